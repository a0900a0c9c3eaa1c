// Ghost refresh of a decomposed rank BY TILES (include/die_hip.h die_pic_ghost_pack / die_pic_ghost_merge; die_amd/dist.py
// DistEnv._refresh_ghosts_tiles).  No reference counterpart (the reference has no distributed code).
//
// The agents of a rank are held in a tile-binned layout (die_pic_layout: per tile a segment of stayers, then of the agents
// that left the tile in the last step) over planes whose halo is a whole number of tiles deep.  Then nothing about a refresh
// needs a look at every agent:
//   * an agent is OWNED iff it stands on an interior tile — the interior tiles' agents are kept, the halo tiles' are dropped;
//   * what a neighbour needs as its ghosts are the agents standing on the interior tiles next to that side (the band), tile
//     by tile: its halo tiles become copies of them (agent coordinates are global: nothing is translated);
//   * the new layout is, tile by tile, either "what stands on this interior tile now" or "what arrived for this halo tile".
// "What stands on tile t" = t's stayers + the leavers of the 8 tiles around it that landed on t — the gather the agent kernel
// of the step does anyway (csrc/die_pic.hip k_pic_forward_move); the counts are known before anything is copied (s + inc).
#include "die_common.h"

#define RF_BLOCK 256
#define RF_SIDES 8
// words of the summary (int64): [0] agents after the refresh, [1] of them owned, [2, 10) sent per side, [10, 18) arrived
// per side, [18] flags
#define RF_SUM_SENT 2
#define RF_SUM_ARRIVED 10
#define RF_SUM_FLAGS 18
#define RF_FLAG_COUNT 1u        // a tile holds another number of agents than its per-tile words say
#define RF_FLAG_SEND_CAP 2u     // a band holds more agents than a message
#define RF_FLAG_RECV 4u         // a received count is impossible (more than the message holds)
#define RF_FLAG_CAPACITY 8u     // the agents do not fit the local arrays

struct RfSide {
    int tx0, ty0, ntx, nty;     // band: tiles of the interior next to this side (what the neighbour gets) …
    int hx0, hy0;               // … halo: same shape, where the message arriving from that neighbour goes
    uint32_t cap;
    uint32_t *send_counts, *send_rec;
    const uint32_t *recv_counts, *recv_rec;
};

struct RfArgs {
    die_geo g;
    int xs, ys, NTX, NTY;
    int ix0, ix1, iy0, iy1;     // interior tiles [ix0, ix1) × [iy0, iy1)
    int n_sides;
    int first[RF_SIDES + 1];    // band tiles of the sides before side k
    die_pic_layout src, dst;
    RfSide side[RF_SIDES];
    long long* summary;
    uint32_t capacity;
    int phase;                  // die_pic_ghost_merge_phase: 0 the whole new layout; 1 its interior tiles (needs nothing that arrives); 2 its halo tiles
    // die_pic_ghost_inplace: the interior tiles' segments stay where the step left them (layout `src` remains the layout the next step
    // READS); only the halo tiles get new segments, behind everything the arrays hold.  tail[t]: where halo tile t's new segment
    // begins (scratch, one word per tile)
    int inplace;
    uint32_t* tail;
};

__device__ __forceinline__ int rf_wrap(int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); }
__device__ __forceinline__ int rf_tile_of(const RfArgs& a, uint32_t X, uint32_t Y) {
    const int r = die_plane_coord(die_cell((int64_t)X, a.g.gW), a.g.ox, a.g.W, a.g.gW);
    const int c = die_plane_coord(die_cell((int64_t)Y, a.g.gH), a.g.oy, a.g.H, a.g.gH);
    return (r >> a.xs) * a.NTY + (c >> a.ys);
}
__device__ __forceinline__ void rf_flag(const RfArgs& a, uint32_t f) { atomicOr((unsigned long long*)&a.summary[RF_SUM_FLAGS], (unsigned long long)f); }

// Σ v(i) over i < n, by the whole workgroup (n is small: the tiles of one side)
template <class F> __device__ __forceinline__ uint32_t rf_block_sum(int n, F v) {
    __shared__ uint32_t s_red[RF_BLOCK / DIE_WAVE];
    uint32_t t = 0;
    for (int i = threadIdx.x; i < n; i += RF_BLOCK) t += v(i);
    t = (uint32_t)die_wave_sum((long long)t);
    __syncthreads();
    if ((threadIdx.x & (DIE_WAVE - 1)) == 0) s_red[threadIdx.x / DIE_WAVE] = t;
    __syncthreads();
    uint32_t r = 0;
    for (int w = 0; w < RF_BLOCK / DIE_WAVE; ++w) r += s_red[w];
    return r;
}

// The agents standing on tile (tx, ty) in layout L → put(i, x, y, agent_food, slot, heading_hi, heading_lo), i = 0 … count − 1
// (stayers first, in their order; arrivals in any order — nothing depends on the order inside a tile: results are keyed by slot).
// Returns the count (uniform).  Three dependent round trips: the 9 tiles' words, then every stayer and every candidate
// arrival (the leavers of the 8 tiles around, as ONE index range) with all six streams in flight, then the stores.
template <class PUT> __device__ __forceinline__ uint32_t rf_gather(const RfArgs& a, const die_pic_layout& L, int tx, int ty, PUT put) {
    __shared__ uint32_t s_cnt, s_base[9], s_pre[10];
    const int t = tx * a.NTY + ty;
    __syncthreads();                                            // (s_* may still be read by a previous call's stragglers)
    if (threadIdx.x < 9) {
        const int k = threadIdx.x, nb = rf_wrap(tx + k / 3 - 1, a.NTX) * a.NTY + rf_wrap(ty + k % 3 - 1, a.NTY);
        uint32_t o = L.off[nb], s = L.s[nb], n = L.n[nb];
        if (s > n) { s = n = 0; rf_flag(a, RF_FLAG_COUNT); }
        // [4] = the tile itself: its stayers; the others: their leavers (a neighbour that IS the tile — fewer than 3 tiles along
        // an axis — cannot occur: die_pic needs 3×3)
        s_base[k] = k == 4 ? o : o + s;
        s_pre[k] = k == 4 ? s : (nb == t ? 0u : n - s);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t own = s_pre[4];
        uint32_t run = 0;
        for (int k = 0; k < 9; ++k) { const uint32_t len = k == 4 ? 0u : s_pre[k]; s_pre[k] = run; run += len; }
        s_pre[9] = run;                                         // candidates; s_pre[k] = first candidate of neighbour k (k = 4: empty)
        s_cnt = own;
    }
    __syncthreads();
    const uint32_t o = s_base[4], own = s_cnt, ncand = s_pre[9];
    // RF_U items per thread and round with ALL their loads in flight before the first store (the kernels here are bound by the
    // chain of dependent round trips of a workgroup, not by bytes: one item per thread and trip made the pack of 560 band tiles
    // 94 µs); a tile of ≈ 600 stayers and ≈ 650 candidates is one round each
    constexpr int RF_U = 3;
    for (uint32_t b = 0; b < own; b += RF_U * RF_BLOCK) {
        uint32_t X[RF_U], Y[RF_U], AF[RF_U], SL[RF_U], HH[RF_U], HL[RF_U];
#pragma unroll
        for (int u = 0; u < RF_U; ++u) {
            const uint32_t i = b + (uint32_t)u * RF_BLOCK + threadIdx.x;
            if (i < own) {
                const uint32_t j = o + i;
                X[u] = L.x[j]; Y[u] = L.y[j]; AF[u] = __float_as_uint(L.agent_food[j]); SL[u] = L.slot[j]; HH[u] = L.heading_hi[j]; HL[u] = L.heading_lo[j];
            }
        }
#pragma unroll
        for (int u = 0; u < RF_U; ++u) {
            const uint32_t i = b + (uint32_t)u * RF_BLOCK + threadIdx.x;
            if (i < own) put(i, X[u], Y[u], AF[u], SL[u], HH[u], HL[u]);
        }
    }
    __syncthreads();                                            // (s_cnt is read above, added to below)
    const int lane = threadIdx.x & (DIE_WAVE - 1);
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t b = 0; b < ncand; b += RF_U * RF_BLOCK) {                              // wave-uniform trip counts throughout
        uint32_t X[RF_U], Y[RF_U], AF[RF_U], SL[RF_U], HH[RF_U], HL[RF_U];
        bool hit[RF_U];
#pragma unroll
        for (int u = 0; u < RF_U; ++u) {
            const uint32_t c = b + (uint32_t)u * RF_BLOCK + threadIdx.x;
            hit[u] = false;
            X[u] = Y[u] = AF[u] = SL[u] = HH[u] = HL[u] = 0;
            if (c < ncand) {
                int k = 0;
                while (k < 8 && c >= s_pre[k + 1]) ++k;
                const uint32_t j = s_base[k] + (c - s_pre[k]);
                X[u] = L.x[j]; Y[u] = L.y[j]; AF[u] = __float_as_uint(L.agent_food[j]); SL[u] = L.slot[j]; HH[u] = L.heading_hi[j]; HL[u] = L.heading_lo[j];
            }
        }
#pragma unroll
        for (int u = 0; u < RF_U; ++u) {
            const uint32_t c = b + (uint32_t)u * RF_BLOCK + threadIdx.x;
            hit[u] = c < ncand && rf_tile_of(a, X[u], Y[u]) == t;
            const unsigned long long m = __ballot(hit[u]);
            if (!m) continue;
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
            at = __shfl(at, 0, DIE_WAVE);
            if (hit[u]) put(at + (uint32_t)__popcll(m & below), X[u], Y[u], AF[u], SL[u], HH[u], HL[u]);
        }
    }
    __syncthreads();
    return s_cnt;
}

// one workgroup per band tile: the tile's agents → the message of its side, behind those of the side's earlier tiles
__global__ __launch_bounds__(RF_BLOCK) void k_pic_ghost_pack(RfArgs a) {
    int k = 0;
    while (k + 1 < a.n_sides && (int)blockIdx.x >= a.first[k + 1]) ++k;
    const RfSide& S = a.side[k];
    const int i = (int)blockIdx.x - a.first[k], ti = i / S.nty, tj = i - ti * S.nty;
    const die_pic_layout& L = a.src;
    auto expect = [&](int q) { const int t = (S.tx0 + q / S.nty) * a.NTY + S.ty0 + q % S.nty; return L.s[t] + L.inc[t]; };
    const uint32_t before = rf_block_sum(i, expect), cap = S.cap;
    uint32_t* rec = S.send_rec;
    const uint32_t c = rf_gather(a, L, S.tx0 + ti, S.ty0 + tj, [&](uint32_t q, uint32_t X, uint32_t Y, uint32_t af, uint32_t sl, uint32_t hh, uint32_t hl) {
        const uint32_t at = before + q;
        if (at >= cap) return;
        rec[at] = X; rec[cap + at] = Y; rec[2 * cap + at] = af; rec[3 * cap + at] = sl; rec[4 * cap + at] = hh; rec[5 * cap + at] = hl;
    });
    if (threadIdx.x == 0) {
        S.send_counts[i] = c;
        if (c != expect(i)) rf_flag(a, RF_FLAG_COUNT);
        if (before + c > cap) rf_flag(a, RF_FLAG_SEND_CAP);
        atomicAdd((unsigned long long*)&a.summary[RF_SUM_SENT + k], (unsigned long long)c);
    }
}

// which side's message fills halo tile (tx, ty), and which of its tiles; −1: an interior tile (or no neighbour there)
__device__ __forceinline__ int rf_halo_side(const RfArgs& a, int tx, int ty, int& i) {
    for (int k = 0; k < a.n_sides; ++k) {
        const RfSide& S = a.side[k];
        if (tx >= S.hx0 && tx < S.hx0 + S.ntx && ty >= S.hy0 && ty < S.hy0 + S.nty) { i = (tx - S.hx0) * S.nty + ty - S.hy0; return k; }
    }
    i = 0;
    return -1;
}

// one workgroup: the new segment sizes and offsets (dst's per-tile words: every agent a stayer), the totals.  The segments of
// the INTERIOR tiles come first (in tile order), those of the halo tiles behind them: an interior tile's offset then depends on
// nothing that arrives from a neighbour, and the step after a refresh can run on the interior while the messages are in flight
// (phase 1: interior tiles; phase 2: halo tiles, from the interior's total in summary[1]; phase 0: both).
__global__ __launch_bounds__(1024) void k_pic_ghost_scan(RfArgs a) {
    __shared__ uint32_t s[1024];
    const int NT = a.NTX * a.NTY, per = (NT + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(lo + per, NT);
    auto count = [&](int t, bool& interior) {
        const int tx = t / a.NTY, ty = t - tx * a.NTY;
        interior = tx >= a.ix0 && tx < a.ix1 && ty >= a.iy0 && ty < a.iy1;
        if (interior) return a.src.s[t] + a.src.inc[t];
        int i;
        const int k = rf_halo_side(a, tx, ty, i);
        if (k < 0) return 0u;
        if (a.phase == 1) return 0u;                            // (nothing has arrived yet)
        const uint32_t c = a.side[k].recv_counts[i];
        if (c > a.side[k].cap) { rf_flag(a, RF_FLAG_RECV); return 0u; }
        return c;
    };
    uint32_t own_total = a.phase == 2 ? (uint32_t)a.summary[1] : 0u;
    for (int pass = 0; pass < 2; ++pass) {                      // 0: interior tiles, 1: halo tiles
        const bool active = pass == 0 ? a.phase != 2 : a.phase != 1;
        uint32_t sum = 0;
        if (active) for (int t = lo; t < hi; ++t) { bool in; const uint32_t c = count(t, in); sum += in == (pass == 0) ? c : 0u; }
        __syncthreads();
        s[threadIdx.x] = sum;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const uint32_t v = (int)threadIdx.x >= o ? s[threadIdx.x - o] : 0u;
            __syncthreads();
            s[threadIdx.x] += v;
            __syncthreads();
        }
        const uint32_t total = s[1023];
        if (active) {
            uint32_t run = (pass == 0 ? 0u : own_total) + s[threadIdx.x] - sum;
            for (int t = lo; t < hi; ++t) {
                bool in;
                const uint32_t c = count(t, in);
                if (in != (pass == 0)) continue;
                // (a segment that would end behind the arrays is published EMPTY — the capacity flag below is raised for it —: with the
                // refresh under the next step, that step's kernels are queued before the host reads the flag and must never index past
                // the arrays' end)
                const bool fits = (unsigned long long)run + c <= (unsigned long long)a.capacity;
                a.dst.off[t] = fits ? run : 0u; a.dst.n[t] = fits ? c : 0u; a.dst.s[t] = fits ? c : 0u; a.dst.inc[t] = 0;
                run += c;
            }
        }
        if (pass == 0 && a.phase != 2) own_total = total;
        if (pass == 1 && active && threadIdx.x == 0) {
            a.summary[0] = (long long)own_total + (long long)total;
            if (own_total + total > a.capacity) rf_flag(a, RF_FLAG_CAPACITY);
        }
    }
    if (a.phase != 2 && threadIdx.x == 0) {
        a.summary[1] = (long long)own_total;
        if (own_total > a.capacity) rf_flag(a, RF_FLAG_CAPACITY);
    }
    if (a.phase != 1 && (int)threadIdx.x < a.n_sides) {
        const RfSide& S = a.side[threadIdx.x];
        long long t = 0;
        for (int i = 0; i < S.ntx * S.nty; ++i) t += (long long)S.recv_counts[i];
        a.summary[RF_SUM_ARRIVED + threadIdx.x] = t;
    }
}

// The scan of the refresh in place (die_pic_ghost_inplace), one workgroup.  The generic scan above walks its tiles one dependent load
// after the other (32 µs for the 4 624 tiles of a rank's planes: a chain of round trips, nothing else); here a thread requests the
// words of all its tiles first.  phase 1: the interior tiles' places in the layout the step writes (dst), summary[1]; phase 2: the halo
// tiles' places behind them (dst), the totals, the arrived counts per side, and tail[t] — where halo tile t's NEW segment of the
// layout the step reads (src) begins: behind everything the arrays hold, room for what arrived + all its old leavers (so the places
// follow from the per-tile words alone).
#define RF_PER 8                // tiles per thread the unrolled loads cover (8 192 tiles; more: the plain loop)
__global__ __launch_bounds__(1024) void k_pic_ghost_scan_inplace(RfArgs a) {
    __shared__ uint32_t s[1024];
    __shared__ unsigned long long s_arr[RF_SIDES];
    const int NT = a.NTX * a.NTY, per = (NT + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(lo + per, NT);
    if (threadIdx.x < RF_SIDES) s_arr[threadIdx.x] = 0ull;
    __syncthreads();
    // this thread's tiles: interior → the agents standing on it (s + inc); halo → what arrived for it, and its old leavers
    uint32_t cnt[RF_PER], room[RF_PER], endv = 0;
    int kind[RF_PER];                                           // 1 interior, 0 halo, −1 none
    auto classify = [&](int t, uint32_t& c, uint32_t& r, uint32_t& e, int& side) {
        const int tx = t / a.NTY, ty = t - tx * a.NTY;
        const bool interior = tx >= a.ix0 && tx < a.ix1 && ty >= a.iy0 && ty < a.iy1;
        c = r = e = 0; side = -1;
        if (interior) { c = a.src.s[t] + a.src.inc[t]; return 1; }
        if (a.phase == 1) return 0;
        int i;
        const int k = rf_halo_side(a, tx, ty, i);
        if (k >= 0) {
            c = a.side[k].recv_counts[i];
            if (c > a.side[k].cap) { rf_flag(a, RF_FLAG_RECV); c = 0; }
            side = k;
        }
        const uint32_t o_ = a.src.off[t], s_ = a.src.s[t], n_ = a.src.n[t];
        r = c + (s_ > n_ ? 0u : n_ - s_);
        e = o_ + n_;
        return 0;
    };
    uint32_t own_sum = 0, halo_sum = 0, room_sum = 0;
    if (per <= RF_PER) {
#pragma unroll
        for (int q = 0; q < RF_PER; ++q) {
            const int t = lo + q;
            kind[q] = -1; cnt[q] = room[q] = 0;
            if (q < per && t < hi) {
                uint32_t e; int side;
                kind[q] = classify(t, cnt[q], room[q], e, side);
                if (kind[q] == 1 && a.phase == 2) endv = max(endv, a.src.off[t] + a.src.n[t]);
                endv = max(endv, e);
                if (side >= 0) atomicAdd(&s_arr[side], (unsigned long long)cnt[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < RF_PER; ++q) { if (kind[q] == 1) own_sum += cnt[q]; else if (kind[q] == 0) { halo_sum += cnt[q]; room_sum += room[q]; } }
    } else {
        for (int t = lo; t < hi; ++t) {
            uint32_t c, r, e; int side;
            const int kd = classify(t, c, r, e, side);
            if (kd == 1 && a.phase == 2) endv = max(endv, a.src.off[t] + a.src.n[t]);
            endv = max(endv, e);
            if (side >= 0) atomicAdd(&s_arr[side], (unsigned long long)c);
            if (kd == 1) own_sum += c; else { halo_sum += c; room_sum += r; }
        }
    }
    // inclusive scan over the 1024 threads: shuffles inside a wave, the 16 waves' totals through LDS (two barriers; the textbook
    // ten-step form with two barriers per step was most of this kernel's time)
    const int lane = threadIdx.x & (DIE_WAVE - 1), wave = threadIdx.x / DIE_WAVE;
    auto block_scan = [&](uint32_t v, uint32_t& total) {
        uint32_t x = v;
#pragma unroll
        for (int o = 1; o < DIE_WAVE; o <<= 1) { const uint32_t u = __shfl_up(x, o, DIE_WAVE); if (lane >= o) x += u; }
        __syncthreads();                                        // (s may still be read by the previous call's stragglers)
        if (lane == DIE_WAVE - 1) s[wave] = x;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 1024 / DIE_WAVE; ++w) { const uint32_t t_ = s[w]; all += t_; if (w < wave) before += t_; }
        total = all;
        return x + before;
    };
    auto each = [&](auto f) {                                   // f(q-th tile of this thread, kind, count, room)
        if (per <= RF_PER) {
#pragma unroll
            for (int q = 0; q < RF_PER; ++q) if (kind[q] >= 0) f(lo + q, kind[q], cnt[q], room[q]);
        } else {
            for (int t = lo; t < hi; ++t) { uint32_t c, r, e; int side; const int kd = classify(t, c, r, e, side); f(t, kd, c, r); }
        }
    };
    if (a.phase == 1) {
        uint32_t own_total;
        uint32_t run = block_scan(own_sum, own_total) - own_sum;
        each([&](int t, int kd, uint32_t c, uint32_t) {
            if (kd != 1) return;
            const bool fits = (unsigned long long)run + c <= (unsigned long long)a.capacity;
            a.dst.off[t] = fits ? run : 0u; a.dst.n[t] = fits ? c : 0u; a.dst.s[t] = fits ? c : 0u; a.dst.inc[t] = 0;
            run += c;
        });
        if (threadIdx.x == 0) {
            a.summary[1] = (long long)own_total;
            if (own_total > a.capacity) rf_flag(a, RF_FLAG_CAPACITY);
        }
        return;
    }
    // phase 2
    const uint32_t own_total = (uint32_t)a.summary[1];
    uint32_t halo_total, room_total;
    uint32_t run = own_total + block_scan(halo_sum, halo_total) - halo_sum;
    each([&](int t, int kd, uint32_t c, uint32_t) {
        if (kd != 0) return;
        const bool fits = (unsigned long long)run + c <= (unsigned long long)a.capacity;        // (else EMPTY: see k_pic_ghost_scan)
        a.dst.off[t] = fits ? run : 0u; a.dst.n[t] = fits ? c : 0u; a.dst.s[t] = fits ? c : 0u; a.dst.inc[t] = 0;
        run += c;
    });
    const uint32_t room_incl = block_scan(room_sum, room_total);
    uint32_t e_ = endv;
#pragma unroll
    for (int o = DIE_WAVE / 2; o > 0; o >>= 1) e_ = max(e_, (uint32_t)__shfl_xor((int)e_, o, DIE_WAVE));
    __syncthreads();
    if (lane == 0) s[wave] = e_;
    __syncthreads();
    uint32_t end_old = 0;
#pragma unroll
    for (int w = 0; w < 1024 / DIE_WAVE; ++w) end_old = max(end_old, s[w]);
    uint32_t at = end_old + room_incl - room_sum;
    each([&](int t, int kd, uint32_t, uint32_t r) {
        if (kd != 0) return;
        a.tail[t] = at;
        at += r;
    });
    if (threadIdx.x == 0) {
        a.summary[0] = (long long)own_total + (long long)halo_total;
        if (own_total + halo_total > a.capacity || (unsigned long long)end_old + room_total > a.capacity) rf_flag(a, RF_FLAG_CAPACITY);
    }
    if ((int)threadIdx.x < a.n_sides) a.summary[RF_SUM_ARRIVED + threadIdx.x] = (long long)s_arr[threadIdx.x];
}

// in place (die_pic_ghost_inplace, phase 2), one workgroup per HALO tile: its new segment of layout src, at tail[t] — what arrived
// for it as stayers, then its old leavers that stand on an interior tile as leavers (the agent kernel of the interior tile next to it
// picks them up as arrivals, as after any step) — and its three words.  The interior tiles' leavers that stand on a halo tile stay
// where they are and are nobody's: the agent kernel of a halo tile takes no arrivals in the step behind a refresh (die_pic.halo_fresh)
// — the copy that counts arrived with the message.
__global__ __launch_bounds__(RF_BLOCK) void k_pic_ghost_halo(RfArgs a) {
    const int tx = blockIdx.y, ty = blockIdx.x, t = tx * a.NTY + ty;
    if (tx >= a.ix0 && tx < a.ix1 && ty >= a.iy0 && ty < a.iy1) return;
    const die_pic_layout& L = a.src;
    __shared__ uint32_t s_keep;
    if (threadIdx.x == 0) s_keep = 0;
    uint32_t o_old = L.off[t], s_old = L.s[t], n_old = L.n[t];
    if (s_old > n_old) { s_old = n_old = 0; if (threadIdx.x == 0) rf_flag(a, RF_FLAG_COUNT); }
    const uint32_t base = a.tail[t], capacity = a.capacity;
    int i;
    const int k = rf_halo_side(a, tx, ty, i);
    uint32_t c = 0;
    // A new segment that would end behind the arrays is not laid at all: the tile is left EMPTY (off = s = n = 0), so that the step's
    // kernels — which are already queued behind this one when the host reads the flag — never index past the arrays' end.  (Round 5's
    // development build bounded the stores below but still published off = tail[t], s, n for such a tile: the agent kernel of the step
    // that followed then read x[tail[t] + i] beyond the allocation — the memory fault of gpurun_out/r5_t8.log)
    {
        uint32_t c_in = 0;
        if (k >= 0) { c_in = a.side[k].recv_counts[i]; if (c_in > a.side[k].cap) c_in = 0; }
        if ((unsigned long long)base + c_in + (n_old - s_old) > (unsigned long long)capacity) {
            if (threadIdx.x == 0) { rf_flag(a, RF_FLAG_CAPACITY); L.off[t] = 0; L.s[t] = 0; L.n[t] = 0; }
            return;
        }
    }
    if (k >= 0) {
        const RfSide& S = a.side[k];
        const uint32_t* counts = S.recv_counts;
        c = counts[i];
        if (c > S.cap) c = 0;                                   // (flagged by the scan)
        const uint32_t before = rf_block_sum(i, [&](int q) { return counts[q]; }), cap = S.cap;
        const uint32_t* rec = S.recv_rec;
        constexpr int U = 3;                                    // (all loads of a round before its first store: see rf_gather)
        for (uint32_t b = 0; b < c; b += U * RF_BLOCK) {
            uint32_t v[U][6];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t q = b + (uint32_t)u * RF_BLOCK + threadIdx.x, from = before + q, at = base + q;
                ok[u] = q < c;
                if (ok[u] && (from >= cap || at >= capacity)) { rf_flag(a, from >= cap ? RF_FLAG_RECV : RF_FLAG_CAPACITY); ok[u] = false; }
                if (ok[u]) for (int w = 0; w < 6; ++w) v[u][w] = rec[(uint32_t)w * cap + from];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t at = base + b + (uint32_t)u * RF_BLOCK + threadIdx.x;
                if (!ok[u]) continue;
                L.x[at] = v[u][0]; L.y[at] = v[u][1]; L.agent_food[at] = __uint_as_float(v[u][2]);
                L.slot[at] = v[u][3]; L.heading_hi[at] = v[u][4]; L.heading_lo[at] = v[u][5];
            }
        }
    }
    __syncthreads();
    // the old leavers that stand on an interior tile
    const int lane = threadIdx.x & (DIE_WAVE - 1);
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t b = s_old; b < n_old; b += RF_BLOCK) {                // wave-uniform trip count
        const uint32_t j = o_old + b + threadIdx.x;
        const bool in_range = b + threadIdx.x < n_old;
        uint32_t X = 0, Y = 0, AF = 0, SL = 0, HH = 0, HL = 0;
        if (in_range) { X = L.x[j]; Y = L.y[j]; AF = __float_as_uint(L.agent_food[j]); SL = L.slot[j]; HH = L.heading_hi[j]; HL = L.heading_lo[j]; }
        bool keep = false;
        if (in_range) {
            const int tt = rf_tile_of(a, X, Y), ttx = tt / a.NTY, tty = tt - ttx * a.NTY;
            keep = ttx >= a.ix0 && ttx < a.ix1 && tty >= a.iy0 && tty < a.iy1;
        }
        const unsigned long long m = __ballot(keep);
        if (!m) continue;
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(&s_keep, (uint32_t)__popcll(m));
        at = __shfl(at, 0, DIE_WAVE);
        if (keep) {
            const uint32_t q = base + c + at + (uint32_t)__popcll(m & below);
            if (q < capacity) { L.x[q] = X; L.y[q] = Y; L.agent_food[q] = __uint_as_float(AF); L.slot[q] = SL; L.heading_hi[q] = HH; L.heading_lo[q] = HL; }
            else rf_flag(a, RF_FLAG_CAPACITY);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { L.off[t] = base; L.s[t] = c; L.n[t] = c + s_keep; }
}

// one workgroup per tile of the planes: its segment of the new layout
__global__ __launch_bounds__(RF_BLOCK) void k_pic_ghost_merge(RfArgs a) {
    const int tx = blockIdx.y, ty = blockIdx.x, t = tx * a.NTY + ty;
    const uint32_t base = a.dst.off[t], c = a.dst.n[t], capacity = a.capacity;
    const die_pic_layout& D = a.dst;
    const bool interior = tx >= a.ix0 && tx < a.ix1 && ty >= a.iy0 && ty < a.iy1;
    if (interior ? a.phase == 2 : a.phase == 1) return;
    if (interior) {
        const uint32_t got = rf_gather(a, a.src, tx, ty, [&](uint32_t q, uint32_t X, uint32_t Y, uint32_t af, uint32_t sl, uint32_t hh, uint32_t hl) {
            const uint32_t at = base + q;
            if (q >= c || at >= capacity) return;
            D.x[at] = X; D.y[at] = Y; D.agent_food[at] = __uint_as_float(af); D.slot[at] = sl; D.heading_hi[at] = hh; D.heading_lo[at] = hl;
        });
        if (threadIdx.x == 0 && got != c) rf_flag(a, RF_FLAG_COUNT);
        return;
    }
    int i;
    const int k = rf_halo_side(a, tx, ty, i);
    if (k < 0 || c == 0) return;
    const RfSide& S = a.side[k];
    const uint32_t* counts = S.recv_counts;
    const uint32_t before = rf_block_sum(i, [&](int q) { return counts[q]; }), cap = S.cap;
    const uint32_t* rec = S.recv_rec;
    constexpr int U = 3;                                        // (all loads of a round before its first store: see rf_gather)
    for (uint32_t b = 0; b < c; b += U * RF_BLOCK) {
        uint32_t v[U][6];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t q = b + (uint32_t)u * RF_BLOCK + threadIdx.x, from = before + q, at = base + q;
            ok[u] = q < c;
            if (ok[u] && (from >= cap || at >= capacity)) { rf_flag(a, from >= cap ? RF_FLAG_RECV : RF_FLAG_CAPACITY); ok[u] = false; }
            if (ok[u]) for (int w = 0; w < 6; ++w) v[u][w] = rec[(uint32_t)w * cap + from];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t at = base + b + (uint32_t)u * RF_BLOCK + threadIdx.x;
            if (!ok[u]) continue;
            D.x[at] = v[u][0]; D.y[at] = v[u][1]; D.agent_food[at] = __uint_as_float(v[u][2]);
            D.slot[at] = v[u][3]; D.heading_hi[at] = v[u][4]; D.heading_lo[at] = v[u][5];
        }
    }
}

// the layout the next step WRITES gets the same segments (die_pic_bin leaves the two layouts in the same state)
__global__ __launch_bounds__(RF_BLOCK) void k_pic_ghost_words(RfArgs a) {
    const int t = blockIdx.x * RF_BLOCK + threadIdx.x;
    if (t >= a.NTX * a.NTY) return;
    if (a.phase) {
        const int tx = t / a.NTY, ty = t - tx * a.NTY;
        const bool interior = tx >= a.ix0 && tx < a.ix1 && ty >= a.iy0 && ty < a.iy1;
        if (interior ? a.phase == 2 : a.phase == 1) return;
    }
    const uint32_t c = a.dst.n[t];
    a.src.off[t] = a.dst.off[t]; a.src.n[t] = c; a.src.s[t] = c; a.src.inc[t] = 0;
}

static int rf_fill(RfArgs& a, const die_medium* m, const die_pic* p, int32_t from, int32_t n_sides, const die_pic_side* sides,
                   int64_t* summary, const char* who) {
    DIE_REQUIRE(m && p && summary && (from == 0 || from == 1) && n_sides >= 0 && n_sides <= RF_SIDES && (sides || n_sides == 0), "%s: bad argument", who);
    DIE_REQUIRE(m->gW > 0 && m->own_x1 > 0, "%s: the planes must be a tile of a decomposed world with its owned cells set (die_medium.gW, own_*)", who);
    const int xs = p->tile_xs, ys = p->tile_ys, TX = 1 << xs, TY = 1 << ys;
    DIE_REQUIRE(xs >= 2 && xs <= 8 && ys >= 2 && ys <= 8 && m->W % TX == 0 && m->H % TY == 0 && m->W / TX >= 3 && m->H / TY >= 3,
                "%s: the planes do not split into at least 3x3 whole tiles", who);
    DIE_REQUIRE(m->own_x0 % TX == 0 && m->own_x1 % TX == 0 && m->own_y0 % TY == 0 && m->own_y1 % TY == 0,
                "%s: the owned cells [%d, %d) x [%d, %d) are not whole %dx%d tiles", who, m->own_x0, m->own_x1, m->own_y0, m->own_y1, TX, TY);
    a.g = die_geo_of(m);
    a.xs = xs; a.ys = ys; a.NTX = m->W >> xs; a.NTY = m->H >> ys;
    a.ix0 = m->own_x0 >> xs; a.ix1 = m->own_x1 >> xs; a.iy0 = m->own_y0 >> ys; a.iy1 = m->own_y1 >> ys;
    a.n_sides = n_sides;
    a.src = p->layout[from]; a.dst = p->layout[1 - from];
    for (int l = 0; l < 2; ++l) {
        const die_pic_layout& L = p->layout[l];
        DIE_REQUIRE(L.x && L.y && L.agent_food && L.slot && L.heading_hi && L.heading_lo && L.off && L.n && L.s && L.inc, "%s: null pointer in layout %d", who, l);
    }
    DIE_REQUIRE(a.src.x != a.dst.x && a.src.off != a.dst.off, "%s: the two layouts must be different arrays", who);
    a.first[0] = 0;
    for (int k = 0; k < n_sides; ++k) {
        const die_pic_side& S = sides[k];
        DIE_REQUIRE(S.ntx > 0 && S.nty > 0 && S.cap > 0 && S.cap < ((int64_t)1 << 31) && S.send_counts && S.send_rec && S.recv_counts && S.recv_rec,
                    "%s: side %d: bad shape or null pointer", who, k);
        DIE_REQUIRE(S.tx0 >= a.ix0 && S.tx0 + S.ntx <= a.ix1 && S.ty0 >= a.iy0 && S.ty0 + S.nty <= a.iy1, "%s: side %d: the band must lie inside the interior tiles", who, k);
        DIE_REQUIRE(S.hx0 >= 0 && S.hx0 + S.ntx <= a.NTX && S.hy0 >= 0 && S.hy0 + S.nty <= a.NTY &&
                    (S.hx0 + S.ntx <= a.ix0 || S.hx0 >= a.ix1 || S.hy0 + S.nty <= a.iy0 || S.hy0 >= a.iy1), "%s: side %d: the halo block must lie outside the interior tiles", who, k);
        RfSide& R = a.side[k];
        R.tx0 = S.tx0; R.ty0 = S.ty0; R.ntx = S.ntx; R.nty = S.nty; R.hx0 = S.hx0; R.hy0 = S.hy0; R.cap = (uint32_t)S.cap;
        R.send_counts = S.send_counts; R.send_rec = S.send_rec; R.recv_counts = S.recv_counts; R.recv_rec = S.recv_rec;
        a.first[k + 1] = a.first[k] + S.ntx * S.nty;
    }
    for (int k = n_sides; k < RF_SIDES; ++k) { a.side[k] = RfSide{}; a.first[k + 1] = a.first[n_sides]; }
    a.summary = (long long*)summary;
    a.capacity = 0;
    a.phase = 0;
    a.inplace = 0; a.tail = nullptr;
    return DIE_OK;
}

extern "C" int die_pic_ghost_pack(const die_medium* m, const die_pic* p, int32_t from, int32_t n_sides, const die_pic_side* sides,
                                  int64_t* summary, void* stream) {
    RfArgs a;
    const int rc = rf_fill(a, m, p, from, n_sides, sides, summary, "die_pic_ghost_pack");
    if (rc != DIE_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(summary, 0, (size_t)DIE_PIC_GHOST_SUMMARY_WORDS * 8, s);
    if (e != hipSuccess) { die_set_error("die_pic_ghost_pack: memset failed: %s", hipGetErrorString(e)); return DIE_ERR_HIP; }
    if (a.first[n_sides] > 0) k_pic_ghost_pack<<<a.first[n_sides], RF_BLOCK, 0, s>>>(a);
    DIE_CHECK_LAUNCH("die_pic_ghost_pack");
    return DIE_OK;
}

extern "C" int die_pic_ghost_merge(const die_medium* m, const die_pic* p, int32_t from, int32_t n_sides, const die_pic_side* sides,
                                   int64_t capacity, int64_t* summary, void* stream) {
    return die_pic_ghost_merge_phase(m, p, from, n_sides, sides, capacity, summary, 0, stream);
}

extern "C" int die_pic_ghost_merge_phase(const die_medium* m, const die_pic* p, int32_t from, int32_t n_sides, const die_pic_side* sides,
                                         int64_t capacity, int64_t* summary, int32_t phase, void* stream) {
    RfArgs a;
    const int rc = rf_fill(a, m, p, from, n_sides, sides, summary, "die_pic_ghost_merge");
    if (rc != DIE_OK) return rc;
    DIE_REQUIRE(capacity > 0 && capacity < ((int64_t)1 << 31), "die_pic_ghost_merge: capacity %lld", (long long)capacity);
    DIE_REQUIRE(phase >= 0 && phase <= 2, "die_pic_ghost_merge: phase %d", phase);
    a.capacity = (uint32_t)capacity;
    a.phase = phase;
    hipStream_t s = (hipStream_t)stream;
    k_pic_ghost_scan<<<1, 1024, 0, s>>>(a);
    k_pic_ghost_merge<<<dim3(a.NTY, a.NTX), RF_BLOCK, 0, s>>>(a);
    k_pic_ghost_words<<<(a.NTX * a.NTY + RF_BLOCK - 1) / RF_BLOCK, RF_BLOCK, 0, s>>>(a);
    DIE_CHECK_LAUNCH("die_pic_ghost_merge");
    return DIE_OK;
}

extern "C" int die_pic_ghost_inplace(const die_medium* m, const die_pic* p, int32_t from, int32_t n_sides, const die_pic_side* sides,
                                     int64_t capacity, int64_t* summary, int32_t phase, uint32_t* tail, void* stream) {
    RfArgs a;
    const int rc = rf_fill(a, m, p, from, n_sides, sides, summary, "die_pic_ghost_inplace");
    if (rc != DIE_OK) return rc;
    DIE_REQUIRE(capacity > 0 && capacity < ((int64_t)1 << 31), "die_pic_ghost_inplace: capacity %lld", (long long)capacity);
    DIE_REQUIRE((phase == 1 || phase == 2) && tail, "die_pic_ghost_inplace: phase %d / null scratch", phase);
    a.capacity = (uint32_t)capacity;
    a.phase = phase;
    a.inplace = 1; a.tail = tail;
    hipStream_t s = (hipStream_t)stream;
    // phase 1: the places of the interior tiles' segments in the layout the step WRITES (dst) — the step can start on the interior;
    // phase 2: those of the halo tiles behind them, and the halo tiles' new segments of the layout it reads (src), in place
    k_pic_ghost_scan_inplace<<<1, 1024, 0, s>>>(a);
    if (phase == 2) k_pic_ghost_halo<<<dim3(a.NTY, a.NTX), RF_BLOCK, 0, s>>>(a);
    DIE_CHECK_LAUNCH("die_pic_ghost_inplace");
    return DIE_OK;
}
