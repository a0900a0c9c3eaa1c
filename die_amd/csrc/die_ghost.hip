// Ghost-agent refresh of the decomposed step (die_amd/dist.py), device side.
//
// Every M steps a rank re-seats its agent set: agents standing on interior cells stay (they are "owned" from
// now on, whoever owned them before), everything else is dropped ("holes"), and copies of the owned agents
// standing within the halo depth of a side go to the neighbour on that side as its new ghosts.
//   die_ghost_plan          classifies every local agent and writes, in ascending index order (deterministic:
//                           count per block → scan over blocks → fill), one index list per side and the hole list;
//   die_records_gather_dev  packs the records of one list straight into its message (count read on the device);
//   die_records_scatter_at  writes arrived records into the holes / the tail.
// No reference counterpart (the reference is a single-process numpy program).
#include "die_common.h"

#define GH_MAXD 8
#define GH_LISTS (GH_MAXD + 2)          // sides…, holes, owned (count only)
#define GH_BLOCKS 1024
#define GH_ARR_MAX 16

struct GhostArgs {
    die_geo g;
    int64_t N;
    const uint32_t* x;
    const uint32_t* y;
    int Wi, Hi, hx, hy, x0, y0;         // interior size, halo depth, world cell of the interior's origin
    int nd;
    int dx[GH_MAXD], dy[GH_MAXD];
    uint16_t* mask;                     // N membership words
    int32_t* blk;                       // GH_LISTS × GH_BLOCKS: counts, then exclusive bases
    int32_t* lists[GH_MAXD + 1];
    int64_t caps[GH_MAXD + 1];
    int64_t* totals;                    // nd + 2: per side, holes, owned
    int64_t chunk;                      // agents per block (multiple of the block size)
};

__device__ __forceinline__ uint32_t ghost_mask(const GhostArgs& a, int64_t n) {
    const int cx = die_cell((int64_t)a.x[n], a.g.gW), cy = die_cell((int64_t)a.y[n], a.g.gH);
    int lx = cx - a.x0, ly = cy - a.y0;                        // offset from the interior's origin, periodic
    lx = lx < 0 ? lx + a.g.gW : lx;
    ly = ly < 0 ? ly + a.g.gH : ly;
    if (!(lx < a.Wi && ly < a.Hi)) return 1u << GH_MAXD;       // not on an interior cell: a hole
    uint32_t m = 1u << (GH_MAXD + 1);                          // owned
    const bool lox = lx < a.hx, hix = lx >= a.Wi - a.hx, loy = ly < a.hy, hiy = ly >= a.Hi - a.hy;
    for (int k = 0; k < a.nd; ++k) {
        const bool inx = a.dx[k] == 0 || (a.dx[k] < 0 ? lox : hix);
        const bool iny = a.dy[k] == 0 || (a.dy[k] < 0 ? loy : hiy);
        if (inx && iny) m |= 1u << k;
    }
    return m;
}

__global__ __launch_bounds__(DIE_BLOCK) void k_ghost_count(GhostArgs a) {
    __shared__ int s_cnt[GH_LISTS];
    if (threadIdx.x < GH_LISTS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    int cnt[GH_LISTS];
#pragma unroll
    for (int l = 0; l < GH_LISTS; ++l) cnt[l] = 0;
    const int64_t lo = (int64_t)blockIdx.x * a.chunk, hi = min(lo + a.chunk, a.N);
    for (int64_t n = lo + threadIdx.x; n < hi; n += DIE_BLOCK) {
        const uint32_t m = ghost_mask(a, n);
        a.mask[n] = (uint16_t)m;
#pragma unroll
        for (int l = 0; l < GH_LISTS; ++l) cnt[l] += (m >> l) & 1u;
    }
#pragma unroll
    for (int l = 0; l < GH_LISTS; ++l) {
        int c = cnt[l];
        for (int o = DIE_WAVE / 2; o > 0; o >>= 1) c += __shfl_down(c, o, DIE_WAVE);
        if ((threadIdx.x & (DIE_WAVE - 1)) == 0 && c) atomicAdd(&s_cnt[l], c);   // integer adds: order-independent
    }
    __syncthreads();
    if (threadIdx.x < GH_LISTS) a.blk[threadIdx.x * GH_BLOCKS + blockIdx.x] = s_cnt[threadIdx.x];
}

__global__ __launch_bounds__(GH_BLOCKS) void k_ghost_scan(GhostArgs a) {
    __shared__ int s[GH_BLOCKS];
    for (int l = 0; l < GH_LISTS; ++l) {
        const int v = a.blk[l * GH_BLOCKS + threadIdx.x];
        s[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < GH_BLOCKS; o <<= 1) {                     // Hillis–Steele inclusive scan
            const int t = (int)threadIdx.x >= o ? s[threadIdx.x - o] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        a.blk[l * GH_BLOCKS + threadIdx.x] = s[threadIdx.x] - v;     // exclusive base of this block
        if (threadIdx.x == GH_BLOCKS - 1) {
            const int out = l < GH_MAXD ? (l < a.nd ? l : -1) : a.nd + (l - GH_MAXD);
            if (out >= 0) a.totals[out] = s[threadIdx.x];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(DIE_BLOCK) void k_ghost_fill(GhostArgs a) {
    constexpr int NW = DIE_BLOCK / DIE_WAVE;
    __shared__ int s_wave[GH_MAXD + 1][NW];
    __shared__ int s_base[GH_MAXD + 1];
    if (threadIdx.x <= GH_MAXD) s_base[threadIdx.x] = a.blk[threadIdx.x * GH_BLOCKS + blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & (DIE_WAVE - 1), wv = threadIdx.x / DIE_WAVE;
    const int64_t lo = (int64_t)blockIdx.x * a.chunk, hi = min(lo + a.chunk, a.N);
    for (int64_t t0 = lo; t0 < hi; t0 += DIE_BLOCK) {                 // tiles in ascending order
        const int64_t n = t0 + threadIdx.x;
        const uint32_t m = n < hi ? a.mask[n] : 0u;
        int rank[GH_MAXD + 1];
#pragma unroll
        for (int l = 0; l <= GH_MAXD; ++l) {
            const unsigned long long b = __ballot((m >> l) & 1u);
            rank[l] = __popcll(b & ((1ull << lane) - 1ull));
            if (lane == 0) s_wave[l][wv] = __popcll(b);
        }
        __syncthreads();
#pragma unroll
        for (int l = 0; l <= GH_MAXD; ++l) {
            if ((m >> l) & 1u) {
                int off = s_base[l] + rank[l];
                for (int w = 0; w < wv; ++w) off += s_wave[l][w];
                const int li = l < GH_MAXD ? l : a.nd;                // list index: sides, then holes
                if (l == GH_MAXD || l < a.nd) { if ((int64_t)off < a.caps[li]) a.lists[li][off] = (int32_t)n; }
            }
        }
        __syncthreads();
        if (threadIdx.x <= GH_MAXD) {
            int add = 0;
            for (int w = 0; w < NW; ++w) add += s_wave[threadIdx.x][w];
            s_base[threadIdx.x] += add;
        }
        __syncthreads();
    }
}

extern "C" int64_t die_ghost_workspace_bytes(int64_t N) {
    if (N < 0) return -1;
    return ((N * 2 + 255) & ~(int64_t)255) + (int64_t)GH_LISTS * GH_BLOCKS * 4;
}

extern "C" int die_ghost_plan(const die_medium* m, const die_agents* a, int32_t n_dirs, const int8_t* dirs,
                              int32_t* const* lists, const int64_t* caps, int64_t* totals, void* ws, int64_t ws_bytes,
                              void* stream) {
    DIE_REQUIRE(m && a && lists && caps && totals && ws, "die_ghost_plan: null argument");
    DIE_REQUIRE(m->gW > 0 && m->own_x1 > m->own_x0 && m->own_y1 > m->own_y0, "die_ghost_plan: the medium is not a ghost-agent tile");
    DIE_REQUIRE(n_dirs >= 0 && n_dirs <= GH_MAXD && (n_dirs == 0 || dirs), "die_ghost_plan: 0..%d sides", GH_MAXD);
    DIE_REQUIRE(a->N >= 0 && a->N < (int64_t)1 << 31 && (a->N == 0 || (a->x && a->y)), "die_ghost_plan: bad agent arrays");
    DIE_REQUIRE(ws_bytes >= die_ghost_workspace_bytes(a->N), "die_ghost_plan: workspace too small");
    GhostArgs k;
    k.g = die_geo_of(m); k.N = a->N; k.x = a->x; k.y = a->y;
    k.hx = m->own_x0; k.hy = m->own_y0; k.Wi = m->own_x1 - m->own_x0; k.Hi = m->own_y1 - m->own_y0;
    k.x0 = m->ox + m->own_x0; k.y0 = m->oy + m->own_y0;
    k.x0 = ((k.x0 % m->gW) + m->gW) % m->gW; k.y0 = ((k.y0 % m->gH) + m->gH) % m->gH;
    k.nd = n_dirs;
    for (int i = 0; i < GH_MAXD; ++i) {
        k.dx[i] = i < n_dirs ? dirs[2 * i] : 0; k.dy[i] = i < n_dirs ? dirs[2 * i + 1] : 0;
        DIE_REQUIRE(i >= n_dirs || ((k.dx[i] || k.dy[i]) && k.dx[i] >= -1 && k.dx[i] <= 1 && k.dy[i] >= -1 && k.dy[i] <= 1),
                    "die_ghost_plan: bad side %d", i);
    }
    for (int i = 0; i <= GH_MAXD; ++i) {
        k.lists[i] = i <= n_dirs ? lists[i] : nullptr; k.caps[i] = i <= n_dirs ? caps[i] : 0;
        DIE_REQUIRE(i > n_dirs || (lists[i] && caps[i] > 0), "die_ghost_plan: null list %d", i);
    }
    k.mask = (uint16_t*)ws;
    k.blk = (int32_t*)((char*)ws + ((a->N * 2 + 255) & ~(int64_t)255));
    k.totals = totals;
    k.chunk = ((a->N + GH_BLOCKS - 1) / GH_BLOCKS + DIE_BLOCK - 1) / DIE_BLOCK * DIE_BLOCK;
    if (k.chunk < DIE_BLOCK) k.chunk = DIE_BLOCK;
    hipStream_t s = (hipStream_t)stream;
    k_ghost_count<<<GH_BLOCKS, DIE_BLOCK, 0, s>>>(k);
    k_ghost_scan<<<1, GH_BLOCKS, 0, s>>>(k);
    k_ghost_fill<<<GH_BLOCKS, DIE_BLOCK, 0, s>>>(k);
    DIE_CHECK_LAUNCH("die_ghost_plan");
    return DIE_OK;
}

// ---- records with device-side counts / 32-bit index lists ---------------------------------------------------
struct RecDevArgs {
    int n;
    char* arr[GH_ARR_MAX];
    int esz[GH_ARR_MAX];
    const int32_t* idx;
    const int64_t* count_dev;   // gather: number of valid indices (device)
    int64_t count, pitch;       // scatter: host count; pitch = row length of the record matrix
    int32_t* rec;
    int64_t* header;            // gather: receives the true count (may exceed pitch: the receiver checks)
};

template <bool GATHER>
__global__ __launch_bounds__(DIE_BLOCK) void k_records_dev(RecDevArgs a) {
    int64_t count = a.count;
    if (GATHER) {
        const int64_t c = *a.count_dev;
        if (blockIdx.x == 0 && threadIdx.x == 0 && a.header) *a.header = c;
        count = c < a.pitch ? c : a.pitch;
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += stride) {
        const int64_t s = a.idx[j];
        for (int k = 0; k < a.n; ++k) {
            int32_t* r = a.rec + (int64_t)k * a.pitch + j;
            if (a.esz[k] == 4) { if (GATHER) *r = ((const int32_t*)a.arr[k])[s]; else ((int32_t*)a.arr[k])[s] = *r; }
            else { if (GATHER) *r = (int32_t)((const uint8_t*)a.arr[k])[s]; else ((uint8_t*)a.arr[k])[s] = (uint8_t)*r; }
        }
    }
}

static int fill_rec(RecDevArgs& k, void* const* arrays, const int32_t* esz, int32_t n, const char* who) {
    DIE_REQUIRE(arrays && esz && n >= 1 && n <= GH_ARR_MAX, "%s: 1..%d arrays", who, GH_ARR_MAX);
    k.n = n;
    for (int i = 0; i < GH_ARR_MAX; ++i) {
        k.arr[i] = i < n ? (char*)arrays[i] : nullptr;
        k.esz[i] = i < n ? esz[i] : 0;
        DIE_REQUIRE(i >= n || (arrays[i] && (esz[i] == 4 || esz[i] == 1)), "%s: array %d must be 4- or 1-byte", who, i);
    }
    return DIE_OK;
}

extern "C" int die_records_gather_dev(void* const* arrays, const int32_t* elem_bytes, int32_t n, const int32_t* idx,
                                      const int64_t* count_dev, int64_t cap, int32_t* records_out, int64_t* header_out,
                                      void* stream) {
    RecDevArgs k;
    int rc = fill_rec(k, arrays, elem_bytes, n, "die_records_gather_dev");
    if (rc != DIE_OK) return rc;
    DIE_REQUIRE(idx && count_dev && records_out && cap > 0, "die_records_gather_dev: null argument");
    k.idx = idx; k.count_dev = count_dev; k.count = 0; k.pitch = cap; k.rec = records_out; k.header = header_out;
    int64_t g = (cap + DIE_BLOCK - 1) / DIE_BLOCK;
    k_records_dev<true><<<(int)(g < 1024 ? g : 1024), DIE_BLOCK, 0, (hipStream_t)stream>>>(k);
    DIE_CHECK_LAUNCH("die_records_gather_dev");
    return DIE_OK;
}

extern "C" int die_records_scatter_at(void* const* arrays, const int32_t* elem_bytes, int32_t n, const int32_t* idx,
                                      int64_t count, int64_t pitch, const int32_t* records_in, void* stream) {
    if (count == 0) return DIE_OK;
    RecDevArgs k;
    int rc = fill_rec(k, arrays, elem_bytes, n, "die_records_scatter_at");
    if (rc != DIE_OK) return rc;
    DIE_REQUIRE(idx && records_in && count > 0 && pitch >= count, "die_records_scatter_at: bad arguments");
    k.idx = idx; k.count_dev = nullptr; k.count = count; k.pitch = pitch; k.rec = (int32_t*)records_in; k.header = nullptr;
    int64_t g = (count + DIE_BLOCK - 1) / DIE_BLOCK;
    k_records_dev<false><<<(int)(g < 1024 ? g : 1024), DIE_BLOCK, 0, (hipStream_t)stream>>>(k);
    DIE_CHECK_LAUNCH("die_records_scatter_at");
    return DIE_OK;
}

// ---- all sides in one launch; arrivals and compaction entirely on the device ------------------------------------
// The host used to sit in the middle of a refresh: read the counts, build index tensors, launch one gather / scatter
// per side — the GPU idled while it did.  die_ghost_pack and die_ghost_apply take the counts from device memory
// (the plan's totals, the arrived messages' headers); the host reads them once, AFTER everything is enqueued.
struct GhostIO {
    int n_arr_arrays;
    char* arr[GH_ARR_MAX];
    int esz[GH_ARR_MAX];
    int nd;
    int64_t cap[GH_MAXD], hdr_off[GH_MAXD], rec_off[GH_MAXD];
    char* buf;                       // message buffer (send: written, receive: read)
    const int32_t* lists[GH_MAXD];   // pack: index list per side
    const int64_t* totals;           // nd + 2 device words from die_ghost_plan: per side, holes, owned
    const int32_t* holes;            // apply
    const uint16_t* mask;            // apply: the plan's membership words (bit GH_MAXD = hole)
    int64_t n;                       // apply: local agents before the refresh
    int64_t capacity;                // apply: entries every per-agent array holds; arrivals beyond it are dropped (the
                                     // host sees n_new > capacity in the summary and raises)
    int64_t* n_new;                  // apply: device word, local agents afterwards
};

__global__ __launch_bounds__(DIE_BLOCK) void k_ghost_pack(GhostIO a) {
    const int k = blockIdx.y;
    const int64_t total = a.totals[k];
    const int64_t count = total < a.cap[k] ? total : a.cap[k];
    if (blockIdx.x == 0 && threadIdx.x == 0) *(int64_t*)(a.buf + a.hdr_off[k]) = total;     // the receiver checks total <= cap
    int32_t* rec = (int32_t*)(a.buf + a.rec_off[k]);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += stride) {
        const int64_t s = a.lists[k][j];
        for (int f = 0; f < a.n_arr_arrays; ++f) {
            int32_t* r = rec + (int64_t)f * a.cap[k] + j;
            *r = a.esz[f] == 4 ? ((const int32_t*)a.arr[f])[s] : (int32_t)((const uint8_t*)a.arr[f])[s];
        }
    }
}

__device__ __forceinline__ int64_t ghost_arrived(const GhostIO& a, int k) {
    const int64_t c = *(const int64_t*)(a.buf + a.hdr_off[k]);
    return c < 0 ? 0 : (c < a.cap[k] ? c : a.cap[k]);
}

// arrival j (messages in side order) goes into hole j, or behind the old end once the holes are used up
__global__ __launch_bounds__(DIE_BLOCK) void k_ghost_arrivals(GhostIO a) {
    const int k = blockIdx.y;
    int64_t prefix = 0;
    for (int q = 0; q < k; ++q) prefix += ghost_arrived(a, q);
    const int64_t count = ghost_arrived(a, k), H = a.totals[a.nd];
    const int32_t* rec = (const int32_t*)(a.buf + a.rec_off[k]);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const int64_t j = prefix + i;
        const int64_t dst = j < H ? (int64_t)a.holes[j] : a.n + (j - H);
        if (dst >= a.capacity) continue;          // never write past the arrays: k_ghost_tail reports the unclamped count
        for (int f = 0; f < a.n_arr_arrays; ++f) {
            const int32_t v = rec[(int64_t)f * a.cap[k] + i];
            if (a.esz[f] == 4) ((int32_t*)a.arr[f])[dst] = v; else ((uint8_t*)a.arr[f])[dst] = (uint8_t)v;
        }
    }
}

// fewer arrivals than holes: the array shrinks to n_new = n − (H − arrivals); the kept entries of the cut tail
// [n_new, n) move, in order, into the remaining holes below n_new (there are exactly as many).  One workgroup:
// the net change of a refresh is a small fraction of the band.
__global__ __launch_bounds__(1024) void k_ghost_tail(GhostIO a) {
    __shared__ int s[1024];
    __shared__ long long s_base;
    int64_t n_arr = 0;
    for (int q = 0; q < a.nd; ++q) n_arr += ghost_arrived(a, q);
    const int64_t H = a.totals[a.nd];
    const int64_t n_new = n_arr >= H ? a.n + (n_arr - H) : a.n - (H - n_arr);
    if (threadIdx.x == 0) {
        // everything the host wants to know, in one place: [n_new, holes, owned, sent per side…, arrived per side…]
        a.n_new[0] = n_new; a.n_new[1] = H; a.n_new[2] = a.totals[a.nd + 1];
        for (int q = 0; q < a.nd; ++q) { a.n_new[3 + q] = a.totals[q]; a.n_new[3 + a.nd + q] = *(const int64_t*)(a.buf + a.hdr_off[q]); }
        s_base = 0;
    }
    __syncthreads();
    if (n_arr >= H) return;
    for (int64_t p0 = n_new; p0 < a.n; p0 += 1024) {
        const int64_t p = p0 + threadIdx.x;
        const int kept = (p < a.n && !((a.mask[p] >> GH_MAXD) & 1u)) ? 1 : 0;
        s[threadIdx.x] = kept;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int t = (int)threadIdx.x >= o ? s[threadIdx.x - o] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (kept) {
            const int64_t r = (int64_t)s_base + s[threadIdx.x] - 1;
            const int64_t dst = a.holes[n_arr + r];
            for (int f = 0; f < a.n_arr_arrays; ++f) {
                if (a.esz[f] == 4) ((int32_t*)a.arr[f])[dst] = ((const int32_t*)a.arr[f])[p];
                else ((uint8_t*)a.arr[f])[dst] = ((const uint8_t*)a.arr[f])[p];
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_base += s[1023];
        __syncthreads();
    }
}

static int fill_io(GhostIO& k, void* const* arrays, const int32_t* esz, int32_t F, int32_t nd, const int64_t* caps,
                   const int64_t* hdr_off, const int64_t* rec_off, const char* who) {
    DIE_REQUIRE(arrays && esz && F >= 1 && F <= GH_ARR_MAX, "%s: 1..%d arrays", who, GH_ARR_MAX);
    DIE_REQUIRE(nd >= 0 && nd <= GH_MAXD && (nd == 0 || (caps && hdr_off && rec_off)), "%s: 0..%d sides", who, GH_MAXD);
    k.n_arr_arrays = F; k.nd = nd;
    for (int i = 0; i < GH_ARR_MAX; ++i) {
        k.arr[i] = i < F ? (char*)arrays[i] : nullptr; k.esz[i] = i < F ? esz[i] : 0;
        DIE_REQUIRE(i >= F || (arrays[i] && (esz[i] == 4 || esz[i] == 1)), "%s: array %d must be 4- or 1-byte", who, i);
    }
    for (int i = 0; i < GH_MAXD; ++i) {
        k.cap[i] = i < nd ? caps[i] : 0; k.hdr_off[i] = i < nd ? hdr_off[i] : 0; k.rec_off[i] = i < nd ? rec_off[i] : 0;
        k.lists[i] = nullptr;
        DIE_REQUIRE(i >= nd || (caps[i] > 0 && hdr_off[i] % 8 == 0 && rec_off[i] % 4 == 0), "%s: bad message layout %d", who, i);
    }
    return DIE_OK;
}

extern "C" int die_ghost_pack(void* const* arrays, const int32_t* elem_bytes, int32_t n_arrays, int32_t n_dirs,
                              int32_t* const* lists, const int64_t* totals, const int64_t* caps, const int64_t* hdr_off,
                              const int64_t* rec_off, void* send_buf, void* stream) {
    if (n_dirs == 0) return DIE_OK;
    GhostIO k;
    int rc = fill_io(k, arrays, elem_bytes, n_arrays, n_dirs, caps, hdr_off, rec_off, "die_ghost_pack");
    if (rc != DIE_OK) return rc;
    DIE_REQUIRE(lists && totals && send_buf, "die_ghost_pack: null argument");
    int64_t maxcap = 1;
    for (int i = 0; i < n_dirs; ++i) { k.lists[i] = lists[i]; DIE_REQUIRE(lists[i], "die_ghost_pack: null list %d", i); if (caps[i] > maxcap) maxcap = caps[i]; }
    k.totals = totals; k.buf = (char*)send_buf; k.holes = nullptr; k.mask = nullptr; k.n = 0; k.capacity = 0; k.n_new = nullptr;
    int64_t g = (maxcap + DIE_BLOCK - 1) / DIE_BLOCK;
    dim3 grid((unsigned)(g < 256 ? g : 256), (unsigned)n_dirs);
    k_ghost_pack<<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>(k);
    DIE_CHECK_LAUNCH("die_ghost_pack");
    return DIE_OK;
}

extern "C" int die_ghost_apply(void* const* arrays, const int32_t* elem_bytes, int32_t n_arrays, int32_t n_dirs,
                               const int64_t* totals, const int64_t* caps, const int64_t* hdr_off, const int64_t* rec_off,
                               const void* recv_buf, const int32_t* holes, const void* plan_ws, int64_t n_local,
                               int64_t capacity, int64_t* n_new_out, void* stream) {
    GhostIO k;
    int rc = fill_io(k, arrays, elem_bytes, n_arrays, n_dirs, caps, hdr_off, rec_off, "die_ghost_apply");
    if (rc != DIE_OK) return rc;
    DIE_REQUIRE(totals && holes && plan_ws && n_new_out && (n_dirs == 0 || recv_buf) && n_local >= 0, "die_ghost_apply: null argument");
    DIE_REQUIRE(capacity >= n_local, "die_ghost_apply: capacity %lld < %lld local agents", (long long)capacity, (long long)n_local);
    k.totals = totals; k.buf = (char*)recv_buf; k.holes = holes; k.mask = (const uint16_t*)plan_ws; k.n = n_local; k.capacity = capacity;
    k.n_new = n_new_out;
    hipStream_t s = (hipStream_t)stream;
    if (n_dirs > 0) {
        int64_t maxcap = 1;
        for (int i = 0; i < n_dirs; ++i) if (caps[i] > maxcap) maxcap = caps[i];
        int64_t g = (maxcap + DIE_BLOCK - 1) / DIE_BLOCK;
        dim3 grid((unsigned)(g < 256 ? g : 256), (unsigned)n_dirs);
        k_ghost_arrivals<<<grid, DIE_BLOCK, 0, s>>>(k);
    }
    k_ghost_tail<<<1, 1024, 0, s>>>(k);
    DIE_CHECK_LAUNCH("die_ghost_apply");
    return DIE_OK;
}
