// Tile-binned step (gfx950): Agent.forward + Env.step of core/env.py:101-131 / core/agent/gradient.py:96-124 with the
// agents held in EXACT tile order, so that "last writer wins" (core/env.py:211) is resolved in LDS and no claim plane
// exists in HBM.  No reference counterpart for the data structure; the arithmetic per agent and per cell is the same
// code as the classic path (die_forward.h, the field sweep of die_env.hip) and gives the same bits.
//
// Why: in the classic step the 8-byte claim plane is 46 % of the HBM traffic (2.5 M scattered 64-bit atomics fetch and
// write back nearly every line of a 134 MB plane, the sweep reads it again with its halo rows) and the agent kernel is
// bound by the number of L1 misses in flight, not by bytes.  Binning removes both:
//
//   layout    the world is cut into tiles of 2^XS × 2^YS cells.  A layout is a permutation of the agent arrays made of one
//             segment per tile, [off[t], off[t] + n[t]): first the s[t] agents that stand on tile t ("stayers"), then the
//             n[t] − s[t] agents that stood on it one step ago and have walked onto a neighbouring tile ("leavers").  The
//             agents standing on t NOW are its stayers plus those leavers of its 8 neighbours whose cell lies in t.
//   K1        k_pic_forward_move, one workgroup per tile over exactly that set: the chem tile ± the probe reach and the food
//             tile are copied into LDS with 16-byte loads (no gather ever leaves the CU; the loads are issued before
//             anything else, they depend on the tile index only), then forward (4 chem taps, food under the agent), move,
//             feeding of the agent, reward partial; the agent is written to the OTHER layout as a stayer (front of the
//             segment) or a leaver (back) — positions from two LDS counters, one arrival counted per leaver in
//             inc[destination].  Because an agent moves less than a tile per step, a segment of size |set| always fits:
//             no capacity, no overflow.  The action is stored only if the caller passed arrays for it (ACT); a
//             PhysarumAgent's can be re-derived afterwards (die_pic_action_physarum).
//   K2        k_pic_resolve, one workgroup per tile over the same kind of set in the new layout: 64-bit LDS atomicMax of
//             (slot + 1) << 32 | deposit bits per cell, then the tile of the deposit plane is written with coalesced
//             16-byte stores: the winner's deposit, or DIE_DEP_EMPTY, and the food of the occupied cells is reduced
//             (food −= rate·food, core/env.py:222-228; PIC_K2_FEED).  An extra workgroup turns (s, inc) into the segment
//             sizes and offsets of the next step (exclusive scan over the tiles).
//   sweep     k_diffuse_rows<T, R, 2, true> (die_env.hip): deposit + gaussian + decay, reading 4 bytes per cell of deposit
//             plane instead of 8 bytes of claims (and no food at all when K2 has done the feeding); an extra workgroup
//             reduces the reward partials.
//
// Per step the agent arrays are streamed once in and once out by K1 and (x, y, slot, deposit) once by K2; the planes are
// read through the caches by K1 and streamed by the sweep.  The 'agents' channel (claim plane) is not maintained on this
// path: die_agents_mark_owner materialises it when somebody asks (DeviceMedium.occupied / owner_slots / render).
#include "die_forward.h"
#include <stdlib.h>

// (PIC_K2_FEED, PIC_K1_BLOCK, PIC_K2_BLOCK, PIC_STAGE_FOOD, PIC_K1_MINW: knobs of the A/B builds of scratch/build_variant.sh;
// the values below are the measured best: LABBOOK.md, rounds 2–5)
#ifndef PIC_K2_FEED
#define PIC_K2_FEED 1
#endif
// A/B knobs (scratch/build_variant.sh).  PIC_NT: which streams of the two step kernels are non-temporal accesses (`nt`:
// the line is first in line for eviction from L2) — one bit per stream, see the pic_ld / pic_st call sites:
//   0 agent kernel: the six agent streams it reads      1 … the agent_food / heading streams it writes
//   2 … x, y, slot, deposit, rim lists it writes         3 field kernel: agent streams and rim lists it reads
//   4 field kernel: food tile read   5 agent kernel: food tile read   6 field kernel: chem store   7 agent kernel: chem window
//   8 field kernel: chem window      9 field kernel: food store
// PIC_PRIO = s_setprio around the phases that issue the long loads.
#ifndef PIC_NT
#define PIC_NT 0      // (3 buys the field kernel 2.5 µs at 4096² fp32 and costs the agent kernel 3–5 % at 8192² and with fp16 planes: LABBOOK.md, round 3)
#endif
typedef uint32_t pic_u4v __attribute__((ext_vector_type(4)));
typedef uint32_t pic_u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ constexpr bool pic_nt(int bit) { return bit >= 0 && ((PIC_NT >> bit) & 1); }
template <int BIT, class V> __device__ __forceinline__ V pic_ld(const V* a) {
    if constexpr (pic_nt(BIT)) return __builtin_nontemporal_load(a); else return *a;
}
template <int BIT, class V> __device__ __forceinline__ void pic_st(V* a, V v) {
    if constexpr (pic_nt(BIT)) __builtin_nontemporal_store(v, a); else *a = v;
}
template <int BIT> __device__ __forceinline__ uint4 pic_ld4(const void* a) {
    if constexpr (pic_nt(BIT)) { const pic_u4v t = __builtin_nontemporal_load((const pic_u4v*)a); return make_uint4(t.x, t.y, t.z, t.w); }
    else return *(const uint4*)a;
}
template <int BIT> __device__ __forceinline__ void pic_st4(void* a, uint4 v) {
    if constexpr (pic_nt(BIT)) { pic_u4v t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; __builtin_nontemporal_store(t, (pic_u4v*)a); }
    else *(uint4*)a = v;
}
template <int BIT> __device__ __forceinline__ uint2 pic_ld2(const void* a) {
    if constexpr (pic_nt(BIT)) { const pic_u2v t = __builtin_nontemporal_load((const pic_u2v*)a); return make_uint2(t.x, t.y); }
    else return *(const uint2*)a;
}
template <int BIT> __device__ __forceinline__ void pic_st2(void* a, uint2 v) {
    if constexpr (pic_nt(BIT)) { pic_u2v t; t.x = v.x; t.y = v.y; __builtin_nontemporal_store(t, (pic_u2v*)a); }
    else *(uint2*)a = v;
}
// st_sel: the store's cache policy chosen at run time (wave-uniform flag): non-temporal when the step's state does not fit the
// 256 MiB Infinity Cache anyway (KbArgs.nt_out)
template <typename T> struct Vec4;
template <> struct Vec4<float> {
    template <int BIT = -1> static __device__ __forceinline__ void ld(const float* p, float v[4]) {
        const uint4 t = pic_ld4<BIT>(p);
        v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
    }
    template <int BIT = -1> static __device__ __forceinline__ void st(float* p, const float v[4]) {
        pic_st4<BIT>(p, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
    }
    template <int BIT = -1> static __device__ __forceinline__ void st_sel(bool nt, float* p, const float v[4]) {
        if (pic_nt(BIT) || !nt) { st<BIT>(p, v); return; }
        pic_u4v t; t.x = __float_as_uint(v[0]); t.y = __float_as_uint(v[1]); t.z = __float_as_uint(v[2]); t.w = __float_as_uint(v[3]);
        __builtin_nontemporal_store(t, (pic_u4v*)p);
    }
};
template <> struct Vec4<__half> {
    template <int BIT = -1> static __device__ __forceinline__ void ld(const __half* p, float v[4]) {
        const uint2 t = pic_ld2<BIT>(p);
        const __half2 a = *(const __half2*)&t.x, b = *(const __half2*)&t.y;
        v[0] = __low2float(a); v[1] = __high2float(a); v[2] = __low2float(b); v[3] = __high2float(b);
    }
    template <int BIT = -1> static __device__ __forceinline__ void st(__half* p, const float v[4]) {
        const __half2 a = __halves2half2(die_f2h(v[0]), die_f2h(v[1])), b = __halves2half2(die_f2h(v[2]), die_f2h(v[3]));
        uint2 t; t.x = *(const uint32_t*)&a; t.y = *(const uint32_t*)&b;
        pic_st2<BIT>(p, t);
    }
    template <int BIT = -1> static __device__ __forceinline__ void st_sel(bool nt, __half* p, const float v[4]) {
        if (pic_nt(BIT) || !nt) { st<BIT>(p, v); return; }
        const __half2 a = __halves2half2(die_f2h(v[0]), die_f2h(v[1])), b = __halves2half2(die_f2h(v[2]), die_f2h(v[3]));
        pic_u2v t; t.x = *(const uint32_t*)&a; t.y = *(const uint32_t*)&b;
        __builtin_nontemporal_store(t, (pic_u2v*)p);
    }
};

// PIC_PRIO_K1 / PIC_PRIO_KB: s_setprio levels, one hex digit per point of the kernel (agent kernel: start, agent streams
// issued, chunk loop, after the chunk loop; field kernel: start, window loads issued, x pass, unused)
// PIC_KARG: the agent kernel reads its array pointers from the kernel-argument segment where it uses them (scalar loads)
// instead of carrying all 26 of them — most of them spilled to vector lanes — through its chunk loop
#ifndef PIC_KARG
#define PIC_KARG 1
#endif
#ifndef PIC_PRIO_K1
#define PIC_PRIO_K1 0x3003      // (the waves of a starting workgroup issue their loads ahead of the resident workgroups' chunk loops: 83.5 → 81.4 µs;
#endif                          //  raising the chunk loop, or the waves that take a second chunk: worse — LABBOOK.md, rounds 3 and 6)
#ifndef PIC_PRIO_KB
#define PIC_PRIO_KB 0x0000
#endif
#define PIC_SETPRIO(word, i) do { if ((i) == 0 ? (((word) >> 12) & 3) != 0 : ((((word) >> (12 - 4 * (i))) & 3) != (((word) >> (16 - 4 * (i))) & 3))) \
                                      __builtin_amdgcn_s_setprio(((word) >> (12 - 4 * (i))) & 3); } while (0)
#ifndef PIC_R6
#define PIC_R6 1                // A/B: 0 = round 5's prologue and epilogue of the agent kernel (divisions and gridDim in the prologue, priority raised behind it, the tile's last words from wave 0 alone)
#endif
#ifndef PIC_K1_BLOCK
#define PIC_K1_BLOCK 512       // ≈ 614 agents stand on a 64×64 tile at ratio 0.15: one or two trips of the loop
#endif
#ifndef PIC_K2_BLOCK
#define PIC_K2_BLOCK 512
#endif
#define PIC_MAX_MARGIN 24      // probe reach (cells) up to which K1 stages the chem tile in LDS
#ifndef PIC_STAGE_FOOD
#define PIC_STAGE_FOOD 1
#endif

struct PicLayout {
    uint32_t *x, *y;
    float* agent_food;
    uint32_t* slot;
    uint32_t *hhi, *hlo;            // heading, float64 halves
    uint32_t *off, *n, *s, *inc;    // per tile
};

struct PicArgs {
    die_geo g;
    int ntx, nty, xs, ys;           // tiles per axis, log2 of the tile shape
    // divisions the host has done for the kernels' first instructions (a workgroup's PROLOGUE — kernel entry to its first loads — ran
    // 0.86 µs on a loaded CU, a twelfth of its life, half of it four runtime integer divisions: profiles/r06_cu_timeline_4096.txt):
    uint32_t xcd_wb_mul;            // ceil(2^32 / wb), wb = tiles per row / 8 (pic_xcd_tile); 0: divide
    int rp_c, rp_f;                 // rows per staging pass of the chem / food block = workgroup size / 16-byte vectors per staged row
    int margin;                     // K1 stages chem of the tile ± margin cells (a multiple of the 16-byte vector width)
    int fm_r, fm_c;                 // … and food of the tile ± fm_r rows / fm_c columns (periodic): the cells an agent of the tile can walk onto
    uint32_t mg_c, mg_f;            // ceil(2^20 / 16-byte vectors per staged row) of the chem / food block (PicStageRows)
    PicLayout in, out;
    float* dep;                     // N: deposit of every agent, `out` order (K1 → K2)
    float *adx, *ady, *adep;        // the action handed back to the caller, `in` order
    const void* food;
    float rate_feed, w_dep, w_dist;
    int boundary, cost;
    long long* part_gain;           // one fixed-point partial per tile
    uint32_t* error;                // device word, sticky: bit 0 segment bookkeeping broken, bit 1 an agent jumped further than a
                                    // tile, bit 2 a rim record left the 3×3 neighbourhood
    // two-launch form (die_pic.rim != NULL): per tile the list of its agents that matter to another tile's field kernel —
    // those that walked off the tile, and those within R cells of one of the borders of the tile they stand on.  Entry i of
    // tile t: a code byte at rim_code[t·rim_cap + i] = ((ddx + 1)·3 + ddy + 1)·9 + ex·3 + ey (where the agent's tile lies from
    // t; which of ITS borders are near: 0 low, 1 none, 2 high) and a 16-byte record (x, y, slot, deposit bits) at
    // rim[t·rim_cap + i] — so the reader fetches 2 KB of codes for its 9 lists and ONE 16-byte line per agent that matters
    // (four 4-byte gathers from the segment arrays per such agent were a quarter of the field kernel's memory requests)
    uint4* rim;
    uint8_t* rim_code;
    uint32_t* rim_cnt;              // [tile]: entries the tile had (may exceed rim_cap: the reader then scans the segment)
    int rim_cap, rim_r;             // gaussian radius R = width of the rim
    // a launch over a SUBSET of the tiles (die_pic.sub_*: a decomposed rank steps the tiles that need nothing from its neighbours
    // while the ghost refresh's messages are in flight, the others afterwards): 0 all tiles; 1 the rectangle only (the grid is the
    // rectangle); 2 all but the rectangle (full grid, the rectangle's workgroups return at once)
    int sub_mode, sub_tx0, sub_ty0, sub_ntx, sub_nty;
    int halo_fresh;                 // die_pic.halo_fresh
    // GradientAgent with momentum (inertia ≠ 0: die_pic.prev_grad): _prev_grad in `in` order / where the step leaves it, `out` order
    const float *ipgx, *ipgy;
    float *opgx, *opgy;
    // ORDER TABLE (die_pic.order; NULL: the band mapping of pic_xcd_tile): XCD j = linear workgroup id mod 8 takes the tile
    // order[j·order_len + k], k = id div 8 — a permutation of band j's tiles with the crowded ones first (k_pic_order), so that the tiles
    // that live longest do not make up a launch's tail
    const uint16_t* order;
    int order_len;                  // tiles per band = ntx · (nty / 8)
};
// Workgroups are handed to the 8 XCDs round robin (linear workgroup id modulo 8), and every XCD has its own L2.  With the plain
// (blockIdx.y, blockIdx.x) = (tx, ty) mapping the tiles that share cache lines — the margins of their staged windows: a row of a
// window starts 16–48 bytes before a 256-byte boundary and so touches a 128-byte line of each neighbour along y, the margin rows are
// the neighbours' along x — always sit in DIFFERENT L2s, and every shared line is fetched from memory twice.  PIC_XCD_MAP:
//   2 (default): a band of columns per XCD — XCD j takes the tiles with ty in [j·nty/8, (j + 1)·nty/8), walked row of tiles by row of
//      tiles: neighbours along y meet in one L2 at the same time, neighbours along x nty/8 workgroups later, and at any time the 8 XCDs
//      read the same rows of the planes at different columns (4096² fp32: agent kernel FETCH_SIZE −26 %, 79.8 → 77.6 µs)
//   1: one contiguous eighth of the tiles per XCD, in memory order — fewer fetches too (−21 %) but SLOWER (82.1 vs 77.4 µs): the 8 XCDs
//      then walk addresses that differ by exact multiples of an eighth of every array, i.e. the same memory channels at the same time
//   0: plain
#ifndef PIC_XCD_MAP
#define PIC_XCD_MAP 2
#endif
template <bool REVERSE = false>
__device__ __forceinline__ void pic_xcd_tile(int& tx, int& ty, int ntx, uint32_t row0, uint32_t wb_mul, uint32_t nty_) {      // ntx: rows of tiles; row0: grid rows ahead of the tiles' (the field kernel's extra row); wb_mul: PicArgs.xcd_wb_mul; nty_ = gridDim.x = tiles per row (from the arguments: gridDim is a dependent load through the dispatch packet)
#if PIC_XCD_MAP == 1
    if (((gridDim.x * (uint32_t)ntx) & 7u) == 0) {
        const uint32_t L = (blockIdx.y - row0) * gridDim.x + blockIdx.x, G8 = (gridDim.x * (uint32_t)ntx) >> 3;
        const uint32_t nl = (L & 7u) * G8 + (L >> 3);
        tx = (int)(nl / gridDim.x); ty = (int)(nl - (uint32_t)tx * gridDim.x);
    }
#elif PIC_XCD_MAP == 2
    // (gridDim.x = tiles per row.  Bands of wb = floor(nty / 8) columns; the nty mod 8 columns left over — a decomposed rank's planes:
    // 68 tiles per row — come last, in the plain order)
    const uint32_t nty = PIC_R6 ? nty_ : gridDim.x, wb = nty >> 3, banded = (wb << 3) * (uint32_t)ntx;
    const uint32_t L = (blockIdx.y - row0) * nty + blockIdx.x;
    if (wb == 0) return;
    if (L < banded) {
        const uint32_t k = REVERSE ? wb * (uint32_t)ntx - 1u - (L >> 3) : L >> 3;      // (REVERSE: the band walked from its far end — A/B only)
        tx = (int)(PIC_R6 && wb_mul ? __umulhi(k, wb_mul) : k / wb); ty = (int)((L & 7u) * wb + (k - (uint32_t)tx * wb));
    } else {
        const uint32_t r = L - banded, rem = nty - (wb << 3);
        tx = (int)(r / rem); ty = (int)((wb << 3) + (r - (uint32_t)tx * rem));
    }
#endif
}

// tile of linear workgroup L under an order table; false: no tile (a grid rounded up to whole rows, or — never, unless the table is
// broken — an entry beyond the tiles: such a workgroup returns, the step's bookkeeping word then reports the missing tile)
// Only the last PIC_ORDER_SPAN tiles of a band are ever out of band order (k_pic_order): workgroups ahead of them map by arithmetic
// (`use_table` false: the caller takes pic_xcd_tile), the others read ONE table entry — by a SCALAR load of the 32-bit word that holds
// it: as a 16-bit vector load the lookup put a vector-memory round trip (1–2 µs on a loaded CU) in front of every workgroup's first
// loads, which cost an 8192² / 16384² world 2–3 % where the order buys little.
#define PIC_ORDER_SPAN 512
// (Both kernels follow the table.  In a world without crowds the field kernel is ≈ 2 µs slower as soon as the AGENT kernel follows it —
// whatever its own order: measured with the field kernel in band order, 64.3 against 62.1 µs, and following the table too, 63.4 — while
// the agent kernel gains 1–3 µs; with crowds the field kernel gains 14 µs: 82 → 68 µs at world step 3 000, tiles whose rim lists overflow.)
__device__ __forceinline__ uint32_t pic_sload(const uint32_t* a) {
    uint32_t w;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(a) : "memory");
    return w;
}
__device__ __forceinline__ bool pic_order_tile(const PicArgs& p, uint32_t L, int& tx, int& ty, bool& use_table) {
    const uint32_t k = L >> 3;
    use_table = k + (uint32_t)PIC_ORDER_SPAN >= (uint32_t)p.order_len;
    if (!use_table) return true;
    if (k >= (uint32_t)p.order_len) return false;
    const uint32_t e = (L & 7u) * (uint32_t)p.order_len + k;
    const uint32_t t = (pic_sload((const uint32_t*)p.order + (e >> 1)) >> ((e & 1u) * 16u)) & 0xFFFFu;
    if (t >= (uint32_t)(p.ntx * p.nty)) return false;
    tx = (int)(t / (uint32_t)p.nty); ty = (int)(t - (uint32_t)tx * (uint32_t)p.nty);
    return true;
}

__device__ __forceinline__ bool pic_sub_tile(const PicArgs& p, int& tx, int& ty) {        // false: not this launch's tile
    if (p.sub_mode == 1) { tx += p.sub_tx0; ty += p.sub_ty0; return true; }
    if (p.sub_mode == 2) return !(tx >= p.sub_tx0 && tx < p.sub_tx0 + p.sub_ntx && ty >= p.sub_ty0 && ty < p.sub_ty0 + p.sub_nty);
    return true;
}

#define PA_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// (PA_BARRIER instead of __syncthreads() where only LDS is handed over: __syncthreads() is a workgroup-scope fence too and makes
// every wave wait for its outstanding global stores and atomics — in the agent kernel that drained each tile's stores at every barrier)
// plane row / column of the cell an agent stands on.  TILED: the planes are a tile of a decomposed world (die_medium.gW > 0) —
// every kernel below treats that tile as periodic; what this brings in across its outer edge stays in the outermost cells
// of the halo, which the ghost-agent decomposition discards anyway (die_amd/dist.py)
template <bool TILED> __device__ __forceinline__ int pic_row(const die_geo& g, uint32_t X) {
    const int c = die_cell((int64_t)X, g.gW);
    return TILED ? die_plane_coord(c, g.ox, g.W, g.gW) : c;
}
template <bool TILED> __device__ __forceinline__ int pic_col(const die_geo& g, uint32_t Y) {
    const int c = die_cell((int64_t)Y, g.gH);
    return TILED ? die_plane_coord(c, g.oy, g.H, g.gH) : c;
}
template <bool TILED> __device__ __forceinline__ int pic_tile_of(const PicArgs& p, uint32_t X, uint32_t Y) {
    return (pic_row<TILED>(p.g, X) >> p.xs) * p.nty + (pic_col<TILED>(p.g, Y) >> p.ys);
}

// The 9 index ranges that hold the agents standing on `tile` in layout L: [0] its own stayers, [1..8] the leavers of the
// neighbours (periodic neighbourhood; under the 'limit' boundary nothing crosses the seam and those ranges simply fail
// the tile test).  base[r] = first array index, pre[r] = exclusive prefix of the lengths, pre[9] = total candidates.
// Two halves, so that the (dependent) loads of the per-tile words are in flight while the caller stages its tile.
// (The loaded words are only TOUCHED in pic_ranges_finish: arithmetic on them right behind the loads made the compiler wait
// for them — a full memory round trip under load — before it issued the caller's tile loads at all; found in the ISA in
// round 3: `s_waitcnt vmcnt(0)` between the three per-tile loads and the eight tile loads.)
struct PicMeta { uint32_t o, s, n; };

__device__ __forceinline__ int pic_wrap(int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); }     // v in [−n, 2n)

__device__ __forceinline__ PicMeta pic_meta_load(const PicLayout& L, int tx, int ty, int ntx, int nty) {
    PicMeta mt = {0u, 0u, 0u};
    if (threadIdx.x < 9) {
        const int q = threadIdx.x;            // 0: (0, 0); 1..8: the ring
        const int k = q == 0 ? 4 : (q <= 4 ? q - 1 : q);
        const int dx = k / 3 - 1, dy = k % 3 - 1;
        const int nx = pic_wrap(tx + dx, ntx), ny = pic_wrap(ty + dy, nty);
        const int t = nx * nty + ny;
        mt.o = L.off[t]; mt.s = L.s[t]; mt.n = L.n[t];
    }
    return mt;
}

__device__ __forceinline__ void pic_ranges_finish(const PicMeta& mt, uint32_t* base, uint32_t* pre, bool no_arrivals = false) {
    __shared__ uint32_t s_len[9];
    if (threadIdx.x < 9) {
        const bool own = threadIdx.x == 0;
        base[threadIdx.x] = own ? mt.o : mt.o + mt.s;
        s_len[threadIdx.x] = mt.s > mt.n ? 0u : (own ? mt.s : (no_arrivals ? 0u : mt.n - mt.s));     // (s > n: broken bookkeeping — never loop over garbage)
    }
    PA_BARRIER();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int q = 0; q < 9; ++q) { pre[q] = run; run += s_len[q]; }
        pre[9] = run;
    }
    PA_BARRIER();
}

// copy a rows × (vpr·V)-element block of a plane into LDS with 16-byte accesses.  A thread keeps ONE column vector
// (lane group of 2^cs ≥ vpr lanes per row; surplus lanes copy the last column again) and walks down the rows in steps
// of blockDim >> cs: per vector one add, two clamps, a multiply-add and the address — the first cut, with a division for
// the row and tests around every load, spent a quarter of the kernel's instructions here.  Rows / columns outside the
// world are never read by anybody (probes clamp at the world's edge): their loads are clamped into the plane, no branch.
// WRAP: rows / columns beyond the plane are their periodic images (the food under an agent that has just walked across the
// world's seam; H is a multiple of the vector width, so a column vector never straddles the seam).
template <typename T, int NTB = -1, bool WRAP = false>
struct PicStage {
    const T* plane;
    int gx0, gy0, vpr, rows, W, H, cs;
    __device__ __forceinline__ uint4 load(int row, int gyc) const {
        int gx = gx0 + min(row, rows - 1);
        if (WRAP) { gx += gx < 0 ? W : 0; gx -= gx >= W ? W : 0; }
        else gx = min(max(gx, 0), W - 1);
        return pic_ld4<NTB>(plane + (__mul24(gx, H) + gyc));
    }
    __device__ __forceinline__ int column() const { return min((int)threadIdx.x & ((1 << cs) - 1), vpr - 1); }
    __device__ __forceinline__ int col_cell(int cv) const {
        int gy = gy0 + cv * (16 / (int)sizeof(T));
        if (WRAP) { gy += gy < 0 ? H : 0; gy -= gy >= H ? H : 0; return gy; }
        return min(max(gy, 0), H - 16 / (int)sizeof(T));
    }
    // the first NB rows of every thread: requested at the very top of the kernel (issue), written to LDS once the
    // per-tile words and the agent streams have been requested too (commit) …
    template <int NB> __device__ __forceinline__ void issue(uint4 (&v)[NB]) const {
        const int gyc = col_cell(column()), r0 = (int)threadIdx.x >> cs, RP = (int)blockDim.x >> cs;
#pragma unroll
        for (int q = 0; q < NB; ++q) v[q] = load(r0 + q * RP, gyc);
    }
    template <int NB> __device__ __forceinline__ void commit(T* dst, const uint4 (&v)[NB]) const {
        const int cv = column(), r0 = (int)threadIdx.x >> cs, RP = (int)blockDim.x >> cs;
#pragma unroll
        for (int q = 0; q < NB; ++q) ((uint4*)dst)[__mul24(min(r0 + q * RP, rows - 1), vpr) + cv] = v[q];
        // … and whatever a larger tile / a smaller workgroup leaves over
        const int gyc = col_cell(cv);
        for (int row = r0 + NB * RP; row < rows; row += RP) ((uint4*)dst)[__mul24(row, vpr) + cv] = load(row, gyc);
    }
};

// The same copy with lane groups of EXACTLY `vpr` threads per row (row r0 = thread / vpr by a multiply-shift the host has
// verified, blockDim / vpr rows per pass; the last blockDim % vpr threads idle): the power-of-two groups above leave the lanes
// beyond `vpr` idle — 10 of 32 for the 88-cell chem rows, 14 of 32 for the food rows — so the chem tile ± 12 cells takes 4
// vectors per thread instead of 6 and the food tile with its margin 3.  A thread keeps its column; per vector: one add, the
// row's wrap or clamp, a multiply-add, the address.
template <typename T, int NTB, bool WRAP>
struct PicStageRows {
    const T* plane;
    int gx0, gy0, vpr, rows, W, H;
    uint32_t mg;                    // ceil(2^20 / vpr)
    int rp;                         // rows per pass = blockDim / vpr
    __device__ __forceinline__ uint4 load(int row, int gy) const {
        int gx = gx0 + min(row, rows - 1);
        if (WRAP) { gx += gx < 0 ? W : 0; gx -= gx >= W ? W : 0; }
        else gx = min(max(gx, 0), W - 1);
        return pic_ld4<NTB>(plane + (__mul24(gx, H) + gy));
    }
    __device__ __forceinline__ void where(int& r0, int& cv, int& gy) const {
        r0 = (int)(((uint32_t)threadIdx.x * mg) >> 20);
        cv = (int)threadIdx.x - r0 * vpr;
        gy = gy0 + cv * (16 / (int)sizeof(T));
        if (WRAP) { gy += gy < 0 ? H : 0; gy -= gy >= H ? H : 0; }
        else gy = min(max(gy, 0), H - 16 / (int)sizeof(T));
    }
    // the first NB rows of every thread: requested at the very top of the kernel (issue), written to LDS once the
    // per-tile words and the agent streams have been requested too (commit) …
    template <int NB> __device__ __forceinline__ void issue(uint4 (&v)[NB]) const {
        int r0, cv, gy;
        where(r0, cv, gy);
#pragma unroll
        for (int q = 0; q < NB; ++q) v[q] = load(r0 + q * rp, gy);          // (idle threads: r0 = rp, rows clamp: a harmless copy)
    }
    template <int NB> __device__ __forceinline__ void commit(T* dst, const uint4 (&v)[NB]) const {
        int r0, cv, gy;
        where(r0, cv, gy);
        if (r0 >= rp) return;                                            // the blockDim % vpr threads without a column
        uint4* d = (uint4*)dst + (r0 * vpr + cv);
#pragma unroll
        for (int q = 0; q < NB; ++q) if (r0 + q * rp < rows) d[q * rp * vpr] = v[q];
        // … and whatever a larger tile / a smaller workgroup leaves over
        for (int row = r0 + NB * rp; row < rows; row += rp) ((uint4*)dst)[row * vpr + cv] = load(row, gy);
    }
};

#define PIC_RIM_CAP_MAX 224      // entries of a tile's rim list (k_pic_resolve_diffuse reads the codes of 9 lists with one word per thread)
#ifndef PIC_LIST_CAP
#define PIC_LIST_CAP 512        // arrivals of one tile compacted per round (≈ 80 arrive in the benchmark world)
#endif
// (The first round appends up to one candidate per thread: a list shorter than the workgroup is written past its end.  A
// round-2 experiment build with a shorter list faulted that way — most probably the `n1` run of gpurun_out/sw3_n1.err,
// DESIGN.md §9 — hence the assertion; tests/test_gpu_parity.py::test_tile_binned_step_with_a_crowd_crossing_one_border
// drives the rounds below beyond the first on the shipped kernel.)
static_assert(PIC_LIST_CAP >= PIC_K1_BLOCK, "the first round of candidates is one per thread");

// Diagnostic build only (-DPIC_STAMPS; scratch/pic_stamps.py): s_memtime at the phase boundaries of K1, written by lane 0
// of wave 0 behind the error word (the caller allocates 2 + 40·tiles words: 16 64-bit stamps per tile, [0, 8) the agent
// kernel's, [8, 16) the field kernel's; then 4 more per tile, below).  No stamp executes in the shipped kernel.
#if defined(PIC_STAMPS) && !defined(PIC_STAMPS_AGENTS_ONLY)
#ifdef PIC_STAMPS_RT            // s_memrealtime: ONE 100 MHz clock for the whole GPU (s_memtime's counters are per XCD and not comparable)
#define PIC_CLOCK "s_memrealtime"
#else
#define PIC_CLOCK "s_memtime"
#endif
#define PIC_NOW(t_) asm volatile(PIC_CLOCK " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory")
#define PIC_STAMP(k) do { if (threadIdx.x == 0) { unsigned long long t_; PIC_NOW(t_); \
                          ((unsigned long long*)(p.error + 2))[(size_t)tile * 16 + (k)] = t_; } } while (0)
// round 6: where a workgroup ran and when it ENTERED the kernel — 4 more 64-bit words per tile behind the 16 stamps (the caller
// allocates 2 + 40·tiles words): [0] agent kernel: XCC_ID << 32 | HW_ID, [1] its entry time, [2] / [3] the field kernel's.
// (kernel-entry time is taken before the tile is known: PIC_ENTRY_DECL at the top, PIC_ENTRY_STORE once `tile` exists)
#define PIC_ENTRY_DECL unsigned long long t_entry_ = 0; if (threadIdx.x == 0) PIC_NOW(t_entry_)
#define PIC_ENTRY_STORE(which) do { if (threadIdx.x == 0) { uint32_t hw_, xcc_; \
                                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_)); \
                                unsigned long long* e_ = (unsigned long long*)(p.error + 2) + (size_t)p.ntx * p.nty * 16 + (size_t)tile * 4 + 2 * (which); \
                                e_[0] = ((unsigned long long)xcc_ << 32) | hw_; e_[1] = t_entry_; } } while (0)
#else
#define PIC_STAMP(k) do { } while (0)
#define PIC_ENTRY_DECL do { } while (0)
#define PIC_ENTRY_STORE(which) do { } while (0)
#endif

#ifndef PIC_K1_MINW
#define PIC_K1_MINW 6
#endif
#ifndef PIC_STATIC_CHUNKS
#define PIC_STATIC_CHUNKS 0     // 1: wave w takes chunks w, w + waves, … instead of drawing them from an LDS counter
#endif
// element `idx` of a 4-byte-per-agent array: a 32-bit byte offset on the array's (scalar) base — the host refuses worlds of
// 2^30 agents —, so every stream of an agent shares ONE offset register instead of a 64-bit address of its own
#ifndef PIC_NT_OUT
#define PIC_NT_OUT (-1)         // field kernel's plane stores non-temporal: −1 by the state's size (die_pic_forward_env_step), 0 never, 1 always
#endif
#ifndef PIC_K1_NT_LD
#define PIC_K1_NT_LD 0          // agent kernel: its six agent streams (own segment: read once) loaded non-temporally (A/B)
#endif
#define PIC_LDN(base, type, idx) (PIC_K1_NT_LD ? __builtin_nontemporal_load(&PIC_AT(base, type, idx)) : PIC_AT(base, type, idx))
#ifndef PIC_FOOD_COLS
#define PIC_FOOD_COLS 0         // 1: the agent kernel's food block has a margin of columns too (the round-4 first cut: every new cell from LDS)
#endif
#ifndef PIC_TB
#define PIC_TB true             // the agent kernel reads a PhysarumAgent's random turn bit from the step's table (false: one Philox block per agent whose turn is random)
#endif
#define PIC_AT(base, type, idx) (*(type*)((char*)(base) + (size_t)(uint32_t)((uint32_t)(idx) << 2)))
// (Two persistent forms of this kernel — a tile queue with the next tile's agents prefetched, and one 16-wave workgroup per CU with
// LDS-DMA loader waves — were built, bit-equal, and measured slower in round 4 (96 / 101 µs against 75–81): LABBOOK.md,
// scratch/refuted_r04/.)
template <typename T, int KIND, bool STAGE, bool ACT, bool RIM, bool TILED, bool MOM = false>
// (waves per SIMD the compiler must leave room for: with fp16 planes the staged windows are 25 KB per workgroup, FOUR workgroups fit a
// CU's LDS and the registers have to fit 8 waves per SIMD too — 78 scalar registers + 29 spilled instead of 106 + 17; with fp32
// planes, 49 KB, only three fit whatever the registers)
#ifndef PIC_K1_MINW_F16
#define PIC_K1_MINW_F16 8
#endif
__global__ __launch_bounds__(PIC_K1_BLOCK, (sizeof(T) == 2 ? PIC_K1_MINW_F16 : PIC_K1_MINW)) void k_pic_forward_move(FwdArgs f, PicArgs p) {
    PIC_ENTRY_DECL;
    // (a starting workgroup's FIRST instructions already run at raised priority: its ≈ 300 instructions of prologue otherwise queue
    // behind the resident workgroups' chunk loops — round 6, profiles/r06_cu_timeline_4096.txt)
    if (PIC_R6 && ((PIC_PRIO_K1 >> 12) & 3) != 0) __builtin_amdgcn_s_setprio((PIC_PRIO_K1 >> 12) & 3);
    // (Requesting every argument the prologue needs with the FIRST scalar loads changed nothing, 75.1 against 74.6–75.3 µs: the compiler
    // re-loads them where it uses them.  LABBOOK.md, round 6.)
    // what die_pic_forward_env_step has checked on the host, spelled out for the compiler: the momentum / noise / graph
    // replay paths of the shared forward code and the scalar registers that feed them drop out of this kernel (it was
    // spilling scalar registers to vector lanes: ≈ 290 of its 1 900 vector instructions were v_readlane / v_writelane)
    // MOM (GradientAgent with inertia / noise, gradient.py:82-91): the momentum path stays; _prev_grad is read from the `in`
    // arrays at the agent's index and written to the `out` arrays where the agent goes
    f.pgx = MOM ? (float*)p.ipgx : nullptr; f.pgy = MOM ? (float*)p.ipgy : nullptr; f.step_base = nullptr; f.mask = nullptr;
    if (!MOM) { f.inertia = 0.f; f.noise_scale = 0.f; }
    f.normalized = 1;
    f.g = p.g;                            // one copy of the geometry
#if PIC_KARG
    // The array pointers are needed at a few places each — the streams once per chunk, the epilogue's once per tile.  Kept in
    // scalar registers from the kernel's entry they (106 registers + 70 spilled to vector lanes, every use a v_readlane and a
    // hazard nop: vector-issue slots of a kernel that is half vector issue) cost more than re-reading them from the
    // kernel-argument segment right where they are used: volatile, so that the compiler neither hoists nor keeps them.
    // The two by-value arguments lie in the segment one after the other, each 8-byte aligned.
    struct KArgs { FwdArgs f; PicArgs p; };
    static_assert(alignof(FwdArgs) == 8 && alignof(PicArgs) == 8 && sizeof(FwdArgs) % 8 == 0, "kernel-argument layout of k_pic_forward_move");
    const volatile KArgs __attribute__((address_space(4)))* ka = (const volatile KArgs __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
#define PIC_KP(field, type) ((type)ka->p.field)
#else
#define PIC_KP(field, type) (p.field)
#endif
    extern __shared__ __align__(16) unsigned char pic_smem[];     // STAGE: chem of the tile ± margin, then food of the tile ± its margin
    __shared__ uint32_t s_base[9], s_pre[10];                     // ranges of the current tile
    __shared__ unsigned long long s_cnt;                           // stayers | leavers << 21 | rim entries << 42: one LDS atomic per wave and chunk
    __shared__ uint32_t s_next, s_nlist;
    __shared__ uint32_t s_inc[9];                                  // arrivals this tile sends to each neighbour: (ddx + 1)·3 + ddy + 1
    __shared__ __align__(4) uint8_t s_rimc[RIM ? PIC_RIM_CAP_MAX : 4];   // RIM: the codes of this tile's list (flushed as words)
    __shared__ uint32_t s_list[PIC_LIST_CAP];
    __shared__ long long s_gain[PIC_K1_BLOCK / DIE_WAVE];
    __shared__ uint32_t s_alv[TILED ? PIC_K1_BLOCK / DIE_WAVE : 1];   // TILED: agents this rank accounts for (die_medium.own_*)
    const int NT = p.ntx * p.nty;
    const int TX = 1 << p.xs, TY = 1 << p.ys;
    const int lane = threadIdx.x & (DIE_WAVE - 1), wave = threadIdx.x / DIE_WAVE, nwaves = blockDim.x / DIE_WAVE;
    const unsigned long long below = (1ull << lane) - 1ull;
    const T* food = (const T*)p.food;
    constexpr int SV = 16 / (int)sizeof(T);
    const int P = p.margin, pitch = TY + 2 * P, rows = TX + 2 * P;
    const int FR = p.fm_r, FC = p.fm_c, fpitch = TY + 2 * FC, frows = TX + 2 * FR;
    // the tile of this workgroup
    int tx = (int)blockIdx.y, ty = (int)blockIdx.x;
    bool by_table = false;
    if (p.order && !pic_order_tile(p, blockIdx.y * (uint32_t)p.nty + blockIdx.x, tx, ty, by_table)) return;
    if (!by_table && p.sub_mode == 0) pic_xcd_tile(tx, ty, p.ntx, 0, p.xcd_wb_mul, (uint32_t)p.nty);
    if (!pic_sub_tile(p, tx, ty)) return;
    const int tile = tx * p.nty + ty;
    (void)NT;
    // the per-tile counters
    if (threadIdx.x == 0) { s_cnt = 0ull; s_next = (uint32_t)(nwaves * DIE_WAVE); s_nlist = 0; }
    if (threadIdx.x < 9) s_inc[threadIdx.x] = 0;
    // what the first pass needs from memory besides the windows: this thread's first candidate arrival and the six streams of
    // this wave's first chunk of stayers
    uint32_t cj = 0, cX = 0, cY = 0, pX = 0, pY = 0, pS = 0, pHh = 0, pHl = 0;
    float pA = 0.f;
    bool chas = false;
    auto prefetch_agents = [&]() {
        const uint32_t own = s_pre[1], ncand = s_pre[9] - own, base0 = s_base[0];
        // this thread's first candidate arrival and the agent streams of this wave's first chunk of stayers
        cj = 0; cX = 0; cY = 0;
        chas = threadIdx.x < ncand;
        if (chas) {
            const uint32_t idx = own + threadIdx.x;
            int r = 1;
            while (idx >= s_pre[r + 1]) ++r;
            cj = s_base[r] + (idx - s_pre[r]);
            cX = PIC_AT(p.in.x, const uint32_t, cj);
            cY = PIC_AT(p.in.y, const uint32_t, cj);
        }
        const uint32_t pidx = (uint32_t)(wave * DIE_WAVE + lane);
        pX = 0; pY = 0; pS = 0; pHh = 0; pHl = 0; pA = 0.f;
        if (pidx < own) {
            const uint32_t j = base0 + pidx;
            pX = PIC_LDN(p.in.x, const uint32_t, j); pY = PIC_LDN(p.in.y, const uint32_t, j); pS = PIC_LDN(p.in.slot, const uint32_t, j);
            pHh = PIC_LDN(p.in.hhi, const uint32_t, j); pHl = PIC_LDN(p.in.hlo, const uint32_t, j);
            pA = PIC_LDN(p.in.agent_food, const float, j);
        }
    };
    const int x0 = tx << p.xs, y0 = ty << p.ys;
    PIC_ENTRY_STORE(0);
    PIC_STAMP(0);
    PIC_SETPRIO(PIC_PRIO_K1, 0);
    // 1st round trip: the per-tile words (small arrays, L2-resident).  Requested FIRST: vector loads return in order, so a
    // word requested behind the tile loads would only arrive after all of them (stamps: 6 600 cycles for this phase).
    const PicMeta mt = pic_meta_load(p.in, tx, ty, p.ntx, p.nty);
    // the tiles to stage depend on nothing but the tile index: their loads go out next and overlap both round trips
    const PicStageRows<T, 7, false> st_c = {(const T*)f.chem, x0 - P, y0 - P, pitch / SV, rows, p.g.W, p.g.H, p.mg_c, PIC_R6 ? p.rp_c : (int)blockDim.x / (pitch / SV)};
    const PicStageRows<T, 5, true> st_f = {food, x0 - FR, y0 - FC, fpitch / SV, frows, p.g.W, p.g.H, p.mg_f, PIC_R6 ? p.rp_f : (int)blockDim.x / (fpitch / SV)};
    uint4 sc[4], sf[3];                   // 64×64 tile, 512 threads: chem ± 12 cells = 88 × 22 vectors, food ± (3, 4) = 70 × 18
    if (STAGE) {
        st_c.issue(sc);
        st_f.issue(sf);
    }
    const uint32_t obase = p.out.off[tile], on = p.out.n[tile];
    // (TILED, the step behind a refresh in place — die_pic_ghost_inplace: a halo tile holds exactly what arrived for it; the neighbours'
    // leavers that stand on it are stale copies of agents that came with the message)
    pic_ranges_finish(mt, s_base, s_pre, TILED && p.halo_fresh && p.g.own_x1 > 0 && (x0 < p.g.own_x0 || x0 >= p.g.own_x1 || y0 < p.g.own_y0 || y0 >= p.g.own_y1));
    prefetch_agents();
    PIC_STAMP(1);
    const uint32_t own = s_pre[1], ncand = s_pre[9] - own, base0 = s_base[0];
    PIC_SETPRIO(PIC_PRIO_K1, 1);
    FwdTileMem<T, TILED> tm;
    tm.g = p.g;
    T* s_food = nullptr;
    if (STAGE) {
        T* s_chem = (T*)pic_smem;
        s_food = s_chem + rows * pitch;
        st_c.commit(s_chem, sc);
        st_f.commit(s_food, sf);
        tm.food = s_food; tm.fx0 = x0 - FR; tm.fy0 = y0 - FC; tm.fpitch = fpitch;
        tm.chem = s_chem; tm.cx0 = x0 - P; tm.cy0 = y0 - P; tm.pitch = pitch;
    }
    PIC_STAMP(2);
    long long gsum = 0;
    uint32_t nowned = 0;
    // Rounds: the tile's own stayers plus (at most PIC_LIST_CAP per round) the neighbours' leavers that landed here,
    // compacted into s_list first so that the heavy part below runs on full waves.  One round unless a crowd arrives.
    for (uint32_t cb = 0; cb == 0 || cb < ncand; cb += PIC_LIST_CAP) {
        const uint32_t cend = min(cb + (uint32_t)PIC_LIST_CAP, ncand);
        for (uint32_t c0 = cb; c0 < cend; c0 += blockDim.x) {      // wave-uniform trip count
            const uint32_t c = c0 + threadIdx.x;
            bool hit = false;
            uint32_t j = 0;
            if (c0 == 0) {                                         // loaded above
                j = cj;
                hit = chas && pic_tile_of<TILED>(p, cX, cY) == tile;
            } else if (c < cend) {
                const uint32_t idx = own + c;
                int r = 1;
                while (idx >= s_pre[r + 1]) ++r;
                j = s_base[r] + (idx - s_pre[r]);
                hit = pic_tile_of<TILED>(p, PIC_AT(p.in.x, const uint32_t, j), PIC_AT(p.in.y, const uint32_t, j)) == tile;
            }
            const unsigned long long m = __ballot(hit);
            uint32_t at = 0;
            if (lane == 0 && m) at = atomicAdd(&s_nlist, (uint32_t)__popcll(m));
            at = __shfl(at, 0, DIE_WAVE);
            if (hit) s_list[at + (uint32_t)__popcll(m & below)] = j;
        }
        PA_BARRIER();                                           // publishes the staged tiles and the list
        PIC_STAMP(3);
        PIC_SETPRIO(PIC_PRIO_K1, 2);
        const uint32_t n_own = cb == 0 ? own : 0u, count = n_own + s_nlist;
        bool first = cb == 0;
        uint32_t cprev = 0;
        for (;;) {                         // a wave's first chunk is fixed (its streams are already here), then it takes
            uint32_t c = 0;                // chunks of 64 items from the counter until none are left
            if (first) {
                c = (uint32_t)(wave * DIE_WAVE);
            } else if (PIC_STATIC_CHUNKS) {
                c = cprev + (uint32_t)(nwaves * DIE_WAVE);
            } else {
                if (lane == 0) c = atomicAdd(&s_next, (uint32_t)DIE_WAVE);
                c = __shfl(c, 0, DIE_WAVE);
            }
            cprev = c;
            if (c >= count) break;
            const uint32_t idx = c + lane;
            const bool act = idx < count;
            bool stay = false;
            uint32_t X = 0, Y = 0, sid = 0, hh = 0, hl = 0;
            float af = 0.f, dep = 0.f;
            [[maybe_unused]] float pux = 0.f, puy = 0.f;
            double hd = 0.0;
            uint32_t code = 0;
            bool listed = false;
            if (act) {
                const uint32_t j = idx < n_own ? base0 + idx : s_list[idx - n_own];
                if (first && idx < n_own) {                      // (this wave's prefetched chunk: idx = wave·64 + lane < own)
                    X = pX; Y = pY; sid = pS; hh = pHh; hl = pHl; af = pA;
                } else {
                    // (the pointers first — their scalar loads go out together —, then all six streams in flight together)
                    const uint32_t *ix_ = PIC_KP(in.x, const uint32_t*), *iy_ = PIC_KP(in.y, const uint32_t*), *is_ = PIC_KP(in.slot, const uint32_t*);
                    const uint32_t *ihh_ = PIC_KP(in.hhi, const uint32_t*), *ihl_ = PIC_KP(in.hlo, const uint32_t*);
                    const float* ia_ = PIC_KP(in.agent_food, const float*);
                    X = PIC_LDN(ix_, const uint32_t, j); Y = PIC_LDN(iy_, const uint32_t, j); sid = PIC_LDN(is_, const uint32_t, j);
                    hh = PIC_LDN(ihh_, const uint32_t, j); hl = PIC_LDN(ihl_, const uint32_t, j);
                    af = PIC_LDN(ia_, const float, j);
                }
                hd = __hiloint2double((int)hh, (int)hl);
                const FwdOut o = STAGE ? die_forward_agent_mem<T, KIND, false, FwdTileMem<T, TILED>, PIC_TB, false>(f, tm, X, Y, hd, sid, (int64_t)j)
                                       : die_forward_agent_mem<T, KIND, false, FwdGlobalMem<T, false>, PIC_TB, false>(f, FwdGlobalMem<T, false>(f), X, Y, hd, sid, (int64_t)j);
                if (MOM) { pux = o.ux; puy = o.uy; }
                if (ACT && p.adx) { PIC_AT(p.adx, float, j) = o.dx; PIC_AT(p.ady, float, j) = o.dy; PIC_AT(p.adep, float, j) = o.dep; }   // ACT = false: the caller passed no action arrays
                // _agent_move (core/env.py:163-172)
                // (a step is shorter than a tile — checked on the host — so the fixed-point increment needs no float64 path)
                if (p.boundary == DIE_BOUNDARY_WRAP) {
                    X += (uint32_t)die_q32_small(o.dx);
                    Y += (uint32_t)die_q32_small(o.dy);
                } else {
                    const int64_t qx = (int64_t)X + die_q32_small(o.dx), qy = (int64_t)Y + die_q32_small(o.dy);
                    X = (uint32_t)(qx < 0 ? 0 : (qx > 0xFFFFFFFFLL ? 0xFFFFFFFFLL : qx));
                    Y = (uint32_t)(qy < 0 ? 0 : (qy > 0xFFFFFFFFLL ? 0xFFFFFFFFLL : qy));
                }
                const int gcx = die_cell_u(X, p.g.gW), gcy = die_cell_u(Y, p.g.gH);                      // world cell …
                const int cx = TILED ? die_plane_coord(gcx, p.g.ox, p.g.W, p.g.gW) : gcx;                // … and where the planes hold it
                const int cy = TILED ? die_plane_coord(gcy, p.g.oy, p.g.H, p.g.gH) : gcy;
                const int ntx_ = cx >> p.xs, nty_ = cy >> p.ys;
                stay = ntx_ == tx && nty_ == ty;
                // _agent_feed for this (alive) agent (core/env.py:220-243): the food under it BEFORE this step's consumption —
                // from the staged food block, whose margin holds every cell an agent of the tile can reach in one step (a flat
                // load from "LDS or global memory" here made every wave drain its outstanding stores before each chunk)
                float fnew;
                if (STAGE) {
                    int rx = cx - x0, ry = cy - y0;                                    // relative to the tile, periodic
                    rx += rx < -FR ? p.g.W : 0; rx -= rx >= TX + FR ? p.g.W : 0;
                    ry += ry < -FC ? p.g.H : 0; ry -= ry >= TY + FC ? p.g.H : 0;
                    rx = min(max(rx + FR, 0), frows - 1);                              // (a longer jump is an error, flagged below: never out of the block)
                    // the block has a margin of ROWS only by default (fm_c = 0): a row of the plane starts on a 256-byte boundary of
                    // the tile, so whole rows cost no partial cache lines, while ± 4 columns made every row touch two more 128-byte
                    // lines (70 rows × 4 lines instead of 64 × 2: + 65 MB of fetches per step at 4096²).  The few agents that leave
                    // the tile's columns (≈ 1 % per step at the benchmark's step length) read their cell from global memory
                    if (ry >= -FC && ry < TY + FC) fnew = die_ld(s_food, (int64_t)(rx * fpitch + ry + FC));
                    else fnew = die_ld(food, (int64_t)cx * p.g.H + cy);
                } else {
                    fnew = die_ld(food, (int64_t)cx * p.g.H + cy);
                }
                const float consumed = p.rate_feed * fnew;
                const float cost = p.cost == DIE_COST_LINEAR ? p.w_dep * fabsf(o.dep) + p.w_dist * die_sqrt1(o.dx * o.dx + o.dy * o.dy) : 0.f;
                const float gained = consumed - cost;
                af += gained;
                // (a ghost is its owner's to count; die_owned on the plane element: a cell beyond the planes maps to an edge element, never owned)
                if (!TILED || p.g.own_x1 == 0 || (cx >= p.g.own_x0 && cx < p.g.own_x1 && cy >= p.g.own_y0 && cy < p.g.own_y1)) { gsum += die_fix(gained); ++nowned; }
                hd = o.heading;
                dep = o.dep;
                int ddx = 0, ddy = 0;
                if (!stay) {
                    ddx = ntx_ - tx; ddy = nty_ - ty;
                    ddx = ddx > 1 ? ddx - p.ntx : (ddx < -1 ? ddx + p.ntx : ddx);
                    ddy = ddy > 1 ? ddy - p.nty : (ddy < -1 ? ddy + p.nty : ddy);
                    if (ddx < -1 || ddx > 1 || ddy < -1 || ddy > 1) { atomicOr(p.error, 2u); ddx = ddy = 0; }
                    else atomicAdd(&s_inc[(ddx + 1) * 3 + ddy + 1], 1u);   // one global atomic per neighbour at the end (2.5 M
                }                                                          // agents: 11 µs of contended global atomics otherwise)
                if (RIM) {
                    // the field kernel of a tile diffuses that tile's cells and so needs the deposits on the R cells around it
                    // too: listed for it are the agents that walked off this tile and those that stand within R cells of a
                    // border of their (new) tile
                    const int lx = cx & (TX - 1), ly = cy & (TY - 1), Rr = p.rim_r;
                    const int ex = lx < Rr ? 0 : (lx >= TX - Rr ? 2 : 1), ey = ly < Rr ? 0 : (ly >= TY - Rr ? 2 : 1);
                    code = (uint32_t)(((ddx + 1) * 3 + ddy + 1) * 9 + ex * 3 + ey);
                    listed = !stay || ex != 1 || ey != 1;
                }
            }
            // positions: stayers fill the segment from the front, leavers from the back, listed agents their list; the three
            // counters ride in one 64-bit word (21 bits each: the host refuses tiles of 2 M agents), so a wave pays ONE LDS
            // atomic per chunk
            const unsigned long long m_stay = __ballot(act && stay), m_leave = __ballot(act && !stay), m_rim = RIM ? __ballot(act && listed) : 0ull;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt, (unsigned long long)__popcll(m_stay) | ((unsigned long long)__popcll(m_leave) << 21) |
                                                    ((unsigned long long)__popcll(m_rim) << 42));
            base = __shfl(base, 0, DIE_WAVE);
            const uint32_t bf = (uint32_t)base & 0x1FFFFFu, bb = (uint32_t)(base >> 21) & 0x1FFFFFu, br = (uint32_t)(base >> 42) & 0x1FFFFFu;
            const uint32_t k = stay ? bf + (uint32_t)__popcll(m_stay & below) : on - 1u - (bb + (uint32_t)__popcll(m_leave & below));
            if (RIM && act && listed) {
                const uint32_t at = br + (uint32_t)__popcll(m_rim & below);
                if (at < (uint32_t)p.rim_cap) {
                    s_rimc[at] = (uint8_t)code;
                    pic_st4<2>(&PIC_KP(rim, uint4*)[(size_t)tile * p.rim_cap + at], make_uint4(X, Y, sid, __float_as_uint(dep)));
                }
            }
            if (act) {
                if (k < on) {
                    const uint32_t q = obase + k;
                    uint32_t *ox_ = PIC_KP(out.x, uint32_t*), *oy_ = PIC_KP(out.y, uint32_t*), *os_ = PIC_KP(out.slot, uint32_t*);
                    uint32_t *ohh_ = PIC_KP(out.hhi, uint32_t*), *ohl_ = PIC_KP(out.hlo, uint32_t*);
                    float *oa_ = PIC_KP(out.agent_food, float*), *od_ = PIC_KP(dep, float*);
#ifndef PIC_K1_NT_ST
#define PIC_K1_NT_ST 0          // bit 0: x, y, slot, deposit (the field kernel reads them next); bit 1: agent_food, heading — non-temporal stores (A/B)
#endif
#define PIC_ST(bit, base, type, idx, val) do { if ((PIC_K1_NT_ST >> (bit)) & 1) __builtin_nontemporal_store((type)(val), &PIC_AT(base, type, idx)); else PIC_AT(base, type, idx) = (val); } while (0)
                    PIC_ST(0, ox_, uint32_t, q, X);                // (x, y, slot, deposit are read again by the field kernel)
                    PIC_ST(0, oy_, uint32_t, q, Y);
                    PIC_ST(1, oa_, float, q, af);
                    PIC_ST(0, os_, uint32_t, q, sid);
                    PIC_ST(1, ohh_, uint32_t, q, (uint32_t)__double2hiint(hd));
                    PIC_ST(1, ohl_, uint32_t, q, (uint32_t)__double2loint(hd));
                    PIC_ST(0, od_, float, q, dep);
                    if (MOM && p.opgx) { PIC_AT(p.opgx, float, q) = pux; PIC_AT(p.opgy, float, q) = puy; }
                } else {
                    atomicOr(PIC_KP(error, uint32_t*), 1u);
                }
            }
            first = false;
        }
        if (cb + PIC_LIST_CAP < ncand) {                           // another round (rare): reset the work counters
            PA_BARRIER();
            if (threadIdx.x == 0) { s_next = 0; s_nlist = 0; }
            PA_BARRIER();
        }
    }
    PIC_STAMP(4);
    PIC_SETPRIO(PIC_PRIO_K1, 3);
    gsum = die_wave_sum(gsum);
    if (lane == 0) s_gain[threadIdx.x / DIE_WAVE] = gsum;
    if (TILED) {
        const long long c = die_wave_sum((long long)nowned);
        if (lane == 0) s_alv[threadIdx.x / DIE_WAVE] = (uint32_t)c;
    }
    // The tile's last words — arrival counts, rim codes, reward partial, stayer count — go out from THREE waves side by side, and the
    // pointers they go to are requested ahead of the barrier.  (Round 6: all of it in wave 0, one dependent scalar load after the other,
    // kept a workgroup's slot 0.64 µs beyond its last barrier — profiles/r06_cu_timeline_4096.txt; the slot is not refilled before the
    // last wave has ended.)
    const int w_inc = PIC_R6 && nwaves > 1 ? 1 : 0, w_rim = PIC_R6 && nwaves > 2 ? 2 : 0;
    uint32_t* e_inc = nullptr; uint8_t* e_rimc = nullptr; uint32_t* e_rimn = nullptr; long long* e_gain = nullptr;
    if (wave == w_inc) e_inc = PIC_KP(out.inc, uint32_t*);
    if (RIM && wave == w_rim) { e_rimc = PIC_KP(rim_code, uint8_t*); e_rimn = PIC_KP(rim_cnt, uint32_t*); }
    if (wave == 0) e_gain = PIC_KP(part_gain, long long*);
    PA_BARRIER();
    PIC_STAMP(5);
    if (wave == w_inc && lane < 9 && s_inc[lane]) {
        const int ddx = lane / 3 - 1, ddy = lane % 3 - 1;
        atomicAdd(&e_inc[pic_wrap(tx + ddx, p.ntx) * p.nty + pic_wrap(ty + ddy, p.nty)], s_inc[lane]);
    }
    if (RIM && wave == w_rim) {
        // (a segment too long for the 24-bit positions counts as an overflowing list: the reader scans it)
        const uint32_t nr = on >= (1u << 21) ? (uint32_t)p.rim_cap + 1u : (uint32_t)(s_cnt >> 42) & 0x1FFFFFu;
        for (uint32_t i = (uint32_t)lane; i < (min(nr, (uint32_t)p.rim_cap) + 3u) / 4u; i += DIE_WAVE)
            pic_st<2>(&((uint32_t*)e_rimc)[((size_t)tile * p.rim_cap) / 4 + i], ((const uint32_t*)s_rimc)[i]);
        if (lane == 0) e_rimn[tile] = nr;
    }
    if (wave == 0) {
        long long t = die_wave_sum(lane < nwaves ? s_gain[lane] : 0ll);          // (integers: any order)
        [[maybe_unused]] long long c = 0;
        if (TILED) c = die_wave_sum(lane < nwaves ? (long long)s_alv[lane] : 0ll);
        if (lane == 0) {
            e_gain[tile] = t;
            if (TILED) e_gain[(size_t)p.ntx * p.nty + tile] = c;               // second half of the array: owned agents per tile
            const uint32_t nfront = (uint32_t)s_cnt & 0x1FFFFFu, nback = (uint32_t)(s_cnt >> 21) & 0x1FFFFFu;
            p.out.s[tile] = nfront;
            if (nfront + nback != on || on >= (1u << 21)) atomicOr(p.error, 1u);
        }
    }
    PIC_STAMP(7);
#if defined(PIC_STAMPS) && !defined(PIC_STAMPS_AGENTS_ONLY)
    // (diagnostic: when have wave 0's own stores — the epilogue's, issued a moment ago — been acknowledged?)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PIC_STAMP(6);
#endif
}

// ---- dead slots on the tile-binned path (the reference's default layout: max_agents = W·H slots, core/data_init.py:143-144) ----------
// A slot that never lived still acts, moves, burns and "consumes" (core/agent/gradient.py:96-124 has no alive masking;
// core/env.py:163-172 moves every slot; :224-243 feeds every slot from rate·food·(agents > 0) and sums `gained` over all of them) —
// it only never claims, deposits or marks a cell.  The binned layouts hold the alive agents in their tiles' segments, entries
// [0, n_alive), and the dead slots behind them, entries [n_alive, N), in any fixed order; per step, between the agent kernel and the
// field kernel:  k_pic_mark — the cells the alive agents stand on NOW, one byte per cell (the 'agents' channel of this step, which the
// binned path never materialises: 16 MB at 4096²) — and k_pic_dead — forward from global memory, move, feeding from the
// food plane as it still is BEFORE the field kernel's consumption, reward partial per workgroup.
__global__ __launch_bounds__(DIE_BLOCK) void k_pic_mark(PicArgs p, uint32_t n_alive, uint8_t* occ) {
    // one BYTE per cell, plain stores (every writer writes 1: no atomics — a bit per cell with atomicOr took 83 µs for 2.5 M agents,
    // neighbouring lanes hitting the same words)
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_alive; j += stride) {
        const int cx = die_cell_u(p.out.x[j], p.g.gW), cy = die_cell_u(p.out.y[j], p.g.gH);
        occ[(uint32_t)cx * (uint32_t)p.g.H + (uint32_t)cy] = 1;
    }
}

template <typename T, int KIND>
__global__ __launch_bounds__(DIE_BLOCK) void k_pic_dead(FwdArgs f, PicArgs p, uint32_t first, uint32_t count, const uint8_t* occ, long long* part, uint32_t nslots) {
    f.pgx = nullptr; f.pgy = nullptr; f.step_base = nullptr; f.mask = nullptr;
    f.inertia = 0.f; f.noise_scale = 0.f; f.normalized = 1;
    f.g = p.g;
    const FwdGlobalMem<T, false> mem(f);
    const T* food = (const T*)p.food;
    long long gsum = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint32_t j = first + i;
        uint32_t X = p.in.x[j], Y = p.in.y[j];
        const uint32_t sid = p.in.slot[j];
        const double hd = __hiloint2double((int)p.in.hhi[j], (int)p.in.hlo[j]);
        const FwdOut o = die_forward_agent_mem<T, KIND, false, FwdGlobalMem<T, false>, true>(f, mem, X, Y, hd, sid, (int64_t)j);
        if (p.adx) { p.adx[j] = o.dx; p.ady[j] = o.dy; p.adep[j] = o.dep; }
        if (p.boundary == DIE_BOUNDARY_WRAP) {
            X += (uint32_t)die_q32_small(o.dx);
            Y += (uint32_t)die_q32_small(o.dy);
        } else {
            const int64_t qx = (int64_t)X + die_q32_small(o.dx), qy = (int64_t)Y + die_q32_small(o.dy);
            X = (uint32_t)(qx < 0 ? 0 : (qx > 0xFFFFFFFFLL ? 0xFFFFFFFFLL : qx));
            Y = (uint32_t)(qy < 0 ? 0 : (qy > 0xFFFFFFFFLL ? 0xFFFFFFFFLL : qy));
        }
        const int cx = die_cell_u(X, p.g.gW), cy = die_cell_u(Y, p.g.gH);
        const uint32_t c = (uint32_t)cx * (uint32_t)p.g.H + (uint32_t)cy;
        // _agent_feed (core/env.py:224-233): a dead slot "consumes" iff somebody alive stands on its cell
        const bool occupied = occ[c] != 0;
        const float consumed = occupied ? p.rate_feed * die_ld(food, (int64_t)c) : 0.f;
        const float cost = p.cost == DIE_COST_LINEAR ? p.w_dep * fabsf(o.dep) + p.w_dist * die_sqrt1(o.dx * o.dx + o.dy * o.dy) : 0.f;
        const float gained = consumed - cost;
        gsum += die_fix(gained);
        p.out.x[j] = X; p.out.y[j] = Y; p.out.agent_food[j] = p.in.agent_food[j] + gained; p.out.slot[j] = sid;
        p.out.hhi[j] = (uint32_t)__double2hiint(o.heading); p.out.hlo[j] = (uint32_t)__double2loint(o.heading);
        p.dep[j] = o.dep;
    }
    __shared__ long long s_g[DIE_BLOCK / DIE_WAVE];
    gsum = die_wave_sum(gsum);
    if ((threadIdx.x & (DIE_WAVE - 1)) == 0) s_g[threadIdx.x / DIE_WAVE] = gsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < DIE_BLOCK / DIE_WAVE; ++w) t += s_g[w];
        if (gridDim.x <= nslots) part[blockIdx.x] = t;
        else atomicAdd((unsigned long long*)&part[blockIdx.x % nslots], (unsigned long long)t);      // (integers: any order; the slots were zeroed)
    }
}

// The random turn bits of one step, one Philox block per 128 slot ids (die_rng.h die_turn_word): word w of the table for
// w < words.  Launched by die_pic_forward_env_step when the table does not hold this step's bits yet; in steady state the
// field kernel's spare workgroups fill it for the NEXT step.
__device__ __forceinline__ void pic_turn_bits_fill(uint32_t* table, int64_t words, uint64_t seed, uint32_t step, int64_t first, int64_t stride) {
    for (int64_t b = first; b * 4 < words; b += stride) {
        const die_u32x4 r = die_philox((uint32_t)b, 0u, step, DIE_STREAM_TURN, (uint32_t)seed, (uint32_t)(seed >> 32));
        if (b * 4 + 3 < words) *(uint4*)(table + b * 4) = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
        else for (int q = 0; q < 4 && b * 4 + q < words; ++q) table[b * 4 + q] = r.v[q];
    }
}
__global__ __launch_bounds__(DIE_BLOCK) void k_turn_bits(uint32_t* table, int64_t words, uint64_t seed, uint32_t step) {
    pic_turn_bits_fill(table, words, seed, step, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// K2.  blockIdx.x == number of tiles: the scan workgroup (sizes and offsets of the layout the NEXT step writes:
// n = s + inc of the layout just written, exclusive scan; its arrival counters are cleared for that step).
// FEED: the tile's half of _agent_feed (core/env.py:222-228), food −= rate·food on the occupied cells, is done here — K2 is
// bound by latency and has memory bandwidth to spare, the field sweep is bound by bandwidth (PIC_K2_FEED, measured below).
template <typename T, int XS, int YS, bool FEED>
__global__ __launch_bounds__(PIC_K2_BLOCK) void k_pic_resolve(PicArgs p, float* dep_plane) {
    constexpr int TX = 1 << XS, TY = 1 << YS;
    __shared__ __align__(16) unsigned long long s_claim[TX * TY];
    __shared__ uint32_t s_base[9], s_pre[10];
    const int NT = p.ntx * p.nty;
    if ((int)blockIdx.y == p.ntx) {                                // the extra grid row: its first workgroup scans
        if (blockIdx.x != 0) return;
        __shared__ uint32_t s_sum[PIC_K2_BLOCK];
        const int per = (NT + PIC_K2_BLOCK - 1) / PIC_K2_BLOCK;
        const int lo = threadIdx.x * per, hi = min(lo + per, NT);
        uint32_t sum = 0;
        for (int t = lo; t < hi; ++t) sum += p.out.s[t] + p.out.inc[t];
        s_sum[threadIdx.x] = sum;
        __syncthreads();
        for (int o = 1; o < PIC_K2_BLOCK; o <<= 1) {
            const uint32_t v = (int)threadIdx.x >= o ? s_sum[threadIdx.x - o] : 0u;
            __syncthreads();
            s_sum[threadIdx.x] += v;
            __syncthreads();
        }
        uint32_t run = s_sum[threadIdx.x] - sum;
        for (int t = lo; t < hi; ++t) {
            const uint32_t c = p.out.s[t] + p.out.inc[t];
            p.in.off[t] = run;
            p.in.n[t] = c;
            p.in.inc[t] = 0;
            run += c;
        }
        return;
    }
    const int tx = blockIdx.y, ty = blockIdx.x;
    const PicMeta mt = pic_meta_load(p.out, tx, ty, p.ntx, p.nty);
    // FEED: this thread's 4-cell groups of the food tile, requested now, needed after the claims are in
    constexpr int V = 16 / (int)sizeof(T), FG = (TX * TY / 4 + PIC_K2_BLOCK - 1) / PIC_K2_BLOCK;
    float fd[FG][4];
    if (FEED) {
        T* food = (T*)p.food;
#pragma unroll
        for (int q = 0; q < FG; ++q) {
            const int i = ((int)threadIdx.x + q * PIC_K2_BLOCK) * 4;
            if (i < TX * TY) { const int row = i / TY, col = i - row * TY; Vec4<T>::ld(food + (int64_t)(tx * TX + row) * p.g.H + ty * TY + col, fd[q]); }
        }
        (void)V;
    }
    for (int i = threadIdx.x; i < TX * TY / 2; i += PIC_K2_BLOCK) ((ulonglong2*)s_claim)[i] = make_ulonglong2(0ull, 0ull);
    pic_ranges_finish(mt, s_base, s_pre);                          // (its barriers also cover the zeroing)
    const uint32_t total = s_pre[9], own = s_pre[1];
    // two agents per thread and trip (≈ 615 stand on a 64×64 tile, 512 threads): their eight loads are in flight together
    // instead of one round trip after the other
    for (uint32_t i0 = threadIdx.x; i0 < total; i0 += 2 * PIC_K2_BLOCK) {
        uint32_t X[2], Y[2], sid[2], db[2];
        bool has[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t idx = i0 + u * PIC_K2_BLOCK;
            has[u] = idx < total;
            if (has[u]) {
                int r = 0;
                while (idx >= s_pre[r + 1]) ++r;
                const uint32_t j = s_base[r] + (idx - s_pre[r]);
                X[u] = p.out.x[j]; Y[u] = p.out.y[j]; sid[u] = p.out.slot[j]; db[u] = __float_as_uint(p.dep[j]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!has[u]) continue;
            const uint32_t idx = i0 + u * PIC_K2_BLOCK;
            const int cx = die_cell((int64_t)X[u], p.g.gW), cy = die_cell((int64_t)Y[u], p.g.gH);
            if (idx >= own && ((cx >> XS) != tx || (cy >> YS) != ty)) continue;
            // highest slot on the cell wins (core/env.py:211), its deposit rides in the low word
            atomicMax(&s_claim[(cx & (TX - 1)) * TY + (cy & (TY - 1))], ((unsigned long long)(sid[u] + 1u) << 32) | (unsigned long long)db[u]);
        }
    }
    __syncthreads();
    static_assert(TY % 4 == 0, "16-byte stores");
#pragma unroll
    for (int g = 0; g < FG; ++g) {
        const int i = ((int)threadIdx.x + g * PIC_K2_BLOCK) * 4;
        if (i >= TX * TY) break;
        const int row = i / TY, col = i - row * TY;
        const ulonglong2 c01 = *(const ulonglong2*)&s_claim[i], c23 = *(const ulonglong2*)&s_claim[i + 2];
        const unsigned long long c[4] = {c01.x, c01.y, c23.x, c23.y};
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = c[q] ? (uint32_t)c[q] : DIE_DEP_EMPTY;
        const int64_t off = (int64_t)(tx * TX + row) * p.g.H + ty * TY + col;
        *(uint4*)(dep_plane + off) = make_uint4(o[0], o[1], o[2], o[3]);
        if (FEED && (c[0] | c[1] | c[2] | c[3])) {
            bool changed = false;                               // (cells without food stay as they are: an unchanged group is not written)
#pragma unroll
            for (int q = 0; q < 4; ++q) if (c[q] && fd[g][q] != 0.f) { fd[g][q] = fd[g][q] - p.rate_feed * fd[g][q]; changed = true; }
            if (changed) Vec4<T>::st((T*)p.food + off, fd[g]);
        }
    }
}

// ---- two-launch form: K2 and the field sweep in one kernel per tile ------------------------------------------------------
// k_pic_resolve_diffuse, one workgroup per tile d over a window of the tile ± R cells (R = gaussian radius):
//   claims    32-bit LDS atomicMax of slot + 1 per window cell — "the highest alive slot on a cell writes" (core/env.py:211) —
//             for the agents standing on d (its stayers + the neighbours' leavers that landed on it, as in k_pic_resolve) AND
//             for the agents standing within R cells outside d (their deposits reach d's cells through the gaussian).  Both
//             kinds are found in the rim lists the agent kernel left for the 9 tiles around d (per tile: its agents that
//             walked off it or stand near a border, with a code that says where) — lists at fixed places, requested at the
//             very top with everything else that depends on the tile index only; a list that overflowed: that tile's
//             segment is scanned by coordinates instead (slow, correct);
//   deposit   every such agent looks its cell up again; the winner adds its deposit to the chem window staged in LDS
//             (chem[c] = chem[c] + deposit, core/env.py:211: one writer per cell, no atomic);
//   feeding   food −= rate·food on d's occupied cells (core/env.py:222-228), 16-byte read-modify-write of the groups that hold one;
//   diffusion the separable gaussian over the window: axis 0 first like scipy (symmetric pairs summed first, outermost pair
//             first: the arithmetic of k_diffuse_rows, die_env.hip, bit for bit), × (1 − decay) (core/env.py:136-145), 16-byte
//             stores of d's cells of chem_next.
// No deposit plane: the 67 MB written by K2 and read back (with halo rows) by the sweep at 4096² are gone, and so is a
// launch.  Two extra workgroups: the scan of the next step's segment sizes (as in k_pic_resolve) and the reduction of the
// agent kernel's reward partials (as in the sweep).
struct KbArgs {
    const void* chem;
    void* chem_next;
    const uint4* rim;
    const uint8_t* rim_code;
    const uint32_t* rim_cnt;
    int food_infinite;
    float keep;
    float w[2 * 4 + 1];
    die_step_result* result;
    long long alive_const;
    long long* status_out;                                  // where the reduction workgroup copies the error word (die_pic.status_out), or NULL
    const uint32_t* error;
    int nt_out;                                             // the chem and food stores are non-temporal (die_pic_forward_env_step decides by the state's size)
    int n_part;                                             // reward partials in part_gain: one per tile, + (dead slots) one per workgroup of k_pic_dead
    uint32_t* turn_bits;                                    // the NEXT step's random turn bits (pic_turn_bits_fill), or NULL
    int64_t turn_words;
    uint64_t turn_seed;
    uint32_t turn_step;
};

template <int XS, int YS> struct KbShape {
    static constexpr int BLOCK = ((1 << XS) * (1 << YS) >= 4096) ? 512 : 256;
    static constexpr int NE = 4;                                    // rim-list entries per thread: the 4 code bytes of one word
    static constexpr int RIM_CAP = NE * BLOCK / 9 / 4 * 4;          // 224 / 112 (a multiple of 4: a word never straddles two lists)
    static_assert(RIM_CAP <= PIC_RIM_CAP_MAX, "the agent kernel's list holds it");
};

// (The four barriers below as LDS-only barriers — s_waitcnt lgkmcnt(0) + s_barrier, no wait for the food stores' acknowledgements —
// measured in round 5: 58.4–59.6 against 58.6–60.2 µs, nothing.)
#ifndef PIC_KB_MINW
#define PIC_KB_MINW 8           // 4 workgroups of 512 threads per CU (A/B: scratch/build_variant.sh)
#endif
template <typename T, int XS, int YS, int R, bool TILED>
__global__ __launch_bounds__((KbShape<XS, YS>::BLOCK), PIC_KB_MINW) void k_pic_resolve_diffuse(PicArgs p, KbArgs a) {
    PIC_ENTRY_DECL;
    constexpr int TX = 1 << XS, TY = 1 << YS, BLOCK = KbShape<XS, YS>::BLOCK;
    constexpr int A = 16 / (int)sizeof(T);                 // cells per 16-byte vector
    constexpr int WR = TX + 2 * R, WC = TY + 2 * R;        // the window
    constexpr int CP = TY + 2 * A, NV = CP / A;            // staged columns [y0 − A, y0 + TY + A): whole vectors; vectors per row
    static_assert(R >= 1 && R <= 4 && R <= A, "the rim lies inside one vector beside the tile");
    extern __shared__ __align__(16) unsigned char kb_smem[];
    float* s_chem = (float*)kb_smem;                        // WR × CP, window cell (r, c) at r·CP + c − R + A
    uint32_t* s_claim = (uint32_t*)(s_chem + WR * CP);      // WR × WC; after the deposits: s_tmp, TX × CP (the x pass)
    float* s_tmp = (float*)s_claim;
    // the 9 tiles around d, [0] d itself, [1..8] the ring: segment in layout `out`, rim list
    __shared__ uint32_t s_off[9], s_n[9], s_rn[9];          // segment offset, size; entries of the rim list
    __shared__ uint32_t s_own[2];                           // d's stayers: first index, count
    __shared__ unsigned char s_lut[9 * 81 + 3];             // (list, code) → (ux + 1)·3 + uy + 1 of an agent that matters here, else 0xFF
    constexpr int CAPR = KbShape<XS, YS>::RIM_CAP, NE = KbShape<XS, YS>::NE;
    const int NT = p.ntx * p.nty;
    // The extra grid row — next offsets, reward, turn bits: work that depends on the agent kernel only — is the FIRST row of the grid:
    // its workgroups are dispatched ahead of the tiles', so the step's result reaches an Env(sync=True) caller (pinned host memory)
    // while the tiles are still being swept, and that caller's next launches are queued before this kernel ends.  (As the last row,
    // round 4, the result arrived with the kernel's end and every sync=True step paid a launch latency: 6 300 against 7 300 steps/s.)
    const uint32_t row0 = p.sub_mode == 1 ? 0u : 1u;        // (a launch over a rectangle of tiles has no extra row)
    if (row0 && blockIdx.y == 0) {
        if (blockIdx.x == 0) {                              // sizes and offsets of the layout the NEXT step writes
            uint32_t* s_sum = (uint32_t*)kb_smem;           // BLOCK words (the window arrays are not used by this workgroup)
            const int per = (NT + BLOCK - 1) / BLOCK;
            const int lo = threadIdx.x * per, hi = min(lo + per, NT);
            uint32_t sum = 0;
            for (int t = lo; t < hi; ++t) sum += p.out.s[t] + p.out.inc[t];
            s_sum[threadIdx.x] = sum;
            __syncthreads();
            for (int o = 1; o < BLOCK; o <<= 1) {
                const uint32_t v = (int)threadIdx.x >= o ? s_sum[threadIdx.x - o] : 0u;
                __syncthreads();
                s_sum[threadIdx.x] += v;
                __syncthreads();
            }
            uint32_t run = s_sum[threadIdx.x] - sum;
            for (int t = lo; t < hi; ++t) {
                const uint32_t c = p.out.s[t] + p.out.inc[t];
                p.in.off[t] = run;
                p.in.n[t] = c;
                p.in.inc[t] = 0;
                run += c;
            }
        } else if (blockIdx.x == 1 && a.result) {           // reward: the agent kernel's per-tile partials (integers: any order)
            long long* s_g = (long long*)kb_smem;           // BLOCK 64-bit words
            long long t = 0;
            for (int base = threadIdx.x; base < a.n_part; base += BLOCK * 8) {
                long long v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const int i = base + q * BLOCK; v[q] = i < a.n_part ? p.part_gain[i] : 0; }
#pragma unroll
                for (int q = 0; q < 8; ++q) t += v[q];
            }
            s_g[threadIdx.x] = t;
            __syncthreads();
            for (int o = BLOCK / 2; o > 0; o >>= 1) {
                if ((int)threadIdx.x < o) s_g[threadIdx.x] += s_g[threadIdx.x + o];
                __syncthreads();
            }
            long long alive = a.alive_const;
            if (TILED) {                                    // agents standing on cells this rank accounts for (counted by the agent kernel)
                const long long g0 = s_g[0];
                __syncthreads();
                long long c = 0;
                for (int i = threadIdx.x; i < NT; i += BLOCK) c += p.part_gain[(size_t)NT + i];
                s_g[threadIdx.x] = c;
                __syncthreads();
                for (int o = BLOCK / 2; o > 0; o >>= 1) {
                    if ((int)threadIdx.x < o) s_g[threadIdx.x] += s_g[threadIdx.x + o];
                    __syncthreads();
                }
                alive = s_g[0];
                __syncthreads();
                if (threadIdx.x == 0) s_g[0] = g0;
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                die_store_result_f64(&a.result->reward, (double)s_g[0] / DIE_FIX_ONE); die_store_result_i64((long long*)&a.result->num_alive, alive);
                if (a.status_out) die_store_result_i64(a.status_out, (long long)*a.error);     // (set by the agent kernel: a kernel boundary lies in between)
            }
        } else if (blockIdx.x >= 2 && a.turn_bits) {        // the rest of the row: the next step's turn bits (this step's agent kernel is done with the table)
            pic_turn_bits_fill(a.turn_bits, a.turn_words, a.turn_seed, a.turn_step, (int64_t)(blockIdx.x - 2) * BLOCK + threadIdx.x,
                               (int64_t)(gridDim.x - 2) * BLOCK);
        }
        return;
    }
    int tx = (int)(blockIdx.y - row0), ty = blockIdx.x;
#ifndef PIC_XCD_MAP_KB
#define PIC_XCD_MAP_KB 1        // 2: bands walked from their far end (does an XCD's L2 keep the agent kernel's last tiles across the kernel boundary? no: same counters)
#endif
    bool by_table = false;
    if (p.order && !pic_order_tile(p, (blockIdx.y - row0) * (uint32_t)p.nty + blockIdx.x, tx, ty, by_table)) return;
    if (!by_table && PIC_XCD_MAP_KB && p.sub_mode == 0) pic_xcd_tile<PIC_XCD_MAP_KB == 2>(tx, ty, p.ntx, row0, p.xcd_wb_mul, (uint32_t)p.nty);
    if (!pic_sub_tile(p, tx, ty)) return;
    const int x0 = tx << XS, y0 = ty << YS;
    const int W = p.g.W, H = p.g.H;
    [[maybe_unused]] const int tile = tx * p.nty + ty;
    PIC_ENTRY_STORE(1);
    PIC_STAMP(8);
    PIC_SETPRIO(PIC_PRIO_KB, 0);
    const T* chem = (const T*)a.chem;
    T* food = (T*)p.food;
    // 1. everything that depends on the tile index only goes out first: the per-tile words, the rim lists, the chem window,
    //    the food tile
    auto ring_tile = [&](int q) {
        const int k3 = q == 0 ? 4 : (q <= 4 ? q - 1 : q);
        return pic_wrap(tx + k3 / 3 - 1, p.ntx) * p.nty + pic_wrap(ty + k3 % 3 - 1, p.nty);
    };
    uint32_t m_o = 0, m_s = 0, m_n = 0, m_r = 0;
    if (threadIdx.x < 9) {
        const int t = ring_tile(threadIdx.x);
        m_o = p.out.off[t]; m_s = p.out.s[t]; m_n = p.out.n[t]; m_r = a.rim_cnt[t];      // (not touched before the other loads are out)
    }
    // the codes of rim entries 4·tid .. 4·tid + 3 of the flattened 9 × CAPR (list e / CAPR, entry e % CAPR; whether an entry
    // exists is known once the counts are here)
    static_assert(NE == 4 && CAPR % 4 == 0 && 9 * CAPR <= 4 * BLOCK, "one word of codes per thread");
    const int rl = 4 * (int)threadIdx.x / CAPR, ri = 4 * (int)threadIdx.x - rl * CAPR;
    const uint32_t rcodes = rl < 9 ? pic_ld<3>(&((const uint32_t*)a.rim_code)[((size_t)ring_tile(rl) * CAPR + ri) / 4]) : 0u;
    constexpr int NCV = (WR * NV + BLOCK - 1) / BLOCK;
    static_assert(NCV <= 3, "three window vectors per thread at most");       // (named registers: as an array they went to scratch)
    auto window_load = [&](int q) {
        const int i = min((int)threadIdx.x + q * BLOCK, WR * NV - 1);       // (surplus threads load the last vector again: no branch)
        const int r = i / NV, v = i - r * NV;
        return pic_ld4<8>(chem + ((int64_t)pic_wrap(x0 - R + r, W) * H + pic_wrap(y0 - A + v * A, H)));
    };
    // (Round 5, both measured and dropped: the tile's own stayers requested right here, from scalar-loaded segment words — a barrier
    // and an LDS hand-over earlier than the claims pass: 59.5–60.0 against 57.9–58.6 µs; the four barriers below as LDS-only barriers,
    // no wait for the food stores' acknowledgements: 58.4–59.6 against 58.6–60.2.  A shorter chain per workgroup buys nothing.)
    uint4 cv0 = window_load(0), cv1 = cv0, cv2 = cv0;
    if constexpr (NCV > 1) cv1 = window_load(1);
    if constexpr (NCV > 2) cv2 = window_load(2);
    PIC_SETPRIO(PIC_PRIO_KB, 1);
    for (int i = threadIdx.x; i < WR * WC / 4; i += BLOCK) ((uint4*)s_claim)[i] = make_uint4(0u, 0u, 0u, 0u);
    static_assert((WR * WC) % 4 == 0, "16-byte zeroing");
    if (threadIdx.x < 9) {
        if (m_s > m_n) m_s = m_n = 0;                       // (broken bookkeeping — never loop over garbage)
        s_off[threadIdx.x] = m_o; s_n[threadIdx.x] = m_n; s_rn[threadIdx.x] = m_r;
        if (threadIdx.x == 0) { s_own[0] = m_o; s_own[1] = m_s; }
    }
    // (while the loads fly) which (list, code) pairs matter to d, and from which tile relative to d
    for (int t = threadIdx.x; t < 9 * 81; t += BLOCK) {
        const int q = t / 81, c = t - q * 81, k3 = q == 0 ? 4 : (q <= 4 ? q - 1 : q), dd = c / 9, e = c - dd * 9;
        int ux = k3 / 3 - 1 + dd / 3 - 1, uy = k3 % 3 - 1 + dd % 3 - 1;   // the agent's tile relative to d, in [−2, 2] …
        ux += ux < 0 ? p.ntx : 0; ux -= 2 * ux > p.ntx ? p.ntx : 0;        // … as the nearest periodic image (3 tiles: 2 ≡ −1)
        uy += uy < 0 ? p.nty : 0; uy -= 2 * uy > p.nty ? p.nty : 0;
        const int sx = e / 3 - 1, sy = e % 3 - 1;                          // which neighbours of its tile it stands close to
        const bool here = (ux == 0 || ux == -sx) && (uy == 0 || uy == -sy) && !(q == 0 && dd == 4);   // (d's stayers: taken directly)
        s_lut[t] = here ? (unsigned char)((ux + 1) * 3 + uy + 1) : (unsigned char)0xFF;
    }
    __syncthreads();
    PIC_STAMP(9);
    // the chem window into LDS (as float): requested first, so it is here when the per-tile words are — its registers are
    // free for the agents' data
    auto window_commit = [&](int q, const uint4 u) {
        const int i = (int)threadIdx.x + q * BLOCK;
        if (i >= WR * NV) return;
        const int r = i / NV, v = i - r * NV;
        float* dst = s_chem + r * CP + v * A;
        if constexpr (sizeof(T) == 4) {
            *(uint4*)dst = u;
        } else {
            auto lo = [](uint32_t w) { return __half2float(__ushort_as_half((unsigned short)(w & 0xFFFFu))); };
            auto hi = [](uint32_t w) { return __half2float(__ushort_as_half((unsigned short)(w >> 16))); };
            *(float4*)dst = make_float4(lo(u.x), hi(u.x), lo(u.y), hi(u.y));
            *(float4*)(dst + 4) = make_float4(lo(u.z), hi(u.z), lo(u.w), hi(u.w));
        }
    };
    window_commit(0, cv0);
    if constexpr (NCV > 1) window_commit(1, cv1);
    if constexpr (NCV > 2) window_commit(2, cv2);
    const uint32_t own0 = s_own[0], nown = s_own[1];
    // window cell of an agent standing on the tile at (ux, uy) tiles from d; 0xFFFFFFFF: outside the window
    auto window_cell = [&](uint32_t X, uint32_t Y, int ux, int uy) {
        const int cx = pic_row<TILED>(p.g, X), cy = pic_col<TILED>(p.g, Y);
        const int r = ux * TX + (cx & (TX - 1)) + R, c = uy * TY + (cy & (TY - 1)) + R;
        return (r >= 0 && r < WR && c >= 0 && c < WC) ? (uint32_t)(r * WC + c) | ((uint32_t)r << 16) : 0xFFFFFFFFu;   // (row kept: no division later)
    };
    static_assert(WR * WC < (1 << 16) && WR < (1 << 15), "window cell and row in one word");
    // code byte m of this thread's word → where the record lies and the tile code, or 0xFFFFFFFF when the entry does not
    // exist / does not matter here
    auto rim_decode = [&](int m, uint32_t& uc) {
        uc = 0xFFu;
        if (rl >= 9 || (uint32_t)(ri + m) >= s_rn[rl] || s_rn[rl] > (uint32_t)CAPR) return 0xFFFFFFFFu;
        const uint32_t code = (rcodes >> (8 * m)) & 255u;
        if (code >= 81u) return 0xFFFFFFFFu;
        uc = s_lut[rl * 81 + code];
        return uc == 0xFFu ? 0xFFFFFFFFu : (uint32_t)(ri + m);
    };
    // a tile whose list overflowed: every agent of its segment that stands in the window (for d itself: its leavers only)
    // (Crowded tiles — 3 000 agents and more on a tile once the trails have formed, rim lists long overflowed — lived 60 µs here and WERE the
    // field kernel's duration at world step 3 000: one agent per thread and trip, coordinates → cell → slot → deposit one dependent round
    // trip after the other.  PIC_KB_U agents per thread and trip, all their loads requested before the first is used: round 6.)
#ifndef PIC_KB_U
#define PIC_KB_U 4
#endif
    auto scan_segment = [&](int l, bool apply) {
        const uint32_t lo = l == 0 ? own0 + nown : s_off[l], hi = s_off[l] + s_n[l];
        for (uint32_t j0 = lo + threadIdx.x; j0 < hi; j0 += PIC_KB_U * BLOCK) {
            uint32_t X[PIC_KB_U], Y[PIC_KB_U], S[PIC_KB_U], D[PIC_KB_U];
#pragma unroll
            for (int u = 0; u < PIC_KB_U; ++u) {
                const uint32_t j = j0 + (uint32_t)u * BLOCK;
                X[u] = Y[u] = S[u] = D[u] = 0u;
                if (j < hi) { X[u] = p.out.x[j]; Y[u] = p.out.y[j]; S[u] = p.out.slot[j]; if (apply) D[u] = __float_as_uint(p.dep[j]); }
            }
#pragma unroll
            for (int u = 0; u < PIC_KB_U; ++u) {
                if (j0 + (uint32_t)u * BLOCK >= hi) continue;
                const int cx = pic_row<TILED>(p.g, X[u]), cy = pic_col<TILED>(p.g, Y[u]);
                int rr = cx - x0, rc = cy - y0;             // periodic distance to the tile's origin
                rr = rr > W / 2 ? rr - W : (rr < -(W / 2) ? rr + W : rr);
                rc = rc > H / 2 ? rc - H : (rc < -(H / 2) ? rc + H : rc);
                if (rr < -R || rr >= TX + R || rc < -R || rc >= TY + R) continue;
                const uint32_t widx = (uint32_t)((rr + R) * WC + rc + R), s1 = S[u] + 1u;
                if (!apply) atomicMax(&s_claim[widx], s1);
                else if (s_claim[widx] == s1) {
                    float* c = &s_chem[(rr + R) * CP + rc + A];
                    *c = die_as_stored<T>(*c + __uint_as_float(D[u]));
                }
            }
        }
    };
    uint32_t over = 0;
#pragma unroll
    for (int l = 0; l < 9; ++l) over |= s_rn[l] > (uint32_t)CAPR ? 1u << l : 0u;
    // 2. claims: the tile's own stayers two per thread and trip and this thread's rim entries, all their loads in flight
    //    together; the first trip and the entries stay in registers for the deposit pass
    constexpr int FG = (TX * TY / 4 + BLOCK - 1) / BLOCK;
    float fd[FG][4];
    uint32_t cw[2 + NE], cs[2 + NE], cd[2 + NE];
    {
        uint32_t X[2 + NE], Y[2 + NE], uc[NE];
        const uint32_t rj[NE] = {rim_decode(0, uc[0]), rim_decode(1, uc[1]), rim_decode(2, uc[2]), rim_decode(3, uc[3])};
        const uint4* rrec = a.rim + (size_t)ring_tile(rl < 9 ? rl : 0) * CAPR;
#pragma unroll
        for (int u = 0; u < 2 + NE; ++u) {
            cw[u] = 0xFFFFFFFFu; cs[u] = 0u; cd[u] = 0u; X[u] = Y[u] = 0u;
            if (u < 2) {
                if (threadIdx.x + u * BLOCK < nown) {
                    const uint32_t j = own0 + threadIdx.x + u * BLOCK;
                    X[u] = pic_ld<3>(&p.out.x[j]); Y[u] = pic_ld<3>(&p.out.y[j]); cs[u] = pic_ld<3>(&p.out.slot[j]) + 1u; cd[u] = __float_as_uint(pic_ld<3>(&p.dep[j])); cw[u] = 0u;
                }
            } else if (rj[u - 2] != 0xFFFFFFFFu) {
                const uint4 q = pic_ld4<3>(&rrec[rj[u - 2]]);
                X[u] = q.x; Y[u] = q.y; cs[u] = q.z + 1u; cd[u] = q.w; cw[u] = 0u;
            }
        }
        // (the food tile is needed last: requested behind the agents' data — loads return in order)
        if (!a.food_infinite) {
#pragma unroll
            for (int q = 0; q < FG; ++q) {
                const int i = ((int)threadIdx.x + q * BLOCK) * 4;
                if (i < TX * TY) { const int row = i / TY, col = i - row * TY; Vec4<T>::template ld<4>(food + (int64_t)(x0 + row) * H + y0 + col, fd[q]); }
            }
        }
#pragma unroll
        for (int u = 0; u < 2 + NE; ++u) {
            if (cw[u] == 0xFFFFFFFFu) continue;
            cw[u] = u < 2 ? window_cell(X[u], Y[u], 0, 0) : window_cell(X[u], Y[u], (int)uc[u - 2] / 3 - 1, (int)uc[u - 2] % 3 - 1);
            if (cw[u] != 0xFFFFFFFFu) atomicMax(&s_claim[cw[u] & 0xFFFFu], cs[u]);
        }
    }
    for (uint32_t i0 = threadIdx.x + 2 * BLOCK; i0 < nown; i0 += PIC_KB_U * BLOCK) {         // (tiles of more than 1 024 stayers)
        uint32_t X[PIC_KB_U], Y[PIC_KB_U], S[PIC_KB_U];
#pragma unroll
        for (int u = 0; u < PIC_KB_U; ++u) {
            const uint32_t i = i0 + (uint32_t)u * BLOCK, j = own0 + i;
            X[u] = Y[u] = S[u] = 0u;
            if (i < nown) { X[u] = p.out.x[j]; Y[u] = p.out.y[j]; S[u] = p.out.slot[j]; }
        }
#pragma unroll
        for (int u = 0; u < PIC_KB_U; ++u) {
            if (i0 + (uint32_t)u * BLOCK >= nown) continue;
            const uint32_t w_ = window_cell(X[u], Y[u], 0, 0);
            if (w_ != 0xFFFFFFFFu) atomicMax(&s_claim[w_ & 0xFFFFu], S[u] + 1u);
        }
    }
    if (over) for (int l = 0; l < 9; ++l) if (over >> l & 1u) scan_segment(l, false);
    PIC_STAMP(10);
    __syncthreads();
    PIC_STAMP(11);
    // 4. deposits of the winners; feeding of the tile's occupied cells
    auto deposit = [&](uint32_t widx, uint32_t s1, uint32_t db) {
        if (widx == 0xFFFFFFFFu || s_claim[widx & 0xFFFFu] != s1) return;
        const uint32_t r = widx >> 16, c = (widx & 0xFFFFu) - r * WC;
        float* q = &s_chem[r * CP + c - R + A];
        *q = die_as_stored<T>(*q + __uint_as_float(db));
    };
#pragma unroll
    for (int u = 0; u < 2 + NE; ++u) deposit(cw[u], cs[u], cd[u]);
    for (uint32_t i0 = threadIdx.x + 2 * BLOCK; i0 < nown; i0 += PIC_KB_U * BLOCK) {
        uint32_t X[PIC_KB_U], Y[PIC_KB_U], S[PIC_KB_U], D[PIC_KB_U];
#pragma unroll
        for (int u = 0; u < PIC_KB_U; ++u) {
            const uint32_t i = i0 + (uint32_t)u * BLOCK, j = own0 + i;
            X[u] = Y[u] = S[u] = D[u] = 0u;
            if (i < nown) { X[u] = p.out.x[j]; Y[u] = p.out.y[j]; S[u] = p.out.slot[j]; D[u] = __float_as_uint(p.dep[j]); }
        }
#pragma unroll
        for (int u = 0; u < PIC_KB_U; ++u)
            if (i0 + (uint32_t)u * BLOCK < nown) deposit(window_cell(X[u], Y[u], 0, 0), S[u] + 1u, D[u]);
    }
    if (over) for (int l = 0; l < 9; ++l) if (over >> l & 1u) scan_segment(l, true);
    if (!a.food_infinite) {
#pragma unroll
        for (int g = 0; g < FG; ++g) {
            const int i = ((int)threadIdx.x + g * BLOCK) * 4;
            if (i >= TX * TY) break;
            const int row = i / TY, col = i - row * TY;
            const uint32_t* c = &s_claim[(row + R) * WC + col + R];
            // (an occupied cell without food stays as it is — 0 − rate·0: about half of the benchmark world's cells — and a group in
            // which nothing changes is not written: the field kernel's time follows its WRITTEN bytes, 59.6–60.9 → 56.7–57.0 µs;
            // requesting the food only for the groups that hold an occupied cell, i.e. half the loads, changed nothing)
            const bool occ[4] = {c[0] != 0u && fd[g][0] != 0.f, c[1] != 0u && fd[g][1] != 0.f, c[2] != 0u && fd[g][2] != 0.f, c[3] != 0u && fd[g][3] != 0.f};
            if (occ[0] || occ[1] || occ[2] || occ[3]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (occ[q]) fd[g][q] = fd[g][q] - p.rate_feed * fd[g][q];
                Vec4<T>::template st_sel<9>(a.nt_out != 0, food + (int64_t)(x0 + row) * H + y0 + col, fd[g]);
            }
        }
    }
    __syncthreads();
    PIC_STAMP(12);
    PIC_SETPRIO(PIC_PRIO_KB, 2);
    // 5. x pass (axis 0): column c of the window, RB output rows per item, the 2R + 1 rows of the stencil in registers
    constexpr int RB = TX >= 64 ? 16 : 8;
    for (int item = threadIdx.x; item < WC * (TX / RB); item += BLOCK) {
        const int rb = item / WC, c = item - rb * WC;
        const float* src = s_chem + (rb * RB) * CP + c - R + A;
        float win[2 * R + 1];
#pragma unroll
        for (int k = 0; k < 2 * R; ++k) win[k + 1] = src[k * CP];
#pragma unroll
        for (int o = 0; o < RB; ++o) {
#pragma unroll
            for (int k = 0; k < 2 * R; ++k) win[k] = win[k + 1];
            win[2 * R] = src[(o + 2 * R) * CP];
            float t = a.w[R] * win[R];                                       // centre, then symmetric pairs from the outermost
#pragma unroll                                                               // inwards (k_diffuse_rows: identical sums)
            for (int k = 0; k < R; ++k) t += (win[k] + win[2 * R - k]) * a.w[k];
            s_tmp[(rb * RB + o) * CP + c - R + A] = t;
        }
    }
    __syncthreads();
    PIC_STAMP(13);
    // 6. y pass (axis 1), decay, 16-byte (fp16: 8-byte) stores of the tile's cells
    T* dst = (T*)a.chem_next;
    constexpr int LO = A - R, LA = LO & ~3, NX = (LO - LA + 4 + 2 * R + 3) / 4;    // aligned float4 reads around the 4 cells
    for (int item = threadIdx.x; item < TX * (TY / 4); item += BLOCK) {
        const int row = item / (TY / 4), cg = item - row * (TY / 4);
        float xf[NX * 4];
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const float4 v = *(const float4*)&s_tmp[row * CP + LA + 4 * cg + 4 * q];
            xf[4 * q] = v.x; xf[4 * q + 1] = v.y; xf[4 * q + 2] = v.z; xf[4 * q + 3] = v.w;
        }
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cc = LO - LA + R + j;                                  // this cell in xf
            float t = a.w[R] * xf[cc];
#pragma unroll
            for (int k = 0; k < R; ++k) t += (xf[cc - R + k] + xf[cc + R - k]) * a.w[k];
            o[j] = t * a.keep;
        }
        Vec4<T>::template st_sel<6>(a.nt_out != 0, dst + (int64_t)(x0 + row) * H + y0 + 4 * cg, o);
    }
    PIC_STAMP(14);
}

// ---- order table (die_pic.order): which workgroup of a launch takes which tile --------------------------------------
// One workgroup per XCD band j (pic_xcd_tile: columns [j·wb, (j + 1)·wb) of tiles, walked row of tiles by row of tiles): the band's
// tiles sorted by the number of 8-wave rounds their population costs the agent kernel, descending, band order among equals (a stable
// counting sort over 8 classes) — the crowded tiles, and with them the tiles whose rim lists overflow in the field kernel, start
// first; neighbouring tiles of one class still run side by side in one L2.  n = populations of the layout the coming step reads.
#define PIC_ORDER_BLOCK 512
#ifndef PIC_ORDER_PERIOD
#define PIC_ORDER_PERIOD 32     // steps between two rebuilds of the table (an agent walks 3/4 of a tile meanwhile; 8 / 16 / 32: the same rates, the rebuild is 4.8 µs + a kernel boundary)
#endif
// Only the LAST PIC_ORDER_SPAN tiles of a band are sorted; the tiles ahead of them keep the band order.  What matters is that no crowded
// tile sits in a launch's last rounds — a crowded tile in the middle of a launch delays nobody —, and tiles that are neighbours in space
// should stay neighbours in time (they share the margins of their windows in one L2): with their bands of 2 048 / 8 192 tiles sorted as
// a whole, or span by span, an 8192² / 16384² world LOST 2–3 % at the bench's window, where the 512-tile bands of 4096² gained.
__global__ __launch_bounds__(PIC_ORDER_BLOCK) void k_pic_order(const uint32_t* n, int ntx, int nty, uint16_t* order) {
    __shared__ uint32_t s_cnt[8][PIC_ORDER_BLOCK];         // [class][thread]: tiles of that class in this thread's stretch of the span
    __shared__ uint32_t s_first[8];                        // first place of a class
    __shared__ uint32_t s_crowded;                         // tiles of four and more rounds in all eight bands' last spans
    const int j = blockIdx.x, wb = nty >> 3, blen = wb * ntx;
    // spans are counted from the band's END: the last one is whole, the first one takes what is left
    const int q1 = blen - (int)(gridDim.y - 1 - blockIdx.y) * PIC_ORDER_SPAN, q0 = max(q1 - PIC_ORDER_SPAN, 0), len = q1 - q0;
    const int per = (len + PIC_ORDER_BLOCK - 1) / PIC_ORDER_BLOCK;
    const int lo = q0 + min((int)threadIdx.x * per, len), hi = min(lo + per, q0 + len);
    auto tile_at = [&](int q) { return (q / wb) * nty + j * wb + q % wb; };
    if (blockIdx.y + 1 != gridDim.y) {                      // not the band's last span: band order
        for (int q = lo; q < hi; ++q) order[(size_t)j * blen + q] = (uint16_t)tile_at(q);
        return;
    }
#ifndef PIC_ORDER_MIN_CROWDED
#define PIC_ORDER_MIN_CROWDED 96 // tiles of four and more rounds, per 4 096 tiles of the eight bands' last spans TOGETHER, from which every band's span is sorted (A/B: 0 = always)
#endif
#ifndef PIC_ORDER_RMIN
#define PIC_ORDER_RMIN 0        // A/B: tiles of at most this many rounds count as one class (band order among them) …
#endif
#ifndef PIC_ORDER_RMAX
#define PIC_ORDER_RMAX 7        // … and so do tiles of at least this many
#endif
    auto cls = [&](int t) { const uint32_t r = ((n[t] + 63u) / 64u + 7u) / 8u; return 7 - (int)min(max(r, (uint32_t)PIC_ORDER_RMIN), (uint32_t)PIC_ORDER_RMAX); };      // 0: the most crowded
    uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = lo; q < hi; ++q) {
        const int k = cls(tile_at(q));
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] += i == k ? 1u : 0u;      // (no dynamically indexed registers: they would go to scratch)
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s_cnt[i][threadIdx.x] = c[i];
    __syncthreads();
    {                                                       // exclusive scan of every class's row: wave w takes class w, 64 entries per trip
        static_assert(PIC_ORDER_BLOCK / DIE_WAVE == 8, "one wave per class");
        const int w = threadIdx.x / DIE_WAVE, lane = threadIdx.x & (DIE_WAVE - 1);
        uint32_t run = 0;
        for (int b = 0; b < PIC_ORDER_BLOCK; b += DIE_WAVE) {
            const uint32_t v = s_cnt[w][b + lane];
            uint32_t x = v;
#pragma unroll
            for (int o = 1; o < DIE_WAVE; o <<= 1) { const uint32_t u = __shfl_up(x, o, DIE_WAVE); if (lane >= o) x += u; }
            s_cnt[w][b + lane] = run + x - v;
            run += __shfl(x, DIE_WAVE - 1, DIE_WAVE);
        }
        if (lane == 0) s_first[w] = run;                    // (the class's total, for now)
    }
    __syncthreads();
    // Spans without crowds stay in band order: sorting costs the L2 sharing of neighbouring tiles' windows (counted traffic of a step
    // 567 → 613 MB at the bench's window) and pays once enough tiles of four and more rounds — more than 1 536 agents: where the agent
    // kernel's workgroups live longest and the field kernel's rim lists overflow — could end up in the tail.  The bench world, 32-step
    // runs without / with every band sorted (profiles/r06_order_table_shipped.txt): 79–90 such tiles (world steps 416–544) 134.4 / 136.1 µs
    // a step; 105–138 (steps 704–832) 140–147 / 141–139; 237–251 (steps 1 024–1 152) 145 / 138; 234 (step 3 000) 160 / 138.5.
    // All eight bands sort, or none: a launch lasts as long as its slowest band, so a band sorted next to one left alone pays the
    // sorting's cost without its gain (a threshold per band: −1…−1.7 % at world steps 400–700, where some bands passed it).  Every
    // band's workgroup therefore counts the crowded tiles of all eight last spans itself (8 × 512 populations, out of L2).
    if (threadIdx.x == 0) s_crowded = 0;
    __syncthreads();
    {
        uint32_t cr = 0;
        for (int jj = 0; jj < 8; ++jj)
            for (int q = lo; q < hi; ++q) cr += ((n[(q / wb) * nty + jj * wb + q % wb] + 63u) / 64u + 7u) / 8u >= 4u ? 1u : 0u;
        if (cr) atomicAdd(&s_crowded, cr);
    }
    __syncthreads();
    const bool sorted = (uint64_t)s_crowded * 4096u >= (uint64_t)PIC_ORDER_MIN_CROWDED * 8u * (uint32_t)len;
    if (!sorted) {
        for (int q = lo; q < hi; ++q) order[(size_t)j * blen + q] = (uint16_t)tile_at(q);
        return;
    }
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 8; ++i) { const uint32_t v = s_first[i]; s_first[i] = run; run += v; }
    }
    __syncthreads();
    uint32_t at[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) at[i] = s_first[i] + s_cnt[i][threadIdx.x];
    for (int q = lo; q < hi; ++q) {
        const int t = tile_at(q), k = cls(t);
        uint32_t place = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) if (i == k) { place = at[i]; at[i] += 1u; }
        order[(size_t)j * blen + q0 + place] = (uint16_t)t;
    }
}

// ---- (re)binning: any order of the agent arrays → a layout with every agent a stayer ---------------------------
struct PicBinArgs {
    die_geo g;
    int tiled;                      // the planes are a tile of a decomposed world: bin by the plane cell (pic_row / pic_col)
    int64_t N;
    int nty, xs, ys;
    const uint32_t *x, *y, *slot;
    const float* agent_food;
    const uint32_t *hhi, *hlo;
    PicLayout out;
    uint32_t* cursor;
    const uint8_t* alive;           // NULL: every entry is alive.  Dead slots (the reference's default layout: N = W·H slots of which
    uint32_t n_alive;               // 85 % never lived, core/data_init.py:143-144) go behind the tiles' segments, entries [n_alive, N)
    uint32_t* dead_cursor;
    const float *pgx, *pgy;         // GradientAgent with inertia: _prev_grad travels along (NULL: none)
    float *opgx, *opgy;
};

__global__ __launch_bounds__(DIE_BLOCK) void k_pic_hist(PicBinArgs a, uint32_t* hist) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x; b < a.N; b += stride) {
        const int64_t n = b + threadIdx.x;
        const bool active = n < a.N && (!a.alive || a.alive[n]);
        uint32_t key = 0;
        if (active) key = a.tiled ? (uint32_t)((pic_row<true>(a.g, a.x[n]) >> a.xs) * a.nty + (pic_col<true>(a.g, a.y[n]) >> a.ys))
                                  : (uint32_t)((pic_row<false>(a.g, a.x[n]) >> a.xs) * a.nty + (pic_col<false>(a.g, a.y[n]) >> a.ys));
        wave_grouped_add<false>(hist, key, active);
    }
}

// one workgroup: off = exclusive scan of the tile counts; both layouts start with the same segments, all stayers
__global__ __launch_bounds__(1024) void k_pic_bin_scan(const uint32_t* hist, int NT, PicLayout a, PicLayout b, uint32_t* cursor,
                                                       uint32_t* error) {
    __shared__ uint32_t s[1024];
    const int per = (NT + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(lo + per, NT);
    uint32_t sum = 0;
    for (int t = lo; t < hi; ++t) sum += hist[t];
    s[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t v = (int)threadIdx.x >= o ? s[threadIdx.x - o] : 0u;
        __syncthreads();
        s[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = s[threadIdx.x] - sum;
    for (int t = lo; t < hi; ++t) {
        const uint32_t c = hist[t];
        cursor[t] = run;
        a.off[t] = run; a.n[t] = c; a.s[t] = c; a.inc[t] = 0;
        b.off[t] = run; b.n[t] = c; b.s[t] = c; b.inc[t] = 0;
        run += c;
    }
}

__global__ __launch_bounds__(DIE_BLOCK) void k_pic_scatter(PicBinArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x; b < a.N; b += stride) {
        const int64_t n = b + threadIdx.x;
        const bool inside = n < a.N, active = inside && (!a.alive || a.alive[n]);
        const uint32_t X = inside ? a.x[n] : 0u, Y = inside ? a.y[n] : 0u;
        uint32_t key = 0;
        if (active) key = a.tiled ? (uint32_t)((pic_row<true>(a.g, X) >> a.xs) * a.nty + (pic_col<true>(a.g, Y) >> a.ys))
                                  : (uint32_t)((pic_row<false>(a.g, X) >> a.xs) * a.nty + (pic_col<false>(a.g, Y) >> a.ys));
        uint32_t j = wave_grouped_add<true>(a.cursor, key, active);
        // a dead slot: any free entry behind the segments (results are keyed by the slot id, not by the array order)
        const uint32_t jd = wave_grouped_add<true>(a.dead_cursor, 0u, inside && !active);
        if (!inside) continue;
        if (!active) j = a.n_alive + jd;
        a.out.x[j] = X;
        a.out.y[j] = Y;
        a.out.agent_food[j] = a.agent_food[n];
        a.out.slot[j] = a.slot ? a.slot[n] : (uint32_t)n;
        a.out.hhi[j] = a.hhi[n];
        a.out.hlo[j] = a.hlo[n];
        if (a.pgx) { a.opgx[j] = a.pgx[n]; a.opgy[j] = a.pgy[n]; }
    }
}

// the 'agents' channel on demand: highest alive slot per cell, as the claim pass of the classic step leaves it
__global__ __launch_bounds__(DIE_BLOCK) void k_mark_owner(die_geo g, int64_t N, int epoch, const uint32_t* x, const uint32_t* y,
                                                          const uint32_t* slot, const uint8_t* alive, unsigned long long* owner) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        if (!alive[n]) continue;
        const int64_t c = die_local(g, die_cell((int64_t)x[n], g.gW), die_cell((int64_t)y[n], g.gH));
        atomicMax(&owner[c], die_claim(epoch, slot ? (int64_t)slot[n] : n, 0.f));
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
int die_sweep_dep_plane(const die_medium* m, const die_dynamics* d, const float* dep_plane, const long long* part_gain, int n_part,
                        die_step_result* result, long long alive_const, void* stream);   // die_env.hip
bool fused_step_shape_ok(const die_medium* m, const die_dynamics* d);                     // die_env.hip: periodic planes, H % 4 == 0, radius 1..4

static bool pic_shape_ok(int xs, int ys) { return (xs == 6 && ys == 6) || (xs == 5 && ys == 7) || (xs == 5 && ys == 6) || (xs == 4 && ys == 5); }

static int pic_check(const die_medium* m, const die_pic* p, const char* who) {
    DIE_REQUIRE(m && p, "%s: null argument", who);
    DIE_REQUIRE(pic_shape_ok(p->tile_xs, p->tile_ys), "%s: tile shape 2^%d x 2^%d is not compiled in (64x64, 32x128, 32x64, 16x32)", who,
                p->tile_xs, p->tile_ys);
    const int TX = 1 << p->tile_xs, TY = 1 << p->tile_ys;
    if (m->W % TX || m->H % TY || m->W / TX < 3 || m->H / TY < 3) {
        die_set_error("%s: a %dx%d world does not split into at least 3x3 whole tiles of %dx%d cells", who, m->W, m->H, TX, TY);
        return DIE_ERR_UNSUPPORTED;
    }
    DIE_REQUIRE(p->N > 0 && p->N < ((int64_t)1 << 30), "%s: bad agent count (32-bit byte offsets into the agent arrays)", who);
    for (int l = 0; l < 2; ++l) {
        const die_pic_layout& L = p->layout[l];
        DIE_REQUIRE(L.x && L.y && L.agent_food && L.slot && L.heading_hi && L.heading_lo && L.off && L.n && L.s && L.inc, "%s: null pointer in layout %d", who, l);
    }
    DIE_REQUIRE(p->layout[0].x != p->layout[1].x && p->layout[0].off != p->layout[1].off, "%s: the two layouts must be different arrays", who);
    DIE_REQUIRE(p->dep && p->part_gain && p->error && ((p->rim && p->rim_code && p->rim_cnt) || p->dep_plane), "%s: null workspace pointer", who);
    return DIE_OK;
}

static PicLayout pic_layout(const die_pic_layout& L) {
    PicLayout o;
    o.x = L.x; o.y = L.y; o.agent_food = L.agent_food; o.slot = L.slot; o.hhi = L.heading_hi; o.hlo = L.heading_lo;
    o.off = L.off; o.n = L.n; o.s = L.s; o.inc = L.inc;
    return o;
}

extern "C" int64_t die_pic_rim_cap(int32_t tile_xs, int32_t tile_ys) {
    if (!pic_shape_ok(tile_xs, tile_ys)) return -1;
    return (1 << tile_xs) * (1 << tile_ys) >= 4096 ? KbShape<6, 6>::RIM_CAP : KbShape<4, 5>::RIM_CAP;
}

// Does a step with these parameters take the two-launch form (given rim lists)?  ONE place for the rule: the step below and the
// host side (die_amd/pic.py: whether the deposit plane of the three-launch form has to exist, whether status_out is written).
static bool pic_two_launch_rule(int worldmax, int tile_xs, int tile_ys, float scale, float diffuse_sigma, int diffuse_mode) {
    const int TX = 1 << tile_xs, TY = 1 << tile_ys;
    const float reach = fabsf(scale) * (float)(worldmax - 1);
    const int R = (int)(4.0 * (double)diffuse_sigma + 0.5);
    // an agent changes cell by at most floor(reach) + 1 per axis (+ 1 across the world's seam, where cell W − 1 and cell 0 are the
    // same point of the coordinate circle — labels linspace(0, 1, W), core/data_init.py:95-112) and must not come within R cells
    // of the FAR border of the tile it walks onto
    return diffuse_mode == DIE_DIFFUSE_WRAP && R >= 1 && R <= 4 && (int)floorf(reach) + 2 + R <= (TX < TY ? TX : TY);
}
// Per-axis bound of the vector a GradientAgent's action is `scale` times (include/die_hip.h die_pic_step_bound)
#define PIC_NOISE_MAX 2.67f          // 0.4·sqrt(−2·ln 2^-32) = 2.6642: the largest 0.4-sigma Box–Muller normal of die_forward.h / die_init_heading
extern "C" float die_pic_step_bound(float inertia, float noise_scale) {
    if (inertia == 0.f && noise_scale == 0.f) return 1.f;
    if (!(inertia >= 0.f && inertia < 1.f)) return INFINITY;
    const float ns = fabsf(noise_scale) * PIC_NOISE_MAX;
    if (inertia == 0.f) return 1.f + ns;
    const float fix = 1.f + ns / (1.f - inertia);
    return fix > PIC_NOISE_MAX ? fix : PIC_NOISE_MAX;
}
extern "C" int32_t die_pic_two_launch(int32_t world_max, int32_t tile_xs, int32_t tile_ys, float scale, float diffuse_sigma, int32_t diffuse_mode) {
    if (!pic_shape_ok(tile_xs, tile_ys) || world_max < 2) return -1;
    return pic_two_launch_rule(world_max, tile_xs, tile_ys, scale, diffuse_sigma, diffuse_mode) ? 1 : 0;
}

extern "C" int64_t die_pic_tiles(int32_t W, int32_t H, int32_t tile_xs, int32_t tile_ys) {
    if (W < 1 || H < 1 || !pic_shape_ok(tile_xs, tile_ys)) return -1;
    return (int64_t)(W >> tile_xs) * (H >> tile_ys);
}

static int pic_bin(const die_medium* m, const die_agents* a, const uint32_t* heading_hi, const uint32_t* heading_lo,
                   const float* prev_gx, const float* prev_gy, const die_pic* p, int32_t into, void* stream) {
    int rc = pic_check(m, p, "die_pic_bin");
    if (rc != DIE_OK) return rc;
    DIE_REQUIRE(a && a->N == p->N && a->x && a->y && a->agent_food && heading_hi && heading_lo, "die_pic_bin: bad agent arrays");
    DIE_REQUIRE(into == 0 || into == 1, "die_pic_bin: layout index %d", into);
    DIE_REQUIRE(a->x != p->layout[into].x, "die_pic_bin: the agents are already held in layout %d: bin into the other one", into);
    const bool dead = p->n_alive > 0 && p->n_alive < p->N;
    DIE_REQUIRE(p->n_alive >= 0 && p->n_alive <= p->N && (!dead || a->alive), "die_pic_bin: n_alive %lld of %lld slots needs the alive flags", (long long)p->n_alive, (long long)p->N);
    const int NT = (int)die_pic_tiles(m->W, m->H, p->tile_xs, p->tile_ys);
    hipStream_t s = (hipStream_t)stream;
    uint32_t* hist = (uint32_t*)p->part_gain;                 // 2·NT 64-bit words of scratch: histogram, cursors, the dead slots' cursor
    uint32_t* cursor = hist + NT;
    hipError_t e = hipMemsetAsync(hist, 0, (size_t)(2 * NT + 1) * 4, s);
    if (e != hipSuccess) { die_set_error("die_pic_bin: memset failed: %s", hipGetErrorString(e)); return DIE_ERR_HIP; }
    PicBinArgs b;
    b.g = die_geo_of(m); b.tiled = m->gW > 0; b.N = a->N; b.nty = m->H >> p->tile_ys; b.xs = p->tile_xs; b.ys = p->tile_ys;
    b.x = a->x; b.y = a->y; b.slot = a->slot; b.agent_food = a->agent_food; b.hhi = heading_hi; b.hlo = heading_lo;
    b.out = pic_layout(p->layout[into]); b.cursor = cursor;
    b.alive = dead ? a->alive : nullptr; b.n_alive = dead ? (uint32_t)p->n_alive : (uint32_t)p->N; b.dead_cursor = hist + 2 * NT;
    b.pgx = prev_gx; b.pgy = prev_gy; b.opgx = p->prev_grad[into][0]; b.opgy = p->prev_grad[into][1];
    DIE_REQUIRE((prev_gx != nullptr) == (prev_gy != nullptr) && (!prev_gx || (b.opgx && b.opgy && b.opgx != prev_gx)),
                "die_pic_bin: _prev_grad given without die_pic.prev_grad[%d] to carry it into", into);
    int64_t g = (a->N + DIE_BLOCK - 1) / DIE_BLOCK;
    const int grid = (int)(g < 4096 ? g : 4096);
    k_pic_hist<<<grid, DIE_BLOCK, 0, s>>>(b, hist);
    k_pic_bin_scan<<<1, 1024, 0, s>>>(hist, NT, pic_layout(p->layout[0]), pic_layout(p->layout[1]), cursor, p->error);
    k_pic_scatter<<<grid, DIE_BLOCK, 0, s>>>(b);
    DIE_CHECK_LAUNCH("die_pic_bin");
    return DIE_OK;
}
extern "C" int die_pic_bin(const die_medium* m, const die_agents* a, const uint32_t* heading_hi, const uint32_t* heading_lo,
                           const die_pic* p, int32_t into, void* stream) {
    return pic_bin(m, a, heading_hi, heading_lo, nullptr, nullptr, p, into, stream);
}
extern "C" int die_pic_bin_momentum(const die_medium* m, const die_agents* a, const uint32_t* heading_hi, const uint32_t* heading_lo,
                                    const float* prev_gx, const float* prev_gy, const die_pic* p, int32_t into, void* stream) {
    return pic_bin(m, a, heading_hi, heading_lo, prev_gx, prev_gy, p, into, stream);
}

template <int XS, int YS>
static void launch_resolve(const PicArgs& k, float* dep_plane, int NT, bool f32, bool feed, hipStream_t s) {
    if (f32) {
        if (feed) k_pic_resolve<float, XS, YS, true><<<dim3(k.nty, k.ntx + 1), PIC_K2_BLOCK, 0, s>>>(k, dep_plane);
        else k_pic_resolve<float, XS, YS, false><<<dim3(k.nty, k.ntx + 1), PIC_K2_BLOCK, 0, s>>>(k, dep_plane);
    } else {
        if (feed) k_pic_resolve<__half, XS, YS, true><<<dim3(k.nty, k.ntx + 1), PIC_K2_BLOCK, 0, s>>>(k, dep_plane);
        else k_pic_resolve<__half, XS, YS, false><<<dim3(k.nty, k.ntx + 1), PIC_K2_BLOCK, 0, s>>>(k, dep_plane);
    }
}

template <typename T, bool STAGE, bool RIM, bool TILED = false>
static void launch_forward_move(int kind, const FwdArgs& f, const PicArgs& k, int NT, int block, size_t lds, hipStream_t s, bool mom = false) {
    // (an order table covers exactly the tiles: 8 · order_len = ntx · nty workgroups, the same grid)
    const dim3 grid(k.sub_mode == 1 ? k.sub_nty : k.nty, k.sub_mode == 1 ? k.sub_ntx : k.ntx);
    if constexpr (!TILED) {
        if (kind != DIE_AGENT_PHYSARUM && mom) {
            k_pic_forward_move<T, DIE_AGENT_GRADIENT, STAGE, true, RIM, false, true><<<grid, block, lds, s>>>(f, k);
            return;
        }
    }
    if (kind != DIE_AGENT_PHYSARUM) k_pic_forward_move<T, DIE_AGENT_GRADIENT, STAGE, true, RIM, TILED><<<grid, block, lds, s>>>(f, k);
    else if (k.adx) k_pic_forward_move<T, DIE_AGENT_PHYSARUM, STAGE, true, RIM, TILED><<<grid, block, lds, s>>>(f, k);
    else k_pic_forward_move<T, DIE_AGENT_PHYSARUM, STAGE, false, RIM, TILED><<<grid, block, lds, s>>>(f, k);
}

int die_gaussian_taps(float sigma, double* w);             // die_env.hip (scipy.ndimage._gaussian_kernel1d); w holds 2·8 + 1 taps

template <typename T, int XS, int YS, bool TILED>
static void launch_resolve_diffuse(const PicArgs& k, const KbArgs& a, int R, hipStream_t s) {
    constexpr int TX = 1 << XS, TY = 1 << YS, A = 16 / (int)sizeof(T), CP = TY + 2 * A;
    const int WR = TX + 2 * R, WC = TY + 2 * R;
    const size_t lds = ((size_t)WR * CP + (size_t)(WR * WC > TX * CP ? WR * WC : TX * CP)) * 4;
    // (a rectangle of tiles only: no extra grid row — the scan / reduction / turn bits belong to the launch that completes the step)
    const dim3 grid(k.sub_mode == 1 ? k.sub_nty : k.nty, k.sub_mode == 1 ? k.sub_ntx : k.ntx + 1);
    constexpr int B = KbShape<XS, YS>::BLOCK;
    switch (R) {
        case 1: k_pic_resolve_diffuse<T, XS, YS, 1, TILED><<<grid, B, lds, s>>>(k, a); break;
        case 2: k_pic_resolve_diffuse<T, XS, YS, 2, TILED><<<grid, B, lds, s>>>(k, a); break;
        case 3: k_pic_resolve_diffuse<T, XS, YS, 3, TILED><<<grid, B, lds, s>>>(k, a); break;
        default: k_pic_resolve_diffuse<T, XS, YS, 4, TILED><<<grid, B, lds, s>>>(k, a); break;
    }
}

template <typename T, bool TILED>
static void launch_resolve_diffuse_shape(int xs, int ys, const PicArgs& k, const KbArgs& a, int R, hipStream_t s) {
    if (xs == 6 && ys == 6) launch_resolve_diffuse<T, 6, 6, TILED>(k, a, R, s);
    else if (xs == 5 && ys == 7) launch_resolve_diffuse<T, 5, 7, TILED>(k, a, R, s);
    else if (xs == 5 && ys == 6) launch_resolve_diffuse<T, 5, 6, TILED>(k, a, R, s);
    else launch_resolve_diffuse<T, 4, 5, TILED>(k, a, R, s);
}

extern "C" int die_pic_forward_env_step(const die_medium* m, const die_pic* p, int32_t from, die_gradient_agent* g,
                                        const die_action* act, const die_dynamics* d, die_step_result* result, void* stream) {
    int rc = pic_check(m, p, "die_pic_forward_env_step");
    if (rc != DIE_OK) return rc;
    DIE_REQUIRE(g && d && result && (from == 0 || from == 1), "die_pic_forward_env_step: null argument");
    DIE_REQUIRE(m->chem_next && m->chem_next != m->chem, "die_pic_forward_env_step: chem_next must be a second plane");
    const bool dead = p->n_alive > 0 && p->n_alive < p->N;          // dead slots behind the segments (die_pic.n_alive)
    DIE_REQUIRE((!d->has_dead_slots || dead) && !d->agents_die && !m->sense_mask && !d->staged,
                "die_pic_forward_env_step: every slot alive, or the dead slots behind the segments (die_pic.n_alive); no agents_die, no sense mask");
    DIE_REQUIRE(!dead || (p->occ && m->gW <= 0), "die_pic_forward_env_step: dead slots need the occupancy bitmap (die_pic.occ) and a single-tile world");
    // momentum (GradientAgent's inertia / noise, gradient.py:82-91): _prev_grad rides in die_pic.prev_grad, the step is bounded by
    // |scale|·die_pic_step_bound.  (g->prev_gx / prev_gy are ignored here: the state is the layouts'.)
    const bool mom = g->inertia != 0.f || g->noise_scale != 0.f;
    const float ubound = die_pic_step_bound(g->inertia, g->noise_scale);
    DIE_REQUIRE(g->normalized_grad && !g->step_base && (!mom || (g->kind == DIE_AGENT_GRADIENT && ubound < 1e6f)),
                "die_pic_forward_env_step: the step length must be bounded (normalised gradient; momentum for a GradientAgent with inertia < 1 only), no graph replay");
    DIE_REQUIRE(!mom || (!dead && m->gW <= 0 && !p->sub_mode), "die_pic_forward_env_step: momentum combines with neither dead slots nor a decomposed world's tile");
    const float* const* pg_in = (const float* const*)p->prev_grad[from];
    float* const* pg_out = (float* const*)p->prev_grad[1 - from];
    DIE_REQUIRE(g->inertia == 0.f || (pg_in[0] && pg_in[1] && pg_out[0] && pg_out[1] && pg_in[0] != pg_out[0]),
                "die_pic_forward_env_step: inertia needs die_pic.prev_grad of both layouts");
    if (d->boundary != DIE_BOUNDARY_WRAP && d->boundary != DIE_BOUNDARY_LIMIT) {
        die_set_error("die_pic_forward_env_step: boundary %d is not representable in Q0.32", d->boundary);
        return DIE_ERR_UNSUPPORTED;
    }
    DIE_REQUIRE(d->cost == DIE_COST_LINEAR || d->cost == DIE_COST_ZERO, "die_pic_forward_env_step: bad cost operator %d", d->cost);
    const int TX = 1 << p->tile_xs, TY = 1 << p->tile_ys;
    const bool tiled = m->gW > 0;                           // a decomposed world's tile: lengths are fractions of the WORLD
    const int worldmax = tiled ? (m->gW > m->gH ? m->gW : m->gH) : (m->W > m->H ? m->W : m->H);
    const float step_scale = g->scale * ubound;
    const float reach = fabsf(step_scale) * (float)(worldmax - 1);     // cells per step, at most
    if (!(reach <= (float)((TX < TY ? TX : TY) - 1))) {
        die_set_error("die_pic_forward_env_step: a step of %.1f cells does not stay within the neighbouring %dx%d tiles", (double)reach, TX, TY);
        return DIE_ERR_UNSUPPORTED;
    }
    const die_pic_layout& Lin = p->layout[from];
    die_agents a;
    a.N = p->N; a.x = Lin.x; a.y = Lin.y; a.alive = nullptr; a.agent_food = Lin.agent_food; a.slot = Lin.slot;
    die_gradient_agent gg = *g;
    gg.heading_hi = Lin.heading_hi; gg.heading_lo = Lin.heading_lo;
    gg.prev_gx = g->inertia != 0.f ? (float*)pg_in[0] : nullptr; gg.prev_gy = g->inertia != 0.f ? (float*)pg_in[1] : nullptr;
    FwdArgs f;
    rc = die_fill_fwd_args(f, m, &a, &gg, act, "die_pic_forward_env_step");
    if (rc != DIE_OK) return rc;
    PicArgs k;
    k.g = die_geo_of(m); k.ntx = m->W >> p->tile_xs; k.nty = m->H >> p->tile_ys; k.xs = p->tile_xs; k.ys = p->tile_ys;
    k.in = pic_layout(Lin); k.out = pic_layout(p->layout[1 - from]);
    k.dep = p->dep;
    k.adx = act ? act->dx : nullptr; k.ady = act ? act->dy : nullptr; k.adep = act ? act->deposit : nullptr;
    k.food = m->food; k.rate_feed = d->rate_feed; k.w_dep = d->cost_w_deposit; k.w_dist = d->cost_w_dist;
    k.boundary = d->boundary; k.cost = d->cost;
    k.part_gain = (long long*)p->part_gain; k.error = p->error;
    k.sub_mode = p->sub_mode; k.sub_tx0 = p->sub_tx0; k.sub_ty0 = p->sub_ty0; k.sub_ntx = p->sub_ntx; k.sub_nty = p->sub_nty;
    k.halo_fresh = m->gW > 0 ? p->halo_fresh : 0;
    const bool keep_pg = mom && g->inertia != 0.f;
    k.ipgx = keep_pg ? pg_in[0] : nullptr; k.ipgy = keep_pg ? pg_in[1] : nullptr; k.opgx = keep_pg ? pg_out[0] : nullptr; k.opgy = keep_pg ? pg_out[1] : nullptr;
    k.order = nullptr; k.order_len = 0;
    DIE_REQUIRE(p->sub_mode >= 0 && p->sub_mode <= 2, "die_pic_forward_env_step: sub_mode %d", p->sub_mode);
    if (p->sub_mode) {
        DIE_REQUIRE(p->stages == 1 || p->stages == 2, "die_pic_forward_env_step: a subset of the tiles is one launch (stages 1 or 2), not a whole step");
        DIE_REQUIRE(p->sub_tx0 >= 0 && p->sub_ty0 >= 0 && p->sub_ntx >= 0 && p->sub_nty >= 0 && p->sub_tx0 + p->sub_ntx <= k.ntx && p->sub_ty0 + p->sub_nty <= k.nty,
                    "die_pic_forward_env_step: the rectangle of tiles lies outside the planes");
        if (p->sub_mode == 1 && (p->sub_ntx == 0 || p->sub_nty == 0)) return DIE_OK;      // an empty rectangle: nothing to launch
    }
    const int NT = k.ntx * k.nty;
    hipStream_t s = (hipStream_t)stream;
    // K1 stages chem of the tile ± the probe reach in LDS when that fits: an agent's probe cell lies at most
    // floor(|sense_offset|·(size − 1)) + 1 cells from its own cell, the gradient taps one further
    const int esz = m->dtype == DIE_F32 ? 4 : 2, V = 16 / esz;
    int P = (int)floorf(fabsf(g->sense_offset) * (float)(worldmax - 1)) + 2;
    P = (P + V - 1) / V * V;
    const bool stage = P <= PIC_MAX_MARGIN;
    k.margin = stage ? P : 0;
    // the food block: the tile ± the cells an agent can walk onto in one step (floor(reach) + 1 by the move, one more across the
    // world's seam), columns in whole vectors
    k.fm_r = stage ? (int)floorf(reach) + 2 : 0;
    // (columns: PIC_FOOD_COLS = 1 only)
    k.fm_c = PIC_FOOD_COLS ? (k.fm_r + V - 1) / V * V : 0;
    const int vpr_c = (TY + 2 * k.margin) / V, vpr_f = (TY + 2 * k.fm_c) / V;     // 16-byte vectors per staged row
    k.mg_c = ((1u << 20) + (uint32_t)vpr_c - 1u) / (uint32_t)vpr_c;
    k.mg_f = ((1u << 20) + (uint32_t)vpr_f - 1u) / (uint32_t)vpr_f;
    {   // k / wb of pic_xcd_tile as one multiply (checked for every k the mapping can see)
        const uint32_t wb = (uint32_t)k.nty >> 3;
        k.xcd_wb_mul = wb > 1 ? (uint32_t)((((uint64_t)1 << 32) + wb - 1) / wb) : 0u;
        bool ok = true;
        for (uint32_t q = 0; ok && k.xcd_wb_mul && q < wb * (uint32_t)k.ntx; ++q) ok = (uint32_t)(((uint64_t)q * k.xcd_wb_mul) >> 32) == q / wb;
        if (!ok) k.xcd_wb_mul = 0;
    }
    if (stage) {                                            // (the multiply-shift division of PicStageRows, checked for every thread)
        bool ok = true;
        for (int i = 0; ok && i < PIC_K1_BLOCK; ++i) ok = (int)(((uint32_t)i * k.mg_c) >> 20) == i / vpr_c && (int)(((uint32_t)i * k.mg_f) >> 20) == i / vpr_f;
        DIE_REQUIRE(ok && vpr_c <= 64 && vpr_f <= 64, "die_pic_forward_env_step: staged rows of %d / %d vectors", vpr_c, vpr_f);
    }
    DIE_REQUIRE(!stage || (m->W < (1 << 23) && m->H < (1 << 23) && (int64_t)m->W * m->H < (1ll << 31)),
                "die_pic_forward_env_step: plane too large for the 24-bit index arithmetic of the staging loop");
    DIE_REQUIRE(!stage || m->H % V == 0, "die_pic_forward_env_step: plane rows must be whole 16-byte vectors");
    const size_t lds = stage ? ((size_t)(TX + 2 * P) * (TY + 2 * P) + (size_t)(TX + 2 * k.fm_r) * (TY + 2 * k.fm_c)) * esz : 0;
    const int stages = p->stages ? p->stages : 7;          // bit 0: agent kernel, bit 1: resolve + scan, bit 2: field sweep
    // two launches (one field kernel per tile, fed by the agent kernel's rim lists) when the caller gave the lists and every
    // agent that matters to a tile sits in one of the 9 segments around it: an agent changes cell by at most floor(reach) + 1 per axis and must not come within R cells of the
    // FAR border of the tile it walks onto
    const int R = (int)(4.0 * (double)d->diffuse_sigma + 0.5);
    const bool two = p->rim != nullptr && p->rim_code != nullptr && p->rim_cnt != nullptr &&
                     pic_two_launch_rule(worldmax, p->tile_xs, p->tile_ys, step_scale, d->diffuse_sigma, d->diffuse_mode);
    if (!two) DIE_REQUIRE(p->dep_plane, "die_pic_forward_env_step: this step needs the three-launch form: dep_plane is null");
    DIE_REQUIRE(two || !p->sub_mode, "die_pic_forward_env_step: subsets of the tiles exist in the two-launch form only");
    DIE_REQUIRE(!(dead && p->sub_mode), "die_pic_forward_env_step: subsets of the tiles and dead slots do not combine");
    if (tiled && !(two && stage)) {
        die_set_error("die_pic_forward_env_step: a decomposed world's tile runs the two-launch form with staged tiles only (probe reach %d, radius %d)", P, R);
        return DIE_ERR_UNSUPPORTED;
    }
    if (tiled && ((m->gW != m->W && m->gW - m->W < 2 * k.margin + 2) || (m->gH != m->H && m->gH - m->H < 2 * k.margin + 2))) {
        die_set_error("die_pic_forward_env_step: the planes (%d x %d) nearly span the world (%d x %d): the taps around an agent are mapped "
                      "from its own cell (FwdTileMem::home), which needs %d cells of world beyond the planes", m->W, m->H, m->gW, m->gH, 2 * k.margin + 2);
        return DIE_ERR_UNSUPPORTED;
    }
    k.rim = (uint4*)p->rim; k.rim_code = p->rim_code; k.rim_cnt = p->rim_cnt; k.rim_cap = (int)die_pic_rim_cap(p->tile_xs, p->tile_ys); k.rim_r = R;
    const bool feed_in_k2 = PIC_K2_FEED && !d->food_infinite;
    int block = p->k1_threads > 0 ? p->k1_threads : (TX * TY >= 4096 ? 512 : 256);
    DIE_REQUIRE(block % DIE_WAVE == 0 && block >= DIE_WAVE && block <= PIC_K1_BLOCK, "die_pic_forward_env_step: k1_threads %d", block);
    k.rp_c = stage ? block / vpr_c : 1; k.rp_f = stage ? block / vpr_f : 1;
    // the random turn bits of this step (PhysarumAgent): a table over the slot ids, filled by the previous step's field kernel
    // or — the first step, a changed seed, a step counter that did not advance by one — right here
    const bool physarum = g->kind == DIE_AGENT_PHYSARUM;
    const int64_t turn_words = (p->turn_slots + 127) / 128 * 4;
    if (physarum) {
        DIE_REQUIRE(p->turn_bits && p->turn_slots > 0, "die_pic_forward_env_step: turn_bits / turn_slots (a table over every slot id) missing");
        f.turn_bits = p->turn_bits;
        if ((stages & 1) && (!p->turn_ready || !two) && !g->turn_sign) {      // (only the two-launch form's field kernel fills it ahead)
            const int64_t blocks = (turn_words / 4 + DIE_BLOCK - 1) / DIE_BLOCK;
            k_turn_bits<<<(int)(blocks < 1024 ? blocks : 1024), DIE_BLOCK, 0, s>>>(p->turn_bits, turn_words, g->seed, g->step);
        }
    }
    // the order table (die_pic.order): both kernels of the two-launch form take their tiles from it; rebuilt from the populations of the
    // layout this step reads before its agent kernel — the first time, and every PIC_ORDER_PERIOD-th step
    if (p->order && two && !tiled && !p->sub_mode && (k.nty & 7) == 0 && NT <= 65536) {
        k.order = p->order; k.order_len = NT >> 3;
        if ((stages & 1) && (!p->order_ready || (g->step % PIC_ORDER_PERIOD) == 0))
            k_pic_order<<<dim3(8, ((NT >> 3) + PIC_ORDER_SPAN - 1) / PIC_ORDER_SPAN), PIC_ORDER_BLOCK, 0, s>>>(k.in.n, k.ntx, k.nty, p->order);
    }
    if (stages & 1) {
#define DIE_PIC_K1(T, STAGE, LDS) do { if (two) launch_forward_move<T, STAGE, true>(g->kind, f, k, NT, block, LDS, s, mom); \
                                       else launch_forward_move<T, STAGE, false>(g->kind, f, k, NT, block, LDS, s, mom); } while (0)
        if (tiled) {
            if (m->dtype == DIE_F32) launch_forward_move<float, true, true, true>(g->kind, f, k, NT, block, lds, s);
            else launch_forward_move<__half, true, true, true>(g->kind, f, k, NT, block, lds, s);
        } else if (m->dtype == DIE_F32) {
            if (stage) DIE_PIC_K1(float, true, lds);
            else DIE_PIC_K1(float, false, 0);
        } else {
            if (stage) DIE_PIC_K1(__half, true, lds);
            else DIE_PIC_K1(__half, false, 0);
        }
#undef DIE_PIC_K1
    }
    // dead slots (two-launch form): the cells the alive agents stand on now, then the slots that never lived
#ifndef PIC_DEAD_BLOCKS_CAP
#define PIC_DEAD_BLOCKS_CAP 0          // workgroups of k_pic_dead: 0 = one slot per thread, else at most this many (grid-stride)
#endif
    const int64_t dead_want = dead ? (p->N - p->n_alive + DIE_BLOCK - 1) / DIE_BLOCK : 0;
    const int64_t dead_launch = PIC_DEAD_BLOCKS_CAP > 0 && dead_want > PIC_DEAD_BLOCKS_CAP ? PIC_DEAD_BLOCKS_CAP : dead_want;
    const int dead_blocks = (int)(dead_launch < NT ? dead_launch : NT);       // reward partials the field kernel sums behind the tiles'
    if (dead) {
        DIE_REQUIRE(two, "die_pic_forward_env_step: dead slots exist in the two-launch form only");
        if (stages & 1) {
            hipError_t e = hipMemsetAsync(p->occ, 0, (size_t)m->W * m->H, s);
            if (e != hipSuccess) { die_set_error("die_pic_forward_env_step: memset failed: %s", hipGetErrorString(e)); return DIE_ERR_HIP; }
            const int64_t gm = (p->n_alive + DIE_BLOCK - 1) / DIE_BLOCK;
            k_pic_mark<<<(int)gm, DIE_BLOCK, 0, s>>>(k, (uint32_t)p->n_alive, (uint8_t*)p->occ);
            long long* part = (long long*)p->part_gain + NT;
            if (dead_launch > NT) {
                e = hipMemsetAsync(part, 0, (size_t)NT * 8, s);
                if (e != hipSuccess) { die_set_error("die_pic_forward_env_step: memset failed: %s", hipGetErrorString(e)); return DIE_ERR_HIP; }
            }
            const uint32_t first = (uint32_t)p->n_alive, count = (uint32_t)(p->N - p->n_alive);
            const int dg = (int)dead_launch;
            if (m->dtype == DIE_F32) {
                if (physarum) k_pic_dead<float, DIE_AGENT_PHYSARUM><<<dg, DIE_BLOCK, 0, s>>>(f, k, first, count, (const uint8_t*)p->occ, part, (uint32_t)NT);
                else k_pic_dead<float, DIE_AGENT_GRADIENT><<<dg, DIE_BLOCK, 0, s>>>(f, k, first, count, (const uint8_t*)p->occ, part, (uint32_t)NT);
            } else {
                if (physarum) k_pic_dead<__half, DIE_AGENT_PHYSARUM><<<dg, DIE_BLOCK, 0, s>>>(f, k, first, count, (const uint8_t*)p->occ, part, (uint32_t)NT);
                else k_pic_dead<__half, DIE_AGENT_GRADIENT><<<dg, DIE_BLOCK, 0, s>>>(f, k, first, count, (const uint8_t*)p->occ, part, (uint32_t)NT);
            }
        }
    }
    if (two) {
        if (stages & 2) {                                   // (bit 2 alone: nothing — the sweep is part of this kernel)
            if (!tiled && !fused_step_shape_ok(m, d)) {
                die_set_error("die_pic_forward_env_step: only for periodic planes with H %% 4 == 0 and gaussian radius 1..4");
                return DIE_ERR_UNSUPPORTED;
            }
            KbArgs a;
            double wd[2 * 8 + 1];
            die_gaussian_taps(d->diffuse_sigma, wd);
            a.chem = m->chem; a.chem_next = m->chem_next; a.rim = (const uint4*)p->rim; a.rim_code = p->rim_code; a.rim_cnt = p->rim_cnt;
            a.food_infinite = d->food_infinite; a.keep = (float)(1.0 - (double)d->rate_decay_chem);
            for (int q = 0; q <= 2 * R; ++q) a.w[q] = (float)wd[q];
            a.result = result; a.alive_const = dead ? p->n_alive : p->N; a.status_out = (long long*)p->status_out; a.error = p->error;
            a.n_part = NT + (dead ? dead_blocks : 0);
            // cache policy of the field kernel's two plane stores.  Measured (profiles/r04_nt_stores_by_size.txt): with the planes and
            // agent arrays beyond the 256 MiB Infinity Cache (4096² fp32: 342 MB, 8192²) non-temporal stores make the step 3.5 % / 2 %
            // faster — the lines would be evicted before the next kernel reads them anyway and only displace the windows' shared
            // lines from L2 —, below it (4096² fp16: 241 MB, 2048²) they cost 1–3 %: the next kernel finds them in the cache
            {
                const double state = 3.0 * (double)m->W * m->H * esz + 2.0 * 28.0 * (double)p->N;
                a.nt_out = PIC_NT_OUT < 0 ? (state > 256.0 * 1024 * 1024 ? 1 : 0) : PIC_NT_OUT;
            }
            a.turn_bits = physarum && !g->turn_sign && k.nty > 2 ? p->turn_bits : nullptr;
            a.turn_words = turn_words; a.turn_seed = g->seed; a.turn_step = g->step + 1u;
            if (tiled) {
                if (m->dtype == DIE_F32) launch_resolve_diffuse_shape<float, true>(p->tile_xs, p->tile_ys, k, a, R, s);
                else launch_resolve_diffuse_shape<__half, true>(p->tile_xs, p->tile_ys, k, a, R, s);
            } else if (m->dtype == DIE_F32) launch_resolve_diffuse_shape<float, false>(p->tile_xs, p->tile_ys, k, a, R, s);
            else launch_resolve_diffuse_shape<__half, false>(p->tile_xs, p->tile_ys, k, a, R, s);
        }
        DIE_CHECK_LAUNCH("die_pic_forward_env_step");
        return DIE_OK;
    }
    if (stages & 2) {
        const bool f32 = m->dtype == DIE_F32;
        if (p->tile_xs == 6 && p->tile_ys == 6) launch_resolve<6, 6>(k, p->dep_plane, NT, f32, feed_in_k2, s);
        else if (p->tile_xs == 5 && p->tile_ys == 7) launch_resolve<5, 7>(k, p->dep_plane, NT, f32, feed_in_k2, s);
        else if (p->tile_xs == 5 && p->tile_ys == 6) launch_resolve<5, 6>(k, p->dep_plane, NT, f32, feed_in_k2, s);
        else launch_resolve<4, 5>(k, p->dep_plane, NT, f32, feed_in_k2, s);
    }
    DIE_CHECK_LAUNCH("die_pic_forward_env_step");
    if (!(stages & 4)) return DIE_OK;
    die_dynamics dsweep = *d;
    if (feed_in_k2) dsweep.food_infinite = 1;               // K2 has fed the occupied cells: the sweep leaves the food alone
    return die_sweep_dep_plane(m, &dsweep, p->dep_plane, (const long long*)p->part_gain, NT, result, p->N, stream);
}

// n_steps × (forward + step) without the host in between (include/die_hip.h die_pic_run): the loop of examples/minimal_run.py:23-25 for a
// caller that reads nothing back on the way.  Every step is die_pic_forward_env_step with the roles of the two layouts and of the
// two chem planes exchanged and the Philox step counter advanced — the same launches, the same bits.
static thread_local int32_t g_pic_run_done = 0;        // whole steps the last die_pic_run of this thread had enqueued when it returned
extern "C" int32_t die_pic_run_completed(void) { return g_pic_run_done; }
extern "C" int die_pic_run(const die_medium* m, const die_pic* p, int32_t from, const die_gradient_agent* g, const die_dynamics* d,
                           int32_t n_steps, die_step_result* results, void* stream) {
    g_pic_run_done = 0;
    DIE_REQUIRE(m && p && g && d && results && n_steps >= 0 && (from == 0 || from == 1), "die_pic_run: null argument");
    DIE_REQUIRE(p->stages == 0 && p->sub_mode == 0, "die_pic_run: whole steps only (stages = 0, all tiles)");
    die_medium mm = *m;
    die_pic pp = *p;
    die_gradient_agent gg = *g;
    for (int32_t i = 0; i < n_steps; ++i) {
        const int rc = die_pic_forward_env_step(&mm, &pp, (from + i) & 1, &gg, nullptr, d, results + i, stream);
        if (rc != DIE_OK) return rc;
        g_pic_run_done = i + 1;
        void* t = mm.chem; mm.chem = mm.chem_next; mm.chem_next = t;                    // Env.step: swap_chem
        gg.step += 1u;                                                                   // the agent object's call counter
        // (the two-launch form's field kernel has left the next step's turn bits in the table; the three-launch form fills it itself)
        pp.turn_ready = 1;
        pp.order_ready = 1;                                                              // (the first step has built the order table)
    }
    return DIE_OK;
}

// The action of the step that WROTE layout `lay`, re-derived from what that step left behind: a normalised PhysarumAgent
// without momentum moves by scale·polar2xy(1, heading') (die_forward.h: ux += 0 after polar2xy, dx = ux·scale) and its
// deposit went to p->dep — the same arithmetic on the same inputs, so the same bits as the action K1 would have stored.
__global__ __launch_bounds__(DIE_BLOCK) void k_pic_action_physarum(int64_t N, const uint32_t* hhi, const uint32_t* hlo, const float* dep,
                                                                  float scale, float* dx, float* dy, float* adep) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        float ux, uy;
        die_polar2xy_heading(__hiloint2double((int)hhi[n], (int)hlo[n]), 1.f, &ux, &uy);
        ux += 0.f;
        uy += 0.f;
        dx[n] = ux * scale;
        dy[n] = uy * scale;
        adep[n] = dep[n];
    }
}

extern "C" int die_pic_action_physarum(const die_pic* p, int32_t lay, const die_gradient_agent* g, const die_action* act, void* stream) {
    DIE_REQUIRE(p && g && act && (lay == 0 || lay == 1), "die_pic_action_physarum: null argument");
    DIE_REQUIRE(g->kind == DIE_AGENT_PHYSARUM && g->normalized_grad && g->inertia == 0.f && g->noise_scale == 0.f,
                "die_pic_action_physarum: a normalised PhysarumAgent without inertia or noise");
    const die_pic_layout& L = p->layout[lay];
    DIE_REQUIRE(p->N > 0 && act->N == p->N, "die_pic_action_physarum: the action holds %lld entries, the layout %lld", (long long)act->N, (long long)p->N);
    DIE_REQUIRE(L.heading_hi && L.heading_lo && p->dep && act->dx && act->dy && act->deposit, "die_pic_action_physarum: null array");
    int64_t grid = (p->N + DIE_BLOCK - 1) / DIE_BLOCK;
    k_pic_action_physarum<<<(int)(grid < 8192 ? grid : 8192), DIE_BLOCK, 0, (hipStream_t)stream>>>(
        p->N, L.heading_hi, L.heading_lo, p->dep, g->scale, act->dx, act->dy, act->deposit);
    DIE_CHECK_LAUNCH("die_pic_action_physarum");
    return DIE_OK;
}

extern "C" int die_agents_mark_owner(const die_medium* m, const die_agents* a, void* stream) {
    DIE_REQUIRE(m && a && m->owner && a->x && a->y && a->alive && a->N > 0, "die_agents_mark_owner: null argument");
    DIE_REQUIRE(m->epoch >= 1 && m->epoch <= DIE_OWNER_EPOCH_MAX, "die_agents_mark_owner: bad epoch %d", m->epoch);
    int64_t g = (a->N + DIE_BLOCK - 1) / DIE_BLOCK;
    k_mark_owner<<<(int)(g < 8192 ? g : 8192), DIE_BLOCK, 0, (hipStream_t)stream>>>(die_geo_of(m), a->N, m->epoch, a->x, a->y, a->slot,
                                                                                    a->alive, (unsigned long long*)m->owner);
    DIE_CHECK_LAUNCH("die_agents_mark_owner");
    return DIE_OK;
}
