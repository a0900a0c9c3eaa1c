// Shared device/host helpers for libdie_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/die_hip.h"

#define DIE_WAVE 64
#ifndef DIE_BLOCK
#define DIE_BLOCK 256
#endif

void die_set_error(const char* fmt, ...);

#define DIE_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) {                              \
            die_set_error(__VA_ARGS__);             \
            return DIE_ERR_ARG;                     \
        }                                           \
    } while (0)

// The three words of a step's result (reward, num_alive, status) may live in pinned HOST memory that the caller polls while the
// kernel that writes them is still running (Env(sync=True): include/die_hip.h die_host_device_pointer).  Written as system-scope
// atomic stores (write-through, `sc0 sc1`), so that their visibility to the host does not rest on the allocation being
// fine-grained / uncached or on the kernel's end; the host waits for each word's sentinel on its own, so no order between them is
// needed and no fence is issued (a release fence here would write back the XCD's whole L2 in the middle of the field kernel).
__device__ __forceinline__ void die_store_result_f64(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void die_store_result_i64(long long* p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

#define DIE_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            die_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return DIE_ERR_HIP;                                                       \
        }                                                                             \
    } while (0)

static inline int die_grid_for(int64_t n, int block = DIE_BLOCK) { return (int)((n + block - 1) / block); }

// ---- field element access (f32 / f16 planes) ---------------------------------------------
__device__ __forceinline__ float die_ld(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float die_ld(const __half* p, int64_t i) { return __half2float(p[i]); }
__device__ __forceinline__ void die_st(float* p, int64_t i, float v) { p[i] = v; }
// fp32 → fp16 of a value that has been ROUNDED TO fp32 first.  Without the barrier the compiler folds the arithmetic that
// produced `v` and the conversion into one mixed-precision instruction (v_fma_mixlo_f16: a single rounding straight to
// fp16) in some instantiations and not in others — the results then differ in the last fp16 bit whenever the fp32 value is
// an exact tie between two halves (found by tests/fuzz_cases.py fuzz_binned: f − 0.35·f on fp16 food).
__device__ __forceinline__ float die_round_f32(float v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ __half die_f2h(float v) { return __float2half(die_round_f32(v)); }
__device__ __forceinline__ void die_st(__half* p, int64_t i, float v) { p[i] = die_f2h(v); }
// what a value becomes when it is stored in a plane of T and read back (the staged step stores chem + deposit before it
// diffuses; the fused sweeps keep the sum in registers and have to round it the same way to give the same bits)
template <typename T> __device__ __forceinline__ float die_as_stored(float v);
template <> __device__ __forceinline__ float die_as_stored<float>(float v) { return v; }
template <> __device__ __forceinline__ float die_as_stored<__half>(float v) { return __half2float(die_f2h(v)); }

// ---- Q0.32 fixed-point coordinates -------------------------------------------------------
// nearest label of linspace(0, 1, n) to X / 2^32, P may lie outside [0, 2^32) (probe offsets):
// clamp like pandas get_indexer(method='nearest') does (core/utils.py:39-54).
__device__ __forceinline__ int die_cell_u(uint32_t X, int n) {       // X in [0, 2^32): one 32×32→64 multiply-add, never clamps
    return (int)(((uint64_t)X * (uint32_t)(n - 1) + 0x80000000ull) >> 32);
}
__device__ __forceinline__ int die_cell(int64_t P, int n) {
    // (P·(n−1) + 2^31) >> 32 clamped to [0, n−1]: below 0 the quotient is ≤ 0, from 2^32 on it is ≥ n−1, so only the low
    // word needs the multiply (a zero-extended coordinate folds to die_cell_u at compile time)
    const int hi = (int)(P >> 32);
    const int c = die_cell_u((uint32_t)P, n);
    return hi == 0 ? c : (hi < 0 ? 0 : n - 1);
}

// nearest label to (X + off) / 2^32 for a coordinate X and a signed 32-bit offset (a probe): the 33-bit sum never
// materialises — a carry out of the 32-bit add (off >= 0) is "beyond the last label", a borrow (off < 0) "before the first"
__device__ __forceinline__ int die_cell_off(uint32_t X, int32_t off, int n) {
    const uint32_t lo = X + (uint32_t)off;
    const bool out = off >= 0 ? lo < X : lo > X;
    const int c = die_cell_u(lo, n);
    return out ? (off >= 0 ? n - 1 : 0) : c;
}

// square root to 1 ulp (v_sqrt_f32) instead of the correctly rounded library form (15 instructions).  Arguments below
// 2^-126 (denormals) count as 0: lengths that small are below the resolution of everything they are added to.
__device__ __forceinline__ float die_sqrt1(float v) { return __builtin_amdgcn_sqrtf(v); }

// float displacement (fraction of the unit square) → Q0.32 increment
// v·2^32 is exact in fp32 (a power-of-two scaling), so below 2^31 the nearest integer comes from the 32-bit convert;
// larger displacements take the float64 route.  Same value either way.
// the same for |v| < 0.5 (a guarantee of the caller: the tile-binned step bounds both the probe offset and the step length
// by a tile): no float64 fallback, no branch
__device__ __forceinline__ int64_t die_q32_small(float v) { return (int64_t)__float2int_rn(v * 4294967296.0f); }
__device__ __forceinline__ int64_t die_q32(float v) {
    const float t = v * 4294967296.0f;
    if (fabsf(t) < 2147483520.0f) return (int64_t)__float2int_rn(t);
    return (int64_t)__double2ll_rn((double)v * 4294967296.0);
}

__device__ __forceinline__ uint32_t die_owner_word(int epoch, int64_t slot) {
    return ((uint32_t)epoch << DIE_OWNER_EPOCH_SHIFT) | (uint32_t)(slot + 1);
}
__device__ __forceinline__ bool die_owner_occupied(uint32_t w, int epoch) {
    return (w >> DIE_OWNER_EPOCH_SHIFT) == (uint32_t)epoch;
}
// 64-bit claim: ownership word above the fp32 bits of the claimant's deposit
__device__ __forceinline__ unsigned long long die_claim(int epoch, int64_t slot, float deposit) {
    return ((unsigned long long)die_owner_word(epoch, slot) << 32) | (unsigned long long)__float_as_uint(deposit);
}
__device__ __forceinline__ bool die_claim_occupied(unsigned long long k, int epoch) {
    return (uint32_t)(k >> (32 + DIE_OWNER_EPOCH_SHIFT)) == (uint32_t)epoch;
}
__device__ __forceinline__ float die_claim_deposit(unsigned long long k) { return __uint_as_float((uint32_t)k); }

// The reward is accumulated in 32.32 fixed point: integer sums are associative, so reward does not depend on the order
// of the agent arrays, on the grid size or on the decomposition — bit for bit (a slot's gain is rounded to 2^-32 ≈
// 2.3e-10 once; the float64 sum it replaces differed by rounding noise between orders).
#define DIE_FIX_ONE 4294967296.0
__device__ __forceinline__ long long die_fix(float g) {
    const float t = g * 4294967296.0f;                       // exact (power-of-two scaling)
    if (fabsf(t) < 2147483520.0f) return (long long)__float2int_rn(t);
    return __double2ll_rn((double)g * DIE_FIX_ONE);
}

// "no agent on this cell" in the deposit plane of the tile-binned step (a NaN no deposit can carry)
#define DIE_DEP_EMPTY 0xFFFFFFFFu

// ---- tile geometry (die_medium.gW/gH/ox/oy) ----------------------------------------------------
struct die_geo {
    int W, H;        // local plane dims (pitch = H)
    int gW, gH;      // world size used for coordinate → cell
    int ox, oy;      // global cell of local element (0, 0)
    int own_x0, own_y0, own_x1, own_y1;   // local cells this rank accounts for in reward / num_alive (x1 == 0: all)
};
static inline die_geo die_geo_of(const die_medium* m) {
    die_geo g;
    g.W = m->W; g.H = m->H;
    g.gW = m->gW > 0 ? m->gW : m->W; g.gH = m->gW > 0 ? m->gH : m->H;
    g.ox = m->gW > 0 ? m->ox : 0; g.oy = m->gW > 0 ? m->oy : 0;
    const bool own = m->gW > 0 && m->own_x1 > 0;
    g.own_x0 = own ? m->own_x0 : 0; g.own_y0 = own ? m->own_y0 : 0;
    g.own_x1 = own ? m->own_x1 : 0; g.own_y1 = own ? m->own_y1 : 0;
    return g;
}
// world cell → element of the local planes.  In a decomposed world the halo holds periodic images, so a
// cell on the far side of the world's seam maps into the halo (gx − ox modulo gW); then clamped: a slot can
// for one call sit outside the tile that holds it (zeroed by the lifecycle, about to migrate) and must never
// turn into an out-of-bounds access.  For periodic single-tile planes all of this is a no-op.
__device__ __forceinline__ int64_t die_local(const die_geo& g, int gx, int gy) {
    int lx = gx - g.ox, ly = gy - g.oy;
    lx = lx >= g.W ? lx - g.gW : (lx < 0 ? lx + g.gW : lx);      // beyond the planes: try the periodic image
    ly = ly >= g.H ? ly - g.gH : (ly < 0 ? ly + g.gH : ly);
    lx = lx < 0 ? 0 : (lx >= g.W ? g.W - 1 : lx);
    ly = ly < 0 ? 0 : (ly >= g.H ? g.H - 1 : ly);
    return (int64_t)lx * g.H + ly;
}

// world cell → row (or column) of the local planes for the TILE-BINNED kernels: the periodic image inside the planes or, for a
// cell the planes do not hold (an agent that walked out of a decomposed tile's halo), the NEAREST edge — an agent's local
// cell never jumps, so "an agent moves less than a tile per step" survives the plane's edge (die_local above tries one
// periodic image and clamps: fine for a lookup, but it sends the first row past the bottom edge to row 0).
// c in [0, gn), o = global cell of local element 0, n = local extent, gn = world extent
__device__ __forceinline__ int die_plane_coord(int c, int o, int n, int gn) {
    int l = c - o;
    l += l < 0 ? gn : 0;
    l -= l >= gn ? gn : 0;
    return l < n ? l : (l - n < gn - l ? n - 1 : 0);
}

// does this rank account for the agent standing on world cell (gx, gy)?  (ghost-agent decomposition)
__device__ __forceinline__ bool die_owned(const die_geo& g, int gx, int gy) {
    if (g.own_x1 == 0) return true;
    int lx = gx - g.ox, ly = gy - g.oy;
    lx = lx >= g.W ? lx - g.gW : (lx < 0 ? lx + g.gW : lx);
    ly = ly >= g.H ? ly - g.gH : (ly < 0 ? ly + g.gH : ly);
    return lx >= g.own_x0 && lx < g.own_x1 && ly >= g.own_y0 && ly < g.own_y1;
}

// ---- wave / block reductions -------------------------------------------------------------
__device__ __forceinline__ double die_wave_sum(double v) {
#pragma unroll
    for (int o = DIE_WAVE / 2; o > 0; o >>= 1) v += __shfl_down(v, o, DIE_WAVE);
    return v;
}
__device__ __forceinline__ long long die_wave_sum(long long v) {
#pragma unroll
    for (int o = DIE_WAVE / 2; o > 0; o >>= 1) v += __shfl_down(v, o, DIE_WAVE);
    return v;
}

// Groups the active lanes of the wave by key (no memory traffic), then every group's first lane issues ONE atomicAdd of
// the group's size — all of a wave's atomics are in flight together — and the lanes get base + rank inside the group
// (lane order): a unique slot of the bucket's range when `counter` is a cursor.  RETURNING = false: histogram only.
template <bool RETURNING>
__device__ __forceinline__ uint32_t wave_grouped_add(uint32_t* counter, uint32_t key, bool active) {
    const int lane = threadIdx.x & (DIE_WAVE - 1);
    int my_lead = lane;
    uint32_t my_rank = 0, my_count = 0;
    bool todo = active;
    unsigned long long pending = __ballot(todo);
    while (pending) {
        const int lead = __ffsll((long long)pending) - 1;
        const uint32_t kk = __shfl(key, lead, DIE_WAVE);
        const bool match = todo && key == kk;
        const unsigned long long mask = __ballot(match);
        if (match) {
            my_lead = lead;
            my_rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            if (lane == lead) my_count = (uint32_t)__popcll(mask);
            todo = false;
        }
        pending &= ~mask;
    }
    uint32_t base = 0;
    if (active && lane == my_lead) {
        if (RETURNING) base = atomicAdd(&counter[key], my_count);
        else atomicAdd(&counter[key], my_count);
    }
    if (!RETURNING) return 0;
    base = __shfl(base, my_lead, DIE_WAVE);
    return base + my_rank;
}

