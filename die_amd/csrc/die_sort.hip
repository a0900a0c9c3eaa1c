// Spatial re-ordering of the agent arrays (gfx950).  Not a reference function: the reference keeps
// agents in slot order forever (core/data_init.py:133-150).  On device the order of the arrays is
// free — every per-agent computation is keyed by the carried slot id (Philox counters, ownership
// words) — and it decides performance: a wave's 64 gathers / atomics cost one cache line each unless
// neighbouring array entries are neighbours on the grid (scratch/kbench.hip: 2.5 M scattered
// atomicMax 98 µs in random order vs 57 µs bucket-sorted).  Agents are therefore bucket-sorted every
// few steps: key = (ix / 16, iy / 32) bucket in row-major bucket order, then a counting sort written for this
// case — histogram over the buckets (wave-aggregated atomics), one-workgroup exclusive scan, and a scatter pass that
// takes positions from per-bucket cursors and moves every per-slot array, the slot ids and any arrays an Agent object
// attached in the same sweep: 4 launches, ≈ 60 µs at 2.5 M agents against ≈ 125–157 µs for hipCUB's LSD radix sort
// of (key, index) pairs (two onesweep passes + five 5-µs memsets) followed by a gather pass.  The order INSIDE a
// bucket is whatever order the waves reach the cursors in: nothing observable depends on it (results are keyed by
// slot id, the reward is summed in fixed point).
#include "die_common.h"

#define DIE_SORT_MAX_EXTRA 4
#ifndef DIE_SORT_XSHIFT
#define DIE_SORT_XSHIFT 4      // bucket = 2^XSHIFT rows × 2^YSHIFT columns (16×32: best of the shapes tried)
#endif
#ifndef DIE_SORT_YSHIFT
#define DIE_SORT_YSHIFT 5
#endif

__device__ __forceinline__ uint32_t sort_key(const die_geo& g, uint32_t X, uint32_t Y, int nby) {
    int ix = die_cell((int64_t)X, g.gW) - g.ox, iy = die_cell((int64_t)Y, g.gH) - g.oy;
    ix = ix < 0 ? 0 : (ix >= g.W ? g.W - 1 : ix);
    iy = iy < 0 ? 0 : (iy >= g.H ? g.H - 1 : iy);
    return (uint32_t)((ix >> DIE_SORT_XSHIFT) * nby + (iy >> DIE_SORT_YSHIFT));
}

__global__ __launch_bounds__(DIE_BLOCK) void k_sort_hist(die_geo g, int64_t N, const uint32_t* x, const uint32_t* y, int nby,
                                                         uint32_t* hist) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x; b < N; b += stride) {       // wave-uniform trip count
        const int64_t n = b + threadIdx.x;
        const bool active = n < N;
        wave_grouped_add<false>(hist, active ? sort_key(g, x[n], y[n], nby) : 0u, active);
    }
}

// exclusive scan of the bucket counts by ONE workgroup: cursor[b] = first position of bucket b.  A thread owns `per`
// consecutive buckets (per % 4 == 0: 16-byte loads, all of them in flight before the first use).
#ifndef SORT_SCAN_MAX_PER
#define SORT_SCAN_MAX_PER 64
#endif
__global__ __launch_bounds__(1024) void k_sort_scan(const uint32_t* hist, uint32_t* cursor, int nb, int per) {
    __shared__ uint32_t s[1024];
    const int lo = threadIdx.x * per;
    uint4 v[SORT_SCAN_MAX_PER / 4];
    uint32_t sum = 0;
#pragma unroll
    for (int q = 0; q < SORT_SCAN_MAX_PER / 4; ++q) {
        v[q] = make_uint4(0, 0, 0, 0);
        if (4 * q < per && lo + 4 * q < nb) v[q] = *(const uint4*)(hist + lo + 4 * q);   // nb is padded to a multiple of 4
    }
#pragma unroll
    for (int q = 0; q < SORT_SCAN_MAX_PER / 4; ++q) sum += v[q].x + v[q].y + v[q].z + v[q].w;
    s[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t t = (int)threadIdx.x >= o ? s[threadIdx.x - o] : 0u;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = s[threadIdx.x] - sum;
#pragma unroll
    for (int q = 0; q < SORT_SCAN_MAX_PER / 4; ++q) {
        if (4 * q < per && lo + 4 * q < nb) {
            uint4 o;
            o.x = run; run += v[q].x;
            o.y = run; run += v[q].y;
            o.z = run; run += v[q].z;
            o.w = run; run += v[q].w;
            *(uint4*)(cursor + lo + 4 * q) = o;
        }
    }
}

// many buckets (≥ 64 K: 8192² and up): plain loop per thread
__global__ __launch_bounds__(1024) void k_sort_scan_big(const uint32_t* hist, uint32_t* cursor, int nb) {
    __shared__ uint32_t s[1024];
    const int per = (nb + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(lo + per, nb);
    uint32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += hist[i];
    s[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t t = (int)threadIdx.x >= o ? s[threadIdx.x - o] : 0u;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = s[threadIdx.x] - sum;
    for (int i = lo; i < hi; ++i) { const uint32_t c = hist[i]; cursor[i] = run; run += c; }
}

struct ScatterArgs {
    die_geo g;
    int64_t N;
    int nby;
    uint32_t* cursor;
    const uint32_t *x, *y, *slot;
    const uint8_t* alive;
    const float* agent_food;
    uint32_t *ox, *oy, *oslot;
    uint8_t* oalive;
    float* oagent_food;
    int n_extra;
    const float* ein[DIE_SORT_MAX_EXTRA];
    float* eout[DIE_SORT_MAX_EXTRA];
};

__global__ __launch_bounds__(DIE_BLOCK) void k_sort_scatter(ScatterArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x; b < a.N; b += stride) {
        const int64_t n = b + threadIdx.x;
        const bool active = n < a.N;
        const uint32_t X = active ? a.x[n] : 0u, Y = active ? a.y[n] : 0u;
        const uint32_t j = wave_grouped_add<true>(a.cursor, active ? sort_key(a.g, X, Y, a.nby) : 0u, active);
        if (!active) continue;
        a.ox[j] = X;
        a.oy[j] = Y;
        a.oalive[j] = a.alive[n];
        a.oagent_food[j] = a.agent_food[n];
        a.oslot[j] = a.slot ? a.slot[n] : (uint32_t)n;
#pragma unroll
        for (int e = 0; e < DIE_SORT_MAX_EXTRA; ++e)
            if (e < a.n_extra) a.eout[e][j] = a.ein[e][n];
    }
}

static int64_t n_buckets(int W, int H) { return (int64_t)((W >> DIE_SORT_XSHIFT) + 1) * ((H >> DIE_SORT_YSHIFT) + 1); }

extern "C" int64_t die_sort_workspace_bytes(int32_t W, int32_t H, int64_t N) {
    if (W < 1 || H < 1 || N < 1 || N >= ((int64_t)1 << 31)) return -1;
    return 2 * ((((n_buckets(W, H) + 3) & ~(int64_t)3) * 4 + 255) & ~(int64_t)255);
}

extern "C" int die_agents_sort(const die_medium* m, const die_agents* in, const die_agents* out, int32_t n_extra,
                               const float* const* extra_in, float* const* extra_out, void* ws, int64_t ws_bytes,
                               void* stream) {
    DIE_REQUIRE(m && in && out && ws, "die_agents_sort: null argument");
    DIE_REQUIRE(in->N > 0 && in->N == out->N && in->N < ((int64_t)1 << 31), "die_agents_sort: bad slot counts");
    DIE_REQUIRE(in->x && in->y && in->alive && in->agent_food && out->x && out->y && out->alive && out->agent_food && out->slot,
                "die_agents_sort: null device pointer (out->slot is required)");
    DIE_REQUIRE(in->x != out->x && in->agent_food != out->agent_food, "die_agents_sort: in and out must be different arrays");
    DIE_REQUIRE(n_extra >= 0 && n_extra <= DIE_SORT_MAX_EXTRA, "die_agents_sort: at most %d attached arrays", DIE_SORT_MAX_EXTRA);
    DIE_REQUIRE(n_extra == 0 || (extra_in && extra_out), "die_agents_sort: attached arrays missing");
    const int64_t need = die_sort_workspace_bytes(m->W, m->H, in->N);
    DIE_REQUIRE(ws_bytes >= need, "die_agents_sort: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)need);
    const int64_t N = in->N;
    char* w = (char*)ws;
    hipStream_t s = (hipStream_t)stream;
    int64_t g = (N + DIE_BLOCK - 1) / DIE_BLOCK;
#ifndef DIE_SORT_GRID_CAP
#define DIE_SORT_GRID_CAP 4096
#endif
    const int grid = (int)(g < DIE_SORT_GRID_CAP ? g : DIE_SORT_GRID_CAP);
    const int nby = (m->H >> DIE_SORT_YSHIFT) + 1;
    for (int i = 0; i < n_extra; ++i)
        DIE_REQUIRE(extra_in[i] && extra_out[i] && extra_in[i] != extra_out[i], "die_agents_sort: bad attached array %d", i);
    const int64_t nb = n_buckets(m->W, m->H);
    const int64_t nb4 = (nb + 3) & ~(int64_t)3;            // the scan reads 16 bytes at a time
    uint32_t* hist = (uint32_t*)w;
    uint32_t* cursor = (uint32_t*)(w + ((nb4 * 4 + 255) & ~(int64_t)255));
    hipError_t e = hipMemsetAsync(hist, 0, (size_t)nb4 * 4, s);
    if (e != hipSuccess) { die_set_error("die_agents_sort: memset failed: %s", hipGetErrorString(e)); return DIE_ERR_HIP; }
    k_sort_hist<<<grid, DIE_BLOCK, 0, s>>>(die_geo_of(m), N, in->x, in->y, nby, hist);
    const int per = (int)(((nb4 + 1023) / 1024 + 3) & ~3);
    if (per <= SORT_SCAN_MAX_PER) k_sort_scan<<<1, 1024, 0, s>>>(hist, cursor, (int)nb4, per);
    else k_sort_scan_big<<<1, 1024, 0, s>>>(hist, cursor, (int)nb4);
    ScatterArgs q;
    q.g = die_geo_of(m); q.N = N; q.nby = nby; q.cursor = cursor;
    q.x = in->x; q.y = in->y; q.slot = in->slot; q.alive = in->alive; q.agent_food = in->agent_food;
    q.ox = out->x; q.oy = out->y; q.oslot = out->slot; q.oalive = out->alive; q.oagent_food = out->agent_food;
    q.n_extra = n_extra;
    for (int i = 0; i < DIE_SORT_MAX_EXTRA; ++i) { q.ein[i] = i < n_extra ? extra_in[i] : nullptr; q.eout[i] = i < n_extra ? extra_out[i] : nullptr; }
    k_sort_scatter<<<grid, DIE_BLOCK, 0, s>>>(q);
    DIE_CHECK_LAUNCH("die_agents_sort");
    return DIE_OK;
}
