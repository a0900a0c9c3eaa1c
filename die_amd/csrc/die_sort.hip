// Spatial re-ordering of the agent arrays (gfx950).  Not a reference function: the reference keeps
// agents in slot order forever (core/data_init.py:133-150).  On device the order of the arrays is
// free — every per-agent computation is keyed by the carried slot id (Philox counters, ownership
// words) — and it decides performance: a wave's 64 gathers / atomics cost one cache line each unless
// neighbouring array entries are neighbours on the grid (scratch/kbench.hip: 2.5 M scattered
// atomicMax 98 µs in random order vs 57 µs bucket-sorted).  Agents are therefore bucket-sorted every
// few steps: key = (ix / 8, iy / 64) bucket in row-major bucket order (a wave then touches
// ≈ 8 rows × 2 lines), LSD radix sort of (key, index) pairs by rocPRIM/hipCUB, one gather pass that
// permutes every per-slot array, the slot ids and any arrays an Agent object attached.
#include "die_common.h"
#include <hipcub/hipcub.hpp>

#define DIE_SORT_MAX_EXTRA 4
#ifndef DIE_SORT_XSHIFT
#define DIE_SORT_XSHIFT 4      // bucket = 2^XSHIFT rows × 2^YSHIFT columns (16×32: best of the shapes tried)
#endif
#ifndef DIE_SORT_YSHIFT
#define DIE_SORT_YSHIFT 5
#endif

__global__ __launch_bounds__(DIE_BLOCK) void k_sort_keys(die_geo g, int64_t N, const uint32_t* x, const uint32_t* y,
                                                         int nby, uint32_t* key, uint32_t* val) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        // row / column inside the local planes (a decomposed tile: world cell − tile origin)
        int ix = die_cell((int64_t)x[n], g.gW) - g.ox, iy = die_cell((int64_t)y[n], g.gH) - g.oy;
        ix = ix < 0 ? 0 : (ix >= g.W ? g.W - 1 : ix);
        iy = iy < 0 ? 0 : (iy >= g.H ? g.H - 1 : iy);
        key[n] = (uint32_t)((ix >> DIE_SORT_XSHIFT) * nby + (iy >> DIE_SORT_YSHIFT));
        val[n] = (uint32_t)n;
    }
}

struct PermArgs {
    int64_t N;
    const uint32_t* idx;        // new position → old position
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

__global__ __launch_bounds__(DIE_BLOCK) void k_permute(PermArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < a.N; j += stride) {
        const uint32_t s = a.idx[j];
        a.ox[j] = a.x[s];
        a.oy[j] = a.y[s];
        a.oalive[j] = a.alive[s];
        a.oagent_food[j] = a.agent_food[s];
        a.oslot[j] = a.slot ? a.slot[s] : s;
#pragma unroll
        for (int e = 0; e < DIE_SORT_MAX_EXTRA; ++e)
            if (e < a.n_extra) a.eout[e][j] = a.ein[e][s];
    }
}

static int key_bits(int W, int H) {
    const int64_t nb = (int64_t)((W >> DIE_SORT_XSHIFT) + 1) * ((H >> DIE_SORT_YSHIFT) + 1);
    int b = 1;
    while (((int64_t)1 << b) < nb) ++b;
    return b;
}

static size_t cub_bytes(int64_t N, int bits) {
    size_t bytes = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                       (uint32_t*)nullptr, (int)N, 0, bits, (hipStream_t)0);
    return bytes;
}

extern "C" int64_t die_sort_workspace_bytes(int32_t W, int32_t H, int64_t N) {
    if (W < 1 || H < 1 || N < 1 || N >= ((int64_t)1 << 31)) return -1;
    const int64_t arr = ((N * 4 + 255) & ~(int64_t)255);
    return 4 * arr + (int64_t)((cub_bytes(N, key_bits(W, H)) + 255) & ~(size_t)255);
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
    const int64_t arr = ((N * 4 + 255) & ~(int64_t)255);
    char* w = (char*)ws;
    uint32_t* key_in = (uint32_t*)w;
    uint32_t* key_out = (uint32_t*)(w + arr);
    uint32_t* val_in = (uint32_t*)(w + 2 * arr);
    uint32_t* val_out = (uint32_t*)(w + 3 * arr);
    void* tmp = w + 4 * arr;
    const int bits = key_bits(m->W, m->H);
    size_t tmp_bytes = cub_bytes(N, bits);
    hipStream_t s = (hipStream_t)stream;
    int64_t g = (N + DIE_BLOCK - 1) / DIE_BLOCK;
    const int grid = (int)(g < 4096 ? g : 4096);
    k_sort_keys<<<grid, DIE_BLOCK, 0, s>>>(die_geo_of(m), N, in->x, in->y, (m->H >> DIE_SORT_YSHIFT) + 1, key_in, val_in);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key_in, key_out, val_in, val_out, (int)N, 0, bits, s);
    if (e != hipSuccess) {
        die_set_error("die_agents_sort: radix sort failed: %s", hipGetErrorString(e));
        return DIE_ERR_HIP;
    }
    PermArgs p;
    p.N = N; p.idx = val_out;
    p.x = in->x; p.y = in->y; p.slot = in->slot; p.alive = in->alive; p.agent_food = in->agent_food;
    p.ox = out->x; p.oy = out->y; p.oslot = out->slot; p.oalive = out->alive; p.oagent_food = out->agent_food;
    p.n_extra = n_extra;
    for (int i = 0; i < DIE_SORT_MAX_EXTRA; ++i) {
        p.ein[i] = i < n_extra ? extra_in[i] : nullptr;
        p.eout[i] = i < n_extra ? extra_out[i] : nullptr;
        DIE_REQUIRE(i >= n_extra || (p.ein[i] && p.eout[i] && p.ein[i] != p.eout[i]), "die_agents_sort: bad attached array %d", i);
    }
    k_permute<<<grid, DIE_BLOCK, 0, s>>>(p);
    DIE_CHECK_LAUNCH("die_agents_sort");
    return DIE_OK;
}
