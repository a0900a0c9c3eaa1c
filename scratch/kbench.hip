// Micro-benchmarks of divergent-access rates on MI355X for the agent kernels' access shapes.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 scratch/kbench.hip -o scratch/kbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_access(const uint32_t* idx, int64_t n, float* table, uint32_t* utable, float* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n; i += stride) {
        uint32_t c = idx[i];
        if (MODE == 0) acc += table[c];                          // gather
        if (MODE == 1) table[c] = (float)i;                      // scattered store
        if (MODE == 2) atomicMax(&utable[c], (uint32_t)i);       // scattered atomic max (no return)
        if (MODE == 3) table[c] = table[c] + 1.0f;               // scattered RMW
        if (MODE == 4) atomicAdd(&table[c], 1.0f);               // float atomic add
        if (MODE == 6) atomicMax(&((unsigned long long*)table)[c], ((unsigned long long)i << 32) | 7ull);   // u64 atomic max, 8-B cells
        if (MODE == 5) acc += table[c] + table[c + 4096] + table[c + 8192] + table[c + 1] ; // 4 gathers 3 rows
        if (MODE == 7 || MODE == 8) {                            // the probe's 4 taps: (x±1, y), (x, y±1); 7 = linear plane, 8 = 4x8-cell tiles of 128 B
            const int x = (int)(c >> 12) + 1, y = (int)(c & 4095);
            const int ym = y > 0 ? y - 1 : 0, yp = y < 4095 ? y + 1 : 4095;
            auto at = [&](int xx, int yy) -> int64_t {
                if (MODE == 7) return (int64_t)xx * 4096 + yy;
                return ((int64_t)(xx >> 2) * (4096 >> 3) + (yy >> 3)) * 32 + (xx & 3) * 8 + (yy & 7);
            };
            acc += table[at(x - 1, y)] + table[at(x + 1, y)] + table[at(x, ym)] + table[at(x, yp)];
        }
    }
    if (MODE == 0 || MODE == 5 || MODE == 7 || MODE == 8) { if (acc == 12345.678f) out[0] = acc; }
}

template <int MODE>
float run(const uint32_t* idx, int64_t n, float* table, float* out, int grid, int reps = 7) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        k_access<MODE><<<grid, 256>>>(idx, n, table, (uint32_t*)table, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    const int W = 4096, H = 4096; const int64_t C = (int64_t)W * H;
    const int64_t K = 2518584;
    // index sets: (a) spatially ordered sparse (every ~6.67th cell, jittered), (b) random permutation of (a),
    // (c) locally shuffled within windows of 4096 agents (≈ tile-sorted but unordered inside)
    std::mt19937 rng(1);
    std::vector<uint32_t> ordered; ordered.reserve(K);
    { std::uniform_real_distribution<double> u(0, 1); for (int64_t c = 0; c < C - 8200 && (int64_t)ordered.size() < K; ++c) if (u(rng) < 0.1502) ordered.push_back((uint32_t)c); }
    int64_t n = ordered.size();
    std::vector<uint32_t> shuffled = ordered; std::shuffle(shuffled.begin(), shuffled.end(), rng);
    std::vector<uint32_t> local = ordered; for (int64_t s = 0; s + 4096 <= n; s += 4096) std::shuffle(local.begin() + s, local.begin() + s + 4096, rng);
    // tile order: agents sorted by 64x64 tile, random inside the tile
    std::vector<uint32_t> tiled = ordered;
    std::stable_sort(tiled.begin(), tiled.end(), [&](uint32_t a, uint32_t b) { auto t = [&](uint32_t c) { return ((c / H) / 64) * (H / 64) + ((c % H) / 64); }; return t(a) < t(b); });
    // bucket order: sort by (ix/8, iy/64) bucket, random inside the bucket; then displaced copies
    auto bucket_sorted = [&](int disp) {
        std::vector<uint32_t> v = ordered;
        auto b = [&](uint32_t c) { return ((c / H) / 8) * (H / 64) + ((c % H) / 64); };
        std::shuffle(v.begin(), v.end(), rng);
        std::stable_sort(v.begin(), v.end(), [&](uint32_t a, uint32_t c2) { return b(a) < b(c2); });
        if (disp) { std::uniform_int_distribution<int> d(-disp, disp); for (auto& c : v) { int ix = (int)(c / H) + d(rng), iy = (int)(c % H) + d(rng); ix = std::min(std::max(ix, 0), W - 4); iy = std::min(std::max(iy, 0), H - 2); c = (uint32_t)(ix * H + iy); } }
        return v;
    };
    std::vector<uint32_t> bk0 = bucket_sorted(0), bk3 = bucket_sorted(3), bk6 = bucket_sorted(6), bk12 = bucket_sorted(12);
    std::vector<uint32_t> ord3 = ordered; { std::uniform_int_distribution<int> d(-3, 3); for (auto& c : ord3) { int ix = (int)(c / H) + d(rng), iy = (int)(c % H) + d(rng); ix = std::min(std::max(ix, 0), W - 4); iy = std::min(std::max(iy, 0), H - 2); c = (uint32_t)(ix * H + iy); } }
    uint32_t *d_idx; float *table, *out;
    CK(hipMalloc(&d_idx, n * 4)); CK(hipMalloc(&table, C * 8)); CK(hipMalloc(&out, 16)); CK(hipMemset(table, 0, C * 8));
    const char* names[] = {"gather", "store", "atomicMax", "rmw", "atomicAddF", "gather4"};
    struct { const char* name; std::vector<uint32_t>* v; } sets[] = {{"ordered", &ordered}, {"ordered+-3", &ord3}, {"bucket8x64", &bk0}, {"bucket+-3", &bk3}, {"bucket+-6", &bk6}, {"bucket+-12", &bk12}, {"random", &shuffled}};
    for (int grid : {2048}) {
        for (auto& s : sets) {
            CK(hipMemcpy(d_idx, s.v->data(), n * 4, hipMemcpyHostToDevice));
            float t[6];
            t[0] = run<0>(d_idx, n, table, out, grid); t[1] = run<1>(d_idx, n, table, out, grid); t[2] = run<2>(d_idx, n, table, out, grid);
            t[3] = run<3>(d_idx, n, table, out, grid); t[4] = run<4>(d_idx, n, table, out, grid); t[5] = run<5>(d_idx, n, table, out, grid);
            float t6 = run<6>(d_idx, n, table, out, grid);
            float t7 = run<7>(d_idx, n, table, out, grid), t8 = run<8>(d_idx, n, table, out, grid);
            { std::vector<uint32_t> h2(*s.v); CK(hipMemcpy(d_idx, h2.data(), n * 4, hipMemcpyHostToDevice)); }
            printf("grid %5d %-10s n=%lld : taps linear %.1f us, taps 4x8-tiled %.1f us | atomicMax64 %.1f us ", grid, s.name, (long long)n, t7, t8, t6);
            for (int m = 0; m < 6; ++m) printf("  %s %.1f us (%.0f G/s)", names[m], t[m], n / t[m] * 1e-3 * (m == 5 ? 4 : 1));
            printf("\n");
        }
    }
    return 0;
}
