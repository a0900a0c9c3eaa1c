// Is it cheaper to stage the bounding box of a workgroup's probe cells in LDS than to gather 4 taps per agent
// from global memory?  Agents bucket-sorted (BX x BY cells per bucket) with +-D cells of drift; probes 11 cells away.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define W 4096
#define H 4096

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_global(const int* px, const int* py, int64_t n, const float* chem, float* out) {
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    int x = px[i], y = py[i];
    float v = chem[(int64_t)(x - 1) * H + y] + chem[(int64_t)(x + 1) * H + y] + chem[(int64_t)x * H + y - 1] + chem[(int64_t)x * H + y + 1];
    out[i] = v;
}

template <int BLOCK, int MAXCELLS>
__global__ __launch_bounds__(BLOCK) void k_lds(const int* px, const int* py, int64_t n, const float* chem, float* out, int* nfallback) {
    __shared__ float patch[MAXCELLS];
    __shared__ int s_min[2], s_max[2];
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool live = i < n;
    int x = live ? px[i] : 0, y = live ? py[i] : 0;
    if (threadIdx.x == 0) { s_min[0] = s_min[1] = 1 << 30; s_max[0] = s_max[1] = -1; }
    __syncthreads();
    if (live) { atomicMin(&s_min[0], x - 1); atomicMax(&s_max[0], x + 1); atomicMin(&s_min[1], y - 1); atomicMax(&s_max[1], y + 1); }
    __syncthreads();
    const int x0 = s_min[0], y0 = s_min[1] & ~3, rows = s_max[0] - x0 + 1, cols = ((s_max[1] - y0 + 1) + 3) & ~3;
    float v;
    if (rows * cols <= MAXCELLS) {
        const int c4 = cols >> 2;
        for (int k = threadIdx.x; k < rows * c4; k += BLOCK) {
            const int r = k / c4, c = (k - r * c4) << 2;
            const float4 t = *(const float4*)(chem + (int64_t)(x0 + r) * H + y0 + c);
            *(float4*)(patch + r * cols + c) = t;
        }
        __syncthreads();
        const int lx = x - x0, ly = y - y0;
        v = live ? patch[(lx - 1) * cols + ly] + patch[(lx + 1) * cols + ly] + patch[lx * cols + ly - 1] + patch[lx * cols + ly + 1] : 0.f;
    } else {
        if (threadIdx.x == 0) atomicAdd(nfallback, 1);
        v = live ? chem[(int64_t)(x - 1) * H + y] + chem[(int64_t)(x + 1) * H + y] + chem[(int64_t)x * H + y - 1] + chem[(int64_t)x * H + y + 1] : 0.f;
    }
    if (live) out[i] = v;
}

template <typename F> float timeit(F f, int reps = 9) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms * 1e3f); }
    std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2];
}

int main() {
    std::mt19937 rng(1);
    std::vector<int> ax, ay;
    { std::uniform_real_distribution<double> u(0, 1); for (int x = 16; x < W - 16; ++x) for (int y = 16; y < H - 16; ++y) if (u(rng) < 0.15) { ax.push_back(x); ay.push_back(y); } }
    const int64_t n = ax.size();
    float* chem; CK(hipMalloc(&chem, (size_t)W * H * 4)); CK(hipMemset(chem, 0, (size_t)W * H * 4));
    int *dx, *dy, *nf; float* out; CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dy, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&nf, 4));
    for (int bxs : {4, 5, 6}) for (int bys : {5, 6, 7}) for (int D : {0, 6, 12}) {
        std::vector<int64_t> ord(n); for (int64_t i = 0; i < n; ++i) ord[i] = i;
        std::shuffle(ord.begin(), ord.end(), rng);
        const int nby = (H >> bys) + 1;
        std::stable_sort(ord.begin(), ord.end(), [&](int64_t a, int64_t b) { return (ax[a] >> bxs) * nby + (ay[a] >> bys) < (ax[b] >> bxs) * nby + (ay[b] >> bys); });
        std::uniform_int_distribution<int> dd(-D, D), ang(0, 11);
        std::vector<int> px(n), py(n);
        for (int64_t i = 0; i < n; ++i) {
            int x = ax[ord[i]] + (D ? dd(rng) : 0), y = ay[ord[i]] + (D ? dd(rng) : 0);
            double a = ang(rng) * 0.5235987755982988;
            x += (int)lrint(10.2 * cos(a)); y += (int)lrint(10.2 * sin(a));
            px[i] = std::min(std::max(x, 1), W - 2); py[i] = std::min(std::max(y, 1), H - 2);
        }
        CK(hipMemcpy(dx, px.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, py.data(), n * 4, hipMemcpyHostToDevice));
        float tg = timeit([&] { k_global<256><<<(n + 255) / 256, 256>>>(dx, dy, n, chem, out); });
        CK(hipMemset(nf, 0, 4));
        float t256 = timeit([&] { k_lds<256, 12288><<<(n + 255) / 256, 256>>>(dx, dy, n, chem, out, nf); });
        int f256; CK(hipMemcpy(&f256, nf, 4, hipMemcpyDeviceToHost)); CK(hipMemset(nf, 0, 4));
        float t1024 = timeit([&] { k_lds<1024, 16384><<<(n + 1023) / 1024, 1024>>>(dx, dy, n, chem, out, nf); });
        int f1024; CK(hipMemcpy(&f1024, nf, 4, hipMemcpyDeviceToHost));
        printf("bucket %2dx%3d drift %2d: global %.1f us | lds wg256 %.1f us (fallback %.1f%%) | lds wg1024 %.1f us (fallback %.1f%%)\n",
               1 << bxs, 1 << bys, D, tg, t256, 100.0 * f256 / 9 / ((n + 255) / 256), t1024, 100.0 * f1024 / 9 / ((n + 1023) / 1024));
    }
    return 0;
}
