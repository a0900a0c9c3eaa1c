// micro-benchmark: shapes of a plain streaming copy (16 bytes per lane) on this box — which one is the ceiling?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const u4v* __restrict__ src, u4v* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], &dst[i + u * stride]); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
// contiguous chunk per workgroup (each workgroup streams its own slice)
template <int U>
__global__ __launch_bounds__(256) void k_copy_chunk(const u4v* __restrict__ src, u4v* __restrict__ dst, int64_t n) {
    const int64_t per = (n + gridDim.x - 1) / gridDim.x, lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    int64_t i = lo + threadIdx.x;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
        u4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) dst[i + u * 256] = v[u];
    }
    for (; i < hi; i += 256) dst[i] = src[i];
}
template <class F> static void run(const char* name, F launch, int64_t bytes) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s %7.1f GB/s (read + written)\n", name, 2.0 * bytes * 20 / (ms * 1e-3) / 1e9);
}
int main() {
    const int64_t bytes = 512ll << 20, n = bytes / 16;
    u4v *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes);
    run("hipMemcpyAsync D2D", [&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }, bytes);
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        char nm[96];
        snprintf(nm, 96, "grid-stride, 4 in flight, %d workgroups", g); run(nm, [&] { k_copy<4, false><<<g, 256>>>(s, d, n); }, bytes);
        snprintf(nm, 96, "grid-stride, 8 in flight, %d workgroups", g); run(nm, [&] { k_copy<8, false><<<g, 256>>>(s, d, n); }, bytes);
        snprintf(nm, 96, "grid-stride, 4 in flight, nt, %d workgroups", g); run(nm, [&] { k_copy<4, true><<<g, 256>>>(s, d, n); }, bytes);
        snprintf(nm, 96, "own slice, 4 in flight, %d workgroups", g); run(nm, [&] { k_copy_chunk<4><<<g, 256>>>(s, d, n); }, bytes);
    }
    run("one vector per thread (n / 256 workgroups)", [&] { k_copy<1, false><<<(unsigned)(n / 256), 256>>>(s, d, n); }, bytes);
    return 0;
}
