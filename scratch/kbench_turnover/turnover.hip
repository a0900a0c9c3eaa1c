// How long does a resident-workgroup slot stay empty between two workgroups?  (Round 5: the step's two kernels keep only 73-78 % of
// their slots busy — profiles/r05_workgroup_timeline_4096.txt.  Is that the hardware's dispatch, or the kernels' own epilogue / store drain?)
// A grid of `nwg` workgroups of `threads` threads with `lds` bytes of dynamic LDS; every workgroup stamps s_memrealtime (100 MHz, one
// clock for the GPU) at its start and at its end and in between either SPINS for ~`us` microseconds (no memory traffic at all) or
// also streams `kb` KB in and out per workgroup (stores outstanding when it ends).  Host: launch span, mean life, workgroups in flight
// on average, and the gap per turnover = life x (slots / in-flight - 1).
// -DWAVES_MAX=6: the compiler pads the register allocation so that at most 6 waves per SIMD fit (3 workgroups of 8 waves per CU by REGISTERS).
// build: hipcc -O3 --offload-arch=gfx950 turnover.hip -o turnover ; run: ./turnover [nwg=4096] [threads=512] [lds=52000] [us=10] [kb=0] [slots=768] [streams=1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long rt() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

#ifndef WAVES_MAX
#define WAVES_MAX 8
#endif
__global__ __attribute__((amdgpu_waves_per_eu(1, WAVES_MAX))) void k_turn(unsigned long long* stamps, int ticks, const uint4* src, uint4* dst, int vec_per_wg, int first) {
    extern __shared__ unsigned char smem[];
    const unsigned wg = blockIdx.x + first;
#ifdef VGPR_PAD
    asm volatile("v_mov_b32 v79, 0" ::: "v79");           // 80 VGPRs allocated: 6 waves per SIMD, i.e. THREE 8-wave workgroups per CU by registers
#endif
    const unsigned long long t0 = rt();
    if (threadIdx.x == 0) smem[0] = 1;
    if (vec_per_wg > 0) {                                       // stream in, (spin), stream out: the stores are in flight at the end
        const size_t base = (size_t)wg * vec_per_wg;
        uint4 acc = make_uint4(0, 0, 0, 0);
        for (int i = threadIdx.x; i < vec_per_wg; i += blockDim.x) { const uint4 v = src[base + i]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
#ifdef STORES_FIRST           // the stores go out BEFORE the spin: nothing is outstanding when the workgroup ends
        for (int i = threadIdx.x; i < vec_per_wg; i += blockDim.x) dst[base + i] = acc;
        while ((long long)(rt() - t0) < ticks) __builtin_amdgcn_s_sleep(4);
#else
        while ((long long)(rt() - t0) < ticks) __builtin_amdgcn_s_sleep(4);
        for (int i = threadIdx.x; i < vec_per_wg; i += blockDim.x) dst[base + i] = acc;
#endif
    } else {
        while ((long long)(rt() - t0) < ticks) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
    if (threadIdx.x == 0) { stamps[2 * wg] = t0; stamps[2 * wg + 1] = rt(); }
}

int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 4096, threads = argc > 2 ? atoi(argv[2]) : 512, lds = argc > 3 ? atoi(argv[3]) : 52000;
    const int us = argc > 4 ? atoi(argv[4]) : 10, kb = argc > 5 ? atoi(argv[5]) : 0, slots = argc > 6 ? atoi(argv[6]) : 768;
    const int ns = argc > 7 ? atoi(argv[7]) : 1;              // the grid as `ns` launches of nwg / ns workgroups on `ns` streams, all in flight together
    std::vector<hipStream_t> st(ns);
    for (auto& q : st) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    unsigned long long* d; CK(hipMalloc(&d, (size_t)nwg * 16));
    const int vec = kb * 1024 / 16;
    uint4 *src = nullptr, *dst = nullptr;
    if (vec) { CK(hipMalloc(&src, (size_t)nwg * vec * 16)); CK(hipMalloc(&dst, (size_t)nwg * vec * 16)); CK(hipMemset(src, 1, (size_t)nwg * vec * 16)); }
    CK(hipFuncSetAttribute((const void*)k_turn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int rep = 0; rep < 3; ++rep) {
        for (int q = 0; q < ns; ++q) k_turn<<<nwg / ns, threads, lds, st[q]>>>(d, us * 100, src, dst, vec, q * (nwg / ns));
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(2 * (size_t)nwg);
    CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long s0 = ~0ull, e1 = 0; double life = 0;
    for (int i = 0; i < nwg; ++i) { s0 = std::min(s0, h[2 * i]); e1 = std::max(e1, h[2 * i + 1]); life += (double)(h[2 * i + 1] - h[2 * i]); }
    const double dur = (double)(e1 - s0) * 0.01, mean = life / nwg * 0.01, inflight = life * 0.01 / dur;
    if (nwg % slots == 0)      // whole rounds: every slot turns over nwg / slots − 1 times, no partial last round in the average
        printf("   WHOLE ROUNDS (%d): span / rounds - life = %.2f us per turnover (incl. 1/rounds of the launch ramp)\n", nwg / slots, dur / (nwg / slots) - mean);
    printf("%d stream(s): nwg %d x %d threads, %d B LDS, spin %d us, %d KB in + out per workgroup: launch %.1f us, life mean %.2f us, %.0f workgroups in flight on average of %d slots "
           "(%.0f %%), gap per turnover %.2f us\n", ns, nwg, threads, lds, us, kb, dur, mean, inflight, slots, 100.0 * inflight / slots, mean * (slots / inflight - 1.0));
    return 0;
}
