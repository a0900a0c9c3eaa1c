// What does a grid-wide barrier cost on MI355X beside a kernel boundary?  (one cooperative launch for small
// worlds, agent phase | grid barrier | field phase, against two launches.)
//   A: `iters` × (kernel, kernel) of empty bodies, back to back on one stream          → time per kernel boundary
//   B: ONE kernel, `iters` grid barriers (arrive: one atomic per workgroup; wait: spin on a generation word, s_sleep between polls)
//   C: as B, with the barrier's release through a device-scope fence on data every workgroup wrote before it (what the step needs:
//      the field phase reads the agent phase's stores of OTHER workgroups)
// build: hipcc -O3 --offload-arch=gfx950 gridbar.hip -o gridbar ; run: ./gridbar [workgroups=256] [threads=512] [iters=2000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_empty(unsigned* sink) { if (sink && threadIdx.x == 1024) sink[0] = 1; }

__device__ __forceinline__ bool grid_barrier(unsigned* count, volatile unsigned* gen, unsigned nwg, unsigned& my_gen, unsigned* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();                                       // this workgroup's stores are visible device-wide
        const unsigned g = my_gen;
        if (atomicAdd(count, 1u) == nwg - 1u) {
            *count = 0u;
            __threadfence();
            atomicAdd((unsigned*)gen, 1u);                     // release
        } else {
            unsigned spins = 0;
            while (__atomic_load_n((unsigned*)gen, __ATOMIC_RELAXED) == g) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { atomicOr(err, 1u); ok = false; break; }     // (never hang the box)
            }
        }
        __threadfence();
        my_gen = g + 1u;
    }
    __syncthreads();
    return ok;
}

// D: two levels — groups of 16 workgroups arrive on their own counter (own 128-byte line), the last of a group on the root;
// the release is one word per group (16 pollers per line instead of 256 on one)
__device__ __forceinline__ bool grid_barrier2(unsigned* w, unsigned nwg, unsigned& my_gen, unsigned* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned g = my_gen, grp = blockIdx.x >> 4, ngrp = (nwg + 15u) >> 4;
        const unsigned members = min(16u, nwg - (grp << 4));
        unsigned* cnt = w + 64 + grp * 32;                      // group counters, one line each
        volatile unsigned* rel = w + 64 + grp * 32 + 1;         // group release words (same line as the counter: written once per barrier)
        bool releaser = false;
        if (atomicAdd(cnt, 1u) == members - 1u) {
            *cnt = 0u;
            if (atomicAdd(w + 32, 1u) == ngrp - 1u) { w[32] = 0u; releaser = true; }
        }
        if (releaser) {
            __threadfence();
            for (unsigned q = 0; q < ngrp; ++q) __atomic_store_n(w + 64 + q * 32 + 1, g + 1u, __ATOMIC_RELEASE);
        } else {
            unsigned spins = 0;
            while (__atomic_load_n((unsigned*)rel, __ATOMIC_RELAXED) == g) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { atomicOr(err, 1u); ok = false; break; }
            }
        }
        __threadfence();
        my_gen = g + 1u;
    }
    __syncthreads();
    return ok;
}

template <bool DATA>
__global__ void k_barriers2(unsigned* w, unsigned* err, int iters, float* data, float* out) {
    unsigned my_gen = 0;
    float acc = 0.f;
    const unsigned nwg = gridDim.x;
    for (int i = 0; i < iters; ++i) {
        if (DATA) data[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (float)i + acc;
        if (!grid_barrier2(w, nwg, my_gen, err)) break;
        if (DATA) acc += __builtin_nontemporal_load(&data[(size_t)((blockIdx.x + 1) % nwg) * blockDim.x + threadIdx.x]) * 1e-9f;
    }
    if (DATA) out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <bool DATA>
__global__ void k_barriers(unsigned* count, unsigned* gen, unsigned* err, int iters, float* data, float* out) {
    unsigned my_gen = 0;
    float acc = 0.f;
    const unsigned nwg = gridDim.x;
    for (int i = 0; i < iters; ++i) {
        if (DATA) data[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (float)i + acc;
        if (!grid_barrier(count, gen, nwg, my_gen, err)) break;
        if (DATA) acc += __builtin_nontemporal_load(&data[(size_t)((blockIdx.x + 1) % nwg) * blockDim.x + threadIdx.x]) * 1e-9f;
    }
    if (DATA) out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256, threads = argc > 2 ? atoi(argv[2]) : 512, iters = argc > 3 ? atoi(argv[3]) : 2000;
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    if (wgs > cus * 2) { printf("refusing %d workgroups on %d CUs: they must all be resident\n", wgs, cus); return 1; }
    unsigned* words; float *data, *out;
    CK(hipMalloc(&words, 65536)); CK(hipMemset(words, 0, 65536));
    CK(hipMalloc(&data, (size_t)wgs * threads * 4)); CK(hipMalloc(&out, (size_t)wgs * threads * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) { k_empty<<<wgs, threads>>>(nullptr); k_empty<<<wgs, threads>>>(nullptr); }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("A  %d x %d threads: %d x (kernel, kernel): %.2f us per kernel boundary (launch + drain of an empty kernel)\n", wgs, threads, iters, ms * 1e3 / (2.0 * iters));
        CK(hipMemset(words, 0, 64));
        CK(hipEventRecord(e0, 0));
        k_barriers<false><<<wgs, threads>>>(words, words + 1, words + 2, iters, data, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("B  one kernel, %d grid barriers: %.2f us per barrier\n", iters, ms * 1e3 / iters);
        CK(hipMemset(words, 0, 64));
        CK(hipEventRecord(e0, 0));
        k_barriers<true><<<wgs, threads>>>(words, words + 1, words + 2, iters, data, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h[4]; CK(hipMemcpy(h, words, 16, hipMemcpyDeviceToHost));
        printf("C  one kernel, %d x (store, grid barrier, load of the neighbour workgroup's stores): %.2f us per barrier%s\n", iters, ms * 1e3 / iters, h[2] ? "  (SPIN LIMIT HIT)" : "");
        CK(hipMemset(words, 0, 65536));
        CK(hipEventRecord(e0, 0));
        k_barriers2<false><<<wgs, threads>>>(words, words + 2, iters, data, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("D  one kernel, %d two-level grid barriers (groups of 16): %.2f us per barrier\n", iters, ms * 1e3 / iters);
        CK(hipMemset(words, 0, 65536));
        CK(hipEventRecord(e0, 0));
        k_barriers2<true><<<wgs, threads>>>(words, words + 2, iters, data, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, words, 16, hipMemcpyDeviceToHost));
        printf("E  as D with the store / load of the neighbour's data around it: %.2f us per barrier%s\n", iters ? ms * 1e3 / iters : 0.0, h[2] ? "  (SPIN LIMIT HIT)" : "");
    }
    return 0;
}
