// micro-test: does global_load_lds_dwordx4 accept a source address that is only 4-byte aligned?  (GPU box only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
extern __shared__ __align__(16) unsigned char smem[];
typedef const void __attribute__((address_space(1)))* gptr;
typedef void __attribute__((address_space(3)))* lptr;
__global__ void k(const uint32_t* src, uint32_t* dst, int shift) {
    uint32_t* buf = (uint32_t*)smem;
    const int lane = threadIdx.x;
    __builtin_amdgcn_global_load_lds((gptr)(src + shift + 4 * lane), (lptr)buf, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) dst[i] = buf[i];
}
int main() {
    uint32_t *s, *d;
    hipMalloc(&s, 4096 * 4); hipMalloc(&d, 256 * 4);
    std::vector<uint32_t> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    hipMemcpy(s, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipMemset(d, 0xFF, 256 * 4);
        k<<<1, 64, 4096>>>(s, d, shift);
        hipError_t e = hipDeviceSynchronize();
        std::vector<uint32_t> o(256);
        hipMemcpy(o.data(), d, 256 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += o[i] != (uint32_t)(i + shift);
        printf("shift %d: %s, %d of 256 words wrong (first words %u %u %u %u %u)\n", shift, hipGetErrorString(e), bad, o[0], o[1], o[2], o[3], o[4]);
    }
    return 0;
}
