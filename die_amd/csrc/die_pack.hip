// Pack / unpack helpers of the decomposed step (die_amd/dist.py): rectangular blocks of the padded
// planes ⇄ contiguous message buffers (halo exchange), and agent records ⇄ per-agent arrays
// (migration).  One launch moves every block / every array of a message instead of one tiny copy each.
#include "die_common.h"

#define DIE_PACK_MAX 16

struct RectArgs {
    int n;
    char* plane[DIE_PACK_MAX];
    int pitch[DIE_PACK_MAX], r0[DIE_PACK_MAX], c0[DIE_PACK_MAX], cols[DIE_PACK_MAX], esz[DIE_PACK_MAX];
    int64_t first[DIE_PACK_MAX + 1];   // prefix sums of element counts
    int64_t boff[DIE_PACK_MAX];        // byte offset of each block inside the buffer
    char* buf;
};

// MODE 0: plane → buffer, 1: buffer → plane, 2: plane = max(plane, buffer) on unsigned 64-bit words (claim merge)
template <int MODE>
__global__ __launch_bounds__(DIE_BLOCK) void k_rects(RectArgs a) {
    const int64_t total = a.first[a.n];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int k = 0;
        while (i >= a.first[k + 1]) ++k;
        const int64_t e = i - a.first[k];
        const int r = (int)(e / a.cols[k]), c = (int)(e - (int64_t)r * a.cols[k]);
        char* p = a.plane[k] + ((int64_t)(a.r0[k] + r) * a.pitch[k] + a.c0[k] + c) * a.esz[k];
        char* b = a.buf + a.boff[k] + e * a.esz[k];
        if (MODE == 2) {
            // blocks of one message may overlap (a corner band lies inside two edge bands): atomic
            atomicMax((unsigned long long*)p, *(const unsigned long long*)b);
        } else if (a.esz[k] == 8) { if (MODE == 0) *(uint64_t*)b = *(const uint64_t*)p; else *(uint64_t*)p = *(const uint64_t*)b; }
        else if (a.esz[k] == 4) { if (MODE == 0) *(uint32_t*)b = *(const uint32_t*)p; else *(uint32_t*)p = *(const uint32_t*)b; }
        else { if (MODE == 0) *(uint16_t*)b = *(const uint16_t*)p; else *(uint16_t*)p = *(const uint16_t*)b; }
    }
}

// The same copy in 16-byte vectors, one vector per thread (the streaming shape: die_stream_copy below), 32-bit index arithmetic:
// when every block's columns, pitch, first column and buffer offset are whole vectors (the halo bands of a decomposed rank are:
// halos are whole tiles wide).  The element-wise kernel above, with its 64-bit division and its grid-stride loop, moved the 35 MB of
// a rank's eight field bands in 27–33 µs; this one in ≈ 10.
template <int MODE>
__global__ __launch_bounds__(DIE_BLOCK) void k_rects_vec(RectArgs a, int vshift) {       // a.first / cols / c0 / pitch in VECTORS, boff in bytes
    const uint32_t total = (uint32_t)a.first[a.n];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int k = 0;
    while (i >= (uint32_t)a.first[k + 1]) ++k;
    const uint32_t e = i - (uint32_t)a.first[k], cols = (uint32_t)a.cols[k];
    const uint32_t r = e / cols, c = e - r * cols;
    uint4* p = (uint4*)a.plane[k] + ((size_t)(a.r0[k] + r) * a.pitch[k] + a.c0[k] + c);
    uint4* b = (uint4*)(a.buf + a.boff[k]) + e;
    if (MODE == 0) *b = *p; else *p = *b;
}

static int rects(const die_rect* r, int32_t n, void* buf, int mode, void* stream, const char* who) {
    DIE_REQUIRE(r && buf && n >= 1 && n <= DIE_PACK_MAX, "%s: 1..%d blocks", who, DIE_PACK_MAX);
    RectArgs a;
    a.n = n; a.buf = (char*)buf; a.first[0] = 0;
    for (int k = 0; k < n; ++k) {
        DIE_REQUIRE(r[k].plane && r[k].r1 > r[k].r0 && r[k].c1 > r[k].c0 && r[k].r0 >= 0 && r[k].c0 >= 0 && r[k].c1 <= r[k].pitch,
                    "%s: bad block %d", who, k);
        DIE_REQUIRE(r[k].elem_bytes == 2 || r[k].elem_bytes == 4 || r[k].elem_bytes == 8, "%s: element size %d", who, r[k].elem_bytes);
        DIE_REQUIRE(r[k].buf_offset % r[k].elem_bytes == 0, "%s: misaligned block %d", who, k);
        a.plane[k] = (char*)r[k].plane; a.pitch[k] = r[k].pitch; a.r0[k] = r[k].r0; a.c0[k] = r[k].c0;
        a.cols[k] = r[k].c1 - r[k].c0; a.esz[k] = r[k].elem_bytes; a.boff[k] = r[k].buf_offset;
        a.first[k + 1] = a.first[k] + (int64_t)(r[k].r1 - r[k].r0) * a.cols[k];
    }
    for (int k = n; k < DIE_PACK_MAX; ++k) { a.plane[k] = nullptr; a.pitch[k] = a.r0[k] = a.c0[k] = a.cols[k] = a.esz[k] = 0; a.boff[k] = 0; a.first[k + 1] = a.first[n]; }
    if (mode == 2)
        for (int k = 0; k < n; ++k) DIE_REQUIRE(a.esz[k] == 8, "%s: max-merge is for 8-byte claim words", who);
    if (mode != 2) {                                           // the vector form where every block is made of whole 16-byte vectors
        bool vec = ((uintptr_t)buf % 16) == 0 && a.first[n] / 2 < (1ll << 31);
        for (int k = 0; vec && k < n; ++k) {
            const int V = 16 / a.esz[k];
            vec = a.esz[k] == a.esz[0] && a.cols[k] % V == 0 && a.c0[k] % V == 0 && a.pitch[k] % V == 0 && a.boff[k] % 16 == 0 &&
                  ((uintptr_t)a.plane[k] % 16) == 0;
        }
        if (vec) {
            RectArgs v = a;
            const int V = 16 / a.esz[0];
            for (int k = 0; k < n; ++k) { v.cols[k] = a.cols[k] / V; v.c0[k] = a.c0[k] / V; v.pitch[k] = a.pitch[k] / V; v.first[k + 1] = v.first[k] + (a.first[k + 1] - a.first[k]) / V; }
            for (int k = n; k < DIE_PACK_MAX; ++k) v.first[k + 1] = v.first[n];
            const int64_t gv = (v.first[n] + DIE_BLOCK - 1) / DIE_BLOCK;
            if (gv > 0 && gv < (1ll << 31)) {
                if (mode == 0) k_rects_vec<0><<<(int)gv, DIE_BLOCK, 0, (hipStream_t)stream>>>(v, 0);
                else k_rects_vec<1><<<(int)gv, DIE_BLOCK, 0, (hipStream_t)stream>>>(v, 0);
                DIE_CHECK_LAUNCH(who);
                return DIE_OK;
            }
        }
    }
    int64_t g = (a.first[n] + DIE_BLOCK - 1) / DIE_BLOCK;
    const int grid = (int)(g < 2048 ? (g > 0 ? g : 1) : 2048);
    if (mode == 0) k_rects<0><<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>(a);
    else if (mode == 1) k_rects<1><<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>(a);
    else k_rects<2><<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>(a);
    DIE_CHECK_LAUNCH(who);
    return DIE_OK;
}

extern "C" int die_rects_pack(const die_rect* r, int32_t n, void* buf, void* stream) { return rects(r, n, buf, 0, stream, "die_rects_pack"); }
extern "C" int die_rects_unpack(const die_rect* r, int32_t n, const void* buf, void* stream) { return rects(r, n, (void*)buf, 1, stream, "die_rects_unpack"); }
extern "C" int die_rects_unpack_max(const die_rect* r, int32_t n, const void* buf, void* stream) { return rects(r, n, (void*)buf, 2, stream, "die_rects_unpack_max"); }

struct RecArgs {
    int n;
    char* arr[DIE_PACK_MAX];
    int esz[DIE_PACK_MAX];
    const int64_t* idx;
    int64_t count;
    int32_t* rec;       // (n, count) 4-byte words, row-major
};

template <bool GATHER>
__global__ __launch_bounds__(DIE_BLOCK) void k_records(RecArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < a.count; j += stride) {
        const int64_t s = a.idx[j];
        for (int k = 0; k < a.n; ++k) {
            int32_t* r = a.rec + (int64_t)k * a.count + j;
            if (a.esz[k] == 4) { if (GATHER) *r = ((const int32_t*)a.arr[k])[s]; else ((int32_t*)a.arr[k])[s] = *r; }
            else { if (GATHER) *r = (int32_t)((const uint8_t*)a.arr[k])[s]; else ((uint8_t*)a.arr[k])[s] = (uint8_t)*r; }
        }
    }
}

static int records(void* const* arrays, const int32_t* esz, int32_t n, const int64_t* idx, int64_t count, int32_t* rec, bool gather,
                   void* stream, const char* who) {
    DIE_REQUIRE(arrays && esz && n >= 1 && n <= DIE_PACK_MAX, "%s: 1..%d arrays", who, DIE_PACK_MAX);
    if (count == 0) return DIE_OK;
    DIE_REQUIRE(idx && rec && count > 0, "%s: null index / record buffer", who);
    RecArgs a;
    a.n = n; a.idx = idx; a.count = count; a.rec = rec;
    for (int k = 0; k < DIE_PACK_MAX; ++k) {
        a.arr[k] = k < n ? (char*)arrays[k] : nullptr;
        a.esz[k] = k < n ? esz[k] : 0;
        DIE_REQUIRE(k >= n || (arrays[k] && (esz[k] == 4 || esz[k] == 1)), "%s: array %d must be 4- or 1-byte", who, k);
    }
    int64_t g = (count + DIE_BLOCK - 1) / DIE_BLOCK;
    const int grid = (int)(g < 2048 ? g : 2048);
    if (gather) k_records<true><<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>(a);
    else k_records<false><<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>(a);
    DIE_CHECK_LAUNCH(who);
    return DIE_OK;
}

extern "C" int die_records_gather(void* const* arrays, const int32_t* elem_bytes, int32_t n, const int64_t* idx, int64_t count,
                                  int32_t* records_out, void* stream) {
    return records(arrays, elem_bytes, n, idx, count, records_out, true, stream, "die_records_gather");
}
extern "C" int die_records_scatter(void* const* arrays, const int32_t* elem_bytes, int32_t n, const int64_t* idx, int64_t count,
                                   const int32_t* records_in, void* stream) {
    return records(arrays, elem_bytes, n, idx, count, (int32_t*)records_in, false, stream, "die_records_scatter");
}


// ---- a plain streaming copy, for the bench's ceiling ------------------------------------------------------------------
// What this GPU moves when nothing but a stream is asked of it: ONE 16-byte vector per thread, bytes / 4096 workgroups of 256
// threads.  The shape matters (scratch/kbench_dma/copy_sweep.hip, 512 MB, one MI355X): this one 6.21 TB/s (MI355X_MICROARCH.md:
// 6.29 for a float4 copy); a workgroup per contiguous slice with four vectors in flight 5.35–5.69; grid-stride loops of 1 024 –
// 16 384 workgroups 4.3–5.4; hipMemcpyAsync / torch's copy_ 4.8–4.9 (the "copy ceiling" rounds 1–3 quoted).  Short-lived
// workgroups dispatched in address order sweep the memory sequentially; a grid-stride loop makes every workgroup hop through the
// whole buffer.  bench.py times it through the C ABI and quotes both step kernels against it (roofline.stream_ceiling_gbs).
__global__ __launch_bounds__(256) void k_stream_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

extern "C" int die_stream_copy(const void* src, void* dst, int64_t bytes, void* stream) {
    DIE_REQUIRE(src && dst && bytes > 0 && bytes % 16 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0,
                "die_stream_copy: 16-byte aligned buffers of a multiple of 16 bytes");
    const int64_t n = bytes / 16, g = (n + 255) / 256;
    DIE_REQUIRE(g < ((int64_t)1 << 31), "die_stream_copy: at most 2^31 workgroups (8 TiB)");
    k_stream_copy<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>((const uint4*)src, (uint4*)dst, n);
    DIE_CHECK_LAUNCH("die_stream_copy");
    return DIE_OK;
}

// The device address of pinned host memory (hipHostGetDevicePointer) through the runtime instance this library is linked
// against — the one that launches the kernels.  die_amd/env.py uses it for the synchronous step's result buffer (round 3 opened
// libamdhip64 by its SONAME with ctypes: on another ROCm that can load a second runtime into the process — ADVICE r3).
extern "C" int die_host_device_pointer(void* host, int32_t device, void** dev_out) {
    DIE_REQUIRE(host && dev_out, "die_host_device_pointer: null argument");
    int cur = -1;
    hipError_t e = hipGetDevice(&cur);
    if (e == hipSuccess && device >= 0 && device != cur) e = hipSetDevice(device);
    if (e == hipSuccess) e = hipHostGetDevicePointer(dev_out, host, 0);
    if (cur >= 0 && device >= 0 && device != cur) (void)hipSetDevice(cur);
    if (e != hipSuccess) { (void)hipGetLastError(); die_set_error("die_host_device_pointer: %s", hipGetErrorString(e)); return DIE_ERR_HIP; }
    return DIE_OK;
}
