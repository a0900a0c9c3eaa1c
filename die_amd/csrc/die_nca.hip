// NeuralAutomataAgent sensing (gfx950) — core/agent/evo.py:45-118 (ConvolutionModel: a stack of bias-free Conv2d with
// 'same' padding — `boundary` = torch's padding_mode: 'circular' by default, 'zeros' / 'reflect' / 'replicate' —, one Tanh at the end) and :150-174 (forward: per-agent gather of the transformed medium at
// the agent's cell, core/utils.py:56-65, times the action coefficients).
//
//   k_conv_circular   one layer: out[o, x, y] = Σ_i Σ_a Σ_b w[o, i, a, b] · in[i, (x + a − r) mod W, (y + b − r) mod H]
//                     (torch's Conv2d is a cross-correlation; medium axes are (channel, x, y), so kernel rows run along
//                     x).  A workgroup stages the (16 + 2r) × (64 + 2r) input tile of every input channel in LDS and the
//                     layer's weights next to it; a thread produces 4 consecutive y of every output channel (16-byte
//                     stores).  The first layer reads the medium's own planes: fp32 / fp16 fields, and the 'agents'
//                     channel straight from the claim plane (occupied ⇔ epoch tag).  The last layer applies tanh.
//   k_gather_scale    action[c, n] = plane[c][cell(x_n), cell(y_n)] · coef[c] for EVERY slot (only_alive = False).
//
// Roofline: HBM.  A 3→3 channel 3×3 layer is 81 MAC per cell against 24 bytes per cell (3 planes in, 3 out): 6.75 flop
// per byte, far below the ≈ 20 flop/byte at which fp32 vector math (157 TFLOP/s) meets 8 TB/s — no MFMA: the matrix
// cores would sit idle behind the same memory stream.
#include "die_common.h"

#define NCA_TX 16
#define NCA_TY 64
#define NCA_MAXC 4
#define NCA_MAXK 7

struct ConvArgs {
    const void* in[NCA_MAXC];
    int kind[NCA_MAXC];          // die_conv_plane.kind
    float* out[NCA_MAXC];
    const float* w;              // [cout][cin][k][k]
    int W, H, cin, cout, k, epoch, apply_tanh;
    int pad;                     // die_pad_mode: how cells beyond the field are read ('same' padding of torch's Conv2d)
};

// index of the cell that stands in for coordinate v of an axis of n cells, or −1 for "reads as zero" (torch.nn.functional.pad:
// 'circular' wraps, 'zeros' pads with 0, 'reflect' mirrors WITHOUT repeating the edge cell, 'replicate' repeats it)
__device__ __forceinline__ int nca_pad_index(int v, int n, int mode) {
    if (v >= 0 && v < n) return v;
    if (mode == DIE_PAD_CIRCULAR) { v %= n; return v < 0 ? v + n : v; }
    if (mode == DIE_PAD_ZEROS) return -1;
    if (mode == DIE_PAD_REPLICATE) return v < 0 ? 0 : n - 1;
    if (n == 1) return 0;
    const int period = 2 * (n - 1);                  // 'reflect': … 2 1 | 0 1 2 … n−1 | n−2 n−3 …
    v %= period; v = v < 0 ? v + period : v;
    return v < n ? v : period - v;
}

__device__ __forceinline__ float nca_load(const void* p, int kind, int64_t i, int epoch) {
    if (kind == DIE_PLANE_F32) return ((const float*)p)[i];
    if (kind == DIE_PLANE_F16) return __half2float(((const __half*)p)[i]);
    return die_claim_occupied(((const unsigned long long*)p)[i], epoch) ? 1.f : 0.f;
}

template <int K>
__global__ __launch_bounds__(DIE_BLOCK) void k_conv_circular(ConvArgs a) {
    constexpr int R = K / 2, LX = NCA_TX + 2 * R, LY = NCA_TY + 2 * R + 1;     // odd pitch: conflict-free column walks
    extern __shared__ __align__(16) float nca_smem[];
    float* s_in = nca_smem;                                  // [cin][LX][LY]
    float* s_w = nca_smem + a.cin * LX * LY;                 // [cout][cin][K][K]
    const int x0 = blockIdx.y * NCA_TX, y0 = blockIdx.x * NCA_TY;
    const int nw = a.cout * a.cin * K * K;
    for (int i = threadIdx.x; i < nw; i += DIE_BLOCK) s_w[i] = a.w[i];
    constexpr int LYV = NCA_TY + 2 * R;
    for (int c = 0; c < a.cin; ++c) {
        for (int i = threadIdx.x; i < LX * LYV; i += DIE_BLOCK) {
            const int li = i / LYV, lj = i - li * LYV;
            // (rows / columns past the last tile's outputs wrap like a circular field whatever the mode: never used)
            const int vx = x0 - R + li, vy = y0 - R + lj;
            const int gx = nca_pad_index(vx < a.W + R ? vx : vx % a.W, a.W, a.pad), gy = nca_pad_index(vy < a.H + R ? vy : vy % a.H, a.H, a.pad);
            s_in[(c * LX + li) * LY + lj] = (gx < 0 || gy < 0) ? 0.f : nca_load(a.in[c], a.kind[c], (int64_t)gx * a.H + gy, a.epoch);
        }
    }
    __syncthreads();
    const int ti = threadIdx.x / (NCA_TY / 4), tj = (threadIdx.x % (NCA_TY / 4)) * 4;     // 16 rows × 16 column quads
    float acc[NCA_MAXC][4];
#pragma unroll
    for (int o = 0; o < NCA_MAXC; ++o) acc[o][0] = acc[o][1] = acc[o][2] = acc[o][3] = 0.f;
    for (int c = 0; c < a.cin; ++c) {
#pragma unroll
        for (int ka = 0; ka < K; ++ka) {
            float row[4 + K - 1];
#pragma unroll
            for (int q = 0; q < 4 + K - 1; ++q) row[q] = s_in[(c * LX + ti + ka) * LY + tj + q];
#pragma unroll
            for (int o = 0; o < NCA_MAXC; ++o) {
                if (o < a.cout) {
#pragma unroll
                    for (int kb = 0; kb < K; ++kb) {
                        const float wv = s_w[((o * a.cin + c) * K + ka) * K + kb];
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[o][q] += wv * row[q + kb];
                    }
                }
            }
        }
    }
    const int gx = x0 + ti, gy = y0 + tj;
    if (gx >= a.W) return;
#pragma unroll
    for (int o = 0; o < NCA_MAXC; ++o) {
        if (o < a.cout) {
            float v[4] = {acc[o][0], acc[o][1], acc[o][2], acc[o][3]};
            if (a.apply_tanh) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = tanhf(v[q]);
            }
            float* dst = a.out[o] + (int64_t)gx * a.H + gy;
            if (gy + 3 < a.H && (a.H & 3) == 0) *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (gy + q < a.H) dst[q] = v[q];
            }
        }
    }
}

extern "C" int die_conv2d_circular(int32_t W, int32_t H, int32_t cin, const die_conv_plane* in, int32_t epoch, int32_t cout,
                                   float* const* out, int32_t k, const float* weights, int32_t apply_tanh, void* stream) {
    return die_conv2d(W, H, cin, in, epoch, cout, out, k, weights, apply_tanh, DIE_PAD_CIRCULAR, stream);
}

extern "C" int die_conv2d(int32_t W, int32_t H, int32_t cin, const die_conv_plane* in, int32_t epoch, int32_t cout,
                          float* const* out, int32_t k, const float* weights, int32_t apply_tanh, int32_t padding_mode, void* stream) {
    DIE_REQUIRE(padding_mode >= DIE_PAD_CIRCULAR && padding_mode <= DIE_PAD_REPLICATE, "die_conv2d: bad padding mode %d", padding_mode);
    if (padding_mode == DIE_PAD_REFLECT && (k / 2 >= W || k / 2 >= H)) {      // (torch refuses it too)
        die_set_error("die_conv2d: 'reflect' padding of %d cells needs a field larger than that (%dx%d)", k / 2, W, H);
        return DIE_ERR_ARG;
    }
    DIE_REQUIRE(W >= 1 && H >= 1, "die_conv2d_circular: bad size %dx%d", W, H);
    DIE_REQUIRE(cin >= 1 && cin <= NCA_MAXC && cout >= 1 && cout <= NCA_MAXC, "die_conv2d_circular: 1..%d channels (got %d -> %d)",
                NCA_MAXC, cin, cout);
    if (!(k == 1 || k == 3 || k == 5 || k == 7)) {
        die_set_error("die_conv2d_circular: kernel size %d (odd sizes up to %d: 'same' padding of an even kernel is asymmetric)", k, NCA_MAXK);
        return DIE_ERR_UNSUPPORTED;
    }
    DIE_REQUIRE(in && out && weights, "die_conv2d_circular: null argument");
    ConvArgs a;
    for (int c = 0; c < NCA_MAXC; ++c) {
        a.in[c] = c < cin ? in[c].data : nullptr;
        a.kind[c] = c < cin ? in[c].kind : 0;
        a.out[c] = c < cout ? out[c] : nullptr;
        DIE_REQUIRE(c >= cin || (in[c].data && in[c].kind >= DIE_PLANE_F32 && in[c].kind <= DIE_PLANE_AGENTS), "die_conv2d_circular: bad input plane %d", c);
        DIE_REQUIRE(c >= cout || out[c], "die_conv2d_circular: null output plane %d", c);
        for (int q = 0; q < cout && c < cin; ++q) DIE_REQUIRE((const void*)out[q] != in[c].data, "die_conv2d_circular: in-place convolution");
    }
    a.w = weights; a.W = W; a.H = H; a.cin = cin; a.cout = cout; a.k = k; a.epoch = epoch; a.apply_tanh = apply_tanh; a.pad = padding_mode;
    const int R = k / 2;
    const size_t lds = ((size_t)cin * (NCA_TX + 2 * R) * (NCA_TY + 2 * R + 1) + (size_t)cout * cin * k * k) * sizeof(float);
    dim3 grid((H + NCA_TY - 1) / NCA_TY, (W + NCA_TX - 1) / NCA_TX);
    hipStream_t s = (hipStream_t)stream;
    switch (k) {
        case 1: k_conv_circular<1><<<grid, DIE_BLOCK, lds, s>>>(a); break;
        case 3: k_conv_circular<3><<<grid, DIE_BLOCK, lds, s>>>(a); break;
        case 5: k_conv_circular<5><<<grid, DIE_BLOCK, lds, s>>>(a); break;
        default: k_conv_circular<7><<<grid, DIE_BLOCK, lds, s>>>(a); break;
    }
    DIE_CHECK_LAUNCH("die_conv2d_circular");
    return DIE_OK;
}

struct GatherArgs {
    die_geo g;
    int64_t N;
    const uint32_t *x, *y;
    const float* plane[3];
    float coef[3];
    float* out[3];
};

__global__ __launch_bounds__(DIE_BLOCK) void k_gather_scale(GatherArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < a.N; n += stride) {
        const int64_t c = die_local(a.g, die_cell((int64_t)a.x[n], a.g.gW), die_cell((int64_t)a.y[n], a.g.gH));
#pragma unroll
        for (int q = 0; q < 3; ++q) a.out[q][n] = a.plane[q][c] * a.coef[q];
    }
}

extern "C" int die_gather_scale(const die_medium* m, const die_agents* ag, const float* const* planes, const float* coefs,
                                const die_action* out, void* stream) {
    DIE_REQUIRE(m && ag && planes && coefs && out, "die_gather_scale: null argument");
    DIE_REQUIRE(ag->N > 0 && out->N == ag->N && ag->x && ag->y && out->dx && out->dy && out->deposit, "die_gather_scale: bad arrays");
    GatherArgs a;
    a.g = die_geo_of(m); a.N = ag->N; a.x = ag->x; a.y = ag->y;
    for (int q = 0; q < 3; ++q) { a.plane[q] = planes[q]; a.coef[q] = coefs[q]; DIE_REQUIRE(planes[q], "die_gather_scale: null plane %d", q); }
    a.out[0] = out->dx; a.out[1] = out->dy; a.out[2] = out->deposit;
    int64_t g = (ag->N + DIE_BLOCK - 1) / DIE_BLOCK;
    k_gather_scale<<<(int)(g < 8192 ? g : 8192), DIE_BLOCK, 0, (hipStream_t)stream>>>(a);
    DIE_CHECK_LAUNCH("die_gather_scale");
    return DIE_OK;
}
