// EnvRenderer frames on the device (core/render.py:76-110): one sweep over the medium produces
//   the medium image   (W, H, 3) float32 = (agents, env_food, chem1) as R, G, B           (:92-101, 'rgb' colours)
//   the trace image    (W, H, 4) float32 = colormap(trace), trace ← trace·decay + agents   (:9-30, :103-110)
//   optionally         (W, H, 3) uint8   = clip(medium image, 0, 1)·255 — what a plot loop needs to download.
// The colormap is matplotlib's lookup table handed in by the caller (N colours + under, over, bad rows); the index
// arithmetic follows matplotlib.colors.Colormap.__call__ for float input.
#include "die_common.h"

struct RenderArgs {
    const unsigned long long* owner;
    const void* food;
    const void* chem;
    int64_t cells;
    int epoch;
    float* trace;          // W*H state, updated in place
    float decay;
    const float* lut;      // (lut_n + 3, 4)
    int lut_n;
    float* rgb;            // W*H*3 or NULL
    float* rgba;           // W*H*4 or NULL
    uint8_t* rgb8;         // W*H*3 or NULL
};

__device__ __forceinline__ uint8_t to_u8(float v) {
    v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
    return (uint8_t)(v * 255.f + 0.5f);
}

template <typename T>
__global__ __launch_bounds__(DIE_BLOCK) void k_render(RenderArgs a) {
    const T* food = (const T*)a.food;
    const T* chem = (const T*)a.chem;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < a.cells; c += stride) {
        const float occ = die_claim_occupied(a.owner[c], a.epoch) ? 1.f : 0.f;
        const float f = die_ld(food, c), ch = die_ld(chem, c);
        if (a.rgb) { a.rgb[3 * c] = occ; a.rgb[3 * c + 1] = f; a.rgb[3 * c + 2] = ch; }
        if (a.rgb8) { a.rgb8[3 * c] = to_u8(occ); a.rgb8[3 * c + 1] = to_u8(f); a.rgb8[3 * c + 2] = to_u8(ch); }
        if (a.trace) {
            const float t = a.trace[c] * a.decay + occ;
            a.trace[c] = t;
            if (a.rgba) {
                double xa = (double)t * (double)a.lut_n;                  // xa *= N
                int idx;
                if (xa != xa) idx = a.lut_n + 2;                           // bad
                else {
                    if (xa == (double)a.lut_n) xa = (double)(a.lut_n - 1); // 1.0 maps to the last colour
                    idx = xa < 0.0 ? a.lut_n : (xa >= (double)a.lut_n ? a.lut_n + 1 : (int)xa);
                }
                const float4 col = *(const float4*)(a.lut + 4 * idx);
                *(float4*)(a.rgba + 4 * c) = col;
            }
        }
    }
}

extern "C" int die_render_frames(const die_medium* m, float* trace, float trace_decay, const float* lut, int32_t lut_n,
                                 float* rgb_out, float* rgba_out, uint8_t* rgb8_out, void* stream) {
    DIE_REQUIRE(m && m->owner && m->food && m->chem && m->W >= 1 && m->H >= 1, "die_render_frames: bad medium");
    DIE_REQUIRE(m->dtype == DIE_F32 || m->dtype == DIE_F16, "die_render_frames: bad field dtype %d", m->dtype);
    DIE_REQUIRE(m->epoch >= 1 && m->epoch <= DIE_OWNER_EPOCH_MAX, "die_render_frames: bad epoch %d", m->epoch);
    DIE_REQUIRE(!rgba_out || (trace && lut && lut_n >= 1), "die_render_frames: the trace image needs the trace state and a lookup table");
    RenderArgs a;
    a.owner = (const unsigned long long*)m->owner; a.food = m->food; a.chem = m->chem; a.cells = (int64_t)m->W * m->H;
    a.epoch = m->epoch; a.trace = trace; a.decay = trace_decay; a.lut = lut; a.lut_n = lut_n;
    a.rgb = rgb_out; a.rgba = rgba_out; a.rgb8 = rgb8_out;
    int64_t g = (a.cells + DIE_BLOCK - 1) / DIE_BLOCK;
    const int grid = (int)(g < 8192 ? g : 8192);
    if (m->dtype == DIE_F32) k_render<float><<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>(a);
    else k_render<__half><<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>(a);
    DIE_CHECK_LAUNCH("die_render_frames");
    return DIE_OK;
}

// ---- Env._get_sense_mask (core/env.py:276-290) ---------------------------------------------------------------
// mask = ceil(round(gaussian(agents, σ), decimals)): which cells lie in the blurred neighbourhood of an agent.
// skimage.filters.gaussian defaults: mode 'nearest' (indices clamp at the edges), truncate 4 → radius int(4σ+.5);
// scipy filters axis 0 first, then axis 1, in float64.  Two plain passes with float64 accumulation (the mask is a
// threshold: fp32 sums could flip cells on it); the option is off by default and not on the benchmark path.
#define SM_MAXR 16
struct SenseArgs {
    const unsigned long long* owner;
    int W, H, epoch, R;
    double w[2 * SM_MAXR + 1];
    double* tmp;
    uint8_t* mask;
    double scale;          // 10^decimals
};

__global__ __launch_bounds__(DIE_BLOCK) void k_sense_x(SenseArgs a) {
    const int64_t total = (int64_t)a.W * a.H, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += stride) {
        const int i = (int)(c / a.H), j = (int)(c - (int64_t)i * a.H);
        double t = 0.0;
        for (int k = -a.R; k <= a.R; ++k) {
            const int r = min(max(i + k, 0), a.W - 1);
            t += a.w[k + a.R] * (die_claim_occupied(a.owner[(int64_t)r * a.H + j], a.epoch) ? 1.0 : 0.0);
        }
        a.tmp[c] = t;
    }
}

__global__ __launch_bounds__(DIE_BLOCK) void k_sense_y(SenseArgs a) {
    const int64_t total = (int64_t)a.W * a.H, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += stride) {
        const int i = (int)(c / a.H), j = (int)(c - (int64_t)i * a.H);
        double t = 0.0;
        for (int k = -a.R; k <= a.R; ++k) {
            const int q = min(max(j + k, 0), a.H - 1);
            t += a.w[k + a.R] * a.tmp[(int64_t)i * a.H + q];
        }
        a.mask[c] = rint(t * a.scale) >= 1.0 ? 1 : 0;        // np.round (half to even), then ceil of a value in [0, 1]
    }
}

extern "C" int die_sense_mask(const die_medium* m, float sigma, int32_t decimals, uint8_t* mask_out, double* tmp, void* stream) {
    DIE_REQUIRE(m && m->owner && mask_out && tmp && m->W >= 1 && m->H >= 1, "die_sense_mask: null argument");
    DIE_REQUIRE(m->gW <= 0, "die_sense_mask: periodic single-tile planes only");
    DIE_REQUIRE(m->epoch >= 1 && m->epoch <= DIE_OWNER_EPOCH_MAX, "die_sense_mask: bad epoch %d", m->epoch);
    DIE_REQUIRE(sigma > 0.f && decimals >= 0 && decimals <= 9, "die_sense_mask: bad sigma / decimals");
    SenseArgs a;
    a.R = (int)(4.0 * (double)sigma + 0.5);
    DIE_REQUIRE(a.R >= 1 && a.R <= SM_MAXR, "die_sense_mask: sigma %g needs radius %d (max %d)", (double)sigma, a.R, SM_MAXR);
    double sum = 0.0;
    for (int k = -a.R; k <= a.R; ++k) { a.w[k + a.R] = exp(-0.5 / ((double)sigma * (double)sigma) * k * k); sum += a.w[k + a.R]; }
    for (int k = 0; k <= 2 * a.R; ++k) a.w[k] /= sum;
    a.owner = (const unsigned long long*)m->owner; a.W = m->W; a.H = m->H; a.epoch = m->epoch; a.tmp = tmp; a.mask = mask_out;
    a.scale = pow(10.0, (double)decimals);
    int64_t g = ((int64_t)m->W * m->H + DIE_BLOCK - 1) / DIE_BLOCK;
    const int grid = (int)(g < 8192 ? g : 8192);
    k_sense_x<<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>(a);
    k_sense_y<<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>(a);
    DIE_CHECK_LAUNCH("die_sense_mask");
    return DIE_OK;
}
