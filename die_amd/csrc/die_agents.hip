// Agent.forward kernels (gfx950): GradientAgent / PhysarumAgent / BrownianAgent / ConstAgent.
// One thread per agent slot; every slot acts, alive or not, as in the reference
// (core/agent/gradient.py:96-124 has no alive masking).
//
// Memory behaviour: per slot 12 B of state in (x, y, heading), 4 B out (heading), 12 B out
// (action), five gathers (4 chem taps around the probe cell + food at the own cell).  No LDS:
// there is no reuse between slots that the L2 does not already provide.
#include "die_forward.h"
#include <math.h>
#include <string.h>

template <typename T, int KIND>
__global__ __launch_bounds__(DIE_BLOCK) void k_gradient_forward(FwdArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < a.N; n += stride) {
        const uint32_t sid = a.slot ? a.slot[n] : (uint32_t)n;     // reference slot id: keys the random streams
        const FwdOut o = die_forward_agent<T, KIND>(a, a.x[n], a.y[n], die_heading_ld(a.heading_hi, a.heading_lo, n), sid, n);
        die_heading_st(a.heading_hi, a.heading_lo, n, o.heading);
        a.dx[n] = o.dx;
        a.dy[n] = o.dy;
        a.dep[n] = o.dep;
    }
}

__global__ __launch_bounds__(DIE_BLOCK) void k_brownian_forward(int64_t N, const uint8_t* alive, const uint32_t* slot, double s, double dep,
                                                                 uint64_t seed, uint32_t step, float* dx, float* dy,
                                                                 float* dp) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        const die_u32x4 r = die_draw(seed, step, (uint64_t)(slot ? slot[n] : (uint32_t)n), DIE_STREAM_BROWNIAN);
        const double m = alive[n] ? 1.0 : 0.0;
        // (b-a)*u.round(3)+a, × alive (core/data_init.py:168-169,248-253)
        dx[n] = (float)((2.0 * s * (die_round3_units(r.v[0]) / 1000.0) - s) * m);
        dy[n] = (float)((2.0 * s * (die_round3_units(r.v[1]) / 1000.0) - s) * m);
        dp[n] = (float)((dep * (die_round3_units(r.v[2]) / 1000.0)) * m);
    }
}

__global__ __launch_bounds__(DIE_BLOCK) void k_const_forward(int64_t N, float vx, float vy, float vd, float* dx, float* dy,
                                                             float* dp) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        dx[n] = vx;
        dy[n] = vy;
        dp[n] = vd;
    }
}

static int agent_grid(int64_t N) {
    int64_t g = (N + DIE_BLOCK - 1) / DIE_BLOCK;
    const int64_t cap = 256 * 32;   // 256 CUs × 8 blocks of 4 waves, ×4 rounds: enough to fill and balance
    return (int)(g < cap ? (g > 0 ? g : 1) : cap);
}

static int fwd_grid(int64_t N) {
    int64_t g = (N + DIE_BLOCK - 1) / DIE_BLOCK;
#ifndef DIE_FWD_GRID_CAP
#define DIE_FWD_GRID_CAP (256 * 16)
#endif
    const int64_t cap = DIE_FWD_GRID_CAP;
    return (int)(g < cap ? (g > 0 ? g : 1) : cap);
}

// Largest x >= 0 that np.isclose(0, x, rtol, atol) accepts: |x| <= atol + rtol·|x| as numpy evaluates it (product rounded, then
// the sum; core/agent/gradient.py:177-178).  The accepted set is an interval [0, X] (for rtol < 1 the right-hand side grows
// a hundred times slower than x), so X is found by bisection over the bit patterns of the non-negative doubles — the device
// then decides with ONE comparison per test, and the tie at the threshold falls exactly where numpy puts it.
static double isclose_bound(double atol, double rtol) {
    auto ok = [&](double x) { volatile double m = rtol * x; volatile double s = atol + m; return x <= s; };
    if (!ok(0.0)) return -1.0;
    if (!(rtol < 1.0)) return HUGE_VAL;
    uint64_t lo = 0, hi;
    double top = 2.0 * atol / (1.0 - rtol) + 1e-300;
    if (ok(top)) return HUGE_VAL;
    memcpy(&hi, &top, 8);
    while (hi - lo > 1) {                       // ok(lo), !ok(hi)
        const uint64_t mid = lo + (hi - lo) / 2;
        double x;
        memcpy(&x, &mid, 8);
        if (ok(x)) lo = mid; else hi = mid;
    }
    double x;
    memcpy(&x, &lo, 8);
    return x;
}

int die_fill_fwd_args(FwdArgs& k, const die_medium* m, const die_agents* a, const die_gradient_agent* g,
                      const die_action* out, const char* who) {
    DIE_REQUIRE(m && a && g, "%s: null argument", who);
    const die_geo geo = die_geo_of(m);
    DIE_REQUIRE(geo.gW >= 2 && geo.gH >= 2, "%s: field must be at least 2x2 (got %dx%d)", who, geo.gW, geo.gH);
    DIE_REQUIRE(a->N > 0, "%s: no agent slots", who);
    DIE_REQUIRE(!out || (out->N == a->N && out->dx && out->dy && out->deposit), "%s: action has %lld slots, agents %lld",
                who, (long long)(out ? out->N : 0), (long long)a->N);
    DIE_REQUIRE(m->chem && m->food && a->x && a->y && g->heading_hi && g->heading_lo, "%s: null device pointer", who);
    DIE_REQUIRE(m->dtype == DIE_F32 || m->dtype == DIE_F16, "%s: bad field dtype %d", who, m->dtype);
    DIE_REQUIRE(g->kind == DIE_AGENT_GRADIENT || g->kind == DIE_AGENT_PHYSARUM, "%s: bad kind %d", who, g->kind);
    DIE_REQUIRE(g->inertia == 0.f || (g->prev_gx && g->prev_gy), "%s: inertia needs prev_gx/prev_gy", who);
    k.g = geo; k.N = a->N;
    k.chem = m->chem; k.food = m->food; k.mask = m->sense_mask; k.x = a->x; k.y = a->y; k.slot = a->slot;
    k.heading_hi = g->heading_hi; k.heading_lo = g->heading_lo; k.pgx = g->prev_gx; k.pgy = g->prev_gy; k.turn_sign = g->turn_sign;
    k.dx = out ? out->dx : nullptr; k.dy = out ? out->dy : nullptr; k.dep = out ? out->deposit : nullptr;
    k.scale = g->scale; k.deposit = g->deposit; k.inertia = g->inertia; k.sense_offset = g->sense_offset;
    k.noise_scale = g->noise_scale; k.grad_clip = g->grad_clip; k.turn_rad = g->turn_radians;
    k.sense_rad = g->sense_radians; k.rtol = g->turn_tolerance; k.normalized = g->normalized_grad;
    k.seed = g->seed; k.step = g->step; k.step_base = g->step_base;
    k.atol = k.turn_rad * k.rtol;
    k.x_turn = isclose_bound(k.atol, 1e-2);
    k.x_grad = isclose_bound(1e-8, 1e-5);
    const double pi = 3.141592653589793;
    k.c_turn = k.x_turn < 0.0 ? 2.f : (k.x_turn >= pi ? -2.f : (float)cos(k.x_turn));
    k.c_sense = k.sense_rad < 0.0 ? 2.f : (k.sense_rad >= pi ? -2.f : (float)cos(k.sense_rad));
    k.t_grad = (float)tan(k.x_grad);
    k.turn_bits = nullptr;
    return DIE_OK;
}

extern "C" int die_gradient_forward(const die_medium* m, const die_agents* a, die_gradient_agent* g, die_action* out,
                                    void* stream) {
    DIE_REQUIRE(out, "die_gradient_forward: null action");
    FwdArgs k;
    int rc = die_fill_fwd_args(k, m, a, g, out, "die_gradient_forward");
    if (rc != DIE_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int grid = fwd_grid(a->N);
    if (m->dtype == DIE_F32) {
        if (g->kind == DIE_AGENT_PHYSARUM) k_gradient_forward<float, DIE_AGENT_PHYSARUM><<<grid, DIE_BLOCK, 0, s>>>(k);
        else k_gradient_forward<float, DIE_AGENT_GRADIENT><<<grid, DIE_BLOCK, 0, s>>>(k);
    } else {
        if (g->kind == DIE_AGENT_PHYSARUM) k_gradient_forward<__half, DIE_AGENT_PHYSARUM><<<grid, DIE_BLOCK, 0, s>>>(k);
        else k_gradient_forward<__half, DIE_AGENT_GRADIENT><<<grid, DIE_BLOCK, 0, s>>>(k);
    }
    DIE_CHECK_LAUNCH("die_gradient_forward");
    return DIE_OK;
}

extern "C" int die_brownian_forward(const die_agents* a, float move_scale, float deposit_scale, uint64_t seed,
                                    uint32_t step, die_action* out, void* stream) {
    DIE_REQUIRE(a && out, "die_brownian_forward: null argument");
    DIE_REQUIRE(a->N > 0 && out->N == a->N, "die_brownian_forward: action has %lld slots, agents %lld",
                (long long)out->N, (long long)a->N);
    DIE_REQUIRE(a->alive && out->dx && out->dy && out->deposit, "die_brownian_forward: null device pointer");
    k_brownian_forward<<<agent_grid(a->N), DIE_BLOCK, 0, (hipStream_t)stream>>>(
        a->N, a->alive, a->slot, (double)move_scale, (double)deposit_scale, seed, step, out->dx, out->dy, out->deposit);
    DIE_CHECK_LAUNCH("die_brownian_forward");
    return DIE_OK;
}

extern "C" int die_const_forward(int64_t N, float dx, float dy, float deposit, die_action* out, void* stream) {
    DIE_REQUIRE(out && N > 0 && out->N == N, "die_const_forward: bad action");
    DIE_REQUIRE(out->dx && out->dy && out->deposit, "die_const_forward: null device pointer");
    k_const_forward<<<agent_grid(N), DIE_BLOCK, 0, (hipStream_t)stream>>>(N, dx, dy, deposit, out->dx, out->dy,
                                                                            out->deposit);
    DIE_CHECK_LAUNCH("die_const_forward");
    return DIE_OK;
}


// GradientAgent.render (core/agent/gradient.py:126-135): 0.5·(stack(gx, gy, 0) + 1) of the gradient FIELD of
// _get_gradient (:55-71) — np.gradient (central, one-sided at the four edges), g / |g| with 0/0 → 0, components of cells
// whose norm is below grad_clip zeroed.  forward() never builds that field (it reads 4 taps per agent); this kernel
// makes the image when somebody asks for it.
template <typename T>
__global__ __launch_bounds__(DIE_BLOCK) void k_gradient_rgb(const T* chem, int W, int H, int normalized, float grad_clip, float* rgb) {
    const int64_t C = (int64_t)W * H, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < C; c += stride) {
        const int ix = (int)(c / H), iy = (int)(c - (int64_t)ix * H);
        const int xm = ix > 0 ? ix - 1 : 0, xp = ix < W - 1 ? ix + 1 : W - 1, ym = iy > 0 ? iy - 1 : 0, yp = iy < H - 1 ? iy + 1 : H - 1;
        const float gx = xp > xm ? (die_ld(chem, (int64_t)xp * H + iy) - die_ld(chem, (int64_t)xm * H + iy)) * ((xp - xm) == 2 ? 0.5f : 1.0f) : 0.f;
        const float gy = yp > ym ? (die_ld(chem, (int64_t)ix * H + yp) - die_ld(chem, (int64_t)ix * H + ym)) * ((yp - ym) == 2 ? 0.5f : 1.0f) : 0.f;
        const float norm = sqrtf(gx * gx + gy * gy);
        float ux = gx, uy = gy;
        if (normalized) { ux = norm > 0.f ? gx / norm : 0.f; uy = norm > 0.f ? gy / norm : 0.f; }
        if (grad_clip >= 0.f && !(norm >= grad_clip)) { ux = 0.f; uy = 0.f; }
        rgb[c * 3 + 0] = 0.5f * (ux + 1.f);
        rgb[c * 3 + 1] = 0.5f * (uy + 1.f);
        rgb[c * 3 + 2] = 0.5f;
    }
}

extern "C" int die_gradient_render(const void* chem, int32_t W, int32_t H, int32_t dtype, int32_t normalized, float grad_clip,
                                   float* rgb_out, void* stream) {
    DIE_REQUIRE(chem && rgb_out && W >= 2 && H >= 2, "die_gradient_render: bad arguments");
    DIE_REQUIRE(dtype == DIE_F32 || dtype == DIE_F16, "die_gradient_render: bad dtype %d", dtype);
    const int grid = agent_grid((int64_t)W * H);
    if (dtype == DIE_F32) k_gradient_rgb<float><<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>((const float*)chem, W, H, normalized, grad_clip, rgb_out);
    else k_gradient_rgb<__half><<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>((const __half*)chem, W, H, normalized, grad_clip, rgb_out);
    DIE_CHECK_LAUNCH("die_gradient_render");
    return DIE_OK;
}
