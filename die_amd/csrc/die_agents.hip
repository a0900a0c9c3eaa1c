// Agent.forward kernels (gfx950): GradientAgent / PhysarumAgent / BrownianAgent / ConstAgent.
// One thread per agent slot; every slot acts, alive or not, as in the reference
// (core/agent/gradient.py:96-124 has no alive masking).
//
// Memory behaviour: per slot 12 B of state in (x, y, heading), 4 B out (heading), 12 B out
// (action), five gathers (4 chem taps around the probe cell + food at the own cell).  No LDS:
// there is no reuse between slots that the L2 does not already provide.
#include "die_common.h"
#include "die_rng.h"

struct FwdArgs {
    die_geo g;
    int64_t N;
    const void* chem;
    const void* food;
    const uint32_t* x;
    const uint32_t* y;
    const uint32_t* slot;
    float* heading;
    float* pgx;
    float* pgy;
    const int8_t* turn_sign;
    float* dx;
    float* dy;
    float* dep;
    float scale, deposit, inertia, sense_offset, noise_scale, grad_clip, turn_rad, sense_rad, rtol;
    int normalized;
    uint64_t seed;
    uint32_t step;
};

#define DIE_PI_F 3.14159265358979323846f
#define DIE_2PI_F 6.28318530717958647692f

// sin/cos for |x| ≤ ~2π (headings live in (−π, π]): Cody–Waite reduction by π/2 and the
// cephes single-precision minimax polynomials; ≤ 1.5 ulp, branch-free, ~30 VALU — the
// library sincosf carries a large-argument path this kernel can never take.
__device__ __forceinline__ void die_sincos(float x, float* s, float* c) {
    const float k = rintf(x * 0.636619772367581343f);        // x / (π/2)
    const int q = (int)k;
    float r = fmaf(k, -1.5703125f, x);                        // π/2 split in three parts
    r = fmaf(k, -4.837512969970703125e-4f, r);
    r = fmaf(k, -7.549789948768648e-8f, r);
    const float z = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    ps = fmaf(ps * z, r, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    pc = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    const float ss = (q & 1) ? pc : ps;
    const float cc = (q & 1) ? ps : pc;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}

// core/utils.py:178-180 for |r| < 3π: into (-π, π]
__device__ __forceinline__ float renorm_rad(float r) {
    if (r > DIE_PI_F) r -= DIE_2PI_F;
    if (r <= -DIE_PI_F) r += DIE_2PI_F;
    return r;
}

// Agents per thread per loop trip: the U slots' state loads, then their 5U gathers, are issued
// back to back before any use, so each wave keeps U× the memory requests in flight (the kernel
// is bound by gather latency/throughput, not by VALU or bytes — profiles/README.md).
#ifndef DIE_FWD_UNROLL
#define DIE_FWD_UNROLL 1
#endif

template <typename T, int KIND>
__global__ __launch_bounds__(DIE_BLOCK) void k_gradient_forward(FwdArgs a) {
    constexpr int U = DIE_FWD_UNROLL;
    const T* chem = (const T*)a.chem;
    const T* food = (const T*)a.food;
    const die_geo g = a.g;
    const int W = g.gW, H = g.gH;           // world size: probes clamp at the world's edge
    const int64_t chunk = (int64_t)DIE_BLOCK * U;
    for (int64_t base = (int64_t)blockIdx.x * chunk; base < a.N; base += (int64_t)gridDim.x * chunk) {
        int64_t n[U];
        bool live[U];
        uint32_t X[U], Y[U], sid[U];
        float d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            n[u] = base + (int64_t)u * DIE_BLOCK + threadIdx.x;
            live[u] = n[u] < a.N;
            const int64_t m = live[u] ? n[u] : 0;
            X[u] = a.x[m]; Y[u] = a.y[m]; d[u] = a.heading[m];
            sid[u] = a.slot ? a.slot[m] : (uint32_t)m;          // reference slot id: keys the random streams
        }
        float cxm[U], cxp[U], cym[U], cyp[U], f_own[U], wx[U], wy[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float sd, cd;
            die_sincos(d[u], &sd, &cd);
            // probe cell: agents + sense_offset·(cos d, sin d), nearest label, clamped (gradient.py:73-76,105)
            const int px = die_cell((int64_t)X[u] + die_q32(a.sense_offset * cd), W);
            const int py = die_cell((int64_t)Y[u] + die_q32(a.sense_offset * sd), H);
            // np.gradient at the probe cell: central inside, one-sided at the four edges (gradient.py:57)
            const int xm = px > 0 ? px - 1 : 0, xp = px < W - 1 ? px + 1 : W - 1;
            const int ym = py > 0 ? py - 1 : 0, yp = py < H - 1 ? py + 1 : H - 1;
#ifdef DIE_ABL_NOGATHER
            cxm[u] = (float)xm; cxp[u] = (float)xp * 1.5f; cym[u] = (float)ym; cyp[u] = (float)(yp + py);
#else
            cxm[u] = die_ld(chem, die_local(g, xm, py)); cxp[u] = die_ld(chem, die_local(g, xp, py));
            cym[u] = die_ld(chem, die_local(g, px, ym)); cyp[u] = die_ld(chem, die_local(g, px, yp));
#endif
            wx[u] = (xp - xm) == 2 ? 0.5f : 1.0f;
            wy[u] = (yp - ym) == 2 ? 0.5f : 1.0f;
            // food under the agent (gradient.py:114-116)
            const int cx = die_cell((int64_t)X[u], W), cy = die_cell((int64_t)Y[u], H);
#ifdef DIE_ABL_NOFOOD
            f_own[u] = (float)(cx + cy);
#else
            f_own[u] = die_ld(food, die_local(g, cx, cy));
#endif
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float gx = (cxp[u] - cxm[u]) * wx[u];
            const float gy = (cyp[u] - cym[u]) * wy[u];
            const float norm = sqrtf(gx * gx + gy * gy);
            float ux = gx, uy = gy;
            if (a.normalized) {                       // g / |g| with 0/0 → 0 (gradient.py:60-63)
                ux = norm > 0.f ? gx / norm : 0.f;
                uy = norm > 0.f ? gy / norm : 0.f;
            }
            if (a.grad_clip >= 0.f && !(norm >= a.grad_clip)) ux = uy = 0.f;   // gradient.py:64-66

            float d_new = d[u];
            float dep_mask = 1.0f;
            bool heading_from_vector = true;
            if (KIND == DIE_AGENT_PHYSARUM) {
                // _discrete_turn / _choose_turn (gradient.py:168-208)
                const float dr = sqrtf(ux * ux + uy * uy);
#ifdef DIE_ABL_NOMATH
                const float drads = uy * 0.5f + ux;
#else
                const float drads = atan2f(uy, ux);
#endif
                const float delta = renorm_rad(d[u] - drads);
                const float atol = a.turn_rad * a.rtol;
                const bool und_grad = fabsf(drads) <= 1e-8f + 1e-5f * fabsf(drads);
                const bool und_turn = fabsf(delta) <= atol + 1e-2f * fabsf(delta);
                const bool unseen = fabsf(delta) > a.sense_rad;
                const bool und = und_grad || und_turn || unseen;
                float sgn;
                if (und) {
                    if (a.turn_sign) sgn = live[u] ? (float)a.turn_sign[sid[u]] : 1.f;
                    else sgn = (die_draw(a.seed, a.step, (uint64_t)sid[u], DIE_STREAM_TURN).v[0] & 1u) ? 1.f : -1.f;
                } else {
                    sgn = delta > atol ? -1.f : 1.f;  // right (clockwise) / left
                }
                const float d2 = renorm_rad(d[u] + sgn * a.turn_rad);
                float s2, c2;
                die_sincos(d2, &s2, &c2);
                const float r = a.normalized ? 1.f : dr;
                ux = r * c2;
                uy = r * s2;
                dep_mask = (und_grad || und_turn) ? 0.1f : 1.0f;   // clip(mask, .1, 1) (gradient.py:210-214)
                d_new = d2;
                heading_from_vector = !a.normalized;               // |g| may be 0 there: angle(0) = 0
            }
            // _process_momentum (gradient.py:82-91)
            if (a.inertia != 0.f || a.noise_scale != 0.f) {
                float nx = 0.f, ny = 0.f;
                if (a.noise_scale != 0.f) {
                    const die_u32x4 r = die_draw(a.seed, a.step, (uint64_t)sid[u], DIE_STREAM_NOISE);
                    const float u1 = ((float)r.v[0] + 1.0f) * 2.3283064365386963e-10f;
                    const float u2 = (float)r.v[1] * 2.3283064365386963e-10f;
                    const float rad = 0.4f * sqrtf(-2.0f * logf(u1));
                    float sn, cn;
                    sincosf(DIE_2PI_F * u2, &sn, &cn);
                    nx = rad * cn;
                    ny = rad * sn;
                }
                const float ox = (a.pgx && live[u]) ? a.pgx[n[u]] : 0.f, oy = (a.pgy && live[u]) ? a.pgy[n[u]] : 0.f;
                ux = (1.f - a.inertia) * ux + a.inertia * ox + a.noise_scale * nx;
                uy = (1.f - a.inertia) * uy + a.inertia * oy + a.noise_scale * ny;
                heading_from_vector = true;
            }
            if (heading_from_vector) d_new = atan2f(uy, ux);          // get_radians (gradient.py:110)
#ifdef DIE_ABL_NOSTORE
            if (live[u] && ux == 123.456f) {
#else
            if (live[u]) {
#endif
                if (a.pgx) { a.pgx[n[u]] = ux; a.pgy[n[u]] = uy; }
                a.heading[n[u]] = d_new;
                a.dx[n[u]] = ux * a.scale;
                a.dy[n[u]] = uy * a.scale;
                a.dep[n[u]] = a.deposit * f_own[u] * dep_mask;
            }
        }
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
    const int64_t chunk = (int64_t)DIE_BLOCK * DIE_FWD_UNROLL;
    int64_t g = (N + chunk - 1) / chunk;
#ifndef DIE_FWD_GRID_CAP
#define DIE_FWD_GRID_CAP (256 * 16)
#endif
    const int64_t cap = DIE_FWD_GRID_CAP;
    return (int)(g < cap ? (g > 0 ? g : 1) : cap);
}

extern "C" int die_gradient_forward(const die_medium* m, const die_agents* a, die_gradient_agent* g, die_action* out,
                                    void* stream) {
    DIE_REQUIRE(m && a && g && out, "die_gradient_forward: null argument");
    DIE_REQUIRE(m->W >= 2 && m->H >= 2, "die_gradient_forward: field must be at least 2x2 (got %dx%d)", m->W, m->H);
    DIE_REQUIRE(a->N > 0 && out->N == a->N, "die_gradient_forward: action has %lld slots, agents %lld",
                (long long)out->N, (long long)a->N);
    DIE_REQUIRE(m->chem && m->food && a->x && a->y && g->heading && out->dx && out->dy && out->deposit,
                "die_gradient_forward: null device pointer");
    DIE_REQUIRE(m->dtype == DIE_F32 || m->dtype == DIE_F16, "die_gradient_forward: bad field dtype %d", m->dtype);
    DIE_REQUIRE(g->kind == DIE_AGENT_GRADIENT || g->kind == DIE_AGENT_PHYSARUM, "die_gradient_forward: bad kind %d",
                g->kind);
    DIE_REQUIRE(g->inertia == 0.f || (g->prev_gx && g->prev_gy), "die_gradient_forward: inertia needs prev_gx/prev_gy");
    FwdArgs k;
    k.g = die_geo_of(m); k.N = a->N;
    k.chem = m->chem; k.food = m->food; k.x = a->x; k.y = a->y; k.slot = a->slot;
    k.heading = g->heading; k.pgx = g->prev_gx; k.pgy = g->prev_gy; k.turn_sign = g->turn_sign;
    k.dx = out->dx; k.dy = out->dy; k.dep = out->deposit;
    k.scale = g->scale; k.deposit = g->deposit; k.inertia = g->inertia; k.sense_offset = g->sense_offset;
    k.noise_scale = g->noise_scale; k.grad_clip = g->grad_clip; k.turn_rad = g->turn_radians;
    k.sense_rad = g->sense_radians; k.rtol = g->turn_tolerance; k.normalized = g->normalized_grad;
    k.seed = g->seed; k.step = g->step;
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
