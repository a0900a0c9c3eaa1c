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
    int W, H;
    int64_t N;
    const void* chem;
    const void* food;
    const uint32_t* x;
    const uint32_t* y;
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

// core/utils.py:178-180 for |r| < 3π: into (-π, π]
__device__ __forceinline__ float renorm_rad(float r) {
    if (r > DIE_PI_F) r -= DIE_2PI_F;
    if (r <= -DIE_PI_F) r += DIE_2PI_F;
    return r;
}

template <typename T, int KIND>
__global__ __launch_bounds__(DIE_BLOCK) void k_gradient_forward(FwdArgs a) {
    const T* chem = (const T*)a.chem;
    const T* food = (const T*)a.food;
    const int W = a.W, H = a.H;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < a.N; n += stride) {
        const uint32_t X = a.x[n], Y = a.y[n];
        const float d = a.heading[n];
        float sd, cd;
        sincosf(d, &sd, &cd);
        // probe cell: agents + sense_offset·(cos d, sin d), nearest label, clamped (gradient.py:73-76,105)
        const int px = die_cell((int64_t)X + die_q32(a.sense_offset * cd), W);
        const int py = die_cell((int64_t)Y + die_q32(a.sense_offset * sd), H);
        // np.gradient at the probe cell: central inside, one-sided at the four edges (gradient.py:57)
        const int xm = px > 0 ? px - 1 : 0, xp = px < W - 1 ? px + 1 : W - 1;
        const int ym = py > 0 ? py - 1 : 0, yp = py < H - 1 ? py + 1 : H - 1;
        const float cxm = die_ld(chem, (int64_t)xm * H + py), cxp = die_ld(chem, (int64_t)xp * H + py);
        const float cym = die_ld(chem, (int64_t)px * H + ym), cyp = die_ld(chem, (int64_t)px * H + yp);
        // food under the agent (gradient.py:114-116)
        const int cx = die_cell((int64_t)X, W), cy = die_cell((int64_t)Y, H);
        const float f_own = die_ld(food, (int64_t)cx * H + cy);

        const float gx = (cxp - cxm) * ((xp - xm) == 2 ? 0.5f : 1.0f);
        const float gy = (cyp - cym) * ((yp - ym) == 2 ? 0.5f : 1.0f);
        const float norm = sqrtf(gx * gx + gy * gy);
        float ux = gx, uy = gy;
        if (a.normalized) {                       // g / |g| with 0/0 → 0 (gradient.py:60-63)
            ux = norm > 0.f ? gx / norm : 0.f;
            uy = norm > 0.f ? gy / norm : 0.f;
        }
        if (a.grad_clip >= 0.f && !(norm >= a.grad_clip)) ux = uy = 0.f;   // gradient.py:64-66

        float d_new = d;
        float dep_mask = 1.0f;
        bool heading_from_vector = true;
        if (KIND == DIE_AGENT_PHYSARUM) {
            // _discrete_turn / _choose_turn (gradient.py:168-208)
            const float dr = sqrtf(ux * ux + uy * uy);
            const float drads = atan2f(uy, ux);
            const float delta = renorm_rad(d - drads);
            const float atol = a.turn_rad * a.rtol;
            const bool und_grad = fabsf(drads) <= 1e-8f + 1e-5f * fabsf(drads);
            const bool und_turn = fabsf(delta) <= atol + 1e-2f * fabsf(delta);
            const bool unseen = fabsf(delta) > a.sense_rad;
            const bool und = und_grad || und_turn || unseen;
            float sgn;
            if (und) {
                if (a.turn_sign) sgn = (float)a.turn_sign[n];
                else sgn = (die_draw(a.seed, a.step, (uint64_t)n, DIE_STREAM_TURN).v[0] & 1u) ? 1.f : -1.f;
            } else {
                sgn = delta > atol ? -1.f : 1.f;  // right (clockwise) / left
            }
            const float d2 = renorm_rad(d + sgn * a.turn_rad);
            float s2, c2;
            sincosf(d2, &s2, &c2);
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
                const die_u32x4 r = die_draw(a.seed, a.step, (uint64_t)n, DIE_STREAM_NOISE);
                const float u1 = ((float)r.v[0] + 1.0f) * 2.3283064365386963e-10f;
                const float u2 = (float)r.v[1] * 2.3283064365386963e-10f;
                const float rad = 0.4f * sqrtf(-2.0f * logf(u1));
                float sn, cn;
                sincosf(DIE_2PI_F * u2, &sn, &cn);
                nx = rad * cn;
                ny = rad * sn;
            }
            const float ox = a.pgx ? a.pgx[n] : 0.f, oy = a.pgy ? a.pgy[n] : 0.f;
            ux = (1.f - a.inertia) * ux + a.inertia * ox + a.noise_scale * nx;
            uy = (1.f - a.inertia) * uy + a.inertia * oy + a.noise_scale * ny;
            heading_from_vector = true;
        }
        if (a.pgx) { a.pgx[n] = ux; a.pgy[n] = uy; }
        if (heading_from_vector) d_new = atan2f(uy, ux);          // get_radians (gradient.py:110)
        a.heading[n] = d_new;
        a.dx[n] = ux * a.scale;
        a.dy[n] = uy * a.scale;
        a.dep[n] = a.deposit * f_own * dep_mask;
    }
}

__global__ __launch_bounds__(DIE_BLOCK) void k_brownian_forward(int64_t N, const uint8_t* alive, double s, double dep,
                                                                 uint64_t seed, uint32_t step, float* dx, float* dy,
                                                                 float* dp) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        const die_u32x4 r = die_draw(seed, step, (uint64_t)n, DIE_STREAM_BROWNIAN);
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
    k.W = m->W; k.H = m->H; k.N = a->N;
    k.chem = m->chem; k.food = m->food; k.x = a->x; k.y = a->y;
    k.heading = g->heading; k.pgx = g->prev_gx; k.pgy = g->prev_gy; k.turn_sign = g->turn_sign;
    k.dx = out->dx; k.dy = out->dy; k.dep = out->deposit;
    k.scale = g->scale; k.deposit = g->deposit; k.inertia = g->inertia; k.sense_offset = g->sense_offset;
    k.noise_scale = g->noise_scale; k.grad_clip = g->grad_clip; k.turn_rad = g->turn_radians;
    k.sense_rad = g->sense_radians; k.rtol = g->turn_tolerance; k.normalized = g->normalized_grad;
    k.seed = g->seed; k.step = g->step;
    hipStream_t s = (hipStream_t)stream;
    const int grid = agent_grid(a->N);
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
        a->N, a->alive, (double)move_scale, (double)deposit_scale, seed, step, out->dx, out->dy, out->deposit);
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
