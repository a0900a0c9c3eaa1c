// GradientAgent / PhysarumAgent forward() for ONE agent slot (core/agent/gradient.py:96-124,
// 168-219), shared by k_gradient_forward (die_agents.hip) and the fused k_forward_move_claim
// (die_env.hip).
#pragma once
#include "die_common.h"
#include "die_rng.h"

struct FwdArgs {
    die_geo g;
    int64_t N;
    const void* chem;
    const void* food;
    const uint8_t* mask;       // Dynamics.apply_sense_mask: cells with 0 read as 0 (core/env.py:276-295); NULL = all visible
    const uint32_t* x;
    const uint32_t* y;
    const uint32_t* slot;
    uint32_t* heading_hi;      // _direction_rads, float64 in two 4-byte halves (die_gradient_agent)
    uint32_t* heading_lo;
    float* pgx;
    float* pgy;
    const int8_t* turn_sign;
    float* dx;
    float* dy;
    float* dep;
    float scale, deposit, inertia, sense_offset, noise_scale, grad_clip;
    double turn_rad, sense_rad, rtol;
    int normalized;
    uint64_t seed;
    uint32_t step;
    const uint32_t* step_base;   // device word added to step (graph replay), or NULL
    // The decisions of _choose_turn (gradient.py:168-192) as comparisons with constants (die_fill_fwd_args):
    double x_turn, x_grad;       // largest |x| that np.isclose(0, x, rtol=1e-2, atol=turn·rtol) / np.isclose(0, x, rtol=1e-5) accept
    double atol;                 // turn_rad · rtol
    float c_turn, c_sense;       // cos(x_turn), cos(sense_rad): the same tests on the cosine of the angle (generic directions)
    float t_grad;                // tan(x_grad)
    const uint32_t* turn_bits;   // the step's random turn bits, one per slot id (die_rng.h die_turn_word), or NULL: evaluate Philox
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

// renormalize_radians (core/utils.py:178-180) as numpy evaluates it in float64: (r − π) % (−2π) + π, `%` being
// np.remainder (fmod, then the result takes the divisor's sign).  Every operation is an IEEE one, so the value — down
// to the bit that decides an exact tie — is the reference's.
#define DIE_PI_D 3.141592653589793
#define DIE_2PI_D 6.283185307179586
__device__ __forceinline__ double renorm_rad(double r) {
    // fmod(a, −2π) for |a| of a few π: the quotient is a small integer, so a − q·b is exact in one fma (an fmod result
    // is always representable); a quotient misjudged by the rounding of a / b shows as a wrong sign or size and is
    // put right — the library fmod's general loop is 10 µs of the agent kernel at 2.5 M agents
    const double a = r - DIE_PI_D, b = -DIE_2PI_D;
    double m = fma(-trunc(a * (-1.0 / DIE_2PI_D)), b, a);        // (a quotient estimate: no float64 division)
    if (m != 0.0 && (m < 0.0) != (a < 0.0)) m += a < 0.0 ? -DIE_2PI_D : DIE_2PI_D;
    if (fabs(m) >= DIE_2PI_D) m -= a < 0.0 ? -DIE_2PI_D : DIE_2PI_D;
    // np.remainder: the result takes the divisor's sign
    if (m != 0.0) { if (!(m < 0.0)) m += b; }
    else m = -0.0;
    return m + DIE_PI_D;
}

// The same value for the arguments this path produces (a heading in (−π, 2π] plus or minus an angle in [−π, π]: a = r − π
// in (−3π, 2π)): there fmod's quotient is 0 or 1 and the function is a − 2π (a > 0), a + 2π (a <= −2π, exact) or a, then
// + π — three additions instead of the general form's twenty-odd float64 operations (two of these per agent and step were a
// tenth of the agent kernel's vector-issue time).  Arguments outside that range take the general form.
__device__ __forceinline__ double renorm_rad_fast(double r) {
    const double a = r - DIE_PI_D;
    if (__builtin_expect(!(fabs(a + 0.75 * DIE_PI_D) < 2.75 * DIE_PI_D), 0)) return renorm_rad(r);      // a outside (−3.5π, 2π)
    const double k = a > 0.0 ? -DIE_2PI_D : (a <= -DIE_2PI_D ? DIE_2PI_D : 0.0);
    return (a + k) + DIE_PI_D;
}

__device__ __forceinline__ double die_heading_ld(const uint32_t* hi, const uint32_t* lo, int64_t n) {
    return __hiloint2double((int)hi[n], (int)lo[n]);
}
__device__ __forceinline__ void die_heading_st(uint32_t* hi, uint32_t* lo, int64_t n, double d) {
    hi[n] = (uint32_t)__double2hiint(d);
    lo[n] = (uint32_t)__double2loint(d);
}

// np.angle(x + 1j*y) as numpy evaluates it (core/utils.py:158-169): 1j*y is (0·y − 0) + (0 + y)j and the
// sum with x adds the real parts, so y = −0 becomes +0 and x = −0 survives only next to a negative y:
// angle(−0, −0) = +π, every other pair of zeros gives 0.  Only exact zeros are affected.
__device__ __forceinline__ float die_np_angle(float x, float y) {
    const float re = x + (0.f * y - 0.f);
    const float im = 0.f + (0.f + y);
    return atan2f(im, re);
}

// polar2xy(r, heading) of _discrete_turn (core/utils.py:158-169): (r + 0j)·(cos + i·sin) with its zero signs, on the
// float64 heading rounded to fp32.  One function, because die_pic_action_physarum (die_pic.hip) re-derives a PhysarumAgent's
// action from the stored heading with it and must reproduce the step's bits.
__device__ __forceinline__ void die_polar2xy_heading(double heading, float r, float* ux, float* uy) {
    float s2, c2;
    die_sincos((float)heading, &s2, &c2);
    *ux = r * c2 - 0.f * s2;
    *uy = r * s2 + 0.f * c2;
}

struct FwdOut {
    float dx, dy, dep;
    double heading;
    float ux, uy;              // the vector the action is `scale` times: with momentum, the new _prev_grad (gradient.py:89)
};

// The 4 chem taps of np.gradient at the probe cell (px, py) — central inside the world, one-sided at its four edges
// (gradient.py:57) — and the food under the agent's own cell (cx, cy).
struct FwdTaps { float cxm, cxp, cym, cyp, f_own; };

// Reads the 4 chem taps around the probe cell and the food under the agent, decides the turn,
// applies momentum, updates _prev_grad in place (when kept) and returns heading' and the action.
// EXT = false compiles the sense-mask test out (the benchmark path; chosen at launch when mask == NULL).
// Where the taps come from: global memory (the planes of FwdArgs) …
template <typename T, bool EXT>
struct FwdGlobalMem {
    static constexpr bool kSmallOffsets = false;
    const T* chem;
    const T* food;
    const uint8_t* mask;
    die_geo g;
    __device__ __forceinline__ explicit FwdGlobalMem(const FwdArgs& a) : chem((const T*)a.chem), food((const T*)a.food), mask(a.mask), g(a.g) {}
    // the agent sees medium.where(sense_mask, 0) (core/env.py:292-295): a hidden cell reads as 0
    __device__ __forceinline__ float seen(const T* p, const int64_t i) const { return (EXT && mask && !mask[i]) ? 0.f : die_ld(p, i); }
    __device__ __forceinline__ FwdTaps taps(int px, int py, int xm, int xp, int ym, int yp, int cx, int cy) const {
        FwdTaps t;
        t.cxm = seen(chem, die_local(g, xm, py)); t.cxp = seen(chem, die_local(g, xp, py));
        t.cym = seen(chem, die_local(g, px, ym)); t.cyp = seen(chem, die_local(g, px, yp));
        t.f_own = seen(food, die_local(g, cx, cy));
        return t;
    }
};
// … or a tile staged in LDS (die_pic.hip): chem with a margin of the probe reach around the tile, food of the tile itself
template <typename T, bool TILED = false>
struct FwdTileMem {
    static constexpr bool kSmallOffsets = true;       // the staged margin bounds |sense_offset| far below half the world
    const T* chem;          // plane element (lx, ly) at (lx − cx0)·pitch + (ly − cy0)
    const T* food;          // plane element (lx, ly) at (lx − fx0)·fpitch + (ly − fy0)
    int cx0, cy0, pitch, fx0, fy0, fpitch;
    die_geo g;              // TILED (the planes are a tile of a decomposed world): world cell → plane element first
    // TILED: where the planes hold the cells around an agent that stands INSIDE them, from ONE mapping of the agent's own cell
    // (cx, cy): probes clamp at the world's edge, so a tap never lies across the world's seam from the agent, and
    // die_plane_coord(gx) == clamp(die_plane_coord(cx) + (gx − cx), 0, W − 1) for every tap within the staged margin — provided
    // the world is larger than the planes by more than twice that margin, or the planes span it (checked on the host,
    // die_pic_forward_env_step).  Ten mappings per agent were 8 % of the agent kernel's instructions.
    __device__ __forceinline__ FwdTaps taps(int px, int py, int xm, int xp, int ym, int yp, int cx, int cy) const {
        FwdTaps t;
        if (TILED) {
            const int hx = die_plane_coord(cx, g.ox, g.W, g.gW) - cx, hy = die_plane_coord(cy, g.oy, g.H, g.gH) - cy;
            auto lx = [&](int gx) { return min(max(gx + hx, 0), g.W - 1) - cx0; };
            auto ly = [&](int gy) { return min(max(gy + hy, 0), g.H - 1) - cy0; };
            const int rp = lx(px) * pitch, cp = ly(py);
            t.cxm = die_ld(chem, (int64_t)(lx(xm) * pitch + cp)); t.cxp = die_ld(chem, (int64_t)(lx(xp) * pitch + cp));
            t.cym = die_ld(chem, (int64_t)(rp + ly(ym))); t.cyp = die_ld(chem, (int64_t)(rp + ly(yp)));
            t.f_own = die_ld(food, (int64_t)((cx + hx - fx0) * fpitch + (cy + hy - fy0)));
        } else {
            // one multiply for the probe cell, the four taps by their distance from it (0 where the world ends)
            const int c = (px - cx0) * pitch + (py - cy0);
            t.cxm = die_ld(chem, (int64_t)(c - (px - xm) * pitch)); t.cxp = die_ld(chem, (int64_t)(c + (xp - px) * pitch));
            t.cym = die_ld(chem, (int64_t)(c - (py - ym))); t.cyp = die_ld(chem, (int64_t)(c + (yp - py)));
            t.f_own = die_ld(food, (int64_t)((cx - fx0) * fpitch + (cy - fy0)));
        }
        return t;
    }
    // … and by plane element, for a caller that has mapped the cell already
    __device__ __forceinline__ float food_plane(int px_, int py_) const { return die_ld(food, (int64_t)((px_ - fx0) * fpitch + (py_ - fy0))); }
};

// g / |g| with 0/0 → 0 (gradient.py:60-63) and |g| itself, to ≈ 2 ulp: the components are scaled by a power of two that
// brings the larger one to [0.5, 1) (exact; no square overflows or vanishes, whatever the chem values' magnitude — they
// decay towards the denormals), then one v_rsq_f32 and two products.  (The correctly rounded sqrt and two IEEE divisions
// this replaces were 50 instructions of every forward.)
__device__ __forceinline__ void die_normalize2(float gx, float gy, float* ux, float* uy, float* norm) {
    const float gm = fmaxf(fabsf(gx), fabsf(gy));
    const int e = __builtin_amdgcn_frexp_expf(gm);
    const float sx = ldexpf(gx, -e), sy = ldexpf(gy, -e);
    const float n2 = sx * sx + sy * sy;
    const float inv = __builtin_amdgcn_rsqf(n2);
    const bool ok = n2 > 0.f;
    *ux = ok ? sx * inv : 0.f;
    *uy = ok ? sy * inv : 0.f;
    *norm = ok ? ldexpf(n2 * inv, e) : 0.f;
}

// TB: the random turn bits come from the step's table (FwdArgs.turn_bits, filled by die_turn_bits_fill) instead of a Philox
// evaluation per agent — the same bits (die_rng.h).
// PGSTORE = false: _prev_grad is read at index n but NOT updated in place — the caller stores FwdOut.ux / uy where the agent
// goes (the tile-binned step: the agent's index changes with the step).
template <typename T, int KIND, bool EXT, class MEM, bool TB = false, bool PGSTORE = true>
__device__ __forceinline__ FwdOut die_forward_agent_mem(const FwdArgs& a, const MEM& mem, const uint32_t X, const uint32_t Y, const double d64,
                                                       const uint32_t sid, const int64_t n) {
    const float d = (float)d64;             // trigonometry in fp32 (1e-7 of a cell on the probe), decisions in float64
    const die_geo g = a.g;
    const int W = g.gW, H = g.gH;           // world size: probes clamp at the world's edge
#ifndef DIE_TB_LAZY
#define DIE_TB_LAZY 0      // 1: the table word is loaded only by the lanes whose turn IS random (A/B: scratch/variants.log)
#endif
    uint32_t tbits = 0;
    if (TB && KIND == DIE_AGENT_PHYSARUM && !DIE_TB_LAZY) tbits = a.turn_bits[sid >> 5];      // (requested first: needed last)
    float sd, cd;
    die_sincos(d, &sd, &cd);
    // probe cell: agents + sense_offset·(cos d, sin d), nearest label, clamped (gradient.py:73-76,105)
    const int px = MEM::kSmallOffsets ? die_cell_off(X, (int32_t)die_q32_small(a.sense_offset * cd), W) : die_cell((int64_t)X + die_q32(a.sense_offset * cd), W);
    const int py = MEM::kSmallOffsets ? die_cell_off(Y, (int32_t)die_q32_small(a.sense_offset * sd), H) : die_cell((int64_t)Y + die_q32(a.sense_offset * sd), H);
    // np.gradient at the probe cell: central inside, one-sided at the four edges (gradient.py:57)
    const int xm = px > 0 ? px - 1 : 0, xp = px < W - 1 ? px + 1 : W - 1;
    const int ym = py > 0 ? py - 1 : 0, yp = py < H - 1 ? py + 1 : H - 1;
    const int cx = die_cell_u(X, W), cy = die_cell_u(Y, H);
    // the taps, and the food under the agent (gradient.py:114-116)
    const FwdTaps t = mem.taps(px, py, xm, xp, ym, yp, cx, cy);
    const float gx = (t.cxp - t.cxm) * ((xp - xm) == 2 ? 0.5f : 1.0f);
    const float gy = (t.cyp - t.cym) * ((yp - ym) == 2 ? 0.5f : 1.0f);
    float nx, ny, norm;
    die_normalize2(gx, gy, &nx, &ny, &norm);
    float ux = a.normalized ? nx : gx, uy = a.normalized ? ny : gy;
    // grad *= (norm >= grad_clip) (gradient.py:64-66): a masked component becomes a SIGNED zero, and
    // np.angle(∓0 ∓0j) below is 0, −0, π or −π by quadrant — a sub-threshold gradient with gx < 0 is
    // therefore NOT "undetermined" in the reference.  Keep the signs.
    const bool clipped = a.grad_clip >= 0.f && !(norm >= a.grad_clip);
    if (clipped) { ux = copysignf(0.f, ux); uy = copysignf(0.f, uy); }

    double d_new = d64;
    float dep_mask = 1.0f;
    bool heading_from_vector = true;
    if (KIND == DIE_AGENT_PHYSARUM) {
        // _discrete_turn / _choose_turn (gradient.py:168-208).  np.angle(ux + 1j·uy) (core/utils.py:158-169): 1j·uy is
        // (0·uy − 0) + (0 + uy)j and the sum with ux adds the real parts, so uy = −0 becomes +0 and ux = −0 survives only
        // next to a negative uy.
        const float re = ux + (0.f * uy - 0.f);
        const float im = 0.f + (0.f + uy);
        bool und_grad, und_turn, unseen, right;
        if (im == 0.f || re == 0.f) {
            // an axis-aligned direction — every sub-threshold gradient, and the symmetric situations in which the reference's
            // decisions are exact ties (a probe on the axis of an isolated deposit sees a gradient exactly 90° off the heading):
            // the angle is an exact float64 constant and the tests are the reference's float64 operations
            const double drads = im == 0.f ? ((re < 0.f || (re == 0.f && signbit(re))) ? DIE_PI_D : 0.0) : copysign(0.5 * DIE_PI_D, (double)im);
            const double delta = renorm_rad_fast(d64 - drads);
            und_grad = fabs(drads) <= a.x_grad;
            und_turn = fabs(delta) <= a.x_turn;
            unseen = fabs(delta) > a.sense_rad;
            right = delta > a.atol;
        } else {
            // any other direction: the same tests on |u|·cos and |u|·sin of delta = heading − angle(u), from the heading's
            // (cos, sin) — no arctangent, no float64.  A decision differs from the float64 one only within ≈ 3e-7 rad of a
            // threshold (the fp32 arctangent this replaces was no closer).
            const float nrm = a.normalized ? 1.f : norm;
            const float c = cd * ux + sd * uy, s = sd * ux - cd * uy;
            und_grad = ux > 0.f && fabsf(uy) <= a.t_grad * ux;
            und_turn = c >= a.c_turn * nrm;
            unseen = c < a.c_sense * nrm;
            right = s > 0.f;
        }
        const bool und = und_grad || und_turn || unseen;
        double sgn;
        if (und) {
            if (a.turn_sign) sgn = (double)a.turn_sign[sid];
            else if (TB) { if (DIE_TB_LAZY) tbits = a.turn_bits[sid >> 5]; sgn = ((tbits >> (sid & 31u)) & 1u) ? 1.0 : -1.0; }
            else sgn = die_turn_bit(a.seed, a.step + (a.step_base ? *a.step_base : 0u), sid) ? 1.0 : -1.0;
        } else {
            sgn = right ? -1.0 : 1.0;         // right (clockwise) / left
        }
        const double d2_64 = renorm_rad_fast(d64 + sgn * a.turn_rad);
        const float r = a.normalized ? 1.f : (clipped ? 0.f : norm);      // abs(z) of _discrete_turn
        die_polar2xy_heading(d2_64, r, &ux, &uy);
        dep_mask = (und_grad || und_turn) ? 0.1f : 1.0f;   // clip(mask, .1, 1) (gradient.py:210-214)
        d_new = d2_64;            // (the reference re-derives it as angle(exp(i·d2)): the same value up to one ulp of libm noise)
        heading_from_vector = !a.normalized;               // |g| may be 0 there: angle(0) = 0
    }
    // _process_momentum (gradient.py:82-91)
    if (a.inertia != 0.f || a.noise_scale != 0.f) {
        float nx_ = 0.f, ny_ = 0.f;
        if (a.noise_scale != 0.f) {
            const die_u32x4 r = die_draw(a.seed, a.step + (a.step_base ? *a.step_base : 0u), (uint64_t)sid, DIE_STREAM_NOISE);
            const float u1 = ((float)r.v[0] + 1.0f) * 2.3283064365386963e-10f;
            const float u2 = (float)r.v[1] * 2.3283064365386963e-10f;
            const float rad = 0.4f * sqrtf(-2.0f * logf(u1));
            float sn, cn;
            sincosf(DIE_2PI_F * u2, &sn, &cn);
            nx_ = rad * cn;
            ny_ = rad * sn;
        }
        const float ox = a.pgx ? a.pgx[n] : 0.f, oy = a.pgy ? a.pgy[n] : 0.f;
        ux = (1.f - a.inertia) * ux + a.inertia * ox + a.noise_scale * nx_;
        uy = (1.f - a.inertia) * uy + a.inertia * oy + a.noise_scale * ny_;
        heading_from_vector = true;
    }
    else {                                                 // (1−0)·g + 0·prev + 0·noise: the sum ends with "+ (+0)", which turns −0 into +0
        ux += 0.f;
        uy += 0.f;
    }
    if (PGSTORE && a.pgx) { a.pgx[n] = ux; a.pgy[n] = uy; }
    if (heading_from_vector) d_new = (double)die_np_angle(ux, uy);    // get_radians (gradient.py:110)
    FwdOut o;
    o.heading = d_new;
    o.dx = ux * a.scale;
    o.dy = uy * a.scale;
    o.dep = a.deposit * t.f_own * dep_mask;
    o.ux = ux; o.uy = uy;
    return o;
}

template <typename T, int KIND, bool EXT = true>
__device__ __forceinline__ FwdOut die_forward_agent(const FwdArgs& a, const uint32_t X, const uint32_t Y, const double d,
                                                   const uint32_t sid, const int64_t n) {
    return die_forward_agent_mem<T, KIND, EXT>(a, FwdGlobalMem<T, EXT>(a), X, Y, d, sid, n);
}

// host side: validate and fill FwdArgs (shared by die_gradient_forward and die_forward_env_step)
int die_fill_fwd_args(FwdArgs& k, const die_medium* m, const die_agents* a, const die_gradient_agent* g,
                      const die_action* out, const char* who);
