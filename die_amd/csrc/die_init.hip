// DataInitializer on device (gfx950) — core/data_init.py:92-253 as used by Env._init_data
// (core/env.py:74-86) and the agent constructors (core/agent/gradient.py:42-43,163).
//
//   k_init_medium   with_agents(ratio) (:222-226) + synthetic with_food (:228-231) + zero chem
//   k_count / k_scan_blocks / k_scatter
//                   agents_from_medium (:133-150): stream compaction of the occupied cells in
//                   row-major order into slots [0, K) — three passes, 4096 cells per workgroup
//   k_init_heading  _get_some_noise → get_radians → discretize
#include "die_common.h"
#include "die_rng.h"

#define SCAN_ITEMS 16
#define SCAN_TILE (DIE_BLOCK * SCAN_ITEMS)

struct FoodArgs {
    int n_waves;
    float scale;
    int perlin_octaves;        // > 0: Perlin food, _mask(perlin(x·octaves, y·octaves).round(3), mask_above=threshold) (:228-231)
    float threshold;
    double fx[8], fy[8], phase[8], amp[8];
};

// _mask (core/data_init.py:181-185) of a value already rounded to 3 decimals
__device__ __forceinline__ double mask_range(double v, double below, double above) { return (below <= v && v <= above) ? v : 0.0; }

// food value of world cell (ix, iy): with_food_perlin (:228-231) on the linspace(0, 1, n) labels, or the sinusoid mix
__device__ __forceinline__ float init_food_value(const die_geo& g, const FoodArgs& fa, uint64_t seed, int ix, int iy) {
    if (fa.perlin_octaves > 0) {
        const double x = g.gW > 1 ? (double)ix / (double)(g.gW - 1) : 0.0, y = g.gH > 1 ? (double)iy / (double)(g.gH - 1) : 0.0;
        const double p = rint(die_perlin2(seed, x * fa.perlin_octaves, y * fa.perlin_octaves) * 1000.0) / 1000.0;
        return (float)mask_range(p, 0.0, (double)fa.threshold);
    }
    const double x = (double)ix / g.gW, y = (double)iy / g.gH;
    double s = 0.0;
    for (int k = 0; k < fa.n_waves; ++k)
        s += fa.amp[k] * sin(6.283185307179586476925 * (fa.fx[k] * x + fa.fy[k] * y) + fa.phase[k]);
    s = 2.0 * (double)fa.scale * s;
    s = s < 0.0 ? 0.0 : (s > (double)fa.scale ? (double)fa.scale : s);
    return (float)(rint(s * 1000.0) / 1000.0);   // .round(3) (:196)
}

template <typename T>
__global__ __launch_bounds__(DIE_BLOCK) void k_init_medium(die_geo g, uint64_t* owner, T* food, T* chem, double ratio,
                                                           uint64_t seed, FoodArgs fa) {
    const int W = g.W, H = g.H;
    const int64_t C = (int64_t)W * H;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < C; c += stride) {
        // ceil(u·[0 ≤ u ≤ ratio]) with u = random_sample().round(3): occupied iff 0 < u ≤ ratio
        // world cell of this element (a decomposed tile's halo wraps around the world)
        const int lx = (int)(c / H), ly = (int)(c - (int64_t)lx * H);
        int ix = (lx + g.ox) % g.gW, iy = (ly + g.oy) % g.gH;
        ix = ix < 0 ? ix + g.gW : ix;
        iy = iy < 0 ? iy + g.gH : iy;
        const uint64_t gc = (uint64_t)ix * (uint64_t)g.gH + (uint64_t)iy;
        const int r = die_round3_units(die_draw(seed, 0, gc, DIE_STREAM_INIT_AGENTS).v[0]);
        const double u = r / 1000.0;
        owner[c] = (r > 0 && u <= ratio) ? 1ull : 0ull;    // provisional flag; k_scatter writes the claim word
        die_st(food, c, init_food_value(g, fa, seed, ix, iy));
        die_st(chem, c, 0.f);
    }
}

__global__ __launch_bounds__(DIE_BLOCK) void k_count(const uint64_t* flag, int64_t C, int32_t* block_sum) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) cnt += (base + i < C && flag[base + i] != 0) ? 1 : 0;
    __shared__ int s[DIE_BLOCK];
    s[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = DIE_BLOCK / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sum[blockIdx.x] = s[0];
}

// single workgroup: exclusive scan of nb block sums → block_off (int64), total → *total
__global__ __launch_bounds__(DIE_BLOCK) void k_scan_blocks(const int32_t* block_sum, int nb, int64_t* block_off,
                                                            int64_t* total, int64_t capacity) {
    __shared__ long long s[DIE_BLOCK];
    const int per = (nb + DIE_BLOCK - 1) / DIE_BLOCK;
    const int lo = threadIdx.x * per, hi = min(lo + per, nb);
    long long t = 0;
    for (int i = lo; i < hi; ++i) t += block_sum[i];
    s[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int i = 0; i < DIE_BLOCK; ++i) { long long v = s[i]; s[i] = run; run += v; }
        total[0] = run < capacity ? run : capacity;
        total[1] = run > capacity ? 1 : 0;           // overflow flag: more agents than slots
    }
    __syncthreads();
    long long run = s[threadIdx.x];
    for (int i = lo; i < hi; ++i) { block_off[i] = run; run += block_sum[i]; }
}

__global__ __launch_bounds__(DIE_BLOCK) void k_scatter(die_geo g, uint64_t* owner, const int64_t* block_off, int64_t N,
                                                       uint32_t* x, uint32_t* y, uint8_t* alive, float* agent_food,
                                                       uint64_t seed) {
    const int H = g.H, W = g.gW;             // labels are world coordinates: linspace(0, 1, gW)
    const int64_t C = (int64_t)g.W * g.H;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int flags = 0, cnt = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < C && owner[base + i] != 0) { flags |= 1 << i; ++cnt; }
    __shared__ int s[DIE_BLOCK];
    s[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 1; o < DIE_BLOCK; o <<= 1) {          // Hillis–Steele inclusive scan
        int v = (int)threadIdx.x >= o ? s[threadIdx.x - o] : 0;
        __syncthreads();
        s[threadIdx.x] += v;
        __syncthreads();
    }
    int64_t k = block_off[blockIdx.x] + (s[threadIdx.x] - cnt);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t c = base + i;
        if (c >= C) break;
        if (flags & (1 << i)) {
            if (k < N) {
                const int lx = (int)(c / H), ly = (int)(c - (int64_t)lx * H);
                const int ix = lx + g.ox, iy = ly + g.oy;
                // x = linspace(0, 1, W)[ix] in Q0.32; the last label 1.0 is held as 2^32 − 1
                const double qx = W > 1 ? (double)ix / (double)(W - 1) * 4294967296.0 : 0.0;
                const double qy = g.gH > 1 ? (double)iy / (double)(g.gH - 1) * 4294967296.0 : 0.0;
                const long long X = __double2ll_rn(qx), Y = __double2ll_rn(qy);
                x[k] = (uint32_t)(X > 0xFFFFFFFFLL ? 0xFFFFFFFFLL : X);
                y[k] = (uint32_t)(Y > 0xFFFFFFFFLL ? 0xFFFFFFFFLL : Y);
                alive[k] = 1;
                // get_random(K, 0.1, 1.0) = 0.9·u.round(3) + 0.1 (:140, :168-169)
                const int r = die_round3_units(die_draw(seed, 0, (uint64_t)k, DIE_STREAM_INIT_AGENT_FOOD).v[0]);
                agent_food[k] = (float)(0.9 * (r / 1000.0) + 0.1);
                owner[c] = die_claim(1, k, 0.f);
            } else {
                owner[c] = 0;
            }
            ++k;
        }
    }
}

__global__ __launch_bounds__(DIE_BLOCK) void k_zero_tail(const int64_t* total, int64_t N, uint32_t* x, uint32_t* y,
                                                         uint8_t* alive, float* agent_food) {
    const int64_t K = total[0];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = K + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        x[n] = 0; y[n] = 0; alive[n] = 0; agent_food[n] = 0.f;
    }
}

__global__ __launch_bounds__(DIE_BLOCK) void k_init_heading(uint32_t* hhi, uint32_t* hlo, float* pgx, float* pgy, int64_t N, double turn,
                                                            uint64_t seed) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        const die_u32x4 r = die_draw(seed, 0, (uint64_t)n, DIE_STREAM_INIT_HEADING);
        // Box–Muller pair: its polar angle is 2π·u2, wrapped to (−π, π] like np.angle
        const double u1 = ((double)r.v[0] + 1.0) * (1.0 / 4294967296.0);
        const double u2 = (double)r.v[1] * (1.0 / 4294967296.0);
        const double rad = 0.4 * sqrt(-2.0 * log(u1));
        const double gx = rad * cos(6.283185307179586476925 * u2), gy = rad * sin(6.283185307179586476925 * u2);
        double ang = atan2(gy, gx);
        if (turn > 0.0) ang = floor(ang / turn) * turn;      // discretize (core/utils.py:183-184)
        const double h = (double)(float)ang;                 // float64 state holding the fp32 rounding of the lattice angle
        hhi[n] = (uint32_t)__double2hiint(h);
        hlo[n] = (uint32_t)__double2loint(h);
        if (pgx) { pgx[n] = (float)gx; pgy[n] = (float)gy; }
    }
}

static int init_grid(int64_t n) {
    int64_t g = (n + DIE_BLOCK - 1) / DIE_BLOCK;
    return (int)(g < 8192 ? (g > 0 ? g : 1) : 8192);
}

int64_t die_ws_scan_bytes(int32_t W, int32_t H) {
    const int64_t nb = ((int64_t)W * H + SCAN_TILE - 1) / SCAN_TILE;
    return ((nb * 4 + 255) & ~(int64_t)255) + ((nb * 8 + 255) & ~(int64_t)255);
}

extern "C" int die_init_medium(const die_medium* m, double agent_ratio, uint64_t seed, const die_food_spec* food,
                               void* stream) {
    DIE_REQUIRE(m && food, "die_init_medium: null argument");
    DIE_REQUIRE(m->W >= 1 && m->H >= 1 && m->owner && m->food && m->chem, "die_init_medium: bad medium");
    DIE_REQUIRE(food->n_waves >= 0 && food->n_waves <= 8, "die_init_medium: n_waves %d outside 0..8", food->n_waves);
    DIE_REQUIRE(m->dtype == DIE_F32 || m->dtype == DIE_F16, "die_init_medium: bad dtype %d", m->dtype);
    FoodArgs fa;
    fa.n_waves = food->n_waves;
    fa.scale = food->scale;
    fa.perlin_octaves = food->perlin_octaves;
    fa.threshold = food->threshold;
    DIE_REQUIRE(food->perlin_octaves >= 0 && food->perlin_octaves < (1 << 19), "die_init_medium: bad perlin_octaves %d", food->perlin_octaves);
    for (int i = 0; i < 8; ++i) { fa.fx[i] = food->fx[i]; fa.fy[i] = food->fy[i]; fa.phase[i] = food->phase[i]; fa.amp[i] = food->amp[i]; }
    const int grid = init_grid((int64_t)m->W * m->H);
    if (m->dtype == DIE_F32)
        k_init_medium<float><<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>(die_geo_of(m), m->owner, (float*)m->food,
                                                                           (float*)m->chem, agent_ratio, seed, fa);
    else
        k_init_medium<__half><<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>(die_geo_of(m), m->owner, (__half*)m->food,
                                                                            (__half*)m->chem, agent_ratio, seed, fa);
    DIE_CHECK_LAUNCH("die_init_medium");
    return DIE_OK;
}

extern "C" int die_init_agents(const die_medium* m, const die_agents* a, uint64_t seed, int64_t* num_alive_dev, void* ws,
                               int64_t ws_bytes, void* stream) {
    DIE_REQUIRE(m && a && num_alive_dev && ws, "die_init_agents: null argument");
    DIE_REQUIRE(m->W >= 1 && m->H >= 1 && m->owner, "die_init_agents: bad medium");
    DIE_REQUIRE(a->N > 0 && a->x && a->y && a->alive && a->agent_food, "die_init_agents: bad agents");
    DIE_REQUIRE(ws_bytes >= die_workspace_bytes(m->W, m->H, a->N), "die_init_agents: workspace too small");
    const int64_t C = (int64_t)m->W * m->H;
    const int64_t nb = (C + SCAN_TILE - 1) / SCAN_TILE;
    DIE_REQUIRE(nb < (1ll << 31), "die_init_agents: field too large");
    char* w = (char*)ws + (int64_t)8192 * 8 * 3;        // after the step partials (die_env.hip WS_PARTS)
    int32_t* block_sum = (int32_t*)w;
    int64_t* block_off = (int64_t*)(w + ((nb * 4 + 255) & ~(int64_t)255));
    hipStream_t s = (hipStream_t)stream;
    k_count<<<(int)nb, DIE_BLOCK, 0, s>>>(m->owner, C, block_sum);
    k_scan_blocks<<<1, DIE_BLOCK, 0, s>>>(block_sum, (int)nb, block_off, num_alive_dev, a->N);
    k_scatter<<<(int)nb, DIE_BLOCK, 0, s>>>(die_geo_of(m), m->owner, block_off, a->N, a->x, a->y, a->alive, a->agent_food, seed);
    k_zero_tail<<<init_grid(a->N), DIE_BLOCK, 0, s>>>(num_alive_dev, a->N, a->x, a->y, a->alive, a->agent_food);
    DIE_CHECK_LAUNCH("die_init_agents");
    return DIE_OK;
}

extern "C" int die_init_heading(uint32_t* heading_hi, uint32_t* heading_lo, float* prev_gx, float* prev_gy, int64_t N, double turn_radians,
                                uint64_t seed, void* stream) {
    DIE_REQUIRE(heading_hi && heading_lo && N > 0, "die_init_heading: bad arguments");
    DIE_REQUIRE((prev_gx == nullptr) == (prev_gy == nullptr), "die_init_heading: prev_gx/prev_gy must come together");
    k_init_heading<<<init_grid(N), DIE_BLOCK, 0, (hipStream_t)stream>>>(heading_hi, heading_lo, prev_gx, prev_gy, N, turn_radians,
                                                                          seed);
    DIE_CHECK_LAUNCH("die_init_heading");
    return DIE_OK;
}

// ---- food flow: WaveSequence.get_flow_operator (core/data_init.py:29-38,71-89) --------------------------------
// food ← scale·z(x, y, t) + (1 − decay)·food with the reference's running-wave field z; x varies along the last
// axis and y along the first (core/utils.py:113-118 builds the grid from the reversed sizes).  float64 arithmetic,
// one rounding to the field dtype.  Tiles evaluate z at their world cells.
template <typename T>
__global__ __launch_bounds__(DIE_BLOCK) void k_food_flow_wave(T* food, die_geo g, double t, double scale, double keep) {
    const int64_t total = (int64_t)g.W * g.H;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const double pi = 3.141592653589793;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int li = (int)(i / g.H), lj = (int)(i - (int64_t)li * g.H);
        int gi = (li + g.ox) % g.gW, gj = (lj + g.oy) % g.gH;
        gi = gi < 0 ? gi + g.gW : gi;
        gj = gj < 0 ? gj + g.gH : gj;
        // np.linspace(0, 1, n)[k] = k·(1/(n−1)); then (v − 0.5)·2
        const double x = ((double)gj * (1.0 / (double)(g.gH - 1)) - 0.5) * 2.0;
        const double y = ((double)gi * (1.0 / (double)(g.gW - 1)) - 0.5) * 2.0;
        const double r = sqrt(x * x + y * y);
        const double rwave = r + cos(pi * x) + sin(0.4 * pi * y);
        const double z_waves = cos(1.0 * pi * (rwave + t));
        const double z_islands = sin(pi * x * 3.0 + t) + cos(pi * y * 3.0 + t);
        const double z = (1.0 - 0.25) * z_waves + 0.25 * z_islands;
        die_st(food, i, (float)(scale * z + keep * (double)die_ld(food, i)));
    }
}

extern "C" int die_food_flow_wave(const die_medium* m, double t, double scale, double decay, void* stream) {
    DIE_REQUIRE(m && m->food && m->W >= 1 && m->H >= 1, "die_food_flow_wave: bad medium");
    DIE_REQUIRE(m->dtype == DIE_F32 || m->dtype == DIE_F16, "die_food_flow_wave: bad field dtype %d", m->dtype);
    const die_geo g = die_geo_of(m);
    DIE_REQUIRE(g.gW >= 2 && g.gH >= 2, "die_food_flow_wave: the world must be at least 2x2");
    const int64_t total = (int64_t)m->W * m->H;
    const int grid = init_grid(total);
    if (m->dtype == DIE_F32) k_food_flow_wave<float><<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>((float*)m->food, g, t, scale, 1.0 - decay);
    else k_food_flow_wave<__half><<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>((__half*)m->food, g, t, scale, 1.0 - decay);
    DIE_CHECK_LAUNCH("die_food_flow_wave");
    return DIE_OK;
}


// PerlinNoiseSequence.__getitem__ (core/data_init.py:55-69) inside FieldSequence.get_flow_operator (:29-38):
// food ← scale · round(noise((x, y, t)), 3) + (1 − decay) · food with x, y the linspace(0, 1, n) labels of the world cell and
// `noise` the 3-D gradient noise at (x, y, t) · octaves.
template <typename T>
__global__ __launch_bounds__(DIE_BLOCK) void k_food_flow_perlin(T* food, die_geo g, double t, double octaves, double scale, double keep, uint64_t seed) {
    const int64_t total = (int64_t)g.W * g.H;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int li = (int)(i / g.H), lj = (int)(i - (int64_t)li * g.H);
        int gi = (li + g.ox) % g.gW, gj = (lj + g.oy) % g.gH;
        gi = gi < 0 ? gi + g.gW : gi;
        gj = gj < 0 ? gj + g.gH : gj;
        const double x = (double)gi / (double)(g.gW - 1), y = (double)gj / (double)(g.gH - 1);
        const double z = rint(die_perlin3(seed, x * octaves, y * octaves, t * octaves) * 1000.0) / 1000.0;
        die_st(food, i, (float)(scale * z + keep * (double)die_ld(food, i)));
    }
}

extern "C" int die_food_flow_perlin(const die_medium* m, double t, int32_t octaves, double scale, double decay, uint64_t seed, void* stream) {
    DIE_REQUIRE(m && m->food && m->W >= 1 && m->H >= 1, "die_food_flow_perlin: bad medium");
    DIE_REQUIRE(m->dtype == DIE_F32 || m->dtype == DIE_F16, "die_food_flow_perlin: bad field dtype %d", m->dtype);
    DIE_REQUIRE(octaves >= 1, "die_food_flow_perlin: octaves %d", octaves);
    const die_geo g = die_geo_of(m);
    DIE_REQUIRE(g.gW >= 2 && g.gH >= 2, "die_food_flow_perlin: the world must be at least 2x2");
    const int64_t total = (int64_t)m->W * m->H;
    const int grid = init_grid(total);
    if (m->dtype == DIE_F32) k_food_flow_perlin<float><<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>((float*)m->food, g, t, (double)octaves, scale, 1.0 - decay, seed);
    else k_food_flow_perlin<__half><<<grid, DIE_BLOCK, 0, (hipStream_t)stream>>>((__half*)m->food, g, t, (double)octaves, scale, 1.0 - decay, seed);
    DIE_CHECK_LAUNCH("die_food_flow_perlin");
    return DIE_OK;
}

// ---- DataInitializer builder steps on plain fp32 arrays (core/data_init.py:171-253) --------------------------------
// with_const (:214-216), with_noise / get_random (:168-169,218-220), with_agents (:222-226), with_food_perlin / with_chem
// (:228-236) fill one channel; build / build_agents (:238-253) multiply by the static mask and hand the channels over.
struct FieldOpArgs {
    float* dst;
    int64_t n;
    int op;                    // die_field_op
    int W, H;                  // perlin: the field shape (labels linspace(0, 1, W) × linspace(0, 1, H))
    double a, b;               // const: a; noise: range [a, b]; agents: ratio = b; perlin: threshold = b, octaves = a
    uint64_t seed;
    uint32_t step, word;
};

__global__ __launch_bounds__(DIE_BLOCK) void k_field_op(FieldOpArgs q) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < q.n; i += stride) {
        double v = 0.0;
        if (q.op == DIE_FIELD_CONST) {
            v = q.a;
        } else if (q.op == DIE_FIELD_NOISE) {                 // (b − a)·random_sample().round(3) + a
            const int r = die_round3_units(die_draw(q.seed, q.step, (uint64_t)i, DIE_STREAM_BUILDER).v[q.word & 3u]);
            v = (q.b - q.a) * (r / 1000.0) + q.a;
        } else if (q.op == DIE_FIELD_AGENTS) {                // ceil(_mask(u, mask_above=ratio)): the stream of die_init_medium
            const int r = die_round3_units(die_draw(q.seed, q.step, (uint64_t)i, DIE_STREAM_INIT_AGENTS).v[0]);
            v = (r > 0 && r / 1000.0 <= q.b) ? 1.0 : 0.0;
        } else {                                              // DIE_FIELD_PERLIN
            const int ix = (int)(i / q.H), iy = (int)(i - (int64_t)ix * q.H);
            const double x = q.W > 1 ? (double)ix / (double)(q.W - 1) : 0.0, y = q.H > 1 ? (double)iy / (double)(q.H - 1) : 0.0;
            const double p = rint(die_perlin2(q.seed + q.step, x * q.a, y * q.a) * 1000.0) / 1000.0;
            v = mask_range(p, 0.0, q.b);
        }
        q.dst[i] = (float)v;
    }
}

extern "C" int die_field_fill(float* dst, int64_t n, int32_t op, int32_t W, int32_t H, double a, double b, uint64_t seed,
                              uint32_t step, uint32_t word, void* stream) {
    DIE_REQUIRE(dst && n > 0, "die_field_fill: bad array");
    DIE_REQUIRE(op >= DIE_FIELD_CONST && op <= DIE_FIELD_PERLIN, "die_field_fill: bad op %d", op);
    DIE_REQUIRE(op != DIE_FIELD_PERLIN || (W >= 1 && H >= 1 && (int64_t)W * H == n && a >= 1.0 && a < 524288.0), "die_field_fill: bad perlin shape");
    FieldOpArgs q;
    q.dst = dst; q.n = n; q.op = op; q.W = W; q.H = H; q.a = a; q.b = b; q.seed = seed; q.step = step; q.word = word;
    k_field_op<<<init_grid(n), DIE_BLOCK, 0, (hipStream_t)stream>>>(q);
    DIE_CHECK_LAUNCH("die_field_fill");
    return DIE_OK;
}

// build (:238-246): the three channels × mask into the medium's own representation (occupied cells get the
// provisional flag die_init_agents turns into claim words).  Any channel / the mask may be NULL (zeros / ones).
template <typename T>
__global__ __launch_bounds__(DIE_BLOCK) void k_medium_from_fields(int64_t C, uint64_t* owner, T* food, T* chem, const float* fa,
                                                                  const float* ff, const float* fc, const float* mask, float mask_scalar) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < C; c += stride) {
        const float mk = mask ? mask[c] : mask_scalar;
        owner[c] = (fa && fa[c] * mk > 0.f) ? 1ull : 0ull;
        die_st(food, c, ff ? ff[c] * mk : 0.f);
        die_st(chem, c, fc ? fc[c] * mk : 0.f);
    }
}

extern "C" int die_medium_from_fields(const die_medium* m, const float* agents, const float* food, const float* chem,
                                      const float* mask, float mask_scalar, void* stream) {
    DIE_REQUIRE(m && m->owner && m->food && m->chem && m->W >= 1 && m->H >= 1, "die_medium_from_fields: bad medium");
    DIE_REQUIRE(m->dtype == DIE_F32 || m->dtype == DIE_F16, "die_medium_from_fields: bad dtype %d", m->dtype);
    const int64_t C = (int64_t)m->W * m->H;
    if (m->dtype == DIE_F32)
        k_medium_from_fields<float><<<init_grid(C), DIE_BLOCK, 0, (hipStream_t)stream>>>(C, m->owner, (float*)m->food, (float*)m->chem,
                                                                                         agents, food, chem, mask, mask_scalar);
    else
        k_medium_from_fields<__half><<<init_grid(C), DIE_BLOCK, 0, (hipStream_t)stream>>>(C, m->owner, (__half*)m->food, (__half*)m->chem,
                                                                                          agents, food, chem, mask, mask_scalar);
    DIE_CHECK_LAUNCH("die_medium_from_fields");
    return DIE_OK;
}
