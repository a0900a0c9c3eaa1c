// Env.step kernels (gfx950) — core/env.py:101-131.
//
//   k_move_claim   _agent_move (:163-172) + ownership claim for _agent_deposit_and_layout (:204-215)
//                  + _agent_feed (:220-243) for alive slots (their own cell is occupied by definition)
//   k_resolve      winner of each occupied cell writes chem += deposit and food -= rate·food;
//                  dead slots finish their feed (they consume iff somebody alive owns their cell);
//                  _agent_lifecycle (:245-261) when agents_die; alive count
//   k_reduce       fixed-order sum of the per-block partials → die_step_result (deterministic)
//   k_diffuse      _medium_diffuse_decay (:136-145): separable gaussian, periodic, × (1 − decay)
//
// Why two agent passes: every slot on a cell must read the food value from BEFORE the step's
// consumption (co-located agents each get the full amount, :224-225), so all reads of `food`
// (pass 1) are separated from the single write per occupied cell (pass 2) by a kernel boundary.
// "Last writer wins" of the fancy-index assignment at :211 is an atomicMax on the slot id.
#include "die_forward.h"

#ifndef DIE_MAX_PARTIALS
#define DIE_MAX_PARTIALS 8192
#endif
// workgroup size of the per-agent step kernels (claim pass, dead-slot pass).  Swept on MI355X at 2.5 M agents, step time in
// µs: 192: 211.5, 256: 206.6, 320: 199.9, 384: 203.4, 448: 204.5, 512: 202.8, 640: 208.0, 768: 201.8 — 5 waves per
// workgroup it is (one agent per thread under the 8192-workgroup cap, and 20 instead of 24 waves per CU: the kernel
// is bound by L1 misses in flight, fewer waves thrash the L1 less).
#ifndef DIE_STEP_BLOCK
#define DIE_STEP_BLOCK 320
#endif

struct StepArgs {
    die_geo g;
    int epoch;
    int do_move, do_claim;     // die_agent_move / die_agent_claim_feed run one half each
    int tile_w, tile_h, tiles_y;
    int32_t* tile_of;
    int64_t N;
    unsigned long long* owner;
    void* food;
    void* chem;
    uint32_t* x;
    uint32_t* y;
    const uint32_t* slot;  // reference slot ids (NULL = identity)
    uint8_t* alive;
    float* agent_food;
    const float* dx;
    const float* dy;
    const float* dep;
    float rate_feed, w_dep, w_dist;
    int boundary, cost, food_infinite, agents_die, has_dead;
    int skip_scatter;      // fused step: the winner's chem/food writes are done by k_diffuse_rows
    float* stash;          // N floats, only when has_dead
    long long* part_gain;  // gridDim.x fixed-point sums (die_fix)
    long long* part_alive; // gridDim.x
};

__device__ __forceinline__ float action_cost(const StepArgs& a, float dx, float dy, float dep) {
    // linear_action_cost (core/env.py:29-35) or zero_cost (:38-39)
    return a.cost == DIE_COST_LINEAR ? a.w_dep * fabsf(dep) + a.w_dist * die_sqrt1(dx * dx + dy * dy) : 0.f;
}

__device__ __forceinline__ void block_sum_store(long long g, long long c, long long* pg, long long* pc) {
    __shared__ long long sg[DIE_STEP_BLOCK / DIE_WAVE];
    __shared__ long long sc[DIE_STEP_BLOCK / DIE_WAVE];
    g = die_wave_sum(g);
    c = die_wave_sum(c);
    const int lane = threadIdx.x & (DIE_WAVE - 1), wv = threadIdx.x / DIE_WAVE;
    if (lane == 0) { sg[wv] = g; sc[wv] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long tg = 0;
        long long tc = 0;
        for (int i = 0; i < DIE_STEP_BLOCK / DIE_WAVE; ++i) { tg += sg[i]; tc += sc[i]; }
        if (pg) pg[blockIdx.x] = tg;
        if (pc) pc[blockIdx.x] = tc;
    }
}

// One slot of _agent_move + claim + _agent_feed (alive slots); returns its `gained` (0 for dead slots).
template <typename T, bool EXT = true>
__device__ __forceinline__ float move_claim_one(const StepArgs& a, const int64_t n, uint32_t X, uint32_t Y, const float dx,
                                                const float dy, const float dep, const uint32_t sid, long long& owned_alive) {
    const T* food = (const T*)a.food;
    const die_geo g = a.g;
    if (a.do_move) {
        if (a.boundary == DIE_BOUNDARY_WRAP) {            // (xy + dxdy) % 1.
            X += (uint32_t)die_q32(dx);
            Y += (uint32_t)die_q32(dy);
        } else {                                          // clip(0, 1); 1.0 is held as 2^32 − 1
            int64_t px = (int64_t)X + die_q32(dx), py = (int64_t)Y + die_q32(dy);
            X = (uint32_t)(px < 0 ? 0 : (px > 0xFFFFFFFFLL ? 0xFFFFFFFFLL : px));
            Y = (uint32_t)(py < 0 ? 0 : (py > 0xFFFFFFFFLL ? 0xFFFFFFFFLL : py));
        }
        a.x[n] = X;
        a.y[n] = Y;
    }
    const int cx = die_cell((int64_t)X, g.gW), cy = die_cell((int64_t)Y, g.gH);
    if (a.tile_of) a.tile_of[n] = (cx / a.tile_w) * a.tiles_y + cy / a.tile_h;
    if (!a.do_claim) return 0.f;
    const int64_t c = die_local(g, cx, cy);
    const float consumed = a.rate_feed * die_ld(food, c);
    if (a.alive[n]) {
        // Claim: one 64-bit atomicMax (a plain store + a repair pass was measured: 70 + 38 µs against 105 µs, no gain)
        atomicMax(&a.owner[c], die_claim(a.epoch, (int64_t)sid, dep));
        const float gained = consumed - action_cost(a, dx, dy, dep);
        a.agent_food[n] += gained;
        if (EXT) {                                     // ghost-agent tiles (compiled out of the single-tile kernel)
            if (!die_owned(g, cx, cy)) return 0.f;     // a ghost: the rank that owns this cell accounts for it
            ++owned_alive;
        }
        return gained;
    }
    if (a.has_dead) a.stash[n] = consumed;
    return 0.f;
}

template <typename T>
__global__ __launch_bounds__(DIE_STEP_BLOCK) void k_move_claim(StepArgs a) {
    long long gsum = 0;
    long long cnt = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < a.N; n += stride) {
        const uint32_t sid = a.slot ? a.slot[n] : (uint32_t)n;
        gsum += die_fix(move_claim_one<T>(a, n, a.x[n], a.y[n], a.dx[n], a.dy[n], a.do_claim ? a.dep[n] : 0.f, sid, cnt));
    }
    if (a.do_claim) block_sum_store(gsum, cnt, a.part_gain, a.part_alive);
}

// Agent.forward fused with the first half of Env.step: the action stays in registers between the two
// (it is still written out for the caller, but never read back), x / y / slot are loaded once.
template <typename T, int KIND, bool EXT>
__device__ __forceinline__ void forward_move_claim_body(FwdArgs& f, StepArgs& a) {
    long long gsum = 0;
    long long cnt = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // (giving every XCD a contiguous eighth of the sorted array instead of round-robin workgroups: 178 µs vs 122 µs)
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < a.N; n += stride) {
        const uint32_t sid = a.slot ? a.slot[n] : (uint32_t)n;
        const uint32_t X = a.x[n], Y = a.y[n];
        const FwdOut o = die_forward_agent<T, KIND, EXT>(f, X, Y, die_heading_ld(f.heading_hi, f.heading_lo, n), sid, n);
        die_heading_st(f.heading_hi, f.heading_lo, n, o.heading);
        if (f.dx) { f.dx[n] = o.dx; f.dy[n] = o.dy; f.dep[n] = o.dep; }
        gsum += die_fix(move_claim_one<T, EXT>(a, n, X, Y, o.dx, o.dy, o.dep, sid, cnt));
    }
    if (a.do_claim) block_sum_store(gsum, cnt, a.part_gain, a.part_alive);
}

// Agent.forward fused with the first half of Env.step: the action stays in registers between the two
// (it is still written out for the caller, but never read back), x / y / slot are loaded once.
// LEAN: the host has checked that the agent has no momentum, no noise, a normalised gradient and no graph-replay step
// word (PhysarumAgent's defaults) — spelled out for the compiler, those paths of the shared forward code and the scalar
// registers that feed them drop out of the kernel (die_pic.hip: −30 % vector instructions, −4 % time).
__device__ __forceinline__ void fwd_args_lean(FwdArgs& f) {
    f.pgx = nullptr; f.pgy = nullptr; f.step_base = nullptr;
    f.inertia = 0.f; f.noise_scale = 0.f; f.normalized = 1;
}
static bool fwd_is_lean(const die_gradient_agent* g) {
    return g->kind == DIE_AGENT_PHYSARUM && g->inertia == 0.f && g->noise_scale == 0.f && g->normalized_grad && !g->prev_gx && !g->prev_gy &&
           !g->step_base;
}

template <typename T, int KIND, bool EXT = true, bool LEAN = false>
__global__ __launch_bounds__(DIE_STEP_BLOCK) void k_forward_move_claim(FwdArgs f, StepArgs a) {
    if (LEAN) fwd_args_lean(f);
    forward_move_claim_body<T, KIND, EXT>(f, a);
}

// R replicas of one world shape in one launch (blockIdx.y = replica): every array of replica r lies `r` strides behind
// replica 0's, the Philox key is seed + r·seed_stride, its agent count n[r] — otherwise the kernel above.
struct BatchArgs {
    int64_t cells, agents;          // strides: cells per plane, agent slots per replica
    uint64_t seed_stride;
    int64_t n[DIE_MAX_REPLICAS];
};

template <typename T, int KIND, bool LEAN = false>
__global__ __launch_bounds__(DIE_STEP_BLOCK) void k_forward_move_claim_batch(FwdArgs f, StepArgs a, BatchArgs b) {
    if (LEAN) fwd_args_lean(f);
    const int r = blockIdx.y;
    const int64_t pc = b.cells * r, pa = b.agents * r;
    f.chem = (const T*)f.chem + pc; f.food = (const T*)f.food + pc;
    f.x += pa; f.y += pa; f.heading_hi += pa; f.heading_lo += pa;
    if (f.slot) f.slot += pa;
    if (f.dx) { f.dx += pa; f.dy += pa; f.dep += pa; }
    f.N = b.n[r]; f.seed += b.seed_stride * (uint64_t)r;
    a.owner += pc; a.food = (T*)a.food + pc; a.chem = (T*)a.chem + pc;
    a.x += pa; a.y += pa; a.alive += pa; a.agent_food += pa;
    if (a.slot) a.slot += pa;
    a.N = b.n[r];
    a.part_gain += (int64_t)DIE_MAX_PARTIALS * 3 * r;               // replica r's own workspace partials
    a.part_alive = a.part_gain + 2 * DIE_MAX_PARTIALS;
    forward_move_claim_body<T, KIND, false>(f, a);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.part_alive[0] = a.N;  // every slot is alive: the sweep's reduction reads the count here
}

template <typename T>
__global__ __launch_bounds__(DIE_STEP_BLOCK) void k_resolve(StepArgs a) {
    T* food = (T*)a.food;
    T* chem = (T*)a.chem;
    long long gsum = 0;
    long long alive_cnt = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < a.N; n += stride) {
        const uint32_t X = a.x[n], Y = a.y[n];
        const int cx = die_cell((int64_t)X, a.g.gW), cy = die_cell((int64_t)Y, a.g.gH);
        const int64_t c = die_local(a.g, cx, cy);
        const bool owned = die_owned(a.g, cx, cy);
        bool alive = a.alive[n] != 0;
        if (alive && !a.skip_scatter) {
            if ((uint32_t)(a.owner[c] >> 32) == die_owner_word(a.epoch, a.slot ? (int64_t)a.slot[n] : n)) {       // highest alive slot on the cell
                die_st(chem, c, die_ld(chem, c) + a.dep[n]);
                if (!a.food_infinite) {
                    const float f = die_ld(food, c);
                    die_st(food, c, f - a.rate_feed * f);
                }
            }
        } else if (!alive && a.has_dead) {
            const float consumed = die_claim_occupied(a.owner[c], a.epoch) ? a.stash[n] : 0.f;
            const float gained = consumed - action_cost(a, a.dx[n], a.dy[n], a.dep[n]);
            a.agent_food[n] += gained;
            if (owned) gsum += die_fix(gained);
        }
        if (a.agents_die && !(a.agent_food[n] > 1e-4f)) {         // where(have_food, 0): every channel
            a.x[n] = 0; a.y[n] = 0; a.alive[n] = 0; a.agent_food[n] = 0.f;
            alive = false;
        }
        alive_cnt += (alive && owned) ? 1 : 0;
    }
    block_sum_store(gsum, alive_cnt, a.part_gain, a.part_alive);
}

__global__ __launch_bounds__(1024) void k_reduce(const long long* pa, int na, const long long* pb, int nb,
                                                  const long long* pc, int nc, die_step_result* out,
                                                  long long alive_const) {
    // one block; integer sums (fixed-point gains, counts): exact in any order
    __shared__ long long sg[1024];
    __shared__ long long sc[1024];
    long long g = 0;
    long long c = 0;
    for (int i = threadIdx.x; i < na; i += 1024) g += pa[i];
    for (int i = threadIdx.x; i < nb; i += 1024) g += pb[i];
    for (int i = threadIdx.x; i < nc; i += 1024) c += pc[i];
    sg[threadIdx.x] = g;
    sc[threadIdx.x] = c;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sg[threadIdx.x] += sg[threadIdx.x + o]; sc[threadIdx.x] += sc[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {         // (`out` may be pinned host memory a sync=True caller polls: die_common.h die_store_result_*)
        die_store_result_f64(&out->reward, (double)sg[0] / DIE_FIX_ONE);
        die_store_result_i64((long long*)&out->num_alive, alive_const >= 0 ? alive_const : sc[0]);
    }
}

// ---- diffusion --------------------------------------------------------------------------
#define DIF_TX 16      // tile rows (x, stride H)
#define DIF_TY 256     // tile columns (y, contiguous)
#define DIF_MAXR 8

struct DiffuseArgs {
    const void* src;
    void* dst;
    int W, H;
    float keep;                 // 1 − decay
    int mode;                   // die_diffuse_mode: what lies beyond the edges
    float w[2 * DIF_MAXR + 1];  // w[k + R], k = −R..R
};

__device__ __forceinline__ int wrap_idx(int v, int n) {
    v %= n;
    return v < 0 ? v + n : v;
}

// scipy.ndimage boundary modes (what skimage.filters.gaussian passes through, core/env.py:140-143): the index that
// stands in for v outside [0, n), or −1 for 'constant' (cval = 0)
__device__ __forceinline__ int edge_idx(int v, int n, int mode) {
    if (v >= 0 && v < n) return v;
    switch (mode) {
        case DIE_DIFFUSE_NEAREST: return v < 0 ? 0 : n - 1;
        case DIE_DIFFUSE_REFLECT: {                    // d c b a | a b c d | d c b a   (period 2n)
            int m = v % (2 * n);
            m = m < 0 ? m + 2 * n : m;
            return m < n ? m : 2 * n - 1 - m;
        }
        case DIE_DIFFUSE_MIRROR: {                     // d c b | a b c d | c b a       (period 2n − 2)
            if (n == 1) return 0;
            const int p = 2 * n - 2;
            int m = v % p;
            m = m < 0 ? m + p : m;
            return m < n ? m : p - m;
        }
        case DIE_DIFFUSE_CONSTANT: return -1;
        default: return wrap_idx(v, n);
    }
}

// LDS-tiled separable gaussian on the torus: the (TX+2R)×(TY+2R) input tile is staged once,
// filtered along x (axis 0, as scipy does first) into a second LDS plane, then along y.
template <typename T, int R>
__global__ __launch_bounds__(DIE_BLOCK) void k_diffuse(DiffuseArgs a) {
    constexpr int LW = DIF_TY + 2 * R + 1;    // odd pitch: bank-conflict-free column walks
    constexpr int LH = DIF_TX + 2 * R;
    __shared__ float s_in[LH * LW];
    __shared__ float s_mid[DIF_TX * LW];
    const T* src = (const T*)a.src;
    T* dst = (T*)a.dst;
    const int x0 = blockIdx.y * DIF_TX, y0 = blockIdx.x * DIF_TY;
    const int H = a.H, W = a.W;
    constexpr int LWV = DIF_TY + 2 * R;       // valid columns
    for (int idx = threadIdx.x; idx < LH * LWV; idx += DIE_BLOCK) {
        const int li = idx / LWV, lj = idx - li * LWV;
        const int gx = edge_idx(x0 - R + li, W, a.mode), gy = edge_idx(y0 - R + lj, H, a.mode);
        s_in[li * LW + lj] = (gx < 0 || gy < 0) ? 0.f : die_ld(src, (int64_t)gx * H + gy);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < DIF_TX * LWV; idx += DIE_BLOCK) {
        const int i = idx / LWV, j = idx - i * LWV;
        float t = a.w[R] * s_in[(i + R) * LW + j];            // scipy's correlate1d on a symmetric kernel: the centre, then
#pragma unroll
        for (int k = 0; k < R; ++k) t += (s_in[(i + k) * LW + j] + s_in[(i + 2 * R - k) * LW + j]) * a.w[k];   // pairs, outermost first
        s_mid[i * LW + j] = t;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < DIF_TX * DIF_TY; idx += DIE_BLOCK) {
        const int i = idx / DIF_TY, j = idx - i * DIF_TY;
        const int gx = x0 + i, gy = y0 + j;
        if (gx < W && gy < H) {
            float o = a.w[R] * s_mid[i * LW + j + R];
#pragma unroll
            for (int k = 0; k < R; ++k) o += (s_mid[i * LW + j + k] + s_mid[i * LW + j + 2 * R - k]) * a.w[k];
            die_st(dst, (int64_t)gx * H + gy, o * a.keep);
        }
    }
}

// ---- row-marching diffusion, optionally fused with deposit + feeding -----------------------
// One wave owns a strip of 248 columns (62 lanes × 4 contiguous cells; lanes 0 and 63 only load
// the 4-column halo to the left / right) and marches down DIF_ROWS rows.  Per row a lane loads its
// 4 cells with one 16-byte access, keeps the last 2R+1 rows in registers (x pass = axis 0 first,
// as scipy does), and gets the y-pass neighbours from the adjacent lanes with wave shuffles — no
// LDS, no barrier.  FUSED: the same sweep reads the 64-bit claim of every cell it loads and adds the
// winner's deposit before filtering (core/env.py:211) and, for the cells it owns, performs the
// feeding update food −= rate·food on occupied cells (:222-228): the per-cell scatter of the step
// becomes two coalesced streams.
#ifndef DIF_ROWS
#define DIF_ROWS 16     // rows per wave: 16 → ≈17 waves per CU at 4096² (32: 87 µs, 16: 75 µs, 8: 86 µs for the fused sweep)
#endif
#define DIF_WCOLS 248          // output columns per wave
#ifndef DIF_BLOCK
#define DIF_BLOCK 128          // waves are independent (no LDS, no barrier): small workgroups balance better
#endif

struct RowsArgs {
    const void* src;
    void* dst;
    const unsigned long long* claim;   // FUSED == 1: the 64-bit claim plane
    const float* dep;                  // FUSED == 2: per cell the winner's deposit, or DIE_DEP_EMPTY (tile-binned step, die_pic.hip)
    void* food;                        // FUSED only
    int W, H, epoch, food_infinite;
    int64_t rep_cells;                 // batched replicas (gridDim.z > 1): plane stride in cells; partials / results stride per replica
    int halo;                          // tile mode: cells within `halo` of the array border belong to neighbours
    int rpw;                           // rows per wave (rows_per_wave): DIF_ROWS on large fields, fewer on small ones
    int wrapx, wrapy;                  // tile mode: this axis spans the whole world (one rank along it) and is periodic
    // k_reduce folded in: workgroup (0, 0) first sums the claim pass's `n_part` partial gains (a kernel boundary lies
    // between their producer and this sweep) in k_reduce's order and writes the step result — one launch less per step
    const long long* part_gain;
    const long long* part_alive;       // NULL: num_alive = alive_const; else the sum of the claim pass's counts (ghost tiles)
    int n_part;
    die_step_result* result;
    long long alive_const;
    float keep, rate_feed;
    float w[2 * 4 + 1];
};

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ void ld(const float* p, float v[4]) { const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ void st(float* p, const float v[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec4<__half> {
    static __device__ __forceinline__ void ld(const __half* p, float v[4]) {
        const uint2 t = *(const uint2*)p;
        const __half2 a = *(const __half2*)&t.x, b = *(const __half2*)&t.y;
        v[0] = __low2float(a); v[1] = __high2float(a); v[2] = __low2float(b); v[3] = __high2float(b);
    }
    static __device__ __forceinline__ void st(__half* p, const float v[4]) {
        const __half2 a = __halves2half2(die_f2h(v[0]), die_f2h(v[1])), b = __halves2half2(die_f2h(v[2]), die_f2h(v[3]));
        uint2 t; t.x = *(const uint32_t*)&a; t.y = *(const uint32_t*)&b;
        *(uint2*)p = t;
    }
};

template <typename T, int R, int FUSED, bool WRAP>
__global__ __launch_bounds__(DIF_BLOCK) void k_diffuse_rows(RowsArgs a) {
    static_assert(R >= 1 && R <= 4, "one halo lane of 4 columns per side");
    if (gridDim.z > 1) {                                    // replica blockIdx.z of a batch: same shape, arrays one stride apart
        const int64_t pc = a.rep_cells * blockIdx.z;
        a.src = (const T*)a.src + pc; a.dst = (T*)a.dst + pc;
        if (FUSED == 1) a.claim += pc;
        if (FUSED) a.food = (T*)a.food + pc;
        if (a.result) {
            a.part_gain += (int64_t)DIE_MAX_PARTIALS * 3 * blockIdx.z;
            if (a.part_alive) a.part_alive += (int64_t)DIE_MAX_PARTIALS * 3 * blockIdx.z;
            a.result += blockIdx.z;
        }
    }
    const int row0 = FUSED && a.result ? 1 : 0;             // grid rows ahead of the field's
    if (row0 && blockIdx.y == 0) {
        // an extra row of workgroups ahead of the field: (0, 0) reduces, the others have nothing to do.  (Doing it as a
        // prologue of a field workgroup made the whole sweep 10 µs longer: all its workgroups are resident at once, so
        // the kernel ends when its slowest workgroup does.)  The FIRST row since round 5: the reduction depends on the claim
        // pass only, and dispatched first it hands the step's result to an Env(sync=True) caller (pinned host memory) while
        // the field is still being swept — the caller's next launches are queued before this kernel ends (die_pic.hip's field
        // kernel does the same).
        if (blockIdx.x != 0) return;
        __shared__ long long s_g[DIF_BLOCK];
        // 8 loads in flight per thread: this workgroup must not outlast the sweep (one load at a time it took ≈ 45 µs per
        // array, and with the second array of ghost tiles it made the whole launch 100 µs instead of 71)
        auto strided_sum = [&](const long long* p) {
            long long t = 0;
            for (int base = threadIdx.x; base < a.n_part; base += DIF_BLOCK * 8) {
                long long v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const int i = base + q * DIF_BLOCK; v[q] = i < a.n_part ? p[i] : 0; }
#pragma unroll
                for (int q = 0; q < 8; ++q) t += v[q];
            }
            return t;
        };
        long long g = strided_sum(a.part_gain);                    // fixed point: any order
        s_g[threadIdx.x] = g;
        __syncthreads();
        for (int o = DIF_BLOCK / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) s_g[threadIdx.x] += s_g[threadIdx.x + o];
            __syncthreads();
        }
        long long cnt = 0;
        if (a.part_alive) {                                    // integer sum: any order
            __shared__ long long s_c[DIF_BLOCK];
            cnt = strided_sum(a.part_alive);
            s_c[threadIdx.x] = cnt;
            __syncthreads();
            for (int o = DIF_BLOCK / 2; o > 0; o >>= 1) {
                if ((int)threadIdx.x < o) s_c[threadIdx.x] += s_c[threadIdx.x + o];
                __syncthreads();
            }
            cnt = s_c[0];
        }
        if (threadIdx.x == 0) {
            die_store_result_f64(&a.result->reward, (double)s_g[0] / DIE_FIX_ONE);
            die_store_result_i64((long long*)&a.result->num_alive, a.part_alive ? cnt : a.alive_const);
        }
        return;
    }
    const T* src = (const T*)a.src;
    T* dst = (T*)a.dst;
    T* food = (T*)a.food;
    const int W = a.W, H = a.H;
    const int lane = threadIdx.x & (DIE_WAVE - 1);
    const int strip = blockIdx.x * (DIF_BLOCK / DIE_WAVE) + (threadIdx.x >> 6);
    const int yb = strip * DIF_WCOLS;                       // first output column of this wave
    if (yb >= H) return;                                    // whole wave idle (nothing below synchronises)
    const int nout = min(DIF_WCOLS, H - yb) / 4;            // output lanes are 1..nout (H % 4 == 0)
    const bool need = lane <= nout + 1;                     // + the two halo lanes
    const bool outl = lane >= 1 && lane <= nout;
    const bool wx = WRAP || a.wrapx, wy = WRAP || a.wrapy;
    const int hx = wx ? 0 : a.halo, hy = wy ? 0 : a.halo;
    const int col = wy ? wrap_idx(yb + 4 * (lane - 1), H)    // 16-byte aligned since H % 4 == 0
                       : min(max(yb + 4 * (lane - 1), 0), H - 4);     // tile: clamp, border ring is don't-care
    const int x0 = ((int)blockIdx.y - row0) * a.rpw;
    const int rows = min(a.rpw, W - x0);

    float win[2 * R + 1][4];
#pragma unroll
    for (int k = 0; k <= 2 * R; ++k) { win[k][0] = win[k][1] = win[k][2] = win[k][3] = 0.f; }

    auto load_row = [&](int i, float v[4]) {
        v[0] = v[1] = v[2] = v[3] = 0.f;
        if (!need) return;
        const int r = wx ? wrap_idx(x0 + i, W) : min(max(x0 + i, 0), W - 1);
        const int64_t off = (int64_t)r * H + col;
        Vec4<T>::ld(src + off, v);
        if (FUSED) {
            bool occ[4];
            bool any = false;
            if (FUSED == 1) {
                const ulonglong2 c01 = *(const ulonglong2*)(a.claim + off), c23 = *(const ulonglong2*)(a.claim + off + 2);
                const unsigned long long c[4] = {c01.x, c01.y, c23.x, c23.y};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    occ[j] = die_claim_occupied(c[j], a.epoch);
                    if (occ[j]) v[j] = die_as_stored<T>(v[j] + die_claim_deposit(c[j]));     // chem[cell] + deposit of the last writer
                    any |= occ[j];
                }
            } else {
                const uint4 d4 = *(const uint4*)(a.dep + off);
                const uint32_t d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    occ[j] = d[j] != DIE_DEP_EMPTY;
                    if (occ[j]) v[j] = die_as_stored<T>(v[j] + __uint_as_float(d[j]));
                    any |= occ[j];
                }
            }
            // feeding: this wave owns rows [x0, x0+rows) × its output lanes (tile mode: interior cells only)
            const bool own_row = i >= 0 && i < rows && r >= hx && r < W - hx;
            if (any && outl && own_row && !a.food_infinite) {
                float f[4];
                Vec4<T>::ld(food + off, f);
                bool changed = false;                           // (an occupied cell without food stays as it is: a group in which nothing changes is not written)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool mine = col + j >= hy && col + j < H - hy;
                    if (occ[j] && mine && f[j] != 0.f) { f[j] = f[j] - a.rate_feed * f[j]; changed = true; }
                }
                if (changed) Vec4<T>::st(food + off, f);
            }
        }
    };

    float nxt[4];
    load_row(-R, nxt);
    for (int i = -R; i < rows + R; ++i) {
        float cur[4] = {nxt[0], nxt[1], nxt[2], nxt[3]};
        if (i + 1 < rows + R) load_row(i + 1, nxt);            // prefetch one row ahead
#pragma unroll
        for (int k = 0; k < 2 * R; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) win[k][j] = win[k + 1][j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) win[2 * R][j] = cur[j];
        const int orow = i - R;                                // the window is centred on this row
        if (orow < 0) continue;
        float xf[4 + 2 * 4];                                   // [4 − R .. 4 + 4 + R): own 4 at [4..8)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = a.w[R] * win[R][j];                      // centre, then symmetric pairs from the outermost inwards: mirror
#pragma unroll
            for (int k = 0; k < R; ++k) t += (win[k][j] + win[2 * R - k][j]) * a.w[k];      // cells get identical sums, as in scipy
            xf[4 + j] = t;
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            xf[4 - R + j] = __shfl_up(xf[4 + 4 - R + j], 1, DIE_WAVE);      // left lane's last R cells
            xf[8 + j] = __shfl_down(xf[4 + j], 1, DIE_WAVE);                 // right lane's first R cells
        }
        if (outl) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = a.w[R] * xf[4 + j];
#pragma unroll
                for (int k = 0; k < R; ++k) t += (xf[4 + j - R + k] + xf[4 + j + R - k]) * a.w[k];
                o[j] = t * a.keep;
            }
            Vec4<T>::st(dst + (int64_t)(x0 + orow) * H + col, o);
        }
    }
}

// A wave marches down its rows one dependent load at a time (one row of prefetch): on a large field the other waves of
// the CU hide that, on a small one the chain IS the kernel (256²: 20 iterations ≈ 21 µs).  Fewer rows per wave there:
// more, shorter waves (each re-reads 2R halo rows, which costs nothing when the field fits the L2).  Same bits.
static int rows_per_wave(int W, int H, int replicas) {
    const int64_t cells = (int64_t)W * H * replicas;
    return cells >= (1ll << 23) ? DIF_ROWS : cells >= (1ll << 21) ? 8 : cells >= (1ll << 19) ? 4 : 2;
}

template <typename T, int FUSED, bool WRAP = true>
static int launch_rows(RowsArgs a, int R, hipStream_t s, int replicas = 1) {
    const int strips = (a.H + DIF_WCOLS - 1) / DIF_WCOLS;
    constexpr int WPB = DIF_BLOCK / DIE_WAVE;
    a.rpw = rows_per_wave(a.W, a.H, replicas);
    dim3 grid((strips + WPB - 1) / WPB, (a.W + a.rpw - 1) / a.rpw + (FUSED && a.result ? 1 : 0), replicas);
    switch (R) {
        case 1: k_diffuse_rows<T, 1, FUSED, WRAP><<<grid, DIF_BLOCK, 0, s>>>(a); break;
        case 2: k_diffuse_rows<T, 2, FUSED, WRAP><<<grid, DIF_BLOCK, 0, s>>>(a); break;
        case 3: k_diffuse_rows<T, 3, FUSED, WRAP><<<grid, DIF_BLOCK, 0, s>>>(a); break;
        case 4: k_diffuse_rows<T, 4, FUSED, WRAP><<<grid, DIF_BLOCK, 0, s>>>(a); break;
        default: return DIE_ERR_UNSUPPORTED;
    }
    return DIE_OK;
}

static bool rows_kernel_applies(int W, int H, int R) { return R <= 4 && H % 4 == 0 && H >= 4 && W >= 1; }

static int gaussian_taps(float sigma, double* w) {
    // scipy.ndimage._gaussian_kernel1d: radius int(4σ + .5), exp(−x²/2σ²) normalised
    const int R = (int)(4.0 * (double)sigma + 0.5);
    double sum = 0.0;
    for (int k = -R; k <= R && R <= DIF_MAXR; ++k) { w[k + R] = exp(-0.5 / ((double)sigma * (double)sigma) * k * k); sum += w[k + R]; }
    for (int k = 0; k <= 2 * R && R <= DIF_MAXR; ++k) w[k] /= sum;
    return R;
}

int die_gaussian_taps(float sigma, double* w) { return gaussian_taps(sigma, w); }      // (die_pic.hip)

template <typename T>
static int launch_diffuse(const DiffuseArgs& a, int R, hipStream_t s) {
    dim3 grid((a.H + DIF_TY - 1) / DIF_TY, (a.W + DIF_TX - 1) / DIF_TX);
    switch (R) {
        case 1: k_diffuse<T, 1><<<grid, DIE_BLOCK, 0, s>>>(a); break;
        case 2: k_diffuse<T, 2><<<grid, DIE_BLOCK, 0, s>>>(a); break;
        case 3: k_diffuse<T, 3><<<grid, DIE_BLOCK, 0, s>>>(a); break;
        case 4: k_diffuse<T, 4><<<grid, DIE_BLOCK, 0, s>>>(a); break;
        case 5: k_diffuse<T, 5><<<grid, DIE_BLOCK, 0, s>>>(a); break;
        case 6: k_diffuse<T, 6><<<grid, DIE_BLOCK, 0, s>>>(a); break;
        case 7: k_diffuse<T, 7><<<grid, DIE_BLOCK, 0, s>>>(a); break;
        case 8: k_diffuse<T, 8><<<grid, DIE_BLOCK, 0, s>>>(a); break;
        default: die_set_error("die_diffuse_decay: radius %d not supported (sigma too large)", R); return DIE_ERR_UNSUPPORTED;
    }
    return DIE_OK;
}

static int diffuse_decay_mode(const void* src, void* dst, int32_t W, int32_t H, int32_t dtype, float sigma, float decay,
                              int32_t mode, void* stream) {
    DIE_REQUIRE(src && dst && src != dst, "die_diffuse_decay: src/dst must be distinct non-null planes");
    DIE_REQUIRE(W >= 1 && H >= 1, "die_diffuse_decay: bad size %dx%d", W, H);
    DIE_REQUIRE(sigma > 0.f, "die_diffuse_decay: sigma must be positive");
    DIE_REQUIRE(dtype == DIE_F32 || dtype == DIE_F16, "die_diffuse_decay: bad dtype %d", dtype);
    // scipy.ndimage._gaussian_kernel1d: radius int(4σ + .5), exp(−x²/2σ²) normalised
    const int R = (int)(4.0 * (double)sigma + 0.5);
    DIE_REQUIRE(R >= 1, "die_diffuse_decay: sigma %g gives an empty kernel", (double)sigma);
    if (R > DIF_MAXR) {
        die_set_error("die_diffuse_decay: sigma %g needs radius %d > %d", (double)sigma, R, DIF_MAXR);
        return DIE_ERR_UNSUPPORTED;
    }
    DIE_REQUIRE(mode >= DIE_DIFFUSE_WRAP && mode <= DIE_DIFFUSE_CONSTANT, "die_diffuse_decay: bad boundary mode %d", mode);
    if (mode == DIE_DIFFUSE_WRAP && rows_kernel_applies(W, H, R)) {
        RowsArgs ra;
        double wd[2 * DIF_MAXR + 1];
        gaussian_taps(sigma, wd);
        ra.src = src; ra.dst = dst; ra.claim = nullptr; ra.dep = nullptr; ra.food = nullptr; ra.W = W; ra.H = H; ra.epoch = 0; ra.halo = 0; ra.rep_cells = 0;
        ra.wrapx = ra.wrapy = 1; ra.part_gain = nullptr; ra.part_alive = nullptr; ra.n_part = 0; ra.result = nullptr; ra.alive_const = 0;
        ra.food_infinite = 1; ra.keep = (float)(1.0 - (double)decay); ra.rate_feed = 0.f;
        for (int k = 0; k <= 2 * R; ++k) ra.w[k] = (float)wd[k];
        int rc2 = dtype == DIE_F32 ? launch_rows<float, 0>(ra, R, (hipStream_t)stream)
                                   : launch_rows<__half, 0>(ra, R, (hipStream_t)stream);
        if (rc2 != DIE_OK) return rc2;
        DIE_CHECK_LAUNCH("die_diffuse_decay");
        return DIE_OK;
    }
    DiffuseArgs a;
    a.src = src; a.dst = dst; a.W = W; a.H = H; a.mode = mode;
    a.keep = (float)(1.0 - (double)decay);
    double w[2 * DIF_MAXR + 1], sum = 0.0;
    for (int k = -R; k <= R; ++k) { w[k + R] = exp(-0.5 / ((double)sigma * (double)sigma) * k * k); sum += w[k + R]; }
    for (int k = 0; k <= 2 * R; ++k) a.w[k] = (float)(w[k] / sum);
    int rc = dtype == DIE_F32 ? launch_diffuse<float>(a, R, (hipStream_t)stream)
                              : launch_diffuse<__half>(a, R, (hipStream_t)stream);
    if (rc != DIE_OK) return rc;
    DIE_CHECK_LAUNCH("die_diffuse_decay");
    return DIE_OK;
}

extern "C" int die_diffuse_decay(const void* src, void* dst, int32_t W, int32_t H, int32_t dtype, float sigma,
                                 float decay, void* stream) {
    return diffuse_decay_mode(src, dst, W, H, dtype, sigma, decay, DIE_DIFFUSE_WRAP, stream);
}

extern "C" int die_diffuse_decay_mode(const void* src, void* dst, int32_t W, int32_t H, int32_t dtype, float sigma,
                                      float decay, int32_t mode, void* stream) {
    return diffuse_decay_mode(src, dst, W, H, dtype, sigma, decay, mode, stream);
}

// ---- step driver ----------------------------------------------------------------------
static int step_grid(int64_t N) {
    int64_t g = (N + DIE_STEP_BLOCK - 1) / DIE_STEP_BLOCK;
#ifndef DIE_STEP_GRID_CAP
#define DIE_STEP_GRID_CAP 8192
#endif
    const int64_t cap = DIE_STEP_GRID_CAP;   // ≤ DIE_MAX_PARTIALS partial sums for k_reduce (8192 measured 3 % faster than 2048)
    return (int)(g < cap ? (g > 0 ? g : 1) : cap);
}

// workspace layout: [part_gain_a | part_gain_b | part_alive | scan scratch (die_init) | stash (N floats)]
static const int64_t WS_PARTS = (int64_t)DIE_MAX_PARTIALS * 8 * 3;

int64_t die_ws_scan_bytes(int32_t W, int32_t H);   // die_init.hip

extern "C" int64_t die_workspace_bytes(int32_t W, int32_t H, int64_t N) {
    if (W < 1 || H < 1 || N < 0) return -1;
    int64_t b = WS_PARTS + die_ws_scan_bytes(W, H) + N * (int64_t)sizeof(float);
    return (b + 255) & ~(int64_t)255;
}

static int fill_args(StepArgs& k, const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                     void* ws, int64_t ws_bytes, const char* who) {
    DIE_REQUIRE(m && a && d && ws, "%s: null argument", who);
    DIE_REQUIRE(m->W >= 1 && m->H >= 1 && a->N > 0, "%s: bad sizes", who);
    DIE_REQUIRE((int64_t)a->N <= (int64_t)DIE_OWNER_SLOT_MASK - 1, "%s: too many slots for the ownership word", who);
    DIE_REQUIRE(m->epoch >= 1 && m->epoch <= DIE_OWNER_EPOCH_MAX, "%s: epoch %d outside 1..%d", who, m->epoch,
                DIE_OWNER_EPOCH_MAX);
    DIE_REQUIRE(m->owner && m->food && m->chem && a->x && a->y && a->alive && a->agent_food, "%s: null device pointer", who);
    DIE_REQUIRE(m->dtype == DIE_F32 || m->dtype == DIE_F16, "%s: bad field dtype %d", who, m->dtype);
    DIE_REQUIRE(ws_bytes >= die_workspace_bytes(m->W, m->H, a->N), "%s: workspace too small (%lld < %lld)", who,
                (long long)ws_bytes, (long long)die_workspace_bytes(m->W, m->H, a->N));
    if (d->boundary != DIE_BOUNDARY_WRAP && d->boundary != DIE_BOUNDARY_LIMIT) {
        die_set_error("%s: boundary %d (pass-through) leaves [0,1) and is not representable in Q0.32", who, d->boundary);
        return DIE_ERR_UNSUPPORTED;
    }
    DIE_REQUIRE(d->cost == DIE_COST_LINEAR || d->cost == DIE_COST_ZERO, "%s: bad cost operator %d", who, d->cost);
    if (act) {
        DIE_REQUIRE(act->N == a->N && act->dx && act->dy && act->deposit, "%s: bad action array", who);
        k.dx = act->dx; k.dy = act->dy; k.dep = act->deposit;
    } else {
        k.dx = k.dy = k.dep = nullptr;
    }
    k.g = die_geo_of(m); k.epoch = m->epoch; k.N = a->N;
    k.do_move = 1; k.do_claim = 1; k.tile_w = k.tile_h = k.tiles_y = 1; k.tile_of = nullptr;
    k.owner = (unsigned long long*)m->owner; k.food = m->food; k.chem = m->chem;
    k.x = a->x; k.y = a->y; k.slot = a->slot; k.alive = a->alive; k.agent_food = a->agent_food;
    k.rate_feed = d->rate_feed; k.w_dep = d->cost_w_deposit; k.w_dist = d->cost_w_dist;
    k.boundary = d->boundary; k.cost = d->cost; k.food_infinite = d->food_infinite; k.agents_die = d->agents_die;
    k.has_dead = d->has_dead_slots || d->agents_die;
    k.skip_scatter = 0;
    char* w = (char*)ws;
    k.part_gain = nullptr;
    // ghost-agent tiles: num_alive is the number of alive slots on OWNED cells, counted by the claim pass
    k.part_alive = k.g.own_x1 > 0 ? (long long*)ws + 2 * DIE_MAX_PARTIALS : nullptr;
    k.stash = (float*)(w + WS_PARTS + die_ws_scan_bytes(m->W, m->H));
    return DIE_OK;
}

extern "C" int die_agent_move_claim(const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                                    void* ws, int64_t ws_bytes, void* stream) {
    StepArgs k;
    int rc = fill_args(k, m, a, act, d, ws, ws_bytes, "die_agent_move_claim");
    if (rc != DIE_OK) return rc;
    DIE_REQUIRE(act, "die_agent_move_claim: null action");
    k.part_gain = (long long*)ws;
    const int grid = step_grid(a->N);
    if (m->dtype == DIE_F32) k_move_claim<float><<<grid, DIE_STEP_BLOCK, 0, (hipStream_t)stream>>>(k);
    else k_move_claim<__half><<<grid, DIE_STEP_BLOCK, 0, (hipStream_t)stream>>>(k);
    DIE_CHECK_LAUNCH("die_agent_move_claim");
    return DIE_OK;
}

extern "C" int die_agent_move(const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                              int32_t tile_w, int32_t tile_h, int32_t tiles_y, int32_t* tile_of, void* stream) {
    DIE_REQUIRE(m && a && act && d && tile_of, "die_agent_move: null argument");
    DIE_REQUIRE(tile_w >= 1 && tile_h >= 1 && tiles_y >= 1, "die_agent_move: bad tile shape");
    DIE_REQUIRE(a->N > 0 && act->N == a->N && a->x && a->y && act->dx && act->dy, "die_agent_move: bad arrays");
    if (d->boundary != DIE_BOUNDARY_WRAP && d->boundary != DIE_BOUNDARY_LIMIT) {
        die_set_error("die_agent_move: boundary %d is not representable in Q0.32", d->boundary);
        return DIE_ERR_UNSUPPORTED;
    }
    StepArgs k = {};
    k.g = die_geo_of(m); k.N = a->N; k.x = a->x; k.y = a->y; k.dx = act->dx; k.dy = act->dy;
    k.boundary = d->boundary; k.do_move = 1; k.do_claim = 0;
    k.tile_w = tile_w; k.tile_h = tile_h; k.tiles_y = tiles_y; k.tile_of = tile_of;
    k_move_claim<float><<<step_grid(a->N), DIE_STEP_BLOCK, 0, (hipStream_t)stream>>>(k);
    DIE_CHECK_LAUNCH("die_agent_move");
    return DIE_OK;
}

extern "C" int die_forward_move(const die_medium* m, const die_agents* a, die_gradient_agent* g, const die_action* act,
                                const die_dynamics* d, int32_t tile_w, int32_t tile_h, int32_t tiles_y, int32_t* tile_of,
                                void* stream) {
    DIE_REQUIRE(m && a && g && act && d && tile_of, "die_forward_move: null argument");
    DIE_REQUIRE(tile_w >= 1 && tile_h >= 1 && tiles_y >= 1, "die_forward_move: bad tile shape");
    if (d->boundary != DIE_BOUNDARY_WRAP && d->boundary != DIE_BOUNDARY_LIMIT) {
        die_set_error("die_forward_move: boundary %d is not representable in Q0.32", d->boundary);
        return DIE_ERR_UNSUPPORTED;
    }
    FwdArgs f;
    int rc = die_fill_fwd_args(f, m, a, g, act, "die_forward_move");
    if (rc != DIE_OK) return rc;
    StepArgs k = {};
    k.g = die_geo_of(m); k.N = a->N; k.x = a->x; k.y = a->y; k.slot = a->slot;
    k.boundary = d->boundary; k.do_move = 1; k.do_claim = 0;
    k.tile_w = tile_w; k.tile_h = tile_h; k.tiles_y = tiles_y; k.tile_of = tile_of;
    const int grid = step_grid(a->N);
    hipStream_t s = (hipStream_t)stream;
    if (m->dtype == DIE_F32) {
        if (g->kind == DIE_AGENT_PHYSARUM) k_forward_move_claim<float, DIE_AGENT_PHYSARUM><<<grid, DIE_STEP_BLOCK, 0, s>>>(f, k);
        else k_forward_move_claim<float, DIE_AGENT_GRADIENT><<<grid, DIE_STEP_BLOCK, 0, s>>>(f, k);
    } else {
        if (g->kind == DIE_AGENT_PHYSARUM) k_forward_move_claim<__half, DIE_AGENT_PHYSARUM><<<grid, DIE_STEP_BLOCK, 0, s>>>(f, k);
        else k_forward_move_claim<__half, DIE_AGENT_GRADIENT><<<grid, DIE_STEP_BLOCK, 0, s>>>(f, k);
    }
    DIE_CHECK_LAUNCH("die_forward_move");
    return DIE_OK;
}

extern "C" int die_agent_claim_feed(const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                                    void* ws, int64_t ws_bytes, void* stream) {
    StepArgs k;
    int rc = fill_args(k, m, a, act, d, ws, ws_bytes, "die_agent_claim_feed");
    if (rc != DIE_OK) return rc;
    DIE_REQUIRE(act, "die_agent_claim_feed: null action");
    k.do_move = 0;
    k.part_gain = (long long*)ws;
    const int grid = step_grid(a->N);
    if (m->dtype == DIE_F32) k_move_claim<float><<<grid, DIE_STEP_BLOCK, 0, (hipStream_t)stream>>>(k);
    else k_move_claim<__half><<<grid, DIE_STEP_BLOCK, 0, (hipStream_t)stream>>>(k);
    DIE_CHECK_LAUNCH("die_agent_claim_feed");
    return DIE_OK;
}

extern "C" int die_diffuse_decay_tile(const void* src, void* dst, int32_t W, int32_t H, int32_t dtype, float sigma,
                                      float decay, void* stream) {
    DIE_REQUIRE(src && dst && src != dst, "die_diffuse_decay_tile: src/dst must be distinct non-null planes");
    DIE_REQUIRE(dtype == DIE_F32 || dtype == DIE_F16, "die_diffuse_decay_tile: bad dtype %d", dtype);
    DIE_REQUIRE(sigma > 0.f, "die_diffuse_decay_tile: sigma must be positive");
    const int R = (int)(4.0 * (double)sigma + 0.5);
    if (!(R >= 1 && R <= 4 && H % 4 == 0 && H >= 8 && W >= 2 * R + 1)) {
        die_set_error("die_diffuse_decay_tile: needs H %% 4 == 0, H >= 8, radius 1..4 (W=%d H=%d sigma=%g)", W, H, (double)sigma);
        return DIE_ERR_UNSUPPORTED;
    }
    RowsArgs ra;
    double wd[2 * DIF_MAXR + 1];
    gaussian_taps(sigma, wd);
    ra.src = src; ra.dst = dst; ra.claim = nullptr; ra.dep = nullptr; ra.food = nullptr; ra.W = W; ra.H = H; ra.epoch = 0; ra.halo = 0; ra.rep_cells = 0;
    ra.wrapx = ra.wrapy = 0; ra.part_gain = nullptr; ra.part_alive = nullptr; ra.n_part = 0; ra.result = nullptr; ra.alive_const = 0;
    ra.food_infinite = 1; ra.keep = (float)(1.0 - (double)decay); ra.rate_feed = 0.f;
    for (int k = 0; k <= 2 * R; ++k) ra.w[k] = (float)wd[k];
    int rc = dtype == DIE_F32 ? launch_rows<float, 0, false>(ra, R, (hipStream_t)stream)
                              : launch_rows<__half, 0, false>(ra, R, (hipStream_t)stream);
    if (rc != DIE_OK) return rc;
    DIE_CHECK_LAUNCH("die_diffuse_decay_tile");
    return DIE_OK;
}

extern "C" int die_agent_resolve(const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                                 void* ws, int64_t ws_bytes, void* stream) {
    StepArgs k;
    int rc = fill_args(k, m, a, act, d, ws, ws_bytes, "die_agent_resolve");
    if (rc != DIE_OK) return rc;
    DIE_REQUIRE(act, "die_agent_resolve: null action");
    k.part_gain = (long long*)ws + DIE_MAX_PARTIALS;
    k.part_alive = (long long*)ws + 2 * DIE_MAX_PARTIALS;
    const int grid = step_grid(a->N);
    if (m->dtype == DIE_F32) k_resolve<float><<<grid, DIE_STEP_BLOCK, 0, (hipStream_t)stream>>>(k);
    else k_resolve<__half><<<grid, DIE_STEP_BLOCK, 0, (hipStream_t)stream>>>(k);
    DIE_CHECK_LAUNCH("die_agent_resolve");
    return DIE_OK;
}

extern "C" int die_step_reduce_ex(const die_agents* a, die_step_result* result, void* ws, int64_t ws_bytes,
                                  int32_t with_second_pass, int64_t alive_const, void* stream) {
    DIE_REQUIRE(a && result && ws, "die_step_reduce_ex: null argument");
    DIE_REQUIRE(ws_bytes >= WS_PARTS, "die_step_reduce_ex: workspace too small");
    const int g = step_grid(a->N);
    // with_second_pass: 0 = claim-pass gains, num_alive = alive_const; 1 = + the dead-slot pass's gains and its
    // alive count; 2 = claim-pass gains and the claim pass's count of owned alive slots (ghost-agent tiles);
    // 3 = both passes' gains, num_alive = alive_const (reference-compatible lifecycle: the count is frozen)
    const bool both = with_second_pass == 1 || with_second_pass == 3;
    k_reduce<<<1, 1024, 0, (hipStream_t)stream>>>((const long long*)ws, g, (const long long*)ws + DIE_MAX_PARTIALS, both ? g : 0,
                                                  (const long long*)ws + 2 * DIE_MAX_PARTIALS,
                                                  (with_second_pass == 1 || with_second_pass == 2) ? g : 0, result,
                                                  (with_second_pass == 1 || with_second_pass == 2) ? -1 : alive_const);
    DIE_CHECK_LAUNCH("die_step_reduce_ex");
    return DIE_OK;
}

// _agent_lifecycle (core/env.py:245-250) alone: where(agent_food > 1e-4, agents, 0) on every channel
__global__ __launch_bounds__(DIE_BLOCK) void k_lifecycle(int64_t N, uint32_t* x, uint32_t* y, uint8_t* alive, float* agent_food) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride)
        if (!(agent_food[n] > 1e-4f)) { x[n] = 0; y[n] = 0; alive[n] = 0; agent_food[n] = 0.f; }
}

extern "C" int die_agents_lifecycle(const die_agents* a, void* stream) {
    DIE_REQUIRE(a && a->N > 0 && a->x && a->y && a->alive && a->agent_food, "die_agents_lifecycle: bad agents");
    int64_t g = (a->N + DIE_BLOCK - 1) / DIE_BLOCK;
    k_lifecycle<<<(int)(g < 8192 ? g : 8192), DIE_BLOCK, 0, (hipStream_t)stream>>>(a->N, a->x, a->y, a->alive, a->agent_food);
    DIE_CHECK_LAUNCH("die_agents_lifecycle");
    return DIE_OK;
}

extern "C" int die_step_reduce(const die_agents* a, const die_dynamics* d, die_step_result* result, void* ws,
                               int64_t ws_bytes, void* stream) {
    DIE_REQUIRE(a && d && result && ws, "die_step_reduce: null argument");
    DIE_REQUIRE(ws_bytes >= WS_PARTS, "die_step_reduce: workspace too small");
    const int g = step_grid(a->N);
    k_reduce<<<1, 1024, 0, (hipStream_t)stream>>>((const long long*)ws, g, (const long long*)ws + DIE_MAX_PARTIALS, g,
                                                       (const long long*)ws + 2 * DIE_MAX_PARTIALS, g,
                                                       result, -1);
    DIE_CHECK_LAUNCH("die_step_reduce");
    return DIE_OK;
}

static int deposit_feed_diffuse(const die_medium* m, const die_dynamics* d, int halo, bool tile, void* stream, const char* who,
                                const long long* part_gain = nullptr, int n_part = 0, die_step_result* result = nullptr,
                                long long alive_const = 0, const long long* part_alive = nullptr, const float* dep_plane = nullptr);
extern "C" int die_agent_dead_slots(const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                                    void* ws, int64_t ws_bytes, void* stream);

static bool fused_step_applies(const die_medium* m, const die_dynamics* d) {
    const int R = (int)(4.0 * (double)d->diffuse_sigma + 0.5);
    return m->gW <= 0 && d->diffuse_mode == DIE_DIFFUSE_WRAP && rows_kernel_applies(m->W, m->H, R) && R >= 1 && !d->staged;
}

bool fused_step_shape_ok(const die_medium* m, const die_dynamics* d) { return fused_step_applies(m, d); }      // (die_pic.hip)

// everything of die_env_step after the claims are in place (fused path)
static int env_step_tail(const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                         die_step_result* result, void* ws, int64_t ws_bytes, void* stream) {
    // deposits and feeding ride on the diffusion sweep; the per-agent second pass is only needed for
    // dead slots / lifecycle (it then skips the winner's scatter)
    const bool second_pass = d->has_dead_slots || d->agents_die;
    if (second_pass) {
        int rc = die_agent_dead_slots(m, a, act, d, ws, ws_bytes, stream);
        if (rc != DIE_OK) return rc;
    }
    const int g = step_grid(a->N);
    if (!second_pass)                                       // an extra workgroup of the sweep does the reduction (same summation order)
        return deposit_feed_diffuse(m, d, 0, false, stream, "die_env_step(diffuse+deposit+feed+reduce)", (const long long*)ws, g,
                                    result, a->N);
    k_reduce<<<1, 1024, 0, (hipStream_t)stream>>>((const long long*)ws, g, (const long long*)ws + DIE_MAX_PARTIALS,
                                                  g, (const long long*)ws + 2 * DIE_MAX_PARTIALS, g, result, -1);
    DIE_CHECK_LAUNCH("die_env_step(reduce)");
    return deposit_feed_diffuse(m, d, 0, false, stream, "die_env_step(diffuse+deposit+feed)");
}

extern "C" int die_env_step(const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                            die_step_result* result, void* ws, int64_t ws_bytes, void* stream) {
    DIE_REQUIRE(m && a && act && d && result, "die_env_step: null argument");
    DIE_REQUIRE(m->chem_next && m->chem_next != m->chem, "die_env_step: chem_next must be a second plane");
    int rc = die_agent_move_claim(m, a, act, d, ws, ws_bytes, stream);
    if (rc != DIE_OK) return rc;
    if (fused_step_applies(m, d)) return env_step_tail(m, a, act, d, result, ws, ws_bytes, stream);
    rc = die_agent_resolve(m, a, act, d, ws, ws_bytes, stream);
    if (rc != DIE_OK) return rc;
    rc = die_step_reduce(a, d, result, ws, ws_bytes, stream);
    if (rc != DIE_OK) return rc;
    if (m->gW > 0) {
        die_set_error("die_env_step: a decomposed tile needs halo exchange and migration between the stages; "
                      "drive die_agent_move / die_agent_claim_feed / die_medium_deposit_feed_diffuse_tile instead");
        return DIE_ERR_UNSUPPORTED;
    }
    return diffuse_decay_mode(m->chem, m->chem_next, m->W, m->H, m->dtype, d->diffuse_sigma, d->rate_decay_chem, d->diffuse_mode,
                              stream);
}

static int forward_move_claim(const die_medium* m, const die_agents* a, die_gradient_agent* g, const die_action* act,
                              const die_dynamics* d, void* ws, int64_t ws_bytes, void* stream, bool tile_ok, const char* who) {
    DIE_REQUIRE(m && a && g && act && d, "%s: null argument", who);
    if (!tile_ok && !fused_step_applies(m, d)) {   // first half of the fused step: refuse before touching anything
        die_set_error("%s: only for periodic planes with H %% 4 == 0 and gaussian radius 1..4", who);
        return DIE_ERR_UNSUPPORTED;
    }
    FwdArgs f;
    int rc = die_fill_fwd_args(f, m, a, g, act, who);
    if (rc != DIE_OK) return rc;
    StepArgs k;
    rc = fill_args(k, m, a, act, d, ws, ws_bytes, who);
    if (rc != DIE_OK) return rc;
    k.part_gain = (long long*)ws;
    const int grid = step_grid(a->N);
    hipStream_t s = (hipStream_t)stream;
    // the sense-mask and ownership tests are compiled out of the plain single-tile kernel (they cost ≈ 5 % there)
    const bool ext = f.mask != nullptr || k.g.own_x1 > 0;
#define DIE_FMC(T, KIND, LEAN) do { if (ext) k_forward_move_claim<T, KIND, true, LEAN><<<grid, DIE_STEP_BLOCK, 0, s>>>(f, k); \
                                    else k_forward_move_claim<T, KIND, false, LEAN><<<grid, DIE_STEP_BLOCK, 0, s>>>(f, k); } while (0)
    const bool lean = fwd_is_lean(g);
    if (m->dtype == DIE_F32) {
        if (g->kind != DIE_AGENT_PHYSARUM) DIE_FMC(float, DIE_AGENT_GRADIENT, false);
        else if (lean) DIE_FMC(float, DIE_AGENT_PHYSARUM, true);
        else DIE_FMC(float, DIE_AGENT_PHYSARUM, false);
    } else {
        if (g->kind != DIE_AGENT_PHYSARUM) DIE_FMC(__half, DIE_AGENT_GRADIENT, false);
        else if (lean) DIE_FMC(__half, DIE_AGENT_PHYSARUM, true);
        else DIE_FMC(__half, DIE_AGENT_PHYSARUM, false);
    }
#undef DIE_FMC
    DIE_CHECK_LAUNCH(who);
    return DIE_OK;
}

extern "C" int die_forward_move_claim(const die_medium* m, const die_agents* a, die_gradient_agent* g, const die_action* act,
                                      const die_dynamics* d, void* ws, int64_t ws_bytes, void* stream) {
    return forward_move_claim(m, a, g, act, d, ws, ws_bytes, stream, false, "die_forward_move_claim");
}

extern "C" int die_forward_move_claim_tile(const die_medium* m, const die_agents* a, die_gradient_agent* g,
                                           const die_action* act, const die_dynamics* d, void* ws, int64_t ws_bytes, void* stream) {
    DIE_REQUIRE(m && m->gW > 0, "die_forward_move_claim_tile: the planes must be a tile of a decomposed world");
    return forward_move_claim(m, a, g, act, d, ws, ws_bytes, stream, true, "die_forward_move_claim_tile");
}

extern "C" int die_env_step_finish(const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                                   die_step_result* result, void* ws, int64_t ws_bytes, void* stream) {
    DIE_REQUIRE(m && a && act && d && result && ws, "die_env_step_finish: null argument");
    if (!fused_step_applies(m, d)) {
        die_set_error("die_env_step_finish: only for periodic planes with H %% 4 == 0 and gaussian radius 1..4");
        return DIE_ERR_UNSUPPORTED;
    }
    return env_step_tail(m, a, act, d, result, ws, ws_bytes, stream);
}

extern "C" int die_forward_env_step(const die_medium* m, const die_agents* a, die_gradient_agent* g, const die_action* act,
                                    const die_dynamics* d, die_step_result* result, void* ws, int64_t ws_bytes,
                                    void* stream) {
    DIE_REQUIRE(m && a && g && act && d && result, "die_forward_env_step: null argument");
    DIE_REQUIRE(m->chem_next && m->chem_next != m->chem, "die_forward_env_step: chem_next must be a second plane");
    if (!fused_step_applies(m, d)) {
        die_set_error("die_forward_env_step: only for periodic planes with H %% 4 == 0 and gaussian radius 1..4");
        return DIE_ERR_UNSUPPORTED;
    }
    int rc = die_forward_move_claim(m, a, g, act, d, ws, ws_bytes, stream);
    if (rc != DIE_OK) return rc;
    return env_step_tail(m, a, act, d, result, ws, ws_bytes, stream);
}

extern "C" int64_t die_batch_workspace_bytes(int32_t replicas) {
    return replicas >= 1 && replicas <= DIE_MAX_REPLICAS ? (int64_t)replicas * WS_PARTS : -1;
}

extern "C" int die_forward_env_step_batch(const die_medium* m, const die_agents* a, die_gradient_agent* g, const die_action* act,
                                          const die_dynamics* d, const die_batch* b, die_step_result* results, void* ws,
                                          int64_t ws_bytes, void* stream) {
    const char* who = "die_forward_env_step_batch";
    DIE_REQUIRE(m && a && g && d && b && results && ws, "%s: null argument", who);
    DIE_REQUIRE(b->replicas >= 1 && b->replicas <= DIE_MAX_REPLICAS, "%s: 1..%d replicas", who, DIE_MAX_REPLICAS);
    DIE_REQUIRE(ws_bytes >= die_batch_workspace_bytes(b->replicas), "%s: workspace too small", who);
    DIE_REQUIRE(m->gW <= 0 && !m->sense_mask && !d->has_dead_slots && !d->agents_die && !d->staged,
                "%s: periodic single-tile replicas with every slot alive (no agents_die, no sense mask)", who);
    DIE_REQUIRE(m->chem_next && m->chem_next != m->chem, "%s: chem_next must be a second plane", who);
    DIE_REQUIRE(b->plane_stride >= (int64_t)m->W * m->H && b->agent_stride >= a->N, "%s: strides smaller than a replica", who);
    if (!fused_step_applies(m, d)) {
        die_set_error("%s: only for periodic planes with H %% 4 == 0 and gaussian radius 1..4", who);
        return DIE_ERR_UNSUPPORTED;
    }
    FwdArgs f;
    int rc = die_fill_fwd_args(f, m, a, g, act, who);
    if (rc != DIE_OK) return rc;
    // fill_args checks the per-replica workspace of the single-world step; here only the partial arrays are used
    DIE_REQUIRE(m->epoch >= 1 && m->epoch <= DIE_OWNER_EPOCH_MAX && m->owner && a->alive && a->agent_food, "%s: bad medium / agents", who);
    if (d->boundary != DIE_BOUNDARY_WRAP && d->boundary != DIE_BOUNDARY_LIMIT) {
        die_set_error("%s: boundary %d is not representable in Q0.32", who, d->boundary);
        return DIE_ERR_UNSUPPORTED;
    }
    DIE_REQUIRE(d->cost == DIE_COST_LINEAR || d->cost == DIE_COST_ZERO, "%s: bad cost operator %d", who, d->cost);
    StepArgs k = {};
    k.g = die_geo_of(m); k.epoch = m->epoch; k.N = a->N; k.do_move = 1; k.do_claim = 1; k.tile_w = k.tile_h = k.tiles_y = 1;
    k.owner = (unsigned long long*)m->owner; k.food = m->food; k.chem = m->chem;
    k.x = a->x; k.y = a->y; k.slot = a->slot; k.alive = a->alive; k.agent_food = a->agent_food;
    k.dx = act ? act->dx : nullptr; k.dy = act ? act->dy : nullptr; k.dep = act ? act->deposit : nullptr;
    k.rate_feed = d->rate_feed; k.w_dep = d->cost_w_deposit; k.w_dist = d->cost_w_dist;
    k.boundary = d->boundary; k.cost = d->cost; k.food_infinite = d->food_infinite;
    k.part_gain = (long long*)ws;
    BatchArgs ba;
    ba.cells = b->plane_stride; ba.agents = b->agent_stride; ba.seed_stride = b->seed_stride;
    int64_t nmax = 0;
    for (int r = 0; r < DIE_MAX_REPLICAS; ++r) {
        ba.n[r] = r < b->replicas ? b->n[r] : 0;
        DIE_REQUIRE(r >= b->replicas || (b->n[r] >= 1 && b->n[r] <= b->agent_stride), "%s: replica %d has %lld agents", who, r, (long long)b->n[r]);
        if (ba.n[r] > nmax) nmax = ba.n[r];
    }
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(step_grid(nmax), b->replicas);
    if (m->dtype == DIE_F32) {
        if (g->kind != DIE_AGENT_PHYSARUM) k_forward_move_claim_batch<float, DIE_AGENT_GRADIENT><<<grid, DIE_STEP_BLOCK, 0, s>>>(f, k, ba);
        else if (fwd_is_lean(g)) k_forward_move_claim_batch<float, DIE_AGENT_PHYSARUM, true><<<grid, DIE_STEP_BLOCK, 0, s>>>(f, k, ba);
        else k_forward_move_claim_batch<float, DIE_AGENT_PHYSARUM><<<grid, DIE_STEP_BLOCK, 0, s>>>(f, k, ba);
    } else {
        if (g->kind != DIE_AGENT_PHYSARUM) k_forward_move_claim_batch<__half, DIE_AGENT_GRADIENT><<<grid, DIE_STEP_BLOCK, 0, s>>>(f, k, ba);
        else if (fwd_is_lean(g)) k_forward_move_claim_batch<__half, DIE_AGENT_PHYSARUM, true><<<grid, DIE_STEP_BLOCK, 0, s>>>(f, k, ba);
        else k_forward_move_claim_batch<__half, DIE_AGENT_PHYSARUM><<<grid, DIE_STEP_BLOCK, 0, s>>>(f, k, ba);
    }
    DIE_CHECK_LAUNCH(who);
    // the field sweep of every replica in one launch (gridDim.z), each with its own reduction workgroup
    const int R = (int)(4.0 * (double)d->diffuse_sigma + 0.5);
    RowsArgs ra;
    double wd[2 * DIF_MAXR + 1];
    gaussian_taps(d->diffuse_sigma, wd);
    ra.src = m->chem; ra.dst = m->chem_next; ra.claim = (const unsigned long long*)m->owner; ra.dep = nullptr; ra.food = m->food;
    ra.W = m->W; ra.H = m->H; ra.epoch = m->epoch; ra.food_infinite = d->food_infinite; ra.halo = 0; ra.rep_cells = b->plane_stride;
    ra.wrapx = ra.wrapy = 1;
    ra.part_gain = (const long long*)ws; ra.part_alive = (const long long*)ws + 2 * DIE_MAX_PARTIALS; ra.n_part = (int)grid.x;
    ra.result = results; ra.alive_const = 0;
    ra.keep = (float)(1.0 - (double)d->rate_decay_chem); ra.rate_feed = d->rate_feed;
    for (int q = 0; q <= 2 * R; ++q) ra.w[q] = (float)wd[q];
    rc = m->dtype == DIE_F32 ? launch_rows<float, 1, true>(ra, R, s, b->replicas) : launch_rows<__half, 1, true>(ra, R, s, b->replicas);
    if (rc != DIE_OK) return rc;
    DIE_CHECK_LAUNCH(who);
    return DIE_OK;
}

static int deposit_feed_diffuse(const die_medium* m, const die_dynamics* d, int halo, bool tile, void* stream,
                                const char* who, const long long* part_gain, int n_part, die_step_result* result,
                                long long alive_const, const long long* part_alive, const float* dep_plane) {
    DIE_REQUIRE(m && d, "%s: null argument", who);
    DIE_REQUIRE((m->owner || dep_plane) && m->food && m->chem && m->chem_next && m->chem_next != m->chem, "%s: null or aliased plane", who);
    DIE_REQUIRE(m->dtype == DIE_F32 || m->dtype == DIE_F16, "%s: bad dtype %d", who, m->dtype);
    DIE_REQUIRE(dep_plane || (m->epoch >= 1 && m->epoch <= DIE_OWNER_EPOCH_MAX), "%s: bad epoch %d", who, m->epoch);
    const int R = (int)(4.0 * (double)d->diffuse_sigma + 0.5);
    if (!(rows_kernel_applies(m->W, m->H, R) && R >= 1 && (!tile || (m->H >= 8 && m->W >= 2 * R + 1 && halo >= 0)))) {
        die_set_error("%s: needs H %% 4 == 0 and radius 1..4 (W=%d H=%d sigma=%g halo=%d)", who, m->W, m->H,
                      (double)d->diffuse_sigma, halo);
        return DIE_ERR_UNSUPPORTED;
    }
    RowsArgs ra;
    double wd[2 * DIF_MAXR + 1];
    gaussian_taps(d->diffuse_sigma, wd);
    ra.src = m->chem; ra.dst = m->chem_next; ra.claim = (const unsigned long long*)m->owner; ra.dep = dep_plane; ra.food = m->food;
    ra.W = m->W; ra.H = m->H; ra.epoch = m->epoch; ra.food_infinite = d->food_infinite; ra.halo = halo; ra.rep_cells = 0;
    ra.wrapx = tile && m->gW > 0 && m->W == m->gW; ra.wrapy = tile && m->gW > 0 && m->H == m->gH;
    ra.part_gain = part_gain; ra.part_alive = part_alive; ra.n_part = n_part; ra.result = result; ra.alive_const = alive_const;
    ra.keep = (float)(1.0 - (double)d->rate_decay_chem); ra.rate_feed = d->rate_feed;
    for (int k = 0; k <= 2 * R; ++k) ra.w[k] = (float)wd[k];
    int rc;
    if (dep_plane) {
        DIE_REQUIRE(!tile, "%s: the deposit-plane sweep is for periodic single-tile planes", who);
        rc = m->dtype == DIE_F32 ? launch_rows<float, 2, true>(ra, R, (hipStream_t)stream)
                                 : launch_rows<__half, 2, true>(ra, R, (hipStream_t)stream);
    }
    else if (tile) rc = m->dtype == DIE_F32 ? launch_rows<float, 1, false>(ra, R, (hipStream_t)stream)
                                            : launch_rows<__half, 1, false>(ra, R, (hipStream_t)stream);
    else rc = m->dtype == DIE_F32 ? launch_rows<float, 1, true>(ra, R, (hipStream_t)stream)
                                  : launch_rows<__half, 1, true>(ra, R, (hipStream_t)stream);
    if (rc != DIE_OK) return rc;
    DIE_CHECK_LAUNCH(who);
    return DIE_OK;
}

// the field half of the tile-binned step (die_pic.hip): the sweep reads the winners' deposits from a float plane
int die_sweep_dep_plane(const die_medium* m, const die_dynamics* d, const float* dep_plane, const long long* part_gain, int n_part,
                        die_step_result* result, long long alive_const, void* stream) {
    DIE_REQUIRE(dep_plane, "die_pic_step: null deposit plane");
    if (!fused_step_applies(m, d)) {
        die_set_error("die_pic_step: only for periodic planes with H %% 4 == 0 and gaussian radius 1..4");
        return DIE_ERR_UNSUPPORTED;
    }
    return deposit_feed_diffuse(m, d, 0, false, stream, "die_pic_step(diffuse+deposit+feed+reduce)", part_gain, n_part, result,
                                alive_const, nullptr, dep_plane);
}

extern "C" int die_medium_deposit_feed_diffuse(const die_medium* m, const die_dynamics* d, void* stream) {
    DIE_REQUIRE(m && m->gW <= 0, "die_medium_deposit_feed_diffuse: periodic single-tile planes only "
                                 "(decomposed tiles: die_medium_deposit_feed_diffuse_tile)");
    return deposit_feed_diffuse(m, d, 0, false, stream, "die_medium_deposit_feed_diffuse");
}

extern "C" int die_medium_deposit_feed_diffuse_tile(const die_medium* m, const die_dynamics* d, int32_t halo, void* stream) {
    return deposit_feed_diffuse(m, d, halo, true, stream, "die_medium_deposit_feed_diffuse_tile");
}

extern "C" int die_tile_sweep_reduce(const die_medium* m, const die_agents* a, const die_dynamics* d, int32_t halo,
                                     die_step_result* result, void* ws, int64_t ws_bytes, void* stream) {
    DIE_REQUIRE(m && a && d && result && ws, "die_tile_sweep_reduce: null argument");
    DIE_REQUIRE(ws_bytes >= WS_PARTS, "die_tile_sweep_reduce: workspace too small");
    DIE_REQUIRE(!(d->has_dead_slots || d->agents_die), "die_tile_sweep_reduce: a dead-slot pass needs die_step_reduce_ex");
    const bool counted = m->gW > 0 && m->own_x1 > 0;       // ghost tiles: the claim pass counted the owned alive slots
    return deposit_feed_diffuse(m, d, halo, true, stream, "die_tile_sweep_reduce", (const long long*)ws, step_grid(a->N), result,
                                a->N, counted ? (const long long*)ws + 2 * DIE_MAX_PARTIALS : nullptr);
}

extern "C" int die_agent_dead_slots(const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                                    void* ws, int64_t ws_bytes, void* stream) {
    StepArgs k;
    int rc = fill_args(k, m, a, act, d, ws, ws_bytes, "die_agent_dead_slots");
    if (rc != DIE_OK) return rc;
    DIE_REQUIRE(act, "die_agent_dead_slots: null action");
    k.part_gain = (long long*)ws + DIE_MAX_PARTIALS;
    k.part_alive = (long long*)ws + 2 * DIE_MAX_PARTIALS;
    k.skip_scatter = 1;
    const int grid = step_grid(a->N);
    if (m->dtype == DIE_F32) k_resolve<float><<<grid, DIE_STEP_BLOCK, 0, (hipStream_t)stream>>>(k);
    else k_resolve<__half><<<grid, DIE_STEP_BLOCK, 0, (hipStream_t)stream>>>(k);
    DIE_CHECK_LAUNCH("die_agent_dead_slots");
    return DIE_OK;
}

// ---- error plumbing ---------------------------------------------------------------------
static thread_local char g_err[512] = "";

void die_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* die_last_error(void) { return g_err; }
extern "C" int die_abi_version(void) { return DIE_ABI_VERSION; }
