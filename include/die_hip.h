/* die_hip.h — C ABI of libdie_hip.so: the MI355X (gfx950) implementation of the grid-update
 * hot path of gkirgizov/die (Env.step field dynamics + Agent.forward + data_init allocation).
 *
 * The reference has no FFI: the boundary it offers is the Python class API of core/env.py,
 * core/agent/ (all modules) and core/data_init.py.  The host side of this project
 * (the die_amd package) keeps those classes and calls the entry points below through ctypes; each
 * entry point names the reference function(s) it replaces.  Paths are relative to the
 * reference checkout.
 *
 * Conventions
 *   - plain C, no exceptions; every call returns DIE_OK (0) or a negative die_status and
 *     leaves a message for die_last_error() (thread-local).
 *   - all pointers in the structs are DEVICE pointers (HBM) owned by the caller; the library
 *     allocates nothing and keeps no state.  `stream` is a hipStream_t (NULL = default
 *     stream); every call only enqueues work on it and returns without synchronising.
 *   - field planes are W×H row-major, element (ix, iy) at ix*H + iy — the (channel, x, y)
 *     layout of core/data_init.py:95-112 with one plane per channel.
 *   - agent coordinates are Q0.32 fixed point: x = X / 2^32 in [0, 1).  The nearest-cell
 *     lookup of core/utils.py:39-54 is then exact integer arithmetic,
 *     cell = clamp((X*(W-1) + 2^31) >> 32, 0, W-1), and the `% 1.` wrap of
 *     core/env.py:155 is the natural 32-bit overflow.
 *   - the 'agents' medium channel (core/base_types.py:32) is held as a 64-bit claim per cell:
 *     high word (epoch << 27) | (slot + 1) of the highest-index alive agent standing on the
 *     cell in step `epoch`, low word the fp32 bits of that agent's deposit.  A cell is occupied
 *     iff claim >> 59 == current epoch (epoch in 1..31, the caller zeroes the plane when it
 *     wraps: once in 31 steps; 27 bits hold up to 134 M world slots).  One 64-bit atomicMax per agent: highest slot wins == the "last writer wins" of
 *     core/env.py:211, and the winner's deposit reaches the diffusion sweep without a second
 *     per-agent pass.
 */
#ifndef DIE_HIP_H
#define DIE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIE_ABI_VERSION 23

typedef enum die_status {
    DIE_OK = 0,
    DIE_ERR_ARG = -1,         /* null pointer, bad size, unsupported enum value */
    DIE_ERR_HIP = -2,         /* a HIP runtime call or kernel launch failed */
    DIE_ERR_UNSUPPORTED = -3  /* valid in the reference, not implemented on device */
} die_status;

typedef enum die_dtype { DIE_F32 = 0, DIE_F16 = 1 } die_dtype;

/* core/env.py:24-26 BoundaryCondition; anything else passes coordinates through (:158-161) */
typedef enum die_boundary { DIE_BOUNDARY_WRAP = 0, DIE_BOUNDARY_LIMIT = 1, DIE_BOUNDARY_NONE = 2 } die_boundary;

/* core/env.py:29-39 action-cost operators available on device */
typedef enum die_cost { DIE_COST_LINEAR = 0, DIE_COST_ZERO = 1 } die_cost;

/* core/agent/gradient.py: which forward() a call computes */
typedef enum die_agent_kind { DIE_AGENT_GRADIENT = 0, DIE_AGENT_PHYSARUM = 1 } die_agent_kind;

#define DIE_OWNER_EPOCH_SHIFT 27
#define DIE_OWNER_EPOCH_MAX 31
#define DIE_OWNER_SLOT_MASK 0x07FFFFFFu

/* The (3, W, H) medium of core/env.py:75-79 / core/base_types.py:32. */
typedef struct die_medium {
    int32_t W, H;
    int32_t dtype;       /* die_dtype of food / chem / chem_next */
    int32_t epoch;       /* current ownership epoch, 1..DIE_OWNER_EPOCH_MAX */
    uint64_t* owner;     /* 'agents' channel, W*H claim words (see header comment) */
    void* food;          /* 'env_food', W*H */
    void* chem;          /* 'chem1', W*H, current */
    void* chem_next;     /* W*H, receives the diffused plane; the caller swaps after die_env_step */
    /* Domain decomposition.  gW == 0: the planes ARE the periodic W×H world.
     * gW > 0: the planes are one rank's tile of a gW×gH world, halo included: local element
     * (i, j) is global cell (ox + i, oy + j); agent coordinates stay global; nothing wraps
     * inside the tile (the halo is filled by the caller's exchange). */
    int32_t gW, gH;
    int32_t ox, oy;
    /* Ghost-agent decomposition: a rank also steps copies of its neighbours' agents that stand in
     * its halo, so reward / num_alive must only count the agents standing on the cells this rank accounts for:
     * local rows [own_x0, own_x1) × columns [own_y0, own_y1).  own_x1 == 0: every cell counts.
     * An axis with W == gW (one rank along it) has no halo and is periodic inside the tile. */
    int32_t own_x0, own_y0, own_x1, own_y1;
    /* Dynamics.apply_sense_mask (core/env.py:276-295): W*H bytes from die_sense_mask; the forward kernels read the
     * chem / food of a cell with mask 0 as 0.  NULL: everything is visible. */
    const uint8_t* sense_mask;
} die_medium;

/* The (4, N) agent array of core/data_init.py:114-150, structure of arrays. */
typedef struct die_agents {
    int64_t N;           /* slots (alive or not) */
    uint32_t* x;         /* Q0.32 */
    uint32_t* y;         /* Q0.32 */
    uint8_t* alive;      /* 1 / 0 */
    float* agent_food;
    uint32_t* slot;      /* reference slot id of each array entry, or NULL = identity.  The arrays may be
                            held in any order (die_agents_sort keeps them spatially sorted); Philox
                            counters and ownership words always use the slot id, so results do not
                            depend on the order. */
} die_agents;

/* The (3, N) action array of core/data_init.py:152-157. */
typedef struct die_action {
    int64_t N;
    float* dx;
    float* dy;
    float* deposit;
} die_action;

/* core/env.py:42-61 Dynamics, the fields the device path honours. */
/* scipy.ndimage boundary modes as skimage.filters.gaussian passes them through (core/env.py:140-143) */
typedef enum die_diffuse_mode { DIE_DIFFUSE_WRAP = 0, DIE_DIFFUSE_NEAREST = 1, DIE_DIFFUSE_REFLECT = 2, DIE_DIFFUSE_MIRROR = 3,
                                DIE_DIFFUSE_CONSTANT = 4 } die_diffuse_mode;

typedef struct die_dynamics {
    float rate_feed;
    float rate_decay_chem;
    float diffuse_sigma;
    int32_t boundary;        /* die_boundary */
    int32_t cost;            /* die_cost */
    float cost_w_deposit;    /* 0.02 (core/env.py:29) */
    float cost_w_dist;       /* 0.01 */
    int32_t food_infinite;
    int32_t agents_die;
    int32_t has_dead_slots;  /* 0: caller guarantees every slot is alive (skips the dead-slot feed pass) */
    int32_t diffuse_mode;    /* die_diffuse_mode (Dynamics.diffuse_mode, core/env.py:49,142); only WRAP takes the fused sweep */
    int32_t staged;          /* 1: die_env_step runs its stages one kernel each (move/claim, resolve, reduce, diffuse) even where
                                the fused field sweep applies — same bits, for cross-checks (tests) */
} die_dynamics;

/* core/agent/gradient.py:19-28,139-151 constructor arguments + per-agent state. */
typedef struct die_gradient_agent {
    int32_t kind;            /* die_agent_kind */
    int32_t normalized_grad;
    float scale;
    float deposit;
    float inertia;
    float sense_offset;
    float noise_scale;
    float grad_clip;         /* < 0: None */
    /* The heading and the turn decision are float64, like the reference's (core/agent/gradient.py:168-208): the decision
     * `abs(dir_delta) > sense_radians` is an exact tie whenever a probe sits on the symmetry axis of a deposit, and which way
     * it falls is decided by the low bits of the heading — fp32 headings resolved ~0.1 % of agents differently per step. */
    double turn_radians;     /* physarum only: np.radians(turn_angle) */
    double sense_radians;
    double turn_tolerance;
    uint32_t* heading_hi;    /* N + N words: _direction_rads as float64, high and low halves in two arrays (state, read and */
    uint32_t* heading_lo;    /* written) — two 4-byte arrays travel through the sort / migration machinery like any other */
    float* prev_gx;          /* N, _prev_grad[0]; may be NULL when inertia == 0 */
    float* prev_gy;          /* N */
    const int8_t* turn_sign; /* N entries ±1 (indexed by SLOT id) replacing the random turn, or NULL → Philox(seed, step, slot) */
    uint64_t seed;
    uint32_t step;           /* forward-call counter, the Philox step word */
    uint32_t reserved2;
    const uint32_t* step_base; /* device word added to `step`, or NULL: lets a captured hipGraph of many steps be replayed
                                  (the launch arguments are frozen, the counter is not) */
} die_gradient_agent;

/* Device-resident result of one die_env_step (read it back after synchronising). */
typedef struct die_step_result {
    double reward;           /* sum of `gained` over every slot (core/env.py:119-120), accumulated in 32.32 fixed point:
                                identical bits whatever the order of the agent arrays or the decomposition */
    int64_t num_alive;       /* core/env.py:263-265 */
} die_step_result;

int die_abi_version(void);
const char* die_last_error(void);

/* Bytes of scratch die_env_step / die_init_agents need for this problem size. */
int64_t die_workspace_bytes(int32_t W, int32_t H, int64_t N);

/* ---- Agent.forward ------------------------------------------------------------------
 * GradientAgent.forward / PhysarumAgent.forward (core/agent/gradient.py:96-124, 168-219):
 * one forward probe of the normalised np.gradient of chem1 (4 taps around the probe cell,
 * :55-76), discrete turn, momentum, deposit; writes the action and updates agent state.
 * Every slot acts, alive or not, as in the reference. */
int die_gradient_forward(const die_medium* m, const die_agents* a, die_gradient_agent* g,
                         die_action* out, void* stream);

/* GradientAgent.render (core/agent/gradient.py:126-135): rgb_out (W, H, 3) fp32 = 0.5 * (stack(gx, gy, 0) + 1) of the
 * normalised, clipped np.gradient field of the chem plane (_get_gradient, :55-71). */
int die_gradient_render(const void* chem, int32_t W, int32_t H, int32_t dtype, int32_t normalized, float grad_clip,
                        float* rgb_out, void* stream);

/* BrownianAgent.forward (core/agent/static.py:40-50): (b-a)*u.round(3)+a per channel, × alive. */
int die_brownian_forward(const die_agents* a, float move_scale, float deposit_scale,
                         uint64_t seed, uint32_t step, die_action* out, void* stream);

/* ConstAgent.forward (core/agent/static.py:20-27). */
int die_const_forward(int64_t N, float dx, float dy, float deposit, die_action* out, void* stream);

/* ---- Env.step -------------------------------------------------------------------------
 * core/env.py:101-131 in one call: _agent_move (:163-172), _agent_deposit_and_layout
 * (:204-215), _agent_feed (:220-243), _agent_lifecycle (:245-261, agents_die only),
 * _medium_diffuse_decay (:136-145) and the reward / num_agents reductions (:117-126).
 * `m->epoch` must already be the new step's epoch.  On return (stream order) the diffused
 * plane is in m->chem_next.  `result` is a device pointer. */
int die_env_step(const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                 die_step_result* result, void* workspace, int64_t workspace_bytes, void* stream);

/* Agent.forward + Env.step in one call for the common loop `env.step(agent.forward(obs))`
 * (examples/minimal_run.py:24-25): GradientAgent/PhysarumAgent.forward is fused with the move /
 * claim / feeding pass, so the action is written to `act` for the caller but never read back and
 * the agent coordinates are loaded once.  Same results as die_gradient_forward followed by
 * die_env_step.  DIE_ERR_UNSUPPORTED when the fused field sweep does not apply (decomposed tile,
 * H % 4 != 0, gaussian radius > 4): call the two separately then. */
int die_forward_env_step(const die_medium* m, const die_agents* a, die_gradient_agent* g, const die_action* act,
                         const die_dynamics* d, die_step_result* result, void* workspace, int64_t workspace_bytes,
                         void* stream);

/* R replicas of one world shape stepped in ONE launch pair (BASELINE configs[4]: batched env replicas; small grids are
 * launch-bound one at a time).  `m`, `a`, `g`, `act` describe replica 0; every array of replica r starts r strides further
 * (planes: plane_stride elements, per-agent arrays: agent_stride elements; results[r]); replica r holds n[r] agents (all
 * alive) and draws from Philox key g->seed + r * seed_stride — so replica r computes exactly what a stand-alone world with
 * that seed computes.  `act` may be NULL (no action written).  Same restrictions as die_forward_env_step. */
#define DIE_MAX_REPLICAS 64
typedef struct die_batch {
    int32_t replicas;
    int32_t reserved;
    int64_t plane_stride;
    int64_t agent_stride;
    uint64_t seed_stride;
    int64_t n[DIE_MAX_REPLICAS];
} die_batch;
int64_t die_batch_workspace_bytes(int32_t replicas);
int die_forward_env_step_batch(const die_medium* m, const die_agents* a, die_gradient_agent* g, const die_action* act,
                               const die_dynamics* d, const die_batch* b, die_step_result* results, void* workspace,
                               int64_t workspace_bytes, void* stream);

/* First kernel of die_forward_env_step alone (forward + move + claim + feeding of alive slots). */
int die_forward_move_claim(const die_medium* m, const die_agents* a, die_gradient_agent* g, const die_action* act,
                           const die_dynamics* d, void* workspace, int64_t workspace_bytes, void* stream);

/* Everything of die_forward_env_step / die_env_step after the claims are in place (dead-slot pass if
 * needed, reduction, fused field sweep) — lets a caller put work that only touches the agent arrays
 * (die_agents_sort) on another stream, next to the sweep. */
int die_env_step_finish(const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                        die_step_result* result, void* workspace, int64_t workspace_bytes, void* stream);

/* The same kernel on a tile of a decomposed world (claims may land in the halo; the caller merges them
 * across ranks and runs die_medium_deposit_feed_diffuse_tile). */
int die_forward_move_claim_tile(const die_medium* m, const die_agents* a, die_gradient_agent* g, const die_action* act,
                                const die_dynamics* d, void* workspace, int64_t workspace_bytes, void* stream);

/* The stages of die_env_step, individually (tests and custom update cycles such as
 * examples/simple_agents.py:16-30 `_manual_step`). */
int die_agent_move_claim(const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                         void* workspace, int64_t workspace_bytes, void* stream);
int die_agent_resolve(const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                      void* workspace, int64_t workspace_bytes, void* stream);
int die_step_reduce(const die_agents* a, const die_dynamics* d, die_step_result* result,
                    void* workspace, int64_t workspace_bytes, void* stream);
/* The field half of a step in ONE sweep (what die_env_step runs after die_agent_move_claim when
 * H % 4 == 0 and the gaussian radius is <= 4): every cell's claim is read, the winner's deposit
 * added (core/env.py:211), occupied cells fed (food -= rate*food, :222-228), and the result
 * diffused and decayed (:136-145) into m->chem_next.  Replaces die_agent_resolve's per-agent
 * scatter + die_diffuse_decay; DIE_ERR_UNSUPPORTED for other shapes (use those two instead). */
int die_medium_deposit_feed_diffuse(const die_medium* m, const die_dynamics* d, void* stream);
/* Same sweep on a halo-padded tile of a decomposed world (no wrap; chem AND claims of the halo
 * must have been exchanged): deposits are applied everywhere, feeding only to the cells at
 * least `halo` away from the array border, the outer `radius` ring of chem_next is unspecified. */
int die_medium_deposit_feed_diffuse_tile(const die_medium* m, const die_dynamics* d, int32_t halo, void* stream);
/* die_medium_deposit_feed_diffuse_tile + die_step_reduce_ex in one launch (no dead-slot pass): an extra workgroup of
 * the sweep sums the claim pass's partials in k_reduce's order (ghost tiles: and its counts of owned alive slots). */
int die_tile_sweep_reduce(const die_medium* medium, const die_agents* agents, const die_dynamics* dynamics, int32_t halo,
                          die_step_result* result, void* ws, int64_t ws_bytes, void* stream);
/* The part of die_agent_resolve that does not touch the field: dead slots finish their feed,
 * lifecycle (agents_die), alive count — for callers that let the field sweep do the scatter. */
int die_agent_dead_slots(const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                         void* workspace, int64_t workspace_bytes, void* stream);
/* Decomposed step, first half: positions only (core/env.py:163-172).  tile_of[n] receives the
 * index tile_x * tiles_y + tile_y of the interior tile (tile_w × tile_h cells each) the agent now
 * stands on, so the caller can migrate it before die_agent_claim_feed. */
int die_agent_move(const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                   int32_t tile_w, int32_t tile_h, int32_t tiles_y, int32_t* tile_of, void* stream);
/* die_gradient_forward fused with die_agent_move (decomposed worlds; the action is written to `act`). */
int die_forward_move(const die_medium* m, const die_agents* a, die_gradient_agent* g, const die_action* act,
                     const die_dynamics* d, int32_t tile_w, int32_t tile_h, int32_t tiles_y, int32_t* tile_of, void* stream);
/* die_step_reduce when the caller knows whether the dead-slot pass ran: without it only the
 * move/claim partials are summed and num_alive = alive_const (with_second_pass: 0 claim pass only; 1 both passes and the
 * dead-slot pass's alive count; 2 ghost tiles; 3 both passes' gains with num_alive = alive_const). */
int die_step_reduce_ex(const die_agents* a, die_step_result* result, void* workspace, int64_t workspace_bytes,
                       int32_t with_second_pass, int64_t alive_const, void* stream);
/* _agent_lifecycle (core/env.py:245-250) on its own: every channel of the slots with agent_food <= 1e-4 becomes 0.  Used by
 * the reference-compatible agents_die mode (die_amd/env.py Env._step_compat), whose claim / feeding lookups run on a frozen
 * copy of the agent array — the reference's stale AgentIndexer (core/env.py:249 vs core/utils.py:22). */
int die_agents_lifecycle(const die_agents* a, void* stream);
/* Decomposed step, second half: claim + feeding of die_agent_move_claim without the move. */
int die_agent_claim_feed(const die_medium* m, const die_agents* a, const die_action* act, const die_dynamics* d,
                         void* workspace, int64_t workspace_bytes, void* stream);
/* gaussian × (1 − decay) on a halo-padded tile: no wrap; exact for cells at least `radius` away
 * from the array border, the outer ring of dst is unspecified. */
int die_diffuse_decay_tile(const void* src, void* dst, int32_t W, int32_t H, int32_t dtype,
                           float sigma, float decay, void* stream);
/* gaussian(sigma, mode='wrap') × (1 − decay): src → dst, W×H planes of `dtype`. */
int die_diffuse_decay(const void* src, void* dst, int32_t W, int32_t H, int32_t dtype,
                      float sigma, float decay, void* stream);
/* … with any of scipy's boundary modes (LDS-tiled kernel; 'wrap' with H % 4 == 0 and radius <= 4 takes the row sweep) */
int die_diffuse_decay_mode(const void* src, void* dst, int32_t W, int32_t H, int32_t dtype, float sigma, float decay,
                           int32_t mode, void* stream);

/* ---- DataInitializer (core/data_init.py:92-253, core/env.py:74-86) ------------------- */
typedef struct die_food_spec {
    int32_t n_waves;             /* sinusoid mix (<= 8 waves), used when perlin_octaves == 0 */
    float scale;
    int32_t perlin_octaves;      /* > 0: with_food_perlin (:228-231): 2-D gradient noise of that many lattice cells per unit */
    float threshold;             /*      length, .round(3), values outside [0, threshold] masked to 0 (Env: 8, 1.0) */
    double fx[8], fy[8], phase[8], amp[8];
} die_food_spec;

/* with_agents(ratio) + with_food(...) + zero chem: fills owner (epoch 1), food, chem. */
int die_init_medium(const die_medium* m, double agent_ratio, uint64_t seed, const die_food_spec* food,
                    void* stream);
/* agents_from_medium (:133-150): occupied cells in row-major order → slots [0, K); slots
 * >= K are zeroed.  K is written to *num_alive_dev (device int64).  Fails at run time
 * (K clipped, flag in num_alive_dev[1]) when K > a->N. */
int die_init_agents(const die_medium* m, const die_agents* a, uint64_t seed, int64_t* num_alive_dev,
                    void* workspace, int64_t workspace_bytes, void* stream);
/* GradientAgent/PhysarumAgent.__init__ state (:42-43,163): heading from N(0,.4) noise,
 * discretised to the turn lattice when turn_radians > 0; stored as the float64 of its fp32 rounding. */
int die_init_heading(uint32_t* heading_hi, uint32_t* heading_lo, float* prev_gx, float* prev_gy, int64_t N, double turn_radians,
                     uint64_t seed, void* stream);

/* DataInitializer's builder steps (core/data_init.py:171-253) on plain fp32 device arrays: one channel per call.
 *   DIE_FIELD_CONST   with_const (:214-216): dst = a
 *   DIE_FIELD_NOISE   with_noise / get_random (:168-169,218-220): (b - a) * random_sample().round(3) + a, the uniform being
 *                     word `word` (0..3) of Philox(seed, step, element)
 *   DIE_FIELD_AGENTS  with_agents (:222-226): ceil(_mask(random_sample().round(3), mask_above = b)), the random stream of
 *                     die_init_medium at the same `step` (0 there)
 *   DIE_FIELD_PERLIN  with_food_perlin / with_chem (:228-236): _mask(perlin.round(3), mask_above = b) on the W x H labels,
 *                     octaves = a (see die_food_spec), lattice seed `seed + step` */
typedef enum die_field_op { DIE_FIELD_CONST = 0, DIE_FIELD_NOISE = 1, DIE_FIELD_AGENTS = 2, DIE_FIELD_PERLIN = 3 } die_field_op;
int die_field_fill(float* dst, int64_t n, int32_t op, int32_t W, int32_t H, double a, double b, uint64_t seed, uint32_t step,
                   uint32_t word, void* stream);
/* build (:238-246): channels x mask -> the medium (occupied cells flagged for die_init_agents; chem / food converted to
 * the field dtype).  A NULL channel is zeros, a NULL mask is the scalar. */
int die_medium_from_fields(const die_medium* m, const float* agents, const float* food, const float* chem, const float* mask,
                           float mask_scalar, void* stream);

/* Env._medium_resource_dynamics (core/env.py:147-150) with the flow operator of WaveSequence.get_flow_operator
 * (core/data_init.py:29-38): food <- scale * z(x, y, t) + (1 - decay) * food, z = WaveSequence.__getitem__(t)
 * (core/data_init.py:71-89) evaluated at the world cell of every local element; float64 arithmetic. */
int die_food_flow_wave(const die_medium* medium, double t, double scale, double decay, void* stream);
/* The same with PerlinNoiseSequence.__getitem__(t) (core/data_init.py:55-69) as the field: round(noise((x, y, t)), 3), `noise` =
 * 3-D gradient noise at (x, y, t) * octaves on the linspace(0, 1, n) labels, lattice gradients from Philox(seed, point). */
int die_food_flow_perlin(const die_medium* medium, double t, int32_t octaves, double scale, double decay, uint64_t seed, void* stream);

/* Env._get_sense_mask (core/env.py:276-290): mask = ceil(round(gaussian(agents channel, sigma, mode 'nearest',
 * truncate 4), decimals)) as W*H bytes (the reference uses sigma 2.0, 3 decimals); float64 accumulation, axis 0
 * then axis 1 as scipy does.  tmp: W*H doubles of device scratch.  Periodic single-tile planes only. */
int die_sense_mask(const die_medium* medium, float sigma, int32_t decimals, uint8_t* mask_out, double* tmp, void* stream);

/* EnvRenderer.render (core/render.py:76-110) on the device, one sweep: rgb_out (W, H, 3) float32 = (agents,
 * env_food, chem1); trace <- trace * trace_decay + agents (FieldTrace.update, :29-30) and rgba_out (W, H, 4) =
 * lut[index(trace)] with matplotlib's Colormap.__call__ index rule, lut = (lut_n + 3, 4) float32 (colours, under,
 * over, bad); rgb8_out (W, H, 3) uint8 = clip(rgb, 0, 1) * 255.  Any output may be NULL (trace may be NULL when
 * rgba_out is). */
int die_render_frames(const die_medium* medium, float* trace, float trace_decay, const float* lut, int32_t lut_n,
                      float* rgb_out, float* rgba_out, uint8_t* rgb8_out, void* stream);

/* ---- spatial re-ordering of the agent arrays (no reference counterpart; see die_sort.hip) ----
 * Writes `in` permuted into `out` (different arrays, out->slot required) so that array
 * neighbours are grid neighbours: stable sort by the (ix/8, iy/64) bucket of each agent's cell.
 * Up to 4 further per-slot float arrays (agent-object state such as headings) are permuted
 * alongside.  Results of forward/step are independent of the order. */
int64_t die_sort_workspace_bytes(int32_t W, int32_t H, int64_t N);
int die_agents_sort(const die_medium* m, const die_agents* in, const die_agents* out, int32_t n_extra,
                    const float* const* extra_in, float* const* extra_out, void* workspace,
                    int64_t workspace_bytes, void* stream);

/* ---- tile-binned step (die_pic.hip; no reference counterpart for the data structure) --------------------------------
 * The fast path of `env.step(agent.forward(obs))` (examples/minimal_run.py:24-25) for worlds in which every slot is alive
 * and an agent moves less than a tile per step: the agent arrays are kept in EXACT tile order (tiles of 2^tile_xs ×
 * 2^tile_ys cells; 64×64, 32×128, 32×64 and 16×32 are compiled in), "last writer wins" of core/env.py:211 is resolved in LDS
 * and the field sweep reads a 4-byte deposit plane — no claim plane, no global atomics on cells.  Same bits as
 * die_forward_env_step.  A layout = the per-agent arrays plus per-tile words: segment offset / size, number of stayers
 * (agents of the segment that stand on the tile; the rest have walked onto a neighbouring tile), arrivals.  The step reads
 * layout[from] and writes layout[1 - from]; the caller alternates.  The 'agents' channel (m->owner) is NOT maintained by this
 * path: die_agents_mark_owner rebuilds it on demand. */
typedef struct die_pic_layout {
    uint32_t* x;             /* N, Q0.32 */
    uint32_t* y;
    float* agent_food;       /* N */
    uint32_t* slot;          /* N, reference slot ids (always materialised) */
    uint32_t* heading_hi;    /* N, the agent object's _direction_rads (float64 halves, die_gradient_agent) in this order */
    uint32_t* heading_lo;
    uint32_t *off, *n, *s, *inc;   /* die_pic_tiles() words each */
} die_pic_layout;

typedef struct die_pic {
    int32_t tile_xs, tile_ys;    /* log2 of the tile shape */
    int64_t N;                   /* agents = slots, all alive */
    die_pic_layout layout[2];
    float* dep;                  /* N floats of scratch */
    float* dep_plane;            /* three-launch form only (may be NULL when rim is given and the step qualifies), W*H: per cell the deposit of the highest slot standing
                                    on it, or 0xFFFFFFFF */
    void* part_gain;             /* 2 * die_pic_tiles() 64-bit words: reward partials per tile, then (tiles of a decomposed world:
                                    die_medium.gW > 0) the owned agents per tile; also the binning scratch */
    uint32_t* error;             /* device word, 0 = fine; sticky bits after a step: 1 segment bookkeeping broken, 2 an agent moved
                                    further than a tile.  Never cleared by the library */
    int32_t k1_threads;          /* tuning: workgroup size of the agent kernel (multiple of 64, <= 512); 0 = default */
    int32_t stages;              /* 0 = the whole step; else a bit mask of the launches to run (per-kernel timing: bench.py):
                                    1 agent kernel (ONE issue per step: it adds to layout[1 - from].inc), 2 claim resolution +
                                    next offsets, 4 field sweep (two-launch form: the sweep is part of the kernel of bit 2) */
    /* two-launch form (rim != NULL): the agent kernel also lists, per tile, the agents of the tile's new segment that matter
     * to ANOTHER tile's field — those that walked off the tile and those that stand within the gaussian radius of a border —
     * and ONE kernel per tile reads the lists of the 9 tiles around it (fixed places: requested with the first loads),
     * resolves the claims of its tile + rim in LDS, adds the winners' deposits, diffuses, decays and feeds: no deposit
     * plane, one launch less.  A list that overflows costs time (that tile's segment is then scanned), never correctness. */
    void* rim;                   /* die_pic_tiles() * die_pic_rim_cap() records of 16 bytes (x, y, slot, deposit bits), or NULL: three
                                    launches (claim resolution writes dep_plane, die_env.hip's sweep reads it) */
    uint8_t* rim_code;           /* die_pic_tiles() * die_pic_rim_cap() bytes: where each listed agent stands (see die_pic.hip) */
    uint32_t* rim_cnt;           /* die_pic_tiles() words */
    int64_t* status_out;         /* two-launch form, may be NULL: the step copies *error here next to writing `result` — a caller that
                                    places it behind its die_step_result reads reward, num_alive and the error word in ONE copy (both
                                    may be device-visible pinned host memory: one thread writes the three words with system-scope atomic stores, so their visibility does not wait for the kernel's end) */
    /* PhysarumAgent's random turn (core/agent/gradient.py:183) on this path: one bit per slot id from a table the library
     * fills, one Philox block per 128 slots (same bits as the stand-alone forward evaluates per agent: csrc/die_rng.h
     * die_turn_word).  The field kernel of a step fills it for (g->seed, g->step + 1); a caller that steps with the same
     * seed and consecutive g->step sets turn_ready = 1 from the second step on, else the step fills the table first. */
    uint32_t* turn_bits;         /* 4 * ceil(turn_slots / 128) words */
    int64_t turn_slots;          /* slot ids are < turn_slots (>= N; a decomposed world: the world's slot count) */
    int32_t turn_ready;          /* the table already holds the bits of (g->seed, g->step) */
    /* ORDER TABLE of the two-launch form's workgroups (ABI 23; may be NULL: workgroup id -> tile by XCD bands of columns, as before).
     * order[j * (tiles / 8) + k] = the tile the k-th workgroup of XCD j takes (linear workgroup id = 8 k + j): a permutation of band j's
     * tiles, written by the library (k_pic_order: crowded tiles first, band order among equals) from the populations of the layout a
     * step reads — when order_ready == 0 and every 32nd step (g->step % 32 == 0).  Only WHICH workgroup takes a tile changes, never a
     * result.  Late in a run, when the agents have aggregated (tiles of 3 000 agents beside tiles of 100), the crowded tiles — and the
     * tiles whose rim lists overflow — no longer make up a launch's tail: 166 -> 144 us per step at world step 3 000 of the benchmark
     * world (round 6).  Used when tiles-per-row % 8 == 0, tiles <= 65 536, an undivided world, all tiles in one launch; ignored otherwise.
     * The caller allocates die_pic_tiles() 16-bit words (4-byte aligned) and sets order_ready = 1 once a step has run with them (the
     * table stays a valid permutation across re-binning: it is only stale then). */
    int32_t order_ready;
    uint16_t* order;
    /* ONE launch (stages = 1 or 2) over a subset of the tiles: sub_mode 0 all tiles; 1 only the rectangle [sub_tx0, sub_tx0 +
     * sub_ntx) x [sub_ty0, sub_ty0 + sub_nty) of tiles; 2 all tiles but that rectangle — and, for stages = 2, the workgroups that
     * complete a step (next offsets, reward, turn bits).  A decomposed rank steps the tiles that need nothing from its neighbours
     * while its ghost refresh's messages are in flight, the others once they have arrived (die_amd/dist.py).  The caller issues
     * every tile exactly once per stage; two-launch form only. */
    int32_t sub_mode, sub_tx0, sub_ty0, sub_ntx, sub_nty;
    int32_t halo_fresh;          /* decomposed tiles, the step behind die_pic_ghost_inplace: the agent kernel takes no arrivals on tiles
                                    outside the owned cells (die_medium.own_*) */
    /* The reference's default slot layout on this path (core/data_init.py:143-144: max_agents = W*H slots, most of which never
     * lived): the tiles' segments hold the n_alive alive agents, entries [0, n_alive) of the arrays, the dead slots lie behind them,
     * entries [n_alive, N), where die_pic_bin puts them.  A dead slot acts, moves, burns and "consumes" like the reference's
     * (core/env.py:163-172, 224-243) — from the occupancy map `occ` (one BYTE per cell: the cells alive agents stand on in this
     * step; scratch, W*H bytes) — and never claims, deposits or marks a cell.  0 (or N): every slot is alive.
     * Two-launch form, single-tile worlds. */
    int64_t n_alive;
    void* occ;
    /* GradientAgent with momentum on this path (core/agent/gradient.py:82-91: inertia and / or noise; normalised gradient).
     * prev_grad[l][0 / 1]: _prev_grad's x / y component in the order of layout[l] (N floats each) — the step reads layout[from]'s
     * and writes layout[1 - from]'s, die_pic_bin carries them from g's arrays into layout[into]'s; all NULL when inertia = 0
     * (noise alone keeps no state).  The step length the tiles must hold is |scale| * die_pic_step_bound(inertia, noise_scale). */
    float* prev_grad[2][2];
} die_pic;

/* entries per tile of die_pic.rim for a tile shape, or -1 if the shape is not compiled in */
int64_t die_pic_rim_cap(int32_t tile_xs, int32_t tile_ys);
/* 1 if die_pic_forward_env_step takes the two-launch form for these parameters when rim lists are given (world_max = the longer
 * axis of the WORLD in cells), 0 if it takes three launches (then dep_plane must exist and status_out is not written), -1 for a
 * tile shape that is not compiled in.  The library's own rule: callers need not restate it. */
/* Largest |component| of the vector a GradientAgent's action is `scale` times, per axis, for a normalised gradient: 1 without
 * momentum; with it the fixed point of |u'| <= (1 - inertia) + inertia |u| + noise_scale * 2.67 (2.67 = the largest value of the
 * library's 0.4-sigma Box-Muller normals, which also bounds the initial _prev_grad), i.e.
 * max(2.67, 1 + 2.67 noise_scale / (1 - inertia)); inertia = 0: 1 + 2.67 noise_scale.  +inf for inertia >= 1.  Callers pass
 * scale * bound where the rules below ask for `scale`. */
float die_pic_step_bound(float inertia, float noise_scale);
int32_t die_pic_two_launch(int32_t world_max, int32_t tile_xs, int32_t tile_ys, float scale, float diffuse_sigma, int32_t diffuse_mode);
/* number of tiles (words per per-tile array), or -1 if the shape is not compiled in */
int64_t die_pic_tiles(int32_t W, int32_t H, int32_t tile_xs, int32_t tile_ys);
/* Bin agents held in any order (die_agents; `heading` in the same order) into layout[into]; both layouts' per-tile words
 * are initialised.  DIE_ERR_UNSUPPORTED unless the world splits into at least 3×3 whole tiles. */
int die_pic_bin(const die_medium* m, const die_agents* a, const uint32_t* heading_hi, const uint32_t* heading_lo, const die_pic* p,
                int32_t into, void* stream);
/* … and a GradientAgent's _prev_grad (core/agent/gradient.py:42,89), same order, into p->prev_grad[into] (both NULL: die_pic_bin) */
int die_pic_bin_momentum(const die_medium* m, const die_agents* a, const uint32_t* heading_hi, const uint32_t* heading_lo,
                         const float* prev_gx, const float* prev_gy, const die_pic* p, int32_t into, void* stream);
/* GradientAgent/PhysarumAgent.forward (core/agent/gradient.py:96-124) + Env.step (core/env.py:101-131) on binned agents:
 * two launches when p->rim is given and floor(|scale| * (max(W, H) - 1)) + 2 + gaussian radius <= tile (forward + move + feeding +
 * re-binning + rim lists; per-tile claim resolution + deposit + diffusion + feeding + next offsets + reward), else three
 * (forward + move + feeding + re-binning; LDS claim resolution + next offsets; field sweep + reward).  Same bits.  The
 * agent state (g->heading_* are ignored: layout[from].heading_*) moves with the agents; `act` receives the action in the order
 * of layout[from].  Requires: every slot alive, no agents_die / sense mask, normalised gradient without inertia or noise
 * and |scale| * (max(W, H) - 1) <= tile - 1 (else DIE_ERR_UNSUPPORTED / DIE_ERR_ARG: use die_forward_env_step).
 * The planes may be a tile of a decomposed world (die_medium.gW > 0; two-launch form only): agents are binned by the plane
 * cell that holds their world cell (nearest edge for agents beyond the planes), the planes are treated as periodic — what
 * that brings in across their outer edge stays in the outermost cells of the halo, which a ghost-agent decomposition
 * discards —, probes clamp at the WORLD's edge, and reward / num_alive count the agents on cells the rank owns (own_*).  Per axis
 * the planes must either span the world or leave 2 * (probe margin) + 2 cells of it uncovered (the taps around an agent are
 * mapped from one mapping of its own cell; else DIE_ERR_UNSUPPORTED). */
int die_pic_forward_env_step(const die_medium* m, const die_pic* p, int32_t from, die_gradient_agent* g, const die_action* act,
                             const die_dynamics* d, die_step_result* result, void* stream);
/* n_steps x die_pic_forward_env_step(act = NULL) without the host in between — `for i in range(n): obs, ... = env.step(agent.forward(obs))`
 * (examples/minimal_run.py:23-25) for a caller that reads nothing back on the way: step i reads layout[(from + i) & 1] and writes the
 * other one, reads the chem plane step i - 1 wrote (m->chem and m->chem_next exchange roles every step: after an odd n_steps the
 * current plane is m->chem_next), draws from (g->seed, g->step + i), and leaves its result in results[i].  p->stages = 0, no
 * subset of the tiles; p->turn_ready as for the first step.  Same launches, same bits as n_steps separate calls. */
int die_pic_run(const die_medium* m, const die_pic* p, int32_t from, const die_gradient_agent* g, const die_dynamics* d,
                int32_t n_steps, die_step_result* results, void* stream);
/* Steps the calling thread's last die_pic_run had enqueued when it returned: n_steps on success; on an error the number of
 * whole steps already in the stream (arguments are checked by every step, so a failure behind the first one means a failed
 * launch) — the caller adopts the state those steps reach (layout (from + done) & 1, the chem plane roles exchanged `done`
 * times, step counter + done) before it reports the error.  ABI 22. */
int32_t die_pic_run_completed(void);
/* `act` of die_pic_forward_env_step may be NULL: the action then stays in registers.  For a normalised PhysarumAgent it can
 * still be produced afterwards — until the next step overwrites p->dep — from what the step left in layout[lay] (the layout it
 * WROTE): (dx, dy) = scale * polar2xy(1, heading'), deposit = p->dep; same bits as the action the step would have stored, in
 * the order of layout[lay] (core/agent/gradient.py:110-124: the action PhysarumAgent.forward returns). */
int die_pic_action_physarum(const die_pic* p, int32_t lay, const die_gradient_agent* g, const die_action* act, void* stream);

/* ---- Ghost refresh by tiles (die_amd/dist.py DistEnv._refresh_ghosts_tiles; no reference counterpart) ----
 * For a rank of a ghost-agent decomposition whose agents are held in a tile-binned layout (layout[from], as a step or
 * die_pic_bin left it) over planes whose halo is whole tiles deep (die_medium.own_* on tile borders).  An agent is owned iff it
 * stands on an interior tile, so a refresh never looks at every agent:
 *   die_pic_ghost_pack   for every side, tile by tile of its band (tile rectangle [tx0, tx0+ntx) x [ty0, ty0+nty) of the
 *                        interior, row-major): the agents standing on the tile — its stayers and the neighbouring tiles'
 *                        leavers that landed on it — go to send_rec (six streams of `cap` words each: x | y | agent_food |
 *                        slot | heading_hi | heading_lo, the tiles' agents one tile after the other), their number to
 *                        send_counts[tile];
 *   (the caller exchanges the messages: what a neighbour packed for its side -d arrives as recv_* of side d)
 *   die_pic_ghost_merge  layout[1 - from] := tile by tile, the agents standing on an interior tile / the agents that arrived
 *                        for a halo tile (rectangle [hx0, hx0+ntx) x [hy0, hy0+nty), same shape as the band); every agent a
 *                        stayer, both layouts' per-tile words equal — the state die_pic_bin leaves.  Halo tiles no side
 *                        fills become empty.  At most `capacity` array entries are written.
 * summary (device, DIE_PIC_GHOST_SUMMARY_WORDS int64; cleared by die_pic_ghost_pack): [0] agents after the refresh, [1] of
 * them owned, [2 + k] agents sent to side k, [10 + k] agents arrived from side k, [18] flags (DIE_PIC_GHOST_*): the caller
 * reads it once, after die_pic_ghost_merge, and must not use the new layout if a flag is set. */
#define DIE_PIC_GHOST_SUMMARY_WORDS 19
#define DIE_PIC_GHOST_BAD_COUNT 1      /* a tile holds another number of agents than its per-tile words say */
#define DIE_PIC_GHOST_SEND_CAP 2       /* a band holds more agents than `cap` */
#define DIE_PIC_GHOST_BAD_RECV 4       /* a received count is impossible */
#define DIE_PIC_GHOST_CAPACITY 8       /* the agents do not fit `capacity` */
typedef struct die_pic_side {
    int32_t tx0, ty0, ntx, nty;        /* band (tiles of the planes) */
    int32_t hx0, hy0;                  /* halo block filled by the message arriving from this side's neighbour */
    int64_t cap;                       /* agents a message holds */
    uint32_t* send_counts;             /* ntx * nty */
    uint32_t* send_rec;                /* 6 * cap */
    const uint32_t* recv_counts;
    const uint32_t* recv_rec;
} die_pic_side;
int die_pic_ghost_pack(const die_medium* m, const die_pic* p, int32_t from, int32_t n_sides, const die_pic_side* sides,
                       int64_t* summary, void* stream);
int die_pic_ghost_merge(const die_medium* m, const die_pic* p, int32_t from, int32_t n_sides, const die_pic_side* sides,
                        int64_t capacity, int64_t* summary, void* stream);
/* The same in two halves, for a refresh whose messages travel while the next step runs on the tiles that need nothing from a
 * neighbour: phase 1 = the interior tiles' segments (they come first in the new layout: their offsets depend on nothing that
 * arrives; summary[1] = their total), phase 2 = the halo tiles' segments behind them, from the received counts; phase 0 = both
 * (die_pic_ghost_merge).  Phase 1 before phase 2, die_pic_ghost_pack before both. */
int die_pic_ghost_merge_phase(const die_medium* m, const die_pic* p, int32_t from, int32_t n_sides, const die_pic_side* sides,
                              int64_t capacity, int64_t* summary, int32_t phase, void* stream);
/* The refresh IN PLACE, for a step that follows at once and reads layout[from] (not layout[1 - from], as after die_pic_ghost_merge):
 * the interior tiles' segments stay where the step before left them — stayers, then leavers, as the agent kernel reads them
 * anyway — and nothing of them is copied.  phase 1: the places of the interior tiles' segments in the layout the coming step WRITES
 * (layout[1 - from].off / .n; summary[1]); phase 2, once the messages are here: the halo tiles' places behind them (summary[0],
 * arrived counts), and in layout[from] every halo tile gets a NEW segment behind everything the arrays hold — what arrived for it as
 * stayers, then those of its old leavers that stand on an interior tile (a ghost that walked into the interior is owned now) — and
 * its three words.  The coming step must run its agent kernel on the halo tiles with die_pic.halo_fresh = 1 (they take no arrivals:
 * an interior tile's leaver that stands on a halo tile is the stale copy of an agent that arrived with the message).  Entries beyond
 * the old end of the arrays are used: at most die_pic_tiles() words of `tail` scratch, `capacity` array entries. */
int die_pic_ghost_inplace(const die_medium* m, const die_pic* p, int32_t from, int32_t n_sides, const die_pic_side* sides,
                          int64_t capacity, int64_t* summary, int32_t phase, uint32_t* tail, void* stream);
/* Rebuild the 'agents' channel from the agent arrays: atomicMax of (m->epoch, slot) claims (deposit bits 0) for every
 * alive agent.  The caller advances m->epoch (or zeroes the plane) first. */
int die_agents_mark_owner(const die_medium* m, const die_agents* a, void* stream);

/* ---- NeuralAutomataAgent sensing (core/agent/evo.py:45-118,150-174; die_nca.hip) -----------------------------------
 * One layer of ConvolutionModel: a bias-free Conv2d with 'same' circular padding over (channel, x, y) planes,
 *   out[o, x, y] = sum_i sum_a sum_b w[o, i, a, b] * in[i, (x + a - r) mod W, (y + b - r) mod H],  r = k / 2,
 * k odd (1, 3, 5, 7), at most 4 channels in and out, weights (cout, cin, k, k) fp32 in device memory; the last layer of the
 * stack applies tanh (:97-99).  Input planes may be the medium's own: fp32 / fp16 fields, or the 'agents' channel read
 * from the claim plane (occupied cells are 1.0; `epoch` = die_medium.epoch).  Outputs are fp32 planes, never an input. */
typedef enum die_plane_kind { DIE_PLANE_F32 = 0, DIE_PLANE_F16 = 1, DIE_PLANE_AGENTS = 2 } die_plane_kind;
typedef struct die_conv_plane {
    const void* data;
    int32_t kind;            /* die_plane_kind */
    int32_t reserved;
} die_conv_plane;
int die_conv2d_circular(int32_t W, int32_t H, int32_t cin, const die_conv_plane* in, int32_t epoch, int32_t cout,
                        float* const* out, int32_t k, const float* weights, int32_t apply_tanh, void* stream);
/* The same layer with any `boundary` of ConvolutionModel (core/agent/evo.py:51,86 → torch.nn.Conv2d padding_mode). */
typedef enum die_pad_mode { DIE_PAD_CIRCULAR = 0, DIE_PAD_ZEROS = 1, DIE_PAD_REFLECT = 2, DIE_PAD_REPLICATE = 3 } die_pad_mode;
int die_conv2d(int32_t W, int32_t H, int32_t cin, const die_conv_plane* in, int32_t epoch, int32_t cout,
               float* const* out, int32_t k, const float* weights, int32_t apply_tanh, int32_t padding_mode, void* stream);
/* NeuralAutomataAgent.forward's per-agent read-out (core/agent/evo.py:161-170, core/utils.py:56-65): for EVERY slot
 * action[c, n] = planes[c][cell(x_n), cell(y_n)] * coefs[c], c = dx, dy, deposit1. */
int die_gather_scale(const die_medium* m, const die_agents* a, const float* const* planes, const float* coefs,
                     const die_action* out, void* stream);

/* ---- message packing for decomposed worlds (die_amd/dist.py; no reference counterpart) ----------
 * A block [r0, r1) x [c0, c1) of a row-major plane (pitch in elements, 2/4/8-byte elements) copied
 * to / from byte offset buf_offset of one contiguous message buffer; up to 16 blocks per launch. */
typedef struct die_rect {
    void* plane;
    int32_t pitch, r0, r1, c0, c1, elem_bytes;
    int64_t buf_offset;
} die_rect;
int die_rects_pack(const die_rect* rects, int32_t n, void* buf, void* stream);
int die_rects_unpack(const die_rect* rects, int32_t n, const void* buf, void* stream);
/* buffer → plane with plane = max(plane, buffer) on unsigned 64-bit words: merges the claims a
 * neighbour's stray agents made on this rank's cells (guard-band decomposition). */
int die_rects_unpack_max(const die_rect* rects, int32_t n, const void* buf, void* stream);
/* Agent records: word (k, j) of the (n, count) int32 matrix is element idx[j] of array k (4-byte
 * arrays bit-copied, 1-byte arrays widened): migration packs leavers and writes arrivals with one
 * launch each. */
int die_records_gather(void* const* arrays, const int32_t* elem_bytes, int32_t n, const int64_t* idx, int64_t count,
                       int32_t* records_out, void* stream);
int die_records_scatter(void* const* arrays, const int32_t* elem_bytes, int32_t n, const int64_t* idx, int64_t count,
                        const int32_t* records_in, void* stream);

/* ---- Ghost-agent refresh (die_amd/dist.py DistEnv._refresh_ghosts; no reference counterpart) ----
 * die_ghost_plan classifies every local agent of a ghost-agent tile (die_medium.own_* set): agents standing on an
 * interior cell are owned; of those, the ones within the halo depth (own_x0 / own_y0 cells) of side
 * (dirs[2k], dirs[2k+1]) ∈ {-1,0,1}² are listed in lists[k] — their copies become that neighbour's ghosts; every
 * other agent (a ghost, or an agent that walked out) is listed in lists[n_dirs] (holes).  Lists hold ascending
 * 32-bit array indices (deterministic: count per block, scan, fill), at most caps[k] entries each;
 * totals[0..n_dirs) = full list lengths (may exceed caps: the caller checks), totals[n_dirs] = holes,
 * totals[n_dirs + 1] = owned agents.  ws: die_ghost_workspace_bytes(N) bytes of device memory. */
int64_t die_ghost_workspace_bytes(int64_t N);
int die_ghost_plan(const die_medium* medium, const die_agents* agents, int32_t n_dirs, const int8_t* dirs,
                   int32_t* const* lists, const int64_t* caps, int64_t* totals, void* ws, int64_t ws_bytes, void* stream);
/* die_records_gather with the count on the device: packs min(*count_dev, cap) records of the 32-bit index list into
 * the (n, cap) matrix of a message and writes the true count to header_out (may be NULL). */
int die_records_gather_dev(void* const* arrays, const int32_t* elem_bytes, int32_t n, const int32_t* idx,
                           const int64_t* count_dev, int64_t cap, int32_t* records_out, int64_t* header_out, void* stream);
/* die_records_scatter for a 32-bit index list and a record matrix with row pitch `pitch` >= count. */
int die_records_scatter_at(void* const* arrays, const int32_t* elem_bytes, int32_t n, const int32_t* idx, int64_t count,
                           int64_t pitch, const int32_t* records_in, void* stream);
/* The refresh without the host in the middle.  Message k of a buffer = [int64 count at hdr_off[k]] [(n_arrays, caps[k])
 * int32 record matrix at rec_off[k]] (field blocks follow: die_rects_pack).  die_ghost_pack fills every side's records
 * and header from the plan's lists / totals in ONE launch.  die_ghost_apply consumes a received buffer: arrival j (sides
 * in order) overwrites holes[j], or is appended behind n_local once the holes are used up; with fewer arrivals than
 * holes the kept entries of the cut tail move into the remaining holes (plan_ws = the workspace die_ghost_plan wrote
 * its membership words to).  `capacity` = entries each array holds: an arrival that would land at or beyond it is dropped,
 * never written (n_new_out[0] still reports the unclamped count, so the caller sees the overflow).  All counts are read on the device; n_new_out (3 + 2 * n_dirs words) receives what the host
 * wants afterwards: [new number of local agents, holes, owned, sent per side…, arrived per side… (raw headers)]. */
int die_ghost_pack(void* const* arrays, const int32_t* elem_bytes, int32_t n_arrays, int32_t n_dirs, int32_t* const* lists,
                   const int64_t* totals, const int64_t* caps, const int64_t* hdr_off, const int64_t* rec_off, void* send_buf,
                   void* stream);
int die_ghost_apply(void* const* arrays, const int32_t* elem_bytes, int32_t n_arrays, int32_t n_dirs, const int64_t* totals,
                    const int64_t* caps, const int64_t* hdr_off, const int64_t* rec_off, const void* recv_buf,
                    const int32_t* holes, const void* plan_ws, int64_t n_local, int64_t capacity, int64_t* n_new_out, void* stream);

/* A plain streaming copy of `bytes` (a multiple of 16; both buffers 16-byte aligned) device bytes, 16 bytes per lane: what bench.py
 * measures as this GPU's streaming ceiling (roofline.stream_ceiling_gbs), beside the 8 TB/s peak.  No reference counterpart. */
int die_stream_copy(const void* src, void* dst, int64_t bytes, void* stream);

/* Device-visible address of pinned host memory on `device` (-1: the current one), through the HIP runtime this library is linked
 * against (hipHostGetDevicePointer).  Env(sync=True) lets the step's last kernel write its three result words there. */
int die_host_device_pointer(void* host, int32_t device, void** dev_out);

#ifdef __cplusplus
}
#endif
#endif /* DIE_HIP_H */
