#!/usr/bin/env python3
"""bench.py — env steps/sec of the Physarum grid step (agent.forward + env.step) on MI355X.

Contract (see the task brief): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON
line on rank 0.  With N > 1 and no rank environment the command launches its own N ranks
(torch.distributed.run) before touching the GPU; a failure of the decomposed path ends the run
with a non-zero status — there is no fallback to independent replicas.  A "step" is one
`obs, … = env.step(agent.forward(obs))` (examples/minimal_run.py:24-25) over a synthetic 4096x4096 fp32 grid
(BASELINE.json configs[2]) with the state already resident in HBM.  On the tile-binned path a PhysarumAgent's
action is NOT stored by the step: it stays in registers and is re-derived, bit for bit, when somebody reads it
(`--eager-actions` stores it every step; that rate is reported beside the headline, and the roofline figures are
given on both byte bases: the contract's 12·C + 104·K and the 12·C + 92·K that remain without the stored action).  Next to it: `roofline` for the dominant kernel (HIP-event
timed here, algorithmic bytes from DESIGN.md §5) and `cpu_baseline` (the float64 numpy oracle,
a *port*, timed on this box's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# Untimed steps before --warmup.  A FIXED count (round 2 pre-warmed by wall clock: 289 steps on one box, 306 on another — the
# timed steps then sampled different world states).  192 steps ≈ 30 ms of GPU work (first launches, lazily created state and
# the clock ramp are behind us) and the timed steps of the driver's command are world steps 197..216: the regime in which the
# deposited trail network exists (chem max ≈ 0.85 around step 200 — scratch/longrun.py — and fades after step ≈ 1000 as the
# food runs out; the line reports chem_max right after the timed steps).
PREWARM_STEPS = 192


def kernel_source_sha():
    """sha1 over the translation unit of the kernels `roofline` names (csrc/die_pic.hip and every header it pulls in):
    stamps which build a committed PMC file belongs to."""
    import hashlib
    h = hashlib.sha1()
    for f in ('die_amd/csrc/die_pic.hip', 'die_amd/csrc/die_forward.h', 'die_amd/csrc/die_common.h', 'die_amd/csrc/die_rng.h', 'include/die_hip.h'):
        h.update(os.path.basename(f).encode())
        h.update(open(os.path.join(ROOT, f), 'rb').read())
    return h.hexdigest()[:16]


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no rank environment: this process becomes the launcher.  It starts N
    fresh rank processes with torch.distributed.run BEFORE anything here has touched the GPU, relays their output
    (rank 0 prints the JSON line) and exits with their status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.run(cmd, env=env).returncode


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=200)
    p.add_argument('--warmup', type=int, default=20)
    p.add_argument('--size', type=int, default=4096)
    p.add_argument('--ratio', type=float, default=0.15)
    p.add_argument('--seed', type=int, default=1234)
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-steps', type=int, default=6)
    p.add_argument('--kernel-reps', type=int, default=20)
    p.add_argument('--sort-every', type=int, default=None, help='classic step: re-sort period (default: Env decides: 8, or never on worlds below 2^19 cells)')
    p.add_argument('--fields', choices=['f32', 'f16'], default='f32', help='dtype of the field channels (configs[4] uses f16)')
    p.add_argument('--migrate-every', type=int, default=8, help='decomposed runs: agents / ghosts cross ranks every M steps (1 = every step)')
    p.add_argument('--dist-mode', choices=['ghost', 'guard'], default='ghost',
                   help='decomposed runs: ghost = communication-avoiding (ghost agents, nothing crosses ranks for M steps); '
                        'guard = claims merged and halos exchanged every step, strays handed over every M steps')
    p.add_argument('--no-refresh-overlap', action='store_true', help='decomposed ghost-agent runs: refresh right behind its step (default: the refresh waits for the next step and its messages travel under that step\'s interior tiles)')
    p.add_argument('--no-pic', action='store_true', help='classic step (claim plane + bucket sort) instead of the tile-binned one')
    p.add_argument('--pic-tile', default='', help='tuning: log2 tile shape of the tile-binned step, e.g. 6,6')
    p.add_argument('--pic-threads', type=int, default=0, help='tuning: workgroup size of the tile-binned agent kernel')
    p.add_argument('--pic-three-launches', action='store_true', help='tile-binned step in its three-launch form (K2 + deposit plane + sweep) instead of one field kernel per tile')
    p.add_argument('--eager-actions', action='store_true', help='tile-binned step: store the action of every step (default: it stays in registers and is re-derived when read)')
    p.add_argument('--replicas', type=int, default=0, help='batched env replicas on one GPU (BASELINE configs[4]): R worlds of --size in one launch pair; value = replica-steps/s')
    p.add_argument('--force-dist', action='store_true', help='use the decomposed path even on one rank (testing)')
    p.add_argument('--prewarm', type=int, default=PREWARM_STEPS, help='untimed world steps before --warmup (fixed: every box times the same world steps)')
    p.add_argument('--step-events-in-timed-region', action='store_true', help='record the per-step HIP events inside the timed region (rounds 1-3 did; default: in a second, untimed pass of K steps)')
    p.add_argument('--no-extras', action='store_true', help='skip the side measurements (sync=True, reference-default slot count)')
    return p.parse_args()


def contract_bytes(C, K, bf=4):
    """B = C·3·b_f + K·(32 + 24 + 6·b_f + 4 + 5·b_f) per env step — fp32 fields 12·C + 104·K, fp16 fields 6·C + 82·K."""
    return 3 * bf * C + (60 + 11 * bf) * K


def algorithmic_bytes(C, K, action_stored=True, bf=4):
    """DESIGN.md §5 / SURVEY.md §8(d): B = 3·b_f·C + (60 + 11·b_f)·K per env step (b_f = bytes per field element: fp32 fields
    12·C + 104·K, fp16 fields 6·C + 82·K); per kernel, what an ideal version of that kernel has to move.
    `action_stored=False` (tile-binned step with a PhysarumAgent: the action stays in registers and is re-derived only if the
    caller reads it): the 12 bytes per agent of the action are not claimed."""
    act = 12 if action_stored else 0
    if bf != 4:                           # fp16 field channels (the classic kernels' lines are not split out there)
        return {
            'k_pic_forward_move': (32 + act + 6 * bf) * K,        # state R+W 32, action W 12 (if stored), 6 gathers of b_f
            'k_pic_resolve': (4 + 2 * bf) * K,
            'k_diffuse_rows_dep': 2 * bf * C + 2 * bf * K,
            'k_pic_resolve_diffuse': 2 * bf * C + (4 + 4 * bf) * K,   # chem R + W per cell; per agent claim 4, food RMW, chem RMW
            'k_forward_move_claim': (48 + 6 * bf) * K, 'k_diffuse_rows_fused': 2 * bf * C + (4 + 4 * bf) * K,
            'k_gradient_forward': (28 + 5 * bf) * K, 'k_move_claim': (40 + bf) * K,
            'step': 3 * bf * C + (48 + act + 11 * bf) * K,
        }
    return {
        # fused forward + move + claim + feed: x,y R+W 16, heading R+W 8, agent_food R+W 8, action W 12,
        # 6 gathers (4 chem taps, food at old and new cell) 24, claim 4
        'k_forward_move_claim': 72 * K,
        'k_gradient_forward': 48 * K,     # stand-alone forward: x,y,heading R 12 + heading W 4 + action W 12 + 5 gathers 20
        'k_move_claim': 44 * K,           # stand-alone: x,y RW 16 + agent_food RW 8 + action R 12 + food gather 4 + claim 4
        'k_diffuse_rows_fused': 8 * C + 20 * K,   # chem R + W per cell; per agent chem RMW 8 + food RMW 8 + mark 4
        # tile-binned step: the same contract split over its three launches (the claim of the contract is K2's)
        'k_pic_forward_move': (56 + act) * K,     # state R+W 32, action W 12 (if stored), 6 gathers 24
        'k_pic_resolve': 12 * K,          # the ownership claim 4 + food RMW of the occupied cell 8
        'k_diffuse_rows_dep': 8 * C + 8 * K,      # chem R + W per cell; per agent chem RMW 8
        # two-launch form: K2 and the sweep in one kernel per tile (the sum of the two lines above)
        'k_pic_resolve_diffuse': 8 * C + 20 * K,
        'step': 12 * C + (92 + act) * K,
    }


PMC_FILE = os.path.join(ROOT, 'profiles', 'current_pmc_traffic_per_kernel_avg.json')
PMC_NAMES = {'k_gradient_forward': 'void k_gradient_forward<float, 1>', 'k_move_claim': 'void k_move_claim<float>',
             'k_forward_move_claim': 'void k_forward_move_claim<float, 1, false, true>',
             'k_diffuse_rows_fused': 'void k_diffuse_rows<float, 2, 1, true>',
             'k_pic_forward_move': 'void k_pic_forward_move<float, 1, true, false, true, false, false>', 'k_pic_resolve': 'void k_pic_resolve<float, 6, 6, true>',
             'k_pic_resolve_diffuse': 'void k_pic_resolve_diffuse<float, 6, 6, 2, false>',
             'k_diffuse_rows_dep': 'void k_diffuse_rows<float, 2, 2, true>'}
PMC_DTYPE = ['float']        # '__half' when --fields f16 (main sets it): the instantiation names of the fp16 field kernels
WIDE_STREAM_KERNELS = ('k_diffuse_rows_fused', 'k_diffuse_rows_dep', 'k_pic_forward_move', 'k_pic_resolve', 'k_pic_resolve_diffuse')
def pmc_traffic(kernel, K=0):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same command
    (profiles/README.md): (2 x FETCH_SIZE + WRITE_SIZE) KiB.  On gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B while the L2 fetches
    128-byte lines (MI355X_MICROARCH.md: exactly half of a wide coalesced stream): doubled for the kernels whose reads are
    coalesced streams.  Calibrated on this path (round 4, profiles/r04_final_ea_requests.txt): TCC_EA0_RDREQ of the agent kernel
    x 128 B = 208 MB against the 194 MB + lists it must read at least (both planes once, the agent streams once) — the
    4-byte-per-lane agent streams are requests of whole lines too, so nothing is exempt from the factor (rounds 2-3 exempted them
    and reported 628 MB per step where this rule gives 729).
    The file carries the sha of the kernel sources it was taken from: when the sources have changed since, the figure
    would be stale and is reported as null."""
    try:
        doc = json.load(open(PMC_FILE))
        if doc.get('kernel_source_sha') != kernel_source_sha():
            return None, f'{os.path.relpath(PMC_FILE, ROOT)} was taken from another build of the kernels (sha {doc.get("kernel_source_sha")}): not reported'
        name = PMC_NAMES[kernel].replace('<float', '<' + PMC_DTYPE[0])
        c = doc[name] if name in doc else doc[[k for k in doc if k.startswith(name.rstrip('>'))][0]]       # (trailing template arguments may have been added)
        fetch = c['FETCH_SIZE'] * 1024
        if kernel in WIDE_STREAM_KERNELS:
            fetch *= 2
        return int(fetch + c['WRITE_SIZE'] * 1024), (f'{os.path.relpath(PMC_FILE, ROOT)} (separate --pmc passes of this command, '
                                                      f'kernel sources sha {doc["kernel_source_sha"]}; FETCH_SIZE x 2: 128-byte lines)')
    except Exception as e:
        return None, f'unavailable: {type(e).__name__}'


def time_kernels(env, agent, reps):
    """Average launch duration (µs) of each kernel of the step, HIP events on the launch stream."""
    import torch
    from die_amd import _lib
    obs = env._get_current_obs
    out = {}

    def timed(fn, prep=None):
        """Average device time of what `fn` enqueues.  A ~100 µs spin kernel is queued first, so that the start
        event, the launch and the stop event are all in the queue before the GPU reaches them: the host's
        enqueue latency stays out of the measurement (it agrees with rocprofv3's kernel durations)."""
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            args = prep() if prep else ()
            torch.cuda._sleep(250000)
            a.record()
            fn(*args)
            b.record()
            torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / reps * 1e3

    import ctypes as C
    from die_amd.device_array import _ptr, stream_ptr

    if env._pic is not None and env._pic.held is not None and env._pic.held[0] is env.agents.x:
        # tile-binned step: REAL steps whose three launches are issued one call each (die_pic.stages) with a HIP event
        # between them — every kernel meets the cache state it meets in the timed loop
        two = env._pic.two_launch(env, agent)
        names = ('k_pic_forward_move', 'k_pic_resolve_diffuse') if two else ('k_pic_forward_move', 'k_pic_resolve', 'k_diffuse_rows_dep')
        nk = len(names)
        n = max(reps, 20)
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(nk + 2)] for _ in range(n)]
        o = env._get_current_obs
        for e in evs:
            env._pic_events = e
            o, *_ = env.step(agent.forward(o))
            e[nk + 1].record()            # nothing was enqueued since e[nk]: what an empty event interval costs
        env._pic_events = None
        torch.cuda.synchronize()
        for k, name in enumerate(names):
            out[name] = sum(e[k].elapsed_time(e[k + 1]) for e in evs) / n * 1e3
        # an interval between two events contains the kernel AND the completion / dispatch gap around it (the
        # intervals add up to the step): rocprofv3's kernel durations are shorter by about this much per launch
        out['_empty_event_interval'] = sum(e[nk].elapsed_time(e[nk + 1]) for e in evs) / n * 1e3
        return out

    env.sort_agents()                     # the timed loop re-sorts every few steps: measure in that regime
    obs = env._get_current_obs
    sort_every, env._sort_every = env._sort_every, 0
    calls = [0]

    def front_prep():                     # host work of one fused forward + move/claim launch
        calls[0] += 1
        if sort_every > 0 and calls[0] % sort_every == 0:
            env.sort_agents()             # same re-sort cadence as the timed loop
        act = agent.forward(obs)          # pending: its kernel runs inside die_forward_move_claim
        env.medium.next_epoch()
        return (act, env.medium.c_struct(), env.agents.c_struct(), act.raw_struct(), env._c_dynamics())

    def front(act, m, a, u, d):           # k_forward_move_claim alone, through the C ABI
        _lib.check(_lib.lib.die_forward_move_claim(C.byref(m), C.byref(a), C.byref(act.g_struct), C.byref(u), C.byref(d),
                                                   _ptr(env._workspace), env._workspace.numel(), stream_ptr(env.device)),
                   'die_forward_move_claim')
        agent._forward_consumed(act)
    out['k_forward_move_claim'] = timed(front, front_prep)
    out['k_diffuse_rows_fused'] = timed(env._medium_deposit_feed_diffuse)
    env._sort_every = sort_every
    return out


def copy_ceiling_gbs(device, mb=512, reps=10):
    """Measured device-copy ceiling of this box: read + write bytes of a plain D2D copy per second."""
    import torch
    src = torch.empty(mb << 20, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    dst.copy_(src)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        dst.copy_(src)
    b.record()
    torch.cuda.synchronize()
    return 2.0 * (mb << 20) * reps / (a.elapsed_time(b) * 1e-3) / 1e9


def stream_ceiling_gbs(device, mb=512, reps=20):
    """The in-tree streaming copy (die_stream_copy: one 16-byte vector per thread) through the C ABI: read + written bytes per
    second — what this box's GPU moves when nothing but a stream is asked of it (MI355X_MICROARCH.md measures
    6.29 TB/s for a float4 copy; the shapes that do worse: scratch/kbench_dma/copy_sweep.hip)."""
    import ctypes as C
    import torch
    from die_amd import _lib
    from die_amd.device_array import _ptr, stream_ptr
    src = torch.zeros(mb << 20, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    call = lambda: _lib.check(_lib.lib.die_stream_copy(_ptr(src), _ptr(dst), mb << 20, stream_ptr(torch.device(device))), 'die_stream_copy')
    for _ in range(3):
        call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        call()
    b.record()
    torch.cuda.synchronize()
    return 2.0 * (mb << 20) * reps / (a.elapsed_time(b) * 1e-3) / 1e9


def side_measurements(args, device, agent_kw, torch, die_amd):
    """What the headline does not show: the same workload (a) through the Gym API's synchronous form —
    `Env(sync=True)`: float reward + info dict, i.e. one host read per step — and (b) with the reference's default slot
    count, `max_agents=None` → N = W·H slots of which 85 % are dead and still move, burn and count in `reward`
    (core/data_init.py:143-144): the classic step with its dead-slot pass.  Not the metric: steps/s each, own worlds."""
    W = H = args.size
    dt_f = torch.float16 if args.fields == 'f16' else torch.float32
    out = {}
    # (c) SURVEY §8(d)'s stress case: the LITERAL normalised parameters of examples/minimal_run.py:42 (sense_offset .04, scale
    #     .006 — a 164-cell probe and a 25-cell step at 4096²) instead of the cell-unit-constant ones; with the path it took
    literal = dict(agent_kw, sense_offset=0.04, scale=0.006)
    # (d) the reference's default GradientAgent momentum (inertia .9, noise .025, sense_offset 0: core/agent/gradient.py:19-30) at the
    #     benchmark's step length: _prev_grad travels through the tile-binned layouts (die_pic.prev_grad)
    momentum = dict(scale=agent_kw['scale'], sense_offset=0.0, inertia=0.9, noise_scale=0.025)
    for name, kw, n, akw in (('sync_true_steps_per_s', dict(max_agents='alive', sync=True), 100, agent_kw),
                             ('reference_default_slots_steps_per_s', dict(max_agents=None, sync=False), 30, agent_kw),
                             ('literal_normalised_parameters_steps_per_s', dict(max_agents='alive', sync=False), 60, literal),
                             ('gradient_agent_default_momentum_steps_per_s', dict(max_agents='alive', sync=False), 60, momentum)):
        try:
            env = die_amd.Env((W, H), die_amd.Dynamics(init_agent_ratio=args.ratio), seed=args.seed, device=device, field_dtype=dt_f, **kw)
            if akw is momentum:
                agent = die_amd.GradientAgent(max_agents=env.agents.N, seed=args.seed, **akw)
            else:
                agent = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=args.seed, **akw)
            obs = env._get_current_obs
            for _ in range(max(20, n // 2)):
                obs, *_ = env.step(agent.forward(obs))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                obs, *_ = env.step(agent.forward(obs))
            torch.cuda.synchronize()
            out[name] = round(n / (time.perf_counter() - t0), 1)
            if kw['max_agents'] is None:
                out['reference_default_slots'] = int(env.agents.N)
            if akw is momentum:
                binned = getattr(env, '_pic', None) is not None and env._pic.held is not None
                out['gradient_agent_default_momentum'] = {'inertia': 0.9, 'noise_scale': 0.025, 'step_cells': round(akw['scale'] * (W - 1), 2),
                                                          'path': 'tile-binned, two launches (action stored every step)' if binned else 'classic step'}
            if akw is literal:
                binned = getattr(env, '_pic', None) is not None and env._pic.held is not None
                out['literal_normalised_parameters'] = {'sense_offset': 0.04, 'scale': 0.006, 'probe_cells': round(0.04 * (W - 1), 1), 'step_cells': round(0.006 * (W - 1), 1),
                                                        'path': ('tile-binned, ' + ('two' if env._pic.two_launch(env, agent) else 'three') + ' launches') if binned
                                                        else 'classic step (claim plane, re-sort every 8 steps): the probe reaches further than the 24 cells the tile-binned agent kernel stages'}
            del env, agent, obs
            torch.cuda.empty_cache()
        except Exception as e:               # a side measurement never takes the headline down
            out[name] = f'failed: {type(e).__name__}: {e}'
    # (e) Env.run: the same steps as ONE library call (die_pic_run: a C loop over the step — no Python between two steps)
    try:
        env = die_amd.Env((W, H), die_amd.Dynamics(init_agent_ratio=args.ratio), seed=args.seed, device=device, field_dtype=dt_f, max_agents='alive', sync=False)
        agent = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=args.seed, **agent_kw)
        env.run(agent, 40)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.run(agent, 100)
        torch.cuda.synchronize()
        out['env_run_library_loop_steps_per_s'] = round(100 / (time.perf_counter() - t0), 1)
        out['env_run_library_loop'] = 'die_pic_run' if getattr(env, 'library_runs', 0) else 'Python step loop (this world does not take the tile-binned two-launch step)'
        del env, agent
        torch.cuda.empty_cache()
    except Exception as e:
        out['env_run_library_loop_steps_per_s'] = f'failed: {type(e).__name__}: {e}'
    return out


def cpu_baseline(env, agent_kw, n_steps, seed):
    """The float64 numpy/scipy oracle on the same initial state (downloaded from the device)."""
    import numpy as np
    from oracle import cpu_ref as R
    medium, agents = env.medium.to_numpy(), env.agents.to_numpy()
    ref_env = R.RefEnv(medium, agents)
    ref_agent = R.RefPhysarumAgent(agents.shape[1], seed=seed, **agent_kw)
    obs = ref_env.obs
    obs, *_ = ref_env.step(ref_agent.forward(obs))       # warm-up step (first-touch, caches)
    t0 = time.perf_counter()
    for _ in range(n_steps):
        obs, *_ = ref_env.step(ref_agent.forward(obs))
    dt = time.perf_counter() - t0
    return n_steps / dt, dt


def other_cpu_baselines(seed):
    """BASELINE.md §3's two smaller CPU cases, for the record (not the metric): BASELINE configs[0] Brownian 256x256 x 300
    steps (examples/minimal_run.py:32-36) and Physarum 1024x1024 x 50 steps, same float64 oracle."""
    from oracle import cpu_ref as R
    out = []
    for name, (W, H), ratio, steps, mk in (
            ('BrownianAgent 256x256, 300 steps, ratio 0.05', (256, 256), 0.05, 300, lambda N: R.RefBrownianAgent(move_scale=0.01, seed=seed)),
            ('PhysarumAgent 1024x1024, 50 steps, ratio 0.15', (1024, 1024), 0.15, 50,
             lambda N: R.RefPhysarumAgent(N, seed=seed, scale=1.53 / 1023, sense_offset=10.2 / 1023))):
        medium, agents = R.synthetic_init(W, H, ratio, seed=seed)
        agents = agents[:, :int(agents[2].sum())].copy()              # alive-only slots, like the GPU workload
        env, agent = R.RefEnv(medium, agents), mk(agents.shape[1])
        obs = env.obs
        obs, *_ = env.step(agent.forward(obs))
        t0 = time.perf_counter()
        for _ in range(steps):
            obs, *_ = env.step(agent.forward(obs))
        dt = time.perf_counter() - t0
        out.append({'workload': name, 'value': round(steps / dt, 3), 'unit': 'env steps/s', 'seconds': round(dt, 2)})
    return out


def bench_replicas(args, device, agent_kw):
    """`--replicas R`: R worlds of --size stepped by one launch pair per step (die_forward_env_step_batch), against the
    same R worlds stepped one Env at a time.  Not the contract metric: its own line, value = replica-steps per second."""
    import torch
    import die_amd
    from die_amd.batch import BatchedEnv, BatchedPhysarumAgent
    W = H = args.size
    dt = torch.float16 if args.fields == 'f16' else torch.float32
    benv = BatchedEnv((W, H), die_amd.Dynamics(init_agent_ratio=args.ratio), replicas=args.replicas, seed=args.seed, field_dtype=dt,
                      device=device)
    bag = BatchedPhysarumAgent(benv, seed=args.seed, **agent_kw)
    res = torch.empty((benv.R, 2), dtype=torch.float64, device=device)

    def run(n):
        for _ in range(n):
            benv.step(bag, res)
    run(max(args.warmup, 40))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    dt_b = time.perf_counter() - t0
    # the same worlds one Env at a time (classic path on small grids; whatever Env picks)
    envs = [die_amd.Env((W, H), die_amd.Dynamics(init_agent_ratio=args.ratio), seed=args.seed + r, max_agents='alive', device=device,
                        sync=False, field_dtype=dt) for r in range(min(benv.R, 4))]
    ags = [die_amd.PhysarumAgent(max_agents=e.agents.N, seed=args.seed + r, **agent_kw) for r, e in enumerate(envs)]
    obs = [e._get_current_obs for e in envs]

    def run1(n):
        for _ in range(n):
            for i, e in enumerate(envs):
                obs[i], *_ = e.step(ags[i].forward(obs[i]))
    run1(40)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run1(args.steps)
    torch.cuda.synchronize()
    dt_1 = (time.perf_counter() - t0) / len(envs)
    C, K = W * H, sum(benv.n)
    B = (12 if args.fields == 'f32' else 6) * C * benv.R + (104 if args.fields == 'f32' else 82) * K
    line = {'metric': 'batched env replica-steps/sec', 'value': round(args.steps * benv.R / dt_b, 1), 'unit': 'replica steps/s', 'n_gpus': 1,
            'steps': args.steps, 'warmup': max(args.warmup, 40), 'ms_per_step': round(dt_b / args.steps * 1e3, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.fields, 'data': 'synthetic',
            'config': {'workload': f'{benv.R} batched replicas of PhysarumAgent {W}x{H} {args.fields} (BASELINE configs[4] form), one launch pair per step',
                       'alive_agents': K, 'one_env_at_a_time_replica_steps_per_s': round(args.steps / dt_1, 1),
                       'speedup_vs_one_at_a_time': round((args.steps * benv.R / dt_b) / (args.steps / dt_1), 2)},
            'roofline': {'bound': 'hbm', 'kernel': 'whole batched step', 'achieved': round(B / (dt_b / args.steps) / 1e9, 1), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(B / (dt_b / args.steps) / 1e9 / HBM_PEAK_GBS, 4), 'traffic': None,
                         'algorithmic_bytes_per_launch': B}}
    print(json.dumps(line), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and not args.force_dist:
        sys.exit(launch_ranks(args))              # nothing above this line touches the GPU
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist_on = world > 1 or args.force_dist
    backend = os.environ.get('DIE_DIST_BACKEND', 'nccl')      # 'gloo': rehearsal with several ranks sharing one GPU
    if dist_on:
        import datetime
        import torch.distributed as dist
        n_dev = torch.cuda.device_count()
        if backend == 'nccl' and world > n_dev:
            sys.exit(f'bench.py: {world} ranks over RCCL need {world} GPUs, this box has {n_dev} '
                     '(DIE_DIST_BACKEND=gloo rehearses the decomposed path with the ranks sharing one GPU)')
        if n_dev == 1:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        kw = dict(device_id=torch.device(f'cuda:{local_rank}')) if backend == 'nccl' else {}
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300), **kw)
    device = torch.device(f'cuda:{local_rank}')
    torch.cuda.set_device(device)

    import die_amd
    W = H = args.size
    agent_kw = dict(scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1), turn_angle=30, sense_angle=90,
                    turn_tolerance=0.1, deposit=4.0)
    if args.replicas > 0 and not dist_on:
        return bench_replicas(args, device, agent_kw)
    GRIDS = {1: (1, 1), 2: (1, 2), 4: (2, 2), 8: (2, 4)}
    mode, denv = 'single GPU', None
    # A failure anywhere in the decomposed path ends the run with a non-zero status (torch.distributed.run then stops the
    # other ranks): the scaling record shows the failure instead of a replica number under the same metric name.
    if dist_on:
        # weak scaling: every rank owns one W×H tile of a (W·Px)×(H·Py) torus — 2-D domain decomposition with
        # chem-halo exchange and agent migration over RCCL point-to-point (die_amd/dist.py)
        grid = GRIDS.get(world) or (1, world)
        from die_amd.dist import DistEnv
        gW, gH = W * grid[0], H * grid[1]
        # same cell-unit parameters as the single-GPU workload (10.2-cell probe, 1.53-cell step); a non-square world is
        # anisotropic in cells (offsets are fractions of the unit square): size them on the longer axis
        agent_kw.update(scale=1.53 / (max(gW, gH) - 1), sense_offset=10.2 / (max(gW, gH) - 1))
        denv = DistEnv((gW, gH), grid, die_amd.Dynamics(init_agent_ratio=args.ratio), probe_reach=11,
                       device=device, seed=args.seed, sort_every=8 if args.sort_every is None else args.sort_every,
                       migrate_every=args.migrate_every, max_step_cells=1.6, ghosts=args.dist_mode == 'ghost',
                       ghost_headroom=1.3,     # the synthetic world stays uniform (measured fill 0.5 of 2x over 1200 steps)
                       overlap=not args.no_refresh_overlap)
        how = (f'ghost agents, halo ({denv.geo.hx}, {denv.geo.hy}) re-seated every {denv.migrate_every} steps' if denv.ghosts else
               f'halo {denv.geo.h}, claims merged every step, strays handed over every {denv.migrate_every} steps')
        mode = (f'{grid[0]}x{grid[1]} domain decomposition of a {gW}x{gH} torus, {how}, '
                f'{"RCCL" if backend == "nccl" else backend} point-to-point')
        env = denv
        agent = die_amd.PhysarumAgent(max_agents=denv.capacity, seed=args.seed, **agent_kw)   # one seed: streams are keyed by world slot id
    else:
        env = die_amd.Env((W, H), die_amd.Dynamics(init_agent_ratio=args.ratio), seed=args.seed,
                          max_agents='alive', device=device, sync=False, sort_every=args.sort_every,
                          field_dtype=torch.float16 if args.fields == 'f16' else torch.float32, pic=not args.no_pic)
        agent = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=args.seed, **agent_kw)
        if args.eager_actions:
            env._pic_lazy_actions = False
        if args.pic_tile:
            env._pic_tile = tuple(int(v) for v in args.pic_tile.split(','))
        env._pic_k1_threads = args.pic_threads
        env._pic_fused = not args.pic_three_launches

    def barrier():
        if dist_on:
            dist.barrier()

    def timed_run(env, agent):
        """Pre-warm (not part of the contract's W): a FIXED number of steps (PREWARM_STEPS; at least two full re-sort /
        refresh periods), so that lazily created state, first launches and the clock ramp are behind us whatever --warmup says
        and every box times the same world steps.  Then W warm-up steps, then EXACTLY K timed steps between barrier +
        synchronize pairs, one HIP event after every step."""
        obs = env._get_current_obs
        period = max(args.sort_every if args.sort_every is not None else 8, 1)
        if dist_on:
            period = max(period, env.migrate_every)
        n_pre = max(args.prewarm, 2 * period + 1)
        for _ in range(n_pre):
            obs, res, *_ = env.step(agent.forward(obs))
        torch.cuda.synchronize()
        for _ in range(args.warmup):
            obs, res, *_ = env.step(agent.forward(obs))
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        inline = args.step_events_in_timed_region
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if inline:
            ev[0].record()
        for i in range(args.steps):
            obs, res, *_ = env.step(agent.forward(obs))
            if inline:
                ev[i + 1].record()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if not inline:
            # the per-step distribution (step_ms) from a SECOND pass of K steps, outside the timed region: an event after every
            # step is a marker packet between the step's last kernel and the next step's first one — it costs the loop ≈ 2 % that a
            # caller's loop does not pay (measured: profiles/r04_step_events_cost.txt)
            last = res.clone()
            ev[0].record()
            for i in range(args.steps):
                obs, res2, *_ = env.step(agent.forward(obs))
                ev[i + 1].record()
            torch.cuda.synchronize()
            res = last
        per_step = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
        return dt, res, per_step, n_pre

    dt, res, per_step, n_pre = timed_run(env, agent)
    last_reward, last_alive = env.read_result(res)
    env.check()                # (the tile-binned step's sticky error word: a loop that reads nothing back has to ask)
    K = env.agents.N
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # whole job: the world is `world` tiles of W×H cells, so one world step = `world` 4096²-grid steps
    steps_per_s = args.steps / dt * world
    med = per_step[len(per_step) // 2]
    line = {
        'metric': 'env steps/sec on 4096^2 Physarum grid', 'value': round(steps_per_s, 2), 'unit': 'env steps/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.fields, 'data': 'synthetic',
        'decomposed': bool(dist_on),
        'config': {'workload': f'PhysarumAgent {W}x{H} {args.fields} fields, agent ratio {args.ratio} (BASELINE configs[2]); '
                               "alive-only agent slots (max_agents='alive'), sync=False (the result words are read once, after the timed steps); "
                               'step = obs, ... = env.step(agent.forward(obs)); tile-binned path: the action stays in registers '
                               'and is re-derived bit-identically when read (see step_kind)',
                   'grid': [W, H], 'alive_agents': K, 'agent_slots': K, 'steps_per_rank': args.steps,
                   'parallelism': mode, 'prewarm_steps': n_pre, 'timed_world_steps': [n_pre + args.warmup + 1, n_pre + args.warmup + args.steps],
                   'step_ms_from': 'HIP events inside the timed region' if args.step_events_in_timed_region else 'a second pass of K steps with an event after every step (outside the timed region)',
                   'last_reward': round(last_reward, 3), 'last_num_agents': last_alive},
        # one HIP event after every timed step (rank 0's stream): device-side step times, host launch gaps included
        'step_ms': {'median': round(med, 4), 'mean': round(sum(per_step) / len(per_step), 4), 'min': round(per_step[0], 4),
                    'max': round(per_step[-1], 4), 'median_steps_per_s': round(1e3 / med * world, 1)},
    }
    if rank == 0 and dist_on:
        Kw = int(getattr(denv, 'world_agents', K * world))
        Bw = contract_bytes(W * H * world, Kw, 2 if args.fields == 'f16' else 4)
        line['config'].update(alive_agents=Kw, agent_slots=Kw, local_agents_rank0=K)
        if getattr(denv, 'ghost_fill', None) is not None:
            line['config']['ghost_message_fill_max_rank0'] = round(denv.ghost_fill, 3)
        # which step and which refresh the ranks took (rank 0's counts over the whole run: pre-warm, warm-up, timed steps)
        # (counts over the WHOLE run incl. the untimed per-step-event pass behind the timed steps)
        line['config'].update(steps_total_rank0=int(denv._steps), tile_binned_steps_rank0=int(getattr(denv, 'pic_steps', 0)),
                              refreshes_by_tiles_rank0=int(getattr(denv, 'tile_refreshes', 0)),
                              refreshes_under_the_next_steps_interior_rank0=int(getattr(denv, 'overlapped_refreshes', 0)),
                              refreshes_in_place_rank0=int(getattr(denv, 'inplace_refreshes', 0)),
                              band_tiles_packed_under_the_previous_step_rank0=int(getattr(denv, 'early_packs', 0)))
        line['roofline'] = {'bound': 'hbm', 'kernel': 'whole step (all ranks)', 'achieved': round(Bw / (dt / args.steps) / 1e9, 1),
                            'peak': HBM_PEAK_GBS * world, 'unit': 'GB/s', 'frac': round(Bw / (dt / args.steps) / 1e9 / (HBM_PEAK_GBS * world), 4),
                            'traffic': None, 'algorithmic_bytes_per_launch': Bw}
        print(json.dumps(line), flush=True)
    elif rank == 0:
        C = W * H
        binned = getattr(env, '_pic', None) is not None and env._pic.held is not None
        lazy_action = binned and env._pic.lazy_actions
        bf = 2 if args.fields == 'f16' else 4
        PMC_DTYPE[0] = '__half' if bf == 2 else 'float'
        Bc = contract_bytes(C, K, bf)
        B = algorithmic_bytes(C, K, action_stored=not lazy_action, bf=bf)
        line['config']['step_kind'] = (('tile-binned, ' + ('two' if env._pic.two_launch(env, agent) else 'three') + ' launches') if binned else 'classic') + \
            (', action kept in registers (re-derived bit-identically when read; --eager-actions stores it every step)' if lazy_action else '')
        if binned:
            line['config']['tile'] = [1 << env._pic.xs, 1 << env._pic.ys]
            line['config']['order_table'] = bool(env._pic.order is not None)     # die_pic.order: crowded tiles first inside every XCD band, rebuilt every 32nd step (DIE_PIC_ORDER=0: band mapping)
        if lazy_action:
            # the same loop with the action of every step stored, as round 1 did: reported beside the headline, not instead of it
            env._pic.flush_lazy()
            env._pic.lazy_actions = False
            o = env._get_current_obs
            n_e = max(args.steps, 20)
            for _ in range(10):
                o, *_ = env.step(agent.forward(o))
            torch.cuda.synchronize()
            t_e = time.perf_counter()
            for _ in range(n_e):
                o, *_ = env.step(agent.forward(o))
            torch.cuda.synchronize()
            stored_sps = n_e / (time.perf_counter() - t_e)
            line['config']['steps_per_s_with_the_action_stored_every_step'] = round(stored_sps, 1)
            env._pic.lazy_actions = True
        else:
            stored_sps = steps_per_s
        kt = time_kernels(env, agent, args.kernel_reps)
        empty_interval = kt.pop('_empty_event_interval', None)
        intervals = dict(kt)
        if empty_interval is not None:
            # tile-binned step: the figures are event-to-event intervals inside real steps; what an interval with nothing in
            # it costs (completion + dispatch of the event packets) is measured in the same loop and taken off, which
            # brings them onto rocprofv3's kernel durations (profiles/README.md)
            kt = {k: max(v - empty_interval, 0.0) for k, v in kt.items()}
        dom = max(kt, key=kt.get)
        ach = B[dom] / (kt[dom] * 1e-6) / 1e9
        traffic, traffic_src = pmc_traffic(dom, K)
        line['roofline'] = {
            'bound': 'hbm', 'kernel': dom, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_src,
            'avg_launch_us': round(kt[dom], 2), 'algorithmic_bytes_per_launch': B[dom],
            'kernels_us': {k: round(v, 2) for k, v in kt.items()},
            'kernels_gbs': {k: round(B[k] / (v * 1e-6) / 1e9, 1) for k, v in kt.items()},
            'copy_ceiling_gbs': round(copy_ceiling_gbs(device), 1),               # torch's uint8 copy_ (rounds 1-3 quoted this)
            'stream_ceiling_gbs': round(stream_ceiling_gbs(device), 1),           # die_stream_copy, 16 bytes per lane, through the C ABI
            'kernels_frac_of_peak': {k: round(B[k] / (v * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) for k, v in kt.items()},
            'empty_event_interval_us': None if empty_interval is None else round(empty_interval, 2),
            'kernel_event_intervals_us': {k: round(v, 2) for k, v in intervals.items()},
            # the whole step.  The contract's bytes B = 3·b_f·C + (60 + 11·b_f)·K (fp32 12·C + 104·K, fp16 6·C + 82·K)
            # include the action's hand-off (W 12 + R 12 per agent): `frac_contract_action_stored` prices them on the loop that
            # really stores the action every step (config.steps_per_s_with_the_action_stored_every_step) — THE figure to quote
            # against the contract.  `frac` = the bytes the headline's loop owes (the action stays in registers: 12 per agent
            # less) over the headline's time.  `frac_contract_bytes_over_headline_time` (rounds 2-4 called it frac_contract)
            # mixes the contract's bytes with the time of the loop that does not store the action: for comparison with the
            # earlier rounds only.
            'step': {'algorithmic_bytes_contract': Bc,
                     'frac_contract_action_stored': round(Bc * stored_sps / 1e9 / HBM_PEAK_GBS, 4),
                     'frac_contract_bytes_over_headline_time': round(Bc / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                     'algorithmic_bytes': B['step'],
                     'achieved': round(B['step'] / (dt / args.steps) / 1e9, 1),
                     'frac': round(B['step'] / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                     'frac_median_step': round(B['step'] / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        sc = line['roofline']['stream_ceiling_gbs']
        line['roofline']['kernels_frac_of_stream_ceiling'] = {k: round(v / sc, 4) for k, v in line['roofline']['kernels_gbs'].items()}
        # (read here, i.e. after the timed steps AND the untimed passes behind them — per-step events, stored actions, kernel timing)
        line['config']['chem_max_after_all_passes'] = round(float(env.medium.chem.float().max().item()), 4)
        if not args.no_extras:
            line['config']['side_measurements'] = side_measurements(args, device, agent_kw, torch, die_amd)
        if not args.no_cpu_baseline:
            v, cpu_dt = cpu_baseline(env, agent_kw, args.cpu_steps, args.seed)
            line['cpu_baseline'] = {
                'value': round(v, 4), 'unit': 'env steps/s', 'cores': 1, 'kind': 'port',
                'sample': f'{args.cpu_steps} steps of the same {W}x{H} workload ({K} alive-only slots) after 1 warm-up '
                          f'step, {cpu_dt:.1f} s; float64 numpy/scipy oracle (index arithmetic instead of the '
                          "reference's pandas label lookups, so faster than the reference itself); host has "
                          f'{os.cpu_count()} cores, numpy/scipy kernels on this path run on 1 thread',
                'other_sizes': other_cpu_baselines(args.seed)}
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
