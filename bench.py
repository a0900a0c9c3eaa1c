#!/usr/bin/env python3
"""bench.py — env steps/sec of the Physarum grid step (agent.forward + env.step) on MI355X.

Contract (see the task brief): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON
line on rank 0.  A "step" is one PhysarumAgent.forward + one Env.step (action handed over in
HBM, as the Gym API does) over a synthetic 4096x4096 fp32 grid (BASELINE.json configs[2]) with
the state already resident in HBM.  Next to it: `roofline` for the dominant kernel (HIP-event
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
    p.add_argument('--sort-every', type=int, default=8)
    p.add_argument('--fields', choices=['f32', 'f16'], default='f32', help='dtype of the field channels (configs[4] uses f16)')
    p.add_argument('--migrate-every', type=int, default=8, help='decomposed runs: agents / ghosts cross ranks every M steps (1 = every step)')
    p.add_argument('--dist-mode', choices=['ghost', 'guard'], default='ghost',
                   help='decomposed runs: ghost = communication-avoiding (ghost agents, nothing crosses ranks for M steps); '
                        'guard = claims merged and halos exchanged every step, strays handed over every M steps')
    p.add_argument('--force-dist', action='store_true', help='use the decomposed path even on one rank (testing)')
    return p.parse_args()


def algorithmic_bytes(C, K):
    """DESIGN.md §5 / SURVEY.md §8(d), fp32 fields: B = 12·C + 104·K per env step; per kernel, what an ideal
    version of that kernel has to move."""
    return {
        # fused forward + move + claim + feed: x,y R+W 16, heading R+W 8, agent_food R+W 8, action W 12,
        # 6 gathers (4 chem taps, food at old and new cell) 24, claim 4
        'k_forward_move_claim': 72 * K,
        'k_gradient_forward': 48 * K,     # stand-alone forward: x,y,heading R 12 + heading W 4 + action W 12 + 5 gathers 20
        'k_move_claim': 44 * K,           # stand-alone: x,y RW 16 + agent_food RW 8 + action R 12 + food gather 4 + claim 4
        'k_diffuse_rows_fused': 8 * C + 20 * K,   # chem R + W per cell; per agent chem RMW 8 + food RMW 8 + mark 4
        'step': 12 * C + 104 * K,
    }


PMC_FILE = os.path.join(ROOT, 'profiles', 'r01_v7_pmc_traffic_per_kernel_avg.json')
PMC_NAMES = {'k_gradient_forward': 'void k_gradient_forward<float, 1>', 'k_move_claim': 'void k_move_claim<float>',
             'k_forward_move_claim': 'void k_forward_move_claim<float, 1, false>',
             'k_diffuse_rows_fused': 'void k_diffuse_rows<float, 2, true, true>'}


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same
    command (profiles/README.md): (FETCH_SIZE + WRITE_SIZE) KiB.  FETCH_SIZE under-counts wide coalesced
    streams by 2x on gfx950 (applied to the diffusion sweep only); None if the file is absent."""
    try:
        c = json.load(open(PMC_FILE))[PMC_NAMES[kernel]]
        fetch = c['FETCH_SIZE'] * (2 if kernel == 'k_diffuse_rows_fused' else 1)
        return int((fetch + c['WRITE_SIZE']) * 1024)
    except Exception:
        return None


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

    env.sort_agents()                     # the timed loop re-sorts every few steps: measure in that regime
    obs = env._get_current_obs
    sort_every, env._sort_every = env._sort_every, 0

    import ctypes as C
    from die_amd.device_array import _ptr, stream_ptr

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


def main():
    args = parse()
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import datetime
        import torch.distributed as dist
        backend = os.environ.get('DIE_DIST_BACKEND', 'nccl')      # 'gloo': rehearsal with several ranks on one GPU
        if torch.cuda.device_count() == 1:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        kw = dict(device_id=torch.device(f'cuda:{local_rank}')) if backend == 'nccl' else {}
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300), **kw)
    elif args.gpus > 1:
        sys.exit('launch N>1 with torch.distributed.run (one rank per GPU)')
    device = torch.device(f'cuda:{local_rank}')
    torch.cuda.set_device(device)

    import die_amd
    W = H = args.size
    agent_kw = dict(scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1), turn_angle=30, sense_angle=90,
                    turn_tolerance=0.1, deposit=4.0)
    GRIDS = {1: (1, 1), 2: (1, 2), 4: (2, 2), 8: (2, 4)}
    mode, denv = 'single GPU', None
    if dist_on:
        # weak scaling: every rank owns one W×H tile of a (W·Px)×(H·Py) torus — 2-D domain decomposition with
        # chem-halo exchange and agent migration over RCCL point-to-point (die_amd/dist.py)
        grid = GRIDS.get(world) or (1, world)
        try:
            from die_amd.dist import DistEnv
            gW, gH = W * grid[0], H * grid[1]
            # same cell-unit parameters as the single-GPU workload (10.2-cell probe, 1.53-cell step)
            # (a non-square world is anisotropic in cells, offsets being fractions of the unit square: size them on the longer axis)
            agent_kw.update(scale=1.53 / (max(gW, gH) - 1), sense_offset=10.2 / (max(gW, gH) - 1))
            denv = DistEnv((gW, gH), grid, die_amd.Dynamics(init_agent_ratio=args.ratio), probe_reach=11,
                           device=device, seed=args.seed, sort_every=args.sort_every,
                           migrate_every=args.migrate_every, max_step_cells=1.6, ghosts=args.dist_mode == 'ghost',
                           ghost_headroom=1.3)     # the synthetic world stays uniform (measured fill 0.5 of 2x over 1200 steps)
            how = (f'ghost agents, halo ({denv.geo.hx}, {denv.geo.hy}) re-seated every {denv.migrate_every} steps' if denv.ghosts else
                   f'halo {denv.geo.h}, claims merged every step, strays handed over every {denv.migrate_every} steps')
            mode = (f'{grid[0]}x{grid[1]} domain decomposition of a {gW}x{gH} torus, {how}, '
                    f'{"RCCL" if backend == "nccl" else backend} point-to-point')
        except Exception as e:           # keep the scaling run alive: independent replicas, and say so
            denv = None
            mode = f'{world} independent grid replicas (decomposition unavailable: {type(e).__name__}: {e})'
    if denv is not None:
        # trial steps before committing to the decomposed path: a transport that fails at the first exchanges must
        # not cost the scaling run its line (all ranks agree on the outcome; on failure: replicas, and say so)
        ok, why = 1, ''
        try:
            trial_agent = die_amd.PhysarumAgent(max_agents=denv.capacity, seed=args.seed, **agent_kw)   # one seed: streams are keyed by world slot id
            o = denv._get_current_obs
            for _ in range(2 * denv.migrate_every + 1):          # covers two refreshes / hand-overs over the real transport
                o, res, *_ = denv.step(trial_agent.forward(o))
            denv.read_result(res)                                # ghost mode: every world agent has exactly one owner
            torch.cuda.synchronize()
        except Exception as e:
            ok, why = 0, f'{type(e).__name__}: {e}'
        try:
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        except Exception as e:
            ok, why = 0, why or f'{type(e).__name__}: {e}'
        if ok:
            agent = trial_agent
        else:
            denv = None
            agent_kw.update(scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
            mode = f'{world} independent grid replicas (decomposed step failed: {why or "on another rank"})'
    def barrier():
        if dist_on:
            try:
                dist.barrier()
            except Exception:
                if denv is not None:       # the decomposed run needs it; replicas after a transport failure do not
                    raise

    def timed_run(env, agent):
        obs = env._get_current_obs
        results = []
        for _ in range(args.warmup):
            obs, res, *_ = env.step(agent.forward(obs))
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            obs, res, *_ = env.step(agent.forward(obs))
            results.append(res)
        torch.cuda.synchronize()
        barrier()
        return time.perf_counter() - t0, results

    def replica():
        e = die_amd.Env((W, H), die_amd.Dynamics(init_agent_ratio=args.ratio), seed=args.seed + rank,
                        max_agents='alive', device=device, sync=False, sort_every=args.sort_every,
                        field_dtype=torch.float16 if args.fields == 'f16' else torch.float32)
        return e, die_amd.PhysarumAgent(max_agents=e.agents.N, seed=args.seed + rank, **agent_kw)

    if denv is not None:
        env = denv
        ok, why = 1, ''
        try:
            dt, results = timed_run(env, agent)
            last_reward, last_alive = env.read_result(results[-1])
        except Exception as e:                                   # e.g. a ghost refresh that overflows its messages
            ok, why = 0, f'{type(e).__name__}: {e}'
        try:
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        except Exception as e:
            ok, why = 0, why or f'{type(e).__name__}: {e}'
        if not ok:
            denv = None
            agent_kw.update(scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
            mode = f'{world} independent grid replicas (decomposed run failed: {why or "on another rank"})'
    if denv is None:
        env, agent = replica()
        dt, results = timed_run(env, agent)
        last_reward, last_alive = env.read_result(results[-1])
    K = env.agents.N
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # whole job: the world is `world` tiles of W×H cells, so one world step = `world` 4096²-grid steps
    steps_per_s = args.steps / dt * world
    line = {
        'metric': 'env steps/sec on 4096^2 Physarum grid', 'value': round(steps_per_s, 2), 'unit': 'env steps/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.fields, 'data': 'synthetic',
        'config': {'workload': f'PhysarumAgent {W}x{H} {args.fields} fields, agent ratio {args.ratio} (BASELINE configs[2]); '
                               'step = PhysarumAgent.forward + Env.step with the action handed over in HBM',
                   'grid': [W, H], 'alive_agents': K, 'agent_slots': K, 'steps_per_rank': args.steps,
                   'parallelism': mode,
                   'last_reward': round(last_reward, 3), 'last_num_agents': last_alive},
    }
    if rank == 0 and denv is not None:
        Kw = int(getattr(denv, 'world_agents', K * world))
        Bw = (12 * W * H * world + 104 * Kw)
        line['config'].update(alive_agents=Kw, agent_slots=Kw, local_agents_rank0=K)
        if getattr(denv, 'ghost_fill', None) is not None:
            line['config']['ghost_message_fill_max_rank0'] = round(denv.ghost_fill, 3)
        line['roofline'] = {'bound': 'hbm', 'kernel': 'whole step (all ranks)', 'achieved': round(Bw / (dt / args.steps) / 1e9, 1),
                            'peak': HBM_PEAK_GBS * world, 'unit': 'GB/s', 'frac': round(Bw / (dt / args.steps) / 1e9 / (HBM_PEAK_GBS * world), 4),
                            'traffic': None, 'algorithmic_bytes_per_launch': Bw}
        print(json.dumps(line), flush=True)
    elif rank == 0:
        C = W * H
        B = algorithmic_bytes(C, K)
        kt = time_kernels(env, agent, args.kernel_reps)
        dom = max(kt, key=kt.get)
        ach = B[dom] / (kt[dom] * 1e-6) / 1e9
        line['roofline'] = {
            'bound': 'hbm', 'kernel': dom, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': pmc_traffic(dom),
            'traffic_source': 'profiles/r01_v7_pmc_traffic_per_kernel_avg.json (separate --pmc passes of this command)',
            'avg_launch_us': round(kt[dom], 2), 'algorithmic_bytes_per_launch': B[dom],
            'kernels_us': {k: round(v, 2) for k, v in kt.items()},
            'kernels_gbs': {k: round(B[k] / (v * 1e-6) / 1e9, 1) for k, v in kt.items()},
            'copy_ceiling_gbs': round(copy_ceiling_gbs(device), 1),
            'step': {'algorithmic_bytes': B['step'],
                     'achieved': round(B['step'] / (dt / args.steps) / 1e9, 1),
                     'frac': round(B['step'] / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            v, cpu_dt = cpu_baseline(env, agent_kw, args.cpu_steps, args.seed)
            line['cpu_baseline'] = {
                'value': round(v, 4), 'unit': 'env steps/s', 'cores': 1, 'kind': 'port',
                'sample': f'{args.cpu_steps} steps of the same {W}x{H} workload ({K} alive-only slots) after 1 warm-up '
                          f'step, {cpu_dt:.1f} s; float64 numpy/scipy oracle (index arithmetic instead of the '
                          "reference's pandas label lookups, so faster than the reference itself); host has "
                          f'{os.cpu_count()} cores, numpy/scipy kernels on this path run on 1 thread'}
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
