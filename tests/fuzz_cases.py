"""Randomised parity sweeps (GPU vs oracle, and device paths vs each other).  `python -m tests.fuzz_cases
step|forward|paths|init|binned [cases] [seed]` runs a long sweep; tests/test_gpu_fuzz.py runs short ones."""
import os
import sys

import numpy as np
import torch

import die_amd
from oracle import cpu_ref as R
from tests.test_gpu_parity import f32, quantised_action, random_state, ref_dyn


def fuzz_step(n_cases=100, seed=0, verbose=True):
    rs = np.random.RandomState(seed)
    fails = 0
    skipped = 0
    for case in range(n_cases):
        W = int(rs.choice([2, 3, 4, 5, 7, 8, 12, 16, 31, 32, 33, 64, 100, 128, 250, 256]))
        H = int(rs.choice([2, 3, 4, 8, 12, 16, 20, 36, 60, 64, 128, 244, 248, 252, 256, 260, 500, 512]))
        N = int(rs.choice([1, 2, 5, 64, 257, 1000, 4096, 20000]))
        K = int(rs.randint(0, N + 1))
        sigma = float(rs.choice([0.3, 0.5, 0.8, 1.0, 1.2]))
        dyn = die_amd.Dynamics(boundary=die_amd.BoundaryCondition(rs.choice(['wrap', 'limit'])), food_infinite=bool(rs.rand() < 0.2),
                               agents_die=bool(rs.rand() < 0.2), op_action_cost=die_amd.zero_cost if rs.rand() < 0.2 else die_amd.linear_action_cost,
                               diffuse_sigma=sigma, rate_feed=float(rs.choice([0.1, 0.5])), rate_decay_chem=float(rs.choice([0.0, 0.1, 0.3])),
                               diffuse_mode=str(rs.choice(['wrap', 'wrap', 'wrap', 'nearest', 'reflect', 'mirror', 'constant'])))
        sort_every = int(rs.choice([0, 1]))
        try:
            medium, agents = random_state(W, H, N, K, rs, collide=float(rs.choice([0.0, 0.3, 0.9])))
            action = quantised_action(N, rs, float(rs.choice([0.5 / max(W, H), 3.0 / max(W, H), 0.4])))
            rd = ref_dyn(dyn)
            for f in ('rate_feed', 'rate_decay_chem', 'diffuse_sigma'):
                setattr(rd, f, float(np.float32(getattr(rd, f))))
            ref = R.RefEnv(medium, agents, rd)
            env = die_amd.Env.from_numpy(medium, agents, dyn, sort_every=sort_every)
            for step in range(2):
                _, want_r, want_t, _, want_i = ref.step(action)
                _, r, t, _, i = env.step(action)
            ga, gm = env.agents.to_numpy(), env.medium.to_numpy()
            tol_xy = 2.0 ** -32 if dyn.boundary.value == 'limit' else 0.0
            assert np.abs(ga[:2] - ref.agents[:2]).max() <= tol_xy, 'xy'
            assert np.array_equal(ga[2], ref.agents[2]), 'alive'
            assert np.array_equal(gm[0], ref.medium[0]), 'agents channel'
            assert i['num_agents'] == want_i['num_agents'] and t == want_t, 'info'
            assert np.allclose(ga[3], ref.agents[3], rtol=1e-5, atol=2e-7), 'agent_food'
            assert np.allclose(gm[1], ref.medium[1], rtol=1e-5, atol=1e-8), 'food'
            assert np.allclose(gm[2], ref.medium[2], rtol=2e-5, atol=2e-7), 'chem'
            assert abs(r - want_r) <= 1e-5 * np.abs(ref.last_gained).sum() + 1e-9, 'reward'
        except NotImplementedError as e:
            skipped += 1
            continue
        except Exception as e:
            fails += 1
            print(f'CASE {case} FAILED W={W} H={H} N={N} K={K} sigma={sigma} dyn={dyn} sort={sort_every}: {type(e).__name__} {e}', flush=True)
            if fails > 10:
                break
    return fails


def fuzz_forward(n_cases=100, seed=0, verbose=True):
    rs = np.random.RandomState(seed)
    fails = 0
    worst = 0.0
    for case in range(n_cases):
        W = int(rs.choice([2, 5, 16, 33, 64, 200])); H = int(rs.choice([2, 7, 12, 64, 130, 256]))
        N = int(rs.choice([1, 50, 3000, 20000]))
        medium, agents = random_state(W, H, N, int(0.7 * N), rs)
        phys = rs.rand() < 0.6
        kw = dict(scale=float(rs.choice([0.001, 0.01, 0.05])), deposit=float(rs.choice([1.0, 4.0, 4.5])),
                  sense_offset=float(rs.choice([0.0, 0.01, 0.04, 0.3])), normalized_grad=bool(rs.rand() < 0.8),
                  grad_clip=None if rs.rand() < 0.2 else float(rs.choice([1e-5, 1e-3])))
        if phys:
            kw.update(turn_angle=int(rs.choice([20, 30, 35, 45])), sense_angle=int(rs.choice([60, 100, 120])), turn_tolerance=float(rs.choice([0.05, 0.1, 0.2])),
                      inertia=float(rs.choice([0.0, 0.0, 0.5])), noise_scale=float(rs.choice([0.0, 0.0, 0.02])))
            ref = R.RefPhysarumAgent(N, seed=case, **kw); dev = die_amd.PhysarumAgent(max_agents=N, seed=case, **kw)
        else:
            kw.update(inertia=float(rs.choice([0.0, 0.9])), noise_scale=float(rs.choice([0.0, 0.025])))
            ref = R.RefGradientAgent(N, seed=case, **kw); dev = die_amd.GradientAgent(max_agents=N, seed=case, **kw)
        prev = f32(rs.normal(0, .4, (2, N)))
        ref._prev_grad = prev.copy()
        dir0 = f32(ref._direction_rads); ref._direction_rads = dir0.copy()
        want = ref.forward((agents, medium))
        env = die_amd.Env.from_numpy(medium, agents)
        dev.set_state(dir0, prev if kw['inertia'] else None)
        got = dev.forward(env._get_current_obs).to_numpy()
        atol = 1e-6 * kw['scale'] + 1e-9
        bad = ~(np.isclose(got[0], want[0], rtol=1e-5, atol=atol) & np.isclose(got[1], want[1], rtol=1e-5, atol=atol) & np.isclose(got[2], want[2], rtol=1e-5, atol=1e-8))
        frac = bad.mean(); worst = max(worst, frac if N >= 3000 else 0)
        if bad.sum() > max(3, 3e-3 * N):
            fails += 1
            print(f'CASE {case} W={W} H={H} N={N} phys={phys} kw={kw}: {bad.sum()} bad of {N}', flush=True)
    return fails


def fuzz_paths(n_cases=100, seed=0, verbose=True):
    rs = np.random.RandomState(seed)
    fails = 0
    for case in range(n_cases):
        W = int(rs.choice([4, 16, 33, 64, 128])); H = int(rs.choice([4, 8, 12, 64, 244, 252, 256]))
        N = int(rs.choice([5, 300, 4000])); K = int(rs.randint(0, N + 1))
        medium, agents = random_state(W, H, N, K, rs, collide=float(rs.choice([0.0, 0.5])))
        dyn = dict(boundary=die_amd.BoundaryCondition(rs.choice(['wrap', 'limit'])), agents_die=bool(rs.rand() < 0.3), food_infinite=bool(rs.rand() < 0.2),
                   diffuse_sigma=float(rs.choice([0.5, 0.8])))
        phys = rs.rand() < 0.7
        f16 = bool(rs.rand() < 0.3)
        turn = np.radians(30); dir0 = f32(np.floor(rs.uniform(-np.pi, np.pi, N) / turn) * turn); prev = f32(rs.normal(0, .4, (2, N)))
        outs = []
        for variant in ('default', 'staged', 'eager', 'sorted'):
            env = die_amd.Env.from_numpy(medium, agents, die_amd.Dynamics(**dyn), sort_every=1 if variant == 'sorted' else 0,
                                         staged=variant == 'staged', field_dtype=torch.float16 if f16 else torch.float32)
            if phys: ag = die_amd.PhysarumAgent(max_agents=N, seed=3, scale=2.0 / max(W, H), sense_offset=0.05)
            else: ag = die_amd.GradientAgent(max_agents=N, seed=3, scale=0.01, sense_offset=0.03, inertia=0.8, noise_scale=0.02)
            ag.set_state(dir0, None if phys else prev)
            ag.lazy = variant != 'eager'
            obs = env._get_current_obs
            held = []
            for _ in range(4):
                a = ag.forward(obs)
                held.append(a)
                obs, *_ = env.step(a)
            outs.append((env.medium.to_numpy(), env.agents.to_numpy(), ag.direction_rads_numpy(), np.stack([a.to_numpy() for a in held])))
        for v, o in zip(('staged', 'eager', 'sorted'), outs[1:]):
            for a, b in zip(outs[0], o):
                if not np.array_equal(a, b):
                    fails += 1; print(f'CASE {case} variant {v} differs: W={W} H={H} N={N} K={K} phys={phys} f16={f16} dyn={dyn}', flush=True); break
    return fails


def fuzz_binned(n_cases=60, seed=0, verbose=True):
    """Tile-binned step against the classic step on random worlds: tile shapes, world shapes (3..6 tiles per axis), agent
    densities from sparse to several agents per cell, boundaries, rates, fp16 fields, step lengths up to the tile limit,
    actions read every step / now and then / never, switches between the two paths mid-run.  Everything must be equal bit for
    bit: fields, agents, headings, actions, rewards."""
    rs = np.random.RandomState(seed)
    fails = 0
    for case in range(n_cases):
        xs, ys = [(4, 5), (5, 6), (6, 6), (5, 7)][rs.randint(4)]
        TX, TY = 1 << xs, 1 << ys
        W, H = TX * int(rs.randint(3, 6)), TY * int(rs.randint(3, 5))
        if rs.rand() < 0.4:                          # 8 .. 17 tiles per row: the workgroup → tile mapping by XCD bands (+ its remainder columns) is on the path
            W, H = TX * 3, TY * int(rs.randint(8, 18))
        N = int(rs.choice([50, 2000, 20000, W * H // 2]))
        reach_pick = int(rs.randint(3))
        # dead slots behind the segments (the reference's default slot layout) / a GradientAgent's momentum: short steps only
        dead = float(rs.choice([0.3, 0.6, 0.85])) if (rs.rand() < 0.3 and reach_pick < 2 and N >= 2000) else 0.0
        medium, agents = random_state(W, H, N, N - int(dead * N), rs, collide=float(rs.choice([0.0, 0.3, 0.9])))
        f16 = bool(rs.rand() < 0.3)
        dyn = dict(boundary=die_amd.BoundaryCondition(rs.choice(['wrap', 'limit'])), food_infinite=bool(rs.rand() < 0.2),
                   diffuse_sigma=float(rs.choice([0.4, 0.5, 0.8, 1.0])), rate_feed=float(rs.choice([0.1, 0.35])),
                   rate_decay_chem=float(rs.choice([0.01, 0.2])))
        reach = [0.7, 1.53, min(TX, TY) - 1.001][reach_pick]                          # cells per step
        probe = float(rs.choice([1.2, 10.2, 21.5]))
        kw = dict(scale=reach / (max(W, H) - 1), sense_offset=probe / (max(W, H) - 1), sense_angle=float(rs.choice([60, 90, 120])),
                  deposit=float(rs.choice([1.0, 4.0])))
        turn = np.radians(30); dir0 = f32(np.floor(rs.uniform(-np.pi, np.pi, N) / turn) * turn)
        read_mode = rs.choice(['every', 'some', 'never'])
        switch_at = int(rs.randint(2, 7)) if rs.rand() < 0.4 else None
        poke = rs.choice([0, 0, 0, 1, 2, 3, 4], size=8)
        se = int(rs.choice([0, 2, 3]))
        gradient = bool(rs.rand() < 0.3)             # GradientAgent: the binned step stores its action
        inertia, noise = 0.0, 0.0
        if gradient and not dead and reach_pick < 2 and rs.rand() < 0.6:              # … with momentum: _prev_grad travels through the layouts
            inertia, noise = float(rs.choice([0.0, 0.5, 0.9])), float(rs.choice([0.0, 0.025]))
        prev0 = f32(np.clip(rs.randn(2, N) * 0.4, -2.6, 2.6))
        two_agents = bool(rs.rand() < 0.2)           # two agent objects take turns on one env (each with its own headings)
        fused = bool(rs.rand() < 0.8) or dead > 0    # two-launch form (one field kernel per tile) / three launches (dead slots: two-launch form only)
        only = os.environ.get('FUZZ_ONLY')
        if only is not None and int(only) != case:
            continue
        if only is not None:
            print(f'case {case}: W={W} H={H} N={N} tile=({xs},{ys}) f16={f16} read={read_mode} switch={switch_at} poke={poke.tolist()} sort_every={se} gradient={gradient} two={two_agents} fused={fused} dead={dead} inertia={inertia} noise={noise}', flush=True)
        outs = []
        for pic in (True, False):
            env = die_amd.Env.from_numpy(medium, agents, die_amd.Dynamics(**dyn), sort_every=se if pic else 3, pic=pic,
                                         field_dtype=torch.float16 if f16 else torch.float32)
            env._pic_tile = (xs, ys) if pic else None
            env._pic_fused = fused
            def make(sd):
                if gradient:
                    g = die_amd.GradientAgent(max_agents=N, seed=sd, scale=kw['scale'], sense_offset=kw['sense_offset'], deposit=kw['deposit'],
                                              inertia=inertia, noise_scale=noise, normalized_grad=True)
                    g.set_state(dir0, prev_grad=prev0.copy() if inertia else None)
                    return g
                g = die_amd.PhysarumAgent(max_agents=N, seed=sd, **kw)
                g.set_state(dir0)
                return g
            ags = [make(7), make(8)] if two_agents else [make(7)]
            obs = env._get_current_obs
            acts, rewards, held = [], [], []
            for i in range(8):
                if pic and switch_at is not None:
                    env._pic_enabled = not (switch_at <= i < switch_at + 2)      # two classic steps in between
                ag = ags[(i // 2) % len(ags)]
                a = ag.forward(obs)
                try:
                    obs, rew, _, _, info = env.step(a)
                except RuntimeError as e:
                    fails += 1
                    print(f'CASE {case} step {i} pic={pic}: {e}: W={W} H={H} N={N} tile=({xs},{ys}) f16={f16} dyn={dyn} kw={kw} reach={reach} probe={probe} '
                          f'switch={switch_at}', flush=True)
                    acts = None                                 # (no accessor after a bookkeeping error: slot ids may be garbage)
                    break
                rewards.append((rew, info['num_agents']))
                if pic and poke[i] == 1: env.medium.occupied()                   # observers in between: the claim plane is rebuilt,
                elif pic and poke[i] == 2: env.agents.to_numpy()                 # the arrays read,
                elif pic and poke[i] == 3: env.sort_agents()                     # re-ordered behind the binned step's back,
                elif pic and poke[i] == 4: env.render_rgb8()                     # a frame rendered
                if read_mode == 'every' or (read_mode == 'some' and i % 3 == 0):
                    acts.append(a.to_numpy())
                elif read_mode == 'some' and i % 3 == 1:
                    held.append(a)
            if acts is None:
                outs = None
                break
            acts += [a.to_numpy() for a in held]
            if pic and (env._pic is None or (switch_at is None and poke[7] != 3 and env._pic.held[0] is not env.agents.x)):
                fails += 1; print(f'CASE {case}: the binned path did not run (W={W} H={H} tile=({xs},{ys}) reach={reach})', flush=True)
            outs.append((env.medium.to_numpy(), env.agents.to_numpy(), np.stack([g.direction_rads_numpy() for g in ags]), np.array(rewards),
                         np.stack(acts) if acts else np.zeros(0)))
        for name, a, b in zip(('medium', 'agents', 'heading', 'rewards', 'actions'), *(outs or ((), ()))):
            if a.shape != b.shape or not np.array_equal(a, b):
                fails += 1
                if name == 'actions' and a.shape == b.shape:
                    name = 'actions ' + str([bool(np.array_equal(a[k], b[k])) for k in range(a.shape[0])])
                if name == 'medium' and a.shape == b.shape:
                    name = 'medium ' + str([bool(np.array_equal(a[c], b[c])) for c in range(3)]) + f' max chem diff {np.abs(a[2] - b[2]).max():.3g} food diff {np.abs(a[1] - b[1]).max():.3g}'
                print(f'CASE {case} {name} differs: W={W} H={H} N={N} tile=({xs},{ys}) f16={f16} gradient={gradient} two={two_agents} dead={dead} inertia={inertia} noise={noise} dyn={dyn} kw={kw} read={read_mode} switch={switch_at}', flush=True)
                break
        if verbose and case % 10 == 9:
            print(f'  binned: {case + 1} cases, {fails} failures', flush=True)
    return fails


def fuzz_batched(n_cases=30, seed=0, verbose=True):
    """Batched replicas (die_forward_env_step_batch) against stand-alone runs: random shapes (incl. fields whose rows are
    not a multiple of 4 — the batch refuses those cleanly), replica counts, densities, boundaries, rates, fp16."""
    from die_amd.batch import BatchedEnv, BatchedPhysarumAgent
    rs = np.random.RandomState(seed)
    fails = 0
    for case in range(n_cases):
        W = int(rs.choice([8, 33, 64, 100, 256])); H = int(rs.choice([8, 12, 64, 200, 252]))
        R = int(rs.choice([1, 2, 3, 7, 16])); f16 = bool(rs.rand() < 0.3)
        dyn = dict(init_agent_ratio=float(rs.choice([0.02, 0.15, 0.6])), boundary=die_amd.BoundaryCondition(rs.choice(['wrap', 'limit'])),
                   food_infinite=bool(rs.rand() < 0.2), diffuse_sigma=float(rs.choice([0.4, 0.5, 0.8, 1.0])),
                   rate_feed=float(rs.choice([0.1, 0.35])), rate_decay_chem=float(rs.choice([0.01, 0.2])))
        kw = dict(scale=float(rs.choice([0.7, 1.53, 3.0])) / (max(W, H) - 1), sense_offset=float(rs.choice([1.2, 10.2])) / (max(W, H) - 1),
                  sense_angle=float(rs.choice([60, 90, 120])))
        steps, s0, a0 = int(rs.choice([3, 9])), int(rs.randint(1000)), int(rs.randint(1000))
        dt = torch.float16 if f16 else torch.float32
        try:
            benv = BatchedEnv((W, H), die_amd.Dynamics(**dyn), replicas=R, seed=s0, field_dtype=dt)
            bag = BatchedPhysarumAgent(benv, seed=a0, **kw)
            rew, alive = BatchedEnv.read_results(benv.run(bag, steps))
        except NotImplementedError as e:
            if verbose: print(f'  case {case}: refused ({str(e)[:80]})', flush=True)
            continue
        for r in range(R):
            env = die_amd.Env((W, H), die_amd.Dynamics(**dyn), seed=s0 + r, max_agents='alive', field_dtype=dt, pic=False)
            ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=a0 + r, **kw)
            obs = env._get_current_obs
            want = []
            for _ in range(steps):
                obs, rw, _, _, info = env.step(ag.forward(obs))
                want.append((rw, info['num_agents']))
            m, a = benv.replica_numpy(r)
            ok = np.array_equal(m, env.medium.to_numpy()) and np.array_equal(a, env.agents.to_numpy()) and \
                np.array_equal(bag.direction_rads_numpy(r), ag.direction_rads_numpy()) and \
                np.array_equal(rew[:, r], np.array([w[0] for w in want])) and np.array_equal(alive[:, r], np.array([w[1] for w in want]))
            if not ok:
                fails += 1
                print(f'CASE {case} replica {r} differs: W={W} H={H} R={R} f16={f16} dyn={dyn} kw={kw} steps={steps}', flush=True)
                break
        if verbose and case % 10 == 9:
            print(f'  batched: {case + 1} cases, {fails} failures', flush=True)
    return fails


def fuzz_nca(n_cases=40, seed=0, verbose=True):
    """NeuralAutomataAgent.forward on the device against the oracle: odd field shapes (rows not a multiple of 4, fields smaller
    than the kernel), random kernel stacks, fp16 fields, with / without the agents channel."""
    import torch as th
    rs = np.random.RandomState(seed)
    fails = 0
    for case in range(n_cases):
        W = int(rs.choice([2, 3, 5, 13, 16, 37, 64, 130])); H = int(rs.choice([2, 4, 7, 9, 30, 64, 66, 257]))
        ks = tuple(int(k) for k in rs.choice([1, 3, 5, 7], size=int(rs.randint(1, 4))))
        N = max(W * H // 3, 2); K = max(W * H // 5, 1)
        medium, agents = random_state(W, H, N, K, rs, collide=0.2)
        f16 = bool(rs.rand() < 0.3); with_agents = bool(rs.rand() < 0.6)
        th.manual_seed(int(rs.randint(10 ** 6)))
        ag = die_amd.NeuralAutomataAgent(scale=0.07, deposit=1.5, with_agent_channel=with_agents, kernel_sizes=ks)
        ag.model.init_weights()
        env = die_amd.Env.from_numpy(medium, agents, field_dtype=th.float16 if f16 else th.float32)
        action = ag.forward(env._get_current_obs)
        ws = [k.weight.detach().numpy().astype(np.float64) for k in ag.model.conv_layers()]
        m = env.medium.to_numpy()                                   # (fp16 fields: the oracle sees the rounded values)
        want = R.nca_forward((agents, m), ws, 0.07, 1.5, with_agents)
        got = action.to_numpy()
        if not np.allclose(got, want, rtol=1e-5, atol=2e-5):
            fails += 1
            print(f'CASE {case}: W={W} H={H} kernels={ks} f16={f16} agents_channel={with_agents} max err {np.abs(got - want).max():.3g}', flush=True)
        if verbose and case % 10 == 9:
            print(f'  nca: {case + 1} cases, {fails} failures', flush=True)
    return fails


def fuzz_init(n_cases=100, seed=0, verbose=True):
    rs = np.random.RandomState(seed)
    fails = 0
    for case in range(n_cases):
        W = int(rs.choice([2, 3, 17, 64, 100, 255, 256, 513])); H = int(rs.choice([2, 5, 12, 64, 127, 128, 300, 1024]))
        ratio = float(rs.choice([0.0, 0.001, 0.05, 0.15, 0.5, 1.0])); seed = int(rs.randint(0, 2 ** 31)) * int(rs.choice([1, 2 ** 20 + 7]))
        try:
            want_m, want_a = R.synthetic_init(W, H, ratio, seed)
            K = int(want_m[0].sum())
            if K == 0:
                continue
            env = die_amd.Env((W, H), die_amd.Dynamics(init_agent_ratio=ratio), seed=seed)
            m, a = env.medium.to_numpy(), env.agents.to_numpy()
            assert np.array_equal(m[0], want_m[0]), 'seeding'
            assert env._num_seeded == K
            assert np.abs(a[:2] - want_a[:2]).max() <= 2.0 ** -32, 'xy'
            assert np.array_equal(a[2], want_a[2]) and np.allclose(a[3], want_a[3], rtol=1e-6), 'alive/food'
            d = np.abs(m[1] - want_m[1]); assert (d > 1e-6).mean() <= 1e-4 and d.max() <= 1.001e-3, 'food field'
            ag = die_amd.PhysarumAgent(max_agents=W * H, seed=seed); ag._alloc_state('cuda:0')
            ref = R.RefPhysarumAgent(W * H, seed=seed)
            assert np.mean(np.abs(ag.direction_rads_numpy() - ref._direction_rads) > 1e-6) <= 2e-4, 'heading'
        except Exception as e:
            fails += 1; print(f'CASE {case} W={W} H={H} ratio={ratio} seed={seed}: {type(e).__name__} {e}', flush=True)
    return fails



if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'step'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    f = {'step': fuzz_step, 'forward': fuzz_forward, 'paths': fuzz_paths, 'init': fuzz_init, 'binned': fuzz_binned, 'batched': fuzz_batched, 'nca': fuzz_nca}[which](n, seed)
    print(f'{which}: {n} cases, {f} failures')
