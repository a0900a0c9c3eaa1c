"""Randomised decomposed-vs-single-device equality (ranks share the GPU, gloo)."""
import os, sys, socket, tempfile; sys.path.insert(0, '.')
import numpy as np, torch

def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); return s.getsockname()[1]

GHOST = os.environ.get('FUZZ_GHOST', '0') == '1'
PIC = os.environ.get('FUZZ_PIC', '0') == '1'          # ghost mode on worlds whose ranks qualify for the tile-binned step

def make_case(seed):
    rs = np.random.RandomState(seed)
    if PIC:
        grid = [(1, 2), (2, 1), (2, 2), (1, 3), (1, 1)][rs.randint(5)]
        Wi = int(rs.choice([128, 192, 256])); Hi = int(rs.choice([128, 192, 256]))
        if os.environ.get('FUZZ_BIG') == '1':             # planes that take the 64x64 tiles of the headline world
            grid = [(1, 2), (2, 1), (2, 2)][rs.randint(3)]
            Wi = int(rs.choice([512, 640, 768])); Hi = int(rs.choice([512, 640, 1024]))
        W, H = Wi * grid[0], Hi * grid[1]
        N = int(rs.choice([4000, 20000, 60000]))
        if os.environ.get('FUZZ_BIG') == '1':
            N = int(W * H * rs.choice([0.02, 0.15]))
        me = int(rs.choice([1, 2, 3]))
        reach = int(np.ceil(max(6.2 / (W - 1), 0.03) * (max(W, H) - 1)))
        loss = reach + 1 + int(np.ceil(max(1.53 / (W - 1), 0.01 * 1.5) * (max(W, H) - 1) + 0.5)) + 2
        while me > 1 and 2 * (me * loss + 3) > min(Wi if grid[0] > 1 else 10 ** 6, Hi if grid[1] > 1 else 10 ** 6):
            me -= 1                      # (the ghost halo must fit twice into a rank's interior)
        return dict(grid=grid, W=W, H=H, N=N, K=N, agent='physarum', boundary=str(rs.choice(['wrap', 'limit'])), agents_die=False,
                    steps=int(rs.choice([7, 10])), migrate_every=me, sort_every=0, seed=seed, ghosts=True)
    if GHOST:        # ghost-agent mode: bounded steps only, tiles at least twice the halo
        grid = [(1, 2), (2, 1), (2, 2), (1, 3), (3, 1), (1, 4)][rs.randint(6)]
        if os.environ.get('FUZZ_GRID'):
            grid = tuple(int(v) for v in os.environ['FUZZ_GRID'].split('x'))
            assert grid[0] * grid[1] <= 5, 'the GPU box allows 6 processes on the card, the parent is one of them'
        me = int(rs.choice([1, 2, 3]))
        Wi = int(rs.choice([112, 128, 160])); Hi = int(rs.choice([112, 128, 144]))
        W, H = Wi * grid[0], Hi * grid[1]
        N = int(rs.choice([1500, 9000])); K = int(N * rs.choice([0.6, 1.0]))
        reach = int(np.ceil(max(6.2 / (W - 1), 0.03) * (max(W, H) - 1)))
        step = max(1.53 / (W - 1), 0.01 * 1.5) * (max(W, H) - 1) + 0.5
        loss = reach + 1 + int(np.ceil(step)) + 2
        while me > 1 and 2 * (me * loss + 3) > min(Wi if grid[0] > 1 else 10 ** 6, Hi if grid[1] > 1 else 10 ** 6):
            me -= 1
        return dict(grid=grid, W=W, H=H, N=N, K=K, agent=str(rs.choice(['physarum', 'gradient'])), boundary=str(rs.choice(['wrap', 'limit'])),
                    agents_die=False, steps=int(rs.choice([7, 10])), migrate_every=me, sort_every=int(rs.choice([0, 2, 3])), seed=seed, ghosts=True)
    grid = [(1, 2), (2, 1), (2, 2), (1, 4), (4, 1), (1, 3), (3, 1)][rs.randint(7)]
    Wi = int(rs.choice([48, 64, 80])); Hi = int(rs.choice([44, 64, 96]))
    W, H = Wi * grid[0], Hi * grid[1]
    N = int(rs.choice([500, 3000])); K = int(N * rs.choice([0.6, 1.0]))
    agent = rs.choice(['physarum', 'brownian', 'gradient'])
    boundary = rs.choice(['wrap', 'limit'])
    die = bool(rs.rand() < 0.3)
    me = 1 if agent == 'brownian' else int(rs.choice([1, 3, 6]))
    return dict(grid=grid, W=W, H=H, N=N, K=K, agent=str(agent), boundary=str(boundary), agents_die=die, steps=8, migrate_every=me,
                sort_every=int(rs.choice([0, 2])), seed=seed)

_make_case = make_case
def make_case(seed):                    # FUZZ_STEPS=n: long runs (many refreshes on the same layouts)
    case = _make_case(seed)
    if os.environ.get('FUZZ_STEPS'):
        case['steps'] = int(os.environ['FUZZ_STEPS'])
    return case

def build(case, die_amd):
    from tests.test_gpu_parity import random_state, f32
    rs = np.random.RandomState(case['seed'] + 1000)
    medium, agents = random_state(case['W'], case['H'], case['N'], case['K'], rs, collide=0.2)
    turn = np.radians(30)
    dir0 = f32(np.floor(rs.uniform(-np.pi, np.pi, case['N']) / turn) * turn)
    prev = f32(rs.normal(0, .4, (2, case['N'])))
    dyn = die_amd.Dynamics(boundary=die_amd.BoundaryCondition(case['boundary']), agents_die=case['agents_die'])
    return medium, agents, dir0, prev, dyn

def make_agent(case, die_amd, n_slots):
    W = case['W']
    if case['agent'] == 'physarum':
        return die_amd.PhysarumAgent(max_agents=n_slots, seed=9, scale=1.53 / (W - 1), sense_offset=6.2 / (W - 1), sense_angle=100)
    if case['agent'] == 'gradient':
        return die_amd.GradientAgent(max_agents=n_slots, seed=9, scale=0.01, sense_offset=0.03, inertia=0.9, noise_scale=0.025)
    return die_amd.BrownianAgent(move_scale=0.3, deposit_scale=0.5, seed=9)      # jumps across several tiles

def worker(rank, size, port, case, out):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    try:
        import die_amd
        from die_amd.dist import DistEnv
        medium, agents, dir0, prev, dyn = build(case, die_amd)
        env = DistEnv.from_global_numpy(medium, agents, case['grid'], dyn, probe_reach=int(np.ceil(max(6.2 / (case['W'] - 1), 0.03) * (max(case['W'], case['H']) - 1))), device='cuda:0', sort_every=case['sort_every'],
                                        capacity=case['N'] * int(os.environ.get('FUZZ_CAP_MULT', '1')) + 64,      # (FUZZ_CAP_MULT=8: room for the refresh in place)
                                        migrate_every=case['migrate_every'], ghosts=case.get('ghosts', False),
                                        max_step_cells=max(1.53 / (case['W'] - 1), 0.01 * 1.5) * (max(case['W'], case['H']) - 1) + 0.5)
        ag = make_agent(case, die_amd, env.capacity)
        if case['agent'] != 'brownian':
            sl = env.local_slots()
            d = torch.zeros(env.capacity, dtype=torch.float32, device='cuda:0'); d[:env.agents.N] = torch.from_numpy(dir0.astype(np.float32)).cuda()[sl]
            p = None
            if case['agent'] == 'gradient':
                p = torch.zeros((2, env.capacity), dtype=torch.float32, device='cuda:0'); p[:, :env.agents.N] = torch.from_numpy(prev.astype(np.float32)).cuda()[:, sl]
            ag.set_state_local(env.agents, d, p)
        obs = env._get_current_obs
        mat = np.random.RandomState(case['seed'] + 77).rand(case['steps']) < float(os.environ.get('FUZZ_MATERIALISE', '0'))
        for i in range(case['steps']):
            act = ag.forward(obs)
            if mat[i] and case['agent'] != 'brownian':      # (FUZZ_MATERIALISE=p: the caller reads the action before the step, with probability p)
                act.to_numpy()
            obs, res = env.step(act)
        world = env.gather_world()
        if hasattr(env, 'check'):
            env.check()
        if rank == 0:
            np.savez(out, medium=world[0], agents=world[1], pic_steps=getattr(env, 'pic_steps', 0), plane=np.array([env.geo.W, env.geo.H]),
                     inplace=getattr(env, 'inplace_refreshes', 0), tile_refreshes=getattr(env, 'tile_refreshes', 0), early=getattr(env, 'early_packs', 0), tile=np.array(env._pic_tile if getattr(env, '_pic_tile', None) else [0, 0]),
                     overlapped=getattr(env, 'overlapped_refreshes', 0))
    finally:
        dist.destroy_process_group()

if __name__ == '__main__':
    import torch.multiprocessing as mp
    import die_amd
    fails = 0
    for seed in range(int(os.environ.get('FUZZ_FIRST', '0')), int(os.environ.get('FUZZ_FIRST', '0')) + int(os.environ.get('FUZZ_CASES', '8'))):
        case = make_case(seed)
        out = os.path.join(tempfile.gettempdir(), f'fd_{seed}.npz')
        size = case['grid'][0] * case['grid'][1]
        try:
            mp.spawn(worker, args=(size, free_port(), case, out), nprocs=size, join=True)
            got = np.load(out)
            medium, agents, dir0, prev, dyn = build(case, die_amd)
            env = die_amd.Env.from_numpy(medium, agents, dyn, sort_every=0)
            ag = make_agent(case, die_amd, case['N'])
            if case['agent'] != 'brownian':
                ag.set_state(dir0, prev if case['agent'] == 'gradient' else None)
            obs = env._get_current_obs
            for _ in range(case['steps']):
                obs, *_ = env.step(ag.forward(obs))
            m, a = env.medium.to_numpy(), env.agents.to_numpy()
            if case['boundary'] == 'limit':
                assert np.abs(got['agents'][:2] - a[:2]).max() == 0
            assert np.array_equal(got['agents'], a), 'agents'
            assert np.array_equal(got['medium'], m), 'medium'
            print('ok  ', case, 'binned steps', int(got['pic_steps']), 'plane', got['plane'].tolist(), 'refreshes by tiles', int(got['tile_refreshes']), 'in place', int(got['inplace']), 'packed early', int(got['early']), 'overlapped', int(got['overlapped']), 'tile', got['tile'].tolist(), flush=True)
        except Exception as e:
            fails += 1; print('FAIL', case, type(e).__name__, str(e)[-400:].replace(chr(10), ' | '), flush=True)
    print(f'fuzz dist: {fails} failures', flush=True)
