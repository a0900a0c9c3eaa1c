"""Decomposed world vs single GPU: several ranks share the one GPU of the test box and talk over
gloo (host-staged); the decomposed run must reproduce the single-device run bit for bit (Philox
and ownership are keyed by world slot ids, the tile kernels do the same arithmetic)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _state(W, H, N, K, seed):
    from tests.test_gpu_parity import f32, random_state
    rs = np.random.RandomState(seed)
    medium, agents = random_state(W, H, N, K, rs, collide=0.1)
    turn = np.radians(30)
    dir0 = f32(np.floor(rs.uniform(-np.pi, np.pi, N) / turn) * turn)
    return medium, agents, dir0


def _wave_dynamics(die_amd, W, H, kind=True):
    """'dyn-pred' of examples/simple_agents.py:95-100: food flows in running waves (`kind='perlin'`: in drifting Perlin noise,
    core/data_init.py:55-69)."""
    if kind == 'limit':                  # (no food flow: agents stop at the world's edge and pile up there — local counts grow)
        return die_amd.Dynamics(boundary=die_amd.BoundaryCondition.limit)
    if kind == 'perlin':
        seq = die_amd.PerlinNoiseSequence((W, H), dt=0.05, octaves=6, seed=5)
    else:
        seq = die_amd.WaveSequence((W, H), dt=0.01)
    return die_amd.Dynamics(op_food_flow=seq.get_flow_operator(scale=0.5, decay=0.5))


def _worker(rank, size, port, grid, W, H, N, K, steps, sort_every, overlap, migrate_every, out_path, backend='gloo', ghosts=False,
            wave=False, f16=False, read_actions=False, materialise_at=(), capacity=None, final=None, expect_error=None):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    if backend == 'gloo-torch-refresh':          # the ghost refresh written with torch index operations (cross-check path)
        os.environ['DIE_GHOST_REFRESH'] = 'torch'
        backend = 'gloo'
    dev = 'cuda:0'
    if backend == 'nccl':        # one rank: its periodic self-neighbour messages go through RCCL send/recv
        os.environ['DIE_DIST_SELF_VIA_BACKEND'] = '1'
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=rank, world_size=size, device_id=torch.device('cuda:0'))
    elif backend == 'nccl-multi':  # one rank per physical GPU, RCCL between distinct peers (needs a multi-GPU node)
        dev = f'cuda:{rank}'
        torch.cuda.set_device(rank)
        dist.init_process_group('nccl', rank=rank, world_size=size, device_id=torch.device(dev))
    elif expect_error is not None:   # (a rank that did NOT fail would wait for its dead peers: bounded)
        import datetime
        dist.init_process_group('gloo', rank=rank, world_size=size, timeout=datetime.timedelta(seconds=240))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=size)
    try:
        import die_amd
        from die_amd.dist import DistEnv
        medium, agents, dir0 = _state(W, H, N, K, 5)
        kw = dict(scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1), sense_angle=100)
        def make(cap_):
            return DistEnv.from_global_numpy(medium, agents, grid, _wave_dynamics(die_amd, W, H, wave) if wave else None, probe_reach=11,
                                             device=dev, sort_every=sort_every,
                                             overlap=overlap, migrate_every=migrate_every, max_step_cells=1.6, ghosts=ghosts,
                                             field_dtype=torch.float16 if f16 else torch.float32, **({'capacity': cap_} if cap_ else {}))
        if capacity == 'tight':              # arrays that hold the agents a rank starts with and a few entries more — on every rank
            probe = make(None)
            probe._refresh_ghosts(None, after_step=True)      # (collective: the ghosts a rank holds from its first refresh on)
            n0 = torch.tensor([probe.agents.N], dtype=torch.int64)
            del probe
            dist.all_reduce(n0, op=dist.ReduceOp.MAX)
            capacity = int(n0.item()) + 16
        env = make(capacity)
        cap = env.capacity
        agent = die_amd.PhysarumAgent(max_agents=cap, seed=9, **kw)
        local = torch.zeros(cap, dtype=torch.float32, device=dev)
        local[:env.agents.N] = torch.from_numpy(dir0.astype(np.float32)).to(dev)[env.local_slots()]
        agent.set_state_local(env.agents, local)
        obs = env._get_current_obs
        rewards = []
        try:
            for i in range(steps):
                act = agent.forward(obs)
                if i in materialise_at:      # the stand-alone forward runs NOW, on what the rank holds now (a ghost refresh
                    act.to_numpy()           # that was left for this step has to happen first — the halos are past their window)
                obs, res = env.step(act)
                rewards.append(env.read_result(res))
                if read_actions:             # the binned step kept it in registers: re-derived from what the step left behind
                    a = act.to_numpy()
                    assert a.shape[0] == 3 and np.isfinite(a).all()
        except RuntimeError as e:
            if expect_error is None or expect_error not in str(e):
                raise
            torch.cuda.synchronize()         # (whatever was queued behind the refresh has run: no fault)
            np.savez(f'{out_path}.rank{rank}.npz', error=str(e), step=i, inplace=getattr(env, 'inplace_refreshes', 0))
            return
        if expect_error is not None:
            raise AssertionError(f'rank {rank}: the run was expected to fail with "{expect_error}"')
        if ghosts and env._all_alive:       # every local entry is alive, also those a refresh appended beyond the old count
            assert bool(env.agents.alive[:env.agents.N].all())
        if final == 'sort':                  # the tile order is voided while a refresh is still waiting for the next step
            assert env._refresh_due, 'the run was meant to end on a deferred refresh'
            env.sort_agents()
        world = env.gather_world()
        if hasattr(env, 'check'):
            env.check()
        if rank == 0:
            np.savez(out_path, medium=world[0], agents=world[1], rewards=np.array(rewards), pic_steps=getattr(env, 'pic_steps', 0),
                     plane=np.array([env.geo.W, env.geo.H]), tile_refreshes=getattr(env, 'tile_refreshes', 0),
                     overlapped=getattr(env, 'overlapped_refreshes', 0), inplace=getattr(env, 'inplace_refreshes', 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('grid,sort_every,overlap,migrate_every,backend', [
    ((1, 2), 0, True, 1, 'gloo'), ((2, 1), 3, False, 1, 'gloo'), ((2, 2), 0, False, 1, 'gloo'), ((2, 2), 2, True, 1, 'gloo'),
    ((2, 2), 2, True, 4, 'gloo'), ((1, 2), 0, False, 3, 'gloo'), ((2, 1), 3, True, 5, 'gloo'),
    # the RCCL transport itself: a 1x1 "decomposition" whose 8 periodic neighbours are the rank itself, every halo /
    # claim-merge message sent and received through RCCL (8 messages to one peer per exchange, matched in issue order)
    ((1, 1), 2, True, 1, 'nccl'), ((1, 1), 0, True, 4, 'nccl')])
def test_decomposed_run_equals_single_device_run(tmp_path, grid, sort_every, overlap, migrate_every, backend):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import torch.multiprocessing as mp
    import die_amd
    W, H, N, K, steps = 128, 96, 2000, 1800, 12
    out = str(tmp_path / 'dist.npz')
    size = grid[0] * grid[1]
    mp.spawn(_worker, args=(size, _free_port(), grid, W, H, N, K, steps, sort_every, overlap, migrate_every, out, backend),
             nprocs=size, join=True)
    got = np.load(out)

    medium, agents, dir0 = _state(W, H, N, K, 5)
    env = die_amd.Env.from_numpy(medium, agents, sort_every=0)
    agent = die_amd.PhysarumAgent(max_agents=N, seed=9, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1), sense_angle=100)
    agent.set_state(dir0)
    obs = env._get_current_obs
    rewards = []
    for _ in range(steps):
        obs, rew, _, _, info = env.step(agent.forward(obs))
        rewards.append((rew, info['num_agents']))
    m, a = env.medium.to_numpy(), env.agents.to_numpy()
    assert np.array_equal(got['agents'], a)
    assert np.array_equal(got['medium'][0], m[0])
    assert np.array_equal(got['medium'][1], m[1])
    assert np.array_equal(got['medium'][2], m[2])
    r = np.array(rewards)
    assert np.array_equal(got['rewards'][:, 1], r[:, 1])
    assert np.array_equal(got['rewards'][:, 0], r[:, 0])          # fixed-point accumulation: exact in any decomposition


def _single_device_run(W, H, N, K, steps, wave=False, f16=False):
    import die_amd
    medium, agents, dir0 = _state(W, H, N, K, 5)
    env = die_amd.Env.from_numpy(medium, agents, _wave_dynamics(die_amd, W, H, wave) if wave else None, sort_every=0,
                                 field_dtype=torch.float16 if f16 else torch.float32)
    agent = die_amd.PhysarumAgent(max_agents=N, seed=9, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1), sense_angle=100)
    agent.set_state(dir0)
    obs = env._get_current_obs
    rewards = []
    for _ in range(steps):
        obs, rew, _, _, info = env.step(agent.forward(obs))
        rewards.append((rew, info['num_agents']))
    return env.medium.to_numpy(), env.agents.to_numpy(), np.array(rewards)


@pytest.mark.parametrize('grid,sort_every,refresh_every,backend,wave', [
    ((1, 2), 0, 2, 'gloo', False), ((2, 1), 3, 3, 'gloo', False), ((2, 2), 2, 2, 'gloo', False), ((2, 2), 0, 3, 'gloo', True),
    ((2, 2), 4, 1, 'gloo', False), ((1, 1), 2, 4, 'nccl', True), ((2, 2), 2, 2, 'gloo-torch-refresh', False),
    ((1, 2), 3, 3, 'gloo-f16', False), ((2, 2), 2, 3, 'gloo', 'perlin')])
def test_ghost_agent_mode_equals_single_device_run(tmp_path, grid, sort_every, refresh_every, backend, wave):
    """Communication-avoiding mode: ghosts of the neighbours' border agents are stepped locally, nothing crosses
    ranks for `refresh_every` steps; world state and rewards must equal the single-device run bit for bit
    (dead slots included: N > K)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import torch.multiprocessing as mp
    W, H, N, K, steps = 256, 192, 7000, 6400, 13
    out = str(tmp_path / 'dist.npz')
    size = grid[0] * grid[1]
    f16 = backend.endswith('-f16')                                   # fp16 field channels (BASELINE configs[4])
    backend = backend.replace('-f16', '')
    mp.spawn(_worker, args=(size, _free_port(), grid, W, H, N, K, steps, sort_every, False, refresh_every, out, backend, True, wave,
                            f16), nprocs=size, join=True)
    got = np.load(out)
    m, a, r = _single_device_run(W, H, N, K, steps, wave, f16)
    assert np.array_equal(got['agents'], a)
    for c in range(3):
        assert np.array_equal(got['medium'][c], m[c])
    assert np.array_equal(got['rewards'][:, 1], r[:, 1])
    assert np.array_equal(got['rewards'][:, 0], r[:, 0])          # fixed-point accumulation: exact in any decomposition


@pytest.mark.parametrize('overlap', [False, True, 'in place'])
@pytest.mark.parametrize('grid,refresh_every,backend,wave,plane', [
    ((2, 2), 2, 'gloo', False, (320, 256)), ((1, 2), 3, 'gloo', True, (384, 256)), ((2, 1), 2, 'gloo-f16', False, (320, 256)),
    ((1, 1), 4, 'nccl', False, (384, 256)), ((1, 2), 2, 'gloo', 'limit', (384, 256))])
def test_ghost_agent_mode_with_the_tile_binned_step(tmp_path, grid, refresh_every, backend, wave, plane, overlap):
    """A rank of the ghost-agent decomposition takes the step the single GPU takes: the tile-binned two-launch step on its
    padded tile (die_pic.hip TILED: agents binned by the plane cell that holds their world cell, ownership-masked reward,
    probes clamped at the WORLD's edge; the halo is rounded up until the planes split into whole tiles); the refresh goes
    by tiles and leaves the tile order intact.  Every slot alive (the binned step's precondition).  Gathered world and rewards equal the single-device
    run bit for bit; the worker reports how many of its steps took the binned path: all of them.
    `overlap`: a refresh is left for the step that follows it and its messages (one per peer) travel under that step's interior
    tiles — agent kernel and field kernel on the tiles that need nothing from a neighbour, then the halo tiles' segments, then
    both kernels on the rest (DistEnv._refresh_ghosts_tiles(step=...)): the same bits.  'in place' (round 5; what a rank with the
    benchmark's proportions takes — the small worlds here need their arrays enlarged for it): that step reads the layout the step before
    left, the interior tiles' segments are not copied, only the halo tiles get new segments (die_pic_ghost_inplace)."""
    in_place = overlap == 'in place'
    overlap = bool(overlap)
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import torch.multiprocessing as mp
    W, H, N, steps = 384, 256, 12000, 9
    out = str(tmp_path / 'dist.npz')
    size = grid[0] * grid[1]
    f16 = backend.endswith('-f16')
    backend = backend.replace('-f16', '')
    mp.spawn(_worker, args=(size, _free_port(), grid, W, H, N, N, steps, 0, overlap, refresh_every, out, backend, True, wave, f16, False, (),
                            8 * N if in_place else None),
             nprocs=size, join=True)
    got = np.load(out)
    assert int(got['inplace']) == ((steps - 1) // refresh_every if size > 1 and in_place else 0)
    assert int(got['pic_steps']) == steps, 'the ranks did not take the tile-binned step'
    assert tuple(got['plane']) == plane                     # (halo rounded up so that the planes are whole tiles)
    # every refresh after a step went by tiles (csrc/die_pic_refresh.hip: no per-agent classification, no re-bin afterwards)
    assert int(got['tile_refreshes']) == (steps // refresh_every if size > 1 else 0)
    # … and, with overlap, inside the step that followed (the last one, if the run ends on a refresh, is flushed by the reader)
    assert int(got['overlapped']) == ((steps - 1) // refresh_every if size > 1 and overlap else 0)
    m, a, r = _single_device_run(W, H, N, N, steps, wave, f16)
    assert np.array_equal(got['agents'], a)
    for c in range(3):
        assert np.array_equal(got['medium'][c], m[c])
    assert np.array_equal(got['rewards'][:, 1], r[:, 1])
    assert np.array_equal(got['rewards'][:, 0], r[:, 0])


@pytest.mark.parametrize('grid,refresh_every', [((1, 2), 1), ((2, 2), 1), ((2, 1), 2)])
def test_action_read_after_a_step_that_ran_inside_a_ghost_refresh(tmp_path, grid, refresh_every):
    """A step that runs inside a deferred ghost refresh works on the NEW layout (another number of local agents than the one its
    action was built for): reading that action afterwards re-derives it for the new count (PicState._rebuilder).  Found by
    scratch/fuzz_dist.py with a refresh at every step ("the action holds n entries, the layout m"); the other dist tests never
    read an action."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import torch.multiprocessing as mp
    W, H, N, steps = 384, 256, 12000, 7
    out = str(tmp_path / 'dist.npz')
    size = grid[0] * grid[1]
    mp.spawn(_worker, args=(size, _free_port(), grid, W, H, N, N, steps, 0, True, refresh_every, out, 'gloo', True, False, False, True),
             nprocs=size, join=True)
    got = np.load(out)
    assert int(got['pic_steps']) == steps and int(got['overlapped']) == (steps - 1) // refresh_every
    m, a, r = _single_device_run(W, H, N, N, steps, False, False)
    assert np.array_equal(got['agents'], a)
    for c in range(3):
        assert np.array_equal(got['medium'][c], m[c])
    assert np.array_equal(got['rewards'], r)


@pytest.mark.parametrize('grid,refresh_every,at', [((1, 2), 2, (2, 5)), ((2, 2), 1, (1, 2, 4)), ((2, 1), 3, (3,))])
def test_action_materialised_before_the_step_that_follows_a_deferred_refresh(tmp_path, grid, refresh_every, at):
    """With the refresh left for the next step (overlap), a forward that does NOT run fused into that step — here: the action is read
    between forward() and step() — must not sense the stale halos: DeviceMedium.before_sense makes the deferred refresh happen first
    (collectively: every rank runs the same program).  ADVICE r4: before the fix such a run silently left the single-device run."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import torch.multiprocessing as mp
    W, H, N, steps = 384, 256, 12000, 7
    out = str(tmp_path / 'dist.npz')
    size = grid[0] * grid[1]
    mp.spawn(_worker, args=(size, _free_port(), grid, W, H, N, N, steps, 0, True, refresh_every, out, 'gloo', True, False, False, False, at),
             nprocs=size, join=True)
    got = np.load(out)
    m, a, r = _single_device_run(W, H, N, N, steps, False, False)
    assert np.array_equal(got['agents'], a)
    for c in range(3):
        assert np.array_equal(got['medium'][c], m[c])
    assert np.array_equal(got['rewards'], r)


@pytest.mark.parametrize('grid,refresh_every', [((1, 2), 3), ((2, 2), 2)])
def test_deferred_refresh_then_sort_then_gather(tmp_path, grid, refresh_every):
    """A run that ENDS on a refresh left for the next step, then sort_agents() (which voids the tile order), then gather_world(): the
    flushed refresh goes agent by agent (native path) with no action to re-count.  ADVICE r5: `action.N = n_new` on None."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import torch.multiprocessing as mp
    W, H, N = 384, 256, 12000
    steps = 2 * refresh_every                # (the last step is followed by a refresh, which overlap defers)
    out = str(tmp_path / 'dist.npz')
    size = grid[0] * grid[1]
    mp.spawn(_worker, args=(size, _free_port(), grid, W, H, N, N, steps, 0, True, refresh_every, out, 'gloo', True, False, False, False, (),
                            None, 'sort'), nprocs=size, join=True)
    got = np.load(out)
    m, a, r = _single_device_run(W, H, N, N, steps, False, False)
    assert np.array_equal(got['agents'], a)
    for c in range(3):
        assert np.array_equal(got['medium'][c], m[c])
    assert np.array_equal(got['rewards'], r)


@pytest.mark.parametrize('grid,refresh_every', [((2, 2), 2), ((1, 2), 2)])
def test_refresh_in_place_out_of_room_fails_cleanly_on_every_rank(tmp_path, monkeypatch, grid, refresh_every):
    """The ghost refresh in place (die_pic_ghost_inplace) with arrays a few entries longer than the agents they hold — the host's
    own test of the room switched off (DIE_REFRESH_IN_PLACE_FORCE=1) so that the KERNELS' guards are what is exercised: the halo
    tiles' new segments do not fit behind the arrays' old end, the scan raises RF_FLAG_CAPACITY, no halo tile is laid past the end
    (such a tile is published empty), the step queued behind the refresh runs without touching memory beyond the arrays, and
    every rank raises the same clean RuntimeError.  (Round 5's development build faulted here on one rank while another raised:
    gpurun_out/r5_t8.log)"""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import torch.multiprocessing as mp
    monkeypatch.setenv('DIE_REFRESH_IN_PLACE_FORCE', '1')
    W, H, N, steps = 384, 256, 12000, 6
    out = str(tmp_path / 'dist.npz')
    size = grid[0] * grid[1]
    mp.spawn(_worker, args=(size, _free_port(), grid, W, H, N, N, steps, 0, True, refresh_every, out, 'gloo', True, False, False, False, (),
                            'tight', None, 'do not fit the local arrays'), nprocs=size, join=True)
    for rank in range(size):
        got = np.load(f'{out}.rank{rank}.npz')
        assert 'do not fit the local arrays' in str(got['error'])
        assert int(got['step']) == refresh_every           # the step that carried the first deferred refresh
        assert int(got['inplace']) == 1                    # … which took the in-place path


@pytest.mark.parametrize('ghosts,migrate_every', [(True, 3), (False, 1), (False, 4)])
def test_two_physical_gpus_over_rccl(tmp_path, ghosts, migrate_every):
    """The RCCL transport between DISTINCT ranks (persistent P2P op lists, several messages per peer matched by issue
    order, the chem halo on its own stream): one rank per physical GPU.  The pool's test boxes have one GPU, so this is
    skipped there — until it has run on a multi-GPU node the multi-GPU path counts as unverified."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    import torch.multiprocessing as mp
    W, H, N, K, steps = 256, 192, 7000, 6400, 13
    out = str(tmp_path / 'dist.npz')
    mp.spawn(_worker, args=(2, _free_port(), (1, 2), W, H, N, K, steps, 2, True, migrate_every, out, 'nccl-multi', ghosts),
             nprocs=2, join=True)
    got = np.load(out)
    m, a, r = _single_device_run(W, H, N, K, steps)
    assert np.array_equal(got['agents'], a)
    for c in range(3):
        assert np.array_equal(got['medium'][c], m[c])
    assert np.array_equal(got['rewards'], r)


# ---------------------------------------------------------------------------------------------------------
# regression cases found by scratch/fuzz_dist.py: starved slots (agents_die) jump to world cell (0, 0) and must
# be re-homed before the next forward(); Brownian agents jump across several tiles (general routing); 'limit'.
def _case_state(case):
    import die_amd
    from tests.test_gpu_parity import f32, random_state
    rs = np.random.RandomState(case['seed'])
    medium, agents = random_state(case['W'], case['H'], case['N'], case['K'], rs, collide=0.2)
    turn = np.radians(30)
    dir0 = f32(np.floor(rs.uniform(-np.pi, np.pi, case['N']) / turn) * turn)
    dyn = die_amd.Dynamics(boundary=die_amd.BoundaryCondition(case['boundary']), agents_die=case['agents_die'])
    return medium, agents, dir0, dyn


def _case_agent(case, n_slots):
    import die_amd
    if case['agent'] == 'physarum':
        return die_amd.PhysarumAgent(max_agents=n_slots, seed=9, scale=1.53 / (case['W'] - 1), sense_offset=6.2 / (case['W'] - 1),
                                     sense_angle=100)
    return die_amd.BrownianAgent(move_scale=0.3, deposit_scale=0.5, seed=9)


def _case_worker(rank, size, port, case, out_path):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    try:
        from die_amd.dist import DistEnv
        medium, agents, dir0, dyn = _case_state(case)
        reach = int(np.ceil(6.2 / (case['W'] - 1) * (max(case['W'], case['H']) - 1)))
        env = DistEnv.from_global_numpy(medium, agents, case['grid'], dyn, probe_reach=reach, device='cuda:0',
                                        sort_every=case['sort_every'], capacity=case['N'] + 64)
        ag = _case_agent(case, env.capacity)
        if case['agent'] == 'physarum':
            local = torch.zeros(env.capacity, dtype=torch.float32, device='cuda:0')
            local[:env.agents.N] = torch.from_numpy(dir0.astype(np.float32)).cuda()[env.local_slots()]
            ag.set_state_local(env.agents, local)
        obs = env._get_current_obs
        for _ in range(case['steps']):
            obs, res = env.step(ag.forward(obs))
        world = env.gather_world()
        if rank == 0:
            np.savez(out_path, medium=world[0], agents=world[1])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('case', [
    dict(grid=(1, 2), W=48, H=88, N=500, K=300, agent='physarum', boundary='wrap', agents_die=True, steps=8, sort_every=0, seed=100),
    dict(grid=(1, 3), W=32, H=96, N=3000, K=3000, agent='brownian', boundary='limit', agents_die=True, steps=6, sort_every=2, seed=1),
], ids=['starved-slots-rehomed', 'brownian-jumps-limit'])
def test_decomposed_regressions(tmp_path, case):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import torch.multiprocessing as mp
    import die_amd
    out = str(tmp_path / 'case.npz')
    size = case['grid'][0] * case['grid'][1]
    mp.spawn(_case_worker, args=(size, _free_port(), case, out), nprocs=size, join=True)
    got = np.load(out)
    medium, agents, dir0, dyn = _case_state(case)
    env = die_amd.Env.from_numpy(medium, agents, dyn, sort_every=0)
    ag = _case_agent(case, case['N'])
    if case['agent'] == 'physarum':
        ag.set_state(dir0)
    obs = env._get_current_obs
    for _ in range(case['steps']):
        obs, *_ = env.step(ag.forward(obs))
    assert np.array_equal(got['agents'], env.agents.to_numpy())
    assert np.array_equal(got['medium'], env.medium.to_numpy())


# ---------------------------------------------------------------------------------------------------------
# BASELINE configs[3] at FULL size: an 8192² world over 2×2 ranks (sharing the test box's one GPU, gloo), ghost-agent
# mode with the bench's refresh period.  State travels through files (the in-memory gather of the small cases would
# pickle gigabytes): the single-device run writes the initial world, every rank writes its interior tile and owned agents.
def _full_worker(rank, size, port, grid, W, H, steps, sort_every, refresh_every, tmp):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    try:
        import die_amd
        from die_amd.dist import DistEnv
        medium = np.load(os.path.join(tmp, 'medium.npy'), mmap_mode='r')
        agents = np.load(os.path.join(tmp, 'agents.npy'), mmap_mode='r')
        dir0 = np.load(os.path.join(tmp, 'dir0.npy'))
        env = DistEnv.from_global_numpy(medium, agents, grid, None, probe_reach=11, device='cuda:0', sort_every=sort_every,
                                        overlap=True, migrate_every=refresh_every, max_step_cells=1.6, ghosts=True,
                                        ghost_headroom=1.3)
        del medium, agents
        agent = die_amd.PhysarumAgent(max_agents=env.capacity, seed=3, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
        local = torch.zeros(env.capacity, dtype=torch.float32, device='cuda:0')
        local[:env.agents.N] = torch.from_numpy(dir0).cuda()[env.local_slots()]
        agent.set_state_local(env.agents, local)
        obs = env._get_current_obs
        rewards = []
        for _ in range(steps):
            obs, res = env.step(agent.forward(obs))
            rewards.append(env.read_result(res))
        assert env.overlapped_refreshes == (steps - 1) // refresh_every       # (the refresh travelled under the step that followed it)
        # … in place: at the benchmark's proportions the halo tiles' new segments fit behind the arrays' old end by default
        assert getattr(env, 'inplace_refreshes', 0) == (steps - 1) // refresh_every
        own = env.owned_mask()
        g, A, n = env.geo, env.agents, env.agents.N
        ri, ci = g.interior()
        np.savez(os.path.join(tmp, f'rank{rank}.npz'), x0=g.x0, y0=g.y0,
                 occ=env.medium.occupied()[ri, ci].cpu().numpy(), food=env.medium.food[ri, ci].cpu().numpy(),
                 chem=env.medium.chem[ri, ci].cpu().numpy(), slots=A.slot[:n][own].cpu().numpy(),
                 x=A.x[:n][own].cpu().numpy(), y=A.y[:n][own].cpu().numpy(), alive=A.alive[:n][own].cpu().numpy(),
                 agent_food=A.agent_food[:n][own].cpu().numpy(), rewards=np.array(rewards))
    finally:
        dist.destroy_process_group()


def test_configs3_8192_world_2x2_ranks_equals_single_device(tmp_path):
    """8192×8192 Physarum world, 2×2 domain decomposition (four ranks on the one GPU, gloo), ghost agents re-seated every
    8 steps, 10 steps: the gathered world equals the single-device 8192² run bit for bit, and that run satisfies the
    size-independent invariants of the 4096² test."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import torch.multiprocessing as mp
    import die_amd
    from die_amd.device_array import unpermute
    from tests.test_gpu_parity import check_step_invariants
    W = H = 8192
    steps, tmp = 10, str(tmp_path)
    env = die_amd.Env((W, H), die_amd.Dynamics(init_agent_ratio=0.15), seed=77, max_agents='alive', sort_every=8)
    K = env.agents.N
    assert abs(K / (W * H) - 0.15) < 0.002
    agent = die_amd.PhysarumAgent(max_agents=K, seed=3, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
    agent._alloc_state('cuda:0')
    np.save(os.path.join(tmp, 'medium.npy'), env.medium.to_numpy().astype(np.float32))
    np.save(os.path.join(tmp, 'agents.npy'), env.agents.to_numpy())
    np.save(os.path.join(tmp, "dir0.npy"), agent._direction_rads.cpu().numpy().astype(np.float32))
    rewards = check_step_invariants(env, agent, steps, W, H, 1.53 / (W - 1))
    want = dict(occ=env.medium.occupied().cpu().numpy(), food=env.medium.food.cpu().numpy(), chem=env.medium.chem.cpu().numpy(),
                x=unpermute(env.agents.x, env.agents.slot).cpu().numpy(), y=unpermute(env.agents.y, env.agents.slot).cpu().numpy(),
                agent_food=unpermute(env.agents.agent_food, env.agents.slot).cpu().numpy())
    del env, agent
    torch.cuda.empty_cache()

    grid = (2, 2)
    mp.spawn(_full_worker, args=(4, _free_port(), grid, W, H, steps, 8, 8, tmp), nprocs=4, join=True)
    seen = np.zeros(K, dtype=bool)
    for r in range(4):
        got = np.load(os.path.join(tmp, f'rank{r}.npz'))
        x0, y0 = int(got['x0']), int(got['y0'])
        Wi, Hi = got['chem'].shape
        for name in ('occ', 'food', 'chem'):
            assert np.array_equal(got[name], want[name][x0:x0 + Wi, y0:y0 + Hi]), f'rank {r}: {name}'
        sl = got['slots'].astype(np.int64)
        assert not seen[sl].any()
        seen[sl] = True
        assert got['alive'].all()
        for name in ('x', 'y', 'agent_food'):
            assert np.array_equal(got[name], want[name][sl]), f'rank {r}: agents {name}'
        assert np.array_equal(got['rewards'], np.array(rewards))     # every rank holds the all-reduced world result
    assert seen.all()                                                # every world agent has exactly one owner
