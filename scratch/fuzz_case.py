"""Replay one case of tests.fuzz_cases.fuzz_binned (same RNG stream) with the bookkeeping checked after every step and no
accessor that would index with possibly broken slot ids."""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch, die_amd
from tests.test_gpu_parity import f32, random_state
target, seed = int(sys.argv[1]), int(sys.argv[2])
rs = np.random.RandomState(seed)
for case in range(target + 1):
    xs, ys = [(4, 5), (5, 6), (6, 6), (5, 7)][rs.randint(4)]
    TX, TY = 1 << xs, 1 << ys
    W, H = TX * int(rs.randint(3, 6)), TY * int(rs.randint(3, 5))
    N = int(rs.choice([50, 2000, 20000, W * H // 2]))
    medium, agents = random_state(W, H, N, N, rs, collide=float(rs.choice([0.0, 0.3, 0.9])))
    f16 = bool(rs.rand() < 0.3)
    dyn = dict(boundary=die_amd.BoundaryCondition(rs.choice(['wrap', 'limit'])), food_infinite=bool(rs.rand() < 0.2),
               diffuse_sigma=float(rs.choice([0.4, 0.5, 0.8, 1.0])), rate_feed=float(rs.choice([0.1, 0.35])),
               rate_decay_chem=float(rs.choice([0.01, 0.2])))
    reach = float(rs.choice([0.7, 1.53, min(TX, TY) - 1.001]))
    probe = float(rs.choice([1.2, 10.2, 21.5]))
    kw = dict(scale=reach / (max(W, H) - 1), sense_offset=probe / (max(W, H) - 1), sense_angle=float(rs.choice([60, 90, 120])),
              deposit=float(rs.choice([1.0, 4.0])))
    turn = np.radians(30); dir0 = f32(np.floor(rs.uniform(-np.pi, np.pi, N) / turn) * turn)
    read_mode = rs.choice(['every', 'some', 'never'])
    switch_at = int(rs.randint(2, 7)) if rs.rand() < 0.4 else None
    se = int(rs.choice([0, 2, 3]))
print('case', target, dict(W=W, H=H, N=N, tile=(xs, ys), f16=f16, reach=reach, probe=probe, switch=switch_at, sort_every=se, read=read_mode), dyn)
env = die_amd.Env.from_numpy(medium, agents, die_amd.Dynamics(**dyn), sort_every=se, pic=True, sync=False,
                             field_dtype=torch.float16 if f16 else torch.float32)
env._pic_tile = (xs, ys)
ag = die_amd.PhysarumAgent(max_agents=N, seed=7, **kw)
ag.set_state(dir0)
obs = env._get_current_obs
for i in range(8):
    if switch_at is not None:
        env._pic_enabled = not (switch_at <= i < switch_at + 2)
    a = ag.forward(obs)
    obs, res, *_ = env.step(a)
    torch.cuda.synchronize()
    P = env._pic
    if P is None:
        print('step', i, 'no pic state'); continue
    err = int(P.error[0].item())
    binned = P.held is not None and P.held[0] is env.agents.x
    msg = f'step {i}: binned={binned} error={err}'
    if binned:
        meta = P.meta[P.cur].cpu().numpy().astype(np.int64)              # off, n, s, inc of the layout that holds the agents
        x, y = P.held[0].cpu().numpy().astype(np.int64) & 0xFFFFFFFF, P.held[1].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        cx = np.clip((x * (W - 1) + (1 << 31)) >> 32, 0, W - 1); cy = np.clip((y * (H - 1) + (1 << 31)) >> 32, 0, H - 1)
        tile = (cx >> xs) * (H >> ys) + (cy >> ys)
        NT = meta.shape[1]
        bad = 0
        for t in range(NT):
            o, n, s = meta[0, t], meta[1, t], meta[2, t]
            bad += int((tile[o:o + s] != t).sum())
        msg += f' sum n={meta[1].sum()} (N={N}) stayers on wrong tile={bad} max n={meta[1].max()} leavers={int((meta[1]-meta[2]).sum())}'
    print(msg, flush=True)
    if err:
        break
