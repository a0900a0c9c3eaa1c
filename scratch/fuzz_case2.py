"""Replay one fuzz_binned case with the binned and the classic step side by side; report the first differing cells."""
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
print('case', target, dict(W=W, H=H, N=N, tile=(xs, ys), f16=f16, reach=reach, probe=probe), dyn, kw)
envs = []
for pic in (True, False):
    env = die_amd.Env.from_numpy(medium, agents, die_amd.Dynamics(**dyn), sort_every=0, pic=pic, field_dtype=torch.float16 if f16 else torch.float32)
    env._pic_tile = (xs, ys) if pic else None
    ag = die_amd.PhysarumAgent(max_agents=N, seed=7, **kw); ag.set_state(dir0)
    envs.append([env, ag, env._get_current_obs])
for i in range(8):
    pre = [e[0].medium.to_numpy() for e in envs]
    acts = []
    for e in envs:
        a = e[1].forward(e[2]); e[2] = e[0].step(a)[0]; acts.append(a.to_numpy())
    m = [e[0].medium.to_numpy() for e in envs]; ag_ = [e[0].agents.to_numpy() for e in envs]
    same = [bool(np.array_equal(m[0][c], m[1][c])) for c in range(3)]
    print('step', i, 'medium same', same, 'agents same', bool(np.array_equal(ag_[0], ag_[1])), 'actions same', bool(np.array_equal(acts[0], acts[1])))
    if not all(same):
        for c, name in ((1, 'food'), (2, 'chem')):
            bad = np.argwhere(m[0][c] != m[1][c])
            print(' ', name, 'differs at', len(bad), 'cells; first:', bad[:4].tolist())
            for (x, y) in bad[:4]:
                print('    cell', (x, y), 'before', pre[0][c][x, y], pre[1][c][x, y], 'binned', m[0][c][x, y], 'classic', m[1][c][x, y], 'occupied', m[0][0][x, y],
                      'neighbours occupied', m[0][0][max(x-1,0):x+2, max(y-1,0):y+2].sum())
        break
