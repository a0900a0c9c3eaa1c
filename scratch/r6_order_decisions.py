"""Which rebuilds of the order table sort (bands out of band order) and what 32 steps then take — the bench world, from a given world step."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
import die_amd
W = 4096
starts = [int(a) for a in sys.argv[1:]] or [192, 400, 700, 1000]
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False)
ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
done = 0
for upto in starts:
    upto = (upto + 31) // 32 * 32
    env.run(ag, upto - done); done = upto
    torch.cuda.synchronize()
    line = []
    for rep in range(4):
        t0 = time.perf_counter()
        env.run(ag, 32); done += 32
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 32 * 1e6
        pic = env._pic
        n = pic.meta[pic.cur][1].cpu().numpy().astype(np.int64)
        crowded = int((-(-(-(-n // 64)) // 8) >= 4).sum())
        moved = '-'
        if pic.order is not None:
            o = (pic.order.cpu().numpy().astype(np.int64) & 0xFFFF)[:pic.NT]
            ntx, nty = W >> pic.xs, W >> pic.ys
            wb, per = nty // 8, ntx * (nty // 8)
            moved = sum(o[j * per:(j + 1) * per].tolist() != [(q // wb) * nty + j * wb + q % wb for q in range(per)] for j in range(8))
        line.append(f'{dt:6.1f} us/step (bands sorted {moved}, crowded tiles now {crowded})')
    print(f'from step {upto}: ' + '; '.join(line), flush=True)
