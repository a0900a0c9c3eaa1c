"""Workgroup lifetimes of the two kernels of the binned step against the tiles' populations at the bench's window (world step ~205):
how much of a launch is the tail of the crowded tiles?  Needs the -DPIC_STAMPS build (scratch/build_stamps.sh; DIE_AMD_LIB)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
import die_amd
W = 4096; STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 205
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False)
ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
obs = env._get_current_obs
for _ in range(STEPS):
    obs, *_ = env.step(ag.forward(obs))
torch.cuda.synchronize()
pic = env._pic
raw = pic.error[2:2 + 32 * pic.NT].cpu().numpy().view(np.uint64).reshape(-1, 16).astype(np.float64)      # (16 stamps per tile; round 6's build keeps 4 more words per tile behind them: scratch/r6_cu_timeline.py)
n_in = pic.meta[1 - pic.cur][1].cpu().numpy().astype(np.int64)       # the layout the last step READ: agents per tile the agent kernel processed
n_out = pic.meta[pic.cur][1].cpu().numpy().astype(np.int64)
rim = pic.rim_cnt.cpu().numpy().astype(np.int64)
cap = int(die_amd._lib.lib.die_pic_rim_cap(pic.xs, pic.ys))
# (stamps = s_memrealtime: ONE 100 MHz clock for the whole GPU — 10 ns ticks; the stamped build of this script replaces s_memtime by it)
TICK_US = 0.01
for name, a, b, pop, slots in (('agent kernel', 0, 5, n_in, 768), ('field kernel', 8, 14, n_out, 1024)):
    t0, t1 = raw[:, a], raw[:, b]
    ok = (t0 > 0) & (t1 > t0)
    life = (t1 - t0) * TICK_US
    s0, e1 = t0[ok].min(), t1[ok].max()
    dur = (e1 - s0) * TICK_US
    heavy = pop > 2 * pop.mean()
    done = np.sort((t1[ok] - s0) * TICK_US)
    print(f'{name}: launch {dur:6.1f} us (first workgroup start -> last end); workgroup life mean {life[ok].mean():5.1f} median {np.median(life[ok]):5.1f} p99 {np.percentile(life[ok], 99):5.1f} max {life[ok].max():5.1f} us; '
          f'sum of lives / launch = {life[ok].sum() / dur:5.0f} workgroups in flight on average of {slots} slots')
    print(f'   population per tile mean {pop.mean():.0f} p99 {np.percentile(pop, 99):.0f} max {pop.max()}; corr(life, population) {np.corrcoef(life[ok], pop[ok])[0, 1]:.2f}; '
          f'{heavy.sum()} tiles with > 2x the mean: life {life[heavy & ok].mean():.1f} us (others {life[~heavy & ok].mean():.1f})')
    print(f'   50 / 90 / 99 / 99.9 % of the tiles done at {done[len(done)//2]:.1f} / {done[int(.9*len(done))]:.1f} / {done[int(.99*len(done))]:.1f} / {done[int(.999*len(done))]:.1f} us of {dur:.1f}')
    # in-flight workgroups over time (1 us bins)
    edges = np.arange(0, dur + 1, 1.0)
    st, en = (t0[ok] - s0) * TICK_US, (t1[ok] - s0) * TICK_US
    inflight = [(int(((st <= x) & (en > x)).sum())) for x in edges]
    print('   workgroups in flight at every 4th us:', inflight[::4])
    last = np.argsort(t1)[-6:]
    print('   the 6 tiles that end last: population', pop[last].tolist(), 'life', np.round(life[last], 1).tolist(), 'start at', np.round((t0[last] - s0) * TICK_US, 1).tolist())
print(f'rim lists: {int((rim > cap).sum())} of {len(rim)} tiles overflow their {cap} entries (the field kernel scans those tiles\' segments)')
