"""Where the host's time per step goes on the tile-binned path (wrappers with perf_counter_ns around the methods; the GPU
runs far behind, nothing waits).  usage: python3 scratch/host_sections.py"""
import os, sys, time, functools, collections
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch, die_amd
from die_amd import pic as P, env as E, device_array as D, _lib
acc = collections.Counter(); cnt = collections.Counter()
def wrap(obj, name, label=None):
    f = getattr(obj, name); label = label or f'{getattr(obj, "__name__", obj)}.{name}'
    @functools.wraps(f)
    def g(*a, **k):
        t = time.perf_counter_ns(); r = f(*a, **k); acc[label] += time.perf_counter_ns() - t; cnt[label] += 1; return r
    setattr(obj, name, g)
W = 2048
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False)
env._pic_tile = (6, 6)
ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
obs = env._get_current_obs
for _ in range(50):
    obs, *_ = env.step(ag.forward(obs))
torch.cuda.synchronize()
wrap(E.Env, 'step'); wrap(E.Env, '_pic_step'); wrap(E.Env, '_pic_applies'); wrap(type(ag), 'forward', 'agent.forward')
wrap(P.PicState, 'step', 'PicState.step'); wrap(P.PicState, '_struct'); wrap(P.PicState, '_out_tensors'); wrap(P.PicState, '_adopt')
wrap(P.PicState, 'flush_lazy'); wrap(P.PicState, '_rebuilder'); wrap(D.DeviceMedium, 'c_struct', 'Medium.c_struct'); wrap(E.Env, '_c_dynamics')
orig = _lib.lib.die_pic_forward_env_step
def call(*a):
    t = time.perf_counter_ns(); r = orig(*a); acc['C call die_pic_forward_env_step'] += time.perf_counter_ns() - t; cnt['C call die_pic_forward_env_step'] += 1; return r
_lib.lib.die_pic_forward_env_step = call
n = 300                                                      # (short: the launch queue must not fill, or the C call blocks on the GPU)
t0 = time.perf_counter_ns()
for _ in range(n):
    obs, *_ = env.step(ag.forward(obs))
tot = (time.perf_counter_ns() - t0) / n / 1e3
torch.cuda.synchronize()
print(f'host per step {tot:.1f} us')
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f'  {k:40s} {v / n / 1e3:7.2f} us/step  ({cnt[k] / n:.1f} calls)')
