"""Host time per step on the classic path (small worlds), Env(sync=True) loop: wrappers with perf_counter_ns.
usage: python3 scratch/host_sections_classic.py [size]"""
import os, sys, time, functools, collections
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch, die_amd
from die_amd import env as E, device_array as D, _lib
acc = collections.Counter(); cnt = collections.Counter()
def wrap(obj, name, label=None):
    f = getattr(obj, name); label = label or f'{getattr(obj, "__name__", obj)}.{name}'
    @functools.wraps(f)
    def g(*a, **k):
        t = time.perf_counter_ns(); r = f(*a, **k); acc[label] += time.perf_counter_ns() - t; cnt[label] += 1; return r
    setattr(obj, name, g)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=True)
ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
obs = env._get_current_obs
for _ in range(50):
    obs, *_ = env.step(ag.forward(obs))
wrap(E.Env, 'step'); wrap(E.Env, '_pic_step'); wrap(E.Env, '_read_host_result'); wrap(E.Env, '_agents_changed'); wrap(E.Env, 'sort_agents')
wrap(type(ag), 'forward', 'agent.forward'); wrap(D.DeviceMedium, 'c_struct', 'Medium.c_struct'); wrap(D.DeviceAgents, 'c_struct', 'Agents.c_struct')
wrap(D.DeviceMedium, 'next_epoch', 'Medium.next_epoch'); wrap(E.Env, '_c_dynamics'); wrap(D.PendingAction, 'raw_struct', 'PendingAction.raw_struct')
for name in ('die_forward_env_step', 'die_env_step'):
    orig = getattr(_lib.lib, name)
    def call(*a, _o=orig, _n=name):
        t = time.perf_counter_ns(); r = _o(*a); acc['C call ' + _n] += time.perf_counter_ns() - t; cnt['C call ' + _n] += 1; return r
    setattr(_lib.lib, name, call)
n = 400
t0 = time.perf_counter_ns()
for _ in range(n):
    obs, *_ = env.step(ag.forward(obs))
tot = (time.perf_counter_ns() - t0) / n / 1e3
print(f'{W}x{W}: {tot:.1f} us per synchronous step')
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f'  {k:40s} {v / n / 1e3:7.2f} us/step  ({cnt[k] / n:.2f} calls)')
