"""K1 (k_pic_forward_move) alone, repeated on the same valid input layout: safe for ablation builds whose outputs are
garbage (nothing consumes them).  usage: DIE_AMD_LIB=... python scratch/k1_micro.py [reps]"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch, die_amd
W = 4096
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False)
ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
obs = env._get_current_obs
import die_amd._lib as L
real = os.environ.get('DIE_AMD_LIB', '').endswith('libdie_hip.so') or not os.environ.get('DIE_AMD_LIB')
# the state is prepared by the shipped library semantics only when this build is sane; ablation builds run stage 1 only
act = ag.forward(obs)
from die_amd.pic import PicState, pick_tile
env._pic_applies(act)
env._pic = PicState(env, env._pic_tile)
env._pic.bin(env, ag)
act.rebind(env.agents)
res = torch.empty(2, dtype=torch.float64, device='cuda')
dyn = env._c_dynamics()
for stages, name in ((1, 'K1'),):
    for _ in range(3):
        env._pic.run_stage(env, act, dyn, res, stages)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        env._pic.run_stage(env, act, dyn, res, stages)
    b.record()
    torch.cuda.synchronize()
    print(f'{os.path.basename(os.environ.get("DIE_AMD_LIB", "libdie_hip.so"))}: {name} {a.elapsed_time(b) / reps * 1e3:.1f} us (freshly binned agents, every agent a stayer)')
