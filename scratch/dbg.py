import sys; sys.path.insert(0,'.')
import numpy as np, torch
import die_amd
from die_amd.device_array import from_q32, to_q32
from oracle import cpu_ref as R, rng as orng
W,H=64,48
medium, agents = R.synthetic_init(W,H,0.15,seed=7)
medium[1]=medium[1].astype(np.float32); medium[2]=R.diffuse_decay(np.random.RandomState(0).rand(W,H),1.0,0.0).astype(np.float32)
agents[:2]=from_q32(to_q32(agents[:2])); agents[3]=agents[3].astype(np.float32)
N=agents.shape[1]
kw=dict(scale=1.53/(W-1), sense_offset=10.2/(W-1))
ref=R.RefPhysarumAgent(N, seed=3, **kw); dir0=ref._direction_rads.astype(np.float32).astype(np.float64); ref._direction_rads=dir0.copy()
env=die_amd.Env.from_numpy(medium, agents)
sign=orng.turn_signs(3,0,N)
want=ref.forward((agents,medium))
und=ref.last_undetermined
for mode in ('seeded','explicit'):
    dev=die_amd.PhysarumAgent(max_agents=N, seed=3, **kw); dev.set_state(dir0)
    if mode=='explicit': dev.set_turn_signs(sign)
    got=dev.forward(env._get_current_obs).to_numpy()
    bad=~np.isclose(got,want,rtol=1e-5,atol=1e-9).all(axis=0)
    print(mode, bad.sum(), 'of', N, 'und', und.sum(), 'bad&und', (bad&und).sum(), 'first bad', np.nonzero(bad)[0][:10])
gd=dev._direction_rads.cpu().numpy()
for n in np.nonzero(bad)[0][:6]:
    print(n, 'alive',agents[2,n],'xy',agents[0,n]*(W-1),agents[1,n]*(H-1),'dir0',np.degrees(dir0[n]),'got',got[:,n]*(W-1),'want',want[:,n]*(W-1),'gotdir',np.degrees(gd[n]),'wantdir',np.degrees(ref._direction_rads[n]), 'und', und[n])
