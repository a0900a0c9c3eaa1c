"""Round 6: where does a resident-workgroup slot's idle time go?  Per-CU timeline of one step's two kernels at the bench's window from a
-DPIC_STAMPS -DPIC_STAMPS_RT build (scratch/build_stamps.sh with STAMP_EXTRA=-DPIC_STAMPS_RT; DIE_AMD_LIB): every workgroup leaves its
kernel-ENTRY time, its phase stamps, its end and the CU it ran on (HW_ID / XCC_ID).  s_memrealtime: 10 ns ticks, one clock for the GPU."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
import die_amd
W = 4096; STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 205
fields = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == 'f16') else torch.float32
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False, field_dtype=fields)
ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
obs = env._get_current_obs
for _ in range(STEPS):
    obs, *_ = env.step(ag.forward(obs))
torch.cuda.synchronize()
pic = env._pic
NT = pic.NT
raw = pic.error[2:].cpu().numpy().view(np.uint64)
st = raw[:NT * 16].reshape(NT, 16).astype(np.float64)
hw = raw[NT * 16:NT * 20].reshape(NT, 4)
TICK = 0.01
pop = pic.meta[1 - pic.cur][1].cpu().numpy().astype(np.int64)
for name, hwi, a, b, end, slots in (('agent kernel', 0, 0, 5, 7, 3), ('field kernel', 2, 8, 14, 14, 4)):
    ident, entry = hw[:, hwi], hw[:, hwi + 1].astype(np.float64)
    s0, s1, e1 = st[:, a], st[:, b], st[:, end]
    ok = (entry > 0) & (s0 >= entry) & (e1 >= s1) & (s1 > s0)
    t0 = entry[ok].min()
    entry, s0, s1, e1 = (entry - t0) * TICK, (s0 - t0) * TICK, (s1 - t0) * TICK, (e1 - t0) * TICK
    xcc = (ident >> np.uint64(32)).astype(np.int64) & 0xF
    cu = (ident.astype(np.int64) >> 8) & 0xFF                  # CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    key = xcc * 256 + cu
    keys = np.unique(key[ok])
    dur = e1[ok].max()
    print(f'== {name}: {ok.sum()} workgroups on {len(keys)} CUs of {len(np.unique(xcc[ok]))} XCDs; launch {dur:.1f} us (first entry -> last end)')
    print(f'   entry -> first stamp (prologue) mean {np.mean((s0 - entry)[ok]):.2f} us p95 {np.percentile((s0 - entry)[ok], 95):.2f};  '
          f'life (first stamp -> behind the last barrier) mean {np.mean((s1 - s0)[ok]):.2f};  last barrier -> end (epilogue, wave 0) mean {np.mean((e1 - s1)[ok]):.2f} p95 {np.percentile((e1 - s1)[ok], 95):.2f}')
    if name == 'agent kernel' and st[:, 6].any():
        ack = (st[:, 6] - st[:, 7]) * TICK
        print(f'   wave 0: end of the epilogue -> its stores acknowledged (s_waitcnt vmcnt(0)): mean {ack[ok].mean():.2f} us median {np.median(ack[ok]):.2f} p90 {np.percentile(ack[ok], 90):.2f}')
    per_cu = np.array([(key[ok] == k).sum() for k in keys])
    print(f'   workgroups per CU: min {per_cu.min()} mean {per_cu.mean():.1f} max {per_cu.max()};  agents per CU: max / mean {max(pop[ok][key[ok] == k].sum() for k in keys) / (pop[ok].sum() / len(keys)):.3f}')
    xend = [e1[ok & (xcc == x)].max() for x in np.unique(xcc[ok])]
    print('   last end per XCD (us):', np.round(xend, 1).tolist())
    # per CU: resident workgroups over time (entry .. end), gaps between a workgroup's end and the next entry on that CU while the queue was not empty
    gaps, conc = [], np.zeros(slots + 3)
    last_entry_all = entry[ok].max()
    for k in keys:
        m = ok & (key == k)
        ev = sorted([(t, +1) for t in entry[m]] + [(t, -1) for t in e1[m]])
        n, prev = 0, 0.0
        for t, d in ev:
            if t > prev:
                conc[min(n, slots + 2)] += min(t, last_entry_all) - min(prev, last_entry_all)      # (only while workgroups were still being dispatched somewhere)
            n += d; prev = t
        en, ex = np.sort(entry[m]), np.sort(e1[m])
        # the i-th end (in time order) frees a slot; the (slots + i)-th entry takes one: gap = entry - end
        for i in range(len(en) - slots):
            gaps.append(en[slots + i] - ex[i])
    gaps = np.array(gaps)
    # how fast CAN workgroups be launched?  entries per microsecond (whole GPU) around the first round's simultaneous turnover
    hist = np.bincount(np.clip(entry[ok].astype(int), 0, 29), minlength=30)
    print('   kernel entries per us (whole GPU), us 0..29:', hist.tolist())
    frac = conc / conc.sum()
    print('   share of (CU x time until the last dispatch) with 0 / 1 / 2 / 3 / 4 … workgroups between entry and end:', np.round(frac, 3).tolist())
    print(f'   mean resident workgroups per CU over that time: {sum(i * f for i, f in enumerate(frac)):.2f} of {slots}')
    print(f'   i-th end -> (i + {slots})-th entry on the same CU: mean {gaps.mean():.2f} us median {np.median(gaps):.2f} p10 {np.percentile(gaps, 10):.2f} p90 {np.percentile(gaps, 90):.2f} ({(gaps < 0).mean() * 100:.0f} % negative: the slot was not the one that freed i-th)')
