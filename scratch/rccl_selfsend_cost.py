"""What a grouped point-to-point exchange costs by NUMBER of messages (RCCL, one rank sending to itself — the only peer a
one-GPU box offers): k messages of 1.3 MB / k each, batch_isend_irecv + wait, 200 exchanges.  usage: python scratch/rccl_selfsend_cost.py"""
import os, time, sys
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
total = 1300000
for k in (1, 2, 3, 5, 8):
    n = total // k
    s = [torch.zeros(n, dtype=torch.uint8, device='cuda') for _ in range(k)]
    r = [torch.zeros(n, dtype=torch.uint8, device='cuda') for _ in range(k)]
    ops = [dist.P2POp(dist.isend, t, 0) for t in s] + [dist.P2POp(dist.irecv, t, 0) for t in r]
    for _ in range(20):
        for w in dist.batch_isend_irecv(ops): w.wait()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); ev0.record()
    for _ in range(200):
        for w in dist.batch_isend_irecv(ops): w.wait()
    ev1.record(); torch.cuda.synchronize()
    print(f'{k} messages of {n} bytes: {ev0.elapsed_time(ev1) / 200 * 1e3:7.1f} us per exchange on the device, {(time.perf_counter() - t0) / 200 * 1e6:7.1f} us wall', flush=True)
dist.destroy_process_group()
