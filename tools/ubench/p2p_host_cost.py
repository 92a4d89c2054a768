#!/usr/bin/env python3
"""Host cost of one halo exchange as bench.py's hooks issue it: a cached batch_isend_irecv (self send/recv on a 1-rank RCCL
group, 2 x 64 KiB like the 4096^2 strip) + wait(), with a 100 us kernel between start and wait.  Prints host us per call
and the steady-state GPU-side period."""
import os, time, sys
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
send = torch.zeros(2 * 2 * 4096, dtype=torch.float64, device="cuda")
recv = torch.zeros_like(send)
ops = [dist.P2POp(dist.irecv, recv, 0), dist.P2POp(dist.isend, send, 0)]
big = torch.zeros(64 << 20, dtype=torch.float64, device="cuda")   # 512 MiB: a copy_ of it takes ~150 us
dst = torch.empty_like(big)
def once(with_kernel):
    works = dist.batch_isend_irecv(ops)
    if with_kernel:
        dst.copy_(big)
    for w in works:
        w.wait()
for with_kernel in (False, True):
    for _ in range(20):
        once(with_kernel)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        once(with_kernel)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"with_kernel={with_kernel}: host issue {1e6 * (t1 - t0) / n:.1f} us/exchange, end-to-end {1e6 * (t2 - t0) / n:.1f} us/iteration", flush=True)
dist.destroy_process_group()
