#!/usr/bin/env python3
"""Cost of one torch.distributed all_reduce (RCCL, this process group) as the BA reduce hook pays it: host time per call
and GPU-side gap, for the sizes the hook sends."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl")
for n in (16, 2000, 2003000):
    t = torch.ones(n, dtype=torch.float64, device="cuda")
    for _ in range(5): dist.all_reduce(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): dist.all_reduce(t)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # interleaved with a small kernel on the current stream, synchronised every time (what a host-paced loop sees)
    t3 = time.perf_counter()
    for _ in range(100):
        t.add_(1.0); dist.all_reduce(t); t.add_(1.0); torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"n={n}: enqueue {1e6 * (t1 - t0) / 200:.1f} us/call, drained {1e6 * (t2 - t0) / 200:.1f} us/call, kernel+allreduce+kernel+sync {1e6 * (t4 - t3) / 100:.1f} us", flush=True)
dist.destroy_process_group()
