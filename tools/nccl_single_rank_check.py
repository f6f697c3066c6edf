#!/usr/bin/env python3
"""One-rank RCCL sanity check for the calls bench.py makes at N > 1 (init with device_id, all_reduce
probe, async all_gather_into_tensor with wait(), barrier(device_ids=...))."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
p = torch.ones(1, device="cuda")
dist.all_reduce(p)
a = torch.arange(8, dtype=torch.float64, device="cuda")
out = torch.empty(8, dtype=torch.float64, device="cuda")
w = dist.all_gather_into_tensor(out, a, async_op=True)
w.wait()
torch.cuda.synchronize()
assert torch.equal(out, a)
dist.barrier(device_ids=[0])
dist.destroy_process_group()
print("rccl single-rank check ok")
