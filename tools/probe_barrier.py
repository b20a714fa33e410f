"""Host cost of dist.barrier() / a tiny all_reduce on RCCL (one rank rehearsal: RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29514)."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.ones(1, device="cuda")
for name, fn in (("barrier", dist.barrier), ("all_reduce(1 float)+sync", lambda: (dist.all_reduce(t), torch.cuda.synchronize()))):
    for i in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        print(f"{name} #{i}: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
dist.destroy_process_group()
