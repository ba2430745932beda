"""Probe (and `-m gpu` test body): torch.distributed's nccl backend IS RCCL on ROCm -- create a one-rank communicator on
device 0, all-reduce device memory, barrier.  (Two ranks need two GPUs; this box has one.)  usage: rccl_one_rank.py [port]"""
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
port = int(sys.argv[1]) if len(sys.argv) > 1 else 29731
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
t = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.SUM)
t64 = torch.tensor([1.5, 2.0], dtype=torch.float64, device="cuda")      # bench.py reduces float64 timings
dist.all_reduce(t64, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("ok", float(t[12345]), float(t64[0]), dist.get_backend(), dist.get_world_size())
dist.destroy_process_group()
