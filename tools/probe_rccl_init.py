"""Does RCCL bootstrap on this box without NCCL_SOCKET_IFNAME?  world size 1, one all_reduce.  usage: python tools/probe_rccl_init.py [ifname]"""
import os, sys, time
import torch
import torch.distributed as dist
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = "29533"
if len(sys.argv) > 1:
    os.environ["NCCL_SOCKET_IFNAME"] = sys.argv[1]
os.environ.setdefault("NCCL_DEBUG", "WARN")
t0 = time.time()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
x = torch.ones(1 << 20, device="cuda")
dist.all_reduce(x)
torch.cuda.synchronize()
print("ok", os.environ.get("NCCL_SOCKET_IFNAME"), x[0].item(), f"{time.time() - t0:.1f} s", flush=True)
dist.destroy_process_group()
