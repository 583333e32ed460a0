"""Latency of the packed all-gather the sharded search issues per batch (run under torch.distributed.run).
Single-rank on a one-GPU box is only a floor for the call overhead; the real number needs N > 1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
dev = torch.device("cuda", torch.cuda.current_device())
dist.init_process_group("nccl", device_id=dev)
nbytes = 256 * 5 * 12
loc = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
all_ = torch.zeros(world * nbytes, dtype=torch.uint8, device=dev)
host = torch.empty(world * nbytes, dtype=torch.uint8).pin_memory()
for _ in range(20): dist.all_gather_into_tensor(all_, loc)
torch.cuda.synchronize()
n = 200
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(n): dist.all_gather_into_tensor(all_, loc)
e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
if rank == 0:
    print(f"all_gather {nbytes} B x{world}: host enqueue {1e6*(t1-t0)/n:.1f} us/call, device {1e3*e0.elapsed_time(e1)/n:.1f} us/call, wall {1e6*(t2-t0)/n:.1f}")
e0.record()
for _ in range(n): host.copy_(all_, non_blocking=True)
e1.record(); torch.cuda.synchronize()
if rank == 0: print(f"D2H copy: device {1e3*e0.elapsed_time(e1)/n:.1f} us/call")
dist.barrier(); dist.destroy_process_group()
