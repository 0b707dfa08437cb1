"""Can two ranks share one GPU under RCCL?  (Only to exercise the SlabRing/NCCL code path on a 1-GPU box.)"""
import os
import sys
import torch
import torch.distributed as dist

sys.path.insert(0, ".")
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.full((4,), rank, device="cuda", dtype=torch.int64)
dist.all_reduce(t)
print(rank, "allreduce ok", t.tolist(), flush=True)
a = torch.full((16,), rank + 10, device="cuda", dtype=torch.uint8)
b = torch.zeros(16, device="cuda", dtype=torch.uint8)
ops = [dist.P2POp(dist.isend, a, (rank + 1) % world), dist.P2POp(dist.irecv, b, (rank - 1) % world)]
for w in dist.batch_isend_irecv(ops):
    w.wait()
torch.cuda.synchronize()
print(rank, "p2p ok", b[:4].tolist(), flush=True)
dist.destroy_process_group()
