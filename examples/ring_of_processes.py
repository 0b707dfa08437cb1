#!/usr/bin/env python3
"""One process per slab (the reference is one process for all GPUs): every rank owns rows [rank * Y, (rank + 1) * Y) of a periodic
lattice and the library moves the ghost rows -- RCCL between GPUs, the peer transport (hipIpcMemHandle) where that does not come up,
e.g. when the ranks share a device.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29700 examples/ring_of_processes.py"""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ising_gpu_amd as ig  # noqa: E402

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
ndev = torch.cuda.device_count()
device = local % ndev
torch.cuda.set_device(device)
shared = world > ndev                                                   # more ranks than GPUs: RCCL refuses, the peer transport does not
dist.init_process_group("gloo" if shared else "nccl")
X, Y = 32768, 8192                                                      # per rank
slab = ig.IsingSlab(X, Y, device=device, seed=1234, temp=ig.CRIT_TEMP_F32, nslabs=world, slab=rank)
ring = ig.open_native_ring(slab, transports=("ipc",) if shared else ("ipc", "rccl"))
assert ring is not None, "no ring transport came up"
ring.init()                                                             # (open_native_ring swept once to try the transport: start over)
ring.sweep(128)
up, down = ring.count()                                                 # collective: totals over all slabs
bonds = ring.bond_equal()
if rank == 0:
    print(f"{world} slabs of {Y} x {X}: after {ring.it} sweeps up {up}, down {down}, e = {ig.energy_per_spin(bonds, X * Y * world):.6f}")
ring.close()
slab.close()
dist.barrier()
dist.destroy_process_group()
