"""One process per GPU over the nccl (= RCCL) backend: tools/ring_ranks_nccl.py native|p2p|allgather.
native    -> ising_gpu_amd.NativeRing (the ring inside libising_hip.so: second stream + ncclSend/ncclRecv)
p2p       -> the unmodified ising_gpu_amd.SlabRing over torch.distributed batch_isend_irecv: a library-owned slab (ballot layout:
             ghost rows Y/2 = 32 deep here, through ising_ghost_ptrs / ising_sweep_ghost) and a torch-owned one (one row per colour half-sweep)
allgather -> SlabRing(exchange="allgather")
Every rank compares its slab and the global counts with the CPU oracle, for the ballot and the dense layout.
Launch: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 --master-port 29541 tools/ring_ranks_nccl.py native"""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ising_gpu_amd as ig  # noqa: E402
import oracle  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "native"
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
X, Y, seed, temp, sweeps = 8192, 64, 2024, ig.CRIT_TEMP_F32, 5
# native: a slab that owns its buffer (ballot layout: ghost rows Y/2 = 32 deep here, one exchange per 16 sweeps) and a torch-owned one (one halo row)
cases = [(lay, own) for lay in (ig.LAYOUT_BALLOT, ig.LAYOUT_DENSE) for own in ((True, False) if mode in ("native", "p2p") else (False,))]
for layout, own in cases:
    if own and mode == "native":
        class _B:  # same shape as HipSlabBackend for what follows
            slab = ig.IsingSlab(X, Y, device=local, seed=seed, temp=temp, nslabs=world, slab=rank, layout=layout)
        backend = _B
    elif own:
        backend = ig.HipSlabBackend(ig.IsingSlab(X, Y, device=local, seed=seed, temp=temp, nslabs=world, slab=rank, layout=layout))
    else:
        backend = ig.HipSlabBackend.create(X, Y, device=local, seed=seed, temp=temp, nslabs=world, slab=rank, layout=layout)
    ring = ig.NativeRing(backend.slab) if mode == "native" else ig.SlabRing(backend, exchange=mode)
    ring.init()
    if own and mode == "p2p":
        assert ring.ghost_rows == (32 if layout == ig.LAYOUT_BALLOT else 1), ring.ghost_rows
    ring.sweep(2).sweep(sweeps - 2)
    ring.quiesce()
    torch.cuda.synchronize()
    orc = oracle.OracleLattice(X, Y * world, seed=seed, temp=temp).init().sweep(sweeps)
    lo, hi = rank * Y, (rank + 1) * Y
    ok = np.array_equal(backend.slab.read(ig.BLACK), orc.black[lo:hi]) and np.array_equal(backend.slab.read(ig.WHITE), orc.white[lo:hi])
    tot, bond = ring.count(), ring.bond_equal()
    good = ok and tot == orc.count() and bond == orc.bond_equal()
    print(f"rank {rank} {mode} layout {layout} {'library-owned' if own else 'torch-owned'} buffer: slab {'==' if ok else '!='} oracle rows [{lo},{hi}); counts {tot} bond {bond} "
          f"{'==' if good else '!='} oracle", flush=True)
    assert good
    if mode == "native":
        ring.close()
    backend.slab.close()
dist.barrier()
dist.destroy_process_group()
