"""The multi-process slab ring (ising_gpu_amd.ring.SlabRing + HipSlabBackend) with several ranks on ONE GPU.
RCCL refuses two ranks per device, and gloo moves host memory, so the exchange is staged: the edge rows (slices of the
same torch-owned device buffer bench.py hands to RCCL) are copied to host tensors after a device synchronise, gloo
moves those, and the received rows are copied into the device halo rows before the next dependent launch.  What this
exercises is SlabRing's schedule (edges -> post -> interior, waits, init exchange) across real processes with the HIP
kernels; what it cannot exercise is RCCL's stream-ordered transport.  Each rank checks its slab against the CPU oracle.
Launch: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 --master-port 29533 tools/ring_two_ranks_one_gpu.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ising_gpu_amd as ig  # noqa: E402
import oracle  # noqa: E402


class StagedSlabRing(ig.SlabRing):
    """SlabRing whose row exchange goes through host staging buffers (see module docstring)."""

    def _post(self, color):
        send_top, send_bot, recv_top, recv_bot = self.b.halo_tensors(color)
        torch.cuda.synchronize()  # RCCL would order the sends after the edge-row kernel on the stream
        host = [send_bot.cpu(), send_top.cpu(), torch.empty_like(recv_top, device="cpu"), torch.empty_like(recv_bot, device="cpu")]
        ops = [dist.P2POp(dist.isend, host[0], self.next, self.group), dist.P2POp(dist.isend, host[1], self.prev, self.group),
               dist.P2POp(dist.irecv, host[2], self.prev, self.group), dist.P2POp(dist.irecv, host[3], self.next, self.group)]
        self._pending[color] = (dist.batch_isend_irecv(ops), host, recv_top, recv_bot)

    def _wait_rows(self, color):
        pend = self._pending[color]
        if pend:
            works, host, recv_top, recv_bot = pend
            for w in works:
                w.wait()
            recv_top.copy_(host[2])
            recv_bot.copy_(host[3])
        self._pending[color] = None


    # the deep exchange (library-owned ballot slabs with ghost rows), staged the same way
    def _post_deep(self):
        torch.cuda.synchronize()
        for color in (ig.BLACK, ig.WHITE):
            send_top, send_bot, recv_top, recv_bot = self.b.ghost_tensors(color)
            host = [send_bot.cpu(), send_top.cpu(), torch.empty_like(recv_top, device="cpu"), torch.empty_like(recv_bot, device="cpu")]
            ops = [dist.P2POp(dist.isend, host[0], self.next, self.group), dist.P2POp(dist.isend, host[1], self.prev, self.group),
                   dist.P2POp(dist.irecv, host[2], self.prev, self.group), dist.P2POp(dist.irecv, host[3], self.next, self.group)]
            self._pending_deep[color] = (dist.batch_isend_irecv(ops), host, recv_top, recv_bot)
        self._deep_posted = True

    def _wait_deep(self):
        for color in (ig.BLACK, ig.WHITE):
            pend = self._pending_deep[color]
            if pend:
                works, host, recv_top, recv_bot = pend
                for w in works:
                    w.wait()
                recv_top.copy_(host[2])
                recv_bot.copy_(host[3])
                self.b.ghost_delivered(color)
            self._pending_deep[color] = None


rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
X, Y, seed, temp, sweeps = 8192, 64, 2024, ig.CRIT_TEMP_F32, 5
for layout in (ig.LAYOUT_BALLOT, ig.LAYOUT_DENSE):
    backend = ig.HipSlabBackend.create(X, Y, device=0, seed=seed, temp=temp, nslabs=world, slab=rank, layout=layout)
    ring = StagedSlabRing(backend).init()
    ring.sweep(sweeps)
    ring.quiesce()
    torch.cuda.synchronize()
    orc = oracle.OracleLattice(X, Y * world, seed=seed, temp=temp).init().sweep(sweeps)
    lo, hi = rank * Y, (rank + 1) * Y
    ok = np.array_equal(backend.slab.read(ig.BLACK), orc.black[lo:hi]) and np.array_equal(backend.slab.read(ig.WHITE), orc.white[lo:hi])
    tot = ring.count()
    print(f"rank {rank} layout {layout}: slab {'==' if ok else '!='} oracle rows [{lo},{hi}); global counts {tot} {'==' if tot == orc.count() else '!='} oracle", flush=True)
    assert ok and tot == orc.count()
    backend.slab.close()

# Library-owned ballot slabs keep ghost rows (here Y/2 = 32 deep): 32 rows of both colours every 16 sweeps, one fused launch
# in between (ising_ghost_ptrs / ising_ghost_delivered / ising_sweep_ghost under SlabRing._sweep_deep).  21 sweeps = two
# launches (16 + 5) with an exchange between them, then 16 more on ghost rows that the last exchange left current: a launch
# of all 32 levels, after which no ghost row is valid until the next exchange (the bond sum reads rows -1 / Y).
sweeps = (21, 16)
slab = ig.IsingSlab(X, Y, device=0, seed=seed, temp=temp, nslabs=world, slab=rank, layout=ig.LAYOUT_BALLOT)
ring = StagedSlabRing(ig.HipSlabBackend(slab)).init()
assert ring.ghost_rows == 32, ring.ghost_rows
orc = oracle.OracleLattice(X, Y * world, seed=seed, temp=temp).init()
for n in sweeps:
    ring.sweep(n)
    orc.sweep(n)
    bond = ring.bond_equal()
    ring.quiesce()
    torch.cuda.synchronize()
    lo, hi = rank * Y, (rank + 1) * Y
    ok = np.array_equal(slab.read(ig.BLACK), orc.black[lo:hi]) and np.array_equal(slab.read(ig.WHITE), orc.white[lo:hi])
    tot = ring.count()
    print(f"rank {rank} ghost rows {ring.ghost_rows}, {ring.it} sweeps: slab {'==' if ok else '!='} oracle rows [{lo},{hi}); global counts {tot} "
          f"{'==' if tot == orc.count() else '!='} oracle; bond sum {'==' if bond == orc.bond_equal() else '!='} oracle", flush=True)
    assert ok and tot == orc.count() and bond == orc.bond_equal()
slab.close()
dist.barrier()
dist.destroy_process_group()
