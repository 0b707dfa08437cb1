"""The N>1 orchestration (ising_gpu_amd/ring.py SlabRing) on CPU: world_size 2 and 3 over gloo, with the CPU
oracle as the slab backend, must reproduce the single-lattice oracle bit for bit (decomposition invariance)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from ising_gpu_amd.ring import SlabRing

X, YTOT, SEED, TEMP, SWEEPS = 2048, 96, 77, 2.1, 3


class OracleBackend:
    """Adapts oracle.OracleSlab (test infrastructure) to the SlabBackend protocol of the product's ring."""

    def __init__(self, slab):
        self.s = slab
        self._t = {}
        for c in (0, 1):
            self._t[c] = (torch.from_numpy(slab.lat[c, 0].view(np.uint8)), torch.from_numpy(slab.lat[c, -1].view(np.uint8)),
                          torch.from_numpy(slab.halo[c, 0].view(np.uint8)), torch.from_numpy(slab.halo[c, 1].view(np.uint8)))

    def init(self):
        self.s.init()

    def update_all(self, it, color):
        self.s.update_rows(it, color, 0, self.s.Y)

    def update_edges(self, it, color):
        self.s.update_rows(it, color, 0, 1)
        self.s.update_rows(it, color, self.s.Y - 1, self.s.Y)

    def update_interior(self, it, color):
        self.s.update_rows(it, color, 1, self.s.Y - 1)

    def halo_tensors(self, color):
        return self._t[color]

    def count_up_down(self):
        up = self.s.count_up()
        return up, 2 * self.s.Y * self.s.X // 2 - up

    def bond_equal(self):
        return self.s.bond_equal()


class OracleGhostBackend(OracleBackend):
    """The same with ghost rows G deep: the optional deep-exchange part of the protocol (ising_ghost_ptrs and friends)."""

    def __init__(self, slab):
        super().__init__(slab)
        G, Y = slab.G, slab.Y
        self.delivered = [0, 0]
        self._g = {c: tuple(torch.from_numpy(slab.ext[c, a:b].reshape(-1).view(np.uint8))
                            for a, b in ((G, 2 * G), (Y, Y + G), (0, G), (Y + G, Y + 2 * G))) for c in (0, 1)}

    def ghost_depth(self):
        return self.s.G

    def ghost_tensors(self, color):
        return self._g[color]

    def ghost_delivered(self, color):
        self.delivered[color] += 1

    def sweep_ghost(self, first_it, nsweeps):
        assert self.delivered[0] and self.delivered[1], "launch before both colours were delivered"
        self.delivered = [0, 0]
        self.s.sweep_ghost(first_it, nsweeps)

    def bond_equal(self):
        G, Y = self.s.G, self.s.Y
        self.s.halo[1, 0], self.s.halo[1, 1] = self.s.ext[1, G - 1], self.s.ext[1, Y + G]  # rows -1 / Y of the ghost rows
        return self.s.bond_equal()


def _worker(rank, world, port, q, exchange="p2p", ghost=0, sweeps=SWEEPS):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle.set_threads(1)
    try:
        if ghost:
            slab = oracle.OracleGhostSlab(X, YTOT // world, SEED, TEMP, world, rank, ghost)
            ring = SlabRing(OracleGhostBackend(slab), exchange=exchange).init()
        else:
            slab = oracle.OracleSlab(X, YTOT // world, SEED, TEMP, world, rank)
            ring = SlabRing(OracleBackend(slab), exchange=exchange).init()
        ring.sweep(sweeps)
        up, down = ring.count()
        bond = ring.bond_equal()
        ring.quiesce()
        q.put((rank, slab.lat.copy(), up, down, bond))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,exchange,ghost,sweeps", [(2, "p2p", 0, SWEEPS), (3, "p2p", 0, SWEEPS), (2, "allgather", 0, SWEEPS),
                                                        (3, "allgather", 0, SWEEPS), (2, "p2p", 8, 8), (3, "p2p", 8, 9), (3, "p2p", 16, 5)])
def test_ring_matches_single_lattice(world, exchange, ghost, sweeps):
    """ghost > 0: the deep exchange -- `ghost` rows of both colours every ghost/2 sweeps, one multi-level update in between."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, exchange, ghost, sweeps)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = oracle.OracleLattice(X, YTOT, seed=SEED, temp=TEMP).init().sweep(sweeps)
    full = np.concatenate([r[1] for r in res], axis=1)
    assert np.array_equal(full[0], ref.black)
    assert np.array_equal(full[1], ref.white)
    for r in res:
        assert (r[2], r[3]) == ref.count()
        assert r[4] == ref.bond_equal()
