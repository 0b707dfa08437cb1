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


def _worker(rank, world, port, q, exchange="p2p"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle.set_threads(1)
    try:
        slab = oracle.OracleSlab(X, YTOT // world, SEED, TEMP, world, rank)
        ring = SlabRing(OracleBackend(slab), exchange=exchange).init()
        ring.sweep(SWEEPS)
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


@pytest.mark.parametrize("world,exchange", [(2, "p2p"), (3, "p2p"), (2, "allgather"), (3, "allgather")])
def test_ring_matches_single_lattice(world, exchange):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = oracle.OracleLattice(X, YTOT, seed=SEED, temp=TEMP).init().sweep(SWEEPS)
    full = np.concatenate([r[1] for r in res], axis=1)
    assert np.array_equal(full[0], ref.black)
    assert np.array_equal(full[1], ref.white)
    for r in res:
        assert (r[2], r[3]) == ref.count()
        assert r[4] == ref.bond_equal()
