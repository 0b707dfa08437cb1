"""nslabs > 1 on ONE GPU: several slab contexts on device 0 exchanging halo rows in-process (LocalRing) must
reproduce the single-slab result bit for bit -- exercises the halo-row kernel path, the boundary/interior strip
split and the zero-copy torch views of the library's halo buffers that the RCCL path uses."""
import numpy as np
import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nslabs,strip", [(2, 16), (4, 8), (3, 16)])
def test_local_ring_matches_single_slab(gpu, nslabs, strip):
    X, Y, seed, temp, sweeps = 4096, 48 * nslabs, 2024, ig.CRIT_TEMP_F32, 5
    with ig.IsingSlab(X, Y, seed=seed, temp=temp) as one:
        one.init().sweep(sweeps)
        ref_b, ref_w, ref_cnt, ref_bond = one.read(ig.BLACK), one.read(ig.WHITE), one.count(), one.bond_equal()
    slabs = [ig.IsingSlab(X, Y // nslabs, seed=seed, temp=temp, nslabs=nslabs, slab=k, strip_rows=strip) for k in range(nslabs)]
    try:
        ring = ig.LocalRing([ig.HipSlabBackend(s) for s in slabs]).init()
        ring.sweep(sweeps)
        got_b = np.concatenate([s.read(ig.BLACK) for s in slabs])
        got_w = np.concatenate([s.read(ig.WHITE) for s in slabs])
        assert np.array_equal(got_b, ref_b) and np.array_equal(got_w, ref_w)
        assert ring.count() == ref_cnt
        assert ring.bond_equal() == ref_bond
    finally:
        for s in slabs:
            s.close()


def test_single_rank_slabring_is_plain_sweep(gpu, oracle_mod):
    with ig.IsingSlab(2048, 64, seed=11, temp=2.0) as s:
        ring = ig.SlabRing(ig.HipSlabBackend(s)).init()
        ring.sweep(4)
        orc = oracle_mod.OracleLattice(2048, 64, seed=11, temp=2.0).init().sweep(4)
        assert np.array_equal(s.read(ig.BLACK), orc.black) and np.array_equal(s.read(ig.WHITE), orc.white)
        assert ring.count() == orc.count()


def test_local_ring_on_torch_owned_buffers(gpu):
    """HipSlabBackend.create: torch allocates the slab buffers (the C-ABI sees plain pointers); halo tensors are
    slices of those tensors -- the form the RCCL path sends and receives."""
    X, Y, n, seed, temp, sweeps = 4096, 144, 3, 77, 2.0, 4
    with ig.IsingSlab(X, Y, seed=seed, temp=temp) as one:
        one.init().sweep(sweeps)
        ref_b, ref_w = one.read(ig.BLACK), one.read(ig.WHITE)
    backs = [ig.HipSlabBackend.create(X, Y // n, seed=seed, temp=temp, nslabs=n, slab=k) for k in range(n)]
    try:
        ring = ig.LocalRing(backs).init()
        ring.sweep(sweeps)
        assert np.array_equal(np.concatenate([b.slab.read(ig.BLACK) for b in backs]), ref_b)
        assert np.array_equal(np.concatenate([b.slab.read(ig.WHITE) for b in backs]), ref_w)
        for b in backs:
            for t in b.halo_tensors(ig.BLACK):
                assert t.untyped_storage().data_ptr() == b._buffers["lattice"].untyped_storage().data_ptr()
    finally:
        for b in backs:
            b.slab.close()
