"""The slab ring inside the C-ABI (ising_ring_* / ising_rank_*, csrc/ising_ring.cpp): edge rows, halo delivery on a
second stream per slab, interior rows.  On one GPU the copy transport carries n slabs of one device and the RCCL
transport carries a ring of ONE slab (ncclSend/ncclRecv to itself: RCCL refuses two ranks on one device); with two or
more GPUs the tests at the end run the real thing (they skip on a 1-GPU box)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _single(X, Y, seed, temp, sweeps, **kw):
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, **kw) as one:
        one.init().sweep(sweeps)
        return one.read(ig.BLACK), one.read(ig.WHITE), one.count(), one.bond_equal()


@pytest.mark.parametrize("layout", [ig.LAYOUT_BALLOT, ig.LAYOUT_DENSE, ig.LAYOUT_NIBBLE])
@pytest.mark.parametrize("nslabs", [2, 3, 8])
@pytest.mark.parametrize("store", [None, "1"])
def test_capi_ring_copy_transport_matches_single_slab(gpu, monkeypatch, nslabs, layout, store):
    """Slabs of one device: with ghost rows (the ballot layout's default) fused launches that take turns, with copies of the
    ghost rows in between; without them -- or with ISING_RING_STORE=1 -- every launch stores its edge rows straight into the
    neighbours' halo rows."""
    if store:
        monkeypatch.setenv("ISING_RING_STORE", store)
    X, Yk, seed, temp, sweeps = 8192, 32, 99, ig.CRIT_TEMP_F32, 6
    ref_b, ref_w, ref_cnt, ref_bond = _single(X, Yk * nslabs, seed, temp, sweeps, layout=layout)
    ring = ig.SlabSet([ig.IsingSlab(X, Yk, seed=seed, temp=temp, nslabs=nslabs, slab=k, layout=layout) for k in range(nslabs)])
    try:
        ring.init()
        assert ring.transport == ig.TRANSPORT_COPY
        ring.sweep(2).sweep(sweeps - 2)
        assert ring.count() == ref_cnt
        assert ring.bond_equal() == ref_bond
        assert np.array_equal(np.concatenate([s.read(ig.BLACK) for s in ring.slabs]), ref_b)
        assert np.array_equal(np.concatenate([s.read(ig.WHITE) for s in ring.slabs]), ref_w)
    finally:
        ring.close()


def test_capi_ring_rejects_mixed_layouts(gpu):
    slabs = [ig.IsingSlab(8192, 32, nslabs=2, slab=0, layout=ig.LAYOUT_BALLOT), ig.IsingSlab(8192, 32, nslabs=2, slab=1, layout=ig.LAYOUT_DENSE)]
    ring = ig.SlabSet(slabs)
    try:
        with pytest.raises(ig.IsingError, match="layout"):
            ring.init()
    finally:
        ring.close()


def test_capi_ring_ballot_slabs_turn_dense_together(gpu, oracle_mod):
    """T <= 0 has no integer thresholds: every ballot slab of the ring must become dense before the first row travels."""
    X, Yk, n, seed = 8192, 32, 3, 5
    ring = ig.SlabSet([ig.IsingSlab(X, Yk, seed=seed, temp=1.5, nslabs=n, slab=k, layout=ig.LAYOUT_BALLOT) for k in range(n)])
    try:
        ring.init().sweep(2)
        ring.set_temperature(-1.0)
        ring.sweep(2)
        assert all(s.current_layout() == ig.LAYOUT_DENSE for s in ring.slabs)
        orc = oracle_mod.OracleLattice(X, Yk * n, seed=seed, temp=1.5).init().sweep(2)
        orc.temp = -1.0
        orc.sweep(2)
        assert ring.count() == orc.count()
        assert np.array_equal(np.concatenate([s.read(ig.BLACK) for s in ring.slabs]), orc.black)
    finally:
        ring.close()


@pytest.mark.parametrize("layout", [ig.LAYOUT_BALLOT, ig.LAYOUT_DENSE])
def test_ring_of_one_over_rccl(gpu, layout):
    """ising_rank_*: the library's RCCL ring with one rank -- the slab's own edge rows travel through ncclSend/ncclRecv on
    the comm stream into its halo rows (ring_halo), instead of the mirror rows the kernels keep for a plain single slab."""
    import torch  # noqa: F401  (the RCCL copy torch ships is the one the process already holds)
    assert ig.rccl_version() > 0
    X, Y, seed, temp, sweeps = 8192, 64, 31, ig.CRIT_TEMP_F32, 7
    ref_b, ref_w, ref_cnt, ref_bond = _single(X, Y, seed, temp, sweeps, layout=layout)
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, layout=layout, ring_halo=True) as s:
        ring = ig.NativeRing(s).init()
        ring.sweep(3).sweep(sweeps - 3)
        ring.quiesce()
        assert ring.count() == ref_cnt
        assert ring.bond_equal() == ref_bond
        assert np.array_equal(s.read(ig.BLACK), ref_b) and np.array_equal(s.read(ig.WHITE), ref_w)
        ring.close()


def test_ring_of_one_copy_transport_and_transport_switch(gpu):
    X, Y, seed, temp, sweeps = 4096, 48, 8, 2.0, 5
    ref_b, ref_w, ref_cnt, _ = _single(X, Y, seed, temp, sweeps)
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, ring_halo=True) as s:
        ring = ig.SlabSet([s]).init()
        assert ring.transport == ig.TRANSPORT_COPY
        ring.sweep(sweeps)
        assert ring.count() == ref_cnt
        assert np.array_equal(s.read(ig.BLACK), ref_b) and np.array_equal(s.read(ig.WHITE), ref_w)
        with pytest.raises(ig.IsingError):  # a plain sweep would need the kernels' mirror rows
            s.sweep(1)


def test_caller_buffers_are_validated(gpu):
    import torch
    small = torch.empty(1024, dtype=torch.uint8, device="cuda")
    with pytest.raises(ig.IsingError, match="needs"):
        ig.IsingSlab(8192, 64, lattice_mem=small.data_ptr(), lattice_mem_bytes=small.numel())
    host = np.zeros(1 << 20, dtype=np.uint8)
    with pytest.raises(ig.IsingError):
        ig.IsingSlab(8192, 64, lattice_mem=host.ctypes.data)
    assert ig.required_bytes(8192, 64, ig.LAYOUT_BALLOT) * 4 == ig.required_bytes(8192, 64) == ig.required_bytes(8192, 64, ig.LAYOUT_NIBBLE)


# ---- two or more GPUs (the driver's multi-GPU node; skipped on the 1-GPU box) ---------------------------------------
def _ngpu():
    return ig.device_count()


@pytest.mark.parametrize("transport", [ig.TRANSPORT_RCCL, ig.TRANSPORT_COPY])
def test_multi_device_ring_single_process(gpu, oracle_mod, transport):
    n = min(_ngpu(), 4)
    if n < 2:
        pytest.skip("needs >= 2 GPUs")
    X, Yk, seed, temp, sweeps = 8192, 64, 2024, ig.CRIT_TEMP_F32, 6
    orc = oracle_mod.OracleLattice(X, Yk * n, seed=seed, temp=temp).init().sweep(sweeps)
    ring = ig.SlabSet([ig.IsingSlab(X, Yk, seed=seed, temp=temp, nslabs=n, slab=k, device=k) for k in range(n)])
    try:
        ring.set_transport(transport)
        ring.init().sweep(sweeps)
        assert ring.transport == transport
        assert ring.count() == orc.count()
        assert ring.bond_equal() == orc.bond_equal()
        assert np.array_equal(np.concatenate([s.read(ig.BLACK) for s in ring.slabs]), orc.black)
        assert np.array_equal(np.concatenate([s.read(ig.WHITE) for s in ring.slabs]), orc.white)
    finally:
        ring.close()


@pytest.mark.parametrize("mode", ["native", "p2p", "allgather"])
def test_multi_process_ring_over_nccl(gpu, mode):
    """One process per GPU under torch.distributed.run, backend nccl (= RCCL): the library's own RCCL ring (NativeRing)
    and the unmodified torch.distributed SlabRing (p2p and all-gather exchange), every rank against the CPU oracle."""
    n = min(_ngpu(), 4)
    if n < 2:
        pytest.skip("needs >= 2 GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "tools", "ring_ranks_nccl.py"), mode],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("slab == oracle rows") == 2 * n and "!=" not in r.stdout, r.stdout[-3000:]


# ---- the path a scaling run of bench.py takes: ballot ring slabs, 64 ghost rows, fused launches, the exchange overlapped -----------
def _scale(transport, world, port, *mode, timeout=1800):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("ISING_RING_GHOST", "ISING_RING_OVERLAP", "ISING_FUSED"):
        env.pop(k, None)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), os.path.join(ROOT, "tools", "ring_ranks_scale.py"), transport, *mode],
                          capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.parametrize("world,workload,port", [(2, "config3", 29561), (4, "config3", 29562), (8, "config3", 29563), (8, "strong", 29564),
                                                 (4, "strong", 29565), (2, "config4", 29566)])
def test_scaling_path_ipc_ranks_reproduce_the_goldens(gpu, world, workload, port):
    """bench.py's own slabs at N ranks over the library's peer ring (ISING_TRANSPORT_IPC): every rank a device of its own where the
    node has them, sharing devices where it has fewer (a 1-GPU box runs all of it) -- 64 ghost rows, fused launches of up to 32
    sweeps, the exchange overlapped with the launches; counts after 5 / 21 / 25 sweeps = the oracle's goldens of the total lattice."""
    r = _scale("ipc", world, port, "golden", workload)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    npts = 2 if workload == "config4" else 3
    assert r.stdout.count("== oracle golden") == npts * world and "!=" not in r.stdout, r.stdout[-3000:]
    assert r.stdout.count("exchange stats:") == world


@pytest.mark.parametrize("world,workload,port", [(2, "config3", 29571), (4, "config3", 29572), (8, "config3", 29573), (8, "strong", 29574), (8, "config4", 29575)])
def test_scaling_path_rccl_ranks_reproduce_the_goldens(gpu, world, workload, port):
    """The same over the library's RCCL ring (ncclSend / ncclRecv on the comm stream, one GPU per rank): what the driver's scaling run
    executes first.  Needs `world` GPUs."""
    if _ngpu() < world:
        pytest.skip(f"needs >= {world} GPUs")
    r = _scale("rccl", world, port, "golden", workload)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    npts = 2 if workload == "config4" else 3
    assert r.stdout.count("== oracle golden") == npts * world and "!=" not in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("transport,world,port", [("ipc", 2, 29581), ("ipc", 4, 29582), ("rccl", 2, 29583), ("rccl", 4, 29584), ("rccl", 8, 29585)])
def test_scaling_path_full_state_against_the_oracle(gpu, transport, world, port):
    """The same schedule on slabs small enough for the CPU oracle to follow (16384 x 2048 per rank, three exchanges deep): every
    rank's FULL state, the counts and the bond sum."""
    if transport == "rccl" and _ngpu() < world:
        pytest.skip(f"needs >= {world} GPUs")
    r = _scale(transport, world, port, "state")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("== oracle rows") == 3 * world and "!=" not in r.stdout, r.stdout[-3000:]


def test_ring_slabs_keep_ghost_rows_where_they_can(gpu, monkeypatch):
    """Ballot ring slabs that own their buffer keep ghost rows (min(64, Y/2) deep; ising_sweep_info: the ring sweeps them in
    fused launches of half as many sweeps), with -J too (the ghost rows' couplings are generated in place); the dense layout, a
    caller-owned buffer and ISING_RING_GHOST=1 do not."""
    import torch  # noqa: F401
    with ig.IsingSlab(8192, 256, nslabs=2, slab=0, layout=ig.LAYOUT_BALLOT) as s:
        assert s.fused and s.max_sweeps_per_launch == 32
    with ig.IsingSlab(8192, 32, nslabs=2, slab=0, layout=ig.LAYOUT_BALLOT) as s:
        assert s.fused and s.max_sweeps_per_launch == 8
    with ig.IsingSlab(8192, 128, nslabs=2, slab=0, layout=ig.LAYOUT_DENSE) as s:
        assert not s.fused and s.max_sweeps_per_launch == 0
    with ig.IsingSlab(8192, 128, nslabs=2, slab=0, layout=ig.LAYOUT_BALLOT, J_prob=0.2) as s:
        assert s.fused and s.max_sweeps_per_launch == 32
    b = ig.HipSlabBackend.create(8192, 128, device=0, nslabs=2, slab=0, layout=ig.LAYOUT_BALLOT)
    try:
        assert not b.slab.fused
    finally:
        b.slab.close()
    monkeypatch.setenv("ISING_RING_GHOST", "1")
    with ig.IsingSlab(8192, 128, nslabs=2, slab=0, layout=ig.LAYOUT_BALLOT) as s:
        assert not s.fused


def _deliver(slabs, torch):
    """The caller's own transport for the deep exchange surface: device-to-device copies between the slabs' ghost blocks."""
    from ising_gpu_amd.ring import _DevMem
    dev = torch.device("cuda", 0)
    n = len(slabs)
    for s in slabs:
        s.synchronize()
    for color in (ig.BLACK, ig.WHITE):
        blocks = []
        for s in slabs:
            depth, ptrs, nb = s.ghost_ptrs(color)
            blocks.append([torch.as_tensor(_DevMem(p, nb), device=dev) for p in ptrs])
        for k in range(n):
            send_top, send_bot, _, _ = blocks[k]
            blocks[(k + 1) % n][2].copy_(send_bot)  # next slab's rows above row 0 <- my last rows
            blocks[(k - 1) % n][3].copy_(send_top)  # previous slab's rows below row Y-1 <- my first rows
        for s in slabs:
            s.ghost_delivered(color)
    torch.cuda.synchronize()


@pytest.mark.parametrize("nslabs,Yk,sweeps", [(2, 64, (16, 5, 1)), (3, 32, (8, 8, 3)), (1, 64, (7, 16))])
def test_deep_exchange_surface_with_a_callers_own_copies(gpu, nslabs, Yk, sweeps):
    """ising_ghost_ptrs / ising_ghost_delivered / ising_sweep_ghost: what a caller with its own transport (MPI, ...) drives.
    Here the 'transport' is a device copy; the slabs of one lattice, swept G/2 sweeps per exchange, equal the single slab."""
    import torch
    X, seed, temp = 8192, 4242, ig.CRIT_TEMP_F32
    total = sum(sweeps)
    ref_b, ref_w, ref_cnt, _ = _single(X, Yk * nslabs, seed, temp, total, layout=ig.LAYOUT_BALLOT)
    slabs = [ig.IsingSlab(X, Yk, seed=seed, temp=temp, nslabs=nslabs, slab=k, layout=ig.LAYOUT_BALLOT, ring_halo=nslabs == 1) for k in range(nslabs)]
    try:
        G = slabs[0].ghost_ptrs(ig.BLACK)[0]
        assert G == min(64, Yk // 2)
        for s in slabs:
            s.init()
        with pytest.raises(ig.IsingError, match="not current"):
            slabs[0].sweep_ghost(1)  # nothing delivered yet
        for n in sweeps:
            left = n
            while left:
                ns = min(left, G // 2)
                _deliver(slabs, torch)
                for s in slabs:
                    s.sweep_ghost(ns)
                left -= ns
        with pytest.raises(ig.IsingError, match="not current"):
            slabs[0].sweep_ghost(1)  # the launch made the ghost rows stale
        _deliver(slabs, torch)
        with pytest.raises(ig.IsingError, match="at most"):
            slabs[0].sweep_ghost(G // 2 + 1)
        assert np.array_equal(np.concatenate([s.read(ig.BLACK) for s in slabs]), ref_b)
        assert np.array_equal(np.concatenate([s.read(ig.WHITE) for s in slabs]), ref_w)
        ups = sum(s.count()[0] for s in slabs)
        assert ups == ref_cnt[0]
    finally:
        for s in slabs:
            s.close()


def test_deep_exchange_surface_is_one_row_deep_without_ghost_rows(gpu):
    with ig.IsingSlab(8192, 32, nslabs=2, slab=0, layout=ig.LAYOUT_DENSE) as s:
        depth, ptrs, nb = s.ghost_ptrs(ig.WHITE)
        assert depth == 1 and (ptrs, nb) == s.halo_ptrs(ig.WHITE)
        s.init()
        s.ghost_delivered(ig.BLACK)
        s.ghost_delivered(ig.WHITE)
        with pytest.raises(ig.IsingError, match="ghost rows"):
            s.sweep_ghost(1)
    with ig.IsingSlab(8192, 32, layout=ig.LAYOUT_BALLOT) as s:  # a lone slab wraps in place
        with pytest.raises(ig.IsingError, match="wrap"):
            s.ghost_ptrs(ig.BLACK)


@pytest.mark.parametrize("transport,world,port", [("ipc", 2, 29591), ("ipc", 4, 29592), ("rccl", 2, 29593), ("rccl", 8, 29594)])
def test_scaling_path_print_points_inside_the_launches(gpu, transport, world, port):
    """ising_rank_sweep_counted: the reference's -p N counts of the WHOLE lattice, taken by every rank's deep launches over its own rows and summed over the rank
    transport -- against the CPU oracle, calls of uneven lengths, the state afterwards."""
    if transport == "rccl" and _ngpu() < world:
        pytest.skip(f"needs >= {world} GPUs")
    r = _scale(transport, world, port, "counted")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("counts == oracle") == 3 * world and r.stdout.count("slab == oracle rows") == world and "!=" not in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("layout,nslabs,Y", [(ig.LAYOUT_BALLOT, 2, 256), (ig.LAYOUT_BALLOT, 3, 128), (ig.LAYOUT_BALLOT, 1, 512), (ig.LAYOUT_DENSE, 2, 64)])
def test_ring_sweep_counted_one_process(gpu, oracle_mod, monkeypatch, layout, nslabs, Y):
    """ising_ring_sweep_counted, all slabs in one process: inside the deep launches on the ballot layout (ghost rows: the slabs' own rows only are counted), sweeps and
    counts in turn on the dense layout -- the same counts and state as the CPU oracle either way."""
    X, seed = 8192, 313
    # (ballot slabs: REQUIRE the counts inside the launches -- ISING_RING_COUNTED=2 turns a silent fall-back into an error; the copies then travel on the comm streams)
    monkeypatch.setenv("ISING_RING_COUNTED", "2" if layout == ig.LAYOUT_BALLOT else "1")
    if layout == ig.LAYOUT_BALLOT:
        monkeypatch.setenv("ISING_RING_INLINE", "0")
    orc = oracle_mod.OracleLattice(X, Y * nslabs, seed=seed, temp=ig.CRIT_TEMP_F32).init()
    ring = ig.SlabSet([ig.IsingSlab(X, Y, seed=seed, temp=ig.CRIT_TEMP_F32, nslabs=nslabs, slab=k, layout=layout, ring_halo=(nslabs == 1)) for k in range(nslabs)])
    try:
        ring.init()
        for n, every, energy in ((40, 16, False), (70, 16, True), (9, 1, True), (23, 100, False)):
            got = ring.sweep_counted(n, every, energy)  # (energy: the bond sum of the whole lattice, every slab's white levels over its own rows)
            want = []
            for _ in range(n):
                orc.sweep(1)
                if orc.it % every == 0:
                    want.append(orc.count() + ((orc.bond_equal(),) if energy else ()))
            assert got == want, (layout, nslabs, n, every)
        assert np.array_equal(np.concatenate([s.read(ig.BLACK) for s in ring.slabs]), orc.black)
        assert np.array_equal(np.concatenate([s.read(ig.WHITE) for s in ring.slabs]), orc.white)
        assert ring.count() == orc.count()
    finally:
        ring.close()
