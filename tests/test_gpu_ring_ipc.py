"""The N > 1 path of the C-ABI with REAL processes on a 1-GPU box: ising_rank_* over ISING_TRANSPORT_IPC (the reference's
own multi-GPU mechanism -- direct peer access to the neighbours' rows, optimized/main.cu:1496-1537, :1637-1642 -- across
processes: hipIpcMemHandle-mapped ghost rows, epoch counters in POSIX shared memory; csrc/ising_ipc.cpp).  2 and 3
processes share device 0 (RCCL refuses that; this transport does not): every rank compares its slab, the global counts
and the bond sum with the CPU oracle on small slabs (deep schedule with ghost rows, one halo row on two streams, -J), and
two ranks holding the bench's 65536^2 slabs reproduce the oracle's golden ring counts after 0 / 5 / 21 / 25 sweeps."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, port, mode, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("ISING_RING_GHOST", None)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), os.path.join(ROOT, "tools", "ring_ranks_ipc.py"), mode],
                          capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.parametrize("world,port", [(2, 29551), (3, 29552)])
def test_rank_ring_over_ipc_processes_share_one_gpu(gpu, world, port):
    r = _launch(world, port, "small")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    # 5 configurations, 3 + 2 + 2 + 2 + 1 sweep calls, every rank reports each; nothing differs
    assert r.stdout.count("== oracle") >= (2 * 10 + 1) * world and "!=" not in r.stdout, r.stdout[-3000:]
    assert r.stdout.count("== oracle (sub-lattice run)") == world, r.stdout[-3000:]
    # ... and the three configurations without couplings checkpoint, continue, load and continue again
    assert r.stdout.count("continuation == oracle") == 3 * world, r.stdout[-3000:]
    assert r.stdout.count("ghost rows 32") == 7 * world, r.stdout[-3000:]  # (5 sweep reports + the two checkpoint reports of the deep configurations)
    assert r.stdout.count("doomed checkpoint calls failed on this rank, the ring went on") == 3 * world, r.stdout[-3000:]


def test_two_ranks_over_ipc_reproduce_the_bench_ring_golden(gpu):
    r = _launch(2, 29553, "golden")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("== oracle golden") == 2 * 4 and "!=" not in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("world,port", [(2, 29554), (3, 29555)])
def test_ranks_over_ipc_soak_against_the_lone_slab(gpu, world, port):
    """24000 sweeps = 750+ overlapped exchanges between real processes, uneven call lengths: counts and bond sum of the ring ==
    the whole lattice swept as one slab, at six checkpoints."""
    r = _launch(world, port, "soak")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("== lone slab") == 6 * world and "!=" not in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("transport", ["ipc", "rccl"])
@pytest.mark.parametrize("overlap", ["0", "1", "2"])
def test_deep_exchange_schedules_agree(gpu, oracle_mod, monkeypatch, transport, overlap):
    """ISING_RING_OVERLAP: the exchange between two launches (0), in the running launch's tail with the next launch waiting on
    the stream (1, default), free-running (2; RCCL keeps the wait whatever is asked): same spins, uneven call lengths."""
    import ising_gpu_amd as ig
    monkeypatch.setenv("ISING_RING_OVERLAP", overlap)
    X, Y, seed = 16384, 512, 12
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=oracle_mod.CRIT_TEMP).init()
    slab = ig.IsingSlab(X, Y, seed=seed, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, ring_halo=True)
    ring = ig.NativeRing(slab, transport=transport).init()
    for n in (33, 1, 64, 7):
        ring.sweep(n)
        orc.sweep(n)
        assert ring.count() == orc.count() and ring.bond_equal() == orc.bond_equal()
    ring.quiesce()
    assert np.array_equal(slab.read(ig.BLACK), orc.black) and np.array_equal(slab.read(ig.WHITE), orc.white)
    ring.close()
    slab.close()


@pytest.mark.parametrize("transport", ["rccl", "ipc"])
def test_ring_of_one_checkpoints_through_the_rank_calls(gpu, oracle_mod, tmp_path, transport):
    """ising_rank_checkpoint_save / _load on a ring of one (their stage outcomes travel through the transport's all-reduce): the
    continuation repeats itself, and calls that cannot succeed return errors instead of hanging."""
    import ising_gpu_amd as ig
    X, Y, seed = 8192, 256, 5
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=oracle_mod.CRIT_TEMP).init()
    slab = ig.IsingSlab(X, Y, seed=seed, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, ring_halo=True)
    ring = ig.NativeRing(slab, transport=transport).init()
    ring.sweep(7)
    path = str(tmp_path / "one.ckpt")
    ring.checkpoint_save(path)
    ring.sweep(40)
    ring.checkpoint_load(path)
    assert ring.it == 7
    ring.sweep(40)
    orc.sweep(47)
    assert ring.count() == orc.count() and ring.bond_equal() == orc.bond_equal()
    with pytest.raises(ig.IsingError):
        ring.checkpoint_load(str(tmp_path / "missing.ckpt"))
    with pytest.raises(ig.IsingError):
        ring.checkpoint_save(str(tmp_path / "no_such_dir" / "x.ckpt"))
    ring.sweep(2)
    orc.sweep(2)
    assert ring.count() == orc.count()
    ring.close()
    slab.close()


@pytest.mark.parametrize("layout_name", ["ballot", "dense"])
def test_ring_of_one_over_ipc_in_process(gpu, oracle_mod, layout_name):
    """A ring of ONE slab attached to itself: its edge rows travel through the transport's copies and counters into its own
    halo / ghost rows -- what one rank of a ring executes, without a second process."""
    import ising_gpu_amd as ig
    layout = {"ballot": ig.LAYOUT_BALLOT, "dense": ig.LAYOUT_DENSE}[layout_name]
    X, Y, seed = 8192, 128, 77
    slab = ig.IsingSlab(X, Y, seed=seed, temp=ig.CRIT_TEMP_F32, layout=layout, ring_halo=True)
    slab.ipc_attach([slab.ipc_export()])
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=oracle_mod.CRIT_TEMP).init()
    slab.init()
    slab.rank_exchange(ig.BLACK)
    slab.rank_exchange(ig.WHITE)
    for n in (3, 40):
        slab.rank_sweep(n)
        orc.sweep(n)
        assert slab.rank_bond_equal() == orc.bond_equal()
        assert slab.rank_count() == orc.count()
        slab.rank_wait(-1)
        assert np.array_equal(slab.read(ig.BLACK), orc.black) and np.array_equal(slab.read(ig.WHITE), orc.white)
    slab.rank_detach()
    # detached: the rank calls are refused, a fresh export / attach works again
    with pytest.raises(ig.IsingError):
        slab.rank_sweep(1)
    slab.ipc_attach([slab.ipc_export()])
    slab.init()
    slab.rank_exchange(ig.BLACK)
    slab.rank_exchange(ig.WHITE)
    slab.rank_sweep(2)
    assert slab.rank_count() == oracle_mod.OracleLattice(X, Y, seed=seed, temp=oracle_mod.CRIT_TEMP).init().sweep(2).count()
    slab.close()


def test_ipc_attach_rejects_foreign_blobs(gpu):
    import ising_gpu_amd as ig
    a = ig.IsingSlab(8192, 64, nslabs=2, slab=0, layout=ig.LAYOUT_DENSE)
    b = ig.IsingSlab(8192, 128, nslabs=2, slab=1, layout=ig.LAYOUT_DENSE)  # another shape
    blobs = [a.ipc_export(), b.ipc_export()]
    with pytest.raises(ig.IsingError):
        a.ipc_attach(blobs)
    with pytest.raises(ig.IsingError):
        a.ipc_attach(blobs[:1])
    with pytest.raises(ig.IsingError):
        a.ipc_attach([bytes(256), bytes(256)])
    a.close()
    b.close()


@pytest.mark.parametrize("split", [None, "1"])
@pytest.mark.parametrize("ghost", ["8", "64"])
@pytest.mark.parametrize("epochs", ["1", "2", "3", None])
def test_several_exchange_epochs_in_one_launch(gpu, oracle_mod, monkeypatch, ghost, epochs, split):
    """Round 6 (ising_ring.cpp: epochs_per_launch; UpdateParams.epoch_sh): over the peer transport with a device to itself a ring slab's persistent launch carries
    several exchange epochs -- the trapezoid starts over, the edge units wait for each exchange inside the running launch -- instead of ending with every exchange.
    Same spins as the oracle's whatever the number of epochs a launch carries (1 = the form of rounds 3-5), ghost rows 8 and 64 deep, calls that end inside an epoch;
    the print points inside the launches (counts and bond sums) as well; and the launches really are fewer than the exchanges."""
    import ising_gpu_amd as ig
    monkeypatch.setenv("ISING_RING_GHOST", ghost)
    if split is None:  # (the fused form; "1": the split form -- draw units and word units --, whose launches carry epochs too from round 6 on)
        monkeypatch.delenv("ISING_SPLIT", raising=False)
    else:
        monkeypatch.setenv("ISING_SPLIT", split)
    if epochs is None:
        monkeypatch.delenv("ISING_RING_EPOCHS", raising=False)
    else:
        monkeypatch.setenv("ISING_RING_EPOCHS", epochs)
    X, Y, seed = 16384, 512, 12
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=oracle_mod.CRIT_TEMP).init()
    slab = ig.IsingSlab(X, Y, seed=seed, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, ring_halo=True)
    ring = ig.NativeRing(slab, transport="ipc").init()
    assert slab.split == (split == "1")
    slab.exchange_stats_begin(256)
    total = 0
    for n in (33, 1, 64, 7, 130):
        ring.sweep(n)
        orc.sweep(n)
        total += n
        assert ring.count() == orc.count() and ring.bond_equal() == orc.bond_equal(), (n, total)
    st = slab.exchange_stats_fetch()
    G = int(ghost)
    per_call = [-(-2 * n // G) for n in (33, 1, 64, 7, 130)]  # exchange epochs of every call
    assert st["exchanges"] == sum(per_call)
    E = {None: 64, "1": 1, "2": 2, "3": 3}[epochs]
    assert st["launches"] == sum(-(-e // E) for e in per_call), st
    ring.quiesce()
    assert np.array_equal(slab.read(ig.BLACK), orc.black) and np.array_equal(slab.read(ig.WHITE), orc.white)
    # the reference's print points, taken inside the launches of several epochs
    series = ring.sweep_counted(40, 8, True)
    want = []
    for k in range(40):
        orc.sweep(1)
        total += 1
        if total % 8 == 0:
            want.append((*orc.count(), orc.bond_equal()))
    assert [tuple(p) for p in series] == want
    ring.close()
    slab.close()


def test_ring_slab_of_few_tickets_runs_split_launches_of_several_epochs(gpu, oracle_mod, monkeypatch):
    """Round 6: where ising_create's rule gives a ring slab the split form (few tickets a level at sixteen-row strips -- the slab of a strong-scaling split), the
    ring's persistent launches of several exchange epochs run it by themselves (no ISING_SPLIT): same spins as the oracle's, print points (which keep the fused
    form: their slots are laid out by its strips) included."""
    import ising_gpu_amd as ig
    for k in ("ISING_SPLIT", "ISING_RING_EPOCHS", "ISING_RING_GHOST"):
        monkeypatch.delenv(k, raising=False)
    X, Y, seed = 8192, 8192, 21  # (the rule: few tickets a level at sixteen-row strips, 8192 rows and more)
    slab = ig.IsingSlab(X, Y, seed=seed, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, ring_halo=True)
    if not slab.split:
        slab.close()
        pytest.skip("the rule applies on a whole MI355X (eight XCDs)")
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=oracle_mod.CRIT_TEMP).init()
    ring = ig.NativeRing(slab, transport="ipc").init()
    slab.exchange_stats_begin(64)
    for n in (70, 33):
        ring.sweep(n)
        orc.sweep(n)
        assert ring.count() == orc.count() and ring.bond_equal() == orc.bond_equal(), n
    st = slab.exchange_stats_fetch()
    assert st["exchanges"] == 3 + 2 and st["launches"] == 2  # (70 and 33 sweeps at 32 an epoch: one launch each)
    series = ring.sweep_counted(24, 8, True)
    want = []
    for _ in range(24):
        orc.sweep(1)
        if orc.it % 8 == 0:
            want.append((*orc.count(), orc.bond_equal()))
    assert [tuple(p) for p in series] == want and len(want) == 3
    ring.quiesce()
    assert np.array_equal(slab.read(ig.BLACK), orc.black) and np.array_equal(slab.read(ig.WHITE), orc.white)
    ring.close()
    slab.close()


def test_ring_slab_times_both_forms_and_keeps_the_faster(gpu, monkeypatch):
    """... and which of the two forms a ring slab's long launches take is measured on the box (a warm and a timed launch of each, the lone slabs' guard): afterwards
    the slab runs the one that was faster, the record says which, and the spins are those of the same slab swept with the guard off."""
    import ising_gpu_amd as ig
    for k in ("ISING_SPLIT", "ISING_RING_EPOCHS", "ISING_RING_GHOST", "ISING_GUARD"):
        monkeypatch.delenv(k, raising=False)
    X, Y, seed = 16384, 8192, 5

    def run(guard):
        monkeypatch.setenv("ISING_GUARD", guard)
        slab = ig.IsingSlab(X, Y, seed=seed, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, ring_halo=True)
        if not slab.split:
            slab.close()
            return None
        ring = ig.NativeRing(slab, transport="ipc").init()
        for _ in range(6):
            ring.sweep(512)  # (launches of several epochs each: a warm and a timed one per form, then the form that stays)
        ring.quiesce()
        out = (ring.count(), ring.bond_equal(), slab.guard_info())
        ring.close()
        slab.close()
        return out
    off = run("0")
    if off is None:
        pytest.skip("the rule applies on a whole MI355X (eight XCDs)")
    on = run("1")
    assert on[:2] == off[:2]
    g = on[2]
    assert off[2]["form_state"] == 0  # (guard off: nothing timed, the table's form)
    assert g["form_state"] == 3 and g["split_flips_per_ns"] > 0 and g["fused_flips_per_ns"] > 0
    assert (g["split_kept"] == 1) == (not g["fused_flips_per_ns"] > 1.02 * g["split_flips_per_ns"])
    print(f"ring of one {Y} x {X}: split {g['split_flips_per_ns']:.0f}, fused {g['fused_flips_per_ns']:.0f} flips/ns in the timed launches: kept {'split' if g['split_kept'] else 'fused'}")
