"""The fused launch form of the ballot kernel (ising_sweep: several colour half-sweeps per launch, in-order tickets,
per-strip completion counters, write-through hand-over between workgroups -- csrc/ising_ballot.hip).  By default only
lattices from 2^26 spins up take it (tests/test_gpu_fullsize.py, the 65536^2 README runs); here it is forced on smaller
ones (ISING_FUSED=1 is read when a slab is created) so that every seam of it is compared with the oracle word for word:
strip order from both ends, periodic wrap through the mirror rows, launches of 1 .. 32 sweeps and the cuts between them,
one wave column (X = 8192, four strips per workgroup) and several, partly empty workgroups, -J, temperature changes
between launches, and the 8-wave workgroup variant."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def fused(monkeypatch):
    monkeypatch.setenv("ISING_FUSED", "1")


def _same(s, orc):
    return np.array_equal(s.read(ig.BLACK), orc.black) and np.array_equal(s.read(ig.WHITE), orc.white)


@pytest.mark.parametrize("X,Y,strip", [(8192, 16, 0), (8192, 48, 8), (8192, 272, 1), (16384, 160, 4), (24576, 96, 2), (32768, 1024, 8)])
def test_fused_matches_oracle_state(gpu, oracle_mod, fused, X, Y, strip):
    seed, temp = 2025, ig.CRIT_TEMP_F32
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp).init()
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, layout=ig.LAYOUT_BALLOT, strip_rows=strip) as s:
        assert s.fused
        s.init()
        for n in (1, 2, 3, 1, 7):  # launches of 2, 4, 6, 2, 14 levels; the iteration counter runs on across them
            s.sweep(n)
            orc.sweep(n)
            assert _same(s, orc), (X, Y, s.it)
            assert s.count() == orc.count() and s.bond_equal() == orc.bond_equal()


@pytest.mark.parametrize("J", [0.0, 0.3])
def test_fused_streaming_variant_matches_oracle(gpu, oracle_mod, fused, monkeypatch, J):
    """Lattices larger than the memory-side cache mark their lattice words non-temporal (a kernel instantiation of its
    own, ISING_FUSED_NT forces it): same results."""
    monkeypatch.setenv("ISING_FUSED_NT", "1")
    X, Y, seed = 16384, 160, 77
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=ig.CRIT_TEMP_F32).init()
    with ig.IsingSlab(X, Y, seed=seed, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, strip_rows=4, J_prob=J) as s:
        s.init()
        if J:
            orc.init_couplings(J)
            s.init_couplings()
        assert s.fused
        for n in (1, 3, 5):
            s.sweep(n)
            orc.sweep(n)
            assert _same(s, orc) and s.count() == orc.count() and s.bond_equal() == orc.bond_equal()


def test_fused_batches_of_32_sweeps_and_the_cut_between_them(gpu, oracle_mod, fused):
    X, Y, seed = 8192, 64, 11
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=2.0).init().sweep(70)
    with ig.IsingSlab(X, Y, seed=seed, temp=2.0, layout=ig.LAYOUT_BALLOT) as s:
        s.init().sweep(70)  # 32 + 32 + 6 sweeps = three launches
        assert _same(s, orc) and s.count() == orc.count()


def test_fused_with_couplings_and_temperature_changes(gpu, oracle_mod, fused):
    X, Y, seed = 16384, 64, 5
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=1.5).init().init_couplings(0.3)
    with ig.IsingSlab(X, Y, seed=seed, temp=1.5, layout=ig.LAYOUT_BALLOT, J_prob=0.3) as s:
        s.init().init_couplings()
        assert s.fused
        for t in (1.5, 2.5, 0.9):
            s.set_temperature(t)
            orc.temp = float(np.float32(t))
            s.sweep(3)
            orc.sweep(3)
            assert _same(s, orc), t
    # a temperature without integer thresholds ends the fused form (and the ballot layout) on the spot
    with ig.IsingSlab(X, Y, seed=seed, temp=1.5, layout=ig.LAYOUT_BALLOT) as s:
        s.init().sweep(2)
        s.set_temperature(-1.0)
        assert not s.fused
        s.sweep(2)
        orc2 = oracle_mod.OracleLattice(X, Y, seed=seed, temp=1.5).init().sweep(2)
        orc2.temp = -1.0
        orc2.sweep(2)
        assert s.current_layout() == ig.LAYOUT_DENSE and _same(s, orc2)


def test_fused_equals_plain_at_16384_full_size(gpu, monkeypatch):
    """BASELINE config 2's lattice on both launch forms: identical packed state after 40 sweeps."""
    got = {}
    for form in ("0", "1"):
        monkeypatch.setenv("ISING_FUSED", form)
        with ig.IsingSlab(16384, 16384, seed=1234, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT) as s:
            assert s.fused == (form == "1")
            s.init().sweep(40)
            got[form] = (s.count(), s.bond_equal(), s.read_bits(ig.BLACK).tobytes(), s.read_bits(ig.WHITE).tobytes())
    assert got["0"] == got["1"]


def test_bench_line_on_a_small_lattice(gpu, oracle_mod):
    """bench.py end to end (the driver's command shape) on a lattice that takes a second: one JSON line with the contract's
    keys, the roofline and cpu_baseline objects, and counts equal to the oracle's."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--x", "8192", "--y", "8192",
                        "--preheat-ms", "5", "--cpu-threads", "8"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in b, key
    assert b["n_gpus"] == 1 and b["steps"] == 4 and b["warmup"] == 2 and b["unit"] == "flips/ns" and b["value"] > 100
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm_reference_accounting", "hbm_real"):
        assert key in b["roofline"], key
    # roofline.frac is the contract's number: 1.5 B per flip x flips per launch / average launch duration / 8 TB/s; the roof that
    # binds in fact is the vector ALU (`bound`), and the ratio to the draw-only ceiling of the same job rides along as frac_valu_ceiling
    rf = b["roofline"]
    assert rf["bound"] == "valu" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["frac"] == rf["frac_hbm_1p5B"]
    alg = 1.5 * 8192 * 8192 / 2.0 * rf["half_sweeps_per_launch"]
    assert abs(rf["achieved"] - alg / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 0.01 * rf["achieved"]  # reproducible from the launch duration alone
    assert 0 < rf["frac_valu_ceiling"] < 1.2 and abs(rf["frac_valu_ceiling"] - rf["kernel_sites_ns"] / rf["valu_ceiling_sites_ns"]) < 1e-3
    ref = rf["hbm_reference_accounting"]
    assert ref["peak"] == 8000.0 and abs(ref["frac"] - ref["achieved"] / ref["peak"]) < 1e-3 and ref["frac"] == rf["frac"]
    assert abs(ref["achieved"] / 1.5 - rf["kernel_sites_ns"]) < 0.01 * rf["kernel_sites_ns"]  # the same launch time in both
    small = b["default_lattice_2048"]
    assert small["frac"] == small["frac_hbm_1p5B"] and 0 < small["frac_of_plateau"] < 2
    many = b["batched_31x2048"]  # (round 6: 31 of them at 31 temperatures in one launch per pass)
    assert many["member0_counts_equal_lone_lattice"] is True and many["lattices"] == 31 and many["value"] > 100 and many["frac_of_plateau"] > 0
    # the reference's methodology: the same sweeps with the counts read back every 16 inside the timed region
    leg = b["with_counts_every_16"]
    assert leg["final_counts_equal_first_leg"] is True and leg["counts_in_timed_region"] == 1 and leg["value"] > 100
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in b["cpu_baseline"], key
    orc = oracle_mod.OracleLattice(8192, 8192, seed=1234, temp=oracle_mod.CRIT_TEMP).init().sweep(6)
    assert (b["config"]["up"], b["config"]["down"]) == orc.count()


@pytest.mark.parametrize("ring,port,exchange", [("native", 29551, "ipc-native"), ("native-rccl", 29553, "rccl-native"), ("torch", 29552, "p2p-ghost64")])
def test_bench_ring_code_path_with_one_rank(gpu, oracle_mod, ring, port, exchange):
    """bench.py's N > 1 code path as far as one GPU can run it: under torch.distributed.run with ONE rank and --force-ring the
    slab is a ring of one -- torch.distributed (nccl) is initialised, and either the library's own ring runs -- the peer transport (primary from round 6 on:
    the rank maps its own rows through hipIpcMemHandle and pushes its edge rows into them), or with --transport rccl its RCCL communicator
    with the id torch broadcast, the ghost rows through ncclSend/ncclRecv on the comm stream --, or the
    fallback behind it: the same deep schedule with torch.distributed send/recv on the library-owned rows."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port",
                        str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-ring", "--ring", ring.split("-")[0], "--steps", "4", "--warmup", "2",
                        "--x", "8192", "--y", "8192", "--preheat-ms", "5", "--layout", "ballot"] + (["--transport", "rccl"] if ring == "native-rccl" else []),
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    b = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert b["config"]["exchange"] == exchange and b["config"]["nranks"] == 1
    if ring == "native":  # (auto: both of the library's transports come up on a ring of one -- the peer transport is held against RCCL over 40 sweeps before it is kept)
        notes = " ".join(b["transport_attempts"]["rank0"])
        assert "ipc and rccl agree after 40 sweeps" in notes, notes
    if ring.startswith("native"):  # the library's ring says where the time around its exchanges went (one launch + exchange per 32 sweeps)
        xs = b["exchange_stats"]
        assert xs["exchanges_per_rank"] >= 1 and xs["launch_ms"]["mean"] > 0 and xs["exchange_ms"]["max"] >= xs["exchange_ms"]["mean"] > 0
        assert len(xs["go_after_end_ms"]["by_rank_mean"]) == 1
    else:
        assert "exchange_stats" not in b
    orc = oracle_mod.OracleLattice(8192, 8192, seed=1234, temp=oracle_mod.CRIT_TEMP).init().sweep(6)
    assert (b["config"]["up"], b["config"]["down"]) == orc.count() and b["config"]["rank_up"] == [orc.count()[0]]


def test_a_transport_that_disagrees_is_not_kept(gpu, oracle_mod):
    """open_native_ring's cross-check: with the first transport's counts made to differ (test aid), the ring that comes back is the second transport's."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ISING_TEST_RING_CROSSCHECK_PERTURB="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29554",
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-ring", "--steps", "4", "--warmup", "2", "--x", "8192", "--y", "8192", "--preheat-ms", "5",
                        "--layout", "ballot", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    b = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert b["config"]["exchange"] == "rccl-native"
    assert "DISAGREE after 40 sweeps: rccl kept" in " ".join(b["transport_attempts"]["rank0"])
    orc = oracle_mod.OracleLattice(8192, 8192, seed=1234, temp=oracle_mod.CRIT_TEMP).init().sweep(6)
    assert (b["config"]["up"], b["config"]["down"]) == orc.count()


def test_fused_launch_gives_up_instead_of_hanging(gpu, oracle_mod, monkeypatch):
    """Completion counters that can never arrive (the host's bases out of step with the device, as after a faulted launch):
    the units' polls are bounded, the launch ends, the next synchronise reports ISING_E_STATE within a second -- not the
    watchdog --, tickets and counters start over and the slab sweeps correctly again."""
    import time
    from ising_gpu_amd import _lib
    monkeypatch.setenv("ISING_FUSED", "1")
    X, Y, seed = 8192, 2048, 31
    with ig.IsingSlab(X, Y, seed=seed, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT) as s:
        assert s.fused
        s.init().sweep(3)
        s.synchronize()
        s.debug_fault(1, 1 << 13)  # level 1 waits for counts 2^20 too high; give up after 8192 polls (a few tens of ms)
        t0 = time.perf_counter()
        s.sweep(4)
        with pytest.raises(ig.IsingError) as e:
            s.synchronize()
        assert e.value.code == _lib.E_STATE and "gave up" in str(e.value)
        assert time.perf_counter() - t0 < 1.0
        s.synchronize()  # the error was consumed, the context is usable
        orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=oracle_mod.CRIT_TEMP).init().sweep(5)
        s.init().sweep(5)
        assert np.array_equal(s.read(ig.BLACK), orc.black) and np.array_equal(s.read(ig.WHITE), orc.white)
        assert s.count() == orc.count()
    # the same through a blocking observable, on a batch
    slabs = [ig.IsingSlab(X, 256, seed=k, temp=2.0, layout=ig.LAYOUT_BALLOT) for k in (1, 2)]
    with ig.IsingBatch(slabs) as b:
        b.init().sweep(2)
        slabs[0].synchronize()
        slabs[0].debug_fault(1, 1 << 13)
        b.sweep(2)  # (the batch has counters of its own: nothing wrong with them, this launch completes)
        b.measure_enqueue()
        assert len(b.measure_fetch()) == 1
        slabs[0].it = 4
        slabs[0].sweep(2)  # member 0 alone, on ITS counters: gives up
        with pytest.raises(ig.IsingError):
            slabs[0].count()
        slabs[0].init().sweep(1)
        assert slabs[0].count() == oracle_mod.OracleLattice(X, 256, seed=1, temp=2.0).init().sweep(1).count()
    for s in slabs:
        s.close()


def test_batched_launch_that_gives_up_is_reported_once_and_by_any_batch_call(gpu, oracle_mod):
    """a batched launch raises the BATCH's own abort word.  A call on a member in between (which checks and clears the
    member's word and knows nothing of the batch's tickets) must not swallow it, the next batch call -- a sweep, not only a fetch --
    reports ISING_E_STATE within a second, and the batch then runs correctly again without waiting for stale counters."""
    import time
    from ising_gpu_amd import _lib
    X, Y = 8192, 256
    slabs = [ig.IsingSlab(X, Y, seed=k, temp=2.0, layout=ig.LAYOUT_BALLOT) for k in (5, 6, 7)]
    with ig.IsingBatch(slabs) as b:
        b.init().sweep(2)
        slabs[0].synchronize()
        b.debug_fault(1 << 13)
        t0 = time.perf_counter()
        b.sweep(2)                      # gives up after 8192 polls
        slabs[0].synchronize()          # a member call: waits for the launch, finds nothing wrong with the MEMBER
        assert slabs[1].count()[0] > 0  # (a blocking member observable as well)
        with pytest.raises(ig.IsingError) as e:
            b.sweep(1)                  # the next batch call of any kind reports it
        assert e.value.code == _lib.E_STATE and "gave up" in str(e.value)
        assert time.perf_counter() - t0 < 2.0
        t0 = time.perf_counter()
        b.init().sweep(3).measure_enqueue()
        meas = b.measure_fetch()
        assert time.perf_counter() - t0 < 2.0  # (no launch waits ~10 s for counters that are out of step)
        for r, k in enumerate((5, 6, 7)):
            o = oracle_mod.OracleLattice(X, Y, seed=k, temp=2.0).init().sweep(3)
            assert meas[0][r] == (*o.count(), o.bond_equal())
            assert np.array_equal(slabs[r].read(ig.BLACK), o.black)
    for s in slabs:
        s.close()


def test_contexts_driven_from_threads_of_their_own(gpu, oracle_mod, fused):
    """One context per thread, each on a private stream (ctypes drops the GIL, so the C-ABI calls really overlap): the fused
    launches of four lattices share the chip -- a workgroup only draws a ticket once it runs, so the launches cannot starve each
    other's parents --, the launcher's per-device caches are asked for the first time from four threads at once, and every
    thread's errors stay its own (ising_last_error is per thread)."""
    import threading
    shapes = [(8192, 512, 11, 2.0), (16384, 256, 12, 2.269), (8192, 1024, 13, 2.5), (24576, 128, 14, 1.8)]
    want = [oracle_mod.OracleLattice(X, Y, seed=sd, temp=t).init().sweep(9) for X, Y, sd, t in shapes]
    out, errs = [None] * len(shapes), []
    gate = threading.Barrier(len(shapes))

    def run(k):
        try:
            X, Y, sd, t = shapes[k]
            gate.wait()
            with ig.IsingSlab(X, Y, seed=sd, temp=t, layout=ig.LAYOUT_BALLOT).use_private_stream() as s:
                assert s.fused
                s.init().sweep(4).sweep(5)
                with pytest.raises(ig.IsingError) as e:  # an error of this thread's own ...
                    s.update_color(0, 7, 0, Y)
                assert "colour" in str(e.value)
                out[k] = (s.read(ig.BLACK), s.read(ig.WHITE), s.count(), s.bond_equal())
        except BaseException as ex:  # noqa: BLE001
            errs.append((k, repr(ex)))

    th = [threading.Thread(target=run, args=(k,)) for k in range(len(shapes))]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not errs and not any(t.is_alive() for t in th), errs
    for k, o in enumerate(want):
        assert np.array_equal(out[k][0], o.black) and np.array_equal(out[k][1], o.white)
        assert out[k][2] == o.count() and out[k][3] == o.bond_equal()


@pytest.mark.parametrize("transport", [None, "rccl", "ipc"])
def test_aged_counters_change_nothing(gpu, oracle_mod, fused, transport):
    """The fused launches' completion counters are monotone and start over past 2^30; the overlapped exchange's counters (edge
    units done, exchange epochs) are compared by signed difference and wrap at 2^32.  ising_debug_fault(2) ages them all as
    billions of sweeps would (device and host record together): a lone slab and a ring of one (both transports) go on bit for
    bit."""
    X, Y, seed = 16384, 512, 515
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=oracle_mod.CRIT_TEMP).init()
    with ig.IsingSlab(X, Y, seed=seed, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, ring_halo=transport is not None) as s:
        ring = ig.NativeRing(s, transport=transport).init() if transport else None
        drv = ring if ring else s.init()
        for n, age in ((33, True), (70, True), (5, False)):
            drv.sweep(n)
            orc.sweep(n)
            if age:
                if ring:
                    ring.quiesce()
                s.debug_fault(2, 0)
            assert drv.count() == orc.count() and drv.bond_equal() == orc.bond_equal()
        if ring:
            ring.quiesce()
        assert np.array_equal(s.read(ig.BLACK), orc.black) and np.array_equal(s.read(ig.WHITE), orc.white)
        if ring:
            ring.close()


@pytest.mark.parametrize("nt", ["0", "1"])
@pytest.mark.parametrize("X,Y,XSL,YSL,strip", [(16384, 256, 2048, 16, 0), (16384, 256, 4096, 64, 8), (16384, 256, 8192, 128, 0), (16384, 256, 16384, 256, 16),
                                               (8192, 128, 2048, 32, 2), (32768, 64, 16384, 16, 1), (32768, 96, 32768, 48, 0)])
def test_fused_launches_carry_sublattices(gpu, oracle_mod, fused, monkeypatch, X, Y, XSL, YSL, strip, nt):
    """Sub-lattices (--xsl / --ysl, optimized/main.cu:1423-1462) in fused launches: a strip's parents wrap inside its block's
    strips, the row above a block's first row is the block's last.  Widths of 2048 / 4096 (periods inside a wave), 8192 and
    more; strips as tall as a block and shorter; both lattice-word flavours."""
    monkeypatch.setenv("ISING_FUSED_NT", nt)
    seed, temp = 414, ig.CRIT_TEMP_F32
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp, XSL=XSL, YSL=YSL).init()
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, layout=ig.LAYOUT_BALLOT, strip_rows=strip, XSL=XSL, YSL=YSL) as s:
        assert s.fused and YSL % s.strip_rows == 0
        s.init()
        for n in (1, 2, 9, 33):
            s.sweep(n)
            orc.sweep(n)
            assert _same(s, orc), (X, Y, XSL, YSL, s.it)
            assert s.count() == orc.count() and s.bond_equal() == orc.bond_equal()


def test_ring_slabs_of_sublattices_sweep_alone_in_fused_launches(gpu, oracle_mod, fused):
    """Nothing crosses slabs when there are sub-lattices: the ring sweeps every slab on its own, in fused launches."""
    X, Y, n, XSL, YSL, seed = 16384, 128, 2, 4096, 64, 5
    orc = oracle_mod.OracleLattice(X, Y * n, seed=seed, temp=1.5, XSL=XSL, YSL=YSL).init().sweep(7)
    slabs = [ig.IsingSlab(X, Y, seed=seed, temp=1.5, nslabs=n, slab=k, layout=ig.LAYOUT_BALLOT, XSL=XSL, YSL=YSL) for k in range(n)]
    assert all(s.fused for s in slabs)
    ring = ig.SlabSet(slabs).init()
    ring.sweep(3).sweep(4)
    assert ring.count() == orc.count()
    for k, s in enumerate(slabs):
        assert np.array_equal(s.read(ig.BLACK), orc.black[k * Y:(k + 1) * Y]) and np.array_equal(s.read(ig.WHITE), orc.white[k * Y:(k + 1) * Y])
    ring.close()


@pytest.mark.parametrize("cap", [None, "5", "64"])
def test_fused_launches_longer_than_32_sweeps(gpu, oracle_mod, fused, monkeypatch, cap):
    """Small lattices carry many sweeps per launch (~50 ms worth, up to 4096: a launch costs ~60 us whatever it carries);
    ISING_FUSED_MAX_SWEEPS caps it.  150 sweeps as one launch, as 30 and as 3 launches: the same spins."""
    if cap:
        monkeypatch.setenv("ISING_FUSED_MAX_SWEEPS", cap)
    X, Y, seed = 8192, 128, 606
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=ig.CRIT_TEMP_F32).init().sweep(150)
    with ig.IsingSlab(X, Y, seed=seed, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT) as s:
        assert s.fused and s.max_sweeps_per_launch == (int(cap) if cap else 4096)
        s.init().sweep(150)
        assert _same(s, orc) and s.count() == orc.count()


@pytest.mark.parametrize("layout", [ig.LAYOUT_BALLOT, ig.LAYOUT_DENSE], ids=["ballot", "dense"])
def test_sublattices_do_not_see_each_other(gpu, layout):
    """A property that needs no oracle, at 2^26 spins: with --xsl/--ysl every block is a periodic system of its own (optimized/main.cu:
    1423-1462).  Complementing ONE block of the start leaves every other block's trajectory untouched, and that block's trajectory is
    the complement of what it was (the plain model's spin-flip symmetry; the random numbers do not depend on the state)."""
    X, Y, XSL, YSL, n = 16384, 4096, 4096, 1024, 12
    by, bx = 2, 1                                   # the block that starts complemented
    rows = slice(by * YSL, (by + 1) * YSL)
    words = slice(bx * XSL // 32, (bx + 1) * XSL // 32)  # 16 sites of a colour per 64-bit word, XSL/2 sites of each colour per block row
    ones = np.uint64(0x1111111111111111)
    with ig.IsingSlab(X, Y, seed=77, temp=2.2, layout=layout, XSL=XSL, YSL=YSL) as s:
        s.init()
        start = [s.read(c) for c in (ig.BLACK, ig.WHITE)]
        s.sweep(n)
        ref = [s.read(c) for c in (ig.BLACK, ig.WHITE)]
        s.init()
        for c in (ig.BLACK, ig.WHITE):
            a = start[c].copy()
            a[rows, words] ^= ones
            s.write(c, a)
        s.sweep(n)
        for c in (ig.BLACK, ig.WHITE):
            want = ref[c].copy()
            want[rows, words] ^= ones
            assert np.array_equal(s.read(c), want), c


@pytest.mark.parametrize("X,Y,H,every,calls", [(8192, 64, 0, 1, (3, 5)), (8192, 128, 4, 3, (7, 2, 70)), (16384, 96, 0, 16, (40, 100)), (24576, 48, 16, 5, (11, 64, 66)),
                                                (10240, 64, 2, 4, (9, 130))])
@pytest.mark.parametrize("energy", [False, True], ids=["counts", "counts+energy"])
def test_counts_taken_inside_the_fused_launches(gpu, oracle_mod, monkeypatch, X, Y, H, every, calls, energy):
    """ising_sweep_counted: the reference's print points (countSpins whenever the iteration is a multiple of -p, optimized/main.cu:1806-1810)
    taken INSIDE the fused launches by the units that store the words.  Every count equals the oracle's at that iteration -- calls that end
    between two print points, launches of 64 sweeps and more than one launch per call, a partly dead wave column --, and the state
    afterwards is the oracle's: counting changes nothing.  `energy` (round 5): the bond sum of ising_bond_equal at the same points, taken by the white
    level of every measured sweep -- north_star's energy series next to the magnetisation's."""
    monkeypatch.setenv("ISING_FUSED", "1")
    orc = oracle_mod.OracleLattice(X, Y, seed=77, temp=oracle_mod.CRIT_TEMP).init()
    with ig.IsingSlab(X, Y, seed=77, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, strip_rows=H) as s:
        assert s.fused
        s.init()
        for n in calls:
            got = s.sweep_counted(n, every, energy)
            want = []
            for _ in range(n):
                orc.sweep(1)
                if orc.it % every == 0:
                    want.append(orc.count() + ((orc.bond_equal(),) if energy else ()))
            assert got == want, (n, s.it)
        assert _same(s, orc) and s.count() == orc.count()


@pytest.mark.parametrize("kw", [dict(layout=ig.LAYOUT_DENSE), dict(layout=ig.LAYOUT_NIBBLE), dict(layout=ig.LAYOUT_BALLOT, XSL=4096, YSL=32), dict(layout=ig.LAYOUT_BALLOT, J_prob=0.3)],
                         ids=["dense", "nibble", "sub-lattices", "couplings"])
def test_counted_sweeps_where_the_counts_cannot_ride_in_the_launch(gpu, oracle_mod, kw):
    """The same call on the other layouts, with sub-lattices and with couplings sweeps and counts in turn: the same numbers."""
    X, Y = 8192, 64
    okw = {k: v for k, v in kw.items() if k in ("XSL", "YSL")}
    orc = oracle_mod.OracleLattice(X, Y, seed=5, temp=2.0, **okw).init()
    with ig.IsingSlab(X, Y, seed=5, temp=2.0, **kw) as s:
        s.init()
        if "J_prob" in kw:
            s.init_couplings()
            orc.init_couplings(kw["J_prob"])
        energy = "J_prob" not in kw  # (ising_bond_equal knows no couplings)
        got = s.sweep_counted(10, 4, energy) + s.sweep_counted(7, 4, energy)
        want = []
        for _ in range(17):
            orc.sweep(1)
            if orc.it % 4 == 0:
                want.append(orc.count() + ((orc.bond_equal(),) if energy else ()))
        assert got == want and len(want) == 4
        assert _same(s, orc)


@pytest.mark.parametrize("n,port", [(2, 29561), (8, 29562)])
def test_the_drivers_multi_gpu_command_verbatim(gpu, n, port):
    """the command the driver runs for SCALE_rNN.json -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps 20 --warmup 5`, nothing added -- on this box, where the N ranks share the one GPU: processes, rendezvous, the peer
    transport, the ring schedule and the parity check against the committed goldens all execute; what the line reports about the ranks and their links is there
    to be read on the first real node (optimized/main.cu:1496-1537, :1763-1805 is what it replaces)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "20", "--warmup", "5"], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == n and b["steps"] == 20 and b["warmup"] == 5 and b["unit"] == "flips/ns" and b["value"] > 100
    cfg = b["config"]
    assert cfg["nranks"] == n and cfg["parity_checked"] is True and ("inside the launch" in cfg["exchange_overlap"] or "tail" in cfg["exchange_overlap"])
    assert len(b["ranks"]) == n and all(rk["pci_bus_id"] and rk["lone_slab_flips_per_ns"] > 0 for rk in b["ranks"])
    assert sorted(b["transport_attempts"]) == sorted(f"rank{k}" for k in range(n)) and all(any(a.endswith(": ok") for a in v) for v in b["transport_attempts"].values())
    assert b["expected"]["lone_slab_flips_per_ns_sum"] > 0 and 0 < b["efficiency_vs_lone_slab"] < 1.5
    assert b["distinct_devices"] >= 1 and len(cfg["rank_up"]) == n and sum(cfg["rank_up"]) == cfg["up"]
