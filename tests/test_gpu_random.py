"""Randomised parity sweep: random lattice shapes, temperatures, seeds, strip heights, layouts, sub-lattices, couplings
and row partitions (fixed RNG seed, so the cases are the same on every run) against the CPU oracle, bit for bit."""
import numpy as np
import pytest

import ising_gpu_amd as ig
from test_gpu_parity import _compare

pytestmark = pytest.mark.gpu
# a one-off wider hunt: ISING_TEST_RANDOM_SCALE=10 ISING_TEST_RANDOM_SEED=1 python -m pytest tests/test_gpu_random.py -m gpu
_SCALE = int(__import__("os").environ.get("ISING_TEST_RANDOM_SCALE", "1"))
_SEED = int(__import__("os").environ.get("ISING_TEST_RANDOM_SEED", "0"))


def _cases(n):
    rng = np.random.default_rng(20260928 + _SEED)
    out = []
    for k in range(n):
        layout = [ig.LAYOUT_BALLOT, ig.LAYOUT_DENSE, ig.LAYOUT_NIBBLE][k % 3]
        gx = int(rng.choice([4, 8, 12, 16])) if layout == ig.LAYOUT_BALLOT else int(rng.integers(1, 9))
        X = 2048 * gx
        Y = 16 * int(rng.integers(1, 9))
        temp = float(np.float32(rng.choice([0.4, 1.0, 1.5, 2.0, float(ig.CRIT_TEMP_F32), 2.6, 3.5, 6.0])))
        seed = int(rng.integers(1, 2**40))
        strip = int(rng.choice([0, 1, 2, 4, 8, 16]))
        if Y % max(strip, 1):
            strip = 0
        sl = None
        if rng.random() < 0.35:
            xs = [w for w in (2048, 4096, 8192, 16384) if X % w == 0 and (layout != ig.LAYOUT_BALLOT or w <= 4096 or w % 8192 == 0)]
            ys = [h for h in (16, 32, 48, 64) if Y % h == 0]
            sl = (int(rng.choice(xs)), int(rng.choice(ys)))
        jp = float(rng.choice([0.1, 0.5, 0.9])) if rng.random() < 0.3 else None
        sweeps = int(rng.integers(1, 5))
        out.append(pytest.param(layout, X, Y, temp, seed, strip, sl, jp, sweeps, id=f"{k}-L{layout}-{X}x{Y}-T{temp:.2f}-s{strip}-sl{sl}-J{jp}"))
    return out


@pytest.mark.parametrize("layout,X,Y,temp,seed,strip,sl,jp,sweeps", _cases(36 * _SCALE))
def test_random_configuration(gpu, oracle_mod, layout, X, Y, temp, seed, strip, sl, jp, sweeps):
    kw = dict(XSL=sl[0], YSL=sl[1]) if sl else {}
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp, **kw).init()
    if jp is not None:
        orc.init_couplings(jp)
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, strip_rows=strip, layout=layout, J_prob=jp, **kw) as s:
        s.init()
        if jp is not None:
            s.init_couplings()
        # first sweep as a random partition of the rows into launches, the rest as whole-colour launches
        cuts = sorted(set([0, Y] + [int(c) for c in np.random.default_rng(seed).integers(0, Y + 1, size=3)]))
        for color in (ig.BLACK, ig.WHITE):
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                s.update_color(1, color, lo, hi)
        s.it = 1
        orc.sweep(1)
        _compare(s, orc, "partitioned sweep")
        s.sweep(sweeps)
        orc.sweep(sweeps)
        _compare(s, orc, "after sweeps")
        assert s.count() == orc.count()
        if jp is None:
            assert s.bond_equal() == orc.bond_equal()


def _fused_cases(n):
    rng = np.random.default_rng(777 + _SEED)
    out = []
    for k in range(n):
        X = int(rng.choice([8192, 8192, 16384, 24576, 12288, 32768]))
        Y = int(rng.choice([16, 32, 48, 64, 96, 112, 160, 256, 272]))
        strip = int(rng.choice([0, 1, 2, 4, 8]))
        if Y % max(strip, 1):
            strip = 0
        rng.random()  # (8-wave workgroups, gone; the draw stays so that the other choices keep their values)
        nt = int(rng.random() < 0.4)
        wgs = rng.choice(["", "1", "2", "3", "5", "17", "64"])
        jp = float(rng.choice([0.2, 0.7])) if (rng.random() < 0.3 and X % 8192 == 0) else None
        temp = float(np.float32(rng.choice([1.0, 2.0, float(ig.CRIT_TEMP_F32), 3.0])))
        seed = int(rng.integers(1, 2**40))
        out.append(pytest.param(X, Y, strip, nt, str(wgs), jp, temp, seed, id=f"{k}-{X}x{Y}-s{strip}-nt{nt}-g{wgs or 'auto'}-J{jp}-T{temp:.2f}"))
    return out


@pytest.mark.parametrize("X,Y,strip,nt,wgs,jp,temp,seed", _fused_cases(28 * _SCALE))
def test_random_fused_configuration(gpu, oracle_mod, monkeypatch, X, Y, strip, nt, wgs, jp, temp, seed):
    """The fused launch form under random shapes and switches: streaming instantiation, strip
    heights, grids from one workgroup (every unit waits for its parents, one after the other) to what the chip holds,
    launches of 1 .. 9 sweeps."""
    monkeypatch.setenv("ISING_FUSED", "1")
    monkeypatch.setenv("ISING_FUSED_NT", str(nt))
    if wgs:
        monkeypatch.setenv("ISING_FUSED_WGS", wgs)
    else:
        monkeypatch.delenv("ISING_FUSED_WGS", raising=False)
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp).init()
    if jp is not None:
        orc.init_couplings(jp)
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, strip_rows=strip, layout=ig.LAYOUT_BALLOT, J_prob=jp) as s:
        s.init()
        if jp is not None:
            s.init_couplings()
        assert s.fused
        for n in (1, 9, 2):
            s.sweep(n)
            orc.sweep(n)
            _compare(s, orc, f"after {s.it} sweeps")
        assert s.count() == orc.count()
        if jp is None:
            assert s.bond_equal() == orc.bond_equal()


def _ring_cases(n):
    rng = np.random.default_rng(4711 + _SEED)
    out = []
    for k in range(n):
        nslabs = int(rng.choice([1, 2, 3, 4]))
        X = int(rng.choice([8192, 16384, 24576, 12288]))
        Yk = int(rng.choice([16, 32, 48, 64, 96, 160]))
        ghost = str(rng.choice(["", "4", "8", "16", "64"]))
        rng.choice(["single", "plain"])  # (a switch of an earlier launch policy; the draw stays so that the other choices keep their values)
        rng.random()  # (8-wave workgroups, gone: as above)
        t2 = str(rng.choice(["", "0", "2", "4"]))
        wgs = str(rng.choice(["", "1", "3", "17"]))
        strip = int(rng.choice([0, 1, 2, 4, 8]))
        if Yk % max(strip, 1):
            strip = 0
        inline = str(rng.choice(["0", "1"]))
        temp = float(np.float32(rng.choice([1.0, 2.0, float(ig.CRIT_TEMP_F32), 3.0])))
        seed = int(rng.integers(1, 2**40))
        sweeps = (int(rng.integers(1, 20)), int(rng.integers(1, 40)))
        jp = float(rng.choice([0.2, 0.7])) if (rng.random() < 0.35 and X % 8192 == 0) else None
        out.append(pytest.param(nslabs, X, Yk, ghost, t2, wgs, strip, inline, temp, seed, sweeps, jp,
                                id=f"{k}-{nslabs}x{X}x{Yk}-G{ghost or 'auto'}-t{t2 or 'auto'}-g{wgs or 'auto'}-s{strip}-i{inline}-n{sweeps[0]}+{sweeps[1]}-J{jp}"))
    return out


@pytest.mark.parametrize("nslabs,X,Yk,ghost,t2,wgs,strip,inline,temp,seed,sweeps,jp", _ring_cases(32 * _SCALE))
def test_random_ring_with_ghost_rows(gpu, oracle_mod, monkeypatch, nslabs, X, Yk, ghost, t2, wgs, strip, inline, temp, seed, sweeps, jp):
    """Ring slabs with ghost rows under random shapes and switches: 1 .. 4 slabs of one device (copy transport on the comm
    streams or inline; a ring of one sends to itself), ghost rows 4 .. 64 deep, one- to eight-row units, several ticket
    counters, grids from one workgroup
    up, -J couplings, two sweep calls whose lengths do not line up with the exchange period -- against the oracle's single lattice."""
    monkeypatch.setenv("ISING_RING_STORE", "0")  # (slabs of one device would otherwise store straight into each other's halo rows)
    monkeypatch.setenv("ISING_RING_INLINE", inline)
    for name, val in (("ISING_RING_GHOST", ghost), ("ISING_FUSED_TICKETS2", t2), ("ISING_FUSED_WGS", wgs)):
        if val:
            monkeypatch.setenv(name, val)
        else:
            monkeypatch.delenv(name, raising=False)
    orc = oracle_mod.OracleLattice(X, Yk * nslabs, seed=seed, temp=temp).init()
    if jp is not None:
        orc.init_couplings(jp)
    ring = ig.SlabSet([ig.IsingSlab(X, Yk, seed=seed, temp=temp, nslabs=nslabs, slab=k, layout=ig.LAYOUT_BALLOT, strip_rows=strip, ring_halo=nslabs == 1,
                                    J_prob=jp) for k in range(nslabs)])
    try:
        ring.init()
        if jp is not None:  # (-J, part of SlabSet.init: the ghost rows' couplings are generated in place, global row around the ring)
            assert np.array_equal(np.concatenate([s.read_couplings(ig.BLACK) for s in ring.slabs]), orc.hamB)
            assert np.array_equal(np.concatenate([s.read_couplings(ig.WHITE) for s in ring.slabs]), orc.hamW)
        G = ring.slabs[0].ghost_ptrs(ig.BLACK)[0]
        assert G == (min(int(ghost or 64), Yk // 2) & ~1) and ring.slabs[0].max_sweeps_per_launch == G // 2
        for n in sweeps:
            ring.sweep(n)
            orc.sweep(n)
            assert np.array_equal(np.concatenate([s.read(ig.BLACK) for s in ring.slabs]), orc.black), f"black after {ring.it} sweeps"
            assert np.array_equal(np.concatenate([s.read(ig.WHITE) for s in ring.slabs]), orc.white), f"white after {ring.it} sweeps"
        assert ring.count() == orc.count()
        if jp is None:
            assert ring.bond_equal() == orc.bond_equal()
    finally:
        ring.close()


def _batch_cases(n):
    rng = np.random.default_rng(4242 + _SEED)
    out = []
    for k in range(n):
        X = int(rng.choice([8192, 8192, 10240, 16384, 24576]))
        Y = int(rng.choice([16, 32, 48, 64, 96, 128, 256]))
        nlat = int(rng.integers(1, 7))
        temps = [float(np.float32(t)) for t in rng.choice([1.0, 1.7, 2.0, float(ig.CRIT_TEMP_F32), 2.6, 3.4], size=nlat)]
        seeds = [int(v) for v in rng.integers(1, 2**40, size=nlat)]
        wgs = str(rng.choice(["", "1", "2", "5", "33"]))
        cap = str(rng.choice(["", "2", "7", "40"]))
        nt = int(rng.random() < 0.3)
        calls = [int(v) for v in rng.integers(1, 46, size=int(rng.integers(1, 4)))]
        out.append(pytest.param(X, Y, temps, seeds, wgs, cap, nt, calls, id=f"{k}-{nlat}x{X}x{Y}-g{wgs or 'auto'}-cap{cap or 'auto'}-nt{nt}-{calls}"))
    return out


@pytest.mark.parametrize("X,Y,temps,seeds,wgs,cap,nt,calls", _batch_cases(24 * _SCALE))
def test_random_batch_configuration(gpu, oracle_mod, monkeypatch, X, Y, temps, seeds, wgs, cap, nt, calls):
    """Batched fused launches under random shapes: 1 .. 6 lattices with their own temperatures and seeds, grids from one
    workgroup up, launches cut at 2 / 7 / 40 sweeps or not at all, both lattice-word flavours; every member = its oracle, and
    the one-launch measurement = count + bond sum of every member."""
    for name, val in (("ISING_FUSED_WGS", wgs), ("ISING_FUSED_MAX_SWEEPS", cap)):
        if val:
            monkeypatch.setenv(name, val)
        else:
            monkeypatch.delenv(name, raising=False)
    monkeypatch.setenv("ISING_FUSED_NT", str(nt))
    slabs = [ig.IsingSlab(X, Y, seed=s, temp=t, layout=ig.LAYOUT_BALLOT) for t, s in zip(temps, seeds)]
    orcs = [oracle_mod.OracleLattice(X, Y, seed=s, temp=t).init() for t, s in zip(temps, seeds)]
    with ig.IsingBatch(slabs) as b:
        b.init()
        for n in calls:
            b.sweep(n).measure_enqueue()
            for o in orcs:
                o.sweep(n)
        meas = b.measure_fetch()
        assert len(meas) == len(calls)
        for r, (s, o) in enumerate(zip(slabs, orcs)):
            _compare(s, o, f"member {r}")
            assert meas[-1][r] == (*o.count(), o.bond_equal())
    for s in slabs:
        s.close()


def _subl_fused_cases(n):
    rng = np.random.default_rng(9090 + _SEED)
    out = []
    for k in range(n):
        X = int(rng.choice([8192, 16384, 32768]))
        XSL = int(rng.choice([w for w in (2048, 4096, 8192, 16384, 32768) if X % w == 0]))
        YSL = int(rng.choice([16, 32, 48, 64, 128]))
        Y = YSL * int(rng.integers(1, 5))
        strip = int(rng.choice([h for h in (0, 1, 2, 4, 8, 16) if h == 0 or YSL % h == 0]))
        wgs = str(rng.choice(["", "1", "3", "9"]))
        nt = int(rng.random() < 0.4)
        temp = float(np.float32(rng.choice([1.2, 2.0, float(ig.CRIT_TEMP_F32), 3.0])))
        seed = int(rng.integers(1, 2**40))
        calls = [int(v) for v in rng.integers(1, 12, size=2)]
        out.append(pytest.param(X, Y, XSL, YSL, strip, wgs, nt, temp, seed, calls, id=f"{k}-{X}x{Y}-sl{XSL}x{YSL}-s{strip}-g{wgs or 'auto'}-nt{nt}"))
    return out


@pytest.mark.parametrize("X,Y,XSL,YSL,strip,wgs,nt,temp,seed,calls", _subl_fused_cases(24 * _SCALE))
def test_random_sublattices_in_fused_launches(gpu, oracle_mod, monkeypatch, X, Y, XSL, YSL, strip, wgs, nt, temp, seed, calls):
    monkeypatch.setenv("ISING_FUSED", "1")
    monkeypatch.setenv("ISING_FUSED_NT", str(nt))
    if wgs:
        monkeypatch.setenv("ISING_FUSED_WGS", wgs)
    else:
        monkeypatch.delenv("ISING_FUSED_WGS", raising=False)
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp, XSL=XSL, YSL=YSL).init()
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, strip_rows=strip, layout=ig.LAYOUT_BALLOT, XSL=XSL, YSL=YSL) as s:
        assert s.fused
        s.init()
        for n in calls:
            s.sweep(n)
            orc.sweep(n)
            _compare(s, orc, f"after {s.it} sweeps")
            assert s.count() == orc.count() and s.bond_equal() == orc.bond_equal()
