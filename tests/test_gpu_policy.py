"""The launch-shape policy of ising_create (csrc/ising_capi.cpp: fused_shape, fused_wgs_for, the split form's rule) defended by measurement: on the shapes
where round 4's probes found cliffs (8192 x 1536, 16384 x 2176, 65536 x 1024, 131072 x 2048), BASELINE config 2 and the slabs a strong-scaling split of
65536^2 produces, the library's choice must not be more than 3 % behind the best of its four neighbours -- half and twice the strip height, one workgroup
per CU fewer and more -- and the other form of launch (fused / split), measured right here, on this box.  A driver, clock or firmware change that moves a cliff under the table's
feet fails this test instead of silently costing a third of the rate (8192 x 1536 ran 1355 against 1699 flips/ns one grid step apart).

Not a parity test: results are compared across shapes only as a sanity check (every cell's counts after the same sweeps are equal)."""
import os

import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu
TC = ig.CRIT_TEMP_F32
TOL = 0.03

SHAPES = [  # (X, Y): lattice columns, rows
    (8192, 1536), (16384, 2176), (65536, 1024), (131072, 2048),      # the cliffs of round 4's wide / small probes
    (8192, 8192), (16384, 16384),                                     # 2^26 spins; BASELINE config 2
    (65536, 32768), (65536, 16384), (65536, 8192),                    # north_star's 65536^2 over 2 / 4 / 8 GPUs: a rank's rows as a lone slab
    (24576, 24576), (8192, 4096), (32768, 4096), (24576, 4096),
]


def _rate(X, Y, H=0, wgs=0, monkeypatch=None, check=None, split=None):
    """flips/ns of ising_sweep on a lone X x Y lattice at strip height H and wgs workgroups per CU (0: the library's choice), in the form of launch the library
    picks (split = None) or the one asked for (0: fused, 1: split); best of 3 pieces of ~25 ms"""
    monkeypatch.setenv("ISING_GUARD", "0")  # (the table's choice against its neighbours AS ASKED FOR: the run-time guard would move a neighbour off its cliff)
    for k, v in (("ISING_FUSED_WGS", 256 * wgs if wgs else None), ("ISING_SPLIT", split)):
        if v is None:
            monkeypatch.delenv(k, raising=False)
        else:
            monkeypatch.setenv(k, str(v))
    sweeps = max(32, min(2048, int(25e-3 * 3.2e12 / (X * Y)) // 32 * 32))  # (3.2 flips/ns = 3.2e12 flips/s)
    with ig.IsingSlab(X, Y, seed=1234, temp=TC, strip_rows=H) as s:
        shape = s.launch_shape()[:2] + (0, s.split, s.fused, s.layout)
        s.init().sweep(32)
        got = s.count()
        if check is not None:
            assert got == check, (X, Y, H, wgs, shape)
        s.sweep_timed(sweeps)
        best = max(X * Y * sweeps / (s.sweep_timed(sweeps) * 1e6) for _ in range(3))
    return best, shape, got


@pytest.fixture(scope="module")
def warm_clock(gpu):
    """the shader clock ramps over the first ~50 ms of load: nobody's first measurement should pay for it"""
    with ig.IsingSlab(16384, 16384, seed=1, temp=TC) as s:
        s.init()
        for _ in range(4):
            s.sweep_timed(64)
    return True


@pytest.mark.parametrize("X,Y", SHAPES)
def test_library_choice_within_3_percent_of_its_neighbours(gpu, warm_clock, monkeypatch, X, Y):
    monkeypatch.setenv("ISING_ABORT_POLLS", "40000")  # (a neighbour that lands beyond a cliff must cost milliseconds, not the watchdog's seconds)
    mine, shape, counts = _rate(X, Y, monkeypatch=monkeypatch)
    H, wg = shape[0], shape[1]
    if not shape[4]:
        pytest.skip(f"{Y} x {X}: no fused launches here (layout {shape[5]}): nothing of the launch-shape tables applies")
    cells = {}
    for h, w in ((H // 2, wg), (2 * H, wg), (H, wg - 1), (H, wg + 1)):
        if h < 1 or h > 16 or Y % h or w < 1 or w > 6:
            continue
        try:
            r, shp, _ = _rate(X, Y, h, w, monkeypatch, check=counts)
        except ig.IsingError:
            continue  # (a shape the library refuses, or one that gave up beyond a cliff)
        if shp[3] != shape[3]:
            continue  # (the other form of launch: not a neighbour)
        cells[(h, w, None)] = r
    # ... and the other form of launch, at the shape the library gives it (split launches apply to lattices the memory-side cache holds)
    other = 0 if shape[3] else 1
    try:
        r, shp, _ = _rate(X, Y, 0, 0, monkeypatch, check=counts, split=other)
        if bool(shp[3]) == bool(other):
            cells[(shp[0], shp[1], other)] = r
    except ig.IsingError:
        pass
    # (the other FORM of launch is held to 5 %: where the rule says "fused" the split form at the fused form's strips is no shape the library would ever pick -- it
    # wants taller strips --, and round 6's boxes put it within 3-4 % either way at 65536 x 1024)
    def handicap(key, r):
        return r * (1.0 - 0.05) / (1.0 - TOL) if key[2] is not None else r
    best = max((handicap(k, r) for k, r in cells.items()), default=0.0)
    for _ in range(3):  # again, both sides (up to three times): a single slow -- or lucky -- piece must not fail the suite
        if mine >= (1.0 - TOL) * best:
            break
        mine = max(mine, _rate(X, Y, monkeypatch=monkeypatch)[0])
        hb, wb, sb = max(cells, key=lambda k: handicap(k, cells[k]))
        cells[(hb, wb, sb)] = _rate(X, Y, 0 if sb is not None else hb, 0 if sb is not None else wb, monkeypatch, split=sb)[0]
        best = max(handicap(k, r) for k, r in cells.items())
    print(f"{Y} x {X}: library H={H} wgs={wg} split={int(shape[3])} {mine:.0f} flips/ns; neighbours "
          + ", ".join(f"H={h} wgs={w}{'' if sp is None else (' split' if sp else ' fused')}: {r:.0f}" for (h, w, sp), r in cells.items()))
    assert mine >= (1.0 - TOL) * best, (f"{Y} x {X}: the library's H={H}, {wg} per CU ({'split' if shape[3] else 'fused'}) runs {mine:.0f} flips/ns, "
                                        f"a neighbour {best:.0f}: {cells}")


QUAD_SHAPES = [(2048, 2048), (4096, 4096), (2048, 16384), (6144, 6144), (8192, 1024), (8192, 2048), (4096, 16384), (10240, 1024), (12288, 768), (6144, 1024), (16384, 512)]


@pytest.mark.parametrize("X,Y", QUAD_SHAPES)
def test_quad_rule_within_3_percent_of_the_other_path(gpu, warm_clock, monkeypatch, X, Y):
    """Round 5's rule for the quad path (ising_capi.cpp: quad_pick -- one and two blocks of 2048 columns up to 2^26 spins, three up to 6144 rows, four up to 1024) against the
    library WITHOUT it / WITH it where the rule says no, measured here: the choice must not be more than 3 % behind."""
    def rate(quad):
        if quad is None:
            monkeypatch.delenv("ISING_QUAD", raising=False)
        else:
            monkeypatch.setenv("ISING_QUAD", str(quad))
        sweeps = max(64, min(4096, int(25e-3 * 2.5e12 / (X * Y)) // 64 * 64))
        with ig.IsingSlab(X, Y, seed=1234, temp=TC) as s:
            is_quad = s.quad
            s.init().sweep(64)
            counts = s.count()
            s.sweep_timed(sweeps)
            return max(X * Y * sweeps / (s.sweep_timed(sweeps) * 1e6) for _ in range(3)), is_quad, counts
    mine, picked, counts = rate(None)
    other, other_quad, counts2 = rate(0 if picked else 1)
    assert counts == counts2 and other_quad != picked
    for _ in range(3):  # (again, both sides, as above)
        if mine >= (1.0 - TOL) * other:
            break
        mine = max(mine, rate(None)[0])
        other = rate(0 if picked else 1)[0]
    print(f"{Y} x {X}: the library ({'quad' if picked else 'no quad'}) {mine:.0f} flips/ns, the other way {other:.0f}")
    assert mine >= (1.0 - TOL) * other, f"{Y} x {X}: the library's choice ({'quad' if picked else 'no quad'}) runs {mine:.0f} flips/ns, the other path {other:.0f}"


RING_SHAPES = [(65536, 32768), (65536, 16384), (65536, 8192), (131072, 16384)]  # a rank's slab of 65536^2 over 2 / 4 / 8 GPUs; BASELINE config 4's slab


@pytest.mark.parametrize("X,Y", RING_SHAPES)
def test_ring_slab_choice_within_3_percent_of_its_neighbours(gpu, warm_clock, monkeypatch, X, Y):
    """the shapes of RING slabs (launch rows = Y + 2 G ghost rows, five workgroups per CU at most, the exchange next to a persistent launch of several
    epochs) under the same defence as the lone slabs': a ring of ONE over the peer (IPC) transport -- what a rank of an N-rank ring executes except a link -- at the
    library's choice against half / twice the strip height, one workgroup per CU fewer / more, the split form, and one launch per exchange (ISING_RING_EPOCHS=1,
    the form of rounds 3-5).  Counts after the same sweeps are equal in every cell."""
    monkeypatch.setenv("ISING_ABORT_POLLS", "40000")
    sweeps = max(64, min(2048, int(40e-3 * 3.2e12 / (X * Y)) // 64 * 64))

    def rate(H=0, wgs=0, split=None, epochs=None, check=None):
        for k, v in (("ISING_FUSED_WGS", 256 * wgs if wgs else None), ("ISING_SPLIT", split), ("ISING_RING_EPOCHS", epochs)):
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, str(v))
        slab = ig.IsingSlab(X, Y, seed=1234, temp=TC, ring_halo=True, strip_rows=H)
        try:
            ring = ig.NativeRing(slab, transport="ipc").init()
            shape = slab.launch_shape()[:2] + (slab.sweep_form(sweeps)[0] == 3,)
            ring.sweep(64)
            ring.quiesce()
            got = ring.count()
            if check is not None:
                assert got == check, (X, Y, H, wgs, split, epochs)
            best = 0.0
            import time
            for _ in range(3):
                t0 = time.perf_counter()
                ring.sweep(sweeps)
                ring.quiesce()
                best = max(best, X * Y * sweeps / (time.perf_counter() - t0) * 1e-9)
            ring.close()
            return best, shape, got
        finally:
            slab.close()

    mine, shape, counts = rate()
    H, wg, is_split = shape
    cells = {}
    for h, w in ((H // 2, wg), (2 * H, wg), (H, wg - 1), (H, wg + 1)):
        if h < 1 or h > 16 or Y % h or w < 1 or w > 6:
            continue
        try:
            r, shp, _ = rate(h, w, check=counts)
        except ig.IsingError:
            continue
        if shp[2] == is_split:
            cells[(h, w, "")] = r
    for name, kw in (("split" if not is_split else "fused", {"split": 0 if is_split else 1}), ("a launch per exchange", {"epochs": 1})):
        try:
            r, shp, _ = rate(check=counts, **kw)
            cells[(shp[0], shp[1], name)] = r
        except ig.IsingError:
            pass
    best = max(cells.values(), default=0.0)
    for _ in range(3):
        if mine >= (1.0 - TOL) * best:
            break
        mine = max(mine, rate()[0])
        hb, wb, nb = max(cells, key=cells.get)
        kw = {"": {"H": hb, "wgs": wb}, "split": {"split": 1}, "fused": {"split": 0}, "a launch per exchange": {"epochs": 1}}[nb]
        cells[(hb, wb, nb)] = rate(**kw)[0]
        best = max(cells.values())
    print(f"ring of one {Y} x {X}: library H={H} wgs={wg} split={int(is_split)} {mine:.0f} flips/ns; neighbours "
          + ", ".join(f"H={h} wgs={w} {nm}: {r:.0f}" for (h, w, nm), r in cells.items()))
    assert mine >= (1.0 - TOL) * best, f"ring slab {Y} x {X}: the library's H={H}, {wg} per CU runs {mine:.0f} flips/ns, a neighbour {best:.0f}: {cells}"


# ---- the run-time guard under the tables (ising_update.cpp: guard_*) -------------------------------------------------------------------------
def _guarded_run(X, Y, monkeypatch, env, launches=6):
    """a fresh slab under `env`: `launches` full-length fused launches through ising_sweep (the guard acts between them), then the rate of three more calls; returns
    (flips/ns, guard record, counts, sweeps done)"""
    for k in ("ISING_FUSED_WGS", "ISING_SPLIT", "ISING_GUARD", "ISING_GUARD_EXPECT", "ISING_FUSED_MAX_SWEEPS"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    monkeypatch.setenv("ISING_ABORT_POLLS", "400000")
    with ig.IsingSlab(X, Y, seed=1234, temp=TC, layout=ig.LAYOUT_BALLOT) as s:
        assert s.fused
        per = s.max_sweeps_per_launch
        s.init()
        for _ in range(launches):
            s.sweep(per)
        info = s.guard_info()
        rate = max(X * Y * per / (s.sweep_timed(per) * 1e6) for _ in range(3))
        return rate, info, s.count(), s.launch_shape()[:2]


def test_guard_tries_the_neighbours_and_keeps_the_fastest(gpu, warm_clock, monkeypatch):
    """The mechanism, made deterministic: with an expectation nothing can meet (ISING_GUARD_EXPECT) the table's shape is timed twice, every neighbour once, and the
    fastest of them stays -- within noise of the best of the same shapes asked for one by one; the spins do not depend on any of it."""
    X, Y = 8192, 8192
    r_off, g_off, counts_off, shape_off = _guarded_run(X, Y, monkeypatch, {"ISING_GUARD": 0, "ISING_SPLIT": 0})
    assert g_off["state"] == 0 and g_off["switched"] == 0
    r_on, g, counts_on, shape_on = _guarded_run(X, Y, monkeypatch, {"ISING_GUARD": 1, "ISING_SPLIT": 0, "ISING_GUARD_EXPECT": 1e6})
    assert counts_on == counts_off  # (same seed, same number of sweeps: a shape never changes results)
    assert g["state"] == 3 and (g["table_strip_rows"], g["table_wg_per_cu"]) == shape_off
    assert 4 <= g["launches_timed"] <= 10  # the table's shape twice, two or three neighbours, a step or two further where it kept paying
    assert (g["strip_rows"], g["wg_per_cu"]) == shape_on and g["kept_flips_per_ns"] >= g["table_flips_per_ns"] > 0
    print(f"{Y} x {X}: table {shape_off} {g['table_flips_per_ns']:.0f} flips/ns in the guard's launch, kept {shape_on} {g['kept_flips_per_ns']:.0f}; afterwards {r_on:.0f} against {r_off:.0f} without the guard")
    assert r_on >= 0.95 * r_off
    # a healthy box meets the expectation at once: one timed launch, nothing tried
    _, g2, counts2, shape2 = _guarded_run(X, Y, monkeypatch, {"ISING_GUARD": 1, "ISING_SPLIT": 0})
    assert counts2 == counts_off
    assert g2["state"] == 3 and (g2["switched"] == 1) == (shape2 != shape_off) and g2["kept_flips_per_ns"] >= g2["table_flips_per_ns"]
    # ... and on the plateau a healthy box settles after one timed launch
    _, g3, _, shape3 = _guarded_run(32768, 32768, monkeypatch, {"ISING_GUARD": 1}, launches=3)
    if g3["table_flips_per_ns"] >= 0.8 * g3["expected_flips_per_ns"]:
        assert g3["state"] == 3 and g3["switched"] == 0 and g3["launches_timed"] == 1


@pytest.mark.parametrize("X,Y", [(8192, 1536), (16384, 2176), (16384, 16384), (65536, 1024)])
def test_guard_recovers_from_an_injected_cliff(gpu, warm_clock, monkeypatch, X, Y):
    """An injected bad shape -- ISING_FUSED_WGS one step past a cliff of THIS box, found here by measuring the grids of one to six workgroups per CU with the guard off --
    recovers to within 5 % of its best neighbour inside the guard's handful of launches (the table's shape twice, each neighbour once, on down the slope while it
    pays); results bit-identical."""
    rates = {}
    for wg in range(1, 7):
        try:
            rates[wg], _, _, _ = _guarded_run(X, Y, monkeypatch, {"ISING_GUARD": 0, "ISING_SPLIT": 0, "ISING_FUSED_WGS": 256 * wg}, launches=1)
        except ig.IsingError:
            rates[wg] = 0.0
            continue
    cliff = [wg for wg in range(2, 7) if rates[wg] < 0.85 * rates[wg - 1]]
    print(f"{Y} x {X}: flips/ns by workgroups per CU {', '.join(f'{w}: {r:.0f}' for w, r in rates.items())}")
    if not cliff:
        pytest.skip(f"{Y} x {X}: no cliff between one and six workgroups per CU on this box")
    wg = cliff[0]
    best_nb = max(rates[wg - 1], rates.get(wg + 1, 0.0))
    r, g, c3, shape = _guarded_run(X, Y, monkeypatch, {"ISING_GUARD": 1, "ISING_SPLIT": 0, "ISING_FUSED_WGS": 256 * wg}, launches=10)
    print(f"{Y} x {X}: injected {wg} per CU ({rates[wg]:.0f} flips/ns): the guard kept {shape} after {g['launches_timed']} timed launches, {r:.0f} flips/ns (best neighbour {best_nb:.0f})")
    assert g["state"] == 3 and g["switched"] == 1 and g["launches_timed"] <= 10
    assert r >= 0.95 * best_nb
    assert c3 == _guarded_run(X, Y, monkeypatch, {"ISING_GUARD": 0, "ISING_SPLIT": 0, "ISING_FUSED_WGS": 256 * wg}, launches=10)[2]  # (same seed, same sweeps)


@pytest.mark.parametrize("X,Y", [(65536, 8192), (24576, 24576), (8192, 8192)])
def test_guard_picks_the_form_of_long_calls(gpu, warm_clock, monkeypatch, X, Y):
    """Where the table gives long calls the split form, the guard times one split and one fused launch on this box and keeps the faster: afterwards the slab runs
    within 5 % of the better of the two forms asked for by name, with the same spins."""
    def run(env, calls=4):
        for k in ("ISING_FUSED_WGS", "ISING_SPLIT", "ISING_GUARD", "ISING_GUARD_EXPECT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, str(v))
        with ig.IsingSlab(X, Y, seed=1234, temp=TC) as s:
            per = s.max_sweeps_per_launch
            n = max(2 * per, ((1 << 36) // (X * Y) + per - 1) // per * per)  # a call of 2^36 flips and more: the table's split form applies
            s.init()
            for _ in range(calls):
                s.sweep(n)
            info = s.guard_info()
            rate = max(X * Y * n / (s.sweep_timed(n) * 1e6) for _ in range(2))
            return rate, info, s.count(), s.sweep_form(n)[0]
    r_split, g0, counts, form_s = run({"ISING_GUARD": 0})
    if form_s != 3:
        pytest.skip(f"{Y} x {X}: the table does not give this lattice the split form here")
    r_fused, _, counts_f, form_f = run({"ISING_GUARD": 0, "ISING_SPLIT": 0})
    r, g, counts_g, form_g = run({"ISING_GUARD": 1})
    assert counts == counts_f == counts_g and form_f == 1
    assert g["form_state"] == 3 and g["split_flips_per_ns"] > 0 and g["fused_flips_per_ns"] > 0
    assert (g["split_kept"] == 1) == (form_g == 3) == (not g["fused_flips_per_ns"] > 1.02 * g["split_flips_per_ns"])
    print(f"{Y} x {X}: split {r_split:.0f}, fused {r_fused:.0f} flips/ns by name; the guard timed {g['split_flips_per_ns']:.0f} / {g['fused_flips_per_ns']:.0f} and kept "
          f"{'split' if g['split_kept'] else 'fused'}: {r:.0f}")
    assert r >= 0.95 * max(r_split, r_fused)  # (three separate contexts, a handful of launches each: 5 % is what box-to-box noise allows a suite)
