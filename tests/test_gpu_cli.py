"""The command-line front (ising_gpu_amd/cuIsing) against the reference's transcript format and the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu
CLI = os.path.join(os.path.dirname(ig.LIB_PATH), "cuIsing")


def run(args, cwd=None):
    assert os.path.exists(CLI), "cuIsing not built"
    r = subprocess.run([CLI] + args, capture_output=True, text=True, cwd=cwd, timeout=600)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_readme_transcript_lines_65536(gpu):
    """optimized/README.md:94-131 replayed as one slab: './cuIsing -y 65536 -x 65536 -n 32 -p 16 -t 1.5' must print the
    README's magnetisation lines character for character (the counts are decomposition independent)."""
    out = run(["-y", "65536", "-x", "65536", "-n", "32", "-p", "16", "-t", "1.5"])
    assert "Initial magnetization:  0.000000, up_s:   2147484090, dw_s:   2147483206\n" in out
    assert "        magnetization:  0.000043, up_s:   2147575418, dw_s:   2147391878 (iter:       16)\n" in out
    assert "        magnetization:  0.000074, up_s:   2147641872, dw_s:   2147325424 (iter:       32)\n" in out
    assert "Final   magnetization:  0.000074, up_s:   2147641872, dw_s:   2147325424 (iter:       32)\n" in out
    # run-configuration block as the reference prints it (README.md:107-125)
    for line in ("\tspin/word: 16\n", "\tspins: 4294967296\n", "\tseed: 463463564571\n", "\titerations: 32\n",
                 "\tblock (X, Y): 16, 16\n", "\ttile  (X, Y): 32, 16\n", "\tprint magn. every 16 steps\n",
                 "\ttemp: 1.500000 (0.661030*T_crit)\n", "\ttemp update not set\n", "\tnot using Hamiltonian buffer\n",
                 "\ttotal lattice size:         65536 x    65536\n", "\tmemory: 2048.00 MB (2048.00 MB per GPU)\n"):
        assert line in out, line
    assert re.search(r"Kernel execution time for 32 update steps: \d\.\d+E[+-]\d+ ms, \d+\.\d\d flips/ns \(BW: \d+\.\d\d GB/s\)", out)


def test_two_slabs_on_one_gpu_match_readme_2gpu_run(gpu):
    """The README run itself is 2 devices x 32768 rows: '-y 32768 -x 65536 -d 2' (both slabs mapped to device 0)."""
    out = run(["-y", "32768", "-x", "65536", "-n", "16", "-p", "16", "-d", "2", "-t", "1.5", "--devmap", "0,0"])
    assert "\tlocal lattice size:         32768 x    65536\n" in out
    assert "\tlocal lattice shape: 2 x    32768 x     2048 (   134217728 ulls)\n" in out
    assert "\tmemory: 2048.00 MB (1024.00 MB per GPU)\n" in out
    assert "\tGPU  1 done\n" in out
    assert "Initial magnetization:  0.000000, up_s:   2147484090, dw_s:   2147483206\n" in out
    assert "        magnetization:  0.000043, up_s:   2147575418, dw_s:   2147391878 (iter:       16)\n" in out


def test_cli_series_energy_and_dump_vs_oracle(gpu, oracle_mod, tmp_path):
    X, Y, n, seed = 2048, 64, 24, 4242
    out = run(["-x", str(X), "-y", str(Y), "-n", str(n), "-p", "8", "-a", "1", "-s", str(seed), "-o", "--energy"], cwd=tmp_path)
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=oracle_mod.CRIT_TEMP).init()
    for it in (8, 16, 24):
        orc.sweep(it - orc.it)
        up, dw = orc.count()
        m = abs(up - dw) / (X * Y)
        assert f"        magnetization: {m:9.6f}, up_s: {up:12d}, dw_s: {dw:12d} (iter: {it:8d})\n" in out
        assert f"        energy/spin:   {orc.energy_per_spin():9.6f} (iter: {it:8d})\n" in out
    dump = tmp_path / f"lattice_{Y}x{X}_T_{oracle_mod.CRIT_TEMP:f}_IT_{n:08d}_0.txt"
    assert dump.exists(), os.listdir(tmp_path)
    assert dump.read_bytes() == orc.dump_rows(0, Y)


def test_cli_ramp_and_exppr(gpu, oracle_mod):
    X, Y, seed = 2048, 32, 5
    out = run(["-x", str(X), "-y", str(Y), "-n", "160", "-e", "-t", "1.0", "-u", "0.5,50", "-s", str(seed)])
    assert "tempUpdStep: 0.500000, tempUpdFreq: 50\n" in out
    assert "Changing temperature to 1.500000\n" in out and "Changing temperature to 2.500000\n" in out
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=1.0).init()
    t = np.float32(1.0)
    for it in range(1, 154):
        orc.sweep(1)
        if it % 50 == 0:
            t = np.float32(t + np.float32(0.5))
            orc.temp = float(t)
    up, dw = orc.count()
    m = abs(up - dw) / (X * Y)
    # first exponential-series print is iteration 153 (MIN_EXP_TIME = 152, optimized/main.cu:68,:1827-1831)
    assert f"        magnetization: {m:9.6f} (^2: {m*m:9.6f}), up_s: {up:12d}, dw_s: {dw:12d} (iter: {153:8d})\n" in out


def test_cli_rejects_bad_sizes_like_the_reference(gpu):
    r = subprocess.run([CLI, "-x", "1024", "-y", "1024"], capture_output=True, text=True)
    assert "Please specify an X dim multiple of 2048" in r.stderr


def test_cli_sublattices_transcript(gpu):
    out = run(["-y", "32768", "-x", "65536", "-n", "16", "-p", "16", "-d", "2", "-t", "1.5", "--xsl", "2048", "--ysl", "2048", "--devmap", "0,0"])
    assert "\tusing sub-lattices:\n" in out
    assert "\t\tno. of sub-lattices per GPU:      512\n" in out
    assert "\t\tno. of sub-lattices (total):     1024\n" in out
    assert "\t\tsub-lattices size:              2048 x    2048\n" in out
    assert "        magnetization:  0.000052, up_s:   2147594634, dw_s:   2147372662 (iter:       16)\n" in out  # README.md:188


# ---- the reference's multi-GPU transcripts, every magnetisation line (optimized/README.md:240-249 and :307-316) -----------
README_2GPU = [  # ./cuIsing -y 65536 -x 65536 -n 128 -p 16 -d 2 -t 1.5   (2 x A100, README.md:205-252; 2 x H100 :327-370)
    "Initial magnetization:  0.000005, up_s:   4294989182, dw_s:   4294945410\n",
    "        magnetization:  0.000082, up_s:   4294617248, dw_s:   4295317344 (iter:       16)\n",
    "        magnetization:  0.000249, up_s:   4293898346, dw_s:   4296036246 (iter:       32)\n",
    "        magnetization:  0.000503, up_s:   4292806461, dw_s:   4297128131 (iter:       48)\n",
    "        magnetization:  0.000725, up_s:   4291852263, dw_s:   4298082329 (iter:       64)\n",
    "        magnetization:  0.000904, up_s:   4291086016, dw_s:   4298848576 (iter:       80)\n",
    "        magnetization:  0.001097, up_s:   4290256223, dw_s:   4299678369 (iter:       96)\n",
    "        magnetization:  0.001245, up_s:   4289621029, dw_s:   4300313563 (iter:      112)\n",
    "        magnetization:  0.001418, up_s:   4288877118, dw_s:   4301057474 (iter:      128)\n",
    "Final   magnetization:  0.001418, up_s:   4288877118, dw_s:   4301057474 (iter:      128)\n",
]
README_8GPU = [  # ./cuIsing -y 65536 -x 65536 -n 128 -p 16 -d 8 -t 1.5   (8 x A100, README.md:255-319; 8 x H100 :375-436)
    "Initial magnetization:  0.000010, up_s:  17179689306, dw_s:  17180049062\n",
    "        magnetization:  0.000203, up_s:  17176389528, dw_s:  17183348840 (iter:       16)\n",
    "        magnetization:  0.000402, up_s:  17172963073, dw_s:  17186775295 (iter:       32)\n",
    "        magnetization:  0.000539, up_s:  17170610910, dw_s:  17189127458 (iter:       48)\n",
    "        magnetization:  0.000642, up_s:  17168843228, dw_s:  17190895140 (iter:       64)\n",
    "        magnetization:  0.000749, up_s:  17167009008, dw_s:  17192729360 (iter:       80)\n",
    "        magnetization:  0.000865, up_s:  17165014291, dw_s:  17194724077 (iter:       96)\n",
    "        magnetization:  0.000941, up_s:  17163708078, dw_s:  17196030290 (iter:      112)\n",
    "        magnetization:  0.001023, up_s:  17162287230, dw_s:  17197451138 (iter:      128)\n",
    "Final   magnetization:  0.001023, up_s:  17162287230, dw_s:  17197451138 (iter:      128)\n",
]


def peer_matrix_block(n):
    """The "GPUs direct access matrix" block of an n-device run in which every pair is linked, as optimized/main.cu:1508-1531 prints it."""
    return ("GPUs direct access matrix:\n       " + "".join(f"{i:4d}" for i in range(n)) + "\n"
            + "".join(f"GPU {i:2d}:" + "   V" * n + "\n" for i in range(n)) + "\n")


def test_readme_two_gpu_transcript_every_line_at_its_own_decomposition(gpu):
    """README.md:205-252 at the reference's own decomposition: 2 slabs of 65536 x 65536 (both on device 0; one MI355X holds
    the whole 131072 x 65536 lattice).  All nine magnetisation lines and the final one, character for character."""
    out = run(["-y", "65536", "-x", "65536", "-n", "128", "-p", "16", "-d", "2", "-t", "1.5", "--devmap", "0,0"])
    for line in README_2GPU + ["\tspins: 8589934592\n", "\tgrid  (X, Y): 32, 4096\n", "\tlocal lattice size:         65536 x    65536\n",
                               "\ttotal lattice size:        131072 x    65536\n", "\tlocal lattice shape: 2 x    65536 x     2048 (   268435456 ulls)\n",
                               "\ttotal lattice shape: 2 x   131072 x     2048 (   536870912 ulls)\n", "\tmemory: 4096.00 MB (2048.00 MB per GPU)\n",
                               "Setting up multi-gpu configuration:\n", "\tGPU  0 done\n", "\tGPU  1 done\n"]:
        assert line in out, line
    # the block between "Using GPUs" and "Run configuration" (optimized/main.cu:1508-1537; README.md:213-216): slabs of one device reach each other
    assert "ECC on)\n\n" + peer_matrix_block(2) + "Run configuration:\n" in out


@pytest.mark.parametrize("ndev,lines", [(2, "README_2GPU"), (8, "README_8GPU")])
def test_readme_multi_gpu_transcripts_on_that_many_devices(gpu, ndev, lines):
    """`cuIsing -d N` as the reference's README types it -- no --devmap: N devices, one slab each, one host thread driving all of them
    (optimized/main.cu:1764-1805), RCCL on the comm streams.  Needs N GPUs; the 1-GPU box runs the same lattices through --devmap above."""
    import ising_gpu_amd as ig
    if ig.device_count() < ndev:
        pytest.skip(f"needs >= {ndev} GPUs")
    out = run(["-y", "65536", "-x", "65536", "-n", "128", "-p", "16", "-d", str(ndev), "-t", "1.5"])
    for line in globals()[lines] + [f"\tGPU {ndev - 1:2d} done\n"]:
        assert line in out, line
    assert peer_matrix_block(ndev) + "Run configuration:\n" in out  # (a node whose devices all see each other, as the README's)


def test_readme_two_gpu_transcript_as_one_slab(gpu):
    """The same lattice as ONE slab of 131072 rows (fused launches, no ring): the counts do not depend on the decomposition."""
    out = run(["-y", "131072", "-x", "65536", "-n", "128", "-p", "16", "-t", "1.5"])
    for line in README_2GPU:
        assert line in out, line


def test_readme_eight_gpu_transcript_every_line_at_its_own_decomposition(gpu):
    """README.md:255-319: 8 slabs of 65536 x 65536 = 2^35 spins (4 GiB at 1 bit per spin: all eight slabs on device 0), 128
    sweeps, printed every 16: all nine magnetisation lines and the final one, character for character."""
    out = run(["-y", "65536", "-x", "65536", "-n", "128", "-p", "16", "-d", "8", "-t", "1.5", "--devmap", "0,0,0,0,0,0,0,0"])
    for line in README_8GPU + ["\tspins: 34359738368\n", "\ttotal lattice size:        524288 x    65536\n",
                               "\ttotal lattice shape: 2 x   524288 x     2048 (  2147483648 ulls)\n", "\tmemory: 16384.00 MB (2048.00 MB per GPU)\n",
                               "\tGPU  7 done\n"]:
        assert line in out, line
    assert "ECC on)\n\n" + peer_matrix_block(8) + "Run configuration:\n" in out  # README.md:270-280


def test_readme_eight_gpu_transcript_as_one_slab(gpu):
    out = run(["-y", "524288", "-x", "65536", "-n", "128", "-p", "16", "-t", "1.5"])
    for line in README_8GPU:
        assert line in out, line


def test_baseline_config2_command_line_at_full_length(gpu):
    """BASELINE config 2 as a user types it -- 16384 x 16384, T = T_c (-a 1), seed 1234, 10^5 sweeps, a line every 10000 -- against
    the oracle's full-length golden: every magnetisation line, character for character."""
    import json
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config2_16384_full.json")))
    out = run(["-x", "16384", "-y", "16384", "-n", "100000", "-a", "1", "-s", "1234", "-p", "10000"])
    n = 16384 * 16384
    pts = {p["sweeps"]: p for p in fx["points"]}
    for it in range(10000, 100001, 10000):
        p = pts[it]
        assert (f"        magnetization: {abs(p['up'] - p['down']) / n:9.6f}, up_s: {p['up']:12d}, dw_s: {p['down']:12d} "
                f"(iter: {it:8d})\n") in out, it
    p = pts[100000]
    assert f"Final   magnetization: {abs(p['up'] - p['down']) / n:9.6f}, up_s: {p['up']:12d}, dw_s: {p['down']:12d} (iter:   100000)\n" in out
    m = re.search(r"Kernel execution time for 100000 update steps: \S+ ms, (\d+\.\d\d) flips/ns", out)
    assert m and float(m.group(1)) > 2500.0, out[-400:]  # (3300 on an idle MI355X; the floor only catches a fall back to a slow path)


# ---- the corners of the print logic (round-4 verdict): -e beyond its first point, -e with -p, -m with -p and with -e, -s 0 ---------------
def _exp_points(nsteps):
    """The reference's -e series (optimized/main.cu:1211-1228) as 1-based iterations: 153, then every rung of rint(2^(k/4)) that has at least
    doubled the last point (+ 1: the loop prints when its 0-based index equals the list entry, :1827-1831)."""
    pts, rung, k = [152], 0, 0
    while k < nsteps and rung < nsteps:
        rung = int(np.rint(2.0 ** (k / 4.0)))
        if rung >= 2 * pts[-1] and len(pts) < 200:
            pts.append(rung)
        k += 1
    return [p + 1 for p in pts]


def test_exp_series_is_the_reference_s():
    # (worked by hand from the reference's loop: 2^(33/4) = 304.4, 2^(37/4) = 608.9, 2^(41/4) = 1217.7, 2^(45/4) = 2435.5 -> 2435 < 2436, so 2^(46/4) = 2896.3)
    assert _exp_points(3000)[:5] == [153, 305, 610, 1219, 2897]


def _series(oracle_mod, X, Y, seed, temp, iters):
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp).init()
    out = {}
    for it in iters:
        orc.sweep(it - orc.it)
        up, dw = orc.count()
        out[it] = (abs(up - dw) / (X * Y), up, dw)
    return out


def test_cli_exppr_series_and_print_freq_ignored(gpu, oracle_mod):
    """-e over four points of the series (153, 305, 610, 1219); with -p next to it printFreq is ignored (optimized/main.cu:1473-1475)."""
    X, Y, seed, n = 2048, 32, 77, 1300
    pts = [p for p in _exp_points(n) if p <= n]
    assert pts == [153, 305, 610, 1219]
    ser = _series(oracle_mod, X, Y, seed, 2.0, pts + [n])
    for extra in ([], ["-p", "100"]):
        out = run(["-x", str(X), "-y", str(Y), "-n", str(n), "-e", "-t", "2.0", "-s", str(seed)] + extra)
        assert "\tprint magn. following exponential series\n" in out
        lines = [ln for ln in out.splitlines() if ln.startswith("        magnetization:")]
        assert len(lines) == len(pts), lines  # (and none at multiples of 100)
        for it, ln in zip(pts, lines):
            m, up, dw = ser[it]
            assert ln == f"        magnetization: {m:9.6f} (^2: {m*m:9.6f}), up_s: {up:12d}, dw_s: {dw:12d} (iter: {it:8d})"
        m, up, dw = ser[n]
        assert f"Final   magnetization: {m:9.6f}, up_s: {up:12d}, dw_s: {dw:12d} (iter: {n:8d})\n" in out


def _first_hit(ser, iters, k):
    """a target magnetisation that the k-th print point is the first to come within 1e-3 of (or None)"""
    tgt = ser[iters[k]][0]
    return tgt if all(abs(ser[it][0] - tgt) >= 1.0e-3 for it in iters[:k]) else None


def test_cli_magn_early_exit_with_print(gpu, oracle_mod):
    """-m with -p: the run stops at the first print point whose magnetisation is within 1e-3 of the target; the final line and the
    performance line report that iteration (optimized/main.cu:65, :1819-1824: j is bumped before the break)."""
    X, Y, seed, p, n = 2048, 64, 99, 4, 200
    iters = list(range(p, n + 1, p))
    ser = _series(oracle_mod, X, Y, seed, 1.0, iters)  # T = 1.0: the magnetisation of a 2^17-spin lattice moves by more than 1e-3 between points
    k = next(k for k in range(3, len(iters)) if _first_hit(ser, iters, k) is not None)
    tgt, stop = ser[iters[k]][0], iters[k]
    out = run(["-x", str(X), "-y", str(Y), "-n", str(n), "-p", str(p), "-t", "1.0", "-s", str(seed), "-m", f"{tgt:.9f}"])
    assert f"\tearly exit if magn. == {tgt:f}+-0.001000\n" in out
    lines = [ln for ln in out.splitlines() if ln.startswith("        magnetization:")]
    assert len(lines) == k + 1
    m, up, dw = ser[stop]
    assert lines[-1] == f"        magnetization: {m:9.6f}, up_s: {up:12d}, dw_s: {dw:12d} (iter: {stop:8d})"
    assert f"Final   magnetization: {m:9.6f}, up_s: {up:12d}, dw_s: {dw:12d} (iter: {stop:8d})\n" in out
    assert re.search(rf"Kernel execution time for {stop} update steps:", out)
    # a target nothing comes near: the run goes to its end
    out = run(["-x", str(X), "-y", str(Y), "-n", "40", "-p", str(p), "-t", "1.0", "-s", str(seed), "-m", "2.0"])
    assert "(iter:       40)\n" in out and re.search(r"Kernel execution time for 40 update steps:", out)
    # without -p / -e the target is never looked at (and the configuration block does not mention it)
    out = run(["-x", str(X), "-y", str(Y), "-n", "8", "-t", "1.0", "-s", str(seed), "-m", f"{ser[iters[1]][0]:.9f}"])
    assert "early exit" not in out and "(iter:        8)\n" in out


def test_cli_magn_early_exit_with_exppr(gpu, oracle_mod):
    """-m with -e: looked at at the series' points only (optimized/main.cu:1840-1845)."""
    X, Y, seed, n = 2048, 32, 31, 700
    pts = [153, 305, 610]
    ser = _series(oracle_mod, X, Y, seed, 1.0, pts)
    tgt = _first_hit(ser, pts, 1)
    if tgt is None:
        pytest.skip("the first two points of this lattice lie within 1e-3 of each other")
    out = run(["-x", str(X), "-y", str(Y), "-n", str(n), "-e", "-t", "1.0", "-s", str(seed), "-m", f"{tgt:.9f}"])
    lines = [ln for ln in out.splitlines() if ln.startswith("        magnetization:")]
    assert len(lines) == 2 and lines[1].endswith("(iter:      305)")
    m, up, dw = ser[305]
    assert f"Final   magnetization: {m:9.6f}, up_s: {up:12d}, dw_s: {dw:12d} (iter: {305:8d})\n" in out
    assert re.search(r"Kernel execution time for 305 update steps:", out)


def test_cli_random_seed_is_printed_and_reproducible(gpu):
    """-s 0 draws a seed (optimized/main.cu:1329-1334); the configuration block prints it and a second run with that seed repeats the run."""
    args = ["-x", "2048", "-y", "32", "-n", "12", "-p", "4", "-a", "1"]
    out = run(args + ["-s", "0"])
    seed = int(re.search(r"\tseed: (\d+)\n", out).group(1))
    assert 0 < seed <= 0x7FFFFFFFF
    again = run(args + ["-s", str(seed)])
    pick = lambda o: [ln for ln in o.splitlines() if "magnetization" in ln]  # noqa: E731
    assert pick(out) == pick(again) and len(pick(out)) == 5


@pytest.mark.parametrize("X,Y,layout", [(8192, 2048, None), (8192, 128, "ballot"), (2048, 64, None)], ids=["fused-auto", "ballot-small", "dense-tiles"])
def test_cli_print_with_energy_rides_in_the_launches(gpu, oracle_mod, X, Y, layout):
    """`-p N --energy` (north_star: magnetisation AND energy time-series): the same lines as the reference's order of events gives, whether the print
    points ride inside the fused launches (ising_sweep_counted with its bond sums) or the run sweeps and measures in turn."""
    seed, n, p = 606, 40, 8
    args = ["-x", str(X), "-y", str(Y), "-n", str(n), "-p", str(p), "-a", "1", "-s", str(seed), "--energy"] + (["--layout", layout] if layout else [])
    out = run(args)
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=oracle_mod.CRIT_TEMP).init()
    assert f"Initial energy/spin:   {orc.energy_per_spin():9.6f}\n" in out
    for it in range(p, n + 1, p):
        orc.sweep(it - orc.it)
        up, dw = orc.count()
        m = abs(up - dw) / (X * Y)
        assert f"        magnetization: {m:9.6f}, up_s: {up:12d}, dw_s: {dw:12d} (iter: {it:8d})\n        energy/spin:   {orc.energy_per_spin():9.6f} (iter: {it:8d})\n" in out, it
    assert f"Final   energy/spin:   {orc.energy_per_spin():9.6f}\n" in out


def test_single_device_run_prints_no_peer_matrix(gpu):
    """optimized/main.cu:1496: the direct-access block belongs to runs on several devices only."""
    out = run(["-x", "2048", "-y", "2048", "-n", "4"])
    assert "GPUs direct access matrix" not in out and "ECC on)\n\nRun configuration:\n" in out
