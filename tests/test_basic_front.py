"""BASELINE config 1's command line: oracle/ising_basic_cpu has the surface of the reference's basic_python/ising_basic.py
(flags :43-53, transcript :208-259, lattice file :137-151) over the CPU baseline's algorithm (oracle/basic_cpu.c)."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
EXE = os.path.join(ROOT, "oracle", "ising_basic_cpu")


@pytest.fixture(scope="module", autouse=True)
def _built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])  # (a no-op when __graft_entry__.build() ran)
    assert os.path.exists(EXE)


def test_basic_front_report_block_and_lattice_file(tmp_path, oracle_mod):
    n, m, w, it, seed = 64, 96, 5, 20, 77
    r = subprocess.run([EXE, "-x", str(n), "-y", str(m), "-w", str(w), "-n", str(it), "-a", "0.9", "-s", str(seed), "-o", "-t", "2"],
                       capture_output=True, text=True, cwd=tmp_path, timeout=120)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    head, report = out.split("REPORT:\n")
    assert head == "Starting warmup...\nStarting trial iterations...\nCompleted 1/20 iterations...\n"
    lines = report.splitlines()
    assert lines[0] == "\tnGPUs: 0" and lines[1] == "\ttemperature: 0.9 * 2.26918531421" and lines[2] == f"\tseed: {seed}"
    assert lines[3] == f"\twarmup iterations: {w}" and lines[4] == f"\ttrial iterations: {it}" and lines[5] == f"\tlattice dimensions: {n} x {m}"
    assert re.fullmatch(r"\telapsed time: [0-9.e+-]+ sec", lines[6]) and re.fullmatch(r"\tupdates per ns: [0-9.e+-]+", lines[7])
    mabs = float(re.fullmatch(r"\taverage magnetism \(absolute\): ([0-9.e+-]+)", lines[8]).group(1))
    # the same run through the library the bench's cpu_baseline uses: same spins, so the same magnetisation and the same file
    b = oracle_mod.BasicCpuIsing(n, m, alpha=0.9, seed=seed)
    b.sweeps(w + it)
    mm, _ = b.observables()
    assert mabs == abs(mm)
    lat = np.loadtxt(tmp_path / "final_rank0.txt", dtype=np.int8)
    assert lat.shape == (n, m)
    want = np.zeros((n, m), dtype=np.int8)  # write_lattice, ising_basic.py:137-151
    want[0::2, 0::2], want[0::2, 1::2] = b.black[0::2], b.white[0::2]
    want[1::2, 1::2], want[1::2, 0::2] = b.black[1::2], b.white[1::2]
    assert np.array_equal(lat, want)


def test_basic_front_rejects_what_the_reference_rejects(tmp_path):
    r = subprocess.run([EXE, "-x", "64", "-y", "63"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode != 0 and "lattice_m must be an even value" in r.stderr
