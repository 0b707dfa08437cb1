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
