"""The scripts under examples/ run as a user would run them (one GPU; the ring example with two processes sharing it)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_quickstart(gpu):
    out = _run([os.path.join("examples", "quickstart.py")])
    # BASELINE config 2's lattice: the counts after 256 sweeps are the oracle's golden ones
    assert "after 256 sweeps: up 134168297, down 134267159" in out, out
    assert "C(r) =" in out and "'it': 256" in out


def test_temperature_series(gpu):
    out = _run([os.path.join("examples", "temperature_series.py")])
    rows = re.findall(r"T = ([\d.]+): <\|m\|> = ([\d.]+) \(exact ([\d.]+)\)", out)
    assert len(rows) == 4
    for _, m, exact in rows:
        assert abs(float(m) - float(exact)) < 2e-3


def test_ring_of_processes_on_one_gpu(gpu):
    out = _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29711",
                os.path.join("examples", "ring_of_processes.py")])
    m = re.search(r"2 slabs of 8192 x 32768: after 128 sweeps up (\d+), down (\d+)", out)
    assert m and int(m.group(1)) + int(m.group(2)) == 2 * 8192 * 32768, out


def test_c_caller(gpu, tmp_path):
    """examples/c_caller.c: pedantic C99 against include/ising_hip.h, linked with libising_hip.so, run on the GPU: BASELINE config 2's
    lattice after 0 and 16 sweeps = the oracle's golden counts and bond sum."""
    import json
    exe = str(tmp_path / "c_caller")
    lib = os.path.join(ROOT, "ising_gpu_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_caller.c"), "-o", exe, "-L", lib, "-lising_hip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    pts = {p["sweeps"]: p for p in json.load(open(os.path.join(ROOT, "tests", "golden", "config2_16384.json")))["points"]}
    assert f"sweeps 0: up {pts[0]['up']} down {pts[0]['down']}\n" in r.stdout
    assert f"sweeps 16: up {pts[16]['up']} down {pts[16]['down']} bond_equal {pts[16]['bond_equal']} (" in r.stdout
