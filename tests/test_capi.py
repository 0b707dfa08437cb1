"""CPU checks of the drop-in boundary: libising_hip.so loads and exports every symbol include/ising_hip.h
declares (no compute calls without a GPU), the ctypes prototypes cover the header, and the product never imports
the oracle."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def header_symbols(which=("ising_hip.h", "ising_hip_testing.h")):
    """Functions the headers under include/ declare (ising_hip_testing.h: the test-only entry points)."""
    txt = "".join(open(os.path.join(ROOT, "include", h)).read() for h in which)
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ising_[a-z_]+)\s*\(", txt)))


def test_headers_under_include_are_all_known():
    assert sorted(f for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h")) == ["ising_hip.h", "ising_hip_testing.h"]
    assert header_symbols(("ising_hip_testing.h",)) == ["ising_batch_debug_fault", "ising_debug_fault", "ising_debug_launch_shape"]      # test aids stay out of the boundary header
    assert "ising_debug_fault" not in header_symbols(("ising_hip.h",))


def test_library_exports_every_declared_symbol():
    from ising_gpu_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libising_hip.so not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared under include/ but not exported"


def test_ctypes_prototypes_cover_header():
    from ising_gpu_amd import _lib
    assert sorted(_lib.PROTOTYPES) == header_symbols()
    _lib.load()


def test_header_cites_reference_lines():
    txt = open(os.path.join(ROOT, "include", "ising_hip.h")).read()
    assert txt.count("optimized/main.cu:") >= 15


def test_no_gpu_means_loud_failure_not_fallback():
    import ising_gpu_amd as ig
    if ig.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(ig.IsingError):
        ig.IsingSlab(2048, 16)


def test_argument_validation_messages_without_gpu():
    """Size rules of optimized/main.cu:1412-1421 are enforced before any device work."""
    import ising_gpu_amd as ig
    for X, Y in ((1024, 16), (2048, 8), (0, 16)):
        with pytest.raises(ig.IsingError) as e:
            ig.IsingSlab(X, Y)
        assert "multiple of" in str(e.value) or "no HIP device" in str(e.value)


def test_integration_doc_maps_every_entry_point():
    """INTEGRATION.md's table says, for every function of the header, which reference interface it replaces (or that it is new).
    Families are abbreviated there (`ising_batch_create` / `_sweep`, `ising_init_couplings*`, `ising_required_bytes[_layout]`)."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    doc += " ".join(m.group(1) + m.group(2) for m in re.finditer(r"(ising_[a-z_0-9]+)\[(_[a-z_0-9]+)\]", doc))
    missing = []
    for sym in header_symbols():
        family, tail = sym.rsplit("_", 1)
        lines = [ln for ln in doc.splitlines() if family in ln]
        if sym in doc or (family + "*") in doc or any(("`_" + tail + "`") in ln or ("/ `_" + tail) in ln or ("_" + tail + "`") in ln for ln in lines):
            continue
        missing.append(sym)
    assert not missing, missing


def test_header_is_plain_c_and_a_c_caller_links(tmp_path):
    """The boundary is a C ABI: include/ising_hip.h compiles as pedantic C99, and a C program (what a cgo / JNI / ctypes-free host
    would be) links against libising_hip.so and calls it -- only entry points that need no device here."""
    src = tmp_path / "caller.c"
    src.write_text(r"""
#include <stdio.h>
#include <string.h>
#include "ising_hip.h"
int main(void) {
	ising_config cfg;
	memset(&cfg, 0, sizeof cfg);
	cfg.nslabs = 1; cfg.temp = 2.0f;
	/* optimized/main.cu:1412-1421: X must be a multiple of 2048 -- refused before any device work */
	cfg.X = 1000; cfg.Y = 16;
	ising_ctx *ctx = NULL;
	int rc = ising_create(&cfg, &ctx);
	printf("%zu %zu %d %d\n", ising_required_bytes(2048, 16), ising_required_bytes_layout(2048, 16, ISING_LAYOUT_DENSE), rc != ISING_OK,
	       ising_last_error()[0] != 0);
	return 0;
}
""")
    exe = tmp_path / "caller"
    lib = os.path.join(ROOT, "ising_gpu_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", lib, "-lising_hip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    full = 2 * (16 + 2) * (2048 // 32) * 8
    assert out.stdout.split() == [str(full), str(full // 4), "1", "1"], out.stdout


def test_product_never_imports_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ising_gpu_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M) or "ising_oracle" in txt or "oracle/" in txt:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_bench_helpers_and_full_size_goldens():
    """bench.py's golden lookup (the counts it checks itself against) and the committed full-size fixtures."""
    import json
    import bench
    gold = os.path.join(ROOT, "tests", "golden")
    # the point the round-1 driver run hit: 65536^2, T_c, seed 1234, 5 + 20 sweeps
    assert bench.golden_counts(65536, 65536, 1234, 25) == (2146725784, 2148241512)
    assert bench.golden_counts(65536, 65536, 1234, 0) == (2147471050, 2147496246)
    assert bench.golden_counts(65536, 65536, 1234, 26) is None and bench.golden_counts(65536, 65536, 99, 25) is None
    for n in (2, 4, 8):
        up, down = bench.golden_counts(65536, 65536 * n, 1234, 25)
        assert up + down == n * 65536 * 65536
    c4 = json.load(open(os.path.join(gold, "config4_131072.json")))
    for pt in c4["points"]:
        assert sum(pt["slab_up"]) == pt["up"] and pt["up"] + pt["down"] == 131072 * 131072
    # the 8-slab ring's initial state is the single lattice's: same counts as config 4's?  (other geometry: 524288 x 65536)
    r8 = [r for r in json.load(open(os.path.join(gold, "ring_65536_tc.json")))["rings"] if r["nslabs"] == 8][0]
    assert r8["Ytot"] == 524288 and r8["points"][0]["sweeps"] == 0


def test_every_point_of_a_scaling_run_has_a_golden():
    """for each workload of bench.py at N = 1, 2, 4, 8 and both sweep counts a scaling run uses (the driver's
    --warmup 5 --steps 20 = 25, the default 16 + 128 = 144) the oracle's counts of the TOTAL lattice are committed -- so that
    `parity_checked` is true or false, never null, at every point.  The lookup keys on the total lattice, not on the number of
    slabs: the strong-scaling splits of 65536^2 all hit bench_65536_tc.json (optimized/main.cu:514, :1590-1591)."""
    import bench
    for name, (x, rows_of, kind) in bench.WORKLOADS.items():
        for n in (1, 2, 4, 8):
            ytot = rows_of(n) * n
            for sweeps in (0, 25, 144):
                got = bench.golden_counts(x, ytot, 1234, sweeps)
                assert got is not None, (name, n, sweeps)
                assert got[0] + got[1] == x * ytot, (name, n, sweeps)
            if kind == "strong":
                assert ytot == 65536 and bench.golden_counts(x, ytot, 1234, 25) == (2146725784, 2148241512)
    # one physical fact that ties the files together: lattices with the same number of Philox streams start with the same number of up
    # spins (latticeInit_k draws per (block row, column group) stream, optimized/main.cu:107-149) -- 131072 x 32768 and 65536^2
    assert bench.golden_counts(131072, 32768, 1234, 0) == bench.golden_counts(65536, 65536, 1234, 0)
    # and config 4 at N = 8 is the lattice config4_131072.json pinned in round 2 (points 0 there and in scaling.json agree)
    rec = bench.golden_records()[(131072, 131072, 1234)]
    assert {0, 1, 2, 5, 25, 144} <= set(rec)


def test_hand_waited_loads_of_the_fused_kernels_pass_the_isa_check():
    """The fused ballot kernels issue their lattice loads as inline assembly and wait for them by hand (ising_ballot.hip);
    the build checks the generated ISA: nothing touches such a register before the covering s_waitcnt.  Here: the check
    ran on the sources as they are, and it saw all eleven loads in each of the ten fused kernels (with / without couplings and
    sub-lattices, streamed or not, batched, with in-launch counts)."""
    csrc = os.path.join(ROOT, "ising_gpu_amd", "csrc")
    subprocess.check_call(["make", "-s", "-C", csrc, "check-asm"])   # (no-op when the library was built from these sources)
    report = open(os.path.join(csrc, ".ising_ballot.s.report")).read().splitlines()
    fused = [ln for ln in report if "ballot_update_k" in ln and "Lb1ELi" in ln]   # ballot_update_k<SUBL, USEJ, FUSED = true, NT>
    # (six lattice / mask loads of the row loop; the unit's first two rows at the two places a unit may wait late -- five more loads with the row-end word)
    assert len(fused) == 10 and all(ln.endswith("11 inline-assembly loads checked") for ln in fused), report
    # the split launches' word units (round 5): two loads for a unit's first rows, four per row; the looks of its waits carry their own
    split = [ln for ln in report if "ballot_split_k" in ln]
    assert len(split) == 2 and all(ln.endswith("6 inline-assembly loads checked") for ln in split), report
    assert not any("touches" in ln for ln in report)
