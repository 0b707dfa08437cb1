"""one table of every ISING_* environment switch (ising_switch_table in csrc/ising_capi.cpp = docs/SWITCHES.md), held against the sources -- a
variable that the library or its Python mirror reads and the table does not list fails here, and so does a stale docs/SWITCHES.md."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_switches  # noqa: E402


def documented():
    return set(re.findall(r"^\| `(ISING_[A-Z0-9_]+)` \|", gen_switches.table(), re.M))


def test_every_environment_read_in_the_sources_is_in_the_table():
    names = documented()
    assert len(names) >= 30
    read = {}
    for path in glob.glob(os.path.join(ROOT, "ising_gpu_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "ising_gpu_amd", "*.py")):
        if not path.endswith((".cpp", ".hip", ".hpp", ".h", ".py")):
            continue
        src = open(path).read()
        # C++: getenv("ISING_X") and read_policy's num("ISING_X", ...); Python: os.environ.get("ISING_X" / os.environ["ISING_X"]
        for m in re.finditer(r'(?:getenv|num)\(\s*"(ISING_[A-Z0-9_]+)"|environ(?:\.get\(|\[)\s*"(ISING_[A-Z0-9_]+)"', src):
            read.setdefault(m.group(1) or m.group(2), set()).add(os.path.basename(path))
    assert read, "the scan found no environment reads at all: the patterns are out of date"
    missing = {k: sorted(v) for k, v in read.items() if k not in names}
    assert not missing, f"environment switches read in the sources but absent from ising_switch_table: {missing}"
    unread = sorted(n for n in names if n not in read)
    assert not unread, f"ising_switch_table lists switches nothing reads: {unread}"


def test_docs_switches_md_is_the_library_table():
    assert open(os.path.join(ROOT, "docs", "SWITCHES.md")).read() == gen_switches.text(), "docs/SWITCHES.md is stale: python tools/gen_switches.py"
