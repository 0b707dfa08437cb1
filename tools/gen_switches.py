#!/usr/bin/env python3
"""Writes docs/SWITCHES.md from the library's own table of environment switches (ising_switch_table; needs no GPU).  `--check`: exit 1 when the file is stale."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ising_gpu_amd import _lib  # noqa: E402

HEAD = """# Environment switches of libising_hip.so

A/B and test aids: the defaults are what `ising_create` and the ring pick.  Read ONCE per context, in `ising_create`, into `ising_ctx::pol`
(`ising_host::read_policy`, `csrc/ising_capi.cpp`); nothing in the library calls `getenv` afterwards (`ISING_RCCL_LIB` is process-wide, read when librccl is first
opened).  A ring takes its ring-level switches from slab 0.  The table below is `ising_switch_table()`'s output (regenerate: `python tools/gen_switches.py`);
`tests/test_switches.py` fails when a source file reads a variable the table does not list, or when this file and the library disagree.

"""
TAIL = """
Compile-time (`make variant NAME=... DEFS=...`, never the product library): `ISING_FUSED_TRACE`, `ISING_FUSED_TRACE_COUNTS`, `ISING_QUAD_TRACE`,
`ISING_FUSED_STAGGER=n`, `ISING_POLL_SLEEP=n`, `ISING_BAL_NUM_SGPR`, `ISING_BAL_THREADS`.
"""


def table() -> str:
    lib = _lib.load()
    n = C.c_size_t()
    lib.ising_switch_table(None, 0, C.byref(n))
    buf = C.create_string_buffer(n.value)
    if lib.ising_switch_table(buf, n.value, C.byref(n)) != 0:
        raise SystemExit("ising_switch_table failed")
    return buf.value.decode()


def text() -> str:
    return HEAD + table() + TAIL


if __name__ == "__main__":
    path = os.path.join(ROOT, "docs", "SWITCHES.md")
    if "--check" in sys.argv:
        raise SystemExit(0 if os.path.exists(path) and open(path).read() == text() else 1)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, "w").write(text())
    print(path)
