"""CPU oracle for the packed checkerboard-Metropolis path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product (``ising_gpu_amd``) must never do so.  See ``ising_oracle.c`` for the reference
file:line map and ``basic_cpu.c`` for the byte-per-spin baseline.
"""
from .pyoracle import (  # noqa: F401
    BLACK, WHITE, CRIT_TEMP, SEED_DEF, build, lib, OracleLattice, philox4x32_10, uniform, exp_table,
    site_draw, BasicCpuIsing, OracleSlab, OracleGhostSlab, set_threads, max_threads,
)
