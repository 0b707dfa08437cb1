"""ctypes front for libising_oracle.so (TEST INFRASTRUCTURE ONLY, see package docstring)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libising_oracle.so")

BLACK, WHITE = 0, 1
CRIT_TEMP = float(np.float32(2.26918531421))  # optimized/main.cu:42
SEED_DEF = 463463564571  # optimized/main.cu:63

_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc, no reference sources involved)."""
    srcs = [os.path.join(_HERE, f) for f in ("ising_oracle.c", "basic_cpu.c", "Makefile")]
    stale = not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u64p = C.POINTER(C.c_uint64)
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.orc_uniform.argtypes = [C.c_uint32]
        L.orc_uniform.restype = C.c_float
        L.orc_exp_table.argtypes = [C.c_float, C.POINTER(C.c_float)]
        L.orc_init.argtypes = [u64p, u64p, C.c_int64, C.c_int64, C.c_uint64]
        L.orc_init.restype = C.c_int
        L.orc_update_color.argtypes = [u64p, u64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_uint64,
                                       C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.orc_update_color.restype = C.c_int
        L.orc_sweep.argtypes = [u64p, u64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_uint64,
                                C.c_int, C.c_int, C.c_float]
        L.orc_sweep.restype = C.c_int
        L.orc_count.argtypes = [u64p, u64p, C.c_int64, C.c_int64, u64p, u64p]
        L.orc_bond_equal.argtypes = [u64p, u64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64]
        L.orc_bond_equal.restype = C.c_int64
        L.orc_dump_rows.argtypes = [u64p, u64p, C.c_int64, C.c_int64, C.c_int64, C.c_char_p]
        L.orc_dump_rows.restype = C.c_int64
        L.orc_corr.argtypes = [u64p, u64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_int64)]
        L.orc_ham_init_black.argtypes = [u64p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_float]
        L.orc_ham_init_white.argtypes = [u64p, u64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64]
        L.orc_update_color_J.argtypes = [u64p, u64p, u64p, u64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_uint64,
                                         C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.orc_update_color_J.restype = C.c_int
        L.orc_site_draw.argtypes = [C.c_int64, C.c_uint64, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int]
        L.orc_site_draw.restype = C.c_uint32
        L.orc_init_slab.argtypes = [u64p, u64p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64]
        L.orc_init_slab.restype = C.c_int
        L.orc_update_color_slab.argtypes = [u64p, u64p, u64p, u64p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64,
                                            C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int64, C.c_int64]
        L.orc_update_color_slab.restype = C.c_int
        L.orc_bond_equal_slab.argtypes = [u64p, u64p, u64p, u64p, C.c_int64, C.c_int64, C.c_int64]
        L.orc_bond_equal_slab.restype = C.c_int64
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_max_threads.restype = C.c_int
        i8p = C.POINTER(C.c_int8)
        f32p = C.POINTER(C.c_float)
        L.basic_init.argtypes = [i8p, i8p, C.c_int64, C.c_int64, C.c_uint64, f32p]
        L.basic_sweeps.argtypes = [i8p, i8p, C.c_int64, C.c_int64, C.c_float, C.c_uint64, C.c_int64, C.c_int64, f32p]
        L.basic_observables.argtypes = [i8p, i8p, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def _u64(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return tuple(int(v) for v in o)


def uniform(x: int) -> float:
    return float(lib().orc_uniform(C.c_uint32(x)))


def exp_table(temp: float) -> np.ndarray:
    t = (C.c_float * 10)()
    lib().orc_exp_table(C.c_float(temp), t)
    return np.array(list(t), dtype=np.float32).reshape(2, 5)


def site_draw(X, seed, it, color, row, word, nib) -> int:
    return int(lib().orc_site_draw(X, C.c_uint64(seed), it, color, row, word, nib))


class OracleLattice:
    """Whole (all-slab) lattice on the host in the reference's packed layout: black[Ytot][X/32], white[...]."""

    def __init__(self, X: int, Ytot: int, seed: int = SEED_DEF, temp: float = 0.1 * CRIT_TEMP,
                 XSL: int = 0, YSL: int = 0):
        if X % 2048 or Ytot % 16:
            raise ValueError("X must be a multiple of 2048 and Y of 16 (optimized/main.cu:1412-1421)")
        self.X, self.Y, self.seed, self.temp = X, Ytot, seed, float(np.float32(temp))
        self.XSL, self.YSL = XSL, YSL
        self.lld = X // 32
        self.black = np.zeros((Ytot, self.lld), dtype=np.uint64)
        self.white = np.zeros((Ytot, self.lld), dtype=np.uint64)
        self.it = 0  # completed sweeps

    def init(self):
        rc = lib().orc_init(_u64(self.black), _u64(self.white), self.X, self.Y, C.c_uint64(self.seed))
        assert rc == 0
        self.it = 0
        return self

    def init_couplings(self, prob: float):
        """-J prob: hamiltInitB_k with seed+1, then hamiltInitW_k (optimized/main.cu:1729-1742)."""
        p = float(np.float32(min(max(0.0, prob), 1.0)))
        self.hamB = np.zeros_like(self.black)
        self.hamW = np.zeros_like(self.white)
        lib().orc_ham_init_black(_u64(self.hamB), self.X, self.Y, 0, C.c_uint64(self.seed + 1), C.c_float(p))
        lib().orc_ham_init_white(_u64(self.hamB), _u64(self.hamW), self.X, self.Y, self.XSL, self.YSL)
        return self

    def update_color(self, it: int, color: int):
        tab = (C.c_float * 10)()
        lib().orc_exp_table(C.c_float(self.temp), tab)
        rc = lib().orc_update_color(_u64(self.black), _u64(self.white), self.X, self.Y, self.XSL, self.YSL,
                                    C.c_uint64(self.seed), it, color, tab)
        assert rc == 0, rc

    def sweep(self, n: int = 1):
        if getattr(self, "hamB", None) is not None:
            tab = (C.c_float * 10)()
            lib().orc_exp_table(C.c_float(self.temp), tab)
            for it in range(self.it + 1, self.it + 1 + n):
                for color in (BLACK, WHITE):
                    rc = lib().orc_update_color_J(_u64(self.black), _u64(self.white), _u64(self.hamB), _u64(self.hamW),
                                                  self.X, self.Y, self.XSL, self.YSL, C.c_uint64(self.seed), it, color, tab)
                    assert rc == 0, rc
            self.it += n
            return self
        rc = lib().orc_sweep(_u64(self.black), _u64(self.white), self.X, self.Y, self.XSL, self.YSL,
                             C.c_uint64(self.seed), self.it + 1, n, C.c_float(self.temp))
        assert rc == 0, rc
        self.it += n
        return self

    def count(self):
        up, dw = C.c_uint64(), C.c_uint64()
        lib().orc_count(_u64(self.black), _u64(self.white), self.X, self.Y, C.byref(up), C.byref(dw))
        return int(up.value), int(dw.value)

    def bond_equal(self) -> int:
        return int(lib().orc_bond_equal(_u64(self.black), _u64(self.white), self.X, self.Y, self.XSL, self.YSL))

    def energy_per_spin(self) -> float:
        n = self.X * self.Y
        return -(2 * self.bond_equal() - 2 * n) / n

    def corr(self, ncorr: int = 128):
        out = (C.c_int64 * ncorr)()
        lib().orc_corr(_u64(self.black), _u64(self.white), self.X, self.Y, self.XSL, self.YSL, ncorr, out)
        return [int(v) for v in out]

    def dump_rows(self, row0: int, nrows: int) -> bytes:
        buf = C.create_string_buffer(nrows * (self.X + 1))
        n = lib().orc_dump_rows(_u64(self.black), _u64(self.white), self.X, row0, nrows, buf)
        return buf.raw[:n]


class BasicCpuIsing:
    """Byte-per-spin baseline (basic_python/ising_basic.py algorithm) on host cores."""

    def __init__(self, n: int, m: int, alpha: float = 1.0, seed: int = 1234):
        self.n, self.m, self.alpha, self.seed = n, m, alpha, seed
        self.black = np.empty((n, m // 2), dtype=np.int8)
        self.white = np.empty((n, m // 2), dtype=np.int8)
        self._scratch = np.empty(n * (m // 2), dtype=np.float32)
        self.it = 0
        p8 = C.POINTER(C.c_int8)
        lib().basic_init(self.black.ctypes.data_as(p8), self.white.ctypes.data_as(p8), n, m, C.c_uint64(seed),
                         self._scratch.ctypes.data_as(C.POINTER(C.c_float)))

    def sweeps(self, k: int):
        p8 = C.POINTER(C.c_int8)
        lib().basic_sweeps(self.black.ctypes.data_as(p8), self.white.ctypes.data_as(p8), self.n, self.m,
                           C.c_float(self.alpha), C.c_uint64(self.seed), self.it, k,
                           self._scratch.ctypes.data_as(C.POINTER(C.c_float)))
        self.it += k

    def observables(self):
        p8 = C.POINTER(C.c_int8)
        ms, b = C.c_int64(), C.c_int64()
        lib().basic_observables(self.black.ctypes.data_as(p8), self.white.ctypes.data_as(p8), self.n, self.m,
                                C.byref(ms), C.byref(b))
        nm = self.n * self.m
        return ms.value / nm, -b.value / nm  # magnetisation per spin, energy per spin


def set_threads(n: int):
    lib().orc_set_threads(int(n))


def max_threads() -> int:
    return int(lib().orc_max_threads())


class OracleSlab:
    """One slab of a ring on the host (the CPU counterpart of one device in optimized/main.cu's -d N runs).

    Holds Y rows of both colours starting at global row slab*Y plus one received halo row above and below per
    colour.  update_color() restricted to a row range restates a partial launch of spinUpdateV_2D_k.
    """

    def __init__(self, X: int, Y: int, seed: int, temp: float, nslabs: int, slab: int):
        self.X, self.Y, self.seed, self.temp = X, Y, seed, float(np.float32(temp))
        self.nslabs, self.slab = nslabs, slab
        self.lld = X // 32
        self.lat = np.zeros((2, Y, self.lld), dtype=np.uint64)
        self.halo = np.zeros((2, 2, self.lld), dtype=np.uint64)  # [colour][top,bot]

    def init(self):
        rc = lib().orc_init_slab(_u64(self.lat[0]), _u64(self.lat[1]), self.X, self.Y, self.slab * self.Y,
                                 C.c_uint64(self.seed))
        assert rc == 0

    def update_rows(self, it: int, color: int, r_lo: int, r_hi: int):
        tab = (C.c_float * 10)()
        lib().orc_exp_table(C.c_float(self.temp), tab)
        other = 1 - color
        rc = lib().orc_update_color_slab(_u64(self.lat[color]), _u64(self.lat[other]), _u64(self.halo[other, 0]),
                                         _u64(self.halo[other, 1]), self.X, self.Y, self.slab * self.Y,
                                         C.c_uint64(self.seed), it, color, tab, r_lo, r_hi)
        assert rc == 0, rc

    def count_up(self) -> int:
        return int(sum(int(np.unpackbits(self.lat[c].view(np.uint8)).sum(dtype=np.int64)) for c in (0, 1)))

    def bond_equal(self) -> int:
        return int(lib().orc_bond_equal_slab(_u64(self.lat[0]), _u64(self.lat[1]), _u64(self.halo[1, 0]),
                                             _u64(self.halo[1, 1]), self.X, self.Y, self.slab * self.Y))


class OracleGhostSlab(OracleSlab):
    """OracleSlab with G ghost rows on either side (the CPU counterpart of a ring slab of the product's ballot layout,
    csrc/ising_ring.cpp: sweep_deep): rows [-G, Y+G) of both colours in one array, rows -G..-1 / Y..Y+G-1 being copies of the
    neighbouring slabs' rows.  sweep_ghost() updates rows [-(G-1), Y+G-1) level by level -- the ghost rows with the draws
    their owners make (global row, around the ring) -- which leaves the slab's own rows exact for up to G levels."""

    def __init__(self, X: int, Y: int, seed: int, temp: float, nslabs: int, slab: int, ghost: int):
        super().__init__(X, Y, seed, temp, nslabs, slab)
        self.G = ghost
        self.ext = np.zeros((2, Y + 2 * ghost, self.lld), dtype=np.uint64)
        self.lat = self.ext[:, ghost:ghost + Y]  # the slab's own rows, in place

    def _level(self, it: int, color: int):
        tab = (C.c_float * 10)()
        lib().orc_exp_table(C.c_float(self.temp), tab)
        total, G, n = self.nslabs * self.Y, self.G, self.Y + 2 * self.G
        other = 1 - color
        e = 1
        while e < n - 1:  # blocks of consecutive global rows (the ring closes inside the ghost rows of the end slabs)
            g0 = (self.slab * self.Y + e - G) % total
            rows = min(n - 1 - e, total - g0)
            # rows [e, e + rows) as the first rows of a "slab" that starts at global row g0 (the oracle wants a row count that
            # is a multiple of 16; only [0, rows) of it is updated, which reads rows e - 1 .. e + rows)
            k16 = -(-rows // 16) * 16
            rc = lib().orc_update_color_slab(_u64(self.ext[color, e:]), _u64(self.ext[other, e:]), _u64(self.ext[other, e - 1]),
                                             _u64(self.ext[other, e + rows]), self.X, k16, g0, C.c_uint64(self.seed), it, color,
                                             tab, 0, rows)
            assert rc == 0, rc
            e += rows

    def sweep_ghost(self, first_it: int, nsweeps: int):
        assert 2 * nsweeps <= self.G
        for level in range(2 * nsweeps):
            self._level(first_it + level // 2, level & 1)

