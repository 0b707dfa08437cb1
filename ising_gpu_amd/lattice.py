"""Host-side mirror of the reference's driver operations for one slab (optimized/main.cu main(), :1230-1926).

`IsingSlab` wraps one `ising_ctx` of libising_hip.so.  Method names follow the reference's steps: init
(latticeInit_k launches :1708-1726), update_color / sweep (spinUpdateV_2D_k launches :1763-1805), count
(countSpins :831-868), set_temperature (:1848-1859), dump (dumpLattice :1140-1209).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import BLACK, WHITE, IsingConfig, IsingError, check


def device_count() -> int:
    n = C.c_int(0)
    lib = _lib.load()
    rc = lib.ising_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class IsingSlab:
    """One slab (rows [slab*Y, (slab+1)*Y) of an (nslabs*Y) x X periodic lattice) resident on one GPU."""

    def __init__(self, X: int, Y: int, seed: int = _lib.SEED_DEF, temp: float = 0.1 * _lib.CRIT_TEMP_F32,
                 nslabs: int = 1, slab: int = 0, device: int = 0, strip_rows: int = 0, kernel: int = _lib.KERNEL_AUTO,
                 XSL: int = 0, YSL: int = 0, J_prob: float | None = None, lattice_mem: int = 0, coupling_mem: int = 0,
                 layout: int = _lib.LAYOUT_AUTO, ring_halo: bool = False, lattice_mem_bytes: int = 0,
                 coupling_mem_bytes: int = 0):
        self._lib = _lib.load()
        self.cfg = IsingConfig(X=X, Y=Y, nslabs=nslabs, slab=slab, seed=seed, temp=float(np.float32(temp)),
                               device=device, strip_rows=strip_rows, kernel=kernel, XSL=XSL, YSL=YSL,
                               lattice_mem=lattice_mem or None, coupling_mem=coupling_mem or None, layout=layout,
                               use_J=0 if J_prob is None else 1, J_prob=0.0 if J_prob is None else float(J_prob),
                               ring_halo=1 if ring_halo else 0, lattice_mem_bytes=lattice_mem_bytes,
                               coupling_mem_bytes=coupling_mem_bytes)
        self.use_J = J_prob is not None
        self._h = C.c_void_p()
        check(self._lib.ising_create(C.byref(self.cfg), C.byref(self._h)))
        self.X, self.Y, self.nslabs, self.slab = X, Y, nslabs, slab
        self.lld = X // 32
        self.it = 0  # completed sweeps (the next sweep uses the reference's it = self.it + 1)
        sr, ns = C.c_int(), C.c_int()
        check(self._lib.ising_strip_info(self._h, C.byref(sr), C.byref(ns)))
        self.strip_rows, self.nstrips = sr.value, ns.value
        lay = C.c_int()
        check(self._lib.ising_layout(self._h, C.byref(lay)))
        self.layout = lay.value

    # -- lifetime ----------------------------------------------------------------------------------------
    def close(self):
        if self._h:
            self._lib.ising_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- driver steps ------------------------------------------------------------------------------------
    def set_stream(self, hip_stream: int):
        check(self._lib.ising_set_stream(self._h, C.c_void_p(hip_stream)))

    def measure_enqueue(self):
        """Enqueue (up count, bond sum) of the state at this point of the stream; nothing waits (measure_fetch does)."""
        check(self._lib.ising_measure_enqueue(self._h))
        return self

    def measure_fetch(self):
        """Wait for the stream; [(up, down, bond_equal), ...] of all measurements enqueued since the last fetch."""
        cap = 4096
        up, bond, n = (C.c_uint64 * cap)(), (C.c_int64 * cap)(), C.c_int()
        check(self._lib.ising_measure_fetch(self._h, up, bond, cap, C.byref(n)))
        tot = self.X * self.Y
        return [(int(up[i]), tot - int(up[i]), int(bond[i])) for i in range(n.value)]

    def use_private_stream(self):
        """A non-blocking stream of this slab's own: independent slabs on one GPU then run side by side."""
        check(self._lib.ising_use_private_stream(self._h))
        return self

    def synchronize(self):
        check(self._lib.ising_synchronize(self._h))

    def init(self):
        check(self._lib.ising_init_lattice(self._h))
        self.it = 0
        return self

    def init_couplings(self):
        """-J: hamiltInitB_k (seed+1) + hamiltInitW_k (single slab, or sub-lattices)."""
        check(self._lib.ising_init_couplings(self._h))
        return self

    def init_couplings_black(self):
        check(self._lib.ising_init_couplings_black(self._h))

    def init_couplings_white(self):
        check(self._lib.ising_init_couplings_white(self._h))

    def read_couplings(self, which: int) -> np.ndarray:
        out = np.empty((self.Y, self.lld), dtype=np.uint64)
        check(self._lib.ising_read_couplings(self._h, which, 0, self.Y, out.ctypes.data_as(C.c_void_p)))
        return out

    def write_couplings(self, which: int, words: np.ndarray):
        """A whole coupling array in the reference's form: (Y, X/32) uint64, one nibble per site, bits <up, down, left, right>."""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        assert words.shape == (self.Y, self.lld), words.shape
        check(self._lib.ising_write_couplings(self._h, which, words.ctypes.data_as(C.c_void_p)))

    def swap_couplings(self):
        """The two coupling arrays change places: every colour's update reads its own sites' bonds (symmetric +-J model)."""
        check(self._lib.ising_swap_couplings(self._h))
        return self

    def current_layout(self) -> int:
        """Device layout right now (a ballot slab turns dense when a temperature has no integer thresholds)."""
        lay = C.c_int()
        check(self._lib.ising_layout(self._h, C.byref(lay)))
        self.layout = lay.value
        return self.layout

    def set_temperature(self, temp: float):
        check(self._lib.ising_set_temperature(self._h, C.c_float(float(np.float32(temp)))))

    def tables(self):
        tab = (C.c_float * 10)()
        thr = (C.c_uint64 * 5)()
        check(self._lib.ising_get_tables(self._h, tab, thr))
        return np.array(list(tab), dtype=np.float32).reshape(2, 5), [int(v) for v in thr]

    def update_color(self, it: int, color: int, row_lo: int = 0, row_hi: int | None = None):
        check(self._lib.ising_update_color(self._h, it, color, row_lo, self.Y if row_hi is None else row_hi))

    def update_edges(self, it: int, color: int):
        check(self._lib.ising_update_edges(self._h, it, color))

    def sweep(self, n: int = 1):
        check(self._lib.ising_sweep(self._h, self.it + 1, n))
        self.it += n
        return self

    def sweep_counted(self, n: int, every: int, energy: bool = False):
        """n sweeps with the up-spin count after every iteration that is a multiple of `every` (ising_sweep_counted: inside the fused
        launches where there are any); returns [(up, down), ...] -- with `energy` [(up, down, bond_equal), ...]: the bond sum at the same points."""
        cap = n // every + 2
        ups, k = (C.c_uint64 * cap)(), C.c_int()
        eqs = (C.c_int64 * cap)() if energy else None
        check(self._lib.ising_sweep_counted(self._h, self.it + 1, n, every, ups, eqs, cap, C.byref(k)))
        self.it += n
        tot = self.X * self.Y
        return [(int(ups[i]), tot - int(ups[i])) + ((int(eqs[i]),) if energy else ()) for i in range(k.value)]

    def kernel_clock(self, enable: bool = True):
        """measurement aid (ising_kernel_clock): fused launches leave the marks their clock is computed from"""
        check(self._lib.ising_kernel_clock(self._h, 1 if enable else 0))
        return self

    def kernel_clock_fetch(self):
        """(mean, min, max) MHz over the XCDs of the last fused launch (ising_kernel_clock_fetch; blocking)"""
        m, lo, hi = C.c_double(), C.c_double(), C.c_double()
        check(self._lib.ising_kernel_clock_fetch(self._h, C.byref(m), C.byref(lo), C.byref(hi)))
        return m.value, lo.value, hi.value

    @property
    def fused(self) -> bool:
        """True when sweep() issues fused launches (several colour half-sweeps per launch, ising_sweep_info)."""
        f, m = C.c_int(), C.c_int()
        check(self._lib.ising_sweep_info(self._h, C.byref(f), C.byref(m)))
        return f.value in (1, 3)

    @property
    def split(self) -> bool:
        """True when sweep()'s LONG calls (2^35 flips and more; ISING_SPLIT=1: all) take the split form (draw units and word units with tickets of their own,
        ising_sweep_info); sweep_form(n) answers for a call of n sweeps."""
        f, m = C.c_int(), C.c_int()
        check(self._lib.ising_sweep_info(self._h, C.byref(f), C.byref(m)))
        return f.value == 3

    def sweep_form(self, nsweeps: int):
        """(form, strip rows, workgroups per CU) of the launches sweep(nsweeps) issues (ising_sweep_form; form as ising_sweep_info, 3 only when THIS call runs split launches)."""
        v = [C.c_int() for _ in range(3)]
        check(self._lib.ising_sweep_form(self._h, nsweeps, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def guard_info(self) -> dict:
        """The run-time guard under the fused launches' shape table (ising_shape_guard_info): state (0 off, 1 timing the table's shape, 2 trying neighbours,
        3 settled), whether it switched, the table's shape and rate, the shape and rate that stayed.  Blocks while a timed launch is in flight."""
        class _G(C.Structure):
            _fields_ = [(k, C.c_int32) for k in ("state", "switched", "launches_timed", "table_strip_rows", "table_wg_per_cu", "strip_rows", "wg_per_cu")] + \
                       [(k, C.c_float) for k in ("expected_flips_per_ns", "table_flips_per_ns", "kept_flips_per_ns")] + \
                       [("form_state", C.c_int32), ("split_kept", C.c_int32), ("split_flips_per_ns", C.c_float), ("fused_flips_per_ns", C.c_float)]
        g = _G()
        check(self._lib.ising_shape_guard_info(self._h, C.byref(g)))
        return {k: getattr(g, k) for k, _ in _G._fields_}

    @property
    def tiled(self) -> bool:
        """True when sweep() issues tile launches (small lattices on the dense layout: several sweeps per launch, every workgroup on a
        tile + halo of its own, ising_sweep_info)."""
        f, m = C.c_int(), C.c_int()
        check(self._lib.ising_sweep_info(self._h, C.byref(f), C.byref(m)))
        return f.value == 2

    @property
    def quad(self) -> bool:
        """True when sweep() runs on the quad layout (small lattices: one launch per pass of several sweeps = the word pass on tiles + halo next to the
        drawing workgroups that make the accept masks of the pass to come, one stream; ising_quad.hip, ising_sweep_info)."""
        f, m = C.c_int(), C.c_int()
        check(self._lib.ising_sweep_info(self._h, C.byref(f), C.byref(m)))
        return f.value == 4

    @property
    def max_sweeps_per_launch(self) -> int:
        """Sweeps one fused launch carries at most (0: one launch per colour); for a ring slab with ghost rows: between
        two exchanges of the ring."""
        f, m = C.c_int(), C.c_int()
        check(self._lib.ising_sweep_info(self._h, C.byref(f), C.byref(m)))
        return int(m.value)

    def sweep_timed(self, n: int) -> float:
        ms = C.c_float()
        check(self._lib.ising_sweep_timed(self._h, self.it + 1, n, C.byref(ms)))
        self.it += n
        return ms.value

    def count(self):
        up, dw = C.c_uint64(), C.c_uint64()
        check(self._lib.ising_count(self._h, C.byref(up), C.byref(dw)))
        return int(up.value), int(dw.value)

    def bond_equal(self) -> int:
        a = C.c_int64()
        check(self._lib.ising_bond_equal(self._h, C.byref(a)))
        return int(a.value)

    def correlations(self, ncorr: int = 128):
        """Exact two-point sums for distances 1..ncorr (getCorr2D_k); divide by 2*X*Y for the reference's output."""
        out = (C.c_int64 * ncorr)()
        check(self._lib.ising_correlations(self._h, ncorr, out))
        return [int(v) for v in out]

    def halo_ptrs(self, color: int):
        p = [C.c_void_p() for _ in range(4)]
        nb = C.c_size_t()
        check(self._lib.ising_halo_ptrs(self._h, color, *[C.byref(x) for x in p], C.byref(nb)))
        return [x.value for x in p], nb.value

    def ghost_ptrs(self, color: int):
        """-> (depth G, [send_top, send_bot, recv_top, recv_bot], bytes per block): the deep exchange surface
        (ising_ghost_ptrs); G == 1 when the slab keeps no ghost rows."""
        p = [C.c_void_p() for _ in range(4)]
        nb, depth = C.c_size_t(), C.c_int()
        check(self._lib.ising_ghost_ptrs(self._h, color, C.byref(depth), *[C.byref(x) for x in p], C.byref(nb)))
        return depth.value, [x.value for x in p], nb.value

    def ghost_delivered(self, color: int):
        check(self._lib.ising_ghost_delivered(self._h, color))

    def sweep_ghost(self, n: int = 1):
        """n <= G/2 sweeps as one fused launch over the slab and its ghost rows (both colours delivered first)."""
        check(self._lib.ising_sweep_ghost(self._h, self.it + 1, n))
        self.it += n
        return self

    def launch_shape(self):
        """Test aid (ising_debug_launch_shape): (strip rows, workgroups per CU, split lead) of this slab's fused launches."""
        v = [C.c_int() for _ in range(3)]
        check(self._lib.ising_debug_launch_shape(self._h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def debug_fault(self, what: int = 1, arg: int = 0):
        """Test aid (ising_debug_fault): leave the host's record of the completion counters out of step with the device."""
        check(self._lib.ising_debug_fault(self._h, what, arg))

    def device_ptr(self, color: int):
        p, nb = C.c_void_p(), C.c_size_t()
        check(self._lib.ising_device_ptr(self._h, color, C.byref(p), C.byref(nb)))
        return p.value, nb.value

    def read(self, color: int, row0: int = 0, nrows: int | None = None) -> np.ndarray:
        nrows = self.Y - row0 if nrows is None else nrows
        out = np.empty((nrows, self.lld), dtype=np.uint64)
        check(self._lib.ising_read_packed(self._h, color, row0, nrows, out.ctypes.data_as(C.c_void_p)))
        return out

    def write(self, color: int, rows: np.ndarray, row0: int = 0):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        check(self._lib.ising_write_packed(self._h, color, row0, rows.shape[0], rows.ctypes.data_as(C.c_void_p)))

    def read_bits(self, color: int, row0: int = 0, nrows: int | None = None) -> np.ndarray:
        """Rows at 1 bit per spin: X/64 uint32 per row, one word per reference 128-bit vector (ising_read_bits)."""
        nrows = self.Y - row0 if nrows is None else nrows
        out = np.empty((nrows, self.X // 64), dtype=np.uint32)
        check(self._lib.ising_read_bits(self._h, color, row0, nrows, out.ctypes.data_as(C.c_void_p)))
        return out

    def write_bits(self, color: int, rows: np.ndarray, row0: int = 0):
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        check(self._lib.ising_write_bits(self._h, color, row0, rows.shape[0], rows.ctypes.data_as(C.c_void_p)))

    def dump(self, prefix: str):
        check(self._lib.ising_dump_text(self._h, prefix.encode()))

    # -- one slab per process: the ring over RCCL inside the library (ising_rank_*) -------------------------
    def rank_attach(self, unique_id: bytes):
        """ncclCommInitRank(nslabs, id, slab): collective over all ranks of the ring."""
        if len(unique_id) != _lib.RCCL_ID_BYTES:
            raise ValueError("the RCCL unique id has %d bytes" % _lib.RCCL_ID_BYTES)
        buf = C.create_string_buffer(bytes(unique_id), _lib.RCCL_ID_BYTES)
        check(self._lib.ising_rank_attach(self._h, buf))

    def ipc_export(self) -> bytes:
        """This rank's blob for the IPC transport (ising_ipc_export): hipIpcMemHandles of its arrays + its flag segment."""
        buf = C.create_string_buffer(_lib.IPC_BLOB_BYTES)
        check(self._lib.ising_ipc_export(self._h, buf))
        return buf.raw

    def ipc_attach(self, blobs):
        """Every rank's blob in slab order (ising_ipc_attach): maps the neighbours' rows; the rank_* calls then run on them."""
        blobs = list(blobs)
        if any(len(b) != _lib.IPC_BLOB_BYTES for b in blobs):
            raise ValueError("an IPC blob has %d bytes" % _lib.IPC_BLOB_BYTES)
        buf = C.create_string_buffer(b"".join(bytes(b) for b in blobs), _lib.IPC_BLOB_BYTES * len(blobs))
        check(self._lib.ising_ipc_attach(self._h, buf, len(blobs)))

    def rank_detach(self, abort: bool = False):
        check(self._lib.ising_rank_detach(self._h, 1 if abort else 0))

    def rank_exchange(self, color: int):
        check(self._lib.ising_rank_exchange(self._h, color))

    def rank_init_couplings(self):
        check(self._lib.ising_rank_init_couplings(self._h))

    def rank_sweep(self, n: int = 1):
        check(self._lib.ising_rank_sweep(self._h, self.it + 1, n))
        self.it += n
        return self

    def rank_sweep_counted(self, n: int, every: int, energy: bool = False):
        """rank_sweep with the whole lattice's up-spin count after every iteration that is a multiple of `every` (ising_rank_sweep_counted: inside
        the deep launches where the ring sweeps through ghost rows; collective); returns [(up, down), ...], with `energy` [(up, down, bond_equal), ...]."""
        cap = n // every + 2
        ups, k = (C.c_uint64 * cap)(), C.c_int()
        eqs = (C.c_int64 * cap)() if energy else None
        check(self._lib.ising_rank_sweep_counted(self._h, self.it + 1, n, every, ups, eqs, cap, C.byref(k)))
        self.it += n
        tot = self.X * self.Y * self.cfg.nslabs
        return [(int(ups[i]), tot - int(ups[i])) + ((int(eqs[i]),) if energy else ()) for i in range(k.value)]

    def rank_wait(self, timeout_ms: int = -1):
        check(self._lib.ising_rank_wait(self._h, timeout_ms))

    def rank_count(self):
        up, dw = C.c_uint64(), C.c_uint64()
        check(self._lib.ising_rank_count(self._h, C.byref(up), C.byref(dw)))
        return int(up.value), int(dw.value)

    def exchange_stats_begin(self, max_exchanges: int = 64):
        """Sample the next deep exchanges of this ring slab (ising_exchange_stats_begin)."""
        check(self._lib.ising_exchange_stats_begin(self._h, max_exchanges))

    def exchange_stats_fetch(self) -> dict:
        """Wait for the slab's streams; {exchanges, launch_ms_mean, ..., gap_ms_max} over the sampled exchanges."""
        st = _lib.ExchangeStats()
        check(self._lib.ising_exchange_stats_fetch(self._h, C.byref(st)))
        return {n: (int(getattr(st, n)) if n in ("exchanges", "launches") else round(float(getattr(st, n)), 4)) for n, _ in st._fields_}

    def rank_checkpoint_save(self, path: str):
        check(self._lib.ising_rank_checkpoint_save(self._h, str(path).encode(), self.it))

    def rank_checkpoint_load(self, path: str) -> int:
        it = C.c_int64()
        check(self._lib.ising_rank_checkpoint_load(self._h, str(path).encode(), C.byref(it)))
        self.it = int(it.value)
        return self.it

    def rank_bond_equal(self) -> int:
        a = C.c_int64()
        check(self._lib.ising_rank_bond_equal(self._h, C.byref(a)))
        return int(a.value)


class IsingBatch:
    """Independent lattices of one shape on one GPU advancing together (ising_batch_*): one fused launch carries a level of
    every member, one more launch measures all of them.  The members stay ordinary IsingSlab objects."""

    def __init__(self, slabs):
        self.slabs = list(slabs)
        self._lib = _lib.load()
        arr = (C.c_void_p * len(self.slabs))(*[s._h for s in self.slabs])
        self._h = C.c_void_p()
        check(self._lib.ising_batch_create(arr, len(self.slabs), C.byref(self._h)))
        self.it = 0
        h, w, n = C.c_int(), C.c_int(), C.c_int()
        check(self._lib.ising_batch_info(self._h, C.byref(h), C.byref(w), C.byref(n)))
        self.strip_rows, self.wg_per_cu, self.n = h.value, w.value, n.value
        c, t, wv = C.c_int(), C.c_int(), C.c_int()
        check(self._lib.ising_batch_quad_info(self._h, C.byref(c), C.byref(t), C.byref(wv)))
        #: a batch of quad-path lattices: (row groups per tile, sweeps per pass, waves per workgroup); None: a ballot batch
        self.quad_shape = (c.value, t.value, wv.value) if c.value else None

    def close(self):
        if self._h:
            self._lib.ising_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def init(self):
        for s in self.slabs:
            s.init()
        self.it = 0
        return self

    def sweep(self, n: int = 1):
        check(self._lib.ising_batch_sweep(self._h, self.it + 1, n))
        self.it += n
        for s in self.slabs:
            s.it = self.it
        return self

    def sweep_counted(self, n: int, every: int, energy: bool = False):
        """`n` sweeps of every member with the reference's print points (iterations that are multiples of `every`):
        [[(up, down, bond_equal or None) per member] per point] (ising_batch_sweep_counted)."""
        cap = max(1, n // every + 1)
        up, bond, k = (C.c_uint64 * (cap * self.n))(), (C.c_int64 * (cap * self.n))(), C.c_int()
        check(self._lib.ising_batch_sweep_counted(self._h, self.it + 1, n, every, up, bond if energy else None, cap, C.byref(k)))
        self.it += n
        for s in self.slabs:
            s.it = self.it
        tot = self.slabs[0].X * self.slabs[0].Y
        return [[(int(up[i * self.n + r]), tot - int(up[i * self.n + r]), int(bond[i * self.n + r]) if energy else None) for r in range(self.n)]
                for i in range(k.value)]

    def measure_enqueue(self):
        check(self._lib.ising_batch_measure_enqueue(self._h))
        return self

    def debug_fault(self, polls: int = 0):
        """Test aid (ising_batch_debug_fault): the batch's completion counters out of step with the device."""
        check(self._lib.ising_batch_debug_fault(self._h, polls))

    def measure_fetch(self):
        """[[(up, down, bond_equal) per member] per measurement], in enqueue order."""
        cap = 1024
        up, bond, k = (C.c_uint64 * (cap * self.n))(), (C.c_int64 * (cap * self.n))(), C.c_int()
        check(self._lib.ising_batch_measure_fetch(self._h, up, bond, cap, C.byref(k)))
        tot = self.slabs[0].X * self.slabs[0].Y
        return [[(int(up[i * self.n + r]), tot - int(up[i * self.n + r]), int(bond[i * self.n + r])) for r in range(self.n)] for i in range(k.value)]


def philox_ceiling(device: int = 0) -> float:
    """sites/ns of a draw-only kernel (one Philox4x32-10 output per site, nothing else): the update kernels' VALU ceiling."""
    v = C.c_double()
    check(_lib.load().ising_philox_ceiling(device, C.byref(v)))
    return v.value


def philox_ceiling_clocked(device: int = 0, min_ms: float = 25.0):
    """(sites/ns, shader MHz) of the draw-only kernel averaged over launches that last min_ms (ising_philox_ceiling_clocked)."""
    v, mhz = C.c_double(), C.c_double()
    check(_lib.load().ising_philox_ceiling_clocked(device, float(min_ms), C.byref(v), C.byref(mhz)))
    return v.value, mhz.value


def rccl_version() -> int:
    """Version code of the RCCL the library opened at run time; raises IsingError when there is none."""
    v = C.c_int()
    check(_lib.load().ising_rccl_available(C.byref(v)))
    return v.value


def rccl_unique_id() -> bytes:
    buf = C.create_string_buffer(_lib.RCCL_ID_BYTES)
    check(_lib.load().ising_rccl_unique_id(buf))
    return buf.raw


def required_bytes(X: int, Y: int, layout: int | None = None) -> int:
    """Size of a caller-owned device buffer: the coupling arrays (layout None: 4 bit per site in every layout) or the
    spin arrays of a given device layout (a quarter of that for the 1 bit/spin layouts)."""
    if layout is None:
        return int(_lib.load().ising_required_bytes(X, Y))
    return int(_lib.load().ising_required_bytes_layout(X, Y, layout))


class SlabSet:
    """All slabs of a ring in ONE process (the reference's own process model, optimized/main.cu:1763-1805), driven through
    the ising_ring_* entry points: edge rows, halo delivery on the slabs' comm streams (RCCL between devices, copies
    on one device), interior rows."""

    def __init__(self, slabs):
        self.slabs = list(slabs)
        self._lib = _lib.load()
        self._arr = (C.c_void_p * len(self.slabs))(*[s._h for s in self.slabs])
        self.n = len(self.slabs)
        self.it = 0

    def set_transport(self, transport: int):
        check(self._lib.ising_ring_set_transport(self._arr, self.n, transport))
        return self

    @property
    def transport(self) -> int:
        t = C.c_int()
        check(self._lib.ising_ring_transport(self._arr, self.n, C.byref(t)))
        return t.value

    def init(self):
        for s in self.slabs:
            s.init()
        self.it = 0
        for color in (BLACK, WHITE):
            check(self._lib.ising_ring_exchange(self._arr, self.n, color))
        if self.slabs[0].use_J:
            check(self._lib.ising_ring_init_couplings(self._arr, self.n))
        return self

    def exchange(self):
        for color in (BLACK, WHITE):
            check(self._lib.ising_ring_exchange(self._arr, self.n, color))
        return self

    def sweep_counted(self, n: int, every: int, energy: bool = False):
        """sweep with the whole lattice's up-spin count after every iteration that is a multiple of `every` (ising_ring_sweep_counted); with
        `energy` the bond sum too: [(up, down, bond_equal), ...]."""
        cap = n // every + 2
        ups, k = (C.c_uint64 * cap)(), C.c_int()
        eqs = (C.c_int64 * cap)() if energy else None
        check(self._lib.ising_ring_sweep_counted(self._arr, self.n, self.it + 1, n, every, ups, eqs, cap, C.byref(k)))
        self.it += n
        for s in self.slabs:
            s.it = self.it
        tot = sum(s.X * s.Y for s in self.slabs)
        return [(int(ups[i]), tot - int(ups[i])) + ((int(eqs[i]),) if energy else ()) for i in range(k.value)]

    def sweep(self, n: int = 1):
        check(self._lib.ising_ring_sweep(self._arr, self.n, self.it + 1, n))
        self.it += n
        for s in self.slabs:
            s.it = self.it
        return self

    def synchronize(self):
        check(self._lib.ising_ring_synchronize(self._arr, self.n))

    def count(self):
        up, dw = C.c_uint64(), C.c_uint64()
        check(self._lib.ising_ring_count(self._arr, self.n, C.byref(up), C.byref(dw)))
        return int(up.value), int(dw.value)

    def bond_equal(self) -> int:
        a = C.c_int64()
        check(self._lib.ising_ring_bond_equal(self._arr, self.n, C.byref(a)))
        return int(a.value)

    def set_temperature(self, temp: float):
        for s in self.slabs:
            s.set_temperature(temp)

    def checkpoint_save(self, path: str):
        check(self._lib.ising_ring_checkpoint_save(self._arr, self.n, str(path).encode(), self.it))

    def checkpoint_load(self, path: str, apply_temperature: bool = True):
        """Loads the spins; the run continues at the temperature the checkpoint was written at (apply_temperature=False keeps
        the slabs' own), so a resume cannot silently go on at another T."""
        it = C.c_int64()
        check(self._lib.ising_ring_checkpoint_load(self._arr, self.n, str(path).encode(), C.byref(it)))
        if apply_temperature:
            self.set_temperature(checkpoint_info(path)["temp"])
        self.it = int(it.value)
        for s in self.slabs:
            s.it = self.it
        return self.exchange()

    def close(self):
        for s in self.slabs:
            s.close()


class _CheckpointInfo(C.Structure):
    _fields_ = [("X", C.c_int32), ("Y_total", C.c_int32), ("nslabs_written", C.c_int32), ("XSL", C.c_int32), ("YSL", C.c_int32),
                ("use_J", C.c_int32), ("temp", C.c_float), ("J_prob", C.c_float), ("seed", C.c_uint64), ("it", C.c_int64)]


def checkpoint_info(path: str) -> dict:
    """Header of a binary checkpoint (ising_checkpoint_info_read): geometry, seed, completed sweeps, temperature, settings."""
    info = _CheckpointInfo()
    check(_lib.load().ising_checkpoint_info_read(str(path).encode(), C.byref(info)))
    return {name: getattr(info, name) for name, _ in _CheckpointInfo._fields_}


def magnetization(up: int, down: int) -> float:
    """|up - down| / N as the reference prints it (optimized/main.cu:1748)."""
    return abs(float(up) - float(down)) / float(up + down)


def energy_per_spin(bond_equal_total: int, nspins: int) -> float:
    """E/N = -(2A - 2N)/N with A the black-site aligned-neighbour count over the whole lattice."""
    return -(2 * bond_equal_total - 2 * nspins) / nspins


def ring_correlations(slabs, ncorr: int = 128):
    """Totals over all slabs of a single-process ring (ising_ring_correlations)."""
    lib = _lib.load()
    arr = (C.c_void_p * len(slabs))(*[s._h for s in slabs])
    out = (C.c_int64 * ncorr)()
    check(lib.ising_ring_correlations(arr, len(slabs), ncorr, out))
    return [int(v) for v in out]
