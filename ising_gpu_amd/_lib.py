"""ctypes binding of libising_hip.so (the C-ABI declared in include/ising_hip.h).

The shared library is the product; this module only declares prototypes.  There is NO CPU fallback: if the
library is missing or cannot be loaded the import of any compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ISING_LIB", os.path.join(_HERE, "libising_hip.so"))  # ISING_LIB: perf-investigation builds only

BLACK, WHITE = 0, 1
HAM_BLACK = 2  # the black coupling array as a third plane for halo exchange
CRIT_TEMP_F32 = 2.2691853046417236  # float32(2.26918531421f), CRIT_TEMP optimized/main.cu:42
SEED_DEF = 463463564571  # optimized/main.cu:63
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_FAST = 0, 1, 2
LAYOUT_AUTO, LAYOUT_NIBBLE, LAYOUT_DENSE, LAYOUT_BALLOT = 0, 1, 2, 3
TRANSPORT_AUTO, TRANSPORT_COPY, TRANSPORT_RCCL, TRANSPORT_IPC = 0, 1, 2, 3
E_ARG, E_HIP, E_STATE, E_NOGPU, E_RCCL, E_TIMEOUT, E_IO = 1, 2, 3, 4, 5, 6, 7
RCCL_ID_BYTES = 128
IPC_BLOB_BYTES = 256


class IsingConfig(C.Structure):
    _fields_ = [
        ("X", C.c_int32), ("Y", C.c_int32), ("nslabs", C.c_int32), ("slab", C.c_int32),
        ("seed", C.c_uint64), ("temp", C.c_float), ("device", C.c_int32),
        ("strip_rows", C.c_int32), ("kernel", C.c_int32), ("XSL", C.c_int32), ("YSL", C.c_int32),
        ("lattice_mem", C.c_void_p), ("coupling_mem", C.c_void_p),
        ("layout", C.c_int32), ("use_J", C.c_int32), ("J_prob", C.c_float), ("ring_halo", C.c_int32),
        ("lattice_mem_bytes", C.c_size_t), ("coupling_mem_bytes", C.c_size_t),
    ]


class ExchangeStats(C.Structure):
    """ising_exchange_stats (include/ising_hip.h): where a ring slab's time goes around its exchanges, in milliseconds."""
    _fields_ = [("exchanges", C.c_int32)] + [(n, C.c_float) for n in (
        "launch_ms_mean", "launch_ms_max", "exchange_ms_mean", "exchange_ms_max",
        "go_after_end_ms_mean", "go_after_end_ms_max", "gap_ms_mean", "gap_ms_max")] + [("launches", C.c_int32)]


class IsingError(RuntimeError):
    def __init__(self, msg, code=0):
        super().__init__(msg)
        self.code = code


_lib = None

# name -> (restype, argtypes); every symbol include/ising_hip.h and include/ising_hip_testing.h declare
PROTOTYPES = {
    "ising_last_error": (C.c_char_p, []),
    "ising_switch_table": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ising_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "ising_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ising_device_peer_access": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "ising_philox_ceiling": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "ising_philox_ceiling_clocked": (C.c_int, [C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ising_kernel_clock": (C.c_int, [C.c_void_p, C.c_int]),
    "ising_kernel_clock_fetch": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ising_required_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "ising_required_bytes_layout": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "ising_create": (C.c_int, [C.POINTER(IsingConfig), C.POINTER(C.c_void_p)]),
    "ising_destroy": (C.c_int, [C.c_void_p]),
    "ising_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ising_synchronize": (C.c_int, [C.c_void_p]),
    "ising_init_lattice": (C.c_int, [C.c_void_p]),
    "ising_init_couplings": (C.c_int, [C.c_void_p]),
    "ising_init_couplings_black": (C.c_int, [C.c_void_p]),
    "ising_init_couplings_white": (C.c_int, [C.c_void_p]),
    "ising_read_couplings": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "ising_write_couplings": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ising_swap_couplings": (C.c_int, [C.c_void_p]),
    "ising_ring_init_couplings": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "ising_set_temperature": (C.c_int, [C.c_void_p, C.c_float]),
    "ising_get_tables": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]),
    "ising_update_color": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ising_update_edges": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ising_strip_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ising_sweep": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ising_sweep_counted": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int)]),
    "ising_ring_sweep_counted": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int)]),
    "ising_rank_sweep_counted": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int)]),
    "ising_sweep_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ising_sweep_form": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ising_sweep_timed": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "ising_halo_ptrs": (C.c_int, [C.c_void_p, C.c_int] + [C.POINTER(C.c_void_p)] * 4 + [C.POINTER(C.c_size_t)]),
    "ising_ghost_ptrs": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)] + [C.POINTER(C.c_void_p)] * 4 + [C.POINTER(C.c_size_t)]),
    "ising_ghost_delivered": (C.c_int, [C.c_void_p, C.c_int]),
    "ising_sweep_ghost": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ising_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ising_bond_equal": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ising_read_packed": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "ising_write_packed": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "ising_read_bits": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "ising_write_bits": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "ising_checkpoint_info_read": (C.c_int, [C.c_char_p, C.c_void_p]),
    "ising_ring_checkpoint_save": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_int64]),
    "ising_ring_checkpoint_load": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.POINTER(C.c_int64)]),
    "ising_rank_checkpoint_save": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "ising_rank_checkpoint_load": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "ising_device_ptr": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "ising_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "ising_debug_fault": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ising_batch_debug_fault": (C.c_int, [C.c_void_p, C.c_int]),
    "ising_debug_launch_shape": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ising_dump_text": (C.c_int, [C.c_void_p, C.c_char_p]),
    "ising_ring_set_transport": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "ising_ring_transport": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]),
    "ising_ring_count": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ising_ring_bond_equal": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int64)]),
    "ising_rccl_available": (C.c_int, [C.POINTER(C.c_int)]),
    "ising_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "ising_rank_attach": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ising_batch_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]),
    "ising_batch_destroy": (C.c_int, [C.c_void_p]),
    "ising_batch_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ising_batch_sweep": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ising_batch_measure_enqueue": (C.c_int, [C.c_void_p]),
    "ising_batch_sweep_counted": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int)]),
    "ising_shape_guard_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ising_batch_quad_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ising_batch_measure_fetch": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int)]),
    "ising_ipc_export": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ising_ipc_attach": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "ising_rank_detach": (C.c_int, [C.c_void_p, C.c_int]),
    "ising_rank_exchange": (C.c_int, [C.c_void_p, C.c_int]),
    "ising_rank_init_couplings": (C.c_int, [C.c_void_p]),
    "ising_rank_sweep": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ising_rank_wait": (C.c_int, [C.c_void_p, C.c_int]),
    "ising_rank_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ising_rank_bond_equal": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ising_exchange_stats_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "ising_exchange_stats_fetch": (C.c_int, [C.c_void_p, C.POINTER(ExchangeStats)]),
    "ising_ring_exchange": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "ising_ring_sweep": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    "ising_use_private_stream": (C.c_int, [C.c_void_p]),
    "ising_measure_enqueue": (C.c_int, [C.c_void_p]),
    "ising_measure_fetch": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int)]),
    "ising_ring_synchronize": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "ising_correlations": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]),
    "ising_ring_correlations": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_int64)]),
}


def _share_torch_hip_runtime():
    """A torch ROCm wheel ships its own libamdhip64.so (and librccl.so on top of it) with the same SONAME as the
    system's.  libising_hip.so binds to whichever copy the process loaded first, so a process that loads this library
    BEFORE torch ends up with two HIP runtimes -- and stream handles, events and allocation records of one mean nothing
    to the other.  When torch is installed (it need not be imported) its copy is loaded first, so that the library,
    torch and torch's RCCL share one runtime.  ISING_HIP_RUNTIME=system keeps the system runtime instead."""
    if os.environ.get("ISING_HIP_RUNTIME", "auto") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001 -- best effort: the library still works on its own runtime
        pass


def load() -> C.CDLL:
    """Load libising_hip.so; raises IsingError (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IsingError(f"{LIB_PATH} is missing: build it with `make -C ising_gpu_amd/csrc` "
                             "(or __graft_entry__.build()); there is no CPU fallback")
        _share_torch_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int):
    if rc != 0:
        msg = load().ising_last_error()
        raise IsingError(f"libising_hip error {rc}: {msg.decode() if msg else '?'}", rc)
