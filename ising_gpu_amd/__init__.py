"""ising_gpu_amd -- MI355X-native (gfx950) 2-D Ising checkerboard-Metropolis engine.

The compute path is libising_hip.so (hand-written HIP kernels behind the C-ABI of include/ising_hip.h); this
package is the thin host-side mirror of the reference driver (optimized/main.cu) plus the multi-GPU slab ring.
The plain ctypes binding (IsingSlab, SlabSet) needs numpy only; the torch.distributed ring classes of `ring.py` are
imported on first use.
"""
from ._lib import BLACK, WHITE, HAM_BLACK, CRIT_TEMP_F32, SEED_DEF, KERNEL_AUTO, KERNEL_GENERIC, KERNEL_FAST, LAYOUT_AUTO, LAYOUT_NIBBLE, LAYOUT_DENSE, LAYOUT_BALLOT, TRANSPORT_AUTO, TRANSPORT_COPY, TRANSPORT_RCCL, TRANSPORT_IPC, IsingError, LIB_PATH  # noqa: F401
from .lattice import IsingSlab, SlabSet, IsingBatch, checkpoint_info, device_count, magnetization, energy_per_spin, ring_correlations, required_bytes, rccl_version, rccl_unique_id, philox_ceiling, philox_ceiling_clocked  # noqa: F401

_RING_NAMES = ("SlabRing", "LocalRing", "HipSlabBackend", "NativeRing", "open_ring", "open_native_ring")


def __getattr__(name):
    if name in _RING_NAMES:
        from . import ring  # needs torch
        return getattr(ring, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
