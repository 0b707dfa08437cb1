"""ising_gpu_amd -- MI355X-native (gfx950) 2-D Ising checkerboard-Metropolis engine.

The compute path is libising_hip.so (hand-written HIP kernels behind the C-ABI of include/ising_hip.h); this
package is the thin host-side mirror of the reference driver (optimized/main.cu) plus the multi-GPU slab ring.
"""
from ._lib import BLACK, WHITE, HAM_BLACK, CRIT_TEMP_F32, SEED_DEF, KERNEL_AUTO, KERNEL_GENERIC, KERNEL_FAST, KERNEL_LUT, LAYOUT_AUTO, LAYOUT_NIBBLE, LAYOUT_DENSE, LAYOUT_BALLOT, IsingError, LIB_PATH  # noqa: F401
from .lattice import IsingSlab, device_count, magnetization, energy_per_spin, ring_correlations, required_bytes  # noqa: F401
from .ring import SlabRing, LocalRing, HipSlabBackend  # noqa: F401
