"""One process per slab on a real GPU: the SlabRing + HipSlabBackend pair bench.py runs under torchrun, here with 2 and
3 ranks sharing device 0.  RCCL refuses two ranks per device, so the rows travel through host staging buffers over
gloo (tools/ring_two_ranks_one_gpu.py: same schedule, same device buffers and kernels, other transport); every rank
checks its slab and the global counts against the CPU oracle -- in both forms of the exchange: one row per colour
half-sweep (torch-owned slabs), and ghost rows 32 deep every 16 sweeps through the C-ABI's deep exchange surface
(ising_ghost_ptrs / ising_ghost_delivered / ising_sweep_ghost) on library-owned ballot slabs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,port", [(2, 29533), (3, 29534)])
def test_slab_ring_processes_share_one_gpu(gpu, world, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "ring_two_ranks_one_gpu.py")],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    # (ranks may interleave their lines) every rank reports both layouts, nothing differs
    # ... and the library-owned slab with ghost rows after both of its sweep phases
    assert r.stdout.count("slab == oracle rows") == 4 * world and "!=" not in r.stdout, r.stdout[-3000:]
    assert r.stdout.count("ghost rows 32") == 2 * world, r.stdout[-3000:]
