#!/usr/bin/env python3
"""bench.py -- spin-flips/ns of the checkerboard-Metropolis hot loop on N MI355X (weak scaling).

A "step" is one full lattice sweep (black half-sweep + white half-sweep = two launches of the update kernel,
the reference's hot loop optimized/main.cu:1763-1805) over this rank's slab.  Per-GPU workload (fixed as N
grows => weak scaling): X = 65536 columns x Y = 65536 rows at T = T_c (CRIT_TEMP, optimized/main.cu:42),
the 65536^2 lattice BASELINE.json's target is quoted on; with N ranks the lattice is (N*65536) x 65536, slabs
along Y, one RCCL row exchange per colour half-sweep (ising_gpu_amd/ring.py).  The lattice is generated on the
device from the seed (there is no input data): "synthetic".

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
                bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_FLIP = 1.5         # reference accounting, optimized/main.cu:1887-1889


def cpu_baseline(args):
    """Reported CPU baseline on the host cores of this box (rank 0, N=1 only): the byte-per-spin algorithm of
    basic_python/ising_basic.py restated in oracle/basic_cpu.c, BASELINE.json configs[0] (1024x1024, alpha 1,
    seed 1234, 100 warm-up + 1000 timed sweeps); plus the bit-exact packed oracle on a bounded sample."""
    import oracle
    oracle.build()
    ncpu = os.cpu_count() or 1
    threads = max(1, min(ncpu, 32, args.cpu_threads or 32))
    oracle.set_threads(threads)
    b = oracle.BasicCpuIsing(1024, 1024, alpha=1.0, seed=1234)
    b.sweeps(100)
    t0 = time.perf_counter()
    b.sweeps(1000)
    dt = time.perf_counter() - t0
    m, e = b.observables()
    out = {
        "value": round(1024 * 1024 * 1000 / dt * 1e-9, 4), "unit": "flips/ns", "cores": threads, "kind": "port",
        "sample": "basic (byte-per-spin) algorithm of basic_python/ising_basic.py, 1024x1024, alpha=1, seed 1234, "
                  "100 warm-up + 1000 timed sweeps, OpenMP over rows (oracle/basic_cpu.c); parity unpinned, "
                  f"|m|={abs(m):.4f} e={e:.4f}",
        "host_cpus": ncpu,
    }
    # second figure: the packed bit-exact oracle (same results as the GPU engine), 8192^2 x 4 sweeps
    threads2 = max(1, min(ncpu, 64))
    oracle.set_threads(threads2)
    L = oracle.OracleLattice(8192, 8192, seed=1234, temp=oracle.CRIT_TEMP).init()
    L.sweep(1)
    t0 = time.perf_counter()
    L.sweep(4)
    dt2 = time.perf_counter() - t0
    out["packed_oracle"] = {"value": round(8192 * 8192 * 4 / dt2 * 1e-9, 4), "unit": "flips/ns", "cores": threads2,
                            "sample": "bit-exact packed oracle (oracle/ising_oracle.c), 8192x8192, T=Tc, 4 sweeps"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--x", type=int, default=65536, help="columns (per-GPU slab and total)")
    ap.add_argument("--y", type=int, default=65536, help="rows per GPU")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--strip-rows", type=int, default=0)
    ap.add_argument("--layout", choices=["auto", "nibble", "dense", "ballot"], default="auto", help="device layout of the spin arrays")
    ap.add_argument("--exchange", choices=["p2p", "allgather"], default=None,
                    help="N > 1: how the edge rows travel (default p2p send/recv, or ISING_RING_EXCHANGE)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    # must be in the environment before the HIP/HSA runtime initialises (RCCL P2P needs dmabuf IPC on this host driver)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import ising_gpu_amd as ig

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # torch owns the slab's device buffer, so the rows RCCL sends/receives are slices of an ordinary torch tensor
    backend = ig.HipSlabBackend.create(args.x, args.y, device=local_rank, seed=args.seed, temp=ig.CRIT_TEMP_F32,
                                       nslabs=world, slab=rank, strip_rows=args.strip_rows,
                                       layout={"auto": ig.LAYOUT_AUTO, "nibble": ig.LAYOUT_NIBBLE, "dense": ig.LAYOUT_DENSE, "ballot": ig.LAYOUT_BALLOT}[args.layout])
    slab = backend.slab
    ring = ig.SlabRing(backend, exchange=args.exchange)
    ring.init()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ring.sweep(args.warmup)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    ring.sweep(args.steps)
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)  # HIP events on the stream the kernels were launched on (torch's current stream)
    if world > 1:
        t = torch.tensor([dt, ev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, ev_ms = float(t[0]), float(t[1])

    up, down = ring.count()
    spins_per_gpu = args.x * args.y
    total_flips = float(spins_per_gpu) * world * args.steps
    value = total_flips / (dt * 1e9)

    layout_name, layout_text, kernel_name = {
        ig.LAYOUT_NIBBLE: ("nibble", "reference 4 bit/spin", "update_k<0>"),
        ig.LAYOUT_DENSE: ("dense", "dense 1 bit/spin", "dense_update_k<0>"),
        ig.LAYOUT_BALLOT: ("ballot", "1 bit/spin in wave-ballot order", "ballot_update_k"),
    }[slab.current_layout()]
    if rank == 0:
        # dominant kernel: update_k; 2 full-slab launches per step (with N>1 each colour adds one tiny edge-row launch)
        launches = 2 * args.steps
        avg_launch_ms = ev_ms / launches
        alg_bytes_per_launch = BYTES_PER_FLIP * spins_per_gpu / 2.0  # src read + dst read + dst write of one colour
        achieved = alg_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
        traffic = None
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof):
            try:
                with open(prof) as f:
                    tj = json.load(f)
                if tj.get("x") == args.x and tj.get("y") == args.y and tj.get("device_layout") == layout_name:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "spin-flips/ns (whole node) at T=Tc", "value": round(value, 2), "unit": "flips/ns",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt * 1e3 / args.steps, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{args.y * world}x{args.x} lattice ({args.y}x{args.x} per GPU), T=Tc, seed {args.seed}, "
                                   "Philox4x32-10 per site; device layout " + layout_text
                                   + ", results identical to the reference's packed state", "x": args.x, "y_per_gpu": args.y,
                       "parallelism": f"slab{world}" + (f" ({ring.exchange} row exchange)" if world > 1 else ""), "strip_rows": slab.strip_rows, "device_layout": layout_name,
                       "up": up, "down": down},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": kernel_name, "avg_launch_ms": round(avg_launch_ms, 5),
                         "algorithmic_bytes_per_launch": alg_bytes_per_launch},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
