"""The path a scaling run of bench.py takes, one process per rank under torch.distributed.run: ring slabs on the
ballot layout that sweep in fused launches between exchanges of 64 ghost rows, the exchange overlapped with the launches
(csrc/ising_ring.cpp: sweep_deep_overlapped) -- through the library's RCCL ring (every rank a device of its own) or its RCCL-free peer
ring (ISING_TRANSPORT_IPC; ranks spread over the devices there are, sharing them when there are fewer: a 1-GPU box runs all of it).

  golden <workload>   bench.py's slab of <workload> (config3: 65536 x 65536 per rank, config4: 131072 columns x 16384 rows, strong:
                      65536 columns x 65536 / N rows) -- counts after 5 / 21 / 25 sweeps (uneven calls: one crosses an exchange) against
                      the CPU oracle's goldens for the TOTAL lattice (tests/golden: bench.golden_records)
  counted             16384 x 2048 per rank: ising_rank_sweep_counted's print points (inside the deep launches) and the state afterwards against the CPU oracle
  state               16384 columns x 2048 rows per rank, 64 ghost rows, fused + overlapped as above: FULL state of every rank, counts
                      and bond sum against the CPU oracle after 3, 36 and 71 sweeps (three exchanges deep)

Launch: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/ring_ranks_scale.py rccl|ipc golden config3
"""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
if len(sys.argv) > 2 and sys.argv[2] == "counted":
    os.environ["ISING_RING_COUNTED"] = "2"  # the print points INSIDE the deep launches, or an error (no silent fall-back to sweep-and-count)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ising_gpu_amd as ig  # noqa: E402

transport, mode = sys.argv[1], sys.argv[2]
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
ndev = torch.cuda.device_count()
if transport == "rccl" and ndev < world:
    raise SystemExit(f"the RCCL ring needs a device per rank ({world} ranks, {ndev} devices)")
dev = local % max(1, ndev)
torch.cuda.set_device(dev)
if transport == "rccl":
    dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
else:
    dist.init_process_group("gloo")


def open_ring(X, Y, seed):
    slab = ig.IsingSlab(X, Y, device=dev, seed=seed, temp=ig.CRIT_TEMP_F32, nslabs=world, slab=rank, layout=ig.LAYOUT_BALLOT)
    ring = ig.NativeRing(slab, transport=transport).init()
    # the default schedule of a scaling run: 64 ghost rows, fused launches of 32 sweeps between exchanges, overlapped
    assert slab.current_layout() == ig.LAYOUT_BALLOT and slab.fused and slab.max_sweeps_per_launch == 32, (slab.fused, slab.max_sweeps_per_launch)
    assert slab.ghost_ptrs(ig.BLACK)[0] == 64
    return slab, ring


def check_golden(workload):
    import bench
    x, rows_of, _ = bench.WORKLOADS[workload]
    y = rows_of(world)
    rec = bench.golden_records().get((x, y * world, 1234))
    assert rec, f"no golden for the {y * world} x {x} lattice"
    slab, ring = open_ring(x, y, 1234)
    slab.exchange_stats_begin(16)
    points = [s for s in (5, 21, 25) if s in rec]  # (scaling.json holds 0 / 5 / 25 / 144; the round-2 files also 21)
    assert len(points) >= 2, sorted(rec)
    for sweeps in points:
        ring.sweep(sweeps - ring.it)
        tot = ring.count()
        ok = tot == tuple(rec[sweeps])
        print(f"rank {rank} {transport} {workload} N={world} ({y} x {x} per rank, device {dev}) after {ring.it} sweeps: counts {tot} {'==' if ok else '!='} oracle golden", flush=True)
        assert ok
    st = slab.exchange_stats_fetch()
    print(f"rank {rank} exchange stats: {st}", flush=True)
    assert st["exchanges"] == len(points) and st["launch_ms_mean"] > 0
    ring.close()
    slab.close()


def check_state():
    import oracle
    oracle.set_threads(min(16, os.cpu_count() or 1))
    X, Y, seed = 16384, 2048, 97
    slab, ring = open_ring(X, Y, seed)
    orc = oracle.OracleLattice(X, Y * world, seed=seed, temp=oracle.CRIT_TEMP).init()
    for n in (3, 33, 35):
        ring.sweep(n)
        orc.sweep(n)
        tot, bond = ring.count(), ring.bond_equal()
        ring.quiesce()
        lo, hi = rank * Y, (rank + 1) * Y
        ok = np.array_equal(slab.read(ig.BLACK), orc.black[lo:hi]) and np.array_equal(slab.read(ig.WHITE), orc.white[lo:hi])
        good = ok and tot == orc.count() and bond == orc.bond_equal()
        print(f"rank {rank} {transport} state N={world} after {ring.it} sweeps: slab {'==' if ok else '!='} oracle rows [{lo},{hi}); counts {tot} bond {bond} "
              f"{'==' if good else '!='} oracle", flush=True)
        assert good
    ring.close()
    slab.close()


def check_counted():
    """Print points inside the deep launches (ising_rank_sweep_counted): 16384 x 2048 per rank, calls of uneven lengths, counts of every iteration that
    is a multiple of 16 (then 5) against the CPU oracle, the state afterwards."""
    import oracle
    oracle.set_threads(min(16, os.cpu_count() or 1))
    X, Y, seed = 16384, 2048, 131
    slab, ring = open_ring(X, Y, seed)
    orc = oracle.OracleLattice(X, Y * world, seed=seed, temp=oracle.CRIT_TEMP).init()
    for n, every in ((40, 16), (33, 16), (23, 5)):
        got = ring.sweep_counted(n, every)
        want = []
        for _ in range(n):
            orc.sweep(1)
            if orc.it % every == 0:
                want.append(orc.count())
        ok = got == want
        print(f"rank {rank} {transport} counted N={world} sweeps {orc.it - n + 1}..{orc.it} every {every}: {len(got)} counts {'==' if ok else '!='} oracle", flush=True)
        assert ok, (got, want)
    ring.quiesce()
    lo, hi = rank * Y, (rank + 1) * Y
    ok = np.array_equal(slab.read(ig.BLACK), orc.black[lo:hi]) and np.array_equal(slab.read(ig.WHITE), orc.white[lo:hi]) and ring.count() == orc.count()
    print(f"rank {rank} {transport} counted N={world} state after {ring.it} sweeps: slab {'==' if ok else '!='} oracle rows [{lo},{hi})", flush=True)
    assert ok
    ring.close()
    slab.close()


if mode == "golden":
    check_golden(sys.argv[3] if len(sys.argv) > 3 else "config3")
elif mode == "state":
    check_state()
elif mode == "counted":
    check_counted()
else:
    raise SystemExit(f"unknown mode {mode!r}")
dist.barrier()
dist.destroy_process_group()
