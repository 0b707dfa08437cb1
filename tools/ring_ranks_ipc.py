"""One process per slab over the library's RCCL-free peer transport (ISING_TRANSPORT_IPC: hipIpcMemHandle-mapped ghost rows,
epoch counters in POSIX shared memory; csrc/ising_ipc.cpp), with the ranks SHARING device 0 when the box has one GPU --
RCCL refuses that, this transport does not -- or owning one GPU each (LOCAL_RANK < device count).  torch.distributed
(gloo) is only the launcher's channel: it carries the attachment blobs once and the barrier at the end.

  small   8192-column slabs against the CPU oracle, full state + counts + bond sum on every rank: the deep schedule (ballot
          layout, ghost rows Y/2 deep: fused launches between exchanges, several sweep calls incl. one longer than the ghost
          rows carry), one halo row per colour half-sweep on two streams (dense layout; ballot layout with ISING_RING_GHOST=1),
          and -J couplings (the black coupling rows travel once at initialisation).
  golden  the bench's slabs (65536 x 65536 per rank, T_c, seed 1234): counts after 0 / 5 / 21 / 25 sweeps against the oracle's
          golden ring counts (tests/golden/ring_65536_tc.json, N = world).

  soak    32768 x 2048 per rank, 24000 sweeps in uneven calls: counts and bond sum against the whole lattice as one slab.

Launch: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 --master-port 29551 tools/ring_ranks_ipc.py small|golden|soak"""
import json
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ising_gpu_amd as ig  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "small"
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
dev = local % max(1, torch.cuda.device_count())
torch.cuda.set_device(dev)
dist.init_process_group("gloo")


def check_small():
    import tempfile
    import oracle
    ckdir = [tempfile.mkdtemp(prefix="ising_ckpt_") if rank == 0 else None]
    dist.broadcast_object_list(ckdir, src=0)
    ckdir = ckdir[0]
    oracle.set_threads(min(16, os.cpu_count() or 1))
    X, Y, seed, temp = 8192, 64, 2024, ig.CRIT_TEMP_F32
    # (layout, ISING_RING_GHOST, J_prob, sweep calls)
    cases = [(ig.LAYOUT_BALLOT, None, None, (2, 19, 16)), (ig.LAYOUT_DENSE, None, None, (2, 3)), (ig.LAYOUT_BALLOT, "1", None, (2, 3)),
             (ig.LAYOUT_BALLOT, None, 0.3, (5, 17)), (ig.LAYOUT_DENSE, None, 0.3, (4,))]
    # sub-lattices: nothing crosses slabs, every rank sweeps on its own (fused launches); only the totals are collective
    with ig.IsingSlab(X, Y, device=dev, seed=seed, temp=temp, nslabs=world, slab=rank, layout=ig.LAYOUT_BALLOT, XSL=4096, YSL=32) as sl:
        ring = ig.NativeRing(sl, transport="ipc").init()
        orc = oracle.OracleLattice(X, Y * world, seed=seed, temp=temp, XSL=4096, YSL=32).init().sweep(6)
        ring.sweep(2).sweep(4)
        tot, bond = ring.count(), ring.bond_equal()
        lo, hi = rank * Y, (rank + 1) * Y
        ok = np.array_equal(sl.read(ig.BLACK), orc.black[lo:hi]) and np.array_equal(sl.read(ig.WHITE), orc.white[lo:hi]) and tot == orc.count() and bond == orc.bond_equal()
        print(f"rank {rank} ipc sub-lattices 4096 x 32 after 6 sweeps: slab, counts, bond sum {'==' if ok else '!='} oracle (sub-lattice run)", flush=True)
        assert ok
        ring.close()
    for layout, ghost_env, jprob, calls in cases:
        if ghost_env is None:
            os.environ.pop("ISING_RING_GHOST", None)
        else:
            os.environ["ISING_RING_GHOST"] = ghost_env
        slab = ig.IsingSlab(X, Y, device=dev, seed=seed, temp=temp, nslabs=world, slab=rank, layout=layout, J_prob=jprob, ring_halo=world == 1)
        ring = ig.NativeRing(slab, transport="ipc").init()
        depth = slab.ghost_ptrs(ig.BLACK)[0]
        want_depth = 32 if (layout == ig.LAYOUT_BALLOT and ghost_env is None) else 1
        assert depth == want_depth, (depth, want_depth)
        orc = oracle.OracleLattice(X, Y * world, seed=seed, temp=temp)
        orc.init()
        if jprob is not None:
            orc.init_couplings(jprob)
        for n in calls:
            ring.sweep(n)
            orc.sweep(n)
            bond = ring.bond_equal()
            tot = ring.count()
            ring.quiesce()
            lo, hi = rank * Y, (rank + 1) * Y
            ok = np.array_equal(slab.read(ig.BLACK), orc.black[lo:hi]) and np.array_equal(slab.read(ig.WHITE), orc.white[lo:hi])
            good = ok and tot == orc.count() and bond == orc.bond_equal()
            print(f"rank {rank} ipc layout {layout} ghost rows {depth} J {jprob} after {ring.it} sweeps: slab {'==' if ok else '!='} oracle rows [{lo},{hi}); "
                  f"counts {tot} bond {bond} {'==' if good else '!='} oracle", flush=True)
            assert good
        if jprob is None:
            # checkpoint written by the ranks together, continued, loaded back: the continuation repeats itself; and the file
            # is the one a single process writes (global row order): rank 0 loads it into ONE slab and compares the counts
            path = os.path.join(ckdir, f"ring_{layout}_{ghost_env}.ckpt")
            ring.checkpoint_save(path)
            at = ring.it
            ring.sweep(3)
            want = (ring.count(), ring.bond_equal())
            ring.checkpoint_load(path)
            assert ring.it == at
            ring.sweep(3)
            got = (ring.count(), ring.bond_equal())
            orc.sweep(3)
            ok = got == want == (orc.count(), orc.bond_equal())
            print(f"rank {rank} ipc layout {layout} ghost rows {depth}: checkpoint at {at}, continuation {'==' if ok else '!='} oracle", flush=True)
            assert ok
            # calls that cannot succeed fail on EVERY rank, at once (the ranks agree on each stage's outcome), and leave the ring usable
            for doomed in (lambda: ring.checkpoint_load(os.path.join(ckdir, "missing.ckpt")),
                           lambda: ring.checkpoint_save(os.path.join(ckdir, "no_such_dir", "x.ckpt"))):
                try:
                    doomed()
                    raise SystemExit("a checkpoint call that cannot succeed returned")
                except ig.IsingError:
                    pass
            ring.sweep(1)
            orc.sweep(1)
            assert (ring.count(), ring.bond_equal()) == (orc.count(), orc.bond_equal())
            print(f"rank {rank} ipc layout {layout} ghost rows {depth}: doomed checkpoint calls failed on this rank, the ring went on", flush=True)
            if rank == 0:
                with ig.IsingSlab(X, Y * world, device=dev, seed=seed, temp=temp, layout=ig.LAYOUT_DENSE) as one:
                    single = ig.SlabSet([one])
                    single.checkpoint_load(path)
                    single.sweep(3)
                    assert single.it == at + 3 and single.count() == want[0]
        ring.close()
        slab.close()
    os.environ.pop("ISING_RING_GHOST", None)


def check_golden():
    gold = [r for r in json.load(open(os.path.join(ROOT, "tests", "golden", "ring_65536_tc.json")))["rings"] if r["nslabs"] == world]
    assert gold, f"no golden ring of {world} slabs"
    gold = gold[0]
    slab = ig.IsingSlab(gold["X"], gold["Ytot"] // world, device=dev, seed=gold["seed"], temp=ig.CRIT_TEMP_F32, nslabs=world, slab=rank)
    ring = ig.NativeRing(slab, transport="ipc").init()
    assert slab.fused and slab.max_sweeps_per_launch == 32  # ghost rows 64 deep, fused launches between exchanges
    for pt in gold["points"]:
        ring.sweep(pt["sweeps"] - ring.it)
        tot = ring.count()
        ok = tot == (pt["up"], pt["down"])
        print(f"rank {rank} ipc golden N={world} after {ring.it} sweeps: counts {tot} {'==' if ok else '!='} oracle golden", flush=True)
        assert ok
    ring.close()
    slab.close()


def check_soak():
    """Thousands of exchanges between real processes next to running fused launches (small slabs: the exchange has ~1 ms per
    launch), uneven call lengths; counts and bond sum at every checkpoint against the whole lattice as ONE slab (rank 0 runs it)."""
    X, Y, seed, total, ncheck = 32768, 2048, 31, int(os.environ.get("ISING_SOAK_SWEEPS", "24000")), 6  # (ISING_SOAK_SWEEPS: a longer one-off hunt)
    slab = ig.IsingSlab(X, Y, device=dev, seed=seed, temp=ig.CRIT_TEMP_F32, nslabs=world, slab=rank, layout=ig.LAYOUT_BALLOT)
    ring = ig.NativeRing(slab, transport="ipc").init()
    ref = ig.IsingSlab(X, Y * world, device=dev, seed=seed, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT).init() if rank == 0 else None
    step = total // ncheck
    for k in range(ncheck):
        n = 0
        while n < step:
            m = min(step - n, (97, 32, 5, 64, 1, 33)[(n + k) % 6])
            ring.sweep(m)
            n += m
        got = [ring.count(), ring.bond_equal()]
        want = [None]
        if rank == 0:
            ref.sweep(step)
            want = [[ref.count(), ref.bond_equal()]]
        dist.broadcast_object_list(want, src=0)
        ok = [tuple(got[0]), got[1]] == [tuple(want[0][0]), want[0][1]]
        print(f"rank {rank} ipc soak N={world} after {ring.it} sweeps: counts and bond sum {'==' if ok else '!='} lone slab", flush=True)
        assert ok
    ring.close()
    slab.close()


if mode == "golden":
    check_golden()
elif mode == "soak":
    check_soak()
else:
    check_small()
dist.barrier()
dist.destroy_process_group()
