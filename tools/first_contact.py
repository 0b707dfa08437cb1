"""First contact with a multi-GPU node: under a minute, one process per rank, every step says what it did and what it got.

  1. who is who       every rank's device ordinal, PCI bus id, name, its row of hipDeviceCanAccessPeer; do N ranks own N devices?
  2. per transport    (rccl, ipc -- the library's two rank transports, csrc/ising_ring.cpp / ising_ipc.cpp): attach (ncclCommInitRank, or the
                      hipIpcMemHandle maps of the neighbours' ghost rows + the shared epoch counters), its time, then TWO deep launches with
                      ONE exchange between them and one behind (64 sweeps of a 65536-column slab: 2 colours x 2 neighbours x 256 KiB per
                      exchange, the bench's row size), the exchange statistics, and the result against the same lattice swept as ONE lone slab
                      on rank 0's device (counts, bond sum, a CRC of every rank's rows) -- the lone path is what the test-suite pins on the oracle.
  3. summary          one JSON line on rank 0: per transport "ok" or the first error string of every rank; exit code 0 only when at least one
                      transport is green on every rank.

The control plane is gloo (a node where RCCL does not come up must still get its report).  Ranks may share a device (a 1-GPU box runs this with N
processes on one GPU: RCCL then refuses -- reported, not fatal -- and the peer transport carries the ring).

Launch: tools/first_contact.sh [N]      (= python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... tools/first_contact.py)"""
import json
import os
import sys
import time
import zlib

t_start = time.perf_counter()
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ising_gpu_amd as ig  # noqa: E402

X, YK, SEED, SWEEPS = 65536, 512, 1234, 64
rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
ndev = torch.cuda.device_count()
if ndev < 1:
    raise SystemExit("first_contact: no GPU")
dev = local % ndev
torch.cuda.set_device(dev)
if world > 1 or "MASTER_ADDR" in os.environ:
    sys.stdout.flush()
    saved = os.dup(1)  # (gloo announces its connections on stdout)
    os.dup2(2, 1)
    try:
        dist.init_process_group("gloo")
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def gather(obj):
    if world == 1:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def say(msg):
    print(f"[first_contact rank {rank}] {msg}", file=sys.stderr, flush=True)


props = torch.cuda.get_device_properties(dev)
bus = getattr(props, "pci_bus_id", None)
pci = None if bus is None else f"{getattr(props, 'pci_domain_id', 0):04x}:{bus:02x}:{getattr(props, 'pci_device_id', 0):02x}.0"
peers = []
for d in range(ndev):
    try:
        peers.append(1 if d == dev else int(torch.cuda.can_device_access_peer(dev, d)))
    except Exception:  # noqa: BLE001
        peers.append(-1)
who = gather({"rank": rank, "device": dev, "pci_bus_id": pci, "name": props.name, "peer_access_row": peers})
shared = len({w["pci_bus_id"] or w["device"] for w in who}) < world

# the reference result: the whole lattice as ONE lone slab on rank 0's device
ref = [None]
if rank == 0:
    with ig.IsingSlab(X, YK * world, device=dev, seed=SEED, temp=ig.CRIT_TEMP_F32) as s:
        s.init().sweep(SWEEPS)
        b, w = s.read(ig.BLACK), s.read(ig.WHITE)
        crcs = [(zlib.crc32(np.ascontiguousarray(b[k * YK:(k + 1) * YK]).tobytes()), zlib.crc32(np.ascontiguousarray(w[k * YK:(k + 1) * YK]).tobytes())) for k in range(world)]
        ref = [{"count": s.count(), "bond": s.bond_equal(), "crcs": crcs}]
if world > 1:
    dist.broadcast_object_list(ref, src=0)
ref = ref[0]

report = {}
for tr in ("rccl", "ipc"):
    rec = {"ok": False, "error": None}
    slab = ring = None
    if tr == "rccl" and shared and world > 1:
        report[tr] = {"ok_on_every_rank": False, "skipped": "ranks share a device: RCCL refuses duplicate GPUs in one communicator (the peer transport carries such rings)"}
        continue
    try:
        slab = ig.IsingSlab(X, YK, device=dev, seed=SEED, temp=ig.CRIT_TEMP_F32, nslabs=world, slab=rank, ring_halo=world == 1)
        t0 = time.perf_counter()
        ring = ig.NativeRing(slab, transport=tr)
        ring.init()
        ring.quiesce()
        torch.cuda.synchronize()
        rec["attach_and_init_s"] = round(time.perf_counter() - t0, 3)
        slab.exchange_stats_begin(16)
        t0 = time.perf_counter()
        ring.sweep(SWEEPS)  # ghost rows 64 deep: two launches of 32 sweeps, an exchange behind each
        ring.quiesce()
        torch.cuda.synchronize()
        rec["two_launches_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        st = slab.exchange_stats_fetch()
        rec["exchange_stats"] = {k: st[k] for k in ("exchanges", "launch_ms_mean", "exchange_ms_mean", "exchange_ms_max", "go_after_end_ms_mean", "go_after_end_ms_max", "gap_ms_mean") if k in st}
        cnt, bond = ring.count(), ring.bond_equal()
        mine = (zlib.crc32(np.ascontiguousarray(slab.read(ig.BLACK)).tobytes()), zlib.crc32(np.ascontiguousarray(slab.read(ig.WHITE)).tobytes()))
        rec["counts_equal_lone_slab"] = tuple(cnt) == tuple(ref["count"]) and bond == ref["bond"]
        rec["rows_equal_lone_slab"] = tuple(mine) == tuple(ref["crcs"][rank])
        rec["ok"] = bool(rec["counts_equal_lone_slab"] and rec["rows_equal_lone_slab"])
        if not rec["ok"]:
            rec["error"] = f"result differs from the lone slab's: counts {cnt} bond {bond} vs {ref['count']} {ref['bond']}; rows {'equal' if rec['rows_equal_lone_slab'] else 'differ'}"
    except Exception as e:  # noqa: BLE001 -- the report is the point
        rec["error"] = f"{type(e).__name__}: {e}"
    say(f"transport {tr}: {'ok' if rec['ok'] else rec['error']}")
    allrec = gather(rec)
    every = all(r["ok"] for r in allrec)
    try:
        if ring is not None:
            ring.close(abort=not every)
        elif slab is not None:
            slab.rank_detach(True)
    except Exception:  # noqa: BLE001
        pass
    if slab is not None:
        try:
            slab.close()
        except Exception:  # noqa: BLE001
            pass
    report[tr] = {"ok_on_every_rank": every, "by_rank": ["ok" if r["ok"] else r["error"] for r in allrec],
                  "attach_and_init_s_max": max((r.get("attach_and_init_s") or 0.0) for r in allrec),
                  "two_launches_ms_max": max((r.get("two_launches_ms") or 0.0) for r in allrec),
                  "exchange_stats_by_rank": [r.get("exchange_stats") for r in allrec]}

if rank == 0:
    try:
        rccl = ig.rccl_version()
    except Exception:  # noqa: BLE001
        rccl = None
    line = {"first_contact": True, "world": world, "physical_gpus": ndev, "ranks_share_devices": shared, "ranks": who,
            "versions": {"hip": getattr(torch.version, "hip", None), "rccl": rccl, "torch": torch.__version__},
            "lattice": f"{YK * world} x {X} ({YK} rows per rank), T_c, seed {SEED}, {SWEEPS} sweeps = two deep launches per rank, an exchange of 64 ghost rows of both colours behind each",
            "expected": {"exchange_ms_max": 0.45, "go_after_end_ms_max": 0.0,
                         "what": "one-GPU probes (profiles/strong_slab_probe_r04.txt): on real links an exchange (1 MiB per rank) ends before the launch whose tail hides it; "
                                 "ranks that share a device time-slice it and say nothing about links"},
            "transports": report, "seconds": round(time.perf_counter() - t_start, 1)}
    print(json.dumps(line), flush=True)
ok_any = any(v["ok_on_every_rank"] for v in report.values())
if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()
sys.exit(0 if ok_any else 1)
