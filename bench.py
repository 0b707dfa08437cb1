#!/usr/bin/env python3
"""bench.py -- spin-flips/ns of the checkerboard-Metropolis hot loop on N MI355X.

A "step" is one full lattice sweep (black half-sweep + white half-sweep, the reference's hot loop
optimized/main.cu:1763-1805) over this rank's slab.  The lattice is generated on the device from the seed ("synthetic"),
T = T_c (CRIT_TEMP, optimized/main.cu:42), slabs along Y (optimized/main.cu:1590-1591: the lattice is ndev*Y x X).  Workloads:

  config3 (default; weak scaling)   65536 columns x 65536 rows per GPU: the 65536^2 lattice BASELINE.json's target is quoted on
                                    (configs[2]); with N ranks the lattice is (N*65536) x 65536
  config4 (weak scaling)            131072 columns x 16384 rows per GPU: BASELINE config 4's slab (131072^2 over 8 GPUs) at every N
  strong  (strong scaling)          the 65536^2 lattice as a whole, 65536/N rows per GPU (north_star: "spin-flips/ns on a 65536^2
                                    lattice ... reported at 1, 2, 4 and 8 GPUs")

Every point of a scaling run checks itself: the counts after warm-up + steps are compared with the CPU oracle's committed
goldens for the TOTAL lattice (tests/golden/bench_65536_tc.json, ring_65536_tc.json, config4_131072.json, scaling.json: 0 / 5 / 25 /
144 sweeps = the driver's --warmup 5 --steps 20 and the default 16 + 128, for all three workloads at N = 1, 2, 4, 8) --
`config.parity_checked`; a mismatch is a non-zero exit.

  N = 1   the slab sweeps itself (ising_sweep), `batch` sweeps per call (batch = the largest divisor <= 32 of
          gcd(steps, warmup)).  ising_sweep issues fused launches from 1.5 * 2^24 spins up (ISING_FUSED=0: one launch per
          colour): every call is ONE launch of 2 * batch colour half-sweeps, so that every launch of the run -- warm-up
          included -- is the same piece of work and the rocprofv3 per-kernel average agrees with the HIP-event average
          reported here (the default 128 + 16: pieces of 16).  Where that common piece would be under 16 sweeps (--steps 20
          --warmup 5) the timed steps and the warm-up are cut on their own (one call of 20, one of 5): `config.sweeps_per_call`
          and `roofline.half_sweeps_per_launch` describe the timed launches.
  N > 1   one process per GPU; the ring lives inside libising_hip.so (ising_rank_*: second HIP stream + RCCL send/recv);
          ballot ring slabs keep ghost rows 64 deep, exchange 64 rows of both colours every 32 sweeps and run one fused launch
          in between (sweep_deep_overlapped).  Transports, in this order, every rank agreeing on each outcome: the library's RCCL ring,
          the library's RCCL-free peer ring (hipIpcMemHandle-mapped ghost rows, ISING_TRANSPORT_IPC), the torch.distributed
          ring (p2p on ghost rows, p2p / all-gather with one row per colour half-sweep on a torch-owned slab); the JSON line
          says which one ran, and `exchange_stats` where each rank's time went around its exchanges (launch, exchange, how far
          past the launch's end the exchange ran, gap to the next launch: max / mean over ranks and exchanges).  With more
          ranks than GPUs (a 1-GPU box running --gpus 2) the ranks share devices: gloo carries the control plane and the peer
          ring the rows (RCCL refuses two ranks per device).  `python bench.py --gpus N` without a launcher starts itself
          under torch.distributed.run.

The JSON line: `value` = bare sweeps (the contract's timed region); `with_counts_every_16` = a second leg over the same sweeps with
the magnetisation of every 16th sweep (and of the last) inside the timed region, as every number the reference publishes includes it
(optimized/main.cu:1806-1810); `with_counts_and_energy_every_16` = a third leg with the bond sum (north_star's energy series) at the same points -- both taken inside the launches: ising_sweep_counted at N = 1, ising_rank_sweep_counted on the library's ring --; `roofline.frac` = SURVEY 8(d)'s number (1.5 B per flip x flips per launch / average launch duration / 8 TB/s: reproducible from the rocprofv3 summary under profiles/),
`roofline.bound` = "valu" -- the roof that binds in fact, with the ratio to a draw-only kernel measured in the same job as `frac_valu_ceiling` -- and the device's real HBM traffic next to them.

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
                bench.py --gpus N --steps K --warmup W [--workload config3|config4|strong]
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_FLIP = 1.5         # reference accounting, optimized/main.cu:1887-1889


def cgroup_cpus():
    """CPUs' worth of time the container's cgroup grants (cgroup v2 cpu.max), or None."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else round(int(quota) / int(period), 2)
    except (OSError, ValueError):
        return None


def exchange_overlap(ring_name):
    """What "the exchange next to the launches" means per transport: measured on a ring of one, profiles/strong_slab_probe_r04.txt."""
    if ring_name.startswith("rccl"):
        return ("tail-serialised: RCCL's send/recv kernel (132 vector registers a lane) finds no room next to the persistent grid and runs when the launch's workgroups "
                "retire; the next launch waits for it (ev_go).  Ring of one: +0.01..0.04 ms past the launch's end, a gap of 11-44 us per 32 sweeps (<= 0.9 %)")
    if ring_name.startswith("ipc"):
        return ("inside the launch: one-lane kernels and peer stores on the comm stream move the rows while the launch works on the slab's interior; a device per rank: ONE "
                "persistent launch carries several exchange epochs (ISING_RING_EPOCHS), its edge units wait for each exchange in place -- no launch boundary per exchange")
    if ring_name == "none":
        return None
    return "between launches (torch.distributed ring: the rows travel when the launch has ended)"


def first_contact_block(args, ig, torch, dist, world, rank, device, ndev, shared, attempts, layout, ctl, value, xstats):
    """Keys that make the first run on a multi-GPU node explain itself (collective: every rank calls it; rank 0 uses the result).
    transport_attempts: per rank, every transport tried with "ok" or its error string; ranks: device ordinal, PCI bus id, the row of the peer-access matrix,
    name; versions: HIP runtime and RCCL as the library loaded them; expected: the rate of THIS slab shape as a lone slab (periodic in itself) measured on
    this rank's device right here (~0.3 s), the envelope of an exchange (one-GPU probes, profiles/strong_slab_probe_r04.txt: 2 x 2 x 256 KiB per exchange
    at X = 65536; <= 0.45 ms, ending before the launch whose tail hides it); efficiency_vs_lone_slab = value / sum of the ranks' lone-slab rates."""
    props = torch.cuda.get_device_properties(device)
    bus = getattr(props, "pci_bus_id", None)
    pci = None if bus is None else f"{getattr(props, 'pci_domain_id', 0):04x}:{bus:02x}:{getattr(props, 'pci_device_id', 0):02x}.0"
    peers = []
    for d in range(ndev):
        try:
            peers.append(1 if d == device else int(torch.cuda.can_device_access_peer(device, d)))
        except Exception:  # noqa: BLE001
            peers.append(-1)
    # the lone slab: same shape, same device, wraps in itself; a preheat, then the best of three timed pieces
    lone = None
    try:
        with ig.IsingSlab(args.x, args.y, device=device, seed=args.seed, temp=ig.CRIT_TEMP_F32, strip_rows=args.strip_rows, layout=layout) as s:
            s.init()
            piece = max(1, min(32, s.max_sweeps_per_launch or 32))
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.15:
                s.sweep(piece)
                s.synchronize()
            for _ in range(3):
                t0 = time.perf_counter()
                s.sweep(piece)
                s.synchronize()
                r = args.x * args.y * piece / (time.perf_counter() - t0) * 1e-9
                lone = r if lone is None else max(lone, r)
    except ig.IsingError as e:
        lone = None
        attempts = attempts + [{"transport": "lone-slab-probe", "ok": False, "error": str(e)}]
    mine = {"rank": rank, "device": device, "pci_bus_id": pci, "name": props.name, "peer_access_row": peers, "lone_slab_flips_per_ns": None if lone is None else round(lone, 1),
            "transport_attempts": attempts}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rank != 0:
        return None
    devices = [r["pci_bus_id"] or f"ordinal{r['device']}" for r in per_rank]
    lone_sum = sum(r["lone_slab_flips_per_ns"] or 0.0 for r in per_rank)
    try:
        rccl = ig.rccl_version()
    except Exception:  # noqa: BLE001
        rccl = None
    out = {
        "ranks": [{k: r[k] for k in ("rank", "device", "pci_bus_id", "name", "peer_access_row", "lone_slab_flips_per_ns")} for r in per_rank],
        "distinct_devices": len(set(devices)),
        "ranks_own_their_device": len(set(devices)) == world and not shared,
        "transport_attempts": {f"rank{r['rank']}": [(a["transport"] + ": " + ("ok" if a["ok"] else str(a["error"])) + (f" [{a['cross_check']}]" if a.get("cross_check") else "")
                                                    + ("" if a.get("all_ranks_ok", a["ok"]) == a["ok"] else " (another rank failed)")) for a in r["transport_attempts"]] for r in per_rank},
        "versions": {"hip": getattr(torch.version, "hip", None), "rccl": rccl, "torch": torch.__version__},
        "expected": {"lone_slab_flips_per_ns_sum": round(lone_sum, 1),
                     "what": "each rank's slab shape as a lone slab (periodic in itself) on its own device, measured in this job; a ring slab does 128 of Y + 128 rows again "
                             "(the ghost rows) and exchanges once per 32 sweeps",
                     "ghost_row_overhead": round(128.0 / (args.y + 128.0), 4),
                     "exchange_ms_max": 0.45, "go_after_end_ms_max": 0.0,
                     "exchange_what": "one-GPU probes (profiles/strong_slab_probe_r04.txt): an exchange moves 2 colours x 2 neighbours x 64 rows (256 KiB each at X = 65536) and starts when the "
                                      "launch's edge strips finish their last level, ~0.4 ms before the launch ends -- on real links it should end before the launch does"},
        "efficiency_vs_lone_slab": None if lone_sum <= 0 else round(value / lone_sum, 4),
    }
    if xstats is not None:
        out["expected"]["exchange_within_envelope"] = bool(xstats["exchange_ms"]["max"] <= 0.45 and xstats["go_after_end_ms"]["max"] <= 0.0)
    return out


def default_lattice_leg(args, device):
    """The reference's DEFAULT lattice (2048 x 2048, optimized/main.cu:1395-1410) next to the headline's: what `cuIsing` without -x/-y runs.  Round 5: the quad
    path (ising_quad.hip: one launch per pass of 8 sweeps = the word pass on tiles + halo next to the draws of the pass to come) against the tile launches of
    round 4 (ISING_QUAD=0), same seed and sweeps, final counts compared (both paths are held against the oracle bit for bit by the GPU tests)."""
    import ising_gpu_amd as ig
    X = Y = 2048
    n = 4096
    out = {}
    for name, env in (("value", None), ("without_quad_path", "0")):
        old = os.environ.get("ISING_QUAD")
        if env is None:
            os.environ.pop("ISING_QUAD", None)
        else:
            os.environ["ISING_QUAD"] = env
        try:
            with ig.IsingSlab(X, Y, device=device, seed=args.seed, temp=ig.CRIT_TEMP_F32) as s:
                s.init().sweep(256)
                s.sweep_timed(n)
                ms = min(s.sweep_timed(n) for _ in range(3))
                out[name] = round(X * Y * n / (ms * 1e6), 1)
                out[name + "_form"] = "quad" if s.quad else ("tiles" if s.tiled else ("fused" if s.fused else "one launch per colour"))
                out[name + "_counts"] = list(s.count())
        finally:
            if old is None:
                os.environ.pop("ISING_QUAD", None)
            else:
                os.environ["ISING_QUAD"] = old
    return {"value": out["value"], "unit": "flips/ns", "lattice": f"{Y}x{X}", "sweeps_per_call": n, "form": out["value_form"],
            "frac_hbm_1p5B": round(out["value"] * 1.5 / 8000.0, 4), "without_quad_path_frac_hbm_1p5B": round(out["without_quad_path"] * 1.5 / 8000.0, 4),
            "without_quad_path": out["without_quad_path"], "without_quad_path_form": out["without_quad_path_form"],
            "counts_equal": out["value_counts"] == out["without_quad_path_counts"], "up_down": out["value_counts"],
            "what": "the reference's default lattice (cuIsing without -x/-y, optimized/main.cu:1395-1410) at T = Tc: best of three calls of 4096 sweeps; "
                    "`without_quad_path` = the same with ISING_QUAD=0 (round 4's tile launches)"}


def batched_small_leg(args, device, lone_counts):
    """31 lattices of 2048 x 2048 -- a temperature series at the reference's default size, T = 1.5 .. 3.0 -- as ONE batch (round 6: the tiles of all of them in one
    quad_pass_k launch per pass, ising_batch_sweep).  The reference's own many-small-systems mode (--xsl/--ysl, optimized/main.cu:1423-1457; README.md:148-198: at
    its big-lattice rate) knows one temperature.  Member 0 runs at T = Tc with the bench's seed: its counts after the same sweeps must be the lone lattice's."""
    import ising_gpu_amd as ig
    X = Y = 2048
    n_lat, n = 31, 4096
    temps = [ig.CRIT_TEMP_F32] + [1.5 + 0.05 * k for k in range(1, n_lat)]
    slabs = [ig.IsingSlab(X, Y, device=device, seed=args.seed, temp=t) for t in temps]
    try:
        with ig.IsingBatch(slabs) as b:
            b.init().sweep(256)
            slabs[0].synchronize()
            best = 0.0
            for _ in range(4):  # 256 + 4 x 4096 sweeps = the default-lattice leg's
                t0 = time.perf_counter()
                b.sweep(n)
                slabs[0].synchronize()
                best = max(best, X * Y * n_lat * n / (time.perf_counter() - t0) * 1e-9)
            counts0 = list(slabs[0].count())
            shape = b.quad_shape
    finally:
        for s in slabs:
            s.close()
    return {"value": round(best, 1), "unit": "flips/ns", "lattices": n_lat, "lattice": f"{Y}x{X}", "sweeps_per_call": n, "frac": round(best * 1.5 / 8000.0, 4),
            "form": "quad batch" if shape else "ballot batch", "tile_row_groups_sweeps_per_pass_waves": list(shape) if shape else None,
            "member0_counts_equal_lone_lattice": counts0 == list(lone_counts),
            "what": "31 lattices of the reference's default size at 31 temperatures in one launch per pass (ising_batch_sweep); best of four calls of 4096 sweeps, host clock"}


def cpu_baseline(args):
    """Reported CPU baseline on the host cores of this box (rank 0, N=1 only): the byte-per-spin algorithm of
    basic_python/ising_basic.py restated in oracle/basic_cpu.c, BASELINE.json configs[0] (1024x1024, alpha 1,
    seed 1234, 100 warm-up + 1000 timed sweeps); plus the bit-exact packed oracle on a bounded sample."""
    import oracle
    oracle.build()
    ncpu = os.cpu_count() or 1
    # the fastest OpenMP team for this 1 MiB lattice on this host, out of 16 / 32 / 64 / 128 threads (all four stated):
    # a team as wide as the host's 256 hardware threads spends its time in barriers (two per sweep)
    teams = [args.cpu_threads] if args.cpu_threads else [t for t in (16, 32, 64, 128) if t <= ncpu] or [ncpu]
    by_team = {}
    for threads in teams:
        oracle.set_threads(threads)
        b = oracle.BasicCpuIsing(1024, 1024, alpha=1.0, seed=1234)
        b.sweeps(100)
        t0 = time.perf_counter()
        b.sweeps(1000)
        dt = time.perf_counter() - t0
        m, e = b.observables()
        by_team[threads] = (round(1024 * 1024 * 1000 / dt * 1e-9, 4), m, e)
    threads = max(by_team, key=lambda t: by_team[t][0])
    value, m, e = by_team[threads]
    out = {
        "value": value, "unit": "flips/ns", "cores": threads, "kind": "port",
        "sample": "basic (byte-per-spin) algorithm of basic_python/ising_basic.py, 1024x1024, alpha=1, seed 1234, "
                  "100 warm-up + 1000 timed sweeps, OpenMP over rows (oracle/basic_cpu.c; the same run from the command line: "
                  "oracle/ising_basic_cpu -x 1024 -y 1024 -a 1 -s 1234 -w 100 -n 1000); parity unpinned, "
                  f"|m|={abs(m):.4f} e={e:.4f}; fastest of the teams tried",
        "by_threads": {str(t): v[0] for t, v in by_team.items()},
        "host_cpus": ncpu,
        "cgroup_cpus": cgroup_cpus(),  # what the container may actually use (cpu.max quota / period); None: no limit found
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


def golden_records():
    """Every committed full-size oracle record, keyed on the TOTAL lattice: {(X, Ytot, seed): {sweeps: (up, down)}}.  Results do
    not depend on the decomposition (optimized/main.cu:514: the Philox stream id uses the global block row; :1590-1591: the
    lattice is ndev*Y x X), so one record serves every split of its lattice into slabs."""
    gold = os.path.join(ROOT, "tests", "golden")
    recs = {}

    def add(fx, seed=None):
        key = (fx["X"], fx["Ytot"], fx.get("seed", seed))
        pts = recs.setdefault(key, {})
        for pt in fx["points"]:
            pts[pt["sweeps"]] = (pt["up"], pt["down"])

    def load(name):
        try:
            return json.load(open(os.path.join(gold, name)))
        except (OSError, ValueError):
            return None
    for name in ("bench_65536_tc.json", "config4_131072.json"):  # make_golden_big.py
        fx = load(name)
        if fx:
            add(fx)
    fx = load("ring_65536_tc.json")                              # make_golden_big.py: (N * 65536) x 65536
    for r in (fx or {}).get("rings", []):
        add(r)
    fx = load("scaling.json")                                    # make_golden_scaling.py: every point of a scaling run
    for r in (fx or {}).get("lattices", []):
        add(r, seed=fx.get("seed"))
    return recs


def golden_counts(x, y_total, seed, sweeps):
    """(up, down) the oracle found for the x-column, y_total-row lattice after `sweeps` sweeps at T_c, or None."""
    try:
        return golden_records().get((x, y_total, seed), {}).get(sweeps)
    except (KeyError, TypeError):
        return None


WORKLOADS = {  # name -> (columns, rows per rank at N ranks, "weak" | "strong")
    "config3": (65536, lambda n: 65536, "weak"),
    "config4": (131072, lambda n: 16384, "weak"),
    "strong": (65536, lambda n: 65536 // n, "strong"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--x", type=int, default=0, help="columns (per-GPU slab and total)")
    ap.add_argument("--y", type=int, default=0, help="rows per GPU")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="config3",
                    help="config3 (default, weak scaling): 65536 x 65536 per GPU, the size BASELINE's metric is quoted on; config4 (weak): "
                         "131072 columns x 16384 rows per GPU (BASELINE config 4: 131072^2 over 8 GPUs, the same slab at every N); strong: the "
                         "65536^2 lattice as a whole, 65536 / N rows per GPU (north_star: 'a 65536^2 lattice ... at 1, 2, 4 and 8 GPUs'); "
                         "--x / --y override the slab")
    ap.add_argument("--strip-rows", type=int, default=0)
    ap.add_argument("--layout", choices=["auto", "nibble", "dense", "ballot"], default="auto", help="device layout of the spin arrays")
    ap.add_argument("--ring", choices=["native", "torch"], default="native",
                    help="N > 1: the ring inside libising_hip.so (RCCL on a second stream) or the torch.distributed one")
    ap.add_argument("--transport", choices=["auto", "rccl", "ipc"], default="auto",
                    help="N > 1, native ring: the peer transport over hipIpcMemHandle (primary: its exchange runs inside the launch), RCCL send/recv "
                         "(fallback), or (auto) the first of the two, in that order, that comes up on every rank")
    ap.add_argument("--exchange", choices=["p2p", "allgather"], default=None,
                    help="N > 1, torch ring: how the edge rows travel (forces --ring torch)")
    ap.add_argument("--preheat-ms", type=float, default=150.0,
                    help="untimed sweeps before the warm-up until this much wall time has passed (the lattice is initialised "
                         "again afterwards): from a cold start the GPU needs ~35 ms under load to reach its steady clock")
    ap.add_argument("--force-ring", action="store_true",
                    help="N = 1: run the N > 1 code path anyway -- torch.distributed + the library's RCCL ring with a ring of "
                         "ONE slab (its edge rows travel through ncclSend/ncclRecv to itself); what a 1-GPU box can test of it")
    ap.add_argument("--no-counts-leg", action="store_true",
                    help="skip the second timed leg (the same sweeps with the magnetisation read back every 16, as every published "
                         "number of the reference includes them: optimized/main.cu:1806-1810)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-small-leg", action="store_true", help="skip the side leg on the reference's default 2048 x 2048 lattice")
    ap.add_argument("--no-alu-probe", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()
    wx, wy_of, scaling = WORKLOADS[args.workload]
    if args.x or args.y:
        scaling = "weak"  # a slab given by hand is the same slab at every N
    args.x, args.y = args.x or wx, args.y or wy_of(max(1, args.gpus))

    # must be in the environment before the HIP/HSA runtime initialises (RCCL P2P needs dmabuf IPC on this host driver)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start one process per rank ourselves (the contract's own command line, on a free port)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
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
    # more ranks than GPUs (what a 1-GPU box can run of an N-rank job): ranks share devices round robin
    ndev = torch.cuda.device_count()
    shared = world > ndev
    device = local_rank % ndev
    torch.cuda.set_device(device)
    ringed = world > 1 or args.force_ring
    # control plane (ids, blobs, barriers, checksums): RCCL through torch when every rank has a device of its own, gloo otherwise
    ctl = "cpu" if shared else "cuda"
    if ringed:
        if "MASTER_ADDR" not in os.environ:  # --force-ring without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29577"), RANK="0", WORLD_SIZE="1")
        if shared:
            sys.stdout.flush()
            saved = os.dup(1)  # gloo announces its connections on stdout; the contract is ONE JSON line there
            os.dup2(2, 1)
            try:
                dist.init_process_group("gloo")
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    local_rank = device

    layout = {"auto": ig.LAYOUT_AUTO, "nibble": ig.LAYOUT_NIBBLE, "dense": ig.LAYOUT_DENSE, "ballot": ig.LAYOUT_BALLOT}[args.layout]
    log = (lambda m: print(f"[bench rank {rank}] {m}", file=sys.stderr, flush=True))
    attempts = []  # every ring transport tried on this rank, with its error string (first contact with a multi-GPU node must explain itself)
    if not ringed:
        slab = ig.IsingSlab(args.x, args.y, device=local_rank, seed=args.seed, temp=ig.CRIT_TEMP_F32, strip_rows=args.strip_rows, layout=layout)
        ring, ring_name = None, "none"
        slab.init()
    else:
        ring = None
        if args.ring == "native" and not args.exchange:
            # the library's own ring on a slab that owns its buffer: ballot ring slabs then keep ghost rows 64 deep, exchange
            # every 32 sweeps and run fused launches in between (csrc/ising_ring.cpp: sweep_deep)
            slab = ig.IsingSlab(args.x, args.y, device=local_rank, seed=args.seed, temp=ig.CRIT_TEMP_F32, nslabs=world, slab=rank,
                                strip_rows=args.strip_rows, layout=layout, ring_halo=world == 1)
            # (auto: the peer transport first -- its exchange runs inside the launch, DESIGN 5 --, RCCL behind it; ranks that share a device have only the first)
            transports = {"auto": ("ipc",) if shared else ("ipc", "rccl"), "rccl": ("rccl",), "ipc": ("ipc",)}[args.transport]
            ring = ig.open_native_ring(slab, log=log, transports=transports, attempts=attempts)
            if ring is None:
                slab.close()
            else:
                ring_name = ring.exchange
        if ring is None and shared:
            raise SystemExit("bench: ranks share a device and the library's peer ring did not come up (torch's rings need RCCL, which refuses two ranks per device)")
        if ring is None and args.exchange in (None, "p2p"):
            # torch.distributed send/recv on a slab that owns its buffer: the same deep schedule (ghost rows, one exchange per
            # 32 sweeps: SlabRing._sweep_deep), the rows wrapped zero-copy as tensors
            slab = ig.IsingSlab(args.x, args.y, device=local_rank, seed=args.seed, temp=ig.CRIT_TEMP_F32, nslabs=world, slab=rank,
                                strip_rows=args.strip_rows, layout=layout, ring_halo=world == 1)
            try:
                ring, ring_name = ig.open_ring(ig.HipSlabBackend(slab), prefer="torch", exchange="p2p", log=log, attempts_out=attempts)
                if ring.ghost_rows > 1:
                    ring_name += f"-ghost{ring.ghost_rows}"
            except ig.IsingError as e:  # (every rank gets here together: open_ring agrees on each attempt's outcome)
                log(f"torch ring on a library-owned slab did not come up: {e}")
                slab.close()
        if ring is None:
            # torch owns the slab's device buffer, so the rows the torch ring hands to RCCL are slices of an ordinary tensor
            backend = ig.HipSlabBackend.create(args.x, args.y, device=local_rank, seed=args.seed, temp=ig.CRIT_TEMP_F32,
                                               nslabs=world, slab=rank, strip_rows=args.strip_rows, layout=layout, ring_halo=world == 1)
            slab = backend.slab
            ring, ring_name = ig.open_ring(backend, prefer="torch", exchange=args.exchange, log=log, attempts_out=attempts)

    # sweeps per ising_sweep call: the largest common divisor of steps and warm-up that one fused launch can carry (32), so
    # that every launch of the run is the same piece of work -- unless that leaves pieces under 16 sweeps (the driver's
    # --steps 20 --warmup 5): then each phase is cut on its own (a fused launch costs ~60 us whatever it carries)
    def largest_piece(n):
        return max(d for d in range(1, 33) if n % d == 0)
    batch = largest_piece(math.gcd(args.steps, args.warmup) if args.warmup else args.steps)
    batch_warm = batch
    if batch < 16:
        batch, batch_warm = largest_piece(args.steps), (largest_piece(args.warmup) if args.warmup else 1)

    def advance(n, piece=None):
        """n sweeps in pieces of `piece` (default: the timed phase's), all asynchronous"""
        piece = piece or batch
        if ring is not None:
            ring.sweep(n)
            return
        for _ in range(n // piece):
            slab.sweep(piece)
        if n % piece:
            slab.sweep(n % piece)

    def restart():
        if ring is not None:
            ring.init()
        else:
            slab.init()

    def barrier():
        if ringed:
            dist.barrier()
        torch.cuda.synchronize()

    # clock warm-up (untimed, then forgotten: the lattice starts over)
    preheat_sweeps = 0
    if args.preheat_ms > 0:
        barrier()
        t0 = time.perf_counter()
        advance(batch)
        torch.cuda.synchronize()
        more = max(0, math.ceil(args.preheat_ms / max((time.perf_counter() - t0) * 1e3, 1e-3)) - 1)
        if ringed:  # every rank must do the same number of sweeps: rank 0's estimate counts
            t = torch.tensor([more], dtype=torch.int64, device=ctl)
            dist.broadcast(t, src=0)
            more = int(t[0])
        more = min(more, 4096)
        for _ in range(more):
            advance(batch)
        preheat_sweeps = (1 + more) * batch
        restart()
    elif ring is not None:
        restart()  # (opening a ring runs one sweep through the transport before it is trusted: start over)

    # the library's ring reports where a slab's time goes around its exchanges (ising_exchange_stats_*: HIP events on the launches'
    # dispatch packets and on the comm stream); sampled over the timed leg on every rank
    stats_slab = ring.slab if isinstance(ring, ig.NativeRing) else None
    clock_slab = slab if ring is None else stats_slab  # the library's own launches leave the marks their shader clock is computed from
    if clock_slab is not None:
        clock_slab.kernel_clock(True)
    advance(args.warmup, batch_warm)
    barrier()
    if stats_slab is not None:
        stats_slab.exchange_stats_begin(256)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # HIP events on the stream the kernels are launched on (torch's current stream = the slab's stream)
    advance(args.steps)
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    if ringed:
        t = torch.tensor([dt, ev_ms], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, ev_ms = float(t[0]), float(t[1])
    sclk_kernel = None
    if clock_slab is not None:
        try:
            sclk_kernel = clock_slab.kernel_clock_fetch()  # (mean, min, max) MHz over the XCDs of the timed leg's LAST fused launch
        except ig.IsingError:
            sclk_kernel = None                             # (no fused launch: one launch per colour, tile launches)
        clock_slab.kernel_clock(False)
    xstats = None
    if stats_slab is not None:
        mine = stats_slab.exchange_stats_fetch()
        per_rank = [mine]
        if world > 1:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
        xstats = {"exchanges_per_rank": per_rank[0]["exchanges"], "launches_per_rank": per_rank[0]["launches"],
                  "what": "per exchange of 64 ghost rows of both colours (RCCL: one per fused launch; the peer (IPC) transport with a device per rank: several exchange "
                          "epochs inside ONE persistent launch -- launches_per_rank < exchanges_per_rank --, the figures below then describe each launch's LAST exchange): "
                          "launch = the fused launch; exchange = edge strips of "
                          "the launch done -> neighbours' rows in place; go_after_end = the exchange's end relative to the END of the launch whose rows "
                          "it carries (negative: hidden in the launch's tail; positive: the next launch waited that long -- a slow link or a "
                          "neighbour that is behind); gap = end of a launch -> start of the next.  ms; mean over ranks of the per-rank means, max over everything"}
        for k in ("launch_ms", "exchange_ms", "go_after_end_ms", "gap_ms"):
            xstats[k] = {"mean": round(sum(r[k + "_mean"] for r in per_rank) / len(per_rank), 4), "max": round(max(r[k + "_max"] for r in per_rank), 4),
                         "by_rank_mean": [r[k + "_mean"] for r in per_rank]}

    up, down = ring.count() if ring is not None else slab.count()
    rank_up = [slab.count()[0]]
    if ringed:
        g = [None] * world
        dist.all_gather_object(g, rank_up[0])
        rank_up = g
    spins_per_gpu = args.x * args.y
    total_flips = float(spins_per_gpu) * world * args.steps
    value = total_flips / (dt * 1e9)
    gold = golden_counts(args.x, args.y * world, args.seed, args.warmup + args.steps)
    parity = None if gold is None else (gold == (up, down))

    # Second timed leg, the reference's methodology: every number it publishes was measured with the magnetisation read back
    # inside the timed loop (-p 16: countSpins every 16 sweeps and after the last one, optimized/main.cu:1806-1810, :1862-1874;
    # BASELINE.md).  Same lattice, same sweeps, from the start; the final counts must be the first leg's.
    def counted_leg(energy):
        """one more timed leg over the same sweeps with the print points inside the timed region; energy: the bond sum at every point too"""
        restart()
        advance(args.warmup, batch_warm)
        if ring is None:
            slab.sweep_counted(0, 16, energy)  # (a call of no sweeps: the slots of the in-launch counts are allocated outside the timed region, like the lattice)
        elif hasattr(ring, "sweep_counted"):
            ring.sweep_counted(0, 16, energy)
        barrier()
        t0 = time.perf_counter()
        ncounts, done, last, series = 0, 0, None, None
        if ring is None:
            # a lone slab: the print points ride INSIDE the fused launches (ising_sweep_counted: the units that store the words count
            # them) -- every iteration that is a multiple of 16, as the reference's loop prints -- and the final count as ever
            series = slab.sweep_counted(args.steps, 16, energy)
            ncounts = len(series) + 1
            last = slab.count()
        elif hasattr(ring, "sweep_counted"):
            # the library's ring: every rank's deep launches count their own rows (ising_rank_sweep_counted), the sums travel over the rank transport
            try:
                series = ring.sweep_counted(args.steps, 16, energy)
                ncounts = len(series) + 1
            except Exception as e:  # noqa: BLE001  (a library error here must not cost the line its first leg: every rank returns the same outcome, the call agrees on it)
                log(f"bench: counts leg: {e}; sweeping and counting in turn instead")
                restart()
                advance(args.warmup, batch_warm)
                barrier()
                t0 = time.perf_counter()
                while done < args.steps:
                    n = min(16, args.steps - done)
                    advance(n, min(batch, n))
                    ring.count()
                    ncounts += 1
                    done += n
            last = ring.count()
        else:
            while done < args.steps:
                n = min(16, args.steps - done)
                advance(n, min(batch, n))
                last = ring.count()  # (blocking: the ranks' counters come back and are summed, as :860-866)
                ncounts += 1
                done += n
        barrier()
        dt2 = time.perf_counter() - t0
        if ringed:
            t = torch.tensor([dt2], dtype=torch.float64, device=ctl)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t[0])
        leg = {"value": round(total_flips / (dt2 * 1e9), 2), "unit": "flips/ns", "ms_per_step": round(dt2 * 1e3 / args.steps, 5),
               "counts_in_timed_region": ncounts, "final_counts_equal_first_leg": last == (up, down),
               "what": "the same steps with the up/down counts" + (" and the bond sum (energy)" if energy else "") + " of every iteration that is a multiple of 16, and the counts "
                       "of the last one, inside the timed region (the reference's -p 16 methodology, optimized/main.cu:1806-1810"
                       + ("; the energy series is north_star's, the reference computes none" if energy else "") + ")"
                       + ("; N = 1: taken inside the fused launches (ising_sweep_counted), read back with the last" if ring is None else
                          ("; the library's ring: taken inside every rank's deep launches (ising_rank_sweep_counted), summed over the rank transport"
                           if hasattr(ring, "sweep_counted") else "; rings: a blocking count of all ranks every 16 sweeps"))}
        if energy and series:
            n_tot = float(spins_per_gpu) * world
            leg["last_point"] = {"iteration": (args.warmup + args.steps) // 16 * 16, "up": series[-1][0], "bond_equal": series[-1][2],
                                 "energy_per_spin": round(-(2.0 * series[-1][2] - 2.0 * n_tot) / n_tot, 9)}
        return leg

    # Further timed legs, the reference's methodology: every number it publishes was measured with the magnetisation read back
    # inside the timed loop (-p 16: countSpins every 16 sweeps and after the last one, optimized/main.cu:1806-1810, :1862-1874;
    # BASELINE.md).  Same lattice, same sweeps, from the start; the final counts must be the first leg's.  The third leg adds
    # north_star's energy series at the same points.
    counts_leg = energy_leg = None
    if not args.no_counts_leg:
        counts_leg = counted_leg(False)
        if ring is None or hasattr(ring, "sweep_counted"):
            energy_leg = counted_leg(True)

    # First contact with a multi-GPU node: who owns which device, what each transport attempt said, and what this slab shape does
    # as a LONE slab on this rank's device in this job -- so that the line reads as good or bad without a second run.
    first_contact = None
    if ringed:
        first_contact = first_contact_block(args, ig, torch, dist, world, rank, device, ndev, shared, attempts, layout, ctl, value, xstats)

    layout_name, layout_text = {
        ig.LAYOUT_NIBBLE: ("nibble", "reference 4 bit/spin"),
        ig.LAYOUT_DENSE: ("dense", "dense 1 bit/spin"),
        ig.LAYOUT_BALLOT: ("ballot", "1 bit/spin in wave-ballot order"),
    }[slab.current_layout()]
    if rank == 0:
        # dominant kernel = the update kernel.  N = 1 on the ballot layout: fused launches of `batch` sweeps; otherwise two
        # full-slab launches per step (with N > 1 each colour adds one tiny edge-row launch).
        fused = layout_name == "ballot" and slab.fused
        if fused and ringed:  # the library's ring on ghost rows: fused launches of up to max_sweeps_per_launch sweeps between exchanges
            per = max(1, slab.max_sweeps_per_launch)
            launches = (args.steps + per - 1) // per
            if xstats is not None and 0 < xstats["launches_per_rank"] < launches:  # (several exchange epochs per persistent launch: the launches the library really issued)
                launches = xstats["launches_per_rank"]
            half_sweeps_per_launch = 2.0 * args.steps / launches
        else:
            half_sweeps_per_launch = 2 * batch if fused else 1
            launches = args.steps // batch + (1 if args.steps % batch else 0) if fused else 2 * args.steps
        avg_launch_ms = ev_ms / launches
        kernel = {"ballot": "ballot_update_k", "dense": "dense_update_k", "nibble": "update_k"}[layout_name] + ("<fused>" if fused else "")
        # (a) the reference's accounting, SURVEY 8(d): 1.5 B per flip = source read + destination read + write at 4 bit/spin
        alg_bytes_per_launch = BYTES_PER_FLIP * spins_per_gpu / 2.0 * half_sweeps_per_launch
        hbm_ref = alg_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
        hbm_reference = {"achieved": round(hbm_ref, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_ref / HBM_PEAK_GBS, 4),
                         "bytes_per_flip": BYTES_PER_FLIP, "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                         "what": "the reference's own accounting (optimized/main.cu:1887-1889): 1.5 B per flip at its 4 bit/spin, whatever the device stores"}
        # (b) what the device really moves: HBM bytes per launch from the PMC passes of tools/profile.sh (separate runs; FETCH_SIZE /
        # WRITE_SIZE per the guide), at this launch shape
        hbm_real, traffic = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tj.get("x") == args.x and tj.get("y") == args.y and tj.get("device_layout") == layout_name and tj.get("fused", False) == fused:
                traffic = tj["hbm_bytes_per_half_sweep"] * half_sweeps_per_launch
                same = tj.get("half_sweeps_per_launch") == half_sweeps_per_launch
                real = traffic / (avg_launch_ms * 1e-3) / 1e9
                hbm_real = {"achieved": round(real, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(real / HBM_PEAK_GBS, 4),
                            "traffic_bytes_per_launch": traffic,
                            "device_algorithmic_bytes_per_launch": tj.get("device_bytes_algorithmic_per_half_sweep", 0) * half_sweeps_per_launch,
                            "source": tj.get("tag", "profiles/traffic.json") + (
                                " (PMC passes of an earlier run of this launch shape, not this run)" if same else
                                f" (PMC passes of an earlier run with {tj.get('half_sweeps_per_launch')} colour half-sweeps per launch, scaled per half-sweep)")}
        except (OSError, ValueError, KeyError):
            pass
        # (c) the roof that binds: the vector ALU.  One 32-bit Philox4x32-10 output per site is forced by bit-exact parity; a kernel that
        # only draws (same job, same chip, same clocks) is the ceiling.
        sites_per_launch = spins_per_gpu / 2.0 * half_sweeps_per_launch
        kern_rate = sites_per_launch / (avg_launch_ms * 1e6)  # sites/ns per GPU inside the kernel
        ceil, ceil_err, sclk_ceiling = None, None, None
        if not args.no_alu_probe:
            try:
                ceil, sclk_ceiling = ig.philox_ceiling_clocked(local_rank, 25.0)  # an average over >= 25 ms of launches, like the kernel it is compared with
            except ig.IsingError as e:
                ceil_err = str(e)
        # `frac` is the CONTRACT's number: algorithmic bytes per launch (1.5 B per flip x the flips of one launch) / the kernel's average launch
        # duration (HIP events on the launches' stream) / 8 TB/s.  `bound` says which roof binds in fact -- the vector ALU: one 32-bit Philox4x32-10 output per
        # site is forced by bit-exact parity --, and the ratio to a kernel that only draws (same job, same chip, same clocks) rides along as frac_valu_ceiling.
        roof = {"bound": "valu" if ceil else "hbm", "achieved": hbm_reference["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_reference["frac"],
                "frac_what": "SURVEY 8(d): 1.5 B per flip (the reference's accounting, optimized/main.cu:1887-1889) x flips per launch / average launch duration / 8 TB/s"}
        if ceil:
            roof.update({"frac_valu_ceiling": round(kern_rate / ceil, 4), "valu_ceiling_sites_ns": round(ceil, 1), "kernel_sites_ns": round(kern_rate, 1),
                         "frac_valu_ceiling_of_value": round(value / world / ceil, 4),
                         "valu_ceiling_what": "draw-only kernel measured in this job (ising_philox_ceiling: Philox4x32-10, one 32-bit output per site exactly as the "
                                              "update kernels draw them, no accept test, no lattice, no memory traffic)",
                         "note": "the kernel is bound by the vector ALU, not by HBM (SQ_ACTIVE_INST_VALU x waves per SIMD ~ 1, real HBM use under a fifth of peak): "
                                 "`frac` is the contract's HBM accounting, `frac_valu_ceiling` the fraction of the roof that binds, hbm_real the device's real traffic"})
        elif ceil_err:
            roof["alu_probe_error"] = ceil_err
        # SURVEY 8(d)'s figures as SCALAR keys (the harness keeps only those of `roofline`): the reference's 1.5 B/flip accounting against 8 TB/s, the
        # device's real HBM rate, real traffic over the device's algorithmic bytes; and the shader clocks of the two kernels `frac` compares
        roof.update({"frac_hbm_1p5B": hbm_reference["frac"], "hbm_achieved_gbs": hbm_reference["achieved"], "hbm_peak_gbs": HBM_PEAK_GBS,
                     "hbm_real_gbs": hbm_real["achieved"] if hbm_real else None, "hbm_real_frac": hbm_real["frac"] if hbm_real else None,
                     "traffic_over_algorithmic": (round(hbm_real["traffic_bytes_per_launch"] / hbm_real["device_algorithmic_bytes_per_launch"], 4)
                                                  if hbm_real and hbm_real["device_algorithmic_bytes_per_launch"] else None),
                     "sclk_mhz_kernel": round(sclk_kernel[0], 1) if sclk_kernel else None,
                     "sclk_mhz_kernel_min_xcd": round(sclk_kernel[1], 1) if sclk_kernel else None,
                     "sclk_mhz_kernel_max_xcd": round(sclk_kernel[2], 1) if sclk_kernel else None,
                     "sclk_mhz_ceiling": round(sclk_ceiling, 1) if sclk_ceiling else None,
                     "frac_at_equal_clock": (round((kern_rate / sclk_kernel[0]) / (ceil / sclk_ceiling), 4) if ceil and sclk_kernel and sclk_ceiling else None),
                     "sclk_what": "shader cycles / time of the 100 MHz counter, both read inside the kernel by one wave per XCD: the timed leg's last fused launch, "
                                  "and the last of the draw-only kernel's launches (an average over >= 25 ms of them)"})
        roof.update({"traffic": traffic, "kernel": kernel, "avg_launch_ms": round(avg_launch_ms, 5), "launches": launches,
                     "half_sweeps_per_launch": half_sweeps_per_launch, "sites_per_launch": sites_per_launch,
                     "hbm_reference_accounting": hbm_reference, "hbm_real": hbm_real})
        line = {
            "metric": "spin-flips/ns (whole node) at T=Tc", "value": round(value, 2), "unit": "flips/ns",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt * 1e3 / args.steps, 5), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {args.y * world}x{args.x} lattice ({args.y}x{args.x} per GPU), T=Tc, seed {args.seed}, "
                                   "Philox4x32-10 per site; device layout " + layout_text
                                   + ", results identical to the reference's packed state", "x": args.x, "y_per_gpu": args.y, "y_total": args.y * world,
                       "parallelism": f"slab{world}", "nranks": world, "physical_gpus": min(world, ndev), "exchange": ring_name, "exchange_overlap": exchange_overlap(ring_name), "strip_rows": slab.strip_rows,
                       "device_layout": layout_name, "sweeps_per_call": batch, "warmup_sweeps_per_call": batch_warm, "preheat_ms": args.preheat_ms, "preheat_sweeps": preheat_sweeps,
                       "up": up, "down": down, "rank_up": rank_up, "parity_checked": parity,
                       "parity_source": None if gold is None else "tests/golden (CPU oracle on the whole lattice, same seed, same number of sweeps; any decomposition)"},
            "roofline": roof,
        }
        if counts_leg is not None:
            line["with_counts_every_16"] = counts_leg
        if energy_leg is not None:
            line["with_counts_and_energy_every_16"] = energy_leg
        if xstats is not None:
            line["exchange_stats"] = xstats
        if first_contact is not None:
            line.update(first_contact)
        if shared:  # more ranks than devices: `value` is what the physical GPUs delivered together, not a scaling point
            line["config"]["note"] = (f"{world} ranks share {ndev} physical GPU(s): the N > 1 path (processes, peer transport, ring schedule) executed and "
                                      "checked, not a scaling measurement")
        if parity is False:
            line["config"]["parity_expected"] = list(gold)
        if not ringed and not args.no_small_leg:
            try:
                leg2048 = default_lattice_leg(args, local_rank)
                leg2048["frac"] = leg2048["frac_hbm_1p5B"]  # (SURVEY 8(d)'s number for this lattice, as roofline.frac is for the headline's)
                leg2048["frac_of_plateau"] = round(leg2048["value"] / (value / world), 4)  # against the large-lattice rate of the same job
                line["default_lattice_2048"] = leg2048
                try:
                    legb = batched_small_leg(args, local_rank, leg2048["up_down"])
                    legb["frac_of_plateau"] = round(legb["value"] / (value / world), 4)
                    line["batched_31x2048"] = legb
                except Exception as e:  # noqa: BLE001
                    line["batched_31x2048"] = {"error": str(e)}
            except Exception as e:  # noqa: BLE001  (a side leg must not cost the line)
                line["default_lattice_2048"] = {"error": str(e)}
        if not ringed and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)

    if ringed:
        dist.barrier()
        dist.destroy_process_group()
    if parity is False:
        raise SystemExit(f"bench: counts {(up, down)} differ from the oracle's {gold}")
    for leg in (counts_leg, energy_leg):
        if leg is not None and not leg["final_counts_equal_first_leg"]:
            raise SystemExit("bench: a leg with print points every 16 sweeps ended on other counts than the first leg")


if __name__ == "__main__":
    main()
