#!/usr/bin/env python3
"""Where the deep exchange sits in time: a ring of one slab (its own first / last 64 rows travel through the transport) under
rocprofv3 --kernel-trace.
  run:      rocprofv3 --kernel-trace -d DIR -o trace -- python tools/overlap_trace.py rccl|ipc [X Y sweeps]
  analyze:  python tools/overlap_trace.py --analyze DIR
For every fused launch of the update kernel: the kernels of the OTHER stream (the wait for the launch's edge strips, RCCL's
send/recv kernel or the peer copies, the counters) that start inside the launch's interval, relative to the launch's end --
with the overlapped schedule the whole exchange lies inside the interval of the launch whose rows it carries."""
import glob
import os
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])

if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
    db = glob.glob(os.path.join(sys.argv[2], "**", "*.db"), recursive=True)[0]
    rows = sqlite3.connect(db).execute("select name, start, end, grid_x, stream_id from kernels order by start").fetchall()
    big = [r for r in rows if "ballot_update_k" in r[0] and (r[2] - r[1]) > 2e6]  # fused launches of several sweeps
    t0 = big[0][1]
    inside_all, outside_all = 0, 0
    for i, b in enumerate(big):
        print(f"fused launch {i}: {(b[1] - t0) / 1e3:10.1f} .. {(b[2] - t0) / 1e3:10.1f} us ({(b[2] - b[1]) / 1e3:.1f} us), stream {b[4]}")
        nxt = big[i + 1][1] if i + 1 < len(big) else b[2] + 2_000_000
        for r in rows:
            if r is b or "ballot_update_k" in r[0] and (r[2] - r[1]) > 2e6:
                continue
            if b[1] <= r[1] < nxt and r[4] != b[4]:
                where = "inside the launch" if r[2] <= b[2] else ("straddles its end" if r[1] < b[2] else "after it")
                inside_all += where == "inside the launch"
                outside_all += where != "inside the launch"
                print(f"      {r[0][:56]:56s} {(r[1] - b[2]) / 1e3:+10.1f} .. {(r[2] - b[2]) / 1e3:+10.1f} us from the launch's end  ({where})")
    for name, v, sg, lds, wg in sqlite3.connect(db).execute("select distinct name, vgpr_count, sgpr_count, lds_size, workgroup_x from kernels where name like '%rccl%' or name like '%ballot_update_k%' or name like '%copyBuffer%' or name like '%counter_%' or name like '%ipc_%'").fetchall():
        print(f"   registers: {name[:60]:60s} vgpr {v} sgpr {sg} lds {lds} workgroup {wg}")
    gaps = [(b2[1] - b1[2]) / 1e3 for b1, b2 in zip(big, big[1:])]
    print(f"{len(big)} fused launches; gaps between consecutive ones (us): {[round(g, 1) for g in gaps]}")
    print(f"kernels of the comm stream inside a launch's interval: {inside_all}, straddling its end or after it: {outside_all}")
    sys.exit(0)

import torch  # noqa: E402,F401
import ising_gpu_amd as ig  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "rccl"
X, Y, sweeps = (int(v) for v in (sys.argv[2:5] if len(sys.argv) >= 5 else (65536, 65536, 128)))
with ig.IsingSlab(X, Y, seed=1, temp=ig.CRIT_TEMP_F32, ring_halo=True) as s:
    ring = ig.NativeRing(s, transport=mode).init()
    ring.sweep(32)
    ring.quiesce()
    ring.sweep(sweeps)
    ring.quiesce()
    print("counts", ring.count())
    ring.close()
