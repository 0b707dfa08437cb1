"""Summarise rocprofv3 rocpd databases (tools/profile.sh output) into a small text + JSON report.

Usage: python tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<name> [profiles/traffic.json]
(writes <name>.txt and <name>.json; with a third argument also the traffic file bench.py reads)
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB and are
collected in separate passes; on gfx950 FETCH_SIZE tallies 128-B requests of wide coalesced streams as 64 B, so
the read side is doubled ("corrected") -- both raw and corrected figures are reported.
"""
import glob
import json
import os
import sqlite3
import sys


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return con.execute(sql).fetchall()
    finally:
        con.close()


def bench_line(path):
    if os.path.exists(path):
        for ln in open(path):
            if ln.startswith("{"):
                return json.loads(ln)
    return None


def main():
    src, dst = sys.argv[1], sys.argv[2]
    lines = []
    out = {}
    tr = glob.glob(os.path.join(src, "trace", "*.db"))
    if tr:
        b = bench_line(os.path.join(src, "bench_under_rocprof.json"))
        hs = (b or {}).get("roofline", {}).get("half_sweeps_per_launch", 1)
        lines.append(f"== kernel-trace --stats (bench.py, default 128 steps + 16 warm-up; {hs} colour half-sweep(s) per update launch), per kernel ==")
        lines.append(f"{'kernel':28s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
        rows = q(tr[0], "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc")
        tot = sum(r[2] for r in rows)
        for n, c, s, a, mn, mx in rows:
            lines.append(f"{n[:28]:28s} {c:6d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.2f}")
            if "update_k" in n or "ballot_split_k" in n:
                out["update_k_avg_us"] = a / 1e3
                out["update_k_calls"] = c
        vg = q(tr[0], "select distinct name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x from kernels where name like '%update_k%' or name like '%ballot_split_k%'")
        for r in vg:
            lines.append(f"   {r[0]}: vgpr {r[1]} agpr {r[2]} sgpr {r[3]} lds {r[4]} scratch {r[5]} grid {r[6]} wg {r[7]}")
    lines.append("")
    lines.append("== PMC passes (bench.py --steps 20 --warmup 5 --preheat-ms 0: the driver's launch shape), the timed launch's update_k dispatch ==")
    pm = {}
    for db in sorted(glob.glob(os.path.join(src, "pmc_*", "*.db"))):
        try:
            # the timed launch only: the longest update_k dispatches (the warm-up launch carries a quarter of its levels)
            rows = q(db, "select counter_name, avg(counter_value), count(*), avg(duration) from pmc_events where (name like '%update_k%' or name like '%ballot_split_k%') "
                         "and duration > 0.6 * (select max(duration) from pmc_events where name like '%update_k%' or name like '%ballot_split_k%') group by counter_name")
        except Exception as e:
            lines.append(f"{db}: {e}")
            continue
        for name, avg, cnt, dur in rows:
            pm[name] = avg
            lines.append(f"{name:24s} avg {avg:18.1f} over {cnt} dispatches (avg duration {dur/1e3:.1f} us under the profiler)")
    pb = None
    for f in sorted(glob.glob(os.path.join(src, "pmc_*.json"))):
        pb = pb or bench_line(f)
    phs = (pb or {}).get("roofline", {}).get("half_sweeps_per_launch", 1)
    if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
        rd, wr = pm["FETCH_SIZE"] * 1024.0, pm["WRITE_SIZE"] * 1024.0
        out.update(fetch_bytes_raw=rd, fetch_bytes_corrected=2 * rd, write_bytes=wr, half_sweeps_per_launch=phs,
                   hbm_bytes_per_launch=2 * rd + wr, hbm_bytes_per_launch_uncorrected=rd + wr,
                   hbm_bytes_per_half_sweep=(2 * rd + wr) / phs)
        lines.append("")
        lines.append(f"HBM per update_k launch ({phs} colour half-sweep(s)): FETCH_SIZE {rd/1e9:.3f} GB raw (x2 gfx950 correction = {2*rd/1e9:.3f} GB), "
                     f"WRITE_SIZE {wr/1e9:.3f} GB -> {(2*rd+wr)/1e9:.3f} GB corrected ({(rd+wr)/1e9:.3f} GB uncorrected) "
                     f"= {(2*rd+wr)/phs/1e9:.3f} GB per colour half-sweep")
    if "SQ_WAVE_CYCLES" in pm and "SQ_BUSY_CYCLES" in pm:
        lines.append(f"VALU instructions per wave: {pm.get('SQ_INSTS_VALU', 0)/max(pm.get('SQ_WAVES', 1), 1):.1f}; "
                     f"SALU per wave: {pm.get('SQ_INSTS_SALU', 0)/max(pm.get('SQ_WAVES', 1), 1):.1f}")
        wc = pm["SQ_WAVE_CYCLES"]
        for k in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in pm:
                lines.append(f"{k}/SQ_WAVE_CYCLES = {pm[k]/wc:.3f}")
    if "TCC_HIT_sum" in pm:
        lines.append(f"L2 hit rate: {pm['TCC_HIT_sum']/(pm['TCC_HIT_sum']+pm['TCC_MISS_sum']):.3f}")
    out["pmc"] = pm
    bj = os.path.join(src, "bench_under_rocprof.json")
    if os.path.exists(bj):
        for ln in open(bj):
            if ln.startswith("{"):
                b = json.loads(ln)
                out["x"], out["y"] = b["config"]["x"], b["config"]["y_per_gpu"]
                lines.append("")
                lines.append("bench.py line under the profiler: " + ln.strip())
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    open(dst + ".txt", "w").write("\n".join(lines) + "\n")
    json.dump(out, open(dst + ".json", "w"), indent=1)
    if len(sys.argv) > 3 and pb and "hbm_bytes_per_half_sweep" in out:
        cfg, roof = pb["config"], pb["roofline"]
        spins = cfg["x"] * cfg["y_per_gpu"]
        bits = 4 if cfg["device_layout"] == "nibble" else 1
        tj = {"x": cfg["x"], "y": cfg["y_per_gpu"], "device_layout": cfg["device_layout"], "fused": roof["kernel"].endswith("<fused>"),
              "tag": dst + ".txt", "hbm_bytes_per_half_sweep": out["hbm_bytes_per_half_sweep"],
              "fetch_bytes_raw_per_launch": out["fetch_bytes_raw"], "write_bytes_per_launch": out["write_bytes"],
              "half_sweeps_per_launch": phs,
              "device_bytes_algorithmic_per_half_sweep": 3 * (spins // 2) * bits // 8,
              "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), KiB -> bytes, FETCH_SIZE doubled per the gfx950 "
                      "note in MI355X_MICROARCH.md (HBM section); the timed launch of bench.py --steps 20 --warmup 5 --preheat-ms 0 (the driver's command)"}
        json.dump(tj, open(sys.argv[3], "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
