"""Summarise the rocprofv3 databases of tools/profile_cmd.sh (any command: the CLI on BASELINE configs 2 and 5).
Usage: python tools/summarize_cmd_prof.py gpurun_out/prof_<tag> profiles/<name>.txt <spins per lattice> [sweeps x lattices]
Per kernel: calls, total, average; for the update kernel: HBM bytes (FETCH_SIZE x 2 per the gfx950 note of
MI355X_MICROARCH.md + WRITE_SIZE, KiB -> bytes, separate passes) summed over ALL its dispatches, against the 1 bit/spin
algorithmic bytes and the reference's 1.5 B/flip accounting of the same sweeps."""
import glob
import os
import sqlite3
import sys


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return con.execute(sql).fetchall()
    finally:
        con.close()


def main():
    src, dst, spins = sys.argv[1], sys.argv[2], float(sys.argv[3])
    sweeps = float(sys.argv[4]) if len(sys.argv) > 4 else None  # lattice sweeps of the whole run (all lattices)
    L = ["command: " + open(os.path.join(src, "command.txt")).read().strip(), ""]
    tr = glob.glob(os.path.join(src, "trace", "*.db"))
    upd_ns = None
    if tr:
        rows = q(tr[0], "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc")
        tot = sum(r[2] for r in rows)
        L.append("== rocprofv3 --kernel-trace --stats, per kernel ==")
        L.append(f"{'kernel':34s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
        for n, c, s, a, mn, mx in rows:
            L.append(f"{n[:34]:34s} {c:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.2f}")
            if ("update_k" in n or "tile_k" in n or "quad_pass_k" in n) and upd_ns is None:
                upd_ns = s
        for r in q(tr[0], "select distinct name, vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x from kernels where name like '%update_k%' or name like '%tile_k%' or name like '%quad_pass_k%'"):
            L.append(f"   {r[0]}: vgpr {r[1]} sgpr {r[2]} lds {r[3]} grid {r[4]} wg {r[5]}")
        L.append("")
    pm = {}
    for db in sorted(glob.glob(os.path.join(src, "pmc_*", "*.db"))):
        for name, tot, cnt, dur in q(db, "select counter_name, sum(counter_value), count(*), sum(duration) from pmc_events where name like '%update_k%' or name like '%tile_k%' or name like '%quad_pass_k%' group by counter_name"):
            pm[name] = (tot, cnt, dur)
            L.append(f"{name:22s} sum {tot:20.1f} over {cnt} update_k / tile_k / quad_pass_k dispatches ({dur/1e6:.2f} ms under the profiler)")
    if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
        rd, wr = pm["FETCH_SIZE"][0] * 1024.0, pm["WRITE_SIZE"][0] * 1024.0
        hbm = 2 * rd + wr
        L.append("")
        L.append(f"HBM bytes of all update_k dispatches: FETCH_SIZE {rd/1e9:.3f} GB raw (x2 gfx950 correction = {2*rd/1e9:.3f} GB) + WRITE_SIZE {wr/1e9:.3f} GB = {hbm/1e9:.3f} GB")
        if sweeps:
            alg1 = 3 * spins / 8 * sweeps  # per sweep: both colours, (source read + destination read + write) at 1 bit per spin
            L.append(f"  {sweeps:.0f} lattice sweeps of {spins:.0f} spins: {hbm/sweeps/2/1e6:.3f} MB per colour half-sweep; algorithmic at 1 bit/spin {alg1/1e9:.3f} GB "
                     f"-> measured / algorithmic = {hbm/alg1:.3f}; the reference's 1.5 B/flip accounting {1.5*spins*sweeps/1e9:.1f} GB")
            if upd_ns:
                L.append(f"  update_k time (trace pass) {upd_ns/1e6:.2f} ms -> {spins*sweeps/upd_ns:.1f} flips/ns in the kernel, {1.5*spins*sweeps/upd_ns:.1f} GB/s by the reference's accounting "
                         f"= {1.5*spins*sweeps/upd_ns/8000:.3f} of 8 TB/s; real HBM rate {hbm/upd_ns:.1f} GB/s")
    if "SQ_WAVE_CYCLES" in pm and "SQ_ACTIVE_INST_VALU" in pm:
        L.append(f"SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {pm['SQ_ACTIVE_INST_VALU'][0]/pm['SQ_WAVE_CYCLES'][0]:.3f}; VALU instructions per wave {pm['SQ_INSTS_VALU'][0]/max(pm['SQ_WAVES'][0],1):.0f}")
    if "TCC_HIT_sum" in pm:
        L.append(f"L2 hit rate {pm['TCC_HIT_sum'][0]/(pm['TCC_HIT_sum'][0]+pm['TCC_MISS_sum'][0]):.3f}")
    run = os.path.join(src, "run_under_rocprof.txt")
    if os.path.exists(run):
        tail = [ln for ln in open(run).read().splitlines() if "flips/ns" in ln]
        L += ["", "the command's own timing line(s) under the profiler:"] + tail[-3:]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    open(dst, "w").write("\n".join(L) + "\n")
    print("\n".join(L))


if __name__ == "__main__":
    main()
