#!/bin/bash
# Durations of the quad path's launches by kind (rocprofv3 --kernel-trace): calls of ONE pass each are a draws-only launch + a words-only launch; a long call's
# launches carry both.  Usage: tools/quad_launch_times.sh X Y C T WAVES
X=$1; Y=$2; C=$3; T=$4; W=$5
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/qlt_${X}_${Y}_${C}_${T}_${W}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ISING_QUAD=1 ISING_QUAD_C=$C ISING_QUAD_T=$T ISING_QUAD_WAVES=$W
rocprofv3 --kernel-trace -f csv -d $OUT/one -o run -- python $R/tools/quad_run.py $X $Y $T 200 > /dev/null 2> $OUT/one.err
rocprofv3 --kernel-trace -f csv -d $OUT/long -o run -- python $R/tools/quad_run.py $X $Y $((T * 200)) 2 > /dev/null 2> $OUT/long.err
python3 - <<PY
import csv, glob, statistics
for tag in ("one", "long"):
    f = glob.glob("$OUT/" + tag + "/**/*kernel_trace.csv", recursive=True)
    if not f:
        print(tag, "no trace"); continue
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0 for r in csv.DictReader(open(f[0])) if "quad_pass_k" in r["Kernel_Name"]]
    d = d[len(d) // 4:]  # (past the warm-up)
    if tag == "one":
        a, b = d[0::2], d[1::2]
        print(f"$Y x $X ($C, $T, $W) one pass a call: launches alternate {statistics.median(a):.2f} us / {statistics.median(b):.2f} us (draws only / words only, in launch order)")
    else:
        print(f"$Y x $X ($C, $T, $W) long calls: {statistics.median(d):.2f} us a launch of $T sweeps (words + draws), {len(d)} launches")
PY
