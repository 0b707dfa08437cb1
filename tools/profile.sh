#!/bin/bash
# rocprofv3 passes for the headline workload (run on the GPU box through gpurun).  Usage: tools/profile.sh <tag> [bench args]
# Writes rocpd databases and the bench lines under gpurun_out/prof_<tag>/; tools/summarize_prof.py turns them into the
# text + JSON summaries that are committed under profiles/.
# Counter passes run on their own (never together with --stats / system traces) and one counter group per run, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes for FETCH_SIZE / WRITE_SIZE.
TAG=${1:-r02}
shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-alu-probe --no-counts-leg $*"   # the default bench command: 128 steps, 16 warm-up
# counter passes: the driver's own launch shape -- bench.py --steps 20 --warmup 5 = a warm-up launch of 10 colour half-sweeps and the
# timed launch of 40; tools/summarize_prof.py keeps the timed launch's dispatch (the longest) only
SHORT="python $R/bench.py ${SHORT_STEPS:---steps 20 --warmup 5} --preheat-ms 0 --no-cpu-baseline --no-alu-probe --no-counts-leg $*"   # (one timed launch of 40 colour half-sweeps: the counts leg would add launches of other lengths)
rocprofv3 --kernel-trace --stats -S -T -d $OUT/trace -o bench -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/trace_stderr.txt
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace -T --pmc $C -d $OUT/pmc_$N -o pmc -- $SHORT > $OUT/pmc_$N.json 2> $OUT/pmc_$N.stderr.txt || echo "pmc pass $C failed" >> $OUT/errors.txt
done
find $OUT -name "*.db" | head -50
