#!/bin/bash
# rocprofv3 passes for the headline workload (run on the GPU box through gpurun).  Usage: tools/profile.sh <tag>
# Writes CSV/summary files under gpurun_out/prof_<tag>/; copy what should be judged into profiles/.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline"   # the default bench command: 128 steps, 16 warm-up
SHORT="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -S -T -d $OUT/trace -o bench -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/trace_stderr.txt
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace -T --pmc $C -d $OUT/pmc_$N -o pmc -- $SHORT > $OUT/pmc_$N.json 2> $OUT/pmc_$N.stderr.txt || echo "pmc pass $C failed" >> $OUT/errors.txt
done
find $OUT -name "*.csv" | head -50
