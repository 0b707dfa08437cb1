#!/bin/bash
# rocprofv3 passes for ANY command (BASELINE configs 2 and 5 through the CLI).  Usage: tools/profile_cmd.sh <tag> <command ...>
# One --kernel-trace --stats pass, then FETCH_SIZE and WRITE_SIZE in passes of their own (never together with --stats or a
# system trace: /opt/skills/guides/MI355X_MICROARCH.md, HBM section).  tools/summarize_cmd_prof.py turns gpurun_out/prof_<tag>
# into the text summary committed under profiles/.
TAG=$1
shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "$*" > $OUT/command.txt
rocprofv3 --kernel-trace --stats -S -T -d $OUT/trace -o run -- "$@" > $OUT/run_under_rocprof.txt 2> $OUT/trace_stderr.txt
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace -T --pmc $C -d $OUT/pmc_$N -o pmc -- "$@" > $OUT/pmc_$N.txt 2> $OUT/pmc_$N.stderr.txt || echo "pmc pass $C failed" >> $OUT/errors.txt
done
find $OUT -name "*.db" | head -20
