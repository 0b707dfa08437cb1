cd /tmp && export TMPDIR=/tmp
export ISING_QUAD_C=4 ISING_QUAD_T=8 ISING_QUAD_WAVES=12 ISING_QUAD_BATCH=512 ISING_QUAD_NBUF=2
for lib in "" qdt1 qd4; do for ch in 8 2; do
  if [ -n "$lib" ]; then export ISING_LIB=$GRAFT_REPO_ROOT/ising_gpu_amd/libising_hip_$lib.so; else unset ISING_LIB; fi
  ISING_QUAD_CHUNK=$ch rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q11/tr -- python $GRAFT_REPO_ROOT/tools/quad_run.py 2048 2048 512 2 > /dev/null 2>&1
  echo "== lib ${lib:-product} chunk $ch (solo: 512 sweeps of 2048^2 per draw launch)"; python $GRAFT_REPO_ROOT/tools/quad_timeline.py $GRAFT_REPO_ROOT/gpurun_out/q11/tr 1 | grep -E "quad_draw|quad_word"
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/q11/tr
done; done
