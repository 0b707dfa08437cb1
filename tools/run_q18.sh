mkdir -p gpurun_out/q18
for sh in "4 8 8" "4 8 12"; do set -- $sh
  echo "== C=$1 T=$2 waves=$3: word-part breakdown (trace build)"; ISING_LIB=$PWD/ising_gpu_amd/libising_hip_qtrace.so ISING_QUAD_C=$1 ISING_QUAD_T=$2 ISING_QUAD_WAVES=$3 python tools/quad_run.py 2048 2048 512 2 2>&1 | grep -v amdgpu
  (cd /tmp && export TMPDIR=/tmp; ISING_QUAD_C=$1 ISING_QUAD_T=$2 ISING_QUAD_WAVES=$3 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q18/tr -- python $GRAFT_REPO_ROOT/tools/quad_run.py 2048 2048 64 3 > /dev/null 2>&1)
  python tools/quad_timeline.py gpurun_out/q18/tr 14; rm -rf gpurun_out/q18/tr
done
