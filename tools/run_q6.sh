mkdir -p gpurun_out/q6
for sh in "4 8 8" "4 8 16" "4 8 4" "8 8 16"; do set -- $sh; echo "== solo passes: C=$1 T=$2 waves=$3"; ISING_LIB=$PWD/ising_gpu_amd/libising_hip_qtrace.so ISING_QUAD_BATCH=512 ISING_QUAD_NBUF=2 ISING_QUAD_C=$1 ISING_QUAD_T=$2 ISING_QUAD_WAVES=$3 python tools/quad_run.py 2048 2048 512 2; done > gpurun_out/q6/trace.txt 2>&1
cat gpurun_out/q6/trace.txt
