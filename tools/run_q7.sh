mkdir -p gpurun_out/q7
(timeout 600 python -m pytest tests/test_gpu_quad.py -x -q 2>&1 | tail -15) > gpurun_out/q7/test.log
for sh in "4 8 8" "4 8 16" "4 8 12" "2 8 12"; do set -- $sh; echo "== solo passes: C=$1 T=$2 waves=$3"; ISING_LIB=$PWD/ising_gpu_amd/libising_hip_qtrace.so ISING_QUAD_BATCH=512 ISING_QUAD_NBUF=2 ISING_QUAD_C=$1 ISING_QUAD_T=$2 ISING_QUAD_WAVES=$3 python tools/quad_run.py 2048 2048 512 2; done > gpurun_out/q7/trace.txt 2>&1
python tools/quad_probe.py --shapes 4,8,8,64:4,8,16,64:2,8,8,64:4,8,12,64:8,8,16,64:2,8,12,64:4,12,16,72:4,16,16,64 2048 2048 > gpurun_out/q7/probe_2048.txt 2>&1
python tools/quad_probe.py --shapes 4,8,8,32:4,8,12,32:4,8,16,32:8,8,16,32 4096 4096 > gpurun_out/q7/probe_4096.txt 2>&1
cat gpurun_out/q7/test.log gpurun_out/q7/trace.txt gpurun_out/q7/probe_2048.txt gpurun_out/q7/probe_4096.txt 
