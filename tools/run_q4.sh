mkdir -p gpurun_out/q4
python tools/quad_probe.py --shapes 4,8,8,16:4,8,8,32:4,8,8,64:4,8,8,128:4,8,8,64,1:4,8,8,64,2:4,8,8,64,8:4,8,16,64:8,8,16,64:4,8,12,64:2,8,8,64:4,4,8,64:4,6,8,66:4,12,16,72 2048 2048 > gpurun_out/q4/probe_2048.txt 2>&1
python tools/quad_probe.py --shapes 4,8,8,16:4,8,8,32:4,8,12,32:8,8,16,32:4,8,12,16,2:4,8,12,64 4096 4096 > gpurun_out/q4/probe_4096.txt 2>&1
for sh in "4 8 8" "4 8 16"; do set -- $sh; echo "== C=$1 T=$2 waves=$3"; ISING_LIB=$PWD/ising_gpu_amd/libising_hip_qtrace.so ISING_QUAD_BATCH=64 ISING_QUAD_C=$1 ISING_QUAD_T=$2 ISING_QUAD_WAVES=$3 python tools/quad_run.py 2048 2048 1024 2; done > gpurun_out/q4/trace.txt 2>&1
cd /tmp && export TMPDIR=/tmp; export ISING_QUAD_C=4 ISING_QUAD_T=8 ISING_QUAD_WAVES=8 ISING_QUAD_BATCH=64; rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q4/tr -- python $GRAFT_REPO_ROOT/tools/quad_run.py 2048 2048 1024 2 > $GRAFT_REPO_ROOT/gpurun_out/q4/run.txt 2>&1; cd $GRAFT_REPO_ROOT; python tools/quad_timeline.py gpurun_out/q4/tr 50 > gpurun_out/q4/timeline.txt; rm -rf gpurun_out/q4/tr
cat gpurun_out/q4/probe_2048.txt gpurun_out/q4/probe_4096.txt gpurun_out/q4/trace.txt gpurun_out/q4/timeline.txt
