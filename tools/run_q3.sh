mkdir -p gpurun_out/q3
python tools/quad_probe.py --shapes 4,8,8:4,8,16:4,8,8,8:4,8,8,16:8,8,16,16:4,8,12:4,16,16,16:2,8,8,16 2048 2048 > gpurun_out/q3/probe_2048.txt 2>&1
python tools/quad_probe.py --shapes 4,8,8:4,8,16:4,8,8,8:4,8,12,16:8,8,16,16 4096 4096 > gpurun_out/q3/probe_4096.txt 2>&1
for sh in "4 8 8" "4 8 16" "4 16 16"; do set -- $sh; echo "== C=$1 T=$2 waves=$3"; ISING_LIB=$PWD/ising_gpu_amd/libising_hip_qtrace.so ISING_QUAD_C=$1 ISING_QUAD_T=$2 ISING_QUAD_WAVES=$3 python tools/quad_run.py 2048 2048 1024 2; done > gpurun_out/q3/trace.txt 2>&1
cd /tmp && export TMPDIR=/tmp; export ISING_QUAD_C=4 ISING_QUAD_T=8 ISING_QUAD_WAVES=8; rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q3/tr -- python $GRAFT_REPO_ROOT/tools/quad_run.py 2048 2048 1024 2 > $GRAFT_REPO_ROOT/gpurun_out/q3/run.txt 2>&1; cd $GRAFT_REPO_ROOT; python tools/quad_timeline.py gpurun_out/q3/tr 50 > gpurun_out/q3/timeline.txt; rm -rf gpurun_out/q3/tr
cat gpurun_out/q3/probe_2048.txt gpurun_out/q3/probe_4096.txt gpurun_out/q3/trace.txt gpurun_out/q3/timeline.txt
