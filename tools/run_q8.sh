mkdir -p gpurun_out/q8
cd /tmp && export TMPDIR=/tmp
export ISING_QUAD_C=4 ISING_QUAD_T=8 ISING_QUAD_WAVES=12
ISING_QUAD_BATCH=512 ISING_QUAD_NBUF=2 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q8/tr1 -- python $GRAFT_REPO_ROOT/tools/quad_run.py 2048 2048 512 2 > $GRAFT_REPO_ROOT/gpurun_out/q8/run1.txt 2>&1
ISING_QUAD_BATCH=64 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q8/tr2 -- python $GRAFT_REPO_ROOT/tools/quad_run.py 2048 2048 1024 2 > $GRAFT_REPO_ROOT/gpurun_out/q8/run2.txt 2>&1
cd $GRAFT_REPO_ROOT
echo "== solo (one batch of 512 sweeps per call)"; python tools/quad_timeline.py gpurun_out/q8/tr1 12
echo "== pipelined (batches of 64)"; python tools/quad_timeline.py gpurun_out/q8/tr2 40
rm -rf gpurun_out/q8/tr1 gpurun_out/q8/tr2
