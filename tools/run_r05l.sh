mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05l_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05l_pytest.txt
timeout 600 python -m pytest tests/test_gpu_policy.py -q -s > gpurun_out/r05l_policy.txt 2>&1
SHORT_STEPS="--steps 32 --warmup 32" ISING_SPLIT=1 bash tools/profile.sh r05_split16k --x 16384 --y 16384 --strip-rows 16 > gpurun_out/r05l_profile.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_prof.py gpurun_out/prof_r05_split16k gpurun_out/rocprof_r05_config2_split >> gpurun_out/r05l_profile.log 2>&1
rm -rf gpurun_out/prof_r05*/trace gpurun_out/prof_r05*/pmc_*/
timeout 900 bash tools/strong_slab_probe.sh > gpurun_out/r05l_strong_slab.txt 2>&1
