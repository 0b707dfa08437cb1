mkdir -p gpurun_out
# the headline workload at HEAD (regenerates profiles/traffic.json), then BASELINE config 2 in the split and in the fused form
bash tools/profile.sh r05 > gpurun_out/r05j_profile.log 2>&1
bash tools/profile.sh r05_split16k --x 16384 --y 16384 >> gpurun_out/r05j_profile.log 2>&1
ISING_SPLIT=0 bash tools/profile.sh r05_fused16k --x 16384 --y 16384 >> gpurun_out/r05j_profile.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_prof.py gpurun_out/prof_r05 gpurun_out/rocprof_r05 gpurun_out/traffic.json >> gpurun_out/r05j_profile.log 2>&1
python tools/summarize_prof.py gpurun_out/prof_r05_split16k gpurun_out/rocprof_r05_config2_split >> gpurun_out/r05j_profile.log 2>&1
python tools/summarize_prof.py gpurun_out/prof_r05_fused16k gpurun_out/rocprof_r05_config2_fused >> gpurun_out/r05j_profile.log 2>&1
rm -rf gpurun_out/prof_r05*/trace gpurun_out/prof_r05*/pmc_*/  # (the databases are large; the summaries are what is kept)
