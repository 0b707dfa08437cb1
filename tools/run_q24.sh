mkdir -p gpurun_out/q24
R=$GRAFT_REPO_ROOT
bash tools/profile_cmd.sh quad2048 python $R/tools/quad_run.py 2048 2048 4096 3 > /dev/null 2>&1
python tools/summarize_cmd_prof.py gpurun_out/prof_quad2048 gpurun_out/q24/rocprof_r05_quad_2048.txt 4194304 12352
bash tools/profile_cmd.sh quad4096 python $R/tools/quad_run.py 4096 4096 1024 3 > /dev/null 2>&1
python tools/summarize_cmd_prof.py gpurun_out/prof_quad4096 gpurun_out/q24/rocprof_r05_quad_4096.txt 16777216 3136
rm -rf gpurun_out/prof_quad2048 gpurun_out/prof_quad4096
