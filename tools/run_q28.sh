mkdir -p gpurun_out/q28
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_cmd.sh quad2048b python $R/tools/quad_run.py 2048 2048 4096 3 > /dev/null 2>&1
python tools/summarize_cmd_prof.py gpurun_out/prof_quad2048b gpurun_out/q28/rocprof_r05_quad_2048.txt 4194304 12352 | tail -22
rm -rf gpurun_out/prof_quad2048b
