python tools/perf_probe.py 8192 8192 512 1,2,4 0 3,2 2>&1 | grep "X="
python tools/perf_probe.py 16384 16384 256 2,4,8 0 3 2>&1 | grep "X="
