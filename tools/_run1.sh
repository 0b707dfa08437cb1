set -x
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04a/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04a/bench_20_5.json 2> gpurun_out/r04a/bench_20_5.err
timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04a/bench_n2.json 2> gpurun_out/r04a/bench_n2.err
timeout 300 python bench.py --gpus 2 --workload strong --steps 20 --warmup 5 > gpurun_out/r04a/bench_n2_strong.json 2> gpurun_out/r04a/bench_n2_strong.err
cat gpurun_out/r04a/pytest.txt; cat gpurun_out/r04a/*.json; tail -5 gpurun_out/r04a/*.err
