mkdir -p gpurun_out/r04f
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -8 > gpurun_out/r04f/pytest.txt
cat gpurun_out/r04f/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04f/bench_20_5.json 2> gpurun_out/r04f/bench_20_5.err
timeout 300 python bench.py > gpurun_out/r04f/bench_default.json 2> gpurun_out/r04f/bench_default.err
python tools/policy_probe.py 2>&1 | tee gpurun_out/r04f/policy_probe.txt | tail -30
