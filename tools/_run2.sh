mkdir -p gpurun_out/r04b
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r04b/pytest.txt
cat gpurun_out/r04b/pytest.txt
