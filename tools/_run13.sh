mkdir -p gpurun_out/r04z
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -rf 2>&1 | tail -60 > gpurun_out/r04z/pytest2.txt
cat gpurun_out/r04z/pytest2.txt
