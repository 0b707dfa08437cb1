mkdir -p gpurun_out/r04z
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r04z/pytest.txt
cat gpurun_out/r04z/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04z/bench_20_5.json 2> gpurun_out/r04z/bench_20_5.err
timeout 300 python bench.py > gpurun_out/r04z/bench_default.json 2> gpurun_out/r04z/bench_default.err
for W in config3 config4 strong; do for N in 1 2 4 8; do
  timeout 600 python bench.py --gpus $N --workload $W --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04z/bench_scaling_${W}_n${N}.json 2> gpurun_out/r04z/bench_scaling_${W}_n${N}.err || echo "FAILED $W $N"
done; done
bash tools/run_baseline_configs.sh > gpurun_out/r04z/configs.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04z/bench*.json")):
    try:
        b=json.load(open(f))
        print(f.split("/")[-1], b["value"], b["n_gpus"], b["scaling"], b["config"]["parity_checked"], b["roofline"]["frac"], b.get("with_counts_every_16",{}).get("value"), b.get("exchange_stats",{}).get("exchange_ms",{}).get("mean"))
    except Exception as e:
        print(f, "ERR", e)
PY
grep -E "Kernel execution|^T = 2.25" gpurun_out/r04z/configs.txt
