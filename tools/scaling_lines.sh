#!/bin/bash
# Every point of a scaling run of bench.py as processes on ONE GPU (the peer transport): config3 / config4 / strong x N = 2, 4, 8 at the driver's 5 + 20 sweeps.
# Usage (GPU box): bash tools/scaling_lines.sh <tag>   ->  gpurun_out/bench_<tag>_scaling_<workload>_n<N>.json
TAG=${1:-r04z}
cd ${GRAFT_REPO_ROOT:-/root/repo}
for w in config3 config4 strong; do
  for n in 2 4 8; do
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --steps 20 --warmup 5 --workload $w 2>/dev/null | tail -1 > gpurun_out/bench_${TAG}_scaling_${w}_n$n.json
    python - <<P
import json
try:
    d = json.loads(open("gpurun_out/bench_${TAG}_scaling_${w}_n$n.json").read())
    print("$w N=$n:", d["value"], "with counts", d["with_counts_every_16"]["value"], "parity_checked", d["config"].get("parity_checked"), "final counts equal", d["with_counts_every_16"]["final_counts_equal_first_leg"])
except Exception as e:
    print("$w N=$n: FAILED", e)
P
  done
done
