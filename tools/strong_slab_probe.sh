#!/bin/bash
# What ONE rank of a strong-scaling run of the 65536^2 lattice delivers ("at 8192 rows per slab the ghost-row redundancy is ~1.5 % per level and
# nobody has looked"): its slab alone (no ring: the upper bound for that shape) and as a ring of one over the library's RCCL ring and its peer (IPC) ring -- 64 ghost
# rows, fused launches, the exchange overlapped: everything a rank of an N-rank ring executes except a real link.  flips/ns, default 16 + 128 sweeps.
for Y in 65536 32768 16384 8192; do
  L=$(python bench.py --y $Y --no-cpu-baseline --no-alu-probe --no-counts-leg 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.readline()); print(b['value'], b['config']['strip_rows'])")
  R=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --force-ring --y $Y --no-cpu-baseline --no-alu-probe --no-counts-leg 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(b['value'], b['config']['strip_rows'], b['config']['exchange'], b['exchange_stats']['go_after_end_ms']['mean'], b['exchange_stats']['gap_ms']['mean'])")
  I=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --force-ring --transport ipc --y $Y --no-cpu-baseline --no-alu-probe --no-counts-leg 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(b['value'], b['config']['exchange'], b['exchange_stats']['go_after_end_ms']['mean'], b['exchange_stats']['gap_ms']['mean'])")
  echo "65536 x $Y (N = $((65536 / Y))): lone slab [flips/ns, H] $L   ring of one, RCCL [flips/ns, H, transport, exchange end vs launch end ms, gap ms] $R   ring of one, IPC $I"
done
