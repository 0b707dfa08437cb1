#!/bin/bash
# First contact with a multi-GPU node: tools/first_contact.sh [N]   (N = ranks; default: the GPUs of the node, at least 2)
# One process per rank (ranks share devices when N exceeds the GPUs): who owns which device, RCCL and the peer (IPC) transport each attached,
# two deep launches and the exchanges around them checked against the same lattice as a lone slab.  < 60 s; ONE JSON line on stdout; exit 0 = a transport is green.
cd "$(dirname "$0")/.." || exit 2
export HSA_ENABLE_IPC_MODE_LEGACY=0
N=${1:-$(python - <<'PY'
import torch
print(max(2, torch.cuda.device_count()))
PY
)}
PORT=$(python - <<'PY'
import socket
s = socket.socket(); s.bind(("127.0.0.1", 0)); print(s.getsockname()[1])
PY
)
exec timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" tools/first_contact.py
