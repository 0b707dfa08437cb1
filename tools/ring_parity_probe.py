#!/usr/bin/env python3
"""Ring of one at the bench size against the oracle's golden counts (tests/golden/bench_65536_tc.json: 25 sweeps), through
every schedule / transport / buffer owner.  usage: ring_parity_probe.py"""
import json, os, sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch
import torch.distributed as dist
import ising_gpu_amd as ig
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29588", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
gold = {p["sweeps"]: (p["up"], p["down"]) for p in json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bench_65536_tc.json")))["points"]}
X = Y = 65536
def check(name, make):
    res = []
    for rep in range(3):
        ring, closer = make()
        ring.init()
        got = {}
        done = 0
        for upto in (5, 21, 25):
            ring.sweep(upto - done); done = upto
            got[upto] = ring.count()
        res.append(all(got[k] == gold[k] for k in got))
        closer()
    print(f"{name}: {res}", flush=True)
def native_lib():
    s = ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, ring_halo=True)
    r = ig.NativeRing(s)
    return r, lambda: (r.close(), s.close())
def native_torch():
    b = ig.HipSlabBackend.create(X, Y, device=0, seed=1234, temp=ig.CRIT_TEMP_F32, nslabs=1, slab=0, ring_halo=True)
    r = ig.NativeRing(b.slab)
    return r, lambda: (r.close(), b.slab.close())
def slabset():
    s = ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, ring_halo=True)
    r = ig.SlabSet([s])
    return r, s.close
for name, env, mk in (("NativeRing (RCCL), library buffer", {}, native_lib), ("NativeRing (RCCL), torch-owned buffer", {}, native_torch),
                      ("SlabSet copy transport, default", {}, slabset), ("SlabSet copy, two streams", {"ISING_RING_INLINE": "0", "ISING_RING_STORE": "0"}, slabset),
                      ("SlabSet copy, one stream, edge launch + copies", {"ISING_RING_INLINE": "1", "ISING_RING_STORE": "0"}, slabset)):
    for k in ("ISING_RING_INLINE", "ISING_RING_STORE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        check(name, mk)
    except Exception as e:
        print(f"{name}: {type(e).__name__}: {e}")
dist.destroy_process_group()
