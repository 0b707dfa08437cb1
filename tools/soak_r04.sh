#!/bin/bash
# Round-4 soak of the fused launches whose units draw before they wait for their parents (the regime where parents ARE late: levels of 512 .. 2048 tickets
# on five and six workgroups per CU): the ballot layout against the dense kernel (an independent implementation), counts and bond sums at every checkpoint,
# final states word for word; then the randomised parity hunts at ten times their usual width.  Usage (GPU box): tools/soak_r04.sh > gpurun_out/soak_r04.txt
set -x
python tools/soak_ballot.py 8192 4096 1000000 10 11
python tools/soak_ballot.py 8192 8192 1000000 10 12
python tools/soak_ballot.py 16384 8192 400000 8 13
python tools/soak_ballot.py 16384 16384 200000 8 14
python tools/soak_ballot.py 32768 32768 40000 4 15
python tools/soak_ballot.py 65536 65536 10000 4 16
for SEED in 51 52 53; do ISING_TEST_RANDOM_SCALE=10 ISING_TEST_RANDOM_SEED=$SEED python -m pytest tests/test_gpu_random.py -q -p no:cacheprovider 2>&1 | tail -2; done
