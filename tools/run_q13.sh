for n in 256 512 1024 2048 4096 8192; do python tools/quad_run.py 2048 2048 $n 3; done
for n in 256 1024 4096; do python tools/quad_run.py 8192 2048 $n 3; done
