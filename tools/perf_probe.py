"""Quick single-GPU throughput probe: flips/ns of the hot loop for a few lattice sizes / strip heights.

Usage: python tools/perf_probe.py [X Y sweeps strip_rows[,strip_rows...]] ...
"""
import sys
import time

sys.path.insert(0, ".")
import ising_gpu_amd as ig


def run(X, Y, sweeps, strip, kernel=ig.KERNEL_AUTO, layout=0):
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, strip_rows=strip, kernel=kernel, layout=layout) as s:
        s.init()
        s.sweep(2)
        s.synchronize()
        best = 1e30
        for _ in range(3):
            ms = s.sweep_timed(sweeps)
            best = min(best, ms)
        flips = X * Y * sweeps / (best * 1e6)
        print(f"X={X} Y={Y} strip={s.strip_rows:3d} kernel={kernel} layout={s.layout} sweeps={sweeps}: {best/sweeps:8.3f} ms/sweep "
              f"{flips:8.1f} flips/ns  ({1.5*flips:7.1f} GB/s algorithmic)", flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    if not args:
        for strip in (1, 4, 8, 16, 32, 64):
            run(65536, 65536, 8, strip)
        for strip in (1, 2, 4, 8, 16):
            run(16384, 16384, 32, strip)
        run(131072, 16384, 8, 0)
        run(8192, 8192, 64, 0)
        run(65536, 65536, 4, 16, ig.KERNEL_GENERIC)
    else:
        X, Y, sweeps = int(args[0]), int(args[1]), int(args[2])
        kernels = [int(v) for v in args[4].split(",")] if len(args) > 4 else [ig.KERNEL_AUTO]
        layouts = [int(v) for v in args[5].split(",")] if len(args) > 5 else [0]
        for layout in layouts:
            for kernel in kernels:
                for strip in [int(v) for v in args[3].split(",")]:
                    run(X, Y, sweeps, strip, kernel, layout)
