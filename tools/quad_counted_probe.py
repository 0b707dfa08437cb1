import sys, time
sys.path.insert(0, "/root/repo")
import ising_gpu_amd as ig
X = Y = 2048
with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32) as s:
    s.init().sweep(256); s.synchronize()
    n = 16384
    def t(f):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); f(); s.synchronize(); best = min(best, time.perf_counter() - t0)
        return X * Y * n / best * 1e-9, best / n * 1e6
    print("sweep", t(lambda: s.sweep(n)))
    for every in (8, 16, 64, 1024, 16384):
        print("counted every", every, t(lambda: s.sweep_counted(n, every)), "with energy", t(lambda: s.sweep_counted(n, every, True)))
