"""one bench workload at a given N, for traces:  python tools/bench_one.py <c2|c3> N [steps]"""
import sys, time
sys.path.insert(0, '.')
import bench
import lsqfitgp_amd as lgp
cfg = dict(bench.CONFIGS[sys.argv[1]]); cfg['n'] = int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = lgp.Context.default()
w = bench.Workload(cfg, ctx)
for _ in range(2):
    w.step()
ctx.sync()
t0 = time.perf_counter()
for _ in range(steps):
    w.step()
ctx.sync()
print(f"{cfg['kernel']} N={cfg['n']}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step")
