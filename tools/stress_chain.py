"""python tools/stress_chain.py [reps] -- the chain-kernel driver against the launch-per-operation driver (LGP_CHAIN=0) on the same random
positive definite matrices, sizes up to the chain driver's limit, several repetitions each: the same factor to rounding, info 0"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
os.environ['LGP_CHAIN'] = '0'
import lsqfitgp_amd as lgp
launch = lgp.Context(0)
del os.environ['LGP_CHAIN']
chain = lgp.Context(0)
rng = np.random.default_rng(5)
bad = 0
for n in (1152, 2048, 3200, 4096, 6144, 8192, 10240, 12288):
    x = np.sort(rng.uniform(0, n / 16, n))
    K = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2) + 0.01 * np.eye(n)
    ref = None
    for r in range(reps):
        for name, ctx in (('launch', launch), ('chain', chain)):
            if name == 'launch' and ref is not None:
                continue
            d = lgp.Chol(K, ctx=ctx)
            L = np.tril(d.factor().to_host())
            if name == 'launch':
                ref = L
                continue
            err = np.max(np.abs(L - ref)) / np.max(np.abs(ref))
            ok = np.isfinite(err) and err < 1e-9
            bad += not ok
            print(f'n={n} rep {r}: max |L_chain - L_launch| / max |L| = {err:.2e} {"ok" if ok else "WRONG"}', flush=True)
print('wrong results:', bad)
sys.exit(1 if bad else 0)
