import sys, time
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from lsqfitgp_amd import _fit
from workloads import load_abalone
X, y = load_abalone(); X = X.astype(float); y = y.astype(float)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
orig = _fit.empbayes_fit._derivs
def dbg(self, z, fisher):
    hp = self._space.unflat(z)
    t = time.perf_counter()
    try:
        out = orig(self, z, fisher)
        print('z', np.round(z, 4), {k: float('%.5g' % v) for k, v in hp.items()}, 'f=%.17g' % out[0], '|g|=%.3g' % np.linalg.norm(out[1]), '%.1f ms' % ((time.perf_counter() - t) * 1e3), flush=True)
        return out
    except Exception as e:
        print('z', z, hp, 'FAILED', repr(e), flush=True)
        raise
_fit.empbayes_fit._derivs = dbg
t = time.perf_counter()
b = lgp.bayestree.bart(X[:n], y[:n], fitkw=dict(verbosity=1))
print('total', time.perf_counter() - t, b.fit.minresult.nit)
print(b)
