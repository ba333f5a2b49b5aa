import sys, time
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from workloads import load_abalone
X, y = load_abalone(); X = X.astype(float); y = y.astype(float)
n = 4096
from lsqfitgp_amd.bayestree._bart import _Design, _Outcome, _Regression
ctx = lgp.Context(0)
model = _Regression(_Design(X[:n]), _Outcome(y[:n], None, n), {}, False, ctx)
hp = {'alpha': 0.7059, 'beta': 1.449, 'k': 2.1277, 'sigma2': 4.5297}
mode = sys.argv[1]
ref = None
for trial in range(12):
    dec, r, dK, dr, names = model.likelihood_pieces(hp, mlkw=dict(epsrel=0))
    if mode == 'sums':
        ctx.sync()
        sums = [float(np.abs(m.to_host()).sum()) for m in dK]
        g1 = np.asarray(dec.normal_density_derivs(r, dK, dr)[1])
        g2 = np.asarray(dec.normal_density_derivs(r, dK, dr)[1])
        sums2 = [float(np.abs(m.to_host()).sum()) for m in dK]
        print(trial, sums == sums2, sums, g1, g2, flush=True)
    elif mode == 'twice':
        g1 = np.asarray(dec.normal_density_derivs(r, dK, dr)[1])
        g2 = np.asarray(dec.normal_density_derivs(r, dK, dr)[1])
        g3 = np.asarray(dec.normal_density_derivs(r, dK, dr)[1])
        print(trial, np.array_equal(g1, g2), np.array_equal(g2, g3), g1, g2, g3, flush=True)
    elif mode == 'keep':
        keep = getattr(model, '_keep', [])
        keep.append((dec, dK))         # nothing is ever released: no block reuse
        model._keep = keep
        g1 = np.asarray(dec.normal_density_derivs(r, dK, dr)[1])
        print(trial, g1, flush=True)
