""" stress: the objective + gradient of bayestree.bart at a fixed point, over and over: every evaluation must give the same numbers """
import sys, time
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from workloads import load_abalone
X, y = load_abalone(); X = X.astype(float); y = y.astype(float)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
from lsqfitgp_amd.bayestree._bart import _Design, _Outcome, _Regression
ctx = lgp.Context(0)
model = _Regression(_Design(X[:n]), _Outcome(y[:n], None, n), {}, False, ctx)
hps = [{'alpha': 0.7059, 'beta': 1.449, 'k': 2.1277, 'sigma2': 4.5297}, {'alpha': 0.70424, 'beta': 1.4584, 'k': 2.3096, 'sigma2': 1.4327}]
ref = {}
bad = 0
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    i = trial % 2
    try:
        dec, r, dK, dr, names = model.likelihood_pieces(hps[i], mlkw=dict(epsrel=0))
        v, g, _ = dec.normal_density_derivs(r, dK, dr, fisher=False)
        g = np.asarray(g)
        if i not in ref:
            ref[i] = (v, g)
        same = v == ref[i][0] and np.array_equal(g, ref[i][1])
        if not same:
            bad += 1
        print(trial, 'ok' if same else 'DIFFERENT', repr(v), g, flush=True)
    except np.linalg.LinAlgError as e:
        bad += 1
        print(trial, 'FAILED info', getattr(e, 'info', None), flush=True)
print('bad', bad)
