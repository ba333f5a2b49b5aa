"""how far apart are bart(X, y) and bart(X, a + b y)?  python tools/bart_equivariance.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from tests.test_gpu_fit import bart_problem
rng = np.random.default_rng(3)
X, y = bart_problem(rng)
offset, scale = 0.4703189, 0.5294714
for kw in (dict(), dict(minkw=dict(method='l-bfgs-b', options=dict(maxls=4, maxiter=100, ftol=0, gtol=1e-9))), dict(minkw=dict(method='bfgs'))):
    b1 = lgp.bayestree.bart(X, y, fitkw=dict(verbosity=0, **kw))
    b2 = lgp.bayestree.bart(X, offset + y * scale, fitkw=dict(verbosity=0, **kw))
    m1, c1 = b1.pred(); m2, c2 = b2.pred()
    print(kw, 'iterations', b1.fit.minresult.nit, b2.fit.minresult.nit, 'mean', np.max(np.abs(m2 - (offset + m1 * scale)) / np.abs(m2)),
          'cov', np.max(np.abs(c2 - c1 * scale ** 2)) / np.max(np.abs(c2)), 'z', np.max(np.abs(b1.fit.zmean - b2.fit.zmean)),
          'fun', b1.fit.minresult.fun, b2.fit.minresult.fun, b1.fit.minresult.message)
