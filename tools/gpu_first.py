"""ad-hoc first GPU check: gemm variants, potrf, Chol vs oracle, gram, GP C1, timings"""
import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np
from scipy import linalg
import lsqfitgp_amd as lgp
from lsqfitgp_amd import _lib
from lsqfitgp_amd._GP import _gemm
import oracle

ctx = lgp.Context.default()
print(ctx.info())
rng = np.random.default_rng(0)

def relerr(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)

# ---- gemm
for dt in (np.float64, np.float32):
    for (m, n, k) in [(5, 7, 3), (128, 128, 16), (100, 1, 1000), (300, 257, 129), (1000, 100, 4100), (513, 515, 33)]:
        for ta in (0, 1):
            for tb in (0, 1):
                A = rng.standard_normal((k, m) if ta else (m, k)).astype(dt)
                B = rng.standard_normal((n, k) if tb else (k, n)).astype(dt)
                C0 = rng.standard_normal((m, n)).astype(dt)
                Cd = lgp.DeviceArray.from_host(C0)
                _gemm(ta, tb, lgp.DeviceArray.from_host(A), lgp.DeviceArray.from_host(B), 0.5, -1.5, Cd)
                ref = 0.5 * (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64) - 1.5 * C0
                e = relerr(Cd.to_host(), ref)
                tol = 1e-13 if dt == np.float64 else 2e-5
                if e > tol:
                    print('GEMM FAIL', dt.__name__, m, n, k, ta, tb, e)
print('gemm checked')

# ---- Chol vs oracle, well conditioned
for dt, tol in ((np.float64, 1e-12), (np.float32, 1e-3)):
    for n in (1, 2, 10, 100, 128, 129, 300, 513, 1000, 1500):
        Q = rng.standard_normal((n, n))
        K = (Q @ Q.T / n + np.eye(n) * 0.5) * rng.choice([0.01, 1, 37.0])
        K = K.astype(dt)
        t0 = time.time()
        d = lgp.Chol(K)
        o = oracle.Chol(K)
        L = d.factor().to_host()
        e = relerr(L, o._L)
        A = rng.standard_normal((n, 7)).astype(dt)
        r = rng.standard_normal(n).astype(dt)
        e2 = relerr(d.ginv_linear(A), o.ginv_linear(A))
        e3 = relerr(d.pinv_bilinear(A, r), o.pinv_bilinear(A, r))
        e4 = relerr(d.ginv_quad(A), o.ginv_quad(A))
        e5 = abs(d.minus_log_normal_density(r, value=True)[0] - o.minus_log_normal_density(r, value=True)[0])
        e6 = relerr(d.correlate(A), o.correlate(A)); e7 = relerr(d.back_correlate(A), o.back_correlate(A))
        e8 = relerr(d.ginv_diagquad(A), o.ginv_diagquad(A))
        ok = max(e, e2, e3, e4, e6, e7, e8) < tol * 100 and np.triu(L, 1).max(initial=0) == 0
        print(dt.__name__, n, 'L %.1e ginv %.1e bil %.1e quad %.1e mlnd %.1e corr %.1e %.1e dq %.1e eps %.3e/%.3e' % (e, e2, e3, e4, e5, e6, e7, e8, d.eps, o.eps), 'OK' if ok else 'FAIL')

# non-PD must raise
try:
    lgp.Chol(-np.eye(5)); print('FAIL no raise')
except np.linalg.LinAlgError as ex:
    print('non-PD raises OK:', ex)

# ---- gram
x = rng.uniform(-5, 5, 777); y = rng.uniform(-5, 5, 333)
for kd, ko in ((lgp.ExpQuad(), oracle.ExpQuad()), (lgp.ExpQuad(scale=2.5, loc=0.3) * 3 + 1, oracle.ExpQuad(scale=2.5, loc=0.3) * 3 + 1), (lgp.Maternp(p=2, scale=1.3), oracle.Maternp(p=2, scale=1.3)), (lgp.Maternp(p=0), oracle.Maternp(p=0)), (lgp.Maternp(p=5), oracle.Maternp(p=5))):
    G = kd.gram(x, y).to_host(); R = ko(x[:, None], y[None, :])
    ulp = np.max(np.abs(G - R) / np.spacing(np.abs(R)))
    print('gram 1d max ulp', ulp, 'diag', kd(x[:5], x[:5]))
xs = np.zeros(500, dtype=[('a', float), ('b', float), ('c', float)])
for f in 'abc': xs[f] = rng.uniform(0, 16, 500)
G = lgp.Maternp(p=2).gram(xs, xs).to_host(); R = oracle.Maternp(p=2)(xs[:, None], xs[None, :])
print('gram 3d matern max ulp', np.max(np.abs(G - R) / np.spacing(np.abs(R))), 'sym', np.abs(G - G.T).max())

# ---- GP C1
x = rng.uniform(-5, 5, 512); xt = rng.uniform(-10, 10, 100); yv = np.tanh(x)
gd = lgp.GP(lgp.ExpQuad(), checkpos=False, checksym=False).addx(x, 'data').addx(xt, 'test')
go = oracle.GP(oracle.ExpQuad(), checkpos=False, checksym=False).addx(x, 'data').addx(xt, 'test')
m1, c1 = gd.predfromdata({'data': yv}, 'test', raw=True)
m2, c2 = go.predfromdata({'data': yv}, 'test', raw=True)
print('C1 mean rel', relerr(m1, m2), 'cov 2norm rel', np.linalg.norm(c1 - c2, 2) / np.linalg.norm(go.prior('test', raw=True), 2),
      'mll', gd.marginal_likelihood({'data': yv}), go.marginal_likelihood({'data': yv}))

# ---- timings
for n in (4096, 8192, 16384):
    x = rng.uniform(0, n / 16, n)
    t0 = time.time(); K = lgp.ExpQuad().gram(x, x); ctx.sync(); t1 = time.time()
    d = lgp.Chol(K); ctx.sync(); t2 = time.time()
    d2 = lgp.Chol(K); ctx.sync(); t3 = time.time()
    print('n', n, 'gram %.1f ms' % ((t1 - t0) * 1e3), 'chol first %.1f ms second %.1f ms -> %.1f TFLOP/s' % ((t2 - t1) * 1e3, (t3 - t2) * 1e3, n**3 / 3 / (t3 - t2) / 1e12))
    if n <= 8192:
        o = oracle.Chol(K.to_host())
        print('   L rel err vs oracle', relerr(d.factor().to_host(), o._L), 'eps', d.eps, o.eps)
