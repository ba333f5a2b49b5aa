"""
Generates tests/golden/*.npz.  These vectors are produced by the CPU oracle
(oracle/, a restatement of the reference's algorithm on LAPACK), NOT by the
reference itself: lsqfitgp cannot be imported in the build container (no jax /
gvar), see oracle/__init__.py.  They freeze the oracle's outputs on small seeded
inputs so that the GPU box can be checked without recomputing them and so that
a later change to the oracle is noticed.

    python tests/golden/make_golden.py
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402


def c1():
    """ config C1 of SURVEY.md 8(d): ExpQuad 1-D, N=512 / M=100, seed 1, no noise (jitter path);
    plus an in-domain test set, where the posterior mean is well conditioned """
    rng = np.random.default_rng(1)
    x = rng.uniform(-5, 5, 512)
    xs = rng.uniform(-10, 10, 100)
    xin = rng.uniform(-5, 5, 100)
    y = np.tanh(x)
    gp = (oracle.GP(oracle.ExpQuad(), checkpos=False, checksym=False)
          .addx(x, 'data').addx(xs, 'test').addx(xin, 'testin'))
    m, c = gp.predfromdata({'data': y}, 'test', raw=True)
    mi, ci = gp.predfromdata({'data': y}, 'testin', raw=True)
    mll = gp.marginal_likelihood({'data': y})
    np.savez_compressed(os.path.join(HERE, 'c1_expquad_n512.npz'), x=x, xs=xs, xin=xin, y=y, mean=m, cov=c,
                        mean_in=mi, cov_in=ci, mll=mll, eps=gp._solver(('data',)).eps)


def c2small():
    """ config C2 at reduced size: ExpQuad 1-D, N=1024, 16 points per lengthscale, noise 0.01 I, seed 2 """
    rng = np.random.default_rng(2)
    n = 1024
    x = rng.uniform(0, n / 16, n)
    xs = rng.uniform(0, n / 16, 100)
    y = np.sin(x / 4) + 0.1 * rng.standard_normal(n)
    gp = oracle.GP(oracle.ExpQuad(), checkpos=False, checksym=False).addx(x, 'data').addx(xs, 'test')
    gc = {('data', 'data'): 0.01 * np.eye(n)}
    m, c = gp.predfromdata({'data': y}, 'test', gc, raw=True)
    mll = gp.marginal_likelihood({'data': y}, gc)
    np.savez_compressed(os.path.join(HERE, 'c2_expquad_n1024.npz'), x=x, xs=xs, y=y, mean=m, cov=c, mll=mll)


def c3small():
    """ config C3 at reduced size: Matern-5/2 3-D, N=1024, noise 0.01 I, seed 3 """
    rng = np.random.default_rng(3)
    n = 1024
    dt = [('f0', float), ('f1', float), ('f2', float)]
    x = np.empty(n, dt); xs = np.empty(100, dt)
    for f in 'f0', 'f1', 'f2':
        x[f] = rng.uniform(0, 16, n)
    for f in 'f0', 'f1', 'f2':
        xs[f] = rng.uniform(0, 16, 100)
    y = sum(np.sin(x[f]) for f in ('f0', 'f1', 'f2')) + 0.1 * rng.standard_normal(n)
    gp = oracle.GP(oracle.Maternp(p=2, scale=1.), checkpos=False, checksym=False).addx(x, 'data').addx(xs, 'test')
    gc = {('data', 'data'): 0.01 * np.eye(n)}
    m, c = gp.predfromdata({'data': y}, 'test', gc, raw=True)
    mll = gp.marginal_likelihood({'data': y}, gc)
    np.savez_compressed(os.path.join(HERE, 'c3_matern52_n1024.npz'), x0=x['f0'], x1=x['f1'], x2=x['f2'],
                        xs0=xs['f0'], xs1=xs['f1'], xs2=xs['f2'], y=y, mean=m, cov=c, mll=mll)


def c5small():
    """ config C5 structure at reduced size: BART Gram (maxd=10, reset=[2,4,6,8]) on synthetic integer-valued
    covariates, p=10, N=256, fp32 and fp64 """
    rng = np.random.default_rng(5)
    n, p = 256, 10
    X = np.concatenate([rng.integers(0, 2, (n, 3)), rng.integers(0, 40, (n, 7))], axis=1).astype(float)
    splits = oracle.BART.splits_from_coord(X)
    ix = oracle.BART.indices_from_coord(X, splits).astype(np.int32)
    out = {}
    for name, flt in (('f64', np.float64), ('f32', np.float32)):
        kw = dict(alpha=flt(0.95), beta=flt(2), gamma=flt(1), weights=np.ones(p, flt))
        k = oracle.BART(splits=(splits[0], None), indices=True, maxd=10, reset=[2, 4, 6, 8], **kw)
        ixs = np.array([tuple(r) for r in ix], dtype=[(f'f{i}', np.int32) for i in range(p)])
        out['K_' + name] = k(ixs[:, None], ixs[None, :])
        k2 = oracle.BART(splits=(splits[0], None), indices=True, maxd=2, **kw)
        out['K2_' + name] = k2(ixs[:, None], ixs[None, :])
    np.savez_compressed(os.path.join(HERE, 'c5_bart_n256.npz'), X=X, nsplits=splits[0], ix=ix, **out)


from workloads import load_abalone, c5_problem  # noqa: E402


C5_GRAM_ROWS = [0, 1, 2, 3, 500, 1777, 3000, 4095]


def c5full():
    """ config C5 on its real workload: first 4096 rows of the Abalone table for training, the remaining 81 rows
    held out (the file has 4177 rows, so the "100-row" test block of SURVEY.md 8(d) is the 81 rows there are);
    outputs of the oracle: posterior mean at 'trainmean' (M = N) with the diagonal and one 64 x 64 corner of its
    covariance, mean and covariance at 'testmean', log marginal likelihood (epsrel = 0, ``bayestree/_bart.py:222``),
    8 rows of the unscaled fp32 BART Gram matrix, and the norms the GPU test derives its fp32 tolerance from """
    X, y = load_abalone()
    n = 4096
    gp, given, kern, pts = c5_problem(oracle, X, y, n)
    mt, ct = gp.predfromdata(given, 'trainmean', raw=True)
    ms, cs = gp.predfromdata(given, 'testmean', raw=True)
    mll = gp.marginal_likelihood(given, epsrel=0)
    base = oracle.BART(splits=(oracle.BART.splits_from_coord(X[:n])[0], None), indices=True, alpha=np.float32(0.95),
                       beta=np.float32(2), gamma=np.float32(1), maxd=10, reset=[2, 4, 6, 8], weights=np.ones(10, np.float32))
    rows = base(pts[C5_GRAM_ROWS][:, None], pts[None, :n])
    # sensitivity of the posterior mean to a relative perturbation of the Gram entries (see tests/test_gpu_gp.py):
    # K = Kxx (train x train, assembled, float64), alpha = K^-1 r
    Kxx = np.asarray(gp.prior('train', raw=True), float)
    r = np.asarray(given['train'], float)
    alpha = np.linalg.solve(Kxx, r)
    Ktt = np.asarray(gp.prior('trainmean', raw=True), float)
    evmin = np.linalg.eigvalsh(Kxx)[0]
    np.savez_compressed(os.path.join(HERE, 'c5_abalone_n4096.npz'),
                        mean_train=mt, covdiag_train=np.diag(ct).copy(), covcorner_train=ct[:64, :64].copy(),
                        mean_test=ms, cov_test=cs, mll=mll, gram_rows=rows, gram_row_index=np.array(C5_GRAM_ROWS),
                        norm_alpha=np.linalg.norm(alpha), fro_Ktt=np.linalg.norm(Ktt), norm2_Ktt=np.linalg.norm(Ktt, 2),
                        evmin_Kxx=evmin, norm_mean_train=np.linalg.norm(mt))


# one parameter set per kernel of the reference's public list (BART apart: c5small / c5full); the points of each kernel's domain
KERNEL_CASES = {
    'ExpQuad': {}, 'Maternp': dict(p=2), 'Matern': dict(nu=1.3), 'GammaExp': dict(gamma=1.3), 'Cauchy': dict(alpha=1.5, beta=2.5),
    'Wendland': dict(k=1, alpha=2), 'CausalExpQuad': dict(alpha=1.5), 'Log': {}, 'Constant': {}, 'White': {}, 'Linear': {},
    'Expon': {}, 'HoleEffect': {}, 'Periodic': dict(outerscale=0.7), 'Cos': {}, 'Sinc': {}, 'Celerite': dict(gamma=0.8, B=0.3),
    'Harmonic': dict(Q=0.8), 'Circular': dict(tau=5, c=0.3), 'StationaryFracBrownian': dict(H=0.3), 'Wiener': {}, 'WienerIntegral': {},
    'OrnsteinUhlenbeck': {}, 'BrownianBridge': {}, 'FracBrownian': dict(H=0.3, K=0.7), 'Decaying': dict(alpha=1.5), 'Taylor': {},
    'NNKernel': dict(sigma0=0.7), 'Gibbs': {}, 'Rescaling': {}, 'Bessel': dict(nu=1.3), 'Color': dict(n=4), 'Pink': dict(dw=2.),
    'Zeta': dict(nu=1.3), 'MA': dict(w=[1., 0.5, -0.25]), 'AR': dict(phi=[0.6, -0.2], maxlag=64),
}
KERNEL_DOMAIN = {'Wiener': (0, 5), 'WienerIntegral': (0, 5), 'OrnsteinUhlenbeck': (0, 5), 'BrownianBridge': (0, 1), 'Decaying': (0, 5), 'Taylor': (-2, 2)}


def kernel_points(name):
    rng = np.random.default_rng(sum(map(ord, name)))
    if name in ('MA', 'AR'):
        return rng.integers(0, 60, 24), rng.integers(0, 60, 16)
    lo, hi = KERNEL_DOMAIN.get(name, (-4, 4))
    return rng.uniform(lo, hi, 24), rng.uniform(lo, hi, 16)


def kernels():
    """ a 24 x 16 block of every kernel (scale 1.3 where the kernel takes one), frozen from the oracle: a change to a core of the
    oracle or of the device is noticed whichever side it happens on """
    out = {}
    for name, kw in KERNEL_CASES.items():
        x, y = kernel_points(name)
        kwargs = dict(kw) if name in ('MA', 'AR') else dict(kw, scale=1.3)
        out[name] = getattr(oracle, name)(**kwargs)(x[:, None], y[None, :])
    np.savez_compressed(os.path.join(HERE, 'kernel_blocks.npz'), **out)


if __name__ == '__main__':
    import sys as _sys
    which = _sys.argv[1:] or ['c1', 'c2small', 'c3small', 'c5small', 'c5full', 'kernels']
    for name in which:
        globals()[name]()
    print('written', sorted(f for f in os.listdir(HERE) if f.endswith('.npz')))
