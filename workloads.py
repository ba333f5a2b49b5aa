"""
Configuration C5 on its real workload (shared by bench.py, tests/golden/make_golden.py and the GPU parity tests):
the reference's data file and the GP that ``bayestree.bart`` builds from it.
"""

import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests', 'golden')      # the data file is a test fixture


def load_abalone():
    """ ``bart-data.txt.gz`` is a byte-for-byte gzip of the reference's data file ``examples/bart-data.txt`` (the UCI
    Abalone table, 4177 rows: Sex, Length, Diameter, Height, Whole, Shucked, Viscera, Shell, Rings).  Covariates as
    ``examples/bart.py:37-55`` builds them: Sex -> 3 one-hot columns (F, I, M) followed by the 7 numeric columns
    (p = 10); outcome = Rings.  Everything float32 (config C5 of SURVEY.md 8(d)). """
    import gzip
    with gzip.open(os.path.join(HERE, 'bart-data.txt.gz'), 'rt') as f:
        rows = [line.strip().split(',') for line in f if line.strip()]
    sex = np.array([r[0] for r in rows])
    num = np.array([[float(v) for v in r[1:8]] for r in rows])
    y = np.array([float(r[8]) for r in rows])
    X = np.concatenate([(sex[:, None] == np.array(['F', 'I', 'M'])[None, :]).astype(float), num], axis=1)
    return X.astype(np.float32), y.astype(np.float32)


def c5_prepare(mod, X, y, n):
    """ the data preparation of ``bayestree.bart`` (``bayestree/_bart.py:152-173``) at the fixed hyperparameters of
    SURVEY.md 8(d): splits from the n training rows, bin indices of every row, alpha = 0.95, beta = 2, maxd = 10,
    reset = [2, 4, 6, 8], gamma = 1, k = 2, sigma2 = mean((y - ybar)^2), unit weights; everything float32 / int32 """
    f = np.float32
    p = X.shape[1]
    ytr = y[:n]
    ymin, ymax = ytr.min(), ytr.max()
    mu_mu = f((ymax + ymin) / 2)
    k_sigma_mu = f((ymax - ymin) / 2)
    lam2 = f((k_sigma_mu / f(2)) ** 2)
    sigma2 = f(np.mean((ytr - ytr.mean()) ** 2))
    splits = mod.BART.splits_from_coord(X[:n])
    ix = mod.BART.indices_from_coord(X, splits).astype(np.int32)
    pts = np.array([tuple(r) for r in ix], dtype=[(f'f{i}', np.int32) for i in range(p)])
    kern = mod.BART(splits=(splits[0], None), indices=True, alpha=f(0.95), beta=f(2), gamma=f(1), maxd=10,
                    reset=[2, 4, 6, 8], weights=np.ones(p, f)) * lam2
    return dict(n=n, pts=pts, kern=kern, sigma2=sigma2, k_sigma_mu=k_sigma_mu, given={'train': ytr - mu_mu})


def c5_gp(mod, prep, diag=None):
    """ the GP that ``bayestree.bart`` builds (``bayestree/_bart.py:175-194``, marginalised mean) plus the held-out rows as
    'testmean'.  ``diag``: how the noise block is given -- default the dense ``numpy.diag`` of the reference API;
    ``lsqfitgp_amd.diag`` builds the same block on the device """
    n, pts, f = prep['n'], prep['pts'], np.float32
    noise = np.diag(np.full(n, prep['sigma2'])).astype(f) if diag is None else diag(np.full(n, prep['sigma2'], f))
    return (mod.GP(prep['kern'], checkpos=False, checksym=False, solver='chol')
            .addx(pts[:n], 'trainmean')
            .addcov(noise, 'trainnoise')
            .addcov(np.asarray(prep['k_sigma_mu'] ** 2, f), 'mean')
            .addtransf({'trainmean': 1, 'trainnoise': 1, 'mean': 1}, 'train')
            .addx(pts[n:], 'testmean'))


def c5_problem(mod, X, y, n):
    """ ``(gp, given, kernel, points)`` of configuration C5: `c5_prepare` + `c5_gp`; ``mod`` is the oracle or the device
    package (same surface) """
    prep = c5_prepare(mod, X, y, n)
    return c5_gp(mod, prep), prep['given'], prep['kern'], prep['pts']
