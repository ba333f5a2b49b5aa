"""
CPU checks of the configuration-C5 fixtures: the data file is the reference's own ``examples/bart-data.txt`` (by hash),
and the oracle reproduces the rows of the BART Gram matrix frozen in ``tests/golden/c5_abalone_n4096.npz`` bit for bit
(the full N = 4096 fixture takes the oracle ~30 s to regenerate: ``python tests/golden/make_golden.py c5full``).
"""

import gzip
import hashlib
import os

import numpy as np

import oracle
from workloads import load_abalone, c5_problem, HERE

REFERENCE_SHA256 = 'de37cdcdcaaa50c309d514f248f7c2302a5f1f88c168905eba23fe2fbc78449f'     # sha256 of examples/bart-data.txt in the reference checkout


def test_data_file_is_the_reference_file():
    with gzip.open(os.path.join(HERE, 'bart-data.txt.gz'), 'rb') as f:
        raw = f.read()
    assert hashlib.sha256(raw).hexdigest() == REFERENCE_SHA256
    ref = '/root/reference/examples/bart-data.txt'
    if os.path.exists(ref):          # build container only
        with open(ref, 'rb') as f:
            assert f.read() == raw
    X, y = load_abalone()
    assert X.shape == (4177, 10) and X.dtype == np.float32 and y.shape == (4177,)
    assert np.all(X[:, :3].sum(axis=1) == 1)               # one-hot Sex
    assert (y.min(), y.max()) == (1, 29)


def test_oracle_reproduces_gram_rows():
    g = np.load(os.path.join(HERE, 'c5_abalone_n4096.npz'))
    X, y = load_abalone()
    n = 4096
    splits = oracle.BART.splits_from_coord(X[:n])
    assert list(splits[0]) == [1, 1, 1, 133, 110, 50, 2404, 1502, 873, 917]
    ix = oracle.BART.indices_from_coord(X, splits).astype(np.int32)
    pts = np.array([tuple(r) for r in ix], dtype=[(f'f{i}', np.int32) for i in range(10)])
    f = np.float32
    base = oracle.BART(splits=(splits[0], None), indices=True, alpha=f(0.95), beta=f(2), gamma=f(1), maxd=10,
                       reset=[2, 4, 6, 8], weights=np.ones(10, f))
    rows = base(pts[g['gram_row_index']][:, None], pts[None, :n])
    assert rows.dtype == np.float32
    np.testing.assert_array_equal(rows, g['gram_rows'])
    for k, i in enumerate(g['gram_row_index']):
        assert rows[k, i] == 1                              # correlation of a point with itself


def test_problem_structure_small():
    """ the C5 GP on the first 300 rows through the oracle: shapes, dtype promotion (fp32 kernel blocks meet float64
    zero blocks, ``_GP/_elements.py:616``), finite likelihood """
    X, y = load_abalone()
    gp, given, kern, pts = c5_problem(oracle, X[:330], y[:330], 300)
    m, c = gp.predfromdata(given, 'testmean', raw=True)
    assert m.shape == (30,) and c.shape == (30, 30) and m.dtype == np.float64
    assert np.all(np.linalg.eigvalsh(c) > -1e-6)
    assert np.isfinite(gp.marginal_likelihood(given, epsrel=0))
