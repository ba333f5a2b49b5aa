"""
``tests/golden/kernel_blocks.npz``: a 24 x 16 block of every kernel of the reference's public list (one parameter set each, seeded
points), frozen from the oracle by ``tests/golden/make_golden.py kernels``.  The oracle must still reproduce it (CPU), and the device
must agree with it (GPU): a change to a core is noticed whichever side it happens on.
"""

import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location('make_golden', os.path.join(HERE, 'golden', 'make_golden.py'))
make_golden = importlib.util.module_from_spec(spec)
spec.loader.exec_module(make_golden)
GOLD = np.load(os.path.join(HERE, 'golden', 'kernel_blocks.npz'))
NAMES = sorted(make_golden.KERNEL_CASES)


def build(mod, name):
    kw = make_golden.KERNEL_CASES[name]
    return getattr(mod, name)(**(dict(kw) if name in ('MA', 'AR') else dict(kw, scale=1.3)))


def test_fixture_lists_every_kernel():
    import oracle
    public = {n for n in dir(oracle) if n[0].isupper() and n not in ('GP', 'Chol', 'Decomposition', 'Kernel', 'BART', 'Categorical', 'BagOfWords')
              and callable(getattr(oracle, n))}
    assert public == set(NAMES) == set(GOLD.files)


@pytest.mark.parametrize('name', NAMES)
def test_oracle_reproduces_the_fixture(name):
    import oracle
    x, y = make_golden.kernel_points(name)
    np.testing.assert_allclose(build(oracle, name)(x[:, None], y[None, :]), GOLD[name], rtol=1e-13, atol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_device_agrees_with_the_fixture(name):
    import lsqfitgp_amd as lgp
    x, y = make_golden.kernel_points(name)
    got = build(lgp, name).gram(x, y).to_host()
    # (Color: the reference's two-branch evaluation, which the fixture freezes, is itself good to ~1e-9 at n = 4)
    tol = 2e-9 if name == 'Color' else 2e-13
    np.testing.assert_allclose(got, GOLD[name], rtol=0, atol=tol * max(1., np.abs(GOLD[name]).max()))
