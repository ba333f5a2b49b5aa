"""
CPU-side checks of the product boundary: the C-ABI library exists, loads, and
exports every symbol ``include/lsqfitgp_hip.h`` declares (no compute calls);
the ctypes table mirrors the header; the package fails loudly without a GPU;
nothing in the product imports the oracle.
"""

import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, 'include', 'lsqfitgp_hip.h')).read()
    return set(re.findall(r'\b(lgp_[a-z0-9_]+)\s*\(', hdr))


def test_library_built_and_loads():
    from lsqfitgp_amd import _lib
    assert os.path.exists(_lib.library_path()), 'run __graft_entry__.build() first'
    lib = _lib.load()
    assert lib.lgp_version() == 1


def test_exports_every_declared_symbol():
    from lsqfitgp_amd import _lib
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.library_path()], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if ' T ' in line}
    declared = _declared()
    assert declared, 'header parse failed'
    assert not (declared - exported), f'declared but not exported: {sorted(declared - exported)}'
    assert not {e for e in exported - declared if e.startswith('lgp_')}, 'exported lgp_ symbols missing from the header'


def test_ctypes_table_matches_header():
    from lsqfitgp_amd import _lib
    assert set(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    for name in _lib.SIGNATURES:
        assert getattr(lib, name).argtypes is not None


def test_no_gpu_fails_loudly():
    """ without a HIP device every entry point of the package raises LibraryError: no CPU fallback """
    import lsqfitgp_amd as lgp
    from lsqfitgp_amd import _lib
    count = ctypes.c_int(0)
    if _lib.load().lgp_device_count(ctypes.byref(count)) == 0 and count.value > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(lgp.LibraryError):
        lgp.Chol(np.eye(3))
    with pytest.raises(lgp.LibraryError):
        lgp.ExpQuad()(np.arange(3.)[:, None], np.arange(3.)[None, :])
    with pytest.raises(lgp.LibraryError):
        lgp.GP(lgp.ExpQuad()).addx(np.arange(3.), 'x').prior('x', raw=True)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'lsqfitgp_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, re.M), f'{f} imports the oracle'
                assert 'oracle/' not in text or f.endswith('.md'), f'{f} references oracle/'


def test_host_bracket_folding():
    """ host logic of the BART wrapper: reset depths -> recursion slices (``_kernels/_bart.py:415-447``) """
    from lsqfitgp_amd._kernels import _bart_brackets, _bart_probs
    assert _bart_brackets(11, [2, 4, 6, 8]) == [(0, 10, 5)]
    assert _bart_brackets(5, 2) == [(0, 4, 2)]
    assert _bart_brackets(3, None) == [(0, 2, 1)]
    assert _bart_brackets(4, [1]) == [(0, 1, 1), (1, 3, 1)]
    assert _bart_brackets(7, [1, 3]) == [(0, 1, 1), (1, 3, 1), (3, 6, 1)]
    assert _bart_brackets(7, [3]) == [(0, 3, 1), (3, 6, 1)]            # three levels deep: never folded
    assert _bart_brackets(7, [2, 4], altinput=False) == [(0, 2, 1), (2, 4, 1), (4, 6, 1)]
    assert _bart_brackets(7, [2, 4], debug=True) == [(0, 2, 1), (2, 4, 1), (4, 6, 1)]
    assert _bart_brackets(6, [2, 4]) == [(0, 4, 2), (4, 5, 1)]
    assert _bart_brackets(4, [1, 2]) == [(0, 3, 3)]
    assert _bart_brackets(1, None) == [(0, 0, 1)]
    with pytest.raises(AssertionError):
        _bart_brackets(5, [3, 2])
    pnt = 0.95 / (1 + np.arange(11)) ** 2.
    probs = _bart_probs(pnt, 0, 10, 5)
    expected = np.array([1, pnt[9], pnt[10], 1, pnt[7], pnt[8], 1, pnt[5], pnt[6], 1, pnt[3], pnt[4], pnt[0], pnt[1], pnt[2]])
    np.testing.assert_array_equal(probs, expected)      # deepest slice first
    import oracle  # the oracle's restatement of the same lines must agree
    with np.errstate(all='ignore'):
        n = np.array([5, 7]); ix = np.array([1, 6]); iy = np.array([3, 2])
        a = oracle.BART.correlation(n, ix, iy, maxd=10, reset=[2, 4, 6, 8], altinput=True)
        g = 1.
        for k in range(5):
            g = oracle.BART.correlation(n, ix, iy, pnt=probs[3 * k:3 * k + 3], gamma=g, altinput=True)
        np.testing.assert_allclose(a, g, rtol=1e-15)


def test_host_leaf_columns():
    from lsqfitgp_amd._kernels import _leaf_columns
    x = np.zeros(4, dtype=[('a', float), ('b', float, 2), ('c', [('d', int), ('e', float)])])
    x['a'] = np.arange(4); x['b'] = np.arange(8).reshape(4, 2); x['c']['d'] = 7; x['c']['e'] = 9
    cols = _leaf_columns(x)
    assert len(cols) == 5
    np.testing.assert_array_equal(cols[1], x['b'][:, 0])
    np.testing.assert_array_equal(cols[2], x['b'][:, 1])
    np.testing.assert_array_equal(cols[3], 7)


def test_splits_and_indices_host(rng):
    """ BART.splits_from_coord / indices_from_coord are host-side (sort/unique): compare with the oracle """
    import lsqfitgp_amd as lgp
    import oracle
    X = rng.integers(0, 6, (40, 3)).astype(float)
    l1, s1 = lgp.BART.splits_from_coord(X)
    l2, s2 = oracle.BART.splits_from_coord(X)
    np.testing.assert_array_equal(l1, l2)
    np.testing.assert_array_equal(s1, s2)
    Y = rng.uniform(-1, 7, (25, 3))
    np.testing.assert_array_equal(lgp.BART.indices_from_coord(Y, (l1, s1)), oracle.BART.indices_from_coord(Y, (l2, s2)))
    x = np.repeat(np.arange(10 * 2.).reshape(-1, 2), 2, axis=0).view('d,d').squeeze(axis=-1)
    length, _ = lgp.BART.splits_from_coord(x)
    assert np.all(length == 9)          # ref tests/kernels/test_bart.py:268-271


def test_derivative_specifications():
    """ the reference's ``tests/test_deriv.py`` on `GP._field_orders`, which reads a derivative specification as ``Deriv`` does
    (``_Deriv.py:24-97``): integers, field names, sequences with multiplicities; the malformed ones raise the same errors """
    from lsqfitgp_amd._GP import GP
    f = GP._field_orders
    assert f(None) == 0 and f(0) == 0 and f(3) == 3
    assert f('a') == {'a': 1}
    assert f(['a', 'b', 'a']) == {'a': 2, 'b': 1}
    assert f([1, 'ciao', 2, 'pippo']) == {'ciao': 1, 'pippo': 2} and len(f([1, 'ciao', 2, 'pippo'])) == 2
    assert f((2, 'f0')) == {'f0': 2}
    assert f([0, 'a']) == 0                                  # zero times along a: no derivative
    with pytest.raises(TypeError):
        f((None,))
    with pytest.raises(ValueError):
        f((1, 2))
    with pytest.raises(TypeError):
        f(object)
    with pytest.raises(ValueError):
        f((1,))
    with pytest.raises(ValueError):
        f(('ciao', 1))
    with pytest.raises(ValueError):
        f(-1)
    with pytest.raises(ValueError):
        f((-1, 'a'))
    with pytest.raises(TypeError):
        f(True)
