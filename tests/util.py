""" comparison helpers restated from the reference's ``tests/util.py:144-232`` """

import numpy as np
from scipy import linalg


def assert_close_matrices(actual, desired, *, rtol=0, atol=0, tozero=False):
    """ ||actual - desired||_2 <= atol + rtol * ||desired||_2 (``tests/util.py:144-200``) """
    actual = np.asarray(actual)
    desired = np.asarray(desired)
    assert actual.shape == desired.shape, (actual.shape, desired.shape)
    if actual.size == 0:
        return
    actual = np.atleast_1d(actual)
    desired = np.atleast_1d(desired)
    diff = actual if tozero else actual - desired
    dnorm = linalg.norm(desired, 2)
    adnorm = linalg.norm(diff, 2)
    ratio = adnorm / dnorm if dnorm else np.nan
    msg = f"""\
matrices actual and {'zero' if tozero else 'desired'} are not close in 2-norm
norm(desired) = {dnorm:.2g}
norm(diff) = {adnorm:.2g}  (atol = {atol:.2g})
ratio = {ratio:.2g}  (rtol = {rtol:.2g})"""
    assert adnorm <= atol + rtol * dnorm, msg


def assert_allclose(actual, desired, *, rtol=0, atol=0, equal_nan=False, **kw):
    """ ``tests/util.py:230-232``: zero default tolerances """
    np.testing.assert_allclose(actual, desired, rtol=rtol, atol=atol, equal_nan=equal_nan, **kw)


def assert_close_decomps(actual, desired, *, rtol=0, atol=0):
    """ ``tests/util.py:220-228`` """
    assert actual.n == desired.n
    def compare(a, b):
        I = np.eye(a.n)
        Z = a.correlate(I)
        zzbz = a.correlate(b.ginv_quad(Z))
        assert_close_matrices(zzbz, Z, rtol=rtol, atol=atol)
    compare(actual, desired)
    compare(desired, actual)


def ulp_distance(a, b):
    """ max |a - b| in units of the spacing of b """
    a = np.asarray(a)
    b = np.asarray(b)
    sp = np.spacing(np.maximum(np.abs(a), np.abs(b)))
    return np.max(np.abs(a - b) / sp) if a.size else 0.
