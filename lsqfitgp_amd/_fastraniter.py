"""
Random samples from a multivariate Gaussian given as mean and covariance:
``lsqfitgp.raniter`` / ``lsqfitgp.sample`` (``src/lsqfitgp/_fastraniter.py:36-121``)
on the device.  The O(m^3) step the reference's guide calls the typical bottleneck of
sampling (``docs/userguide/optim.rst:47-53``) is ``Chol(cov, epsrel=eps)`` (``:96``),
here the blocked MFMA factorisation; every sample is ``mean + L z`` (``:108-109``) with
the factor resident in HBM.  The iid Normal vectors come from the host generator, like in
the reference, so that a seed gives the same stream.
"""

import itertools

import numpy

from ._linalg import Chol


def _flatten(mean, cov):
    """ (flat mean, square covariance, unpack function) for an array/scalar or a dictionary of arrays """
    if hasattr(mean, 'keys'):
        keys = list(mean.keys())
        parts = [numpy.asarray(mean[k], float) for k in keys]
        sizes = [p.size for p in parts]
        stops = numpy.concatenate([[0], numpy.cumsum(sizes)]).astype(int)
        flat = numpy.concatenate([p.reshape(-1) for p in parts]) if parts else numpy.empty(0)
        square = numpy.empty((len(flat), len(flat)))
        for i, ki in enumerate(keys):
            for j, kj in enumerate(keys):
                square[stops[i]:stops[i + 1], stops[j]:stops[j + 1]] = numpy.asarray(cov[ki, kj], float).reshape(sizes[i], sizes[j])

        def unpack(v):
            return {k: v[stops[i]:stops[i + 1]].reshape(parts[i].shape) for i, k in enumerate(keys)}
        return flat, square, unpack
    marr = numpy.asarray(mean, float)
    flat = marr.reshape(-1)
    square = numpy.asarray(cov, float).reshape(len(flat), len(flat))

    def unpack(v):
        return v.reshape(marr.shape) if marr.shape else v.item()
    return flat, square, unpack


def raniter(mean, cov, n=None, eps=None, rng=None, *, ctx=None):
    """
    Take random samples from a multivariate Gaussian (same interface as the reference's `raniter`).

    mean : scalar, array, or dictionary of scalars/arrays
    cov : scalar, array, or dictionary keyed by pairs of keys of ``mean``
    n : int, optional -- maximum number of iterations (default unlimited)
    eps : float, optional -- relative regularisation of the covariance (default: size times machine epsilon,
        the ``epsrel='auto'`` rule of the decomposition)
    rng : seed or ``numpy.random.Generator``

    Yields samples in the format of ``mean`` (dictionaries come back as plain ``dict``).
    """
    flat, square, unpack = _flatten(mean, cov)
    try:
        dec = Chol(square, epsrel='auto' if eps is None else eps, ctx=ctx)
    except numpy.linalg.LinAlgError:
        raise numpy.linalg.LinAlgError('covariance matrix not positive definite with eps={}'.format(eps))
    rng = numpy.random.default_rng(rng)
    for _ in (itertools.count() if n is None else range(n)):
        iid = rng.standard_normal(dec.m)
        yield unpack(flat + dec.correlate(iid))


def sample(*args, **kw):
    """ Shortcut for ``next(raniter(..., n=1))`` """
    return next(raniter(*args, n=1, **kw))
