"""
Gaussian-copula reparametrisation of hyperparameter priors: the subset of
``lsqfitgp.copula`` (``src/lsqfitgp/copula/_copulas.py``) that the callers of the
hot path use (``bayestree/_bart.py:163-171``: ``beta``, ``invgamma``; plus the
other one-parameter families).  A distribution object maps a standard Normal
variable z to the target variable, ``x = F^-1(Phi(z))``, so that the fit
(`empbayes_fit`) works in a space where the prior is N(0, 1).

Host code (scipy): the transforms act on a handful of scalars per iteration of the
fit, not on the O(N^2) / O(N^3) path.
"""

import numpy
from scipy import special, stats


class Distr:
    """ a univariate distribution reparametrised by a standard Normal """

    def invfcn(self, z):  # pragma: no cover
        """ z ~ N(0, 1)  ->  x ~ this distribution """
        raise NotImplementedError

    def fcn(self, x):  # pragma: no cover
        """ inverse of `invfcn` """
        raise NotImplementedError

    def dinvfcn(self, z):
        """ d invfcn / dz = phi(z) / pdf(x) """
        x = self.invfcn(z)
        return stats.norm.pdf(z) / self._pdf(x)

    def __repr__(self):
        args = ', '.join(f'{v!r}' for v in self._args)
        return f'{type(self).__name__}({args})'


def _upper(z):
    """ use survival functions for z > 0, so that the right tail keeps its relative accuracy
    (``_copulas.py:78-88``) """
    return numpy.asarray(z) > 0


class beta(Distr):
    """ Beta(alpha, beta) on (0, 1)  (``_copulas.py:41-50``) """

    def __init__(self, alpha, beta):
        assert alpha > 0 and beta > 0
        self._args = (alpha, beta)
        self._d = stats.beta(alpha, beta)

    def invfcn(self, z):
        z = numpy.asarray(z, float)
        return numpy.where(_upper(z), self._d.isf(special.ndtr(-z)), self._d.ppf(special.ndtr(z)))[()]

    def fcn(self, x):
        x = numpy.asarray(x, float)
        c = self._d.cdf(x)
        return numpy.where(c > 0.5, -special.ndtri(self._d.sf(x)), special.ndtri(c))[()]

    def _pdf(self, x):
        return self._d.pdf(x)


class gamma(Distr):
    """ Gamma(alpha, rate beta)  (``_copulas.py:66-92``) """

    def __init__(self, alpha, beta):
        assert alpha > 0 and beta > 0
        self._args = (alpha, beta)
        self._d = stats.gamma(alpha, scale=1 / beta)

    def invfcn(self, z):
        z = numpy.asarray(z, float)
        return numpy.where(_upper(z), self._d.isf(special.ndtr(-z)), self._d.ppf(special.ndtr(z)))[()]

    def fcn(self, x):
        x = numpy.asarray(x, float)
        c = self._d.cdf(x)
        return numpy.where(c > 0.5, -special.ndtri(self._d.sf(x)), special.ndtri(c))[()]

    def _pdf(self, x):
        return self._d.pdf(x)


class invgamma(Distr):
    """ Inverse-gamma(alpha, scale beta)  (``_copulas.py:122-147``) """

    def __init__(self, alpha, beta):
        assert alpha > 0 and beta > 0
        self._args = (alpha, beta)
        self._d = stats.invgamma(alpha, scale=beta)

    def invfcn(self, z):
        z = numpy.asarray(z, float)
        return numpy.where(_upper(z), self._d.isf(special.ndtr(-z)), self._d.ppf(special.ndtr(z)))[()]

    def fcn(self, x):
        x = numpy.asarray(x, float)
        c = self._d.cdf(x)
        return numpy.where(c > 0.5, -special.ndtri(self._d.sf(x)), special.ndtri(c))[()]

    def _pdf(self, x):
        return self._d.pdf(x)


class halfcauchy(Distr):
    """ Half-Cauchy(gamma)  (``_copulas.py:158-178``) """

    def __init__(self, gamma):
        assert gamma > 0
        self._args = (gamma,)
        self._g = gamma

    def invfcn(self, z):
        z = numpy.asarray(z, float)
        lo = numpy.tan(numpy.pi * special.ndtr(z) / 2)
        with numpy.errstate(divide='ignore'):
            hi = 1 / numpy.tan(numpy.pi * special.ndtr(-z) / 2)
        return (self._g * numpy.where(z < 0, lo, hi))[()]

    def fcn(self, x):
        return special.ndtri(2 / numpy.pi * numpy.arctan(numpy.asarray(x, float) / self._g))

    def _pdf(self, x):
        return stats.halfcauchy.pdf(x, scale=self._g)


class halfnorm(Distr):
    """ Half-Normal(sigma)  (``_copulas.py:180-200``) """

    def __init__(self, sigma):
        assert sigma > 0
        self._args = (sigma,)
        self._s = sigma

    def invfcn(self, z):
        z = numpy.asarray(z, float)
        lo = special.ndtri((1 + special.ndtr(z)) / 2)
        hi = -special.ndtri(special.ndtr(-z) / 2)
        return (self._s * numpy.where(z < 0, lo, hi))[()]

    def fcn(self, x):
        return special.ndtri(2 * special.ndtr(numpy.asarray(x, float) / self._s) - 1)

    def _pdf(self, x):
        return stats.halfnorm.pdf(x, scale=self._s)


class uniform(Distr):
    """ Uniform(a, b)  (``_copulas.py:202-210``) """

    def __init__(self, a, b):
        assert b > a
        self._args = (a, b)

    def invfcn(self, z):
        a, b = self._args
        return a + (b - a) * special.ndtr(numpy.asarray(z, float))

    def fcn(self, x):
        a, b = self._args
        return special.ndtri((numpy.asarray(x, float) - a) / (b - a))

    def _pdf(self, x):
        a, b = self._args
        return numpy.full(numpy.shape(x), 1 / (b - a))[()]


class lognorm(Distr):
    """ Log-Normal(mu, sigma)  (``_copulas.py:212-220``) """

    def __init__(self, mu, sigma):
        assert sigma > 0
        self._args = (mu, sigma)

    def invfcn(self, z):
        mu, sigma = self._args
        return numpy.exp(mu + sigma * numpy.asarray(z, float))

    def fcn(self, x):
        mu, sigma = self._args
        return (numpy.log(numpy.asarray(x, float)) - mu) / sigma

    def dinvfcn(self, z):
        return self._args[1] * self.invfcn(z)


def makedict(variables, prefix=''):
    """ ``copula.makedict`` (``_makedict.py``): here the hyperprior dictionary of `empbayes_fit` takes distribution
    objects as they are, so this only prefixes the keys and checks the values """
    out = {}
    for k, v in variables.items():
        if not isinstance(v, (Distr, tuple, list)) and not hasattr(v, 'mean'):
            raise TypeError(f'hyperprior[{k!r}] must be a copula distribution, (mean, sdev) or a UVal, found {type(v).__name__}')
        out[prefix + k] = v
    return out
