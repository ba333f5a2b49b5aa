"""
``bayestree.bart``: nonparametric regression with the GP limit of BART -- the caller behind
configuration C5 of the hot path (reference: ``src/lsqfitgp/bayestree/_bart.py:35-458``).

Same statistical model, hyperprior, GP keys and public surface (``gp``, ``data``, ``pred``,
``mean / sigma / alpha / beta / meansdev / fit``) as the reference; the code is organised
around three objects of this module instead of closures:

`_Design`      the covariate side: any supported container -> one structured array, the
               splitting grid of the kernel and the integer bin indices of a point set;
`_Outcome`     the outcome side: values, weights, the location / half-range that fix the
               priors of the intercept and of the latent amplitude;
`_Regression`  the model: which GP a hyperparameter dictionary stands for, and -- since there
               is no automatic differentiation here -- the train covariance
               ``lambda^2 B(alpha, beta) + sigma^2 diag(1/w) + m^2 11'`` with its derivatives,
               assembled on the device in one pass per L-BFGS iteration.

Outputs that are gvars in the reference (``sigma``, ``alpha`` ...) are `UVal` (mean and
standard deviation); ``pred(format='gvar')`` is not available.
"""

import ctypes

import numpy
from numpy.lib import recfunctions

from .. import _lib
from .. import copula
from .._device import Context, DeviceArray
from .._fit import empbayes_fit
from .._GP import GP, diag
from .._kernels import BART
from .._linalg import Chol
from .._uncertain import UVal, sqrt as usqrt

# depth truncation of the kernel that the reference fixes for this model (``_bart.py:188-191``)
_KERNEL_DEFAULTS = dict(maxd=10, reset=[2, 4, 6, 8])
# minimiser settings of the reference (``_bart.py:207-214``)
_FIT_DEFAULTS = dict(verbosity=3, raises=False, forward=True,
                     minkw=dict(method='l-bfgs-b', options=dict(maxls=4, maxiter=100)))


class _Design:
    """ covariates as the kernel wants them """

    def __init__(self, x):
        self.x = self.structured(x)
        self.names = list(self.x.dtype.names)
        self.grid = BART.splits_from_coord(self.x)          # (counts per axis, sorted midpoints)
        self.index = self.bins(self.x)

    @staticmethod
    def structured(x):
        """ dataframe-like (``.columns`` + ``.to_numpy()``), (n, p) matrix or structured array -> 1-d structured
        array with one numerical field per covariate (what ``_bart.py:406-434`` accepts) """
        if hasattr(x, 'columns') and hasattr(x, 'to_numpy'):
            table = numpy.asarray(x.to_numpy())
            x = recfunctions.unstructured_to_structured(table, names=[str(c) for c in x.columns])
        else:
            x = numpy.asarray(x)
            if x.dtype.names is None:
                if x.ndim != 2:
                    raise ValueError(f'unstructured covariates must be an (n, p) matrix, found shape {x.shape}')
                x = recfunctions.unstructured_to_structured(x, names=[f'f{j}' for j in range(x.shape[1])])
        if x.ndim != 1:
            raise ValueError(f'covariates must list one point per row, found shape {x.shape}')
        bad = [nm for nm in x.dtype.names if not numpy.issubdtype(x.dtype[nm].base, numpy.number)]
        if bad:
            raise TypeError(f'covariate `{bad[0]}` is not numerical')
        return x

    def bins(self, x):
        """ bin indices of the points ``x`` on this design's grid, as an int32 structured array with the fields of x """
        if x.dtype.names is None or len(x.dtype.names) != len(self.names):
            raise ValueError(f'{len(self.names)} covariates expected')
        table = BART.indices_from_coord(x, self.grid)
        return recfunctions.unstructured_to_structured(numpy.ascontiguousarray(table, numpy.int32), names=list(x.dtype.names))


class _Outcome:
    """ outcomes and weights; centre and half-range of the outcomes """

    def __init__(self, y, weights, n):
        if hasattr(y, 'to_numpy'):
            y = numpy.squeeze(y.to_numpy())
        self.y = numpy.asarray(y, float)
        if self.y.shape != (n,):
            raise ValueError(f'y_train has shape {self.y.shape}, x_train has {n} rows')
        self.weighted = weights is not None
        self.w = numpy.ones(n) if weights is None else numpy.asarray(weights, float)
        if self.w.shape != (n,):
            raise ValueError(f'weights has shape {self.w.shape}, expected ({n},)')
        lo, hi = self.y.min(), self.y.max()
        self.centre = (hi + lo) / 2
        self.halfrange = (hi - lo) / 2
        # scale of the error-variance prior: weighted mean square deviation (``_bart.py:163``)
        self.msd = numpy.mean(self.w * (self.y - self.y.mean()) ** 2)
        self.sdev = self.y.std()


class _Regression:
    """
    The model of ``_bart.py:70-100``:  y_i = mu + lambda f(x_i) + eps_i,  eps_i ~ N(0, sigma^2 / w_i),
    f ~ GP(0, BART(alpha, beta)),  mu ~ N(centre, halfrange^2) (marginalised unless ``free_mean``),
    log sigma^2 ~ N(log msd, 4),  log k ~ N(log 2, 4) with lambda = halfrange / k,  alpha ~ Beta(2, 1),
    beta ~ InvGamma(1, 1).
    """

    def __init__(self, design, outcome, kernelkw, free_mean, ctx):
        self.design, self.outcome = design, outcome
        self.kernelkw = {**_KERNEL_DEFAULTS, **kernelkw}
        self.free_mean = bool(free_mean)
        self.ctx = ctx

    # ---- hyperparameters ---------------------------------------------------------------------------------------------

    def hyperprior(self):
        o = self.outcome
        prior = {
            'alpha': copula.beta(2, 1),
            'beta': copula.invgamma(1, 1),
            'log(k)': (numpy.log(2), 2),
            'log(sigma2)': (numpy.log(o.msd), 2),
        }
        if self.free_mean:
            prior['mean'] = (o.centre, o.halfrange)
        return prior

    def intercept(self, hp):
        return hp['mean'] if self.free_mean else self.outcome.centre

    def amplitude2(self, hp):
        return (self.outcome.halfrange / hp['k']) ** 2

    def kernel(self, hp):
        """ lambda^2 BART(alpha, beta) over bin indices """
        return BART(splits=self.design.grid, indices=True, alpha=hp['alpha'], beta=hp['beta'], **self.kernelkw) * self.amplitude2(hp)

    # ---- the GP a hyperparameter dictionary stands for ------------------------------------------------------------------

    def _observed(self, gp, hp, which, index, w):
        """ keys '<which>mean' (latent), '<which>noise' (errors) and '<which>' (their sum, plus the intercept when it is
        marginalised) -- the key layout of ``_bart.py:187-194,286-302`` """
        gp = gp.addx(index, which + 'mean').addcov(diag(hp['sigma2'] / w), which + 'noise')
        terms = {which + 'mean': 1, which + 'noise': 1}
        if not self.free_mean:
            terms['mean'] = 1
        return gp.addtransf(terms, which)

    def gp(self, hp, **_):
        gp = GP(self.kernel(hp), checkpos=False, checksym=False, solver='chol', ctx=self.ctx)
        if not self.free_mean:
            gp = gp.addcov(self.outcome.halfrange ** 2, 'mean')
        return self._observed(gp, hp, 'train', self.design.index, self.outcome.w)

    def gp_with_test(self, hp, x_test, w_test):
        gp = self.gp(hp)
        if x_test is None:
            return gp
        index = self.design.bins(_Design.structured(x_test))
        if index.dtype != self.design.index.dtype:
            raise ValueError('test covariates must have the fields of the training covariates')
        if w_test is None:
            w_test = numpy.ones(index.shape)
        else:
            w_test = numpy.asarray(w_test, float)
            if w_test.shape != index.shape:
                raise ValueError(f'weights has shape {w_test.shape}, expected {index.shape}')
        return self._observed(gp, hp, 'test', index, w_test)

    def given(self, hp, **_):
        return {'train': self.outcome.y - self.intercept(hp)}

    # ---- likelihood pieces with derivatives (the `gradfactory` protocol of `empbayes_fit`) -------------------------------

    def likelihood_pieces(self, hp, *, mlkw={}, **_):
        """
        ``(decomposition of the train covariance, residuals, [dK/dtheta], dr/dtheta, names)`` for theta = alpha, beta, k,
        sigma2 (, mean): what forward differentiation of `gp` / `given` gives the reference's fit (``_bart.py:213``).

            K        = lambda^2 B + sigma2 diag(1/w) [+ halfrange^2 11']
            dK/dalpha, dK/dbeta   tangents of the BART recursion, computed with B in one kernel pass
            dK/dk    = -2/k lambda^2 B            dK/dsigma2 = diag(1/w)           dK/dmean = 0, dr/dmean = -1
        """
        ctx, lib, F = self.ctx, _lib.load(), ctypes.c_double
        n = len(self.design.index)
        K, tangents = self.kernel(hp).gram_jvp(self.design.index, self.design.index, ctx=ctx)
        dk = DeviceArray.empty((n, n), numpy.float64, ctx)
        _lib.check(lib.lgp_axpby_f64(ctx.handle, n, n, F(-2 / hp['k']), K.ptr, K.ld, F(0), None, 0, dk.ptr, dk.ld), 'lgp_axpby')
        invw = DeviceArray.from_host(1 / self.outcome.w, ctx=ctx)
        ds = DeviceArray.zeros((n, n), numpy.float64, ctx)
        _lib.check(lib.lgp_add_diag_f64(ctx.handle, n, ds.ptr, ds.ld, invw.ptr, F(1)), 'lgp_add_diag')
        _lib.check(lib.lgp_add_diag_f64(ctx.handle, n, K.ptr, K.ld, invw.ptr, F(hp['sigma2'])), 'lgp_add_diag')
        names = ['alpha', 'beta', 'k', 'sigma2']
        dK = list(tangents) + [dk, ds]
        dr = None
        if self.free_mean:
            names.append('mean')
            dK.append(None)
            dr = numpy.zeros((n, len(names)))
            dr[:, -1] = -1
        else:
            _lib.check(lib.lgp_add_scalar_f64(ctx.handle, n, n, K.ptr, K.ld, F(self.outcome.halfrange ** 2)), 'lgp_add_scalar')
        return Chol(K, **mlkw), self.outcome.y - self.intercept(hp), dK, dr, names


class bart:
    """
    Nonparametric Bayesian regression with a GP version of BART.

    Parameters are those of the reference: ``x_train`` (n, p) array, structured array or dataframe-like,
    ``y_train`` (n,), ``weights`` (n,) rescaling the error variance as 1 / weight, ``fitkw`` (overrides for
    `empbayes_fit`), ``kernelkw`` (overrides for `BART`), ``marginalize_mean``.  (sigma, lambda, alpha, beta) are
    set to their marginal MAP in the Gaussian-copula parametrisation of the hyperprior (see `_Regression`).

    Attributes: ``mean``, ``sigma``, ``alpha``, ``beta``, ``meansdev`` (`UVal` or float), ``fit``.
    Methods: ``gp``, ``data``, ``pred``.
    """

    def __init__(self, x_train, y_train, *, weights=None, fitkw={}, kernelkw={}, marginalize_mean=True, ctx=None):
        ctx = ctx or Context.default()
        design = _Design(x_train)
        outcome = _Outcome(y_train, weights, len(design.x))
        model = self._model = _Regression(design, outcome, dict(kernelkw), not marginalize_mean, ctx)

        options = {**_FIT_DEFAULTS, 'mlkw': dict(epsrel=0), 'gradfactory': model.likelihood_pieces}
        options.update(fitkw)
        # the factory arguments the reference exposes as ``fit.gpfactorykw`` (the model object holds the same data)
        factorykw = dict(i_train=design.index, weights=outcome.w, splits=design.grid, mu_mu=outcome.centre)
        factorykw.update(options.pop('gpfactorykw', {}))
        factorykw['mlkw'] = dict(options['mlkw'])
        self.fit = empbayes_fit(model.hyperprior(), model.gp, model.given, gpfactorykw=factorykw, **options)

        p = self.fit.p
        self.alpha, self.beta = p['alpha'], p['beta']
        self.sigma = usqrt(p['sigma2'])
        self.meansdev = outcome.halfrange / p['k']
        self.mean = p['mean'] if model.free_mean else outcome.centre

    # ---- hyperparameter selection ---------------------------------------------------------------------------------------

    def _hyper(self, hp, rng):
        """ 'map' | 'sample' | an explicit dictionary (``_bart.py:232-240``) """
        if isinstance(hp, str):
            if hp == 'map':
                return self.fit.pmean
            if hp == 'sample':
                return self.fit.sample_hyperparameters(rng)
            raise KeyError(hp)
        return hp

    # ---- public surface ---------------------------------------------------------------------------------------------------

    def gp(self, *, hp='map', x_test=None, weights=None, rng=None):
        """
        The centred Gaussian process at the chosen hyperparameters (``_bart.py:242-304``).  Keys: 'Xmean', 'Xnoise' and
        'X' = 'Xmean' + 'Xnoise' (+ the marginalised intercept), X = 'train', and 'test' when ``x_test`` is given
        (``weights``: error-variance weights of the test points).  Add the attribute ``mean`` to predictions.
        """
        return self._model.gp_with_test(self._hyper(hp, rng), x_test, weights)

    def data(self, *, hp='map', rng=None):
        """ ``y_train`` as the dictionary `GP.pred` takes on a GP returned by `gp` (``_bart.py:306-327``) """
        return self._model.given(self._hyper(hp, rng))

    def pred(self, *, hp='map', error=False, format='matrices', x_test=None, weights=None, rng=None):
        """
        Mean and covariance matrix of the Normal posterior of the latent regression function (``error=False``) or of
        the outcome (``error=True``) at ``x_test`` (default: the training covariates); ``_bart.py:329-404``.
        """
        if format == 'gvar':
            raise NotImplementedError('gvar outputs are out of scope of the device path: use format="matrices"')
        if format != 'matrices':
            raise KeyError(format)
        hp = self._hyper(hp, rng)
        model = self._model
        key = ('train' if x_test is None else 'test') + ('' if error else 'mean')
        m, c = model.gp_with_test(hp, x_test, weights).predfromdata(model.given(hp), key, raw=True)
        return m + model.intercept(hp), c

    def __repr__(self):
        o = self._model.outcome
        lines = [
            'BART fit:',
            f'alpha = {self.alpha} (0 -> intercept only, 1 -> any)',
            f'beta = {self.beta} (0 -> any, ∞ -> no interactions)',
            f'mean = {self.mean}',
            f'latent sdev = {self.meansdev} (large -> conservative extrapolation)',
            f'data total sdev = {o.sdev:.3g}',
        ]
        if o.weighted:
            avg = (self.sigma ** 2 * float(numpy.mean(1 / o.w))).sqrt()
            lines += [f'error sdev (avg weighted) = {avg}', f'error sdev (unweighted) = {self.sigma}']
        else:
            lines.append(f'error sdev = {self.sigma}')
        return '\n'.join(lines)
