"""
``bayestree.bart``: nonparametric regression with the GP version of BART, the caller of
configuration C5 of the hot path (``src/lsqfitgp/bayestree/_bart.py:35-458``).  Same
model, hyperprior, fixed settings and public surface as the reference; the Gram matrices
(with their alpha / beta tangents), the factorisations, the likelihood with its gradient
and the predictions run on the GPU, the L-BFGS iteration on the host.

Outputs that are gvars in the reference (``sigma``, ``alpha`` ...) are `UVal` (mean and
standard deviation); ``pred(format='gvar')`` is not available.
"""

import ctypes

import numpy
from numpy.lib import recfunctions

from .. import _lib
from .. import copula
from .. import _fastraniter
from .._device import Context, DeviceArray
from .._fit import empbayes_fit
from .._GP import GP, diag
from .._kernels import BART
from .._linalg import Chol
from .._uncertain import UVal, sqrt as usqrt


class bart:
    """
    Nonparametric Bayesian regression with a GP version of BART.

    Parameters are those of the reference: ``x_train`` (n, p) array, structured array or
    dataframe-like (``.columns`` / ``.to_numpy()``), ``y_train`` (n,), ``weights`` (n,)
    rescaling the error variance as 1 / weight, ``fitkw`` (overrides for `empbayes_fit`),
    ``kernelkw`` (overrides for `BART`), ``marginalize_mean``.

    Model (``_bart.py:70-100``): y_i = mu + lambda f(x_i) + eps_i, eps_i ~ N(0, sigma^2 / w_i),
    f ~ GP(0, BART(alpha, beta)), mu ~ N((max y + min y)/2, (max y - min y)^2/4) marginalised,
    log sigma^2 ~ N(log mean(w (y - ybar)^2), 4), log k ~ N(log 2, 4) with lambda = (max y - min y)/(2k),
    alpha ~ Beta(2, 1), beta ~ InvGamma(1, 1); (sigma, lambda, alpha, beta) at their marginal MAP.

    Attributes: ``mean``, ``sigma``, ``alpha``, ``beta``, ``meansdev`` (`UVal` or float), ``fit``.
    Methods: ``gp``, ``data``, ``pred``.
    """

    def __init__(self, x_train, y_train, *, weights=None, fitkw={}, kernelkw={}, marginalize_mean=True, ctx=None):
        self._ctx = ctx or Context.default()
        x_train = self._to_structured(x_train)
        if hasattr(y_train, 'to_numpy'):
            y_train = y_train.to_numpy().squeeze()
        y_train = numpy.asarray(y_train, float)
        assert y_train.shape == x_train.shape

        self._no_weights = weights is None
        if self._no_weights:
            weights = numpy.ones_like(y_train)
        weights = numpy.asarray(weights, float)
        assert weights.shape == y_train.shape

        # prior mean and variance                                                _bart.py:152-156
        ymin, ymax = numpy.min(y_train), numpy.max(y_train)
        mu_mu = (ymax + ymin) / 2
        k_sigma_mu = (ymax - ymin) / 2

        # splitting points and indices                                           :158-160
        splits = BART.splits_from_coord(x_train)
        i_train = self._toindices(x_train, splits)

        # prior on hyperparams                                                   :162-173
        sigma2_priormean = numpy.mean((y_train - y_train.mean()) ** 2 * weights)
        hyperprior = {
            'alpha': copula.beta(2, 1),                                  # base of tree gen prob
            'beta': copula.invgamma(1, 1),                               # exponent of tree gen prob
            'log(k)': (numpy.log(2), 2),                                 # denominator of prior sdev
            'log(sigma2)': (numpy.log(sigma2_priormean), 2),             # i.i.d. error variance, scaled with weights
            'mean': (mu_mu, k_sigma_mu),                                 # mean of the GP
        }
        if marginalize_mean:
            hyperprior.pop('mean')
        kernelkw = dict(kernelkw)

        def makekernel(hp, splits):
            kw = dict(alpha=hp['alpha'], beta=hp['beta'], maxd=10, reset=[2, 4, 6, 8])
            kw.update(kernelkw)
            return BART(splits=splits, indices=True, **kw) * (k_sigma_mu / hp['k']) ** 2

        # GP factory                                                             :175-194
        def makegp(hp, *, i_train, weights, splits, **_):
            gp = (GP(makekernel(hp, splits), checkpos=False, checksym=False, solver='chol', ctx=self._ctx)
                  .addx(i_train, 'trainmean')
                  .addcov(diag(hp['sigma2'] / weights), 'trainnoise'))      # (= jnp.diag(...) of the reference, built on the device)
            pieces = {'trainmean': 1, 'trainnoise': 1}
            if 'mean' not in hp:
                gp = gp.addcov(k_sigma_mu ** 2, 'mean')
                pieces.update({'mean': 1})
            return gp.addtransf(pieces, 'train')

        # data factory                                                           :196-198
        def info(hp, *, mu_mu, **_):
            return {'train': y_train - hp.get('mean', mu_mu)}

        # what jax.jacfwd of makegp / info gives the reference's fit (forward=True, :213), written out: the train
        # covariance lambda^2 B(alpha, beta) + diag(sigma^2 / w) + k_sigma_mu^2 11' and its derivatives, device resident
        def derivs(hp, *, i_train, weights, splits, mu_mu, mlkw, **_):
            ctx = self._ctx
            lib = _lib.load()
            n = len(i_train)
            F = ctypes.c_double
            K, (dKa, dKb) = makekernel(hp, splits).gram_jvp(i_train, i_train, ctx=ctx)        # lambda^2 B and its tangents
            dKk = DeviceArray.empty((n, n), numpy.float64, ctx)
            _lib.check(lib.lgp_axpby_f64(ctx.handle, n, n, F(-2 / hp['k']), K.ptr, K.ld, F(0), None, 0, dKk.ptr, dKk.ld), 'lgp_axpby')
            invw = DeviceArray.from_host(1 / weights, ctx=ctx)
            dKs = DeviceArray.zeros((n, n), numpy.float64, ctx)
            _lib.check(lib.lgp_add_diag_f64(ctx.handle, n, dKs.ptr, dKs.ld, invw.ptr, F(1)), 'lgp_add_diag')
            _lib.check(lib.lgp_add_diag_f64(ctx.handle, n, K.ptr, K.ld, invw.ptr, F(hp['sigma2'])), 'lgp_add_diag')
            names = ['alpha', 'beta', 'k', 'sigma2']
            dK = [dKa, dKb, dKk, dKs]
            dr = None
            if 'mean' in hp:
                names.append('mean')
                dK.append(None)                                            # the covariance does not depend on the mean
                dr = numpy.zeros((n, len(names)))
                dr[:, -1] = -1
            else:
                _lib.check(lib.lgp_add_scalar_f64(ctx.handle, n, n, K.ptr, K.ld, F(k_sigma_mu ** 2)), 'lgp_add_scalar')
            r = y_train - hp.get('mean', mu_mu)
            return Chol(K, **mlkw), r, dK, dr, names

        # fit hyperparameters                                                    :200-217
        gpkw = dict(i_train=i_train, weights=weights, splits=splits, mu_mu=mu_mu, mlkw=dict(epsrel=0))
        options = dict(
            verbosity=3,
            raises=False,
            minkw=dict(method='l-bfgs-b', options=dict(maxls=4, maxiter=100)),
            mlkw=dict(epsrel=0),
            forward=True,
            gpfactorykw=gpkw,
            gradfactory=derivs,
        )
        options.update(fitkw)
        gpkw['mlkw'] = dict(options['mlkw'])
        fit = empbayes_fit(hyperprior, makegp, info, **options)

        # extract hyperparameters from minimization result                       :219-224
        self.sigma = usqrt(fit.p['sigma2'])
        self.alpha = fit.p['alpha']
        self.beta = fit.p['beta']
        self.meansdev = k_sigma_mu / fit.p['k']
        self.mean = fit.p.get('mean', mu_mu)
        self.fit = fit
        self._ystd = y_train.std()

    def _gethp(self, hp, rng):
        """ ``_bart.py:232-240`` """
        if not isinstance(hp, str):
            return hp
        elif hp == 'map':
            return self.fit.pmean
        elif hp == 'sample':
            return self.fit.sample_hyperparameters(rng)
        else:
            raise KeyError(hp)

    def gp(self, *, hp='map', x_test=None, weights=None, rng=None):
        """
        Create a Gaussian process with the fitted hyperparameters (``_bart.py:242-272``).  The keys of the GP are
        'Xmean', 'Xnoise' and 'X' with X in {'train', 'test'}, X = Xmean + Xnoise (+ the marginalised mean).
        """
        hp = self._gethp(hp, rng)
        return self._gp(hp, x_test, weights, self.fit.gpfactorykw)

    def _gp(self, hp, x_test, weights, gpfactorykw):
        gp = self.fit.gpfactory(hp, **gpfactorykw)
        if x_test is not None:
            x_test = self._to_structured(x_test)
            i_test = self._toindices(x_test, gpfactorykw['splits'])
            assert i_test.dtype == gpfactorykw['i_train'].dtype
            if weights is not None:
                weights = numpy.asarray(weights, float)
                assert weights.shape == i_test.shape
            else:
                weights = numpy.ones(i_test.shape)
            gp = gp.addx(i_test, 'testmean').addcov(diag(hp['sigma2'] / weights), 'testnoise')
            pieces = {'testmean': 1, 'testnoise': 1}
            if 'mean' not in hp:
                pieces.update({'mean': 1})
            gp = gp.addtransf(pieces, 'test')
        return gp

    def data(self, *, hp='map', rng=None):
        """ the dictionary that represents ``y_train`` for `GP.pred` on a GP returned by `gp` (``_bart.py:306-327``) """
        hp = self._gethp(hp, rng)
        return self.fit.data(hp, **self.fit.gpfactorykw)

    def pred(self, *, hp='map', error=False, format='matrices', x_test=None, weights=None, rng=None):
        """
        Predict the outcome at given locations (``_bart.py:329-404``): mean and covariance matrix of the Normal
        posterior of the latent mean (``error=False``) or of the outcome (``error=True``) at ``x_test`` (default:
        the training covariates).
        """
        hp = self._gethp(hp, rng)
        if format == 'gvar':
            raise NotImplementedError('gvar outputs are out of scope of the device path: use format="matrices"')
        elif format != 'matrices':
            raise KeyError(format)
        gp = self._gp(hp, x_test, weights, self.fit.gpfactorykw)
        data = self.fit.data(hp, **self.fit.gpfactorykw)
        label = 'train' if x_test is None else 'test'
        if not error:
            label += 'mean'
        outmean, outcov = gp.predfromdata(data, label, raw=True)
        return outmean + hp.get('mean', self.fit.gpfactorykw['mu_mu']), outcov

    @classmethod
    def _to_structured(cls, x):
        """ ``_bart.py:406-425``: any supported covariate container -> 1-d structured array, one field per covariate """
        if hasattr(x, 'columns') and hasattr(x, 'to_numpy'):
            names = [str(c) for c in x.columns]
            x = recfunctions.unstructured_to_structured(numpy.asarray(x.to_numpy()), names=names)
        else:
            x = numpy.asarray(x)
            if x.dtype.names is None:
                if x.ndim != 2:
                    raise ValueError(f'unstructured covariates must be an (n, p) matrix, found shape {x.shape}')
                x = recfunctions.unstructured_to_structured(x, names=[f'f{i}' for i in range(x.shape[1])])
        assert x.ndim == 1
        for name in x.dtype.names:
            if not numpy.issubdtype(x.dtype[name].base, numpy.number):
                raise TypeError(f'covariate `{name}` is not numerical')
        return x

    @staticmethod
    def _toindices(x, splits):
        ix = BART.indices_from_coord(x, splits)
        return recfunctions.unstructured_to_structured(numpy.ascontiguousarray(ix, numpy.int32), names=list(x.dtype.names))

    def __repr__(self):
        out = f"""BART fit:
alpha = {self.alpha} (0 -> intercept only, 1 -> any)
beta = {self.beta} (0 -> any, ∞ -> no interactions)
mean = {self.mean}
latent sdev = {self.meansdev} (large -> conservative extrapolation)
data total sdev = {self._ystd:.3g}"""
        if self._no_weights:
            out += f"""
error sdev = {self.sigma}"""
        else:
            weights = numpy.array(self.fit.gpfactorykw['weights'])
            avgsigma = (self.sigma ** 2 * float(numpy.mean(1 / weights))).sqrt()
            out += f"""
error sdev (avg weighted) = {avgsigma}
error sdev (unweighted) = {self.sigma}"""
        return out
