"""
Host driver of the hyperparameter fit: the counterpart of ``lsqfitgp.empbayes_fit``
(``src/lsqfitgp/_fit.py:149-371``) over the device hot path.  It maximises the
marginal posterior of the hyperparameters,

    minus log posterior(z) = minus_log_normal_density(r; K(hp(z))) + 1/2 z'z + const,

in the space z where the hyperprior is a standard Normal (``_fit.py:443-487``), with
``scipy.optimize.minimize`` on the host and every O(N^2) / O(N^3) piece (Gram
build, factorisation, likelihood, its gradient and Fisher matrix) on the GPU.

Differences from the reference, all forced by the absence of jax / gvar here:

* the hyperprior is a dictionary of independent priors: ``(mean, sdev)`` pairs
  (or `UVal`) for Normal variables, `copula` distribution objects for the others.
  gvar's key convention is kept: a key ``'log(x)'`` makes both ``hp['log(x)']``
  and ``hp['x']`` available to ``gpfactory`` (also ``sqrt``, ``erfinv``);
* ``gpfactory`` is not traced.  Its derivatives come, in this order, from (a) ``gradfactory(hp, **gpfactorykw) -> (decomp,
  r, dK, dr, names)``, hand-written analytic derivatives as device matrices; (b) the factory's own arithmetic: the
  hyperparameters are handed over as `Dual` values that carry first-order tangents, and the kernel (``scale``, ``loc``,
  ``kernel * c``, ``kernel + c``; BART's ``alpha`` / ``beta``), `GP.addcov` / ``givencov`` blocks, `GP.addtransf` coefficients
  and the data computed from them yield the derivative of the decomposed matrix and of the residuals next to their values
  (`GP._prior_decomp_tangents`; the first gradient is checked once against central differences, because a factory can drop
  a tangent silently, e.g. ``float(hp['x'])``); (c) central differences of the objective (2 x #hyperparameters device
  evaluations per gradient) for whatever (b) does not cover.  In (a) and (b) the contraction with K^-1 runs on the device
  (`Chol.normal_density_derivs`, ``_linalg/_decomp.py:461-561`` of the reference), Fisher matrix included;
* results are plain dictionaries of floats (``pmean``), of `UVal` (``p``) and of
  covariance blocks (``pcov``) instead of gvars.
"""

import re
import time
import warnings

import numpy
from scipy import optimize, special

from . import copula as _copula
from . import _dual
from ._uncertain import UVal

# gvar.BufferDict's built-in key functions: 'f(x)' in the prior -> x = invf(value) for the model
_KEYFUNCS = {
    'log': (numpy.exp, numpy.exp),                                   # (inverse, derivative of the inverse)
    'sqrt': (numpy.square, lambda v: 2 * v),
    'erfinv': (special.erf, lambda v: 2 / numpy.sqrt(numpy.pi) * numpy.exp(-v * v)),
}
_KEYPATTERN = re.compile(r'^(\w+)\((.+)\)$')


class _Variable:
    """ one scalar hyperparameter: z (standard Normal) -> the values seen by the model """

    def __init__(self, key, prior):
        self.key = key
        m = _KEYPATTERN.match(key)
        self.func = None
        self.inner = None
        if isinstance(prior, _copula.Distr):
            self.dist = prior
            self.mean = self.sdev = None
        else:
            if isinstance(prior, UVal):
                prior = (prior.mean, prior.sdev)
            mean, sdev = prior
            self.dist = None
            self.mean, self.sdev = float(mean), float(sdev)
            if not self.sdev > 0:
                raise ValueError(f'hyperprior[{key!r}]: standard deviation must be positive')
            if m and m.group(1) in _KEYFUNCS:
                self.func, self.inner = m.group(1), m.group(2)

    def primary(self, z):
        return self.dist.invfcn(z) if self.dist is not None else self.mean + self.sdev * z

    def dprimary(self, z):
        return self.dist.dinvfcn(z) if self.dist is not None else self.sdev

    def values(self, z):
        """ {name: value} of everything this variable defines """
        v = float(self.primary(z))
        out = {self.key: v}
        if self.func:
            out[self.inner] = float(_KEYFUNCS[self.func][0](v))
        return out

    def slopes(self, z):
        """ {name: d value / dz} """
        v = float(self.primary(z))
        d = float(self.dprimary(z))
        out = {self.key: d}
        if self.func:
            out[self.inner] = float(_KEYFUNCS[self.func][1](v)) * d
        return out

    def to_z(self, value):
        return float(self.dist.fcn(value)) if self.dist is not None else (float(value) - self.mean) / self.sdev


class _HyperSpace:
    """ flat standardised parametrisation of the hyperprior (``_parse_hyperprior``, ``_fit.py:443-487``) """

    def __init__(self, hyperprior, fix, initial, rng=None):
        if not hasattr(hyperprior, 'keys'):
            raise TypeError('hyperprior must be a dictionary: key -> (mean, sdev) | UVal | copula distribution')
        self.vars = [_Variable(k, v) for k, v in hyperprior.items()]
        names = set()
        for v in self.vars:
            for nm in (v.key, v.inner):
                if nm is not None:
                    if nm in names:
                        raise ValueError(f'duplicate hyperparameter name {nm!r} in hyperprior')
                    names.add(nm)
        fixed = {}
        if fix is not None:
            if not hasattr(fix, 'keys'):
                raise TypeError('fix must be a dictionary of booleans')
            for k, flag in fix.items():
                hit = [v for v in self.vars if k in (v.key, v.inner)]
                if not hit:
                    raise KeyError(f'key {k!r} in fix is missing in hyperprior')
                fixed[hit[0].key] = bool(flag)
        # starting point
        if isinstance(initial, str):
            if initial == 'priormean':
                z0 = numpy.zeros(len(self.vars))
            elif initial == 'priorsample':
                z0 = numpy.random.default_rng(rng).standard_normal(len(self.vars))
            else:
                raise KeyError(initial)
        else:
            if not hasattr(initial, 'keys') or set(initial.keys()) != {v.key for v in self.vars}:
                raise ValueError('initial must be a dictionary with the keys of hyperprior')
            z0 = numpy.array([v.to_z(initial[v.key]) for v in self.vars])
        self.z0 = z0
        self.free = numpy.array([not fixed.get(v.key, False) for v in self.vars], bool)
        self.nfree = int(self.free.sum())

    def full(self, zfree):
        z = self.z0.copy()
        z[self.free] = zfree
        return z

    def unflat(self, zfree):
        z = self.full(numpy.asarray(zfree, float))
        hp = {}
        for v, zi in zip(self.vars, z):
            hp.update(v.values(zi))
        return hp

    def slopes(self, zfree):
        """ name -> (index in the free vector, d value / dz) """
        z = self.full(numpy.asarray(zfree, float))
        out = {}
        pos = numpy.cumsum(self.free) - 1
        for i, (v, zi) in enumerate(zip(self.vars, z)):
            if self.free[i]:
                for nm, d in v.slopes(zi).items():
                    out[nm] = (int(pos[i]), d)
        return out


class empbayes_fit:
    """
    Maximum a posteriori fit of the hyperparameters of a Gaussian process
    (``src/lsqfitgp/_fit.py:149-371``).

    Parameters
    ----------
    hyperprior : dict
        key -> ``(mean, sdev)`` | `UVal` | `copula` distribution.
    gpfactory : callable
        ``gpfactory(hp, **gpfactorykw) -> GP``; ``hp`` is a dictionary of floats.
    data : dict, tuple or callable
        Arguments of ``GP.marginal_likelihood`` (``given`` or ``(given, givencov)``), or a callable
        ``data(hp, **gpfactorykw)`` returning them.
    raises, minkw, gpfactorykw, method, initial, verbosity, covariance, fix, mlkw, additional_loss
        As in the reference.  ``method``: 'nograd' (Nelder-Mead), 'gradient' (BFGS, default), 'fisher'
        (dogleg with Fisher + prior precision; needs ``gradfactory``).
    gradfactory : callable, optional
        ``gradfactory(hp, **gpfactorykw) -> (decomp, r, dK, dr, names)``: analytic derivatives (see the module
        docstring).  ``dK`` is a list of device matrices (or None), ``dr`` an (n, len(names)) array (or None).
    autodiff : bool, default True
        Without a ``gradfactory``: hand the hyperparameters to ``gpfactory`` (and to ``data``, if callable) as values that
        carry first-order tangents (`lsqfitgp_amd.Dual`) and let the kernel, the covariance blocks and the transformation
        coefficients built from them return the derivative of the prior covariance next to the matrix -- the counterpart of
        the reference's ``jax.jacfwd`` of the factory (``_fit.py:598-616``).  Covers kernel ``scale`` / ``loc``, ``kernel * c``,
        ``kernel + c``, `addcov` and ``givencov`` blocks, `addtransf` coefficients and the data; the first operation that is not
        covered (a shape parameter of a kernel, say) switches the fit to central differences for good.
    jit, forward : ignored (there is no tracing compiler here; derivatives are forward-mode).

    Attributes
    ----------
    p : dict of `UVal`;  pmean : dict of floats;  pcov : dict of covariances keyed by pairs of names;
    prior, initial, fix, minresult, minargs, gpfactory, gpfactorykw, data : as in the reference.
    """

    def __init__(self, hyperprior, gpfactory, data, *, raises=True, minkw={}, gpfactorykw={}, jit=True,
                 method='gradient', initial='priormean', verbosity=0, covariance='auto', fix=None, mlkw={},
                 forward=False, additional_loss=None, gradfactory=None, autodiff=True, fdstep=1e-5, rng=None):
        assert callable(gpfactory)
        self._verbosity = int(verbosity)
        self._log('**** call lsqfitgp_amd.empbayes_fit ****', 1)
        space = self._space = _HyperSpace(hyperprior, fix, initial, rng)
        self.prior = dict(hyperprior)
        self.fix = {v.key: not f for v, f in zip(space.vars, space.free)}
        self.initial = space.unflat(space.z0[space.free])
        self.gpfactory = gpfactory
        self.gpfactorykw = dict(gpfactorykw)
        self.gradfactory = gradfactory
        self.data = data
        self._mlkw = dict(mlkw)
        self._additional_loss = additional_loss
        self._fdstep = float(fdstep)
        self._calls = dict(fun=0, grad=0, fisher=0)
        if covariance not in ('auto', 'fisher', 'minhess', 'none'):
            raise KeyError(covariance)
        if method not in ('nograd', 'gradient', 'fisher'):
            raise KeyError(method)
        self._autodiff = bool(autodiff) and gradfactory is None
        self._autodiff_checked = False
        if (method == 'fisher' or covariance == 'fisher') and gradfactory is None and not self._autodiff:
            raise NotImplementedError('the Fisher matrix needs analytic derivatives: pass gradfactory')
        if (method == 'fisher' or covariance == 'fisher') and additional_loss is not None:
            raise NotImplementedError('Fisher matrix not implemented with additional_loss')     # _fit.py:737-741

        x0 = space.z0[space.free].copy()
        if method == 'nograd':
            minargs = dict(fun=self._fun, x0=x0, method='nelder-mead')
        elif method == 'gradient':
            minargs = dict(fun=self._fun_and_jac, jac=True, x0=x0, method='bfgs')
        else:
            minargs = dict(fun=self._fun_and_jac, jac=True, hess=self._fisher, x0=x0, method='dogleg')
        it = [0]
        stamp = [time.perf_counter()]

        def callback(xk, *_):
            it[0] += 1
            now = time.perf_counter()
            self._log(f'iter {it[0]}, time: {(now - stamp[0]) * 1e3:.1f} ms, calls: {self._calls}', 3)
            stamp[0] = now

        minargs.update(callback=callback)
        minargs.update(minkw)
        self._log(f'minimizer method {minargs["method"]!r}, {space.nfree}/{len(space.vars)} free hyperparameters', 2)
        total = time.perf_counter()
        if space.nfree:
            result = optimize.minimize(**minargs)
        else:
            result = optimize.OptimizeResult(x=x0, success=True, message='no free hyperparameters', fun=self._fun(x0))
        total = time.perf_counter() - total

        if result.success:
            self._log(f'minimization succeeded: {result.message}', 1)
        else:
            msg = f'minimization failed: {result.message}'
            if raises:
                raise RuntimeError(msg)
            elif self._verbosity == 0:
                warnings.warn(msg)
            else:
                self._log(msg, 1)

        cov = self._posterior_covariance(method, covariance, result)
        self._log(f'calls: {self._calls}, total time: {total:.2f} s', 1)

        # posterior of the hyperparameters in the model's variables: first-order propagation through hp(z)
        zfree = numpy.asarray(result.x, float)
        self.minresult = result
        self.minargs = minargs
        self.zmean = zfree
        self.zcov = cov
        self.pmean = space.unflat(zfree)
        slopes = space.slopes(zfree)
        names = list(self.pmean)
        J = numpy.zeros((len(names), max(space.nfree, 1)))
        for a, nm in enumerate(names):
            if nm in slopes:
                J[a, slopes[nm][0]] = slopes[nm][1]
        if cov.ndim == 2:
            full = J[:, :cov.shape[0]] @ cov @ J[:, :cov.shape[0]].T
        else:
            full = numpy.full((len(names), len(names)), numpy.nan)
            for a, nm in enumerate(names):
                if nm not in slopes:
                    full[a, :] = full[:, a] = 0
        self.pcov = {(a, b): full[i, j] for i, a in enumerate(names) for j, b in enumerate(names)}
        self.p = {nm: UVal(self.pmean[nm], numpy.sqrt(full[i, i]) if full[i, i] >= 0 else numpy.nan) for i, nm in enumerate(names)}
        if self._verbosity >= 2:
            for nm in names:
                self._log(f'{nm:>20s}  {self.p[nm]}', 2)
        self._log('**** exit lsqfitgp_amd.empbayes_fit ****', 1)

    # ---- logging --------------------------------------------------------------------------------------------------

    def _log(self, message, level):
        if self._verbosity >= level:
            print(message)

    # ---- objective -------------------------------------------------------------------------------------------------

    def _args(self, hp):
        if callable(self.data):
            args = self.data(hp, **self.gpfactorykw)
        else:
            args = self.data
        if not isinstance(args, tuple):
            args = (args,)
        return args

    @staticmethod
    def _prior_value(z):
        # the marginal prior of z is a standard Normal by construction (_fit.py:664-667)
        return 0.5 * (len(z) * numpy.log(2 * numpy.pi) + z @ z)

    def _fun(self, z):
        """ minus log marginal posterior of the hyperparameters (not normalised), ``_fit.py:676-683``.  A covariance matrix
        that cannot be factored gives nan, as the reference's traced function does (its Cholesky raises only outside a trace,
        ``_linalg/_decomp.py:389-391``), so that the minimiser backs off instead of the fit dying """
        self._calls['fun'] += 1
        z = numpy.asarray(z, float)
        hp = self._space.unflat(z)
        try:
            gp = self.gpfactory(hp, **self.gpfactorykw)
            decomp, r = gp._prior_decomp(*self._args(hp), **self._mlkw)
        except numpy.linalg.LinAlgError:
            return numpy.nan
        value = decomp.minus_log_normal_density(numpy.asarray(r, float), value=True)[0]
        loss = 0. if self._additional_loss is None else float(self._additional_loss(hp))
        return float(value) + self._prior_value(z) + loss

    def _loss_grad(self, z):
        if self._additional_loss is None:
            return 0.
        g = numpy.zeros(len(z))
        for i in range(len(z)):
            e = numpy.zeros(len(z)); e[i] = self._fdstep
            g[i] = (self._additional_loss(self._space.unflat(z + e)) - self._additional_loss(self._space.unflat(z - e))) / (2 * self._fdstep)
        return g

    def _derivs(self, z, fisher):
        hp = self._space.unflat(z)
        decomp, r, dK, dr, names = self.gradfactory(hp, **self.gpfactorykw)
        value, grad, fish = decomp.normal_density_derivs(r, dK, dr, fisher=fisher)
        slopes = self._space.slopes(z)
        # chain rule to z: the analytic derivatives are with respect to the model's variables `names`
        T = numpy.zeros((len(names), len(z)))
        for k, nm in enumerate(names):
            if nm not in self._space.unflat(z):
                raise KeyError(f'gradfactory returned a derivative for {nm!r}, which is not a hyperparameter')
            if nm in slopes:
                T[k, slopes[nm][0]] = slopes[nm][1]
        return float(value), T.T @ numpy.asarray(grad, float), (None if fish is None else T.T @ numpy.asarray(fish, float) @ T)

    def _auto_derivs(self, z, fisher):
        """ value, gradient and Fisher matrix in z by forward-mode differentiation of the factory's own arithmetic: the
        hyperparameters go to ``gpfactory`` (and ``data``) as `_dual.Dual` values whose tangents are their slopes in z, and the
        GP assembles the derivative of its prior covariance next to the matrix (`GP._prior_decomp_tangents`).  The counterpart
        of ``jax.jacfwd`` in ``_fit.py:598-616``. """
        values = self._space.unflat(z)
        slopes = self._space.slopes(z)
        hp = {nm: (_dual.Dual(float(v), {slopes[nm][0]: float(slopes[nm][1])}) if nm in slopes else v) for nm, v in values.items()}
        gp = self.gpfactory(hp, **self.gpfactorykw)
        if not hasattr(gp, '_prior_decomp_tangents'):
            raise NotImplementedError(f'{type(gp).__module__}.{type(gp).__name__} does not assemble tangents')
        decomp, r, dK, dr = gp._prior_decomp_tangents(*self._args(hp), tkeys=range(len(z)), **self._mlkw)
        value, grad, fish = decomp.normal_density_derivs(numpy.asarray(r, float), dK, dr, fisher=fisher)
        return float(value), numpy.asarray(grad, float), (None if fish is None else numpy.asarray(fish, float))

    def _analytic(self, z, fisher):
        """ ``(value, gradient, fisher)`` from `gradfactory` if given, else from the tangents the factory's arithmetic carries;
        None when neither is available (then: central differences) """
        if self.gradfactory is not None:
            return self._derivs(z, fisher)
        if self._autodiff:
            try:
                res = self._auto_derivs(z, fisher)
            except (NotImplementedError, TypeError) as exc:
                self._autodiff = False
                self._log(f'no analytic derivatives for this factory ({exc}): central differences', 1)
                return None
            if not self._autodiff_checked:
                # once per fit: a factory can lose a tangent without an error (numpy.full(n, hp[...]), math.exp(hp[...]) turn
                # the hyperparameter into a plain number), so the first gradient is held against central differences --
                # component by component (a dropped tangent whose gradient is small beside the largest one must not pass),
                # with the noise of the differences themselves as the absolute part of the tolerance.  The check only counts
                # as done once the comparison has been made: a LinAlgError inside the differences leaves it for the next point
                g = self._fd_gradient(z)
                self._autodiff_checked = True
                value = abs(float(res[0]))
                atol = 1e2 * numpy.finfo(float).eps * max(value, 1.) / self._fdstep + 1e-3 * self._fdstep ** 2 * max(value, 1.)
                bad = ~numpy.isfinite(res[1]) | (numpy.abs(res[1] - g) > 1e-3 * numpy.abs(g) + atol)
                if numpy.any(bad):
                    self._autodiff = False
                    self._log(f'the tangents of the factory do not reproduce its finite differences at components {numpy.flatnonzero(bad).tolist()} '
                              f'({res[1]} vs {g}, tolerance 1e-3 relative + {atol:.1e}): central differences', 1)
                    return None
            return res
        return None

    def _fd_gradient(self, z):
        """ central differences of the likelihood term alone (what `_analytic` returns the gradient of) """
        g = numpy.empty(len(z))
        h = self._fdstep
        for i in range(len(z)):
            e = numpy.zeros(len(z)); e[i] = h
            vals = []
            for zz in (z + e, z - e):
                hp = self._space.unflat(zz)
                gp = self.gpfactory(hp, **self.gpfactorykw)
                decomp, r = gp._prior_decomp(*self._args(hp), **self._mlkw)
                vals.append(float(decomp.minus_log_normal_density(numpy.asarray(r, float), value=True)[0]))
            g[i] = (vals[0] - vals[1]) / (2 * h)
        return g

    def _fun_and_jac(self, z):
        """ the objective and its gradient (``_fit.py:716-725``) """
        z = numpy.asarray(z, float)
        self._calls['grad'] += 1
        if self.gradfactory is not None or self._autodiff:
            try:
                res = self._analytic(z, False)
            except numpy.linalg.LinAlgError:
                return numpy.nan, numpy.full(len(z), numpy.nan)
            if res is not None:
                value, g, _ = res
                loss = 0. if self._additional_loss is None else float(self._additional_loss(self._space.unflat(z)))
                return value + self._prior_value(z) + loss, g + z + self._loss_grad(z)
        f0 = self._fun(z)
        g = numpy.empty(len(z))
        h = self._fdstep
        for i in range(len(z)):
            e = numpy.zeros(len(z)); e[i] = h
            g[i] = (self._fun(z + e) - self._fun(z - e)) / (2 * h)
        return f0, g

    def _fisher(self, z):
        """ Fisher information plus the prior precision (``_fit.py:733-745``) """
        self._calls['fisher'] += 1
        res = self._analytic(numpy.asarray(z, float), True)
        if res is None:
            raise NotImplementedError('the Fisher matrix needs analytic derivatives: this factory has none, pass gradfactory')
        return res[2] + numpy.eye(len(z))

    def _posterior_covariance(self, method, covariance, result):
        """ ``_fit.py:819-862`` """
        n = len(result.x)
        if covariance == 'auto':
            covariance = 'minhess' if (hasattr(result, 'hess_inv') or hasattr(result, 'hess')) else 'none'
        if covariance == 'fisher':
            prec = result.hess if method == 'fisher' and hasattr(result, 'hess') else self._fisher(result.x)
            return numpy.linalg.inv(numpy.asarray(prec, float))
        if covariance == 'minhess' and self.gradfactory is None and not self._autodiff and method != 'nograd' and n:
            # finite-difference gradients leave the quasi-Newton estimate of the inverse Hessian rough (the line search
            # stops on rounding noise after a few updates); with so few parameters the Hessian of the objective at the mode
            # is cheap to difference directly: n^2 + n + 1 more evaluations
            H = self._fd_hessian(numpy.asarray(result.x, float))
            if numpy.all(numpy.isfinite(H)) and numpy.all(numpy.linalg.eigvalsh(H) > 0):
                return numpy.linalg.inv(H)
        if covariance == 'minhess':
            if hasattr(result, 'hess_inv'):
                hi = result.hess_inv
                if isinstance(hi, optimize.LbfgsInvHessProduct):
                    # the L-BFGS pairs replayed as dense BFGS updates (_fit.py:864-870)
                    bfgs = optimize.BFGS()
                    bfgs.initialize(hi.shape[0], 'inv_hess')
                    for i in range(hi.n_corrs):
                        bfgs.update(hi.sk[i], hi.yk[i])
                    return numpy.asarray(bfgs.get_matrix(), float)
                return numpy.asarray(hi, float)
            if hasattr(result, 'hess'):
                return numpy.linalg.inv(numpy.asarray(result.hess, float))
            raise RuntimeError('the minimizer did not return an estimate of the hessian')
        return numpy.full(n, numpy.nan)

    def _fd_hessian(self, z, h=1e-3):
        n = len(z)
        H = numpy.empty((n, n))
        f0 = self._fun(z)
        E = numpy.eye(n) * h
        fp = [self._fun(z + E[i]) for i in range(n)]
        fm = [self._fun(z - E[i]) for i in range(n)]
        for i in range(n):
            H[i, i] = (fp[i] - 2 * f0 + fm[i]) / (h * h)
            for j in range(i):
                H[i, j] = H[j, i] = (self._fun(z + E[i] + E[j]) - fp[i] - fp[j] + 2 * f0 - fm[i] - fm[j] + self._fun(z - E[i] - E[j])) / (2 * h * h)
        return H

    # ---- convenience -------------------------------------------------------------------------------------------------

    def sample_hyperparameters(self, rng=None):
        """ a draw from the Laplace approximation of the hyperparameter posterior, in the model's variables """
        rng = numpy.random.default_rng(rng)
        z = self.zmean
        if self.zcov.ndim == 2 and len(z):
            w, v = numpy.linalg.eigh(self.zcov)
            z = z + v @ (numpy.sqrt(numpy.maximum(w, 0)) * rng.standard_normal(len(z)))
        return self._space.unflat(z)
