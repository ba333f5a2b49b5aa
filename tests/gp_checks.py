"""
GP-level identity tests ported from the reference (``tests/GP/test_pred.py``,
``tests/GP/test_GP.py``, raw-output modes only), written once and run against
both the CPU oracle (``tests/test_oracle_gp.py``) and the device implementation
(``tests/test_gpu_gp.py``).  ``self.lgp`` is the module under test.
"""

import itertools

import numpy as np
import pytest
from pytest import mark

from . import util


class GPChecks:

    lgp = None          # module-like: GP, ExpQuad, Maternp
    exact = True        # whether "same call twice" identities are bit-exact
    validates = True    # whether the GP class under test implements the reference's argument validation and `decomps` (the
                        # oracle restates the numerics only)

    def _need_validation(self):
        if not self.validates:
            pytest.skip('argument validation / decomps are not part of the numerical oracle')

    def assert_equal(self, a, b):
        if isinstance(a, dict):
            assert a.keys() == b.keys()
            for k in a:
                np.testing.assert_array_equal(a[k], b[k])
        else:
            np.testing.assert_array_equal(a, b)

    # ---- tests/GP/test_pred.py ------------------------------------------------

    def _pred(self, seed, err, **kw):
        lgp = self.lgp
        rng = np.random.default_rng(seed)
        x = rng.uniform(-5, 5, size=20)
        xpred = rng.uniform(-10, 10, size=100)
        gp = lgp.GP(lgp.ExpQuad()).addx(x, 'data').addx(xpred, 'pred')
        y = np.tanh(x)
        givencov = None
        if err:
            # the reference draws correlated errors from a Cauchy-kernel GP (test_pred.py:44-49);
            # here the same role is played by an explicit covariance passed as givencov
            ycov = 0.1 ** 2 / (1 + ((x[:, None] - x[None, :]) / 0.3) ** 2)
            givencov = {('data', 'data'): ycov}
        return gp.pred({'data': y}, 'pred', givencov, **kw)

    @mark.parametrize('err', [False, True])
    def test_pred_raw_modes(self, err, rng):
        """ fromdata=True/False raw modes agree when there are no errors (ref test_pred.py:60-76) """
        seed = rng.integers(2 ** 63)
        m1, cov1 = self._pred(seed, err, fromdata=True, raw=True)
        m2, cov2 = self._pred(seed, err, fromdata=False, raw=True)
        assert m1.shape == (100,) and cov1.shape == (100, 100)
        if not err:
            util.assert_allclose(m1, m2, rtol=1e-6, atol=1e-9)
            util.assert_close_matrices(cov1, cov2, rtol=1e-1)

    def test_double_pred(self, rng):
        """ pred from data == pred from the posterior of the data points (ref test_pred.py:78-88) """
        lgp = self.lgp
        n = 50
        gp = lgp.GP(lgp.ExpQuad())
        ax, bx = rng.standard_normal((2, n))
        gp = gp.addx(ax, 'a').addx(bx, 'b')
        m = rng.standard_normal((n, n))
        ymean = rng.standard_normal(n)
        ycov = m.T @ m
        m1, cov1 = gp.predfromdata({'a': ymean}, 'b', {('a', 'a'): ycov}, raw=True)
        ma, cova = gp.predfromdata({'a': ymean}, ['a'], {('a', 'a'): ycov}, raw=True)
        m2, cov2 = gp.predfromfit(ma, 'b', cova, raw=True)
        util.assert_allclose(m2, m1, rtol=1e-7, atol=1e-9)
        util.assert_close_matrices(cov2, cov1, rtol=1e-4)

    # ---- tests/GP/test_GP.py ----------------------------------------------------

    def test_prior_raw_shape(self):
        """ ref :33-40 """
        lgp = self.lgp
        gp = lgp.GP(lgp.ExpQuad()).addx(np.arange(20).reshape(2, 10), 'x')
        cov = gp.prior(raw=True)
        assert cov['x', 'x'].shape == (2, 10, 2, 10)
        cov = gp.prior('x', raw=True)
        assert cov.shape == (2, 10, 2, 10)

    @mark.parametrize('shape', [(20,), (2, 3)])
    def test_halfmatrix(self, shape, rng):
        """ ref :42-50 """
        lgp = self.lgp
        covs = []
        x = rng.standard_normal(shape)
        for checksym in [False, True]:
            gp = lgp.GP(lgp.ExpQuad(), checksym=checksym, halfmatrix=not checksym)
            gp = gp.addx({'x': x})
            covs.append(gp.prior('x', raw=True))
        self.assert_equal(*covs)

    def test_transf_scalar_and_vector(self, rng):
        """ scalar / vector addtransf (ref :52-80, raw form) """
        lgp = self.lgp
        x = np.linspace(0, 3, 7)
        gp = lgp.GP(lgp.ExpQuad()).addx(x, 'x').addx(x + 0.5, 'z')
        gp = gp.addtransf({'x': 2., 'z': -1.}, 's')
        gp = gp.addtransf({'x': np.arange(21.).reshape(3, 7)}, 'v')
        cov = gp.prior(raw=True)
        kxx, kxz, kzz = cov['x', 'x'], cov['x', 'z'], cov['z', 'z']
        util.assert_allclose(cov['s', 's'], 4 * kxx - 2 * kxz - 2 * kxz.T + kzz, rtol=1e-13, atol=1e-13)
        util.assert_allclose(cov['s', 'x'], 2 * kxx - kxz.T, rtol=1e-13, atol=1e-13)
        A = np.arange(21.).reshape(3, 7)
        util.assert_allclose(cov['v', 'v'], A @ kxx @ A.T, rtol=1e-13)
        util.assert_allclose(cov['v', 'z'], A @ kxz, rtol=1e-13)
        util.assert_allclose(cov['z', 'v'], (A @ kxz).T, rtol=1e-13)

    def test_cov(self, rng):
        """ ref :82-87 """
        lgp = self.lgp
        A = rng.standard_normal((20, 20))
        M1 = A.T @ A
        gp = lgp.GP().addcov(M1, 'M')
        M2 = gp.prior('M', raw=True)
        self.assert_equal(M1, M2)

    def test_no_key(self):
        """ ref :349-354 """
        gp = self.lgp.GP(self.lgp.ExpQuad())
        with pytest.raises(ValueError):
            gp.addx(0)
        with pytest.raises(ValueError):
            gp.addcov(1)

    def test_redundant_key(self):
        """ ref :356-361 """
        gp = self.lgp.GP(self.lgp.ExpQuad())
        with pytest.raises(ValueError):
            gp.addx({0: 0}, 0)
        with pytest.raises(ValueError):
            gp.addcov({0: 1}, 0)

    def test_none_key(self):
        """ ref :363-374 """
        lgp = self.lgp
        gp = lgp.GP(lgp.ExpQuad())
        with pytest.raises(ValueError):
            gp.addx({None: 0})
        gp = lgp.GP(lgp.ExpQuad()).addx(0, 0)
        with pytest.raises(ValueError):
            gp.addtransf({0: 1}, None)
        gp = lgp.GP(lgp.ExpQuad())
        with pytest.raises(ValueError):
            gp.addcov({None: 1})

    def test_nonsense_x(self):
        """ ref :376-379 """
        gp = self.lgp.GP(self.lgp.ExpQuad())
        with pytest.raises(TypeError):
            gp.addcov(None, 0)

    def test_key_already_used(self):
        """ ref :381-392 """
        lgp = self.lgp
        gp = lgp.GP(lgp.ExpQuad()).addx(0, 0)
        with pytest.raises(KeyError):
            gp.addx(0, 0)
        with pytest.raises(KeyError):
            gp.addtransf({0: 1}, 0)
        with pytest.raises(KeyError):
            gp.addcov(1, 0)

    def test_missing_key(self):
        """ ref :419-422 """
        gp = self.lgp.GP(self.lgp.ExpQuad()).addx(0, 0)
        with pytest.raises(KeyError):
            gp.addtransf({1: 1}, 2)

    def test_immutable(self):
        """ every add* returns a new object (``_GP/_base.py:86-104``) """
        lgp = self.lgp
        gp = lgp.GP(lgp.ExpQuad())
        gp2 = gp.addx(np.arange(3.), 'a')
        assert 'a' in gp2._elements and 'a' not in gp._elements
        gp3 = gp2.addcov(np.eye(2), 'c')
        assert 'c' in gp3._elements and 'c' not in gp2._elements

    def test_addcov_wrong_blocks(self, rng):
        """ ref :443-468 """
        gp = self.lgp.GP(self.lgp.ExpQuad())
        with pytest.raises(ValueError):
            gp.addcov(np.zeros((1, 1, 1)), 0)
        with pytest.raises(ValueError):
            gp.addcov(np.zeros((1, 2, 2, 1)), 0)
        with pytest.raises(ValueError):
            gp.addcov(rng.standard_normal((10, 10)), 0)
        with pytest.raises(KeyError):
            gp.addcov({(0, 0): 1, (0, 1): 0})
        with pytest.raises(ValueError):
            gp.addcov({(0, 0): np.ones((2, 2)), (1, 1): np.ones((3, 3)), (0, 1): np.ones((3, 2))})
        with pytest.raises(ValueError):
            gp.addcov({(0, 0): np.ones((2, 2)), (1, 1): np.ones((3, 3)), (0, 1): np.ones((2, 3)), (1, 0): np.zeros((3, 2))})

    def test_addcov_no_checksym(self):
        """ ref :470-477 """
        gp = self.lgp.GP(self.lgp.ExpQuad(), checksym=False)
        gp.addcov({(0, 0): np.ones((2, 2)), (1, 1): np.ones((3, 3)), (0, 1): np.ones((2, 3)), (1, 0): np.zeros((3, 2))})

    def test_addcov_missing_block(self):
        """ ref :479-487 """
        gp = self.lgp.GP(self.lgp.ExpQuad(), checkpos=False)
        gp = gp.addcov({(0, 0): np.ones((2, 2)), (1, 1): np.ones((3, 3)), (0, 1): np.ones((2, 3))})
        prior = gp.prior(raw=True)
        self.assert_equal(prior[0, 1], prior[1, 0].T)

    def test_zero_covblock(self, rng):
        """ ref :509-515 """
        gp = self.lgp.GP()
        a = rng.standard_normal((10, 10))
        m = a.T @ a
        gp = gp.addcov(m, 0).addcov(m, 1)
        prior = gp.prior(raw=True)
        self.assert_equal(prior[0, 1], np.zeros_like(m))

    def test_solver_cache(self):
        """ ref :587-596 """
        lgp = self.lgp
        gp = lgp.GP(lgp.ExpQuad())
        x = np.linspace(0, 1, 10)
        y = np.linspace(1, 2, 10)
        z = np.zeros_like(x)
        gp = gp.addx(x, 0).addx(y, 1)
        m1, c1 = gp.predfromdata({0: z}, 1, raw=True)
        assert (0,) in gp._decompcache
        m2, c2 = gp.predfromdata({0: z}, 1, raw=True)
        self.assert_equal(m1, m2)
        self.assert_equal(c1, c2)

    def test_checkpos(self, rng):
        """ ref :598-610 """
        a = rng.standard_normal((20, 20))
        m = a.T @ a
        w, v = np.linalg.eigh(m)
        w[np.arange(len(w)) % 2 == 1] *= -1
        m = (v * w) @ v.T
        gp = self.lgp.GP().addcov(m, 0)
        with pytest.raises(np.linalg.LinAlgError):
            gp.prior(raw=True)
        gp._checkpositive = False
        gp.prior(raw=True)

    def test_given_checks(self, rng):
        """ ref :637-650 """
        lgp = self.lgp
        gp = lgp.GP(lgp.ExpQuad())
        x, y, z = rng.standard_normal((3, 20))
        gp = gp.addx(x, 0).addx(y, 1)
        with pytest.raises(TypeError):
            gp.predfromdata(0, 1, raw=True)
        with pytest.raises(TypeError):
            gp.predfromdata({0: z}, 1, givencov=0, raw=True)
        with pytest.raises(KeyError):
            gp.predfromdata({2: z}, 1, raw=True)
        with pytest.raises(ValueError):
            gp.predfromdata({0: z[:-1]}, 1, raw=True)
        with pytest.raises(TypeError):
            gp.predfromdata({0: np.empty_like(z, str)}, 1, raw=True)

    def test_zero_givencov(self, rng):
        """ ref :652-660 """
        lgp = self.lgp
        gp = lgp.GP(lgp.ExpQuad())
        x, y, z = rng.standard_normal((3, 20))
        gp = gp.addx(x, 0).addx(y, 1)
        cov = np.zeros(2 * x.shape)
        m1, c1 = gp.predfromdata({0: z}, 1, {(0, 0): cov}, raw=True)
        m2, c2 = gp.predfromdata({0: z}, 1, raw=True)
        self.assert_equal(m1, m2)
        self.assert_equal(c1, c2)

    def test_pred_checks(self, rng):
        """ ref :662-678 """
        lgp = self.lgp
        gp = lgp.GP(lgp.ExpQuad())
        x, y, z = rng.standard_normal((3, 20))
        gp = gp.addx(x, 0).addx(y, 1)
        with pytest.raises(ValueError):
            gp.pred({0: z}, 1, raw=True)
        with pytest.raises(ValueError):
            gp.predfromdata({0: z}, 1, raw=True, keepcorr=True)
        with pytest.raises(ValueError):
            gp.predfromdata({0: np.full_like(z, np.nan)}, 1, raw=True)
        with pytest.raises(ValueError):
            gp.predfromdata({0: z}, 1, {(0, 0): np.full(2 * x.shape, np.nan)}, raw=True)
        a = rng.standard_normal((20, 20))
        with pytest.raises(ValueError):
            gp.predfromdata({0: z}, 1, {(0, 0): a}, raw=True)
        gp._checkfinite = False
        gp.predfromdata({0: np.full_like(z, np.nan)}, 1, raw=True)

    def test_pred_all(self, rng):
        """ ref :680-687 """
        lgp = self.lgp
        gp = lgp.GP(lgp.ExpQuad())
        x, y, z = rng.standard_normal((3, 20))
        gp = gp.addx(x, 0).addx(y, 1)
        m1, c1 = gp.predfromdata({0: z}, raw=True)
        m2, c2 = gp.predfromdata({0: z}, [0, 1], raw=True)
        self.assert_equal(m1, m2)
        self.assert_equal(c1, c2)
        assert set(m1) == {0, 1} and set(c1) == {(0, 0), (0, 1), (1, 0), (1, 1)}
        assert c1[0, 1].shape == (20, 20)

    def test_marginal_likelihood_checks(self, rng):
        """ ref :689-704 (non-gvar part) """
        lgp = self.lgp
        gp = lgp.GP(lgp.ExpQuad())
        x, y = rng.standard_normal((2, 20))
        gp = gp.addx(x, 0)
        z = np.full_like(x, np.nan)
        with pytest.raises(ValueError):
            gp.marginal_likelihood({0: z})
        m = np.full(2 * x.shape, np.nan)
        with pytest.raises(ValueError):
            gp.marginal_likelihood({0: y}, {(0, 0): m})
        a = rng.standard_normal(2 * x.shape)
        with pytest.raises(ValueError):
            gp.marginal_likelihood({0: y}, {(0, 0): a})

    def test_marginal_likelihood_value(self, rng):
        """ log N(y; 0, K + C) against a direct dense evaluation """
        from scipy import stats
        lgp = self.lgp
        gp = lgp.GP(lgp.ExpQuad(scale=0.7), checkpos=False)
        x, y = rng.standard_normal((2, 30))
        gp = gp.addx(x, 0)
        a = rng.standard_normal((30, 30))
        c = a.T @ a / 30 + 0.1 * np.eye(30)
        ml = gp.marginal_likelihood({0: y}, {(0, 0): c})
        K = np.exp(-0.5 * ((x[:, None] - x[None, :]) / 0.7) ** 2) + c
        expected = stats.multivariate_normal(np.zeros(30), K, allow_singular=True).logpdf(y)
        util.assert_allclose(ml, expected, rtol=1e-9)
        # kwargs reach the decomposition (bayestree/_bart.py:222 passes epsrel=0)
        ml0 = gp.marginal_likelihood({0: y}, {(0, 0): c}, epsrel=0)
        util.assert_allclose(ml0, expected, rtol=1e-9)

    def test_decompose_nd(self):
        """ ref :960-966 """
        lgp = self.lgp
        cov = np.array(2.)
        d1 = lgp.GP.decompose(cov)
        d2 = lgp.GP.decompose(cov.reshape(1, 1))
        d3 = lgp.GP.decompose(cov.reshape(1, 1, 1, 1))
        util.assert_close_decomps(d1, d2, rtol=1e-15)
        util.assert_close_decomps(d1, d3, rtol=1e-15)
        m = np.arange(16.).reshape(2, 2, 2, 2)
        m = m + m.transpose(2, 3, 0, 1) + 40 * np.eye(4).reshape(2, 2, 2, 2)
        d4 = lgp.GP.decompose(m)
        assert d4.n == 4

    def test_bart_like_structure(self, rng):
        """ the GP structure of ``bayestree/_bart.py:187-205``: K_train = lambda^2 K + diag(sigma2) + c^2 11' """
        lgp = self.lgp
        n, ntest = 25, 6
        x = rng.standard_normal(n); xt = rng.standard_normal(ntest)
        lam2, sigma2, c2 = 1.7, 0.3, 0.5
        gp = (lgp.GP(lgp.ExpQuad() * lam2, checkpos=False, checksym=False)
              .addx(x, 'trainmean').addx(xt, 'testmean')
              .addcov(sigma2 * np.eye(n), 'trainnoise')
              .addcov(c2, 'mean')
              .addtransf({'trainmean': 1, 'trainnoise': 1, 'mean': 1}, 'train'))
        y = rng.standard_normal(n)
        K = lam2 * np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2)
        Ks = lam2 * np.exp(-0.5 * (x[:, None] - xt[None, :]) ** 2)
        Kss = lam2 * np.exp(-0.5 * (xt[:, None] - xt[None, :]) ** 2)
        # 'mean' has shape (): the scalar transformation broadcasts it over the n training points
        Kt = K + sigma2 * np.eye(n) + c2
        m, cov = gp.predfromdata({'train': y}, 'testmean', raw=True)
        util.assert_allclose(m, Ks.T @ np.linalg.solve(Kt, y), rtol=1e-9, atol=1e-12)
        util.assert_close_matrices(cov, Kss - Ks.T @ np.linalg.solve(Kt, Ks), rtol=1e-9)

    # ---- more of tests/GP/test_GP.py (round 3) ---------------------------------------------------------------------------

    def test_lintransf_checks(self):
        """ ref :205-225 """
        self._need_validation()
        lgp = self.lgp
        gp = lgp.GP(lgp.ExpQuad()).addx(0, 0).addx(0, 1)
        with pytest.raises(KeyError):
            gp.addlintransf(lambda x, y: x + y, [0, 1], 0)
        with pytest.raises(ValueError):
            gp.addlintransf(lambda x, y: x + y, [0, 1], None)
        with pytest.raises(KeyError):
            gp.addlintransf(lambda x, y: x + y, [0, 2], 2)
        with pytest.raises(RuntimeError):
            gp.addlintransf(lambda x, y: 1 + x + y, [0, 1], 2)
        with pytest.raises(RuntimeError):
            gp.addlintransf(lambda x, y: 1, [0, 1], 2)
        gp = gp.addlintransf(lambda x, y: 1 + x + y, [0, 1], 2, checklin=False)
        gp._checklin = False
        gp = gp.addlintransf(lambda x, y: 1 + x + y, [0, 1], 3)
        with pytest.raises(RuntimeError):
            gp.addlintransf(lambda x, y: 1 + x + y, [0, 1], 4, checklin=True)

    def test_lintransf_matmul(self, rng):
        """ ref :267-276 """
        lgp = self.lgp
        x = np.arange(20)
        m = rng.standard_normal((30, len(x)))
        gp = lgp.GP(lgp.ExpQuad()).addx(x, 0).addlintransf(lambda v: m @ v, [0], 1)
        prior = gp.prior([0, 1], raw=True)
        util.assert_allclose(m @ prior[0, 0] @ m.T, prior[1, 1], rtol=1e-11)
        util.assert_allclose(m @ prior[0, 1], prior[1, 1], rtol=1e-11)
        util.assert_allclose(prior[1, 0] @ m.T, prior[1, 1], rtol=1e-11)

    def test_compare_transfs_on_elements(self):
        """ ref :158-203, the element-level half (process-level definitions are out of scope): `addtransf` with diagonal
        matrices / scalars and `addlintransf` with the same map give the same prior """
        lgp = self.lgp
        x = np.arange(20.)
        fa = np.sin(x) + 0.5 * np.cos(x ** 2)
        fb = 1 / (1 + np.exp(-x))
        for ta, tb, fun in ((np.diag(fa), np.diag(fb), lambda a, b: fa * a + fb * b), (2, 3, lambda a, b: 2 * a + 3 * b)):
            gp = (lgp.GP(lgp.ExpQuad()).addx(x, 'ax').addx(x + 0.5, 'bx')
                  .addtransf({'ax': ta, 'bx': tb}, 0).addlintransf(fun, ['ax', 'bx'], 1))
            prior = gp.prior([0, 1], raw=True)
            for pair in prior:
                util.assert_allclose(prior[0, 0], prior[pair], atol=1e-13, rtol=1e-13)

    def test_incompatible_dtypes(self):
        """ ref :451-459 """
        self._need_validation()
        lgp = self.lgp
        gp = lgp.GP(lgp.ExpQuad()).addx(0, 0)
        with pytest.raises(TypeError):
            gp.addx(np.zeros(1, 'd,d'), 1)
        gp = lgp.GP(lgp.ExpQuad()).addx(np.zeros(1, 'd,d'), 0)
        with pytest.raises(TypeError):
            gp.addx(np.zeros(1, 'd,d,d'), 2)

    def test_explicit_deriv_and_missing_field(self):
        """ ref :461-469 """
        self._need_validation()
        lgp = self.lgp
        gp = lgp.GP(lgp.ExpQuad())
        with pytest.raises(ValueError):
            gp.addx(0, 0, deriv='x')
        with pytest.raises(ValueError):
            gp.addx(np.array((0, 0), 'f8,f8'), 0, deriv='x')

    def test_nonsense_tensors_and_broadcast(self):
        """ ref :478-498 """
        lgp = self.lgp
        gp = lgp.GP(lgp.ExpQuad()).addx(0, 0)
        with pytest.raises((TypeError, ValueError)):
            gp.addtransf({0: 'a'}, 1)
        with pytest.raises(ValueError):
            gp.addtransf({0: np.inf}, 1)
        gp = gp.addx([0, 1], 1)
        with pytest.raises(ValueError):
            gp.addtransf({1: [1, 2, 3]}, 2)
        gp = lgp.GP(lgp.ExpQuad()).addx([0, 1], 0).addx([0, 1, 2], 1)
        with pytest.raises(ValueError):
            gp.addtransf({0: 1, 1: 1}, 2)
        with pytest.raises(ValueError):
            lgp.GP(lgp.ExpQuad()).addtransf({}, 2)                     # ref :1098-1101

    def test_partial_derivative(self):
        """ ref :539-551: the derivative along a named field of structured points is the derivative of plain points """
        lgp = self.lgp
        x = np.arange(20.)
        y = np.zeros(len(x), 'f8,f8')
        y['f0'] = x
        cov1 = lgp.GP(lgp.ExpQuad()).addx(y, 0, deriv='f0').prior(0, raw=True)
        cov2 = lgp.GP(lgp.ExpQuad()).addx(x, 0, deriv=1).prior(0, raw=True)
        util.assert_allclose(cov1, cov2, rtol=1e-14, atol=1e-15)

    def test_addcov_checks(self, rng):
        """ ref :563-594 """
        self._need_validation()
        lgp = self.lgp
        a = rng.standard_normal((10, 10))
        b = np.copy(a)
        b[0, 0] = np.inf
        m = b.T @ b
        gp = lgp.GP()
        with pytest.raises(ValueError):
            gp.addcov(a, 0)
        with pytest.raises(ValueError):
            gp.addcov(m, 0)
        lgp.GP(checksym=False).addcov(a, 0)
        lgp.GP(checkfinite=False).addcov(m, 0)
        a = a @ a.T
        gp = lgp.GP()
        dec = lgp.GP.decompose(a)
        with pytest.raises(TypeError):
            gp.addcov({(0, 0): a}, decomps=dec)
        with pytest.raises(KeyError):
            gp.addcov({(0, 0): a}, decomps={1: dec})
        with pytest.raises(TypeError):
            gp.addcov({(0, 0): a}, decomps={0: a})
        b = rng.standard_normal((20, 20))
        bd = lgp.GP.decompose(b @ b.T)
        with pytest.raises(ValueError):
            gp.addcov({(0, 0): a}, decomps={0: bd})

    def test_makecovblock_and_covblock_checks(self, rng):
        """ ref :596-632: the checks that were off when a block was given act when it is used """
        lgp = self.lgp
        a = rng.standard_normal((10, 10))
        b = np.copy(a)
        b[0, 0] = np.inf
        m = b.T @ b
        gp = lgp.GP(checksym=False).addcov(a, 0)
        gp._checksym = True
        with pytest.raises(RuntimeError):
            gp.prior(raw=True)
        gp = lgp.GP(checkfinite=False).addcov(m, 0)
        gp._checkfinite = True
        with pytest.raises(RuntimeError):
            gp.prior(raw=True)
        lgp.GP(checksym=False, checkpos=False).addcov(a, 0).prior(raw=True)
        lgp.GP(checkfinite=False, checkpos=False).addcov(m, 0).prior(raw=True)
        a, b, c, d = rng.standard_normal((4, 10, 10))
        blocks = {(0, 0): a.T @ a, (1, 1): b.T @ b, (0, 1): c, (1, 0): d}
        gp = lgp.GP(checksym=False, checkpos=False).addcov(blocks)
        gp2 = lgp.GP(checksym=False, checkpos=False).addcov(blocks)
        gp._checksym = True
        with pytest.raises(RuntimeError):
            gp.prior(raw=True)
        gp2.prior(raw=True)

    def test_priorpoints_cache(self):
        """ ref :679-687 (raw): the same points under two keys give the same blocks """
        lgp = self.lgp
        x = np.arange(20)
        prior = lgp.GP(lgp.ExpQuad()).addx(x, 0).addx(x, 1).prior([0, 1], raw=True)
        self.assert_equal(prior[0, 0], prior[1, 1])
        self.assert_equal(prior[0, 0], prior[0, 1])

    def test_addcov_decomps(self, rng):
        """ ref :1007-1047 (raw): a decomposition handed to `addcov` is the one used """
        self._need_validation()
        lgp = self.lgp
        a = rng.standard_normal((10, 10))
        a = a @ a.T
        blocks = {(0, 0): a[:5, :5], (0, 1): a[:5, 5:], (1, 0): a[5:, :5], (1, 1): a[5:, 5:]}
        dec = lgp.GP.decompose(blocks[0, 0])
        dec1 = lgp.GP.decompose(blocks[1, 1])
        b = rng.standard_normal(5)

        def makez(**kw):
            return lgp.GP().addcov(blocks, **kw).predfromdata({0: b}, 1, raw=True)

        (m1, c1), (m2, c2), (m3, c3) = makez(), makez(decomps={0: dec}), makez(decomps={0: dec1})
        util.assert_allclose(m1, m2, rtol=1e-9, atol=1e-12)
        util.assert_allclose(c1, c2, rtol=1e-9, atol=1e-12)
        assert not np.allclose(m1, m3, rtol=1e-6)

        def makez2(**kw):
            return lgp.GP().addcov(blocks[0, 0], 0, **kw).addcov(blocks[1, 1], 1).predfromdata({0: b}, 1, raw=True)

        (m1, c1), (m2, c2) = makez2(), makez2(decomps=dec)
        util.assert_allclose(m1, m2, rtol=1e-9, atol=1e-12)
        util.assert_allclose(c1, c2, rtol=1e-9, atol=1e-12)

    def test_transf_outer(self):
        """ ref :1088-1096: axes=0 is an outer product """
        lgp = self.lgp
        t = np.arange(5)
        gp = lgp.GP(lgp.ExpQuad()).addx(np.arange(5), 0).addtransf({0: t}, 1, axes=0)
        cov = gp.prior(raw=True)
        util.assert_allclose(np.einsum('i,jl,k', t, cov[0, 0], t), cov[1, 1], rtol=1e-14, atol=1e-15)

    def test_nochecksym_structured_and_not_kernel(self):
        """ ref :1116-1119, :290-292 """
        self._need_validation()
        lgp = self.lgp
        lgp.GP(lgp.ExpQuad(), checksym=False, halfmatrix=True).addx(np.zeros(1, 'd,d'), 0).prior(0, raw=True)
        with pytest.raises(TypeError):
            lgp.GP(0)
