"""
CPU restatement (numpy) of the raw-output subset of ``lsqfitgp.GP`` -- test
infrastructure only, see ``oracle/__init__.py``.  parity unpinned vs JAX-CPU.

Follows ``src/lsqfitgp/_GP/_gp.py:114-129`` (ctor), ``_GP/_base.py:86-104``
(clone-on-write), ``_GP/_elements.py:158-346,408-649,758-826`` (addx,
addtransf with scalar/tensor coefficients, addcov, covariance blocks, prior)
and ``_GP/_compute.py:45-136,138-305,336-422,430-516`` (solver cache,
pred, marginal_likelihood, decompose).
"""

import copy
import math

import numpy
from scipy.sparse import linalg as sparse_linalg

from . import decomp as _decomp


class _Points:
    def __init__(self, x, deriv=0):
        self.x = x
        self.deriv = deriv
    @property
    def shape(self):
        return self.x.shape
    @property
    def size(self):
        return math.prod(self.shape)


class _LinTransf:
    """ scalar/tensor-coefficient transformation registered by addtransf
    (``_elements.py:248-346``) """
    def __init__(self, tensors, keys, shape, axes):
        self.tensors = tensors
        self.keys = keys
        self.shape = shape
        self.axes = axes
    @property
    def size(self):
        return math.prod(self.shape)


class _Cov:
    def __init__(self, blocks, shape):
        self.blocks = blocks
        self.shape = shape
    @property
    def size(self):
        return math.prod(self.shape)


class GP:

    def __init__(self, covfun=None, *, solver='chol', checkpos=True, checksym=True,
        checkfinite=True, checklin=True, posepsfac=1, halfmatrix=False, **kw):
        """ ``_gp.py:114-129`` """
        self._covfun = covfun
        self._elements = {}
        self._covblocks = {}
        self._decompcache = {}
        self._checkpositive = bool(checkpos)
        self._checksym = bool(checksym)
        self._checkfinite = bool(checkfinite)
        self._posepsfac = float(posepsfac)
        self._halfmatrix = bool(halfmatrix)
        assert not (halfmatrix and checksym)
        decomp = self._getdecomp(solver)
        self._decompclass = lambda K, **kwargs: decomp(K, **kwargs, **kw)
        self._checkpos_cache = []

    def _clone(self):
        new = copy.copy(self)
        new._elements = self._elements.copy()
        new._covblocks = self._covblocks.copy()
        new._decompcache = self._decompcache.copy()
        new._checkpos_cache = list(self._checkpos_cache)
        return new

    # ---- elements ------------------------------------------------------

    @staticmethod
    def _deriv(spec):
        """ ``_Deriv.py:60-97``: None | int | str | iterable of str with int multipliers -> 0, int or {field: order} """
        import collections
        c = collections.Counter()
        if spec is None:
            return 0
        if isinstance(spec, (int, numpy.integer)):
            assert spec >= 0
            return int(spec)
        if isinstance(spec, str):
            return {spec: 1}
        integer = None
        for obj in spec:
            if isinstance(obj, str):
                if integer is not None:
                    if integer:
                        c.update({obj: integer})
                    integer = None
                else:
                    c.update([obj])
            else:
                assert integer is None and obj >= 0
                integer = int(obj)
        assert integer is None
        return dict(c) or 0

    def addx(self, x, key=None, *, deriv=0):
        """ ``_elements.py:158-242`` (default process only; `deriv` as ``Deriv`` reads it, ``_Deriv.py``) """
        deriv = self._deriv(deriv)
        self = self._clone()
        if hasattr(x, 'keys'):
            if key is not None:
                raise ValueError('can not specify key if x is a dictionary')
            if None in x:
                raise ValueError('None key in x not allowed')
        else:
            if key is None:
                raise ValueError('x is not dictionary but key is None')
            x = {key: x}
        for key in x:
            if key in self._elements:
                raise KeyError('key {!r} already in GP'.format(key))
            self._elements[key] = _Points(numpy.asarray(x[key]), deriv)
        return self

    def addtransf(self, tensors, key, *, axes=1):
        """ ``_elements.py:248-346`` """
        self = self._clone()
        if key is None:
            raise ValueError('key can not be None')
        if key in self._elements:
            raise KeyError(f'key {key!r} already in GP')
        for k in tensors:
            if k not in self._elements:
                raise KeyError(k)
        if len(tensors) == 0:
            raise ValueError('empty tensors, undetermined output shape')
        tens = {}
        for k, t in tensors.items():
            t = numpy.asarray(t)
            if self._checkfinite and not numpy.all(numpy.isfinite(t)):
                raise ValueError(f'tensors[{k!r}] contains infs/nans')
            rshape = self._elements[k].shape
            if t.shape and t.shape[t.ndim - axes:] != rshape[:axes]:
                raise ValueError(f'tensors[{k!r}].shape = {t.shape!r} can not be multiplied with shape {rshape!r} with {axes}-axes contraction')
            tens[k] = t
        shapes = [
            t.shape[:t.ndim - axes] + self._elements[k].shape[axes:] if t.shape else self._elements[k].shape
            for k, t in tens.items()
        ]
        shape = numpy.broadcast_shapes(*shapes)
        self._elements[key] = _LinTransf(tens, list(tens.keys()), shape, axes)
        return self

    def addlintransf(self, transf, keys, key, *, checklin=None):
        """ ``_elements.py:348-406``; the coefficient matrices as ``_LinTransf.matrices`` (``:129-147``): the function applied to
        the basis vectors of one input with the others at zero """
        self = self._clone()
        if key is None:
            raise ValueError('key can not be None')
        if key in self._elements:
            raise KeyError(f'key {key!r} already in GP')
        for k in keys:
            if k not in self._elements:
                raise KeyError(k)
        elems = [self._elements[k] for k in keys]
        shape = numpy.asarray(transf(*[numpy.zeros(e.shape) for e in elems])).shape
        tens, axes = {}, {}
        for i, (k, elem) in enumerate(zip(keys, elems)):
            rows = []
            for e in range(elem.size):
                inputs = [numpy.zeros(ej.shape) for ej in elems]
                inputs[i].reshape(-1)[e] = 1
                rows.append(numpy.asarray(transf(*inputs), float).reshape(-1))
            M = numpy.array(rows).reshape(elem.size, -1).T if rows else numpy.zeros((int(numpy.prod(shape, dtype=int)), 0))
            t = M.reshape(tuple(shape) + elem.shape)
            tens[k] = tens[k] + t if k in tens else t
            axes[k] = len(elem.shape)
        self._elements[key] = _LinTransf(tens, list(tens.keys()), tuple(shape), axes)
        return self

    def addcov(self, covblocks, key=None):
        """ ``_elements.py:408-552`` (without ``decomps``) """
        self = self._clone()
        if hasattr(covblocks, 'keys'):
            if key is not None:
                raise ValueError('can not specify key if covblocks is a dictionary')
            if None in covblocks:
                raise ValueError('None key in covblocks not allowed')
        else:
            if key is None:
                raise ValueError('covblocks is not dictionary but key is None')
            covblocks = {(key, key): covblocks}
        shapes = {}
        preblocks = {}
        for keys, block in covblocks.items():
            for key in keys:
                if key in self._elements:
                    raise KeyError(f'key {key!r} already in GP')
            xkey, ykey = keys
            if block is None:
                raise TypeError(f'block {keys!r} is None')
            block = numpy.asarray(block)
            if xkey == ykey:
                if block.ndim % 2 == 1:
                    raise ValueError(f'diagonal block {key!r} has odd number of axes')
                half = block.ndim // 2
                head = block.shape[:half]
                tail = block.shape[half:]
                if head != tail:
                    raise ValueError(f'shape {block.shape!r} of diagonal block {key!r} is not symmetric')
                shapes[xkey] = head
                if self._checksym and not numpy.allclose(block, block.T):
                    raise ValueError(f'diagonal block {key!r} is not symmetric')
            preblocks[keys] = block
        blocks = {}
        for keys, block in preblocks.items():
            if self._checkfinite and not numpy.all(numpy.isfinite(block)):
                raise ValueError(f'block {keys!r} not finite')
            xkey, ykey = keys
            if xkey == ykey:
                size = math.prod(shapes[xkey])
                blocks[keys] = block.reshape((size, size))
            else:
                for key in keys:
                    if key not in shapes:
                        raise KeyError(f'key {key!r} from off-diagonal block {keys!r} not found in diagonal blocks')
                eshape = shapes[xkey] + shapes[ykey]
                if block.shape != eshape:
                    raise ValueError(f'shape {block.shape!r} of block {keys!r} is not {eshape!r} as expected from diagonal blocks')
                xsize = math.prod(shapes[xkey])
                ysize = math.prod(shapes[ykey])
                block = block.reshape((xsize, ysize))
                blocks[keys] = block
                revkeys = keys[::-1]
                if preblocks.get(revkeys) is None:
                    blocks[revkeys] = block.T
        if self._checksym:
            for keys, block in blocks.items():
                xkey, ykey = keys
                if xkey != ykey:
                    blockT = blocks[ykey, xkey]
                    if not numpy.allclose(block.T, blockT):
                        raise ValueError(f'block {keys!r} is not the transpose of block {keys[::-1]!r}')
        for key, shape in shapes.items():
            self._elements[key] = _Cov(blocks, shape)
        return self

    # ---- covariance blocks ----------------------------------------------

    def _makecovblock_points(self, xkey, ykey):
        """ ``_elements.py:554-579`` """
        x = self._elements[xkey]
        y = self._elements[ykey]
        kernel = self._covfun
        if x.deriv or y.deriv:
            kernel = kernel.diff(x.deriv, y.deriv)          # :565
        if x is y and not self._checksym and self._halfmatrix:
            n = x.size
            ix, iy = numpy.triu_indices(n)
            back = numpy.empty((n, n), ix.dtype)
            a = numpy.arange(ix.size)
            back[ix, iy] = a
            back[iy, ix] = a
            flat = x.x.reshape(-1)
            halfcov = kernel(flat[ix], flat[iy])
            return halfcov[back]
        ax = x.x.reshape(-1)[:, None]
        ay = y.x.reshape(-1)[None, :]
        return kernel(ax, ay)

    def _makecovblock_lintransf_any(self, xkey, ykey):
        """ ``_elements.py:581-601`` specialised to the tensordot transformation
        built by addtransf (``:332-344``) """
        x = self._elements[xkey]
        y = self._elements[ykey]
        out = None
        for k in x.keys:
            elem = self._elements[k]
            cov = self._covblock(k, ykey)
            assert cov.shape == (elem.size, y.size)
            cov = cov.reshape(elem.shape + (y.size,))
            t = x.tensors[k]
            if t.shape:
                b = numpy.tensordot(t, cov, x.axes[k] if isinstance(x.axes, dict) else x.axes)
            else:
                b = t * cov
            out = b if out is None else out + b
        out = numpy.broadcast_to(out, x.shape + (y.size,))
        return out.reshape((x.size, y.size))

    def _makecovblock(self, xkey, ykey):
        """ ``_elements.py:603-624`` """
        x = self._elements[xkey]
        y = self._elements[ykey]
        if isinstance(x, _Points) and isinstance(y, _Points):
            cov = self._makecovblock_points(xkey, ykey)
        elif isinstance(x, _LinTransf):
            cov = self._makecovblock_lintransf_any(xkey, ykey)
        elif isinstance(y, _LinTransf):
            cov = self._makecovblock_lintransf_any(ykey, xkey).T
        elif isinstance(x, _Cov) and isinstance(y, _Cov) and x.blocks is y.blocks and (xkey, ykey) in x.blocks:
            cov = x.blocks[xkey, ykey]
        else:
            cov = numpy.zeros((x.size, y.size))
        if self._checkfinite and not numpy.all(numpy.isfinite(cov)):
            raise RuntimeError(f'covariance block {(xkey, ykey)!r} is not finite')
        if self._checksym and xkey == ykey and not numpy.allclose(cov, cov.T):
            raise RuntimeError(f'covariance block {(xkey, ykey)!r} is not symmetric')
        return cov

    def _covblock(self, row, col):
        """ ``_elements.py:626-640`` """
        if (row, col) not in self._covblocks:
            block = self._makecovblock(row, col)
            if row != col:
                if self._checksym:
                    blockT = self._makecovblock(col, row)
                    if not numpy.allclose(block.T, blockT):
                        raise RuntimeError('covariance block {!r} is not symmetric'.format((row, col)))
                self._covblocks[col, row] = block.T
            self._covblocks[row, col] = block
        return self._covblocks[row, col]

    def _assemblecovblocks(self, rowkeys, colkeys=None):
        """ ``_elements.py:642-649`` """
        if colkeys is None:
            colkeys = rowkeys
        return numpy.block([[self._covblock(row, col) for col in colkeys] for row in rowkeys])

    def _checkpos(self, cov):
        """ ``_elements.py:651-668`` (dense eigvalsh stands in for the two lobpcg runs) """
        if len(cov) <= 2000:
            w = numpy.linalg.eigvalsh(numpy.asarray(cov, float))
            mineigv, maxeigv = w[0], w[-1]
        else:
            X = numpy.random.randn(len(cov), 1)
            A = numpy.asarray(cov)
            (mineigv,), _ = sparse_linalg.lobpcg(A, X, largest=False)
            (maxeigv,), _ = sparse_linalg.lobpcg(A, X, largest=True)
        if mineigv < 0:
            bound = -len(cov) * numpy.finfo(cov.dtype).eps * maxeigv * self._posepsfac
            if mineigv < bound:
                msg = 'covariance matrix is not positive definite: '
                msg += 'mineigv = {:.4g} < {:.4g}'.format(mineigv, bound)
                raise numpy.linalg.LinAlgError(msg)

    def _checkpos_keys(self, keys):
        """ ``_elements.py:672-680`` """
        if not self._checkpositive:
            return
        keys = set(keys)
        for prev_keys in self._checkpos_cache:
            if keys.issubset(prev_keys):
                return
        cov = self._assemblecovblocks(list(keys))
        self._checkpos(cov)
        self._checkpos_cache.append(keys)

    def prior(self, key=None, *, raw=False):
        """ ``_elements.py:758-817`` raw branches """
        if not raw:
            raise NotImplementedError('gvar outputs are out of scope: use raw=True')
        if key is None:
            outkeys = list(self._elements)
        elif isinstance(key, list):
            outkeys = key
        else:
            outkeys = None
        self._checkpos_keys([key] if outkeys is None else outkeys)
        if outkeys is not None:
            return {
                (row, col): self._covblock(row, col).reshape(self._elements[row].shape + self._elements[col].shape)
                for row in outkeys for col in outkeys
            }
        return self._covblock(key, key).reshape(2 * self._elements[key].shape)

    def _slices(self, keylist):
        """ ``_elements.py:819-826`` """
        sizes = [self._elements[key].size for key in keylist]
        stops = numpy.pad(numpy.cumsum(sizes), (1, 0))
        return [slice(stops[i - 1], stops[i]) for i in range(1, len(stops))]

    # ---- compute ---------------------------------------------------------

    @staticmethod
    def _getdecomp(solver):
        """ ``_compute.py:424-428`` """
        return {'chol': _decomp.Chol}[solver]

    def _solver(self, keys, ycov=None, **kw):
        """ ``_compute.py:45-94`` """
        keys = tuple(keys)
        if ycov is None:
            cache = self._decompcache.get(keys)
            if cache is not None:
                return cache
        Kxx = self._assemblecovblocks(keys)
        if ycov is not None:
            Kxx = Kxx + ycov
        decomp = self._decompclass(Kxx, **kw)
        if ycov is None:
            self._decompcache[keys] = decomp
        return decomp

    def _flatgiven(self, given, givencov):
        """ ``_compute.py:96-136`` """
        if not hasattr(given, 'keys'):
            raise TypeError('`given` must be dict')
        gcblack = givencov is None or isinstance(givencov, _decomp.Decomposition)
        if not gcblack and not hasattr(givencov, 'keys'):
            raise TypeError('`givenconv` must be None, dict or Decomposition')
        ylist = []
        keylist = []
        for key, l in given.items():
            if key not in self._elements:
                raise KeyError(key)
            l = numpy.asarray(l)
            shape = self._elements[key].shape
            if l.shape != shape:
                raise ValueError('given[{!r}] has shape {!r} different from shape {!r}'.format(key, l.shape, shape))
            if l.dtype != object and not numpy.issubdtype(l.dtype, numpy.number):
                raise TypeError('given[{!r}] has non-numerical dtype {!r}'.format(key, l.dtype))
            ylist.append(l.reshape(-1))
            keylist.append(key)
        if gcblack:
            covblocks = givencov
        else:
            covblocks = [
                [
                    numpy.asarray(givencov[keylist[i], keylist[j]]).reshape(ylist[i].shape + ylist[j].shape)
                    for j in range(len(keylist))
                ]
                for i in range(len(keylist))
            ]
        return ylist, keylist, covblocks

    def _check_ymean(self, ymean):
        if self._checkfinite and not numpy.all(numpy.isfinite(ymean)):
            raise ValueError('mean of `given` is not finite')

    def _check_ycov(self, ycov):
        if ycov is None:
            return
        if self._checkfinite and not numpy.all(numpy.isfinite(ycov)):
            raise ValueError('covariance matrix of `given` is not finite')
        if self._checksym and not numpy.allclose(ycov, ycov.T):
            raise ValueError('covariance matrix of `given` is not symmetric')

    def pred(self, given, key=None, givencov=None, *, fromdata=None, raw=False, keepcorr=None):
        """ ``_compute.py:138-305`` raw branch """
        if fromdata is None:
            raise ValueError('you must specify if `given` is data or fit result')
        fromdata = bool(fromdata)
        raw = bool(raw)
        if keepcorr is None:
            keepcorr = not raw
        if keepcorr and raw:
            raise ValueError('both keepcorr=True and raw=True')
        if not raw:
            raise NotImplementedError('gvar outputs are out of scope: use raw=True')
        strip = False
        if key is None:
            outkeys = list(self._elements)
        elif isinstance(key, list):
            outkeys = key
        else:
            outkeys = [key]
            strip = True
        outslices = self._slices(outkeys)
        ylist, inkeys, ycovblocks = self._flatgiven(given, givencov)
        y = numpy.concatenate(ylist)
        self._checkpos_keys(inkeys + outkeys)
        Kxxs = self._assemblecovblocks(inkeys, outkeys)
        ycov = numpy.block(ycovblocks) if ycovblocks is not None else None
        self._check_ycov(ycov)
        ymean = y
        self._check_ymean(ymean)
        Kxsxs = self._assemblecovblocks(outkeys)
        if fromdata:
            solver = self._solver(inkeys, ycov)
        else:
            solver = self._solver(inkeys)
        mean = solver.pinv_bilinear(Kxxs, ymean)
        cov = Kxsxs - solver.ginv_quad(Kxxs)
        if not fromdata and ycov is not None:
            A = solver.ginv_linear(Kxxs)
            cov = cov + A.T @ ycov @ A
        if not strip:
            meandict = {
                key: mean[slic].reshape(self._elements[key].shape)
                for key, slic in zip(outkeys, outslices)
            }
            covdict = {
                (row, col): cov[rowslice, colslice].reshape(self._elements[row].shape + self._elements[col].shape)
                for row, rowslice in zip(outkeys, outslices)
                for col, colslice in zip(outkeys, outslices)
            }
            return meandict, covdict
        outkey, = outkeys
        return mean.reshape(self._elements[outkey].shape), cov.reshape(2 * self._elements[outkey].shape)

    def predfromfit(self, *args, **kw):
        return self.pred(*args, fromdata=False, **kw)

    def predfromdata(self, *args, **kw):
        return self.pred(*args, fromdata=True, **kw)

    def _prior_decomp(self, given, givencov=None, **kw):
        """ ``_compute.py:336-367`` """
        ylist, inkeys, ycovblocks = self._flatgiven(given, givencov)
        y = numpy.concatenate(ylist)
        self._checkpos_keys(inkeys)
        ymean = y
        self._check_ymean(ymean)
        ycov = numpy.block(ycovblocks) if ycovblocks is not None else None
        self._check_ycov(ycov)
        return self._solver(inkeys, ycov, **kw), ymean

    def marginal_likelihood(self, given, givencov=None, **kw):
        """ ``_compute.py:383-422`` """
        decomp, ymean = self._prior_decomp(given, givencov, **kw)
        mll, _, _, _, _ = decomp.minus_log_normal_density(ymean, value=True)
        return -mll

    @classmethod
    def decompose(cls, posdefmatrix, solver='chol', **kw):
        """ ``_compute.py:430-516`` """
        m = numpy.asarray(posdefmatrix)
        assert m.size > 0
        assert m.ndim % 2 == 0
        half = m.ndim // 2
        head = m.shape[:half]
        tail = m.shape[half:]
        assert head == tail
        n = math.prod(head)
        m = m.reshape(n, n)
        return cls._getdecomp(solver)(m, **kw)
