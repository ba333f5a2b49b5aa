"""
Host-side mirror of the raw-output subset of ``lsqfitgp.GP``
(``src/lsqfitgp/_GP/_gp.py:22-129``): the thin part of the façade needed to
drive the device hot path -- ``addx``, ``addcov``, ``addtransf`` (scalar or
tensor coefficients), ``prior(raw=True)``, ``pred/predfromdata/predfromfit(...,
raw=True)``, ``marginal_likelihood``, ``decompose``.  Same names, keyword
arguments, immutability (every ``add*`` returns a new object,
``_GP/_base.py:86-104``), caches (``_covblocks`` ``_elements.py:43,626-640``,
``_decompcache`` ``_compute.py:35,55-58``) and exceptions as the reference.
Covariance blocks, decompositions and all intermediate matrices stay in HBM;
only the requested outputs are downloaded.  Outputs that would be ``gvar``
objects (``raw=False``) are out of scope and raise ``NotImplementedError``.
"""

import copy
import ctypes
import math

import numpy

from . import _lib
from ._device import Context, DeviceArray
from ._linalg import Decomposition, Chol
from . import _kernels
from . import _dual


class _Element:
    @property
    def size(self):
        return math.prod(self.shape)


class _Points(_Element):
    """ Points where the process is evaluated (``_elements.py:104-117``); `deriv`: order of the derivative of the process taken
    there (implicit variable) """
    def __init__(self, x, deriv=0):
        self.x = x
        self.deriv = deriv
    @property
    def shape(self):
        return self.x.shape


class _LinTransf(_Element):
    """ Linear transformation of other elements with fixed coefficient tensors
    (``_elements.py:119-147`` restricted to what `addtransf` builds, ``:332-344``) """
    def __init__(self, tensors, keys, shape, axes, dtensors=None):
        self.tensors = tensors
        self.keys = keys
        self.shape = shape
        self.axes = axes
        self.dtensors = dtensors or {}          # element key -> {tangent key: array like the tensor}  (`_dual`)


class _Cov(_Element):
    """ User-provided covariance matrix block(s) (``_elements.py:149-156``) """
    def __init__(self, blocks, shape, tangents=None):
        self.blocks = blocks
        self.shape = shape
        self.tangents = tangents if tangents is not None else {}      # (row, col) -> {tangent key: matrix like the block}  (`_dual`)


class Diagonal:
    """
    A diagonal covariance block given by its diagonal: ``addcov(diag(v), key)`` / ``givencov={(k, k): diag(v)}`` mean the
    same as the dense ``numpy.diag(v)`` of the reference API, without materialising (and uploading) n^2 zeros on the host:
    the device matrix is built by a memset and one kernel.  Not in the reference (it takes dense arrays only).
    """

    def __init__(self, values):
        values, self.dvalues = _dual.split(values)          # (tangents of the diagonal, when it was computed from hyperparameters)
        self.values = numpy.asarray(values)
        if self.values.ndim != 1:
            raise ValueError('the diagonal must be a vector')
        self.shape = (len(self.values), len(self.values))
        self.ndim = 2
        self.dtype = self.values.dtype
        self.size = len(self.values) ** 2

    @property
    def T(self):
        return self

    def dense(self):
        return numpy.diag(self.values)

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and not isinstance(shape[0], int) else shape
        if tuple(shape) != self.shape:
            raise ValueError(f'a diagonal block of shape {self.shape} can not be reshaped to {tuple(shape)}')
        return self

    def __array__(self, dtype=None, copy=None):
        a = self.dense()
        return a if dtype is None else a.astype(dtype)


try:                                     # a fast non-cryptographic digest where there is one (10-20 GB/s), else the standard library's
    import xxhash as _xxhash
except ImportError:                      # pragma: no cover
    _xxhash = None


_DIGEST_POOL = None


def _digest(arr):
    """ digest of every byte of a host array (strided arrays are read in C order).  Large arrays are hashed in chunks on a few
    threads (xxhash releases the interpreter lock): the 512 MB of a dense N = 8192 block in ~3 ms instead of ~20 """
    a = numpy.ascontiguousarray(arr)
    buf = a.view(numpy.uint8).reshape(-1) if a.dtype.names is None and a.dtype != object else numpy.frombuffer(a.tobytes(), numpy.uint8)
    if _xxhash is None:
        import hashlib
        return hashlib.blake2b(buf, digest_size=16).digest()
    if buf.size < (8 << 20):
        return _xxhash.xxh3_128_digest(buf)
    global _DIGEST_POOL
    if _DIGEST_POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _DIGEST_POOL = ThreadPoolExecutor(max_workers=max(2, min(8, os.cpu_count() or 2)), thread_name_prefix='lgp-digest')
    step = 4 << 20
    parts = _DIGEST_POOL.map(lambda off: _xxhash.xxh3_128_digest(buf[off:off + step]), range(0, buf.size, step))
    return _xxhash.xxh3_128_digest(b''.join(parts))


def diag(values):
    """ ``lsqfitgp_amd.diag(v)``: see `Diagonal` """
    return Diagonal(values)


def _allclose_dev(A, B):
    """ ``jnp.allclose(A, B)`` on device matrices """
    f, F = _lib.fn('lgp_allclose', A.dtype)
    flag = ctypes.c_int()
    _lib.check(f(A.ctx.handle, A.shape[0], A.shape[1], A.ptr, A.ld, B.ptr, B.ld, F(1e-5), F(1e-8), ctypes.byref(flag)), 'lgp_allclose')
    return bool(flag.value)


def _check_dev(A, sym):
    """ (nonfinite, asymmetric) flags of a device matrix """
    f, F = _lib.fn('lgp_check_matrix', A.dtype)
    flags = (ctypes.c_int * 2)()
    _lib.check(f(A.ctx.handle, A.shape[0], A.shape[1], A.ptr, A.ld, int(bool(sym)), F(1e-5), F(1e-8), flags), 'lgp_check_matrix')
    return bool(flags[0]), bool(flags[1])


def _axpby(alpha, A, beta, B, out=None):
    """ out = alpha*A + beta*B on device (B may be None) """
    ref = A if A is not None else B
    if out is None:
        out = DeviceArray.empty(ref.shape, ref.dtype, ref.ctx)
    f, F = _lib.fn('lgp_axpby', ref.dtype)
    _lib.check(f(ref.ctx.handle, ref.shape[0], ref.shape[1], F(alpha), None if A is None else A.ptr, 0 if A is None else A.ld,
                 F(beta), None if B is None else B.ptr, 0 if B is None else B.ld, out.ptr, out.ld), 'lgp_axpby')
    return out


def _gemm(ta, tb, A, B, alpha=1., beta=0., C=None):
    f, F = _lib.fn('lgp_gemm', A.dtype)
    m = A.shape[1] if ta else A.shape[0]
    k = A.shape[0] if ta else A.shape[1]
    n = B.shape[0] if tb else B.shape[1]
    if C is None:
        C = DeviceArray.empty((m, n), A.dtype, A.ctx)
    _lib.check(f(A.ctx.handle, int(ta), int(tb), m, n, k, F(alpha), A.ptr, A.ld, B.ptr, B.ld, F(beta), C.ptr, C.ld), 'lgp_gemm')
    return C


def _astype_dev(A, dtype):
    """ dtype conversion on the device (mixed-precision assembly: float32 kernel blocks next to float64 blocks) """
    dtype = numpy.dtype(dtype)
    if A.dtype == dtype:
        return A
    out = DeviceArray.empty(A.shape, dtype, A.ctx)
    if A.size:
        A2, o2 = A.as2d(), out.as2d()
        name = 'lgp_convert_f32_f64' if dtype == numpy.float64 else 'lgp_convert_f64_f32'
        _lib.check(getattr(_lib.load(), name)(A.ctx.handle, A2.shape[0], A2.shape[1], o2.ptr, o2.ld, A2.ptr, A2.ld), name)
    return out


class _Zero:
    """ A covariance block that is zero by construction: two elements without a common ancestor (``_elements.py:616``,
    ``jnp.zeros`` there, float64).  It is never written to HBM unless somebody needs the matrix itself: linear transformations
    skip it (it still takes part in the result's dtype, as the reference's dense zeros do) and the assembly clears its region. """

    __slots__ = ('shape', 'ctx')
    dtype = numpy.dtype(numpy.float64)

    def __init__(self, shape, ctx):
        self.shape = tuple(int(v) for v in shape)
        self.ctx = ctx

    @property
    def size(self):
        return self.shape[0] * self.shape[1]

    def transpose(self):
        return _Zero(self.shape[::-1], self.ctx)

    def dense(self):
        return DeviceArray.zeros(self.shape, self.dtype, self.ctx)

    def to_host(self):
        return numpy.zeros(self.shape, self.dtype)


def _transposed(block):
    """ the transposed block: nothing to do for zeros and for blocks known to be exactly symmetric (the Gram matrix of a point set
    with itself, an uploaded diagonal, and whatever is merely another name for one of them) """
    if isinstance(block, _Zero):
        return block.transpose()
    if getattr(block, '_symm', False) and block.shape[0] == block.shape[1]:
        return block
    return block.transpose()


class GP:
    """
    Object that represents a Gaussian process (device-resident subset).

    Parameters mirror ``lsqfitgp.GP`` (``_gp.py:114-129``): ``covfun`` is a
    kernel object of this package, ``solver`` must be ``'chol'``, the
    ``check*`` flags and ``posepsfac``/``halfmatrix`` keep their meaning, and
    extra keyword arguments (``epsrel``, ``epsabs``) go to the decomposition.
    ``fuse_solve`` (not in the reference; default True): when `pred` has to factor ``Kxx + ycov`` for this call only (a
    nonzero ``givencov``: such decompositions are never cached, ``_compute.py:55-58``), ``[Kxxs | y]`` is carried through
    the factorisation (`Chol` ``rhs=``) instead of being solved afterwards by a latency-bound chain of small kernels.
    Cached decompositions (no ``givencov``, or one that is exactly zero) always use the separate solve, so that a cache hit
    reproduces the first call bit for bit.
    ``reuse_givencov`` (not in the reference; default True): the reference factors ``Kxx + ycov`` anew in every call that is
    given a ``givencov`` (``_compute.py:55-58``), so ``predfromdata(given, key, givencov)`` followed by
    ``marginal_likelihood(given, givencov)`` -- the usual pair -- factors the same matrix twice.  Here the decomposition of the
    last such call is kept and serves the next call on the same object when the SAME ``givencov`` object is passed again with
    the same keys and decomposition arguments and the same CONTENT: a digest of every byte of its host blocks is compared, so an
    in-place edit of any entry between the two calls is noticed (one pass over the blocks, 10-20 GB/s; the reference's calls make
    several).  The numbers are those of the factorisation that was kept.
    """

    DefaultProcess = 'default'

    def __init__(self, covfun=None, *, solver='chol', checkpos=True, checksym=True,
        checkfinite=True, checklin=True, posepsfac=1, halfmatrix=False, ctx=None, fuse_solve=True, reuse_givencov=True, **kw):
        if covfun is not None and not isinstance(covfun, _kernels.Kernel):
            raise TypeError('covariance function must be a lsqfitgp_amd kernel object')
        self._covfun = covfun
        self._ctx = ctx
        self._fuse_solve = bool(fuse_solve)
        self._reuse_givencov = bool(reuse_givencov)
        self._givencov_cache = None    # (keys, fingerprint of givencov, decomposition kwargs, decomposition) of the last call
        self._elements = {}        # key -> _Element
        self._covblocks = {}       # (key, key) -> DeviceArray (2d flattened)
        self._tanblocks = {}       # (key, key) -> {tangent key: DeviceArray}: derivatives of the blocks (`_dual`)
        self._decompcache = {}     # tuple of keys -> Decomposition
        self._checkpositive = bool(checkpos)
        self._posepsfac = float(posepsfac)
        self._checksym = bool(checksym)
        self._checkfinite = bool(checkfinite)
        self._checklin = bool(checklin)
        self._halfmatrix = bool(halfmatrix)
        assert not (halfmatrix and checksym)
        self._dtype = None
        decomp = self._getdecomp(solver)
        self._decomptype = decomp
        self._decompkw = dict(kw)
        self._decompclass = lambda K, **kwargs: decomp(K, **kwargs, **kw)
        self._checkpos_cache = []

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = Context.default()
        return self._ctx

    def _clone(self):
        """ ``_GP/_base.py:86-104`` clone-on-write """
        # deferred block checks are settled first: the flag words and the list of pending keys belong to ONE object
        self._resolve_checks()
        new = copy.copy(self)
        new._checkflags = None
        new._checkkeys = []
        new._elements = self._elements.copy()
        new._covblocks = self._covblocks.copy()
        new._tanblocks = self._tanblocks.copy()
        new._decompcache = self._decompcache.copy()
        new._checkpos_cache = list(self._checkpos_cache)
        return new

    # ---- elements (``_GP/_elements.py``) -------------------------------------

    @staticmethod
    def _keyed(arg, key, what):
        """ the ``(arg, key)`` calling convention shared by `addx` and `addcov`: either a dictionary and no key, or one
        object and its key.  Returns a dictionary, with the reference's error messages (``_elements.py:199-212,463-476``) """
        if hasattr(arg, 'keys'):
            if key is not None:
                raise ValueError(f'can not specify key if {what} is a dictionary')
            if None in arg:
                raise ValueError(f'None key in {what} not allowed')
            return arg, True
        if key is None:
            raise ValueError(f'{what} is not dictionary but key is None')
        return None, False

    def _require_new(self, key):
        if key in self._elements:
            raise KeyError(f'key {key!r} already in GP')

    @staticmethod
    def _field_orders(spec):
        """ a derivative specification as the reference's ``Deriv`` reads it (``_Deriv.py:24-97``) -> 0, a positive integer (the
        implicit variable) or a dictionary field name -> order: an integer; a field name; a sequence of names, an integer in
        front of a name multiplying it (``['a', 2, 'b']``: once along a, twice along b) """
        if spec is None:
            return 0
        if isinstance(spec, bool):
            raise TypeError('argument must be None, int, str, or iterable')
        if isinstance(spec, (int, numpy.integer)):
            if spec < 0:
                raise ValueError(f'degree {spec} is negative')
            return int(spec)
        if isinstance(spec, str):
            return {spec: 1}
        if not numpy.iterable(spec):
            raise TypeError('argument must be None, int, str, or iterable')
        out, times = {}, None
        for obj in spec:
            if isinstance(obj, str):
                n = 1 if times is None else times
                if n:
                    out[obj] = out.get(obj, 0) + n
                times = None
            elif isinstance(obj, (int, numpy.integer)) and not isinstance(obj, bool):
                if times is not None:
                    raise ValueError('consecutive integers in iterable')
                if obj < 0:
                    raise ValueError(f'degree {obj} is negative')
                times = int(obj)
            else:
                raise TypeError('objects in iterable must be int or str')
        if times is not None:
            raise ValueError('dangling derivative order')
        return out or 0

    def addx(self, x, key=None, *, deriv=0, proc=DefaultProcess):
        """ Add points where the Gaussian process is evaluated (``_elements.py:158-242``).  ``deriv``: derivative of the process
        taken at the points, an integer order along the variable of one-dimensional points or, for structured points, along
        named fields (``'f0'``, ``(2, 'f0')``, ``('f0', 'f1')``: what ``Deriv`` of the reference accepts) """
        deriv = self._field_orders(deriv)
        if proc != self.DefaultProcess:
            raise KeyError(f'process named {proc!r} not found')
        points, isdict = self._keyed(x, key, 'x')
        if not isdict:
            points = {key: x}
        new = self._clone()
        for k, value in points.items():
            new._require_new(k)
            arr = numpy.asarray(value)
            # all the point sets of a GP must share a dtype family (plain or the same structure): the kernel is called on
            # pairs of them                                                      (_elements.py:222-233)
            if new._dtype is None:
                new._dtype = arr.dtype
            else:
                try:
                    new._dtype = numpy.result_type(new._dtype, arr.dtype)
                except TypeError:
                    raise TypeError('x[{!r}].dtype = {!r} not compatible with {!r}'.format(k, arr.dtype, new._dtype))
            orders = deriv
            if isinstance(deriv, dict):
                # orders along the leaf fields, in the order in which the kernels read them (`_kernels._leaf_columns`)
                names = arr.dtype.names
                if names is None:
                    raise ValueError(f'derivative along named fields {sorted(deriv)} of points without fields')      # _ops.py:141-148
                missing = [nm for nm in deriv if nm not in names]
                if missing:
                    raise ValueError(f'derivative along missing field {missing[0]!r}')
                if any(arr.dtype[nm].shape or arr.dtype[nm].names for nm in names):
                    raise NotImplementedError('derivatives along fields of points with array-valued or nested fields')
                orders = tuple(deriv.get(nm, 0) for nm in names)
            elif deriv and arr.dtype.names is not None:
                raise ValueError('an integer derivative order needs points without fields: name the field')        # _ops.py:141-148
            new._elements[k] = _Points(arr, orders)
        return new

    def addtransf(self, tensors, key, *, axes=1):
        """ Apply a linear transformation with fixed coefficients to already
        specified process values (``_elements.py:248-346``) """
        if key is None:
            raise ValueError('key can not be None')
        self._require_new(key)
        missing = [k for k in tensors if k not in self._elements]
        if missing:
            raise KeyError(missing[0])
        if len(tensors) == 0:
            raise ValueError('empty tensors, undetermined output shape')
        coeffs, dcoeffs, outshapes = {}, {}, []
        for k, t in tensors.items():
            t, dt = _dual.split(t)
            t = numpy.asarray(t)
            if dt:
                dcoeffs[k] = {key: numpy.broadcast_to(numpy.asarray(v, float), t.shape) for key, v in dt.items()}
            if self._checkfinite and not numpy.all(numpy.isfinite(t)):
                raise ValueError(f'tensors[{k!r}] contains infs/nans')
            eshape = self._elements[k].shape
            if t.ndim == 0:
                outshapes.append(eshape)                       # a scalar multiplies the element as it is
            else:
                if t.shape[t.ndim - axes:] != eshape[:axes]:
                    raise ValueError(f'tensors[{k!r}].shape = {t.shape!r} can not be multiplied with shape {eshape!r} with {axes}-axes contraction')
                outshapes.append(t.shape[:t.ndim - axes] + eshape[axes:])
            coeffs[k] = t
        try:
            shape = numpy.broadcast_shapes(*outshapes)
        except ValueError:
            left = ', '.join(repr(t.shape) for t in coeffs.values())
            right = ', '.join(repr(self._elements[k].shape) for k in coeffs)
            raise ValueError(f'can not broadcast tensors with shapes [{left}] contracted with arrays with shapes [{right}]')
        new = self._clone()
        new._elements[key] = _LinTransf(coeffs, list(coeffs), tuple(shape), axes, dcoeffs)
        return new

    def addlintransf(self, transf, keys, key, *, checklin=None):
        """
        Define a finite linear transformation of the evaluated process by a function (``_elements.py:348-406``):
        ``transf(array1, array2, ...) -> array`` takes one array per key of ``keys`` (with the shapes of those elements) and
        must be linear in them.  The reference traces the function with jax and applies it to the covariance blocks; here it
        is an ordinary function of numpy arrays, evaluated ONCE per basis vector on the host to obtain its coefficient tensors
        (``_LinTransf.matrices``, ``:129-147``), which then act on the device blocks as those of `addtransf` do.
        ``checklin`` (default: the GP's setting): verify on random inputs that the function is linear and not affine
        (``RuntimeError`` otherwise, as the reference).
        """
        if key is None:
            raise ValueError('key can not be None')
        self._require_new(key)
        keys = list(keys)
        for k in keys:
            if k not in self._elements:
                raise KeyError(k)
        if not keys:
            raise ValueError('empty keys, undetermined output shape')
        shapes = [self._elements[k].shape for k in keys]
        zeros = [numpy.zeros(sh) for sh in shapes]
        out0 = numpy.asarray(transf(*zeros), float)
        shape = out0.shape
        if checklin is None:
            checklin = self._checklin
        if checklin:
            rng = numpy.random.default_rng(202310)
            a = [rng.standard_normal(sh) for sh in shapes]
            b = [rng.standard_normal(sh) for sh in shapes]
            fa, fb = numpy.asarray(transf(*a), float), numpy.asarray(transf(*b), float)
            fab = numpy.asarray(transf(*[2. * u - 3. * v for u, v in zip(a, b)]), float)
            scale = max(float(numpy.max(numpy.abs(fa), initial=0.)), float(numpy.max(numpy.abs(fb), initial=0.)), 1.)
            if numpy.any(out0 != 0) or not numpy.allclose(fab, 2. * fa - 3. * fb, rtol=1e-9, atol=1e-9 * scale):
                raise RuntimeError('the transformation is not linear')                 # _elements.py:386-388
        tensors, axes = {}, {}
        for pos, (k, sh) in enumerate(zip(keys, shapes)):
            size = math.prod(sh)
            cols = numpy.empty((math.prod(shape), size))
            args = [z.copy() for z in zeros]
            flat = args[pos].reshape(-1)
            for e in range(size):
                flat[e] = 1.
                cols[:, e] = numpy.asarray(transf(*args), float).reshape(-1)
                flat[e] = 0.
            if self._checkfinite and not numpy.all(numpy.isfinite(cols)):
                raise ValueError(f'the transformation of {k!r} contains infs/nans')
            if k in tensors:                     # the same element passed twice: its coefficients add up
                tensors[k] = tensors[k] + cols.reshape(shape + sh)
            else:
                tensors[k] = cols.reshape(shape + sh)
            axes[k] = len(sh)
        new = self._clone()
        new._elements[key] = _LinTransf(tensors, list(tensors), tuple(shape), axes)
        return new

    def addcov(self, covblocks, key=None, *, decomps=None):
        """ Add user-defined prior covariance matrix blocks (``_elements.py:408-552``) """
        given, isdict = self._keyed(covblocks, key, 'covblocks')
        if isdict:
            if decomps is not None and not hasattr(decomps, 'keys'):
                raise TypeError('covblocks is dictionary but decomps is not')
            decomps = dict(decomps or {})
        else:
            given = {(key, key): covblocks}
            decomps = {} if decomps is None else {key: decomps}

        # pass 1: the diagonal blocks fix the shape of every new element
        shapes, raw, rawtan = {}, {}, {}
        for pair, block in given.items():
            for k in pair:
                self._require_new(k)
            if block is None:
                raise TypeError(f'block {pair!r} is None')
            block, dblock = _dual.split(block)
            if isinstance(block, Diagonal) and block.dvalues:
                dblock = {key: Diagonal(numpy.broadcast_to(numpy.asarray(v, float), block.values.shape)) for key, v in block.dvalues.items()}
            if dblock:
                rawtan[pair] = dblock
            if not isinstance(block, Diagonal):
                block = numpy.asarray(block)
            row, col = pair
            if row == col:
                if block.ndim % 2:
                    raise ValueError(f'diagonal block {row!r} has odd number of axes')
                head, tail = block.shape[:block.ndim // 2], block.shape[block.ndim // 2:]
                if head != tail:
                    raise ValueError(f'shape {block.shape!r} of diagonal block {row!r} is not symmetric')
                if self._checksym and not isinstance(block, Diagonal) and not numpy.allclose(block, block.T):
                    raise ValueError(f'diagonal block {row!r} is not symmetric')
                shapes[row] = head
            raw[pair] = block

        for k, dec in decomps.items():
            if k not in shapes:
                raise KeyError(f'key {k!r} in decomps not found in diagonal blocks')
            if not isinstance(dec, Decomposition):
                raise TypeError(f'decomps[{k!r}] = {dec!r} is not a decomposition')
            size = math.prod(shapes[k])
            if dec.n != size:
                raise ValueError(f'decomposition matrix size {dec.n} != diagonal block size {size} for key {k!r}')

        # pass 2: flatten every block to a matrix; an off-diagonal block given once also serves as its transpose
        flat = {}
        for pair, block in raw.items():
            if self._checkfinite and not numpy.all(numpy.isfinite(block.values if isinstance(block, Diagonal) else block)):
                raise ValueError(f'block {pair!r} not finite')
            row, col = pair
            if row != col:
                for k in pair:
                    if k not in shapes:
                        raise KeyError(f'key {k!r} from off-diagonal block {pair!r} not found in diagonal blocks')
                expected = shapes[row] + shapes[col]
                if block.shape != expected:
                    raise ValueError(f'shape {block.shape!r} of block {pair!r} is not {expected!r} as expected from diagonal blocks')
            flat[pair] = block.reshape(math.prod(shapes[row]), math.prod(shapes[col]))
        for (row, col), mat in list(flat.items()):
            if row != col and raw.get((col, row)) is None:
                flat[col, row] = mat.T
        if self._checksym:
            for (row, col), mat in flat.items():
                if row != col and not numpy.allclose(mat.T, flat[col, row]):
                    raise ValueError(f'block {(row, col)!r} is not the transpose of block {(col, row)!r}')

        # tangents of the blocks (`_dual`), flattened like the blocks
        flattan = {}
        for (row, col), d in rawtan.items():
            rs, cs = math.prod(shapes[row]), math.prod(shapes[col])
            flattan[row, col] = {key: (v if isinstance(v, Diagonal) else numpy.broadcast_to(numpy.asarray(v, float), raw[row, col].shape).reshape(rs, cs))
                                 for key, v in d.items()}
            if row != col and (col, row) not in rawtan:
                flattan[col, row] = {key: v.T for key, v in flattan[row, col].items()}
        new = self._clone()
        for k, shape in shapes.items():
            new._elements[k] = _Cov(flat, shape, flattan)
            if decomps.get(k) is not None:
                new._decompcache[k,] = decomps[k]
        return new

    # ---- covariance blocks -----------------------------------------------------

    def _upload_block(self, block):
        if isinstance(block, Diagonal):
            dtype = numpy.float32 if block.dtype == numpy.float32 else numpy.float64
            out = DeviceArray.zeros(block.shape, dtype, self.ctx)
            if block.shape[0]:
                vec = DeviceArray.from_host(block.values, dtype=dtype, ctx=self.ctx)
                f, F = _lib.fn('lgp_add_diag', dtype)
                _lib.check(f(self.ctx.handle, block.shape[0], out.ptr, out.ld, vec.ptr, F(1)), 'lgp_add_diag')
                out._diagvec = vec          # (a sum that this block joins adds the vector to its diagonal instead of reading the matrix)
            out._symm = True
            return out
        block = numpy.asarray(block)
        dtype = numpy.float32 if block.dtype == numpy.float32 else numpy.float64
        return DeviceArray.from_host(block, dtype=dtype, ctx=self.ctx)

    def _makecovblock_points(self, xkey, ykey):
        """ ``_elements.py:554-579``: HOT LOOP 1, one Gram-kernel launch """
        x = self._elements[xkey]
        y = self._elements[ykey]
        kernel = self._covfun
        if kernel is None:
            raise ValueError('no covariance function: GP(covfun) is required to add points')
        xf = x.x.reshape(-1)
        def orders(e, other):          # per-field orders on both sides when either names its fields
            if isinstance(e.deriv, tuple) or not isinstance(other.deriv, tuple):
                return e.deriv
            return (0,) * len(other.deriv)
        kw = {'deriv': (orders(x, y), orders(y, x))} if (x.deriv or y.deriv) else {}       # kernel.linop('diff', x.deriv, y.deriv), :565
        if x is y and not self._checksym and self._halfmatrix:
            # the reference evaluates the upper triangle as a flat vector and gathers (:567-573);
            # the device Gram kernels are exactly symmetric, so the full tile evaluation gives the same matrix
            out = kernel.gram(xf, xf, ctx=self.ctx, **kw)
        else:
            yf = xf if y is x else y.x.reshape(-1)
            out = kernel.gram(xf, yf, ctx=self.ctx, **kw)
        if x is y and not kw:
            out._symm = True              # k(x_i, x_j) and k(x_j, x_i) are the same arithmetic: its transpose is itself
        return out

    def _makecovblock_lintransf_any(self, xkey, ykey):
        """ ``_elements.py:581-601`` for the tensordot transformation of `addtransf` (``:332-344``) """
        x = self._elements[xkey]
        y = self._elements[ykey]
        # the terms first: blocks that are zero by construction are skipped (not read, not even allocated) but count for the
        # dtype, as the reference's dense float64 zeros do
        terms, dtypes = [], []
        for k in x.keys:
            cov = self._covblock(k, ykey)
            assert cov.shape == (self._elements[k].size, y.size)
            dtypes.append(cov.dtype)
            if not isinstance(cov, _Zero):
                terms.append((k, cov))
        if not terms:
            return _Zero((x.size, y.size), self.ctx)
        dt = numpy.result_type(*dtypes)
        if len(terms) == 1:
            k, cov = terms[0]
            t = x.tensors[k]
            if not t.shape and self._elements[k].size == x.size and float(t) == 1. and cov.dtype == dt:
                return cov                                # the block under another name: nothing to compute
        out = None
        for k, cov in terms:
            elem = self._elements[k]
            t = x.tensors[k]
            if t.shape:
                # tensordot(t, cov.reshape(elem.shape + (ysize,)), axes): (out.., contracted..) x (contracted.., rest.., ysize)
                axes = x.axes[k] if isinstance(x.axes, dict) else x.axes
                csize = math.prod(elem.shape[:axes])
                rest = math.prod(elem.shape[axes:])
                osize = math.prod(t.shape[:t.ndim - axes])
                if (osize * rest) != x.size:
                    raise NotImplementedError('broadcasting between transformed elements of different shapes')
                cov = _astype_dev(cov, dt)
                td = DeviceArray.from_host(t.reshape(osize, csize), dtype=dt, ctx=self.ctx)
                covr = cov if cov.contiguous else cov.copy()
                covr = covr.reshape_contiguous((csize, rest * y.size))
                if out is None:
                    out = _gemm(False, False, td, covr).reshape_contiguous((x.size, y.size))
                else:
                    _gemm(False, False, td, covr, 1., 1., out.reshape_contiguous((osize, rest * y.size)))
            elif elem.size == x.size:
                vec = getattr(cov, '_diagvec', None)
                if out is None:
                    conv = _astype_dev(cov, dt)
                    if conv is not cov:                   # converted: a matrix of our own already
                        out = conv if float(t) == 1. else _axpby(float(t), conv, 0., None, conv)
                    else:
                        out = _axpby(float(t), cov, 0., None)
                    if len(terms) == 1:
                        # the block in another precision / scale: what is known of it still holds (symmetric; a diagonal;
                        # finite, if nothing was multiplied in)
                        out._symm = getattr(cov, '_symm', False)
                        if vec is not None:
                            out._diagvec = _axpby(float(t), _astype_dev(vec, dt).as2d(), 0., None)
                        if float(t) == 1.:
                            out._checks_passed = getattr(cov, '_checks_passed', 0) & 1
                elif vec is not None and out.shape[0] == out.shape[1]:
                    f, F = _lib.fn('lgp_add_diag', dt)        # a diagonal block joins the sum through its diagonal alone
                    _lib.check(f(self.ctx.handle, out.shape[0], out.ptr, out.ld, _astype_dev(vec, dt).ptr, F(float(t))), 'lgp_add_diag')
                else:
                    _axpby(float(t), _astype_dev(cov, dt), 1., out, out)
            elif elem.size == 1:
                # a scalar element broadcast over the output (e.g. the marginalised mean of
                # bayestree/_bart.py:203): out += t * ones (x.size, 1) @ cov (1, ysize)
                ones = DeviceArray.from_host(numpy.ones((x.size, 1)), dtype=dt, ctx=self.ctx)
                cov = _astype_dev(cov, dt)
                if out is None:
                    out = _gemm(False, False, ones, cov, float(t), 0.)
                else:
                    _gemm(False, False, ones, cov, float(t), 1., out)
            else:
                raise NotImplementedError('broadcasting between transformed elements of different shapes')
        return out

    def _makecovblock(self, xkey, ykey):
        """ ``_elements.py:603-624`` """
        x = self._elements[xkey]
        y = self._elements[ykey]
        if isinstance(x, _Points) and isinstance(y, _Points):
            cov = self._makecovblock_points(xkey, ykey)
        elif isinstance(x, _LinTransf):
            cov = self._makecovblock_lintransf_any(xkey, ykey)
        elif isinstance(y, _LinTransf):
            cov = _transposed(self._makecovblock_lintransf_any(ykey, xkey))
        elif isinstance(x, _Cov) and isinstance(y, _Cov) and x.blocks is y.blocks and (xkey, ykey) in x.blocks:
            cov = self._upload_block(x.blocks[xkey, ykey])
        else:
            cov = _Zero((x.size, y.size), self.ctx)
        if (self._checkfinite or (self._checksym and xkey == ykey)) and cov.size and not isinstance(cov, _Zero):
            self._enqueue_check(cov, (xkey, ykey))
        return cov

    # The finite / symmetric verdicts of the blocks (``_elements.py:618-622``) are collected on the device and read once per
    # assembled matrix instead of once per block: every read-back is a device synchronisation (0.12 ms each, 15 per step of
    # the BART workload).  The error is the reference's, raised before the assembled matrix is used.
    _CHECK_SLOTS = 64

    def _enqueue_check(self, cov, key):
        if getattr(self, '_checkflags', None) is None:
            self._checkflags = DeviceArray.zeros((2 * self._CHECK_SLOTS,), numpy.int32, self.ctx)
            self._checkkeys = []
        if len(self._checkkeys) == self._CHECK_SLOTS:
            self._resolve_checks()
        sym = bool(self._checksym and key[0] == key[1])
        need = (1 if self._checkfinite else 0) | (2 if sym else 0)
        if getattr(cov, '_checks_passed', 0) & need == need:
            return                                        # the same matrix under another key
        cov._checks_passed = getattr(cov, '_checks_passed', 0) | need      # (or about to be judged: a failure clears it, below)
        slot = len(self._checkkeys)
        f, F = _lib.fn('lgp_check_matrix_dev', cov.dtype)
        _lib.check(f(cov.ctx.handle, cov.shape[0], cov.shape[1], cov.ptr, cov.ld, int(sym), F(1e-5), F(1e-8), self._checkflags.ptr + 8 * slot),
                   'lgp_check_matrix_dev')
        self._checkkeys.append((key, sym, cov))

    def _resolve_checks(self):
        keys = getattr(self, '_checkkeys', None)
        if not keys:
            return
        flags = self._checkflags.to_host()
        self._checkkeys = []
        if flags.any():
            self._checkflags.fill_zero()
            bad = [(key, sym, slot, cov) for slot, (key, sym, cov) in enumerate(keys)
                   if (self._checkfinite and flags[2 * slot]) or (sym and flags[2 * slot + 1])]
            for key, _, _, cov in bad:                  # a block that failed is not kept (the reference raises before it caches),
                cov._checks_passed = 0                  # under any of its names
                for k in [k for k, v in self._covblocks.items() if v is cov or k == key or k == key[::-1]]:
                    self._covblocks.pop(k, None)
            for key, sym, slot, _ in bad:
                if self._checkfinite and flags[2 * slot]:
                    raise RuntimeError(f'covariance block {key!r} is not finite')
                raise RuntimeError(f'covariance block {key!r} is not symmetric')

    def _covblock(self, row, col):
        """ ``_elements.py:626-640``: block cache, transposed block stored too """
        if (row, col) not in self._covblocks:
            block = self._makecovblock(row, col)
            if row != col:
                blockT = None
                if self._checksym:
                    blockT = self._makecovblock(col, row)
                    bt = _transposed(block)
                    if isinstance(bt, _Zero) != isinstance(blockT, _Zero):
                        bt, blockT = (b.dense() if isinstance(b, _Zero) else b for b in (bt, blockT))
                    if not isinstance(bt, _Zero) and (bt.dtype != blockT.dtype or not _allclose_dev(bt, blockT)):
                        if bt.dtype == blockT.dtype:
                            raise RuntimeError('covariance block {!r} is not symmetric'.format((row, col)))
                    blockT = bt
                self._covblocks[col, row] = blockT if blockT is not None else _transposed(block)
            self._covblocks[row, col] = block
        return self._covblocks[row, col]

    # ---- first-order tangents of the blocks with respect to hyperparameters (`_dual`): what jax.jacfwd of the reference's
    # gpfactory yields for the prior covariance (``_fit.py:598-616``), block by block through the same structure as the values

    def _apply_tensor(self, x, k, t, cov, out):
        """ out (+)= t applied to cov(k, .) as element k's coefficient in the transformation x (the operations of
        `_makecovblock_lintransf_any`, one term) """
        elem = self._elements[k]
        t = numpy.asarray(t)
        ysize = cov.shape[1]
        dt = cov.dtype if out is None else numpy.result_type(cov.dtype, out.dtype)
        cov = _astype_dev(cov, dt)
        if out is not None and out.dtype != dt:
            out = _astype_dev(out, dt)
        if t.shape:
            axes = x.axes[k] if isinstance(x.axes, dict) else x.axes
            csize = math.prod(elem.shape[:axes])
            rest = math.prod(elem.shape[axes:])
            osize = math.prod(t.shape[:t.ndim - axes])
            if (osize * rest) != x.size:
                raise NotImplementedError('broadcasting between transformed elements of different shapes')
            td = DeviceArray.from_host(t.reshape(osize, csize), dtype=dt, ctx=self.ctx)
            covr = (cov if cov.contiguous else cov.copy()).reshape_contiguous((csize, rest * ysize))
            if out is None:
                return _gemm(False, False, td, covr).reshape_contiguous((x.size, ysize))
            _gemm(False, False, td, covr, 1., 1., out.reshape_contiguous((osize, rest * ysize)))
            return out
        if elem.size == x.size:
            return _axpby(float(t), cov, 0., None) if out is None else _axpby(float(t), cov, 1., out, out)
        if elem.size == 1:
            ones = DeviceArray.from_host(numpy.ones((x.size, 1)), dtype=dt, ctx=self.ctx)
            return _gemm(False, False, ones, cov, float(t), 0.) if out is None else _gemm(False, False, ones, cov, float(t), 1., out)
        raise NotImplementedError('broadcasting between transformed elements of different shapes')

    def _tangents_lintransf(self, xkey, ykey):
        x = self._elements[xkey]
        out = {}
        for k in x.keys:
            t = x.tensors[k]
            for key, dcov in self._tangents_of(k, ykey).items():              # t d(cov)
                out[key] = self._apply_tensor(x, k, t, dcov, out.get(key))
            dts = x.dtensors.get(k)
            if dts:                                                           # d(t) cov
                cov = self._covblock(k, ykey)
                if not isinstance(cov, _Zero):
                    for key, dt in dts.items():
                        out[key] = self._apply_tensor(x, k, dt, cov, out.get(key))
        return out

    def _maketangents(self, xkey, ykey):
        x, y = self._elements[xkey], self._elements[ykey]
        if isinstance(x, _Points) and isinstance(y, _Points):
            kernel = self._covfun
            if kernel is None or not kernel.tangent_keys():
                return {}
            xf = x.x.reshape(-1)
            yf = xf if y is x else y.x.reshape(-1)
            kw = {}
            if x.deriv or y.deriv:            # (as in `_makecovblock`: per-field orders on both sides when either names its fields)
                orders = lambda e, other: e.deriv if (isinstance(e.deriv, tuple) or not isinstance(other.deriv, tuple)) else (0,) * len(other.deriv)
                kw = {'deriv': (orders(x, y), orders(y, x))}
            _, tans = kernel.gram_tangents(xf, yf, ctx=self.ctx, **kw)
            return tans
        if isinstance(x, _LinTransf):
            return self._tangents_lintransf(xkey, ykey)
        if isinstance(y, _LinTransf):
            return {key: T.transpose() for key, T in self._tangents_lintransf(ykey, xkey).items()}
        if isinstance(x, _Cov) and isinstance(y, _Cov) and x.blocks is y.blocks and (xkey, ykey) in x.tangents:
            return {key: self._upload_block(v) for key, v in x.tangents[xkey, ykey].items()}
        return {}

    def _tangents_of(self, row, col):
        """ ``{tangent key: d cov(row, col)}`` as device matrices; keys the block does not depend on are absent """
        if (row, col) not in self._tanblocks:
            tans = self._maketangents(row, col)
            self._tanblocks[row, col] = tans
            if row != col:
                self._tanblocks[col, row] = {key: T.transpose() for key, T in tans.items()}
        return self._tanblocks[row, col]

    def _assembletangents(self, keys, tkeys, dtype):
        """ d(_assemblecovblocks(keys)) / d(tangent key), one matrix per entry of ``tkeys`` """
        sizes = [self._elements[k].size for k in keys]
        n = sum(sizes)
        mats = []
        for tk in tkeys:
            M = None
            r0 = 0
            for row, rs in zip(keys, sizes):
                c0 = 0
                for col, cs in zip(keys, sizes):
                    T = self._tangents_of(row, col).get(tk)
                    if T is not None and T.size:
                        if len(keys) == 1:
                            M = _astype_dev(T, dtype)
                        else:
                            if M is None:
                                M = DeviceArray.zeros((n, n), dtype, self.ctx)
                            M.block(r0, r0 + rs, c0, c0 + cs).assign(_astype_dev(T, dtype))
                    c0 += cs
                r0 += rs
            mats.append(M)
        return mats

    def _prior_decomp_tangents(self, given, givencov=None, tkeys=(), **kw):
        """
        `_prior_decomp` plus the first-order tangents of what the likelihood is made of: returns
        ``(decomp, r, dK, dr)`` with ``dK`` a list of device matrices (one per tangent key in ``tkeys``, the derivative of the
        decomposed matrix) and ``dr`` an (n, len(tkeys)) host matrix (the derivative of the residuals) or None -- the inputs of
        `Chol.normal_density_derivs`.  Hyperparameters enter through the kernel (`Kernel.gram_tangents`), through `addcov` blocks
        and `addtransf` coefficients computed from `_dual.Dual` values, through ``givencov`` blocks and through ``given``.
        """
        tkeys = list(tkeys)
        plain, dgiven = {}, {}
        for k, v in given.items():
            plain[k], dgiven[k] = _dual.split(v)
        gc_plain, dgc = givencov, {}
        if givencov is not None and hasattr(givencov, 'keys'):
            gc_plain = {}
            for pair, block in givencov.items():
                b, db = _dual.split(block)
                if isinstance(b, Diagonal) and b.dvalues:
                    db = {key: Diagonal(numpy.broadcast_to(numpy.asarray(v, float), b.values.shape)) for key, v in b.dvalues.items()}
                gc_plain[pair] = b
                if db:
                    dgc[pair] = db
        decomp, ymean = self._prior_decomp(plain, gc_plain, **kw)
        inkeys = list(plain)
        dtype = decomp._dtype
        mats = self._assembletangents(inkeys, tkeys, dtype)
        sizes = [self._elements[k].size for k in inkeys]
        n = sum(sizes)
        offs = dict(zip(inkeys, numpy.concatenate(([0], numpy.cumsum(sizes)))[:-1]))
        owned = set()
        for (row, col), db in dgc.items():
            for key, v in db.items():
                if key not in tkeys:
                    continue
                i = tkeys.index(key)
                if mats[i] is None:
                    mats[i] = DeviceArray.zeros((n, n), dtype, self.ctx)
                    owned.add(i)
                elif i not in owned:                    # (possibly a cached tangent block itself: never written to)
                    mats[i] = mats[i].copy()
                    owned.add(i)
                rs, cs = self._elements[row].size, self._elements[col].size
                blk = _astype_dev(self._upload_block(v if isinstance(v, Diagonal) else numpy.broadcast_to(numpy.asarray(v, float), numpy.shape(gc_plain[row, col])).reshape(rs, cs)), dtype)
                dst = mats[i].block(int(offs[row]), int(offs[row]) + rs, int(offs[col]), int(offs[col]) + cs)
                _axpby(1., blk, 1., dst, dst)
        for i, M in enumerate(mats):
            if M is None:
                mats[i] = DeviceArray.zeros((n, n), dtype, self.ctx)
        dr = None
        if any(dgiven.values()):
            dr = numpy.zeros((n, len(tkeys)))
            for k in inkeys:
                for key, v in dgiven[k].items():
                    if key in tkeys:
                        shape = numpy.shape(plain[k])
                        dr[int(offs[k]):int(offs[k]) + self._elements[k].size, tkeys.index(key)] = numpy.broadcast_to(numpy.asarray(v, float), shape).reshape(-1)
        return decomp, ymean, mats, dr

    def _assemblecovblocks(self, rowkeys, colkeys=None):
        """ ``_elements.py:642-649`` (jnp.block): strided device copies into one matrix """
        if colkeys is None:
            colkeys = rowkeys
        blocks = [[self._covblock(row, col) for col in colkeys] for row in rowkeys]
        self._resolve_checks()
        if len(blocks) == 1 and len(blocks[0]) == 1:
            b = blocks[0][0]
            return b.dense() if isinstance(b, _Zero) else b
        dtype = numpy.result_type(*[b.dtype for r in blocks for b in r])
        nr = sum(r[0].shape[0] for r in blocks)
        nc = sum(b.shape[1] for b in blocks[0])
        anyzero = any(isinstance(b, _Zero) and b.size for r in blocks for b in r)
        out = (DeviceArray.zeros if anyzero else DeviceArray.empty)((nr, nc), dtype, self.ctx)      # (one clear for all the zero blocks)
        r0 = 0
        for r in blocks:
            c0 = 0
            for b in r:
                if b.size and not isinstance(b, _Zero):
                    out.block(r0, r0 + b.shape[0], c0, c0 + b.shape[1]).assign(_astype_dev(b, dtype))
                c0 += b.shape[1]
            r0 += r[0].shape[0]
        return out

    def _extreme_eigenvalues(self, cov, steps=24):
        """ estimates of the smallest and largest eigenvalue of a symmetric device matrix.  The reference runs two LOBPCG
        iterations from one random vector on the host copy (``_elements.py:651-668``); here: exact for small matrices,
        otherwise `steps` Lanczos iterations entirely on the device (``lgp_lanczos``: two launches per step, no host round
        trip; the host only diagonalises the steps x steps tridiagonal matrix).  Like LOBPCG's, the Ritz values bracket from inside:
        the estimate of the minimum is an upper bound of it. """
        n = cov.shape[0]
        if n <= 256:
            w = numpy.linalg.eigvalsh(cov.to_host().astype(float))
            return w[0], w[-1]
        rng = numpy.random.default_rng(0x5eed)
        q = rng.standard_normal(n)
        q /= numpy.linalg.norm(q)
        steps = min(steps, n)
        alpha, beta = numpy.zeros(steps), numpy.zeros(steps)
        f, _ = _lib.fn('lgp_lanczos', cov.dtype)
        _lib.check(f(self.ctx.handle, n, cov.ptr, cov.ld, steps, q.ctypes.data, alpha.ctypes.data, beta.ctypes.data), 'lgp_lanczos')
        stop = numpy.flatnonzero(beta == 0)                # an invariant subspace: the recurrence ended there
        if len(stop):
            alpha, beta = alpha[:stop[0] + 1], beta[:stop[0] + 1]
        k = len(alpha)
        T = numpy.diag(alpha) + numpy.diag(beta[:k - 1], 1) + numpy.diag(beta[:k - 1], -1)
        w = numpy.linalg.eigvalsh(T)
        return w[0], w[-1]

    def _checkpos(self, cov):
        """ ``_elements.py:651-668``: reject ``mineigv < -n eps maxeigv posepsfac`` """
        mineigv, maxeigv = self._extreme_eigenvalues(cov)
        assert mineigv <= maxeigv
        if mineigv < 0:
            bound = -len(cov) * numpy.finfo(cov.dtype).eps * maxeigv * self._posepsfac
            if mineigv < bound:
                raise numpy.linalg.LinAlgError('covariance matrix is not positive definite: mineigv = {:.4g} < {:.4g}'.format(mineigv, bound))

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
        """ Prior covariance (``_elements.py:758-817``); ``raw=True`` only """
        raw = bool(raw)
        if not raw:
            raise NotImplementedError('gvar outputs are out of scope of the device path: use raw=True')
        if key is None:
            outkeys = list(self._elements)
        elif isinstance(key, list):
            outkeys = key
        else:
            outkeys = None
        self._checkpos_keys([key] if outkeys is None else outkeys)
        if outkeys is not None:
            blocks = {(row, col): self._covblock(row, col) for row in outkeys for col in outkeys}
            self._resolve_checks()
            return {rc: b.to_host().reshape(self._elements[rc[0]].shape + self._elements[rc[1]].shape) for rc, b in blocks.items()}
        block = self._covblock(key, key)
        self._resolve_checks()
        return block.to_host().reshape(2 * self._elements[key].shape)

    def _slices(self, keylist):
        """ ``_elements.py:819-826`` """
        sizes = [self._elements[key].size for key in keylist]
        stops = numpy.concatenate(([0], numpy.cumsum(sizes)))        # (numpy.pad costs 10 us)
        return [slice(stops[i - 1], stops[i]) for i in range(1, len(stops))]

    # ---- compute (``_GP/_compute.py``) ---------------------------------------------

    @staticmethod
    def _getdecomp(solver):
        """ ``_compute.py:424-428`` """
        return {'chol': Chol}[solver]

    @staticmethod
    def _fingerprint(givencov):
        """ what a ``givencov`` dictionary holds: per block its pair, shape, dtype and a digest of EVERY byte of a host array (an
        in-place edit of any entry changes it), or address, shape and write count of a device array (`DeviceArray.assign` /
        `fill_zero` bump the count; stores made behind the class's back through the C-ABI are not seen) """
        if givencov is None or not hasattr(givencov, 'items'):
            return None
        parts = []
        for pair, block in givencov.items():
            values = block.values if isinstance(block, Diagonal) else block
            if isinstance(values, DeviceArray):
                parts.append((pair, type(block).__name__, values.shape, values.ptr, values.ld, values.writes))
                continue
            arr = numpy.asarray(values)
            parts.append((pair, type(block).__name__, arr.shape, str(arr.dtype), _digest(arr)))
        return tuple(parts)

    def _solver(self, keys, ycov=None, rhs_builder=None, **kw):
        """ Decomposition of the covariance of ``keys`` plus ``ycov``
        (``_compute.py:45-94``): HOT LOOP 2.  ``rhs_builder(dtype)`` (optional) supplies right-hand sides to carry
        through a factorisation that is not already cached """
        keys = tuple(keys)
        if ycov is None:
            cache = self._decompcache.get(keys)
            if cache is not None:
                return cache
        Kxx = self._assemblecovblocks(keys)
        if ycov is not None:
            if ycov.dtype != Kxx.dtype:
                dt = numpy.result_type(ycov.dtype, Kxx.dtype)
                ycov, Kxx = _astype_dev(ycov, dt), _astype_dev(Kxx, dt)
            Kxx = _axpby(1., Kxx, 1., ycov)                               # Kxx + ycov, :85
        if rhs_builder is not None:
            kw = dict(kw, rhs=rhs_builder(Kxx.dtype))
        decomp = self._decompclass(Kxx, **kw)                              # :88
        if ycov is None:
            self._decompcache[keys] = decomp
        return decomp

    def _flatgiven(self, given, givencov):
        """ ``_compute.py:96-136``: the data as flat vectors in dictionary order, and the matching grid of covariance blocks """
        if not hasattr(given, 'keys'):
            raise TypeError('`given` must be dict')
        opaque = givencov is None or isinstance(givencov, Decomposition)
        if not opaque and not hasattr(givencov, 'keys'):
            raise TypeError('`givenconv` must be None, dict or Decomposition')
        keys, values = [], []
        for k, v in given.items():
            if k not in self._elements:
                raise KeyError(k)
            v = numpy.asarray(v)
            want = self._elements[k].shape
            if v.shape != want:
                raise ValueError('given[{!r}] has shape {!r} different from shape {!r}'.format(k, v.shape, want))
            if v.dtype == object:
                raise NotImplementedError('gvar inputs are out of scope of the device path')
            if not numpy.issubdtype(v.dtype, numpy.number):
                raise TypeError('given[{!r}] has non-numerical dtype {!r}'.format(k, v.dtype))
            keys.append(k)
            values.append(v.reshape(-1))
        if opaque:
            if givencov is not None:
                raise NotImplementedError('Decomposition as givencov (Woodbury) is un-implemented in the reference too')
            return values, keys, None
        if len(keys) == 1 and isinstance(givencov[keys[0], keys[0]], Diagonal):
            d = givencov[keys[0], keys[0]]
            if d.shape[0] != values[0].size:
                raise ValueError(f'diagonal covariance of size {d.shape[0]} for {values[0].size} values')
            return values, keys, d                      # stays a diagonal: built on the device, nothing dense on the host
        grid = [[numpy.asarray(givencov[ki, kj]).reshape(vi.shape + vj.shape) for kj, vj in zip(keys, values)]
                for ki, vi in zip(keys, values)]
        return values, keys, grid

    @staticmethod
    def _join_ycov(blocks):
        """ the covariance of the data as one matrix (``_compute.py:225-227``).  A covariance that is exactly zero is the
        same as none -- the reference pins that identity bit for bit (tests/GP/test_GP.py:652-660) -- and is treated as none,
        so that the cached decomposition of the prior serves it """
        if blocks is None:
            return None
        if isinstance(blocks, Diagonal):
            return None if not numpy.any(blocks.values) else blocks
        ycov = numpy.block(blocks)
        return None if not numpy.any(ycov) else ycov

    def _check_ymean(self, ymean):
        """ ``_compute.py:369-372`` """
        if self._checkfinite and not numpy.all(numpy.isfinite(ymean)):
            raise ValueError('mean of `given` is not finite')

    def _check_ycov(self, ycov):
        """ ``_compute.py:374-381`` (ycov is a host array here: it comes from the caller) """
        if ycov is None:
            return
        if isinstance(ycov, Diagonal):
            if self._checkfinite and not numpy.all(numpy.isfinite(ycov.values)):
                raise ValueError('covariance matrix of `given` is not finite')
            return
        if self._checkfinite and not numpy.all(numpy.isfinite(ycov)):
            raise ValueError('covariance matrix of `given` is not finite')
        if self._checksym and not numpy.allclose(ycov, ycov.T):
            raise ValueError('covariance matrix of `given` is not symmetric')

    def pred(self, given, key=None, givencov=None, *, fromdata=None, raw=False, keepcorr=None):
        """ Compute the posterior (``_compute.py:138-305``), raw outputs """
        if fromdata is None:
            raise ValueError('you must specify if `given` is data or fit result')
        fromdata = bool(fromdata)
        raw = bool(raw)
        if keepcorr is None:
            keepcorr = not raw
        if keepcorr and raw:
            raise ValueError('both keepcorr=True and raw=True')
        if not raw:
            raise NotImplementedError('gvar outputs are out of scope of the device path: use raw=True')

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

        if fromdata and strip and len(inkeys) == 1:
            fast = self._pred_one_call(inkeys[0], outkeys[0], y, ycovblocks, givencov)
            if fast is not None:
                return fast

        self._checkpos_keys(inkeys + outkeys)

        if tuple(inkeys) not in self._decompcache or ycovblocks is not None:
            # the blocks of the matrix to be factored first: their kernels (the N x N Gram matrix) run while the host prepares the
            # small blocks below, which otherwise keep the GPU waiting for launches (0.5 ms of the 5.4 ms BART step)
            for row in inkeys:
                for col in inkeys:
                    self._covblock(row, col)

        Kxxs = self._assemblecovblocks(inkeys, outkeys)                   # :230

        ycov = self._join_ycov(ycovblocks)
        self._check_ycov(ycov)
        ymean = y
        self._check_ymean(ymean)

        Kxsxs = self._assemblecovblocks(outkeys)                          # :252
        kept = self._cached_given(inkeys, givencov, ycov, {}) if fromdata else None
        ycov_dev = None if (ycov is None or kept is not None) else self._upload_block(ycov)
        n, m = Kxxs.shape
        built = {}

        def build_rhs(dtype):
            # W = [Kxxs | y]; one forward solve serves mean and covariance (:259-260)
            W = DeviceArray.empty((n, m + 1), dtype, self.ctx)
            W.block(0, n, 0, m).assign(_astype_dev(Kxxs, dtype))
            W.block(0, n, m, m + 1).assign(DeviceArray.from_host(ymean.reshape(n, 1), dtype=dtype, ctx=self.ctx))
            built['W'] = W
            return W

        # A factorisation that is computed in this call carries W through it and the solve is free (`Chol` ``rhs=``).  The
        # reference pins "zero givencov == no givencov" and "cache hit == first call" bit for bit (tests/GP/test_GP.py:587-596,
        # 652-660): a zero ycov has become None above, and a decomposition found in the cache serves the same question (same
        # output keys, same data) from the solution it carried -- the very numbers of the first call -- and any other question by a
        # separate solve.
        rhs_key = (tuple(outkeys), ymean.dtype.str, ymean.tobytes())
        carry_ok = self._fuse_solve and issubclass(self._decomptype, Chol) and fromdata
        carry = build_rhs if (carry_ok and (ycov_dev is not None or tuple(inkeys) not in self._decompcache)) else None
        if fromdata:
            solver = kept
            if solver is None:
                solver = self._solver(inkeys, ycov_dev, rhs_builder=carry)   # :255
                if ycov_dev is not None:
                    self._keep_given(inkeys, givencov, {}, solver)
        else:
            solver = self._solver(inkeys)

        dtype = solver._dtype
        mean_d = DeviceArray.empty((m,), dtype, self.ctx)
        cov_d = DeviceArray.empty((m, m), dtype, self.ctx)
        Kss = _astype_dev(Kxsxs, dtype)
        carried = 'W' in built and getattr(solver, '_solved_for', None) is built['W']
        if carried:
            solver._solved_key = rhs_key
            solver._solved_for = True                                     # (the right-hand sides themselves need not stay in HBM)
        elif fromdata and getattr(solver, '_solved_key', None) == rhs_key and solver._solved is not None and solver._solved.shape == (m + 1, n):
            carried = True
        f, F = _lib.fn('lgp_posterior_solved' if carried else 'lgp_posterior', dtype)
        # (the density of y under this decomposition comes out of the same solve: a `marginal_likelihood` of the same data that
        # finds the decomposition in the cache need not solve again)
        dens = F() if fromdata else None
        densref = None if dens is None else ctypes.byref(dens)
        if carried:
            Wt = solver._solved                                           # (m + 1, n) = (L^-1 [Kxxs | y])^T
            _lib.check(f(self.ctx.handle, n, m, solver._L.ptr, solver._L.ld, Wt.ptr, Wt.ld,
                         Kss.ptr, Kss.ld, mean_d.ptr, cov_d.ptr, cov_d.ld, densref), 'lgp_posterior_solved')
        else:
            W = build_rhs(dtype)
            _lib.check(f(self.ctx.handle, n, m, solver._L.ptr, solver._L.ld, solver._dblk.ptr, W.ptr, W.ld,
                         Kss.ptr, Kss.ld, mean_d.ptr, cov_d.ptr, cov_d.ld, densref), 'lgp_posterior')
        if dens is not None:
            solver._density_memo = (numpy.array(ymean, copy=True), solver._dtype.type(dens.value))
        if not fromdata and ycov_dev is not None:
            # cov += A' ycov A, A = K^-1 Kxxs = L^-T (L^-1 Kxxs)             (:262-269)
            A = solver._solved.block(0, m, 0, n).transpose() if carried else W.block(0, n, 0, m).copy()
            solver._trsm(A, 1)
            yA = _gemm(False, False, _astype_dev(ycov_dev, dtype), A)
            _gemm(True, False, A, yA, 1., 1., cov_d)
        mean = mean_d.to_host()
        cov = cov_d.to_host()

        if not strip:                                                     # :286-300
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
        outkey, = outkeys                                                 # :301-305
        return mean.reshape(self._elements[outkey].shape), cov.reshape(2 * self._elements[outkey].shape)

    def _pred_one_call(self, inkey, outkey, y, ycovblocks, givencov=None):
        """ The plain question -- one set of points conditioned on, one predicted, one isotropic kernel the fused entry point covers,
        no checks that need the assembled blocks -- in ONE call of the library instead of fifteen (``lgp_predict_isotropic``: the
        cross blocks straight into the rows carried below the factor, the Gram build + regularisation + factorisation of
        ``lgp_gram_chol_factor_rhs`` with Kxx never materialised, the posterior of ``lgp_posterior_solved``, one upload of the points
        and one download of the results).  Same arithmetic as the general path with ``fuse_solve`` (the kernels are the
        same); what it saves is host work and round trips: at N = 1000 the general path spends more time in Python than the GPU in kernels
        (the reference's own benchmark, ``docs/userguide/optim.rst:88-159``).  Returns None when the question is not of this kind. """
        if self._checkpositive or self._checksym or not self._fuse_solve or self._decomptype is not Chol or self._decompkw:
            return None
        x, xs = self._elements.get(inkey), self._elements.get(outkey)
        kernel = self._covfun
        if not (isinstance(x, _Points) and isinstance(xs, _Points)) or x is xs or x.deriv or xs.deriv:
            return None
        if not isinstance(kernel, _kernels._Isotropic) or kernel._kind not in _kernels.FUSED_KINDS or kernel._dim is not None:
            return None
        if (inkey,) in self._decompcache:
            return None
        if not all(v is None or isinstance(v, (int, float, numpy.integer, numpy.floating)) for v in (kernel._scale, kernel._loc)) \
                or any(not isinstance(c, (int, float, numpy.integer, numpy.floating)) for _, c in kernel._outops) or any(kernel._doutops):
            return None                                   # (hyperparameters with tangents: the general path keeps them)
        noise = None
        if ycovblocks is not None:
            if not isinstance(ycovblocks, Diagonal) or ycovblocks.dvalues:
                return None                               # (a dense covariance of the data has to be uploaded and added: general path)
            if numpy.any(ycovblocks.values):
                noise = numpy.asarray(ycovblocks.values, float)
            if noise is not None and self._cached_given([inkey], givencov, True, {}) is not None:
                return None                               # (factored by an earlier call with this very givencov: the general path reuses it)
        xa, xsa = x.x.reshape(-1), xs.x.reshape(-1)
        n, m = xa.size, xsa.size
        if n == 0 or m == 0 or n > 16384:
            return None                                   # (large problems: the host work does not matter, and the blocks are kept for reuse)
        cols = _kernels._leaf_columns(numpy.asarray(kernel._select(xa)))
        colss = _kernels._leaf_columns(numpy.asarray(kernel._select(xsa)))
        if len(cols) != len(colss):
            return None
        dtype = _kernels._float_dtype_of(cols + colss, kernel._scale, kernel._loc)
        if numpy.dtype(dtype) != numpy.float64 or y.dtype.kind not in 'fiu':
            return None
        d = len(cols)
        pts = numpy.empty((d, n + m))
        for i in range(d):
            pts[i, :n] = cols[i]
            pts[i, n:] = colss[i]
        if self._checkfinite:
            # (finite points give finite values of these kernels; the blocks themselves are never assembled here)
            if not numpy.all(numpy.isfinite(pts)):
                return None
            self._check_ymean(y)
            if noise is not None and not numpy.all(numpy.isfinite(noise)):
                return None
        ctx = self.ctx
        lib = _lib.load()
        F = ctypes.c_double
        # ONE call: the points, y and the noise go up, Kxxs^T lands in the rows carried below the factor, the fused Gram build +
        # factorisation and the posterior run, covariance + mean + density come down together (lgp_predict_isotropic)
        P = DeviceArray.empty((d, n + m), numpy.float64, ctx)
        nx = -(-(m + 1) // 128) * 128                                     # carried rows, padded to whole tiles
        L = DeviceArray.empty((n + nx, n), numpy.float64, ctx)
        dblk = DeviceArray.empty((lib.lgp_chol_dblk_size(n),), numpy.float64, ctx)
        work = DeviceArray.empty((m * m + m + 2 + n,), numpy.float64, ctx)
        nops, codes, vals = _kernels._ops_arrays(kernel._outops, F)
        npar, hpar = kernel._par_args(F)
        yh = numpy.ascontiguousarray(y, float).reshape(n)
        nh = None if noise is None else numpy.ascontiguousarray(noise, float).reshape(n)
        hout = numpy.empty(m * m + m + 2)
        eps = F()
        info = _lib.check(lib.lgp_predict_isotropic_f64(ctx.handle, kernel._kind, int(kernel._p), npar, hpar, d, pts.ctypes.data, n, m,
                                                        int(kernel._loc is not None), F(kernel._loc or 0), int(kernel._scale is not None), F(kernel._scale or 1),
                                                        nops, codes, vals, yh.ctypes.data, None if nh is None else nh.ctypes.data, P.ptr, L.ptr, L.ld, nx,
                                                        dblk.ptr, work.ptr, hout.ctypes.data, ctypes.byref(eps)), 'lgp_predict_isotropic')
        if info != 0:
            raise numpy.linalg.LinAlgError('cholesky decomposition not finite, probably matrix not pos def numerically')
        cov = hout[:m * m].reshape(m, m).copy()
        mean = hout[m * m:m * m + m].copy()
        dens = F(hout[m * m + m])
        nz = None if nh is None else work.rows(m * m + m + 2, m * m + m + 2 + n)
        # the factorisation stays, as on the general path: a `marginal_likelihood` of the same data (with the same givencov) finds it,
        # with the density of y that came out of this solve; another question solves against it
        def matrix(kernel=kernel, xa=xa, nz=nz):
            K = kernel.gram(xa, xa, ctx=ctx)
            if nz is not None:
                _lib.check(lib.lgp_add_diag_f64(ctx.handle, n, K.ptr, K.ld, nz.ptr, F(0.)), 'lgp_add_diag')
            return K
        solver = Chol._adopt(L.block(0, n, 0, n), dblk, eps.value, solved=L.block(n, n + m + 1, 0, n), matrix=matrix, lower_only=True)
        ymean = numpy.asarray(y)
        solver._solved_key = ((outkey,), ymean.dtype.str, ymean.tobytes())
        solver._density_memo = (numpy.array(ymean, copy=True), numpy.float64(dens.value))
        if noise is None:
            self._decompcache[(inkey,)] = solver
        else:
            self._keep_given([inkey], givencov, {}, solver)
        return mean.reshape(xs.shape), cov.reshape(2 * xs.shape)

    def predfromfit(self, *args, **kw):
        """ Like `pred` with ``fromdata=False`` """
        return self.pred(*args, fromdata=False, **kw)

    def predfromdata(self, *args, **kw):
        """ Like `pred` with ``fromdata=True`` """
        return self.pred(*args, fromdata=True, **kw)

    def _prior_decomp(self, given, givencov=None, **kw):
        """ ``_compute.py:336-367`` """
        ylist, inkeys, ycovblocks = self._flatgiven(given, givencov)
        y = numpy.concatenate(ylist)
        self._checkpos_keys(inkeys)
        ymean = y
        self._check_ymean(ymean)
        ycov = self._join_ycov(ycovblocks)
        self._check_ycov(ycov)
        if ycov is not None:
            decomp = self._cached_given(inkeys, givencov, True, kw)
            if decomp is not None:
                return decomp, ymean
        ycov_dev = None if ycov is None else self._upload_block(ycov)
        decomp = self._solver(inkeys, ycov_dev, **kw)
        if ycov is not None:
            self._keep_given(inkeys, givencov, kw, decomp)
        return decomp, ymean

    def _cached_given(self, keys, givencov, ycov, kw):
        """ the decomposition kept by the last call with a ``givencov``, if this call is about the same matrix (see the class
        docstring, ``reuse_givencov``) """
        c = self._givencov_cache
        if not self._reuse_givencov or c is None or ycov is None:
            return None
        # (the dictionary that was kept is held by the cache entry, c[4]: its address cannot be recycled for another object)
        if c[4] is givencov and c[0] == tuple(keys) and c[2] == tuple(sorted(kw.items(), key=lambda kv: kv[0])) and c[1] == self._fingerprint(givencov):
            return c[3]
        return None

    def _keep_given(self, keys, givencov, kw, decomp):
        if self._reuse_givencov:
            try:
                kwkey = tuple(sorted(kw.items(), key=lambda kv: kv[0]))
                hash(kwkey)
            except TypeError:
                return
            self._givencov_cache = (tuple(keys), self._fingerprint(givencov), kwkey, decomp, givencov)

    def marginal_likelihood(self, given, givencov=None, **kw):
        """ Logarithm of the probability of the data (``_compute.py:383-422``) """
        decomp, ymean = self._prior_decomp(given, givencov, **kw)
        mll, _, _, _, _ = decomp.minus_log_normal_density(ymean, value=True)
        return -mll

    @classmethod
    def decompose(cls, posdefmatrix, solver='chol', **kw):
        """ Decompose a nonnegative definite matrix (``_compute.py:430-516``) """
        m = posdefmatrix if isinstance(posdefmatrix, DeviceArray) else numpy.asarray(posdefmatrix)
        assert m.size > 0
        assert m.ndim % 2 == 0
        half = m.ndim // 2
        head = m.shape[:half]
        tail = m.shape[half:]
        assert head == tail
        n = math.prod(head)
        if not isinstance(m, DeviceArray):
            m = m.reshape(n, n)
        decompcls = cls._getdecomp(solver)
        return decompcls(m, **kw)
