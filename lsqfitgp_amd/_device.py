"""
Device context and device-resident arrays (HBM) for the host-side mirror of
lsqfitgp's hot path.  Everything here is plumbing over the C-ABI
(``lgp_ctx_*``, ``lgp_malloc``, ``lgp_upload`` ...); there is no compute.
"""

import ctypes
import math
import os
import threading
import weakref

import numpy

from . import _lib


class Context:
    """ one GPU: a ``lgp_ctx`` (stream + workspace).  ``Context.default()`` is
    the process-wide context on ``LGP_DEVICE`` / ``LOCAL_RANK`` / device 0. """

    _default = None
    _lock = threading.Lock()

    def __init__(self, device=0):
        lib = _lib.load()
        count = ctypes.c_int(0)
        st = lib.lgp_device_count(ctypes.byref(count))
        if st != 0 or count.value == 0:
            raise _lib.LibraryError(
                'no HIP device visible: lsqfitgp_amd computes on an MI355X (gfx950) GPU '
                'and has no CPU fallback (' + _lib.last_error() + ')')
        handle = ctypes.c_void_p()
        _lib.check(lib.lgp_ctx_create(int(device), ctypes.byref(handle)), 'lgp_ctx_create')
        self._lib = lib
        self._handle = handle
        self.device = int(device)
        self.rank = 0
        self.nranks = 1
        # lgp_free synchronises the device (hipFree does); the buffers that the host classes drop all the time are
        # therefore collected here and freed in batches, large ones at once
        self._graveyard = _Graveyard(lib, handle)
        _lib._on_enomem.append(self._graveyard.flush)
        self._finalizer = weakref.finalize(self, _destroy_context, lib, handle, self._graveyard)

    @property
    def handle(self):
        return self._handle

    @classmethod
    def default(cls):
        with cls._lock:
            if cls._default is None:
                dev = os.environ.get('LGP_DEVICE')
                if dev is None:
                    dev = os.environ.get('LOCAL_RANK', '0')
                    count = ctypes.c_int(0)
                    _lib.load().lgp_device_count(ctypes.byref(count))
                    dev = int(dev) % max(count.value, 1)
                cls._default = cls(int(dev))
            return cls._default

    def sync(self):
        _lib.check(self._lib.lgp_sync(self._handle), 'lgp_sync')

    def info(self):
        name = ctypes.create_string_buffer(256)
        cus = ctypes.c_int()
        clock = ctypes.c_int()
        mem = ctypes.c_size_t()
        _lib.check(self._lib.lgp_device_info(self._handle, name, 256, ctypes.byref(cus), ctypes.byref(clock), ctypes.byref(mem)))
        return dict(name=name.value.decode(), cus=cus.value, clock_khz=clock.value, mem_bytes=mem.value)

    # ---- events (HIP-event timing on the context's stream)
    def event(self):
        ev = ctypes.c_void_p()
        _lib.check(self._lib.lgp_event_create(self._handle, ctypes.byref(ev)))
        return ev

    def record(self, ev):
        _lib.check(self._lib.lgp_event_record(self._handle, ev))

    def elapsed_ms(self, start, stop):
        ms = ctypes.c_float()
        _lib.check(self._lib.lgp_event_elapsed_ms(self._handle, start, stop, ctypes.byref(ms)))
        return ms.value

    def event_destroy(self, ev):
        self._lib.lgp_event_destroy(self._handle, ev)

    # ---- RCCL
    def comm_init(self, rank, nranks, unique_id):
        buf = ctypes.create_string_buffer(bytes(unique_id), _lib.LGP_COMM_ID_BYTES)
        _lib.check(self._lib.lgp_comm_init(self._handle, int(rank), int(nranks), buf), 'lgp_comm_init')
        self.rank, self.nranks = int(rank), int(nranks)

    def comm_init_host(self, rank, nranks, collective, sendrecv=False):
        """ alternative transport: ``collective(op, array, root)`` performs the collective in place on a host numpy
        view (op 0: broadcast of a uint8 array from ``root``; 1 / 2: all-reduce sum / max of a float64 array).
        ``sendrecv=True``: the transport also moves bytes between two ranks (op 5: from rank ``root & 0xffff`` to rank
        ``root >> 16``, called on every rank), and broadcasts travel as scatter + all-gather (`comm_set_bcast('scatter')`) """
        import numpy

        def trampoline(user, op, hbuf, count, root):
            try:
                if op in (0, 5):
                    arr = numpy.ctypeslib.as_array((ctypes.c_ubyte * count).from_address(hbuf))
                else:
                    arr = numpy.ctypeslib.as_array((ctypes.c_double * count).from_address(hbuf))
                collective(int(op), arr, int(root))
                return 0
            except Exception:      # reported through the library's error path
                import traceback
                traceback.print_exc()
                return 1

        cbtype = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int)
        self._hostcb = cbtype(trampoline)           # keep the callback object alive as long as the context
        _lib.check(self._lib.lgp_comm_init_host(self._handle, int(rank), int(nranks), ctypes.cast(self._hostcb, ctypes.c_void_p), None),
                   'lgp_comm_init_host')
        self.rank, self.nranks = int(rank), int(nranks)
        if sendrecv:
            self.comm_set_bcast('scatter')

    def comm_set_bcast(self, algo='auto', min_bytes=1 << 20):
        """ how `lgp_bcast` moves a buffer: 'auto' | 'plain' | 'scatter' (see include/lsqfitgp_hip.h) """
        _lib.check(self._lib.lgp_comm_set_bcast(self._handle, {'auto': 0, 'plain': 1, 'scatter': 2}[algo], int(min_bytes)), 'lgp_comm_set_bcast')

    @staticmethod
    def bcast_plan(nbytes, nranks, root):
        """ the transfers of a scatter + all-gather broadcast: (scatter, gather), lists of (src, dst, offset, length) """
        import numpy
        lib = _lib.load()
        ns = ctypes.c_size_t()
        cnt = lib.lgp_bcast_plan(int(nbytes), int(nranks), int(root), None, 0, ctypes.byref(ns))
        if cnt < 0:
            _lib.check(int(cnt), 'lgp_bcast_plan')
        buf = numpy.zeros((max(cnt, 1), 4), numpy.uint64)
        lib.lgp_bcast_plan(int(nbytes), int(nranks), int(root), buf.ctypes.data, int(cnt), ctypes.byref(ns))
        rows = [tuple(int(v) for v in r) for r in buf[:cnt]]
        return rows[:ns.value], rows[ns.value:]

    def bcast_check(self, nbytes=8 << 20, root=0):
        """ one broadcast of ``nbytes`` of a known pattern from ``root`` with the current algorithm, checked on every rank:
        None if this rank holds the pattern afterwards, else what went wrong (an exception's text, or 'wrong contents') """
        import numpy
        n = max(1, nbytes // 8)
        pattern = (numpy.arange(n, dtype=numpy.float64) * 1.25 + 3.).reshape(1, n)
        try:
            buf = DeviceArray.from_host(pattern if self.rank == root else numpy.zeros((1, n)), ctx=self)
            _lib.check(self._lib.lgp_bcast(self._handle, buf.ptr, n * 8, int(root)), 'lgp_bcast')
            self.sync()
            return None if numpy.array_equal(buf.to_host(), pattern) else 'wrong contents after the broadcast'
        except Exception as exc:
            return repr(exc)

    def dist_stats(self, enable=None):
        """ `lgp_dist_stats`: the last sharded factorisation measured from inside, as a dict; ``enable`` switches the measurement
        on / off for the calls that follow """
        out = (ctypes.c_double * 8)()
        _lib.check(self._lib.lgp_dist_stats(self._handle, -1 if enable is None else int(bool(enable)), out), 'lgp_dist_stats')
        keys = ('wall_ms', 'wait_ms', 'pack_ms', 'bytes_from_peers', 'bytes_own', 'pieces', 'bcast_algo', 'owner_via_comm_stream')
        d = dict(zip(keys, (float(v) for v in out)))
        d['bcast_algo'] = {0: None, 1: 'broadcast', 2: 'scatter + all-gather'}[int(d['bcast_algo'])]
        return d

    def comm_info(self):
        """ (rank, number of ranks, transport) as the communicator reports them; transport 'none' | 'rccl' | 'host' """
        r, n, t = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(self._lib.lgp_comm_info(self._handle, ctypes.byref(r), ctypes.byref(n), ctypes.byref(t)), 'lgp_comm_info')
        return r.value, n.value, ('none', 'rccl', 'host')[t.value]

    @staticmethod
    def comm_unique_id():
        buf = ctypes.create_string_buffer(_lib.LGP_COMM_ID_BYTES)
        _lib.check(_lib.load().lgp_comm_unique_id(buf), 'lgp_comm_unique_id')
        return buf.raw


class _Graveyard:
    """ device allocations whose owner died.  They are kept for reuse by size (a new-GP step of the callers allocates the same
    dozen blocks again and again, and lgp_free synchronises the device: 4 ms of a 13 ms step on the BART workload went
    there): every operation of the host classes is enqueued on the context's stream -- the library joins its internal
    streams back into it before it returns -- so a block handed out again is only touched by work that is ordered after
    everything that used it before.  Beyond CACHE_BYTES (or for blocks above BLOCK_BYTES) pointers wait for lgp_free instead,
    flushed when more than LIMIT_BYTES or LIMIT_COUNT are pending; everything is given back when an allocation fails and
    when the context goes away. """

    LIMIT_BYTES = 1 << 30
    LIMIT_COUNT = 256
    CACHE_BYTES = 40 << 30         # (of 288 GB: the matrices of a C3-sized step, 8.6 GB each, are reused instead of freed and mapped again)
    BLOCK_BYTES = 16 << 30

    def __init__(self, lib, handle):
        self.lib, self.handle = lib, handle
        self.pending, self.nbytes = [], 0
        self.cache, self.cached_bytes = {}, 0
        self.enabled = os.environ.get('LGP_HOST_POOL', '1') != '0'
        # blocks handed out and not yet buried; when the context object dies first (the garbage collector finalises the members of
        # a reference cycle in no particular order) the library context is kept until the last of them comes back
        self.live, self.closing, self.destroyed = 0, False, False
        # `bury` runs from weakref finalizers, i.e. from the garbage collector in whatever thread triggers it
        self._lock = threading.RLock()

    def take(self, nbytes):
        """ a cached block of exactly this size, or None """
        with self._lock:
            blocks = self.cache.get(nbytes)
            if not blocks:
                return None
            self.cached_bytes -= nbytes
            return blocks.pop()

    def bury(self, ptr, nbytes):
        """ one decision per block, taken inside one critical section: dropped (the library context is gone: its memory went with
        it), freed with the context (closing), cached, or queued for lgp_free """
        destroy = full = False
        with self._lock:
            self.live -= 1
            if self.destroyed:
                return
            if self.closing:
                self.pending.append(ptr)
                self.nbytes += nbytes
                destroy = self.live <= 0
                if destroy:
                    self.destroyed = True            # nobody else may touch the handle from here on
                    pending = self._take_all()
            elif self.enabled and nbytes <= self.BLOCK_BYTES and self.cached_bytes + nbytes <= self.CACHE_BYTES:
                self.cache.setdefault(nbytes, []).append(ptr)
                self.cached_bytes += nbytes
            else:
                self.pending.append(ptr)
                self.nbytes += nbytes
                full = self.nbytes > self.LIMIT_BYTES or len(self.pending) > self.LIMIT_COUNT
        if destroy:
            for p in pending:
                self.lib.lgp_free(self.handle, ctypes.c_void_p(p))
            self.lib.lgp_ctx_destroy(self.handle)
        elif full:
            self.flush(cache=False)

    def _take_all(self, cache=True):
        """ (lock held) everything that waits for lgp_free, and the cache with it """
        pending, self.pending, self.nbytes = self.pending, [], 0
        if cache:
            for blocks in self.cache.values():
                pending.extend(blocks)
            self.cache, self.cached_bytes = {}, 0
        return pending

    def flush(self, cache=True):
        with self._lock:
            if self.destroyed:
                return
            pending = self._take_all(cache)
        for ptr in pending:
            self.lib.lgp_free(self.handle, ctypes.c_void_p(ptr))


def _destroy_context(lib, handle, graveyard):
    try:
        _lib._on_enomem.remove(graveyard.flush)
    except ValueError:
        pass
    with graveyard._lock:
        if graveyard.destroyed:
            return
        if graveyard.live > 0:          # buffers of this context are still to be finalised: the last one destroys it (`bury`)
            graveyard.closing = True
            return
        graveyard.destroyed = True      # from here on `bury` drops what comes back and `take` finds nothing
        pending = graveyard._take_all()
    for ptr in pending:
        lib.lgp_free(handle, ctypes.c_void_p(ptr))
    lib.lgp_ctx_destroy(handle)


class _Buffer:
    """ owns one lgp_malloc allocation """

    def __init__(self, ctx, nbytes):
        ptr = ctypes.c_void_p(ctx._graveyard.take(int(nbytes)))
        if ptr.value is None:
            st = ctx._lib.lgp_malloc(ctx.handle, int(nbytes), ctypes.byref(ptr))
            if st == _lib.LGP_ENOMEM and (ctx._graveyard.pending or ctx._graveyard.cached_bytes):      # give back what is kept and retry
                ctx._graveyard.flush()
                st = ctx._lib.lgp_malloc(ctx.handle, int(nbytes), ctypes.byref(ptr))
            _lib.check(st, f'lgp_malloc({nbytes})')
        self.ctx = ctx
        self.ptr = ptr.value
        self.nbytes = int(nbytes)
        with ctx._graveyard._lock:
            ctx._graveyard.live += 1
        self._finalizer = weakref.finalize(self, ctx._graveyard.bury, ptr.value, int(nbytes))


_POISON = os.environ.get('LGP_POISON', '0') not in ('', '0')
_ALLOWED = (numpy.dtype(numpy.float64), numpy.dtype(numpy.float32), numpy.dtype(numpy.int32))


class DeviceArray:
    """
    A 1-D or 2-D row-major array resident in HBM: pointer + shape + leading
    dimension.  Views (``block``/``rows``/``col``) share the buffer.
    ``numpy.asarray(a)`` / ``a.to_host()`` download it.
    """

    __array_priority__ = 1000

    def __init__(self, buffer, ptr, shape, dtype, ld):
        self._buffer = buffer
        self.ptr = int(ptr)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = numpy.dtype(dtype)
        self.ld = int(ld)       # elements between rows (2-D) ; 1 for 1-D

    # ---- construction
    @classmethod
    def empty(cls, shape, dtype=numpy.float64, ctx=None):
        ctx = ctx or Context.default()
        dtype = numpy.dtype(dtype)
        assert dtype in _ALLOWED, dtype
        if isinstance(shape, (int, numpy.integer)):
            shape = (int(shape),)
        shape = tuple(int(s) for s in shape)
        assert 1 <= len(shape) <= 2, shape
        n = math.prod(shape)
        buf = _Buffer(ctx, max(n, 1) * dtype.itemsize)
        if _POISON:        # debugging aid (LGP_POISON=1): fresh blocks read as NaN / -1, so that a read of unwritten memory shows
            _lib.check(ctx._lib.lgp_memset(ctx.handle, buf.ptr, 0xff, buf.nbytes))
        return cls(buf, buf.ptr, shape, dtype, shape[-1] if len(shape) == 2 else 1)

    @classmethod
    def zeros(cls, shape, dtype=numpy.float64, ctx=None):
        out = cls.empty(shape, dtype, ctx)
        out.fill_zero()
        return out

    @classmethod
    def from_host(cls, a, dtype=None, ctx=None):
        a = numpy.ascontiguousarray(a, dtype=dtype)
        out = cls.empty(a.shape if a.ndim else (1,), a.dtype, ctx)
        if a.size:
            _lib.check(out.ctx._lib.lgp_upload(out.ctx.handle, out.ptr, a.ctypes.data, a.nbytes), 'lgp_upload')
        return out

    # ---- properties
    @property
    def ctx(self):
        return self._buffer.ctx

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return math.prod(self.shape)

    @property
    def itemsize(self):
        return self.dtype.itemsize

    @property
    def contiguous(self):
        return self.ndim == 1 or self.ld == self.shape[1] or self.shape[0] <= 1

    def __len__(self):
        return self.shape[0]

    # ---- views
    def block(self, r0, r1, c0, c1):
        assert self.ndim == 2 and 0 <= r0 <= r1 <= self.shape[0] and 0 <= c0 <= c1 <= self.shape[1]
        return DeviceArray(self._buffer, self.ptr + (r0 * self.ld + c0) * self.itemsize, (r1 - r0, c1 - c0), self.dtype, self.ld)

    def rows(self, r0, r1):
        if self.ndim == 1:
            assert 0 <= r0 <= r1 <= self.shape[0]
            return DeviceArray(self._buffer, self.ptr + r0 * self.itemsize, (r1 - r0,), self.dtype, 1)
        return self.block(r0, r1, 0, self.shape[1])

    def as2d(self):
        """ a vector as an (n, 1) column """
        if self.ndim == 2:
            return self
        return DeviceArray(self._buffer, self.ptr, (self.shape[0], 1), self.dtype, 1)

    def reshape_contiguous(self, shape):
        assert self.contiguous and math.prod(shape) == self.size
        shape = tuple(shape)
        return DeviceArray(self._buffer, self.ptr, shape, self.dtype, shape[-1] if len(shape) == 2 else 1)

    # ---- data movement
    @property
    def writes(self):
        """ how often the buffer behind this array (or any view of it) was overwritten through `assign` / `fill_zero` """
        return getattr(self._buffer, 'writes', 0)

    def _written(self):
        try:
            self._buffer.writes = getattr(self._buffer, 'writes', 0) + 1
        except AttributeError:            # a buffer object without attributes (foreign memory): nothing to count on
            pass

    def fill_zero(self):
        self._written()
        if self.contiguous:
            _lib.check(self.ctx._lib.lgp_memset(self.ctx.handle, self.ptr, 0, self.size * self.itemsize))
        else:
            for i in range(self.shape[0]):
                _lib.check(self.ctx._lib.lgp_memset(self.ctx.handle, self.ptr + i * self.ld * self.itemsize, 0, self.shape[1] * self.itemsize))

    def to_host(self):
        out = numpy.empty(self.shape, self.dtype)
        if self.size == 0:
            return out
        if self.contiguous:
            _lib.check(self.ctx._lib.lgp_download(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes), 'lgp_download')
        else:
            tmp = self.copy()
            _lib.check(self.ctx._lib.lgp_download(self.ctx.handle, out.ctypes.data, tmp.ptr, out.nbytes), 'lgp_download')
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.to_host()
        return a if dtype is None else a.astype(dtype, copy=False)

    def copy(self):
        out = DeviceArray.empty(self.shape, self.dtype, self.ctx)
        out.assign(self)
        return out

    def assign(self, src):
        """ self[...] = src (device array of the same shape and dtype) """
        assert src.shape == self.shape and src.dtype == self.dtype, (src.shape, self.shape, src.dtype, self.dtype)
        if self.size == 0:
            return
        self._written()
        rows, cols = (self.shape if self.ndim == 2 else (1, self.shape[0]))
        _lib.check(self.ctx._lib.lgp_copy2d(self.ctx.handle, self.ptr, max(self.ld, cols), src.ptr, max(src.ld, cols), rows, cols, self.itemsize), 'lgp_copy2d')

    def transpose(self):
        assert self.ndim == 2 and self.dtype.kind == 'f'
        out = DeviceArray.empty((self.shape[1], self.shape[0]), self.dtype, self.ctx)
        if self.size:
            f, _ = _lib.fn('lgp_transpose', self.dtype)
            _lib.check(f(self.ctx.handle, out.ptr, out.ld, self.ptr, self.ld, self.shape[0], self.shape[1]), 'lgp_transpose')
        return out

    def __repr__(self):
        return f'DeviceArray(shape={self.shape}, dtype={self.dtype}, ld={self.ld}, device={self.ctx.device})'


def asdevice(a, dtype=None, ctx=None):
    """ DeviceArray as is (dtype must match), host array uploaded """
    if isinstance(a, DeviceArray):
        if dtype is not None and numpy.dtype(dtype) != a.dtype:
            raise TypeError(f'device array has dtype {a.dtype}, expected {numpy.dtype(dtype)}')
        return a
    return DeviceArray.from_host(numpy.asarray(a), dtype=dtype, ctx=ctx)
