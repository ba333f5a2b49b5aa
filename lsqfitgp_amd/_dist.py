"""
1-D block-column-cyclic Cholesky across the GPUs of one node (the reference has no
distributed code, so this has no counterpart there and is judged on parity with the single-GPU factor).

One process per GPU.  Block column ``b`` (columns ``[b*nb, (b+1)*nb)``) lives on rank ``b % G`` with all its
rows.  Right-looking; the panels are factored, broadcast and consumed in row PIECES (``csrc/dist.hip``): piece 0 of
panel b is its diagonal square (it stays on its owner: the trailing updates only read rows below a panel's diagonal
block), pieces 1.. are slabs of the rows below.  Rows r of panel b+1 depend on rows r of panel b and on the two squares
only, so a slab moves from panel to panel (update of block column b+1 with the slab's rows of panel b, substitution
against square b+1, broadcast) independently of the slabs below it: the dependent chain of the factorisation is that of
one slab, the others follow as a pipeline.  Every stream of every rank takes the pieces in ONE common order (key
``2 b + i``), a topological order of all the dependencies:

    owner(b), piece 0       update the square with panel b-1, factor it                                  [look-ahead stream 1]
    owner(b), piece i > 0   update the piece's rows of block column b with panel b-1, solve them
                            against the square, pack them                                                [look-ahead stream 2]
    every rank              broadcast piece (b, i) from owner(b)                                         [comm stream, RCCL]
    every rank              update its block columns c >= b+2, rows of the piece, in one staircase launch  [compute stream]

The only other collectives are scalars (``max`` for the Gershgorin bound, ``sum`` for the log-determinant).

The product runs the schedule inside the library (``lgp_chol_factor_dist``); the list of pieces is the library's
(``lgp_chol_dist_plan``, host only) and `factor_schedule` below executes that same list over an abstract ``ops`` backend,
so that ownership, ordering, the dependencies and the panel slots are exercised with world_size > 1 on CPU (gloo + numpy
backend in ``tests/test_dist_gloo.py``).
"""

import ctypes

import os

import numpy

from . import _lib
from ._device import Context, DeviceArray


def kernel_dtype(kernel, x):
    """ dtype of the kernel's Gram blocks on the points x (float32 kernels evaluate in float32, `_kernels.py`) """
    return numpy.dtype(kernel._result_dtype(x[:0], x[:0]))


def _astype(A, dtype):
    from ._GP import _astype_dev
    return _astype_dev(A, dtype)


class Layout:
    """ ownership and offsets of the block columns """

    def __init__(self, n, nb, rank, nranks):
        assert n >= 1 and nb >= 1 and 0 <= rank < nranks
        self.n, self.nb, self.rank, self.nranks = int(n), int(nb), int(rank), int(nranks)
        self.nblocks = (self.n + self.nb - 1) // self.nb

    def owner(self, b):
        return b % self.nranks

    def local_index(self, b):
        return b // self.nranks

    def col0(self, b):
        return b * self.nb

    def width(self, b):
        return min(self.nb, self.n - b * self.nb)

    def rows(self, b):
        """ rows of the panel of block b: from its diagonal down """
        return self.n - b * self.nb

    def owned(self, rank=None):
        rank = self.rank if rank is None else rank
        return [b for b in range(self.nblocks) if self.owner(b) == rank]

    @property
    def nlocal(self):
        return len(self.owned())

    @property
    def local_cols(self):
        return max(self.nlocal, 1) * self.nb


def dist_plan(n, nb, nx=0, nranks=1, slab=None):
    """ the library's plan of the sharded factorisation: ``(pieces, nslots, slab)`` with ``pieces`` a list of
    ``(b, i, r0, r1)`` in the order in which every stream of every rank takes them.  ``slab=None``: the library's default
    for ``nranks`` (what `lgp_chol_factor_dist` uses).  Host only. """
    lib = _lib.load()
    slab_arg = ctypes.c_size_t(-1).value if slab is None else int(slab)
    ns, so = ctypes.c_size_t(), ctypes.c_size_t()
    count = lib.lgp_chol_dist_plan(n, nb, nx, nranks, slab_arg, None, 0, ctypes.byref(ns), ctypes.byref(so))
    if count < 0:
        raise _lib.LibraryError('lgp_chol_dist_plan: ' + _lib.last_error())
    buf = (ctypes.c_ulonglong * (4 * count))()
    lib.lgp_chol_dist_plan(n, nb, nx, nranks, slab_arg, buf, count, ctypes.byref(ns), ctypes.byref(so))
    pieces = [tuple(int(buf[4 * t + k]) for k in range(4)) for t in range(count)]
    return pieces, int(ns.value), int(so.value)


class Task:
    """ one enqueued operation of the sharded factorisation: ``kind`` on ``stream`` for piece ``piece = (b, i, r0, r1)``;
    it starts after the operations enqueued before it on its stream and after the records ``waits`` (ids of earlier
    tasks whose completion was recorded as an event), like a kernel behind ``hipStreamWaitEvent`` calls """
    __slots__ = ('index', 'stream', 'kind', 'piece', 'waits', 'extra')

    def __init__(self, index, stream, kind, piece, waits, extra=None):
        self.index, self.stream, self.kind, self.piece, self.waits, self.extra = index, stream, kind, piece, waits, extra

    def __repr__(self):
        return f'Task({self.index}, {self.stream}, {self.kind}, {self.piece}, waits={self.waits})'


def factor_tasks(layout, plan=None, nx=0, owner_comm=None):
    """
    What ``lgp_chol_factor_dist`` (``csrc/dist.hip``) enqueues on this rank, operation by operation, in the same order and
    with the same event waits: a list of `Task`.  (A readable restatement: ``tests/test_dist_trace.py`` holds it to the records
    that the C++ itself writes down, ``lgp_chol_dist_trace``.)  ``owner_comm``: ``'stream'`` -- a rank's own pieces are broadcast
    from the communication stream behind an event (the library's default with several ranks), ``'order'`` -- from the stream
    that packed them (one rank; ``LGP_DIST_OWNER_COMM``).  Streams: ``'s1'`` (squares), ``'s3'`` (rows below them), ``'sx'``
    (broadcasts), ``'sc'`` (bulk updates).  Kinds, for a piece ``(b, i, r0, r1)``:

        'U'      owner(b): rows of the piece of block column b -= P(b-1)[rows] P(b-1)[rows of block b]'  (i = 0: the square, on
                 's1'; i > 0: on 'sc', enqueued with the piece of panel b-1 that holds those rows, in front of its bulk update)
        'SQ'     owner(b), i = 0: the diagonal square factored (the library records one event per 128-column block of it and
                 lets the rows below follow block by block; here the square is one operation)
        'SOLVE'  owner(b), i > 0: the rows solved against the square
        'PACK'   owner(b), i > 0: the rows copied into panel slot b % nslots (for the broadcast: the owner's own updates read the rows
                 where they lie, behind 'SOLVE')
        'BCAST'  every rank, i > 0: the piece broadcast from owner(b) (a collective: same order on every rank); the owner issues it
                 on 'sx' behind the event of 'PACK' (or on 's3' right behind 'PACK': ``owner_comm='order'``), the others on 'sx' 
        'BULK'   every rank, i > 0: block columns ``extra`` (owned, >= b + 2) restricted to the rows of the piece
    """
    L, me = layout, layout.rank
    if owner_comm is None:
        owner_comm = 'order' if L.nranks <= 1 else 'stream'
    if plan is None:
        plan = dist_plan(L.n, L.nb, nx, L.nranks)
    pieces, nslots, _ = plan
    by_panel = {}
    for p in pieces:
        by_panel.setdefault(p[0], []).append(p)
    for v in by_panel.values():
        v.sort(key=lambda p: p[1])

    def cover(b, r):
        for p in by_panel[b]:
            if p[2] <= r < p[3]:
                return p[1]
        return by_panel[b][-1][1]

    tasks = []
    rec = {}                     # event name -> id of the task whose completion it last recorded (hipEventRecord)

    def push(stream, kind, piece, wait_names, extra=None):
        waits = sorted({rec[w] for w in wait_names if w in rec})        # (waiting on an event never recorded is a no-op)
        tasks.append(Task(len(tasks), stream, kind, piece, waits, extra))
        return tasks[-1].index

    def slot_events(b):
        if b < nslots:
            return []
        s = b % nslots
        ev = [('free_sc', s), ('free_l1', s)]
        prev = b - nslots                      # a panel of this rank's own is read in place by its updates: its slot by the broadcast alone
        if L.owner(prev) == me and len(by_panel[prev]) > 1:
            ev.append(('recv', prev, by_panel[prev][-1][1]))
        return ev

    def avail(b, i):
        """ the event behind which the rows of piece i of panel b may be read by this rank's updates: solved (own panel, read in
        the factor itself) or received (another rank's, read in its slot) """
        return ('solved', b, i) if L.owner(b) == me else ('recv', b, i)

    mine = L.owned()
    for p in pieces:
        b, i, r0, r1 = p
        last = i == by_panel[b][-1][1]
        if L.owner(b) == me:
            if i == 0:
                if b >= 1:          # U1 on the stream of the squares
                    w = [avail(b - 1, cover(b - 1, b * L.nb))]
                    if b >= 2:
                        w.append(('bulk', b - 2, cover(b - 2, b * L.nb)))
                    push('s1', 'U', p, w)
                t = push('s1', 'SQ', p, [])
                rec['sq', b] = t
                if b >= 1:
                    rec['free_l1', (b - 1) % nslots] = t
            else:                   # (U2 of these rows: on the compute stream, with the piece of panel b - 1 that holds them, below)
                rec['solved', b, i] = push('s3', 'SOLVE', p, [('sq', b)] + ([('u2', b - 1, cover(b - 1, r0))] if b >= 1 else []))
                rec['ready', b, i] = push('s3', 'PACK', p, slot_events(b))
        if i == 0:
            continue
        # (the owner: from the communication stream behind the pack's event, or from the stream that packed in its order with no
        # event between the two; the others on the communication stream once the slot is free)
        if L.owner(b) == me:
            rec['recv', b, i] = push('s3', 'BCAST', p, []) if owner_comm == 'order' else push('sx', 'BCAST', p, [('ready', b, i)])
        else:
            rec['recv', b, i] = push('sx', 'BCAST', p, slot_events(b))
        if b + 1 < L.nblocks and L.owner(b + 1) == me:
            # U2: the rows of this piece below the square of the look-ahead column b + 1 = one piece of panel b + 1
            r2 = max(r0, min((b + 2) * L.nb, L.n))
            t = None
            if r2 < r1:
                q = next(q for q in by_panel[b + 1] if q[2] == r2)
                assert q[3] == r1 and q[1] >= 1
                t = push('sc', 'U', q, [avail(b, i), avail(b, cover(b, (b + 1) * L.nb))])
            else:
                t = push('sc', 'NOP', p, [avail(b, i)])
            rec['u2', b, i] = t
        cols = [c for c in mine if c >= b + 2 and c * L.nb < r1]
        w = [avail(b, i)]
        if cols:
            w.append(avail(b, cover(b, cols[0] * L.nb)))
        t = push('sc', 'BULK', p, w, cols)
        rec['bulk', b, i] = t
        if last:
            rec['free_sc', b % nslots] = t
    return tasks, nslots


def factor_schedule(layout, ops, plan=None, nx=0):
    """ `factor_tasks` executed eagerly, in enqueue order, on ``ops`` (methods named after the task kinds in lower case,
    called with the piece, the panel slot and, for the bulk update, the block columns) """
    tasks, nslots = factor_tasks(layout, plan, nx)
    for t in tasks:
        getattr(ops, t.kind.lower())(t.piece, t.piece[0] % nslots, *([t.extra] if t.kind == 'BULK' else []))
    return tasks


def forward_solve_schedule(layout, ops):
    """
    Fan-in forward solve  X = L^-1 W  with L sharded by block columns and W (N x k, k small) replicated:
    every rank keeps the updates it generates in a local accumulator S (N x k); for block b the pending rows
    S[rows_b] are summed over the ranks (one small all-reduce of nb x k values), the owner subtracts them,
    solves with its diagonal block and adds L[below, b] X_b to its accumulator.  X_b stays on owner(b):
    the quadratic forms that follow are local partial sums plus one all-reduce.  ``ops`` provides
    ``sum_pending(b)`` (every rank), ``solve_block(b)`` and ``accumulate(b)`` (owner only).
    """
    for b in range(layout.nblocks):
        ops.sum_pending(b)
        if layout.owner(b) == layout.rank:
            ops.solve_block(b)
            if b + 1 < layout.nblocks:
                ops.accumulate(b)


class DistributedChol:
    """
    Regularised Cholesky of a kernel Gram matrix sharded by block columns over the ranks of ``ctx``'s communicator: the
    distributed counterpart of ``Chol(kernel.gram(x, x) + noise * I)`` (``_linalg/_decomp.py:380-393``) for any kernel
    object of this package (isotropic or BART, with scalar ``*`` / ``+``).  The power-of-two diagonal scaling, the
    Gershgorin jitter (bound reduced with a max over the ranks) and the final rescaling follow the reference step by
    step; the factorisation itself is ONE library call per rank (``lgp_chol_factor_dist``).
    """

    PANEL_PAD = 16          # padding (doubles) of the packed panels' leading dimension, see csrc/dist.hip

    def __init__(self, kernel, x, *, noise=0., epsrel='auto', epsabs=0, nb=1024, ctx=None, carry=None):
        """ ``carry=(xs, y)`` (optional): test points and data whose forward solve ``L^-1 [K(x, xs) | y]`` rides through the
        factorisation as extra rows of every rank's block columns (as in ``lgp_chol_factor_rhs``); `posterior` then
        needs no solve and no per-panel all-reduce """
        self.ctx = ctx or Context.default()
        lib = self.lib = _lib.load()
        import os, time
        timing = os.environ.get('LGP_DIST_TIMING') == '1'
        self.phases = {}

        def mark(name, t0=[None]):
            if timing:
                self.ctx.sync()
                now = time.perf_counter()
                if t0[0] is not None:
                    self.phases[name] = (now - t0[0]) * 1e3
                t0[0] = now
        mark('start')
        x = numpy.asarray(x).reshape(-1)
        n = len(x)
        self.layout = L = Layout(n, nb, self.ctx.rank, self.ctx.nranks)
        self.n = n
        self._carry = None
        nx = 0
        if carry is not None:
            xs_c = numpy.asarray(carry[0]).reshape(-1)
            y_c = numpy.asarray(carry[1], float).reshape(-1)
            if len(y_c) != n:
                raise ValueError(f'y has {len(y_c)} values, x has {n} points')
            mq = len(xs_c) + 1
            nx = -(-mq // 128) * 128                     # whole 128-row tiles
            self._carry = (xs_c, y_c, mq)
        self.nx = nx
        kdt = kernel_dtype(kernel, x)
        # isotropic cores that the fused kernel covers: a block column is built ALREADY regularised (scaled lower triangle, zeros
        # above, noise on the diagonal) together with the scales of every row and its Gershgorin sums -- one pass where the generic
        # route below makes four (Gram, scales, scaling, column sums)
        from ._kernels import _Isotropic
        fused = (isinstance(kernel, _Isotropic) and kdt == numpy.float64 and os.environ.get('LGP_DIST_FUSED_GRAM', '1') != '0'
                 and type(kernel).gram is _Isotropic.gram
                 and kernel._kind in (_lib.LGP_KERNEL_EXPQUAD, _lib.LGP_KERNEL_MATERNP, _lib.LGP_KERNEL_GAMMAEXP, _lib.LGP_KERNEL_CAUCHY)
                 and -(-n // nb) >= L.nranks)           # (every rank owns a block column: each writes the scales itself)
        if fused and n % nb == 0:                        # (a ragged last block keeps zeros in its padding columns)
            self.Lloc = DeviceArray.empty((n + nx, L.local_cols), numpy.float64, self.ctx)
            if nx:
                self.Lloc.rows(n, n + nx).fill_zero()
        else:
            self.Lloc = DeviceArray.zeros((n + nx, L.local_cols), numpy.float64, self.ctx)
        self.dblk_per_block = lib.lgp_chol_dblk_size(nb)
        self.dblk = DeviceArray.empty((max(L.nlocal, 1) * self.dblk_per_block,), numpy.float64, self.ctx)
        F = ctypes.c_double
        h = self.ctx.handle
        # ---- Gram build: every rank evaluates its own block columns in place, no communication; the points (replicated:
        # 8 d N bytes) go to the device once
        xdev = kernel.prepare(x, ctx=self.ctx)
        xsdev = kernel.prepare(self._carry[0], ctx=self.ctx, dtype=xdev.dtype) if self._carry is not None else None
        ydev = DeviceArray.from_host(self._carry[1].reshape(1, n), ctx=self.ctx) if self._carry is not None else None
        svec = DeviceArray.zeros((n,), numpy.float64, self.ctx)
        sums = DeviceArray.zeros((L.local_cols,), numpy.float64, self.ctx)
        if fused:
            from ._kernels import _ops_arrays
            d = xdev.array.shape[0]
            nops, codes, vals = _ops_arrays(kernel._outops, F)
            npar, hpar = kernel._par_args(F)
        onecall = fused and nb % 512 == 0 and L.nlocal > 0
        # the carried rows of all the rank's columns in one Gram launch and one copy (its points gathered in local column order)
        batch_carry = self._carry is not None and n % nb == 0 and L.nlocal > 0 and kdt == numpy.float64
        if batch_carry:
            xs_c, y_c, mq = self._carry
            cols = numpy.concatenate([numpy.arange(L.col0(b), L.col0(b) + L.width(b)) for b in L.owned()])
            xloc = kernel.prepare(x[cols], ctx=self.ctx, dtype=xdev.dtype)
            kernel.gram(xsdev, xloc, ctx=self.ctx, out=self.Lloc.block(n, n + mq - 1, 0, len(cols)))
            self.Lloc.block(n + mq - 1, n + mq, 0, len(cols)).assign(DeviceArray.from_host(y_c[cols].reshape(1, -1), ctx=self.ctx))
        if onecall:                                      # the rank's whole share in one launch: local block q = global block q nranks + rank
            out = self.Lloc.block(0, n, 0, L.nlocal * nb)
            _lib.check(lib.lgp_gram_reg_cols_f64(h, kernel._kind, int(kernel._p), npar, hpar, d, xdev.array.ptr, xdev.array.ld, n, L.rank * nb, L.nlocal * nb,
                                                 nb, L.nranks * nb,
                                                 int(kernel._loc is not None), F(kernel._loc or 0), int(kernel._scale is not None), F(kernel._scale or 1),
                                                 nops, codes, vals, F(noise or 0.), None, out.ptr, out.ld, svec.ptr, sums.ptr),
                       'lgp_gram_reg_cols')
        for b in L.owned():
            lb, w, c0 = L.local_index(b), L.width(b), L.col0(b)
            out = self.Lloc.block(0, n, lb * nb, lb * nb + w)
            if onecall:
                pass
            elif fused:
                _lib.check(lib.lgp_gram_reg_cols_f64(h, kernel._kind, int(kernel._p), npar, hpar, d, xdev.array.ptr, xdev.array.ld, n, c0, w, 0, 0,
                                                     int(kernel._loc is not None), F(kernel._loc or 0), int(kernel._scale is not None), F(kernel._scale or 1),
                                                     nops, codes, vals, F(noise or 0.), None, out.ptr, out.ld, svec.ptr, sums.ptr + lb * nb * 8),
                           'lgp_gram_reg_cols')
            else:
                blk = kernel.gram(xdev, xdev.sub(c0, c0 + w), ctx=self.ctx, out=out if out.dtype == kdt else None)
                if blk is not out:                           # float32 kernels: promoted into the float64 factor storage
                    out.assign(_astype(blk, numpy.float64))
                if noise:
                    diag = self.Lloc.block(c0, c0 + w, lb * nb, lb * nb + w)
                    _lib.check(lib.lgp_add_diag_f64(h, w, diag.ptr, diag.ld, None, F(noise)), 'lgp_add_diag')
            if self._carry is not None and not batch_carry:      # K(xs, x[block]) and y[block] below the matrix rows
                xs_c, y_c, mq = self._carry
                rows_k = self.Lloc.block(n, n + mq - 1, lb * nb, lb * nb + w)
                kb = kernel.gram(xsdev, xdev.sub(c0, c0 + w), ctx=self.ctx, out=rows_k if rows_k.dtype == kdt else None)
                if kb is not rows_k:
                    rows_k.assign(_astype(kb, numpy.float64))
                self.Lloc.block(n + mq - 1, n + mq, lb * nb, lb * nb + w).assign(ydev.block(0, 1, c0, c0 + w))
        mark('gram')
        # ---- s = diag_scale_pow2(K) (_decomp.py:356-361): own columns, then shared with one all-reduce
        if not fused:
            for b in L.owned():
                lb, w, c0 = L.local_index(b), L.width(b), L.col0(b)
                diag = self.Lloc.block(c0, c0 + w, lb * nb, lb * nb + w)
                _lib.check(lib.lgp_diag_scale_pow2_f64(h, w, diag.ptr, diag.ld, svec.ptr + c0 * 8), 'lgp_diag_scale_pow2')
            _lib.check(lib.lgp_allreduce_sum_f64(h, svec.ptr, n), 'lgp_allreduce_sum')
        self.ctx.sync()
        s_host = svec.to_host()
        self._unit = bool(numpy.all(s_host == 1))
        if not self._unit:                               # K~ = K / s / s[:, None] (:385); carried rows: columns / s
            for b in L.owned():
                lb, w, c0 = L.local_index(b), L.width(b), L.col0(b)
                blk = self.Lloc.block(0, n, lb * nb, lb * nb + w)
                if not fused:
                    _lib.check(lib.lgp_scale_rows_cols_f64(h, n, w, blk.ptr, blk.ld, svec.ptr, svec.ptr + c0 * 8, 1), 'lgp_scale_rows_cols')
                if nx:
                    ext = self.Lloc.block(n, n + nx, lb * nb, lb * nb + w)
                    _lib.check(lib.lgp_scale_rows_cols_f64(h, nx, w, ext.ptr, ext.ld, None, svec.ptr + c0 * 8, 1), 'lgp_scale_rows_cols')
        # ---- jitter: eps = epsrel * max_i sum_j |K~_ij| + epsabs (:245-255,349-354); column sums of the owned block columns
        # (= row sums of the symmetric matrix), maximum on the device, one scalar all-reduce
        meps = numpy.finfo(float).eps
        if isinstance(epsrel, str):
            epsrel = n * meps
        if isinstance(epsabs, str):
            epsabs = meps
        if not fused:
            for b in L.owned():
                lb, w = L.local_index(b), L.width(b)
                blk = self.Lloc.block(0, n, lb * nb, lb * nb + w)          # matrix rows only (not the carried rows)
                _lib.check(lib.lgp_colabssum_f64(h, n, w, blk.ptr, blk.ld, sums.ptr + lb * nb * 8), 'lgp_colabssum')
        red = DeviceArray.zeros((1,), numpy.float64, self.ctx)
        if L.nlocal:
            _lib.check(lib.lgp_max_f64(h, L.nlocal * nb, sums.ptr, red.ptr), 'lgp_max')
        _lib.check(lib.lgp_allreduce_max_f64(h, red.ptr, 1), 'lgp_allreduce_max')
        self.ctx.sync()
        eps_scaled = epsrel * float(red.to_host()[0]) + epsabs
        self.eps = eps_scaled * float(numpy.min(s_host * s_host))     # :393
        for b in L.owned():
            lb, w, c0 = L.local_index(b), L.width(b), L.col0(b)
            diag = self.Lloc.block(c0, c0 + w, lb * nb, lb * nb + w)
            _lib.check(lib.lgp_add_diag_f64(h, w, diag.ptr, diag.ld, None, F(eps_scaled)), 'lgp_add_diag')
        mark('regularisation')
        # ---- factorisation: the whole look-ahead schedule inside the library
        ldp = nb + self.PANEL_PAD
        _, nslots, self.slab = dist_plan(n, nb, nx, L.nranks)
        # (without peers nothing is packed or sent: the library takes no slots then)
        self._panels = DeviceArray.empty((nslots * (n + nx) * ldp if L.nranks > 1 or os.environ.get('LGP_DIST_FORCE_COMM') else 1,), numpy.float64, self.ctx)
        info = ctypes.c_int()
        _lib.check(lib.lgp_chol_factor_dist_f64(h, n, nb, nx, self.Lloc.ptr, self.Lloc.ld, self.dblk.ptr, self._panels.ptr, ldp, nslots,
                                                None if self._unit else svec.ptr, ctypes.byref(info)), 'lgp_chol_factor_dist')
        self._panels = None
        mark('factorisation')
        bad = DeviceArray.from_host(numpy.array([float(info.value != 0)]), ctx=self.ctx)
        _lib.check(lib.lgp_allreduce_max_f64(h, bad.ptr, 1))
        self.ctx.sync()
        if bad.to_host()[0] != 0:
            raise numpy.linalg.LinAlgError('cholesky decomposition not finite, probably matrix not pos def numerically')

    def logdet(self):
        """ sum_i log L_ii, local diagonal blocks + one scalar all-reduce (``_decomp.py:486``) """
        L, lib = self.layout, self.lib
        tot = 0.
        val = ctypes.c_double()
        if L.nranks == 1:                # the local matrix is the matrix: one diagonal, one call (and one host round trip instead of n / nb)
            _lib.check(lib.lgp_logdet_diag_f64(self.ctx.handle, L.n, self.Lloc.ptr, self.Lloc.ld, ctypes.byref(val)))
            return float(val.value)
        for b in L.owned():
            lb, w, c0 = L.local_index(b), L.width(b), L.col0(b)
            diag = self.Lloc.block(c0, c0 + w, lb * L.nb, lb * L.nb + w)
            _lib.check(lib.lgp_logdet_diag_f64(self.ctx.handle, w, diag.ptr, diag.ld, ctypes.byref(val)))
            tot += val.value
        red = DeviceArray.from_host(numpy.array([tot]), ctx=self.ctx)
        _lib.check(lib.lgp_allreduce_sum_f64(self.ctx.handle, red.ptr, 1))
        self.ctx.sync()
        return float(red.to_host()[0])

    def local_factor(self):
        """ host copy of the locally owned block columns of L: dict block -> (n - col0) x width, lower part """
        out = {}
        L = self.layout
        for b in L.owned():
            lb, w, c0 = L.local_index(b), L.width(b), L.col0(b)
            blk = self.Lloc.block(c0, L.n, lb * L.nb, lb * L.nb + w).to_host()
            blk[:w] = numpy.tril(blk[:w])
            out[b] = blk
        return out


    # ---- posterior / likelihood on the sharded factor --------------------------------------------------

    def _gemm(self, ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc):
        _lib.check(self.lib.lgp_gemm_f64(self.ctx.handle, int(ta), int(tb), m, n, k, ctypes.c_double(alpha), A, lda, B, ldb,
                                         ctypes.c_double(beta), C, ldc), 'lgp_gemm')

    def posterior(self, kernel, x, xs, y):
        """
        Posterior mean and covariance at ``xs`` and minus-log-density of ``y`` (``_GP/_compute.py:254-260``,
        ``_linalg/_decomp.py:405-420,484-488``) with the factor sharded over the ranks: W = [Kxxs | y] is built
        on every rank (the Gram build is O(N M)), the forward solve is `forward_solve_schedule`, and
        mean = X' x_y, Q = X' X, |x_y|^2 are local partial sums over the owned row blocks + one all-reduce.
        Every rank returns the same (mean, cov, mlnd).
        """
        L, lib, ctx = self.layout, self.lib, self.ctx
        n, nb = L.n, L.nb
        xs = numpy.asarray(xs).reshape(-1)
        m = len(xs)
        k = m + 1
        Kss = kernel.gram(xs, xs, ctx=ctx)
        if self._carry is not None and self._carry[2] == k and numpy.array_equal(self._carry[0], xs) \
                and numpy.array_equal(self._carry[1], numpy.asarray(y, float).reshape(-1)):
            # the solve was carried through the factorisation: rows n .. n+k-1 of the owned block columns hold
            # (L^-1 [Kxxs | y])^T restricted to those columns; [X'X | X'x_y ; . | x_y'x_y] = sum over columns + one all-reduce
            acc = DeviceArray.zeros((k, k), numpy.float64, ctx)
            if L.nlocal:
                Wt = self.Lloc.block(n, n + k, 0, L.local_cols)         # (the padding columns of a ragged last block are zero)
                self._gemm(False, True, k, k, Wt.shape[1], 1., Wt.ptr, Wt.ld, Wt.ptr, Wt.ld, 0., acc.ptr, acc.ld)
            _lib.check(lib.lgp_allreduce_sum_f64(ctx.handle, acc.ptr, k * k), 'lgp_allreduce_sum')
            ctx.sync()
            G = acc.to_host()
            mean = G[:m, m].copy()
            cov = Kss.to_host() - G[:m, :m]
            mlnd = 0.5 * (n * numpy.log(2 * numpy.pi) + 2 * self.logdet() + G[m, m])
            return mean, cov, mlnd
        Kxxs = kernel.gram(numpy.asarray(x).reshape(-1), xs, ctx=ctx)
        W = DeviceArray.empty((n, k), numpy.float64, ctx)
        W.block(0, n, 0, m).assign(Kxxs)
        W.block(0, n, m, k).assign(DeviceArray.from_host(numpy.asarray(y, float).reshape(n, 1), ctx=ctx))
        S = DeviceArray.zeros((n, k), numpy.float64, ctx)
        dist = self

        class Ops:
            def sum_pending(self, b):
                c0, w = L.col0(b), L.width(b)
                _lib.check(lib.lgp_allreduce_sum_f64(ctx.handle, S.ptr + c0 * k * 8, w * k), 'lgp_allreduce_sum')

            def solve_block(self, b):
                c0, w, lb = L.col0(b), L.width(b), L.local_index(b)
                Wb = W.block(c0, c0 + w, 0, k)
                Sb = S.block(c0, c0 + w, 0, k)
                _lib.check(lib.lgp_axpby_f64(ctx.handle, w, k, ctypes.c_double(1.), Wb.ptr, Wb.ld, ctypes.c_double(-1.), Sb.ptr, Sb.ld, Wb.ptr, Wb.ld))
                D = dist.Lloc.block(c0, c0 + w, lb * nb, lb * nb + w)
                wsp = dist.dblk.rows(lb * dist.dblk_per_block, (lb + 1) * dist.dblk_per_block)
                _lib.check(lib.lgp_trsm_f64(ctx.handle, w, D.ptr, D.ld, wsp.ptr, 0, Wb.ptr, Wb.ld, k), 'lgp_trsm')

            def accumulate(self, b):
                c0, w, lb = L.col0(b), L.width(b), L.local_index(b)
                below = n - (c0 + w)
                A = dist.Lloc.block(c0 + w, n, lb * nb, lb * nb + w)
                Wb = W.block(c0, c0 + w, 0, k)
                Sbel = S.block(c0 + w, n, 0, k)
                dist._gemm(False, False, below, k, w, 1., A.ptr, A.ld, Wb.ptr, Wb.ld, 1., Sbel.ptr, Sbel.ld)

        forward_solve_schedule(L, Ops())
        # local partial sums over the owned (solved) row blocks: [X'X | X'x_y ; . | x_y'x_y] = Xfull' Xfull
        acc = DeviceArray.zeros((k, k), numpy.float64, ctx)
        for b in L.owned():
            c0, w = L.col0(b), L.width(b)
            Wb = W.block(c0, c0 + w, 0, k)
            self._gemm(True, False, k, k, w, 1., Wb.ptr, Wb.ld, Wb.ptr, Wb.ld, 1., acc.ptr, acc.ld)
        _lib.check(lib.lgp_allreduce_sum_f64(ctx.handle, acc.ptr, k * k), 'lgp_allreduce_sum')
        ctx.sync()
        G = acc.to_host()
        mean = G[:m, m].copy()
        cov = Kss.to_host() - G[:m, :m]
        mlnd = 0.5 * (n * numpy.log(2 * numpy.pi) + 2 * self.logdet() + G[m, m])
        return mean, cov, mlnd
