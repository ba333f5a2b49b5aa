"""
1-D block-column-cyclic Cholesky across the GPUs of one node (the reference has no
distributed code, so this has no counterpart there and is judged on parity with the single-GPU factor).

One process per GPU.  Block column ``b`` (columns ``[b*nb, (b+1)*nb)``) lives on rank ``b % G`` with all its
rows.  Right-looking with one-panel look-ahead:

    step b:   every rank has panel b (rows b*nb.., nb columns) in panel slot b%2
              owner(b+1):  update block column b+1 with panel b, factor it, pack it        [compute stream]
              all ranks:   broadcast panel b+1 from owner(b+1) into slot (b+1)%2            [comm stream, RCCL]
              all ranks:   update their other owned block columns c > b with panel b        [compute stream]

so the (N - k) x nb panel message (one ``ncclBroadcast`` per panel, every peer fed over its own xGMI link)
travels while the trailing updates (MFMA) run.  The only other collectives are scalars
(``max`` for the Gershgorin bound, ``sum`` for the log-determinant).

The product runs the schedule inside the library (``lgp_chol_factor_dist``, ``csrc/dist.hip``); `factor_schedule` below
states the same schedule over an abstract ``ops`` backend so that ownership, ordering and the panel slots can be exercised
with world_size > 1 on CPU (gloo + numpy backend in ``tests/test_dist_gloo.py``).
"""

import ctypes

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


def factor_schedule(layout, ops):
    """
    The look-ahead schedule described in the module docstring.  ``ops`` provides

        panel_factor(b), pack_panel(b, slot)                      compute stream, owner(b) only
        update(c, b, slot)                                        compute stream, owner(c) only:
                                                                  block column c -= P_b[rows of c] P_b[block rows of c]'
        update_rest(cols, b, slot)   (optional)                   the same for a run of consecutive owned block columns
        bcast_panel(b, slot)                                      comm stream, every rank
        compute_mark_ready(slot) / comm_wait_ready(slot)          owner: panel packed before it is sent
        comm_mark_received(slot) / compute_wait_received(slot)    panel arrived before it is used
        compute_mark_free(slot) / comm_wait_free(slot)            slot no longer read before it is overwritten
    """
    L, me = layout, layout.rank
    if L.owner(0) == me:
        ops.panel_factor(0)
        ops.pack_panel(0, 0)
        ops.compute_mark_ready(0)
        ops.comm_wait_ready(0)
    ops.bcast_panel(0, 0)
    ops.comm_mark_received(0)
    mine = L.owned()
    for b in range(L.nblocks):
        slot, nslot = b % 2, (b + 1) % 2
        ops.compute_wait_received(slot)
        nxt = b + 1
        if nxt < L.nblocks:
            if L.owner(nxt) == me:
                ops.update(nxt, b, slot)
                ops.panel_factor(nxt)
                ops.pack_panel(nxt, nslot)
                ops.compute_mark_ready(nslot)
                ops.comm_wait_ready(nslot)
            if b >= 1:
                ops.comm_wait_free(nslot)
            ops.bcast_panel(nxt, nslot)
            ops.comm_mark_received(nslot)
        rest = [c for c in mine if c > b and c != nxt]
        if rest:
            if hasattr(ops, 'update_rest'):
                ops.update_rest(rest, b, slot)          # all of them in one launch
            else:
                for c in rest:
                    ops.update(c, b, slot)
        ops.compute_mark_free(slot)
    ops.finish()


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
        self.Lloc = DeviceArray.zeros((n + nx, L.local_cols), numpy.float64, self.ctx)
        self.dblk_per_block = lib.lgp_chol_dblk_size(nb)
        self.dblk = DeviceArray.empty((max(L.nlocal, 1) * self.dblk_per_block,), numpy.float64, self.ctx)
        F = ctypes.c_double
        h = self.ctx.handle
        # ---- Gram build: every rank evaluates its own block columns in place, no communication; the points (replicated:
        # 8 d N bytes) go to the device once
        kdt = kernel_dtype(kernel, x)
        xdev = kernel.prepare(x, ctx=self.ctx)
        xsdev = kernel.prepare(self._carry[0], ctx=self.ctx, dtype=xdev.dtype) if self._carry is not None else None
        ydev = DeviceArray.from_host(self._carry[1].reshape(1, n), ctx=self.ctx) if self._carry is not None else None
        for b in L.owned():
            lb, w, c0 = L.local_index(b), L.width(b), L.col0(b)
            out = self.Lloc.block(0, n, lb * nb, lb * nb + w)
            blk = kernel.gram(xdev, xdev.sub(c0, c0 + w), ctx=self.ctx, out=out if out.dtype == kdt else None)
            if blk is not out:                           # float32 kernels: promoted into the float64 factor storage
                out.assign(_astype(blk, numpy.float64))
            if noise:
                diag = self.Lloc.block(c0, c0 + w, lb * nb, lb * nb + w)
                _lib.check(lib.lgp_add_diag_f64(h, w, diag.ptr, diag.ld, None, F(noise)), 'lgp_add_diag')
            if self._carry is not None:                  # K(xs, x[block]) and y[block] below the matrix rows
                xs_c, y_c, mq = self._carry
                rows_k = self.Lloc.block(n, n + mq - 1, lb * nb, lb * nb + w)
                kb = kernel.gram(xsdev, xdev.sub(c0, c0 + w), ctx=self.ctx, out=rows_k if rows_k.dtype == kdt else None)
                if kb is not rows_k:
                    rows_k.assign(_astype(kb, numpy.float64))
                self.Lloc.block(n + mq - 1, n + mq, lb * nb, lb * nb + w).assign(ydev.block(0, 1, c0, c0 + w))
        mark('gram')
        # ---- s = diag_scale_pow2(K) (_decomp.py:356-361): own columns, then shared with one all-reduce
        svec = DeviceArray.zeros((n,), numpy.float64, self.ctx)
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
        sums = DeviceArray.zeros((L.local_cols,), numpy.float64, self.ctx)
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
        self._panels = DeviceArray.empty((2 * (n + nx) * ldp,), numpy.float64, self.ctx)
        info = ctypes.c_int()
        _lib.check(lib.lgp_chol_factor_dist_f64(h, n, nb, nx, self.Lloc.ptr, self.Lloc.ld, self.dblk.ptr, self._panels.ptr, ldp,
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
