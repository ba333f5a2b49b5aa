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

The schedule (`factor_schedule`) is pure Python over an ``ops`` backend so that it can be exercised with
world_size > 1 on CPU (gloo + numpy backend in ``tests/test_dist_gloo.py``); `DeviceOps` is the product backend
(C-ABI kernels + RCCL).
"""

import ctypes

import numpy

from . import _lib
from ._device import Context, DeviceArray


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


class DeviceOps:
    """ product backend: C-ABI kernels on the local GPU, RCCL broadcasts on the auxiliary stream """

    def __init__(self, ctx, layout, Lloc, dblk, nx=0):
        """ nx: extra rows below the n matrix rows of ``Lloc`` (right-hand sides carried through the factorisation:
        they are "rows below" for every panel, like in lgp_chol_factor_rhs) """
        self.ctx, self.layout, self.Lloc, self.dblk, self.nx = ctx, layout, Lloc, dblk, int(nx)
        self.lib = _lib.load()
        n, nb = layout.n, layout.nb
        self.P = [DeviceArray.empty((n + self.nx, nb), numpy.float64, ctx) for _ in range(2)]
        self.ev = {k: [ctx.event(), ctx.event()] for k in ('ready', 'recv', 'free')}
        self.dblk_per_block = self.lib.lgp_chol_dblk_size(nb)

    # -- streams
    def _compute(self):
        _lib.check(self.lib.lgp_select_stream(self.ctx.handle, 0))

    def _comm(self):
        _lib.check(self.lib.lgp_select_stream(self.ctx.handle, 1))

    def _mark(self, kind, slot):
        self.ctx.record(self.ev[kind][slot])

    def _wait(self, kind, slot):
        _lib.check(self.lib.lgp_stream_wait_event(self.ctx.handle, self.ev[kind][slot]))

    def compute_mark_ready(self, slot):
        self._compute(); self._mark('ready', slot)

    def comm_wait_ready(self, slot):
        self._comm(); self._wait('ready', slot); self._compute()

    def comm_mark_received(self, slot):
        self._comm(); self._mark('recv', slot); self._compute()

    def compute_wait_received(self, slot):
        self._compute(); self._wait('recv', slot)

    def compute_mark_free(self, slot):
        self._compute(); self._mark('free', slot)

    def comm_wait_free(self, slot):
        self._comm(); self._wait('free', slot); self._compute()

    # -- data
    def _block(self, b):
        """ device view of the locally owned block column b from its diagonal down: rows(b) x width(b) """
        L = self.layout
        lb = L.local_index(b)
        return self.Lloc.block(L.col0(b), L.n + self.nx, lb * L.nb, lb * L.nb + L.width(b))

    def panel_factor(self, b):
        L = self.layout
        blk = self._block(b)
        w = self.dblk.rows(L.local_index(b) * self.dblk_per_block, (L.local_index(b) + 1) * self.dblk_per_block)
        _lib.check(self.lib.lgp_panel_factor_f64(self.ctx.handle, L.rows(b) + self.nx, L.width(b), blk.ptr, blk.ld, w.ptr, L.col0(b)), 'lgp_panel_factor')

    def pack_panel(self, b, slot):
        L = self.layout
        blk = self._block(b)
        dst = self.P[slot]
        _lib.check(self.lib.lgp_copy2d(self.ctx.handle, dst.ptr, L.width(b), blk.ptr, blk.ld, L.rows(b) + self.nx, L.width(b), 8), 'lgp_copy2d')

    def bcast_panel(self, b, slot):
        L = self.layout
        self._comm()
        _lib.check(self.lib.lgp_bcast(self.ctx.handle, self.P[slot].ptr, (L.rows(b) + self.nx) * L.width(b) * 8, L.owner(b)), 'lgp_bcast')
        self._compute()

    def update(self, c, b, slot):
        L = self.layout
        kb = L.width(b)
        off = L.col0(c) - L.col0(b)                 # first row of block c inside panel b
        A = self.P[slot].ptr + off * kb * 8
        blk = self._block(c)
        _lib.check(self.lib.lgp_update_nt_f64(self.ctx.handle, L.rows(c) + self.nx, L.width(c), kb, A, kb, A, kb, blk.ptr, blk.ld, 1), 'lgp_update_nt')

    def update_rest(self, cols, b, slot):
        """ every remaining owned block column with panel b in one "staircase" launch (lgp_update_nt_cyclic_f64) """
        L = self.layout
        kb = L.width(b)
        lb0 = L.local_index(cols[0])
        assert [L.local_index(c) for c in cols] == list(range(lb0, lb0 + len(cols)))
        g0 = L.col0(cols[0])
        A = self.P[slot].ptr + (g0 - L.col0(b)) * kb * 8
        ncols = sum(L.width(c) for c in cols)
        C = self.Lloc.block(g0, L.n + self.nx, lb0 * L.nb, lb0 * L.nb + ncols)
        if L.nb % 128:
            for c in cols:                              # block widths that are not whole tiles: one launch per block column
                self.update(c, b, slot)
            return
        _lib.check(self.lib.lgp_update_nt_cyclic_f64(self.ctx.handle, L.n - g0 + self.nx, ncols, kb, A, kb, C.ptr, C.ld, L.nb, L.nranks), 'lgp_update_nt_cyclic')

    def finish(self):
        self._compute()
        self.ctx.sync()


class DistributedChol:
    """
    Regularised Cholesky of a kernel Gram matrix sharded by block columns over the ranks of ``ctx``'s
    communicator: the distributed counterpart of ``Chol(kernel.gram(x, x) + noise * I)`` for kernels with unit
    diagonal (ExpQuad, Maternp, BART correlations), i.e. ``diag_scale_pow2`` is the identity
    (``_linalg/_decomp.py:356-361``); the jitter follows ``_decomp.py:245-255,349-354`` with the Gershgorin
    bound reduced (max) over the ranks.
    """

    def __init__(self, kernel, x, *, noise=0., epsrel='auto', epsabs=0, nb=1024, ctx=None, carry=None):
        """ ``carry=(xs, y)`` (optional): test points and data whose forward solve ``L^-1 [K(x, xs) | y]`` rides through the
        factorisation as extra rows of every rank's block columns (as in ``lgp_chol_factor_rhs``); `posterior` then
        needs no solve and no per-panel all-reduce """
        from ._kernels import _Isotropic, _leaf_columns
        if not isinstance(kernel, _Isotropic):
            raise NotImplementedError('the distributed Gram build covers the isotropic kernels')
        if kernel._outops:
            raise NotImplementedError('distributed path: kernels with unit diagonal only (no output algops)')
        self.ctx = ctx or Context.default()
        lib = self.lib = _lib.load()
        cols = _leaf_columns(numpy.asarray(x).reshape(-1))
        n = len(cols[0])
        self.layout = L = Layout(n, nb, self.ctx.rank, self.ctx.nranks)
        self.n = n
        xd = DeviceArray.from_host(numpy.stack([numpy.asarray(c, float) for c in cols]), ctx=self.ctx)   # replicated covariates: 8 d N bytes
        d = len(cols)
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
        self.dblk = DeviceArray.empty((max(L.nlocal, 1) * lib.lgp_chol_dblk_size(nb),), numpy.float64, self.ctx)
        F = ctypes.c_double
        # ---- Gram build: every rank evaluates its own block columns in place, no communication
        for b in L.owned():
            lb, w, c0 = L.local_index(b), L.width(b), L.col0(b)
            out = self.Lloc.block(0, n, lb * nb, lb * nb + w)
            yb = xd.block(0, d, c0, c0 + w)
            _lib.check(lib.lgp_gram_isotropic_f64(self.ctx.handle, kernel._kind, int(kernel._p), *kernel._par_args(F), d, xd.ptr, xd.ld, n, yb.ptr, yb.ld, w,
                                                  int(kernel._loc is not None), F(kernel._loc or 0), int(kernel._scale is not None), F(kernel._scale or 1),
                                                  0, None, None, 0, out.ptr, out.ld), 'lgp_gram_isotropic')
            if noise:
                diag = self.Lloc.block(c0, c0 + w, lb * nb, lb * nb + w)
                _lib.check(lib.lgp_add_diag_f64(self.ctx.handle, w, diag.ptr, diag.ld, None, F(noise)), 'lgp_add_diag')
            if self._carry is not None:                  # K(xs, x[block]) and y[block] below the matrix rows
                xs_c, y_c, mq = self._carry
                xsd = DeviceArray.from_host(numpy.stack([numpy.asarray(c_, float) for c_ in _leaf_columns(xs_c)]), ctx=self.ctx)
                rows_k = self.Lloc.block(n, n + mq - 1, lb * nb, lb * nb + w)
                _lib.check(lib.lgp_gram_isotropic_f64(self.ctx.handle, kernel._kind, int(kernel._p), *kernel._par_args(F), d, xsd.ptr, xsd.ld, mq - 1,
                                                      yb.ptr, yb.ld, w, int(kernel._loc is not None), F(kernel._loc or 0),
                                                      int(kernel._scale is not None), F(kernel._scale or 1), 0, None, None, 0, rows_k.ptr, rows_k.ld),
                           'lgp_gram_isotropic')
                self.Lloc.block(n + mq - 1, n + mq, lb * nb, lb * nb + w).assign(DeviceArray.from_host(y_c[c0:c0 + w].reshape(1, w), ctx=self.ctx))
        if 1 + noise > 2 or 1 + noise < 0.5:
            raise NotImplementedError('distributed path assumes diag_scale_pow2 == 1 (diagonal in [0.5, 2))')
        # ---- jitter: eps = epsrel * max_i sum_j |K_ij| + epsabs, column sums of the owned block columns
        meps = numpy.finfo(float).eps
        if isinstance(epsrel, str):
            epsrel = n * meps
        if isinstance(epsabs, str):
            epsabs = meps
        sums = DeviceArray.zeros((L.local_cols,), numpy.float64, self.ctx)
        gmax = 0.
        for b in L.owned():
            lb, w = L.local_index(b), L.width(b)
            blk = self.Lloc.block(0, n, lb * nb, lb * nb + w)          # matrix rows only (not the carried rows)
            _lib.check(lib.lgp_colabssum_f64(self.ctx.handle, n, w, blk.ptr, blk.ld, sums.ptr + lb * nb * 8), 'lgp_colabssum')
        if L.nlocal:
            gmax = float(numpy.max(sums.to_host()[:L.nlocal * nb]))
        red = DeviceArray.from_host(numpy.array([gmax]), ctx=self.ctx)
        _lib.check(lib.lgp_allreduce_max_f64(self.ctx.handle, red.ptr, 1), 'lgp_allreduce_max')
        self.ctx.sync()
        self.eps = epsrel * float(red.to_host()[0]) + epsabs
        for b in L.owned():
            lb, w, c0 = L.local_index(b), L.width(b), L.col0(b)
            diag = self.Lloc.block(c0, c0 + w, lb * nb, lb * nb + w)
            _lib.check(lib.lgp_add_diag_f64(self.ctx.handle, w, diag.ptr, diag.ld, None, F(self.eps)), 'lgp_add_diag')
        # ---- factorisation
        self.ops = DeviceOps(self.ctx, L, self.Lloc, self.dblk, nx)
        info0 = ctypes.c_int()
        _lib.check(lib.lgp_get_info(self.ctx.handle, ctypes.byref(info0)))       # reset
        factor_schedule(L, self.ops)
        info = ctypes.c_int()
        _lib.check(lib.lgp_get_info(self.ctx.handle, ctypes.byref(info)))
        bad = DeviceArray.from_host(numpy.array([float(info.value != 0)]), ctx=self.ctx)
        _lib.check(lib.lgp_allreduce_max_f64(self.ctx.handle, bad.ptr, 1))
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
                wsp = dist.dblk.rows(lb * dist.ops.dblk_per_block, (lb + 1) * dist.ops.dblk_per_block)
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
