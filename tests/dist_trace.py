"""
Decoder, static checker and numpy executor of the schedule that ``lgp_chol_factor_dist_f64`` (csrc/dist.hip) REALLY enqueues.

The library writes every enqueue of the sharded factorisation down -- event waits and records, launches, copies, broadcasts,
each with its stream and the operand ranges it was handed -- either instead of performing it (``lgp_chol_dist_trace``: no
device needed, this is what the CPU suite uses) or while performing it (``LGP_DIST_TRACE=<path>`` on a GPU box).  Nothing here
restates the schedule: the records come out of the C++ that runs on the device.

* `hazards` -- the static check.  Two operations of one rank CONFLICT when one writes what the other reads or writes.  Every
  conflicting pair must be ordered by what HIP orders: the sequence within a stream and the event edges (a wait refers to the
  LAST record of its event at the time the wait is enqueued).  Updates ``C -= A B'`` commute arithmetically, so a simulator that
  compares values cannot see two of them racing on one tile; this check can.
* `simulate` -- the dynamic check: all ranks in one process, numpy in place of the kernels, the next operation drawn at random
  among those whose stream predecessors and event waits are complete (a broadcast: when it is at the head of every rank's
  communicator queue); the gathered factor must be LAPACK's.

Test infrastructure (like everything under tests/); record layout: csrc/common.h.
"""

import ctypes

import numpy as np
from scipy import linalg

FIELDS = 20
WAIT, RECORD, MARK = 1, 2, 3
GEMM, SUBST, POTF2, COPY2D, BCAST, CHAINSQ, TAIL, EXTRACT, SCALE, MEMSET = 10, 11, 12, 13, 14, 15, 16, 17, 18, 19
KIND_NAMES = {WAIT: 'WAIT', RECORD: 'RECORD', MARK: 'MARK', GEMM: 'GEMM', SUBST: 'SUBST', POTF2: 'POTF2', COPY2D: 'COPY2D', BCAST: 'BCAST',
              CHAINSQ: 'CHAINSQ', TAIL: 'TAIL', EXTRACT: 'EXTRACT', SCALE: 'SCALE', MEMSET: 'MEMSET'}
STREAMS = ('sc', 's1', 's3', 'sx')
DBLK_UNIT = 2 * 128 * 128 + 128 + 264 * 64          # LGP_DBLK (csrc/common.h): workspace of one 128-column diagonal block


def dry_trace(n, nb, nx, rank, world, slab=None, flags=-1):
    """ the records of one rank's schedule, from the library, without a device: array (count, 20) of int64 """
    from lsqfitgp_amd import _lib
    lib = _lib.load()
    sl = ctypes.c_size_t(-1).value if slab is None else int(slab)
    cnt = lib.lgp_chol_dist_trace(n, nb, nx, rank, world, sl, flags, None, 0)
    if cnt < 0:
        raise _lib.LibraryError('lgp_chol_dist_trace: ' + _lib.last_error())
    buf = np.zeros((cnt, FIELDS), np.int64)
    got = lib.lgp_chol_dist_trace(n, nb, nx, rank, world, sl, flags, buf.ctypes.data, cnt)
    assert got == cnt
    return buf


def load_trace(path):
    """ a file written by a live call under LGP_DIST_TRACE: (records, status) """
    raw = np.fromfile(path, np.int64)
    assert raw[0] == 0x3143525450474c and raw[1] == FIELDS, 'not a schedule trace'
    return raw[4:4 + raw[2] * FIELDS].reshape(-1, FIELDS).copy(), int(raw[3])


class Node:
    """ one record: ``kind`` on ``stream`` (index into STREAMS); operations carry `reads` / `writes` (lists of rectangles
    ``(space, r0, r1, c0, c1)``), `ctx` (the piece it belongs to) and `args` """
    __slots__ = ('index', 'kind', 'stream', 'args', 'reads', 'writes', 'ctx', 'event', 'clock')

    def __init__(self, index, kind, stream, args):
        self.index, self.kind, self.stream, self.args = index, kind, stream, args
        self.reads, self.writes, self.ctx, self.event, self.clock = [], [], None, None, None

    @property
    def is_op(self):
        return self.kind >= GEMM and self.kind != MEMSET

    def __repr__(self):
        return f'{KIND_NAMES.get(self.kind, self.kind)}#{self.index}@{STREAMS[self.stream] if self.stream < 4 else self.stream} {self.ctx}'


class Schedule:
    """ the decoded schedule of one rank """

    def __init__(self, rec):
        rec = np.asarray(rec, np.int64).reshape(-1, FIELDS)
        head = rec[0]
        assert head[0] == MARK and head[2] == -1, 'the first record is the header'
        (self.n, self.nb, self.nx, self.rank, self.world, self.slab, self.nslots, self.ldl, self.ldp, self.dblk_per_block,
         self.owner_via_stream) = (int(v) for v in head[3:14])
        self.rows_all = self.n + self.nx
        self.nodes = []
        ctx = None
        for k, r in enumerate(rec[1:]):
            kind, stream, a = int(r[0]), int(r[1]), [int(v) for v in r[2:]]
            if kind == MARK:
                ctx = (('lane', 'comm', 'bulk', 'u1')[a[0]], a[1], a[2], a[3], a[4])
                continue
            nd = Node(len(self.nodes), kind, stream, a)
            nd.ctx = ctx
            if kind in (WAIT, RECORD):
                nd.event = a[0]
            else:
                self._access(nd)
            self.nodes.append(nd)
        self._clocks()

    # ---- operand ranges
    def _rc(self, space, off, ld=None):
        """ (row, col) of an element offset inside its space; space 1: row = slot * rows_all + row inside the slot """
        if space == 0:
            return off // self.ldl, off % self.ldl
        if space == 1:
            return off // self.ldp, off % self.ldp
        assert space == 2
        return off // DBLK_UNIT, 0

    def _rect(self, space, off, rows, cols):
        r, c = self._rc(space, off)
        if space == 2:
            return (2, r, r + max(1, -(-cols // 128)), 0, 1)
        ld = self.ldl if space == 0 else self.ldp
        assert c + cols <= ld, 'operand crosses its leading dimension'
        return (space, r, r + rows, c, c + cols)

    def _access(self, nd):
        a, k = nd.args, nd.kind
        if k == GEMM:
            m, n, kk, spA, offA, lda, spB, offB, ldb, spC, offC, ldc, lower, cyc_nb, cyc_g1, roff, tt = a[:17]
            assert tt == 2, 'the factorisation only issues C -= A B^T'
            assert lda == (self.ldl if spA == 0 else self.ldp) and ldb == (self.ldl if spB == 0 else self.ldp) and ldc == self.ldl and spC == 0
            nd.reads.append(self._rect(spA, offA, m, kk))
            rB, cB = self._rc(spB, offB)
            rC, cC = self._rc(spC, offC)
            for j0 in range(0, n, 128):
                j1 = min(j0 + 128, n)
                first = 0
                if lower:            # element (i, j) is produced iff  j + cshift(j) - roff <= i  (gemm.hip: mshift)
                    cshift = cyc_nb * cyc_g1 * (j0 // cyc_nb) if cyc_nb else 0
                    first = max(0, j0 + cshift - roff)
                if first >= m:
                    continue         # (tiles entirely above the diagonal are neither computed nor read)
                cs = cyc_nb * cyc_g1 * (j0 // cyc_nb) if (lower and cyc_nb) else 0       # (B row = GLOBAL column: local column + cshift)
                nd.reads.append((spB, rB + j0 + cs, rB + j1 + cs, cB, cB + kk))
                nd.writes.append((0, rC + first, rC + m, cC + j0, cC + j1))
        elif k == SUBST:
            offw, jb, spX, offX, ldx, rows = a[:6]
            nd.reads.append(self._rect(2, offw, 1, jb))
            nd.writes.append(self._rect(spX, offX, rows, jb))
        elif k == POTF2:
            spA, offA, lda, nb, offw = a[:5]
            nd.writes.append(self._rect(spA, offA, nb, nb))
            nd.writes.append(self._rect(2, offw, 1, nb))
        elif k == COPY2D:
            spD, offD, ldd, spS, offS, lds, rows, cols = a[:8]
            nd.reads.append(self._rect(spS, offS, rows, cols))
            nd.writes.append(self._rect(spD, offD, rows, cols))
        elif k == BCAST:
            sp, off, elements, root = a[:4]
            assert sp == 1 and elements % self.ldp == 0
            rect = self._rect(1, off, elements // self.ldp, self.ldp)
            (nd.reads if root == self.rank else nd.writes).append(rect)
        elif k == CHAINSQ:
            spP, offP, ldl, kb, offw = a[:5]
            nd.writes.append(self._rect(spP, offP, kb, kb))
            nd.writes.append(self._rect(2, offw, 1, kb))
        elif k == EXTRACT:
            spL, offL, ldl, n, offw = a[:5]
            nd.reads.append(self._rect(spL, offL, n, n))
            nd.writes.append(self._rect(2, offw, 1, n))
        elif k == SCALE:
            rows, cols = a[:2]
            nd.writes.append((0, 0, rows, 0, cols))
        elif k == TAIL:
            pass                      # (one rank: the trailing matrix goes to the single-GPU chain driver, which orders its own streams)
        elif k == MEMSET:
            pass
        else:
            raise ValueError(f'unknown record kind {k}')

    # ---- what HIP orders: stream order + event edges
    def _clocks(self):
        """ clock[s] of a node = index of the latest node of stream s that is complete before the node starts (-1: none) """
        ns = 1 + max(nd.stream for nd in self.nodes)
        last_on = [-1] * ns
        last_rec = {}
        for nd in self.nodes:
            prev = last_on[nd.stream]
            clock = list(self.nodes[prev].clock) if prev >= 0 else [-1] * ns
            if prev >= 0:
                clock[nd.stream] = prev
            if nd.kind == WAIT and nd.event in last_rec:        # (a wait for an event never recorded in this call is a no-op)
                src = self.nodes[last_rec[nd.event]]
                other = list(src.clock)
                other[src.stream] = max(other[src.stream], src.index)
                clock = [max(x, y) for x, y in zip(clock, other)]
            nd.clock = clock
            if nd.kind == RECORD:
                last_rec[nd.event] = nd.index
            last_on[nd.stream] = nd.index
        self.nstreams = ns

    def ordered(self, a, b):
        """ a (enqueued first) completes before b starts """
        return b.clock[a.stream] >= a.index

    @property
    def ops(self):
        return [nd for nd in self.nodes if nd.is_op]

    def hazards(self, limit=20):
        """ conflicting accesses that nothing orders: list of (earlier op, later op, rectangle, 'WAW' | 'RAW' | 'WAR') """
        ops = [nd for nd in self.ops if nd.kind != TAIL]
        if not ops:
            return []
        rects, owner, write = [], [], []
        for i, nd in enumerate(ops):
            for r in nd.reads:
                rects.append(r); owner.append(i); write.append(False)
            for r in nd.writes:
                rects.append(r); owner.append(i); write.append(True)
        R = np.array(rects, np.int64).reshape(-1, 5)
        owner = np.array(owner)
        write = np.array(write)
        out = []
        start = np.searchsorted(owner, np.arange(len(ops)))          # first rectangle of each op (rects are grouped by op in order)
        for i, b in enumerate(ops):
            lo, hi = start[i], (start[i + 1] if i + 1 < len(ops) else len(R))
            if lo == 0:
                continue
            E = R[:lo]
            for q in range(lo, hi):
                sp, r0, r1, c0, c1 = R[q]
                hit = (E[:, 0] == sp) & (E[:, 1] < r1) & (r0 < E[:, 2]) & (E[:, 3] < c1) & (c0 < E[:, 4]) & (write[:lo] | write[q])
                for p in np.flatnonzero(hit):
                    a = ops[owner[p]]
                    if not self.ordered(a, b):
                        kind = 'WAW' if write[p] and write[q] else ('RAW' if write[p] else 'WAR')
                        out.append((a, b, tuple(int(v) for v in R[q]), kind))
                        if len(out) >= limit:
                            return out
        return out

    def unrecorded_waits(self):
        """ waits whose event has no record enqueued before them in this call (they would refer to a record of an EARLIER call) """
        seen, bad = set(), []
        for nd in self.nodes:
            if nd.kind == RECORD:
                seen.add(nd.event)
            elif nd.kind == WAIT and nd.event not in seen:
                bad.append(nd)
        return bad


# ---- numpy execution of the records ----------------------------------------------------------------------------------------

class RankState:
    """ the memory of one rank (block columns, panel slots) and its queues """

    def __init__(self, sched, K):
        s = self.s = sched
        self.L = np.zeros((s.rows_all, s.ldl))
        nlocal = 0
        for b in range(s.rank, -(-s.n // s.nb), s.world):
            c0, w = b * s.nb, min(s.nb, s.n - b * s.nb)
            self.L[:, nlocal * s.nb:nlocal * s.nb + w] = np.where(np.arange(s.rows_all)[:, None] >= c0 + np.arange(w)[None, :], K[:, c0:c0 + w], 0.)
            nlocal += 1
        self.nlocal = nlocal
        self.P = np.full((s.nslots * s.rows_all, s.ldp), np.nan)
        self.queues = {}
        for nd in s.nodes:
            self.queues.setdefault(nd.stream, []).append(nd)
        self.pos = {st: 0 for st in self.queues}
        self.done = np.zeros(len(s.nodes), bool)
        self.recorded = {}            # event -> index of the node whose completion its last ENQUEUED record stands for ... resolved statically below
        self.bcasts = [nd.index for nd in s.nodes if nd.kind == BCAST]
        self.bpos = 0
        # a wait refers to the last record of its event enqueued before it (host order = node order)
        last, self.wait_for = {}, {}
        for nd in s.nodes:
            if nd.kind == RECORD:
                last[nd.event] = nd.index
            elif nd.kind == WAIT and nd.event in last:
                self.wait_for[nd.index] = last[nd.event]

    def mat(self, space):
        return self.L if space == 0 else self.P

    def view(self, space, off, rows, cols):
        r, c = self.s._rc(space, off)
        return self.mat(space)[r:r + rows, c:c + cols]

    def ready(self, nd):
        w = self.wait_for.get(nd.index)
        return w is None or self.done[w]

    def run(self, nd):
        a, k = nd.args, nd.kind
        if k == GEMM:
            m, n, kk, spA, offA, lda, spB, offB, ldb, spC, offC, ldc, lower, cyc_nb, cyc_g1, roff, tt = a[:17]
            A, C = self.view(spA, offA, m, kk), self.view(spC, offC, m, n)
            j = np.arange(n)
            cshift = cyc_nb * cyc_g1 * (j // cyc_nb) if (lower and cyc_nb) else 0 * j
            rB, cB = self.s._rc(spB, offB)
            B = self.mat(spB)[rB + j + cshift, cB:cB + kk]           # (row of B = GLOBAL column of the block-cyclic layout)
            if lower:
                mask = (j + cshift - roff)[None, :] <= np.arange(m)[:, None]
                # rows of B that belong to tiles entirely above the diagonal are not read at all: they may not have arrived
                need = mask.any(axis=0)
                Bn = np.where(need[:, None], B, 0.)
                assert np.all(np.isfinite(Bn[need])) and np.all(np.isfinite(A)), f'{nd}: an operand is not there yet'
                C -= np.where(mask, A @ Bn.T, 0.)
            else:
                assert np.all(np.isfinite(A)) and np.all(np.isfinite(B)), f'{nd}: an operand is not there yet'
                C -= A @ B.T
        elif k == SUBST:
            offw, jb, spX, offX, ldx, rows = a[:6]
            X = self.view(spX, offX, rows, jb)
            D = self.diag[offw // DBLK_UNIT]
            X[...] = linalg.solve_triangular(D, X.T, lower=True).T
        elif k == POTF2:
            spA, offA, lda, nb, offw = a[:5]
            A = self.view(spA, offA, nb, nb)
            A[...] = np.tril(linalg.cholesky(np.tril(A) + np.tril(A, -1).T, lower=True)) + np.triu(A, 1)
            self.diag[offw // DBLK_UNIT] = np.tril(A).copy()
        elif k == CHAINSQ:
            spP, offP, ldl, kb, offw = a[:5]
            A = self.view(spP, offP, kb, kb)
            A[...] = np.tril(linalg.cholesky(np.tril(A) + np.tril(A, -1).T, lower=True)) + np.triu(A, 1)
            for q in range(0, kb, 128):
                self.diag[offw // DBLK_UNIT + q // 128] = np.tril(A[q:q + 128, q:q + 128]).copy()
        elif k == COPY2D:
            spD, offD, ldd, spS, offS, lds, rows, cols = a[:8]
            self.view(spD, offD, rows, cols)[...] = self.view(spS, offS, rows, cols)
        elif k in (EXTRACT, MEMSET, WAIT, RECORD):
            pass
        else:
            raise ValueError(f'cannot execute {nd}')

    diag = None


def simulate(scheds, K, seed):
    """ every rank's records executed over numpy in a random order that respects stream order, event waits and the communicator's
    issue order; returns the list of RankState """
    rng = np.random.default_rng(seed)
    world = len(scheds)
    ranks = [RankState(s, K) for s in scheds]
    for rk in ranks:
        rk.diag = {}
    order = [[tuple(rk.s.nodes[i].args[:4]) for i in rk.bcasts] for rk in ranks]
    assert all(len(o) == len(order[0]) for o in order), 'the ranks issue different numbers of broadcasts'
    for o in order:                # same (elements, root) sequence on every rank; offsets too (the slots are laid out alike)
        assert o == order[0], 'the ranks issue their broadcasts in different orders'
    remaining = sum(len(rk.s.nodes) for rk in ranks)
    while remaining:
        runnable, heads = [], []
        for r, rk in enumerate(ranks):
            for st, q in rk.queues.items():
                if rk.pos[st] < len(q):
                    nd = q[rk.pos[st]]
                    if not rk.ready(nd):
                        continue
                    if nd.kind == BCAST:
                        if rk.bcasts[rk.bpos] == nd.index:
                            heads.append((r, st, nd))
                    else:
                        runnable.append((r, st, nd))
        if len(heads) == world:
            runnable.append((-1, None, heads))
        assert runnable, f'deadlock with {remaining} records left'
        r, st, nd = runnable[rng.integers(len(runnable))]
        if r < 0:
            sp, off, elements, root = nd[0][2].args[:4]
            src = ranks[root]
            rows = elements // src.s.ldp
            data = src.view(1, off, rows, src.s.ldp).copy()
            w = min(src.s.nb, src.s.ldp)
            assert np.all(np.isfinite(data[:, :1])), 'broadcast of rows that were not packed'
            for rr, ss, dd in nd:
                rk = ranks[rr]
                if rr != root:
                    rk.view(1, off, rows, rk.s.ldp)[...] = data
                rk.done[dd.index] = True
                rk.pos[ss] += 1
                rk.bpos += 1
                remaining -= 1
        else:
            rk = ranks[r]
            rk.run(nd)
            rk.done[nd.index] = True
            rk.pos[st] += 1
            remaining -= 1
    return ranks


def gather(ranks):
    """ the factor (and the carried rows) from the ranks' block columns """
    s0 = ranks[0].s
    full = np.zeros((s0.rows_all, s0.n))
    for rk in ranks:
        s = rk.s
        for q, b in enumerate(range(s.rank, -(-s.n // s.nb), s.world)):
            c0, w = b * s.nb, min(s.nb, s.n - b * s.nb)
            blk = rk.L[:, q * s.nb:q * s.nb + w].copy()
            full[:, c0:c0 + w] = np.where(np.arange(s.rows_all)[:, None] >= c0 + np.arange(w)[None, :], blk, 0.)
    return full
