"""
The N > 1 path on CPU.  ``lsqfitgp_amd._dist.factor_tasks`` restates, operation by operation, what the library's
``lgp_chol_factor_dist`` (csrc/dist.hip) enqueues on its four streams with which event waits, over the library's own plan
of pieces (``lgp_chol_dist_plan``); a numpy backend stands in for the device kernels (scipy supplies the factorisation of
the squares -- this is the checker side).  Two executions:

* a SIMULATOR of all ranks in one process that honours ONLY what HIP honours -- the order within a stream, the event
  waits, and the rendezvous of a collective (a broadcast completes when it is at the head of every rank's communication
  stream) -- and otherwise picks the next operation at random: no deadlock for any interleaving tried, no read of a panel
  slot that does not hold the expected rows (arrived too late, overwritten too early), and the gathered factor equals the
  single-process Cholesky.  World sizes up to 8, with and without carried right-hand sides, ragged sizes;
* ``torch.distributed`` (gloo, 127.0.0.1) with world_size 2 and 3: the same operations executed eagerly in enqueue order
  with real broadcasts between processes.
"""

import os
import socket

import numpy as np
import pytest
from scipy import linalg

from lsqfitgp_amd._dist import Layout, factor_tasks, factor_schedule, dist_plan


def test_layout():
    L = Layout(10, 3, 1, 2)
    assert L.nblocks == 4
    assert [L.owner(b) for b in range(4)] == [0, 1, 0, 1]
    assert L.owned() == [1, 3] and L.owned(0) == [0, 2]
    assert [L.width(b) for b in range(4)] == [3, 3, 3, 1]
    assert [L.rows(b) for b in range(4)] == [10, 7, 4, 1]
    assert L.local_index(3) == 1 and L.local_cols == 6
    L1 = Layout(5, 8, 0, 1)
    assert L1.nblocks == 1 and L1.width(0) == 5


class NumpyOps:
    """ numpy stand-in for the device kernels of one rank.  ``K``: the (n + nx) x n matrix whose top n x n part is the
    symmetric matrix to factor and whose last nx rows are carried right-hand sides (B^T); every read of a panel slot checks
    that the rows it reads belong to the panel it expects """

    def __init__(self, layout, K, nslots, dist=None):
        self.dist, self.L = dist, layout
        L = layout
        self.rows_all = K.shape[0]
        self.blocks = {b: K[L.col0(b):, L.col0(b):L.col0(b) + L.width(b)].copy() for b in L.owned()}
        self.P = [np.full((self.rows_all, L.nb), np.nan) for _ in range(nslots)]
        self.tag = [np.full(self.rows_all, -1) for _ in range(nslots)]
        self.solved = {b: np.zeros(self.rows_all, bool) for b in L.owned()}     # rows of an own panel that are final
        self.nslots = nslots
        self.log = []

    def _read(self, b, r0, r1):
        if self.L.owner(b) == self.L.rank:          # an own panel is read where it lies, once its rows are solved
            assert np.all(self.solved[b][r0:r1]), f'rows {r0}:{r1} of own panel {b} are not solved yet'
            c0 = self.L.col0(b)
            return self.blocks[b][r0 - c0:r1 - c0]
        slot = b % self.nslots
        assert np.all(self.tag[slot][r0:r1] == b), f'rows {r0}:{r1} of slot {slot} do not hold panel {b}'
        return self.P[slot][r0:r1, :self.L.width(b)]

    def u(self, p, slot):
        b, i, r0, r1 = p
        L = self.L
        c0, w = L.col0(b), L.width(b)
        A = self._read(b - 1, r0, r1)
        B = self._read(b - 1, c0, c0 + w)
        self.blocks[b][r0 - c0:r1 - c0] -= A @ B.T
        self.log.append(('u', b, i))

    def nop(self, p, slot):
        pass

    def sq(self, p, slot):
        b = p[0]
        w = self.L.width(b)
        blk = self.blocks[b]
        blk[:w] = linalg.cholesky(np.tril(blk[:w]) + np.tril(blk[:w], -1).T, lower=True)
        self.log.append(('sq', b))

    def solve(self, p, slot):
        b, i, r0, r1 = p
        c0, w = self.L.col0(b), self.L.width(b)
        blk = self.blocks[b]
        blk[r0 - c0:r1 - c0] = linalg.solve_triangular(blk[:w], blk[r0 - c0:r1 - c0].T, lower=True).T
        self.solved[b][r0:r1] = True

    def pack(self, p, slot):
        b, i, r0, r1 = p
        c0, w = self.L.col0(b), self.L.width(b)
        self.P[slot][r0:r1, :w] = self.blocks[b][r0 - c0:r1 - c0]
        self.tag[slot][r0:r1] = b

    def bcast(self, p, slot):
        b, i, r0, r1 = p
        w = self.L.width(b)
        buf = np.ascontiguousarray(self.P[slot][r0:r1, :w])
        if self.dist is not None:
            import torch
            self.dist.broadcast(torch.from_numpy(buf), src=self.L.owner(b))
        self.P[slot][r0:r1, :w] = buf
        self.tag[slot][r0:r1] = b
        self.log.append(('bcast', b, i))

    def bulk(self, p, slot, cols):
        b, i, r0, r1 = p
        L = self.L
        for c in cols:
            cc, wc = L.col0(c), L.width(c)
            rs = max(r0, cc)
            A = self._read(b, rs, r1)
            B = self._read(b, cc, cc + wc)
            self.blocks[c][rs - cc:r1 - cc] -= A @ B.T
        self.log.append(('bulk', b, i))


def _matrix(n, seed=0, nx=0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, n / 8, n)
    K = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2) + 0.05 * np.eye(n)
    if nx:
        K = np.concatenate([K, rng.standard_normal((nx, n))])
    return K


def _gather(n, nx, nb, world, blocks_of_rank):
    full = np.zeros((n + nx, n))
    for rank in range(world):
        L = Layout(n, nb, rank, world)
        blocks = blocks_of_rank(rank)
        assert sorted(blocks) == L.owned()
        for b, blk in blocks.items():
            w = L.width(b)
            blk = blk.copy(); blk[:w] = np.tril(blk[:w])
            full[L.col0(b):, L.col0(b):L.col0(b) + w] = blk
    return full


def _check_factor(full, K, n):
    ref = linalg.cholesky(K[:n], lower=True)
    np.testing.assert_allclose(full[:n], ref, rtol=1e-11, atol=1e-13)
    if K.shape[0] > n:        # carried rows: (L^-1 B)^T
        np.testing.assert_allclose(full[n:], linalg.solve_triangular(ref, K[n:].T, lower=True).T, rtol=1e-9, atol=1e-11)


def simulate(world, n, nb, nx, slab, seed, owner_comm=None):
    """ all ranks in one process; the next operation is drawn at random among those whose stream predecessors and event waits
    are complete (a broadcast: on every rank at once) """
    rng = np.random.default_rng(seed)
    K = _matrix(n, 0, nx)
    plan = dist_plan(n, nb, nx, world, slab)
    ranks = []
    for r in range(world):
        L = Layout(n, nb, r, world)
        tasks, nslots = factor_tasks(L, plan, nx, owner_comm)
        queues = {}
        for t in tasks:
            queues.setdefault(t.stream, []).append(t)
        # the broadcasts of a rank in issue order: the communicator runs them in that order whatever stream they were issued on
        # (the owner of a panel sends from the stream that packed it, 's3'; the others receive on 'sx')
        ranks.append(dict(ops=NumpyOps(L, K, nslots), tasks=tasks, queues=queues, pos={s: 0 for s in queues}, done=set(), nslots=nslots,
                          bcasts=[t.index for t in tasks if t.kind == 'BCAST'], bpos=0))
        for t in tasks:
            if t.kind == 'BCAST':
                in_order = owner_comm == 'order' or (owner_comm is None and world == 1)
                assert t.stream == ('s3' if L.owner(t.piece[0]) == r and in_order else 'sx')
    # every rank issues the same broadcasts in the same order
    order = [[t.piece for t in rk['tasks'] if t.kind == 'BCAST'] for rk in ranks]
    assert all(o == order[0] for o in order)
    remaining = sum(len(rk['tasks']) for rk in ranks)
    steps = 0
    while remaining:
        ready = []
        for r, rk in enumerate(ranks):
            for s, q in rk['queues'].items():
                if rk['pos'][s] < len(q):
                    t = q[rk['pos'][s]]
                    if all(w in rk['done'] for w in t.waits):
                        ready.append((r, s, t))
        # a collective needs every rank
        heads = [(r, s, t) for r, s, t in ready if t.kind == 'BCAST' and ranks[r]['bcasts'][ranks[r]['bpos']] == t.index]
        runnable = [(r, s, t) for r, s, t in ready if t.kind != 'BCAST']
        if len(heads) == world:
            assert len({t.piece for _, _, t in heads}) == 1
            runnable.append(('all', 'sx', heads))
        assert runnable, f'deadlock after {steps} operations, {remaining} left'
        r, s, t = runnable[rng.integers(len(runnable))]
        if r == 'all':
            piece = t[0][2].piece
            b, i, r0, r1 = piece
            root = ranks[b % world]
            slot = b % root['nslots']
            w = root['ops'].L.width(b)
            src = root['ops']
            assert np.all(src.tag[slot][r0:r1] == b), 'broadcast of rows that were not packed'
            data = src.P[slot][r0:r1, :w].copy()
            for rr, ss, tt in t:
                rk = ranks[rr]
                rk['ops'].P[slot][r0:r1, :w] = data
                rk['ops'].tag[slot][r0:r1] = b
                rk['done'].add(tt.index)
                rk['pos'][ss] += 1
                rk['bpos'] += 1
                remaining -= 1
        else:
            rk = ranks[r]
            getattr(rk['ops'], t.kind.lower())(t.piece, t.piece[0] % rk['nslots'], *([t.extra] if t.kind == 'BULK' else []))
            rk['done'].add(t.index)
            rk['pos'][s] += 1
            remaining -= 1
        steps += 1
    full = _gather(n, nx, nb, world, lambda rank: ranks[rank]['ops'].blocks)
    _check_factor(full, K, n)


@pytest.mark.parametrize('world,n,nb,nx,slab', [
    (1, 37, 8, 0, 0), (1, 40, 8, 3, 16), (2, 45, 8, 0, 16), (2, 64, 8, 5, 8), (3, 50, 7, 0, 14), (3, 61, 8, 4, 24),
    (4, 96, 8, 0, 16), (5, 70, 6, 2, 12), (8, 128, 8, 0, 32), (8, 150, 8, 6, 16), (2, 16, 16, 0, 16), (8, 130, 8, 3, 0),
])
def test_schedule_simulated_with_random_interleaving(world, n, nb, nx, slab):
    for seed in range(6):
        simulate(world, n, nb, nx, slab, seed, ('order', 'stream')[seed % 2])


def test_plan_properties():
    """ the library's plan: every row of every panel in exactly one piece, piece 0 the square, slabs cut at absolute multiples of
    the slab height, the order a topological order of the row-wise dependencies, enough panel slots for it """
    for n, nb, nx, world, slab in [(65536, 1024, 128, 8, None), (32768, 1024, 128, 1, None), (2600, 1024, 128, 2, 2048), (45, 8, 3, 2, 16)]:
        pieces, nslots, slab_used = dist_plan(n, nb, nx, world, slab)
        pos = {(b, i): k for k, (b, i, r0, r1) in enumerate(pieces)}
        by = {}
        for b, i, r0, r1 in pieces:
            by.setdefault(b, []).append((i, r0, r1))
        assert sorted(by) == list(range(-(-n // nb)))
        for b, v in by.items():
            v.sort()
            assert [i for i, _, _ in v] == list(range(len(v)))
            assert v[0][1] == b * nb and v[0][2] == min((b + 1) * nb, n)
            for (i, r0, r1), (j, s0, s1) in zip(v, v[1:]):
                assert r1 == s0 and r0 < r1
                if i >= 1 and slab_used:
                    assert r1 % slab_used == 0
            assert v[-1][2] == n + nx or (len(v) == 1 and nx == 0)
            for i, r0, r1 in v:
                if b >= 1:
                    # the piece of the previous panel with the same rows, and the one with the rows of block b, come first
                    for r in (r0, b * nb):
                        j = next(jj for jj, t0, t1 in by[b - 1] if t0 <= r < t1)
                        assert pos[b - 1, j] < pos[b, i]
                if i >= 1:
                    assert pos[b, i - 1] < pos[b, i]
        maxp = max(len(v) for v in by.values())
        assert nslots >= maxp // 2 + 2
        if world == 1 and slab is None:
            assert slab_used == 0 and maxp == 2
    assert dist_plan(65536, 1024, 128, 8)[2] == 8192


def test_schedule_single_rank():
    n, nb = 37, 8
    K = _matrix(n)
    L = Layout(n, nb, 0, 1)
    tasks, nslots = factor_tasks(L)
    ops = NumpyOps(L, K, nslots)
    factor_schedule(L, ops)
    full = _gather(n, 0, nb, 1, lambda rank: ops.blocks)
    _check_factor(full, K, n)
    # look-ahead: square 1 waits for panel 0's rows only -- an own panel: for their SOLVE, neither for the copy into the slot nor
    # for the broadcast -- not for the bulk update of the other block columns with panel 0 (which runs beside it on the compute stream)
    u10 = next(t for t in tasks if t.kind == 'U' and t.piece[:2] == (1, 0))
    assert [tasks[w].kind for w in u10.waits] == ['SOLVE'] and u10.stream == 's1'
    bulk01 = next(t for t in tasks if t.kind == 'BULK' and t.piece[:2] == (0, 1))
    assert bulk01.stream == 'sc' and bulk01.extra == [2, 3, 4]


def _worker(rank, world, port, n, nb, nx, slab, tmpdir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        K = _matrix(n, 0, nx)
        L = Layout(n, nb, rank, world)
        plan = dist_plan(n, nb, nx, world, slab)
        ops = NumpyOps(L, K, plan[1], dist)
        factor_schedule(L, ops, plan, nx)
        np.savez(os.path.join(tmpdir, f'rank{rank}.npz'), **{str(b): blk for b, blk in ops.blocks.items()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('world,n,nb,nx,slab', [(2, 45, 8, 0, 16), (3, 50, 7, 4, 14), (2, 16, 16, 0, None)])
def test_schedule_gloo(world, n, nb, nx, slab, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, nb, nx, slab, str(tmp_path)), nprocs=world, join=True)
    K = _matrix(n, 0, nx)

    def blocks(rank):
        data = np.load(os.path.join(tmp_path, f'rank{rank}.npz'))
        return {int(k): data[k] for k in data.files}
    _check_factor(_gather(n, nx, nb, world, blocks), K, n)


# ---- fan-in forward solve ------------------------------------------------------------------------------

class NumpySolveOps:
    def __init__(self, layout, Lfull, W, dist=None):
        import torch
        self.torch, self.dist, self.layout = torch, dist, layout
        L = layout
        self.cols = {b: Lfull[:, L.col0(b):L.col0(b) + L.width(b)].copy() for b in L.owned()}
        self.W = W.copy()
        self.S = np.zeros_like(W)
        self.solved = {}

    def sum_pending(self, b):
        L = self.layout
        c0, w = L.col0(b), L.width(b)
        buf = np.ascontiguousarray(self.S[c0:c0 + w])
        if self.dist is not None:
            t = self.torch.from_numpy(buf)
            self.dist.all_reduce(t)
        self.S[c0:c0 + w] = buf

    def solve_block(self, b):
        L = self.layout
        c0, w = L.col0(b), L.width(b)
        rhs = self.W[c0:c0 + w] - self.S[c0:c0 + w]
        self.W[c0:c0 + w] = linalg.solve_triangular(self.cols[b][c0:c0 + w], rhs, lower=True)
        self.solved[b] = self.W[c0:c0 + w].copy()

    def accumulate(self, b):
        L = self.layout
        c0, w = L.col0(b), L.width(b)
        self.S[c0 + w:] += self.cols[b][c0 + w:] @ self.W[c0:c0 + w]


def _solve_worker(rank, world, port, n, nb, tmpdir):
    import torch.distributed as dist
    from lsqfitgp_amd._dist import forward_solve_schedule
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        K = _matrix(n)
        Lf = linalg.cholesky(K, lower=True)
        W = np.random.default_rng(3).standard_normal((n, 5))
        L = Layout(n, nb, rank, world)
        ops = NumpySolveOps(L, Lf, W, dist)
        forward_solve_schedule(L, ops)
        np.savez(os.path.join(tmpdir, f'solve{rank}.npz'), **{str(b): v for b, v in ops.solved.items()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,n,nb', [(2, 45, 8), (3, 41, 6)])
def test_forward_solve_gloo(world, n, nb, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_solve_worker, args=(world, port, n, nb, str(tmp_path)), nprocs=world, join=True)
    K = _matrix(n)
    Lf = linalg.cholesky(K, lower=True)
    W = np.random.default_rng(3).standard_normal((n, 5))
    ref = linalg.solve_triangular(Lf, W, lower=True)
    got = np.zeros_like(ref)
    for rank in range(world):
        data = np.load(os.path.join(tmp_path, f'solve{rank}.npz'))
        L = Layout(n, nb, rank, world)
        for key in data.files:
            b = int(key)
            got[L.col0(b):L.col0(b) + L.width(b)] = data[key]
    np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-13)
