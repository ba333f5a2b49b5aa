"""
The N > 1 path on CPU: the block-column-cyclic look-ahead schedule of ``lsqfitgp_amd/_dist.py`` run with
world_size 2 and 3 over ``torch.distributed`` (gloo, 127.0.0.1) with a numpy backend standing in for the device
kernels (scipy supplies the panel factorisation -- this is the checker side, the product backend is
``DeviceOps``).  Verifies ownership, offsets, ordering and the panel slots: the gathered factor must equal the
single-process Cholesky.
"""

import os
import socket

import numpy as np
import pytest
from scipy import linalg

from lsqfitgp_amd._dist import Layout, factor_schedule


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
    """ numpy stand-in for DeviceOps (eager: the stream/event hooks only record the order they were called in) """

    def __init__(self, layout, K, dist=None):
        import torch
        self.torch, self.dist, self.layout = torch, dist, layout
        L = layout
        self.blocks = {b: K[L.col0(b):, L.col0(b):L.col0(b) + L.width(b)].copy() for b in L.owned()}
        self.P = [np.zeros((L.n, L.nb)), np.zeros((L.n, L.nb))]
        self.log = []
        self.state = {'recv': [None, None], 'free': [True, True]}

    def panel_factor(self, b):
        L = self.layout
        w = L.width(b)
        blk = self.blocks[b]
        D = linalg.cholesky(np.tril(blk[:w]) + np.tril(blk[:w], -1).T, lower=True)
        blk[:w] = D
        if L.rows(b) > w:
            blk[w:] = linalg.solve_triangular(D, blk[w:].T, lower=True).T
        self.log.append(('factor', b))

    def pack_panel(self, b, slot):
        L = self.layout
        assert self.state['free'][slot], 'panel slot overwritten while still in use'
        self.P[slot][:L.rows(b), :L.width(b)] = self.blocks[b]

    def bcast_panel(self, b, slot):
        L = self.layout
        if L.owner(b) != L.rank:
            assert self.state['free'][slot], 'panel slot overwritten while still in use'
        buf = np.ascontiguousarray(self.P[slot][:L.rows(b), :L.width(b)])
        if self.dist is not None:
            t = self.torch.from_numpy(buf)
            self.dist.broadcast(t, src=L.owner(b))
        self.P[slot][:L.rows(b), :L.width(b)] = buf
        self.state['recv'][slot] = b
        self.state['free'][slot] = False
        self.log.append(('bcast', b, slot))

    def update(self, c, b, slot):
        L = self.layout
        assert self.state['recv'][slot] == b, 'update with a panel that has not arrived'
        assert L.owner(c) == L.rank
        off = L.col0(c) - L.col0(b)
        kb = L.width(b)
        A = self.P[slot][off:L.rows(b), :kb]
        self.blocks[c] -= A @ A[:L.width(c)].T
        self.log.append(('update', c, b))

    def compute_mark_ready(self, slot): pass
    def comm_wait_ready(self, slot): pass
    def comm_mark_received(self, slot): pass
    def compute_wait_received(self, slot):
        assert self.state['recv'][slot] is not None
    def compute_mark_free(self, slot):
        self.state['free'][slot] = True
    def comm_wait_free(self, slot):
        assert self.state['free'][slot]
    def finish(self): pass


def _matrix(n, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, n / 8, n)
    return np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2) + 0.05 * np.eye(n)


def test_schedule_single_rank():
    n, nb = 37, 8
    K = _matrix(n)
    L = Layout(n, nb, 0, 1)
    ops = NumpyOps(L, K)
    factor_schedule(L, ops)
    full = np.zeros((n, n))
    for b, blk in ops.blocks.items():
        w = L.width(b)
        blk = blk.copy(); blk[:w] = np.tril(blk[:w])
        full[L.col0(b):, L.col0(b):L.col0(b) + w] = blk
    np.testing.assert_allclose(full, linalg.cholesky(K, lower=True), rtol=1e-12, atol=1e-14)
    # look-ahead order: panel b+1 is factored before the remaining updates with panel b
    first_update_other = next(i for i, e in enumerate(ops.log) if e[0] == 'update' and e[1] == 2 and e[2] == 0)
    assert ops.log.index(('factor', 1)) < first_update_other


def _worker(rank, world, port, n, nb, tmpdir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        K = _matrix(n)
        L = Layout(n, nb, rank, world)
        ops = NumpyOps(L, K, dist)
        factor_schedule(L, ops)
        np.savez(os.path.join(tmpdir, f'rank{rank}.npz'), **{str(b): blk for b, blk in ops.blocks.items()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('world,n,nb', [(2, 45, 8), (3, 50, 7), (2, 16, 16)])
def test_schedule_gloo(world, n, nb, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, nb, str(tmp_path)), nprocs=world, join=True)
    K = _matrix(n)
    ref = linalg.cholesky(K, lower=True)
    full = np.zeros((n, n))
    for rank in range(world):
        data = np.load(os.path.join(tmp_path, f'rank{rank}.npz'))
        L = Layout(n, nb, rank, world)
        assert sorted(int(k) for k in data.files) == L.owned()
        for key in data.files:
            b = int(key); w = L.width(b)
            blk = data[key].copy(); blk[:w] = np.tril(blk[:w])
            full[L.col0(b):, L.col0(b):L.col0(b) + w] = blk
    np.testing.assert_allclose(full, ref, rtol=1e-12, atol=1e-14)


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
