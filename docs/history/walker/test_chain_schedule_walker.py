"""
Host logic of the alternative chain kernel (csrc/chainw.hip, walker.inc; lgp_chain_schedule_walker, contexts created with LGP_CHAIN_IMPL=walker): the order in which the resident workers take the tasks
of the diagonal squares must be a topological order of the blocked right-looking Cholesky (the algorithm LAPACK potrf runs
inside jlinalg.cholesky, src/lsqfitgp/_linalg/_decomp.py:388), must contain every task exactly once, and executing the
tasks in that order on the CPU -- with the far rows and the trailing updates done where the streams of potrf_chain do them --
must give the factor.  No device needed.
"""

import ctypes

import numpy as np
import pytest
from scipy import linalg

from lsqfitgp_amd import _lib

P, S, G = 0, 1, 2
NB = 1024


def schedule(n, with_lists=False):
    lib = _lib.load()
    m = lib.lgp_chain_schedule_walker(n, None, 0)
    assert m > 0, _lib.last_error()
    buf = np.zeros(2 * m, np.uint32)
    assert lib.lgp_chain_schedule_walker(n, buf.ctypes.data_as(ctypes.c_void_p), m) == m
    w0, r = buf[0::2], buf[1::2]
    tasks = [(int(a & 127), int(a >> 24), int((a >> 8) & 255), int(b), int((a >> 16) & 255)) for a, b in zip(w0, r)]   # type, k, j, r, c
    if with_lists:
        return tasks, [bool(a & 128) for a in w0]
    return tasks


@pytest.mark.parametrize('n', [256, 1024, 1152, 2048, 4096 + 384, 8192])
def test_complete_and_topological(n):
    """ P(k, j) stands for the walker's step j of panel k (csrc/walker.inc): potf2 of block j, the two row units of block j + 1
    solved against it, block j + 1 updated with them.  S / G are the workers' tasks: every other row unit, every other tile. """
    tasks = schedule(n)
    want = set()
    for k in range((n + NB - 1) // NB):
        nq = min(NB, n - NB * k) // 128
        for j in range(nq):
            want.add((P, k, j, 0, 0))
            for r in range(2 * j + 4, 2 * nq):
                want.add((S, k, j, r, 0))
                for c in range(2 * j + 2, r + 1):
                    want.add((G, k, j, r, c))
    assert len(tasks) == len(want) and set(tasks) == want
    # what the device code waits for (chain_kernel, walker_body.inc): flags and per-tile step counters of the task's panel
    flagP, flagS, cnt = set(), set(), {}
    lastk = 0
    for t, k, j, r, c in tasks:
        assert k >= lastk                       # panels follow one another
        lastk = k
        nq = min(NB, n - NB * k) // 128
        if t == P:
            # block j itself: the workers' updates of steps < j - 1 and the walker's own of step j - 1 (in registers)
            for tile in ((2 * j, 2 * j), (2 * j + 1, 2 * j), (2 * j + 1, 2 * j + 1)):
                assert cnt.get((k,) + tile, 0) >= j, ('P', k, j, tile)
            flagP.add((k, j))
            if j + 1 < nq:
                # the rows of block j + 1 and that block, with every update of the steps before this one
                for tile in ((2 * j + 2, 2 * j), (2 * j + 2, 2 * j + 1), (2 * j + 3, 2 * j), (2 * j + 3, 2 * j + 1),
                             (2 * j + 2, 2 * j + 2), (2 * j + 3, 2 * j + 2), (2 * j + 3, 2 * j + 3)):
                    assert cnt.get((k,) + tile, 0) >= j, ('W', k, j, tile)
                flagS.add((k, 2 * j + 2, j))
                flagS.add((k, 2 * j + 3, j))
                for tile in ((2 * j + 2, 2 * j + 2), (2 * j + 3, 2 * j + 2), (2 * j + 3, 2 * j + 3)):
                    cnt[(k,) + tile] = j + 1
        elif t == S:
            assert (k, j) in flagP
            assert cnt.get((k, r, 2 * j), 0) >= j and cnt.get((k, r, 2 * j + 1), 0) >= j, ('S', k, j, r)
            flagS.add((k, r, j))
        else:
            assert (k, r, j) in flagS and (k, c, j) in flagS, ('G', k, j, r, c)
            assert cnt.get((k, r, c), 0) == j, ('G order', k, j, r, c)      # updates of one tile in step order, never two at once
            cnt[(k, r, c)] = j + 1


def test_near_list():
    """ the tasks marked as the near list -- the rows of the block after next: what the walker reads one block later -- are
    exactly those, in step order; they have workers of their own (chain_kernel) """
    tasks, near = schedule(2048, with_lists=True)
    got = [t for t, f in zip(tasks, near) if f and t[1] == 1]
    want = []
    for j in range(6):
        rows = (2 * j + 4, 2 * j + 5)
        want += [(S, 1, j, r, 0) for r in rows]
        for c in range(2 * j + 2, 2 * j + 6):
            want += [(G, 1, j, r, c) for r in rows if c <= r]
    assert got == want
    assert not any(f for t, f in zip(tasks, near) if t[0] == P)      # the walker takes no list task


@pytest.mark.parametrize('n,nx', [(640, 0), (1024 + 384, 5), (2048 + 128, 0)])
def test_executing_the_schedule_factors_the_matrix(n, nx):
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n))
    K = a @ a.T + n * np.eye(n)
    B = rng.standard_normal((n, nx))
    Lw = linalg.cholesky(K, lower=True)
    want = np.vstack([Lw, linalg.solve_triangular(Lw, B, lower=True).T])
    M = np.vstack([np.tril(K), B.T])            # rows n .. n + nx - 1: carried right-hand sides

    def finish_panel(k):
        """ what the streams do once the square of panel k is factored: far rows, then the update of everything to the right """
        k0, kb = NB * k, min(NB, n - NB * k)
        sq = slice(k0, k0 + kb)
        M[k0 + kb:, sq] = linalg.solve_triangular(M[sq, sq], M[k0 + kb:, sq].T, lower=True).T
        t0 = k0 + kb
        if t0 < n:
            upd = M[t0:, sq] @ M[t0:n, sq].T
            M[t0:, t0:n] -= np.vstack([np.tril(upd[:n - t0]), upd[n - t0:]])

    cur = 0
    for t, k, j, r, c in schedule(n):
        if k != cur:
            finish_panel(cur)
            cur = k
        k0 = NB * k
        cj = slice(k0 + 128 * j, k0 + 128 * j + 128)
        if t == P:
            M[cj, cj] = linalg.cholesky(M[cj, cj], lower=True)
            if 128 * (j + 1) < min(NB, n - k0):       # the walker: the rows of block j + 1 solved, that block updated
                nj = slice(k0 + 128 * j + 128, k0 + 128 * j + 256)
                M[nj, cj] = linalg.solve_triangular(M[cj, cj], M[nj, cj].T, lower=True).T
                M[nj, nj] -= np.tril(M[nj, cj] @ M[nj, cj].T)
        elif t == S:
            rr = slice(k0 + 64 * r, k0 + 64 * r + 64)
            M[rr, cj] = linalg.solve_triangular(M[cj, cj], M[rr, cj].T, lower=True).T
        else:
            rr, cc = slice(k0 + 64 * r, k0 + 64 * r + 64), slice(k0 + 64 * c, k0 + 64 * c + 64)
            upd = M[rr, cj] @ M[cc, cj].T
            M[rr, cc] -= np.tril(upd) if r == c else upd
    finish_panel(cur)
    np.testing.assert_allclose(M, want, rtol=1e-10, atol=1e-10)


def test_rejects_what_the_kernel_does_not_cover():
    lib = _lib.load()
    assert lib.lgp_chain_schedule_walker(1000, None, 0) < 0       # not a multiple of 128
    assert lib.lgp_chain_schedule_walker(0, None, 0) < 0
