"""
GPU tests of the distributed pieces on ONE device: the device backend of the block-column schedule
(``DeviceOps`` + ``DistributedChol`` with a single rank) against the oracle, and the RCCL plumbing through a
one-rank communicator (ncclCommInitRank / ncclBroadcast / ncclAllReduce really run).  The multi-rank schedule
itself is covered on CPU by tests/test_dist_gloo.py, and on the device by two processes that share the GPU and
exchange panels through the host-callback transport (gloo), since RCCL refuses two ranks on one device.
"""

import ctypes

import numpy as np
import pytest

import oracle
from . import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lgp():
    import lsqfitgp_amd
    lsqfitgp_amd.Context.default()
    return lsqfitgp_amd


@pytest.mark.parametrize('n,nb,noise', [(300, 128, 0.01), (1500, 256, 0.01), (1100, 1024, 0.), (2500, 1024, 0.05)])
def test_distributed_chol_single_rank(lgp, n, nb, noise, rng):
    from lsqfitgp_amd._dist import DistributedChol
    x = rng.uniform(0, n / 16, n)
    d = DistributedChol(lgp.ExpQuad(), x, noise=noise, nb=nb)
    K = oracle.ExpQuad()(x[:, None], x[None, :]) + noise * np.eye(n)
    o = oracle.Chol(K)
    util.assert_allclose(d.eps, o.eps, rtol=1e-13)
    full = np.zeros((n, n))
    for b, blk in d.local_factor().items():
        c0 = d.layout.col0(b)
        full[c0:, c0:c0 + blk.shape[1]] = blk
    if noise:
        util.assert_close_matrices(full, o._L, rtol=1e-11)
    util.assert_close_matrices(full @ full.T, o._L @ o._L.T, rtol=1e-13)
    # noise-free: the ~1e-6 pivots of the numerically singular matrix make the log-determinant itself ill conditioned
    util.assert_allclose(d.logdet(), np.sum(np.log(np.diag(o._L))), rtol=1e-10 if noise else 1e-5, atol=1e-8)


@pytest.mark.parametrize('n,nb,slab,q', [(1500, 256, 512, 0), (3300, 1024, 1024, 5), (2000, 128, 384, 3), (2048, 256, 256, 0)])
def test_distributed_chol_single_rank_in_slabs(lgp, n, nb, slab, q, rng, monkeypatch):
    """ one rank, but with the panels cut into row slabs as for several ranks (LGP_DIST_SLAB): pieces of a panel solved, packed,
    "broadcast" and consumed one after the other, the bulk updates as row-offset staircase launches; with carried rows """
    from lsqfitgp_amd._dist import DistributedChol
    monkeypatch.setenv('LGP_DIST_SLAB', str(slab))
    x = rng.uniform(0, n / 16, n)
    xs = rng.uniform(0, n / 16, q) if q else None
    y = np.sin(x / 4) + 0.1 * rng.standard_normal(n)
    d = DistributedChol(lgp.ExpQuad(), x, noise=0.01, nb=nb, carry=(xs, y) if q else None)
    assert d.slab == -(-slab // nb) * nb
    K = oracle.ExpQuad()(x[:, None], x[None, :]) + 0.01 * np.eye(n)
    o = oracle.Chol(K)
    util.assert_allclose(d.eps, o.eps, rtol=1e-13)
    full = np.zeros((n, n))
    for b, blk in d.local_factor().items():
        c0 = d.layout.col0(b)
        full[c0:, c0:c0 + blk.shape[1]] = blk
    util.assert_close_matrices(full, o._L, rtol=1e-11)
    if q:
        mean, cov, mlnd = d.posterior(lgp.ExpQuad(), x, xs, y)
        mo = o.pinv_bilinear(oracle.ExpQuad()(x[:, None], xs[None, :]), y)
        assert np.linalg.norm(mean - mo) <= 1e-8 * np.linalg.norm(mo)
        util.assert_allclose(mlnd, o.minus_log_normal_density(y, value=True)[0], rtol=1e-10)


@pytest.mark.parametrize('n,nb', [(1500, 256), (2500, 1024), (1100, 128)])
def test_distributed_carried_posterior_single_rank(lgp, n, nb, rng):
    """ DistributedChol(carry=(xs, y)): posterior from the carried rows equals the fan-in solve and the oracle """
    from lsqfitgp_amd._dist import DistributedChol
    x = rng.uniform(0, n / 16, n); xs = rng.uniform(0, n / 16, 23)
    y = np.sin(x / 4) + 0.1 * rng.standard_normal(n)
    k = lgp.ExpQuad()
    a = DistributedChol(k, x, noise=0.01, nb=nb, carry=(xs, y))
    b = DistributedChol(k, x, noise=0.01, nb=nb)
    ma, ca, la = a.posterior(k, x, xs, y)
    mb, cb, lb_ = b.posterior(k, x, xs, y)
    assert a.eps == b.eps
    assert np.linalg.norm(ma - mb) <= 1e-11 * np.linalg.norm(mb)
    assert np.linalg.norm(ca - cb, 2) <= 1e-11
    util.assert_allclose(la, lb_, rtol=1e-12)
    K = oracle.ExpQuad()(x[:, None], x[None, :]) + 0.01 * np.eye(n)
    o = oracle.Chol(K)
    mo = o.pinv_bilinear(oracle.ExpQuad()(x[:, None], xs[None, :]), y)
    assert np.linalg.norm(ma - mo) <= 1e-8 * np.linalg.norm(mo)
    # different xs or y than the carried ones: falls back to the fan-in solve
    m2, _, _ = a.posterior(k, x, xs[:5], y)
    assert np.linalg.norm(m2 - mo[:5]) <= 1e-8 * np.linalg.norm(mo[:5])


def test_distributed_matern_3d(lgp, rng):
    from lsqfitgp_amd._dist import DistributedChol
    n = 900
    dt = [('a', float), ('b', float), ('c', float)]
    x = np.empty(n, dt)
    for f in 'abc':
        x[f] = rng.uniform(0, 8, n)
    d = DistributedChol(lgp.Maternp(p=2, scale=1.5), x, noise=0.01, nb=256)
    K = oracle.Maternp(p=2, scale=1.5)(x[:, None], x[None, :]) + 0.01 * np.eye(n)
    o = oracle.Chol(K)
    full = np.zeros((n, n))
    for b, blk in d.local_factor().items():
        c0 = d.layout.col0(b)
        full[c0:, c0:c0 + blk.shape[1]] = blk
    util.assert_close_matrices(full, o._L, rtol=1e-11)


def test_rccl_single_rank_communicator(lgp):
    """ a 1-rank RCCL communicator on a private context: unique id, init, broadcast, all-reduce, destroy """
    from lsqfitgp_amd import _lib
    ctx = lgp.Context(0)
    uid = lgp.Context.comm_unique_id()
    assert len(uid) == _lib.LGP_COMM_ID_BYTES and any(uid)
    ctx.comm_init(0, 1, uid)
    a = lgp.DeviceArray.from_host(np.arange(1000.), ctx=ctx)
    lib = _lib.load()
    _lib.check(lib.lgp_bcast(ctx.handle, a.ptr, 8000, 0), 'lgp_bcast')
    _lib.check(lib.lgp_allreduce_sum_f64(ctx.handle, a.ptr, 1000), 'allreduce sum')
    _lib.check(lib.lgp_allreduce_max_f64(ctx.handle, a.ptr, 1000), 'allreduce max')
    ctx.sync()
    np.testing.assert_array_equal(a.to_host(), np.arange(1000.))
    _lib.check(lib.lgp_comm_destroy(ctx.handle))


def test_distributed_posterior_single_rank(lgp, rng):
    """ fan-in solve + partial quadratic forms on the device backend (one rank) against the oracle """
    from lsqfitgp_amd._dist import DistributedChol
    n, m = 1300, 40
    x = rng.uniform(0, n / 16, n); xs = rng.uniform(0, n / 16, m)
    y = np.sin(x / 4) + 0.1 * rng.standard_normal(n)
    kern = lgp.ExpQuad()
    d = DistributedChol(kern, x, noise=0.01, nb=256)
    mean, cov, mlnd = d.posterior(kern, x, xs, y)
    go = oracle.GP(oracle.ExpQuad(), checkpos=False, checksym=False).addx(x, 'data').addx(xs, 'test')
    gc = {('data', 'data'): 0.01 * np.eye(n)}
    mo, co = go.predfromdata({'data': y}, 'test', gc, raw=True)
    lo = go.marginal_likelihood({'data': y}, gc)
    assert np.linalg.norm(mean - mo) <= 1e-8 * np.linalg.norm(mo)
    assert np.linalg.norm(cov - co, 2) <= 1e-8 * np.linalg.norm(go.prior('test', raw=True), 2)
    assert abs(-mlnd - lo) <= 1e-8 * abs(lo) + 1e-6


@pytest.mark.parametrize('n,nb,world,carry,amp,slab,bcast', [(1500, 256, 2, 0, 1, 512, 'plain'), (2600, 1024, 2, 1, 1, None, 'plain'), (2900, 256, 3, 1, 1, 768, 'scatter'),
                                                              (1700, 256, 2, 1, 37, 256, 'scatter'), (4100, 128, 4, 1, 1, 1024, 'scatter'), (2900, 256, 3, 0, 1, 512, 'plain'),
                                                              (3300, 256, 5, 1, 1, 512, 'scatter')])
def test_ranks_share_the_gpu(lgp, tmp_path, n, nb, world, carry, amp, slab, bcast):
    """ world size 2 to 5 on the device: real kernels, real stream/event choreography of the panel look-ahead, panels
    exchanged through the host-callback transport -- as one broadcast ('plain') or as scatter + all-gather between the
    ranks (`lgp_bcast`, three ranks or more; with two it is the same thing); factor, posterior and likelihood against the oracle """
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    m = 37
    # separate interpreters, one per rank
    script = os.path.join(os.path.dirname(__file__), 'two_rank_worker.py')
    env = dict(os.environ, LGP_TEST_CARRY=str(carry), LGP_TEST_AMP=str(amp))        # carry = 1: [Kxxs | y] rides through the factorisation
    env['LGP_TEST_SENDRECV'] = '1' if bcast == 'scatter' else '0'
    if slab is not None:
        env['LGP_DIST_SLAB'] = str(slab)                                            # (default for several ranks: 8 nb)
    procs = [subprocess.Popen([sys.executable, script, str(rank), str(world), str(port), str(n), str(nb), str(m), str(tmp_path)],
                              cwd=os.path.dirname(os.path.dirname(__file__)), env=env) for rank in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    rng = np.random.default_rng(7)
    x = rng.uniform(0, n / 16, n)
    xs = rng.uniform(0, n / 16, m)
    y = np.sin(x / 4) + 0.1 * rng.standard_normal(n)
    K = amp * (oracle.ExpQuad()(x[:, None], x[None, :]) + 0.01 * np.eye(n))
    o = oracle.Chol(K)
    Kxs = amp * oracle.ExpQuad()(x[:, None], xs[None, :])
    Kss = amp * oracle.ExpQuad()(xs[:, None], xs[None, :])
    mean_o = o.pinv_bilinear(Kxs, y)
    cov_o = Kss - o.ginv_quad(Kxs)
    mlnd_o = o.minus_log_normal_density(y, value=True)[0]
    full = np.zeros((n, n))
    seen = set()
    report = []                                                 # everything is compared before anything is asserted: a failure says what differs
    for rank in range(world):
        r = np.load(tmp_path / f'rank{rank}.npz')
        util.assert_allclose(r['eps'], o.eps, rtol=1e-13)
        em = np.linalg.norm(r['mean'] - mean_o) / np.linalg.norm(mean_o)
        ec = np.linalg.norm(r['cov'] - cov_o, 2) / np.linalg.norm(Kss, 2)
        el = abs(r['mlnd'] - mlnd_o) / abs(mlnd_o)
        if em > 1e-8 or ec > 1e-8 or el > 1e-10:
            report.append(f'rank {rank}: mean {em:.2e} cov {ec:.2e} density {el:.2e}')
        for b in r['owned']:
            assert int(b) not in seen and int(b) % world == rank      # block-column cyclic ownership
            seen.add(int(b))
            blk = r[f'b{b}']
            c0 = int(b) * nb
            full[c0:, c0:c0 + blk.shape[1]] = blk
            err = np.abs(blk - o._L[c0:, c0:c0 + blk.shape[1]]).max(axis=1)
            if err.max() > 1e-9:
                rows = np.flatnonzero(err > 1e-9) + c0
                report.append(f'block column {int(b)} (rank {rank}): max error {err.max():.2e} in rows {rows.min()}..{rows.max()} ({len(rows)} rows)')
    assert not report, '\n'.join(report)
    assert len(seen) == -(-n // nb)
    util.assert_close_matrices(full, o._L, rtol=1e-11)


@pytest.mark.parametrize('n,nb,world,slab,jitter', [(1100, 128, 4, 256, None), (1700, 256, 2, 256, 5)])
def test_live_schedule_is_the_dry_schedule(lgp, tmp_path, n, nb, world, slab, jitter):
    """ what the library enqueues on the device (LGP_DIST_TRACE) is, record for record, what its dry run writes down
    (lgp_chol_dist_trace) -- the schedule that tests/test_dist_trace.py proves free of unordered conflicts on CPU; with
    LGP_DIST_JITTER (spin kernels of random length in front of randomly chosen enqueues) the results stay right """
    import os
    import socket
    import subprocess
    import sys
    from . import dist_trace as dt
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    m = 37
    script = os.path.join(os.path.dirname(__file__), 'two_rank_worker.py')
    env = dict(os.environ, LGP_TEST_CARRY='1', LGP_TEST_AMP='1', LGP_TEST_SENDRECV='1', LGP_DIST_SLAB=str(slab), LGP_DIST_TRACE=str(tmp_path / 'trace'),
               LGP_CHAIN='0')                                      # (whether squares go to the chain kernel is a run-time question: off here, as in the dry run)
    if jitter is not None:
        env['LGP_DIST_JITTER'] = str(jitter)
    procs = [subprocess.Popen([sys.executable, script, str(rank), str(world), str(port), str(n), str(nb), str(m), str(tmp_path)],
                              cwd=os.path.dirname(os.path.dirname(__file__)), env=env) for rank in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    for rank in range(world):
        live, status = dt.load_trace(str(tmp_path / f'trace.rank{rank}'))
        assert status == 0
        dry = dt.dry_trace(n, nb, 128, rank, world, slab, 1)
        assert live.shape == dry.shape and np.array_equal(live, dry), f'rank {rank}: the live schedule differs from the dry run'
    rng = np.random.default_rng(7)
    x = rng.uniform(0, n / 16, n)
    xs = rng.uniform(0, n / 16, m)
    y = np.sin(x / 4) + 0.1 * rng.standard_normal(n)
    o = oracle.Chol(oracle.ExpQuad()(x[:, None], x[None, :]) + 0.01 * np.eye(n))
    mean_o = o.pinv_bilinear(oracle.ExpQuad()(x[:, None], xs[None, :]), y)
    for rank in range(world):
        r = np.load(tmp_path / f'rank{rank}.npz')
        assert np.linalg.norm(r['mean'] - mean_o) <= 1e-8 * np.linalg.norm(mean_o)
        for b in r['owned']:
            c0 = int(b) * nb
            util.assert_close_matrices(r[f'b{b}'], o._L[c0:, c0:c0 + r[f'b{b}'].shape[1]], rtol=1e-10)


def _sharded_against_single_gpu(lgp, tmp_path, n, world, slab=None):
    """ the sharded factorisation of configuration C4's workload (block columns of 1024, `world` ranks sharing the GPU, panel
    pieces through the host-staged transport, [Kxxs | y] carried through the factorisation) against the single-GPU driver on the
    same inputs: eps, the action of the factor on random vectors (L Z, L^T Z), its diagonal, posterior mean / covariance and the
    log-density """
    import os
    import socket
    import subprocess
    import sys
    from lsqfitgp_amd import _lib
    nb, m = 1024, 37
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    script = os.path.join(os.path.dirname(__file__), 'two_rank_worker.py')
    env = dict(os.environ, LGP_TEST_CARRY='1', LGP_TEST_BIG='1')
    if slab is not None:
        env['LGP_DIST_SLAB'] = str(slab)
    procs = [subprocess.Popen([sys.executable, script, str(rank), str(world), str(port), str(n), str(nb), str(m), str(tmp_path)],
                              cwd=os.path.dirname(os.path.dirname(__file__)), env=env) for rank in range(world)]
    # single-GPU reference while the ranks run (the same inputs as tests/two_rank_worker.py)
    rng = np.random.default_rng(7)
    x = rng.uniform(0, n / 16, n)
    xs = rng.uniform(0, n / 16, m)
    y = np.sin(x / 4) + 0.1 * rng.standard_normal(n)
    kern = lgp.ExpQuad()
    K = kern.gram(x, x)
    ctx = lgp.Context.default()
    assert _lib.load().lgp_add_diag_f64(ctx.handle, n, K.ptr, K.ld, None, ctypes.c_double(0.01)) == 0
    Kxs = kern.gram(x, xs).to_host()
    one = lgp.Chol(K, rhs=np.concatenate([Kxs, y[:, None]], axis=1))
    del K
    S = one.solved()
    mean1 = S[:, :m].T @ S[:, m]
    Kss = np.exp(-0.5 * (xs[:, None] - xs[None, :]) ** 2)
    cov1 = Kss - S[:, :m].T @ S[:, :m]
    mlnd1 = one.minus_log_normal_density(y, value=True)[0]
    Z = np.random.default_rng(11).standard_normal((n, 2))
    LZ1 = one.correlate(Z)
    LtZ1 = one.back_correlate(Z)
    diag1 = np.concatenate([np.diag(one.factor().block(r0, r0 + 4096, r0, r0 + 4096).to_host()) for r0 in range(0, n, 4096)])
    for p in procs:
        assert p.wait(timeout=1000) == 0
    LZ = np.zeros((n, 2))
    seen = set()
    for rank in range(world):
        r = np.load(tmp_path / f'rank{rank}.npz')
        assert int(r['slab']) == (8 * nb if slab is None else slab)
        util.assert_allclose(r['eps'], one.eps, rtol=1e-13)
        assert np.linalg.norm(r['mean'] - mean1) <= 1e-10 * np.linalg.norm(mean1)
        assert np.linalg.norm(r['cov'] - cov1, 2) <= 1e-10 * np.linalg.norm(Kss, 2)
        util.assert_allclose(r['mlnd'], mlnd1, rtol=1e-12)
        LZ += r['LZ']
        for b in r['owned']:
            b = int(b)
            assert b not in seen and b % world == rank
            seen.add(b)
            util.assert_close_matrices(r[f'lt{b}'], LtZ1[b * nb:(b + 1) * nb], rtol=1e-11)
            util.assert_allclose(r[f'dg{b}'], diag1[b * nb:(b + 1) * nb], rtol=1e-11)
    assert len(seen) == n // nb
    util.assert_close_matrices(LZ, LZ1, rtol=1e-12)


def test_sharded_path_two_ranks_large(lgp, tmp_path):
    """ N = 16384: 16 block columns, 2 ranks, slabs of 4096 rows """
    _sharded_against_single_gpu(lgp, tmp_path, 16384, 2, slab=4096)


def test_sharded_c4_two_ranks_full_size(lgp, tmp_path):
    """ BASELINE configuration C4 at its own size through the sharded path: N = 65536, 64 block columns of 1024, 2 ranks (2 x 17.2 GB
    of block columns + 6 panel slots each on the one 288 GB GPU), the default slabs of 8192 rows: 64 panels in 8 + 7 + ... pieces,
    every panel slot reused ten times, the staircase update at full width with row offsets, the carried rows.  (RCCL itself
    needs one GPU per rank: what travels here goes through the host-staged transport.) """
    _sharded_against_single_gpu(lgp, tmp_path, 65536, 2)


def test_sharded_three_ranks(lgp, tmp_path):
    """ three ranks (block columns dealt out 0, 1, 2, 0, ...; the look-ahead column's owner changes every panel), N = 12288 """
    _sharded_against_single_gpu(lgp, tmp_path, 12288, 3, slab=2048)


@pytest.mark.parametrize('nb', [256, 1024])
def test_distributed_scaled_kernel_and_bart(lgp, nb, rng):
    """ the sharded path is not restricted to unit-diagonal isotropic kernels: `kernel * c` (power-of-two diagonal scaling,
    eps rescaled by min s^2, L = L~ s[:, None]) and the BART kernel, against the oracle's Chol of the same matrix """
    from lsqfitgp_amd._dist import DistributedChol
    n = 1500
    x = rng.uniform(0, n / 16, n)
    for amp, noise in ((37., 0.3), (0.003, 1e-5)):
        d = DistributedChol(lgp.ExpQuad() * amp, x, noise=noise, nb=nb)
        K = amp * oracle.ExpQuad()(x[:, None], x[None, :]) + noise * np.eye(n)
        o = oracle.Chol(K)
        util.assert_allclose(d.eps, o.eps, rtol=1e-13)
        full = np.zeros((n, n))
        for b, blk in d.local_factor().items():
            c0 = d.layout.col0(b)
            full[c0:, c0:c0 + blk.shape[1]] = blk
        util.assert_close_matrices(full, o._L, rtol=1e-11)
        util.assert_allclose(d.logdet(), np.sum(np.log(np.diag(o._L))), rtol=1e-10)
    # BART (float64), unstructured integer-valued covariates
    p = 4
    X = rng.integers(0, 30, (700, p)).astype(float)
    splits = oracle.BART.splits_from_coord(X)
    from numpy.lib import recfunctions
    Xs = recfunctions.unstructured_to_structured(X)
    kd = lgp.BART(splits=splits, maxd=4, reset=2) * 2.5
    ko = oracle.BART(splits=splits, maxd=4, reset=2) * 2.5
    d = DistributedChol(kd, Xs, noise=0.1, nb=nb)
    o = oracle.Chol(ko(Xs[:, None], Xs[None, :]) + 0.1 * np.eye(700))
    util.assert_allclose(d.eps, o.eps, rtol=1e-12)
    full = np.zeros((700, 700))
    for b, blk in d.local_factor().items():
        c0 = d.layout.col0(b)
        full[c0:, c0:c0 + blk.shape[1]] = blk
    util.assert_close_matrices(full, o._L, rtol=1e-10)


def test_distributed_composite_kernel(lgp, rng):
    """ sums and products of kernels (and a host-evaluated factor) through the sharded path: every operand's points are prepared
    once, the block columns are combined on the device """
    from lsqfitgp_amd._dist import DistributedChol
    n = 1300
    x = rng.uniform(0, 40, n)
    std = lambda v: 1 + 0.3 * np.sin(v)
    make = lambda M: (M.ExpQuad(scale=2.) * M.Periodic(scale=1.5) + M.Maternp(p=1, scale=5.) * 0.5) * M.Rescaling(stdfun=std)
    d = DistributedChol(make(lgp), x, noise=0.05, nb=256)
    K = make(oracle)(x[:, None], x[None, :]) + 0.05 * np.eye(n)
    o = oracle.Chol(K)
    util.assert_allclose(d.eps, o.eps, rtol=1e-12)
    full = np.zeros((n, n))
    for b, blk in d.local_factor().items():
        c0 = d.layout.col0(b)
        full[c0:, c0:c0 + blk.shape[1]] = blk
    util.assert_close_matrices(full, o._L, rtol=1e-10)


@pytest.mark.parametrize('n,nb,G,rank', [(3000, 512, 1, 0), (3000, 512, 3, 1), (2560, 1024, 2, 1), (1300, 512, 2, 0)])
def test_fused_block_column_kernel(lgp, n, nb, G, rank):
    """ ``lgp_gram_reg_cols``: a rank's block-cyclic share of the regularised Gram matrix in one launch -- the scaled lower triangle
    of its columns (zeros above the diagonal), the scales of every row and the Gershgorin sums of its columns over all rows --
    against numpy on the oracle's Gram matrix; ragged last blocks included """
    import ctypes
    from lsqfitgp_amd import _lib
    from lsqfitgp_amd._kernels import _ops_arrays
    rng = np.random.default_rng(n + G)
    x = np.empty(n, [('a', float), ('b', float)])
    x['a'] = rng.uniform(0, 30, n); x['b'] = rng.uniform(0, 30, n)
    kd = lgp.Maternp(p=2, scale=1.7) * 37.0
    K = (oracle.Maternp(p=2, scale=1.7) * 37.0)(x[:, None], x[None, :]) + 0.3 * np.eye(n)
    s = 2.0 ** np.rint(0.5 * np.log2(np.diag(K)))
    Kt = K / s[:, None] / s[None, :]
    blocks = [b for b in range(-(-n // nb)) if b % G == rank]
    cols = np.concatenate([np.arange(b * nb, min(n, (b + 1) * nb)) for b in blocks])
    wloc = len(blocks) * nb
    ctx = lgp.Context.default()
    lib = _lib.load()
    xdev = kd.prepare(x, ctx=ctx)
    out = lgp.DeviceArray.zeros((n, wloc), np.float64, ctx)
    svec = lgp.DeviceArray.zeros((n,), np.float64, ctx)
    sums = lgp.DeviceArray.zeros((wloc,), np.float64, ctx)
    F = ctypes.c_double
    nops, codes, vals = _ops_arrays(kd._outops, F)
    npar, hpar = kd._par_args(F)
    _lib.check(lib.lgp_gram_reg_cols_f64(ctx.handle, kd._kind, int(kd._p), npar, hpar, 2, xdev.array.ptr, xdev.array.ld, n, rank * nb, wloc, nb, G * nb,
                                         0, F(0), 1, F(1.7), nops, codes, vals, F(0.3), None, out.ptr, out.ld, svec.ptr, sums.ptr), 'lgp_gram_reg_cols')
    got = out.to_host()
    np.testing.assert_array_equal(svec.to_host(), s)
    for q, b in enumerate(blocks):
        c0, c1 = b * nb, min(n, (b + 1) * nb)
        ref = np.tril(Kt, 0)[:, c0:c1]
        np.testing.assert_allclose(got[:, q * nb:q * nb + (c1 - c0)], ref, rtol=0, atol=4e-15)
        assert np.all(got[:c0, q * nb:q * nb + (c1 - c0)] == 0)
        np.testing.assert_allclose(sums.to_host()[q * nb:q * nb + (c1 - c0)], np.abs(Kt[:, c0:c1]).sum(0), rtol=1e-13)


@pytest.mark.parametrize('n,nb,amp', [(2100, 512, 1.), (1536, 256, 37.), (3072, 1024, 0.003)])
def test_fused_and_generic_gram_build_agree(lgp, rng, monkeypatch, n, nb, amp):
    """ the two ways `DistributedChol` builds a rank's share (the fused block-column kernel for the isotropic cores it covers --
    one launch when the block width is a multiple of 512, one per block otherwise -- and the generic kernel.gram + scaling +
    column sums, ``LGP_DIST_FUSED_GRAM=0``): same jitter to rounding, same posterior; non-unit scales included (amp) """
    x = rng.uniform(0, n / 16, n)
    xs = rng.uniform(0, n / 16, 40)
    y = np.sin(x) + 0.1 * rng.standard_normal(n)
    from lsqfitgp_amd._dist import DistributedChol
    kern = lgp.Maternp(p=2, scale=1.3) * amp
    out = []
    for flag in ('1', '0'):
        monkeypatch.setenv('LGP_DIST_FUSED_GRAM', flag)
        d = DistributedChol(kern, x, noise=0.01 * amp, nb=nb, carry=(xs, y))
        out.append((d.eps, d.posterior(kern, x, xs, y)))
    assert out[0][0] == pytest.approx(out[1][0], rel=1e-12)
    for a, b in zip(out[0][1], out[1][1]):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9 * amp)


def test_bench_gpus_2_launches_two_ranks():
    """ `python bench.py --gpus 2` by itself (no external launcher, no torch): two fresh rank processes sharing the one GPU through
    the host-staged transport run the sharded step and rank 0's line says so (VERDICT r03 "next" 1).  RCCL with more than one
    rank is NOT covered by this: one GPU per box. """
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'LGP_RDZV_PORT', 'LGP_DEVICE')}
    env['LGP_DIST_TRANSPORT'] = 'host'
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--config', 'c2', '--replica-config', 'c2', '--steps', '1', '--warmup', '1',
                          '--no-cpu-baseline', '--no-sweep'], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['scaling'] == 'strong' and out['transport'] == 'host' and out['rccl_ranks'] is None
    assert out['config']['N'] == 8192 and out['steps'] == 1
    # the sharded line's roofline: panel traffic against the links, per-rank phase times from inside the library (lgp_dist_stats)
    roof = out['roofline']
    assert roof['bound'] == 'xgmi' and roof['unit'] == 'GB/s' and roof['peak'] == 153. and out['bcast_algo'] == roof['bcast_algo'] == 'broadcast'
    assert roof['bytes_from_peers_per_rank_per_step'] > 0.3 * 8192 * 8192 * 8 / 2 / 2 and len(roof['per_rank_ms']) == 2
    for r in roof['per_rank_ms']:
        assert r['factorisation_wall'] > 0 and 0 <= r['waiting_for_pieces'] <= r['factorisation_wall'] and r['pack'] > 0
    assert roof['achieved'] is None or roof['frac'] == pytest.approx(roof['achieved'] / roof['peak'])
    assert 'sharded_path_error' not in out
    assert out['replicas']['scaling'] == 'weak' and out['replicas']['value'] > 0
    assert 'torch' not in res.stderr.lower()
    # the sharded step computed the same log marginal likelihood as the single-GPU step of the same workload
    sys.path.insert(0, root)
    import bench
    import lsqfitgp_amd as lgp
    wl = bench.Workload(bench.CONFIGS['c2'], lgp.Context.default())
    wl.step()
    lgp.Context.default().sync()
    assert abs(out['log_marginal_likelihood'] - wl.results['mll']) <= 1e-9 * abs(wl.results['mll'])


def test_bench_gpus_3_tries_both_broadcasts_and_a_second_workload():
    """ three ranks on the one GPU (host-staged transport): the sharded step is measured with the transport's broadcast first, then
    with scatter + all-gather behind one checked broadcast, the better one is the line; a second workload rides as
    `sharded_c3` (here a small one, named through the environment) """
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'LGP_RDZV_PORT', 'LGP_DEVICE')}
    env.update(LGP_DIST_TRANSPORT='host', LGP_BENCH_STRONG_CONFIG='r4096')
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '3', '--config', 'c2', '--replica-config', 'r4096', '--steps', '1', '--warmup', '1',
                          '--no-cpu-baseline', '--no-sweep'], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 3 and sorted(out['bcast_attempts_ms_per_step']) == ['plain', 'scatter'] and 'sharded_path_error' not in out
    best = min(out['bcast_attempts_ms_per_step'], key=out['bcast_attempts_ms_per_step'].get)
    assert out['bcast_algo'] == {'plain': 'broadcast', 'scatter': 'scatter + all-gather'}[best]
    assert out['roofline']['links_used_per_gpu'] == (2 if best == 'scatter' else 1)
    c3 = out['sharded_c3']
    assert c3['config']['N'] == 4096 and c3['roofline']['bound'] == 'xgmi' and c3['value'] > 0

