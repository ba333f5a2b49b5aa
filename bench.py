#!/usr/bin/env python3
"""
bench.py -- the hot path of BASELINE.json on MI355X: dense GP fit+predict in fp64
(kernel Gram build -> scaling + Gershgorin jitter -> blocked Cholesky -> triangular solves ->
posterior mean / covariance + log marginal likelihood), through the C-ABI of
liblsqfitgp_hip.so with every buffer resident in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5] [--no-cpu-baseline] [--no-sweep]

One "step" = one full pass of the path over one batch of synthetic input:
    c2  ExpQuad 1-D, N = 8192,  M = 100, y = sin(x/4) + noise, givencov 0.01 I
    c3  Matern-5/2, 3-D covariates, N = 32768, M = 100, noise 0.01 I      (default at --gpus 1)
    c4  ExpQuad 1-D, N = 65536, M = 100, noise 0.01 I                     (default at --gpus N > 1, sharded)
    c5  BART p = 10 (fp32 Gram) on the first 4096 rows of the reference's bart-data.txt, through the GP / BART / Chol
        host classes: predict the 81 held-out rows + marginal likelihood
With --gpus N > 1 the step is ONE C4 problem sharded by block columns over N ranks, one process per GPU
(lsqfitgp_amd._dist.DistributedChol: RCCL panel broadcasts over xGMI with look-ahead, "scaling": "strong"); N independent C3
replicas are timed first and reported under "replicas" (no data-path collective).  The ranks come from either side:
  * `python bench.py --gpus N` by itself: this process never touches the GPU, starts N fresh child processes
    (RANK / LOCAL_RANK / WORLD_SIZE in their environment), relays rank 0's JSON line and exits non-zero if a child does;
  * `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: the launcher's ranks are used as they are.
Rendezvous, barriers and the max-over-ranks of the timings go through lsqfitgp_amd/_rendezvous.py (standard-library
sockets on 127.0.0.1): bench.py imports no torch.

Prints ONE JSON line (rank 0): whole-job throughput in TFLOP/s over the algorithmic flop count
N^3/3 + N^2 (M+1) + N M^2, the roofline of the dominant kernel (syrk trailing update of the Cholesky,
timed live with HIP events on the stream it runs on) and a CPU baseline (the numpy/scipy oracle on bounded samples of the
same workloads, on this box's host cores).
"""

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6     # 256 CU x 4 SIMD x 2048 flop / 64 cycles x 2.4 GHz (v_mfma_f64_16x16x4_f64: 64 cycles
                                 # per instruction, measured by tools/f64_rate_probe.hip; equals the MI355X datasheet figure)

CONFIGS = {
    'c2': dict(name='C2 ExpQuad 1-D fp64 N=8192 M=100: Gram build + Cholesky + solve', kernel='expquad', d=1, n=8192, m=100, seed=2),
    'c3': dict(name='C3 Matern-5/2 3-D fp64 N=32768 M=100: Gram build + Cholesky + solve', kernel='matern52', d=3, n=32768, m=100, seed=3),
    'c4': dict(name='C4 ExpQuad 1-D fp64 N=65536 M=100: Gram build + Cholesky + solve', kernel='expquad', d=1, n=65536, m=100, seed=4),
    # (not a BASELINE configuration: a small second workload for rehearsing the multi-rank line on one GPU, tests/test_gpu_dist.py)
    'r4096': dict(name='rehearsal Matern-5/2 3-D fp64 N=4096 M=100', kernel='matern52', d=3, n=4096, m=100, seed=5),
}


def synth(cfg, n=None):
    """ synthetic inputs of SURVEY.md 8(d) """
    n = n or cfg['n']
    m = cfg['m']
    rng = np.random.default_rng(cfg['seed'])
    if cfg['kernel'] == 'expquad':
        hi = n / 16.                               # 16 points per lengthscale
        x = rng.uniform(0, hi, (1, n))
        xs = rng.uniform(0, hi, (1, m))
        y = np.sin(x[0] / 4) + 0.1 * rng.standard_normal(n)
    else:
        x = rng.uniform(0, 16, (3, n))
        xs = rng.uniform(0, 16, (3, m))
        y = np.sin(x).sum(axis=0) + 0.1 * rng.standard_normal(n)
    return x, xs, y


def flops_fit_predict(n, m):
    return n ** 3 / 3 + n ** 2 * (m + 1) + n * m ** 2


class Workload:
    """ one replica of the path on one GPU, buffers preallocated """

    def __init__(self, cfg, ctx):
        import lsqfitgp_amd as lgp
        from lsqfitgp_amd import _lib
        self.lgp, self._lib, self.lib, self.ctx, self.cfg = lgp, _lib, _lib.load(), ctx, cfg
        n, m, d = cfg['n'], cfg['m'], cfg['d']
        x, xs, y = synth(cfg)
        DA = lgp.DeviceArray
        self.x = DA.from_host(x, ctx=ctx)            # SoA d x N
        self.xs = DA.from_host(xs, ctx=ctx)
        self.y = DA.from_host(y.reshape(n, 1), ctx=ctx)
        self.fused = not os.environ.get('LGP_BENCH_SEPARATE_SOLVE')
        # fused Gram: lgp_gram_chol_factor_rhs writes tril((Kxx + ycov) / s / s') straight into the factor (K is never
        # materialised: one N x N allocation instead of two); LGP_BENCH_SEPARATE_GRAM=1 times the three-pass path instead
        self.fused_gram = self.fused and not os.environ.get('LGP_BENCH_SEPARATE_GRAM')
        # like GP.pred's one-call path (_GP.py, _pred_one_call): LGP_FACTOR_LOWER_ONLY -- the prediction reads the factor's lower triangle
        # only, so the Gram kernel stores no zeros above the diagonal (lgp_zero_upper writes them for a consumer that wants the full
        # matrix).  LGP_BENCH_FULL_FACTOR=1 times the step with the zeros stored: +0.46 ms at C3 (fused Gram kernel 1.62 against 1.16 ms)
        self.lower_only = 0 if os.environ.get('LGP_BENCH_FULL_FACTOR') else _lib.LGP_FACTOR_LOWER_ONLY
        self.K = None if self.fused_gram else DA.empty((n, n), ctx=ctx)           # Kxx + ycov
        # fused: the factor with m + 1 extra rows [Kxxs | y]^T that the factorisation turns into (L^-1 [Kxxs | y])^T
        # (m + 1 padded with zero rows to whole 128-row tiles, so that the extra tiles take the kernels' aligned fast path)
        self.nx = -(-(m + 1) // 128) * 128
        self.L = DA.empty((n + self.nx, n) if self.fused else (n, n), ctx=ctx)
        if self.fused:
            self.L.block(n, n + self.nx, 0, n).fill_zero()
        self.yrow = DA.from_host(y.reshape(1, n), ctx=ctx)
        self.dblk = DA.empty((self.lib.lgp_chol_dblk_size(n),), ctx=ctx)
        self.W = DA.empty((n, m + 1), ctx=ctx)       # separate-solve path: [Kxxs | y] -> L^-1 [Kxxs | y]
        self.Kss = DA.empty((m, m), ctx=ctx)
        self.mean = DA.empty((m,), ctx=ctx)
        self.cov = DA.empty((m, m), ctx=ctx)
        self.kind = _lib.LGP_KERNEL_EXPQUAD if cfg['kernel'] == 'expquad' else _lib.LGP_KERNEL_MATERNP
        self.p = 0 if cfg['kernel'] == 'expquad' else 2
        self.noise = 0.01
        self.results = {}

    def _gram(self, a, b, out):
        ck = self._lib.check
        d = self.cfg['d']
        ck(self.lib.lgp_gram_isotropic_f64(self.ctx.handle, self.kind, self.p, 0, None, d, a.ptr, a.ld, a.shape[1], b.ptr, b.ld, b.shape[1],
                                           0, 0., 1, 1., 0, None, None, int(a is b), out.ptr, out.ld), 'gram')

    def step(self, timers=None):
        """ Gram build, decomposition, posterior, log marginal likelihood -- all enqueued on the context's stream """
        lib, ctx, ck = self.lib, self.ctx, self._lib.check
        n, m = self.cfg['n'], self.cfg['m']
        ev = timers
        if ev:
            ctx.record(ev[0])
        # ---- Gram build (_GP/_elements.py:575-577): Kxx, Kxxs, Kxsxs; noise on the diagonal (_compute.py:85)
        if not self.fused_gram:
            self._gram(self.x, self.x, self.K)
            ck(lib.lgp_add_diag_f64(ctx.handle, n, self.K.ptr, self.K.ld, None, self.noise), 'add_diag')
        if self.fused:
            self._gram(self.xs, self.x, self.L.block(n, n + m, 0, n))        # Kxxs^T straight into the extra rows
            self.L.block(n + m, n + m + 1, 0, n).assign(self.yrow)
        else:
            self._gram(self.x, self.xs, self.W.block(0, n, 0, m))
            self.W.block(0, n, m, m + 1).assign(self.y)
        self._gram(self.xs, self.xs, self.Kss)
        if ev:
            ctx.record(ev[1])
        # ---- decomposition (_linalg/_decomp.py:380-393); fused: the forward solve of [Kxxs | y] rides along
        eps = ctypes.c_double()
        if self.fused_gram:
            info = ck(lib.lgp_gram_chol_factor_rhs_f64(ctx.handle, self.kind, self.p, 0, None, self.cfg['d'], self.x.ptr, self.x.ld, n, 0, 0., 1, 1., 0, None, None,
                                                       self.noise, None, 0., 0., self._lib.LGP_EPS_AUTO_REL | self.lower_only, self.L.ptr, self.L.ld, self.nx, self.dblk.ptr,
                                                       ctypes.byref(eps)), 'gram_chol_factor_rhs')
        elif self.fused:
            info = ck(lib.lgp_chol_factor_rhs_f64(ctx.handle, n, self.K.ptr, self.K.ld, 0., 0., self._lib.LGP_EPS_AUTO_REL,
                                                  self.L.ptr, self.L.ld, self.nx, self.dblk.ptr, ctypes.byref(eps)), 'chol_factor_rhs')
        else:
            info = ck(lib.lgp_chol_factor_f64(ctx.handle, n, self.K.ptr, self.K.ld, 0., 0., self._lib.LGP_EPS_AUTO_REL,
                                              self.L.ptr, self.L.ld, self.dblk.ptr, ctypes.byref(eps)), 'chol_factor')
        if info != 0:
            raise np.linalg.LinAlgError('cholesky decomposition not finite, probably matrix not pos def numerically')
        if ev:
            ctx.record(ev[2])
        # ---- posterior mean / covariance (_GP/_compute.py:259-260) and log marginal likelihood (:420-422)
        # one forward solve of [Kxxs | y] serves the mean, the covariance and the likelihood
        val = ctypes.c_double()
        if self.fused:
            Wt = self.L.block(n, n + m + 1, 0, n)
            ck(lib.lgp_posterior_solved_f64(ctx.handle, n, m, self.L.ptr, self.L.ld, Wt.ptr, Wt.ld,
                                            self.Kss.ptr, self.Kss.ld, self.mean.ptr, self.cov.ptr, self.cov.ld, ctypes.byref(val)), 'posterior')
        else:
            ck(lib.lgp_posterior_f64(ctx.handle, n, m, self.L.ptr, self.L.ld, self.dblk.ptr, self.W.ptr, self.W.ld,
                                     self.Kss.ptr, self.Kss.ld, self.mean.ptr, self.cov.ptr, self.cov.ld, ctypes.byref(val)), 'posterior')
        if ev:
            ctx.record(ev[3])
        self.results = dict(eps=eps.value, mll=-val.value)

    def posterior_host(self):
        return self.mean.to_host(), self.cov.to_host()


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _cpu_share():
    """ CPUs this process may really use: the affinity mask, cut by the cgroup CPU quota when there is one (a 1-GPU box is a
    16-CPU share of a 256-CPU host: 64 BLAS threads on it ran LAPACK potrf at 30 GFLOP/s in round 2) """
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith('cpu.max'):
                if parts[0] != 'max':
                    cpus = min(cpus, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as g:
                        cpus = min(cpus, max(1, int(q / int(g.read().split()[0]))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, cpus)


def _oracle_fit_predict(cfg, n):
    """ one fit+predict pass of the numpy / scipy oracle on the first n points of the workload, factoring ONCE (as the GPU step
    does): Gram build, Chol of Kxx + noise, posterior mean / covariance and log marginal likelihood from the same factor.
    Returns (seconds by phase, total seconds). """
    import oracle
    m = cfg['m']
    x, xs, y = synth(cfg, n)
    if cfg['kernel'] == 'expquad':
        k = oracle.ExpQuad()
        a, b = x[0], xs[0]
    else:
        k = oracle.Maternp(p=2, scale=1.)
        dt = [(f'f{i}', float) for i in range(cfg['d'])]
        a = np.empty(n, dt); b = np.empty(m, dt)
        for i, f in enumerate(a.dtype.names):
            a[f] = x[i]; b[f] = xs[i]
    t0 = time.perf_counter()
    Kxx = k(a[:, None], a[None, :])
    Kxx[np.diag_indices(n)] += 0.01
    Kxs = k(a[:, None], b[None, :])
    Kss = k(b[:, None], b[None, :])
    t1 = time.perf_counter()
    dec = oracle.Chol(Kxx)                                                   # _linalg/_decomp.py:380-393
    t2 = time.perf_counter()
    mean = dec.pinv_bilinear(Kxs, y)                                         # _GP/_compute.py:259
    cov = Kss - dec.ginv_quad(Kxs)                                           # :260
    mll = -dec.minus_log_normal_density(y, value=True)[0]                    # :420-422
    t3 = time.perf_counter()
    assert np.isfinite(mll) and mean.shape == (m,) and cov.shape == (m, m)
    return dict(gram=t1 - t0, decomposition=t2 - t1, solves=t3 - t2), t3 - t0


def _backend_fit_predict(be, cfg, n):
    """ the same pass through the C++ / OpenMP CPU backend (oracle/cpu_backend.cpp, the C-ABI of the HIP library on host
    arrays): Gram blocks, Chol.__init__ with [Kxxs | y] solved afterwards, posterior, log marginal likelihood """
    m = cfg['m']
    x, xs, y = synth(cfg, n)
    kind, p = (be.EXPQUAD, 0) if cfg['kernel'] == 'expquad' else (be.MATERNP, 2)
    t0 = time.perf_counter()
    Kxx = be.gram(kind, p, x, x)
    Kxx[np.diag_indices(n)] += 0.01
    W = np.empty((n, m + 1))
    W[:, :m] = be.gram(kind, p, x, xs)
    W[:, m] = y
    Kss = be.gram(kind, p, xs, xs)
    t1 = time.perf_counter()
    L, eps = be.chol(Kxx)
    t2 = time.perf_counter()
    mean, cov, mlnd = be.posterior(L, W, Kss)
    t3 = time.perf_counter()
    assert np.isfinite(mlnd) and mean.shape == (m,) and cov.shape == (m, m)
    return dict(gram=t1 - t0, decomposition=t2 - t1, solves=t3 - t2), t3 - t0, -mlnd


def cpu_baseline(main_cfg):
    """ CPU baseline on this box's host cores: the C++ / OpenMP backend behind the
    library's C-ABI (`kind` "port": a restatement of the reference's path, not the reference), one warm-up + best of N, with
    all the CPUs this process may use and with one; beside it the numpy / scipy oracle (LAPACK potrf / trsm, what JAX-CPU lowers
    to) with its BLAS threads capped to the same CPUs.  C2 at N = 8192, C3 at N = 8192 and at its full N = 32768 (best of 3), C5's
    Gram rows with the numpy oracle (256 of the 4096 rows timed, the full matrix extrapolated and labelled so).  About 80 s in all. """
    from threadpoolctl import threadpool_limits, threadpool_info
    from oracle import cpu as cpu_backend
    info = threadpool_info()
    cpus = _cpu_share()
    out = {}

    def stats(cfg, n, t, ph, threads, reps):
        return dict(N=n, M=cfg['m'], seconds=t, phases_s=ph, tflops=flops_fit_predict(n, cfg['m']) / t / 1e12,
                    cholesky_gflops=n ** 3 / 3 / ph['decomposition'] / 1e9, threads=threads, best_of=reps)

    def best_backend(cfg, n, threads, reps):
        be = cpu_backend.CpuBackend(threads)
        _backend_fit_predict(be, cfg, min(n, 1024))                          # warm-up (thread pool, page faults of the library)
        runs = [_backend_fit_predict(be, cfg, n) for _ in range(reps)]
        ph, t, mll = min(runs, key=lambda r: r[1])
        d = stats(cfg, n, t, ph, threads, reps)
        d['log_marginal_likelihood'] = mll
        return d

    def best_oracle(cfg, n, threads, reps):
        with threadpool_limits(limits=threads):
            _oracle_fit_predict(cfg, min(n, 1024))
            runs = [_oracle_fit_predict(cfg, n) for _ in range(reps)]
        ph, t = min(runs, key=lambda r: r[1])
        return stats(cfg, n, t, ph, threads, reps)

    for name in ('c2', 'c3'):
        cfg = CONFIGS[name]
        out[name] = {'all_cores': best_backend(cfg, 8192, cpus, 5), 'one_core': best_backend(cfg, 4096, 1, 3),
                     'numpy_scipy_oracle_all_cores': best_oracle(cfg, 8192, cpus, 2)}
    try:
        out['c3']['full_size_all_cores'] = best_backend(CONFIGS['c3'], CONFIGS['c3']['n'], cpus, 3)
    except MemoryError as exc:
        out['c3']['full_size_all_cores'] = {'error': repr(exc)}
    try:        # C5: rows of the BART Gram matrix of the real workload by the numpy oracle + the f64 factorisation by the backend
        import oracle
        from workloads import load_abalone, c5_prepare
        X, y = load_abalone()
        prep = c5_prepare(oracle, X, y, 4096)
        rows = 256
        t0 = time.perf_counter()
        with threadpool_limits(limits=cpus):
            blk = prep['kern'](prep['pts'][:rows, None], prep['pts'][None, :4096])
        tg = time.perf_counter() - t0
        be = cpu_backend.CpuBackend(cpus)
        K = np.eye(4096) * 5 + 0.5
        t0 = time.perf_counter(); be.chol(K); tc = time.perf_counter() - t0
        out['c5'] = {'bart_gram_pairs_per_s': rows * 4096 / tg, 'bart_gram_rows_timed': rows, 'bart_gram_full_matrix_s_extrapolated': tg * 4096 / rows,
                     'cholesky_4096_s': tc, 'threads': cpus,
                     'note': 'numpy oracle Gram rows (vectorised over pairs) + C++ backend Cholesky; the Gram time of the full 4096 x 4096 matrix is EXTRAPOLATED from '
                             f'the {rows} rows timed (the pairs are independent: linear in the rows)'}
        assert blk.shape == (rows, 4096)
    except Exception as exc:
        out['c5'] = {'error': repr(exc)}
    key = 'c2' if main_cfg['kernel'] == 'expquad' else 'c3'
    top = out[key].get('full_size_all_cores') if main_cfg['n'] == CONFIGS[key]['n'] and 'error' not in out[key].get('full_size_all_cores', {'error': 1}) \
        else out[key]['all_cores']
    return dict(value=top['tflops'], unit='TFLOP/s', cores=int(top['threads']), kind='port',
                sample=(f"{main_cfg['kernel']} d={main_cfg['d']} N={top['N']} M={top['M']} fit+predict with the C++/OpenMP CPU backend behind the library's C-ABI "
                        f"(oracle/cpu_backend.cpp), factored once, best of {top['best_of']} after a warm-up, {top['seconds']:.2f} s"),
                seconds=top['seconds'], cholesky_gflops=top['cholesky_gflops'], host_cpus=os.cpu_count(), cpus_usable=cpus, cpu_model=_cpu_model(),
                blas=[p.get('internal_api') for p in info],
                one_core={'value': out[key]['one_core']['tflops'], 'N': out[key]['one_core']['N'], 'seconds': out[key]['one_core']['seconds'],
                          'cholesky_gflops': out[key]['one_core']['cholesky_gflops']},
                by_config=out)


def size_sweep(ctx, main_cfg):
    """ "vs N" part of BASELINE.json's metric: the same step at smaller sizes and on configuration C2, 1 warmup + 3 timed
    steps each (6 up to N = 16384; a fraction of a second in all); not part of `value` """
    out = []
    cases = [dict(CONFIGS['c2'])]
    if main_cfg['n'] < CONFIGS['c4']['n']:
        cases.append(dict(CONFIGS['c4']))           # N = 65536 on this one GPU (2 x 32 GiB)
    for n in (4096, 8192, 16384):
        if n < main_cfg['n']:
            c = dict(main_cfg)
            c['n'] = n
            c['name'] = f"{main_cfg['kernel']} {main_cfg['d']}-D fp64 N={n} M={main_cfg['m']}"
            cases.append(c)
    for c in cases:
        wl = Workload(c, ctx)
        wl.step()
        ctx.sync()
        reps = (6 if c['n'] <= 16384 else 3) if c['n'] <= 32768 else 2
        evs = [[ctx.event() for _ in range(4)] for _ in range(reps)]
        t0 = time.perf_counter()
        for k in range(reps):
            wl.step(evs[k])
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        phs = np.array([[ctx.elapsed_ms(e[i], e[i + 1]) for i in range(3)] for e in evs])
        ph = phs.mean(axis=0)
        out.append({'workload': c['name'], 'N': c['n'], 'ms_per_step': dt * 1e3, 'value': flops_fit_predict(c['n'], c['m']) / dt / 1e12,
                    'cholesky_ms': float(ph[1]), 'cholesky_tflops': (c['n'] ** 3 / 3) / (ph[1] * 1e-3) / 1e12,
                    'cholesky_min_ms': float(phs[:, 1].min()), 'timed_steps': reps,
                    'cholesky_phase': 'fused Gram build of Kxx + regularisation + factorisation with the carried solve, HIP events, mean (and min) over the timed steps'})
        del wl
    try:
        out.append(c5_step(ctx, steps=3, warmup=1))
        out[-1]['full_fit'] = c5_fit(ctx)
    except Exception as exc:                      # the sweep is informative: never let it take the headline number down
        out.append({'workload': 'C5', 'error': repr(exc)})
    return out


def default_path(ctx, cfg_name='c2', reps=3):
    """ the same fit+predict through the drop-in host classes with their defaults (GP.predfromdata + marginal_likelihood,
    reference call sequence, dense givencov uploaded by the caller) next to the C-ABI step that `value` is measured on """
    import lsqfitgp_amd as lgp
    cfg = CONFIGS[cfg_name]
    n, m = cfg['n'], cfg['m']
    x, xs, y = synth(cfg)
    kern = lgp.ExpQuad() if cfg['kernel'] == 'expquad' else lgp.Maternp(p=2, scale=1.)
    if cfg['d'] == 1:
        a, b = x[0], xs[0]
    else:
        dt = [(f'f{i}', float) for i in range(cfg['d'])]
        a = np.empty(n, dt); b = np.empty(m, dt)
        for i, f in enumerate(a.dtype.names):
            a[f] = x[i]; b[f] = xs[i]
    gc = {('data', 'data'): lgp.diag(np.full(n, 0.01))}
    out = {}
    for label, fuse in (('carried_solve_default', True), ('separate_solve', False)):
        def step():
            gp = lgp.GP(kern, checkpos=False, checksym=False, fuse_solve=fuse, ctx=ctx).addx(a, 'data').addx(b, 'test')
            mean, cov = gp.predfromdata({'data': y}, 'test', gc, raw=True)
            return gp.marginal_likelihood({'data': y}, gc)
        step(); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            mll = step()
        ctx.sync()
        out[label + '_ms'] = (time.perf_counter() - t0) / reps * 1e3
    out['workload'] = (cfg['name'] + ' via GP.predfromdata + GP.marginal_likelihood with the same givencov object: the reference call sequence; '
                       'the decomposition of the first call serves the second (GP(reuse_givencov=True), one factorisation where the reference does two)')
    out['log_marginal_likelihood'] = float(mll)
    return out


def small_n_latency(ctx):
    """ the only numbers the reference publishes (docs/userguide/optim.rst:88-159, "unspecified laptop"): `doinference` =
    GP(ExpQuad()).addx(x, 'data').addx(xplot, 'plot').predfromdata({'data': data}, 'plot', raw=True) + sqrt(diag(cov)), data =
    zeros(N), x = linspace(0, 10, N), 100 plot points, N = 10 and 1000; there: 7.1 / 33.2 ms eager, 16.0 ms at N = 1000 without the
    checks, 0.041 / 4.8 ms jitted (where the checks are off) -- AVERAGES over repetitions (IPython %timeit).  Here through the host
    classes, wall clock over 20 calls: the MEAN is what stands beside the reference's figures, the minimum is given too. """
    import lsqfitgp_amd as lgp

    def doinference(data, **options):
        x = np.linspace(0, 10, len(data))
        xplot = np.linspace(0, 10, 100)
        mean, cov = lgp.GP(lgp.ExpQuad(), ctx=ctx, **options).addx(x, 'data').addx(xplot, 'plot').predfromdata({'data': data}, 'plot', raw=True)
        return mean, np.sqrt(np.diag(cov))

    def best(n, **options):
        data = np.zeros(n)
        doinference(data, **options)
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            doinference(data, **options)
            ts.append(time.perf_counter() - t0)
        return float(np.mean(ts)) * 1e3, min(ts) * 1e3

    out = {'workload': 'docs/userguide/optim.rst:88-159 doinference (ExpQuad, N data points on [0, 10], 100 plot points, raw=True), wall clock, mean of 20 calls (like the reference\'s %timeit averages; min beside it)',
           'reference_published_ms': {'N=10 eager': 7.102, 'N=10 jit': 0.041, 'N=1000 eager': 33.204, 'N=1000 jit': 4.804,
                                      'N=1000 eager, checkpos=False, checksym=False': 16.029, 'N=1000 jit, checks off': 5.010,
                                      'hardware': 'unspecified laptop CPU (the reference states none)'}}
    for n in (10, 1000):
        out[f'N={n} default options ms'], out[f'N={n} default options min ms'] = best(n)
        out[f'N={n} checkpos=False, checksym=False ms'], out[f'N={n} checkpos=False, checksym=False min ms'] = best(n, checkpos=False, checksym=False)
    return out


def c5_step(ctx, steps=3, warmup=1):
    """ configuration C5 on its real workload: the GP that bayestree.bart builds (``bayestree/_bart.py:187-205``) at the fixed
    hyperparameters of SURVEY.md 8(d) on the first 4096 rows of the reference's examples/bart-data.txt (p = 10, fp32 kernel,
    81 held-out rows), driven through the host classes as a user of the reference would: one step = new GP ->
    predfromdata('testmean') + marginal_likelihood(epsrel=0).  Roofline object: the BART Gram kernel. """
    import lsqfitgp_amd as lgp
    from workloads import load_abalone, c5_prepare, c5_gp
    X, y = load_abalone()
    n = 4096
    prep = c5_prepare(lgp, X, y, n)                  # data preparation (splits, bin indices): once, like bayestree.bart does
    given, kern, pts = prep['given'], prep['kern'], prep['pts']

    def step():
        gp = c5_gp(lgp, prep, diag=lgp.diag)         # a new GP per step: every block is rebuilt on the device
        mean, cov = gp.predfromdata(given, 'testmean', raw=True)
        return mean, cov, gp.marginal_likelihood(given, epsrel=0), kern, pts

    for _ in range(warmup):
        step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    kern, pts = res[3], res[4]
    # the BART Gram kernel alone, N x N fp32, HIP events on the context's stream: bin indices uploaded once (kernel.prepare, what
    # the callers that rebuild the matrix per hyperparameter step do), output preallocated
    e0, e1 = ctx.event(), ctx.event()
    dp = kern.prepare(pts[:n], ctx=ctx)
    gout = lgp.DeviceArray.empty((n, n), np.float32, ctx)
    kern.gram(dp, dp, ctx=ctx, out=gout)
    best = 1e9
    for _ in range(5):
        ctx.record(e0)
        kern.gram(dp, dp, ctx=ctx, out=gout)
        ctx.record(e1)
        best = min(best, ctx.elapsed_ms(e0, e1))
    nt = len(pts) - n
    return {'workload': f'C5 BART kernel p=10 fp32 on bart-data.txt, N={n} train / {nt} held out, via GP/BART/Chol host classes',
            'N': n, 'M': nt, 'ms_per_step': dt * 1e3, 'value': flops_fit_predict(n, nt) / dt / 1e12,
            'log_marginal_likelihood': float(res[2]),
            'bart_gram': bart_roofline(n, best)}


VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 4          # wave64 vector instructions per ns: one per 4 cycles per SIMD, 4 SIMDs x 256 CUs x 2.4 GHz


def bart_roofline(n, ms):
    """ the BART Gram kernel is bound by vector-ALU ISSUE, not by HBM (4 bytes written per pair = 0.17 TB/s): its roofline is
    the wave64 VALU instruction rate.  Instructions per pair from the committed counter pass (profiles/rNN_c5_bart_valu.json, the newest:
    SQ_INSTS_VALU of one launch / pairs, p = 10 axes x 5 iterations of the depth-2 closed form, fp32 without contraction);
    achieved = that count x pairs of this launch / its duration measured live """
    per_pair = source = None
    for tag in ('r05', 'r04', 'r03'):                        # the newest committed counter pass (tools/pmc_bart.sh)
        try:
            with open(os.path.join(ROOT, 'profiles', f'{tag}_c5_bart_valu.json')) as f:
                d = json.load(f)
            per_pair = d['SQ_INSTS_VALU'] / d['pairs']       # wave instructions per pair (each covers 64 pairs' worth of lanes)
            source = f'profiles/{tag}_c5_bart_valu.json'
            break
        except Exception:
            continue
    out = {'kernel': 'gram_bart_kernel<float> (N x N)', 'ms': ms, 'gpairs_per_s': n * n / (ms * 1e-3) / 1e9,
           'hbm_written_GB_per_s': 4. * n * n / (ms * 1e-3) / 1e9,
           'note': 'bin indices uploaded once (kernel.prepare), output preallocated; includes the per-call upload of the per-axis constants'}
    if per_pair is not None:
        ach = per_pair * n * n / (ms * 1e-3) / 1e9
        out.update({'bound': 'valu', 'achieved': ach, 'peak': VALU_PEAK_GINSTR, 'unit': 'G wave-instructions/s', 'frac': ach / VALU_PEAK_GINSTR,
                    'valu_lane_instructions_per_pair': per_pair * 64, 'instruction_count_source': source})
    return out


def c5_fit(ctx):
    """ the caller of configuration C5 end to end: ``bayestree.bart`` on the first 4096 rows of bart-data.txt (float64, the
    reference's default dtype), i.e. the whole L-BFGS hyperparameter fit with every evaluation (BART Gram + alpha / beta
    tangents, factorisation, likelihood gradient) on the device, then the prediction at the 81 held-out rows """
    import lsqfitgp_amd as lgp
    from workloads import load_abalone
    X, y = load_abalone()
    X = X.astype(float); y = y.astype(float)
    n = 4096
    t0 = time.perf_counter()
    b = lgp.bayestree.bart(X[:n], y[:n], fitkw=dict(verbosity=0), ctx=ctx)
    t1 = time.perf_counter()
    m, c = b.pred(x_test=X[n:], error=True)
    t2 = time.perf_counter()
    evals = b.fit._calls['grad'] + b.fit._calls['fun']
    return {'workload': f'bayestree.bart(x[:{n}], y[:{n}]) on bart-data.txt, f64: L-BFGS fit of (alpha, beta, k, sigma2) + prediction of {len(y) - n} held-out rows',
            'fit_s': t1 - t0, 'iterations': int(b.fit.minresult.nit), 'evaluations': int(evals), 'ms_per_evaluation': (t1 - t0) / max(evals, 1) * 1e3,
            'ms_per_iteration': (t1 - t0) / max(int(b.fit.minresult.nit), 1) * 1e3, 'predict_ms': (t2 - t1) * 1e3,
            'map': {k: float(v) for k, v in b.fit.pmean.items()}, 'heldout_rmse': float(np.sqrt(np.mean((m - y[n:]) ** 2))),
            'heldout_sdev': float(np.std(y[n:]))}


def main_c5(args, ctx):
    r = c5_step(ctx, steps=args.steps, warmup=args.warmup)
    try:
        fit = c5_fit(ctx)
    except Exception as exc:
        fit = {'error': repr(exc)}
    g = r['bart_gram']
    print(json.dumps({
        'metric': 'GP fit+predict throughput (Gram build + Cholesky + solves; flops N^3/3 + N^2(M+1) + N M^2)',
        'value': r['value'], 'unit': 'TFLOP/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': r['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32 Gram, f64 assembly and solve (as the reference: zero cross blocks are float64)', 'data': 'reference data file (examples/bart-data.txt)',
        'config': {'workload': r['workload'], 'N': r['N'], 'M': r['M'], 'p': 10, 'parallelism': 'single GPU'},
        'log_marginal_likelihood': r['log_marginal_likelihood'],
        'roofline': {'kernel': g['kernel'], 'bound': g.get('bound'), 'achieved': g.get('achieved'), 'peak': g.get('peak'), 'unit': g.get('unit'),
                     'frac': g.get('frac'), 'traffic': None, 'avg_launch_ms': g['ms'], 'gpairs_per_s': g['gpairs_per_s'],
                     'hbm_written_GB_per_s': g['hbm_written_GB_per_s'], 'valu_lane_instructions_per_pair': g.get('valu_lane_instructions_per_pair')},
        'full_fit': fit,
        'cpu_baseline': None}), flush=True)


def pmc_traffic(config):
    """ HBM bytes per trailing-update launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE
    x2 per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE, collected in their own runs of this same
    command); counters cannot be read live from inside the process, so None when no committed pass matches """
    for tag in ('r05', 'r04', 'r03', 'r02', 'r01'):          # the newest committed pass that holds numbers
        path = os.path.join(ROOT, 'profiles', f'{tag}_{config}_pmc_traffic.json')
        try:
            with open(path) as f:
                d = json.load(f)
            return d['hbm_read_bytes_per_launch_corrected'] + d['hbm_write_bytes_per_launch'], os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def init_transport(ctx, rdzv, rank, world):
    """ RCCL communicator over the ranks (one GPU each), or -- LGP_DIST_TRANSPORT=host -- panels staged through host memory and
    the rendezvous sockets (lets several ranks rehearse the sharded path on ONE GPU).  Every rank learns whether every rank
    succeeded, so that a failure can fall back to replicas instead of hanging in a collective.  Returns (ok, message). """
    import lsqfitgp_amd as lgp
    err = ''
    if os.environ.get('LGP_DIST_TRANSPORT') == 'host':
        try:
            ctx.comm_init_host(rank, world, rdzv.collective)
        except Exception as exc:
            err = repr(exc)
    else:
        # every rank loads librccl first (lgp_comm_unique_id does): a missing library must not leave the others blocked inside
        # ncclCommInitRank
        uid = b''
        try:
            uid = lgp.Context.comm_unique_id()
        except Exception as exc:
            err = repr(exc)
        loaded = rdzv.allreduce(0. if err else 1., 'min') > 0
        uid = rdzv.broadcast_bytes(uid, 0)
        if loaded:
            try:
                ctx.comm_init(rank, world, uid)
            except Exception as exc:
                err = repr(exc)
        elif not err:
            err = 'another rank could not load librccl'
    ok = rdzv.allreduce(0. if err else 1., 'min') > 0
    return ok, err


def dist_step_factory(cfg, ctx):
    """ one step of the sharded path: Gram build of the owned block columns + regularisation + look-ahead factorisation with
    [Kxxs | y] carried through it + posterior mean / covariance / log marginal likelihood (one all-reduce) """
    import lsqfitgp_amd as lgp
    from lsqfitgp_amd._dist import DistributedChol
    x, xs, y = synth(cfg)
    kern = lgp.ExpQuad() if cfg['kernel'] == 'expquad' else lgp.Maternp(p=2, scale=1.)
    if cfg['d'] == 1:
        pts, tpts = x[0], xs[0]
    else:
        dt = [(f'f{i}', float) for i in range(cfg['d'])]
        pts = np.empty(cfg['n'], dt); tpts = np.empty(cfg['m'], dt)
        for i, f in enumerate(pts.dtype.names):
            pts[f] = x[i]; tpts[f] = xs[i]

    def step():
        d = DistributedChol(kern, pts, noise=0.01, nb=1024, ctx=ctx, carry=(tpts, y))
        mean, cov, mlnd = d.posterior(kern, pts, tpts, y)
        return -mlnd
    return step


def timed(step, args, sync, rdzv):
    """ W untimed + exactly K timed steps, bracketed by barrier + device sync on both sides; max over the ranks """
    for _ in range(args.warmup):
        step()
    sync(); rdzv.barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync(); rdzv.barrier(); sync()
    return rdzv.allreduce(time.perf_counter() - t0, 'max'), out


METRIC_SHARDED = 'GP fit+predict fp64 throughput, one problem sharded over the GPUs (flops N^3/3 + N^2(M+1) + N M^2)'
METRIC = 'GP fit+predict fp64 throughput (Gram build + Cholesky + solves; flops N^3/3 + N^2(M+1) + N M^2)'


XGMI_LINK_GBS = 153.        # MI355X_MICROARCH.md: 7 xGMI links per GPU, ~153 GB/s each, point to point (no switch)


def sharded_measure(cfg, ctx, args, rdzv, world):
    """ the timed sharded steps, then ONE more step with the library's inside view switched on (lgp_dist_stats: timing event pairs
    on the compute stream, kept out of the timed region): every rank's wall / wait / pack times and the bytes its updates read
    from other ranks, gathered on every rank """
    step = dist_step_factory(cfg, ctx)
    elapsed, mll = timed(step, args, ctx.sync, rdzv)
    ctx.dist_stats(True)
    step()
    ctx.sync()
    st = ctx.dist_stats(False)
    per_rank = [json.loads(b.decode()) for b in rdzv.allgather_bytes(json.dumps(st).encode())]
    return dict(elapsed=elapsed, mll=mll, per_rank=per_rank)


def xgmi_roofline(cfg, res, world, steps):
    """ the sharded line's roofline: the panel traffic against the xGMI links.  achieved = the panel bytes a rank's updates read
    from other ranks / the time its compute stream stood idle waiting for them (the slowest rank's), peak = one link (153 GB/s)
    for the transport's broadcast, world - 1 links for scatter + all-gather; `link_bound_ms` = those bytes over the peak: what the
    step would take if nothing but the links bound it.  frac > 1 means the panels arrive faster than the updates ask for
    them at that peak, i.e. the links do not bound the step """
    pr = res['per_rank']
    worst = max(pr, key=lambda d: d['wait_ms'])
    algo = worst['bcast_algo']
    links = (world - 1) if algo == 'scatter + all-gather' else 1
    peak = XGMI_LINK_GBS * links
    step_ms = res['elapsed'] / steps * 1e3
    achieved = worst['bytes_from_peers'] / (worst['wait_ms'] * 1e-3) / 1e9 if worst['wait_ms'] > 0 else None
    return {'bound': 'xgmi', 'achieved': achieved, 'peak': peak, 'unit': 'GB/s', 'frac': achieved / peak if achieved else None, 'traffic': None,
            'bcast_algo': algo, 'links_used_per_gpu': links, 'bytes_from_peers_per_rank_per_step': worst['bytes_from_peers'],
            'link_bound_ms': worst['bytes_from_peers'] / (peak * 1e9) * 1e3,
            'per_rank_ms': [{'rank': r, 'factorisation_wall': d['wall_ms'], 'waiting_for_pieces': d['wait_ms'], 'pack': d['pack_ms'],
                             'compute_and_the_rest': d['wall_ms'] - d['wait_ms']} for r, d in enumerate(pr)],
            'step_ms': step_ms, 'wait_frac_of_step_worst_rank': worst['wait_ms'] / step_ms if step_ms > 0 else None,
            'owner_via_comm_stream': bool(worst['owner_via_comm_stream']),
            'note': 'measured on one extra, untimed step with timing events on the compute stream (lgp_dist_stats); compute-bound when the waits are a small '
                    'fraction of the step, link-bound when step_ms approaches link_bound_ms, chain-bound (the slab pipeline) when it waits although link_bound_ms is small'}


def main_multi(args, ctx, rdzv, rank, world):
    """ --gpus N > 1: N independent C3 replicas first (no data-path collective: always measurable), then BASELINE configuration C4
    (or --config) as ONE problem sharded by block columns over the N GPUs -- the headline line when it runs -- and C3 sharded the
    same way (strong scaling from the driver's own N = 1 number).  RCCL with more than one rank has never run on the builder's
    one-GPU boxes, so the attempts are ordered by how well-trodden their RCCL path is, each under a watchdog that prints the best
    line there is so far: replicas -> sharded with plain ncclBroadcast -> sharded with scatter + all-gather (grouped
    ncclSend/ncclRecv; first one checked broadcast: on an error or wrong contents every rank drops back to ncclBroadcast). """
    import threading
    import lsqfitgp_amd as lgp
    lgp.Context._default = ctx
    cfg = CONFIGS[args.config if args.config_given else 'c4']
    # replicas: every rank runs the single-GPU step on its own GPU (C3; --replica-config for rehearsals on a shared GPU)
    rcfg = CONFIGS[args.replica_config]
    wl = Workload(rcfg, ctx)
    relapsed, _ = timed(lambda: wl.step(), args, ctx.sync, rdzv)
    del wl
    replicas = {'workload': rcfg['name'], 'scaling': 'weak', 'value': flops_fit_predict(rcfg['n'], rcfg['m']) * args.steps * world / relapsed / 1e12,
                'unit': 'TFLOP/s', 'ms_per_step': relapsed / args.steps * 1e3, 'parallelism': f'{world} independent replicas, one per GPU (no collective)'}

    def replicas_line(err):
        return {'metric': METRIC, 'value': replicas['value'], 'unit': 'TFLOP/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': replicas['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
                'data': 'synthetic', 'config': {'workload': rcfg['name'], 'N': rcfg['n'], 'M': rcfg['m'], 'parallelism': replicas['parallelism']},
                'roofline': None, 'cpu_baseline': None, 'rccl_ranks': None, 'sharded_path_error': err}

    comm = (rank, 1, 'none')

    def sharded_line(c, res, extra):
        n, m = c['n'], c['m']
        transport = {'host': 'host-staged (rendezvous sockets)', 'rccl': 'RCCL over xGMI'}.get(comm[2], comm[2])
        roof = xgmi_roofline(c, res, world, args.steps)
        out = {'metric': METRIC_SHARDED,
               'value': flops_fit_predict(n, m) * args.steps / res['elapsed'] / 1e12, 'unit': 'TFLOP/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': res['elapsed'] / args.steps * 1e3,
               'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
               'config': {'workload': c['name'], 'N': n, 'M': m,
                          'parallelism': f'1-D block-column-cyclic Cholesky over {world} GPUs, nb=1024, {transport} panel broadcast ({roof["bcast_algo"]}) with look-ahead (lgp_chol_factor_dist)'},
               # what the communicator itself reports (ncclCommCount), None when the panels did not travel by RCCL
               'rccl_ranks': comm[1] if comm[2] == 'rccl' else None, 'transport': comm[2], 'bcast_algo': roof['bcast_algo'],
               'log_marginal_likelihood': res['mll'], 'roofline': roof, 'cpu_baseline': None, 'replicas': replicas}
        out.update(extra)
        return out

    # one line, printed once: by the main thread at the end or by a watchdog with the best there is so far
    state = {'line': replicas_line('the sharded path was not reached'), 'code': 3, 'printed': False, 'stage': 'communicator set-up'}
    lock = threading.Lock()

    def emit():
        with lock:
            if state['printed']:
                return False
            state['printed'] = True
            if rank == 0:
                print(json.dumps(state['line']), flush=True)
            return True

    # (a sharded C4 step is under a second on 2-8 GPUs; communicator set-up tens of seconds at most: a stalled attempt should not
    #  hold the line back for long)
    limit = float(os.environ.get('LGP_BENCH_SHARDED_TIMEOUT', 120 + 10 * (args.steps + args.warmup)))

    def give_up():
        with lock:
            if not state['printed']:
                state['line']['sharded_path_error'] = f'{state["stage"]} did not finish within {limit:.0f} s (communicator or schedule stalled)'
        emit()
        os._exit(state['code'])   # a process that has touched the GPU and is stuck in a collective: leave; non-zero unless a sharded line is out

    def arm():
        d = threading.Timer(limit, give_up)
        d.daemon = True
        d.start()
        return d

    dog = arm()
    ok, err = init_transport(ctx, rdzv, rank, world)
    errors = {}
    if ok:
        comm = ctx.comm_info()
        want = os.environ.get('LGP_BENCH_BCAST', 'both' if world >= 3 else 'plain')       # plain | scatter | both
        attempts = {'plain': ['plain'], 'scatter': ['scatter'], 'both': ['plain', 'scatter']}[want]
        results = {}
        for algo in attempts:
            dog.cancel(); dog = arm()
            with lock:
                state['stage'] = f'the sharded step with bcast_algo={algo}'
            try:
                ctx.comm_set_bcast(algo, 1 << 20)
                if algo == 'scatter':
                    # first use of grouped ncclSend / ncclRecv between these ranks: one checked broadcast; every rank must agree
                    bad = ctx.bcast_check(8 << 20, 0)
                    if rdzv.allreduce(1. if bad else 0., 'max') > 0:
                        errors[algo] = bad or 'another rank failed the checked scatter + all-gather broadcast'
                        ctx.comm_set_bcast('plain', 1 << 20)
                        continue
                res = sharded_measure(cfg, ctx, args, rdzv, world)
                results[algo] = res
                best = min(results, key=lambda a: results[a]['elapsed'])
                extra = {'bcast_attempts_ms_per_step': {a: r['elapsed'] / args.steps * 1e3 for a, r in results.items()}}
                if errors:
                    extra['sharded_path_error'] = '; '.join(f'{a}: {e}' for a, e in errors.items())
                with lock:
                    state['line'], state['code'] = sharded_line(cfg, results[best], extra), 0
            except Exception as exc:             # (a rank that fails here leaves the others in a collective: reported, not hidden)
                errors[algo] = repr(exc)
                break
        strong = os.environ.get('LGP_BENCH_STRONG_CONFIG', '' if args.config_given else 'c3')
        if results and strong in CONFIGS and cfg is not CONFIGS[strong]:
            # C3 sharded: the strong-scaling curve whose N = 1 point is the single-GPU bench line
            dog.cancel(); dog = arm()
            with lock:
                state['stage'] = 'the sharded C3 step'
            try:
                best = min(results, key=lambda a: results[a]['elapsed'])
                ctx.comm_set_bcast(best, 1 << 20)
                r3 = sharded_measure(CONFIGS[strong], ctx, args, rdzv, world)
                l3 = sharded_line(CONFIGS[strong], r3, {})
                with lock:
                    state['line']['sharded_c3'] = {k: l3[k] for k in ('value', 'unit', 'ms_per_step', 'scaling', 'config', 'bcast_algo', 'log_marginal_likelihood', 'roofline')}
            except Exception as exc:
                errors['c3'] = repr(exc)
    else:
        errors['communicator'] = err
    with lock:
        if errors and not state['printed']:
            state['line']['sharded_path_error'] = '; '.join(f'{a}: {e}' for a, e in errors.items())
            if state['code'] != 0:
                state['line'] = replicas_line(state['line']['sharded_path_error'])
    dog.cancel()                 # (before the line goes out: a watchdog that fires now must not print a second one)
    emit()
    code = state['code']

    # the teardown under a watchdog of its own: a rank stuck in the last barrier must not hang the job (the line is out)
    dog = threading.Timer(60., lambda: os._exit(code))
    dog.daemon = True
    dog.start()
    try:
        rdzv.barrier()
        rdzv.close()
    except Exception:
        pass
    dog.cancel()
    return code


def main_stub(args, rdzv, rank, world):
    """ LGP_BENCH_STUB=1: the launcher, the rendezvous and the timing protocol with a step that does not touch the GPU (the CPU test
    of `--gpus N`; tests/test_bench_launcher.py) """
    elapsed, _ = timed(lambda: time.sleep(0.01), args, lambda: None, rdzv)
    pids = rdzv.allgather_bytes(str(os.getpid()).encode())
    if rank == 0:
        print(json.dumps({'metric': 'stub step (no GPU work): launcher + rendezvous + timing protocol', 'value': args.steps * world / elapsed, 'unit': 'steps/s',
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
                          'scaling': 'weak', 'stub': True, 'rank_pids': [int(p) for p in pids], 'local_ranks_seen': world}), flush=True)
    rdzv.barrier()
    rdzv.close()
    return 0


def launch_ranks(args):
    """ `python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here.  This process makes no GPU call
    (it imports nothing of the package), so the children are ordinary fresh processes, one per GPU; rank 0's standard output is
    relayed, the others' goes to standard error; the exit status is non-zero if any rank's is, and a rank that outlives
    LGP_BENCH_LAUNCH_TIMEOUT seconds (default 1500) is ended by its exact pid. """
    import socket
    import subprocess
    import threading
    import secrets
    n = args.gpus
    # no port is picked here (a port that was free a moment ago may be taken when rank 0 binds it): rank 0 binds an ephemeral one
    # and publishes it in a private file named after this process and the job name below (lsqfitgp_amd/_rendezvous.py)
    job = str(20000 + secrets.randbelow(40000))
    token = secrets.token_hex(16)
    procs = []
    for r in range(n):
        env = {k: v for k, v in os.environ.items() if k != 'LGP_RDZV_PORT'}
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=job,
                   LGP_RDZV_TOKEN=token)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: what RCCL needs between processes on this image
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    deadline = time.monotonic() + float(os.environ.get('LGP_BENCH_LAUNCH_TIMEOUT', 1500))
    grace = float(os.environ.get('LGP_BENCH_LAUNCH_GRACE', 90))
    failed_at = None
    codes = [None] * n
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        now = time.monotonic()
        if failed_at is None and any(c not in (None, 0) for c in codes):
            failed_at = now            # a rank has failed: the others get `grace` seconds (their own watchdogs print the fallback line)
        if now > deadline or (failed_at is not None and now - failed_at > grace):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.kill()           # the exact process this launcher started
                    codes[i] = p.wait()
            break
        time.sleep(0.1)
    th.join(timeout=10)
    bad = [c for c in codes if c != 0]
    if bad:
        sys.stderr.write(f'bench.py: rank exit codes {codes}\n')
        return bad[0] if bad[0] > 0 else 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', choices=sorted(CONFIGS) + ['c5'], default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-sweep', action='store_true', help='skip the smaller-N / C2 / C4 / C5 runs reported under "vs_N"')
    ap.add_argument('--dist', action='store_true', default=os.environ.get('LGP_BENCH_DIST') == '1',
                    help='single process: run the sharded code path with one rank (the multi-rank default of --gpus N > 1)')
    ap.add_argument('--replica-config', choices=sorted(CONFIGS), default='c3',
                    help='--gpus N > 1: the single-GPU workload every rank runs as an independent replica (default c3)')
    args = ap.parse_args()
    args.config_given = args.config is not None
    if args.config is None:
        args.config = 'c3'

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(args)           # no launcher around us: start the ranks; this process never touches the GPU

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != max(args.gpus, 1) and world > 1:
        sys.stderr.write(f'bench.py: --gpus {args.gpus} but the launcher started {world} ranks: reporting n_gpus = {world}\n')
    rdzv = None
    if os.environ.get('LGP_BENCH_STUB') == '1' and os.environ.get('LGP_BENCH_STUB_FAIL_RANK') == str(rank):
        return 7                            # (test of the launcher: a rank that dies before the rendezvous)
    if world > 1 or args.dist:
        from lsqfitgp_amd._rendezvous import Rendezvous       # standard-library sockets; before any GPU call
        rdzv = Rendezvous(rank, world)
        if os.environ.get('LGP_BENCH_STUB') == '1':
            return main_stub(args, rdzv, rank, world)

    import lsqfitgp_amd as lgp              # fails loudly without the HIP library / a GPU
    device = os.environ.get('LGP_DEVICE')
    if device is None:
        device = local_rank
        if os.environ.get('LGP_DIST_TRANSPORT') == 'host':      # rehearsal: more ranks than GPUs share what there is
            cnt = ctypes.c_int()
            from lsqfitgp_amd import _lib as _l
            _l.load().lgp_device_count(ctypes.byref(cnt))
            device = local_rank % max(cnt.value, 1)
    ctx = lgp.Context(int(device))
    if args.config == 'c5':
        lgp.Context._default = ctx
        return main_c5(args, ctx)
    if rdzv is not None:
        return main_multi(args, ctx, rdzv, rank, world)
    cfg = CONFIGS[args.config]
    lgp.Context._default = ctx
    wl = Workload(cfg, ctx)
    lib, _lib = wl.lib, wl._lib

    for _ in range(args.warmup):
        wl.step()
    ctx.sync()

    # ---- timed region: exactly K steps
    evs = [[ctx.event() for _ in range(4)] for _ in range(args.steps)]
    _lib.check(lib.lgp_profile_begin(ctx.handle))
    t0 = time.perf_counter()
    for k in range(args.steps):
        wl.step(evs[k])
    ctx.sync()
    t1 = time.perf_counter()
    launches = ctypes.c_size_t(); tot_ms = ctypes.c_double(); tot_fl = ctypes.c_double(); max_fl = ctypes.c_double(); max_ms = ctypes.c_double()
    _lib.check(lib.lgp_profile_end(ctx.handle, ctypes.byref(launches), ctypes.byref(tot_ms), ctypes.byref(tot_fl), ctypes.byref(max_fl), ctypes.byref(max_ms)))
    elapsed = t1 - t0

    n, m = cfg['n'], cfg['m']
    ms_per_step = elapsed / args.steps * 1e3
    total_flops = flops_fit_predict(n, m) * args.steps
    phases = np.array([[ctx.elapsed_ms(e[i], e[i + 1]) for i in range(3)] for e in evs]).mean(axis=0)
    achieved = tot_fl.value / (tot_ms.value * 1e-3) / 1e12 if tot_ms.value > 0 else None
    out = {
        'metric': METRIC,
        'value': total_flops / elapsed / 1e12,
        'unit': 'TFLOP/s',
        'n_gpus': 1,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': ms_per_step,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {'workload': cfg['name'], 'N': n, 'M': m, 'd': cfg['d'], 'parallelism': 'single GPU',
                   'factor': ('complete: zeros stored above the diagonal (LGP_BENCH_FULL_FACTOR=1)' if os.environ.get('LGP_BENCH_FULL_FACTOR') else
                              'lower triangle only, as GP.pred does (LGP_FACTOR_LOWER_ONLY: the prediction never reads above the diagonal; '
                              'LGP_BENCH_FULL_FACTOR=1 stores the zeros too, +0.46 ms at C3)')},
        'phases_ms': {'gram (cross blocks; Kxx is built inside the decomposition call)': float(phases[0]), 'gram + regularisation + decomposition': float(phases[1]), 'solves+likelihood': float(phases[2])},
        'cholesky_tflops': (n ** 3 / 3) / (phases[1] * 1e-3) / 1e12,
        'log_marginal_likelihood': wl.results.get('mll'),
        'roofline': {
            'kernel': 'gemm_kernel<double,NT> lower (syrk trailing update of the blocked Cholesky, v_mfma_f64_16x16x4_f64)',
            'bound': 'mfma',
            'achieved': achieved,
            'peak': FP64_MFMA_PEAK_TFLOPS,
            'unit': 'TFLOP/s',
            'frac': achieved / FP64_MFMA_PEAK_TFLOPS if achieved else None,
            'traffic': pmc_traffic(args.config)[0],
            'traffic_source': pmc_traffic(args.config)[1],       # (counters cannot be read from inside the process: a committed rocprofv3 --pmc pass of this command)
            'launches': int(launches.value),
            'avg_launch_ms': tot_ms.value / max(launches.value, 1),
            'avg_launch_gflop': tot_fl.value / max(launches.value, 1) / 1e9,
            'largest_launch_tflops': (max_fl.value / (max_ms.value * 1e-3) / 1e12) if max_ms.value > 0 else None,
            'coverage': ('the timed launches are the trailing updates that keep 128 x 128 tiles (the smaller ones go to the 64-tile instantiation, '
                         'which the other kernels of the factorisation share): %d per step, %.0f %% of the N^3/3 flops of the factorisation'
                         % (launches.value // max(args.steps, 1), 100. * tot_fl.value / max(args.steps, 1) / (n ** 3 / 3))),
            'whole_cholesky_frac_of_peak': (n ** 3 / 3) / (phases[1] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
        },
    }
    # the N x N Gram kernel alone against the HBM roofline (inside the step it is fused with the regularisation passes)
    try:
        scratch = lgp.DeviceArray.empty((n, n), ctx=ctx)
        e0, e1 = ctx.event(), ctx.event()
        wl._gram(wl.x, wl.x, scratch)
        gms = 1e9
        for _ in range(3):
            ctx.record(e0); wl._gram(wl.x, wl.x, scratch); ctx.record(e1)
            gms = min(gms, ctx.elapsed_ms(e0, e1))
        out['gram_build'] = {'kernel': 'gram_iso_kernel (N x N, stand-alone, best of 3; the same array on both sides: one triangle evaluated and mirrored '
                                       'for the cores heavier than ExpQuad)', 'bound': 'hbm', 'ms': gms, 'achieved': 8. * n * n / (gms * 1e-3) / 1e9,
                             'peak': 8000., 'write_peak': 6300., 'frac_of_write_peak': 8. * n * n / (gms * 1e-3) / 1e9 / 6300., 'unit': 'GB/s'}
        del scratch
    except Exception as exc:
        out['gram_build'] = {'error': repr(exc)}
    del wl
    if not args.no_sweep:
        out['vs_N'] = size_sweep(ctx, cfg)
        try:
            out['default_path'] = default_path(ctx, 'c2')
        except Exception as exc:
            out['default_path'] = {'error': repr(exc)}
        try:
            out['small_n_latency'] = sn = small_n_latency(ctx)
            # (also as rows of the size sweep: the reference's own benchmark, through the host classes with default options)
            for n in (10, 1000):
                out['vs_N'].append({'workload': f'docs/userguide/optim.rst:88-159 predfromdata ExpQuad N={n} M=100 raw=True via the host classes, default options, wall clock mean of 20 calls',
                                    'N': n, 'M': 100, 'ms_per_step': sn[f'N={n} default options ms'], 'ms_checks_off': sn[f'N={n} checkpos=False, checksym=False ms'],
                                    'min_ms_per_step': sn[f'N={n} default options min ms'], 'min_ms_checks_off': sn[f'N={n} checkpos=False, checksym=False min ms'],
                                    'reference_published_ms': {'eager': sn['reference_published_ms'][f'N={n} eager'], 'jit': sn['reference_published_ms'][f'N={n} jit']}})
        except Exception as exc:
            out['small_n_latency'] = {'error': repr(exc)}
    out['cpu_baseline'] = None if args.no_cpu_baseline else cpu_baseline(cfg)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    sys.exit(main() or 0)
