#!/usr/bin/env python3
"""
bench.py -- the hot path of BASELINE.json on MI355X: dense GP fit+predict in fp64
(kernel Gram build -> scaling + Gershgorin jitter -> blocked Cholesky -> triangular solves ->
posterior mean / covariance + log marginal likelihood), through the C-ABI of
liblsqfitgp_hip.so with every buffer resident in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5] [--no-cpu-baseline]

One "step" = one full pass of the path over one batch of synthetic input:
    c2  ExpQuad 1-D, N = 8192,  M = 100, y = sin(x/4) + noise, givencov 0.01 I
    c3  Matern-5/2, 3-D covariates, N = 32768, M = 100, noise 0.01 I      (default at --gpus 1)
    c4  ExpQuad 1-D, N = 65536, M = 100, noise 0.01 I
    c5  BART p = 10 (fp32 Gram), N = 4096, M = 100, through the GP / BART / Chol host classes
With --gpus N > 1 (launched by torch.distributed.run, one process per GPU) every rank runs an independent
replica of the workload on its own GPU ("replicas only" until the block-column distributed factorisation is
the default) -- no data-path collective; torch.distributed (gloo) is used only for the barrier
and the max-over-ranks of the timing.

Prints ONE JSON line (rank 0): whole-job throughput in TFLOP/s over the algorithmic flop count
N^3/3 + N^2 (M+1) + N M^2, the roofline of the dominant kernel (syrk trailing update of the Cholesky,
timed live with HIP events on the stream it runs on) and a CPU baseline (the numpy/scipy oracle on a
bounded sample of the same workload, on this box's host cores).
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
        self.K = DA.empty((n, n), ctx=ctx)           # Kxx + ycov
        self.fused = not os.environ.get('LGP_BENCH_SEPARATE_SOLVE')
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
        if self.fused:
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


def cpu_baseline(cfg, budget_s=25.0):
    """ the oracle (numpy/scipy restatement of the reference's path) on a bounded sample: the same workload at
    a reduced N chosen so that the CPU work stays within ~10-30 s """
    import oracle
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    m = cfg['m']
    n = min(cfg['n'], 8192)
    x, xs, y = synth(cfg, n)

    def run(n_):
        xx, xxs, yy = x[:, :n_], xs, y[:n_]
        if cfg['kernel'] == 'expquad':
            k = oracle.ExpQuad()
            a, b = xx[0], xxs[0]
        else:
            k = oracle.Maternp(p=2, scale=1.)
            dt = [('f0', float), ('f1', float), ('f2', float)]
            a = np.empty(n_, dt); b = np.empty(m, dt)
            for i, f in enumerate(('f0', 'f1', 'f2')):
                a[f] = xx[i]; b[f] = xxs[i]
        t0 = time.perf_counter()
        gp = oracle.GP(k, checkpos=False, checksym=False).addx(a, 'data').addx(b, 'test')
        gc = {('data', 'data'): 0.01 * np.eye(n_)}
        mean, cov = gp.predfromdata({'data': yy}, 'test', gc, raw=True)
        mll = gp.marginal_likelihood({'data': yy}, gc)
        return time.perf_counter() - t0

    t = run(n)
    while t > budget_s and n > 1024:
        n //= 2
        t = run(n)
    tf = flops_fit_predict(n, m) / t / 1e12
    return dict(value=tf, unit='TFLOP/s', cores=int(threads), kind='port',
                sample=f'{cfg["kernel"]} d={cfg["d"]} N={n} M={m} fit+predict with the numpy/scipy oracle (LAPACK potrf/trsm), {t:.2f} s wall, one run',
                seconds=t, host_cpus=os.cpu_count())


def size_sweep(ctx, main_cfg):
    """ "vs N" part of BASELINE.json's metric: the same step at smaller sizes and on configuration C2, 1 warmup + 3 timed
    steps each (a fraction of a second in all); not part of `value` """
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
        reps = 3 if c['n'] <= 32768 else 2
        evs = [[ctx.event() for _ in range(4)] for _ in range(reps)]
        t0 = time.perf_counter()
        for k in range(reps):
            wl.step(evs[k])
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        ph = np.array([[ctx.elapsed_ms(e[i], e[i + 1]) for i in range(3)] for e in evs]).mean(axis=0)
        out.append({'workload': c['name'], 'N': c['n'], 'ms_per_step': dt * 1e3, 'value': flops_fit_predict(c['n'], c['m']) / dt / 1e12,
                    'cholesky_ms': float(ph[1]), 'cholesky_tflops': (c['n'] ** 3 / 3) / (ph[1] * 1e-3) / 1e12})
        del wl
    return out


def main_c5(args, ctx):
    """ configuration C5: the GP that bayestree.bart builds (``bayestree/_bart.py:187-205``) at fixed hyperparameters on
    synthetic Abalone-like covariates (p = 10, N = 4096 train, M = 100 test), driven through the host classes
    (``GP`` / ``BART`` / ``Chol``) exactly as a user of the reference would: fp32 BART Gram, fp64 assembly and solve.
    One step = new GP -> predfromdata(testmean) + marginal_likelihood.  Roofline object: the BART Gram kernel (HBM). """
    import lsqfitgp_amd as lgp
    rng = np.random.default_rng(5)
    n, nt, p = 4096, 100, 10
    f = np.float32
    X = np.concatenate([rng.integers(0, 2, (n + nt, 3)), rng.integers(0, 200, (n + nt, 7)) / 200.], axis=1).astype(f)
    yv = (10 * X[:n, 4] + 3 * X[:n, 0] + rng.standard_normal(n)).astype(f)
    k_sigma_mu = f((yv.max() - yv.min()) / 2)
    lam2 = f((k_sigma_mu / f(2)) ** 2)
    sigma2 = f(np.mean((yv - yv.mean()) ** 2))
    splits = lgp.BART.splits_from_coord(X[:n])
    ix = lgp.BART.indices_from_coord(X, splits).astype(np.int32)
    pts = np.array([tuple(r) for r in ix], dtype=[(f'f{i}', np.int32) for i in range(p)])
    kern = lgp.BART(splits=(splits[0], None), indices=True, alpha=f(0.95), beta=f(2), gamma=f(1), maxd=10, reset=[2, 4, 6, 8],
                    weights=np.ones(p, f)) * lam2
    noise = np.diag(np.full(n, sigma2)).astype(f)
    given = {'train': yv - f(yv.mean())}

    def step():
        gp = (lgp.GP(kern, checkpos=False, checksym=False, solver='chol')
              .addx(pts[:n], 'trainmean').addx(pts[n:], 'testmean')
              .addcov(noise, 'trainnoise').addcov(np.asarray(k_sigma_mu ** 2, f), 'mean')
              .addtransf({'trainmean': 1, 'trainnoise': 1, 'mean': 1}, 'train'))
        mean, cov = gp.predfromdata(given, 'testmean', raw=True)
        return mean, cov, gp.marginal_likelihood(given, epsrel=0)

    for _ in range(args.warmup):
        step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    ctx.sync()
    dt = (time.perf_counter() - t0) / args.steps
    # the BART Gram kernel alone, N x N fp32, HIP events on the context's stream
    e0, e1 = ctx.event(), ctx.event()
    kern.gram(pts[:n], pts[:n], ctx=ctx)
    best = 1e9
    for _ in range(5):
        ctx.record(e0)
        kern.gram(pts[:n], pts[:n], ctx=ctx)
        ctx.record(e1)
        best = min(best, ctx.elapsed_ms(e0, e1))
    gbps = 4. * n * n / (best * 1e-3) / 1e9
    print(json.dumps({
        'metric': 'GP fit+predict throughput (Gram build + Cholesky + solves; flops N^3/3 + N^2(M+1) + N M^2)',
        'value': flops_fit_predict(n, nt) / dt / 1e12, 'unit': 'TFLOP/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32 Gram, f64 assembly and solve (as the reference: zero cross blocks are float64)', 'data': 'synthetic',
        'config': {'workload': 'C5 BART kernel p=10 N=4096 M=100 via GP/BART/Chol host classes (bayestree.bart GP at fixed hyperparameters)',
                   'N': n, 'M': nt, 'p': p, 'parallelism': 'single GPU'},
        'log_marginal_likelihood': float(res[2]),
        'roofline': {'kernel': 'gram_bart_kernel<float> (N x N, includes the host-side per-call setup of kernel.gram)', 'bound': 'hbm',
                     'achieved': gbps, 'peak': 8000., 'unit': 'GB/s', 'frac': gbps / 8000., 'traffic': None, 'avg_launch_ms': best},
        'cpu_baseline': None}), flush=True)


def pmc_traffic(config):
    """ HBM bytes per trailing-update launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE
    x2 per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE, collected in their own runs of this same
    command); counters cannot be read live from inside the process, so None when no committed pass matches """
    path = os.path.join(ROOT, 'profiles', f'r01_{config}_pmc_traffic.json')
    try:
        with open(path) as f:
            d = json.load(f)
        return d['hbm_read_bytes_per_launch_corrected'] + d['hbm_write_bytes_per_launch']
    except Exception:
        return None


def main_dist(args, cfg, ctx, dist, rank, world):
    """ opt-in: Gram build + regularisation + Cholesky of ONE N x N problem sharded by block columns over the ranks
    (lsqfitgp_amd/_dist.py); strong scaling, value = N^3/3 flops / time """
    import torch
    import lsqfitgp_amd as lgp
    from lsqfitgp_amd._dist import DistributedChol
    if os.environ.get('LGP_DIST_TRANSPORT') == 'host':
        # panels through host memory and gloo (lgp_comm_init_host): lets several ranks share one GPU, e.g. to rehearse the
        # multi-rank path on a single-GPU box; the default is RCCL over xGMI
        def collective(op, arr, root):
            t = torch.from_numpy(arr)
            if op == 0:
                dist.broadcast(t, src=root)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 1 else dist.ReduceOp.MAX)
        ctx.comm_init_host(rank, world, collective)
    else:
        uid = [lgp.Context.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(rank, world, uid[0])
    lgp.Context._default = ctx
    x, xs, y = synth(cfg)
    kern = lgp.ExpQuad() if cfg['kernel'] == 'expquad' else lgp.Maternp(p=2, scale=1.)
    if cfg['d'] == 1:
        pts = x[0]
    else:
        pts = np.empty(cfg['n'], [(f'f{i}', float) for i in range(cfg['d'])])
        for i in range(cfg['d']):
            pts[f'f{i}'] = x[i]

    if cfg['d'] == 1:
        tpts = xs[0]
    else:
        tpts = np.empty(cfg['m'], pts.dtype)
        for i in range(cfg['d']):
            tpts[f'f{i}'] = xs[i]

    def step():
        d = DistributedChol(kern, pts, noise=0.01, nb=1024, ctx=ctx, carry=(tpts, y))   # Gram build + jitter + factorisation, solve carried
        mean, cov, mlnd = d.posterior(kern, pts, tpts, y)                     # fan-in solve, mean, covariance, likelihood
        return -mlnd

    for _ in range(args.warmup):
        step()
    ctx.sync(); dist.barrier(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ld = step()
    ctx.sync(); dist.barrier(); ctx.sync()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0])
    n = cfg['n']
    if rank == 0:
        print(json.dumps({
            'metric': 'GP fit+predict fp64 throughput, one problem sharded over the GPUs (flops N^3/3 + N^2(M+1) + N M^2)',
            'value': flops_fit_predict(n, cfg['m']) * args.steps / elapsed / 1e12, 'unit': 'TFLOP/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': cfg['name'], 'N': n, 'M': cfg['m'],
                       'parallelism': f'1-D block-column-cyclic over {world} GPUs, nb=1024, ' + ('host-staged gloo' if os.environ.get('LGP_DIST_TRANSPORT') == 'host' else 'RCCL') + ' panel broadcast with look-ahead'},
            'log_marginal_likelihood': ld, 'roofline': None, 'cpu_baseline': None}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', choices=sorted(CONFIGS) + ['c5'], default='c3')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-sweep', action='store_true', help='skip the smaller-N / C2 runs reported under "vs_N"')
    ap.add_argument('--dist', action='store_true', default=os.environ.get('LGP_BENCH_DIST') == '1',
                    help='N > 1: one block-column-cyclic factorisation sharded over the ranks (RCCL panel broadcasts) '
                         'instead of independent replicas')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    if world > 1:
        import torch                       # plumbing only: rendezvous, barrier, max-over-ranks
        import torch.distributed as dist
        dist.init_process_group('gloo', rank=rank, world_size=world)
        os.environ.setdefault('LGP_DEVICE', str(local_rank))

    import lsqfitgp_amd as lgp              # fails loudly without the HIP library / a GPU
    ctx = lgp.Context(int(os.environ.get('LGP_DEVICE', local_rank)))
    if args.config == 'c5':
        lgp.Context._default = ctx
        return main_c5(args, ctx)
    cfg = CONFIGS[args.config]
    if args.dist:
        if dist is None:                      # single process: still goes through the sharded code path (1 rank)
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29511')
            dist.init_process_group('gloo', rank=0, world_size=1)
        return main_dist(args, cfg, ctx, dist, rank, world)
    wl = Workload(cfg, ctx)
    lib, _lib = wl.lib, wl._lib

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
        ctx.sync()

    for _ in range(args.warmup):
        wl.step()
    barrier()

    # ---- timed region: exactly K steps
    evs = [[ctx.event() for _ in range(4)] for _ in range(args.steps)]
    _lib.check(lib.lgp_profile_begin(ctx.handle))
    t0 = time.perf_counter()
    for k in range(args.steps):
        wl.step(evs[k])
    barrier()
    t1 = time.perf_counter()
    launches = ctypes.c_size_t(); tot_ms = ctypes.c_double(); tot_fl = ctypes.c_double(); max_fl = ctypes.c_double(); max_ms = ctypes.c_double()
    _lib.check(lib.lgp_profile_end(ctx.handle, ctypes.byref(launches), ctypes.byref(tot_ms), ctypes.byref(tot_fl), ctypes.byref(max_fl), ctypes.byref(max_ms)))
    elapsed = t1 - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    n, m = cfg['n'], cfg['m']
    ms_per_step = elapsed / args.steps * 1e3
    total_flops = flops_fit_predict(n, m) * args.steps * world
    phases = np.array([[ctx.elapsed_ms(e[i], e[i + 1]) for i in range(3)] for e in evs]).mean(axis=0)

    if rank == 0:
        achieved = tot_fl.value / (tot_ms.value * 1e-3) / 1e12 if tot_ms.value > 0 else None
        out = {
            'metric': 'GP fit+predict fp64 throughput (Gram build + Cholesky + solves; flops N^3/3 + N^2(M+1) + N M^2)',
            'value': total_flops / elapsed / 1e12,
            'unit': 'TFLOP/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': cfg['name'], 'N': n, 'M': m, 'd': cfg['d'],
                       'parallelism': 'single GPU' if world == 1 else f'{world} independent replicas, one per GPU (no collective)'},
            'phases_ms': {'gram': float(phases[0]), 'decomposition': float(phases[1]), 'solves+likelihood': float(phases[2])},
            'cholesky_tflops': (n ** 3 / 3) / (phases[1] * 1e-3) / 1e12,
            # Gram build against the HBM roofline: bytes written for Kxx, Kxxs, Kxsxs / time of the phase
            'gram_build': {'bound': 'hbm', 'achieved': 8. * (n * n + n * m + m * m) / (phases[0] * 1e-3) / 1e9, 'peak': 8000., 'unit': 'GB/s'},
            'log_marginal_likelihood': wl.results.get('mll'),
            'roofline': {
                'kernel': 'gemm_kernel<double,NT> lower (syrk trailing update of the blocked Cholesky, v_mfma_f64_16x16x4_f64)',
                'bound': 'mfma',
                'achieved': achieved,
                'peak': FP64_MFMA_PEAK_TFLOPS,
                'unit': 'TFLOP/s',
                'frac': achieved / FP64_MFMA_PEAK_TFLOPS if achieved else None,
                'traffic': pmc_traffic(args.config),
                'launches': int(launches.value),
                'avg_launch_ms': tot_ms.value / max(launches.value, 1),
                'avg_launch_gflop': tot_fl.value / max(launches.value, 1) / 1e9,
                'largest_launch_tflops': (max_fl.value / (max_ms.value * 1e-3) / 1e12) if max_ms.value > 0 else None,
            },
        }
        if world == 1 and not args.no_sweep:
            out['vs_N'] = size_sweep(ctx, cfg)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
