"""python tools/hunt_dist.py [reps] [n nb world slab] -- the multi-rank configuration that once returned a wrong posterior mean
(tests/test_gpu_dist.py::test_ranks_share_the_gpu[4100-128-4-...]) repeated INSIDE one set of rank processes, with the schedule
perturbed by LGP_DIST_JITTER (spin kernels of random length in front of randomly chosen enqueues, csrc/dist.hip): one
process start-up for `reps` factorisations instead of one per factorisation.  Every repetition every rank compares its block
columns (matrix rows and carried rows) with scipy's factor and says WHICH block columns / rows differ.

Environment: JITTER=<seed> (default 1; 0 = off), OWNER=order|stream (LGP_DIST_OWNER_COMM), SENDRECV=0|1 (scatter + all-gather),
TRACE=<path> (LGP_DIST_TRACE of the last repetition).  Parent mode (no RANK in the environment) starts the ranks.
"""
import os
import socket
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def parent():
    reps = sys.argv[1] if len(sys.argv) > 1 else '50'
    rest = sys.argv[2:6] if len(sys.argv) >= 6 else ['4100', '128', '4', '1024']
    world = int(rest[2])
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HUNT_PORT=str(port), LGP_DIST_SLAB=rest[3])
    jit = os.environ.get('JITTER', '1')
    if jit != '0':
        env['LGP_DIST_JITTER'] = jit
    if os.environ.get('OWNER'):
        env['LGP_DIST_OWNER_COMM'] = os.environ['OWNER']
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), reps] + rest, env=dict(env, RANK=str(r))) for r in range(world)]
    rc = [p.wait() for p in procs]
    print('exit codes', rc, flush=True)
    sys.exit(max(rc))


def rank_main():
    from scipy import linalg
    reps, n, nb, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rank, m = int(os.environ['RANK']), 37
    from lsqfitgp_amd._rendezvous import Rendezvous
    dist = Rendezvous(rank, world, port=int(os.environ['HUNT_PORT']))
    import lsqfitgp_amd as lgp
    from lsqfitgp_amd._dist import DistributedChol
    ctx = lgp.Context(0)
    lgp.Context._default = ctx
    ctx.comm_init_host(rank, world, dist.collective, sendrecv=os.environ.get('SENDRECV', '1') == '1')
    rng = np.random.default_rng(7)
    x = rng.uniform(0, n / 16, n)
    xs = rng.uniform(0, n / 16, m)
    y = np.sin(x / 4) + 0.1 * rng.standard_normal(n)
    K = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2) + 0.01 * np.eye(n)
    s = np.ones(n)
    eps = n * np.finfo(float).eps * np.abs(K).sum(axis=1).max()
    Lref = linalg.cholesky(K + eps * np.eye(n), lower=True)
    W = np.concatenate([np.exp(-0.5 * (x[:, None] - xs[None, :]) ** 2), y[:, None]], axis=1)
    Xref = linalg.solve_triangular(Lref, W, lower=True).T          # the carried rows: (L^-1 [Kxxs | y])^T
    mean_ref = Xref[:m] @ Xref[m]
    kern = lgp.ExpQuad()
    nbad = 0
    t0 = time.time()
    for rep in range(reps):
        if rep == reps - 1 and os.environ.get('TRACE'):
            os.environ['LGP_DIST_TRACE'] = os.environ['TRACE']
        d = DistributedChol(kern, x, noise=0.01, nb=nb, ctx=ctx, carry=(xs, y))
        mean, cov, mlnd = d.posterior(kern, x, xs, y)
        msgs = []
        em = np.linalg.norm(mean - mean_ref) / np.linalg.norm(mean_ref)
        if em > 1e-8:
            msgs.append(f'mean {em:.2e}')
        L = d.layout
        for b in L.owned():
            lb, w, c0 = L.local_index(b), L.width(b), L.col0(b)
            blk = d.Lloc.block(c0, n + m + 1, lb * nb, lb * nb + w).to_host()
            blk[:w] = np.tril(blk[:w])
            ref = np.concatenate([Lref[c0:, c0:c0 + w], Xref[:, c0:c0 + w]])
            err = np.abs(blk - ref).max(axis=1)
            if err.max() > 1e-9:
                rows = np.flatnonzero(err > 1e-9) + c0
                msgs.append(f'block column {b}: max error {err.max():.2e} in rows {rows.min()}..{rows.max()} ({len(rows)} rows; carried rows start at {n})')
        if msgs:
            nbad += 1
            print(f'rank {rank} rep {rep} WRONG: ' + '; '.join(msgs), flush=True)
        del d
        if rank == 0 and (rep + 1) % 10 == 0:
            print(f'rep {rep + 1} of {reps}: {time.time() - t0:.0f} s', flush=True)
    bad_any = dist.allreduce(float(nbad), 'sum')
    if rank == 0:
        print(f'HUNT n={n} nb={nb} world={world} reps={reps} jitter={os.environ.get("LGP_DIST_JITTER")} owner={os.environ.get("LGP_DIST_OWNER_COMM")} '
              f'sendrecv={os.environ.get("SENDRECV", "1")}: {int(bad_any)} wrong results (summed over ranks)', flush=True)
    dist.barrier()
    dist.close()
    sys.exit(1 if nbad else 0)


if __name__ == '__main__':
    if 'RANK' in os.environ:
        rank_main()
    else:
        parent()
