"""python tools/flaky_dist.py [reps] -- the 4-rank configuration of tests/test_gpu_dist.py::test_ranks_share_the_gpu run several times;
on a mismatch says WHAT differs (which block columns of the factor, the posterior, the density) instead of stopping at the first assert"""
import os, sys, socket, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
n, nb, world, m = 4100, 128, 4, 37
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(7)
x = rng.uniform(0, n / 16, n); xs = rng.uniform(0, n / 16, m)
y = np.sin(x / 4) + 0.1 * rng.standard_normal(n)
K = oracle.ExpQuad()(x[:, None], x[None, :]) + 0.01 * np.eye(n)
o = oracle.Chol(K)
Kxs = oracle.ExpQuad()(x[:, None], xs[None, :])
mean_o = o.pinv_bilinear(Kxs, y)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stress = []
if os.environ.get('STRESS'):        # CPU contention, as under pytest where the parent computes the oracle meanwhile
    stress = [subprocess.Popen([sys.executable, '-c', 'import numpy as np\na=np.random.rand(1500,1500)\nwhile True: a@a']) for _ in range(int(os.environ['STRESS']))]
for rep in range(reps):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    tmp = tempfile.mkdtemp()
    env = dict(os.environ, LGP_TEST_CARRY='1', LGP_TEST_AMP='1', LGP_TEST_SENDRECV=os.environ.get('SENDRECV', '1'), LGP_DIST_SLAB='1024')
    procs = [subprocess.Popen([sys.executable, os.path.join(root, 'tests', 'two_rank_worker.py'), str(r), str(world), str(port), str(n), str(nb), str(m), tmp], cwd=root, env=env)
             for r in range(world)]
    rc = [p.wait(timeout=600) for p in procs]
    bad = []
    for r in range(world):
        d = np.load(os.path.join(tmp, f'rank{r}.npz'))
        em = np.linalg.norm(d['mean'] - mean_o) / np.linalg.norm(mean_o)
        for b in d['owned']:
            blk = d[f'b{b}']; c0 = int(b) * nb
            ref = o._L[c0:, c0:c0 + blk.shape[1]]
            err = np.abs(blk - ref).max(axis=1)
            if err.max() > 1e-9:
                rows = np.flatnonzero(err > 1e-9) + c0
                bad.append((int(b), float(err.max()), int(rows.min()), int(rows.max()), len(rows)))
    print(f'rep {rep}: rc {rc} mean rel err {em:.2e} bad block columns (b, max err, first row, last row, rows): {bad[:12]}', flush=True)

for p in stress:
    p.kill()
