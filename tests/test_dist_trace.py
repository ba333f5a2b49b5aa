"""
The schedule of the sharded factorisation as the C++ REALLY enqueues it (``lgp_chol_dist_trace``: csrc/dist.hip run without a
device, every enqueue written down instead of performed), checked on CPU:

* statically -- every pair of conflicting accesses of a rank ordered by stream order + event edges (`Schedule.hazards`),
* dynamically -- the records executed over numpy by all ranks in random orders that respect what HIP and the communicator
  order: no deadlock, no operand read before it is there, the gathered factor equal to LAPACK's,
* against the readable restatement ``lsqfitgp_amd._dist.factor_tasks``: same operations on the same streams in the same order
  with the same event edges.

World sizes 1-8, both ways of issuing the owner's broadcast, the parametrisations of the on-device multi-rank tests
(tests/test_gpu_dist.py), and C3 / C4 at full size (static check only).  No reference counterpart.
"""

import numpy as np
import pytest
from scipy import linalg

from lsqfitgp_amd._dist import Layout, dist_plan, factor_tasks

from . import dist_trace as dt

# (n, nb, world, nx, slab) of tests/test_gpu_dist.py::test_ranks_share_the_gpu (nx = 128 where it carries right-hand sides)
ON_DEVICE = [(1500, 256, 2, 0, 512), (2600, 1024, 2, 128, None), (2900, 256, 3, 128, 768), (1700, 256, 2, 128, 256), (4100, 128, 4, 128, 1024),
             (2900, 256, 3, 0, 512), (3300, 256, 5, 128, 512)]
SMALL = [(37, 8, 1, 0, 0), (40, 8, 1, 3, 16), (45, 8, 2, 0, 16), (64, 8, 2, 5, 8), (50, 7, 3, 0, 14), (61, 8, 3, 4, 24), (96, 8, 4, 0, 16), (70, 6, 5, 2, 12),
         (128, 8, 8, 0, 32), (150, 8, 8, 6, 16), (16, 16, 2, 0, 16), (130, 8, 8, 3, 0), (700, 128, 2, 0, None), (1100, 128, 4, 128, 256), (900, 256, 3, 128, 512),
         (640, 128, 1, 128, None), (1300, 256, 5, 128, 256)]


def _matrix(n, nx, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, n / 8, n)
    K = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2) + 0.05 * np.eye(n)
    if nx:
        B = np.zeros((nx, n))
        B[:min(nx, 5)] = rng.standard_normal((min(nx, 5), n))
        K = np.concatenate([K, B])
    return K


def _schedules(n, nb, world, nx, slab, flags):
    return [dt.Schedule(dt.dry_trace(n, nb, nx, r, world, slab, flags)) for r in range(world)]


@pytest.mark.parametrize('n,nb,world,nx,slab', ON_DEVICE + SMALL)
@pytest.mark.parametrize('flags', [0, 1, 3])
def test_no_unordered_conflict(n, nb, world, nx, slab, flags):
    """ flags: 0 the owner broadcasts from the stream that packed, 1 from the communication stream behind an event, 3 = 1 + the
    squares go to the chain kernel where the library would use it """
    if flags == 3 and nb < 256:
        pytest.skip('the chain kernel takes squares of at least 256 columns')
    for s in _schedules(n, nb, world, nx, slab, flags):
        assert (s.n, s.nb, s.nx, s.rank < world, s.world) == (n, nb, nx, True, world)
        assert s.owner_via_stream == (flags & 1)
        hz = s.hazards()
        assert not hz, f'rank {s.rank}: ' + '; '.join(f'{k} on {rect}: {a} -> {b}' for a, b, rect, k in hz[:5])
        # a wait whose event is not recorded earlier in the call is a no-op: only the slot-free event of the squares' stream, on
        # ranks that never read that slot there, may be one
        for w in s.unrecorded_waits():
            assert w.ctx[0] in ('comm', 'lane') and w.ctx[2] >= 1, w


@pytest.mark.parametrize('n,nb,world', [(32768, 1024, 1), (65536, 1024, 2), (65536, 1024, 4), (65536, 1024, 8), (32768, 1024, 8)])
def test_no_unordered_conflict_at_full_size(n, nb, world):
    """ C3 / C4 with the library's defaults for that many ranks (chain kernel assumed usable) """
    for rank in sorted({0, world - 1}):
        s = dt.Schedule(dt.dry_trace(n, nb, 128, rank, world, None, 3 if world > 1 else 2))
        assert not s.hazards()
        assert s.slab == (0 if world == 1 else 8192)
    assert dt.Schedule(dt.dry_trace(n, nb, 128, 0, world, None, -1)).owner_via_stream == (world > 1)


def test_the_checker_sees_a_missing_edge():
    """ the static check is not vacuous: without the event edges of one stream it reports conflicts """
    rec = dt.dry_trace(1100, 128, 128, 1, 4, 256, 1)
    s = dt.Schedule(rec)
    assert not s.hazards()
    cut = rec.copy()
    waits_sc = np.flatnonzero((cut[:, 0] == dt.WAIT) & (cut[:, 1] == 0))
    cut = np.delete(cut, waits_sc[3:], axis=0)                  # the compute stream stops waiting for arrivals
    hz = dt.Schedule(cut).hazards(limit=1000)
    assert hz and all(b.stream == 0 or a.stream == 0 for a, b, _, _ in hz)
    cut = np.delete(rec, np.flatnonzero((rec[:, 0] == dt.WAIT) & (rec[:, 1] == 2)), axis=0)     # the rows below the squares stop waiting
    assert dt.Schedule(cut).hazards()


@pytest.mark.parametrize('n,nb,world,nx,slab', SMALL + [(1500, 256, 2, 0, 512), (1700, 256, 2, 128, 256)])
def test_trace_executed_in_random_order(n, nb, world, nx, slab):
    K = _matrix(n, nx)
    ref = linalg.cholesky(K[:n], lower=True)
    for flags in (0, 1) + ((3,) if nb >= 256 else ()):
        scheds = _schedules(n, nb, world, nx, slab, flags)
        for seed in range(3 if n < 300 else 1):
            full = dt.gather(dt.simulate(scheds, K, seed + 10 * flags))
            np.testing.assert_allclose(full[:n], ref, rtol=1e-11, atol=1e-13)
            if nx:
                np.testing.assert_allclose(full[n:], linalg.solve_triangular(ref, K[n:].T, lower=True).T, rtol=1e-9, atol=1e-11)


# ---- the restatement against the records ----------------------------------------------------------------------------------

def _composites(s):
    """ the trace as the restatement's vocabulary: list of [stream, kind, key, waits] with key = (panel, first row) and waits = the
    composites whose completion its event waits refer to """
    groups, cur = [], None
    for nd in s.nodes:
        if nd.ctx is None:
            continue
        if cur is None or cur[0] is not nd.ctx:
            cur = (nd.ctx, [])
            groups.append(cur)
        cur[1].append(nd)
    comps, of_node = [], {}

    def add(stream, kind, key, nodes):
        comps.append([dt.STREAMS[stream], kind, key, nodes])
        for nd in nodes:
            of_node[nd.index] = len(comps) - 1

    for ctx, nodes in groups:
        what, b, i, r0, r1 = ctx
        st = nodes[0].stream
        if what == 'u1':
            add(st, 'U', (b, r0), nodes)
        elif what == 'lane' and i == 0:
            add(st, 'SQ', (b, r0), nodes)
        elif what == 'lane':
            k = next(j for j, nd in enumerate(nodes) if nd.kind == dt.RECORD) + 1           # ... up to "these rows are solved"
            add(st, 'SOLVE', (b, r0), nodes[:k])
            if any(nd.kind == dt.COPY2D for nd in nodes[k:]):
                add(st, 'PACK', (b, r0), nodes[k:])
        elif what == 'comm':
            add(st, 'BCAST', (b, r0), nodes)
        elif what == 'bulk':
            head = [nodes[0]]                                                               # the wait for the piece itself: both parts
            rest = nodes[1:]
            u2 = [j for j, nd in enumerate(rest) if nd.kind == dt.GEMM and not nd.args[12]]
            has_u2_event = s.rank == (b + 1) % s.world and b + 1 < -(-s.n // s.nb)
            k = 0
            if has_u2_event:
                k = next(j for j, nd in enumerate(rest) if nd.kind == dt.RECORD) + 1
                if u2:
                    row = s._rc(0, rest[u2[0]].args[10])[0]
                    add(st, 'U', (b + 1, row), head + rest[:k])
                else:
                    add(st, 'NOP', (b, r0), head + rest[:k])
            add(st, 'BULK', (b, r0), head + rest[k:])
    last_rec = {}
    waits = [set() for _ in comps]
    for nd in s.nodes:
        if nd.kind == dt.RECORD:
            last_rec[nd.event] = nd.index
        elif nd.kind == dt.WAIT and nd.event in last_rec and nd.index in of_node:
            src = of_node.get(last_rec[nd.event])
            if src is not None:
                # a group's shared leading wait belongs to each composite made from the group
                for c in {c for c in range(len(comps)) if nd in comps[c][3]}:
                    if src != c:
                        waits[c].add(src)
    return [(st, kind, key, frozenset((comps[w][0], comps[w][1], comps[w][2]) for w in ws)) for (st, kind, key, _), ws in zip(comps, waits)]


def _restated(layout, nx, slab, owner_comm):
    plan = dist_plan(layout.n, layout.nb, nx, layout.nranks, slab)
    tasks, _ = factor_tasks(layout, plan, nx, owner_comm)
    out = []
    for t in tasks:
        b, i, r0, r1 = t.piece
        ws = frozenset((tasks[w].stream, tasks[w].kind, (tasks[w].piece[0], tasks[w].piece[2])) for w in t.waits)
        out.append((t.stream, t.kind, (b, r0), ws))
    return out


@pytest.mark.parametrize('n,nb,world,nx,slab', SMALL + ON_DEVICE)
def test_restatement_is_what_the_library_enqueues(n, nb, world, nx, slab):
    for flags, owner_comm in ((0, 'order'), (1, 'stream')):
        for rank in range(world):
            s = dt.Schedule(dt.dry_trace(n, nb, nx, rank, world, slab, flags))
            if any(nd.kind == dt.TAIL for nd in s.nodes):
                continue           # (one rank, large: the end goes to the single-GPU chain driver, which the restatement does not describe)
            got = _composites(s)
            want = _restated(Layout(n, nb, rank, world), nx, slab, owner_comm)
            if world == 1:         # (without peers nothing is packed or sent)
                want = [t for t in want if t[1] not in ('PACK', 'BCAST')]
                want = [(st, k, key, frozenset(w for w in ws if w[1] not in ('PACK', 'BCAST'))) for st, k, key, ws in want]
            assert [t[:3] for t in got] == [t[:3] for t in want], f'rank {rank}: different operations'
            for g, w in zip(got, want):
                assert g[3] == w[3], f'rank {rank}: {g[:3]} waits for {sorted(g[3])}, restated as {sorted(w[3])}'
