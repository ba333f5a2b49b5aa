"""
Loader of the reference-held golden vectors under tests/golden/reference_special/ (mpmath values the reference's own tests
compare its special functions with) and the inputs they belong to, regenerated as the reference's tests generate them
(``tests/conftest.py:37-42``, ``tests/test_special.py:146-154,190-226,415-424``).
"""
import gzip
import json
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_special')


def _hook(obj):
    c = obj.get('__class__')
    if c == 'array':
        return np.array(*obj.get('args', ()), **obj.get('kw', {}))
    if c == 'complex':
        return complex(*obj.get('args', ()))
    return obj


def load(name):
    with gzip.open(os.path.join(HERE, name + '.json.gz'), 'rt') as f:
        return json.load(f, object_hook=_hook)


def reference_rng(nodeid):
    """ the reference's per-test generator (``tests/conftest.py:37-42``) """
    return np.random.default_rng(np.array([nodeid], np.bytes_).view(np.uint8))


def expn_case():
    """ n (10,), x (10,), E_n(-ix) (10, 10) complex (rows: n) """
    rng = reference_rng('tests/test_special.py::test_expn')
    x = rng.uniform(0, 30, 10)
    return np.arange(2, 12), x, load('test_expn')['sol']


PZ_S = {'near1': np.array([1 + 1e-15]), 'half': 0.5 + np.arange(1, 15), 'even': np.arange(2, 15, 2.), 'odd': np.arange(3, 15, 2.)}
PZ_D = {'at': 0., 'close': 1e-4, 'veryclose': 1e-13}
PZ_SGN = {'pos': 1, 'neg': -1}


def periodic_zeta_cases():
    """ (id, x (52,), s (ns,), Li_s(e^{2 pi i x}) (ns, 52) complex) for every stored parametrisation """
    x = np.linspace(-1, 2, 52)
    out = []
    for sid, s0 in PZ_S.items():
        for did, d in PZ_D.items():
            for gid, sgn in PZ_SGN.items():
                name = f'test_periodic_zeta[{sid}-{did}-{gid}-real]'
                if not os.path.exists(os.path.join(HERE, name + '.json.gz')):
                    continue                       # the parametrisations the reference skips (d = 0 with sgn < 0, s <= 1)
                s = s0 + sgn * d
                out.append((f'{sid}-{did}-{gid}', x, s, load(name)['z1']))
    return out


def hurwitz_case():
    """ s (100,), a (100,), zeta(s, a) (100, 100) """
    return np.linspace(-10, 0, 100), np.linspace(0, 1, 100), load('test_hurwitz_zeta')['z1']
