"""
TEST INFRASTRUCTURE (part of the CPU oracle).  Derivatives of the kernels with an elementary core by SYMBOLIC differentiation
(sympy) evaluated in extended precision (mpmath): the checker's independent route to what the reference gets from jax.grad of
the core when points are added with ``deriv=`` (``src/lsqfitgp/_Kernel/_ops.py:115-211``).  The device evaluates the same
derivatives in truncated Taylor arithmetic (``lsqfitgp_amd/csrc/gram_jet.hip``); nothing is shared between the two.

A core announces its expression through ``core.symbolic = (kind, builder)``:
    'delta'  builder(d): a function of the signed difference of one-dimensional points;
    'abs'    builder(a): a function of a = |difference| >= 0, continued as an even function: sign(0) = +1;
    'circ'   builder(t): a function of the distance t of |difference| to the nearest integer;
    'xy'     builder(x, y, upper): of the two points themselves; upper: the branch x <= y;
    'iso'    builder(r2): of the squared distance over the fields;
    'dots'   builder(xy, xx, yy): of the three dot products x.y, x.x, y.y over the fields;
an optional third entry is the radius of a compact support (the expression is then the one inside it), a fourth a dictionary
``dps`` / ``offset``: working digits and the offset from which the limit at coincident points is taken (default 120, 1e-10).
"""

import numpy


def symbolic_diff(kernel, da, db, Kernel):
    """ the kernel derived da = {field: order} times in its first argument and db in its second (one field per side) """
    import mpmath
    import sympy as sp
    kind, build = kernel._core.symbolic[:2]
    support = kernel._core.symbolic[2] if len(kernel._core.symbolic) > 2 else None       # compact support: zero for arguments >= support
    opts = kernel._core.symbolic[3] if len(kernel._core.symbolic) > 3 else {}            # working precision / offset at coincidences
    scale = float(kernel._scale if kernel._scale is not None else 1.)
    factor = 1.
    for op, c in kernel._outops:
        if op == 'mul':
            factor = factor * c
    order = sum(da.values()) + sum(db.values())
    factor = factor / scale ** order
    fx, fy = list(da.items()), list(db.items())
    assert len(fx) <= 1 and len(fy) <= 1, 'one field per side'
    (xf, a), (yf, b) = (fx[0] if fx else (None, 0)), (fy[0] if fy else (None, 0))

    def mp_eval(expr, syms):
        f = sp.lambdify(syms, expr, modules='mpmath')
        return lambda *v: f(*[t if isinstance(t, mpmath.mpf) else mpmath.mpf(float(t)) for t in v])

    def column(x):
        x = numpy.asarray(x)
        if x.dtype.names is not None:
            assert len(x.dtype.names) == 1
            x = x[x.dtype.names[0]]
        return numpy.asarray(x, float)

    if kind in ('delta', 'abs', 'circ'):
        z = sp.Symbol('z', positive=True) if kind != 'delta' else sp.Symbol('z', real=True)
        q = a + b
        fn = mp_eval(sp.diff(build(z), z, q) if q else build(z), [z])
        small = opts.get('small')                              # (threshold, builder): the branch the reference's core takes below it
        fn_small = mp_eval(sp.diff(small[1](z), z, q) if q else small[1](z), [z]) if small else None

        def core(x, y):
            d = column(x) - column(y)
            out = numpy.empty(d.shape)
            with mpmath.workdps(70):
                for idx in numpy.ndindex(d.shape):
                    v = d[idx]
                    sgn = 1. if v >= 0 else -1.
                    if kind == 'delta':
                        val = fn(v if v != 0 else 1e-12)           # (removable singularities, sin(pi d) / (pi d): the limit from a small offset)
                    else:
                        arg = abs(v)
                        if kind == 'circ':
                            fr = arg - numpy.floor(arg)
                            sgn = sgn * (1. if fr <= 0.5 else -1.)
                            arg = min(fr, 1 - fr)
                        if small is not None and arg < small[0]:
                            val = fn_small(arg) * sgn ** q
                        else:
                            val = 0. if (support is not None and arg >= support) else fn(arg if arg > 0 else 1e-12) * sgn ** q
                    out[idx] = float(mpmath.re(val)) * (-1.) ** b      # d/dy of a function of x - y (real expressions written with complex parts: Color)
            return out
        return Kernel(core, scale=kernel._scale, loc=kernel._loc, outops=(('mul', factor),))

    if kind == 'xy':
        X, Y = sp.symbols('X Y', real=True)
        exprs = {br: mp_eval(sp.diff(build(X, Y, br), *([X] * a + [Y] * b)) if a + b else build(X, Y, br), [X, Y]) for br in (True, False)}

        def core(x, y):
            x, y = numpy.broadcast_arrays(column(x), column(y))
            out = numpy.empty(x.shape)
            with mpmath.workdps(40):
                for idx in numpy.ndindex(x.shape):
                    out[idx] = float(exprs[bool(x[idx] <= y[idx])](x[idx], y[idx]))
            return out
        return Kernel(core, scale=kernel._scale, loc=kernel._loc, outops=(('mul', factor),))

    if kind == 'dots':
        # U: the x side's differentiated coordinate (field f), Z: the y side's (field g); every other coordinate enters as a number.
        # f = g (or one side only): x.y = U Z + Rxy;  f != g: x.y = U Vf + Xg Z + Rxy with Vf = y_f, Xg = x_g;  x.x = U^2 + Rxx, y.y = Z^2 + Ryy
        U, Z, Vf, Xg, Rxy, Rxx, Ryy = sp.symbols('U Z Vf Xg Rxy Rxx Ryy', real=True)
        split = bool(a and b and xf != yf)
        expr = build((U * Vf + Xg * Z + Rxy) if split else (U * Z + Rxy), U ** 2 + Rxx, Z ** 2 + Ryy)
        fn = mp_eval(sp.diff(expr, *([U] * a + [Z] * b)) if a + b else expr, [U, Z, Vf, Xg, Rxy, Rxx, Ryy])

        def core(x, y):
            x = numpy.asarray(x); y = numpy.asarray(y)
            names = x.dtype.names
            if names is None:
                cols = {None: (numpy.asarray(x, float), numpy.asarray(y, float))}
            else:
                cols = {nm: (numpy.asarray(x[nm], float), numpy.asarray(y[nm], float)) for nm in names}
            fa = xf if a else (yf if b else next(iter(cols)))          # the field U stands for
            fb = yf if split else fa                                     # ... and Z
            shape = numpy.broadcast_shapes(*(c[0].shape for c in cols.values()), *(c[1].shape for c in cols.values()))
            bc = {k: (numpy.broadcast_to(v[0], shape), numpy.broadcast_to(v[1], shape)) for k, v in cols.items()}
            out = numpy.empty(shape)
            with mpmath.workdps(60):
                for idx in numpy.ndindex(shape):
                    u, z = bc[fa][0][idx], bc[fb][1][idx]
                    vf, xg = (bc[fa][1][idx], bc[fb][0][idx]) if split else (0., 0.)
                    rxy = sum(bc[k][0][idx] * bc[k][1][idx] for k in bc if k not in (fa, fb))
                    rxx = sum(bc[k][0][idx] ** 2 for k in bc if k != fa)
                    ryy = sum(bc[k][1][idx] ** 2 for k in bc if k != fb)
                    out[idx] = float(fn(u, z, vf, xg, rxy, rxx, ryy))
            return out
        return Kernel(core, scale=kernel._scale, loc=kernel._loc, outops=(('mul', factor),))

    assert kind == 'iso'
    # symbols: the differentiated coordinates and the sum R of the squares of the other fields' differences
    U, V, W, Z, R = sp.symbols('U V W Z R', real=True)          # x_f, y_f (field of the x side), x_g, y_g (field of the y side)
    same = xf == yf or a == 0 or b == 0
    if same:
        expr = sp.diff(build((U - V) ** 2 + R), *([U] * a + [V] * b))
        fn = mp_eval(expr, [U, V, R])
    else:
        expr = sp.diff(build((U - V) ** 2 + (W - Z) ** 2 + R), *([U] * a + [Z] * b))
        fn = mp_eval(expr, [U, V, W, Z, R])

    def core(x, y):
        x = numpy.asarray(x); y = numpy.asarray(y)
        names = x.dtype.names
        if names is None:
            cols = {None: (numpy.asarray(x, float), numpy.asarray(y, float))}
        else:
            cols = {nm: (numpy.asarray(x[nm], float), numpy.asarray(y[nm], float)) for nm in names}
        f1 = xf if a else yf
        f2 = None if same else yf
        shape = numpy.broadcast_shapes(*(c[0].shape for c in cols.values()), *(c[1].shape for c in cols.values()))
        bc = {k: (numpy.broadcast_to(v[0], shape), numpy.broadcast_to(v[1], shape)) for k, v in cols.items()}
        out = numpy.empty(shape)
        for idx in numpy.ndindex(shape):
            rest = sum((bc[k][0][idx] - bc[k][1][idx]) ** 2 for k in bc if k not in (f1, f2))
            u, v = bc[f1][0][idx], bc[f1][1][idx]
            args = [u, v] + ([bc[f2][0][idx], bc[f2][1][idx]] if f2 is not None else []) + [rest]
            coincide = rest == 0 and u == v and (f2 is None or args[2] == args[3])
            # a coincidence: the limit, from a small offset (error ~ the offset) in extended precision
            with mpmath.workdps(opts.get('dps', 120) if coincide else 120):
                if coincide:
                    args[0] = mpmath.mpf(float(u)) + mpmath.mpf(opts.get('offset', '1e-10'))
                out[idx] = float(fn(*args))
        return out
    return Kernel(core, scale=kernel._scale, loc=kernel._loc, outops=(('mul', factor),))
