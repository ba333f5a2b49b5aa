"""
Host side of the Zeta kernel (``_kernels/_zeta.py:37-70``, ``_special/_zeta.py``): k(Δ) = Re Li_s(e^{2πiΔ}) / ζ(s), s = 1 + 2ν.

The reference evaluates the periodic zeta function through Hurwitz zeta series with guards against the cancellations near integer
s.  Here the device evaluates, for θ = 2π·(Δ mod 1 reflected into [0, 1/2]) ≤ π, the classical expansion of the polylogarithm on
the unit circle

    Re Li_s(e^{iθ}) = Γ(1-s) sin(πs/2) θ^(s-1) + Σ_n (-1)^n ζ(s-2n) θ^(2n) / (2n)!      (|θ| < 2π, s not an integer)

with the coefficients computed once per kernel on the host (this module): 36 terms (the ratio of consecutive terms is at most
1/4).  Γ(1-s) sin(πs/2) = π / (2 Γ(s) cos(πs/2)) is finite at even s; at odd s = 2m + 1 (integer ν) it has a pole that cancels
against the pole of ζ(s-2m): with s = 2m + 1 + ε the two terms are regrouped as θ^(2m) (A expm1(ε log θ) + D), D = A + c_m from
Laurent / Taylor expansions in ε (Stieltjes constants, polygamma), which tends to the logarithmic form
(-1)^m θ^(2m) (H_2m - log θ) / (2m)! at ε = 0.  For s >= 10 the Fourier series itself, 50 terms (as the reference,
``_special/_zeta.py:96-105,121-131``); ν = 0 is 1 at the integers and 0 elsewhere (``_kernels/_zeta.py:66-68``).
"""

import numpy
from scipy import special

# γ_0 .. γ_5: ζ(1 + ε) = 1/ε + Σ_k (-1)^k γ_k ε^k / k!
_STIELTJES = (0.5772156649015329, -0.0728158454836767, -0.00969036319287232, 0.00205383442030335, 0.00232537006546730, 0.000793323817301063)


def _laurent_at_one(e):
    """ ζ(1 + e) - 1/e for small e """
    return _STIELTJES[0] + sum((-1.) ** j * _STIELTJES[j] * e ** j / special.gamma(j + 1) for j in range(1, 6))


def _eta(z, n=40):
    """ Dirichlet eta by P. Borwein's algorithm 2 (2000): -1/d_n Σ_k (-1)^k (d_k - d_n) / (k + 1)^z, error ~ (3 + √8)^-n """
    k = numpy.arange(n + 1)
    d = numpy.cumsum(n * numpy.exp(special.gammaln(n + k) - special.gammaln(n - k + 1) - special.gammaln(2 * k + 1) + k * numpy.log(4.)))
    kk = numpy.arange(n)
    return -numpy.sum((-1.) ** kk * (d[kk] - d[n]) / (kk + 1.) ** z) / d[n]


def zeta_real(z):
    """ Riemann zeta of a real argument, z != 1 """
    if abs(z - 1) < 0.01:
        return 1 / (z - 1) + _laurent_at_one(z - 1)
    if z > 1:
        return float(special.zeta(z, 1))
    if z >= 0:
        return _eta(z) / (1 - 2. ** (1 - z))
    return 2 * (2 * numpy.pi) ** (z - 1) * numpy.sin(numpy.pi * z / 2) * special.gamma(1 - z) * zeta_real(1 - z)


NTERMS = 36
NFOURIER = 50
MODE_FOURIER, MODE_POWER, MODE_INTEGERS = 0, 1, 2


def coefficients(nu):
    """ what ``lgp_gram_pseries`` takes for Zeta(nu): ``(mode, uselog, table, (A, pw, eps, D))``, everything already divided by ζ(s):
    MODE_FOURIER:  k = Σ_j table[j-1] cos(j θ);  MODE_POWER:  k = Σ_n table[n] θ^(2n) + θ^pw (A g + D) with g = 1 (uselog 0: D = 0),
    log θ (uselog 1, eps = 0) or expm1(eps log θ) (uselog 1, eps != 0);  MODE_INTEGERS:  k = 1 where θ = 0 """
    nu = float(nu)
    s = 1 + 2 * nu
    if nu == 0:
        return MODE_INTEGERS, 0, numpy.zeros(0), (0., 0., 0., 0.)
    if s >= 10:
        j = numpy.arange(1., NFOURIER + 1)
        return MODE_FOURIER, 0, j ** -s / zeta_real(s), (0., 0., 0., 0.)
    zs = zeta_real(s)
    m = int(round((s - 1) / 2))
    e = s - (2 * m + 1)                                   # s = 2 m + 1 + e
    near = abs(e) < 1e-3                                  # (m = 0, s -> 1+ included: both poles are there too; found by the reference's
                                                          # golden values for s = 1 + 1e-15, tests/test_reference_special.py)
    c = numpy.zeros(NTERMS)
    for n in range(NTERMS):
        z = s - 2 * n
        if near and n == m:
            continue                                       # regrouped with the power term below
        if z < -20:                                        # reflection, with Γ(1 - z) / (2n)! through logarithms
            w = 1 - z
            c[n] = (-1.) ** n * 2 * (2 * numpy.pi) ** (z - 1) * numpy.sin(numpy.pi * z / 2) * numpy.exp(special.gammaln(w) - special.gammaln(2 * n + 1)) * float(special.zeta(w, 1))
        else:
            c[n] = (-1.) ** n * zeta_real(z) / special.gamma(2 * n + 1)
    if not near:
        A = numpy.pi / (2 * special.gamma(s) * numpy.cos(numpy.pi * s / 2))
        return MODE_POWER, 0, c / zs, (A / zs, s - 1, 0., 0.)
    sign = (-1.) ** m
    f2m = special.gamma(2 * m + 1)
    if e == 0:
        A = -sign / f2m                                   # times log θ
        D = sign * numpy.sum(1. / numpy.arange(1, 2 * m + 1)) / f2m
    else:
        xx = (numpy.pi * e / 2) ** 2
        logu = xx / 6 + xx ** 2 / 180 + xx ** 3 / 2835 + xx ** 4 / 37800      # log(x / sin x), x = π e / 2: as the quotient it is 1 + 4e-9 at
        u = numpy.exp(logu)                                                   # e = 1e-4 and its logarithm keeps 8 digits: 2.7e-13 in D, found
                                                                              # by the reference's golden values (tests/test_reference_special.py)
        dl = sum(e ** (j + 1) * special.polygamma(j, 2 * m + 1.) / special.gamma(j + 2) for j in range(6))      # lgamma(2m+1+e) - lgamma(2m+1)
        A = -sign * u * numpy.exp(-dl) / (e * f2m)        # = π / (2 Γ(s) cos(πs/2))
        D = sign / f2m * (-numpy.expm1(logu - dl) / e + _laurent_at_one(e))                                      # = A + c_m
    return MODE_POWER, 1, c / zs, (A / zs, 2. * m, e, D / zs)
