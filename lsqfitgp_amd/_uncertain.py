"""
Minimal value-with-uncertainty scalar for the outputs of `empbayes_fit` and
`bayestree.bart` (the reference returns ``gvar`` objects there; gvar is not a
dependency of this package and correlated error propagation is out of scope of
the device path).  First-order propagation through the few operations the
callers need; the string form is gvar's compact ``mean(sdev)`` notation so that
printed results read like the reference's (``docs/userguide/hyper.rst:66-70``:
``2.16(17)``).
"""

import math


class UVal:
    """ a number with a standard deviation """

    __slots__ = ('mean', 'sdev')

    def __init__(self, mean, sdev=0.):
        self.mean = float(mean)
        self.sdev = abs(float(sdev))

    # -- first-order propagation (independent operands)
    @staticmethod
    def _parts(x):
        return (x.mean, x.sdev) if isinstance(x, UVal) else (float(x), 0.)

    def __add__(self, o):
        m, s = self._parts(o)
        return UVal(self.mean + m, math.hypot(self.sdev, s))

    __radd__ = __add__

    def __neg__(self):
        return UVal(-self.mean, self.sdev)

    def __sub__(self, o):
        return self + (-o if isinstance(o, UVal) else -float(o))

    def __rsub__(self, o):
        return (-self) + o

    def __mul__(self, o):
        m, s = self._parts(o)
        return UVal(self.mean * m, math.hypot(self.sdev * m, self.mean * s))

    __rmul__ = __mul__

    def __truediv__(self, o):
        m, s = self._parts(o)
        return UVal(self.mean / m, math.hypot(self.sdev / m, self.mean * s / (m * m)))

    def __rtruediv__(self, o):
        o = float(o)
        return UVal(o / self.mean, abs(o) * self.sdev / (self.mean * self.mean))

    def __pow__(self, e):
        e = float(e)
        return UVal(self.mean ** e, abs(e * self.mean ** (e - 1)) * self.sdev)

    def sqrt(self):
        r = math.sqrt(self.mean)
        return UVal(r, self.sdev / (2 * r) if r else math.inf)

    def exp(self):
        v = math.exp(self.mean)
        return UVal(v, v * self.sdev)

    def log(self):
        return UVal(math.log(self.mean), self.sdev / abs(self.mean))

    def __float__(self):
        return self.mean

    def __repr__(self):
        return str(self)

    def __str__(self):
        m, s = self.mean, self.sdev
        if not (math.isfinite(m) and math.isfinite(s)):
            return f'{m:g} +- {s:g}'
        if s == 0:
            return f'{m:g}(0)'
        # two significant digits of the uncertainty, the mean rounded to the same decimal place
        dec = 1 - math.floor(math.log10(s))
        if dec >= 2:
            digits = round(s * 10 ** dec)
            if digits >= 100:          # 0.0996 -> 0.10
                dec -= 1
                digits = round(s * 10 ** dec)
            if dec >= 2:
                return f'{m:.{dec}f}({digits})'
        if s < 9.95:
            return f'{m:.1f}({s:.1f})'
        return f'{m:.0f}({s:.0f})'


def sqrt(x):
    return x.sqrt() if isinstance(x, UVal) else math.sqrt(x)


def exp(x):
    return x.exp() if isinstance(x, UVal) else math.exp(x)


def log(x):
    return x.log() if isinstance(x, UVal) else math.log(x)
