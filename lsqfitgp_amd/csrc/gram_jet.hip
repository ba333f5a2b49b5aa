// gram_jet.hip -- Gram blocks of DERIVED kernels: d^a/dx^a d^b/dy^b k(x, y), a, b <= 2, along one field per side, for the kernels
// of the reference whose core is an elementary expression.  Replaces what `kernel.linop('diff', xderiv, yderiv)` gives the
// reference through jax.grad of the core (src/lsqfitgp/_Kernel/_ops.py:115-211) when points are added with `deriv=`
// (_GP/_elements.py:158-242, 554-579).
//
// No closed forms per kernel and order: the core is evaluated ONCE in truncated Taylor arithmetic.  A value is a polynomial
// sum_{i, j <= 2} c_ij u^i v^j in the two perturbations (u of the x side's field, v of the y side's), i.e. an element of
// R[u, v] / (u^3, v^3); sums and products are those of the ring, an elementary function f acts through
// f(a) = sum_{k <= 4} f^(k)(a_00) / k! (a - a_00)^k (the nilpotent part has no term of total degree 0, so its fifth power vanishes).
// The derivative asked for is a! b! c_ab.  ExpQuad keeps its Hermite form (gram.hip, any fields, total order 4 per field) and
// Maternp in one dimension its polynomial form; Maternp over several fields comes here, as a function of s = (2p+1) r^2 whose
// derivatives e^-t R_k(t) the host derives (R_{k+1} = (R_k' - R_k) / (2 t), exact for k < p: no square root of a jet at r = 0).
// One thread per pair: derivative blocks are the small blocks of a problem, not its N x N matrix.
#include "common.h"
#include "bessel_k_dev.h"
#include <limits>

namespace {

constexpr int JD = 3;            // orders 0 .. 2 per side

template <typename T>
struct Jet {
  T c[JD][JD];
  __device__ __forceinline__ Jet() {}
  __device__ __forceinline__ explicit Jet(T v) {
#pragma unroll
    for (int i = 0; i < JD; ++i)
#pragma unroll
      for (int j = 0; j < JD; ++j) c[i][j] = T(0);
    c[0][0] = v;
  }
  __device__ __forceinline__ T val() const { return c[0][0]; }
};

template <typename T> __device__ __forceinline__ Jet<T> operator+(const Jet<T> &a, const Jet<T> &b) {
  Jet<T> r;
#pragma unroll
  for (int i = 0; i < JD; ++i)
#pragma unroll
    for (int j = 0; j < JD; ++j) r.c[i][j] = a.c[i][j] + b.c[i][j];
  return r;
}
template <typename T> __device__ __forceinline__ Jet<T> operator-(const Jet<T> &a, const Jet<T> &b) {
  Jet<T> r;
#pragma unroll
  for (int i = 0; i < JD; ++i)
#pragma unroll
    for (int j = 0; j < JD; ++j) r.c[i][j] = a.c[i][j] - b.c[i][j];
  return r;
}
template <typename T> __device__ __forceinline__ Jet<T> operator-(const Jet<T> &a) {
  Jet<T> r;
#pragma unroll
  for (int i = 0; i < JD; ++i)
#pragma unroll
    for (int j = 0; j < JD; ++j) r.c[i][j] = -a.c[i][j];
  return r;
}
template <typename T> __device__ __forceinline__ Jet<T> operator*(const Jet<T> &a, const Jet<T> &b) {
  Jet<T> r(T(0));
#pragma unroll
  for (int i = 0; i < JD; ++i)
#pragma unroll
    for (int j = 0; j < JD; ++j)
#pragma unroll
      for (int k = 0; k + i < JD; ++k)
#pragma unroll
        for (int l = 0; l + j < JD; ++l) r.c[i + k][j + l] += a.c[i][j] * b.c[k][l];
  return r;
}
template <typename T> __device__ __forceinline__ Jet<T> operator*(const Jet<T> &a, T s) {
  Jet<T> r;
#pragma unroll
  for (int i = 0; i < JD; ++i)
#pragma unroll
    for (int j = 0; j < JD; ++j) r.c[i][j] = a.c[i][j] * s;
  return r;
}
template <typename T> __device__ __forceinline__ Jet<T> operator*(T s, const Jet<T> &a) { return a * s; }
template <typename T> __device__ __forceinline__ Jet<T> operator+(const Jet<T> &a, T s) { Jet<T> r = a; r.c[0][0] += s; return r; }
template <typename T> __device__ __forceinline__ Jet<T> operator+(T s, const Jet<T> &a) { return a + s; }
template <typename T> __device__ __forceinline__ Jet<T> operator-(const Jet<T> &a, T s) { return a + (-s); }
template <typename T> __device__ __forceinline__ Jet<T> operator-(T s, const Jet<T> &a) { return (-a) + s; }

// f(a) from the Taylor coefficients f[k] = f^(k)(a_00) / k!, k = 0 .. 4
template <typename T> __device__ __forceinline__ Jet<T> compose(const Jet<T> &a, const T (&f)[5]) {
  Jet<T> n = a;
  n.c[0][0] = T(0);
  Jet<T> r = f[4] * n;                 // Horner in the nilpotent part
  r = (r + f[3]) * n;
  r = (r + f[2]) * n;
  r = (r + f[1]) * n;
  return r + f[0];
}
template <typename T> __device__ __forceinline__ Jet<T> jexp(const Jet<T> &a) {
  const T e = exp(a.val());
  const T f[5] = {e, e, e / T(2), e / T(6), e / T(24)};
  return compose(a, f);
}
template <typename T> __device__ __forceinline__ Jet<T> jsin(const Jet<T> &a) {
  const T s = sin(a.val()), c = cos(a.val());
  const T f[5] = {s, c, -s / T(2), -c / T(6), s / T(24)};
  return compose(a, f);
}
template <typename T> __device__ __forceinline__ Jet<T> jcos(const Jet<T> &a) {
  const T s = sin(a.val()), c = cos(a.val());
  const T f[5] = {c, -s, -c / T(2), s / T(6), c / T(24)};
  return compose(a, f);
}
template <typename T> __device__ __forceinline__ Jet<T> jcosh(const Jet<T> &a) {
  const T s = sinh(a.val()), c = cosh(a.val());
  const T f[5] = {c, s, c / T(2), s / T(6), c / T(24)};
  return compose(a, f);
}
template <typename T> __device__ __forceinline__ Jet<T> jsinh(const Jet<T> &a) {
  const T s = sinh(a.val()), c = cosh(a.val());
  const T f[5] = {s, c, s / T(2), c / T(6), s / T(24)};
  return compose(a, f);
}
// a^p, real p, a_00 > 0 (a_00 <= 0: zero, the continuation the kernels with a compact support want: (1 - t)_+^p)
template <typename T> __device__ __forceinline__ Jet<T> jpow(const Jet<T> &a, T p) {
  const T x = a.val();
  if (!(x > T(0))) return Jet<T>(T(0));
  const T v = pow(x, p), ix = T(1) / x;
  T f[5];
  f[0] = v;
  f[1] = f[0] * p * ix;
  f[2] = f[1] * (p - T(1)) * ix / T(2);
  f[3] = f[2] * (p - T(2)) * ix / T(3);
  f[4] = f[3] * (p - T(3)) * ix / T(4);
  return compose(a, f);
}
template <typename T> __device__ __forceinline__ Jet<T> jrecip(const Jet<T> &a) {
  const T ix = T(1) / a.val();
  const T f[5] = {ix, -ix * ix, ix * ix * ix, -ix * ix * ix * ix, ix * ix * ix * ix * ix};
  return compose(a, f);
}
// asin(a), |a_00| < 1
template <typename T> __device__ __forceinline__ Jet<T> jasin(const Jet<T> &a) {
  const T t = a.val(), u = T(1) / (T(1) - t * t), r = sqrt(u);        // u = 1 / (1 - t^2)
  const T f[5] = {asin(t), r, t * r * u / T(2), (T(1) + T(2) * t * t) * r * u * u / T(6), (T(9) * t + T(6) * t * t * t) * r * u * u * u / T(24)};
  return compose(a, f);
}
// |a| as an even function continued from the right: sign(0) = +1 (the kernels that take |delta| and are derivable at 0 have
// h'(0) = 0, so that d^2/d delta^2 h(|delta|) at 0 is h''(0+), which this gives; a sign that vanishes at 0 would give 0)
template <typename T> __device__ __forceinline__ Jet<T> jabs(const Jet<T> &a) { return a.val() >= T(0) ? a : -a; }
template <typename T> __device__ __forceinline__ Jet<T> jmin(const Jet<T> &a, const Jet<T> &b) { return a.val() <= b.val() ? a : b; }
template <typename T> __device__ __forceinline__ Jet<T> jmax(const Jet<T> &a, const Jet<T> &b) { return a.val() <= b.val() ? b : a; }

// sin(pi a) / (pi a) and its derivatives: the series of the even function below |pi x| = 1, sines and cosines above
template <typename T> __device__ __forceinline__ Jet<T> jsinc(const Jet<T> &a) {
  const T PI = T(3.141592653589793238462643383279502884);
  const T z = PI * a.val();
  T g[5];                                   // derivatives of sin(z) / z with respect to z, divided by k!
  if (fabs(z) < T(1)) {
    // sin z / z = sum_m (-1)^m z^(2m) / (2m+1)!;  k-th derivative term by term, 12 terms (z^24 / 25! = 6e-26)
    const T z2 = z * z;
    T d0 = T(0), d1 = T(0), d2 = T(0), d3 = T(0), d4 = T(0);
    T c = T(1);                             // (-1)^m / (2m+1)!
    T zp = T(1);                            // z^(2m)
    for (int m = 0; m < 13; ++m) {
      const T e = T(2 * m);
      d0 += c * zp;
      if (m >= 1) {
        d1 += c * e * zp;                                   // times 1 / z below
        d2 += c * e * (e - T(1)) * zp;                      // times 1 / z^2
      }
      if (m >= 2) {
        d3 += c * e * (e - T(1)) * (e - T(2)) * zp;         // times 1 / z^3
        d4 += c * e * (e - T(1)) * (e - T(2)) * (e - T(3)) * zp;
      }
      zp *= z2;
      c = -c / (T(2 * m + 2) * T(2 * m + 3));
    }
    // (the divisions by powers of z: exact cancellation of the leading powers is avoided by dividing the accumulated sums, whose
    //  lowest term is z^2 (d1, d2) or z^4 (d3, d4); at z = 0 the limits are -1/3 for d2 and 1/5 for d4, 0 for the odd ones)
    if (z == T(0)) {
      g[0] = T(1); g[1] = T(0); g[2] = T(-1) / T(6); g[3] = T(0); g[4] = T(1) / T(120);
    } else {
      g[0] = d0; g[1] = d1 / z; g[2] = d2 / (z * z) / T(2); g[3] = d3 / (z * z * z) / T(6); g[4] = d4 / (z * z * z * z) / T(24);
    }
  } else {
    const T s = sin(z), c = cos(z), iz = T(1) / z;
    // f = s / z; f' = (c - f) / z ... by the recurrence f_(k+1) = (d^k/dz^k cos-or-sin term - (k+1) f_k) / z from z f = sin z
    const T t0 = s * iz;
    const T t1 = (c - t0) * iz;
    const T t2 = (-s - T(2) * t1) * iz;
    const T t3 = (-c - T(3) * t2) * iz;
    const T t4 = (s - T(4) * t3) * iz;
    g[0] = t0; g[1] = t1; g[2] = t2 / T(2); g[3] = t3 / T(6); g[4] = t4 / T(24);
  }
  T f[5];
  T pk = T(1);
  for (int k = 0; k < 5; ++k) { f[k] = g[k] * pk; pk *= PI; }      // chain rule of z = pi a
  return compose(a, f);
}

struct JetArgs {
  int kernel, p, d, n, m;
  int fx, fy, ax, by;          // field differentiated on each side (-1: none) and the orders (0 .. 2)
  int has_loc, has_scale, xy, posabs;
  double loc, scale, factor;   // factor: the product of the multiplicative output operations (added constants drop out of a derivative)
  double par[8];
  double coef[5][24];          // Maternp: R_k as Laurent polynomials in t, entry i the coefficient of t^(i - 7);  Wendland: row 0 = the
                               // polynomial, highest power first
  long ldx, ldy, ldo;
};

template <typename T, typename S>
__device__ __forceinline__ S jet_core(const JetArgs &g, const S &r2, const S &X, const S &Y) {
  // r2: squared distance over the fields (isotropic kernels); X, Y: the two (transformed) coordinates of one-variable kernels;
  // stationary kernels of one variable take delta = X - Y
  const int k = g.kernel;
  if (k == LGP_KERNEL_EXPQUAD) return jexp(T(-0.5) * r2);
  if (k == LGP_KERNEL_GAMMAEXP) return jexp(-r2);                                    // gamma = 2 only (the host refuses the others)
  if (k == LGP_KERNEL_CAUCHY) return jpow(T(1) + r2 * T(1. / g.par[1]), T(g.par[2])); // alpha = 2: (1 + r2 / beta)^(-beta / 2)
  if (k == LGP_KERNEL_MATERNP) {
    // s = (2p+1) r2, t = sqrt(s): f^(k)(s) = e^-t R_k(t), R_k a Laurent polynomial (negative powers beyond the p-th derivative:
    // they meet the matching powers of the nilpotent part, which carries delta^2 ~ t^2 per order, and the products stay finite)
    const S s = T(2 * g.p + 1) * r2;
    const T t = sqrt(s.val() + T(1e-30)), e = exp(-t), it = T(1) / t;
    T f[5];
    T fact = T(1);
    for (int q = 0; q < 5; ++q) {
      T poly = T(g.coef[q][23]);
      for (int i = 22; i >= 7; --i) poly = poly * t + T(g.coef[q][i]);
      T neg = T(0);
      for (int i = 0; i < 7; ++i) neg = neg * it + T(g.coef[q][i]);     // t^-7 .. t^-1: Horner in 1 / t, lowest power first
      f[q] = e * (poly + neg * it) / fact;
      fact *= T(q + 1);
    }
    return compose(s, f);
  }
  if (k == LGP_KERNEL_MATERN) {
    // real order: G(q) = c2 (x/2)^nu K_nu(x), x^2 = 2 nu q, has G^(k)(q) = (-nu / 2)^k F_k(x) (bessel_k_dev.h); at q = 0 the limits
    // c2 (-nu / 2)^k Gamma(nu - k) / 2 from the host (row 1 of the table; 0 beyond the orders that exist: they meet powers of a
    // vanishing distance, and the host has refused the derivatives that the process does not have)
    const double q = (double)r2.val();
    T f[5];
    if (q > 0.) {
      lgp_bessel::MaternNu c;
      c.nu = g.coef[0][0]; c.mu = g.coef[0][1]; c.gampl = g.coef[0][2]; c.gammi = g.coef[0][3];
      c.gam1 = g.coef[0][4]; c.gam2 = g.coef[0][5]; c.fact = g.coef[0][6]; c.c2 = g.coef[0][7]; c.nl = g.p;
      double F[5];
      lgp_bessel::matern_ladder(c, sqrt(2. * c.nu * q), F);
      double fk = 1., fact = 1.;
      for (int i = 0; i < 5; ++i) {
        f[i] = T(F[i] * fk / fact);
        fk *= -0.5 * c.nu;
        fact *= (double)(i + 1);
      }
    } else {
      for (int i = 0; i < 5; ++i) f[i] = T(g.coef[1][i]);
    }
    return compose(r2, f);
  }
  if (k == LGP_KERNEL_BESSEL) {
    // k_nu(x) = Gamma(nu + 1) (x / 2)^-nu J_nu(x), x^2 = s^2 q: k_nu' = -x / (2 (nu + 1)) k_{nu+1} makes
    // G^(k)(q) = (-s^2 / 4)^k / ((nu + 1) .. (nu + k)) k_{nu+k}(x), an entire function of q (1 at x = 0 for every order)
    const double nu = g.par[0], s2 = g.par[3], x = sqrt(s2 * fmax((double)r2.val(), 0.));
    T f[5];
    double c = 1., gn = g.par[1];
    for (int i = 0; i < 5; ++i) {
      f[i] = T(c * lgp_bessel::bessel_jmod(nu + i, g.p + i, gn, g.par[2], x));
      gn *= nu + i + 1.;
      c *= -0.25 * s2 / ((nu + i + 1.) * (double)(i + 1));
    }
    return compose(r2, f);
  }
  if (k == LGP_KERNEL_CAUSALEXPQUAD) return jexp(T(-0.5) * r2);                       // alpha = 0 only (the host refuses the others): r2 of the softened distances
  if (k == LGP_KERNEL_NNKERNEL) {                        // r2 carries x.y, X carries x.x, Y carries y.y (all fields); par = {q}
    const T q = T(g.par[0]);
    const S denom = (T(1) + T(2) * (X + q)) * (T(1) + T(2) * (Y + q));
    return T(0.63661977236758134308) * jasin(T(2) * (r2 + q) * jrecip(denom));
  }
  if (k == LGP_KERNEL_CONSTANT) return S(T(1));
  if (k == LGP_KERNEL_LINEAR) return r2;                                             // (r2 carries sum_f X_f Y_f for this kernel)
  const S delta = X - Y;
  if (k == LGP_KERNEL_COS) return jcos(delta);
  if (k == LGP_KERNEL_PERIODIC) {
    const S q = jsin(T(0.5) * delta) * T(1. / g.par[0]);
    return jexp(T(-2) * (q * q));
  }
  if (k == LGP_KERNEL_SINC) return jsinc(delta);
  if (k == LGP_KERNEL_COLOR) {
    // k(x) = (n - 1) Re E_n(-i x), x = |delta|: E_n' = -E_{n-1} makes d^m/dx^m E_n(-i x) = i^m E_{n-m}(-i x); psi(n - m) from the host
    // (par[m]); at x = 0, E_k = 1 / (k - 1).  Orders with n - m < 2 only meet derivatives the host has refused.
    const S a = jabs(delta);
    const double x = (double)a.val();
    T f[5];
    double fact = 1.;
    for (int m = 0; m < 5; ++m) {
      const int kk = g.p - m;
      double v = 0.;
      if (kk >= 2 || (kk >= 1 && x > 0.)) {
        const bool im = (m & 1) != 0;
        if (x == 0.) v = im ? 0. : 1. / (double)(kk - 1);
        else v = im ? lgp_bessel::expn_imag_re<true>(kk, g.par[m], x) : lgp_bessel::expn_imag_re<false>(kk, g.par[m], x);
        if (m == 1 || m == 2) v = -v;                    // Re(i E) = -Im E, Re(-E) = -Re E, Re(-i E) = Im E, Re(E)
      }
      f[m] = T((double)(g.p - 1) * v / fact);
      fact *= (double)(m + 1);
    }
    return compose(a, f);
  }
  if (k == LGP_KERNEL_PSERIES) {
    // Zeta: the expansion that the value core sums (gram.hip, lsqfitgp_amd/_zeta.py), differentiated term by term in
    // theta = 2 pi t, t the distance of |delta| to the nearest integer.  p & 15 = mode (0: sum_j c_j cos(j theta), 1: sum_n c_n
    // theta^(2n) + theta^pw (A g + D)), p & 16 = with the logarithm; par = {A, pw, eps, D}; coef = the table, ntab = p >> 8.
    const S ad = jabs(delta) + (sizeof(T) == 8 ? T(2.220446049250313e-16) : T(1.1920929e-07f));
    const S xx = ad - floor(ad.val());
    const S tt = jmin(xx, T(1) - xx);
    const S th = tt * T(6.283185307179586476925286766559);
    const double t = (double)th.val();
    const int mode = g.p & 15, uselog = (g.p >> 4) & 1, ntab = g.p >> 8;
    const double *tab = &g.coef[0][0];
    double dv[5] = {0., 0., 0., 0., 0.};                             // d^m k / d theta^m
    if (mode == 0) {
      for (int j = ntab; j >= 1; --j) {
        double sn, cs;
        sincos((double)j * t, &sn, &cs);
        const double c = tab[j - 1], j1 = (double)j, j2 = j1 * j1;
        dv[0] += c * cs; dv[1] -= c * j1 * sn; dv[2] -= c * j2 * cs; dv[3] += c * j2 * j1 * sn; dv[4] += c * j2 * j2 * cs;
      }
    } else {
      // the even power series: Horner per order on the falling factorials of 2n
      for (int m = 0; m < 5; ++m) {
        double acc = 0.;
        for (int n = ntab - 1; n >= 0; --n) {
          const int e = 2 * n - m;                                   // exponent of theta in the m-th derivative of theta^(2n)
          if (e < 0) continue;
          double ff = 1.;
          for (int q = 0; q < m; ++q) ff *= (double)(2 * n - q);
          acc += tab[n] * ff * (e == 0 ? 1. : pow(t, (double)e));
        }
        dv[m] = acc;
      }
      // the power term theta^pw (A g + D): g = 1; log theta (eps = 0); expm1(eps log theta) = theta^eps - 1
      const double A = g.par[0], pw = g.par[1], eps = g.par[2], D = g.par[3];
      auto addpow = [&](double coef, double ex) {                      // coef theta^ex and its derivatives
        double ff = coef;
        for (int m = 0; m < 5; ++m) {
          const double e = ex - (double)m;
          if (t > 0.) dv[m] += ff * pow(t, e);
          else if (e == 0.) dv[m] += ff;                               // (theta = 0: positive powers vanish; negative ones only meet derivatives the process does not have)
          ff *= e;
        }
      };
      if (!uselog) addpow(A + D, pw);
      else if (eps != 0.) { addpow(A, pw + eps); addpow(D - A, pw); }
      else {
        addpow(D, pw);
        if (t > 0.) {                                                  // A theta^pw log theta: d^m = theta^(pw-m) (a_m log theta + b_m), a' = a (pw - m), b' = b (pw - m) + a
          double am = A, bm = 0.;
          const double lt = log(t);
          for (int m = 0; m < 5; ++m) {
            dv[m] += pow(t, pw - (double)m) * (am * lt + bm);
            bm = bm * (pw - (double)m) + am;
            am *= pw - (double)m;
          }
        }
      }
    }
    const T f[5] = {T(dv[0]), T(dv[1]), T(dv[2] / 2.), T(dv[3] / 6.), T(dv[4] / 24.)};
    return compose(th, f);
  }
  if (k == LGP_KERNEL_PINK) {
    // k(x) = (Ci(a x) - Ci(x)) / L, a = 1 + dw, L = log1p(dw), x = |delta|; derivable once: Ci' = cos / x makes k' and k'' elementary
    // (their series in x below 1e-2, where the closed forms cancel); where x dw < sqrt(eps), cos(x (1 + dw / 2)) as the value core does
    const S a = jabs(delta);
    const double x = (double)a.val(), dw = g.par[0], L = g.par[1], aa = 1. + dw;
    double k0, k1, k2;
    if (x * dw < g.par[2]) {
      const double m = 1. + 0.5 * dw;
      k0 = cos(x * m); k1 = -m * sin(x * m); k2 = -m * m * k0;
    } else {
      k0 = (lgp_bessel::expn_imag_re(1, -0.5772156649015329, x) - lgp_bessel::expn_imag_re(1, -0.5772156649015329, x * aa)) / L;
      if (x < 1e-2) {
        // cos(a x) - cos x = sum_j (-1)^j (a^(2j) - 1) x^(2j) / (2j)!
        double t1 = 0., t2 = 0., a2j = 1., x2j2 = 1., fac = 1.;          // x^(2j-2), (2j)!
        for (int j = 1; j < 8; ++j) {
          a2j *= aa * aa;
          fac *= (double)(2 * j - 1) * (double)(2 * j);
          const double c = ((j & 1) ? -1. : 1.) * (a2j - 1.) / fac;
          t1 += c * x2j2 * x;
          t2 += c * (double)(2 * j - 1) * x2j2;
          x2j2 *= x * x;
        }
        k1 = t1 / L; k2 = t2 / L;
      } else {
        const double ca = cos(aa * x), c1 = cos(x), ix = 1. / x;
        k1 = (ca - c1) * ix / L;
        k2 = ((-aa * sin(aa * x) + sin(x)) * ix - (ca - c1) * ix * ix) / L;
      }
    }
    const T f[5] = {T(k0), T(k1), T(0.5 * k2), T(0), T(0)};
    return compose(a, f);
  }
  if (k == LGP_KERNEL_CELERITE) {                                                    // B = gamma (derivable once)
    const S a = jabs(delta);
    return jexp(T(-g.par[0]) * a) * (jcos(a) + T(g.par[1]) * jsin(a));
  }
  if (k == LGP_KERNEL_HARMONIC) {
    const S tau = jabs(delta), tauQ = tau * T(1. / g.par[0]);
    if (g.p == 0) {
      const S pexp = jexp(T(g.par[2]) * tauQ), mexp = jexp(T(-(1. + g.par[1])) * tauQ);
      return T(0.5) * (pexp + mexp + (pexp - mexp) * T(1. / g.par[1]));
    } else if (g.p == 1) {
      const S et = T(g.par[1]) * tauQ;
      return jexp(-tauQ) * (jcosh(et) + jsinh(et) * T(1. / g.par[1]));
    } else if (g.p == 2) {
      return (T(1) + tauQ) * jexp(-tauQ) + jexp(-tauQ) * (T(1. - g.par[0]) * (tau * tau) * (T(1) + tau * T(1. / 3.)));
    } else {
      const S et = T(g.par[1]) * tauQ;
      return jexp(-tauQ) * (jcos(et) + jsin(et) * T(1. / g.par[1]));
    }
  }
  if (k == LGP_KERNEL_WENDLAND) {                                                    // one-dimensional points: r = |delta| + eps
    const S r = jabs(delta) + (sizeof(T) == 8 ? T(2.220446049250313e-16) : T(1.1920929e-07f));
    S poly(T(g.coef[0][0]));
    for (int i = 1; i <= g.p; ++i) poly = poly * r + T(g.coef[0][i]);
    return jpow(T(1) - r, T(g.par[0])) * poly;
  }
  if (k == LGP_KERNEL_CIRCULAR) {
    const S a = jabs(delta) + (sizeof(T) == 8 ? T(2.220446049250313e-16) : T(1.1920929e-07f));
    const S x = a - floor(a.val());
    const S t = jmin(x, T(1) - x);
    return (T(1) + t * T(g.par[0] / g.par[1])) * jpow(T(1) - t * T(1. / g.par[1]), T(g.par[0]));
  }
  if (k == LGP_KERNEL_DECAYING) return jpow(X + Y + T(1), T(-g.par[0]));
  if (k == LGP_KERNEL_WIENERINTEGRAL) {
    const S a = jmin(X, Y), b = jmax(X, Y);
    return T(0.5) * (a * a) * (b - a * T(1. / 3.));
  }
  if (k == LGP_KERNEL_TAYLOR) {
    // h(w) = I0(2 sqrt(w)) = sum_j w^j / j!^2 (J0 of the same argument where w = x y < 0): h^(m)(w) = sum_j w^j / (j! (j + m)!) =
    // |w|^(-m/2) I_m or J_m (2 sqrt|w|).  The series while it neither overflows the patience nor cancels (|w| <= 30); beyond, the
    // library's I0, I1 with the recurrence upwards (harmless at arguments >= 11 for four orders) and its J_m
    const S w = X * Y;
    const double w0 = (double)w.val(), aw = fabs(w0);
    T f[5];
    if (aw <= 30.) {
      double mfact = 1.;
      for (int m = 0; m < 5; ++m) {
        double term = 1. / mfact, sum = term;                       // j = 0: 1 / (0! m!)
        for (int j = 1; j < 200; ++j) {
          term *= w0 / ((double)j * (double)(j + m));
          sum += term;
          if (fabs(term) < 1e-18 * fabs(sum)) break;
        }
        f[m] = T(sum / mfact);                                      // h^(m) / m!
        mfact *= (double)(m + 1);
      }
    } else {
      const double z = 2. * sqrt(aw), iz = 1. / z;
      double b[5];
      if (w0 > 0.) {
        b[0] = cyl_bessel_i0(z); b[1] = cyl_bessel_i1(z);
        for (int m = 1; m < 4; ++m) b[m + 1] = b[m - 1] - 2. * (double)m * iz * b[m];
      } else {
        for (int m = 0; m < 5; ++m) b[m] = jn(m, z);
      }
      double pw = 1., mfact = 1.;
      const double isq = 2. * iz;                                   // 1 / sqrt|w|
      for (int m = 0; m < 5; ++m) {
        f[m] = T(b[m] * pw / mfact);
        pw *= isq;
        mfact *= (double)(m + 1);
      }
    }
    return compose(w, f);
  }
  if (k == LGP_KERNEL_FRACBROWNIAN) return X * Y;                                    // H = 1, K = 1: (x^2 + y^2 - (x - y)^2) / 2
  if (k == LGP_KERNEL_STATIONARYFRACBROWNIAN) return S(T(1));                        // H = 1: ((d+1)^2 + (d-1)^2 - 2 d^2) / 2
  return S(T(NAN));
}

template <typename T>
__global__ void __launch_bounds__(256) gram_jet_kernel(const T *__restrict__ x, const T *__restrict__ y, T *out, JetArgs g) {
  typedef Jet<T> S;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)g.n * g.m) return;
  const int i = (int)(idx / g.m), j = (int)(idx % g.m);
  const T sc = g.has_scale ? T(g.scale) : T(1), lc = g.has_loc ? T(g.loc) : T(0);
  S r2(T(0)), X(T(0)), Y(T(0));
  for (int f = 0; f < g.d; ++f) {
    // the reference transforms each argument: (x - loc) / scale (_Kernel/_ops.py:292-326), true division
    T xv = x[(long)f * g.ldx + i], yv = y[(long)f * g.ldy + j];
    if (g.has_loc) { xv -= lc; yv -= lc; }
    if (g.has_scale) { xv /= sc; yv /= sc; }
    S xf(xv), yf(yv);
    if (f == g.fx) xf.c[1][0] = T(1);            // u: the perturbation of x along its field (in transformed units)
    if (f == g.fy) yf.c[0][1] = T(1);            // v: of y
    if (g.xy == 3) {                                 // the three dot products over the fields
      r2 = r2 + xf * yf;
      X = X + xf * xf;
      Y = Y + yf * yf;
    } else if (g.xy == 2) {
      r2 = r2 + xf * yf;
    } else {
      S df = xf - yf;
      if (g.posabs) df = jabs(df) + (sizeof(T) == 8 ? T(2.220446049250313e-16) : T(1.1920929e-07f));
      r2 = r2 + df * df;
    }
    if (f == 0 && g.xy != 3) { X = xf; Y = yf; }
  }
  const S v = jet_core<T, S>(g, r2, X, Y);
  const T fact[3] = {T(1), T(1), T(2)};
  // chain rule of the transformation: 1 / scale per derivative
  T chain = T(g.factor);
  for (int q = 0; q < g.ax + g.by; ++q) chain /= sc;
  out[(long)i * g.ldo + j] = v.c[g.ax][g.by] * fact[g.ax] * fact[g.by] * chain;
}

template <typename T>
int gram_jet_launch(lgp_ctx *ctx, int kernel, int p, int npar, const T *hpar, int ncoef, const T *hcoef, int d, const T *dx, size_t ldx, size_t n,
                    const T *dy, size_t ldy, size_t m, int has_loc, T loc, int has_scale, T scale, T factor, int fx, int ax, int fy, int by, T *dout,
                    size_t ldo) {
  LGP_REQUIRE(ctx != nullptr && (n == 0 || m == 0 || (dx != nullptr && dy != nullptr && dout != nullptr)), "null pointer");
  LGP_REQUIRE(d >= 1 && d <= 64 && npar >= 0 && npar <= 8 && (npar == 0 || hpar != nullptr), "bad kernel parameters");
  LGP_REQUIRE(ax >= 0 && ax <= 2 && by >= 0 && by <= 2 && fx >= -1 && fx < d && fy >= -1 && fy < d, "derivative orders 0 .. 2 along one field per side");
  LGP_REQUIRE((ax == 0 || fx >= 0) && (by == 0 || fy >= 0), "a derivative needs its field");
  LGP_REQUIRE(!has_scale || (scale > T(0) && scale < T(INFINITY)), "scale must be positive and finite");
  LGP_REQUIRE(ncoef >= 0 && ncoef <= 5 * 24 && (ncoef == 0 || hcoef != nullptr), "bad coefficient table");
  const bool onevar = kernel == LGP_KERNEL_COS || kernel == LGP_KERNEL_PERIODIC || kernel == LGP_KERNEL_SINC || kernel == LGP_KERNEL_CELERITE ||
                      kernel == LGP_KERNEL_HARMONIC || kernel == LGP_KERNEL_WENDLAND || kernel == LGP_KERNEL_CIRCULAR || kernel == LGP_KERNEL_DECAYING ||
                      kernel == LGP_KERNEL_WIENERINTEGRAL || kernel == LGP_KERNEL_FRACBROWNIAN || kernel == LGP_KERNEL_STATIONARYFRACBROWNIAN ||
                      kernel == LGP_KERNEL_TAYLOR || kernel == LGP_KERNEL_COLOR || kernel == LGP_KERNEL_PINK || kernel == LGP_KERNEL_PSERIES;
  const bool iso = kernel == LGP_KERNEL_EXPQUAD || kernel == LGP_KERNEL_GAMMAEXP || kernel == LGP_KERNEL_CAUCHY || kernel == LGP_KERNEL_MATERNP ||
                   kernel == LGP_KERNEL_CONSTANT || kernel == LGP_KERNEL_LINEAR || kernel == LGP_KERNEL_MATERN || kernel == LGP_KERNEL_BESSEL ||
                   kernel == LGP_KERNEL_CAUSALEXPQUAD || kernel == LGP_KERNEL_NNKERNEL;
  LGP_REQUIRE(onevar || iso, "no derivative core for this kernel");
  LGP_REQUIRE(!onevar || d == 1, "a kernel of one variable takes one-dimensional points");
  LGP_REQUIRE(kernel != LGP_KERNEL_MATERNP || (p >= 0 && p <= 16 && ncoef == 5 * 24), "Maternp: order 0 .. 16 and the five polynomials");
  LGP_REQUIRE(kernel != LGP_KERNEL_WENDLAND || (p >= 0 && p <= 16 && ncoef >= p + 1), "Wendland: the polynomial");
  if (n == 0 || m == 0) return LGP_OK;
  LGP_REQUIRE(ldx >= n && ldy >= m && ldo >= m && n * m < (size_t(1) << 40), "bad dimensions");
  DeviceGuard dg(ctx->device);
  JetArgs g;
  memset(&g, 0, sizeof(g));
  g.kernel = kernel; g.p = p; g.d = d; g.n = (int)n; g.m = (int)m;
  g.fx = ax ? fx : -1; g.fy = by ? fy : -1; g.ax = ax; g.by = by;
  g.has_loc = has_loc; g.has_scale = has_scale; g.loc = (double)loc; g.scale = (double)scale; g.factor = (double)factor;
  g.xy = kernel == LGP_KERNEL_LINEAR ? 2 : (kernel == LGP_KERNEL_NNKERNEL ? 3 : 0);
  g.posabs = kernel == LGP_KERNEL_CAUSALEXPQUAD ? 1 : 0;           // input='posabs' (_Kernel/_isotropic.py:67-78)
  for (int i = 0; i < npar; ++i) g.par[i] = (double)hpar[i];
  for (int i = 0; i < ncoef; ++i) g.coef[i / 24][i % 24] = (double)hcoef[i];
  LGP_REQUIRE(kernel != LGP_KERNEL_PSERIES || (npar == 4 && (p >> 8) <= 5 * 24 && ncoef >= (p >> 8) && ((p & 15) == 0 || (p & 15) == 1)),
              "PSeries: p = mode | uselog << 4 | ntable << 8, {A, pw, eps, D}, the table as coefficients");
  LGP_REQUIRE(kernel != LGP_KERNEL_NNKERNEL || (npar == 1 && hpar[0] > T(0)), "NNKernel takes {sigma0^2}");
  if (kernel == LGP_KERNEL_PINK) {                       // hpar = {dw}: par = {dw, log1p(dw), sqrt(eps)} as the value core (_spectral.py:62-66)
    LGP_REQUIRE(npar == 1 && hpar[0] >= T(0) && hpar[0] < std::numeric_limits<T>::infinity(), "Pink takes {dw}, dw >= 0");
    g.par[0] = (double)hpar[0]; g.par[1] = log1p((double)hpar[0]); g.par[2] = sqrt((double)std::numeric_limits<T>::epsilon());
  }
  if (kernel == LGP_KERNEL_COLOR) {                      // p = n: psi(n - m), m = 0 .. 4, for the series of E_{n-m}
    LGP_REQUIRE(npar == 0 && p >= 2, "Color takes the integer n >= 2 as p");
    for (int m = 0; m < 5; ++m) {
      double psi = -0.5772156649015329;
      for (int j = 1; j < p - m; ++j) psi += 1. / (double)j;
      g.par[m] = psi;
    }
  }
  if (kernel == LGP_KERNEL_BESSEL) {                     // hpar = {nu}: as the value core (gram.hip), par = {nu, Gamma(nu + 1), Gamma(1 + nu - floor(nu)), (2 + nu / 2)^2}
    LGP_REQUIRE(npar == 1 && hpar[0] >= T(0) && hpar[0] <= T(96), "Bessel takes {nu} in [0, 96]");
    const double nu = (double)hpar[0];
    const int fl = (int)nu;
    LGP_REQUIRE(d <= 2 * (fl + 1), "Bessel(nu) is a kernel in at most 2 (floor(nu) + 1) dimensions");       // _matern.py:97-99
    g.p = fl;
    g.par[0] = nu; g.par[1] = tgamma(nu + 1.); g.par[2] = tgamma(1. + nu - (double)fl); g.par[3] = (2. + nu / 2.) * (2. + nu / 2.);
  }
  if (kernel == LGP_KERNEL_MATERN) {                     // hpar = {nu}: the constants of the Bessel evaluation and the limits at r = 0
    LGP_REQUIRE(npar == 1 && hpar[0] > T(0) && hpar[0] <= T(100), "Matern takes {nu} in (0, 100]");
    const double nu = (double)hpar[0];
    const lgp_bessel::MaternNu c = lgp_bessel::matern_nu_constants(nu);
    g.p = c.nl;
    g.coef[0][0] = c.nu; g.coef[0][1] = c.mu; g.coef[0][2] = c.gampl; g.coef[0][3] = c.gammi;
    g.coef[0][4] = c.gam1; g.coef[0][5] = c.gam2; g.coef[0][6] = c.fact; g.coef[0][7] = c.c2;
    double fk = 1., fact = 1.;
    for (int i = 0; i < 5; ++i) {
      g.coef[1][i] = nu - i > 0. ? 0.5 * (c.c2 * tgamma(nu - i)) * fk / fact : 0.;
      fk *= -0.5 * nu;
      fact *= (double)(i + 1);
    }
  }
  g.ldx = (long)ldx; g.ldy = (long)ldy; g.ldo = (long)ldo;
  const size_t total = n * m;
  hipLaunchKernelGGL((gram_jet_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, dx, dy, dout, g);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

}  // namespace

extern "C" int lgp_gram_deriv_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int ncoef, const double *hcoef, int d, const double *dx,
                                  size_t ldx, size_t n, const double *dy, size_t ldy, size_t m, int has_loc, double loc, int has_scale, double scale,
                                  double factor, int fx, int ax, int fy, int by, double *dout, size_t ldo) {
  return gram_jet_launch<double>(ctx, kernel, p, npar, hpar, ncoef, hcoef, d, dx, ldx, n, dy, ldy, m, has_loc, loc, has_scale, scale, factor, fx, ax, fy, by,
                                 dout, ldo);
}
extern "C" int lgp_gram_deriv_f32(lgp_ctx *ctx, int kernel, int p, int npar, const float *hpar, int ncoef, const float *hcoef, int d, const float *dx,
                                  size_t ldx, size_t n, const float *dy, size_t ldy, size_t m, int has_loc, float loc, int has_scale, float scale,
                                  float factor, int fx, int ax, int fy, int by, float *dout, size_t ldo) {
  return gram_jet_launch<float>(ctx, kernel, p, npar, hpar, ncoef, hcoef, d, dx, ldx, n, dy, ldy, m, has_loc, loc, has_scale, scale, factor, fx, ax, fy, by,
                                dout, ldo);
}
