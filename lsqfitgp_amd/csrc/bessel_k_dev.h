// bessel_k_dev.h -- Bessel functions of real order on the device: K_nu as the Matern kernel of real order needs it, J_nu as the Bessel
// kernel does, E_n on the imaginary axis as Color and Pink do (gram.hip: values and the scale tangents; gram_jet.hip: derivative
// observations).
#pragma once
#include <hip/hip_runtime.h>

namespace lgp_bessel {

// Matern of real order (_kernels/_matern.py:56-76 through _special/_bessel.py:72-82): 2 / Gamma(nu) (x / 2)^nu K_nu(x), x = sqrt(2 nu r2),
// 1 at x = 0.  K_nu by Temme's method (N. M. Temme, J. Comput. Phys. 19 (1975) 324): nu = mu + nl, |mu| <= 1/2; K_mu and K_{mu+1}
// from the series in x^2/4 for x < 2 and from Steed's continued fraction otherwise, then nl steps of the upward recurrence (stable
// for K).  The constants that depend on nu alone (1/Gamma(1 +- mu), the two Temme combinations of them, pi mu / sin(pi mu),
// 2/Gamma(nu)) come from the host.  Also returns x K_{nu-1}(x) in units of K's prefactor: dk/dx = -c (x/2)^nu K_{nu-1}(x).
struct MaternNu { double nu, mu, gampl, gammi, gam1, gam2, fact, c2; int nl; };
// K_mu(x), K_{mu+1}(x) and log(x / 2)
__device__ inline void bessel_k_pair(const MaternNu &c, double x, double &kmu, double &k1, double &lhalf) {
  const double eps = 1e-16;
  const double mu2 = c.mu * c.mu, xi = 1. / x, xi2 = 2. * xi;
  if (x < 2.) {
    const double d = log(xi2);
    lhalf = -d;
    double e = c.mu * d;
    const double sh = fabs(e) < eps ? 1. : sinh(e) / e;
    double ff = c.fact * (c.gam1 * cosh(e) + c.gam2 * sh * d);
    double sum = ff;
    e = exp(e);
    double p = 0.5 * e / c.gampl, q = 0.5 / (e * c.gammi), cc = 1., sum1 = p;
    const double q4 = 0.25 * x * x;
    for (int i = 1; i < 500; ++i) {
      ff = (i * ff + p + q) / (i * i - mu2);
      cc *= q4 / i;
      p /= i - c.mu;
      q /= i + c.mu;
      const double del = cc * ff;
      sum += del;
      sum1 += cc * (p - i * ff);
      if (fabs(del) < fabs(sum) * eps) break;
    }
    kmu = sum;
    k1 = sum1 * xi2;
  } else {
    lhalf = log(0.5 * x);
    double b = 2. * (1. + x), d = 1. / b, h = d, delh = d, q1 = 0., q2 = 1.;
    const double a1 = 0.25 - mu2;
    double q = a1, cc = a1, a = -a1, s = 1. + q * delh;
    for (int i = 2; i < 500; ++i) {
      a -= 2 * (i - 1);
      cc = -a * cc / i;
      const double qnew = (q1 - b * q2) / a;
      q1 = q2; q2 = qnew;
      q += cc * qnew;
      b += 2.;
      d = 1. / (b + a * d);
      delh = (b * d - 1.) * delh;
      h += delh;
      const double dels = q * delh;
      s += dels;
      if (fabs(dels / s) < eps) break;
    }
    h = a1 * h;
    kmu = sqrt(1.5707963267948966 / x) * exp(-x) / s;
    k1 = kmu * (c.mu + x + 0.5 - h) * xi;
  }
}

__device__ inline double matern_real(const MaternNu &c, double x, double *xdk) {
  const double xi = 1. / x, xi2 = 2. * xi;
  double kmu, k1, lhalf;
  bessel_k_pair(c, x, kmu, k1, lhalf);
  double km1 = k1 - 2. * c.mu * xi * kmu;                // K_{mu-1} = K_{1-mu} = K_{mu+1} - (2 mu / x) K_mu, used when nl == 0
  for (int i = 1; i <= c.nl; ++i) {
    const double kn = (c.mu + i) * xi2 * k1 + kmu;
    km1 = kmu; kmu = k1; k1 = kn;
  }
  // 2/Gamma(nu) and (x/2)^nu are both tiny where K_nu is huge (and the power huge where K_nu is tiny): pair the factors so that
  // no intermediate product leaves the normal range
  const double pw = exp(c.nu * lhalf);
  *xdk = (km1 > 1. ? (c.c2 * km1) * pw : (c.c2 * pw) * km1) * x;
  return kmu > 1. ? (c.c2 * kmu) * pw : (c.c2 * pw) * kmu;
}

// F_k = c2 (x / 2)^(nu - k) K_{nu - k}(x), k = 0 .. 4, x > 0: the kernel is G(q) = F_0(x) with x^2 = 2 nu q, and
// d/dx [(x/2)^m K_m] = -(x/2)^m K_{m-1} makes G^(k)(q) = (-nu / 2)^k F_k(x).  Orders below mu by the recurrence downwards,
// K_{m-1} = K_{m+1} - (2 m / x) K_m (K_{-m} = K_m): it cancels for small x, by a relative x^(2 mu) or so, where those orders only
// enter a derivative multiplied by matching powers of the distance.
__device__ inline void matern_ladder(const MaternNu &c, double x, double (&F)[5]) {
  const double xi2 = 2. / x;
  double kmu, k1, lhalf;
  bessel_k_pair(c, x, kmu, k1, lhalf);
  double w[5];                                           // K of orders o, o - 1, .., o - 4 with o = mu + (steps done)
  w[0] = kmu;
  w[1] = k1 - c.mu * xi2 * kmu;
  w[2] = w[0] - (c.mu - 1.) * xi2 * w[1];
  w[3] = w[1] - (c.mu - 2.) * xi2 * w[2];
  w[4] = w[2] - (c.mu - 3.) * xi2 * w[3];
  for (int i = 1; i <= c.nl; ++i) {
    const double kn = (c.mu + i) * xi2 * k1 + w[0];      // K_{mu+i+1}
    w[4] = w[3]; w[3] = w[2]; w[2] = w[1]; w[1] = w[0]; w[0] = k1;
    k1 = kn;
  }
  const double ih = exp(-lhalf);                         // 2 / x
  double pw = exp(c.nu * lhalf);
  for (int k = 0; k < 5; ++k) {
    F[k] = w[k] > 1. ? (c.c2 * w[k]) * pw : (c.c2 * pw) * w[k];
    pw *= ih;
  }
}

// Gamma(nu + 1) (x / 2)^-nu J_nu(x), real nu >= 0 (Bessel kernel, _kernels/_matern.py:102-119 through _special/_bessel.py:45-50): 1 at
// x = 0.  Three regions: the power series sum_k (-x^2/4)^k / (k! (nu+1)_k) while it does not cancel (x^2 <= 8 (nu + 1), or x <= 3);
// Hankel's asymptotic expansion above max(30, nu^2); between them Miller's backward recurrence from an even order N > x down to the
// fractional order mu = nu - floor(nu), normalised by (x/2)^mu = sum_k (mu + 2k) Gamma(mu + k) / k! J_{mu+2k}(x).  Against scipy's
// jv: <= 1e-14 absolute for nu in [0, 30], x in [0, 3000].
__device__ inline double bessel_jmod(double nu, int n, double gnu1, double gmu1, double x) {
  if (x == 0.) return 1.;
  if (x * x <= 8. * (nu + 1.) || x <= 3.) {
    const double q = -0.25 * x * x;
    double term = 1., sum = 1.;
    for (int k = 1; k < 300; ++k) {
      term *= q / ((double)k * (nu + (double)k));
      sum += term;
      if (fabs(term) < 1e-17 * fabs(sum)) break;
    }
    return sum;
  }
  if (x > fmax(30., nu * nu)) {
    const double mu4 = 4. * nu * nu;
    double P = 1., Q = 0., t = 1.;
    for (int k = 1; k < 60; ++k) {
      const double odd = (double)(2 * k - 1);
      const double tn = t * (mu4 - odd * odd) / ((double)k * 8. * x);
      if (fabs(tn) > fabs(t) && (double)k > nu) break;
      t = tn;
      const bool neg = ((k >> 1) & 1) != 0;
      if (k & 1) Q += neg ? -t : t;
      else P += neg ? -t : t;
      if (fabs(t) < 1e-17) break;
    }
    double sn, cs;
    sincos(x - (0.5 * nu + 0.25) * 3.141592653589793238462643383279502884, &sn, &cs);
    const double J = sqrt(0.6366197723675814 / x) * (P * cs - Q * sn);
    return gnu1 * exp(-nu * log(0.5 * x)) * J;
  }
  const double mu = nu - (double)n, big = fmax(x, nu);
  int N = (int)(big + 25. + 8. * cbrt(big));
  N += N & 1;
  const int K = N >> 1;
  double ck = exp(lgamma(mu + (double)K) - lgamma((double)K + 1.)) * (mu + 2. * (double)K);
  double fp = 0., f = 1e-300, S = 0., fn = 0.;
  const double tox = 2. / x;
  for (int j = N; j >= 0; --j) {
    if ((j & 1) == 0) {
      const int k = j >> 1;
      S += ck * f;
      if (k >= 2) ck = ck * (mu + (double)(2 * k - 2)) / (mu + (double)(2 * k)) * (double)k / (mu + (double)(k - 1));
      else if (k == 1) ck = gmu1;
    }
    if (j == n) fn = f;
    if (j > 0) {
      const double fm = (mu + (double)j) * tox * f - fp;
      fp = f; f = fm;
      if (fabs(f) > 1e250) { f *= 1e-250; fp *= 1e-250; S *= 1e-250; fn *= 1e-250; }
    }
  }
  return gnu1 * exp(-(double)n * log(0.5 * x)) * fn / S;
}

// Re E_n(-i x), x > 0, n >= 1: the generalised exponential integral on the imaginary axis (Color: (n - 1) Re E_n(-i x),
// _kernels/_spectral.py:73-111; Pink through Ci(x) = -Re E_1(-i x), :41-66).  For x <= 2 the power series of E_n (DLMF 8.19.8 /
// 6.6.6 for n = 1): sum_{m != n-1} -(i x)^m / ((m - n + 1) m!) + (i x)^(n-1) / (n-1)! (psi(n) - ln x + i pi/2), real part; above,
// the continued fraction e^(-z) / (z + n - 1 n / (z + n + 2 - 2 (n + 1) / (z + n + 4 - ...))) at z = -i x by the modified Lentz
// recurrence in complex arithmetic (92 steps at x = 2, 8 at x = 100).  The two agree to 4e-15 where they overlap (x in [1, 4],
// n = 1 .. 50); psi(n) comes from the host.  (The reference switches between the closed form 8.19.7 and a 20-term asymptotic
// series at a "knee" where both have lost digits -- 1e-4 at n = 20; the values here are good to the last few bits everywhere.)
// (IM: the imaginary part instead -- d/dx Re E_n(-i x) = -Im E_{n-1}(-i x), the scale tangent of Color)
template <bool IM = false>
__device__ inline double expn_imag_re(int n, double psin, double x) {
  const int nm1 = n - 1;
  if (x <= 2.) {
    const double a = psin - log(x), b = 1.5707963267948966;
    const int r = nm1 & 3;
    // Re / Im of i^(n-1) (a + i b)
    const double lead = IM ? (r == 0 ? b : (r == 1 ? a : (r == 2 ? -b : -a))) : (r == 0 ? a : (r == 1 ? -b : (r == 2 ? -a : b)));
    double ans = nm1 == 0 ? lead : (IM ? 0. : 1. / (double)nm1), fact = 1.;
    for (int m = 1; m < 200; ++m) {
      fact *= x / (double)m;
      if (m == nm1) ans += lead * fact;
      else if (((m & 1) != 0) == IM) ans -= ((m & 2) ? -fact : fact) / (double)(m - nm1);      // Re i^m = (-1)^(m/2), Im i^m = (-1)^((m-1)/2)
      if (((m & 1) != 0) == IM && m > 1 && fact * (1. + fabs(lead)) < 1e-17 * (fabs(ans) + 1e-300)) break;
    }
    return ans;
  }
  // complex Lentz: b = z + n, c = 1 / tiny, d = 1 / b, h = d; a_i = -i (n - 1 + i), b += 2
  double br = (double)n, bi = -x;
  double den = br * br + bi * bi;
  double dr = br / den, di = -bi / den;
  double cr = 1e300, ci = 0.;
  double hr = dr, hi = di;
  for (int i = 1; i < 400; ++i) {
    const double ai = -(double)i * (double)(nm1 + i);
    br += 2.;
    // d = 1 / (a d + b)
    const double tr = ai * dr + br, ti = ai * di + bi;
    den = tr * tr + ti * ti;
    dr = tr / den; di = -ti / den;
    // c = b + a / c
    den = cr * cr + ci * ci;
    const double qr = br + ai * cr / den, qi = bi - ai * ci / den;
    cr = qr; ci = qi;
    const double er = cr * dr - ci * di, ei = cr * di + ci * dr;             // del = c d
    const double nr = hr * er - hi * ei, ni = hr * ei + hi * er;
    hr = nr; hi = ni;
    if (fabs(er - 1.) + fabs(ei) < 1e-16) break;
  }
  double sn, cs;
  sincos(x, &sn, &cs);
  return IM ? hr * sn + hi * cs : hr * cs - hi * sn;                         // Im / Re of h e^(i x)
}

// The constants of `MaternNu` that depend on nu alone (host).  1 / Gamma(1 + z) = sum_k c_k z^(k-1) (Abramowitz & Stegun 6.1.34);
// Temme's gam2 and gam1 are its even and (minus its) odd part.
inline MaternNu matern_nu_constants(double nu) {
  MaternNu c;
  const int nl = (int)(nu + 0.5);
  const double mu = nu - nl, mu2 = mu * mu;
  static const double rg[26] = {1.0000000000000000, 0.5772156649015329, -0.6558780715202538, -0.0420026350340952, 0.1665386113822915,
    -0.0421977345555443, -0.0096219715278770, 0.0072189432466630, -0.0011651675918591, -0.0002152416741149, 0.0001280502823882,
    -0.0000201348547807, -0.0000012504934821, 0.0000011330272320, -0.0000002056338417, 0.0000000061160950, 0.0000000050020075,
    -0.0000000011812746, 0.0000000001043427, 0.0000000000077823, -0.0000000000036968, 0.0000000000005100, -0.0000000000000206,
    -0.0000000000000054, 0.0000000000000014, 0.0000000000000001};
  double gam1 = 0., gam2 = 0.;
  for (int k = 25; k >= 1; k -= 2) gam1 = gam1 * mu2 + rg[k];
  for (int k = 24; k >= 0; k -= 2) gam2 = gam2 * mu2 + rg[k];
  gam1 = -gam1;
  const double pimu = 3.141592653589793238462643383279502884 * mu;
  c.nu = nu; c.mu = mu; c.nl = nl;
  c.gampl = gam2 - mu * gam1; c.gammi = gam2 + mu * gam1; c.gam1 = gam1; c.gam2 = gam2;
  c.fact = fabs(pimu) < 1e-16 ? 1. : pimu / sin(pimu);
  c.c2 = nu == 0. ? 0. : 2. / tgamma(nu);
  return c;
}

}  // namespace lgp_bessel
