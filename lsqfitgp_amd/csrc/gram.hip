// gram.hip -- Gram-matrix build for gfx950: the `kernel(x[:,None], y[None,:])` call of
// src/lsqfitgp/_GP/_elements.py:575-577 for the three kernels of the hot path.
//
//   isotropic  : u = (x - loc)/scale per field (_Kernel/_ops.py:292-326), r2 = sum_f (u_f - v_f)^2 in field order
//                (_Kernel/_isotropic.py:61-81, _Kernel/_util.py:74-99), then
//                  ExpQuad  exp(-1/2 * r2)                                     (_kernels/_basic.py:75)
//                  Maternp  x = sqrt((2p+1) r2 + 1e-30); exp(-x) * Horner(x)   (_kernels/_matern.py:48-49, _special/_bessel.py:103-110)
//   BART       : depth <= 2 closed forms of the altinput correlation with `repeat` folding
//                (_kernels/_bart.py:628-757), digamma at integers from a harmonic-number table.
//
// Both are HBM-write bound (8 or 4 bytes per pair out, O(d) bytes per point in): a workgroup owns a
// 32 x 512 output tile, the transformed coordinates of its 32 rows and 512 columns are staged once in LDS
// (so (x-loc)/scale is evaluated once per point, exactly as the reference does before differencing), every
// lane produces 16 contiguous bytes per row so a wave stores 1 KiB (fp64: 128 columns) contiguously.
// This file is compiled with -ffp-contract=off: the reference evaluates square-then-add without fused
// multiply-add and the entries are compared to a few ulp.
#include "common.h"
#include "bessel_k_dev.h"
#include <limits>

namespace {

constexpr int GT_ROWS = 32;
constexpr int GT_THREADS = 256;
constexpr int MAX_OPS = 8;
constexpr int MAX_P = 16;

template <typename T>
struct OutOps {
  int n;
  int code[MAX_OPS];
  T val[MAX_OPS];
  __device__ __forceinline__ T apply(T v) const {
    for (int i = 0; i < n; ++i) v = (code[i] == LGP_OP_MUL) ? v * val[i] : v + val[i];
    return v;
  }
  // derivative of apply(v) with respect to v: the product of the multiplications
  __device__ __forceinline__ T slope() const {
    T f = T(1);
    for (int i = 0; i < n; ++i) f = (code[i] == LGP_OP_MUL) ? f * val[i] : f;
    return f;
  }
};

template <typename T>
struct IsoArgs {
  const T *x;
  const T *y;
  T *out;
  T *dsc;          // optional second output: d out / d scale (same shape and leading dimension)
  long ldx, ldy, ldo;
  int n, m, d;
  int kernel, p;
  int has_loc, has_scale;
  T loc, scale;
  T coef[MAX_P];   // Horner coefficients c_{k+1}/c_k, k = p-1..0 in evaluation order
  T twop1;         // 2p+1
  T par[4];        // GammaExp: gamma, gamma/2;  Cauchy: alpha/2, beta, -beta/alpha  (host-evaluated like the reference's Python floats)
                   // Wendland: nu + k;  CausalExpQuad: alpha / 4
  int flag;        // GammaExp: gamma == 2;  Cauchy: alpha == 2  (the branches that act on r2 itself);  MaternpDiff: odd total order
  int xy;          // 0; 1: kernels of x and y themselves, one variable: the core receives y in place of r2 and x in place of d0; 2: d0 = sum_f x_f y_f
  int posabs;      // input='posabs' of the reference (_Kernel/_isotropic.py:67-78): per field (|x - y| + eps)^2, the core takes the root
  int cbase;       // fused kernel, column-range mode: first column of the launch (the last is m - 1)
  int cyc_nb, cyc_stride, wloc;   // ... block-cyclic columns: local block q (width cyc_nb) starts at global column cbase + q cyc_stride; wloc local columns
  const T *table;  // LagTable: device array of the values at the integer lags 0 .. ntable - 1, `fill` beyond
  int ntable;
  T fill;
  int dord[8];     // ExpQuadDiff: total derivative order (x side + y side) along each of the first 8 fields
  int lower_only;  // fused kernel: leave the entries above the diagonal outside the tiles that straddle it unwritten
  OutOps<T> ops;
};

template <typename T> struct VecOf;
template <> struct VecOf<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int W = 2; };
template <> struct VecOf<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int W = 4; };

// ---- exp(-x) and sqrt for the two cores that fill the large Gram matrices (ExpQuad, Maternp), fp64.  The library versions spend a
// third of their instructions on cases these callers cannot reach (the square root scales its argument against subnormals and
// checks for zero / infinity; the exponential handles both signs and overflow) and on copying the polynomial's coefficients from
// registers every time (v_fmac needs its addend in the destination): 19 + 32 instructions per entry, of ~117 in the fused Gram
// kernel's loop.  Here: 8 + 17 and one clamp, the coefficients in scalar registers (a table in constant memory, read once per wave).
__constant__ double LGP_EXP_Q[10] = {          // exp(r) = 1 + r (1 + r q(r)) on |r| <= ln 2 / 2, q of degree 9 (highest first): Chebyshev fit of
    2.510038549551032e-08, 2.7620088445409746e-07, 2.7557268459997064e-06, 2.4801521295954376e-05, 0.00019841269863053618,      // (exp(r) - 1 - r) / r^2
    0.0013888888917213717, 0.008333333333330062, 0.04166666666662413, 0.16666666666666669, 0.5000000000000001};                // (mpmath, 60 digits):
                                                                                        // 0.07 ulp of approximation error with these doubles
// exp(-x) for 0 <= x <= 1000 or NaN (the callers clamp)
template <typename T>
__device__ __forceinline__ T lgp_exp_neg_raw(T x) { return exp(-x); }
template <>
__device__ __forceinline__ double lgp_exp_neg_raw<double>(double x) {
  const double n = __builtin_rint(x * -1.4426950408889634);        // rint(-x log2 e) <= 0
  double r = __builtin_fma(n, -6.93147180369123816490e-01, -x);    // -x - n ln 2, ln 2 in two pieces (n ln2_hi is exact)
  r = __builtin_fma(n, -1.90821492927058770002e-10, r);
  double p = LGP_EXP_Q[0];
#pragma unroll
  for (int k = 1; k < 10; ++k) {
    // (three-address form with the coefficient in scalar registers: left to itself the compiler keeps the table in vector
    //  registers and copies each coefficient into the destination of a v_fmac, one v_mov_b64 per step)
    const double c = LGP_EXP_Q[k];
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(r), "s"(c));
  }
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return ldexp(p, (int)n);
}
// exp(-x) for x >= 0 (+inf gives 0: beyond 745.2 the result is zero; the comparison keeps a NaN)
template <typename T>
__device__ __forceinline__ T lgp_exp_neg(T x) { return exp(-x); }
template <>
__device__ __forceinline__ double lgp_exp_neg<double>(double x) { return lgp_exp_neg_raw<double>(!(x > 800.0) ? x : 800.0); }

#ifndef LGP_SQRT_TWO_CORRECTIONS
#define LGP_SQRT_TWO_CORRECTIONS 0
#endif
// sqrt(a) for a > 0 in the normal range, not infinite (NaN stays NaN): the callers add 1e-30 and clamp
template <typename T>
__device__ __forceinline__ T lgp_sqrt_pos(T a) { return sqrt(a); }
template <>
__device__ __forceinline__ double lgp_sqrt_pos<double>(double a) {
  const double y = __builtin_amdgcn_rsq(a);                        // ~23 bits
  double g = a * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, a);
#if LGP_SQRT_TWO_CORRECTIONS
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, a);
#endif
  return __builtin_fma(d, h, g);               // (one residual correction: g is good to 2^-45 before it; the library's second one only decides the last rounding)
}
// x = sqrt(a), a > 0, limited to 800 (exp(-x) is zero from 745.2 on, whatever polynomial multiplies it; keeps a NaN), and exp(-x)
template <typename T>
__device__ __forceinline__ T lgp_sqrt_exp(T a, T *e) {
  const T x = sqrt(a);
  *e = exp(-x);
  return x;
}
template <>
__device__ __forceinline__ double lgp_sqrt_exp<double>(double a, double *e) {
  const double x = lgp_sqrt_pos<double>(!(a > 640000.0) ? a : 640000.0);
  *e = lgp_exp_neg_raw<double>(x);
  return x;
}

template <typename T>
__device__ __forceinline__ T matern_real_of(const IsoArgs<T> &g, T r2, T *xdk) {
  if (g.flag) { *xdk = T(0); return r2 == T(0) ? T(1) : T(0); }          // nu = 0: _matern.py:74 "the correct limit is white"
  const double x = sqrt((double)g.twop1 * (double)r2);
  double dk = 0., k = 1.;
  if (x > 0.) {
    lgp_bessel::MaternNu c;
    c.nu = (double)g.coef[0]; c.mu = (double)g.coef[1]; c.gampl = (double)g.coef[2]; c.gammi = (double)g.coef[3];
    c.gam1 = (double)g.coef[4]; c.gam2 = (double)g.coef[5]; c.fact = (double)g.coef[6]; c.c2 = (double)g.coef[7]; c.nl = g.p;
    k = lgp_bessel::matern_real(c, x, &dk);
    if (!(k <= 1.)) {
      // x so small that K_nu overflows (x < 0.06 at nu = 100, 1e-8 at nu = 33, 1e-76 at nu = 4): the series
      // sum_k (-x^2/4)^k / (k! (nu-1) ... (nu-k)); its companion in x^(2 nu) is far below the last bit there
      const double q = 0.25 * x * x, nu = c.nu;
      k = nu > 4. ? 1. - q / (nu - 1.) * (1. - q / (2. * (nu - 2.)) * (1. - q / (3. * (nu - 3.)))) : 1.;
      dk = nu > 4. ? 2. * q / (nu - 1.) * (1. - q / (nu - 2.)) : 0.;
    }
  }
  *xdk = T(dk);
  return T(k);
}

// the cores that need loops of their own (one compile-time core, CORE_SPECIAL, for all of them)
template <typename T>
__device__ __forceinline__ T special_core(const IsoArgs<T> &g, T r2) {
  if (g.kernel == LGP_KERNEL_BESSEL)                     // par = {nu, Gamma(nu + 1), Gamma(1 + nu - floor(nu))}, p = floor(nu), twop1 = (2 + nu/2)^2
    return T(lgp_bessel::bessel_jmod((double)g.par[0], g.p, (double)g.par[1], (double)g.par[2], sqrt((double)g.twop1 * (double)r2)));
  const double x = (double)sqrt(r2);                      // one variable: sqrt(d^2) = |d| exactly
  if (g.kernel == LGP_KERNEL_PSERIES) {
    // even 1-periodic functions of the difference given by host-computed coefficients (Zeta kernel, _kernels/_zeta.py:37-70):
    // theta = 2 pi (|d| mod 1, reflected into [0, 1/2]); p = 0: sum_j table[j-1] cos(j theta); p = 1: sum_n table[n] theta^(2n) +
    // theta^pw (A g + D), g = 1 | log theta | expm1(eps log theta) (flag: with the logarithm); p = 2: 1 where theta = 0.
    // par = {A, pw, eps, D}
    double y = x - floor(x);
    if (y > 0.5) y = 1. - y;
    const double th = 6.283185307179586476925286766559 * y;
    if (g.p == 2) return y == 0. ? T(1) : T(0);
    if (!(x < 1e300)) return T(NAN);
    if (g.p == 0) {
      double acc = 0.;
      for (int j = g.ntable; j >= 1; --j) acc += (double)g.table[j - 1] * cos((double)j * th);    // smallest terms first
      return T(acc);
    }
    const double t2 = th * th;
    double acc = 0.;
    for (int n = g.ntable - 1; n >= 0; --n) acc = acc * t2 + (double)g.table[n];
    const double A = (double)g.par[0], pw = (double)g.par[1], eps = (double)g.par[2];
    if (th > 0.) {
      const double lt = log(th);
      const double gl = g.flag ? (eps == 0. ? lt : expm1(eps * lt)) : 1.;
      acc += exp(pw * lt) * (A * gl + (double)g.par[3]);
    } else if (pw == 0. && g.flag && eps > 0.) {
      acc += (double)g.par[3] - A;                        // s -> 1+ (m = 0): theta^0 (A expm1(eps log 0) + D) = D - A = zeta(s) / zeta(s)
    }
    return T(acc);
  }
  if (g.kernel == LGP_KERNEL_COLOR) {                     // par = {psi(n)}, p = n
    if (x == 0.) return T(1);
    return T((double)(g.p - 1) * lgp_bessel::expn_imag_re(g.p, (double)g.par[0], x));
  } else {                                                // Pink: par = {dw, log1p(dw), sqrt(eps)}
    const double dw = (double)g.par[0];
    if (x * dw < (double)g.par[2]) return T(cos(x * (1. + 0.5 * dw)));
    return T((lgp_bessel::expn_imag_re(1, -0.5772156649015329, x) - lgp_bessel::expn_imag_re(1, -0.5772156649015329, x * (1. + dw))) / (double)g.par[1]);
  }
}

// -x k'(x) of the same cores, x the (transformed) distance: times 1 / scale it is dk / dscale (x = distance / scale)
template <typename T>
__device__ __forceinline__ T special_core_mxdk(const IsoArgs<T> &g, T r2) {
  if (g.kernel == LGP_KERNEL_BESSEL) {                    // k_nu' = -X / (2 (nu + 1)) k_{nu+1}(X), X = s x: -x dk/dx = X^2 / (2 (nu + 1)) k_{nu+1}(X)
    const double nu = (double)g.par[0], X2 = (double)g.twop1 * (double)r2;
    return T(X2 / (2. * (nu + 1.)) * lgp_bessel::bessel_jmod(nu + 1., g.p + 1, (double)g.par[1] * (nu + 1.), (double)g.par[2], sqrt(X2)));
  }
  const double x = (double)sqrt(r2);
  if (x == 0.) return T(0);
  if (g.kernel == LGP_KERNEL_COLOR)                       // k' = -(n - 1) Im E_{n-1}(-i x); par[1] = psi(n - 1)
    return T(x * (double)(g.p - 1) * lgp_bessel::expn_imag_re<true>(g.p - 1, (double)g.par[1], x));
  if (g.kernel == LGP_KERNEL_PINK) {                      // Ci'(x) = cos x / x
    const double dw = (double)g.par[0];
    if (x * dw < (double)g.par[2]) return T(x * (1. + 0.5 * dw) * sin(x * (1. + 0.5 * dw)));
    return T(-(cos(x * (1. + dw)) - cos(x)) / (double)g.par[1]);
  }
  return T(NAN);                                          // (PSeries: no tangent; refused by the launcher)
}

// d0: the signed difference of the (transformed) first coordinates, used by the derivative kernel only
// NU: with the real-order Matern branch (the pairs kernel; the Gram kernels have a compile-time core for it, CORE_MATERNNU, so that
// its loops do not weigh on the registers of the other generic cores)
template <typename T, bool NU = false>
__device__ __forceinline__ T iso_core(const IsoArgs<T> &g, T r2, T d0 = T(0), T d1 = T(0)) {
  if (g.kernel == LGP_KERNEL_EXPQUAD) {
    return lgp_exp_neg<T>(T(0.5) * r2);
  } else if (g.kernel == LGP_KERNEL_EXPQUAD_DIFF) {
    // mixed derivative of exp(-|u|^2 / 2), u = x - y field by field: along field f, a_f times in x and b_f times in y, it is
    // prod_f (-1)^{a_f} He_{a_f + b_f}(u_f) exp(-|u|^2 / 2) (probabilists' Hermite polynomials; the signs and the 1 / scale^order
    // of the chain rule are an output operation of the caller).  d0 arrives as the product of the Hermite factors (iso_aux).
    // What jax's elementwise_grad of the core gives for _Kernel/_ops.py:115-211 (`diff`) on ExpQuad, _kernels/_basic.py:75.
    return d0 * lgp_exp_neg<T>(T(0.5) * r2);
  } else if (g.kernel == LGP_KERNEL_MATERNP_DIFF) {
    // derivative of total order q of K(u) = exp(-t) P_p(t), t = sqrt((2p+1) u^2 + 1e-30), in one dimension (d0 = u):
    // K^(q)(u) = (c sign u)^q exp(-t) Q_q(t), Q_0 = P_p, Q_{q+1} = Q_q' - Q_q; the coefficients of Q_q (degree g.p, ascending)
    // come from the host, c^q and the chain-rule factors are an output operation of the caller; flag: q odd
    T et;
    const T t = lgp_sqrt_exp<T>(g.twop1 * r2 + T(1e-30), &et);
    T poly = g.coef[g.p];
    for (int k = g.p - 1; k >= 0; --k) poly = poly * t + g.coef[k];
    const T v = et * poly;
    return g.flag ? (d0 > T(0) ? v : (d0 < T(0) ? -v : T(0))) : v;
  } else if (g.kernel == LGP_KERNEL_WENDLAND) {          // _kernels/_wendland.py:32-84: (1 - r)_+^(nu + k) poly(r), r < 1
    const T r = sqrt(r2);
    T poly = g.coef[0];
    for (int k = 1; k <= g.p; ++k) poly = poly * r + g.coef[k];     // polyval: highest power first
    return r < T(1) ? pow(T(1) - r, g.par[0]) * poly : T(0);
  } else if (g.kernel == LGP_KERNEL_CAUSALEXPQUAD) {     // _kernels/_basic.py:346-360: erfc(alpha / 4 r) exp(-1/2 r^2)
    const T r = sqrt(r2);
    return erfc(g.par[0] * r) * exp(T(-0.5) * (r * r));
  } else if (g.kernel == LGP_KERNEL_LOG) {               // _kernels/_basic.py:381-392: log1p(r) / r
    const T r = sqrt(r2);
    return log1p(r) / r;
  } else if (g.kernel == LGP_KERNEL_CONSTANT) {          // _kernels/_basic.py:34-46
    return T(1);
  } else if (g.kernel == LGP_KERNEL_WHITE) {             // _kernels/_basic.py:48-59: d0 is the product over the fields of (x_f == y_f)
    return d0;
  } else if (g.kernel == LGP_KERNEL_EXPON) {             // _kernels/_basic.py:254-269: d0 is the signed difference of the one field
    return exp(-fabs(d0));
  } else if (g.kernel == LGP_KERNEL_HOLEEFFECT) {        // _kernels/_basic.py:299-310
    const T a = fabs(d0);
    return (T(1) - a) * exp(-a);
  } else if (g.kernel == LGP_KERNEL_PERIODIC) {          // _kernels/_basic.py:187-208
    const T q = sin(d0 / T(2)) / g.par[0];
    return exp(T(-2) * (q * q));
  } else if (g.kernel == LGP_KERNEL_COS) {               // _kernels/_spectral.py:25-38
    return cos(d0);
  } else if (g.kernel == LGP_KERNEL_SINC) {              // _kernels/_spectral.py:113-122, _special/_sinc.py:25-32
    const T a = fabs(d0) + (sizeof(T) == 8 ? T(2.220446049250313e-16) : T(1.1920929e-07f));
    const T pa = T(3.141592653589793238462643383279502884) * a;
    if (a < T(0.1)) {                                    // six terms of sum (-1)^m z^m / (2 m + 1)!, z = (pi a)^2
      const T z = pa * pa;
      return T(1) + z * (T(-1) / T(6) + z * (T(1) / T(120) + z * (T(-1) / T(5040) + z * (T(1) / T(362880) + z * (T(-1) / T(39916800))))));
    }
    return sin(pa) / pa;
  } else if (g.kernel == LGP_KERNEL_CELERITE) {          // _kernels/_celerite.py:34-52: par = {gamma, B}, input='abs'
    const T a = fabs(d0);
    return exp(-g.par[0] * a) * (cos(a) + g.par[1] * sin(a));
  } else if (g.kernel == LGP_KERNEL_HARMONIC) {          // _kernels/_celerite.py:54-121: par = {Q, eta Q, sqrt(1 - Q^2) - 1}, p = branch
    const T tau = fabs(d0), tauQ = tau / g.par[0];
    if (g.p == 0) {                                      // Q < 1/2
      const T pexp = exp(g.par[2] * tauQ), mexp = exp(-(T(1) + g.par[1]) * tauQ);
      return (pexp + mexp + (pexp - mexp) / g.par[1]) / T(2);
    } else if (g.p == 1) {                               // 1/2 <= Q < 1
      const T et = g.par[1] * tauQ;
      return exp(-tauQ) * (cosh(et) + sinh(et) / g.par[1]);
    } else if (g.p == 2) {                               // Q = 1: (1 + tau) exp(-tau)  (the second term of _harmonic vanishes)
      return (T(1) + tauQ) * exp(-tauQ) + exp(-tau / g.par[0]) * (T(1) - g.par[0]) * (tau * tau) * (T(1) + tau / T(3));
    } else {                                             // Q > 1
      const T et = g.par[1] * tauQ;
      return exp(-tauQ) * (cos(et) + sin(et) / g.par[1]);
    }
  } else if (g.kernel == LGP_KERNEL_CIRCULAR) {          // _kernels/_wendland.py:88-113: par = {tau, c}, input='posabs'
    const T a = fabs(d0) + (sizeof(T) == 8 ? T(2.220446049250313e-16) : T(1.1920929e-07f));
    const T x = a - floor(a);                            // delta % 1 for delta > 0
    const T t = fmin(x, T(1) - x);
    return (T(1) + g.par[0] * t / g.par[1]) * pow(fmax(T(1) - t / g.par[1], T(0)), g.par[0]);
  } else if (g.kernel == LGP_KERNEL_LAGTABLE) {          // _kernels/_arma.py:31-63,180-189: a table over the integer lag |x - y|
    const T a = fabs(d0);
    return a < (T)g.ntable ? g.table[(long)a] : g.fill;  // (a NaN lag compares false: the fill value)
  } else if (g.kernel == LGP_KERNEL_LINEAR) {            // _kernels/_basic.py:76-91: d0 = sum_f x_f y_f
    return d0;
  } else if (g.kernel == LGP_KERNEL_WIENER) {            // _kernels/_randomwalk.py:26-41 (from here on: x = d0, y = r2, one variable)
    return fmin(d0, r2);
  } else if (g.kernel == LGP_KERNEL_ORNSTEINUHLENBECK) { // _kernels/_randomwalk.py:112-132
    return exp(-fabs(d0 - r2)) - exp(-(d0 + r2));
  } else if (g.kernel == LGP_KERNEL_BROWNIANBRIDGE) {    // _kernels/_randomwalk.py:134-151
    return fmin(d0, r2) - d0 * r2;
  } else if (g.kernel == LGP_KERNEL_DECAYING) {          // _kernels/_basic.py:362-379: 1 / (x + y + 1)^alpha
    return T(1) / pow(d0 + r2 + T(1), g.par[0]);
  } else if (g.kernel == LGP_KERNEL_FRACBROWNIAN) {      // _kernels/_randomwalk.py:46-68: par = {2 H, K, 1 / 2^K, 2 H K}
    return g.par[2] * (pow(pow(fabs(d0), g.par[0]) + pow(fabs(r2), g.par[0]), g.par[1]) - pow(fabs(d0 - r2), g.par[3]));
  } else if (g.kernel == LGP_KERNEL_WIENERINTEGRAL) {    // _kernels/_randomwalk.py:93-110
    const T a = fmin(d0, r2), b = fmax(d0, r2);
    return T(0.5) * (a * a) * (b - a / T(3));
  } else if (g.kernel == LGP_KERNEL_GIBBS) {             // _kernels/_basic.py:156-185: d0 = s(x), d1 = s(y) (the last coordinate row), r2 over the others
    const T den = d0 * d0 + d1 * d1;
    return sqrt(T(2) * d0 * d1 / den) * exp(-r2 / den);
  } else if (g.kernel == LGP_KERNEL_TAYLOR) {            // _kernels/_basic.py:393-408: I0(2 sqrt(x y)), J0 where x y < 0
    const T mul = d0 * r2, val = T(2) * sqrt(fabs(mul));
    return mul >= T(0) ? T(cyl_bessel_i0((double)val)) : T(j0((double)val));
  } else if (g.kernel == LGP_KERNEL_STATIONARYFRACBROWNIAN) {   // _kernels/_randomwalk.py:156-172 (stationary: d0 = delta); par = {2 H}
    return T(0.5) * (pow(fabs(d0 + T(1)), g.par[0]) + pow(fabs(d0 - T(1)), g.par[0]) - T(2) * pow(fabs(d0), g.par[0]));
  } else if (NU && g.kernel == LGP_KERNEL_MATERN) {      // _kernels/_matern.py:56-76
    T unused;
    return matern_real_of<T>(g, r2, &unused);
  } else if (NU && (g.kernel == LGP_KERNEL_COLOR || g.kernel == LGP_KERNEL_PINK || g.kernel == LGP_KERNEL_BESSEL || g.kernel == LGP_KERNEL_PSERIES)) {
    return special_core<T>(g, r2);
  } else if (g.kernel == LGP_KERNEL_GAMMAEXP) {          // _kernels/_basic.py:111-113
    return g.flag ? exp(-r2) : exp(-pow(r2, g.par[1]));
  } else if (g.kernel == LGP_KERNEL_CAUCHY) {            // _kernels/_basic.py:340-344
    const T power = g.flag ? r2 : pow(r2, g.par[0]);
    return pow(T(1) + power / g.par[1], g.par[2]);
  } else {
    T r2s = g.twop1 * r2;
    T e;
    const T x = lgp_sqrt_exp<T>(r2s + T(1e-30), &e);
    T poly = T(1);
    for (int k = 0; k < g.p; ++k) poly = T(1) + poly * g.coef[k] * T(2) * x;
    return e * poly;
  }
}

// what the derivative cores need beside r2: ExpQuadDiff the product over the fields of the Hermite factors He_{order_f}(u_f),
// MaternpDiff (one field) the signed difference itself
template <typename T>
__device__ __forceinline__ T iso_aux(const IsoArgs<T> &g, int f, T xv, T yv, T df, T aux) {
  if (g.xy == 1) return xv;
  if (g.xy == 2) return f == 0 ? xv * yv : aux + xv * yv;
  if (g.kernel == LGP_KERNEL_EXPQUAD_DIFF) {
    const int o = f < 8 ? g.dord[f] : 0;
    const T u2 = df * df;
    T he = T(1);
    if (o == 1) he = df;
    else if (o == 2) he = u2 - T(1);
    else if (o == 3) he = df * (u2 - T(3));
    else if (o == 4) he = u2 * (u2 - T(6)) + T(3);
    return (f == 0 ? T(1) : aux) * he;
  }
  if (g.kernel == LGP_KERNEL_WHITE) return (f == 0 ? T(1) : aux) * (df == T(0) ? T(1) : T(0));
  return f == 0 ? df : aux;
}
// the per-field term of r2: the squared difference, or (|difference| + eps)^2 for input='posabs'
template <typename T>
__device__ __forceinline__ T iso_sq(const IsoArgs<T> &g, T df, T yv) {
  if (g.xy == 1) return yv;
  if (g.posabs) {
    const T a = fabs(df) + (sizeof(T) == 8 ? T(2.220446049250313e-16) : T(1.1920929e-07f));
    return a * a;
  }
  return df * df;
}

// Compile-time selection of the core for the kernels that evaluate n x m pairs: CORE_GENERIC keeps the run-time branches of
// iso_core; the others fix the kernel (and the Matern order, so that the Horner loop unrolls and its coefficients become
// immediates) -- the inner loop of the C3 Gram build lost its branch on g.kernel and its loop on g.p per element.
constexpr int CORE_GENERIC = 0, CORE_EXPQUAD = 1, CORE_MATERN1 = 2, CORE_MATERN2 = 3, CORE_MATERN3 = 4, CORE_MATERNNU = 5, CORE_SPECIAL = 6;

template <typename T, int CORE>
__device__ __forceinline__ T iso_core_t(const IsoArgs<T> &g, T r2, T d0 = T(0), T d1 = T(0)) {
  if constexpr (CORE == CORE_EXPQUAD) {
    return lgp_exp_neg<T>(T(0.5) * r2);
  } else if constexpr (CORE == CORE_MATERN1 || CORE == CORE_MATERN2 || CORE == CORE_MATERN3) {
    constexpr int P = CORE == CORE_MATERN1 ? 1 : (CORE == CORE_MATERN2 ? 2 : 3);
    const T r2s = T(2 * P + 1) * r2;
    T e;
    const T x = lgp_sqrt_exp<T>(r2s + T(1e-30), &e);
    T poly = T(1);
#pragma unroll
    for (int k = 0; k < P; ++k) poly = T(1) + poly * (g.coef[k] * T(2)) * x;    // (bit for bit iso_core's poly * coef * 2 * x: doubling is exact, and
                                                                                 //  the doubled coefficient is uniform -- one multiplication less per step and entry)
    return e * poly;
  } else if constexpr (CORE == CORE_MATERNNU) {
    T unused;
    return matern_real_of<T>(g, r2, &unused);
  } else if constexpr (CORE == CORE_SPECIAL) {
    return special_core<T>(g, r2);
  } else {
    return iso_core<T>(g, r2, d0, d1);
  }
}

__host__ inline int core_of(int kernel, int p) {
  if (kernel == LGP_KERNEL_EXPQUAD) return CORE_EXPQUAD;
  if (kernel == LGP_KERNEL_MATERNP && p >= 1 && p <= 3) return CORE_MATERN1 + (p - 1);
  if (kernel == LGP_KERNEL_MATERN) return CORE_MATERNNU;
  if (kernel == LGP_KERNEL_COLOR || kernel == LGP_KERNEL_PINK || kernel == LGP_KERNEL_BESSEL || kernel == LGP_KERNEL_PSERIES) return CORE_SPECIAL;
  return CORE_GENERIC;
}

// core value and its derivative with respect to `scale` (the inputs are divided by scale before differencing, so
// r2 = R2 / scale^2 and d r2 / d scale = -2 r2 / scale): analytic forms of the cores above.  With scale=None the
// derivative is taken at scale = 1.
template <typename T>
__device__ __forceinline__ T iso_core_dscale(const IsoArgs<T> &g, T r2, T &dk) {
  const T inv_s = g.has_scale ? T(1) / g.scale : T(1);
  if (g.kernel == LGP_KERNEL_EXPQUAD) {                  // k = exp(-r2/2): dk/dscale = k r2 / scale
    const T k = lgp_exp_neg<T>(T(0.5) * r2);
    dk = k * r2 * inv_s;
    return k;
  } else if (g.kernel == LGP_KERNEL_GAMMAEXP) {          // k = exp(-r2^(gamma/2)): dk/dscale = k gamma r2^(gamma/2) / scale
    const T pw = g.flag ? r2 : pow(r2, g.par[1]);
    const T k = exp(-pw);
    dk = k * g.par[0] * pw * inv_s;
    return k;
  } else if (g.kernel == LGP_KERNEL_CAUCHY) {            // k = (1 + pw/beta)^(-beta/alpha), pw = r2^(alpha/2): dk/dscale = k pw / ((1 + pw/beta) scale)
    const T pw = g.flag ? r2 : pow(r2, g.par[0]);
    const T base = T(1) + pw / g.par[1];
    const T k = pow(base, g.par[2]);
    dk = k * pw / base * inv_s;
    return k;
  } else {                                               // Maternp: k = exp(-x) poly(x), x = sqrt((2p+1) r2 + 1e-30); dx/dscale = -r2s / (x scale)
    const T r2s = g.twop1 * r2;
    T e;
    const T x = lgp_sqrt_exp<T>(r2s + T(1e-30), &e);
    T poly = T(1), dpoly = T(0);
    for (int k = 0; k < g.p; ++k) {
      const T c2 = g.coef[k] * T(2);
      dpoly = dpoly * c2 * x + poly * c2;                // d/dx of (1 + poly c 2 x)
      poly = T(1) + poly * c2 * x;
    }
    dk = e * (poly - dpoly) * r2s / x * inv_s;           // -(dk/dx) r2s / (x scale)
    return e * poly;
  }
}

// D > 0: compile-time field count (coordinates of the lane's columns live in registers); D == 0: runtime d
// SYM (both point sets the same, `symmetric` of the C-ABI, _elements.py:567-573 `halfmatrix`): only the tiles that reach the
// diagonal or lie left of it are in the grid -- numbered consecutively, so that the XCDs (workgroup i runs on XCD i mod 8) get
// equal shares -- and a tile entirely left of its rows' GT_COLS-block also stores its transpose: thread t keeps its W columns of
// 16 rows in registers and writes them as W runs of 16 consecutive entries of the rows col0 + c0 + e.  The entries are exactly
// symmetric by construction ((u - v)^2 == (v - u)^2), so the matrix is the one the full grid writes, at half the evaluations.
template <typename T, int D, bool GRAD = false, int CORE = CORE_GENERIC, bool SYM = false>
__global__ void __launch_bounds__(GT_THREADS) gram_iso_kernel(const IsoArgs<T> g) {
  constexpr int W = VecOf<T>::W;
  constexpr int GT_COLS = GT_THREADS * W;
  constexpr int RPB = GT_COLS / GT_ROWS;              // tile rows per GT_COLS-block of rows
  typedef typename VecOf<T>::type vec_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *su = reinterpret_cast<T *>(smem_raw);            // [d][GT_ROWS]
  const int d = D > 0 ? D : g.d;
  T *sv = su + d * GT_ROWS;                            // [d][GT_COLS]

  const int ntc = (g.m + GT_COLS - 1) / GT_COLS;
  int tr, tc;
  if (SYM) {
    // row group G (RPB tile rows) has G + 1 tiles per row: RPB G (G + 1) / 2 tiles before it
    const long b = blockIdx.x;
    long G = (long)((sqrt(1.0 + 8.0 * (double)b / RPB) - 1.0) * 0.5);
    while (RPB * (G + 1) * (G + 2) / 2 <= b) ++G;
    while (RPB * G * (G + 1) / 2 > b) --G;
    const long rem = b - RPB * G * (G + 1) / 2;
    tr = (int)(G * RPB + rem / (G + 1));
    tc = (int)(rem % (G + 1));
  } else {
    tr = blockIdx.x / ntc;
    tc = blockIdx.x % ntc;
  }
  const int row0 = tr * GT_ROWS, col0 = tc * GT_COLS;
  const int tid = threadIdx.x;
  const bool mirror = SYM && tc < tr / RPB;            // (workgroup-uniform) entirely left of the rows' block: nobody else writes its transpose

  for (int idx = tid; idx < d * GT_ROWS; idx += GT_THREADS) {
    const int f = idx / GT_ROWS, r = idx % GT_ROWS;
    T v = T(0);
    if (row0 + r < g.n) {
      v = g.x[(long)f * g.ldx + row0 + r];
      if (g.has_loc) v = v - g.loc;
      if (g.has_scale) v = v / g.scale;
    }
    su[idx] = v;
  }
  for (int idx = tid; idx < d * GT_COLS; idx += GT_THREADS) {
    const int f = idx / GT_COLS, c = idx % GT_COLS;
    T v = T(0);
    if (col0 + c < g.m) {
      v = g.y[(long)f * g.ldy + col0 + c];
      if (g.has_loc) v = v - g.loc;
      if (g.has_scale) v = v / g.scale;
    }
    sv[idx] = v;
  }
  __syncthreads();

  const int c0 = tid * W;
  T vreg[D > 0 ? D : 1][W];
  if (D > 0) {
#pragma unroll
    for (int f = 0; f < D; ++f)
#pragma unroll
      for (int e = 0; e < W; ++e) vreg[f][e] = sv[f * GT_COLS + c0 + e];
  }
  const bool colfull = (col0 + c0 + W <= g.m) && ((g.ldo % W) == 0) && ((reinterpret_cast<uintptr_t>(g.out) % 16) == 0);
  const int rmax = (g.n - row0) < GT_ROWS ? (g.n - row0) : GT_ROWS;
  const T slope = GRAD ? g.ops.slope() : T(0);
  constexpr int MR = 16;                              // rows per transposed run
  auto do_row = [&](int r, T (&res)[W]) {
    T dres[W];
#pragma unroll
    for (int e = 0; e < W; ++e) {
      T r2 = T(0), d0 = T(0), d1 = T(0);
      if (D > 0) {
#pragma unroll
        for (int f = 0; f < D; ++f) {
          const T xv = su[f * GT_ROWS + r], yv = vreg[f][e];
          const T df = xv - yv;
          if (CORE == CORE_GENERIC && g.xy == 3 && f == D - 1) {       // (Gibbs: the last row holds the per-point scales)
            d0 = xv; d1 = yv;
            continue;
          }
          const T sq = CORE == CORE_GENERIC ? iso_sq<T>(g, df, yv) : df * df;
          r2 = (f == 0) ? sq : r2 + sq;
          if (CORE == CORE_GENERIC) d0 = iso_aux<T>(g, f, xv, yv, df, d0);
        }
      } else {
        for (int f = 0; f < d; ++f) {
          const T xv = su[f * GT_ROWS + r], yv = sv[f * GT_COLS + c0 + e];
          const T df = xv - yv;
          if (CORE == CORE_GENERIC && g.xy == 3 && f == d - 1) {
            d0 = xv; d1 = yv;
            continue;
          }
          const T sq = CORE == CORE_GENERIC ? iso_sq<T>(g, df, yv) : df * df;
          r2 = (f == 0) ? sq : r2 + sq;
          if (CORE == CORE_GENERIC) d0 = iso_aux<T>(g, f, xv, yv, df, d0);
        }
      }
      if (GRAD) {
        T dk;
        if constexpr (CORE == CORE_MATERNNU) {             // dk/dx = -c (x/2)^nu K_{nu-1}(x), dx/dscale = -x / scale
          T xdk;
          const T kv = matern_real_of<T>(g, r2, &xdk);
          dk = g.has_scale ? xdk / g.scale : xdk;
          res[e] = g.ops.apply(kv);
        } else if constexpr (CORE == CORE_SPECIAL) {
          dk = special_core_mxdk<T>(g, r2);
          if (g.has_scale) dk = dk / g.scale;
          res[e] = g.ops.apply(special_core<T>(g, r2));
        } else {
          res[e] = g.ops.apply(iso_core_dscale<T>(g, r2, dk));
        }
        dres[e] = slope * dk;
      } else {
        res[e] = g.ops.apply(iso_core_t<T, CORE>(g, r2, d0, d1));
      }
    }
    T *po = g.out + (long)(row0 + r) * g.ldo + col0 + c0;
    T *pd = GRAD ? g.dsc + (long)(row0 + r) * g.ldo + col0 + c0 : nullptr;
    if (colfull && (!GRAD || (reinterpret_cast<uintptr_t>(g.dsc) % 16) == 0)) {
      vec_t v;
#pragma unroll
      for (int e = 0; e < W; ++e) v[e] = res[e];
      *reinterpret_cast<vec_t *>(po) = v;
      if (GRAD) {
#pragma unroll
        for (int e = 0; e < W; ++e) v[e] = dres[e];
        *reinterpret_cast<vec_t *>(pd) = v;
      }
    } else {
#pragma unroll
      for (int e = 0; e < W; ++e)
        if (col0 + c0 + e < g.m) {
          po[e] = res[e];
          if (GRAD) pd[e] = dres[e];
        }
    }
  };
  if (!SYM || !mirror) {
    for (int r = 0; r < rmax; ++r) {
      T res[W];
      do_row(r, res);
    }
  } else {
    // (measured instead: the transpose staged through LDS so that neighbouring lanes write one row -- slower, 2.6 against 2.2 ms at
    // N = 32768: what the mirrored half costs is HBM write locality, 32-entry runs 256 KB apart, not the coalescing of a store)
    // (round 5: taller tiles, so that a mirrored row receives 128 or 512 consecutive entries from one workgroup instead of 32: 2.50 and
    //  2.73 ms against 2.15 -- slower again; with the lean exp / sqrt the full grid takes 1.95-2.2 ms, so the symmetric form no longer gains)
    for (int rb = 0; rb < rmax; rb += MR) {
      T keep[MR][W];
      const int cnt = (rmax - rb) < MR ? (rmax - rb) : MR;
#pragma unroll
      for (int q = 0; q < MR; ++q)
        if (q < cnt) do_row(rb + q, keep[q]);
#pragma unroll
      for (int e = 0; e < W; ++e) {
        T *pt = g.out + (long)(col0 + c0 + e) * g.ldo + row0 + rb;      // row col0 + c0 + e < n: the tile lies left of the diagonal block
        if (cnt == MR && ((g.ldo % W) == 0) && ((reinterpret_cast<uintptr_t>(g.out) % 16) == 0)) {
#pragma unroll
          for (int q = 0; q < MR; q += W) {
            vec_t v;
#pragma unroll
            for (int w2 = 0; w2 < W; ++w2) v[w2] = keep[q + w2][e];
            *reinterpret_cast<vec_t *>(pt + q) = v;
          }
        } else {
#pragma unroll
          for (int q = 0; q < MR; ++q)
            if (q < cnt) pt[q] = keep[q][e];
        }
      }
    }
  }
}

// out[i] = ops(core(r2(x_i, y_i))): the elementwise (broadcast) form of CrossKernel.__call__
// (_Kernel/_crosskernel.py:192-200) for equally shaped x and y, also used by `halfmatrix`
template <typename T>
__global__ void __launch_bounds__(256) gram_iso_pairs_kernel(const IsoArgs<T> g) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= g.n) return;
  T r2 = T(0), d0 = T(0), d1 = T(0);
  for (int f = 0; f < g.d; ++f) {
    T u = g.x[(long)f * g.ldx + i], v = g.y[(long)f * g.ldy + i];
    if (g.has_loc) { u = u - g.loc; v = v - g.loc; }
    if (g.has_scale) { u = u / g.scale; v = v / g.scale; }
    if (g.xy == 3 && f == g.d - 1) {
      d0 = u; d1 = v;
      continue;
    }
    const T df = u - v;
    const T sq = iso_sq<T>(g, df, v);
    r2 = (f == 0) ? sq : r2 + sq;
    d0 = iso_aux<T>(g, f, u, v, df, d0);
  }
  g.out[i] = g.ops.apply(iso_core<T, true>(g, r2, d0, d1));
}

template <typename T>
int gram_iso_launch(lgp_ctx *ctx, int kernel, int p, int npar, const T *hpar, int d, const T *dx, size_t ldx, size_t n, const T *dy, size_t ldy, size_t m,
                    int has_loc, T loc, int has_scale, T scale, int nops, const int *hops_code, const T *hops_val, T *dout,
                    size_t ldo, bool pairs = false, T *ddsc = nullptr, bool symmetric = false, const T *dtable = nullptr, size_t ntable = 0, T fill = T(0)) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  if (n == 0 || m == 0) return LGP_OK;
  if (pairs) { LGP_REQUIRE(n == m, "pairs mode needs as many x as y points"); ldo = m; }
  LGP_REQUIRE(dx != nullptr && dy != nullptr && dout != nullptr, "null pointer");
  LGP_REQUIRE(kernel >= LGP_KERNEL_EXPQUAD && kernel <= LGP_KERNEL_PSERIES, "unknown isotropic kernel");
  LGP_REQUIRE((kernel != LGP_KERNEL_LAGTABLE && kernel != LGP_KERNEL_PSERIES) || ((dtable != nullptr || ntable == 0) && ntable < (1u << 30) && ddsc == nullptr), "LagTable takes a device table (lgp_gram_lagtable), no scale tangent");
  LGP_REQUIRE(kernel < LGP_KERNEL_EXPON || kernel == LGP_KERNEL_LINEAR || kernel == LGP_KERNEL_GIBBS || kernel == LGP_KERNEL_MATERN || kernel == LGP_KERNEL_BESSEL || d == 1, "this kernel is a function of one variable (maxdim = 1)");
  LGP_REQUIRE(kernel != LGP_KERNEL_GIBBS || (d >= 2 && !has_loc && !has_scale), "Gibbs takes the per-point scales as the last coordinate row, inputs already transformed");
  LGP_REQUIRE(d >= 1 && d <= 64, "field count must be in [1,64]");
  LGP_REQUIRE(p >= 0 && (p <= MAX_P || kernel == LGP_KERNEL_COLOR || kernel == LGP_KERNEL_PSERIES), "Maternp order must be in [0,16]");
  const bool derived = kernel == LGP_KERNEL_EXPQUAD_DIFF || kernel == LGP_KERNEL_MATERNP_DIFF;
  const bool posabs = kernel == LGP_KERNEL_WENDLAND || kernel == LGP_KERNEL_CAUSALEXPQUAD || kernel == LGP_KERNEL_LOG;
  LGP_REQUIRE(!(derived || posabs) || ddsc == nullptr, "no scale tangent for this kernel");
  LGP_REQUIRE(kernel != LGP_KERNEL_EXPQUAD_DIFF || (d <= 8 && ((npar == 0 && d == 1 && p <= 4) || (npar == d && hpar != nullptr))),
              "ExpQuadDiff: at most 8 fields, hpar = the total derivative order along each field (at most 4 each)");
  LGP_REQUIRE(kernel != LGP_KERNEL_MATERNP_DIFF || (d == 1 && p >= 1 && p < MAX_P && npar == p + 2 && hpar != nullptr),
              "MaternpDiff: one-dimensional points, hpar = {parity of the order, the p + 1 coefficients of Q}");
  LGP_REQUIRE(nops >= 0 && nops <= MAX_OPS, "at most 8 output operations");
  LGP_REQUIRE(n < (1u << 30) && m < (1u << 30), "too many points");
  LGP_REQUIRE(ldx >= n && ldy >= m && ldo >= m, "leading dimension too small");
  LGP_REQUIRE(!has_scale || (scale > T(0) && scale < T(INFINITY)), "scale must be positive and finite");   // _ops.py:324-325
  LGP_REQUIRE(!has_loc || (loc > T(-INFINITY) && loc < T(INFINITY)), "loc must be finite");                 // _ops.py:306-307
  DeviceGuard dg(ctx->device);
  IsoArgs<T> g;
  g.x = dx; g.y = dy; g.out = dout; g.dsc = ddsc;
  LGP_REQUIRE(!(pairs && ddsc), "the scale derivative is available for the Gram form only");
  g.ldx = (long)ldx; g.ldy = (long)ldy; g.ldo = (long)ldo;
  g.n = (int)n; g.m = (int)m; g.d = d;
  g.kernel = kernel; g.p = p;
  g.has_loc = has_loc; g.has_scale = has_scale; g.loc = loc; g.scale = scale;
  // _bessel.py:106-109: for k in reversed(range(p)): c = (p - k) / ((2p - k)(k + 1)); poly = 1 + poly*c*2*x
  for (int i = 0; i < MAX_P; ++i) g.coef[i] = T(0);
  int pos = 0;
  for (int k = (p <= MAX_P ? p : 0) - 1; k >= 0; --k) g.coef[pos++] = (T)((double)(p - k) / ((double)(2 * p - k) * (double)(k + 1)));
  g.twop1 = (T)(2 * p + 1);
  g.par[0] = g.par[1] = g.par[2] = T(0);
  g.table = dtable; g.ntable = (int)ntable; g.fill = fill;
  g.cbase = 0; g.cyc_nb = 0; g.cyc_stride = 0; g.wloc = 0;
  g.flag = 0;
  g.posabs = posabs ? 1 : 0;
  g.xy = kernel == LGP_KERNEL_LINEAR ? 2 : ((kernel >= LGP_KERNEL_WIENER && kernel <= LGP_KERNEL_WIENERINTEGRAL) || kernel == LGP_KERNEL_TAYLOR ? 1 : (kernel == LGP_KERNEL_GIBBS ? 3 : 0));
  for (int f = 0; f < 8; ++f) g.dord[f] = 0;
  g.lower_only = 0;
  if (kernel == LGP_KERNEL_EXPQUAD_DIFF) {
    if (npar == 0) g.dord[0] = p;
    for (int f = 0; f < npar; ++f) {
      LGP_REQUIRE(hpar[f] >= T(0) && hpar[f] <= T(4) && hpar[f] == T((int)hpar[f]), "derivative orders must be integers in [0, 4]");
      g.dord[f] = (int)hpar[f];
    }
  } else if (kernel == LGP_KERNEL_MATERNP_DIFF) {
    g.flag = hpar[0] != T(0);
    for (int k = 0; k <= p; ++k) g.coef[k] = hpar[1 + k];
  } else if (kernel == LGP_KERNEL_WENDLAND) {     // hpar = {nu + k, the k + 1 coefficients of the polynomial in r, highest power first}
    LGP_REQUIRE(p <= 3 && npar == p + 2 && hpar != nullptr && hpar[0] >= T(1), "Wendland takes k in [0, 3], {nu + k, k + 1 coefficients}");
    g.par[0] = hpar[0];
    for (int k = 0; k <= p; ++k) g.coef[k] = hpar[1 + k];
  } else if (kernel == LGP_KERNEL_CAUSALEXPQUAD) {   // hpar = {alpha / 4}
    LGP_REQUIRE(npar == 1 && hpar != nullptr && hpar[0] >= T(0), "CausalExpQuad takes {alpha / 4}, alpha >= 0");
    g.par[0] = hpar[0];
  } else if (kernel == LGP_KERNEL_LOG) {
    LGP_REQUIRE(npar == 0, "this kernel takes no real parameters");
  } else if (kernel == LGP_KERNEL_CELERITE) {            // hpar = {gamma, B}
    LGP_REQUIRE(npar == 2 && hpar != nullptr && hpar[0] >= T(0) && hpar[0] < std::numeric_limits<T>::infinity() && fabs(hpar[1]) <= hpar[0], "Celerite takes {gamma, B}, |B| <= gamma");
    g.par[0] = hpar[0]; g.par[1] = hpar[1];
  } else if (kernel == LGP_KERNEL_HARMONIC) {            // hpar = {Q, eta Q, sqrt(1 - Q^2) - 1}; p = 0: Q < 1/2, 1: Q < 1, 2: Q = 1, 3: Q > 1
    LGP_REQUIRE(npar == 3 && hpar != nullptr && hpar[0] > T(0) && p >= 0 && p <= 3, "Harmonic takes {Q, eta Q, sqrt(1 - Q^2) - 1} and the branch");
    for (int k = 0; k < 3; ++k) g.par[k] = hpar[k];
  } else if (kernel == LGP_KERNEL_CIRCULAR) {            // hpar = {tau, c}
    LGP_REQUIRE(npar == 2 && hpar != nullptr && hpar[0] >= T(4) && hpar[1] > T(0) && hpar[1] <= T(0.5), "Circular takes {tau >= 4, c in (0, 1/2]}");
    g.par[0] = hpar[0]; g.par[1] = hpar[1];
  } else if (kernel == LGP_KERNEL_DECAYING) {            // hpar = {alpha}
    LGP_REQUIRE(npar == 1 && hpar != nullptr && hpar[0] >= T(0), "Decaying takes {alpha}, alpha >= 0");
    g.par[0] = hpar[0];
  } else if (kernel == LGP_KERNEL_FRACBROWNIAN) {        // hpar = {2 H, K, 1 / 2^K, 2 H K}
    LGP_REQUIRE(npar == 4 && hpar != nullptr && hpar[0] > T(0) && hpar[0] <= T(2) && hpar[1] > T(0) && hpar[1] <= T(1), "FracBrownian takes {2H, K, 2^-K, 2HK}, H and K in (0, 1]");
    for (int k = 0; k < 4; ++k) g.par[k] = hpar[k];
  } else if (kernel == LGP_KERNEL_STATIONARYFRACBROWNIAN) {   // hpar = {2 H}
    LGP_REQUIRE(npar == 1 && hpar != nullptr && hpar[0] > T(0) && hpar[0] <= T(2), "StationaryFracBrownian takes {2H}, H in (0, 1]");
    g.par[0] = hpar[0];
  } else if (kernel == LGP_KERNEL_PERIODIC) {            // hpar = {outerscale}
    LGP_REQUIRE(npar == 1 && hpar != nullptr && hpar[0] > T(0) && hpar[0] < std::numeric_limits<T>::infinity(), "Periodic takes {outerscale} in (0, inf)");
    g.par[0] = hpar[0];
  } else if (kernel == LGP_KERNEL_PSERIES) {             // p = mode | 16 uselog, hpar = {A, pw, eps, D}
    LGP_REQUIRE(npar == 4 && hpar != nullptr && (p & 15) <= 2 && ntable <= 4096, "PSeries takes the mode in p and {A, pw, eps, D}");
    for (int k = 0; k < 4; ++k) g.par[k] = hpar[k];
    g.flag = (p >> 4) & 1;
    g.p = p & 15;
  } else if (kernel == LGP_KERNEL_BESSEL) {              // hpar = {nu}
    LGP_REQUIRE(npar == 1 && hpar != nullptr && hpar[0] >= T(0) && hpar[0] <= T(100), "Bessel takes {nu} in [0, 100]");   // _matern.py:117
    const double nu = (double)hpar[0];
    const int fl = (int)nu;
    LGP_REQUIRE(d <= 2 * (fl + 1), "Bessel(nu) is a kernel in at most 2 (floor(nu) + 1) dimensions");                                // _matern.py:97-99
    g.p = fl;
    g.par[0] = (T)nu; g.par[1] = (T)tgamma(nu + 1.); g.par[2] = (T)tgamma(1. + nu - (double)fl);
    g.twop1 = (T)((2. + nu / 2.) * (2. + nu / 2.));                                                                                   // _matern.py:118
  } else if (kernel == LGP_KERNEL_COLOR) {               // p = n
    LGP_REQUIRE(npar == 0 && p >= 2, "Color takes the integer n >= 2 as p, no real parameters");       // _spectral.py:110
    double psi = -0.5772156649015329;
    for (int k = 1; k < p - 1; ++k) psi += 1. / (double)k;
    g.par[1] = (T)psi;                                   // psi(n - 1): the scale tangent
    g.par[0] = (T)(psi + 1. / (double)(p - 1));          // psi(n)
  } else if (kernel == LGP_KERNEL_PINK) {                // hpar = {dw}
    LGP_REQUIRE(npar == 1 && hpar != nullptr && hpar[0] >= T(0) && hpar[0] < std::numeric_limits<T>::infinity(), "Pink takes {dw}, dw >= 0");
    g.par[0] = hpar[0]; g.par[1] = (T)log1p((double)hpar[0]); g.par[2] = (T)sqrt((double)std::numeric_limits<T>::epsilon());   // _spectral.py:62-66
  } else if (kernel == LGP_KERNEL_MATERN) {              // hpar = {nu}
    LGP_REQUIRE(npar == 1 && hpar != nullptr && hpar[0] >= T(0) && hpar[0] <= T(100), "Matern takes {nu} in [0, 100]");   // _matern.py:73
    const double nu = (double)hpar[0];
    g.flag = nu == 0.;
    g.twop1 = (T)(2. * (nu == 0. ? 1. : nu));            // _matern.py:74
    const lgp_bessel::MaternNu c = lgp_bessel::matern_nu_constants(nu);
    g.p = c.nl;
    g.coef[0] = (T)c.nu; g.coef[1] = (T)c.mu; g.coef[2] = (T)c.gampl; g.coef[3] = (T)c.gammi;
    g.coef[4] = (T)c.gam1; g.coef[5] = (T)c.gam2; g.coef[6] = (T)c.fact; g.coef[7] = (T)c.c2;
  } else if (kernel == LGP_KERNEL_GAMMAEXP) {            // hpar = {gamma, gamma / 2}
    LGP_REQUIRE(npar == 2 && hpar != nullptr, "GammaExp takes {gamma, gamma/2}");
    LGP_REQUIRE(hpar[0] > T(0) && hpar[0] <= T(2), "gamma must be in (0, 2]");                 // _basic.py:109-110
    g.par[0] = hpar[0]; g.par[1] = hpar[1];
    g.flag = hpar[0] == T(2);
  } else if (kernel == LGP_KERNEL_CAUCHY) {       // hpar = {alpha, beta, -beta / alpha}
    LGP_REQUIRE(npar == 3 && hpar != nullptr, "Cauchy takes {alpha, beta, -beta/alpha}");
    LGP_REQUIRE(hpar[0] > T(0) && hpar[0] <= T(2) && hpar[1] > T(0), "alpha must be in (0, 2], beta positive");   // _basic.py:336-338
    g.par[0] = hpar[0] / T(2); g.par[1] = hpar[1]; g.par[2] = hpar[2];
    g.flag = hpar[0] == T(2);
  } else {
    LGP_REQUIRE(npar == 0, "this kernel takes no real parameters");
  }
  g.ops.n = nops;
  for (int i = 0; i < MAX_OPS; ++i) { g.ops.code[i] = 0; g.ops.val[i] = T(0); }
  for (int i = 0; i < nops; ++i) {
    LGP_REQUIRE(hops_code[i] == LGP_OP_MUL || hops_code[i] == LGP_OP_ADD, "unknown output operation");
    g.ops.code[i] = hops_code[i];
    g.ops.val[i] = hops_val[i];
  }
  if (pairs) {
    hipLaunchKernelGGL((gram_iso_pairs_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, g);
    LGP_LAUNCH_CHECK();
    return LGP_OK;
  }
  constexpr int W = VecOf<T>::W;
  constexpr int GT_COLS = GT_THREADS * W;
  const size_t ntr = (n + GT_ROWS - 1) / GT_ROWS, ntc = (m + GT_COLS - 1) / GT_COLS;
  LGP_REQUIRE(ntr * ntc < (1ull << 31), "gram grid too large");
  size_t shmem = (size_t)d * (GT_ROWS + GT_COLS) * sizeof(T);
  dim3 gd((unsigned)(ntr * ntc)), bd(GT_THREADS);
  if (ddsc && core_of(kernel, p) == CORE_SPECIAL) {
    switch (d) {
      case 1: hipLaunchKernelGGL((gram_iso_kernel<T, 1, true, CORE_SPECIAL>), gd, bd, shmem, ctx->stream, g); break;
      case 2: hipLaunchKernelGGL((gram_iso_kernel<T, 2, true, CORE_SPECIAL>), gd, bd, shmem, ctx->stream, g); break;
      case 3: hipLaunchKernelGGL((gram_iso_kernel<T, 3, true, CORE_SPECIAL>), gd, bd, shmem, ctx->stream, g); break;
      default: hipLaunchKernelGGL((gram_iso_kernel<T, 0, true, CORE_SPECIAL>), gd, bd, shmem, ctx->stream, g); break;
    }
  } else if (ddsc && kernel == LGP_KERNEL_MATERN) {
    switch (d) {
      case 1: hipLaunchKernelGGL((gram_iso_kernel<T, 1, true, CORE_MATERNNU>), gd, bd, shmem, ctx->stream, g); break;
      case 2: hipLaunchKernelGGL((gram_iso_kernel<T, 2, true, CORE_MATERNNU>), gd, bd, shmem, ctx->stream, g); break;
      case 3: hipLaunchKernelGGL((gram_iso_kernel<T, 3, true, CORE_MATERNNU>), gd, bd, shmem, ctx->stream, g); break;
      default: hipLaunchKernelGGL((gram_iso_kernel<T, 0, true, CORE_MATERNNU>), gd, bd, shmem, ctx->stream, g); break;
    }
  } else if (ddsc) {
    switch (d) {
      case 1: hipLaunchKernelGGL((gram_iso_kernel<T, 1, true>), gd, bd, shmem, ctx->stream, g); break;
      case 2: hipLaunchKernelGGL((gram_iso_kernel<T, 2, true>), gd, bd, shmem, ctx->stream, g); break;
      case 3: hipLaunchKernelGGL((gram_iso_kernel<T, 3, true>), gd, bd, shmem, ctx->stream, g); break;
      default: hipLaunchKernelGGL((gram_iso_kernel<T, 0, true>), gd, bd, shmem, ctx->stream, g); break;
    }
  } else {
    // one triangle when both point sets are the same array (and the core is one of the plain ones: the derived kernels are not
    // symmetric entry by entry)
    // ... and where it pays: the mirrored half is written in runs of 16 entries 8 N bytes apart, which HBM takes at ~3.9 TB/s
    // against 5.3-5.6 TB/s for whole tile rows.  ExpQuad, write-bound on the full grid (1.5-1.6 ms at N = 32768), would lose
    // (2.2 ms); the cores with more arithmetic per entry gain (Matern-5/2 3-D: 2.4 -> 2.2 ms); fp32 runs would be 64 bytes
    // (round 5, lean exp / sqrt: Matern-5/2 3-D of a point set with itself 2.20-2.21 ms on the full grid, 2.02-2.07 as one triangle, same box)
    static const bool nosym = getenv("LGP_GRAM_FULL") != nullptr, allsym = getenv("LGP_GRAM_HALF") != nullptr;
    const bool sym = symmetric && !nosym && dx == dy && n == m && ldx == ldy && n >= (size_t)GT_COLS &&
                     kernel != LGP_KERNEL_EXPQUAD_DIFF && kernel != LGP_KERNEL_MATERNP_DIFF &&
                     (allsym || (sizeof(T) == 8 && core_of(kernel, p) != CORE_EXPQUAD));
    if (sym) {
      constexpr size_t RPB = GT_COLS / GT_ROWS;
      size_t tiles = 0;
      for (size_t tr = 0; tr < ntr; ++tr) tiles += tr / RPB + 1;
      gd = dim3((unsigned)tiles);
    }
#define LGP_ISO_LAUNCH(DD, CC)                                                                                   \
    do {                                                                                                         \
      if (sym) hipLaunchKernelGGL((gram_iso_kernel<T, DD, false, CC, true>), gd, bd, shmem, ctx->stream, g);     \
      else hipLaunchKernelGGL((gram_iso_kernel<T, DD, false, CC>), gd, bd, shmem, ctx->stream, g);               \
    } while (0)
#define LGP_ISO_BY_CORE(DD)                                   \
    switch (core_of(kernel, p)) {                             \
      case CORE_EXPQUAD: LGP_ISO_LAUNCH(DD, CORE_EXPQUAD); break;   \
      case CORE_MATERN1: LGP_ISO_LAUNCH(DD, CORE_MATERN1); break;   \
      case CORE_MATERN2: LGP_ISO_LAUNCH(DD, CORE_MATERN2); break;   \
      case CORE_MATERN3: LGP_ISO_LAUNCH(DD, CORE_MATERN3); break;   \
      case CORE_MATERNNU: LGP_ISO_LAUNCH(DD, CORE_MATERNNU); break; \
      case CORE_SPECIAL: LGP_ISO_LAUNCH(DD, CORE_SPECIAL); break;   \
      default: LGP_ISO_LAUNCH(DD, CORE_GENERIC); break;             \
    }
    switch (d) {
      case 1: LGP_ISO_BY_CORE(1); break;
      case 2: LGP_ISO_BY_CORE(2); break;
      case 3: LGP_ISO_BY_CORE(3); break;
      default: LGP_ISO_BY_CORE(0); break;
    }
#undef LGP_ISO_BY_CORE
#undef LGP_ISO_LAUNCH
  }
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

// ------------------------------------------------------------------------------------------------
// Gram build fused with the regularisation passes of Chol.__init__ (_decomp.py:384-387): one pass over the pairs writes
// tril(K / s / s') straight into the factor's storage (zeros above the diagonal) and leaves the Gershgorin sums, instead of
// K -> HBM -> row sums (read n^2) -> scaled lower copy (read n^2 / 2, write n^2).  K = kernel(x, x) + diag(noise) is never
// materialised.  s_i = 2^rint(log2(K_ii) / 2) needs only the diagonal, K_ii = ops(core(0)) + noise_i, so every workgroup
// derives the scales of its own rows and columns; being powers of two, multiplying by 1 / s is the division.  Each thread
// owns W columns of a 128 x (256 W) tile and accumulates |K~_ij| down its columns: the column sums of the FULL symmetric
// matrix are the Gershgorin row sums, and the partial sums of the n / 128 tile rows are added in a fixed order afterwards
// (deterministic, no atomics).  The upper triangle is evaluated only for those sums.
constexpr int GR_ROWS = 128;
constexpr int GR_BATCH = GT_THREADS / 32;      // rows whose sums along the row are formed together (gram_reg_kernel)

template <typename T>
__device__ __forceinline__ T pow2_scale(T d) {      // _decomp.py:356-361
  return d != T(0) ? exp2(rint(T(0.5) * log2(d))) : T(1);
}

// Tiles entirely above the diagonal are not evaluated (round 3: they were, for the sums only -- 1.3 of the 3.3 ms at C3): their
// part of column j's sum, sum_{i in the tile's rows} |K~_ij| = sum |K~_ji|, is the sum ALONG ROW j of the mirrored entries, and
// the mirror of an entry of such a tile always lies in a tile entirely BELOW the diagonal (the column blocks of 256 W entries are
// whole multiples of the 128-row blocks).  So: tiles above: nothing; tiles below: column sums as before plus the sums along
// their rows (per wave by lane shuffles, the four waves' partials added in fixed order) -> rpart[tile column][row]; tiles that
// straddle the diagonal: evaluated in full as before, column sums only.  Gershgorin_j = sum_tr part[tr][j] + sum_tc rpart[tc][j].
// COLS (the sharded factorisation, csrc/dist.hip / _dist.py): the block of the columns [g.cbase, g.m) only, every row -- a rank's own
// block column evaluated in full (no other rank holds the mirrored entries), column sums only; svec for ALL rows
template <typename T, int D, int CORE, bool COLS = false>
__global__ void __launch_bounds__(GT_THREADS) gram_reg_kernel(const IsoArgs<T> g, T noise, const T *__restrict__ noisev, T *__restrict__ svec,
                                                              T *__restrict__ part, T *__restrict__ rpart) {
  constexpr int W = VecOf<T>::W;
  constexpr int GT_COLS = GT_THREADS * W;
  typedef typename VecOf<T>::type vec_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *su = reinterpret_cast<T *>(smem_raw);            // [d][GR_ROWS] transformed row coordinates
  const int d = D > 0 ? D : g.d;
  T *sv = su + d * GR_ROWS;                            // [d][GT_COLS]
  T *sinv = sv + d * GT_COLS;                          // [GR_ROWS] 1 / s_i of the rows;  then [GR_ROWS] their noise
  T *snz = sinv + GR_ROWS;
  T *srow = snz + GR_ROWS;                             // [GR_ROWS] sums along the rows (tiles below the diagonal)
  T *sbat = srow + GR_ROWS;                            // [GR_BATCH][32] partial sums of a batch of rows (within the 4 GR_ROWS entries the sums own)
  const int n = g.n;
  // Only the tiles that reach the diagonal or lie below it are in the grid (numbered consecutively: row group G of RPB tile rows
  // has G + 1 tiles per row).  The entries above the diagonal of the factor's storage are zero: a tile entirely below the
  // diagonal also writes the zeros of its MIRROR IMAGE (GT_COLS rows x GR_ROWS columns; together the mirrors are exactly the
  // tiles above the diagonal), one round of GT_THREADS 16-byte stores per row of its own loop, so that the stores of the zeros
  // are issued between the evaluations instead of by workgroups of their own (which made the kernel's time the SUM of the two
  // kinds of work: zeros alone 0.8 ms, evaluated tiles alone 1.9 ms, together 3.1 ms at N = 32768 on the same box).
  constexpr int RPB = GT_COLS / GR_ROWS;
  int tr, tc;
  if (COLS) {
    const int ntcl = (g.wloc + GT_COLS - 1) / GT_COLS;
    tr = blockIdx.x / ntcl;
    tc = blockIdx.x % ntcl;
  } else {
    const long b = blockIdx.x;
    long G = (long)((sqrt(1.0 + 8.0 * (double)b / RPB) - 1.0) * 0.5);
    while (RPB * (G + 1) * (G + 2) / 2 <= b) ++G;
    while (RPB * G * (G + 1) / 2 > b) --G;
    const long rem = b - RPB * G * (G + 1) / 2;
    tr = (int)(G * RPB + rem / (G + 1));
    tc = (int)(rem % (G + 1));
  }
  // COLS: local column lcol0 of the launch's storage -> global column col0 (block-cyclic: a tile lies within one block, GT_COLS
  // divides the block width); cend: one past the last column this tile may touch
  const int lcol0 = tc * GT_COLS;
  const int cq = (COLS && g.cyc_nb) ? lcol0 / g.cyc_nb : 0;
  const int row0 = tr * GR_ROWS, col0 = COLS ? g.cbase + (g.cyc_nb ? cq * g.cyc_stride + (lcol0 - cq * g.cyc_nb) : lcol0) : lcol0;
  const int cblk_end = (COLS && g.cyc_nb) ? g.cbase + cq * g.cyc_stride + g.cyc_nb : g.m;
  const int cend = COLS ? (cblk_end < n ? cblk_end : n) : n;
  if (COLS && col0 >= cend) return;
  if (row0 >= n) return;                                 // (the last row group may be ragged)
  const int tid = threadIdx.x;
  const T k0 = g.ops.apply(iso_core_t<T, CORE>(g, T(0)));     // the kernel on its diagonal: r2 = 0 exactly, as for the pairs (i, i)
  const bool below = !COLS && col0 + GT_COLS - 1 < row0;   // every entry strictly below the diagonal: sums along the rows as well
  const bool zeros = below && !g.lower_only;               // ... and the zeros of the mirror image
  constexpr int TPR = GR_ROWS / W, RPR = GT_THREADS / TPR;     // zero rounds: threads per mirrored row, rows per round
  const bool zfull = ((g.ldo % W) == 0) && ((reinterpret_cast<uintptr_t>(g.out) % 16) == 0);
  auto zero_round = [&](int zr) {
    const long zi = (long)col0 + (long)zr * RPR + tid / TPR;   // < row0 <= n - 1
    const int zj = row0 + (tid % TPR) * W;
    T *pz = g.out + zi * g.ldo + zj;
    if (zfull && zj + W <= n) {
      vec_t z;
#pragma unroll
      for (int e = 0; e < W; ++e) z[e] = T(0);
      *reinterpret_cast<vec_t *>(pz) = z;
    } else {
#pragma unroll
      for (int e = 0; e < W; ++e)
        if (zj + e < n) pz[e] = T(0);
    }
  };
  for (int idx = tid; idx < d * GR_ROWS; idx += GT_THREADS) {
    const int f = idx / GR_ROWS, r = idx % GR_ROWS;
    T v = T(0);
    if (row0 + r < n) {
      v = g.x[(long)f * g.ldx + row0 + r];
      if (g.has_loc) v = v - g.loc;
      if (g.has_scale) v = v / g.scale;
    }
    su[idx] = v;
  }
  for (int idx = tid; idx < d * GT_COLS; idx += GT_THREADS) {
    const int f = idx / GT_COLS, c = idx % GT_COLS;
    T v = T(0);
    if (col0 + c < cend) {
      v = g.x[(long)f * g.ldx + col0 + c];
      if (g.has_loc) v = v - g.loc;
      if (g.has_scale) v = v / g.scale;
    }
    sv[idx] = v;
  }
  for (int r = tid; r < GR_ROWS; r += GT_THREADS) {
    const T nz = (row0 + r < n) ? (noisev ? noisev[row0 + r] : noise) : T(0);
    snz[r] = nz;
    const T si = pow2_scale<T>(k0 + nz);
    sinv[r] = T(1) / si;
    if (COLS && tc == 0 && row0 + r < n) svec[row0 + r] = si;
  }
  __syncthreads();

  const int c0 = tid * W;
  T vreg[D > 0 ? D : 1][W], cinv[W], cnz[W], acc[W];
#pragma unroll
  for (int e = 0; e < W; ++e) {
    const int j = col0 + c0 + e;
    cnz[e] = (j < cend) ? (noisev ? noisev[j] : noise) : T(0);
    const T sj = pow2_scale<T>(k0 + cnz[e]);
    cinv[e] = T(1) / sj;
    acc[e] = T(0);
    if (!COLS && tr == RPB * tc && j < n) svec[j] = sj;        // the first tile row that holds this column block
    if (D > 0) {
#pragma unroll
      for (int f = 0; f < D; ++f) vreg[f][e] = sv[f * GT_COLS + c0 + e];
    }
  }
  const bool colfull = (col0 + c0 + W <= cend) && ((g.ldo % W) == 0) && ((reinterpret_cast<uintptr_t>(g.out) % 16) == 0) && (!COLS || (g.cbase % W) == 0);
  const int rmax = (n - row0) < GR_ROWS ? (n - row0) : GR_ROWS;
  // INTERIOR (workgroup-uniform): a whole tile strictly below the diagonal with every column in range and aligned stores -- no entry
  // is on the diagonal, above it or outside the matrix, so the selects of the general body go
  auto rows = [&](auto interior_c) {
  constexpr bool INTERIOR = decltype(interior_c)::value;
  for (int r = 0; r < rmax; ++r) {
    const int i = row0 + r;
    const T rinv = sinv[r];
    T res[W];
#pragma unroll
    for (int e = 0; e < W; ++e) {
      const int j = col0 + c0 + e;
      T r2 = T(0);
      if (D > 0) {
#pragma unroll
        for (int f = 0; f < D; ++f) {
          const T df = su[f * GR_ROWS + r] - vreg[f][e];
          const T sq = CORE == CORE_GENERIC ? iso_sq<T>(g, df, vreg[f][e]) : df * df;
          r2 = (f == 0) ? sq : r2 + sq;
        }
      } else {
        for (int f = 0; f < d; ++f) {
          const T df = su[f * GR_ROWS + r] - sv[f * GT_COLS + c0 + e];
          const T sq = CORE == CORE_GENERIC ? iso_sq<T>(g, df, sv[f * GT_COLS + c0 + e]) : df * df;
          r2 = (f == 0) ? sq : r2 + sq;
        }
      }
      T v = g.ops.apply(iso_core_t<T, CORE>(g, r2));
      if (!INTERIOR && i == j) v = v + cnz[e];          // Kxx + ycov (_compute.py:85), noise on the diagonal
      v = v * cinv[e] * rinv;                           // K / s / s[:, None] (:385): exact, s is a power of two
      if (INTERIOR || j < cend) acc[e] += fabs(v);
      res[e] = (INTERIOR || j <= i) ? v : T(0);
    }
    if (zeros) zero_round(r);
    if (below) {
      // sum along the row over this tile's columns, in a fixed order (round 5): three exchanges among eight neighbouring lanes, the 32
      // partial sums of a row parked in LDS, and once per eight rows 32 threads per row finish them with five exchanges -- ~11
      // instructions per row and thread instead of the ~30 of a 64-lane exchange tree (12 ds_bpermute) per row
      T rs = T(0);
#pragma unroll
      for (int e = 0; e < W; ++e) rs += (INTERIOR || col0 + c0 + e < n) ? fabs(res[e]) : T(0);
#pragma unroll
      for (int off = 1; off <= 4; off <<= 1) rs += __shfl_xor(rs, off, 64);
      if ((tid & 7) == 0) sbat[(r & (GR_BATCH - 1)) * 32 + (tid >> 3)] = rs;
      if ((r & (GR_BATCH - 1)) == GR_BATCH - 1 || r == rmax - 1) {
        __syncthreads();
        const int q = tid >> 5, seg = tid & 31, rq = (r & ~(GR_BATCH - 1)) + q;      // GT_THREADS / 32 = GR_BATCH rows
        T v = sbat[q * 32 + seg];
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (seg == 0 && rq <= r) srow[rq] = v;
        __syncthreads();
      }
    }
    T *po = g.out + (long)i * g.ldo + lcol0 + c0;
    if (INTERIOR || colfull) {
      vec_t v;
#pragma unroll
      for (int e = 0; e < W; ++e) v[e] = res[e];
      *reinterpret_cast<vec_t *>(po) = v;
    } else {
#pragma unroll
      for (int e = 0; e < W; ++e)
        if (col0 + c0 + e < cend) po[e] = res[e];
    }
  }
  };
  const bool interior = below && row0 + GR_ROWS <= n && col0 + GT_COLS <= cend && ((g.ldo % W) == 0) && ((reinterpret_cast<uintptr_t>(g.out) % 16) == 0);
  if (interior) rows(std::true_type{});
  else rows(std::false_type{});
  if (zeros)
    for (int zr = rmax; zr < GR_ROWS; ++zr) zero_round(zr);      // (ragged last tile row: the rounds its loop did not reach)
#pragma unroll
  for (int e = 0; e < W; ++e)
    if (col0 + c0 + e < cend) part[(long)tr * (COLS ? g.wloc : n) + lcol0 + c0 + e] = acc[e];
  if (below) {
    __syncthreads();
    for (int r = tid; r < rmax; r += GT_THREADS)
      rpart[(long)tc * n + row0 + r] = srow[r];
  }
}

template <typename T>
int gram_reg_launch(lgp_ctx *ctx, int kernel, int p, int npar, const T *hpar, int d, const T *dx, size_t ldx, size_t n, int has_loc, T loc,
                    int has_scale, T scale, int nops, const int *hops_code, const T *hops_val, T noise, const T *dnoise, T *dL, size_t ldl, T *dsvec,
                    T *dpart, T *drpart, size_t *ntr_out, bool cols = false, size_t c0 = 0, size_t w = 0, size_t cyc_nb = 0, size_t cyc_stride = 0,
                    bool lower_only = false) {
  LGP_REQUIRE(ctx != nullptr && dx != nullptr && dL != nullptr && dsvec != nullptr && dpart != nullptr && (cols || drpart != nullptr), "null pointer");
  LGP_REQUIRE(!cols || (w >= 1 && c0 < n && (cyc_nb || c0 + w <= n)), "column range outside the matrix");
  LGP_REQUIRE(!cols || cyc_nb == 0 || (cyc_nb % (GT_THREADS * VecOf<T>::W) == 0 && cyc_stride >= cyc_nb), "block width: a multiple of the tile width");
  LGP_REQUIRE(kernel == LGP_KERNEL_EXPQUAD || kernel == LGP_KERNEL_MATERNP || kernel == LGP_KERNEL_GAMMAEXP || kernel == LGP_KERNEL_CAUCHY,
              "unknown isotropic kernel");
  LGP_REQUIRE(d >= 1 && d <= 64 && p >= 0 && p <= MAX_P && nops >= 0 && nops <= MAX_OPS, "bad kernel parameters");
  LGP_REQUIRE(n >= 1 && n < (1u << 30) && ldx >= n && ldl >= (cols ? w : n), "bad dimensions");
  LGP_REQUIRE(!has_scale || (scale > T(0) && scale < T(INFINITY)), "scale must be positive and finite");
  LGP_REQUIRE(!has_loc || (loc > T(-INFINITY) && loc < T(INFINITY)), "loc must be finite");
  DeviceGuard dg(ctx->device);
  IsoArgs<T> g;
  g.x = dx; g.y = dx; g.out = dL; g.dsc = nullptr;
  g.ldx = (long)ldx; g.ldy = (long)ldx; g.ldo = (long)ldl;
  g.n = (int)n; g.m = (int)n; g.d = d;
  g.kernel = kernel; g.p = p;
  g.has_loc = has_loc; g.has_scale = has_scale; g.loc = loc; g.scale = scale;
  for (int i = 0; i < MAX_P; ++i) g.coef[i] = T(0);
  int pos = 0;
  for (int k = p - 1; k >= 0; --k) g.coef[pos++] = (T)((double)(p - k) / ((double)(2 * p - k) * (double)(k + 1)));
  g.twop1 = (T)(2 * p + 1);
  g.posabs = 0;
  g.xy = 0;
  g.table = nullptr; g.ntable = 0; g.fill = T(0);
  g.cbase = 0; g.cyc_nb = 0; g.cyc_stride = 0; g.wloc = 0;
  for (int f = 0; f < 8; ++f) g.dord[f] = 0;
  g.lower_only = lower_only ? 1 : 0;
  g.par[0] = g.par[1] = g.par[2] = T(0);
  g.flag = 0;
  if (kernel == LGP_KERNEL_GAMMAEXP) {
    LGP_REQUIRE(npar == 2 && hpar != nullptr && hpar[0] > T(0) && hpar[0] <= T(2), "GammaExp takes {gamma, gamma/2}, gamma in (0, 2]");
    g.par[0] = hpar[0]; g.par[1] = hpar[1]; g.flag = hpar[0] == T(2);
  } else if (kernel == LGP_KERNEL_CAUCHY) {
    LGP_REQUIRE(npar == 3 && hpar != nullptr && hpar[0] > T(0) && hpar[0] <= T(2) && hpar[1] > T(0), "Cauchy takes {alpha, beta, -beta/alpha}");
    g.par[0] = hpar[0] / T(2); g.par[1] = hpar[1]; g.par[2] = hpar[2]; g.flag = hpar[0] == T(2);
  } else {
    LGP_REQUIRE(npar == 0, "this kernel takes no real parameters");
  }
  g.ops.n = nops;
  for (int i = 0; i < MAX_OPS; ++i) { g.ops.code[i] = 0; g.ops.val[i] = T(0); }
  for (int i = 0; i < nops; ++i) {
    LGP_REQUIRE(hops_code[i] == LGP_OP_MUL || hops_code[i] == LGP_OP_ADD, "unknown output operation");
    g.ops.code[i] = hops_code[i];
    g.ops.val[i] = hops_val[i];
  }
  constexpr int W = VecOf<T>::W;
  constexpr int GT_COLS = GT_THREADS * W;
  const size_t ntr = (n + GR_ROWS - 1) / GR_ROWS, ntc = (n + GT_COLS - 1) / GT_COLS;
  LGP_REQUIRE(ntr * ntc < (1ull << 31), "gram grid too large");
  const size_t shmem = ((size_t)d * (GR_ROWS + GT_COLS) + 6 * GR_ROWS) * sizeof(T);
  constexpr size_t RPB = (size_t)GT_COLS / GR_ROWS;
  if (cols) {
    // a column range, every row (a rank's block column of the sharded factorisation): plain grid, every tile writes its partial sums
    g.cbase = (int)c0;
    g.m = (int)((cyc_nb ? n : c0 + w));
    g.wloc = (int)w; g.cyc_nb = (int)cyc_nb; g.cyc_stride = (int)cyc_stride;
    const size_t ntcl = (w + GT_COLS - 1) / GT_COLS;
    LGP_REQUIRE(ntr * ntcl < (1ull << 31), "gram grid too large");
    dim3 gdc((unsigned)(ntr * ntcl)), bdc(GT_THREADS);
#define LGP_REGC_LAUNCH(DD, CC) hipLaunchKernelGGL((gram_reg_kernel<T, DD, CC, true>), gdc, bdc, shmem, ctx->stream, g, noise, dnoise, dsvec, dpart, drpart)
#define LGP_REGC_BY_CORE(DD)                                   \
    switch (core_of(kernel, p)) {                               \
      case CORE_EXPQUAD: LGP_REGC_LAUNCH(DD, CORE_EXPQUAD); break;   \
      case CORE_MATERN2: LGP_REGC_LAUNCH(DD, CORE_MATERN2); break;   \
      default: LGP_REGC_LAUNCH(DD, CORE_GENERIC); break;             \
    }
    switch (d) {
      case 1: LGP_REGC_BY_CORE(1); break;
      case 2: LGP_REGC_BY_CORE(2); break;
      case 3: LGP_REGC_BY_CORE(3); break;
      default: LGP_REGC_BY_CORE(0); break;
    }
#undef LGP_REGC_BY_CORE
#undef LGP_REGC_LAUNCH
    LGP_LAUNCH_CHECK();
    *ntr_out = ntr;
    return LGP_OK;
  }
  size_t tiles = 0;
  for (size_t tr = 0; tr < (ntr + RPB - 1) / RPB * RPB; ++tr) tiles += tr / RPB + 1;     // whole row groups (the kernel skips rows past n)
  dim3 gd((unsigned)tiles), bd(GT_THREADS);
  // the partial column sums of the tiles above the diagonal, which no workgroup owns any more: zero
  // ... and the sums along the rows: one slot per tile column, zero where no tile below the diagonal contributes
  if (drpart == dpart + ntr * n) {
    LGP_HIP(hipMemsetAsync(dpart, 0, (ntr + ntc) * n * sizeof(T), ctx->stream));
  } else {
    LGP_HIP(hipMemsetAsync(dpart, 0, ntr * n * sizeof(T), ctx->stream));
    LGP_HIP(hipMemsetAsync(drpart, 0, ntc * n * sizeof(T), ctx->stream));
  }
#define LGP_REG_LAUNCH(DD, CC) hipLaunchKernelGGL((gram_reg_kernel<T, DD, CC>), gd, bd, shmem, ctx->stream, g, noise, dnoise, dsvec, dpart, drpart)
#define LGP_REG_BY_CORE(DD)                                   \
  switch (core_of(kernel, p)) {                               \
    case CORE_EXPQUAD: LGP_REG_LAUNCH(DD, CORE_EXPQUAD); break;   \
    case CORE_MATERN2: LGP_REG_LAUNCH(DD, CORE_MATERN2); break;   \
    default: LGP_REG_LAUNCH(DD, CORE_GENERIC); break;             \
  }
  switch (d) {
    case 1: LGP_REG_BY_CORE(1); break;
    case 2: LGP_REG_BY_CORE(2); break;
    case 3: LGP_REG_BY_CORE(3); break;
    default: LGP_REG_BY_CORE(0); break;
  }
#undef LGP_REG_BY_CORE
#undef LGP_REG_LAUNCH
  LGP_LAUNCH_CHECK();
  *ntr_out = ntr;
  return LGP_OK;
}

// ------------------------------------------------------------------------------------------------
// BART

constexpr int BART_MAX_P = 64;
constexpr int BART_MAX_ROWS = 24;   // repeat*size <= 24 probabilities
constexpr int BT_ROWS = 16, BT_COLS = 256;

constexpr int BART_MAX_TAN = 2;    // forward-mode tangents carried beside the value (alpha, beta)

template <typename T>
struct BartArgs {
  const int32_t *ix;
  const int32_t *iy;
  T *out;
  const T *gamma_mat;   // optional per-pair gamma (n x m, ld ldo), may alias out
  const T *psi;         // psi[k] = digamma(k) for k = 1..kmax (psi[0] unused)
  long ldo;
  int n, m, p;
  int size, repeat;
  T gamma;
  int nsplits[BART_MAX_P];
  T w[BART_MAX_P];
  T probs[BART_MAX_ROWS];
  OutOps<T> ops;
  // pair-independent per-axis constants of the depth-2 closed form, evaluated once on the host in T arithmetic with the
  // operations and the order of the per-pair code (so the Gram entries do not change): valid when `pre` (global split totals)
  int pre;
  int sym;               // Gram form on one point set (ix == iy): evaluate one triangle, write both
  T wn[BART_MAX_P];      // [n != 0] w / n
  T invWm[BART_MAX_P];   // 1 / (Wn - [n != 0] w)
  T wiW[BART_MAX_P];     // w * (1 / Wn)
  T Wn, invWn;
  // forward-mode tangents: d probs / d theta_t laid out like probs; outputs tan[t] (ld ldo), optional per-pair tangents of
  // gamma from the deeper slice gtan[t] (may alias tan[t])
  int ntan;
  T dprobs[BART_MAX_TAN][BART_MAX_ROWS];
  T *tan[BART_MAX_TAN];
  const T *gtan[BART_MAX_TAN];
};

template <typename T> __device__ __forceinline__ T bart_div(T a, T b) { return a / b; }
// fp32: hardware reciprocal (1 ulp) instead of the ~10-instruction IEEE division sequence; the entries stay within a few
// ulp of the oracle (the reference's own tolerance for re-orderings of this fp32 evaluation is 16-64 ulp)
template <> __device__ __forceinline__ float bart_div<float>(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

// geometry of one pair: everything that does not depend on the splitting probabilities
template <typename T>
struct BartGeom {
  bool anyn0;
  T Wn, inv_Wn, sum_term, S, sumi;
};

// X(a), Y(a), N(a) give the (already weight-masked) indices and split totals of axis a.  PRE: the per-axis constants of
// BartArgs are valid (N(a) is the same for every pair)
template <typename T, bool PRE, typename FX, typename FY, typename FN>
__device__ __forceinline__ BartGeom<T> bart_geometry(const BartArgs<T> &g, FX X, FY Y, FN N) {
  const int p = g.p;
  const T one = T(1);
  BartGeom<T> q;
  q.anyn0 = false;
  q.Wn = q.inv_Wn = q.sum_term = q.S = q.sumi = T(0);
  for (int a = 0; a < p; ++a) q.anyn0 |= (X(a) != Y(a));    // reference: 64-bit hash inequality (:674-681)
  if (g.size == 1) return q;
  if (PRE) {
    q.Wn = g.Wn;
  } else {
    T Wn = T(0);                                            // :694
    for (int a = 0; a < p; ++a)
      if (N(a)) Wn += g.w[a];
    q.Wn = Wn;
  }
  if (g.size == 2) {                                        // :697-708
    T sum_term = T(0);
    for (int a = 0; a < p; ++a) {
      const int na = N(a), x = X(a), y = Y(a);
      const int n0 = x > y ? x - y : y - x;
      const T wn = PRE ? g.wn[a] : (na ? g.w[a] / T(na) : T(0));
      sum_term += wn * T(n0);
    }
    q.sum_term = sum_term;
    return q;
  }
  // size == 3, :711-757
  const T inv_Wn = PRE ? g.invWn : one / q.Wn;
  q.inv_Wn = inv_Wn;
  T S = T(0);
  for (int a = 0; a < p; ++a) {
    const int na = N(a), x = X(a), y = Y(a);
    const int n0 = x > y ? x - y : y - x;
    const T wn = PRE ? g.wn[a] : (na ? g.w[a] / T(na) : T(0));
    S += wn * T(na - n0);
  }
  T sumi = T(0);
  for (int a = 0; a < p; ++a) {
    const T w = g.w[a];
    const int na = N(a), x = X(a), y = Y(a);
    const bool xlty = x < y;
    const int mn = xlty ? x : y, mx = xlty ? y : x;
    const int n0 = mx - mn;
    const int nminus0 = mx, nplus0 = na - mn, nout = na - n0;
    const T inv_Wnmod = PRE ? g.invWm[a] : one / (q.Wn - (na ? w : T(0)));
    const T inv_Wnminus = nplus0 ? inv_Wn : inv_Wnmod;
    const T inv_Wnplus = nminus0 ? inv_Wn : inv_Wnmod;
    const T wn = PRE ? g.wn[a] : (na ? w / T(na) : T(0));
    const T wiW = PRE ? g.wiW[a] : w * inv_Wn;
    const T t = wn * T(n0);
    const T terms1 = (S + t) * (inv_Wnminus + inv_Wnplus + inv_Wn * T(nout - 2));
    T terms2 = nplus0 ? (PRE ? bart_div<T>(wiW * T(n0), T(nplus0)) : wiW * T(n0) / T(nplus0)) : w * inv_Wnmod;
    terms2 += nminus0 ? (PRE ? bart_div<T>(wiW * T(n0), T(nminus0)) : wiW * T(n0) / T(nminus0)) : w * inv_Wnmod;
    const T psin2 = T(2) * g.psi[na ? na : 1];     // (pair-independent in Gram mode: a uniform load)
    const T psiminus = g.psi[1 + mx];          // where(xlty, psi(1+iy), psi(1+ix)) = psi(1 + max)
    const T psiplus = g.psi[1 + na - mn];      // where(xlty, psi(1+n-ix), psi(1+n-iy)) = psi(1 + n - min)
    const T terms3 = wiW * T(n0) * (psin2 - psiminus - psiplus);
    sumi += wn * (terms1 - terms2 - terms3);
  }
  q.S = S;
  q.sumi = sumi;
  return q;
}

// the recursion over the `repeat` probability rows; NT tangents ride along (dgam[t] on entry = d gamma / d theta_t)
template <typename T, int NT>
__device__ __forceinline__ T bart_iterate(const BartArgs<T> &g, const BartGeom<T> &q, T gam, T (&dgam)[NT > 0 ? NT : 1]) {
  const T one = T(1);
  if (g.p == 0) {                                         // no covariates (:651-652)
#pragma unroll
    for (int t = 0; t < NT; ++t) dgam[t] = T(0);
    return one;
  }
  if (g.size == 1) {                                      // :685-691
    for (int k = 0; k < g.repeat; ++k) {
#pragma unroll
      for (int t = 0; t < NT; ++t) dgam[t] = q.anyn0 ? dgam[t] * g.probs[k] - (one - gam) * g.dprobs[t][k] : T(0);
      gam = q.anyn0 ? one - (one - gam) * g.probs[k] : one;
    }
    return gam;
  }
  if (g.size == 2) {                                      // :697-708
    for (int k = 0; k < g.repeat; ++k) {
      const T P0 = g.probs[2 * k], P1 = g.probs[2 * k + 1];
      const T Q = one - P1 + gam * P1;
      const T inner = P0 - P0 / q.Wn * q.sum_term;
      const T result = one - P0 + Q * inner;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const T dP0 = g.dprobs[t][2 * k], dP1 = g.dprobs[t][2 * k + 1];
        const T dQ = dgam[t] * P1 + (gam - one) * dP1;
        const T dinner = dP0 - dP0 / q.Wn * q.sum_term;
        dgam[t] = q.anyn0 ? dQ * inner + Q * dinner - dP0 : T(0);
      }
      gam = q.anyn0 ? result : one;
    }
    return gam;
  }
  for (int k = 0; k < g.repeat; ++k) {                    // :749-756
    const T P0 = g.probs[3 * k], P1 = g.probs[3 * k + 1], P2 = g.probs[3 * k + 2];
    const T Q = one + P2 * (gam - one);
    const T core = Q * q.sumi - q.S;
    const T sump = q.S + P1 * core;
    const T outer = q.inv_Wn * sump - one;
    const T result = one + P0 * outer;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const T dP0 = g.dprobs[t][3 * k], dP1 = g.dprobs[t][3 * k + 1], dP2 = g.dprobs[t][3 * k + 2];
      const T dQ = dP2 * (gam - one) + P2 * dgam[t];
      const T dsump = dP1 * core + P1 * (dQ * q.sumi);
      dgam[t] = q.anyn0 ? dP0 * outer + P0 * (q.inv_Wn * dsump) : T(0);
    }
    gam = q.anyn0 ? result : one;
  }
  return gam;
}

template <typename T, int NT>
__global__ void __launch_bounds__(BT_COLS) gram_bart_kernel(const BartArgs<T> g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int32_t *sx = reinterpret_cast<int32_t *>(smem_raw);      // [BT_ROWS][p]
  int32_t *sy = sx + BT_ROWS * g.p;                          // [p][BT_COLS]  (transposed: conflict-free)
  const int p = g.p;
  const int ntc = (g.m + BT_COLS - 1) / BT_COLS;
  int tr, tc;
  if (g.sym) {
    // One triangle when both point sets are the same (what `halfmatrix` of the reference does, _elements.py:567-573): the
    // kernel is bound by vector-ALU issue (715 instructions per pair at C5), so half the pairs is about half the time.  Only
    // the tiles that reach the diagonal or lie below it exist in the grid: tile row tr has tr / R + 1 of them (R = BT_COLS /
    // BT_ROWS), numbered consecutively so that the XCDs, which take workgroups round-robin, get equal shares (skipping tiles of
    // a rectangular grid left the XCDs of the high tile columns almost idle: 330 instead of 354 us for 53 % of the tiles).
    constexpr int R = BT_COLS / BT_ROWS;
    const int idx = (int)blockIdx.x;
    int grp = (int)((sqrtf(1.0f + 0.5f * (float)idx) - 1.0f) * 0.5f);          // largest grp with (R / 2) grp (grp + 1) <= idx ...
    while ((R / 2) * (grp + 1) * (grp + 2) <= idx) ++grp;                         // ... made exact
    while ((R / 2) * grp * (grp + 1) > idx) --grp;
    const int rem = idx - (R / 2) * grp * (grp + 1);
    tr = R * grp + rem / (grp + 1);
    tc = rem % (grp + 1);
  } else {
    tr = blockIdx.x / ntc;
    tc = blockIdx.x % ntc;
  }
  const int row0 = tr * BT_ROWS, col0 = tc * BT_COLS;
  const int tid = threadIdx.x;
  // a tile entirely below the diagonal also writes its transpose, which lies in tiles entirely above (the 256-column blocks
  // are whole multiples of the 16-row blocks); tiles that straddle the diagonal are evaluated in full
  const bool mirror = g.sym && (col0 + BT_COLS - 1 < row0);
  // zero-weight axes are ignored: n, ix, iy -> 0 (_bart.py:670-672)
  for (int idx = tid; idx < BT_ROWS * p; idx += BT_COLS) {
    const int r = idx / p, a = idx % p;
    int v = 0;
    if (row0 + r < g.n && g.w[a] != T(0)) v = g.ix[(long)(row0 + r) * p + a];
    sx[idx] = v;
  }
  for (int idx = tid; idx < BT_COLS * p; idx += BT_COLS) {
    const int c = idx / p, a = idx % p;
    int v = 0;
    if (col0 + c < g.m && g.w[a] != T(0)) v = g.iy[(long)(col0 + c) * p + a];
    sy[a * BT_COLS + c] = v;
  }
  __syncthreads();
  const int col = col0 + tid;
  if (col >= g.m) return;
  const T slope = NT > 0 ? g.ops.slope() : T(0);
  const int rmax = (g.n - row0) < BT_ROWS ? (g.n - row0) : BT_ROWS;
  for (int r = 0; r < rmax; ++r) {
    const long o = (long)(row0 + r) * g.ldo + col;
    const T gam0 = g.gamma_mat ? g.gamma_mat[o] : g.gamma;
    T dgam[NT > 0 ? NT : 1];
#pragma unroll
    for (int t = 0; t < NT; ++t) dgam[t] = g.gtan[t] ? g.gtan[t][o] : T(0);
    const BartGeom<T> q = bart_geometry<T, true>(
        g, [&](int a) { return sx[r * p + a]; }, [&](int a) { return sy[a * BT_COLS + tid]; },
        [&](int a) { return (g.w[a] != T(0)) ? g.nsplits[a] : 0; });
    const T gam = bart_iterate<T, NT>(g, q, gam0, dgam);
    const T val = g.ops.apply(gam);
    g.out[o] = val;
    const long om = (long)col * g.ldo + row0 + r;       // the mirrored entry
    if (mirror) g.out[om] = val;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      g.tan[t][o] = slope * dgam[t];
      if (mirror) g.tan[t][om] = slope * dgam[t];
    }
  }
}

// elementwise form: pair i uses ix[i], iy[i] and, if npair != NULL, its own split totals npair[i]
// (the fully vectorised BART.correlation(..., altinput=True) of _bart.py:301-455)
template <typename T>
__global__ void __launch_bounds__(256) bart_pairs_kernel(const BartArgs<T> g, const int32_t *npair) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= g.n) return;
  const int p = g.p;
  const int32_t *x = g.ix + i * p, *y = g.iy + i * p, *nn = npair ? npair + i * p : nullptr;
  const T gam0 = g.gamma_mat ? g.gamma_mat[i] : g.gamma;
  T nodual[1];
  const BartGeom<T> q = bart_geometry<T, false>(
      g, [&](int a) { return (g.w[a] != T(0)) ? x[a] : 0; }, [&](int a) { return (g.w[a] != T(0)) ? y[a] : 0; },
      [&](int a) { return (g.w[a] != T(0)) ? (nn ? nn[a] : g.nsplits[a]) : 0; });
  g.out[i] = g.ops.apply(bart_iterate<T, 0>(g, q, gam0, nodual));
}

// Recursion slices deeper than two levels (_bart.py:759-806): the correlation at a level is 1 - P (1 - sum_i w_i / n_i sum_k C_k / Wn)
// with C_k the correlation one level down when the cell is cut at split k of axis i (k runs over the splits that do not
// separate the two points: the first nminus_i cut on the left and leave k splits there, the others cut on the right).  E
// explicit levels by depth-first search with an explicit stack (no recursion on the device), the last three probabilities by
// the closed form above.  One thread per pair, the split counts of the pair in private arrays: the slow path -- (p n)^E closed
// forms per pair, as in the reference, whose documentation recommends `reset` instead.
constexpr int BART_DEEP_MAX_E = 8, BART_DEEP_MAX_P = 16;
template <typename T>
struct BartDeep {
  int E;
  T top[BART_DEEP_MAX_E];      // splitting probabilities of the explicit levels (the closed form's three are BartArgs::probs)
};

template <typename T>
__global__ void __launch_bounds__(128) bart_deep_kernel(const BartArgs<T> g, const BartDeep<T> dp, const int32_t *npair, int pairs) {
  const long idx = (long)blockIdx.x * 128 + threadIdx.x;
  const long total = pairs ? (long)g.n : (long)g.n * g.m;
  if (idx >= total) return;
  const long i = pairs ? idx : idx / g.m, j = pairs ? idx : idx % g.m;
  const int p = g.p;
  const int32_t *x = g.ix + i * p, *y = g.iy + j * p, *nn = npair ? npair + idx * p : nullptr;
  const long oidx = pairs ? idx : i * g.ldo + j;
  const T gam0 = g.gamma_mat ? g.gamma_mat[oidx] : g.gamma;
  const T one = T(1);
  int nminus[BART_DEEP_MAX_P], n0[BART_DEEP_MAX_P], nplus[BART_DEEP_MAX_P];
  bool anyn0 = false;
  for (int a = 0; a < p; ++a) {
    const bool on = g.w[a] != T(0);                      // zero-weight axes are ignored (:670-672)
    const int xa = on ? x[a] : 0, ya = on ? y[a] : 0, na = on ? (nn ? nn[a] : g.nsplits[a]) : 0;
    const int mn = xa < ya ? xa : ya, mx = xa < ya ? ya : xa;
    nminus[a] = mn; n0[a] = mx - mn; nplus[a] = na - mx;
    anyn0 |= xa != ya;
  }
  T result = one;
  if (anyn0 && p > 0) {
    struct Frame { int i, k, kend, smin, splus; T sumn, sump, Wn; };
    Frame st[BART_DEEP_MAX_E];
    int lvl = 0;
    T ret = T(0);
    bool entering = true;
    for (;;) {
      if (entering) {
        if (lvl == dp.E) {
          // the closed form on the current cell: ix = nminus, iy = nminus + n0, n = nminus + n0 + nplus
          T nodual[1];
          const BartGeom<T> q = bart_geometry<T, false>(
              g, [&](int a) { return nminus[a]; }, [&](int a) { return nminus[a] + n0[a]; }, [&](int a) { return nminus[a] + n0[a] + nplus[a]; });
          ret = bart_iterate<T, 0>(g, q, gam0, nodual);
          --lvl;
          entering = false;
          continue;
        }
        Frame &f = st[lvl];
        T Wn = T(0);
        for (int a = 0; a < p; ++a)
          if (nminus[a] + n0[a] + nplus[a]) Wn += g.w[a];
        f.Wn = Wn; f.sump = T(0); f.i = -1; f.k = 0; f.kend = 0; f.sumn = T(0);
      } else {
        Frame &f = st[lvl];
        f.sumn += ret;
        nminus[f.i] = f.smin; nplus[f.i] = f.splus;
        ++f.k;
      }
      Frame &f = st[lvl];
      // the next cut of this level, if any
      bool descend = false;
      for (;;) {
        if (f.i >= 0 && f.k < f.kend) { descend = true; break; }
        if (f.i >= 0) {
          const int ni = f.smin + n0[f.i] + f.splus;
          if (ni) f.sump += g.w[f.i] * f.sumn / T(ni);
        }
        ++f.i;
        if (f.i >= p) break;
        f.smin = nminus[f.i]; f.splus = nplus[f.i];
        const int ni = f.smin + n0[f.i] + f.splus;
        f.k = 0; f.kend = ni ? f.smin + f.splus : 0; f.sumn = T(0);
      }
      if (descend) {
        if (f.k < f.smin) nminus[f.i] = f.k;
        else nplus[f.i] = f.k - f.smin;
        ++lvl;
        entering = true;
        continue;
      }
      ret = one - dp.top[lvl] * (one - f.sump / f.Wn);     // :806 (anyn0 holds on every level: n0 never changes)
      if (lvl == 0) { result = ret; break; }
      --lvl;
      entering = false;
    }
  }
  g.out[oidx] = g.ops.apply(result);
}

template <typename T>
int gram_bart_launch(lgp_ctx *ctx, int p, const int32_t *dix, size_t n, const int32_t *diy, size_t m, const int32_t *hnsplits,
                     const T *hweights, int size, int repeat, const T *hprobs, T gamma, const T *dgamma_mat, const T *dpsi,
                     size_t npsi, int nops, const int *hops_code, const T *hops_val, T *dout, size_t ldo, bool pairs = false,
                     const int32_t *dnpair = nullptr, int ntan = 0, const T *hdprobs = nullptr, T *const *hdtan = nullptr,
                     const T *const *hdgtan = nullptr) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  if (n == 0 || m == 0) return LGP_OK;
  if (pairs) { LGP_REQUIRE(n == m, "pairs mode needs as many x as y points"); ldo = m; }
  LGP_REQUIRE(p >= 0 && p <= BART_MAX_P, "number of covariates must be in [0,64]");
  const bool deep = size > 3;
  LGP_REQUIRE(size >= 1 && size <= 3 + BART_DEEP_MAX_E, "recursion slice of at most 11 probabilities");
  LGP_REQUIRE(!deep || (repeat == 1 && ntan == 0 && p <= BART_DEEP_MAX_P), "slices deeper than two levels: no repetition, no tangents, at most 16 covariates");
  LGP_REQUIRE(repeat >= 1 && size * repeat <= BART_MAX_ROWS, "too many probabilities");
  LGP_REQUIRE(dout != nullptr && (p == 0 || (dix != nullptr && diy != nullptr)), "null pointer");
  LGP_REQUIRE(nops >= 0 && nops <= MAX_OPS, "at most 8 output operations");
  LGP_REQUIRE(n < (1u << 30) && m < (1u << 30) && ldo >= m, "bad dimensions");
  LGP_REQUIRE(ntan >= 0 && ntan <= BART_MAX_TAN && (ntan == 0 || (!pairs && hdprobs != nullptr && hdtan != nullptr)),
              "at most 2 tangents, Gram form only, with their probability derivatives and outputs");
  DeviceGuard dg(ctx->device);
  BartArgs<T> g;
  memset(&g, 0, sizeof(g));
  g.ix = dix; g.iy = diy; g.out = dout; g.gamma_mat = dgamma_mat; g.psi = dpsi;
  g.ldo = (long)ldo; g.n = (int)n; g.m = (int)m; g.p = p; g.size = deep ? 3 : size; g.repeat = repeat; g.gamma = gamma;
  BartDeep<T> dp;
  memset(&dp, 0, sizeof(dp));
  if (deep) {
    dp.E = size - 3;
    for (int l = 0; l < dp.E; ++l) dp.top[l] = hprobs[l];
    hprobs += dp.E;               // the closed form at the bottom takes the last three
    size = 3;
  }
  int maxn = 0;
  for (int a = 0; a < p; ++a) {
    LGP_REQUIRE(hweights[a] >= T(0), "weights must be nonnegative");
    g.w[a] = hweights[a];
    if (hnsplits) {
      LGP_REQUIRE(hnsplits[a] >= 0, "negative split count");
      g.nsplits[a] = hnsplits[a];
      if (hnsplits[a] > maxn) maxn = hnsplits[a];
    }
  }
  LGP_REQUIRE(hnsplits != nullptr || (pairs && dnpair != nullptr) || p == 0, "split totals missing");
  // with per-pair totals the caller guarantees npsi >= max total + 2
  if (size == 3) LGP_REQUIRE(dpsi != nullptr && npsi >= (size_t)maxn + 2, "digamma table too short");
  for (int i = 0; i < size * repeat; ++i) g.probs[i] = hprobs[i];
  g.ntan = ntan;
  for (int t = 0; t < ntan; ++t) {
    for (int i = 0; i < size * repeat; ++i) g.dprobs[t][i] = hdprobs[t * size * repeat + i];
    LGP_REQUIRE(hdtan[t] != nullptr, "null tangent output");
    g.tan[t] = hdtan[t];
    g.gtan[t] = hdgtan ? hdgtan[t] : nullptr;
  }
  g.ops.n = nops;
  for (int i = 0; i < nops; ++i) {
    LGP_REQUIRE(hops_code[i] == LGP_OP_MUL || hops_code[i] == LGP_OP_ADD, "unknown output operation");
    g.ops.code[i] = hops_code[i];
    g.ops.val[i] = hops_val[i];
  }
  if (deep) {
    const size_t total = pairs ? n : n * m;
    LGP_REQUIRE((total + 127) / 128 < (1ull << 31), "too many pairs");
    hipLaunchKernelGGL((bart_deep_kernel<T>), dim3((unsigned)((total + 127) / 128)), dim3(128), 0, ctx->stream, g, dp, dnpair, pairs ? 1 : 0);
    LGP_LAUNCH_CHECK();
    return LGP_OK;
  }
  if (pairs) {
    hipLaunchKernelGGL((bart_pairs_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, g, dnpair);
    LGP_LAUNCH_CHECK();
    return LGP_OK;
  }
  // Gram form: the split totals are global, so everything that depends on the axis alone is evaluated here, once, with
  // the same T-precision operations in the same order as the per-pair code (zero-weight axes count as n = 0, :670-672)
  {
    volatile T Wn = T(0);                                   // (volatile: plain T arithmetic, no contraction or widening)
    for (int a = 0; a < p; ++a)
      if (g.w[a] != T(0) && g.nsplits[a]) Wn = Wn + g.w[a];
    g.Wn = Wn;
    volatile T inv = T(1) / g.Wn;
    g.invWn = inv;
    for (int a = 0; a < p; ++a) {
      const int na = g.w[a] != T(0) ? g.nsplits[a] : 0;
      volatile T wn = na ? g.w[a] / T(na) : T(0);
      volatile T den = g.Wn - (na ? g.w[a] : T(0));
      volatile T im = T(1) / den;
      volatile T wi = g.w[a] * g.invWn;
      g.wn[a] = wn; g.invWm[a] = im; g.wiW[a] = wi;
    }
    g.pre = 1;
  }
  {
    // the same points on both sides, and a per-pair gamma / tangent input (a deeper slice chained in place) that is this
    // launch's own output or absent: then out(i, j) and out(j, i) are the same number up to the order of two operands
    static const bool nosym = getenv("LGP_BART_FULL") != nullptr;
    bool sym = !nosym && dix == diy && n == m && (dgamma_mat == nullptr || dgamma_mat == dout);
    for (int t = 0; t < ntan; ++t) sym = sym && (g.gtan[t] == nullptr || g.gtan[t] == g.tan[t]);
    g.sym = sym ? 1 : 0;
  }
  const size_t ntr = (n + BT_ROWS - 1) / BT_ROWS, ntc = (m + BT_COLS - 1) / BT_COLS;
  LGP_REQUIRE(ntr * ntc < (1ull << 31), "gram grid too large");
  const size_t shmem = (size_t)(p > 0 ? p : 1) * (BT_ROWS + BT_COLS) * sizeof(int32_t);
  size_t ntiles = ntr * ntc;
  if (g.sym) {      // the tiles that reach the diagonal or lie below it: tile row tr has tr / R + 1
    constexpr size_t R = BT_COLS / BT_ROWS;
    ntiles = 0;
    for (size_t r0 = 0; r0 < ntr; r0 += R) ntiles += (std::min(ntr, r0 + R) - r0) * (r0 / R + 1);
  }
  const dim3 gd((unsigned)ntiles), bd(BT_COLS);
  if (ntan == 0) hipLaunchKernelGGL((gram_bart_kernel<T, 0>), gd, bd, shmem, ctx->stream, g);
  else if (ntan == 1) hipLaunchKernelGGL((gram_bart_kernel<T, 1>), gd, bd, shmem, ctx->stream, g);
  else hipLaunchKernelGGL((gram_bart_kernel<T, 2>), gd, bd, shmem, ctx->stream, g);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

}  // namespace

extern "C" int lgp_gram_isotropic_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int d, const double *dx, size_t ldx, size_t n, const double *dy,
                                      size_t ldy, size_t m, int has_loc, double loc, int has_scale, double scale, int nops,
                                      const int *hops_code, const double *hops_val, int symmetric, double *dout, size_t ldo) {
  // (symmetric: (u-v)^2 == (v-u)^2 bit for bit, so one triangle evaluated and mirrored is the matrix the full evaluation writes)
  return gram_iso_launch<double>(ctx, kernel, p, npar, hpar, d, dx, ldx, n, dy, ldy, m, has_loc, loc, has_scale, scale, nops, hops_code, hops_val,
                                 dout, ldo, false, nullptr, symmetric != 0);
}
extern "C" int lgp_gram_isotropic_f32(lgp_ctx *ctx, int kernel, int p, int npar, const float *hpar, int d, const float *dx, size_t ldx, size_t n, const float *dy,
                                      size_t ldy, size_t m, int has_loc, float loc, int has_scale, float scale, int nops,
                                      const int *hops_code, const float *hops_val, int symmetric, float *dout, size_t ldo) {
  return gram_iso_launch<float>(ctx, kernel, p, npar, hpar, d, dx, ldx, n, dy, ldy, m, has_loc, loc, has_scale, scale, nops, hops_code, hops_val,
                                dout, ldo, false, nullptr, symmetric != 0);
}
extern "C" int lgp_gram_bart_f32(lgp_ctx *ctx, int p, const int32_t *dix, size_t n, const int32_t *diy, size_t m,
                                 const int32_t *hnsplits, const float *hweights, int size, int repeat, const float *hprobs, float gamma,
                                 const float *dgamma_mat, const float *dpsi, size_t npsi, int nops, const int *hops_code,
                                 const float *hops_val, float *dout, size_t ldo) {
  return gram_bart_launch<float>(ctx, p, dix, n, diy, m, hnsplits, hweights, size, repeat, hprobs, gamma, dgamma_mat, dpsi, npsi, nops,
                                 hops_code, hops_val, dout, ldo);
}
extern "C" int lgp_gram_bart_f64(lgp_ctx *ctx, int p, const int32_t *dix, size_t n, const int32_t *diy, size_t m,
                                 const int32_t *hnsplits, const double *hweights, int size, int repeat, const double *hprobs,
                                 double gamma, const double *dgamma_mat, const double *dpsi, size_t npsi, int nops,
                                 const int *hops_code, const double *hops_val, double *dout, size_t ldo) {
  return gram_bart_launch<double>(ctx, p, dix, n, diy, m, hnsplits, hweights, size, repeat, hprobs, gamma, dgamma_mat, dpsi, npsi, nops,
                                  hops_code, hops_val, dout, ldo);
}

extern "C" int lgp_gram_lagtable_f64(lgp_ctx *ctx, const double *dtable, size_t ntable, double fill, const double *dx, size_t ldx, size_t n, const double *dy,
                                     size_t ldy, size_t m, int has_loc, double loc, int has_scale, double scale, int nops, const int *hops_code,
                                     const double *hops_val, int symmetric, double *dout, size_t ldo) {
  return gram_iso_launch<double>(ctx, LGP_KERNEL_LAGTABLE, 0, 0, nullptr, 1, dx, ldx, n, dy, ldy, m, has_loc, loc, has_scale, scale, nops, hops_code, hops_val,
                                 dout, ldo, false, nullptr, symmetric != 0, dtable, ntable, fill);
}
extern "C" int lgp_gram_lagtable_f32(lgp_ctx *ctx, const float *dtable, size_t ntable, float fill, const float *dx, size_t ldx, size_t n, const float *dy,
                                     size_t ldy, size_t m, int has_loc, float loc, int has_scale, float scale, int nops, const int *hops_code,
                                     const float *hops_val, int symmetric, float *dout, size_t ldo) {
  return gram_iso_launch<float>(ctx, LGP_KERNEL_LAGTABLE, 0, 0, nullptr, 1, dx, ldx, n, dy, ldy, m, has_loc, loc, has_scale, scale, nops, hops_code, hops_val,
                                dout, ldo, false, nullptr, symmetric != 0, dtable, ntable, fill);
}
extern "C" int lgp_gram_pseries_f64(lgp_ctx *ctx, int mode, int uselog, const double *dtable, size_t ntable, const double *hpar4, const double *dx, size_t ldx,
                                    size_t n, const double *dy, size_t ldy, size_t m, int has_loc, double loc, int has_scale, double scale, int nops,
                                    const int *hops_code, const double *hops_val, int symmetric, double *dout, size_t ldo) {
  return gram_iso_launch<double>(ctx, LGP_KERNEL_PSERIES, (mode & 15) | (uselog ? 16 : 0), 4, hpar4, 1, dx, ldx, n, dy, ldy, m, has_loc, loc, has_scale, scale, nops,
                                 hops_code, hops_val, dout, ldo, false, nullptr, symmetric != 0, dtable, ntable, 0.);
}
extern "C" int lgp_gram_pseries_f32(lgp_ctx *ctx, int mode, int uselog, const float *dtable, size_t ntable, const float *hpar4, const float *dx, size_t ldx,
                                    size_t n, const float *dy, size_t ldy, size_t m, int has_loc, float loc, int has_scale, float scale, int nops,
                                    const int *hops_code, const float *hops_val, int symmetric, float *dout, size_t ldo) {
  return gram_iso_launch<float>(ctx, LGP_KERNEL_PSERIES, (mode & 15) | (uselog ? 16 : 0), 4, hpar4, 1, dx, ldx, n, dy, ldy, m, has_loc, loc, has_scale, scale, nops,
                                hops_code, hops_val, dout, ldo, false, nullptr, symmetric != 0, dtable, ntable, 0.f);
}
extern "C" int lgp_pairs_pseries_f64(lgp_ctx *ctx, int mode, int uselog, const double *dtable, size_t ntable, const double *hpar4, const double *dx, size_t ldx,
                                     const double *dy, size_t ldy, size_t npairs, int has_loc, double loc, int has_scale, double scale, int nops,
                                     const int *hops_code, const double *hops_val, double *dout) {
  return gram_iso_launch<double>(ctx, LGP_KERNEL_PSERIES, (mode & 15) | (uselog ? 16 : 0), 4, hpar4, 1, dx, ldx, npairs, dy, ldy, npairs, has_loc, loc, has_scale, scale,
                                 nops, hops_code, hops_val, dout, npairs, true, nullptr, false, dtable, ntable, 0.);
}
extern "C" int lgp_pairs_pseries_f32(lgp_ctx *ctx, int mode, int uselog, const float *dtable, size_t ntable, const float *hpar4, const float *dx, size_t ldx,
                                     const float *dy, size_t ldy, size_t npairs, int has_loc, float loc, int has_scale, float scale, int nops,
                                     const int *hops_code, const float *hops_val, float *dout) {
  return gram_iso_launch<float>(ctx, LGP_KERNEL_PSERIES, (mode & 15) | (uselog ? 16 : 0), 4, hpar4, 1, dx, ldx, npairs, dy, ldy, npairs, has_loc, loc, has_scale, scale,
                                nops, hops_code, hops_val, dout, npairs, true, nullptr, false, dtable, ntable, 0.f);
}
extern "C" int lgp_pairs_lagtable_f64(lgp_ctx *ctx, const double *dtable, size_t ntable, double fill, const double *dx, size_t ldx, const double *dy, size_t ldy,
                                      size_t npairs, int has_loc, double loc, int has_scale, double scale, int nops, const int *hops_code,
                                      const double *hops_val, double *dout) {
  return gram_iso_launch<double>(ctx, LGP_KERNEL_LAGTABLE, 0, 0, nullptr, 1, dx, ldx, npairs, dy, ldy, npairs, has_loc, loc, has_scale, scale, nops, hops_code,
                                 hops_val, dout, npairs, true, nullptr, false, dtable, ntable, fill);
}
extern "C" int lgp_pairs_lagtable_f32(lgp_ctx *ctx, const float *dtable, size_t ntable, float fill, const float *dx, size_t ldx, const float *dy, size_t ldy,
                                      size_t npairs, int has_loc, float loc, int has_scale, float scale, int nops, const int *hops_code,
                                      const float *hops_val, float *dout) {
  return gram_iso_launch<float>(ctx, LGP_KERNEL_LAGTABLE, 0, 0, nullptr, 1, dx, ldx, npairs, dy, ldy, npairs, has_loc, loc, has_scale, scale, nops, hops_code,
                                hops_val, dout, npairs, true, nullptr, false, dtable, ntable, fill);
}

namespace {
template <typename T>
__global__ void __launch_bounds__(256) regcols_reduce_kernel(size_t w, size_t ntr, const T *part, T *sums) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= w) return;
  T acc = T(0);
  for (size_t t = 0; t < ntr; ++t) acc += part[t * w + j];      // fixed order: deterministic
  sums[j] = acc;
}
template <typename T>
int gram_reg_cols_impl(lgp_ctx *ctx, int kernel, int p, int npar, const T *hpar, int d, const T *dx, size_t ldx, size_t n, size_t c0, size_t w, size_t cyc_nb, size_t cyc_stride, int has_loc,
                       T loc, int has_scale, T scale, int nops, const int *hops_code, const T *hops_val, T noise, const T *dnoise, T *dblock, size_t ldb,
                       T *dsvec, T *dsums) {
  LGP_REQUIRE(ctx != nullptr && dsums != nullptr, "null pointer");
  if (n == 0 || w == 0) return LGP_OK;
  const size_t ntr = (n + GR_ROWS - 1) / GR_ROWS;
  if (c0 == 0 && w == n && (cyc_nb == 0 || cyc_stride == cyc_nb)) {
    // the range is the whole matrix (the sharded path with one rank): the mirrored entries are all here, so the lower triangle
    // alone is evaluated, as for the single-GPU factorisation -- its tiles add their sums along the rows to the columns' as well
    const size_t ntc = (n + GT_THREADS * VecOf<T>::W - 1) / (GT_THREADS * VecOf<T>::W);
    T *part = nullptr;
    LGP_CHECK(lgp_ws_alloc(ctx, ctx->stream, (ntr + ntc) * n * sizeof(T), reinterpret_cast<void **>(&part)));
    WsGuard part_guard(ctx, ctx->stream, part);
    size_t got = 0;
    LGP_CHECK(gram_reg_launch<T>(ctx, kernel, p, npar, hpar, d, dx, ldx, n, has_loc, loc, has_scale, scale, nops, hops_code, hops_val, noise, dnoise, dblock, ldb,
                                 dsvec, part, part + ntr * n, &got));
    hipLaunchKernelGGL((regcols_reduce_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, ntr + ntc, part, dsums);
    LGP_LAUNCH_CHECK();
    return LGP_OK;
  }
  T *part = nullptr;
  LGP_CHECK(lgp_ws_alloc(ctx, ctx->stream, ntr * w * sizeof(T), reinterpret_cast<void **>(&part)));
  WsGuard part_guard(ctx, ctx->stream, part);
  LGP_HIP(hipMemsetAsync(part, 0, ntr * w * sizeof(T), ctx->stream));
  size_t got = 0;
  LGP_CHECK(gram_reg_launch<T>(ctx, kernel, p, npar, hpar, d, dx, ldx, n, has_loc, loc, has_scale, scale, nops, hops_code, hops_val, noise, dnoise, dblock, ldb,
                               dsvec, part, nullptr, &got, true, c0, w, cyc_nb, cyc_stride));
  // (columns of the padding behind a ragged last block: no tile wrote their partial sums)

  hipLaunchKernelGGL((regcols_reduce_kernel<T>), dim3((unsigned)((w + 255) / 256)), dim3(256), 0, ctx->stream, w, ntr, part, dsums);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}
}  // namespace

extern "C" int lgp_gram_reg_cols_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int d, const double *dx, size_t ldx, size_t n, size_t c0, size_t w,
                                     size_t cyc_nb, size_t cyc_stride, int has_loc, double loc, int has_scale, double scale, int nops, const int *hops_code, const double *hops_val, double noise,
                                     const double *dnoise, double *dblock, size_t ldb, double *dsvec, double *dsums) {
  return gram_reg_cols_impl<double>(ctx, kernel, p, npar, hpar, d, dx, ldx, n, c0, w, cyc_nb, cyc_stride, has_loc, loc, has_scale, scale, nops, hops_code, hops_val, noise, dnoise, dblock,
                                    ldb, dsvec, dsums);
}

extern "C" int lgp_gram_reg_cols_f32(lgp_ctx *ctx, int kernel, int p, int npar, const float *hpar, int d, const float *dx, size_t ldx, size_t n, size_t c0, size_t w,
                                     size_t cyc_nb, size_t cyc_stride, int has_loc, float loc, int has_scale, float scale, int nops, const int *hops_code, const float *hops_val, float noise,
                                     const float *dnoise, float *dblock, size_t ldb, float *dsvec, float *dsums) {
  return gram_reg_cols_impl<float>(ctx, kernel, p, npar, hpar, d, dx, ldx, n, c0, w, cyc_nb, cyc_stride, has_loc, loc, has_scale, scale, nops, hops_code, hops_val, noise, dnoise, dblock,
                                   ldb, dsvec, dsums);
}

// ---- elementwise ("pairs") forms: out[i] = k(x_i, y_i) ------------------------------------------------
extern "C" int lgp_pairs_isotropic_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int d, const double *dx, size_t ldx, const double *dy, size_t ldy,
                                       size_t npairs, int has_loc, double loc, int has_scale, double scale, int nops,
                                       const int *hops_code, const double *hops_val, double *dout) {
  return gram_iso_launch<double>(ctx, kernel, p, npar, hpar, d, dx, ldx, npairs, dy, ldy, npairs, has_loc, loc, has_scale, scale, nops, hops_code,
                                 hops_val, dout, npairs, true);
}
extern "C" int lgp_pairs_isotropic_f32(lgp_ctx *ctx, int kernel, int p, int npar, const float *hpar, int d, const float *dx, size_t ldx, const float *dy, size_t ldy,
                                       size_t npairs, int has_loc, float loc, int has_scale, float scale, int nops,
                                       const int *hops_code, const float *hops_val, float *dout) {
  return gram_iso_launch<float>(ctx, kernel, p, npar, hpar, d, dx, ldx, npairs, dy, ldy, npairs, has_loc, loc, has_scale, scale, nops, hops_code,
                                hops_val, dout, npairs, true);
}
extern "C" int lgp_pairs_bart_f32(lgp_ctx *ctx, int p, const int32_t *dix, const int32_t *diy, size_t npairs, const int32_t *hnsplits,
                                  const int32_t *dnpair, const float *hweights, int size, int repeat, const float *hprobs, float gamma,
                                  const float *dgamma_vec, const float *dpsi, size_t npsi, int nops, const int *hops_code,
                                  const float *hops_val, float *dout) {
  return gram_bart_launch<float>(ctx, p, dix, npairs, diy, npairs, hnsplits, hweights, size, repeat, hprobs, gamma, dgamma_vec, dpsi, npsi,
                                 nops, hops_code, hops_val, dout, npairs, true, dnpair);
}
extern "C" int lgp_pairs_bart_f64(lgp_ctx *ctx, int p, const int32_t *dix, const int32_t *diy, size_t npairs, const int32_t *hnsplits,
                                  const int32_t *dnpair, const double *hweights, int size, int repeat, const double *hprobs, double gamma,
                                  const double *dgamma_vec, const double *dpsi, size_t npsi, int nops, const int *hops_code,
                                  const double *hops_val, double *dout) {
  return gram_bart_launch<double>(ctx, p, dix, npairs, diy, npairs, hnsplits, hweights, size, repeat, hprobs, gamma, dgamma_vec, dpsi,
                                  npsi, nops, hops_code, hops_val, dout, npairs, true, dnpair);
}

// ---- Gram matrix together with its derivative with respect to `scale` (forward-mode tangent of _Kernel/_ops.py:310-326
// through the cores; what jax.jacfwd of the reference's gpfactory yields for a `scale` hyperparameter, _fit.py:679-690) ----
extern "C" int lgp_gram_isotropic_dscale_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int d, const double *dx, size_t ldx,
                                             size_t n, const double *dy, size_t ldy, size_t m, int has_loc, double loc, int has_scale,
                                             double scale, int nops, const int *hops_code, const double *hops_val, double *dout,
                                             double *ddscale, size_t ldo) {
  LGP_REQUIRE(ddscale != nullptr, "null derivative output");
  return gram_iso_launch<double>(ctx, kernel, p, npar, hpar, d, dx, ldx, n, dy, ldy, m, has_loc, loc, has_scale, scale, nops, hops_code, hops_val,
                                 dout, ldo, false, ddscale);
}
extern "C" int lgp_gram_isotropic_dscale_f32(lgp_ctx *ctx, int kernel, int p, int npar, const float *hpar, int d, const float *dx, size_t ldx,
                                             size_t n, const float *dy, size_t ldy, size_t m, int has_loc, float loc, int has_scale, float scale,
                                             int nops, const int *hops_code, const float *hops_val, float *dout, float *ddscale, size_t ldo) {
  LGP_REQUIRE(ddscale != nullptr, "null derivative output");
  return gram_iso_launch<float>(ctx, kernel, p, npar, hpar, d, dx, ldx, n, dy, ldy, m, has_loc, loc, has_scale, scale, nops, hops_code, hops_val,
                                dout, ldo, false, ddscale);
}

// ---- BART Gram matrix with forward-mode tangents of the splitting probabilities: what jax.jacfwd of the reference's
// gpfactory yields for the alpha / beta hyperparameters of bayestree.bart (bayestree/_bart.py:187-205, _fit.py:679-690).
// hdprobs: HOST ntan x (repeat*size) derivatives of hprobs; hdtan: HOST array of ntan DEVICE output matrices (n x m, ld ldo);
// hdgtan: HOST array of ntan DEVICE per-pair tangents of gamma from the deeper slice (may alias hdtan; NULL: zero) ----
extern "C" int lgp_gram_bart_jvp_f32(lgp_ctx *ctx, int p, const int32_t *dix, size_t n, const int32_t *diy, size_t m, const int32_t *hnsplits,
                                     const float *hweights, int size, int repeat, const float *hprobs, float gamma, const float *dgamma_mat,
                                     const float *dpsi, size_t npsi, int nops, const int *hops_code, const float *hops_val, float *dout,
                                     size_t ldo, int ntan, const float *hdprobs, float *const *hdtan, const float *const *hdgtan) {
  return gram_bart_launch<float>(ctx, p, dix, n, diy, m, hnsplits, hweights, size, repeat, hprobs, gamma, dgamma_mat, dpsi, npsi, nops,
                                 hops_code, hops_val, dout, ldo, false, nullptr, ntan, hdprobs, hdtan, hdgtan);
}
extern "C" int lgp_gram_bart_jvp_f64(lgp_ctx *ctx, int p, const int32_t *dix, size_t n, const int32_t *diy, size_t m, const int32_t *hnsplits,
                                     const double *hweights, int size, int repeat, const double *hprobs, double gamma, const double *dgamma_mat,
                                     const double *dpsi, size_t npsi, int nops, const int *hops_code, const double *hops_val, double *dout,
                                     size_t ldo, int ntan, const double *hdprobs, double *const *hdtan, const double *const *hdgtan) {
  return gram_bart_launch<double>(ctx, p, dix, n, diy, m, hnsplits, hweights, size, repeat, hprobs, gamma, dgamma_mat, dpsi, npsi, nops,
                                  hops_code, hops_val, dout, ldo, false, nullptr, ntan, hdprobs, hdtan, hdgtan);
}

// ---- Gram build + Chol.__init__ in one call: the matrix K = kernel(x, x) + diag(noise) goes straight into the factor ----
// ---- the whole prediction of a SMALL problem in one launch of one workgroup (round 5) --------------------------------------------
// docs/userguide/optim.rst:88-159 times `predfromdata` at N = 10: there the general path's fifteen launches (two Gram kernels, the
// fused build, scales, eps, a 34 us potf2 of a 10 x 10 block, substitutions, products, three copies) cost 0.25 ms of the 0.31.
// For n <= 64, m <= 256, n (n + m + 2) <= 18000 doubles of LDS everything happens here: points transformed, K = kernel(x, x) +
// noise built in full, s = pow2 scales (_decomp.py:356-361), K~ = K / s / s', Gershgorin bound and jitter (:349-354,245-255),
// Cholesky of K~ + eps I (:388), [Kxxs | y] scaled and solved against it, mean = X' x_y, cov = Kxsxs - X' X (_compute.py:254-260),
// log-determinant and |x_y|^2 (:484-488).  Inputs are read from, and the results written to, pinned host memory that the
// device maps: no copy engine, one launch, one synchronisation.  The factor, L = s L~, and the solved rows go to the device
// buffers the caller keeps for the questions that follow (Chol._adopt).
struct SmallIO {
  const double *hpts, *hy, *hnoise;     // mapped host memory: coordinate f of point i at hpts[f (n + m) + i]; noise may be null
  double *hout;                         // mapped host memory: cov (m m), mean (m), sum log L_ii, |L^-1 y|^2, eps, eps min s^2, info
  double *dP, *dL, *dnoise;             // device: the points, the factor with the carried rows below, the noise vector
  long ldl;
  int nx;
};

__global__ void __launch_bounds__(256) predict_small_kernel(const IsoArgs<double> g, const SmallIO io) {
  typedef double T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n = g.n, m = g.m, d = g.d, np = n + m, lda = n + 1, mw = m + 1;
  T *pt = reinterpret_cast<T *>(smem_raw);           // [d][n + m] transformed coordinates
  T *A = pt + (size_t)d * np;                        // [n][n + 1]
  T *W = A + (size_t)n * lda;                        // [n][m + 1]: Kxxs | y, then X = L~^-1 of it
  T *sinv = W + (size_t)n * mw;                      // [n] 1 / s_i
  T *red = sinv + n;                                 // [256] reductions
  __shared__ T sh_eps, sh_piv;
  __shared__ int sh_bad;
  const int tid = threadIdx.x;
  // Several workgroups: each repeats the small factorisation in its own LDS (a dependent chain a single workgroup is alone with
  // anyway) and takes its share of the m x m posterior covariance -- 10^4 kernel evaluations at m = 100, which one workgroup
  // needed 40 of the kernel's 63 us for (N = 10).  Workgroup 0 writes everything else.
  const bool first = blockIdx.x == 0;
  const int nwg = gridDim.x;
  for (int idx = tid; idx < d * np; idx += 256) {
    T v = io.hpts[idx];
    if (first) io.dP[idx] = v;
    if (g.has_loc) v = v - g.loc;
    if (g.has_scale) v = v / g.scale;
    pt[idx] = v;
  }
  const T k0 = g.ops.apply(iso_core<T>(g, T(0)));
  for (int i = tid; i < n; i += 256) {
    const T nz = io.hnoise ? io.hnoise[i] : T(0);
    if (io.hnoise && first) io.dnoise[i] = nz;
    sinv[i] = T(1) / pow2_scale<T>(k0 + nz);
  }
  if (tid == 0) sh_bad = 0;
  __syncthreads();
  auto pair = [&](int i, int j) {                    // kernel of points i and j (indices into the n + m points), output operations applied
    T r2 = T(0);
    for (int f = 0; f < d; ++f) {
      const T df = pt[f * np + i] - pt[f * np + j];
      r2 = (f == 0) ? df * df : r2 + df * df;
    }
    return g.ops.apply(iso_core<T>(g, r2));
  };
  for (int idx = tid; idx < n * n; idx += 256) {
    const int i = idx / n, j = idx % n;
    T v = pair(i, j);
    if (i == j) v = v + (io.hnoise ? io.hnoise[i] : T(0));
    A[i * lda + j] = v * sinv[j] * sinv[i];
  }
  for (int idx = tid; idx < n * mw; idx += 256) {
    const int i = idx / mw, c = idx % mw;
    const T v = c < m ? pair(i, n + c) : io.hy[i];
    W[i * mw + c] = v * sinv[i];
  }
  __syncthreads();
  // Gershgorin: max_i sum_j |K~_ij|, and min s^2
  T mx = -INFINITY, mn = INFINITY;
  int isnan_ = 0;
  for (int i = tid; i < n; i += 256) {
    T rs = T(0);
    for (int j = 0; j < n; ++j) rs += fabs(A[i * lda + j]);
    if (rs != rs) isnan_ = 1;
    mx = rs > mx ? rs : mx;
    const T si = T(1) / sinv[i], s2 = si * si;
    mn = s2 < mn ? s2 : mn;
  }
  red[tid] = isnan_ ? T(NAN) : mx;
  __syncthreads();
  if (tid == 0) {
    T v = -INFINITY;
    bool bad = false;
    for (int t = 0; t < 256 && t < n; ++t) { if (red[t] != red[t]) bad = true; v = red[t] > v ? red[t] : v; }
    const T maxeig = bad ? T(NAN) : v;
    sh_eps = T(n) * T(2.220446049250313e-16) * maxeig;      // epsrel = 'auto', epsabs = 0 (_decomp.py:245-255)
  }
  __syncthreads();
  red[tid] = mn;
  const T eps = sh_eps;
  __syncthreads();
  if (tid == 0 && first) {
    T v = INFINITY;
    for (int t = 0; t < 256 && t < n; ++t) v = red[t] < v ? red[t] : v;
    io.hout[(size_t)m * m + m + 2] = eps;
    io.hout[(size_t)m * m + m + 3] = eps * v;               // eps min s^2 (:393)
  }
  for (int i = tid; i < n; i += 256) A[i * lda + i] += eps;
  __syncthreads();
  // right-looking Cholesky of the lower triangle, column by column
  for (int k = 0; k < n; ++k) {
    if (tid == 0) {
      const T dk = A[k * lda + k];
      if (!(dk > T(0)) || dk + dk == dk) { if (!sh_bad) sh_bad = k + 1; sh_piv = T(NAN); }
      else sh_piv = sqrt(dk);
    }
    __syncthreads();
    const T lk = sh_piv;
    for (int i = k + tid; i < n; i += 256) A[i * lda + k] = (i == k) ? lk : A[i * lda + k] / lk;
    __syncthreads();
    const int rem = n - k - 1;
    for (int idx = tid; idx < rem * rem; idx += 256) {
      const int i = k + 1 + idx / rem, j = k + 1 + idx % rem;
      if (j <= i) A[i * lda + j] -= A[i * lda + k] * A[j * lda + k];
    }
    __syncthreads();
  }
  // X = L~^-1 W, one column per thread
  for (int c = tid; c < mw; c += 256) {
    for (int i = 0; i < n; ++i) {
      T acc = W[i * mw + c];
      for (int j = 0; j < i; ++j) acc -= A[i * lda + j] * W[j * mw + c];
      W[i * mw + c] = acc / A[i * lda + i];
    }
  }
  __syncthreads();
  // posterior: straight to the mapped host buffer
  for (int idx = blockIdx.x * 256 + tid; idx < m * m; idx += 256 * nwg) {
    const int a = idx / m, b = idx % m;
    T q = T(0);
    for (int i = 0; i < n; ++i) q += W[i * mw + a] * W[i * mw + b];
    io.hout[idx] = pair(n + a, n + b) - q;
  }
  if (!first) return;
  for (int a = tid; a < m; a += 256) {
    T q = T(0);
    for (int i = 0; i < n; ++i) q += W[i * mw + a] * W[i * mw + m];
    io.hout[(size_t)m * m + a] = q;
  }
  if (tid == 0) {
    T ld = T(0), q = T(0);
    for (int i = 0; i < n; ++i) { ld += log(A[i * lda + i] / sinv[i]); q += W[i * mw + m] * W[i * mw + m]; }
    io.hout[(size_t)m * m + m] = ld;
    io.hout[(size_t)m * m + m + 1] = q;
    io.hout[(size_t)m * m + m + 4] = (T)sh_bad;
  }
  // the factor L = s L~ (zeros above the diagonal) and the solved rows, for the caller's later questions
  for (int idx = tid; idx < n * n; idx += 256) {
    const int i = idx / n, j = idx % n;
    io.dL[(long)i * io.ldl + j] = j <= i ? A[i * lda + j] / sinv[i] : T(0);
  }
  for (int idx = tid; idx < io.nx * n; idx += 256) {
    const int c = idx / n, i = idx % n;
    io.dL[(long)(n + c) * io.ldl + i] = c < mw ? W[i * mw + c] : T(0);
  }
}

// whether lgp_predict_isotropic takes this road, and the road itself (api.hip)
bool lgp_predict_small_fits(int kernel, int d, size_t n, size_t m) {
  static const bool off = getenv("LGP_PREDICT_SMALL") != nullptr && atoi(getenv("LGP_PREDICT_SMALL")) == 0;
  const bool kind = kernel == LGP_KERNEL_EXPQUAD || kernel == LGP_KERNEL_MATERNP || kernel == LGP_KERNEL_GAMMAEXP || kernel == LGP_KERNEL_CAUCHY;
  return !off && kind && n >= 1 && n <= 64 && m >= 1 && m <= 256 && d >= 1 && (size_t)d * (n + m) + n * (n + m + 3) + 256 <= 18000;
}

int lgp_predict_small_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int d, const double *hpts, size_t n, size_t m, int has_loc, double loc,
                          int has_scale, double scale, int nops, const int *hops_code, const double *hops_val, const double *hy, const double *hnoise,
                          double *dP, double *dL, size_t ldl, size_t nx, double *ddblk, double *dnoise, double *hout, double *heps_out, int *info_out) {
  typedef double T;
  LGP_REQUIRE(lgp_predict_small_fits(kernel, d, n, m), "not a small problem");
  LGP_REQUIRE(p >= 0 && p <= MAX_P && nops >= 0 && nops <= MAX_OPS, "bad kernel parameters");
  LGP_REQUIRE(!has_scale || (scale > T(0) && scale < T(INFINITY)), "scale must be positive and finite");
  LGP_REQUIRE(!has_loc || (loc > T(-INFINITY) && loc < T(INFINITY)), "loc must be finite");
  DeviceGuard dg(ctx->device);
  const size_t np = n + m, nin = (size_t)d * np + 2 * n, nout = m * m + m + 5;
  if (ctx->h_small == nullptr) LGP_HIP(hipHostMalloc(&ctx->h_small, (size_t)1 << 20, hipHostMallocDefault));     // (18000 + 256^2 + ... doubles fit)
  T *hin = static_cast<T *>(ctx->h_small), *hres = hin + ((nin + 63) / 64) * 64;
  memcpy(hin, hpts, (size_t)d * np * sizeof(T));
  memcpy(hin + (size_t)d * np, hy, n * sizeof(T));
  if (hnoise) memcpy(hin + (size_t)d * np + n, hnoise, n * sizeof(T));
  void *dev = nullptr;
  LGP_HIP(hipHostGetDevicePointer(&dev, ctx->h_small, 0));
  T *din = static_cast<T *>(dev), *dres = din + (hres - hin);
  IsoArgs<T> g;
  g.x = g.y = nullptr; g.out = nullptr; g.dsc = nullptr;
  g.ldx = g.ldy = g.ldo = 0;
  g.n = (int)n; g.m = (int)m; g.d = d;
  g.kernel = kernel; g.p = p;
  g.has_loc = has_loc; g.has_scale = has_scale; g.loc = loc; g.scale = scale;
  for (int i = 0; i < MAX_P; ++i) g.coef[i] = T(0);
  int pos = 0;
  for (int k = p - 1; k >= 0; --k) g.coef[pos++] = (T)((double)(p - k) / ((double)(2 * p - k) * (double)(k + 1)));
  g.twop1 = (T)(2 * p + 1);
  g.posabs = 0; g.xy = 0;
  g.table = nullptr; g.ntable = 0; g.fill = T(0);
  g.cbase = 0; g.cyc_nb = 0; g.cyc_stride = 0; g.wloc = 0;
  for (int f = 0; f < 8; ++f) g.dord[f] = 0;
  g.lower_only = 0;
  g.par[0] = g.par[1] = g.par[2] = g.par[3] = T(0);
  g.flag = 0;
  if (kernel == LGP_KERNEL_GAMMAEXP) {
    LGP_REQUIRE(npar == 2 && hpar != nullptr && hpar[0] > T(0) && hpar[0] <= T(2), "GammaExp takes {gamma, gamma/2}, gamma in (0, 2]");
    g.par[0] = hpar[0]; g.par[1] = hpar[1]; g.flag = hpar[0] == T(2);
  } else if (kernel == LGP_KERNEL_CAUCHY) {
    LGP_REQUIRE(npar == 3 && hpar != nullptr && hpar[0] > T(0) && hpar[0] <= T(2) && hpar[1] > T(0), "Cauchy takes {alpha, beta, -beta/alpha}");
    g.par[0] = hpar[0] / T(2); g.par[1] = hpar[1]; g.par[2] = hpar[2]; g.flag = hpar[0] == T(2);
  } else {
    LGP_REQUIRE(npar == 0, "this kernel takes no real parameters");
  }
  g.ops.n = nops;
  for (int i = 0; i < MAX_OPS; ++i) { g.ops.code[i] = 0; g.ops.val[i] = T(0); }
  for (int i = 0; i < nops; ++i) {
    LGP_REQUIRE(hops_code[i] == LGP_OP_MUL || hops_code[i] == LGP_OP_ADD, "unknown output operation");
    g.ops.code[i] = hops_code[i];
    g.ops.val[i] = hops_val[i];
  }
  SmallIO io;
  io.hpts = din; io.hy = din + (size_t)d * np; io.hnoise = hnoise ? din + (size_t)d * np + n : nullptr;
  io.hout = dres;
  io.dP = dP; io.dL = dL; io.dnoise = dnoise; io.ldl = (long)ldl; io.nx = (int)nx;
  const size_t shmem = ((size_t)d * np + n * (n + 1) + n * (m + 1) + n + 256) * sizeof(T);
  static bool attr_set = false;
  if (!attr_set) {
    LGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(predict_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set = true;
  }
  // one workgroup per 1024 entries of the posterior covariance, at most 16 (LGP_PREDICT_SMALL_WGS)
  static const int wgs_env = getenv("LGP_PREDICT_SMALL_WGS") ? atoi(getenv("LGP_PREDICT_SMALL_WGS")) : 0;
  size_t nwg = wgs_env > 0 ? (size_t)wgs_env : (m * m + 1023) / 1024;
  nwg = nwg < 1 ? 1 : (nwg > 16 ? 16 : nwg);
  hipLaunchKernelGGL(predict_small_kernel, dim3((unsigned)nwg), dim3(256), shmem, ctx->stream, g, io);
  LGP_LAUNCH_CHECK();
  LGP_HIP(hipStreamSynchronize(ctx->stream));
  memcpy(hout, hres, (m * m + m + 2) * sizeof(T));
  if (heps_out) *heps_out = hres[m * m + m + 3];
  *info_out = (int)hres[m * m + m + 4];
  (void)nout;
  // the images of the diagonal blocks that later solves read (D, D', 1 / diag): enqueued, not waited for
  if (*info_out == 0) LGP_CHECK(lgp_extract_dblk_public(ctx, ctx->stream, dL, ldl, n, ddblk));
  return LGP_OK;
}

template <typename T>
int lgp_chol_from_prepared(lgp_ctx *ctx, size_t n, T *L, size_t ldl, T *dblk, size_t nrhs, T *ws, size_t ntr, T epsrel, T epsabs, int flags,
                           T *heps_out);   // potrf.hip

template <typename T>
static int gram_chol_impl(lgp_ctx *ctx, int kernel, int p, int npar, const T *hpar, int d, const T *dx, size_t ldx, size_t n, int has_loc, T loc,
                          int has_scale, T scale, int nops, const int *hops_code, const T *hops_val, T noise, const T *dnoise, T epsrel, T epsabs,
                          int flags, T *dL, size_t ldl, size_t nrhs, T *ddblk, T *heps_out) {
  LGP_REQUIRE(ctx != nullptr && ddblk != nullptr, "null pointer");
  LGP_REQUIRE(n >= 1 && n < (1u << 30), "matrix size out of range");
  DeviceGuard dg(ctx->device);
  const size_t ntr = (n + GR_ROWS - 1) / GR_ROWS;
  const size_t ntc_r = (n + GT_THREADS * VecOf<T>::W - 1) / (GT_THREADS * VecOf<T>::W);
  // workspace: s (n), Gershgorin sums (n), stats (16), partial column sums (ntr x n) followed by the sums along the rows
  // (ntc x n): the reduction adds all (ntr + ntc) x n partials in fixed order
  T *ws = nullptr;
  LGP_CHECK(lgp_ws_alloc(ctx, ctx->stream, (2 * n + 16 + (ntr + ntc_r) * n) * sizeof(T), reinterpret_cast<void **>(&ws)));
  WsGuard ws_guard(ctx, ctx->stream, ws);      // (the calls below end with a stream synchronisation: nothing is using it by then)
  size_t got = 0;
  int st = gram_reg_launch<T>(ctx, kernel, p, npar, hpar, d, dx, ldx, n, has_loc, loc, has_scale, scale, nops, hops_code, hops_val, noise, dnoise, dL,
                              ldl, ws, ws + 2 * n + 16, ws + 2 * n + 16 + ntr * n, &got, false, 0, 0, 0, 0, (flags & LGP_FACTOR_LOWER_ONLY) != 0);
  if (st == LGP_OK) st = lgp_chol_from_prepared<T>(ctx, n, dL, ldl, ddblk, nrhs, ws, ntr + ntc_r, epsrel, epsabs, flags, heps_out);
  return st;
}

extern "C" int lgp_gram_chol_factor_rhs_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int d, const double *dx, size_t ldx, size_t n,
                                            int has_loc, double loc, int has_scale, double scale, int nops, const int *hops_code,
                                            const double *hops_val, double noise, const double *dnoise, double epsrel, double epsabs, int eps_flags,
                                            double *dL, size_t ldl, size_t nrhs, double *ddblk, double *heps_out) {
  return gram_chol_impl<double>(ctx, kernel, p, npar, hpar, d, dx, ldx, n, has_loc, loc, has_scale, scale, nops, hops_code, hops_val, noise, dnoise,
                                epsrel, epsabs, eps_flags, dL, ldl, nrhs, ddblk, heps_out);
}
extern "C" int lgp_gram_chol_factor_rhs_f32(lgp_ctx *ctx, int kernel, int p, int npar, const float *hpar, int d, const float *dx, size_t ldx, size_t n,
                                            int has_loc, float loc, int has_scale, float scale, int nops, const int *hops_code, const float *hops_val,
                                            float noise, const float *dnoise, float epsrel, float epsabs, int eps_flags, float *dL, size_t ldl,
                                            size_t nrhs, float *ddblk, float *heps_out) {
  return gram_chol_impl<float>(ctx, kernel, p, npar, hpar, d, dx, ldx, n, has_loc, loc, has_scale, scale, nops, hops_code, hops_val, noise, dnoise,
                               epsrel, epsabs, eps_flags, dL, ldl, nrhs, ddblk, heps_out);
}
