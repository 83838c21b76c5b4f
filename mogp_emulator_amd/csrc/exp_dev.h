// The covariance kernels' exponential.  Plain arithmetic, so the SAME text compiles for the device (hipcc) and for the host (g++:
// tests/c/exp_check.cpp holds it to long double) -- only the two bit-cast helpers differ.
#pragma once
#include "exp_tab.h"
#ifdef __HIPCC__
#define MOGP_EXP_FN __device__ __forceinline__
#define MOGP_HI(x) __double2hiint(x)
#define MOGP_LO(x) __double2loint(x)
#define MOGP_HILO(h, l) __hiloint2double(h, l)
#else
#include <cmath>
#include <cstring>
#define MOGP_EXP_FN inline
static inline int MOGP_HI(double x) { long long b; std::memcpy(&b, &x, 8); return (int)(b >> 32); }
static inline int MOGP_LO(double x) { long long b; std::memcpy(&b, &x, 8); return (int)b; }
static inline double MOGP_HILO(int h, int l) { long long b = ((long long)h << 32) | (unsigned)l; double x; std::memcpy(&x, &b, 8); return x; }
#endif

namespace mogp {

// exp(-x) (HALF: exp(-x/2)) for x >= 0 in 14 vector-ALU instructions where ocml's exp costs ~35 inside the covariance kernels,
// which are bound by vector-ALU issue (profiles/r04_*_pmc_sq_valu_B64.txt): -x = (256 m + j) ln2/256 - w with |w| <= ln2/512,
// exp = 2^m * tab[j] * (1 + p(w)), p of degree 4 (economised Taylor: truncation 3e-18).  The integer 256 m + j is read from the low
// mantissa bits of x * 256/ln2 + 1.5 * 2^52 (no conversion), the reduction uses ln2/256 as hi + lo through two fmas, 2^m is applied with
// v_ldexp_f64 (gradual underflow as libm's).  Error <= 1.3 ulp over [0, 1500] (tests/c/exp_check.cpp holds the same arithmetic on
// the host to long double); arguments above 768 (HALF: 1536) give 0 like libm, NaN stays NaN (only the high word is clamped).
// HALF folds the factor 1/2 into the constants by exact powers of two: lean_exp_neg<true>(x) == lean_exp_neg<false>(x / 2) bit for bit.
template <bool HALF>
MOGP_EXP_FN double lean_exp_neg(double x, const double* tab) {
  constexpr double XMAX = HALF ? 1536.0 : 768.0;
  constexpr int XMAX_HI = HALF ? 0x40980000 : 0x40880000;
  int hi = MOGP_HI(x);
  hi = (x > XMAX) ? XMAX_HI : hi;
  x = MOGP_HILO(hi, MOGP_LO(x));
  constexpr double MAGIC = 6755399441055744.0;   // 1.5 * 2^52
  constexpr double S = HALF ? 0.5 : 1.0;
  constexpr double A2 = (EXPN_CHI / 2) * (EXPN_CHI / 2);
  constexpr double C1 = -1.0 + A2 * A2 / 384.0, C2 = 0.5, C3 = -1.0 / 6.0 - A2 / 96.0, C4 = 1.0 / 24.0;
  const double kd = __builtin_fma(x, -EXPN_L * S, MAGIC);
  const double kn = kd - MAGIC;
  const int ki = MOGP_LO(kd);
  double w = __builtin_fma(kn, EXPN_CHI / S, x);
  w = __builtin_fma(kn, EXPN_CLO / S, w);
  double q = __builtin_fma(w, C4 * S * S * S * S, C3 * S * S * S);
  q = __builtin_fma(q, w, C2 * S * S);
  q = __builtin_fma(q, w, C1 * S);
  const double p = q * w;
  const double t = tab[ki & 255];
  const double y = __builtin_fma(t, p, t);
  return __builtin_ldexp(y, ki >> 8);
}

}  // namespace mogp
