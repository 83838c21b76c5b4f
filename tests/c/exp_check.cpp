// Host check of the covariance kernels' exponential (mogp_emulator_amd/csrc/exp_dev.h): the SAME source text the device compiles,
// held to long double expl over the argument ranges the kernels meet.  Prints "max_ulp_full max_ulp_half mismatch_half_vs_full specials_ok".
// Built and run by tests/test_host_boundary.py::test_lean_exp_matches_long_double (g++ -O2 -ffp-contract=off).
#include "exp_dev.h"
#include <cstdio>
#include <cstdint>
#include <cmath>
static const double TAB[256] = {MOGP_EXP_TAB_VALUES};
static uint64_t st = 88172645463325252ull;
static double urand() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) * (1.0 / 9007199254740992.0); }
static double ulps(double y, long double t) {
  const double td = (double)t;
  const long double ul = (long double)std::nextafter(std::fabs(td), INFINITY) - (long double)std::fabs(td);
  return (double)(fabsl((long double)y - t) / ul);
}
int main(int argc, char** argv) {
  const long N = argc > 1 ? atol(argv[1]) : 4000000;
  double m1 = 0, m2 = 0;
  long mism = 0;
  for (long it = 0; it < N; ++it) {
    const double u = urand();
    double x;
    switch (it & 3) {
      case 0: x = u * 1400.0; break;                 // down to the smallest normal numbers (denormal results round twice: excluded from the ulp bar)
      case 1: x = u * 40.0; break;
      case 2: x = std::exp(-40.0 * u); break;
      default: x = u * 1e-3;
    }
    const double y1 = mogp::lean_exp_neg<false>(x, TAB), y2 = mogp::lean_exp_neg<true>(x, TAB);
    if (x < 708.0) m1 = std::fmax(m1, ulps(y1, expl(-(long double)x)));
    m2 = std::fmax(m2, ulps(y2, expl(-(long double)x / 2)));
    if (mogp::lean_exp_neg<true>(2.0 * x, TAB) != y1) ++mism;
  }
  bool ok = true;
  const double sp[] = {0.0, 1e-300, 5e-324, 700.0, 744.0, 745.0, 745.2, 750.0, 767.9, 768.0, 769.0, 1535.0, 1536.0, 1537.0, 1e10, 1e308, INFINITY};
  for (double x : sp) {
    const double a = mogp::lean_exp_neg<false>(x, TAB), b = mogp::lean_exp_neg<true>(x, TAB);
    const double ra = std::exp(-x), rb = std::exp(-x / 2);
    // denormal results: within one denormal step; otherwise 2 ulp
    ok = ok && std::fabs(a - ra) <= std::fmax(2 * (std::nextafter(ra, INFINITY) - ra), 5e-324);
    ok = ok && std::fabs(b - rb) <= std::fmax(2 * (std::nextafter(rb, INFINITY) - rb), 5e-324);
  }
  ok = ok && std::isnan(mogp::lean_exp_neg<false>(NAN, TAB)) && std::isnan(mogp::lean_exp_neg<true>(NAN, TAB));
  ok = ok && mogp::lean_exp_neg<false>(0.0, TAB) == 1.0 && mogp::lean_exp_neg<true>(0.0, TAB) == 1.0;
  printf("%.4f %.4f %ld %d\n", m1, m2, mism, ok ? 1 : 0);
  return 0;
}
