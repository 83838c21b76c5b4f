/* Plain-C consumer of include/mogp_hip.h: proves the drop-in boundary needs nothing but a C compiler
 * (no Python, no torch types).  Fits the reference's 2 x 3 fixture (tests/test_GaussianProcess.py:16-22)
 * and prints the known answers of SURVEY.md section 8c item 1 for the caller to compare. */
#include <stdio.h>
#include <stdlib.h>
#include "mogp_hip.h"

#define CHECK(call)                                                      \
  do {                                                                   \
    if ((call) != 0) {                                                   \
      fprintf(stderr, "%s failed: %s\n", #call, mogp_last_error());      \
      return 2;                                                          \
    }                                                                    \
  } while (0)

int main(void) {
  if (!mogp_have_compatible_device()) {
    fprintf(stderr, "no gfx950 device\n");
    return 3;
  }
  const double X[6] = {1., 2., 3., 4., 5., 6.};
  const double t[2] = {2., 4.};
  const double theta[4] = {1., 1., 1., 1.};
  mogp_densegp* gp = mogp_densegp_create(X, 2, 3, t, 16, NULL, MOGP_SQUARED_EXPONENTIAL, MOGP_NUG_FIXED, 0.0);
  if (!gp) {
    fprintf(stderr, "create failed: %s\n", mogp_last_error());
    return 2;
  }
  double logpost = 0., grad[4], alpha[2], mean = 0., var = 0.;
  const double xs[3] = {2., 3., 4.};
  CHECK(mogp_densegp_get_logpost(gp, theta, 4, &logpost));
  CHECK(mogp_densegp_fit(gp, theta, 4));
  CHECK(mogp_densegp_logpost_deriv(gp, grad, 4));
  CHECK(mogp_densegp_get_invQt(gp, alpha));
  CHECK(mogp_densegp_predict_variance_batch(gp, xs, 1, 3, &mean, &var, 1));
  printf("logpost %.15e\nalpha %.15e %.15e\ngrad3 %.15e\nmean %.15e\nvar %.15e\n", logpost, alpha[0], alpha[1], grad[3], mean, var);
  /* error path: wrong theta length must fail with the reference's message, not crash */
  if (mogp_densegp_fit(gp, theta, 3) == 0) return 4;
  printf("error %s\n", mogp_last_error());
  mogp_densegp_destroy(gp);
  return 0;
}
