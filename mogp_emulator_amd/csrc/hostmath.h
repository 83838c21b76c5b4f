// Host-side O(D) scalar pieces of the path: priors, mean functions, parameter transforms.
// Counterparts of mogp_gpu/src/gppriors.hpp, meanfunc.hpp, gpparams.hpp (arithmetic follows the
// CPU oracle: Priors.py:291-354, 842-1128; GPParams.py:35-147).
#pragma once
#include <cmath>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

namespace mogp {

enum { PRIOR_INVGAMMA = 0, PRIOR_GAMMA = 1, PRIOR_LOGNORMAL = 2, PRIOR_WEAK = 3 };
enum { NUG_ADAPTIVE = 0, NUG_FIT = 1, NUG_FIXED = 2, NUG_PIVOT = 3 };

struct Prior {
  int type = PRIOR_WEAK;
  double shape = 0., scale = 0.;

  double logp(double x) const {
    switch (type) {
      case PRIOR_INVGAMMA: return shape * std::log(scale) - std::lgamma(shape) - (shape + 1.) * std::log(x) - scale / x;
      case PRIOR_GAMMA: return -shape * std::log(scale) - std::lgamma(shape) + (shape - 1.) * std::log(x) - x / scale;
      case PRIOR_LOGNORMAL: {
        const double u = std::log(x / scale) / shape;
        return -0.5 * u * u - 0.5 * std::log(2. * M_PI) - std::log(x) - std::log(shape);
      }
      default: return 0.;
    }
  }
  double dlogpdx(double x) const {
    switch (type) {
      case PRIOR_INVGAMMA: return -(shape + 1.) / x + scale / (x * x);
      case PRIOR_GAMMA: return (shape - 1.) / x - 1. / scale;
      case PRIOR_LOGNORMAL: return -std::log(x / scale) / (shape * shape) / x - 1. / x;
      default: return 0.;
    }
  }
  // draw of the scaled variable (Priors.py sample_x); weak priors sample the RAW variable
  template <class RNG>
  double sample_x(RNG& rng) const {
    switch (type) {
      case PRIOR_INVGAMMA: { std::gamma_distribution<double> g(shape, 1.0); return scale / g(rng); }
      case PRIOR_GAMMA: { std::gamma_distribution<double> g(shape, scale); return g(rng); }
      case PRIOR_LOGNORMAL: { std::normal_distribution<double> nd(std::log(scale), shape); return std::exp(nd(rng)); }
      default: return 0.;
    }
  }
};

// theta_data layout: [corr_raw (D) | log sigma^2 | log nugget (fit only)]
struct Priors {
  std::vector<Prior> corr;
  Prior cov, nug;
  bool created = false;

  double logp(const std::vector<double>& th, int D, int nug_type) const {
    if (!created) return 0.;
    double lp = 0.;
    for (int d = 0; d < D; ++d) lp += corr[d].logp(std::exp(-0.5 * th[d]));
    lp += cov.logp(std::exp(th[D]));
    if (nug_type == NUG_FIT) lp += nug.logp(std::exp(th[D + 1]));
    return lp;
  }
  void dlogpdtheta(const std::vector<double>& th, int D, int nug_type, double* out) const {
    const int nd = D + 1 + (nug_type == NUG_FIT ? 1 : 0);
    for (int i = 0; i < nd; ++i) out[i] = 0.;
    if (!created) return;
    for (int d = 0; d < D; ++d) {
      const double l = std::exp(-0.5 * th[d]);
      out[d] = corr[d].dlogpdx(l) * (-0.5 * l);
    }
    const double s2 = std::exp(th[D]);
    out[D] = cov.dlogpdx(s2) * s2;
    if (nug_type == NUG_FIT) {
      const double eta = std::exp(th[D + 1]);
      out[D + 1] = nug.dlogpdx(eta) * eta;
    }
  }
  template <class RNG>
  void sample(RNG& rng, int D, int nug_type, double* out) const {
    std::uniform_real_distribution<double> U(0., 1.);
    auto draw = [&](const Prior& p, bool corr_tr) {
      if (!created || p.type == PRIOR_WEAK) return 5. * (U(rng) - 0.5);     // WeakPrior.sample, Priors.py:636-649
      const double x = p.sample_x(rng);
      return corr_tr ? -2. * std::log(x) : std::log(x);
    };
    for (int d = 0; d < D; ++d) out[d] = draw(created ? corr[d] : Prior(), true);
    out[D] = draw(cov, false);
    if (nug_type == NUG_FIT) out[D + 1] = draw(nug, false);
  }
};

// Mean functions evaluated on the host (meanfunc.hpp).  kind: 0 zero, 1 fixed, 2 const, 3 poly.
struct MeanFunc {
  int kind = 0;
  double value = 0.;
  std::vector<int> dims, powers;

  int n_params() const { return kind == 2 ? 1 : (kind == 3 ? 1 + (int)dims.size() : 0); }
  void check(int np, int D) const {
    if (np != n_params()) throw std::runtime_error("Expected params list of length " + std::to_string(n_params()));
    for (int d : dims)
      if (d >= D) throw std::runtime_error("Dimension index must be less than D");
  }
  void mean_f(const double* xs, int m, int D, const double* p, int np, double* out) const {
    check(np, D);
    for (int i = 0; i < m; ++i) {
      double v = 0.;
      if (kind == 1) v = value;
      else if (kind == 2) v = p[0];
      else if (kind == 3) {
        v = p[0];
        for (size_t t = 0; t < dims.size(); ++t) v += p[t + 1] * std::pow(xs[(size_t)i * D + dims[t]], powers[t]);
      }
      out[i] = v;
    }
  }
  // out (n_params, m)
  void mean_deriv(const double* xs, int m, int D, const double* p, int np, double* out) const {
    check(np, D);
    if (kind == 2) for (int i = 0; i < m; ++i) out[i] = 1.;
    if (kind == 3) {
      for (int i = 0; i < m; ++i) out[i] = 1.;
      for (size_t t = 0; t < dims.size(); ++t)
        for (int i = 0; i < m; ++i) out[(t + 1) * m + i] = std::pow(xs[(size_t)i * D + dims[t]], powers[t]);
    }
  }
  // out (D, m)
  void mean_inputderiv(const double* xs, int m, int D, const double* p, int np, double* out) const {
    check(np, D);
    for (size_t e = 0; e < (size_t)D * m; ++e) out[e] = 0.;
    if (kind == 3)
      for (size_t t = 0; t < dims.size(); ++t)
        for (int i = 0; i < m; ++i)
          out[(size_t)dims[t] * m + i] += p[t + 1] * powers[t] * std::pow(xs[(size_t)i * D + dims[t]], powers[t] - 1);
  }
};

}  // namespace mogp
