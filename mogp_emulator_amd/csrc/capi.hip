// extern "C" surface of libmogp_hip.so (include/mogp_hip.h).  Every entry point converts C++
// exceptions into (non-zero status, thread-local message); the Python shim raises RuntimeError
// with that message, matching the std::runtime_error -> RuntimeError mapping of pybind11 in the
// reference (bindings.cu).
#include <cstring>
#include <limits>
#include <memory>

#include "../../include/mogp_hip.h"
#include "engine.h"

using namespace mogp;

static thread_local std::string g_err;

struct mogp_meanfunc { MeanFunc mf; };
struct mogp_densegp { Engine* eng; int idx; bool owns; };
struct mogp_mogp { std::unique_ptr<Engine> eng; std::vector<mogp_densegp> views; double nug_size0; int nug_type0; };

#define GUARD(body)                      \
  try {                                  \
    body;                                \
    return 0;                            \
  } catch (const std::exception& e) {    \
    g_err = e.what();                    \
    return 1;                            \
  } catch (...) {                        \
    g_err = "unknown error";             \
    return 1;                            \
  }

static void set_priors(Engine* eng, int i, int n_corr, const int* ct, const double* cp, int covt, const double* covp, int nugt,
                       const double* nugp) {
  if (n_corr != eng->NC) throw std::runtime_error("number of correlation priors must equal the number of correlation parameters");
  Priors pr;
  pr.corr.resize(n_corr);
  for (int d = 0; d < n_corr; ++d) {
    pr.corr[d].type = ct[d];
    pr.corr[d].shape = cp[2 * d];
    pr.corr[d].scale = cp[2 * d + 1];
  }
  pr.cov.type = covt; pr.cov.shape = covp[0]; pr.cov.scale = covp[1];
  pr.nug.type = nugt; pr.nug.shape = nugp[0]; pr.nug.scale = nugp[1];
  pr.created = true;
  eng->gp[i].pri = pr;
  eng->gp[i].logpost_stale = true;      // the cached log-posterior was computed with the old priors
}

static void kernel_eval_impl(int kernel_type, int what, const double* x1, int n1, const double* x2, int n2, int D, const double* params,
                             int n_params, double* out) {
  {
    if (kernel_type < 0 || kernel_type > 4) throw std::runtime_error("Unrecognized kernel type\n");
    if (what < 0 || what > 2) throw std::runtime_error("kernel_eval: what must be 0 (f), 1 (deriv) or 2 (inputderiv)");
    if (n1 < 1 || n2 < 1 || D < 1) throw std::runtime_error("kernel inputs must have shape (n, D) with n, D >= 1");
    const bool uniform = kernel_type >= 3;
    const int nc = uniform ? 1 : D;
    if (n_params != nc + 1) throw std::runtime_error("Expected params list of length " + std::to_string(nc + 1));
    const int dk = kernel_type == 3 ? 0 : (kernel_type == 4 ? 1 : kernel_type);
    std::vector<double> P(D + 1);
    for (int d = 0; d < D; ++d) P[d] = std::exp(params[uniform ? 0 : d]);
    P[D] = std::exp(params[nc]);
    const size_t planes = what == 0 ? 1 : (what == 1 ? (size_t)D + 1 : (size_t)D);
    const size_t cnt = planes * (size_t)n1 * n2;
    double *d1 = nullptr, *d2 = nullptr, *dP = nullptr, *dO = nullptr;
    auto cleanup = [&]() { for (double* p : {d1, d2, dP, dO}) if (p) hipFree(p); };
    try {
      hip_check(hipMalloc(reinterpret_cast<void**>(&d1), (size_t)n1 * D * 8), "hipMalloc");
      hip_check(hipMalloc(reinterpret_cast<void**>(&d2), (size_t)n2 * D * 8), "hipMalloc");
      hip_check(hipMalloc(reinterpret_cast<void**>(&dP), P.size() * 8), "hipMalloc");
      hip_check(hipMalloc(reinterpret_cast<void**>(&dO), cnt * 8), "hipMalloc");
      hip_check(hipMemcpy(d1, x1, (size_t)n1 * D * 8, hipMemcpyHostToDevice), "hipMemcpy");
      hip_check(hipMemcpy(d2, x2, (size_t)n2 * D * 8, hipMemcpyHostToDevice), "hipMemcpy");
      hip_check(hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice), "hipMemcpy");
      launch_kernel_object(dk, d1, n1, d2, n2, D, dP, what, dO, nullptr);
      std::vector<double> tmp(cnt);
      hip_check(hipMemcpy(tmp.data(), dO, cnt * 8, hipMemcpyDeviceToHost), "hipMemcpy");
      hip_check(hipGetLastError(), "kernel_object_kernel");
      if (what == 1 && uniform) {
        // one shared length scale: d/dtheta_0 = sum of the per-dimension planes (Kernel.py:338-376); then the sigma^2 plane
        const size_t pl = (size_t)n1 * n2;
        for (size_t e = 0; e < pl; ++e) {
          double s = 0.;
          for (int d = 0; d < D; ++d) s += tmp[(size_t)d * pl + e];
          out[e] = s;
          out[pl + e] = tmp[(size_t)D * pl + e];
        }
      } else {
        std::memcpy(out, tmp.data(), cnt * 8);
      }
    } catch (...) {
      cleanup();
      throw;
    }
    cleanup();
  }
}


extern "C" {

const char* mogp_last_error(void) { return g_err.c_str(); }
const char* mogp_version(void) { return "mogp-hip 0.1 (gfx950)"; }

int mogp_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}
int mogp_have_compatible_device(void) {
  int c = mogp_device_count();
  for (int d = 0; d < c; ++d) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, d) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) return 1;
  }
  return 0;
}
int mogp_set_device(int device) { GUARD(hip_check(hipSetDevice(device), "hipSetDevice")); }

// ---- mean functions ---------------------------------------------------------------------------
mogp_meanfunc* mogp_meanfunc_zero(void) { auto* m = new mogp_meanfunc; m->mf.kind = 0; return m; }
mogp_meanfunc* mogp_meanfunc_fixed(double v) { auto* m = new mogp_meanfunc; m->mf.kind = 1; m->mf.value = v; return m; }
mogp_meanfunc* mogp_meanfunc_const(void) { auto* m = new mogp_meanfunc; m->mf.kind = 2; return m; }
mogp_meanfunc* mogp_meanfunc_poly(const int* dims, const int* powers, int nterms) {
  auto* m = new mogp_meanfunc;
  m->mf.kind = 3;
  m->mf.dims.assign(dims, dims + nterms);
  m->mf.powers.assign(powers, powers + nterms);
  return m;
}
void mogp_meanfunc_destroy(mogp_meanfunc* m) { delete m; }
int mogp_meanfunc_n_params(const mogp_meanfunc* m) { return m->mf.n_params(); }
int mogp_meanfunc_mean_f(const mogp_meanfunc* m, const double* xs, int mm, int D, const double* p, int np, double* out) {
  GUARD(m->mf.mean_f(xs, mm, D, p, np, out));
}
int mogp_meanfunc_mean_deriv(const mogp_meanfunc* m, const double* xs, int mm, int D, const double* p, int np, double* out) {
  GUARD(m->mf.mean_deriv(xs, mm, D, p, np, out));
}
int mogp_meanfunc_mean_inputderiv(const mogp_meanfunc* m, const double* xs, int mm, int D, const double* p, int np, double* out) {
  GUARD(m->mf.mean_inputderiv(xs, mm, D, p, np, out));
}

// ---- DenseGP_GPU --------------------------------------------------------------------------------
static mogp_densegp* densegp_create(const double* inputs, int n, int D, const double* targets, unsigned testing_size,
                                    const mogp_meanfunc* mean, int kernel_type, int nugget_type, double nugget_size, bool analytic) {
  try {
    MeanFunc mf;
    if (mean) mf = mean->mf;
    auto* h = new mogp_densegp;
    h->eng = new Engine(inputs, n, D, targets, 1, testing_size, mf, kernel_type, nugget_type, nugget_size, analytic);
    h->idx = 0;
    h->owns = true;
    return h;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
mogp_densegp* mogp_densegp_create(const double* inputs, int n, int D, const double* targets, unsigned testing_size,
                                  const mogp_meanfunc* mean, int kernel_type, int nugget_type, double nugget_size) {
  return densegp_create(inputs, n, D, targets, testing_size, mean, kernel_type, nugget_type, nugget_size, false);
}
mogp_densegp* mogp_densegp_create_analytic_mean(const double* inputs, int n, int D, const double* targets, unsigned testing_size,
                                                const mogp_meanfunc* mean, int kernel_type, int nugget_type, double nugget_size) {
  return densegp_create(inputs, n, D, targets, testing_size, mean, kernel_type, nugget_type, nugget_size, true);
}
int mogp_densegp_set_mean_priors(mogp_densegp* h, int q, const double* b, const double* Binv, const double* Binv_b, double logdetB) {
  GUARD(h->eng->set_mean_priors(h->idx, q, b, Binv, Binv_b, logdetB));
}
int mogp_densegp_n_beta(const mogp_densegp* h) { return h->eng->q; }
int mogp_densegp_get_beta(const mogp_densegp* h, double* out) {
  const auto& b = h->eng->gp[h->idx].beta;
  for (size_t c = 0; c < b.size(); ++c) out[c] = b[c];
  return 0;
}
void mogp_densegp_destroy(mogp_densegp* h) {
  if (!h || !h->owns) return;
  delete h->eng;
  delete h;
}
int mogp_densegp_n(const mogp_densegp* h) { return h->eng->n; }
int mogp_densegp_D(const mogp_densegp* h) { return h->eng->D; }
int mogp_densegp_n_corr(const mogp_densegp* h) { return h->eng->NC; }
int mogp_densegp_n_params(const mogp_densegp* h) { return h->eng->n_data(h->idx); }
int mogp_densegp_n_mean(const mogp_densegp* h) { return h->eng->n_mean(); }
int mogp_densegp_n_data(const mogp_densegp* h) { return h->eng->n_data(h->idx); }
int mogp_densegp_inputs(const mogp_densegp* h, double* out) {
  std::memcpy(out, h->eng->hX.data(), h->eng->hX.size() * sizeof(double));
  return 0;
}
int mogp_densegp_targets(const mogp_densegp* h, double* out) {
  std::memcpy(out, h->eng->hT.data() + (size_t)h->idx * h->eng->n, h->eng->n * sizeof(double));
  return 0;
}
int mogp_densegp_theta_fit_status(const mogp_densegp* h) { return h->eng->gp[h->idx].has_data ? 1 : 0; }
int mogp_densegp_reset_theta_fit_status(mogp_densegp* h) {
  GPState& g = h->eng->gp[h->idx];
  g.has_data = false;
  g.factored = g.linv = g.kinv = false;
  std::fill(g.data.begin(), g.data.end(), 0.);
  std::fill(g.meanp.begin(), g.meanp.end(), 0.);
  return 0;
}
int mogp_densegp_get_theta(const mogp_densegp* h, double* data_out, double* mean_out) {
  const GPState& g = h->eng->gp[h->idx];
  if (data_out) std::memcpy(data_out, g.data.data(), g.data.size() * sizeof(double));
  if (mean_out && !g.meanp.empty()) std::memcpy(mean_out, g.meanp.data(), g.meanp.size() * sizeof(double));
  return 0;
}
int mogp_densegp_create_gppriors(mogp_densegp* h, int n_corr, const int* ct, const double* cp, int covt, const double* covp, int nugt,
                                 const double* nugp) {
  GUARD(set_priors(h->eng, h->idx, n_corr, ct, cp, covt, covp, nugt, nugp));
}
int mogp_densegp_priors_logp(const mogp_densegp* h, const double* th, int len, double* out) {
  GUARD({
    if (len != h->eng->n_data(h->idx)) throw std::runtime_error("Shape of new GPParams object does not match existing one");
    std::vector<double> v(th, th + len);
    *out = h->eng->gp[h->idx].pri.logp(v, h->eng->NC, h->eng->gp[h->idx].nug_type);
  });
}
int mogp_densegp_priors_dlogpdtheta(const mogp_densegp* h, const double* th, int len, double* out) {
  GUARD({
    if (len != h->eng->n_data(h->idx)) throw std::runtime_error("Shape of new GPParams object does not match existing one");
    std::vector<double> v(th, th + len);
    h->eng->gp[h->idx].pri.dlogpdtheta(v, h->eng->NC, h->eng->gp[h->idx].nug_type, out);
  });
}
int mogp_densegp_priors_sample(mogp_densegp* h, double* out) {
  GUARD({
    static std::mt19937_64 r(std::random_device{}());
    const int nm = h->eng->n_mean();
    for (int k = 0; k < nm; ++k) out[k] = 0.;
    h->eng->gp[h->idx].pri.sample(r, h->eng->NC, h->eng->gp[h->idx].nug_type, out + nm);
  });
}
int mogp_densegp_fit(mogp_densegp* h, const double* theta, int len) { GUARD(h->eng->fit_one(h->idx, theta, len)); }
int mogp_densegp_get_logpost(mogp_densegp* h, const double* theta, int len, double* out) {
  GUARD({
    Engine* e = h->eng;
    const int i = h->idx;
    if (len != e->n_theta(i)) throw std::runtime_error("Shape of new GPParams object does not match existing one");
    const GPState& g = e->gp[i];
    bool close = g.has_data && g.factored && !g.logpost_stale;
    if (close) {   // gpparams.hpp:204-210 test_close: ||theta - current|| < 1e-8
      double d2 = 0.;
      const int nm = e->n_mean();
      for (int k = 0; k < nm; ++k) d2 += (theta[k] - g.meanp[k]) * (theta[k] - g.meanp[k]);
      for (size_t k = 0; k < g.data.size(); ++k) d2 += (theta[nm + k] - g.data[k]) * (theta[nm + k] - g.data[k]);
      close = std::sqrt(d2) < 1e-8;
    }
    if (!close) e->fit_one(i, theta, len);
    *out = e->gp[i].logpost;
  });
}
int mogp_densegp_logpost_deriv(mogp_densegp* h, double* out, int len) {
  GUARD({
    Engine* e = h->eng;
    if (len < e->n_theta(h->idx)) throw std::runtime_error("logpost_deriv: the result buffer passed was too small");
    if (!e->gp[h->idx].factored) throw std::runtime_error("logpost_deriv: hyperparameters have not been fit");
    std::vector<int> ids{h->idx};
    e->grad_current(ids, out, len);
  });
}
static void check_batch(const mogp_densegp* h, int m, int D, int out_len, const char* small_msg) {
  if (D != h->eng->D) throw std::runtime_error("testing points must have D columns");
  if (out_len < m) throw std::runtime_error(small_msg);
  if ((unsigned)m > h->eng->testing_size)
    throw std::runtime_error("predict_variance_batch: More test points were passed than the maximum batch size");
}
int mogp_densegp_predict(mogp_densegp* h, const double* testing, int D, double* mean_out) {
  GUARD({
    if (D != h->eng->D) throw std::runtime_error("testing point must have D entries");
    std::vector<int> ids{h->idx};
    h->eng->predict(ids, testing, 1, false, mean_out, nullptr, 1, false, nullptr);
  });
}
int mogp_densegp_predict_variance(mogp_densegp* h, const double* testing, int D, double* mean_out, double* var_out) {
  GUARD({
    if (D != h->eng->D) throw std::runtime_error("testing point must have D entries");
    std::vector<int> ids{h->idx};
    h->eng->predict(ids, testing, 1, false, mean_out, var_out, 1, false, nullptr);
  });
}
int mogp_densegp_predict_batch(mogp_densegp* h, const double* testing, int m, int D, double* mean_out, int out_len) {
  GUARD({
    check_batch(h, m, D, out_len, "predict_batch: the result buffer passed was too small to hold the result");
    std::vector<int> ids{h->idx};
    h->eng->predict(ids, testing, m, false, mean_out, nullptr, m, false, nullptr);
  });
}
int mogp_densegp_predict_variance_batch(mogp_densegp* h, const double* testing, int m, int D, double* mean_out, double* var_out, int out_len) {
  GUARD({
    check_batch(h, m, D, out_len, "predict_variance_batch: The result buffer passed was too small to hold the variance");
    std::vector<int> ids{h->idx};
    h->eng->predict(ids, testing, m, false, mean_out, var_out, m, false, nullptr);
  });
}
int mogp_densegp_predict_deriv(mogp_densegp* h, const double* testing, int m, int D, double* out, int out_rows, int out_cols) {
  GUARD({
    if (out_rows < m || out_cols != h->eng->D)
      throw std::runtime_error("predict_deriv: the result buffer passed was the wrong shape to hold the result");
    check_batch(h, m, D, m, "");
    std::vector<int> ids{h->idx};
    h->eng->predict(ids, testing, m, false, nullptr, nullptr, m, false, out);        // derivatives only: no cross covariance
  });
}
int mogp_densegp_predict_full_cov(mogp_densegp* h, const double* testing, int m, int D, double* mean_out, double* cov_out) {
  GUARD({
    if (D != h->eng->D) throw std::runtime_error("testing points must have D columns");
    std::vector<int> ids{h->idx};
    h->eng->predict_full_cov(ids, testing, m, mean_out, cov_out);
  });
}
int mogp_densegp_implausibility(mogp_densegp* h, const double* testing, int m, int D, double obs, double obs_var, double discrepancy,
                                int include_nugget, double* out) {
  GUARD({
    if (D != h->eng->D) throw std::runtime_error("testing points must have D columns");
    std::vector<int> ids{h->idx};
    h->eng->implausibility(ids, testing, m, &obs, &obs_var, &discrepancy, include_nugget != 0, 0, out);
  });
}
int mogp_densegp_loo_variance(mogp_densegp* h, double* out) { GUARD(h->eng->loo_variance(h->idx, out)); }
int mogp_densegp_get_K(mogp_densegp* h, double* out) { GUARD(h->eng->get_K(h->idx, out)); }
int mogp_densegp_get_invQ(mogp_densegp* h, double* out) { GUARD(h->eng->get_invQ(h->idx, out)); }
int mogp_densegp_get_invQt(mogp_densegp* h, double* out) { GUARD(h->eng->get_invQt(h->idx, out)); }
int mogp_densegp_get_cholesky_lower(mogp_densegp* h, double* out) { GUARD(h->eng->get_chol(h->idx, out)); }
int mogp_densegp_get_pivot(mogp_densegp* h, int* P_out, int* rank_out) { GUARD(h->eng->get_pivot(h->idx, P_out, rank_out)); }
int mogp_pivot_cholesky(const double* A, int n, double* L_out, int* P_out, int* rank_out) {
  GUARD(Engine::pivot_cholesky(A, n, L_out, P_out, rank_out));
}
double mogp_densegp_get_nugget_size(const mogp_densegp* h) { return h->eng->nugget_size(h->idx); }
int mogp_densegp_set_nugget_size(mogp_densegp* h, double v) {
  GPState& g = h->eng->gp[h->idx];
  // a fixed nugget is part of the factored matrix: a new value invalidates the factor, alpha and the log-posterior
  // (the reference keeps serving the stale ones, densegp_gpu.hpp:125-135); the emulator has to be fit again
  if (g.nug_type == NUG_FIXED && v != g.nug_size) {
    g.has_data = false;
    g.factored = g.linv = g.kinv = false;
  }
  g.nug_size = v;
  if (g.nug_type == NUG_FIT && !g.data.empty()) g.data[g.data.size() - 1] = v;   // gpparams.hpp:167-172
  return 0;
}
int mogp_densegp_get_nugget_type(const mogp_densegp* h) { return h->eng->gp[h->idx].nug_type; }
int mogp_densegp_set_nugget_type(mogp_densegp* h, int t) {
  GUARD({
    if (t < 0 || t > 3) throw std::runtime_error("Unrecognized nugget_type");
    GPState& g = h->eng->gp[h->idx];
    if (t != g.nug_type) {
      g.nug_type = t;
      g.data.assign(h->eng->NC + 1 + (t == NUG_FIT ? 1 : 0), 0.);
      g.has_data = false;
      g.factored = g.linv = g.kinv = false;
    }
  });
}
int mogp_densegp_get_kernel_type(const mogp_densegp* h) { return h->eng->kernel_type; }
int mogp_fit_single_GP_MAP(mogp_densegp* h, int n_tries, const double* theta0, int theta0_len) {
  GUARD({
    std::vector<int> ids{h->idx};
    h->eng->fit_map(ids, n_tries, theta0, theta0_len);
  });
}

// ---- MultiOutputGP_GPU ---------------------------------------------------------------------------
static mogp_mogp* mogp_create(const double* inputs, int n, int D, const double* targets, int n_out, unsigned testing_size,
                              const mogp_meanfunc* mean, int kernel_type, int nugget_type, double nugget_size, bool analytic);
mogp_mogp* mogp_mogp_create(const double* inputs, int n, int D, const double* targets, int n_out, unsigned testing_size,
                            const mogp_meanfunc* mean, int kernel_type, int nugget_type, double nugget_size) {
  return mogp_create(inputs, n, D, targets, n_out, testing_size, mean, kernel_type, nugget_type, nugget_size, false);
}
mogp_mogp* mogp_mogp_create_analytic_mean(const double* inputs, int n, int D, const double* targets, int n_out, unsigned testing_size,
                                          const mogp_meanfunc* mean, int kernel_type, int nugget_type, double nugget_size) {
  return mogp_create(inputs, n, D, targets, n_out, testing_size, mean, kernel_type, nugget_type, nugget_size, true);
}
static mogp_mogp* mogp_create(const double* inputs, int n, int D, const double* targets, int n_out, unsigned testing_size,
                              const mogp_meanfunc* mean, int kernel_type, int nugget_type, double nugget_size, bool analytic) {
  try {
    MeanFunc mf;
    if (mean) mf = mean->mf;
    auto* h = new mogp_mogp;
    h->eng.reset(new Engine(inputs, n, D, targets, n_out, testing_size, mf, kernel_type, nugget_type, nugget_size, analytic));
    h->views.resize(n_out);
    for (int i = 0; i < n_out; ++i) h->views[i] = mogp_densegp{h->eng.get(), i, false};
    h->nug_size0 = nugget_size;
    h->nug_type0 = nugget_type;
    return h;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void mogp_mogp_destroy(mogp_mogp* h) { delete h; }
int mogp_mogp_n(const mogp_mogp* h) { return h->eng->n; }
int mogp_mogp_D(const mogp_mogp* h) { return h->eng->D; }
int mogp_mogp_n_emulators(const mogp_mogp* h) { return h->eng->B; }
int mogp_mogp_inputs(const mogp_mogp* h, double* out) {
  std::memcpy(out, h->eng->hX.data(), h->eng->hX.size() * sizeof(double));
  return 0;
}
int mogp_mogp_targets(const mogp_mogp* h, double* out) {
  std::memcpy(out, h->eng->hT.data(), h->eng->hT.size() * sizeof(double));
  return 0;
}
mogp_densegp* mogp_mogp_emulator(mogp_mogp* h, int index) {
  if (index < 0 || index >= h->eng->B) {
    g_err = "Invalid emulator index";
    return nullptr;
  }
  return &h->views[index];
}
int mogp_mogp_get_nugget_type(const mogp_mogp* h) { return h->nug_type0; }
double mogp_mogp_get_nugget_size(const mogp_mogp* h) { return h->nug_size0; }
int mogp_mogp_get_fitted_indices(const mogp_mogp* h, int* out) {
  int c = 0;
  for (int i = 0; i < h->eng->B; ++i)
    if (h->eng->gp[i].has_data) out[c++] = i;
  return c;
}
int mogp_mogp_get_unfitted_indices(const mogp_mogp* h, int* out) {
  int c = 0;
  for (int i = 0; i < h->eng->B; ++i)
    if (!h->eng->gp[i].has_data) out[c++] = i;
  return c;
}
int mogp_mogp_reset_fit_status(mogp_mogp* h) {
  for (int i = 0; i < h->eng->B; ++i) mogp_densegp_reset_theta_fit_status(&h->views[i]);
  return 0;
}
int mogp_mogp_create_priors_for_emulator(mogp_mogp* h, int index, int n_corr, const int* ct, const double* cp, int covt, const double* covp,
                                         int nugt, const double* nugp) {
  GUARD({
    if (index < 0 || index >= h->eng->B) throw std::runtime_error("Invalid emulator index for setting priors");
    set_priors(h->eng.get(), index, n_corr, ct, cp, covt, covp, nugt, nugp);
  });
}
int mogp_mogp_eval(mogp_mogp* h, const double* thetas, int n_rows, int n_cols, double* logpost_out, double* grad_out, int* ok_out) {
  GUARD({
    Engine* e = h->eng.get();
    if (n_rows != e->B) throw std::runtime_error("thetas must have one row per emulator");
    std::vector<int> ids(e->B);
    std::vector<const double*> th(e->B);
    for (int i = 0; i < e->B; ++i) {
      if (n_cols != e->n_theta(i)) throw std::runtime_error("Shape of new GPParams object does not match existing one");
      ids[i] = i;
      th[i] = thetas + (size_t)i * n_cols;
    }
    std::vector<double> f(e->B);
    std::vector<int> ok(e->B);
    e->eval(ids, th, grad_out != nullptr, f.data(), grad_out, n_cols, ok.data());
    if (logpost_out) std::memcpy(logpost_out, f.data(), sizeof(double) * e->B);
    if (ok_out) std::memcpy(ok_out, ok.data(), sizeof(int) * e->B);
  });
}
int mogp_mogp_fit(mogp_mogp* h, const double* thetas, int n_rows, int n_cols) {
  GUARD({
    Engine* e = h->eng.get();
    std::vector<int> ok(e->B);
    if (mogp_mogp_eval(h, thetas, n_rows, n_cols, nullptr, nullptr, ok.data())) throw std::runtime_error(g_err);
    for (int i = 0; i < e->B; ++i)
      if (!ok[i] && !(e->gp[i].nug_type == NUG_PIVOT && e->gp[i].factored)) {
        if (e->gp[i].nug_type == NUG_ADAPTIVE) throw std::runtime_error("All attempts at factorization failed. Last return code 1");
        throw std::runtime_error("Unable to factorize matrix using selected nugget type");
      }
  });
}
int mogp_mogp_fit_emulator(mogp_mogp* h, int index, const double* theta, int len) {
  GUARD({
    if (index < 0 || index >= h->eng->B) throw std::runtime_error("Invalid emulator index");
    h->eng->fit_one(index, theta, len);
  });
}
static std::vector<int> fitted_ids(const mogp_mogp* h) {
  std::vector<int> ids;
  for (int i = 0; i < h->eng->B; ++i)
    if (h->eng->gp[i].has_data && h->eng->gp[i].factored) ids.push_back(i);
  return ids;
}
// results of fitted emulators go to their own rows; rows of unfitted emulators are untouched
static void mogp_predict_common(mogp_mogp* h, const double* testing, int m, int D, double* means, double* vars, double* derivs) {
  Engine* e = h->eng.get();
  if (D != e->D) throw std::runtime_error("testing points must have D columns");
  std::vector<int> ids = fitted_ids(h);
  if (ids.empty()) return;
  const size_t nf = ids.size();
  if ((int)nf == e->B) {             // every emulator fitted: results go straight into the caller's arrays (means == null: derivatives only)
    e->predict(ids, testing, m, false, means, vars, m, false, derivs);
    return;
  }
  std::vector<double> mm(means ? nf * m : 0), vv(vars ? nf * m : 0), dd(derivs ? nf * m * D : 0);
  e->predict(ids, testing, m, false, means ? mm.data() : nullptr, vars ? vv.data() : nullptr, m, false, derivs ? dd.data() : nullptr);
  for (size_t k = 0; k < nf; ++k) {
    if (means) std::memcpy(means + (size_t)ids[k] * m, mm.data() + k * m, m * sizeof(double));
    if (vars) std::memcpy(vars + (size_t)ids[k] * m, vv.data() + k * m, m * sizeof(double));
    if (derivs) std::memcpy(derivs + (size_t)ids[k] * m * D, dd.data() + k * m * D, (size_t)m * D * sizeof(double));
  }
}
int mogp_mogp_predict_batch(mogp_mogp* h, const double* testing, int m, int D, double* means) {
  GUARD(mogp_predict_common(h, testing, m, D, means, nullptr, nullptr));
}
int mogp_mogp_predict_variance_batch(mogp_mogp* h, const double* testing, int m, int D, double* means, double* vars) {
  GUARD(mogp_predict_common(h, testing, m, D, means, vars, nullptr));
}
int mogp_mogp_predict_deriv(mogp_mogp* h, const double* testing, int m, int D, double* derivs) {
  GUARD(mogp_predict_common(h, testing, m, D, nullptr, nullptr, derivs));
}
int mogp_mogp_predict_full_cov(mogp_mogp* h, const double* testing, int m, int D, double* means, double* covs) {
  GUARD({
    Engine* e = h->eng.get();
    if (D != e->D) throw std::runtime_error("testing points must have D columns");
    std::vector<int> ids = fitted_ids(h);
    if (ids.empty()) return 0;
    if ((int)ids.size() == e->B) {
      e->predict_full_cov(ids, testing, m, means, covs);
    } else {
      const size_t nf = ids.size();
      const size_t mm = (size_t)m * m;
      std::vector<double> mu(nf * m);
      std::vector<double> cc(nf * mm);
      e->predict_full_cov(ids, testing, m, mu.data(), cc.data());
      for (size_t k = 0; k < nf; ++k) {
        std::memcpy(means + (size_t)ids[k] * m, mu.data() + k * m, m * sizeof(double));
        std::memcpy(covs + (size_t)ids[k] * mm, cc.data() + k * mm, mm * sizeof(double));
      }
    }
  });
}
int mogp_mogp_implausibility(mogp_mogp* h, const double* testing, int m, int D, const double* obs, const double* obs_var,
                             const double* discrepancy, int include_nugget, int rank, double* out) {
  GUARD({
    Engine* e = h->eng.get();
    if (D != e->D) throw std::runtime_error("testing points must have D columns");
    std::vector<int> ids = fitted_ids(h);
    if ((int)ids.size() != e->B) throw std::runtime_error("Hyperparameters have not been fit for this Gaussian Process");
    e->implausibility(ids, testing, m, obs, obs_var, discrepancy, include_nugget != 0, rank, out);
  });
}
// device-resident prediction: d_means / d_vars (n_emulators, m) and d_derivs (n_emulators, m, D) are device buffers (d_vars, d_derivs may be
// null); rows of emulators that are not fit are filled with NaN (MultiOutputGP_GPU.py:288-296)
#define HIPCK(x) hip_check((x), #x)
static void mogp_predict_dev_common(mogp_mogp* h, const double* d_testing, int m, int D, double* d_means, double* d_vars, double* d_derivs) {
  Engine* e = h->eng.get();
  if (D != e->D) throw std::runtime_error("testing points must have D columns");
  if (m <= 0) return;
  // the same contract whatever the fit status is: the test points are required, any of the three outputs may be null
  // (Engine::predict: means == nullptr = derivatives only), but not all of them
  if (!d_testing) throw std::runtime_error("device-resident predict: the test points pointer is null");
  if (!d_means && !d_vars && !d_derivs) throw std::runtime_error("device-resident predict: no output buffer given");
  std::vector<int> ids = fitted_ids(h);
  if ((int)ids.size() == e->B) {
    e->predict(ids, d_testing, m, true, d_means, d_vars, m, true, d_derivs);
    return;
  }
  // some emulators are not fit: the fitted ones are predicted into scratch rows and copied to their places, the others become NaN (the
  // all-ones bit pattern is a quiet NaN: a memset on the engine's stream instead of one blocking host copy per row and array)
  const size_t nf = ids.size(), row = (size_t)m, drow = (size_t)m * D;
  hipStream_t st = e->stream;
  std::vector<char> fitted(e->B, 0);
  for (int i : ids) fitted[i] = 1;
  for (int i = 0; i < e->B; ++i) {
    if (fitted[i]) continue;
    if (d_means) HIPCK(hipMemsetAsync(d_means + (size_t)i * row, 0xFF, row * sizeof(double), st));
    if (d_vars) HIPCK(hipMemsetAsync(d_vars + (size_t)i * row, 0xFF, row * sizeof(double), st));
    if (d_derivs) HIPCK(hipMemsetAsync(d_derivs + (size_t)i * drow, 0xFF, drow * sizeof(double), st));
  }
  if (nf == 0) {
    HIPCK(hipStreamSynchronize(st));
    return;
  }
  double *tm = nullptr, *tv = nullptr, *td = nullptr;
  auto release = [&] {
    for (double* p : {tm, tv, td})
      if (p) hipFree(p);
  };
  try {
    if (d_means) HIPCK(hipMalloc((void**)&tm, nf * row * sizeof(double)));
    if (d_vars) HIPCK(hipMalloc((void**)&tv, nf * row * sizeof(double)));
    if (d_derivs) HIPCK(hipMalloc((void**)&td, nf * drow * sizeof(double)));
    e->predict(ids, d_testing, m, true, tm, tv, m, true, td);
    // runs of consecutive fitted emulators go in one copy each
    for (size_t k = 0; k < nf;) {
      size_t len = 1;
      while (k + len < nf && ids[k + len] == ids[k] + (int)len) ++len;
      if (d_means) HIPCK(hipMemcpyAsync(d_means + (size_t)ids[k] * row, tm + k * row, len * row * sizeof(double), hipMemcpyDeviceToDevice, st));
      if (d_vars) HIPCK(hipMemcpyAsync(d_vars + (size_t)ids[k] * row, tv + k * row, len * row * sizeof(double), hipMemcpyDeviceToDevice, st));
      if (d_derivs) HIPCK(hipMemcpyAsync(d_derivs + (size_t)ids[k] * drow, td + k * drow, len * drow * sizeof(double), hipMemcpyDeviceToDevice, st));
      k += len;
    }
    HIPCK(hipStreamSynchronize(st));
  } catch (...) {
    hipStreamSynchronize(st);
    release();
    throw;
  }
  release();
}
#undef HIPCK
int mogp_mogp_predict_variance_batch_dev(mogp_mogp* h, const double* d_testing, int m, int D, double* d_means, double* d_vars) {
  GUARD(mogp_predict_dev_common(h, d_testing, m, D, d_means, d_vars, nullptr));
}
int mogp_mogp_predict_dev(mogp_mogp* h, const double* d_testing, int m, int D, double* d_means, double* d_vars, double* d_derivs) {
  GUARD(mogp_predict_dev_common(h, d_testing, m, D, d_means, d_vars, d_derivs));
}
int mogp_fit_GP_MAP(mogp_mogp* h, int n_tries, const double* theta0, int theta0_len) {
  GUARD({
    std::vector<int> ids(h->eng->B);
    for (int i = 0; i < h->eng->B; ++i) ids[i] = i;
    h->eng->fit_map(ids, n_tries, theta0, theta0_len);
  });
}
int mogp_set_fit_options(int max_iter, double ftol, double gtol, unsigned long long seed) {
  FitOptions& o = fit_options();
  if (max_iter > 0) o.max_iter = max_iter;
  if (ftol > 0) o.ftol = ftol;
  if (gtol > 0) o.gtol = gtol;
  o.seed = seed;
  return 0;
}

// ---- stand-alone kernel objects (bindings.cu:340-361, kernel.hpp:47-107) --------------------------------------
// kernel_type as in the enum (0 SqExp, 1 Matern52, 2 ProductMat52, 3 UniformSqExp, 4 UniformMat52); params = [corr_raw.., log sigma^2]
// (n_corr + 1 entries, n_corr = 1 for the uniform kernels); what = 0 kernel_f -> out (n1, n2), 1 kernel_deriv -> out
// (n_corr + 1, n1, n2), 2 kernel_inputderiv -> out (n2, n1, D) (the reference's flat order)
int mogp_kernel_eval(int kernel_type, int what, const double* x1, int n1, const double* x2, int n2, int D, const double* params, int n_params,
                     double* out) {
  GUARD(kernel_eval_impl(kernel_type, what, x1, n1, x2, n2, D, params, n_params, out));
}

// ---- measurement hooks ----------------------------------------------------------------------------
int mogp_profile_enable(int on) { prof_enable(on != 0); return 0; }
int mogp_profile_reset(void) { prof_reset(); return 0; }
int mogp_profile_schedule(int schedule, int single_stream) {
  schedule_override().schedule = schedule;
  schedule_override().single_stream = single_stream != 0;
  return 0;
}
static int task_table_out(int n_plus_rhs, bool ahead, int* out, int capacity) {
  // host-only: the per-emulator task order of the one-launch Cholesky for a matrix of NP = roundup(n_plus_rhs, 128) rows
  const int NP = (n_plus_rhs + TILE - 1) / TILE * TILE;
  const std::vector<int> tb = mchol_task_table(NP, ahead);
  if (out)
    for (int i = 0; i < (int)tb.size() && i < capacity; ++i) out[i] = tb[i];
  return (int)tb.size();
}
int mogp_mchol_task_table(int n_plus_rhs, int* out, int capacity) { return task_table_out(n_plus_rhs, false, out, capacity); }
int mogp_mchol_task_table_ahead(int n_plus_rhs, int* out, int capacity) { return task_table_out(n_plus_rhs, true, out, capacity); }
int mogp_profile_counter(const char* name, long long* out) {
  const long long v = prof_counter(name);
  if (v < 0 || !out) {
    g_err = std::string("unknown counter ") + (name ? name : "(null)");
    return 1;
  }
  *out = v;
  return 0;
}
int mogp_profile_get(const char* tag, double* total_ms, long long* launches, double* alg_flops, double* alg_bytes) {
  return prof_get(tag, total_ms, launches, alg_flops, alg_bytes) ? 0 : 1;
}
void* mogp_dev_malloc(unsigned long long bytes) {
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) { g_err = "hipMalloc failed"; return nullptr; }
  return p;
}
int mogp_dev_free(void* p) { GUARD(hip_check(hipFree(p), "hipFree")); }
int mogp_dev_upload(void* d, const void* s, unsigned long long bytes) { GUARD(hip_check(hipMemcpy(d, s, bytes, hipMemcpyHostToDevice), "upload")); }
int mogp_dev_download(void* d, const void* s, unsigned long long bytes) { GUARD(hip_check(hipMemcpy(d, s, bytes, hipMemcpyDeviceToHost), "download")); }
int mogp_dev_synchronize(void) { GUARD(hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize")); }

}  // extern "C"
