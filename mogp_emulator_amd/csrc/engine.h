// Engine: B independent zero/parametric-mean GPs that share one input matrix X, resident on one
// MI355X.  It is the device-side state behind both DenseGP_GPU (B = 1, densegp_gpu.hpp:36-123) and
// MultiOutputGP_GPU (B = n_emulators, multioutputgp_gpu.hpp:35-287).  Where the reference loops
// over emulators with OpenMP and serialises them on the default stream, every operation here is
// ONE batched launch sequence over an index list of emulators.
#pragma once
#include <hip/hip_runtime.h>
#include <functional>
#include <random>
#include <string>
#include <map>
#include <vector>

#include "hostmath.h"
#include "launch.h"

namespace mogp {

struct GPState {
  std::vector<double> data;      // n_data: corr_raw (NC), log sigma^2, [log nugget]
  std::vector<double> meanp;     // n_mean
  bool has_data = false;
  int nug_type = NUG_ADAPTIVE;
  double nug_size = 0.;          // adaptive: jitter found by the last fit; fixed: the constant
  Priors pri;
  double logpost = 0.;
  bool logpost_stale = false;    // the priors changed since `logpost` was computed (the factorisation itself is still valid)
  bool factored = false;         // A holds L (and y) for `data`
  bool linv = false, kinv = false;
  double nugget_used = 0.;       // value actually added to the diagonal in the last factorisation
  // nugget="pivot": the factor in A, alpha, L^-1, K^-1 and this emulator's copy of the inputs are in pivoted order
  bool permuted = false;
  bool kinv_split = false;       // rank < n: Kinv was formed without the rows of L^-1 of the skipped pivots (kept in w2)
  int rank = 0;                  // pivots accepted by the last pivoted factorisation (n = full rank)
  std::vector<double> beta;      // analytic mean coefficients (q), GaussianProcess.py:669-670
  std::vector<double> LA;        // q x q lower Cholesky factor of A = H^T K^-1 H + B^-1
  // informative mean priors beta ~ N(b, B) of the analytic mean (Priors.py:423-581); empty = weak
  std::vector<double> mp_b, mp_Binv, mp_Binvb;
  double mp_logdetB = 0.;
};

class Engine {
 public:
  // analytic_mean: the coefficients of a const / polynomial mean are integrated out analytically (CPU
  // GaussianProcess semantics) instead of living in theta (reference GPU semantics)
  Engine(const double* X, int n, int D, const double* targets, int B, unsigned testing_size, const MeanFunc& mean,
         int kernel_type, int nug_type, double nug_size, bool analytic_mean = false);
  ~Engine();
  Engine(const Engine&) = delete;

  int n, D, NP, LD, PS, B, kernel_type;
  // kernel_type: 0 SquaredExponential, 1 Matern52 (reference enum, types.hpp:29-35) and the CPU-only kernels of
  // Kernel.py:946-997: 2 ProductMat52, 3 UniformSqExp, 4 UniformMat52.  NC = number of correlation parameters
  // (1 for the uniform kernels, which run the device kernels 0 / 1 with one shared length scale).
  int NC = 0;
  bool uniform() const { return kernel_type == 3 || kernel_type == 4; }
  int device_kernel() const { return kernel_type == 3 ? 0 : (kernel_type == 4 ? 1 : kernel_type); }
  size_t MS;
  unsigned testing_size;
  MeanFunc mean;
  std::vector<GPState> gp;
  std::vector<double> hX, hT;    // host copies (inputs()/targets())

  bool analytic = false;
  int q = 0, R = 1, RA = 1;      // analytic mean columns, right-hand-side rows (1 + q), rows of alpha per emulator
  // MeanPriors(mean = b, cov = B) of emulator i: b (q), B^-1 (q x q), B^-1 b (q), log|B|; q_in = 0 resets to weak
  void set_mean_priors(int i, int q_in, const double* b, const double* Binv, const double* Binvb, double logdetB);
  int n_mean() const { return analytic ? 0 : mean.n_params(); }
  int n_data(int i) const { return NC + 1 + (gp[i].nug_type == NUG_FIT ? 1 : 0); }
  int n_theta(int i) const { return n_mean() + n_data(i); }
  double nugget_size(int i) const;

  // Batched objective (+ gradient) at per-emulator thetas (full vectors [mean | data]).
  // ok[k] = 1 when the factorisation succeeded.  Never throws for numerical failure.
  void eval(const std::vector<int>& ids, const std::vector<const double*>& thetas, bool want_grad, double* f, double* grad,
            int grad_ld, int* ok);
  // fit(theta) for one emulator: throws std::runtime_error on failure (densegp_gpu.hpp:556-570)
  void fit_one(int i, const double* theta, int len);
  void grad_current(const std::vector<int>& ids, double* grad, int grad_ld);

  // predictions for emulators `ids` (must be fitted). Xs host (m, D) unless xs_on_device.
  // means/vars: (ids.size(), m) row-major with leading dimension out_ld, derivs (ids.size(), m, D) or null; host unless
  // out_on_device.  Mean-function terms (parametric or analytic) are added on the device either way.
  void predict(const std::vector<int>& ids, const double* Xs, int m, bool xs_on_device, double* means, double* vars,
               long out_ld, bool out_on_device, double* derivs);

  void get_K(int i, double* out);
  // HistoryMatching.get_implausibility (HistoryMatching.py:197-276) fused behind the batched prediction:
  // obs / obs_var / discrepancy per entry of ids; out (m) host.  Query points are processed in device chunks.
  void implausibility(const std::vector<int>& ids, const double* Xs, int m, const double* obs, const double* obs_var,
                      const double* discrepancy, bool include_nugget, int rank, double* out);
  // leave-one-out predictive variance of emulator i at its own training inputs (MICEFastGP.fast_predict for every index)
  void loo_variance(int i, double* out);
  // predict(full_cov=True), GaussianProcess.py:899-911: means (nb, m), covs (nb, m, m) host buffers, nugget NOT included
  void predict_full_cov(const std::vector<int>& ids, const double* Xs, int m, double* means, double* covs);
  void get_invQ(int i, double* out);
  void get_invQt(int i, double* out);
  void get_chol(int i, double* out);
  // pivot order P of the last fit (A[P][:, P] = L L^T, ChoInvPivot.P); identity for the other nugget types
  void get_pivot(int i, int* perm_out, int* rank_out);
  // pivot_cholesky(A) of linalg/cholesky.py:284-327 for an arbitrary symmetric matrix (host buffers, row-major)
  static void pivot_cholesky(const double* A, int n, double* L_out, int* P_out, int* rank_out);

  // multi-start MAP fit of emulators `ids` (fitting.hpp:61-128): all (emulator, start) runs through one slot pool
  void fit_map(const std::vector<int>& ids, int n_tries, const double* theta0, int theta0_len);
  // the optimiser runs as a slot pool on emulators `slots` of this engine: next(pos, x0, tag) hands slot `pos` its next run (false: none
  // left), done(tag, f, x) receives a run's end point (f = +inf, x empty: failed)
  void run_pool(const std::vector<int>& slots, const std::function<bool(int, std::vector<double>&, int&)>& next,
                const std::function<void(int, double, const std::vector<double>&)>& done);
  // slot `slot` of this (replica) engine takes targets, nugget type and priors of emulator i of `src`
  void retarget(int slot, const Engine& src, int i);

  hipStream_t stream = nullptr;      // main stream: covariance build, trailing updates, everything else
  hipStream_t pstream = nullptr;     // look-ahead stream: panel factorisations
  std::vector<hipEvent_t> evPanel, evUpd;
  std::vector<hipStream_t> gstreams;  // extra streams for independent emulator groups
  hipEvent_t evReady = nullptr;
  hipEvent_t evGroup[15] = {};

 private:
  void upload_params(const std::vector<int>& ids);
  void upload_idx(const std::vector<int>& ids);
  // defer_info: leave the status words on the device (the caller reads them together with its own results: one
  // synchronisation per evaluation instead of two)
  void factorize(const std::vector<int>& ids, std::vector<int>& info, bool defer_info = false);
  void factorize_blocked(const std::vector<int>& ids, std::vector<int>& info, bool defer_info);
  void read_info(std::vector<int>& info, bool defer_info);
  std::vector<int> idx_on_device;          // what dIdx holds (upload_idx skips an identical list)
  void factorize_pivot(const std::vector<int>& ids, std::vector<int>& info);
  void ensure_pivot_buffers();
  void unpermute(int i, double* vec) const;          // vec (n) from pivoted to training order, in place
  void panel(const BatchView& v, int o, int w, hipStream_t st);
  void ensure_linv(const std::vector<int>& ids);
  void ensure_kinv(const std::vector<int>& ids, bool for_gradient = false);
  BatchView view(int nb) const;
  void build_cov(const BatchView& v, const ZeroRanges& zero = ZeroRanges());
  std::vector<char> z_armed;     // per emulator: its solution row holds the sentinel pattern of the one-launch back substitution
  void set_theta(int i, const double* theta);
  void ensure_predict_scratch(int nb, int MC);

  double *dX = nullptr, *dP = nullptr, *dT = nullptr, *dA = nullptr, *dLinv = nullptr, *dKinv = nullptr, *dAlpha = nullptr;
  uint32_t* sigU1 = nullptr;     // signal word of the stream memory operations of the look-ahead schedule (hipMallocSignalMemory)
  uint32_t sig_epoch = 1;
  bool can_waitval = false;      // hipDeviceAttributeCanUseStreamWaitValue of the engine's device
  int device = 0;                // HIP device the engine was created on
  // one-launch Cholesky (kernels_mchol.hip): task table, control words, per-column packs
  int* dMcTable = nullptr;       // the task order of one emulator (mchol_task_table)
  int mc_ntasks = 0;
  unsigned* dMcCtrl = nullptr;
  size_t mc_ctrl_ints = 0;
  int mc_slots = 0;              // batch slots dMcCtrl / dMcPacks are sized for (grown to the largest one-launch batch seen)
  double* dMcPacks = nullptr;
  bool mc_used = false;          // the last factorisation of this engine went through the one-launch kernel (its abort word is live)
  bool mc_force_legacy = false;  // transient: repeat a factorisation with a multi-launch schedule after an abort
  int n_cu = 256;
  int* dBsFlags = nullptr;       // hand-off flags of the one-launch back substitution (B x ceil(n/128)), compared with bs_epoch
  int bs_epoch = 0;
  double *dRes = nullptr, *hRes = nullptr;   // per emulator [log-det, status, Gram matrix]: device buffer and its pinned host mirror
  double *dGradOut = nullptr, *dGradPartial = nullptr;
  int *dInfo = nullptr, *dIdx = nullptr;
  double* dLpack = nullptr;
  double *dH = nullptr, *dZ = nullptr, *dM = nullptr;
  // nugget="pivot": per-emulator inputs in pivot order (B*n*D), pivot order (B*n), rank (B), scratch (B*2*NP)
  double *dXp = nullptr, *dPivWork = nullptr;
  int *dPerm = nullptr, *dRank = nullptr;
  std::vector<int> hPerm;
  std::map<int, double*> w2;     // emulator -> (n - rank) x LD rows of L^-1 of the skipped pivots (gradient path)
  void drop_w2(int i);
  std::vector<double> hH;        // q x n design-matrix columns      // packed transposed diagonal block + reciprocal diagonal (potf2 -> trsm)
  double* hP = nullptr;          // pinned host copy of the parameter blocks (B * PS doubles)
  // predict scratch
  double *dXs = nullptr, *dKs = nullptr, *dMean = nullptr, *dVar = nullptr, *dVarPartial = nullptr, *dDeriv = nullptr;
  size_t capXs = 0, capKs = 0, capMean = 0, capVar = 0, capVarPartial = 0, capDeriv = 0;
  double *dMeanFin = nullptr, *dMeanAux = nullptr;   // finished means when the dot products have their own rows; staging of the mean-function terms
  size_t capMeanFin = 0, capMeanAux = 0;
  std::mt19937_64 rng;
};

// optimiser options (mogp_set_fit_options)
struct FitOptions {
  int max_iter = 200;
  double ftol = 1e-9;   // |f_k - f_{k+1}| <= ftol * max(1, |f|)   (dlib objective_delta_stop_strategy(1e-9), fitting.hpp:92)
  double gtol = 1e-6;
  unsigned long long seed = 0;
};
FitOptions& fit_options();

// measurement hooks (mogp_profile_schedule): force the Cholesky schedule / serialise it onto one stream so that the
// HIP-event time of a kernel is its time alone on the device; -1 / false = the library's own choice.
// FOR MEASUREMENT ONLY: process-global, read by every engine at the start of a factorisation and not synchronised --
// set it while no evaluation is in flight (bench.py does), never from a thread that races with one.
struct ScheduleOverride {
  int schedule = -1;       // 0 two emulator groups, 1 right-looking, 3 look-ahead, 4 one launch (task queue), 5 the multi-launch schedule of the regime
  bool single_stream = false;
};
ScheduleOverride& schedule_override();

void hip_check(hipError_t e, const char* what);
void prof_enable(bool on);
void prof_reset();
bool prof_get(const char* tag, double* ms, long long* launches, double* flops, double* bytes);
// process-wide diagnostic counters (mogp_profile_counter): "backsolve_timeouts" = back substitutions repeated with the
// multi-launch path after a wait of the one-launch chain timed out
long long prof_counter(const char* name);

}  // namespace mogp
