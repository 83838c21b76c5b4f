/*
 * mogp_hip.h -- C ABI of libmogp_hip.so: an MI355X (gfx950) native GP fit + predict
 * backend that drops in behind mogp_emulator's GPU seam.
 *
 * The seam it replaces is the pybind11 module `libgpgpu`
 * (reference: mogp_gpu/src/bindings.cu:13-621, imported by mogp_emulator/LibGPGPU.py:5-14).
 * Every entry point below names the reference binding (file:line) whose behaviour it
 * provides.  The signatures are plain C: pointers + sizes, no torch / numpy / Eigen types.
 *
 * Conventions
 *  - all arrays are IEEE double, C-contiguous (row major); outputs are caller-allocated
 *    buffers filled in place (Eigen::Ref semantics of types.hpp:15-18).
 *  - `theta` passed to fit = [mean params..., corr_raw (D), log sigma^2, (log nugget)],
 *    length n_mean + n_data (densegp_gpu.hpp:493-503).
 *  - every call returns 0 on success, non-zero on failure; mogp_last_error() returns the
 *    message of the last failure on the calling thread.  The Python shim turns that into
 *    RuntimeError(msg) exactly where the reference throws std::runtime_error.
 *  - calls on one handle are not thread safe; distinct handles may be used concurrently.
 *  - pointers whose name starts with `d_` are DEVICE pointers (HBM resident); all others
 *    are host pointers.
 */
#ifndef MOGP_HIP_H
#define MOGP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* enums: values are those of mogp_gpu/src/types.hpp:29-35 (bindings.cu:585-598) */
/* 2-4: kernels the reference only has on the CPU (Kernel.py:946-997); the uniform kernels have one correlation parameter */
enum mogp_kernel_type { MOGP_SQUARED_EXPONENTIAL = 0, MOGP_MATERN52 = 1, MOGP_PRODUCT_MATERN52 = 2, MOGP_UNIFORM_SQUARED_EXPONENTIAL = 3,
                        MOGP_UNIFORM_MATERN52 = 4 };
/* 3: the CPU class's nugget="pivot" (GPParams.py:185-186; cholesky_factor(..., "pivot"), linalg/cholesky.py:182-184) */
enum mogp_nugget_type { MOGP_NUG_ADAPTIVE = 0, MOGP_NUG_FIT = 1, MOGP_NUG_FIXED = 2, MOGP_NUG_PIVOT = 3 };
enum mogp_prior_type { MOGP_PRIOR_INVGAMMA = 0, MOGP_PRIOR_GAMMA = 1, MOGP_PRIOR_LOGNORMAL = 2, MOGP_PRIOR_WEAK = 3 };

typedef struct mogp_meanfunc mogp_meanfunc;   /* BaseMeanFunc   meanfunc.hpp:24-66            */
typedef struct mogp_densegp mogp_densegp;     /* DenseGP_GPU    densegp_gpu.hpp:36-867        */
typedef struct mogp_mogp mogp_mogp;           /* MultiOutputGP_GPU multioutputgp_gpu.hpp:35-287 */

/* ---- library-level ------------------------------------------------------------------ */
const char* mogp_last_error(void);
/* have_compatible_device(), bindings.cu:600 / util.hpp:40-47 */
int mogp_have_compatible_device(void);
int mogp_device_count(void);
/* select the HIP device used by handles created afterwards on this thread (one process per GPU) */
int mogp_set_device(int device);
const char* mogp_version(void);

/* ---- mean functions (bindings.cu:365-413, meanfunc.hpp) ------------------------------ */
mogp_meanfunc* mogp_meanfunc_zero(void);                       /* ZeroMeanFunc()  */
mogp_meanfunc* mogp_meanfunc_fixed(double value);              /* FixedMeanFunc(v) */
mogp_meanfunc* mogp_meanfunc_const(void);                      /* ConstMeanFunc()  */
/* PolyMeanFunc([[dim,power],...]); parameters = [const, coeff_1 .. coeff_nterms] */
mogp_meanfunc* mogp_meanfunc_poly(const int* dims, const int* powers, int nterms);
void mogp_meanfunc_destroy(mogp_meanfunc*);
int mogp_meanfunc_n_params(const mogp_meanfunc*);
/* mean_f(xs (m,D), params) -> out (m) ; wrong params length -> "Expected params list of length N" */
int mogp_meanfunc_mean_f(const mogp_meanfunc*, const double* xs, int m, int D, const double* params, int n_params, double* out);
/* mean_deriv -> out (n_params, m) ; mean_inputderiv -> out (D, m)   (meanfunc.hpp layouts) */
int mogp_meanfunc_mean_deriv(const mogp_meanfunc*, const double* xs, int m, int D, const double* params, int n_params, double* out);
int mogp_meanfunc_mean_inputderiv(const mogp_meanfunc*, const double* xs, int m, int D, const double* params, int n_params, double* out);

/* ---- DenseGP_GPU (bindings.cu:14-257) -------------------------------------------------- */
/* DenseGP_GPU(inputs(n,D), targets(n), testing_size, meanfunc, kernel_type, nugget_type, nugget_size)
 * bindings.cu:14-15 / densegp_gpu.hpp:777-865.  The mean function is cloned. */
mogp_densegp* mogp_densegp_create(const double* inputs, int n, int D, const double* targets, unsigned testing_size,
                                  const mogp_meanfunc* mean, int kernel_type, int nugget_type, double nugget_size);
/* Same constructor, but the coefficients of a Const / Poly mean function are integrated out
 * analytically with weak (flat) mean priors -- the CPU class's semantics (GaussianProcess.py:640-700
 * fit, :860-940 predict; linalg_utils.py:5-121) instead of living in theta.  n_mean() is then 0 and
 * the fitted coefficients are read with mogp_densegp_get_beta.  At most 7 mean terms. */
mogp_densegp* mogp_densegp_create_analytic_mean(const double* inputs, int n, int D, const double* targets, unsigned testing_size,
                                                const mogp_meanfunc* mean, int kernel_type, int nugget_type, double nugget_size);
/* MeanPriors(mean = b, cov = B) for the analytic mean (Priors.py:423-581, GPPriors.mean): the caller passes b (q),
 * B^-1 (q x q row-major), B^-1 b (q) and log|B|; q = 0 restores weak priors.  Works on borrowed handles
 * (mogp_mogp_emulator) too, so every emulator can carry its own prior. */
int mogp_densegp_set_mean_priors(mogp_densegp*, int q, const double* b, const double* Binv, const double* Binv_b, double logdetB);
int mogp_densegp_n_beta(const mogp_densegp*);                        /* GPParams.n_mean, GPParams.py */
int mogp_densegp_get_beta(const mogp_densegp*, double* out /* n_beta */);   /* theta.mean, GaussianProcess.py:669 */
/* the reference never frees (py::nodelete); the shim may.  Must not be called on a handle
 * obtained from mogp_mogp_emulator(). */
void mogp_densegp_destroy(mogp_densegp*);

int mogp_densegp_n(const mogp_densegp*);                 /* n()        bindings.cu:19 */
int mogp_densegp_D(const mogp_densegp*);                 /* D()        :23 */
int mogp_densegp_n_corr(const mogp_densegp*);            /* n_corr()   :27 */
int mogp_densegp_n_params(const mogp_densegp*);          /* n_params() :41  (D+1+[fit]) */
int mogp_densegp_n_mean(const mogp_densegp*);            /* get_theta().get_n_mean() */
int mogp_densegp_n_data(const mogp_densegp*);            /* get_theta().get_n_data() */
int mogp_densegp_inputs(const mogp_densegp*, double* out /* n*D */);   /* inputs()  :31 */
int mogp_densegp_targets(const mogp_densegp*, double* out /* n */);     /* targets() :36 */
int mogp_densegp_theta_fit_status(const mogp_densegp*);  /* :45 */
int mogp_densegp_reset_theta_fit_status(mogp_densegp*);  /* :50 */
/* get_theta() :55 -- copies: data (n_data) and mean (n_mean) */
int mogp_densegp_get_theta(const mogp_densegp*, double* data_out, double* mean_out);
/* create_gppriors(n_corr, [(type,[a,b])...], (type,[a,b]), (type,[a,b]))  :67 / densegp_gpu.hpp:214-232 */
int mogp_densegp_create_gppriors(mogp_densegp*, int n_corr, const int* corr_types, const double* corr_params /* 2*n_corr */,
                                 int cov_type, const double* cov_params /* 2 */, int nug_type, const double* nug_params /* 2 */);
/* GPPriors.logp / dlogpdtheta / sample of the attached priors (bindings.cu:547-566) */
int mogp_densegp_priors_logp(const mogp_densegp*, const double* theta_data, int len, double* out);
int mogp_densegp_priors_dlogpdtheta(const mogp_densegp*, const double* theta_data, int len, double* out /* len */);
int mogp_densegp_priors_sample(mogp_densegp*, double* out /* n_mean + n_data */);
/* fit(theta) :147-165 / densegp_gpu.hpp:483-612 */
int mogp_densegp_fit(mogp_densegp*, const double* theta, int len);
/* get_logpost(theta) :199-206 / densegp_gpu.hpp:639-661 (refits iff theta moved) */
int mogp_densegp_get_logpost(mogp_densegp*, const double* theta, int len, double* out);
/* logpost_deriv(out) :246-256 / densegp_gpu.hpp:663-767 ; out has n_mean + n_data entries */
int mogp_densegp_logpost_deriv(mogp_densegp*, double* out, int len);
/* predict(testing (D)) :71-80 ; predict_variance :83-95 */
int mogp_densegp_predict(mogp_densegp*, const double* testing, int D, double* mean_out);
int mogp_densegp_predict_variance(mogp_densegp*, const double* testing, int D, double* mean_out, double* var_out);
/* predict_batch(testing (m,D), out (m)) :98-110 / densegp_gpu.hpp:300-338 */
int mogp_densegp_predict_batch(mogp_densegp*, const double* testing, int m, int D, double* mean_out, int out_len);
/* predict_variance_batch(testing, mean, var) :113-129 / densegp_gpu.hpp:341-408 (var WITHOUT nugget) */
int mogp_densegp_predict_variance_batch(mogp_densegp*, const double* testing, int m, int D, double* mean_out, double* var_out, int out_len);
/* predict_deriv(testing, out (m,D)) :132-144 / densegp_gpu.hpp:411-448 */
int mogp_densegp_predict_deriv(mogp_densegp*, const double* testing, int m, int D, double* out, int out_rows, int out_cols);
/* predict(full_cov=True) of the CPU class, GaussianProcess.py:899-911: mean_out (m) and the
 * m x m predictive covariance sigma^2 k(X*,X*) - (L^-1 K*)^T (L^-1 K*) [+ analytic-mean term], nugget NOT added.
 * m is not limited by testing_size; the call fails if the device scratch (2 n m + m^2 doubles per emulator) exceeds 64 GB. */
int mogp_densegp_predict_full_cov(mogp_densegp*, const double* testing, int m, int D, double* mean_out, double* cov_out /* m*m */);
/* get_K :168 (sigma^2 k(X,X), no nugget) ; get_invQ :177 ; get_invQt :187 ; get_cholesky_lower :232
 * get_cholesky_lower fills `out` so that tril(out^T) == L  (densegp_gpu.hpp:478-481) */
/* Consumers of the batched prediction, fused on the device so that a query sweep returns one
 * score per point instead of means and variances:
 *  - implausibility |z - E f(x)| / sqrt(Var f(x) [+ nugget] + discrepancy + obs_var)  (HistoryMatching.py:197-276);
 *    zero / fixed mean functions only;
 *  - leave-one-out predictive variance at every training input, 1 / [K^-1]_ii: what MICEFastGP.fast_predict(index)
 *    computes for one index from a Woodbury downdate (SequentialDesign.py:705-747), for all indices at once. */
int mogp_densegp_implausibility(mogp_densegp*, const double* testing, int m, int D, double obs, double obs_var, double discrepancy,
                                int include_nugget, double* out /* m */);
int mogp_densegp_loo_variance(mogp_densegp*, double* out /* n */);
int mogp_densegp_get_K(mogp_densegp*, double* out /* n*n */);
int mogp_densegp_get_invQ(mogp_densegp*, double* out /* n*n */);
int mogp_densegp_get_invQt(mogp_densegp*, double* out /* n */);
int mogp_densegp_get_cholesky_lower(mogp_densegp*, double* out /* n*n */);
/* ChoInvPivot.P of the current fit (linalg/cholesky.py:82-104): K[P][:, P] = L L^T with L = get_cholesky_lower; the identity
 * and rank n for the other nugget types.  rank_out may be NULL. */
int mogp_densegp_get_pivot(mogp_densegp*, int* P_out /* n */, int* rank_out);
/* pivot_cholesky(A), linalg/cholesky.py:284-327 (LAPACK dpstrf + the replacement diagonal of the skipped rows), for any
 * symmetric matrix with positive diagonal: host buffers, row-major; L_out lower triangular, P_out zero-based. */
int mogp_pivot_cholesky(const double* A, int n, double* L_out /* n*n */, int* P_out /* n */, int* rank_out);
double mogp_densegp_get_nugget_size(const mogp_densegp*);     /* :209 */
int mogp_densegp_set_nugget_size(mogp_densegp*, double);      /* :213 */
int mogp_densegp_get_nugget_type(const mogp_densegp*);        /* :217 */
int mogp_densegp_set_nugget_type(mogp_densegp*, int);         /* :221 */
int mogp_densegp_get_kernel_type(const mogp_densegp*);        /* :225 */
/* fit_GP_MAP(DenseGP_GPU&, n_tries, theta0)  bindings.cu:602-603 / fitting.hpp:61-120.
 * theta0_len == 0 means "no theta0".  Failed starts are skipped; if all fail the emulator is
 * left "not fit" (fitting.hpp:111-113) and the call still returns 0. */
int mogp_fit_single_GP_MAP(mogp_densegp*, int n_tries, const double* theta0, int theta0_len);

/* ---- MultiOutputGP_GPU (bindings.cu:260-337) ------------------------------------------- */
/* MultiOutputGP_GPU(inputs(n,D), targets (n_out,n), testing_size, meanfunc, kernel, nugget_type, nugget_size) */
mogp_mogp* mogp_mogp_create(const double* inputs, int n, int D, const double* targets, int n_out, unsigned testing_size,
                            const mogp_meanfunc* mean, int kernel_type, int nugget_type, double nugget_size);
mogp_mogp* mogp_mogp_create_analytic_mean(const double* inputs, int n, int D, const double* targets, int n_out, unsigned testing_size,
                                          const mogp_meanfunc* mean, int kernel_type, int nugget_type, double nugget_size);
void mogp_mogp_destroy(mogp_mogp*);
int mogp_mogp_n(const mogp_mogp*);
int mogp_mogp_D(const mogp_mogp*);
int mogp_mogp_n_emulators(const mogp_mogp*);
int mogp_mogp_inputs(const mogp_mogp*, double* out);
int mogp_mogp_targets(const mogp_mogp*, double* out /* n_out*n */);
/* emulator(i): borrowed DenseGP handle sharing the container's device state (bindings.cu:275) */
mogp_densegp* mogp_mogp_emulator(mogp_mogp*, int index);
int mogp_mogp_get_nugget_type(const mogp_mogp*);
double mogp_mogp_get_nugget_size(const mogp_mogp*);
/* get_fitted_indices / get_unfitted_indices: writes up to n_emulators ints, returns the count */
int mogp_mogp_get_fitted_indices(const mogp_mogp*, int* out);
int mogp_mogp_get_unfitted_indices(const mogp_mogp*, int* out);
int mogp_mogp_reset_fit_status(mogp_mogp*);
/* create_priors_for_emulator(i, ...) multioutputgp_gpu.hpp:124-134 */
int mogp_mogp_create_priors_for_emulator(mogp_mogp*, int index, int n_corr, const int* corr_types, const double* corr_params,
                                         int cov_type, const double* cov_params, int nug_type, const double* nug_params);
/* fit(thetas (n_out, n_params)) multioutputgp_gpu.hpp:223-228 : ONE batched device pass over all emulators */
int mogp_mogp_fit(mogp_mogp*, const double* thetas, int n_rows, int n_cols);
int mogp_mogp_fit_emulator(mogp_mogp*, int index, const double* theta, int len);
/* batched objective / gradient for all emulators at once (what fit_GP_MAP iterates on):
 * logpost_out (n_out), grad_out (n_out, n_params) may be NULL; ok_out (n_out) 1 = factorised */
int mogp_mogp_eval(mogp_mogp*, const double* thetas, int n_rows, int n_cols, double* logpost_out, double* grad_out, int* ok_out);
/* predict_batch(testing (m,D), out (n_out,m)) :170-179 ; predict_variance_batch :182-192 ; predict_deriv (n_out,m,D) :195-203
 * unfit emulators are skipped (their rows are left untouched), as in the reference. */
int mogp_mogp_predict_batch(mogp_mogp*, const double* testing, int m, int D, double* means);
int mogp_mogp_predict_variance_batch(mogp_mogp*, const double* testing, int m, int D, double* means, double* vars);
int mogp_mogp_predict_deriv(mogp_mogp*, const double* testing, int m, int D, double* derivs);
/* full predictive covariance of every fitted emulator in one batched pass: means (n_out, m), covs (n_out, m, m) */
/* multi-output implausibility: obs / obs_var / discrepancy (n_out); out[j] = the (rank+1)-th largest of the n_out
 * per-output implausibilities at query point j (rank 0 = maximum; forced to 0 for one output; rank <= 15) */
int mogp_mogp_implausibility(mogp_mogp*, const double* testing, int m, int D, const double* obs, const double* obs_var,
                             const double* discrepancy, int include_nugget, int rank, double* out /* m */);
int mogp_mogp_predict_full_cov(mogp_mogp*, const double* testing, int m, int D, double* means, double* covs);
/* predict_variance_batch (multioutputgp_gpu.hpp:182-192) with DEVICE pointers: inputs already resident in HBM, results stay in HBM
 * (every mean function; rows of emulators that are not fit are filled with NaN, MultiOutputGP_GPU.py:288-296) */
int mogp_mogp_predict_variance_batch_dev(mogp_mogp*, const double* d_testing, int m, int D, double* d_means, double* d_vars);
/* the same with the input derivatives of predict_deriv (multioutputgp_gpu.hpp:195-203) in the same pass: d_derivs (n_out, m, D) device
 * buffer; d_vars and d_derivs may be NULL.  What one rank of a sharded predict hands to the single RCCL gather (dist.py). */
int mogp_mogp_predict_dev(mogp_mogp*, const double* d_testing, int m, int D, double* d_means, double* d_vars, double* d_derivs);
/* fit_GP_MAP(MultiOutputGP_GPU&, n_tries, theta0) bindings.cu:604-605 / fitting.hpp:122-128.
 * All emulators advance in lock-step: each L-BFGS iteration is one batched device evaluation. */
int mogp_fit_GP_MAP(mogp_mogp*, int n_tries, const double* theta0, int theta0_len);
/* optimiser controls (not in the reference API; defaults reproduce fitting.hpp's stop rule 1e-9) */
int mogp_set_fit_options(int max_iter, double ftol, double gtol, unsigned long long seed);

/* ---- stand-alone kernel objects: SquaredExponentialKernel / Matern52Kernel .kernel_f / kernel_deriv / kernel_inputderiv
   (mogp_gpu/src/bindings.cu:340-361, kernel.hpp:47-107; CPU counterparts Kernel.py:99-173).
   kernel_type as the enum above (+ 2 ProductMat52, 3 UniformSqExp, 4 UniformMat52); x1 (n1, D), x2 (n2, D) row-major;
   params = [corr_raw.., log sigma^2] (n_corr + 1, n_corr = D or 1 for the uniform kernels).
   what = 0: out (n1, n2)            = sigma^2 k(x1_i, x2_j)
   what = 1: out (n_corr + 1, n1, n2) = d/d theta_p (last plane: d/d log sigma^2 = K)
   what = 2: out (n2, n1, D)         = d/d x1_i[d]   (the flat order of the reference's CUDA kernel, kernel.cu:86-100) */
int mogp_kernel_eval(int kernel_type, int what, const double* x1, int n1, const double* x2, int n2, int D, const double* params,
                     int n_params, double* out);

/* ---- measurement hooks (bench.py only) -------------------------------------------------- */
/* when enabled, HIP events are recorded on the launch stream around every launch of the tagged kernels */
int mogp_profile_enable(int on);
int mogp_profile_reset(void);
/* force the Cholesky schedule (0 two emulator groups, 1 right-looking, 3 look-ahead, 4 one launch / task queue, 5 the multi-launch
   schedule the library would pick without the one-launch kernel, -1 the library's choice) and / or
   serialise it onto one stream, so that the HIP-event time of a kernel is its time alone on the device */
int mogp_profile_schedule(int schedule, int single_stream);
/* sums over launches since reset: total milliseconds, launch count, algorithmic flops and bytes */
int mogp_profile_get(const char* kernel_tag, double* total_ms, long long* launches, double* alg_flops, double* alg_bytes);
/* The per-emulator task order of the one-launch Cholesky (kernels_mchol.hip) for n_plus_rhs = n + number of right-hand-side rows,
   padded to a multiple of 128: entry = (type << 30) | (block column c << 15) | 64-row block r; type 0 D(c): diagonal block, 1 G(s, c): lower
   64 x 64 tile s = 0, 1, 2 of the diagonal block (s in the r field) receives the panels 0 .. c-2, 2 T(r, c): panel tile receives the panels 0 .. c-1 and is solved.  Writes up to
   `capacity` entries, returns the number of tasks.  Host-only (no device needed): the CPU suite checks that the order is
   topological, which is what the kernel's forward-progress argument rests on. */
int mogp_mchol_task_table(int n_plus_rhs, int* out, int capacity);
/* The order launches with enough workgroups per queue use (kernels_mchol.hip, mchol_task_table): the same tasks, the band tasks of an
   iteration -- entries with bit 29 set; the block column is bits 15..28 -- listed directly in front of the diagonal block they wait for, at
   most 6 places ahead of it with only such entries in between.  The CPU suite checks exactly that bound. */
int mogp_mchol_task_table_ahead(int n_plus_rhs, int* out, int capacity);
/* process-wide diagnostic counters: "backsolve_timeouts" = back substitutions that were repeated with the multi-launch
   path because a wait of the one-launch chain timed out (0 in normal operation); "mchol_aborts" = factorisations the
   one-launch Cholesky gave up on and a multi-launch schedule repeated (0 in normal operation); "objective_evals" / "gradient_evals" =
   emulator objective evaluations so far (all / with gradient), e.g. to turn a fit_GP_MAP wall time into evaluations/s;
   "lbfgs_runs" / "lbfgs_iterations" / "linesearch_shortened" / "linesearch_lengthened" = optimiser runs started by fit_GP_MAP,
   their accepted steps, and the line-search trial points that failed the sufficient-decrease / the curvature test;
   round 6, the slot pool of fit_GP_MAP: "pool_rounds" = batched optimiser rounds, "pool_slot_rounds" = sum over them of the runs that took
   part (/ rounds = mean batch), "retargets" / "retarget_us" = replica slots handed to another emulator and the host time that took,
   "replica_engine_build_us" / "replica_pool_us" = host time constructing (or re-taking) the replica engine / inside the pool,
   "replica_engines_reused" = multi-start fits that ran on the replica engine the previous one left behind (MOGP_REPLICA_CACHE) */
int mogp_profile_counter(const char* name, long long* out);
/* device memory helpers so a host program can hand device-resident buffers to the *_dev calls */
void* mogp_dev_malloc(unsigned long long bytes);
int mogp_dev_free(void* d_ptr);
int mogp_dev_upload(void* d_dst, const void* src, unsigned long long bytes);
int mogp_dev_download(void* dst, const void* d_src, unsigned long long bytes);
int mogp_dev_synchronize(void);

#ifdef __cplusplus
}
#endif
#endif /* MOGP_HIP_H */
