#include "engine.h"

#include <algorithm>
#include <memory>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>

namespace mogp {

void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}
#define HIPCK(x) hip_check((x), #x)

FitOptions& fit_options() {
  static FitOptions o;
  return o;
}
ScheduleOverride& schedule_override() {
  static ScheduleOverride o;
  return o;
}

// ---------------------------------------------------------------------------------------------
// profiling registry (bench.py): HIP events on the launch stream around tagged kernels
// ---------------------------------------------------------------------------------------------
namespace {
struct ProfRec {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  double flops = 0., bytes = 0., ms_done = 0.;
  long long launches = 0;
  hipEvent_t pending = nullptr;
};
bool g_prof_on = false;
std::map<std::string, ProfRec> g_prof;
std::mutex g_prof_mu;
}  // namespace

void prof_begin(const char* tag, hipStream_t s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec& r = g_prof[tag];
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  hipEventRecord(e, s);
  r.pending = e;
}
void prof_end(const char* tag, hipStream_t s, double flops, double bytes) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec& r = g_prof[tag];
  if (!r.pending) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  hipEventRecord(e, s);
  r.ev.emplace_back(r.pending, e);
  r.pending = nullptr;
  r.flops += flops;
  r.bytes += bytes;
  r.launches += 1;
}
void prof_enable(bool on) { g_prof_on = on; }
// sets a flag for the lifetime of a scope (cleared again when the scope is left through an exception)
struct FlagGuard {
  bool& b;
  explicit FlagGuard(bool& f) : b(f) { b = true; }
  ~FlagGuard() { b = false; }
};
static std::atomic<long long> g_bs_timeouts{0}, g_obj_evals{0}, g_grad_evals{0}, g_mc_aborts{0};
static std::atomic<long long> g_lb_iters{0}, g_ls_short{0}, g_ls_long{0}, g_lb_runs{0}, g_pool_rounds{0}, g_pool_slot_rounds{0}, g_rep_build_us{0}, g_rep_pool_us{0}, g_retarget_us{0}, g_retargets{0}, g_rep_reused{0};
long long prof_counter(const char* name) {
  const std::string s(name ? name : "");
  if (s == "backsolve_timeouts") return g_bs_timeouts.load();
  if (s == "mchol_aborts") return g_mc_aborts.load();          // one-launch factorisations repeated with a multi-launch schedule
  if (s == "objective_evals") return g_obj_evals.load();      // emulator objective evaluations (with or without gradient)
  if (s == "gradient_evals") return g_grad_evals.load();      // of which with gradient
  // optimiser statistics of fit_GP_MAP: runs started, accepted L-BFGS steps, line-search trial points that were shortened
  // (sufficient decrease failed) / lengthened (curvature condition failed)
  if (s == "lbfgs_runs") return g_lb_runs.load();
  if (s == "lbfgs_iterations") return g_lb_iters.load();
  if (s == "linesearch_shortened") return g_ls_short.load();
  if (s == "linesearch_lengthened") return g_ls_long.load();
  // slot pool of fit_GP_MAP: batched optimiser rounds, and the sum over rounds of the slots that took part (/ rounds = mean batch)
  if (s == "pool_rounds") return g_pool_rounds.load();
  if (s == "pool_slot_rounds") return g_pool_slot_rounds.load();
  if (s == "replica_engine_build_us") return g_rep_build_us.load();      // host time spent constructing replica engines (allocations)
  if (s == "retarget_us") return g_retarget_us.load();                   // host time inside Engine::retarget (slot takes another emulator's targets)
  if (s == "retargets") return g_retargets.load();
  if (s == "replica_engines_reused") return g_rep_reused.load();          // multi-start fits that took the cached replica engine
  if (s == "replica_pool_us") return g_rep_pool_us.load();               // ... from there to the end of fit_map's replica block (pool + its destruction excluded)
  return -1;
}
bool prof_is_on() { return g_prof_on; }
void prof_reset() {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& kv : g_prof) {
    for (auto& p : kv.second.ev) {
      hipEventDestroy(p.first);
      hipEventDestroy(p.second);
    }
  }
  g_prof.clear();
}
bool prof_get(const char* tag, double* ms, long long* launches, double* flops, double* bytes) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  auto it = g_prof.find(tag);
  if (it == g_prof.end()) return false;
  ProfRec& r = it->second;
  for (auto& p : r.ev) {
    hipEventSynchronize(p.second);
    float t = 0.f;
    hipEventElapsedTime(&t, p.first, p.second);
    r.ms_done += t;
    hipEventDestroy(p.first);
    hipEventDestroy(p.second);
  }
  r.ev.clear();
  *ms = r.ms_done;
  *launches = r.launches;
  *flops = r.flops;
  *bytes = r.bytes;
  return true;
}

// ---------------------------------------------------------------------------------------------
static int roundup(int x, int m) { return (x + m - 1) / m * m; }

template <class T>
static T* dalloc(size_t count) {
  T* p = nullptr;
  HIPCK(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T)));
  return p;
}
template <class T>
static void grow(T*& p, size_t& cap, size_t need) {
  if (need <= cap) return;
  if (p) HIPCK(hipFree(p));
  p = dalloc<T>(need);
  cap = need;
}

Engine::Engine(const double* X, int n_, int D_, const double* targets, int B_, unsigned testing_size_, const MeanFunc& mean_,
               int kernel_type_, int nug_type, double nug_size, bool analytic_mean)
    : n(n_), D(D_), B(B_), kernel_type(kernel_type_), testing_size(testing_size_), mean(mean_) {
  analytic = analytic_mean && mean.n_params() > 0;
  q = analytic ? mean.n_params() : 0;
  R = 1 + q;
  if (R > RMAX) throw std::runtime_error("analytic mean: at most " + std::to_string(RMAX - 1) + " mean-function terms are supported");
  if (analytic && q >= n_) throw std::runtime_error("analytic mean: more mean-function terms than training points");
  if (n < 1 || D < 1 || B < 1) throw std::runtime_error("inputs must have shape (n, D) with n, D >= 1");
  if (kernel_type < 0 || kernel_type > 4) throw std::runtime_error("Unrecognized kernel type\n");
  if (D < 1) throw std::runtime_error("inputs must have at least one column");
  if (D > MAX_D)
    throw std::runtime_error("at most " + std::to_string(MAX_D) + " input dimensions are supported by the device kernels (" +
                             std::to_string(D) + " given): the per-tile copy of the inputs must fit the 160 KB LDS");
  if (nug_type < 0 || nug_type > 3) throw std::runtime_error("Unrecognized nugget_type");
  for (int d : mean.dims)
    if (d >= D) throw std::runtime_error("Dimension index must be less than " + std::to_string(D));
  NP = roundup(n + R, TILE);
  // Row stride = NP.  A non-power-of-two stride (NP + 16) was tried against L2 set aliasing of the 16 KiB-strided tile rows
  // and measured SLOWER on MI355X (fit 7.5 -> 8.2 ms, fit+grad 15.3 -> 16.2 ms; predictive variance unchanged).
  LD = NP;
  MS = (size_t)NP * LD;
  NC = uniform() ? 1 : D;
  PS = D + 2;
  hX.assign(X, X + (size_t)n * D);
  hT.assign(targets, targets + (size_t)B * n);
  gp.resize(B);
  for (auto& g : gp) {
    g.nug_type = nug_type;
    g.nug_size = nug_size;
    g.data.assign(NC + 1 + (nug_type == NUG_FIT ? 1 : 0), 0.);
    g.meanp.assign(n_mean(), 0.);
    g.beta.assign(q, 0.);
  }
  {
    // stream memory operations (look-ahead schedule) are a property of the device THIS engine lives on
    int dev = 0, ok = 0;
    int cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) n_cu = cus;
    can_waitval = hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ok, hipDeviceAttributeCanUseStreamWaitValue, dev) == hipSuccess && ok != 0;
    device = dev;
  }
  HIPCK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  HIPCK(hipEventCreateWithFlags(&evReady, hipEventDisableTiming));
  for (auto& e : evGroup) HIPCK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  {
    int lo = 0, hi = 0;   // numerically lower = higher priority
    HIPCK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCK(hipStreamCreateWithPriority(&pstream, hipStreamNonBlocking, hi));
  }
  dX = dalloc<double>((size_t)n * D);
  dP = dalloc<double>((size_t)B * PS);
  dT = dalloc<double>((size_t)B * n);
  dA = dalloc<double>((size_t)B * MS);
  RA = (R > 1) ? R + 1 : 1;
  dAlpha = dalloc<double>((size_t)B * RA * LD);
  if (R > 1) {
    dZ = dalloc<double>((size_t)B * R * LD);
    dM = dalloc<double>((size_t)B * (RMAX + 1) * RMAX);
    // design matrix columns: row c of mean_deriv = d mean / d beta_c = basis function c evaluated at X
    hH.assign((size_t)q * n, 0.);
    std::vector<double> dummy(q, 0.);
    mean.mean_deriv(X, n, D, dummy.data(), q, hH.data());
    dH = dalloc<double>(hH.size());
    HIPCK(hipMemcpy(dH, hH.data(), hH.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  dRes = dalloc<double>((size_t)B * RES_STRIDE);
  HIPCK(hipHostMalloc(reinterpret_cast<void**>(&hRes), (size_t)B * RES_STRIDE * sizeof(double), hipHostMallocDefault));
  dInfo = dalloc<int>(B);
  dIdx = dalloc<int>(B);
  dLpack = dalloc<double>((size_t)B * lpack128_doubles_per_emulator());
  // (pinned: the parameter block goes up in front of every evaluation, and an asynchronous copy from pageable memory is staged by the runtime)
  HIPCK(hipHostMalloc(reinterpret_cast<void**>(&hP), (size_t)B * PS * sizeof(double), hipHostMallocDefault));
  std::fill(hP, hP + (size_t)B * PS, 0.);
  HIPCK(hipMemcpy(dX, hX.data(), hX.size() * sizeof(double), hipMemcpyHostToDevice));
  // residual targets for parameter-free means are fixed once
  std::vector<double> res(hT);
  if (!analytic && mean.n_params() == 0 && mean.kind == 1)
    for (auto& x : res) x -= mean.value;
  HIPCK(hipMemcpy(dT, res.data(), res.size() * sizeof(double), hipMemcpyHostToDevice));
  rng.seed(fit_options().seed ? fit_options().seed : std::random_device{}());
}

Engine::~Engine() {
  for (void* p : {(void*)dX, (void*)dP, (void*)dT, (void*)dA, (void*)dLinv, (void*)dKinv, (void*)dAlpha, (void*)dRes,
                  (void*)dGradOut, (void*)dGradPartial, (void*)dInfo, (void*)dIdx, (void*)dXs, (void*)dKs, (void*)dMean, (void*)dVar,
                  (void*)dVarPartial, (void*)dDeriv, (void*)dLpack, (void*)dH, (void*)dZ, (void*)dM, (void*)dXp, (void*)dPivWork,
                  (void*)dPerm, (void*)dRank, (void*)dMeanFin, (void*)dMeanAux})
    if (p) hipFree(p);
  if (hRes) hipHostFree(hRes);
  if (hP) hipHostFree(hP);
  if (dBsFlags) hipFree(dBsFlags);
  for (void* p : {(void*)dMcTable, (void*)dMcCtrl, (void*)dMcPacks})
    if (p) hipFree(p);
  if (sigU1) hipFree(sigU1);
  for (auto& kv : w2) hipFree(kv.second);
  for (auto st : gstreams) hipStreamDestroy(st);
  if (evReady) hipEventDestroy(evReady);
  for (auto e : evGroup) if (e) hipEventDestroy(e);
  for (auto e : evPanel) hipEventDestroy(e);
  for (auto e : evUpd) hipEventDestroy(e);
  if (pstream) hipStreamDestroy(pstream);
  if (stream) hipStreamDestroy(stream);
}

double Engine::nugget_size(int i) const {
  const GPState& g = gp[i];
  if (g.nug_type == NUG_FIT) return std::exp(g.data[NC + 1]);   // zero-initialised data -> 1 before the first fit   // gpparams.hpp:176-182
  return g.nug_size;
}

// K build of the slots in dIdx.  With one right-hand side it also presets the solution rows to the all-ones pattern the one-launch
// back substitution polls for (backsolve_chain_kernel<SENT>); `z_armed` records that, so that a chain launched on a row that has
// not been preset since its last solve arms it itself instead of reading stale values as "already there".
void Engine::build_cov(const BatchView& v, const ZeroRanges& zero) {
  launch_cov_build(v, stream, zero);
  if (z_armed.size() != (size_t)B) z_armed.assign(B, 0);
  for (int i : idx_on_device) z_armed[i] = (R == 1 && v.Z != nullptr) ? 1 : 0;
}

BatchView Engine::view(int nb) const {
  BatchView v;
  v.n = n; v.D = D; v.NP = NP; v.LD = LD; v.MS = MS; v.PS = PS; v.kernel_type = device_kernel();
  v.X = dXp ? dXp : dX; v.XS = dXp ? (size_t)n * D : 0; v.P = dP; v.T = dT; v.A = dA; v.Linv = dLinv; v.Kinv = dKinv; v.alpha = dAlpha;
  v.idx = dIdx; v.nb = nb;
  v.R = R; v.RA = RA; v.H = dH; v.Z = (R > 1) ? dZ : dAlpha;
  return v;
}

void Engine::upload_idx(const std::vector<int>& ids) {
  if (ids == idx_on_device) return;      // the optimiser and the benchmark evaluate the same list again and again
  HIPCK(hipMemcpyAsync(dIdx, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice, stream));
  HIPCK(hipStreamSynchronize(stream));   // ids may be a temporary
  idx_on_device = ids;
}

void Engine::set_mean_priors(int i, int q_in, const double* b, const double* Binv, const double* Binvb, double logdetB) {
  GPState& g = gp[i];
  if (q_in == 0) {
    g.mp_b.clear(); g.mp_Binv.clear(); g.mp_Binvb.clear(); g.mp_logdetB = 0.;
  } else {
    if (!analytic) throw std::runtime_error("mean priors need the analytic mean function (analytic_mean=True)");
    if (q_in != q) throw std::runtime_error("mean priors must have one entry per mean-function term (" + std::to_string(q) + ")");
    g.mp_b.assign(b, b + q);
    g.mp_Binv.assign(Binv, Binv + (size_t)q * q);
    g.mp_Binvb.assign(Binvb, Binvb + q);
    g.mp_logdetB = logdetB;
  }
  g.has_data = false;
  g.factored = g.linv = g.kinv = false;
}

void Engine::set_theta(int i, const double* theta) {
  GPState& g = gp[i];
  const int nm = n_mean(), nd = n_data(i);
  g.meanp.assign(theta, theta + nm);
  g.data.assign(theta + nm, theta + nm + nd);
  g.has_data = false;
  g.factored = g.linv = g.kinv = false;
  if (g.nug_type == NUG_ADAPTIVE) g.nug_size = 0.;   // gpparams.hpp:118-126
  if (nm > 0) {
    std::vector<double> m(n), r(n);
    mean.mean_f(hX.data(), n, D, g.meanp.data(), nm, m.data());
    for (int k = 0; k < n; ++k) r[k] = hT[(size_t)i * n + k] - m[k];
    HIPCK(hipMemcpyAsync(dT + (size_t)i * n, r.data(), n * sizeof(double), hipMemcpyHostToDevice, stream));
    HIPCK(hipStreamSynchronize(stream));
  }
}

void Engine::upload_params(const std::vector<int>& ids) {
  for (int i : ids) {
    const GPState& g = gp[i];
    double* p = hP + (size_t)i * PS;
    for (int d = 0; d < D; ++d) p[d] = std::exp(g.data[uniform() ? 0 : d]);
    p[D] = std::exp(g.data[NC]);
    p[D + 1] = g.nugget_used;
  }
  HIPCK(hipMemcpyAsync(dP, hP, (size_t)B * PS * sizeof(double), hipMemcpyHostToDevice, stream));
}

// Blocked right-looking Cholesky of K + nugget I for the emulators in `ids` (one batched sequence).
// Recursive panel: a block column of width w is factored as [left half] -> update of the right half
// (K = w/2, MFMA) -> [right half], down to 64-wide leaves (potf2 + trsm).  The outer block is 512
// wide so the big trailing update runs with K = 512: per 128x128 tile the MFMA work then clearly outweighs the
// read-modify-write of C (256 KB per tile), which it does not at K = 128 (measured on C5: 256 -> 45.7 ms, 512 -> 41.7 ms).
// One 128-wide block column [c, c+128), rows [c, NP), K = [k0, k1).  With few 128 x 128 tiles in the launch (a single
// large matrix: (NP - c)/128 <= 125 workgroups on 256 CUs) the 64 x 64 tiling gives 4x the workgroups and the launch
// takes one short tile instead of one long one.
static void update_column_block(const BatchView& v, int c, int k0, int k1, hipStream_t st) {
  if ((long)v.nb * ((v.NP - c) / TILE) < 512L) launch_update_narrow_pair(v, c, k0, k1, st);
  else launch_update_wide(v, c, k0, k1, st);
}

void Engine::panel(const BatchView& v, int o, int w, hipStream_t st) {
  if (w == TILE) {
    launch_panel128(v, o, dInfo, dLpack, st);      // 128 x 128 diagonal block + 128-wide panel solve
    return;
  }
  int h = TILE;                      // largest power of two below w (w is a multiple of 128)
  while (2 * h < w) h *= 2;
  panel(v, o, h, st);
  for (int c = o + h; c < o + w; c += TILE) update_column_block(v, c, o, o + h, st);
  panel(v, o + h, w - h, st);
}

// Look-ahead schedule on two HIP streams: as soon as the columns of the NEXT outer block have
// received the update from panel k (U_a, main stream), panel k+1 is factored on the panel stream
// while the main stream applies panel k to the rest of the trailing matrix (U_b).  The
// latency-bound panel kernels (potf2 / trsm, few workgroups) thereby run underneath the MFMA
// trailing update instead of in front of it.
void Engine::ensure_pivot_buffers() {
  if (dXp) return;
  dPerm = dalloc<int>((size_t)B * n);
  dRank = dalloc<int>(B);
  dPivWork = dalloc<double>((size_t)B * pstrf_work_doubles(NP));
  hPerm.resize((size_t)B * n);
  for (int i = 0; i < B; ++i)
    for (int k = 0; k < n; ++k) hPerm[(size_t)i * n + k] = k;
  double* xp = dalloc<double>((size_t)B * n * D);
  for (int i = 0; i < B; ++i)
    HIPCK(hipMemcpyAsync(xp + (size_t)i * n * D, dX, (size_t)n * D * sizeof(double), hipMemcpyDeviceToDevice, stream));
  HIPCK(hipStreamSynchronize(stream));
  dXp = xp;
}

// nugget="pivot" (cholesky_factor(K, nugget, "pivot"), linalg/cholesky.py:182-184): K without nugget, factored with
// diagonal pivoting; afterwards the emulator's inputs are held in pivot order, so that every later kernel (prediction,
// gradient, L^-1, K^-1) works on an ordinary lower-triangular factor of k(Xp, Xp) and never sees the permutation.
void Engine::factorize_pivot(const std::vector<int>& ids, std::vector<int>& info) {
  const int nb = (int)ids.size();
  ensure_pivot_buffers();
  for (int i : ids) gp[i].nugget_used = 0.;
  upload_idx(ids);
  upload_params(ids);
  BatchView v = view(nb);
  v.X = dX;          // the covariance is built in training order; the interchanges happen inside the factorisation
  v.XS = 0;
  build_cov(v);
  launch_pstrf_begin(v, dPerm, dRank, dInfo, dPivWork, stream);
  std::vector<int> rank(B, 0), inf(B, 0);
  std::vector<int> active(ids), stopped;
  for (int k0 = 0; k0 < n && !active.empty(); k0 += NBI) {
    const bool first_half = (k0 % TILE) == 0;
    launch_pstrf_panel(v, k0, std::min(NBI, n - k0), dPerm, dRank, dPivWork, stream);
    HIPCK(hipMemcpyAsync(rank.data(), dRank, B * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCK(hipStreamSynchronize(stream));
    std::vector<int> still;
    for (int i : active) {
      if (rank[i] < 0) still.push_back(i);
      else if (rank[i] < n) stopped.push_back(i);
    }
    if (still.size() != active.size()) {
      active.swap(still);
      if (active.empty()) break;
      upload_idx(active);
      v = view((int)active.size());
      v.X = dX;
      v.XS = 0;
    }
    // rank-64 update of everything to the right of the panel (the 128-wide tiles start at multiples of 128)
    if (first_half) launch_update_narrow(v, k0 + NBI, k0, k0 + NBI, stream);
    launch_update_trailing(v, first_half ? k0 + TILE : k0 + NBI, k0, k0 + NBI, stream);
  }
  if (!stopped.empty()) {
    upload_idx(stopped);
    BatchView t = view((int)stopped.size());
    launch_pstrf_tail(t, dPerm, dRank, dX, nullptr, stream);
  }
  upload_idx(ids);
  v = view(nb);
  launch_pstrf_end(v, stream);
  launch_permute_rows(v, dX, dPerm, dXp, stream);
  HIPCK(hipMemcpyAsync(inf.data(), dInfo, B * sizeof(int), hipMemcpyDeviceToHost, stream));
  HIPCK(hipMemcpyAsync(rank.data(), dRank, B * sizeof(int), hipMemcpyDeviceToHost, stream));
  HIPCK(hipMemcpyAsync(hPerm.data(), dPerm, hPerm.size() * sizeof(int), hipMemcpyDeviceToHost, stream));
  HIPCK(hipStreamSynchronize(stream));
  HIPCK(hipGetLastError());
  if (info.size() != (size_t)B) info.assign(B, 0);
  for (int i : ids) {
    info[i] = inf[i];
    gp[i].rank = rank[i];
    gp[i].permuted = true;
  }
}

void Engine::factorize(const std::vector<int>& ids, std::vector<int>& info, bool defer_info) {
  std::vector<int> piv, rest;
  for (int i : ids) (gp[i].nug_type == NUG_PIVOT ? piv : rest).push_back(i);
  if (piv.empty()) {
    // an emulator that was pivoted earlier goes back to training order
    for (int i : rest)
      if (gp[i].permuted) {
        HIPCK(hipMemcpyAsync(dXp + (size_t)i * n * D, dX, (size_t)n * D * sizeof(double), hipMemcpyDeviceToDevice, stream));
        for (int k = 0; k < n; ++k) hPerm[(size_t)i * n + k] = k;
        gp[i].permuted = false;
        gp[i].rank = 0;
      }
    factorize_blocked(rest, info, defer_info);
    return;
  }
  std::vector<int> tmp;
  if (!rest.empty()) {
    factorize(rest, tmp);
    info = tmp;
  } else {
    info.assign(B, 0);
  }
  factorize_pivot(piv, info);
}

void Engine::read_info(std::vector<int>& info, bool defer_info) {
  if (defer_info) return;
  info.assign(B, 0);
  HIPCK(hipMemcpyAsync(info.data(), dInfo, B * sizeof(int), hipMemcpyDeviceToHost, stream));
  HIPCK(hipStreamSynchronize(stream));
  HIPCK(hipGetLastError());
}

void Engine::factorize_blocked(const std::vector<int>& ids, std::vector<int>& info, bool defer_info) {
  const int nb = (int)ids.size();
  upload_idx(ids);
  upload_params(ids);
  BatchView v = view(nb);
  // schedule: 4 = one launch / task queue (default, kernels_mchol.hip); the multi-launch schedules (fall-back, > 2048 tiles per step):
  // 3 = left-looking with look-ahead, 0 = left-looking in two emulator groups, 1 = right-looking + look-ahead
  static const int forced = [] {
    const char* e = getenv("MOGP_CHOL");
    if (!e) return -1;
    if (e[0] == 'r') return 1;
    if (std::string(e) == "mchol") return 4;
    return (std::string(e) == "left") ? 0 : 3;
  }();
  // Measured (fit, ms; look-ahead / two groups / right-looking): 8 x n=2000 1.75 / 1.97 / 1.89, 16 x 2.21 / 2.38 / 2.44,
  // 32 x 3.32 / 3.34 / 3.65, 64 x 5.47 / 5.37 / 6.76, 16 x n=5000 18.9 / 19.8 / -, 2 x n=5000 6.59 / - / 6.42,
  // 1 x n=16000 59.8 / - / 38.4: one matrix has too few tiles per block column for a left-looking pass (right-looking),
  // a large batch fills the machine with the update of ONE emulator group while the other factors its panels.
  const long tiles64 = (long)nb * (NP / 64), tiles128 = (long)nb * (NP / TILE);
  const ScheduleOverride& ovr = schedule_override();
  // Default: the ONE-LAUNCH task-queue kernel (schedule 4) up to 2048 128-tiles per block-column step.  Fit, ms, one launch /
  // best multi-launch schedule: 8 x n=2000 1.12 / 1.60, 16 x 1.54 / 1.98, 32 x 2.58 / 3.05, 64 x 4.73 - 4.84 / 5.06, 120 x 8.46 /
  // 8.83, 2 x n=5000 2.86 / 5.27, 16 x n=5000 14.4 / 17.5, n=16000 26.3 / 34.5, 3 x n=700 0.40 / 0.54, 64 x n=1000 1.04 / 1.06.
  // Beyond that (the replica engines of a multi-start fit: 240 x n=2000 17.3 / 16.8) and for thousands of single-block
  // matrices (2000 x n=100: 0.62 / 0.57, one task each) the two-group multi-launch schedule stays.
  const int legacy = tiles64 < 256 ? 1 : (tiles128 >= 1024 ? 0 : 3);
  static const bool mc_default = [] { const char* e = getenv("MOGP_MCHOL"); return !e || atoi(e) != 0; }();
  // Round 6: re-measured on the round-5 kernels, the one-launch kernel wins at every batch size -- 128 / 256 / 512 x n=2000: 7.71 / 15.24 /
  // 30.68 ms against 8.87 / 17.54 / 33.70 with the two-group schedule, 1024 x n=1000 11.19 / 11.73, 2048 x n=500 4.50 / 4.91, 4096 x n=250
  // 2.34 / 2.75, 64 x n=5000 52.2 / 56.9 (profiles/r06_big_batch.txt) -- so the bound is now the pack memory alone (147 KB per emulator and
  // block column: 16384 tiles = 2.4 GB); rounds 2-5 stopped at 2048 tiles (measured on the round-2 kernel: 240 x n=2000 17.3 / 16.8).
  const bool mc_regime = tiles128 < 16384 && (NP > TILE || nb <= 512);
  int schedule = ovr.schedule >= 0 ? ovr.schedule : (forced >= 0 ? forced : ((mc_default && mc_regime) ? 4 : legacy));
  if (schedule == 5) schedule = legacy;            // (mogp_profile_schedule(5, ..): the multi-launch schedule of this regime)
  // the one-launch kernel addresses an emulator's matrix through a 32-bit buffer offset; after an abort the multi-launch
  // schedule of the same regime takes over
  if (schedule == 4 && (mc_force_legacy || MS * sizeof(double) >= (size_t)1 << 32)) schedule = legacy;
  mc_used = schedule == 4;
  if (schedule == 4) {
    // ONE LAUNCH: persistent workgroups take the tasks of all block columns from a dependency-ordered queue (kernels_mchol.hip)
    if (!dMcTable) {
      std::vector<int> tb = mchol_task_table(NP);
      const std::vector<int> ta = mchol_task_table(NP, true);
      mc_ntasks = (int)tb.size();
      tb.insert(tb.end(), ta.begin(), ta.end());              // [in-order | band-ahead]: launch_mchol picks
      dMcTable = dalloc<int>(tb.size());
      HIPCK(hipMemcpy(dMcTable, tb.data(), tb.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    if (nb > mc_slots) {
      // control rows and packs are per batch SLOT of a launch, sized for the largest launch seen so far -- not for the engine's B: a
      // few-emulator retry on an engine whose full batch stays on the multi-launch schedules (B * NP / 128 >= 2048) would otherwise
      // allocate B packs per block column (4.7 GB at B = 2000, n = 2000)
      HIPCK(hipStreamSynchronize(stream));
      if (dMcCtrl) HIPCK(hipFree(dMcCtrl));
      if (dMcPacks) HIPCK(hipFree(dMcPacks));
      dMcCtrl = nullptr;
      dMcPacks = nullptr;
      mc_slots = nb;
      mc_ctrl_ints = mchol_ctrl_ints(NP, mc_slots);
      dMcCtrl = dalloc<unsigned>(mc_ctrl_ints);
      dMcPacks = dalloc<double>(mchol_pack_doubles(NP, mc_slots));
    }
    // (the info words and the kernel's control words are cleared by the K build: two memset commands less in front of a small fit)
    ZeroRanges zr;
    zr.p[0] = reinterpret_cast<unsigned*>(dInfo); zr.n[0] = (unsigned)B;
    zr.p[1] = dMcCtrl; zr.n[1] = (unsigned)mchol_ctrl_ints(NP, nb);
    build_cov(v, zr);
    launch_mchol(v, dMcCtrl, mchol_ctrl_ints(NP, nb), dMcTable, mc_ntasks, dMcPacks, dInfo, n_cu, stream, true);
    if (!defer_info) {
      read_info(info, false);
      unsigned aborted = 0;
      HIPCK(hipMemcpy(&aborted, dMcCtrl, sizeof(unsigned), hipMemcpyDeviceToHost));
      if (aborted) {
        g_mc_aborts += 1;
        FlagGuard legacy_only(mc_force_legacy);          // reset also when the repeat throws
        factorize_blocked(ids, info, false);
      }
    }
    return;
  }
  if (schedule == 3) {
    // LEFT-LOOKING WITH LOOK-AHEAD.  Block column c receives the panels 0 .. c-2 in one long-K MFMA pass U1(c) on the main
    // stream -- every element of the trailing matrix is read-modified-written once, at the K depth where the MFMA main
    // loop runs best -- WHILE the panel stream works on block column c-1:
    //     panel stream (high priority):  U2(c): column c -= panel c-1 (K = 128)  ->  128 x 128 diagonal block  ->  panel solve
    //     main stream:                   U1(c+2): column c+2 -= panels 0 .. c    (needs the panel solve of column c)
    // The whole dependent chain of a block column (short update, diagonal block, panel solve) sits in ONE stream: a
    // cross-stream event wait costs ~12 us on this stack when the waiter is already blocked (kernel trace), and the
    // earlier schedules paid two of them per block column.  The main stream is one block column ahead, so its events
    // have normally fired by the time the panel stream asks.  Replaces the two-emulator-group schedule (5.37 ms at
    // 64 x n=2000), the right-looking schedule of small batches and of a single large matrix.
    std::vector<int> cols;
    for (int o = 0; o < n + R; o += TILE) cols.push_back(o);
    const int K = (int)cols.size();
    while ((int)evPanel.size() < K + 1) {
      hipEvent_t a, b;
      HIPCK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
      HIPCK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
      evPanel.push_back(a);
      evUpd.push_back(b);
    }
    constexpr long tail_threshold = 1100L;
    auto long_update = [&](int o, int k1, hipStream_t st) {
      // 64 x 64 tiles unless the launch has several rounds of 128 x 128 ones (measured 7.6 vs 8.3 ms at 64 x n=2000)
      if ((long)nb * ((NP - o) / TILE) >= tail_threshold) launch_update_wide(v, o, 0, k1, st);
      else launch_update_narrow_pair(v, o, 0, k1, st);
    };
    hipStream_t pst = ovr.single_stream ? stream : pstream;
    HIPCK(hipMemsetAsync(dInfo, 0, B * sizeof(int), stream));
    build_cov(v);
    HIPCK(hipEventRecord(evReady, stream));
    HIPCK(hipStreamWaitEvent(pst, evReady, 0));
    // "U1(c) done" in front of U2(c) sits in the dependent chain although U1(c) has normally finished a block column earlier, and
    // an event wait costs the panel stream ~11 us even then.  As a stream memory operation on one signal word (the main stream
    // writes base + c behind U1(c), the panel stream waits for >= base + c) a satisfied wait is a memory poll: fit 1.62 -> 1.57 ms
    // at 8 x n=2000, 2.05 -> 1.94 at 16, 3.05 -> 2.94 at 32, 1.19 -> 1.14 at 64 x n=1000.  A waiter that really has to wait is
    // served later by the poll than by the event (n = 5000: 7.7 -> 8.0 ms at 4 emulators; the right-looking schedule, whose
    // waits are all of that kind: 5.2 -> 5.5 ms at 2 x n=5000, 34.3 -> 35.3 at n=16000; the other direction, panel -> U1, too),
    // so it is used up to NP = 3072.
    const bool wv = can_waitval && !ovr.single_stream && NP <= 3072;
    if (wv && !sigU1) {
      HIPCK(hipExtMallocWithFlags(reinterpret_cast<void**>(&sigU1), 8, hipMallocSignalMemory));
      HIPCK(hipMemset(sigU1, 0, 8));
    }
    if (wv && sig_epoch > 0xF0000000u) {      // the compare is >=: start over long before the counter wraps
      HIPCK(hipStreamSynchronize(stream));
      HIPCK(hipStreamSynchronize(pst));
      HIPCK(hipMemset(sigU1, 0, 8));
      sig_epoch = 1;
    }
    const uint32_t sig_base = sig_epoch;
    if (wv) sig_epoch += (uint32_t)K + 1;
    for (int c = 0; c < K; ++c) {
      const int o = cols[c];
      if (c >= 1) {
        if (c >= 2) {
          if (wv) HIPCK(hipStreamWaitValue32(pst, sigU1, sig_base + (uint32_t)c, hipStreamWaitValueGte, 0xFFFFFFFFu));
          else HIPCK(hipStreamWaitEvent(pst, evUpd[c], 0));                     // U1(c) done
        }
        launch_update_narrow_pair(v, o, o - TILE, o, pst);                      // U2(c): panel c-1 -> column c
      }
      panel(v, o, TILE, pst);
      HIPCK(hipEventRecord(evPanel[c], pst));
      if (c + 2 < K) {
        HIPCK(hipStreamWaitEvent(stream, evPanel[c], 0));
        long_update(cols[c + 2], cols[c + 1], stream);                          // U1(c+2): panels 0 .. c -> column c+2
        if (wv) HIPCK(hipStreamWriteValue32(stream, sigU1, sig_base + (uint32_t)(c + 2), 0));
        else HIPCK(hipEventRecord(evUpd[c + 2], stream));
      }
    }
    HIPCK(hipStreamWaitEvent(stream, evPanel[K - 1], 0));
    read_info(info, defer_info);
    return;
  }
  if (schedule == 0) {
    // Two independent emulator groups on separate streams: while one group runs its
    // latency-bound panel kernels (diagonal block / panel solve: few workgroups) the other group's MFMA update fills the
    // machine.  More than two streams collapse (round 1, 64 x n=2000: 1 group 6.58 ms, 2 groups 6.23 ms, 3 groups 8.3 ms,
    // 4 groups 14.2 ms -- the same when replayed from a captured hipGraph, so it is not host launch overhead).
    constexpr long tail_threshold = 1100L;
    const int G = ovr.single_stream ? 1 : std::min(2, std::max(1, nb / 8));
    while ((int)gstreams.size() < G - 1) {
      hipStream_t st;
      HIPCK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      gstreams.push_back(st);
    }
    HIPCK(hipMemsetAsync(dInfo, 0, B * sizeof(int), stream));
    build_cov(v);
    HIPCK(hipEventRecord(evReady, stream));
    std::vector<BatchView> gv(G, v);
    std::vector<hipStream_t> gs(G, stream);
    for (int g = 0; g < G; ++g) {
      const int lo = (int)((long)nb * g / G), hi = (int)((long)nb * (g + 1) / G);
      gv[g].idx = dIdx + lo;
      gv[g].nb = hi - lo;
      if (g > 0) {
        gs[g] = gstreams[g - 1];
        HIPCK(hipStreamWaitEvent(gs[g], evReady, 0));
      }
    }
    for (int o = 0; o < n + R; o += TILE)
      for (int g = 0; g < G; ++g) {
        if (o > 0) {
          // with fewer than ~4 128-tiles per CU (always true at n=2000 x 64, measured 7.6 vs 8.3 ms) use
          // 64x64 tiles: 4x the workgroups, 3 resident per CU, better balance and latency hiding; both 64-wide
          // halves of the block column go in one launch (7.56 -> 6.84 ms: the partially filled last round of
          // workgroups is paid once instead of twice)
          const long tiles128 = (long)gv[g].nb * ((NP - o) / TILE);
          if (tiles128 >= tail_threshold) launch_update_wide(gv[g], o, 0, o, gs[g]);
          else launch_update_narrow_pair(gv[g], o, 0, o, gs[g]);
        }
        panel(gv[g], o, TILE, gs[g]);
      }
    for (int g = 1; g < G; ++g) {
      HIPCK(hipEventRecord(evGroup[g - 1], gs[g]));
      HIPCK(hipStreamWaitEvent(stream, evGroup[g - 1], 0));
    }
    read_info(info, defer_info);
    return;
  }
  HIPCK(hipMemsetAsync(dInfo, 0, B * sizeof(int), stream));
  build_cov(v);
  std::vector<int> starts;
  // outer block = K depth of the trailing update (256 / 512 / 1024 -> C5 fit 45.7 / 41.7 / 44.3 ms, round 1)
  constexpr int OUTERW = 512;
  for (int o = 0; o < n + R; o += OUTERW) starts.push_back(o);
  const int K = (int)starts.size();
  while ((int)evPanel.size() < K + 1) {
    hipEvent_t a, b;
    HIPCK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
    evPanel.push_back(a);
    evUpd.push_back(b);
  }
  auto width = [&](int k) { return std::min(OUTERW, NP - starts[k]); };
  HIPCK(hipEventRecord(evUpd[K], stream));                 // K build done
  HIPCK(hipStreamWaitEvent(pstream, evUpd[K], 0));
  panel(v, starts[0], width(0), pstream);
  HIPCK(hipEventRecord(evPanel[0], pstream));
  for (int k = 0; k < K; ++k) {
    const int o = starts[k], w = width(k);
    HIPCK(hipStreamWaitEvent(stream, evPanel[k], 0));
    if (k + 1 < K) {
      const int on = starts[k + 1], wn = width(k + 1);
      for (int c = on; c < on + wn; c += TILE) update_column_block(v, c, o, o + w, stream);     // U_a
      HIPCK(hipEventRecord(evUpd[k], stream));
      HIPCK(hipStreamWaitEvent(pstream, evUpd[k], 0));
      panel(v, on, wn, pstream);
      HIPCK(hipEventRecord(evPanel[k + 1], pstream));
      launch_update_trailing(v, on + wn, o, o + w, stream);                                     // U_b
    } else {
      launch_update_trailing(v, o + w, o, o + w, stream);   // (empty unless padding rows remain)
    }
  }
  read_info(info, defer_info);
}

void Engine::eval(const std::vector<int>& ids, const std::vector<const double*>& thetas, bool want_grad, double* f, double* grad,
                  int grad_ld, int* ok) {
  const int nb = (int)ids.size();
  if (nb == 0) return;
  g_obj_evals += nb;
  if (want_grad) g_grad_evals += nb;
  for (int k = 0; k < nb; ++k) {
    set_theta(ids[k], thetas[k]);
    GPState& g = gp[ids[k]];
    g.nugget_used = (g.nug_type == NUG_FIXED) ? g.nug_size : (g.nug_type == NUG_FIT ? std::exp(g.data[NC + 1]) : 0.0);
  }
  // One synchronisation per evaluation: the factorisation leaves its status words on the device, log-det / Gram / alpha
  // (and L^-1 on the gradient path) are launched right behind it, and everything is read back together.  An emulator
  // whose factorisation failed has produced garbage there -- it is either retried (adaptive jitter) or reported.
  std::vector<double> logdet(B, 0.), gram((size_t)B * RMAX * RMAX, 0.);
  auto after_factor = [&](const std::vector<int>& list, std::vector<int>* info_out) {
    for (int i : list) gp[i].factored = true;                // provisional (ensure_linv checks it)
    if (info_out) info_out->assign(B, 0);
    // second pass (rare): emulators whose one-launch back substitution gave up waiting are solved again with the
    // multi-launch path, which has no inter-workgroup waits
    std::vector<int> todo(list);
    for (int pass = 0; pass < 2 && !todo.empty(); ++pass) {
      upload_idx(todo);
      BatchView v = view((int)todo.size());
      bool chained = false, res_done = false;
      // single right-hand side: the one-launch chain (MOGP_BACKSOLVE=1: per-block launches, the path a timed-out chain falls back to)
      static const bool chain = [] { const char* e = getenv("MOGP_BACKSOLVE"); return !e; }();
      const bool use_chain = chain && R == 1 && pass == 0;
      // the chain on stream `st` (flags and sentinel rows prepared on the same stream, in front of it)
      auto launch_chain = [&](hipStream_t st) {
        const size_t nfl = (size_t)B * ((n + 127) / 128);
        if (!dBsFlags) {
          dBsFlags = dalloc<int>(nfl + B);                 // flags, then one status word per emulator
          HIPCK(hipMemsetAsync(dBsFlags, 0, (nfl + B) * sizeof(int), st));
        }
        if (bs_epoch > 0x7FFFFF00) {                       // flags and status are compared with the epoch: start over before it wraps
          HIPCK(hipMemsetAsync(dBsFlags, 0, (nfl + B) * sizeof(int), st));
          bs_epoch = 0;
        }
        if (z_armed.size() != (size_t)B) z_armed.assign(B, 0);
        for (int i : todo) {
          if (!z_armed[i]) HIPCK(hipMemsetAsync(dAlpha + (size_t)i * RA * LD, 0xFF, (size_t)LD * sizeof(double), st));
          z_armed[i] = 0;                                  // consumed by this solve
        }
        res_done = launch_backsolve_chain(v, dBsFlags, ++bs_epoch, dBsFlags + nfl, n_cu, st, dInfo, dRes, mc_used ? dMcCtrl : nullptr);
        chained = true;
      };
      if (want_grad) {
        if (use_chain) {
          // Round 6: alpha by the SAME one-launch back substitution as a plain fit -- on the panel stream, UNDER the triangular inversion
          // (both only read the factor): the gemv with L^-1 that used to follow the inversion (0.25 ms for 64 x n=2000, HBM-bound) and the
          // logdet launch leave the path, and eval(grad=True) returns bit for bit the alpha, log-determinant and log-posterior of eval(grad=False).
          // fit + gradient, gemv behind the inversion / chain under it: 64 x n=2000 10.87 -> 10.81 ms, 8 x 1.751 -> 1.711, one matrix 0.918 -> 0.889.
          HIPCK(hipEventRecord(evReady, stream));            // the factor and the index list are in place
          HIPCK(hipStreamWaitEvent(pstream, evReady, 0));
          launch_chain(pstream);
          HIPCK(hipEventRecord(evGroup[14], pstream));
          ensure_linv(todo);                                 // (fresh factors: every listed emulator needs it, so the index list on the device stays as it is)
          HIPCK(hipStreamWaitEvent(stream, evGroup[14], 0));
        } else {
          // L^-1 is needed anyway, so K^-1 [t, H] = L^-T Y is one fully parallel gemv with it
          ensure_linv(todo);
          upload_idx(todo);
          v = view((int)todo.size());
          launch_alpha_from_linv(v, stream);
        }
      } else if (use_chain) {
        launch_chain(stream);
      } else {
        launch_backsolve(v, stream);
      }
      // (after the solves: it also collects the status words)
      if (!res_done) launch_logdet(v, dInfo, dRes, stream, chained ? dBsFlags + (size_t)B * ((n + 127) / 128) : nullptr, bs_epoch, mc_used ? dMcCtrl : nullptr);
      // status words, log-determinants and Gram matrices come back in ONE copy into pinned host memory
      HIPCK(hipMemcpyAsync(hRes, dRes, (size_t)B * RES_STRIDE * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIPCK(hipStreamSynchronize(stream));
      HIPCK(hipGetLastError());
      std::vector<int> again;
      for (int i : todo) {
        const double* r = hRes + (size_t)i * RES_STRIDE;
        if ((int)r[1] == BACKSOLVE_TIMEOUT) {
          again.push_back(i);
          continue;
        }
        logdet[i] = r[0];
        if (info_out) (*info_out)[i] = (int)r[1];
        std::memcpy(gram.data() + (size_t)i * RMAX * RMAX, r + 2, sizeof(double) * RMAX * RMAX);
      }
      g_bs_timeouts += (long long)again.size();
      todo.swap(again);
    }
  };
  std::vector<int> info;
  bool has_pivot = false;
  for (int i : ids) has_pivot = has_pivot || gp[i].nug_type == NUG_PIVOT;
  factorize(ids, info, !has_pivot);                          // (the pivoted path reads its status after every panel anyway)
  after_factor(ids, has_pivot ? nullptr : &info);
  if (!has_pivot) {
    // the one-launch Cholesky gave up on a wait (never observed; the bound exists so that nothing can hang): its abort word
    // came back as the status of every emulator -- factorise again with the multi-launch schedule
    bool aborted = false;
    for (int i : ids) aborted = aborted || info[i] == MCHOL_ABORTED;
    if (aborted) {
      g_mc_aborts += 1;
      for (int i : ids) gp[i].factored = gp[i].linv = gp[i].kinv = false;      // (L^-1 of the unusable factor may have been formed)
      {
        FlagGuard legacy_only(mc_force_legacy);        // reset also when the repeat throws
        factorize(ids, info, true);
      }
      after_factor(ids, &info);
    }
  }
  std::vector<char> good(B, 0);
  std::vector<int> failed;
  for (int i : ids) {
    if (info[i] == 0) good[i] = 1;
    else {
      failed.push_back(i);
      gp[i].factored = gp[i].linv = gp[i].kinv = false;
    }
  }
  // adaptive jitter ladder: linalg/cholesky.py:268-279 -- jitter = mean(diag K) * 1e-6, x10 per try, 5 tries.
  // diag K = sigma^2 k(0) = sigma^2 exactly, so mean(diag K) = sigma^2.
  std::vector<double> jitter(B, 0.);
  {
    std::vector<int> retry, recovered;
    for (int i : failed)
      if (gp[i].nug_type == NUG_ADAPTIVE) {
        jitter[i] = std::exp(gp[i].data[NC]) * 1e-6;
        retry.push_back(i);
      }
    for (int attempt = 0; attempt < 5 && !retry.empty(); ++attempt) {
      std::vector<int> todo;
      for (int i : retry)
        if (std::isfinite(jitter[i])) {
          gp[i].nugget_used = jitter[i];
          todo.push_back(i);
        }
      if (todo.empty()) break;
      factorize(todo, info);
      std::vector<int> still;
      for (int i : todo) {
        if (info[i] == 0) {
          good[i] = 1;
          gp[i].nug_size = jitter[i];
          recovered.push_back(i);
        } else {
          jitter[i] *= 10;
          still.push_back(i);
        }
      }
      retry.swap(still);
    }
    if (!recovered.empty()) after_factor(recovered, nullptr);
  }
  std::vector<int> okids;
  for (int i : ids)
    if (good[i]) okids.push_back(i);
  for (int i : ids) {
    gp[i].factored = good[i] != 0;
    if (!good[i]) gp[i].linv = gp[i].kinv = false;
  }
  std::vector<double> hM;
  if (R > 1) hM.assign((size_t)B * (RMAX + 1) * RMAX, 0.);
  for (int k = 0; k < nb; ++k) {
    const int i = ids[k];
    GPState& g = gp[i];
    double val = std::numeric_limits<double>::quiet_NaN();
    bool fine = good[i];
    int n_coeff = n;
    if (fine) {
      const double* G = gram.data() + (size_t)i * RMAX * RMAX;
      double quad = G[0], logdetA = 0.;
      if (R > 1) {
        // Analytic mean with priors beta ~ N(b, B) (weak: B^-1 = 0, b = 0), from the Gram matrix G = [t,H]^T K^-1 [t,H]:
        //   A = H^T K^-1 H + B^-1 (calc_Ainv, linalg_utils.py:5-40),  r = H^T K^-1 (t - H b),
        //   beta_hat = A^-1 (r + B^-1 b) (calc_mean_params, :88-121),  quadratic form (t-Hb)^T K^-1 (t-Hb) - r^T A^-1 r
        const bool weak = g.mp_b.empty();
        std::vector<double> Am((size_t)q * q), rv(q), bb(q, 0.);
        if (!weak) bb = g.mp_b;
        for (int r = 0; r < q; ++r) {
          double s = G[(1 + r) * RMAX];
          for (int c = 0; c < q; ++c) {
            s -= G[(1 + r) * RMAX + (1 + c)] * bb[c];
            Am[r * q + c] = G[(1 + r) * RMAX + (1 + c)] + (weak ? 0. : g.mp_Binv[r * q + c]);
          }
          rv[r] = s;
        }
        if (!weak) {
          double bSb = 0., bv = 0.;
          for (int r = 0; r < q; ++r) {
            bv += bb[r] * G[(1 + r) * RMAX];
            for (int c = 0; c < q; ++c) bSb += bb[r] * G[(1 + r) * RMAX + (1 + c)] * bb[c];
          }
          quad = G[0] - 2. * bv + bSb;
        }
        g.LA.assign((size_t)q * q, 0.);
        for (int r = 0; r < q && fine; ++r)
          for (int c = 0; c <= r; ++c) {
            double s = Am[r * q + c];
            for (int p = 0; p < c; ++p) s -= g.LA[r * q + p] * g.LA[c * q + p];
            if (r == c) {
              if (!(s > 0.)) { fine = false; break; }
              g.LA[r * q + r] = std::sqrt(s);
            } else {
              g.LA[r * q + c] = s / g.LA[c * q + c];
            }
          }
        if (fine) {
          auto solveA = [&](std::vector<double> x) {          // A^-1 x by the two triangular solves with LA
            for (int r = 0; r < q; ++r) {
              double s = x[r];
              for (int p = 0; p < r; ++p) s -= g.LA[r * q + p] * x[p];
              x[r] = s / g.LA[r * q + r];
            }
            for (int r = q - 1; r >= 0; --r) {
              double s = x[r];
              for (int p = r + 1; p < q; ++p) s -= g.LA[p * q + r] * x[p];
              x[r] = s / g.LA[r * q + r];
            }
            return x;
          };
          std::vector<double> w(q), Linv((size_t)q * q, 0.);
          for (int r = 0; r < q; ++r) {               // w = LA^-1 r
            double s = rv[r];
            for (int p = 0; p < r; ++p) s -= g.LA[r * q + p] * w[p];
            w[r] = s / g.LA[r * q + r];
            quad -= w[r] * w[r];
            logdetA += 2. * std::log(g.LA[r * q + r]);
          }
          const std::vector<double> bgrad = solveA(rv);           // beta' = A^-1 r: residual of the gradient's quadratic form
          std::vector<double> rhs(rv);
          if (!weak)
            for (int r = 0; r < q; ++r) rhs[r] += g.mp_Binvb[r];
          g.beta = solveA(rhs);
          for (int c = 0; c < q; ++c) {               // LA^-1 (lower), column by column
            for (int r = c; r < q; ++r) {
              double s = (r == c) ? 1. : 0.;
              for (int p = c; p < r; ++p) s -= g.LA[r * q + p] * Linv[p * q + c];
              Linv[r * q + c] = s / g.LA[r * q + r];
            }
          }
          // combination matrix over Z = K^-1 [t, h_1..h_q]:
          //   row 0 -> K^-1 (t - H beta_hat) (predictions);  row c -> g_c = sum_d (LA^-1)[c][d] K^-1 h_d  (d log|A|);
          //   row R -> K^-1 (t - H (b + beta')) (gradient of the quadratic form; = row 0 with weak priors)
          double* M = hM.data() + (size_t)i * (RMAX + 1) * RMAX;
          M[0] = 1.;
          M[R * RMAX] = 1.;
          for (int c = 0; c < q; ++c) {
            M[1 + c] = -g.beta[c];
            M[R * RMAX + 1 + c] = -(bb[c] + bgrad[c]);
          }
          for (int c = 0; c < q; ++c)
            for (int d = 0; d <= c; ++d) M[(1 + c) * RMAX + (1 + d)] = Linv[c * q + d];
          if (weak) n_coeff = n - q;                  // GaussianProcess.py:674-677
          else logdetA += g.mp_logdetB;               // + log|B| (priors.mean.logdet_cov)
        }
      }
      if (fine) {
        // GaussianProcess.py:679-685; densegp_gpu.hpp:604-611
        val = 0.5 * (quad + logdet[i] + logdetA + n_coeff * std::log(2.0 * M_PI)) - g.pri.logp(g.data, NC, g.nug_type);
        if (!std::isfinite(val)) fine = false;
      }
    }
    // nugget="pivot": the reference never fails a fit whose pivoted factorisation exists; with many skipped rows the
    // log-posterior over the tiny replacement diagonal is inf / nan there as well, and the emulator is left "fit" with
    // that value (garbage in, garbage out) instead of raising.  ok[] still reports it, so the optimiser steps away.
    const bool pivot_kept = good[i] && g.nug_type == NUG_PIVOT;
    g.has_data = fine || pivot_kept;
    g.factored = fine || pivot_kept;
    g.logpost = val;
    g.logpost_stale = false;
    if (f) f[k] = val;
    if (ok) ok[k] = fine ? 1 : 0;
    if (!fine) good[i] = 0;
  }
  if (R > 1 && !okids.empty()) {
    HIPCK(hipMemcpyAsync(dM, hM.data(), hM.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    upload_idx(okids);
    launch_combine_rows(view((int)okids.size()), dM, stream);
    HIPCK(hipStreamSynchronize(stream));
  }
  if (want_grad && grad) {
    std::vector<int> gids;
    std::vector<int> pos;
    for (int k = 0; k < nb; ++k)
      if (good[ids[k]]) {
        gids.push_back(ids[k]);
        pos.push_back(k);
      }
    if (!gids.empty()) {
      std::vector<double> tmp((size_t)gids.size() * grad_ld);
      grad_current(gids, tmp.data(), grad_ld);
      for (size_t q = 0; q < gids.size(); ++q)
        std::memcpy(grad + (size_t)pos[q] * grad_ld, tmp.data() + q * grad_ld, sizeof(double) * n_theta(gids[q]));
    }
  }
}

void Engine::fit_one(int i, const double* theta, int len) {
  if (len != n_theta(i)) throw std::runtime_error("Shape of new GPParams object does not match existing one");
  double f;
  int ok;
  std::vector<int> ids{i};
  std::vector<const double*> th{theta};
  eval(ids, th, false, &f, nullptr, 0, &ok);
  if (!ok && !(gp[i].nug_type == NUG_PIVOT && gp[i].factored)) {
    if (gp[i].nug_type == NUG_ADAPTIVE) throw std::runtime_error("All attempts at factorization failed. Last return code 1");
    throw std::runtime_error("Unable to factorize matrix using selected nugget type");
  }
}

void Engine::ensure_linv(const std::vector<int>& ids) {
  std::vector<int> need;
  for (int i : ids) {
    if (!gp[i].factored) throw std::runtime_error("emulator has not been fit");
    if (!gp[i].linv) need.push_back(i);
  }
  if (need.empty()) return;
  if (!dLinv) {
    dLinv = dalloc<double>((size_t)B * MS);
    dKinv = dalloc<double>((size_t)B * MS);
    dGradOut = dalloc<double>((size_t)B * (D + 3));
    dGradPartial = dalloc<double>((size_t)B * grad_num_tiles(n) * (D + 3));
  }
  upload_idx(need);
  BatchView v = view((int)need.size());
  launch_trtri(v, stream);
  for (int i : need) {
    gp[i].linv = true;
    gp[i].kinv = false;   // trtri used Kinv as scratch
  }
}

void Engine::drop_w2(int i) {
  auto it = w2.find(i);
  if (it == w2.end()) return;
  hipFree(it->second);
  w2.erase(it);
}

// for_gradient: an emulator whose pivoted factorisation skipped rows gets K^-1 WITHOUT the rows of L^-1 of the skipped
// pivots (they go to w2 and enter the gradient through launch_grad_lowrank, see kernels_cov.hip); otherwise the full K^-1.
void Engine::ensure_kinv(const std::vector<int>& ids, bool for_gradient) {
  ensure_linv(ids);
  std::vector<int> need;
  for (int i : ids) {
    const bool deficient = gp[i].permuted && gp[i].rank < n;
    const bool want_split = for_gradient && deficient;
    if (!gp[i].kinv || gp[i].kinv_split != want_split) need.push_back(i);
  }
  if (need.empty()) return;
  const size_t rowb = (size_t)LD * sizeof(double), wb = (size_t)n * sizeof(double);
  std::vector<int> split;
  for (int i : need)
    if (for_gradient && gp[i].permuted && gp[i].rank < n) {
      const int m = n - gp[i].rank;
      drop_w2(i);
      double* buf = dalloc<double>((size_t)m * LD);
      w2[i] = buf;
      double* rows = dLinv + (size_t)i * MS + (size_t)gp[i].rank * LD;
      HIPCK(hipMemcpy2DAsync(buf, rowb, rows, rowb, wb, m, hipMemcpyDeviceToDevice, stream));
      HIPCK(hipMemset2DAsync(rows, rowb, 0, wb, m, stream));
      split.push_back(i);
    }
  upload_idx(need);
  BatchView v = view((int)need.size());
  launch_kinv(v, n_cu, stream);
  for (int i : split) {
    double* rows = dLinv + (size_t)i * MS + (size_t)gp[i].rank * LD;
    HIPCK(hipMemcpy2DAsync(rows, rowb, w2[i], rowb, wb, n - gp[i].rank, hipMemcpyDeviceToDevice, stream));
  }
  for (int i : need) {
    gp[i].kinv = true;
    gp[i].kinv_split = false;
  }
  for (int i : split) gp[i].kinv_split = true;
}

void Engine::grad_current(const std::vector<int>& ids, double* grad, int grad_ld) {
  if (ids.empty()) return;
  ensure_kinv(ids, true);
  upload_idx(ids);
  BatchView v = view((int)ids.size());
  launch_grad(v, dGradPartial, dGradOut, stream);
  for (int i : ids)
    if (gp[i].kinv_split) {
      const int m = n - gp[i].rank;
      double* part = dalloc<double>((size_t)m * ((n + 127) / 128) * (D + 1));
      launch_grad_lowrank(v, i, w2.at(i), m, part, dGradOut, stream);
      HIPCK(hipStreamSynchronize(stream));
      hipFree(part);
    }
  const int NQ = D + 3;
  std::vector<double> out((size_t)B * NQ);
  HIPCK(hipMemcpyAsync(out.data(), dGradOut, out.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIPCK(hipStreamSynchronize(stream));
  HIPCK(hipGetLastError());
  const int nm = n_mean();
  std::vector<double> dpr(D + 2);
  for (size_t q = 0; q < ids.size(); ++q) {
    const int i = ids[q];
    const GPState& g = gp[i];
    double* gr = grad + q * grad_ld;
    const double* o = out.data() + (size_t)i * NQ;
    g.pri.dlogpdtheta(g.data, NC, g.nug_type, dpr.data());
    // GaussianProcess.py:751-780: 0.5 sum W dK/dtheta_p  -  d log prior / d theta_p
    if (uniform()) {
      // one shared length scale: dr2/dtheta_0 = r2 (Kernel.py:338-376) = the sum of the per-dimension terms
      double s = 0.;
      for (int p = 0; p < D; ++p) s += o[p];
      gr[nm] = s - dpr[0];
    } else {
      for (int p = 0; p < D; ++p) gr[nm + p] = o[p] - dpr[p];
    }
    gr[nm + NC] = o[D] - dpr[NC];
    if (g.nug_type == NUG_FIT) gr[nm + NC + 1] = 0.5 * std::exp(g.data[NC + 1]) * (o[D + 1] - o[D + 2]) - dpr[NC + 1];
    if (nm > 0) {
      // densegp_gpu.hpp:734-747: -(d mean / d beta)^T alpha
      std::vector<double> a(n), md((size_t)nm * n);
      HIPCK(hipMemcpy(a.data(), dAlpha + (size_t)i * RA * LD, n * sizeof(double), hipMemcpyDeviceToHost));
      unpermute(i, a.data());              // the mean derivatives below are in training order
      mean.mean_deriv(hX.data(), n, D, g.meanp.data(), nm, md.data());
      for (int p = 0; p < nm; ++p) {
        double s = 0.;
        for (int k = 0; k < n; ++k) s += md[(size_t)p * n + k] * a[k];
        gr[p] = -s;
      }
    }
  }
}

void Engine::ensure_predict_scratch(int nb, int MC) {
  grow(dKs, capKs, (size_t)nb * MC * LD);
  // partial sums per row tile
  const size_t nti = (n + 127) / 128;
  grow(dVarPartial, capVarPartial, (size_t)nb * nti * MC);
}

void Engine::predict(const std::vector<int>& ids, const double* Xs, int m, bool xs_on_device, double* means, double* vars, long out_ld,
                     bool out_on_device, double* derivs) {
  const int nb = (int)ids.size();
  if (nb == 0 || m == 0) return;
  for (int i : ids)
    if (!gp[i].factored) throw std::runtime_error("emulator has not been fit");
  if (vars) ensure_linv(ids);
  upload_idx(ids);
  BatchView v = view(nb);
  const double* dXsrc = Xs;
  if (!xs_on_device) {
    grow(dXs, capXs, (size_t)m * D);
    HIPCK(hipMemcpyAsync(dXs, Xs, (size_t)m * D * sizeof(double), hipMemcpyHostToDevice, stream));
    dXsrc = dXs;
  }
  // dots: R rows of dot products per emulator (row 0 = k*^T K^-1 (t - H beta), rows 1.. = k*^T K^-1 h_c); with R = 1 that row IS
  // the mean (before the mean-function term) and lands in the result rows directly.  fm / fv / fd: where the finished means /
  // variances / derivatives live on the device -- the caller's buffers (out_on_device) or the engine's, copied out at the end.
  double* fm = means;
  double* fv = vars;
  double* fd = derivs;
  long ld = out_ld;
  const bool want_mean = means != nullptr;       // derivatives only (mogp_*_predict_deriv): no cross covariance, no mean
  if (!want_mean && vars) throw std::runtime_error("predict: variances without means");
  if (!out_on_device && want_mean) {
    grow(dMeanFin, capMeanFin, (size_t)nb * m);
    fm = dMeanFin;
    ld = m;
    if (vars) {
      grow(dVar, capVar, (size_t)nb * m);
      fv = dVar;
    }
  }
  if (!out_on_device && derivs) {
    grow(dDeriv, capDeriv, (size_t)nb * m * D);
    fd = dDeriv;
  }
  double* dots = fm;
  long dots_ld = ld;
  if (R > 1 && want_mean) {
    grow(dMean, capMean, (size_t)nb * R * m);
    dots = dMean;
    dots_ld = m;
  }
  const int MPtot = roundup(m, 128);
  static const double budget = [] { const char* e = getenv("MOGP_KS_BUDGET_GB"); return (e ? atof(e) : 12.0) * 1e9; }();  // cross-covariance chunk (12 GB: one chunk for 64 x n=2000 x m=10^4)
  // never more than half of what the device has free right now (several engines / ranks per GPU, smaller devices);
  // what is already allocated for the chunk counts as free
  double cap = budget;
  if (vars) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) cap = std::min(cap, 0.5 * ((double)free_b + (double)capKs * sizeof(double)));
  }
  long MC = (long)(cap / ((double)nb * LD * 8.0)) / 128 * 128;
  MC = std::max<long>(128, std::min<long>(MC, MPtot));
  if (vars) ensure_predict_scratch(nb, (int)MC);
  for (int c0 = 0; want_mean && c0 < m; c0 += (int)MC) {
    const int mc = std::min<int>((int)MC, m - c0);
    const int MPc = roundup(mc, 128);
    launch_cross_cov_mean(v, dXsrc + (size_t)c0 * D, mc, MPc, vars ? dKs : nullptr, dots + c0, (int)dots_ld, stream);
    if (vars) launch_predict_var(v, dKs, mc, MPc, dVarPartial, fv + c0, (int)ld, n_cu, stream);
  }
  if (derivs) launch_predict_deriv(v, dXsrc, m, fd, (long)m * D, stream);
  if (mean.kind != 0 || R > 1) {
    // mean-function terms, on the device: the basis columns from the device-resident test points (mean_basis_kernel; round 6 -- the host used to
    // evaluate them, with a read-back of the points when they were handed over in device memory), then predict_mean_finish_kernel
    const int nterm = (mean.kind == 3) ? (int)mean.dims.size() : 0;
    const int nbasis = 1 + nterm;
    const int qq = R - 1;
    if (R > 1 && qq != nbasis) throw std::runtime_error("predict: analytic mean with an unexpected number of columns");
    // device block: basis | dbasis | coef | LA | (ints) dims, powers -- the first two filled on the device, the rest staged from the host
    const size_t o_basis = 0, o_dbasis = o_basis + (size_t)nbasis * m, o_coef = o_dbasis + (size_t)nterm * m,
                 o_la = o_coef + (size_t)nb * nbasis, o_int = o_la + (size_t)nb * qq * qq, total = o_int + (size_t)nterm + 1;
    std::vector<double> st(total - o_coef, 0.);
    for (int k = 0; k < nb; ++k) {
      const GPState& g = gp[ids[k]];
      double* c = st.data() + (size_t)k * nbasis;
      if (R > 1) {
        for (int t = 0; t < qq; ++t) c[t] = g.beta[t];
        for (int e = 0; e < qq * qq; ++e) st[(o_la - o_coef) + (size_t)k * qq * qq + e] = g.LA[e];
      } else if (mean.kind == 1) c[0] = mean.value;
      else
        for (int t = 0; t < nbasis; ++t) c[t] = g.meanp[t];
    }
    // (ints packed behind the doubles of the same staging block: copied in, not written through a punned pointer -- ADVICE r5)
    std::vector<int> hi(2 * (size_t)nterm + 2, 0);
    for (int t = 0; t < nterm; ++t) {
      hi[t] = mean.dims[t];
      hi[nterm + t] = mean.powers[t];
    }
    std::memcpy(st.data() + (o_int - o_coef), hi.data(), 2 * (size_t)nterm * sizeof(int));
    grow(dMeanAux, capMeanAux, total);
    HIPCK(hipMemcpyAsync(dMeanAux + o_coef, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    const int* di = reinterpret_cast<const int*>(dMeanAux + o_int);
    launch_mean_basis(dXsrc, m, D, nterm, di, di + nterm, dMeanAux + o_basis, dMeanAux + o_dbasis, stream);
    launch_predict_mean_finish(nb, m, D, R, nbasis, dMeanAux + o_basis, dMeanAux + o_coef, R > 1 ? dots : nullptr, dMeanAux + o_la, fm,
                               (R > 1 && vars) ? fv : nullptr, ld, derivs ? nterm : 0, dMeanAux + o_dbasis, di, di + nterm, fd, stream);
    HIPCK(hipStreamSynchronize(stream));      // `st` is the source of an asynchronous copy
  }
  if (!out_on_device) {
    if (!want_mean) {
    } else if (out_ld == m) {        // contiguous result arrays: one transfer each instead of one per emulator
      HIPCK(hipMemcpyAsync(means, fm, (size_t)nb * m * sizeof(double), hipMemcpyDeviceToHost, stream));
      if (vars) HIPCK(hipMemcpyAsync(vars, fv, (size_t)nb * m * sizeof(double), hipMemcpyDeviceToHost, stream));
    } else {
      for (int k = 0; k < nb; ++k) {
        HIPCK(hipMemcpyAsync(means + (size_t)k * out_ld, fm + (size_t)k * m, m * sizeof(double), hipMemcpyDeviceToHost, stream));
        if (vars) HIPCK(hipMemcpyAsync(vars + (size_t)k * out_ld, fv + (size_t)k * m, m * sizeof(double), hipMemcpyDeviceToHost, stream));
      }
    }
    if (derivs) HIPCK(hipMemcpyAsync(derivs, fd, (size_t)nb * m * D * sizeof(double), hipMemcpyDeviceToHost, stream));
  }
  HIPCK(hipStreamSynchronize(stream));
  HIPCK(hipGetLastError());
}

void Engine::predict_full_cov(const std::vector<int>& ids, const double* Xs, int m, double* means, double* covs) {
  const int nb = (int)ids.size();
  if (nb == 0 || m == 0) return;
  for (int i : ids)
    if (!gp[i].factored) throw std::runtime_error("emulator has not been fit");
  const int MP = roundup(m, 128);
  const double need = (double)nb * 8.0 * ((double)LD * MP * 2.0 + (double)m * m);
  if (need > 64.0e9)
    throw std::runtime_error("full_cov: " + std::to_string(nb) + " x " + std::to_string(m) +
                             " test points need more than 64 GB of device scratch; use fewer points per call");
  ensure_linv(ids);
  upload_idx(ids);
  BatchView v = view(nb);
  double *dXf = nullptr, *dKf = nullptr, *dV = nullptr, *dC = nullptr, *dDots = nullptr;
  try {
    dXf = dalloc<double>((size_t)m * D);
    dKf = dalloc<double>((size_t)nb * MP * LD);
    dV = dalloc<double>((size_t)nb * NP * MP);
    dC = dalloc<double>((size_t)nb * m * m);
    dDots = dalloc<double>((size_t)nb * R * m);
    HIPCK(hipMemcpyAsync(dXf, Xs, (size_t)m * D * sizeof(double), hipMemcpyHostToDevice, stream));
    launch_cross_cov_mean(v, dXf, m, MP, dKf, dDots, m, stream);
    launch_cov_self_batch(v, dXf, m, dC, stream);
    launch_predict_fullcov(v, dKf, m, MP, dV, dC, stream);
    std::vector<double> dots((size_t)nb * R * m);
    HIPCK(hipMemcpyAsync(dots.data(), dDots, dots.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCK(hipMemcpyAsync(covs, dC, (size_t)nb * m * m * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCK(hipStreamSynchronize(stream));
    HIPCK(hipGetLastError());
    std::vector<double> mv(m), Hs((size_t)q * m), rm((size_t)q * m), dummy(std::max(q, 1), 0.);
    if (R > 1) mean.mean_deriv(Xs, m, D, dummy.data(), q, Hs.data());
    for (int k = 0; k < nb; ++k) {
      const GPState& g = gp[ids[k]];
      const double* dk = dots.data() + (size_t)k * R * m;
      double* mu = means + (size_t)k * m;
      for (int j = 0; j < m; ++j) mu[j] = dk[j];
      if (R > 1) {
        // + h(x*)^T beta and + (LA^-1 R)^T (LA^-1 R), R = H*^T - H^T K^-1 k*   (calc_R, linalg_utils.py:123-168)
        for (int j = 0; j < m; ++j) {
          for (int c = 0; c < q; ++c) {
            mu[j] += g.beta[c] * Hs[(size_t)c * m + j];
            double s = Hs[(size_t)c * m + j] - dk[(size_t)(1 + c) * m + j];
            for (int p = 0; p < c; ++p) s -= g.LA[c * q + p] * rm[(size_t)p * m + j];
            rm[(size_t)c * m + j] = s / g.LA[c * q + c];
          }
        }
        double* Ck = covs + (size_t)k * m * m;
        for (int i = 0; i < m; ++i)
          for (int c = 0; c < q; ++c) {
            const double ri = rm[(size_t)c * m + i];
            const double* rc = rm.data() + (size_t)c * m;
            double* row = Ck + (size_t)i * m;
            for (int j = 0; j < m; ++j) row[j] += ri * rc[j];
          }
      } else if (mean.kind != 0) {
        mean.mean_f(Xs, m, D, g.meanp.data(), n_mean(), mv.data());
        for (int j = 0; j < m; ++j) mu[j] += mv[j];
      }
    }
  } catch (...) {
    for (double* p : {dXf, dKf, dV, dC, dDots}) if (p) hipFree(p);
    throw;
  }
  for (double* p : {dXf, dKf, dV, dC, dDots}) if (p) hipFree(p);
}

void Engine::implausibility(const std::vector<int>& ids, const double* Xs, int m, const double* obs, const double* obs_var,
                            const double* discrepancy, bool include_nugget, int rank, double* out) {
  const int nb = (int)ids.size();
  if (nb == 0 || m == 0) return;
  for (int i : ids)
    if (!gp[i].factored) throw std::runtime_error("emulator has not been fit");
  if (R > 1 || n_mean() > 0)
    throw std::runtime_error("implausibility: the fused device path supports zero / fixed mean functions only");
  if (nb == 1) rank = 0;                                       // HistoryMatching.py:254-255
  if (rank < 0) throw std::runtime_error("rank must be a non-negative integer");
  if (rank >= nb) throw std::runtime_error("rank must be less than the number of observations");
  if (rank > IMPLAUS_MAX_RANK) throw std::runtime_error("rank above " + std::to_string(IMPLAUS_MAX_RANK) + " is not supported on the device");
  std::vector<double> prm((size_t)nb * 3);
  for (int k = 0; k < nb; ++k) {
    if (discrepancy[k] < 0.) throw std::runtime_error("Model discrepancy variance cannot be negative");
    if (obs_var[k] < 0.) throw std::runtime_error("observation variance cannot be negative");
    prm[3 * k] = obs[k];
    prm[3 * k + 1] = obs_var[k] + discrepancy[k] + (include_nugget ? nugget_size(ids[k]) : 0.);
    prm[3 * k + 2] = (mean.kind == 1) ? mean.value : 0.;
  }
  ensure_linv(ids);
  upload_idx(ids);
  BatchView v = view(nb);
  const int MPtot = roundup(m, 128);
  long MC = (long)(6.0e9 / ((double)nb * LD * 8.0)) / 128 * 128;
  MC = std::max<long>(128, std::min<long>(MC, MPtot));
  ensure_predict_scratch(nb, (int)MC);
  grow(dXs, capXs, (size_t)MC * D);
  grow(dMean, capMean, (size_t)nb * MC);
  grow(dVar, capVar, (size_t)nb * MC);
  double* dPrm = dalloc<double>(prm.size());
  double* dOut = dalloc<double>((size_t)MC);
  try {
    HIPCK(hipMemcpyAsync(dPrm, prm.data(), prm.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    for (int c0 = 0; c0 < m; c0 += (int)MC) {
      const int mc = std::min<int>((int)MC, m - c0);
      const int MPc = roundup(mc, 128);
      HIPCK(hipMemcpyAsync(dXs, Xs + (size_t)c0 * D, (size_t)mc * D * sizeof(double), hipMemcpyHostToDevice, stream));
      launch_cross_cov_mean(v, dXs, mc, MPc, dKs, dMean, (int)MC, stream);
      launch_predict_var(v, dKs, mc, MPc, dVarPartial, dVar, (int)MC, n_cu, stream);
      launch_implausibility(nb, dMean, dVar, (int)MC, mc, dPrm, rank, dOut, stream);
      HIPCK(hipMemcpyAsync(out + c0, dOut, (size_t)mc * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIPCK(hipStreamSynchronize(stream));      // dXs is re-used by the next chunk
    }
    HIPCK(hipGetLastError());
  } catch (...) {
    hipFree(dPrm);
    hipFree(dOut);
    throw;
  }
  hipFree(dPrm);
  hipFree(dOut);
}

void Engine::loo_variance(int i, double* out) {
  if (!gp[i].factored) throw std::runtime_error("emulator has not been fit");
  std::vector<int> ids{i};
  ensure_linv(ids);
  upload_idx(ids);
  double* tmp = dalloc<double>((size_t)n);
  launch_loo_variance(view(1), tmp, n, stream);
  HIPCK(hipMemcpyAsync(out, tmp, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIPCK(hipStreamSynchronize(stream));
  hipFree(tmp);
  unpermute(i, out);
}

void Engine::unpermute(int i, double* vec) const {
  if (!gp[i].permuted) return;
  const int* P = hPerm.data() + (size_t)i * n;
  std::vector<double> tmp(vec, vec + n);
  for (int k = 0; k < n; ++k) vec[P[k]] = tmp[k];
}

void Engine::get_pivot(int i, int* perm_out, int* rank_out) {
  if (!gp[i].factored) throw std::runtime_error("emulator has not been fit");
  for (int k = 0; k < n; ++k) perm_out[k] = gp[i].permuted ? hPerm[(size_t)i * n + k] : k;
  if (rank_out) *rank_out = gp[i].permuted ? gp[i].rank : n;
}

void Engine::get_K(int i, double* out) {
  if (!gp[i].has_data) throw std::runtime_error("emulator has not been fit");
  double* tmp = dalloc<double>((size_t)n * n);
  std::vector<int> ids{i};
  upload_params(ids);
  BatchView v = view(1);
  v.X = dX;          // training order, also for a pivoted emulator
  v.XS = 0;
  launch_cov_full(v, i, tmp, stream);
  HIPCK(hipMemcpyAsync(out, tmp, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIPCK(hipStreamSynchronize(stream));
  hipFree(tmp);
}

void Engine::get_invQ(int i, double* out) {
  std::vector<int> ids{i};
  ensure_kinv(ids, false);
  double* tmp = dalloc<double>((size_t)n * n);
  launch_extract(dKinv + (size_t)i * MS, LD, n, tmp, 2, stream);
  HIPCK(hipMemcpyAsync(out, tmp, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIPCK(hipStreamSynchronize(stream));
  hipFree(tmp);
  if (gp[i].permuted) {          // K^-1 of the pivoted matrix back to training order
    const int* P = hPerm.data() + (size_t)i * n;
    std::vector<double> t(out, out + (size_t)n * n);
    for (int a = 0; a < n; ++a)
      for (int b = 0; b < n; ++b) out[(size_t)P[a] * n + P[b]] = t[(size_t)a * n + b];
  }
}

void Engine::get_invQt(int i, double* out) {
  if (!gp[i].factored) throw std::runtime_error("emulator has not been fit");
  HIPCK(hipMemcpy(out, dAlpha + (size_t)i * RA * LD, n * sizeof(double), hipMemcpyDeviceToHost));
  unpermute(i, out);
}

void Engine::pivot_cholesky(const double* Ain, int n, double* L_out, int* P_out, int* rank_out) {
  if (n < 1) throw std::runtime_error("A must have shape (n,n)");
  // _check_cholesky_inputs, linalg/cholesky.py:196-222
  for (int i = 0; i < n; ++i) {
    if (!(Ain[(size_t)i * n + i] > 0.)) throw std::runtime_error("not pd: non-positive diagonal elements");
    for (int j = 0; j < i; ++j) {
      const double a = Ain[(size_t)i * n + j], b = Ain[(size_t)j * n + i];
      if (!(std::fabs(a - b) <= 1e-7 * std::fabs(b))) throw std::runtime_error("A must be symmetric");
    }
  }
  // padded copy: NP x NP with the identity beyond n, so that the blocked trailing update can be used as it is
  const int NPp = roundup(n, TILE);
  std::vector<double> hA((size_t)NPp * NPp, 0.);
  for (int i = 0; i < NPp; ++i) {
    if (i < n) std::memcpy(hA.data() + (size_t)i * NPp, Ain + (size_t)i * n, (size_t)n * sizeof(double));
    else hA[(size_t)i * NPp + i] = 1.0;
  }
  double* dA_ = dalloc<double>(hA.size());
  double* dA0 = dalloc<double>((size_t)n * n);
  double* dW = dalloc<double>(pstrf_work_doubles(NPp));
  int* dI = dalloc<int>((size_t)n + 2);
  BatchView v{};
  v.n = n; v.D = 1; v.NP = NPp; v.LD = NPp; v.MS = (size_t)NPp * NPp; v.PS = 0; v.kernel_type = 0;
  v.A = dA_; v.R = 0; v.RA = 0; v.idx = nullptr; v.nb = 1;
  HIPCK(hipMemcpy(dA_, hA.data(), hA.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCK(hipMemcpy(dA0, Ain, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice));
  launch_pstrf_begin(v, dI, dI + n, dI + n + 1, dW, nullptr);
  int rk = -1;
  for (int k0 = 0; k0 < n; k0 += NBI) {
    const bool first_half = (k0 % TILE) == 0;
    launch_pstrf_panel(v, k0, std::min(NBI, n - k0), dI, dI + n, dW, nullptr);
    HIPCK(hipMemcpy(&rk, dI + n, sizeof(int), hipMemcpyDeviceToHost));
    if (rk >= 0) break;
    if (first_half) launch_update_narrow(v, k0 + NBI, k0, k0 + NBI, nullptr);
    launch_update_trailing(v, first_half ? k0 + TILE : k0 + NBI, k0, k0 + NBI, nullptr);
  }
  launch_pstrf_tail(v, dI, dI + n, nullptr, dA0, nullptr);
  std::vector<int> hi((size_t)n + 2);
  HIPCK(hipMemcpy(hi.data(), dI, hi.size() * sizeof(int), hipMemcpyDeviceToHost));
  HIPCK(hipMemcpy(hA.data(), dA_, hA.size() * sizeof(double), hipMemcpyDeviceToHost));
  hipFree(dA_); hipFree(dA0); hipFree(dW); hipFree(dI);
  for (int i = 0; i < n; ++i) std::memcpy(L_out + (size_t)i * n, hA.data() + (size_t)i * NPp, (size_t)n * sizeof(double));
  if (hi[n + 1] != 0) throw std::runtime_error("not pd: no positive pivot");
  for (int i = 0; i < n; ++i) {
    P_out[i] = hi[i];
    for (int j = i + 1; j < n; ++j) L_out[(size_t)i * n + j] = 0.;
  }
  if (rank_out) *rank_out = hi[n];
}

void Engine::get_chol(int i, double* out) {
  if (!gp[i].factored) throw std::runtime_error("emulator has not been fit");
  double* tmp = dalloc<double>((size_t)n * n);
  launch_extract(dA + (size_t)i * MS, LD, n, tmp, 1, stream);
  HIPCK(hipMemcpyAsync(out, tmp, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIPCK(hipStreamSynchronize(stream));
  hipFree(tmp);
}

// ---------------------------------------------------------------------------------------------
// Multi-start MAP fit (fitting.hpp:61-128, fitting.py:219-266).  Every optimiser "round" is ONE batched device evaluation
// (objective + gradient) of every run that needs one; every run is its own L-BFGS (memory 10) with an Armijo/curvature
// line search and advances independently of the others (slot pool, run_pool below).  Optimiser trajectory parity with
// dlib / scipy is unpinned; the end-point objective is what the tests compare.
// ---------------------------------------------------------------------------------------------
namespace {
struct Lbfgs {
  int np = 0;
  std::vector<double> x, g, d, xt, gt;   // current point/gradient, direction, trial point/gradient
  double f = 0., ft = 0., step = 1., slope = 0.;
  std::vector<std::vector<double>> S, Y;
  std::vector<double> rho;
  int iter = 0, ls_iter = 0;
  enum { NEED_F0, LINESEARCH, DONE, FAILED } state = NEED_F0;
  double step_lo = 0., step_hi = 0.;      // bracketing for the curvature condition

  void direction() {
    d = g;
    const int h = (int)S.size();
    std::vector<double> al(h);
    for (int i = h - 1; i >= 0; --i) {
      double a = 0.;
      for (int k = 0; k < np; ++k) a += S[i][k] * d[k];
      a *= rho[i];
      al[i] = a;
      for (int k = 0; k < np; ++k) d[k] -= a * Y[i][k];
    }
    if (h > 0) {
      double yy = 0., sy = 0.;
      for (int k = 0; k < np; ++k) { yy += Y[h - 1][k] * Y[h - 1][k]; sy += S[h - 1][k] * Y[h - 1][k]; }
      const double gam = sy / yy;
      for (int k = 0; k < np; ++k) d[k] *= gam;
    }
    for (int i = 0; i < h; ++i) {
      double b = 0.;
      for (int k = 0; k < np; ++k) b += Y[i][k] * d[k];
      b *= rho[i];
      for (int k = 0; k < np; ++k) d[k] += S[i][k] * (al[i] - b);
    }
    for (int k = 0; k < np; ++k) d[k] = -d[k];
    slope = 0.;
    for (int k = 0; k < np; ++k) slope += g[k] * d[k];
    if (!(slope < 0.)) {   // not a descent direction: restart from steepest descent
      S.clear(); Y.clear(); rho.clear();
      for (int k = 0; k < np; ++k) d[k] = -g[k];
      slope = 0.;
      for (int k = 0; k < np; ++k) slope -= g[k] * g[k];
    }
  }
  void trial() {
    for (int k = 0; k < np; ++k) xt[k] = x[k] + step * d[k];
  }
};
}  // namespace

// the process-wide replica engine kept between multi-start fits (never destroyed at exit: the HIP runtime may be gone by then)
static std::unique_ptr<Engine>& replica_cache() {
  static std::unique_ptr<Engine>* p = new std::unique_ptr<Engine>();
  return *p;
}
static std::mutex& replica_cache_mutex() {
  static std::mutex* m = new std::mutex();
  return *m;
}

// The optimiser runs of a multi-start fit as a SLOT POOL (round 6).  `slots` are emulators of THIS engine; a slot
// carries one L-BFGS run at a time.  Every round is ONE batched objective (+ gradient) evaluation of the slots whose run needs one; a run
// that ends (converged, out of iterations, failed) hands its slot to the next pending run IN THE SAME ROUND -- `next(pos, x0, tag)` fills
// the slot (and, on a replica engine, gives it the targets and priors of the run's emulator), `done(tag, f, x)` receives the end point
// (f = +inf and x empty for a failed run) -- so every evaluation is full until the queue drains.  The reference runs one optimiser per
// emulator with no coupling at all (mogp_gpu/src/fitting.hpp:122-127 OpenMP over emulators; fitting.py:333-335 Pool.starmap); rounds 1-5
// ran fixed passes of starts in lock-step, each ending on its slowest run (64 x 15 starts: 71 % of the raw fit+gradient rate).
// A run's trajectory depends on its own evaluations only, and those are bit-identical whatever the batch they are part of (one-launch
// Cholesky), so the end points do not depend on the schedule.
void Engine::run_pool(const std::vector<int>& slots, const std::function<bool(int, std::vector<double>&, int&)>& next,
                      const std::function<void(int, double, const std::vector<double>&)>& done) {
  const FitOptions& opt = fit_options();
  const int ns = (int)slots.size();
  std::vector<Lbfgs> st(ns);
  std::vector<int> tag(ns, -1), evals(ns, 0);
  std::vector<char> live(ns, 0);
  const int eval_cap = opt.max_iter * 25;
  auto start_run = [&](int pos) {
    std::vector<double> x0;
    int t = -1;
    if (!next(pos, x0, t)) {
      live[pos] = 0;
      return;
    }
    Lbfgs s;
    s.np = n_theta(slots[pos]);
    if ((int)x0.size() != s.np) throw std::runtime_error("fit_map: starting point of the wrong length");
    s.g.resize(s.np); s.d.resize(s.np); s.gt.resize(s.np);
    s.x = x0;
    s.xt = s.x;
    st[pos] = std::move(s);
    tag[pos] = t;
    evals[pos] = 0;
    live[pos] = 1;
    g_lb_runs += 1;
  };
  auto finish_run = [&](int pos) {
    const Lbfgs& s = st[pos];
    const bool good = !(s.state == Lbfgs::FAILED || s.state == Lbfgs::NEED_F0) && std::isfinite(s.f);
    done(tag[pos], good ? s.f : std::numeric_limits<double>::infinity(), good ? s.x : std::vector<double>());
    start_run(pos);
  };
  for (int pos = 0; pos < ns; ++pos) start_run(pos);
  const double c1 = 1e-4, c2 = 0.9;      // Armijo / weak curvature
  static const int lazy_env = [] { const char* e = getenv("MOGP_LAZY_GRAD"); return e ? atoi(e) : -1; }();
  const bool lazy_grad = lazy_env < 0 ? n >= 512 : lazy_env != 0;
  for (;;) {
    std::vector<int> act, actid;
    std::vector<const double*> th;
    for (int pos = 0; pos < ns; ++pos)
      if (live[pos]) {
        act.push_back(pos);
        actid.push_back(slots[pos]);
        th.push_back(st[pos].xt.data());
      }
    if (act.empty()) break;
    g_pool_rounds += 1;
    g_pool_slot_rounds += (long long)act.size();
    int maxnp = 0;
    for (int pos : act) maxnp = std::max(maxnp, st[pos].np);
    std::vector<double> fv(act.size()), gv(act.size() * (size_t)maxnp);
    std::vector<int> okv(act.size());
    if (lazy_grad) {
      // The objective of every active run first; the gradient (L^-1, K^-1, the fused reduction: 2/3 of an evaluation) only
      // where the optimiser will look at it -- a trial point that fails the sufficient-decrease test is shortened without.
      // 8 - 14 % of the trial points of the benchmark fits fail it: 64 emulators x 15 starts of n = 2000, 10 / 100
      // iterations: 2.58 -> 2.34 s / 8.11 -> 7.42 s, same optima.  The second synchronisation per round costs small
      // problems more than it saves (n = 200, 15 starts: 0.045 -> 0.051 s), so it is used from n = 512 (MOGP_LAZY_GRAD=0 / 1).
      eval(actid, th, false, fv.data(), nullptr, 0, okv.data());
      std::vector<int> gids, gpos;
      for (size_t q = 0; q < act.size(); ++q) {
        const Lbfgs& s = st[act[q]];
        if (!okv[q]) continue;
        if (s.state == Lbfgs::NEED_F0 || fv[q] <= s.f + c1 * s.step * s.slope) {
          gids.push_back(actid[q]);
          gpos.push_back((int)q);
        }
      }
      if (!gids.empty()) {
        g_grad_evals += (long long)gids.size();
        std::vector<double> tmp(gids.size() * (size_t)maxnp);
        grad_current(gids, tmp.data(), maxnp);
        for (size_t k = 0; k < gids.size(); ++k) std::memcpy(gv.data() + (size_t)gpos[k] * maxnp, tmp.data() + k * maxnp, sizeof(double) * maxnp);
      }
    } else {
      eval(actid, th, true, fv.data(), gv.data(), maxnp, okv.data());
    }
    for (size_t q = 0; q < act.size(); ++q) {
      const int pos = act[q];
      Lbfgs& s = st[pos];
      const bool ok = okv[q] != 0;
      const double* gq = gv.data() + q * maxnp;
      bool gfinite = ok;
      evals[pos] += 1;
      // (a trial point that fails the sufficient-decrease test is judged by its objective alone)
      const bool armijo_fail = s.state == Lbfgs::LINESEARCH && ok && !(fv[q] <= s.f + c1 * s.step * s.slope);
      if (ok && !armijo_fail) for (int k = 0; k < s.np; ++k) gfinite = gfinite && std::isfinite(gq[k]);
      if (s.state == Lbfgs::NEED_F0) {
        if (!gfinite) { s.state = Lbfgs::FAILED; finish_run(pos); continue; }
        s.f = fv[q];
        s.g.assign(gq, gq + s.np);
        s.direction();
        double gn = 0.;
        for (int k = 0; k < s.np; ++k) gn += s.g[k] * s.g[k];
        gn = std::sqrt(gn);
        if (gn <= opt.gtol) { s.state = Lbfgs::DONE; finish_run(pos); continue; }
        s.step = std::min(1.0, 1.0 / gn);
        s.step_lo = 0.; s.step_hi = 0.; s.ls_iter = 0;
        s.trial();
        s.state = Lbfgs::LINESEARCH;
        continue;
      }
      // line search step: Armijo (1e-4) + weak curvature (0.9) by bisection/expansion
      bool accept = false;
      if (!gfinite || !(fv[q] <= s.f + c1 * s.step * s.slope)) {
        s.step_hi = s.step;
        s.step = 0.5 * (s.step_lo + s.step_hi);
        g_ls_short += 1;
      } else {
        double st_slope = 0.;
        for (int k = 0; k < s.np; ++k) st_slope += gq[k] * s.d[k];
        if (st_slope < c2 * s.slope && s.ls_iter < 10) {
          s.step_lo = s.step;
          s.step = (s.step_hi > 0.) ? 0.5 * (s.step_lo + s.step_hi) : 2.0 * s.step;
          g_ls_long += 1;
        } else {
          accept = true;
        }
      }
      s.ls_iter++;
      if (!accept) {
        if (s.ls_iter > 40 || s.step < 1e-20 || evals[pos] >= eval_cap) {
          // could not make progress along d: take what we have
          s.state = Lbfgs::DONE;
          finish_run(pos);
          continue;
        }
        s.trial();
        continue;
      }
      // accept xt
      std::vector<double> sv(s.np), yv(s.np);
      double sy = 0., yy = 0.;
      for (int k = 0; k < s.np; ++k) {
        sv[k] = s.xt[k] - s.x[k];
        yv[k] = gq[k] - s.g[k];
        sy += sv[k] * yv[k];
        yy += yv[k] * yv[k];
      }
      const double fold = s.f;
      s.x = s.xt;
      s.f = fv[q];
      s.g.assign(gq, gq + s.np);
      if (sy > 1e-10 * yy && yy > 0.) {
        s.S.push_back(sv); s.Y.push_back(yv); s.rho.push_back(1.0 / sy);
        if (s.S.size() > 10) { s.S.erase(s.S.begin()); s.Y.erase(s.Y.begin()); s.rho.erase(s.rho.begin()); }
      }
      s.iter++;
      g_lb_iters += 1;
      double gmax = 0.;
      for (int k = 0; k < s.np; ++k) gmax = std::max(gmax, std::fabs(s.g[k]));
      if (std::fabs(fold - s.f) <= opt.ftol * std::max(1.0, std::fabs(s.f)) || gmax <= opt.gtol || s.iter >= opt.max_iter || evals[pos] >= eval_cap) {
        s.state = Lbfgs::DONE;
        finish_run(pos);
        continue;
      }
      s.direction();
      s.step = 1.0; s.step_lo = 0.; s.step_hi = 0.; s.ls_iter = 0;
      s.trial();
    }
  }
}

// Slot `slot` of this (replica) engine becomes a copy of emulator `i` of `src`: targets (host + device, with the fixed-mean
// subtraction of the constructor), nugget type / size, hyper-parameter and mean priors; no hyper-parameters, no factor.
void Engine::retarget(int slot, const Engine& src, int i) {
  std::copy(src.hT.begin() + (size_t)i * n, src.hT.begin() + (size_t)(i + 1) * n, hT.begin() + (size_t)slot * n);
  std::vector<double> res(hT.begin() + (size_t)slot * n, hT.begin() + (size_t)(slot + 1) * n);
  if (!analytic && mean.n_params() == 0 && mean.kind == 1)
    for (auto& x : res) x -= mean.value;
  HIPCK(hipMemcpyAsync(dT + (size_t)slot * n, res.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream));
  HIPCK(hipStreamSynchronize(stream));          // `res` is a temporary
  const GPState& s = src.gp[i];
  GPState d;
  d.nug_type = s.nug_type;
  d.nug_size = s.nug_size;
  d.pri = s.pri;
  d.mp_b = s.mp_b; d.mp_Binv = s.mp_Binv; d.mp_Binvb = s.mp_Binvb; d.mp_logdetB = s.mp_logdetB;
  d.data.assign(s.data.size(), 0.);
  d.meanp.assign(n_mean(), 0.);
  d.beta.assign(q, 0.);
  drop_w2(slot);
  gp[slot] = std::move(d);
}

// Multi-start MAP fit (fitting.hpp:61-128, fitting.py:219-266): n_tries L-BFGS runs per emulator, the best end point wins.
// All (emulator, start) runs go through ONE slot pool (run_pool).  The runs are independent, so as many of them as pay run
// CONCURRENTLY on a replica engine whose slots take the targets and priors of whatever run they are handed (retarget): small
// problems, whose batches do not fill the GPU, then cost about one start instead of n_tries, and large ones keep every batched
// evaluation full until the queue of runs drains.  MOGP_PARALLEL_STARTS=0: no replica engine -- the slots are this engine's own
// emulators, each working through its own starts one after the other (still without waiting for its neighbours).
void Engine::fit_map(const std::vector<int>& ids_in, int n_tries, const double* theta0, int theta0_len) {
  std::vector<int> ids(ids_in);
  if (ids.empty()) return;
  if (n_tries < 1) throw std::runtime_error("number of attempts must be positive");
  for (int i : ids)
    if (theta0_len > 0 && theta0_len != n_theta(i)) throw std::runtime_error("length of theta0 must equal n_params of GP.");
  const int ne = (int)ids.size();
  // starting points: start 0 = theta0 if given, everything else drawn from the priors (Priors.py:394-418)
  std::vector<std::vector<std::vector<double>>> x0(n_tries, std::vector<std::vector<double>>(ne));
  for (int s = 0; s < n_tries; ++s)
    for (int e = 0; e < ne; ++e) {
      std::vector<double>& x = x0[s][e];
      x.assign(n_theta(ids[e]), 0.);
      if (s == 0 && theta0_len > 0) x.assign(theta0, theta0 + theta0_len);
      else gp[ids[e]].pri.sample(rng, NC, gp[ids[e]].nug_type, x.data() + n_mean());
    }
  std::vector<double> best_f(ne, std::numeric_limits<double>::infinity());
  std::vector<int> best_s(ne, -1);
  std::vector<std::vector<double>> best_x(ne);
  // run tag = s * ne + e; of equal end points the one of the earlier start wins, whatever order the runs finish in
  auto done = [&](int tag, double f, const std::vector<double>& x) {
    const int e = tag % ne, s = tag / ne;
    if (!x.empty() && std::isfinite(f) && (f < best_f[e] || (f == best_f[e] && s < best_s[e]))) {
      best_f[e] = f;
      best_s[e] = s;
      best_x[e] = x;
    }
  };
  // how many replica slots: what fits beside this engine (A, L^-1, K^-1 per slot plus the small per-emulator buffers), what pays
  // (replicas beyond what fills the device buy nothing and their buffers are fresh allocations: 64 x 15 starts of n = 2000 as ONE batch
  // of 960 replicas took 3.7 - 5.8 s, capped at 256 2.44 s, 128: 2.52 s -- round 2), MOGP_START_REPLICAS overrides the cap
  static const bool parallel_starts = [] { const char* e = getenv("MOGP_PARALLEL_STARTS"); return !e || e[0] != '0'; }();
  const long total = (long)ne * n_tries;
  long slots_n = ne;
  if (parallel_starts && n_tries > 1) {
    size_t free_b = 0, total_b = 0;
    long fit = total;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const double per_emu = 3.0 * (double)MS * sizeof(double) + 16.0 * LD * sizeof(double);
      fit = (long)std::max(1.0, std::floor(0.5 * (double)free_b / per_emu));
    }
    static const long replica_cap = [] { const char* e = getenv("MOGP_START_REPLICAS"); return e ? atol(e) : 0L; }();
    const long cap = replica_cap > 0 ? replica_cap : std::max<long>(ne, 4095 / std::max(1, NP / TILE) + 1);
    slots_n = std::min(total, std::min(fit, cap));
    if (slots_n > 8) slots_n -= slots_n % 8;      // (batches that are multiples of 8 give every XCD whole emulators)
  }
  if (slots_n <= ne || !parallel_starts || n_tries == 1) {
    // no replicas: slot e IS emulator ids[e] and works through its own starts
    std::vector<int> nexts(ne, 0);
    run_pool(ids,
             [&](int pos, std::vector<double>& x, int& tag) {
               if (nexts[pos] >= n_tries) return false;
               const int s = nexts[pos]++;
               x = x0[s][pos];
               tag = s * ne + pos;
               return true;
             },
             done);
  } else {
    // ONE replica engine of slots_n slots; the queue of runs in start-major order (all emulators' start 0 first)
    std::vector<double> targets((size_t)slots_n * n);
    for (long k = 0; k < slots_n; ++k) {
      const int e = (int)(k % ne);
      std::copy(hT.begin() + (size_t)ids[e] * n, hT.begin() + (size_t)(ids[e] + 1) * n, targets.begin() + (size_t)k * n);
    }
    const auto tc0 = std::chrono::steady_clock::now();
    // The replica engine of the LAST multi-start fit of the process is kept (one engine, process-wide) and taken again when the next fit
    // has the same shape (n, D, slots, kernel, mean function, device): its 3 x slots matrices are tens of GB, and on SOME boxes fresh
    // allocations of that size cost 0.9 - 1.5 s per fit (bench.py run behind the GPU test suite on the same box: fit_GP_MAP 3.1 instead of
    // 2.15 s, every call with 256 slots; on a fresh box 2.13 s from the first call; writing 200 GB from another process beforehand did not
    // reproduce it -- profiles/r06_fitmap_context_and_grad_chain.txt).  A fit of another shape frees it.  MOGP_REPLICA_CACHE=0: every fit
    // builds and frees its own.
    static const bool cache_on = [] { const char* e = getenv("MOGP_REPLICA_CACHE"); return !e || atoi(e) != 0; }();
    std::unique_ptr<Engine> rep;
    {
      std::lock_guard<std::mutex> lk(replica_cache_mutex());
      std::unique_ptr<Engine>& slot = replica_cache();
      int dev = -1;
      (void)hipGetDevice(&dev);
      if (slot && cache_on && slot->n == n && slot->D == D && slot->B == (int)slots_n && slot->kernel_type == kernel_type && slot->analytic == analytic &&
          slot->testing_size == testing_size && slot->device == dev && slot->mean.kind == mean.kind && slot->mean.value == mean.value &&
          slot->mean.dims == mean.dims && slot->mean.powers == mean.powers) {
        rep = std::move(slot);
        rep->hX = hX;
        HIPCK(hipMemcpy(rep->dX, hX.data(), hX.size() * sizeof(double), hipMemcpyHostToDevice));
        g_rep_reused += 1;
      } else {
        slot.reset();                                   // (frees the old one BEFORE the new one is allocated)
      }
    }
    if (!rep)
      rep.reset(new Engine(hX.data(), n, D, targets.data(), (int)slots_n, testing_size, mean, kernel_type, gp[ids[0]].nug_type, gp[ids[0]].nug_size, analytic));
    // hand the engine back to the cache when this block is left normally (an exception destroys it)
    struct Keep {
      std::unique_ptr<Engine>& rep;
      bool on;
      ~Keep() {
        if (!on || !rep || std::uncaught_exceptions() > 0) return;
        std::lock_guard<std::mutex> lk(replica_cache_mutex());
        replica_cache() = std::move(rep);
      }
    } keep{rep, cache_on};
    g_rep_build_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tc0).count();
    std::vector<int> holds(slots_n, -1);          // which emulator (index into ids) a slot's targets and priors belong to
    std::vector<int> rslots(slots_n);
    for (long k = 0; k < slots_n; ++k) rslots[k] = (int)k;
    long next_run = 0;
    Engine* self = this;
    const auto tp0 = std::chrono::steady_clock::now();
    struct PoolTimer {
      std::chrono::steady_clock::time_point t0;
      ~PoolTimer() { g_rep_pool_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); }
    } pool_timer{tp0};
    rep->run_pool(rslots,
                  [&](int pos, std::vector<double>& x, int& tag) {
                    if (next_run >= total) return false;
                    const long r = next_run++;
                    const int s = (int)(r / ne), e = (int)(r % ne);
                    if (holds[pos] != e) {
                      const auto tr0 = std::chrono::steady_clock::now();
                      rep->retarget(pos, *self, ids[e]);
                      g_retarget_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tr0).count();
                      g_retargets += 1;
                      holds[pos] = e;
                    }
                    x = x0[s][e];
                    tag = (int)r;
                    return true;
                  },
                  done);
  }
  // refit at the best point of every emulator (fitting.hpp:115-117); failures -> "not fit" (:111-113)
  std::vector<int> fin;
  std::vector<const double*> th;
  for (int e = 0; e < ne; ++e) {
    if (best_x[e].empty()) {
      gp[ids[e]].has_data = false;
      gp[ids[e]].factored = false;
    } else {
      fin.push_back(ids[e]);
      th.push_back(best_x[e].data());
    }
  }
  if (!fin.empty()) {
    std::vector<double> fv(fin.size());
    std::vector<int> okv(fin.size());
    eval(fin, th, false, fv.data(), nullptr, 0, okv.data());
  }
}

}  // namespace mogp
