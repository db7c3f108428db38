// pgb_host.h -- part of pgbart_hip.hip (not a standalone header): host side: handles, the C ABI of include/pgbart.h, slot enqueueing.
// ------------------------------------------------------------------ host side
enum { PK_CTRL = 0, PK_ROWS = 1, PK_LL = 2, PK_COUNT = 3 };
static std::atomic<int> g_live_handles{0};  // samplers alive in this process (see feed_until_flag)
static thread_local char g_err[512];
static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}
static int fail_hip(hipError_t e, const char* what) {
  snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
  return PGB_E_DEVICE;
}
#define HIPCHK(expr)                                     \
  do {                                                   \
    hipError_t e_ = (expr);                              \
    if (e_ != hipSuccess) return fail_hip(e_, #expr);    \
  } while (0)

struct pgb_handle {
  pgb_settings s;
  Dev d;
  Dev* d_dev;                          // device-resident copy passed to every kernel
  volatile unsigned long long* flag;   // pinned host word written by k_ctrl (completed asteps)
  long long steps_target;              // asteps requested so far
  hipStream_t stream;
  hipStream_t stream_out;              // pgb_step_host: the step's results leave on this stream, past the idle slots
  std::vector<void*> allocs;
  std::vector<size_t> alloc_bytes;     // per allocation: payload size ...
  std::vector<char> alloc_owned;       // ... and whether it is a hipMalloc of its own (else a piece of `slab`)
  void* slab;                          // small records share one allocation (see dalloc_bytes)
  size_t slab_used;
  long long slot;  // next slot index (parity = slot & 1)
  int st_cur, alpha_cur;  // mirrors of Ctrl::st_cur / alpha_cur at the last idle point
  int have_data, have_y;
  int has_subset;  // any SubsetSplit column: selects the row-pass instance
  int rows_grid;   // workgroups of the persistent row-pass grid (dispatch costs ~3.5 ns each)
  int ll_grid;     // ... of the log-likelihood pass
  int rows_mk_cap; // K-vector row pass: workgroups its instance keeps resident (0: not queried yet)
  // the instance of k_loglik this sampler launches (family / outputs / response)
  void (*ll_kernel)(const Dev*, int, int, const Cmd*, const Ctrl*, const Job*, const Acc*, const InitAcc*);
  int sigma_dirty;
  double inv_sigma2;
  double lik_param2;
  int lower_host;      // mirror of the batch cursor
  int last_lower, last_n;
  double slots_per_step;  // running estimate of the working slots one astep needs ...
  double slots_var;       // ... and of their variance
  pgb_counters ctr;
  // profiling: events attached to the dispatches of a region, per kernel (PK_*); the row pass is
  // the dominant kernel pgb_profile reports
  int prof;
  std::vector<hipEvent_t> ev[PK_COUNT];
  size_t ev_used[PK_COUNT];
  double prof_ms[PK_COUNT];
  long long prof_launches[PK_COUNT];
  int prof_wgs[PK_COUNT];
  // host-facing results of the last pgb_step_host (mapped pinned block written by k_export_step)
  unsigned char* out_host;   // host address
  unsigned char* out_dev;    // the same block as the device sees it
  StepOutLayout out_layout;
  double* st_dense;          // [K][n] staging of sum_trees in HBM
  int out_valid;             // the block holds the trees of the last step (pgb_export_trees(0) reads it)
  int poisoned;              // a step was abandoned half-way (callback error, stuck state machine): the device
                             // state is undefined until pgb_checkpoint_load restores an idle image
  // pgb_step_async: a worker thread feeds the state machine while the caller goes on
  // callback family: the host evaluates the per-row log-likelihood once per slot
  pgb_loglik_fn cb_fn;
  void* cb_ctx;
  std::vector<double> y_host, off_host;
  std::vector<int32_t> rules_host;  // the PGB_RULE_* of the columns: every exported split node carries its own
  int device;
  std::thread worker;
  int job_running, job_rc;
  char job_err[512];
  long long* prof_buf;    // device-clock stamps (allocated on first use)
  long long prof_slot0;   // first slot of the profiled region (device-clock stamps)
  double prof_clock_ms;   // sum over launches of max(end) - min(start), 100 MHz device clock
  long long prof_clock_launches;
};

// A running pgb_step_async job owns the handle: every other entry point waits for it first.
static int join_async(pgb_handle* h) {
  if (h->worker.joinable()) h->worker.join();
  if (h->job_running) {
    h->job_running = 0;
    if (h->job_rc != PGB_OK) {
      const int rc = h->job_rc;
      h->job_rc = PGB_OK;
      return fail(rc, h->job_err);
    }
  }
  return PGB_OK;
}
#define JOIN_ASYNC(h)                          \
  do {                                         \
    int rcj_ = join_async(h);                  \
    if (rcj_ != PGB_OK) return rcj_;           \
  } while (0)
// Entry points that run or read the chain refuse a poisoned handle (see pgb_handle::poisoned).
#define REFUSE_POISONED(h)                                                                                  \
  do {                                                                                                      \
    if ((h)->poisoned)                                                                                      \
      return fail(PGB_E_STATE, "an earlier step of this sampler was abandoned half-way (log-likelihood "    \
                               "callback error or stuck state machine): its state is undefined; restore a " \
                               "checkpoint (pgb_checkpoint_load) or create a new sampler");                 \
  } while (0)

// Device allocations.  The small records every slot touches first (control words, commands, job records,
// split statistics, counters, split weights ...) are carved out of ONE slab: a dozen separate hipMallocs put
// each of them on a page of its own, and the first loads of every (latency-bound) control kernel then paid a
// TLB miss each -- most of the ~1.7 us a fresh kernel waited for its first data.
#define SLAB_BYTES ((size_t)2 << 20)
#define SLAB_ITEM_MAX ((size_t)192 << 10)
static int dalloc_bytes(pgb_handle* h, void** p, size_t bytes) {
  static const bool use_slab = !(getenv("PGB_NO_SLAB") && atoi(getenv("PGB_NO_SLAB")));
  void* q = nullptr;
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (use_slab && need <= SLAB_ITEM_MAX) {
    if (!h->slab) {
      hipError_t e = hipMalloc(&h->slab, SLAB_BYTES);
      if (e != hipSuccess) return fail_hip(e, "hipMalloc");
      h->slab_used = 0;
    }
    if (h->slab_used + need <= SLAB_BYTES) {
      q = (char*)h->slab + h->slab_used;
      h->slab_used += need;
      h->alloc_owned.push_back(0);
    }
  }
  if (!q) {
    hipError_t e = hipMalloc(&q, bytes + 256);
    if (e != hipSuccess) return fail_hip(e, "hipMalloc");
    h->alloc_owned.push_back(1);
  }
  h->allocs.push_back(q);
  h->alloc_bytes.push_back(bytes);
  *p = q;
  return PGB_OK;
}
template <typename T>
static int dalloc(pgb_handle* h, T** p, size_t count) {
  void* q = nullptr;
  const int rc = dalloc_bytes(h, &q, count * sizeof(T));
  if (rc == PGB_OK) *p = (T*)q;
  return rc;
}

extern "C" const char* pgb_last_error(void) { return g_err; }
extern "C" const char* pgb_backend_name(void) { return "hip-gfx950"; }
extern "C" int32_t pgb_max_particles(void) { return PGB_MAX_PARTICLES; }
extern "C" int32_t pgb_abi_version(void) { return PGB_ABI_VERSION; }

// The instance of k_loglik a sampler launches, chosen once: per number of outputs (loops unrolled for
// K = 2, 3, 4), per family for single-output constant leaves (one family's code per instance).
typedef void (*ll_kernel_t)(const Dev*, int, int, const Cmd*, const Ctrl*, const Job*, const Acc*, const InitAcc*);
static ll_kernel_t select_ll_kernel(int K, bool lin, int family) {
  if (K > 1 && lin) return k_loglik<0, -1, true>;
  if (K > 1) {  // constant K-vector leaves: softmax (factorised evaluation, any K) or Normal mean / scale (K = 2)
    if (family != PGB_FAMILY_CATEGORICAL) return k_loglik<2, PGB_FAMILY_NORMAL_MEANSCALE, false>;
    switch (K) {
      case 2: return k_loglik<2, PGB_FAMILY_CATEGORICAL, false>;
      case 3: return k_loglik<3, PGB_FAMILY_CATEGORICAL, false>;
      case 4: return k_loglik<4, PGB_FAMILY_CATEGORICAL, false>;
      default: return k_loglik<0, PGB_FAMILY_CATEGORICAL, false>;
    }
  }
  if (lin) return k_loglik<1, -1, true>;  // linear leaves: one instance, family read at run time
  switch (family) {
    case PGB_FAMILY_BERNOULLI_PROBIT: return k_loglik<1, PGB_FAMILY_BERNOULLI_PROBIT, false>;
    case PGB_FAMILY_BERNOULLI_LOGIT: return k_loglik<1, PGB_FAMILY_BERNOULLI_LOGIT, false>;
    case PGB_FAMILY_POISSON_LOG: return k_loglik<1, PGB_FAMILY_POISSON_LOG, false>;
    case PGB_FAMILY_NEGBIN_LOG: return k_loglik<1, PGB_FAMILY_NEGBIN_LOG, false>;
    case PGB_FAMILY_ASYMLAPLACE: return k_loglik<1, PGB_FAMILY_ASYMLAPLACE, false>;
    case PGB_FAMILY_GAMMA_LOG: return k_loglik<1, PGB_FAMILY_GAMMA_LOG, false>;
    case PGB_FAMILY_CALLBACK: return k_loglik<1, PGB_FAMILY_CALLBACK, false>;
    default: return k_loglik<1, PGB_FAMILY_STUDENT_T, false>;
  }
}

#define PGB_STR2(x) #x
#define PGB_STR(x) PGB_STR2(x)
extern "C" int pgb_create(const pgb_settings* s, void* stream, pgb_handle** out) {
  if (!s || !out) return fail(PGB_E_INVALID, "null argument");
  if (s->n < 1 || s->p < 1 || s->m < 1) return fail(PGB_E_INVALID, "n, p, m must be >= 1");
  if (s->n >= (1ll << 31) - CH) return fail(PGB_E_UNSUPPORTED, "n too large");
  if (s->num_particles < 2 || s->num_particles > PGB_MAX_PARTICLES)
    return fail(PGB_E_INVALID, PGB_MAX_PARTICLES == 64
                                   ? "num_particles must be in [2, 64] (libpgbart_hip_p128.so takes up to 128)"
                                   : "num_particles must be in [2, 128]");
  if (s->family == PGB_FAMILY_CATEGORICAL) {
    if (s->n_outputs < 2 || s->n_outputs > PGB_MAX_OUTPUTS)
      return fail(PGB_E_INVALID, "CATEGORICAL needs 2 <= n_outputs <= " PGB_STR(PGB_MAX_OUTPUTS));
  } else if (s->family == PGB_FAMILY_NORMAL_MEANSCALE) {
    if (s->n_outputs != 2) return fail(PGB_E_INVALID, "NORMAL_MEANSCALE needs n_outputs == 2");
  } else if (s->family == PGB_FAMILY_NORMAL || s->family == PGB_FAMILY_BERNOULLI_PROBIT ||
             s->family == PGB_FAMILY_BERNOULLI_LOGIT || s->family == PGB_FAMILY_POISSON_LOG ||
             s->family == PGB_FAMILY_NEGBIN_LOG || s->family == PGB_FAMILY_ASYMLAPLACE ||
             s->family == PGB_FAMILY_STUDENT_T || s->family == PGB_FAMILY_GAMMA_LOG ||
             s->family == PGB_FAMILY_CALLBACK) {
    if (s->n_outputs != 1) return fail(PGB_E_INVALID, "this family has a single output");
    if (s->family == PGB_FAMILY_CALLBACK && s->response != PGB_RESPONSE_CONSTANT)
      return fail(PGB_E_UNSUPPORTED, "the callback family has constant leaves");
  } else {
    return fail(PGB_E_UNSUPPORTED, "unknown family");
  }
  if (s->batch_tune < 1 || s->batch_draw < 1) return fail(PGB_E_INVALID, "batch sizes must be >= 1");
  if (s->compat & ~PGB_COMPAT_ALL) return fail(PGB_E_INVALID, "unknown compat bits (pgbart_spec.h: PGB_COMPAT_*)");
  if (s->response != PGB_RESPONSE_CONSTANT) {
    if (s->response != PGB_RESPONSE_LINEAR && s->response != PGB_RESPONSE_MIX)
      return fail(PGB_E_UNSUPPORTED, "unknown response");
  }
  int ndev = 0;
  hipError_t e0 = hipGetDeviceCount(&ndev);
  if (e0 != hipSuccess || ndev < 1) {
    snprintf(g_err, sizeof g_err, "no HIP device visible (%s, count %d); this backend has no CPU fallback",
             hipGetErrorString(e0), ndev);
    return PGB_E_DEVICE;
  }
  pgb_handle* h = new pgb_handle();
  g_live_handles.fetch_add(1);
  h->s = *s;
  h->stream = (hipStream_t)stream;
  h->stream_out = nullptr;
  h->slab = nullptr;
  h->slab_used = 0;
  h->slot = 0;
  h->has_subset = 0;
  h->rows_grid = 1024;
  h->rows_mk_cap = 0;
  h->prof_buf = nullptr;
  h->prof = 0;
  for (int k = 0; k < PK_COUNT; ++k) {
    h->ev_used[k] = 0;
    h->prof_ms[k] = 0.0;
    h->prof_launches[k] = 0;
    h->prof_wgs[k] = 0;
  }
  h->cb_fn = nullptr;
  h->cb_ctx = nullptr;
  h->out_host = h->out_dev = nullptr;
  h->st_dense = nullptr;
  h->out_valid = 0;
  h->poisoned = 0;
  h->job_running = 0;
  h->job_rc = PGB_OK;
  h->device = 0;
  (void)hipGetDevice(&h->device);
  h->prof_clock_ms = 0.0;
  h->prof_clock_launches = 0;
  h->prof_slot0 = 0;
  if (const char* e = getenv("PGB_ROWS_GRID")) h->rows_grid = atoi(e) > 0 ? atoi(e) : h->rows_grid;
  h->ll_grid = h->rows_grid;
  if (const char* e = getenv("PGB_LL_GRID")) h->ll_grid = atoi(e) > 0 ? atoi(e) : h->ll_grid;
  h->st_cur = 0;
  h->alpha_cur = 0;
  h->d_dev = nullptr;
  h->flag = nullptr;
  h->steps_target = 0;
  h->inv_sigma2 = 1.0;
  h->lik_param2 = 1.0;
  h->sigma_dirty = 1;
  h->slots_per_step = 0.0;
  h->slots_var = 0.0;
  memset(&h->ctr, 0, sizeof h->ctr);
  Dev& d = h->d;
  memset(&d, 0, sizeof d);
  d.n = s->n;
  d.nchunks = (int)((s->n + CH - 1) / CH);
  d.n_pad = (long long)d.nchunks * CH;
  d.cc_stride = (d.nchunks + 7) & ~7;
  d.p = s->p;
  d.m = s->m;
  d.P = s->num_particles;
  d.family = s->family;
  d.K = s->n_outputs;
  d.response = s->response;
  d.lin_R = pgb_pow2(s->range_exp - 1);
  d.inv_R = pgb_pow2(1 - s->range_exp);
  d.rows_target = ROWS_TARGET_ITEMS;
  d.rows_target_init = ROWS_TARGET_ITEMS_INIT;
  if (const char* e = getenv("PGB_ROWS_TARGET")) d.rows_target = atoi(e) > 0 ? atoi(e) : d.rows_target;
  d.ll_target = d.rows_target;
  d.compat = s->compat;
  if (const char* e = getenv("PGB_LL_TARGET")) d.ll_target = atoi(e) > 0 ? atoi(e) : d.ll_target;
  // The likelihood pass is a persistent grid too, and its instances differ a lot in registers (probit 88
  // VGPRs, K = 4 softmax 154): the grid is what the chosen instance can keep resident -- 5 workgroups per
  // CU for probit, 3 for K = 4 -- so that no workgroup waits for another to finish; the pass aims for at
  // least as many work items.  (cfg4: 1024 -> 1280 workgroups, k_loglik 40.3 -> 37.5 us.)
  h->ll_kernel = select_ll_kernel(d.K, d.response != PGB_RESPONSE_CONSTANT, d.family);
  if (d.family != PGB_FAMILY_NORMAL && !getenv("PGB_LL_GRID")) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)h->ll_kernel, BT, 0) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && per_cu > 0 && cus > 0) {
      // (K = 2, 3, 4: the instances are compiled for PGB_LLK_WGS workgroups per CU and measured best there --
      //  cfg5: 768 workgroups 27.2-27.9 us, 1024 28.1-31.2 us, 1280 30.5-31.0 us -- whatever else would fit)
      if (d.K >= 2 && d.K <= 4 && d.response == PGB_RESPONSE_CONSTANT && per_cu > PGB_LLK_WGS) per_cu = PGB_LLK_WGS;
      long long g = (long long)per_cu * cus;
      if (g < 256) g = 256;
      if (g > 2048) g = 2048;
      h->ll_grid = (int)g;
      if (!getenv("PGB_LL_TARGET")) d.ll_target = h->ll_grid > 1024 ? h->ll_grid : 1024;
    }
  }
  if (const char* e = getenv("PGB_ROWS_TARGET_INIT")) d.rows_target_init = atoi(e) > 0 ? atoi(e) : d.rows_target_init;
  // One launch per SMC round where the round is latency-bound: Normal likelihood, one output, constant
  // leaves, and few enough (particle, chunk) pairs that a work item holds <= GMAXF particles.
  d.batch_tune = s->batch_tune;
  d.batch_draw = s->batch_draw;
  d.seed = s->seed;
  d.init_leaf = s->init_leaf;
  d.mdouble = (double)s->m;
  d.sc = pgb_make_scales(s->n, s->range_exp);
  int rc;
  double *XT, *y, *st, *rs_mean, *rs_m2, *prior_leaf;
  long long *alpha, *cdfS;
  double2* pack;
  uint8_t *tree_lid, *lid;
  uint16_t* cc;
  int32_t *vi, *rules, *col_nan;
#define DA(ptr, cnt) \
  if ((rc = dalloc(h, &ptr, (size_t)(cnt))) != PGB_OK) { pgb_destroy(h); return rc; }
  DA(XT, (size_t)d.p * d.n_pad);
  DA(y, d.n_pad);
  double* off;
  DA(off, (size_t)d.K * d.n_pad);
  d.off = off;
  const int K = d.K, KX = d.K - 1;
  DA(st, (size_t)2 * K * d.n_pad);
  DA(pack, d.n_pad);
  DA(rs_mean, (size_t)K * d.n_pad);
  DA(rs_m2, (size_t)K * d.n_pad);
  if (KX > 0) {  // K-vector leaves: extension outputs
    DA(d.packx, (size_t)KX * d.n_pad);
    DA(d.pvx, (size_t)2 * MAXP * MAXN * KX);
    DA(d.pqx, (size_t)2 * MAXP * MAXN * KX);
    DA(d.tvx, (size_t)d.m * MAXN * KX);
    DA(d.accx, (size_t)2 * MAXP * AX_PER);
    DA(d.iax, (size_t)2 * IA_SLOTS * 2 * KX);
    DA(d.lvx, (size_t)2 * 2 * 256 * KX);
    DA(d.jqx, (size_t)2 * MAXP * KX);
    DA(d.jvx, (size_t)2 * MAXP * KX);
    DA(d.jzx, (size_t)2 * MAXP * KX * 2);
    DA(d.lsdx, (size_t)2 * KXMAX);
    DA(d.finx, (size_t)2 * MAXP * KX);
    if (s->family == PGB_FAMILY_CATEGORICAL && s->response == PGB_RESPONSE_CONSTANT) {
      // the row part of the factorised softmax: scratch of one tree update (no checkpoint carries it)
      // (zeroed once below: the passes of a tree's later rounds read the padded rows of the last chunk as well, which
      //  the pass that starts the tree never writes)
      DA(d.cat_e, (size_t)K * d.n_pad);
      DA(d.cat_a, d.n_pad);
      DA(d.cat_c, d.n_pad);
    }
  }
  DA(tree_lid, (size_t)d.m * d.n_pad);
  DA(lid, (size_t)NGEN * MAXP * d.n_pad);
  DA(cc, (size_t)CC_ROUNDS * MAXP * 2 * d.cc_stride);
  if (s->family == PGB_FAMILY_CALLBACK) {
    DA(d.cb_mu, (size_t)MAXP * d.n_pad);
    DA(d.cb_side, (size_t)MAXP * d.n_pad);
    h->y_host.assign((size_t)d.n, 0.0);
    h->off_host.assign((size_t)d.n, 0.0);
  }
  DA(d.trees, d.m);
  DA(d.parts, 2 * MAXP);
  DA(d.jobs, 2 * MAXP);
  DA(d.acc, 2 * MAXP * ACC_PER);
  DA(d.accl, 2 * MAXP * LL_PER);
  DA(d.jobl, 2 * MAXP);
  DA(d.initacc, 2 * IA_SLOTS);
  DA(d.cmd, 2);
  DA(d.ctrl, 2);
  DA(d.counters, 8);
  DA(vi, d.p);
  DA(alpha, 2 * d.p);
  DA(cdfS, 2 * d.p);
  DA(prior_leaf, PGB_MAX_DEPTH);
  DA(rules, d.p);
  DA(col_nan, d.p);
  int32_t* col_ex;
  DA(col_ex, d.p);
  d.col_ex = col_ex;
  if (d.response != PGB_RESPONSE_CONSTANT) {
    DA(d.plin, (size_t)2 * MAXP * MAXN);
    DA(d.tlin, (size_t)d.m * MAXN);
    DA(d.lvl, (size_t)2 * 2 * 256);
    DA(d.accu, (size_t)2 * MAXP * ACC_PER);
    if (KX > 0) {
      DA(d.psx, (size_t)2 * MAXP * MAXN * KX);
      DA(d.tsx, (size_t)d.m * MAXN * KX);
      DA(d.lsx, (size_t)2 * 2 * 256 * KX);
      DA(d.accux, (size_t)2 * MAXP * AX_PER);
    }
  }
#undef DA
  d.XT = XT; d.y = y; d.st = st; d.pack = pack; d.rs_mean = rs_mean; d.rs_m2 = rs_m2;
  d.tree_lid = tree_lid; d.lid = lid; d.cc = cc; d.vi = vi; d.alpha = alpha; d.cdfS = cdfS;
  d.rules = rules; d.col_nan = col_nan; d.prior_leaf = prior_leaf;
  hipStream_t sm = h->stream;
  hipError_t e;
#define HC(expr) \
  if ((e = (expr)) != hipSuccess) { int r_ = fail_hip(e, #expr); pgb_destroy(h); return r_; }
  {
    void* hp = nullptr;
    HC(hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent));
    h->flag = (volatile unsigned long long*)hp;
    h->flag[0] = 0;  // asteps whose last tree has been accepted (its FINAL row pass may still be running)
    h->flag[1] = 0;  // asteps that are complete on the device (published by the first idle slot after them)
    h->flag[2] = 0;  // slots whose control kernel has started (the credit the host enqueues against)
    h->flag[4] = 0;  // set by k_nonfinite (pgb_set_response / pgb_set_offset)
    // (h->stream_out: see pgb_set_output_stream)
    void* dp = nullptr;
    HC(hipHostGetDevicePointer(&dp, hp, 0));
    d.host_flag = (unsigned long long*)dp;
#ifdef PGB_TRACE
    if ((rc = dalloc(h, &d.trace, (size_t)TRACE_SLOTS * TRACE_W)) != PGB_OK) { pgb_destroy(h); return rc; }
    HC(hipMemsetAsync(d.trace, 0, (size_t)TRACE_SLOTS * TRACE_W * sizeof(long long), sm));
#endif
    {  // host-facing results of pgb_step_host: mapped pinned block + dense sum_trees staging
      int cap_trees = s->batch_tune > s->batch_draw ? s->batch_tune : s->batch_draw;
      if (cap_trees > s->m) cap_trees = s->m;
      h->out_layout = stepout_layout(d.p, cap_trees, d.K, d.response != PGB_RESPONSE_CONSTANT);
      void* hp2 = nullptr;
      HC(hipHostMalloc(&hp2, (size_t)h->out_layout.bytes, hipHostMallocMapped | hipHostMallocCoherent));
      memset(hp2, 0, (size_t)h->out_layout.bytes);
      h->out_host = (unsigned char*)hp2;
      void* dp2 = nullptr;
      HC(hipHostGetDevicePointer(&dp2, hp2, 0));
      h->out_dev = (unsigned char*)dp2;
      if ((rc = dalloc(h, &h->st_dense, (size_t)d.K * d.n)) != PGB_OK) { pgb_destroy(h); return rc; }
    }
    if ((rc = dalloc(h, &h->d_dev, 1)) != PGB_OK) { pgb_destroy(h); return rc; }
    HC(hipMemcpyAsync(h->d_dev, &d, sizeof(Dev), hipMemcpyHostToDevice, sm));
  }
  HC(hipMemcpyAsync(prior_leaf, s->prior_leaf, PGB_MAX_DEPTH * sizeof(double), hipMemcpyHostToDevice, sm));
  HC(hipMemsetAsync(y, 0, d.n_pad * sizeof(double), sm));
  HC(hipMemsetAsync(off, 0, (size_t)d.K * d.n_pad * sizeof(double), sm));
  HC(hipMemsetAsync(pack, 0, d.n_pad * sizeof(double2), sm));
  HC(hipMemsetAsync(rs_mean, 0, (size_t)K * d.n_pad * sizeof(double), sm));
  HC(hipMemsetAsync(rs_m2, 0, (size_t)K * d.n_pad * sizeof(double), sm));
  if (KX > 0) {
    HC(hipMemsetAsync(d.packx, 0, (size_t)KX * d.n_pad * sizeof(double), sm));
    HC(hipMemsetAsync(d.pvx, 0, (size_t)2 * MAXP * MAXN * KX * sizeof(double), sm));
    HC(hipMemsetAsync(d.pqx, 0, (size_t)2 * MAXP * MAXN * KX * sizeof(long long), sm));
    HC(hipMemsetAsync(d.accx, 0, (size_t)2 * MAXP * AX_PER * sizeof(long long), sm));
    HC(hipMemsetAsync(d.iax, 0, (size_t)2 * IA_SLOTS * 2 * KX * sizeof(long long), sm));
    HC(hipMemsetAsync(d.lvx, 0, (size_t)2 * 2 * 256 * KX * sizeof(double), sm));
    HC(hipMemsetAsync(d.jqx, 0, (size_t)2 * MAXP * KX * sizeof(long long), sm));
    HC(hipMemsetAsync(d.jvx, 0, (size_t)2 * MAXP * KX * sizeof(double), sm));
    HC(hipMemsetAsync(d.jzx, 0, (size_t)2 * MAXP * KX * 2 * sizeof(double), sm));
    HC(hipMemsetAsync(d.finx, 0, (size_t)2 * MAXP * KX * sizeof(FinX), sm));
    if (d.cat_e) {
      HC(hipMemsetAsync(d.cat_e, 0, (size_t)K * d.n_pad * sizeof(double), sm));
      HC(hipMemsetAsync(d.cat_a, 0, d.n_pad * sizeof(double), sm));
      HC(hipMemsetAsync(d.cat_c, 0, d.n_pad, sm));
    }
    hipLaunchKernelGGL(k_fill_f64, dim3(1), dim3(256), 0, sm, d.lsdx, (long long)2 * KXMAX, s->init_leaf_sd);
    // every accepted tree starts as a stump whose K-vector leaf is init_leaf
    hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)(((size_t)d.m * MAXN * KX + 255) / 256)), dim3(256), 0, sm,
                       d.tvx, (long long)d.m * MAXN * KX, s->init_leaf);
  }
  HC(hipMemsetAsync(lid, PGB_ORPHAN, (size_t)NGEN * MAXP * d.n_pad, sm));
  HC(hipMemsetAsync(cc, 0, (size_t)CC_ROUNDS * MAXP * 2 * d.cc_stride * sizeof(uint16_t), sm));
  HC(hipMemsetAsync(d.parts, 0, 2 * MAXP * sizeof(DPart), sm));
  HC(hipMemsetAsync(d.jobs, 0, 2 * MAXP * sizeof(Job), sm));
  HC(hipMemsetAsync(d.acc, 0, 2 * MAXP * ACC_PER * sizeof(Acc), sm));
  HC(hipMemsetAsync(d.accl, 0, 2 * MAXP * LL_PER * sizeof(AccL), sm));
  HC(hipMemsetAsync(d.jobl, 0, 2 * MAXP * sizeof(JobL), sm));
  HC(hipMemsetAsync(d.initacc, 0, 2 * IA_SLOTS * sizeof(InitAcc), sm));
  HC(hipMemsetAsync(d.cmd, 0, 2 * sizeof(Cmd), sm));
  HC(hipMemsetAsync(d.counters, 0, 8 * sizeof(unsigned long long), sm));
  HC(hipMemsetAsync(vi, 0, d.p * sizeof(int32_t), sm));
  HC(hipMemsetAsync(col_nan, 0, d.p * sizeof(int32_t), sm));
  HC(hipMemsetAsync(col_ex, 0, d.p * sizeof(int32_t), sm));
  if (d.response != PGB_RESPONSE_CONSTANT) {
    const long long n1 = (long long)2 * MAXP * MAXN, n2 = (long long)d.m * MAXN, n3 = 2 * 2 * 256;
    hipLaunchKernelGGL(k_init_linp, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, sm, d.plin, n1);
    hipLaunchKernelGGL(k_init_linp, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, sm, d.tlin, n2);
    hipLaunchKernelGGL(k_init_linp, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, sm, d.lvl, n3);
    HC(hipMemsetAsync(d.accu, 0, (size_t)2 * MAXP * ACC_PER * sizeof(AccU), sm));
    if (KX > 0) {
      HC(hipMemsetAsync(d.psx, 0, (size_t)2 * MAXP * MAXN * KX * sizeof(double), sm));
      HC(hipMemsetAsync(d.tsx, 0, (size_t)d.m * MAXN * KX * sizeof(double), sm));
      HC(hipMemsetAsync(d.lsx, 0, (size_t)2 * 2 * 256 * KX * sizeof(double), sm));
      HC(hipMemsetAsync(d.accux, 0, (size_t)2 * MAXP * AX_PER * sizeof(long long), sm));
    }
  }
  Ctrl c0;
  memset(&c0, 0, sizeof c0);
  c0.phase = PH_IDLE;
  c0.leaf_sd = s->init_leaf_sd;
  c0.inv_sigma2 = 1.0;
  c0.lik_param2 = 1.0;
  Ctrl cc2[2] = {c0, c0};
  HC(hipMemcpyAsync(d.ctrl, cc2, sizeof cc2, hipMemcpyHostToDevice, sm));
  hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)((2 * K * d.n_pad + 255) / 256)), dim3(256), 0, sm, st,
                     2 * K * d.n_pad, s->init_sum);
  hipLaunchKernelGGL(k_init_tree_lid, dim3((unsigned)((d.n_pad * d.m + 255) / 256)), dim3(256), 0, sm,
                     tree_lid, d.n, d.n_pad, d.m);
  hipLaunchKernelGGL(k_init_trees, dim3((d.m + 63) / 64), dim3(64), 0, sm, d.trees, d.m, d.n,
                     s->init_leaf);
  HC(hipGetLastError());
  HC(hipStreamSynchronize(sm));
#undef HC
  *out = h;
  return PGB_OK;
}

extern "C" int pgb_destroy(pgb_handle* h) {
  if (!h) return PGB_OK;
  if (h->worker.joinable()) h->worker.join();
  // pgb_step_host returns while up to FEED_MIN idle slots are still queued on the sampler's stream (they write
  // the pinned flag words and the control records): nothing below may be freed under them, and torch's
  // streams are non-blocking, so no implicit ordering can be relied on
  (void)hipStreamSynchronize(h->stream);
  if (h->stream_out) (void)hipStreamSynchronize(h->stream_out);
  (void)hipGetLastError();
  for (int k = 0; k < PK_COUNT; ++k)
    for (hipEvent_t e : h->ev[k]) (void)hipEventDestroy(e);
  if (h->flag) (void)hipHostFree((void*)h->flag);
  if (h->out_host) (void)hipHostFree((void*)h->out_host);
  for (size_t i = 0; i < h->allocs.size(); ++i)
    if (h->alloc_owned[i]) (void)hipFree(h->allocs[i]);
  if (h->slab) (void)hipFree(h->slab);
  g_live_handles.fetch_sub(1);
  delete h;
  return PGB_OK;
}

// 16-bit order keys of p columns of a column-major matrix ([p][n_pad] doubles -> [p][n_pad] keys; see k_rows<..., F32>
// and k_key_*): one column at a time -- stage as float32, radix sort, equi-depth boundaries, keys.  Scratch:
// [n] staged | [n] sorted | boundaries | counter | the sort's temporary storage, allocated and freed here.
static hipError_t order_keys_build(const double* XT, long long n, long long n_pad, int p, uint16_t* keys, hipStream_t sm) {
  const size_t nn = (size_t)n;
  size_t tmp_bytes = 0;
  hipError_t e = hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, (const float*)nullptr, (float*)nullptr, (int)nn, 0, 32, sm);
  if (e != hipSuccess) return e;
  const size_t fl = (2 * nn + PGB_KEY_BOUNDS + 64) * sizeof(float);
  char* scratch = nullptr;
  e = hipMalloc((void**)&scratch, fl + tmp_bytes + 256);
  if (e != hipSuccess) return e;
  float* staged = (float*)scratch;
  float* sorted = staged + nn;
  float* bnd = sorted + nn;
  unsigned* miss = (unsigned*)(bnd + PGB_KEY_BOUNDS);
  void* tmp = scratch + fl;
  for (int c = 0; c < p && e == hipSuccess; ++c) {
    const double* col = XT + (size_t)c * n_pad;
    e = hipMemsetAsync(miss, 0, sizeof(unsigned), sm);
    if (e != hipSuccess) break;
    hipLaunchKernelGGL(k_key_stage, dim3(1024), dim3(BT), 0, sm, col, staged, n, miss);
    e = hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, (const float*)staged, sorted, (int)nn, 0, 32, sm);
    if (e != hipSuccess) break;
    hipLaunchKernelGGL(k_key_bounds, dim3(256), dim3(BT), 0, sm, (const float*)sorted, n, (const unsigned*)miss, bnd);
    hipLaunchKernelGGL(k_key_assign, dim3(2048), dim3(BT), 0, sm, col, (const float*)bnd, keys + (size_t)c * n_pad, n_pad);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(sm);
  (void)hipFree(scratch);
  return e;
}

extern "C" int pgb_set_data(pgb_handle* h, const double* X_dev, int64_t ldx, const int32_t* rules_host,
                            const double* split_prior_host) {
  if (!h || !X_dev || !rules_host || !split_prior_host) return fail(PGB_E_INVALID, "null argument");
  JOIN_ASYNC(h);
  h->out_valid = 0;
  Dev& d = h->d;
  if (ldx < d.p) return fail(PGB_E_INVALID, "ldx < p");
  double mx = 0.0;
  h->has_subset = 0;
  for (int j = 0; j < d.p; ++j) {
    if (rules_host[j] == PGB_RULE_SUBSET) h->has_subset = 1;
    if (rules_host[j] != PGB_RULE_CONTINUOUS && rules_host[j] != PGB_RULE_ONEHOT &&
        rules_host[j] != PGB_RULE_SUBSET)
      return fail(PGB_E_UNSUPPORTED, "unknown split rule");
    if (!(split_prior_host[j] > 0.0)) return fail(PGB_E_INVALID, "split_prior must be positive");
    if (split_prior_host[j] > mx) mx = split_prior_host[j];
  }
  d.max_prior = mx;
  d.alpha_unit = pgb_alpha_unit(mx);
  h->rules_host.assign(rules_host, rules_host + d.p);
  hipStream_t sm = h->stream;
  HIPCHK(hipMemcpyAsync((void*)d.rules, rules_host, d.p * sizeof(int32_t), hipMemcpyHostToDevice, sm));
  // the prior is staged in the (not yet used) running-sd buffer and quantised on the device
  HIPCHK(hipMemcpyAsync(d.rs_mean, split_prior_host, (size_t)(d.p < d.n_pad ? d.p : 0) * sizeof(double),
                        hipMemcpyHostToDevice, sm));
  HIPCHK(hipMemsetAsync((void*)d.col_nan, 0, d.p * sizeof(int32_t), sm));
  dim3 grid((unsigned)(d.n_pad / 32), (unsigned)((d.p + 31) / 32));
  hipLaunchKernelGGL(k_transpose, grid, dim3(BT), 0, sm, X_dev, (long long)ldx, (double*)d.XT, d.n,
                     d.n_pad, d.p, (int32_t*)d.col_nan);
  if (h->has_subset) {  // category codes outside 0 .. 51 are refused, not clamped (include/pgbart.h, Limits)
    int* bad_dev = nullptr;
    int bad = 0;
    HIPCHK(hipMalloc((void**)&bad_dev, sizeof(int)));
    hipError_t e = hipMemsetAsync(bad_dev, 0, sizeof(int), sm);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_subset_check, dim3((unsigned)d.p), dim3(BT), 0, sm, d.XT, d.n, d.n_pad, (const int32_t*)d.rules, bad_dev);
      e = hipMemcpyAsync(&bad, bad_dev, sizeof(int), hipMemcpyDeviceToHost, sm);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(sm);
    (void)hipFree(bad_dev);
    HIPCHK(e);
    if (bad) {
      char msg[160];
      snprintf(msg, sizeof msg, "SubsetSplit column %d: categories must be integer codes in [0, %d) (NaN = missing)", bad - 1,
               PGB_SUBSET_BITS);
      return fail(PGB_E_INVALID, msg);
    }
  }
  // A design matrix that does not fit the 256 MiB Infinity Cache streams from HBM in every row pass: the row pass
  // (single output, K = 2..4) then reads 16-bit order keys of the split column (k_rows<..., F32>: a quarter of the
  // bytes; built here, one column at a time: stage as float32, sort, equi-depth boundaries, keys); smaller matrices
  // stay on the float64 path (cache-resident, latency-bound: keys only add work).
  {
    size_t min_bytes = (size_t)192 << 20;
    if (const char* e = getenv("PGB_X32_MIN_MB")) min_bytes = (size_t)atoll(e) << 20;
    const size_t count = (size_t)d.p * d.n_pad;
    if (d.response == PGB_RESPONSE_CONSTANT && !h->has_subset && count * sizeof(double) >= min_bytes) {
      if (!d.XK16) {
        uint16_t* xk = nullptr;
        int rck = dalloc(h, &xk, count);
        if (rck != PGB_OK) return rck;
        d.XK16 = xk;
        h->rows_mk_cap = 0;  // (another instance of the K-vector row pass from here on)
      }
      hipError_t e = order_keys_build(d.XT, (long long)d.n, (long long)d.n_pad, d.p, (uint16_t*)d.XK16, sm);
      if (e != hipSuccess) return fail_hip(e, "order keys of the design matrix");
    }
  }
  double* prior_stage = nullptr;
  if (d.p >= d.n_pad) {  // more columns than padded rows: stage through a temporary
    HIPCHK(hipMalloc((void**)&prior_stage, d.p * sizeof(double)));
    HIPCHK(hipMemcpyAsync(prior_stage, split_prior_host, d.p * sizeof(double), hipMemcpyHostToDevice, sm));
  }
  hipLaunchKernelGGL(k_init_alpha, dim3(1), dim3(64), 0, sm, prior_stage ? prior_stage : d.rs_mean, mx,
                     d.alpha, d.cdfS, d.p);
  if (d.response != PGB_RESPONSE_CONSTANT) {  // exponent bound of every column (u = x 2^-ex)
    double* amax_dev = nullptr;
    HIPCHK(hipMalloc((void**)&amax_dev, d.p * sizeof(double)));
    hipLaunchKernelGGL(k_colmax, dim3((unsigned)d.p), dim3(BT), 0, sm, d.XT, d.n, d.n_pad, amax_dev);
    std::vector<double> amax(d.p);
    HIPCHK(hipMemcpyAsync(amax.data(), amax_dev, d.p * sizeof(double), hipMemcpyDeviceToHost, sm));
    HIPCHK(hipStreamSynchronize(sm));
    (void)hipFree(amax_dev);
    std::vector<int32_t> ex(d.p);
    for (int j = 0; j < d.p; ++j) ex[j] = pgb_col_exponent(amax[j]);
    HIPCHK(hipMemcpyAsync((void*)d.col_ex, ex.data(), d.p * sizeof(int32_t), hipMemcpyHostToDevice, sm));
    HIPCHK(hipStreamSynchronize(sm));
  }
  HIPCHK(hipMemsetAsync(d.rs_mean, 0, d.n_pad * sizeof(double), sm));
  HIPCHK(hipMemcpyAsync(h->d_dev, &d, sizeof(Dev), hipMemcpyHostToDevice, sm));  // alpha_unit, max_prior
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(sm));
  if (prior_stage) (void)hipFree(prior_stage);
  h->have_data = 1;
  return PGB_OK;
}

extern "C" int pgb_set_response(pgb_handle* h, const double* y_dev) {
  if (!h || !y_dev) return fail(PGB_E_INVALID, "null argument");
  JOIN_ASYNC(h);
  h->out_valid = 0;
  HIPCHK(hipMemcpyAsync((void*)h->d.y, y_dev, h->d.n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (h->s.family == PGB_FAMILY_CALLBACK)
    HIPCHK(hipMemcpyAsync(h->y_host.data(), y_dev, h->d.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  h->flag[4] = 0;
  hipLaunchKernelGGL(k_nonfinite, dim3(256), dim3(BT), 0, h->stream, (const double*)h->d.y, (long long)h->d.n,
                     (long long)h->d.n_pad, 1, __builtin_inf(), h->d.host_flag + 4);
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->flag[4]) {
    h->have_y = 0;
    return fail(PGB_E_INVALID, "the response has non-finite values");
  }
  h->have_y = 1;
  return PGB_OK;
}

extern "C" int pgb_set_offset(pgb_handle* h, const double* offset_dev) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  JOIN_ASYNC(h);
  h->out_valid = 0;
  if (h->s.family == PGB_FAMILY_NORMAL)
    return fail(PGB_E_UNSUPPORTED, "offsets are for the per-row families (a Normal model fits observed - offset)");
  if (offset_dev)  // [K][n] -> the padded [K][n_pad] rows
    HIPCHK(hipMemcpy2DAsync((void*)h->d.off, h->d.n_pad * sizeof(double), offset_dev, h->d.n * sizeof(double),
                            h->d.n * sizeof(double), (size_t)h->d.K, hipMemcpyDeviceToDevice, h->stream));
  else
    HIPCHK(hipMemsetAsync((void*)h->d.off, 0, (size_t)h->d.K * h->d.n_pad * sizeof(double), h->stream));
  if (h->s.family == PGB_FAMILY_CALLBACK) {
    if (offset_dev)
      HIPCHK(hipMemcpyAsync(h->off_host.data(), offset_dev, h->d.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    else
      h->off_host.assign((size_t)h->d.n, 0.0);
  }
  if (h->d.has_off != (offset_dev ? 1 : 0)) {
    h->d.has_off = offset_dev ? 1 : 0;
    HIPCHK(hipMemcpyAsync(h->d_dev, &h->d, sizeof(Dev), hipMemcpyHostToDevice, h->stream));
  }
  h->flag[4] = 0;
  if (offset_dev)
    hipLaunchKernelGGL(k_nonfinite, dim3(256), dim3(BT), 0, h->stream, (const double*)h->d.off, (long long)h->d.n,
                       (long long)h->d.n_pad, (int)h->d.K, (double)PGB_MAX_OFFSET, h->d.host_flag + 4);
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->flag[4]) {  // (a linear predictor must be finite; the offset is zeroed so that the chain stays usable)
    HIPCHK(hipMemsetAsync((void*)h->d.off, 0, (size_t)h->d.K * h->d.n_pad * sizeof(double), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return fail(PGB_E_INVALID, "the offset has non-finite values or values beyond +-1e6 (PGB_MAX_OFFSET)");
  }
  return PGB_OK;
}

extern "C" int pgb_set_likelihood(pgb_handle* h, const double* params, int32_t n_params) {
  if (!h || !params) return fail(PGB_E_INVALID, "null argument");
  JOIN_ASYNC(h);
  if (h->s.family == PGB_FAMILY_NORMAL) {
    if (n_params != 1 || !(params[0] > 0.0)) return fail(PGB_E_INVALID, "NORMAL needs sigma > 0");
    h->inv_sigma2 = 1.0 / (params[0] * params[0]);
    h->sigma_dirty = 1;
  } else if (h->s.family == PGB_FAMILY_NEGBIN_LOG || h->s.family == PGB_FAMILY_GAMMA_LOG) {
    // the slot doubles as "the family's parameter"
    if (n_params != 1 || !(params[0] > 0.0)) return fail(PGB_E_INVALID, "NEGBIN_LOG / GAMMA_LOG need alpha > 0");
    h->inv_sigma2 = params[0];
    h->sigma_dirty = 1;
  } else if (h->s.family == PGB_FAMILY_ASYMLAPLACE) {
    if (n_params != 2 || !(params[0] > 0.0) || !(params[1] > 0.0) || !(params[1] < 1.0))
      return fail(PGB_E_INVALID, "ASYMLAPLACE needs b > 0 and 0 < q < 1");
    h->inv_sigma2 = params[0];
    h->lik_param2 = params[1];
    h->sigma_dirty = 1;
  } else if (h->s.family == PGB_FAMILY_STUDENT_T) {
    if (n_params != 2 || !(params[0] > 0.0) || !(params[1] > 0.0))
      return fail(PGB_E_INVALID, "STUDENT_T needs sigma > 0 and nu > 0");
    h->inv_sigma2 = params[0];
    h->lik_param2 = params[1];
    h->sigma_dirty = 1;
  } else if (n_params != 0) {
    return fail(PGB_E_INVALID, "this family has no parameters");
  }
  return PGB_OK;
}

// Profiling: the events are attached to the dispatch itself (hipExtLaunchKernelGGL), i.e. they carry
// the start / end timestamps of the kernel's own AQL packet -- the interval rocprofv3 reports --
// rather than bracketing the launch with two extra barrier packets.
static int prof_events(pgb_handle* h, int k, hipEvent_t* e0, hipEvent_t* e1) {
  *e0 = *e1 = nullptr;
  if (!h->prof) return PGB_OK;
  if (h->ev_used[k] + 2 > h->ev[k].size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)
      return fail(PGB_E_DEVICE, "hipEventCreate");
    h->ev[k].push_back(a);
    h->ev[k].push_back(b);
  }
  *e0 = h->ev[k][h->ev_used[k]];
  *e1 = h->ev[k][h->ev_used[k] + 1];
  h->ev_used[k] += 2;
  return PGB_OK;
}
#define LAUNCH_KT(PK_, KERN, GRID_, THREADS_, ...)                                               \
  do {                                                                                           \
    hipEvent_t e0_, e1_;                                                                         \
    int rc_ = prof_events(h, (PK_), &e0_, &e1_);                                                 \
    if (rc_ != PGB_OK) return rc_;                                                               \
    h->prof_wgs[(PK_)] = (int)(GRID_).x;                                                         \
    if (h->prof) hipExtLaunchKernelGGL((KERN), (GRID_), dim3(THREADS_), 0, h->stream, e0_, e1_, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERN), (GRID_), dim3(THREADS_), 0, h->stream, __VA_ARGS__);          \
  } while (0)
#define LAUNCH_K(PK_, KERN, GRID_, ...) LAUNCH_KT(PK_, KERN, GRID_, BT, __VA_ARGS__)

static int enqueue_slots(pgb_handle* h, int count) {
  Dev& d = h->d;
  const bool lin = d.response != PGB_RESPONSE_CONSTANT;
  long long want = (long long)d.nchunks * (d.P - 1);
  if (want < d.n_pad / BT) want = d.n_pad / BT;
  if (want > h->rows_grid) want = h->rows_grid;
  if (d.K > 1) {
    // the K-vector instances differ in registers (K = 4: 3 workgroups per CU, the run-time-K instance 2-3): a
    // persistent grid larger than what stays resident leaves the surplus workgroups waiting for a first round of
    // items to finish (cfg5, in-kernel stamps: the last quarter of a 1024 grid started 22 us into a 27 us launch)
    if (h->rows_mk_cap == 0) {
      const bool f32 = d.XK16 != nullptr;
      const void* kf = lin ? (const void*)k_rows_mk<0, true>
                       : d.K == 2 ? (f32 ? (const void*)k_rows_mk<2, false, true> : (const void*)k_rows_mk<2, false>)
                       : d.K == 3 ? (f32 ? (const void*)k_rows_mk<3, false, true> : (const void*)k_rows_mk<3, false>)
                       : d.K == 4 ? (f32 ? (const void*)k_rows_mk<4, false, true> : (const void*)k_rows_mk<4, false>)
                                  : (f32 ? (const void*)k_rows_mk<0, false, true> : (const void*)k_rows_mk<0, false>);
      int per_cu = 0, cus = 0;
      h->rows_mk_cap = h->rows_grid;
      if (!getenv("PGB_ROWS_GRID") && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kf, BT, 0) == hipSuccess &&
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && per_cu > 0 && cus > 0)
        h->rows_mk_cap = per_cu * cus < h->rows_grid ? per_cu * cus : h->rows_grid;
    }
    if (want > h->rows_mk_cap) want = h->rows_mk_cap;
  }
  dim3 gctrl((unsigned)(d.P - 1)), grows((unsigned)want);
  long long wantl = (long long)d.nchunks * (d.P - 1);
  if (wantl > h->ll_grid) wantl = h->ll_grid;
  dim3 gll((unsigned)wantl);
  const Dev* dd = (const Dev*)h->d_dev;
  for (int i = 0; i < count; ++i) {
    int par = (int)(h->slot & 1);
#define CTRL_ARGS(nwg) dd, par, (int)(nwg), d.ctrl, (const InitAcc*)d.initacc, (const Job*)d.jobs, (const Acc*)d.acc, (const DPart*)d.parts, (const uint16_t*)d.XK16
    if (d.K > 1 && lin) LAUNCH_K(PK_CTRL, (k_ctrl<true, true>), gctrl, CTRL_ARGS(gctrl.x));
    else if (d.K > 1 && d.XK16) LAUNCH_K(PK_CTRL, (k_ctrl<true, false, true>), gctrl, CTRL_ARGS(gctrl.x));
    else if (d.K > 1) LAUNCH_K(PK_CTRL, (k_ctrl<true, false>), gctrl, CTRL_ARGS(gctrl.x));
    else if (lin) LAUNCH_K(PK_CTRL, (k_ctrl<false, true>), gctrl, CTRL_ARGS(gctrl.x));
    else if (d.XK16) LAUNCH_K(PK_CTRL, (k_ctrl<false, false, true>), dim3((unsigned)d.P), CTRL_ARGS(d.P));
    else LAUNCH_K(PK_CTRL, (k_ctrl<false, false>), dim3((unsigned)d.P), CTRL_ARGS(d.P));  // + the workgroup that builds the label tables ahead
#undef CTRL_ARGS
#define ROWS_ARGS dd, par, (const Cmd*)d.cmd, (const Job*)d.jobs
    if (d.K > 1 && lin) {  // linear leaves: one instance for any K
      LAUNCH_K(PK_ROWS, (k_rows_mk<0, true>), grows, dd, par, (const Cmd*)d.cmd, (const Job*)d.jobs);
    } else if (d.K == 2) {
      if (d.XK16) LAUNCH_K(PK_ROWS, (k_rows_mk<2, false, true>), grows, dd, par, (const Cmd*)d.cmd, (const Job*)d.jobs);
      else LAUNCH_K(PK_ROWS, (k_rows_mk<2, false>), grows, dd, par, (const Cmd*)d.cmd, (const Job*)d.jobs);
    } else if (d.K == 3) {
      if (d.XK16) LAUNCH_K(PK_ROWS, (k_rows_mk<3, false, true>), grows, dd, par, (const Cmd*)d.cmd, (const Job*)d.jobs);
      else LAUNCH_K(PK_ROWS, (k_rows_mk<3, false>), grows, dd, par, (const Cmd*)d.cmd, (const Job*)d.jobs);
    } else if (d.K == 4) {
      if (d.XK16) LAUNCH_K(PK_ROWS, (k_rows_mk<4, false, true>), grows, dd, par, (const Cmd*)d.cmd, (const Job*)d.jobs);
      else LAUNCH_K(PK_ROWS, (k_rows_mk<4, false>), grows, dd, par, (const Cmd*)d.cmd, (const Job*)d.jobs);
    } else if (d.K > 1) {
      if (d.XK16) LAUNCH_K(PK_ROWS, (k_rows_mk<0, false, true>), grows, dd, par, (const Cmd*)d.cmd, (const Job*)d.jobs);
      else LAUNCH_K(PK_ROWS, (k_rows_mk<0, false>), grows, dd, par, (const Cmd*)d.cmd, (const Job*)d.jobs);
    } else {
      const bool nrm = h->s.family == PGB_FAMILY_NORMAL;
      if (lin && h->has_subset) {  // (a linear leaf regresses on whatever column its parent split on)
        if (nrm) LAUNCH_K(PK_ROWS, (k_rows<true, true, true>), grows, ROWS_ARGS);
        else LAUNCH_K(PK_ROWS, (k_rows<true, false, true>), grows, ROWS_ARGS);
      } else if (lin) {
        if (nrm) LAUNCH_K(PK_ROWS, (k_rows<false, true, true>), grows, ROWS_ARGS);
        else LAUNCH_K(PK_ROWS, (k_rows<false, false, true>), grows, ROWS_ARGS);
      } else if (h->has_subset) {
        if (nrm) LAUNCH_K(PK_ROWS, (k_rows<true, true, false>), grows, ROWS_ARGS);
        else LAUNCH_K(PK_ROWS, (k_rows<true, false, false>), grows, ROWS_ARGS);
      } else if (d.XK16 != nullptr) {  // 16-bit order keys of the split columns (matrix larger than the Infinity Cache)
        if (nrm) LAUNCH_K(PK_ROWS, (k_rows<false, true, false, true>), grows, ROWS_ARGS);
        else LAUNCH_K(PK_ROWS, (k_rows<false, false, false, true>), grows, ROWS_ARGS);
      } else {
        if (nrm) LAUNCH_K(PK_ROWS, (k_rows<false, true, false>), grows, ROWS_ARGS);
        else LAUNCH_K(PK_ROWS, (k_rows<false, false, false>), grows, ROWS_ARGS);
      }
    }
#undef ROWS_ARGS
    if (d.family != PGB_FAMILY_NORMAL)  // per-row log-likelihood of the rows this round re-labelled
      LAUNCH_K(PK_LL, h->ll_kernel, gll, dd, par, (int)gll.x, (const Cmd*)d.cmd, (const Ctrl*)d.ctrl, (const Job*)d.jobs,
               (const Acc*)d.acc, (const InitAcc*)d.initacc);
    h->slot += 1;
  }
  HIPCHK(hipGetLastError());
  return PGB_OK;
}

static int harvest_profile(pgb_handle* h) {
  for (int k = 0; k < PK_COUNT; ++k) {
    for (size_t i = 0; i + 1 < h->ev_used[k]; i += 2) {
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, h->ev[k][i], h->ev[k][i + 1]));
      h->prof_ms[k] += ms;
      h->prof_launches[k] += 1;
    }
    h->ev_used[k] = 0;
  }
  return PGB_OK;
}

// ---- callback family: the host half of a slot (after {k_ctrl ; k_rows ; k_loglik<callback>} have run)
// k_loglik left, per active particle, every row's side and the linear predictor of the rows of the split
// leaf; the callback evaluates them in ONE call, the values are clamped and quantised like the built-in
// families' and their sums per (particle, side) go where k_loglik would have put them.  A slot that
// starts a tree also needs the log-likelihood of a fresh stump and of the current tree over all rows
// (what the INIT part of k_rows computes for the built-in families).
static int callback_host_phase(pgb_handle* h, int par) {
  Dev& d = h->d;
  const size_t n = (size_t)d.n;
  struct CmdHead { int32_t kind, tree_old, tree_new, sel_gen, sel_slot, tune, dst_gen, st_cur; } ch;
  HIPCHK(hipMemcpy(&ch, &d.cmd[par], sizeof ch, hipMemcpyDeviceToHost));
  if (!(ch.kind & CMD_PARTITION)) return PGB_OK;
  std::vector<Job> jobs(MAXP);
  HIPCHK(hipMemcpy(jobs.data(), d.jobs + (size_t)par * MAXP, sizeof(Job) * MAXP, hipMemcpyDeviceToHost));
  std::vector<double> yy, mm, out;
  std::vector<int64_t> rows;
  std::vector<int> owner;  // (particle << 2) | side of every gathered row
  std::vector<uint8_t> side(n);
  std::vector<double> mu(n);
  for (int p = 1; p < d.P; ++p) {
    if (!jobs[p].active) continue;
    HIPCHK(hipMemcpy(side.data(), d.cb_side + (size_t)p * d.n_pad, n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(mu.data(), d.cb_mu + (size_t)p * d.n_pad, n * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i)
      if (side[i] < 3) {
        rows.push_back((int64_t)i);
        yy.push_back(h->y_host[i]);
        mm.push_back(mu[i]);
        owner.push_back((p << 2) | side[i]);
      }
  }
  size_t n_part = yy.size();
  const bool init = (ch.kind & CMD_INIT) != 0;
  if (init) {  // stump and current tree over all rows: [n_part, n_part + n) stump, [n_part + n, n_part + 2n) current
    std::vector<double2> pack(n);
    std::vector<double> noi(n);
    HIPCHK(hipMemcpy(pack.data(), d.pack, n * sizeof(double2), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(noi.data(), d.st + (size_t)(ch.st_cur ^ 1) * d.n_pad, n * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
      rows.push_back((int64_t)i);
      yy.push_back(h->y_host[i]);
      mm.push_back((noi[i] + h->off_host[i]) + d.init_leaf);
    }
    for (size_t i = 0; i < n; ++i) {
      rows.push_back((int64_t)i);
      yy.push_back(h->y_host[i]);
      mm.push_back(pack[i].x + h->off_host[i]);
    }
  }
  if (yy.empty()) return PGB_OK;
  out.assign(yy.size(), 0.0);
  if (h->cb_fn(h->cb_ctx, rows.data(), yy.data(), mm.data(), (int64_t)yy.size(), out.data()) != 0)
    return fail(PGB_E_STATE, "the log-likelihood callback reported an error");
  unsigned sat = 0;
  std::vector<AccL> sums(MAXP, AccL{0, 0, 0, 0});
  for (size_t i = 0; i < n_part; ++i) {
    const long long q = pgb_quant(pgb_clamp_loglik(out[i]), d.sc.cl, &sat);
    AccL& a = sums[owner[i] >> 2];
    const int sd = owner[i] & 3;
    if (sd == 0) a.llL += q; else if (sd == 1) a.llR += q; else a.llN += q;
  }
  for (int p = 1; p < d.P; ++p)
    if (jobs[p].active)  // copy 0 of the particle's record (k_ctrl zeroed all copies in this slot)
      HIPCHK(hipMemcpy(&d.accl[((size_t)par * MAXP + p) * LL_PER], &sums[p], sizeof(AccL), hipMemcpyHostToDevice));
  if (init) {
    long long C = 0, E0 = 0;
    for (size_t i = 0; i < n; ++i) C += pgb_quant(pgb_clamp_loglik(out[n_part + i]), d.sc.cl, &sat);
    for (size_t i = 0; i < n; ++i) E0 += pgb_quant(pgb_clamp_loglik(out[n_part + n + i]), d.sc.cl, &sat);
    InitAcc ia;
    HIPCHK(hipMemcpy(&ia, &d.initacc[(size_t)par * IA_SLOTS], sizeof ia, hipMemcpyDeviceToHost));
    ia.C += C;
    ia.E0 += E0;
    HIPCHK(hipMemcpy(&d.initacc[(size_t)par * IA_SLOTS], &ia, sizeof ia, hipMemcpyHostToDevice));
  }
  if (sat) {
    unsigned long long cs = 0;
    HIPCHK(hipMemcpy(&cs, &d.counters[4], sizeof cs, hipMemcpyDeviceToHost));
    cs += sat;
    HIPCHK(hipMemcpy(&d.counters[4], &cs, sizeof cs, hipMemcpyHostToDevice));
  }
  return PGB_OK;
}

// ---- feeding the state machine ---------------------------------------------------------------------------
// The device publishes two progress words besides the step count: flag[2] = slots whose control kernel has
// started (Ctrl::slot_no, one posted store per slot) and flag[1] = steps that are complete.  The host
// enqueues against flag[2] like against a credit: up to FEED_AHEAD slots beyond the executed one while the
// step is expected to need them (running mean - 1 sd of the slots such a call took), FEED_MIN beyond it
// after that -- so a finished step leaves at most a few idle slots (~2.6 us each) behind it.  No events, no
// barrier packets: the throttle costs the device nothing and the host one read of pinned memory.
static double watchdog_seconds() {
  static const double v = getenv("PGB_WATCHDOG_S") ? atof(getenv("PGB_WATCHDOG_S")) : 20.0;
  return v > 0.1 ? v : 0.1;
}
#define FEED_AHEAD 24
#define FEED_MIN 6
#define FEED_RUN 8
static int feed_until_flag(pgb_handle* h, int n_steps) {
  Dev& d = h->d;
  const long long start = h->slot;
  long long cap = start + (long long)n_steps * (PGB_MAX_NODES + 3) * (d.m + 1) + 64;
  int rc;
  if (h->s.family == PGB_FAMILY_CALLBACK) {  // one slot at a time, the host evaluates between slots
    if (!h->cb_fn) return fail(PGB_E_INVALID, "pgb_set_loglik_callback first");
    while (*h->flag < (unsigned long long)h->steps_target) {
      const int par = (int)(h->slot & 1);
      if ((rc = enqueue_slots(h, 1)) != PGB_OK) return rc;
      HIPCHK(hipStreamSynchronize(h->stream));
      if ((rc = callback_host_phase(h, par)) != PGB_OK) {
        h->poisoned = 1;  // steps_target is ahead of the flag and the device sits in the middle of a round
        return rc;
      }
      if (h->slot > cap) {
        h->poisoned = 1;
        return fail(PGB_E_STATE, "sampler state machine did not finish");
      }
    }
    return PGB_OK;
  }
  static const int feed_ahead = getenv("PGB_FEED_AHEAD") ? atoi(getenv("PGB_FEED_AHEAD")) : FEED_AHEAD;
  static const int feed_min = getenv("PGB_FEED_MIN") ? atoi(getenv("PGB_FEED_MIN")) : FEED_MIN;
  static const double feed_sd = getenv("PGB_FEED_SD") ? atof(getenv("PGB_FEED_SD")) : 1.0;
  // slots this call is expected to need at least: mean - feed_sd * sd of the running per-step estimate
  long long expect_end = 1ll << 60;
  if (h->slots_per_step > 0.0) {
    const double sd = h->slots_var > 0.0 ? __builtin_sqrt(h->slots_var) : 0.15 * h->slots_per_step;
    double e = h->slots_per_step * n_steps - feed_sd * sd * __builtin_sqrt((double)n_steps);
    expect_end = start + (long long)(e > 1.0 ? e : 1.0);
  }
  long long idle_polls = 0;
  int refill = 0;
  // lost-device watchdog by the WALL clock (a poll count means different times on different hosts, and a false
  // trip poisons the chain): PGB_WATCHDOG_S seconds without a single slot starting; the deadline moves whenever
  // the device publishes progress
  long long seen = (long long)h->flag[2];
  auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds((long long)(watchdog_seconds() * 1e3));
  while (*h->flag < (unsigned long long)h->steps_target) {
    const long long executed = (long long)h->flag[2];
    long long limit = executed + feed_ahead;
    if (limit > expect_end) limit = expect_end;
    if (limit < executed + feed_min) limit = executed + feed_min;
    // far from the expected end the queue is topped up in runs of FEED_RUN slots (several chains share the
    // runtime's launch path: slot-by-slot refills of four chains at once cost them 30 % of their aggregate);
    // near it, slot by slot
    if (refill == 0 && h->slot < limit) {
      const bool far = limit == executed + feed_ahead;
      if (!far || h->slot - executed <= feed_ahead - FEED_RUN) refill = far ? FEED_RUN : 1;
    }
    if (refill > 0 && h->slot < limit) {
      refill -= 1;
      if ((rc = enqueue_slots(h, 1)) != PGB_OK) return rc;
      if (h->slot > cap) {
        h->poisoned = 1;
        return fail(PGB_E_STATE, "sampler state machine did not finish");
      }
      idle_polls = 0;
    } else {
      refill = 0;
      for (int i = 0; i < 16; ++i) __builtin_ia32_pause();
      // several chains feed their own streams from their own threads (chains.sample_chains, bench.py's
      // concurrent chains): a waiting feeder gives its core away instead of spinning against the others
      if (g_live_handles.load(std::memory_order_relaxed) > 1) sched_yield();
      // a device that stops publishing progress (lost GPU, a kernel that faulted) ends the call, not the process
      if ((++idle_polls & 0x3FFF) == 0) {
        const auto now = std::chrono::steady_clock::now();
        if ((long long)h->flag[2] != seen) {
          seen = (long long)h->flag[2];
          deadline = now + std::chrono::milliseconds((long long)(watchdog_seconds() * 1e3));
        } else if (now > deadline) {
          hipError_t e = hipStreamQuery(h->stream);
          h->poisoned = 1;
          if (e != hipSuccess && e != hipErrorNotReady) return fail_hip(e, "device stopped making progress");
          return fail(PGB_E_STATE, "device stopped publishing slot progress");
        }
      }
    }
  }
  return PGB_OK;
}

static int run_until_idle(pgb_handle* h, int n_steps) {
  Dev& d = h->d;
  int rc;
  if ((rc = feed_until_flag(h, n_steps)) != PGB_OK) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->prof && (rc = harvest_profile(h)) != PGB_OK) return rc;
  Ctrl c;
  HIPCHK(hipMemcpy(&c, &d.ctrl[h->slot & 1], sizeof c, hipMemcpyDeviceToHost));
  if (c.phase != PH_IDLE) return fail(PGB_E_STATE, "device not idle after the progress flag fired");
  h->st_cur = c.st_cur;
  h->alpha_cur = c.alpha_cur;
  return PGB_OK;
}

static int begin_steps(pgb_handle* h, int tune, int n_steps) {
  Dev& d = h->d;
  if (!h->have_data || !h->have_y) return fail(PGB_E_INVALID, "set_data/set_response first");
  REFUSE_POISONED(h);
  int par = (int)(h->slot & 1);
  h->out_valid = 0;
  hipLaunchKernelGGL(k_begin, dim3(1), dim3(256), 0, h->stream, h->d_dev, par, tune, n_steps,
                     h->inv_sigma2, h->lik_param2, h->sigma_dirty);
  h->steps_target += n_steps;
  h->sigma_dirty = 0;
  // host mirror of the batch cursor ([U] PGBART.astep batching)
  for (int i = 0; i < n_steps; ++i) {
    int bs = tune ? d.batch_tune : d.batch_draw;
    int upper = h->lower_host + bs;
    if (upper > d.m) upper = d.m;
    h->last_lower = h->lower_host;
    h->last_n = upper - h->lower_host;
    h->lower_host = upper < d.m ? upper : 0;
  }
  return PGB_OK;
}

static void counters_from(pgb_handle* h, const unsigned long long* c) {
  h->ctr.particle_steps = (int64_t)c[0];
  h->ctr.tree_updates = (int64_t)c[1];
  h->ctr.rows_touched = (int64_t)c[2];
  h->ctr.rounds = (int64_t)c[3];
  h->ctr.saturations = (int64_t)c[4];
  h->ctr.slots = (int64_t)c[5];
  h->ctr.partitions = (int64_t)c[6];
}

// running estimate of the (working) slots one astep needs, from the device's own count
static void note_step_slots(pgb_handle* h, long long slots_before, int n_steps) {
  const double per = (double)(h->ctr.slots - slots_before) / (double)(n_steps > 0 ? n_steps : 1);
  if (per <= 0.0) return;
  if (h->slots_per_step <= 0.0) {
    h->slots_per_step = per;
    return;
  }
  // exponentially weighted mean / variance (weight 1/8); a batch of n steps averages n of them
  const double dev = per - h->slots_per_step;
  h->slots_per_step += 0.125 * dev;
  h->slots_var = 0.875 * (h->slots_var + 0.125 * dev * dev * (n_steps > 0 ? n_steps : 1));
}

static int fetch_counters(pgb_handle* h, pgb_counters* out) {
  unsigned long long c[8];
  HIPCHK(hipMemcpyAsync(c, h->d.counters, sizeof c, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  counters_from(h, c);
  if (out) *out = h->ctr;
  return PGB_OK;
}

extern "C" int pgb_set_output_stream(pgb_handle* h, void* stream) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  JOIN_ASYNC(h);
  h->stream_out = (hipStream_t)stream;  // owned by the caller
  return PGB_OK;
}

extern "C" int pgb_set_loglik_callback(pgb_handle* h, pgb_loglik_fn fn, void* ctx) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  JOIN_ASYNC(h);
  if (h->s.family != PGB_FAMILY_CALLBACK)
    return fail(PGB_E_INVALID, "the sampler was not created with the callback family");
  h->cb_fn = fn;
  h->cb_ctx = ctx;
  return PGB_OK;
}

extern "C" int pgb_step(pgb_handle* h, int32_t tune, double* sum_trees_dev_out, int32_t* vi_counts_host_out,
                        pgb_counters* counters_out) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  JOIN_ASYNC(h);
  int rc;
  const long long slots0 = h->ctr.slots;
  if ((rc = begin_steps(h, tune, 1)) != PGB_OK) return rc;
  if ((rc = run_until_idle(h, 1)) != PGB_OK) return rc;
  if (sum_trees_dev_out)  // [K][n] out of the padded [K][n_pad] buffer
    HIPCHK(hipMemcpy2DAsync(sum_trees_dev_out, h->d.n * sizeof(double),
                            h->d.st + (size_t)h->st_cur * h->d.K * h->d.n_pad, h->d.n_pad * sizeof(double),
                            h->d.n * sizeof(double), (size_t)h->d.K, hipMemcpyDeviceToDevice, h->stream));
  if (vi_counts_host_out)
    HIPCHK(hipMemcpyAsync(vi_counts_host_out, h->d.vi, h->d.p * sizeof(int32_t), hipMemcpyDeviceToHost,
                          h->stream));
  if ((rc = fetch_counters(h, counters_out)) != PGB_OK) return rc;
  note_step_slots(h, slots0, 1);
  return PGB_OK;
}

// PGBART.astep's return path in one device -> host transaction: the state machine runs to its idle
// point, k_export_step writes the small results (trees of this step, vi, counters, control words)
// into the mapped pinned block and densifies sum_trees, one DMA moves sum_trees to the caller,
// ONE stream synchronisation ends the call.
extern "C" int pgb_step_host(pgb_handle* h, int32_t tune, double* sum_trees_host_out,
                             int32_t* vi_counts_host_out, pgb_counters* counters_out) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  JOIN_ASYNC(h);
  Dev& d = h->d;
  int rc;
  const long long slots0 = h->ctr.slots;
  if ((rc = begin_steps(h, tune, 1)) != PGB_OK) return rc;
  if ((rc = feed_until_flag(h, 1)) != PGB_OK) return rc;
  const int nt = h->last_n;
  if (nt > h->out_layout.cap_trees) return fail(PGB_E_STATE, "step batch exceeds the export block");
  // The progress word fires when the last tree is accepted; its FINAL row pass runs in that same slot, and the
  // first idle slot behind it publishes "step complete" (flag[1]).  From there the results leave on a stream of
  // their own: the idle slots still queued on the sampler's stream (a few, ~2.6 us each) drain
  // while the export, the DMA and the caller's host work go on.
  hipStream_t so = h->stream;
  const bool early = h->s.family != PGB_FAMILY_CALLBACK && !h->prof && h->stream_out != nullptr;
  if (early) {
    int extra = 0;
    long long polls = 0;
    auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds((long long)(watchdog_seconds() * 1e3));
    long long seen2 = (long long)h->flag[2];
    while (h->flag[1] < (unsigned long long)h->steps_target) {
      // (a device that faults between "last tree accepted" and the first idle slot would otherwise hold this
      //  loop for ever, with the GIL released.  As in feed_until_flag the deadline MOVES while the device
      //  publishes progress -- slots starting -- so that a FINAL pass that takes longer than PGB_WATCHDOG_S on a
      //  time-sliced GPU, under a profiler or at very large n does not poison a healthy chain: round-4 ADVICE)
      if ((++polls & 0x3FFF) == 0) {
        const auto now = std::chrono::steady_clock::now();
        if ((long long)h->flag[2] != seen2) {
          seen2 = (long long)h->flag[2];
          deadline = now + std::chrono::milliseconds((long long)(watchdog_seconds() * 1e3));
        } else if (now > deadline) {
          hipError_t e = hipStreamQuery(h->stream);
          h->poisoned = 1;
          if (e != hipSuccess && e != hipErrorNotReady) return fail_hip(e, "device stopped before the step-complete word");
          return fail(PGB_E_STATE, "the step-complete word never fired");
        }
      }
      // every enqueued slot has started and none of them was an idle one behind the finishing slot: one more
      if ((long long)h->flag[2] >= h->slot) {
        if (h->flag[1] >= (unsigned long long)h->steps_target) break;
        if (++extra > 16) { h->poisoned = 1; return fail(PGB_E_STATE, "the step-complete word never fired"); }
        if ((rc = enqueue_slots(h, 1)) != PGB_OK) return rc;
      }
      for (int i = 0; i < 16; ++i) __builtin_ia32_pause();
    }
    so = h->stream_out;
  }
  const long long dense_blocks = sum_trees_host_out ? ((long long)d.K * d.n + BT - 1) / BT : 0;
  long long grid = dense_blocks < 512 ? dense_blocks : 512;
  if (grid < nt) grid = nt;
  if (grid < 1) grid = 1;
  // sum_trees goes straight into the caller's buffer when that is device-accessible pinned memory (what
  // PySampler hands in): the export kernel writes it over PCIe itself -- one kernel instead of a densify
  // kernel, a copy kernel and the gap between them.  Pageable memory: densify in HBM, then one DMA.
  double* direct = nullptr;
  if (sum_trees_host_out && !(getenv("PGB_NO_DIRECT_OUT") && atoi(getenv("PGB_NO_DIRECT_OUT")))) {
    // (asked every time: the same address may be pinned in one call and ordinary memory in the next)
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, sum_trees_host_out, 0) == hipSuccess && dp) direct = (double*)dp;
    else (void)hipGetLastError();  // not pinned: the DMA path
  }
  hipLaunchKernelGGL(k_export_step, dim3((unsigned)grid), dim3(BT), 0, so, (const Dev*)h->d_dev,
                     (int)(h->slot & 1), h->last_lower, nt, h->out_dev, h->out_layout,
                     sum_trees_host_out ? (direct ? direct : h->st_dense) : nullptr);
  if (sum_trees_host_out && !direct)
    HIPCHK(hipMemcpyAsync(sum_trees_host_out, h->st_dense, (size_t)d.K * d.n * sizeof(double),
                          hipMemcpyDeviceToHost, so));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(so));
  if (h->prof && (rc = harvest_profile(h)) != PGB_OK) return rc;
  const StepOutHdr* H = (const StepOutHdr*)h->out_host;
  if (H->phase != PH_IDLE) return fail(PGB_E_STATE, "device not idle after the progress flag fired");
  h->st_cur = H->st_cur;
  h->alpha_cur = H->alpha_cur;
  counters_from(h, H->counters);
  note_step_slots(h, slots0, 1);
  h->out_valid = 1;
  if (vi_counts_host_out) memcpy(vi_counts_host_out, h->out_host + h->out_layout.vi, (size_t)d.p * sizeof(int32_t));
  if (counters_out) *counters_out = h->ctr;
  return PGB_OK;
}

extern "C" int pgb_step_async(pgb_handle* h, int32_t tune, int32_t n_steps) {
  if (!h || n_steps < 1) return fail(PGB_E_INVALID, "bad argument");
  JOIN_ASYNC(h);
  if (!h->have_data || !h->have_y) return fail(PGB_E_INVALID, "set_data/set_response first");
  REFUSE_POISONED(h);
  h->job_running = 1;
  h->job_rc = PGB_OK;
  h->job_err[0] = 0;
  h->worker = std::thread([h, tune, n_steps]() {
    int rc = PGB_OK;
    if (hipSetDevice(h->device) != hipSuccess) rc = fail(PGB_E_DEVICE, "hipSetDevice in the step worker");
    if (rc == PGB_OK) rc = begin_steps(h, tune, n_steps);
    if (rc == PGB_OK) rc = run_until_idle(h, n_steps);
    h->job_rc = rc;
    if (rc != PGB_OK) snprintf(h->job_err, sizeof h->job_err, "%s", g_err);  // this thread's message
  });
  return PGB_OK;
}

extern "C" int pgb_sync(pgb_handle* h, pgb_counters* counters_out) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  JOIN_ASYNC(h);
  return fetch_counters(h, counters_out);
}

// the trees describe themselves (include/pgbart.h, pgb_tree_arrays::rule): how each split node sends a row left
static void fill_node_rules(const pgb_handle* h, pgb_tree_arrays* out, int total) {
  if (!out->rule) return;
  const int p = (int)h->rules_host.size();
  for (int g = 0; g < total; ++g) {
    const int j = out->var[g];
    out->rule[g] = j >= 0 && j < p ? h->rules_host[j] : PGB_RULE_CONTINUOUS;
  }
}

// the trees of the last pgb_step_host, already in host memory (k_export_step)
static int export_from_block(pgb_handle* h, pgb_tree_arrays* out) {
  const StepOutHdr* H = (const StepOutHdr*)h->out_host;
  const StepOutLayout& L = h->out_layout;
  const int nt = H->n_trees, total = H->total_nodes, K = H->K;
  if (!out->var) {
    out->n_trees = nt;
    out->n_outputs = K;
    out->total_nodes = total;
    return PGB_OK;
  }
  if (out->n_trees != nt || out->total_nodes != total) return fail(PGB_E_INVALID, "size mismatch");
  const unsigned char* B = h->out_host;
  for (int t = 0; t < nt; ++t) out->tree_id[t] = H->first + t;
  memcpy(out->node_off, B + L.node_off, (size_t)(nt + 1) * 4);
  memcpy(out->var, B + L.var, (size_t)total * 4);
  memcpy(out->left, B + L.left, (size_t)total * 4);
  memcpy(out->right, B + L.right, (size_t)total * 4);
  memcpy(out->split, B + L.split, (size_t)total * 8);
  memcpy(out->count, B + L.count, (size_t)total * 8);
  memcpy(out->value, B + L.value, (size_t)total * 8 * K);
  fill_node_rules(h, out, total);
  if (out->slope && out->xbar && out->svar) {
    if (H->lin) {
      memcpy(out->slope, B + L.slope, (size_t)total * 8 * K);
      memcpy(out->xbar, B + L.xbar, (size_t)total * 8);
      memcpy(out->svar, B + L.svar, (size_t)total * 4);
    } else {
      memset(out->slope, 0, (size_t)total * 8 * K);
      memset(out->xbar, 0, (size_t)total * 8);
      for (int g = 0; g < total; ++g) out->svar[g] = -1;
    }
  }
  return PGB_OK;
}

extern "C" int pgb_export_trees(pgb_handle* h, int32_t which, pgb_tree_arrays* out) {
  if (!h || !out) return fail(PGB_E_INVALID, "null argument");
  JOIN_ASYNC(h);
  REFUSE_POISONED(h);
  Dev& d = h->d;
  if (which == 0 && h->out_valid) return export_from_block(h, out);
  int first = which == 0 ? h->last_lower : 0;
  int nt = which == 0 ? h->last_n : d.m;
  std::vector<DTree> host(nt);
  if (nt > 0) {
    HIPCHK(hipMemcpyAsync(host.data(), d.trees + first, sizeof(DTree) * nt, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  int total = 0;
  for (int t = 0; t < nt; ++t) total += host[t].n_nodes;
  const int K = d.K, KX = d.K - 1;
  if (!out->var) {
    out->n_trees = nt;
    out->n_outputs = K;
    out->total_nodes = total;
    return PGB_OK;
  }
  std::vector<double> hx;
  if (KX > 0 && nt > 0) {
    hx.resize((size_t)nt * MAXN * KX);
    HIPCHK(hipMemcpy(hx.data(), d.tvx + (size_t)first * MAXN * KX, hx.size() * sizeof(double),
                     hipMemcpyDeviceToHost));
  }
  if (out->n_trees != nt || out->total_nodes != total) return fail(PGB_E_INVALID, "size mismatch");
  std::vector<LinP> hl;
  std::vector<double> hsx;  // slopes of outputs 1..K-1
  const bool want_lin = out->slope && out->xbar && out->svar;
  if (want_lin && d.response != PGB_RESPONSE_CONSTANT && nt > 0) {
    hl.resize((size_t)nt * MAXN);
    HIPCHK(hipMemcpy(hl.data(), d.tlin + (size_t)first * MAXN, hl.size() * sizeof(LinP), hipMemcpyDeviceToHost));
    if (KX > 0) {
      hsx.resize((size_t)nt * MAXN * KX);
      HIPCHK(hipMemcpy(hsx.data(), d.tsx + (size_t)first * MAXN * KX, hsx.size() * sizeof(double),
                       hipMemcpyDeviceToHost));
    }
  }
  int off = 0;
  for (int t = 0; t < nt; ++t) {
    const DTree& T = host[t];
    out->tree_id[t] = first + t;
    out->node_off[t] = off;
    for (int k = 0; k < T.n_nodes; ++k) {
      const DNode& z = T.nd[k];
      out->var[off + k] = z.var;
      out->split[off + k] = z.var >= 0 ? z.split : 0.0;
      out->left[off + k] = z.var >= 0 ? (int32_t)z.left : -1;
      out->right[off + k] = z.var >= 0 ? (int32_t)z.right : -1;
      out->count[off + k] = z.cnt;
      out->value[(size_t)(off + k) * K] = z.var < 0 ? z.value : 0.0;
      if (want_lin) {
        const bool islin = z.var < 0 && !hl.empty() && hl[(size_t)t * MAXN + k].svar >= 0;
        out->slope[(size_t)(off + k) * K] = islin ? hl[(size_t)t * MAXN + k].slope : 0.0;
        for (int o = 1; o < K; ++o)
          out->slope[(size_t)(off + k) * K + o] = islin ? hsx[((size_t)t * MAXN + k) * KX + o - 1] : 0.0;
        out->xbar[off + k] = islin ? hl[(size_t)t * MAXN + k].xbar : 0.0;
        out->svar[off + k] = islin ? (int32_t)hl[(size_t)t * MAXN + k].svar : -1;
      }
      for (int o = 1; o < K; ++o)
        out->value[(size_t)(off + k) * K + o] = z.var < 0 ? hx[((size_t)t * MAXN + k) * KX + o - 1] : 0.0;
    }
    off += T.n_nodes;
  }
  out->node_off[nt] = off;
  fill_node_rules(h, out, total);
  return PGB_OK;
}

extern "C" int pgb_export_trees_packed(pgb_handle* h, int32_t which, void* host_buf, int64_t cap_bytes,
                                       int64_t* bytes_out) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  const int rc = pgb_export_trees_packed_via(h, which, host_buf, cap_bytes, bytes_out,
                                             h->s.response != PGB_RESPONSE_CONSTANT);
  if (rc == PGB_E_NOMEM) return fail(rc, "packed tree record does not fit the buffer (*bytes_out has the size)");
  return rc;
}

// (pgb_get_state: pgb_checkpoint.h, next to the image that needs the same idle-state read)

extern "C" int pgb_get_split_weights(pgb_handle* h, double* out) {
  if (!h || !out) return fail(PGB_E_INVALID, "null argument");
  JOIN_ASYNC(h);
  std::vector<long long> a((size_t)h->d.p);
  HIPCHK(hipMemcpyAsync(a.data(), h->d.alpha + (size_t)h->alpha_cur * h->d.p, h->d.p * sizeof(long long),
                        hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  // in units of the caller's prior: prior_j + number of tuning counts (up to 2^-24 rounding)
  for (int j = 0; j < h->d.p; ++j) out[j] = (double)a[j] * (h->d.max_prior * pgb_pow2(-PGB_ALPHA_BITS));
  return PGB_OK;
}

extern "C" int pgb_predict(const pgb_tree_arrays* trees, const int32_t* forest_tree_idx, int32_t n_forests,
                           int32_t m, const double* X_dev, int64_t n_rows, int32_t p, int64_t ldx,
                           const int32_t* excluded_host, int32_t n_excluded,
                           double* out_dev, void* stream) {
  if (!trees || !forest_tree_idx || !X_dev || !out_dev) return fail(PGB_E_INVALID, "null argument");
  if (trees->n_outputs < 1 || trees->n_outputs > PGB_MAX_OUTPUTS) return fail(PGB_E_INVALID, "n_outputs");
  if (n_forests < 1 || n_rows < 1) return PGB_OK;
  // a malformed history (truncated file, mismatched m) is an error, not an out-of-bounds walk
  if (trees->n_trees < 0 || trees->total_nodes < 0) return fail(PGB_E_INVALID, "tree arrays are inconsistent (node_off / left / right)");
  for (long long i = 0; i < (long long)n_forests * m; ++i)
    if (forest_tree_idx[i] < 0 || forest_tree_idx[i] >= trees->n_trees)
      return fail(PGB_E_INVALID, "forest_tree_idx entry outside [0, n_trees)");
  for (int t = 0; t < trees->n_trees; ++t) {
    const int base = trees->node_off[t], end = trees->node_off[t + 1];
    if (base < 0 || end <= base || end > trees->total_nodes)
      return fail(PGB_E_INVALID, "tree arrays are inconsistent (node_off / left / right)");
    for (int g = base; g < end; ++g) {
      if (trees->var[g] < 0) continue;
      if (trees->var[g] >= p) return fail(PGB_E_INVALID, "a tree splits on a column X does not have");
      if (trees->rule && trees->rule[g] != PGB_RULE_CONTINUOUS && trees->rule[g] != PGB_RULE_ONEHOT &&
          trees->rule[g] != PGB_RULE_SUBSET)
        return fail(PGB_E_INVALID, "a split node carries an unknown split rule");
      if (trees->left[g] < 0 || trees->right[g] < 0 || trees->left[g] >= end - base || trees->right[g] >= end - base)
        return fail(PGB_E_INVALID, "tree arrays are inconsistent (node_off / left / right)");
    }
  }
  hipStream_t sm = (hipStream_t)stream;
  const int K = trees->n_outputs, N = trees->total_nodes, NT = trees->n_trees;
  std::vector<uint8_t> excl((size_t)p, 0);
  for (int e = 0; e < n_excluded; ++e)
    if (excluded_host[e] >= 0 && excluded_host[e] < p) excl[excluded_host[e]] = 1;
  const bool lin = trees->slope && trees->xbar && trees->svar;
  // one upload buffer: [PNode N | FNode N | value N K | (slope N K | xbar N) | root NT | fidx | (svar N)]
  const size_t o_fn = (size_t)N * sizeof(PNode);
  const size_t o_val = o_fn + (size_t)N * sizeof(FNode);
  const size_t o_slope = o_val + (size_t)N * K * 8;
  const size_t o_xbar = o_slope + (lin ? (size_t)N * K * 8 : 0);
  const size_t o_root = o_xbar + (lin ? (size_t)N * 8 : 0);
  const size_t o_f = o_root + (size_t)NT * sizeof(int2);
  const size_t o_svar = o_f + (size_t)n_forests * m * 4;
  const size_t bytes = o_svar + (lin ? (size_t)N * 4 : 0);
  std::vector<uint8_t> hb(bytes);
  PNode* hn = (PNode*)hb.data();
  FNode* hf = (FNode*)(hb.data() + o_fn);
  int2* hroot = (int2*)(hb.data() + o_root);
  std::vector<int> depth_of;
  bool cont = true;  // every split of every tree is `x <= v`: the instance without the rule dispatch
  int32_t* hsvar = (int32_t*)(hb.data() + o_svar);
  for (int t = 0; t < NT; ++t) {
    const int base = trees->node_off[t], end = trees->node_off[t + 1];
    // depth of the tree and whether it can take the fixed-length walk (<= 255 nodes, children after
    // their parent as both backends build them, no split on an excluded variable)
    const int nn = end - base;
    bool general = nn > 255 || nn < 1;
    int depth = 0;
    depth_of.assign((size_t)(nn > 0 ? nn : 1), 0);
    for (int k = 0; k < nn && !general; ++k) {
      const int g = base + k;
      if (trees->var[g] < 0) continue;
      const int l = trees->left[g], r = trees->right[g];
      if (l <= k || r <= k || l >= nn || r >= nn) { general = true; break; }
      if (trees->var[g] < p && excl[trees->var[g]]) general = true;
      depth_of[l] = depth_of[r] = depth_of[k] + 1;
      if (depth_of[k] + 1 > depth) depth = depth_of[k] + 1;
    }
    if (depth > 255) general = true;
    hroot[t] = make_int2(base, general ? 0x100 : depth);
    for (int g = base; g < end; ++g) {
      PNode z;
      z.var = trees->var[g];
      if (z.var >= p) return fail(PGB_E_INVALID, "a tree splits on a column X does not have");
      z.left = z.var >= 0 ? base + trees->left[g] : -1;
      z.right = z.var >= 0 ? base + trees->right[g] : -1;
      const int rule = z.var >= 0 && trees->rule ? trees->rule[g] : PGB_RULE_CONTINUOUS;  // the node's own
      if (rule != PGB_RULE_CONTINUOUS) cont = false;
      z.flags = z.var >= 0 ? ((rule << 1) | (excl[z.var] ? 1 : 0)) : 0;
      z.split = trees->split[g];
      z.cnt = (double)trees->count[g];
      hn[g] = z;
      FNode f;
      f.pad = 0;
      if (z.var >= 0 && !general) {
        f.split = z.split; f.var = z.var; f.left = (uint8_t)trees->left[g]; f.right = (uint8_t)trees->right[g];
      } else {  // a leaf points at itself: x <= +inf for every value that is not NaN
        f.split = __builtin_inf(); f.var = 0; f.left = f.right = (uint8_t)((g - base) & 255);
      }
      hf[g] = f;
      if (lin) {
        const int js = trees->svar[g];
        hsvar[g] = (js >= 0 && js < p && !excl[js]) ? js : -1;
      }
    }
  }
  memcpy(hb.data() + o_val, trees->value, (size_t)N * K * 8);
  if (lin) {
    memcpy(hb.data() + o_slope, trees->slope, (size_t)N * K * 8);
    memcpy(hb.data() + o_xbar, trees->xbar, (size_t)N * 8);
  }
  memcpy(hb.data() + o_f, forest_tree_idx, (size_t)n_forests * m * 4);
  uint8_t* db = nullptr;
  HIPCHK(hipMalloc((void**)&db, bytes));
  hipError_t e = hipMemcpyAsync(db, hb.data(), bytes, hipMemcpyHostToDevice, sm);
  if (e != hipSuccess) { (void)hipFree(db); return fail_hip(e, "hipMemcpyAsync"); }
  PredTrees T;
  T.node = (const PNode*)db;
  T.value = (const double*)(db + o_val);
  T.slope = lin ? (const double*)(db + o_slope) : nullptr;
  T.xbar = lin ? (const double*)(db + o_xbar) : nullptr;
  T.fnode = (const FNode*)(db + o_fn);
  T.root = (const int2*)(db + o_root);
  T.svar = lin ? (const int32_t*)(db + o_svar) : nullptr;
  // one wave per workgroup; enough workgroups to fill the chip, each looping over its share of forests
  const long long gx = (n_rows + PRED_BT - 1) / PRED_BT;
  long long want_wgs = p <= PRED_LDS_MAXP ? 16384 : 4096;  // measured at cfg2 (32 forests x 100k rows): 10.0 vs 11.7 ms
  if (const char* ev = getenv("PGB_PRED_WGS")) want_wgs = atoll(ev) > 0 ? atoll(ev) : want_wgs;
  long long gy = (want_wgs + gx - 1) / gx;
  if (gy > n_forests) gy = n_forests;
  if (gy < 1) gy = 1;
  dim3 grid((unsigned)gx, (unsigned)gy);
#define LAUNCH_PRED(L_, C_, LDS_)                                                                               \
  hipLaunchKernelGGL((k_predict<L_, C_>), grid, dim3(PRED_BT), (LDS_), sm, T, (const int32_t*)(db + o_f), n_forests, \
                     m, K, (int)p, X_dev, (long long)n_rows, (long long)ldx, out_dev)
  const size_t lds = (size_t)p * 65 * sizeof(double);
  if (p <= PRED_LDS_MAXP) {
    if (cont) LAUNCH_PRED(true, true, lds);
    else LAUNCH_PRED(true, false, lds);
  } else {
    if (cont) LAUNCH_PRED(false, true, 0);
    else LAUNCH_PRED(false, false, 0);
  }
#undef LAUNCH_PRED
  e = hipGetLastError();
  hipError_t e2 = hipStreamSynchronize(sm);
  (void)hipFree(db);
  if (e != hipSuccess) return fail_hip(e, "k_predict launch");
  if (e2 != hipSuccess) return fail_hip(e2, "k_predict");
  return PGB_OK;
}

