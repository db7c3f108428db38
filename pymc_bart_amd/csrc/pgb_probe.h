// pgb_probe.h -- part of pgbart_hip.hip (not a standalone header): test hooks pgbh_*.
// ------------------------------------------------------------------ device probes of the numeric contract
// include/pgbart_spec.h is compiled twice -- by gcc into the oracle and by hipcc into the kernels -- so
// "HIP == oracle" says nothing about it unless the DEVICE compile of each function is looked at by
// itself.  These hooks run the device compile on arrays the caller supplies (host pointers; the hook
// stages them): tests/test_spec_device_gpu.py compares the results bit for bit with the host compile
// (pgbo_* of the oracle library: another compiler, another FMA contraction policy, another libm if any
// leaked in) and, independently of both, with SciPy.  Not part of the sampler ABI (include/pgbart.h);
// nothing in the product calls them.

struct ProbeArgs {
  int what, family, K;
  long long n;
  const double *a, *b, *c, *d;          // inputs
  const long long *ia, *ib, *ic, *id_, *ie, *if_, *ig;
  double *o0, *o1, *o2, *o3;            // outputs
  long long* oi;
  unsigned* osat;
  double p0, p1, p2, p3, p4;
  unsigned long long seed;
  unsigned u0, u1, u2, u3;
};

enum {
  PROBE_LOGLIKQ = 1,   // pgb_loglik1q, tables read from global memory
  PROBE_BERN_LDS = 2,  // pgb_loglik_bern_s on the signed predictor, tables staged in LDS (the form k_loglik runs)
  PROBE_MULTI = 3,     // pgb_loglik (K linear predictors)
  PROBE_LOG_NDTR = 4,
  PROBE_MATH = 5,      // pgb_exp, pgb_log, pgb_sincos2pi
  PROBE_NORMAL2 = 6,
  PROBE_DRAW2 = 7,
  PROBE_QUANT = 8,
  PROBE_LEAF = 9,  // pgb_leaf_value, pgb_leaf_sse
  PROBE_LIN = 10,  // pgb_lin_fit, pgb_lin_sse
  PROBE_GO_LEFT = 11,
  PROBE_MATH_T = 12,   // pgb_exp_t, pgb_log_t (the table-driven forms of the per-row likelihoods), tables in LDS
  PROBE_MULTI_LDS = 13,  // pgb_loglikq_t with every table in LDS (the form k_loglik<K> runs)
  PROBE_CAT_F = 14,      // pgb_loglik_cat_f: the factorised softmax of constant leaves, tables in LDS
};

__global__ __launch_bounds__(256) void k_probe(ProbeArgs A) {
  // every table of the per-row likelihood math staged in LDS, as the likelihood pass does
  __shared__ double s_lphi[PGB_LPHI_SIZE];
  __shared__ double s_expt[PGB_EXPT_SIZE];
  __shared__ __attribute__((aligned(16))) double s_logt[PGB_LOGT_SIZE];
  pgb_lltabs lds;
  lds.lphi = s_lphi;
  lds.expt = s_expt;
  lds.logt = s_logt;
  const pgb_lltabs glob = pgb_lltabs_default();
  if (A.what == PROBE_BERN_LDS || A.what == PROBE_MATH_T || A.what == PROBE_MULTI_LDS || A.what == PROBE_CAT_F) {
    for (int i = threadIdx.x; i < PGB_LPHI_SIZE; i += blockDim.x) s_lphi[i] = glob.lphi[i];
    for (int i = threadIdx.x; i < PGB_EXPT_SIZE; i += blockDim.x) s_expt[i] = glob.expt[i];
    for (int i = threadIdx.x; i < PGB_LOGT_SIZE; i += blockDim.x) s_logt[i] = glob.logt[i];
    __syncthreads();
  }
  __shared__ double s_cat[3][PGB_MAX_OUTPUTS];  // PROBE_CAT_F: the leaf values v, and d, w of pgb_cat_side
  __shared__ int s_cat_fast;
  if (A.what == PROBE_CAT_F) {
    if (threadIdx.x == 0) {
      for (int k = 0; k < A.K; ++k) s_cat[0][k] = A.c[k];
      s_cat_fast = pgb_cat_side(A.K, s_cat[0], s_expt, s_cat[1], s_cat[2]);
    }
    __syncthreads();
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += (long long)gridDim.x * blockDim.x) {
    switch (A.what) {
      case PROBE_LOGLIKQ:
        A.o0[i] = pgb_loglik1q(A.family, A.a[i], A.b[i], A.p0, A.p1, &glob);
        break;
      case PROBE_BERN_LDS: {
        // the response as a sign mask, exactly as the likelihood pass flips the predictor
        const unsigned long long ysgn = A.a[i] > 0.5 ? 0ull : 0x8000000000000000ull;
        const double smu = pgb_u2d(pgb_d2u(A.b[i]) ^ ysgn);
        A.o0[i] = pgb_loglik_bern_s(A.family, smu, &lds);
        break;
      }
      case PROBE_MULTI: {
        double mu[PGB_MAX_OUTPUTS];
        for (int k = 0; k < A.K; ++k) mu[k] = A.b[i * A.K + k];
        A.o0[i] = pgb_loglik(A.family, A.K, A.a[i], mu);
        break;
      }
      case PROBE_MULTI_LDS: {
        double mu[PGB_MAX_OUTPUTS];
        for (int k = 0; k < A.K; ++k) mu[k] = A.b[i * A.K + k];
        A.o0[i] = pgb_loglikq_t(A.family, A.K, A.a[i], mu, A.p0, A.p1, &lds);
        break;
      }
      case PROBE_CAT_F: {  // a = y, b = eta [n][K], c = v [K]; the child's part: made once, in LDS (see below)
        double eta[PGB_MAX_OUTPUTS];
        for (int k = 0; k < A.K; ++k) eta[k] = A.b[i * A.K + k];
        A.o0[i] = pgb_loglik_cat_f(A.K, A.a[i], eta, s_cat[0], s_cat[1], s_cat[2], s_cat_fast, &lds);
        break;
      }
      case PROBE_MATH_T:
        A.o0[i] = pgb_exp_t(A.a[i], s_expt);
        A.o1[i] = pgb_log_t(A.a[i], s_logt);
        break;
      case PROBE_LOG_NDTR:
        A.o0[i] = pgb_log_ndtr(A.a[i]);
        break;
      case PROBE_MATH: {
        A.o0[i] = pgb_exp(A.a[i]);
        A.o1[i] = pgb_log(A.a[i]);
        double s, c;
        pgb_sincos2pi(A.a[i], &s, &c);
        A.o2[i] = s;
        A.o3[i] = c;
        break;
      }
      case PROBE_NORMAL2: {
        double z0, z1;
        pgb_normal2(A.a[i], A.b[i], &z0, &z1);
        A.o0[i] = z0;
        A.o1[i] = z1;
        break;
      }
      case PROBE_DRAW2: {
        const pgb_u2 u = pgb_draw2(A.seed, A.u0, A.u1, A.u2, A.u3, (uint32_t)i);
        A.o0[i] = u.u0;
        A.o1[i] = u.u1;
        break;
      }
      case PROBE_QUANT: {
        unsigned sat = 0;
        A.oi[i] = pgb_quant(A.a[i], A.p0, &sat);
        A.osat[i] = sat;
        break;
      }
      case PROBE_LEAF:
        A.o0[i] = pgb_leaf_value(A.ia[i], A.ib[i], A.p0, A.p2, A.a[i], A.p3);
        A.o1[i] = pgb_leaf_sse(A.ia[i], A.ic[i], A.id_[i], A.o0[i], A.p0, A.p1);
        break;
      case PROBE_LIN: {
        const pgb_linfit f = pgb_lin_fit(A.ia[i], A.ib[i], A.ic[i], A.id_[i], A.ie[i], A.p0, A.p1, A.p2);
        A.o0[i] = f.slope_u;
        A.o1[i] = f.ubar;
        A.o2[i] = f.var_u;
        A.o3[i] = pgb_lin_sse(A.a[i], f, A.if_[i], A.ig[i], A.p0);
        break;
      }
      case PROBE_GO_LEFT:
        A.oi[i] = pgb_go_left(A.family, A.a[i], A.b[i]);
        break;
    }
  }
}

namespace probe {
struct Stage {  // device copies of the caller's host arrays, freed on scope exit
  std::vector<void*> bufs;
  bool ok = true;
  template <typename T>
  T* in(const T* host, long long cnt) {
    T* d = out<T>(cnt);
    if (d && host && hipMemcpy(d, host, sizeof(T) * (size_t)cnt, hipMemcpyHostToDevice) != hipSuccess) ok = false;
    return d;
  }
  template <typename T>
  T* out(long long cnt) {
    void* d = nullptr;
    if (hipMalloc(&d, sizeof(T) * (size_t)(cnt > 0 ? cnt : 1)) != hipSuccess) {
      ok = false;
      return nullptr;
    }
    bufs.push_back(d);
    return (T*)d;
  }
  template <typename T>
  void back(T* host, const T* dev, long long cnt) {
    if (host && dev && hipMemcpy(host, dev, sizeof(T) * (size_t)cnt, hipMemcpyDeviceToHost) != hipSuccess) ok = false;
  }
  ~Stage() {
    for (void* b : bufs) (void)hipFree(b);
  }
};
static int run(const ProbeArgs& A, Stage& st) {
  if (!st.ok) return fail(PGB_E_DEVICE, "probe: staging failed");
  if (A.n <= 0) return PGB_OK;
  long long g = (A.n + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_probe, dim3((unsigned)g), dim3(256), 0, 0, A);
  if (hipDeviceSynchronize() != hipSuccess) return fail(PGB_E_DEVICE, "probe: kernel failed");
  return PGB_OK;
}
}  // namespace probe

extern "C" {
int pgbh_loglikq(int family, const double* y, const double* mu, int64_t n, double param, double param2, double* out) {
  probe::Stage st;
  ProbeArgs A{};
  A.what = PROBE_LOGLIKQ; A.family = family; A.n = n; A.p0 = param; A.p1 = param2;
  A.a = st.in(y, n); A.b = st.in(mu, n); A.o0 = st.out<double>(n);
  int rc = probe::run(A, st);
  if (rc == PGB_OK) st.back(out, A.o0, n);
  return st.ok ? rc : fail(PGB_E_DEVICE, "probe: copy failed");
}
int pgbh_loglik_bern_lds(int family, const double* y, const double* mu, int64_t n, double* out) {
  probe::Stage st;
  ProbeArgs A{};
  A.what = PROBE_BERN_LDS; A.family = family; A.n = n;
  A.a = st.in(y, n); A.b = st.in(mu, n); A.o0 = st.out<double>(n);
  int rc = probe::run(A, st);
  if (rc == PGB_OK) st.back(out, A.o0, n);
  return st.ok ? rc : fail(PGB_E_DEVICE, "probe: copy failed");
}
int pgbh_loglik_multi(int family, int K, const double* y, const double* mu /* [n][K] */, int64_t n, double* out) {
  if (K < 1 || K > PGB_MAX_OUTPUTS) return fail(PGB_E_INVALID, "probe: K");
  probe::Stage st;
  ProbeArgs A{};
  A.what = PROBE_MULTI; A.family = family; A.K = K; A.n = n;
  A.a = st.in(y, n); A.b = st.in(mu, n * K); A.o0 = st.out<double>(n);
  int rc = probe::run(A, st);
  if (rc == PGB_OK) st.back(out, A.o0, n);
  return st.ok ? rc : fail(PGB_E_DEVICE, "probe: copy failed");
}
int pgbh_loglik_multi_lds(int family, int K, const double* y, const double* mu /* [n][K] */, int64_t n, double param,
                          double param2, double* out) {
  if (K < 1 || K > PGB_MAX_OUTPUTS) return fail(PGB_E_INVALID, "probe: K");
  probe::Stage st;
  ProbeArgs A{};
  A.what = PROBE_MULTI_LDS; A.family = family; A.K = K; A.n = n; A.p0 = param; A.p1 = param2;
  A.a = st.in(y, n); A.b = st.in(mu, n * K); A.o0 = st.out<double>(n);
  int rc = probe::run(A, st);
  if (rc == PGB_OK) st.back(out, A.o0, n);
  return st.ok ? rc : fail(PGB_E_DEVICE, "probe: copy failed");
}
int pgbh_loglik_cat_f(int K, const double* y, const double* eta /* [n][K] */, const double* v /* [K] */, int64_t n,
                      double* out) {
  if (K < 1 || K > PGB_MAX_OUTPUTS) return fail(PGB_E_INVALID, "probe: K");
  probe::Stage st;
  ProbeArgs A{};
  A.what = PROBE_CAT_F; A.K = K; A.n = n;
  A.a = st.in(y, n); A.b = st.in(eta, n * K); A.c = st.in(v, (int64_t)K); A.o0 = st.out<double>(n);
  int rc = probe::run(A, st);
  if (rc == PGB_OK) st.back(out, A.o0, n);
  return st.ok ? rc : fail(PGB_E_DEVICE, "probe: copy failed");
}
// the order keys pgb_set_data builds for a column (k_key_* + the radix sort), on a column the caller supplies
int pgbh_order_keys(const double* x, int64_t n, uint16_t* keys) {  // (tests/test_spec_device_gpu.py)
  probe::Stage st;
  double* dx = st.in(x, n);
  uint16_t* dk = st.out<uint16_t>(n);
  if (!st.ok) return fail(PGB_E_DEVICE, "probe: allocation / copy failed");
  const hipError_t e = order_keys_build(dx, (long long)n, (long long)n, 1, dk, nullptr);
  if (e != hipSuccess) return fail_hip(e, "probe: order keys");
  st.back(keys, dk, n);
  return st.ok ? PGB_OK : fail(PGB_E_DEVICE, "probe: copy failed");
}
int pgbh_math_t(const double* x, int64_t n, double* e, double* l) {
  probe::Stage st;
  ProbeArgs A{};
  A.what = PROBE_MATH_T; A.n = n;
  A.a = st.in(x, n);
  A.o0 = st.out<double>(n); A.o1 = st.out<double>(n);
  int rc = probe::run(A, st);
  if (rc == PGB_OK) { st.back(e, A.o0, n); st.back(l, A.o1, n); }
  return st.ok ? rc : fail(PGB_E_DEVICE, "probe: copy failed");
}
int pgbh_log_ndtr(const double* x, int64_t n, double* out) {
  probe::Stage st;
  ProbeArgs A{};
  A.what = PROBE_LOG_NDTR; A.n = n;
  A.a = st.in(x, n); A.o0 = st.out<double>(n);
  int rc = probe::run(A, st);
  if (rc == PGB_OK) st.back(out, A.o0, n);
  return st.ok ? rc : fail(PGB_E_DEVICE, "probe: copy failed");
}
int pgbh_math(const double* x, int64_t n, double* e, double* l, double* s, double* c) {
  probe::Stage st;
  ProbeArgs A{};
  A.what = PROBE_MATH; A.n = n;
  A.a = st.in(x, n);
  A.o0 = st.out<double>(n); A.o1 = st.out<double>(n); A.o2 = st.out<double>(n); A.o3 = st.out<double>(n);
  int rc = probe::run(A, st);
  if (rc == PGB_OK) { st.back(e, A.o0, n); st.back(l, A.o1, n); st.back(s, A.o2, n); st.back(c, A.o3, n); }
  return st.ok ? rc : fail(PGB_E_DEVICE, "probe: copy failed");
}
int pgbh_normal2(const double* u0, const double* u1, int64_t n, double* z0, double* z1) {
  probe::Stage st;
  ProbeArgs A{};
  A.what = PROBE_NORMAL2; A.n = n;
  A.a = st.in(u0, n); A.b = st.in(u1, n);
  A.o0 = st.out<double>(n); A.o1 = st.out<double>(n);
  int rc = probe::run(A, st);
  if (rc == PGB_OK) { st.back(z0, A.o0, n); st.back(z1, A.o1, n); }
  return st.ok ? rc : fail(PGB_E_DEVICE, "probe: copy failed");
}
/* draws (seed, iter, round, particle, purpose, sub = 0 .. n-1) */
int pgbh_draw2(uint64_t seed, uint32_t iter, uint32_t round, uint32_t particle, uint32_t purpose, int64_t n,
               double* u0, double* u1) {
  probe::Stage st;
  ProbeArgs A{};
  A.what = PROBE_DRAW2; A.n = n; A.seed = seed; A.u0 = iter; A.u1 = round; A.u2 = particle; A.u3 = purpose;
  A.o0 = st.out<double>(n); A.o1 = st.out<double>(n);
  int rc = probe::run(A, st);
  if (rc == PGB_OK) { st.back(u0, A.o0, n); st.back(u1, A.o1, n); }
  return st.ok ? rc : fail(PGB_E_DEVICE, "probe: copy failed");
}
int pgbh_quant(const double* x, int64_t n, double scale, int64_t* q, uint32_t* sat) {
  probe::Stage st;
  ProbeArgs A{};
  A.what = PROBE_QUANT; A.n = n; A.p0 = scale;
  A.a = st.in(x, n); A.oi = st.out<long long>(n); A.osat = st.out<unsigned>(n);
  int rc = probe::run(A, st);
  if (rc == PGB_OK) { st.back((long long*)q, A.oi, n); st.back((unsigned*)sat, A.osat, n); }
  return st.ok ? rc : fail(PGB_E_DEVICE, "probe: copy failed");
}
int pgbh_leaf(const int64_t* cnt, const int64_t* q_st, const int64_t* q_r, const int64_t* q_r2, const double* z,
              int64_t n, double inv_c1, double inv_c2, double m, double leaf_sd, double* value, double* sse) {
  probe::Stage st;
  ProbeArgs A{};
  A.what = PROBE_LEAF; A.n = n; A.p0 = inv_c1; A.p1 = inv_c2; A.p2 = m; A.p3 = leaf_sd;
  A.ia = st.in((const long long*)cnt, n); A.ib = st.in((const long long*)q_st, n);
  A.ic = st.in((const long long*)q_r, n); A.id_ = st.in((const long long*)q_r2, n);
  A.a = st.in(z, n);
  A.o0 = st.out<double>(n); A.o1 = st.out<double>(n);
  int rc = probe::run(A, st);
  if (rc == PGB_OK) { st.back(value, A.o0, n); st.back(sse, A.o1, n); }
  return st.ok ? rc : fail(PGB_E_DEVICE, "probe: copy failed");
}
int pgbh_lin(const int64_t* cnt, const int64_t* q_u, const int64_t* q_uu, const int64_t* q_us, const int64_t* q_st,
             const int64_t* q_ur, const int64_t* q_r, const double* sse_const, int64_t n, double inv_c1, double inv_R,
             double m, double* slope_u, double* ubar, double* var_u, double* sse) {
  probe::Stage st;
  ProbeArgs A{};
  A.what = PROBE_LIN; A.n = n; A.p0 = inv_c1; A.p1 = inv_R; A.p2 = m;
  A.ia = st.in((const long long*)cnt, n); A.ib = st.in((const long long*)q_u, n);
  A.ic = st.in((const long long*)q_uu, n); A.id_ = st.in((const long long*)q_us, n);
  A.ie = st.in((const long long*)q_st, n); A.if_ = st.in((const long long*)q_ur, n);
  A.ig = st.in((const long long*)q_r, n);
  A.a = st.in(sse_const, n);
  A.o0 = st.out<double>(n); A.o1 = st.out<double>(n); A.o2 = st.out<double>(n); A.o3 = st.out<double>(n);
  int rc = probe::run(A, st);
  if (rc == PGB_OK) { st.back(slope_u, A.o0, n); st.back(ubar, A.o1, n); st.back(var_u, A.o2, n); st.back(sse, A.o3, n); }
  return st.ok ? rc : fail(PGB_E_DEVICE, "probe: copy failed");
}
int pgbh_go_left(int rule, const double* x, const double* v, int64_t n, int64_t* out) {
  probe::Stage st;
  ProbeArgs A{};
  A.what = PROBE_GO_LEFT; A.family = rule; A.n = n;
  A.a = st.in(x, n); A.b = st.in(v, n); A.oi = st.out<long long>(n);
  int rc = probe::run(A, st);
  if (rc == PGB_OK) st.back((long long*)out, A.oi, n);
  return st.ok ? rc : fail(PGB_E_DEVICE, "probe: copy failed");
}
}  // extern "C"
