/*
 * pgbart_oracle.c -- CPU restatement of particle-Gibbs BART (PGBART.astep).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (pymc_bart_amd/) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / CPU baseline.
 *
 * PARITY UNPINNED against the reference binary: pymc-bart 0.13.1 delegates the
 * sampler to the external Rust wheel `bartrs>=0.4.0` (requirements.txt:6,
 * pymc_bart/pymc_bart.py:2, tests/test_bart.py:4); its source is not vendored and
 * cannot be built or imported here (no Rust toolchain, no network), and the
 * reference's tests pin no numeric outputs on this path.  This file
 * therefore restates the published algorithm -- pymc-bart's pgbart.py / tree.py
 * as they existed up to 0.12.x and as summarised in SURVEY.md Appendix A -- under
 * the numeric contract of include/pgbart_spec.h, and is pinned by (i) the
 * behavioural assertions of the reference's own tests (tests/test_oracle_*.py cite
 * them) and (ii) the golden vectors of the one exactly-runnable piece, the
 * variable-inclusion codec (utils.py:1368-1398, tests/golden/vi_codec.json).
 *
 * Implements the same C ABI as the HIP library (include/pgbart.h) with "device"
 * pointers being ordinary host pointers.  It deliberately uses a DIFFERENT data
 * structure from the HIP backend -- per-node sorted row-index segments in a shared
 * arena (the upstream idx_data_points design) instead of per-particle leaf-label
 * arrays -- so that agreement between the two is evidence, not tautology.  Only the
 * numeric primitives of pgbart_spec.h (Philox, exp/log, fixed point) are shared.
 *
 * Where each step follows upstream ([U] = upstream recall):
 *   astep batching ............. o_step()          [U] PGBART.astep
 *   residual removal ........... o_tree_begin()    [U] sum_trees_noi = sum_trees - tree.predict
 *   init_particles ............. o_tree_begin()    [U] PGBART.init_particles
 *   sample_tree / grow_tree .... o_particle_step() [U] ParticleTree.sample_tree, grow_tree
 *   draw_leaf_value ............ pgb_leaf_value    [U] draw_leaf_value ("constant" response)
 *   update_weight .............. pgb_leaf_sse      [U] PGBART.update_weight (Normal family,
 *                                                   evaluated from exact sufficient statistics)
 *   normalize/resample ......... o_resample()      [U] normalize, systematic, inverse_cdf
 *   get_particle_tree .......... o_tree_end()      [U] (one categorical draw from the weights)
 *   tuning ..................... o_tree_end()      [U] alpha_vec counts, RunningSd (CHANGELOG.md:413)
 *   prediction ................. pgb_predict       utils.py:60-71, CHANGELOG.md:410-411
 * Deviations from upstream, all deliberate and documented in DESIGN.md:
 *   - counter-based RNG instead of the NumPy stream;
 *   - a fresh particle's weight is the likelihood of its stump (upstream leaves 0; pgb_settings.compat bit 0
 *     restores upstream's rule, bit 1 the empty right child of a one-hot split);
 *   - the final particle is one categorical draw (same distribution as upstream's
 *     systematic()[randint]);
 *   - trees are capped at PGB_MAX_NODES nodes and PGB_MAX_DEPTH depth.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pgbart.h"
#include "pgbart_image.h"
#include "pgbart_pack.h"
#include "pgbart_spec.h"

#define MAXN PGB_MAX_NODES
#define PGB_STR2(x) #x
#define PGB_STR(x) PGB_STR2(x)

static __thread char g_err[256];
static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}
const char* pgb_last_error(void) { return g_err; }
const char* pgb_backend_name(void) { return "oracle-cpu"; }
int32_t pgb_max_particles(void) { return PGB_MAX_PARTICLES; }
int32_t pgb_abi_version(void) { return PGB_ABI_VERSION; }

typedef struct {
  int32_t var; /* -1 leaf */
  int32_t left, right, depth;
  int32_t label; /* leaf label (left child inherits its parent's) */
  double split;
  int64_t cnt;
  int64_t q_st, q_r, q_r2; /* fixed-point sums over the node rows */
  double value, sse;
  double valx[PGB_MAX_OUTPUTS - 1];    /* leaf value of outputs 1..K-1 (K-vector leaves) */
  int64_t q_stx[PGB_MAX_OUTPUTS - 1];  /* sum of sum_trees over the node rows, outputs 1..K-1 */
  int64_t ll;  /* non-Normal families: fixed-point log-likelihood of the node's rows */
  double slope, xbar; /* linear response: the leaf predicts value + slope (x[svar] - xbar) */
  double slopex[PGB_MAX_OUTPUTS - 1]; /* ... slopes of outputs 1..K-1 (svar / xbar are shared) */
  int32_t svar;       /* ... svar = -1: constant leaf                                       */
  int64_t seg; /* arena offset of the sorted row list (oracle only) */
} onode;

typedef struct {
  int32_t n_nodes, n_leaves, next_pop;
  double sse_tot, sse_orph;
  int64_t ll_tot, ll_orph; /* Bernoulli families */
  onode nd[MAXN];
} otree;

struct pgb_handle {
  pgb_settings s;
  pgb_scales sc;
  double* X; /* column-major p x n */
  double* y;
  double* off; /* per-row offset of the linear predictor (single-output per-row families) */
  int32_t* rules;
  int64_t* alpha_vec; /* p  integer split weights (pgb_alpha_init + counts * unit) */
  int64_t* cdf;       /* p  prefix sums the sampler currently uses */
  int64_t alpha_unit;
  double max_prior;
  int* col_has_nan;
  int* col_ex;       /* linear response: exponent bound of every column (u = x 2^-ex) */
  double lin_R, inv_R;
  double* st; /* sum_trees n */
  double* r;  /* y - noi */
  double* oldv;
  double* rs_mean;
  double* rs_m2;
  int64_t rs_count;
  double leaf_sd;                       /* output 0 */
  double leaf_sdx[PGB_MAX_OUTPUTS - 1]; /* outputs 1..K-1 */
  double inv_sigma2;
  double lik_param2; /* second scalar parameter of the two-parameter families */
  int64_t iter;
  int32_t lower;
  otree* trees;     /* m accepted trees */
  uint8_t* lid;     /* m x n leaf labels of accepted trees */
  otree* part;      /* P particles (index 0 unused) */
  otree* part2;     /* resampling scratch */
  int32_t* arena;   /* row-index segments */
  int64_t arena_len, arena_cap;
  int32_t* vi;      /* p */
  int32_t* last_ids;
  int32_t n_last;
  pgb_counters ctr;
  double sse0; /* reference particle (Normal) */
  int64_t ll0; /* reference particle (Bernoulli families) */
  int have_data, have_y;
  pgb_loglik_fn cb_fn; /* PGB_FAMILY_CALLBACK */
  void* cb_ctx;
  int cb_failed;
};

/* ------------------------------------------------------------------ helpers */
static void copy_tree(otree* d, const otree* s) {
  d->n_nodes = s->n_nodes;
  d->n_leaves = s->n_leaves;
  d->next_pop = s->next_pop;
  d->sse_tot = s->sse_tot;
  d->sse_orph = s->sse_orph;
  d->ll_tot = s->ll_tot;
  d->ll_orph = s->ll_orph;
  memcpy(d->nd, s->nd, sizeof(onode) * (size_t)s->n_nodes);
}

static void build_cdf(pgb_handle* h) {
  /* [U] SampleSplittingVariable: cumulative split weights (exact integer prefix sums) */
  int64_t c = 0;
  for (int j = 0; j < h->s.p; ++j) {
    c += h->alpha_vec[j];
    h->cdf[j] = c;
  }
}

static int sample_var(const pgb_handle* h, double u) {
  /* [U] rvs(): inverse CDF on one uniform */
  return pgb_sample_var(h->cdf, h->s.p, u);
}

static int64_t arena_alloc(pgb_handle* h, int64_t len) {
  if (h->arena_len + len > h->arena_cap) {
    int64_t nc = h->arena_cap * 2;
    while (nc < h->arena_len + len) nc *= 2;
    int32_t* na = (int32_t*)realloc(h->arena, sizeof(int32_t) * (size_t)nc);
    if (!na) abort();
    h->arena = na;
    h->arena_cap = nc;
  }
  int64_t off = h->arena_len;
  h->arena_len += len;
  return off;
}

/* ------------------------------------------------------------------ ABI: lifecycle */
int pgb_create(const pgb_settings* s, void* stream, pgb_handle** out) {
  (void)stream;
  if (!s || !out) return fail(PGB_E_INVALID, "null argument");
  if (s->n < 1 || s->p < 1 || s->m < 1) return fail(PGB_E_INVALID, "n, p, m must be >= 1");
  if (s->num_particles < 2 || s->num_particles > PGB_MAX_PARTICLES)
    return fail(PGB_E_INVALID, "num_particles must be in [2, PGB_MAX_PARTICLES]");
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
  pgb_handle* h = (pgb_handle*)calloc(1, sizeof *h);
  if (!h) return fail(PGB_E_NOMEM, "calloc");
  h->s = *s;
  h->sc = pgb_make_scales(s->n, s->range_exp);
  int64_t n = s->n;
  int p = s->p, m = s->m, P = s->num_particles, K = s->n_outputs;
  h->X = (double*)malloc(sizeof(double) * (size_t)n * p);
  h->y = (double*)malloc(sizeof(double) * n);
  h->off = (double*)calloc((size_t)n * (size_t)s->n_outputs, sizeof(double)); /* [K][n] */
  h->rules = (int32_t*)calloc(p, sizeof(int32_t));
  h->alpha_vec = (int64_t*)malloc(sizeof(int64_t) * p);
  h->cdf = (int64_t*)malloc(sizeof(int64_t) * p);
  h->col_has_nan = (int*)calloc(p, sizeof(int));
  h->col_ex = (int*)calloc(p, sizeof(int));
  h->lin_R = pgb_pow2(s->range_exp - 1);
  h->inv_R = pgb_pow2(1 - s->range_exp);
  h->st = (double*)malloc(sizeof(double) * n * K); /* [K][n] */
  h->r = (double*)malloc(sizeof(double) * n);
  h->oldv = (double*)malloc(sizeof(double) * n * K);
  h->rs_mean = (double*)calloc((size_t)n * K, sizeof(double));
  h->rs_m2 = (double*)calloc((size_t)n * K, sizeof(double));
  h->trees = (otree*)calloc(m, sizeof(otree));
  h->lid = (uint8_t*)calloc((size_t)m * n, 1);
  h->part = (otree*)calloc(P, sizeof(otree));
  h->part2 = (otree*)calloc(P, sizeof(otree));
  h->arena_cap = 4 * n + 1024;
  h->arena = (int32_t*)malloc(sizeof(int32_t) * (size_t)h->arena_cap);
  h->vi = (int32_t*)calloc(p, sizeof(int32_t));
  h->last_ids = (int32_t*)calloc(m, sizeof(int32_t));
  for (int64_t i = 0; i < n; ++i) h->arena[i] = (int32_t)i; /* root segment: all rows, ascending */
  for (int64_t i = 0; i < n * K; ++i) h->st[i] = s->init_sum;
  for (int t = 0; t < m; ++t) {
    otree* T = &h->trees[t];
    T->n_nodes = 1;
    T->n_leaves = 1;
    T->next_pop = 1;
    onode* z = &T->nd[0];
    memset(z, 0, sizeof *z);
    z->var = -1;
    z->left = z->right = -1;
    z->cnt = n;
    z->value = s->init_leaf;
    z->svar = -1;
    for (int k = 1; k < K; ++k) z->valx[k - 1] = s->init_leaf;
  }
  h->leaf_sd = s->init_leaf_sd;
  for (int k = 1; k < K; ++k) h->leaf_sdx[k - 1] = s->init_leaf_sd;
  h->inv_sigma2 = 1.0;
  h->lik_param2 = 1.0;
  *out = h;
  return PGB_OK;
}

int pgb_destroy(pgb_handle* h) {
  if (!h) return PGB_OK;
  free(h->X); free(h->y); free(h->off); free(h->rules); free(h->alpha_vec); free(h->cdf); free(h->col_has_nan); free(h->col_ex);
  free(h->st); free(h->r); free(h->oldv); free(h->rs_mean); free(h->rs_m2); free(h->trees);
  free(h->lid); free(h->part); free(h->part2); free(h->arena); free(h->vi); free(h->last_ids);
  free(h);
  return PGB_OK;
}

int pgb_set_data(pgb_handle* h, const double* X, int64_t ldx, const int32_t* rules,
                 const double* split_prior) {
  if (!h || !X || !rules || !split_prior) return fail(PGB_E_INVALID, "null argument");
  int64_t n = h->s.n;
  int p = h->s.p;
  if (ldx < p) return fail(PGB_E_INVALID, "ldx < p");
  double mx = 0.0;
  for (int j = 0; j < p; ++j) {
    if (!(split_prior[j] > 0.0)) return fail(PGB_E_INVALID, "split_prior must be positive");
    if (split_prior[j] > mx) mx = split_prior[j];
  }
  h->max_prior = mx;
  h->alpha_unit = pgb_alpha_unit(mx);
  for (int j = 0; j < p; ++j) {
    if (rules[j] != PGB_RULE_CONTINUOUS && rules[j] != PGB_RULE_ONEHOT && rules[j] != PGB_RULE_SUBSET)
      return fail(PGB_E_UNSUPPORTED, "unknown split rule");
    h->rules[j] = rules[j];
    h->alpha_vec[j] = pgb_alpha_init(split_prior[j], mx);
    int has = 0;
    for (int64_t i = 0; i < n; ++i) {
      double x = X[i * ldx + j];
      h->X[(size_t)j * n + i] = x;
      if (x != x) has = 1;
    }
    h->col_has_nan[j] = has;
    double amax = 0.0;
    for (int64_t i = 0; i < n; ++i) {
      double a = h->X[(size_t)j * n + i];
      a = a < 0.0 ? -a : a;
      if (a > amax) amax = a; /* NaN compares false */
    }
    h->col_ex[j] = pgb_col_exponent(amax);
    if (rules[j] == PGB_RULE_SUBSET) /* category codes outside 0 .. 51 are refused, not clamped */
      for (int64_t i = 0; i < n; ++i) {
        double x = h->X[(size_t)j * n + i];
        if (x == x && !(x >= 0.0 && x <= (double)(PGB_SUBSET_BITS - 1) && x == (double)(int)x)) {
          static __thread char msg[128];
          snprintf(msg, sizeof msg, "SubsetSplit column %d: categories must be integer codes in [0, %d) (NaN = missing)",
                   j, PGB_SUBSET_BITS);
          return fail(PGB_E_INVALID, msg);
        }
      }
  }
  build_cdf(h);
  h->have_data = 1;
  return PGB_OK;
}

int pgb_set_response(pgb_handle* h, const double* y) {
  if (!h || !y) return fail(PGB_E_INVALID, "null argument");
  for (int64_t i = 0; i < h->s.n; ++i)
    if (!(y[i] - y[i] == 0.0)) {
      h->have_y = 0;
      return fail(PGB_E_INVALID, "the response has non-finite values");
    }
  memcpy(h->y, y, sizeof(double) * h->s.n);
  h->have_y = 1;
  return PGB_OK;
}

int pgb_set_offset(pgb_handle* h, const double* off) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  if (h->s.family == PGB_FAMILY_NORMAL)
    return fail(PGB_E_UNSUPPORTED, "offsets are for the per-row families (a Normal model fits observed - offset)");
  if (off) {
    for (int64_t i = 0; i < h->s.n * h->s.n_outputs; ++i)
      if (!(off[i] - off[i] == 0.0) || !(fabs(off[i]) <= PGB_MAX_OFFSET)) {  /* (a linear predictor must be finite and of bounded size; the offset is zeroed) */
        memset(h->off, 0, sizeof(double) * h->s.n * h->s.n_outputs);
        return fail(PGB_E_INVALID, "the offset has non-finite values or values beyond +-1e6 (PGB_MAX_OFFSET)");
      }
    memcpy(h->off, off, sizeof(double) * h->s.n * h->s.n_outputs);
  } else {
    memset(h->off, 0, sizeof(double) * h->s.n * h->s.n_outputs);
  }
  return PGB_OK;
}

int pgb_set_likelihood(pgb_handle* h, const double* params, int32_t n_params) {
  if (!h || !params) return fail(PGB_E_INVALID, "null argument");
  if (h->s.family == PGB_FAMILY_NORMAL) {
    if (n_params != 1 || !(params[0] > 0.0)) return fail(PGB_E_INVALID, "NORMAL needs sigma > 0");
    h->inv_sigma2 = 1.0 / (params[0] * params[0]);
  } else if (h->s.family == PGB_FAMILY_NEGBIN_LOG || h->s.family == PGB_FAMILY_GAMMA_LOG) {
    if (n_params != 1 || !(params[0] > 0.0)) return fail(PGB_E_INVALID, "NEGBIN_LOG / GAMMA_LOG need alpha > 0");
    h->inv_sigma2 = params[0];
  } else if (h->s.family == PGB_FAMILY_ASYMLAPLACE) {
    if (n_params != 2 || !(params[0] > 0.0) || !(params[1] > 0.0) || !(params[1] < 1.0))
      return fail(PGB_E_INVALID, "ASYMLAPLACE needs b > 0 and 0 < q < 1");
    h->inv_sigma2 = params[0];
    h->lik_param2 = params[1];
  } else if (h->s.family == PGB_FAMILY_STUDENT_T) {
    if (n_params != 2 || !(params[0] > 0.0) || !(params[1] > 0.0))
      return fail(PGB_E_INVALID, "STUDENT_T needs sigma > 0 and nu > 0");
    h->inv_sigma2 = params[0];
    h->lik_param2 = params[1];
  } else if (n_params != 0) {
    return fail(PGB_E_INVALID, "this family has no parameters");
  }
  return PGB_OK;
}

/* ------------------------------------------------------------------ one tree update */
/* per-row log-likelihood of the non-Normal families at linear predictor(s) mu[0..K-1] */
static double o_loglik(const pgb_handle* h, int64_t row, double y, const double* mu) {
  if (h->s.family == PGB_FAMILY_CALLBACK) { /* the host callback, one row at a time */
    double out = 0.0;
    if (!h->cb_fn || h->cb_fn(h->cb_ctx, &row, &y, mu, 1, &out) != 0) {
      ((pgb_handle*)h)->cb_failed = 1;
      return -2047.0;
    }
    return pgb_clamp_loglik(out);
  }
  /* inv_sigma2 doubles as "the family's scalar parameter" for the non-Normal families */
  return pgb_loglikq(h->s.family, h->s.n_outputs, y, mu, h->inv_sigma2, h->lik_param2);
}

static void o_tree_begin(pgb_handle* h, int tree_id) {
  /* [U] sum_trees_noi = sum_trees - old_tree.predict(); init_particles */
  const pgb_settings* s = &h->s;
  int64_t n = s->n;
  const int K = s->n_outputs;
  const otree* T = &h->trees[tree_id];
  static __thread double lv[PGB_MAX_OUTPUTS][256];
  for (int o = 0; o < K; ++o)
    for (int k = 0; k < 256; ++k) lv[o][k] = 0.0;
  for (int k = 0; k < T->n_nodes; ++k)
    if (T->nd[k].var < 0) {
      lv[0][T->nd[k].label] = T->nd[k].value;
      for (int o = 1; o < K; ++o) lv[o][T->nd[k].label] = T->nd[k].valx[o - 1];
    }
  const uint8_t* lid = h->lid + (size_t)tree_id * n;
  /* linear response: label -> (slope, xbar, column) of the tree being replaced */
  static __thread double lb[256], lx[256], lbx[PGB_MAX_OUTPUTS - 1][256];
  static __thread int lj[256];
  for (int k = 0; k < 256; ++k) { lb[k] = 0.0; lx[k] = 0.0; lj[k] = -1; }
  if (s->response != PGB_RESPONSE_CONSTANT)
    for (int k = 0; k < T->n_nodes; ++k)
      if (T->nd[k].var < 0) {
        lb[T->nd[k].label] = T->nd[k].slope;
        lx[T->nd[k].label] = T->nd[k].xbar;
        lj[T->nd[k].label] = T->nd[k].svar;
        for (int o = 1; o < K; ++o) lbx[o - 1][T->nd[k].label] = T->nd[k].svar >= 0 ? T->nd[k].slopex[o - 1] : 0.0;
      }
  unsigned sat = 0;
  int64_t A = 0, B = 0, C = 0, E0 = 0;
  int64_t Ax[PGB_MAX_OUTPUTS - 1] = {0};
  const int normal = s->family == PGB_FAMILY_NORMAL;
  for (int64_t i = 0; i < n; ++i) {
    double o = lv[0][lid[i]];
    if (lj[lid[i]] >= 0) o = pgb_leaf_pred(o, lb[lid[i]], lx[lid[i]], h->X[(size_t)lj[lid[i]] * n + i]);
    double noi = h->st[i] - o;
    h->oldv[i] = o;
    A += pgb_quant(h->st[i], h->sc.c1, &sat);
    if (normal) {
      double r = h->y[i] - noi;
      h->r[i] = r;
      B += pgb_quant(r, h->sc.c1, &sat);
      C += pgb_quant(r * r, h->sc.c2, &sat);
      double e = r - o;
      E0 += pgb_quant(e * e, h->sc.c2, &sat);
    } else {
      /* non-Normal families: C = log-lik of a fresh stump, E0 = log-lik of the current tree */
      double mu_stump[PGB_MAX_OUTPUTS], mu_cur[PGB_MAX_OUTPUTS];
      mu_stump[0] = (noi + h->off[i]) + s->init_leaf;
      mu_cur[0] = h->st[i] + h->off[i];
      for (int k = 1; k < K; ++k) {
        const double stk = h->st[(size_t)k * n + i];
        double ok = lv[k][lid[i]];
        if (lj[lid[i]] >= 0)
          ok = pgb_leaf_pred(ok, lbx[k - 1][lid[i]], lx[lid[i]], h->X[(size_t)lj[lid[i]] * n + i]);
        const double noik = stk - ok;
        h->oldv[(size_t)k * n + i] = ok;
        Ax[k - 1] += pgb_quant(stk, h->sc.c1, &sat);
        mu_stump[k] = (noik + h->off[(size_t)k * n + i]) + s->init_leaf;
        mu_cur[k] = stk + h->off[(size_t)k * n + i];
      }
      h->r[i] = 0.0;
      if (s->family == PGB_FAMILY_CATEGORICAL && s->response == PGB_RESPONSE_CONSTANT) {
        /* softmax, constant leaves: the stump in the factorised form of the contract (pgb_loglik_cat_f) -- every
           output predicts init_leaf, so d = 0 and w = 1 exactly and the value is a_c - log sum_k E_k */
        const pgb_lltabs tb = pgb_lltabs_default();
        double eta[PGB_MAX_OUTPUTS], v0[PGB_MAX_OUTPUTS], d0[PGB_MAX_OUTPUTS], w0[PGB_MAX_OUTPUTS];
        eta[0] = noi + h->off[i];
        for (int k = 1; k < K; ++k)
          eta[k] = (h->st[(size_t)k * n + i] - h->oldv[(size_t)k * n + i]) + h->off[(size_t)k * n + i];
        for (int k = 0; k < K; ++k) v0[k] = s->init_leaf;
        const int fast = pgb_cat_side(K, v0, tb.expt, d0, w0);
        C += pgb_quant(pgb_loglik_cat_f(K, h->y[i], eta, v0, d0, w0, fast, &tb), h->sc.cl, &sat);
      } else
      C += pgb_quant(o_loglik(h, i, h->y[i], mu_stump), h->sc.cl, &sat);
      E0 += pgb_quant(o_loglik(h, i, h->y[i], mu_cur), h->sc.cl, &sat);
    }
  }
  h->ctr.saturations += sat;
  h->sse0 = (double)E0 * h->sc.inv_c2;
  h->ll0 = E0;
  h->arena_len = n; /* keep the root segment, drop everything else */
  for (int q = 1; q < s->num_particles; ++q) {
    otree* Pq = &h->part[q];
    Pq->n_nodes = 1;
    Pq->n_leaves = 1;
    Pq->next_pop = 0;
    onode* z = &Pq->nd[0];
    memset(z, 0, sizeof *z);
    z->var = -1;
    z->left = z->right = -1;
    z->depth = 0;
    z->label = 0;
    z->svar = -1;
    z->cnt = n;
    z->q_st = A;
    z->q_r = B;
    z->q_r2 = C;
    z->value = s->init_leaf;
    for (int k = 1; k < K; ++k) {
      z->q_stx[k - 1] = Ax[k - 1];
      z->valx[k - 1] = s->init_leaf;
    }
    z->sse = pgb_leaf_sse(n, B, C, z->value, h->sc.inv_c1, h->sc.inv_c2);
    z->ll = C;
    z->seg = 0;
    Pq->sse_tot = z->sse;
    Pq->sse_orph = 0.0;
    Pq->ll_tot = C;
    Pq->ll_orph = 0;
  }
}

/* non-Normal families: fixed-point log-likelihood of `cnt` rows predicting the K-vector `v` from
 * this tree ([U] update_weight restricted to the rows whose prediction changed). */
static int64_t o_seg_loglik_lin(pgb_handle* h, const int32_t* seg, int64_t cnt, const double* v, double slope,
                                const double* slopex, double xbar, int svar) {
  unsigned sat = 0;
  int64_t acc = 0;
  const int K = h->s.n_outputs;
  const int64_t n = h->s.n;
  if (h->s.family == PGB_FAMILY_CATEGORICAL && h->s.response == PGB_RESPONSE_CONSTANT) {
    /* softmax, constant leaves: the factorised form of the contract (pgb_loglik_cat_f) -- the part of the
       (particle, child) once per segment, the part of the row per row */
    const pgb_lltabs tb = pgb_lltabs_default();
    double d[PGB_MAX_OUTPUTS], w[PGB_MAX_OUTPUTS];
    const int fast = pgb_cat_side(K, v, tb.expt, d, w);
    for (int64_t k = 0; k < cnt; ++k) {
      int32_t i = seg[k];
      double eta[PGB_MAX_OUTPUTS];
      for (int o = 0; o < K; ++o)
        eta[o] = (h->st[(size_t)o * n + i] - h->oldv[(size_t)o * n + i]) + h->off[(size_t)o * n + i];
      acc += pgb_quant(pgb_loglik_cat_f(K, h->y[i], eta, v, d, w, fast, &tb), h->sc.cl, &sat);
    }
    h->ctr.saturations += sat;
    return acc;
  }
  for (int64_t k = 0; k < cnt; ++k) {
    int32_t i = seg[k];
    double mu[PGB_MAX_OUTPUTS];
    for (int o = 0; o < K; ++o) {
      double vo = v[o];
      if (svar >= 0) vo = pgb_leaf_pred(vo, o ? slopex[o - 1] : slope, xbar, h->X[(size_t)svar * n + i]);
      mu[o] = ((h->st[(size_t)o * n + i] - h->oldv[(size_t)o * n + i]) + h->off[(size_t)o * n + i]) + vo;
    }
    if (K == 1) {
      double vi = v[0];
      if (svar >= 0) vi = pgb_leaf_pred(vi, slope, xbar, h->X[(size_t)svar * n + i]);
      mu[0] = ((h->st[i] - h->oldv[i]) + h->off[i]) + vi;
    }
    acc += pgb_quant(o_loglik(h, i, h->y[i], mu), h->sc.cl, &sat);
  }
  h->ctr.saturations += sat;
  return acc;
}
static int64_t o_seg_loglik(pgb_handle* h, const int32_t* seg, int64_t cnt, const double* v) {
  return o_seg_loglik_lin(h, seg, cnt, v, 0.0, NULL, 0.0, -1);
}

/* [U] ParticleTree.sample_tree + grow_tree for particle q in round `round`. */
static void o_particle_step(pgb_handle* h, int q, uint32_t round) {
  const pgb_settings* s = &h->s;
  otree* T = &h->part[q];
  if (T->next_pop >= T->n_nodes) return;
  h->ctr.particle_steps += 1;
  int l = T->next_pop++;
  onode nd = T->nd[l];
  uint32_t it = (uint32_t)h->iter;
  pgb_u2 u = pgb_draw2(s->seed, it, round, (uint32_t)q, PGB_RNG_PROPOSE, 0);
  double pl = nd.depth < PGB_MAX_DEPTH ? s->prior_leaf[nd.depth] : 1.0;
  if (!(pl < u.u0)) return;                 /* stays a leaf */
  if (T->n_nodes + 2 > MAXN) return;        /* node cap */
  if (nd.cnt < 2) return;                   /* [U] needs more than one candidate */
  int j = sample_var(h, u.u1);
  const double* xc = h->X + (size_t)j * s->n;
  const int32_t* seg = h->arena + nd.seg;
  double v = 0.0;
  int found = 0;
  for (uint32_t tr = 0; tr < PGB_SELECT_TRIES && !found; ++tr) {
    pgb_u2 us = pgb_draw2(s->seed, it, round, (uint32_t)q, PGB_RNG_SELECT, tr);
    int64_t k = (int64_t)(us.u0 * (double)nd.cnt);
    if (k > nd.cnt - 1) k = nd.cnt - 1;
    double x = xc[seg[k]];
    if (x == x) {
      v = h->rules[j] == PGB_RULE_SUBSET ? pgb_subset_value(us.u1, x) : x;
      found = 1;
    }
  }
  if (!found) return;
  h->ctr.rows_touched += nd.cnt;
  h->ctr.partitions += 1;
  int rule = h->rules[j];
  /* stable partition of the sorted segment; NaN rows fall in neither child [U] */
  int64_t offL = arena_alloc(h, nd.cnt);
  int64_t offR = arena_alloc(h, nd.cnt);
  seg = h->arena + nd.seg; /* arena may have moved */
  int32_t* sl = h->arena + offL;
  int32_t* sr = h->arena + offR;
  int64_t cL = 0, cR = 0, cN = 0;
  int64_t aL = 0, bL = 0, c2L = 0, aN = 0, bN = 0, c2N = 0;
  int64_t aLx[PGB_MAX_OUTPUTS - 1] = {0}, aNx[PGB_MAX_OUTPUTS - 1] = {0};
  const int K = s->n_outputs;
  const int normal = s->family == PGB_FAMILY_NORMAL;
  int32_t* sn = NULL; /* NaN-dropped rows (non-Normal families need their log-likelihood) */
  if (!normal && h->col_has_nan[j]) sn = (int32_t*)malloc(sizeof(int32_t) * (size_t)nd.cnt);
  /* linear response: sums of u = x 2^-ex over the two children (see pgb_lin_fit) */
  const int lin = s->response != PGB_RESPONSE_CONSTANT;
  const double uscale = pgb_pow2(-h->col_ex[j]);
  int64_t uL[4] = {0, 0, 0, 0}, uR[4] = {0, 0, 0, 0}; /* q_u, q_uu, q_us, q_ur */
  int64_t uLx[PGB_MAX_OUTPUTS - 1] = {0}, uRx[PGB_MAX_OUTPUTS - 1] = {0}; /* q_us of outputs 1..K-1 */
  for (int64_t k = 0; k < nd.cnt; ++k) {
    int32_t i = seg[k];
    double x = xc[i];
    /* saturation of these very values was already counted in o_tree_begin */
    int64_t qa = pgb_quant(h->st[i], h->sc.c1, NULL);
    int64_t qb = pgb_quant(h->r[i], h->sc.c1, NULL);
    int64_t qc = pgb_quant(h->r[i] * h->r[i], h->sc.c2, NULL);
    if (x != x) {
      if (sn) sn[cN] = i;
      cN++; aN += qa; bN += qb; c2N += qc;
      for (int o = 1; o < K; ++o) aNx[o - 1] += pgb_quant(h->st[(size_t)o * s->n + i], h->sc.c1, NULL);
    } else if (pgb_go_left(rule, x, v)) {
      sl[cL++] = i; aL += qa; bL += qb; c2L += qc;
      for (int o = 1; o < K; ++o) aLx[o - 1] += pgb_quant(h->st[(size_t)o * s->n + i], h->sc.c1, NULL);
      if (lin) {
        const double uu = x * uscale;
        uL[0] += pgb_quant(uu * h->lin_R, h->sc.c1, NULL);
        uL[1] += pgb_quant((uu * uu) * h->lin_R, h->sc.c1, NULL);
        uL[2] += pgb_quant(uu * h->st[i], h->sc.c1, NULL);
        uL[3] += pgb_quant(uu * h->r[i], h->sc.c1, NULL);
        for (int o = 1; o < K; ++o) uLx[o - 1] += pgb_quant(uu * h->st[(size_t)o * s->n + i], h->sc.c1, NULL);
      }
    } else {
      sr[cR++] = i;
      if (lin) {
        const double uu = x * uscale;
        uR[0] += pgb_quant(uu * h->lin_R, h->sc.c1, NULL);
        uR[1] += pgb_quant((uu * uu) * h->lin_R, h->sc.c1, NULL);
        uR[2] += pgb_quant(uu * h->st[i], h->sc.c1, NULL);
        uR[3] += pgb_quant(uu * h->r[i], h->sc.c1, NULL);
        for (int o = 1; o < K; ++o) uRx[o - 1] += pgb_quant(uu * h->st[(size_t)o * s->n + i], h->sc.c1, NULL);
      }
    }
  }
  double zero_v[PGB_MAX_OUTPUTS] = {0};
  /* give back the unused tail of the two segments */
  /* (segments are [offL, offL+cL) and [offR, offR+cR); the slack is simply wasted) */
  if (cR == 0 && pgb_empty_right_fails(rule, s->compat)) {
    /* [U] a one-hot / subset split needs two distinct values: the grow fails and the node stays a leaf.
       (PGB_COMPAT_ONEHOT_EMPTY_CHILD: a one-hot split grows its empty right leaf like a continuous one.)
       Rows with a missing split value have been dropped by the partition; the leaf sheds them
       (an identity when there are none).  Same arithmetic as the HIP backend. */
    onode* pn = &T->nd[l];
    double new_sse = pgb_leaf_sse(cL, bL, c2L, nd.value, h->sc.inv_c1, h->sc.inv_c2);
    T->sse_orph += (double)c2N * h->sc.inv_c2;
    T->sse_tot = (T->sse_tot - nd.sse) + new_sse;
    if (!normal) {
      double pv[PGB_MAX_OUTPUTS];
      pv[0] = nd.value;
      for (int o = 1; o < K; ++o) pv[o] = nd.valx[o - 1];
      int64_t llL = o_seg_loglik(h, sl, cL, pv);
      int64_t llN = sn ? o_seg_loglik(h, sn, cN, zero_v) : 0;
      T->ll_orph += llN;
      T->ll_tot = (T->ll_tot - nd.ll) + llL;
      pn->ll = llL;
      free(sn);
    }
    pn->cnt = cL;
    for (int o = 1; o < K; ++o) pn->q_stx[o - 1] = aLx[o - 1];
    pn->q_st = aL;
    pn->q_r = bL;
    pn->q_r2 = c2L;
    pn->sse = new_sse;
    pn->seg = offL;
    return;
  }
  int64_t aR = nd.q_st - aL - aN, bR = nd.q_r - bL - bN, c2R = nd.q_r2 - c2L - c2N;
  /* the rows dropped by NaN now predict 0 from this tree */
  T->sse_orph += (double)c2N * h->sc.inv_c2;
  pgb_u2 ul = pgb_draw2(s->seed, it, round, (uint32_t)q, PGB_RNG_LEAF, 0);
  double z0, z1;
  pgb_normal2(ul.u0, ul.u1, &z0, &z1);
  int L = T->n_nodes, R = T->n_nodes + 1;
  onode* pn = &T->nd[l];
  pn->var = j;
  pn->split = v;
  pn->left = L;
  pn->right = R;
  onode* a = &T->nd[L];
  onode* b = &T->nd[R];
  memset(a, 0, sizeof *a);
  memset(b, 0, sizeof *b);
  a->var = b->var = -1;
  a->left = a->right = b->left = b->right = -1;
  a->depth = b->depth = nd.depth + 1;
  a->label = nd.label;
  b->label = T->n_leaves;
  a->cnt = cL; a->q_st = aL; a->q_r = bL; a->q_r2 = c2L; a->seg = offL;
  b->cnt = cR; b->q_st = aR; b->q_r = bR; b->q_r2 = c2R; b->seg = offR;
  a->value = pgb_leaf_value(cL, aL, h->sc.inv_c1, (double)s->m, z0, h->leaf_sd);
  b->value = pgb_leaf_value(cR, aR, h->sc.inv_c1, (double)s->m, z1, h->leaf_sd);
  for (int o = 1; o < K; ++o) { /* K-vector leaves: one Box-Muller pair per output */
    pgb_u2 uk = pgb_draw2(s->seed, it, round, (uint32_t)q, PGB_RNG_LEAF, (uint32_t)o);
    double zk0, zk1;
    pgb_normal2(uk.u0, uk.u1, &zk0, &zk1);
    const int64_t aLk = aLx[o - 1], aRk = nd.q_stx[o - 1] - aLx[o - 1] - aNx[o - 1];
    a->q_stx[o - 1] = aLk;
    b->q_stx[o - 1] = aRk;
    a->valx[o - 1] = pgb_leaf_value(cL, aLk, h->sc.inv_c1, (double)s->m, zk0, h->leaf_sdx[o - 1]);
    b->valx[o - 1] = pgb_leaf_value(cR, aRk, h->sc.inv_c1, (double)s->m, zk1, h->leaf_sdx[o - 1]);
  }
  a->sse = pgb_leaf_sse(cL, bL, c2L, a->value, h->sc.inv_c1, h->sc.inv_c2);
  b->sse = pgb_leaf_sse(cR, bR, c2R, b->value, h->sc.inv_c1, h->sc.inv_c2);
  a->svar = b->svar = -1;
  if (lin) { /* [U] fast_linear_fit on the split variable; "mix": a fair coin per child */
    int linL = 1, linR = 1;
    if (s->response == PGB_RESPONSE_MIX) {
      pgb_u2 um = pgb_draw2(s->seed, it, round, (uint32_t)q, PGB_RNG_MIX, 0);
      linL = um.u0 < 0.5;
      linR = um.u1 < 0.5;
    }
    const double xs = pgb_pow2(h->col_ex[j]);
    if (linL) {
      pgb_linfit f = pgb_lin_fit(cL, uL[0], uL[1], uL[2], aL, h->sc.inv_c1, h->inv_R, (double)s->m);
      /* K-vector leaves: one slope per output on the shared regressor; the leaf is linear
         when any of them is non-zero */
      int any = f.slope_u != 0.0;
      for (int o = 1; o < K; ++o) {
        pgb_linfit fk = pgb_lin_fit(cL, uL[0], uL[1], uLx[o - 1], a->q_stx[o - 1], h->sc.inv_c1, h->inv_R, (double)s->m);
        a->slopex[o - 1] = fk.slope_u * uscale;
        any |= fk.slope_u != 0.0;
      }
      if (any) {
        a->svar = j;
        a->slope = f.slope_u * uscale; /* per unit of x */
        a->xbar = f.ubar * xs;
        a->sse = pgb_lin_sse(a->sse, f, uL[3], bL, h->sc.inv_c1);
      }
    }
    if (linR) {
      pgb_linfit f = pgb_lin_fit(cR, uR[0], uR[1], uR[2], aR, h->sc.inv_c1, h->inv_R, (double)s->m);
      int any = f.slope_u != 0.0;
      for (int o = 1; o < K; ++o) {
        pgb_linfit fk = pgb_lin_fit(cR, uR[0], uR[1], uRx[o - 1], b->q_stx[o - 1], h->sc.inv_c1, h->inv_R, (double)s->m);
        b->slopex[o - 1] = fk.slope_u * uscale;
        any |= fk.slope_u != 0.0;
      }
      if (any) {
        b->svar = j;
        b->slope = f.slope_u * uscale;
        b->xbar = f.ubar * xs;
        b->sse = pgb_lin_sse(b->sse, f, uR[3], bR, h->sc.inv_c1);
      }
    }
  }
  T->sse_tot = ((T->sse_tot - nd.sse) + a->sse) + b->sse;
  if (!normal) {
    sl = h->arena + offL;
    sr = h->arena + offR;
    double va[PGB_MAX_OUTPUTS], vb[PGB_MAX_OUTPUTS];
    va[0] = a->value;
    vb[0] = b->value;
    for (int o = 1; o < K; ++o) {
      va[o] = a->valx[o - 1];
      vb[o] = b->valx[o - 1];
    }
    a->ll = o_seg_loglik_lin(h, sl, cL, va, a->slope, a->slopex, a->xbar, a->svar);
    b->ll = o_seg_loglik_lin(h, sr, cR, vb, b->slope, b->slopex, b->xbar, b->svar);
    if (sn) T->ll_orph += o_seg_loglik(h, sn, cN, zero_v);
    T->ll_tot = ((T->ll_tot - nd.ll) + a->ll) + b->ll;
    free(sn);
  }
  T->n_nodes += 2;
  T->n_leaves += 1;
}

static double o_logw(const pgb_handle* h, const otree* T) {
  /* [U] ParticleTree.log_weight = 0 until the particle's first successful grow (compat bit 0) */
  if ((h->s.compat & PGB_COMPAT_FRESH_WEIGHT_ZERO) && T->n_nodes == 1) return 0.0;
  if (h->s.family != PGB_FAMILY_NORMAL) return (double)(T->ll_tot + T->ll_orph) * h->sc.inv_cl;
  return (T->sse_tot + T->sse_orph) * (-0.5 * h->inv_sigma2);
}

/* [U] resample: normalize (softmax + 1e-12) + systematic resampling of particles 1..P-1.
 * Cumulative weights and the inverse-CDF walk are pgb_weights_scan / pgb_pick (numeric contract). */
static void o_resample(pgb_handle* h, uint32_t round) {
  int P = h->s.num_particles, Lc = P - 1;
  double lw[PGB_MAX_PARTICLES], W[PGB_MAX_PARTICLES];
  for (int q = 1; q < P; ++q) lw[q] = o_logw(h, &h->part[q]);
  pgb_weights_scan(lw, 1, Lc, W);
  pgb_u2 u = pgb_draw2(h->s.seed, (uint32_t)h->iter, round, 0, PGB_RNG_RESAMPLE, 0);
  for (int i = 0; i < Lc; ++i) {
    double ui = (u.u0 + (double)i) / (double)Lc;
    int a = pgb_pick(W, 1, Lc, ui);
    copy_tree(&h->part2[i + 1], &h->part[a]);
  }
  otree* t = h->part;
  h->part = h->part2;
  h->part2 = t;
}

static void o_tree_end(pgb_handle* h, int tree_id, int tune) {
  const pgb_settings* s = &h->s;
  int64_t n = s->n;
  int P = s->num_particles;
  double lw[PGB_MAX_PARTICLES], W[PGB_MAX_PARTICLES];
  lw[0] = s->family == PGB_FAMILY_NORMAL ? h->sse0 * (-0.5 * h->inv_sigma2) : (double)h->ll0 * h->sc.inv_cl;
  for (int q = 1; q < P; ++q) lw[q] = o_logw(h, &h->part[q]);
  pgb_weights_scan(lw, 0, P, W);
  pgb_u2 u = pgb_draw2(s->seed, (uint32_t)h->iter, 0, 0, PGB_RNG_FINAL, 0);
  int sel = pgb_pick(W, 0, P, u.u0);
  uint8_t* lid = h->lid + (size_t)tree_id * n;
  otree* T = &h->trees[tree_id];
  if (sel > 0) {
    copy_tree(T, &h->part[sel]);
    memset(lid, PGB_ORPHAN, (size_t)n);
    for (int k = 0; k < T->n_nodes; ++k)
      if (T->nd[k].var < 0) {
        const int32_t* seg = h->arena + T->nd[k].seg;
        for (int64_t c = 0; c < T->nd[k].cnt; ++c) lid[seg[c]] = (uint8_t)T->nd[k].label;
      }
  }
  double lv[256];
  for (int k = 0; k < 256; ++k) lv[k] = 0.0;
  for (int k = 0; k < T->n_nodes; ++k)
    if (T->nd[k].var < 0) lv[T->nd[k].label] = T->nd[k].value;
  static __thread double lb[256], lx[256];
  static __thread int lj[256];
  for (int k = 0; k < 256; ++k) { lb[k] = 0.0; lx[k] = 0.0; lj[k] = -1; }
  if (s->response != PGB_RESPONSE_CONSTANT)
    for (int k = 0; k < T->n_nodes; ++k)
      if (T->nd[k].var < 0) {
        lb[T->nd[k].label] = T->nd[k].slope;
        lx[T->nd[k].label] = T->nd[k].xbar;
        lj[T->nd[k].label] = T->nd[k].svar;
      }
  /* [U] sum_trees = sum_trees_noi + new_tree.predict() */
  if (tune) h->rs_count += 1;
  unsigned sat = 0;
  int64_t qstd = 0;
  int64_t qstdx[PGB_MAX_OUTPUTS - 1] = {0};
  const int K = s->n_outputs;
  for (int o = 0; o < K; ++o) {
    double* st = h->st + (size_t)o * n;
    double* ov = h->oldv + (size_t)o * n;
    double* rmean = h->rs_mean + (size_t)o * n;
    double* rm2 = h->rs_m2 + (size_t)o * n;
    if (o > 0) { /* label -> value (and slope) table of output o */
      for (int k = 0; k < 256; ++k) lv[k] = 0.0;
      for (int k = 0; k < T->n_nodes; ++k)
        if (T->nd[k].var < 0) {
          lv[T->nd[k].label] = T->nd[k].valx[o - 1];
          if (T->nd[k].svar >= 0) lb[T->nd[k].label] = T->nd[k].slopex[o - 1];
        }
    }
    for (int64_t i = 0; i < n; ++i) {
      double nv = lv[lid[i]];
      if (lj[lid[i]] >= 0) nv = pgb_leaf_pred(nv, lb[lid[i]], lx[lid[i]], h->X[(size_t)lj[lid[i]] * n + i]);
      double noi = st[i] - ov[i];
      st[i] = noi + nv;
      if (tune) { /* [U] RunningSd.update (Welford) */
        double cntf = (double)h->rs_count;
        double delta = nv - rmean[i];
        double mean = rmean[i] + delta / cntf;
        double delta2 = nv - mean;
        double m2 = rm2[i] + delta * delta2;
        rmean[i] = mean;
        rm2[i] = m2;
        int64_t qs = pgb_quant(PGB_SQRT(m2 / cntf), h->sc.c1, &sat);
        if (o == 0) qstd += qs; else qstdx[o - 1] += qs;
      }
    }
  }
  h->ctr.saturations += sat;
  if (tune) {
    if (h->iter > s->m) build_cdf(h); /* [U] ssv rebuilt before this tree's counts are added */
    for (int k = 0; k < T->n_nodes; ++k)
      if (T->nd[k].var >= 0) h->alpha_vec[T->nd[k].var] += h->alpha_unit;
    /* [U] leaf_sd = RunningSd.update(new tree's predictions) from the third update on; a running sd of exactly
     * 0 is not adopted (deviation 12, pgb_tuned_leaf_sd) */
    h->leaf_sd = pgb_tuned_leaf_sd(h->leaf_sd, h->iter, qstd, h->sc.inv_c1, n);
    for (int o = 1; o < K; ++o)
      h->leaf_sdx[o - 1] = pgb_tuned_leaf_sd(h->leaf_sdx[o - 1], h->iter, qstdx[o - 1], h->sc.inv_c1, n);
  } else {
    for (int k = 0; k < T->n_nodes; ++k)
      if (T->nd[k].var >= 0) h->vi[T->nd[k].var] += 1;
  }
  h->ctr.tree_updates += 1;
}

static int o_step(pgb_handle* h, int tune) {
  const pgb_settings* s = &h->s;
  if (!h->have_data || !h->have_y) return fail(PGB_E_INVALID, "set_data/set_response first");
  if (s->family == PGB_FAMILY_CALLBACK && !h->cb_fn) return fail(PGB_E_INVALID, "pgb_set_loglik_callback first");
  /* include/pgbart.h: a step abandoned on a callback error poisons the handle until a checkpoint is loaded */
  if (h->cb_failed)
    return fail(PGB_E_STATE, "an earlier step of this sampler was abandoned half-way (log-likelihood "
                             "callback error or stuck state machine): its state is undefined; restore a "
                             "checkpoint (pgb_checkpoint_load) or create a new sampler");
  memset(h->vi, 0, sizeof(int32_t) * s->p);
  int bs = tune ? s->batch_tune : s->batch_draw;
  int upper = h->lower + bs;
  if (upper > s->m) upper = s->m;
  h->n_last = 0;
  for (int tree_id = h->lower; tree_id < upper; ++tree_id) {
    h->iter += 1;
    h->last_ids[h->n_last++] = tree_id;
    o_tree_begin(h, tree_id);
    for (uint32_t round = 0;; ++round) {
      for (int q = 1; q < s->num_particles; ++q) o_particle_step(h, q, round);
      h->ctr.rounds += 1;
      int stop = 1;
      for (int q = 1; q < s->num_particles; ++q)
        if (h->part[q].next_pop < h->part[q].n_nodes) stop = 0;
      if (stop) break;
      o_resample(h, round);
    }
    o_tree_end(h, tree_id, tune);
  }
  h->lower = upper < s->m ? upper : 0;
  return PGB_OK;
}

int pgb_set_output_stream(pgb_handle* h, void* stream) { /* no streams on this backend */
  (void)stream;
  if (!h) return fail(PGB_E_INVALID, "null handle");
  return PGB_OK;
}

int pgb_set_loglik_callback(pgb_handle* h, pgb_loglik_fn fn, void* ctx) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  if (h->s.family != PGB_FAMILY_CALLBACK) return fail(PGB_E_INVALID, "the sampler was not created with the callback family");
  h->cb_fn = fn;
  h->cb_ctx = ctx;
  return PGB_OK;
}

int pgb_step(pgb_handle* h, int32_t tune, double* sum_trees_out, int32_t* vi_out,
             pgb_counters* counters_out) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  int rc = o_step(h, tune);
  if (rc) return rc;
  if (h->cb_failed) return fail(PGB_E_STATE, "the log-likelihood callback reported an error");
  if (sum_trees_out) memcpy(sum_trees_out, h->st, sizeof(double) * h->s.n * h->s.n_outputs);
  if (vi_out) memcpy(vi_out, h->vi, sizeof(int32_t) * h->s.p);
  if (counters_out) *counters_out = h->ctr;
  return PGB_OK;
}

/* host outputs: on this backend "device" memory is host memory, so the two calls coincide */
int pgb_step_host(pgb_handle* h, int32_t tune, double* sum_trees_out, int32_t* vi_out,
                  pgb_counters* counters_out) {
  return pgb_step(h, tune, sum_trees_out, vi_out, counters_out);
}

int pgb_step_async(pgb_handle* h, int32_t tune, int32_t n_steps) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  for (int i = 0; i < n_steps; ++i) {
    int rc = o_step(h, tune);
    if (rc) return rc;
  }
  return PGB_OK;
}

int pgb_sync(pgb_handle* h, pgb_counters* counters_out) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  if (counters_out) *counters_out = h->ctr;
  return PGB_OK;
}

int pgb_export_trees(pgb_handle* h, int32_t which, pgb_tree_arrays* out) {
  if (!h || !out) return fail(PGB_E_INVALID, "null argument");
  if (h->cb_failed) return fail(PGB_E_STATE, "an earlier step of this sampler was abandoned half-way; restore a checkpoint");
  int nt = which == 0 ? h->n_last : h->s.m;
  int total = 0;
  for (int t = 0; t < nt; ++t) total += h->trees[which == 0 ? h->last_ids[t] : t].n_nodes;
  if (!out->var) {
    out->n_trees = nt;
    out->n_outputs = h->s.n_outputs;
    out->total_nodes = total;
    return PGB_OK;
  }
  if (out->n_trees != nt || out->total_nodes != total) return fail(PGB_E_INVALID, "size mismatch");
  int off = 0;
  for (int t = 0; t < nt; ++t) {
    int id = which == 0 ? h->last_ids[t] : t;
    const otree* T = &h->trees[id];
    out->tree_id[t] = id;
    out->node_off[t] = off;
    for (int k = 0; k < T->n_nodes; ++k) {
      const onode* z = &T->nd[k];
      out->var[off + k] = z->var;
      out->split[off + k] = z->var >= 0 ? z->split : 0.0;
      out->left[off + k] = z->left;
      out->right[off + k] = z->right;
      out->count[off + k] = z->cnt;
      /* the trees describe themselves (include/pgbart.h): utils.py:124-127 rebuilds predictors without rules */
      if (out->rule) out->rule[off + k] = z->var >= 0 ? h->rules[z->var] : PGB_RULE_CONTINUOUS;
      const int K = h->s.n_outputs;
      out->value[(size_t)(off + k) * K] = z->var < 0 ? z->value : 0.0;
      if (out->slope && out->xbar && out->svar) {
        const int islin = z->var < 0 && h->s.response != PGB_RESPONSE_CONSTANT && z->svar >= 0;
        out->slope[(size_t)(off + k) * K] = islin ? z->slope : 0.0;
        for (int o = 1; o < K; ++o) out->slope[(size_t)(off + k) * K + o] = islin ? z->slopex[o - 1] : 0.0;
        out->xbar[off + k] = islin ? z->xbar : 0.0;
        out->svar[off + k] = islin ? z->svar : -1;
      }
      for (int o = 1; o < K; ++o) out->value[(size_t)(off + k) * K + o] = z->var < 0 ? z->valx[o - 1] : 0.0;
    }
    off += T->n_nodes;
  }
  out->node_off[nt] = off;
  return PGB_OK;
}

int pgb_export_trees_packed(pgb_handle* h, int32_t which, void* host_buf, int64_t cap_bytes, int64_t* bytes_out) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  int rc = pgb_export_trees_packed_via(h, which, host_buf, cap_bytes, bytes_out, h->s.response != PGB_RESPONSE_CONSTANT);
  if (rc == PGB_E_NOMEM) return fail(rc, "packed tree record does not fit the buffer (*bytes_out has the size)");
  return rc;
}

int pgb_get_state(pgb_handle* h, double* leaf_sd_out, int64_t* iter_out, int32_t* lower_out) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  if (leaf_sd_out) {
    leaf_sd_out[0] = h->leaf_sd;
    for (int o = 1; o < h->s.n_outputs; ++o) leaf_sd_out[o] = h->leaf_sdx[o - 1];
  }
  if (iter_out) *iter_out = h->iter;
  if (lower_out) *lower_out = h->lower;
  return PGB_OK;
}

int pgb_get_split_weights(pgb_handle* h, double* out) {
  if (!h || !out) return fail(PGB_E_INVALID, "null argument");
  /* in units of the caller's prior: prior_j + number of tuning counts (up to 2^-24 rounding) */
  for (int j = 0; j < h->s.p; ++j)
    out[j] = (double)h->alpha_vec[j] * (h->max_prior * pgb_pow2(-PGB_ALPHA_BITS));
  return PGB_OK;
}

/* ------------------------------------------------------------------ prediction */
/* [U] Tree.predict with `excluded` (CHANGELOG.md:410-411): at a split on an excluded
 * variable or a NaN value, the count-weighted mean of both subtrees. */
static void o_predict_rec(const pgb_tree_arrays* T, int base, int k, const double* x,
                          const uint8_t* excl, int K, double w, double* acc) {
  for (;;) {
    int g = base + k;
    if (T->var[g] < 0) {
      int js = -1; /* linear leaf; a missing / excluded regressor: the mean */
      if (T->svar && T->svar[g] >= 0 && !excl[T->svar[g]] && x[T->svar[g]] == x[T->svar[g]]) js = T->svar[g];
      for (int o = 0; o < K; ++o) {
        double vo = T->value[(size_t)g * K + o];
        if (js >= 0) vo = pgb_leaf_pred(vo, T->slope[(size_t)g * K + o], T->xbar[g], x[js]);
        acc[o] += w * vo;
      }
      return;
    }
    int j = T->var[g];
    double xv = x[j];
    if (excl[j] || xv != xv) {
      int l = T->left[g], r = T->right[g];
      double cl = (double)T->count[base + l], cr = (double)T->count[base + r];
      double tot = cl + cr;
      if (!(tot > 0.0)) return;
      o_predict_rec(T, base, l, x, excl, K, w * (cl / tot), acc);
      o_predict_rec(T, base, r, x, excl, K, w * (cr / tot), acc);
      return;
    }
    int rule = T->rule ? T->rule[g] : PGB_RULE_CONTINUOUS; /* the node's own rule */
    int go_left = pgb_go_left(rule, xv, T->split[g]);
    k = go_left ? T->left[g] : T->right[g];
  }
}

/* A malformed history (truncated file, mismatched m) must be an error, not an out-of-bounds walk:
 * 1 = forest index outside the tree list, 2 = inconsistent node arrays, 3 = split column >= p,
 * 4 = unknown split rule on a node. */
static int pgb_validate_forest(const pgb_tree_arrays* T, const int32_t* fidx, int32_t n_forests, int32_t m,
                               int32_t p) {
  for (int64_t i = 0; i < (int64_t)n_forests * m; ++i)
    if (fidx[i] < 0 || fidx[i] >= T->n_trees) return 1;
  if (T->n_trees < 0 || T->total_nodes < 0) return 2;
  for (int t = 0; t < T->n_trees; ++t) {
    const int base = T->node_off[t], end = T->node_off[t + 1];
    if (base < 0 || end <= base || end > T->total_nodes) return 2;
    for (int g = base; g < end; ++g) {
      if (T->var[g] < 0) continue;
      if (T->var[g] >= p) return 3;
      if (T->rule && T->rule[g] != PGB_RULE_CONTINUOUS && T->rule[g] != PGB_RULE_ONEHOT && T->rule[g] != PGB_RULE_SUBSET)
        return 4;
      if (T->left[g] < 0 || T->right[g] < 0 || T->left[g] >= end - base || T->right[g] >= end - base) return 2;
    }
  }
  return 0;
}

int pgb_predict(const pgb_tree_arrays* trees, const int32_t* forest_tree_idx, int32_t n_forests,
                int32_t m, const double* X, int64_t n_rows, int32_t p, int64_t ldx,
                const int32_t* excluded, int32_t n_excluded, double* out, void* stream) {
  (void)stream;
  if (!trees || !forest_tree_idx || !X || !out) return fail(PGB_E_INVALID, "null argument");
  int K = trees->n_outputs;
  {
    int vrc = pgb_validate_forest(trees, forest_tree_idx, n_forests, m, p);
    if (vrc == 1) return fail(PGB_E_INVALID, "forest_tree_idx entry outside [0, n_trees)");
    if (vrc == 2) return fail(PGB_E_INVALID, "tree arrays are inconsistent (node_off / left / right)");
    if (vrc == 3) return fail(PGB_E_INVALID, "a tree splits on a column X does not have");
    if (vrc == 4) return fail(PGB_E_INVALID, "a split node carries an unknown split rule");
  }
  uint8_t* excl = (uint8_t*)calloc(p, 1);
  for (int e = 0; e < n_excluded; ++e)
    if (excluded[e] >= 0 && excluded[e] < p) excl[excluded[e]] = 1;
  double acc[PGB_MAX_OUTPUTS];
  for (int d = 0; d < n_forests; ++d)
    for (int64_t i = 0; i < n_rows; ++i) {
      for (int o = 0; o < K; ++o) acc[o] = 0.0;
      for (int t = 0; t < m; ++t) {
        int ti = forest_tree_idx[(size_t)d * m + t];
        o_predict_rec(trees, trees->node_off[ti], 0, X + i * ldx, excl, K, 1.0, acc);
      }
      for (int o = 0; o < K; ++o) out[((size_t)d * K + o) * n_rows + i] = acc[o];
    }
  free(excl);
  return PGB_OK;
}

int pgb_profile(pgb_handle* h, int32_t enable, double* kernel_ms_out, int64_t* launches_out) {
  (void)h;
  (void)enable;
  if (kernel_ms_out) *kernel_ms_out = 0.0;
  if (launches_out) *launches_out = 0;
  return PGB_OK;
}

int pgb_profile_clock(pgb_handle* h, double* kernel_ms_out, int64_t* launches_out) {
  (void)h;
  if (kernel_ms_out) *kernel_ms_out = 0.0;
  if (launches_out) *launches_out = 0;
  return PGB_OK;
}

int pgb_profile_kernel(pgb_handle* h, int32_t which, double* kernel_ms_out, int64_t* launches_out,
                       int32_t* workgroups_out) {
  (void)h;
  (void)which;
  if (kernel_ms_out) *kernel_ms_out = 0.0;
  if (launches_out) *launches_out = 0;
  if (workgroups_out) *workgroups_out = 0;
  return PGB_OK;
}

/* ------------------------------------------------------------------ checkpoint / resume
 * The chain image of include/pgbart_image.h: the record every backend writes and reads.  This backend keeps its
 * accepted trees as node tables + one label byte per (tree, row), which is what the image holds; the per-node
 * sufficient statistics and row segments of a tree are particle state and do not outlive the update. */
static int32_t o_total_nodes(const pgb_handle* h) {
  int32_t N = 0;
  for (int t = 0; t < h->s.m; ++t) N += h->trees[t].n_nodes;
  return N;
}

int pgb_checkpoint_size(pgb_handle* h, int64_t* bytes_out) {
  if (!h || !bytes_out) return fail(PGB_E_INVALID, "null argument");
  *bytes_out = pgb_image_bytes(h->s.n, h->s.p, h->s.m, h->s.n_outputs, o_total_nodes(h));
  return PGB_OK;
}

int pgb_checkpoint_save(pgb_handle* h, void* host_buf, int64_t bytes) {
  if (!h || !host_buf) return fail(PGB_E_INVALID, "null argument");
  if (!h->have_data || !h->have_y) return fail(PGB_E_INVALID, "set_data/set_response first");
  if (h->cb_failed) return fail(PGB_E_STATE, "an earlier step of this sampler was abandoned half-way; restore a checkpoint");
  const pgb_settings* s = &h->s;
  const int64_t n = s->n;
  const int K = s->n_outputs, m = s->m, p = s->p;
  const int32_t N = o_total_nodes(h);
  const int64_t need = pgb_image_bytes(n, p, m, K, N);
  if (bytes < need) return fail(PGB_E_INVALID, "checkpoint buffer too small");
  pgb_image_header hd;
  pgb_image_begin(host_buf, need, s, N, pgb_backend_name(), &hd);
  hd.iter = h->iter;
  hd.rs_count = h->rs_count;
  hd.lower = h->lower;
  hd.last_lower = h->n_last > 0 ? h->last_ids[0] : 0;
  hd.last_n = h->n_last;
  hd.leaf_sd[0] = h->leaf_sd;
  for (int o = 1; o < K; ++o) hd.leaf_sd[o] = h->leaf_sdx[o - 1];
  hd.lik_param[0] = h->inv_sigma2;
  hd.lik_param[1] = h->lik_param2;
  hd.ctr = h->ctr;
  memcpy(host_buf, &hd, sizeof hd);
  pgb_image_view v;
  pgb_image_bind(host_buf, &hd, &v);
  memcpy(v.sum_trees, h->st, sizeof(double) * (size_t)n * K);
  memcpy(v.rs_mean, h->rs_mean, sizeof(double) * (size_t)n * K);
  memcpy(v.rs_m2, h->rs_m2, sizeof(double) * (size_t)n * K);
  memcpy(v.alpha, h->alpha_vec, sizeof(int64_t) * (size_t)p);
  memcpy(v.cdf, h->cdf, sizeof(int64_t) * (size_t)p);
  int32_t g = 0;
  for (int t = 0; t < m; ++t) {
    const otree* T = &h->trees[t];
    v.node_off[t] = g;
    for (int k = 0; k < T->n_nodes; ++k, ++g) {
      const onode* z = &T->nd[k];
      const int leaf = z->var < 0;
      const int islin = leaf && s->response != PGB_RESPONSE_CONSTANT && z->svar >= 0;
      v.var[g] = z->var;
      v.left[g] = leaf ? -1 : z->left;
      v.right[g] = leaf ? -1 : z->right;
      v.depth[g] = z->depth;
      v.label[g] = z->label;
      v.svar[g] = islin ? z->svar : -1;
      v.count[g] = z->cnt;
      v.split[g] = leaf ? 0.0 : z->split;
      v.xbar[g] = islin ? z->xbar : 0.0;
      for (int o = 0; o < K; ++o) {
        v.value[(size_t)g * K + o] = !leaf ? 0.0 : o ? z->valx[o - 1] : z->value;
        v.slope[(size_t)g * K + o] = !islin ? 0.0 : o ? z->slopex[o - 1] : z->slope;
      }
    }
  }
  v.node_off[m] = g;
  memcpy(v.lid, h->lid, (size_t)m * (size_t)n);
  return PGB_OK;
}

int pgb_checkpoint_load(pgb_handle* h, const void* host_buf, int64_t bytes) {
  if (!h || !host_buf) return fail(PGB_E_INVALID, "null argument");
  if (!h->have_data || !h->have_y) return fail(PGB_E_INVALID, "set_data/set_response first");
  const char* why = pgb_image_check(host_buf, bytes, &h->s);
  if (why) return fail(PGB_E_INVALID, why);
  const pgb_settings* s = &h->s;
  const int64_t n = s->n;
  const int K = s->n_outputs, m = s->m, p = s->p;
  pgb_image_header hd;
  memcpy(&hd, host_buf, sizeof hd);
  pgb_image_view v;
  pgb_image_bind((void*)host_buf, &hd, &v);
  memcpy(h->st, v.sum_trees, sizeof(double) * (size_t)n * K);
  memcpy(h->rs_mean, v.rs_mean, sizeof(double) * (size_t)n * K);
  memcpy(h->rs_m2, v.rs_m2, sizeof(double) * (size_t)n * K);
  memcpy(h->alpha_vec, v.alpha, sizeof(int64_t) * (size_t)p);
  memcpy(h->cdf, v.cdf, sizeof(int64_t) * (size_t)p);
  for (int t = 0; t < m; ++t) {
    otree* T = &h->trees[t];
    const int32_t base = v.node_off[t], nn = v.node_off[t + 1] - base;
    memset(T, 0, sizeof *T);
    T->n_nodes = nn;
    T->next_pop = nn;
    for (int k = 0; k < nn; ++k) {
      const int32_t g = base + k;
      onode* z = &T->nd[k];
      z->var = v.var[g];
      z->left = v.left[g];
      z->right = v.right[g];
      z->depth = v.depth[g];
      z->label = v.label[g];
      z->split = v.split[g];
      z->cnt = v.count[g];
      z->svar = v.svar[g];
      z->xbar = v.xbar[g];
      z->value = v.value[(size_t)g * K];
      z->slope = v.slope[(size_t)g * K];
      for (int o = 1; o < K; ++o) {
        z->valx[o - 1] = v.value[(size_t)g * K + o];
        z->slopex[o - 1] = v.slope[(size_t)g * K + o];
      }
      if (z->var < 0) T->n_leaves += 1;
    }
  }
  memcpy(h->lid, v.lid, (size_t)m * (size_t)n);
  h->rs_count = hd.rs_count;
  h->iter = hd.iter;
  h->lower = hd.lower;
  h->n_last = hd.last_n;
  for (int i = 0; i < hd.last_n; ++i) h->last_ids[i] = hd.last_lower + i;
  h->leaf_sd = hd.leaf_sd[0];
  for (int o = 1; o < K; ++o) h->leaf_sdx[o - 1] = hd.leaf_sd[o];
  h->inv_sigma2 = hd.lik_param[0];
  h->lik_param2 = hd.lik_param[1];
  h->ctr = hd.ctr;
  memset(h->vi, 0, sizeof(int32_t) * (size_t)p);
  h->cb_failed = 0;
  return PGB_OK;
}

/* ------------------------------------------------------------------ test hooks
 * (pgbo_*: exported only by the oracle so that tests can pin the numeric primitives of
 * pgbart_spec.h against known-answer vectors and libm) */
void pgbo_philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                 uint32_t* out) {
  pgb_u32x4 r = pgb_philox4x32_10(k0, k1, c0, c1, c2, c3);
  for (int i = 0; i < 4; ++i) out[i] = r.v[i];
}
void pgbo_draw2(uint64_t seed, uint32_t iter, uint32_t round, uint32_t particle, uint32_t purpose,
                uint32_t sub, double* out) {
  pgb_u2 u = pgb_draw2(seed, iter, round, particle, purpose, sub);
  out[0] = u.u0;
  out[1] = u.u1;
}
void pgbo_math(const double* x, int64_t n, double* e, double* l, double* s, double* c) {
  for (int64_t i = 0; i < n; ++i) {
    e[i] = pgb_exp(x[i]);
    l[i] = pgb_log(x[i]);
    pgb_sincos2pi(x[i], &s[i], &c[i]);
  }
}
void pgbo_normal2(const double* u0, const double* u1, int64_t n, double* z0, double* z1) {
  for (int64_t i = 0; i < n; ++i) pgb_normal2(u0[i], u1[i], &z0[i], &z1[i]);
}
int64_t pgbo_quant(double x, double scale, uint32_t* sat) { return pgb_quant(x, scale, sat); }
void pgbo_scales(int64_t n, int range_exp, double* out6) {
  pgb_scales s = pgb_make_scales(n, range_exp);
  out6[0] = s.c1; out6[1] = s.c2; out6[2] = s.cl;
  out6[3] = s.inv_c1; out6[4] = s.inv_c2; out6[5] = s.inv_cl;
}
int64_t pgbo_sizeof_settings(void) { return (int64_t)sizeof(pgb_settings); }
int64_t pgbo_sizeof_counters(void) { return (int64_t)sizeof(pgb_counters); }
int64_t pgbo_sizeof_tree_arrays(void) { return (int64_t)sizeof(pgb_tree_arrays); }
void pgbo_scan64(double* x) { pgb_scan64(x); }
int pgbo_pick(const double* lw, int first, int cnt, double u, double* W_out) {
  double W[PGB_MAX_PARTICLES];
  pgb_weights_scan(lw, first, cnt, W);
  if (W_out) memcpy(W_out, W, 64 * sizeof(double)); /* (the test hook hands in 64 entries) */
  return pgb_pick(W, first, cnt, u);
}
void pgbo_loglik(int family, const double* y, const double* mu, int64_t n, double* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = pgb_loglik1(family, y[i], mu[i]);
}
void pgbo_log_ndtr(const double* x, int64_t n, double* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = pgb_log_ndtr(x[i]);
}

/* ---- hooks for the oracle-INDEPENDENT checks of the shared numeric layer (tests/test_spec_independent.py:
 *      every function below is compiled into both backends from include/pgbart_spec.h, so HIP == oracle says
 *      nothing about it; these expose them to scipy / NumPy) */
void pgbo_loglikq(int family, const double* y, const double* mu, int64_t n, double param, double param2,
                  double* out) {
  const pgb_lltabs tb = pgb_lltabs_default();
  for (int64_t i = 0; i < n; ++i) out[i] = pgb_loglik1q(family, y[i], mu[i], param, param2, &tb);
}
/* the table-driven exp / log of the per-row likelihoods (pgb_exp_t, pgb_log_t) */
void pgbo_math_t(const double* x, int64_t n, double* e, double* l) {
  for (int64_t i = 0; i < n; ++i) {
    e[i] = pgb_exp_t(x[i], pgb_tab_exp());
    l[i] = pgb_log_t(x[i], pgb_tab_log());
  }
}
void pgbo_loglik_multi(int family, int K, const double* y, const double* mu /* [n][K] */, int64_t n, double* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = pgb_loglik(family, K, y[i], mu + i * K);
}
/* the factorised softmax of constant leaves (pgb_loglik_cat_f): eta [n][K] row parts, v [K] leaf values */
void pgbo_loglik_cat_f(int K, const double* y, const double* eta, const double* v, int64_t n, double* out) {
  const pgb_lltabs tb = pgb_lltabs_default();
  double d[PGB_MAX_OUTPUTS], w[PGB_MAX_OUTPUTS];
  const int fast = pgb_cat_side(K, v, tb.expt, d, w);
  for (int64_t i = 0; i < n; ++i) out[i] = pgb_loglik_cat_f(K, y[i], eta + i * K, v, d, w, fast, &tb);
}
void pgbo_lin_fit(int64_t cnt, int64_t q_u, int64_t q_uu, int64_t q_us, int64_t q_st, double inv_c1, double inv_R,
                  double m, double* out3) {
  pgb_linfit f = pgb_lin_fit(cnt, q_u, q_uu, q_us, q_st, inv_c1, inv_R, m);
  out3[0] = f.slope_u; out3[1] = f.ubar; out3[2] = f.var_u;
}
double pgbo_lin_sse(double sse_const, double slope_u, double ubar, double var_u, int64_t q_ur, int64_t q_r,
                    double inv_c1) {
  pgb_linfit f = {slope_u, ubar, var_u};
  return pgb_lin_sse(sse_const, f, q_ur, q_r, inv_c1);
}
double pgbo_leaf_sse(int64_t cnt, int64_t q_r, int64_t q_r2, double v, double inv_c1, double inv_c2) {
  return pgb_leaf_sse(cnt, q_r, q_r2, v, inv_c1, inv_c2);
}
double pgbo_leaf_value(int64_t cnt, int64_t q_st, double inv_c1, double m, double z, double leaf_sd) {
  return pgb_leaf_value(cnt, q_st, inv_c1, m, z, leaf_sd);
}
int pgbo_sample_var(const int64_t* S, int p, double u) { return pgb_sample_var(S, p, u); }
int64_t pgbo_alpha_init(double prior, double max_prior) { return pgb_alpha_init(prior, max_prior); }
int64_t pgbo_alpha_unit(double max_prior) { return pgb_alpha_unit(max_prior); }
double pgbo_subset_value(double u1, double x) { return pgb_subset_value(u1, x); }
int pgbo_go_left(int rule, double x, double v) { return pgb_go_left(rule, x, v); }
int pgbo_col_exponent(double amax) { return pgb_col_exponent(amax); }
