/*
 * pgbart.h -- C ABI of the MI355X-native particle-Gibbs BART sampler.
 *
 * This is the drop-in boundary for the hot path of pymc-bart 0.13.1: the native
 * sampler classes the reference imports from the external `bartrs` wheel
 *     pymc_bart/pymc_bart.py:2   PySampler, PyBartSettings, TreeArrays, PosteriorSampler
 *     tests/test_bart.py:4,231   bartrs.PGBART([rv], num_particles=...)
 * The reference binds them through PyO3; a maintainer binds THIS library with the
 * ctypes stubs shown in INTEGRATION.md (pymc_bart_amd/_abi.py is that stub).
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types.  "dev" pointers are device
 *     (HBM) addresses owned by the caller (torch tensors kept alive by Python);
 *     "host" pointers are ordinary memory.  All scratch is owned by the handle.
 *   - every function returns 0 on success or a negative PGB_E_* code; the message
 *     is available from pgb_last_error() (thread-local).  No exception crosses.
 *   - a handle is bound to one device and one HIP stream and is single-threaded.
 *     Calls are stream-synchronous at return unless stated otherwise.
 *   - numerics follow include/pgbart_spec.h: results are a pure function of
 *     (settings, data, seed) and are bit-identical across conforming backends.
 *
 * Limits (compile-time sizes of device records; every violation is reported by pgb_create /
 * pgb_set_data as PGB_E_INVALID or PGB_E_UNSUPPORTED with a message, never silently clamped).
 * The reference bounds none of these (tests/test_bart.py:231 takes any num_particles,
 * :117,155 any shape=(K, n)); upstream's defaults and tests sit far inside all of them.
 *   num_particles        2 .. pgb_max_particles(): 64 in libpgbart_hip.so (one particle per lane of a wave64), 128
 *                                   in libpgbart_hip_p128.so (the same source built with -DPGB_MAX_PARTICLES=128: two
 *                                   particles per lane of the control kernel, cumulative weights as two chained
 *                                   64-entry scans -- pgb_weights_scan); a chain of <= 64 particles is bit-identical
 *                                   on either build.  The Python binding picks the build by num_particles.
 *   n_outputs (K)        1 .. 16   (PGB_MAX_OUTPUTS; K = 2, 3, 4 have unrolled kernel instances, any other K runs in
 *                                  tiles of four outputs: no K-sized array in registers, no scratch)
 *   nodes per tree       <= 255    (leaf labels are bytes; label 255 = dropped row); a tree that
 *                                   would grow past it stops splitting (P ~ 0 under the prior)
 *   tree depth           <= 64     (prior_leaf[64]; upstream cuts its table where P(leaf) >= 0.9999,
 *                                   depth ~ 97 at alpha = 0.95, beta = 2: entries beyond 64 are 1)
 *   SubsetSplit columns  integer category codes 0 .. 51 or NaN (the split value is a 52-bit mask in a double);
 *                        pgb_set_data checks every value and returns PGB_E_INVALID naming the column
 *   response linear/mix  any split rule (a leaf regresses on the column its parent split on, upstream's
 *                        fast_linear_fit; on a SubsetSplit column that is the category code)
 *   tree updates         < 2^32 per chain: the Philox counter takes the low 32 bits of the tree-update counter
 *                        (pgb_get_state's iter), so the random numbers of update i + 2^32 repeat those of update i
 *                        (about 4 days of cfg2-sized asteps at 12 k tree updates/s); nothing is refused
 *   offsets              |offset| <= 1e6 (PGB_MAX_OFFSET), finite; responses finite (pgb_set_offset / pgb_set_response)
 *   n                    < 2^31 - 1024 rows;  p, m >= 1 (bounded by memory: per row the device holds 8 p bytes
 *                        of the design matrix (+ 2 p for its 16-bit order keys when it exceeds the Infinity Cache),
 *                        m bytes of tree labels, 8 x pgb_max_particles() bytes of particle labels (8 generations) and
 *                        ~ 64 K bytes of running statistics -- 30 M rows x 4 columns x 5 trees: 17 GB, in the suite)
 */
#ifndef PGBART_H
#define PGBART_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGB_OK 0
#define PGB_E_INVALID -1  /* bad argument / call order        */
#define PGB_E_DEVICE -2   /* HIP runtime error                */
#define PGB_E_NOMEM -3    /* allocation failed                */
#define PGB_E_STATE -4    /* sampler state machine stuck      */
#define PGB_E_UNSUPPORTED -5

typedef struct pgb_handle pgb_handle;

/* Mirrors bartrs' PyBartSettings (pymc_bart/pymc_bart.py:2) + the BART op
 * attributes the step method snapshots at construction (bart.py:141-158).     */
typedef struct {
  int64_t n;              /* rows of X                                          */
  int32_t p;              /* columns of X                                       */
  int32_t m;              /* number of trees                 (bart.py:119)      */
  int32_t num_particles;  /* incl. the reference particle    (test_bart.py:231) */
  int32_t n_outputs;      /* K: leaves are K-vectors         (test_bart.py:117) */
  int32_t family;         /* PGB_FAMILY_*                                       */
  int32_t batch_tune;     /* trees updated per step while tuning                */
  int32_t batch_draw;     /* trees updated per step after tuning                */
  int32_t range_exp;      /* fixed-point range: |sum_trees|,|y - mu| < 2^range_exp */
  int32_t response;       /* PGB_RESPONSE_*: how leaf values are computed (bart.py:88-90)   */
  int32_t compat;         /* PGB_COMPAT_* bits (pgbart_spec.h): 0 = this sampler; bit 0 / bit 1 switch one of
                             the two distribution-changing deviations from upstream back */
  uint64_t seed;          /* Philox key                                         */
  double init_sum;        /* initial sum_trees value = mean(Y)   (bart.py:148)  */
  double init_leaf;       /* initial leaf value      = mean(Y)/m                */
  double init_leaf_sd;    /* std(Y)/sqrt(m)  (3/sqrt(m) for 0/1 responses)      */
  double prior_leaf[64];  /* P(node at depth d stays a leaf) = 1 - alpha(1+d)^-beta
                             (bart.py:107-109); entries >= 1 stop growth        */
} pgb_settings;

/* Work counters, maintained identically by every backend. */
typedef struct {
  int64_t particle_steps; /* non-reference particles popped with a non-empty queue */
  int64_t tree_updates;   /* trees re-sampled                                    */
  int64_t rows_touched;   /* sum of leaf sizes over executed split partitions    */
  int64_t rounds;         /* SMC rounds                                          */
  int64_t saturations;    /* fixed-point saturation events (should stay 0)       */
  int64_t slots;          /* backend-specific: kernel slots consumed             */
  int64_t partitions;     /* executed split partitions (grow attempts that found a split value) */
} pgb_counters;

/* SoA tree storage: the counterpart of bartrs' TreeArrays (pymc_bart/pymc_bart.py:2).
 * Trees are concatenated; tree t owns nodes [node_off[t], node_off[t+1]).
 * Node 0 of each tree is its root; children indices are tree-local; a leaf has
 * left == right == -1.  `value` holds K doubles per node (leaves only).
 * `count` is the number of training rows in the node ("nvalue" upstream) and is
 * what prediction uses to average over an excluded / NaN split.               */
typedef struct {
  int32_t n_trees;
  int32_t n_outputs;
  int32_t total_nodes;
  int32_t* tree_id;   /* [n_trees]      which of the m slots the tree occupies  */
  int32_t* node_off;  /* [n_trees + 1]                                          */
  int32_t* var;       /* [total_nodes]  split variable, -1 for a leaf           */
  double* split;      /* [total_nodes]  split value                             */
  int32_t* left;      /* [total_nodes]                                          */
  int32_t* right;     /* [total_nodes]                                          */
  int64_t* count;     /* [total_nodes]                                          */
  double* value;      /* [total_nodes * n_outputs]                              */
  /* linear response (NULL = not wanted): a leaf predicts value + slope * (x[svar] - xbar);
   * svar = -1 (slope 0) for constant leaves                                              */
  double* slope;      /* [total_nodes * n_outputs]                              */
  double* xbar;       /* [total_nodes]                                          */
  int32_t* svar;      /* [total_nodes]                                          */
  /* The trees describe themselves: the PGB_RULE_* each split node was grown under (0 for a leaf), so that a
   * history rebuilt by the reference's unmodified call PosteriorSampler.from_history(batches, baseline_forest,
   * op.m, op.n_outputs) (utils.py:124-127 -- it passes no split rules) predicts one-hot / subset splits
   * correctly.  Filled by every export; read by pgb_predict (NULL there = every split is continuous).        */
  int32_t* rule;      /* [total_nodes]                                          */
} pgb_tree_arrays;

const char* pgb_last_error(void);
const char* pgb_backend_name(void); /* "hip-gfx950" or "oracle-cpu" */
int32_t pgb_max_particles(void);    /* PGB_MAX_PARTICLES of this build (64 or 128): the largest num_particles pgb_create
                                       accepts.  Replaces nothing in the reference (tests/test_bart.py:231 takes any int) */

/* Create a sampler.  `stream` is a hipStream_t (NULL = default stream); ignored by
 * CPU backends.  Replaces PGBART.__init__ / PySampler construction.            */
int pgb_create(const pgb_settings* settings, void* stream, pgb_handle** out);
int pgb_destroy(pgb_handle* h);

/* Snapshot the design matrix.  X is row-major n x p doubles with leading
 * dimension ldx (as produced by bart.py:209-210 preprocess_xy), NaN = missing.
 * The backend keeps its own device-resident copy (column-major) -- X need not
 * stay alive.  rules[p] are PGB_RULE_*, split_prior[p] > 0 (bart.py:139: empty
 * prior => all ones, done by the caller).                                      */
int pgb_set_data(pgb_handle* h, const double* X_dev, int64_t ldx, const int32_t* rules_host,
                 const double* split_prior_host);

/* Observed response of the likelihood, n doubles (class index for categorical).  Every value must be finite:
 * PGB_E_INVALID otherwise (the linear predictor of a row is a sum of finite leaf values and of what this call and
 * pgb_set_offset hand in; the per-row likelihood tables are addressed by its bits -- pgbart_spec.h, pgb_lphi_t). */
int pgb_set_response(pgb_handle* h, const double* y_dev);

/* Per-row offset of the linear predictor(s) for the per-row families (everything but NORMAL, where
 * the caller subtracts the other terms from the response instead): the likelihood sees
 * offset + sum_trees -- the contribution of the other additive terms of the model at the current
 * point (a second BART variable, a log-exposure, ...).  EXACTLY K*n doubles, layout [K][n] (the call takes
 * no size: the caller guarantees it, as the ctypes stub does); NULL resets to 0.  A non-finite value, or one beyond
 * +-PGB_MAX_OFFSET = 1e6 (the per-row likelihood tables are addressed by the bits of the linear predictor, without a
 * clamp), is refused with PGB_E_INVALID and the offset is reset to 0 (the chain stays usable). */
int pgb_set_offset(pgb_handle* h, const double* offset_dev);

/* Likelihood parameters at the current point of the other model variables
 * (upstream evaluates model.datalogp; here a closed family).  NORMAL: {sigma}.  */
int pgb_set_likelihood(pgb_handle* h, const double* params_host, int32_t n_params);

/* Likelihoods outside the closed family (family PGB_FAMILY_CALLBACK, one output, constant leaves):
 * `fn` receives n (row index, observed value, linear predictor = sum of trees + offset) triples and writes
 * the n per-row log-likelihoods; it returns 0 on success.  A batch lists one particle after the other, each
 * with ascending row indices (a row may occur once per particle).  Upstream evaluates model.datalogp through PyTensor
 * for every particle; here the device produces the predictors of the rows each round re-labelled, the
 * host evaluates them in one call per round and the fixed-point sums go back to the device -- the slow
 * fallback SURVEY.md section 7 asks for (a slot then costs a device round trip instead of ~17 us).
 * `fn` must be a pure function of each (row, y_i, mu_i).
 * A non-zero return abandons the astep half-way: that call returns PGB_E_STATE and the handle is POISONED --
 * every later pgb_step* / pgb_export_trees returns PGB_E_STATE (the chain's state is undefined) until
 * pgb_checkpoint_load restores an idle image, or the handle is destroyed. */
typedef int (*pgb_loglik_fn)(void* ctx, const int64_t* row, const double* y, const double* mu, int64_t n,
                             double* loglik_out);
int pgb_set_loglik_callback(pgb_handle* h, pgb_loglik_fn fn, void* ctx);

/* One PGBART.astep: re-sample the next batch of trees.
 *   sum_trees_dev_out  K*n doubles (layout [K][n]), may be NULL
 *   vi_counts_host_out p int32: split-variable counts of the accepted trees of
 *                      this step (zeros while tuning), may be NULL
 *   counters_out       cumulative counters, may be NULL                          */
int pgb_step(pgb_handle* h, int32_t tune, double* sum_trees_dev_out, int32_t* vi_counts_host_out,
             pgb_counters* counters_out);

/* The same astep with HOST outputs -- what PGBART.astep hands back to PyMC's trace (SURVEY.md 8a
 * a2: sum_trees (K,n) device->host per step): sum_trees_host_out receives K*n doubles (device-accessible
 * pinned memory -- hipHostMalloc -- is written by the export kernel itself; pageable memory works through one
 * more copy), vi / counters as in pgb_step.  The
 * trees this step re-sampled are fetched in the same device->host transaction, so a following
 * pgb_export_trees(h, 0, ...) is served from host memory without touching the device.           */
int pgb_step_host(pgb_handle* h, int32_t tune, double* sum_trees_host_out, int32_t* vi_counts_host_out,
                  pgb_counters* counters_out);

/* Optional: a second stream (a hipStream_t of the sampler's device, owned by the caller) on which the results of
 * pgb_step_host leave the device -- the export kernel and the copy of sum_trees then run past the few idle slots
 * still queued on the sampler's own stream instead of behind them.  NULL (the default): the sampler's stream.
 * HIP multiplexes streams onto 4 hardware queues; give every sampler of a process the SAME output stream (an
 * otherwise idle one) so that it does not take a queue away from a concurrently running chain.  Ignored by CPU
 * backends.                                                                                                   */
int pgb_set_output_stream(pgb_handle* h, void* stream);

/* Asynchronous variant for throughput runs: pgb_step_async starts `n_steps` asteps and RETURNS
 * while they run (a worker thread owned by the handle feeds the device state machine; CPU backends
 * may run them before returning); pgb_sync waits for them, reports their error if any and fills
 * the counters.  Every other call on the handle waits for a running job first.                 */
int pgb_step_async(pgb_handle* h, int32_t tune, int32_t n_steps);
int pgb_sync(pgb_handle* h, pgb_counters* counters_out);

/* Tree export.  Call once with out->... pointers NULL to get sizes (n_trees,
 * total_nodes filled), allocate, call again.  which = 0: the trees updated by the
 * last step (a "batch"); which = 1: all m current trees (a "baseline forest").  */
int pgb_export_trees(pgb_handle* h, int32_t which, pgb_tree_arrays* out);

/* The same export as ONE self-describing record in a caller buffer, in one call (include/pgbart_pack.h
 * documents the record: a 4-int header, the int32 arrays, the 8-byte arrays; the linear-response arrays only
 * when the sampler has linear leaves).  *bytes_out receives the record size; when cap_bytes is too small
 * (or host_buf is NULL) the call returns PGB_E_NOMEM and *bytes_out says how much is needed.  This is the
 * form the per-draw batches of the tree history travel in (utils.py:124-127).                          */
int pgb_export_trees_packed(pgb_handle* h, int32_t which, void* host_buf, int64_t cap_bytes, int64_t* bytes_out);

/* Current sampler scalars: leaf_sd[K], iter, lower cursor. */
int pgb_get_state(pgb_handle* h, double* leaf_sd_out, int64_t* iter_out, int32_t* lower_out);
/* Current split-variable weights alpha_vec[p] (prior + tuning counts). */
int pgb_get_split_weights(pgb_handle* h, double* alpha_vec_host_out);

/* Prediction from stored trees (PosteriorSampler.sample_posterior, utils.py:66-69):
 * out[d][k][row] = sum over the trees of forest d of the leaf value reached by
 * X[row,:]; at a split on an excluded variable or a NaN value the result is the
 * count-weighted mean of both subtrees.  `forest_tree_idx` is [n_forests][m]
 * indices into `trees`.  X_dev row-major n_rows x p; out_dev n_forests*K*n_rows.
 * How a split sends a row left (x <= v, x == v, x in the subset v) is read from the
 * node itself (trees_host->rule): the call takes no per-column rules, as the
 * reference's from_history / sample_posterior take none.                        */
int pgb_predict(const pgb_tree_arrays* trees_host, const int32_t* forest_tree_idx_host,
                int32_t n_forests, int32_t m, const double* X_dev, int64_t n_rows, int32_t p,
                int64_t ldx, const int32_t* excluded_host, int32_t n_excluded, double* out_dev,
                void* stream);

/* Profiling aid for bench.py: when enabled, the backend brackets every launch of
 * its dominant kernel with events on its stream and accumulates their duration. */
int pgb_profile(pgb_handle* h, int32_t enable, double* kernel_ms_out, int64_t* launches_out);
/* The same region seen by the device clock: per launch, last reading of any workgroup minus the
 * first reading of any workgroup (the interval a kernel trace reports); valid after
 * pgb_profile(h, 0, ...).  CPU backends report 0. */
int pgb_profile_clock(pgb_handle* h, double* kernel_ms_out, int64_t* launches_out);
/* Per-kernel view of the last profiled region (valid after pgb_profile(h, 0, ...)): which = 0 the
 * control kernel, 1 the row pass, 2 the per-row log-likelihood pass.
 * Total event time, launches, and the workgroups of one launch.  CPU backends report 0.          */
int pgb_profile_kernel(pgb_handle* h, int32_t which, double* kernel_ms_out, int64_t* launches_out,
                       int32_t* workgroups_out);

/* Checkpoint / resume of one chain (what pickling the reference's step method into a PyMC worker
 * process carries: reference SURVEY 8b "must be picklable"; tree hand-off bart.py:134-135).
 * The blob is the CHAIN IMAGE of include/pgbart_image.h: the state of the chain at an idle point (between
 * asteps) in a layout that belongs to no backend -- sum_trees, the m accepted trees with every row's leaf label,
 * the running-sd accumulators, the split weights, leaf_sd[K], iter, lower, the likelihood parameters, the counters.
 * Any backend that implements this ABI continues a chain any other wrote, bit for bit: pgb_create with the
 * SAME settings, pgb_set_data / pgb_set_response (/ pgb_set_offset) with the same data, then pgb_checkpoint_load
 * (the random numbers are addressed by (seed, iter, ...), so the image carries no generator state beyond the
 * iteration counter).  An image carries a layout version; a record that is truncated, inconsistent or written for
 * other settings is refused with PGB_E_INVALID and a message.  Loading also clears a poisoned handle (see the
 * callback); saving refuses one.  The size depends on the trees: ask pgb_checkpoint_size right before saving. */
int pgb_checkpoint_size(pgb_handle* h, int64_t* bytes_out);
int pgb_checkpoint_save(pgb_handle* h, void* host_buf, int64_t bytes);
int pgb_checkpoint_load(pgb_handle* h, const void* host_buf, int64_t bytes);

/* The revision of THIS header a library was built against (PGB_ABI_VERSION): bumped whenever a signature, a
 * struct of this file or the chain image changes, so that a binding refuses a stale library instead of calling it
 * with shifted arguments.  Replaces nothing in the reference (PyO3 checks its own module at import). */
#define PGB_ABI_VERSION 6
int32_t pgb_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PGBART_H */
