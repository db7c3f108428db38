/*
 * pgbart_image.h -- the CHAIN IMAGE of pgb_checkpoint_save / pgb_checkpoint_load (include/pgbart.h): one chain at an
 * idle point (between asteps), in a layout that belongs to no backend.  Every backend that implements the ABI writes
 * and reads exactly this record, so a chain started on one continues, bit for bit, on another: the gfx950 library
 * after its burn-in -> the CPU restatement under oracle/ (the steady-state parity tests), the 64-particle build <->
 * the 128-particle build, a PyMC worker process -> the parent.
 *
 * What a chain IS between two asteps ([U] the attributes PGBART keeps on `self`, SURVEY.md Appendix A "State"):
 *   sum_trees (K, n), the m accepted trees with the leaf every training row sits in, the running-sd accumulators of
 *   the tuning phase, the split weights (alpha_vec) and the prefix sums the split-variable sampler currently uses,
 *   leaf_sd[K], the tree-update counter `iter` (which also addresses the random numbers: pgbart_spec.h), the batch
 *   cursor `lower`, the likelihood parameters last handed in, and the work counters.  Nothing else: particles,
 *   residuals, label generations, job records ... live for one tree update.
 * The reference has no counterpart file: upstream pickles the step method (SURVEY 8b "must be picklable") and its
 * per-chain tree history travels as (baseline_forest, batches) (utils.py:124-127) -- this record is what the
 * step method's pickle carries here.
 *
 * Record (little endian; every section starts on a multiple of 8 bytes, zero padded):
 *   pgb_image_header
 *   double  sum_trees[K][n], rs_mean[K][n], rs_m2[K][n]
 *   int64   alpha[p]            integer split weights (pgb_alpha_init + counts * pgb_alpha_unit)
 *   int64   cdf[p]              the prefix sums the sampler draws from (rebuilt from alpha at the times upstream
 *                               rebuilds its SampleSplittingVariable: they may lag alpha)
 *   int32   node_off[m + 1]     tree t owns nodes [node_off[t], node_off[t+1]); N = node_off[m]
 *   int32   var[N], left[N], right[N], depth[N], label[N], svar[N]
 *                               var -1 / left = right = -1: a leaf; children indices are tree-local and larger than
 *                               their parent's; label: the leaf label of the node's rows in `lid` (a split node keeps
 *                               the label its left-most leaf inherited; only leaves' labels are read); svar: the
 *                               regressor of a linear leaf, -1 for a constant one
 *   int64   count[N];  double split[N], xbar[N], value[N][K], slope[N][K]
 *   uint8   lid[m][n]           leaf label of every training row in every tree (PGB_ORPHAN = 255: dropped by a
 *                               missing split value)
 */
#ifndef PGBART_IMAGE_H
#define PGBART_IMAGE_H

#include <stdint.h>
#include <string.h>

#include "pgbart.h"
#include "pgbart_spec.h"

#define PGB_IMAGE_VERSION 1

typedef struct {
  char magic[8];        /* "PGBIMAGE"                                                        */
  int32_t version;      /* PGB_IMAGE_VERSION                                                 */
  int32_t header_bytes; /* sizeof(pgb_image_header) of the writer                            */
  int64_t total_bytes;  /* header + sections                                                 */
  pgb_settings s;       /* must equal the loading sampler's settings                         */
  int64_t iter;         /* tree updates so far                                               */
  int64_t rs_count;     /* updates the running sd has seen                                   */
  int32_t lower;        /* batch cursor: first tree of the next astep                        */
  int32_t last_lower, last_n; /* the batch of the last astep (pgb_export_trees(h, 0, ...))   */
  int32_t total_nodes;  /* N                                                                 */
  double leaf_sd[PGB_MAX_OUTPUTS];
  double lik_param[2];  /* as pgb_set_likelihood left them (NORMAL: {1 / sigma^2, -})         */
  pgb_counters ctr;     /* (`slots` is backend-specific and carried as found)                */
  char writer[16];      /* pgb_backend_name() of the writer -- information, never checked    */
} pgb_image_header;

typedef struct {
  double *sum_trees, *rs_mean, *rs_m2;
  int64_t *alpha, *cdf;
  int32_t *node_off, *var, *left, *right, *depth, *label, *svar;
  int64_t* count;
  double *split, *xbar, *value, *slope;
  uint8_t* lid;
} pgb_image_view;

static inline int64_t pgb_image_pad8(int64_t b) { return (b + 7) & ~(int64_t)7; }

static inline int64_t pgb_image_bytes(int64_t n, int32_t p, int32_t m, int32_t K, int32_t N) {
  int64_t b = pgb_image_pad8((int64_t)sizeof(pgb_image_header));
  b += 3 * 8 * (int64_t)K * n;
  b += 2 * 8 * (int64_t)p;
  b += pgb_image_pad8(4 * ((int64_t)m + 1));
  b += pgb_image_pad8(4 * 6 * (int64_t)N);
  b += 8 * (3 * (int64_t)N + 2 * (int64_t)N * K);
  b += pgb_image_pad8((int64_t)m * n);
  return b;
}

/* Point `v` into a record buffer whose header says (n, p, m, K, N).  The int32 node arrays are laid out as ONE
 * padded block of 6 N words. */
static inline void pgb_image_bind(void* buf, const pgb_image_header* hd, pgb_image_view* v) {
  const int64_t n = hd->s.n, K = hd->s.n_outputs, N = hd->total_nodes;
  const int32_t p = hd->s.p, m = hd->s.m;
  char* q = (char*)buf + pgb_image_pad8((int64_t)sizeof(pgb_image_header));
  v->sum_trees = (double*)q; q += 8 * K * n;
  v->rs_mean = (double*)q; q += 8 * K * n;
  v->rs_m2 = (double*)q; q += 8 * K * n;
  v->alpha = (int64_t*)q; q += 8 * (int64_t)p;
  v->cdf = (int64_t*)q; q += 8 * (int64_t)p;
  v->node_off = (int32_t*)q; q += pgb_image_pad8(4 * ((int64_t)m + 1));
  v->var = (int32_t*)q;
  v->left = v->var + N;
  v->right = v->left + N;
  v->depth = v->right + N;
  v->label = v->depth + N;
  v->svar = v->label + N;
  q += pgb_image_pad8(4 * 6 * N);
  v->count = (int64_t*)q; q += 8 * N;
  v->split = (double*)q; q += 8 * N;
  v->xbar = (double*)q; q += 8 * N;
  v->value = (double*)q; q += 8 * N * K;
  v->slope = (double*)q; q += 8 * N * K;
  v->lid = (uint8_t*)q;
}

/* Start a record: zero the buffer's header and padding words, stamp it.  `bytes` = pgb_image_bytes(...). */
static inline void pgb_image_begin(void* buf, int64_t bytes, const pgb_settings* s, int32_t total_nodes,
                                   const char* writer, pgb_image_header* hd) {
  memset(hd, 0, sizeof *hd);
  memcpy(hd->magic, "PGBIMAGE", 8);
  hd->version = PGB_IMAGE_VERSION;
  hd->header_bytes = (int32_t)sizeof(pgb_image_header);
  hd->total_bytes = bytes;
  hd->s = *s;
  hd->total_nodes = total_nodes;
  strncpy(hd->writer, writer, sizeof hd->writer - 1);
  /* the padding after the odd-sized sections (never read, but an image is compared and hashed as bytes) */
  pgb_image_view v;
  pgb_image_bind(buf, hd, &v);
  memset((char*)buf, 0, (size_t)pgb_image_pad8((int64_t)sizeof(pgb_image_header)));
  memset(v.node_off, 0, (size_t)pgb_image_pad8(4 * ((int64_t)s->m + 1)));
  memset(v.var, 0, (size_t)pgb_image_pad8(4 * 6 * (int64_t)total_nodes));
  if (((int64_t)s->m * s->n) & 7) memset(v.lid + (((int64_t)s->m * s->n) & ~(int64_t)7), 0, 8);
}

/* Is `buf` an image this sampler can continue?  Returns NULL when it is, else what is wrong (a static string).
 * Checks everything a loader indexes by: a truncated or foreign record is an error, never an out-of-bounds walk. */
static inline const char* pgb_image_check(const void* buf, int64_t bytes, const pgb_settings* mine) {
  if (bytes < (int64_t)sizeof(pgb_image_header)) return "checkpoint truncated";
  pgb_image_header hd;
  memcpy(&hd, buf, sizeof hd);
  if (memcmp(hd.magic, "PGBIMAGE", 8) != 0) return "not a pgbart checkpoint";
  if (hd.version != PGB_IMAGE_VERSION || hd.header_bytes != (int32_t)sizeof(pgb_image_header))
    return "checkpoint layout version differs from this build's (written by another release)";
  if (memcmp(&hd.s, mine, sizeof(pgb_settings)) != 0) return "checkpoint settings differ from this sampler's settings";
  const int32_t m = hd.s.m, p = hd.s.p, K = hd.s.n_outputs, N = hd.total_nodes;
  if (N < m || (int64_t)N > (int64_t)m * PGB_MAX_NODES) return "checkpoint is inconsistent (node count)";
  if (hd.total_bytes != pgb_image_bytes(hd.s.n, p, m, K, N) || bytes < hd.total_bytes) return "checkpoint truncated";
  if (hd.lower < 0 || hd.lower >= m || hd.last_lower < 0 || hd.last_n < 0 || hd.last_lower + hd.last_n > m ||
      hd.iter < 0 || hd.rs_count < 0)
    return "checkpoint is inconsistent (cursor)";
  pgb_image_view v;
  pgb_image_bind((void*)buf, &hd, &v);
  if (v.node_off[0] != 0 || v.node_off[m] != N) return "checkpoint is inconsistent (node offsets)";
  for (int32_t t = 0; t < m; ++t) {
    const int32_t base = v.node_off[t], nn = v.node_off[t + 1] - base;
    if (nn < 1 || nn > PGB_MAX_NODES || base < 0 || base + nn > N) return "checkpoint is inconsistent (node offsets)";
    for (int32_t k = 0; k < nn; ++k) {
      const int32_t g = base + k;
      if (v.var[g] >= p) return "checkpoint is inconsistent (split column)";
      if (v.var[g] >= 0) {
        if (v.left[g] <= k || v.right[g] <= k || v.left[g] >= nn || v.right[g] >= nn)
          return "checkpoint is inconsistent (children)";
      } else if (v.label[g] < 0 || v.label[g] >= PGB_ORPHAN || v.svar[g] >= p) {
        return "checkpoint is inconsistent (leaf)";
      }
      if (v.depth[g] < 0 || v.depth[g] > PGB_MAX_NODES) return "checkpoint is inconsistent (depth)";
    }
  }
  return (const char*)0;
}

#endif /* PGBART_IMAGE_H */
