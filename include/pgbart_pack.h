/*
 * pgbart_pack.h -- the packed tree record of pgb_export_trees_packed (include/pgbart.h), and its
 * backend-independent implementation on top of pgb_export_trees.  Included by both backends.
 *
 * Record (little endian, every array contiguous):
 *   int32  n_trees, n_outputs (K), total_nodes (N), flags (bit 0: the linear-response arrays are present;
 *          bit 1: the per-node split rules are present -- always set by this version, records written before
 *          round 5 lack the array and every split of theirs reads as continuous)
 *   int32  tree_id[n_trees], node_off[n_trees + 1], var[N], left[N], right[N], rule[N], (svar[N])
 *   -- zero padding to a multiple of 8 bytes --
 *   double split[N];  int64 count[N];  double value[N * K];  (double slope[N * K];  double xbar[N])
 * Counterpart of the reference's TreeArrays crossing the PyO3 boundary in one object
 * (pymc_bart/pymc_bart.py:2); the per-draw batches of utils.py:124-127 travel in this form.
 */
#ifndef PGBART_PACK_H
#define PGBART_PACK_H

#include <stdint.h>
#include <string.h>

#include "pgbart.h"

#define PGB_PACK_LINEAR 1
#define PGB_PACK_RULES 2

static inline int64_t pgb_packed_bytes(int32_t nt, int32_t N, int32_t K, int lin) {
  int64_t ints = 4 + (int64_t)nt + (nt + 1) + 4 * (int64_t)N + (lin ? N : 0);
  int64_t head = (ints * 4 + 7) & ~(int64_t)7;
  return head + 8 * ((int64_t)N * 2 + (int64_t)N * K + (lin ? (int64_t)N * K + N : 0));
}

/* Point the arrays of `out` into a record buffer laid out for (nt, N, K, lin) and write its header. */
static inline void pgb_packed_bind(void* buf, int32_t nt, int32_t N, int32_t K, int lin, pgb_tree_arrays* out) {
  int32_t* ip = (int32_t*)buf;
  ip[0] = nt; ip[1] = K; ip[2] = N; ip[3] = PGB_PACK_RULES | (lin ? PGB_PACK_LINEAR : 0);
  int32_t* q = ip + 4;
  out->n_trees = nt; out->n_outputs = K; out->total_nodes = N;
  out->tree_id = q; q += nt;
  out->node_off = q; q += nt + 1;
  out->var = q; q += N;
  out->left = q; q += N;
  out->right = q; q += N;
  out->rule = q; q += N;
  out->svar = lin ? q : (int32_t*)0; q += lin ? N : 0;
  int64_t ints = (int64_t)(q - ip);
  if (ints & 1) *q = 0; /* the padding word */
  double* d = (double*)((char*)buf + ((ints * 4 + 7) & ~(int64_t)7));
  out->split = d; d += N;
  out->count = (int64_t*)d; d += N;
  out->value = d; d += (int64_t)N * K;
  out->slope = lin ? d : (double*)0; d += lin ? (int64_t)N * K : 0;
  out->xbar = lin ? d : (double*)0;
}

/* pgb_export_trees_packed for any backend that has pgb_export_trees; `lin`: the sampler has linear leaves. */
static inline int pgb_export_trees_packed_via(pgb_handle* h, int32_t which, void* host_buf, int64_t cap_bytes,
                                              int64_t* bytes_out, int lin) {
  pgb_tree_arrays sz;
  memset(&sz, 0, sizeof sz);
  int rc = pgb_export_trees(h, which, &sz); /* size query: array pointers NULL */
  if (rc != PGB_OK) return rc;
  const int64_t need = pgb_packed_bytes(sz.n_trees, sz.total_nodes, sz.n_outputs, lin);
  if (bytes_out) *bytes_out = need;
  if (!host_buf || cap_bytes < need) return PGB_E_NOMEM; /* *bytes_out says how much is needed */
  pgb_tree_arrays out;
  memset(&out, 0, sizeof out);
  pgb_packed_bind(host_buf, sz.n_trees, sz.total_nodes, sz.n_outputs, lin, &out);
  return pgb_export_trees(h, which, &out);
}

#endif /* PGBART_PACK_H */
