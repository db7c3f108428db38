// k_export.h -- part of pgbart_hip.hip (not a standalone header): k_export_step: the host-facing result of one astep.
// ------------------------------------------------------------------ k_export_step
// What PGBART.astep returns to PyMC besides sum_trees: the trees this step
// re-sampled as compact SoA arrays (the layout of pgb_tree_arrays), the variable-inclusion counts,
// the work counters and the idle-point control words.  The kernel writes them straight into a
// MAPPED PINNED host block (posted PCIe writes of a few KB), so that the whole return path of an
// astep is one kernel + one DMA of sum_trees + one stream synchronisation -- no small blocking
// copies.  It also gathers sum_trees [K][n_pad] into a dense [K][n] staging buffer in HBM, which
// the host then moves with a single hipMemcpyAsync.
struct StepOutHdr {  // head of the mapped block
  int32_t n_trees, total_nodes, first, K;
  int32_t phase, st_cur, alpha_cur, lin;
  unsigned long long counters[8];
  long long steps_done, pad;
};
struct StepOutLayout {  // byte offsets inside the mapped block (host and device agree on these)
  long long vi, node_off, var, left, right, svar, split, count, xbar, value, slope, bytes;
  int32_t cap_trees, cap_nodes;
};
__host__ __device__ inline StepOutLayout stepout_layout(int p, int cap_trees, int K, bool lin) {
  StepOutLayout L;
  const long long cap = (long long)cap_trees * MAXN;
  long long o = (long long)sizeof(StepOutHdr);
  auto take = [&](long long bytes) { const long long at = o; o += (bytes + 63) & ~63ll; return at; };
  L.vi = take((long long)p * 4);
  L.node_off = take((long long)(cap_trees + 1) * 4);
  L.var = take(cap * 4);
  L.left = take(cap * 4);
  L.right = take(cap * 4);
  L.svar = take(lin ? cap * 4 : 0);
  L.split = take(cap * 8);
  L.count = take(cap * 8);
  L.xbar = take(lin ? cap * 8 : 0);
  L.value = take(cap * 8 * K);
  L.slope = take(lin ? cap * 8 * K : 0);
  L.bytes = o;
  L.cap_trees = cap_trees;
  L.cap_nodes = (int32_t)cap;
  return L;
}

// grid: max(n_trees, blocks that cover the sum_trees gather); block b < n_trees exports tree first + b
__global__ __launch_bounds__(BT) void k_export_step(const Dev* __restrict__ Sp, int par, int first, int n_trees,
                                                    unsigned char* __restrict__ blob, StepOutLayout L,
                                                    double* __restrict__ st_dense /* [K][n] or null */) {
  const DevG& S = *reinterpret_cast<const DevG*>(Sp);
  const int tid = threadIdx.x, b = blockIdx.x;
  const Ctrl c = S.ctrl[par];
  const int K = S.K, KX = S.K - 1;
  const bool lin = S.response != PGB_RESPONSE_CONSTANT;
  if (st_dense != nullptr) {  // sum_trees of the idle point, without the row padding
    const double* src = S.st + (size_t)c.st_cur * K * S.n_pad;
    const long long tot = (long long)K * S.n;
    for (long long e = (long long)b * BT + tid; e < tot; e += (long long)gridDim.x * BT) {
      const long long k = e / S.n, i = e - k * S.n;
      st_dense[e] = src[(size_t)k * S.n_pad + i];
    }
  }
  if (b == 0) {
    StepOutHdr* H = (StepOutHdr*)blob;
    int32_t* vi = (int32_t*)(blob + L.vi);
    for (int j = tid; j < S.p; j += BT) vi[j] = S.vi[j];
    if (tid < 8) H->counters[tid] = S.counters[tid];
    if (tid == 0) {  // (node offsets and the node total: written by the blocks that export the trees)
      H->n_trees = n_trees;
      if (n_trees == 0) {
        H->total_nodes = 0;
        *(int32_t*)(blob + L.node_off) = 0;
      }
      H->first = first;
      H->K = K;
      H->phase = c.phase;
      H->st_cur = c.st_cur;
      H->alpha_cur = c.alpha_cur;
      H->lin = lin ? 1 : 0;
      H->steps_done = c.steps_done;
    }
  }
  if (b >= n_trees) return;
  // every block computes its own offset (no dependency between blocks): the node counts of the trees before it
  // are loaded by as many threads at once and summed -- a serial walk over 20 trees was 20 dependent round trips
  // to memory, most of this kernel's 6.7 us on the critical path of every astep
  __shared__ long long s_off[4];
  long long part[1] = {0};
  for (int t = tid; t < b; t += BT) part[0] += S.trees[first + t].n_nodes;
  block_sum<1>(part, s_off);
  if (tid == 0) s_off[0] = part[0];
  __syncthreads();
  const int base = (int)s_off[0];
  const DTree* T = &S.trees[first + b];
  const int nn = T->n_nodes;
  if (tid == 0) {
    int32_t* off = (int32_t*)(blob + L.node_off);
    off[b] = base;
    if (b == n_trees - 1) {
      off[n_trees] = base + nn;
      ((StepOutHdr*)blob)->total_nodes = base + nn;
    }
  }
  int32_t* o_var = (int32_t*)(blob + L.var) + base;
  int32_t* o_left = (int32_t*)(blob + L.left) + base;
  int32_t* o_right = (int32_t*)(blob + L.right) + base;
  double* o_split = (double*)(blob + L.split) + base;
  long long* o_count = (long long*)(blob + L.count) + base;
  double* o_value = (double*)(blob + L.value) + (size_t)base * K;
  for (int k = tid; k < nn; k += BT) {
    const DNode z = T->nd[k];
    const bool leaf = z.var < 0;
    o_var[k] = z.var;
    o_split[k] = leaf ? 0.0 : z.split;
    o_left[k] = leaf ? -1 : (int32_t)z.left;
    o_right[k] = leaf ? -1 : (int32_t)z.right;
    o_count[k] = z.cnt;
    o_value[(size_t)k * K] = leaf ? z.value : 0.0;
    for (int o = 1; o < K; ++o)
      o_value[(size_t)k * K + o] = leaf ? S.tvx[((size_t)(first + b) * MAXN + k) * KX + o - 1] : 0.0;
    if (lin) {
      const LinP lp = S.tlin[(size_t)(first + b) * MAXN + k];
      const bool islin = leaf && lp.svar >= 0;
      ((int32_t*)(blob + L.svar))[base + k] = islin ? (int32_t)lp.svar : -1;
      ((double*)(blob + L.xbar))[base + k] = islin ? lp.xbar : 0.0;
      double* o_slope = (double*)(blob + L.slope) + (size_t)(base + k) * K;
      o_slope[0] = islin ? lp.slope : 0.0;
      for (int o = 1; o < K; ++o)
        o_slope[o] = islin ? S.tsx[((size_t)(first + b) * MAXN + k) * KX + o - 1] : 0.0;
    }
  }
}
