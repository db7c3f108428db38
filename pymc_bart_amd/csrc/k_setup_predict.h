// k_setup_predict.h -- part of pgbart_hip.hip (not a standalone header): set-up kernels and the prediction kernel.
// ------------------------------------------------------------------ setup kernels
// X row-major [n][ldx] -> XT column-major [p][n_pad]; LDS-tiled 32x32 transpose so that both
// the read and the write are coalesced.  Also flags columns that contain NaN.
// 16-bit order keys of the column-major design matrix (see k_rows<..., F32>), one column at a time:
//   k_key_stage   the column's values rounded to float32 (a missing value: the canonical positive NaN, which a
//                 radix sort of the bit patterns puts behind +inf), the number of missing values counted;
//   (hipcub radix sort of the staged values)
//   k_key_bounds  boundary i (1 .. PGB_KEY_BOUNDS) = the value at rank i m / (PGB_KEY_BOUNDS + 1) of the m
//                 non-missing values: equi-depth bins, so that a split value shares its key with m / 65 535 rows;
//   k_key_assign  key(x) = number of boundaries <= float32(x) (binary search; non-decreasing in x), 0xFFFF if missing.
#define PGB_KEY_BOUNDS 65534
__global__ __launch_bounds__(BT) void k_key_stage(const double* __restrict__ col, float* __restrict__ out, long long n,
                                                  unsigned* __restrict__ n_missing) {
  unsigned miss = 0;
  for (long long i = (long long)blockIdx.x * BT + threadIdx.x; i < n; i += (long long)gridDim.x * BT) {
    const double x = col[i];
    const bool m = x != x;
    out[i] = m ? __uint_as_float(0x7FC00000u) : (float)x;
    miss += m ? 1u : 0u;
  }
  if (miss) atomicAdd(n_missing, miss);
}
__global__ __launch_bounds__(BT) void k_key_bounds(const float* __restrict__ sorted, long long n,
                                                   const unsigned* __restrict__ n_missing, float* __restrict__ bnd) {
  const long long m = n - (long long)*n_missing;
  for (int i = blockIdx.x * BT + threadIdx.x; i < PGB_KEY_BOUNDS; i += gridDim.x * BT)
    bnd[i] = m > 0 ? sorted[((long long)(i + 1) * m) / (PGB_KEY_BOUNDS + 1)] : __uint_as_float(0x7F800000u);
}
__global__ __launch_bounds__(BT) void k_key_assign(const double* __restrict__ col, const float* __restrict__ bnd,
                                                   uint16_t* __restrict__ key, long long n_pad) {
  for (long long i = (long long)blockIdx.x * BT + threadIdx.x; i < n_pad; i += (long long)gridDim.x * BT) {
    const double xd = col[i];
    const float x = (float)xd;
    int lo = 0, hi = PGB_KEY_BOUNDS;  // first boundary > x
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (bnd[mid] <= x) lo = mid + 1;
      else hi = mid;
    }
    key[i] = xd != xd ? (uint16_t)0xFFFFu : (uint16_t)lo;
  }
}

__global__ __launch_bounds__(BT) void k_transpose(const double* __restrict__ X, long long ldx,
                                                  double* __restrict__ XT, long long n,
                                                  long long n_pad, int p, int32_t* col_nan) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const long long r0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  for (int k = ty; k < 32; k += 8) {
    long long r = r0 + k;
    int c = c0 + tx;
    tile[k][tx] = (r < n && c < p) ? X[r * ldx + c] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    int c = c0 + k;
    long long r = r0 + tx;
    if (c < p && r < n_pad) {
      double x = tile[tx][k];
      XT[(size_t)c * n_pad + r] = x;
      if (x != x) col_nan[c] = 1;
    }
  }
}

__global__ void k_init_linp(LinP* p, long long n) {  // constant leaves everywhere
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = LinP{0.0, 0.0, -1};
}
// per-column max |x| (NaN ignored) of the column-major copy: one workgroup per column
__global__ __launch_bounds__(BT) void k_colmax(const double* __restrict__ XT, long long n, long long n_pad,
                                               double* __restrict__ amax) {
  __shared__ double sm[BT];
  const double* c = XT + (size_t)blockIdx.x * n_pad;
  double a = 0.0;
  for (long long i = threadIdx.x; i < n; i += BT) {
    double v = c[i];
    v = v < 0.0 ? -v : v;
    if (v > a) a = v;
  }
  sm[threadIdx.x] = a;
  __syncthreads();
  for (int o = BT / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o && sm[threadIdx.x + o] > sm[threadIdx.x]) sm[threadIdx.x] = sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) amax[blockIdx.x] = sm[0];
}
// SubsetSplit columns hold integer category codes 0 .. PGB_SUBSET_BITS - 1 (NaN = missing); anything else would be
// clamped by pgb_subset_code, so pgb_set_data refuses it: one workgroup per column, *bad = 1 + the first such column
__global__ __launch_bounds__(BT) void k_subset_check(const double* __restrict__ XT, long long n, long long n_pad,
                                                     const int32_t* __restrict__ rules, int* __restrict__ bad) {
  if (rules[blockIdx.x] != PGB_RULE_SUBSET) return;
  const double* c = XT + (size_t)blockIdx.x * n_pad;
  bool b = false;
  for (long long i = threadIdx.x; i < n; i += BT) {
    const double v = c[i];
    if (v == v && !(v >= 0.0 && v <= (double)(PGB_SUBSET_BITS - 1) && v == (double)(int)v)) b = true;
  }
  if (b) atomicMax(bad, (int)blockIdx.x + 1);
}
// any non-finite value, or one beyond +-limit, in `rows` rows of n values, `stride` apart?  -> *flag = 1 (the
// host's mapped word).  The linear predictor of a row must be finite and of bounded size: the likelihood tables are
// addressed by its bits (pgb_lphi_t, pgb_exp_t; PGB_MAX_OFFSET).
__global__ __launch_bounds__(BT) void k_nonfinite(const double* __restrict__ a, long long n, long long stride, int rows,
                                                  double limit, unsigned long long* __restrict__ flag) {
  bool b = false;
  for (int r = 0; r < rows; ++r)
    for (long long i = (long long)blockIdx.x * BT + threadIdx.x; i < n; i += (long long)gridDim.x * BT) {
      const double v = a[(size_t)r * stride + i];
      if (!(v - v == 0.0) || !(__builtin_fabs(v) <= limit)) b = true;
    }
  if (__ballot(b) && (threadIdx.x & 63) == 0) *flag = 1ull;
}
__global__ void k_fill_f64(double* a, long long n, double v) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}

__global__ void k_init_tree_lid(uint8_t* a, long long n, long long n_pad, int m) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pad * m) a[i] = (i % n_pad) < n ? 0 : PGB_ORPHAN;
}

__global__ void k_init_trees(DTree* trees, int m, long long n, double init_leaf) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m) return;
  DTree* T = &trees[t];
  T->n_nodes = 1;
  T->n_leaves = 1;
  DNode z;
  memset(&z, 0, sizeof z);
  z.var = -1;
  z.cc_row = -1;
  z.cnt = (int32_t)n;
  z.value = init_leaf;
  T->nd[0] = z;
}

// integer split weights from the user's prior + their prefix sums (numeric contract:
// pgb_alpha_init / pgb_sample_var)
__global__ void k_init_alpha(const double* prior, double max_prior, long long* alpha, long long* cdfS, int p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    long long cs = 0;
    for (int j = 0; j < p; ++j) {
      const long long a = pgb_alpha_init(prior[j], max_prior);
      alpha[j] = a;
      cs += a;
      cdfS[j] = cs;
    }
  }
}

// ------------------------------------------------------------------ prediction
// out[d][k][row] = sum over the m trees of forest d of the leaf value reached by X[row,:]
// (PosteriorSampler.sample_posterior, utils.py:66-69); excluded / NaN splits average both
// subtrees by their training counts (CHANGELOG.md:410-411).  One thread per (row, forest).
//
// A traversal is a chain of dependent loads, so a node is ONE 32-byte record (packed on the host
// per call: split variable, children as pool-wide indices, split rule and "excluded" folded into
// flags, the split value and the training count) instead of a lookup in five arrays plus the rule
// and exclusion tables: one memory round trip per level for the node and one for the row's value.
struct PNode {
  int32_t var;          // split variable, -1: leaf
  int32_t left, right;  // pool-wide node indices
  int32_t flags;        // bit 0: the split variable is excluded; bits 1..2: split rule
  double split;
  double cnt;           // training rows in the node (exact in a double)
};
// The same tree for the walk that needs no marginalisation (16 bytes, children tree-local): leaves
// point at themselves with split = +inf, so a walk is exactly `depth` steps for every row -- no
// leaf test, no divergence -- and ends on its leaf.
struct FNode {
  double split;
  int32_t var;
  uint8_t left, right;
  uint16_t pad;
};
struct PredTrees {
  const PNode* node;      // [total_nodes]
  const FNode* fnode;     // [total_nodes]
  const int2* root;       // [n_trees] {pool-wide index of the root, depth | 0x100 when the tree needs the general walk}
  const double* value;    // [total_nodes][K]
  // linear leaves (svar == nullptr: none); svar is -1 for constant leaves AND for excluded regressors
  const double* slope;    // [total_nodes][K]
  const double* xbar;     // [total_nodes]
  const int32_t* svar;    // [total_nodes]
};

// The row's values: the threads of a wave own consecutive rows of a row-major matrix, so a read of
// "my row, my split variable" touches 64 different cache lines per instruction -- that, not the node
// chain, bounded the first version (44 G tree-traversals/s at cfg2).  LDSX: the wave first copies its
// 64 rows (coalesced) into LDS, transposed [column][lane] with a pad of one; the traversal reads
// from there.  One wave per workgroup; the workgroup loops over a share of the forests so that the
// staged rows serve several.  (p <= PRED_LDS_MAXP: 64 KB of LDS; wider matrices use global reads.)
#define PRED_BT 64
#ifndef PRED_WALKS
#define PRED_WALKS 4 /* interleaved fixed-length walks per lane */
#endif
#ifndef PRED_LDS_MAXP
#define PRED_LDS_MAXP 126
#endif
// CONT: every column follows the ContinuousSplit rule (the usual case): one compare per level.
template <bool LDSX, bool CONT>
__global__ __launch_bounds__(PRED_BT) void k_predict(PredTrees T, const int32_t* __restrict__ forest_idx,
                                                     int n_forests, int m, int K, int p,
                                                     const double* __restrict__ X, long long n_rows,
                                                     long long ldx, double* __restrict__ out) {
  extern __shared__ double s_x[];  // LDSX: [p][65]
  const int lane = threadIdx.x;
  const long long row0 = (long long)blockIdx.x * PRED_BT;
  const long long row = row0 + lane;
  if constexpr (LDSX) {
    const long long rows_here = n_rows - row0 < PRED_BT ? n_rows - row0 : PRED_BT;
    if (ldx == p) {  // the block of rows is contiguous: fully coalesced copy
      const double* __restrict__ src = X + row0 * ldx;
      const int tot = (int)rows_here * p;
      for (int i = lane; i < tot; i += PRED_BT) s_x[(i % p) * 65 + i / p] = src[i];
    } else {
      for (int r = 0; r < (int)rows_here; ++r)
        for (int j = lane; j < p; j += PRED_BT) s_x[j * 65 + r] = X[(row0 + r) * ldx + j];
    }
    __syncthreads();
  }
  if (row >= n_rows) return;
  const double* __restrict__ x = X + row * ldx;
  auto xval = [&](int j) -> double {
    if constexpr (LDSX) return s_x[j * 65 + lane];
    else return x[j];
  };
  // Rows without a missing value take the fixed-length walk through every tree that does not split
  // on an excluded variable (decided per wave / per tree, so that the walk stays uniform).
  bool clean = CONT;
  if (CONT) {
    bool nan = false;
    for (int j = 0; j < p; ++j) {
      const double v = xval(j);
      nan = nan || v != v;
    }
    clean = __ballot(nan) == 0ull;
  }
  int stk_node[PGB_MAX_DEPTH + 2];
  double stk_w[PGB_MAX_DEPTH + 2];
  for (int d = blockIdx.y; d < n_forests; d += gridDim.y) {
    double acc[PGB_MAX_OUTPUTS];
    for (int o = 0; o < K; ++o) acc[o] = 0.0;
    // Trees are taken PRED_WALKS at a time when all of them qualify for the fixed-length walk: that
    // many independent chains of (node, value) loads per lane; the roots of the next group are
    // requested before the current group is walked.  Otherwise one tree at a time (the explicit
    // stack -- private memory -- is touched only by walks that marginalise).
    const int32_t* __restrict__ fi = forest_idx + (size_t)d * m;
    int2 rnx[PRED_WALKS];
    bool have_nx = false;
    for (int t = 0; t < m; ++t) {
      if (clean && t + PRED_WALKS <= m) {
        int2 rw[PRED_WALKS];
        int any_general = 0;
#pragma unroll
        for (int w = 0; w < PRED_WALKS; ++w) {
          rw[w] = have_nx ? rnx[w] : T.root[fi[t + w]];
          any_general |= rw[w].y & 0x100;
        }
        have_nx = false;
        if (!any_general) {
          if (t + 2 * PRED_WALKS <= m) {
#pragma unroll
            for (int w = 0; w < PRED_WALKS; ++w) rnx[w] = T.root[fi[t + PRED_WALKS + w]];
            have_nx = true;
          }
          int steps = 0, gw[PRED_WALKS];
#pragma unroll
          for (int w = 0; w < PRED_WALKS; ++w) {
            steps = rw[w].y > steps ? rw[w].y : steps;  // a finished walk idles on its leaf
            gw[w] = 0;
          }
          for (int l = 0; l < steps; ++l) {
            uint4 q[PRED_WALKS];
#pragma unroll
            for (int w = 0; w < PRED_WALKS; ++w) q[w] = ((const uint4*)(T.fnode + rw[w].x))[gw[w]];
#pragma unroll
            for (int w = 0; w < PRED_WALKS; ++w)
              asm volatile("" : "+v"(q[w].x), "+v"(q[w].y), "+v"(q[w].z), "+v"(q[w].w));  // whole 16-byte loads
#pragma unroll
            for (int w = 0; w < PRED_WALKS; ++w) {
              const double xv = xval((int)q[w].z);
              gw[w] = xv <= __hiloint2double((int)q[w].y, (int)q[w].x) ? (int)(q[w].w & 255u) : (int)((q[w].w >> 8) & 255u);
            }
          }
#pragma unroll
          for (int w = 0; w < PRED_WALKS; ++w) {  // tree t first, then t + 1, ...: the order of the plain loop
            const int hg = rw[w].x + gw[w];
            int js = -1;
            if (T.svar != nullptr) js = T.svar[hg];
            for (int o = 0; o < K; ++o) {
              double vo = T.value[(size_t)hg * K + o];
              if (js >= 0) vo = pgb_leaf_pred(vo, T.slope[(size_t)hg * K + o], T.xbar[hg], xval(js));
              acc[o] += vo;  // (the general walk adds 1.0 * vo: the same bits)
            }
          }
          t += PRED_WALKS - 1;
          continue;
        }
      }
      const int2 rt = T.root[fi[t]];
      if (clean && !(rt.y & 0x100)) {
        const uint4* __restrict__ fn = (const uint4*)(T.fnode + rt.x);
        int gl = 0;
        for (int l = 0; l < rt.y; ++l) {
          uint4 q = fn[gl];
          asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w));  // one 16-byte load, not three sunk ones
          const double xv = xval((int)q.z);
          gl = xv <= __hiloint2double((int)q.y, (int)q.x) ? (int)(q.w & 255u) : (int)((q.w >> 8) & 255u);
        }
        const int gg = rt.x + gl;
        int js = -1;
        if (T.svar != nullptr) js = T.svar[gg];
        for (int o = 0; o < K; ++o) {
          double vo = T.value[(size_t)gg * K + o];
          if (js >= 0) vo = pgb_leaf_pred(vo, T.slope[(size_t)gg * K + o], T.xbar[gg], xval(js));
          acc[o] += vo;
        }
        continue;
      }
      int g = rt.x;
      double w = 1.0;
      int sp = 0;
      for (;;) {
        // the record as two 16-byte words, requested together (a struct copy is split into per-field
        // loads that the compiler sinks to their uses: three dependent round trips per level)
        const uint4* __restrict__ np = (const uint4*)(T.node + g);
        uint4 n0 = np[0], n1 = np[1];
        // (an empty asm that "uses" all eight words: without it the loads are narrowed and sunk again)
        asm volatile("" : "+v"(n0.x), "+v"(n0.y), "+v"(n0.z), "+v"(n0.w), "+v"(n1.x), "+v"(n1.y), "+v"(n1.z), "+v"(n1.w));
        PNode nd;
        nd.var = (int32_t)n0.x; nd.left = (int32_t)n0.y; nd.right = (int32_t)n0.z; nd.flags = (int32_t)n0.w;
        nd.split = __hiloint2double((int)n1.y, (int)n1.x);
        bool done = false;  // this branch of the walk has ended
        if (nd.var < 0) {
          int js = -1;  // linear leaf; a missing / excluded regressor: the mean
          double xs = 0.0;
          if (T.svar != nullptr) {
            js = T.svar[g];
            if (js >= 0) {
              xs = xval(js);
              if (xs != xs) js = -1;
            }
          }
          for (int o = 0; o < K; ++o) {
            double vo = T.value[(size_t)g * K + o];
            if (js >= 0) vo = pgb_leaf_pred(vo, T.slope[(size_t)g * K + o], T.xbar[g], xs);
            acc[o] += w * vo;
          }
          done = true;
        } else {
          const double xv = xval(nd.var);
          if ((nd.flags & 1) || xv != xv) {
            const double cl = T.node[nd.left].cnt, cr = T.node[nd.right].cnt;
            const double tot = cl + cr;
            if (!(tot > 0.0)) {
              done = true;
            } else {  // depth-first, left first (same summation order as the oracle's recursion)
              stk_node[sp] = nd.right;
              stk_w[sp] = w * (cr / tot);
              ++sp;
              g = nd.left;
              w = w * (cl / tot);
            }
          } else {
            const bool gl = CONT ? xv <= nd.split : pgb_go_left(nd.flags >> 1, xv, nd.split) != 0;
            g = gl ? nd.left : nd.right;
          }
        }
        if (done) {
          if (sp == 0) break;
          --sp;
          g = stk_node[sp];
          w = stk_w[sp];
        }
      }
    }
    for (int o = 0; o < K; ++o) out[((size_t)d * K + o) * n_rows + row] = acc[o];
  }
}
