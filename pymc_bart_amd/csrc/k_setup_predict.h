// k_setup_predict.h -- part of pgbart_hip.hip (not a standalone header): set-up kernels and the prediction kernel.
// ------------------------------------------------------------------ setup kernels
// X row-major [n][ldx] -> XT column-major [p][n_pad]; LDS-tiled 32x32 transpose so that both
// the read and the write are coalesced.  Also flags columns that contain NaN.
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
struct PredTrees {
  const int32_t* node_off;
  const int32_t* var;
  const double* split;
  const int32_t* left;
  const int32_t* right;
  const long long* count;
  const double* value;
  // linear leaves (svar == nullptr: none)
  const double* slope;
  const double* xbar;
  const int32_t* svar;
};

__global__ __launch_bounds__(BT) void k_predict(PredTrees T, const int32_t* forest_idx, int n_forests,
                                                int m, int K, const double* __restrict__ X,
                                                long long n_rows, int p, long long ldx,
                                                const int32_t* rules, const uint8_t* excl,
                                                double* out) {
  const long long row = (long long)blockIdx.x * BT + threadIdx.x;
  const int d = blockIdx.y;
  if (row >= n_rows) return;
  const double* x = X + row * ldx;
  double acc[PGB_MAX_OUTPUTS];
  for (int o = 0; o < K; ++o) acc[o] = 0.0;
  int stk_node[PGB_MAX_DEPTH + 2];
  double stk_w[PGB_MAX_DEPTH + 2];
  for (int t = 0; t < m; ++t) {
    const int base = T.node_off[forest_idx[(size_t)d * m + t]];
    int sp = 0;
    stk_node[0] = 0;
    stk_w[0] = 1.0;
    sp = 1;
    while (sp > 0) {
      --sp;
      int k = stk_node[sp];
      double w = stk_w[sp];
      for (;;) {
        const int g = base + k;
        const int j = T.var[g];
        if (j < 0) {
          int js = -1;  // linear leaf; a missing / excluded regressor: the mean
          if (T.svar != nullptr) {
            js = T.svar[g];
            if (js >= 0 && (excl[js] || x[js] != x[js])) js = -1;
          }
          for (int o = 0; o < K; ++o) {
            double vo = T.value[(size_t)g * K + o];
            if (js >= 0) vo = pgb_leaf_pred(vo, T.slope[(size_t)g * K + o], T.xbar[g], x[js]);
            acc[o] += w * vo;
          }
          break;
        }
        const double xv = x[j];
        if (excl[j] || xv != xv) {
          const int l = T.left[g], r = T.right[g];
          const double cl = (double)T.count[base + l], cr = (double)T.count[base + r];
          const double tot = cl + cr;
          if (!(tot > 0.0)) break;
          // depth-first, left first (same summation order as the oracle's recursion)
          stk_node[sp] = r;
          stk_w[sp] = w * (cr / tot);
          ++sp;
          k = l;
          w = w * (cl / tot);
          continue;
        }
        const bool gl = pgb_go_left(rules[j], xv, T.split[g]) != 0;
        k = gl ? T.left[g] : T.right[g];
      }
    }
  }
  for (int o = 0; o < K; ++o) out[((size_t)d * K + o) * n_rows + row] = acc[o];
}

