// pgb_leaf_values.h -- part of pgbart_hip.hip (not a standalone header): k_begin and the leaf-value routines shared by k_ctrl and k_loglik.
// ------------------------------------------------------------------ k_begin
__global__ void k_begin(const Dev* __restrict__ Sp, int par, int tune, int n_steps, double inv_sigma2, double lik_param2,
                        int set_sigma) {
  const DevG& S = *reinterpret_cast<const DevG*>(Sp);
  Ctrl* c = &S.ctrl[par];
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    c->tune = tune;
    c->steps_left = n_steps;
    if (set_sigma) {
      c->inv_sigma2 = inv_sigma2;
      c->lik_param2 = lik_param2;
    }
    int bs = tune ? S.batch_tune : S.batch_draw;
    int upper = c->lower + bs;
    if (upper > S.m) upper = S.m;
    c->batch_n = upper - c->lower;
    c->k = 0;
    c->phase = PH_BEGIN;
  }
  for (int j = threadIdx.x; j < S.p; j += blockDim.x) S.vi[j] = 0;
}


// Leaf values of the two children a PARTITION pass created for one particle ([U] draw_leaf_value),
// from the pass statistics.  Shared by k_ctrl (which stores them) and k_loglik (which needs them
// one launch earlier) so that both evaluate EXACTLY the same expressions.
struct ChildVals {
  int ok;  // 1: split, -1: failed one-hot split (the node stays a leaf and keeps its value)
  int cL, cR, cN;
  long long aL, aR;
  double vL, vR;
};
__device__ __forceinline__ ChildVals child_values(const DevG& S, int rule, int cnt, long long p_q_st,
                                                  double p_value, unsigned long long a_cnts, long long a_aL,
                                                  long long a_aN, double z0, double z1, double leaf_sd) {
  ChildVals c;
  c.cL = (int)(a_cnts & 0xFFFFFFFFull);
  c.cN = (int)(a_cnts >> 32);
  c.cR = cnt - c.cL - c.cN;
  c.aL = a_aL;
  c.aR = p_q_st - a_aL - a_aN;
  if (c.cR == 0 && pgb_empty_right_fails(rule, S.compat)) {
    c.ok = -1;
    c.vL = p_value;
    c.vR = 0.0;
  } else {
    c.ok = 1;
    c.vL = pgb_leaf_value(c.cL, c.aL, S.sc.inv_c1, S.mdouble, z0, leaf_sd);
    c.vR = pgb_leaf_value(c.cR, c.aR, S.sc.inv_c1, S.mdouble, z1, leaf_sd);
  }
  return c;
}


// ---- K-vector leaves: extension outputs 1..K-1 ------------------------------------------------
#define KXMAX (PGB_MAX_OUTPUTS - 1)
// leaf_sd of extension output k (0-based), with the pending update of a FINAL pass resolved the
// same way as for output 0
__device__ __forceinline__ double leaf_sd_x(const DevG& S, const Ctrl& c, int ctrl_par, int acc_par, int k) {
  const double cur = S.lsdx[ctrl_par * KXMAX + k];
  if (!(c.pend_leafsd && c.pend_iter > 2)) return cur;
  const int KX = S.K - 1;
  long long q = 0;
  for (int sl = 0; sl < IA_SLOTS; ++sl) q += S.iax[((size_t)acc_par * IA_SLOTS + sl) * 2 * KX + KX + k];
  return pgb_tuned_leaf_sd(cur, c.pend_iter, q, S.sc.inv_c1, S.n);
}
__device__ __forceinline__ long long root_A_x(const DevG& S, int acc_par, int k) {
  const int KX = S.K - 1;
  long long q = 0;
  for (int sl = 0; sl < IA_SLOTS; ++sl) q += S.iax[((size_t)acc_par * IA_SLOTS + sl) * 2 * KX + k];
  return q;
}
// children of one particle, extension output k: values and sums ([U] draw_leaf_value per output;
// one Box-Muller pair per output, RNG sub-index = output)
struct ChildX {
  double vL, vR;
  long long aL, aR;
  double sL, sR;  // linear response: slopes on the shared regressor (0 for a constant leaf)
};
// z0 / z1: the Box-Muller pair addressed by (iter, round, particle, LEAF, sub = k + 1), drawn one slot ahead by
// the control kernel that proposed the split (Dev::jzx) -- the consumers (the next control kernel and this
// slot's likelihood pass, in EVERY workgroup's prologue) used to redo Philox + log + sqrt + sincos per output.
__device__ __forceinline__ ChildX child_values_x(const DevG& S, int ok, int cL, int cR, long long aLk,
                                                 long long aNk, long long pq, double pv, double z0, double z1,
                                                 double lsd) {
  ChildX c;
  c.sL = c.sR = 0.0;
  c.aL = aLk;
  c.aR = pq - aLk - aNk;
  if (ok == 1) {
    c.vL = pgb_leaf_value(cL, c.aL, S.sc.inv_c1, S.mdouble, z0, lsd);
    c.vR = pgb_leaf_value(cR, c.aR, S.sc.inv_c1, S.mdouble, z1, lsd);
  } else {
    c.vL = pv;  // failed one-hot split: the leaf keeps its value
    c.vR = 0.0;
  }
  return c;
}

