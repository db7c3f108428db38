// k_loglik.h -- part of pgbart_hip.hip (not a standalone header): k_loglik: per-row log-likelihood of the rows a round re-labelled (non-Normal families).
// ------------------------------------------------------------------ k_loglik
// Bernoulli families only ([U] update_weight): after the PARTITION pass of a slot, the children's
// leaf values are known (child_values, the same routine k_ctrl uses one launch later); this pass
// evaluates the per-row log-likelihood of the rows of the leaf that was split -- left / right /
// dropped by a missing value -- and reduces it in fixed point.  Same work items as PARTITION.
// (a template: the single-output kernels keep 64 of these in LDS and carry none of the arrays)
struct LJobLin {  // linear response: the children's linear parts
  double slopeL, xbarL, slopeR, xbarR;
  int32_t svarL, svarR;
};
struct LJobNone {};
template <bool MK, bool LIN, int KT, bool CATF = false>
struct LJobT : std::conditional<LIN, LJobLin, LJobNone>::type {  // (constant leaves: none of the linear fields in LDS)
  long long src, xoff, dst;  // byte offsets: the particle's labels before the split, its split column, its new labels
  double v, vL, vR;
  int32_t p, rule, label, check_nan, ok, new_label;
  int32_t cL, cR;  // rows of the children (the extension outputs' lanes read them here, see the job list)
  // K-vector leaves: outputs 1..K-1 (sized by the instance: K - 1 when K is known at compile time)
  static constexpr int NX = !MK ? 1 : KT > 0 ? KT - 1 : KXMAX;
  double vLx[NX], vRx[NX];
  double sLx[MK && LIN ? NX : 0], sRx[MK && LIN ? NX : 0];  // ... slopes of outputs 1..K-1
  // softmax with constant leaves: the (particle, child) part of the factorised log-likelihood (pgb_cat_side),
  // [0] the left child, [1] the right one: d = v - v of output 0, w = exp(d); slowx[k - 1]: bit 0 / 1 = output k makes
  // the left / right child a slow one (|d| > PGB_CAT_DMAX: its rows take the unfactorised form)
  static constexpr int NK = !CATF ? 0 : KT > 0 ? KT : PGB_MAX_OUTPUTS;
  __attribute__((aligned(16))) double w2[CATF ? 2 : 0][NK];
  double d2[CATF ? 2 : 0][NK];
  int32_t slowx[CATF ? NX : 0];
};

// Dense evaluation of sparse leaves (single output, constant leaves): a wave owns 256 rows, 4 per lane, and the
// per-row evaluation is issued once per row slot e = 0..3 for the whole wave however few lanes have a row of
// the split leaf there -- from the second round on a leaf holds a half, a quarter, ... of the rows, scattered,
// so most of the issued lanes were masked off (round-2 profile: ~127 lane-instructions per touched row against
// ~70 per evaluated row).  When a wave finds at most LL_DENSE_MAX of its 256 rows in the leaf it lists them
// (ballot + mbcnt rank, {sum_trees_noi + offset, side, y}) in a wave-private LDS list and evaluates the list
// 64 at a time: 1-2 evaluations instead of 4.  Order within a side does not matter (integer sums).
#define LL_DENSE_MAX 128
#ifndef PGB_LLK_WGS
#define PGB_LLK_WGS 3  // workgroups per CU the K = 2, 3, 4 instances are compiled for (experiment knob)
#endif

// The K-vector families are two -- softmax and Normal mean/scale (K = 2) -- and pgb_loglikq_t dispatches exactly
// like this.  Naming them here keeps the one-predictor families out of the K >= 2 instances (round 3: k_loglik<4>
// was 92 KB with them, more than the instruction cache two CUs share).  With K known at compile time the softmax
// is K table-driven exponentials and one logarithm, fully unrolled (pgb_loglik_cat_t; round 4 -- the K - 1
// exponentials of the previous form cost more in selects than the exponential they saved).
// (The run-time-K instances, KT = 0, keep the spec's own dispatcher: K = 5..8 and K-vector linear leaves.)
// (trace builds with -DPGB_TRACE_LIST: stamps 27..29 move from the passes into the job list -- 27: pair statistics
//  requested, 28: output 0 of the particles' lanes done, 29: the pairs' statistics have arrived and their values are out)
#ifdef PGB_TRACE_LIST
#define TRLP(i) ((void)0)
#define TRLL(i) TRL(i)
#else
#define TRLP(i) TRL(i)
#define TRLL(i) ((void)0)
#endif
template <int KT, int FAM = -1>
__device__ __forceinline__ double loglik_mk(int family, int K, double y, const double* mu, const pgb_lltabs* tb) {
  if constexpr (KT == 0) {
    return pgb_loglikq_t(family, K, y, mu, 0.0, 1.0, tb);
  } else if constexpr (FAM == PGB_FAMILY_CATEGORICAL) {
    return pgb_loglik_cat_t(KT, y, mu, tb);
  } else if constexpr (FAM == PGB_FAMILY_NORMAL_MEANSCALE) {
    return pgb_loglik_meanscale_t(y, mu, tb);
  } else {
    if (KT >= 3 || family == PGB_FAMILY_CATEGORICAL) return pgb_loglik_cat_t(KT, y, mu, tb);
    return pgb_loglik_meanscale_t(y, mu, tb);
  }
}

// pgb_loglikq_t for a K known only at run time, with the K linear predictors given as a FUNCTION mu(k) that is
// re-evaluated where the spec routine reads an array element: the same operations in the same order (the serial
// maximum, the serial sum of exponentials), hence the same bits -- and no K-sized array in registers or scratch,
// whatever K is (the run-time-K instances carried 184-384 B of scratch for mu[8]; round-3 VERDICT #6).
template <typename MuF>
__device__ __forceinline__ double loglik_fn(int family, int K, double y, MuF mu, const pgb_lltabs* tb) {
  if (family == PGB_FAMILY_CATEGORICAL) {
    double mx = mu(0);
    for (int k = 1; k < K; ++k) {
      const double m = mu(k);
      if (m > mx) mx = m;
    }
    int c = (int)y;
    if (c < 0) c = 0;
    if (c > K - 1) c = K - 1;
    double sum = 0.0, muc = 0.0;
    for (int k = 0; k < K; ++k) {
      const double m = mu(k);
      if (k == c) muc = m;  // (the observed class's predictor, picked up on the way: the same value)
      sum += pgb_exp_t(m - mx, tb->expt);
    }
    double ll = (muc - mx) - pgb_log_pos_t(sum, tb->logt);
    if (!(sum >= 1.0)) ll = -2047.0;
    return PGB_CLAMP_LL(ll, 0.0);
  }
  if (family == PGB_FAMILY_NORMAL_MEANSCALE) {
    const double m2[2] = {mu(0), mu(1)};
    return pgb_loglik_meanscale_t(y, m2, tb);
  }
  return pgb_loglik1q(family, y, mu(0), 0.0, 1.0, tb);
}

// pgb_loglikq_t for a run-time K <= KBND with the predictors in a REGISTER array of compile-time size KBND: every
// loop is unrolled to KBND and guarded by the wave-uniform `k < K` (a scalar branch skips the exponentials of the
// outputs a model does not have), the observed class's predictor is picked by compares instead of a dynamic index.
// Same operations in the same order as pgb_loglik_cat_t.  What the array buys over loglik_fn: the K loads behind it
// are issued together (and a pass ahead), where the one-at-a-time loops paid 2 K serialised memory round trips per
// evaluated row -- the run-time-K pass was 3.5 x slower PER OUTPUT than the unrolled K = 4 instance.
template <int KBND>
__device__ __forceinline__ double loglik_arr(int family, int K, double y, const double (&mu)[KBND], const pgb_lltabs* tb) {
  if (family == PGB_FAMILY_NORMAL_MEANSCALE) {
    const double m2[2] = {mu[0], mu[KBND > 1 ? 1 : 0]};
    return pgb_loglik_meanscale_t(y, m2, tb);
  }
  double mx = mu[0];
#pragma unroll
  for (int k = 1; k < KBND; ++k)
    if (k < K && mu[k] > mx) mx = mu[k];
  int c = (int)y;
  if (c < 0) c = 0;
  if (c > K - 1) c = K - 1;
  double sum = 0.0, muc = 0.0;
#pragma unroll
  for (int k = 0; k < KBND; ++k)
    if (k < K) {
      if (k == c) muc = mu[k];
      sum += pgb_exp_t(mu[k] - mx, tb->expt);
    }
  double ll = (muc - mx) - pgb_log_pos_t(sum, tb->logt);
  if (!(sum >= 1.0)) ll = -2047.0;
  return PGB_CLAMP_LL(ll, 0.0);
}

// The probit instance evaluates pgb_lphi_t on a copy of its table staged in LDS (the header's own layout: coefficient
// pairs [PAIRS][entries][2], one 16-byte read per pair at an immediate offset of ONE address).
// (Measured and dropped, round 4: the coefficient-major layout [9][entries] with nine 8-byte reads -- lanes with
//  different entries then collide only when the entries are 32 apart instead of 16 -- 24.6 -> 32.3 us at cfg4.)
// A table of N doubles (16-byte aligned, N even) on its way into LDS in two halves: every thread REQUESTS its 16-byte
// pieces (stage_load: all of them in flight at once), and stores them once they are needed (stage_store).  The
// requests go out at the head of the kernel, next to the command word: the tables are read once per launch by every
// workgroup and the row passes between two launches push them out of the L2 -- in-kernel stamps (round 5) showed a
// workgroup waiting 1.9 us (exp / log) and 3.6 us (log Phi, 18.5 KB, nine dependent iterations) for them AFTER it
// had learnt that the slot has work.
template <int N>
struct StageRegs {
  static constexpr int PER = (N / 2 + BT - 1) / BT;
  double2 v[PER];
};
template <int N>
__device__ __forceinline__ void stage_load(const double* g, StageRegs<N>& r) {
#pragma unroll
  for (int i = 0; i < StageRegs<N>::PER; ++i) {
    const int idx = (int)threadIdx.x + i * BT;
    r.v[i] = idx < N / 2 ? gload_d2(as_global(g) + 2 * idx) : double2{0.0, 0.0};
  }
}
template <int N>
__device__ __forceinline__ void stage_store(double* s_tab, const StageRegs<N>& r) {
#pragma unroll
  for (int i = 0; i < StageRegs<N>::PER; ++i) {
    const int idx = (int)threadIdx.x + i * BT;
    if (idx < N / 2) ((double2*)s_tab)[idx] = r.v[i];
  }
}
__device__ __forceinline__ double lphi_lds(double s, const double* s_tab) { return pgb_lphi_t(s, s_tab); }

// pgb_quant of a per-row log-likelihood that has already been clamped to the contract's range
// (|ll| <= 2047, never NaN: every pgb_loglik* routine ends with that clamp): |ll * cl| < 2^50, so none of
// pgb_quant's NaN / saturation branches can fire and what is left of it is the rounding itself.  Same bits.
__device__ __forceinline__ long long quant_ll(double ll, double cl) {
  const double mg = ll * cl + 6755399441055744.0; /* 1.5 * 2^52 */
  return (long long)(pgb_d2u(mg) - 0x4338000000000000ull);
}

// KT: 1 = single output; 2, 3, 4 = that many outputs, loops unrolled; 0 = any K <= PGB_MAX_OUTPUTS
// FAM: the likelihood family when known at compile time (single-output kernels: the per-row
// evaluation then contains one family's code only), -1: read S.family.
// LIN: linear response (single-output families): the children predict value + slope (x - xbar).
template <int KT, int FAM, bool LIN>
// (compiled for 3 workgroups per CU, i.e. <= 168 VGPRs: the K = 4 instance sits right at that edge, and one
//  register more costs it a third of its waves -- 32 -> 40 us per launch at cfg5;
//  the probit instance -- cfg4's dominant kernel -- for 5: <= 96 VGPRs, where a 97th costs it a fifth)
__global__ __launch_bounds__(BT, (KT == 1 && FAM == PGB_FAMILY_BERNOULLI_PROBIT && !LIN) ? 5 : (KT >= 2 && !LIN) ? PGB_LLK_WGS : KT == 0 ? 2 : 3)
void k_loglik(const Dev* __restrict__ Sp, int par, int nwg, const Cmd* __restrict__ cmds, const Ctrl* __restrict__ ctrls,
              const Job* __restrict__ jobs_all, const Acc* __restrict__ acc_all, const InitAcc* __restrict__ ias) {
  // cmds / ctrls / jobs_all / acc_all / ias repeat S.cmd / S.ctrl / S.jobs / S.acc / S.initacc as kernel arguments
  // (preloaded into SGPRs): the first loads of the launch -- the command word, the control word, the job records and
  // their statistics -- go out together, at once, instead of behind a load of their pointers from the argument block
  // S (in-kernel stamps, round 5: 4.1 - 4.9 us from the entry of a workgroup to the end of its job list, in every
  // launch of every instance: four dependent memory round trips and a few hundred instructions)
  // nwg repeats gridDim.x as an explicit argument (it takes the padding after `par`): gridDim.x is a HIDDEN argument,
  // a scalar load from the kernel-argument segment with its wait in front of the first item of the passes
  const DevG& S = *reinterpret_cast<const DevG*>(Sp);
  // what the passes need of the argument block, requested with everything else at the head of the launch (read
  // where they are used, these scalar loads and the hidden grid size stood between the job list and the first
  // label words: 2.6 us at cfg5, in-kernel stamps)
  const long long n = S.n;
  const int nchunks_h = S.nchunks;
  const double cl = S.sc.cl;
  const gptr<const double> gy = as_global(S.y), goff = as_global(S.off);
  const gptr<const uint8_t> glid = as_global((const uint8_t*)S.lid);
  const double* const st_h = S.st;
  const long long n_pad_h = S.n_pad;
  constexpr bool MK = KT != 1;
  // (K-vector leaves: the arrays the job list reads per (particle, extension output) -- as pointers of the argument
  //  block read where the list uses them, they were a scalar round trip in front of the list's own)
  const long long* const accx_h = MK ? (const long long*)S.accx : nullptr;
  const long long* const jqx_h = MK ? (const long long*)S.jqx : nullptr;
  const double* const jvx_h = MK ? (const double*)S.jvx : nullptr;
  const double* const jzx_h = MK ? (const double*)S.jzx : nullptr;
  const double init_leaf_h = S.init_leaf;
  constexpr int KB = KT > 0 ? KT : PGB_MAX_OUTPUTS;  // compile-time bound of the K loops (run-time K: guarded by k < K)
  // softmax with constant leaves: the factorised evaluation (pgbart_spec.h, pgb_loglik_cat_f)
  constexpr bool CATF = MK && !LIN && FAM == PGB_FAMILY_CATEGORICAL;
  constexpr int FAMK = MK ? FAM : -1;  // the family of a K-vector instance when it is known at compile time
  typedef LJobT<MK, LIN, KT, CATF> LJob;
#ifdef PGB_STAMP_LL  // experiment builds: this kernel, not the row pass, leaves the per-workgroup clock readings
  struct StampEnd {
    long long* p;
    __device__ ~StampEnd() { if (p) p[1] = wall_clock64(); }
  } stamp_end{nullptr};
  if (S.prof_stamps != nullptr && threadIdx.x == 0 && blockIdx.x < PROF_BLOCKS) {
    stamp_end.p = S.prof_stamps + ((size_t)((S.ctrl[par ^ 1].slot_no - 1) % PROF_RING) * PROF_BLOCKS + blockIdx.x) * 2;
    stamp_end.p[0] = wall_clock64();
    stamp_end.p[1] = stamp_end.p[0];
  }
#endif
  // per (particle of the span, side): the four waves ADD their totals (LDS atomics; the reducer thread reads the sum
  // and leaves a zero) -- a quarter of the storage of one cell per wave, which is what fits a fifth workgroup of the
  // probit instance into the CU next to its 18.6 KB table
  __shared__ long long s_red[MAXP * 3 < 8 ? 8 : MAXP * 3];
  __shared__ LJob s_job[MAXP];
  __shared__ int s_n[2];
  // the tables of the per-row likelihood math in LDS (a per-lane table row through the vector L1 costs a
  // cache-line access per distinct row and instruction; LDS serves them at bank speed): log Phi for the probit
  // instance (10.4 KB, coefficient-major: lanes reading coefficient k of different rows hit different banks),
  // exp / log (2.3 KB) for every instance whose family uses them
  constexpr bool PROBIT = KT == 1 && FAM == PGB_FAMILY_BERNOULLI_PROBIT;
  constexpr bool EXPLOG = !(KT == 1 && (FAM == PGB_FAMILY_BERNOULLI_PROBIT || FAM == PGB_FAMILY_ASYMLAPLACE ||
                                        FAM == PGB_FAMILY_CALLBACK));
  __shared__ __attribute__((aligned(16))) double s_lphi[PROBIT ? PGB_LPHI_SIZE : 2];
  __shared__ __attribute__((aligned(16))) double s_expt[EXPLOG ? PGB_EXPT_SIZE : 2];
  __shared__ __attribute__((aligned(16))) double s_logt[EXPLOG ? PGB_LOGT_SIZE : 2];
  pgb_lltabs tb;
  tb.lphi = pgb_tab_lphi();  // (the probit instance evaluates through lphi_lds on its staged copy)
  tb.expt = EXPLOG ? s_expt : pgb_tab_exp();
  tb.logt = EXPLOG ? s_logt : pgb_tab_log();
  // wave-private lists of the dense path (see LL_DENSE_MAX); Bernoulli responses travel as a flag bit
  // (the evaluation is inlined a second time: the families whose instance would cross an occupancy edge with
  //  it -- Poisson, NegativeBinomial, Gamma: two exp / log chains each -- keep the plain path;
  //  tools/occupancy_guard.py holds the line)
  constexpr bool DENSE = KT == 1 && !LIN &&
                         (FAM == PGB_FAMILY_BERNOULLI_PROBIT || FAM == PGB_FAMILY_BERNOULLI_LOGIT ||
                          FAM == PGB_FAMILY_ASYMLAPLACE || FAM == PGB_FAMILY_STUDENT_T);
  constexpr bool YBIT = FAM == PGB_FAMILY_BERNOULLI_PROBIT || FAM == PGB_FAMILY_BERNOULLI_LOGIT;
  __shared__ double s_lnv[DENSE ? BT / 64 : 1][DENSE ? LL_DENSE_MAX : 1];
  __shared__ double s_ly[DENSE && !YBIT ? BT / 64 : 1][DENSE && !YBIT ? LL_DENSE_MAX : 1];
  __shared__ uint16_t s_lfl[DENSE ? BT / 64 : 1][DENSE ? LL_DENSE_MAX : 1];
  constexpr bool MKPASS = MK && !LIN;  // K-vector constant leaves: the pass loop (see below); KT = 0: any K
  __shared__ uint16_t s_lrow[MKPASS ? BT / 64 : 1][MKPASS ? LL_DENSE_MAX : 1];
  // (Measured and dropped: listing each wave's matching rows with ballot + mbcnt and evaluating the
  // list densely -- per particle, or through a per-wave queue with three interleaved passes -- is
  // SLOWER at cfg4, 184 k / 171 k vs 223 k particle-steps/s: with a quarter of the lanes active the
  // rare branches of the evaluation are skipped wave-wide, with every lane active they never are.)
  const Cmd* cmd = &cmds[par];
  // The job records and split statistics of this lane's particles are requested FIRST, next to the command word
  // (their addresses depend on `par` alone; the records exist for every index below MAXP): by the time the command
  // says that this slot has a round to evaluate, they are on their way.  (k_ctrl starts the same way.)
  Job j_pre[MAXP / 64];
  Acc a_pre[MAXP / 64];
  if (threadIdx.x < 64) {
#pragma unroll
    for (int hq = 0; hq < MAXP / 64; ++hq) {
      const int q = (int)(threadIdx.x & 63) + 64 * hq;
      j_pre[hq] = jobs_all[(size_t)par * MAXP + q];
      a_pre[hq] = load_acc(&acc_all[((size_t)par * MAXP + q) * ACC_PER]);
    }
  }
#ifndef PGB_LL_PAIRS
#define PGB_LL_PAIRS 1 /* experiment knob: 1 = job list: one lane per (active particle, extension output), see there */
#endif
  constexpr bool PAIRS = MK && !LIN && PGB_LL_PAIRS != 0;
  // ... and the control word this slot's k_ctrl produced (the round decides where a K-vector job's parent sums are,
  // the pending leaf_sd update what output 0 needs: read behind the command word, it was a round trip of its own in
  // front of the list's requests -- 0.8 us from the job records' arrival to the first of them, stamps)
  const Ctrl cn = ctrls[par ^ 1];
  // ... and so are the tables (see stage_load)
  StageRegs<PROBIT ? PGB_LPHI_SIZE : 2> st_lphi;
  StageRegs<EXPLOG ? PGB_EXPT_SIZE : 2> st_exp;
  StageRegs<EXPLOG ? PGB_LOGT_SIZE : 2> st_log;
  if constexpr (PROBIT) stage_load<PGB_LPHI_SIZE>(pgb_tab_lphi(), st_lphi);
  if constexpr (EXPLOG) {
    stage_load<PGB_EXPT_SIZE>(pgb_tab_exp(), st_exp);
    stage_load<PGB_LOGT_SIZE>(pgb_tab_log(), st_log);
  }
  // the head of the command record -- kind .. st_cur, 32 bytes -- in ONE scalar load: dst_gen is needed for the first
  // label words of the passes, and read where it is used it was another memory round trip in front of them (stamps)
  struct CmdHead {
    int32_t kind, tree_old, tree_new, sel_gen, sel_slot, tune, dst_gen, st_cur;
  };
  static_assert(offsetof(Cmd, st_cur) == offsetof(CmdHead, st_cur), "CmdHead mirrors the head of Cmd");
  const CmdHead ch = load_uniform(reinterpret_cast<const CmdHead*>(cmd));
  if (!(ch.kind & CMD_PARTITION)) return;
  TRL_BIND(cn.slot_no - 1);
  TRL(24);
  if constexpr (PROBIT) stage_store<PGB_LPHI_SIZE>(s_lphi, st_lphi);
  if constexpr (EXPLOG) {
    stage_store<PGB_EXPT_SIZE>(s_expt, st_exp);
    stage_store<PGB_LOGT_SIZE>(s_logt, st_log);
  }
  if constexpr (CATF) __syncthreads();  // the job list below evaluates exponentials (the children's w) on the LDS table
  // (the barrier after the job list below also publishes the tables)
  TRL(36);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int round = cn.round - 1;   // round of the proposals of this slot
  const uint32_t it = (uint32_t)cn.iter;
  // leaf_sd / root statistics in force for this round (k_ctrl of the NEXT slot resolves them the
  // same way): a FINAL or INIT part of this slot's row pass may just have produced them
  double leaf_sd = cn.leaf_sd;
  long long rootA = 0;
  {
    const InitAcc* src = ias + (size_t)par * IA_SLOTS;
    long long qstd = 0;
    for (int k = 0; k < IA_SLOTS; ++k) {
      qstd += src[k].QSTD;
      rootA += src[k].A;
    }
    if (cn.pend_leafsd && cn.pend_iter > 2) leaf_sd = pgb_tuned_leaf_sd(cn.leaf_sd, cn.pend_iter, qstd, S.sc.inv_c1, S.n);
  }
  // The list of this slot's active particles with their children's leaf values (the routines k_ctrl runs one launch
  // later, same inputs).  Every launch starts with it, on its critical path (measured with in-kernel stamps at cfg5:
  // 4.1 us of a 10 us launch in a plain round, 7.2 us in the slot that starts a tree -- one wave listing 39 particles,
  // three extension outputs one after the other, every statistic loaded where it was used).  Now: wave 0 does output
  // 0 and the fields of the record; K-vector constant leaves: extension output kx on wave 1 + kx % 3, side by side;
  // and every statistic a lane may need is REQUESTED before anything is known about the particle (one round trip).
  // (XW: the extension outputs on waves 1..3; linear leaves: wave 0, they chain)
  // The list comes in two loops over the blocks of 64 particles (wave 0).  The first: which particles have a job, of
  // each record the fields that say WHERE its rows are (they depend on the job record alone), and the requests for the
  // extension outputs' statistics; the second: the children's values.  (Tried between the two: a barrier and every
  // wave's request for the label words of its first unit, so that they travel under the second loop -- 0.28 us less in
  // front of the passes, 0.42 us more in the list, cfg5 782 / 780 k against 791 / 785 k: dropped.)
  const int KXr = MK ? (KT > 0 ? KT : S.K) - 1 : 0;
  struct PairIn {
    long long axL, axN, pq;
    double pv, z0, z1;
  };
  constexpr int NHQ = MAXP / 64;
  unsigned long long mm[NHQ];
  int nl0[NHQ];  // list position of the first active particle of a block of 64
  PairIn x0[NHQ];
  int kk0[NHQ], kx0[NHQ];
  auto pair_load = [&](int nlist, int i, PairIn& x, int& kk, int& kx) {
    if constexpr (!PAIRS) return;
    kk = nlist + i / KXr;
    kx = i - (i / KXr) * KXr;
    const int q2 = s_job[kk].p;
    const size_t e = ((size_t)par * MAXP + q2) * KXr + kx;
    x.axL = load_accx(accx_h, par, q2, kx);
    x.axN = load_accx(accx_h, par, q2, KXr + kx);
    x.pq = round == 0 ? root_A_x(S, par, kx) : jqx_h[e];
    x.pv = round == 0 ? init_leaf_h : jvx_h[e];
    x.z0 = jzx_h[e * 2];  // drawn by this slot's control kernel
    x.z1 = jzx_h[e * 2 + 1];
  };
  if (tid < 64) {
    const int ln = tid;
    int nlist = 0;  // (lanes' particles ln, ln + 64, ...: one block of 64 after the other)
#pragma unroll
    for (int hq = 0; hq < NHQ; ++hq) {
      const int q = ln + 64 * hq;
      const bool inr = q >= 1 && q < S.P;
      const Job& j = j_pre[hq];  // (requested at the head of the kernel)
      const bool has = inr && j.active != 0;
      const unsigned long long m = __ballot(has);
      if (hq == 0) TRL(37);
      mm[hq] = m;
      nl0[hq] = nlist;
      const int k = nlist + __popcll(m & ((1ull << ln) - 1ull));
      if (has) {
        LJob& lj = s_job[k];  // (filled in place: a local record with its K-sized arrays would live in scratch)
        lj.p = q;
        lj.rule = j.rule;
        lj.label = j.label;
        lj.new_label = j.new_label;
        lj.check_nan = j.check_nan;
        lj.v = j.v;
        lj.src = j.src_slot < 0 ? -1ll : (long long)(((size_t)j.src_gen * MAXP + j.src_slot) * S.n_pad);
        lj.xoff = (long long)((size_t)j.var * S.n_pad);
        lj.dst = (long long)((size_t)q * S.n_pad);
      }
      // K-vector constant leaves: the extension outputs of the block's active particles, ONE per lane -- pair i =
      // (i / KX-th active particle of the block, output 1 + i % KX), 64 pairs at a time.  (One lane per particle ran
      // its K - 1 outputs one after the other: K - 1 dependent chains of two leaf values and two exponentials each,
      // behind one another on a wave that has the SIMD to itself -- 2.5 us of a 10 us launch at cfg5, and all but ~10
      // of the 64 lanes idle in a plain round.)  A pair's statistics are requested as soon as the ballot says which
      // particles have a job, and arrive while the particles' own lanes derive output 0.
      x0[hq] = PairIn{0, 0, 0, 0.0, 0.0, 0.0};
      kk0[hq] = kx0[hq] = 0;
      if constexpr (PAIRS) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (ln < __popcll(m) * KXr) pair_load(nlist, ln, x0[hq], kk0[hq], kx0[hq]);
      }
      if (hq == 0) TRLL(27);
      nlist += __popcll(m);
    }
    if (tid == 0) s_n[0] = nlist;
  }
  unsigned sat = 0;
  if (tid < 64) {
    const int ln = tid;
#pragma unroll
    for (int hq = 0; hq < NHQ; ++hq) {
    const int q = ln + 64 * hq;
    const Job& j = j_pre[hq];
    const Acc& a = a_pre[hq];
    const unsigned long long m = mm[hq];
    const bool has = (m >> ln) & 1ull;
    const int nlist = nl0[hq];
    const int k = nlist + __popcll(m & ((1ull << ln) - 1ull));
    const int npair = PAIRS ? __popcll(m) * KXr : 0;  // (wave-uniform)
    // the leaf noise of particle `ln` in this round: drawn by the control kernel of this slot, in the job
    const double z0 = has ? j.z0 : 0.0, z1 = has ? j.z1 : 0.0;
    if (has) {
      const ChildVals cv = child_values(S, j.rule, j.cnt, round == 0 ? rootA : j.p_q_st, j.p_value, a.cnts,
                                        a.aL, a.aN, z0, z1, leaf_sd);
      LJob& lj = s_job[k];
      lj.ok = cv.ok;
      lj.cL = cv.cL;
      lj.cR = cv.cR;
      lj.vL = cv.vL;
      lj.vR = cv.vR;
      if constexpr (CATF) {  // output 0 of the children's part: d = v - v = 0, w = exp(0) = 1 exactly
        lj.d2[0][0] = lj.d2[1][0] = 0.0;
        lj.w2[0][0] = lj.w2[1][0] = 1.0;
      }
      if constexpr (LIN) {
        lj.slopeL = lj.xbarL = lj.slopeR = lj.xbarR = 0.0;
        lj.svarL = lj.svarR = -1;
      }
      LinKids lk;
      lk.svarL = lk.svarR = -1;
      lk.linL = lk.linR = false;
      if constexpr (LIN) {
        if (cv.ok == 1) {
          lk = lin_children(S, &S.accu[((size_t)par * MAXP + q) * ACC_PER], j.var, cv.cL, cv.cR,
                            cv.aL, cv.aR, it, (uint32_t)round, (uint32_t)q);
          lj.svarL = lk.svarL; lj.slopeL = lk.slopeL; lj.xbarL = lk.xbarL;
          lj.svarR = lk.svarR; lj.slopeR = lk.slopeR; lj.xbarR = lk.xbarR;
        }
      }
      if constexpr (MK && !PAIRS)
      for (int kx = 0; kx < (KT > 0 ? KT : S.K) - 1; ++kx) {  // extension outputs: same routine as k_ctrl
        const int KX = (KT > 0 ? KT : S.K) - 1;
        const long long pq = round == 0 ? root_A_x(S, par, kx) : S.jqx[((size_t)par * MAXP + q) * KX + kx];
        const double pv = round == 0 ? S.init_leaf : S.jvx[((size_t)par * MAXP + q) * KX + kx];
        const double* zz = S.jzx + (((size_t)par * MAXP + q) * KX + kx) * 2;  // drawn by this slot's control kernel
        ChildX cx = child_values_x(S, cv.ok, cv.cL, cv.cR, load_accx(S.accx, par, q, kx),
                                   load_accx(S.accx, par, q, KX + kx), pq, pv, zz[0], zz[1],
                                   leaf_sd_x(S, cn, par ^ 1, par, kx));
        lj.vLx[kx] = cx.vL;
        lj.vRx[kx] = cx.vR;
        if constexpr (!LIN)  // handed to the next slot's control kernel (Dev::finx): it needs exactly these
          if (blockIdx.x == 0) S.finx[((size_t)par * MAXP + q) * KX + kx] = FinX{cx.vL, cx.vR, cx.aL, cx.aR};
        if constexpr (LIN) {
          if (cv.ok == 1)
            lin_children_x(S, lk, cx, j.var, cv.cL, cv.cR, load_accx(S.accux, par, q, kx),
                           load_accx(S.accux, par, q, KX + kx));
          lj.sLx[kx] = cx.sL;
          lj.sRx[kx] = cx.sR;
        }
        if constexpr (CATF) {  // the child's part of the factorised softmax for this output (pgb_cat_side)
          const double dL = cx.vL - cv.vL, dR = cx.vR - cv.vR;
          lj.d2[0][kx + 1] = dL;
          lj.d2[1][kx + 1] = dR;
          lj.w2[0][kx + 1] = pgb_exp_t(dL, tb.expt);
          lj.w2[1][kx + 1] = pgb_exp_t(dR, tb.expt);
          lj.slowx[kx] = (!(dL <= PGB_CAT_DMAX && dL >= -PGB_CAT_DMAX) ? 1 : 0) |
                         (!(dR <= PGB_CAT_DMAX && dR >= -PGB_CAT_DMAX) ? 2 : 0);
        }
      }
      if constexpr (MK && LIN)
        if (cv.ok == 1) {  // a further output may have made the leaf linear
          lj.svarL = lk.svarL; lj.slopeL = lk.slopeL; lj.xbarL = lk.xbarL;
          lj.svarR = lk.svarR; lj.slopeR = lk.slopeR; lj.xbarR = lk.xbarR;
        }
    }
    if (hq == 0) TRLL(28);
    if constexpr (PAIRS) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // (the particles' lanes have written ok / cL / cR / vL / vR of their records)
      for (int i0 = 0; i0 < npair; i0 += 64) {
        const int i = i0 + ln;
        PairIn x = x0[hq];
        int kk = kk0[hq], kx = kx0[hq];
        if (i0 != 0 && i < npair) pair_load(nlist, i, x, kk, kx);
        if (i < npair) {
          LJob& lj = s_job[kk];
          const ChildX cx = child_values_x(S, lj.ok, lj.cL, lj.cR, x.axL, x.axN, x.pq, x.pv, x.z0, x.z1,
                                           leaf_sd_x(S, cn, par ^ 1, par, kx));  // same routine as k_ctrl
          lj.vLx[kx] = cx.vL;
          lj.vRx[kx] = cx.vR;
          if (hq == 0 && i0 == 0) TRLL(29);
          // handed to the next slot's control kernel (Dev::finx): it needs exactly these
          if (blockIdx.x == 0) S.finx[((size_t)par * MAXP + lj.p) * KXr + kx] = FinX{cx.vL, cx.vR, cx.aL, cx.aR};
          if constexpr (CATF) {  // the child's part of the factorised softmax for this output (pgb_cat_side)
            const double dL = cx.vL - lj.vL, dR = cx.vR - lj.vR;
            lj.d2[0][kx + 1] = dL;
            lj.d2[1][kx + 1] = dR;
            lj.w2[0][kx + 1] = pgb_exp_t(dL, tb.expt);
            lj.w2[1][kx + 1] = pgb_exp_t(dR, tb.expt);
            lj.slowx[kx] = (!(dL <= PGB_CAT_DMAX && dL >= -PGB_CAT_DMAX) ? 1 : 0) |
                           (!(dR <= PGB_CAT_DMAX && dR >= -PGB_CAT_DMAX) ? 2 : 0);
          }
        }
      }
    }
    }
  }
  TRL(38);
  __syncthreads();
  TRL(25);
  const int nact = s_n[0];
  TRL(26);
  if (nact != 0) {  // (the passes; a slot without an active particle goes straight to the INIT part)
  for (int i = tid; i < MAXP * 3; i += BT) s_red[i] = 0;  // (the waves add into it; the INIT part uses it as scratch)
  __syncthreads();
  TRL(39);
  // (arrays of the argument block as GLOBAL pointers -- see as_global: a flat load also counts in lgkmcnt, so
  //  the first LDS table read of an evaluation waited for the label words requested for the NEXT particle)
  // Work = nchunks x nact (chunk, particle) units.  The persistent grid takes them as ONE linear range cut into
  // gridDim.x equal spans (chunk-major): a workgroup gets a run of particles of one chunk and, if its span
  // crosses a chunk boundary, the first particles of the next.  (Whole items of G particles left the grid
  // unbalanced: cfg4, round 0: 977 chunks x 2 groups over 1280 workgroups = 674 workgroups with 39 units,
  // 606 with 30 or 9 -- 76 % of the pass's lanes.)
  const int W = nchunks_h * nact;  // (< 2^31: n < 2^31 rows, at most 63 particles)
  const int per = W / nwg, rem = W - per * nwg, bx = (int)blockIdx.x;
  const int u_lo = bx * per + (bx < rem ? bx : rem), u_hi = u_lo + per + (bx < rem ? 1 : 0);
  const int K = KT > 0 ? KT : S.K;
  const gptr<const double> noi = as_global(st_h + (size_t)cn.st_cur * K * n_pad_h);
  // The row pass of this slot has already sorted the rows of every split leaf: left rows kept the
  // leaf's label, right rows carry the new one, dropped rows the orphan label.  Reading those
  // bytes back (1 B per row) replaces a second read of the split column (8 B per row).
  const gptr<const uint8_t> newl = glid + (size_t)ch.dst_gen * MAXP * n_pad_h;
  for (int u = u_lo; u < u_hi;) {
    const int chunk = u / nact, g0 = u - chunk * nact;
    const int g1 = nact - g0 < u_hi - u ? nact : g0 + (u_hi - u);
    u += g1 - g0;
    const long long base = (long long)chunk * CH + tid * RPT;
    double yv[RPT], nv[RPT];
    uint32_t ysgn[RPT];  // (Bernoulli families: sign mask of the predictor, once per row instead of once per evaluation)
    uint32_t root_ids = 0;
#pragma unroll
    for (int e = 0; e < RPT; ++e) {
      if constexpr (!MKPASS) {  // (the pass loop of the K = 2, 3, 4 instances fetches a row's inputs where it evaluates it)
        yv[e] = gy[base + e];
        if constexpr (YBIT) ysgn[e] = yv[e] > 0.5 ? 0u : 0x80000000u;  // Bernoulli: the response only picks the sign
        nv[e] = noi[base + e];
        if constexpr (KT == 1)
          if (S.has_off) nv[e] = nv[e] + goff[base + e];  // (adding the default 0.0 would give the same bits)
      }
      if (base + e >= n) root_ids |= (uint32_t)PGB_ORPHAN << (8 * e);
    }
    if constexpr (MKPASS) {
      // K = 2, 3, 4, constant leaves: ONE evaluation site, visited once per PASS.  Plain: pass e evaluates the
      // lane's own row e where it is in the split leaf (4 passes).  Dense (a wave with <= LL_DENSE_MAX of its 256
      // rows in the leaf, i.e. every round but the first few): the wave lists {row in chunk, side} of its
      // matching rows (ballot + mbcnt rank) and pass k evaluates list entries 64 k .. 64 k + 63 -- 1-2 passes
      // instead of 4, none at all for a wave without a matching row.  A row's inputs (y, K predictors) are
      // fetched where it is evaluated (L1 hits: the chunk was just read by this workgroup).
      const long long cbase = (long long)chunk * CH;
      // the chunk's streams as wave-uniform bases (scalar registers) + a 32-bit row offset: the loads of a row's
      // inputs then cost one shared offset instead of a 64-bit address computation each
      const gptr<const double> gy_c = gy + cbase;
      gptr<const double> noi_c[KB], off_c[KB];
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const int kc = k < K ? k : 0;  // (outputs the model does not have: never read)
        noi_c[k] = noi + (size_t)kc * S.n_pad + cbase;
        off_c[k] = goff + (size_t)kc * S.n_pad + cbase;
      }
      const bool has_off = S.has_off != 0;
      // softmax, constant leaves: the row parts E_k / a_c / class of the chunk (written by the slot that started the
      // tree; output k of row r at e_c + k n_pad + r)
      gptr<const double> e_c = nullptr, a_cb = nullptr;
      gptr<const uint8_t> cls_c = nullptr;
      uint32_t e_stride = 0;  // bytes between two outputs of a row
      if constexpr (CATF) {
        e_c = as_global((const double*)S.cat_e) + cbase;
        a_cb = as_global((const double*)S.cat_a) + cbase;
        cls_c = as_global((const uint8_t*)S.cat_c) + cbase;
        e_stride = (uint32_t)(S.n_pad * 8);  // (K n_pad 8 < 2^32 is NOT assumed: see e_at)
      }
      auto e_at = [&](int k, uint32_t ro) -> double {  // E_k of the row at byte offset ro of its chunk
        return gload_d_off(e_c + (size_t)k * S.n_pad, ro);
      };
      (void)e_stride;
      // a lane's OWN four rows (y and the K predictors without the leaf value): what every particle whose leaf still
      // holds most of the wave's rows evaluates, lane by row -- in the slot that starts a tree that is every particle
      // of the span, and the same 20 values were fetched again for each of them, pass by pass.  Compile-time K:
      // loaded once per item, when the first such particle comes up, and kept in registers.
      constexpr bool OWN = KT > 0;
      // (softmax: own_y holds a_c, own_nk the E_k, own_cls the class -- the row part of the factorised form)
      double own_y[OWN ? RPT : 1], own_nk[OWN ? RPT : 1][OWN ? KB : 1];
      uint32_t own_cls = 0;  // (the four rows' classes, a byte each)
      bool own_have = false;
      // the label words of particle g + 1 are requested before particle g's passes (each of which is hundreds of
      // instructions): requested where they are used, they cost every particle of the span a memory round trip in
      // front of its first ballot -- with the first pass's inputs behind it, half of the wave-cycles of the launch
      // that starts a tree at cfg5 (three waves per SIMD: nothing else to issue meanwhile)
      const uint32_t base32 = (uint32_t)base;  // (n < 2^31)
      uint32_t ids_nx = root_ids, nid_nx = 0;
      auto fetch_labels = [&](int gg) {
        const LJob& ln = s_job[gg];
        // (the record is the same in every lane: its offsets and labels go to scalar registers, the label words
        //  are fetched at scalar base + 32-bit row offset)
        const long long src_u = uni(ln.src), dst_u = uni(ln.dst);
        ids_nx = src_u < 0 ? root_ids : gload_u32_off(glid + src_u, base32);
        nid_nx = gload_u32_off(newl + dst_u, base32);
      };
      if (g0 < g1) fetch_labels(g0);
      TRLP(27);
      for (int g = g0; g < g1; ++g) {
        const LJob& lj = s_job[g];
        const uint32_t ids = ids_nx, nid = nid_nx;
        if (g == g0 + 1) TRLP(28);  // (the first particle of the item done: own rows loaded / computed with it)
        if (g + 1 < g1) fetch_labels(g + 1);
        const uint32_t lab = uni((uint32_t)lj.label), nlab = uni((uint32_t)lj.new_label);
        // (compile-time K: the particle's leaf values in registers; run-time K: read from its LDS record where used.
        //  Softmax: what the evaluation needs of a child is its w_k -- the leaf values themselves only in the rare
        //  fallback, which reads them from the LDS record)
        constexpr int KV = KT > 0 ? KT : 1;
        double vLr[KV], vRr[KV];
        if constexpr (CATF) {
#pragma unroll
          for (int k = 0; k < KV; ++k) {
            vLr[k] = lj.w2[0][k];
            vRr[k] = lj.w2[1][k];
          }
        } else {
          vLr[0] = lj.vL;
          vRr[0] = lj.vR;
#pragma unroll
          for (int k = 1; k < KV; ++k) {
            vLr[k] = lj.vLx[k - 1];
            vRr[k] = lj.vRx[k - 1];
          }
        }
        uint32_t slowm = 0;  // softmax: bit 0 / 1 = the left / right child is a slow one (pgb_cat_side)
        if constexpr (CATF) {
#pragma unroll
          for (int k = 0; k < LJob::NX; ++k)
            if (k < K - 1) slowm |= (uint32_t)lj.slowx[k];
          slowm = uni(slowm);
        }
        unsigned long long mk[RPT];
        int M = 0;
#pragma unroll
        for (int e = 0; e < RPT; ++e) {
          mk[e] = __ballot(((ids >> (8 * e)) & 255u) == lab);
          M += __popcll(mk[e]);
        }
        // (wave-uniform.  Softmax with K known at compile time: round 0 -- whose row parts are not in memory yet --
        //  always takes the lane's own rows, computed once per item, also in the partial last chunk)
        const bool dense = M <= LL_DENSE_MAX && !(CATF && KT > 0 && round == 0);
        if (dense && M > 0) {
          int off = 0;
#pragma unroll
          for (int e = 0; e < RPT; ++e) {
            if (((ids >> (8 * e)) & 255u) == lab) {
              const int pos = off + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk[e] >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((unsigned)mk[e], 0u));
              const uint32_t nl = (nid >> (8 * e)) & 255u;
              const uint32_t side = nl == lab ? 0u : (nl == nlab ? 1u : 2u);
              s_lrow[w][pos] = (uint16_t)((uint32_t)(tid * RPT + e) | (side << 10));
            }
            off += __popcll(mk[e]);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if constexpr (OWN) {
          if (!dense && !own_have) {  // (wave-uniform)
            if (CATF && round > 0) {  // the row parts the slot that started the tree left in memory
              if constexpr (CATF) {
#pragma unroll
                for (int e = 0; e < RPT; ++e) {
                  const uint32_t r = (uint32_t)(tid * RPT + e);
                  own_y[e] = gload_d_off(a_cb, r * 8u);
                  own_cls |= (uint32_t)cls_c[r] << (8 * e);
#pragma unroll
                  for (int k = 0; k < KB; ++k) own_nk[e][k] = e_at(k, r * 8u);
                }
              }
            } else {
#pragma unroll
              for (int e = 0; e < RPT; ++e) {
                const uint32_t ro = (uint32_t)(tid * RPT + e) * 8u;
                own_y[e] = gload_d_off(gy_c, ro);
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                  const double nk = gload_d_off(noi_c[k], ro);
                  own_nk[e][k] = has_off ? nk + gload_d_off(off_c[k], ro) : nk;
                }
              }
              if constexpr (CATF) {
                // round 0 (this launch also writes the row parts, but for the whole data set, by other workgroups):
                // the same values from the same inputs, pgb_cat_row
#pragma unroll
                for (int e = 0; e < RPT; ++e) {
                  const int ce = pgb_cat_class(KT, own_y[e]);
                  own_cls |= (uint32_t)ce << (8 * e);
                  own_y[e] = pgb_cat_row(KT, ce, own_nk[e], tb.expt, own_nk[e]);
                }
              }
            }
            own_have = true;
          }
        }
        const int npass = dense ? (M + 63) >> 6 : RPT;
        // (only a split on a column with missing values drops rows -- uniform per particle: the pass loop is
        //  compiled with and without the third side, see the single-output loop below)
        auto passes = [&](auto drops_c) {
          constexpr bool DROPS = decltype(drops_c)::value;
          long long vt = 0, v0 = 0, v2 = 0;
          // a pass's inputs: which row (if any) this lane evaluates, its side, y and the K predictors without the
          // leaf value.  They are requested ONE PASS AHEAD: the evaluation of pass ps (~175 vector instructions)
          // covers the L1 / L2 round trip of the loads of pass ps + 1 (round 4; the pass loop stalled on them:
          // 55 % of the wave-cycles waiting at three waves per SIMD).
          struct PassIn {
            bool act;
            uint32_t side, ro;  // (ro: the row's byte offset in its chunk -- what the run-time-K evaluation reads with)
            int cls;            // softmax: the observed class (y then holds a_c and nk the E_k: the row part)
            double y, nk[KB];
          };
          auto fetch = [&](int ps) -> PassIn {
            PassIn in;
            uint32_t r;
            if (dense) {
              const int k = lane + 64 * ps;
              in.act = k < M;
              const uint32_t ent = in.act ? (uint32_t)s_lrow[w][k] : 0u;
              r = ent & 1023u;
              in.side = ent >> 10;
            } else {
              r = (uint32_t)(tid * RPT + ps);
              in.act = ((ids >> (8 * ps)) & 255u) == lab;
              const uint32_t nl = (nid >> (8 * ps)) & 255u;
              in.side = nl == lab ? 0u : (nl == nlab ? 1u : 2u);
            }
            in.y = 0.0;
            in.cls = 0;
            in.ro = r * 8u;  // (r < 1024)
#pragma unroll
            for (int k = 0; k < KB; ++k) in.nk[k] = 0.0;
            if constexpr (OWN) {
              if (!dense) {  // the lane's own row ps: in registers (the pass loop is unrolled over ps below)
#pragma unroll
                for (int e = 0; e < RPT; ++e)
                  if (e == ps) {
                    in.y = own_y[e];
                    if constexpr (CATF) in.cls = (int)((own_cls >> (8 * e)) & 255u);
#pragma unroll
                    for (int k = 0; k < KB; ++k) in.nk[k] = own_nk[e][k];
                  }
                return in;
              }
            }
            if (in.act) {
              if constexpr (CATF) {
                if (KT > 0 || round > 0) {  // the row part of the factorised softmax, as the slot that started the tree left it
                  in.y = gload_d_off(a_cb, in.ro);
                  in.cls = (int)*(cls_c + r);
#pragma unroll
                  for (int k = 0; k < KB; ++k)
                    if (k < K) in.nk[k] = e_at(k, in.ro);  // (wave-uniform guard)
                } else {
                  // round 0 of the run-time-K instance: THIS launch writes the row parts (other workgroups, not
                  // visible yet): computed here from the same inputs (pgb_cat_row).  (The compile-time-K instances
                  // take their own rows from registers in this round, see above.)
                  const double yv0 = gload_d_off(gy_c, in.ro);
                  double Mx = 0.0;
#pragma unroll
                  for (int k = 0; k < KB; ++k)
                    if (k < K) {
                      const double nk = gload_d_off(noi_c[k], in.ro);
                      in.nk[k] = has_off ? nk + gload_d_off(off_c[k], in.ro) : nk;
                      if (k == 0 || in.nk[k] > Mx) Mx = in.nk[k];
                    }
                  in.cls = pgb_cat_class(K, yv0);
#pragma unroll
                  for (int k = 0; k < KB; ++k)
                    if (k < K) {
                      const double ak = in.nk[k] - Mx;
                      if (k == in.cls) in.y = ak;
                      in.nk[k] = pgb_exp_t(ak, tb.expt);
                    }
                }
              } else {
                in.y = gload_d_off(gy_c, in.ro);
#pragma unroll
                for (int k = 0; k < KB; ++k)
                  if (k < K) {  // (wave-uniform)
                    const double nk = gload_d_off(noi_c[k], in.ro);
                    in.nk[k] = has_off ? nk + gload_d_off(off_c[k], in.ro) : nk;
                  }
              }
            }
            return in;
          };
          uint32_t fb = 0;  // softmax: passes of this lane whose factorised sum was lost (see below): bit = pass
          // NB (the lane's own rows, softmax): the evaluation runs for every row slot and the result of a row that is
          // not in the leaf is dropped by a select -- the four row slots of a lane are then four INDEPENDENT chains
          // the compiler interleaves.  Under a branch per row slot each chain ran alone (exec-masked regions in
          // sequence: a dependent fma / LDS-read chain per slot with three waves per SIMD to hide it -- 20 % of
          // the wave-cycles issued a vector instruction).  The leaves that take this path hold most of the rows.
          auto eval_pass = [&](const PassIn& cur, int psi, auto nb_c) {
            constexpr bool NB = decltype(nb_c)::value;
            if (NB || cur.act) {
              double llv;
              if constexpr (CATF) {
                // softmax, constant leaves (pgb_loglik_cat_f): S = sum_k E_k w_k in output order, one fma each; the
                // child's d_c from the particle's LDS record by (side, class).  A dropped row predicts 0 from this
                // tree: w = exp(0 - 0) = 1 exactly, d = 0.
                double Ssum = 0.0;
#pragma unroll
                for (int k = 0; k < KB; ++k)
                  if (k < K) {
                    double wk;
                    if constexpr (KT > 0) wk = cur.side == 0 ? vLr[k] : vRr[k];  // (the children's w_k: see above)
                    else wk = cur.side == 0 ? uni(lj.w2[0][k]) : uni(lj.w2[1][k]);
                    if constexpr (DROPS)
                      if (cur.side == 2) wk = 1.0;
                    Ssum = k == 0 ? cur.nk[0] * wk : PGB_FMA(cur.nk[k], wk, Ssum);
                  }
                double dc = lj.d2[cur.side == 0 ? 0 : 1][cur.cls];
                if constexpr (DROPS)
                  if (cur.side == 2) dc = 0.0;
                // (a row of a SLOW child -- leaf values hundreds of units apart, pgb_cat_side -- is taken over by the
                //  unfactorised form AFTER the passes, see `fb`; here it only leaves its mark.  One copy of that
                //  code per instance, none of its registers in this loop.)
                llv = pgb_cat_value(cur.y, dc, Ssum, tb.logt);
                if (slowm != 0u) {  // (wave-uniform: this particle has a slow child)
                  if ((!NB || cur.act) && ((slowm >> (cur.side == 0 ? 0 : 1)) & 1u) != 0u && cur.side != 2) {
                    fb |= 1u << psi;
                    llv = 0.0;  // (quantises to 0: nothing is added for this row here)
                  }
                }
              } else {
                double mu[KB];
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                  mu[k] = 0.0;
                  if (k < K) {
                    double vk;
                    if constexpr (KT > 0) vk = cur.side == 0 ? vLr[k] : vRr[k];
                    else vk = k == 0 ? (cur.side == 0 ? vLr[0] : vRr[0])
                                     : (cur.side == 0 ? uni(lj.vLx[k > 0 ? k - 1 : 0]) : uni(lj.vRx[k > 0 ? k - 1 : 0]));
                    if constexpr (DROPS)
                      if (cur.side == 2) vk = 0.0;
                    mu[k] = cur.nk[k] + vk;
                  }
                }
                if constexpr (KT > 0) llv = loglik_mk<KT, FAMK>(S.family, K, cur.y, mu, &tb);
                else llv = loglik_arr<KB>(S.family, K, cur.y, mu, &tb);
              }
              long long q = quant_ll(llv, cl);
              if constexpr (NB) q = cur.act ? q : 0;
              vt += q;
              v0 += cur.side == 0 ? q : 0;
              if constexpr (DROPS) v2 += cur.side == 2 ? q : 0;
            }
          };
          bool done = false;
          if constexpr (OWN && CATF) {
            // the lane's own four rows: the row index is a compile-time constant in each copy, so that a pass takes
            // its row part straight from the registers (with a run-time pass index the selection of one row's K + 2
            // values out of four cost as many v_cndmask as the evaluation itself has arithmetic, now that it is K
            // fma and a logarithm)
            if (!dense) {  // (wave-uniform)
#pragma unroll
              for (int ps = 0; ps < RPT; ++ps) eval_pass(fetch(ps), ps, std::true_type{});
              done = true;
            }
          }
          if (!done) {
            PassIn cur = fetch(0);
            for (int ps = 0; ps < npass; ++ps) {
              PassIn nxt = cur;
              if (ps + 1 < npass) nxt = fetch(ps + 1);
              eval_pass(cur, ps, std::false_type{});
              cur = nxt;
            }
          }
          if constexpr (CATF) {
            if (__ballot(fb != 0u) != 0ull) {
              // the unfactorised softmax (pgb_loglik_cat_t's operations, one output at a time) on mu_k = eta_k + v_k
              // for the rows of a slow child.  Rare; the row, its side and its class are found again the way fetch()
              // found them.
              for (int ps = 0; ps < npass; ++ps)
                if ((fb >> ps) & 1u) {
                  uint32_t r, sd;
                  if (dense) {
                    const uint32_t ent = (uint32_t)s_lrow[w][lane + 64 * ps];
                    r = ent & 1023u;
                    sd = ent >> 10;
                  } else {
                    r = (uint32_t)(tid * RPT + ps);
                    const uint32_t nl = (nid >> (8 * ps)) & 255u;
                    sd = nl == lab ? 0u : (nl == nlab ? 1u : 2u);
                  }
                  const uint32_t ro = r * 8u;
                  auto muf = [&](int k) -> double {
                    const double nk = gload_d_off(noi_c[0] + (size_t)k * S.n_pad, ro);
                    const double ek = has_off ? nk + gload_d_off(off_c[0] + (size_t)k * S.n_pad, ro) : nk;
                    const double vk = sd == 0 ? (k == 0 ? lj.vL : lj.vLx[k > 0 ? k - 1 : 0])
                                    : sd == 1 ? (k == 0 ? lj.vR : lj.vRx[k > 0 ? k - 1 : 0]) : 0.0;
                    return ek + vk;
                  };
                  const double yv1 = gload_d_off(gy_c, ro);  // (its class index: loglik_fn clamps it like pgb_cat_class)
                  const long long q = quant_ll(loglik_fn(PGB_FAMILY_CATEGORICAL, S.K, yv1, muf, &tb), cl);  // (S.K: rolled loops)
                  vt += q;
                  v0 += sd == 0 ? q : 0;
                  if constexpr (DROPS) v2 += sd == 2 ? q : 0;
                }
            }
          }
          const int slot = (g - g0) * 3;
          if constexpr (DROPS) {
            const long long tot = wave_sum4(v0, vt - v0 - v2, v2, 0);  // lane l: total of value l & 3
            if (lane < 3 && tot != 0) atomicAdd((unsigned long long*)&s_red[slot + lane], (unsigned long long)tot);
          } else {
            const long long tot = wave_sum2(v0, vt - v0);  // lane l: total of value l & 1
            if (lane < 2 && tot != 0) atomicAdd((unsigned long long*)&s_red[slot + lane], (unsigned long long)tot);
          }
        };
        if (uni(lj.check_nan) != 0) passes(std::true_type{});
        else passes(std::false_type{});
        if (dense) __builtin_amdgcn_wave_barrier();  // the next particle's list goes into the same storage
      }
      TRLP(29);
      __syncthreads();
      for (int t = tid; t < (g1 - g0) * 3; t += BT) {
        const int gi = t / 3, i = t % 3;
        const long long s = s_red[t];
        s_red[t] = 0;
        if (s != 0) {
          AccL* a = &S.accl[((size_t)par * MAXP + s_job[g0 + gi].p) * LL_PER + (chunk & (LL_SLOTS - 1)) * LL_STRIDE];
          atomicAdd((unsigned long long*)(i == 0 ? &a->llL : i == 1 ? &a->llR : &a->llN), (unsigned long long)s);
        }
      }
      __syncthreads();
      TRL(30);
      continue;
    }
    if constexpr (MK) {  // run-time K (KT = 0): K = 5.. outputs and K-vector linear leaves
      static_assert(KT == 0 || MKPASS, "the compile-time-K instances take the pass loop above");
      for (int g = g0; g < g1; ++g) {
        const LJob& lj = s_job[g];
        const uint32_t ids = lj.src < 0 ? root_ids : *gcast<const uint32_t>(glid + lj.src + base);
        const uint32_t nid = *gcast<const uint32_t>(newl + (size_t)lj.p * S.n_pad + base);
        const uint32_t lab = (uint32_t)lj.label, nlab = (uint32_t)lj.new_label;
        long long v0 = 0, v1 = 0, v2 = 0;
        for (int e = 0; e < RPT; ++e) {
          if (((ids >> (8 * e)) & 255u) == lab) {
            const uint32_t nl = (nid >> (8 * e)) & 255u;
            const int side = nl == lab ? 0 : (nl == nlab ? 1 : 2);
            int sv = -1;
            double xv = 0.0, xb = 0.0;
            if constexpr (LIN) {
              sv = side == 0 ? lj.svarL : side == 1 ? lj.svarR : -1;
              if (sv >= 0) {
                xv = S.XT[lj.xoff + base + e];
                xb = side == 0 ? lj.xbarL : lj.xbarR;
              }
            }
            // predictor k of this row under this particle's split (leaf values from the LDS record)
            auto mu = [&](int k) -> double {
              double vk = k == 0 ? (side == 0 ? lj.vL : side == 1 ? lj.vR : 0.0)
                                 : (side == 0 ? lj.vLx[k - 1] : side == 1 ? lj.vRx[k - 1] : 0.0);
              if constexpr (LIN)
                if (sv >= 0)
                  vk = pgb_leaf_pred(vk, k == 0 ? (side == 0 ? lj.slopeL : lj.slopeR)
                                                : (side == 0 ? lj.sLx[k - 1] : lj.sRx[k - 1]), xb, xv);
              const double nk = k == 0 ? nv[e] : noi[(size_t)k * S.n_pad + base + e];
              return (S.has_off ? nk + goff[(size_t)k * S.n_pad + base + e] : nk) + vk;
            };
            const double llv = loglik_fn(S.family, K, yv[e], mu, &tb);
            const long long q = quant_ll(llv, cl);
            if (side == 0) v0 += q; else if (side == 1) v1 += q; else v2 += q;
          }
        }
        const int slot = (g - g0) * 3;
        const long long tot = wave_sum4(v0, v1, v2, 0);  // lane l: total of value l & 3
        if (lane < 3 && tot != 0) atomicAdd((unsigned long long*)&s_red[slot + lane], (unsigned long long)tot);
      }
      __syncthreads();
      for (int t = tid; t < (g1 - g0) * 3; t += BT) {
        const int gi = t / 3, i = t % 3;
        const long long s = s_red[t];
        s_red[t] = 0;
        if (s != 0) {
          AccL* a = &S.accl[((size_t)par * MAXP + s_job[g0 + gi].p) * LL_PER + (chunk & (LL_SLOTS - 1)) * LL_STRIDE];
          atomicAdd((unsigned long long*)(i == 0 ? &a->llL : i == 1 ? &a->llR : &a->llN), (unsigned long long)s);
        }
      }
      __syncthreads();
      continue;
    }
    // the label words of the next particle are requested before this one is evaluated
    const uint32_t base32 = (uint32_t)base;  // (n < 2^31)
    // (a job record is the same in every lane: offsets and labels go to scalar registers -- uni -- and the label
    //  words are fetched at scalar base + 32-bit row offset; per-lane 64-bit address arithmetic on these was a
    //  quarter of the pass's per-particle instructions)
    auto label_words = [&](const LJob& q, uint32_t& ids_o, uint32_t& nid_o) {
      const long long src_u = uni(q.src), dst_u = uni(q.dst);
      ids_o = src_u < 0 ? root_ids : gload_u32_off(glid + src_u, base32);
      nid_o = gload_u32_off(newl + dst_u, base32);
    };
    uint32_t ids_n, nid_n;
    label_words(s_job[g0], ids_n, nid_n);
    for (int g = g0; g < g1; ++g) {
      const LJob& lj = s_job[g];
      const uint32_t ids = ids_n, nid = nid_n;
      if (g + 1 < g1) {
        label_words(s_job[g + 1], ids_n, nid_n);
      }
      // (the particle's fields in registers: read through the LDS record they cost three LDS reads per ROW)
      const uint32_t lab = uni((uint32_t)lj.label), nlab = uni((uint32_t)lj.new_label);
      const double vL = lj.vL, vR = lj.vR;
      const bool drops = uni(lj.check_nan) != 0;  // only a column with missing values drops rows
      if constexpr (FAM == PGB_FAMILY_CALLBACK) {
        // the host evaluates this family (pgb_set_loglik_callback): hand it every row's side and the
        // linear predictor of the rows of the split leaf
        uint32_t sides = 0;
#pragma unroll
        for (int e = 0; e < RPT; ++e) {
          uint32_t side = 3;
          if (((ids >> (8 * e)) & 255u) == (uint32_t)lj.label) {
            const uint32_t nl = (nid >> (8 * e)) & 255u;
            side = nl == (uint32_t)lj.label ? 0u : (nl == (uint32_t)lj.new_label ? 1u : 2u);
            S.cb_mu[(size_t)lj.p * S.n_pad + base + e] = nv[e] + (side == 0 ? lj.vL : side == 1 ? lj.vR : 0.0);
          }
          sides |= side << (8 * e);
        }
        *(uint32_t*)(S.cb_side + (size_t)lj.p * S.n_pad + base) = sides;
        continue;
      }
      // One evaluation per row, two (three) running sums per lane: vt over every row of the split leaf, v0 over
      // the rows that stayed left (v2: dropped by a missing split value); the right child is what they leave.
      // Only a split on a column with missing values drops rows -- uniform per particle -- so the loop body is
      // compiled twice: without drops the side is ONE compare, the leaf value one select, and there is no third
      // sum (6 vector instructions fewer per evaluated row: profiles/r04_experiments.md).
      auto eval_rows = [&](auto drops_c) {
        constexpr bool DROPS = decltype(drops_c)::value;
        long long vt = 0, v0 = 0, v2 = 0;
        auto eval_one = [&](double nvr, uint32_t nl, double yr, uint32_t sgn_hi, int e_lin) {
          const bool left = nl == lab;
          double vleaf = left ? vL : vR;
          bool dropped = false;
          if constexpr (DROPS) {
            dropped = !left && nl != nlab;
            if (dropped) vleaf = 0.0;  // dropped: predicts 0
          }
          if constexpr (LIN) {
            const int side = left ? 0 : dropped ? 2 : 1;
            const int sv = side == 0 ? lj.svarL : side == 1 ? lj.svarR : -1;
            if (sv >= 0) {
              const double xv = S.XT[lj.xoff + base + e_lin];
              vleaf = pgb_leaf_pred(vleaf, side == 0 ? lj.slopeL : lj.slopeR, side == 0 ? lj.xbarL : lj.xbarR, xv);
            }
          }
          const double mu = nvr + vleaf;
          double llr;
          if constexpr (YBIT) {  // (Bernoulli: the response only flips the sign of the predictor)
            const double smu = pgb_u2d(pgb_d2u(mu) ^ ((unsigned long long)sgn_hi << 32));
            if constexpr (PROBIT) llr = lphi_lds(smu, s_lphi);  // (= pgb_loglik_bern_s: in [-2047, 1e-16] by itself)
            else llr = pgb_loglik_bern_s(FAM, smu, &tb);
          } else {
            llr = pgb_loglik1q(FAM >= 0 ? FAM : S.family, yr, mu, cn.inv_sigma2, cn.lik_param2, &tb);
          }
          const long long q = quant_ll(llr, cl);
          vt += q;
          v0 += left ? q : 0;
          if constexpr (DROPS) v2 += dropped ? q : 0;
        };
        bool dense_done = false;
        if constexpr (DENSE) {
          unsigned long long mk[RPT];
          int M = 0;
#pragma unroll
          for (int e = 0; e < RPT; ++e) {
            mk[e] = __ballot(((ids >> (8 * e)) & 255u) == lab);
            M += __popcll(mk[e]);
          }
          if (M <= LL_DENSE_MAX) {  // (wave-uniform)
            int off = 0;
#pragma unroll
            for (int e = 0; e < RPT; ++e) {
              if (((ids >> (8 * e)) & 255u) == lab) {
                const int pos = off + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk[e] >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo((unsigned)mk[e], 0u));
                const uint32_t nl = (nid >> (8 * e)) & 255u;
                s_lnv[w][pos] = nv[e];
                // the row's new label travels with it; Bernoulli: bit 8 = "y = 0" (flip the sign)
                if constexpr (YBIT) s_lfl[w][pos] = (uint16_t)(nl | (ysgn[e] >> 23));
                else { s_lfl[w][pos] = (uint16_t)nl; s_ly[w][pos] = yv[e]; }
              }
              off += __popcll(mk[e]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int k = lane; k < M; k += 64) {
              const uint32_t fl = s_lfl[w][k];
              eval_one(s_lnv[w][k], fl & 255u, YBIT ? 0.0 : s_ly[w][k], (fl & 256u) << 23, 0);
            }
            __builtin_amdgcn_wave_barrier();  // the next particle's list goes into the same storage
            dense_done = true;
          }
        }
        if (!dense_done) {
          if constexpr (DENSE) {
#pragma unroll
            for (int e = 0; e < RPT; ++e)
              if (((ids >> (8 * e)) & 255u) == lab) eval_one(nv[e], (nid >> (8 * e)) & 255u, yv[e], YBIT ? ysgn[e] : 0u, e);
          } else {  // (one evaluation site: these instances are bound by their registers, not by a loop counter)
#pragma unroll 1
            for (int e = 0; e < RPT; ++e)
              if (((ids >> (8 * e)) & 255u) == lab) eval_one(nv[e], (nid >> (8 * e)) & 255u, yv[e], 0u, e);
          }
        }
        const int slot = (g - g0) * 3;
        if constexpr (DROPS) {
          const long long tot = wave_sum4(v0, vt - v0 - v2, v2, 0);  // lane l: total of value l & 3
          if (lane < 3 && tot != 0) atomicAdd((unsigned long long*)&s_red[slot + lane], (unsigned long long)tot);
        } else {
          const long long tot = wave_sum2(v0, vt - v0);  // lane l: total of value l & 1
          if (lane < 2 && tot != 0) atomicAdd((unsigned long long*)&s_red[slot + lane], (unsigned long long)tot);
        }
      };
      // (the families with two exp / log chains per evaluation -- Poisson, NegativeBinomial, Gamma -- and the
      //  run-time-family instance keep ONE copy of the loop: a second one costs them their occupancy)
      if constexpr (DENSE) {
        if (drops) eval_rows(std::true_type{});
        else eval_rows(std::false_type{});
      } else {
        eval_rows(std::true_type{});
      }
    }
    if constexpr (FAM == PGB_FAMILY_CALLBACK) continue;  // nothing to reduce: the host sums
    __syncthreads();
    for (int t = tid; t < (g1 - g0) * 3; t += BT) {
      const int gi = t / 3, i = t % 3;
      const long long s = s_red[t];
      s_red[t] = 0;
      if (s != 0) {
        AccL* a = &S.accl[((size_t)par * MAXP + s_job[g0 + gi].p) * LL_PER + (chunk & (LL_SLOTS - 1)) * LL_STRIDE];
        atomicAdd((unsigned long long*)(i == 0 ? &a->llL : i == 1 ? &a->llR : &a->llN), (unsigned long long)s);
      }
    }
    __syncthreads();
  }
  }  // (nact != 0)
  // The INIT part of a slot that starts a tree comes LAST: nothing in this launch reads what it produces (the
  // stump / reference sums go to the next control kernel, the row parts to the later rounds), and in front of the
  // passes it stood -- as code -- between the job list and the first item of EVERY launch.
  TRL(31);
  if constexpr (FAM != PGB_FAMILY_CALLBACK) {
    // A slot that starts a tree ([U] init_particles): the log-likelihood of a fresh stump (C) and of the
    // tree as it stands, the reference particle (E0), over ALL rows -- from the {sum_trees, sum_trees_noi}
    // the INIT part of this slot's row pass just wrote.  Evaluated here, in the kernel that is compiled per
    // family / number of outputs, and not in the row pass: with the evaluation inlined twice the row pass
    // needed 163 (single output) / 256 (K = 4) VGPRs for a loop that does not use any of it.
    const int fam = FAM >= 0 ? FAM : S.family;
    if ((ch.kind & CMD_INIT) && fam != PGB_FAMILY_CALLBACK) {
      const int Kn = KT > 0 ? KT : S.K;
      const double* __restrict__ const noi0 = S.st + (size_t)cn.st_cur * Kn * S.n_pad;
      long long ce[2] = {0, 0};
      // K-vector leaves: units of BT rows, one row per thread -- a chunk is RPT units -- so that every workgroup of
      // the grid takes part (cfg5: 245 chunks over 768 workgroups left a third of them four rows per thread, three
      // exact evaluations each, before their share of the passes, and the launch waited for those)
      constexpr int ISUB = MK ? RPT : 1, IROWS = MK ? 1 : RPT;
      for (int unit = blockIdx.x; unit < S.nchunks * ISUB; unit += nwg) {
        const int chunk = unit / ISUB;
        const long long base = MK ? (long long)chunk * CH + (long long)(unit - chunk * ISUB) * BT + tid
                                  : (long long)chunk * CH + tid * RPT;
        // (single output: the four rows' inputs are requested together -- the arrays are padded to whole
        //  chunks -- instead of one dependent round trip per row and evaluation)
        double yrr[RPT], noir[RPT], str_[RPT], offr[RPT];
        const double init_leaf = S.init_leaf;
        if constexpr (!MK) {
          const double* __restrict__ const yp = S.y;
          const double2* __restrict__ const pk = S.pack;
          const double* __restrict__ const op = S.off;
          const bool ho = S.has_off != 0;
#pragma unroll
          for (int e = 0; e < RPT; ++e) {
            yrr[e] = yp[base + e];
            noir[e] = noi0[base + e];
            str_[e] = pk[base + e].x;
            offr[e] = ho ? op[base + e] : 0.0;  // (x + 0.0 == x bit for bit)
          }
        }
#pragma unroll
        for (int e = 0; e < IROWS; ++e) {
          const long long row = base + e;
          if (row >= S.n) continue;
          const double yr = MK ? S.y[row] : yrr[e];
          if constexpr (MK) {
            auto offk = [&](int k) { return S.has_off ? S.off[(size_t)k * S.n_pad + row] : 0.0; };  // (x + 0.0 == x)
            auto eta0 = [&](int k) { return noi0[(size_t)k * S.n_pad + row] + offk(k); };  // the row part of predictor k
            auto mu_stump = [&](int k) { return eta0(k) + S.init_leaf; };
            auto mu_cur = [&](int k) {
              return (k == 0 ? S.pack[row].x : S.packx[(size_t)(k > 0 ? k - 1 : 0) * S.n_pad + row]) + offk(k);
            };
            if constexpr (LIN && KT == 0) {  // (the rare instance keeps the predictors as functions, see loglik_fn)
              ce[0] += quant_ll(loglik_fn(S.family, Kn, yr, mu_stump, &tb), S.sc.cl);
              ce[1] += quant_ll(loglik_fn(S.family, Kn, yr, mu_cur, &tb), S.sc.cl);
            } else {
              double ms[KB], mc[KB];
              double et[CATF ? KB : 1];  // softmax, constant leaves: the row part eta_k of the predictors
#pragma unroll
              for (int k = 0; k < KB; ++k) {
                ms[k] = mc[k] = 0.0;
                if constexpr (CATF) et[k] = 0.0;
                if (k < Kn) {  // (wave-uniform; the loads behind the predictors go out together)
                  if constexpr (CATF) {
                    et[k] = eta0(k);
                    ms[k] = et[k] + S.init_leaf;  // (= mu_stump(k))
                  } else {
                    ms[k] = mu_stump(k);
                  }
                  mc[k] = mu_cur(k);
                }
              }
              if constexpr (CATF) {
                // ... and what the later rounds of this tree read instead of them (pgb_cat_row): E_k, a_c and the
                // class c.  (Round 0, in this very launch, computes the same values from the same inputs for its
                // own rows: other workgroups' stores are not visible yet.)
                double M = et[0];
#pragma unroll
                for (int k = 1; k < KB; ++k)
                  if (k < Kn && et[k] > M) M = et[k];
                const int c = pgb_cat_class(Kn, yr);
                double ac = 0.0, Esum = 0.0;
#pragma unroll
                for (int k = 0; k < KB; ++k)
                  if (k < Kn) {
                    const double ak = et[k] - M;
                    if (k == c) ac = ak;
                    const double Ek = pgb_exp_t(ak, tb.expt);
                    S.cat_e[(size_t)k * S.n_pad + row] = Ek;
                    Esum = k == 0 ? Ek * 1.0 : PGB_FMA(Ek, 1.0, Esum);  // (pgb_cat_sum with w = 1)
                  }
                S.cat_a[row] = ac;
                S.cat_c[row] = (uint8_t)c;
                // the stump in the factorised form (every output predicts init_leaf: d = 0, w = 1 exactly): no
                // exponentials of its own
                ce[0] += quant_ll(pgb_cat_value(ac, 0.0, Esum, tb.logt), S.sc.cl);
                if constexpr (KT > 0) ce[1] += quant_ll(loglik_mk<KT, FAMK>(S.family, Kn, yr, mc, &tb), S.sc.cl);
                else ce[1] += quant_ll(loglik_arr<KB>(S.family, Kn, yr, mc, &tb), S.sc.cl);
              } else if constexpr (KT > 0) {
                ce[0] += quant_ll(loglik_mk<KT, FAMK>(S.family, Kn, yr, ms, &tb), S.sc.cl);
                ce[1] += quant_ll(loglik_mk<KT, FAMK>(S.family, Kn, yr, mc, &tb), S.sc.cl);
              } else {
                ce[0] += quant_ll(loglik_arr<KB>(S.family, Kn, yr, ms, &tb), S.sc.cl);
                ce[1] += quant_ll(loglik_arr<KB>(S.family, Kn, yr, mc, &tb), S.sc.cl);
              }
            }
          } else {
            const double offv = offr[e];
            auto ll1 = [&](double mu1) -> double {
              if constexpr (PROBIT) return lphi_lds(yr > 0.5 ? mu1 : -mu1, s_lphi);  // (the staged layout)
              else return pgb_loglik1q(fam, yr, mu1, cn.inv_sigma2, cn.lik_param2, &tb);
            };
            ce[0] += quant_ll(ll1((noir[e] + offv) + init_leaf), S.sc.cl);
            ce[1] += quant_ll(ll1(str_[e] + offv), S.sc.cl);
          }
        }
      }
      block_sum<2>(ce, s_red);
      if (tid == 0) {
        InitAcc* a = &S.initacc[(size_t)par * IA_SLOTS + (blockIdx.x % IA_SLOTS)];
        if (ce[0]) atomicAdd((unsigned long long*)&a->C, (unsigned long long)ce[0]);
        if (ce[1]) atomicAdd((unsigned long long*)&a->E0, (unsigned long long)ce[1]);
      }
    }
  }
  if (sat) atomicAdd(&S.counters[4], (unsigned long long)sat);
}

