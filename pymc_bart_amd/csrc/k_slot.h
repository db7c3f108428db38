// k_slot.h -- part of pgbart_hip.hip (not a standalone header): k_slot: one SMC round in ONE launch.
// ------------------------------------------------------------------ k_slot
// The two-kernel slot {k_ctrl ; k_rows} pays two dependent kernel boundaries (~2.3 us each on
// MI355X) and a command / job hand-off through memory per SMC round; at n = 100k they are a third
// of the round.  k_slot is the same round as ONE launch for the configuration the headline metric
// is quoted on (Normal likelihood, one output, constant leaves):
//
//   * EVERY workgroup of the row grid first finishes the previous round for all particles --
//     statistics -> leaf values -> weights -> resampling, one particle per lane of wave 0, exactly
//     the arithmetic of k_ctrl -- while waves 1..3 draw the proposals' random numbers; the result
//     (ancestors, popped nodes, attempts) lives in LDS, so there is nothing to hand over;
//   * it then makes the growth proposal only for the <= GMAXF particles of ITS work item (split
//     variable, exact k-th-row selection) and streams its 1024 rows for them (the row pass of
//     k_rows, unchanged);
//   * the bookkeeping that has to be written exactly once (node tables of the new particles, job
//     records, the accepted tree, the control word) is spread over the workgroups as "duties".
//
// Redundant control work costs nothing here: the chain is latency-bound and the SIMDs would idle.
// Statistics are kept in a ring of THREE buffers (read: previous slot, write: this slot, zero: next
// slot), because nobody can clear a buffer between "every workgroup has read it" and "the first
// workgroup adds to it" without a grid-wide barrier.  Results are bit-identical to the two-kernel
// path and to the oracle: same numeric contract, same addressed draws, integer row sums.
#define GMAXF 16    /* particles per work item */
#ifndef SLOT_TEAMS
#define SLOT_TEAMS 3 /* 256-thread row teams per workgroup; ONE workgroup per CU runs the control phase once */
#endif
#define SLOT_BT (SLOT_TEAMS * BT)
#define CDF_LDS 256 /* split-variable prefix sums are staged in LDS: p <= CDF_LDS */

struct FinS {  // an old particle after its pending split (Normal family, constant leaves)
  int ok, cL, cR;
  int nn_old, n_nodes, n_leaves, next_pop;
  int loc_gen, loc_slot;
  int node, var, new_label, ccL, ccR;
  int depth, label;
  long long aL, aR, bL, bR, c2L, c2R;
  double split, vL, vR, sseL, sseR, sse_tot, sse_orph;
};
struct Prop {  // the proposal of new particle p, minus the split value
  int anc, node, attempt, haswork, copy_if_idle, var, cnt, cc_row, label, depth;
  int src_gen, src_slot, n_nodes, n_leaves, next_pop, pos;  // header after the pop; pos: index in the work list
  long long q_st, q_r, q_r2;
  double sse, value, sse_tot, sse_orph;
};
struct SJob {  // a work-list entry of this workgroup's group, after the split-row selection
  long long src, xoff;
  double v;
  int32_t p, active, copy, check_nan, rule, label, new_label, ccL, ccR, pad;
};

// first j with thr <= (double)S[j] (p - 1 if none): the prefix sums are integers and increase, so
// the bisection returns what the linear scan of pgb_sample_var returns
__device__ __forceinline__ int sample_var_bisect(const long long* Sarr, int p, double u) {
  const double thr = u * (double)Sarr[p - 1];
  int lo = 0, hi = p - 1;  // answer in [lo, hi]
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (thr <= (double)Sarr[mid]) hi = mid;
    else lo = mid + 1;
  }
  return lo;
}

// new particle's node table := ancestor's table with the pending split applied (all threads of the
// workgroup; the same patches as k_ctrl)
__device__ __forceinline__ void copy_patched(DNode* __restrict__ dst, const DNode* __restrict__ src, const FinS& f,
                                             bool r1, const InitAcc& ia, double root_sse, int ttid) {
  const int nn = f.nn_old;
  for (int i = ttid; i < nn; i += BT) {
    DNode z = src[i];
    if (r1 && i == 0) {
      z.q_st = ia.A;
      z.q_r = ia.B;
      z.q_r2 = ia.C;
      z.sse = root_sse;
    }
    if (f.ok == 1 && i == f.node) {
      z.var = f.var;
      z.split = f.split;
      z.left = (uint8_t)nn;
      z.right = (uint8_t)(nn + 1);
    } else if (f.ok == -1 && i == f.node) {
      z.cnt = f.cL;
      z.q_st = f.aL;
      z.q_r = f.bL;
      z.q_r2 = f.c2L;
      z.sse = f.sseL;
      z.cc_row = f.ccL;
    }
    dst[i] = z;
  }
  if (f.ok == 1 && ttid >= BT - 2) {
    const bool isL = ttid == BT - 2;
    DNode z;
    memset(&z, 0, sizeof z);
    z.var = -1;
    z.depth = (uint8_t)(f.depth + 1);
    z.label = isL ? (uint8_t)f.label : (uint8_t)f.new_label;
    z.cnt = isL ? f.cL : f.cR;
    z.q_st = isL ? f.aL : f.aR;
    z.q_r = isL ? f.bL : f.bR;
    z.q_r2 = isL ? f.c2L : f.c2R;
    z.value = isL ? f.vL : f.vR;
    z.sse = isL ? f.sseL : f.sseR;
    z.cc_row = isL ? f.ccL : f.ccR;
    dst[nn + (isL ? 0 : 1)] = z;
  }
}

// [U] get_split_value on ONE wave: the k-th row (ascending) of the popped leaf, k = floor(u cnt),
// redrawn while the split column is missing there.  Returns found; *v_out the split value.
__device__ __forceinline__ int select_split_value(const Dev& S, const Prop& pr, int p, uint32_t itp, uint32_t rr,
                                                  double u_try0, double u1_try0, double* v_out) {
  const int j = pr.var;
  const double* xc = S.XT + (size_t)j * S.n_pad;
  const bool subset_rule = S.rules[j] == PGB_RULE_SUBSET;
  const uint8_t* lid = pr.src_slot >= 0 ? S.lid + ((size_t)pr.src_gen * MAXP + pr.src_slot) * S.n_pad : nullptr;
  const uint16_t* ccr = pr.cc_row >= 0 ? S.cc + (size_t)pr.cc_row * S.nchunks : nullptr;
  const int ncnt = pr.cnt, nlabel = pr.label;
  const int per = (S.nchunks + 63) / 64;
  const int c0 = lane_id() * per;
  int c1 = c0 + per;
  if (c1 > S.nchunks) c1 = S.nchunks;
  int part = 0, pre = 0;
  if (lid != nullptr) {
    for (int cc = c0; cc < c1; ++cc) part += ccr[cc];
    pre = wave_incl_scan(part) - part;
  }
  int found = 0;
  double v = 0.0;
  for (uint32_t tr = 0; tr < PGB_SELECT_TRIES && !found; ++tr) {
    double us = u_try0, us1 = u1_try0;
    if (tr > 0) {
      const pgb_u2 ud = pgb_draw2(S.seed, itp, rr, (uint32_t)p, PGB_RNG_SELECT, tr);
      us = ud.u0;
      us1 = ud.u1;
    }
    long long k = (long long)(us * (double)ncnt);
    if (k > ncnt - 1) k = ncnt - 1;
    long long row;
    if (lid == nullptr) {
      row = k;  // untouched root: every row belongs to it
    } else {
      const bool own = (long long)pre <= k && k < (long long)pre + part;
      int cstar = 0, kk = 0;
      if (own) {
        kk = (int)(k - pre);
        cstar = c0;
        while (kk >= ccr[cstar]) {
          kk -= ccr[cstar];
          ++cstar;
        }
      }
      const int ol = (int)__ffsll((long long)__ballot(own)) - 1;
      cstar = __builtin_amdgcn_readlane(cstar, ol);
      kk = __builtin_amdgcn_readlane(kk, ol);
      const uint4 ids = *(const uint4*)(lid + (size_t)cstar * CH + lane_id() * 16);
      const uint32_t wds[4] = {ids.x, ids.y, ids.z, ids.w};
      int mcnt = 0;
#pragma unroll
      for (int wd = 0; wd < 4; ++wd)
#pragma unroll
        for (int e = 0; e < 4; ++e) mcnt += (((wds[wd] >> (8 * e)) & 255u) == (uint32_t)nlabel);
      const int pre2 = wave_incl_scan(mcnt) - mcnt;
      const bool own2 = pre2 <= kk && kk < pre2 + mcnt;
      int off = 0;
      if (own2) {
        int rem = kk - pre2;
        for (int bb = 0; bb < 16; ++bb) {
          if (((wds[bb >> 2] >> (8 * (bb & 3))) & 255u) == (uint32_t)nlabel) {
            if (rem == 0) {
              off = bb;
              break;
            }
            --rem;
          }
        }
      }
      const int ol2 = (int)__ffsll((long long)__ballot(own2)) - 1;
      off = __builtin_amdgcn_readlane(off, ol2);
      row = (long long)cstar * CH + ol2 * 16 + off;
    }
    const double x = xc[row];
    found = (x == x) ? 1 : 0;
    v = x;
    if (found && subset_rule) v = pgb_subset_value(us1, x);
  }
  *v_out = v;
  return found;
}

template <bool SUB>
__global__ __launch_bounds__(SLOT_BT, 1) void k_slot(const Dev* __restrict__ Sp, int par, int s3, Ctrl* __restrict__ ctrls) {
  const Dev& S = *Sp;
  constexpr int NRED = 7;
  constexpr int NW = SLOT_BT / 64;  // waves per workgroup
  __shared__ FinS s_fin[MAXP];
  __shared__ DNode s_pop[MAXP];  // node each OLD particle would pop next
  __shared__ Prop s_prop[MAXP];
  __shared__ SJob s_job[MAXP];   // indexed by position in the work list
  __shared__ long long s_red[SLOT_TEAMS][GMAXF * NRED * 4];
  __shared__ double s_lv[2][256];  // label -> leaf value: [0] accepted tree (FINAL), [1] next tree (INIT)
  __shared__ double s_prior[PGB_MAX_DEPTH];
  __shared__ double s_coin[2][MAXP], s_uvar[2][MAXP], s_usel[2][MAXP], s_usel1[2][MAXP];
  __shared__ int s_var[2][MAXP];
  __shared__ long long s_cdf[2][CDF_LDS];
  __shared__ int s_anc[MAXP], s_list[MAXP];
  __shared__ int s_i[8];
  __shared__ unsigned long long s_duty[SLOT_TEAMS];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.x;
  const int team = tid / BT, ttid = tid % BT, tw = ttid >> 6;  // row team, thread / wave inside the team
  const int P = S.P, Lc = P - 1;
  const int rR = (s3 + 2) % 3, rW = s3, rZ = (s3 + 1) % 3;  // statistics ring: read / write / zero
  const Ctrl c = load_uniform(&ctrls[par]);
  Ctrl* co = &ctrls[par ^ 1];
  InitAcc ia;
  {
    const InitAcc* src = S.initacc + (size_t)rR * IA_SLOTS;
    ia = load_uniform(&src[0]);
#pragma unroll
    for (int k = 1; k < IA_SLOTS; ++k) {
      const InitAcc t = load_uniform(&src[k]);
      ia.A += t.A; ia.B += t.B; ia.C += t.C; ia.E0 += t.E0; ia.QSTD += t.QSTD;
    }
  }
  long long* pstamp = nullptr;
  if (S.prof_stamps != nullptr && tid == 0 && b < PROF_BLOCKS) {
    pstamp = S.prof_stamps + ((size_t)(c.slot_no % PROF_RING) * PROF_BLOCKS + b) * 2;
    pstamp[0] = wall_clock64();
    pstamp[1] = pstamp[0];
  }
#define PROF_END() do { if (pstamp) pstamp[1] = wall_clock64(); } while (0)
#ifdef PGB_TRACE
#define TRS(i) do { if (b == 1 && tid == 0) S.trace[(size_t)(c.slot_no % TRACE_SLOTS) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define TRS(i) ((void)0)
#endif
  TRS(0);

  double leaf_sd = c.leaf_sd;
  if (c.pend_leafsd && c.pend_iter > 2) leaf_sd = ((double)ia.QSTD * S.sc.inv_c1) / (double)S.n;

  // the buffers the NEXT slot accumulates into
  for (int p = 1 + b; p < P; p += gridDim.x)
    if (tid < ACC_SLOTS) {
      Acc z;
      memset(&z, 0, sizeof z);
      S.acc[((size_t)rZ * MAXP + p) * ACC_PER + tid * ACC_STRIDE] = z;
    }
  if (b == 0 && tid < IA_SLOTS) S.initacc[(size_t)rZ * IA_SLOTS + tid] = InitAcc{0, 0, 0, 0, 0, 0, 0, 0};

  if (c.phase == PH_IDLE) {
    if (b == 0 && tid == 0) {
      Ctrl o = c;
      o.slot_no = c.slot_no + 1;
      o.leaf_sd = leaf_sd;
      o.pend_leafsd = 0;
      *co = o;
    }
    PROF_END();
    return;
  }
  if (b == 0 && tid == 0) atomicAdd(&S.counters[5], 1ull);  // slots that did work

  const bool begin = c.phase == PH_BEGIN;  // first tree of a step: nothing to finish
  const int r = c.round;
  const uint32_t it = (uint32_t)c.iter;
  const DPart* OT = S.parts + (size_t)par * MAXP;
  DPart* NT = S.parts + (size_t)(par ^ 1) * MAXP;
  const Job* JP = S.jobs + (size_t)(par ^ 1) * MAXP;
  Job* JN = S.jobs + (size_t)par * MAXP;
  const bool rebuild = !begin && c.tune && c.iter > S.m;
  const long long nchunks = S.nchunks;

  // tree-level bookkeeping IF this slot ends the tree (known from the control word alone)
  const int tree_old = c.lower + c.k;
  const bool more = (c.k + 1 < c.batch_n);
  const bool next_step = (!more && c.steps_left > 1);
  int lower_next = c.lower, k_next = c.k + 1, batch_next = c.batch_n;
  if (!more) {
    const int upper = c.lower + c.batch_n;
    lower_next = upper < S.m ? upper : 0;
    k_next = 0;
    const int bs = c.tune ? S.batch_tune : S.batch_draw;
    int up2 = lower_next + bs;
    if (up2 > S.m) up2 = S.m;
    batch_next = up2 - lower_next;
  }
  const int tree_next = begin ? tree_old : lower_next + k_next;  // the tree a fresh proposal starts

  // ---- staging, consumed after the first barrier: prior table, split-variable prefix sums, empty label
  //      tables; the LAST team fetches the nodes the label tables may be built from
  if (tid < PGB_MAX_DEPTH) s_prior[tid] = S.prior_leaf[tid];
  {
    const long long* cdfS = S.cdfS + (size_t)c.cdf_cur * S.p;
    for (int j = tid; j < S.p; j += SLOT_BT) s_cdf[0][j] = cdfS[j];
  }
  if (tid < 256) {
    s_lv[0][tid] = 0.0;
    s_lv[1][tid] = 0.0;
  }
  // node `ttid` of the current tree (kept if the reference particle wins) and of the next tree
  int ko_var = 0, ko_label = 0, kn_var = 0, kn_label = 0;
  double ko_value = 0.0, kn_value = 0.0;
  bool ko_has = false, kn_has = false;
  if (team == SLOT_TEAMS - 1) {
    if (!begin && ttid < S.trees[tree_old].n_nodes) {
      const DNode z = S.trees[tree_old].nd[ttid];
      ko_var = z.var; ko_label = z.label; ko_value = z.value; ko_has = true;
    }
    if (ttid < S.trees[tree_next].n_nodes) {
      const DNode z = S.trees[tree_next].nd[ttid];
      kn_var = z.var; kn_label = z.label; kn_value = z.value; kn_has = true;
    }
  }
  // rows of the team's first work item in a plain round ({sum_trees, r}: independent of everything above)
  double2 pre_pack[RPT];
  const bool have_pre = c.phase == PH_ROUND;
  {
    const long long base0 = (long long)(((long long)b * SLOT_TEAMS + team) % nchunks) * CH + ttid * RPT;
#pragma unroll
    for (int e = 0; e < RPT; ++e) pre_pack[e] = have_pre ? S.pack[base0 + e] : make_double2(0.0, 0.0);
  }

  bool stop = false;
  int sel = 0;
  double sse0 = c.sse0;
  const bool r1 = r == 1;
  const double root_sse = pgb_leaf_sse(S.n, ia.B, ia.C, S.init_leaf, S.sc.inv_c1, S.sc.inv_c2);

  // =================================================================== finish + draws
  if (w == 0) {
    if (!begin) {
      if (r1) sse0 = (double)ia.E0 * S.sc.inv_c2;
      const int q = tid;
      const bool isp = q >= 1 && q < P;
      double lw = 0.0;
      bool pending = false;
      Job j;
      Acc a;
      DNode popn;
      memset(&popn, 0, sizeof popn);
      if (isp) {
        j = JP[q];
        a = load_acc(&S.acc[((size_t)rR * MAXP + q) * ACC_PER]);
        if (j.h_next_pop < j.h_n_nodes) popn = OT[q].nd[j.h_next_pop];
      }
      double z0, z1, u_res;
      {
        const pgb_u2 ul = pgb_draw2(S.seed, it, (uint32_t)(r - 1), (uint32_t)q,
                                    q == 0 ? PGB_RNG_RESAMPLE : PGB_RNG_LEAF, 0);
        u_res = readlane_d(ul.u0, 0);
        pgb_normal2(ul.u0, ul.u1, &z0, &z1);
      }
      if (isp) {
        FinS& f = s_fin[q];
        if (r1) {  // round-0 jobs were written before the root statistics existed
          j.p_q_st = ia.A;
          j.p_q_r = ia.B;
          j.p_q_r2 = ia.C;
          j.p_sse = root_sse;
          j.h_sse_tot = root_sse;
          j.h_sse_orph = 0.0;
        }
        f.ok = 0;
        f.nn_old = j.h_n_nodes;
        f.n_nodes = j.h_n_nodes;
        f.n_leaves = j.h_n_leaves;
        f.next_pop = j.h_next_pop;
        f.sse_tot = j.h_sse_tot;
        f.sse_orph = j.h_sse_orph;
        f.loc_gen = j.src_gen;
        f.loc_slot = j.src_slot;
        if (j.copy) {
          f.loc_gen = c.lid_gen;
          f.loc_slot = q;
        }
        if (j.active) {
          const ChildVals cv = child_values(S, j.rule, j.cnt, j.p_q_st, j.p_value, a.cnts, a.aL, a.aN, z0, z1, leaf_sd);
          const int cL = cv.cL, cR = cv.cR;
          f.loc_gen = c.lid_gen;
          f.loc_slot = q;
          f.ok = cv.ok;
          f.node = j.node;
          f.cL = cL;
          f.aL = a.aL; f.bL = a.bL; f.c2L = a.c2L;
          f.ccL = j.ccL;
          f.sse_orph = j.h_sse_orph + (double)a.c2N * S.sc.inv_c2;
          if (cv.ok == -1) {
            f.sseL = pgb_leaf_sse(cL, f.bL, f.c2L, j.p_value, S.sc.inv_c1, S.sc.inv_c2);
            f.sse_tot = (j.h_sse_tot - j.p_sse) + f.sseL;
          } else {
            f.cR = cR;
            f.var = j.var; f.split = j.v; f.new_label = j.new_label;
            f.ccR = j.ccR;
            f.depth = j.p_depth; f.label = j.label;
            f.aR = cv.aR;
            f.bR = j.p_q_r - a.bL - a.bN;
            f.c2R = j.p_q_r2 - a.c2L - a.c2N;
            f.vL = cv.vL;
            f.vR = cv.vR;
            f.sseL = pgb_leaf_sse(cL, f.bL, f.c2L, f.vL, S.sc.inv_c1, S.sc.inv_c2);
            f.sseR = pgb_leaf_sse(cR, f.bR, f.c2R, f.vR, S.sc.inv_c1, S.sc.inv_c2);
            f.sse_tot = ((j.h_sse_tot - j.p_sse) + f.sseL) + f.sseR;
            f.n_nodes = j.h_n_nodes + 2;
            f.n_leaves = j.h_n_leaves + 1;
          }
        }
        s_pop[q] = popn;
        pending = f.next_pop < f.n_nodes;
        lw = (f.sse_tot + f.sse_orph) * (-0.5 * c.inv_sigma2);
      }
      stop = __ballot(pending) == 0ull;
      // cumulative weights in the contract's scan order (pgb_weights_scan): lanes [first, first + cnt)
      const int first = stop ? 0 : 1, cnt = stop ? P : Lc;
      if (stop && q == 0) lw = sse0 * (-0.5 * c.inv_sigma2);  // the reference particle
      const bool act = q >= first && q < first + cnt;
      const double mx = wave_max_d(act ? lw : -1.0e308);
      double W = act ? pgb_exp(lw - mx) + 1e-12 : 0.0;
#define PGB_SCAN_STEP(ctrl_, rm_)                                                          \
  {                                                                                        \
    const int tl = __builtin_amdgcn_update_dpp(0, __double2loint(W), ctrl_, rm_, 0xf, 0);  \
    const int th = __builtin_amdgcn_update_dpp(0, __double2hiint(W), ctrl_, rm_, 0xf, 0);  \
    W = W + __hiloint2double(th, tl);                                                      \
  }
      PGB_SCAN_STEP(0x111, 0xf)
      PGB_SCAN_STEP(0x112, 0xf)
      PGB_SCAN_STEP(0x114, 0xf)
      PGB_SCAN_STEP(0x118, 0xf)
      PGB_SCAN_STEP(0x142, 0xa)
      PGB_SCAN_STEP(0x143, 0xc)
#undef PGB_SCAN_STEP
      TRS(2);
      const int last = first + cnt - 1;
      const double total = readlane_d(W, last);
      // [U] systematic resampling: ancestor of every new particle (lane = new particle), or the final
      // choice among all P particles; pgb_pick: first i in [first, last) with !(u total > W[i]).
      // The cumulative weight of lane i is broadcast with v_readlane: no LDS round trips.
      double u_mine;
      if (!stop) u_mine = (u_res + (double)(q - 1)) / (double)Lc;
      else u_mine = pgb_draw2(S.seed, it, 0, 0, PGB_RNG_FINAL, 0).u0;
      const double thr = u_mine * total;
      int pick = last;
      for (int i = last - 1; i >= first; --i) {
        const double Wi = readlane_d(W, i);
        if (!(thr > Wi)) pick = i;
      }
      if (isp) s_anc[q] = pick;
      if (tid == 0) {
        s_i[0] = stop ? 1 : 0;
        s_i[1] = pick;  // lane 0's pick is the final choice when the tree ends
      }
    }
  } else if (w <= 2) {
    // proposals' draws for every new particle p = lane: set 0 = round r of this tree, set 1 = round 0
    // of the next tree; coin + split variable, and the first split-row draw
    const int set = w - 1, p = lane;
    if (p >= 1 && p < P) {
      const uint32_t itp = set ? it + 1u : it, rr = set ? 0u : (uint32_t)r;
      const pgb_u2 u = pgb_draw2(S.seed, itp, rr, (uint32_t)p, PGB_RNG_PROPOSE, 0);
      const pgb_u2 us = pgb_draw2(S.seed, itp, rr, (uint32_t)p, PGB_RNG_SELECT, 0);
      s_coin[set][p] = u.u0;
      s_uvar[set][p] = u.u1;  // the variable itself is drawn after the barrier (set 1 may need the rebuilt sampler)
      s_usel[set][p] = us.u0;
      s_usel1[set][p] = us.u1;
    }
  } else if (w == 3) {
    // the sampler of the NEXT tree is rebuilt from the weights when this tree ends while tuning
    if (rebuild) {
      const long long* alpha = S.alpha + (size_t)c.alpha_cur * S.p;
      long long carry = 0;
      for (int base = 0; base < S.p; base += 64) {
        const int j = base + lane;
        const long long run = wave_sum_dpp(j < S.p ? alpha[j] : 0) + carry;
        if (j < S.p) s_cdf[1][j] = run;
        carry = ((long long)__builtin_amdgcn_readlane((int)(run >> 32), 63) << 32) |
                (unsigned)__builtin_amdgcn_readlane((int)run, 63);
      }
    }
  }
  TRS(3);
  __syncthreads();
  TRS(4);
  if (!begin) {
    stop = s_i[0] != 0;
    sel = stop ? s_i[1] : 0;
  }
  // split variables ([U] SampleSplittingVariable.rvs): waves 1, 2, lane = new particle
  if (w >= 1 && w <= 2) {
    const int set = w - 1, p = lane;
    if (p >= 1 && p < P) {
      const long long* cdf = (set && rebuild) ? s_cdf[1] : s_cdf[0];
      s_var[set][p] = sample_var_bisect(cdf, S.p, s_uvar[set][p]);
    }
  }

  const bool has_init = begin || (stop && (more || next_step));
  const bool fresh = has_init;
  const bool do_final = stop, do_init = has_init;
  const bool lone_final = stop && !has_init;
  const int tree_new = tree_next;
  const int set = fresh ? 1 : 0;
  const int rr = fresh ? 0 : r;
  const uint32_t itp = fresh ? it + 1u : it;
  const int dst_gen = (c.lid_gen + 1) % NGEN;

  // ---- label -> value tables of the accepted tree and of the next tree
  if (do_final) {
    if (sel == 0) {
      if (ko_has && ko_var < 0) s_lv[0][ko_label] = ko_value;
    } else {
      const FinS& F = s_fin[sel];
      const DNode* src = OT[sel].nd;
      for (int i = tid; i < F.nn_old; i += SLOT_BT) {
        const DNode z = src[i];
        const bool split_now = F.ok == 1 && i == F.node;
        if (z.var < 0 && !split_now) s_lv[0][z.label] = z.value;
      }
      if (F.ok == 1 && tid == 0) {
        s_lv[0][F.label] = F.vL;
        s_lv[0][F.new_label] = F.vR;
      }
    }
  }
  if (do_init && !(do_final && tree_new == tree_old) && kn_has && kn_var < 0) s_lv[1][kn_label] = kn_value;
  __syncthreads();
  if (do_init && do_final && tree_new == tree_old && tid < 256) s_lv[1][tid] = s_lv[0][tid];  // m == 1: the tree just accepted

  // =================================================================== proposals (wave 0, lane = new particle)
  TRS(5);
  if (w == 0 && !lone_final) {
    const int p = lane;
    bool haswork = false;
    Prop pr;
    memset(&pr, 0, sizeof pr);
    pr.node = -1;
    pr.pos = -1;
    const bool isp = p >= 1 && p < P;
    if (isp) {
      if (fresh) {
        pr.anc = 0;
        pr.src_gen = 0;
        pr.src_slot = -1;
        pr.n_nodes = 1;
        pr.n_leaves = 1;
        pr.next_pop = 1;
        pr.node = 0;
        pr.cnt = (int32_t)S.n;
        pr.cc_row = -1;
        pr.value = S.init_leaf;
      } else {
        const int anc = s_anc[p];
        const FinS& F = s_fin[anc];
        pr.anc = anc;
        pr.src_gen = F.loc_gen;
        pr.src_slot = F.loc_slot;
        pr.n_nodes = F.n_nodes;
        pr.n_leaves = F.n_leaves;
        pr.next_pop = F.next_pop;
        pr.sse_tot = F.sse_tot;
        pr.sse_orph = F.sse_orph;
        const int np = F.next_pop;
        if (np < F.n_nodes) {
          pr.node = np;
          pr.next_pop = np + 1;
          if (np < F.nn_old) {
            const DNode& nd = s_pop[anc];
            pr.depth = nd.depth; pr.label = nd.label; pr.cnt = nd.cnt; pr.cc_row = nd.cc_row;
            pr.q_st = nd.q_st; pr.q_r = nd.q_r; pr.q_r2 = nd.q_r2; pr.sse = nd.sse; pr.value = nd.value;
            if (r1 && np == 0) {
              pr.q_st = ia.A; pr.q_r = ia.B; pr.q_r2 = ia.C; pr.sse = root_sse;
            }
          } else {
            const bool isL = np == F.nn_old;
            pr.depth = F.depth + 1;
            pr.label = isL ? F.label : F.new_label;
            pr.cnt = isL ? F.cL : F.cR;
            pr.q_st = isL ? F.aL : F.aR;
            pr.q_r = isL ? F.bL : F.bR;
            pr.q_r2 = isL ? F.c2L : F.c2R;
            pr.sse = isL ? F.sseL : F.sseR;
            pr.value = isL ? F.vL : F.vR;
            pr.cc_row = isL ? F.ccL : F.ccR;
          }
        }
      }
      if (pr.node >= 0) {
        const double pl = pr.depth < PGB_MAX_DEPTH ? s_prior[pr.depth] : 1.0;
        pr.attempt = ((pl < s_coin[set][p]) && (pr.n_nodes + 2 <= MAXN) && (pr.cnt >= 2)) ? 1 : 0;
      }
      pr.var = s_var[set][p];
      // labels are only rewritten when a particle splits; an idle particle is copied forward only when
      // its generation is the next to be reused
      pr.copy_if_idle = (pr.src_slot >= 0 && pr.src_gen == (dst_gen + 1) % NGEN) ? 1 : 0;
      haswork = pr.attempt || pr.copy_if_idle;
      pr.haswork = haswork ? 1 : 0;
    }
    const unsigned long long m = __ballot(haswork);
    if (haswork) {
      pr.pos = __popcll(m & ((1ull << p) - 1ull));
      s_list[pr.pos] = p;
    }
    if (isp) s_prop[p] = pr;
    if (p == 0) s_i[2] = __popcll(m);
  }
  __syncthreads();
  TRS(6);

  // =================================================================== rows
  uint8_t* tl_old = do_final ? S.tree_lid + (size_t)tree_old * S.n_pad : nullptr;
  const uint8_t* tl_new = do_init ? S.tree_lid + (size_t)tree_new * S.n_pad : nullptr;
  int sel_slot = -2, sel_gen = 0;
  if (do_final && sel >= 1) {
    sel_slot = s_fin[sel].loc_slot;
    sel_gen = s_fin[sel].loc_gen;
  }
  const uint8_t* sel_lid = (do_final && sel_slot >= 0) ? S.lid + ((size_t)sel_gen * MAXP + sel_slot) * S.n_pad : nullptr;
  const long long rs_count = c.rs_count + (c.tune ? 1 : 0);
  const double cntf = (double)rs_count;
  double* const st_in = S.st + (size_t)c.st_cur * S.n_pad;
  double* const st_out = S.st + (size_t)(do_init ? c.st_cur ^ 1 : c.st_cur) * S.n_pad;
  const double c1 = S.sc.c1, c2 = S.sc.c2;
  const long long n = S.n, n_pad = S.n_pad;

  if (!lone_final) {
    const int nact = s_i[2];
    const int target = do_init ? S.rows_target_init : S.rows_target;
    int G = (int)((nact * nchunks + target - 1) / target);
    if (G < 1) G = 1;
    if (G > GMAXF) G = GMAXF;
    int ngroups = (nact + G - 1) / G;
    if (ngroups < 1) ngroups = 1;  // an INIT must run (and the duties be done) even if no particle has work
    const long long nitems = nchunks * ngroups;  // item = group * nchunks + chunk: a workgroup's teams share a group
    if ((long long)b * SLOT_TEAMS >= nitems) { PROF_END(); return; }  // no item, hence no duty
    uint8_t* __restrict__ const dst0 = S.lid + (size_t)dst_gen * MAXP * S.n_pad;
    const uint8_t* __restrict__ const lid0 = S.lid;
    const double* __restrict__ const XT = S.XT;
    long long iv[5] = {0, 0, 0, 0, 0};  // INIT/FINAL statistics: A, B, C, E0, QSTD
    unsigned sat = 0;
    int sel_lo = 0, sel_hi = 0;  // work-list range whose split values this workgroup already has
    bool first_item = true;
    for (long long it0 = (long long)b * SLOT_TEAMS; it0 < nitems; it0 += (long long)gridDim.x * SLOT_TEAMS) {
      const long long item = it0 + team;
      const bool valid = item < nitems;
      const int chunk = valid ? (int)(item % nchunks) : 0, grp = valid ? (int)(item / nchunks) : 0;
      const int g0 = valid ? grp * G : 0, g1 = valid ? ((g0 + G < nact) ? g0 + G : nact) : 0;
      // ---- split values of the particles of this pass's groups (the teams' items are consecutive, so
      //      their groups are): one wave per particle, all waves of the workgroup
      {
        const long long last_item = (it0 + SLOT_TEAMS - 1 < nitems) ? it0 + SLOT_TEAMS - 1 : nitems - 1;
        const int lo = (int)(it0 / nchunks) * G;
        int hi = ((int)(last_item / nchunks) + 1) * G;
        if (hi > nact) hi = nact;
        if (lo < sel_lo || hi > sel_hi) {
          if (!first_item) __syncthreads();
          for (int i = lo + w; i < hi; i += NW) {
            const int p = s_list[i];
            const Prop& pr = s_prop[p];
            int found = 0;
            double v = 0.0;
            if (pr.attempt) found = select_split_value(S, pr, p, itp, (uint32_t)rr, s_usel[set][p], s_usel1[set][p], &v);
            if (lane == 0) {
              SJob sj;
              const int jv = pr.var;
              sj.p = p;
              sj.active = found;
              sj.copy = (!found && pr.copy_if_idle) ? 1 : 0;
              sj.check_nan = S.col_nan[jv];
              sj.rule = S.rules[jv];
              sj.label = pr.label;
              sj.new_label = pr.n_leaves;
              sj.ccL = ((rr * MAXP + p) * 2);
              sj.ccR = sj.ccL + 1;
              sj.v = v;
              sj.src = pr.src_slot < 0 ? -1ll : (long long)(((size_t)pr.src_gen * MAXP + pr.src_slot) * S.n_pad);
              sj.xoff = (long long)((size_t)jv * S.n_pad);
              sj.pad = 0;
              s_job[i] = sj;
            }
          }
          __syncthreads();
          sel_lo = lo;
          sel_hi = hi;
        }
      }
      TRS(7);
      const long long base = (long long)chunk * CH + ttid * RPT;
      long long qa[RPT], qb[RPT], qc[RPT];
#pragma unroll
      for (int e = 0; e < RPT; ++e) qa[e] = qb[e] = qc[e] = 0;
      if (valid && do_init) {
        // ---- this slot starts a tree: FINAL of the previous tree + INIT on the fly (as k_rows)
        const bool writer = grp == 0;
        uint32_t ids_next = *(const uint32_t*)(tl_new + base);
        uint32_t ids_sel = 0;
        if (do_final) {
          if (sel_slot == -2) {
            ids_sel = *(const uint32_t*)(tl_old + base);  // old tree kept
          } else {
            if (sel_lid) {
              ids_sel = *(const uint32_t*)(sel_lid + base);
            } else {  // untouched root: label 0 (pad rows: orphan)
#pragma unroll
              for (int e = 0; e < RPT; ++e)
                if (base + e >= n) ids_sel |= (uint32_t)PGB_ORPHAN << (8 * e);
            }
            if (writer) *(uint32_t*)(tl_old + base) = ids_sel;
          }
          if (tree_new == tree_old) ids_next = ids_sel;
        }
        double st4[RPT], y4[RPT], mean4[RPT], m24[RPT];
        {
          const double2* __restrict__ sp = (const double2*)(st_in + base);
          const double2* __restrict__ yp = (const double2*)(S.y + base);
          const double2 s01 = sp[0], s23 = sp[1], y01 = yp[0], y23 = yp[1];
          st4[0] = s01.x; st4[1] = s01.y; st4[2] = s23.x; st4[3] = s23.y;
          y4[0] = y01.x; y4[1] = y01.y; y4[2] = y23.x; y4[3] = y23.y;
          const bool upd = do_final && c.tune && writer;
#pragma unroll
          for (int e = 0; e < RPT; ++e) {
            mean4[e] = upd ? S.rs_mean[base + e] : 0.0;
            m24[e] = upd ? S.rs_m2[base + e] : 0.0;
          }
        }
#pragma unroll
        for (int e = 0; e < RPT; ++e) {
          const long long row = base + e;
          if (row >= n) continue;
          double st = st4[e];
          if (do_final) {
            const double nv = s_lv[0][(ids_sel >> (8 * e)) & 255u];
            st = st + nv;
            if (c.tune && writer) {  // [U] RunningSd.update (Welford)
              const double mean0 = mean4[e], m20 = m24[e];
              const double delta = nv - mean0;
              const double mean = mean0 + delta / cntf;
              const double delta2 = nv - mean;
              const double m2 = m20 + delta * delta2;
              S.rs_mean[row] = mean;
              S.rs_m2[row] = m2;
              iv[4] += pgb_quant(PGB_SQRT(m2 / cntf), c1, &sat);
            }
          }
          const double o = s_lv[1][(ids_next >> (8 * e)) & 255u];
          const double noi = st - o;
          const double rres = y4[e] - noi;
          unsigned sat1 = 0;
          qa[e] = pgb_quant(st, c1, &sat1);
          qb[e] = pgb_quant(rres, c1, &sat1);
          qc[e] = pgb_quant(rres * rres, c2, &sat1);
          if (writer) {
            S.pack[row] = make_double2(st, rres);
            st_out[row] = noi;
            sat += sat1;
            iv[0] += qa[e];
            iv[1] += qb[e];
            iv[2] += qc[e];
            const double er = rres - o;
            iv[3] += pgb_quant(er * er, c2, &sat);
          }
        }
      } else if (valid) {
        if (!(first_item && have_pre)) {
#pragma unroll
          for (int e = 0; e < RPT; ++e) pre_pack[e] = S.pack[base + e];
        }
#pragma unroll
        for (int e = 0; e < RPT; ++e) {
          qa[e] = pgb_quant(pre_pack[e].x, c1, nullptr);
          qb[e] = pgb_quant(pre_pack[e].y, c1, nullptr);
          qc[e] = pgb_quant(pre_pack[e].y * pre_pack[e].y, c2, nullptr);
        }
      }
      first_item = false;
      TRS(8);
      uint32_t root_ids = 0;
#pragma unroll
      for (int e = 0; e < RPT; ++e)
        if (base + e >= n) root_ids |= (uint32_t)PGB_ORPHAN << (8 * e);
      // software pipeline over the particles of the group (as k_rows)
      uint32_t nx_ids = root_ids;
      double2 nx0 = {0.0, 0.0}, nx1 = {0.0, 0.0};
      if (g0 < g1) {
        const SJob& rn = s_job[g0];
        if (rn.src >= 0) nx_ids = *(const uint32_t*)(lid0 + rn.src + base);
        if (rn.active) {
          const double2* __restrict__ xn = (const double2*)(XT + rn.xoff + base);
          nx0 = xn[0];
          nx1 = xn[1];
        }
      }
      long long* const red = s_red[team];
      for (int g = g0; g < g1; ++g) {
        const SJob& rj = s_job[g];
        const uint32_t ids = nx_ids;
        const double2 t0 = nx0, t1 = nx1;
        if (g + 1 < g1) {
          const SJob& rn = s_job[g + 1];
          nx_ids = rn.src < 0 ? root_ids : *(const uint32_t*)(lid0 + rn.src + base);
          if (rn.active) {
            const double2* __restrict__ xn = (const double2*)(XT + rn.xoff + base);
            nx0 = xn[0];
            nx1 = xn[1];
          }
        }
        uint32_t out = ids;
        uint8_t* __restrict__ const dp = dst0 + (size_t)rj.p * n_pad + base;
        if (!rj.active) {
          if (rj.copy) *(uint32_t*)dp = out;  // forced refresh only
          continue;
        }
        const double x[RPT] = {t0.x, t0.y, t1.x, t1.y};
        const int slot = (g - g0) * NRED;
        if (!rj.check_nan) {
          long long v0 = 0, v1 = 0, v2 = 0, v3 = 0;  // cnts(L | R<<20), aL, bL, c2L
#pragma unroll
          for (int e = 0; e < RPT; ++e) {
            if (((ids >> (8 * e)) & 255u) == (uint32_t)rj.label) {
              if (go_left_t<SUB>(rj.rule, x[e], rj.v)) {
                v0 += 1;
                v1 += qa[e]; v2 += qb[e]; v3 += qc[e];
              } else {
                out = (out & ~(255u << (8 * e))) | ((uint32_t)rj.new_label << (8 * e));
                v0 += 1ll << 20;
              }
            }
          }
          *(uint32_t*)dp = out;
          const long long tot = wave_sum4(v0, v1, v2, v3);
          if (lane < 4) red[(slot + lane) * 4 + tw] = tot;
        } else {
          long long v[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int e = 0; e < RPT; ++e) {
            if (((ids >> (8 * e)) & 255u) == (uint32_t)rj.label) {
              const double xv = x[e];
              if (xv != xv) {
                out = (out & ~(255u << (8 * e))) | ((uint32_t)PGB_ORPHAN << (8 * e));
                v[0] += 1ll << 40;
                v[4] += qa[e]; v[5] += qb[e]; v[6] += qc[e];
              } else if (go_left_t<SUB>(rj.rule, xv, rj.v)) {
                v[0] += 1;
                v[1] += qa[e]; v[2] += qb[e]; v[3] += qc[e];
              } else {
                out = (out & ~(255u << (8 * e))) | ((uint32_t)rj.new_label << (8 * e));
                v[0] += 1ll << 20;
              }
            }
          }
          *(uint32_t*)dp = out;
          const long long ta = wave_sum4(v[0], v[1], v[2], v[3]);
          const long long tb = wave_sum4(v[4], v[5], v[6], 0);
          if (lane < 4) red[(slot + lane) * 4 + tw] = ta;
          else if (lane < 7) red[(slot + lane) * 4 + tw] = tb;
        }
      }
      TRS(9);
      // which new particles this team writes out: (group(p), p % nchunks) == (grp, chunk); particles
      // without work belong to group 0
      if (tw == 0) {
        bool mine = false;
        if (valid && lane >= 1 && lane < P) {
          const Prop& pr = s_prop[lane];
          const int pg = pr.haswork ? pr.pos / G : 0;
          mine = pg == grp && (lane % nchunks) == chunk;
        }
        const unsigned long long dm = __ballot(mine);
        if (lane == 0) s_duty[team] = dm;
      }
      __syncthreads();
      for (int t = ttid; t < (g1 - g0) * NRED; t += BT) {
        const int gi = t / NRED, i = t % NRED;
        const SJob& rj = s_job[g0 + gi];
        if (!rj.active || (i >= 4 && !rj.check_nan)) continue;
        const long long s = red[t * 4] + red[t * 4 + 1] + red[t * 4 + 2] + red[t * 4 + 3];
        Acc* a = &S.acc[((size_t)rW * MAXP + rj.p) * ACC_PER + (chunk & (ACC_SLOTS - 1)) * ACC_STRIDE];
        if (i == 0) {
          const int cL = (int)(s & 0xFFFFF), cR = (int)((s >> 20) & 0xFFFFF), cN = (int)(s >> 40);
          S.cc[(size_t)rj.ccL * S.nchunks + chunk] = (uint16_t)cL;
          S.cc[(size_t)rj.ccR * S.nchunks + chunk] = (uint16_t)cR;
          if (cL | cN) atomicAdd(&a->cnts, (unsigned long long)cL | ((unsigned long long)cN << 32));
        } else if (s != 0) {
          long long* dst = i == 1 ? &a->aL : i == 2 ? &a->bL : i == 3 ? &a->c2L : i == 4 ? &a->aN : i == 5 ? &a->bN : &a->c2N;
          atomicAdd((unsigned long long*)dst, (unsigned long long)s);
        }
      }
      TRS(10);
      // ---- duties of this item
      for (unsigned long long dm = s_duty[team]; dm != 0ull; dm &= dm - 1ull) {
        const int p = (int)__ffsll((long long)dm) - 1;
        const Prop& pr = s_prop[p];
        DPart* me = &NT[p];
        if (fresh) {
          if (ttid == 0) {
            DNode z;
            memset(&z, 0, sizeof z);
            z.var = -1;
            z.cc_row = -1;
            z.cnt = (int32_t)S.n;
            z.value = S.init_leaf;
            me->nd[0] = z;
          }
        } else {
          copy_patched(me->nd, OT[pr.anc].nd, s_fin[pr.anc], r1, ia, root_sse, ttid);
        }
        if (ttid == 0) {
          Job job;
          memset(&job, 0, sizeof job);
          job.src_gen = pr.src_gen;
          job.src_slot = pr.src_slot;
          job.h_n_nodes = pr.n_nodes;
          job.h_n_leaves = pr.n_leaves;
          job.h_next_pop = pr.next_pop;
          job.h_sse_tot = pr.sse_tot;
          job.h_sse_orph = pr.sse_orph;
          if (pr.node >= 0) atomicAdd(&S.counters[0], 1ull);
          if (pr.haswork) {
            const SJob& sj = s_job[pr.pos];
            job.copy = sj.copy;
            if (sj.active) {
              job.active = 1;
              job.node = pr.node;
              job.label = pr.label;
              job.new_label = pr.n_leaves;
              job.var = pr.var;
              job.rule = sj.rule;
              job.check_nan = sj.check_nan;
              job.ccL = sj.ccL;
              job.ccR = sj.ccR;
              job.cnt = pr.cnt;
              job.v = sj.v;
              job.p_q_st = pr.q_st;
              job.p_q_r = pr.q_r;
              job.p_q_r2 = pr.q_r2;
              job.p_sse = pr.sse;
              job.p_value = pr.value;
              job.p_depth = pr.depth;
              atomicAdd(&S.counters[2], (unsigned long long)pr.cnt);
              atomicAdd(&S.counters[6], 1ull);
            }
          }
          JN[p] = job;
          me->n_nodes = pr.n_nodes;
          me->n_leaves = pr.n_leaves;
          me->next_pop = pr.next_pop;
          me->loc_gen = pr.src_gen;
          me->loc_slot = pr.src_slot;
          me->sse_tot = pr.sse_tot;
          me->sse_orph = pr.sse_orph;
        }
      }
      TRS(11);
    }
    if (do_init) {  // statistics of the INIT (+FINAL) part: one sum per wave, one atomic per wave and value
      unsigned any_sat = sat;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const long long t = wave_sum_dpp(iv[i]);
        if (lane == 63 && t != 0) {
          InitAcc* a = &S.initacc[(size_t)rW * IA_SLOTS + ((b * NW + w) % IA_SLOTS)];
          long long* dst = i == 0 ? &a->A : i == 1 ? &a->B : i == 2 ? &a->C : i == 3 ? &a->E0 : &a->QSTD;
          atomicAdd((unsigned long long*)dst, (unsigned long long)t);
        }
      }
      if (any_sat) atomicAdd(&S.counters[4], (unsigned long long)any_sat);
    }
  } else {
    // ---------------- lone FINAL (last tree of the last requested step): one row per thread and item
    long long v4 = 0;
    unsigned sat = 0;
    const long long nitems = (S.n_pad + SLOT_BT - 1) / SLOT_BT;
    for (long long item = b; item < nitems; item += gridDim.x) {
      const long long row = item * SLOT_BT + tid;
      if (row >= S.n) continue;
      double st = st_in[row];
      uint32_t id_sel;
      if (sel_slot == -2) {
        id_sel = tl_old[row];
      } else {
        id_sel = sel_lid ? (uint32_t)sel_lid[row] : 0u;
        tl_old[row] = (uint8_t)id_sel;
      }
      const double nv = s_lv[0][id_sel];
      st = st + nv;
      if (c.tune) {
        const double mean0 = S.rs_mean[row], m20 = S.rs_m2[row];
        const double delta = nv - mean0;
        const double mean = mean0 + delta / cntf;
        const double delta2 = nv - mean;
        const double m2 = m20 + delta * delta2;
        S.rs_mean[row] = mean;
        S.rs_m2[row] = m2;
        v4 += pgb_quant(PGB_SQRT(m2 / cntf), S.sc.c1, &sat);
      }
      st_out[row] = st;
    }
    if (sat) atomicAdd(&S.counters[4], (unsigned long long)sat);
    if (c.tune) {
      const long long t = wave_sum_dpp(v4);
      if (lane == 63 && t != 0) {
        InitAcc* a = &S.initacc[(size_t)rW * IA_SLOTS + ((b * NW + w) % IA_SLOTS)];
        atomicAdd((unsigned long long*)&a->QSTD, (unsigned long long)t);
      }
    }
    // no proposal follows: empty job records
    for (int p = 1 + b; p < P; p += gridDim.x)
      if (tid == 0) {
        Job z;
        memset(&z, 0, sizeof z);
        JN[p] = z;
      }
  }

  // =================================================================== once per slot: workgroup 0
  if (b == 0) {
    if (stop) {
      // the accepted tree: store it, count its split variables ([U] tuning / variable inclusion)
      const bool grown = sel >= 1;
      DTree* T = &S.trees[tree_old];
      if (grown && team == 0) {
        const FinS& F = s_fin[sel];
        copy_patched(T->nd, OT[sel].nd, F, r1, ia, root_sse, ttid);
        if (tid == 0) {
          T->n_nodes = F.n_nodes;
          T->n_leaves = F.n_leaves;
        }
      }
      if (c.tune) {
        const long long* alpha = S.alpha + (size_t)c.alpha_cur * S.p;
        long long* alpha_o = S.alpha + (size_t)(c.alpha_cur ^ 1) * S.p;
        if (rebuild) {
          long long* cdf_o = S.cdfS + (size_t)(c.cdf_cur ^ 1) * S.p;
          for (int j = tid; j < S.p; j += SLOT_BT) cdf_o[j] = s_cdf[1][j];
        }
        for (int j = tid; j < S.p; j += SLOT_BT) alpha_o[j] = alpha[j];
      }
      __syncthreads();
      if (tid == 0) {
        // split variables of the accepted tree (old nodes; the pending split's variable; none of a stump)
        long long* alpha_o = S.alpha + (size_t)(c.alpha_cur ^ 1) * S.p;
        const DNode* snd = grown ? OT[sel].nd : T->nd;
        const int nn = grown ? s_fin[sel].nn_old : T->n_nodes;
        for (int i = 0; i < nn; ++i) {
          int v = snd[i].var;
          if (grown && s_fin[sel].ok == 1 && i == s_fin[sel].node) v = s_fin[sel].var;
          if (v >= 0) {
            if (c.tune) alpha_o[v] += S.alpha_unit;
            else S.vi[v] += 1;
          }
        }
        atomicAdd(&S.counters[1], 1ull);
        atomicAdd(&S.counters[3], 1ull);
      }
    }
    if (tid == 0) {
      Ctrl o = c;
      o.slot_no = c.slot_no + 1;
      o.leaf_sd = leaf_sd;
      o.pend_leafsd = 0;
      if (lone_final) {
        o.rs_count = rs_count;
        o.pend_leafsd = c.tune ? 1 : 0;
        o.pend_iter = c.iter;
        o.round = 0;
        o.k = k_next;
        o.lower = lower_next;
        o.batch_n = batch_next;
        o.phase = PH_IDLE;
        o.steps_left = 0;
        o.steps_done = c.steps_done + 1;
        if (c.tune) o.alpha_cur = c.alpha_cur ^ 1;
        if (rebuild) o.cdf_cur = c.cdf_cur ^ 1;
        *co = o;
        __hip_atomic_store(S.host_flag, (unsigned long long)o.steps_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        o.lid_gen = dst_gen;
        o.sse0 = sse0;
        o.phase = PH_ROUND;
        if (!fresh) {
          o.round = r + 1;
          atomicAdd(&S.counters[3], 1ull);  // round r-1 is complete
        } else {
          o.round = 1;
          o.iter = c.iter + 1;
          o.st_cur = c.st_cur ^ 1;  // INIT writes sum_trees_noi into the other buffer
          if (stop) {
            o.rs_count = rs_count;
            o.pend_leafsd = c.tune ? 1 : 0;
            o.pend_iter = c.iter;
            o.k = k_next;
            o.lower = lower_next;
            o.batch_n = batch_next;
            if (!more) {
              o.steps_left = c.steps_left - 1;
              o.steps_done = c.steps_done + 1;
            }
            if (c.tune) o.alpha_cur = c.alpha_cur ^ 1;
            if (rebuild) o.cdf_cur = c.cdf_cur ^ 1;
          }
        }
        *co = o;
        if (fresh && stop && !more)
          __hip_atomic_store(S.host_flag, (unsigned long long)o.steps_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  TRS(12);
  PROF_END();
#undef PROF_END
#undef TRS
}
