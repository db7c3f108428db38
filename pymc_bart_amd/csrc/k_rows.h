// k_rows.h -- part of pgbart_hip.hip (not a standalone header): k_rows: the single-output row pass (PARTITION / INIT / FINAL).
// ------------------------------------------------------------------ k_rows
// Persistent grid (<= 1024 workgroups): work items are looped over, because on MI355X the
// dispatch of a workgroup costs ~3-4 ns and a (chunk x particle) grid of thousands of
// workgroups was dispatch-bound, not bandwidth-bound (profiles/r01_*).
//
// PARTITION item = (1024-row chunk, group of G particles with work).  The workgroup loads and
// quantises {sum_trees, r} of its rows ONCE and then, for each particle of the group, relabels
// the rows of the leaf being split and reduces the left child's statistics.  Particles without
// work in this round are not touched at all: their labels stay where they are (NGEN generations).
// FINAL/INIT item = 256 rows.

// Work items a row pass aims for (measured on cfg2: 640 for plain rounds, 768 for the fused
// FINAL+INIT+round-0 pass whose INIT part is repeated by every particle group).
#define ROWS_TARGET_ITEMS 640
#define ROWS_TARGET_ITEMS_INIT 768

struct RJob {  // the fields of a Job the row pass needs, cached in LDS
  long long src;   // byte offset of the source labels in S.lid, -1: implicit root labels
  long long xoff;  // element offset of the split column in S.XT
  double v;
  double uscale;  // linear response: 2^-ex of the split column
  int32_t p, active, check_nan, rule, label, new_label, ccL, ccR;
  int32_t vkey, pad;  // order key of v (shadow instances)
};

// LIN: linear response (Normal family only): leaves predict value + slope (x[svar] - xbar); the
// partition additionally reduces the sums pgb_lin_fit needs for both children.
// F32 ("shadow" instances): the split column is read as 16-BIT ORDER KEYS (XK16, built by pgb_set_data when the
// design matrix does not fit the Infinity Cache -- cfg4: 800 MB): a quarter of the bytes of the pass's largest stream.
// key(x) = number of the column's 65 534 equi-depth boundaries (of its float32-rounded values) that are <= x, so
// key(x) < key(v) implies x < v and key(x) > key(v) implies x > v; only EQUAL keys -- one row in 65 536 -- need the
// float64 value, which that lane then fetches.  A missing value has key 0xFFFF.  Same decisions, bit for bit.
// (Continuous / one-hot rules, constant leaves.  Round 3 / early round 4: a float32 shadow, half the bytes.)
__device__ __forceinline__ uint2 gload_k4(gptr<const uint16_t> p) {  // the keys of four adjacent rows: one 8-byte load
  typedef uint32_t v2u __attribute__((ext_vector_type(2)));
  const v2u v = *(gptr<const v2u>)p;
  return make_uint2(v.x, v.y);
}
template <bool SUB, bool NORMAL, bool LIN, bool F32 = false>
__global__ __launch_bounds__(BT, (NORMAL && !LIN) ? 3 : 2) void k_rows(const Dev* __restrict__ Sp, int par,
                                                              const Cmd* __restrict__ cmds,
                                                              const Job* __restrict__ jobs_all) {
  // cmds / jobs_all repeat S.cmd / S.jobs as kernel arguments: their first loads then do not wait
  // for the load of the argument block S itself (one dependent memory round trip less)
  const DevG& S = *reinterpret_cast<const DevG*>(Sp);
  constexpr int NRED = LIN ? 15 : 7;  // values reduced per particle
  __shared__ long long s_red[MAXP * NRED * 4];
  // constant leaves, split column without missing values: the sums of a particle go straight into LDS with atomic
  // adds (lane l to entry l mod NE of the value) instead of through a wave butterfly -- one ds_add per value in
  // place of 21-42 vector instructions per wave and particle; the reducer threads add the NE entries up
  // (F32 instances only: at cfg2 -- float64 columns in the Infinity Cache, items of five particles -- the
  //  butterflies are faster: k_rows 6.62 us against 6.82 with the atomics, A/B on one box)
  constexpr bool ATOM = F32 && !LIN;
  constexpr int NVA = !ATOM ? 1 : (NORMAL ? 4 : 2);
  constexpr int NE = !ATOM ? 1 : (NORMAL ? 8 : 16) / (MAXP / 64);
  __shared__ unsigned long long s_acc[ATOM ? MAXP * NVA * NE : 1];
  __shared__ double s_lv[2][256];
  __shared__ LinP s_ll[LIN ? 2 : 1][LIN ? 256 : 1];  // label -> linear part: [0 new | 1 next]
  __shared__ RJob s_job[MAXP];
  __shared__ int s_n[2];
  const Cmd* cmd = &cmds[par];
  const int kind = cmd->kind;
#ifndef PGB_ROWS_JPRE
#define PGB_ROWS_JPRE 1 /* experiment knob: 0 = the job records are requested where the list is made, behind the command word */
#endif
  // The job records of this lane's particles (wave 0: the particle list below) are requested WITH the command word:
  // their addresses depend on `par` alone and the records exist for every index below MAXP.  Requested where the list
  // is made they were a second round trip behind it in every workgroup of every launch (stamps 12 -> 13: 1.24 us at
  // cfg2).  (k_ctrl and k_loglik start the same way.)
  Job j_pre[MAXP / 64];
  if constexpr (PGB_ROWS_JPRE != 0) {
    if (threadIdx.x < 64) {
#pragma unroll
      for (int hq = 0; hq < MAXP / 64; ++hq) j_pre[hq] = jobs_all[(size_t)par * MAXP + threadIdx.x + 64 * hq];
    }
  }
  TRR_BIND(S.ctrl[par ^ 1].slot_no - 1);  // (stamp 12: entry)
  // profiling: every workgroup leaves its first and last device-clock reading; the host takes
  // min(start) .. max(end) per launch -- the interval rocprofv3 reports for the dispatch
  long long* pstamp = nullptr;
  if (S.prof_stamps != nullptr && threadIdx.x == 0 && blockIdx.x < PROF_BLOCKS && !PGB_STAMP_LL_ON) {
    pstamp = S.prof_stamps + ((size_t)((S.ctrl[par ^ 1].slot_no - 1) % PROF_RING) * PROF_BLOCKS + blockIdx.x) * 2;
    pstamp[0] = wall_clock64();
    pstamp[1] = pstamp[0];
  }
#define PROF_END() do { if (pstamp) pstamp[1] = wall_clock64(); } while (0)
  if (kind == CMD_NOOP) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const bool do_final = (kind & CMD_FINAL) != 0, do_init = (kind & CMD_INIT) != 0;
  const bool do_part = (kind & CMD_PARTITION) != 0;
  constexpr bool normal = NORMAL;  // compiled per family class: the Normal instance carries no log-likelihood code

  if (do_final || do_init) {
    for (int i = tid; i < 256; i += BT) {
      s_lv[0][i] = (!LIN && cmd->sel_slot == -2) ? cmd->lv_keep[i] : cmd->lv_new[i];
      s_lv[1][i] = cmd->lv_next[i];
      if constexpr (LIN) {
        s_ll[0][i] = S.lvl[((size_t)par * 2 + 0) * 256 + i];
        s_ll[1][i] = S.lvl[((size_t)par * 2 + 1) * 256 + i];
      }
    }
  }
  // (arrays of the argument block as GLOBAL pointers: see as_global)
  const gptr<uint8_t> tl_old = as_global(do_final ? S.tree_lid + (size_t)cmd->tree_old * S.n_pad : nullptr);
  const gptr<const uint8_t> tl_new = as_global(do_init ? (const uint8_t*)S.tree_lid + (size_t)cmd->tree_new * S.n_pad : nullptr);
  const gptr<const uint8_t> sel_lid = as_global(
      (do_final && cmd->sel_slot >= 0) ? (const uint8_t*)S.lid + ((size_t)cmd->sel_gen * MAXP + cmd->sel_slot) * S.n_pad : nullptr);
  const gptr<double> rs_mean = as_global(S.rs_mean), rs_m2 = as_global(S.rs_m2);
  const gptr<double> pack = as_global((double*)S.pack);  // [n_pad] pairs {sum_trees, r}
  const double cntf = (double)cmd->rs_count;
  // sum_trees buffers: an INIT reads st_in and writes sum_trees_noi to st_out (other workgroups of
  // the same chunk still read st_in); a lone FINAL updates st_in in place
  const gptr<double> st_in = as_global(S.st + (size_t)cmd->st_cur * S.n_pad);
  const gptr<double> st_out = as_global(S.st + (size_t)(do_init ? cmd->st_cur ^ 1 : cmd->st_cur) * S.n_pad);

  if (do_part) {
    if constexpr (ATOM)
      for (int i = tid; i < MAXP * NVA * NE; i += BT) s_acc[i] = 0ull;
    const Job* jobs = jobs_all + (size_t)par * MAXP;
    // list of particles with work in this pass (split or forced label refresh); their job
    // fields are cached in LDS once per workgroup
    if (tid < 64) {
      int nlist = 0;  // (wave 0 lists the particles with work: lanes' particles tid, tid + 64, ... one block after the other)
      bool plain = true;   // every particle with work splits the root of a fresh stump on a continuous column without NaNs
      bool common = true;  // every split is on a continuous column without NaNs (label-refresh-only jobs allowed)
#pragma unroll
      for (int hq = 0; hq < MAXP / 64; ++hq) {
      const int q = tid + 64 * hq;
      Job j;
      j.active = 0;
      j.copy = 0;
      if (q >= 1 && q < S.P) j = PGB_ROWS_JPRE != 0 ? j_pre[hq] : jobs[q];  // (requested at the head of the kernel)
      const bool has = (j.active | j.copy) != 0;
      const unsigned long long m = __ballot(has);
      if constexpr (F32 && !LIN) {
        const bool cont = !j.check_nan && j.rule == PGB_RULE_CONTINUOUS;
        if (__any(has && j.active && !cont)) common = false;
        if (__any(has && !(j.active && cont && j.src_slot < 0 && j.label == 0))) plain = false;
      }
      if (has) {
        const int k = nlist + __popcll(m & ((1ull << tid) - 1ull));
        RJob rj;
        rj.p = q;
        rj.active = j.active;
        rj.check_nan = j.check_nan;
        rj.rule = j.rule;
        rj.label = j.label;
        rj.new_label = j.new_label;
        rj.ccL = j.ccL;
        rj.ccR = j.ccR;
        rj.v = j.v;
        rj.vkey = j.vkey;
        rj.pad = 0;
        rj.src = j.src_slot < 0 ? -1ll : (long long)(((size_t)j.src_gen * MAXP + j.src_slot) * S.n_pad);
        rj.xoff = (long long)((size_t)j.var * S.n_pad);
        rj.uscale = 1.0;
        if constexpr (LIN) rj.uscale = j.active ? pgb_pow2(-S.col_ex[j.var]) : 1.0;
        s_job[k] = rj;
      }
      nlist += __popcll(m);
      }
      if (tid == 0) {
        s_n[0] = nlist;
        if constexpr (F32 && !LIN) s_n[1] = plain ? 2 : common ? 1 : 0;
      }
    }
    __syncthreads();
    TRR(13, 0);
    const int nact = s_n[0];
    const int pass_kind = (F32 && !LIN) ? s_n[1] : 0;  // 2: plain round, 1: common round, 0: general (shadow instances only)
    const bool all_plain = pass_kind == 2;
    if (nact == 0 && !do_init) { PROF_END(); return; }
    const int target = do_init ? S.rows_target_init : S.rows_target;
    int G = (nact * S.nchunks + target - 1) / target;
    if (G < 1) G = 1;
    int ngroups = (nact + G - 1) / G;
    if (ngroups < 1) ngroups = 1;  // an INIT must run even if no particle splits
    const int nitems = S.nchunks * ngroups;
    const gptr<uint8_t> __restrict__ dst0 = as_global(S.lid + (size_t)cmd->dst_gen * MAXP * S.n_pad);
    const gptr<const uint8_t> __restrict__ lid0 = as_global((const uint8_t*)S.lid);
    const gptr<const double> __restrict__ XT = as_global(S.XT);
    const gptr<const uint16_t> __restrict__ XK = as_global(S.XK16);
    const gptr<const double> __restrict__ yarr = as_global(S.y);
    const gptr<uint16_t> ccp = as_global(S.cc);
    const double c1 = S.sc.c1, c2 = S.sc.c2;
    const long long n = S.n, n_pad = S.n_pad;
    long long iv[5] = {0, 0, 0, 0, 0};  // INIT/FINAL statistics: A, B, C, E0, QSTD
    unsigned sat = 0;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
      const int chunk = item % S.nchunks, grp = item / S.nchunks;
      const long long base = (long long)chunk * CH + tid * RPT;
      // rows of this thread: {sum_trees, r} quantised once, reused for every particle of the group
      long long qa[RPT], qb[RPT], qc[RPT];
      double strow[RPT], rrow[RPT];  // linear response: the unquantised {sum_trees, r} of the rows
#pragma unroll
      for (int e = 0; e < RPT; ++e) strow[e] = rrow[e] = 0.0;
      if (do_init) {
        // ---- this slot starts a tree: finish the previous tree (FINAL) and compute the new
        // residuals (INIT) on the fly; the first group of each chunk also writes them back
        const bool writer = grp == 0;
        uint32_t ids_next = *gcast<const uint32_t>(tl_new + base);
        uint32_t ids_sel = 0;
        if (do_final) {
          if (cmd->sel_slot == -2) {
            ids_sel = *gcast<const uint32_t>(tl_old + base);  // old tree kept
          } else {
            if (sel_lid) {
              ids_sel = *gcast<const uint32_t>(sel_lid + base);
            } else {  // untouched root: label 0 (pad rows: orphan)
#pragma unroll
              for (int e = 0; e < RPT; ++e)
                if (base + e >= n) ids_sel |= (uint32_t)PGB_ORPHAN << (8 * e);
            }
            if (writer) *gcast<uint32_t>(tl_old + base) = ids_sel;
          }
          if (cmd->tree_new == cmd->tree_old) ids_next = ids_sel;
        }
        // every input of the thread's four rows is requested BEFORE the first result is stored:
        // the stores below may alias the loads as far as the compiler knows, so loads left inside
        // the loop would be issued one row (one memory round trip) at a time
        double st4[RPT], y4[RPT], mean4[RPT], m24[RPT];
        {
          const gptr<const double> sp = st_in + base, yp = yarr + base;
          const double2 s01 = gload_d2(sp), s23 = gload_d2(sp + 2), y01 = gload_d2(yp), y23 = gload_d2(yp + 2);
          st4[0] = s01.x; st4[1] = s01.y; st4[2] = s23.x; st4[3] = s23.y;
          y4[0] = y01.x; y4[1] = y01.y; y4[2] = y23.x; y4[3] = y23.y;
          const bool upd = do_final && cmd->tune && writer;
#pragma unroll
          for (int e = 0; e < RPT; ++e) {
            mean4[e] = upd ? rs_mean[base + e] : 0.0;
            m24[e] = upd ? rs_m2[base + e] : 0.0;
          }
        }
#pragma unroll
        for (int e = 0; e < RPT; ++e) {
          const long long row = base + e;
          qa[e] = qb[e] = qc[e] = 0;
          if (row >= n) continue;
          double st = st4[e];  // sum_trees at a step boundary, sum_trees_noi inside an update
          if (do_final) {
            // [U] sum_trees = sum_trees_noi + new_tree.predict()
            double nv = s_lv[0][(ids_sel >> (8 * e)) & 255u];
            if constexpr (LIN) {
              const LinP lp = s_ll[0][(ids_sel >> (8 * e)) & 255u];
              if (lp.svar >= 0) nv = pgb_leaf_pred(nv, lp.slope, lp.xbar, XT[(size_t)lp.svar * n_pad + row]);
            }
            st = st + nv;
            if (cmd->tune && writer) {  // [U] RunningSd.update (Welford)
              const double mean0 = mean4[e], m20 = m24[e];
              const double delta = nv - mean0;
              const double mean = mean0 + delta / cntf;
              const double delta2 = nv - mean;
              const double m2 = m20 + delta * delta2;
              rs_mean[row] = mean;
              rs_m2[row] = m2;
              iv[4] += pgb_quant(PGB_SQRT(m2 / cntf), c1, &sat);
            }
          }
          // [U] sum_trees_noi = sum_trees - old_tree.predict()
          double o = s_lv[1][(ids_next >> (8 * e)) & 255u];
          if constexpr (LIN) {
            const LinP lp = s_ll[1][(ids_next >> (8 * e)) & 255u];
            if (lp.svar >= 0) o = pgb_leaf_pred(o, lp.slope, lp.xbar, XT[(size_t)lp.svar * n_pad + row]);
          }
          const double noi = st - o;
          const double yv = y4[e];
          const double r = normal ? yv - noi : 0.0;  // Bernoulli families: no residual algebra
          unsigned sat1 = 0;
          qa[e] = pgb_quant(st, c1, &sat1);
          qb[e] = normal ? pgb_quant(r, c1, &sat1) : 0;  // (per-row families: r is 0)
          qc[e] = normal ? pgb_quant(r * r, c2, &sat1) : 0;
          strow[e] = st;
          rrow[e] = r;
          if (writer) {  // saturation is counted where the values are produced, once
            gstore_d2(pack + 2 * row, st, r);
            st_out[row] = noi;
            sat += sat1;
            iv[0] += qa[e];
            iv[1] += qb[e];
            if (normal) {
              iv[2] += qc[e];
              const double er = r - o;
              iv[3] += pgb_quant(er * er, c2, &sat);
            }
            // (per-row families: C, the log-likelihood of a fresh stump, and E0, of the current tree, are
            //  summed by k_loglik, which runs after this pass and is compiled per family)
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < RPT; ++e) {
          const double2 sr = gload_d2(pack + 2 * (base + e));
          strow[e] = sr.x;
          rrow[e] = sr.y;
          qa[e] = pgb_quant(sr.x, c1, nullptr);
          // (per-row families carry no residual algebra: r is 0 in `pack`, and the constants let the
          //  compiler drop the two sums from every particle's reduction)
          qb[e] = normal ? pgb_quant(sr.y, c1, nullptr) : 0;
          qc[e] = normal ? pgb_quant(sr.y * sr.y, c2, nullptr) : 0;
        }
      }
      uint32_t root_ids = 0;
#pragma unroll
      for (int e = 0; e < RPT; ++e)
        if (base + e >= n) root_ids |= (uint32_t)PGB_ORPHAN << (8 * e);
      const int g0 = grp * G, g1 = (g0 + G < nact) ? g0 + G : nact;
      // general rounds: software pipeline over the particles of the group -- the labels and split-column values of
      // particle g + 1 are requested before particle g is relabelled and reduced
      uint32_t nx_ids = root_ids;
      double2 nx0 = {0.0, 0.0}, nx1 = {0.0, 0.0};
      uint2 nxk = {0u, 0u};
      auto fetch1 = [&](int gg) {
        nx_ids = root_ids;
        if (gg < g1) {
          const RJob& rn = s_job[gg];
          if (rn.src >= 0) nx_ids = *gcast<const uint32_t>(lid0 + rn.src + base);
          if (rn.active) {
            if constexpr (F32) {
              nxk = gload_k4(XK + rn.xoff + base);
            } else {
              const gptr<const double> xn = XT + rn.xoff + base;
              nx0 = gload_d2(xn);
              nx1 = gload_d2(xn + 2);
            }
          }
        }
      };
      const bool plain_item = F32 && !LIN && all_plain && g1 > g0;
      const bool common_item = pass_kind == 1 && g1 > g0;
      if (!plain_item && !common_item) fetch1(g0);
      TRR(14, 0);  // rows of the item loaded and quantised (INIT part done)
      const bool full_chunk = (long long)(chunk + 1) * CH <= n;  // every row of the chunk is a row of the data
      int g_first = g0;  // particles g0 .. g_first - 1 of this item went through the plain round
      // ---- plain round: EVERY particle of the pass splits the root of a fresh stump (implicit root labels: all
      // rows of a full chunk are in the leaf) on a continuous column without missing values -- the slot that starts
      // a tree, half of this kernel's time at cfg4, where one workgroup walks ONE item of 39 particles.  One stage
      // of prefetch left that walk waiting a whole memory round trip per particle (~1.4 us x 39; fewer bytes and
      // fewer instructions -- no label compare, LDS atomics for butterflies -- changed nothing before that was gone:
      // profiles/r04_experiments.md section 8): here the keys of PD particles' split columns are in flight, in
      // registers that are never moved (the loop is unrolled by PD: a register shift would wait for the loads it moves).
      if constexpr (!LIN && F32) {  // (F32: the data sets beyond the Infinity Cache; the smaller ones keep their registers)
        if (plain_item) {
          constexpr int PD = 4;
          uint2 pf[PD];
          // (every load and store of the loop is unconditional -- past the end the last particle's column is
          //  requested again and dropped -- so that the number of operations in flight behind the one being
          //  waited for is the same on every path and the wait can leave them in flight)
          auto fetch = [&](int gg, uint2& ff) { ff = gload_k4(XK + uni(s_job[gg < g1 ? gg : g1 - 1].xoff) + base); };
          // one particle from its stage registers; `more`: request the column of particle g + PD into them
          auto stage = [&](int g, uint2& ff, bool more) {
            const RJob& rj = s_job[g];
            const double r_v = uni(rj.v);
            const uint32_t r_vk = uni((uint32_t)rj.vkey);
            const uint32_t nw = uni((uint32_t)rj.new_label);
            const long long xo = uni(rj.xoff);
            const gptr<uint8_t> __restrict__ dp = dst0 + (size_t)uni(rj.p) * n_pad + base;
            const uint32_t xk[RPT] = {ff.x & 0xFFFFu, ff.x >> 16, ff.y & 0xFFFFu, ff.y >> 16};
            bool L[RPT];
#pragma unroll
            for (int e = 0; e < RPT; ++e) L[e] = xk[e] < r_vk;
            // equal keys (the row of the split value and its bin mates) are decided on the float64 values
            if (__any((xk[0] == r_vk) | (xk[1] == r_vk) | (xk[2] == r_vk) | (xk[3] == r_vk))) {
#pragma unroll
              for (int e = 0; e < RPT; ++e)
                if (xk[e] == r_vk) L[e] = XT[xo + base + e] <= r_v;
            }
            if (more) fetch(g + PD, ff);
            uint32_t out = root_ids, cl = 0, cr = RPT;
            long long a1 = 0, a2 = 0, a3 = 0;
            if (full_chunk) {
#pragma unroll
              for (int e = 0; e < RPT; ++e) {
                out |= L[e] ? 0u : (nw << (8 * e));
                cl += L[e] ? 1u : 0u;
                a1 += L[e] ? qa[e] : 0ll;
                if constexpr (NORMAL) { a2 += L[e] ? qb[e] : 0ll; a3 += L[e] ? qc[e] : 0ll; }
              }
              cr -= cl;
            } else {  // the last chunk of the data: rows past the end carry the orphan label and count nowhere
              cr = 0;
#pragma unroll
              for (int e = 0; e < RPT; ++e) {
                const bool in = base + e < n;
                out |= (in && !L[e]) ? (nw << (8 * e)) : 0u;
                cl += (in && L[e]) ? 1u : 0u;
                cr += (in && !L[e]) ? 1u : 0u;
                a1 += (in && L[e]) ? qa[e] : 0ll;
                if constexpr (NORMAL) { a2 += (in && L[e]) ? qb[e] : 0ll; a3 += (in && L[e]) ? qc[e] : 0ll; }
              }
            }
            *gcast<uint32_t>(dp) = out;
            unsigned long long* ap = &s_acc[(g - g0) * (NVA * NE) + (lane & (NE - 1))];
            atomicAdd(ap, (unsigned long long)(cl | (cr << 20)));
            atomicAdd(ap + NE, (unsigned long long)a1);
            if constexpr (NORMAL) {
              atomicAdd(ap + 2 * NE, (unsigned long long)a2);
              atomicAdd(ap + 3 * NE, (unsigned long long)a3);
            }
          };
#pragma unroll
          for (int d = 0; d < PD; ++d) {
            pf[d] = uint2{0u, 0u};
            fetch(g0 + d, pf[d]);
          }
          const int g_main = g0 + (g1 - g0) / PD * PD;  // whole blocks of PD particles, then the rest from their stages
          for (int gb = g0; gb < g_main; gb += PD) {
#pragma unroll
            for (int d = 0; d < PD; ++d) stage(gb + d, pf[d], true);
          }
#pragma unroll
          for (int d = 0; d < PD; ++d)
            if (g_main + d < g1) stage(g_main + d, pf[d], false);
          g_first = g1;
        }
      }
      // ---- common round (shadow instances): every split of the pass is on a continuous column without missing
      // values (particles that only carry their labels forward included).  The later rounds walk ONE item of 20-39
      // particles per workgroup with an L2 / Infinity-Cache round trip per particle behind one stage of prefetch; here
      // the labels and keys of PD particles are in flight (same construction as the plain round).
      if constexpr (!LIN && F32) {
        if (common_item) {
          constexpr int PD = 4;
          uint2 ck[PD];
          uint32_t cl4[PD];
          auto fetchc = [&](int gg, uint2& ff, uint32_t& fl) {
            const RJob& rn = s_job[gg < g1 ? gg : g1 - 1];
            ff = gload_k4(XK + uni(rn.xoff) + base);  // (a label-only job: column 0, not used)
            const long long so = uni(rn.src);
            fl = *gcast<const uint32_t>(lid0 + (so < 0 ? 0ll : so) + base);  // (an implicit root: loaded, not used)
          };
          auto stagec = [&](int g, uint2& ff, uint32_t& fl, bool more) {
            const RJob& rj = s_job[g];
            const bool act = uni(rj.active) != 0;
            const double r_v = uni(rj.v);
            const uint32_t r_vk = uni((uint32_t)rj.vkey);
            const uint32_t nw = uni((uint32_t)rj.new_label), lb = uni((uint32_t)rj.label);
            const long long xo = uni(rj.xoff);
            const gptr<uint8_t> __restrict__ dp = dst0 + (size_t)uni(rj.p) * n_pad + base;
            const uint32_t ids = uni(rj.src) < 0 ? root_ids : fl;
            const uint32_t xk[RPT] = {ff.x & 0xFFFFu, ff.x >> 16, ff.y & 0xFFFFu, ff.y >> 16};
            bool in[RPT], L[RPT];
#pragma unroll
            for (int e = 0; e < RPT; ++e) {
              in[e] = act && ((ids >> (8 * e)) & 255u) == lb;  // row of the leaf being split
              L[e] = xk[e] < r_vk;
            }
            if (__any((in[0] && xk[0] == r_vk) | (in[1] && xk[1] == r_vk) | (in[2] && xk[2] == r_vk) | (in[3] && xk[3] == r_vk))) {
#pragma unroll
              for (int e = 0; e < RPT; ++e)
                if (in[e] && xk[e] == r_vk) L[e] = XT[xo + base + e] <= r_v;  // equal keys: the float64 values decide
            }
            if (more) fetchc(g + PD, ff, fl);
            uint32_t out = ids, cl = 0, cr = 0;
            long long a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int e = 0; e < RPT; ++e) {
              const bool le = in[e] && L[e], ri = in[e] && !L[e];
              out = ri ? ((out & ~(255u << (8 * e))) | (nw << (8 * e))) : out;
              cl += le ? 1u : 0u;
              cr += ri ? 1u : 0u;
              a1 += le ? qa[e] : 0ll;
              if constexpr (NORMAL) { a2 += le ? qb[e] : 0ll; a3 += le ? qc[e] : 0ll; }
            }
            *gcast<uint32_t>(dp) = out;
            if ((cl | cr) != 0) {  // (lanes that hold a row of the leaf)
              unsigned long long* ap = &s_acc[(g - g0) * (NVA * NE) + (lane & (NE - 1))];
              atomicAdd(ap, (unsigned long long)(cl | (cr << 20)));
              atomicAdd(ap + NE, (unsigned long long)a1);
              if constexpr (NORMAL) {
                atomicAdd(ap + 2 * NE, (unsigned long long)a2);
                atomicAdd(ap + 3 * NE, (unsigned long long)a3);
              }
            }
          };
#pragma unroll
          for (int d = 0; d < PD; ++d) {
            ck[d] = uint2{0u, 0u};
            cl4[d] = 0u;
            fetchc(g0 + d, ck[d], cl4[d]);
          }
          const int g_main = g0 + (g1 - g0) / PD * PD;
          for (int gb = g0; gb < g_main; gb += PD) {
#pragma unroll
            for (int d = 0; d < PD; ++d) stagec(gb + d, ck[d], cl4[d], true);
          }
#pragma unroll
          for (int d = 0; d < PD; ++d)
            if (g_main + d < g1) stagec(g_main + d, ck[d], cl4[d], false);
          g_first = g1;
        }
      }
      for (int g = g_first; g < g1; ++g) {
        const RJob& rj = s_job[g];
        const uint32_t ids = nx_ids;
        const double2 t0 = nx0, t1 = nx1;
        const uint2 tk = nxk;
        fetch1(g + 1);
        uint32_t out = ids;
        const gptr<uint8_t> __restrict__ dp = dst0 + (size_t)rj.p * n_pad + base;
        // (the particle's split in registers: read through the LDS record, the value, the rule and the labels
        //  were fetched again for every ROW -- the compiler cannot keep LDS reads across the stores of the
        //  reduction -- with a wait for the LDS each time)
        const double r_v = rj.v, r_uscale = rj.uscale;
        const int r_rule = rj.rule;
        const uint32_t r_label = (uint32_t)rj.label, r_new = (uint32_t)rj.new_label;
        if (!rj.active) {  // forced refresh only
          *gcast<uint32_t>(dp) = out;
          continue;
        }
        const double x[RPT] = {t0.x, t0.y, t1.x, t1.y};
        const uint32_t xk[RPT] = {tk.x & 0xFFFFu, tk.x >> 16, tk.y & 0xFFFFu, tk.y >> 16};
        const uint32_t r_vk = (uint32_t)rj.vkey;
        const gptr<const double> __restrict__ xcol = XT + rj.xoff + base;  // (F32: equal keys only)
        // go left?  F32: decided on the order keys unless they are equal (see the template comment)
        auto left_of = [&](int e) -> bool {
          if constexpr (F32) {
            if (xk[e] != r_vk) return r_rule == PGB_RULE_CONTINUOUS ? xk[e] < r_vk : false;
            return go_left_t<SUB>(r_rule, xcol[e], r_v);
          } else {
            return go_left_t<SUB>(r_rule, x[e], r_v);
          }
        };
        const int slot = (g - g0) * NRED;
        if (!rj.check_nan) {  // common case: the split column has no missing values
          long long v0 = 0, v1 = 0, v2 = 0, v3 = 0;  // cnts(L | R<<20), aL, bL, c2L
#pragma unroll
          for (int e = 0; e < RPT; ++e) {
            if (((ids >> (8 * e)) & 255u) == r_label) {
              if (left_of(e)) {
                v0 += 1;
                v1 += qa[e];
                if constexpr (NORMAL) { v2 += qb[e]; v3 += qc[e]; }
              } else {
                out = (out & ~(255u << (8 * e))) | (r_new << (8 * e));
                v0 += 1ll << 20;
              }
            }
          }
          *gcast<uint32_t>(dp) = out;
          if constexpr (!ATOM && (NORMAL || LIN)) {
            const long long tot = wave_sum4(v0, v1, v2, v3);  // lane l: total of value l & 3
            if (lane < 4) s_red[(slot + lane) * 4 + w] = tot;
          } else if constexpr (!ATOM) {
            const long long tot = wave_sum2(v0, v1);  // lane l: total of value l & 1
            if (lane < 2) s_red[(slot + lane) * 4 + w] = tot;
          } else if (v0 != 0) {  // (lanes that hold a row of the leaf)
            // per-row families: the weights come from the likelihood pass, a split only needs the children's counts
            // and the left child's sum of sum_trees -- two values.  (Measured and dropped, round 4: counts as wave
            // votes with a one-value reduction, 20.8 -> 22.9 us at cfg4: the votes serialise on the scalar unit;
            // one loop for both NaN cases: 21.9 us.)
            unsigned long long* ap = &s_acc[(g - g0) * (NVA * NE) + (lane & (NE - 1))];
            atomicAdd(ap, (unsigned long long)v0);
            atomicAdd(ap + NE, (unsigned long long)v1);
            if constexpr (NORMAL) {
              atomicAdd(ap + 2 * NE, (unsigned long long)v2);
              atomicAdd(ap + 3 * NE, (unsigned long long)v3);
            }
          }
        } else {
          long long v[7] = {0, 0, 0, 0, 0, 0, 0};  // cnts(L | R<<20 | N<<40), aL, bL, c2L, aN, bN, c2N
#pragma unroll
          for (int e = 0; e < RPT; ++e) {
            if (((ids >> (8 * e)) & 255u) == r_label) {
              const bool missing = F32 ? (xk[e] == 0xFFFFu) : (x[e] != x[e]);  // (the key of a missing value)
              if (missing) {
                out = (out & ~(255u << (8 * e))) | ((uint32_t)PGB_ORPHAN << (8 * e));
                v[0] += 1ll << 40;
                v[4] += qa[e]; v[5] += qb[e]; v[6] += qc[e];
              } else if (left_of(e)) {
                v[0] += 1;
                v[1] += qa[e]; v[2] += qb[e]; v[3] += qc[e];
              } else {
                out = (out & ~(255u << (8 * e))) | (r_new << (8 * e));
                v[0] += 1ll << 20;
              }
            }
          }
          *gcast<uint32_t>(dp) = out;
          const long long ta = wave_sum4(v[0], v[1], v[2], v[3]);
          const long long tb = wave_sum4(v[4], v[5], v[6], 0);
          if (lane < 4) s_red[(slot + lane) * 4 + w] = ta;
          else if (lane < 7) s_red[(slot + lane) * 4 + w] = tb;  // lane 4..6: value (lane & 3) of the second set
        }
        if constexpr (LIN) {  // sums of u = x 2^-ex over the two children (see pgb_lin_fit)
          long long ul[4] = {0, 0, 0, 0}, ur[4] = {0, 0, 0, 0};
#pragma unroll
          for (int e = 0; e < RPT; ++e) {
            const double xv = x[e];
            if (((ids >> (8 * e)) & 255u) == r_label && xv == xv) {
              const double uu = xv * r_uscale;
              const long long q0 = pgb_quant(uu * S.lin_R, c1, nullptr);
              const long long q1 = pgb_quant((uu * uu) * S.lin_R, c1, nullptr);
              const long long q2 = pgb_quant(uu * strow[e], c1, nullptr);
              const long long q3 = pgb_quant(uu * rrow[e], c1, nullptr);
              const bool gl = go_left_t<SUB>(r_rule, xv, r_v);
              ul[0] += gl ? q0 : 0; ul[1] += gl ? q1 : 0; ul[2] += gl ? q2 : 0; ul[3] += gl ? q3 : 0;
              ur[0] += gl ? 0 : q0; ur[1] += gl ? 0 : q1; ur[2] += gl ? 0 : q2; ur[3] += gl ? 0 : q3;
            }
          }
          const long long tl = wave_sum4(ul[0], ul[1], ul[2], ul[3]);
          const long long tr = wave_sum4(ur[0], ur[1], ur[2], ur[3]);
          if (lane < 4) {
            s_red[(slot + 7 + lane) * 4 + w] = tl;
            s_red[(slot + 11 + lane) * 4 + w] = tr;
          }
        }
      }
      __syncthreads();
      // one thread per (particle of the group, statistic): combine the 4 waves, publish
      for (int t = tid; t < (g1 - g0) * NRED; t += BT) {
        const int gi = t / NRED, i = t % NRED;
        const RJob& rj = s_job[g0 + gi];
        if (!rj.active || (i >= 4 && i < 7 && !rj.check_nan)) continue;
        if constexpr (!NORMAL && !LIN)
          if (i == 2 || i == 3 || i == 5 || i == 6) continue;  // (no residual algebra: nothing was reduced)
        long long s;
        if (ATOM && !rj.check_nan) {  // (i < NVA here: the other values were skipped above)
          unsigned long long* ap = &s_acc[(gi * NVA + i) * NE];
          unsigned long long u = 0;
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            u += ap[e];
            ap[e] = 0ull;
          }
          s = (long long)u;
        } else {
          s = s_red[t * 4] + s_red[t * 4 + 1] + s_red[t * 4 + 2] + s_red[t * 4 + 3];
        }
        if constexpr (LIN) {
          if (i >= 7) {
            AccU* au = &S.accu[((size_t)par * MAXP + rj.p) * ACC_PER + (chunk & (ACC_SLOTS - 1)) * ACC_STRIDE];
            if (s != 0) atomicAdd((unsigned long long*)(i < 11 ? &au->uL[i - 7] : &au->uR[i - 11]), (unsigned long long)s);
            continue;
          }
        }
        Acc* a = &S.acc[((size_t)par * MAXP + rj.p) * ACC_PER + (chunk & (ACC_SLOTS - 1)) * ACC_STRIDE];
        if (i == 0) {
          const int cL = (int)(s & 0xFFFFF), cR = (int)((s >> 20) & 0xFFFFF), cN = (int)(s >> 40);
          ccp[(size_t)rj.ccL * (F32 ? S.cc_stride : S.nchunks) + chunk] = (uint16_t)cL;
          ccp[(size_t)rj.ccR * (F32 ? S.cc_stride : S.nchunks) + chunk] = (uint16_t)cR;
          if (cL | cN) atomicAdd(&a->cnts, (unsigned long long)cL | ((unsigned long long)cN << 32));
        } else if (s != 0) {
          long long* dst = i == 1 ? &a->aL : i == 2 ? &a->bL : i == 3 ? &a->c2L : i == 4 ? &a->aN : i == 5 ? &a->bN : &a->c2N;
          atomicAdd((unsigned long long*)dst, (unsigned long long)s);
        }
      }
      __syncthreads();
    }
    TRR(15, 0);  // item loop done
    if (do_init) {  // statistics of the INIT (+FINAL) part, accumulated by the writer groups only
      block_sum<5>(iv, s_red);
      if (sat) atomicAdd(&S.counters[4], (unsigned long long)sat);
      if (tid == 0) {
        InitAcc* a = &S.initacc[(size_t)par * IA_SLOTS + (blockIdx.x % IA_SLOTS)];
        if (iv[0]) atomicAdd((unsigned long long*)&a->A, (unsigned long long)iv[0]);
        if (iv[1]) atomicAdd((unsigned long long*)&a->B, (unsigned long long)iv[1]);
        if (iv[2]) atomicAdd((unsigned long long*)&a->C, (unsigned long long)iv[2]);
        if (iv[3]) atomicAdd((unsigned long long*)&a->E0, (unsigned long long)iv[3]);
        if (iv[4]) atomicAdd((unsigned long long*)&a->QSTD, (unsigned long long)iv[4]);
      }
    }
    PROF_END();
    return;
  }

  // ---------------- lone FINAL (last tree of the last requested step): 256 rows per item
  __syncthreads();
  long long v[5] = {0, 0, 0, 0, 0};
  unsigned sat = 0;
  const int nitems = (int)(S.n_pad / BT);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const long long row = (long long)item * BT + tid;
    if (row >= S.n) continue;
    double st = st_in[row];
    uint32_t id_sel;
    if (cmd->sel_slot == -2) {
      id_sel = tl_old[row];  // old tree kept
    } else {
      id_sel = sel_lid ? (uint32_t)sel_lid[row] : 0u;  // untouched root: label 0
      tl_old[row] = (uint8_t)id_sel;
    }
    // [U] sum_trees = sum_trees_noi + new_tree.predict()
    double nv = s_lv[0][id_sel];
    if constexpr (LIN) {
      const LinP lp = s_ll[0][id_sel];
      if (lp.svar >= 0) nv = pgb_leaf_pred(nv, lp.slope, lp.xbar, S.XT[(size_t)lp.svar * S.n_pad + row]);
    }
    st = st + nv;
    if (cmd->tune) {  // [U] RunningSd.update (Welford)
      const double mean0 = rs_mean[row], m20 = rs_m2[row];
      const double delta = nv - mean0;
      const double mean = mean0 + delta / cntf;
      const double delta2 = nv - mean;
      const double m2 = m20 + delta * delta2;
      rs_mean[row] = mean;
      rs_m2[row] = m2;
      v[4] += pgb_quant(PGB_SQRT(m2 / cntf), S.sc.c1, &sat);
    }
    st_out[row] = st;
  }
  block_sum<5>(v, s_red);
  if (sat) atomicAdd(&S.counters[4], (unsigned long long)sat);
  if (tid == 0 && cmd->tune && v[4]) {
    InitAcc* a = &S.initacc[(size_t)par * IA_SLOTS + (blockIdx.x % IA_SLOTS)];
    atomicAdd((unsigned long long*)&a->QSTD, (unsigned long long)v[4]);
  }
  PROF_END();
#undef PROF_END
}

