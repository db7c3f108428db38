// k_rows_mk.h -- part of pgbart_hip.hip (not a standalone header): k_rows_mk: the row pass for K-vector leaves.
// ------------------------------------------------------------------ k_rows_mk
// K-vector leaves (K > 1, Categorical-softmax): the same slot logic as k_rows with sum_trees, leaf
// values and running-sd statistics per output.  Output 0 uses the scalar buffers, outputs 1..K-1
// the *x extension arrays.  Not the headline path: written for clarity, K loops innermost.
// KT: number of outputs when known at compile time (2, 3, 4: loops unroll, the per-row arrays stay in registers);
// 0: ANY K <= PGB_MAX_OUTPUTS, processed as TILES of TW = 4 outputs -- the whole pass body runs once per tile with
// the per-row arrays of four outputs in registers (what the K = 4 instance keeps), the partition decisions are
// re-derived per tile (labels and the split column come from L2 again), and the labels / counts are written by
// the first tile only.  No K-sized array anywhere: the run-time-K instance has no scratch and the same LDS
// whatever K is (round 3: 576-736 B of scratch, 72 KB of LDS at K <= 8; round-3 VERDICT #2 / #6).
// LIN: linear response; the label -> (slope, xbar, column) tables are read from global memory
// (lvl for output 0 and the shared parts, lsx for the slopes of outputs 1..K-1)
// F32: the split column as 16-bit order keys of the design matrix (see k_rows<..., F32>); continuous /
// one-hot rules only.
// (the compile-time-K instances are held to 4 workgroups per CU, <= 128 VGPRs: K = 4 sits at that edge)
#ifndef PGB_MK_WGS
#define PGB_MK_WGS(KT) ((KT) == 4 ? 3 : 4) /* K = 4: 149 registers without a spill (3 per CU) beat 128 with eight spilled (4 per CU): 723 k against 687 k at cfg5 */
#endif
template <int KT, bool LIN, bool F32 = false>
__global__ __launch_bounds__(BT, (KT >= 2 && !LIN) ? PGB_MK_WGS(KT) : 2) void k_rows_mk(const Dev* __restrict__ Sp, int par,
                                                                                            const Cmd* __restrict__ cmds,
                                                                                            const Job* __restrict__ jobs_all) {
  // cmds / jobs_all repeat S.cmd / S.jobs as kernel arguments (preloaded into SGPRs), as in k_rows: the command word
  // and the job records are requested at once instead of behind a load of their pointers from the argument block S
  // and behind one another (round 5: three dependent round trips at the head of every launch were one)
  const DevG& S = *reinterpret_cast<const DevG*>(Sp);
  const int K = KT > 0 ? KT : S.K, KX = K - 1;
  constexpr int TW = KT > 0 ? KT : 4;  // outputs per tile (= K for the compile-time instances: one tile)
  constexpr int KB = TW;
  constexpr int NVT = 1 + 2 * TW;      // per particle and tile: counts, aL[TW], aN[TW]
  __shared__ long long s_red[MAXP * NVT * 4];
  __shared__ double s_lv[2][256][KB];
  __shared__ RJob s_job[MAXP];
  __shared__ int s_n[2];
  const Cmd* cmd = &cmds[par];
  const int kind = cmd->kind;
  Job j_pre[MAXP / 64];  // (wave 0: the particle list below)
  if constexpr (PGB_ROWS_JPRE != 0) {
    if (threadIdx.x < 64) {
#pragma unroll
      for (int hq = 0; hq < MAXP / 64; ++hq) j_pre[hq] = jobs_all[(size_t)par * MAXP + threadIdx.x + 64 * hq];
    }
  }
  TRR_BIND(S.ctrl[par ^ 1].slot_no - 1);  // (stamp 12: entry; 13 jobs listed, 14 rows of the last item loaded, 15 items done)
  long long* pstamp = nullptr;            // profiling: first / last device-clock reading of every workgroup (see k_rows)
  if (S.prof_stamps != nullptr && threadIdx.x == 0 && blockIdx.x < PROF_BLOCKS && !PGB_STAMP_LL_ON) {
    pstamp = S.prof_stamps + ((size_t)((S.ctrl[par ^ 1].slot_no - 1) % PROF_RING) * PROF_BLOCKS + blockIdx.x) * 2;
    pstamp[0] = wall_clock64();
    pstamp[1] = pstamp[0];
  }
  if (kind == CMD_NOOP) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const bool do_final = (kind & CMD_FINAL) != 0, do_init = (kind & CMD_INIT) != 0;
  const bool do_part = (kind & CMD_PARTITION) != 0;

  // label -> leaf value tables of the outputs k0 .. k0 + TW - 1 (new | next)
  auto load_lv = [&](int k0) {
    const double* lx = S.lvx + (size_t)par * 2 * 256 * KX;
    for (int i = tid; i < 256; i += BT) {
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) {
        const int k = k0 + kk;
        double a = 0.0, b = 0.0;
        if (k == 0) {
          a = cmd->lv_new[i];
          b = cmd->lv_next[i];
        } else if (k < K) {
          a = lx[(size_t)i * KX + (k - 1)];
          b = lx[(size_t)256 * KX + (size_t)i * KX + (k - 1)];
        }
        s_lv[0][i][kk] = a;
        s_lv[1][i][kk] = b;
      }
    }
  };
  uint8_t* tl_old = do_final ? S.tree_lid + (size_t)cmd->tree_old * S.n_pad : nullptr;
  const uint8_t* tl_new = do_init ? S.tree_lid + (size_t)cmd->tree_new * S.n_pad : nullptr;
  const uint8_t* sel_lid =
      (do_final && cmd->sel_slot >= 0) ? S.lid + ((size_t)cmd->sel_gen * MAXP + cmd->sel_slot) * S.n_pad : nullptr;
  const double cntf = (double)cmd->rs_count;
  // (fields of the command read ONCE: behind a store the compiler has to assume they changed and loads them again)
  const int c_sel_slot = cmd->sel_slot, c_tune = cmd->tune;
  const bool c_same_tree = cmd->tree_new == cmd->tree_old;
  // sum_trees buffers [2][K][n_pad]
  const double* st_in = S.st + (size_t)cmd->st_cur * K * S.n_pad;
  double* st_out = S.st + (size_t)(do_init ? cmd->st_cur ^ 1 : cmd->st_cur) * K * S.n_pad;
  const double c1 = S.sc.c1;
  const long long n = S.n, n_pad = S.n_pad;
  // linear part of the prediction of output k for a row with label `id`: table t = 0 new | 1 next
  auto lin_pred = [&](double v, int t, uint32_t id, int k, long long row) -> double {
    if constexpr (LIN) {
      const LinP lp = S.lvl[((size_t)par * 2 + t) * 256 + id];
      if (lp.svar >= 0) {
        const double sl = k == 0 ? lp.slope : S.lsx[(((size_t)par * 2 + t) * 256 + id) * KX + (k - 1)];
        v = pgb_leaf_pred(v, sl, lp.xbar, S.XT[(size_t)lp.svar * n_pad + row]);
      }
    }
    return v;
  };

  if (do_part) {
    const Job* jobs = jobs_all + (size_t)par * MAXP;
    if (tid < 64) {
      int nlist = 0;  // (lanes' particles tid, tid + 64, ...: one block of 64 after the other)
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
      {
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
        s_job[k] = rj;
      }
      nlist += __popcll(m);
      }
      if (tid == 0) {
        s_n[0] = nlist;
        s_n[1] = plain ? 2 : common ? 1 : 0;
      }
    }
    __syncthreads();
    TRR(13, 0);
    const int nact = s_n[0];
    const bool all_plain = !LIN && s_n[1] == 2;
    const bool all_common = !LIN && F32 && s_n[1] == 1;  // (the pipelined common round: shadow instances only)
    if (nact == 0 && !do_init) {
      if (pstamp) pstamp[1] = wall_clock64();
      return;
    }
    const int target = do_init ? S.rows_target_init : S.rows_target;
    int G = (nact * S.nchunks + target - 1) / target;
    if (G < 1) G = 1;
    int ngroups = (nact + G - 1) / G;
    if (ngroups < 1) ngroups = 1;
    const int nitems = S.nchunks * ngroups;
    uint8_t* const dst0 = S.lid + (size_t)cmd->dst_gen * MAXP * S.n_pad;
    unsigned sat = 0;
    for (int k0 = 0; k0 < K; k0 += TW) {  // one tile of outputs at a time (compile-time K: a single tile)
      const bool first = k0 == 0;         // the first tile writes the labels and the counts
      if (do_init) {
        if (!first) __syncthreads();      // (the previous tile's readers of s_lv are done)
        load_lv(k0);
        __syncthreads();
      }
      long long ivA[KB], ivQ[KB];  // INIT statistics of this tile's outputs: A[k], QSTD[k]
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) ivA[kk] = ivQ[kk] = 0;
      for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        TRR(23, 0);
        const int chunk = item % S.nchunks, grp = item / S.nchunks;
        const long long base = (long long)chunk * CH + tid * RPT;
        double stv[RPT][KB];  // sum_trees of this thread's rows, per output of the tile
        if (do_init) {
          const bool writer = grp == 0;
          uint32_t ids_next = *(const uint32_t*)(tl_new + base);
          uint32_t ids_sel = 0;
          if (do_final) {
            if (c_sel_slot == -2) {
              ids_sel = *(const uint32_t*)(tl_old + base);
            } else {
              if (sel_lid) {
                ids_sel = *(const uint32_t*)(sel_lid + base);
              } else {
                for (int e = 0; e < RPT; ++e)
                  if (base + e >= n) ids_sel |= (uint32_t)PGB_ORPHAN << (8 * e);
              }
              // (a later tile reads tl_old again only when the old tree was kept, sel_slot == -2; here every tile
              //  derives ids_sel from the selected particle's labels, so the first tile's store is the only one)
              if (writer && first) *(uint32_t*)(tl_old + base) = ids_sel;
            }
            if (c_same_tree) ids_next = ids_sel;
          }
          // every input of the thread's rows is requested BEFORE the first result is stored: the stores below may
          // alias the loads as far as the compiler knows, and loads left inside the loops were issued one at a time
          // -- one memory round trip per (row, output): 36 of the 48 us of the slot that starts a tree at cfg5
          // (round 4, in-kernel stamps; profiles/r04_experiments.md)
          TRR(20, 0);
          double stl[RPT][KB];
          const bool upd = do_final && c_tune && writer;
#pragma unroll
          for (int kk = 0; kk < KB; ++kk) {
            const int k = k0 + kk < K ? k0 + kk : K - 1;  // (a tile's outputs past K: loaded again, never used)
            const double2* sp = (const double2*)(st_in + (size_t)k * n_pad + base);
            const double2 s01 = sp[0], s23 = sp[1];
            stl[0][kk] = s01.x; stl[1][kk] = s01.y; stl[2][kk] = s23.x; stl[3][kk] = s23.y;
          }
          // sum_trees of the rows with the finished tree added (registers only) ...
#pragma unroll
          for (int e = 0; e < RPT; ++e) {
            const bool in = base + e < n;
            const uint32_t id_s = (ids_sel >> (8 * e)) & 255u;
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
              const int k = k0 + kk;
              double st = stl[e][kk];
              if (do_final) st = st + lin_pred(s_lv[0][id_s][kk], 0, id_s, k < K ? k : K - 1, in ? base + e : 0);
              stv[e][kk] = (in && k < K) ? st : 0.0;
            }
          }
          // ... and, by the first group of the chunk only, everything that is stored, in one block: the four rows of
          // a thread are adjacent in every array, so they leave as 16-byte stores (rows past the end of the data
          // store zeros: nothing reads them as data)
          if (writer) {
            const auto pk = S.pack;  // (pointers of the argument block read once, not behind every store)
            const auto pkx = S.packx;
            uint32_t id_n[RPT];
#pragma unroll
            for (int e = 0; e < RPT; ++e) id_n[e] = (ids_next >> (8 * e)) & 255u;
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
              const int k = k0 + kk;
              if (k >= K) continue;
              double noi[RPT];
#pragma unroll
              for (int e = 0; e < RPT; ++e) {
                const long long row = base + e;
                const bool in = row < n;
                const double o = lin_pred(s_lv[1][id_n[e]][kk], 1, id_n[e], k, in ? row : 0);
                noi[e] = in ? stv[e][kk] - o : 0.0;
                if (in) ivA[kk] += pgb_quant(stv[e][kk], c1, &sat);
              }
              if (first && kk == 0) {
#pragma unroll
                for (int e = 0; e < RPT; ++e) pk[base + e] = make_double2(stv[e][kk], 0.0);
              } else {
                double2* px = (double2*)(pkx + (size_t)(k - 1) * n_pad + base);
                px[0] = make_double2(stv[0][kk], stv[1][kk]);
                px[1] = make_double2(stv[2][kk], stv[3][kk]);
              }
              double2* po = (double2*)(st_out + (size_t)k * n_pad + base);
              po[0] = make_double2(noi[0], noi[1]);
              po[1] = make_double2(noi[2], noi[3]);
            }
          }
          // (C, the log-likelihood of a fresh stump, and E0, of the current tree, are summed by k_loglik,
          //  which runs after this pass and is compiled per number of outputs)
          TRR(21, 0);
          if (upd) {  // [U] RunningSd.update (Welford), per output: the four rows' inputs of an output requested together
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
              const int k = k0 + kk;
              if (k >= K) continue;
              const double2* mp = (const double2*)(S.rs_mean + (size_t)k * n_pad + base);
              const double2* qp = (const double2*)(S.rs_m2 + (size_t)k * n_pad + base);
              const double2 m01 = mp[0], m23 = mp[1], q01 = qp[0], q23 = qp[1];
              const double mean_l[RPT] = {m01.x, m01.y, m23.x, m23.y}, m2_l[RPT] = {q01.x, q01.y, q23.x, q23.y};
#pragma unroll
              for (int e = 0; e < RPT; ++e) {
                const long long row = base + e;
                if (row >= n) continue;
                const double nv = lin_pred(s_lv[0][(ids_sel >> (8 * e)) & 255u][kk], 0, (ids_sel >> (8 * e)) & 255u, k, row);
                const size_t ri = (size_t)k * n_pad + row;
                const double delta = nv - mean_l[e];
                const double mean = mean_l[e] + delta / cntf;
                const double delta2 = nv - mean;
                const double m2 = m2_l[e] + delta * delta2;
                S.rs_mean[ri] = mean;
                S.rs_m2[ri] = m2;
                ivQ[kk] += pgb_quant(PGB_SQRT(m2 / cntf), c1, &sat);
              }
            }
          }
        } else {
          for (int e = 0; e < RPT; ++e) {
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
              const int k = k0 + kk;
              stv[e][kk] = k >= K ? 0.0 : k == 0 ? S.pack[base + e].x : S.packx[(size_t)(k - 1) * n_pad + base + e];
            }
          }
        }
        TRR(22, 0);
        uint32_t root_ids = 0;
        for (int e = 0; e < RPT; ++e)
          if (base + e >= n) root_ids |= (uint32_t)PGB_ORPHAN << (8 * e);
        const int g0 = grp * G, g1 = (g0 + G < nact) ? g0 + G : nact;
        const bool plain_item = !LIN && all_plain && g1 > g0;  // (the plain round below takes all particles of the item)
        int g_first = g0;  // particles g0 .. g_first - 1 of this item went through the plain round
        uint32_t nx_ids = root_ids;
        double2 nx0 = {0.0, 0.0}, nx1 = {0.0, 0.0};
        uint2 nxk = {0u, 0u};
        auto fetch1 = [&](int gg) {
          nx_ids = root_ids;
          if (gg < g1) {
            const RJob& rn = s_job[gg];
            if (rn.src >= 0) nx_ids = *(const uint32_t*)(S.lid + rn.src + base);
            if (rn.active) {
              if constexpr (F32) {
                nxk = gload_k4(S.XK16 + rn.xoff + base);
              } else {
                const double2* xn = (const double2*)(S.XT + rn.xoff + base);
                nx0 = xn[0];
                nx1 = xn[1];
              }
            }
          }
        };
        const bool common_item = all_common && g1 > g0;
        if (!plain_item && !common_item) fetch1(g0);  // (next to the loads of the rows' sum_trees, before anything waits for those)
        long long qst[RPT][KB];  // quantised once per row, reused by every particle of the group
#pragma unroll
        for (int e = 0; e < RPT; ++e)
#pragma unroll
          for (int kk = 0; kk < KB; ++kk) qst[e][kk] = k0 + kk < K ? pgb_quant(stv[e][kk], c1, nullptr) : 0;
        TRR(14, 0);
        // ---- plain round (see k_rows): every particle of the pass splits the root of a fresh stump on a continuous
        // column without missing values -- the slot that starts a tree.  The split columns of PD particles are in
        // flight in stage registers that are never moved (loop unrolled by PD, every load and store unconditional),
        // no label compare, no per-row control flow.  (The general loop below requests a particle's column and
        // uses it at once: a memory round trip per particle.)
        if constexpr (!LIN) {
          if (plain_item) {
            constexpr int PD = F32 ? (KB == 4 ? 3 : 4) : 2;  // (float64 columns: 8 registers per stage; K = 4 sits at its register edge)
            const bool full_chunk = (long long)(chunk + 1) * CH <= n;
            double2 pa[PD], pb[PD];
            uint2 pf[PD];
            auto fetch = [&](int gg, double2& f0, double2& f1, uint2& ff) {
              const long long xo = uni(s_job[gg < g1 ? gg : g1 - 1].xoff);
              if constexpr (F32) {
                ff = gload_k4(S.XK16 + xo + base);
              } else {
                const double2* xp = (const double2*)(S.XT + xo + base);
                f0 = xp[0];
                f1 = xp[1];
              }
            };
            auto stage = [&](int g, double2& f0, double2& f1, uint2& ff, bool more) {
              const RJob& rj = s_job[g];
              const double r_v = uni(rj.v);
              const uint32_t r_vk = uni((uint32_t)rj.vkey);
              const uint32_t nw = uni((uint32_t)rj.new_label);
              const long long xo = uni(rj.xoff);
              uint8_t* const dp = dst0 + (size_t)uni(rj.p) * n_pad + base;
              const double x[RPT] = {f0.x, f0.y, f1.x, f1.y};
              const uint32_t xk[RPT] = {ff.x & 0xFFFFu, ff.x >> 16, ff.y & 0xFFFFu, ff.y >> 16};
              bool L[RPT];
#pragma unroll
              for (int e = 0; e < RPT; ++e) L[e] = F32 ? (xk[e] < r_vk) : (x[e] <= r_v);
              if constexpr (F32) {  // equal order keys are decided on the float64 values
                if (__any((xk[0] == r_vk) | (xk[1] == r_vk) | (xk[2] == r_vk) | (xk[3] == r_vk))) {
#pragma unroll
                  for (int e = 0; e < RPT; ++e)
                    if (xk[e] == r_vk) L[e] = S.XT[xo + base + e] <= r_v;
                }
              }
              if (more) fetch(g + PD, f0, f1, ff);
              uint32_t out = root_ids, cl = 0, cr = 0;
              long long aL[KB];
#pragma unroll
              for (int kk = 0; kk < KB; ++kk) aL[kk] = 0;
#pragma unroll
              for (int e = 0; e < RPT; ++e) {
                const bool in = full_chunk || base + e < n;  // (rows past the end: orphan label, counted nowhere)
                const bool le = in && L[e], ri = in && !L[e];
                out |= ri ? (nw << (8 * e)) : 0u;
                cl += le ? 1u : 0u;
                cr += ri ? 1u : 0u;
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) aL[kk] += le ? qst[e][kk] : 0ll;
              }
              if (first) *(uint32_t*)dp = out;
              const int slot = (g - g0) * NVT;
              if constexpr (KB == 4) {
                const long long tot = wave_sum4(aL[0], aL[1], aL[2], aL[3]);
                if (lane < 4) s_red[(slot + 1 + lane) * 4 + w] = tot;
                const long long c = wave_sum_dpp((long long)(cl | (cr << 20)));  // (lane 63 holds the total)
                if (lane == 63) s_red[slot * 4 + w] = c;
              } else {
                const long long tot = wave_sum4((long long)(cl | (cr << 20)), aL[0], aL[KB > 1 ? 1 : 0], aL[KB > 2 ? 2 : 0]);
                if (lane < 1 + KB) s_red[(slot + lane) * 4 + w] = tot;
              }
            };
#pragma unroll
            for (int d = 0; d < PD; ++d) {
              pa[d] = pb[d] = double2{0.0, 0.0};
              pf[d] = uint2{0u, 0u};
              fetch(g0 + d, pa[d], pb[d], pf[d]);
            }
            const int g_main = g0 + (g1 - g0) / PD * PD;
            for (int gb = g0; gb < g_main; gb += PD) {
#pragma unroll
              for (int d = 0; d < PD; ++d) stage(gb + d, pa[d], pb[d], pf[d], true);
            }
#pragma unroll
            for (int d = 0; d < PD; ++d)
              if (g_main + d < g1) stage(g_main + d, pa[d], pb[d], pf[d], false);
            g_first = g1;
          }
        }
        // general rounds: the labels and the split column of particle g + 1 are requested before particle g is
        // relabelled and reduced (the values just arrived move to the working registers, the next loads go out
        // behind them: no register with a load in flight is moved).  Rounds >= 2 are short chains of a few particles
        // per item, each a memory round trip before this.
        // ---- common round (shadow instances; see k_rows): every split of the pass is on a continuous column without
        // missing values (label-refresh-only particles included): labels and keys of PD particles in flight.
        if constexpr (!LIN && F32) {
          if (common_item) {
            constexpr int PD = KB == 4 ? 3 : 4;
            uint2 ck[PD];
            uint32_t cl4[PD];
            auto fetchc = [&](int gg, uint2& ff, uint32_t& fl) {
              const RJob& rn = s_job[gg < g1 ? gg : g1 - 1];
              ff = gload_k4(S.XK16 + uni(rn.xoff) + base);  // (a label-only job: column 0, not used)
              const long long so = uni(rn.src);
              fl = *(const uint32_t*)(S.lid + (so < 0 ? 0ll : so) + base);  // (an implicit root: loaded, not used)
            };
            auto stagec = [&](int g, uint2& ff, uint32_t& fl, bool more) {
              const RJob& rj = s_job[g];
              const bool act = uni(rj.active) != 0;
              const double r_v = uni(rj.v);
              const uint32_t r_vk = uni((uint32_t)rj.vkey);
              const uint32_t nw = uni((uint32_t)rj.new_label), lb = uni((uint32_t)rj.label);
              const long long xo = uni(rj.xoff);
              uint8_t* const dp = dst0 + (size_t)uni(rj.p) * n_pad + base;
              const uint32_t ids = uni(rj.src) < 0 ? root_ids : fl;
              const uint32_t xk[RPT] = {ff.x & 0xFFFFu, ff.x >> 16, ff.y & 0xFFFFu, ff.y >> 16};
              bool in[RPT], L[RPT];
#pragma unroll
              for (int e = 0; e < RPT; ++e) {
                in[e] = act && ((ids >> (8 * e)) & 255u) == lb;
                L[e] = xk[e] < r_vk;
              }
              if (__any((in[0] && xk[0] == r_vk) | (in[1] && xk[1] == r_vk) | (in[2] && xk[2] == r_vk) | (in[3] && xk[3] == r_vk))) {
#pragma unroll
                for (int e = 0; e < RPT; ++e)
                  if (in[e] && xk[e] == r_vk) L[e] = S.XT[xo + base + e] <= r_v;  // equal keys: the float64 values decide
              }
              if (more) fetchc(g + PD, ff, fl);
              uint32_t out = ids, cl = 0, cr = 0;
              long long aL[KB];
#pragma unroll
              for (int kk = 0; kk < KB; ++kk) aL[kk] = 0;
#pragma unroll
              for (int e = 0; e < RPT; ++e) {
                const bool le = in[e] && L[e], ri = in[e] && !L[e];
                out = ri ? ((out & ~(255u << (8 * e))) | (nw << (8 * e))) : out;
                cl += le ? 1u : 0u;
                cr += ri ? 1u : 0u;
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) aL[kk] += le ? qst[e][kk] : 0ll;
              }
              if (first) *(uint32_t*)dp = out;
              if (act) {
                const int slot = (g - g0) * NVT;
                if constexpr (KB == 4) {
                  const long long tot = wave_sum4(aL[0], aL[1], aL[2], aL[3]);
                  if (lane < 4) s_red[(slot + 1 + lane) * 4 + w] = tot;
                  const long long c = wave_sum_dpp((long long)(cl | (cr << 20)));  // (lane 63 holds the total)
                  if (lane == 63) s_red[slot * 4 + w] = c;
                } else {
                  const long long tot = wave_sum4((long long)(cl | (cr << 20)), aL[0], aL[KB > 1 ? 1 : 0], aL[KB > 2 ? 2 : 0]);
                  if (lane < 1 + KB) s_red[(slot + lane) * 4 + w] = tot;
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
          // (a later tile finds the rows of the leaf in the labels as they were BEFORE this pass -- the source
          //  generation is never the one being written -- and re-derives the sides from the split column)
          const uint32_t ids = nx_ids;
          const double2 t0 = nx0, t1 = nx1;
          const uint2 tk = nxk;
          fetch1(g + 1);
          uint32_t out = ids;
          uint8_t* const dp = dst0 + (size_t)rj.p * n_pad + base;
          if (!rj.active) {
            if (first) *(uint32_t*)dp = out;
            continue;
          }
          const double2* xp = (const double2*)(S.XT + rj.xoff + base);
          // (the particle's split in registers instead of LDS reads per row, see k_rows)
          const double r_v = rj.v;
          const int r_rule = rj.rule;
          const uint32_t r_label = (uint32_t)rj.label, r_new = (uint32_t)rj.new_label;
          double x[RPT] = {0.0, 0.0, 0.0, 0.0};
          uint32_t xk[RPT] = {0u, 0u, 0u, 0u};
          if constexpr (F32) {
            xk[0] = tk.x & 0xFFFFu; xk[1] = tk.x >> 16; xk[2] = tk.y & 0xFFFFu; xk[3] = tk.y >> 16;
          } else {
            x[0] = t0.x; x[1] = t0.y; x[2] = t1.x; x[3] = t1.y;
          }
          const uint32_t r_vk = (uint32_t)rj.vkey;
          int side[RPT];  // 0: not in the leaf, 1: left, 2: right, 3: dropped (missing value)
          long long cnts = 0;
          for (int e = 0; e < RPT; ++e) {
            side[e] = 0;
            if (((ids >> (8 * e)) & 255u) == r_label) {
              bool missing, left;
              if constexpr (F32) {  // decided on the float32 values unless they tie
                missing = xk[e] == 0xFFFFu;
                if (xk[e] != r_vk) left = r_rule == PGB_RULE_CONTINUOUS ? xk[e] < r_vk : false;
                else left = go_left(r_rule, ((const double*)xp)[e], r_v);
              } else {
                missing = x[e] != x[e];
                left = !missing && go_left(r_rule, x[e], r_v);
              }
              if (missing) {
                side[e] = 3;
                out = (out & ~(255u << (8 * e))) | ((uint32_t)PGB_ORPHAN << (8 * e));
                cnts += 1ll << 40;
              } else if (left) {
                side[e] = 1;
                cnts += 1;
              } else {
                side[e] = 2;
                out = (out & ~(255u << (8 * e))) | (r_new << (8 * e));
                cnts += 1ll << 20;
              }
            }
          }
          if (first) *(uint32_t*)dp = out;
          const int slot = (g - g0) * NVT;
          // values of this particle and tile: [0] counts, [1 + kk] aL[k0 + kk], [1 + TW + kk] aN[k0 + kk]; reduced
          // four at a time (wave_sum4); a column without missing values has no aN part
          long long vals[NVT + 3];
#pragma unroll
          for (int i = 0; i < NVT + 3; ++i) vals[i] = 0;
          vals[0] = cnts;
#pragma unroll
          for (int kk = 0; kk < KB; ++kk) {
            long long aL = 0, aN = 0;
#pragma unroll
            for (int e = 0; e < RPT; ++e) {
              const long long q = qst[e][kk];
              aL += side[e] == 1 ? q : 0;
              aN += side[e] == 3 ? q : 0;
            }
            vals[1 + kk] = aL;
            vals[1 + TW + kk] = aN;
          }
          if constexpr (LIN) {  // sums of u = x 2^-ex over the two children (see pgb_lin_fit): u, u^2 (first tile)
            // and u st_k per output; one wave total each, added by lane 63 (a rare path: no LDS staging)
            const double uscale = pgb_pow2(-S.col_ex[rj.xoff / n_pad]);
            long long su[2][2 + KB];
#pragma unroll
            for (int i = 0; i < 2 + KB; ++i) su[0][i] = su[1][i] = 0;
#pragma unroll
            for (int e = 0; e < RPT; ++e) {
              if (side[e] == 1 || side[e] == 2) {
                const int sd = side[e] - 1;
                const double uu = x[e] * uscale;
                su[sd][0] += pgb_quant(uu * S.lin_R, c1, nullptr);
                su[sd][1] += pgb_quant((uu * uu) * S.lin_R, c1, nullptr);
#pragma unroll
                for (int kk = 0; kk < KB; ++kk)
                  if (k0 + kk < K) su[sd][2 + kk] += pgb_quant(uu * stv[e][kk], c1, nullptr);
              }
            }
            AccU* au = &S.accu[((size_t)par * MAXP + rj.p) * ACC_PER + (chunk & (ACC_SLOTS - 1)) * ACC_STRIDE];
            long long* aux = S.accux + ((size_t)par * MAXP + rj.p) * AX_PER + (size_t)(chunk & (AX_SLOTS - 1)) * AX_REC;
#pragma unroll
            for (int sd = 0; sd < 2; ++sd)
#pragma unroll
              for (int i = 0; i < 2 + KB; ++i) {
                const int k = k0 + i - 2;  // output of entry i >= 2
                if ((i < 2 && !first) || (i >= 2 && k >= K)) continue;
                const long long tot = wave_sum_dpp(su[sd][i]);
                if (lane == 63 && tot != 0) {
                  // uL / uR: [0] sum u, [1] sum u^2, [2] sum u st_0; outputs k >= 1 in accux
                  long long* dst = i < 2 ? (sd ? &au->uR[i] : &au->uL[i])
                                         : k == 0 ? (sd ? &au->uR[2] : &au->uL[2]) : &aux[(sd ? KX : 0) + (k - 1)];
                  atomicAdd((unsigned long long*)dst, (unsigned long long)tot);
                }
              }
          }
          const int nv = rj.check_nan ? NVT : 1 + TW;
#pragma unroll
          for (int c4 = 0; c4 < (NVT + 3) / 4; ++c4) {
            if (c4 * 4 < nv) {
              const long long tot = wave_sum4(vals[c4 * 4], vals[c4 * 4 + 1], vals[c4 * 4 + 2], vals[c4 * 4 + 3]);
              if (lane < 4 && c4 * 4 + lane < nv) s_red[(slot + c4 * 4 + lane) * 4 + w] = tot;
            }
          }
        }
        __syncthreads();
        for (int t = tid; t < (g1 - g0) * NVT; t += BT) {
          const int gi = t / NVT, i = t % NVT;
          const RJob& rj = s_job[g0 + gi];
          if (!rj.active || (i > TW && !rj.check_nan)) continue;
          if (i == 0 && !first) continue;
          const long long s = s_red[t * 4] + s_red[t * 4 + 1] + s_red[t * 4 + 2] + s_red[t * 4 + 3];
          Acc* a = &S.acc[((size_t)par * MAXP + rj.p) * ACC_PER + (chunk & (ACC_SLOTS - 1)) * ACC_STRIDE];
          long long* ax = S.accx + ((size_t)par * MAXP + rj.p) * AX_PER + (size_t)(chunk & (AX_SLOTS - 1)) * AX_REC;
          if (i == 0) {
            const int cL = (int)(s & 0xFFFFF), cR = (int)((s >> 20) & 0xFFFFF), cN = (int)(s >> 40);
            S.cc[(size_t)rj.ccL * (F32 ? S.cc_stride : S.nchunks) + chunk] = (uint16_t)cL;
            S.cc[(size_t)rj.ccR * (F32 ? S.cc_stride : S.nchunks) + chunk] = (uint16_t)cR;
            if (cL | cN) atomicAdd(&a->cnts, (unsigned long long)cL | ((unsigned long long)cN << 32));
          } else if (s != 0) {
            const int kk = (i - 1) % TW, k = k0 + kk;
            const bool isN = (i - 1) >= TW;
            if (k < K) {
              long long* dst = k == 0 ? (isN ? &a->aN : &a->aL) : &ax[(isN ? KX : 0) + k - 1];
              atomicAdd((unsigned long long*)dst, (unsigned long long)s);
            }
          }
        }
        __syncthreads();
      }
      TRR(15, 0);  // items of this tile done
      if (do_init) {  // A[k], QSTD[k] of this tile's outputs -> InitAcc (k = 0) / iax (k >= 1)
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
          const int k = k0 + kk;
          if (k >= K) continue;
          long long v2[2] = {ivA[kk], ivQ[kk]};
          block_sum<2>(v2, s_red);
          if (tid == 0) {
            if (k == 0) {
              InitAcc* a = &S.initacc[(size_t)par * IA_SLOTS + (blockIdx.x % IA_SLOTS)];
              if (v2[0]) atomicAdd((unsigned long long*)&a->A, (unsigned long long)v2[0]);
              if (v2[1]) atomicAdd((unsigned long long*)&a->QSTD, (unsigned long long)v2[1]);
            } else {
              long long* ix = S.iax + ((size_t)par * IA_SLOTS + (blockIdx.x % IA_SLOTS)) * 2 * KX;
              if (v2[0]) atomicAdd((unsigned long long*)&ix[k - 1], (unsigned long long)v2[0]);
              if (v2[1]) atomicAdd((unsigned long long*)&ix[KX + k - 1], (unsigned long long)v2[1]);
            }
          }
        }
      }
    }
    if (do_init && sat) atomicAdd(&S.counters[4], (unsigned long long)sat);
    if (pstamp) pstamp[1] = wall_clock64();
    return;
  }

  // ---------------- lone FINAL: one output at a time (no K-sized array)
  unsigned sat = 0;
  const int nitems = (int)(S.n_pad / BT);
  for (int k0 = 0; k0 < K; k0 += TW) {
    if (k0 != 0) __syncthreads();
    load_lv(k0);
    __syncthreads();
    long long qs[KB];
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) qs[kk] = 0;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
      const long long row = (long long)item * BT + tid;
      if (row >= S.n) continue;
      uint32_t id_sel;
      if (cmd->sel_slot == -2) {
        id_sel = tl_old[row];
      } else {
        id_sel = sel_lid ? (uint32_t)sel_lid[row] : 0u;
        if (k0 == 0) tl_old[row] = (uint8_t)id_sel;
      }
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) {
        const int k = k0 + kk;
        if (k >= K) continue;
        const size_t ri = (size_t)k * n_pad + row;
        const double nv = lin_pred(s_lv[0][id_sel][kk], 0, id_sel, k, row);
        const double st = st_in[ri] + nv;
        if (cmd->tune) {
          const double mean0 = S.rs_mean[ri], m20 = S.rs_m2[ri];
          const double delta = nv - mean0;
          const double mean = mean0 + delta / cntf;
          const double delta2 = nv - mean;
          const double m2 = m20 + delta * delta2;
          S.rs_mean[ri] = mean;
          S.rs_m2[ri] = m2;
          qs[kk] += pgb_quant(PGB_SQRT(m2 / cntf), c1, &sat);
        }
        st_out[ri] = st;
      }
    }
    if (cmd->tune) {
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) {
        const int k = k0 + kk;
        if (k >= K) continue;
        long long v1[1] = {qs[kk]};
        block_sum<1>(v1, s_red);
        if (tid == 0 && v1[0]) {
          if (k == 0) {
            InitAcc* a = &S.initacc[(size_t)par * IA_SLOTS + (blockIdx.x % IA_SLOTS)];
            atomicAdd((unsigned long long*)&a->QSTD, (unsigned long long)v1[0]);
          } else {
            long long* ix = S.iax + ((size_t)par * IA_SLOTS + (blockIdx.x % IA_SLOTS)) * 2 * KX;
            atomicAdd((unsigned long long*)&ix[KX + k - 1], (unsigned long long)v1[0]);
          }
        }
      }
    }
  }
  if (sat) atomicAdd(&S.counters[4], (unsigned long long)sat);
  if (pstamp) pstamp[1] = wall_clock64();
}

