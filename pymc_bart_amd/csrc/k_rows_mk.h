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
// F32: the split column from the float32 shadow of the design matrix (see k_rows<..., F32>); continuous /
// one-hot rules only.
// (the compile-time-K instances are held to 4 workgroups per CU, <= 128 VGPRs: K = 4 sits at that edge)
template <int KT, bool LIN, bool F32 = false>
__global__ __launch_bounds__(BT, (KT >= 2 && !LIN) ? 4 : 2) void k_rows_mk(const Dev* __restrict__ Sp, int par) {
  const DevG& S = *reinterpret_cast<const DevG*>(Sp);
  const int K = KT > 0 ? KT : S.K, KX = K - 1;
  constexpr int TW = KT > 0 ? KT : 4;  // outputs per tile (= K for the compile-time instances: one tile)
  constexpr int KB = TW;
  constexpr int NVT = 1 + 2 * TW;      // per particle and tile: counts, aL[TW], aN[TW]
  __shared__ long long s_red[MAXP * NVT * 4];
  __shared__ double s_lv[2][256][KB];
  __shared__ RJob s_job[MAXP];
  __shared__ int s_n[2];
  const Cmd* cmd = &S.cmd[par];
  const int kind = cmd->kind;
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
    const Job* jobs = S.jobs + (size_t)par * MAXP;
    if (tid < 64) {
      int nlist = 0;  // (lanes' particles tid, tid + 64, ...: one block of 64 after the other)
#pragma unroll
      for (int hq = 0; hq < MAXP / 64; ++hq) {
      const int q = tid + 64 * hq;
      Job j;
      j.active = 0;
      j.copy = 0;
      if (q >= 1 && q < S.P) j = jobs[q];
      const bool has = (j.active | j.copy) != 0;
      const unsigned long long m = __ballot(has);
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
        rj.src = j.src_slot < 0 ? -1ll : (long long)(((size_t)j.src_gen * MAXP + j.src_slot) * S.n_pad);
        rj.xoff = (long long)((size_t)j.var * S.n_pad);
        s_job[k] = rj;
      }
      nlist += __popcll(m);
      }
      if (tid == 0) s_n[0] = nlist;
    }
    __syncthreads();
    const int nact = s_n[0];
    if (nact == 0 && !do_init) return;
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
        const int chunk = item % S.nchunks, grp = item / S.nchunks;
        const long long base = (long long)chunk * CH + tid * RPT;
        double stv[RPT][KB];  // sum_trees of this thread's rows, per output of the tile
        if (do_init) {
          const bool writer = grp == 0;
          uint32_t ids_next = *(const uint32_t*)(tl_new + base);
          uint32_t ids_sel = 0;
          if (do_final) {
            if (cmd->sel_slot == -2) {
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
            if (cmd->tree_new == cmd->tree_old) ids_next = ids_sel;
          }
          for (int e = 0; e < RPT; ++e) {
            const long long row = base + e;
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) stv[e][kk] = 0.0;
            if (row >= n) continue;
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
              const int k = k0 + kk;
              if (k >= K) continue;
              double st = st_in[(size_t)k * n_pad + row];
              if (do_final) {
                const double nv = lin_pred(s_lv[0][(ids_sel >> (8 * e)) & 255u][kk], 0, (ids_sel >> (8 * e)) & 255u, k, row);
                st = st + nv;
                if (cmd->tune && writer) {  // [U] RunningSd.update (Welford), per output
                  const size_t ri = (size_t)k * n_pad + row;
                  const double mean0 = S.rs_mean[ri], m20 = S.rs_m2[ri];
                  const double delta = nv - mean0;
                  const double mean = mean0 + delta / cntf;
                  const double delta2 = nv - mean;
                  const double m2 = m20 + delta * delta2;
                  S.rs_mean[ri] = mean;
                  S.rs_m2[ri] = m2;
                  ivQ[kk] += pgb_quant(PGB_SQRT(m2 / cntf), c1, &sat);
                }
              }
              const double o = lin_pred(s_lv[1][(ids_next >> (8 * e)) & 255u][kk], 1, (ids_next >> (8 * e)) & 255u, k, row);
              const double noi = st - o;
              stv[e][kk] = st;
              if (writer) {
                if (k == 0) S.pack[row] = make_double2(st, 0.0);
                else S.packx[(size_t)(k - 1) * n_pad + row] = st;
                st_out[(size_t)k * n_pad + row] = noi;
                ivA[kk] += pgb_quant(st, c1, &sat);
              }
            }
            // (C, the log-likelihood of a fresh stump, and E0, of the current tree, are summed by k_loglik,
            //  which runs after this pass and is compiled per number of outputs)
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
        uint32_t root_ids = 0;
        for (int e = 0; e < RPT; ++e)
          if (base + e >= n) root_ids |= (uint32_t)PGB_ORPHAN << (8 * e);
        long long qst[RPT][KB];  // quantised once per row, reused by every particle of the group
#pragma unroll
        for (int e = 0; e < RPT; ++e)
#pragma unroll
          for (int kk = 0; kk < KB; ++kk) qst[e][kk] = k0 + kk < K ? pgb_quant(stv[e][kk], c1, nullptr) : 0;
        const int g0 = grp * G, g1 = (g0 + G < nact) ? g0 + G : nact;
        for (int g = g0; g < g1; ++g) {
          const RJob& rj = s_job[g];
          // (a later tile finds the rows of the leaf in the labels as they were BEFORE this pass -- the source
          //  generation is never the one being written -- and re-derives the sides from the split column)
          const uint32_t ids = rj.src < 0 ? root_ids : *(const uint32_t*)(S.lid + rj.src + base);
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
          float xf[RPT] = {0.f, 0.f, 0.f, 0.f};
          if constexpr (F32) {
            const float4 tf = *(const float4*)(S.XT32 + rj.xoff + base);
            xf[0] = tf.x; xf[1] = tf.y; xf[2] = tf.z; xf[3] = tf.w;
          } else {
            const double2 t0 = xp[0], t1 = xp[1];
            x[0] = t0.x; x[1] = t0.y; x[2] = t1.x; x[3] = t1.y;
          }
          const float r_vf = (float)r_v;
          int side[RPT];  // 0: not in the leaf, 1: left, 2: right, 3: dropped (missing value)
          long long cnts = 0;
          for (int e = 0; e < RPT; ++e) {
            side[e] = 0;
            if (((ids >> (8 * e)) & 255u) == r_label) {
              bool missing, left;
              if constexpr (F32) {  // decided on the float32 values unless they tie
                missing = xf[e] != xf[e];
                if (xf[e] != r_vf) left = r_rule == PGB_RULE_CONTINUOUS ? xf[e] < r_vf : false;
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
            S.cc[(size_t)rj.ccL * S.nchunks + chunk] = (uint16_t)cL;
            S.cc[(size_t)rj.ccR * S.nchunks + chunk] = (uint16_t)cR;
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
}

