// k_ctrl.h -- part of pgbart_hip.hip (not a standalone header): k_ctrl: the control kernel (finish a round, weights, resampling, next proposal).
// ------------------------------------------------------------------ k_ctrl
struct Fin {  // result of finishing the pending split of an old particle (kept in LDS)
  int ok;     // 1: children created, 0: no pending split, -1: failed one-hot split
  int cL, cR;
  int nn_old, n_nodes, n_leaves, next_pop;
  int loc_gen, loc_slot;
  int node, var, new_label, ccL, ccR;
  uint8_t depth, label;
  long long aL, aR, bL, bR, c2L, c2R;
  long long llL, llR, ll_tot, ll_orph;  // Bernoulli families
  double split, vL, vR, sseL, sseR, sse_tot, sse_orph;
  // linear response: the children's linear parts (svar < 0: constant leaf)
  double slopeL, xbarL, slopeR, xbarR;
  int svarL, svarR;
};

// [U] normalize + inverse-CDF pick on ONE wave, one particle per lane: lanes [first, first+cnt)
// hold log-weights.  Cumulative weights are the fixed-order wave scan the numeric contract
// defines (pgb_scan64 / pgb_weights_scan / pgb_pick in include/pgbart_spec.h).
// NH = PGB_MAX_PARTICLES / 64 particles per lane: lane l holds the log-weights of particles l, l + 64, ...; the
// cumulative weights are one DPP scan per block of 64, each later block shifted by the total of the block before.
template <int NH>
__device__ __forceinline__ int wave_pick(const double (&lw)[NH], int first, int cnt, double u) {
  const int lane = threadIdx.x & 63;
  bool act[NH];
  double m = -1.0e308;
#pragma unroll
  for (int hq = 0; hq < NH; ++hq) {
    const int q = lane + 64 * hq;
    act[hq] = q >= first && q < first + cnt;
    if (act[hq] && lw[hq] > m) m = lw[hq];  // (a maximum: any order)
  }
  const double mx = wave_max_d(m);
  double W[NH];
#define PGB_SCAN_STEP(ctrl, rm)                                                       \
  {                                                                                   \
    const int tl = __builtin_amdgcn_update_dpp(0, __double2loint(Wh), ctrl, rm, 0xf, 0); \
    const int th = __builtin_amdgcn_update_dpp(0, __double2hiint(Wh), ctrl, rm, 0xf, 0); \
    Wh = Wh + __hiloint2double(th, tl);                                               \
  }
  double carry = 0.0;
#pragma unroll
  for (int hq = 0; hq < NH; ++hq) {
    double Wh = act[hq] ? pgb_exp(lw[hq] - mx) + 1e-12 : 0.0;
    PGB_SCAN_STEP(0x111, 0xf)  // row_shr:1
    PGB_SCAN_STEP(0x112, 0xf)  // row_shr:2
    PGB_SCAN_STEP(0x114, 0xf)  // row_shr:4
    PGB_SCAN_STEP(0x118, 0xf)  // row_shr:8
    PGB_SCAN_STEP(0x142, 0xa)  // row_bcast:15 -> rows 1, 3
    PGB_SCAN_STEP(0x143, 0xc)  // row_bcast:31 -> rows 2, 3
    if (hq > 0) Wh = Wh + carry;  // (pgb_weights_scan: a later block is shifted by the total before it)
    if (hq + 1 < NH) carry = readlane_d(Wh, 63);
    W[hq] = Wh;
  }
#undef PGB_SCAN_STEP
  const int last = first + cnt - 1;
  double tot = 0.0;
#pragma unroll
  for (int hq = 0; hq < NH; ++hq)
    if ((last >> 6) == hq) tot = readlane_d(W[hq], last & 63);  // (wave-uniform)
  const double thr = u * tot;
#pragma unroll
  for (int hq = 0; hq < NH; ++hq) {
    const int q = lane + 64 * hq;
    const bool hit = act[hq] && (q < last) && !(thr > W[hq]);
    const unsigned long long mk = __ballot(hit);
    if (mk) return 64 * hq + (int)__ffsll((long long)mk) - 1;
  }
  return last;
}

// [U] SampleSplittingVariable.rvs on one wave, from stored prefix sums (pgb_sample_var)
__device__ __forceinline__ int sample_var_prefix(const long long* Sarr, int p, double u) {
  const int lane = threadIdx.x & 63;
  if (p > 64 && p <= 256) {
    // 65 .. 256 columns (cfg4: 100, cfg5: 200): the blocks of 64 prefix sums are requested TOGETHER with the total;
    // the loop below pays a memory round trip per block, one after the other, before the wave can go on to what
    // the slot is waiting for (at cfg5 the pre-draw waves reached the barrier 1.4 us after wave 0)
    long long v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = Sarr[lane + 64 * i < p ? lane + 64 * i : p - 1];
    const double thr = u * (double)Sarr[p - 1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = lane + 64 * i;
      const unsigned long long m = __ballot(j < p && thr <= (double)v[i]);
      if (m) return 64 * i + (int)__ffsll((long long)m) - 1;
    }
    return p - 1;
  }
  const double thr = u * (double)Sarr[p - 1];
  for (int base = 0; base < p; base += 64) {
    const int j = base + lane;
    const bool hit = j < p && thr <= (double)Sarr[j];
    const unsigned long long m = __ballot(hit);
    if (m) return base + (int)__ffsll((long long)m) - 1;
  }
  return p - 1;
}

// The same draw from the integer split weights themselves (sampler being rebuilt by this slot):
// exact prefix sums, 64 variables per step with a running carry.
__device__ __forceinline__ int sample_var_weights(const long long* A, int p, double u) {
  const int lane = threadIdx.x & 63;
  long long part = 0;
  for (int j = lane; j < p; j += 64) part += A[j];
  part = wave_sum_dpp(part);
  const long long tot = ((long long)__builtin_amdgcn_readlane((int)(part >> 32), 63) << 32) |
                        (unsigned)__builtin_amdgcn_readlane((int)part, 63);
  const double thr = u * (double)tot;
  long long carry = 0;
  for (int base = 0; base < p; base += 64) {
    const int j = base + lane;
    long long run = wave_sum_dpp(j < p ? A[j] : 0) + carry;  // inclusive prefix through variable j
    const bool hit = j < p && thr <= (double)run;
    const unsigned long long m = __ballot(hit);
    if (m) return base + (int)__ffsll((long long)m) - 1;
    carry = ((long long)__builtin_amdgcn_readlane((int)(run >> 32), 63) << 32) |
            (unsigned)__builtin_amdgcn_readlane((int)run, 63);
  }
  return p - 1;
}

// linear response: the two children's linear parts from the sums the row pass left (shared by
// k_ctrl and, for the per-row families, k_loglik: the very same arithmetic in both places)
struct LinKids {
  double slopeL, xbarL, slopeR, xbarR;
  int svarL, svarR;
  pgb_linfit fL, fR;
  long long urL, urR;  // sum q(u r) of the children (the Normal family's SSE needs them)
  // what the slopes of further outputs need (K-vector leaves, lin_children_x)
  long long u0L, u1L, u0R, u1R;
  bool linL, linR;
  double uscale, xs;
};
__device__ __forceinline__ LinKids lin_children(const DevG& S, const AccU* __restrict__ copies, int var, int cL, int cR,
                                                long long aL, long long aR, uint32_t it, uint32_t round, uint32_t q) {
  LinKids k;
  k.slopeL = k.xbarL = k.slopeR = k.xbarR = 0.0;
  k.svarL = k.svarR = -1;
  k.fL = pgb_linfit{0.0, 0.0, 0.0};
  k.fR = pgb_linfit{0.0, 0.0, 0.0};
  long long ul[4] = {0, 0, 0, 0}, ur[4] = {0, 0, 0, 0};
  for (int c = 0; c < ACC_SLOTS; ++c) {
    const AccU t = copies[c * ACC_STRIDE];
    for (int i2 = 0; i2 < 4; ++i2) { ul[i2] += t.uL[i2]; ur[i2] += t.uR[i2]; }
  }
  k.urL = ul[3];
  k.urR = ur[3];
  k.u0L = ul[0]; k.u1L = ul[1]; k.u0R = ur[0]; k.u1R = ur[1];
  bool linL = true, linR = true;
  if (S.response == PGB_RESPONSE_MIX) {  // [U] "mix": a fair coin per child
    const pgb_u2 um = pgb_draw2(S.seed, it, round, q, PGB_RNG_MIX, 0);
    linL = um.u0 < 0.5;
    linR = um.u1 < 0.5;
  }
  const int ex = S.col_ex[var];
  const double uscale = pgb_pow2(-ex), xs = pgb_pow2(ex);
  k.linL = linL; k.linR = linR;
  k.uscale = uscale; k.xs = xs;
  if (linL) {
    k.fL = pgb_lin_fit(cL, ul[0], ul[1], ul[2], aL, S.sc.inv_c1, S.inv_R, S.mdouble);
    if (k.fL.slope_u != 0.0) {
      k.svarL = var;
      k.slopeL = k.fL.slope_u * uscale;
      k.xbarL = k.fL.ubar * xs;
    }
  }
  if (linR) {
    k.fR = pgb_lin_fit(cR, ur[0], ur[1], ur[2], aR, S.sc.inv_c1, S.inv_R, S.mdouble);
    if (k.fR.slope_u != 0.0) {
      k.svarR = var;
      k.slopeR = k.fR.slope_u * uscale;
      k.xbarR = k.fR.ubar * xs;
    }
  }
  return k;
}
// K-vector leaves: the slopes of extension output kx of both children (sums of u st_k from accux,
// the sums of u / u^2 are shared with output 0); a leaf is linear when ANY output has a slope.
__device__ __forceinline__ void lin_children_x(const DevG& S, LinKids& lk, ChildX& cx, int var, int cL, int cR,
                                               long long usL, long long usR) {
  if (lk.linL) {
    const pgb_linfit f = pgb_lin_fit(cL, lk.u0L, lk.u1L, usL, cx.aL, S.sc.inv_c1, S.inv_R, S.mdouble);
    cx.sL = f.slope_u * lk.uscale;
    if (f.slope_u != 0.0 && lk.svarL < 0) {
      lk.svarL = var;
      lk.slopeL = lk.fL.slope_u * lk.uscale;
      lk.xbarL = lk.fL.ubar * lk.xs;
    }
  }
  if (lk.linR) {
    const pgb_linfit f = pgb_lin_fit(cR, lk.u0R, lk.u1R, usR, cx.aR, S.sc.inv_c1, S.inv_R, S.mdouble);
    cx.sR = f.slope_u * lk.uscale;
    if (f.slope_u != 0.0 && lk.svarR < 0) {
      lk.svarR = var;
      lk.slopeR = lk.fR.slope_u * lk.uscale;
      lk.xbarR = lk.fR.ubar * lk.xs;
    }
  }
}


// [U] get_split_value: the k-th row (ascending) of a leaf, k = floor(u * cnt), on ONE wave (all 64 lanes
// call this with wave-uniform arguments; no workgroup barrier): the per-chunk row counts the node got
// when it was created are scanned to find the chunk, then the chunk's 1024 label bytes (16 per lane).
// A missing value at the chosen row redraws the row (<= PGB_SELECT_TRIES).  pre0 / pre1: the
// pre-drawn uniforms of the tries ([1 + try]).  Returns found and the split value in every lane.
struct SplitRow {
  int found;
  int vkey;  // order key of the value (XK16), 0 without keys
  double v;
};
// V16: compile the sixteen-counts-in-registers path (below) in -- the KEYS instances of k_ctrl, i.e. the data sets
// beyond the Infinity Cache, which are the ones with hundreds of chunks; the instance the small data sets launch
// (cfg2: a latency kernel at its register edge) carries none of it.
template <bool V16 = false>
__device__ __forceinline__ SplitRow select_split_row(const DevG& S, int j, bool subset_rule, int src_gen, int src_slot,
                                                     int ncnt, int ncc, int nlabel, const double* pre0,
                                                     const double* pre1, const uint16_t* xk16, long long* tr_rec = nullptr) {
  const double* xc = S.XT + (size_t)j * S.n_pad;
  const uint8_t* lid = src_slot >= 0 ? S.lid + ((size_t)src_gen * MAXP + src_slot) * S.n_pad : nullptr;
  const uint16_t* ccr = ncc >= 0 ? S.cc + (size_t)ncc * (V16 ? S.cc_stride : S.nchunks) : nullptr;
  const uint16_t* kc = xk16 ? xk16 + (size_t)j * S.n_pad : nullptr;  // (xk16 = S.XK16 as a preloaded kernel argument)
  SplitRow out;
  out.found = 0;
  out.vkey = 0;
  out.v = 0.0;
  // per-lane partial sums of the node's per-chunk row counts (independent of the retry); the counts of a
  // lane's chunks stay in registers (<= 4 chunks per lane, i.e. n <= 262144: one 8-byte load, no re-read
  // when the k-th row is located; longer columns walk the counts in memory)
  // 257 .. 1024 chunks (n <= 1 048 576 rows: cfg4): a lane owns SIXTEEN chunks whatever n is, and their counts -- 32
  // contiguous bytes of the row, 16-byte aligned because rows are padded to 8 counts (Dev::cc_stride) -- arrive in TWO
  // 16-byte loads and stay in registers.  (The general path below walked them with sixteen 2-byte loads, and then
  // found the chunk of the k-th row with a SECOND round trip -- the owning lane's counts, one per lane, and a scan:
  // 1.4 + 0.4 us of k_ctrl's 7.7 at cfg4 against 0.6 us for the same stage at cfg2, in-kernel stamps.)
#ifndef PGB_SEL_V16
#define PGB_SEL_V16 1 /* experiment knob: 0 = the general path for every size (A/B) */
#endif
  const bool v16 = V16 && PGB_SEL_V16 != 0 && S.nchunks > 256 && S.nchunks <= 1024;  // (wave-uniform)
  const int per = v16 ? 16 : (S.nchunks + 63) / 64;
  const int c0 = lane_id() * per;
  int c1 = c0 + per;
  if (c1 > S.nchunks) c1 = S.nchunks;
  int part = 0, pre = 0;
  unsigned cw[4] = {0u, 0u, 0u, 0u};  // counts of chunks c0 .. c0 + 3
  uint32_t cp[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};  // v16: counts of chunks c0 .. c0 + 15, two per word
  const bool inreg = per <= 4;
  if (lid != nullptr) {
    if (inreg) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < per && c0 + i < S.nchunks) cw[i] = ccr[c0 + i];
      part = (int)(cw[0] + cw[1] + cw[2] + cw[3]);
    } else if (v16) {
      // (counts beyond nchunks inside the padded row are 0: pgb_create zeroes the table and the row passes write
      //  chunks < nchunks only; beyond the padded row nothing is read)
      if (c0 + 8 <= S.cc_stride) {
        const uint4 a = *(const uint4*)(ccr + c0);
        cp[0] = a.x; cp[1] = a.y; cp[2] = a.z; cp[3] = a.w;
      }
      if (c0 + 16 <= S.cc_stride) {
        const uint4 b = *(const uint4*)(ccr + c0 + 8);
        cp[4] = b.x; cp[5] = b.y; cp[6] = b.z; cp[7] = b.w;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) part += (int)((cp[i] & 0xFFFFu) + (cp[i] >> 16));
    } else {
      for (int cc = c0; cc < c1; ++cc) part += ccr[cc];
    }
    pre = wave_incl_scan(part) - part;
  }
  TRS(20);
  for (uint32_t tr = 0; tr < PGB_SELECT_TRIES && !out.found; ++tr) {
    long long k = (long long)(pre0[1 + tr] * (double)ncnt);
    if (k > ncnt - 1) k = ncnt - 1;
    long long row;
    if (lid == nullptr) {
      row = k;  // untouched root: every row belongs to it
    } else {
      // (1) which chunk holds the k-th row
      const bool own = (long long)pre <= k && k < (long long)pre + part;
      int cstar = 0, kk = 0;
      if (own) {
        kk = (int)(k - pre);
        cstar = c0;
        if (inreg) {
#pragma unroll
          for (int i = 0; i < 3; ++i)
            if (kk >= (int)cw[i] && cstar == c0 + i) {
              kk -= (int)cw[i];
              ++cstar;
            }
        } else if (v16) {  // the same walk over the lane's sixteen counts in registers
#pragma unroll
          for (int i = 0; i < 15; ++i) {
            const int ci = (int)((i & 1) ? (cp[i >> 1] >> 16) : (cp[i >> 1] & 0xFFFFu));
            if (kk >= ci && cstar == c0 + i) {
              kk -= ci;
              ++cstar;
            }
          }
        } else if (per > 64) {  // (more than 64 chunks per lane, n > 4M rows: walk the counts)
          while (kk >= ccr[cstar]) {
            kk -= ccr[cstar];
            ++cstar;
          }
        }
      }
      const int ol = (int)__ffsll((long long)__ballot(own)) - 1;
      cstar = __builtin_amdgcn_readlane(cstar, ol);
      kk = __builtin_amdgcn_readlane(kk, ol);
      if (!inreg && !v16 && per <= 64) {
        // the owning lane's chunks, one per lane: a second scan instead of a serial walk of dependent loads
        // (at n = 1M a lane owns 16 chunks)
        const int cl = cstar + lane_id();
        const int cv = (lane_id() < per && cl < S.nchunks) ? (int)ccr[cl] : 0;
        const int inc = wave_incl_scan(cv);
        const bool hit = lane_id() < per && kk < inc && kk >= inc - cv;
        const int hl = (int)__ffsll((long long)__ballot(hit)) - 1;
        kk -= __builtin_amdgcn_readlane(inc - cv, hl);
        cstar += hl;
      }
      TRS(23);
      // (2) which row inside the chunk: 16 label bytes per lane.  Matches as a 16-bit mask: a byte of
      // (word xor label-in-every-byte) is zero exactly where the label matches; the exact zero-byte test
      // ~(((x & 0x7f7f7f7f) + 0x7f7f7f7f) | x | 0x7f7f7f7f) leaves 0x80 in those bytes and nothing else.
      const uint4 ids = *(const uint4*)(lid + (size_t)cstar * CH + lane_id() * 16);
      const uint32_t wds[4] = {ids.x, ids.y, ids.z, ids.w};
      const uint32_t lab4 = (uint32_t)nlabel * 0x01010101u;
      uint32_t mask = 0;
#pragma unroll
      for (int wd = 0; wd < 4; ++wd) {
        const uint32_t x = wds[wd] ^ lab4;
        const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);  // 0x80 per matching byte
        // gather the four flag bits (bits 7, 15, 23, 31) into a nibble
        const uint32_t nib = ((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u);
        mask |= nib << (4 * wd);
      }
      const int mcnt = __popc(mask);
      const int pre2 = wave_incl_scan(mcnt) - mcnt;
      const bool own2 = pre2 <= kk && kk < pre2 + mcnt;
      int off = 0;
      if (own2) {  // position of the (kk - pre2)-th set bit: clear that many low set bits, count trailing zeros
        uint32_t mm = mask;
        for (int rem = kk - pre2; rem > 0; --rem) mm &= mm - 1u;
        off = __ffs((int)mm) - 1;
      }
      const int ol2 = (int)__ffsll((long long)__ballot(own2)) - 1;
      off = __builtin_amdgcn_readlane(off, ol2);
      row = (long long)cstar * CH + ol2 * 16 + off;
    }
    TRS(21);
    const double x = xc[row];
    if (kc) out.vkey = kc[row];  // (requested with the value: the same round trip)
    out.found = (x == x) ? 1 : 0;
    TRS(22);
    out.v = x;
    if (out.found && subset_rule) out.v = pgb_subset_value(pre1[1 + tr], x);
  }
  return out;
}

// MK: K-vector leaves (K > 1).  The single-output instantiation contains none of that code.
// KEYS: the data set has 16-bit order keys (XK16): the key of a split value is read with the value.  A separate
// instance, so that the one the small data sets launch (cfg2) carries none of it (0.07 us per launch otherwise).
template <bool MK, bool LIN, bool KEYS = false>
__global__ __launch_bounds__(BT) __attribute__((amdgpu_waves_per_eu(1, 1)))  // latency kernel: registers, not occupancy
void k_ctrl(const Dev* __restrict__ Sp, int par, int nwg, Ctrl* __restrict__ ctrls, const InitAcc* __restrict__ ias,
            const Job* __restrict__ jobs_all, const Acc* __restrict__ acc_all, const DPart* __restrict__ parts_all,
            const uint16_t* __restrict__ xk16) {
  // nwg repeats gridDim.x as an explicit argument (it takes the padding after `par`): explicit arguments arrive
  // preloaded in SGPRs, gridDim.x is a HIDDEN one and cost a scalar load with its wait in front of the first batch
  // of global loads of every launch
  // ctrls / ias / jobs_all / acc_all / parts_all repeat S.ctrl / S.initacc / S.jobs / S.acc / S.parts as
  // kernel arguments (preloaded into SGPRs): their first loads do not wait for the argument block S
  const DevG& S = *reinterpret_cast<const DevG*>(Sp);  // device-resident: kernel arguments live in host-coherent memory, HBM is closer
  __shared__ Fin s_fin[MAXP];
  __shared__ int s_i[16];
  __shared__ double s_d[4];
  __shared__ double s_pre[2][PGB_SELECT_TRIES + 2];  // [set][0: coin, 1+t: row draw of try t]
  __shared__ double s_pre1[2][PGB_SELECT_TRIES + 2]; // second uniform of the same draws (subset masks)
  __shared__ ChildX s_finx[MK ? MAXP : 1][KXMAX];     // K-vector leaves: children, outputs 1..K-1
  __shared__ double s_prior[PGB_MAX_DEPTH];           // P(leaf | depth): read once by the idle wave 3
  __shared__ DNode s_pop[MAXP];                       // node each OLD particle would pop next (prefetched)
  __shared__ double s_ahead[2][4];                    // [set][z0, z1, u_res, u_fin]: draws made one slot ahead
  __shared__ double s_aheadx[2][MK ? KXMAX : 1][2];   // ... leaf noise of outputs 1..K-1

  TR_DECL();
  TR0();
  // The previous round's job record and split statistics of old particle q = lane (wave 0) are
  // requested FIRST, together with the control word: their addresses depend on `par` alone.  Harmless
  // in idle / first slots (the records exist).
  constexpr int NH = MAXP / 64;  // particles per lane of wave 0: q = lane, lane + 64, ...
  Job j_pre[NH];
  Acc a_pre[NH];
  // (only the lanes that can be particles: the grid has P - 1 or P workgroups, so particles 1 .. nwg (= gridDim.x)
  //  cover particles 1 .. P - 1 -- at P = 40 the other 23 lanes made 37 % of this batch for nothing)
#pragma unroll
  for (int hq = 0; hq < NH; ++hq) {
    const unsigned qq = threadIdx.x + 64u * hq;
    if (threadIdx.x < 64u && qq >= 1 && qq <= (unsigned)nwg) {
      j_pre[hq] = jobs_all[(size_t)(par ^ 1) * MAXP + qq];
      a_pre[hq] = load_acc(&acc_all[((size_t)(par ^ 1) * MAXP + qq) * ACC_PER]);
    }
  }
  if (threadIdx.x >= BT - 64) s_prior[threadIdx.x - (BT - 64)] = S.prior_leaf[threadIdx.x - (BT - 64)];
  const Ctrl c = load_uniform(&ctrls[par]);
  // the credit the host enqueues against: slots whose control kernel has started (a posted store, no wait)
  if (blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(S.host_flag + 2, (unsigned long long)(c.slot_no + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  TR_BIND(c.slot_no);
  Ctrl* co = &ctrls[par ^ 1];
  const int b = blockIdx.x, p = b + 1, tid = threadIdx.x;
  const int P = S.P, Lc = P - 1;
  Cmd* cmd = &S.cmd[par];
  // One workgroup more than there are particles to propose for (single-output constant leaves): it builds
  // the label -> value tables the row pass of THIS slot needs if the slot turns out to end the tree -- of
  // the tree as it stands (the update may keep it) and of the tree that would be updated next.  Both depend
  // on the control word alone, so they are built here, next to the workgroups that decide; a slot that does
  // not end the tree leaves them unused.  (The other instances build their tables where the tree ends.)
  constexpr bool SPEC_LV = !MK && !LIN;
  if constexpr (SPEC_LV) {
    if (b == P - 1) {
      if (c.phase != PH_ROUND) return;
      const int t_old = c.lower + c.k;
      int t_next;
      if (c.k + 1 < c.batch_n) {
        t_next = t_old + 1;
      } else {
        const int upper = c.lower + c.batch_n;
        t_next = upper < S.m ? upper : 0;
      }
      build_lv(S.trees[t_old].nd, S.trees[t_old].n_nodes, cmd->lv_keep);
      if (t_next != t_old) build_lv(S.trees[t_next].nd, S.trees[t_next].n_nodes, cmd->lv_next);
      return;
    }
  }
  InitAcc ia;  // statistics of the previous FINAL/INIT row pass (integer sums over IA_SLOTS lines)
  {
    const InitAcc* src = ias + (size_t)(par ^ 1) * IA_SLOTS;
    ia = load_uniform(&src[0]);
#pragma unroll
    for (int k = 1; k < IA_SLOTS; ++k) {
      const InitAcc t = load_uniform(&src[k]);
      ia.A += t.A; ia.B += t.B; ia.C += t.C; ia.E0 += t.E0; ia.QSTD += t.QSTD;
    }
  }

  // pending leaf_sd from the FINAL pass of the previous slot ([U] RunningSd -> leaf_sd)
  double leaf_sd = c.leaf_sd;
  // (the "third update on" test is repeated out here on purpose: with pend_iter tested only inside the call the
  //  compiler re-ordered the first loads of this kernel and every launch took 0.3 us longer -- 6.45 -> 6.75 us,
  //  A/B of five statement shapes on one box, profiles/r03_experiments.md section 14)
  if (c.pend_leafsd && c.pend_iter > 2) leaf_sd = pgb_tuned_leaf_sd(c.leaf_sd, c.pend_iter, ia.QSTD, S.sc.inv_c1, S.n);

  if (b == 0 && tid == 0 && c.phase != PH_IDLE) atomicAdd(&S.counters[5], 1ull);  // slots that did work
  if (tid < ACC_SLOTS) {
    Acc z;
    memset(&z, 0, sizeof z);
    S.acc[((size_t)par * MAXP + p) * ACC_PER + tid * ACC_STRIDE] = z;
    if constexpr (LIN) {
      AccU zu;
      memset(&zu, 0, sizeof zu);
      S.accu[((size_t)par * MAXP + p) * ACC_PER + tid * ACC_STRIDE] = zu;
    }
  }
  if (tid < LL_SLOTS && S.family != PGB_FAMILY_NORMAL)
    S.accl[((size_t)par * MAXP + p) * LL_PER + tid * LL_STRIDE] = AccL{0, 0, 0, 0};
  if (b == 0 && tid < IA_SLOTS) S.initacc[(size_t)par * IA_SLOTS + tid] = InitAcc{0, 0, 0, 0, 0, 0, 0, 0};
  const int KX = MK ? S.K - 1 : 0;
  if constexpr (MK) {
    if (tid < AX_PER) S.accx[((size_t)par * MAXP + p) * AX_PER + tid] = 0;
    if constexpr (LIN)
      if (tid < AX_PER) S.accux[((size_t)par * MAXP + p) * AX_PER + tid] = 0;
    if (b == 0)
      for (int i = tid; i < IA_SLOTS * 2 * KX; i += BT) S.iax[(size_t)par * IA_SLOTS * 2 * KX + i] = 0;
  }

  if (c.phase == PH_IDLE) {
    if (b == 0 && tid == 0) {
      Ctrl o = c;
      o.slot_no = c.slot_no + 1;
      o.leaf_sd = leaf_sd;
      if constexpr (MK)
        for (int k = 0; k < KX; ++k) S.lsdx[(par ^ 1) * KXMAX + k] = leaf_sd_x(S, c, par, par ^ 1, k);
      o.pend_leafsd = 0;
      o.done_pub = c.steps_done;
      *co = o;
      cmd->kind = CMD_NOOP;
      if (c.done_pub != c.steps_done)  // first idle slot after a step: every row pass of the step has run
        __hip_atomic_store(S.host_flag + 1, (unsigned long long)c.steps_done, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);  // (release: the control word above is visible first)
    }
    return;
  }

  const bool begin = c.phase == PH_BEGIN;  // first tree of a step: nothing to finish
  const bool normal = S.family == PGB_FAMILY_NORMAL;
  TR(1);
  const int r = c.round;  // >= 1 in PH_ROUND: round 0 is proposed by the slot that starts the tree
  const uint32_t it = (uint32_t)c.iter;
  const DPart* OT = parts_all + (size_t)par * MAXP;
  DPart* NT = S.parts + (size_t)(par ^ 1) * MAXP;
  const Job* JP = S.jobs + (size_t)(par ^ 1) * MAXP;  // jobs (+ particle headers) of the previous slot
  Job* JN = S.jobs + (size_t)par * MAXP;
  DPart* me = &NT[p];
  const long long* cdfS = S.cdfS + (size_t)c.cdf_cur * S.p;
  const long long* alpha = S.alpha + (size_t)c.alpha_cur * S.p;
  // the sampler of the NEXT tree is rebuilt from the weights when this tree ends while tuning
  const bool rebuild = !begin && c.tune && c.iter > S.m;

  // Waves 1 and 2 make the draws of the two proposals this slot may need while wave 0 finishes the
  // previous round; they depend only on (iter, round, particle).
  //   set 0: round r of the current tree            (iter,     r, p)
  //   set 1: round 0 of the next tree to be updated (iter + 1, 0, p)
#ifndef PGB_CTRL_FINX
#define PGB_CTRL_FINX 1 /* 0 = waves 1..3 derive the extension outputs of K-vector leaves again (the round-4 form; A/B knob) */
#endif
  if (tid >= 64 && tid < 192) {
    const int set = (tid >> 6) - 1, l = tid & 63;
    const pgb_u2 u = pgb_draw2(S.seed, set ? it + 1u : it, set ? 0u : (uint32_t)r, (uint32_t)p,
                               l == 0 ? PGB_RNG_PROPOSE : PGB_RNG_SELECT, l == 0 ? 0u : (uint32_t)(l - 1));
    TRX(32 + set, blockIdx.x == 1 && l == 0);
    if (l <= PGB_SELECT_TRIES) {
      s_pre[set][l] = u.u0;
      s_pre1[set][l] = u.u1;
    }
    const double u1 = readlane_d(u.u1, 0);
    const int jj = (set && rebuild) ? sample_var_weights(alpha, S.p, u1) : sample_var_prefix(cdfS, S.p, u1);
    TRX(34 + set, blockIdx.x == 1 && l == 0);
    if (l == 0) {
      s_i[8 + set] = jj;
      // the column's rule and NaN flag: fetched here, off the critical path (they used to be two dependent
      // global loads right before the job record is written)
      s_i[10 + set] = S.rules[jj];
      s_i[12 + set] = S.col_nan[jj];
    }
    TRX(9 + set, blockIdx.x == 1 && l == 0);
  }

  // Wave 3: the draws the NEXT slot starts with -- the leaf noise of the children this slot's proposal may
  // create (particle p), the resampling offset of the proposed round and the final-choice draw of its tree
  // -- for both candidate proposals (set 0 / 1 as above).  They travel in the job record / control word.
  if (tid >= 192 && tid < 198) {
    const int l = tid - 192, set = l & 1, kind = l >> 1;  // kind 0: leaf noise, 1: resampling, 2: final choice
    const uint32_t ita = set ? it + 1u : it, ra = set ? 0u : (uint32_t)r;
    if (kind == 0) {
      const pgb_u2 ul = pgb_draw2(S.seed, ita, ra, (uint32_t)p, PGB_RNG_LEAF, 0);
      double z0, z1;
      pgb_normal2(ul.u0, ul.u1, &z0, &z1);
      s_ahead[set][0] = z0;
      s_ahead[set][1] = z1;
    } else if (kind == 1) {
      s_ahead[set][2] = pgb_draw2(S.seed, ita, ra, 0u, PGB_RNG_RESAMPLE, 0).u0;
    } else {
      s_ahead[set][3] = pgb_draw2(S.seed, ita, 0, 0, PGB_RNG_FINAL, 0).u0;
    }
  }
  if constexpr (MK) {  // ... and the leaf noise of outputs 1..K-1 (sub = output), same two candidate proposals
    if (tid >= 198 && tid < 198 + 2 * KX) {
      const int l = tid - 198, set = l & 1, k = l >> 1;
      const pgb_u2 ul = pgb_draw2(S.seed, set ? it + 1u : it, set ? 0u : (uint32_t)r, (uint32_t)p, PGB_RNG_LEAF, (uint32_t)(k + 1));
      double z0, z1;
      pgb_normal2(ul.u0, ul.u1, &z0, &z1);
      s_aheadx[set][k][0] = z0;
      s_aheadx[set][k][1] = z1;
    }
  }

  int anc = p;  // ancestor (old particle index) of new particle p
  int pick0 = p;  // ... as wave 0 itself picked it (scalar; the other waves read it from LDS)
  bool stop = false;
  int sel = 0;
  double sse0 = c.sse0;

  if (!begin) {
    // [U] init_particles: the root statistics of this tree arrive with the INIT pass that ran
    // together with round 0; they are patched in here (round 1)
    const bool r1 = r == 1;
    const double root_sse = pgb_leaf_sse(S.n, ia.B, ia.C, S.init_leaf, S.sc.inv_c1, S.sc.inv_c2);
    // weight of the reference particle p0: its SSE (Normal) or its log-likelihood (Bernoulli)
    if (r1) sse0 = (double)ia.E0 * (normal ? S.sc.inv_c2 : S.sc.inv_cl);
    // -------- wave 0: finish round r-1 for every old particle (lane q <-> old particle q),
    //          then decide stop / ancestor / final choice
    if (tid < 64) {
      double lwv[NH];     // log-weights of this lane's particles
      DNode popv[NH];     // the node each of them pops next (stored after the pick, see below)
      bool ispv[NH];
      bool pend_any = false;
      int n_popped = 0, n_active = 0;
      long long rt_lane = 0;
      const double u_res = c.u_res;
#pragma unroll
      for (int hq = 0; hq < NH; ++hq) {
      const int q = tid + 64 * hq;
      const bool isp = q >= 1 && q < P;
      // this lane's record is built directly in LDS (a register copy with a final struct store
      // defeats scalar replacement and ends up in scratch); slot 0 is unused in this phase
      Fin& f = s_fin[isp ? q : 0];
      double lw = 0.0;
      bool pending = false;
      // RNG + Box-Muller do not depend on memory: they run while the loads below are in flight
      Job j;
      Acc a;
      JobL jl = {0, 0, 0, 0};
      AccL al = {0, 0, 0, 0};
      DNode popn;  // the node this particle pops next if it is an old node (children: from Fin)
      memset(&popn, 0, sizeof popn);
      LinKids lk;  // linear response: kept for the extension outputs below
      lk.svarL = lk.svarR = -1;
      lk.linL = lk.linR = false;
      if (isp) {
        j = j_pre[hq];
        a = a_pre[hq];
        // requested as soon as the job header is here; consumed at the end of this phase
        if (j.h_next_pop < j.h_n_nodes) popn = OT[q].nd[j.h_next_pop];
        if (!normal) {
          jl = S.jobl[(par ^ 1) * MAXP + q];
          al = AccL{0, 0, 0, 0};
          for (int k = 0; k < LL_SLOTS; ++k) {
            const AccL t = S.accl[((size_t)(par ^ 1) * MAXP + q) * LL_PER + k * LL_STRIDE];
            al.llL += t.llL; al.llR += t.llR; al.llN += t.llN;
          }
        }
      }
      // the leaf noise of old particle q's pending split and the resampling offset were drawn one slot
      // ahead (same addresses: (iter, r - 1, q, LEAF) and (iter, r - 1, 0, RESAMPLE)) and arrive with the
      // job record / control word
      const double z0 = isp ? j.z0 : 0.0, z1 = isp ? j.z1 : 0.0;
      if (isp) {
        if (r1) {  // round-0 jobs were written before the root statistics existed
          j.p_q_st = ia.A;
          j.p_q_r = ia.B;
          j.p_q_r2 = ia.C;
          j.p_sse = root_sse;
          j.h_sse_tot = root_sse;
          j.h_sse_orph = 0.0;
          jl.p_ll = ia.C;  // non-Normal families: C carries the stump's log-likelihood
          jl.h_ll_tot = ia.C;
          jl.h_ll_orph = 0;
        }
        f.ok = 0;
        f.nn_old = j.h_n_nodes;
        f.n_nodes = j.h_n_nodes;
        f.n_leaves = j.h_n_leaves;
        f.next_pop = j.h_next_pop;
        f.sse_tot = j.h_sse_tot;
        f.sse_orph = j.h_sse_orph;
        // labels: wherever they were, unless the previous row pass rewrote them (split / refresh)
        f.loc_gen = j.src_gen;
        f.loc_slot = j.src_slot;
        if (j.copy) {
          f.loc_gen = c.lid_gen;
          f.loc_slot = q;
        }
        f.ll_tot = jl.h_ll_tot;
        f.ll_orph = jl.h_ll_orph;
        if (j.active) {
          const ChildVals cv = child_values(S, j.rule, j.cnt, j.p_q_st, j.p_value, a.cnts, a.aL, a.aN, z0, z1, leaf_sd);
          const int cL = cv.cL, cR = cv.cR;
          f.loc_gen = c.lid_gen;  // the row pass wrote this particle's labels here
          f.loc_slot = q;
          f.ok = cv.ok;
          f.node = j.node;
          f.cL = cL;
          f.aL = a.aL; f.bL = a.bL; f.c2L = a.c2L;
          f.ccL = j.ccL;
          f.sse_orph = j.h_sse_orph + (double)a.c2N * S.sc.inv_c2;
          f.ll_orph = jl.h_ll_orph + al.llN;
          f.llL = al.llL;
          if (cv.ok == -1) {
            // [U] a one-hot split needs two distinct values: the grow fails and the node stays a
            // leaf.  No row was relabelled except rows with a missing split value, which the pass
            // dropped; the leaf sheds them (identity when there are none).
            f.sseL = pgb_leaf_sse(cL, f.bL, f.c2L, j.p_value, S.sc.inv_c1, S.sc.inv_c2);
            f.sse_tot = (j.h_sse_tot - j.p_sse) + f.sseL;
            f.ll_tot = (jl.h_ll_tot - jl.p_ll) + al.llL;
          } else {
            f.cR = cR;
            f.var = j.var; f.split = j.v; f.new_label = j.new_label;
            f.ccR = j.ccR;
            f.depth = (uint8_t)j.p_depth; f.label = (uint8_t)j.label;
            f.aR = cv.aR;
            f.bR = j.p_q_r - a.bL - a.bN;
            f.c2R = j.p_q_r2 - a.c2L - a.c2N;
            f.vL = cv.vL;
            f.vR = cv.vR;
            f.sseL = pgb_leaf_sse(cL, f.bL, f.c2L, f.vL, S.sc.inv_c1, S.sc.inv_c2);
            f.sseR = pgb_leaf_sse(cR, f.bR, f.c2R, f.vR, S.sc.inv_c1, S.sc.inv_c2);
            f.svarL = f.svarR = -1;
            f.slopeL = f.xbarL = f.slopeR = f.xbarR = 0.0;
            if constexpr (LIN) {  // [U] fast_linear_fit on the split variable
              lk = lin_children(S, &S.accu[((size_t)(par ^ 1) * MAXP + q) * ACC_PER], j.var, cL, cR,
                                f.aL, f.aR, it, (uint32_t)(r - 1), (uint32_t)q);
              f.svarL = lk.svarL; f.slopeL = lk.slopeL; f.xbarL = lk.xbarL;
              f.svarR = lk.svarR; f.slopeR = lk.slopeR; f.xbarR = lk.xbarR;
              if (normal) {  // the weight of a linear leaf: SSE in closed form
                if (lk.svarL >= 0) f.sseL = pgb_lin_sse(f.sseL, lk.fL, lk.urL, f.bL, S.sc.inv_c1);
                if (lk.svarR >= 0) f.sseR = pgb_lin_sse(f.sseR, lk.fR, lk.urR, f.bR, S.sc.inv_c1);
              }
            }
            f.sse_tot = ((j.h_sse_tot - j.p_sse) + f.sseL) + f.sseR;
            f.llR = al.llR;
            f.ll_tot = ((jl.h_ll_tot - jl.p_ll) + al.llL) + al.llR;
            f.n_nodes = j.h_n_nodes + 2;
            f.n_leaves = j.h_n_leaves + 1;
          }
        }
        if constexpr (MK && LIN) {  // (constant leaves: waves 1..3 do this, see below)
          if (j.active)
          for (int k = 0; k < KX; ++k) {
            const long long pq = r1 ? root_A_x(S, par ^ 1, k) : S.jqx[((size_t)(par ^ 1) * MAXP + q) * KX + k];
            const double pv = r1 ? S.init_leaf : S.jvx[((size_t)(par ^ 1) * MAXP + q) * KX + k];
            const double* zz = S.jzx + (((size_t)(par ^ 1) * MAXP + q) * KX + k) * 2;
            s_finx[q][k] = child_values_x(S, f.ok, f.cL, f.cR, load_accx(S.accx, par ^ 1, q, k),
                                          load_accx(S.accx, par ^ 1, q, KX + k), pq, pv, zz[0], zz[1],
                                          leaf_sd_x(S, c, par, par ^ 1, k));
            if constexpr (LIN)
              if (f.ok == 1)
                lin_children_x(S, lk, s_finx[q][k], j.var, f.cL, f.cR, load_accx(S.accux, par ^ 1, q, k),
                               load_accx(S.accux, par ^ 1, q, KX + k));
          }
          if constexpr (LIN)
            if (j.active && f.ok == 1) {  // a further output may have made the leaf linear
              f.svarL = lk.svarL; f.slopeL = lk.slopeL; f.xbarL = lk.xbarL;
              f.svarR = lk.svarR; f.slopeR = lk.slopeR; f.xbarR = lk.xbarR;
            }
        }
        pending = f.next_pop < f.n_nodes;
        lw = normal ? (f.sse_tot + f.sse_orph) * (-0.5 * c.inv_sigma2)
                    : (double)(f.ll_tot + f.ll_orph) * S.sc.inv_cl;
        // [U] ParticleTree.log_weight = 0 until the particle's first successful grow (upstream-semantics switch,
        // pgbart_spec.h; a root-only tree has never grown, and copies made by the resampling carry the tree)
        if ((S.compat & PGB_COMPAT_FRESH_WEIGHT_ZERO) && f.n_nodes == 1) lw = 0.0;
      }
      lwv[hq] = lw;
      popv[hq] = popn;
      ispv[hq] = isp;
      pend_any = pend_any || pending;
      if (b == 0) {  // (see below)
        n_popped += __popcll(__ballot(isp && j.popped));
        n_active += __popcll(__ballot(isp && j.active));
        rt_lane += isp && j.active ? (long long)j.cnt : 0ll;
      }
      }  // (particles of this lane)
      TR(2);
      if (b == 0) {
        // particle steps / partitions / rows touched of the round the previous slot proposed, counted here
        // from its job records: 39 workgroups adding to one line at the end of a kernel serialise (~12 ns
        // each) and the kernel does not end before the last one is acknowledged
        const long long rt = wave_sum_dpp(rt_lane);
        if (tid == 63) {
          if (n_popped) atomicAdd(&S.counters[0], (unsigned long long)n_popped);
          if (n_active) {
            atomicAdd(&S.counters[6], (unsigned long long)n_active);
            atomicAdd(&S.counters[2], (unsigned long long)rt);
          }
        }
      }
      stop = __ballot(pend_any) == 0ull;
      int pick;
      if (!stop) {
        // [U] systematic resampling of particles 1..P-1: ancestor of new particle p
        const double ui = (u_res + (double)(p - 1)) / (double)Lc;
        pick = wave_pick<NH>(lwv, 1, Lc, ui);
      } else {
        // [U] get_particle_tree: final choice among all P particles (lane 0 = reference particle)
        if (tid == 0) lwv[0] = normal ? sse0 * (-0.5 * c.inv_sigma2) : sse0;
        pick = wave_pick<NH>(lwv, 0, P, c.u_fin);  // (iter, 0, 0, FINAL), drawn one slot ahead
      }
      // (stored only now: the load was requested when the job header arrived and its round trip runs
      //  under the weights, the scan and the pick instead of ending the finish stage)
#pragma unroll
      for (int hq = 0; hq < NH; ++hq)
        if (ispv[hq]) s_pop[tid + 64 * hq] = popv[hq];
      pick0 = pick;
      if (tid == 0) {
        s_i[0] = stop ? 1 : 0;
        s_i[1] = pick;
      }
    }
#if PGB_CTRL_FINX
    if constexpr (MK && !LIN) {
      // K-vector leaves: the children's values and sums of outputs 1..K-1 of every old particle's pending split.
      // Workgroup 0 of the previous slot's likelihood pass derived them for its job list (child_values_x on the
      // statistics this kernel would read) and left them in Dev::finx: waves 1..3 copy them side by side, one
      // contiguous read, where they used to derive them again per (particle, output) lane -- the job record, then
      // the counts and every output's sums, parent values and leaf noise: two dependent round trips of scattered
      // loads in front of the barrier below, 1.4 us after wave 0 was done at cfg5.  (Particles without a pending
      // split: stale entries, never read -- s_finx[anc] is consulted only when Fin::ok says there was one.)
      // (requested before the draws instead of behind them: no gain; wave 3 alone, whose draws are done early:
      //  slower -- profiles/r05_experiments.md section 4)
      if (tid >= 64) {
        const FinX* fx = S.finx + (size_t)(par ^ 1) * MAXP * KX;
        for (int e = tid - 64; e < P * KX; e += BT - 64) {
          const FinX t = fx[e];
          ChildX& d = s_finx[e / KX][e % KX];
          d.vL = t.vL; d.vR = t.vR; d.aL = t.aL; d.aR = t.aR;
          d.sL = d.sR = 0.0;
        }
      }
    }
#else
    if constexpr (MK && !LIN) {
      // K-vector leaves: the children's values of outputs 1..K-1 (a Philox draw, a Box-Muller pair and
      // two leaf values each) are independent of everything wave 0 does above, so waves 1..3 compute
      // them meanwhile -- output k on wave 1 + k % 3, old particle q on lane q -- instead of wave 0
      // doing K - 1 of those chains one after the other.  Same routine, same inputs.
      if (tid >= 64) {
        const int wv = tid >> 6;
#pragma unroll
        for (int hq = 0; hq < NH; ++hq) {
        const int q = (tid & 63) + 64 * hq;
        if (q >= 1 && q < P && JP[q].active) {
          unsigned long long cnts = 0;
#pragma unroll
          for (int cpy = 0; cpy < ACC_SLOTS; ++cpy)
            cnts += S.acc[((size_t)(par ^ 1) * MAXP + q) * ACC_PER + cpy * ACC_STRIDE].cnts;
          const int cL = (int)(cnts & 0xFFFFFFFFull), cN = (int)(cnts >> 32), cR = JP[q].cnt - cL - cN;
          const int ok = (cR == 0 && pgb_empty_right_fails(JP[q].rule, S.compat)) ? -1 : 1;  // as child_values decides
          for (int k = wv - 1; k < KX; k += 3) {
            const long long pq = r1 ? root_A_x(S, par ^ 1, k) : S.jqx[((size_t)(par ^ 1) * MAXP + q) * KX + k];
            const double pv = r1 ? S.init_leaf : S.jvx[((size_t)(par ^ 1) * MAXP + q) * KX + k];
            const double* zz = S.jzx + (((size_t)(par ^ 1) * MAXP + q) * KX + k) * 2;
            s_finx[q][k] = child_values_x(S, ok, cL, cR, load_accx(S.accx, par ^ 1, q, k),
                                          load_accx(S.accx, par ^ 1, q, KX + k), pq, pv, zz[0], zz[1],
                                          leaf_sd_x(S, c, par, par ^ 1, k));
          }
        }
        }
      }
    }
#endif
    __syncthreads();
    stop = s_i[0] != 0;
    if (stop) {
      sel = s_i[1];
      anc = p;  // no resampling in the final slot: particle p finishes itself
    } else {
      anc = s_i[1];
    }
    TR(3);
    // -------- new particle p := old particle anc with its pending split applied (all threads)
    {
      const DPart* A = &OT[anc];
      const Fin& f = s_fin[anc];
      const int nn = f.nn_old;
      // (waves 1..3 copy: wave 0 goes straight on to the proposal -- it would otherwise sit out the
      //  round trip of these loads before popping its node; nothing below reads the new table until
      //  the workgroup barrier of a tree's last slot)
      for (int i = tid - 64; i < nn; i += BT - 64) {
        if (i < 0) continue;
        DNode z = A->nd[i];
        if (r1 && i == 0) {  // root statistics (see above)
          z.q_st = ia.A;
          z.q_r = normal ? ia.B : ia.C;  // Bernoulli families keep the node's log-likelihood here
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
          z.q_r = normal ? f.bL : f.llL;
          z.q_r2 = f.c2L;
          z.sse = f.sseL;
          z.cc_row = f.ccL;
        }
        me->nd[i] = z;
      }
      if constexpr (LIN) {  // linear parts of the leaves
        const LinP* la = S.plin + ((size_t)par * MAXP + anc) * MAXN;
        LinP* lm = S.plin + ((size_t)(par ^ 1) * MAXP + p) * MAXN;
        for (int i = tid; i < nn; i += BT) lm[i] = la[i];
        if (f.ok == 1 && tid < 2) {
          lm[nn + tid] = tid == 0 ? LinP{f.slopeL, f.xbarL, (long long)f.svarL}
                                  : LinP{f.slopeR, f.xbarR, (long long)f.svarR};
        }
      }
      if constexpr (MK) {  // extension outputs of the node table
        const size_t so = ((size_t)par * MAXP + anc) * MAXN * KX, dn = ((size_t)(par ^ 1) * MAXP + p) * MAXN * KX;
        // (waves 1..3 only, like the node table itself: wave 0 goes on to the proposal)
        for (int e = tid - 64; e < nn * KX; e += BT - 64) {
          if (e < 0) continue;
          const int i = e / KX, k = e % KX;
          double v = S.pvx[so + e];
          long long qv = S.pqx[so + e];
          if (r1 && i == 0) qv = root_A_x(S, par ^ 1, k);
          if (f.ok == -1 && i == f.node) qv = s_finx[anc][k].aL;
          S.pvx[dn + e] = v;
          S.pqx[dn + e] = qv;
          if constexpr (LIN) S.psx[dn + e] = S.psx[so + e];
        }
        if (f.ok == 1)
          for (int e = tid - 128; e < 2 * KX; e += BT) {
            if (e < 0) continue;
            const int ch = e / KX, k = e % KX;
            S.pvx[dn + (size_t)(nn + ch) * KX + k] = ch ? s_finx[anc][k].vR : s_finx[anc][k].vL;
            S.pqx[dn + (size_t)(nn + ch) * KX + k] = ch ? s_finx[anc][k].aR : s_finx[anc][k].aL;
            if constexpr (LIN) S.psx[dn + (size_t)(nn + ch) * KX + k] = ch ? s_finx[anc][k].sR : s_finx[anc][k].sL;
          }
      }
      if (f.ok == 1 && tid >= BT - 2) {
        const bool isL = tid == BT - 2;
        DNode z;
        memset(&z, 0, sizeof z);
        z.var = -1;
        z.depth = f.depth + 1;
        z.label = isL ? f.label : (uint8_t)f.new_label;
        z.cnt = isL ? f.cL : f.cR;
        z.q_st = isL ? f.aL : f.aR;
        z.q_r = normal ? (isL ? f.bL : f.bR) : (isL ? f.llL : f.llR);
        z.q_r2 = isL ? f.c2L : f.c2R;
        z.value = isL ? f.vL : f.vR;
        z.sse = isL ? f.sseL : f.sseR;
        z.cc_row = isL ? f.ccL : f.ccR;
        me->nd[nn + (isL ? 0 : 1)] = z;
      }
    }
    if (!stop) {
      // =============================================================== plain round: propose on wave 0
      // Waves 1..3 are done once their copy is issued.  Wave 0 proposes round r for new particle p
      // ([U] ParticleTree.sample_tree / grow_tree) with wave-uniform values only: every lane reads the
      // ancestor's record from LDS (the reads go out together), no value travels through lane 0, LDS and
      // a workgroup barrier.  The slots that end or start a tree take the general path further down.
      if (tid >= 64) return;
      const int a0 = pick0;  // the ancestor, in a scalar register (s_i[1] holds the same value)
      const Fin& F = s_fin[a0];
      const int np = F.next_pop, nn_old = F.nn_old, f_nodes = F.n_nodes, f_leaves = F.n_leaves;
      const int f_gen = F.loc_gen, f_slot = F.loc_slot;
      const double f_sse_tot = F.sse_tot, f_sse_orph = F.sse_orph;
      const double u_coin = s_pre[0][0];
      const int jvar = s_i[8], jrule = s_i[10], jnan = s_i[12];
      const bool has = np < f_nodes;
      DNode nd;
      memset(&nd, 0, sizeof nd);
      if (has) {
        if (np < nn_old) {  // an old node of the ancestor (prefetched by lane a0 of the finish stage)
          nd = s_pop[a0];
          if (r1 && np == 0) nd.q_st = ia.A, nd.q_r = normal ? ia.B : ia.C, nd.q_r2 = ia.C, nd.sse = root_sse;
        } else {  // one of the children the ancestor's pending split just created
          const bool isL = np == nn_old;
          nd.var = -1;
          nd.depth = F.depth + 1;
          nd.label = isL ? F.label : (uint8_t)F.new_label;
          nd.cnt = isL ? F.cL : F.cR;
          nd.q_st = isL ? F.aL : F.aR;
          nd.q_r = normal ? (isL ? F.bL : F.bR) : (isL ? F.llL : F.llR);
          nd.q_r2 = isL ? F.c2L : F.c2R;
          nd.sse = isL ? F.sseL : F.sseR;
          nd.value = isL ? F.vL : F.vR;
          nd.cc_row = isL ? F.ccL : F.ccR;
        }
      }
      const double pl = (has && nd.depth < PGB_MAX_DEPTH) ? s_prior[nd.depth] : 1.0;
      const bool attempt = has && (pl < u_coin) && (f_nodes + 2 <= MAXN) && (nd.cnt >= 2);
      TR(5);
      SplitRow sr;
      sr.found = 0;
      sr.vkey = 0;
      sr.v = 0.0;
      if (attempt) {
        TR(6);
        sr = select_split_row<KEYS>(S, jvar, jrule == PGB_RULE_SUBSET, f_gen, f_slot, nd.cnt, nd.cc_row, nd.label, s_pre[0],
                              s_pre1[0], KEYS ? xk16 : nullptr, TR_REC);
      }
      TR(7);
      Job job;
      memset(&job, 0, sizeof job);
      job.src_gen = f_gen;
      job.src_slot = f_slot;
      job.h_n_nodes = f_nodes;
      job.h_n_leaves = f_leaves;
      job.h_next_pop = np + (has ? 1 : 0);
      job.h_sse_tot = f_sse_tot;
      job.h_sse_orph = f_sse_orph;
      job.popped = has ? 1 : 0;
      if (sr.found) {
        job.active = 1;
        job.node = np;
        job.label = nd.label;
        job.new_label = f_leaves;
        job.var = jvar;
        job.rule = jrule;
        job.check_nan = jnan;
        job.ccL = ((r * MAXP + p) * 2);
        job.ccR = job.ccL + 1;
        job.cnt = nd.cnt;
        job.vkey = sr.vkey;
        job.v = sr.v;
        // parent statistics and the children's leaf noise travel with the job
        job.p_q_st = nd.q_st;
        job.p_q_r = nd.q_r;
        job.p_q_r2 = nd.q_r2;
        job.p_sse = nd.sse;
        job.p_value = nd.value;
        job.p_depth = nd.depth;
        job.z0 = s_ahead[0][0];
        job.z1 = s_ahead[0][1];
      }
      {
        const int dst = (c.lid_gen + 1) % NGEN;
        job.copy = (!job.active && f_slot >= 0 && f_gen == (dst + 1) % NGEN) ? 1 : 0;
      }
      if constexpr (MK)
      if (job.active && tid < KX) {  // extension outputs of the node being split: output k on lane k
        const int k = tid;           // (one lane after the other paid two dependent global loads per output)
        {
          long long pq;
          double pv;
          if (np < nn_old) {
            const size_t so = ((size_t)par * MAXP + a0) * MAXN * KX + (size_t)np * KX + k;
            pq = (r1 && np == 0) ? root_A_x(S, par ^ 1, k) : S.pqx[so];
            pv = S.pvx[so];
            if (F.ok == -1 && np == F.node) pq = s_finx[a0][k].aL;
          } else {
            const bool isL = np == nn_old;
            pq = isL ? s_finx[a0][k].aL : s_finx[a0][k].aR;
            pv = isL ? s_finx[a0][k].vL : s_finx[a0][k].vR;
          }
          S.jqx[((size_t)par * MAXP + p) * KX + k] = pq;
          S.jvx[((size_t)par * MAXP + p) * KX + k] = pv;
          S.jzx[(((size_t)par * MAXP + p) * KX + k) * 2] = s_aheadx[0][k][0];
          S.jzx[(((size_t)par * MAXP + p) * KX + k) * 2 + 1] = s_aheadx[0][k][1];
        }
      }
      if (tid == 0) {
        JN[p] = job;
        if (!normal)  // the node's log-likelihood lives in q_r for these families
          S.jobl[par * MAXP + p] = JobL{job.active ? nd.q_r : 0, F.ll_tot, F.ll_orph, 0};
        me->n_nodes = f_nodes;
        me->n_leaves = f_leaves;
        me->next_pop = job.h_next_pop;
        me->loc_gen = f_gen;
        me->loc_slot = f_slot;
        me->sse_tot = f_sse_tot;
        me->sse_orph = f_sse_orph;
      }
      TRV(17, r);
      TRV(16, attempt);
      TRV(18, 0);
      TRV(19, 0);
      TR(8);
      if (b == 0 && tid == 0) {
        cmd->dst_gen = (c.lid_gen + 1) % NGEN;
        cmd->st_cur = c.st_cur;
        cmd->kind = CMD_PARTITION;
        Ctrl o = c;
        o.slot_no = c.slot_no + 1;
        o.leaf_sd = leaf_sd;
        if constexpr (MK)
          for (int k = 0; k < KX; ++k) S.lsdx[(par ^ 1) * KXMAX + k] = leaf_sd_x(S, c, par, par ^ 1, k);
        o.pend_leafsd = 0;
        o.lid_gen = (c.lid_gen + 1) % NGEN;
        o.sse0 = sse0;
        o.u_res = s_ahead[0][2];
        o.u_fin = s_ahead[0][3];
        o.phase = PH_ROUND;
        o.round = r + 1;
        atomicAdd(&S.counters[3], 1ull);  // round r-1 is complete
        *co = o;
        TRX(11, true);
      }
      return;
    }
  } else {
    __syncthreads();  // waves 1/2 have published their draws
  }
  TR(4);

  // =================================================================== end of a tree
  // bookkeeping of the accepted tree; then (if another tree follows) fall through and propose
  // its round 0 in this very slot
  bool fresh = begin;       // propose round 0 of a new tree (fresh stump) instead of round r
  int tree_new = c.lower + c.k;  // PH_BEGIN: the tree to start
  bool has_init = begin;
  int lower_next = c.lower, k_next = c.k, batch_next = c.batch_n;
  bool more = true;
  if (stop) {
    const Fin& F = s_fin[p];
    __syncthreads();  // the node copy above is complete (this workgroup reads it back below)
    const int tree_old = c.lower + c.k;
    more = (c.k + 1 < c.batch_n);
    const bool next_step = (!more && c.steps_left > 1);
    k_next = c.k + 1;
    if (!more) {
      int upper = c.lower + c.batch_n;
      lower_next = upper < S.m ? upper : 0;
      k_next = 0;
      int bs = c.tune ? S.batch_tune : S.batch_draw;
      int up2 = lower_next + bs;
      if (up2 > S.m) up2 = S.m;
      batch_next = up2 - lower_next;
    }
    tree_new = lower_next + k_next;
    has_init = more || next_step;
    fresh = has_init;

    if (tid == 0) {  // particle header (kept for inspection / export)
      me->n_nodes = F.n_nodes;
      me->n_leaves = F.n_leaves;
      me->next_pop = F.next_pop;
      me->loc_gen = F.loc_gen;
      me->loc_slot = F.loc_slot;
      me->sse_tot = F.sse_tot;
      me->sse_orph = F.sse_orph;
    }
    if (sel >= 1 && p == sel) {
      // accepted a grown particle: store it as the tree and publish its label->value table
      DTree* T = &S.trees[tree_old];
      const int nn = F.n_nodes;
      for (int i = tid; i < nn; i += BT) T->nd[i] = me->nd[i];
      if (tid == 0) {
        T->n_nodes = nn;
        T->n_leaves = F.n_leaves;
        cmd->sel_gen = F.loc_gen;
        cmd->sel_slot = F.loc_slot;  // may be -1 (untouched root labels)
      }
      build_lv(me->nd, nn, cmd->lv_new);
      if constexpr (LIN) {
        const LinP* lm = S.plin + ((size_t)(par ^ 1) * MAXP + p) * MAXN;
        for (int i = tid; i < nn; i += BT) S.tlin[(size_t)tree_old * MAXN + i] = lm[i];
        build_lvl(me->nd, nn, lm, S.lvl + ((size_t)par * 2 + 0) * 256);
      }
      if constexpr (MK) {  // extension outputs: store with the tree, publish label->value tables
        const size_t pn = ((size_t)(par ^ 1) * MAXP + p) * MAXN * KX, tn = (size_t)tree_old * MAXN * KX;
        for (int e = tid; e < nn * KX; e += BT) S.tvx[tn + e] = S.pvx[pn + e];
        build_lvx(me->nd, nn, S.pvx + pn, KX, S.lvx + ((size_t)par * 2 + 0) * 256 * KX);
        if constexpr (LIN) {
          for (int e = tid; e < nn * KX; e += BT) S.tsx[tn + e] = S.psx[pn + e];
          build_lvx(me->nd, nn, S.psx + pn, KX, S.lsx + ((size_t)par * 2 + 0) * 256 * KX);
        }
      }
    }
    if (b == 0) {
      if constexpr (LIN) {
        if (sel == 0)
          build_lvl(S.trees[tree_old].nd, S.trees[tree_old].n_nodes, S.tlin + (size_t)tree_old * MAXN,
                    S.lvl + ((size_t)par * 2 + 0) * 256);
        if (has_init && tree_new != tree_old)
          build_lvl(S.trees[tree_new].nd, S.trees[tree_new].n_nodes, S.tlin + (size_t)tree_new * MAXN,
                    S.lvl + ((size_t)par * 2 + 1) * 256);
      }
      if constexpr (MK) {
        if (sel == 0)
          build_lvx(S.trees[tree_old].nd, S.trees[tree_old].n_nodes, S.tvx + (size_t)tree_old * MAXN * KX, KX,
                    S.lvx + ((size_t)par * 2 + 0) * 256 * KX);
        if (has_init && tree_new != tree_old)
          build_lvx(S.trees[tree_new].nd, S.trees[tree_new].n_nodes, S.tvx + (size_t)tree_new * MAXN * KX, KX,
                    S.lvx + ((size_t)par * 2 + 1) * 256 * KX);
        if constexpr (LIN) {
          if (sel == 0)
            build_lvx(S.trees[tree_old].nd, S.trees[tree_old].n_nodes, S.tsx + (size_t)tree_old * MAXN * KX, KX,
                      S.lsx + ((size_t)par * 2 + 0) * 256 * KX);
          if (has_init && tree_new != tree_old)
            build_lvx(S.trees[tree_new].nd, S.trees[tree_new].n_nodes, S.tsx + (size_t)tree_new * MAXN * KX, KX,
                      S.lsx + ((size_t)par * 2 + 1) * 256 * KX);
        }
      }
      if (sel == 0) {  // the old tree is kept: nobody writes S.trees[tree_old] in this slot
        if (tid == 0) {
          cmd->sel_slot = -2;
          cmd->sel_gen = 0;
        }
        if constexpr (!SPEC_LV) build_lv(S.trees[tree_old].nd, S.trees[tree_old].n_nodes, cmd->lv_new);
      }
      // label table of the next tree to update (a different tree unless m == 1)
      if constexpr (!SPEC_LV)
        if (has_init && tree_new != tree_old)
          build_lv(S.trees[tree_new].nd, S.trees[tree_new].n_nodes, cmd->lv_next);
    }
    // Bookkeeping that needs the accepted tree's split variables is done by the workgroup
    // that owns a complete copy of that tree.
    const bool owner = (sel == 0) ? (b == 0) : (p == sel);
    if (owner) {
      const DNode* snd = sel == 0 ? S.trees[tree_old].nd : me->nd;
      const int nn = sel == 0 ? S.trees[tree_old].n_nodes : F.n_nodes;
      if (c.tune) {
        // [U] the sampler is rebuilt from the weights BEFORE this tree's counts are added; weights
        // and prefix sums are double-buffered (other workgroups read the current ones in this slot)
        long long* alpha_o = S.alpha + (size_t)(c.alpha_cur ^ 1) * S.p;
        if (rebuild) {
          long long* cdf_o = S.cdfS + (size_t)(c.cdf_cur ^ 1) * S.p;
          if (tid < 64) {
            long long carry = 0;
            for (int base = 0; base < S.p; base += 64) {
              const int j = base + tid;
              const long long run = wave_sum_dpp(j < S.p ? alpha[j] : 0) + carry;
              if (j < S.p) cdf_o[j] = run;
              carry = ((long long)__builtin_amdgcn_readlane((int)(run >> 32), 63) << 32) |
                      (unsigned)__builtin_amdgcn_readlane((int)run, 63);
            }
          }
        }
        for (int j = tid; j < S.p; j += BT) alpha_o[j] = alpha[j];
        __syncthreads();
        // (one thread per node, integer atomics: a serial loop of dependent global read-modify-writes on
        //  one thread cost this workgroup -- and with it the slot -- a few microseconds per tree)
        for (int i = tid; i < nn; i += BT) {
          const int v = snd[i].var;
          if (v >= 0) atomicAdd((unsigned long long*)&alpha_o[v], (unsigned long long)S.alpha_unit);
        }
      } else {
        for (int i = tid; i < nn; i += BT) {
          const int v = snd[i].var;
          if (v >= 0) atomicAdd(&S.vi[v], 1);
        }
      }
      if (tree_new == tree_old && has_init) {  // m == 1 corner: next update is this very tree
        __syncthreads();
        build_lv(snd, nn, cmd->lv_next);
        if constexpr (LIN)
          build_lvl(snd, nn, sel == 0 ? S.tlin + (size_t)tree_old * MAXN
                                      : S.plin + ((size_t)(par ^ 1) * MAXP + p) * MAXN,
                    S.lvl + ((size_t)par * 2 + 1) * 256);
        if constexpr (MK)
          build_lvx(snd, nn, sel == 0 ? S.tvx + (size_t)tree_old * MAXN * KX
                                      : S.pvx + ((size_t)(par ^ 1) * MAXP + p) * MAXN * KX,
                    KX, S.lvx + ((size_t)par * 2 + 1) * 256 * KX);
        if constexpr (MK && LIN)
          build_lvx(snd, nn, sel == 0 ? S.tsx + (size_t)tree_old * MAXN * KX
                                      : S.psx + ((size_t)(par ^ 1) * MAXP + p) * MAXN * KX,
                    KX, S.lsx + ((size_t)par * 2 + 1) * 256 * KX);
      }
    }
    if (b == 0 && tid == 0) {
      cmd->tree_old = tree_old;
      cmd->tune = c.tune;
      cmd->rs_count = c.rs_count + (c.tune ? 1 : 0);
      atomicAdd(&S.counters[1], 1ull);
      atomicAdd(&S.counters[3], 1ull);
    }
    if (!has_init) {  // last tree of the last requested step
      if (b == 0 && tid == 0) {
        cmd->kind = CMD_FINAL;
        cmd->st_cur = c.st_cur;
        Ctrl o = c;
        o.slot_no = c.slot_no + 1;
        o.leaf_sd = leaf_sd;
        if constexpr (MK)
          for (int k = 0; k < KX; ++k) S.lsdx[(par ^ 1) * KXMAX + k] = leaf_sd_x(S, c, par, par ^ 1, k);
        o.rs_count = c.rs_count + (c.tune ? 1 : 0);
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
        // progress word the host polls (the row pass of this slot is still to run)
        __hip_atomic_store(S.host_flag, (unsigned long long)o.steps_done, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (tid == 0) {
        Job z;
        memset(&z, 0, sizeof z);
        JN[p] = z;
      }
      return;
    }
  } else if (begin && b == 0) {
    build_lv(S.trees[tree_new].nd, S.trees[tree_new].n_nodes, cmd->lv_next);
    if constexpr (LIN)
      build_lvl(S.trees[tree_new].nd, S.trees[tree_new].n_nodes, S.tlin + (size_t)tree_new * MAXN,
                S.lvl + ((size_t)par * 2 + 1) * 256);
    if constexpr (MK)
      build_lvx(S.trees[tree_new].nd, S.trees[tree_new].n_nodes, S.tvx + (size_t)tree_new * MAXN * KX, KX,
                S.lvx + ((size_t)par * 2 + 1) * 256 * KX);
    if constexpr (MK && LIN)
      build_lvx(S.trees[tree_new].nd, S.trees[tree_new].n_nodes, S.tsx + (size_t)tree_new * MAXN * KX, KX,
                S.lsx + ((size_t)par * 2 + 1) * 256 * KX);
  }

  // =================================================================== propose
  // [U] ParticleTree.sample_tree / grow_tree for new particle p: round r of the current tree, or
  // round 0 of the tree this slot starts (fresh stump, [U] init_particles)
  const int set = fresh ? 1 : 0;
  const int rr = fresh ? 0 : r;  // round of the proposal
  if (fresh) {
    __syncthreads();  // every reader of s_fin[] of the finished tree is done
    if (tid == 0) {  // slot 0 of s_fin is never a particle: it holds the fresh stump
      Fin f0;
      memset(&f0, 0, sizeof f0);
      f0.nn_old = 1;
      f0.n_nodes = 1;
      f0.n_leaves = 1;
      f0.next_pop = 0;
      f0.loc_gen = 0;
      f0.loc_slot = -1;
      s_fin[0] = f0;
      // root node; its statistics are patched in by the next slot
      DNode z;
      memset(&z, 0, sizeof z);
      z.var = -1;
      z.cc_row = -1;
      z.cnt = (int32_t)S.n;
      z.value = S.init_leaf;
      me->nd[0] = z;
      if constexpr (LIN) S.plin[((size_t)(par ^ 1) * MAXP + p) * MAXN] = LinP{0.0, 0.0, -1};
      if constexpr (MK)
      for (int k = 0; k < KX; ++k) {
        S.pvx[((size_t)(par ^ 1) * MAXP + p) * MAXN * KX + k] = S.init_leaf;
        S.pqx[((size_t)(par ^ 1) * MAXP + p) * MAXN * KX + k] = 0;  // patched by the next slot
        if constexpr (LIN) S.psx[((size_t)(par ^ 1) * MAXP + p) * MAXN * KX + k] = 0.0;
      }
    }
    __syncthreads();
  }
  const Fin& F = s_fin[fresh ? 0 : anc];
  Job job;
  memset(&job, 0, sizeof job);
  job.src_gen = F.loc_gen;
  job.src_slot = F.loc_slot;
  job.h_n_nodes = F.n_nodes;
  job.h_n_leaves = F.n_leaves;
  job.h_next_pop = F.next_pop;
  job.h_sse_tot = F.sse_tot;
  job.h_sse_orph = F.sse_orph;
  bool attempt = false;
  int node = -1;
  DNode nd;
  memset(&nd, 0, sizeof nd);
  if (tid == 0) {
    const int np = F.next_pop;
    if (np < F.n_nodes) {
      node = np;
      // the popped node: the root of a fresh stump, an old node of the ancestor, or one of the
      // children just created
      if (fresh) {
        nd.var = -1;
        nd.cc_row = -1;
        nd.cnt = (int32_t)S.n;
        nd.value = S.init_leaf;
      } else if (np < F.nn_old) {
        nd = s_pop[anc];
        if (r == 1 && np == 0) nd.q_st = ia.A, nd.q_r = normal ? ia.B : ia.C, nd.q_r2 = ia.C,
            nd.sse = pgb_leaf_sse(S.n, ia.B, ia.C, S.init_leaf, S.sc.inv_c1, S.sc.inv_c2);
      } else {
        const bool isL = np == F.nn_old;
        nd.var = -1;
        nd.depth = F.depth + 1;
        nd.label = isL ? F.label : (uint8_t)F.new_label;
        nd.cnt = isL ? F.cL : F.cR;
        nd.q_st = isL ? F.aL : F.aR;
        nd.q_r = normal ? (isL ? F.bL : F.bR) : (isL ? F.llL : F.llR);
        nd.q_r2 = isL ? F.c2L : F.c2R;
        nd.sse = isL ? F.sseL : F.sseR;
        nd.value = isL ? F.vL : F.vR;
        nd.cc_row = isL ? F.ccL : F.ccR;
      }
      double pl = nd.depth < PGB_MAX_DEPTH ? s_prior[nd.depth] : 1.0;
      attempt = (pl < s_pre[set][0]) && (F.n_nodes + 2 <= MAXN) && (nd.cnt >= 2);
      s_i[5] = nd.cnt;
      s_i[6] = nd.cc_row;
      s_i[7] = nd.label;
    }
    s_i[3] = attempt ? 1 : 0;
    s_i[4] = node;
  }
  __syncthreads();
  attempt = s_i[3] != 0;
  node = s_i[4];
  job.h_next_pop = F.next_pop + (node >= 0 ? 1 : 0);
  TR(5);
  if (attempt) {
    const int ncnt = s_i[5], ncc = s_i[6], nlabel = s_i[7];
    // runs on wave 0 only (no workgroup barriers)
    if (tid < 64) {
      TR(6);
      const SplitRow sr = select_split_row<KEYS>(S, s_i[8 + set], s_i[10 + set] == PGB_RULE_SUBSET, job.src_gen, job.src_slot,
                                           ncnt, ncc, nlabel, s_pre[set], s_pre1[set], KEYS ? xk16 : nullptr);
      if (tid == 0) {
        s_i[14] = sr.found;  // (not s_i[0]: waves 1..3 read s_i[0..1] after the resampling barrier and no later barrier orders them)
        s_i[15] = sr.vkey;
        s_d[0] = sr.v;
      }
    }
    __syncthreads();
    if (s_i[14]) {
      const int j = s_i[8 + set];
      job.active = 1;
      job.node = node;
      job.label = nlabel;
      job.new_label = F.n_leaves;
      job.var = j;
      job.rule = s_i[10 + set];
      job.check_nan = s_i[12 + set];
      job.ccL = ((rr * MAXP + p) * 2);
      job.ccR = job.ccL + 1;
      job.cnt = ncnt;
      job.vkey = s_i[15];
      job.v = s_d[0];
    }
  }
  TR(7);
  // Labels are only rewritten when a particle splits.  A particle that idles keeps pointing at
  // its old generation; it is copied forward only when that generation is the next to be reused.
  {
    const int dst = (c.lid_gen + 1) % NGEN;
    job.copy = (!job.active && job.src_slot >= 0 && job.src_gen == (dst + 1) % NGEN) ? 1 : 0;
  }
  if (tid == 0) {
    if (job.active) {  // parent statistics travel with the job (the next slot needs nothing else)
      job.p_q_st = nd.q_st;
      job.p_q_r = nd.q_r;
      job.p_q_r2 = nd.q_r2;
      job.p_sse = nd.sse;
      job.p_value = nd.value;
      job.p_depth = nd.depth;
      job.z0 = s_ahead[set][0];
      job.z1 = s_ahead[set][1];
      if constexpr (MK)
      for (int k = 0; k < KX; ++k) {  // extension outputs of the node being split
        long long pq;
        double pv;
        if (fresh) {
          pq = 0;  // root sums are not known yet: patched by the next slot
          pv = S.init_leaf;
        } else if (node < F.nn_old) {
          const size_t so = ((size_t)par * MAXP + anc) * MAXN * KX + (size_t)node * KX + k;
          pq = (r == 1 && node == 0) ? root_A_x(S, par ^ 1, k) : S.pqx[so];
          pv = S.pvx[so];
          if (F.ok == -1 && node == F.node) pq = s_finx[anc][k].aL;
        } else {
          const bool isL = node == F.nn_old;
          pq = isL ? s_finx[anc][k].aL : s_finx[anc][k].aR;
          pv = isL ? s_finx[anc][k].vL : s_finx[anc][k].vR;
        }
        S.jqx[((size_t)par * MAXP + p) * KX + k] = pq;
        S.jvx[((size_t)par * MAXP + p) * KX + k] = pv;
        S.jzx[(((size_t)par * MAXP + p) * KX + k) * 2] = s_aheadx[set][k][0];
        S.jzx[(((size_t)par * MAXP + p) * KX + k) * 2 + 1] = s_aheadx[set][k][1];
      }
    }
    job.popped = node >= 0 ? 1 : 0;
    JN[p] = job;
    if (!normal)  // the node's log-likelihood lives in q_r for these families
      S.jobl[par * MAXP + p] = JobL{job.active ? nd.q_r : 0, F.ll_tot, F.ll_orph, 0};
    me->n_nodes = F.n_nodes;
    me->n_leaves = F.n_leaves;
    me->next_pop = job.h_next_pop;
    me->loc_gen = F.loc_gen;
    me->loc_slot = F.loc_slot;
    me->sse_tot = F.sse_tot;
    me->sse_orph = F.sse_orph;
  }
  TRV(17, r);
  TRV(16, attempt);
  TRV(18, fresh);
  TRV(19, stop);
  TR(8);
  if (b == 0 && tid == 0) {
    cmd->dst_gen = (c.lid_gen + 1) % NGEN;
    cmd->st_cur = c.st_cur;
    Ctrl o = c;
    o.slot_no = c.slot_no + 1;
    o.leaf_sd = leaf_sd;
    if constexpr (MK)
      for (int k = 0; k < KX; ++k) S.lsdx[(par ^ 1) * KXMAX + k] = leaf_sd_x(S, c, par, par ^ 1, k);
    o.pend_leafsd = 0;
    o.lid_gen = (c.lid_gen + 1) % NGEN;
    o.sse0 = sse0;
    o.u_res = s_ahead[set][2];
    o.u_fin = s_ahead[set][3];
    o.phase = PH_ROUND;
    if (!fresh) {
      cmd->kind = CMD_PARTITION;
      o.round = r + 1;
      atomicAdd(&S.counters[3], 1ull);  // round r-1 is complete
    } else {
      // this slot starts a tree: FINAL of the previous one (if any) + INIT + round 0 in one row pass
      cmd->kind = (stop ? CMD_FINAL : 0) | CMD_INIT | CMD_PARTITION;
      cmd->tree_new = tree_new;
      o.round = 1;
      o.iter = c.iter + 1;
      o.st_cur = c.st_cur ^ 1;  // INIT writes sum_trees_noi into the other buffer
      if (stop) {
        o.rs_count = c.rs_count + (c.tune ? 1 : 0);
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
    if (fresh && stop && !more)  // a step completed (its FINAL runs in this slot's row pass)
      __hip_atomic_store(S.host_flag, (unsigned long long)o.steps_done, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    TRX(11, true);
  }
}

