// pgb_checkpoint.h -- part of pgbart_hip.hip (not a standalone header): checkpoint / resume, profiling and debug entry points.
// ---- checkpoint / resume ------------------------------------------------------------------
// The chain image of include/pgbart_image.h -- the record every backend writes and reads, so that a chain continues
// bit for bit on another backend (the CPU restatement picks up a chain after the GPU's burn-in: the steady-state
// parity tests), on the other build of this library (64 <-> 128 particles) or on a fresh handle.  Of this backend's
// device state only what a chain IS between asteps travels: sum_trees, the accepted trees (node tables, K-vector /
// linear leaf parts, one label byte per tree and row), the running-sd accumulators, the split weights and the prefix
// sums in use, and the scalars of the control word.  Everything else -- particles, label generations, job records,
// chunk counts, command blocks, the label -> value tables built one slot ahead -- lives for one tree update and is
// rebuilt by the first slot of the next astep (k_begin -> PH_BEGIN).

// The control word of the idle device with the pending leaf_sd of the last FINAL pass resolved (the first idle slot
// after a step does the same on the device; pgb_get_state and the image must not depend on whether one has run).
static int read_idle_ctrl(pgb_handle* h, Ctrl* c_out, double* leaf_sd /* [K] */) {
  Dev& d = h->d;
  Ctrl c;
  InitAcc ia[IA_SLOTS];
  HIPCHK(hipMemcpyAsync(&c, &d.ctrl[h->slot & 1], sizeof c, hipMemcpyDeviceToHost, h->stream));
  const size_t ia_read = (size_t)((h->slot & 1) ^ 1);  // the last slot's sums
  HIPCHK(hipMemcpyAsync(ia, &d.initacc[ia_read * IA_SLOTS], sizeof ia, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  const bool pend = c.pend_leafsd != 0;
  if (leaf_sd) {
    long long qstd = 0;
    for (int k = 0; k < IA_SLOTS; ++k) qstd += ia[k].QSTD;
    leaf_sd[0] = pend ? pgb_tuned_leaf_sd(c.leaf_sd, c.pend_iter, qstd, d.sc.inv_c1, d.n) : c.leaf_sd;
    const int KX = d.K - 1;
    if (KX > 0) {
      std::vector<long long> ix((size_t)IA_SLOTS * 2 * KX);
      HIPCHK(hipMemcpy(ix.data(), d.iax + ia_read * IA_SLOTS * 2 * KX, ix.size() * sizeof(long long), hipMemcpyDeviceToHost));
      double lsdx[2 * KXMAX];
      HIPCHK(hipMemcpy(lsdx, d.lsdx, sizeof lsdx, hipMemcpyDeviceToHost));
      for (int k = 0; k < KX; ++k) {
        double v = lsdx[(h->slot & 1) * KXMAX + k];
        if (pend) {
          long long q = 0;
          for (int sl = 0; sl < IA_SLOTS; ++sl) q += ix[(size_t)sl * 2 * KX + KX + k];
          v = pgb_tuned_leaf_sd(v, c.pend_iter, q, d.sc.inv_c1, d.n);
        }
        leaf_sd[k + 1] = v;
      }
    }
  }
  if (c_out) *c_out = c;
  return PGB_OK;
}

extern "C" int pgb_get_state(pgb_handle* h, double* leaf_sd_out, int64_t* iter_out, int32_t* lower_out) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  JOIN_ASYNC(h);
  Ctrl c;
  double sd[PGB_MAX_OUTPUTS];
  const int rc = read_idle_ctrl(h, &c, sd);
  if (rc != PGB_OK) return rc;
  if (leaf_sd_out)
    for (int k = 0; k < h->d.K; ++k) leaf_sd_out[k] = sd[k];
  if (iter_out) *iter_out = c.iter;
  if (lower_out) *lower_out = c.lower;
  return PGB_OK;
}

// node counts of the m accepted trees (the first word of every DTree)
static int fetch_tree_sizes(pgb_handle* h, std::vector<int32_t>& nn) {
  nn.assign((size_t)h->d.m, 0);
  HIPCHK(hipMemcpy2D(nn.data(), sizeof(int32_t), h->d.trees, sizeof(DTree), sizeof(int32_t), (size_t)h->d.m,
                     hipMemcpyDeviceToHost));
  return PGB_OK;
}

extern "C" int pgb_checkpoint_size(pgb_handle* h, int64_t* bytes_out) {
  if (!h || !bytes_out) return fail(PGB_E_INVALID, "null argument");
  JOIN_ASYNC(h);
  REFUSE_POISONED(h);
  HIPCHK(hipStreamSynchronize(h->stream));
  std::vector<int32_t> nn;
  int rc = fetch_tree_sizes(h, nn);
  if (rc != PGB_OK) return rc;
  long long N = 0;
  for (int32_t v : nn) N += v;
  *bytes_out = pgb_image_bytes(h->d.n, h->d.p, h->d.m, h->d.K, (int32_t)N);
  return PGB_OK;
}

extern "C" int pgb_checkpoint_save(pgb_handle* h, void* host_buf, int64_t bytes) {
  if (!h || !host_buf) return fail(PGB_E_INVALID, "null argument");
  JOIN_ASYNC(h);
  if (!h->have_data || !h->have_y) return fail(PGB_E_INVALID, "set_data/set_response first");
  REFUSE_POISONED(h);  // (the state of an abandoned step is not a chain)
  Dev& d = h->d;
  const long long n = d.n, n_pad = d.n_pad;
  const int K = d.K, KX = d.K - 1, m = d.m, p = d.p;
  const bool lin = d.response != PGB_RESPONSE_CONSTANT;
  HIPCHK(hipStreamSynchronize(h->stream));  // step calls return idle; this also covers set_* uploads and queued idle slots
  Ctrl c;
  double sd[PGB_MAX_OUTPUTS] = {0};
  int rc = read_idle_ctrl(h, &c, sd);
  if (rc != PGB_OK) return rc;
  if (c.phase != PH_IDLE) return fail(PGB_E_STATE, "device not idle");
  std::vector<DTree> T((size_t)m);
  HIPCHK(hipMemcpy(T.data(), d.trees, sizeof(DTree) * (size_t)m, hipMemcpyDeviceToHost));
  long long N = 0;
  for (int t = 0; t < m; ++t) N += T[t].n_nodes;
  const int64_t need = pgb_image_bytes(n, p, m, K, (int32_t)N);
  if (bytes < need) return fail(PGB_E_INVALID, "checkpoint buffer too small");
  pgb_image_header hd;
  pgb_image_begin(host_buf, need, &h->s, (int32_t)N, pgb_backend_name(), &hd);
  if ((rc = fetch_counters(h, nullptr)) != PGB_OK) return rc;
  hd.iter = c.iter;
  hd.rs_count = c.rs_count;
  hd.lower = c.lower;
  hd.last_lower = h->last_lower;
  hd.last_n = h->last_n;
  for (int k = 0; k < K; ++k) hd.leaf_sd[k] = sd[k];
  hd.lik_param[0] = h->inv_sigma2;
  hd.lik_param[1] = h->lik_param2;
  hd.ctr = h->ctr;
  memcpy(host_buf, &hd, sizeof hd);
  pgb_image_view v;
  pgb_image_bind(host_buf, &hd, &v);
  // [K][n_pad] device rows -> [K][n]
  const size_t w = (size_t)n * sizeof(double), sp = (size_t)n_pad * sizeof(double);
  HIPCHK(hipMemcpy2D(v.sum_trees, w, d.st + (size_t)c.st_cur * K * n_pad, sp, w, (size_t)K, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy2D(v.rs_mean, w, d.rs_mean, sp, w, (size_t)K, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy2D(v.rs_m2, w, d.rs_m2, sp, w, (size_t)K, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(v.alpha, d.alpha + (size_t)c.alpha_cur * p, (size_t)p * sizeof(long long), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(v.cdf, d.cdfS + (size_t)c.cdf_cur * p, (size_t)p * sizeof(long long), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy2D(v.lid, (size_t)n, d.tree_lid, (size_t)n_pad, (size_t)n, (size_t)m, hipMemcpyDeviceToHost));
  std::vector<double> hx, hsx;
  std::vector<LinP> hl;
  if (KX > 0) {
    hx.resize((size_t)m * MAXN * KX);
    HIPCHK(hipMemcpy(hx.data(), d.tvx, hx.size() * sizeof(double), hipMemcpyDeviceToHost));
  }
  if (lin) {
    hl.resize((size_t)m * MAXN);
    HIPCHK(hipMemcpy(hl.data(), d.tlin, hl.size() * sizeof(LinP), hipMemcpyDeviceToHost));
    if (KX > 0) {
      hsx.resize((size_t)m * MAXN * KX);
      HIPCHK(hipMemcpy(hsx.data(), d.tsx, hsx.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
  }
  int32_t g = 0;
  for (int t = 0; t < m; ++t) {
    v.node_off[t] = g;
    for (int k = 0; k < T[t].n_nodes; ++k, ++g) {
      const DNode& z = T[t].nd[k];
      const bool leaf = z.var < 0;
      const bool islin = leaf && lin && hl[(size_t)t * MAXN + k].svar >= 0;
      v.var[g] = z.var;
      v.left[g] = leaf ? -1 : (int32_t)z.left;
      v.right[g] = leaf ? -1 : (int32_t)z.right;
      v.depth[g] = z.depth;
      v.label[g] = z.label;
      v.svar[g] = islin ? (int32_t)hl[(size_t)t * MAXN + k].svar : -1;
      v.count[g] = z.cnt;
      v.split[g] = leaf ? 0.0 : z.split;
      v.xbar[g] = islin ? hl[(size_t)t * MAXN + k].xbar : 0.0;
      for (int o = 0; o < K; ++o) {
        v.value[(size_t)g * K + o] = !leaf ? 0.0 : o ? hx[((size_t)t * MAXN + k) * KX + o - 1] : z.value;
        v.slope[(size_t)g * K + o] = !islin ? 0.0 : o ? hsx[((size_t)t * MAXN + k) * KX + o - 1] : hl[(size_t)t * MAXN + k].slope;
      }
    }
  }
  v.node_off[m] = g;
  return PGB_OK;
}

extern "C" int pgb_checkpoint_load(pgb_handle* h, const void* host_buf, int64_t bytes) {
  if (!h || !host_buf) return fail(PGB_E_INVALID, "null argument");
  JOIN_ASYNC(h);
  if (!h->have_data || !h->have_y) return fail(PGB_E_INVALID, "set_data/set_response first");
  const char* why = pgb_image_check(host_buf, bytes, &h->s);
  if (why) return fail(PGB_E_INVALID, why);
  Dev& d = h->d;
  const long long n = d.n, n_pad = d.n_pad;
  const int K = d.K, KX = d.K - 1, m = d.m, p = d.p;
  const bool lin = d.response != PGB_RESPONSE_CONSTANT;
  pgb_image_header hd;
  memcpy(&hd, host_buf, sizeof hd);
  pgb_image_view v;
  pgb_image_bind((void*)host_buf, &hd, &v);
  // every enqueued slot has run once the stream is idle -- also on a poisoned handle, whose device sits in the
  // middle of a round: the control word written below puts it back to the idle phase
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->stream_out) HIPCHK(hipStreamSynchronize(h->stream_out));
  Ctrl c;
  HIPCHK(hipMemcpy(&c, &d.ctrl[h->slot & 1], sizeof c, hipMemcpyDeviceToHost));
  // which of the double-buffered arrays are "current" is this handle's business: the image goes into those
  const int st_cur = c.st_cur & 1, alpha_cur = c.alpha_cur & 1, cdf_cur = c.cdf_cur & 1;
  const size_t w = (size_t)n * sizeof(double), dp = (size_t)n_pad * sizeof(double);
  HIPCHK(hipMemcpy2D(d.st + (size_t)st_cur * K * n_pad, dp, v.sum_trees, w, w, (size_t)K, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy2D(d.rs_mean, dp, v.rs_mean, w, w, (size_t)K, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy2D(d.rs_m2, dp, v.rs_m2, w, w, (size_t)K, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.alpha + (size_t)alpha_cur * p, v.alpha, (size_t)p * sizeof(long long), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.cdfS + (size_t)cdf_cur * p, v.cdf, (size_t)p * sizeof(long long), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy2D(d.tree_lid, (size_t)n_pad, v.lid, (size_t)n, (size_t)n, (size_t)m, hipMemcpyHostToDevice));
  {
    std::vector<DTree> T((size_t)m);
    std::vector<double> hx, hsx;
    std::vector<LinP> hl;
    if (KX > 0) hx.assign((size_t)m * MAXN * KX, 0.0);
    if (lin) hl.assign((size_t)m * MAXN, LinP{0.0, 0.0, -1});
    if (lin && KX > 0) hsx.assign((size_t)m * MAXN * KX, 0.0);
    memset(T.data(), 0, sizeof(DTree) * (size_t)m);
    for (int t = 0; t < m; ++t) {
      const int32_t base = v.node_off[t], nn = v.node_off[t + 1] - base;
      T[t].n_nodes = nn;
      for (int k = 0; k < nn; ++k) {
        const int32_t g = base + k;
        DNode& z = T[t].nd[k];
        const bool leaf = v.var[g] < 0;
        z.var = v.var[g];
        z.split = v.split[g];
        z.value = leaf ? v.value[(size_t)g * K] : 0.0;
        z.cnt = (int32_t)v.count[g];
        z.cc_row = -1;  // (chunk counts belong to the particle the tree once was)
        z.left = leaf ? 0 : (uint8_t)v.left[g];
        z.right = leaf ? 0 : (uint8_t)v.right[g];
        z.depth = (uint8_t)(v.depth[g] > 255 ? 255 : v.depth[g]);
        z.label = (uint8_t)v.label[g];
        if (leaf) T[t].n_leaves += 1;
        for (int o = 1; o < K; ++o) {
          hx[((size_t)t * MAXN + k) * KX + o - 1] = leaf ? v.value[(size_t)g * K + o] : 0.0;
          if (lin) hsx[((size_t)t * MAXN + k) * KX + o - 1] = v.slope[(size_t)g * K + o];
        }
        if (lin && leaf && v.svar[g] >= 0) hl[(size_t)t * MAXN + k] = LinP{v.slope[(size_t)g * K], v.xbar[g], (long long)v.svar[g]};
      }
      // (K-vector leaves: node tables past the tree's last node keep the initial value, as after pgb_create)
      for (int k = nn; k < MAXN && KX > 0; ++k)
        for (int o = 0; o < KX; ++o) hx[((size_t)t * MAXN + k) * KX + o] = d.init_leaf;
    }
    HIPCHK(hipMemcpy(d.trees, T.data(), sizeof(DTree) * (size_t)m, hipMemcpyHostToDevice));
    if (KX > 0) HIPCHK(hipMemcpy(d.tvx, hx.data(), hx.size() * sizeof(double), hipMemcpyHostToDevice));
    if (lin) HIPCHK(hipMemcpy(d.tlin, hl.data(), hl.size() * sizeof(LinP), hipMemcpyHostToDevice));
    if (lin && KX > 0) HIPCHK(hipMemcpy(d.tsx, hsx.data(), hsx.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  // the control word of an idle device at this point of the chain
  Ctrl o = c;
  o.phase = PH_IDLE;
  o.k = 0;
  o.batch_n = 0;
  o.lower = hd.lower;
  o.tune = 0;
  o.round = 0;
  o.steps_left = 0;
  o.pend_leafsd = 0;
  o.pend_iter = 0;
  o.st_cur = st_cur;
  o.alpha_cur = alpha_cur;
  o.cdf_cur = cdf_cur;
  o.iter = hd.iter;
  o.rs_count = hd.rs_count;
  o.leaf_sd = hd.leaf_sd[0];
  o.inv_sigma2 = hd.lik_param[0];
  o.lik_param2 = hd.lik_param[1];
  o.sse0 = 0.0;
  o.u_res = o.u_fin = 0.0;
  o.steps_done = h->steps_target;  // every requested step counts as complete (a poisoned handle was short of it)
  o.done_pub = h->steps_target;
  o.slot_no = h->slot;
  Ctrl both[2] = {o, o};
  HIPCHK(hipMemcpy(d.ctrl, both, sizeof both, hipMemcpyHostToDevice));
  if (KX > 0) {
    double lsdx[2 * KXMAX];
    for (int k = 0; k < KXMAX; ++k) lsdx[k] = lsdx[KXMAX + k] = k < KX ? hd.leaf_sd[k + 1] : h->s.init_leaf_sd;
    HIPCHK(hipMemcpy(d.lsdx, lsdx, sizeof lsdx, hipMemcpyHostToDevice));
  }
  {
    unsigned long long cs[8] = {(unsigned long long)hd.ctr.particle_steps, (unsigned long long)hd.ctr.tree_updates,
                                (unsigned long long)hd.ctr.rows_touched, (unsigned long long)hd.ctr.rounds,
                                (unsigned long long)hd.ctr.saturations, (unsigned long long)hd.ctr.slots,
                                (unsigned long long)hd.ctr.partitions, 0ull};
    HIPCHK(hipMemcpy(d.counters, cs, sizeof cs, hipMemcpyHostToDevice));
  }
  HIPCHK(hipMemset(d.vi, 0, (size_t)p * sizeof(int32_t)));
  HIPCHK(hipDeviceSynchronize());  // (the copies above are blocking calls on the null stream; the sampler's stream is a non-blocking one)
  h->flag[0] = (unsigned long long)h->steps_target;
  h->flag[1] = (unsigned long long)h->steps_target;
  h->flag[2] = (unsigned long long)h->slot;
  h->st_cur = st_cur;
  h->alpha_cur = alpha_cur;
  h->lower_host = hd.lower;
  h->last_lower = hd.last_lower;
  h->last_n = hd.last_n;
  h->inv_sigma2 = hd.lik_param[0];
  h->lik_param2 = hd.lik_param[1];
  h->sigma_dirty = 1;  // (k_begin hands them to the control word again: harmless)
  h->ctr = hd.ctr;
  h->out_valid = 0;  // the mapped block still holds the trees of the step before the load
  h->poisoned = 0;   // an idle image replaces whatever an abandoned step left behind
  return PGB_OK;
}

extern "C" int pgb_profile(pgb_handle* h, int32_t enable, double* kernel_ms_out, int64_t* launches_out) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  JOIN_ASYNC(h);
  if (kernel_ms_out) *kernel_ms_out = h->prof_ms[PK_ROWS];
  if (launches_out) *launches_out = h->prof_launches[PK_ROWS];
  if (enable && !h->prof) {
    for (int k = 0; k < PK_COUNT; ++k) {
      h->prof_ms[k] = 0.0;
      h->prof_launches[k] = 0;
      h->ev_used[k] = 0;
    }
    h->prof_clock_ms = 0.0;
    h->prof_clock_launches = 0;
    h->prof_slot0 = h->slot;
    if (!h->prof_buf) {
      int rc = dalloc(h, &h->prof_buf, (size_t)PROF_RING * PROF_BLOCKS * 2);
      if (rc != PGB_OK) return rc;
    }
    h->d.prof_stamps = h->prof_buf;
    HIPCHK(hipMemsetAsync(h->d.prof_stamps, 0, (size_t)PROF_RING * PROF_BLOCKS * 2 * sizeof(long long), h->stream));
    HIPCHK(hipMemcpyAsync(h->d_dev, &h->d, sizeof(Dev), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  if (!enable && h->prof && h->d.prof_stamps) {  // harvest the device-clock stamps, stop stamping
    const long long n_launch = h->slot - h->prof_slot0;
    if (n_launch > 0 && n_launch <= PROF_RING) {
      std::vector<long long> st((size_t)PROF_RING * PROF_BLOCKS * 2);
      HIPCHK(hipMemcpy(st.data(), h->d.prof_stamps, st.size() * sizeof(long long), hipMemcpyDeviceToHost));
      for (long long sl = h->prof_slot0; sl < h->slot; ++sl) {
        const long long* row = st.data() + (size_t)(sl % PROF_RING) * PROF_BLOCKS * 2;
        long long lo = 0, hi = 0;
        bool any = false;
        for (int b = 0; b < PROF_BLOCKS; ++b) {
          if (row[2 * b] == 0) continue;
          if (!any || row[2 * b] < lo) lo = row[2 * b];
          if (!any || row[2 * b + 1] > hi) hi = row[2 * b + 1];
          any = true;
        }
        if (any) {
          h->prof_clock_ms += (double)(hi - lo) * 1.0e-5;  // 100 MHz ticks -> ms
          h->prof_clock_launches += 1;
        }
      }
    }
    h->d.prof_stamps = nullptr;
    HIPCHK(hipMemcpy(h->d_dev, &h->d, sizeof(Dev), hipMemcpyHostToDevice));
  }
  h->prof = enable ? 1 : 0;
  return PGB_OK;
}

// Device-clock view of the last profiled region: sum over row-pass launches of
// max(last reading of a workgroup) - min(first reading), and the number of launches seen.
extern "C" int pgb_profile_clock(pgb_handle* h, double* kernel_ms_out, int64_t* launches_out) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  JOIN_ASYNC(h);
  if (kernel_ms_out) *kernel_ms_out = h->prof_clock_ms;
  if (launches_out) *launches_out = h->prof_clock_launches;
  return PGB_OK;
}

extern "C" int pgb_profile_kernel(pgb_handle* h, int32_t which, double* kernel_ms_out, int64_t* launches_out,
                                  int32_t* workgroups_out) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  JOIN_ASYNC(h);
  if (which < 0 || which >= PK_COUNT) return fail(PGB_E_INVALID, "unknown kernel");
  if (kernel_ms_out) *kernel_ms_out = h->prof_ms[which];
  if (launches_out) *launches_out = h->prof_launches[which];
  if (workgroups_out) *workgroups_out = h->prof_wgs[which];
  return PGB_OK;
}

#ifdef PGB_TRACE
// raw device-clock stamps of the row pass (first / last reading of every workgroup, per launch)
extern "C" int pgb_debug_stamps(pgb_handle* h, long long* out, long long slot0, int n_slots) {
  if (!h->prof_buf) return fail(PGB_E_INVALID, "profiling was never enabled");
  for (int i = 0; i < n_slots; ++i)
    HIPCHK(hipMemcpy(out + (size_t)i * PROF_BLOCKS * 2, h->prof_buf + (size_t)((slot0 + i) % PROF_RING) * PROF_BLOCKS * 2,
                     (size_t)PROF_BLOCKS * 2 * sizeof(long long), hipMemcpyDeviceToHost));
  return PGB_OK;
}
extern "C" int pgb_debug_trace(pgb_handle* h, long long* out, int n_slots) {
  HIPCHK(hipMemcpy(out, h->d.trace, (size_t)n_slots * TRACE_W * sizeof(long long), hipMemcpyDeviceToHost));
  return PGB_OK;
}
#endif
