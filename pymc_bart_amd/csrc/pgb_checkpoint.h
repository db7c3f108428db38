// pgb_checkpoint.h -- part of pgbart_hip.hip (not a standalone header): checkpoint / resume, profiling and debug entry points.
// ---- checkpoint / resume ------------------------------------------------------------------
// Layout version of the image: bump when a device record that travels in the payload (Job, Ctrl, Cmd, DPart,
// Acc, DTree ...) or this header changes.  The record sizes are stored as well, so an image written by a
// build with other records is refused by name rather than by a payload-size coincidence.
#define PGB_CKPT_VERSION 4
struct CkptHeader {
  char magic[8];       // "PGBCKPT2"
  int32_t version;     // PGB_CKPT_VERSION
  int32_t rec_bytes[7];  // sizeof Job, Ctrl, Cmd, DPart, Acc, DTree, pgb_counters of the writing build
  char backend[16];    // pgb_backend_name()
  pgb_settings s;      // must equal the loading handle's settings
  long long n_allocs, payload_bytes;
  // host mirrors at the idle point
  long long slot, steps_target, flag;
  int32_t st_cur, alpha_cur, lower_host, last_lower, last_n, sigma_dirty;
  double inv_sigma2, lik_param2;
  pgb_counters ctr;
};

static void ckpt_stamp(CkptHeader* hd) {
  memcpy(hd->magic, "PGBCKPT2", 8);
  hd->version = PGB_CKPT_VERSION;
  const int32_t rb[7] = {(int32_t)sizeof(Job), (int32_t)sizeof(Ctrl), (int32_t)sizeof(Cmd), (int32_t)sizeof(DPart),
                         (int32_t)sizeof(Acc), (int32_t)sizeof(DTree), (int32_t)sizeof(pgb_counters)};
  memcpy(hd->rec_bytes, rb, sizeof rb);
}

static long long ckpt_payload(const pgb_handle* h, long long* n_allocs) {
  long long tot = 0, cnt = 0;
  for (size_t i = 0; i < h->allocs.size(); ++i)
    if (h->alloc_persist[i]) {
      tot += (long long)((h->alloc_bytes[i] + 7) & ~(size_t)7);
      cnt += 1;
    }
  if (n_allocs) *n_allocs = cnt;
  return tot;
}

extern "C" int pgb_checkpoint_size(pgb_handle* h, int64_t* bytes_out) {
  if (!h || !bytes_out) return fail(PGB_E_INVALID, "null argument");
  JOIN_ASYNC(h);
  *bytes_out = (int64_t)sizeof(CkptHeader) + ckpt_payload(h, nullptr);
  return PGB_OK;
}

extern "C" int pgb_checkpoint_save(pgb_handle* h, void* host_buf, int64_t bytes) {
  if (!h || !host_buf) return fail(PGB_E_INVALID, "null argument");
  JOIN_ASYNC(h);
  if (!h->have_data || !h->have_y) return fail(PGB_E_INVALID, "set_data/set_response first");
  CkptHeader hd;
  memset(&hd, 0, sizeof hd);
  ckpt_stamp(&hd);
  snprintf(hd.backend, sizeof hd.backend, "%s", pgb_backend_name());
  hd.s = h->s;
  hd.payload_bytes = ckpt_payload(h, &hd.n_allocs);
  if (bytes < (int64_t)sizeof hd + hd.payload_bytes) return fail(PGB_E_INVALID, "checkpoint buffer too small");
  HIPCHK(hipStreamSynchronize(h->stream));  // step calls return idle; this also covers set_* uploads
  hd.slot = h->slot;
  hd.steps_target = h->steps_target;
  hd.flag = (long long)*h->flag;
  hd.st_cur = h->st_cur;
  hd.alpha_cur = h->alpha_cur;
  hd.lower_host = h->lower_host;
  hd.last_lower = h->last_lower;
  hd.last_n = h->last_n;
  hd.sigma_dirty = h->sigma_dirty;
  hd.inv_sigma2 = h->inv_sigma2;
  hd.lik_param2 = h->lik_param2;
  hd.ctr = h->ctr;
  memcpy(host_buf, &hd, sizeof hd);
  char* o = (char*)host_buf + sizeof hd;
  for (size_t i = 0; i < h->allocs.size(); ++i)
    if (h->alloc_persist[i]) {
      HIPCHK(hipMemcpy(o, h->allocs[i], h->alloc_bytes[i], hipMemcpyDeviceToHost));
      o += (h->alloc_bytes[i] + 7) & ~(size_t)7;
    }
  return PGB_OK;
}

extern "C" int pgb_checkpoint_load(pgb_handle* h, const void* host_buf, int64_t bytes) {
  if (!h || !host_buf) return fail(PGB_E_INVALID, "null argument");
  JOIN_ASYNC(h);
  if (!h->have_data || !h->have_y) return fail(PGB_E_INVALID, "set_data/set_response first");
  if (bytes < (int64_t)sizeof(CkptHeader)) return fail(PGB_E_INVALID, "checkpoint truncated");
  CkptHeader hd;
  memcpy(&hd, host_buf, sizeof hd);
  if (memcmp(hd.magic, "PGBCKPT", 7) != 0) return fail(PGB_E_INVALID, "not a pgbart checkpoint");
  CkptHeader mine;
  ckpt_stamp(&mine);
  if (hd.magic[7] != mine.magic[7] || hd.version != mine.version ||
      memcmp(hd.rec_bytes, mine.rec_bytes, sizeof mine.rec_bytes) != 0)
    return fail(PGB_E_INVALID, "checkpoint layout version differs from this build's (written by another release)");
  if (strncmp(hd.backend, pgb_backend_name(), sizeof hd.backend) != 0)
    return fail(PGB_E_INVALID, "checkpoint was written by a different backend");
  if (memcmp(&hd.s, &h->s, sizeof(pgb_settings)) != 0)
    return fail(PGB_E_INVALID, "checkpoint settings differ from this sampler's settings");
  long long n_allocs = 0;
  const long long payload = ckpt_payload(h, &n_allocs);
  if (hd.n_allocs != n_allocs || hd.payload_bytes != payload || bytes < (int64_t)sizeof hd + payload)
    return fail(PGB_E_INVALID, "checkpoint layout does not match this build");
  HIPCHK(hipStreamSynchronize(h->stream));
  const char* o = (const char*)host_buf + sizeof hd;
  for (size_t i = 0; i < h->allocs.size(); ++i)
    if (h->alloc_persist[i]) {
      HIPCHK(hipMemcpy(h->allocs[i], o, h->alloc_bytes[i], hipMemcpyHostToDevice));
      o += (h->alloc_bytes[i] + 7) & ~(size_t)7;
    }
  h->slot = hd.slot;
  h->steps_target = hd.steps_target;
  h->flag[0] = (unsigned long long)hd.flag;
  h->flag[1] = (unsigned long long)hd.flag;  // an idle image: every recorded step is complete
  h->flag[2] = (unsigned long long)hd.slot;  // ... and every enqueued slot has run
  h->st_cur = hd.st_cur;
  h->alpha_cur = hd.alpha_cur;
  h->lower_host = hd.lower_host;
  h->last_lower = hd.last_lower;
  h->last_n = hd.last_n;
  h->sigma_dirty = hd.sigma_dirty;
  h->inv_sigma2 = hd.inv_sigma2;
  h->lik_param2 = hd.lik_param2;
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
      h->alloc_persist.back() = 0;
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
