// pgb_dev_types.h -- part of pgbart_hip.hip (not a standalone header): device-side records of the slot state machine (nodes, jobs, statistics, the Dev argument block).
// ------------------------------------------------------------------ device structs
struct DNode {  // 64 bytes
  double split, value, sse;
  long long q_st, q_r, q_r2;
  int32_t cnt;
  int32_t var;     // -1 leaf
  int32_t cc_row;  // chunk-count row of this node (-1: root => all rows)
  uint8_t left, right, depth, label;
};

struct DTree {  // an accepted tree
  int32_t n_nodes, n_leaves;
  int32_t pad[2];
  DNode nd[MAXN];
};

struct DPart {  // a particle
  int32_t n_nodes, n_leaves, next_pop;
  int32_t loc_gen, loc_slot;  // where its leaf labels live (slot -1: implicit root labels)
  int32_t pad;
  double sse_tot, sse_orph;
  DNode nd[MAXN];
};

struct Job {  // one particle's work for a PARTITION row pass + what the next k_ctrl needs (128 B)
  int32_t active;
  int32_t copy;  // no split, but the labels must be copied forward (their generation is next to be reused)
  int32_t src_gen, src_slot;
  int32_t node, label, new_label;
  int32_t var, rule, check_nan;
  int32_t ccL, ccR;
  int32_t cnt;
  int32_t vkey;  // 16-bit order key of the split value (shadow instances of the row pass: XK16)
  double v;
  // statistics of the node being split (the parent of the children the pass creates)
  long long p_q_st, p_q_r, p_q_r2;
  double p_sse, p_value;
  // particle header after this round's pop (so that the next k_ctrl needs one load per particle)
  double h_sse_tot, h_sse_orph;
  int32_t h_n_nodes, h_n_leaves, h_next_pop, p_depth;
  int32_t popped;  // a node was popped for this particle in the round this record proposes (a particle step)
  // leaf noise of the two children this split creates ([U] draw_leaf_value: the Box-Muller pair addressed
  // by (iter, round, particle)): drawn by an idle wave of the slot that WRITES the job, so that the slot
  // that finishes the round does not start with ~1.8 us of Philox + log + sqrt + sincos on its one
  // critical wave
  double z0, z1;
};
// (No unions / arrays in records that are copied by value in kernels: they defeat scalar
//  replacement and the copies get demoted to LDS or scratch.)

struct JobL {  // non-Normal families: fixed-point log-likelihoods that travel with a Job
  long long p_ll, h_ll_tot, h_ll_orph, pad;
};
struct AccL {  // non-Normal families (k_loglik): log-likelihood of the left / right children and of
  long long llL, llR, llN, pad;  // the rows dropped by a missing split value
};

struct Acc {  // statistics of the LEFT child and of the NaN-dropped rows (right = parent - both)
  unsigned long long cnts;  // cntL | cntN << 32
  long long aL, bL, c2L, aN, bN, c2N;
  long long pad;
};

#define IA_SLOTS 8 /* the row pass spreads its atomics over this many cache lines */
// Per-particle split statistics: every work item of a particle adds to the particle's record, and
// atomics on ONE cache line serialise (~12 ns each: 98 chunks x 4 values = 4.7 us at cfg2).  The
// record is therefore kept ACC_SLOTS times (item -> slot by chunk); readers sum the copies.
#ifndef ACC_SLOTS
#define ACC_SLOTS 4 /* measured at cfg2: 1 -> 1.52 M, 4 -> 1.62 M, 8 -> 1.58 M, 16 -> 1.51 M particle-steps/s */
#endif
#ifndef ACC_STRIDE
#define ACC_STRIDE 2 /* distance between copies, in records of 64 B: one 128-B line each */
#endif
#define ACC_PER (ACC_SLOTS * ACC_STRIDE)
#define PROF_RING 4096   /* row-pass launches a profiled region may span */
#define PROF_BLOCKS 1024 /* = the largest row grid */
// the same for the log-likelihood sums of k_loglik (32-byte records): 8 copies, 128 B apart
#define LL_SLOTS 8
#define LL_STRIDE 4
#define LL_PER (LL_SLOTS * LL_STRIDE)
// ... and for the extension-output sums of the multi-output row pass: 4 copies of 32 longs (aL[k], aN[k] of the
// outputs 1..K-1: 2 (PGB_MAX_OUTPUTS - 1) = 30 values)
#define AX_SLOTS 4
#define AX_REC 32
#define AX_PER (AX_SLOTS * AX_REC)
struct InitAcc {   // one 64-byte line
  long long A, B, C, E0, QSTD;
  long long pad0, pad1, pad2;
};

enum { CMD_NOOP = 0, CMD_PARTITION = 1, CMD_INIT = 2, CMD_FINAL = 4 /* FINAL|INIT = 6 */ };
enum { PH_IDLE = 0, PH_BEGIN = 1, PH_ROUND = 2 };

// linear response: what a leaf adds to its constant value: slope * (x[svar] - xbar); svar < 0: nothing
struct LinP {
  double slope, xbar;
  long long svar;
};
// linear response: sums of u = x 2^-ex over the left / right child of a split (see pgb_lin_fit):
// q_u, q_uu, q_us, q_ur each
struct AccU {
  long long uL[4], uR[4];
};

struct Cmd {
  int32_t kind;
  int32_t tree_old, tree_new;
  int32_t sel_gen, sel_slot;  // sel_slot == -2: the old tree was kept
  int32_t tune, dst_gen, st_cur;
  long long rs_count;
  double lv_new[256], lv_next[256];
  // label -> value table of the tree being updated as it stands (used instead of lv_new when the update
  // keeps the old tree, sel_slot == -2): built ahead by an idle wave, see k_ctrl (single-output, constant leaves)
  double lv_keep[256];
};

struct Ctrl {
  int32_t phase, k, batch_n, lower;
  int32_t tune, round, lid_gen, steps_left;
  int32_t pend_leafsd, st_cur;  // st_cur: which sum_trees buffer is current
  int32_t alpha_cur, cdf_cur;   // current buffers of the split weights / their prefix sums
  long long iter, rs_count, pend_iter;
  double leaf_sd, inv_sigma2;  // (leaf_sd of outputs 1..K-1: Dev::lsdx -- no arrays in this record,
                               //  the compiler would demote a by-value copy with an indexed array to LDS)
  double lik_param2;  // second scalar parameter of the two-parameter likelihood families
  double sse0;  // SSE of the reference particle (the current tree), fixed at round 0
  long long steps_done;  // asteps completed since creation (mirrored to the host flag)
  long long slot_no;     // k_ctrl launches so far
  // draws the NEXT slot needs before it can do anything else, made one slot ahead (same addresses):
  // the systematic-resampling offset of the round just proposed, the final-choice draw of its tree
  double u_res, u_fin;
  // steps whose LAST row pass has run, as published to the host (host_flag[1]) by the first idle slot after
  // the step: the host then fetches the step's results without waiting for the idle slots queued behind it
  long long done_pub;
};

// Global address space.  A pointer LOADED from memory (every array of the argument block) is a generic pointer
// to the compiler, and a load through it is a FLAT instruction: it counts in lgkmcnt as well as vmcnt, so every
// wait for an LDS read (`s_waitcnt lgkmcnt(0)`) also drains the global loads and stores in flight.  The
// kernels therefore read the block through `DevG`, the same layout with its pointers typed as global
// (address space 1) in the DEVICE pass -- loads become global_load (vmcnt only, SGPR base + VGPR offset),
// atomics global_atomic -- while the host fills in `Dev` with ordinary pointers.  (In the host pass of the
// translation unit the alias is a plain pointer, so that kernel bodies type-check there too.)
#if defined(__HIP_DEVICE_COMPILE__)
template <typename T>
using gptr = T __attribute__((address_space(1)))*;
#else
template <typename T>
using gptr = T*;
#endif
template <bool G, typename T>
using dptr = typename std::conditional<G, gptr<T>, T*>::type;

// children of one job, one extension output (what ChildX carries for a constant leaf), handed from the likelihood pass
// of a slot to the control kernel of the next (Dev::finx)
struct FinX {
  double vL, vR;
  long long aL, aR;
};

template <bool G>
struct DevT {  // kernel argument block (by value)
  long long n, n_pad;
  int32_t p, m, P, nchunks;
  int32_t batch_tune, batch_draw;
  int32_t family, K;  // K = n_outputs; KX = K - 1 extension outputs live in the *x arrays below
  int32_t rows_target, rows_target_init;  // work items the row passes aim for (tuning knobs)
  int32_t ll_target;                      // ... and the log-likelihood pass
  int32_t compat;                         // PGB_COMPAT_* (pgb_settings.compat): upstream-semantics switches
  unsigned long long seed;
  double init_leaf, mdouble;
  pgb_scales sc;
  dptr<G, const double> prior_leaf;  // [PGB_MAX_DEPTH] device copy
  dptr<G, const double> XT;  // [p][n_pad]
  dptr<G, const uint16_t> XK16; // [p][n_pad] 16-bit order keys of XT (null unless the matrix is larger than the Infinity Cache)
  dptr<G, const double> y;   // [n_pad]
  dptr<G, const double> off; // [K][n_pad] offset of the linear predictor (per-row families; 0 by default)
  dptr<G, double> st;        // [2][n_pad] sum_trees (ping-pong, see k_rows)
  dptr<G, double2> pack;     // [n_pad] {sum_trees, y - noi}
  dptr<G, double> rs_mean;
  dptr<G, double> rs_m2;
  dptr<G, uint8_t> tree_lid;  // [m][n_pad]
  dptr<G, uint8_t> lid;       // [NGEN][MAXP][n_pad]
  dptr<G, uint16_t> cc;       // [CC_ROUNDS*MAXP*2][nchunks], or [..][cc_stride] for the instances with order keys (see cc_stride)
  dptr<G, DTree> trees;       // [m]
  dptr<G, DPart> parts;       // [2][P]
  dptr<G, Job> jobs;          // [2][P]
  dptr<G, Acc> acc;           // [2][P][ACC_SLOTS]
  dptr<G, AccL> accl;         // [2][P][LL_SLOTS]   (non-Normal families)
  dptr<G, JobL> jobl;         // [2][P]   (non-Normal families)
  dptr<G, InitAcc> initacc;   // [2][IA_SLOTS]
  dptr<G, Cmd> cmd;           // [2]
  dptr<G, Ctrl> ctrl;         // [2]
  dptr<G, unsigned long long> counters;  // particle_steps, tree_updates, rows_touched, rounds, sat, slots
  dptr<G, int32_t> vi;        // [p]
  dptr<G, long long> alpha;   // [2][p] integer split weights (pgb_alpha_init + counts * alpha_unit)
  dptr<G, long long> cdfS;    // [2][p] their prefix sums, as used by the sampler
  long long alpha_unit;
  double max_prior;
  dptr<G, const int32_t> rules;
  dptr<G, const int32_t> col_nan;
  // ---- K-vector leaves (K > 1): output 0 uses the scalar fields, outputs 1..K-1 these arrays
  dptr<G, double> packx;      // [KX][n_pad]            sum_trees of outputs 1.. (as of INIT, like pack.x)
  dptr<G, double> pvx;        // [2][MAXP][MAXN][KX]    particle leaf values
  dptr<G, long long> pqx;     // [2][MAXP][MAXN][KX]    particle node sums of sum_trees
  dptr<G, double> tvx;        // [m][MAXN][KX]          accepted trees' leaf values
  dptr<G, long long> accx;    // [2][MAXP][AX_SLOTS][AX_REC]  row-pass statistics: aL[k], aN[k] (copies, see AX_SLOTS)
  dptr<G, long long> iax;     // [2][IA_SLOTS][2*KX]    INIT/FINAL statistics: A[k], QSTD[k]
  dptr<G, double> lvx;        // [2][2][256][KX]        label->value tables: [par][0 new | 1 next]
  dptr<G, long long> jqx;     // [2][MAXP][KX]          per job: parent's node sums
  dptr<G, double> jvx;        // [2][MAXP][KX]          per job: parent's leaf values
  dptr<G, double> jzx;        // [2][MAXP][KX][2]       per job: leaf noise of the children, outputs 1..K-1 (drawn one slot ahead like Job::z0 / z1)
  dptr<G, double> lsdx;       // [2][KXMAX]             leaf_sd of outputs 1..K-1 (double-buffered like ctrl)
  // constant K-vector leaves: the children's values / sums of outputs 1..K-1 of every job of a slot, as workgroup 0
  // of the slot's likelihood pass derived them for its job list (child_values_x) -- the control kernel of the NEXT slot
  // needs exactly these and used to derive them again from the same statistics (two dependent round trips of
  // scattered loads per extension wave in front of its ancestor barrier); now one contiguous read
  dptr<G, FinX> finx;         // [2][MAXP][KX]
  // softmax with constant leaves: the ROW part of the factorised log-likelihood (pgbart_spec.h, pgb_loglik_cat_f),
  // written by the likelihood pass of the slot that starts a tree, read by the passes of that tree's later rounds
  dptr<G, double> cat_e;      // [K][n_pad]  E_k = exp(eta_k - max_k eta_k), eta_k = sum_trees_noi_k + offset_k
  dptr<G, double> cat_a;      // [n_pad]     eta_c - max_k eta_k of the observed class c
  dptr<G, uint8_t> cat_c;     // [n_pad]     the observed class c
  // ---- linear response (Normal family, K = 1, continuous columns)
  int32_t response, has_off;  // has_off: an offset of the linear predictor is set (else the array is all 0)
  double lin_R, inv_R;
  dptr<G, const int32_t> col_ex;  // [p] exponent bound of every column
  dptr<G, LinP> plin;             // [2][MAXP][MAXN]  particle leaves
  dptr<G, LinP> tlin;             // [m][MAXN]        accepted trees' leaves
  dptr<G, LinP> lvl;              // [2][2][256]      label -> LinP tables: [par][0 new | 1 next]
  dptr<G, AccU> accu;             // [2][MAXP][ACC_PER]  row-pass sums (copies like acc)
  // ... K-vector leaves: one slope per output on the shared regressor; output 0 lives in LinP,
  // outputs 1..K-1 in arrays laid out like pvx / tvx / lvx / accx
  dptr<G, double> psx;            // [2][MAXP][MAXN][KX]  particle leaf slopes
  dptr<G, double> tsx;            // [m][MAXN][KX]        accepted trees' leaf slopes
  dptr<G, double> lsx;            // [2][2][256][KX]      label -> slope tables: [par][0 new | 1 next]
  dptr<G, long long> accux;       // [2][MAXP][AX_SLOTS][AX_REC]  row-pass sums of u st_k: left [k], right [KX + k]
  // callback family only (null otherwise): what k_loglik hands to the host instead of evaluating it --
  // per particle and row, the linear predictor and the side (0 left, 1 right, 2 dropped, 3 not in the leaf)
  dptr<G, double> cb_mu;      // [MAXP][n_pad]
  dptr<G, uint8_t> cb_side;   // [MAXP][n_pad]
  // profiling only (null otherwise): [PROF_RING][PROF_BLOCKS][2] device-clock stamps of the row pass
  dptr<G, long long> prof_stamps;
  dptr<G, unsigned long long> host_flag;  // pinned host word: number of completed asteps
  dptr<G, long long> trace;               // PGB_TRACE builds only: [TRACE_SLOTS][TRACE_W] wall_clock64 stamps
  // Row stride of `cc` for the instances of the data sets with order keys (k_ctrl<.., KEYS>, k_rows<.., F32>,
  // k_rows_mk<.., F32>): nchunks rounded up to 8 counts, so that a lane's sixteen counts are two aligned 16-byte loads
  // (select_split_row<V16>).  Every other instance strides by nchunks, as before.  (At the END of the block: the
  // instances that do not use it keep every offset -- and with it their register allocation -- as it was.)
  int32_t cc_stride, pad_cc;
};
typedef DevT<false> Dev;   // as the host fills it in and the kernels receive it
typedef DevT<true> DevG;   // as the kernels read it (see above)
static_assert(sizeof(Dev) == sizeof(DevG), "the two views of the argument block must coincide");

#ifdef PGB_TRACE
#define TRACE_SLOTS 4096
#define TRACE_W 40 /* stamps 0..15, 16: attempt, 17: round of the proposal, 18: fresh (this slot starts a tree), 19: stop, 20..23: inside the split-row selection, 24..31: the likelihood pass (TRL) */
// The record of a slot is addressed through `tr_rec` (set once per kernel from the slot number the kernel
// has in registers anyway): a stamp is one clock read and one store, no load.  TR0() keeps the entry
// reading in a register until the slot number is known.
#define TR_DECL() long long tr_t0 = 0; long long* tr_rec = nullptr
#define TR0() do { tr_t0 = wall_clock64(); } while (0)
#define TR_BIND(slot_no) do { tr_rec = S.trace + (size_t)((slot_no) % TRACE_SLOTS) * TRACE_W; if (blockIdx.x == 1 && threadIdx.x == 0) tr_rec[0] = tr_t0; } while (0)
#define TR(i) do { if (blockIdx.x == 1 && threadIdx.x == 0) tr_rec[(i)] = wall_clock64(); } while (0)
#define TRX(i, cond) do { if (cond) tr_rec[(i)] = wall_clock64(); } while (0)
#define TRV(i, val) do { if (blockIdx.x == 1 && threadIdx.x == 0) tr_rec[(i)] = (val); } while (0)
#define TRS(i) do { if (tr_rec != nullptr && blockIdx.x == 1 && threadIdx.x == 0) tr_rec[(i)] = wall_clock64(); } while (0)
#define TR_REC tr_rec
// stamps of the row pass (entries 12..15 of the slot's record), taken by one chosen workgroup
#define TRR_BIND(slot_no)                                                                   \
  const long long tr_t12 = wall_clock64();                                                   \
  long long* tr_rec = S.trace + (size_t)((slot_no) % TRACE_SLOTS) * TRACE_W;                 \
  if (blockIdx.x == 0 && threadIdx.x == 0) tr_rec[12] = tr_t12
#define TRR(i, blk) do { if (blockIdx.x == (blk) && threadIdx.x == 0) tr_rec[(i)] = wall_clock64(); } while (0)
// stamps of the likelihood pass (entries 24..31), taken by workgroup TRL_WG (one in the middle of the grid: it has
// no INIT unit beyond its share and a full span of the passes)
#define TRL_WG 100
#define TRL_BIND(slot_no) long long* trl_rec = S.trace + (size_t)((slot_no) % TRACE_SLOTS) * TRACE_W
#define TRL(i) do { if (blockIdx.x == TRL_WG && threadIdx.x == 0) trl_rec[(i)] = wall_clock64(); } while (0)
#else
#define TR_DECL() ((void)0)
#define TR0() ((void)0)
#define TR_BIND(slot_no) ((void)0)
#define TR(i) ((void)0)
#define TRX(i, cond) ((void)0)
#define TRV(i, val) ((void)0)
#define TRS(i) ((void)0)
#define TR_REC nullptr
#define TRR_BIND(slot_no) ((void)0)
#define TRR(i, blk) ((void)0)
#define TRL_BIND(slot_no) ((void)0)
#define TRL(i) ((void)0)
#endif
#ifdef PGB_STAMP_LL
#define PGB_STAMP_LL_ON 1
#else
#define PGB_STAMP_LL_ON 0
#endif

// A read of a wave-uniform, kernel-invariant record (written by an EARLIER launch) through the
// constant address space: the compiler can then use scalar (SMEM) loads and keep the record in
// SGPRs instead of issuing per-lane flat loads.
#define PGB_CONST_AS __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ T load_uniform(const T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  T out;
  __builtin_memcpy(&out, (const PGB_CONST_AS void*)(unsigned long long)p, sizeof(T));
  return out;
#else
  return *p;
#endif
}

