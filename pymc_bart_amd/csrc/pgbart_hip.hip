// pgbart_hip.hip -- particle-Gibbs BART for MI355X (gfx950 / CDNA4), C ABI of include/pgbart.h.
//
// Replaces the native sampler the reference imports from the external `bartrs` wheel
// (pymc_bart/pymc_bart.py:2; call site tests/test_bart.py:231-235).  Algorithm: SURVEY.md
// Appendix A (upstream PGBART.astep) under the numeric contract of include/pgbart_spec.h.
//
// Design:
//   * The whole astep is a DEVICE-RESIDENT STATE MACHINE.  The host only enqueues identical
//     "slots" = { k_ctrl ; k_rows [; k_loglik] } on one HIP stream and polls a pinned progress
//     word; it never waits for the device inside a tree update.  k_ctrl (one 256-thread
//     workgroup per particle) finishes the previous SMC round from the integer statistics the
//     row pass produced -- leaf values, particle weights, systematic resampling on one wave, the
//     next growth proposal incl. the exact "k-th row of the leaf" selection -- and writes one
//     job per particle.  k_rows streams the rows on a persistent grid: a work item is a 1024-row
//     chunk times a group of particles with work; it relabels the rows of the leaves being split
//     and reduces the children's sufficient statistics (four values per butterfly, wave_sum4).
//     The slot that starts a tree fuses FINAL(previous tree) + INIT + round 0 into one pass.
//   * HBM layout: X column-major (coalesced column streams), {sum_trees, residual} packed as
//     double2 per row, one BYTE leaf label per row per particle in an 8-generation ring (idle
//     particles are never copied; a pass writes only particles that split).
//   * All row reductions are integer (fixed point) => bit-reproducible, independent of
//     launch geometry and atomics order, and identical to the CPU oracle.
//   * Kernel instances are compiled per data set where the hot loop has no registers or
//     instructions to spare: k_rows<SubsetSplit columns?, Normal family?, linear response?>,
//     k_rows_mk<K>, k_loglik<K, family>, k_ctrl<multi-output?, linear response?>.
//   * No MFMA: the path is gather / partition / reduce (HBM / L2 bound).
//
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pgbart.h"
#include "pgbart_spec.h"

#define CH 1024 /* rows per chunk = rows per k_rows workgroup */
#define BT 256  /* threads per workgroup */
#define RPT (CH / BT)
#define MAXN PGB_MAX_NODES
#define MAXP PGB_MAX_PARTICLES
#define CC_ROUNDS 256
#define NGEN 8 /* generations of particle leaf labels (ring) */

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
  double v;
  // statistics of the node being split (the parent of the children the pass creates)
  long long p_q_st, p_q_r, p_q_r2;
  double p_sse, p_value;
  // particle header after this round's pop (so that the next k_ctrl needs one load per particle)
  double h_sse_tot, h_sse_orph;
  int32_t h_n_nodes, h_n_leaves, h_next_pop, p_depth;
  int32_t pad_;
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
// ... and for the extension-output sums of the multi-output row pass: 4 copies of 16 longs (128 B)
#define AX_SLOTS 4
#define AX_REC 16
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
};

struct Dev {  // kernel argument block (by value)
  long long n, n_pad;
  int32_t p, m, P, nchunks;
  int32_t batch_tune, batch_draw;
  int32_t family, K;  // K = n_outputs; KX = K - 1 extension outputs live in the *x arrays below
  int32_t rows_target, rows_target_init;  // work items the row passes aim for (tuning knobs)
  int32_t ll_target, ll_pad;              // ... and the log-likelihood pass
  unsigned long long seed;
  double init_leaf, mdouble;
  pgb_scales sc;
  const double* prior_leaf;  // [PGB_MAX_DEPTH] device copy
  const double* XT;  // [p][n_pad]
  const double* y;   // [n_pad]
  const double* off; // [n_pad] offset of the linear predictor (single-output per-row families; 0 by default)
  double* st;        // [2][n_pad] sum_trees (ping-pong, see k_rows)
  double2* pack;     // [n_pad] {sum_trees, y - noi}
  double* rs_mean;
  double* rs_m2;
  uint8_t* tree_lid;  // [m][n_pad]
  uint8_t* lid;       // [NGEN][MAXP][n_pad]
  uint16_t* cc;       // [CC_ROUNDS*MAXP*2][nchunks]
  DTree* trees;       // [m]
  DPart* parts;       // [2][P]
  Job* jobs;          // [2][P]
  Acc* acc;           // [2][P][ACC_SLOTS]
  AccL* accl;         // [2][P][LL_SLOTS]   (non-Normal families)
  JobL* jobl;         // [2][P]   (non-Normal families)
  InitAcc* initacc;   // [2][IA_SLOTS]
  Cmd* cmd;           // [2]
  Ctrl* ctrl;         // [2]
  unsigned long long* counters;  // particle_steps, tree_updates, rows_touched, rounds, sat, slots
  int32_t* vi;        // [p]
  long long* alpha;   // [2][p] integer split weights (pgb_alpha_init + counts * alpha_unit)
  long long* cdfS;    // [2][p] their prefix sums, as used by the sampler
  long long alpha_unit;
  double max_prior;
  const int32_t* rules;
  const int32_t* col_nan;
  // ---- K-vector leaves (K > 1): output 0 uses the scalar fields, outputs 1..K-1 these arrays
  double* packx;      // [KX][n_pad]            sum_trees of outputs 1.. (as of INIT, like pack.x)
  double* pvx;        // [2][MAXP][MAXN][KX]    particle leaf values
  long long* pqx;     // [2][MAXP][MAXN][KX]    particle node sums of sum_trees
  double* tvx;        // [m][MAXN][KX]          accepted trees' leaf values
  long long* accx;    // [2][MAXP][AX_SLOTS][AX_REC]  row-pass statistics: aL[k], aN[k] (copies, see AX_SLOTS)
  long long* iax;     // [2][IA_SLOTS][2*KX]    INIT/FINAL statistics: A[k], QSTD[k]
  double* lvx;        // [2][2][256][KX]        label->value tables: [par][0 new | 1 next]
  long long* jqx;     // [2][MAXP][KX]          per job: parent's node sums
  double* jvx;        // [2][MAXP][KX]          per job: parent's leaf values
  double* lsdx;       // [2][KXMAX]             leaf_sd of outputs 1..K-1 (double-buffered like ctrl)
  // ---- linear response (Normal family, K = 1, continuous columns)
  int32_t response, has_off;  // has_off: an offset of the linear predictor is set (else the array is all 0)
  double lin_R, inv_R;
  const int32_t* col_ex;  // [p] exponent bound of every column
  LinP* plin;             // [2][MAXP][MAXN]  particle leaves
  LinP* tlin;             // [m][MAXN]        accepted trees' leaves
  LinP* lvl;              // [2][2][256]      label -> LinP tables: [par][0 new | 1 next]
  AccU* accu;             // [2][MAXP][ACC_PER]  row-pass sums (copies like acc)
  // ... K-vector leaves: one slope per output on the shared regressor; output 0 lives in LinP,
  // outputs 1..K-1 in arrays laid out like pvx / tvx / lvx / accx
  double* psx;            // [2][MAXP][MAXN][KX]  particle leaf slopes
  double* tsx;            // [m][MAXN][KX]        accepted trees' leaf slopes
  double* lsx;            // [2][2][256][KX]      label -> slope tables: [par][0 new | 1 next]
  long long* accux;       // [2][MAXP][AX_SLOTS][AX_REC]  row-pass sums of u st_k: left [k], right [KX + k]
  // profiling only (null otherwise): [PROF_RING][PROF_BLOCKS][2] device-clock stamps of the row pass
  long long* prof_stamps;
  unsigned long long* host_flag;  // pinned host word: number of completed asteps
  long long* trace;               // PGB_TRACE builds only: [TRACE_SLOTS][16] wall_clock64 stamps
};

#ifdef PGB_TRACE
#define TRACE_SLOTS 4096
#define TR(i)                                                                              \
  do {                                                                                     \
    if (blockIdx.x == 1 && threadIdx.x == 0)                                               \
      S.trace[(size_t)(S.ctrl[par].slot_no % TRACE_SLOTS) * 16 + (i)] = wall_clock64();         \
  } while (0)
#define TRX(i, cond)                                                                       \
  do {                                                                                     \
    if (cond) S.trace[(size_t)(S.ctrl[par].slot_no % TRACE_SLOTS) * 16 + (i)] = wall_clock64(); \
  } while (0)
// stamps of the row pass (entries 12..15 of the slot's record), taken by one chosen workgroup
#define TRR(i, blk)                                                                                  \
  do {                                                                                               \
    if (blockIdx.x == (blk) && threadIdx.x == 0)                                                     \
      S.trace[(size_t)((S.ctrl[par ^ 1].slot_no - 1) % TRACE_SLOTS) * 16 + (i)] = wall_clock64();    \
  } while (0)
#else
#define TR(i) ((void)0)
#define TRX(i, cond) ((void)0)
#define TRR(i, blk) ((void)0)
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

// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ long long wave_sum(long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// 64-bit wave sum with DPP row shifts/broadcasts (gfx9 DPP); the result lands in lane 63.
__device__ __forceinline__ long long wave_sum_dpp(long long v) {
  int lo = (int)v, hi = (int)(v >> 32);
#define PGB_DPP_STEP(ctrl, rm)                                                   \
  {                                                                              \
    int tl = __builtin_amdgcn_update_dpp(0, lo, ctrl, rm, 0xf, 0);               \
    int th = __builtin_amdgcn_update_dpp(0, hi, ctrl, rm, 0xf, 0);               \
    long long a = ((long long)hi << 32) | (unsigned)lo;                          \
    long long b = ((long long)th << 32) | (unsigned)tl;                          \
    a += b;                                                                      \
    lo = (int)a;                                                                 \
    hi = (int)(a >> 32);                                                         \
  }
  PGB_DPP_STEP(0x111, 0xf)  // row_shr:1
  PGB_DPP_STEP(0x112, 0xf)  // row_shr:2
  PGB_DPP_STEP(0x114, 0xf)  // row_shr:4
  PGB_DPP_STEP(0x118, 0xf)  // row_shr:8
  PGB_DPP_STEP(0x142, 0xa)  // row_bcast:15
  PGB_DPP_STEP(0x143, 0xc)  // row_bcast:31
#undef PGB_DPP_STEP
  return ((long long)hi << 32) | (unsigned)lo;
}

// Wave-wide sums of FOUR 64-bit values at once ("transposed" butterfly): the first two exchange
// steps halve the number of live values instead of carrying all four through every step, so the
// whole reduction costs ~42 VALU instructions instead of 4 x 24.  Integer adds: any order gives
// the same bits.  Exchanges: quad_perm (xor 1, xor 2), masked row shifts (xor 4), row_ror:8
// (xor 8) and the gfx950 v_permlane16_swap / v_permlane32_swap (xor 16, xor 32).
// Returns, in EVERY lane, the wave total of value number (lane & 3).
template <int CTRL>
__device__ __forceinline__ long long dpp_mov64(long long x) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, 0);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(x >> 32), CTRL, 0xf, 0xf, 0);
  return ((long long)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ int dpp_xor4(int x) {
  int t = __builtin_amdgcn_update_dpp(x, x, 0x114, 0xf, 0xa, 0);  // lanes 4-7, 12-15 <- lane - 4
  return __builtin_amdgcn_update_dpp(t, x, 0x104, 0xf, 0x5, 0);   // lanes 0-3, 8-11  <- lane + 4
}
__device__ __forceinline__ long long wave_sum4(long long v0, long long v1, long long v2, long long v3) {
  const int lane = (int)(threadIdx.x & 63);
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
  // xor 1: even lanes keep (v0, v2), odd lanes keep (v1, v3)
  long long k0 = b0 ? v1 : v0, k1 = b0 ? v3 : v2;
  const long long s0 = b0 ? v0 : v1, s1 = b0 ? v2 : v3;
  k0 += dpp_mov64<0xB1>(s0);  // quad_perm [1,0,3,2]
  k1 += dpp_mov64<0xB1>(s1);
  // xor 2: lanes with bit 1 clear keep the first (v0 | v1), the others the second (v2 | v3)
  long long k = b1 ? k1 : k0;
  const long long s = b1 ? k0 : k1;
  k += dpp_mov64<0x4E>(s);  // quad_perm [2,3,0,1]
  // from here on lane l carries value (l & 3)
  {
    const int lo = dpp_xor4((int)k), hi = dpp_xor4((int)(k >> 32));
    k += ((long long)hi << 32) | (unsigned)lo;
  }
  k += dpp_mov64<0x128>(k);  // row_ror:8
  {
    const auto l = __builtin_amdgcn_permlane16_swap((unsigned)k, (unsigned)k, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((unsigned)(k >> 32), (unsigned)(k >> 32), false, false);
    k = (long long)(((unsigned long long)h[0] << 32) | l[0]) + (long long)(((unsigned long long)h[1] << 32) | l[1]);
  }
  {
    const auto l = __builtin_amdgcn_permlane32_swap((unsigned)k, (unsigned)k, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((unsigned)(k >> 32), (unsigned)(k >> 32), false, false);
    k = (long long)(((unsigned long long)h[0] << 32) | l[0]) + (long long)(((unsigned long long)h[1] << 32) | l[1]);
  }
  return k;
}

// sum of the ACC_SLOTS copies of a particle's split statistics
__device__ __forceinline__ Acc load_acc(const Acc* __restrict__ base) {
  Acc a = base[0];
#pragma unroll
  for (int k = 1; k < ACC_SLOTS; ++k) {
    const Acc t = base[k * ACC_STRIDE];
    a.cnts += t.cnts;
    a.aL += t.aL; a.bL += t.bL; a.c2L += t.c2L;
    a.aN += t.aN; a.bN += t.bN; a.c2N += t.c2N;
  }
  return a;
}

// extension-output statistic `idx` (aL[k]: k, aN[k]: KX + k) of a particle, summed over its copies
__device__ __forceinline__ long long load_accx(const long long* __restrict__ accx, int par, int q, int idx) {
  const long long* b = accx + ((size_t)par * MAXP + q) * AX_PER + idx;
  long long s = 0;
#pragma unroll
  for (int k = 0; k < AX_SLOTS; ++k) s += b[k * AX_REC];
  return s;
}

// block-wide sum of NV long long values; result valid in thread 0
template <int NV>
__device__ __forceinline__ void block_sum(long long (&v)[NV], long long* sm /* [NV*4] */) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = wave_sum_dpp(v[i]);
    if (lane == 63) sm[i * 4 + w] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = sm[i * 4] + sm[i * 4 + 1] + sm[i * 4 + 2] + sm[i * 4 + 3];
  }
  __syncthreads();
}

// block-wide exclusive scan of one int per thread (256 threads); returns exclusive prefix,
// total via *tot (all threads)
__device__ __forceinline__ int block_excl_scan(int x, int* sm /* [8] */, int* tot) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = x;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) sm[w] = inc;
  __syncthreads();
  int base = 0;
  for (int i = 0; i < w; ++i) base += sm[i];
  *tot = sm[0] + sm[1] + sm[2] + sm[3];
  __syncthreads();
  return base + inc - x;
}

__device__ __forceinline__ bool go_left(int rule, double x, double v) {
  return pgb_go_left(rule, x, v) != 0;
}
// the two-rule form for data without SubsetSplit columns (the Normal-family row pass is compiled
// both ways: it has no registers to spare for the set-membership test)
template <bool SUB>
__device__ __forceinline__ bool go_left_t(int rule, double x, double v) {
  if (SUB) return pgb_go_left(rule, x, v) != 0;
  return rule == PGB_RULE_CONTINUOUS ? (x <= v) : (x == v);
}

// label -> leaf value table of a node array (ORPHAN and unused labels -> 0)
__device__ __forceinline__ void build_lv(const DNode* nd, int n_nodes, double* lv /*[256] global*/) {
  for (int i = threadIdx.x; i < 256; i += BT) lv[i] = 0.0;
  __syncthreads();
  for (int i = threadIdx.x; i < n_nodes; i += BT)
    if (nd[i].var < 0) lv[nd[i].label] = nd[i].value;
  __syncthreads();
}
// linear response: label -> linear part of the leaf
__device__ __forceinline__ void build_lvl(const DNode* nd, int n_nodes, const LinP* lin, LinP* lv /*[256] global*/) {
  for (int i = threadIdx.x; i < 256; i += BT) lv[i] = LinP{0.0, 0.0, -1};
  __syncthreads();
  for (int i = threadIdx.x; i < n_nodes; i += BT)
    if (nd[i].var < 0) lv[nd[i].label] = lin[i];
  __syncthreads();
}


__device__ __forceinline__ double readlane_d(double v, int lane /* wave-uniform */) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// max over the wave (order-free), DPP reduction to lane 63 + broadcast
__device__ __forceinline__ double wave_max_d(double v) {
#define PGB_MAX_STEP(ctrl, rm)                                                        \
  {                                                                                   \
    int lo = __double2loint(v), hi = __double2hiint(v);                               \
    int tl = __builtin_amdgcn_update_dpp(lo, lo, ctrl, rm, 0xf, 0);                   \
    int th = __builtin_amdgcn_update_dpp(hi, hi, ctrl, rm, 0xf, 0);                   \
    double t = __hiloint2double(th, tl);                                              \
    v = t > v ? t : v;                                                                \
  }
  PGB_MAX_STEP(0x111, 0xf)
  PGB_MAX_STEP(0x112, 0xf)
  PGB_MAX_STEP(0x114, 0xf)
  PGB_MAX_STEP(0x118, 0xf)
  PGB_MAX_STEP(0x142, 0xa)
  PGB_MAX_STEP(0x143, 0xc)
#undef PGB_MAX_STEP
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// inclusive wave scan of one int per lane (DPP row shifts + row broadcasts)
__device__ __forceinline__ int wave_incl_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, 0);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, 0);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, 0);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, 0);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, 0);  // row_bcast:15
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, 0);  // row_bcast:31
  return x;
}

// the same for the extension outputs: lvx[label][k] from a node array + its [node][KX] values
__device__ __forceinline__ void build_lvx(const DNode* nd, int n_nodes, const double* vx, int KX,
                                          double* lvx /*[256][KX] global*/) {
  for (int i = threadIdx.x; i < 256 * KX; i += BT) lvx[i] = 0.0;
  __syncthreads();
  for (int e = threadIdx.x; e < n_nodes * KX; e += BT) {
    const int i = e / KX, k = e % KX;
    if (nd[i].var < 0) lvx[(size_t)nd[i].label * KX + k] = vx[e];
  }
  __syncthreads();
}

// ------------------------------------------------------------------ k_begin
__global__ void k_begin(const Dev* __restrict__ Sp, int par, int tune, int n_steps, double inv_sigma2, double lik_param2,
                        int set_sigma) {
  const Dev& S = *Sp;
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
__device__ __forceinline__ ChildVals child_values(const Dev& S, int rule, int cnt, long long p_q_st,
                                                  double p_value, unsigned long long a_cnts, long long a_aL,
                                                  long long a_aN, double z0, double z1, double leaf_sd) {
  ChildVals c;
  c.cL = (int)(a_cnts & 0xFFFFFFFFull);
  c.cN = (int)(a_cnts >> 32);
  c.cR = cnt - c.cL - c.cN;
  c.aL = a_aL;
  c.aR = p_q_st - a_aL - a_aN;
  if (rule != PGB_RULE_CONTINUOUS && c.cR == 0) {
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
__device__ __forceinline__ double leaf_sd_x(const Dev& S, const Ctrl& c, int ctrl_par, int acc_par, int k) {
  if (!(c.pend_leafsd && c.pend_iter > 2)) return S.lsdx[ctrl_par * KXMAX + k];
  const int KX = S.K - 1;
  long long q = 0;
  for (int sl = 0; sl < IA_SLOTS; ++sl) q += S.iax[((size_t)acc_par * IA_SLOTS + sl) * 2 * KX + KX + k];
  return ((double)q * S.sc.inv_c1) / (double)S.n;
}
__device__ __forceinline__ long long root_A_x(const Dev& S, int acc_par, int k) {
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
__device__ __forceinline__ ChildX child_values_x(const Dev& S, int ok, int cL, int cR, long long aLk,
                                                 long long aNk, long long pq, double pv, uint32_t it,
                                                 uint32_t round, uint32_t particle, int k, double lsd) {
  ChildX c;
  c.sL = c.sR = 0.0;
  c.aL = aLk;
  c.aR = pq - aLk - aNk;
  if (ok == 1) {
    const pgb_u2 u = pgb_draw2(S.seed, it, round, particle, PGB_RNG_LEAF, (uint32_t)(k + 1));
    double z0, z1;
    pgb_normal2(u.u0, u.u1, &z0, &z1);
    c.vL = pgb_leaf_value(cL, c.aL, S.sc.inv_c1, S.mdouble, z0, lsd);
    c.vR = pgb_leaf_value(cR, c.aR, S.sc.inv_c1, S.mdouble, z1, lsd);
  } else {
    c.vL = pv;  // failed one-hot split: the leaf keeps its value
    c.vR = 0.0;
  }
  return c;
}

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
__device__ __forceinline__ int wave_pick(double lw, int first, int cnt, double u) {
  const int lane = threadIdx.x & 63;
  const bool act = lane >= first && lane < first + cnt;
  const double mx = wave_max_d(act ? lw : -1.0e308);
  double W = act ? pgb_exp(lw - mx) + 1e-12 : 0.0;
#define PGB_SCAN_STEP(ctrl, rm)                                                       \
  {                                                                                   \
    const int tl = __builtin_amdgcn_update_dpp(0, __double2loint(W), ctrl, rm, 0xf, 0); \
    const int th = __builtin_amdgcn_update_dpp(0, __double2hiint(W), ctrl, rm, 0xf, 0); \
    W = W + __hiloint2double(th, tl);                                                 \
  }
  PGB_SCAN_STEP(0x111, 0xf)  // row_shr:1
  PGB_SCAN_STEP(0x112, 0xf)  // row_shr:2
  PGB_SCAN_STEP(0x114, 0xf)  // row_shr:4
  PGB_SCAN_STEP(0x118, 0xf)  // row_shr:8
  PGB_SCAN_STEP(0x142, 0xa)  // row_bcast:15 -> rows 1, 3
  PGB_SCAN_STEP(0x143, 0xc)  // row_bcast:31 -> rows 2, 3
#undef PGB_SCAN_STEP
  const int last = first + cnt - 1;
  const double thr = u * readlane_d(W, last);
  const bool hit = act && (lane < last) && !(thr > W);
  const unsigned long long m = __ballot(hit);
  return m ? (int)__ffsll((long long)m) - 1 : last;
}

// [U] SampleSplittingVariable.rvs on one wave, from stored prefix sums (pgb_sample_var)
__device__ __forceinline__ int sample_var_prefix(const long long* Sarr, int p, double u) {
  const int lane = threadIdx.x & 63;
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
__device__ __forceinline__ LinKids lin_children(const Dev& S, const AccU* __restrict__ copies, int var, int cL, int cR,
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
__device__ __forceinline__ void lin_children_x(const Dev& S, LinKids& lk, ChildX& cx, int var, int cL, int cR,
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

// MK: K-vector leaves (K > 1).  The single-output instantiation contains none of that code.
template <bool MK, bool LIN>
__global__ __launch_bounds__(BT) __attribute__((amdgpu_waves_per_eu(1, 1)))  // latency kernel: registers, not occupancy
void k_ctrl(const Dev* __restrict__ Sp, int par, Ctrl* __restrict__ ctrls, const InitAcc* __restrict__ ias) {
  // ctrls / ias repeat S.ctrl / S.initacc as kernel arguments (see k_rows)
  const Dev& S = *Sp;  // device-resident: kernel arguments live in host-coherent memory, HBM is closer
  __shared__ Fin s_fin[MAXP];
  __shared__ int s_i[16];
  __shared__ double s_d[4];
  __shared__ double s_pre[2][PGB_SELECT_TRIES + 2];  // [set][0: coin, 1+t: row draw of try t]
  __shared__ double s_pre1[2][PGB_SELECT_TRIES + 2]; // second uniform of the same draws (subset masks)
  __shared__ ChildX s_finx[MK ? MAXP : 1][KXMAX];     // K-vector leaves: children, outputs 1..K-1
  __shared__ double s_prior[PGB_MAX_DEPTH];           // P(leaf | depth): read once by the idle wave 3
  __shared__ DNode s_pop[MAXP];                       // node each OLD particle would pop next (prefetched)

  TR(0);
  if (threadIdx.x >= BT - 64) s_prior[threadIdx.x - (BT - 64)] = S.prior_leaf[threadIdx.x - (BT - 64)];
  const Ctrl c = load_uniform(&ctrls[par]);
  Ctrl* co = &ctrls[par ^ 1];
  const int b = blockIdx.x, p = b + 1, tid = threadIdx.x;
  const int P = S.P, Lc = P - 1;
  Cmd* cmd = &S.cmd[par];
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
  if (c.pend_leafsd && c.pend_iter > 2) leaf_sd = ((double)ia.QSTD * S.sc.inv_c1) / (double)S.n;

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
      *co = o;
      cmd->kind = CMD_NOOP;
    }
    return;
  }

  const bool begin = c.phase == PH_BEGIN;  // first tree of a step: nothing to finish
  const bool normal = S.family == PGB_FAMILY_NORMAL;
  TR(1);
  const int r = c.round;  // >= 1 in PH_ROUND: round 0 is proposed by the slot that starts the tree
  const uint32_t it = (uint32_t)c.iter;
  const DPart* OT = S.parts + (size_t)par * MAXP;
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
  if (tid >= 64 && tid < 192) {
    const int set = (tid >> 6) - 1, l = tid & 63;
    const pgb_u2 u = pgb_draw2(S.seed, set ? it + 1u : it, set ? 0u : (uint32_t)r, (uint32_t)p,
                               l == 0 ? PGB_RNG_PROPOSE : PGB_RNG_SELECT, l == 0 ? 0u : (uint32_t)(l - 1));
    if (l <= PGB_SELECT_TRIES) {
      s_pre[set][l] = u.u0;
      s_pre1[set][l] = u.u1;
    }
    const double u1 = readlane_d(u.u1, 0);
    const int jj = (set && rebuild) ? sample_var_weights(alpha, S.p, u1) : sample_var_prefix(cdfS, S.p, u1);
    if (l == 0) s_i[8 + set] = jj;
    TRX(9 + set, blockIdx.x == 1 && l == 0);
  }

  int anc = p;  // ancestor (old particle index) of new particle p
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
      const int q = tid;
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
        j = JP[q];
        a = load_acc(&S.acc[((size_t)(par ^ 1) * MAXP + q) * ACC_PER]);
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
      // one Philox evaluation per lane: lane 0 draws the resampling offset, lane q the leaf noise
      double z0, z1, u_res;
      {
        const pgb_u2 ul = pgb_draw2(S.seed, it, (uint32_t)(r - 1), (uint32_t)q,
                                    q == 0 ? PGB_RNG_RESAMPLE : PGB_RNG_LEAF, 0);
        u_res = readlane_d(ul.u0, 0);
        pgb_normal2(ul.u0, ul.u1, &z0, &z1);
      }
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
        if constexpr (MK) {
          if (j.active)
          for (int k = 0; k < KX; ++k) {
            const long long pq = r1 ? root_A_x(S, par ^ 1, k) : S.jqx[((size_t)(par ^ 1) * MAXP + q) * KX + k];
            const double pv = r1 ? S.init_leaf : S.jvx[((size_t)(par ^ 1) * MAXP + q) * KX + k];
            s_finx[q][k] = child_values_x(S, f.ok, f.cL, f.cR, load_accx(S.accx, par ^ 1, q, k),
                                          load_accx(S.accx, par ^ 1, q, KX + k), pq, pv, it,
                                          (uint32_t)(r - 1), (uint32_t)q, k, leaf_sd_x(S, c, par, par ^ 1, k));
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
        s_pop[q] = popn;
        pending = f.next_pop < f.n_nodes;
        lw = normal ? (f.sse_tot + f.sse_orph) * (-0.5 * c.inv_sigma2)
                    : (double)(f.ll_tot + f.ll_orph) * S.sc.inv_cl;
      }
      TR(2);
      stop = __ballot(pending) == 0ull;
      int pick;
      if (!stop) {
        // [U] systematic resampling of particles 1..P-1: ancestor of new particle p
        const double ui = (u_res + (double)(p - 1)) / (double)Lc;
        pick = wave_pick(lw, 1, Lc, ui);
      } else {
        // [U] get_particle_tree: final choice among all P particles (lane 0 = reference particle)
        if (q == 0) lw = normal ? sse0 * (-0.5 * c.inv_sigma2) : sse0;
        const pgb_u2 u_fin = pgb_draw2(S.seed, it, 0, 0, PGB_RNG_FINAL, 0);
        pick = wave_pick(lw, 0, P, u_fin.u0);
      }
      if (tid == 0) {
        s_i[0] = stop ? 1 : 0;
        s_i[1] = pick;
      }
    }
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
      for (int i = tid; i < nn; i += BT) {
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
        for (int e = tid; e < nn * KX; e += BT) {
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
          for (int e = tid; e < 2 * KX; e += BT) {
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
        build_lv(S.trees[tree_old].nd, S.trees[tree_old].n_nodes, cmd->lv_new);
      }
      // label table of the next tree to update (a different tree unless m == 1)
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
        if (tid == 0)
          for (int i = 0; i < nn; ++i)
            if (snd[i].var >= 0) alpha_o[snd[i].var] += S.alpha_unit;
      } else {
        if (tid == 0)
          for (int i = 0; i < nn; ++i)
            if (snd[i].var >= 0) S.vi[snd[i].var] += 1;
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
      atomicAdd(&S.counters[0], 1ull);
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
    // Everything below runs on wave 0 only (no workgroup barriers): the k-th row (ascending) of
    // the leaf, k = floor(u * cnt)   ([U] get_split_value)
    if (tid < 64) {
      const int j = s_i[8 + set];
      const double* xc = S.XT + (size_t)j * S.n_pad;
      const bool subset_rule = S.rules[j] == PGB_RULE_SUBSET;
      const uint8_t* lid =
          job.src_slot >= 0 ? S.lid + ((size_t)job.src_gen * MAXP + job.src_slot) * S.n_pad : nullptr;
      const uint16_t* ccr = ncc >= 0 ? S.cc + (size_t)ncc * S.nchunks : nullptr;
      int found = 0;
      double v = 0.0;
      TR(6);
      // per-lane partial sums of the node's per-chunk row counts (independent of the retry)
      const int per = (S.nchunks + 63) / 64;
      const int c0 = lane_id() * per;
      int c1 = c0 + per;
      if (c1 > S.nchunks) c1 = S.nchunks;
      int part = 0, pre = 0;
      if (lid != nullptr) {
        for (int cc = c0; cc < c1; ++cc) part += ccr[cc];
        pre = wave_incl_scan(part) - part;
      }
      for (uint32_t tr = 0; tr < PGB_SELECT_TRIES && !found; ++tr) {
        long long k = (long long)(s_pre[set][1 + tr] * (double)ncnt);
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
            while (kk >= ccr[cstar]) {
              kk -= ccr[cstar];
              ++cstar;
            }
          }
          const int ol = (int)__ffsll((long long)__ballot(own)) - 1;
          cstar = __builtin_amdgcn_readlane(cstar, ol);
          kk = __builtin_amdgcn_readlane(kk, ol);
          // (2) which row inside the chunk: 16 label bytes per lane
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
        if (found && subset_rule) v = pgb_subset_value(s_pre1[set][1 + tr], x);
      }
      if (tid == 0) {
        s_i[0] = found;
        s_d[0] = v;
      }
    }
    __syncthreads();
    if (s_i[0]) {
      const int j = s_i[8 + set];
      job.active = 1;
      job.node = node;
      job.label = nlabel;
      job.new_label = F.n_leaves;
      job.var = j;
      job.rule = S.rules[j];
      job.check_nan = S.col_nan[j];
      job.ccL = ((rr * MAXP + p) * 2);
      job.ccR = job.ccL + 1;
      job.cnt = ncnt;
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
      atomicAdd(&S.counters[2], (unsigned long long)nd.cnt);
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
      }
    }
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
#ifdef PGB_TRACE
  if (b == 1 && tid == 0) {
    S.trace[(size_t)(c.slot_no % TRACE_SLOTS) * 16 + 15] = r;
    S.trace[(size_t)(c.slot_no % TRACE_SLOTS) * 16 + 14] = attempt;
    TR(8);
  }
#endif
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
};

// LIN: linear response (Normal family only): leaves predict value + slope (x[svar] - xbar); the
// partition additionally reduces the sums pgb_lin_fit needs for both children.
template <bool SUB, bool NORMAL, bool LIN>
__global__ __launch_bounds__(BT, (NORMAL && !LIN) ? 3 : 2) void k_rows(const Dev* __restrict__ Sp, int par,
                                                              const Cmd* __restrict__ cmds,
                                                              const Job* __restrict__ jobs_all) {
  // cmds / jobs_all repeat S.cmd / S.jobs as kernel arguments: their first loads then do not wait
  // for the load of the argument block S itself (one dependent memory round trip less)
  const Dev& S = *Sp;
  constexpr int NRED = LIN ? 15 : 7;  // values reduced per particle
  __shared__ long long s_red[MAXP * NRED * 4];
  __shared__ double s_lv[2][256];
  __shared__ LinP s_ll[LIN ? 2 : 1][LIN ? 256 : 1];  // label -> linear part: [0 new | 1 next]
  __shared__ RJob s_job[MAXP];
  __shared__ int s_n[2];
  const Cmd* cmd = &cmds[par];
  const int kind = cmd->kind;
  TRR(12, 0);
  // profiling: every workgroup leaves its first and last device-clock reading; the host takes
  // min(start) .. max(end) per launch -- the interval rocprofv3 reports for the dispatch
  long long* pstamp = nullptr;
  if (S.prof_stamps != nullptr && threadIdx.x == 0 && blockIdx.x < PROF_BLOCKS) {
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
      s_lv[0][i] = cmd->lv_new[i];
      s_lv[1][i] = cmd->lv_next[i];
      if constexpr (LIN) {
        s_ll[0][i] = S.lvl[((size_t)par * 2 + 0) * 256 + i];
        s_ll[1][i] = S.lvl[((size_t)par * 2 + 1) * 256 + i];
      }
    }
  }
  uint8_t* tl_old = do_final ? S.tree_lid + (size_t)cmd->tree_old * S.n_pad : nullptr;
  const uint8_t* tl_new = do_init ? S.tree_lid + (size_t)cmd->tree_new * S.n_pad : nullptr;
  const uint8_t* sel_lid =
      (do_final && cmd->sel_slot >= 0) ? S.lid + ((size_t)cmd->sel_gen * MAXP + cmd->sel_slot) * S.n_pad : nullptr;
  const double cntf = (double)cmd->rs_count;
  // sum_trees buffers: an INIT reads st_in and writes sum_trees_noi to st_out (other workgroups of
  // the same chunk still read st_in); a lone FINAL updates st_in in place
  double* const st_in = S.st + (size_t)cmd->st_cur * S.n_pad;
  double* const st_out = S.st + (size_t)(do_init ? cmd->st_cur ^ 1 : cmd->st_cur) * S.n_pad;

  if (do_part) {
    const Job* jobs = jobs_all + (size_t)par * MAXP;
    // list of particles with work in this pass (split or forced label refresh); their job
    // fields are cached in LDS once per workgroup
    if (tid < 64) {
      Job j;
      j.active = 0;
      j.copy = 0;
      if (tid >= 1 && tid < S.P) j = jobs[tid];  // one round trip: the whole job
      const bool has = (j.active | j.copy) != 0;
      const unsigned long long m = __ballot(has);
      if (has) {
        const int k = __popcll(m & ((1ull << tid) - 1ull));
        RJob rj;
        rj.p = tid;
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
        rj.uscale = 1.0;
        if constexpr (LIN) rj.uscale = j.active ? pgb_pow2(-S.col_ex[j.var]) : 1.0;
        s_job[k] = rj;
      }
      if (tid == 0) s_n[0] = __popcll(m);
    }
    __syncthreads();
    TRR(13, 0);
    const int nact = s_n[0];
    if (nact == 0 && !do_init) { PROF_END(); return; }
    const int target = do_init ? S.rows_target_init : S.rows_target;
    int G = (nact * S.nchunks + target - 1) / target;
    if (G < 1) G = 1;
    int ngroups = (nact + G - 1) / G;
    if (ngroups < 1) ngroups = 1;  // an INIT must run even if no particle splits
    const int nitems = S.nchunks * ngroups;
    uint8_t* __restrict__ const dst0 = S.lid + (size_t)cmd->dst_gen * MAXP * S.n_pad;
    const uint8_t* __restrict__ const lid0 = S.lid;
    const double* __restrict__ const XT = S.XT;
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
        uint32_t ids_next = *(const uint32_t*)(tl_new + base);
        uint32_t ids_sel = 0;
        if (do_final) {
          if (cmd->sel_slot == -2) {
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
          if (cmd->tree_new == cmd->tree_old) ids_next = ids_sel;
        }
        // every input of the thread's four rows is requested BEFORE the first result is stored:
        // the stores below may alias the loads as far as the compiler knows, so loads left inside
        // the loop would be issued one row (one memory round trip) at a time
        double st4[RPT], y4[RPT], mean4[RPT], m24[RPT];
        {
          const double2* __restrict__ sp = (const double2*)(st_in + base);
          const double2* __restrict__ yp = (const double2*)(S.y + base);
          const double2 s01 = sp[0], s23 = sp[1], y01 = yp[0], y23 = yp[1];
          st4[0] = s01.x; st4[1] = s01.y; st4[2] = s23.x; st4[3] = s23.y;
          y4[0] = y01.x; y4[1] = y01.y; y4[2] = y23.x; y4[3] = y23.y;
          const bool upd = do_final && cmd->tune && writer;
#pragma unroll
          for (int e = 0; e < RPT; ++e) {
            mean4[e] = upd ? S.rs_mean[base + e] : 0.0;
            m24[e] = upd ? S.rs_m2[base + e] : 0.0;
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
              S.rs_mean[row] = mean;
              S.rs_m2[row] = m2;
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
          qb[e] = pgb_quant(r, c1, &sat1);
          qc[e] = pgb_quant(r * r, c2, &sat1);
          strow[e] = st;
          rrow[e] = r;
          if (writer) {  // saturation is counted where the values are produced, once
            S.pack[row] = make_double2(st, r);
            st_out[row] = noi;
            sat += sat1;
            iv[0] += qa[e];
            iv[1] += qb[e];
            if (normal) {
              iv[2] += qc[e];
              const double er = r - o;
              iv[3] += pgb_quant(er * er, c2, &sat);
            } else {
              // C: log-likelihood of a fresh stump, E0: of the current tree (reference particle)
              const double lp = S.ctrl[par ^ 1].inv_sigma2, lp2 = S.ctrl[par ^ 1].lik_param2;  // family parameters
              const double offv = S.has_off ? S.off[row] : 0.0;  // (x + 0.0 == x bit for bit)
              iv[2] += pgb_quant(pgb_loglik1q(S.family, yv, (noi + offv) + S.init_leaf, lp, lp2, pgb_ln_tn(), pgb_ln_tp()), S.sc.cl, &sat);
              iv[3] += pgb_quant(pgb_loglik1q(S.family, yv, st + offv, lp, lp2, pgb_ln_tn(), pgb_ln_tp()), S.sc.cl, &sat);
            }
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < RPT; ++e) {
          const double2 sr = S.pack[base + e];
          strow[e] = sr.x;
          rrow[e] = sr.y;
          qa[e] = pgb_quant(sr.x, c1, nullptr);
          qb[e] = pgb_quant(sr.y, c1, nullptr);
          qc[e] = pgb_quant(sr.y * sr.y, c2, nullptr);
        }
      }
      TRR(14, 0);  // rows of the item loaded and quantised (INIT part done)
      uint32_t root_ids = 0;
#pragma unroll
      for (int e = 0; e < RPT; ++e)
        if (base + e >= n) root_ids |= (uint32_t)PGB_ORPHAN << (8 * e);
      const int g0 = grp * G, g1 = (g0 + G < nact) ? g0 + G : nact;
      // software pipeline over the particles of the group: the labels and split-column values of
      // particle g + 1 are requested before particle g is relabelled and reduced
      uint32_t nx_ids = root_ids;
      double2 nx0 = {0.0, 0.0}, nx1 = {0.0, 0.0};
      if (g0 < g1) {
        const RJob& rn = s_job[g0];
        if (rn.src >= 0) nx_ids = *(const uint32_t*)(lid0 + rn.src + base);
        if (rn.active) {
          const double2* __restrict__ xn = (const double2*)(XT + rn.xoff + base);
          nx0 = xn[0];
          nx1 = xn[1];
        }
      }
      for (int g = g0; g < g1; ++g) {
        const RJob& rj = s_job[g];
        const uint32_t ids = nx_ids;
        const double2 t0 = nx0, t1 = nx1;
        if (g + 1 < g1) {
          const RJob& rn = s_job[g + 1];
          nx_ids = rn.src < 0 ? root_ids : *(const uint32_t*)(lid0 + rn.src + base);
          if (rn.active) {
            const double2* __restrict__ xn = (const double2*)(XT + rn.xoff + base);
            nx0 = xn[0];
            nx1 = xn[1];
          }
        }
        uint32_t out = ids;
        uint8_t* __restrict__ const dp = dst0 + (size_t)rj.p * n_pad + base;
        if (!rj.active) {  // forced refresh only
          *(uint32_t*)dp = out;
          continue;
        }
        const double x[RPT] = {t0.x, t0.y, t1.x, t1.y};
        const int slot = (g - g0) * NRED;
        if (!rj.check_nan) {  // common case: the split column has no missing values
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
          const long long tot = wave_sum4(v0, v1, v2, v3);  // lane l: total of value l & 3
          if (lane < 4) s_red[(slot + lane) * 4 + w] = tot;
        } else {
          long long v[7] = {0, 0, 0, 0, 0, 0, 0};  // cnts(L | R<<20 | N<<40), aL, bL, c2L, aN, bN, c2N
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
          if (lane < 4) s_red[(slot + lane) * 4 + w] = ta;
          else if (lane < 7) s_red[(slot + lane) * 4 + w] = tb;  // lane 4..6: value (lane & 3) of the second set
        }
        if constexpr (LIN) {  // sums of u = x 2^-ex over the two children (see pgb_lin_fit)
          long long ul[4] = {0, 0, 0, 0}, ur[4] = {0, 0, 0, 0};
#pragma unroll
          for (int e = 0; e < RPT; ++e) {
            const double xv = x[e];
            if (((ids >> (8 * e)) & 255u) == (uint32_t)rj.label && xv == xv) {
              const double uu = xv * rj.uscale;
              const long long q0 = pgb_quant(uu * S.lin_R, c1, nullptr);
              const long long q1 = pgb_quant((uu * uu) * S.lin_R, c1, nullptr);
              const long long q2 = pgb_quant(uu * strow[e], c1, nullptr);
              const long long q3 = pgb_quant(uu * rrow[e], c1, nullptr);
              const bool gl = go_left_t<SUB>(rj.rule, xv, rj.v);
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
        const long long s = s_red[t * 4] + s_red[t * 4 + 1] + s_red[t * 4 + 2] + s_red[t * 4 + 3];
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
          S.cc[(size_t)rj.ccL * S.nchunks + chunk] = (uint16_t)cL;
          S.cc[(size_t)rj.ccR * S.nchunks + chunk] = (uint16_t)cR;
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
      const double mean0 = S.rs_mean[row], m20 = S.rs_m2[row];
      const double delta = nv - mean0;
      const double mean = mean0 + delta / cntf;
      const double delta2 = nv - mean;
      const double m2 = m20 + delta * delta2;
      S.rs_mean[row] = mean;
      S.rs_m2[row] = m2;
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

// ------------------------------------------------------------------ k_rows_mk
// K-vector leaves (K > 1, Categorical-softmax): the same slot logic as k_rows with sum_trees, leaf
// values and running-sd statistics per output.  Output 0 uses the scalar buffers, outputs 1..K-1
// the *x extension arrays.  Not the headline path: written for clarity, K loops innermost.
__device__ __forceinline__ double loglik_any(const Dev& S, double y, const double* mu) {
  return pgb_loglik(S.family, S.K, y, mu);
}

// KT: number of outputs when known at compile time (2, 3, 4: loops unroll, the per-row arrays stay
// in registers), 0: any K <= PGB_MAX_OUTPUTS.
// LIN: linear response; the label -> (slope, xbar, column) tables are read from global memory
// (lvl for output 0 and the shared parts, lsx for the slopes of outputs 1..K-1)
template <int KT, bool LIN>
__global__ __launch_bounds__(BT) void k_rows_mk(const Dev* __restrict__ Sp, int par) {
  const Dev& S = *Sp;
  const int K = KT > 0 ? KT : S.K, KX = K - 1;
  constexpr int KB = KT > 0 ? KT : PGB_MAX_OUTPUTS;  // compile-time bound of the K loops
  __shared__ long long s_red[MAXP * (1 + 2 * KB) * 4];
  __shared__ double s_lv[2][256][KB];
  __shared__ RJob s_job[MAXP];
  __shared__ int s_n[2];
  const Cmd* cmd = &S.cmd[par];
  const int kind = cmd->kind;
  if (kind == CMD_NOOP) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const bool do_final = (kind & CMD_FINAL) != 0, do_init = (kind & CMD_INIT) != 0;
  const bool do_part = (kind & CMD_PARTITION) != 0;
  const int NV = 1 + 2 * K;  // per particle: counts, aL[K], aN[K]

  if (do_final || do_init) {
    const double* lx = S.lvx + (size_t)par * 2 * 256 * KX;
    for (int i = tid; i < 256; i += BT) {
      s_lv[0][i][0] = cmd->lv_new[i];
      s_lv[1][i][0] = cmd->lv_next[i];
      for (int k = 0; k < KX; ++k) {
        s_lv[0][i][k + 1] = lx[(size_t)i * KX + k];
        s_lv[1][i][k + 1] = lx[(size_t)256 * KX + (size_t)i * KX + k];
      }
    }
  }
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
      Job j;
      j.active = 0;
      j.copy = 0;
      if (tid >= 1 && tid < S.P) j = jobs[tid];
      const bool has = (j.active | j.copy) != 0;
      const unsigned long long m = __ballot(has);
      if (has) {
        const int k = __popcll(m & ((1ull << tid) - 1ull));
        RJob rj;
        rj.p = tid;
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
      if (tid == 0) s_n[0] = __popcll(m);
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
    long long iv[2 + 2 * PGB_MAX_OUTPUTS];  // C, E0, A[K], QSTD[K]
    for (int i = 0; i < 2 + 2 * K; ++i) iv[i] = 0;
    unsigned sat = 0;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
      const int chunk = item % S.nchunks, grp = item / S.nchunks;
      const long long base = (long long)chunk * CH + tid * RPT;
      double stv[RPT][KB];  // sum_trees of this thread's rows, per output
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
            if (writer) *(uint32_t*)(tl_old + base) = ids_sel;
          }
          if (cmd->tree_new == cmd->tree_old) ids_next = ids_sel;
        }
        for (int e = 0; e < RPT; ++e) {
          const long long row = base + e;
          for (int k = 0; k < K; ++k) stv[e][k] = 0.0;
          if (row >= n) continue;
          double mu_stump[KB], mu_cur[KB];
          for (int k = 0; k < K; ++k) {
            double st = st_in[(size_t)k * n_pad + row];
            if (do_final) {
              const double nv = lin_pred(s_lv[0][(ids_sel >> (8 * e)) & 255u][k], 0, (ids_sel >> (8 * e)) & 255u, k, row);
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
                iv[2 + K + k] += pgb_quant(PGB_SQRT(m2 / cntf), c1, &sat);
              }
            }
            const double o = lin_pred(s_lv[1][(ids_next >> (8 * e)) & 255u][k], 1, (ids_next >> (8 * e)) & 255u, k, row);
            const double noi = st - o;
            stv[e][k] = st;
            mu_stump[k] = noi + S.init_leaf;
            mu_cur[k] = st;
            if (writer) {
              if (k == 0) S.pack[row] = make_double2(st, 0.0);
              else S.packx[(size_t)(k - 1) * n_pad + row] = st;
              st_out[(size_t)k * n_pad + row] = noi;
              iv[2 + k] += pgb_quant(st, c1, &sat);
            }
          }
          if (writer) {
            const double yv = S.y[row];
            iv[0] += pgb_quant(pgb_loglik(S.family, K, yv, mu_stump), S.sc.cl, &sat);  // C: fresh stump
            iv[1] += pgb_quant(pgb_loglik(S.family, K, yv, mu_cur), S.sc.cl, &sat);    // E0: current tree
          }
        }
      } else {
        for (int e = 0; e < RPT; ++e) {
          stv[e][0] = S.pack[base + e].x;
          for (int k = 1; k < K; ++k) stv[e][k] = S.packx[(size_t)(k - 1) * n_pad + base + e];
        }
      }
      uint32_t root_ids = 0;
      for (int e = 0; e < RPT; ++e)
        if (base + e >= n) root_ids |= (uint32_t)PGB_ORPHAN << (8 * e);
      long long qst[RPT][KB];  // quantised once per row, reused by every particle of the group
#pragma unroll
      for (int e = 0; e < RPT; ++e)
#pragma unroll
        for (int k = 0; k < KB; ++k) qst[e][k] = k < K ? pgb_quant(stv[e][k], c1, nullptr) : 0;
      const int g0 = grp * G, g1 = (g0 + G < nact) ? g0 + G : nact;
      for (int g = g0; g < g1; ++g) {
        const RJob& rj = s_job[g];
        const uint32_t ids = rj.src < 0 ? root_ids : *(const uint32_t*)(S.lid + rj.src + base);
        uint32_t out = ids;
        uint8_t* const dp = dst0 + (size_t)rj.p * n_pad + base;
        if (!rj.active) {
          *(uint32_t*)dp = out;
          continue;
        }
        const double2* xp = (const double2*)(S.XT + rj.xoff + base);
        const double2 t0 = xp[0], t1 = xp[1];
        const double x[RPT] = {t0.x, t0.y, t1.x, t1.y};
        int side[RPT];  // 0: not in the leaf, 1: left, 2: right, 3: dropped (missing value)
        long long cnts = 0;
        for (int e = 0; e < RPT; ++e) {
          side[e] = 0;
          if (((ids >> (8 * e)) & 255u) == (uint32_t)rj.label) {
            const double xv = x[e];
            if (xv != xv) {
              side[e] = 3;
              out = (out & ~(255u << (8 * e))) | ((uint32_t)PGB_ORPHAN << (8 * e));
              cnts += 1ll << 40;
            } else if (go_left(rj.rule, xv, rj.v)) {
              side[e] = 1;
              cnts += 1;
            } else {
              side[e] = 2;
              out = (out & ~(255u << (8 * e))) | ((uint32_t)rj.new_label << (8 * e));
              cnts += 1ll << 20;
            }
          }
        }
        *(uint32_t*)dp = out;
        const int slot = (g - g0) * NV;
        // values of this particle: [0] counts, [1 + k] aL[k], [1 + K + k] aN[k]; reduced four at a
        // time (wave_sum4); a column without missing values has no aN part
        long long vals[1 + 2 * KB + 3];
#pragma unroll
        for (int i = 0; i < 1 + 2 * KB + 3; ++i) vals[i] = 0;
        vals[0] = cnts;
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          if (k < K) {
            long long aL = 0, aN = 0;
#pragma unroll
            for (int e = 0; e < RPT; ++e) {
              const long long q = qst[e][k];
              aL += side[e] == 1 ? q : 0;
              aN += side[e] == 3 ? q : 0;
            }
            vals[1 + k] = aL;
            vals[1 + K + k] = aN;
          }
        }
        if constexpr (LIN) {  // sums of u = x 2^-ex over the two children (see pgb_lin_fit): u, u^2 and
          // u st_k per output; one wave total each, added by lane 63 (a rare path: no LDS staging)
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
              for (int k = 0; k < KB; ++k)
                if (k < K) su[sd][2 + k] += pgb_quant(uu * stv[e][k], c1, nullptr);
            }
          }
          AccU* au = &S.accu[((size_t)par * MAXP + rj.p) * ACC_PER + (chunk & (ACC_SLOTS - 1)) * ACC_STRIDE];
          long long* aux = S.accux + ((size_t)par * MAXP + rj.p) * AX_PER + (size_t)(chunk & (AX_SLOTS - 1)) * AX_REC;
#pragma unroll
          for (int sd = 0; sd < 2; ++sd)
#pragma unroll
            for (int i = 0; i < 2 + KB; ++i) {
              if (i >= 2 + K) continue;
              const long long tot = wave_sum_dpp(su[sd][i]);
              if (lane == 63 && tot != 0) {
                long long* dst = i < 3 ? (sd ? &au->uR[i] : &au->uL[i]) : &aux[(sd ? KX : 0) + (i - 3)];
                atomicAdd((unsigned long long*)dst, (unsigned long long)tot);
              }
            }
        }
        const int nv = rj.check_nan ? NV : 1 + K;
#pragma unroll
        for (int c4 = 0; c4 < (1 + 2 * KB + 3) / 4; ++c4) {
          if (c4 * 4 < nv) {
            const long long tot = wave_sum4(vals[c4 * 4], vals[c4 * 4 + 1], vals[c4 * 4 + 2], vals[c4 * 4 + 3]);
            if (lane < 4 && c4 * 4 + lane < nv) s_red[(slot + c4 * 4 + lane) * 4 + w] = tot;
          }
        }
      }
      __syncthreads();
      for (int t = tid; t < (g1 - g0) * NV; t += BT) {
        const int gi = t / NV, i = t % NV;
        const RJob& rj = s_job[g0 + gi];
        if (!rj.active || (i > K && !rj.check_nan)) continue;
        const long long s = s_red[t * 4] + s_red[t * 4 + 1] + s_red[t * 4 + 2] + s_red[t * 4 + 3];
        Acc* a = &S.acc[((size_t)par * MAXP + rj.p) * ACC_PER + (chunk & (ACC_SLOTS - 1)) * ACC_STRIDE];
        long long* ax = S.accx + ((size_t)par * MAXP + rj.p) * AX_PER + (size_t)(chunk & (AX_SLOTS - 1)) * AX_REC;
        if (i == 0) {
          const int cL = (int)(s & 0xFFFFF), cR = (int)((s >> 20) & 0xFFFFF), cN = (int)(s >> 40);
          S.cc[(size_t)rj.ccL * S.nchunks + chunk] = (uint16_t)cL;
          S.cc[(size_t)rj.ccR * S.nchunks + chunk] = (uint16_t)cR;
          if (cL | cN) atomicAdd(&a->cnts, (unsigned long long)cL | ((unsigned long long)cN << 32));
        } else if (s != 0) {
          const int k = (i - 1) % K;
          const bool isN = (i - 1) >= K;
          long long* dst = k == 0 ? (isN ? &a->aN : &a->aL) : &ax[(isN ? KX : 0) + k - 1];
          atomicAdd((unsigned long long*)dst, (unsigned long long)s);
        }
      }
      __syncthreads();
    }
    if (do_init) {
      // C, E0, A[0] (+QSTD[0]) -> InitAcc; A[k>0], QSTD[k>0] -> iax
      long long v5[5] = {iv[2], 0, iv[0], iv[1], iv[2 + K]};
      block_sum<5>(v5, s_red);
      if (tid == 0) {
        InitAcc* a = &S.initacc[(size_t)par * IA_SLOTS + (blockIdx.x % IA_SLOTS)];
        if (v5[0]) atomicAdd((unsigned long long*)&a->A, (unsigned long long)v5[0]);
        if (v5[2]) atomicAdd((unsigned long long*)&a->C, (unsigned long long)v5[2]);
        if (v5[3]) atomicAdd((unsigned long long*)&a->E0, (unsigned long long)v5[3]);
        if (v5[4]) atomicAdd((unsigned long long*)&a->QSTD, (unsigned long long)v5[4]);
      }
      for (int k = 1; k < K; ++k) {
        long long v2[2] = {iv[2 + k], iv[2 + K + k]};
        block_sum<2>(v2, s_red);
        if (tid == 0) {
          long long* ix = S.iax + ((size_t)par * IA_SLOTS + (blockIdx.x % IA_SLOTS)) * 2 * KX;
          if (v2[0]) atomicAdd((unsigned long long*)&ix[k - 1], (unsigned long long)v2[0]);
          if (v2[1]) atomicAdd((unsigned long long*)&ix[KX + k - 1], (unsigned long long)v2[1]);
        }
      }
      if (sat) atomicAdd(&S.counters[4], (unsigned long long)sat);
    }
    return;
  }

  // ---------------- lone FINAL
  __syncthreads();
  long long qs[PGB_MAX_OUTPUTS];
  for (int k = 0; k < K; ++k) qs[k] = 0;
  unsigned sat = 0;
  const int nitems = (int)(S.n_pad / BT);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const long long row = (long long)item * BT + tid;
    if (row >= S.n) continue;
    uint32_t id_sel;
    if (cmd->sel_slot == -2) {
      id_sel = tl_old[row];
    } else {
      id_sel = sel_lid ? (uint32_t)sel_lid[row] : 0u;
      tl_old[row] = (uint8_t)id_sel;
    }
    for (int k = 0; k < K; ++k) {
      const size_t ri = (size_t)k * n_pad + row;
      const double nv = lin_pred(s_lv[0][id_sel][k], 0, id_sel, k, row);
      const double st = st_in[ri] + nv;
      if (cmd->tune) {
        const double mean0 = S.rs_mean[ri], m20 = S.rs_m2[ri];
        const double delta = nv - mean0;
        const double mean = mean0 + delta / cntf;
        const double delta2 = nv - mean;
        const double m2 = m20 + delta * delta2;
        S.rs_mean[ri] = mean;
        S.rs_m2[ri] = m2;
        qs[k] += pgb_quant(PGB_SQRT(m2 / cntf), c1, &sat);
      }
      st_out[ri] = st;
    }
  }
  if (cmd->tune) {
    for (int k = 0; k < K; ++k) {
      long long v1[1] = {qs[k]};
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
  if (sat) atomicAdd(&S.counters[4], (unsigned long long)sat);
}

// ------------------------------------------------------------------ k_loglik
// Bernoulli families only ([U] update_weight): after the PARTITION pass of a slot, the children's
// leaf values are known (child_values, the same routine k_ctrl uses one launch later); this pass
// evaluates the per-row log-likelihood of the rows of the leaf that was split -- left / right /
// dropped by a missing value -- and reduces it in fixed point.  Same work items as PARTITION.
// (a template: the single-output kernels keep 64 of these in LDS and carry none of the arrays)
template <bool MK, bool LIN>
struct LJobT {
  long long src, xoff;
  double v, vL, vR;
  int32_t p, rule, label, check_nan, ok, new_label;
  double vLx[MK ? KXMAX : 1], vRx[MK ? KXMAX : 1];  // K-vector leaves: outputs 1..K-1
  // linear response: the children's linear parts
  double slopeL, xbarL, slopeR, xbarR;
  int32_t svarL, svarR;
  double sLx[MK && LIN ? KXMAX : 1], sRx[MK && LIN ? KXMAX : 1];  // ... slopes of outputs 1..K-1
};

// KT: 1 = single output; 2, 3, 4 = that many outputs, loops unrolled; 0 = any K <= PGB_MAX_OUTPUTS
// FAM: the likelihood family when known at compile time (single-output kernels: the per-row
// evaluation then contains one family's code only), -1: read S.family.
// LIN: linear response (single-output families): the children predict value + slope (x - xbar).
template <int KT, int FAM, bool LIN>
__global__ __launch_bounds__(BT) void k_loglik(const Dev* __restrict__ Sp, int par) {
  const Dev& S = *Sp;
  constexpr bool MK = KT != 1;
  constexpr int KB = KT > 0 ? KT : PGB_MAX_OUTPUTS;
  typedef LJobT<MK, LIN> LJob;
  __shared__ long long s_red[MAXP * 3 * 4];
  __shared__ LJob s_job[MAXP];
  __shared__ int s_n[2];
  // log Phi tables in LDS (single-output Bernoulli path): a per-lane row through the vector L1
  // costs a cache-line access per distinct row and instruction; LDS serves them at bank speed
  constexpr bool PROBIT = KT == 1 && FAM == PGB_FAMILY_BERNOULLI_PROBIT;
  __shared__ double s_ln[PROBIT ? (PGB_LN_TN_ROWS + PGB_LN_TP_ROWS) * 9 : 1];
  // (Measured and dropped: listing each wave's matching rows with ballot + mbcnt and evaluating the
  // list densely -- per particle, or through a per-wave queue with three interleaved passes -- is
  // SLOWER at cfg4, 184 k / 171 k vs 223 k particle-steps/s: with a quarter of the lanes active the
  // rare branches of the evaluation are skipped wave-wide, with every lane active they never are.)
  const Cmd* cmd = &S.cmd[par];
  if (!(cmd->kind & CMD_PARTITION)) return;
  if constexpr (PROBIT) {
    const double* gtn = pgb_ln_tn();
    const double* gtp = pgb_ln_tp();
    for (int i = threadIdx.x; i < PGB_LN_TN_ROWS * 9; i += BT) s_ln[i] = gtn[i];
    for (int i = threadIdx.x; i < PGB_LN_TP_ROWS * 9; i += BT) s_ln[PGB_LN_TN_ROWS * 9 + i] = gtp[i];
    // (the barrier after the job list below also publishes the tables)
  }
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const Ctrl cn = S.ctrl[par ^ 1];  // the state this slot's k_ctrl produced
  const int round = cn.round - 1;   // round of the proposals of this slot
  const uint32_t it = (uint32_t)cn.iter;
  // leaf_sd / root statistics in force for this round (k_ctrl of the NEXT slot resolves them the
  // same way): a FINAL or INIT part of this slot's row pass may just have produced them
  double leaf_sd = cn.leaf_sd;
  long long rootA = 0;
  {
    const InitAcc* src = S.initacc + (size_t)par * IA_SLOTS;
    long long qstd = 0;
    for (int k = 0; k < IA_SLOTS; ++k) {
      qstd += src[k].QSTD;
      rootA += src[k].A;
    }
    if (cn.pend_leafsd && cn.pend_iter > 2) leaf_sd = ((double)qstd * S.sc.inv_c1) / (double)S.n;
  }
  const Job* jobs = S.jobs + (size_t)par * MAXP;
  if (tid < 64) {
    Job j;
    j.active = 0;
    if (tid >= 1 && tid < S.P) j = jobs[tid];
    const bool has = j.active != 0;
    const unsigned long long m = __ballot(has);
    // one Philox evaluation per lane: the leaf noise of particle `tid` in this round
    double z0, z1;
    {
      const pgb_u2 ul = pgb_draw2(S.seed, it, (uint32_t)round, (uint32_t)tid, PGB_RNG_LEAF, 0);
      pgb_normal2(ul.u0, ul.u1, &z0, &z1);
    }
    if (has) {
      const int k = __popcll(m & ((1ull << tid) - 1ull));
      const Acc a = load_acc(&S.acc[((size_t)par * MAXP + tid) * ACC_PER]);
      const ChildVals cv = child_values(S, j.rule, j.cnt, round == 0 ? rootA : j.p_q_st, j.p_value, a.cnts,
                                        a.aL, a.aN, z0, z1, leaf_sd);
      LJob lj;
      lj.p = tid;
      lj.rule = j.rule;
      lj.label = j.label;
      lj.new_label = j.new_label;
      lj.check_nan = j.check_nan;
      lj.ok = cv.ok;
      lj.v = j.v;
      lj.vL = cv.vL;
      lj.vR = cv.vR;
      lj.src = j.src_slot < 0 ? -1ll : (long long)(((size_t)j.src_gen * MAXP + j.src_slot) * S.n_pad);
      lj.xoff = (long long)((size_t)j.var * S.n_pad);
      lj.slopeL = lj.xbarL = lj.slopeR = lj.xbarR = 0.0;
      lj.svarL = lj.svarR = -1;
      LinKids lk;
      lk.svarL = lk.svarR = -1;
      lk.linL = lk.linR = false;
      if constexpr (LIN) {
        if (cv.ok == 1) {
          lk = lin_children(S, &S.accu[((size_t)par * MAXP + tid) * ACC_PER], j.var, cv.cL, cv.cR,
                            cv.aL, cv.aR, it, (uint32_t)round, (uint32_t)tid);
          lj.svarL = lk.svarL; lj.slopeL = lk.slopeL; lj.xbarL = lk.xbarL;
          lj.svarR = lk.svarR; lj.slopeR = lk.slopeR; lj.xbarR = lk.xbarR;
        }
      }
      if constexpr (MK)
      for (int kx = 0; kx < (KT > 0 ? KT : S.K) - 1; ++kx) {  // extension outputs: same routine as k_ctrl
        const int KX = (KT > 0 ? KT : S.K) - 1;
        const long long pq = round == 0 ? root_A_x(S, par, kx) : S.jqx[((size_t)par * MAXP + tid) * KX + kx];
        const double pv = round == 0 ? S.init_leaf : S.jvx[((size_t)par * MAXP + tid) * KX + kx];
        ChildX cx = child_values_x(S, cv.ok, cv.cL, cv.cR, load_accx(S.accx, par, tid, kx),
                                   load_accx(S.accx, par, tid, KX + kx), pq, pv,
                                   it, (uint32_t)round, (uint32_t)tid, kx, leaf_sd_x(S, cn, par ^ 1, par, kx));
        lj.vLx[kx] = cx.vL;
        lj.vRx[kx] = cx.vR;
        if constexpr (LIN) {
          if (cv.ok == 1)
            lin_children_x(S, lk, cx, j.var, cv.cL, cv.cR, load_accx(S.accux, par, tid, kx),
                           load_accx(S.accux, par, tid, KX + kx));
          lj.sLx[kx] = cx.sL;
          lj.sRx[kx] = cx.sR;
        }
      }
      if constexpr (MK && LIN)
        if (cv.ok == 1) {  // a further output may have made the leaf linear
          lj.svarL = lk.svarL; lj.slopeL = lk.slopeL; lj.xbarL = lk.xbarL;
          lj.svarR = lk.svarR; lj.slopeR = lk.slopeR; lj.xbarR = lk.xbarR;
        }
      s_job[k] = lj;
    }
    if (tid == 0) s_n[0] = __popcll(m);
  }
  __syncthreads();
  const int nact = s_n[0];
  if (nact == 0) return;
  int G = (nact * S.nchunks + S.ll_target - 1) / S.ll_target;
  if (G < 1) G = 1;
  const int ngroups = (nact + G - 1) / G;
  const int nitems = S.nchunks * ngroups;
  const int K = KT > 0 ? KT : S.K;
  const double* __restrict__ const noi = S.st + (size_t)cn.st_cur * K * S.n_pad;
  const double cl = S.sc.cl;
  // The row pass of this slot has already sorted the rows of every split leaf: left rows kept the
  // leaf's label, right rows carry the new one, dropped rows the orphan label.  Reading those
  // bytes back (1 B per row) replaces a second read of the split column (8 B per row).
  const uint8_t* __restrict__ const newl = S.lid + (size_t)cmd->dst_gen * MAXP * S.n_pad;
  const long long n = S.n;
  unsigned sat = 0;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int chunk = item % S.nchunks, grp = item / S.nchunks;
    const long long base = (long long)chunk * CH + tid * RPT;
    double yv[RPT], nv[RPT];
    uint32_t root_ids = 0;
#pragma unroll
    for (int e = 0; e < RPT; ++e) {
      yv[e] = S.y[base + e];
      nv[e] = noi[base + e];
      if constexpr (KT == 1)
        if (S.has_off) nv[e] = nv[e] + S.off[base + e];  // (adding the default 0.0 would give the same bits)
      if (base + e >= n) root_ids |= (uint32_t)PGB_ORPHAN << (8 * e);
    }
    if constexpr (MK) {  // K-vector leaves: per-row softmax log-likelihood over all outputs
      const int g0 = grp * G, g1 = (g0 + G < nact) ? g0 + G : nact;
      for (int g = g0; g < g1; ++g) {
        const LJob& lj = s_job[g];
        const uint32_t ids = lj.src < 0 ? root_ids : *(const uint32_t*)(S.lid + lj.src + base);
        const uint32_t nid = *(const uint32_t*)(newl + (size_t)lj.p * S.n_pad + base);
        long long v0 = 0, v1 = 0, v2 = 0;
        for (int e = 0; e < RPT; ++e) {
          if (((ids >> (8 * e)) & 255u) == (uint32_t)lj.label) {
            const uint32_t nl = (nid >> (8 * e)) & 255u;
            const int side = nl == (uint32_t)lj.label ? 0 : (nl == (uint32_t)lj.new_label ? 1 : 2);
            double mu[KB];
            double v0k = side == 0 ? lj.vL : side == 1 ? lj.vR : 0.0;
            int sv = -1;
            double xv = 0.0, xb = 0.0;
            if constexpr (LIN) {
              sv = side == 0 ? lj.svarL : side == 1 ? lj.svarR : -1;
              if (sv >= 0) {
                xv = S.XT[lj.xoff + base + e];
                xb = side == 0 ? lj.xbarL : lj.xbarR;
                v0k = pgb_leaf_pred(v0k, side == 0 ? lj.slopeL : lj.slopeR, xb, xv);
              }
            }
            mu[0] = nv[e] + v0k;
#pragma unroll
            for (int k = 1; k < KB; ++k)
              if (k < K) {
                double vk = side == 0 ? lj.vLx[k - 1] : side == 1 ? lj.vRx[k - 1] : 0.0;
                if constexpr (LIN)
                  if (sv >= 0) vk = pgb_leaf_pred(vk, side == 0 ? lj.sLx[k - 1] : lj.sRx[k - 1], xb, xv);
                mu[k] = noi[(size_t)k * S.n_pad + base + e] + vk;
              }
            const long long q = pgb_quant(pgb_loglik(S.family, K, yv[e], mu), cl, &sat);
            if (side == 0) v0 += q; else if (side == 1) v1 += q; else v2 += q;
          }
        }
        const int slot = (g - g0) * 3;
        const long long tot = wave_sum4(v0, v1, v2, 0);  // lane l: total of value l & 3
        if (lane < 3) s_red[(slot + lane) * 4 + w] = tot;
      }
      __syncthreads();
      for (int t = tid; t < (g1 - g0) * 3; t += BT) {
        const int gi = t / 3, i = t % 3;
        const long long s = s_red[t * 4] + s_red[t * 4 + 1] + s_red[t * 4 + 2] + s_red[t * 4 + 3];
        if (s != 0) {
          AccL* a = &S.accl[((size_t)par * MAXP + s_job[g0 + gi].p) * LL_PER + (chunk & (LL_SLOTS - 1)) * LL_STRIDE];
          atomicAdd((unsigned long long*)(i == 0 ? &a->llL : i == 1 ? &a->llR : &a->llN), (unsigned long long)s);
        }
      }
      __syncthreads();
      continue;
    }
    const int g0 = grp * G, g1 = (g0 + G < nact) ? g0 + G : nact;
    // the label words of the next particle are requested before this one is evaluated
    uint32_t ids_n = s_job[g0].src < 0 ? root_ids : *(const uint32_t*)(S.lid + s_job[g0].src + base);
    uint32_t nid_n = *(const uint32_t*)(newl + (size_t)s_job[g0].p * S.n_pad + base);
    for (int g = g0; g < g1; ++g) {
      const LJob& lj = s_job[g];
      const uint32_t ids = ids_n, nid = nid_n;
      if (g + 1 < g1) {
        const LJob& ln = s_job[g + 1];
        ids_n = ln.src < 0 ? root_ids : *(const uint32_t*)(S.lid + ln.src + base);
        nid_n = *(const uint32_t*)(newl + (size_t)ln.p * S.n_pad + base);
      }
      long long v0 = 0, v1 = 0, v2 = 0;  // llL, llR, llN
#pragma unroll
      for (int e = 0; e < RPT; ++e) {
        if (((ids >> (8 * e)) & 255u) == (uint32_t)lj.label) {
          // ONE evaluation per row: the side only selects the leaf value and the accumulator
          // (separate calls per side would run one after the other on a divergent wave)
          const uint32_t nl = (nid >> (8 * e)) & 255u;
          const int side = nl == (uint32_t)lj.label ? 0 : (nl == (uint32_t)lj.new_label ? 1 : 2);
          double vleaf = side == 0 ? lj.vL : side == 1 ? lj.vR : 0.0;  // dropped: predicts 0
          if constexpr (LIN) {
            const int sv = side == 0 ? lj.svarL : side == 1 ? lj.svarR : -1;
            if (sv >= 0) {
              const double xv = S.XT[lj.xoff + base + e];
              vleaf = pgb_leaf_pred(vleaf, side == 0 ? lj.slopeL : lj.slopeR, side == 0 ? lj.xbarL : lj.xbarR, xv);
            }
          }
          const double mu = nv[e] + vleaf;
          const long long q = pgb_quant(pgb_loglik1q(FAM >= 0 ? FAM : S.family, yv[e], mu, cn.inv_sigma2, cn.lik_param2,
                                                      PROBIT ? s_ln : pgb_ln_tn(),
                                                      PROBIT ? s_ln + PGB_LN_TN_ROWS * 9 : pgb_ln_tp()), cl, &sat);
          v0 += side == 0 ? q : 0;
          v1 += side == 1 ? q : 0;
          v2 += side == 2 ? q : 0;
        }
      }
      const int slot = (g - g0) * 3;
      const long long tot = wave_sum4(v0, v1, v2, 0);  // lane l: total of value l & 3
      if (lane < 3) s_red[(slot + lane) * 4 + w] = tot;
    }
    __syncthreads();
    for (int t = tid; t < (g1 - g0) * 3; t += BT) {
      const int gi = t / 3, i = t % 3;
      const long long s = s_red[t * 4] + s_red[t * 4 + 1] + s_red[t * 4 + 2] + s_red[t * 4 + 3];
      if (s != 0) {
        AccL* a = &S.accl[((size_t)par * MAXP + s_job[g0 + gi].p) * LL_PER + (chunk & (LL_SLOTS - 1)) * LL_STRIDE];
        atomicAdd((unsigned long long*)(i == 0 ? &a->llL : i == 1 ? &a->llR : &a->llN), (unsigned long long)s);
      }
    }
    __syncthreads();
  }
  if (sat) atomicAdd(&S.counters[4], (unsigned long long)sat);
}

// ------------------------------------------------------------------ setup kernels
// X row-major [n][ldx] -> XT column-major [p][n_pad]; LDS-tiled 32x32 transpose so that both
// the read and the write are coalesced.  Also flags columns that contain NaN.
__global__ __launch_bounds__(BT) void k_transpose(const double* __restrict__ X, long long ldx,
                                                  double* __restrict__ XT, long long n,
                                                  long long n_pad, int p, int32_t* col_nan) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const long long r0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  for (int k = ty; k < 32; k += 8) {
    long long r = r0 + k;
    int c = c0 + tx;
    tile[k][tx] = (r < n && c < p) ? X[r * ldx + c] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    int c = c0 + k;
    long long r = r0 + tx;
    if (c < p && r < n_pad) {
      double x = tile[tx][k];
      XT[(size_t)c * n_pad + r] = x;
      if (x != x) col_nan[c] = 1;
    }
  }
}

__global__ void k_init_linp(LinP* p, long long n) {  // constant leaves everywhere
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = LinP{0.0, 0.0, -1};
}
// per-column max |x| (NaN ignored) of the column-major copy: one workgroup per column
__global__ __launch_bounds__(BT) void k_colmax(const double* __restrict__ XT, long long n, long long n_pad,
                                               double* __restrict__ amax) {
  __shared__ double sm[BT];
  const double* c = XT + (size_t)blockIdx.x * n_pad;
  double a = 0.0;
  for (long long i = threadIdx.x; i < n; i += BT) {
    double v = c[i];
    v = v < 0.0 ? -v : v;
    if (v > a) a = v;
  }
  sm[threadIdx.x] = a;
  __syncthreads();
  for (int o = BT / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o && sm[threadIdx.x + o] > sm[threadIdx.x]) sm[threadIdx.x] = sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) amax[blockIdx.x] = sm[0];
}
__global__ void k_fill_f64(double* a, long long n, double v) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}

__global__ void k_init_tree_lid(uint8_t* a, long long n, long long n_pad, int m) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pad * m) a[i] = (i % n_pad) < n ? 0 : PGB_ORPHAN;
}

__global__ void k_init_trees(DTree* trees, int m, long long n, double init_leaf) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m) return;
  DTree* T = &trees[t];
  T->n_nodes = 1;
  T->n_leaves = 1;
  DNode z;
  memset(&z, 0, sizeof z);
  z.var = -1;
  z.cc_row = -1;
  z.cnt = (int32_t)n;
  z.value = init_leaf;
  T->nd[0] = z;
}

// integer split weights from the user's prior + their prefix sums (numeric contract:
// pgb_alpha_init / pgb_sample_var)
__global__ void k_init_alpha(const double* prior, double max_prior, long long* alpha, long long* cdfS, int p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    long long cs = 0;
    for (int j = 0; j < p; ++j) {
      const long long a = pgb_alpha_init(prior[j], max_prior);
      alpha[j] = a;
      cs += a;
      cdfS[j] = cs;
    }
  }
}

// ------------------------------------------------------------------ prediction
// out[d][k][row] = sum over the m trees of forest d of the leaf value reached by X[row,:]
// (PosteriorSampler.sample_posterior, utils.py:66-69); excluded / NaN splits average both
// subtrees by their training counts (CHANGELOG.md:410-411).  One thread per (row, forest).
struct PredTrees {
  const int32_t* node_off;
  const int32_t* var;
  const double* split;
  const int32_t* left;
  const int32_t* right;
  const long long* count;
  const double* value;
  // linear leaves (svar == nullptr: none)
  const double* slope;
  const double* xbar;
  const int32_t* svar;
};

__global__ __launch_bounds__(BT) void k_predict(PredTrees T, const int32_t* forest_idx, int n_forests,
                                                int m, int K, const double* __restrict__ X,
                                                long long n_rows, int p, long long ldx,
                                                const int32_t* rules, const uint8_t* excl,
                                                double* out) {
  const long long row = (long long)blockIdx.x * BT + threadIdx.x;
  const int d = blockIdx.y;
  if (row >= n_rows) return;
  const double* x = X + row * ldx;
  double acc[PGB_MAX_OUTPUTS];
  for (int o = 0; o < K; ++o) acc[o] = 0.0;
  int stk_node[PGB_MAX_DEPTH + 2];
  double stk_w[PGB_MAX_DEPTH + 2];
  for (int t = 0; t < m; ++t) {
    const int base = T.node_off[forest_idx[(size_t)d * m + t]];
    int sp = 0;
    stk_node[0] = 0;
    stk_w[0] = 1.0;
    sp = 1;
    while (sp > 0) {
      --sp;
      int k = stk_node[sp];
      double w = stk_w[sp];
      for (;;) {
        const int g = base + k;
        const int j = T.var[g];
        if (j < 0) {
          int js = -1;  // linear leaf; a missing / excluded regressor: the mean
          if (T.svar != nullptr) {
            js = T.svar[g];
            if (js >= 0 && (excl[js] || x[js] != x[js])) js = -1;
          }
          for (int o = 0; o < K; ++o) {
            double vo = T.value[(size_t)g * K + o];
            if (js >= 0) vo = pgb_leaf_pred(vo, T.slope[(size_t)g * K + o], T.xbar[g], x[js]);
            acc[o] += w * vo;
          }
          break;
        }
        const double xv = x[j];
        if (excl[j] || xv != xv) {
          const int l = T.left[g], r = T.right[g];
          const double cl = (double)T.count[base + l], cr = (double)T.count[base + r];
          const double tot = cl + cr;
          if (!(tot > 0.0)) break;
          // depth-first, left first (same summation order as the oracle's recursion)
          stk_node[sp] = r;
          stk_w[sp] = w * (cr / tot);
          ++sp;
          k = l;
          w = w * (cl / tot);
          continue;
        }
        const bool gl = pgb_go_left(rules[j], xv, T.split[g]) != 0;
        k = gl ? T.left[g] : T.right[g];
      }
    }
  }
  for (int o = 0; o < K; ++o) out[((size_t)d * K + o) * n_rows + row] = acc[o];
}

// ------------------------------------------------------------------ host side
static thread_local char g_err[512];
static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}
static int fail_hip(hipError_t e, const char* what) {
  snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
  return PGB_E_DEVICE;
}
#define HIPCHK(expr)                                     \
  do {                                                   \
    hipError_t e_ = (expr);                              \
    if (e_ != hipSuccess) return fail_hip(e_, #expr);    \
  } while (0)

struct pgb_handle {
  pgb_settings s;
  Dev d;
  Dev* d_dev;                          // device-resident copy passed to every kernel
  volatile unsigned long long* flag;   // pinned host word written by k_ctrl (completed asteps)
  long long steps_target;              // asteps requested so far
  std::vector<hipEvent_t> bundle_ev;   // throttle: at most 3 bundles of slots in flight
  long long bundles;
  hipStream_t stream;
  std::vector<void*> allocs;
  std::vector<size_t> alloc_bytes;     // per allocation: payload size ...
  std::vector<char> alloc_persist;     // ... and whether a checkpoint carries it
  long long slot;  // next slot index (parity = slot & 1)
  int st_cur, alpha_cur;  // mirrors of Ctrl::st_cur / alpha_cur at the last idle point
  int have_data, have_y;
  int has_subset;  // any SubsetSplit column: selects the row-pass instance
  int rows_grid;   // workgroups of the persistent row-pass grid (dispatch costs ~3.5 ns each)
  int ll_grid;     // ... of the log-likelihood pass
  int sigma_dirty;
  double inv_sigma2;
  double lik_param2;
  int lower_host;      // mirror of the batch cursor
  int last_lower, last_n;
  double slots_per_step;  // running estimate
  pgb_counters ctr;
  // profiling of the dominant kernel (k_rows)
  int prof;
  std::vector<hipEvent_t> ev;
  size_t ev_used;
  double prof_ms;
  long long prof_launches;
  long long* prof_buf;    // device-clock stamps (allocated on first use)
  long long prof_slot0;   // first slot of the profiled region (device-clock stamps)
  double prof_clock_ms;   // sum over launches of max(end) - min(start), 100 MHz device clock
  long long prof_clock_launches;
};

template <typename T>
static int dalloc(pgb_handle* h, T** p, size_t count) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, count * sizeof(T) + 256);
  if (e != hipSuccess) return fail_hip(e, "hipMalloc");
  h->allocs.push_back(q);
  h->alloc_bytes.push_back(count * sizeof(T));
  h->alloc_persist.push_back(1);
  *p = (T*)q;
  return PGB_OK;
}
// data, per-tree scratch and pointer tables are rebuilt by create/set_data: not part of a checkpoint
static void transient(pgb_handle* h) { h->alloc_persist.back() = 0; }

extern "C" const char* pgb_last_error(void) { return g_err; }
extern "C" const char* pgb_backend_name(void) { return "hip-gfx950"; }

extern "C" int pgb_create(const pgb_settings* s, void* stream, pgb_handle** out) {
  if (!s || !out) return fail(PGB_E_INVALID, "null argument");
  if (s->n < 1 || s->p < 1 || s->m < 1) return fail(PGB_E_INVALID, "n, p, m must be >= 1");
  if (s->n >= (1ll << 31) - CH) return fail(PGB_E_UNSUPPORTED, "n too large");
  if (s->num_particles < 2 || s->num_particles > PGB_MAX_PARTICLES)
    return fail(PGB_E_INVALID, "num_particles must be in [2, 64]");
  if (s->family == PGB_FAMILY_CATEGORICAL) {
    if (s->n_outputs < 2 || s->n_outputs > PGB_MAX_OUTPUTS)
      return fail(PGB_E_INVALID, "CATEGORICAL needs 2 <= n_outputs <= 8");
  } else if (s->family == PGB_FAMILY_NORMAL_MEANSCALE) {
    if (s->n_outputs != 2) return fail(PGB_E_INVALID, "NORMAL_MEANSCALE needs n_outputs == 2");
  } else if (s->family == PGB_FAMILY_NORMAL || s->family == PGB_FAMILY_BERNOULLI_PROBIT ||
             s->family == PGB_FAMILY_BERNOULLI_LOGIT || s->family == PGB_FAMILY_POISSON_LOG ||
             s->family == PGB_FAMILY_NEGBIN_LOG || s->family == PGB_FAMILY_ASYMLAPLACE ||
             s->family == PGB_FAMILY_STUDENT_T || s->family == PGB_FAMILY_GAMMA_LOG) {
    if (s->n_outputs != 1) return fail(PGB_E_INVALID, "this family has a single output");
  } else {
    return fail(PGB_E_UNSUPPORTED, "unknown family");
  }
  if (s->batch_tune < 1 || s->batch_draw < 1) return fail(PGB_E_INVALID, "batch sizes must be >= 1");
  if (s->response != PGB_RESPONSE_CONSTANT) {
    if (s->response != PGB_RESPONSE_LINEAR && s->response != PGB_RESPONSE_MIX)
      return fail(PGB_E_UNSUPPORTED, "unknown response");
  }
  int ndev = 0;
  hipError_t e0 = hipGetDeviceCount(&ndev);
  if (e0 != hipSuccess || ndev < 1) {
    snprintf(g_err, sizeof g_err, "no HIP device visible (%s, count %d); this backend has no CPU fallback",
             hipGetErrorString(e0), ndev);
    return PGB_E_DEVICE;
  }
  pgb_handle* h = new pgb_handle();
  h->s = *s;
  h->stream = (hipStream_t)stream;
  h->slot = 0;
  h->has_subset = 0;
  h->rows_grid = 1024;
  h->prof_buf = nullptr;
  h->prof_clock_ms = 0.0;
  h->prof_clock_launches = 0;
  h->prof_slot0 = 0;
  if (const char* e = getenv("PGB_ROWS_GRID")) h->rows_grid = atoi(e) > 0 ? atoi(e) : h->rows_grid;
  h->ll_grid = h->rows_grid;
  if (const char* e = getenv("PGB_LL_GRID")) h->ll_grid = atoi(e) > 0 ? atoi(e) : h->ll_grid;
  h->st_cur = 0;
  h->alpha_cur = 0;
  h->d_dev = nullptr;
  h->flag = nullptr;
  h->steps_target = 0;
  h->bundles = 0;
  h->inv_sigma2 = 1.0;
  h->lik_param2 = 1.0;
  h->sigma_dirty = 1;
  h->slots_per_step = 0.0;
  memset(&h->ctr, 0, sizeof h->ctr);
  Dev& d = h->d;
  memset(&d, 0, sizeof d);
  d.n = s->n;
  d.nchunks = (int)((s->n + CH - 1) / CH);
  d.n_pad = (long long)d.nchunks * CH;
  d.p = s->p;
  d.m = s->m;
  d.P = s->num_particles;
  d.family = s->family;
  d.K = s->n_outputs;
  d.response = s->response;
  d.lin_R = pgb_pow2(s->range_exp - 1);
  d.inv_R = pgb_pow2(1 - s->range_exp);
  d.rows_target = ROWS_TARGET_ITEMS;
  d.rows_target_init = ROWS_TARGET_ITEMS_INIT;
  if (const char* e = getenv("PGB_ROWS_TARGET")) d.rows_target = atoi(e) > 0 ? atoi(e) : d.rows_target;
  d.ll_target = d.rows_target;
  d.ll_pad = 0;
  if (const char* e = getenv("PGB_LL_TARGET")) d.ll_target = atoi(e) > 0 ? atoi(e) : d.ll_target;
  if (const char* e = getenv("PGB_ROWS_TARGET_INIT")) d.rows_target_init = atoi(e) > 0 ? atoi(e) : d.rows_target_init;
  d.batch_tune = s->batch_tune;
  d.batch_draw = s->batch_draw;
  d.seed = s->seed;
  d.init_leaf = s->init_leaf;
  d.mdouble = (double)s->m;
  d.sc = pgb_make_scales(s->n, s->range_exp);
  int rc;
  double *XT, *y, *st, *rs_mean, *rs_m2, *prior_leaf;
  long long *alpha, *cdfS;
  double2* pack;
  uint8_t *tree_lid, *lid;
  uint16_t* cc;
  int32_t *vi, *rules, *col_nan;
#define DA(ptr, cnt) \
  if ((rc = dalloc(h, &ptr, (size_t)(cnt))) != PGB_OK) { pgb_destroy(h); return rc; }
  DA(XT, (size_t)d.p * d.n_pad);
  transient(h);
  DA(y, d.n_pad);
  transient(h);
  double* off;
  DA(off, d.n_pad);
  transient(h);
  d.off = off;
  const int K = d.K, KX = d.K - 1;
  DA(st, (size_t)2 * K * d.n_pad);
  DA(pack, d.n_pad);
  DA(rs_mean, (size_t)K * d.n_pad);
  DA(rs_m2, (size_t)K * d.n_pad);
  if (KX > 0) {  // K-vector leaves: extension outputs
    DA(d.packx, (size_t)KX * d.n_pad);
    DA(d.pvx, (size_t)2 * MAXP * MAXN * KX);
    DA(d.pqx, (size_t)2 * MAXP * MAXN * KX);
    DA(d.tvx, (size_t)d.m * MAXN * KX);
    DA(d.accx, (size_t)2 * MAXP * AX_PER);
    DA(d.iax, (size_t)2 * IA_SLOTS * 2 * KX);
    DA(d.lvx, (size_t)2 * 2 * 256 * KX);
    DA(d.jqx, (size_t)2 * MAXP * KX);
    DA(d.jvx, (size_t)2 * MAXP * KX);
    DA(d.lsdx, (size_t)2 * KXMAX);
  }
  DA(tree_lid, (size_t)d.m * d.n_pad);
  DA(lid, (size_t)NGEN * MAXP * d.n_pad);
  transient(h);  // particle labels live for one tree update only
  DA(cc, (size_t)CC_ROUNDS * MAXP * 2 * d.nchunks);
  DA(d.trees, d.m);
  DA(d.parts, 2 * MAXP);
  DA(d.jobs, 2 * MAXP);
  DA(d.acc, 2 * MAXP * ACC_PER);
  DA(d.accl, 2 * MAXP * LL_PER);
  DA(d.jobl, 2 * MAXP);
  DA(d.initacc, 2 * IA_SLOTS);
  DA(d.cmd, 2);
  DA(d.ctrl, 2);
  DA(d.counters, 8);
  DA(vi, d.p);
  DA(alpha, 2 * d.p);
  DA(cdfS, 2 * d.p);
  DA(prior_leaf, PGB_MAX_DEPTH);
  DA(rules, d.p);
  DA(col_nan, d.p);
  int32_t* col_ex;
  DA(col_ex, d.p);
  d.col_ex = col_ex;
  if (d.response != PGB_RESPONSE_CONSTANT) {
    DA(d.plin, (size_t)2 * MAXP * MAXN);
    DA(d.tlin, (size_t)d.m * MAXN);
    DA(d.lvl, (size_t)2 * 2 * 256);
    DA(d.accu, (size_t)2 * MAXP * ACC_PER);
    if (KX > 0) {
      DA(d.psx, (size_t)2 * MAXP * MAXN * KX);
      DA(d.tsx, (size_t)d.m * MAXN * KX);
      DA(d.lsx, (size_t)2 * 2 * 256 * KX);
      DA(d.accux, (size_t)2 * MAXP * AX_PER);
    }
  }
#undef DA
  d.XT = XT; d.y = y; d.st = st; d.pack = pack; d.rs_mean = rs_mean; d.rs_m2 = rs_m2;
  d.tree_lid = tree_lid; d.lid = lid; d.cc = cc; d.vi = vi; d.alpha = alpha; d.cdfS = cdfS;
  d.rules = rules; d.col_nan = col_nan; d.prior_leaf = prior_leaf;
  hipStream_t sm = h->stream;
  hipError_t e;
#define HC(expr) \
  if ((e = (expr)) != hipSuccess) { int r_ = fail_hip(e, #expr); pgb_destroy(h); return r_; }
  {
    void* hp = nullptr;
    HC(hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent));
    h->flag = (volatile unsigned long long*)hp;
    *h->flag = 0;
    void* dp = nullptr;
    HC(hipHostGetDevicePointer(&dp, hp, 0));
    d.host_flag = (unsigned long long*)dp;
#ifdef PGB_TRACE
    if ((rc = dalloc(h, &d.trace, (size_t)TRACE_SLOTS * 16)) != PGB_OK) { pgb_destroy(h); return rc; }
    transient(h);
    HC(hipMemsetAsync(d.trace, 0, (size_t)TRACE_SLOTS * 16 * sizeof(long long), sm));
#endif
    if ((rc = dalloc(h, &h->d_dev, 1)) != PGB_OK) { pgb_destroy(h); return rc; }
    transient(h);  // holds device pointers
    HC(hipMemcpyAsync(h->d_dev, &d, sizeof(Dev), hipMemcpyHostToDevice, sm));
    for (int i = 0; i < 4; ++i) {
      hipEvent_t ev;
      HC(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      h->bundle_ev.push_back(ev);
    }
  }
  HC(hipMemcpyAsync(prior_leaf, s->prior_leaf, PGB_MAX_DEPTH * sizeof(double), hipMemcpyHostToDevice, sm));
  HC(hipMemsetAsync(y, 0, d.n_pad * sizeof(double), sm));
  HC(hipMemsetAsync(off, 0, d.n_pad * sizeof(double), sm));
  HC(hipMemsetAsync(pack, 0, d.n_pad * sizeof(double2), sm));
  HC(hipMemsetAsync(rs_mean, 0, (size_t)K * d.n_pad * sizeof(double), sm));
  HC(hipMemsetAsync(rs_m2, 0, (size_t)K * d.n_pad * sizeof(double), sm));
  if (KX > 0) {
    HC(hipMemsetAsync(d.packx, 0, (size_t)KX * d.n_pad * sizeof(double), sm));
    HC(hipMemsetAsync(d.pvx, 0, (size_t)2 * MAXP * MAXN * KX * sizeof(double), sm));
    HC(hipMemsetAsync(d.pqx, 0, (size_t)2 * MAXP * MAXN * KX * sizeof(long long), sm));
    HC(hipMemsetAsync(d.accx, 0, (size_t)2 * MAXP * AX_PER * sizeof(long long), sm));
    HC(hipMemsetAsync(d.iax, 0, (size_t)2 * IA_SLOTS * 2 * KX * sizeof(long long), sm));
    HC(hipMemsetAsync(d.lvx, 0, (size_t)2 * 2 * 256 * KX * sizeof(double), sm));
    HC(hipMemsetAsync(d.jqx, 0, (size_t)2 * MAXP * KX * sizeof(long long), sm));
    HC(hipMemsetAsync(d.jvx, 0, (size_t)2 * MAXP * KX * sizeof(double), sm));
    hipLaunchKernelGGL(k_fill_f64, dim3(1), dim3(256), 0, sm, d.lsdx, (long long)2 * KXMAX, s->init_leaf_sd);
    // every accepted tree starts as a stump whose K-vector leaf is init_leaf
    hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)(((size_t)d.m * MAXN * KX + 255) / 256)), dim3(256), 0, sm,
                       d.tvx, (long long)d.m * MAXN * KX, s->init_leaf);
  }
  HC(hipMemsetAsync(lid, PGB_ORPHAN, (size_t)NGEN * MAXP * d.n_pad, sm));
  HC(hipMemsetAsync(cc, 0, (size_t)CC_ROUNDS * MAXP * 2 * d.nchunks * sizeof(uint16_t), sm));
  HC(hipMemsetAsync(d.parts, 0, 2 * MAXP * sizeof(DPart), sm));
  HC(hipMemsetAsync(d.jobs, 0, 2 * MAXP * sizeof(Job), sm));
  HC(hipMemsetAsync(d.acc, 0, 2 * MAXP * ACC_PER * sizeof(Acc), sm));
  HC(hipMemsetAsync(d.accl, 0, 2 * MAXP * LL_PER * sizeof(AccL), sm));
  HC(hipMemsetAsync(d.jobl, 0, 2 * MAXP * sizeof(JobL), sm));
  HC(hipMemsetAsync(d.initacc, 0, 2 * IA_SLOTS * sizeof(InitAcc), sm));
  HC(hipMemsetAsync(d.cmd, 0, 2 * sizeof(Cmd), sm));
  HC(hipMemsetAsync(d.counters, 0, 8 * sizeof(unsigned long long), sm));
  HC(hipMemsetAsync(vi, 0, d.p * sizeof(int32_t), sm));
  HC(hipMemsetAsync(col_nan, 0, d.p * sizeof(int32_t), sm));
  HC(hipMemsetAsync(col_ex, 0, d.p * sizeof(int32_t), sm));
  if (d.response != PGB_RESPONSE_CONSTANT) {
    const long long n1 = (long long)2 * MAXP * MAXN, n2 = (long long)d.m * MAXN, n3 = 2 * 2 * 256;
    hipLaunchKernelGGL(k_init_linp, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, sm, d.plin, n1);
    hipLaunchKernelGGL(k_init_linp, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, sm, d.tlin, n2);
    hipLaunchKernelGGL(k_init_linp, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, sm, d.lvl, n3);
    HC(hipMemsetAsync(d.accu, 0, (size_t)2 * MAXP * ACC_PER * sizeof(AccU), sm));
    if (KX > 0) {
      HC(hipMemsetAsync(d.psx, 0, (size_t)2 * MAXP * MAXN * KX * sizeof(double), sm));
      HC(hipMemsetAsync(d.tsx, 0, (size_t)d.m * MAXN * KX * sizeof(double), sm));
      HC(hipMemsetAsync(d.lsx, 0, (size_t)2 * 2 * 256 * KX * sizeof(double), sm));
      HC(hipMemsetAsync(d.accux, 0, (size_t)2 * MAXP * AX_PER * sizeof(long long), sm));
    }
  }
  Ctrl c0;
  memset(&c0, 0, sizeof c0);
  c0.phase = PH_IDLE;
  c0.leaf_sd = s->init_leaf_sd;
  c0.inv_sigma2 = 1.0;
  c0.lik_param2 = 1.0;
  Ctrl cc2[2] = {c0, c0};
  HC(hipMemcpyAsync(d.ctrl, cc2, sizeof cc2, hipMemcpyHostToDevice, sm));
  hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)((2 * K * d.n_pad + 255) / 256)), dim3(256), 0, sm, st,
                     2 * K * d.n_pad, s->init_sum);
  hipLaunchKernelGGL(k_init_tree_lid, dim3((unsigned)((d.n_pad * d.m + 255) / 256)), dim3(256), 0, sm,
                     tree_lid, d.n, d.n_pad, d.m);
  hipLaunchKernelGGL(k_init_trees, dim3((d.m + 63) / 64), dim3(64), 0, sm, d.trees, d.m, d.n,
                     s->init_leaf);
  HC(hipGetLastError());
  HC(hipStreamSynchronize(sm));
#undef HC
  *out = h;
  return PGB_OK;
}

extern "C" int pgb_destroy(pgb_handle* h) {
  if (!h) return PGB_OK;
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->bundle_ev) (void)hipEventDestroy(e);
  if (h->flag) (void)hipHostFree((void*)h->flag);
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
  return PGB_OK;
}

extern "C" int pgb_set_data(pgb_handle* h, const double* X_dev, int64_t ldx, const int32_t* rules_host,
                            const double* split_prior_host) {
  if (!h || !X_dev || !rules_host || !split_prior_host) return fail(PGB_E_INVALID, "null argument");
  Dev& d = h->d;
  if (ldx < d.p) return fail(PGB_E_INVALID, "ldx < p");
  double mx = 0.0;
  h->has_subset = 0;
  for (int j = 0; j < d.p; ++j) {
    if (rules_host[j] == PGB_RULE_SUBSET) h->has_subset = 1;
    if (rules_host[j] != PGB_RULE_CONTINUOUS && rules_host[j] != PGB_RULE_ONEHOT &&
        rules_host[j] != PGB_RULE_SUBSET)
      return fail(PGB_E_UNSUPPORTED, "unknown split rule");
    if (!(split_prior_host[j] > 0.0)) return fail(PGB_E_INVALID, "split_prior must be positive");
    if (split_prior_host[j] > mx) mx = split_prior_host[j];
    if (d.response != PGB_RESPONSE_CONSTANT && rules_host[j] != PGB_RULE_CONTINUOUS)
      return fail(PGB_E_UNSUPPORTED, "response linear/mix needs ContinuousSplit columns");
  }
  d.max_prior = mx;
  d.alpha_unit = pgb_alpha_unit(mx);
  hipStream_t sm = h->stream;
  HIPCHK(hipMemcpyAsync((void*)d.rules, rules_host, d.p * sizeof(int32_t), hipMemcpyHostToDevice, sm));
  // the prior is staged in the (not yet used) running-sd buffer and quantised on the device
  HIPCHK(hipMemcpyAsync(d.rs_mean, split_prior_host, (size_t)(d.p < d.n_pad ? d.p : 0) * sizeof(double),
                        hipMemcpyHostToDevice, sm));
  HIPCHK(hipMemsetAsync((void*)d.col_nan, 0, d.p * sizeof(int32_t), sm));
  dim3 grid((unsigned)(d.n_pad / 32), (unsigned)((d.p + 31) / 32));
  hipLaunchKernelGGL(k_transpose, grid, dim3(BT), 0, sm, X_dev, (long long)ldx, (double*)d.XT, d.n,
                     d.n_pad, d.p, (int32_t*)d.col_nan);
  double* prior_stage = nullptr;
  if (d.p >= d.n_pad) {  // more columns than padded rows: stage through a temporary
    HIPCHK(hipMalloc((void**)&prior_stage, d.p * sizeof(double)));
    HIPCHK(hipMemcpyAsync(prior_stage, split_prior_host, d.p * sizeof(double), hipMemcpyHostToDevice, sm));
  }
  hipLaunchKernelGGL(k_init_alpha, dim3(1), dim3(64), 0, sm, prior_stage ? prior_stage : d.rs_mean, mx,
                     d.alpha, d.cdfS, d.p);
  if (d.response != PGB_RESPONSE_CONSTANT) {  // exponent bound of every column (u = x 2^-ex)
    double* amax_dev = nullptr;
    HIPCHK(hipMalloc((void**)&amax_dev, d.p * sizeof(double)));
    hipLaunchKernelGGL(k_colmax, dim3((unsigned)d.p), dim3(BT), 0, sm, d.XT, d.n, d.n_pad, amax_dev);
    std::vector<double> amax(d.p);
    HIPCHK(hipMemcpyAsync(amax.data(), amax_dev, d.p * sizeof(double), hipMemcpyDeviceToHost, sm));
    HIPCHK(hipStreamSynchronize(sm));
    (void)hipFree(amax_dev);
    std::vector<int32_t> ex(d.p);
    for (int j = 0; j < d.p; ++j) ex[j] = pgb_col_exponent(amax[j]);
    HIPCHK(hipMemcpyAsync((void*)d.col_ex, ex.data(), d.p * sizeof(int32_t), hipMemcpyHostToDevice, sm));
    HIPCHK(hipStreamSynchronize(sm));
  }
  HIPCHK(hipMemsetAsync(d.rs_mean, 0, d.n_pad * sizeof(double), sm));
  HIPCHK(hipMemcpyAsync(h->d_dev, &d, sizeof(Dev), hipMemcpyHostToDevice, sm));  // alpha_unit, max_prior
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(sm));
  if (prior_stage) (void)hipFree(prior_stage);
  h->have_data = 1;
  return PGB_OK;
}

extern "C" int pgb_set_response(pgb_handle* h, const double* y_dev) {
  if (!h || !y_dev) return fail(PGB_E_INVALID, "null argument");
  HIPCHK(hipMemcpyAsync((void*)h->d.y, y_dev, h->d.n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->have_y = 1;
  return PGB_OK;
}

extern "C" int pgb_set_offset(pgb_handle* h, const double* offset_dev) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  if (h->s.family == PGB_FAMILY_NORMAL || h->s.n_outputs != 1)
    return fail(PGB_E_UNSUPPORTED, "offsets are for the single-output per-row families");
  if (offset_dev)
    HIPCHK(hipMemcpyAsync((void*)h->d.off, offset_dev, h->d.n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  else
    HIPCHK(hipMemsetAsync((void*)h->d.off, 0, h->d.n * sizeof(double), h->stream));
  if (h->d.has_off != (offset_dev ? 1 : 0)) {
    h->d.has_off = offset_dev ? 1 : 0;
    HIPCHK(hipMemcpyAsync(h->d_dev, &h->d, sizeof(Dev), hipMemcpyHostToDevice, h->stream));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  return PGB_OK;
}

extern "C" int pgb_set_likelihood(pgb_handle* h, const double* params, int32_t n_params) {
  if (!h || !params) return fail(PGB_E_INVALID, "null argument");
  if (h->s.family == PGB_FAMILY_NORMAL) {
    if (n_params != 1 || !(params[0] > 0.0)) return fail(PGB_E_INVALID, "NORMAL needs sigma > 0");
    h->inv_sigma2 = 1.0 / (params[0] * params[0]);
    h->sigma_dirty = 1;
  } else if (h->s.family == PGB_FAMILY_NEGBIN_LOG || h->s.family == PGB_FAMILY_GAMMA_LOG) {
    // the slot doubles as "the family's parameter"
    if (n_params != 1 || !(params[0] > 0.0)) return fail(PGB_E_INVALID, "NEGBIN_LOG / GAMMA_LOG need alpha > 0");
    h->inv_sigma2 = params[0];
    h->sigma_dirty = 1;
  } else if (h->s.family == PGB_FAMILY_ASYMLAPLACE) {
    if (n_params != 2 || !(params[0] > 0.0) || !(params[1] > 0.0) || !(params[1] < 1.0))
      return fail(PGB_E_INVALID, "ASYMLAPLACE needs b > 0 and 0 < q < 1");
    h->inv_sigma2 = params[0];
    h->lik_param2 = params[1];
    h->sigma_dirty = 1;
  } else if (h->s.family == PGB_FAMILY_STUDENT_T) {
    if (n_params != 2 || !(params[0] > 0.0) || !(params[1] > 0.0))
      return fail(PGB_E_INVALID, "STUDENT_T needs sigma > 0 and nu > 0");
    h->inv_sigma2 = params[0];
    h->lik_param2 = params[1];
    h->sigma_dirty = 1;
  } else if (n_params != 0) {
    return fail(PGB_E_INVALID, "this family has no parameters");
  }
  return PGB_OK;
}

static int enqueue_slots(pgb_handle* h, int count) {
  Dev& d = h->d;
  long long want = (long long)d.nchunks * (d.P - 1);
  if (want < d.n_pad / BT) want = d.n_pad / BT;
  if (want > h->rows_grid) want = h->rows_grid;
  dim3 gctrl((unsigned)(d.P - 1)), grows((unsigned)want);
  long long wantl = (long long)d.nchunks * (d.P - 1);
  if (wantl > h->ll_grid) wantl = h->ll_grid;
  dim3 gll((unsigned)wantl);
  for (int i = 0; i < count; ++i) {
    int par = (int)(h->slot & 1);
    const bool lin = d.response != PGB_RESPONSE_CONSTANT;
    if (d.K > 1 && lin)
      hipLaunchKernelGGL((k_ctrl<true, true>), gctrl, dim3(BT), 0, h->stream, (const Dev*)h->d_dev, par, d.ctrl, (const InitAcc*)d.initacc);
    else if (d.K > 1)
      hipLaunchKernelGGL((k_ctrl<true, false>), gctrl, dim3(BT), 0, h->stream, (const Dev*)h->d_dev, par, d.ctrl, (const InitAcc*)d.initacc);
    else
      if (d.response != PGB_RESPONSE_CONSTANT)
        hipLaunchKernelGGL((k_ctrl<false, true>), gctrl, dim3(BT), 0, h->stream, (const Dev*)h->d_dev, par, d.ctrl, (const InitAcc*)d.initacc);
      else
        hipLaunchKernelGGL((k_ctrl<false, false>), gctrl, dim3(BT), 0, h->stream, (const Dev*)h->d_dev, par, d.ctrl, (const InitAcc*)d.initacc);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->prof) {
      if (h->ev_used + 2 > h->ev.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)
          return fail(PGB_E_DEVICE, "hipEventCreate");
        h->ev.push_back(a);
        h->ev.push_back(b);
      }
      e0 = h->ev[h->ev_used];
      e1 = h->ev[h->ev_used + 1];
      h->ev_used += 2;
    }
    // Profiling: the events are attached to the dispatch itself (hipExtLaunchKernelGGL), i.e. they
    // carry the start / end timestamps of the kernel's own AQL packet -- the interval rocprofv3
    // reports -- rather than bracketing the launch with two extra barrier packets.
#define LAUNCH_ROWS(KERN, ...)                                                                    \
  do {                                                                                            \
    if (h->prof) hipExtLaunchKernelGGL((KERN), grows, dim3(BT), 0, h->stream, e0, e1, 0, (const Dev*)h->d_dev, par, ##__VA_ARGS__); \
    else hipLaunchKernelGGL((KERN), grows, dim3(BT), 0, h->stream, (const Dev*)h->d_dev, par, ##__VA_ARGS__);    \
  } while (0)
#define ROWS_PTRS (const Cmd*)d.cmd, (const Job*)d.jobs
    if (d.K > 1 && lin) {  // linear leaves: one instance for any K
      LAUNCH_ROWS((k_rows_mk<0, true>));
    } else if (d.K == 2) {
      LAUNCH_ROWS((k_rows_mk<2, false>));
    } else if (d.K == 3) {
      LAUNCH_ROWS((k_rows_mk<3, false>));
    } else if (d.K == 4) {
      LAUNCH_ROWS((k_rows_mk<4, false>));
    } else if (d.K > 1) {
      LAUNCH_ROWS((k_rows_mk<0, false>));
    } else {
      const bool nrm = h->s.family == PGB_FAMILY_NORMAL;
      if (d.response != PGB_RESPONSE_CONSTANT) {
        if (nrm) LAUNCH_ROWS((k_rows<false, true, true>), ROWS_PTRS);
        else LAUNCH_ROWS((k_rows<false, false, true>), ROWS_PTRS);
      } else if (h->has_subset) {
        if (nrm) LAUNCH_ROWS((k_rows<true, true, false>), ROWS_PTRS);
        else LAUNCH_ROWS((k_rows<true, false, false>), ROWS_PTRS);
      } else {
        if (nrm) LAUNCH_ROWS((k_rows<false, true, false>), ROWS_PTRS);
        else LAUNCH_ROWS((k_rows<false, false, false>), ROWS_PTRS);
      }
    }
#undef LAUNCH_ROWS
#undef ROWS_PTRS
    if (d.family != PGB_FAMILY_NORMAL) {  // per-row log-likelihood of the rows this round re-labelled
#define LAUNCH_LL(KT_, FAM_) hipLaunchKernelGGL((k_loglik<KT_, FAM_, false>), gll, dim3(BT), 0, h->stream, h->d_dev, par)
      if (d.K > 1 && lin) {
        hipLaunchKernelGGL((k_loglik<0, -1, true>), gll, dim3(BT), 0, h->stream, h->d_dev, par);
      } else if (d.K > 1) {
        switch (d.K) {
          case 2: LAUNCH_LL(2, -1); break;
          case 3: LAUNCH_LL(3, -1); break;
          case 4: LAUNCH_LL(4, -1); break;
          default: LAUNCH_LL(0, -1);
        }
      } else if (d.response != PGB_RESPONSE_CONSTANT) {  // linear leaves: one instance, family read at run time
        hipLaunchKernelGGL((k_loglik<1, -1, true>), gll, dim3(BT), 0, h->stream, h->d_dev, par);
      } else {
        switch (d.family) {
          case PGB_FAMILY_BERNOULLI_PROBIT: LAUNCH_LL(1, PGB_FAMILY_BERNOULLI_PROBIT); break;
          case PGB_FAMILY_BERNOULLI_LOGIT: LAUNCH_LL(1, PGB_FAMILY_BERNOULLI_LOGIT); break;
          case PGB_FAMILY_POISSON_LOG: LAUNCH_LL(1, PGB_FAMILY_POISSON_LOG); break;
          case PGB_FAMILY_NEGBIN_LOG: LAUNCH_LL(1, PGB_FAMILY_NEGBIN_LOG); break;
          case PGB_FAMILY_ASYMLAPLACE: LAUNCH_LL(1, PGB_FAMILY_ASYMLAPLACE); break;
          case PGB_FAMILY_GAMMA_LOG: LAUNCH_LL(1, PGB_FAMILY_GAMMA_LOG); break;
          default: LAUNCH_LL(1, PGB_FAMILY_STUDENT_T);
        }
      }
#undef LAUNCH_LL
    }
    h->slot += 1;
  }
  HIPCHK(hipGetLastError());
  return PGB_OK;
}

static int harvest_profile(pgb_handle* h) {
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
    h->prof_ms += ms;
    h->prof_launches += 1;
  }
  h->ev_used = 0;
  return PGB_OK;
}

// Enqueue bundles of slots until the device reports that all requested asteps are complete.
// The device publishes its progress in a pinned host word (k_ctrl, final slot of a step); the
// host polls it between bundles -- no stream synchronisation inside a step.  At most 3 bundles
// are in flight, so the overshoot after completion is bounded (idle slots cost ~2 x 1.3 us).
#define BUNDLE 8
static int run_until_idle(pgb_handle* h, int n_steps) {
  Dev& d = h->d;
  long long start = h->slot;
  long long cap = start + (long long)n_steps * (PGB_MAX_NODES + 3) * (d.m + 1) + 64;
  int rc;
  while (*h->flag < (unsigned long long)h->steps_target) {
    hipEvent_t ev = h->bundle_ev[h->bundles & 3];
    if (h->bundles >= 3) {
      // wait for bundle (bundles - 3) before reusing its event: keeps <= 3 bundles queued
      HIPCHK(hipEventSynchronize(h->bundle_ev[(h->bundles - 3) & 3]));
      if (*h->flag >= (unsigned long long)h->steps_target) break;
    }
    if ((rc = enqueue_slots(h, BUNDLE)) != PGB_OK) return rc;
    HIPCHK(hipEventRecord(ev, h->stream));
    h->bundles += 1;
    if (h->slot > cap) return fail(PGB_E_STATE, "sampler state machine did not finish");
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->prof && (rc = harvest_profile(h)) != PGB_OK) return rc;
  Ctrl c;
  HIPCHK(hipMemcpy(&c, &d.ctrl[h->slot & 1], sizeof c, hipMemcpyDeviceToHost));
  if (c.phase != PH_IDLE) return fail(PGB_E_STATE, "device not idle after the progress flag fired");
  h->st_cur = c.st_cur;
  h->alpha_cur = c.alpha_cur;
  return PGB_OK;
}

static int begin_steps(pgb_handle* h, int tune, int n_steps) {
  Dev& d = h->d;
  if (!h->have_data || !h->have_y) return fail(PGB_E_INVALID, "set_data/set_response first");
  int par = (int)(h->slot & 1);
  hipLaunchKernelGGL(k_begin, dim3(1), dim3(256), 0, h->stream, h->d_dev, par, tune, n_steps,
                     h->inv_sigma2, h->lik_param2, h->sigma_dirty);
  h->steps_target += n_steps;
  h->sigma_dirty = 0;
  // host mirror of the batch cursor ([U] PGBART.astep batching)
  for (int i = 0; i < n_steps; ++i) {
    int bs = tune ? d.batch_tune : d.batch_draw;
    int upper = h->lower_host + bs;
    if (upper > d.m) upper = d.m;
    h->last_lower = h->lower_host;
    h->last_n = upper - h->lower_host;
    h->lower_host = upper < d.m ? upper : 0;
  }
  return PGB_OK;
}

static int fetch_counters(pgb_handle* h, pgb_counters* out) {
  unsigned long long c[8];
  HIPCHK(hipMemcpyAsync(c, h->d.counters, sizeof c, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->ctr.particle_steps = (int64_t)c[0];
  h->ctr.tree_updates = (int64_t)c[1];
  h->ctr.rows_touched = (int64_t)c[2];
  h->ctr.rounds = (int64_t)c[3];
  h->ctr.saturations = (int64_t)c[4];
  h->ctr.slots = (int64_t)c[5];
  if (out) *out = h->ctr;
  return PGB_OK;
}

extern "C" int pgb_step(pgb_handle* h, int32_t tune, double* sum_trees_dev_out, int32_t* vi_counts_host_out,
                        pgb_counters* counters_out) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  int rc;
  if ((rc = begin_steps(h, tune, 1)) != PGB_OK) return rc;
  if ((rc = run_until_idle(h, 1)) != PGB_OK) return rc;
  if (sum_trees_dev_out)  // [K][n] out of the padded [K][n_pad] buffer
    HIPCHK(hipMemcpy2DAsync(sum_trees_dev_out, h->d.n * sizeof(double),
                            h->d.st + (size_t)h->st_cur * h->d.K * h->d.n_pad, h->d.n_pad * sizeof(double),
                            h->d.n * sizeof(double), (size_t)h->d.K, hipMemcpyDeviceToDevice, h->stream));
  if (vi_counts_host_out)
    HIPCHK(hipMemcpyAsync(vi_counts_host_out, h->d.vi, h->d.p * sizeof(int32_t), hipMemcpyDeviceToHost,
                          h->stream));
  if ((rc = fetch_counters(h, counters_out)) != PGB_OK) return rc;
  return PGB_OK;
}

extern "C" int pgb_step_async(pgb_handle* h, int32_t tune, int32_t n_steps) {
  if (!h || n_steps < 1) return fail(PGB_E_INVALID, "bad argument");
  int rc;
  if ((rc = begin_steps(h, tune, n_steps)) != PGB_OK) return rc;
  return run_until_idle(h, n_steps);
}

extern "C" int pgb_sync(pgb_handle* h, pgb_counters* counters_out) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  return fetch_counters(h, counters_out);
}

extern "C" int pgb_export_trees(pgb_handle* h, int32_t which, pgb_tree_arrays* out) {
  if (!h || !out) return fail(PGB_E_INVALID, "null argument");
  Dev& d = h->d;
  int first = which == 0 ? h->last_lower : 0;
  int nt = which == 0 ? h->last_n : d.m;
  std::vector<DTree> host(nt);
  if (nt > 0) {
    HIPCHK(hipMemcpyAsync(host.data(), d.trees + first, sizeof(DTree) * nt, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  int total = 0;
  for (int t = 0; t < nt; ++t) total += host[t].n_nodes;
  const int K = d.K, KX = d.K - 1;
  if (!out->var) {
    out->n_trees = nt;
    out->n_outputs = K;
    out->total_nodes = total;
    return PGB_OK;
  }
  std::vector<double> hx;
  if (KX > 0 && nt > 0) {
    hx.resize((size_t)nt * MAXN * KX);
    HIPCHK(hipMemcpy(hx.data(), d.tvx + (size_t)first * MAXN * KX, hx.size() * sizeof(double),
                     hipMemcpyDeviceToHost));
  }
  if (out->n_trees != nt || out->total_nodes != total) return fail(PGB_E_INVALID, "size mismatch");
  std::vector<LinP> hl;
  std::vector<double> hsx;  // slopes of outputs 1..K-1
  const bool want_lin = out->slope && out->xbar && out->svar;
  if (want_lin && d.response != PGB_RESPONSE_CONSTANT && nt > 0) {
    hl.resize((size_t)nt * MAXN);
    HIPCHK(hipMemcpy(hl.data(), d.tlin + (size_t)first * MAXN, hl.size() * sizeof(LinP), hipMemcpyDeviceToHost));
    if (KX > 0) {
      hsx.resize((size_t)nt * MAXN * KX);
      HIPCHK(hipMemcpy(hsx.data(), d.tsx + (size_t)first * MAXN * KX, hsx.size() * sizeof(double),
                       hipMemcpyDeviceToHost));
    }
  }
  int off = 0;
  for (int t = 0; t < nt; ++t) {
    const DTree& T = host[t];
    out->tree_id[t] = first + t;
    out->node_off[t] = off;
    for (int k = 0; k < T.n_nodes; ++k) {
      const DNode& z = T.nd[k];
      out->var[off + k] = z.var;
      out->split[off + k] = z.var >= 0 ? z.split : 0.0;
      out->left[off + k] = z.var >= 0 ? (int32_t)z.left : -1;
      out->right[off + k] = z.var >= 0 ? (int32_t)z.right : -1;
      out->count[off + k] = z.cnt;
      out->value[(size_t)(off + k) * K] = z.var < 0 ? z.value : 0.0;
      if (want_lin) {
        const bool islin = z.var < 0 && !hl.empty() && hl[(size_t)t * MAXN + k].svar >= 0;
        out->slope[(size_t)(off + k) * K] = islin ? hl[(size_t)t * MAXN + k].slope : 0.0;
        for (int o = 1; o < K; ++o)
          out->slope[(size_t)(off + k) * K + o] = islin ? hsx[((size_t)t * MAXN + k) * KX + o - 1] : 0.0;
        out->xbar[off + k] = islin ? hl[(size_t)t * MAXN + k].xbar : 0.0;
        out->svar[off + k] = islin ? (int32_t)hl[(size_t)t * MAXN + k].svar : -1;
      }
      for (int o = 1; o < K; ++o)
        out->value[(size_t)(off + k) * K + o] = z.var < 0 ? hx[((size_t)t * MAXN + k) * KX + o - 1] : 0.0;
    }
    off += T.n_nodes;
  }
  out->node_off[nt] = off;
  return PGB_OK;
}

extern "C" int pgb_get_state(pgb_handle* h, double* leaf_sd_out, int64_t* iter_out, int32_t* lower_out) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  Dev& d = h->d;
  Ctrl c;
  InitAcc ia[IA_SLOTS];
  HIPCHK(hipMemcpyAsync(&c, &d.ctrl[h->slot & 1], sizeof c, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(ia, &d.initacc[(size_t)((h->slot & 1) ^ 1) * IA_SLOTS], sizeof ia, hipMemcpyDeviceToHost,
                        h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  double leaf_sd = c.leaf_sd;
  long long qstd = 0;
  for (int k = 0; k < IA_SLOTS; ++k) qstd += ia[k].QSTD;
  if (c.pend_leafsd && c.pend_iter > 2) leaf_sd = ((double)qstd * d.sc.inv_c1) / (double)d.n;
  if (leaf_sd_out) {
    leaf_sd_out[0] = leaf_sd;
    const int KX = d.K - 1;
    if (KX > 0) {
      std::vector<long long> ix((size_t)IA_SLOTS * 2 * KX);
      HIPCHK(hipMemcpy(ix.data(), d.iax + (size_t)((h->slot & 1) ^ 1) * IA_SLOTS * 2 * KX,
                       ix.size() * sizeof(long long), hipMemcpyDeviceToHost));
      double lsdx[2 * KXMAX];
      HIPCHK(hipMemcpy(lsdx, d.lsdx, sizeof lsdx, hipMemcpyDeviceToHost));
      for (int k = 0; k < KX; ++k) {
        double v = lsdx[(h->slot & 1) * KXMAX + k];
        if (c.pend_leafsd && c.pend_iter > 2) {
          long long q = 0;
          for (int sl = 0; sl < IA_SLOTS; ++sl) q += ix[(size_t)sl * 2 * KX + KX + k];
          v = ((double)q * d.sc.inv_c1) / (double)d.n;
        }
        leaf_sd_out[k + 1] = v;
      }
    }
  }
  if (iter_out) *iter_out = c.iter;
  if (lower_out) *lower_out = c.lower;
  return PGB_OK;
}

extern "C" int pgb_get_split_weights(pgb_handle* h, double* out) {
  if (!h || !out) return fail(PGB_E_INVALID, "null argument");
  std::vector<long long> a((size_t)h->d.p);
  HIPCHK(hipMemcpyAsync(a.data(), h->d.alpha + (size_t)h->alpha_cur * h->d.p, h->d.p * sizeof(long long),
                        hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  // in units of the caller's prior: prior_j + number of tuning counts (up to 2^-24 rounding)
  for (int j = 0; j < h->d.p; ++j) out[j] = (double)a[j] * (h->d.max_prior * pgb_pow2(-PGB_ALPHA_BITS));
  return PGB_OK;
}

extern "C" int pgb_predict(const pgb_tree_arrays* trees, const int32_t* forest_tree_idx, int32_t n_forests,
                           int32_t m, const double* X_dev, int64_t n_rows, int32_t p, int64_t ldx,
                           const int32_t* rules_host, const int32_t* excluded_host, int32_t n_excluded,
                           double* out_dev, void* stream) {
  if (!trees || !forest_tree_idx || !X_dev || !out_dev || !rules_host)
    return fail(PGB_E_INVALID, "null argument");
  if (trees->n_outputs < 1 || trees->n_outputs > PGB_MAX_OUTPUTS) return fail(PGB_E_INVALID, "n_outputs");
  if (n_forests < 1 || n_rows < 1) return PGB_OK;
  hipStream_t sm = (hipStream_t)stream;
  const int K = trees->n_outputs, N = trees->total_nodes, NT = trees->n_trees;
  std::vector<uint8_t> excl((size_t)p, 0);
  for (int e = 0; e < n_excluded; ++e)
    if (excluded_host[e] >= 0 && excluded_host[e] < p) excl[excluded_host[e]] = 1;
  // one upload buffer: [node_off | var | left | right | rules | fidx] int32, then 8-byte arrays
  size_t n_i32 = (size_t)(NT + 1) + 3 * (size_t)N + p + (size_t)n_forests * m;
  size_t off8 = ((n_i32 * 4 + 7) / 8) * 8;
  const bool lin = trees->slope && trees->xbar && trees->svar;
  // ... then, for linear leaves, slope / xbar (8-byte) and svar (int32) after the exclusion flags
  const size_t off_lin = ((off8 + (size_t)N * 8 /*split*/ + (size_t)N * 8 /*count*/ + (size_t)N * K * 8 + p + 7) / 8) * 8;
  size_t bytes = lin ? off_lin + (size_t)N * K * 8 + (size_t)N * 12 : off8 + (size_t)N * 16 + (size_t)N * K * 8 + p;
  std::vector<uint8_t> hb(bytes);
  int32_t* hi = (int32_t*)hb.data();
  size_t o = 0;
  memcpy(hi + o, trees->node_off, (NT + 1) * 4); size_t o_off = o; o += NT + 1;
  memcpy(hi + o, trees->var, N * 4); size_t o_var = o; o += N;
  memcpy(hi + o, trees->left, N * 4); size_t o_l = o; o += N;
  memcpy(hi + o, trees->right, N * 4); size_t o_r = o; o += N;
  memcpy(hi + o, rules_host, p * 4); size_t o_rules = o; o += p;
  memcpy(hi + o, forest_tree_idx, (size_t)n_forests * m * 4); size_t o_f = o; o += (size_t)n_forests * m;
  uint8_t* h8 = hb.data() + off8;
  memcpy(h8, trees->split, (size_t)N * 8);
  memcpy(h8 + (size_t)N * 8, trees->count, (size_t)N * 8);
  memcpy(h8 + (size_t)N * 16, trees->value, (size_t)N * K * 8);
  memcpy(h8 + (size_t)N * 16 + (size_t)N * K * 8, excl.data(), p);
  if (lin) {
    memcpy(hb.data() + off_lin, trees->slope, (size_t)N * K * 8);
    memcpy(hb.data() + off_lin + (size_t)N * K * 8, trees->xbar, (size_t)N * 8);
    memcpy(hb.data() + off_lin + (size_t)N * K * 8 + (size_t)N * 8, trees->svar, (size_t)N * 4);
  }
  uint8_t* db = nullptr;
  HIPCHK(hipMalloc((void**)&db, bytes));
  hipError_t e = hipMemcpyAsync(db, hb.data(), bytes, hipMemcpyHostToDevice, sm);
  if (e != hipSuccess) { (void)hipFree(db); return fail_hip(e, "hipMemcpyAsync"); }
  const int32_t* di = (const int32_t*)db;
  PredTrees T;
  T.node_off = di + o_off;
  T.var = di + o_var;
  T.left = di + o_l;
  T.right = di + o_r;
  T.split = (const double*)(db + off8);
  T.count = (const long long*)(db + off8 + (size_t)N * 8);
  T.value = (const double*)(db + off8 + (size_t)N * 16);
  T.slope = lin ? (const double*)(db + off_lin) : nullptr;
  T.xbar = lin ? (const double*)(db + off_lin + (size_t)N * K * 8) : nullptr;
  T.svar = lin ? (const int32_t*)(db + off_lin + (size_t)N * K * 8 + (size_t)N * 8) : nullptr;
  const uint8_t* dexcl = db + off8 + (size_t)N * 16 + (size_t)N * K * 8;
  dim3 grid((unsigned)((n_rows + BT - 1) / BT), (unsigned)n_forests);
  hipLaunchKernelGGL(k_predict, grid, dim3(BT), 0, sm, T, di + o_f, n_forests, m, K, X_dev,
                     (long long)n_rows, p, (long long)ldx, di + o_rules, dexcl, out_dev);
  e = hipGetLastError();
  hipError_t e2 = hipStreamSynchronize(sm);
  (void)hipFree(db);
  if (e != hipSuccess) return fail_hip(e, "k_predict launch");
  if (e2 != hipSuccess) return fail_hip(e2, "k_predict");
  return PGB_OK;
}

// ---- checkpoint / resume ------------------------------------------------------------------
struct CkptHeader {
  char magic[8];       // "PGBCKPT1"
  char backend[16];    // pgb_backend_name()
  pgb_settings s;      // must equal the loading handle's settings
  long long n_allocs, payload_bytes;
  // host mirrors at the idle point
  long long slot, steps_target, flag;
  int32_t st_cur, alpha_cur, lower_host, last_lower, last_n, sigma_dirty;
  double inv_sigma2, lik_param2;
  pgb_counters ctr;
};

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
  *bytes_out = (int64_t)sizeof(CkptHeader) + ckpt_payload(h, nullptr);
  return PGB_OK;
}

extern "C" int pgb_checkpoint_save(pgb_handle* h, void* host_buf, int64_t bytes) {
  if (!h || !host_buf) return fail(PGB_E_INVALID, "null argument");
  if (!h->have_data || !h->have_y) return fail(PGB_E_INVALID, "set_data/set_response first");
  CkptHeader hd;
  memset(&hd, 0, sizeof hd);
  memcpy(hd.magic, "PGBCKPT1", 8);
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
  if (!h->have_data || !h->have_y) return fail(PGB_E_INVALID, "set_data/set_response first");
  if (bytes < (int64_t)sizeof(CkptHeader)) return fail(PGB_E_INVALID, "checkpoint truncated");
  CkptHeader hd;
  memcpy(&hd, host_buf, sizeof hd);
  if (memcmp(hd.magic, "PGBCKPT1", 8) != 0) return fail(PGB_E_INVALID, "not a pgbart checkpoint");
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
  *h->flag = (unsigned long long)hd.flag;
  h->st_cur = hd.st_cur;
  h->alpha_cur = hd.alpha_cur;
  h->lower_host = hd.lower_host;
  h->last_lower = hd.last_lower;
  h->last_n = hd.last_n;
  h->sigma_dirty = hd.sigma_dirty;
  h->inv_sigma2 = hd.inv_sigma2;
  h->lik_param2 = hd.lik_param2;
  h->ctr = hd.ctr;
  return PGB_OK;
}

extern "C" int pgb_profile(pgb_handle* h, int32_t enable, double* kernel_ms_out, int64_t* launches_out) {
  if (!h) return fail(PGB_E_INVALID, "null handle");
  if (kernel_ms_out) *kernel_ms_out = h->prof_ms;
  if (launches_out) *launches_out = h->prof_launches;
  if (enable && !h->prof) {
    h->prof_ms = 0.0;
    h->prof_launches = 0;
    h->ev_used = 0;
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
  if (kernel_ms_out) *kernel_ms_out = h->prof_clock_ms;
  if (launches_out) *launches_out = h->prof_clock_launches;
  return PGB_OK;
}

#ifdef PGB_TRACE
extern "C" int pgb_debug_trace(pgb_handle* h, long long* out, int n_slots) {
  HIPCHK(hipMemcpy(out, h->d.trace, (size_t)n_slots * 16 * sizeof(long long), hipMemcpyDeviceToHost));
  return PGB_OK;
}
#endif
