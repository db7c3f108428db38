// pgb_dev_helpers.h -- part of pgbart_hip.hip (not a standalone header): wave / workgroup primitives and the label -> value table builders.
// ---- global address space (see DevG in pgb_dev_types.h) ----------------------------------------------------
template <typename T>
__device__ __forceinline__ gptr<T> as_global(T* p) { return (gptr<T>)p; }
#if defined(__HIP_DEVICE_COMPILE__)
template <typename T>
__device__ __forceinline__ gptr<T> as_global(gptr<T> p) { return p; }
#endif
template <typename U, typename T>
__device__ __forceinline__ gptr<U> gcast(gptr<T> p) { return (gptr<U>)p; }
// (HIP's double2 / float4 are classes whose copy constructors take generic references: wide loads and stores
//  through a global pointer go through the native vector types)
typedef double pgb_v2f64 __attribute__((ext_vector_type(2)));
typedef float pgb_v4f32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double2 gload_d2(gptr<const double> p) {
  const pgb_v2f64 v = *(gptr<const pgb_v2f64>)p;
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ float4 gload_f4(gptr<const float> p) {
  const pgb_v4f32 v = *(gptr<const pgb_v4f32>)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gstore_d2(gptr<double> p, double a, double b) {
  pgb_v2f64 v;
  v.x = a;
  v.y = b;
  *(gptr<pgb_v2f64>)p = v;
}

// a double at a wave-uniform base + a 32-bit BYTE offset: `global_load_dwordx2 v, v_off, s[base:base+1]` -- no
// 64-bit address computation per load (base[i] with a 32-bit i cannot be selected that way: 8 i may overflow)
__device__ __forceinline__ double gload_d_off(gptr<const double> base, uint32_t byte_off) {
  return *(gptr<const double>)((gptr<const char>)base + byte_off);
}

// a 32-bit word at a wave-uniform base + a 32-bit byte offset (see gload_d_off)
__device__ __forceinline__ uint32_t gload_u32_off(gptr<const uint8_t> base, uint32_t byte_off) {
  return *(gptr<const uint32_t>)(base + byte_off);
}
// Values every lane of the wave holds alike (a record read from LDS by all lanes), moved to scalar registers: what
// is computed from them -- 64-bit offsets, base addresses -- then runs on the scalar unit instead of once per lane.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ long long uni(long long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ double uni(double v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane(__double2loint(v));
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double((int)hi, (int)lo);
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

// The same for TWO values (the per-row families reduce {all rows, left rows} per particle): the first exchange
// halves the live values, five more finish -- 30 vector instructions.  Returns, in every lane, the wave total of
// value number (lane & 1).
__device__ __forceinline__ long long wave_sum2(long long v0, long long v1) {
  const int lane = (int)(threadIdx.x & 63);
  const bool b0 = (lane & 1) != 0;
  long long k = b0 ? v1 : v0;
  const long long s = b0 ? v0 : v1;
  k += dpp_mov64<0xB1>(s);  // quad_perm [1,0,3,2]: lane l now carries value (l & 1) of its pair
  k += dpp_mov64<0x4E>(k);  // quad_perm [2,3,0,1]
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

