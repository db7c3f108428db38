/*
 * pgbart_spec.h -- the NUMERIC CONTRACT of the PGBART C ABI (include/pgbart.h).
 *
 * Every backend that implements the ABI (the gfx950 HIP library, and the CPU
 * restatement under oracle/ that checks it) must produce the same draws for the
 * same (seed, inputs).  That is only possible when the random numbers, the
 * transcendental functions and the reductions are defined independently of
 * execution order.  This header is that definition:
 *
 *   1. RNG       counter-based Philox4x32-10 (Salmon et al., SC'11).  A draw is a
 *                pure function of (seed, tree-update counter, SMC round, particle,
 *                purpose) -- replaces the sequential NumPy stream of upstream
 *                PGBART (SURVEY.md Appendix A "RNG").
 *   2. math      exp / log / sincos / Box-Muller written with + - * / sqrt only
 *                (all IEEE-754 correctly rounded on x86-64 SSE2 and on gfx950),
 *                so host and device results are bit-identical when both sides are
 *                compiled with -ffp-contract=off.
 *   3. sums      every reduction over rows is an integer sum of fixed-point
 *                quantised terms (pgb_quant).  Integer addition is associative,
 *                so a sum does not depend on thread/block/atomic order.
 *
 * The file is plain C99 and also compiles as HIP device code.
 * No reference source corresponds to this file: the reference delegates the
 * sampler to the external `bartrs` wheel (requirements.txt:6).
 */
#ifndef PGBART_SPEC_H
#define PGBART_SPEC_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PGB_HD __host__ __device__ static inline
#else
#define PGB_HD static inline
#endif

/* Both compilers MUST be run with -ffp-contract=off (see __graft_entry__.build): a product and
 * a sum are fused only where the text says so, through PGB_FMA -- one correctly rounded
 * operation on either side (C99 fma / v_fma_f64), used in the Horner chains. */
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
#if defined(__HIPCC__)
#define PGB_FMA(a, b, c) __builtin_fma((a), (b), (c))
#else
#include <math.h>
#define PGB_FMA(a, b, c) fma((a), (b), (c))
#endif

/* ------------------------------------------------------------------ limits */
#define PGB_MAX_NODES 255     /* nodes per tree (127 splits + 128 leaves)        */
#define PGB_MAX_LEAVES 128
#define PGB_ORPHAN 255        /* leaf label of rows dropped by a NaN split value */
#define PGB_MAX_DEPTH 64      /* prior_leaf table length; deeper => never split  */
#define PGB_MAX_PARTICLES 64  /* one particle per lane of a wave64               */
#define PGB_MAX_OUTPUTS 8
#define PGB_SELECT_TRIES 16   /* redraws of the split row when X[row,var] is NaN */

/* split rules (reference names: tests/test_bart.py:143-145, bart.py:100-103) */
#define PGB_RULE_CONTINUOUS 0 /* go left iff x <= v  */
#define PGB_RULE_ONEHOT 1     /* go left iff x == v  */
#define PGB_RULE_SUBSET 2     /* go left iff category x is in the set v (bart.py:100-103)   */
#define PGB_SUBSET_BITS 52    /* categories are integer codes 0..51; the set is a bit mask  */

/* likelihood families (closed family; SURVEY.md 7 "Hard parts") */
#define PGB_FAMILY_NORMAL 0           /* y ~ N(mu, sigma)      params: sigma */
#define PGB_FAMILY_BERNOULLI_PROBIT 1 /* y ~ Bern(Phi(mu))                   */
#define PGB_FAMILY_BERNOULLI_LOGIT 2  /* y ~ Bern(expit(mu))                 */
#define PGB_FAMILY_CATEGORICAL 3      /* y ~ Cat(softmax(mu[0..K-1]))        */
#define PGB_FAMILY_NORMAL_MEANSCALE 4 /* y ~ N(mu[0], |mu[1]|), K = 2        */

/* RNG purposes (high half of counter word 3) */
#define PGB_RNG_PROPOSE 1u  /* u0: prior coin, u1: split variable            */
#define PGB_RNG_SELECT 2u   /* u0: split row (sub = retry index)             */
#define PGB_RNG_LEAF 3u     /* Box-Muller pair -> (left, right) leaf noise   */
#define PGB_RNG_RESAMPLE 4u /* u0: systematic-resampling offset              */
#define PGB_RNG_FINAL 5u    /* u0: final particle choice                     */

/* ------------------------------------------------------------------ Philox */
typedef struct {
  uint32_t v[4];
} pgb_u32x4;

PGB_HD pgb_u32x4 pgb_philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                                   uint32_t c2, uint32_t c3) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  pgb_u32x4 o;
  o.v[0] = c0;
  o.v[1] = c1;
  o.v[2] = c2;
  o.v[3] = c3;
  return o;
}

/* 53-bit uniform in [0,1) from two 32-bit words */
PGB_HD double pgb_u01(uint32_t hi, uint32_t lo) {
  uint64_t x = (((uint64_t)hi << 32) | lo) >> 11;
  return (double)x * 1.1102230246251565404e-16; /* 2^-53 */
}

typedef struct {
  double u0, u1;
} pgb_u2;

/* The draw addressed by (iter, round, particle, purpose, sub) under `seed`. */
PGB_HD pgb_u2 pgb_draw2(uint64_t seed, uint32_t iter, uint32_t round, uint32_t particle,
                        uint32_t purpose, uint32_t sub) {
  pgb_u32x4 x = pgb_philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), particle, round, iter,
                                  (purpose << 16) | (sub & 0xFFFFu));
  pgb_u2 r;
  r.u0 = pgb_u01(x.v[0], x.v[1]);
  r.u1 = pgb_u01(x.v[2], x.v[3]);
  return r;
}

/* ------------------------------------------------------------------ bit casts */
PGB_HD uint64_t pgb_d2u(double x) {
  uint64_t u;
  memcpy(&u, &x, 8);
  return u;
}
PGB_HD double pgb_u2d(uint64_t u) {
  double x;
  memcpy(&x, &u, 8);
  return x;
}

PGB_HD double pgb_pow2(int e) { return pgb_u2d((uint64_t)(e + 1023) << 52); }

/* ------------------------------------------------------------------ split rules */
/* SubsetSplit: the column holds integer category codes; every non-NaN value maps to a code
 * in [0, 52) (out-of-range values clamp -- callers validate).  The split "value" is the set of
 * categories that go left, stored as the integer bit mask M < 2^52 converted to double (exact).
 *   proposal ([U] SubsetSplitRule.get_split_value: each available category joins the set with
 *   probability 1/2): the category of the uniformly chosen row always goes left, every other
 *   category independently with probability 1/2 (52 bits of the unused second uniform of the
 *   SELECT draw).  If no row of the leaf falls outside the set the grow fails, exactly like a
 *   one-hot split on a leaf with a single category (upstream redraws until the subset is
 *   proper; the difference is a failure probability of 2^(1-k) with k categories present). */
PGB_HD int pgb_subset_code(double x) { return x >= 51.0 ? 51 : (x > 0.0 ? (int)x : 0); }
PGB_HD double pgb_subset_value(double u1, double x) {
  uint64_t M = (uint64_t)(u1 * 4503599627370496.0); /* 2^52 */
  M |= (uint64_t)1 << pgb_subset_code(x);
  return (double)M;
}
/* x is not NaN */
PGB_HD int pgb_go_left(int rule, double x, double v) {
  if (rule == PGB_RULE_CONTINUOUS) return x <= v;
  if (rule == PGB_RULE_ONEHOT) return x == v;
  return (int)(((uint64_t)v >> pgb_subset_code(x)) & 1u);
}

/* ------------------------------------------------------------------ exp */
/* exp(x) for the softmax of particle weights: x is clamped to [-700, 700].
 * Cody-Waite reduction x = k ln2 + r, |r| <= ln2/2, degree-13 Taylor in Horner
 * form, scaling by an exactly constructed 2^k.  ~1 ulp; deterministic. */
PGB_HD double pgb_exp(double x) {
  if (!(x == x)) return x;
  if (x > 700.0) x = 700.0;
  if (x < -700.0) x = -700.0;
  double kf = x * 1.4426950408889634074; /* 1/ln2 */
  kf = (kf >= 0.0) ? (double)(int64_t)(kf + 0.5) : (double)(int64_t)(kf - 0.5);
  double r = (x - kf * 6.93147180369123816490e-01) - kf * 1.90821492927058770002e-10;
  double p = 1.6059043836821613e-10;      /* 1/13! */
  p = PGB_FMA(p, r, 2.08767569878681e-09);       /* 1/12! */
  p = PGB_FMA(p, r, 2.505210838544172e-08);      /* 1/11! */
  p = PGB_FMA(p, r, 2.755731922398589e-07);      /* 1/10! */
  p = PGB_FMA(p, r, 2.7557319223985893e-06);     /* 1/9!  */
  p = PGB_FMA(p, r, 2.48015873015873e-05);       /* 1/8!  */
  p = PGB_FMA(p, r, 1.984126984126984e-04);      /* 1/7!  */
  p = PGB_FMA(p, r, 1.388888888888889e-03);      /* 1/6!  */
  p = PGB_FMA(p, r, 8.333333333333333e-03);      /* 1/5!  */
  p = PGB_FMA(p, r, 4.1666666666666664e-02);     /* 1/4!  */
  p = PGB_FMA(p, r, 1.6666666666666666e-01);     /* 1/3!  */
  p = PGB_FMA(p, r, 0.5);
  p = PGB_FMA(p, r, 1.0);
  p = PGB_FMA(p, r, 1.0);
  int64_t k = (int64_t)kf;
  double scale = pgb_u2d((uint64_t)(k + 1023) << 52);
  return p * scale;
}

/* ------------------------------------------------------------------ log */
/* log(x), x > 0 normal.  x = m 2^e, m in [sqrt(1/2), sqrt(2)); s = (m-1)/(m+1);
 * log m = 2 atanh(s) as an odd series to s^23.  Returns -1e300 for x <= 0. */
PGB_HD double pgb_log(double x) {
  if (!(x == x)) return x;
  if (!(x > 0.0)) return -1.0e300;
  uint64_t b = pgb_d2u(x);
  int64_t e = (int64_t)((b >> 52) & 0x7FF) - 1023;
  if (e == -1023) { /* subnormal: rescale */
    x = x * 4503599627370496.0;
    b = pgb_d2u(x);
    e = (int64_t)((b >> 52) & 0x7FF) - 1023 - 52;
  }
  double m = pgb_u2d((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
  if (m > 1.4142135623730951) {
    m = m * 0.5;
    e += 1;
  }
  double f = m - 1.0;
  double s = f / (2.0 + f);
  double z = s * s;
  double q = 4.3478260869565216e-02;  /* 1/23 */
  q = PGB_FMA(q, z, 4.7619047619047616e-02); /* 1/21 */
  q = PGB_FMA(q, z, 5.2631578947368418e-02); /* 1/19 */
  q = PGB_FMA(q, z, 5.8823529411764705e-02); /* 1/17 */
  q = PGB_FMA(q, z, 6.6666666666666666e-02); /* 1/15 */
  q = PGB_FMA(q, z, 7.6923076923076927e-02); /* 1/13 */
  q = PGB_FMA(q, z, 9.0909090909090912e-02); /* 1/11 */
  q = PGB_FMA(q, z, 1.1111111111111111e-01); /* 1/9  */
  q = PGB_FMA(q, z, 1.4285714285714285e-01); /* 1/7  */
  q = PGB_FMA(q, z, 0.2);                    /* 1/5  */
  q = PGB_FMA(q, z, 3.3333333333333331e-01); /* 1/3  */
  q = q * z;
  double lm = 2.0 * s + (2.0 * s) * q;
  double ef = (double)e;
  return ef * 6.93147180369123816490e-01 + (lm + ef * 1.90821492927058770002e-10);
}

/* ------------------------------------------------------------------ sincos */
/* (sin, cos) of 2*pi*u, u in [0,1).  Octant reduction is exact in binary;
 * |w| <= pi/4 Taylor polynomials. */
PGB_HD void pgb_sincos2pi(double u, double* sn, double* cs) {
  double t = u * 8.0;
  int o = (int)t; /* 0..7 */
  if (o > 7) o = 7;
  double f = t - (double)o;
  double y = (o & 1) ? (f - 1.0) : f; /* (-1, 1) */
  int k = ((o + 1) >> 1) & 3;
  double w = y * 7.85398163397448309616e-01; /* pi/4 */
  double w2 = w * w;
  double s = -8.2206352466243295e-18;  /* -1/19! */
  s = PGB_FMA(s, w2, 2.8114572543455206e-15); /*  1/17! */
  s = PGB_FMA(s, w2, -7.6471637318198164e-13); /* -1/15! */
  s = PGB_FMA(s, w2, 1.6059043836821613e-10); /*  1/13! */
  s = PGB_FMA(s, w2, -2.5052108385441720e-08); /* -1/11! */
  s = PGB_FMA(s, w2, 2.7557319223985893e-06); /*  1/9!  */
  s = PGB_FMA(s, w2, -1.9841269841269841e-04); /* -1/7!  */
  s = PGB_FMA(s, w2, 8.3333333333333332e-03); /*  1/5!  */
  s = PGB_FMA(s, w2, -1.6666666666666666e-01); /* -1/3!  */
  s = w + (w * w2) * s;
  double c = 4.1103176233121648e-19;   /*  1/20! */
  c = PGB_FMA(c, w2, -1.5619206968586225e-16); /* -1/18! */
  c = PGB_FMA(c, w2, 4.7794773323873853e-14); /*  1/16! */
  c = PGB_FMA(c, w2, -1.1470745597729725e-11); /* -1/14! */
  c = PGB_FMA(c, w2, 2.0876756987868100e-09); /*  1/12! */
  c = PGB_FMA(c, w2, -2.7557319223985888e-07); /* -1/10! */
  c = PGB_FMA(c, w2, 2.4801587301587302e-05); /*  1/8!  */
  c = PGB_FMA(c, w2, -1.3888888888888889e-03); /* -1/6!  */
  c = PGB_FMA(c, w2, 4.1666666666666664e-02); /*  1/4!  */
  c = PGB_FMA(c, w2, -0.5);
  c = 1.0 + w2 * c;
  switch (k) {
    case 0: *sn = s; *cs = c; break;
    case 1: *sn = c; *cs = -s; break;
    case 2: *sn = -s; *cs = -c; break;
    default: *sn = -c; *cs = s; break;
  }
}

#if defined(__HIPCC__)
#define PGB_SQRT(x) __builtin_sqrt(x)
#else
#include <math.h>
#define PGB_SQRT(x) sqrt(x)
#endif

/* Box-Muller: two independent N(0,1) from two uniforms in [0,1). */
PGB_HD void pgb_normal2(double u0, double u1, double* z0, double* z1) {
  double rad = PGB_SQRT(-2.0 * pgb_log(1.0 - u0));
  double s, c;
  pgb_sincos2pi(u1, &s, &c);
  *z0 = rad * c;
  *z1 = rad * s;
}

/* ------------------------------------------------------------------ log-likelihoods */
/* log Phi(x) (standard normal CDF), deterministic, one polynomial evaluation for either sign.
 * z = |x|.
 *   x <  0: log Phi(-z) = -z^2/2 - log((z + 2) / 2) + LG(t),  t = z / (z + 2) in [0, 1);
 *           LG(t) = log(Phi(-z) exp(z^2/2) / (1 - t)) is smooth and bounded: 16 degree-8 pieces
 *           on equal intervals of t (table tn).  No exp: exact far into the tail.
 *   x >= 0: log Phi(z) itself, 34 degree-8 pieces of width 1/4 on [0, 8.5) (table tp); 0 beyond
 *           (|log Phi(8.5)| < 1e-17).
 * Both signs share ONE Horner evaluation (the sign only selects the table row and the local
 * variable), so a wave with mixed signs does not run two polynomial paths; the x < 0 lanes add
 * one pgb_log.  Tables: tools/fit_log_ndtr.py (Chebyshev-node interpolation against mpmath at 50
 * digits; absolute error 4e-16 / 5e-16).  Absolute error of the function < 4e-13 against
 * scipy.special.log_ndtr on [-38, 38], < 1e-14 on [-5, 5] (tests/test_spec.py). */
PGB_HD double pgb_log_ndtr(double x) {
  static const double tn[16][9] = {
    {-0x1.6c9c157b3b477p-1, -0x1.3cfd24ebe10dcp-6, -0x1.832b258771687p-12, -0x1.80e3c3bf73935p-19, 0x1.a54e8c8c6458dp-25, 0x1.5ced36570cb4dp-29, 0x1.e850a2f45761bp-35, 0x1.27fb972265d12p-41, -0x1.be282ab840064p-47},
    {-0x1.813061f66fc65p-1, -0x1.55b8b1b075af4p-6, -0x1.93bfa88feec9fp-12, -0x1.3d5e3efe0d003p-19, 0x1.4e8f47c595531p-24, 0x1.bd9517f493ad6p-29, 0x1.0c00759bae9c5p-34, 0x1.bbcf7c479c9a7p-43, -0x1.0f52d0d96db94p-45},
    {-0x1.97581a3384203p-1, -0x1.6f6017fa58958p-6, -0x1.a060ab306a625p-12, -0x1.ae03b3b61fb23p-20, 0x1.e9947f0f4d597p-24, 0x1.111549263dae7p-28, 0x1.056f5877dfcdep-34, -0x1.13dd3453762dfp-41, -0x1.02a416e17fc2ep-44},
    {-0x1.af1fb8758e046p-1, -0x1.89a6156c82328p-6, -0x1.a73d55bd2188dp-12, -0x1.1836047a00f82p-21, 0x1.51575494c458dp-23, 0x1.3d44605db935ap-28, 0x1.8b1d8451af6eep-35, -0x1.dc389c3c2d9bdp-40, -0x1.a78f1e25735a5p-44},
    {-0x1.c88deb30b0f8fp-1, -0x1.a41d8576fe21bp-6, -0x1.a62c41cc2b5dap-12, 0x1.f0df230b82755p-21, 0x1.b903d8af3fa82p-23, 0x1.555e133c3552dp-28, 0x1.4644ec353f7fap-37, -0x1.eebcc79347ac5p-39, -0x1.2bdfaefacbd15p-43},
    {-0x1.e3a16da2176ebp-1, -0x1.be3479a30797bp-6, -0x1.9ac405a6c4a8fp-12, 0x1.734befde729e9p-19, 0x1.1138548614803p-22, 0x1.4467878114807p-28, -0x1.ebe7d9a621d34p-35, -0x1.9c9e2aef53f6ep-38, -0x1.5d03b1f2af4cep-43},
    {-0x1.0027518f7da15p+0, -0x1.d731cf7c91c8ap-6, -0x1.82948032efe68p-12, 0x1.4e174c5d350afp-18, 0x1.3e5b939648324p-22, 0x1.df7a5008f3a87p-29, -0x1.549ba4c9694f2p-33, -0x1.1fb9f1ac835d2p-37, -0x1.108026aec59d1p-43},
    {-0x1.0f3e97b5a9e3ep+0, -0x1.ee37b2e549e3ap-6, -0x1.5b8cc26cd7500p-12, 0x1.f4a6b1c64606cp-18, 0x1.574105d058ee0p-22, 0x1.ee5129443239cp-31, -0x1.32dcceb7dc608p-32, -0x1.420cf1e620bc1p-37, 0x1.756d84f3a9214p-46},
    {-0x1.1f02f935ebe92p+0, -0x1.0127054794a79p-5, -0x1.2491e8b677883p-12, 0x1.4fadb8f8aa814p-17, 0x1.4c3fc9b24574ap-22, -0x1.b5c8698b1b9acp-29, -0x1.b3998b2acb512p-32, -0x1.db649f16cf05ep-38, 0x1.5203b7b230769p-42},
    {-0x1.2f58fe63c7ce1p+0, -0x1.093bc15ced211p-5, -0x1.bc6da6b47a7c3p-13, 0x1.9c3b877ed11ecp-17, 0x1.0eecc44d16018p-22, -0x1.1e7499fcc1f63p-27, -0x1.e8ebe66bed359p-32, 0x1.ddeb4463359d1p-41, 0x1.6c2455d8eeac6p-41},
    {-0x1.401d9877f90aep+0, -0x1.0ee8efdb3f2eep-5, -0x1.16c2a3440412ap-13, 0x1.d2719f49864f5p-17, 0x1.33b30782bf583p-23, -0x1.c873e82bb699dp-27, -0x1.8234436a82fbep-32, 0x1.c95e60bb2dae2p-37, 0x1.ca220c8b49f34p-41},
    {-0x1.512797078b338p+0, -0x1.11dee2cc97cffp-5, -0x1.8c10c84f13cc0p-15, 0x1.e57b88d69aa72p-17, -0x1.a3f038bd98429p-28, -0x1.13e2790cc912dp-26, -0x1.7461bce5c2d07p-34, 0x1.a799355c2ace1p-36, 0x1.0793c4290e0abp-41},
    {-0x1.624a5a19095f5p+0, -0x1.1201f4562e06dp-5, 0x1.401f59852b54ap-15, 0x1.ce4f1880ee6dbp-17, -0x1.6196e1bc902e7p-23, -0x1.005201a637877p-26, 0x1.3073ec9269221p-32, 0x1.b7b1246270d04p-36, -0x1.c9b870813670ep-42},
    {-0x1.7359730352325p+0, -0x1.0f7484e419ec7p-5, 0x1.e58ebfa27d1bep-14, 0x1.90027d7407dfdp-17, -0x1.385207572daa2p-22, -0x1.4fcc281d30515p-27, 0x1.2e016f15cd238p-31, 0x1.9bc579d2bc99cp-37, -0x1.4577afa4c1b24p-40},
    {-0x1.842c751330986p+0, -0x1.0a92539b273dap-5, 0x1.78ca958bd1a6ep-13, 0x1.37e90cd2f2f3ep-17, -0x1.7bc7f202cbd79p-22, -0x1.77bfcab1c6b8fp-29, 0x1.3e68315e452adp-31, -0x1.ef8db05c32455p-38, -0x1.20077de85c726p-40},
    {-0x1.94a201257a7ecp+0, -0x1.03dd32ed393cbp-5, 0x1.dbc659b25a226p-13, 0x1.b06ee41e93671p-18, -0x1.7707827460cf7p-22, 0x1.bd0cc8b3129ffp-29, 0x1.aee567e9c475cp-32, -0x1.3176193e2486ap-36, -0x1.05bb65b475210p-42},
  };
  static const double tp[34][9] = {
    {-0x1.3256172f7f1bep-1, 0x1.70aa14147e559p-4, -0x1.3789f7df5bd50p-8, 0x1.3d21436888496p-14, 0x1.3464f17fbfa71p-20, -0x1.bf1c1c2260febp-28, -0x1.0e61fb5735579p-30, -0x1.b296e7a7e0028p-36, 0x1.3482c461d6327p-43},
    {-0x1.bf2c740535475p-2, 0x1.26a4c52e2ce76p-4, -0x1.180d396af5f9ap-8, 0x1.61df01422a853p-14, 0x1.114d199b40162p-20, -0x1.5c1a53d31f0afp-26, -0x1.6326c23229253p-30, -0x1.30a16f2adb08bp-36, 0x1.d4dfaa4cdd552p-41},
    {-0x1.3ca5e181de6b4p-2, 0x1.c9ce8bd89f145p-5, -0x1.eacddce394bc3p-9, 0x1.7fbbd102ff368p-14, 0x1.8a2f50dc14b01p-21, -0x1.3bc8a3053e538p-25, -0x1.83d5038031323p-30, 0x1.8cd24374cc4f0p-39, 0x1.d897ff6a1f8bap-40},
    {-0x1.b18c203eca063p-3, 0x1.584406b6daa59p-5, -0x1.a0f93ee7be814p-9, 0x1.9146918d4ef73p-14, 0x1.31c17459af12cp-22, -0x1.c41c4248c14d0p-25, -0x1.3ea90c2bc1a1cp-30, 0x1.30cc3788cb556p-35, 0x1.3452ea0d7dfd0p-39},
    {-0x1.1de6f2151f49ap-3, 0x1.f2ee6ff575b83p-6, -0x1.556c6794c7badp-9, 0x1.9151f141d5e42p-14, -0x1.41071b0b5d1efp-22, -0x1.0d20f77a279d4p-24, -0x1.dd86c1aee2a88p-32, 0x1.290df995d79bdp-34, 0x1.e2cd06bb6552ap-40},
    {-0x1.69e8b8a516e7fp-4, 0x1.5acab4beea7e1p-6, -0x1.0bc7e5549dd8ep-9, 0x1.7ca7682d0b659p-14, -0x1.f3ddbbc70c9b4p-21, -0x1.08714d8fb461bp-24, 0x1.66b01761e3ec1p-31, 0x1.6b9c949b129dfp-34, 0x1.9423871260a3cp-54},
    {-0x1.b6295bcbdfd74p-5, 0x1.cc5b5188a23c1p-7, -0x1.8fe8f4abb42bap-10, 0x1.53b65dc4f0174p-14, -0x1.8e9a062d3eed2p-20, -0x1.93f060ce62161p-25, 0x1.db6ea26f17bdap-30, 0x1.1c585879119dfp-34, -0x1.38fff73e216a1p-39},
    {-0x1.f9bd6774eea48p-6, 0x1.2294c953210bdp-7, -0x1.1aba2832bd042p-10, 0x1.1b4ab36dbee57p-14, -0x1.ece0699416bccp-20, -0x1.7a321a4b122b0p-26, 0x1.3f9c9d8c66e78p-29, 0x1.25e70b188cd29p-36, -0x1.ec9a2e5df7a77p-39},
    {-0x1.157a5dee91099p-6, 0x1.5b9dadb9c7363p-8, -0x1.78b79bedb59d8p-11, 0x1.b71fecb427b29p-15, -0x1.0128380705364p-19, 0x1.89afa81c54286p-28, 0x1.2a739d30bfb37p-29, -0x1.4478731412747p-35, -0x1.9647a0b544f86p-39},
    {-0x1.20ca757a34e1fp-7, 0x1.88ed3235e31bfp-9, -0x1.d74fda2b16a4ep-12, 0x1.3b2b6c4a5bd9cp-15, -0x1.d383a8b61622ap-20, 0x1.d9b07730b5b3dp-26, 0x1.7f9db27f3af48p-30, -0x1.2b85413a4e092p-34, -0x1.fb4a2d262068bp-41},
    {-0x1.1c8c5555ca7bfp-8, 0x1.a2c29f114ef41p-10, -0x1.1426387512681p-12, 0x1.a22c5e96a0d52p-16, -0x1.784d8d855b555p-20, 0x1.49f3f23eaf574p-25, 0x1.c98a95b53b75ap-32, -0x1.2310d98580c16p-34, 0x1.1e3a0d3e003bbp-40},
    {-0x1.090d201df5328p-9, 0x1.a427eeab1eea2p-11, -0x1.2ea9184e54e13p-13, 0x1.00548bd4d619ap-16, -0x1.0f184d79b4e52p-20, 0x1.4a112d31cdf7cp-25, -0x1.86eb3c89fbd06p-32, -0x1.6e6b61082fa7ap-35, 0x1.03184b49709ccp-39},
    {-0x1.d250071dfcd5bp-11, 0x1.8c7c635d427dap-12, -0x1.360defc2d200fp-14, 0x1.228ee2d6ca864p-17, -0x1.6080a232f9eefp-21, 0x1.0e72261587a38p-25, -0x1.93a0fa528ad93p-31, -0x1.d19f1a16f3ba8p-37, 0x1.bbffe5411ed43p-40},
    {-0x1.831407d9a7e95p-12, 0x1.5fb2319af6024p-13, -0x1.28dc8c7cb5a72p-15, 0x1.30ed3d54f37edp-18, -0x1.a077a2c63670ep-22, 0x1.7d616cfd4f0f5p-26, -0x1.a5fbd8cbd010ep-31, 0x1.ba9f1d55a8167p-38, 0x1.d8edbbb70b11ap-41},
    {-0x1.2f051a65b3d72p-13, 0x1.2526cf65a8058p-14, -0x1.09b5a982f7387p-16, 0x1.28bc4bdbfa22dp-19, -0x1.c19bd6f9449e3p-23, 0x1.dad0e480145e5p-27, -0x1.4fe619e5a98a9p-31, 0x1.f250f857a0bb4p-37, 0x1.a254a18b03c74p-43},
    {-0x1.bf3a7383e9134p-15, 0x1.cb284b6aaed81p-16, -0x1.bcd578224e819p-18, 0x1.0c307f4661eb7p-20, -0x1.bd99854278de6p-24, 0x1.090f23bbda894p-27, -0x1.bfe47b708375cp-32, 0x1.e909f4d424483p-37, -0x1.7a2fb5c004d4bp-43},
    {-0x1.36ff6bb50c785p-16, 0x1.51d1404fd6a3dp-17, -0x1.5c61881af1878p-19, 0x1.c2e41fb428156p-22, -0x1.96f495830ed93p-25, 0x1.0c29a24e7a25ep-28, -0x1.0458f4d04f923p-32, 0x1.66da0a6c2be6ep-37, -0x1.26fa461c824b9p-42},
    {-0x1.9776056d2b406p-18, 0x1.d2fafe190dfacp-19, -0x1.fec35ae17520ep-21, 0x1.60fa892d34f0ep-23, -0x1.578453c252567p-26, 0x1.ef4ae6a91e86fp-30, -0x1.0d97bcf4fd0f8p-33, 0x1.b6f466d1f4431p-38, -0x1.f0face457d2e0p-43},
    {-0x1.f6c726add6fb5p-20, 0x1.2f3620549181ap-20, -0x1.5e96c246aabf9p-22, 0x1.019ce4d54ba3dp-24, -0x1.0ca6c970ebfdap-27, 0x1.a3b9d86dbcde2p-31, -0x1.f7d3785839cecp-35, 0x1.d348182495df9p-39, -0x1.432b921258c68p-43},
    {-0x1.24149f101743cp-21, 0x1.71e584a210d0ap-22, -0x1.c2cfca5b57716p-24, 0x1.5edf60e9e9b74p-26, -0x1.860f2c6b7ae26p-29, 0x1.47ac5d1986b7dp-32, -0x1.ac7736f830f3ep-36, 0x1.baa1c1bbd3b40p-40, -0x1.624cd541a30dfp-44},
    {-0x1.3f7a8f1d851a2p-23, 0x1.a7e88bb593778p-24, -0x1.0f90fc40b1fa3p-25, 0x1.be4360a91c8a7p-28, -0x1.0741c98cf2ccfp-30, 0x1.d8b9c661e3f71p-34, -0x1.4d97d73fd0fedp-37, 0x1.7a08c0017d66dp-41, -0x1.53cafe93bcf8bp-45},
    {-0x1.48eb8d145ee8ep-25, 0x1.c85f9f230afc8p-26, -0x1.32a03fbbcba82p-27, 0x1.092d95485f3e7p-29, -0x1.4ac7e634a10bcp-32, 0x1.3bcd032c02912p-35, -0x1.dd9dac96830b1p-39, 0x1.25995c414ac93p-42, -0x1.22b8601249114p-46},
    {-0x1.3eb3453d3c8a3p-27, 0x1.cd8ea2eded5d1p-28, -0x1.44884acbe4240p-29, 0x1.26a228f3cf755p-31, -0x1.8348a8fa8ae9ep-34, 0x1.877a911c32df8p-37, -0x1.3b62a89e4bdedp-40, 0x1.a12afe702e527p-44, -0x1.c116a6194cae5p-48},
    {-0x1.2293637cac591p-29, 0x1.b6848b23da0b2p-30, -0x1.42095647af517p-31, 0x1.3231154dcfb55p-33, -0x1.a6e1f17ca2737p-36, 0x1.c2ed40fa86fabp-39, -0x1.812a8f95dd8d3p-42, 0x1.1045e9013d921p-45, -0x1.3b96f2e76ec82p-49},
    {-0x1.f289d488f4b4ap-32, 0x1.8762da1505798p-32, -0x1.2ba7af0dfa534p-33, 0x1.29be7ff351b08p-35, -0x1.aef302855fc04p-38, 0x1.e3351ce109f56p-41, -0x1.b3d6f542b560cp-44, 0x1.4791b32b37495p-47, -0x1.95c90d4507bbfp-51},
    {-0x1.9256fc313e2bdp-34, 0x1.4827ed6585235p-34, -0x1.057fd146dc184p-35, 0x1.0f01b3cf7414dp-37, -0x1.9a1ffe478c8f9p-40, 0x1.e233f96a481aep-43, -0x1.c9b08d2c5c2dep-46, 0x1.6c242c6ca08dcp-49, -0x1.df45bdabe2ebfp-53},
    {-0x1.317156a78b172p-36, 0x1.0278910bce75fp-36, -0x1.ac17b078cf0cap-38, 0x1.cdea9be54260bp-40, -0x1.6cafe247b7aa3p-42, 0x1.c080a7d3642e8p-45, -0x1.bea26b6ebe380p-48, 0x1.76c7fadaf8d6cp-51, -0x1.04cae14c22892p-54},
    {-0x1.b437009ea5552p-39, 0x1.7e7fa1d6c4a83p-39, -0x1.48b5af57bc7acp-40, 0x1.70adac47de6b3p-42, -0x1.2f22ecb80c9b2p-44, 0x1.8517f460a42f2p-47, -0x1.957580b5d94a0p-50, 0x1.65b6c334cda2ep-53, -0x1.062691b004d2bp-56},
    {-0x1.24f60a258d773p-41, 0x1.09df85a841032p-41, -0x1.d996269b9630cp-43, 0x1.13a74f72ca4acp-44, -0x1.d74640ed928c5p-47, 0x1.3b0acee4795f4p-49, -0x1.56bc26efc0f41p-52, 0x1.3d0d65e2f55afp-55, -0x1.e7c0cec960cb4p-59},
    {-0x1.721278ef40c2ap-44, 0x1.5b388401e4e2ap-44, -0x1.40181a30a344bp-45, 0x1.823762373f137p-47, -0x1.56b48db748b03p-49, 0x1.dc6731b57a198p-52, -0x1.0dfad40d75551p-54, 0x1.053fe02422c06p-57, -0x1.a49c251042fccp-61},
    {-0x1.b79cff2b8cae9p-47, 0x1.a9fbf63b12f7ep-47, -0x1.9604278b757b1p-48, 0x1.fb1b25b6aa39ep-50, -0x1.d26ac01335795p-52, 0x1.50866d849d162p-54, -0x1.8ca2fc715b887p-57, 0x1.90a49f0f53758p-60, -0x1.50aaa3e4f9640p-63},
    {-0x1.eb0fed119b109p-50, 0x1.eaf36542b8ca6p-50, -0x1.e347990c39ea9p-51, 0x1.380a22390f9fdp-52, -0x1.291841c7f2d8ap-54, 0x1.bc5d2264506d2p-57, -0x1.0fdb9e8de5302p-59, 0x1.1e1cb4c8b1ecep-62, -0x1.f4d1f404086cbp-66},
    {-0x1.01e30a1d54c78p-52, 0x1.09c5972f5074ep-52, -0x1.0decae900b470p-53, 0x1.67fce4ea8ea40p-55, -0x1.625d50c385d1ep-57, 0x1.124ee1cc13d16p-59, -0x1.5be8a1ae7023cp-62, 0x1.7cdb7df3dbf96p-65, -0x1.5a7c048cf453fp-68},
    {-0x1.fd59ae3f7142ep-56, 0x1.0e5011ed79d14p-55, -0x1.1afbd42aa4a19p-56, 0x1.855e848815345p-58, -0x1.8bd3fed38282ap-60, 0x1.3cc0b1c690573p-62, -0x1.9fd8b74267e10p-65, 0x1.d8cccd4785191p-68, -0x1.be4ce2cee174dp-71},
  };
  if (!(x == x)) return x;
  const int neg = x < 0.0;
  const double z = neg ? -x : x;
  const double zp2 = z + 2.0;
  const double t = z / zp2;
  int in = (int)(t * 16.0);
  if (in > 15) in = 15;
  const double zc = z < 8.5 ? z : 8.5; /* keeps the conversion below in range */
  int ip = (int)(zc * 4.0);
  if (ip > 33) ip = 33;
  const double un = t * 32.0 - (double)(2 * in + 1); /* local variables in [-1, 1] */
  const double up = zc * 8.0 - (double)(2 * ip + 1);
  const double* c = neg ? tn[in] : tp[ip];
  const double u = neg ? un : up;
  double g = c[8];
  g = PGB_FMA(g, u, c[7]);
  g = PGB_FMA(g, u, c[6]);
  g = PGB_FMA(g, u, c[5]);
  g = PGB_FMA(g, u, c[4]);
  g = PGB_FMA(g, u, c[3]);
  g = PGB_FMA(g, u, c[2]);
  g = PGB_FMA(g, u, c[1]);
  g = PGB_FMA(g, u, c[0]);
  if (neg) return -0.5 * (z * z) + (g - pgb_log(0.5 * zp2));
  return z < 8.5 ? g : 0.0;
}

/* log(1 + e^t) */
PGB_HD double pgb_softplus(double t) {
  if (t > 36.0) return t;
  return pgb_log(1.0 + pgb_exp(t));
}

/* Per-row log-likelihood of the closed families with one linear predictor mu (K = 1).
 * y is the observed response (0/1 for the Bernoulli families).  Clamped to [-2047, 0] so that
 * n terms fit the fixed-point accumulator (scale cl). */
PGB_HD double pgb_loglik1(int family, double y, double mu) {
  const double smu = y > 0.5 ? mu : -mu;
  double ll = family == PGB_FAMILY_BERNOULLI_PROBIT ? pgb_log_ndtr(smu) : -pgb_softplus(-smu);
  if (!(ll > -2047.0)) ll = -2047.0;
  if (ll > 0.0) ll = 0.0;
  return ll;
}

/* Categorical-softmax over K linear predictors: mu[y] - logsumexp(mu) (serial max / sum in output
 * order), clamped like pgb_loglik1.  y is the class index stored as a double. */
PGB_HD double pgb_loglik_cat(int K, double y, const double* mu) {
  double mx = mu[0];
  for (int k = 1; k < K; ++k)
    if (mu[k] > mx) mx = mu[k];
  double sum = 0.0;
  for (int k = 0; k < K; ++k) sum += pgb_exp(mu[k] - mx);
  int c = (int)y;
  if (c < 0) c = 0;
  if (c > K - 1) c = K - 1;
  double ll = (mu[c] - mx) - pgb_log(sum);
  if (!(ll > -2047.0)) ll = -2047.0;
  if (ll > 0.0) ll = 0.0;
  return ll;
}

/* Normal with BART mean and BART scale (reference tests/test_bart.py:118: Normal(w[0], |w[1]|)):
 * -log|s| - 0.5 ((y - m)/s)^2 (the constant -0.5 log 2pi cancels in the particle weights).
 * |s| is floored at 1e-8; clamped to [-2047, 2047]. */
PGB_HD double pgb_loglik_meanscale(double y, const double* mu) {
  double sd = mu[1] < 0.0 ? -mu[1] : mu[1];
  if (sd < 1e-8) sd = 1e-8;
  const double z = (y - mu[0]) / sd;
  double ll = -pgb_log(sd) - 0.5 * (z * z);
  if (!(ll > -2047.0)) ll = -2047.0;
  if (ll > 2047.0) ll = 2047.0;
  return ll;
}

/* Per-row log-likelihood of every non-Normal(sigma) family at the K linear predictors mu. */
PGB_HD double pgb_loglik(int family, int K, double y, const double* mu) {
  if (family == PGB_FAMILY_CATEGORICAL) return pgb_loglik_cat(K, y, mu);
  if (family == PGB_FAMILY_NORMAL_MEANSCALE) return pgb_loglik_meanscale(y, mu);
  return pgb_loglik1(family, y, mu[0]);
}

/* ------------------------------------------------------------------ fixed point */
/* q = round-to-nearest-even(x * 2^s) saturated to |q| <= 2^50, via the 1.5*2^52
 * trick (exact for |x*2^s| < 2^51).  `scale` = 2^s.  NaN -> 0.  `sat` (may be
 * NULL) is incremented on saturation/NaN so that backends can report it. */
#define PGB_QLIM 1125899906842624.0 /* 2^50 */
PGB_HD int64_t pgb_quant(double x, double scale, unsigned* sat) {
  double t = x * scale;
  if (!(t == t)) {
    t = 0.0;
    if (sat) *sat += 1u;
  }
  if (t > PGB_QLIM) {
    t = PGB_QLIM;
    if (sat) *sat += 1u;
  }
  if (t < -PGB_QLIM) {
    t = -PGB_QLIM;
    if (sat) *sat += 1u;
  }
  double mg = t + 6755399441055744.0; /* 1.5 * 2^52 */
  return (int64_t)(pgb_d2u(mg) - 0x4338000000000000ull);
}

/* Fixed-point scales derived from the data once (pgb_set_data):
 *   frac = min(61 - ceil(log2(n+1)), 50)   bits below the saturation limit
 *   S1   = frac - range_exp                for sum_trees / residual terms (|x| < 2^range_exp)
 *   S2   = frac - 2*range_exp              for squared residual terms
 *   SL   = frac - 11                       for per-row log-likelihood terms (|x| < 2048)
 * so that n saturated terms still fit an int64.                                  */
typedef struct {
  double c1, c2, cl;             /* 2^S1, 2^S2, 2^SL   */
  double inv_c1, inv_c2, inv_cl; /* 2^-S1, 2^-S2, 2^-SL */
} pgb_scales;

PGB_HD pgb_scales pgb_make_scales(int64_t n, int range_exp) {
  int bits = 0;
  while (((int64_t)1 << bits) < n + 1) ++bits;
  int frac = 61 - bits;
  if (frac > 50) frac = 50;
  pgb_scales s;
  s.c1 = pgb_pow2(frac - range_exp);
  s.c2 = pgb_pow2(frac - 2 * range_exp);
  s.cl = pgb_pow2(frac - 11);
  s.inv_c1 = pgb_pow2(-(frac - range_exp));
  s.inv_c2 = pgb_pow2(-(frac - 2 * range_exp));
  s.inv_cl = pgb_pow2(-(frac - 11));
  return s;
}

/* ------------------------------------------------------------------ split-variable sampler */
/* [U] SampleSplittingVariable.  Split weights are integers: A_j = rne(prior_j * 2^24 / max prior)
 * plus PGB_ALPHA_UNIT-scaled tuning counts, so their prefix sums S_j are exact and independent of
 * summation order (any workgroup can rebuild them in parallel).  A draw u picks the first j with
 * u * S_{p-1} <= S_j (as doubles; S < 2^53), fallback p-1. */
#define PGB_ALPHA_BITS 24
PGB_HD int64_t pgb_alpha_unit(double max_prior) {  /* what one tuning count adds */
  int64_t v = pgb_quant(pgb_pow2(PGB_ALPHA_BITS) / max_prior, 1.0, (unsigned*)0);
  return v < 1 ? 1 : v;
}
PGB_HD int64_t pgb_alpha_init(double prior, double max_prior) {
  int64_t v = pgb_quant(prior * (pgb_pow2(PGB_ALPHA_BITS) / max_prior), 1.0, (unsigned*)0);
  return v < 1 ? 1 : v;
}
PGB_HD int pgb_sample_var(const int64_t* S, int p, double u) {
  const double thr = u * (double)S[p - 1];
  for (int j = 0; j < p; ++j)
    if (thr <= (double)S[j]) return j;
  return p - 1;
}

/* ------------------------------------------------------------------ particle weights */
/* Inclusive scan of 64 doubles in the FIXED association order of a wave64 DPP scan
 * (row_shr:1,2,4,8 inside each row of 16 lanes, then row_bcast:15 into rows 1 and 3, then
 * row_bcast:31 into rows 2 and 3).  The numeric contract defines the cumulative particle
 * weights as THIS scan so that the GPU can use 6 cross-lane steps instead of a 64-step serial
 * chain; a CPU backend evaluates the same tree of additions with this function.  Unused
 * entries must be 0.0 (x + 0.0 is exact). */
PGB_HD void pgb_scan64(double* x) {
  double t[64];
  for (int d = 1; d <= 8; d <<= 1) {
    for (int i = 0; i < 64; ++i) t[i] = ((i & 15) >= d) ? x[i] + x[i - d] : x[i];
    for (int i = 0; i < 64; ++i) x[i] = t[i];
  }
  for (int i = 16; i < 32; ++i) x[i] = x[i] + x[15];
  for (int i = 48; i < 64; ++i) x[i] = x[i] + x[47];
  for (int i = 32; i < 64; ++i) x[i] = x[i] + x[31];
}

/* [U] normalize + inverse_cdf: particles occupy entries [first, first+cnt) of a 64-entry
 * array of log-weights.  w_i = exp(lw_i - max) + 1e-12, W = pgb_scan64(w), total = W[last].
 * pgb_pick returns the first i in [first, last) with !(u * total > W[i]), else last. */
PGB_HD void pgb_weights_scan(const double* lw, int first, int cnt, double* W) {
  double mx = lw[first];
  for (int i = first + 1; i < first + cnt; ++i)
    if (lw[i] > mx) mx = lw[i];
  for (int i = 0; i < 64; ++i) W[i] = 0.0;
  for (int i = first; i < first + cnt; ++i) W[i] = pgb_exp(lw[i] - mx) + 1e-12;
  pgb_scan64(W);
}
PGB_HD int pgb_pick(const double* W, int first, int cnt, double u) {
  const int last = first + cnt - 1;
  const double thr = u * W[last];
  for (int i = first; i < last; ++i)
    if (!(thr > W[i])) return i;
  return last;
}

/* ------------------------------------------------------------------ leaf algebra */
/* Normal family: sum of squared errors of a leaf with value v from the integer
 * sufficient statistics (count, sum r, sum r^2), r = y - sum_trees_noi:
 *   SSE = Q2 - 2 v Q1 + cnt v^2          (evaluated in exactly this order). */
PGB_HD double pgb_leaf_sse(int64_t cnt, int64_t q_r, int64_t q_r2, double v, double inv_c1,
                           double inv_c2) {
  double a = (double)q_r2 * inv_c2;
  double b = (double)q_r * inv_c1;
  return (a - (2.0 * v) * b) + ((double)cnt * v) * v;
}

/* Leaf value: mean of sum_trees over the leaf rows / m + noise (upstream
 * draw_leaf_value); empty leaf -> 0. */
PGB_HD double pgb_leaf_value(int64_t cnt, int64_t q_st, double inv_c1, double m, double z,
                             double leaf_sd) {
  if (cnt <= 0) return 0.0;
  double mean = ((double)q_st * inv_c1) / (double)cnt;
  return mean / m + z * leaf_sd;
}

#endif /* PGBART_SPEC_H */
