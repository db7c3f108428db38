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

/* leaf responses (bart.py:88-90; "linear" and "mix" are flagged experimental upstream) */
#define PGB_RESPONSE_CONSTANT 0
#define PGB_RESPONSE_LINEAR 1 /* per-leaf OLS of sum_trees/m on the parent's split variable */
#define PGB_RESPONSE_MIX 2    /* a fair coin per new leaf between the two                   */

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
#define PGB_FAMILY_POISSON_LOG 5      /* y ~ Poisson(exp(mu))                */
#define PGB_FAMILY_NEGBIN_LOG 6       /* y ~ NegBin(mean exp(mu), alpha)     params: alpha */
#define PGB_FAMILY_ASYMLAPLACE 7      /* y ~ AsymmetricLaplace(b, q, mu): quantile regression   params: b, q */
#define PGB_FAMILY_STUDENT_T 8        /* y ~ StudentT(nu, mu, sigma)         params: sigma, nu */
#define PGB_FAMILY_GAMMA_LOG 9        /* y ~ Gamma(alpha, mean exp(mu)), y > 0   params: alpha */
#define PGB_FAMILY_CALLBACK 10        /* log p(y_i | mu_i) evaluated by a HOST callback (pgb_set_loglik_callback):
                                         the slow fallback for likelihoods outside the closed family -- upstream
                                         evaluates the model's datalogp through PyTensor for every particle.
                                         Per-row values are clamped to [-2047, 2047] like the built-in families
                                         and enter the particle weights through the same fixed-point sums. */

/* RNG purposes (high half of counter word 3) */
#define PGB_RNG_PROPOSE 1u  /* u0: prior coin, u1: split variable            */
#define PGB_RNG_SELECT 2u   /* u0: split row (sub = retry index)             */
#define PGB_RNG_LEAF 3u     /* Box-Muller pair -> (left, right) leaf noise   */
#define PGB_RNG_RESAMPLE 4u /* u0: systematic-resampling offset              */
#define PGB_RNG_FINAL 5u    /* u0: final particle choice                     */
#define PGB_RNG_MIX 6u      /* response = mix: u0 / u1 < 1/2 => the left / right child is linear */

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
  /* |kf| <= 1010 after the clamp: 32-bit conversions give the same integers as 64-bit ones and are
   * single instructions on the GPU (f64 <-> i64 is emulated there) */
  /* round half away from zero: one add of +-0.5 carrying kf's sign and ONE conversion (the two-sided form
   * `kf >= 0 ? (int)(kf + 0.5) : (int)(kf - 0.5)` costs the GPU both conversions and two 64-bit selects;
   * same integers, -0.0 included: both give 0) */
  const int32_t k = (int32_t)(kf + __builtin_copysign(0.5, kf));
  kf = (double)k;
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
  double scale = pgb_u2d((uint64_t)(uint32_t)(k + 1023) << 52);
  return p * scale;
}

/* ------------------------------------------------------------------ log */
/* log(x), x > 0 normal.  x = m 2^e, m in [sqrt(1/2), sqrt(2)); s = (m-1)/(m+1);
 * log m = 2 atanh(s) as an odd series to s^23.  Returns -1e300 for x <= 0. */
PGB_HD double pgb_log(double x) {
  if (!(x == x)) return x;
  if (!(x > 0.0)) return -1.0e300;
  uint64_t b = pgb_d2u(x);
  int32_t e = (int32_t)((b >> 52) & 0x7FF) - 1023; /* (32-bit: see pgb_exp) */
  if (e == -1023) { /* subnormal: rescale */
    x = x * 4503599627370496.0;
    b = pgb_d2u(x);
    e = (int32_t)((b >> 52) & 0x7FF) - 1023 - 52;
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
/* log Phi(x) (standard normal CDF), deterministic: one table row, one degree-8 Horner chain.
 * z = |x|.
 *   x <  0: log Phi(-z) = -z^2/2 + F(z),  F(z) = log(Phi(-z) exp(z^2/2)) (smooth, ~ -log z).
 *           F is tabulated on dyadic intervals -- [0, 1/8) and [2^e (1 + s/8), 2^e (1 + (s+1)/8))
 *           for e = -3..9, s = 0..7 -- so the row index and the local variable come straight from
 *           the exponent and the top three mantissa bits of z: no division, no exp, no log.
 *           z >= 1024: -log z - log sqrt(2 pi) - w + 2.5 w^2, w = z^-2 (next term < 1e-17).
 *   x >= 0: log Phi(z) itself on the SAME dyadic intervals, up to 8.5; 0 beyond (|.| < 1e-17).
 * Both signs share the interval arithmetic AND the Horner evaluation: the sign only selects the
 * table, so a wave with mixed signs runs one path.  Tables: tools/fit_log_ndtr.py
 * (Chebyshev-node interpolation against mpmath at 60 digits; absolute error 4e-15 / 7.5e-15).
 * Absolute error of the function < 4e-13 against scipy.special.log_ndtr on [-38, 38] (the
 * rounding of z^2 at |x| ~ 38), < 1e-14 on [-5, 5] (tests/test_spec.py). */
/* the tables live in accessor functions so that a kernel can stage them in LDS (per-lane rows
 * through the vector L1 cost one cache-line access per distinct row and instruction) */
PGB_HD const double* pgb_ln_tn(void) {
  static const double t[105][9] = {
    {-0x1.7c1095dd2ee18p-1, -0x1.8d1ab6a7490c2p-5, 0x1.6660cbdea201bp-11, -0x1.1fdd292b99f46p-17, 0x1.395f6bb4173c0p-24, -0x1.41183ee9009fcp-35, -0x1.8d47d27c3c90bp-37, 0x1.ab76140dcca81p-43, -0x1.99c0bcea901d1p-51},
    {-0x1.978ca807aaec6p-1, -0x1.80c4c05863cebp-8, 0x1.577ce4a842339p-17, -0x1.14df722e48797p-26, 0x1.379c44e0c7e71p-36, -0x1.d9d51d68b3605p-49, -0x1.59ae71e770c59p-55, 0x1.9a1101d9fa280p-64, -0x1.22075289065abp-74},
    {-0x1.9d8a616688a75p-1, -0x1.7e1902be7d586p-8, 0x1.5441ec8d6b7a0p-17, -0x1.1270d4698c3bcp-26, 0x1.36fe2d9f67ba5p-36, -0x1.0cccdc442a87ep-48, -0x1.4e881dbbbf3fcp-55, 0x1.95696a43fbe72p-64, -0x1.3194ff623f4e9p-74},
    {-0x1.a37d78b12600dp-1, -0x1.7b73b3ca3955ep-8, 0x1.510e3e5dd977ap-17, -0x1.100386892cb9cp-26, 0x1.364c7c3a83404p-36, -0x1.2ba51f9efd232p-48, -0x1.438333722b4a8p-55, 0x1.9085db1b7b31ap-64, -0x1.40046f11cdae0p-74},
    {-0x1.a9660784f73c2p-1, -0x1.78d4c4ebd7cc2p-8, 0x1.4de1d5efb6739p-17, -0x1.0d97af1bd3b13p-26, 0x1.3587d5f9db849p-36, -0x1.49768de2d3f3ap-48, -0x1.38a1474c92720p-55, 0x1.8b6abd06cdcb8p-64, -0x1.4d5dce9e1a66fp-74},
    {-0x1.af44274542dcdp-1, -0x1.763c279c5ece1p-8, 0x1.4abcaea76e87fp-17, -0x1.0b2d7367a44f8p-26, 0x1.34b0de18737b1p-36, -0x1.664480a51eb7fp-48, -0x1.2de3cf23539e3p-55, 0x1.861c57b58c46ap-64, -0x1.59a96b24ca01bp-74},
    {-0x1.b517f11b4680fp-1, -0x1.73a9cd5e7a235p-8, 0x1.479ec37b82f90p-17, -0x1.08c4f76e6ae1ap-26, 0x1.33c835aea7302p-36, -0x1.82127323b4df9p-48, -0x1.234c234dae8bdp-55, 0x1.809ed16ef45e3p-64, -0x1.64efaa8c67a25p-74},
    {-0x1.bae17df65f2a6p-1, -0x1.711da7bf53556p-8, 0x1.44880ef850bd3p-17, -0x1.065e5df1f698ap-26, 0x1.32ce7b9dfd859p-36, -0x1.9ce3ff8bbb02dp-48, -0x1.18db7f8cda226p-55, 0x1.7af62ebcb1bfdp-64, -0x1.6f39049f4da4dp-74},
    {-0x1.c0a0e68c34d99p-1, -0x1.6e97a857623c8p-8, 0x1.41788b43c9bd1p-17, -0x1.03f9c87899fbfp-26, 0x1.31c44c7ea6b87p-36, -0x1.b6bcdc56ab6e3p-48, -0x1.0e9303f910949p-55, 0x1.752652306b0f8p-64, -0x1.788dfc8e5d792p-74},
    {-0x1.c92d34eaf340ap-1, -0x1.6ada117e750edp-7, 0x1.3ceeb3796230dp-15, -0x1.0066f62261ff0p-23, 0x1.30177b13337d3p-32, -0x1.dbb83c29b5db5p-43, -0x1.fee795344d074p-50, 0x1.6c2c7211ce8ecp-57, -0x1.84d1b467f9324p-66},
    {-0x1.d47056729a44cp-1, -0x1.65f2488c1b51cp-7, 0x1.36fa8148ce22ap-15, -0x1.f756c94de1a35p-24, 0x1.2da7aab29a5dcp-32, -0x1.04e2aa0aff3cbp-42, -0x1.d7bece205128dp-50, 0x1.5fce514e94d42p-57, -0x1.922a1efb0faf3p-66},
    {-0x1.df8c985eaeba7p-1, -0x1.612217bf91bccp-7, 0x1.31229661a75aep-15, -0x1.edf4050aba21cp-24, 0x1.2b008e864e9e1p-32, -0x1.1a1b807d4bad8p-42, -0x1.b1f5b44a930fdp-50, 0x1.53127b8dddc5ap-57, -0x1.9c49f31b0feb7p-66},
    {-0x1.ea82b5ad37440p-1, -0x1.5c690e794da5dp-7, 0x1.2b66b32224c2dp-15, -0x1.e4a74807e7e3dp-24, 0x1.282694092dae5p-32, -0x1.2d975580fbecap-42, -0x1.8d95280b448cfp-50, 0x1.4611aa33285ebp-57, -0x1.a373cceadf865p-66},
    {-0x1.f55365db52476p-1, -0x1.57c6bd21f8a32p-7, 0x1.25c693235133ap-15, -0x1.db72180e7fb41p-24, 0x1.251dfe7bf7891p-32, -0x1.3f6737e9c54e2p-42, -0x1.6aa372647ffe5p-50, 0x1.38e287b6f4311p-57, -0x1.a7e8b85b5eb0ap-66},
    {-0x1.ffff5cedc1bbbp-1, -0x1.533ab53cb74d0p-7, 0x1.2041ed9f580f1p-15, -0x1.d255d97907b19p-24, 0x1.21eae62832308p-32, -0x1.4f9c7388afdb4p-42, -0x1.49247de5b6270p-50, 0x1.2b99bdd5e87eep-57, -0x1.a9e7c903d7d57p-66},
    {-0x1.0543a5bd016eep+0, -0x1.4ec48977ded62p-7, 0x1.1ad875d3cb387p-15, -0x1.c953d08d36704p-24, 0x1.1e9137ea22e94p-32, -0x1.5e48760b9b0d6p-42, -0x1.291a0dd716436p-50, 0x1.1e4a06b40c5f8p-57, -0x1.a9adc5023f409p-66},
    {-0x1.0a75ef57f2c15p+0, -0x1.4a63cdbc29ac9p-7, 0x1.1589db5fd750ep-15, -0x1.c06d22d85b964p-24, 0x1.1b14b4fb66d7bp-32, -0x1.6b7cb666fe6f3p-42, -0x1.0a83f36258ad8p-50, 0x1.1104406242baap-57, -0x1.a774e14614c97p-66},
    {-0x1.1220f16b3718ep+0, -0x1.43f9fceb7ce00p-6, 0x1.0dc59aed923b5p-13, -0x1.b348a38c67f73p-21, 0x1.15a06d2f35531p-28, -0x1.7cb09a8ba2095p-37, -0x1.beb08a9a327c4p-45, 0x1.fa9fa56f8229ep-51, -0x1.a0d30d2438b6bp-58},
    {-0x1.1c1f745d3dd24p+0, -0x1.3bb415039ffcfp-6, 0x1.03c57ca737789p-13, -0x1.a22b1e89d385ep-21, 0x1.0e00d95ce1a9cp-28, -0x1.8f1e9430979f4p-37, -0x1.558294cad0ea5p-45, 0x1.c7558617c2854p-51, -0x1.92f7168d213bep-58},
    {-0x1.25dd03d41e61dp+0, -0x1.33bc9724315cdp-6, 0x1.f4553862b4dadp-14, -0x1.918a33f448e1fp-21, 0x1.061103dc28a7bp-28, -0x1.9ce0e61377de9p-37, -0x1.eeadf29a9e4aep-46, 0x1.9611e0cb47871p-51, -0x1.80a424757559dp-58},
    {-0x1.2f5c0697d4713p+0, -0x1.2c1065326d3f3p-6, 0x1.e1e4009909353p-14, -0x1.816a33e347736p-21, 0x1.fbce02cb3cf84p-29, -0x1.a678c99c3b69bp-37, -0x1.475c42d7b0a73p-46, 0x1.674eef839e1cdp-51, -0x1.6b1c97d991c74p-58},
    {-0x1.389ecadfabec1p+0, -0x1.24ac793a536d7p-6, 0x1.d03133a95f15fp-14, -0x1.71ce210017166p-21, 0x1.eb2ce0f8bb44bp-29, -0x1.ac60edbe305f0p-37, -0x1.67c7b1b3bf28ep-47, 0x1.3b61a12cb7547p-51, -0x1.537319463fc41p-58},
    {-0x1.41a78715e2c28p+0, -0x1.1d8de5ffc049ap-6, 0x1.bf36963b9e04fp-14, -0x1.62b7d6d6a32b2p-21, 0x1.da6138f0455d8p-29, -0x1.af0c96ddb6325p-37, -0x1.97d33a9ce63c2p-49, 0x1.127f6838e1f56p-51, -0x1.3a8d4fba11cf0p-58},
    {-0x1.4a785a9ef152ep+0, -0x1.16b1d758271a1p-6, 0x1.aeeddce28e51ep-14, -0x1.54282dd7302b3p-21, 0x1.c9894835e4270p-29, -0x1.aee711713151bp-37, 0x1.e5debe98d2026p-49, 0x1.d98745699911fp-52, -0x1.21277fb565947p-58},
    {-0x1.53134ea2d9d05p+0, -0x1.10159253ad32fp-6, 0x1.9f50b77576064p-14, -0x1.461f1cdc5f7fbp-21, 0x1.b8bf44711b2f6p-29, -0x1.ac53693075967p-37, 0x1.3943a681c47dep-47, 0x1.94690a7dfa114p-52, -0x1.07d8afa23844ap-58},
    {-0x1.5f9af900426eap+0, -0x1.069d09fceb65fp-5, 0x1.8918f3ff32ac8p-12, -0x1.320bf4774920dp-18, 0x1.9fdaa5d680c25p-25, -0x1.a4aba2887abd6p-32, 0x1.11d856e3efb96p-40, 0x1.38916e7e1a1dep-45, -0x1.c61a475838261p-51},
    {-0x1.6fa4dab0716c3p+0, -0x1.f587aeeca7bccp-6, 0x1.6d979343d2cbap-12, -0x1.191239556f4ffp-18, 0x1.7f9119412b68dp-25, -0x1.94ffb15c37c6ep-32, 0x1.836148cbe83e2p-40, 0x1.a52ae0c4fa0dbp-46, -0x1.6b688a1fd83f7p-51},
    {-0x1.7ef7d89d2b969p+0, -0x1.df7b2477a8cdfp-6, 0x1.5455b96bf7f30p-12, -0x1.02124a59fb2cdp-18, 0x1.60af9c40b153bp-25, -0x1.80f9b7f0c93ccp-32, 0x1.cd12c8b1e588bp-40, 0x1.03dcfeddb0689p-46, -0x1.1bbc71455de82p-51},
    {-0x1.8da095a14fbe7p+0, -0x1.caf1f195a6a34p-6, 0x1.3d251ce7a4eb4p-12, -0x1.d9e6986633c63p-19, 0x1.437bf3776ba3fp-25, -0x1.6a424bfafa315p-32, 0x1.f7a1a9b995673p-40, 0x1.0ed126e0ce400p-47, -0x1.b019656dafb0fp-52},
    {-0x1.9baaabde4541ap+0, -0x1.b7cc643e71df2p-6, 0x1.27d9f53dd17e2p-12, -0x1.b331f75922dcep-19, 0x1.281e98ed52a71p-25, -0x1.52214f21574e3p-32, 0x1.053ae2aedf26ap-39, 0x1.4eb3cb9154f28p-49, -0x1.402c0a12c7af5p-52},
    {-0x1.a920c29dad3a6p+0, -0x1.a5ed713560ea4p-6, 0x1.144b38a1a9158p-12, -0x1.8fca9255146d7p-19, 0x1.0ea9adf7258b6p-25, -0x1.398d7fcd24e6ep-32, 0x1.05d81ca711f09p-39, -0x1.a7cfdfba56a9dp-50, -0x1.cb8050392fe80p-53},
    {-0x1.b60ca2d523c3bp+0, -0x1.953a871887aaep-6, 0x1.0252b7dc865f0p-12, -0x1.6f731d43be685p-19, 0x1.ee3d78b7e45d7p-26, -0x1.213b39fb51b5ep-32, 0x1.0022e4904dfd7p-39, -0x1.2a80e326439efp-48, -0x1.3c699d4136cfap-53},
    {-0x1.c2774a3dc0351p+0, -0x1.859b6083e6de5p-6, 0x1.e39a4262e308cp-13, -0x1.51eee4d9b2f36p-19, 0x1.c2e6bd51b40edp-26, -0x1.09a9b7b2c6639p-32, 0x1.ec232885441ddp-40, -0x1.abc2a9c50f3a7p-48, -0x1.9acc112dd231ep-54},
    {-0x1.d436e13388a73p+0, -0x1.7001c9c13fa66p-5, 0x1.b6eba906997b7p-11, -0x1.2a749d3eda535p-16, 0x1.889671b0b057fp-22, -0x1.d0d61730fddafp-28, 0x1.c40c70f30eb79p-34, -0x1.0ac49e8c6b4efp-40, -0x1.5dc0f08f88c11p-47},
    {-0x1.ea647d005a2c5p+0, -0x1.563b4cdeb189ep-5, 0x1.834a269ce14d2p-11, -0x1.fb4f37713098fp-17, 0x1.4648e705f68bbp-22, -0x1.81c29ffec1c7bp-28, 0x1.86cc20ff2cb9bp-34, -0x1.1cd5e0d672dd7p-40, 0x1.4de1a56907b74p-51},
    {-0x1.ff0e2b51417a1p+0, -0x1.3f6fa97f0aa74p-5, 0x1.5754132b0733ap-11, -0x1.b0d0839374661p-17, 0x1.0f6e9c0d1de8ep-22, -0x1.3e440580903cep-28, 0x1.49b7c2248b1b4p-34, -0x1.0d8c3e2470fd9p-40, 0x1.90294e1d9db2cp-48},
    {-0x1.092ffed73ad20p+1, -0x1.2b2ecd2fa392dp-5, 0x1.31bf6c807d3a9p-11, -0x1.72c9ab3a0c5b5p-17, 0x1.c481234461dedp-23, -0x1.05c841f295aecp-28, 0x1.11e120ac115bap-34, -0x1.dedaf1a225a63p-41, 0x1.0caa7455e3f79p-47},
    {-0x1.123fcfae50152p+1, -0x1.191b670420d8fp-5, 0x1.117c4d3670f85p-11, -0x1.3f04d383437c4p-17, 0x1.7a3a603cd26cep-23, -0x1.ae36582d54646p-29, 0x1.c287212b758e8p-35, -0x1.99b98b29363edp-41, 0x1.159b3078c6e1cp-47},
    {-0x1.1ac6b30967b45p+1, -0x1.08e795517c23ap-5, 0x1.eb56557a27dfap-12, -0x1.13ae0c52fc8d0p-17, 0x1.3d3008f74ef2ap-23, -0x1.61ac050f89ee9p-29, 0x1.705bc11b64651p-35, -0x1.5678b26c34f01p-41, 0x1.015e4cb3e79e7p-47},
    {-0x1.22d2991d9ac9ep+1, -0x1.f4a44c0f8ba79p-6, 0x1.bb2897d774fc3p-12, -0x1.de8af8afbad50p-18, 0x1.0af92b1c66cefp-23, -0x1.23332c9d61bc8p-29, 0x1.2c372b66d29d5p-35, -0x1.1a04d087ee44cp-41, 0x1.c27f6400a4812p-48},
    {-0x1.2a6f93a34f22ep+1, -0x1.da48aa6c8b26fp-6, 0x1.91411890723c2p-12, -0x1.a1237e1566fd6p-18, 0x1.c32b3acdb56eap-24, -0x1.e0a4a7bc969f9p-30, 0x1.e8c302479423ap-36, -0x1.cc1c43fa219fap-42, 0x1.7d0423845c42ap-48},
    {-0x1.3521a931372ffp+1, -0x1.b73db57ef13c1p-5, 0x1.5c1498e54b348p-10, -0x1.562bf9e45cf5bp-15, 0x1.61174ac909615p-20, -0x1.6a59b0f03fb39p-25, 0x1.673a86b2dcb62p-30, -0x1.4fff42352c5d9p-35, 0x1.1d91efc390878p-40},
    {-0x1.4237980e0a9f5p+1, -0x1.8f6bf62d4a980p-5, 0x1.2363df2f03bd9p-10, -0x1.0a76d9b1ec999p-15, 0x1.0251add7aa40fp-20, -0x1.f7453a01060bdp-26, 0x1.df79d540abb35p-31, -0x1.b664acf5d7247p-36, 0x1.76a7ca7a73d65p-41},
    {-0x1.4e291fc988c3cp+1, -0x1.6de2e052face6p-5, 0x1.edd675a563441p-11, -0x1.a54d4ead4461cp-16, 0x1.8016426bb6360p-21, -0x1.62c5cbf530703p-26, 0x1.4387fbf58b80dp-31, -0x1.1e9950b14697ap-36, 0x1.e3421fc45f15ap-42},
    {-0x1.5922fb9cd60eap+1, -0x1.5150a81c01a4bp-5, 0x1.a70c56f477c59p-11, -0x1.51b31cf0f011ap-16, 0x1.21f7ef5548e90p-21, -0x1.fbeb2265e3d09p-27, 0x1.ba75ec6d365b7p-32, -0x1.79cbf94aaee02p-37, 0x1.36dfb412d3af1p-42},
    {-0x1.6348bfd9c6287p+1, -0x1.38b88434a0bccp-5, 0x1.6df3e67a03215p-11, -0x1.1223ffa785bd9p-16, 0x1.bc37f94fe7c55p-22, -0x1.71201b7a35b2fp-27, 0x1.32e87cdb12811p-32, -0x1.f7b71c0aefca5p-38, 0x1.920318557bd97p-43},
    {-0x1.6cb71ac88e7e1p+1, -0x1.235a6a8faa114p-5, 0x1.3f56d7806a28bp-11, -0x1.c24573fe4451ep-17, 0x1.58e5d6a4938ffp-22, -0x1.1031a31daaf30p-27, 0x1.affcde2d7ce50p-33, -0x1.543154e70ee6ep-38, 0x1.065d0eec1b74fp-43},
    {-0x1.758573dc00849p+1, -0x1.10a263e930f5ap-5, 0x1.18dbe1d8c60d2p-11, -0x1.75b57e7fe540ap-17, 0x1.0f2ba1a96108ap-22, -0x1.9706ca9b70b5ep-28, 0x1.34649ad0dd66ep-33, -0x1.d1d43ce8e8521p-39, 0x1.5a6d6835539b7p-44},
    {-0x1.7dc71b0cfb7dap+1, -0x1.001cebe13e518p-5, 0x1.f18a08d7f1d63p-12, -0x1.392bc530e75aap-17, 0x1.af65d51d91ba1p-23, -0x1.3451c1110be40p-28, 0x1.be6f8050eda1fp-34, -0x1.4359c29a6fef6p-39, 0x1.cf405fcc98cf2p-45},
    {-0x1.89444941c5639p+1, -0x1.d561e70624c15p-5, 0x1.a3aeb3676e872p-10, -0x1.e861b26223c98p-15, 0x1.3833c8136a69dp-19, -0x1.9fdb7c92b4f63p-24, 0x1.19bafb54b9646p-28, -0x1.80b4aa6dd035ap-33, 0x1.04500e9770001p-37},
    {-0x1.972ba3f08010cp+1, -0x1.a617da00ce00ap-5, 0x1.54f1f54903bc7p-10, -0x1.67fcb571196eap-15, 0x1.a3595d666eb32p-20, -0x1.ff0f88597c24ap-25, 0x1.3e1fd7c973937p-29, -0x1.90b968e29f14bp-34, 0x1.f7079828219b2p-39},
    {-0x1.a3bc66d06351dp+1, -0x1.7f51dbe2b799dp-5, 0x1.1a28d48d18992p-10, -0x1.10629cc220221p-15, 0x1.2309114ed7ec9p-20, -0x1.465b80e3fff2cp-25, 0x1.770e9a598d9b2p-30, -0x1.b5602b5a23425p-35, 0x1.fe5f3f41cef38p-40},
    {-0x1.af31dc5999bf7p+1, -0x1.5efb05097d7e3p-5, 0x1.da610944f1caep-11, -0x1.a580e9e5a8cf3p-16, 0x1.9f8e7f1c2d99bp-21, -0x1.af09ea24857fap-26, 0x1.cb591ef120aa8p-31, -0x1.f1d21c1f0c53ap-36, 0x1.0ec5ea3a71bb3p-40},
    {-0x1.b9b950b032245p+1, -0x1.439d29c92b4f8p-5, 0x1.942162e83abf7p-11, -0x1.4c72afc450166p-16, 0x1.300adbe005ef2p-21, -0x1.251fa30225528p-26, 0x1.22e9a1e488669p-31, -0x1.261d7eb030f88p-36, 0x1.2b33dea845269p-41},
    {-0x1.c3761bbdddbd3p+1, -0x1.2c2a6847d01ecp-5, 0x1.5c4497f9b1bcdp-11, -0x1.0a9bf7928415ep-16, 0x1.c687c1c637be5p-22, -0x1.9914cf8355a58p-27, 0x1.7b9c5f9953a04p-32, -0x1.6754a7e59b740p-37, 0x1.56eaeeb99d999p-42},
    {-0x1.cc84527e5f0e8p+1, -0x1.17db7e4e7328bp-5, 0x1.2f255f674a1afp-11, -0x1.b1e309375fc67p-17, 0x1.5a3353edb654dp-22, -0x1.240744ca2a968p-27, 0x1.fc9eee2fc710bp-33, -0x1.c44da12be336bp-38, 0x1.9627b4af7bbbap-43},
    {-0x1.d4fa9fc291064p+1, -0x1.061a460e9e014p-5, 0x1.0a3257beb44a5p-11, -0x1.659ce9919adcbp-17, 0x1.0c1c26536afa1p-22, -0x1.a973405e2ca31p-28, 0x1.5cdd7a1991b9bp-33, -0x1.24611d9bc7913p-38, 0x1.ef83edb17ca4ap-44},
    {-0x1.e0b6eeabb1608p+1, -0x1.de9a9225075d4p-5, 0x1.bc61ed1c07b04p-10, -0x1.113e5a10266b1p-14, 0x1.77883b622489ap-19, -0x1.1171bc4c0e685p-23, 0x1.9c1447c628721p-28, -0x1.3efb564a1b8f2p-32, 0x1.f216ce7053a92p-37},
    {-0x1.eedd4de216516p+1, -0x1.accd37be4b86ap-5, 0x1.652e6292995e7p-10, -0x1.8a91dfba55a57p-15, 0x1.e7bd5c0d880fap-20, -0x1.3fd690c92a866p-24, 0x1.b29fed3655ab9p-29, -0x1.2f75c24ad827ep-33, 0x1.ac63baed6da9dp-38},
    {-0x1.fb9c938c93ca1p+1, -0x1.8458882658651p-5, 0x1.253f1002bbd60p-10, -0x1.25f16deeb52c0p-15, 0x1.4a03a083a2d37p-20, -0x1.8979f9d80ac89p-25, 0x1.e68c03eb71fabp-30, -0x1.352fcc51ee9dap-34, 0x1.8dedce83d9301p-39},
    {-0x1.039aa71ce1a35p+2, -0x1.62d701e46738bp-5, 0x1.ea03398731ddap-11, -0x1.c17363150a8dap-16, 0x1.ce11943135c52p-21, -0x1.f8d8579d0d933p-26, 0x1.1e39dce0542c6p-30, -0x1.4da290e2a68dbp-35, 0x1.8a5363032275bp-40},
    {-0x1.08ec0f90cfd75p+2, -0x1.46a3c458f7bc4p-5, 0x1.9f74e923839c8p-11, -0x1.5f2d2cfe2806ep-16, 0x1.4ce6d1abac0ddp-21, -0x1.4f90393609269p-26, 0x1.5f3d38b13b98ap-31, -0x1.79f865a566b38p-36, 0x1.9cd47a70ddd3cp-41},
    {-0x1.0dd5473889f7ap+2, -0x1.2e94a68568d09p-5, 0x1.64aaee8a786a8p-11, -0x1.1786809f54677p-16, 0x1.eb936ad1386a9p-22, -0x1.cbd21b39019c1p-27, 0x1.bed5982159fd1p-32, -0x1.be71129a296b8p-37, 0x1.c51842ce79ddap-42},
    {-0x1.1265167a20a92p+2, -0x1.19d09fe09f3d2p-5, 0x1.3580c8bf6c0f6p-11, -0x1.c427155aed134p-17, 0x1.72b09466a1d83p-22, -0x1.4367f6e60eebdp-27, 0x1.25393fc067ce8p-32, -0x1.115c51e8b60bcp-37, 0x1.03090d90b3dfbp-42},
    {-0x1.16a7570ead596p+2, -0x1.07b6001f4209fp-5, 0x1.0f18113bae992p-11, -0x1.72d087ebd6de7p-17, 0x1.1cb963cf6b32cp-22, -0x1.d16f6a59ca341p-28, 0x1.8b75b01f6c4ccp-33, -0x1.5980d19ce4341p-38, 0x1.32fc1db6c12d5p-43},
    {-0x1.1c8dec18c1856p+2, -0x1.e10d5eea71bfep-5, 0x1.c332fbe813098p-10, -0x1.19a63bddc6219p-14, 0x1.8ae6c328dcdfep-19, -0x1.26cb8b23949efp-23, 0x1.c9b0e7970faf1p-28, -0x1.6efc2e4100601p-32, 0x1.2a26b17db7a72p-36},
    {-0x1.23a96d6068f22p+2, -0x1.ae90125ffa64fp-5, 0x1.699490b128ef3p-10, -0x1.944f73fa10872p-15, 0x1.fbe752fd81037p-20, -0x1.53d1aa8e72c42p-24, 0x1.d90533adcf30ep-29, -0x1.53c48e0c46502p-33, 0x1.ef4b63c2790c3p-38},
    {-0x1.2a0f248f25e12p+2, -0x1.85a779c27ccc4p-5, 0x1.283590235ed20p-10, -0x1.2be4c2f8a2654p-15, 0x1.55314a1b58133p-20, -0x1.9d9708857a4c2p-25, 0x1.04d3b7b3590afp-29, -0x1.53510c9dc41e4p-34, 0x1.c0686db46298ep-39},
    {-0x1.2fe01556963acp+2, -0x1.63d68aeec9c41p-5, 0x1.ee258cd002983p-11, -0x1.c90bbaf8c62ebp-16, 0x1.db20127a90ac5p-21, -0x1.072d0509e407cp-25, 0x1.2f6a17bf58befp-30, -0x1.68a2dde78c46fp-35, 0x1.b3c86acf688b0p-40},
    {-0x1.3535055a3e525p+2, -0x1.476b1ade86354p-5, 0x1.a26d56fe02b29p-11, -0x1.643459f104a3cp-16, 0x1.54de2248e0327p-21, -0x1.5ba8eb3840dbap-26, 0x1.71119fc19b267p-31, -0x1.93c90c4aa0422p-36, 0x1.c16684c4c5d69p-41},
    {-0x1.3a210473e9474p+2, -0x1.2f331d3e495e4p-5, 0x1.66db2db32f1f7p-11, -0x1.1af59a0ad23eap-16, 0x1.f5aa39831f261p-22, -0x1.da076db4006ffp-27, 0x1.d242d27bfbc72p-32, -0x1.d885eac9b2dcep-37, 0x1.e7624e8c15071p-42},
    {-0x1.3eb30dd25ad51p+2, -0x1.1a50a7fa283b7p-5, 0x1.3726893bf504ap-11, -0x1.c8f884425e5a5p-17, 0x1.79496baac6905p-22, -0x1.4c112c2923499p-27, 0x1.3043dfec373a9p-32, -0x1.1f2d9e89722dfp-37, 0x1.13facaec6fd74p-42},
    {-0x1.42f71e3f24e31p+2, -0x1.081ee99769c82p-5, 0x1.105b8f946eb62p-11, -0x1.76463cdb9ae3fp-17, 0x1.2128e8032699ap-22, -0x1.dc578c483d65cp-28, 0x1.987b9d8c17632p-33, -0x1.68c37fd12ac4cp-38, 0x1.44847cc178fc5p-43},
    {-0x1.48dfd831b6df2p+2, -0x1.e1ac952e85cf9p-5, 0x1.c4f2ea90f189ap-10, -0x1.1bd5d63eb8a32p-14, 0x1.9004700886da2p-19, -0x1.2c89733b3431dp-23, 0x1.d63603fddede4p-28, -0x1.7c792bec2806ap-32, 0x1.384da01e7e311p-36},
    {-0x1.4ffd759594e01p+2, -0x1.af023c3050ab4p-5, 0x1.6ab426a36f2b1p-10, -0x1.96d2f2c7263ccp-15, 0x1.00961ee0d3e8cp-19, -0x1.591df16658a00p-24, 0x1.e35ef935e0574p-29, -0x1.5dbdf7af5c0b8p-33, 0x1.01071630565bfp-37},
    {-0x1.5664b6689a4d7p+2, -0x1.85fc184223a2cp-5, 0x1.28f68613a8a57p-10, -0x1.2d6ba8984d99dp-15, 0x1.58173ca4930c9p-20, -0x1.a2de2885628b2p-25, 0x1.097f19b0230f7p-29, -0x1.5b74fcab18c0bp-34, 0x1.ce47870f0d606p-39},
    {-0x1.5c36cee1a865ep+2, -0x1.6416fd2cbbce3p-5, 0x1.ef3204088d0fdp-11, -0x1.cafc88cd906b3p-16, 0x1.de7d8b0e52c07p-21, -0x1.09f9b17ead1d2p-25, 0x1.33f0d88937e1fp-30, -0x1.6fd6f8653c07dp-35, 0x1.bf006c4292d4bp-40},
    {-0x1.618ca29f78f13p+2, -0x1.479d4fa899f9bp-5, 0x1.a32dcfbfe9784p-11, -0x1.657c27b65a4bbp-16, 0x1.56e950d4bd7ebp-21, -0x1.5eca30ddff3dfp-26, 0x1.75ba3ad81df29p-31, -0x1.9a9af4b034be4p-36, 0x1.cb2dfb73ed0e9p-41},
    {-0x1.667954d123a20p+2, -0x1.2f5afbd4363e4p-5, 0x1.6768bc178626cp-11, -0x1.1bd4e5c3026cap-16, 0x1.f83e6e1fe15f4p-22, -0x1.ddb00c066d1c2p-27, 0x1.d74e388ad21a4p-32, -0x1.df5c3a89bc28ap-37, 0x1.f0779359dc0b8p-42},
    {-0x1.6b0bed90d2b1dp+2, -0x1.1a70d77cb219fp-5, 0x1.3790f32997c04p-11, -0x1.ca3128b9615dep-17, 0x1.7af7da7fb197ap-22, -0x1.4e49e49fafb15p-27, 0x1.331e376f23ca1p-32, -0x1.22c726b3d3820p-37, 0x1.186f282ceb952p-42},
    {-0x1.6f50728e3524ap+2, -0x1.08394497b8f00p-5, 0x1.10ad16ed01c5bp-11, -0x1.77265b8cdf8f0p-17, 0x1.22499febe5cd6p-22, -0x1.df2178916af83p-28, 0x1.9bd589ced10b9p-33, -0x1.6cb81a06e2887p-38, 0x1.49192a87f0c6cp-43},
    {-0x1.7539b65749285p+2, -0x1.e1d48bf17c81ap-5, 0x1.c563a771d9f65p-10, -0x1.1c6327200b13fp-14, 0x1.91507f617bc7cp-19, -0x1.2dffe16782699p-23, 0x1.d96ad1be0951bp-28, -0x1.7ff2e2d6aad32p-32, 0x1.3bf95b7858d6cp-36},
    {-0x1.7c57db400dad9p+2, -0x1.af1ede4dc43b7p-5, 0x1.6afc6f7c95a47p-10, -0x1.97752020e6e26p-15, 0x1.0140a62c3b743p-19, -0x1.5a76260f26e97p-24, 0x1.e60255e261edcp-29, -0x1.604bf6aae77a7p-33, 0x1.037170b22a3a1p-37},
    {-0x1.82bf7ec658a86p+2, -0x1.86114e3ef3cdbp-5, 0x1.2926fa284dc0cp-10, -0x1.2dce07f6cffe5p-15, 0x1.58d2728e74bcfp-20, -0x1.a43420d84e9d8p-25, 0x1.0aaeb74089b22p-29, -0x1.5d8868ef5624cp-34, 0x1.d1d51135102fdp-39},
    {-0x1.8891e159d3d63p+2, -0x1.642722db7a345p-5, 0x1.ef75612e46bc0p-11, -0x1.cb796c2451a28p-16, 0x1.df56946426f05p-21, -0x1.0aaeb6c7cd2eep-25, 0x1.3516634f5ec3cp-30, -0x1.71abb09d8a4d1p-35, 0x1.c1dd154280ae2p-40},
    {-0x1.8de7ee23f17d7p+2, -0x1.47a9e2df53885p-5, 0x1.a35e146272e48p-11, -0x1.65ce7d65e923ap-16, 0x1.576cf814261c2p-21, -0x1.5f9442fe53f7fp-26, 0x1.76e7bcef59f70p-31, -0x1.9c5598032c6a1p-36, 0x1.cdaaa77e1450cp-41},
    {-0x1.92d4cd2f8ad7bp+2, -0x1.2f64f79282949p-5, 0x1.678c37ef91c2fp-11, -0x1.1c0cf2131477fp-16, 0x1.f8e468e3bf9c8p-22, -0x1.de9bf54993803p-27, 0x1.d8942fba80b87p-32, -0x1.e11719609acb1p-37, 0x1.f2c57e0596e91p-42},
    {-0x1.976789d54a2cap+2, -0x1.1a78e63b5f237p-5, 0x1.37ab9d7238e08p-11, -0x1.ca7f977dee844p-17, 0x1.7b63fc8004516p-22, -0x1.4ed8ff6d01d8ep-27, 0x1.33d65451b488fp-32, -0x1.23aff2d43f152p-37, 0x1.198fe6def5397p-42},
    {-0x1.9bac2c008f546p+2, -0x1.083fdd65e32b8p-5, 0x1.10c1835bb47a9p-11, -0x1.775e8eec40690p-17, 0x1.22921cbbd6a92p-22, -0x1.dfd4f7a77ec16p-28, 0x1.9cad96cd537b1p-33, -0x1.6db792bb825b9p-38, 0x1.4a419f65aa47bp-43},
    {-0x1.a19592491b611p+2, -0x1.e1de8c397dd91p-5, 0x1.c57fe2d8b4b1fp-10, -0x1.1c869240fe6d4p-14, 0x1.91a3ce98e552cp-19, -0x1.2e5dee4f0bd9dp-23, 0x1.da3944d0cbfb7p-28, -0x1.80d303db55b9ep-32, 0x1.3ce67c3e36b3ep-36},
    {-0x1.a8b3d91b0286ap+2, -0x1.af2608519ee9ap-5, 0x1.6b0e87f3dd8c2p-10, -0x1.979dc083effe0p-15, 0x1.016b66fd11f78p-19, -0x1.5acc86a6c2245p-24, 0x1.e6abff9522f5dp-29, -0x1.60f07648d7999p-33, 0x1.040d2f0360d71p-37},
    {-0x1.af1b9552c78fdp+2, -0x1.86169ca4e8c83p-5, 0x1.29331a9c0b112p-10, -0x1.2de6aa424b7f1p-15, 0x1.59015be34fe8ap-20, -0x1.a489e2c6ad964p-25, 0x1.0afaec1d93fddp-29, -0x1.5e0dee10c1ec2p-34, 0x1.d2b9d77ef4c0cp-39},
    {-0x1.b4ee0a6fc127ep+2, -0x1.642b2cd99e5ebp-5, 0x1.ef863c720f6f5p-11, -0x1.cb98b00a2b606p-16, 0x1.df8cf1a5183edp-21, -0x1.0adc160b270a6p-25, 0x1.356004716a5d2p-30, -0x1.72215ba445fd9p-35, 0x1.c2952e43c4e2fp-40},
    {-0x1.ba44257ecfc9ep+2, -0x1.47ad080d8cee1p-5, 0x1.a36a27f4b7ea0p-11, -0x1.65e318fe3b2b0p-16, 0x1.578defb655cb2p-21, -0x1.5fc6e3d094699p-26, 0x1.773353bd2b5d9p-31, -0x1.9cc4a4b00ec26p-36, 0x1.ce4a820766f62p-41},
    {-0x1.bf310fc224973p+2, -0x1.2f6776c3cc9cdp-5, 0x1.6795186adf083p-11, -0x1.1c1af8c180871p-16, 0x1.f90df68549ef1p-22, -0x1.ded70beb2dc70p-27, 0x1.d8e5dfd615c05p-32, -0x1.e18626b3aaf97p-37, 0x1.f3598412e8de0p-42},
    {-0x1.c3c3d562448f2p+2, -0x1.1a7aea190477cp-5, 0x1.37b24901f8bdfp-11, -0x1.ca93378dfefcep-17, 0x1.7b7f0d6ff3d1bp-22, -0x1.4efcd503e9bd3p-27, 0x1.3404742c233eap-32, -0x1.23ea4cdc284afp-37, 0x1.19d851aa86d37p-42},
    {-0x1.c8087ed99e7efp+2, -0x1.084183ba5f11dp-5, 0x1.10c69f215d16bp-11, -0x1.776c9e81c264ep-17, 0x1.22a440e2eaed0p-22, -0x1.e001e7c4e5ae4p-28, 0x1.9ce3b35881452p-33, -0x1.6df7965a8a6cfp-38, 0x1.4a8bf1b8cdfd4p-43},
    {-0x1.cdf1edc2a64c6p+2, -0x1.e1e10c75031eap-5, 0x1.c586f275c3671p-10, -0x1.1c8f6e78d8fbep-14, 0x1.91b8a721616d6p-19, -0x1.2e7578a74779ap-23, 0x1.da6cf5ad329e1p-28, -0x1.810b2781aec80p-32, 0x1.3d21e7d8b0201p-36},
    {-0x1.d5103d0f54d82p+2, -0x1.af27d2ea651e5p-5, 0x1.6b130e75ec9f8p-10, -0x1.97a7e9ee46b7ep-15, 0x1.01761922a781bp-19, -0x1.5ae224088c114p-24, 0x1.e6d67739e0e6dp-29, -0x1.6119a64516eedp-33, 0x1.04343131e328dp-37},
    {-0x1.db77ff73c3058p+2, -0x1.8617f04cd5e52p-5, 0x1.293622eff7f50p-10, -0x1.2decd37cba888p-15, 0x1.590d17f7477aap-20, -0x1.a49f57833c2f9p-25, 0x1.0b0dfe3116d17p-29, -0x1.5e2f5a068abd2p-34, 0x1.d2f31f6c1ded3p-39},
    {-0x1.e14a79334a75ap+2, -0x1.642c2f6250b67p-5, 0x1.ef8a7382bae0dp-11, -0x1.cba081b4ed2e5p-16, 0x1.df9a8aa4f00c1p-21, -0x1.0ae76fbc4b3b2p-25, 0x1.357270a41c757p-30, -0x1.723ece3d17e57p-35, 0x1.c2c343814fc76p-40},
    {-0x1.e6a097d3b337ap+2, -0x1.47add15f25179p-5, 0x1.a36d2cffed695p-11, -0x1.65e8404738f24p-16, 0x1.57962e7c6a04cp-21, -0x1.5fd38dcac46d4p-26, 0x1.77463cd771b8ep-31, -0x1.9ce06e1debe23p-36, 0x1.ce7283acd2331p-41},
    {-0x1.eb8d84e50ae1cp+2, -0x1.2f6816943b362p-5, 0x1.679750a20c713p-11, -0x1.1c1e7aa6d3374p-16, 0x1.f9185add1164cp-22, -0x1.dee5d3597aab5p-27, 0x1.d8fa4f03cd7e3p-32, -0x1.e1a1ef660a5e4p-37, 0x1.f37e8e53a5868p-42},
    {-0x1.f0204cc3d0cdap+2, -0x1.1a7b6b134dd3ep-5, 0x1.37b3f3f5c541bp-11, -0x1.ca981fd804048p-17, 0x1.7b85d2331a439p-22, -0x1.4f05cb583ff53p-27, 0x1.340ffdada3d96p-32, -0x1.23f8e5cf9cbbcp-37, 0x1.19ea7012071bap-42},
    {-0x1.f464f80e39b26p+2, -0x1.0841ed518d60cp-5, 0x1.10c7e61d68ca6p-11, -0x1.77702293076d6p-17, 0x1.22a8ca3bf88b4p-22, -0x1.e00d24d21cac0p-28, 0x1.9cf13c10b9855p-33, -0x1.6e07999aaab5cp-38, 0x1.4a9e89a16afefp-43},
  };
  return &t[0][0];
}
PGB_HD const double* pgb_ln_tp(void) {
  static const double t[105][9] = {
    {-0x1.49fdb8239277dp-1, 0x1.845d8f1d630c8p-5, -0x1.3edbb6f305fdbp-10, 0x1.336f39cd291b8p-17, 0x1.37ccb4ad14d39p-24, -0x1.fa9c942d19b25p-34, -0x1.ecbfca37ae037p-37, -0x1.b87b592fdaa63p-43, 0x1.5bbcb0e37cd90p-55},
    {-0x1.2f77317c8eb32p-1, 0x1.6e3bee4ac297dp-8, -0x1.369bab58aa2fap-16, 0x1.3e556129ddbfdp-26, 0x1.33d53f55348f7p-36, -0x1.d8981a3bd8be9p-48, -0x1.115b6ad0eb368p-54, -0x1.b284f214272efp-64, 0x1.5692c50adceeep-75},
    {-0x1.29c7f1a501b73p-1, 0x1.69653d049704cp-8, -0x1.34bc5e1ec175ap-16, 0x1.40bbd98d105ecp-26, 0x1.329dbe41a066dp-36, -0x1.0633c18677858p-47, -0x1.1747358acca92p-54, -0x1.af88c53d000e8p-64, 0x1.a612d78c264aap-75},
    {-0x1.242bfd8e8c34cp-1, 0x1.649610243cf98p-8, -0x1.32d97913cc89fp-16, 0x1.431fc2515f9d7p-26, 0x1.314582f2a1a44p-36, -0x1.20a90accca80ap-47, -0x1.1d2774335deefp-54, -0x1.abeab71dec4b1p-64, 0x1.f87355fe41fa7p-75},
    {-0x1.1ea3370ae2dd8p-1, 0x1.5fce760107fd5p-8, -0x1.30f30040a96acp-16, 0x1.4580d951d39cdp-26, 0x1.2fcbdd26821cep-36, -0x1.3baab8efe2665p-47, -0x1.22f9e62a11f63p-54, -0x1.a7a50bed4b6f8p-64, 0x1.26d76dfca9771p-74},
    {-0x1.192d7fb27e65fp-1, 0x1.5b0e7ce16050fp-8, -0x1.2f08f81275265p-16, 0x1.47dedb0a83cacp-26, 0x1.2e301e39ba759p-36, -0x1.573764ef84289p-47, -0x1.28bc36d24a825p-54, -0x1.a2b214c23aa14p-64, 0x1.52def45b493b0p-74},
    {-0x1.13cab8e4e259ep-1, 0x1.565632f92dd8ap-8, -0x1.2d1b655c97b3ap-16, 0x1.4a39829c18c6cp-26, 0x1.2c71996cdbfa8p-36, -0x1.734d6eeef39f7p-47, -0x1.2e6bfdc7ec38fp-54, -0x1.9d0c33e86dc9ap-64, 0x1.804b61fe35695p-74},
    {-0x1.0e7ac3c8ea55ap-1, 0x1.51a5a6683ae28p-8, -0x1.2b2a4d5aca1f1p-16, 0x1.4c9089cfdc6b8p-26, 0x1.2a8fa42ccbafep-36, -0x1.8feafc5d512a4p-47, -0x1.3406bf2376b0fp-54, -0x1.96ade169f8587p-64, 0x1.af16b1a40412cp-74},
    {-0x1.093d814d1dc44p-1, 0x1.4cfce5388ee04p-8, -0x1.2935b5b31630ep-16, 0x1.4ee3a91c5c9a5p-26, 0x1.2889965d46a08p-36, -0x1.ad0df6251b1abp-47, -0x1.3989ebce5c4f6p-54, -0x1.8f91afce69cb0p-64, 0x1.df39a7aacf0ccp-74},
    {-0x1.018467ffdb9f9p-1, 0x1.460e7f4348fd8p-7, -0x1.26405046a3a61p-14, 0x1.5258658d8371ep-23, 0x1.253b6abbf638ap-32, -0x1.d9b76bc45eb77p-42, -0x1.419cb0e405a2dp-48, -0x1.837310ff7982cp-57, 0x1.14f2bfc2c67b9p-65},
    {-0x1.eef1234068943p-2, 0x1.3cec6b2692c0ep-7, -0x1.2242713deeb34p-14, 0x1.56e3ad506e801p-23, 0x1.204eee02f2126p-32, -0x1.0b7f8f441f10dp-41, -0x1.4bf4b04222166p-48, -0x1.70878c5c94fa6p-57, 0x1.48e29401c095dp-65},
    {-0x1.db6a972994017p-2, 0x1.33ea7c384eddcp-7, -0x1.1e370f1daa101p-14, 0x1.5b5a0d892c174p-23, 0x1.1ac6dc3674b4cp-32, -0x1.2b15177b777b9p-41, -0x1.55bcb89f17d15p-48, -0x1.5a4a5aff07bb2p-57, 0x1.7f1f293611a90p-65},
    {-0x1.c873260ced812p-2, 0x1.2b091d8c935c2p-7, -0x1.1a1e6c4b40b77p-14, 0x1.5fb90ea16bd08p-23, 0x1.149ea07fd42cap-32, -0x1.4b8db4a4483c7p-41, -0x1.5edd1191b327fp-48, -0x1.4098920d43950p-57, 0x1.b76a5ef2ed5f8p-65},
    {-0x1.b608c39c200eap-2, 0x1.2248b80667bfep-7, -0x1.15f8d2adc6002p-14, 0x1.63fe274901ce6p-23, 0x1.0dd1f4ab9ee4ep-32, -0x1.6cd8888d42ed7p-41, -0x1.673d1cb551873p-48, -0x1.23538ae4dc0f3p-57, 0x1.f179b771ce27bp-65},
    {-0x1.a4295d0dc3533p-2, 0x1.19a9b21ae827ap-7, -0x1.11c693e1347c2p-14, 0x1.6826bdc764176p-23, 0x1.065cecb294dcep-32, -0x1.8ee25ca464675p-41, -0x1.6ec3764534abfp-48, -0x1.0261ad1277df8p-57, 0x1.167aec3ac4f70p-64},
    {-0x1.92d2d9462b992p-2, 0x1.112c6f92df003p-7, -0x1.0d880965726a0p-14, 0x1.6c30297c09f73p-23, 0x1.fc78053ddb580p-33, -0x1.b19591354496ap-41, -0x1.75561b571cbfap-48, -0x1.bb5e7e59f5999p-58, 0x1.34bd17871c027p-64},
    {-0x1.820319042b81ap-2, 0x1.08d1514af9591p-7, -0x1.093d94c88d08ep-14, 0x1.7017b48efba38p-23, 0x1.ead84577fc8d0p-33, -0x1.d4da1081fe787p-41, -0x1.7ada95dd4b289p-48, -0x1.6a5e78638c862p-58, 0x1.534a5e637a280p-64},
    {-0x1.69c37441c2b8fp-2, 0x1.f912e7b18768cp-7, -0x1.02b8787b21d46p-12, 0x1.75ad76d0b8250p-20, 0x1.cdcd8b5363f45p-29, -0x1.054b33fbbb93ap-35, -0x1.80ec4897753fep-42, -0x1.c4b50c12c447fp-52, 0x1.811c20180d6fap-56},
    {-0x1.4b320f5951cd0p-2, 0x1.d948db56d33bfp-7, -0x1.f3c2d60c4ac19p-13, 0x1.7c8f3ffcb0a40p-20, 0x1.a2250a168e3adp-29, -0x1.2995d16c6bc01p-35, -0x1.84594ec3d278cp-42, -0x1.2ca8044717f0ep-55, 0x1.bcb08038c1442p-56},
    {-0x1.2e9466bf1b223p-2, 0x1.ba9c2ecc3e5b0p-7, -0x1.e1c66aee9f1ecp-13, 0x1.82b70ae7ecb87p-20, 0x1.70ce203414badp-29, -0x1.4deb5473bcae9p-35, -0x1.81b306229c79cp-42, 0x1.b34a78720e1d5p-52, 0x1.f427912632431p-56},
    {-0x1.13d87ecd5b5f9p-2, 0x1.9d117e7ed5cb7p-7, -0x1.cf84efc0fbf52p-13, 0x1.880e2a34b9062p-20, 0x1.39d2f9d048255p-29, -0x1.71b0b6cf233bep-35, -0x1.783816e191aedp-42, 0x1.e02f23b49f1d3p-51, 0x1.124d020efab12p-55},
    {-0x1.f5d82dc999791p-3, 0x1.80acca6096ba6p-7, -0x1.bd08b30c0b3f3p-13, 0x1.8c7e4b7dda7c2p-20, 0x1.faaeb1a31187ep-30, -0x1.9439e44b2ec3cp-35, -0x1.673fc8ead6ca0p-42, 0x1.7e453b89a5bacp-50, 0x1.2573ad9e9c7fbp-55},
    {-0x1.c7796775409e1p-3, 0x1.657165315a6a9p-7, -0x1.aa5d0aff2d39ap-13, 0x1.8ff1e2bdad4f7p-20, 0x1.773614503f7ffp-30, -0x1.b4cc8b8363e53p-35, -0x1.4e44ebfe64f9cp-42, 0x1.0a35cb100450ap-49, 0x1.31ecc6debd603p-55},
    {-0x1.9c6f555b5beafp-3, 0x1.4b61e45d17ddfp-7, -0x1.978e4972431f9p-13, 0x1.92549e8d9b67dp-20, 0x1.d3ef5705c4157p-31, -0x1.d2a3f8e9ca1b1p-35, -0x1.2cf1089363e80p-42, 0x1.576786f1e8f97p-49, 0x1.362928c62ee18p-55},
    {-0x1.74945c2da9da6p-3, 0x1.328010c684471p-7, -0x1.84a9aa4b271abp-13, 0x1.9393e2d5081d1p-20, 0x1.4fb5fb05aeb16p-32, -0x1.ecf5fbbecb1d7p-35, -0x1.03273b666931dp-42, 0x1.a47a7cb11e94cp-49, 0x1.30c0462371208p-55},
    {-0x1.3e718d02d6b2fp-3, 0x1.0f64c2f92e6f2p-6, -0x1.6848ae45df348p-11, 0x1.932d1d1a97a42p-17, -0x1.35c2bf6451d85p-27, -0x1.060f8e9261686p-29, -0x1.6a0812970a7cdp-37, 0x1.0904019c4e95ep-41, 0x1.13eeb84b07000p-47},
    {-0x1.000734fb251f7p-3, 0x1.c96d2ad95a591p-7, -0x1.42ae4089a631fp-11, 0x1.8e26ad6c41187p-17, -0x1.eaa812923b487p-26, -0x1.112648f0a38c9p-29, -0x1.9927fe816226ep-39, 0x1.45142f8e9845ap-41, 0x1.89d5a35f236a3p-48},
    {-0x1.9765935cec1a3p-4, 0x1.7d630e0d5d897p-7, -0x1.1dcc011b3fda6p-11, 0x1.83d009a4bd2e4p-17, -0x1.a01bcc0220fb2p-25, -0x1.0efbe20c9f30bp-29, 0x1.8f919a8b34cd9p-38, 0x1.67c778736dca7p-41, 0x1.29467c173db8fp-49},
    {-0x1.409b1cbf09397p-4, 0x1.3a6d2d3fff8a0p-7, -0x1.f442f13e11417p-12, 0x1.74349c7eb22fap-17, -0x1.227f90a26574fp-24, -0x1.fc573b1d1fc7fp-30, 0x1.02cc030b938f2p-36, 0x1.67a80e64e87d7p-41, -0x1.3a9c703df9af3p-49},
    {-0x1.f2ea0451cba5dp-5, 0x1.002e0deadf4bep-7, -0x1.b0537a42d4de5p-12, 0x1.5fa87634409ebp-17, -0x1.6d61a570713a7p-24, -0x1.bd7583e508619p-30, 0x1.990832c9dddf5p-36, 0x1.3fcb8f9fbd2d0p-41, -0x1.de8378f153980p-48},
    {-0x1.7fa8bab7b746cp-5, 0x1.9c556b3f0bfbdp-8, -0x1.70a92aeaccb72p-12, 0x1.46c81cd6f88b8p-17, -0x1.ac584de2c8833p-24, -0x1.649547d0f02c1p-30, 0x1.0aa2d5800d27fp-35, 0x1.e384c09858b77p-42, -0x1.7b508e2ec03e3p-47},
    {-0x1.2378b318ff16cp-5, 0x1.479ca233b912cp-8, -0x1.35ffff30917dep-12, 0x1.2a706c9184c97p-17, -0x1.dbc942cda8695p-24, -0x1.f01f07cf3a9afp-31, 0x1.3436f9d0af5efp-35, 0x1.0c522475cb155p-42, -0x1.d7d3fdb1fb815p-47},
    {-0x1.b5603055b1f0dp-6, 0x1.00deb302ea883p-8, -0x1.00e5aea8d98e9p-12, 0x1.0baedeb89db09p-17, -0x1.f94b1ff7d0fe6p-24, -0x1.01413f907f198p-31, 0x1.4445bc4c8ba19p-35, 0x1.6e52392169ce1p-46, -0x1.f2325e79e88f4p-47},
    {-0x1.157a5dee91099p-6, 0x1.5b9dadb9c7363p-8, -0x1.78b79bedb59d8p-11, 0x1.b71fecb427b29p-15, -0x1.0128380705364p-19, 0x1.89afa81c54286p-28, 0x1.2a739d30bfb37p-29, -0x1.4478731412747p-35, -0x1.9647a0b544f86p-39},
    {-0x1.20ca757a34e1fp-7, 0x1.88ed3235e31bfp-9, -0x1.d74fda2b16a4ep-12, 0x1.3b2b6c4a5bd9cp-15, -0x1.d383a8b61622ap-20, 0x1.d9b07730b5b3dp-26, 0x1.7f9db27f3af48p-30, -0x1.2b85413a4e092p-34, -0x1.fb4a2d262068bp-41},
    {-0x1.1c8c5555ca7bfp-8, 0x1.a2c29f114ef41p-10, -0x1.1426387512681p-12, 0x1.a22c5e96a0d52p-16, -0x1.784d8d855b555p-20, 0x1.49f3f23eaf574p-25, 0x1.c98a95b53b75ap-32, -0x1.2310d98580c16p-34, 0x1.1e3a0d3e003bbp-40},
    {-0x1.090d201df5328p-9, 0x1.a427eeab1eea2p-11, -0x1.2ea9184e54e13p-13, 0x1.00548bd4d619ap-16, -0x1.0f184d79b4e52p-20, 0x1.4a112d31cdf7cp-25, -0x1.86eb3c89fbd06p-32, -0x1.6e6b61082fa7ap-35, 0x1.03184b49709ccp-39},
    {-0x1.d250071dfcd5bp-11, 0x1.8c7c635d427dap-12, -0x1.360defc2d200fp-14, 0x1.228ee2d6ca864p-17, -0x1.6080a232f9eefp-21, 0x1.0e72261587a38p-25, -0x1.93a0fa528ad93p-31, -0x1.d19f1a16f3ba8p-37, 0x1.bbffe5411ed43p-40},
    {-0x1.831407d9a7e95p-12, 0x1.5fb2319af6024p-13, -0x1.28dc8c7cb5a72p-15, 0x1.30ed3d54f37edp-18, -0x1.a077a2c63670ep-22, 0x1.7d616cfd4f0f5p-26, -0x1.a5fbd8cbd010ep-31, 0x1.ba9f1d55a8167p-38, 0x1.d8edbbb70b11ap-41},
    {-0x1.2f051a65b3d72p-13, 0x1.2526cf65a8058p-14, -0x1.09b5a982f7387p-16, 0x1.28bc4bdbfa22dp-19, -0x1.c19bd6f9449e3p-23, 0x1.dad0e480145e5p-27, -0x1.4fe619e5a98a9p-31, 0x1.f250f857a0bb4p-37, 0x1.a254a18b03c74p-43},
    {-0x1.bf3a7383e9134p-15, 0x1.cb284b6aaed81p-16, -0x1.bcd578224e819p-18, 0x1.0c307f4661eb7p-20, -0x1.bd99854278de6p-24, 0x1.090f23bbda894p-27, -0x1.bfe47b708375cp-32, 0x1.e909f4d424483p-37, -0x1.7a2fb5c004d4bp-43},
    {-0x1.66a63a5c5cc49p-17, 0x1.904c162c84690p-17, -0x1.a95210a91597ap-18, 0x1.1c98b610b61e5p-19, -0x1.0af44b1efc902p-21, 0x1.707aa65a04dd9p-24, -0x1.7bc711138eba0p-27, 0x1.1eb5238787c5cp-30, -0x1.18d6922f113cep-34},
    {-0x1.110576b9a1bf4p-20, 0x1.51865c55fccecp-20, -0x1.90cfaabd9ba93p-21, 0x1.2f3f372037de0p-22, -0x1.46b494506230dp-24, 0x1.08d50c9cae195p-26, -0x1.4c463c6aa52c1p-29, 0x1.496f12aab779ap-32, -0x1.e966b3e1b833fp-36},
    {-0x1.46a16da817198p-24, 0x1.bb4a8814ac789p-24, -0x1.22e8ee50b33a4p-24, 0x1.eaa07713fd7d4p-26, -0x1.29ba9ccc4b6f1p-27, 0x1.13ad6c553d831p-29, -0x1.93350bb02e56ap-32, 0x1.e59644da7c079p-35, -0x1.c7ae965fd44a4p-38},
    {-0x1.32a35e3ed86a6p-28, 0x1.c56b1317dacbdp-28, -0x1.45e500ec51ae7p-28, 0x1.2ee06c98e6b39p-29, -0x1.98383c9448a0bp-31, 0x1.a79a3ca3c1f89p-33, -0x1.5fc37d45c562dp-35, 0x1.eee9e35f86d60p-38, -0x1.12598805f9e7bp-40},
    {-0x1.c34c28f42587cp-33, 0x1.693085d27541ep-32, -0x1.1a2dff9ba12cep-32, 0x1.1e6d2ed6581a4p-33, -0x1.a802c9fd66d54p-35, 0x1.e63c5db2c425ap-37, -0x1.c26edae7fac1dp-39, 0x1.69aa8a12b2a99p-41, -0x1.c973bd4cf2878p-44},
    {-0x1.041789eb78bb0p-37, 0x1.c0277054d597bp-37, -0x1.7a218946b1d2ep-37, 0x1.a019554fab7cdp-38, -0x1.4f4e5597a6146p-39, 0x1.a45b07a98cbd1p-41, -0x1.acd897a1850c9p-43, 0x1.82fcf54125cc0p-45, -0x1.1135acc074208p-47},
    {-0x1.d53e3e82dad22p-43, 0x1.b10ea310b66a1p-42, -0x1.8875ccf9a1cdap-42, 0x1.d146489f854a6p-43, -0x1.9541f29933f33p-44, 0x1.134c2a508f171p-45, -0x1.32498914a51d2p-47, 0x1.333e452a94dadp-49, -0x1.dd0ffaaba83f9p-52},
    {-0x1.4b13ea9a9f5d0p-48, 0x1.45e730ce79a5bp-47, -0x1.3bb8c58d95591p-47, 0x1.911f4bae0e0c1p-48, -0x1.775d7ac54ff52p-49, 0x1.125ad8986a8fdp-50, -0x1.4a4f09804e4e0p-52, 0x1.6d57c1e286a4dp-54, -0x1.347523c1fc7a4p-56},
    {-0x1.5dbbaccf1a4e0p-57, 0x1.75474ab08fab4p-55, -0x1.8f8d7573271cep-54, 0x1.227db36d8b34bp-53, -0x1.2485c73998cefp-53, 0x1.8874777f85b81p-54, -0x1.1e0eacf944552p-54, 0x1.11e9961103e86p-54, -0x1.e1177bdb5958dp-56},
    {-0x1.3d2d60a5c14aap-70, 0x1.72e892a332377p-68, -0x1.c226c43a4d6d8p-67, 0x1.825ad1b4a76f7p-66, -0x1.a7ecb02437186p-66, 0x1.07f1231375e15p-66, -0x1.de48ecf979d38p-67, 0x1.428237c06c977p-66, -0x1.3bce89806b570p-67},
    {-0x1.abbbd1ab8143dp-85, 0x1.080351f7f294ep-82, -0x1.6f6d57dbee1ccp-81, 0x1.8161a5e51ca11p-80, -0x1.bb7051d9c78b8p-80, 0x1.9a5db6886ea6fp-81, -0x1.02cee884700cap-80, 0x1.04efe571c1e85p-79, -0x1.178a1389d2bd9p-80},
    {-0x1.ac170cfdee0fcp-101, 0x1.03253fb444d98p-98, -0x1.b136afceaff8ep-97, 0x1.273dc4cb76442p-95, -0x1.54adf532645c0p-95, 0x1.9257cd2cfca66p-98, -0x1.57daf93a4fcfep-96, 0x1.26ff61f2ffc39p-94, -0x1.54cfba37dc1a0p-95},
    {-0x1.3d880d577329bp-118, 0x1.3947c0190a32dp-116, -0x1.6d9796d042c2ap-114, 0x1.61e0ac4b7b4ccp-112, -0x1.878f34ce4c54bp-112, -0x1.127f512731b9dp-113, -0x1.bcbf1a6d59ccap-114, 0x1.d76225ffa7073p-111, -0x1.225bbd4e515b1p-111},
    {-0x1.5cbaff5bbf4b5p-137, 0x1.cedb5c260159ep-137, -0x1.b0478edf146d6p-133, 0x1.4d943e8b425f7p-130, -0x1.567d12f7b5b14p-130, -0x1.5b3263d471628p-130, 0x1.0b8131edd1b32p-133, 0x1.0c5f9b1053b2fp-128, -0x1.5d5dbe357aec3p-129},
    {-0x1.1b4c176e39cf0p-157, -0x1.5813a471f8eecp-155, -0x1.534225c0a6560p-153, 0x1.e9f3dcfd0cbf7p-150, -0x1.cd792482815dfp-150, -0x1.964415cb4db14p-149, 0x1.1962c6be470c6p-150, 0x1.b634f57ad3f77p-148, -0x1.2b29084f66bf7p-148},
    {-0x1.543b1842c894ep-179, -0x1.830b2d85e725cp-175, -0x1.260ed7172af27p-175, 0x1.1360c04c11d98p-170, -0x1.e0581e43b5af3p-171, -0x1.28e84f37b9446p-169, 0x1.1a19a937e0301p-170, 0x1.01b73ec694d52p-168, -0x1.6ec943d8a79d4p-169},
    {-0x1.14be5c581bcb8p-214, -0x1.349d1d9d09abdp-194, 0x1.1ae1279485f2ap-194, 0x1.e28d195a1d6f3p-191, -0x1.bb90a2921388bp-191, -0x1.6abf4a2cc9dd8p-189, 0x1.4f972175f107dp-189, 0x1.384aeedc6a8b8p-189, -0x1.26a295a1b5c05p-189},
    {-0x1.02d64a74714f3p-266, -0x1.2b00881b79976p-243, 0x1.19dff59ebc4eap-243, 0x1.d18d06aea4e24p-240, -0x1.b7c212a1d2ab2p-240, -0x1.59f4118bd67f0p-238, 0x1.4888dbbc63414p-238, 0x1.21ce93a13588ap-238, -0x1.16d58209ff3e0p-238},
    {-0x1.1ee774feb43cdp-324, -0x1.49da6139a5334p-298, 0x1.3d19b8760e219p-298, 0x1.000066cc88281p-294, -0x1.ecefd80deb9c7p-295, -0x1.7934acb7b78d0p-293, 0x1.6ca887422c48ep-293, 0x1.362fa828a49dfp-293, -0x1.2e96e9876d6a2p-293},
    {-0x1.78249810921d5p-388, -0x1.a29b68aca9f69p-359, 0x1.980ef2d38a01ap-359, 0x1.4425437979948p-355, -0x1.3c53ed9963f67p-355, -0x1.da92e2f7b8d42p-354, 0x1.d09273d73eba1p-354, 0x1.8136666676149p-354, -0x1.7b86613f2f095p-354},
    {-0x1.2334bd061877ep-457, -0x1.33928486a0099p-425, 0x1.2eda134345b8ap-425, 0x1.db877d6b514dfp-422, -0x1.d49f1e4b9605bp-422, -0x1.5a75e52eded72p-420, 0x1.563a751d3cd9cp-420, 0x1.169a22e59c8a3p-420, -0x1.147ca0e703511p-420},
    {-0x1.09f504c96d386p-532, -0x1.06cf01c63eabep-497, 0x1.04b10db1d1f5dp-497, 0x1.95cf128733ef1p-494, -0x1.92cb0d4daf182p-494, -0x1.269e9647e5b9cp-492, 0x1.24f26be727ca1p-492, 0x1.d695745f31197p-493, -0x1.d5879dde12783p-493},
    {-0x1.1e485467b3945p-613, -0x1.05e646fdaf8bcp-575, 0x1.0537d5efb7c1dp-575, 0x1.94057214248a3p-572, -0x1.932982a2fcc40p-572, -0x1.248aa896c96c1p-570, 0x1.244f75a5b8431p-570, 0x1.d0e093c5133b0p-571, -0x1.d1b6330817bb6p-571},
    {-0x1.6af0e389373a7p-700, -0x1.30fd46d78dc36p-659, 0x1.31737aa335f40p-659, 0x1.d627da7f2230cp-656, -0x1.d709d2e6a5524p-656, -0x1.53bdc65a4a820p-654, 0x1.54b9e996ffad0p-654, 0x1.0ce66b52a5132p-654, -0x1.0e352426ab25bp-654},
    {-0x1.a22175280770ep-841, -0x1.12cec0a25e3cbp-749, 0x1.16fca6c747089p-749, 0x1.a69c2b6524a5ap-746, -0x1.ad0a15930c435p-746, -0x1.2f52ac54110c0p-744, 0x1.33f12f6812aa7p-744, 0x1.d9fbe35c6c453p-745, -0x1.e1374074c0179p-745},
    {-0x0.00000037b23b8p-1022, -0x1.7768a0d8235bcp-946, 0x1.7d2b13de407b1p-946, 0x1.20a70070274bap-942, -0x1.2514dc9232aaep-942, -0x1.9e534e92669f3p-941, 0x1.a4af7254e95f7p-941, 0x1.43adcf492b575p-941, -0x1.48a6ac773c7a1p-941},
    {-0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {-0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {-0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {-0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {-0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {-0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
    {0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0, -0x0.0p+0, 0x0.0p+0, 0x0.0p+0, -0x0.0p+0},
  };
  return &t[0][0];
}
#define PGB_LN_TN_ROWS 105
#define PGB_LN_TP_ROWS 105
PGB_HD double pgb_log_ndtr_t(double x, const double* tn, const double* tp) {
  if (!(x == x)) return x;
  const int neg = x < 0.0;
  const double z = neg ? -x : x;
  /* dyadic interval and local variable from the bits of z (every step below is exact); the same
   * for both signs -- the sign only selects the table */
  const uint64_t zb = pgb_d2u(z);
  const int e = (int)((zb >> 52) & 0x7FF) - 1023;
  const int ec = e > 9 ? 9 : e;
  const int sub = (int)((zb >> 49) & 7);
  const int in = e < -3 ? 0 : 1 + (ec + 3) * 8 + sub;
  /* local variable u in [-1, 1): with t = (mantissa bits below `sub`) / 2^52 in [0, 1/8), u = 16 t - 1.  The 49
   * bits are moved up by 3 under the exponent of 1.0, which is the double 1 + 8 t, and u = 2 (1 + 8 t) - 3 in
   * one fma -- exact, like the textbook form (m - (1 + sub/8)) * 16 - 1 with m the mantissa in [1, 2): every
   * intermediate of either form is representable, so both give the same bits; this one is 5 operations. */
  const double m8 = pgb_u2d(((zb & 0x0001FFFFFFFFFFFFull) << 3) | 0x3FF0000000000000ull); /* 1 + 8 t */
  const double u = e < -3 ? z * 16.0 - 1.0 : PGB_FMA(2.0, m8, -3.0);
  const double* c = (neg ? tn : tp) + in * 9;
  double g = c[8];
  g = PGB_FMA(g, u, c[7]);
  g = PGB_FMA(g, u, c[6]);
  g = PGB_FMA(g, u, c[5]);
  g = PGB_FMA(g, u, c[4]);
  g = PGB_FMA(g, u, c[3]);
  g = PGB_FMA(g, u, c[2]);
  g = PGB_FMA(g, u, c[1]);
  g = PGB_FMA(g, u, c[0]);
  if (neg) {
    if (e > 9) { /* far tail: asymptotic series of the Mills ratio */
      const double w = 1.0 / (z * z);
      g = (-pgb_log(z) - 0.91893853320467274178) + (2.5 * w * w - w);
    }
    return -0.5 * (z * z) + g;
  }
  return z < 8.5 ? g : 0.0;
}

PGB_HD double pgb_log_ndtr(double x) { return pgb_log_ndtr_t(x, pgb_ln_tn(), pgb_ln_tp()); }

/* log(1 + e^t) */
PGB_HD double pgb_softplus(double t) {
  if (t > 36.0) return t;
  return pgb_log(1.0 + pgb_exp(t));
}

/* Per-row log-likelihood of the closed families with one linear predictor mu (K = 1).
 * y is the observed response (0/1 for the Bernoulli families).  Clamped to [-2047, 0] so that
 * n terms fit the fixed-point accumulator (scale cl). */
/* `param`: the family's scalar parameter (NEGBIN_LOG: alpha), 0 otherwise.
 * The count families are written relative to the saturated model (minus half the deviance):
 *   POISSON_LOG:  (y mu - e^mu) - (y log y - y)
 *   NEGBIN_LOG:   (y mu - (alpha + y) log(alpha + e^mu)) - (y log y - (alpha + y) log(alpha + y))
 * The subtracted terms depend on the data (and alpha) only, so they cancel in every particle
 * weight; they make the value a quantity <= 0 that fits the fixed-point range like a log-pmf. */
/*   ASYMLAPLACE(b, q):  -rho_q((y - mu) / b),  rho_q(u) = u (q - [u < 0])   (the check loss)
 *   STUDENT_T(sigma, nu): -((nu + 1) / 2) log(1 + ((y - mu) / sigma)^2 / nu)
 *   GAMMA_LOG(alpha):    -alpha (y e^-mu + mu - 1 - log y)   (relative to the saturated model)
 * (both without their mu-free normalising terms, hence <= 0). */
/* The Bernoulli families on the SIGNED predictor s = mu for y = 1, -mu for y = 0 (the response only picks the
 * sign): callers that evaluate one row for many particles flip the sign with the row's precomputed mask. */
PGB_HD double pgb_loglik_bern_s(int family, double smu, const double* tn, const double* tp) {
  double ll = family == PGB_FAMILY_BERNOULLI_PROBIT ? pgb_log_ndtr_t(smu, tn, tp) : -pgb_softplus(-smu);
  if (!(ll > -2047.0)) ll = -2047.0;
  if (ll > 0.0) ll = 0.0;
  return ll;
}
PGB_HD double pgb_loglik1q(int family, double y, double mu, double param, double param2, const double* tn,
                           const double* tp) {
  double ll;
  if (family == PGB_FAMILY_CALLBACK) return 0.0; /* evaluated on the host, never here */
  if (family == PGB_FAMILY_POISSON_LOG || family == PGB_FAMILY_NEGBIN_LOG) {
    const double yy = y > 0.0 ? y : 0.0;
    const double em = pgb_exp(mu);
    if (family == PGB_FAMILY_POISSON_LOG) {
      const double sat = yy > 0.0 ? yy * pgb_log(yy) - yy : 0.0;
      ll = (yy * mu - em) - sat;
    } else {
      const double ay = param + yy;
      const double sat = yy > 0.0 ? yy * pgb_log(yy) - ay * pgb_log(ay) : -(param * pgb_log(param));
      ll = (yy * mu - ay * pgb_log(param + em)) - sat;
    }
  } else if (family == PGB_FAMILY_GAMMA_LOG) {
    /* -alpha (y e^-mu + mu) minus its maximum over mu, -alpha (1 + log y) */
    const double yy = y > 1.0e-300 ? y : 1.0e-300;
    ll = -param * (((yy * pgb_exp(-mu) + mu) - 1.0) - pgb_log(yy));
  } else if (family == PGB_FAMILY_ASYMLAPLACE) {
    const double u = (y - mu) / param;
    ll = -(u * (u < 0.0 ? param2 - 1.0 : param2));
  } else if (family == PGB_FAMILY_STUDENT_T) {
    const double u = (y - mu) / param;
    ll = (-0.5 * (param2 + 1.0)) * pgb_log(1.0 + (u * u) / param2);
  } else {
    return pgb_loglik_bern_s(family, y > 0.5 ? mu : -mu, tn, tp);
  }
  if (!(ll > -2047.0)) ll = -2047.0;
  if (ll > 0.0) ll = 0.0;
  return ll;
}
PGB_HD double pgb_loglik1p(int family, double y, double mu, double param, const double* tn, const double* tp) {
  return pgb_loglik1q(family, y, mu, param, 1.0, tn, tp);
}
PGB_HD double pgb_loglik1_t(int family, double y, double mu, const double* tn, const double* tp) {
  return pgb_loglik1p(family, y, mu, 0.0, tn, tp);
}
PGB_HD double pgb_loglik1(int family, double y, double mu) {
  return pgb_loglik1_t(family, y, mu, pgb_ln_tn(), pgb_ln_tp());
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

/* the contract's range for a callback's per-row value (NaN -> the lower bound) */
PGB_HD double pgb_clamp_loglik(double ll) {
  if (!(ll > -2047.0)) ll = -2047.0;
  if (ll > 2047.0) ll = 2047.0;
  return ll;
}

/* Per-row log-likelihood of every non-Normal(sigma) family at the K linear predictors mu. */
PGB_HD double pgb_loglikq(int family, int K, double y, const double* mu, double param, double param2) {
  if (family == PGB_FAMILY_CATEGORICAL) return pgb_loglik_cat(K, y, mu);
  if (family == PGB_FAMILY_NORMAL_MEANSCALE) return pgb_loglik_meanscale(y, mu);
  return pgb_loglik1q(family, y, mu[0], param, param2, pgb_ln_tn(), pgb_ln_tp());
}
PGB_HD double pgb_loglikp(int family, int K, double y, const double* mu, double param) {
  return pgb_loglikq(family, K, y, mu, param, 1.0);
}
PGB_HD double pgb_loglik(int family, int K, double y, const double* mu) {
  return pgb_loglikp(family, K, y, mu, 0.0);
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

/* ------------------------------------------------------------------ linear response */
/* [U] draw_leaf_value / fast_linear_fit with response = "linear": a new leaf predicts
 *     value + slope * (x - xbar),   x = its rows' value of the PARENT's split variable,
 * value being the constant-response leaf value (mean(sum_trees)/m + noise) and slope the OLS
 * slope of sum_trees/m on x over the leaf's rows (0 with fewer than 3 rows or no spread).
 * The row pass works with u = x * 2^-ex (ex: exponent bound of the column, |u| <= 1) and
 * reduces, next to the usual sums, the fixed-point sums
 *     q_u = sum q(u R), q_uu = sum q(u^2 R), q_us = sum q(u sum_trees), q_ur = sum q(u r)
 * all at scale c1 (R = 2^(range_exp - 1) lifts u and u^2 to the resolution of the others). */
typedef struct {
  double slope_u; /* slope with respect to u            */
  double ubar;    /* mean of u over the leaf's rows     */
  double var_u;   /* sum (u - ubar)^2                   */
} pgb_linfit;

PGB_HD pgb_linfit pgb_lin_fit(int64_t cnt, int64_t q_u, int64_t q_uu, int64_t q_us, int64_t q_st,
                              double inv_c1, double inv_R, double m) {
  pgb_linfit f;
  f.slope_u = 0.0;
  f.ubar = 0.0;
  f.var_u = 0.0;
  if (cnt < 3) return f;
  const double nn = (double)cnt;
  const double su = ((double)q_u * inv_c1) * inv_R;
  const double suu = ((double)q_uu * inv_c1) * inv_R;
  const double sus = (double)q_us * inv_c1;
  const double sst = (double)q_st * inv_c1;
  f.ubar = su / nn;
  const double var = suu - su * f.ubar;
  const double cov = sus - sst * f.ubar;
  if (!(var > 1.0e-12)) return f; /* no spread (or cancellation noise): constant leaf */
  f.var_u = var;
  f.slope_u = (cov / var) / m;
  return f;
}
/* SSE of the leaf's rows under the linear prediction, from the constant-leaf SSE:
 * sum (r - value - b (u - ubar))^2 = sse_const - 2 b sum r (u - ubar) + b^2 var  (sum (u - ubar) = 0) */
PGB_HD double pgb_lin_sse(double sse_const, pgb_linfit f, int64_t q_ur, int64_t q_r, double inv_c1) {
  const double cru = (double)q_ur * inv_c1 - f.ubar * ((double)q_r * inv_c1);
  double sse = (sse_const - (2.0 * f.slope_u) * cru) + (f.slope_u * f.slope_u) * f.var_u;
  return sse;
}
/* exponent bound of a column: smallest ex >= 0... any integer ex with max|x| <= 2^ex */
PGB_HD int pgb_col_exponent(double amax) {
  if (!(amax > 0.0)) return 0;
  int e = (int)((pgb_d2u(amax) >> 52) & 0x7FF) - 1023 + 1; /* amax < 2^e */
  if (e < -1000) e = -1000;
  return e;
}
/* per-row prediction of a leaf */
PGB_HD double pgb_leaf_pred(double value, double slope, double xbar, double x) {
  return value + slope * (x - xbar);
}

/* leaf_sd after the tree update number `iter` (1-based) whose accepted tree added `qstd` = sum over the rows of
 * quant(running sd of the accepted trees' predictions) to the tuning statistics ([U] RunningSd.update; adopted from
 * the third update on).  A running sd of EXACTLY 0 -- every accepted prediction so far is the same constant, e.g.
 * the untouched stump won the first updates -- is not adopted: leaf values are mean(sum_trees)/m + N(0,1) leaf_sd,
 * so leaf_sd = 0 would never let a leaf move again (DESIGN.md deviation 12).  ONE definition for the oracle, the
 * control kernel, the likelihood pass (which re-derives leaf values), the K-vector outputs and pgb_get_state. */
PGB_HD double pgb_tuned_leaf_sd(double current, int64_t iter, int64_t qstd, double inv_c1, int64_t n) {
  /* (the zero test is made on the double, after the division: a 64-bit integer compare on qstd in the condition made
   *  the control kernel wait for its statistics loads one branch earlier -- +0.34 us per launch, A/B on one box) */
  double sd = current;
  if (iter > 2) sd = ((double)qstd * inv_c1) / (double)n;
  return sd > 0.0 ? sd : current;
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
