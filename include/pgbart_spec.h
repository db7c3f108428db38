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
/* particles: one per lane of a wave64, or -- library built with -DPGB_MAX_PARTICLES=128 -- two per lane (the
 * control kernel finishes lanes' particles q and q + 64 one after the other, the cumulative weights are two
 * 64-entry scans chained by the first block's total: pgb_weights_scan).  A chain of <= 64 particles draws the same
 * numbers from either build. */
#ifndef PGB_MAX_PARTICLES
#define PGB_MAX_PARTICLES 64
#endif
#define PGB_MAX_OUTPUTS 16 /* K-vector leaves: the run-time-K kernels work in tiles of 4 outputs, nothing is sized by it but small records */
#define PGB_SELECT_TRIES 16   /* redraws of the split row when X[row,var] is NaN */

/* split rules (reference names: tests/test_bart.py:143-145, bart.py:100-103) */
#define PGB_RULE_CONTINUOUS 0 /* go left iff x <= v  */
#define PGB_RULE_ONEHOT 1     /* go left iff x == v  */
#define PGB_RULE_SUBSET 2     /* go left iff category x is in the set v (bart.py:100-103)   */
#define PGB_SUBSET_BITS 52    /* categories are integer codes 0..51; the set is a bit mask  */

/* Upstream-semantics switches (pgb_settings.compat).  The default sampler (compat = 0) differs from pymc-bart
 * <= 0.12 as SURVEY.md Appendix A recalls it in two ways that change the sampled distribution (DESIGN.md section
 * 0, deviations 2 and 13); each bit puts upstream's behaviour back, on every backend alike, so that whoever can
 * run the reference binary (bartrs is not in the reference tree) can tell which semantics it follows.
 *   bit 0  a particle that has never grown (a root-only tree) keeps log-weight 0 -- upstream initialises
 *          ParticleTree.log_weight = 0 and calls update_weight only after a successful grow -- instead of the
 *          likelihood of its stump.  The reference particle keeps its likelihood, as upstream (init_particles).
 *   bit 1  a one-hot split whose right child would be empty (every row of the leaf holds the split value) is
 *          grown, with an empty right leaf that predicts 0, instead of failing.  (The subset rule still fails:
 *          upstream redraws the subset until it is proper.)                                                   */
#define PGB_COMPAT_FRESH_WEIGHT_ZERO 1
#define PGB_COMPAT_ONEHOT_EMPTY_CHILD 2
#define PGB_COMPAT_ALL 3

/* Largest |offset| of a linear predictor pgb_set_offset accepts.  The table-driven exp / log-Phi of the per-row
 * families (pgb_exp_t, pgb_lphi_t) take their table index from the bits of the argument without a clamp; they are
 * exact in their saturation (0 / inf) only for |x| < 4.6e7, beyond which the index wraps.  sum_trees is bounded by
 * the fixed-point range (|sum_trees| < 2^range_exp <= 2^20), so bounding the offset bounds the argument. */
#define PGB_MAX_OFFSET 1.0e6

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
/* A partition that sent no row right: does the grow fail (the node stays a leaf)?  Never under the continuous
 * rule (upstream grows the empty child); always under the subset rule; under the one-hot rule unless
 * PGB_COMPAT_ONEHOT_EMPTY_CHILD asks for upstream's behaviour as recalled. */
PGB_HD int pgb_empty_right_fails(int rule, int compat) {
  return rule == PGB_RULE_SUBSET || (rule == PGB_RULE_ONEHOT && !(compat & PGB_COMPAT_ONEHOT_EMPTY_CHILD));
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
/* The per-row log-likelihoods (every family but Normal(sigma)) are evaluated once per (row, particle, round):
 * at n = 1 M rows and 40 particles the likelihood pass is the dominant kernel, and what bounds it
 * is the NUMBER OF VECTOR INSTRUCTIONS of one evaluation.  The three functions below are therefore written for
 * few operations on a machine without fp64 division -- table look-ups addressed by the bits of the argument, short
 * Horner chains of explicit fma -- and still only use + - * fma and integer operations on the bit patterns, so
 * that the x86-64 compile (oracle) and the gfx950 compile (product) give the same bits.  Tables: generated against
 * mpmath by tools/fit_ll_tables.py into pgbart_lltab.h; they live in accessor functions so that a kernel can
 * stage them in LDS (a per-lane row through the vector L1 costs one cache-line access per distinct row).
 * (pgb_exp / pgb_log above stay what the CONTROL path uses -- particle weights, Box-Muller: a few calls per
 * round on one wave, where staging tables would cost more than it saves.) */
#include "pgbart_lltab.h"

typedef struct {
  const double* lphi; /* pgb_tab_lphi() or a copy of it */
  const double* expt; /* pgb_tab_exp()                  */
  const double* logt; /* pgb_tab_log()                  */
} pgb_lltabs;
PGB_HD pgb_lltabs pgb_lltabs_default(void) {
  pgb_lltabs t;
  t.lphi = pgb_tab_lphi();
  t.expt = pgb_tab_exp();
  t.logt = pgb_tab_log();
  return t;
}

#if defined(__HIPCC__)
#define PGB_LDEXP(x, k) __builtin_ldexp((x), (k))
#else
/* exact scaling by 2^k: a multiplication while 2^k is a normal double, ldexp beyond (one correctly rounded
 * result either way -- what v_ldexp_f64 returns) */
PGB_HD double pgb_ldexp_host(double x, int k) {
  if (k < -1021 || k > 1023) return ldexp(x, k);
  return x * pgb_u2d((uint64_t)(uint32_t)(k + 1023) << 52);
}
#define PGB_LDEXP(x, k) pgb_ldexp_host((x), (k))
#endif

/* exp(x) = 2^k * T[j] * e^r with x = (32 k + j) ln2/32 + r, |r| <= ln2/64.
 * n = 32 k + j is the round-to-nearest-even of x * 32/ln2, read from the low word of x * 32/ln2 + 1.5 * 2^52 (one
 * fma; exact two's complement while |n| < 2^31, i.e. |x| < 4.6e7); e^r - 1 = r + r^2/2 + ... + r^6/720 (next term
 * 3.4e-18).  ~1 ulp.  No clamp: results beyond the double range are 0 / inf, a NaN stays a NaN.  17 vector
 * instructions + one table read on gfx950. */
PGB_HD double pgb_exp_t(double x, const double* T) {
  const double km = PGB_FMA(x, 46.16624130844683, 6755399441055744.0); /* 32/ln2, 1.5 * 2^52 */
  const int32_t n = (int32_t)(uint32_t)pgb_d2u(km);
  const double kf = km - 6755399441055744.0;
  double r = PGB_FMA(kf, -2.166084938653512e-02, x);     /* ln2/32: high part 0x1.62e42feep-6 (21 trailing zero bits ... */
  r = PGB_FMA(kf, -5.9631716539705866e-12, r);    /* ... so that kf * high is exact), low part */
  const double Tj = T[n & 31];
  double q = 1.3888888888888889e-03;            /* 1/720 */
  q = PGB_FMA(q, r, 8.3333333333333332e-03);    /* 1/120 */
  q = PGB_FMA(q, r, 4.1666666666666664e-02);    /* 1/24  */
  q = PGB_FMA(q, r, 1.6666666666666666e-01);    /* 1/6   */
  q = PGB_FMA(q, r, 0.5);
  q = PGB_FMA(q, r, 1.0);
  q = q * r;                                    /* e^r - 1 */
  return PGB_LDEXP(PGB_FMA(Tj, q, Tj), n >> 5);
}

/* log(x) for a POSITIVE, NORMAL, FINITE x (the callers below guarantee it; pgb_log_t adds the guards).
 * x = 2^k z with z in [0.6875, 1.375): the high word minus 0x3FE60000 gives k (its top 12 bits) and the table
 * interval (the next 7); r = z * invc - 1 (one fma, |r| < 2^-7), log z = logc + log1p(r),
 * log1p(r) = r + r^2 (-1/2 + r/3 - ... + r^5/7) (next term 1.7e-18 r).  The interval around 1 has invc = 1,
 * logc = 0: log keeps its relative accuracy there.  Absolute error < 2e-16 + 1 ulp.  No division. */
PGB_HD double pgb_log_pos_t(double x, const double* T) {
  const uint64_t ix = pgb_d2u(x);
  const uint32_t tmp = (uint32_t)(ix >> 32) - 0x3FE60000u;
  const int32_t k = (int32_t)tmp >> 20;
  const uint32_t i = (tmp >> 13) & 127u;
  const double z = pgb_u2d(ix - ((uint64_t)(tmp & 0xFFF00000u) << 32));
  const double invc = T[2 * i], logc = T[2 * i + 1];
  const double r = PGB_FMA(z, invc, -1.0);
  const double kd = (double)k;
  double p = 1.4285714285714285e-01;            /*  1/7 */
  p = PGB_FMA(p, r, -1.6666666666666666e-01);   /* -1/6 */
  p = PGB_FMA(p, r, 0.2);                       /*  1/5 */
  p = PGB_FMA(p, r, -0.25);                     /* -1/4 */
  p = PGB_FMA(p, r, 3.3333333333333331e-01);    /*  1/3 */
  p = PGB_FMA(p, r, -0.5);
  const double lo = PGB_FMA(r * r, p, r);                               /* log1p(r)      */
  const double hi = PGB_FMA(kd, 6.93147180369123816490e-01, logc);      /* k ln2 + log c */
  return PGB_FMA(kd, 1.90821492927058770002e-10, hi + lo);
}
/* log(x) for any x: NaN -> NaN, x <= 0 -> -1e300 (like pgb_log), subnormals rescaled, +inf -> +inf */
PGB_HD double pgb_log_t(double x, const double* T) {
  if (!(x == x)) return x;
  if (!(x > 0.0)) return -1.0e300;
  if (x > 1.7976931348623157e308) return x;
  double adj = 0.0;
  if (x < 2.2250738585072014e-308) { /* (one evaluation site: the rare case only moves the argument) */
    x = x * 4503599627370496.0;
    adj = -36.04365338911715; /* 52 ln2 */
  }
  return pgb_log_pos_t(x, T) + adj;
}

/* two adjacent doubles of a 16-byte aligned table as ONE 16-byte read on the device */
#if defined(__HIP_DEVICE_COMPILE__)
typedef double pgb_d2v __attribute__((ext_vector_type(2)));
#define PGB_LD2(p, a, b)                                   \
  do {                                                     \
    const pgb_d2v v_ = *(const pgb_d2v*)(p);               \
    (a) = v_.x;                                            \
    (b) = v_.y;                                            \
  } while (0)
#else
#define PGB_LD2(p, a, b) \
  do {                   \
    (a) = (p)[0];        \
    (b) = (p)[1];        \
  } while (0)
#endif

/* log Phi(s) (standard normal CDF) for the probit likelihood: ONE table entry, one Horner chain (degree
 * PGB_LPHI_DEG = 7), for either sign and any magnitude -- no division, no exp, no log, no branch, no clamp.
 *   t = |s| + 1/8 >= 1/8: the dyadic interval of t (exponent -3..5 and the top PGB_LPHI_SUBBITS = 4 mantissa bits
 *   -> row 0..143) and the local variable u in [-1, 1) (the remaining mantissa bits: u = 2 (1 + 16 frac) - 3,
 *   exact) come from the bits of t; per row and sign the table holds log Phi(+-(t - 1/8)) itself as a polynomial
 *   in u.  Everything from t = 64 on shares the last row, the clamp row: exactly -2047 for s < 0 -- the lower bound of a per-row
 *   log-likelihood; log Phi(-63.875) = -2045.08, so every other row stays above it -- and 0 for s > 0
 *   (log Phi(9) = -1.1e-19).  +-inf therefore give the limits 0 / -2047.
 *   The rounding of t is an argument perturbation of at most 2^-54 |t|: <= 1 ulp of the result.
 * Absolute error against mpmath / scipy.special.log_ndtr: < 1e-15 for s >= -1, < 1 ulp + 6e-15 on [-9, -1],
 * < 2 ulp of the result below (tests/test_spec.py).  The result is in [-2047, 1e-16].
 * The argument must not be NaN: the linear predictor of a row is a sum of finite leaf values and the offsets /
 * responses the boundary checked (pgb_set_offset, pgb_set_response refuse non-finite values); a NaN would read
 * the clamp row of ITS sign bit, which compilers do not agree on.
 * gfx950: 29 vector instructions with the sign flip, quantisation and two running sums + four 16-byte table reads
 * (before round 4: 66 and nine 8-byte reads). */
#define PGB_LPHI_J0 (1020u << PGB_LPHI_SUBBITS) /* (biased exponent of 1/8) << SUBBITS */
PGB_HD double pgb_lphi_t(double s, const double* T) {
  const uint64_t sb = pgb_d2u(s);
  const double t = pgb_u2d(sb & 0x7FFFFFFFFFFFFFFFull) + 0.125;
  const uint64_t tb = pgb_d2u(t);
  uint32_t J = (uint32_t)(tb >> (52 - PGB_LPHI_SUBBITS)); /* (biased exponent << SUBBITS) | top mantissa bits; t > 0 */
  if (J > PGB_LPHI_J0 + (PGB_LPHI_ROWS - 1)) J = PGB_LPHI_J0 + (PGB_LPHI_ROWS - 1);
  const uint32_t ent = ((J - PGB_LPHI_J0) << 1) | (uint32_t)(sb >> 63);
  const double* c = T + 2 * ent;
  /* the mantissa bits below the row bits, moved up under the exponent of 1.0: 1 + 2^SUBBITS frac in [1, 2) */
  const double mm = pgb_u2d(((tb & ((1ull << (52 - PGB_LPHI_SUBBITS)) - 1ull)) << PGB_LPHI_SUBBITS) | 0x3FF0000000000000ull);
  const double u = PGB_FMA(2.0, mm, -3.0);
  double cf[2 * PGB_LPHI_PAIRS];
  for (int p = PGB_LPHI_PAIRS - 1; p >= 0; --p) PGB_LD2(c + p * 2 * PGB_LPHI_ENT, cf[2 * p], cf[2 * p + 1]);
  double g = cf[PGB_LPHI_DEG];
  for (int k = PGB_LPHI_DEG - 1; k >= 0; --k) g = PGB_FMA(g, u, cf[k]);
  return g;
}
PGB_HD double pgb_log_ndtr(double x) { return pgb_lphi_t(x, pgb_tab_lphi()); }

/* the contract's range of a per-row log-likelihood: n terms fit the fixed-point accumulator (scale cl).
 * NaN -> the lower bound.  (Device: v_max_f64 / v_min_f64 -- the same values; a zero may differ in sign, which
 * the fixed-point rounding that follows does not see.) */
#if defined(__HIP_DEVICE_COMPILE__)
#define PGB_CLAMP_LL(ll, hi) __builtin_fmin(__builtin_fmax((ll), -2047.0), (hi))
#else
PGB_HD double pgb_clamp_ll_host(double ll, double hi) {
  if (!(ll > -2047.0)) ll = -2047.0;
  if (ll > hi) ll = hi;
  return ll;
}
#define PGB_CLAMP_LL(ll, hi) pgb_clamp_ll_host((ll), (hi))
#endif

/* log(1 + e^t) */
PGB_HD double pgb_softplus_t(double t, const pgb_lltabs* tb) {
  if (t > 36.0) return t;
  return pgb_log_t(1.0 + pgb_exp_t(t, tb->expt), tb->logt); /* (the guarded form: a NaN stays a NaN) */
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
PGB_HD double pgb_loglik_bern_s(int family, double smu, const pgb_lltabs* tb) {
  if (family == PGB_FAMILY_BERNOULLI_PROBIT) return pgb_lphi_t(smu, tb->lphi); /* in [-2047, 1e-16] by itself */
  return PGB_CLAMP_LL(-pgb_softplus_t(-smu, tb), 0.0);
}
PGB_HD double pgb_loglik1q(int family, double y, double mu, double param, double param2, const pgb_lltabs* tb) {
  double ll;
  if (family == PGB_FAMILY_CALLBACK) return 0.0; /* evaluated on the host, never here */
  if (family == PGB_FAMILY_POISSON_LOG || family == PGB_FAMILY_NEGBIN_LOG) {
    const double yy = y > 0.0 ? y : 0.0;
    const double em = pgb_exp_t(mu, tb->expt);
    if (family == PGB_FAMILY_POISSON_LOG) {
      const double sat = yy > 0.0 ? yy * pgb_log_t(yy, tb->logt) - yy : 0.0;
      ll = (yy * mu - em) - sat;
    } else {
      const double ay = param + yy;
      /* y > 0: y log y - (alpha + y) log(alpha + y);  y = 0: -alpha log alpha.  (Three logarithms per row, not
       * four: the first serves either case -- the same values, one evaluation site less in the kernel.) */
      const double l1 = pgb_log_t(yy > 0.0 ? yy : param, tb->logt);
      const double sat = yy > 0.0 ? yy * l1 - ay * pgb_log_t(ay, tb->logt) : -(param * l1);
      ll = (yy * mu - ay * pgb_log_t(param + em, tb->logt)) - sat;
    }
  } else if (family == PGB_FAMILY_GAMMA_LOG) {
    /* -alpha (y e^-mu + mu) minus its maximum over mu, -alpha (1 + log y) */
    const double yy = y > 1.0e-300 ? y : 1.0e-300;
    ll = -param * (((yy * pgb_exp_t(-mu, tb->expt) + mu) - 1.0) - pgb_log_t(yy, tb->logt));
  } else if (family == PGB_FAMILY_ASYMLAPLACE) {
    const double u = (y - mu) / param;
    ll = -(u * (u < 0.0 ? param2 - 1.0 : param2));
  } else if (family == PGB_FAMILY_STUDENT_T) {
    const double u = (y - mu) / param;
    ll = (-0.5 * (param2 + 1.0)) * pgb_log_t(1.0 + (u * u) / param2, tb->logt);
  } else {
    return pgb_loglik_bern_s(family, y > 0.5 ? mu : -mu, tb);
  }
  return PGB_CLAMP_LL(ll, 0.0);
}
PGB_HD double pgb_loglik1(int family, double y, double mu) {
  const pgb_lltabs tb = pgb_lltabs_default();
  return pgb_loglik1q(family, y, mu, 0.0, 1.0, &tb);
}

/* Categorical-softmax over K linear predictors: mu[y] - logsumexp(mu) (serial max / sum in output
 * order), clamped like pgb_loglik1q.  y is the class index stored as a double.  The sum is >= 1 (the largest
 * predictor contributes exp(0) = 1 exactly) unless a predictor is NaN, which gives the lower bound. */
PGB_HD double pgb_loglik_cat_t(int K, double y, const double* mu, const pgb_lltabs* tb) {
  double mx = mu[0];
  for (int k = 1; k < K; ++k)
    if (mu[k] > mx) mx = mu[k];
  double sum = 0.0;
  for (int k = 0; k < K; ++k) sum += pgb_exp_t(mu[k] - mx, tb->expt);
  int c = (int)y;
  if (c < 0) c = 0;
  if (c > K - 1) c = K - 1;
  double ll = (mu[c] - mx) - pgb_log_pos_t(sum, tb->logt);
  if (!(sum >= 1.0)) ll = -2047.0; /* NaN (or a sum that lost its unit term to one): the lower bound */
  return PGB_CLAMP_LL(ll, 0.0);
}

/* The same softmax log-likelihood for CONSTANT leaves, factorised (round 5).  During one tree update a row i is
 * evaluated once per (particle, round) that re-labels it -- ~ 48 times per tree at 40 particles -- and every time
 * with predictors mu_k = eta_k(i) + v_k: a part that belongs to the ROW (eta_k = sum_trees_noi_k + offset_k, fixed
 * while the tree is updated) and a part that belongs to the (particle, child) (the K leaf values).  With
 *     M = max_k eta_k,  a_k = eta_k - M,  E_k = exp(a_k)        per row, once per tree update
 *     d_k = v_k - v_0,  w_k = exp(d_k)                          per (particle, child), once per round
 * the log-likelihood is (a_c + d_c) - log S with S = sum_k E_k w_k: K fma and ONE logarithm per evaluation instead
 * of K exponentials and one logarithm (the exponentials were two thirds of the likelihood pass at K = 4).
 * The child's part is taken relative to output 0, not to the largest leaf value: each output's d_k, w_k then depends
 * on its own leaf value and output 0's alone (the kernels compute the outputs of a child side by side), and
 * w_0 = exp(0) = 1 exactly.  A child is FAST when every |d_k| <= PGB_CAT_DMAX = 300: then every w_k is a normal
 * double, S >= E_j w_j >= e^-300 for the output j with E_j = 1, S <= K e^300, and a term whose E_k underflowed
 * (a_k < -708) is below e^-408 S -- the factorised value is exact to the last few ulp of log S for ANY row, no
 * per-row test needed.  A child that is not fast (leaf values hundreds of units apart: a chain that has left every
 * sane regime) takes the unfactorised pgb_loglik_cat_t on mu_k = eta_k + v_k for all of its rows.
 * (Linear leaves keep pgb_loglik_cat_t: their leaf value changes from row to row.)
 * Maxima and sums are serial in output order, like everywhere in this contract. */
#ifndef PGB_CAT_DMAX
#define PGB_CAT_DMAX 300.0 /* (a test build of BOTH backends with a tiny value sends every child down the slow path) */
#endif
/* the (particle, child) part: d[k], w[k] from the K leaf values v[k]; returns 1 when the child is fast */
PGB_HD int pgb_cat_side(int K, const double* v, const double* expt, double* d, double* w) {
  int fast = 1;
  for (int k = 0; k < K; ++k) {
    d[k] = v[k] - v[0];
    w[k] = pgb_exp_t(d[k], expt);
    if (!(d[k] <= PGB_CAT_DMAX && d[k] >= -PGB_CAT_DMAX)) fast = 0;
  }
  return fast;
}
/* the row part: E[k] and a_c (c = the observed class, clamped like pgb_loglik_cat_t clamps it) from eta[k] */
PGB_HD int pgb_cat_class(int K, double y) {
  int c = (int)y;
  if (c < 0) c = 0;
  if (c > K - 1) c = K - 1;
  return c;
}
/* (E may be eta itself; the class is matched by compares, not by an index: a register array on the device) */
PGB_HD double pgb_cat_row(int K, int c, const double* eta, const double* expt, double* E) {
  double M = eta[0];
  for (int k = 1; k < K; ++k)
    if (eta[k] > M) M = eta[k];
  double a_c = 0.0;
  for (int k = 0; k < K; ++k) {
    const double ak = eta[k] - M;
    if (k == c) a_c = ak;
    E[k] = pgb_exp_t(ak, expt);
  }
  return a_c;
}
/* S = sum_k E[k] w[k], one fma per output in output order */
PGB_HD double pgb_cat_sum(int K, const double* E, const double* w) {
  double S = E[0] * w[0];
  for (int k = 1; k < K; ++k) S = PGB_FMA(E[k], w[k], S);
  return S;
}
/* the value of a row of a fast child */
PGB_HD double pgb_cat_value(double a_c, double d_c, double S, const double* logt) {
  const double ll = (a_c + d_c) - pgb_log_pos_t(S, logt);
  return PGB_CLAMP_LL(ll, 0.0);
}
/* everything together, as the oracle evaluates one row (the kernels keep E / a_c per row and d / w per child) */
PGB_HD double pgb_loglik_cat_f(int K, double y, const double* eta, const double* v, const double* d, const double* w,
                               int fast, const pgb_lltabs* tb) {
  if (fast) {
    double E[PGB_MAX_OUTPUTS];
    E[0] = 0.0; /* (K >= 1: always overwritten; quiets a compiler that cannot know) */
    const int c = pgb_cat_class(K, y);
    const double a_c = pgb_cat_row(K, c, eta, tb->expt, E);
    return pgb_cat_value(a_c, d[c], pgb_cat_sum(K, E, w), tb->logt);
  }
  double mu[PGB_MAX_OUTPUTS];
  mu[0] = 0.0; /* (see E above) */
  for (int k = 0; k < K; ++k) mu[k] = eta[k] + v[k];
  return pgb_loglik_cat_t(K, y, mu, tb);
}

/* Normal with BART mean and BART scale (reference tests/test_bart.py:118: Normal(w[0], |w[1]|)):
 * -log|s| - 0.5 ((y - m)/s)^2 (the constant -0.5 log 2pi cancels in the particle weights).
 * |s| is floored at 1e-8; clamped to [-2047, 2047]. */
PGB_HD double pgb_loglik_meanscale_t(double y, const double* mu, const pgb_lltabs* tb) {
  double sd = mu[1] < 0.0 ? -mu[1] : mu[1];
  if (!(sd >= 1e-8)) sd = 1e-8; /* (also a NaN scale) */
  if (sd > 1.0e300) sd = 1.0e300;
  const double z = (y - mu[0]) / sd;
  const double ll = -pgb_log_pos_t(sd, tb->logt) - 0.5 * (z * z);
  return PGB_CLAMP_LL(ll, 2047.0);
}

/* the contract's range for a callback's per-row value (NaN -> the lower bound) */
PGB_HD double pgb_clamp_loglik(double ll) {
  if (!(ll > -2047.0)) ll = -2047.0;
  if (ll > 2047.0) ll = 2047.0;
  return ll;
}

/* Per-row log-likelihood of every non-Normal(sigma) family at the K linear predictors mu. */
PGB_HD double pgb_loglikq_t(int family, int K, double y, const double* mu, double param, double param2,
                            const pgb_lltabs* tb) {
  if (family == PGB_FAMILY_CATEGORICAL) return pgb_loglik_cat_t(K, y, mu, tb);
  if (family == PGB_FAMILY_NORMAL_MEANSCALE) return pgb_loglik_meanscale_t(y, mu, tb);
  return pgb_loglik1q(family, y, mu[0], param, param2, tb);
}
PGB_HD double pgb_loglikq(int family, int K, double y, const double* mu, double param, double param2) {
  const pgb_lltabs tb = pgb_lltabs_default();
  return pgb_loglikq_t(family, K, y, mu, param, param2, &tb);
}
PGB_HD double pgb_loglik(int family, int K, double y, const double* mu) {
  return pgb_loglikq(family, K, y, mu, 0.0, 1.0);
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

/* [U] normalize + inverse_cdf: particles occupy entries [first, first+cnt) of a PGB_MAX_PARTICLES-entry
 * array of log-weights.  w_i = exp(lw_i - max) + 1e-12; W = the inclusive scan of w in BLOCKS of 64 entries, each
 * block in the order of pgb_scan64, every later block shifted by the (shifted) total of the block before it -- the
 * tree of additions a GPU evaluates with one DPP scan per block and one add; total = W[last].
 * pgb_pick returns the first i in [first, last) with !(u * total > W[i]), else last. */
PGB_HD void pgb_weights_scan(const double* lw, int first, int cnt, double* W) {
  double mx = lw[first];
  for (int i = first + 1; i < first + cnt; ++i)
    if (lw[i] > mx) mx = lw[i];
  for (int i = 0; i < PGB_MAX_PARTICLES; ++i) W[i] = 0.0;
  for (int i = first; i < first + cnt; ++i) W[i] = pgb_exp(lw[i] - mx) + 1e-12;
  for (int b = 0; b < PGB_MAX_PARTICLES; b += 64) {
    pgb_scan64(W + b);
    if (b > 0)
      for (int i = b; i < b + 64; ++i) W[i] = W[i] + W[b - 1];
  }
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
