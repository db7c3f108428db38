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

/* Both compilers MUST be run with -ffp-contract=off (see __graft_entry__.build). */
#if defined(__clang__)
#pragma clang fp contract(off)
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
  p = p * r + 2.08767569878681e-09;       /* 1/12! */
  p = p * r + 2.505210838544172e-08;      /* 1/11! */
  p = p * r + 2.755731922398589e-07;      /* 1/10! */
  p = p * r + 2.7557319223985893e-06;     /* 1/9!  */
  p = p * r + 2.48015873015873e-05;       /* 1/8!  */
  p = p * r + 1.984126984126984e-04;      /* 1/7!  */
  p = p * r + 1.388888888888889e-03;      /* 1/6!  */
  p = p * r + 8.333333333333333e-03;      /* 1/5!  */
  p = p * r + 4.1666666666666664e-02;     /* 1/4!  */
  p = p * r + 1.6666666666666666e-01;     /* 1/3!  */
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
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
  q = q * z + 4.7619047619047616e-02; /* 1/21 */
  q = q * z + 5.2631578947368418e-02; /* 1/19 */
  q = q * z + 5.8823529411764705e-02; /* 1/17 */
  q = q * z + 6.6666666666666666e-02; /* 1/15 */
  q = q * z + 7.6923076923076927e-02; /* 1/13 */
  q = q * z + 9.0909090909090912e-02; /* 1/11 */
  q = q * z + 1.1111111111111111e-01; /* 1/9  */
  q = q * z + 1.4285714285714285e-01; /* 1/7  */
  q = q * z + 0.2;                    /* 1/5  */
  q = q * z + 3.3333333333333331e-01; /* 1/3  */
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
  s = s * w2 + 2.8114572543455206e-15; /*  1/17! */
  s = s * w2 - 7.6471637318198164e-13; /* -1/15! */
  s = s * w2 + 1.6059043836821613e-10; /*  1/13! */
  s = s * w2 - 2.5052108385441720e-08; /* -1/11! */
  s = s * w2 + 2.7557319223985893e-06; /*  1/9!  */
  s = s * w2 - 1.9841269841269841e-04; /* -1/7!  */
  s = s * w2 + 8.3333333333333332e-03; /*  1/5!  */
  s = s * w2 - 1.6666666666666666e-01; /* -1/3!  */
  s = w + (w * w2) * s;
  double c = 4.1103176233121648e-19;   /*  1/20! */
  c = c * w2 - 1.5619206968586225e-16; /* -1/18! */
  c = c * w2 + 4.7794773323873853e-14; /*  1/16! */
  c = c * w2 - 1.1470745597729725e-11; /* -1/14! */
  c = c * w2 + 2.0876756987868100e-09; /*  1/12! */
  c = c * w2 - 2.7557319223985888e-07; /* -1/10! */
  c = c * w2 + 2.4801587301587302e-05; /*  1/8!  */
  c = c * w2 - 1.3888888888888889e-03; /* -1/6!  */
  c = c * w2 + 4.1666666666666664e-02; /*  1/4!  */
  c = c * w2 - 0.5;
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
/* log Phi(x) (standard normal CDF), deterministic: |x|/sqrt2 < 2.5 -> positive-term series
 * erf(z) = 2/sqrt(pi) e^{-z^2} sum_n 2^n z^(2n+1)/(2n+1)!! (40 terms, reciprocals tabulated);
 * otherwise the continued fraction of erfc evaluated by the division-free forward recurrence
 * (48 steps).  Absolute error < 3e-12 against scipy.special.log_ndtr on [-38, 10]. */
PGB_HD double pgb_log_ndtr(double x) {
  const double rodd[40] = {0.3333333333333333, 0.2, 0.14285714285714285, 0.1111111111111111, 0.09090909090909091, 0.07692307692307693, 0.06666666666666667, 0.058823529411764705, 0.05263157894736842, 0.047619047619047616, 0.043478260869565216, 0.04, 0.037037037037037035, 0.034482758620689655, 0.03225806451612903, 0.030303030303030304, 0.02857142857142857, 0.02702702702702703, 0.02564102564102564, 0.024390243902439025, 0.023255813953488372, 0.022222222222222223, 0.02127659574468085, 0.02040816326530612, 0.0196078431372549, 0.018867924528301886, 0.01818181818181818, 0.017543859649122806, 0.01694915254237288, 0.01639344262295082, 0.015873015873015872, 0.015384615384615385, 0.014925373134328358, 0.014492753623188406, 0.014084507042253521, 0.0136986301369863, 0.013333333333333334, 0.012987012987012988, 0.012658227848101266, 0.012345679012345678};
  const double z = (x < 0.0 ? -x : x) * 0.70710678118654752440;
  if (z < 2.5) {
    double t = z, s = z;
    const double z2 = 2.0 * z * z;
    for (int n = 0; n < 40; ++n) {
      t = (t * z2) * rodd[n];
      s = s + t;
    }
    const double erf = (1.1283791670955125739 * pgb_exp(-(z * z))) * s;
    return x >= 0.0 ? pgb_log(0.5 + 0.5 * erf) : pgb_log(0.5 * (1.0 - erf));
  }
  const double a = 1.0 / (2.0 * z * z);
  double Am = 1.0, A = 1.0, Bm = 0.0, B = 1.0;
  for (int k = 1; k <= 48; ++k) {
    const double ak = (double)k * a;
    const double An = A + ak * Am, Bn = B + ak * Bm;
    Am = A; A = An;
    Bm = B; B = Bn;
  }
  const double logerfc = (-(z * z) - pgb_log(z * 1.7724538509055160273)) + pgb_log(B / A);
  return x >= 0.0 ? pgb_log(1.0 - 0.5 * pgb_exp(logerfc)) : (-0.69314718055994530942 + logerfc);
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
