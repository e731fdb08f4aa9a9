// tq_math.h -- scalar building blocks shared by every kernel of the cosmos hot path.
//
// Everything here is a small inline function on scalars so that (a) the HIP kernels
// (tq_*.hip, compiled for gfx950) and (b) the host-side checker used by the CPU test
// suite (tests/hostcheck, compiled with g++) run the SAME arithmetic.  The checker is
// test infrastructure only; the product never falls back to it.
//
// Contents
//   * Philox4x32-10 counter RNG, uniform / normal draws
//   * Marsaglia-Tsang standard-Gamma sampler (the algorithm torch.distributions.Gamma
//     uses: aten/src/ATen/native/Distributions.h `sample_gamma`)
//   * lgamma / digamma through the Binet function S(a) = lgamma(a) - [(a-1/2)ln a - a + ln sqrt(2 pi)]
//   * implicit reparameterisation gradients of Gamma and Beta/Dirichlet draws
//     (Figurnov et al. 2018; Jankowiak & Obermeyer 2018).  The piecewise scheme and the
//     fitted rational coefficients are those published in PyTorch (BSD-3,
//     ATen/native/Distributions.h: standard_gamma_grad_one, dirichlet_grad_one) because
//     the reference's gradients are *defined* by them (pyro rsample -> torch.autograd).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TQ_HD __host__ __device__ __forceinline__
// The fp64 special functions below are inlined: every kernel is organised so that a lane
// evaluates them at most twice (one guide site per lane), which keeps the code small and
// scratch-free (device function calls would need a scratch-backed stack).
#define TQ_HD_NOINLINE __host__ __device__ __forceinline__
#else
#define TQ_HD inline
#define TQ_HD_NOINLINE inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// Raw v_log_f32 / v_exp_f32 / v_rcp_f32 (1 ulp, one quarter-rate instruction each on CDNA4).
// hipcc's __logf / __frcp_rn expand to 10-12 instruction IEEE sequences (denormal scaling,
// div_scale/div_fmas/div_fixup); every argument on the pixel path is a normal positive number.
#define TQ_FLOG(x) (__builtin_amdgcn_logf(x) * 0.69314718055994530942f)
#define TQ_FEXP(x) __builtin_amdgcn_exp2f((x) * 1.44269504088896340736f)
#define TQ_FRCP(x) __builtin_amdgcn_rcpf(x)
#define TQ_FLOG2(x) __builtin_amdgcn_logf(x)
#define TQ_FEXP2(x) __builtin_amdgcn_exp2f(x)
#define TQ_FSQRT(x) __builtin_amdgcn_sqrtf(x)
#else
#define TQ_FLOG(x) logf(x)
#define TQ_FEXP(x) expf(x)
#define TQ_FRCP(x) (1.0f / (x))
#define TQ_FLOG2(x) log2f(x)
#define TQ_FEXP2(x) exp2f(x)
#define TQ_FSQRT(x) sqrtf(x)
#endif
#define TQ_LOG2E 1.44269504088896340736f


// ------------------------------------------------------------------------------------------
// fp64 reciprocal / square root / logarithm from fp32 hardware seeds + Newton steps (relative error
// < 1e-13, arguments well inside the float range).  The library versions cost 25-100 instructions
// each and dominated the implicit-gradient code, whose results are rounded to float anyway.
// ------------------------------------------------------------------------------------------
TQ_HD double tq_drcp(double b) {
  double y = (double)TQ_FRCP((float)b);
  double e = fma(-b, y, 1.0);
  y = fma(y, e, y);
  e = fma(-b, y, 1.0);
  return fma(y, e, y);
}
TQ_HD double tq_dsqrt(double a) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double r = (double)__builtin_amdgcn_rsqf((float)a);
#else
  const double r = (double)(1.0f / sqrtf((float)a));
#endif
  double g = a * r, h = 0.5 * r;
  double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  e = fma(-h, g, 0.5);
  return fma(g, e, g);
}
TQ_HD double tq_dlog(double z) {
  // z = m 2^e with m in [sqrt(1/2), sqrt(2)); ln m = 2 atanh((m-1)/(m+1))
  union { double d; long long i; } u;
  u.d = z;
  int e = (int)((u.i >> 52) & 0x7ff) - 1023;
  u.i = (u.i & 0x000fffffffffffffLL) | 0x3ff0000000000000LL;
  double m = u.d;
  if (m > 1.41421356237309504880) {
    m *= 0.5;
    e += 1;
  }
  const double s = (m - 1.0) * tq_drcp(m + 1.0);
  const double s2 = s * s;
  double p = 1.0 / 17.0;
  p = fma(p, s2, 1.0 / 15.0);
  p = fma(p, s2, 1.0 / 13.0);
  p = fma(p, s2, 1.0 / 11.0);
  p = fma(p, s2, 1.0 / 9.0);
  p = fma(p, s2, 1.0 / 7.0);
  p = fma(p, s2, 1.0 / 5.0);
  p = fma(p, s2, 1.0 / 3.0);
  p = fma(p, s2, 1.0);
  return fma((double)e, 0.69314718055994530942, 2.0 * s * p);
}

#define TQ_LN_SQRT_2PI 0.91893853320467274178f
#define TQ_LN2 0.69314718055994530942f
#define TQ_PI 3.14159265358979323846f

// ------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  key = (seed_lo, seed_hi); counter = 128 bit.
// ------------------------------------------------------------------------------------------
struct TqPhilox {
  uint32_t c[4];
  uint32_t k[2];
  uint32_t out[4];
  int have;  // unread words in out
};

TQ_HD void tq_philox_round(uint32_t* c, const uint32_t* k) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

TQ_HD void tq_philox_block(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
  uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
  uint32_t k[2] = {key[0], key[1]};
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    tq_philox_round(c, k);
    k[0] += 0x9E3779B9u;
    k[1] += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

// stream = (seed, step, site id, element id): every latent scalar owns a counter sequence,
// so results do not depend on the launch geometry or on how AOIs are sharded over GPUs.
TQ_HD void tq_philox_init(TqPhilox* s, uint64_t seed, uint32_t step, uint32_t site, uint64_t elem) {
  s->k[0] = (uint32_t)seed;
  s->k[1] = (uint32_t)(seed >> 32);
  s->c[0] = 0;  // block counter within the stream
  s->c[1] = (uint32_t)elem;
  s->c[2] = (uint32_t)(elem >> 32) ^ (site << 20);
  s->c[3] = step;
  s->have = 0;
}

TQ_HD uint32_t tq_philox_next(TqPhilox* s) {
  if (s->have == 0) {
    tq_philox_block(s->c, s->k, s->out);
    s->c[0] += 1;
    s->have = 4;
  }
  s->have -= 1;
  return s->out[3 - s->have];
}

// uniform on the open interval (0, 1): 24 random bits, centred
TQ_HD float tq_uniform(TqPhilox* s) {
  return (float)(tq_philox_next(s) >> 8) * 5.9604644775390625e-08f + 2.98023223876953125e-08f;
}

// Box-Muller: both normals of one (u1, u2) pair
TQ_HD void tq_normal_pair(float u1, float u2, float* x1, float* x2) {
#if defined(__HIP_DEVICE_COMPILE__)
  // v_cos_f32 / v_sin_f32 take their argument in revolutions
  const float r = __builtin_amdgcn_sqrtf(-2.0f * TQ_FLOG(u1));
  *x1 = r * __builtin_amdgcn_cosf(u2);
  *x2 = r * __builtin_amdgcn_sinf(u2);
#else
  const float r = sqrtf(-2.0f * logf(u1));
  *x1 = r * cosf(2.0f * TQ_PI * u2);
  *x2 = r * sinf(2.0f * TQ_PI * u2);
#endif
}

// Standard Gamma(alpha, 1) draw; Marsaglia & Tsang (2000) with the alpha < 1 boost (the algorithm of
// torch's sample_gamma).  One Philox block (4 words) feeds TWO independent candidates -- the two
// Box-Muller normals and two uniforms -- evaluated in straight-line code; the first accepted one is
// returned.  A lane needs a second block with probability ~1e-3, so a wave64 almost never loops.
TQ_HD float tq_sample_std_gamma(TqPhilox* s, float alpha) {
  float scale = 1.0f;
  if (alpha < 1.0f) {
    if (alpha == 0.0f) return 0.0f;
    scale = TQ_FEXP2(TQ_FLOG2(1.0f - tq_uniform(s)) * TQ_FRCP(alpha));  // u^(1/alpha) boost of Marsaglia-Tsang
    alpha += 1.0f;
    s->have = 0;  // the candidates below start on a fresh block
  }
  const float d = alpha - 1.0f / 3.0f;
#if defined(__HIP_DEVICE_COMPILE__)
  const float c = __builtin_amdgcn_rsqf(9.0f * d);
#else
  const float c = 1.0f / sqrtf(9.0f * d);
#endif
  for (int it = 0; it < 32; ++it) {  // the bound only guards against a stuck wave
    const float ua = tq_uniform(s), ub = tq_uniform(s);
    const float u[2] = {1.0f - tq_uniform(s), 1.0f - tq_uniform(s)};
    float x[2];
    tq_normal_pair(ua, ub, &x[0], &x[1]);
    float out = -1.0f;
#pragma unroll
    for (int j = 1; j >= 0; --j) {  // candidate 0 has priority: evaluated last so that it overwrites
      const float y = 1.0f + c * x[j];
      const float v = y * y * y;
      const float xx = x[j] * x[j];
      const bool ok = (y > 0.0f) && ((u[j] < 1.0f - 0.0331f * xx * xx) ||
                                     (TQ_FLOG(u[j]) < 0.5f * xx + d * (1.0f - v + TQ_FLOG(fmaxf(v, 1e-37f)))));
      if (ok) out = d * v;
    }
    if (out >= 0.0f) return scale * out;
  }
  return scale * d;  // practically unreachable
}

// ------------------------------------------------------------------------------------------
// lgamma / digamma via the Binet function.
//   lgamma(a)  = (a - 1/2) ln a - a + ln sqrt(2 pi) + S(a)
//   digamma(a) = ln a - 1/(2a) + S'(a)
//   S(a)  =  1/(12a) - 1/(360a^3) + 1/(1260a^5) - 1/(1680a^7)
//   S'(a) = -1/(12a^2) + 1/(120a^4) - 1/(252a^6) + 1/(240a^8)
// Series for a >= 8 (truncation < 3e-10); below that, shift up by 8 with the recurrences
// lgamma(a) = lgamma(a+8) - ln prod_{i<8}(a+i), digamma(a) = digamma(a+8) - sum_{i<8} 1/(a+i).
// ------------------------------------------------------------------------------------------
template <typename T>
TQ_HD void tq_binet_series(T a, T ra, T* S, T* dS) {
  const T r2 = ra * ra;
  *S = ra * (T(1.0 / 12.0) + r2 * (T(-1.0 / 360.0) + r2 * (T(1.0 / 1260.0) + r2 * T(-1.0 / 1680.0))));
  *dS = r2 * (T(-1.0 / 12.0) + r2 * (T(1.0 / 120.0) + r2 * (T(-1.0 / 252.0) + r2 * T(1.0 / 240.0))));
}

// a >= 8: three terms are exact to fp32 (next terms 1/(1680 a^7) < 3e-10, 1/(240 a^8) < 3e-10)
TQ_HD void tq_binet_fast(float ra, float* S, float* dS) {
  const float r2 = ra * ra;
  *S = ra * (1.0f / 12.0f + r2 * (-1.0f / 360.0f + r2 * (1.0f / 1260.0f)));
  *dS = r2 * (-1.0f / 12.0f + r2 * (1.0f / 120.0f + r2 * (-1.0f / 252.0f)));
}

// Binet function and its derivative for any a > 0, given ln a and 1/a.
TQ_HD void tq_binet(float a, float lna, float ra, float* S, float* dS) {
  if (a >= 8.0f) {
    tq_binet_series<float>(a, ra, S, dS);
    return;
  }
  // shifted path: rare on the pixel path (needs background/gain < 8), the rule for the height sites of the guide
  // (concentration ~2), hence the hardware reciprocals / logarithms
  float prod = 1.0f, rsum = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float ai = a + (float)i;
    prod *= ai;
    rsum += TQ_FRCP(ai);
  }
  const float b = a + 8.0f;
  const float lnb = TQ_FLOG(b), rb = TQ_FRCP(b);
  float Sb, dSb;
  tq_binet_series<float>(b, rb, &Sb, &dSb);
  // lgamma(a) = (b-1/2)ln b - b + c + Sb - ln prod  and  S(a) = lgamma(a) - (a-1/2)ln a + a - c
  *S = (b - 0.5f) * lnb - 8.0f + Sb - TQ_FLOG(prod) - (a - 0.5f) * lna;
  // digamma(a) = ln b - 1/(2b) + dSb - rsum ; S'(a) = digamma(a) - ln a + 1/(2a)
  *dS = (lnb - lna) - 0.5f * rb + dSb - rsum + 0.5f * ra;
}

TQ_HD float tq_lgamma(float a) {
  const float lna = logf(a), ra = 1.0f / a;
  float S, dS;
  tq_binet(a, lna, ra, &S, &dS);
  return (a - 0.5f) * lna - a + TQ_LN_SQRT_2PI + S;
}

TQ_HD float tq_digamma(float a) {
  const float lna = logf(a), ra = 1.0f / a;
  float S, dS;
  tq_binet(a, lna, ra, &S, &dS);
  return lna - 0.5f * ra + dS;
}

TQ_HD void tq_lgamma_digamma(float a, float* lg, float* dg) {
  const float lna = TQ_FLOG(a), ra = TQ_FRCP(a);
  float S, dS;
  tq_binet(a, lna, ra, &S, &dS);
  *lg = (a - 0.5f) * lna - a + TQ_LN_SQRT_2PI + S;
  *dg = lna - 0.5f * ra + dS;
}

// double-precision versions (host-side globals math and the mid-range Beta gradient)
TQ_HD_NOINLINE void tq_lgamma_digamma_d(double a, double* lg, double* dg) {
  // shift up to a >= 12 with  lgamma(a) = lgamma(a+n) - ln prod (a+i),  digamma(a) = digamma(a+n) - sum 1/(a+i):
  // one logarithm of the running product instead of one per step
  double prod = 1.0, shift_d = 0.0;
  for (int it = 0; it < 12 && a < 12.0; ++it) {
    prod *= a;
    shift_d += tq_drcp(a);
    a += 1.0;
  }
  const double ra = tq_drcp(a), r2 = ra * ra, lna = tq_dlog(a);
  const double S = ra * (1.0 / 12.0 + r2 * (-1.0 / 360.0 + r2 * (1.0 / 1260.0 + r2 * (-1.0 / 1680.0 + r2 * (1.0 / 1188.0)))));
  const double dS = r2 * (-1.0 / 12.0 + r2 * (1.0 / 120.0 + r2 * (-1.0 / 252.0 + r2 * (1.0 / 240.0 + r2 * (-1.0 / 132.0)))));
  *lg = (a - 0.5) * lna - a + 0.91893853320467274178 + S - tq_dlog(prod);
  *dg = lna - 0.5 * ra + dS - shift_d;
}

// digamma alone: shift to a >= 8 (sum of reciprocals), then the asymptotic series (next term 1/(240 a^8) <= 2.5e-10)
TQ_HD double tq_digamma_fast_d(double a) {
  double shift = 0.0;
#pragma nounroll
  for (int it = 0; it < 8 && a < 8.0; ++it) {
    shift += tq_drcp(a);
    a += 1.0;
  }
  const double ra = tq_drcp(a), r2 = ra * ra;
  return tq_dlog(a) - 0.5 * ra - r2 * (1.0 / 12.0 - r2 * (1.0 / 120.0 - r2 * (1.0 / 252.0))) - shift;
}

// (1 - x)^e for 0 <= x < 1: the logarithm in fp64 (e can be thousands while x is tiny: e ln(1-x) must not inherit the
// rounding of 1-x to fp32), the exponential of the O(1) product in fp32 (relative error ~1e-7)
TQ_HD double tq_pow1m(double x, double e) { return (double)TQ_FEXP((float)(e * tq_dlog(1.0 - x))); }

// ------------------------------------------------------------------------------------------
// numerically safe logistic helpers (unconstrained -> constrained transforms)
// ------------------------------------------------------------------------------------------
// ln(1 + u), u > -1, from ONE hardware logarithm: the rounding of 1 + u is compensated where it matters (|u| < 1/2:
// ln(1 + u) = ln w + (u - (w - 1)) / w + O(eps^2)); libm's log1pf is ~35 instructions and the per-unit routine calls it
// eight times per unit at K = 2
TQ_HD float tq_log1p_fast(float u) {
  const float w = 1.0f + u;
  const float c = TQ_FLOG(w);
  return (fabsf(u) < 0.5f) ? c - ((w - 1.0f) - u) * TQ_FRCP(w) : c;
}
TQ_HD float tq_softplus(float u) {  // ln(1 + e^u) = max(u, 0) + ln(1 + e^-|u|)
  return fmaxf(u, 0.0f) + tq_log1p_fast(TQ_FEXP(-fabsf(u)));
}
TQ_HD float tq_sigmoid(float u) {  // branch-free: e = e^-|u| in (0, 1]
  const float e = TQ_FEXP(-fabsf(u));
  const float r = TQ_FRCP(1.0f + e);
  return u >= 0.0f ? r : e * r;
}

// ------------------------------------------------------------------------------------------
// Implicit reparameterisation gradient  d g / d alpha  of g ~ Gamma(alpha, 1):
//   -(d/dalpha CDF(g; alpha)) / pdf(g; alpha)
// ------------------------------------------------------------------------------------------
TQ_HD_NOINLINE float tq_std_gamma_grad(float alpha_, float x_) {
  // evaluated in double: the saddle-point branch cancels badly in float and this runs
  // once per latent scalar, not per pixel
  const double x = x_, alpha = alpha_;
  if (x < 0.8) {
    // Taylor series of the lower incomplete gamma function in x.  With cdf = x^a s1, d cdf / d a = (ln x - psi(a)) cdf
    // - x^a s2 and pdf = x^(a-1) e^-x (all three without the common 1/Gamma(a)) the powers of x cancel:
    //   -(d cdf / d a) / pdf = x e^x [ s2 - (ln x - psi(a)) s1 ].
    // The reciprocals 1/(a+i) of the series also shift psi(a) to psi(a+6).
    double numer = 1.0, r = tq_drcp(alpha);
    double series1 = r, series2 = r * r, shift = r;
#pragma unroll
    for (int i = 1; i <= 5; ++i) {
      numer *= -x * (1.0 / (double)i);
      r = tq_drcp(alpha + (double)i);
      series1 += numer * r;
      series2 += numer * r * r;
      shift += r;
    }
    const double z = alpha + 6.0, rz = tq_drcp(z), rz2 = rz * rz;
    const double psi = tq_dlog(z) - 0.5 * rz - rz2 * (1.0 / 12.0 - rz2 * (1.0 / 120.0 - rz2 * (1.0 / 252.0 - rz2 * (1.0 / 240.0)))) - shift;
    const double result = x * (double)TQ_FEXP((float)x) * (series2 - (tq_dlog(x) - psi) * series1);
    return (result != result) ? 0.0f : (float)result;
  }
  if (alpha > 8.0) {  // Rice saddle-point expansion
    if (0.9 * alpha <= x && x <= 1.1 * alpha) {
      const double numer_1 = 1.0 + 24.0 * alpha * (1.0 + 12.0 * alpha);
      const double numer_2 = 1440.0 * (alpha * alpha) + 6.0 * x * (53.0 - 120.0 * x) - 65.0 * x * x * tq_drcp(alpha) +
                             alpha * (107.0 + 3600.0 * x);
      const double denom = 1244160.0 * (alpha * alpha) * (alpha * alpha);
      return (float)(numer_1 * numer_2 * tq_drcp(denom));
    }
    const double ra = tq_drcp(alpha);
    const double denom = tq_dsqrt(8.0 * alpha);
    const double rax = tq_drcp(alpha - x);
    const double term2 = denom * rax;
    const double lxa = tq_dlog(x * ra);
    const double t3b = x - alpha - alpha * lxa;
    const double term3 = tq_drcp(t3b * tq_dsqrt(t3b));  // t3b^(-3/2)
    const double term23 = (x < alpha) ? term2 - term3 : term2 + term3;
    const double term1 = lxa * term23 - tq_dsqrt(2.0 * ra) * (alpha + x) * rax * rax;
    const double stirling = 1.0 + ra * (1.0 / 12.0) * (1.0 + ra * (1.0 / 24.0));
    const double numer = x * term1;
    return (float)(-stirling * numer * tq_drcp(denom));
  }
  // bivariate rational approximation in (ln(x/alpha), ln alpha); coefficients: PyTorch (BSD-3)
  const float v = TQ_FLOG(alpha_);
  const float u = TQ_FLOG(x_) - v;
  const float coef_uv[3][8] = {
      {0.16009398f, -0.094634809f, 0.025146376f, -0.0030648343f, 1.0f, 0.32668115f, 0.10406089f, 0.0014179084f},
      {0.53487893f, 0.1298071f, 0.065735949f, -0.0015649758f, 0.16639465f, 0.020070113f, -0.0035938915f, -0.00058392623f},
      {0.040121004f, -0.0065914022f, -0.0026286047f, -0.0013441777f, 0.017050642f, -0.0021309326f, 0.00085092367f,
       -1.5247877e-07f},
  };
  float coef_v[8];
  for (int i = 0; i < 8; ++i) coef_v[i] = coef_uv[0][i] + u * (coef_uv[1][i] + u * coef_uv[2][i]);
  const float p = coef_v[0] + v * (coef_v[1] + v * (coef_v[2] + v * coef_v[3]));
  const float q = coef_v[4] + v * (coef_v[5] + v * (coef_v[6] + v * coef_v[7]));
  return TQ_FEXP(p * TQ_FRCP(q));  // fp32 throughout, as torch evaluates it for float32 tensors (O(1) quantities)
}

// ------------------------------------------------------------------------------------------
// Scaled implicit gradient of x ~ Beta(alpha, total - alpha):
//   -(d/dalpha CDF(x; alpha, beta)) / pdf(x; alpha, beta) / (1 - x)
// (the quantity torch._dirichlet_grad returns; Dirichlet/Beta rsample backward is built on it)
// ------------------------------------------------------------------------------------------
TQ_HD double tq_digamma_d(double a) {
  double lg, dg;
  tq_lgamma_digamma_d(a, &lg, &dg);
  return dg;
}

// fp32 digamma: shift to a >= 6 by the recurrence, then the asymptotic series (next term 1/(240 a^8) < 3e-9)
TQ_HD float tq_digamma_f(float a) {
  float shift = 0.0f;
#pragma nounroll
  for (int it = 0; it < 6 && a < 6.0f; ++it) {
    shift += TQ_FRCP(a);
    a += 1.0f;
  }
  const float ra = TQ_FRCP(a), r2 = ra * ra;
  return TQ_FLOG(a) - 0.5f * ra - r2 * (1.0f / 12.0f - r2 * (1.0f / 120.0f - r2 * (1.0f / 252.0f))) - shift;
}

// psi(total) - psi(alpha) for total > alpha > 0.  In float when the difference is not a small one of two large values
// (total >= 1.1 alpha: absolute error ~3e-7 of each digamma against a difference >= 0.09 -- relative error ~1e-5, far
// inside what torch's own float32 evaluation has); in double otherwise.  Fits that have converged sit almost entirely in the
// first case (c1 ~ c0), and the double branch with its reciprocal loops is then skipped by whole waves.
TQ_HD float tq_digamma_diff(float total, float alpha) {
  if (total >= 1.1f * alpha) return tq_digamma_f(total) - tq_digamma_f(alpha);
  return (float)(tq_digamma_fast_d((double)total) - tq_digamma_fast_d((double)alpha));
}

// The 1-x-small series regime of torch's _dirichlet_grad in fp32 (terms decay like (1-x)^i / i!; checked against
// the fp64 evaluation to 5e-5).  The x-small regime stays in fp64: its alternating series cancels by factors of
// several hundred when beta x is near the regime boundary and needs psi, ln x and 1/(alpha+i) to ~1e-9.
TQ_HD float tq_beta_grad_beta_small_f(float x, float alpha, float beta) {
  // psi(alpha + beta) - psi(beta) cancels when alpha << beta: then the two digammas in fp64 (tq_digamma_diff)
  const float factor = tq_digamma_diff(alpha + beta, beta);
  float numer = 1.0f, betas = 1.0f, dbetas = 0.0f, series = factor * TQ_FRCP(alpha);
#pragma nounroll
  for (int i = 1; i <= 8; ++i) {
    const float ci = (float)i;
    numer *= -x * TQ_FRCP(ci);
    dbetas = dbetas * (beta - ci) + betas;
    betas = betas * (beta - ci);
    series += numer * TQ_FRCP(alpha + ci) * (dbetas + factor * betas);
  }
  const float result = -TQ_FEXP((1.0f - beta) * log1pf(-x)) * series;  // -(1-x)^(1-beta) series
  return (result != result) ? 0.0f : result;
}

TQ_HD double tq_beta_grad_alpha_small(double x, double alpha, double beta) {
  // 1/(alpha+i), i = 0..10, by batch inversion (one fp64 reciprocal of the running product, three multiplications per
  // element); their sum also shifts psi(alpha) to psi(alpha + 11), where the asymptotic series is exact to 1e-12
  double pref[11], r[11];
  double prod = 1.0;
#pragma unroll
  for (int i = 0; i <= 10; ++i) {
    pref[i] = prod;
    prod *= alpha + (double)i;
  }
  double inv = tq_drcp(prod), shift = 0.0;
#pragma unroll
  for (int i = 10; i >= 0; --i) {
    r[i] = inv * pref[i];
    inv *= alpha + (double)i;
    shift += r[i];
  }
  const double z = alpha + 11.0, rz = tq_drcp(z), rz2 = rz * rz;
  const double psi_a = tq_dlog(z) - 0.5 * rz - rz2 * (1.0 / 12.0 - rz2 * (1.0 / 120.0 - rz2 * (1.0 / 252.0))) - shift;
  const double factor = psi_a - tq_digamma_fast_d(alpha + beta) - tq_dlog(x);
  double numer = 1.0;
  double series = r[0] * (factor + r[0]);
#pragma unroll
  for (int i = 1; i <= 10; ++i) {
    numer *= ((double)i - beta) * x * (1.0 / (double)i);
    series += numer * r[i] * (factor + r[i]);
  }
  const double result = x * tq_pow1m(x, -beta) * series;
  return (result != result) ? 0.0 : result;
}

TQ_HD double tq_beta_grad_beta_small(double x, double alpha, double beta) {
  const double factor = tq_digamma_fast_d(alpha + beta) - tq_digamma_fast_d(beta);
  double numer = 1.0, betas = 1.0, dbetas = 0.0, series = factor * tq_drcp(alpha);
#pragma nounroll
  for (int i = 1; i <= 8; ++i) {
    const double ci = (double)i;
    numer *= -x * tq_drcp(ci);
    dbetas = dbetas * (beta - ci) + betas;
    betas = betas * (beta - ci);
    series += numer * tq_drcp(alpha + ci) * (dbetas + factor * betas);
  }
  const double result = -tq_pow1m(x, 1.0 - beta) * series;
  return (result != result) ? 0.0 : result;
}

// local polynomial of the saddle-point gradient around its removable singularity at x = mean
TQ_HD double tq_beta_grad_window(double x, double alpha, double beta) {
  const double total = alpha + beta;
  const double b2 = beta * beta;
  const double poly =
      47.0 * x * b2 * b2 +
      alpha * ((43.0 + 20.0 * (16.0 + 27.0 * beta) * x) * b2 * beta +
               alpha * (3.0 * (59.0 + 180.0 * beta - 90.0 * x) * b2 +
                        alpha * ((453.0 + 1620.0 * beta * (1.0 - x) - 455.0 * x) * beta +
                                 alpha * (8.0 * (1.0 - x) * (135.0 * beta - 11.0)))));
  const double prefactor_num = (1.0 + 12.0 * alpha) * (1.0 + 12.0 * beta);
  const double prefactor_den = 12960.0 * alpha * alpha * alpha * b2 * (1.0 + 12.0 * total) * (total * total) * (1.0 - x);
  return prefactor_num * poly * tq_drcp(prefactor_den);
}

TQ_HD double tq_beta_grad_alpha_mid(double x, double alpha, double beta) {
  const double total = alpha + beta;
  const double rt = tq_drcp(total);
  const double mean = alpha * rt;
  const double sd = tq_dsqrt(alpha * beta * tq_drcp(total + 1.0)) * rt;
  if (mean - 0.1 * sd <= x && x <= mean + 0.1 * sd) return tq_beta_grad_window(x, alpha, beta);
  const double prefactor = -x * tq_drcp(tq_dsqrt(2.0 * alpha * beta * rt));
  const double fa = tq_drcp(12.0 * alpha), fb = tq_drcp(12.0 * beta), ft = tq_drcp(12.0 * total);
  const double stirling = (1.0 + fa + 0.5 * fa * fa) * (1.0 + fb + 0.5 * fb * fb) * tq_drcp(1.0 + ft + 0.5 * ft * ft);
  const double term1_num = 2.0 * (alpha * alpha) * (x - 1.0) + alpha * beta * (x - 1.0) - x * (beta * beta);
  const double axbx = alpha * (x - 1.0) + beta * x;
  const double term1_den = tq_dsqrt(2.0 * alpha * tq_drcp(beta)) * (total * tq_dsqrt(total)) * axbx * axbx;
  const double term1 = term1_num * tq_drcp(term1_den);
  const double la = tq_dlog(alpha * rt * tq_drcp(x));
  const double term2 = 0.5 * la;
  const double term3 = tq_dsqrt(8.0 * alpha * beta * rt) * tq_drcp(axbx);
  const double term4_base = beta * tq_dlog(beta * rt * tq_drcp(1.0 - x)) + alpha * la;
  const double term4 = tq_drcp(term4_base * tq_dsqrt(term4_base));  // term4_base^(-3/2)
  const double term1234 = term1 + term2 * (term3 + (x < mean ? term4 : -term4));
  return stirling * prefactor * term1234;
}

// Both implicit gradients of one Beta(alpha, beta) draw x,
//   ga = tq_dirichlet_grad(x, alpha, alpha+beta),  gb = tq_dirichlet_grad(1-x, beta, alpha+beta),
// when both fall in the saddle-point regime: the two evaluations share total, the two logarithms
// (each is the other's term2 and both enter term4_base), sqrt(2 alpha beta / total), term4 and the
// Stirling ratio; |alpha (x-1) + beta x| is the same up to sign.  Returns false (nothing written) if
// either direction needs a different branch of the piecewise scheme.
// The same polynomial in float: no cancellation (relative error < 1e-6 against the double evaluation over the whole
// window, alpha, beta in [6, 3000]); the products stay inside the float range for alpha, beta < 2000.
TQ_HD float tq_beta_grad_window_f(float x, float alpha, float beta) {
  const float total = alpha + beta;
  const float b2 = beta * beta;
  const float poly =
      47.0f * x * b2 * b2 +
      alpha * ((43.0f + 20.0f * (16.0f + 27.0f * beta) * x) * b2 * beta +
               alpha * (3.0f * (59.0f + 180.0f * beta - 90.0f * x) * b2 +
                        alpha * ((453.0f + 1620.0f * beta * (1.0f - x) - 455.0f * x) * beta +
                                 alpha * (8.0f * (1.0f - x) * (135.0f * beta - 11.0f)))));
  const float prefactor_num = (1.0f + 12.0f * alpha) * (1.0f + 12.0f * beta);
  const float prefactor_den = 12960.0f * alpha * alpha * alpha * b2 * (1.0f + 12.0f * total) * (total * total) * (1.0f - x);
  return prefactor_num * poly * TQ_FRCP(prefactor_den);
}

// ANY_BOUNDARY: the saddle-point formulas for alpha, beta > 6 whatever total x (1-x) is -- for a draw near an edge
// only ONE of the two directions is in the saddle-point regime (the other one is in a series regime); the caller
// keeps the direction it needs (tq_beta_grad_pair_rest).
// (does tq_beta_grad_pair_mid<false> apply?  The sampling kernels sort the draws of a workgroup by regime before they
// evaluate anything: tq_cosmos.hip, tq_site_beta_compact)
TQ_HD bool tq_beta_grad_pair_applies(double x, double alpha, double beta) {
  const double total = alpha + beta;
  const double y = 1.0 - x;
  const double xy = x * y;
  const double boundary = total * xy;
  return boundary >= 2.5 && alpha > 6.0 && beta > 6.0;
}
template <bool ANY_BOUNDARY = false>
TQ_HD bool tq_beta_grad_pair_mid(double x, double alpha, double beta, double* ga, double* gb) {
  const double total = alpha + beta;
  const double y = 1.0 - x;
  const double xy = x * y;
  const double boundary = total * xy;
  if (!((ANY_BOUNDARY || boundary >= 2.5) && alpha > 6.0 && beta > 6.0)) return false;
  // ONE reciprocal, of alpha beta total x y, gives 1/total, 1/x, 1/y, 1/alpha, 1/beta by multiplication (a Newton
  // reciprocal in double is ~7 instructions; this routine is a third of the local sampling kernel)
  const double P = alpha * beta;
  const double R = tq_drcp(P * boundary);
  const double RP = R * P;          // 1 / (total x y)
  const double rt = RP * xy;        // 1 / total
  const double Rxy = RP * total;    // 1 / (x y)
  const double rx = Rxy * y, ry = Rxy * x;
  const double Rab = R * boundary;  // 1 / (alpha beta)
  const double ra = Rab * beta, rb = Rab * alpha;
  const double mean = alpha * rt;
  // |x - mean| <= 0.1 sd, sd^2 = alpha beta / ((total+1) total^2), without the square root
  const double dev = x - mean;
  if (dev * dev * (total + 1.0) * total * total <= 0.01 * P) {
    // removable singularity at x = mean: both directions use the local polynomial (8 % of draws; handled here so
    // that a wave never has to run the generic piecewise code for them) -- in float where its products fit
    if (alpha < 2000.0 && beta < 2000.0) {
      *ga = (double)tq_beta_grad_window_f((float)x, (float)alpha, (float)beta);
      *gb = (double)tq_beta_grad_window_f((float)y, (float)beta, (float)alpha);
    } else {
      *ga = tq_beta_grad_window(x, alpha, beta);
      *gb = tq_beta_grad_window(y, beta, alpha);
    }
    return true;
  }
  const double la = tq_dlog(mean * rx);       // ln(alpha / (total x))
  const double lb = tq_dlog(beta * rt * ry);  // ln(beta / (total (1-x)))
  const double base = beta * lb + alpha * la;  // total * KL(mean || x) > 0
  const double sb = tq_dsqrt(base);
  const double q = tq_dsqrt(P);                // sqrt(alpha beta): sqrt(alpha / beta) = q / beta, sqrt(beta / alpha) = q / alpha
  const double st = tq_dsqrt(total);
  const double axbx = beta * x - alpha * y;    // alpha (x-1) + beta x = total (x - mean)
  const double bs = base * sb;
  const double M = tq_drcp(axbx * bs);         // second (and last) reciprocal: 1 / axbx and base^(-3/2)
  const double r_ax = M * bs;
  const double term4 = M * axbx;
  const double str = st * rt;                  // 1 / sqrt(total)
  const double s2 = 1.41421356237309504880 * q * str;              // sqrt(2 alpha beta / total)
  const double rs2 = 0.70710678118654752440 * st * (q * Rab);      // 1 / s2 (1 / q = q / (alpha beta))
  const float fa = TQ_FRCP(12.0f * (float)alpha), fb = TQ_FRCP(12.0f * (float)beta), ft = TQ_FRCP(12.0f * (float)total);
  const double stirling =
      (double)((1.0f + fa + 0.5f * fa * fa) * (1.0f + fb + 0.5f * fb * fb) * TQ_FRCP(1.0f + ft + 0.5f * ft * ft));
  const double r_den = r_ax * r_ax * (0.70710678118654752440 * str * rt);  // 1 / (sqrt(2) total^(3/2) axbx^2)
  const double term3 = 2.0 * s2 * r_ax;        // sqrt(8 alpha beta / total) / axbx
  const double sgn4 = (x < mean) ? term4 : -term4;
  // direction alpha
  {
    const double num = -(2.0 * alpha * alpha + P) * y - x * beta * beta;
    const double term1 = num * r_den * (q * ra);  // / sqrt(alpha / beta)
    *ga = stirling * (-x * rs2) * (term1 + 0.5 * la * (term3 + sgn4));
  }
  // direction beta: x -> 1-x, alpha <-> beta, axbx -> -axbx, mean -> 1-mean
  {
    const double num = -(2.0 * beta * beta + P) * x - y * alpha * alpha;
    const double term1 = num * r_den * (q * rb);  // / sqrt(beta / alpha)
    *gb = stirling * (-y * rs2) * (term1 + 0.5 * lb * (-term3 - sgn4));
  }
  return true;
}

// rational regime of torch's _dirichlet_grad (neither series regime, alpha or beta <= 6)
TQ_HD float tq_beta_grad_rational(float x_, float alpha_, float total_) {
  const double x = x_, alpha = alpha_, total = total_;
  const double beta = total - alpha;
  // rational correction to an analytic approximation; coefficients: PyTorch (BSD-3)
  const float c[2][3][3][4] = {
      {{{1.003668233f, -0.01061107488f, -0.0657888334f, 0.01201642863f},
        {0.6336835991f, -0.3557432599f, 0.05486251648f, -0.001465281033f},
        {-0.03276231906f, 0.004474107445f, 0.002429354597f, -0.0001557569013f}},
       {{0.221950385f, -0.3187676331f, 0.01799915743f, 0.01074823814f},
        {-0.2951249643f, 0.06219954479f, 0.01535556598f, 0.001550077057f},
        {0.02155310298f, 0.004170831599f, 0.001292462449f, 6.976601077e-05f}},
       {{-0.05980841433f, 0.008441916499f, 0.01085618172f, 0.002319392565f},
        {0.02911413504f, 0.01400243777f, -0.002721828457f, 0.000751041181f},
        {0.005900514878f, -0.001936558688f, -9.495446725e-06f, 5.385558597e-05f}}},
      {{{1.0f, -0.02924021934f, -0.04438342661f, 0.007285809825f},
        {0.6357567472f, -0.3473456711f, 0.05454656494f, -0.002407477521f},
        {-0.03301322327f, 0.004845219414f, 0.00231480583f, -0.0002307248149f}},
       {{0.5925320577f, -0.1757678135f, 0.01505928619f, 0.000564515273f},
        {0.1014815858f, -0.06589186703f, 0.01272886114f, -0.0007316646956f},
        {-0.007258481865f, 0.001096195486f, 0.0003934994223f, -4.12701925e-05f}},
       {{0.06469649321f, -0.0236701437f, 0.002902096474f, -5.896963079e-05f},
        {0.001925008108f, -0.002869809258f, 0.0008000589141f, -6.063713228e-05f},
        {-0.0003477407336f, 6.959756487e-05f, 1.097287507e-05f, -1.650964693e-06f}}},
  };
  // the O(1) rational correction in fp32 (as torch does for float32 tensors); the digamma difference: tq_digamma_diff
  const float u = TQ_FLOG(x_);
  const float a = TQ_FLOG(alpha_) - u;
  const float b = TQ_FLOG(total_) - a;
  const float pow_u[3] = {1.0f, u, u * u};
  const float pow_a[3] = {1.0f, a, a * a};
  float p = 0.0f, q = 0.0f;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const float ua = pow_u[i] * pow_a[j];
      p += ua * (c[0][i][j][0] + b * (c[0][i][j][1] + b * (c[0][i][j][2] + b * c[0][i][j][3])));
      q += ua * (c[1][i][j][0] + b * (c[1][i][j][1] + b * (c[1][i][j][2] + b * c[1][i][j][3])));
    }
  const float approx = x_ * tq_digamma_diff(total_, alpha_) * TQ_FRCP((float)beta);
  return p * TQ_FRCP(q) * approx;
}

TQ_HD int tq_dirichlet_grad_regime(double x, double alpha, double beta, double total) {
  const double boundary = total * x * (1.0 - x);
  if (x <= 0.5 && boundary < 2.5) return 0;     // x-small series (fp64)
  if (x >= 0.5 && boundary < 0.75) return 1;    // (1-x)-small series
  if (alpha > 6.0 && beta > 6.0) return 2;      // saddle point
  return 3;                                     // rational
}

TQ_HD_NOINLINE float tq_dirichlet_grad(float x_, float alpha_, float total_) {
  const double x = x_, alpha = alpha_, total = total_;
  const double beta = total - alpha;
  const int regime = tq_dirichlet_grad_regime(x, alpha, beta, total);
  if (regime == 0) return (float)tq_beta_grad_alpha_small(x, alpha, beta);
  if (regime == 1) return -tq_beta_grad_beta_small_f(1.0f - x_, total_ - alpha_, alpha_);
  if (regime == 2) return (float)tq_beta_grad_alpha_mid(x, alpha, beta);
  return tq_beta_grad_rational(x_, alpha_, total_);
}

// Both directions of one Beta draw, dd[0] = tq_dirichlet_grad(t, c1, size), dd[1] = tq_dirichlet_grad(1 - t, c0, size),
// when tq_beta_grad_pair_mid does not apply -- what the two calls return (bit for bit in the series and rational regimes;
// a direction in the saddle-point regime comes from the pair routine, same formulas, rounding-level differences), but
// arranged by REGIME instead of by direction.  Lanes of a wave fall into all regimes once a fit has converged (absent spots: size 5..20, present
// ones several hundred), and a wave executes every branch some lane takes: called twice, each of the three expensive
// regimes ran twice per wave.  The two directions of a lane never share the x-small series (x <= 1/2 for one means
// 1 - x >= 1/2 for the other), the (1-x)-small series or -- outside the pair routine -- the saddle point, so each of
// series runs ONCE with per-lane arguments from whichever direction needs it; only the (float) rational regime can be
// needed by both.
TQ_HD_NOINLINE void tq_beta_grad_pair_rest(float t, float c1, float c0, float size, float* dd) {
  const float xf[2] = {t, 1.0f - t};
  const float af[2] = {c1, c0};
  const double total = size;
  int regime[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) regime[j] = tq_dirichlet_grad_regime((double)xf[j], (double)af[j], total - (double)af[j], total);
  dd[0] = dd[1] = 0.0f;
  // saddle-point regime of ONE direction (both: tq_beta_grad_pair_mid has returned true and this routine is not
  // called): the pair routine once, keeping the direction(s) that asked for it
#if defined(TQ_DIAG_NO_R2)
  if (false) {
#else
  if (regime[0] == 2 || regime[1] == 2) {
#endif
    double ga = 0.0, gb = 0.0;
    if (tq_beta_grad_pair_mid<true>((double)t, (double)c1, total - (double)c1, &ga, &gb)) {
      if (regime[0] == 2) dd[0] = (float)ga;
      if (regime[1] == 2) dd[1] = (float)gb;
    } else {  // (the two directions disagree about alpha, beta > 6 within rounding: the plain evaluation)
#pragma nounroll
      for (int j = 0; j < 2; ++j)
        if (regime[j] == 2) dd[j] = (float)tq_beta_grad_alpha_mid((double)xf[j], (double)af[j], total - (double)af[j]);
    }
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const bool n0 = regime[0] == r, n1 = regime[1] == r;
    // pass 0: every lane that needs regime r evaluates ONE direction (the first if both do); pass 1: the second
    // direction of lanes that need it for both (x = 1/2 exactly) -- a run-time loop, so the regime's code exists once
#pragma nounroll
    for (int pass = 0; pass < 2; ++pass) {
      const bool act = pass == 0 ? (n0 || n1) : (n0 && n1);
      if (act) {
        const bool first = pass == 0 && n0;
        const float xs = first ? xf[0] : xf[1], as = first ? af[0] : af[1];
        float g;
#if defined(TQ_DIAG_NO_R0)  // (diagnostic builds: scripts/gpu_site_diag.sh)
        if (r == 0) g = xs;
#else
        if (r == 0) g = (float)tq_beta_grad_alpha_small((double)xs, (double)as, total - (double)as);
#endif
#if defined(TQ_DIAG_NO_R1)
        else g = as;
#else
        else g = -tq_beta_grad_beta_small_f(1.0f - xs, size - as, as);
#endif
        if (first) dd[0] = g;
        else dd[1] = g;
      }
    }
  }
#pragma nounroll
  for (int j = 0; j < 2; ++j)
#if defined(TQ_DIAG_NO_R3)
    if (regime[j] == 3) dd[j] = xf[j];
#else
    if (regime[j] == 3) dd[j] = tq_beta_grad_rational(xf[j], af[j], size);
#endif
}
