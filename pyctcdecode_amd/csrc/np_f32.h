// np_f32.h -- numpy's float32 exp and log, restated bit for bit (host + device).
//
// For float32 logits the reference computes its log-softmax in float32 (decoder.py:180-197:  x - max - log(sum(exp(x - max)))
// with numpy float32 ufuncs), and the beam order inside runs of equal scores hinges on the last bit of those values. On the
// machines numpy ships SIMD kernels for -- x86-64 with AVX512F or AVX2 + FMA3, i.e. where the goldens were generated and where
// the reference runs in practice -- float32 exp / log are NOT the C library's: numpy/_core/src/umath/
// loops_exponent_log.dispatch.c.src computes
//   exp(x):  k = rint(x * log2(e)) by the 1.5 * 2^23 trick; r = x - k * ln2 in two fused steps (Cody-Waite, ln2 split in a
//            high part with trailing zero bits and a low part); exp(r) = P5(r) / Q2(r), a rational minimax approximation,
//            every step one fused multiply-add and one IEEE division; result scaled by 2^k;
//   log(x):  x = m * 2^e with m in [0.5, 1) (getmant / getexp); if m <= sqrt(1/2): m = 2m, e = e - 1;  t = m - 1;
//            log(1 + t) = P5(t) / Q5(t);  result = fma(e, ln2, P5 / Q5).
// The constants are numpy's (npy_simd_data.h). Restated from the published algorithm, and pinned against numpy itself:
// tests/test_np_f32.py compares both functions with numpy 2.2.6 on this container for every float32 in the ranges the
// log-softmax can reach when run with --exhaustive (tools/np_f32_exhaustive.py: all 2^32 bit patterns of exp's domain that do
// not overflow, all positive finite floats for log) and on 4 M sampled arguments in the default suite.
// Arguments outside what a log-softmax produces (exp: x > 0 up to overflow handling, NaN; log: x <= 0, inf, NaN, denormals)
// follow numpy's special-case rules as well, so the functions are total.
#pragma once
#include <math.h>
#include <stdint.h>

#include "common.h"

// Every rounding below is numpy's: a multiply followed by an add must stay two roundings wherever numpy has two (compilers
// contract a * b + c into one fused operation by default when the target has one -- hipcc does).
#if defined(__clang__)
#define CTC_NP_F32_FN
#define CTC_NP_F32_BODY _Pragma("clang fp contract(off)")
#else
#define CTC_NP_F32_FN __attribute__((optimize("fp-contract=off")))
#define CTC_NP_F32_BODY
#endif

namespace ctc {

CTC_HD float np_f32_from_bits(uint32_t u) {
  union { uint32_t u; float f; } c;
  c.u = u;
  return c.f;
}
CTC_HD uint32_t np_f32_bits(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  return c.u;
}

// numpy's SIMD float32 exp (AVX512F / AVX2+FMA3 kernels: the same arithmetic)
CTC_NP_F32_FN CTC_HD float np_exp_f32(float x) {
  CTC_NP_F32_BODY
  const float xmax = 88.72283935546875f, xmin = -103.97208404541015625f;
  if (x != x) return x;                       // NaN stays NaN
  if (x >= xmax) return INFINITY;             // (overflow; +inf included)
  if (x <= xmin) return 0.0f;                 // (underflow; -inf included)
  float q = x * 1.442695040888963407359924681001892137f;  // log2(e)
  q = (q + 12582912.0f) - 12582912.0f;        // round to nearest (0x1.8p23)
  float r = fmaf(q, -6.93145752e-1f, x);      // Cody-Waite, high part of ln 2
  r = fmaf(q, -1.42860677e-6f, r);            // ... low part
  float num = fmaf(5.082762527590693718096e-04f, r, 6.757896990527504603057e-03f);
  num = fmaf(num, r, 5.114512081637298353406e-02f);
  num = fmaf(num, r, 2.473615434895520810817e-01f);
  num = fmaf(num, r, 7.257664613233124478488e-01f);
  num = fmaf(num, r, 9.999999999980870924916e-01f);
  float den = fmaf(2.159509375685829852307e-02f, r, -2.742335390411667452936e-01f);
  den = fmaf(den, r, 1.000000000000000000000e+00f);
  const float p = num / den;
  return ldexpf(p, (int)q);                   // scalef: exact, gradual underflow below 2^-126
}

// numpy's SIMD float32 log
CTC_NP_F32_FN CTC_HD float np_log_f32(float x) {
  CTC_NP_F32_BODY
  if (x != x) return x;
  if (x < 0.0f) return NAN;
  if (x == 0.0f) return -INFINITY;
  if (x == INFINITY) return INFINITY;
  // x = m * 2^e, m in [0.5, 1) (denormals are normalised first, as vgetexp / vgetmant do)
  uint32_t u = np_f32_bits(x);
  int e;
  if ((u & 0x7F800000u) == 0u) {
    const float xs = x * 8388608.0f;  // 2^23
    u = np_f32_bits(xs);
    e = (int)(u >> 23) - 126 - 23;
  } else {
    e = (int)(u >> 23) - 126;
  }
  float m = np_f32_from_bits((u & 0x007FFFFFu) | 0x3F000000u);
  float ef = (float)e;
  if (m <= 0.70710678118654752440f) {  // sqrt(1/2)
    m = m + m;
    ef = ef - 1.0f;
  }
  const float t = m - 1.0f;
  float num = fmaf(2.589979117907922693523e-02f, t, 3.808837741388407920751e-01f);
  num = fmaf(num, t, 1.480000633576506585156e+00f);
  num = fmaf(num, t, 2.112677543073053063722e+00f);
  num = fmaf(num, t, 9.999999999999998702752e-01f);
  num = fmaf(num, t, 0.000000000000000000000e+00f);
  float den = fmaf(5.875095403124574342950e-03f, t, 1.546476374983906719538e-01f);
  den = fmaf(den, t, 9.864942958519418960339e-01f);
  den = fmaf(den, t, 2.453006071784736363091e+00f);
  den = fmaf(den, t, 2.612677543073109236779e+00f);
  den = fmaf(den, t, 1.000000000000000000000e+00f);
  const float p = num / den;
  return fmaf(ef, 0.693147180559945309417232121458176568f, p);
}

}  // namespace ctc
