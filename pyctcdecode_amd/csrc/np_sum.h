// np_sum.h -- numpy's float summation order, restated (host + device).
//
// The reference decides "probabilities or logits?" with  math.isclose(logits.sum(axis=1).mean(), 1)  (decoder.py:760,
// rel_tol 1e-9) ON THE INPUT DTYPE. For float32 / float16 inputs the neighbours of 1 are 6e-8 / 5e-4 away, so the test is in
// effect "does the mean round to exactly 1" -- which depends on the order numpy adds in. That order is pinned here:
//   * add.reduce over a contiguous axis = pairwise summation (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum):
//       n < 8      : plain loop starting from 0
//       n <= 128   : eight accumulators r[j] over a[j], a[j+8], ...; ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)); then the
//                    n % 8 leftover elements one by one
//       otherwise  : n2 = n/2 rounded down to a multiple of 8; pairwise(a, n2) + pairwise(a + n2, n - n2)
//     float16 rows are accumulated in float32 by that routine and rounded to float16 once per row;
//   * ndarray.mean (numpy/_core/_methods.py:_mean): the same reduction (float16: with a float32 accumulator), a true
//     division by the count in the accumulator type, float16 results rounded back to float16.
// Checked against numpy 2.2.6 on this container by tests/test_np_sum.py (random C-contiguous arrays, every dtype);
// for other memory layouts numpy iterates differently and so may the last bit.
#pragma once
#include <stdint.h>

#include "common.h"

namespace ctc {

// Acc: the accumulator type (float for float32 / float16 / bfloat16 rows, double for float64 rows); Get(i) -> Acc
template <class Acc, class Get>
CTC_HD Acc np_pairwise_leaf(Get get, int64_t a, int64_t n) {
  if (n < 8) {
    Acc res = (Acc)0;
    for (int64_t i = 0; i < n; ++i) res = res + get(a + i);
    return res;
  }
  Acc r0 = get(a), r1 = get(a + 1), r2 = get(a + 2), r3 = get(a + 3), r4 = get(a + 4), r5 = get(a + 5), r6 = get(a + 6),
      r7 = get(a + 7);
  int64_t i = 8;
  for (; i < n - (n % 8); i += 8) {
    r0 = r0 + get(a + i);
    r1 = r1 + get(a + i + 1);
    r2 = r2 + get(a + i + 2);
    r3 = r3 + get(a + i + 3);
    r4 = r4 + get(a + i + 4);
    r5 = r5 + get(a + i + 5);
    r6 = r6 + get(a + i + 6);
    r7 = r7 + get(a + i + 7);
  }
  Acc res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (; i < n; ++i) res = res + get(a + i);
  return res;
}

// the recursion, without recursion: an explicit stack of (offset, length, state, left sum)
template <class Acc, class Get>
CTC_HD Acc np_pairwise(Get get, int64_t n) {
  constexpr int DEPTH = 28;  // lengths halve down to 128: 2^34 elements
  int64_t off[DEPTH], len[DEPTH];
  Acc left[DEPTH];
  int state[DEPTH];  // 0: nothing done, 1: left half done
  int sp = 0;
  off[0] = 0;
  len[0] = n;
  state[0] = 0;
  left[0] = (Acc)0;
  Acc ret = (Acc)0;
  for (;;) {
    if (len[sp] <= 128) {
      ret = np_pairwise_leaf<Acc>(get, off[sp], len[sp]);
      --sp;
    } else if (state[sp] == 0) {
      int64_t n2 = len[sp] / 2;
      n2 -= n2 % 8;
      state[sp] = 1;
      off[sp + 1] = off[sp];
      len[sp + 1] = n2;
      state[sp + 1] = 0;
      ++sp;
      continue;
    }
    // a child of frame sp has returned `ret` (or the root was a leaf)
    for (;;) {
      if (sp < 0) return ret;
      if (state[sp] == 1) {  // that was the left half: now the right one
        int64_t n2 = len[sp] / 2;
        n2 -= n2 % 8;
        left[sp] = ret;
        state[sp] = 2;
        off[sp + 1] = off[sp] + n2;
        len[sp + 1] = len[sp] - n2;
        state[sp + 1] = 0;
        ++sp;
        break;
      }
      // state 2: both halves done
      ret = left[sp] + ret;
      --sp;
    }
  }
}

// float32 -> float16 / bfloat16 bits, round to nearest even (what numpy's npy_float_to_half does; bfloat16 is not a
// numpy dtype: same rule)
CTC_HD uint16_t f32_to_f16_bits(float f) {
  union { float f; uint32_t u; } c;
  c.f = f;
  const uint32_t x = c.u;
  const uint32_t sign = (x >> 16) & 0x8000u;
  const uint32_t ex = (x >> 23) & 0xFFu, man = x & 0x7FFFFFu;
  if (ex == 0xFFu) return (uint16_t)(sign | 0x7C00u | (man ? 0x200u | (man >> 13) : 0u));  // inf / nan
  const int e = (int)ex - 127 + 15;
  if (e >= 31) return (uint16_t)(sign | 0x7C00u);  // overflow -> inf
  if (e <= 0) {                                      // subnormal half or zero
    if (e < -10) return (uint16_t)sign;
    const uint32_t m = man | 0x800000u;
    const int shift = 14 - e;  // 14..24
    uint32_t h = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1u))) ++h;
    return (uint16_t)(sign | h);
  }
  uint32_t h = ((uint32_t)e << 10) | (man >> 13);
  const uint32_t rem = man & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;  // (a carry into the exponent is the right result)
  return (uint16_t)(sign | h);
}
CTC_HD float f16_bits_to_f32(uint16_t h) {
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t ex = (h >> 10) & 31u, man = h & 1023u;
  union { float f; uint32_t u; } c;
  if (ex == 0) {
    if (man == 0) {
      c.u = sign;
      return c.f;
    }
    int e = -1;
    do {
      ++e;
      man <<= 1;
    } while (!(man & 1024u));
    c.u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 1023u) << 13);
    return c.f;
  }
  if (ex == 31) {
    c.u = sign | 0x7F800000u | (man << 13);
    return c.f;
  }
  c.u = sign | ((ex + 112u) << 23) | (man << 13);
  return c.f;
}
CTC_HD uint16_t f32_to_bf16_bits(float f) {
  union { float f; uint32_t u; } c;
  c.f = f;
  if ((c.u & 0x7F800000u) == 0x7F800000u && (c.u & 0x7FFFFFu)) return (uint16_t)((c.u >> 16) | 0x40u);  // nan
  const uint32_t lsb = (c.u >> 16) & 1u;
  return (uint16_t)((c.u + 0x7FFFu + lsb) >> 16);
}
CTC_HD float bf16_bits_to_f32(uint16_t h) {
  union { float f; uint32_t u; } c;
  c.u = (uint32_t)h << 16;
  return c.f;
}

// logits.sum(axis=1)[t] of a C-contiguous [T, V] matrix of dtype code 0 f32 / 1 f64 / 2 f16 / 3 bf16, as numpy computes
// it, widened (exactly) to double
CTC_HD double np_row_sum(const void* x, int dtype, int64_t t, int64_t V) {
  if (dtype == 1) {
    const double* p = (const double*)x + t * V;
    return np_pairwise<double>([p](int64_t i) { return p[i]; }, V);
  }
  if (dtype == 0) {
    const float* p = (const float*)x + t * V;
    return (double)np_pairwise<float>([p](int64_t i) { return p[i]; }, V);
  }
  const uint16_t* p = (const uint16_t*)x + t * V;
  if (dtype == 2) {
    const float s = np_pairwise<float>([p](int64_t i) { return f16_bits_to_f32(p[i]); }, V);
    return (double)f16_bits_to_f32(f32_to_f16_bits(s));
  }
  const float s = np_pairwise<float>([p](int64_t i) { return bf16_bits_to_f32(p[i]); }, V);
  return (double)bf16_bits_to_f32(f32_to_bf16_bits(s));
}
// row_sums.mean() of T such sums (held as doubles), as numpy computes it, widened to double; T > 0
CTC_HD double np_mean_of_sums(const double* rs, int dtype, int64_t T) {
  if (dtype == 1) return np_pairwise<double>([rs](int64_t i) { return rs[i]; }, T) / (double)T;
  const float m = np_pairwise<float>([rs](int64_t i) { return (float)rs[i]; }, T) / (float)T;
  if (dtype == 2) return (double)f16_bits_to_f32(f32_to_f16_bits(m));
  if (dtype == 3) return (double)bf16_bits_to_f32(f32_to_bf16_bits(m));
  return (double)m;
}
// math.isclose(mean, 1) with the default rel_tol = 1e-9 (decoder.py:760); never true for a non-finite mean
CTC_HD bool np_mean_is_one(double mean) {
  const double d = mean - 1.0, a = mean < 0 ? -mean : mean;
  const double ad = d < 0 ? -d : d;
  return mean == mean && a <= 1.7976931348623157e308 && ad <= 1e-9 * (a > 1.0 ? a : 1.0);
}

}  // namespace ctc
