// wave_ops_hip.h -- wave64 cross-lane helpers shared by the HIP translation units (device code only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ctc {
namespace be {

// ---------------------------------------------------------------------------------------------
// wave helpers (wave64)
// ---------------------------------------------------------------------------------------------
// Wave-wide reductions through the DPP crossbar (row_shr 1/2/4/8 inside each row of 16 lanes, then
// row_bcast15 / row_bcast31 across the rows -- the gfx9 scan pattern): six VALU steps, no LDS round trips
// (the __shfl_xor form goes through ds_bpermute: two dependent LDS-pipeline operations per step for 64 bits).
// The total ends up in lane 63 and is broadcast from there. `ident` fills the lanes a step has no source for.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t ident, uint64_t v) {
  const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)ident, (int)(uint32_t)v, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(ident >> 32), (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xf, false);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ uint64_t bcast_lane63(uint64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
  return ((uint64_t)hi << 32) | lo;
}
#define CTC_DPP_REDUCE(V, IDENT, COMBINE)                 \
  do {                                                    \
    V = COMBINE(V, dpp_u64<0x111, 0xf>(IDENT, V)); /* row_shr:1 */  \
    V = COMBINE(V, dpp_u64<0x112, 0xf>(IDENT, V)); /* row_shr:2 */  \
    V = COMBINE(V, dpp_u64<0x114, 0xf>(IDENT, V)); /* row_shr:4 */  \
    V = COMBINE(V, dpp_u64<0x118, 0xf>(IDENT, V)); /* row_shr:8 */  \
    V = COMBINE(V, dpp_u64<0x142, 0xa>(IDENT, V)); /* row_bcast:15 into rows 1 and 3 */ \
    V = COMBINE(V, dpp_u64<0x143, 0xc>(IDENT, V)); /* row_bcast:31 into rows 2 and 3 */ \
  } while (0)
__device__ __forceinline__ uint64_t comb_add_f64(uint64_t a, uint64_t b) {
  return (uint64_t)__double_as_longlong(__longlong_as_double((long long)a) + __longlong_as_double((long long)b));
}
__device__ __forceinline__ uint64_t comb_max_f64(uint64_t a, uint64_t b) {
  return (uint64_t)__double_as_longlong(fmax(__longlong_as_double((long long)a), __longlong_as_double((long long)b)));
}
__device__ __forceinline__ uint64_t comb_max_u64(uint64_t a, uint64_t b) { return a > b ? a : b; }
// Wave-wide maximum of 32-bit unsigned values: the compiler folds each step into ONE v_max_u32_dpp (lanes a step has no source
// for read 0, the identity) -- six VALU instructions and a v_readlane.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  int x = (int)v;
#define CTC_MAX_STEP(CTRL, ROWS, BOUND)                                                              \
  {                                                                                                  \
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, x, CTRL, ROWS, 0xf, BOUND);          \
    x = (int)(o > (uint32_t)x ? o : (uint32_t)x);                                                    \
  }
  CTC_MAX_STEP(0x111, 0xf, true)
  CTC_MAX_STEP(0x112, 0xf, true)
  CTC_MAX_STEP(0x114, 0xf, true)
  CTC_MAX_STEP(0x118, 0xf, true)
  CTC_MAX_STEP(0x142, 0xa, false)
  CTC_MAX_STEP(0x143, 0xc, false)
#undef CTC_MAX_STEP
  return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}
// Wave-wide maximum of 64-bit unsigned values as two 32-bit ones: the high words, then the low words of the lanes that hold the
// highest high word (~17 VALU instructions; the 64-bit DPP reduction is six rounds of two moves, a 64-bit compare and two
// selects plus the moves that fill the lanes without a source: ~45).
__device__ __forceinline__ uint64_t wave_max_u64_split(uint64_t v) {
  const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
  const uint32_t mh = wave_max_u32(hi);
  const uint32_t ml = wave_max_u32(hi == mh ? lo : 0u);
  return ((uint64_t)mh << 32) | ml;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_min_i32(int v) {
  const int o = __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, CTRL, ROW_MASK, 0xf, false);
  return o < v ? o : v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
  v = dpp_min_i32<0x111, 0xf>(v);
  v = dpp_min_i32<0x112, 0xf>(v);
  v = dpp_min_i32<0x114, 0xf>(v);
  v = dpp_min_i32<0x118, 0xf>(v);
  v = dpp_min_i32<0x142, 0xa>(v);
  v = dpp_min_i32<0x143, 0xc>(v);
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ double wave_sum(double x) {
  uint64_t v = (uint64_t)__double_as_longlong(x);
  CTC_DPP_REDUCE(v, 0ull, comb_add_f64);  // +0.0
  return __longlong_as_double((long long)bcast_lane63(v));
}
__device__ __forceinline__ double wave_max(double x) {
  uint64_t v = (uint64_t)__double_as_longlong(x);
  CTC_DPP_REDUCE(v, 0xFFF0000000000000ull, comb_max_f64);  // -inf
  return __longlong_as_double((long long)bcast_lane63(v));
}

}  // namespace be
}  // namespace ctc
