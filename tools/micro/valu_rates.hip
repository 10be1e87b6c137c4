// valu_rates.hip -- what a wave64 vector instruction costs on one SIMD of gfx950, measured so that the answer cannot be
// better than the hardware: EVERY wave of the launch records its own start / end ticks and the SIMD it ran on
// (HW_REG_HW_ID), and a SIMD's cost per instruction is
//        (last end - first start over the waves of that SIMD) / (instructions those waves issued together).
// Round 5's version timed only wave 0 of each block -- the oldest wave of its SIMD, which wins the issue arbitration -- and
// divided its time by four: a lone-wave latency, not a throughput (it implied 240 TFLOPS of fp32 on a 157 TFLOPS chip).
// This version prints, per instruction and for 1 / 2 / 3 / 4 waves per SIMD: cycles per instruction per SIMD (throughput),
// the mean and the slowest wave's cycles per instruction (what one wave sees), and aborts when v_fma_f32 comes out
// faster than the chip's vector peak allows (2 cycles per wave64 instruction on a SIMD-32).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/micro/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

struct Rec {
  unsigned long long t0, t1;  // s_memtime
  unsigned long long r0, r1;  // s_memrealtime (100 MHz)
  uint32_t hw_id, pad;
};

__device__ __forceinline__ uint32_t hw_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return v;
}

#define BENCH(NAME, ASM)                                                                                   \
  __global__ void k_##NAME(Rec* out, int iters) {                                                          \
    uint32_t a = threadIdx.x, b = threadIdx.x * 3u + 1u, c = threadIdx.x ^ 0x55u, d = threadIdx.x + 7u;    \
    uint64_t p = threadIdx.x * 0x9E3779B97F4A7C15ull + 1, q = p ^ 0xABCDEFull, r = p + 5, s = q + 9;         \
    double x = 1.0 + threadIdx.x, y = 0.5, z = 0.25, w = 2.0;                                              \
    __syncthreads();                                                                                       \
    const unsigned long long r0 = wall_clock64();                                                          \
    const unsigned long long t0 = __builtin_readcyclecounter();                                            \
    for (int i = 0; i < iters; ++i) {                                                                      \
      REP64(asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(p), "+v"(q), "+v"(r), "+v"(s), "+v"(x), "+v"(y), "+v"(z), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");) \
    }                                                                                                      \
    const unsigned long long t1 = __builtin_readcyclecounter();                                            \
    const unsigned long long r1 = wall_clock64();                                                          \
    if ((threadIdx.x & 63) == 0) {                                                                         \
      Rec& o = out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)];                                   \
      o.t0 = t0; o.t1 = t1; o.r0 = r0; o.r1 = r1; o.hw_id = hw_id(); o.pad = 0;                             \
    }                                                                                                      \
    if (a + b + c + d + (uint32_t)(p + q + r + s) + (uint32_t)(x + y + z + w) == 0x12345u) out[0].pad = 1;  \
  }

// each ASM string is FOUR instructions on independent registers
BENCH(add_u32, "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0")
BENCH(xor_b32, "v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0")
BENCH(and_or_b32, "v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %1, %1, %2, %3\n v_and_or_b32 %2, %2, %3, %0\n v_and_or_b32 %3, %3, %0, %1")
BENCH(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0")
BENCH(mul_hi_u32, "v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %2\n v_mul_hi_u32 %2, %2, %3\n v_mul_hi_u32 %3, %3, %0")
BENCH(mul_u32_u24, "v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %0")
BENCH(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1")
BENCH(mad_u64_u32, "v_mad_u64_u32 %4, vcc, %0, %1, %4\n v_mad_u64_u32 %5, vcc, %1, %2, %5\n v_mad_u64_u32 %6, vcc, %2, %3, %6\n v_mad_u64_u32 %7, vcc, %3, %0, %7")
BENCH(lshl_add_u64, "v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 0, %6\n v_lshl_add_u64 %6, %6, 0, %7\n v_lshl_add_u64 %7, %7, 0, %4")
BENCH(lshlrev_b64, "v_lshlrev_b64 %4, 3, %4\n v_lshlrev_b64 %5, 3, %5\n v_lshlrev_b64 %6, 3, %6\n v_lshlrev_b64 %7, 3, %7")
BENCH(lshrrev_b64, "v_lshrrev_b64 %4, 3, %4\n v_lshrrev_b64 %5, 3, %5\n v_lshrrev_b64 %6, 3, %6\n v_lshrrev_b64 %7, 3, %7")
BENCH(cmp_lt_u64, "v_cmp_lt_u64 vcc, %4, %5\n v_cmp_lt_u64 vcc, %5, %6\n v_cmp_lt_u64 vcc, %6, %7\n v_cmp_lt_u64 vcc, %7, %4")
BENCH(cmp_eq_u64, "v_cmp_eq_u64 vcc, %4, %5\n v_cmp_eq_u64 vcc, %5, %6\n v_cmp_eq_u64 vcc, %6, %7\n v_cmp_eq_u64 vcc, %7, %4")
BENCH(cmp_lt_u32, "v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %0")
BENCH(cmp_lt_f64, "v_cmp_lt_f64 vcc, %8, %9\n v_cmp_lt_f64 vcc, %9, %10\n v_cmp_lt_f64 vcc, %10, %11\n v_cmp_lt_f64 vcc, %11, %8")
BENCH(cndmask, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc")
BENCH(fma_f64, "v_fma_f64 %8, %8, %9, %10\n v_fma_f64 %9, %9, %10, %11\n v_fma_f64 %10, %10, %11, %8\n v_fma_f64 %11, %11, %8, %9")
BENCH(add_f64, "v_add_f64 %8, %8, %9\n v_add_f64 %9, %9, %10\n v_add_f64 %10, %10, %11\n v_add_f64 %11, %11, %8")
BENCH(mul_f64, "v_mul_f64 %8, %8, %9\n v_mul_f64 %9, %9, %10\n v_mul_f64 %10, %10, %11\n v_mul_f64 %11, %11, %8")
BENCH(max_f64, "v_max_f64 %8, %8, %9\n v_max_f64 %9, %9, %10\n v_max_f64 %10, %10, %11\n v_max_f64 %11, %11, %8")
BENCH(mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
BENCH(max_u32_dpp, "v_max_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %1, %2, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %2, %3, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %3, %0, %3 row_shr:8 row_mask:0xf bank_mask:0xf")
BENCH(mov_dpp_bcast, "v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_bcast:31 row_mask:0xc bank_mask:0xf")
BENCH(alignbit, "v_alignbit_b32 %0, %0, %1, 7\n v_alignbit_b32 %1, %1, %2, 7\n v_alignbit_b32 %2, %2, %3, 7\n v_alignbit_b32 %3, %3, %0, 7")
BENCH(readlane, "v_readlane_b32 s20, %0, 5\n v_readlane_b32 s21, %1, 6\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 8")
BENCH(readfirstlane, "v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3")
BENCH(bcnt, "v_bcnt_u32_b32 %0, %0, %1\n v_bcnt_u32_b32 %1, %1, %2\n v_bcnt_u32_b32 %2, %2, %3\n v_bcnt_u32_b32 %3, %3, %0")
BENCH(mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0\n v_mbcnt_hi_u32_b32 %1, %2, %1\n v_mbcnt_lo_u32_b32 %2, %3, %2\n v_mbcnt_hi_u32_b32 %3, %0, %3")
BENCH(cvt_f64_u32, "v_cvt_f64_u32 %8, %0\n v_cvt_f64_u32 %9, %1\n v_cvt_f64_u32 %10, %2\n v_cvt_f64_u32 %11, %3")
BENCH(rcp_f64, "v_rcp_f64 %8, %8\n v_rcp_f64 %9, %9\n v_rcp_f64 %10, %10\n v_rcp_f64 %11, %11")
BENCH(exp_f32, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3")
BENCH(log_f32, "v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3")
BENCH(rcp_f32, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3")
BENCH(pk_fma_f32, "v_pk_fma_f32 %4, %4, %5, %6\n v_pk_fma_f32 %5, %5, %6, %7\n v_pk_fma_f32 %6, %6, %7, %4\n v_pk_fma_f32 %7, %7, %4, %5")
BENCH(pk_mul_f32, "v_pk_mul_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %6\n v_pk_mul_f32 %6, %6, %7\n v_pk_mul_f32 %7, %7, %4")
BENCH(fma_f32, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1")
BENCH(max3_f32, "v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1")
BENCH(cvt_f64_f32, "v_cvt_f64_f32 %8, %0\n v_cvt_f64_f32 %9, %1\n v_cvt_f64_f32 %10, %2\n v_cvt_f64_f32 %11, %3")
// SGPR-operand forms
BENCH(cndmask_vcc_set, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_lt_u32 vcc, %2, %3\n v_cndmask_b32 %3, %3, %0, vcc")
BENCH(cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %3, %3, %0, s[20:21]")
BENCH(add_u32_sgpr, "v_add_u32 %0, s20, %0\n v_add_u32 %1, s21, %1\n v_add_u32 %2, s20, %2\n v_add_u32 %3, s21, %3")
BENCH(add_u32_lit, "v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3")
BENCH(mov_vgpr, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0")
BENCH(cmp_to_sgpr, "v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cmp_lt_u32_e64 s[22:23], %1, %2\n v_cmp_lt_u32_e64 s[20:21], %2, %3\n v_cmp_lt_u32_e64 s[22:23], %3, %0")
BENCH(fma_f64_sgpr, "v_fma_f64 %8, %8, %9, s[20:21]\n v_fma_f64 %9, %9, %10, s[20:21]\n v_fma_f64 %10, %10, %11, s[20:21]\n v_fma_f64 %11, %11, %8, s[20:21]")
// ONE dependency chain (every instruction needs its predecessor): the latency a lone chain pays
BENCH(dep_add_u32, "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %2\n v_add_u32 %0, %0, %3\n v_add_u32 %0, %0, %1")
BENCH(dep_fma_f64, "v_fma_f64 %8, %8, %9, %10\n v_fma_f64 %8, %8, %9, %11\n v_fma_f64 %8, %8, %9, %10\n v_fma_f64 %8, %8, %9, %11")
BENCH(dep_fma_f32, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %3\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %3")
BENCH(dep_mad_u64_u32, "v_mad_u64_u32 %4, vcc, %0, %1, %4\n v_mad_u64_u32 %4, vcc, %1, %2, %4\n v_mad_u64_u32 %4, vcc, %2, %3, %4\n v_mad_u64_u32 %4, vcc, %3, %0, %4")
BENCH(dep_cmp_cndmask, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n v_cmp_lt_u32 vcc, %0, %3\n v_cndmask_b32 %0, %0, %1, vcc")
BENCH(dep_dpp, "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf")
// scalar unit (one per CU, shared by the sixteen waves)
BENCH(s_add_u32, "s_add_u32 s20, s20, s21\n s_add_u32 s21, s21, s22\n s_add_u32 s22, s22, s23\n s_add_u32 s23, s23, s20")
BENCH(s_and_b64, "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[22:23], s[22:23], s[20:21]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[22:23], s[22:23], s[20:21]")
BENCH(mix_valu_salu, "v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, s21\n v_add_u32 %2, %2, %3\n s_add_u32 s22, s22, s23")

struct Entry {
  const char* name;
  void (*fn)(Rec*, int);
};
#define E(NAME) {#NAME, k_##NAME}
static const Entry kAll[] = {
    E(add_u32), E(xor_b32), E(and_or_b32), E(mul_lo_u32), E(mul_hi_u32), E(mul_u32_u24), E(mad_u32_u24), E(mad_u64_u32),
    E(lshl_add_u64), E(lshlrev_b64), E(lshrrev_b64), E(cmp_lt_u64), E(cmp_eq_u64), E(cmp_lt_u32), E(cmp_lt_f64), E(cndmask),
    E(fma_f64), E(add_f64), E(mul_f64), E(max_f64), E(mov_dpp), E(max_u32_dpp), E(mov_dpp_bcast), E(alignbit), E(readlane),
    E(readfirstlane), E(bcnt), E(mbcnt), E(cvt_f64_u32), E(rcp_f64), E(exp_f32), E(log_f32), E(rcp_f32), E(pk_fma_f32),
    E(pk_mul_f32), E(fma_f32), E(max3_f32), E(cvt_f64_f32), E(cndmask_vcc_set), E(cndmask_sgpr), E(add_u32_sgpr),
    E(add_u32_lit), E(mov_vgpr), E(cmp_to_sgpr), E(fma_f64_sgpr), E(dep_add_u32), E(dep_fma_f64), E(dep_fma_f32),
    E(dep_mad_u64_u32), E(dep_cmp_cndmask), E(dep_dpp), E(s_add_u32), E(s_and_b64), E(mix_valu_salu)};

struct Stat {
  double per_simd, wave_mean, wave_max, mhz;
  int simds_used, max_waves_on_simd;
};

// HW_ID on gfx9: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx950: xcc in its own register)
static Stat measure(const Entry& e, Rec* out, int waves_per_block, int blocks, int iters) {
  const int threads = waves_per_block * 64, n = blocks * waves_per_block;
  hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(threads), 0, 0, out, 2);
  hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipDeviceSynchronize();
  std::vector<Rec> h(n);
  hipMemcpy(h.data(), out, sizeof(Rec) * n, hipMemcpyDeviceToHost);
  const double insts = iters * 256.0;
  Stat s{};
  double sum_simd = 0, sum_wave = 0, sum_mhz = 0;
  int n_simd = 0;
  for (int b = 0; b < blocks; ++b) {
    // a block's waves share one CU (a workgroup never spans CUs); group them by the SIMD they report
    std::map<uint32_t, std::vector<const Rec*>> by_simd;
    for (int w = 0; w < waves_per_block; ++w) {
      const Rec& r = h[b * waves_per_block + w];
      by_simd[(r.hw_id >> 4) & 3].push_back(&r);
      const double per = (double)(r.t1 - r.t0) / insts;
      sum_wave += per;
      s.wave_max = std::max(s.wave_max, per);
      sum_mhz += (double)(r.t1 - r.t0) / ((double)(r.r1 - r.r0) / 100.0);
    }
    for (auto& kv : by_simd) {
      unsigned long long lo = ~0ull, hi = 0;
      for (const Rec* r : kv.second) lo = std::min(lo, r->t0), hi = std::max(hi, r->t1);
      sum_simd += (double)(hi - lo) / (insts * kv.second.size());
      ++n_simd;
      s.max_waves_on_simd = std::max(s.max_waves_on_simd, (int)kv.second.size());
    }
  }
  s.per_simd = sum_simd / n_simd;
  s.wave_mean = sum_wave / n;
  s.mhz = sum_mhz / n;
  s.simds_used = n_simd;
  return s;
}

int main(int argc, char** argv) {
  Rec* out;
  hipMalloc(&out, sizeof(Rec) * 256 * 16);
  const int iters = 64, blocks = 256;
  printf("# 256 back-to-back instructions x %d iterations per wave; 256 blocks (one per CU unless the dispatcher doubles up:\n", iters);
  printf("# the 'w/simd' column is the largest number of waves any SIMD reported); s_memtime ticks; 'MHz' = s_memtime ticks per\n");
  printf("# microsecond of the 100 MHz s_memrealtime over the same interval.\n");
  printf("# per-SIMD = (last end - first start of the waves on one SIMD) / (instructions they issued together): THROUGHPUT.\n");
  printf("# wave mean / max = one wave's own end - start per instruction: what the chain of one wave sees.\n");
  printf("%-16s", "instruction");
  for (int w = 1; w <= 4; ++w) printf(" | %dw/SIMD: per-SIMD wave-mean wave-max", w);
  printf(" | lone wave/CU | MHz(4w) w/simd\n");
  bool ok = true;
  for (const Entry& e : kAll) {
    printf("%-16s", e.name);
    Stat s4{};
    for (int w = 1; w <= 4; ++w) {
      const Stat s = measure(e, out, 4 * w, blocks, iters);
      printf(" | %8.2f %8.2f %8.2f          ", s.per_simd, s.wave_mean, s.wave_max);
      if (w == 4) s4 = s;
    }
    const Stat lone = measure(e, out, 1, blocks, iters);
    printf(" | %8.2f     | %6.0f %d\n", lone.wave_mean, s4.mhz, s4.max_waves_on_simd);
    fflush(stdout);
    if (!strcmp(e.name, "fma_f32") && s4.per_simd < 1.9 && s4.max_waves_on_simd == 4) {
      printf("!! v_fma_f32 at %.2f cycles per wave64 instruction per SIMD would exceed the chip's 157.3 TFLOPS vector peak "
             "(2 cycles on a SIMD-32): the measurement is wrong\n", s4.per_simd);
      ok = false;
    }
  }
  return ok ? 0 : 1;
}
