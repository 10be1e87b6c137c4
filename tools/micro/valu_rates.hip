// valu_rates.hip -- issue cost of the VALU instructions the beam kernels lean on, in cycles per wave64 instruction on one
// SIMD of gfx950: 256 back-to-back instructions of one kind (four independent dependency chains) between two
// s_memtime reads, one wave per CU and then four waves per SIMD (the throughput figure).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/micro/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

#define BENCH(NAME, ASM)                                                                                   \
  __global__ void k_##NAME(unsigned long long* out, int iters) {                                           \
    uint32_t a = threadIdx.x, b = threadIdx.x * 3u + 1u, c = threadIdx.x ^ 0x55u, d = threadIdx.x + 7u;    \
    uint64_t p = threadIdx.x * 0x9E3779B97F4A7C15ull + 1, q = p ^ 0xABCDEFull, r = p + 5, s = q + 9;         \
    double x = 1.0 + threadIdx.x, y = 0.5, z = 0.25, w = 2.0;                                              \
    unsigned long long t0 = __builtin_readcyclecounter();                                                  \
    for (int i = 0; i < iters; ++i) {                                                                      \
      REP64(asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(p), "+v"(q), "+v"(r), "+v"(s), "+v"(x), "+v"(y), "+v"(z), "+v"(w) : : "vcc");) \
    }                                                                                                      \
    unsigned long long t1 = __builtin_readcyclecounter();                                                  \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                       \
    if (a + b + c + d + (uint32_t)(p + q + r + s) + (uint32_t)(x + y + z + w) == 0x12345u) out[0] = 0;     \
  }

// each ASM string is FOUR instructions on independent registers
BENCH(add_u32, "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0")
BENCH(xor_b32, "v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0")
BENCH(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0")
BENCH(mul_hi_u32, "v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %2\n v_mul_hi_u32 %2, %2, %3\n v_mul_hi_u32 %3, %3, %0")
BENCH(mul_u32_u24, "v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %0")
BENCH(mad_u64_u32, "v_mad_u64_u32 %4, vcc, %0, %1, %4\n v_mad_u64_u32 %5, vcc, %1, %2, %5\n v_mad_u64_u32 %6, vcc, %2, %3, %6\n v_mad_u64_u32 %7, vcc, %3, %0, %7")
BENCH(lshl_add_u64, "v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 0, %6\n v_lshl_add_u64 %6, %6, 0, %7\n v_lshl_add_u64 %7, %7, 0, %4")
BENCH(lshlrev_b64, "v_lshlrev_b64 %4, 3, %4\n v_lshlrev_b64 %5, 3, %5\n v_lshlrev_b64 %6, 3, %6\n v_lshlrev_b64 %7, 3, %7")
BENCH(lshrrev_b64, "v_lshrrev_b64 %4, 3, %4\n v_lshrrev_b64 %5, 3, %5\n v_lshrrev_b64 %6, 3, %6\n v_lshrrev_b64 %7, 3, %7")
BENCH(cmp_lt_u64, "v_cmp_lt_u64 vcc, %4, %5\n v_cmp_lt_u64 vcc, %5, %6\n v_cmp_lt_u64 vcc, %6, %7\n v_cmp_lt_u64 vcc, %7, %4")
BENCH(cmp_lt_u32, "v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %0")
BENCH(cndmask, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc")
BENCH(fma_f64, "v_fma_f64 %8, %8, %9, %10\n v_fma_f64 %9, %9, %10, %11\n v_fma_f64 %10, %10, %11, %8\n v_fma_f64 %11, %11, %8, %9")
BENCH(add_f64, "v_add_f64 %8, %8, %9\n v_add_f64 %9, %9, %10\n v_add_f64 %10, %10, %11\n v_add_f64 %11, %11, %8")
BENCH(mul_f64, "v_mul_f64 %8, %8, %9\n v_mul_f64 %9, %9, %10\n v_mul_f64 %10, %10, %11\n v_mul_f64 %11, %11, %8")
BENCH(max_f64, "v_max_f64 %8, %8, %9\n v_max_f64 %9, %9, %10\n v_max_f64 %10, %10, %11\n v_max_f64 %11, %11, %8")
BENCH(mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
BENCH(max_u32_dpp, "v_max_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %1, %2, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %2, %3, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %3, %0, %3 row_shr:8 row_mask:0xf bank_mask:0xf")
BENCH(alignbit, "v_alignbit_b32 %0, %0, %1, 7\n v_alignbit_b32 %1, %1, %2, 7\n v_alignbit_b32 %2, %2, %3, 7\n v_alignbit_b32 %3, %3, %0, 7")
BENCH(readlane, "v_readlane_b32 s20, %0, 5\n v_readlane_b32 s21, %1, 6\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 8")
BENCH(bcnt, "v_bcnt_u32_b32 %0, %0, %1\n v_bcnt_u32_b32 %1, %1, %2\n v_bcnt_u32_b32 %2, %2, %3\n v_bcnt_u32_b32 %3, %3, %0")
BENCH(cvt_f64_u32, "v_cvt_f64_u32 %8, %0\n v_cvt_f64_u32 %9, %1\n v_cvt_f64_u32 %10, %2\n v_cvt_f64_u32 %11, %3")
BENCH(rcp_f64, "v_rcp_f64 %8, %8\n v_rcp_f64 %9, %9\n v_rcp_f64 %10, %10\n v_rcp_f64 %11, %11")
BENCH(exp_f32, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3")
BENCH(pk_fma_f32, "v_pk_fma_f32 %4, %4, %5, %6\n v_pk_fma_f32 %5, %5, %6, %7\n v_pk_fma_f32 %6, %6, %7, %4\n v_pk_fma_f32 %7, %7, %4, %5")
BENCH(fma_f32, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1")

struct Entry {
  const char* name;
  void (*fn)(unsigned long long*, int);
};
#define E(NAME) {#NAME, k_##NAME}
static const Entry kAll[] = {E(add_u32), E(xor_b32), E(mul_lo_u32), E(mul_hi_u32), E(mul_u32_u24), E(mad_u64_u32), E(lshl_add_u64),
                             E(lshlrev_b64), E(lshrrev_b64), E(cmp_lt_u64), E(cmp_lt_u32), E(cndmask), E(fma_f64), E(add_f64), E(mul_f64),
                             E(max_f64), E(mov_dpp), E(max_u32_dpp), E(alignbit), E(readlane), E(bcnt), E(cvt_f64_u32), E(rcp_f64),
                             E(exp_f32), E(pk_fma_f32), E(fma_f32)};

int main() {
  unsigned long long* out;
  hipMalloc(&out, 8 * 4096);
  const int iters = 64;
  printf("%-14s %18s %18s\n", "instruction", "cyc/inst 1 wave/CU", "cyc/inst 4 waves/SIMD (per SIMD)");
  for (const Entry& e : kAll) {
    double res[2];
    for (int mode = 0; mode < 2; ++mode) {
      const int threads = mode == 0 ? 64 : 1024;  // 1024 threads = 16 waves on one CU = 4 per SIMD
      const int blocks = 256;
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(threads), 0, 0, out, 4);  // warm-up
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(threads), 0, 0, out, iters);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(blocks);
      hipMemcpy(h.data(), out, 8 * blocks, hipMemcpyDeviceToHost);
      double sum = 0;
      for (auto v : h) sum += (double)v;
      const double per_wave = sum / blocks / (iters * 256.0);  // cycles per instruction as one wave sees them
      res[mode] = mode == 0 ? per_wave : per_wave / 4.0;       // four waves share the SIMD: per-SIMD issue cost
    }
    printf("%-14s %18.2f %18.2f\n", e.name, res[0], res[1]);
  }
  return 0;
}
