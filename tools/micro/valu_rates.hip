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
      REP64(asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(p), "+v"(q), "+v"(r), "+v"(s), "+v"(x), "+v"(y), "+v"(z), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");) \
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

// SGPR-operand forms: is a vector instruction that reads a scalar register (a lane mask, a constant) as cheap as one that does not?
BENCH(cndmask_vcc_set, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_lt_u32 vcc, %2, %3\n v_cndmask_b32 %3, %3, %0, vcc")
BENCH(cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %3, %3, %0, s[20:21]")
BENCH(cndmask_indep, "v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %2, %3, %0, vcc\n v_cndmask_b32 %3, %0, %1, vcc")
BENCH(add_u32_sgpr, "v_add_u32 %0, s20, %0\n v_add_u32 %1, s21, %1\n v_add_u32 %2, s20, %2\n v_add_u32 %3, s21, %3")
BENCH(add_u32_lit, "v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3")
BENCH(add_u32_inl, "v_add_u32 %0, 7, %0\n v_add_u32 %1, 7, %1\n v_add_u32 %2, 7, %2\n v_add_u32 %3, 7, %3")
BENCH(mov_from_sgpr, "v_mov_b32 %0, s20\n v_mov_b32 %1, s21\n v_mov_b32 %2, s20\n v_mov_b32 %3, s21")
BENCH(mov_vgpr, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0")
BENCH(cmp_to_sgpr, "v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cmp_lt_u32_e64 s[22:23], %1, %2\n v_cmp_lt_u32_e64 s[20:21], %2, %3\n v_cmp_lt_u32_e64 s[22:23], %3, %0")
BENCH(fma_f64_sgpr, "v_fma_f64 %8, %8, %9, s[20:21]\n v_fma_f64 %9, %9, %10, s[20:21]\n v_fma_f64 %10, %10, %11, s[20:21]\n v_fma_f64 %11, %11, %8, s[20:21]")

__global__ void k_clock(unsigned long long* out, int iters) {
  // both counters around a fixed amount of work (no loop that waits for a clock: a bounded kernel whatever the clocks do)
  uint32_t a = threadIdx.x, b = 3;
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));)
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = r1 - r0;
    out[2] = a;
  }
}

// ---- latencies of the memory paths a beam-kernel frame waits on: dependent chains (each access needs the previous result) ----
// mode 0: scalar loads (constant memory through the scalar cache); 1: LDS reads; 2: global loads that hit in L2 (a 64 KB ring
// per wave, touched once before the timed loop); 3: global loads of a 1 KB ring (vector L1 hits)
__global__ void k_latency(unsigned long long* out, const uint32_t* ring, int mode, int iters, uint32_t ring_words) {
  __shared__ uint32_t lds[1024];
  const uint32_t* mine = ring + (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * ring_words;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = (uint32_t)((i * 37 + 11) & 1023);
  __syncthreads();
  uint32_t idx = threadIdx.x & 63;
  unsigned long long t0 = 0, t1 = 0;
  if (mode == 0) {
    uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x & 15));
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
      s = *(const uint32_t __attribute__((address_space(4)))*)(const void*)(ring + (s & 1023));  // uniform address: s_load_dword
      s = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
    }
    t1 = __builtin_readcyclecounter();
    idx = s;
  } else if (mode == 1) {
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) idx = lds[idx & 1023];
    t1 = __builtin_readcyclecounter();
  } else {
    const uint32_t mask = ring_words - 1;
    for (uint32_t i = threadIdx.x & 63; i < ring_words; i += 64) idx += mine[i] & 1u;  // bring the ring into L2
    idx &= mask;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) idx = mine[idx & mask];
    t1 = __builtin_readcyclecounter();
  }
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
  if (idx == 0xFFFFFFFFu) out[0] = 0;
}

struct Entry {
  const char* name;
  void (*fn)(unsigned long long*, int);
};
#define E(NAME) {#NAME, k_##NAME}
static const Entry kAll[] = {E(add_u32), E(xor_b32), E(mul_lo_u32), E(mul_hi_u32), E(mul_u32_u24), E(mad_u64_u32), E(lshl_add_u64),
                             E(lshlrev_b64), E(lshrrev_b64), E(cmp_lt_u64), E(cmp_lt_u32), E(cndmask), E(fma_f64), E(add_f64), E(mul_f64),
                             E(max_f64), E(mov_dpp), E(max_u32_dpp), E(alignbit), E(readlane), E(bcnt), E(cvt_f64_u32), E(rcp_f64),
                             E(exp_f32), E(pk_fma_f32), E(fma_f32), E(cndmask_vcc_set), E(cndmask_sgpr), E(cndmask_indep),
                             E(add_u32_sgpr), E(add_u32_lit), E(add_u32_inl), E(mov_from_sgpr), E(mov_vgpr), E(cmp_to_sgpr),
                             E(fma_f64_sgpr)};

int main() {
  unsigned long long* out;
  hipMalloc(&out, 8 * 8192);
  const int iters = 64;
  {
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, out, 20000);
    hipDeviceSynchronize();
    unsigned long long h[2];
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("cycle counter: %.1f ticks per microsecond (s_memtime against the 100 MHz s_memrealtime)\n", (double)h[0] / ((double)h[1] / 100.0));
    printf("   (%llu cycle-counter ticks, %llu real-time ticks for 1 280 000 dependent v_add_u32 of one wave)\n", h[0], h[1]);
    fflush(stdout);
  }
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
    fflush(stdout);
  }
  {
    // rings: word i holds the index of the next word to read (a stride that scatters over the ring's cache lines)
    const uint32_t big = 16384, small = 256;  // words per wave: 64 KB (L2) / 1 KB (L1)
    const int waves = 256 * 16;
    std::vector<uint32_t> h((size_t)waves * big);
    for (int w = 0; w < waves; ++w)
      for (uint32_t i = 0; i < big; ++i) h[(size_t)w * big + i] = (i * 1031u + 97u) & (big - 1);
    uint32_t* ring;
    hipMalloc(&ring, h.size() * 4);
    hipMemcpy(ring, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> hs((size_t)waves * small);
    for (int w = 0; w < waves; ++w)
      for (uint32_t i = 0; i < small; ++i) hs[(size_t)w * small + i] = (i * 37u + 11u) & (small - 1);
    uint32_t* ring_s;
    hipMalloc(&ring_s, hs.size() * 4);
    hipMemcpy(ring_s, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    const char* names[4] = {"s_load (scalar cache)", "ds_read_b32", "global_load, 64 KB ring per wave (L2)", "global_load, 1 KB ring per wave (L1)"};
    printf("\n%-42s %22s %22s\n", "dependent access", "cycles, 1 wave per CU", "cycles, 16 waves per CU");
    for (int mode = 0; mode < 4; ++mode) {
      double res[2];
      for (int occ = 0; occ < 2; ++occ) {
        const int threads = occ == 0 ? 64 : 1024, blocks = 256, it = 2000;
        const uint32_t* r = mode == 3 ? ring_s : ring;
        const uint32_t words = mode == 3 ? small : big;
        hipLaunchKernelGGL(k_latency, dim3(blocks), dim3(threads), 0, 0, out, r, mode, 50, words);
        hipLaunchKernelGGL(k_latency, dim3(blocks), dim3(threads), 0, 0, out, r, mode, it, words);
        hipDeviceSynchronize();
        const int n = blocks * (threads / 64);
        std::vector<unsigned long long> hv(n);
        hipMemcpy(hv.data(), out, 8 * n, hipMemcpyDeviceToHost);
        double sum = 0;
        for (auto v : hv) sum += (double)v;
        res[occ] = sum / n / it;
      }
      printf("%-42s %22.0f %22.0f\n", names[mode], res[0], res[1]);
      fflush(stdout);
    }
  }
  return 0;
}
