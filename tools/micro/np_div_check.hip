// np_div_check.hip -- is the short division of np_exp_nonpos_pk (csrc/backend_hip.hip, CTC_NP_SHORT_DIV) IEEE-exact on this
// chip? For EVERY float32 r with |r| <= 0.36 (the reduced argument of numpy's float32 exp is within ln2/2 = 0.3466 of zero):
// num = P5(r), den = Q2(r) as numpy evaluates them, then  num / den  by the compiler's IEEE division against
// v_rcp_f32 + one Newton step + quotient + one fused correction. Prints how many of the ~2.1e9 arguments differ (expected: 0).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/np_div_check tools/micro/np_div_check.hip && /tmp/np_div_check
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

__global__ void check(unsigned long long* bad, uint32_t* first, uint32_t hi) {
#pragma clang fp contract(off)
  unsigned long long mine = 0, mine3 = 0;
  for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u <= hi; u += (uint64_t)gridDim.x * blockDim.x) {
    for (int sgn = 0; sgn < 2; ++sgn) {
      const uint32_t bits = (uint32_t)u | (sgn ? 0x80000000u : 0u);
      const float r = __uint_as_float(bits);
      float num = fmaf(5.082762527590693718096e-04f, r, 6.757896990527504603057e-03f);
      num = fmaf(num, r, 5.114512081637298353406e-02f);
      num = fmaf(num, r, 2.473615434895520810817e-01f);
      num = fmaf(num, r, 7.257664613233124478488e-01f);
      num = fmaf(num, r, 9.999999999980870924916e-01f);
      float den = fmaf(2.159509375685829852307e-02f, r, -2.742335390411667452936e-01f);
      den = fmaf(den, r, 1.0f);
      const float full = num / den;
      const float y0 = __builtin_amdgcn_rcpf(den);
      const float e = fmaf(-den, y0, 1.0f);
      const float y = fmaf(e, y0, y0);
      const float q0 = num * y;
      const float rem = fmaf(-den, q0, num);
      const float q = fmaf(rem, y, q0);
      if (__float_as_uint(q) != __float_as_uint(full)) {
        if (mine == 0) atomicMin(first, bits & 0x7FFFFFFFu);
        ++mine;
      }
      // the same without the Newton step on the reciprocal (three operations behind v_rcp_f32): reported, not used unless exact
      const float q0b = num * y0;
      const float remb = fmaf(-den, q0b, num);
      const float qb = fmaf(remb, y0, q0b);
      if (__float_as_uint(qb) != __float_as_uint(full)) ++mine3;
    }
  }
  if (mine) atomicAdd(bad, mine);
  if (mine3) atomicAdd(bad + 1, mine3);
}

int main() {
  unsigned long long* bad;
  uint32_t* first;
  hipMalloc(&bad, 16);
  hipMalloc(&first, 4);
  hipMemset(bad, 0, 16);
  hipMemset(first, 0xFF, 4);
  const float lim = 0.36f;
  uint32_t hi;
  memcpy(&hi, &lim, 4);
  hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, bad, first, hi);
  hipDeviceSynchronize();
  unsigned long long h = 0, h3 = 0;
  uint32_t f = 0;
  hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
  hipMemcpy(&h3, bad + 1, 8, hipMemcpyDeviceToHost);
  hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
  printf("short division (v_rcp_f32 + Newton + fused correction) vs IEEE division of numpy's exp polynomials: %llu of %llu reduced "
         "arguments differ", h, 2ull * ((unsigned long long)hi + 1ull));
  if (h) printf(" (smallest |r| bits 0x%08x)", f);
  printf("\n(without the Newton step on the reciprocal: %llu differ)\n", h3);
  return h ? 1 : 0;
}
