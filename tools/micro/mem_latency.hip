// mem_latency.hip -- latencies of the memory paths a beam-kernel frame waits on: dependent chains (each access needs the
// previous result), one wave per CU and sixteen waves per CU; every wave's time is recorded and averaged.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mem_latency tools/micro/mem_latency.hip && /tmp/mem_latency
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

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

int main() {
  unsigned long long* out;
  hipMalloc(&out, 8 * 8192);
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
