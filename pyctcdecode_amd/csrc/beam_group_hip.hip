// beam_group_hip.hip -- the workgroup kernel of the beam stage (beam_core.h: one workgroup per utterance) for gfx950, in a
// translation unit of its own.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>

#include "backend.h"
#include "beam_core.h"
#include "wave_ops_hip.h"

namespace ctc {
namespace be {

#define HIP_TRY_G(expr)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      if (err) *err = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
      return -1;                                                                         \
    }                                                                                    \
  } while (0)

// ---------------------------------------------------------------------------------------------
// beam kernel
// ---------------------------------------------------------------------------------------------
struct GpuCtx {
  int tid, nt;
  // LDS-only barrier: waits for this wave's LDS traffic, not for global loads/stores in flight
  // (prefetches and arena stores keep going across it)
  __device__ __forceinline__ void sync() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  // full barrier: also makes this workgroup's global stores visible to its other waves
  __device__ __forceinline__ void sync_mem() { __syncthreads(); }
  // LDS atomics (ds_*): workgroup scope, relaxed -- phases are separated by s_barrier
  __device__ __forceinline__ uint32_t atomic_add(CTC_LDS uint32_t* p, uint32_t v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ void atomic_or(CTC_LDS uint32_t* p, uint32_t v) {
    __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ void atomic_min(CTC_LDS uint32_t* p, uint32_t v) {
    __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ void atomic_max(CTC_LDS uint32_t* p, uint32_t v) {
    __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ void atomic_max64(CTC_LDS uint64_t* p, uint64_t v) {
    __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ uint32_t atomic_cas(CTC_LDS uint32_t* p, uint32_t cmp, uint32_t val) {
    __hip_atomic_compare_exchange_strong(p, &cmp, val, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_WORKGROUP);
    return cmp;
  }
  // max over the 64 lanes of the calling wave (all lanes must call it)
  __device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) { return wave_max_u64_split(v); }
  __device__ __forceinline__ bool is_wave_leader() { return (threadIdx.x & 63) == 0; }
  __device__ __forceinline__ int wave_width() { return 64; }
  __device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
  __device__ __forceinline__ int clz64(uint64_t x) { return __clzll((long long)x); }
  __device__ __forceinline__ int popc64(uint64_t x) { return __popcll(x); }
  __device__ __forceinline__ void use(double x) { asm volatile("" ::"v"(x)); }
  __device__ __forceinline__ unsigned long long clock() { return (unsigned long long)wall_clock64(); }
  __device__ __forceinline__ unsigned long long global_add(unsigned long long* p, unsigned long long v) {
    return atomicAdd(p, v);
  }
};

// Two waves per SIMD (= two 256-thread workgroups per CU, which is also what the 80 KB of LDS allow): the
// second launch-bound caps the allocator at 256 VGPRs -- left alone it drifts above that with small code
// changes and silently halves the residency (measured: 13.2 ms -> 25.4 ms).
template <int BW, int NT, bool MULTI>
__global__ __launch_bounds__(NT, NT > 256 ? 1 : 2) void beam_decode(BeamArgs a, int surv_cap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int u = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
  // compile-time layout: every LDS array sits at a constant offset (ds_* immediate offsets)
  LdsShape shape;
  shape.bw = BW;
  constexpr int CAND = BW <= 128 && NT > 256 ? CAND_CHUNK_WIDE : CAND_CHUNK;  // = group_cand(BW, NT > 256)
  shape.cand = CAND;
  shape.tab = CAND > CAND_CHUNK ? 4 * CAND : 2 * CAND;  // = group_tab(CAND)
  shape.pool = CAND + BW;
  shape.sortn = 2 * CAND;
  shape.surv = surv_cap;
  LdsView view;
  lds_carve(view, (lds_bytes_t)smem, shape);
  UttIO io;
  const int64_t r0 = a.utt_row0[u];
  io.surv_cnt = a.surv_cnt + r0;
  io.surv_id = a.surv_id + (size_t)r0 * a.params.max_surv;
  io.surv_lp = a.surv_lp + (size_t)r0 * a.params.max_surv;
  io.T = (int32_t)(a.utt_row0[u + 1] - r0);
  io.text_nodes = a.text_nodes + a.text_off[u];
  io.text_cap = (uint32_t)(a.text_off[u + 1] - a.text_off[u]);
  io.emit_nodes = a.emit_nodes + a.emit_off[u];
  io.emit_cap = (uint32_t)(a.emit_off[u + 1] - a.emit_off[u]);
  const uint32_t n_lms = MULTI ? a.tables.n_lms : 1u;
  io.start_state = a.start_states ? a.start_states + (size_t)u * n_lms : nullptr;
  io.out_xstates = (MULTI && a.out_xstates) ? a.out_xstates + (size_t)u * a.out_stride * (n_lms - 1) : nullptr;
  io.out = a.out + (size_t)u * a.out_stride;
  io.n_out = a.n_out + u;
  io.status = a.status + u;
  io.tok_pool = a.tok_pool;
  io.tok_pool_head = a.tok_pool_head;
  io.tok_pool_cap = a.tok_pool_cap;
  io.prof = (u == 0) ? a.prof : nullptr;
  io.imports = (a.imports && !a.resident_in) ? a.imports + a.import_off[u] : nullptr;
  io.n_import = (a.imports && !a.resident_in) ? (int32_t)(a.import_off[u + 1] - a.import_off[u]) : 0;
    io.import_xstates = (a.imports && a.import_xstates && !a.resident_in) ? a.import_xstates + (size_t)a.import_off[u] * (n_lms - 1) : nullptr;
  io.first_frame = a.first_frames ? a.first_frames[u] : a.params.first_frame;
  io.cold = nullptr;
  io.pay = nullptr;
  io.carry_out = a.carry_out ? a.carry_out + (size_t)u * a.carry_stride : nullptr;
  io.carry_xstates = (a.carry_out && a.carry_xstates) ? a.carry_xstates + (size_t)u * a.carry_stride * (n_lms - 1) : nullptr;
  io.sstate = a.sstate ? a.sstate + u : nullptr;
  io.emit_start = a.sstate ? a.sstate[u].emit_next : 0u;
  io.want_out = a.want_out;
  if (a.resident_in) {
    io.imports = a.imports + (size_t)u * a.carry_stride;
    io.n_import = (int32_t)a.sstate[u].n_carry;
    io.import_xstates = a.import_xstates ? a.import_xstates + (size_t)u * a.carry_stride * (n_lms - 1) : nullptr;
  }
  GpuCtx ctx{(int)threadIdx.x, NT};
  BeamDecoder<GpuCtx, MULTI> dec(ctx, view, shape, a.tables, a.params, io);
  dec.run();
}

template <int BW, int NT, bool MULTI>
static int launch_beam_t(const BeamArgs& a, const LdsShape& shape, size_t lds, hipStream_t stream, std::string* err) {
  HIP_TRY_G(hipFuncSetAttribute((const void*)beam_decode<BW, NT, MULTI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds));
  hipLaunchKernelGGL((beam_decode<BW, NT, MULTI>), dim3((unsigned)a.n_utts), dim3(NT), lds, stream, a, shape.surv);
  return 0;
}

template <int NT, bool MULTI>
static int launch_beam_nt(const BeamArgs& a, const LdsShape& shape, size_t lds, hipStream_t stream, std::string* err) {
  switch (shape.bw) {
    case 32: return launch_beam_t<32, NT, MULTI>(a, shape, lds, stream, err);
    case 64: return launch_beam_t<64, NT, MULTI>(a, shape, lds, stream, err);
    case 128: return launch_beam_t<128, NT, MULTI>(a, shape, lds, stream, err);
    default: return launch_beam_t<256, NT, MULTI>(a, shape, lds, stream, err);
  }
}


int launch_group(const BeamArgs& a, const LdsShape& shape, size_t lds, int kind, hipStream_t stream, std::string* err) {
  if (kind == 2) return launch_beam_nt<256, true>(a, shape, lds, stream, err);  // MultiLanguageModel
  if (kind == 1) return launch_beam_nt<512, false>(a, shape, lds, stream, err);
  return launch_beam_nt<256, false>(a, shape, lds, stream, err);
}

}  // namespace be
}  // namespace ctc
