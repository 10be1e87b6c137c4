// beam_wave_hip.hip -- the wave kernel of the beam stage (beam_wave.h: one wavefront per utterance) for gfx950, in a
// translation unit of its own (it is the slowest part of the build and the one that is rebuilt most often).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>

#include "backend.h"
#include "beam_core.h"
#include "beam_wave.h"
#include "wave_ops_hip.h"

namespace ctc {
namespace be {

#define HIP_TRY_W(expr)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      if (err) *err = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
      return -1;                                                                         \
    }                                                                                    \
  } while (0)

// ---------------------------------------------------------------------------------------------
// wave kernel: one wavefront per utterance (beam_wave.h)
// ---------------------------------------------------------------------------------------------
typedef const BeamArgs __attribute__((address_space(4))) * KernArgs;  // the kernel's argument block (constant address space)
struct WaveGpuCtx {
  int lane;
  KernArgs ka;
  // Issue priority (s_setprio). The SIMD's arbiter prefers, at equal priority, the OLDEST of its ready waves: of four resident
  // waves the first-dispatched runs at nearly a lone wave's pace and the last gets what is left (tools/micro/valu_rates.hip: 5.1 vs
  // 10 - 17 cycles per instruction; CTCDEC_WAVE_TIMES on the 4096-utterance launch: the four age ranks of a SIMD finish after
  // 12.9 / 13.7 / 14.5 / 15.6 ms on average) -- and the launch lasts as long as its slowest wave (18.0 ms) while the average wave
  // is done after 79 % of that. Two remedies, both a handful of scalar instructions per 16 frames:
  //   rotation (mode 1 + k): the four priority levels go round the waves of a SIMD every 2^k frames -- equal shares
  //     (age ranks: 13.9 .. 14.6 ms; launch 16.8 ms);
  //   by remaining work (mode 32): utterances differ (candidates per frame), so equal shares still end 12 - 17 ms apart. Every
  //     16 frames a wave adds its progress to one counter of the launch and reads back everybody's: whoever has more frames
  //     left than the average runs at a higher level, whoever is ahead at a lower one.
  // Nothing is held for it across the frame loop (the kernel has no scalar register to spare): the mode and the counter's
  // address are re-read from the argument block, and the wave's slot on its SIMD (HW_ID.wave_id) stands in for its age.
  __device__ __forceinline__ void set_prio(uint32_t p) {
    switch (p) {  // (the level is an immediate of the instruction)
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      default: __builtin_amdgcn_s_setprio(3); break;
    }
  }
  // the frames [t, t2) have just been decoded; T: the utterance's length
  __device__ __forceinline__ void frame_done(int t, int t2, int T) {
    if ((((uint32_t)t ^ (uint32_t)t2) >> 4) == 0u) return;  // (every 16 frames)
    const KernArgs k = fresh();
    const int32_t mode = k->prio_mode;
    if (mode == 0) return;
    if (mode < 32) {
      uint32_t hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(hw));
      set_prio((hw + ((uint32_t)t2 >> (mode - 1))) & 3u);
      return;
    }
    const uint32_t units = (((uint32_t)t2 >> 4) - ((uint32_t)t >> 4)) << 4;
    unsigned long long old = 0;
    if (lane == 0) old = atomicAdd(k->progress, (unsigned long long)units);
    const uint32_t done = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)old) + units;  // (launches of < 2^32 frames)
    const float avg_left = (float)((uint32_t)k->total_frames - done) * k->inv_n_utts;
    // (mode 33: a frame of this utterance costs `weight` average frames -- its survivors per frame, counted by the prune stage)
    const float weight = mode == 33 ? k->block_weight[blockIdx.x] : 1.f;
    const float d = (float)(T - t2) * weight - avg_left;  // > 0: behind the launch's average
    set_prio(d > 24.f ? 3u : d > 0.f ? 2u : d > -24.f ? 1u : 0u);
  }
  // mode 33: the heavy utterances run at the top level from their first frame on (they are known before the launch)
  __device__ __forceinline__ void start_prio(int T) {
    const KernArgs k = fresh();
    if (k->prio_mode != 33) return;
    const float d = (float)T * k->block_weight[blockIdx.x] - (float)(uint32_t)k->total_frames * k->inv_n_utts;
    set_prio(d > 24.f ? 3u : d > 0.f ? 2u : d > -24.f ? 1u : 0u);
  }
  // launch constants, re-read where they are used (scalar loads; see WaveDecoder::tab)
  __device__ __forceinline__ KernArgs fresh() const {
    KernArgs p = ka;
    asm volatile("" : "+s"(p));
    return p;
  }
  __device__ __forceinline__ const DeviceTables& tables() const { return *(const DeviceTables*)&fresh()->tables; }
  __device__ __forceinline__ const DecodeParams& params() const { return *(const DecodeParams*)&fresh()->params; }
  // One wave issues its LDS operations in order: what the lanes exchange through LDS only needs the compiler
  // to keep the program order of the accesses.
  __device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // global stores of this wave complete before anything that follows
  __device__ __forceinline__ void mem_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  // every global load / store of this wave issued so far has completed
  // (the builtin, not inline assembly: the compiler's own wait-count bookkeeping sees it and does not wait again)
  __device__ __forceinline__ void vm_wait() { __builtin_amdgcn_s_waitcnt(0x0F70); }  // vmcnt(0), expcnt / lgkmcnt untouched
  __device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
  __device__ __forceinline__ int popc64(uint64_t x) { return __popcll(x); }
  __device__ __forceinline__ int clz64(uint64_t x) { return __clzll((long long)x); }
  __device__ __forceinline__ int ctz64(uint64_t x) { return __builtin_ctzll(x); }
  __device__ __forceinline__ int clz32(uint32_t x) { return __clz((int)x); }
  __device__ __forceinline__ int ctz32(uint32_t x) { return __builtin_ctz(x); }
  __device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
  // the value, but nothing computed from it may be scheduled above this point
  __device__ __forceinline__ uint32_t opaque32(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
  }
  __device__ __forceinline__ uint32_t bcast32(uint32_t v, int src) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src));
  }
  __device__ __forceinline__ uint32_t shfl32(uint32_t v, int src) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v);
  }
  __device__ __forceinline__ uint64_t bcast64(uint64_t v, int src) {
    const int s = __builtin_amdgcn_readfirstlane(src);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, s);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), s);
    return ((uint64_t)hi << 32) | lo;
  }
  __device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) { return wave_max_u64_split(v); }
  __device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
    // (status bits: rare) any lane with a bit set makes it wave-wide
    if (__ballot((v & 0xFFu) != 0u) == 0ull) return v;  // nothing raised (every frame but a handful): one ballot, not eight
    uint32_t r = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b)
      if (__ballot((v >> b) & 1u)) r |= 1u << b;
    return r | (v & ~0xFFu);
  }
  __device__ __forceinline__ uint32_t wave_excl_sum_u32(uint32_t v) {  // (finalisation only)
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
      if ((int)(threadIdx.x & 63) >= off) incl += o;
    }
    return incl - v;
  }
  __device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    // DPP crossbar (row_shr 1/2/4/8, row_bcast 15/31): six VALU steps, no LDS round trips
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
  }
  __device__ __forceinline__ void lds_max_u64(CTC_LDS uint64_t* p, uint64_t v) {
    __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  __device__ __forceinline__ void lds_or_u32(CTC_LDS uint32_t* p, uint32_t v) {
    __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  __device__ __forceinline__ unsigned long long clock() { return (unsigned long long)wall_clock64(); }
  __device__ __forceinline__ unsigned long long global_add(unsigned long long* p, unsigned long long v) {
    return atomicAdd(p, v);
  }
};

// Four waves per SIMD: 128 registers per lane (the allocator is told so: left alone it takes what it likes and halves the
// residency) and, by wave_lds_bytes, at most 10 KB of LDS at beam_width <= 100 -- sixteen utterances per CU.
// (beam_width 101 .. 128: 12.7 KB of LDS allow twelve waves per CU, three per SIMD: 168 registers)
template <int BW, int ORD, bool PROF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BW <= 100 ? 4 : 3, BW <= 100 ? 4 : 3))) void beam_wave(BeamArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int u = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
  WaveLds view;
  wave_lds_carve<BW>(view, (lds_bytes_t)smem);
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
  io.start_state = a.start_states ? a.start_states + (size_t)u : nullptr;
  io.out_xstates = nullptr;
  io.out = a.out + (size_t)u * a.out_stride;
  io.n_out = a.n_out + u;
  io.status = a.status + u;
  io.tok_pool = a.tok_pool;
  io.tok_pool_head = a.tok_pool_head;
  io.tok_pool_cap = a.tok_pool_cap;
  io.prof = (u == 0) ? a.prof : nullptr;
  io.imports = (a.imports && !a.resident_in) ? a.imports + a.import_off[u] : nullptr;
  io.n_import = (a.imports && !a.resident_in) ? (int32_t)(a.import_off[u + 1] - a.import_off[u]) : 0;
  io.import_xstates = nullptr;
  io.first_frame = a.first_frames ? a.first_frames[u] : a.params.first_frame;
  io.cold = a.cold + (size_t)u * 2 * COLD_STRIDE;
  io.pay = a.pay + (size_t)u * a.pay_stride;
  io.carry_out = a.carry_out ? a.carry_out + (size_t)u * a.carry_stride : nullptr;
  io.carry_xstates = (a.carry_out && a.carry_xstates) ? a.carry_xstates + (size_t)u * a.carry_stride * (1u - 1) : nullptr;
  io.sstate = a.sstate ? a.sstate + u : nullptr;
  io.emit_start = a.sstate ? a.sstate[u].emit_next : 0u;
  io.want_out = a.want_out;
  if (a.resident_in) {
    io.imports = a.imports + (size_t)u * a.carry_stride;
    io.n_import = (int32_t)a.sstate[u].n_carry;
    io.import_xstates = a.import_xstates ? a.import_xstates + (size_t)u * a.carry_stride * 0 : nullptr;
  }
  WaveGpuCtx ctx{(int)threadIdx.x, (KernArgs)__builtin_amdgcn_kernarg_segment_ptr()};
  // diagnostics: when did this wave run, and where (the record's address lives in two VECTOR registers across the frame loop:
  // the kernel has none of the scalar kind to spare)
  unsigned long long* rec = a.wave_clock ? a.wave_clock + (size_t)blockIdx.x * 4 : nullptr;
  asm volatile("" : "+v"(rec));
  if (rec && threadIdx.x == 0) {
    rec[0] = wall_clock64();
    rec[3] = (unsigned long long)io.T | ((unsigned long long)(uint32_t)u << 32);
  }
  ctx.start_prio(io.T);
  WaveDecoder<WaveGpuCtx, BW, ORD, PROF> dec(ctx, view, io);
  dec.run();
  if (rec && threadIdx.x == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    rec[1] = wall_clock64();
    rec[2] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
  }
}


template <int BW, int ORD, bool PROF>
static int launch_wave_p(const BeamArgs& a, hipStream_t stream, std::string* err) {
  const size_t lds = wave_lds_bytes<BW>();
  HIP_TRY_W(hipFuncSetAttribute((const void*)beam_wave<BW, ORD, PROF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((beam_wave<BW, ORD, PROF>), dim3((unsigned)a.n_utts), dim3(64), lds, stream, a);
  return 0;
}
template <int BW, int ORD>
static int launch_wave_t(const BeamArgs& a, hipStream_t stream, std::string* err) {
  // (phase timers: their own instantiation, launched only while ctcdec_profile_phases is on)
  return a.prof ? launch_wave_p<BW, ORD, true>(a, stream, err) : launch_wave_p<BW, ORD, false>(a, stream, err);
}
template <int BW>
static int launch_wave_bw(const BeamArgs& a, hipStream_t stream, std::string* err) {
  // n-gram orders compiled in: up to 4 (the usual models; no LM at all runs here too) or up to MAX_CTX + 1
  if (!a.tables.has_lm || a.tables.lm_order <= 4) return launch_wave_t<BW, 4>(a, stream, err);
  return launch_wave_t<BW, MAX_CTX + 1>(a, stream, err);
}

int launch_wave(const BeamArgs& a, hipStream_t stream, std::string* err) {
  switch (wave_bucket(a.params.beam_width)) {
    case 64: return launch_wave_bw<64>(a, stream, err);
    case 100: return launch_wave_bw<100>(a, stream, err);
    default: return launch_wave_bw<128>(a, stream, err);
  }
}

}  // namespace be
}  // namespace ctc
