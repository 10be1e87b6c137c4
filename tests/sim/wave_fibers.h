// wave_fibers.h -- TEST INFRASTRUCTURE ONLY.  Runs SPMD code written for one 64-lane wavefront
// (pyctcdecode_amd/csrc/beam_wave.h) on the CPU: one cooperative fiber per lane, scheduled round-robin;
// every cross-lane operation (ballot, broadcast, reductions, wsync) is a rendezvous of all 64 fibers.
// The simulator checks what the hardware cannot: that every lane reaches the same cross-lane
// operation (same call site tag) -- a lane that skips one, or calls a different one, aborts the run.
// LDS is ordinary memory: between two rendezvous the lanes run one after the other, so code that lets
// lanes exchange data through LDS without a wsync() in between computes with stale or too-new values
// here and fails the parity tests.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <vector>

namespace wavesim {

constexpr int LANES = 64;

#if !defined(__x86_64__)
#error "the wave simulator's context switch is written for x86-64"
#endif

// save callee-saved registers + stack pointer of the running context, load another
extern "C" void wavesim_switch(void** save_sp, void* load_sp);
__asm__(
    ".text\n"
    ".globl wavesim_switch\n"
    ".type wavesim_switch,@function\n"
    "wavesim_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size wavesim_switch,.-wavesim_switch\n");

struct Wave;
static thread_local Wave* g_wave = nullptr;

struct Wave {
  static constexpr size_t STACK = 256 * 1024;
  void* sp[LANES];
  void* main_sp = nullptr;
  std::vector<char> stacks;
  bool done[LANES];
  int current = -1;
  std::function<void(int)> body;
  // rendezvous state
  uint64_t slot[2][LANES];
  uint64_t gen = 0;
  int arrived = 0;
  int tag0 = 0;

  static void trampoline() {
    Wave* w = g_wave;
    const int lane = w->current;
    w->body(lane);
    w->done[lane] = true;
    wavesim_switch(&w->sp[lane], w->main_sp);
    abort();  // a finished fiber is never resumed
  }

  void run(std::function<void(int)> fn) {
    body = std::move(fn);
    stacks.assign(STACK * LANES + 64, 0);
    for (int l = 0; l < LANES; ++l) {
      done[l] = false;
      uintptr_t top = ((uintptr_t)stacks.data() + STACK * (size_t)(l + 1)) & ~(uintptr_t)15;
      void** s = (void**)top;
      *--s = nullptr;                 // fake return address of the trampoline (keeps the ABI alignment)
      *--s = (void*)&trampoline;      // `ret` target of the first switch
      for (int k = 0; k < 6; ++k) *--s = nullptr;  // rbp rbx r12 r13 r14 r15
      sp[l] = (void*)s;
    }
    Wave* prev = g_wave;
    g_wave = this;
    gen = 0;
    arrived = 0;
    for (;;) {
      int live = 0;
      const uint64_t gen_before = gen;
      const int arrived_before = arrived;
      for (int l = 0; l < LANES; ++l) {
        if (done[l]) continue;
        ++live;
        current = l;
        wavesim_switch(&main_sp, sp[l]);
      }
      if (!live) break;
      int still = 0;
      for (int l = 0; l < LANES; ++l) still += done[l] ? 0 : 1;
      if (still && gen == gen_before && arrived == arrived_before) {
        fprintf(stderr, "wavesim: deadlock -- %d lanes wait in a cross-lane operation (tag %d) the others never reach\n",
                arrived, tag0);
        abort();
      }
    }
    g_wave = prev;
  }

  void yield(int lane) { wavesim_switch(&sp[lane], main_sp); }

  // all 64 lanes publish a value; returns the buffer holding everybody's
  const uint64_t* rendezvous(int lane, uint64_t v, int tag) {
    const uint64_t my_gen = gen;
    const int buf = (int)(my_gen & 1u);
    if (arrived == 0) tag0 = tag;
    else if (tag != tag0) {
      fprintf(stderr, "wavesim: lanes diverged -- lane %d is at cross-lane operation %d, others at %d\n", lane, tag, tag0);
      abort();
    }
    slot[buf][lane] = v;
    if (++arrived == LANES) {
      arrived = 0;
      ++gen;
    } else {
      while (gen == my_gen) yield(lane);
    }
    return slot[buf];
  }
};

// the execution context beam_wave.h is written against
struct SimWaveCtx {
  int lane;
  Wave* w;
  const ctc::DeviceTables* tabp;
  const ctc::DecodeParams* prmp;
  const ctc::DeviceTables& tables() const { return *tabp; }
  const ctc::DecodeParams& params() const { return *prmp; }
  const uint64_t* all(uint64_t v, int tag) { return w->rendezvous(lane, v, tag); }
  void wsync() { all(0, 1); }
  void mem_sync() { all(0, 2); }
  void vm_wait() {}  // (device: every global access issued so far has completed)
  void frame_done(int, int, int) {}  // (device: issue priority among the waves of a SIMD)
  uint64_t ballot(bool p) {
    const uint64_t* s = all(p ? 1u : 0u, 3);
    uint64_t m = 0;
    for (int l = 0; l < LANES; ++l) m |= (s[l] & 1ull) << l;
    return m;
  }
  int popc64(uint64_t x) { return __builtin_popcountll(x); }
  int clz64(uint64_t x) { return __builtin_clzll(x); }
  int ctz64(uint64_t x) { return __builtin_ctzll(x); }
  int clz32(uint32_t x) { return __builtin_clz(x); }
  int ctz32(uint32_t x) { return __builtin_ctz(x); }
  uint32_t uni32(uint32_t v) {  // a value every lane must agree on
    const uint64_t* s = all(v, 4);
    for (int l = 1; l < LANES; ++l)
      if (s[l] != s[0]) {
        fprintf(stderr, "wavesim: uni32 called with a non-uniform value (lane %d: %llu, lane 0: %llu)\n", l,
                (unsigned long long)s[l], (unsigned long long)s[0]);
        abort();
      }
    return (uint32_t)s[0];
  }
  uint32_t opaque32(uint32_t v) { return v; }
  uint32_t bcast32(uint32_t v, int src) {
    const uint64_t* s = all(((uint64_t)(uint32_t)src << 32) | v, 5);
    const int from = (int)(s[0] >> 32);
    for (int l = 1; l < LANES; ++l)
      if ((int)(s[l] >> 32) != from) {
        fprintf(stderr, "wavesim: bcast32 with a non-uniform source lane\n");
        abort();
      }
    return (uint32_t)s[from & 63];
  }
  uint32_t shfl32(uint32_t v, int src) {  // lane-wise gather: every lane names its own source lane
    const uint64_t* s = all(v, 11);
    return (uint32_t)s[src & 63];
  }
  uint64_t bcast64(uint64_t v, int src) {
    const int from = (int)uni32((uint32_t)src) & 63;
    const uint64_t* s = all(v, 6);
    return s[from];
  }
  uint64_t wave_max_u64(uint64_t v) {
    const uint64_t* s = all(v, 7);
    uint64_t m = 0;
    for (int l = 0; l < LANES; ++l) m = s[l] > m ? s[l] : m;
    return m;
  }
  uint32_t wave_or_u32(uint32_t v) {
    const uint64_t* s = all(v, 8);
    uint32_t m = 0;
    for (int l = 0; l < LANES; ++l) m |= (uint32_t)s[l];
    return m;
  }
  uint32_t wave_sum_u32(uint32_t v) {
    const uint64_t* s = all(v, 9);
    uint32_t m = 0;
    for (int l = 0; l < LANES; ++l) m += (uint32_t)s[l];
    return m;
  }
  uint32_t wave_excl_sum_u32(uint32_t v) {
    const uint64_t* s = all(v, 10);
    uint32_t m = 0;
    for (int l = 0; l < lane; ++l) m += (uint32_t)s[l];
    return m;
  }
  void lds_max_u64(uint64_t* p, uint64_t v) { if (v > *p) *p = v; }
  void lds_or_u32(uint32_t* p, uint32_t v) { *p |= v; }
  unsigned long long clock() { return 0; }
  unsigned long long global_add(unsigned long long* p, unsigned long long v) {
    unsigned long long o = *p;
    *p = o + v;
    return o;
  }
};

}  // namespace wavesim
