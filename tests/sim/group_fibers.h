// group_fibers.h -- TEST INFRASTRUCTURE ONLY.  Runs the workgroup kernel (pyctcdecode_amd/csrc/beam_core.h: BeamDecoder) on the
// CPU with as many cooperative fibers as the device launch has threads (256 or 512: 4 or 8 "waves" of 64), instead of the one
// sequential thread of SeqCtx. What that checks and SeqCtx cannot:
//   * every code path that depends on the thread count (chunks of at most one candidate per thread, the tiled pair loops,
//     the rows-by-threads splits of the selection) runs as on the device;
//   * barriers: sync() / sync_mem() are rendezvous of ALL fibers, wave-level operations (ballot, wave_max_u64) of the 64
//     fibers of one wave. Between two rendezvous a fiber runs alone, and the WAVE that arrived last runs first afterwards
//     (the scheduler walks the waves downwards; the lanes of a wave upwards, see run()), so a phase that reads what another
//     wave wrote without a barrier in between sees stale or too-new values and fails the parity tests; fibers waiting at different barriers, or a fiber that
//     ends while others wait, abort the run with a message.
// LDS atomics need no care: a fiber is never preempted between rendezvous.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <functional>
#include <vector>

#include "wave_fibers.h"  // wavesim_switch

using wavesim::wavesim_switch;

namespace groupsim {

constexpr int WAVE = 64;

struct Block;
static thread_local Block* g_block = nullptr;

struct Block {
  static constexpr size_t STACK = 192 * 1024;
  int nt = 0;
  std::vector<void*> sp;
  void* main_sp = nullptr;
  std::vector<char> stacks;
  std::vector<char> done;
  int current = -1;
  std::function<void(int)> body;
  struct Rdv {
    std::vector<uint64_t> slot[2];
    uint64_t gen = 0;
    int arrived = 0, tag0 = 0, size = 0;
  };
  Rdv all;                 // workgroup barrier
  std::vector<Rdv> waves;  // one per wave of 64
  uint64_t progress = 0;   // bumped by every completed rendezvous and every fiber that ends
  bool waves_up = false;

  static void trampoline() {
    Block* b = g_block;
    const int t = b->current;
    b->body(t);
    b->done[t] = 1;
    ++b->progress;
    wavesim_switch(&b->sp[t], b->main_sp);
    abort();  // a finished fiber is never resumed
  }

  void run(int threads, std::function<void(int)> fn) {
    nt = threads;
    body = std::move(fn);
    if (stacks.size() < STACK * (size_t)nt + 64) stacks.assign(STACK * (size_t)nt + 64, 0);
    sp.assign(nt, nullptr);
    done.assign(nt, 0);
    for (int t = 0; t < nt; ++t) {
      uintptr_t top = ((uintptr_t)stacks.data() + STACK * (size_t)(t + 1)) & ~(uintptr_t)15;
      void** s = (void**)top;
      *--s = nullptr;             // fake return address of the trampoline (keeps the ABI alignment)
      *--s = (void*)&trampoline;  // `ret` target of the first switch
      for (int k = 0; k < 6; ++k) *--s = nullptr;  // rbp rbx r12 r13 r14 r15
      sp[t] = (void*)s;
    }
    auto reset = [](Rdv& r, int n) {
      r.slot[0].assign(n, 0);
      r.slot[1].assign(n, 0);
      r.gen = 0;
      r.arrived = 0;
      r.size = n;
    };
    reset(all, nt);
    waves.assign((nt + WAVE - 1) / WAVE, Rdv());
    for (size_t w = 0; w < waves.size(); ++w) reset(waves[w], (int)std::min<size_t>(WAVE, nt - w * WAVE));
    Block* prev = g_block;
    g_block = this;
    progress = 0;
    {
      const char* e = getenv("CTCDEC_SIM_GROUP_ORDER");
      waves_up = e && e[0] == 'u';
    }
    for (;;) {
      int live = 0;
      const uint64_t before = progress;
      // Waves downwards (the wave that arrived last at a barrier runs first after it), the lanes of a wave upwards: on the
      // device the lanes of one wave execute in lockstep, so between two lanes of the SAME wave program order holds (lane 0's
      // earlier LDS store lands before lane 5's later one) and the kernel may rely on it; between waves nothing holds.
      // (CTCDEC_SIM_GROUP_ORDER=up walks the waves upwards instead: the suites run both ways, so that a missing barrier
      // shows whichever wave its reader is in.)
      const int n_waves = (nt + WAVE - 1) / WAVE;
      for (int k = 0; k < n_waves; ++k) {
        const int w = waves_up ? k : n_waves - 1 - k;
        for (int t = w * WAVE; t < nt && t < (w + 1) * WAVE; ++t) {
          if (done[t]) continue;
          ++live;
          current = t;
          wavesim_switch(&main_sp, sp[t]);
        }
      }
      if (!live) break;
      int still = 0;
      for (int t = 0; t < nt; ++t) still += done[t] ? 0 : 1;
      if (still && progress == before) {
        fprintf(stderr, "groupsim: deadlock -- %d of %d threads wait at the workgroup barrier (tag %d), %d have ended; wave waits:",
                all.arrived, nt, all.tag0, nt - still);
        for (size_t w = 0; w < waves.size(); ++w) fprintf(stderr, " [%d: %d at tag %d]", (int)w, waves[w].arrived, waves[w].tag0);
        fprintf(stderr, "\n");
        abort();
      }
    }
    g_block = prev;
  }

  void yield(int t) { wavesim_switch(&sp[t], main_sp); }

  const uint64_t* rendezvous(Rdv& r, int t, int idx, uint64_t v, int tag) {
    const uint64_t my_gen = r.gen;
    const int buf = (int)(my_gen & 1u);
    if (r.arrived == 0) r.tag0 = tag;
    else if (tag != r.tag0) {
      fprintf(stderr, "groupsim: threads diverged -- thread %d is at operation %d, others at %d\n", t, tag, r.tag0);
      abort();
    }
    r.slot[buf][idx] = v;
    if (++r.arrived == r.size) {
      r.arrived = 0;
      ++r.gen;
      ++progress;
      yield(t);  // the one that completes a rendezvous does not run ahead of its own wave's lower lanes (see run())
    } else {
      while (r.gen == my_gen) yield(t);
    }
    return r.slot[buf].data();
  }
};

// the execution context beam_core.h's BeamDecoder is written against
struct GroupFiberCtx {
  int tid, nt;
  Block* b;
  Block::Rdv& wave() { return b->waves[tid / WAVE]; }
  int lane() const { return tid % WAVE; }
  void sync() { b->rendezvous(b->all, tid, tid, 0, 1); }
  void sync_mem() { b->rendezvous(b->all, tid, tid, 0, 2); }
  uint32_t atomic_add(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
  void atomic_or(uint32_t* p, uint32_t v) { *p |= v; }
  void atomic_min(uint32_t* p, uint32_t v) { if (v < *p) *p = v; }
  void atomic_max(uint32_t* p, uint32_t v) { if (v > *p) *p = v; }
  void atomic_max64(uint64_t* p, uint64_t v) { if (v > *p) *p = v; }
  uint32_t atomic_cas(uint32_t* p, uint32_t cmp, uint32_t val) { uint32_t o = *p; if (o == cmp) *p = val; return o; }
  unsigned long long clock() { return 0; }
  void use(double) {}
  uint64_t wave_max_u64(uint64_t v) {
    Block::Rdv& r = wave();
    const uint64_t* s = b->rendezvous(r, tid, lane(), v, 3);
    uint64_t m = 0;
    for (int l = 0; l < r.size; ++l) m = s[l] > m ? s[l] : m;
    return m;
  }
  bool is_wave_leader() { return lane() == 0; }
  int wave_width() { return WAVE; }
  uint64_t ballot(bool p) {
    Block::Rdv& r = wave();
    const uint64_t* s = b->rendezvous(r, tid, lane(), p ? 1u : 0u, 4);
    uint64_t m = 0;
    for (int l = 0; l < r.size; ++l) m |= (s[l] & 1ull) << l;
    return m;
  }
  int clz64(uint64_t x) { return __builtin_clzll(x); }
  int popc64(uint64_t x) { return __builtin_popcountll(x); }
  unsigned long long global_add(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
};

}  // namespace groupsim
