// text_wave.h -- the text of a beam from its emission chain, by one 64-lane wavefront (decoder.py:653-667: the words of the
// beam joined by single spaces). Same result, byte for byte, as text_backwards (beam_core.h), which one thread computes
// with three dependent memory round trips per emission (node, label record, label bytes): ~250 emissions of a T=1000
// utterance were 0.4-0.6 ms of nothing but latency at the end of every decode_batch.
//
// Two phases, repeated until the root:
//   1. the chain itself. A child's index is always above its parent's (nodes are appended frame by frame) and the parent
//      usually sits a frame or two back: the wave loads a WINDOW of TEXT_WIN consecutive nodes ending at the current one
//      into LDS (coalesced, all loads in flight together) and lane 0 follows the parent links inside it -- LDS round
//      trips instead of memory ones -- listing the emissions it passes;
//   2. the listed emissions, 64 at a time, one per lane: label records and bytes are fetched side by side, the places of
//      the pieces come from a prefix sum over their lengths, and whether a separator is due before a piece (a word
//      boundary passed since the last bytes, and bytes to its right) from the ballots of "has bytes" and "is a boundary".
// Written against the same execution context as beam_wave.h (lane, wsync, ballot, clz64, wave_excl_sum_u32, wave_sum_u32):
// tests/sim/backend_sim.cpp runs it on 64 fibers next to text_backwards for every decode_batch of the CPU suite.
#pragma once
#include <stdint.h>

#include "beam_core.h"
#include "common.h"

namespace ctc {

#ifndef CTC_TEXT_WIN  // (the simulator builds with small odd sizes: windows and lists then end in every possible place)
#define CTC_TEXT_WIN 512
#define CTC_TEXT_LIST 512
#endif
constexpr uint32_t TEXT_WIN = CTC_TEXT_WIN;    // nodes per window (8 coalesced 16-byte loads per lane)
constexpr uint32_t TEXT_LIST = CTC_TEXT_LIST;  // emissions listed per window at most

struct TextLds {
  CTC_LDS uint32_t* win_parent;  // [TEXT_WIN]
  CTC_LDS uint32_t* win_tok;     // [TEXT_WIN]
  CTC_LDS uint32_t* list;        // [TEXT_LIST] tok_branch of the emissions passed, leaf side first
  CTC_LDS uint32_t* scal;        // [4]
};
constexpr size_t TEXT_LDS_BYTES = (2 * TEXT_WIN + TEXT_LIST + 4) * 4;
CTC_HD void text_lds_carve(TextLds& L, CTC_LDS char* base) {
  L.win_parent = (CTC_LDS uint32_t*)base;
  L.win_tok = L.win_parent + TEXT_WIN;
  L.list = L.win_tok + TEXT_WIN;
  L.scal = L.list + TEXT_LIST;
}

// All 64 lanes call it with the same arguments; returns (to all) where the text starts in scratch[.. cap).
// n_nodes: size of the emission arena = bound of the walk (see text_backwards).
template <class Ctx>
CTC_HD uint32_t wave_text_backwards(Ctx& ctx, const TextLds& L, const EmitNode* nodes, const DeviceTables& tab, uint32_t enode,
                                    uint8_t* scratch, uint32_t cap, uint32_t n_nodes) {
  const uint32_t lane = (uint32_t)ctx.lane;
  uint32_t cur = enode, steps = 0;
  int64_t pos = (int64_t)cap;             // may run below 0: bytes that would land there are dropped (a full scratch area)
  bool emitted = false, pending = false;  // (uniform) state of the backwards walk between the 64-blocks
  while (cur != 0 && cur < n_nodes && steps < n_nodes) {
    const uint32_t hi = cur, lo = hi >= TEXT_WIN - 1u ? hi - (TEXT_WIN - 1u) : 0u;
    for (uint32_t k = lane; k <= hi - lo; k += 64u) {
      const EmitNode en = nodes[lo + k];
      L.win_parent[k] = en.parent;
      L.win_tok[k] = en.tok_branch;
    }
    ctx.wsync();
    if (lane == 0) {
      uint32_t c = cur, n = 0, st = steps;
      while (c != 0 && c >= lo && c <= hi && st < n_nodes && n < TEXT_LIST) {
        L.list[n++] = L.win_tok[c - lo];
        c = L.win_parent[c - lo];
        ++st;
      }
      L.scal[0] = c;
      L.scal[1] = n;
      L.scal[2] = st;
    }
    ctx.wsync();
    cur = L.scal[0];
    const uint32_t n = L.scal[1];
    steps = L.scal[2];
    for (uint32_t base = 0; base < n; base += 64u) {  // (uniform trip count: ballots and scans inside)
      const uint32_t k = base + lane;
      const bool valid = k < n;
      const uint32_t tb = valid ? L.list[k] : 0u;
      const uint32_t br = tb >> 16, tok = tb & 0xFFFFu;
      uint32_t off = 0, len = 0;
      if (valid && br == BR_APPEND) {
        off = tab.tok_text[tok].raw_off;
        len = tab.tok_text[tok].raw_len;
      } else if (valid && br == BR_BOUNDARY) {
        off = tab.tok_text[tok].clean_off;
        len = tab.tok_text[tok].clean_len;
      }
      const bool has = len > 0;
      const bool bound = valid && (br == BR_BOUNDARY || br == BR_SPACE || br == BR_FINAL);
      const uint64_t lmask = ctx.ballot(has), bmask = ctx.ballot(bound);
      const uint64_t lower = (1ull << lane) - 1ull;
      // before this emission (going leaf to root): was a boundary passed since the last bytes? are there bytes at all?
      bool pend, emit;
      const uint64_t prev = lmask & lower;
      if (prev) {
        const uint32_t p = 63u - (uint32_t)ctx.clz64(prev);  // the emission with the last bytes: its own boundary counts
        pend = (bmask & lower & ~((1ull << p) - 1ull)) != 0;
        emit = true;
      } else {
        pend = pending || (bmask & lower) != 0;
        emit = emitted;
      }
      const uint32_t sep = (has && pend && emit) ? 1u : 0u;
      const uint32_t size = has ? len + sep : 0u;
      const uint32_t before = ctx.wave_excl_sum_u32(size);
      const uint32_t total = ctx.wave_sum_u32(size);
      if (has) {
        const int64_t end = pos - (int64_t)before;  // this piece: [end - size, end): the label's bytes, then the separator
        if (sep && end - 1 >= 0) scratch[end - 1] = (uint8_t)' ';
        const int64_t at = end - (int64_t)size;
        for (uint32_t j = 0; j < len; ++j)
          if (at + (int64_t)j >= 0) scratch[at + (int64_t)j] = tab.tok_bytes[off + j];
      }
      if (lmask) {
        const uint32_t p = 63u - (uint32_t)ctx.clz64(lmask);
        pending = (bmask >> p) != 0;
        emitted = true;
      } else {
        pending = pending || bmask != 0;
      }
      pos -= (int64_t)total;
    }
  }
  return pos > 0 ? (uint32_t)pos : 0u;
}

}  // namespace ctc
