// set_order_small.h -- set_order.h for at most SMALL_SET_MAX_KEYS ascending ids below 4095, in 48 table slots.
//
// The same CPython 3.10 algorithm (Objects/setobject.c, restated in set_order.h: cpython_set_order), with every table
// size it can reach for so few keys written out, so that one frame's tables fit 96 bytes and a wavefront can order 64
// frames at once, one frame per lane (frame_prune_fast in backend_hip.hip):
//   * set(ids): 8 slots; the 5th key grows it to 32 (used * 5 >= mask * 3 -> resize for used * 4); the next growth would
//     come with the 19th key;
//   * `| {argmax}` copies the left operand (set_merge): up to 4 keys the copy has the same 8 slots and is the table
//     itself; 5..7 keys are re-inserted, in slot order, into 16 slots (the smallest power of two above used * 2); 8..15
//     keys take 32 slots again: the table itself;
//   * before the argmax is added the copy is resized when (used + 1) * 5 >= mask * 3: only for 4 keys in 8 slots
//     (-> 16 slots); adding the argmax never triggers another growth.
// A slot holds  id | payload << 12  (id < 4095, payload < 16: the caller's index of that id), SMALL_SET_EMPTY when free; only the
// id takes part in hashing and comparison.
// `Tab` is anything with  uint16_t get(uint32_t slot)  and  void put(uint32_t slot, uint16_t v)  over 48 slots:
// X = slots [0, 16), Y = slots [16, 48).
// Checked against set_order.h and against real Python sets by tests/test_set_order_small.py.
#pragma once
#include <stdint.h>

#include "common.h"

namespace ctc {

constexpr uint32_t SMALL_SET_MAX_KEYS = 15;
constexpr uint32_t SMALL_SET_SLOTS = 48;
constexpr uint16_t SMALL_SET_EMPTY = 0xFFFFu;
constexpr uint32_t SMALL_SET_ARGMAX = 15;  // payload of an argmax that is not among the ids (indices stop at 14)
constexpr uint32_t SMALL_SET_ID_BITS = 12;  // ids < 4095 (id 4095 with the argmax payload would read as SMALL_SET_EMPTY)
constexpr uint32_t SMALL_SET_ID_MASK = (1u << SMALL_SET_ID_BITS) - 1u;
constexpr uint32_t SMALL_SET_MAX_ID = SMALL_SET_ID_MASK - 1u;
static_assert(SMALL_SET_MAX_KEYS <= SMALL_SET_ARGMAX, "an index must not read as the argmax marker");

struct SmallSet {
  uint32_t base, mask, used;
};

template <class Tab>
CTC_HD void small_set_clear(Tab& tab, uint32_t base, uint32_t size) {
  for (uint32_t k = 0; k < size; ++k) tab.put(base + k, SMALL_SET_EMPTY);
}

// set_insert_clean: the key is known to be absent and no slot was ever deleted
template <class Tab>
CTC_HD void small_set_insert_clean(Tab& tab, const SmallSet& s, uint16_t entry) {
  const uint32_t key = entry & SMALL_SET_ID_MASK, mask = s.mask;
  uint32_t perturb = key, i = key & mask;
  for (;;) {
    if (tab.get(s.base + i) == SMALL_SET_EMPTY) {
      tab.put(s.base + i, entry);
      return;
    }
    if (i + 9 <= mask) {
      for (uint32_t j = 1; j <= 9; ++j)
        if (tab.get(s.base + i + j) == SMALL_SET_EMPTY) {
          tab.put(s.base + i + j, entry);
          return;
        }
    }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
  }
}

// set_add_entry without its resize (the callers know when one is due): false when the id is already a member
template <class Tab>
CTC_HD bool small_set_add(Tab& tab, const SmallSet& s, uint16_t entry) {
  const uint32_t key = entry & SMALL_SET_ID_MASK, mask = s.mask;
  uint32_t perturb = key, i = key & mask;
  for (;;) {
    uint32_t probes = (i + 9 <= mask) ? 9u : 0u;
    uint32_t e = i;
    for (;;) {
      const uint16_t cur = tab.get(s.base + e);
      if (cur == SMALL_SET_EMPTY) {
        tab.put(s.base + e, entry);
        return true;
      }
      if ((cur & SMALL_SET_ID_MASK) == key) return false;
      ++e;
      if (probes == 0) break;
      --probes;
    }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
  }
}

// every occupied slot of `from`, in slot order, into the cleared table `to`
template <class Tab>
CTC_HD void small_set_rebuild(Tab& tab, const SmallSet& from, SmallSet& to, uint32_t to_base, uint32_t to_size) {
  small_set_clear(tab, to_base, to_size);
  to.base = to_base;
  to.mask = to_size - 1;
  to.used = from.used;
  for (uint32_t k = 0; k <= from.mask; ++k) {
    const uint16_t v = tab.get(from.base + k);
    if (v != SMALL_SET_EMPTY) small_set_insert_clean(tab, to, v);
  }
}

// set(ids[0..n)) | {argmax} for n <= SMALL_SET_MAX_KEYS distinct ascending ids: the table that holds the result
// (iterate its slots base .. base + mask in order); slot payloads are the indices 0..n-1, SMALL_SET_ARGMAX for an
// argmax outside the ids. `id_at(k)` -> id k.
template <class Tab, class IdAt>
CTC_HD SmallSet small_set_order(Tab& tab, uint32_t n, IdAt id_at, uint32_t argmax) {
  SmallSet s{0, 7, 0};
  small_set_clear(tab, 0, 8);
  for (uint32_t k = 0; k < n; ++k) {
    small_set_add(tab, s, (uint16_t)(id_at(k) | (k << SMALL_SET_ID_BITS)));
    s.used += 1;
    if (s.used * 5 >= s.mask * 3) {  // 8 slots, 5th key (32 slots: not before the 19th)
      SmallSet grown;
      small_set_rebuild(tab, s, grown, 16, 32);
      s = grown;
    }
  }
  // the copy made by `|`
  if (s.used >= 5 && s.used <= 7) {
    SmallSet copy;
    small_set_rebuild(tab, s, copy, 0, 16);
    s = copy;
  } else if (s.used == 4) {  // copied as it is, then resized for the key to come
    SmallSet grown;
    small_set_rebuild(tab, s, grown, 16, 16);
    s = grown;
  }
  if (small_set_add(tab, s, (uint16_t)(argmax | (SMALL_SET_ARGMAX << SMALL_SET_ID_BITS)))) s.used += 1;
  return s;
}

}  // namespace ctc
