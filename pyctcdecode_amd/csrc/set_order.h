// set_order.h -- iteration order of CPython's  set(ascending ids) | {argmax}  (decoder.py:445-447).
// The reference iterates the surviving labels of a frame in hash-table slot order of a CPython
// set; that order decides tie-breaks and the BPE force_next_break leak, so the
// frame-prune kernel emits survivors in exactly that order.  This is Objects/setobject.c of
// CPython 3.10 (set_add_entry / set_insert_clean / set_table_resize / set_merge) for small
// non-negative int keys (hash(k) == k), restated over uint16 slot tables.
#pragma once
#include "common.h"

namespace ctc {

constexpr uint16_t SET_EMPTY = 0xFFFFu;

struct SetTab {
  uint16_t* slots;
  uint32_t mask;
  uint32_t used;
};

CTC_HD void set_clear(SetTab& t, uint32_t size) {
  t.mask = size - 1;
  t.used = 0;
  for (uint32_t k = 0; k < size; ++k) t.slots[k] = SET_EMPTY;
}

CTC_HD void set_insert_clean(SetTab& t, uint32_t key) {
  uint32_t mask = t.mask;
  uint32_t perturb = key;
  uint32_t i = key & mask;
  for (;;) {
    if (t.slots[i] == SET_EMPTY) {
      t.slots[i] = (uint16_t)key;
      return;
    }
    if (i + 9 <= mask) {
      for (uint32_t j = 1; j <= 9; ++j)
        if (t.slots[i + j] == SET_EMPTY) {
          t.slots[i + j] = (uint16_t)key;
          return;
        }
    }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
  }
}

// Rebuild `t` at the size CPython picks for `minused`, re-inserting the old slots in slot order.
// `scratch` must hold the old table (mask+1 entries).
CTC_HD void set_resize(SetTab& t, uint32_t minused, uint16_t* scratch) {
  uint32_t newsize = 8;
  while (newsize <= minused) newsize <<= 1;
  uint32_t oldsize = t.mask + 1;
  for (uint32_t k = 0; k < oldsize; ++k) scratch[k] = t.slots[k];
  uint32_t used = t.used;
  set_clear(t, newsize);
  t.used = used;
  for (uint32_t k = 0; k < oldsize; ++k)
    if (scratch[k] != SET_EMPTY) set_insert_clean(t, scratch[k]);
}

CTC_HD void set_add(SetTab& t, uint32_t key, uint16_t* scratch) {
  uint32_t mask = t.mask;
  uint32_t perturb = key;
  uint32_t i = key & mask;
  for (;;) {
    uint32_t probes = (i + 9 <= mask) ? 9u : 0u;
    uint32_t e = i;
    for (;;) {
      uint16_t cur = t.slots[e];
      if (cur == SET_EMPTY) {
        t.slots[e] = (uint16_t)key;
        t.used += 1;
        if (t.used * 5 >= mask * 3) set_resize(t, t.used > 50000 ? t.used * 2 : t.used * 4, scratch);
        return;
      }
      if (cur == key) return;
      ++e;
      if (probes == 0) break;
      --probes;
    }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
  }
}

// Largest table CPython can reach while holding up to n keys (host sizing helper).
inline uint32_t set_table_cap(uint32_t n) {
  uint32_t cap = 8, size = 8;
  for (uint32_t used = 1; used <= n + 1; ++used) {
    if (used * 5 >= (size - 1) * 3) {
      uint32_t minused = used * 4, ns = 8;
      while (ns <= minused) ns <<= 1;
      size = ns;
    }
    if (size > cap) cap = size;
  }
  // The copy made by `|` is sized in one step for used*2 (set_merge) or (used+1)*2 (the resize before the
  // argmax is added): the smallest power of two above that. This can exceed the incremental sizes above
  // (16 keys live in a 32-slot table, their union copy takes 64 slots), so it bounds the capacity too.
  uint32_t merged = 8;
  while (merged <= (n + 1) * 2) merged <<= 1;
  return merged > cap ? merged : cap;
}

// asc: ascending ids (n of them). tabA/tabR/scratch: `cap` uint16 each. out: n+1 entries.
// Returns the number of ids written to out (n, or n+1 when argmax was not among them).
CTC_HD uint32_t cpython_set_order(const uint16_t* asc, uint32_t n, uint32_t argmax, uint16_t* tabA,
                                  uint16_t* tabR, uint16_t* scratch, uint16_t* out) {
  SetTab a{tabA, 0, 0}, r{tabR, 0, 0};
  set_clear(a, 8);
  for (uint32_t k = 0; k < n; ++k) set_add(a, asc[k], scratch);
  set_clear(r, 8);
  if (a.used) {
    if (a.used * 5 >= r.mask * 3) {  // set_merge: one big resize up front
      uint32_t newsize = 8;
      while (newsize <= a.used * 2) newsize <<= 1;
      set_clear(r, newsize);
    }
    if (r.mask == a.mask) {
      for (uint32_t k = 0; k <= a.mask; ++k) r.slots[k] = a.slots[k];
    } else {
      for (uint32_t k = 0; k <= a.mask; ++k)
        if (a.slots[k] != SET_EMPTY) set_insert_clean(r, a.slots[k]);
    }
    r.used = a.used;
  }
  if ((r.used + 1) * 5 >= r.mask * 3) set_resize(r, (r.used + 1) * 2, scratch);
  set_add(r, argmax, scratch);
  uint32_t m = 0;
  for (uint32_t k = 0; k <= r.mask; ++k)
    if (r.slots[k] != SET_EMPTY) out[m++] = r.slots[k];
  return m;
}

}  // namespace ctc
