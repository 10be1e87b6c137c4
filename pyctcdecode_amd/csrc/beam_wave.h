// beam_wave.h -- the prefix-beam recursion as ONE wavefront per utterance (reference:
// BeamSearchDecoderCTC._partial_decode_logits decoder.py:426-556, _finalize_beams :558-602,
// _get_lm_beams :346-424, _merge_beams :211-224, _prune_history :227-258).
//
// Same semantics as beam_core.h (which stays the general path: several language models, beam widths
// above 128, survivor bounds above SURV_CAP) but shaped for what a frame of the recursion really is on
// CDNA4: a few dozen live beams, a handful of surviving labels, ~100 candidates. 64 lanes cover that in
// one or two passes, so nothing here ever waits on an s_barrier:
//   * the beam table is an array of 112-byte records in LDS, read and written 16 bytes at a time
//     (ds_read_b128 / ds_write_b128); a lane gathers the record of whatever beam it needs;
//   * candidates (label s, beam i) are spread densely over the lanes, whole labels per pass;
//   * duplicates are found by a wave-wide hash match: one ds_max_u64 per candidate on
//     (57-bit key tag | 127 - candidate) and one read back -- the smallest candidate of the largest tag
//     owns a slot, losers move on to their next slot; the members of a group announce themselves to
//     their representative through one ds_or on a 128-bit mask, which gives the representative the
//     fold order (ascending beam rank, decoder.py:217-223) and the donor (last arrival) at once;
//   * threshold, top-B and the history prune are one counting sweep over 16-byte {score key, history
//     key} records that every lane reads at the same address (an LDS broadcast);
//   * the next table is built by gathering (pool entry, donor record, label record) per kept rank.
//   * runs of frames whose only survivor is the label every beam already ends in (most frames of a real
//     posterior) are consumed in place, 64 at a time, each checked to leave the order intact (label_run).
// Written for a LONE wave per SIMD: exec-mask branches and misplaced s_waitcnt are what cost here, so the
// per-candidate work is straight-line selects over safe LDS indices, and registers that receive global
// loads are laundered (ctx.opaque32) at their first intended use.
// Diagnostics, all off by default: CTC_WAVE_TRACE (device printf of pool / beam records), CTC_RUN_TRACE
// (host: which frames label_run consumed), CTC_SIM_DEBUG (index checks in the simulator), tick<>() phase
// timers (ctcdec_profile_phases).
// Everything a lane shares with another lane goes through LDS or a cross-lane instruction; `wsync()`
// marks the points where LDS traffic of different lanes meets (on the device a compiler fence -- one
// wave issues its LDS operations in order --, in the 64-fiber test simulator a rendezvous).
#pragma once
#include "beam_core.h"

namespace ctc {

typedef uint32_t u32x4 __attribute__((vector_size(16)));
typedef uint32_t u32x4a __attribute__((vector_size(16), may_alias));  // 16-byte view of a structure in global memory

CTC_HD uint64_t pack64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
CTC_HD double bits_f64(uint64_t u) {
  union { uint64_t u; double d; } c;
  c.u = u;
  return c.d;
}
CTC_HD uint64_t f64_bits(double d) {
  union { uint64_t u; double d; } c;
  c.d = d;
  return c.u;
}
CTC_HD float bits_f32(uint32_t u) {
  union { uint32_t u; float f; } c;
  c.u = u;
  return c.f;
}
CTC_HD uint32_t f32_bits(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  return c.u;
}
CTC_HD u32x4 mk4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  u32x4 r = {a, b, c, d};
  return r;
}
CTC_HD u32x4 mk4q(uint64_t a, uint64_t b) {
  return mk4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}
CTC_HD uint64_t q_lo(u32x4 v) { return pack64(v[0], v[1]); }
CTC_HD uint64_t q_hi(u32x4 v) { return pack64(v[2], v[3]); }

CTC_HD uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

// History of a text: its last n_hist word hashes (newest first) folded to 64 bits. The words are 61-bit
// polynomial hashes: xor under distinct rotations keeps equal tuples equal and makes unequal ones collide with
// probability ~2^-61 (no multiplies: this runs for every completed word).
CTC_HD uint64_t wave_hist_fold(const uint64_t* ring, uint32_t cnt) {
  uint64_t h = 0x9E3779B97F4A7C15ull * (uint64_t)(cnt + 1u);
CTC_UNROLL
  for (int k = 0; k < MAX_CTX; ++k)
    if ((uint32_t)k < cnt) h ^= rotl64(ring[k], 13 * k + 1);
  return h;
}

// bijective 64-bit finaliser (cheaper than mix64: one multiply)
CTC_HD uint64_t fin64(uint64_t x) {
  x ^= x >> 32;
  x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 29;
  return x;
}

// beam record: 7 chunks of 16 bytes
//   0: text_h, part_h          1: logit, meta1, meta2       2: c_text_h, lm_hw     3: c_lm_hw, pscore
//   4: hist_h, c_hist_h        5: text_node, comp_node, emit_node, word_id         6: pstart, pend, depth, -
constexpr int BREC = 7;
constexpr int WAVE_LAB = 64;        // survivors are staged in LDS (ids, modes, label constants) 64 at a time
constexpr int WAVE_SURV_CAP = 480;  // survivors per frame this kernel handles: arrival = s * N + beam has to fit 16 bits

template <int BW>
struct WaveShape {
  static constexpr int SLB = (BW + 63) / 64;      // beam slots per lane (lane = beam phases)
  static constexpr bool BIG = BW > 64;            // frames with more than 64 live beams can occur
  static constexpr int C = BIG ? 128 : 64;        // candidates that share one match table
  static constexpr int P = BW + C;                // pool capacity: the kept beam_width + one pass of new ones
  static constexpr int PE = (P + 63) / 64;        // pool entries per lane
  static constexpr int TS = 2 * C;                // match-table slots
};

struct WaveLds {
  LPtr<u32x4> beams;     // [BW * BREC]
  LPtr<u32x4> surv;      // [WAVE_LAB]       {id, mode word, lp lo, lp hi}
  LPtr<u32x4> lab;       // [WAVE_LAB * 3]   {h_raw, pow_raw} {h_clean, len_raw, len_clean} {flags, start_flags, start_word_id, hot}
  LPtr<double> c_logit;  // [C]
  LPtr<uint32_t> c_br;   // [C]   branch of the candidate (the donor's is what the new beam is built from) | representative << 8
  LPtr<uint64_t> table;  // [TS]
  LPtr<u32x4> gmask;     // [C]   members of the group a candidate represents
  LPtr<u32x4> rank_rec;  // [P]   {score key, history key}; aliases c_logit/table/gmask (used between passes only)
  // pool of merged, scored candidates: 3 chunks per entry
  //   0: score, history key     1: logit, new partial hash     2: arrival | new plen << 16, donor word, word id, meta2
  // donor word: beam index | label << 8 | label is blank << 29 | donor's branch << 30
  LPtr<u32x4> pool;      // [P * 3]
  LPtr<uint32_t> sel;    // [BW]
  LPtr<unsigned long long> prof;  // [W_PROF_N] phase tick accumulators (diagnostics)
  // scalar views of the beam records
  LPtr<uint64_t> b64;
  LPtr<double> bf64;
  LPtr<uint32_t> b32;
  LPtr<int32_t> bi32;
};

template <int BW>
CTC_HD size_t wave_lds_carve(WaveLds& o, lds_bytes_t base) {
  typedef WaveShape<BW> S;
  lds_bytes_t p = base;
  o.beams = lds_take<u32x4>(p, 16 * BREC * BW);
  o.b64.p = (CTC_LDS uint64_t*)o.beams.p;
  o.bf64.p = (CTC_LDS double*)o.beams.p;
  o.b32.p = (CTC_LDS uint32_t*)o.beams.p;
  o.bi32.p = (CTC_LDS int32_t*)o.beams.p;
  o.surv = lds_take<u32x4>(p, 16 * WAVE_LAB);
  o.lab = lds_take<u32x4>(p, 16 * 3 * WAVE_LAB);
  o.pool = lds_take<u32x4>(p, 48 * S::P);
  o.sel = lds_take<uint32_t>(p, 4 * BW);
  o.prof = lds_take<unsigned long long>(p, 8 * 16);
  lds_bytes_t shared0 = p;
  o.c_logit = lds_take<double>(p, 8 * S::C);
  o.c_br = lds_take<uint32_t>(p, 4 * S::C);
  o.table = lds_take<uint64_t>(p, 8 * S::TS);
  o.gmask = lds_take<u32x4>(p, 16 * S::C);
  lds_bytes_t q = shared0;
  o.rank_rec = lds_take<u32x4>(q, 16 * S::P + 4 * S::P);  // + the compaction index list
  if (q > p) p = q;
  return (size_t)(p - base);
}
template <int BW>
CTC_HD size_t wave_lds_bytes() {
  WaveLds tmp;
  return wave_lds_carve<BW>(tmp, (lds_bytes_t) nullptr);
}

// may this decode run on the wave kernel?
CTC_HD bool wave_eligible(const DeviceTables& t, const DecodeParams& p) {
  return t.n_lms <= 1 && p.beam_width <= 128 && p.max_surv <= WAVE_SURV_CAP;
}
CTC_HD int wave_bucket(int beam_width) { return beam_width <= 64 ? 64 : beam_width <= 104 ? 104 : 128; }

constexpr int W_PROF_LOAD = 0, W_PROF_COMP = 1, W_PROF_GEN = 2, W_PROF_MATCH = 3, W_PROF_FOLD = 4, W_PROF_SCORE = 5,
              W_PROF_RANK = 6, W_PROF_BUILD = 7, W_PROF_FINAL = 8, W_PROF_COMPACT = 9, W_PROF_PUSH = 10, W_PROF_PFTOK = 11,
              W_PROF_GATHER = 12, W_PROF_FETCH = 13, W_PROF_BEGIN = 14, W_PROF_RUN = 15, W_PROF_N = 16;

template <class Ctx, int BW>
struct WaveDecoder {
  typedef WaveShape<BW> S;
  static constexpr int SLB = S::SLB;
  static constexpr int C = S::C;
  static constexpr int P = S::P;
  static constexpr int PE = S::PE;
  static constexpr int TS = S::TS;

  Ctx& ctx;
  WaveLds& L;
  const DeviceTables& tab;
  const DecodeParams& prm;
  const UttIO& io;
  const int lane;

  // wave-uniform state (every lane holds the same value)
  int N = 1;
  uint32_t pool_n = 0;
  uint64_t runmax = 0;   // ascending-sortable key of the best score pushed this frame
  uint64_t kth_key = 0;  // after a pool compaction: key a later candidate has to beat
  uint32_t text_next = 1, emit_next = 1, status = 0;
  uint32_t fflag = 0;    // force_next_break (decoder.py:442)
  uint32_t need = 0;     // some label of this frame closes open words
  // survivors of the NEXT frame, one per lane, fetched a frame ahead
  uint32_t pf_cnt = 0, pf_id = 0;
  double pf_lp = 0.0;
  bool pf_live = false;
  uint64_t pt_h_raw = 0, pt_pow_raw = 0, pt_h_clean = 0;
  uint32_t pt_len_raw = 0, pt_len_clean = 0, pt_flags = TK_BLANK, pt_start_flags = 0, pt_start_word_id = 0;
  uint64_t pt_hot_raw = 0;  // the label's TokHot entry as loaded (min_len, complete): folded only where it is consumed
  // completions in flight (lane = beam): source node fetched at the start of the frame, n-gram probes issued
  // before the candidates are generated, everything resolved after the match
  // What the completion of beam (slot, lane) needs, fetched as raw 16-byte chunks and decoded where it is
  // used (a loaded register that is repacked or copied right away has to be waited for right away).
  // Only beam slot 0 (beams 0..63) defers its completion across the candidate generation; slot 1 (more than
  // 64 live beams: rare) completes on the spot, which keeps ~70 registers per lane out of the long live ranges.
  bool cp_cand = false, cp_todo = false;
  uint32_t cp_idx = 0, cp_wid = 0, cp_m2 = 0;
  uint64_t cp_part = 0, cp_text = 0, cp_ring3 = 0;
  double cp_raw = 0.0;
  u32x4 cp_c2, cp_c3, cp_c4, cp_c5, cp_c6;  // TextNode chunks 2..6 of the source node
  // n-gram probes of slot 0 in flight: unigram entry, first table entry of the orders 2..6, their keys
  uint64_t cp_u = 0;
  u32x4 cp_e[MAX_CTX];
  uint64_t cp_k[MAX_CTX];
  int cp_max_n = 1;
  bool comp_pending = false;
  // emission nodes of the beams built last frame: their stores are issued in the NEXT frame right after its
  // last load has been consumed -- on gfx9 stores and loads share vmcnt, so a load consumed while a store is
  // still pending waits for the store's acknowledgement as well (measured: ~5 us per frame when the stores sat
  // in front of the next loads)
  bool em_pending = false;
  bool em_has[SLB];
  uint32_t em_idx[SLB];
  u32x4 em_node[SLB];
  bool tok_pending = false;  // the label constants of the next frame's survivors still have to be fetched
  bool run_ok = false;       // the beam table is the output of a full frame of this launch (label_run's precondition)
  unsigned long long t_last = 0;  // (diagnostics: phase ticks are accumulated in global memory, not registers)

  CTC_HD WaveDecoder(Ctx& c, WaveLds& l, const DeviceTables& t, const DecodeParams& p, const UttIO& i)
      : ctx(c), L(l), tab(t), prm(p), io(i), lane(c.lane) {}

  template <int PHASE>
  CTC_HD void tick() {
    if (io.prof && lane == 0) {
      unsigned long long now = ctx.clock();
      L.prof[PHASE] += now - t_last;
      t_last = now;
    }
  }

  // ---- small helpers -------------------------------------------------------------------------
  CTC_HD static uint64_t asc_key(double s) {
    if (s == 0.0) s = 0.0;
    const uint64_t u = f64_bits(s);
    return (u >> 63) ? ~u : (u | (1ull << 63));
  }
  CTC_HD static double key_to_score(uint64_t u) {
    const uint64_t bits = (u >> 63) ? (u & ~(1ull << 63)) : ~u;
    return bits_f64(bits);
  }
  CTC_HD uint32_t prefix_cnt(uint64_t m) const { return (uint32_t)ctx.popc64(m & ((1ull << lane) - 1ull)); }

  // mode word of a survivor: branch mode | first non-repeating beam << 8 | label flags (TK_*) << 16
  CTC_HD static uint32_t branch_of(uint32_t mode_word, uint32_t c, uint32_t i, uint32_t last_char) {
    if ((mode_word & (TK_BLANK << 16)) || last_char == c) return 0;  // keep prefix (blank / repeat)   decoder.py:452
    const uint32_t mode = mode_word & 0xFFu;
    if (mode == MODE_ALL_B) return BR_BOUNDARY;
    if (mode == MODE_FIRST_B) return i == ((mode_word >> 8) & 0xFFu) ? BR_BOUNDARY : BR_APPEND;
    if (mode == MODE_C) return BR_SPACE;
    return BR_APPEND;
  }

  // ---- survivor prefetch (one frame ahead) --------------------------------------------------
  CTC_HD void prefetch(int t) {
    pf_live = t < io.T;
    tok_pending = pf_live;
    if (!pf_live) return;
    // (every lane loads the same count; it is only read -- and made a scalar -- once it has landed)
    pf_cnt = io.surv_cnt[t];
    if (lane < prm.max_surv) {
      pf_id = io.surv_id[(size_t)t * prm.max_surv + lane];
      pf_lp = io.surv_lp[(size_t)t * prm.max_surv + lane];
    }
  }
  CTC_HD void prefetch_tok() {  // second stage, issued once the ids above have landed
    tok_pending = false;
    if (!pf_live) return;
    // (opaque: keeps the compiler from evaluating anything that depends on the prefetched registers earlier
    // than here -- it once hoisted `lane < pf_cnt` out of the pass loop to right behind the loads, which put
    // a full memory round trip at the start of every frame)
    pf_cnt = ctx.opaque32(pf_cnt);
    pf_id = ctx.opaque32(pf_id);
    if ((uint32_t)lane >= pf_cnt) return;
    const TokInfo& g = tab.tok[pf_id];
    pt_h_raw = g.h_raw;
    pt_pow_raw = g.pow_raw;
    pt_h_clean = g.h_clean;
    pt_len_raw = g.len_raw;
    pt_len_clean = g.len_clean;
    pt_flags = g.flags;
    pt_start_flags = g.start_flags;
    pt_start_word_id = g.start_word_id;
    // (raw: any arithmetic on the loaded words here would wait for them on the spot)
    pt_hot_raw = tab.tok_hot ? *(const uint64_t*)&tab.tok_hot[pf_id] : 0ull;
  }

  // Branch modes of up to 64 labels, one per lane (flags TK_BLANK for a lane without a label). For BPE
  // vocabularies the force_next_break flag threads through the labels in iteration order; each label acts on
  // it as identity / clear / set, so the flag a label sees is that of the last non-identity label before it.
  // first/any: index of the first beam that does not repeat the label (N: none).
  CTC_HD uint32_t mode_block(uint32_t fl, uint32_t c, uint32_t lc0, uint32_t f1) {
    const bool blank = (fl & TK_BLANK) != 0;
    if (!tab.is_bpe) {
      const uint32_t mode = blank ? MODE_A : ((fl & TK_SPACE) ? MODE_C : MODE_D);
      if (ctx.ballot(!blank && mode == MODE_C) != 0ull) need = 1u;
      return mode | ((uint32_t)(N & 0xFF) << 8) | (fl << 16);
    }
    uint32_t first = (uint32_t)N;
    if (!blank) first = (c != lc0) ? 0u : f1;
    const bool any = !blank && first < (uint32_t)N;
    const bool lead = (fl & TK_LEAD) != 0, trail = (fl & TK_TRAIL) != 0;
    const bool sets = any && lead && trail;
    const bool clears = any && !trail;
    const uint64_t m_one = ctx.ballot(sets);
    const uint64_t m_set = m_one | ctx.ballot(clears);
    const uint64_t prior = m_set & ((1ull << lane) - 1ull);
    uint32_t f_in = fflag;
    if (prior) f_in = (uint32_t)((m_one >> (63 - ctx.clz64(prior))) & 1ull);
    uint32_t mode = MODE_D;
    if (blank) mode = MODE_A;
    else if (lead) mode = MODE_ALL_B;
    else if (f_in && any) mode = trail ? MODE_ALL_B : MODE_FIRST_B;
    if (ctx.ballot(any && mode != MODE_D) != 0ull) need = 1u;
    if (m_set) fflag = (uint32_t)((m_one >> (63 - ctx.clz64(m_set))) & 1ull);
    return mode | ((first & 0xFFu) << 8) | (fl << 16);  // (first <= 128; only compared with beam indices < N)
  }

  // ---- completion of a beam's open word: the (text (+) partial) prefix ------------------------
  // One TextNode per completed prefix (the reference's memo entry, decoder.py:387-396); lane = beam.
  struct CompSrc {  // what a completion needs from the beam's record and its text node
    uint32_t wid, m2;
    uint64_t part, text, ring3;
    double raw;
    u32x4 c2, c3, c4, c5, c6;
  };
  CTC_HD void comp_fetch(CompSrc& q, int i, u32x4 k1, u32x4 k5) {
    q.wid = k5[3];
    q.m2 = k1[3];
    const u32x4 k0 = L.beams[i * BREC];
    q.text = q_lo(k0);
    q.part = q_hi(k0);
    // TextNode as 16-byte chunks: 0 text_h, raw_lm | 2 hw_cnt, ring_cnt, state.len, words[0] | 3 words[1..4]
    // | 4 backoff[0..3] | 5 backoff[4], pad, ring[0] | 6 ring[1], ring[2] | 7 ring[3], ring[4]
    const TextNode& sn = io.text_nodes[k5[0]];
    const u32x4a* src = (const u32x4a*)&sn;
    q.raw = sn.raw_lm;
    q.c2 = src[2];
    q.c3 = src[3];
    q.c4 = src[4];
    q.c5 = src[5];
    q.c6 = src[6];
    q.ring3 = sn.ring[3];
  }
  CTC_HD static LmState comp_state(const CompSrc& q) {
    LmState st;
    st.len = (int32_t)q.c2[2];
    st.words[0] = q.c2[3];
    st.words[1] = q.c3[0];
    st.words[2] = q.c3[1];
    st.words[3] = q.c3[2];
    st.words[4] = q.c3[3];
    st.backoff[0] = bits_f32(q.c4[0]);
    st.backoff[1] = bits_f32(q.c4[1]);
    st.backoff[2] = bits_f32(q.c4[2]);
    st.backoff[3] = bits_f32(q.c4[3]);
    st.backoff[4] = bits_f32(q.c5[0]);
    return st;
  }
  // fetch (beams 0..63): issued at the start of the frame, before it is known whether any label closes a word --
  // the loads are cheap and land while the branch modes are worked out. Returns the last label of this lane's
  // beam of slot 0 / 1 through lc[].
  CTC_HD void completions_fetch(uint32_t* lc) {
    cp_cand = false;
    cp_todo = false;
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      lc[j] = 0;
      if (j * 64 >= N) continue;
      const int i = j * 64 + lane;
      if (j == 0) {
        // Every lane loads (a lane without a beam, or whose beam needs no completion, reads the node of beam 0):
        // behind a divergent branch the compiler cannot count the loads in flight, and the first wait for any OLDER
        // load -- the prefetched label constants, a few lines on -- would turn into a wait for these as well.
        const int ii = i < N ? i : 0;
        const u32x4 k1 = L.beams[ii * BREC + 1], k5 = L.beams[ii * BREC + 5];
        lc[0] = i < N ? (k1[2] & 0xFFFFu) : 0u;
        cp_cand = i < N && (k1[2] >> 16) > 0 && k5[1] == 0;
        CompSrc q;
        comp_fetch(q, ii, k1, k5);
        cp_wid = q.wid;
        cp_m2 = q.m2;
        cp_text = q.text;
        cp_part = q.part;
        cp_ring3 = q.ring3;
        cp_raw = q.raw;
        cp_c2 = q.c2;
        cp_c3 = q.c3;
        cp_c4 = q.c4;
        cp_c5 = q.c5;
        cp_c6 = q.c6;
      } else if (i < N) {
        lc[j] = L.b32[i * 28 + 6] & 0xFFFFu;
      }
    }
  }
  CTC_HD CompSrc cp_src() const {
    CompSrc q;
    q.wid = cp_wid;
    q.m2 = cp_m2;
    q.text = cp_text;
    q.part = cp_part;
    q.ring3 = cp_ring3;
    q.raw = cp_raw;
    q.c2 = cp_c2;
    q.c3 = cp_c3;
    q.c4 = cp_c4;
    q.c5 = cp_c5;
    q.c6 = cp_c6;
    return q;
  }
  // node index + the completed text's hash (all the candidate keys need) for the beams of slot j that need it
  CTC_HD uint32_t comp_alloc(bool todo, int i, uint64_t text, uint64_t part) {
    const uint64_t m = ctx.ballot(todo);
    uint32_t idx = text_next + prefix_cnt(m);
    text_next += (uint32_t)ctx.popc64(m);
    if (todo) {
      if (idx + 1 > io.text_cap) {
        status |= ST_TEXT_OVERFLOW;  // (made wave-wide at the end of the frame)
        idx = io.text_cap - 1;
      }
      L.b64[i * 14 + 4] = text_push(text, part);  // c_text_h
      L.b32[i * 28 + 21] = idx;                    // comp_node
    }
    return idx;
  }
  // begin: beams 0..63 get their node index and issue their n-gram probes (resolved by completions_end after
  // the match); beams 64.. (more than 64 live beams: rare) are completed on the spot
  CTC_HD u32x4 opaque128(u32x4 v) { return mk4(ctx.opaque32(v[0]), ctx.opaque32(v[1]), ctx.opaque32(v[2]), ctx.opaque32(v[3])); }
  CTC_HD void completions_begin() {
    // The source node fetched at the start of the frame is first LOOKED AT here: laundering its registers keeps the
    // compiler from hoisting pieces of the work below (the history fold's count, the state length) up to right behind
    // the loads, where their s_waitcnt vmcnt(0) put the fetch's whole round trip in front of the branch modes.
    cp_c2 = opaque128(cp_c2);
    cp_c3 = opaque128(cp_c3);
    cp_c4 = opaque128(cp_c4);
    cp_c5 = opaque128(cp_c5);
    cp_c6 = opaque128(cp_c6);
    cp_raw = bits_f64(pack64(ctx.opaque32((uint32_t)f64_bits(cp_raw)), ctx.opaque32((uint32_t)(f64_bits(cp_raw) >> 32))));
    cp_ring3 = pack64(ctx.opaque32((uint32_t)cp_ring3), ctx.opaque32((uint32_t)(cp_ring3 >> 32)));
    cp_todo = cp_cand;
    if (ctx.ballot(cp_todo) != 0ull) {
      comp_pending = true;
      cp_idx = comp_alloc(cp_todo, lane, cp_text, cp_part);
      if (cp_todo && tab.has_lm) {
        // the keys of all orders share one chain, newest word first (common.h); every first table entry is
        // loaded as one 16-byte tuple straight into the register quad it stays in until completions_end
        const LmState in = comp_state(cp_src());
        const uint32_t wid = cp_wid;
        const UnigramEntry* up = &tab.unigrams[wid];
        cp_u = *(const uint64_t*)up;
        const int in_len = in.len;
        const int max_n = !tab.ngrams ? 1 : ((int)tab.lm_order < in_len + 1 ? (int)tab.lm_order : in_len + 1);
        cp_max_n = max_n;
        uint64_t c = ngram_key_push(ngram_key_begin(), wid);
CTC_UNROLL
        for (int k = 0; k < MAX_CTX; ++k) {
          cp_k[k] = 0;
          cp_e[k] = mk4(0, 0, 0, 0);
          if (max_n >= k + 2) {
            c = ngram_key_push(c, in.words[k]);
            cp_k[k] = ngram_key_end(c, (uint32_t)(k + 2));
            cp_e[k] = *(const u32x4a*)&tab.ngrams[cp_k[k] & tab.ngram_mask];
          }
        }
      }
    }
    if (SLB > 1 && N > 64) {
      const int i = 64 + lane;
      bool todo = false;
      u32x4 k1 = mk4(0, 0, 0, 0), k5 = mk4(0, 0, 0, 0);
      if (i < N) {
        k1 = L.beams[i * BREC + 1];
        k5 = L.beams[i * BREC + 5];
        todo = (k1[2] >> 16) > 0 && k5[1] == 0;
      }
      if (ctx.ballot(todo) != 0ull) {
        CompSrc q;
        q.wid = q.m2 = 0;
        q.text = q.part = q.ring3 = 0;
        q.raw = 0.0;
        q.c2 = q.c3 = q.c4 = q.c5 = q.c6 = mk4(0, 0, 0, 0);
        if (todo) comp_fetch(q, i, k1, k5);
        const uint32_t idx = comp_alloc(todo, i, q.text, q.part);
        if (todo) completion_finish(q, i, idx, false);
      }
    }
  }
  CTC_HD uint64_t opaque64(uint64_t v) { return pack64(ctx.opaque32((uint32_t)v), ctx.opaque32((uint32_t)(v >> 32))); }
  CTC_HD static NgramEntry entry_of(u32x4 r) {
    NgramEntry e;
    e.key = q_lo(r);
    e.prob = bits_f32(r[2]);
    e.backoff = bits_f32(r[3]);
    return e;
  }
  // LM score of the closed word, hot-word count, history ring; the node and the beam's memo fields.
  // probed: the n-gram probes of this lane were issued by completions_begin.
  CTC_HD void completion_finish(const CompSrc& q, int i, uint32_t idx, bool probed) {
    const uint32_t m2 = q.m2;
    double raw = q.raw;
    const LmState in = comp_state(q);
    LmState out = in;
    if (tab.has_lm) {
      float base;
      if (probed) {
        LmProbe p;
        // the probe results are used HERE: this call sits in the pass loop, and everything computed from
        // these loop-invariant registers would otherwise be hoisted to right behind the loads (a memory
        // round trip in front of the candidate generation instead of hidden behind it)
        const uint64_t u = opaque64(cp_u);
        p.u.prob = bits_f32((uint32_t)u);
        p.u.backoff = bits_f32((uint32_t)(u >> 32));
        p.max_n = cp_max_n;
        u32x4 e[MAX_CTX];
CTC_UNROLL
        for (int k = 0; k < MAX_CTX; ++k)
          e[k] = mk4(ctx.opaque32(cp_e[k][0]), ctx.opaque32(cp_e[k][1]), ctx.opaque32(cp_e[k][2]), ctx.opaque32(cp_e[k][3]));
        p.k2 = cp_k[0]; p.k3 = cp_k[1]; p.k4 = cp_k[2]; p.k5 = cp_k[3]; p.k6 = cp_k[4];
        p.s2 = p.k2 & tab.ngram_mask; p.s3 = p.k3 & tab.ngram_mask; p.s4 = p.k4 & tab.ngram_mask;
        p.s5 = p.k5 & tab.ngram_mask; p.s6 = p.k6 & tab.ngram_mask;
        p.e2 = entry_of(e[0]); p.e3 = entry_of(e[1]); p.e4 = entry_of(e[2]); p.e5 = entry_of(e[3]); p.e6 = entry_of(e[4]);
        base = lm_probe_finish(tab, in, q.wid, p, &out);
      } else {
        base = lm_base_score(tab, in, q.wid, &out);
      }
      raw = raw + lm_word_score(tab, prm, base, m2, 0.0, false);
    }
    const uint64_t part_h = q.part;
    const uint32_t cnt = q.c2[0] + ((m2 & M2_HOT_COMPLETE) ? 1u : 0u);
    const double lmhw = raw + prm.hot_weight * (double)cnt;
    const uint32_t rc0 = q.c2[1];
    const uint32_t rc = rc0 + 1 > tab.n_hist ? tab.n_hist : rc0 + 1;
    // history ring, newest first: the closed word, then the source node's (its fifth entry always drops out)
    const uint64_t old0 = pack64(q.c5[2], q.c5[3]), old1 = q_lo(q.c6), old2 = q_hi(q.c6), old3 = q.ring3;
    uint64_t ring[MAX_CTX];
    ring[0] = part_h;
    ring[1] = 1u < rc ? old0 : 0ull;
    ring[2] = 2u < rc ? old1 : 0ull;
    ring[3] = 3u < rc ? old2 : 0ull;
    ring[4] = 4u < rc ? old3 : 0ull;
    const uint64_t hh = wave_hist_fold(ring, rc);
    const uint64_t th = text_push(q.text, part_h);
    u32x4a* dst = (u32x4a*)&io.text_nodes[idx];
    dst[0] = mk4q(th, f64_bits(raw));
    dst[1] = mk4q(f64_bits(lmhw), hh);
    dst[2] = mk4(cnt, rc, (uint32_t)out.len, out.words[0]);
    dst[3] = mk4(out.words[1], out.words[2], out.words[3], out.words[4]);
    dst[4] = mk4(f32_bits(out.backoff[0]), f32_bits(out.backoff[1]), f32_bits(out.backoff[2]), f32_bits(out.backoff[3]));
    dst[5] = mk4(f32_bits(out.backoff[4]), 0u, (uint32_t)ring[0], (uint32_t)(ring[0] >> 32));
    dst[6] = mk4q(ring[1], ring[2]);
    dst[7] = mk4q(ring[3], ring[4]);
    L.bf64[i * 14 + 6] = lmhw;   // c_lm_hw
    L.b64[i * 14 + 9] = hh;      // c_hist_h
  }
  CTC_HD void completions_end() {
    if (cp_todo) completion_finish(cp_src(), lane, cp_idx, true);
    cp_todo = false;
    comp_pending = false;
  }

  CTC_HD void flush_emits() {
    if (!em_pending) return;
CTC_UNROLL
    for (int j = 0; j < SLB; ++j)
      if (em_has[j]) *(u32x4a*)&io.emit_nodes[em_idx[j]] = em_node[j];
    em_pending = false;
  }

  // One entry per lane (`ok` lanes, mask pm; e = the lane's pool index): rank by (key asc = score desc, arrival asc),
  // history duplicates flagged; writes L.sel for the ranks below `want`. The caller has laid the passing entries'
  // (key, history key) out densely in L.rank_rec[0 .. np) (lane order) and fenced: every lane walks that list with
  // broadcast LDS reads (one ds_read_b128 + a handful of VALU per entry, the reads pipelined ahead of their use)
  // instead of 4 v_readlane + scalar bit scanning per entry.
  CTC_HD void rank_lanes(uint64_t key, uint64_t hk, bool ok, uint64_t pm, uint32_t e, bool with_hist, uint32_t want) {
    const uint32_t np = (uint32_t)ctx.popc64(pm);
    uint32_t rank = 0, dup = 0;
    // (the list is padded to a multiple of four with entries that beat nobody: key = ~0)
    if (with_hist) {
      for (uint32_t j = 0; j < np; j += 4u) {
        const u32x4 r0 = L.rank_rec[j], r1 = L.rank_rec[j + 1u], r2 = L.rank_rec[j + 2u], r3 = L.rank_rec[j + 3u];
        const bool b0 = q_lo(r0) < key, b1 = q_lo(r1) < key, b2 = q_lo(r2) < key, b3 = q_lo(r3) < key;
        rank += (b0 ? 1u : 0u) + (b1 ? 1u : 0u) + (b2 ? 1u : 0u) + (b3 ? 1u : 0u);
        // (bitwise, not short-circuit: the latter became a ladder of exec-mask branches)
        dup |= (uint32_t)((b0 & (q_hi(r0) == hk)) | (b1 & (q_hi(r1) == hk)) | (b2 & (q_hi(r2) == hk)) | (b3 & (q_hi(r3) == hk)));
      }
    } else {
      for (uint32_t j = 0; j < np; j += 4u) {
        const u32x4 r0 = L.rank_rec[j], r1 = L.rank_rec[j + 1u], r2 = L.rank_rec[j + 2u], r3 = L.rank_rec[j + 3u];
        rank += (q_lo(r0) < key ? 1u : 0u) + (q_lo(r1) < key ? 1u : 0u) + (q_lo(r2) < key ? 1u : 0u) + (q_lo(r3) < key ? 1u : 0u);
      }
    }
    // Without equal scores the ranks of the passing entries are a permutation of 0 .. np-1; an equal pair shares
    // a rank and makes their sum smaller -- one wave sum instead of an equality count per broadcast entry.
    if (ctx.wave_sum_u32(ok ? rank : 0u) != np * (np - 1u) / 2u) {
      // equal scores (rare): the earlier arrival ranks first (heapq.nlargest is stable)
      const uint32_t arr = ok ? (L.pool[e * 3 + 2][0] & 0xFFFFu) : 0u;
      for (uint64_t t = pm; t; t &= t - 1ull) {
        const int j = ctx.ctz64(t);
        const uint64_t x = ctx.bcast64(key, j), xh = ctx.bcast64(hk, j);
        const uint32_t xa = ctx.bcast32(arr, j);
        const bool before = ok && x == key && xa < arr;
        rank += before ? 1u : 0u;
        dup |= (before && xh == hk) ? 1u : 0u;
      }
    }
    if (ok && rank < want) L.sel[rank] = e | ((with_hist && dup) ? 0u : 0x80000000u);
  }

  // ---- pool ranking --------------------------------------------------------------------------
  // Ranks the pool entries with score >= thr by (score desc, arrival asc); L.sel[r] = pool index of rank r
  // (bit 31: kept by the history prune) for r < min(count, beam_width). Returns the count.
  CTC_HD uint32_t rank_pool(double thr, bool with_hist) {
    const uint32_t n = pool_n;
    const uint32_t want = (uint32_t)prm.beam_width;
    if (n <= 64u) {
      // the usual frame: one entry per lane, keys stay in registers; each passing entry is broadcast in turn
      // (readlane) and counted by the lanes it beats -- no LDS round trips at all
      const uint32_t e = (uint32_t)lane;
      const bool mine = e < n;
      u32x4 p0 = mk4(0, 0, 0, 0);
      if (mine) p0 = L.pool[e * 3];
      const double sc = bits_f64(q_lo(p0));
      const bool ok = mine && sc >= thr;
      const uint64_t pm = ctx.ballot(ok);
      const uint32_t np = (uint32_t)ctx.popc64(pm);
      const uint64_t key = ok ? score_sort_key(sc) : ~0ull, hk = with_hist ? q_hi(p0) : 0ull;
      if (ok) L.rank_rec[prefix_cnt(pm)] = mk4q(key, hk);
      if (lane < 4) L.rank_rec[np + (uint32_t)lane] = mk4q(~0ull, 0ull);
      ctx.wsync();
      rank_lanes(key, hk, ok, pm, e, with_hist, want);
      ctx.wsync();
      return np;
    }
    // Larger pools (heavy frames): the entries that pass the threshold are first compacted (usually far fewer
    // than the pool holds, and mostly <= 64 again), then ranked out of registers the same way, R compacted
    // entries per lane.
    LPtr<uint32_t> ridx;  // pool index of compacted entry q (behind the P rank records)
    ridx.p = (CTC_LDS uint32_t*)(L.rank_rec.p + P);
    uint32_t n_pass = 0;
CTC_UNROLL
    for (int k = 0; k < PE; ++k) {
      if ((uint32_t)(k * 64) >= n) continue;
      const uint32_t e = (uint32_t)(k * 64 + lane);
      u32x4 p0 = mk4(0, 0, 0, 0);
      if (e < n) p0 = L.pool[e * 3];
      const double sc = bits_f64(q_lo(p0));
      const bool ok = e < n && sc >= thr;
      const uint64_t bm = ctx.ballot(ok);
      if (ok) {
        const uint32_t q = n_pass + prefix_cnt(bm);
        L.rank_rec[q] = mk4q(score_sort_key(sc), with_hist ? q_hi(p0) : 0ull);
        ridx[q] = e;
      }
      n_pass += (uint32_t)ctx.popc64(bm);
    }
    if (n_pass <= 64u && lane < 4) L.rank_rec[n_pass + (uint32_t)lane] = mk4q(~0ull, 0ull);  // pads the list for rank_lanes (68 <= P)
    ctx.wsync();
    if (n_pass <= 64u) {  // (all but the heaviest frames) one compacted entry per lane: the same walk as the small pool
      const bool ok = (uint32_t)lane < n_pass;
      u32x4 r = mk4(~0u, ~0u, 0, 0);
      uint32_t e = 0;
      if (ok) {
        r = L.rank_rec[lane];
        e = ridx[lane];
      }
      rank_lanes(q_lo(r), q_hi(r), ok, n_pass >= 64u ? ~0ull : ((1ull << n_pass) - 1ull), e, with_hist, want);
      ctx.wsync();
      return n_pass;
    }
    const uint32_t R = (n_pass + 63u) >> 6;  // compacted entries per lane
    uint64_t key[PE], hk[PE];
    uint32_t rank[PE], same[PE], dup[PE];
CTC_UNROLL
    for (int k = 0; k < PE; ++k) {
      const uint32_t q = (uint32_t)(k * 64 + lane);
      key[k] = ~0ull;
      hk[k] = 0;
      rank[k] = same[k] = dup[k] = 0;
      if ((uint32_t)k < R && q < n_pass) {
        const u32x4 r = L.rank_rec[q];
        key[k] = q_lo(r);
        hk[k] = q_hi(r);
      }
    }
CTC_UNROLL
    for (int kk = 0; kk < PE; ++kk) {  // broadcast every compacted entry in turn
      if ((uint32_t)kk >= R) continue;
      const uint32_t cnt = n_pass - (uint32_t)(kk * 64) < 64u ? n_pass - (uint32_t)(kk * 64) : 64u;
      for (uint32_t j = 0; j < cnt; ++j) {
        const uint64_t x = ctx.bcast64(key[kk], (int)j), xh = ctx.bcast64(hk[kk], (int)j);
CTC_UNROLL
        for (int k = 0; k < PE; ++k) {
          if ((uint32_t)k < R) {
            const bool better = x < key[k];
            rank[k] += better ? 1u : 0u;
            same[k] += x == key[k] ? 1u : 0u;
            dup[k] |= (better && xh == hk[k]) ? 1u : 0u;
          }
        }
      }
    }
    // equal scores (rare): the earlier arrival ranks first (heapq.nlargest is stable)
    bool tie = false;
    uint32_t eidx[PE];
CTC_UNROLL
    for (int k = 0; k < PE; ++k) {
      const uint32_t q = (uint32_t)(k * 64 + lane);
      eidx[k] = ((uint32_t)k < R && q < n_pass) ? ridx[q] : 0u;
      tie = tie || (key[k] != ~0ull && same[k] > 1u);
    }
    if (ctx.ballot(tie) != 0ull) {
      uint32_t arr[PE];
CTC_UNROLL
      for (int k = 0; k < PE; ++k) arr[k] = key[k] != ~0ull ? (L.pool[eidx[k] * 3 + 2][0] & 0xFFFFu) : 0u;
      for (uint32_t j = 0; j < n_pass; ++j) {
        const u32x4 r = L.rank_rec[j];
        const uint64_t x = q_lo(r), xh = q_hi(r);
        const uint32_t xa = L.pool[ridx[j] * 3 + 2][0] & 0xFFFFu;
CTC_UNROLL
        for (int k = 0; k < PE; ++k) {
          const bool before = key[k] != ~0ull && x == key[k] && xa < arr[k];
          rank[k] += before ? 1u : 0u;
          dup[k] |= (before && xh == hk[k]) ? 1u : 0u;
        }
      }
    }
CTC_UNROLL
    for (int k = 0; k < PE; ++k)
      if (key[k] != ~0ull && rank[k] < want) L.sel[rank[k]] = eidx[k] | ((with_hist && dup[k]) ? 0u : 0x80000000u);
    ctx.wsync();
    return n_pass;
  }

  // keep only the best beam_width pool entries (exact: pruning is monotone)
  CTC_HD void compact_pool() {
    const double mx = key_to_score(runmax);
    uint32_t n = rank_pool(mx + prm.beam_prune_logp, false);
    if (n > (uint32_t)prm.beam_width) n = (uint32_t)prm.beam_width;
    u32x4 g0[SLB], g1[SLB], g2[SLB];
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const uint32_t r = (uint32_t)(j * 64 + lane);
      g0[j] = g1[j] = g2[j] = mk4(0, 0, 0, 0);
      if (r < n) {
        const uint32_t e = L.sel[r] & 0x7FFFFFFFu;
#ifdef CTC_SIM_DEBUG
        if (e >= (uint32_t)P) { fprintf(stderr, "compact: r=%u n=%u pool_n=%u sel=%x N=%d\n", r, n, pool_n, L.sel[r], N); abort(); }
#endif
        g0[j] = L.pool[e * 3];
        g1[j] = L.pool[e * 3 + 1];
        g2[j] = L.pool[e * 3 + 2];
      }
    }
    ctx.wsync();
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const uint32_t r = (uint32_t)(j * 64 + lane);
      if (r < n) {
        L.pool[r * 3] = g0[j];
        L.pool[r * 3 + 1] = g1[j];
        L.pool[r * 3 + 2] = g2[j];
      }
    }
    pool_n = n;
    // from now on only a candidate that beats the current beam_width-th best can still matter: later
    // candidates arrive later, so an equal score ranks behind the beam_width entries kept here
    if (n >= (uint32_t)prm.beam_width) {
      const uint32_t r = n - 1;
      uint64_t k = 0;
CTC_UNROLL
      for (int j = 0; j < SLB; ++j)
        if ((int)(r >> 6) == j) k = ctx.bcast64(asc_key(bits_f64(q_lo(g0[j]))), (int)(r & 63u));
      kth_key = k;
    }
    ctx.wsync();
  }

  // ---- candidates ------------------------------------------------------------------------------------
  // A pass takes whole labels: floor(64 * SLB / N) of them, candidate v = j * 64 + lane (up to SLB per lane);
  // with more than 64 live beams that is ONE label and v is the beam index.
  struct Cand {
    bool valid, is_rep, want_p, want_h;
    uint32_t v, bi, ls, ll, lid, mw, br, rep, pl0, m2_0, len_raw, tslot;
    uint64_t kp, ck, pp_key, ph_key;
    uint32_t pp_wid, pp_fl, ph_min, ph_cmp;
    double lg, lmhw;
  };

  // branch, merge key and summed logit of candidate (label l of the staged block = survivor s, beam i);
  // FULL: also the first probe of the prefix / hot-word table of an appended partial word
  template <bool FULL>
  CTC_HD void gen(Cand& c, bool valid, uint32_t v, uint32_t l, uint32_t s, uint32_t i) {
    // Straight-line: a lane without a candidate computes on label 0 / beam 0 (its fields are only looked at behind
    // `valid`), and every branch of the reference's if-ladder (decoder.py:452-534) is a select -- as divergent
    // branches this was 23 exec-mask regions and ~130 register moves for two candidates.
    const uint32_t ll = valid ? l : 0u, ii = valid ? i : 0u;
    const u32x4 sv = L.surv[ll];
    const u32x4 la = L.lab[ll * 3], lb = L.lab[ll * 3 + 1];
    const u32x4 k0 = L.beams[ii * BREC], k1 = L.beams[ii * BREC + 1], k2 = L.beams[ii * BREC + 2];
    c.valid = valid;
    c.is_rep = false;
    c.v = v;
    c.bi = i;
    c.ls = s;
    c.ll = l;
    c.rep = v;
    c.lid = sv[0];
    c.mw = sv[1];
    c.len_raw = lb[2];
    const uint32_t meta1 = k1[2];
    const uint32_t pl = meta1 >> 16;
    c.pl0 = pl;
    c.m2_0 = k1[3];
    // branch (branch_of, as selects)
    const uint32_t mode = sv[1] & 0xFFu;
    const bool keep = (sv[1] & (TK_BLANK << 16)) != 0u || (meta1 & 0xFFFFu) == sv[0];
    const bool first_b = ii == ((sv[1] >> 8) & 0xFFu);
    uint32_t b = BR_APPEND;
    b = mode == MODE_C ? (uint32_t)BR_SPACE : b;
    b = (mode == MODE_FIRST_B && first_b) ? (uint32_t)BR_BOUNDARY : b;
    b = mode == MODE_ALL_B ? (uint32_t)BR_BOUNDARY : b;
    b = keep ? 0u : b;
    c.br = b;
    const bool closes = b == BR_BOUNDARY || b == BR_SPACE;
    const bool app = b == BR_APPEND;
    // merge key parts: the text (the completed one when the open word closes: the rest of the completion may still
    // be in flight, c_text_h is there) and the new partial word
    const uint64_t kt = (closes && pl > 0) ? q_lo(k2) : q_lo(k0);
    const uint64_t p_app = str_concat(q_hi(k0), q_hi(la), q_lo(la));  // pow_raw, h_raw
    uint64_t p = q_hi(k0);
    p = app ? p_app : p;
    p = b == BR_BOUNDARY ? q_lo(lb) : p;  // h_clean
    p = b == BR_SPACE ? 0ull : p;
    c.kp = p;
    c.tslot = 0;
    c.want_p = c.want_h = false;
    c.pp_key = c.ph_key = 0;
    c.pp_wid = c.pp_fl = c.ph_min = c.ph_cmp = 0;
    if (FULL) {  // first probe of the prefix / hot-word table of an appended partial word
      const bool probe = valid && app && p != 0;
      c.tslot = (uint32_t)table_slot(p);
      c.want_p = probe && (k1[3] & PF_ON_TABLE) && tab.prefixes;
      c.want_h = probe && (k1[3] & M2_HOT_ON) && tab.hot;
      if (c.want_p) {
        const PrefixEntry& g = tab.prefixes[c.tslot & tab.prefix_mask];
        c.pp_key = g.key;
        c.pp_wid = g.word_id;
        c.pp_fl = g.flags;
      }
      if (c.want_h) {
        const HotEntry& g = tab.hot[c.tslot & tab.hot_mask];
        c.ph_key = g.key;
        c.ph_min = g.min_len;
        c.ph_cmp = g.complete;
      }
    }
    c.lmhw = bits_f64(q_hi(k2));
    c.ck = fin64(kt ^ rotl64(p, 17) ^ ((uint64_t)(l + 1u) << 56));
    c.lg = bits_f64(q_lo(k1)) + bits_f64(pack64(sv[2], sv[3]));
  }

  // wave-wide hash match on 64-bit keys (valid lanes only): rep = smallest candidate index with the same key.
  // The smallest candidate of the largest tag owns a slot; the others of its key join it, the rest move to their
  // next slot (other key bits, then linear). A candidates per lane (v = j * 64 + lane), all inserted in the same
  // round (the members of a key must see the same slot history); table of 128 * A slots, cleared by the caller.
  template <int A>
  CTC_HD void match(const bool* valid, const uint64_t* ck, uint32_t* rep) {
    constexpr uint32_t MASK = (uint32_t)(128 * A - 1);
    bool open[A];
    uint32_t slot[A];
CTC_UNROLL
    for (int j = 0; j < A; ++j) {
      open[j] = valid[j];
      rep[j] = (uint32_t)(j * 64 + lane);
      slot[j] = (uint32_t)(ck[j] >> 7) & MASK;
    }
    for (uint32_t round = 0;; ++round) {
      bool any_open = false;
CTC_UNROLL
      for (int j = 0; j < A; ++j) any_open = any_open || open[j];
      if (ctx.ballot(any_open) == 0ull) break;
CTC_UNROLL
      for (int j = 0; j < A; ++j) {
        const uint32_t v = (uint32_t)(j * 64 + lane);
        if (open[j]) ctx.lds_max_u64(&L.table[slot[j]], (ck[j] & ~127ull) | (uint64_t)(127u - v));
      }
      ctx.wsync();
      uint64_t got[A];
CTC_UNROLL
      for (int j = 0; j < A; ++j) got[j] = open[j] ? L.table[slot[j]] : 0ull;
CTC_UNROLL
      for (int j = 0; j < A; ++j) {
        if (open[j]) {
          if ((got[j] & ~127ull) == (ck[j] & ~127ull)) {
            rep[j] = 127u - (uint32_t)(got[j] & 127ull);
            open[j] = false;
          } else {
            slot[j] = round < 5u ? ((uint32_t)(ck[j] >> (15 + 8 * round)) & MASK) : ((slot[j] + 1u) & MASK);
          }
        }
      }
      ctx.wsync();
    }
  }

  // Everything this frame still reads from global memory is consumed here, before the frame's stores: the
  // table entries of the appended partials, the ids of the next frame's survivors (their label constants are
  // requested now) and the n-gram probes of the completions, whose nodes are then written together with the
  // emission nodes of the previous frame.
  struct TabView {
    bool on, hon;
    uint32_t pf, nw, hmin, hcomp;
  };
  CTC_HD TabView resolve_tables(const Cand& c) {
    TabView t;
    t.on = t.hon = false;
    t.pf = t.nw = t.hmin = t.hcomp = 0;
    if (c.is_rep && c.br == BR_APPEND) {
      const uint64_t key = c.kp;
      if (c.want_p) {
        uint64_t sp = c.tslot & tab.prefix_mask;
        uint64_t ek = c.pp_key;
        uint32_t nw = c.pp_wid, pf = c.pp_fl;
        while (ek != key && ek != 0) {
          sp = (sp + 1) & tab.prefix_mask;
          const PrefixEntry& g = tab.prefixes[sp];
          ek = g.key;
          nw = g.word_id;
          pf = g.flags;
        }
        t.on = ek == key;
        t.nw = nw;
        t.pf = pf;
      }
      if (c.want_h) {
        uint64_t sh = c.tslot & tab.hot_mask;
        uint64_t ek = c.ph_key;
        uint32_t hmin = c.ph_min, hcomp = c.ph_cmp;
        while (ek != key && ek != 0) {
          sh = (sh + 1) & tab.hot_mask;
          const HotEntry& g = tab.hot[sh];
          ek = g.key;
          hmin = g.min_len;
          hcomp = g.complete;
        }
        t.hon = ek == key;
        t.hmin = hmin;
        t.hcomp = hcomp;
      }
    }
    return t;
  }
  CTC_HD void frame_stores() {
    if (tok_pending) prefetch_tok();
    if (comp_pending) {
      completions_end();
      ctx.wsync();
    }
    flush_emits();
  }

  // partial_score (beam_core.h: language_model.py:141-150, 326-336; decoder.py:363-367, 397-409) as selects, for the
  // single-model kernel: same operations in the same order.
  CTC_HD double partial_score_sel(uint32_t pf_flags, uint32_t hot_min_len, uint32_t plen) const {
    const double pl = (double)plen;
    double s = 0.0;
    if (tab.has_lm) {  // (uniform)
      const bool on_trie = tab.has_trie && (pf_flags & PF_UNI_PREFIX);
      s = prm.unk * (on_trie ? 0.0 : 1.0);
      if (plen > 6) s = s * pl / 6.0;  // (the two fp64 divisions stay behind branches: ~15 instructions each)
    }
    if (hot_min_len > 0) s = prm.hot_weight * pl / (double)hot_min_len;
    return s;
  }

  // fold, score and push the representatives of one pass; A candidates per lane, W = 32-bit words of a member
  // mask in use
  template <int A>
  CTC_HD void tail(Cand* c, const TabView* t) {
    constexpr int W = 2 * A;
    // ---- fold the group's logits in ascending beam rank (decoder.py:217-223); donor = last arrival
    uint32_t imax[A], dbr[A];
    {
      u32x4 gm[A];
      bool more = false;
CTC_UNROLL
      for (int j = 0; j < A; ++j) {
        gm[j] = mk4(0, 0, 0, 0);
        imax[j] = c[j].bi;
        dbr[j] = c[j].br;
        if (c[j].is_rep) {
          gm[j] = L.gmask[c[j].v];
          uint32_t top = 0xFFFFFFFFu;
CTC_UNROLL
          for (int w = 0; w < W; ++w)
            if (gm[j][w]) top = (uint32_t)(w * 32 + 31 - ctx.clz32(gm[j][w]));
          if (top != 0xFFFFFFFFu) {
            imax[j] = c[j].bi + (top - c[j].v);  // members share the label: consecutive beam indices
            dbr[j] = L.c_br[top];
            more = true;
          }
        }
      }
      while (ctx.ballot(more) != 0ull) {
        more = false;
CTC_UNROLL
        for (int j = 0; j < A; ++j) {
          if (c[j].is_rep) {
            uint32_t mbit = 0xFFFFFFFFu;
CTC_UNROLL
            for (int w = W - 1; w >= 0; --w)
              if (gm[j][w]) mbit = (uint32_t)(w * 32 + ctx.ctz32(gm[j][w]));
            if (mbit != 0xFFFFFFFFu) {
CTC_UNROLL
              for (int w = 0; w < W; ++w)
                if ((int)(mbit >> 5) == w) gm[j][w] &= gm[j][w] - 1u;
              c[j].lg = lse2(c[j].lg, L.c_logit[mbit]);
              uint32_t left = 0;
CTC_UNROLL
              for (int w = 0; w < W; ++w) left |= gm[j][w];
              more = more || left != 0u;
            }
          }
        }
      }
    }
    tick<W_PROF_FOLD>();
    // ---- score the representatives (decoder.py:346-424)
    double score[A];
    uint64_t my_key[A];
    uint32_t v_pl[A], v_m2[A], v_wid[A];
    uint64_t pass_key = 0;
CTC_UNROLL
    for (int j = 0; j < A; ++j) {
      // Straight-line (selects, LDS reads at safe indices): as an if-ladder over the four branches this was ~30
      // exec-mask regions per slot. Lanes that represent nothing compute on beam 0 / label 0 and are masked at the end.
      const bool rep = c[j].is_rep;
      const uint32_t i = rep ? c[j].bi : 0u, ll = rep ? c[j].ll : 0u;
      const uint32_t b = c[j].br;
      const u32x4 lb = L.lab[ll * 3 + 1], lc = L.lab[ll * 3 + 2];
      const uint32_t st_wid = L.b32[i * 28 + 23];
      const double st_ps = L.bf64[i * 14 + 7], c_lmhw = L.bf64[i * 14 + 6];
      const bool is0 = b == 0, isB = b == BR_BOUNDARY, isA = b == BR_APPEND;  // else: space
      // boundary: a new word starts with the clean label (or, for a bare boundary mark, nothing yet)
      const uint32_t len_clean = lb[3];
      const uint32_t hminB = lc[3] & 0xFFFFu, hcompB = lc[3] >> 31;
      const bool bw = isB && len_clean > 0;
      const uint32_t m2B = len_clean > 0 ? ((lc[1] & (PF_PARTIAL_MASK | PF_ON_TABLE)) | (hminB ? M2_HOT_ON : 0u) |
                                            (hcompB ? M2_HOT_COMPLETE : 0u) | (hminB << 8))
                                         : EMPTY_PARTIAL_M2;
      // append: what the prefix / hot-word tables say about the longer partial word
      const uint32_t a_pf = t[j].on ? t[j].pf : 0u, a_hmin = t[j].hon ? t[j].hmin : 0u;
      const uint32_t m2A = (t[j].on ? (PF_ON_TABLE | (t[j].pf & PF_PARTIAL_MASK)) : 0u) | (t[j].hon ? M2_HOT_ON : 0u) |
                           ((t[j].hon && t[j].hcomp) ? M2_HOT_COMPLETE : 0u) | (a_hmin << 8);
      const uint32_t q_pl = is0 ? c[j].pl0 : (isB ? len_clean : (isA ? c[j].pl0 + c[j].len_raw : 0u));
      const uint32_t q_m2 = is0 ? c[j].m2_0 : (isB ? m2B : (isA ? m2A : EMPTY_PARTIAL_M2));
      const uint32_t q_wid = is0 ? st_wid : (isB ? (len_clean > 0 ? lc[2] : 0u) : (isA ? (t[j].on ? t[j].nw : 0u) : 0u));
      const double ps_new = partial_score_sel(isB ? lc[1] : a_pf, isB ? hminB : a_hmin, q_pl);
      const double q_ps = is0 ? st_ps : ((bw || isA) ? ps_new : 0.0);
      const double lmhw = (!is0 && !isA && c[j].pl0 > 0) ? c_lmhw : c[j].lmhw;  // boundary / space close the open word
      const double sc = total_score(tab, c[j].lg, lmhw, q_ps, q_pl);
      score[j] = rep ? sc : 0.0;
      my_key[j] = rep ? asc_key(sc) : 0ull;
      v_pl[j] = q_pl;
      v_m2[j] = q_m2;
      v_wid[j] = q_wid;
      if (my_key[j] > pass_key) pass_key = my_key[j];
    }
    pass_key = ctx.wave_max_u64(pass_key);
    if (pass_key > runmax) runmax = pass_key;
    const double thr = key_to_score(runmax) + prm.beam_prune_logp;
    tick<W_PROF_SCORE>();
    // ---- push what can still matter into the pool
CTC_UNROLL
    for (int j = 0; j < A; ++j) {
      const bool push = c[j].is_rep && score[j] >= thr && my_key[j] > kth_key;
      const uint64_t m = ctx.ballot(push);
      const uint32_t k = pool_n + prefix_cnt(m);
      pool_n += (uint32_t)ctx.popc64(m);
      if (push) {
        // (history, partial, last_char) folded to 64 bits (decoder.py:250-254): equality of the folds stands in
        // for equality of the triple (its members are 61/64-bit string hashes already)
        uint64_t hk = 0;
        if (prm.prune_history) {
          const bool closed = (c[j].br == BR_BOUNDARY || c[j].br == BR_SPACE) && c[j].pl0 > 0;
          const uint64_t hh = L.b64[c[j].bi * 14 + (closed ? 9 : 8)];  // c_hist_h : hist_h
          hk = fin64(hh ^ rotl64(c[j].kp, 19) ^ ((uint64_t)(c[j].lid + 1u) << 40));
        }
        L.pool[k * 3] = mk4q(f64_bits(score[j]), hk);
        L.pool[k * 3 + 1] = mk4q(f64_bits(c[j].lg), c[j].kp);
        const uint32_t blank = (c[j].mw >> 16) & TK_BLANK;
        L.pool[k * 3 + 2] = mk4((c[j].ls * (uint32_t)N + c[j].bi) | (v_pl[j] << 16),
                                imax[j] | (c[j].lid << 8) | (blank ? (1u << 29) : 0u) | (dbr[j] << 30), v_wid[j], v_m2[j]);
      }
    }
    ctx.wsync();
    tick<W_PROF_PUSH>();
  }

  // One pass: the candidates of the labels [l0, l1) of the staged block (survivors base + l), A candidates per
  // lane (v = j * 64 + lane; with more than 64 live beams a pass is one label and v the beam index)
  template <int A>
  CTC_HD void pass(uint32_t base, uint32_t l0, uint32_t l1) {
    const uint32_t Nn = (uint32_t)N;
    const uint32_t Q = (l1 - l0) * Nn;
    const uint32_t rcpN = Nn ? 65536u / Nn + 1u : 0u;  // v / N == (v * rcpN) >> 16 for v * N < 65536
    Cand c[A];
    bool valid[A];
    uint64_t ck[A];
    uint32_t rep[A];
CTC_UNROLL
    for (int j = 0; j < A; ++j) {
      const int v = j * 64 + lane;
      L.gmask[v] = mk4(0, 0, 0, 0);
      ((CTC_LDS u32x4*)L.table.p)[v] = mk4(0, 0, 0, 0);  // 128 slots per candidate slot
    }
CTC_UNROLL
    for (int j = 0; j < A; ++j) {
      const uint32_t v = (uint32_t)(j * 64 + lane);
      const uint32_t sl = (v * rcpN) >> 16;
      const uint32_t l = l0 + sl;
      gen<true>(c[j], v < Q, v, l, base + l, v - sl * Nn);
      valid[j] = c[j].valid;
      ck[j] = c[j].ck;
      if (c[j].valid) {
        L.c_logit[v] = c[j].lg;
        L.c_br[v] = c[j].br;
      }
    }
    ctx.wsync();
    tick<W_PROF_GEN>();
    match<A>(valid, ck, rep);
    // members announce themselves to their representative
CTC_UNROLL
    for (int j = 0; j < A; ++j) {
      const uint32_t v = (uint32_t)(j * 64 + lane);
      c[j].rep = rep[j];
      if (valid[j] && rep[j] != v) ctx.lds_or_u32(&((CTC_LDS uint32_t*)L.gmask.p)[rep[j] * 4 + (v >> 5)], 1u << (v & 31u));
      c[j].is_rep = valid[j] && rep[j] == v;
    }
    ctx.wsync();
    tick<W_PROF_MATCH>();
    TabView t[A];
CTC_UNROLL
    for (int j = 0; j < A; ++j) t[j] = resolve_tables(c[j]);
    frame_stores();
    tick<W_PROF_COMP>();
    tail<A>(c, t);
  }

  // ---- one frame ---------------------------------------------------------------------------------
  // ---- runs of single-label frames ----------------------------------------------------------------
  // A frame whose only survivor is the label every live beam already ends in -- a blank after blanks, a token
  // that is held -- extends every beam in place (decoder.py:452-471): logit += p and, for a token, the end frame of
  // the open word. Nothing merges (the merge and history keys are those of the previous frame, which left them
  // distinct), no word completes, and every score moves by the same p: the threshold prune and the stable sort keep
  // each beam where it is. "The same p" holds only up to fp rounding, so each frame's new scores are CHECKED to be
  // still sorted and above the threshold; the first frame where they are not, or that has another survivor set,
  // ends the run and goes through the full path. Real CTC posteriors are mostly such frames (the reference's
  // libri sample: 327 of 371 frames have one survivor). Returns the first frame not consumed (t: none was).
  CTC_HD int label_run(int t, uint32_t lab, bool lab_is_blank) {
    double lg[SLB], rest[SLB], psc[SLB];
    uint32_t pl[SLB];
    bool live[SLB];
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const int i = j * 64 + lane;
      live[j] = i < N;
      lg[j] = rest[j] = psc[j] = 0.0;
      pl[j] = 0;
      if (live[j]) {
        lg[j] = L.bf64[i * 14 + 2];
        rest[j] = L.bf64[i * 14 + 5];
        psc[j] = L.bf64[i * 14 + 7];
        pl[j] = L.b32[i * 28 + 6] >> 16;
      }
    }
    double p = bits_f64(ctx.bcast64(f64_bits(pf_lp), 0));
    double w_lp = 0.0;  // look-ahead window: lane k holds the survivor of frame tt + k
    uint32_t w_n = 0, w_pos = 0;
    int tt = t;
    for (;;) {
      double nl[SLB], sc[SLB];
CTC_UNROLL
      for (int j = 0; j < SLB; ++j) {
        nl[j] = lg[j] + p;
        sc[j] = total_score(tab, nl[j], rest[j], psc[j], pl[j]);
        if (live[j]) L.c_logit[j * 64 + lane] = sc[j];
      }
      ctx.wsync();
      const double thr = L.c_logit[0] + prm.beam_prune_logp;
      bool bad = false;
CTC_UNROLL
      for (int j = 0; j < SLB; ++j) {
        const int i = j * 64 + lane;
        if (live[j]) {
          if (i + 1 < N) bad = bad || !(sc[j] >= L.c_logit[i + 1]);
          bad = bad || !(sc[j] >= thr);
        }
      }
      const bool stop = ctx.ballot(bad) != 0ull;
      ctx.wsync();  // (the next frame's scores go to the same LDS words)
      if (stop) break;
CTC_UNROLL
      for (int j = 0; j < SLB; ++j) lg[j] = nl[j];
      ++tt;
      if (tt >= io.T) break;
      if (w_pos == w_n) {  // look ahead: up to 64 frames, one per lane
        const int f = tt + lane;
        bool q = false;
        if (f < io.T) {
          const uint32_t cnt = io.surv_cnt[f];
          const uint32_t id = io.surv_id[(size_t)f * prm.max_surv];
          w_lp = io.surv_lp[(size_t)f * prm.max_surv];
          q = cnt == 1u && id == lab;
        }
        const uint64_t qm = ctx.ballot(q);
        w_n = ~qm ? (uint32_t)ctx.ctz64(~qm) : 64u;
        w_pos = 0;
        if (w_n == 0) break;
      }
      p = bits_f64(ctx.bcast64(f64_bits(w_lp), (int)w_pos));
      ++w_pos;
    }
    if (tt == t) return t;
#ifdef CTC_RUN_TRACE
    if (lane == 0) fprintf(stderr, "label_run: frames %d..%d label %u N=%d\n", t, tt - 1, lab, N);
#endif
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const int i = j * 64 + lane;
      if (live[j]) {
        L.bf64[i * 14 + 2] = lg[j];
        if (!lab_is_blank) L.bi32[i * 28 + 25] = io.first_frame + tt;  // end frame of the open word: last held frame + 1
      }
    }
    ctx.wsync();
    prefetch(tt);
    prefetch_tok();
    tick<W_PROF_RUN>();
    return tt;
  }

  CTC_HD int step(int t) {
    const int frame = io.first_frame + t;
    const uint32_t ns = ctx.uni32(pf_cnt);
    if (run_ok && ns == 1u && N > 0 && !prm.no_label_runs) {
      const uint32_t lab = ctx.bcast32(pf_id, 0);
      bool same = true;
CTC_UNROLL
      for (int j = 0; j < SLB; ++j) {
        const int i = j * 64 + lane;
        if (i < N) same = same && (L.b32[i * 28 + 6] & 0xFFFFu) == lab;
      }
      if (ctx.ballot(!same) == 0ull) {
        const int t2 = label_run(t, lab, (ctx.bcast32(pt_flags, 0) & TK_BLANK) != 0u);
        if (t2 > t) return t2;
      }
    }
    pool_n = 0;
    runmax = asc_key(-INFINITY);
    kth_key = 0;
    need = 0;
    // this lane's beam(s): last label, and -- if its open word has no completion yet -- its text node
#ifdef CTC_WAVE_TRACE
    if (lane < N) {
      const u32x4 t0 = L.beams[lane * BREC], t1 = L.beams[lane * BREC + 1], t5 = L.beams[lane * BREC + 5];
      printf("TB f=%d N=%d i=%d text=%llx part=%llx logit=%.6f meta1=%x m2=%x tn=%u cn=%u wid=%u\n", frame, N, lane, (unsigned long long)q_lo(t0),
             (unsigned long long)q_hi(t0), bits_f64(q_lo(t1)), t1[2], t1[3], t5[0], t5[1], t5[3]);
    }
#endif
    // Block 0's labels go to LDS first: their registers were requested a frame ago, but the wait for them also covers
    // every younger load (loads and stores share vmcnt) -- it has to come BEFORE the source-node fetch below is issued,
    // or the frame starts with that fetch's full round trip.
    uint32_t id0 = 0, fl0 = TK_BLANK;
    double lp0 = 0.0;
    if ((uint32_t)lane < ns) {
      id0 = pf_id;
      lp0 = pf_lp;
      fl0 = pt_flags;
      L.lab[lane * 3] = mk4q(pt_h_raw, pt_pow_raw);
      L.lab[lane * 3 + 1] = mk4((uint32_t)pt_h_clean, (uint32_t)(pt_h_clean >> 32), pt_len_raw, pt_len_clean);
      const uint32_t pt_hot = ((uint32_t)pt_hot_raw & 0xFFFFu) | ((uint32_t)(pt_hot_raw >> 32) ? 0x80000000u : 0u);
      L.lab[lane * 3 + 2] = mk4(pt_flags, pt_start_flags, pt_start_word_id, pt_hot);
    }
    id0 = ctx.opaque32(id0);  // (pins the consumption above the fetch)
    ctx.wsync();
    uint32_t lc[SLB];
    completions_fetch(lc);
    tick<W_PROF_FETCH>();
    // is there a beam that does not end in beam 0's label, and which is the first?
    uint32_t lc0 = NO_CHAR, f1 = (uint32_t)N;
    {
      lc0 = ctx.bcast32(lc[0], 0);
CTC_UNROLL
      for (int j = SLB - 1; j >= 0; --j) {
        const int i = j * 64 + lane;
        const uint64_t m = ctx.ballot(i < N && lc[j] != lc0);
        if (m) f1 = (uint32_t)(j * 64 + ctx.ctz64(m));
      }
    }
    bool comp_begun = false;
    // survivors in blocks of 64 labels: ids, log-probs, branch modes and label constants -> LDS, then the
    // candidates of the block in passes of whole labels
    for (uint32_t base = 0; base < ns; base += 64u) {
      const uint32_t s = base + (uint32_t)lane;
      const bool mine = s < ns;
      uint32_t id = 0, fl = TK_BLANK;
      double lp = 0.0;
      if (base == 0) {
        if (mine) {
          id = id0;
          lp = lp0;
          fl = fl0;
        }
      } else {
        // (more than 64 survivors in one frame: rare; fetched on the spot)
        ctx.wsync();  // the previous block's passes are done with surv / lab
        if (mine) {
          id = io.surv_id[(size_t)t * prm.max_surv + s];
          lp = io.surv_lp[(size_t)t * prm.max_surv + s];
          const TokInfo& g = tab.tok[id];
          fl = g.flags;
          const uint32_t hot = tab.tok_hot ? ((tab.tok_hot[id].min_len & 0xFFFFu) | (tab.tok_hot[id].complete ? 0x80000000u : 0u)) : 0u;
          L.lab[lane * 3] = mk4q(g.h_raw, g.pow_raw);
          L.lab[lane * 3 + 1] = mk4((uint32_t)g.h_clean, (uint32_t)(g.h_clean >> 32), g.len_raw, g.len_clean);
          L.lab[lane * 3 + 2] = mk4(g.flags, g.start_flags, g.start_word_id, hot);
        }
      }
      const uint32_t mwd = mode_block(fl, id, lc0, f1);
      if (mine) L.surv[lane] = mk4(id, mwd, (uint32_t)f64_bits(lp), (uint32_t)(f64_bits(lp) >> 32));
      if (base == 0) tick<W_PROF_LOAD>();
      if (need && !comp_begun) {
        completions_begin();
        comp_begun = true;
      }
      ctx.wsync();
      if (base == 0) {
        prefetch(t + 1);  // lands while this frame's candidates are processed
        tick<W_PROF_BEGIN>();
      }
      const uint32_t nb = ns - base < 64u ? ns - base : 64u;
      // passes of whole labels (<= 64 * SLB candidates); before a pass that might not fit the pool, the pool is
      // compacted to its best beam_width entries
      // (a pass never brings more candidates than the pool can take right after a compaction: P - beam_width)
      uint32_t room = (uint32_t)P - (uint32_t)prm.beam_width;
      if (room > (uint32_t)(64 * SLB)) room = (uint32_t)(64 * SLB);
      uint32_t per = room / (uint32_t)(N > 0 ? N : 1);
      if (per == 0) per = 1;  // (N <= beam_width <= P - beam_width: one label always fits)
      for (uint32_t l0 = 0; l0 < nb; l0 += per) {
        const uint32_t l1 = l0 + per < nb ? l0 + per : nb;
        const uint32_t q = (l1 - l0) * (uint32_t)N;
        if (pool_n + q > (uint32_t)P) {
          compact_pool();
          tick<W_PROF_COMPACT>();
        }
        if (SLB == 1 || q <= 64u) pass<1>(base, l0, l1);
        else pass<SLB>(base, l0, l1);
      }
    }
    if (comp_pending || em_pending || tok_pending) frame_stores();  // (no pass ran: only without survivors or beams)
    tick<W_PROF_PFTOK>();
    const double thr = key_to_score(runmax) + prm.beam_prune_logp;
    const bool hist = prm.prune_history != 0;
#ifdef CTC_WAVE_TRACE
    if ((uint32_t)lane < pool_n) {
      const u32x4 t0 = L.pool[lane * 3], t2 = L.pool[lane * 3 + 2];
      printf("TR f=%d N=%d ns=%u pool=%u e=%d score=%.6f arr=%u don=%x wid=%u m2=%x\n", frame, N, ns, pool_n, lane, bits_f64(q_lo(t0)), t2[0], t2[1], t2[2], t2[3]);
    }
#endif
    uint32_t n = rank_pool(thr, hist);
    if (n > (uint32_t)prm.beam_width) n = (uint32_t)prm.beam_width;
    tick<W_PROF_RANK>();
    // nothing passed the threshold: only possible with non-finite scores (NaN rows) or a positive
    // beam_prune_logp; the reference then dies on max([]) (decoder.py:545) -- reported through the status
    if (n == 0) status |= ST_NO_BEAMS;
    if (SLB == 1 || n <= 64u) build<1>(frame, n);
    else build<SLB>(frame, n);
    run_ok = true;
    return t + 1;
  }

  // next beam table from the ranked pool (decoder.py:548-554); A = rank slots per lane in use
  template <int A>
  CTC_HD void build(int frame, uint32_t n) {
    // gather everything the new records need, then write them (one table, no double buffer)
    bool kept[A];
    uint32_t dst[A];
    u32x4 o0[A], o1[A], o2[A], o3[A], o4[A], o5[A], o6[A];
    uint32_t n_new = 0;
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) em_has[j] = false;
CTC_UNROLL
    for (int j = 0; j < A; ++j) {
      const uint32_t r = (uint32_t)(j * 64 + lane);
      kept[j] = false;
      dst[j] = 0;
      o0[j] = o1[j] = o2[j] = o3[j] = o4[j] = o5[j] = o6[j] = mk4(0, 0, 0, 0);
      if ((uint32_t)(j * 64) >= n) continue;
      uint32_t w = 0;
      if (r < n) w = L.sel[r];
      kept[j] = r < n && (w >> 31) != 0u;
      const uint64_t km = ctx.ballot(kept[j]);
      dst[j] = n_new + prefix_cnt(km);
      n_new += (uint32_t)ctx.popc64(km);
      // the payload is the donor's (the last-arriving duplicate, decoder.py:221-223), its branch included.
      // Emission nodes: one per kept beam whose label is not a blank / repeat.
      u32x4 e1 = mk4(0, 0, 0, 0), e2 = mk4(0, 0, 0, 0);
      if (kept[j]) {
        const uint32_t idx = w & 0x7FFFFFFFu;
        e1 = L.pool[idx * 3 + 1];
        e2 = L.pool[idx * 3 + 2];
      }
      const uint32_t don = e2[1];
      const uint32_t b = don >> 30, i = don & 0xFFu, c = (don >> 8) & 0xFFFFu;
      const uint64_t em = ctx.ballot(kept[j] && b != 0);
      uint32_t e = emit_next + prefix_cnt(em);
      emit_next += (uint32_t)ctx.popc64(em);
      if (em) em_pending = true;
      if (kept[j]) {
        const u32x4 k0 = L.beams[i * BREC], k1 = L.beams[i * BREC + 1], k2 = L.beams[i * BREC + 2];
        const u32x4 k3 = L.beams[i * BREC + 3], k4 = L.beams[i * BREC + 4], k5 = L.beams[i * BREC + 5];
        const u32x4 k6 = L.beams[i * BREC + 6];
        const uint32_t pl = k1[2] >> 16;
        uint64_t th = q_lo(k0), hh = q_lo(k4);
        const uint64_t ph = q_hi(e1);  // the new partial word's hash (unchanged for a blank / repeat)
        const uint64_t cth = q_lo(k2), chh = q_hi(k4);
        double lmhw = bits_f64(q_hi(k2));
        const double clm = bits_f64(q_lo(k3));
        uint32_t tnode = k5[0], cnode = k5[1], enode = k5[2];
        int32_t pst = (int32_t)k6[0], pen = (int32_t)k6[1];
        uint32_t depth = k6[2];
        const uint32_t m2 = e2[3], wid = e2[2];
        const uint32_t npl = e2[0] >> 16;
        if (b == 0) {
          if (!(don & (1u << 29))) pen = frame + 1;  // a repeated label extends the open word (decoder.py:453-461)
        } else {
          const int32_t wst = pst, wen = pen;
          if (b == BR_BOUNDARY || b == BR_SPACE) {
            if (pl > 0) {  // the open word is completed (decoder.py:483-495, 501-515)
              th = cth;
              hh = chh;
              lmhw = clm;
              tnode = cnode;
            }
            if (b == BR_BOUNDARY) {
              pst = frame;
              pen = frame + 1;
            } else {
              pst = -1;
              pen = -1;
            }
          } else {  // BR_APPEND (decoder.py:518-534)
            pst = pst < 0 ? frame : pst;
            pen = frame + 1;
          }
          cnode = 0;
          if (e >= io.emit_cap) {
            status |= ST_EMIT_OVERFLOW;
            e = io.emit_cap - 1;
          }
          // (stored by flush_emits() in the next frame, or before the final back-trace)
          em_has[j] = true;
          em_idx[j] = e;
          em_node[j] = mk4(enode, c | (b << 16), (uint32_t)wst, (uint32_t)wen);
          enode = e;
          depth += 1;
        }
        double ps = 0.0;
        if (npl > 0) ps = partial_score(tab, prm, m2 & PF_PARTIAL_MASK, (m2 & M2_HOT_ON) ? ((m2 >> 8) & 0xFFFFu) : 0u, npl);
        o0[j] = mk4q(th, ph);
        o1[j] = mk4(e1[0], e1[1], c | (npl << 16), m2);
        o2[j] = mk4q(cth, f64_bits(lmhw));
        o3[j] = mk4q(f64_bits(clm), f64_bits(ps));
        o4[j] = mk4q(hh, chh);
        o5[j] = mk4(tnode, cnode, enode, wid);
        o6[j] = mk4((uint32_t)pst, (uint32_t)pen, depth, 0u);
      }
    }
    ctx.wsync();
    tick<W_PROF_GATHER>();
CTC_UNROLL
    for (int j = 0; j < A; ++j) {
      if (kept[j]) {
        const uint32_t d = dst[j];
        L.beams[d * BREC] = o0[j];
        L.beams[d * BREC + 1] = o1[j];
        L.beams[d * BREC + 2] = o2[j];
        L.beams[d * BREC + 3] = o3[j];
        L.beams[d * BREC + 4] = o4[j];
        L.beams[d * BREC + 5] = o5[j];
        L.beams[d * BREC + 6] = o6[j];
      }
    }
    N = (int)n_new;
    // lanes may have raised status bits on their own
    status = ctx.wave_or_u32(status);
    ctx.wsync();
    tick<W_PROF_BUILD>();
  }

  // ---- init / import ---------------------------------------------------------------------------
  CTC_HD void write_beam(int i, uint64_t text_h, uint64_t part_h, double logit, uint32_t meta1, uint32_t meta2, double lm_hw,
                         double pscore, uint64_t hist_h, uint32_t text_node, uint32_t emit_node, uint32_t word_id,
                         int32_t pstart, int32_t pend, uint32_t depth) {
    const uint64_t lgb = f64_bits(logit);
    L.beams[i * BREC] = mk4q(text_h, part_h);
    L.beams[i * BREC + 1] = mk4((uint32_t)lgb, (uint32_t)(lgb >> 32), meta1, meta2);
    L.beams[i * BREC + 2] = mk4q(0, f64_bits(lm_hw));
    L.beams[i * BREC + 3] = mk4q(f64_bits(0.0), f64_bits(pscore));
    L.beams[i * BREC + 4] = mk4q(hist_h, 0);
    L.beams[i * BREC + 5] = mk4(text_node, 0u, emit_node, word_id);
    L.beams[i * BREC + 6] = mk4((uint32_t)pstart, (uint32_t)pend, depth, 0u);
  }

  CTC_HD void init() {
    text_next = 1;  // text node 0 = empty text
    emit_next = 1;  // emission node 0 = root
    status = 0;
    fflag = 0;
    N = 1;
    if (lane == 0) {
      TextNode root;
      root.text_h = 0;
      root.raw_lm = 0.0;
      root.lm_hw = 0.0;
      root.hw_cnt = 0;
      root.ring_cnt = 0;
      for (int k = 0; k < MAX_CTX; ++k) root.ring[k] = 0;
      root.hist_h = wave_hist_fold(root.ring, 0);
      root.pad0 = 0;
      LmState st;
      st.len = 0;
      for (int k = 0; k < MAX_CTX; ++k) {
        st.words[k] = 0;
        st.backoff[k] = 0.f;
      }
      if (io.start_state && io.start_state->len >= 0) st = *io.start_state;
      root.state = st;
      io.text_nodes[0] = root;
      EmitNode er;
      er.parent = 0;
      er.tok_branch = 0;
      er.wstart = -1;
      er.wend = -1;
      io.emit_nodes[0] = er;
      write_beam(0, 0, 0, 0.0, NO_CHAR, EMPTY_PARTIAL_M2, 0.0, 0.0, root.hist_h, 0, 0, 0, -1, -1, 0);
    }
    if (io.imports && io.n_import > 0) import_beams();
    ctx.mem_sync();
  }

  // streaming: rebuild the beam table from the caller's beams (their order is the rank order)
  CTC_HD void import_beams() {
    const int n = io.n_import;
    for (int i = lane; i < n; i += 64) {
      const ImportBeam& m = io.imports[i];
      const uint32_t node = 1u + (uint32_t)i;  // node 0 is the empty text
      TextNode& tn = io.text_nodes[node];
      tn.text_h = m.text_h;
      tn.raw_lm = m.raw_lm;
      const double lmhw = m.raw_lm + prm.hot_weight * (double)m.hw_cnt;
      tn.lm_hw = lmhw;
      const uint64_t hh = wave_hist_fold(m.ring, m.ring_cnt);
CTC_UNROLL
      for (int k = 0; k < MAX_CTX; ++k) tn.ring[k] = m.ring[k];
      tn.hist_h = hh;
      tn.hw_cnt = m.hw_cnt;
      tn.ring_cnt = m.ring_cnt;
      tn.pad0 = 0;
      tn.state.len = m.state.len;
CTC_UNROLL
      for (int k = 0; k < MAX_CTX; ++k) {
        tn.state.words[k] = m.state.words[k];
        tn.state.backoff[k] = m.state.backoff[k];
      }
      EmitNode en;
      en.parent = 0;
      en.tok_branch = (uint32_t)i | (BR_IMPORT << 16);
      en.wstart = -1;
      en.wend = -1;
      io.emit_nodes[1 + i] = en;
      const double ps = m.plen > 0 ? partial_score(tab, prm, m.m2 & PF_PARTIAL_MASK, (m.m2 & M2_HOT_ON) ? ((m.m2 >> 8) & 0xFFFFu) : 0u, m.plen) : 0.0;
      write_beam(i, m.text_h, m.part_h, m.logit_score, (m.last_char & 0xFFFFu) | (m.plen << 16),
                 m.plen > 0 ? m.m2 : EMPTY_PARTIAL_M2, lmhw, ps, hh, node, 1u + (uint32_t)i, m.word_id, m.pstart, m.pend, 1u);
    }
    text_next = 1u + (uint32_t)n;
    emit_next = 1u + (uint32_t)n;
    N = n;
  }

  // ---- finalisation: _finalize_beams(force_next_word, is_end) + output records (decoder.py:558-602,653-667)
  CTC_HD void finalise() {
    const bool fold = prm.fold != 0, eos = prm.eos != 0;
    pool_n = 0;
    runmax = asc_key(-INFINITY);
    kth_key = 0;
    flush_emits();
    ctx.mem_sync();
    if (fold) {
      uint32_t lc_unused[SLB];
      completions_fetch(lc_unused);
      completions_begin();
      completions_end();
    }
    ctx.mem_sync();
    const uint32_t Q = (uint32_t)N;  // one candidate per beam: N <= BW <= C
    bool valid[SLB], is_rep[SLB];
    uint32_t rep[SLB], donor[SLB];
    uint64_t ck[SLB];
    double lg[SLB];
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const int v = j * 64 + lane;
      L.gmask[v] = mk4(0, 0, 0, 0);
      ((CTC_LDS u32x4*)L.table.p)[v] = mk4(0, 0, 0, 0);
    }
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const uint32_t v = (uint32_t)(j * 64 + lane);
      valid[j] = v < Q;
      is_rep[j] = valid[j];
      rep[j] = v;
      donor[j] = v;
      ck[j] = 0;
      lg[j] = 0.0;
      if (valid[j]) {
        const uint32_t pl = L.b32[v * 28 + 6] >> 16;
        const uint64_t kt = pl > 0 ? L.b64[v * 14 + 4] : L.b64[v * 14];
        ck[j] = fin64(kt ^ 0x165667B19E3779F9ull);
        lg[j] = L.bf64[v * 14 + 2];
        L.c_logit[v] = lg[j];
      }
    }
    ctx.wsync();
    if (fold) {
      match<SLB>(valid, ck, rep);
CTC_UNROLL
      for (int j = 0; j < SLB; ++j) {
        const uint32_t v = (uint32_t)(j * 64 + lane);
        if (valid[j] && rep[j] != v) ctx.lds_or_u32(&((CTC_LDS uint32_t*)L.gmask.p)[rep[j] * 4 + (v >> 5)], 1u << (v & 31u));
        is_rep[j] = valid[j] && rep[j] == v;
      }
      ctx.wsync();
      // fold in ascending beam rank; scored through the donor's (text, next_word) split (decoder.py:387-395)
CTC_UNROLL
      for (int j = 0; j < SLB; ++j) {
        if (is_rep[j]) {
          const uint32_t v = (uint32_t)(j * 64 + lane);
          u32x4 gm = L.gmask[v];
CTC_UNROLL
          for (int w = 0; w < 4; ++w) {
            while (gm[w]) {
              const uint32_t mbit = (uint32_t)(w * 32 + ctx.ctz32(gm[w]));
              gm[w] &= gm[w] - 1u;
              lg[j] = lse2(lg[j], L.c_logit[mbit]);
              donor[j] = mbit;
            }
          }
        }
      }
    }
    // score
    double score[SLB];
    uint64_t pass_key = 0;
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      score[j] = 0.0;
      if (!is_rep[j]) continue;
      const uint32_t v = (uint32_t)(j * 64 + lane);
      if (fold) {
        const uint32_t d = donor[j];
        const uint32_t m2 = L.b32[d * 28 + 7];
        const uint32_t pl = L.b32[d * 28 + 6] >> 16;
        double lmhw;
        if (eos) {
          const TextNode& src = io.text_nodes[L.b32[d * 28 + 20]];
          const uint32_t cnt = src.hw_cnt + ((pl > 0 && (m2 & M2_HOT_COMPLETE)) ? 1u : 0u);
          if (tab.has_lm) {
            LmState st, end;
            st.len = src.state.len;
CTC_UNROLL
            for (int k = 0; k < MAX_CTX; ++k) {
              st.words[k] = src.state.words[k];
              st.backoff[k] = src.state.backoff[k];
            }
            const uint32_t wid = pl > 0 ? L.b32[d * 28 + 23] : 0u;
            const uint32_t wfl = pl > 0 ? m2 : 0u;
            const float base_s = lm_base_score(tab, st, wid, &end);
            double end_score = 0.0;
            if (prm.score_boundary) {
              LmState tmp;
              end_score = (double)lm_base_score(tab, end, tab.eos_id, &tmp);
            }
            const double raw = src.raw_lm + lm_word_score(tab, prm, base_s, wfl, end_score, true);
            lmhw = raw + prm.hot_weight * (double)cnt;
          } else {
            lmhw = prm.hot_weight * (double)cnt;
          }
        } else {
          lmhw = pl > 0 ? L.bf64[d * 14 + 6] : L.bf64[d * 14 + 5];  // memo entry (text (+) word, False)
        }
        score[j] = tab.has_lm ? lg[j] + lmhw : lg[j] + lmhw + 0.0;
      } else {
        score[j] = total_score(tab, lg[j], L.bf64[v * 14 + 5], L.bf64[v * 14 + 7], L.b32[v * 28 + 6] >> 16);
      }
      const uint64_t k = asc_key(score[j]);
      if (k > pass_key) pass_key = k;
    }
    runmax = ctx.wave_max_u64(pass_key);
    if (runmax < asc_key(-INFINITY)) runmax = asc_key(-INFINITY);
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      if ((uint32_t)(j * 64) >= Q) continue;
      const uint64_t m = ctx.ballot(is_rep[j]);
      const uint32_t k = pool_n + prefix_cnt(m);
      pool_n += (uint32_t)ctx.popc64(m);
      if (is_rep[j]) {
        L.pool[k * 3] = mk4q(f64_bits(score[j]), 0);
        L.pool[k * 3 + 1] = mk4q(f64_bits(lg[j]), 0);
        L.pool[k * 3 + 2] = mk4((uint32_t)(j * 64 + lane), donor[j], 0u, 0u);  // arrival = beam rank, donor beam
      }
    }
    ctx.wsync();
    uint32_t n = rank_pool(key_to_score(runmax) + prm.beam_prune_logp, false);
    if (n > (uint32_t)prm.beam_width) n = (uint32_t)prm.beam_width;
    if (n == 0) status |= ST_NO_BEAMS;
    uint32_t n_out = n;
    if (prm.n_best > 0 && n_out > (uint32_t)prm.n_best) n_out = (uint32_t)prm.n_best;
    // output records + back-trace of each returned beam's emission chain
    uint32_t len[SLB], off[SLB];
    uint32_t total = 0;
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const uint32_t r = (uint32_t)(j * 64 + lane);
      len[j] = 0;
      off[j] = 0;
      if (r < n_out) {
        const uint32_t idx = L.sel[r] & 0x7FFFFFFFu;
        const uint32_t d = L.pool[idx * 3 + 2][1];
        len[j] = L.b32[d * 28 + 26] + ((fold && (L.b32[d * 28 + 6] >> 16) > 0) ? 1u : 0u);
      }
      off[j] = total + ctx.wave_excl_sum_u32(len[j]);
      total += ctx.wave_sum_u32(len[j]);
    }
    unsigned long long base = 0;
    if (lane == 0) {
      base = ctx.global_add(io.tok_pool_head, (unsigned long long)total);
      if (base + total > io.tok_pool_cap) {
        status |= ST_TOK_OVERFLOW;
        base = 0;
      }
    }
    base = ctx.bcast64(base, 0);
    status = ctx.wave_or_u32(status);
    const bool tok_ok = !(status & ST_TOK_OVERFLOW);
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const uint32_t r = (uint32_t)(j * 64 + lane);
      if (r >= n_out) continue;
      const uint32_t idx = L.sel[r] & 0x7FFFFFFFu;
      const u32x4 e0 = L.pool[idx * 3], e1 = L.pool[idx * 3 + 1];
      const uint32_t d = L.pool[idx * 3 + 2][1];
      OutBeam& ob = io.out[r];
      ob.logit_score = bits_f64(q_lo(e1));
      ob.lm_score = bits_f64(q_lo(e0));
      const uint32_t meta1 = L.b32[d * 28 + 6];
      const uint32_t pl = meta1 >> 16;
      const bool closes = fold && pl > 0;
      const uint32_t o = (uint32_t)(base + off[j]);
      ob.tok_off = o;
      ob.tok_cnt = tok_ok ? len[j] : 0;
      ob.pad[0] = 0;
      ob.pad[1] = 0;
      ob.last_char = fold ? NO_CHAR : (meta1 & 0xFFFFu);
      ob.pstart = fold ? -1 : L.bi32[d * 28 + 24];
      ob.pend = fold ? -1 : L.bi32[d * 28 + 25];
      // the text's memo entry: raw LM sum and the state after its last word
      const TextNode& node = io.text_nodes[closes ? L.b32[d * 28 + 21] : L.b32[d * 28 + 20]];
      ob.raw_lm = node.raw_lm;
      if (!tab.has_lm) {
        ob.state.len = -1;
CTC_UNROLL
        for (int k = 0; k < MAX_CTX; ++k) {
          ob.state.words[k] = 0;
          ob.state.backoff[k] = 0.f;
        }
      } else if (eos) {
        // last_lm_state: state after the last word, before </s> (language_model.py:357); an empty
        // last word is still scored as a word (decoder.py:387-395)
        const TextNode& src = io.text_nodes[L.b32[d * 28 + 20]];
        LmState st;
        st.len = src.state.len;
CTC_UNROLL
        for (int k = 0; k < MAX_CTX; ++k) {
          st.words[k] = src.state.words[k];
          st.backoff[k] = src.state.backoff[k];
        }
        lm_base_score(tab, st, pl > 0 ? L.b32[d * 28 + 23] : 0u, &ob.state);
      } else {
        ob.state.len = node.state.len;
CTC_UNROLL
        for (int k = 0; k < MAX_CTX; ++k) {
          ob.state.words[k] = node.state.words[k];
          ob.state.backoff[k] = node.state.backoff[k];
        }
      }
      if (tok_ok) {
        uint32_t pos = o + len[j];
        if (closes) {
          EmitNode fin;
          fin.parent = 0;
          fin.tok_branch = BR_FINAL << 16;
          fin.wstart = L.bi32[d * 28 + 24];
          fin.wend = L.bi32[d * 28 + 25];
          io.tok_pool[--pos] = fin;
        }
        uint32_t e = L.b32[d * 28 + 22];
        while (e != 0 && pos > o) {
          const EmitNode en = io.emit_nodes[e];
          io.tok_pool[--pos] = en;
          e = en.parent;
        }
      }
    }
    if (lane == 0) {
      *io.n_out = n_out;
      *io.status = status;
    }
  }

  CTC_HD void run() {
    init();
    if (io.prof && lane == 0) {
      for (int k = 0; k < 16; ++k) L.prof[k] = 0;
      t_last = ctx.clock();
    }
    prefetch(0);
    prefetch_tok();
    for (int t = 0; t < io.T;) t = step(t);
    finalise();
    tick<W_PROF_FINAL>();
    if (io.prof && lane == 0)
      for (int k = 0; k < W_PROF_N; ++k) io.prof[k] = L.prof[k];
  }
};

}  // namespace ctc
