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
// Everything a lane shares with another lane goes through LDS or a cross-lane instruction; `wsync()`
// marks the points where LDS traffic of different lanes meets (on the device a compiler fence -- one
// wave issues its LDS operations in order --, in the 64-fiber test simulator a rendezvous).
#pragma once
#include "beam_core.h"

namespace ctc {

typedef uint32_t u32x4 __attribute__((vector_size(16)));

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
CTC_HD u32x4 mk4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  u32x4 r = {a, b, c, d};
  return r;
}
CTC_HD u32x4 mk4q(uint64_t a, uint64_t b) {
  return mk4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}
CTC_HD uint64_t q_lo(u32x4 v) { return pack64(v[0], v[1]); }
CTC_HD uint64_t q_hi(u32x4 v) { return pack64(v[2], v[3]); }

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
constexpr int WAVE_LAB = 64;        // label records staged in LDS (survivors [0, 64) of the frame)
constexpr int WAVE_SURV_CAP = 160;  // survivors per frame this kernel handles (default token_min_logp: 150)

template <int BW>
struct WaveShape {
  static constexpr int SLB = (BW + 63) / 64;  // beam / candidate slots per lane
  static constexpr int C = 64 * SLB;          // candidates per pass
  static constexpr int P = BW + C;            // pool capacity: the kept beam_width + one pass of new ones
  static constexpr int PE = (P + 63) / 64;    // pool entries per lane
  static constexpr int TS = 2 * C;            // match-table slots
};

struct WaveLds {
  LPtr<u32x4> beams;     // [BW * BREC]
  LPtr<u32x4> surv;      // [WAVE_SURV_CAP]  {id, mode word, lp lo, lp hi}
  LPtr<u32x4> lab;       // [WAVE_LAB * 3]   {h_raw, pow_raw} {h_clean, len_raw, len_clean} {flags, start_flags, start_word_id, hot}
  LPtr<double> c_logit;  // [C]
  LPtr<uint64_t> table;  // [TS]
  LPtr<u32x4> gmask;     // [C]   members of the group a candidate represents
  LPtr<u32x4> rank_rec;  // [P]   {score key, history key}; aliases c_logit/table/gmask (used between passes only)
  LPtr<double> p_score, p_logit;
  LPtr<uint64_t> p_hk;
  LPtr<uint32_t> p_arr, p_don, p_wid, p_m2;
  LPtr<uint32_t> sel;    // [BW]
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
  o.surv = lds_take<u32x4>(p, 16 * WAVE_SURV_CAP);
  o.lab = lds_take<u32x4>(p, 16 * 3 * WAVE_LAB);
  o.p_score = lds_take<double>(p, 8 * S::P);
  o.p_logit = lds_take<double>(p, 8 * S::P);
  o.p_hk = lds_take<uint64_t>(p, 8 * S::P);
  o.p_arr = lds_take<uint32_t>(p, 4 * S::P);
  o.p_don = lds_take<uint32_t>(p, 4 * S::P);
  o.p_wid = lds_take<uint32_t>(p, 4 * S::P);
  o.p_m2 = lds_take<uint32_t>(p, 4 * S::P);
  o.sel = lds_take<uint32_t>(p, 4 * BW);
  lds_bytes_t shared0 = p;
  o.c_logit = lds_take<double>(p, 8 * S::C);
  o.table = lds_take<uint64_t>(p, 8 * S::TS);
  o.gmask = lds_take<u32x4>(p, 16 * S::C);
  lds_bytes_t q = shared0;
  o.rank_rec = lds_take<u32x4>(q, 16 * S::P);
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
CTC_HD int wave_bucket(int beam_width) { return beam_width <= 32 ? 32 : beam_width <= 64 ? 64 : 128; }

constexpr int W_PROF_LOAD = 0, W_PROF_COMP = 1, W_PROF_GEN = 2, W_PROF_MATCH = 3, W_PROF_FOLD = 4, W_PROF_SCORE = 5,
              W_PROF_RANK = 6, W_PROF_BUILD = 7, W_PROF_FINAL = 8, W_PROF_COMPACT = 9, W_PROF_N = 10;

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
  uint32_t pt_len_raw = 0, pt_len_clean = 0, pt_flags = TK_BLANK, pt_start_flags = 0, pt_start_word_id = 0, pt_hot = 0;
  unsigned long long t_last = 0;
  unsigned long long t_acc[W_PROF_N] = {};

  CTC_HD WaveDecoder(Ctx& c, WaveLds& l, const DeviceTables& t, const DecodeParams& p, const UttIO& i)
      : ctx(c), L(l), tab(t), prm(p), io(i), lane(c.lane) {}

  template <int PHASE>
  CTC_HD void tick() {
    if (io.prof && lane == 0) {
      unsigned long long now = ctx.clock();
      t_acc[PHASE] += now - t_last;
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

  struct Lab {  // label constants of one survivor
    uint64_t h_raw, pow_raw, h_clean;
    uint32_t len_raw, len_clean, flags, start_flags, start_word_id, hot_min, hot_complete;
  };
  // label constants of survivor s whose label id is c: LDS for the staged ones, L2 beyond
  CTC_HD Lab label_of(uint32_t s, uint32_t c) const {
    Lab r;
    if (s < (uint32_t)WAVE_LAB) {
      const u32x4 a = L.lab[s * 3], b = L.lab[s * 3 + 1], d = L.lab[s * 3 + 2];
      r.h_raw = q_lo(a);
      r.pow_raw = q_hi(a);
      r.h_clean = q_lo(b);
      r.len_raw = b[2];
      r.len_clean = b[3];
      r.flags = d[0];
      r.start_flags = d[1];
      r.start_word_id = d[2];
      r.hot_min = d[3] & 0xFFFFu;
      r.hot_complete = d[3] >> 31;
    } else {
      const TokInfo& g = tab.tok[c];
      r.h_raw = g.h_raw;
      r.pow_raw = g.pow_raw;
      r.h_clean = g.h_clean;
      r.len_raw = g.len_raw;
      r.len_clean = g.len_clean;
      r.flags = g.flags;
      r.start_flags = g.start_flags;
      r.start_word_id = g.start_word_id;
      r.hot_min = tab.tok_hot ? tab.tok_hot[c].min_len : 0u;
      r.hot_complete = tab.tok_hot ? tab.tok_hot[c].complete : 0u;
    }
    return r;
  }

  CTC_HD static uint32_t branch_of(uint32_t tflags, uint32_t mode_word, uint32_t c, uint32_t i, uint32_t last_char) {
    if ((tflags & TK_BLANK) || last_char == c) return 0;  // keep prefix (blank / repeat)   decoder.py:452
    const uint32_t mode = mode_word & 0xFFu;
    if (mode == MODE_ALL_B) return BR_BOUNDARY;
    if (mode == MODE_FIRST_B) return i == (mode_word >> 8) ? BR_BOUNDARY : BR_APPEND;
    if (mode == MODE_C) return BR_SPACE;
    return BR_APPEND;
  }

  // ---- survivor prefetch (one frame ahead) --------------------------------------------------
  CTC_HD void prefetch(int t) {
    pf_live = t < io.T;
    if (!pf_live) return;
    pf_cnt = ctx.uni32(io.surv_cnt[t]);
    if ((uint32_t)lane < pf_cnt) {
      pf_id = io.surv_id[(size_t)t * prm.max_surv + lane];
      pf_lp = io.surv_lp[(size_t)t * prm.max_surv + lane];
    }
  }
  CTC_HD void prefetch_tok() {  // second stage, issued once the ids above have landed
    if (!pf_live || (uint32_t)lane >= pf_cnt) return;
    const TokInfo& g = tab.tok[pf_id];
    pt_h_raw = g.h_raw;
    pt_pow_raw = g.pow_raw;
    pt_h_clean = g.h_clean;
    pt_len_raw = g.len_raw;
    pt_len_clean = g.len_clean;
    pt_flags = g.flags;
    pt_start_flags = g.start_flags;
    pt_start_word_id = g.start_word_id;
    pt_hot = tab.tok_hot ? ((tab.tok_hot[pf_id].min_len & 0xFFFFu) | (tab.tok_hot[pf_id].complete ? 0x80000000u : 0u)) : 0u;
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
      return mode | ((uint32_t)N << 8);
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
    return mode | (first << 8);
  }

  // ---- completion of a beam's open word: the (text (+) partial) prefix ------------------------
  // One TextNode per completed prefix (the reference's memo entry, decoder.py:387-396); lane = beam.
  CTC_HD void completions() {
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      if (j * 64 >= N) break;
      const int i = j * 64 + lane;
      bool todo = false;
      uint32_t tnode = 0, wid = 0, m2 = 0;
      uint64_t part_h = 0;
      if (i < N) {
        const u32x4 k1 = L.beams[i * BREC + 1], k5 = L.beams[i * BREC + 5];
        todo = (k1[2] >> 16) > 0 && k5[1] == 0;
        tnode = k5[0];
        wid = k5[3];
        m2 = k1[3];
        part_h = L.b64[i * 14 + 1];
      }
      const uint64_t m = ctx.ballot(todo);
      if (!m) continue;
      uint32_t idx = text_next + prefix_cnt(m);
      text_next += (uint32_t)ctx.popc64(m);
      if (todo) {
        if (idx + 1 > io.text_cap) {
          status |= ST_TEXT_OVERFLOW;  // (made uniform at the end of the frame)
          idx = io.text_cap - 1;
        }
        const TextNode& src = io.text_nodes[tnode];
        TextNode& dst = io.text_nodes[idx];
        double raw = src.raw_lm;
        if (tab.has_lm) {
          LmState st;
          st.len = src.state.len;
CTC_UNROLL
          for (int k = 0; k < MAX_CTX; ++k) {
            st.words[k] = src.state.words[k];
            st.backoff[k] = src.state.backoff[k];
          }
          const float base = lm_base_score(tab, st, wid, &dst.state);
          raw = raw + lm_word_score(tab, prm, base, m2, 0.0, false);
        } else {
          dst.state.len = src.state.len;
CTC_UNROLL
          for (int k = 0; k < MAX_CTX; ++k) {
            dst.state.words[k] = src.state.words[k];
            dst.state.backoff[k] = src.state.backoff[k];
          }
        }
        const uint64_t th = text_push(src.text_h, part_h);
        const uint32_t cnt = src.hw_cnt + ((m2 & M2_HOT_COMPLETE) ? 1u : 0u);
        const double lmhw = raw + prm.hot_weight * (double)cnt;
        const uint32_t rc = src.ring_cnt + 1 > tab.n_hist ? tab.n_hist : src.ring_cnt + 1;
        uint64_t hh = 0x9E3779B97F4A7C15ull + rc;
CTC_UNROLL
        for (int k = MAX_CTX - 1; k >= 0; --k) {
          const uint64_t rk = k == 0 ? part_h : ((uint32_t)k < rc ? src.ring[k > 0 ? k - 1 : 0] : 0ull);
          dst.ring[k] = rk;
          if ((uint32_t)k < rc) hh = mix64(hh ^ rk) + 0x632BE59BD9B4E019ull;
        }
        dst.text_h = th;
        dst.raw_lm = raw;
        dst.lm_hw = lmhw;
        dst.hist_h = hh;
        dst.hw_cnt = cnt;
        dst.ring_cnt = rc;
        dst.pad0 = 0;
        L.b64[i * 14 + 4] = th;      // c_text_h
        L.bf64[i * 14 + 6] = lmhw;   // c_lm_hw
        L.b64[i * 14 + 9] = hh;      // c_hist_h
        L.b32[i * 28 + 21] = idx;    // comp_node
      }
    }
  }

  // The partial word a candidate ends up with, seen through the prefix / hot-word tables.
  struct PartView {
    uint32_t pl, m2, wid;
    double ps;
  };

  // ---- pool ranking --------------------------------------------------------------------------
  // Ranks the pool entries with score >= thr by (score desc, arrival asc); L.sel[r] = pool index of rank r
  // (bit 31: kept by the history prune) for r < min(count, beam_width). Returns the count.
  CTC_HD uint32_t rank_pool(double thr, bool with_hist) {
    const uint32_t n = pool_n;
    const uint32_t want = (uint32_t)prm.beam_width;
    uint64_t key[PE], hk[PE];
    bool pass[PE];
    uint32_t n_pass = 0;
CTC_UNROLL
    for (int k = 0; k < PE; ++k) {
      const uint32_t e = (uint32_t)(k * 64 + lane);
      pass[k] = false;
      key[k] = ~0ull;
      hk[k] = 0;
      if ((uint32_t)(k * 64) < n) {
        if (e < n) {
          const double sc = L.p_score[e];
          pass[k] = sc >= thr;
          if (pass[k]) key[k] = score_sort_key(sc);
          hk[k] = with_hist ? L.p_hk[e] : 0ull;
          L.rank_rec[e] = mk4q(key[k], hk[k]);
        }
        n_pass += (uint32_t)ctx.popc64(ctx.ballot(pass[k]));
      }
    }
    ctx.wsync();
    uint32_t rank[PE], same[PE], dup[PE];
CTC_UNROLL
    for (int k = 0; k < PE; ++k) rank[k] = same[k] = dup[k] = 0;
    for (uint32_t j = 0; j < n; ++j) {  // every lane reads the same record: an LDS broadcast
      const u32x4 r = L.rank_rec[j];
      const uint64_t x = q_lo(r), xh = q_hi(r);
CTC_UNROLL
      for (int k = 0; k < PE; ++k) {
        if ((uint32_t)(k * 64) < n) {
          const bool better = x < key[k];
          rank[k] += better ? 1u : 0u;
          same[k] += x == key[k] ? 1u : 0u;
          dup[k] |= (better && xh == hk[k]) ? 1u : 0u;
        }
      }
    }
    // equal scores (rare): the earlier arrival ranks first (heapq.nlargest is stable)
    bool tie = false;
CTC_UNROLL
    for (int k = 0; k < PE; ++k) tie = tie || (pass[k] && same[k] > 1u);
    if (ctx.ballot(tie) != 0ull) {
      uint32_t arr[PE];
CTC_UNROLL
      for (int k = 0; k < PE; ++k) {
        const uint32_t e = (uint32_t)(k * 64 + lane);
        arr[k] = e < n ? L.p_arr[e] : 0u;
      }
      for (uint32_t j = 0; j < n; ++j) {
        const u32x4 r = L.rank_rec[j];
        const uint64_t x = q_lo(r), xh = q_hi(r);
        const uint32_t xa = L.p_arr[j];
CTC_UNROLL
        for (int k = 0; k < PE; ++k) {
          const bool before = pass[k] && x == key[k] && xa < arr[k];
          rank[k] += before ? 1u : 0u;
          dup[k] |= (before && xh == hk[k]) ? 1u : 0u;
        }
      }
    }
CTC_UNROLL
    for (int k = 0; k < PE; ++k) {
      const uint32_t e = (uint32_t)(k * 64 + lane);
      if (pass[k] && rank[k] < want) L.sel[rank[k]] = e | ((with_hist && dup[k]) ? 0u : 0x80000000u);
    }
    ctx.wsync();
    return n_pass;
  }

  // keep only the best beam_width pool entries (exact: pruning is monotone)
  CTC_HD void compact_pool() {
    const double mx = key_to_score(runmax);
    uint32_t n = rank_pool(mx + prm.beam_prune_logp, false);
    if (n > (uint32_t)prm.beam_width) n = (uint32_t)prm.beam_width;
    double g_score[SLB], g_logit[SLB];
    uint64_t g_hk[SLB];
    uint32_t g_arr[SLB], g_don[SLB], g_wid[SLB], g_m2[SLB];
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const uint32_t r = (uint32_t)(j * 64 + lane);
      g_score[j] = 0.0;
      g_logit[j] = 0.0;
      g_hk[j] = 0;
      g_arr[j] = g_don[j] = g_wid[j] = g_m2[j] = 0;
      if (r < n) {
        const uint32_t e = L.sel[r] & 0x7FFFFFFFu;
        g_score[j] = L.p_score[e];
        g_logit[j] = L.p_logit[e];
        g_hk[j] = L.p_hk[e];
        g_arr[j] = L.p_arr[e];
        g_don[j] = L.p_don[e];
        g_wid[j] = L.p_wid[e];
        g_m2[j] = L.p_m2[e];
      }
    }
    ctx.wsync();
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const uint32_t r = (uint32_t)(j * 64 + lane);
      if (r < n) {
        L.p_score[r] = g_score[j];
        L.p_logit[r] = g_logit[j];
        L.p_hk[r] = g_hk[j];
        L.p_arr[r] = g_arr[j];
        L.p_don[r] = g_don[j];
        L.p_wid[r] = g_wid[j];
        L.p_m2[r] = g_m2[j];
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
        if ((int)(r >> 6) == j) k = ctx.bcast64(asc_key(g_score[j]), (int)(r & 63u));
      kth_key = k;
    }
    ctx.wsync();
  }

  // ---- one pass: candidates of the labels [s0, s1) -----------------------------------------------
  CTC_HD void pass(uint32_t s0, uint32_t s1) {
    const uint32_t Nn = (uint32_t)N;
    const uint32_t Q = (s1 - s0) * Nn;
    const uint32_t rcpN = Nn ? 65536u / Nn + 1u : 0u;  // v / N == (v * rcpN) >> 16 for v * N < 65536
    bool valid[SLB], is_rep[SLB];
    uint32_t bi[SLB], ls[SLB], lid[SLB], br[SLB], rep[SLB], pl0[SLB], m2_0[SLB];
    uint64_t kp[SLB], ck[SLB];
    double lg[SLB];
    Lab lb[SLB];
    uint64_t c_hist_sel[SLB];  // history hash the candidate's text ends up with
    double lmhw_sel[SLB];
    PrefixEntry pre_p[SLB];
    HotEntry pre_h[SLB];
    bool want_p[SLB], want_h[SLB];
    // clear the match table and the member masks
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const int v = j * 64 + lane;
      L.gmask[v] = mk4(0, 0, 0, 0);
      ((CTC_LDS u32x4*)L.table.p)[v] = mk4(0, 0, 0, 0);
    }
    // ---- generation: branch, merge key, summed logit
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const uint32_t v = (uint32_t)(j * 64 + lane);
      valid[j] = v < Q;
      is_rep[j] = false;
      bi[j] = ls[j] = lid[j] = br[j] = pl0[j] = m2_0[j] = 0;
      rep[j] = v;
      kp[j] = ck[j] = 0;
      lg[j] = 0.0;
      c_hist_sel[j] = 0;
      lmhw_sel[j] = 0.0;
      want_p[j] = want_h[j] = false;
      pre_p[j].key = 0; pre_p[j].word_id = 0; pre_p[j].flags = 0;
      pre_h[j].key = 0; pre_h[j].min_len = 0; pre_h[j].complete = 0;
      if ((uint32_t)(j * 64) >= Q) continue;
      if (valid[j]) {
        const uint32_t sl = (v * rcpN) >> 16;
        const uint32_t i = v - sl * Nn;
        const uint32_t s = s0 + sl;
        bi[j] = i;
        ls[j] = s;
        const u32x4 sv = L.surv[s];
        lid[j] = sv[0];
        lb[j] = label_of(s, sv[0]);
        const u32x4 k0 = L.beams[i * BREC], k1 = L.beams[i * BREC + 1], k2 = L.beams[i * BREC + 2];
        const u32x4 k4 = L.beams[i * BREC + 4];
        const uint32_t meta1 = k1[2];
        const uint32_t pl = meta1 >> 16;
        pl0[j] = pl;
        m2_0[j] = k1[3];
        const uint32_t b = branch_of(lb[j].flags, sv[1], sv[0], i, meta1 & 0xFFFFu);
        br[j] = b;
        uint64_t kt = q_lo(k0), p = q_hi(k0);
        uint64_t hh = q_lo(k4);
        double lmhw = bits_f64(q_hi(k2));
        if (b == BR_BOUNDARY || b == BR_SPACE) {
          if (pl > 0) {
            kt = q_lo(k2);                      // c_text_h
            hh = q_hi(k4);                      // c_hist_h
            lmhw = L.bf64[i * 14 + 6];          // c_lm_hw
          }
          p = b == BR_BOUNDARY ? lb[j].h_clean : 0;
        } else if (b == BR_APPEND) {
          p = str_concat(p, lb[j].pow_raw, lb[j].h_raw);
          // first probe of both tables issued here: in flight across the match
          if (p != 0) {
            const uint64_t hk = mix64(p);
            want_p[j] = (k1[3] & PF_ON_TABLE) && tab.prefixes;
            want_h[j] = (k1[3] & M2_HOT_ON) && tab.hot;
            if (want_p[j]) pre_p[j] = tab.prefixes[hk & tab.prefix_mask];
            if (want_h[j]) pre_h[j] = tab.hot[hk & tab.hot_mask];
          }
        }
        kp[j] = p;
        c_hist_sel[j] = hh;
        lmhw_sel[j] = lmhw;
        ck[j] = fin64(kt * 0x9E3779B97F4A7C15ull + p * 0xC2B2AE3D27D4EB4Full + (uint64_t)(sl + 1u) * 0x165667B19E3779F9ull);
        lg[j] = bits_f64(q_lo(k1)) + bits_f64(pack64(sv[2], sv[3]));
        L.c_logit[v] = lg[j];
      }
    }
    ctx.wsync();
    tick<W_PROF_GEN>();
    // ---- match: the smallest candidate of the largest tag owns a slot; the others of its key join it, the
    // rest move to their next slot (different key bits, then linear)
    {
      bool open[SLB];
      uint32_t slot[SLB];
CTC_UNROLL
      for (int j = 0; j < SLB; ++j) {
        open[j] = valid[j];
        slot[j] = (uint32_t)(ck[j] >> 7) & (uint32_t)(TS - 1);
      }
      for (uint32_t round = 0;; ++round) {
        bool any_open = false;
CTC_UNROLL
        for (int j = 0; j < SLB; ++j) any_open = any_open || open[j];
        if (ctx.ballot(any_open) == 0ull) break;
CTC_UNROLL
        for (int j = 0; j < SLB; ++j) {
          const uint32_t v = (uint32_t)(j * 64 + lane);
          if (open[j]) ctx.lds_max_u64(&L.table[slot[j]], (ck[j] & ~127ull) | (uint64_t)(127u - v));
        }
        ctx.wsync();
CTC_UNROLL
        for (int j = 0; j < SLB; ++j) {
          if (open[j]) {
            const uint64_t got = L.table[slot[j]];
            if ((got & ~127ull) == (ck[j] & ~127ull)) {
              rep[j] = 127u - (uint32_t)(got & 127ull);
              open[j] = false;
            } else {
              slot[j] = round < 5u ? ((uint32_t)(ck[j] >> (15 + 8 * round)) & (uint32_t)(TS - 1))
                                   : ((slot[j] + 1u) & (uint32_t)(TS - 1));
            }
          }
        }
        ctx.wsync();
      }
    }
    // members announce themselves to their representative
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const uint32_t v = (uint32_t)(j * 64 + lane);
      if (valid[j] && rep[j] != v) ctx.lds_or_u32(&((CTC_LDS uint32_t*)L.gmask.p)[rep[j] * 4 + (v >> 5)], 1u << (v & 31u));
      is_rep[j] = valid[j] && rep[j] == v;
    }
    ctx.wsync();
    tick<W_PROF_MATCH>();
    // ---- fold the group's logits in ascending beam rank (decoder.py:217-223); donor = last arrival
    uint32_t imax[SLB];
    {
      u32x4 gm[SLB];
      bool more = false;
CTC_UNROLL
      for (int j = 0; j < SLB; ++j) {
        const uint32_t v = (uint32_t)(j * 64 + lane);
        gm[j] = mk4(0, 0, 0, 0);
        imax[j] = bi[j];
        if (is_rep[j]) {
          gm[j] = L.gmask[v];
          uint32_t top = 0xFFFFFFFFu;
CTC_UNROLL
          for (int w = 0; w < 4; ++w)
            if (gm[j][w]) top = (uint32_t)(w * 32 + 31 - ctx.clz32(gm[j][w]));
          if (top != 0xFFFFFFFFu) {
            imax[j] = bi[j] + (top - v);  // members share the label: consecutive beam indices
            more = true;
          }
        }
      }
      while (ctx.ballot(more) != 0ull) {
        more = false;
CTC_UNROLL
        for (int j = 0; j < SLB; ++j) {
          if (is_rep[j]) {
            uint32_t mbit = 0xFFFFFFFFu;
CTC_UNROLL
            for (int w = 3; w >= 0; --w)
              if (gm[j][w]) mbit = (uint32_t)(w * 32 + ctx.ctz32(gm[j][w]));
            if (mbit != 0xFFFFFFFFu) {
CTC_UNROLL
              for (int w = 0; w < 4; ++w)
                if ((int)(mbit >> 5) == w) gm[j][w] &= gm[j][w] - 1u;
              lg[j] = lse2(lg[j], L.c_logit[mbit]);
              more = more || (gm[j][0] | gm[j][1] | gm[j][2] | gm[j][3]) != 0u;
            }
          }
        }
      }
    }
    tick<W_PROF_FOLD>();
    // ---- score the representatives (decoder.py:346-424)
    double score[SLB];
    uint64_t my_key[SLB];
    PartView pv[SLB];
    uint64_t pass_key = 0;
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      score[j] = 0.0;
      my_key[j] = 0;
      pv[j].pl = pv[j].m2 = pv[j].wid = 0;
      pv[j].ps = 0.0;
      if (is_rep[j]) {
        const uint32_t i = bi[j];
        const uint32_t b = br[j];
        PartView q;
        if (b == 0) {  // blank / repeat: unchanged
          q.pl = pl0[j];
          q.m2 = m2_0[j];
          q.wid = L.b32[i * 28 + 23];
          q.ps = L.bf64[i * 14 + 7];
        } else if (b == BR_BOUNDARY && lb[j].len_clean > 0) {  // a new word starts with the clean label
          const uint32_t hmin = lb[j].hot_min, hcomp = lb[j].hot_complete;
          q.pl = lb[j].len_clean;
          q.m2 = (lb[j].start_flags & (PF_PARTIAL_MASK | PF_ON_TABLE)) | (hmin ? M2_HOT_ON : 0u) | (hcomp ? M2_HOT_COMPLETE : 0u) | (hmin << 8);
          q.wid = lb[j].start_word_id;
          q.ps = partial_score(tab, prm, lb[j].start_flags, hmin, q.pl);
        } else if (b == BR_APPEND) {
          uint32_t pf = 0, nw = 0, hmin = 0, hcomp = 0;
          bool on = false, hon = false;
          const uint64_t key = kp[j];
          const uint64_t hk = mix64(key);
          if (want_p[j]) {
            uint64_t sp = hk & tab.prefix_mask;
            PrefixEntry ep = pre_p[j];
            while (ep.key != key && ep.key != 0) {
              sp = (sp + 1) & tab.prefix_mask;
              ep = tab.prefixes[sp];
            }
            on = ep.key == key;
            nw = ep.word_id;
            pf = ep.flags;
          }
          if (want_h[j]) {
            uint64_t sh = hk & tab.hot_mask;
            HotEntry eh = pre_h[j];
            while (eh.key != key && eh.key != 0) {
              sh = (sh + 1) & tab.hot_mask;
              eh = tab.hot[sh];
            }
            hon = eh.key == key;
            hmin = eh.min_len;
            hcomp = eh.complete;
          }
          q.pl = pl0[j] + lb[j].len_raw;
          q.m2 = (on ? (PF_ON_TABLE | (pf & PF_PARTIAL_MASK)) : 0u) | (hon ? M2_HOT_ON : 0u) | ((hon && hcomp) ? M2_HOT_COMPLETE : 0u) |
                 ((hon ? hmin : 0u) << 8);
          q.wid = on ? nw : 0;
          q.ps = partial_score(tab, prm, on ? pf : 0u, hon ? hmin : 0u, q.pl);
        } else {  // space, or a bare boundary mark: the open word is empty
          q.pl = 0;
          q.m2 = EMPTY_PARTIAL_M2;
          q.wid = 0;
          q.ps = 0.0;
        }
        pv[j] = q;
        score[j] = total_score(tab, lg[j], lmhw_sel[j], q.ps, q.pl);
        my_key[j] = asc_key(score[j]);
        if (my_key[j] > pass_key) pass_key = my_key[j];
      }
    }
    pass_key = ctx.wave_max_u64(pass_key);
    if (pass_key > runmax) runmax = pass_key;
    const double thr = key_to_score(runmax) + prm.beam_prune_logp;
    tick<W_PROF_SCORE>();
    // ---- push what can still matter into the pool
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      if ((uint32_t)(j * 64) >= Q) continue;
      const bool push = is_rep[j] && score[j] >= thr && my_key[j] > kth_key;
      const uint64_t m = ctx.ballot(push);
      const uint32_t k = pool_n + prefix_cnt(m);
      pool_n += (uint32_t)ctx.popc64(m);
      if (push) {
        L.p_score[k] = score[j];
        L.p_logit[k] = lg[j];
        L.p_arr[k] = ls[j] * Nn + bi[j];
        L.p_don[k] = (ls[j] << 8) | imax[j];
        L.p_wid[k] = pv[j].wid;
        L.p_m2[k] = pv[j].m2;
        // (history, partial, last_char) folded to 64 bits (decoder.py:250-254): equality of the folds stands in
        // for equality of the triple (its members are 61/64-bit string hashes already)
        L.p_hk[k] = fin64(c_hist_sel[j] * 0x9E3779B97F4A7C15ull + kp[j] * 0xC2B2AE3D27D4EB4Full + (uint64_t)(lid[j] + 1u));
      }
    }
    ctx.wsync();
  }

  // ---- one frame ---------------------------------------------------------------------------------
  CTC_HD void step(int t) {
    const int frame = io.first_frame + t;
    const uint32_t ns = pf_cnt;
    pool_n = 0;
    runmax = asc_key(-INFINITY);
    kth_key = 0;
    need = 0;
    // last labels of the live beams: is there a beam that does not end in beam 0's label, and which first?
    uint32_t lc0 = NO_CHAR, f1 = (uint32_t)N;
    {
      uint32_t lc[SLB];
CTC_UNROLL
      for (int j = 0; j < SLB; ++j) {
        const int i = j * 64 + lane;
        lc[j] = i < N ? (L.b32[i * 28 + 6] & 0xFFFFu) : 0u;
      }
      lc0 = ctx.bcast32(lc[0], 0);
CTC_UNROLL
      for (int j = SLB - 1; j >= 0; --j) {
        const int i = j * 64 + lane;
        const uint64_t m = ctx.ballot(i < N && lc[j] != lc0);
        if (m) f1 = (uint32_t)(j * 64 + ctx.ctz64(m));
      }
    }
    // survivors -> LDS (ids, log-probs, branch modes, label constants), 64 labels at a time
    for (uint32_t base = 0; base < ns; base += 64u) {
      const uint32_t s = base + (uint32_t)lane;
      const bool mine = s < ns;
      uint32_t id = 0, fl = TK_BLANK;
      double lp = 0.0;
      if (base == 0) {
        if (mine) {
          id = pf_id;
          lp = pf_lp;
          fl = pt_flags;
          L.lab[s * 3] = mk4q(pt_h_raw, pt_pow_raw);
          L.lab[s * 3 + 1] = mk4((uint32_t)pt_h_clean, (uint32_t)(pt_h_clean >> 32), pt_len_raw, pt_len_clean);
          L.lab[s * 3 + 2] = mk4(pt_flags, pt_start_flags, pt_start_word_id, pt_hot);
        }
      } else if (mine) {
        id = io.surv_id[(size_t)t * prm.max_surv + s];
        lp = io.surv_lp[(size_t)t * prm.max_surv + s];
        fl = tab.tok[id].flags;
      }
      const uint32_t mw = mode_block(fl, id, lc0, f1);
      if (mine) L.surv[s] = mk4(id, mw, (uint32_t)f64_bits(lp), (uint32_t)(f64_bits(lp) >> 32));
    }
    tick<W_PROF_LOAD>();
    if (need) completions();
    ctx.wsync();
    prefetch(t + 1);  // lands while this frame's candidates are processed
    tick<W_PROF_COMP>();
    // labels are taken in passes of whole labels (<= C candidates); before a pass that might not fit the
    // pool, the pool is compacted to its best beam_width entries
    uint32_t per = (uint32_t)C / (uint32_t)(N > 0 ? N : 1);
    if (per == 0) per = 1;
    for (uint32_t s0 = 0; s0 < ns; s0 += per) {
      const uint32_t s1 = s0 + per < ns ? s0 + per : ns;
      if (pool_n + (s1 - s0) * (uint32_t)N > (uint32_t)P) {
        compact_pool();
        tick<W_PROF_COMPACT>();
      }
      pass(s0, s1);
    }
    prefetch_tok();
    finish_frame(frame);
  }

  // threshold prune, top-B, history prune, next beam table (decoder.py:545-554)
  CTC_HD void finish_frame(int frame) {
    const double thr = key_to_score(runmax) + prm.beam_prune_logp;
    const bool hist = prm.prune_history != 0;
    uint32_t n = rank_pool(thr, hist);
    if (n > (uint32_t)prm.beam_width) n = (uint32_t)prm.beam_width;
    tick<W_PROF_RANK>();
    // nothing passed the threshold: only possible with non-finite scores (NaN rows) or a positive
    // beam_prune_logp; the reference then dies on max([]) (decoder.py:545) -- reported through the status
    if (n == 0) status |= ST_NO_BEAMS;
    // gather everything the new records need, then write them (one table, no double buffer)
    bool kept[SLB];
    uint32_t dst[SLB];
    u32x4 o0[SLB], o1[SLB], o2[SLB], o3[SLB], o4[SLB], o5[SLB], o6[SLB];
    uint32_t n_new = 0;
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
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
      // the payload is the donor's (the last-arriving duplicate, decoder.py:221-223): its branch, not the
      // representative's. Emission nodes: one per kept beam whose label is not a blank / repeat.
      uint32_t b = 0, s = 0, i = 0, idx = 0, c = 0;
      Lab lb;
      u32x4 k1 = mk4(0, 0, 0, 0);
      if (kept[j]) {
        idx = w & 0x7FFFFFFFu;
        const uint32_t don = L.p_don[idx];
        s = don >> 8;
        i = don & 0xFFu;
        const u32x4 sv = L.surv[s];
        c = sv[0];
        lb = label_of(s, c);
        k1 = L.beams[i * BREC + 1];
        b = branch_of(lb.flags, sv[1], c, i, k1[2] & 0xFFFFu);
      }
      const uint64_t em = ctx.ballot(kept[j] && b != 0);
      uint32_t e = emit_next + prefix_cnt(em);
      emit_next += (uint32_t)ctx.popc64(em);
      if (kept[j]) {
        const u32x4 k0 = L.beams[i * BREC], k2 = L.beams[i * BREC + 2];
        const u32x4 k3 = L.beams[i * BREC + 3], k4 = L.beams[i * BREC + 4], k5 = L.beams[i * BREC + 5];
        const u32x4 k6 = L.beams[i * BREC + 6];
        const uint32_t pl = k1[2] >> 16;
        uint64_t th = q_lo(k0), ph = q_hi(k0), hh = q_lo(k4);
        const uint64_t cth = q_lo(k2), chh = q_hi(k4);
        double lmhw = bits_f64(q_hi(k2));
        const double clm = bits_f64(q_lo(k3));
        uint32_t tnode = k5[0], cnode = k5[1], enode = k5[2];
        int32_t pst = (int32_t)k6[0], pen = (int32_t)k6[1];
        uint32_t depth = k6[2];
        const uint32_t m2 = L.p_m2[idx], wid = L.p_wid[idx];
        uint32_t npl = pl;
        if (b == 0) {
          if (!(lb.flags & TK_BLANK)) pen = frame + 1;  // decoder.py:453-461
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
              ph = lb.h_clean;
              npl = lb.len_clean;
              pst = frame;
              pen = frame + 1;
            } else {
              ph = 0;
              npl = 0;
              pst = -1;
              pen = -1;
            }
          } else {  // BR_APPEND (decoder.py:518-534)
            ph = str_concat(ph, lb.pow_raw, lb.h_raw);
            npl = pl + lb.len_raw;
            pst = pst < 0 ? frame : pst;
            pen = frame + 1;
          }
          cnode = 0;
          if (e >= io.emit_cap) {
            status |= ST_EMIT_OVERFLOW;
            e = io.emit_cap - 1;
          }
          EmitNode en;
          en.parent = enode;
          en.tok_branch = c | (b << 16);
          en.wstart = wst;
          en.wend = wen;
          io.emit_nodes[e] = en;
          enode = e;
          depth += 1;
        }
        double ps = 0.0;
        if (npl > 0) ps = partial_score(tab, prm, m2 & PF_PARTIAL_MASK, (m2 & M2_HOT_ON) ? ((m2 >> 8) & 0xFFFFu) : 0u, npl);
        o0[j] = mk4q(th, ph);
        const uint64_t lgb = f64_bits(L.p_logit[idx]);
        o1[j] = mk4((uint32_t)lgb, (uint32_t)(lgb >> 32), c | (npl << 16), m2);
        o2[j] = mk4q(cth, f64_bits(lmhw));
        o3[j] = mk4q(f64_bits(clm), f64_bits(ps));
        o4[j] = mk4q(hh, chh);
        o5[j] = mk4(tnode, cnode, enode, wid);
        o6[j] = mk4((uint32_t)pst, (uint32_t)pen, depth, 0u);
      }
    }
    ctx.wsync();
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
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
      root.hist_h = hist_hash(root.ring, 0);
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
      uint64_t hh = 0x9E3779B97F4A7C15ull + m.ring_cnt;
CTC_UNROLL
      for (int k = MAX_CTX - 1; k >= 0; --k) {
        tn.ring[k] = m.ring[k];
        if ((uint32_t)k < m.ring_cnt) hh = mix64(hh ^ m.ring[k]) + 0x632BE59BD9B4E019ull;
      }
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
    ctx.mem_sync();
    if (fold) completions();
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
        ck[j] = fin64(kt * 0x9E3779B97F4A7C15ull + 0x165667B19E3779F9ull);
        lg[j] = L.bf64[v * 14 + 2];
        L.c_logit[v] = lg[j];
      }
    }
    ctx.wsync();
    if (fold) {
      bool open[SLB];
      uint32_t slot[SLB];
CTC_UNROLL
      for (int j = 0; j < SLB; ++j) {
        open[j] = valid[j];
        slot[j] = (uint32_t)(ck[j] >> 7) & (uint32_t)(TS - 1);
      }
      for (uint32_t round = 0;; ++round) {
        bool any_open = false;
CTC_UNROLL
        for (int j = 0; j < SLB; ++j) any_open = any_open || open[j];
        if (ctx.ballot(any_open) == 0ull) break;
CTC_UNROLL
        for (int j = 0; j < SLB; ++j) {
          const uint32_t v = (uint32_t)(j * 64 + lane);
          if (open[j]) ctx.lds_max_u64(&L.table[slot[j]], (ck[j] & ~127ull) | (uint64_t)(127u - v));
        }
        ctx.wsync();
CTC_UNROLL
        for (int j = 0; j < SLB; ++j) {
          if (open[j]) {
            const uint64_t got = L.table[slot[j]];
            if ((got & ~127ull) == (ck[j] & ~127ull)) {
              rep[j] = 127u - (uint32_t)(got & 127ull);
              open[j] = false;
            } else {
              slot[j] = round < 5u ? ((uint32_t)(ck[j] >> (15 + 8 * round)) & (uint32_t)(TS - 1))
                                   : ((slot[j] + 1u) & (uint32_t)(TS - 1));
            }
          }
        }
        ctx.wsync();
      }
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
        L.p_score[k] = score[j];
        L.p_logit[k] = lg[j];
        L.p_arr[k] = (uint32_t)(j * 64 + lane);
        L.p_don[k] = donor[j];
        L.p_wid[k] = 0;
        L.p_m2[k] = 0;
        L.p_hk[k] = 0;
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
        const uint32_t d = L.p_don[idx];
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
      const uint32_t d = L.p_don[idx];
      OutBeam& ob = io.out[r];
      ob.logit_score = L.p_logit[idx];
      ob.lm_score = L.p_score[idx];
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
    if (io.prof && lane == 0) t_last = ctx.clock();
    prefetch(0);
    prefetch_tok();
    for (int t = 0; t < io.T; ++t) step(t);
    finalise();
    tick<W_PROF_FINAL>();
    if (io.prof && lane == 0) {
CTC_UNROLL
      for (int k = 0; k < W_PROF_N; ++k) io.prof[k] = t_acc[k];
    }
  }
};

}  // namespace ctc
