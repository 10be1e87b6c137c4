// beam_wave.h -- the prefix-beam recursion as ONE wavefront per utterance (reference:
// BeamSearchDecoderCTC._partial_decode_logits decoder.py:426-556, _finalize_beams :558-602,
// _get_lm_beams :346-424, _merge_beams :211-224, _prune_history :227-258).
//
// Same semantics as beam_core.h (which stays the general path: several language models, beam widths
// above 128, survivor bounds above WAVE_SURV_CAP) but shaped for what a frame of the recursion really is on
// CDNA4: a few dozen live beams, a handful of surviving labels, ~100 candidates -- and, since round 4, for FOUR
// wavefronts per SIMD: at beam_width 100 a wave owns 9 920 bytes of LDS and is compiled for 128 registers, so
// sixteen utterances are resident per CU and a batch of 4096 utterances is one round on the 256 CUs.
//   * LDS holds only what the candidate passes read per (label, beam): three 16-byte columns per beam
//     ({text hash, partial hash} {logit, last label | length, table view} {lm + hot-word score, hash of text (+) open word}),
//     column-wise so that consecutive lanes read consecutive 16-byte words;
//   * everything else of a beam -- the scores of its pending word completion, its history hashes, its emission
//     chain, the frames of the open word -- is a 64-byte record in global memory (ColdRec, two buffers used
//     alternately by the table builds; L2 resident), fetched by the lanes that need it while they wait for the
//     table probes anyway;
//   * candidates (label s, beam i) are spread densely over the lanes, whole labels per pass, one candidate per
//     lane; with more than 64 live beams a label is two half passes matched together;
//   * duplicates are found by a wave-wide hash match: one ds_max_u64 per candidate on
//     (57-bit key tag | 127 - candidate) and one read back -- the smallest candidate of the largest tag
//     owns a slot, losers move on to their next slot; the members of a group announce themselves to
//     their representative through one ds_or on a 64-bit mask, which gives the representative the
//     fold order (ascending beam rank, decoder.py:217-223) and the donor (last arrival) at once;
//   * the pool of merged, scored candidates keeps 24 bytes per entry in LDS -- {score key, history key} for the
//     ranking walk, {arrival, payload index, donor} -- and the rest (summed logit, new partial word, its table
//     view) in a per-utterance payload line in global memory that only the table build reads; the pool holds
//     beam_width + 28 entries, a push that would not fit first compacts it to its best beam_width (exact:
//     pruning is monotone);
//   * threshold, top-B and the history prune are one counting sweep over the pool's {score key, history
//     key} words that every lane reads at the same address (an LDS broadcast);
//   * the next table is built in place by gathering (pool entry, payload, donor columns, donor record) per kept rank;
//   * runs of frames whose only survivor is the label every beam already ends in (most frames of a real
//     posterior) are consumed in place, 64 at a time, each checked to leave the order intact (label_run).
// Diagnostics, all off by default: CTC_WAVE_TRACE (device printf of pool / beam records), CTC_RUN_TRACE
// (host: which frames label_run consumed), CTC_SIM_DEBUG (index checks in the simulator), CTC_STATS (simulator:
// live beams / survivors / pool entries per frame), tick<>() phase timers (ctcdec_profile_phases).
// Everything a lane shares with another lane goes through LDS, global memory or a cross-lane instruction; `wsync()`
// marks the points where LDS traffic of different lanes meets (on the device a compiler fence -- one
// wave issues its LDS operations in order --, in the 64-fiber test simulator a rendezvous). Global memory written by
// one lane and read by another of the same wave needs no more than program order either: the accesses of a wave
// pass through its CU's vector cache in issue order.
#pragma once
#include "beam_core.h"

namespace ctc {

typedef uint32_t u32x4 __attribute__((vector_size(16)));
typedef uint32_t u32x4a __attribute__((vector_size(16), may_alias));  // 16-byte view of a structure in global memory
typedef uint32_t u32x2a __attribute__((vector_size(8), may_alias));

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

// History of a text: its last n_hist word hashes (newest first) folded to 64 bits. The words are packed pairs
// of 31-bit polynomial hashes: xor under distinct rotations keeps equal tuples equal and makes unequal ones collide with
// probability ~2^-62 (no multiplies: this runs for every completed word).
// NH: the ring entries that can be in use (the decoder's context length: 3 for models up to order 4)
template <int NH = MAX_CTX>
CTC_HD uint64_t wave_hist_fold(const uint64_t* ring, uint32_t cnt) {
  uint64_t h = 0x9E3779B97F4A7C15ull * (uint64_t)(cnt + 1u);
CTC_UNROLL
  for (int k = 0; k < NH; ++k)
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

// ---- layout ------------------------------------------------------------------------------------------------
// A live beam, LDS part: three columns of 16-byte words
//   A: text_h, part_h          B: logit, meta1, meta2          C: lm_hw, c_text_h (the text with the open word closed: valid
//                                                                    with M2_COMP; the merge key of a candidate that closes it;
//                                                                    WITHOUT M2_COMP its low word is the beam's text node, which
//                                                                    the completion starts from -- no trip to the ColdRec first)
// meta1 = last label | code points of the open word << 16; meta2 = prefix-table / hot-word view of the open word, plus
// M2_COMP: the completion of the open word (text (+) word: its TextNode, scores and history hash) exists in the beam's
// ColdRec. The rest of the beam is its ColdRec in global memory (beam_core.h).
constexpr uint32_t M2_COMP = 64u;
constexpr int WAVE_LAB = 16;        // survivors are staged in LDS (ids, modes, label constants) 16 at a time
constexpr int WAVE_SURV_CAP = 480;  // survivors per frame this kernel handles: arrival = s * N + beam has to fit 16 bits

// 65536 / n + 1 and max(1, 64 / n) for the n <= 64 live beams of a candidate pass: looked up (one scalar load), not divided
// (an integer division is ~35 instructions of float reciprocal and fix-up on this hardware)
struct WaveDivTab {
  uint32_t rcp[65];
  uint32_t per[65];
};
static constexpr WaveDivTab WAVE_DIV = {
    {0, 65537, 32769, 21846, 16385, 13108, 10923, 9363, 8193, 7282, 6554, 5958, 5462, 5042, 4682, 4370, 4097, 3856, 3641, 3450, 3277, 3121, 2979, 2850, 2731, 2622, 2521, 2428, 2341, 2260, 2185, 2115, 2049, 1986, 1928, 1873, 1821, 1772, 1725, 1681, 1639, 1599, 1561, 1525, 1490, 1457, 1425, 1395, 1366, 1338, 1311, 1286, 1261, 1237, 1214, 1192, 1171, 1150, 1130, 1111, 1093, 1075, 1058, 1041, 1025},
    {64, 64, 32, 21, 16, 12, 10, 9, 8, 7, 6, 5, 5, 4, 4, 4, 4, 3, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}};

template <int BW>
struct WaveShape {
  static constexpr int SLB = (BW + 63) / 64;  // beam slots per lane (lane = beam phases)
  static constexpr int P = BW + 28;           // pool capacity (a multiple of 4: the ranking walk reads four entries at a time);
                                              // a push that would not fit first compacts the pool to its best beam_width
  static constexpr int PE = (P + 63) / 64;    // pool entries per lane
  static constexpr int TS = 128;              // match-table slots
  static constexpr int STAGE = BW > 64 ? (BW - 64) * 48 : 0;  // build of more than 64 beams: columns of the ranks 64..
  static_assert(P % 4 == 0, "pool capacity must be a multiple of 4");
  static_assert(BW <= 128, "one label's candidates are at most two per lane");
};

struct WaveLds {
  LPtr<u32x4> hA, hB, hC;  // [BW] each
  // pool of merged, scored candidates
  LPtr<u32x4> pk;          // [P]  {score key (ascending = better), history key}
  LPtr<uint64_t> pa;       // [P]  low word: arrival | payload index << 16; high word: donor
  // donor word: beam index | label << 8 | label is blank << 29 | donor's branch << 30
  LPtr<u32x4> stage;       // the regions below, as one: columns of the ranks 64.. while a table of more than 64 beams is built
  LPtr<u32x4> lab;         // [WAVE_LAB * 4]  {id, mode word, lp lo, lp hi} {h_raw, pow_raw} {h_clean, len_raw, len_clean}
                           //                 {flags, start_flags, start_word_id, hot}
  LPtr<uint64_t> table;    // [128] match table of one pass of <= 128 candidates
  // views of the table's words while no match is running (the last rendezvous of a match is behind them):
  LPtr<uint32_t> gmask;    // [64 * 2]  members of the group a candidate represents
  LPtr<double> scr;        // [BW]      label_run: one score per live beam
  LPtr<uint32_t> sel;      // [BW]      rank -> pool index
  // scalar views of the columns
  LPtr<uint64_t> a64, b64, c64;
  LPtr<uint32_t> b32, c32;
};

template <int BW>
CTC_HD size_t wave_lds_carve(WaveLds& o, lds_bytes_t base) {
  typedef WaveShape<BW> S;
  lds_bytes_t p = base;
  o.hA = lds_take<u32x4>(p, 16 * BW);
  o.hB = lds_take<u32x4>(p, 16 * BW);
  o.hC = lds_take<u32x4>(p, 16 * BW);
  o.a64.p = (CTC_LDS uint64_t*)o.hA.p;
  o.b64.p = (CTC_LDS uint64_t*)o.hB.p;
  o.c64.p = (CTC_LDS uint64_t*)o.hC.p;
  o.b32.p = (CTC_LDS uint32_t*)o.hB.p;
  o.c32.p = (CTC_LDS uint32_t*)o.hC.p;
  o.pk = lds_take<u32x4>(p, 16 * S::P);
  o.pa = lds_take<uint64_t>(p, 8 * S::P);
  lds_bytes_t stage0 = p;
  o.stage.p = (CTC_LDS u32x4*)p;
  o.lab = lds_take<u32x4>(p, 16 * 4 * WAVE_LAB);
  lds_bytes_t shared0 = p;
  o.table = lds_take<uint64_t>(p, 8 * S::TS);
  o.gmask.p = (CTC_LDS uint32_t*)shared0;
  o.scr.p = (CTC_LDS double*)shared0;
  o.sel.p = (CTC_LDS uint32_t*)shared0;
  size_t shared = (size_t)(p - shared0);
  if (shared < (size_t)(8 * BW)) shared = align16((size_t)(8 * BW));  // label_run's scores
  if ((size_t)(shared0 - stage0) + shared < (size_t)S::STAGE) shared = (size_t)S::STAGE - (size_t)(shared0 - stage0);
  p = shared0 + shared;
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
CTC_HD int wave_bucket(int beam_width) { return beam_width <= 64 ? 64 : beam_width <= 100 ? 100 : 128; }
// payload lines per utterance: one per candidate that can be pushed in a frame
CTC_HD size_t wave_pay_stride(const DecodeParams& p) { return (size_t)p.max_surv * (size_t)wave_bucket(p.beam_width); }

constexpr int W_PROF_LOAD = 0, W_PROF_COMP = 1, W_PROF_GEN = 2, W_PROF_MATCH = 3, W_PROF_FOLD = 4, W_PROF_SCORE = 5,
              W_PROF_RANK = 6, W_PROF_BUILD = 7, W_PROF_FINAL = 8, W_PROF_COMPACT = 9, W_PROF_PUSH = 10, W_PROF_PFTOK = 11,
              W_PROF_GATHER = 12, W_PROF_BIG = 13, W_PROF_TABLES = 14, W_PROF_RUN = 15, W_PROF_ST_COMP = 16, W_PROF_ST_PUSH = 17,
              W_PROF_ST_BUILD = 18, W_PROF_N = 19;

// ORD: the highest n-gram order compiled in (4 covers the usual models; 6 everything the tables can hold)
// PROF: the phase timers of ctcdec_profile_phases are compiled in (a diagnostics launch; the ~35 timer sites of a frame
//       cost three instructions each even when they do nothing)
template <class Ctx, int BW, int ORD = MAX_CTX + 1, bool PROF = false>
struct WaveDecoder {
  typedef WaveShape<BW> S;
  static constexpr int SLB = S::SLB;
  static constexpr int P = S::P;
  static constexpr int PE = S::PE;

  Ctx& ctx;
  WaveLds& L;
  const UttIO& io;
  const int lane;

  // wave-uniform state (every lane holds the same value)
  int N = 1;
  uint32_t pool_n = 0;
  uint32_t pay_n = 0;    // payload lines used this frame
  uint64_t runmax = 0;   // ascending-sortable key of the best score pushed this frame
  uint64_t kth_key = 0;  // after a pool compaction: key a later candidate has to beat
  uint32_t text_next = 1, emit_next = 1, status = 0;
  uint32_t fflag = 0;    // force_next_break (decoder.py:442)
  uint32_t need = 0;     // some label of this frame closes open words
  uint32_t par = 0;      // which of the two ColdRec buffers holds the current table's records
  // survivors of the NEXT frame, one per lane, fetched a frame ahead
  uint32_t pf_cnt = 0, pf_id = 0;
  double pf_lp = 0.0;
  bool pf_live = false;
  bool run_ok = false;   // the beam table is the output of a full frame of this launch (label_run's precondition)
  unsigned long long t_last = 0;  // (diagnostics: phase ticks are accumulated in global memory)

  CTC_HD WaveDecoder(Ctx& c, WaveLds& l, const UttIO& i) : ctx(c), L(l), io(i), lane(c.lane) {}

  // The scorer tables and the decode parameters are launch constants (the kernel's argument block). Every use fetches
  // what it needs afresh through the execution context -- on the device a scalar load from the argument block behind an
  // opaque copy of its address -- instead of holding ~100 launch constants in scalar registers across the frame loop,
  // from where the compiler spills them to vector-register lanes and pays a v_readlane per use.
  CTC_HD const DeviceTables& tab() const { return ctx.tables(); }
  CTC_HD const DecodeParams& prm() const { return ctx.params(); }
  // ... except the dozen that EVERY candidate pass reads (round 5): a scalar load is a ~70-cycle round trip through the scalar
  // cache that the wave sits out (it issues in order), a pass had ten of them, and a wave's frame is bound by such waits, not
  // by instruction issue (tools/micro/valu_rates.hip). These are fetched once per launch and held -- in scalar registers, or
  // spilled to a vector-register lane, where the reload is one v_readlane.
  struct Hot {
    const PrefixEntry* prefixes;
    uint64_t prefix_mask;
    const HotEntry* hot;
    uint64_t hot_mask;
    double prune_logp, hot_weight, unk;
    uint32_t has_lm, has_trie, prune_history, beam_width;
  } H;
  CTC_HD void load_hot() {
    const DeviceTables& T_ = tab();
    const DecodeParams& P_ = prm();
    H.prefixes = T_.prefixes;
    H.prefix_mask = T_.prefix_mask;
    H.hot = T_.hot;
    H.hot_mask = T_.hot_mask;
    H.prune_logp = P_.beam_prune_logp;
    H.hot_weight = P_.hot_weight;
    H.unk = P_.unk;
    H.has_lm = T_.has_lm ? 1u : 0u;
    H.has_trie = T_.has_trie ? 1u : 0u;
    H.prune_history = P_.prune_history ? 1u : 0u;
    H.beam_width = (uint32_t)P_.beam_width;
  }

  // ---- text nodes -------------------------------------------------------------------------------
  // One node per completed-words prefix (the reference's memo entry, decoder.py:387-396): the raw LM score sum, the
  // hot-word count, the LM state and the ring of the last n_hist word hashes. This kernel keeps them in its own compact
  // layout in the utterance's node arena: with at most three context words (models up to order 4: ORD = 4) a node is 64
  // bytes -- four 16-byte chunks to read and four to write per completed word; longer contexts take 96 of a 128-byte slot.
  //   CTX = 3:  {raw_lm, hw_cnt, ring_cnt | len << 8} {w0, w1, w2, b0} {b1, b2, ring0} {ring1, ring2}
  //   CTX = 5:  {raw_lm, hw_cnt, ring_cnt | len << 8} {w0..w3} {w4, b0, b1, b2} {b3, b4, ring0} {ring1, ring2} {ring3, ring4}
  static constexpr int CTX = (ORD - 1 < 3) ? 3 : (ORD - 1 > MAX_CTX ? MAX_CTX : (ORD - 1 <= 3 ? 3 : MAX_CTX));
  static constexpr int NODE_BYTES = CTX <= 3 ? 64 : 128;
  struct Node {
    double raw;
    uint32_t hw_cnt, ring_cnt;
    LmState st;
    uint64_t ring[MAX_CTX];
  };
  CTC_HD u32x4a* node_ptr(uint32_t idx) const { return (u32x4a*)((char*)io.text_nodes + (size_t)idx * (size_t)NODE_BYTES); }
  CTC_HD void node_load(uint32_t idx, Node& n) const {
    const u32x4a* p = node_ptr(idx);
    const u32x4 c0 = p[0], c1 = p[1], c2 = p[2], c3 = p[3];
    n.raw = bits_f64(q_lo(c0));
    n.hw_cnt = c0[2];
    n.ring_cnt = c0[3] & 0xFFu;
    n.st.len = (int32_t)(c0[3] >> 8);
    if (CTX <= 3) {
      n.st.words[0] = c1[0]; n.st.words[1] = c1[1]; n.st.words[2] = c1[2]; n.st.words[3] = 0; n.st.words[4] = 0;
      n.st.backoff[0] = bits_f32(c1[3]); n.st.backoff[1] = bits_f32(c2[0]); n.st.backoff[2] = bits_f32(c2[1]);
      n.st.backoff[3] = 0.f; n.st.backoff[4] = 0.f;
      n.ring[0] = q_hi(c2); n.ring[1] = q_lo(c3); n.ring[2] = q_hi(c3); n.ring[3] = 0; n.ring[4] = 0;
    } else {
      const u32x4 c4 = p[4], c5 = p[5];
      n.st.words[0] = c1[0]; n.st.words[1] = c1[1]; n.st.words[2] = c1[2]; n.st.words[3] = c1[3]; n.st.words[4] = c2[0];
      n.st.backoff[0] = bits_f32(c2[1]); n.st.backoff[1] = bits_f32(c2[2]); n.st.backoff[2] = bits_f32(c2[3]);
      n.st.backoff[3] = bits_f32(c3[0]); n.st.backoff[4] = bits_f32(c3[1]);
      n.ring[0] = q_hi(c3); n.ring[1] = q_lo(c4); n.ring[2] = q_hi(c4); n.ring[3] = q_lo(c5); n.ring[4] = q_hi(c5);
    }
  }
  CTC_HD void node_store(uint32_t idx, const Node& n) const {
    u32x4a* p = node_ptr(idx);
    const uint64_t rb = f64_bits(n.raw);
    p[0] = mk4((uint32_t)rb, (uint32_t)(rb >> 32), n.hw_cnt, (n.ring_cnt & 0xFFu) | ((uint32_t)n.st.len << 8));
    if (CTX <= 3) {
      p[1] = mk4(n.st.words[0], n.st.words[1], n.st.words[2], f32_bits(n.st.backoff[0]));
      p[2] = mk4(f32_bits(n.st.backoff[1]), f32_bits(n.st.backoff[2]), (uint32_t)n.ring[0], (uint32_t)(n.ring[0] >> 32));
      p[3] = mk4q(n.ring[1], n.ring[2]);
    } else {
      p[1] = mk4(n.st.words[0], n.st.words[1], n.st.words[2], n.st.words[3]);
      p[2] = mk4(n.st.words[4], f32_bits(n.st.backoff[0]), f32_bits(n.st.backoff[1]), f32_bits(n.st.backoff[2]));
      p[3] = mk4(f32_bits(n.st.backoff[3]), f32_bits(n.st.backoff[4]), (uint32_t)n.ring[0], (uint32_t)(n.ring[0] >> 32));
      p[4] = mk4q(n.ring[1], n.ring[2]);
      p[5] = mk4q(n.ring[3], n.ring[4]);
    }
  }

  template <int PHASE>
  CTC_HD void tick() {
    if (PROF && io.prof && lane == 0) {
      unsigned long long now = ctx.clock();
      io.prof[PHASE] += now - t_last;
      t_last = now;
    }
  }

  // ---- small helpers -------------------------------------------------------------------------
  CTC_HD static uint64_t asc_key(double s) {
    const uint64_t u = f64_bits(s + 0.0);  // (-0.0 -> +0.0: they compare equal in Python)
    return u ^ ((uint64_t)((int64_t)u >> 63) | (1ull << 63));  // negative: every bit flipped; else the sign bit set
  }
  CTC_HD static double key_to_score(uint64_t u) {
    return bits_f64(u ^ (~(uint64_t)((int64_t)u >> 63) | (1ull << 63)));
  }
  CTC_HD uint32_t prefix_cnt(uint64_t m) const { return (uint32_t)ctx.popc64(m & ((1ull << lane) - 1ull)); }
  CTC_HD ColdRec* cold_cur() const { return io.cold + (size_t)par * COLD_STRIDE; }
  CTC_HD ColdRec* cold_next() const { return io.cold + (size_t)(par ^ 1u) * COLD_STRIDE; }

  // mode word of a survivor: branch mode | first non-repeating beam << 8 | label flags (TK_*) << 16
  CTC_HD static uint32_t branch_of(uint32_t mode_word, uint32_t c, uint32_t i, uint32_t last_char) {
    if ((mode_word & (TK_BLANK << 16)) || last_char == c) return 0;  // keep prefix (blank / repeat)   decoder.py:452
    const uint32_t mode = mode_word & 0xFFu;
    if (mode == MODE_ALL_B) return BR_BOUNDARY;
    if (mode == MODE_FIRST_B) return i == ((mode_word >> 8) & 0xFFu) ? BR_BOUNDARY : BR_APPEND;
    if (mode == MODE_C) return BR_SPACE;
    return BR_APPEND;
  }

  // ---- survivors of a frame: ids and log-probs one frame ahead, then their label constants -> LDS ----------
  CTC_HD void prefetch(int t) {
    const DecodeParams& P_ = prm();
    pf_live = t < io.T;
    if (!pf_live) return;
    // (every lane loads the same count)
    pf_cnt = io.surv_cnt[t];
    if (lane < WAVE_LAB && lane < P_.max_surv) {
      pf_id = io.surv_id[(size_t)t * P_.max_surv + lane];
      pf_lp = io.surv_lp[(size_t)t * P_.max_surv + lane];
    }
  }
  // label constants of the first WAVE_LAB survivors of the prefetched frame: requested once the passes of the current
  // frame are done with the label block, written to it after the table build. Four lanes per label: lanes 4l .. 4l+2
  // fetch the three 16-byte chunks of TokInfo that ARE the label block's chunks 1..3, lane 4l+3 the label's hot-word view
  // of this call (which goes into the last word of chunk 3) -- four registers across the ranking and the build.
  struct TokRegs {
    u32x4 v;
  };
  CTC_HD void tok_load(TokRegs& r) {
    const DeviceTables& T_ = tab();
    r.v = mk4(0, 0, 0, 0);
    const uint32_t l = (uint32_t)lane >> 2, q = (uint32_t)lane & 3u;
    const uint32_t id = ctx.shfl32(pf_id, (int)l);  // (lanes 0 .. WAVE_LAB-1 hold the ids: l < 16)
    if (!(pf_live && l < pf_cnt)) return;
    if (q < 3u) {
      r.v = ((const u32x4a*)&T_.tok[id])[q];
    } else if (T_.tok_hot) {
      const uint64_t h = *(const uint64_t*)&T_.tok_hot[id];
      r.v[3] = ((uint32_t)h & 0xFFFFu) | ((uint32_t)(h >> 32) ? 0x80000000u : 0u);
    }
  }
  CTC_HD void tok_commit(const TokRegs& r) {
    const uint32_t l = (uint32_t)lane >> 2, q = (uint32_t)lane & 3u;
    const bool mine = pf_live && l < pf_cnt;
    if (mine && q < 3u) L.lab[l * 4 + 1 + q] = r.v;
    ctx.wsync();  // (the hot-word view goes into the last word of chunk 3, behind that chunk's own write)
    if (mine && q == 3u) ((CTC_LDS uint32_t*)L.lab.p)[(l * 4 + 3) * 4 + 3] = r.v[3];
    ctx.wsync();
  }

  // Branch modes of up to WAVE_LAB labels, one per lane (flags TK_BLANK for a lane without a label). For BPE
  // vocabularies the force_next_break flag threads through the labels in iteration order; each label acts on
  // it as identity / clear / set, so the flag a label sees is that of the last non-identity label before it.
  // first/any: index of the first beam that does not repeat the label (N: none).
  CTC_HD uint32_t mode_block(uint32_t fl, uint32_t c, uint32_t lc0, uint32_t f1) {
    const bool blank = (fl & TK_BLANK) != 0;
    if (!tab().is_bpe) {
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
  // One TextNode per completed prefix (the reference's memo entry, decoder.py:387-396); lane = beam. Done on the
  // spot: source node -> n-gram probes -> new node are three dependent global round trips, which the other
  // wavefronts of the SIMD cover. What later phases need of the completion goes to the beam's ColdRec.
  CTC_HD void completions_now() {
    const DeviceTables& T_ = tab();
    const DecodeParams& P_ = prm();
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      if (j * 64 >= N) continue;
      const int i = j * 64 + lane;
      const int ii = i < N ? i : 0;
      const u32x4 k1 = L.hB[ii];
      const bool todo = i < N && (k1[2] >> 16) > 0 && !(k1[3] & M2_COMP);
      const uint64_t m = ctx.ballot(todo);
      if (m == 0ull) continue;
      uint32_t idx = text_next + prefix_cnt(m);
      text_next += (uint32_t)ctx.popc64(m);
      if (!todo) continue;
      if (idx + 1 > io.text_cap) {
        status |= ST_TEXT_OVERFLOW;  // (made wave-wide at the end of the frame)
        idx = io.text_cap - 1;
      }
      ColdRec& cr = cold_cur()[i];
      // the source node's index sits in the beam's column C (no completion yet: see the layout), so the node and the word id
      // are fetched side by side: two dependent round trips for a completion (node, n-gram probes) instead of three
      const uint32_t wid = cr.wid, m2 = k1[3];
      const u32x4 k0 = L.hA[i];
      const uint64_t part_h = q_hi(k0);
      Node sn;
      node_load(L.c32[i * 4 + 2], sn);
      double raw = sn.raw;
      LmState out = sn.st;
      if (T_.has_lm) {
        const float base = lm_base_score<ORD>(T_, sn.st, wid, &out);
        raw = raw + lm_word_score(T_, P_, base, m2, 0.0, false);
      }
      const uint32_t cnt = sn.hw_cnt + ((m2 & M2_HOT_COMPLETE) ? 1u : 0u);
      const double lmhw = raw + P_.hot_weight * (double)cnt;
      const uint32_t rc0 = sn.ring_cnt;
      const uint32_t rc = rc0 + 1 > T_.n_hist ? T_.n_hist : rc0 + 1;
      // history ring, newest first: the closed word, then the source node's (its last entry drops out when the ring is full)
      Node nn;
      nn.raw = raw;
      nn.hw_cnt = cnt;
      nn.ring_cnt = rc;
      nn.st = out;
      nn.ring[0] = part_h;
      nn.ring[1] = 1u < rc ? sn.ring[0] : 0ull;
      nn.ring[2] = 2u < rc ? sn.ring[1] : 0ull;
      nn.ring[3] = 3u < rc ? sn.ring[2] : 0ull;
      nn.ring[4] = 4u < rc ? sn.ring[3] : 0ull;
      const uint64_t hh = wave_hist_fold<CTX>(nn.ring, rc);  // (rc <= n_hist <= CTX)
      node_store(idx, nn);
      L.c64[i * 2 + 1] = text_push(q_lo(k0), part_h);  // the completed text's hash: merge key of the candidates that close the word
      ((u32x4a*)&cr)[0] = mk4q(f64_bits(lmhw), hh);  // {c_lmhw, c_hist_h}: one store
      cr.cnode = idx;
      L.b32[i * 4 + 3] = m2 | M2_COMP;
    }
#ifdef CTC_STORE_PROBE
    tick<W_PROF_COMP>();
    ctx.vm_wait();
    tick<W_PROF_ST_COMP>();
#endif
  }

  // ---- pool ranking --------------------------------------------------------------------------
  // Pool slot k (entries k * 64 + lane) against the whole pool: every lane walks the entries' {score key, history key}
  // words with broadcast LDS reads, four per iteration (the caller padded the list to a multiple of four with keys
  // that beat nobody), counts the entries that beat its own and notes a better one with its history key.
  // An entry below the threshold has a larger key than every entry that passes, so it never counts against one.
  // TIES: equal keys are ordered by arrival (heapq.nlargest is stable) -- only run when the plain walk found some.
  template <bool TIES>
  CTC_HD uint32_t rank_slot(int k, uint32_t n, uint64_t thr_key, bool with_hist, uint32_t want, uint32_t* rank_sum) {
    const uint32_t e = (uint32_t)(k * 64 + lane);
    const bool mine = e < n;
    u32x4 p0 = mk4(~0u, ~0u, 0, 0);
    if (mine) p0 = L.pk[e];
    const bool ok = mine && q_lo(p0) <= thr_key;
    const uint64_t key = ok ? q_lo(p0) : ~0ull, hk = with_hist ? q_hi(p0) : 0ull;
    uint32_t rank = 0, dup = 0;
    if (!TIES) {
      if (with_hist) {
        for (uint32_t j = 0; j < n; j += 4u) {
          const u32x4 r0 = L.pk[j], r1 = L.pk[j + 1u], r2 = L.pk[j + 2u], r3 = L.pk[j + 3u];
          const bool b0 = q_lo(r0) < key, b1 = q_lo(r1) < key, b2 = q_lo(r2) < key, b3 = q_lo(r3) < key;
          rank += (b0 ? 1u : 0u) + (b1 ? 1u : 0u) + (b2 ? 1u : 0u) + (b3 ? 1u : 0u);
          // (bitwise, not short-circuit: the latter became a ladder of exec-mask branches)
          dup |= (uint32_t)((b0 & (q_hi(r0) == hk)) | (b1 & (q_hi(r1) == hk)) | (b2 & (q_hi(r2) == hk)) | (b3 & (q_hi(r3) == hk)));
        }
      } else {
        for (uint32_t j = 0; j < n; j += 4u) {
          const u32x4 r0 = L.pk[j], r1 = L.pk[j + 1u], r2 = L.pk[j + 2u], r3 = L.pk[j + 3u];
          rank += (q_lo(r0) < key ? 1u : 0u) + (q_lo(r1) < key ? 1u : 0u) + (q_lo(r2) < key ? 1u : 0u) + (q_lo(r3) < key ? 1u : 0u);
        }
      }
    } else {
      const uint32_t arr = ok ? ((uint32_t)L.pa[e] & 0xFFFFu) : 0u;
      for (uint32_t j = 0; j < n; ++j) {
        const u32x4 r = L.pk[j];
        const uint32_t xa = (uint32_t)L.pa[j] & 0xFFFFu;
        const bool before = q_lo(r) < key || (q_lo(r) == key && xa < arr);
        rank += before ? 1u : 0u;
        dup |= (before && q_hi(r) == hk) ? 1u : 0u;
      }
    }
    *rank_sum += ctx.wave_sum_u32(ok ? rank : 0u);
    if (ok && rank < want) L.sel[rank] = e | ((with_hist && dup) ? 0u : 0x80000000u);
    return (uint32_t)ctx.popc64(ctx.ballot(ok));
  }

  // Drop the pool entries below the threshold (stable: the rest keep their order). The running maximum only rises
  // during a frame, so entries pushed early are often dead by now; the ranking walk and the compaction are O(entries)
  // per lane and this is O(1).
  CTC_HD void filter_pool(double thr) {
    const uint32_t n = pool_n;
    const uint64_t thr_key = score_sort_key(thr);
    u32x4 g0[PE];
    uint64_t g1[PE];
    uint32_t dst[PE];
    bool ok[PE];
    uint32_t kept = 0;
CTC_UNROLL
    for (int k = 0; k < PE; ++k) {
      const uint32_t e = (uint32_t)(k * 64 + lane);
      ok[k] = false;
      dst[k] = 0;
      g0[k] = mk4(0, 0, 0, 0);
      g1[k] = 0;
      if ((uint32_t)(k * 64) >= n) continue;
      if (e < n) {
        g0[k] = L.pk[e];
        g1[k] = L.pa[e];
        ok[k] = q_lo(g0[k]) <= thr_key;
      }
      const uint64_t m = ctx.ballot(ok[k]);
      dst[k] = kept + prefix_cnt(m);
      kept += (uint32_t)ctx.popc64(m);
    }
    if (kept == n) return;  // (uniform)
    ctx.wsync();
CTC_UNROLL
    for (int k = 0; k < PE; ++k) {
      if (ok[k]) {
        L.pk[dst[k]] = g0[k];
        L.pa[dst[k]] = g1[k];
      }
    }
    pool_n = kept;
    ctx.wsync();
  }

  // Ranks the pool entries with score >= thr by (score desc, arrival asc); L.sel[r] = pool index of rank r
  // (bit 31: kept by the history prune) for r < min(count, beam_width). Returns the count.
  // (heapq.nlargest + _prune_history, decoder.py:165-167, 244-257)
  CTC_HD uint32_t rank_pool(double thr, bool with_hist) {
    if (pool_n > 64u) filter_pool(thr);  // (heavy frames: usually back to one entry per lane)
    const uint32_t n = pool_n;
    const uint32_t want = H.beam_width;
    const uint64_t thr_key = score_sort_key(thr);
    // pad the list to a multiple of four (n <= P, P % 4 == 0: the slots exist)
    if (lane < 3 && (n & 3u) != 0u && n + (uint32_t)lane < ((n + 3u) & ~3u)) L.pk[n + (uint32_t)lane] = mk4q(~0ull, 0ull);
    ctx.wsync();
    uint32_t np = 0, rank_sum = 0;
CTC_UNROLL
    for (int k = 0; k < PE; ++k)
      if ((uint32_t)(k * 64) < n) np += rank_slot<false>(k, n, thr_key, with_hist, want, &rank_sum);
    // Without equal scores the ranks of the passing entries are a permutation of 0 .. np-1; an equal pair shares
    // a rank and makes their sum smaller -- one wave sum per slot instead of an equality count per entry.
    if (rank_sum != np * (np - 1u) / 2u) {
      ctx.wsync();
      uint32_t unused = 0;
CTC_UNROLL
      for (int k = 0; k < PE; ++k)
        if ((uint32_t)(k * 64) < n) rank_slot<true>(k, n, thr_key, with_hist, want, &unused);
    }
    ctx.wsync();
    return np;
  }

  // Make room in the pool: keep (a little more than) its best beam_width entries -- exact, pruning is monotone (SURVEY
  // App. G). Nothing has to be RANKED for that: a bisection on the score key finds a cut K with beam_width <= #(key <= K)
  // <= beam_width + 8 (a dozen ballots instead of a walk of every entry against every other), the entries behind the cut
  // are dropped in place, and from then on only a candidate that beats K can still matter: later candidates arrive
  // later, so an equal score ranks behind the >= beam_width entries kept here. (Many exactly equal scores around the
  // cut: the full ranking decides.)
  CTC_HD void compact_pool() {
    filter_pool(key_to_score(runmax) + H.prune_logp);
    const uint32_t n = pool_n, want = H.beam_width;
    if (n <= want) {
      tick<W_PROF_COMPACT>();
      return;
    }
    uint64_t key[PE];
    uint64_t worst = 0, best_inv = 0;
CTC_UNROLL
    for (int k = 0; k < PE; ++k) {
      const uint32_t e = (uint32_t)(k * 64 + lane);
      key[k] = ~0ull;
      if ((uint32_t)(k * 64) < n && e < n) {
        key[k] = q_lo(L.pk[e]);
        if (key[k] > worst) worst = key[k];
        if (~key[k] > best_inv) best_inv = ~key[k];
      }
    }
    uint64_t lo = ~ctx.wave_max_u64(best_inv), hi = ctx.wave_max_u64(worst);  // #(key <= hi) = n > want
    uint32_t cnt = n;
    while (lo < hi) {
      const uint64_t mid = lo + ((hi - lo) >> 1);
      uint32_t c = 0;
CTC_UNROLL
      for (int k = 0; k < PE; ++k)
        if ((uint32_t)(k * 64) < n) c += (uint32_t)ctx.popc64(ctx.ballot(key[k] <= mid));
      if (c >= want) {
        hi = mid;
        cnt = c;
        if (c <= want + 8u) break;
      } else {
        lo = mid + 1;
      }
    }
    if (cnt > want + 8u) {  // (a pile of equal scores at the cut)
      compact_pool_ranked();
      return;
    }
    filter_pool(key_to_score(~hi));
    kth_key = ~hi;  // (score key = ~ascending key)
    tick<W_PROF_COMPACT>();
  }
  CTC_HD void compact_pool_ranked() {
    const double mx = key_to_score(runmax);
    uint32_t n = rank_pool(mx + H.prune_logp, false);
    if (n > H.beam_width) n = H.beam_width;
    u32x4 g0[SLB];
    uint64_t g1[SLB];
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const uint32_t r = (uint32_t)(j * 64 + lane);
      g0[j] = mk4(0, 0, 0, 0);
      g1[j] = 0;
      if (r < n) {
        const uint32_t e = L.sel[r] & 0x7FFFFFFFu;
        g0[j] = L.pk[e];
        g1[j] = L.pa[e];
      }
    }
    ctx.wsync();
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const uint32_t r = (uint32_t)(j * 64 + lane);
      if (r < n) {
        L.pk[r] = g0[j];
        L.pa[r] = g1[j];
      }
    }
    pool_n = n;
    if (n >= H.beam_width) {
      const uint32_t r = n - 1;
      uint64_t k = 0;
CTC_UNROLL
      for (int j = 0; j < SLB; ++j)
        if ((int)(r >> 6) == j) k = ctx.bcast64(~q_lo(g0[j]), (int)(r & 63u));  // (score key = ~ascending key)
      kth_key = k;
    }
    ctx.wsync();
    tick<W_PROF_COMPACT>();
  }

  // ---- candidates ------------------------------------------------------------------------------------
  // A pass takes whole labels: floor(64 / N) of them, candidate v = lane. With more than 64 live beams one label is
  // two half passes that are matched together (pass_big).
  struct Cand {
    bool valid, is_rep, want_p, want_h;
    uint32_t bi, ls, ll, lid, mw, br, pl0, m2_0, len_raw, tslot;
    uint64_t kp, ck, pp_key, ph_key;
    uint32_t pp_wid, pp_fl, ph_min, ph_cmp;
    double lg;
    double wd;    // of the candidate's beam (ColdRec): the pending completion's lm + hot-word score when the label closes the
    uint64_t wh;  // open word, else the open word's partial score; and the history hash that goes with the branch
  };

  // branch, merge key and summed logit of candidate (label l of the staged block = survivor s, beam i);
  // FULL: also the first probe of the prefix / hot-word table of an appended partial word and the request for the chunk
  // of the beam's ColdRec its score will need.
  template <bool FULL>
  CTC_HD void gen(Cand& c, bool valid, uint32_t l, uint32_t s, uint32_t i) {
    // Straight-line: a lane without a candidate computes on label 0 / beam 0 (its fields are only looked at behind
    // `valid`), and every branch of the reference's if-ladder (decoder.py:452-534) is a select.
    const uint32_t ll = valid ? l : 0u, ii = valid ? i : 0u;
    const u32x4 sv = L.lab[ll * 4];
    const u32x4 la = L.lab[ll * 4 + 1], lb = L.lab[ll * 4 + 2];
    const u32x4 k0 = L.hA[ii], k1 = L.hB[ii];
    c.valid = valid;
    c.is_rep = false;
    c.bi = i;
    c.ls = s;
    c.ll = l;
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
    const bool closing_word = closes && pl > 0;
    // merge key parts: the text (the completed one when the open word closes) and the new partial word
    // (the completed text's hash sits in the beam's column C since its completion was made: completions_now)
    const uint64_t kt = closing_word ? L.c64[ii * 2 + 1] : q_lo(k0);
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
    c.wd = 0.0;
    c.wh = 0;
    if (FULL) {  // first probe of the prefix / hot-word table of an appended partial word
      const bool probe = valid && app && p != 0;
      c.tslot = (uint32_t)table_slot(p);
      c.want_p = probe && (k1[3] & PF_ON_TABLE) && H.prefixes;
      c.want_h = probe && (k1[3] & M2_HOT_ON) && H.hot;
      if (c.want_p) {
        const PrefixEntry& g = H.prefixes[c.tslot & H.prefix_mask];
        c.pp_key = g.key;
        c.pp_wid = g.word_id;
        c.pp_fl = g.flags;
      }
      if (c.want_h) {
        const HotEntry& g = H.hot[c.tslot & H.hot_mask];
        c.ph_key = g.key;
        c.ph_min = g.min_len;
        c.ph_cmp = g.complete;
      }
      // one 16-byte load: chunk 0 {completion's score, its history hash} or chunk 1 {partial score, the text's history hash}
      const u32x4 cw = ((const u32x4a*)(cold_cur() + ii))[closing_word ? 0 : 1];
      c.wd = bits_f64(q_lo(cw));
      c.wh = q_hi(cw);
    }
    // (a real finaliser: the hashes of one-character strings differ in their low bits only, and the match tag drops seven bits)
    c.ck = fin64(kt ^ rotl64(p, 17) ^ ((uint64_t)(l + 1u) << 56));
    c.lg = bits_f64(q_lo(k1)) + bits_f64(pack64(sv[2], sv[3]));
  }

  // wave-wide hash match on 64-bit keys (valid lanes only): rep = smallest candidate index with the same key.
  // The smallest candidate of the largest tag owns a slot; the others of its key join it, the rest move to their
  // next slot (other key bits, then linear). A candidates per lane (v = j * 64 + lane), all inserted in the same
  // round (the members of a key must see the same slot history); 128 slots, cleared by the caller. (A settled owner
  // may later be displaced by a larger tag: its members have read their representative by then.)
  template <int A>
  CTC_HD void match(const bool* valid, const uint64_t* ck, uint32_t* rep) {
    constexpr uint32_t MASK = (uint32_t)(S::TS - 1);
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
  CTC_HD void clear_table() { ((CTC_LDS u32x4*)L.table.p)[lane] = mk4(0, 0, 0, 0); }  // 128 slots = 64 x 16 bytes

  // member masks of the representatives that have members (bit u: candidate u joins), read off the matched
  // representatives with ballots: one round per group of two or more
  template <int A>
  CTC_HD void group_masks(const bool* valid, const uint32_t* rep, uint64_t* mlo, uint64_t* mhi) {
    bool pend[A];
CTC_UNROLL
    for (int h = 0; h < A; ++h) {
      pend[h] = valid[h] && rep[h] != (uint32_t)(h * 64 + lane);
      mlo[h] = mhi[h] = 0ull;
    }
    for (;;) {
      uint64_t b[A];
      uint64_t any = 0ull;
CTC_UNROLL
      for (int h = 0; h < A; ++h) {
        b[h] = ctx.ballot(pend[h]);
        any |= b[h];
      }
      if (any == 0ull) break;
      uint32_t u = 0;
      if (b[0]) u = ctx.bcast32(rep[0], ctx.ctz64(b[0]));
      else if (A > 1) u = ctx.bcast32(rep[A - 1], ctx.ctz64(b[A - 1]));
      uint64_t m[A];
CTC_UNROLL
      for (int h = 0; h < A; ++h) {
        const bool in = pend[h] && rep[h] == u;
        m[h] = ctx.ballot(in);
        pend[h] = pend[h] && !in;
      }
      if ((uint32_t)lane == (u & 63u)) {
CTC_UNROLL
        for (int h = 0; h < A; ++h) {
          if ((u >> 6) == (uint32_t)h) {
            mlo[h] = m[0];
            mhi[h] = A > 1 ? m[A - 1] : 0ull;
          }
        }
      }
    }
  }

  // what the prefix / hot-word tables say about an appended partial word (the first probes were issued by gen)
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
        uint64_t sp = c.tslot & H.prefix_mask;
        uint64_t ek = c.pp_key;
        uint32_t nw = c.pp_wid, pf = c.pp_fl;
        while (ek != key && ek != 0) {
          sp = (sp + 1) & H.prefix_mask;
          const PrefixEntry& g = H.prefixes[sp];
          ek = g.key;
          nw = g.word_id;
          pf = g.flags;
        }
        t.on = ek == key;
        t.nw = nw;
        t.pf = pf;
      }
      if (c.want_h) {
        uint64_t sh = c.tslot & H.hot_mask;
        uint64_t ek = c.ph_key;
        uint32_t hmin = c.ph_min, hcomp = c.ph_cmp;
        while (ek != key && ek != 0) {
          sh = (sh + 1) & H.hot_mask;
          const HotEntry& g = H.hot[sh];
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

  // partial_score (beam_core.h: language_model.py:141-150, 326-336; decoder.py:363-367, 397-409) as selects, for the
  // single-model kernel: same operations in the same order.
  CTC_HD double partial_score_sel(uint32_t pf_flags, uint32_t hot_min_len, uint32_t plen) const {
    const double pl = (double)plen;
    double s = 0.0;
    if (H.has_lm) {  // (uniform)
      const bool on_trie = H.has_trie && (pf_flags & PF_UNI_PREFIX);
      s = H.unk * (on_trie ? 0.0 : 1.0);
      s = plen > 6 ? div_by_6(s * pl) : s;
    }
    if (hot_min_len > 0) s = H.hot_weight * pl / (double)hot_min_len;  // (a real division: behind a branch, ~30 instructions)
    return s;
  }

  // total_score (beam_core.h; decoder.py:363-367, 398-409, 420) on the held has_lm flag
  CTC_HD double total_score_h(double logit, double lm_hw, double ps, uint32_t plen) const {
    if (!H.has_lm) return logit + lm_hw + ps;
    double s = lm_hw;
    if (plen > 0) s = s + ps;
    return logit + s;
  }

  // one pool entry per flagged lane, in lane order; its payload goes to the frame's next free payload line
  CTC_HD void pool_put(bool put, u32x4 key, uint32_t arrival, uint32_t donor, double logit, uint64_t part_h, uint32_t plen,
                       uint32_t wid, uint32_t m2) {
    const uint64_t m = ctx.ballot(put);
    const uint32_t off = prefix_cnt(m);
    const uint32_t k = pool_n + off, q = pay_n + off;
    pool_n += (uint32_t)ctx.popc64(m);
    pay_n += (uint32_t)ctx.popc64(m);
    if (put) {
#ifdef CTC_SIM_DEBUG
      if (k >= (uint32_t)P) { fprintf(stderr, "pool_put: k=%u P=%d\n", k, P); abort(); }
      if (q >= 65536u) { fprintf(stderr, "pool_put: payload line %u\n", q); abort(); }
#endif
      L.pk[k] = key;
      L.pa[k] = pack64(arrival | (q << 16), donor);
      u32x4a* dst = (u32x4a*)&io.pay[q];
      dst[0] = mk4q(f64_bits(logit), part_h);
      dst[1] = mk4(plen, wid, m2, 0u);
    }
  }

  // score the representatives of a pass (decoder.py:346-424) and push what can still matter into the pool;
  // imax / dbr: the group's donor (last arrival: its beam, its branch)
  CTC_HD void score_push(const Cand& c, const TabView& t, uint32_t imax, uint32_t dbr) {
    // Straight-line (selects, LDS reads at safe indices). Lanes that represent nothing compute on beam 0 / label 0
    // and are masked at the end.
    const bool rep = c.is_rep;
    const uint32_t i = rep ? c.bi : 0u, ll = rep ? c.ll : 0u;
    const uint32_t b = c.br;
    const u32x4 lb = L.lab[ll * 4 + 2], lc = L.lab[ll * 4 + 3];
    const double own_lmhw = bits_f64(L.c64[i * 2]);
    const bool is0 = b == 0, isB = b == BR_BOUNDARY, isA = b == BR_APPEND;  // else: space
    // boundary: a new word starts with the clean label (or, for a bare boundary mark, nothing yet)
    const uint32_t len_clean = lb[3];
    const uint32_t hminB = lc[3] & 0xFFFFu, hcompB = lc[3] >> 31;
    const bool bw = isB && len_clean > 0;
    const uint32_t m2B = len_clean > 0 ? ((lc[1] & (PF_PARTIAL_MASK | PF_ON_TABLE)) | (hminB ? M2_HOT_ON : 0u) |
                                          (hcompB ? M2_HOT_COMPLETE : 0u) | (hminB << 8))
                                       : EMPTY_PARTIAL_M2;
    // append: what the prefix / hot-word tables say about the longer partial word
    const uint32_t a_pf = t.on ? t.pf : 0u, a_hmin = t.hon ? t.hmin : 0u;
    const uint32_t m2A = (t.on ? (PF_ON_TABLE | (t.pf & PF_PARTIAL_MASK)) : 0u) | (t.hon ? M2_HOT_ON : 0u) |
                         ((t.hon && t.hcomp) ? M2_HOT_COMPLETE : 0u) | (a_hmin << 8);
    const uint32_t q_pl = is0 ? c.pl0 : (isB ? len_clean : (isA ? c.pl0 + c.len_raw : 0u));
    const uint32_t q_m2 = is0 ? (c.m2_0 & ~M2_COMP) : (isB ? m2B : (isA ? m2A : EMPTY_PARTIAL_M2));
    // (a blank / repeat keeps its beam's open word and with it the word's id: the table build reads it from that beam's ColdRec
    //  -- bit 31 + the beam -- instead of every candidate lane fetching it here)
    const uint32_t q_wid = is0 ? (0x80000000u | c.bi) : (isB ? (len_clean > 0 ? lc[2] : 0u) : (isA ? (t.on ? t.nw : 0u) : 0u));
    const double ps_new = partial_score_sel(isB ? lc[1] : a_pf, isB ? hminB : a_hmin, q_pl);
    const double q_ps = is0 ? c.wd : ((bw || isA) ? ps_new : 0.0);  // (blank / repeat: the open word's score as it is)
    const double lmhw = (!is0 && !isA && c.pl0 > 0) ? c.wd : own_lmhw;  // boundary / space close the open word
    const double sc = total_score_h(c.lg, lmhw, q_ps, q_pl);
    const double score = rep ? sc : 0.0;
    const uint64_t my_key = rep ? asc_key(sc) : 0ull;
    const uint64_t pass_key = ctx.wave_max_u64(my_key);
    if (pass_key > runmax) runmax = pass_key;
    const double thr = key_to_score(runmax) + H.prune_logp;
    tick<W_PROF_SCORE>();
    // (history, partial, last_char) folded to 64 bits (decoder.py:250-254): equality of the folds stands in
    // for equality of the triple (its members are 61/64-bit string hashes already)
    uint64_t hk = 0;
    if (H.prune_history) hk = c.wh ^ rotl64(c.kp, 19) ^ ((uint64_t)(c.lid + 1u) << 40);  // (compared for equality only)
    const u32x4 e0 = mk4q(~my_key, hk);  // (score_sort_key(score) == ~asc_key(score); lanes that push are representatives)
    const uint32_t blank = (c.mw >> 16) & TK_BLANK;
    const uint32_t arrival = c.ls * (uint32_t)N + c.bi;
    const uint32_t donor = imax | (c.lid << 8) | (blank ? (1u << 29) : 0u) | (dbr << 30);
    bool push = rep && score >= thr && my_key > kth_key;
    uint32_t cnt = (uint32_t)ctx.popc64(ctx.ballot(push));
    if (pool_n + cnt > (uint32_t)P) filter_pool(thr);  // room, the cheap way: entries the risen threshold has left behind
    while (pool_n + cnt > (uint32_t)P) {
      // The pool holds the best beam_width so far plus room for two dozen more: make room (exact: pruning is monotone),
      // if need be in several parts.
      compact_pool();
      push = push && my_key > kth_key;
      const uint64_t m = ctx.ballot(push);
      cnt = (uint32_t)ctx.popc64(m);
      if (pool_n + cnt > (uint32_t)P) {
        const bool part = push && prefix_cnt(m) < (uint32_t)P - pool_n;
        pool_put(part, e0, arrival, donor, c.lg, c.kp, q_pl, q_wid, q_m2);
        ctx.wsync();
        push = push && !part;
        cnt = (uint32_t)ctx.popc64(ctx.ballot(push));
      }
    }
    pool_put(push, e0, arrival, donor, c.lg, c.kp, q_pl, q_wid, q_m2);
    ctx.wsync();
    tick<W_PROF_PUSH>();
#ifdef CTC_STORE_PROBE
    ctx.vm_wait();
    tick<W_PROF_ST_PUSH>();
#endif
  }

  // One pass: the candidates of the labels [l0, l1) of the staged block (survivors base + l), N <= 64 beams,
  // candidate v = lane
  CTC_HD void pass1(uint32_t base, uint32_t l0, uint32_t l1) {
    const uint32_t Nn = (uint32_t)N;
    const uint32_t Q = (l1 - l0) * Nn;
    const uint32_t rcpN = WAVE_DIV.rcp[Nn <= 64u ? Nn : 0u];  // v / N == (v * rcpN) >> 16 for v * N < 65536 (N <= 64 here)
    const uint32_t v = (uint32_t)lane;
    clear_table();
    const uint32_t sl = (v * rcpN) >> 16;
    const uint32_t l = l0 + sl;
    Cand c;
    gen<true>(c, v < Q, l, base + l, v - sl * Nn);
    ctx.wsync();
    tick<W_PROF_GEN>();
    uint32_t rep;
    match<1>(&c.valid, &c.ck, &rep);
    // members announce themselves to their representative (the mask words take the place of the table's first half)
    *(CTC_LDS uint64_t*)&L.gmask[lane * 2] = 0ull;
    ctx.wsync();
    if (c.valid && rep != v) ctx.lds_or_u32(&L.gmask[rep * 2 + (v >> 5)], 1u << (v & 31u));
    c.is_rep = c.valid && rep == v;
    ctx.wsync();
    tick<W_PROF_MATCH>();
    // ---- fold the group's logits in ascending beam rank (decoder.py:217-223); donor = last arrival
    // (before the table probes and the ColdRec chunk that gen requested are looked at: the fold needs neither, and whatever it
    //  costs is taken off the wait for them)
    uint32_t imax = c.bi, dbr = c.br;
    {
      uint32_t g0 = 0, g1 = 0;
      bool more = false;
      double lp = 0.0;
      if (c.is_rep) {
        g0 = L.gmask[v * 2];
        g1 = L.gmask[v * 2 + 1];
        uint32_t top = 0xFFFFFFFFu;
        if (g0) top = (uint32_t)(31 - ctx.clz32(g0));
        if (g1) top = (uint32_t)(63 - ctx.clz32(g1));
        if (top != 0xFFFFFFFFu) {
          imax = c.bi + (top - v);  // members share the label: consecutive beam indices
          dbr = branch_of(c.mw, c.lid, imax, L.b32[imax * 4 + 2] & 0xFFFFu);
          const u32x4 sv = L.lab[c.ll * 4];
          lp = bits_f64(pack64(sv[2], sv[3]));
          more = true;
        }
      }
      while (ctx.ballot(more) != 0ull) {
        more = false;
        if (c.is_rep) {
          uint32_t mbit = 0xFFFFFFFFu;
          if (g1) mbit = (uint32_t)(32 + ctx.ctz32(g1));
          if (g0) mbit = (uint32_t)ctx.ctz32(g0);
          if (mbit != 0xFFFFFFFFu) {
            if (mbit < 32u) g0 &= g0 - 1u;
            else g1 &= g1 - 1u;
            // (the member's summed logit: its beam's + the label's, as the member itself computed it)
            c.lg = lse2(c.lg, bits_f64(L.b64[(c.bi + (mbit - v)) * 2]) + lp);
            more = (g0 | g1) != 0u;
          }
        }
      }
    }
    tick<W_PROF_FOLD>();
    const TabView t = resolve_tables(c);
    tick<W_PROF_TABLES>();
    score_push(c, t, imax, dbr);
  }

  // One label with more than 64 live beams (rare): candidate v = beam index, two half passes matched together.
  // Keys of both halves stay in registers; groups are read off the representatives with ballots; everything else
  // of a half is (re)generated when the half is scored, so that no second candidate lives in registers.
  CTC_HD void pass_big(uint32_t base, uint32_t l) {
    const uint32_t Nn = (uint32_t)N;
    const u32x4 sv = L.lab[l * 4];
    const double lp = bits_f64(pack64(sv[2], sv[3]));
    bool valid[2];
    uint64_t ck[2];
    double lg[2];
    uint32_t rep[2];
    clear_table();
CTC_UNROLL
    for (int h = 0; h < 2; ++h) {
      const uint32_t v = (uint32_t)(h * 64 + lane);
      Cand c;
      gen<false>(c, v < Nn, l, base + l, v);
      valid[h] = c.valid;
      ck[h] = c.ck;
      lg[h] = c.lg;
    }
    ctx.wsync();
    match<2>(valid, ck, rep);
    uint64_t mlo[2], mhi[2];
    group_masks<2>(valid, rep, mlo, mhi);
    // fold in ascending beam rank (decoder.py:217-223); a member's summed logit is its beam's logit + the label's
    uint32_t imax[2];
    bool is_rep[2];
    bool more = false;
CTC_UNROLL
    for (int h = 0; h < 2; ++h) {
      const uint32_t v = (uint32_t)(h * 64 + lane);
      is_rep[h] = valid[h] && rep[h] == v;
      imax[h] = v;
      if (is_rep[h]) {
        if (mlo[h]) imax[h] = (uint32_t)(63 - ctx.clz64(mlo[h]));
        if (mhi[h]) imax[h] = (uint32_t)(127 - ctx.clz64(mhi[h]));
      }
      more = more || (is_rep[h] && (mlo[h] | mhi[h]) != 0ull);
    }
    while (ctx.ballot(more) != 0ull) {
      more = false;
CTC_UNROLL
      for (int h = 0; h < 2; ++h) {
        if (is_rep[h] && (mlo[h] | mhi[h]) != 0ull) {
          uint32_t mbit;
          if (mlo[h]) {
            mbit = (uint32_t)ctx.ctz64(mlo[h]);
            mlo[h] &= mlo[h] - 1ull;
          } else {
            mbit = (uint32_t)(64 + ctx.ctz64(mhi[h]));
            mhi[h] &= mhi[h] - 1ull;
          }
          lg[h] = lse2(lg[h], bits_f64(L.b64[mbit * 2]) + lp);
          more = more || (mlo[h] | mhi[h]) != 0ull;
        }
      }
    }
    tick<W_PROF_BIG>();
CTC_UNROLL
    for (int h = 0; h < 2; ++h) {
      const uint32_t v = (uint32_t)(h * 64 + lane);
      Cand c;
      gen<true>(c, v < Nn, l, base + l, v);
      c.is_rep = is_rep[h];
      c.lg = lg[h];
      const uint32_t di = is_rep[h] ? imax[h] : 0u;
      const uint32_t dbr = branch_of(sv[1], sv[0], di, L.b32[di * 4 + 2] & 0xFFFFu);
      const TabView t = resolve_tables(c);
      score_push(c, t, imax[h], dbr);
    }
  }

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
    const DeviceTables& T_ = tab();
    const DecodeParams& P_ = prm();
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
        const u32x4 k1 = L.hB[i];
        lg[j] = bits_f64(q_lo(k1));
        rest[j] = bits_f64(L.c64[i * 2]);
        pl[j] = k1[2] >> 16;
        if (pl[j] > 0) psc[j] = cold_cur()[i].pscore;  // (only an open word has a partial score; total_score looks at it then)
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
        sc[j] = total_score_h(nl[j], rest[j], psc[j], pl[j]);
        if (live[j]) L.scr[j * 64 + lane] = sc[j];
      }
      ctx.wsync();
      const double thr = L.scr[0] + H.prune_logp;
      bool bad = false;
CTC_UNROLL
      for (int j = 0; j < SLB; ++j) {
        const int i = j * 64 + lane;
        if (live[j]) {
          if (i + 1 < N) bad = bad || !(sc[j] >= L.scr[i + 1]);
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
          const uint32_t id = io.surv_id[(size_t)f * P_.max_surv];
          w_lp = io.surv_lp[(size_t)f * P_.max_surv];
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
        L.b64[i * 2] = f64_bits(lg[j]);
        if (!lab_is_blank) cold_cur()[i].pend = io.first_frame + tt;  // end frame of the open word: last held frame + 1
      }
    }
    ctx.wsync();
    prefetch(tt);
    TokRegs tr;
    tok_load(tr);
    tok_commit(tr);
    tick<W_PROF_RUN>();
    return tt;
  }

  // ---- one frame ---------------------------------------------------------------------------------
  CTC_HD int step(int t) {
    const int frame = io.first_frame + t;
    const uint32_t ns = ctx.uni32(pf_cnt);
    if (run_ok && ns == 1u && N > 0 && !prm().no_label_runs) {
      const uint32_t lab = ctx.bcast32(pf_id, 0);
      bool same = true;
CTC_UNROLL
      for (int j = 0; j < SLB; ++j) {
        const int i = j * 64 + lane;
        if (i < N) same = same && (L.b32[i * 4 + 2] & 0xFFFFu) == lab;
      }
      if (ctx.ballot(!same) == 0ull) {
        const int t2 = label_run(t, lab, (L.lab[3][0] & TK_BLANK) != 0u);
        if (t2 > t) return t2;
      }
    }
    pool_n = 0;
    pay_n = 0;
    runmax = asc_key(-INFINITY);
    kth_key = 0;
    need = 0;
#ifdef CTC_WAVE_TRACE
    if (lane < N) {
      const u32x4 t0 = L.hA[lane], t1 = L.hB[lane], t2 = L.hC[lane];
      printf("TB f=%d N=%d i=%d text=%llx part=%llx logit=%.6f meta1=%x m2=%x ctext=%llx\n", frame, N, lane, (unsigned long long)q_lo(t0),
             (unsigned long long)q_hi(t0), bits_f64(q_lo(t1)), t1[2], t1[3], (unsigned long long)q_hi(t2));
    }
#endif
    // this frame's survivors (block 0: ids / log-probs in the prefetch registers, label constants already in LDS),
    // then the request for the next frame's
    const uint32_t id0 = pf_id;
    const double lp0 = pf_lp;
    prefetch(t + 1);  // lands while this frame's candidates are processed
    // is there a beam that does not end in beam 0's label, and which is the first?
    uint32_t lc[SLB];
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const int i = j * 64 + lane;
      lc[j] = i < N ? (L.b32[i * 4 + 2] & 0xFFFFu) : 0u;
    }
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
    bool comp_done = false;
    // survivors in blocks of WAVE_LAB labels: ids, log-probs, branch modes and label constants -> LDS, then the
    // candidates of the block in passes of whole labels
    for (uint32_t base = 0; base < ns; base += (uint32_t)WAVE_LAB) {
      const uint32_t s = base + (uint32_t)lane;
      const bool mine = s < ns && lane < WAVE_LAB;
      uint32_t id = 0, fl = TK_BLANK;
      double lp = 0.0;
      if (base == 0) {
        if (mine) {
          id = id0;
          lp = lp0;
          fl = L.lab[lane * 4 + 3][0];
        }
      } else {
        // (more than WAVE_LAB survivors in one frame: fetched on the spot)
        ctx.wsync();  // the previous block's passes are done with the label block
        if (mine) {
          id = io.surv_id[(size_t)t * prm().max_surv + s];
          lp = io.surv_lp[(size_t)t * prm().max_surv + s];
          const TokInfo& g = tab().tok[id];
          fl = g.flags;
          const uint32_t hot = tab().tok_hot ? ((tab().tok_hot[id].min_len & 0xFFFFu) | (tab().tok_hot[id].complete ? 0x80000000u : 0u)) : 0u;
          L.lab[lane * 4 + 1] = mk4q(g.h_raw, g.pow_raw);
          L.lab[lane * 4 + 2] = mk4((uint32_t)g.h_clean, (uint32_t)(g.h_clean >> 32), g.len_raw, g.len_clean);
          L.lab[lane * 4 + 3] = mk4(g.flags, g.start_flags, g.start_word_id, hot);
        }
      }
      const uint32_t mwd = mode_block(fl, id, lc0, f1);
      if (mine) L.lab[lane * 4] = mk4(id, mwd, (uint32_t)f64_bits(lp), (uint32_t)(f64_bits(lp) >> 32));
      if (base == 0) tick<W_PROF_LOAD>();
      if (need && !comp_done) {
        completions_now();
        comp_done = true;
        tick<W_PROF_COMP>();
      }
      ctx.wsync();
      const uint32_t nb = ns - base < (uint32_t)WAVE_LAB ? ns - base : (uint32_t)WAVE_LAB;
      if (N <= 64) {
        // passes of whole labels (<= 64 candidates)
        const uint32_t per = WAVE_DIV.per[N > 0 ? N : 0];  // whole labels per pass: max(1, 64 / N)
        for (uint32_t l0 = 0; l0 < nb; l0 += per) pass1(base, l0, l0 + per < nb ? l0 + per : nb);
      } else if (SLB > 1) {
        for (uint32_t l = 0; l < nb; ++l) pass_big(base, l);
      }
    }
    // the label constants of the next frame's survivors: requested now that the passes are done with the label
    // block, written to it after the build
    TokRegs tr;
    tok_load(tr);
    tick<W_PROF_PFTOK>();
    const double thr = key_to_score(runmax) + H.prune_logp;
    const bool hist = H.prune_history != 0;
#ifdef CTC_WAVE_TRACE
    if ((uint32_t)lane < pool_n) {
      const u32x4 t0 = L.pk[lane];
      const uint64_t t2 = L.pa[lane];
      printf("TR f=%d N=%d ns=%u pool=%u e=%d key=%llx arr=%x don=%x\n", frame, N, ns, pool_n, lane, (unsigned long long)q_lo(t0), (uint32_t)t2, (uint32_t)(t2 >> 32));
    }
#endif
    uint32_t n = rank_pool(thr, hist);
#ifdef CTC_STATS
    if (lane == 0) fprintf(stderr, "ST %d %u %u %u\n", N, ns, pool_n, n);
#endif
    if (n > H.beam_width) n = H.beam_width;
    tick<W_PROF_RANK>();
    // nothing passed the threshold: only possible with non-finite scores (NaN rows) or a positive
    // beam_prune_logp; the reference then dies on max([]) (decoder.py:545) -- reported through the status
    if (n == 0) status |= ST_NO_BEAMS;
    // Everything requested so far -- the next frame's survivors and their label constants -- is in its registers before the
    // build issues its stores: the accesses of a wave complete in issue order, so a load waited for behind a store waits for
    // that store's acknowledgement too (0.3 - 0.6 us), and the first thing the next frame does is look at its survivors.
    ctx.vm_wait();
    if (SLB == 1 || n <= 64u) {
      tok_commit(tr);  // (the label block is idle from here on; build_big parks records in it)
      build1(frame, n);
    } else {
      build_big(frame, n);
      tok_commit(tr);
    }
    run_ok = true;
    return t + 1;
  }

  // ---- next beam table from the ranked pool (decoder.py:548-554) -----------------------------------
  struct Rec {
    u32x4 o0, o1, o2;
  };
  // the columns of one rank (w = its L.sel word, d = its place in the new table); also writes the beam's ColdRec
  // and, for a label that is not a blank / repeat, its emission node. All lanes call.
  CTC_HD void gather(int frame, uint32_t w, bool kept, uint32_t d, Rec& o) {
    o.o0 = o.o1 = o.o2 = mk4(0, 0, 0, 0);
    // the payload is the donor's (the last-arriving duplicate, decoder.py:221-223), its branch included
    uint64_t a2 = 0;
    if (kept) a2 = L.pa[w & 0x7FFFFFFFu];
    const uint32_t don = (uint32_t)(a2 >> 32);
    const uint32_t b = don >> 30, i = don & 0xFFu, c = (don >> 8) & 0xFFFFu;
    const uint64_t em = ctx.ballot(kept && b != 0);
    uint32_t e = emit_next + prefix_cnt(em);
    emit_next += (uint32_t)ctx.popc64(em);
    if (!kept) return;
    const u32x4a* pay = (const u32x4a*)&io.pay[((uint32_t)a2 >> 16) & 0xFFFFu];
    const u32x4 e1 = pay[0], e2 = pay[1];  // {summed logit, new partial hash} {plen, word id, table view}
    const u32x4a* crp = (const u32x4a*)&cold_cur()[i];
    const u32x4 w0 = crp[0], w1 = crp[1], w2 = crp[2], w3 = crp[3];  // (w3: depth, text node, word id)
    uint32_t depth = w3[0];
    const u32x4 k0 = L.hA[i], k1 = L.hB[i], k2 = L.hC[i];
    const uint32_t pl = k1[2] >> 16;
    uint64_t th = q_lo(k0), hh = q_hi(w1);
    const uint64_t ph = q_hi(e1);  // the new partial word's hash (unchanged for a blank / repeat)
    uint64_t lmhw = q_lo(k2), cth = q_hi(k2);
    uint64_t clm = q_lo(w0), chh = q_hi(w0);
    uint32_t tnode = w3[1], cnode = w2[0], enode = w2[1];
    int32_t pst = (int32_t)w2[2], pen = (int32_t)w2[3];
    uint32_t m2 = e2[2] & ~M2_COMP;
    uint32_t wid = e2[1];
    if (wid & 0x80000000u) {  // the open word of the beam that kept it (score_push): mostly the donor itself
      const uint32_t rb = wid & 0xFFu;
      wid = rb == i ? w3[2] : cold_cur()[rb].wid;
    }
    const uint32_t npl = e2[0];
    if (b == 0) {
      if (!(don & (1u << 29))) pen = frame + 1;  // a repeated label extends the open word (decoder.py:453-461)
      m2 |= k1[3] & M2_COMP;                     // ... and the pending completion stays what it is
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
      clm = 0;
      chh = 0;
      cth = (uint64_t)tnode;  // (no completion of the new open word yet: column C carries the text node, see the layout)
      if (e >= io.emit_cap) {
        status |= ST_EMIT_OVERFLOW;
        e = io.emit_cap - 1;
      }
      *(u32x4a*)&io.emit_nodes[e] = mk4(enode, c | (b << 16), (uint32_t)wst, (uint32_t)wen);
      enode = e;
      depth += 1;
    }
    double ps = 0.0;
    if (npl > 0) ps = partial_score_sel(m2 & PF_PARTIAL_MASK, (m2 & M2_HOT_ON) ? ((m2 >> 8) & 0xFFFFu) : 0u, npl);
    u32x4a* nr = (u32x4a*)&cold_next()[d];
    nr[0] = mk4q(clm, chh);
    nr[1] = mk4q(f64_bits(ps), hh);
    nr[2] = mk4(cnode, enode, (uint32_t)pst, (uint32_t)pen);
    nr[3] = mk4(depth, tnode, wid, 0u);
    o.o0 = mk4q(th, ph);
    o.o1 = mk4(e1[0], e1[1], c | (npl << 16), m2);
    o.o2 = mk4q(lmhw, cth);
  }
  CTC_HD void put_rec(uint32_t d, const Rec& o) {
    L.hA[d] = o.o0;
    L.hB[d] = o.o1;
    L.hC[d] = o.o2;
  }
  CTC_HD void build_done(uint32_t n_new) {
    N = (int)n_new;
    par ^= 1u;
    // lanes may have raised status bits on their own
    status = ctx.wave_or_u32(status);
    ctx.wsync();
    tick<W_PROF_BUILD>();
#ifdef CTC_STORE_PROBE
    ctx.vm_wait();
    tick<W_PROF_ST_BUILD>();
#endif
  }
  // at most 64 ranks (nearly every frame): gather everything the new columns need, then write them in place
  CTC_HD void build1(int frame, uint32_t n) {
    const uint32_t r = (uint32_t)lane;
    uint32_t w = 0;
    if (r < n) w = L.sel[r];
    const bool kept = r < n && (w >> 31) != 0u;
    const uint64_t km = ctx.ballot(kept);
    const uint32_t d = prefix_cnt(km);
    Rec o;
    gather(frame, w, kept, d, o);
    ctx.wsync();
    tick<W_PROF_GATHER>();
    if (kept) put_rec(d, o);
    build_done((uint32_t)ctx.popc64(km));
  }
  // more than 64 ranks: the columns of the ranks 64.. are parked in LDS that is idle during a build (label block +
  // match table) while the ranks 0..63 are gathered, so that only one record per lane lives in registers
  CTC_HD void build_big(int frame, uint32_t n) {
    const uint32_t r1 = (uint32_t)(64 + lane);
    const uint32_t w0 = L.sel[lane];
    uint32_t w1 = 0;
    if (r1 < n) w1 = L.sel[r1];
    const bool kept0 = (w0 >> 31) != 0u, kept1 = r1 < n && (w1 >> 31) != 0u;
    const uint64_t km0 = ctx.ballot(kept0), km1 = ctx.ballot(kept1);
    const uint32_t n0 = (uint32_t)ctx.popc64(km0);
    const uint32_t d0 = prefix_cnt(km0), d1 = n0 + prefix_cnt(km1);
    ctx.wsync();  // (the parking area covers L.sel)
    Rec o;
    gather(frame, w1, kept1, d1, o);
    if (kept1) {
      L.stage[lane * 3] = o.o0;
      L.stage[lane * 3 + 1] = o.o1;
      L.stage[lane * 3 + 2] = o.o2;
    }
    gather(frame, w0, kept0, d0, o);
    ctx.wsync();
    tick<W_PROF_GATHER>();
    if (kept0) put_rec(d0, o);
    if (kept1) {
      o.o0 = L.stage[lane * 3];
      o.o1 = L.stage[lane * 3 + 1];
      o.o2 = L.stage[lane * 3 + 2];
      put_rec(d1, o);
    }
    build_done(n0 + (uint32_t)ctx.popc64(km1));
  }

  // ---- init / import ---------------------------------------------------------------------------
  CTC_HD void write_beam(int i, uint64_t text_h, uint64_t part_h, double logit, uint32_t meta1, uint32_t meta2, double lm_hw,
                         double pscore, uint64_t hist_h, uint32_t text_node, uint32_t emit_node, uint32_t word_id,
                         int32_t pstart, int32_t pend, uint32_t depth) {
    const uint64_t lgb = f64_bits(logit), lmb = f64_bits(lm_hw);
    L.hA[i] = mk4q(text_h, part_h);
    L.hB[i] = mk4((uint32_t)lgb, (uint32_t)(lgb >> 32), meta1, meta2 & ~M2_COMP);
    L.hC[i] = mk4((uint32_t)lmb, (uint32_t)(lmb >> 32), text_node, 0u);  // (no completion yet: the text node, see the layout)
    ColdRec cr;
    cr.c_lmhw = 0.0;
    cr.pscore = pscore;
    cr.hist_h = hist_h;
    cr.c_hist_h = 0;
    cr.cnode = 0;
    cr.enode = emit_node;
    cr.pstart = pstart;
    cr.pend = pend;
    cr.depth = depth;
    cr.tnode = text_node;
    cr.wid = word_id;
    cr.pad = 0;
    cold_cur()[i] = cr;
  }

  CTC_HD void init() {
    text_next = 1;  // text node 0 = empty text
    emit_next = io.emit_start > 0 ? io.emit_start : 1u;  // emission node 0 = root (a resident stream goes on in its arena)
    status = 0;
    fflag = 0;
    par = 0;
    N = 1;
    if (lane == 0) {
      Node root;
      root.raw = 0.0;
      root.hw_cnt = 0;
      root.ring_cnt = 0;
      for (int k = 0; k < MAX_CTX; ++k) root.ring[k] = 0;
      LmState st;
      st.len = 0;
      for (int k = 0; k < MAX_CTX; ++k) {
        st.words[k] = 0;
        st.backoff[k] = 0.f;
      }
      if (io.start_state && io.start_state->len >= 0) st = *io.start_state;
      if (st.len > CTX) st.len = CTX;  // (a state of a model of this order never holds more)
      root.st = st;
      node_store(0, root);
      const uint64_t root_hist = wave_hist_fold(root.ring, 0);
      EmitNode er;
      er.parent = 0;
      er.tok_branch = 0;
      er.wstart = -1;
      er.wend = -1;
      io.emit_nodes[0] = er;
      write_beam(0, 0, 0, 0.0, NO_CHAR, EMPTY_PARTIAL_M2, 0.0, 0.0, root_hist, 0, 0, 0, -1, -1, 0);
    }
    if (io.imports && io.n_import > 0) import_beams();
    ctx.mem_sync();
  }

  // streaming: rebuild the beam table from the caller's beams (their order is the rank order; the host has
  // checked that there are no more of them than the table holds). Beams built by the host are rooted in fresh
  // BR_IMPORT emission nodes; beams carried over on the device (resident streams) keep their emission chains.
  CTC_HD void import_beams() {
    const DeviceTables& T_ = tab();
    const DecodeParams& P_ = prm();
    const int n = io.n_import < BW ? io.n_import : BW;
    const uint32_t emit_base = emit_next;
    const bool host_built = io.imports[0].resident == 0u;
    for (int i = lane; i < n; i += 64) {
      const ImportBeam& m = io.imports[i];
      const uint32_t node = 1u + (uint32_t)i;  // node 0 is the empty text
      Node tn;
      tn.raw = m.raw_lm;
      const double lmhw = m.raw_lm + P_.hot_weight * (double)m.hw_cnt;
      const uint64_t hh = wave_hist_fold(m.ring, m.ring_cnt);
CTC_UNROLL
      for (int k = 0; k < MAX_CTX; ++k) tn.ring[k] = m.ring[k];
      tn.hw_cnt = m.hw_cnt;
      tn.ring_cnt = m.ring_cnt;
      tn.st.len = m.state.len > CTX ? CTX : m.state.len;
CTC_UNROLL
      for (int k = 0; k < MAX_CTX; ++k) {
        tn.st.words[k] = m.state.words[k];
        tn.st.backoff[k] = m.state.backoff[k];
      }
      node_store(node, tn);
      uint32_t enode = m.enode, depth = m.depth;
      if (host_built) {
        EmitNode en;
        en.parent = 0;
        en.tok_branch = (uint32_t)i | (BR_IMPORT << 16);
        en.wstart = -1;
        en.wend = -1;
        enode = emit_base + (uint32_t)i;
        depth = 1u;
        if (enode >= io.emit_cap) {
          status |= ST_EMIT_OVERFLOW;
          enode = io.emit_cap - 1;
        }
        io.emit_nodes[enode] = en;
      }
      const double ps = m.plen > 0 ? partial_score(T_, P_, m.m2 & PF_PARTIAL_MASK, (m.m2 & M2_HOT_ON) ? ((m.m2 >> 8) & 0xFFFFu) : 0u, m.plen) : 0.0;
      write_beam(i, m.text_h, m.part_h, m.logit_score, (m.last_char & 0xFFFFu) | (m.plen << 16),
                 m.plen > 0 ? m.m2 : EMPTY_PARTIAL_M2, lmhw, ps, hh, node, enode, m.word_id, m.pstart, m.pend, depth);
    }
    text_next = 1u + (uint32_t)n;
    if (host_built) emit_next = emit_base + (uint32_t)n;
    N = n;
  }

  // ---- finalisation: _finalize_beams(force_next_word, is_end) + output records (decoder.py:558-602,653-667)
  CTC_HD void finalise() {
    const bool fold = prm().fold != 0, eos = prm().eos != 0;
    pool_n = 0;
    pay_n = 0;
    runmax = asc_key(-INFINITY);
    kth_key = 0;
    ctx.mem_sync();
    if (fold) completions_now();
    ctx.mem_sync();
    const uint32_t Q = (uint32_t)N;  // one candidate per beam
    bool valid[SLB], is_rep[SLB];
    uint32_t rep[SLB], donor[SLB];
    uint64_t ck[SLB];
    double lg[SLB];
    clear_table();
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
        const u32x4 k0 = L.hA[v], k1 = L.hB[v];
        const uint32_t pl = k1[2] >> 16;
        const uint64_t kt = (fold && pl > 0) ? L.c64[v * 2 + 1] : q_lo(k0);  // (folding: every open word has its completion now)
        ck[j] = fin64(kt ^ 0x165667B19E3779F9ull);
        lg[j] = bits_f64(q_lo(k1));
      }
    }
    ctx.wsync();
    if (fold) {
      match<SLB>(valid, ck, rep);
      uint64_t mlo[SLB], mhi[SLB];
      group_masks<SLB>(valid, rep, mlo, mhi);
      // fold in ascending beam rank; scored through the donor's (text, next_word) split (decoder.py:387-395)
CTC_UNROLL
      for (int j = 0; j < SLB; ++j) {
        const uint32_t v = (uint32_t)(j * 64 + lane);
        is_rep[j] = valid[j] && rep[j] == v;
        if (is_rep[j]) {
          while (mlo[j]) {
            const uint32_t mbit = (uint32_t)ctx.ctz64(mlo[j]);
            mlo[j] &= mlo[j] - 1ull;
            lg[j] = lse2(lg[j], bits_f64(L.b64[mbit * 2]));
            donor[j] = mbit;
          }
          while (mhi[j]) {
            const uint32_t mbit = (uint32_t)(64 + ctx.ctz64(mhi[j]));
            mhi[j] &= mhi[j] - 1ull;
            lg[j] = lse2(lg[j], bits_f64(L.b64[mbit * 2]));
            donor[j] = mbit;
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
        const u32x4 d1 = L.hB[d], d2 = L.hC[d];
        const uint32_t m2 = d1[3];
        const uint32_t pl = d1[2] >> 16;
        double lmhw;
        if (eos) {
          const ColdRec dcr = cold_cur()[d];
          Node src;
          node_load(dcr.tnode, src);
          const uint32_t cnt = src.hw_cnt + ((pl > 0 && (m2 & M2_HOT_COMPLETE)) ? 1u : 0u);
          if (tab().has_lm) {
            LmState st = src.st, end;
            const uint32_t wid = pl > 0 ? dcr.wid : 0u;
            const uint32_t wfl = pl > 0 ? m2 : 0u;
            const float base_s = lm_base_score<ORD>(tab(), st, wid, &end);
            double end_score = 0.0;
            if (prm().score_boundary) {
              LmState tmp;
              end_score = (double)lm_base_score<ORD>(tab(), end, tab().eos_id, &tmp);
            }
            const double raw = src.raw + lm_word_score(tab(), prm(), base_s, wfl, end_score, true);
            lmhw = raw + prm().hot_weight * (double)cnt;
          } else {
            lmhw = prm().hot_weight * (double)cnt;
          }
        } else {
          lmhw = pl > 0 ? cold_cur()[d].c_lmhw : bits_f64(q_lo(d2));  // memo entry (text (+) word, False)
        }
        score[j] = tab().has_lm ? lg[j] + lmhw : lg[j] + lmhw + 0.0;
      } else {
        const u32x4 k1 = L.hB[v];
        const uint32_t pl = k1[2] >> 16;
        score[j] = total_score(tab(), lg[j], bits_f64(L.c64[v * 2]), pl > 0 ? cold_cur()[v].pscore : 0.0, pl);
      }
      const uint64_t k = asc_key(score[j]);
      if (k > pass_key) pass_key = k;
    }
    runmax = ctx.wave_max_u64(pass_key);
    if (runmax < asc_key(-INFINITY)) runmax = asc_key(-INFINITY);
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      if ((uint32_t)(j * 64) >= Q) continue;
      // (N <= beam_width <= P: everything fits; arrival = beam rank, donor beam; the exact score rides in the payload)
      pool_put(is_rep[j], mk4q(score_sort_key(score[j]), 0), (uint32_t)(j * 64 + lane), donor[j], lg[j], f64_bits(score[j]), 0u,
               0u, 0u);
    }
    ctx.wsync();
    uint32_t n = rank_pool(key_to_score(runmax) + prm().beam_prune_logp, false);
    if (n > (uint32_t)prm().beam_width) n = (uint32_t)prm().beam_width;
    if (n == 0) status |= ST_NO_BEAMS;
    if (io.carry_out && !eos) carry_beams(n, fold);
    uint32_t n_out = io.want_out ? n : 0u;
    if (prm().n_best > 0 && n_out > (uint32_t)prm().n_best) n_out = (uint32_t)prm().n_best;
    if (prm().texts_only != 0 && n_out > 0) {
      // decode_batch: only the best beam's text is wanted, and a separate launch assembles it (assemble_texts: every
      // utterance's chain walk at once instead of at the tail of this one's life) -- leave it where its chain ends
      if (lane == 0) {
        const uint64_t a2 = L.pa[L.sel[0] & 0x7FFFFFFFu];
        const PoolPay& pp = io.pay[((uint32_t)a2 >> 16) & 0xFFFFu];
        const uint32_t d = (uint32_t)(a2 >> 32);
        OutBeam& ob = io.out[0];
        ob.logit_score = pp.logit;
        ob.lm_score = bits_f64(pp.part_h);
        ob.raw_lm = 0.0;
        ob.tok_off = 0;
        ob.tok_cnt = 0;
        ob.state.len = -1;
        ob.last_char = NO_CHAR;
        ob.pstart = ob.pend = -1;
        ob.pad[0] = 0;
        ob.pad[1] = cold_cur()[d].enode;
        *io.n_out = 1;
        *io.status = status;
      }
      return;
    }
    // output records + back-trace of each returned beam's emission chain
    uint32_t len[SLB], off[SLB];
    uint32_t total = 0;
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      const uint32_t r = (uint32_t)(j * 64 + lane);
      len[j] = 0;
      off[j] = 0;
      if (r < n_out) {
        const uint32_t d = (uint32_t)(L.pa[L.sel[r] & 0x7FFFFFFFu] >> 32);
        len[j] = cold_cur()[d].depth + ((fold && (L.b32[d * 4 + 2] >> 16) > 0) ? 1u : 0u);
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
      const uint64_t a2 = L.pa[L.sel[r] & 0x7FFFFFFFu];
      const PoolPay& pp = io.pay[((uint32_t)a2 >> 16) & 0xFFFFu];
      const uint32_t d = (uint32_t)(a2 >> 32);
      const ColdRec cr = cold_cur()[d];
      OutBeam& ob = io.out[r];
      ob.logit_score = pp.logit;
      ob.lm_score = bits_f64(pp.part_h);
      const u32x4 d1 = L.hB[d], d2 = L.hC[d];
      const uint32_t meta1 = d1[2];
      const uint32_t pl = meta1 >> 16;
      const bool closes = fold && pl > 0;
      const uint32_t o = (uint32_t)(base + off[j]);
      ob.tok_off = o;
      ob.tok_cnt = tok_ok ? len[j] : 0;
      ob.pad[0] = 0;
      ob.pad[1] = 0;
      ob.last_char = fold ? NO_CHAR : (meta1 & 0xFFFFu);
      ob.pstart = fold ? -1 : cr.pstart;
      ob.pend = fold ? -1 : cr.pend;
      // the text's memo entry: raw LM sum and the state after its last word
      Node node;
      node_load(closes ? cr.cnode : cr.tnode, node);
      ob.raw_lm = node.raw;
      if (!tab().has_lm) {
        ob.state.len = -1;
CTC_UNROLL
        for (int k = 0; k < MAX_CTX; ++k) {
          ob.state.words[k] = 0;
          ob.state.backoff[k] = 0.f;
        }
      } else if (eos) {
        // last_lm_state: state after the last word, before </s> (language_model.py:357); an empty
        // last word is still scored as a word (decoder.py:387-395)
        Node src;
        node_load(cr.tnode, src);
        LmState st = src.st, after;
        lm_base_score<ORD>(tab(), st, pl > 0 ? cr.wid : 0u, &after);
        ob.state = after;
      } else {
        ob.state = node.st;
      }
      if (tok_ok) {
        uint32_t pos = o + len[j];
        if (closes) {
          EmitNode fin;
          fin.parent = 0;
          fin.tok_branch = BR_FINAL << 16;
          fin.wstart = cr.pstart;
          fin.wend = cr.pend;
          io.tok_pool[--pos] = fin;
        }
        uint32_t e = cr.enode;
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
      if (io.sstate) {
        io.sstate->n_carry = n;
        io.sstate->emit_next = emit_next;
        io.sstate->status = status;
        io.sstate->pad = 0;
      }
    }
  }

  // Device-resident streams: the ranked beams of this chunk, written where the next chunk's import_beams() reads them
  // (what the reference's caller carries between partial_decode_beams calls, decoder.py:681-728). A beam whose open
  // word the finalisation closed (force_next_word) gets a BR_FINAL emission node for that word.
  CTC_HD void carry_beams(uint32_t n, bool fold) {
CTC_UNROLL
    for (int j = 0; j < SLB; ++j) {
      if ((uint32_t)(j * 64) >= n) continue;
      const uint32_t r = (uint32_t)(j * 64 + lane);
      const bool mine = r < n;
      uint32_t d = 0;
      double lg = 0.0;
      if (mine) {
        const uint64_t a2 = L.pa[L.sel[r] & 0x7FFFFFFFu];
        d = (uint32_t)(a2 >> 32);
        lg = io.pay[((uint32_t)a2 >> 16) & 0xFFFFu].logit;
      }
      const u32x4 d0 = L.hA[d], d1 = L.hB[d], d2 = L.hC[d];
      const uint32_t meta1 = d1[2];
      const uint32_t pl = meta1 >> 16;
      const bool closes = mine && fold && pl > 0;
      const uint64_t cm = ctx.ballot(closes);
      uint32_t e = emit_next + prefix_cnt(cm);
      emit_next += (uint32_t)ctx.popc64(cm);
      if (!mine) continue;
      const ColdRec cr = cold_cur()[d];
      uint32_t enode = cr.enode, depth = cr.depth;
      if (closes) {
        if (e >= io.emit_cap) {
          status |= ST_EMIT_OVERFLOW;
          e = io.emit_cap - 1;
        }
        *(u32x4a*)&io.emit_nodes[e] = mk4(enode, BR_FINAL << 16, (uint32_t)cr.pstart, (uint32_t)cr.pend);
        enode = e;
        depth += 1;
      }
      Node node;
      node_load(closes ? cr.cnode : cr.tnode, node);
      ImportBeam& m = io.carry_out[r];
      m.logit_score = lg;
      m.raw_lm = node.raw;
      m.text_h = closes ? q_hi(d2) : q_lo(d0);  // (the text of the node: its hash is the beam's, or its completion's)
      m.part_h = fold ? 0ull : q_hi(d0);
CTC_UNROLL
      for (int k = 0; k < MAX_CTX; ++k) m.ring[k] = node.ring[k];
      m.ring_cnt = node.ring_cnt;
      m.hw_cnt = node.hw_cnt;
      m.plen = fold ? 0u : pl;
      m.last_char = fold ? NO_CHAR : (meta1 & 0xFFFFu);
      m.m2 = fold ? EMPTY_PARTIAL_M2 : (d1[3] & ~M2_COMP);
      m.word_id = fold ? 0u : cr.wid;
      m.pstart = fold ? -1 : cr.pstart;
      m.pend = fold ? -1 : cr.pend;
      m.state = node.st;
      m.enode = enode;
      m.depth = depth;
      m.resident = 1u;
    }
    status = ctx.wave_or_u32(status);
  }

  CTC_HD void run() {
    load_hot();
    init();
    if (PROF && io.prof && lane == 0) t_last = ctx.clock();
    prefetch(0);
    {
      TokRegs tr;
      tok_load(tr);
      tok_commit(tr);
    }
    for (int t = 0; t < io.T;) {
      const int t2 = step(t);
      ctx.frame_done(t, t2, io.T);
      t = t2;
    }
    finalise();
    tick<W_PROF_FINAL>();
  }
};

}  // namespace ctc
