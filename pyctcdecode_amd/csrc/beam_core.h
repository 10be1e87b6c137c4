// beam_core.h -- the per-utterance CTC prefix-beam recursion (reference:
// BeamSearchDecoderCTC._partial_decode_logits decoder.py:426-556, _finalize_beams :558-602,
// _get_lm_beams :346-424, _merge_beams :211-224, _prune_history :227-258).
//
// One workgroup owns one utterance and walks its frames in order; everything a frame needs
// (live beam table, candidate keys, merge table, candidate pool, sort buffer) lives in LDS.
// Global memory holds only the read-only scorer tables, the per-frame survivor lists written by
// the frame-prune kernel, and two append-only arenas (TextNode, EmitNode).
//
// The code is written against an execution context `Ctx` {tid, nt, sync(), LDS atomics} so the
// same source runs as a HIP workgroup (backend_hip.hip) and as a 1-thread sequential simulation
// (tests/sim, test infrastructure only -- the product never runs it).
#pragma once
#include <math.h>
#include "common.h"

namespace ctc {

// LDS-resident arrays are addressed through 32-bit address_space(3) pointers on the device so that
// every access is a ds_* instruction with a constant offset (the layout is a compile-time constant
// of the kernel instantiation); on the host (sim, sizing) they are ordinary pointers.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CTC_SIM)
#define CTC_LDS __attribute__((address_space(3)))
#else
#define CTC_LDS
#endif
template <class T>
struct LPtr {
  CTC_LDS T* p;
  CTC_HD CTC_LDS T& operator[](int i) const { return p[i]; }
  CTC_HD CTC_LDS T& operator[](uint32_t i) const { return p[i]; }
};
typedef CTC_LDS char* lds_bytes_t;

constexpr uint32_t NO_CHAR = 0xFFFFu;
constexpr uint32_t M2_HOT_ON = 16u;        // partial is a prefix of a hot word
constexpr uint32_t M2_HOT_COMPLETE = 32u;  // partial is itself a hot word
constexpr uint32_t EMPTY_PARTIAL_M2 = PF_ON_TABLE | M2_HOT_ON;

enum : uint32_t { MODE_A = 0, MODE_ALL_B = 1, MODE_FIRST_B = 2, MODE_C = 3, MODE_D = 4 };
enum : uint32_t { ST_TEXT_OVERFLOW = 1u, ST_EMIT_OVERFLOW = 2u, ST_POOL_OVERFLOW = 4u, ST_TOK_OVERFLOW = 8u, ST_NO_BEAMS = 16u };

struct BeamSoA {
  LPtr<double> logit, lm_hw, pscore, c_lm_hw;
  LPtr<uint64_t> text_h, part_h, hist_h, c_text_h, c_hist_h;
  LPtr<uint32_t> text_node, comp_node, emit_node, word_id, meta1, meta2, depth;
  LPtr<int32_t> pstart, pend;
};

// what the recursion needs to know about a label, staged in LDS for the first survivors of a frame
struct TokLite {  // 56 B
  uint64_t h_raw, pow_raw, h_clean;
  uint32_t len_raw, len_clean, flags, start_flags, start_word_id, hot_min, hot_complete, pad;
};
constexpr int TOK_STAGE = 32;

struct Surv {  // one surviving label of the current frame
  uint32_t id;
  uint32_t mode;  // MODE_* | first_non_repeat << 8
  double lp;
};

// sizes that shape the LDS carve-up (host computes the same numbers for the launch)
struct LdsShape {
  int bw;    // beam capacity (beam_width rounded up to 8)
  int cand;  // candidates per chunk
  int tab;   // merge-table slots (a power of two >= 2 * cand)
  int pool;  // pool capacity
  int sortn; // sort-buffer entries (power of two >= pool)
  int surv;  // survivors per frame capacity
};

CTC_HD size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// LDS shape for a beam width and survivor bound (same numbers on host and device)
CTC_HD int beam_bucket(int beam_width) {  // beam-table capacities the kernel is instantiated for
  return beam_width <= 32 ? 32 : beam_width <= 64 ? 64 : beam_width <= 128 ? 128 : 256;
}
constexpr int CAND_CHUNK = 512;  // candidates merged per pass (>= 2 * largest beam bucket, >= 512: the pool's 1024-bucket histogram
                                 // lives in the 2 * cand slots of the merge table)
constexpr int CAND_CHUNK_WIDE = 1024;  // ... when a workgroup has a CU's LDS to itself (beam_decode<*, 512>: at most one
                                       // utterance per CU). A frame of ~2 500 candidates (BASELINE configs[1]) takes 3 chunks and
                                       // one pool compaction instead of 5 and 4.
// candidates per chunk of the workgroup kernel's variants (host and device agree through this one rule)
CTC_HD int group_cand(int bw_bucket, bool wide) { return wide && bw_bucket <= 128 ? CAND_CHUNK_WIDE : CAND_CHUNK; }
// ... and the slots of their merge table: a quarter full at most where LDS is plentiful (linear probing: the longest probe
// chain among a wave's 64 lanes sets the pace), half full otherwise
CTC_HD int group_tab(int cand) { return cand > CAND_CHUNK ? 4 * cand : 2 * cand; }
CTC_HD LdsShape make_shape(int beam_width, int max_surv, int cand = CAND_CHUNK) {
  LdsShape s;
  s.bw = beam_bucket(beam_width);
  s.cand = cand;
  s.tab = group_tab(cand);
  s.pool = cand + s.bw;  // one chunk of fresh candidates + the best beam_width kept so far
  s.sortn = 2 * cand;    // a power of two >= pool for every bucket (bw <= 256 <= cand)
  s.surv = (max_surv + 3) & ~3;
  return s;
}

struct LdsView {
  BeamSoA beams0;  // both beam tables: every array holds 2*bw entries, table k starts at k*bw
  int bw;
  // candidates of the current chunk
  LPtr<uint64_t> ck_text, ck_part;
  LPtr<double> c_logit;
  LPtr<uint32_t> crep, rmin, rmax, rcnt;
  LPtr<uint32_t> table;  // shape.tab slots, stores q+1 | hash bits
  // pool of merged, scored candidates of the current frame
  LPtr<double> p_score, p_logit;
  LPtr<uint32_t> p_arr, p_don;
  LPtr<uint32_t> p_wid, p_m2;  // prefix/hot-word view of the candidate's new partial word
  // sort buffer (aliases the candidate arrays)
  LPtr<uint64_t> s_k0, s_k1;
  LPtr<uint64_t> s_hk;  // folded history-prune key of the first 256 compacted entries (small-set ranking)
  // scalars
  LPtr<uint32_t> scal;  // [0] pool_n [1] text_next [2] emit_next [3] flag [4] need_comp [5] n_sel [6] status [7] n_new [8] tok_len
  LPtr<uint64_t> smax;  // [0] sortable max score [1] token pool base [2] sortable score of the beam_width-th best so far
  LPtr<uint32_t> keep;  // per selected beam: kept by history prune
  LPtr<uint32_t> sel;   // pool indices of the selected candidates in (score desc, arrival asc) order
  LPtr<uint32_t> part;  // 64 partial sums of the bucket histogram (large-set selection)
  LPtr<uint64_t> hk_h, hk_p;  // history-prune keys of the selected beams
  LPtr<uint32_t> hk_c;
  // gather temp used when the pool is compacted (aliases the tail of the candidate arrays)
  LPtr<double> g_score, g_logit;
  LPtr<uint32_t> g_arr, g_don, g_wid, g_m2;
  // survivors of the current frame (last: its size is the only run-time quantity)
  LPtr<TokLite> stok;  // label constants of survivors [0, TOK_STAGE)
  LPtr<Surv> surv;
};

template <class T>
CTC_HD LPtr<T> lds_take(lds_bytes_t& p, size_t bytes) {
  LPtr<T> r;
  r.p = (CTC_LDS T*)p;
  p += align16(bytes);
  return r;
}

CTC_HD void carve_beams(BeamSoA& b, lds_bytes_t& p, int bw) {
  b.logit = lds_take<double>(p, 8 * bw);
  b.lm_hw = lds_take<double>(p, 8 * bw);
  b.pscore = lds_take<double>(p, 8 * bw);
  b.c_lm_hw = lds_take<double>(p, 8 * bw);
  b.text_h = lds_take<uint64_t>(p, 8 * bw);
  b.part_h = lds_take<uint64_t>(p, 8 * bw);
  b.hist_h = lds_take<uint64_t>(p, 8 * bw);
  b.c_text_h = lds_take<uint64_t>(p, 8 * bw);
  b.c_hist_h = lds_take<uint64_t>(p, 8 * bw);
  b.text_node = lds_take<uint32_t>(p, 4 * bw);
  b.comp_node = lds_take<uint32_t>(p, 4 * bw);
  b.emit_node = lds_take<uint32_t>(p, 4 * bw);
  b.word_id = lds_take<uint32_t>(p, 4 * bw);
  b.meta1 = lds_take<uint32_t>(p, 4 * bw);
  b.meta2 = lds_take<uint32_t>(p, 4 * bw);
  b.depth = lds_take<uint32_t>(p, 4 * bw);
  b.pstart = lds_take<int32_t>(p, 4 * bw);
  b.pend = lds_take<int32_t>(p, 4 * bw);
}

// Carves `base` into the view; returns bytes used.
CTC_HD size_t lds_carve(LdsView& o, lds_bytes_t base, const LdsShape& s) {
  lds_bytes_t p = base;
  carve_beams(o.beams0, p, 2 * s.bw);
  o.bw = s.bw;
  o.p_score = lds_take<double>(p, 8 * s.pool);
  o.p_logit = lds_take<double>(p, 8 * s.pool);
  o.p_arr = lds_take<uint32_t>(p, 4 * s.pool);
  o.p_don = lds_take<uint32_t>(p, 4 * s.pool);
  o.p_wid = lds_take<uint32_t>(p, 4 * s.pool);
  o.p_m2 = lds_take<uint32_t>(p, 4 * s.pool);
  o.scal = lds_take<uint32_t>(p, 4 * 16);
  o.smax = lds_take<uint64_t>(p, 8 * 4);
  o.keep = lds_take<uint32_t>(p, 4 * s.bw);
  o.sel = lds_take<uint32_t>(p, 4 * s.bw);
  o.part = lds_take<uint32_t>(p, 4 * 64);
  // one block: the two key arrays of the large-set path, or the 256 folded keys of the small-set path
  o.s_hk = lds_take<uint64_t>(p, 8 * (2 * s.bw > 256 ? 2 * s.bw : 256));
  o.hk_h.p = o.s_hk.p;
  o.hk_p.p = o.s_hk.p + s.bw;
  o.hk_c = lds_take<uint32_t>(p, 4 * s.bw);
  // candidate arrays and the sort buffer share one region
  lds_bytes_t shared0 = p;
  o.ck_text = lds_take<uint64_t>(p, 8 * s.cand);
  o.ck_part = lds_take<uint64_t>(p, 8 * s.cand);
  o.c_logit = lds_take<double>(p, 8 * s.cand);
  o.crep = lds_take<uint32_t>(p, 4 * s.cand);
  o.rmin = lds_take<uint32_t>(p, 4 * s.cand);
  o.rmax = lds_take<uint32_t>(p, 4 * s.cand);
  o.rcnt = lds_take<uint32_t>(p, 4 * s.cand);
  o.table = lds_take<uint32_t>(p, 4 * s.tab);
  lds_bytes_t q = shared0;
  o.s_k0 = lds_take<uint64_t>(q, 8 * s.sortn);
  o.s_k1 = lds_take<uint64_t>(q, 8 * s.sortn);
  o.g_score = lds_take<double>(q, 8 * s.bw);
  o.g_logit = lds_take<double>(q, 8 * s.bw);
  o.g_arr = lds_take<uint32_t>(q, 4 * s.bw);
  o.g_don = lds_take<uint32_t>(q, 4 * s.bw);
  o.g_wid = lds_take<uint32_t>(q, 4 * s.bw);
  o.g_m2 = lds_take<uint32_t>(q, 4 * s.bw);
  if (q > p) p = q;
  o.stok = lds_take<TokLite>(p, sizeof(TokLite) * TOK_STAGE);
  o.surv = lds_take<Surv>(p, sizeof(Surv) * s.surv);
  return (size_t)(p - base);
}

CTC_HD size_t lds_bytes(const LdsShape& s) {
  LdsView tmp;
  return lds_carve(tmp, (lds_bytes_t) nullptr, s);
}

// wave kernel (beam_wave.h): what the candidate passes do not read per (label, beam) lives outside LDS, two buffers of
// COLD_STRIDE records per utterance used alternately by the table builds
struct ColdRec {  // 64 B = four 16-byte chunks; a candidate reads ONE of the first two (one load): chunk 0 when its label
                  // closes the beam's open word, chunk 1 otherwise
  double c_lmhw;           // lm + hot-word score of text (+) open word, once that completion exists (M2_COMP)
  uint64_t c_hist_h;       // hash of the last n_hist words of that completion
  double pscore;           // partial score of the open word
  uint64_t hist_h;         // hash of the text's last n_hist words (decoder.py:250-251)
  uint32_t cnode;          // TextNode of the completion
  uint32_t enode;          // end of the beam's emission chain
  int32_t pstart, pend;    // partial_frames of the open word
  uint32_t depth;          // emission nodes on the beam's chain
  uint32_t tnode;          // text node of the beam's completed words (the wave kernel also keeps it in the beam's LDS column C
                           // while the open word has no completion)
  uint32_t wid;            // word id of the open word in the LM vocabulary (0: none / not a word)
  uint32_t pad;
};
// ... and what only the table build needs of a pooled candidate: one line per push, per utterance and frame
struct PoolPay {  // 32 B
  double logit;            // summed over the merged duplicates
  uint64_t part_h;         // the new open word (finalisation: the exact lm_score)
  uint32_t plen, wid, m2;  // its code points, word id and table view
  uint32_t pad;
};
constexpr int COLD_STRIDE = 128;

// per-utterance global-memory view
struct UttIO {
  const uint32_t* surv_cnt;  // [T]
  const uint16_t* surv_id;   // [T * max_surv]   CPython-set order
  const double* surv_lp;     // [T * max_surv]
  int32_t T;
  TextNode* text_nodes;
  uint32_t text_cap;
  EmitNode* emit_nodes;
  uint32_t emit_cap;
  const LmState* start_state;  // nullptr: LM default; several LMs: n_lms states
  LmState* out_xstates;        // several LMs: [beam_width * (n_lms - 1)] states of LM 1.. per output beam
  OutBeam* out;                // [beam_width]
  uint32_t* n_out;
  uint32_t* status;
  EmitNode* tok_pool;          // global pool of back-traced emission lists
  unsigned long long* tok_pool_head;
  unsigned long long tok_pool_cap;
  unsigned long long* prof;  // optional per-phase cycle accumulators (diagnostics), else nullptr
  const ImportBeam* imports;  // streaming: the caller's live beams (rank order), else nullptr
  const LmState* import_xstates;  // several LMs: [n_import * (n_lms - 1)] states of LM 1.. of those beams
  int32_t n_import;
  int32_t first_frame;        // processed_frames of this utterance (decoder.py:443)
  ColdRec* cold;              // wave kernel: [2 * COLD_STRIDE]
  PoolPay* pay;               // wave kernel: [wave_pay_stride(params)] payload lines of the current frame's pool
  // device-resident streams (nullptr / 0 otherwise): where the finalisation leaves the beams for the next chunk (and
  // their LM states of model 1.. for several LMs), the stream's counters, the first free emission node at entry
  // (0: a fresh arena), and whether output records + emission lists are wanted at all for this chunk
  ImportBeam* carry_out;
  LmState* carry_xstates;
  StreamState* sstate;
  uint32_t emit_start;
  int32_t want_out;
};

// (the launch behind the beam kernel when DecodeParams::texts_only is set: backend_hip.hip assemble_texts)
// ONE thread: the text of a beam (decoder.py:653-667: its words joined by single spaces) from its emission chain, leaf
// to root, written backwards into scratch[.. cap); returns where it starts. Walking backwards a separator is due when a
// word boundary (BR_BOUNDARY / BR_SPACE / BR_FINAL) has been passed since the last bytes and there are bytes to its right.
// max_steps bounds the walk (a chain is never longer than its arena; an utterance that overflowed its arenas is being
// redone and may have left a cycle behind).
CTC_HD uint32_t text_backwards(const EmitNode* emit_nodes, const DeviceTables& tab, uint32_t enode, uint8_t* scratch, uint32_t cap,
                               uint32_t max_steps) {
  uint32_t pos = cap;
  bool emitted = false, pending = false;
  uint32_t steps = 0;
  for (uint32_t e = enode; e != 0 && steps < max_steps; ++steps) {
    const EmitNode en = emit_nodes[e];
    const uint32_t br = en.tok_branch >> 16, tok = en.tok_branch & 0xFFFFu;
    uint32_t off = 0, len = 0;
    if (br == BR_APPEND) {
      off = tab.tok_text[tok].raw_off;
      len = tab.tok_text[tok].raw_len;
    } else if (br == BR_BOUNDARY) {
      off = tab.tok_text[tok].clean_off;
      len = tab.tok_text[tok].clean_len;
    }
    if (len > 0) {
      if (pending && emitted && pos > 0) scratch[--pos] = (uint8_t)' ';
      pending = false;
      for (uint32_t k = len; k > 0 && pos > 0; --k) scratch[--pos] = tab.tok_bytes[off + k - 1];
      emitted = true;
    }
    if (br == BR_BOUNDARY || br == BR_SPACE || br == BR_FINAL) pending = true;
    e = en.parent;
  }
  return pos;
}
constexpr int N_PROF = 24;

// ---------------------------------------------------------------------------------------------
CTC_HD uint64_t score_sort_key(double s) {
  // ascending key order == descending score order; -0.0 and +0.0 compare equal like in Python
  // (s + 0.0 turns -0.0 into +0.0 and leaves every other value alone; the sign then selects "flip all bits" / "flip the
  // sign bit" through an arithmetic shift: five instructions instead of two compares and four selects)
  union { double d; uint64_t u; } c;
  c.d = s + 0.0;
  const uint64_t m = (uint64_t)((int64_t)c.u >> 63);
  return ~(c.u ^ (m | (1ull << 63)));
}

// decoder.py:170-177: s_hi + math.log(1 + math.exp(s_lo - s_hi)). exp and log are written out here (the same source on
// the device and in the CPU simulator) instead of calling the math library: the argument ranges are known -- the exponent
// is in [-37, 0] (below that 1 + e rounds to 1 and the logarithm is exactly 0, as in the reference), the logarithm's
// argument in [1, 2] -- so neither needs the library versions' range checks, denormal and overflow handling: ~75
// instructions instead of ~200 per merged duplicate. Both stay below one ulp (against glibc on 2*10^7 random arguments:
// never more than one ulp apart; the merged score differs in the last bit in 0.2 % of the cases, the rate at which any
// two libms disagree).
CTC_HD double lse_bits_f64(uint64_t u) {
  union { uint64_t u; double d; } c;
  c.u = u;
  return c.d;
}
// The 24 double-precision constants of the two series. On the device they are READ (scalar loads from constant memory
// behind an opaque copy of the table's address) instead of written into the code: an fp64 literal costs two v_mov per use
// -- forty of the ~110 instructions of a merge -- and a scalar-register pair read by the fma costs none.
struct LseTab {
  double log2e, ln2_hi, ln2_lo, e13, e12, e11, e10, e9, e8, e7, e6, e5, e4, e3, e2;  // exp: 1/k!
  double sqrt2, l6, l4, l2, l7, l5, l3, l1;                                           // log: Sun's Lg1 .. Lg7
  double r0, r1, r2;  // 1 / x on [2, 1 + sqrt 2] to ~1e-3: r0 + (x - 2.2) (r1 + (x - 2.2) r2), the seed of three Newton steps
};
#define CTC_LSE_TAB_INIT                                                                                                    \
  {1.4426950408889634074, 6.93147180369123816490e-01, 1.90821492927058770002e-10, 1.0 / 6227020800.0, 1.0 / 479001600.0,      \
   1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0,      \
   1.0 / 6.0, 0.5, 1.4142135623730951, 1.531383769920937332e-01, 2.222219843214978396e-01, 3.999999999940941908e-01,         \
   1.479819860511658591e-01, 1.818357216161805012e-01, 2.857142874366239149e-01, 6.666666666666735130e-01,                  \
   1.0 / 2.2, -1.0 / (2.2 * 2.2), 1.0 / (2.2 * 2.2 * 2.2)}
// a * b + c with c a CONSTANT of that table: the three-operand v_fma_f64 reads it straight from its scalar registers (left
// to itself the compiler picks the two-operand v_fmac_f64, whose addend has to sit in the destination: two v_mov first)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CTC_SIM)
__device__ __forceinline__ double fma_c(double a, double b, double c) {
  double o;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "s"(c));
  return o;
}
#else
inline double fma_c(double a, double b, double c) { return fma(a, b, c); }
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CTC_SIM)
static __device__ __constant__ const LseTab CTC_LSE_TAB = CTC_LSE_TAB_INIT;
__device__ __forceinline__ const LseTab& lse_tab() {
  typedef const LseTab __attribute__((address_space(4))) * P4;
  P4 p = (P4)&CTC_LSE_TAB;
  asm volatile("" : "+s"(p));
  return *(const LseTab*)p;
}
#else
static const LseTab CTC_LSE_TAB = CTC_LSE_TAB_INIT;
inline const LseTab& lse_tab() { return CTC_LSE_TAB; }
#endif

CTC_HD double exp_m37_0(const LseTab& C, double d) {  // e^d, -37 <= d <= 0
  const double kf = rint(d * C.log2e);
  double r = fma(-kf, C.ln2_hi, d);
  r = fma(-kf, C.ln2_lo, r);  // |r| <= ln2 / 2
  // e^r = 1 + r + r^2 q(r), Taylor to r^13 (truncation 4e-18)
  double q = C.e13;
  q = fma_c(q, r, C.e12);
  q = fma_c(q, r, C.e11);
  q = fma_c(q, r, C.e10);
  q = fma_c(q, r, C.e9);
  q = fma_c(q, r, C.e8);
  q = fma_c(q, r, C.e7);
  q = fma_c(q, r, C.e6);
  q = fma_c(q, r, C.e5);
  q = fma_c(q, r, C.e4);
  q = fma_c(q, r, C.e3);
  q = fma_c(q, r, C.e2);
  const double p = fma(r * r, q, r) + 1.0;
  const int32_t k = (int32_t)kf;  // -54 .. 0: the scale is a normal number (a 32-bit conversion: one instruction)
  return p * lse_bits_f64((uint64_t)(uint32_t)(k + 1023) << 52);
}
CTC_HD double log_1_2(const LseTab& C, double x) {  // log x, 1 <= x <= 2 (the classic s = f / (2 + f) series, Sun's coefficients)
  const bool big = x > C.sqrt2;
  const double m = big ? x * 0.5 : x, dk = big ? 1.0 : 0.0;
  const double f = m - 1.0;
  // s = f / (2 + f) without the division (~30 instructions): a quadratic for the reciprocal of the denominator (2 .. 2.42,
  // ~1e-3) refined by three Newton steps (1e-6, 1e-12, 1e-24), all explicit fma's -- the same bits on the device and in the
  // simulator. s is within an ulp or two of the quotient; the series only needs it to about 2^-50.
  const double den = 2.0 + f, dx = f - 0.2;  // (den - 2.2)
  double r = fma_c(dx, fma_c(dx, C.r2, C.r1), C.r0);
  r = fma(fma(-den, r, 1.0), r, r);
  r = fma(fma(-den, r, 1.0), r, r);
  r = fma(fma(-den, r, 1.0), r, r);
  const double s = f * r;
  const double z = s * s, w = z * z;
  const double t1 = w * fma_c(w, fma_c(w, C.l6, C.l4), C.l2);
  const double t2 = z * fma_c(w, fma_c(w, fma_c(w, C.l7, C.l5), C.l3), C.l1);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  return dk * C.ln2_hi - ((hfsq - (s * (hfsq + R) + dk * C.ln2_lo)) - f);
}
CTC_HD double lse2(double a, double b) {
  const bool ge = a >= b;
  const double hi = ge ? a : b, lo = ge ? b : a;
  const double d = lo - hi;
  if (d < -37.0) return hi;  // 1 + e^d == 1
  const LseTab& C = lse_tab();
  return hi + log_1_2(C, 1.0 + exp_m37_0(C, d));  // (a NaN falls through and stays a NaN)
}

CTC_HD uint64_t hist_hash(const uint64_t* ring, uint32_t cnt) {
  uint64_t h = 0x9E3779B97F4A7C15ull + cnt;
CTC_UNROLL
  for (int k = MAX_CTX - 1; k >= 0; --k)
    if ((uint32_t)k < cnt) h = mix64(h ^ ring[k]) + 0x632BE59BD9B4E019ull;
  return h;
}

// x / 6.0, correctly rounded, without a division (language_model.py:334: unk_score * len / AVG_TOKEN_LEN): q = x * RN(1/6),
// one residual step r = x - 6 q (exact in an fma), q + r * RN(1/6). That is the IEEE quotient for every x whose quotient is a
// normal number (Markstein's theorem: RN(1/6) is within half an ulp of 1/6 and q is within an ulp of x/6); checked against
// the division itself on 1.9 * 10^8 arguments -- random doubles over 60 decades, unk_offset * length products, random bit
// patterns -- without a difference. Three instructions instead of the ~30 of a double-precision division.
CTC_HD double div_by_6(double x) {
  const double y = 1.0 / 6.0;
  const double q = x * y;
  return fma(fma(-6.0, q, x), y, q);
}

// language_model.py:141-150 (hot word) / :326-336 (unigram trie) / decoder.py:363-367,397-409
CTC_HD double partial_score(const DeviceTables& t, const DecodeParams& prm, uint32_t pf_flags,
                            uint32_t hot_min_len, uint32_t plen) {
  if (hot_min_len > 0) return prm.hot_weight * (double)plen / (double)hot_min_len;
  if (!t.has_lm) return 0.0;
  bool on_trie = t.has_trie && (pf_flags & PF_UNI_PREFIX);
  double s = prm.unk * (on_trie ? 0.0 : 1.0);
  if (plen > 6) s = div_by_6(s * (double)plen);
  if (t.n_lms > 1) {  // MultiLanguageModel.score_partial_token: np.mean over the models (language_model.py:477-481)
CTC_UNROLL
    for (int k = 1; k < MAX_LMS; ++k) {
      if ((uint32_t)k < t.n_lms) {
        const LmExtra& x = t.x[k - 1];
        const bool on_k = x.has_trie && (pf_flags & (1u << (PF_X_SHIFT + k)));
        double sk = x.unk * (on_k ? 0.0 : 1.0);
        if (plen > 6) sk = div_by_6(sk * (double)plen);
        s = s + sk;
      }
    }
    s = s / (double)t.n_lms;
  }
  return s;
}

CTC_HD double total_score(const DeviceTables& t, double logit, double lm_hw, double ps, uint32_t plen) {
  if (!t.has_lm) return logit + lm_hw + ps;  // decoder.py:363-367
  double s = lm_hw;
  if (plen > 0) s = s + ps;  // decoder.py:398-409
  return logit + s;          // decoder.py:420
}

// language_model.py:338-360 without the EOS term
CTC_HD double lm_word_score_core(bool uniset_nonempty, double alpha, double beta, double unk, double log_base_change,
                                 float base, bool uni_word, bool lm_word, double end_score, bool eos) {
  double lm = (double)base;
  bool oov = (uniset_nonempty && !uni_word) || !lm_word;
  if (oov) lm += unk;
  if (eos) lm = lm + end_score;
  return alpha * lm * log_base_change + beta;
}
CTC_HD double lm_word_score(const DeviceTables& t, const DecodeParams& prm, float base, uint32_t wflags,
                            double end_score, bool eos) {
  return lm_word_score_core(t.uniset_nonempty != 0, prm.alpha, prm.beta, prm.unk, prm.log_base_change, base,
                            (wflags & PF_UNI_WORD) != 0, (wflags & PF_LM_WORD) != 0, end_score, eos);
}

// MULTI: the language model is a MultiLanguageModel (2..MAX_LMS n-gram models, language_model.py:455-502).
// A text then owns n_lms consecutive TextNodes: the first is the node proper, node k only carries model
// k's state. Compile-time so that the single-model kernel contains none of it.
template <class Ctx, bool MULTI = false>
struct BeamDecoder {
  Ctx& ctx;
  LdsView& L;
  const LdsShape& shape;
  const DeviceTables& tab;
  const DecodeParams& prm;
  const UttIO& io;
  int cur;  // live beam buffer
  int N;    // live beams
  uint32_t pf_cnt = 0, pf_id = 0;  // survivors of the NEXT frame, fetched one frame ahead
  double pf_lp = 0.0;
  bool pf_live = false;
  bool flush_nodes = false;        // TextNode stores of this frame still to be completed behind a barrier
  bool run_ok = false;             // the beam table is the output of a full frame of this launch (label_run)
  TokLite pf_tok;                  // ... and the label constants of this thread's survivor
  unsigned long long t_last = 0;
  unsigned long long t_acc[N_PROF] = {};

  // diagnostics: attribute the cycles since the previous tick to `phase` (thread 0, only if asked)
  template <int PHASE>
  CTC_HD void tick() {
    if (io.prof && ctx.tid == 0) {
      unsigned long long now = ctx.clock();
      t_acc[PHASE] += now - t_last;
      t_last = now;
    }
  }

  CTC_HD BeamDecoder(Ctx& c, LdsView& l, const LdsShape& s, const DeviceTables& t, const DecodeParams& p,
                     const UttIO& i)
      : ctx(c), L(l), shape(s), tab(t), prm(p), io(i), cur(0), N(1) {}

  // beam table `which` (0/1) as a by-value bundle of LDS pointers (no run-time indexed struct arrays:
  // those would force the whole view into scratch memory)
  CTC_HD BeamSoA beams_at(int which) const {
    const int off = which * L.bw;
    const BeamSoA& a = L.beams0;
    BeamSoA r;
    r.logit.p = a.logit.p + off;
    r.lm_hw.p = a.lm_hw.p + off;
    r.pscore.p = a.pscore.p + off;
    r.c_lm_hw.p = a.c_lm_hw.p + off;
    r.text_h.p = a.text_h.p + off;
    r.part_h.p = a.part_h.p + off;
    r.hist_h.p = a.hist_h.p + off;
    r.c_text_h.p = a.c_text_h.p + off;
    r.c_hist_h.p = a.c_hist_h.p + off;
    r.text_node.p = a.text_node.p + off;
    r.comp_node.p = a.comp_node.p + off;
    r.emit_node.p = a.emit_node.p + off;
    r.word_id.p = a.word_id.p + off;
    r.meta1.p = a.meta1.p + off;
    r.meta2.p = a.meta2.p + off;
    r.depth.p = a.depth.p + off;
    r.pstart.p = a.pstart.p + off;
    r.pend.p = a.pend.p + off;
    return r;
  }

  CTC_HD uint32_t last_char(const BeamSoA& b, int i) const { return b.meta1[i] & 0xFFFFu; }
  CTC_HD uint32_t plen(const BeamSoA& b, int i) const { return b.meta1[i] >> 16; }

  // ---- completion of beam i's open word: the (text (+) partial) prefix ----------------------
  // The LM state of the text node this thread will complete (beam tid), fetched at the start of the frame:
  // the n-gram probes of the completion phase depend on it, so having it in registers by then turns two
  // dependent memory round trips (node, then probes) into one (probes + the rest of the node together).
  LmState pre_state;
  bool pre_valid = false;

  CTC_HD void prefetch_node(const BeamSoA& b) {
    pre_valid = false;
    const int i = ctx.tid;
    if (tab.has_lm && i < N && plen(b, i) > 0 && b.comp_node[i] == 0) {
      const TextNode& src = io.text_nodes[b.text_node[i]];
      pre_state.len = src.state.len;
CTC_UNROLL
      for (int k = 0; k < MAX_CTX; ++k) {
        pre_state.words[k] = src.state.words[k];
        pre_state.backoff[k] = src.state.backoff[k];
      }
      pre_valid = true;
    }
  }

  CTC_HD void make_completion(const BeamSoA& b, int i) {
    const TextNode& src = io.text_nodes[b.text_node[i]];
    LmState st;
    if (i == ctx.tid && pre_valid) {
      st = pre_state;
    } else {
      st.len = src.state.len;
CTC_UNROLL
      for (int k = 0; k < MAX_CTX; ++k) {
        st.words[k] = src.state.words[k];
        st.backoff[k] = src.state.backoff[k];
      }
    }
    make_completion_from(b, i, src, st);
  }

  CTC_HD uint32_t node_span() const { return MULTI ? tab.n_lms : 1u; }

  // One word under every model of a MultiLanguageModel: model k scores it from its own state in[k] and
  // leaves its new state in out[k]; LanguageModel.score of each (language_model.py:338-360), averaged
  // (:495-501). `uwid` indexes the union word list; each model maps it to its own vocabulary.
  template <class GetIn, class PutOut>
  CTC_HD double multi_word_score(uint32_t uwid, bool eos, GetIn get_in, PutOut put_out) const {
    double sum = 0.0;
CTC_UNROLL
    for (int k = 0; k < MAX_LMS; ++k) {
      if ((uint32_t)k < tab.n_lms) {
        LmState in, out;
        get_in(k, &in);
        double sk;
        if (k == 0) {
          const uint32_t wi = tab.winfo0[uwid];
          const float base = lm_base_score(tab, in, wi & WI_ID_MASK, &out);
          double end = 0.0;
          if (eos && prm.score_boundary) {
            LmState tmp;
            end = (double)lm_base_score(tab, out, tab.eos_id, &tmp);
          }
          sk = lm_word_score_core(tab.uniset_nonempty != 0, prm.alpha, prm.beta, prm.unk, prm.log_base_change, base,
                                  (wi & WI_UNI_WORD) != 0, (wi & WI_LM_WORD) != 0, end, eos);
        } else {
          const LmExtra& x = tab.x[k > 0 ? k - 1 : 0];
          const uint32_t wi = x.winfo[uwid];
          const float base = lm_base_score(x, in, wi & WI_ID_MASK, &out);
          double end = 0.0;
          if (eos && x.score_boundary) {
            LmState tmp;
            end = (double)lm_base_score(x, out, x.eos_id, &tmp);
          }
          sk = lm_word_score_core(x.uniset_nonempty != 0, x.alpha, x.beta, x.unk, prm.log_base_change, base,
                                  (wi & WI_UNI_WORD) != 0, (wi & WI_LM_WORD) != 0, end, eos);
        }
        put_out(k, out);
        sum = sum + sk;
      }
    }
    return sum / (double)tab.n_lms;
  }

  CTC_HD static void copy_state(LmState* d, const LmState& a) {
    d->len = a.len;
CTC_UNROLL
    for (int k = 0; k < MAX_CTX; ++k) {
      d->words[k] = a.words[k];
      d->backoff[k] = a.backoff[k];
    }
  }

  CTC_HD void make_completion_from(const BeamSoA& b, int i, const TextNode& src, const LmState& src_state) {
    uint32_t idx = ctx.atomic_add(&L.scal[1], node_span());
    if (idx + node_span() > io.text_cap) {
      ctx.atomic_or(&L.scal[6], ST_TEXT_OVERFLOW);
      idx = io.text_cap - node_span();
    }
    TextNode& dst = io.text_nodes[idx];
    const uint32_t m2 = b.meta2[i];
    double raw = src.raw_lm;
    if (io.prof) {
      ctx.use(raw);
      tick<11>();
    }
    if (MULTI) {
      const TextNode* src_nodes = &src;
      TextNode* dst_nodes = &dst;
      raw = raw + multi_word_score(
                      b.word_id[i], false,
                      [&](int k, LmState* st) {
                        if (k == 0) copy_state(st, src_state);
                        else copy_state(st, src_nodes[k].state);
                      },
                      [&](int k, const LmState& st) { copy_state(&dst_nodes[k].state, st); });
    } else if (tab.has_lm) {
      float base = lm_base_score(tab, src_state, b.word_id[i], &dst.state);
      if (io.prof) {
        ctx.use((double)base);
        tick<12>();
      }
      raw = raw + lm_word_score(tab, prm, base, m2, 0.0, false);
    } else {
      dst.state.len = src.state.len;
CTC_UNROLL
      for (int k = 0; k < MAX_CTX; ++k) {
        dst.state.words[k] = src.state.words[k];
        dst.state.backoff[k] = src.state.backoff[k];
      }
    }
    const uint64_t wh = b.part_h[i];
    const uint64_t th = text_push(src.text_h, wh);
    const uint32_t cnt = src.hw_cnt + ((m2 & M2_HOT_COMPLETE) ? 1u : 0u);
    const double lmhw = raw + prm.hot_weight * (double)cnt;
    const uint32_t rc = src.ring_cnt + 1 > tab.n_hist ? tab.n_hist : src.ring_cnt + 1;
    // history ring (newest first) and its hash, without run-time indexed temporaries
    uint64_t hh = 0x9E3779B97F4A7C15ull + rc;
CTC_UNROLL
    for (int k = MAX_CTX - 1; k >= 0; --k) {
      uint64_t rk = k == 0 ? wh : ((uint32_t)k < rc ? src.ring[k > 0 ? k - 1 : 0] : 0ull);
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
    b.comp_node[i] = idx;
    b.c_text_h[i] = th;
    b.c_lm_hw[i] = lmhw;
    b.c_hist_h[i] = hh;
    tick<13>();
  }

  // ---- branch of candidate (survivor s, beam i): decoder.py:452,474,500,518 -----------------
  CTC_HD uint32_t branch_of(const BeamSoA& b, uint32_t tflags, uint32_t mode_word, uint32_t c, int i) const {
    if ((tflags & TK_BLANK) || last_char(b, i) == c) return 0;  // keep prefix (blank / repeat)
    uint32_t mode = mode_word & 0xFFu;
    if (mode == MODE_ALL_B) return BR_BOUNDARY;
    if (mode == MODE_FIRST_B) return (uint32_t)i == (mode_word >> 8) ? BR_BOUNDARY : BR_APPEND;
    if (mode == MODE_C) return BR_SPACE;
    return BR_APPEND;
  }

  // ---- one frame ---------------------------------------------------------------------------
  // Label constants of survivor s, read field by field: LDS for the first TOK_STAGE survivors of the
  // frame (staged by load_survivors), L2 beyond. (No struct copies: those would live in scratch.)
  struct TkView {
    const BeamDecoder* d;
    uint32_t s;
#define CTC_TK_FIELD(type, name, global_expr)                                            \
  CTC_HD type name() const {                                                             \
    if (s < (uint32_t)TOK_STAGE) return d->L.stok[s].name;                               \
    const uint32_t c = d->L.surv[s].id;                                                  \
    (void)c;                                                                             \
    return global_expr;                                                                  \
  }
    CTC_TK_FIELD(uint64_t, h_raw, d->tab.tok[c].h_raw)
    CTC_TK_FIELD(uint64_t, pow_raw, d->tab.tok[c].pow_raw)
    CTC_TK_FIELD(uint64_t, h_clean, d->tab.tok[c].h_clean)
    CTC_TK_FIELD(uint32_t, len_raw, d->tab.tok[c].len_raw)
    CTC_TK_FIELD(uint32_t, len_clean, d->tab.tok[c].len_clean)
    CTC_TK_FIELD(uint32_t, flags, d->tab.tok[c].flags)
    CTC_TK_FIELD(uint32_t, start_flags, d->tab.tok[c].start_flags)
    CTC_TK_FIELD(uint32_t, start_word_id, d->tab.tok[c].start_word_id)
    CTC_TK_FIELD(uint32_t, hot_min, (d->tab.tok_hot ? d->tab.tok_hot[c].min_len : 0u))
    CTC_TK_FIELD(uint32_t, hot_complete, (d->tab.tok_hot ? d->tab.tok_hot[c].complete : 0u))
#undef CTC_TK_FIELD
  };
  CTC_HD TkView tok_of(uint32_t s) const { return TkView{this, s}; }

  // stage label c's constants for survivor slot s (field-wise global loads -> LDS stores)
  CTC_HD void stage_tok(uint32_t s, uint32_t c) {
    const TokInfo& g = tab.tok[c];
    L.stok[s].h_raw = g.h_raw;
    L.stok[s].pow_raw = g.pow_raw;
    L.stok[s].h_clean = g.h_clean;
    L.stok[s].len_raw = g.len_raw;
    L.stok[s].len_clean = g.len_clean;
    L.stok[s].flags = g.flags;
    L.stok[s].start_flags = g.start_flags;
    L.stok[s].start_word_id = g.start_word_id;
    L.stok[s].hot_min = tab.tok_hot ? tab.tok_hot[c].min_len : 0u;
    L.stok[s].hot_complete = tab.tok_hot ? tab.tok_hot[c].complete : 0u;
  }

  // issue the loads of frame t's survivor list; they are consumed by load_survivors(t) one frame later
  CTC_HD void prefetch(int t) {
    pf_live = t < io.T;
    if (!pf_live) return;
    pf_cnt = io.surv_cnt[t];
    if (ctx.tid < prm.max_surv) {
      pf_id = io.surv_id[(size_t)t * prm.max_surv + ctx.tid];
      pf_lp = io.surv_lp[(size_t)t * prm.max_surv + ctx.tid];
    }
  }

  // second stage of the prefetch, issued later in the frame when the ids above have landed: the label
  // constants of the next frame's first TOK_STAGE survivors (their address depends on the id)
  CTC_HD void prefetch_tok() {
    if (!pf_live || ctx.tid >= TOK_STAGE || (uint32_t)ctx.tid >= pf_cnt) return;
    const TokInfo& g = tab.tok[pf_id];
    pf_tok.h_raw = g.h_raw;
    pf_tok.pow_raw = g.pow_raw;
    pf_tok.h_clean = g.h_clean;
    pf_tok.len_raw = g.len_raw;
    pf_tok.len_clean = g.len_clean;
    pf_tok.flags = g.flags;
    pf_tok.start_flags = g.start_flags;
    pf_tok.start_word_id = g.start_word_id;
    pf_tok.hot_min = tab.tok_hot ? tab.tok_hot[pf_id].min_len : 0u;
    pf_tok.hot_complete = tab.tok_hot ? tab.tok_hot[pf_id].complete : 0u;
  }

  CTC_HD void store_tok(uint32_t s) {  // survivor slot s <- the prefetched constants of this thread's label
    L.stok[s].h_raw = pf_tok.h_raw;
    L.stok[s].pow_raw = pf_tok.pow_raw;
    L.stok[s].h_clean = pf_tok.h_clean;
    L.stok[s].len_raw = pf_tok.len_raw;
    L.stok[s].len_clean = pf_tok.len_clean;
    L.stok[s].flags = pf_tok.flags;
    L.stok[s].start_flags = pf_tok.start_flags;
    L.stok[s].start_word_id = pf_tok.start_word_id;
    L.stok[s].hot_min = pf_tok.hot_min;
    L.stok[s].hot_complete = pf_tok.hot_complete;
  }

  CTC_HD uint32_t load_survivors(int t) {
    const uint32_t ns = pf_cnt;
    const uint16_t* ids = io.surv_id + (size_t)t * prm.max_surv;
    const double* lps = io.surv_lp + (size_t)t * prm.max_surv;
    if ((uint32_t)ctx.tid < ns) {
      L.surv[ctx.tid].id = pf_id;
      L.surv[ctx.tid].lp = pf_lp;
      if (ctx.tid < TOK_STAGE) store_tok((uint32_t)ctx.tid);
    }
    for (uint32_t s = ctx.tid + ctx.nt; s < ns; s += ctx.nt) {
      uint32_t id = ids[s];
      L.surv[s].id = id;
      L.surv[s].lp = lps[s];
      if (s < (uint32_t)TOK_STAGE) stage_tok(s, id);
    }
    return ns;
  }

  // Branch mode of every surviving label. For BPE vocabularies the force_next_break flag of
  // decoder.py:442,474-482 threads through the labels in iteration order; each label acts on the flag
  // as identity / clear / set, so the flag seen by label s is that of the last non-identity label
  // before it: one ballot pair per 64 labels instead of a serial walk.
  // One block of <= wave-width BPE labels, one per lane of the first wave: flags `fl` / id `c` of this lane's
  // label (TK_BLANK when the lane has none). Returns the lane's mode word; f = running force_next_break flag.
  CTC_HD uint32_t mode_block(const BeamSoA& b, uint32_t lane, uint32_t fl, uint32_t c, uint32_t& f, bool& need) {
    uint32_t first = (uint32_t)N;
    bool any = false;
    if (!(fl & TK_BLANK)) {
      int i = 0;
      while (i < N && last_char(b, i) == c) ++i;  // first beam that does not repeat the label
      first = (uint32_t)i;
      any = i < N;
    }
    const bool lead = (fl & TK_LEAD) != 0, trail = (fl & TK_TRAIL) != 0, blank = (fl & TK_BLANK) != 0;
    // effect on the flag: lead label with a taker -> set to `trail`; other label with a taker and no
    // trailing mark -> clear; everything else -> identity
    const bool sets = !blank && any && lead && trail;
    const bool clears = !blank && any && ((lead && !trail) || (!lead && !trail));
    const uint64_t m_one = ctx.ballot(sets);
    const uint64_t m_set = m_one | ctx.ballot(clears);
    const uint64_t prior = m_set & ((lane >= 64u) ? ~0ull : ((1ull << lane) - 1ull));
    uint32_t f_in = f;
    if (prior) f_in = (uint32_t)((m_one >> (63 - ctx.clz64(prior))) & 1ull);
    uint32_t mode = MODE_D;
    if (blank) mode = MODE_A;
    else if (lead) mode = MODE_ALL_B;
    else if (f_in && any) mode = trail ? MODE_ALL_B : MODE_FIRST_B;
    need = need || (ctx.ballot(!blank && any && mode != MODE_D) != 0ull);
    if (m_set) f = (uint32_t)((m_one >> (63 - ctx.clz64(m_set))) & 1ull);
    return mode | (first << 8);
  }

  CTC_HD void compute_modes(uint32_t ns) {
    const BeamSoA b = beams_at(cur);
    if (!tab.is_bpe) {
      for (uint32_t s = ctx.tid; s < ns; s += ctx.nt) {
        uint32_t fl = tok_of(s).flags();
        uint32_t mode = (fl & TK_BLANK) ? MODE_A : ((fl & TK_SPACE) ? MODE_C : MODE_D);
        if (mode == MODE_C) L.scal[4] = 1u;
        L.surv[s].mode = mode | ((uint32_t)N << 8);
      }
      ctx.sync();
      return;
    }
    const uint32_t W = (uint32_t)ctx.wave_width();
    if ((uint32_t)ctx.tid < W) {  // the first wave
      const uint32_t lane = (uint32_t)ctx.tid;
      uint32_t f = L.scal[3];
      bool need = false;
      for (uint32_t base = 0; base < ns; base += W) {
        const uint32_t s = base + lane;
        const uint32_t fl = s < ns ? tok_of(s).flags() : (uint32_t)TK_BLANK;
        const uint32_t c = s < ns ? L.surv[s].id : 0u;
        const uint32_t mw = mode_block(b, lane, fl, c, f, need);
        if (s < ns) L.surv[s].mode = mw;
      }
      if (lane == 0) {
        L.scal[3] = f;
        if (need) L.scal[4] = 1u;
      }
    }
    ctx.sync();
  }

  // The usual frame (no more survivors than one wave has lanes, all of them staged): the first wave writes
  // the survivor list AND works out the modes from the prefetched registers -- one phase, one barrier.
  CTC_HD void load_survivors_and_modes(uint32_t ns) {
    const BeamSoA b = beams_at(cur);
    if ((uint32_t)ctx.tid < (uint32_t)ctx.wave_width()) {
      const uint32_t lane = (uint32_t)ctx.tid;
      const bool mine = lane < ns;
      const uint32_t fl = mine ? pf_tok.flags : (uint32_t)TK_BLANK;
      uint32_t mw;
      if (!tab.is_bpe) {
        const uint32_t mode = (fl & TK_BLANK) ? MODE_A : ((fl & TK_SPACE) ? MODE_C : MODE_D);
        if (mine && mode == MODE_C) L.scal[4] = 1u;
        mw = mode | ((uint32_t)N << 8);
      } else {
        uint32_t f = L.scal[3];
        bool need = false;
        mw = mode_block(b, lane, fl, pf_id, f, need);
        if (lane == 0) {
          L.scal[3] = f;
          if (need) L.scal[4] = 1u;
        }
      }
      if (mine) {
        L.surv[lane].id = pf_id;
        L.surv[lane].lp = pf_lp;
        L.surv[lane].mode = mw;
        store_tok(lane);
      }
    }
    ctx.sync();
  }

  // push one merged+scored candidate into the pool
  CTC_HD void pool_push(double score, double logit, uint32_t arrival, uint32_t donor, uint32_t wid = 0,
                        uint32_t m2 = 0) {
    uint32_t k = ctx.atomic_add(&L.scal[0], 1u);
    if (k >= (uint32_t)shape.pool) {
      ctx.atomic_or(&L.scal[6], ST_POOL_OVERFLOW);
      return;
    }
    L.p_score[k] = score;
    L.p_logit[k] = logit;
    L.p_arr[k] = arrival;
    L.p_don[k] = donor;
    L.p_wid[k] = wid;
    L.p_m2[k] = m2;
  }

  CTC_HD void clear_table() {
    for (int k = ctx.tid; k < shape.tab; k += ctx.nt) L.table[k] = 0;
  }

  // insert candidate q (keys already in ck_*); candidates of one label occupy `group` consecutive
  // indices and only merge with each other (the key contains last_char). Returns the representative.
  CTC_HD uint32_t table_insert(uint32_t q, uint32_t group, uint32_t* my_slot = nullptr) {
    uint32_t mask = (uint32_t)(shape.tab - 1);
    uint64_t kt = L.ck_text[q], kp = L.ck_part[q];
    uint32_t g = q / group;
    // a slot holds  q + 1 (12 bits: cand <= 2048)  |  the hash's upper 20 bits: a candidate that finds the slot taken
    // compares those first and reads the other's keys (two more dependent LDS round trips of the probe chain, which the
    // slowest lane of the wave dictates) only when they agree
    const uint32_t hsh = key_slot_hash(kt, kp, g);
    uint32_t slot = hsh & mask;
    const uint32_t mine = (q + 1) | (hsh & 0xFFFFF000u);
    for (;;) {
      uint32_t old = ctx.atomic_cas(&L.table[slot], 0u, mine);
      if (old == 0) {
        if (my_slot) *my_slot = slot;
        return q;
      }
      if (((old ^ mine) & 0xFFFFF000u) == 0) {
        const uint32_t r = (old & 0xFFFu) - 1;
        if (L.ck_text[r] == kt && L.ck_part[r] == kp && r / group == g) return r;
      }
      slot = (slot + 1) & mask;
    }
  }

  // Select the pool entries with score >= thr, ordered by (score desc, arrival asc); L.sel[r] = pool
  // index of rank r for r < min(count, beam_width). Returns the count (may exceed beam_width).
  // Entries are first compacted (typically ~25 of ~100 survive the threshold); small sets are ranked by
  // counting (one LDS sweep, no barriers), large ones by a bucket histogram.
  // (scal[5], the compaction counter, is zero on entry: init() and the end of this function see to it.)
  // with_hist: also leave the history-prune key of rank r in hk_*[r] (decoder.py:250-254).
  // !ordered: the caller wants the best min(count, beam_width) entries as a set -- a large set's sel[] then comes in any order.
  CTC_HD uint32_t sort_pool(uint32_t pool_n, double thr, bool with_hist, bool ordered = true) {
    tick<17>();
    for (uint32_t k = ctx.tid; k < pool_n; k += ctx.nt) {
      const double sc = L.p_score[k];
      if (sc >= thr) {
        const uint32_t pos = ctx.atomic_add(&L.scal[5], 1u);
        L.s_k0[pos] = score_sort_key(sc);
        L.s_k1[pos] = ((uint64_t)L.p_arr[k] << 32) | k;
        if (with_hist && pos < 256u) {
          // (history, partial, last_char) folded to 64 bits: equality of the folds stands in for equality of
          // the triple (its members are 61/64-bit string hashes already)
          uint64_t hh, ph;
          uint32_t cc;
          hist_key(k, &hh, &ph, &cc);
          L.s_hk[pos] = mix64(hh ^ mix64(ph + 0x9E3779B97F4A7C15ull * (uint64_t)(cc + 1u)));
        }
      }
    }
    ctx.sync();
    tick<18>();
    const uint32_t n = L.scal[5];
    const uint32_t want = (uint32_t)prm.beam_width;
    if (n <= 256u) {
      for (uint32_t e = ctx.tid; e < n; e += ctx.nt) {
        const uint64_t a0 = L.s_k0[e], a1 = L.s_k1[e];
        uint32_t rank = 0, same = 0, dup = 0;
        uint32_t j = 0;
        if (with_hist) {
          // history prune in the same sweep: an entry is dropped when a better-ranked one has the same
          // (history, partial, last_char) -- first in sorted order wins (decoder.py:248-257)
          const uint64_t ah = L.s_hk[e];
          for (; j + 2 <= n; j += 2) {
            const uint64_t x0 = L.s_k0[j], x1 = L.s_k0[j + 1];
            const uint64_t h0 = L.s_hk[j], h1 = L.s_hk[j + 1];
            rank += (x0 < a0 ? 1u : 0u) + (x1 < a0 ? 1u : 0u);
            same += (x0 == a0 ? 1u : 0u) + (x1 == a0 ? 1u : 0u);
            dup |= ((x0 < a0 && h0 == ah) ? 1u : 0u) | ((x1 < a0 && h1 == ah) ? 1u : 0u);
          }
          for (; j < n; ++j) {
            const uint64_t x = L.s_k0[j];
            rank += x < a0 ? 1u : 0u;
            same += x == a0 ? 1u : 0u;
            dup |= (x < a0 && L.s_hk[j] == ah) ? 1u : 0u;
          }
          if (same > 1u) {  // equal scores (rare): the earlier arrival ranks first (heapq.nlargest is stable)
            for (j = 0; j < n; ++j) {
              const bool before = L.s_k0[j] == a0 && L.s_k1[j] < a1;
              rank += before ? 1u : 0u;
              dup |= (before && L.s_hk[j] == ah) ? 1u : 0u;
            }
          }
        } else {
          for (; j + 4 <= n; j += 4) {  // four independent LDS reads in flight per step
            uint64_t x0 = L.s_k0[j], x1 = L.s_k0[j + 1], x2 = L.s_k0[j + 2], x3 = L.s_k0[j + 3];
            rank += (x0 < a0 ? 1u : 0u) + (x1 < a0 ? 1u : 0u) + (x2 < a0 ? 1u : 0u) + (x3 < a0 ? 1u : 0u);
            same += (x0 == a0 ? 1u : 0u) + (x1 == a0 ? 1u : 0u) + (x2 == a0 ? 1u : 0u) + (x3 == a0 ? 1u : 0u);
          }
          for (; j < n; ++j) {
            const uint64_t x = L.s_k0[j];
            rank += x < a0 ? 1u : 0u;
            same += x == a0 ? 1u : 0u;
          }
          if (same > 1u) {
            for (j = 0; j < n; ++j) rank += (L.s_k0[j] == a0 && L.s_k1[j] < a1) ? 1u : 0u;
          }
        }
        if (rank < want) {
          L.sel[rank] = (uint32_t)(a1 & 0xFFFFFFFFu);
          L.keep[rank] = dup ? 0u : 1u;
        }
      }
      ctx.sync();
      if (ctx.tid == 0) L.scal[5] = 0;
      return n;
    }
    // Large sets (stress inputs, n > 256 > want): exact top-`want` selection by a bucket histogram.
    // Scores are cut into 1024 monotone buckets over [max - width, max]; whole buckets above the one
    // holding the want-th entry are taken, that boundary bucket is resolved by counting, and only the
    // `want` chosen entries are ranked. The (all-zero) merge table is the histogram and is handed back
    // zeroed; the compact index lists live in the idle rmax/rcnt arrays.
    {
      const double mx = sortable_to_max();
      double lo = thr;
      if (!(lo > mx - 64.0)) lo = mx - 64.0;  // clamp (keeps the map monotone; also covers thr = -inf)
      const double scale = mx > lo ? 1023.0 / (mx - lo) : 0.0;
      auto bucket_of = [&](uint32_t e) -> uint32_t {
        const double sc = L.p_score[(uint32_t)(L.s_k1[e] & 0xFFFFFFFFull)];
        double x = (mx - sc) * scale;
        uint32_t bkt = x >= 1023.0 ? 1023u : (uint32_t)x;
        return bkt;
      };
      LPtr<uint32_t> slist = L.rmax;  // chosen entries (compact indices): rmax[0 .. 256)
      LPtr<uint32_t> blist;           // boundary-bucket entries: rmax[256 ..) running on into rcnt (768 slots)
      blist.p = L.rmax.p + 256;
      if (ctx.tid == 0) {
        L.scal[10] = 0;  // chosen so far
        L.scal[11] = 0;  // boundary entries listed
        L.scal[12] = 0;  // boundary bucket
        L.scal[13] = 0;  // entries in the buckets above it
      }
      for (uint32_t e = ctx.tid; e < n; e += ctx.nt) ctx.atomic_add(&L.table[bucket_of(e)], 1u);
      ctx.sync();
      // 64 partial sums of 16 buckets each, then the boundary bucket. (Every loop below reads from addresses that do not
      // depend on what it has read: unrolled, the LDS reads of a step are in flight together -- a thread that walks them
      // one by one pays a round trip each, and these are the only threads at work.)
      for (uint32_t t = ctx.tid; t < 64u; t += ctx.nt) {
        uint32_t c = 0;
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) c += L.table[t * 16u + k];
        L.part[t] = c;
      }
      ctx.sync();
      for (uint32_t t = ctx.tid; t < 64u; t += ctx.nt) {
        uint32_t before = 0;
#pragma unroll
        for (uint32_t k = 0; k < 64u; ++k) {
          const uint32_t v = L.part[k];
          before += k < t ? v : 0u;
        }
        const uint32_t mine = L.part[t];
        if (before < want && want <= before + mine) {  // the want-th entry lies in my 16 buckets
          uint32_t h[16];
#pragma unroll
          for (uint32_t k = 0; k < 16u; ++k) h[k] = L.table[t * 16u + k];
          uint32_t cum = before, bk = 0, bcum = 0;
          bool found = false;
#pragma unroll
          for (uint32_t k = 0; k < 16u; ++k) {
            if (!found && cum + h[k] >= want) {
              found = true;
              bk = k;
              bcum = cum;
            }
            cum += h[k];
          }
          L.scal[12] = t * 16u + bk;
          L.scal[13] = bcum;
        }
      }
      ctx.sync();
      const uint32_t bstar = L.scal[12], above = L.scal[13];
      const uint32_t need = want - above;  // entries still to take from the boundary bucket (>= 1)
      for (uint32_t e = ctx.tid; e < n; e += ctx.nt) {
        const uint32_t bkt = bucket_of(e);
        if (bkt < bstar) slist[ctx.atomic_add(&L.scal[10], 1u)] = e;
        else if (bkt == bstar) blist[ctx.atomic_add(&L.scal[11], 1u)] = e;
      }
      ctx.sync();
      const uint32_t m = L.scal[11];
      for (uint32_t i = ctx.tid; i < m; i += ctx.nt) {  // rank inside the boundary bucket
        const uint32_t e = blist[i];
        const uint64_t a0 = L.s_k0[e], a1 = L.s_k1[e];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < m; ++j) {
          const uint32_t e2 = blist[j];
          const uint64_t b0 = L.s_k0[e2], b1 = L.s_k1[e2];
          rank += ((b0 < a0) || (b0 == a0 && b1 < a1)) ? 1u : 0u;
        }
        if (rank < need) slist[ctx.atomic_add(&L.scal[10], 1u)] = e;
      }
      ctx.sync();
      for (uint32_t k = ctx.tid; k < 1024u; k += ctx.nt) L.table[k] = 0;  // hand the table back zeroed
      if (!ordered) {  // (uniform)
        for (uint32_t i = ctx.tid; i < want; i += ctx.nt) L.sel[i] = (uint32_t)(L.s_k1[slist[i]] & 0xFFFFFFFFull);
        ctx.sync();
        if (ctx.tid == 0) L.scal[5] = 0;
        return n;
      }
      // rank the `want` chosen entries among themselves. Their keys are first copied side by side behind the sort
      // buffer's entries (sortn - pool >= cand - bw >= want slots are free there), then the want x want comparisons are
      // spread over all threads, four to a row, the partial counts summed where the list was.
      LPtr<uint64_t> ca0, ca1;
      ca0.p = L.s_k0.p + shape.pool;
      ca1.p = L.s_k1.p + shape.pool;
      for (uint32_t i = ctx.tid; i < want; i += ctx.nt) {
        const uint32_t e = slist[i];
        ca0[i] = L.s_k0[e];
        ca1[i] = L.s_k1[e];
        slist[i] = 0;  // (the list has served: from here on, the rank of entry i)
      }
      ctx.sync();
      {
        const uint32_t G = ctx.nt >= 4 ? 4u : 1u;  // threads per row
        const uint32_t g = (uint32_t)ctx.tid % G, rows = (uint32_t)ctx.nt / G;
        for (uint32_t i = (uint32_t)ctx.tid / G; i < want; i += rows) {
          const uint64_t a0 = ca0[i], a1 = ca1[i];
          uint32_t rank = 0;
          uint32_t j = g;
          for (; j + 3u * G < want; j += 4u * G) {  // four independent pairs of LDS reads in flight per step
            const uint64_t b0 = ca0[j], b1 = ca0[j + G], b2 = ca0[j + 2u * G], b3 = ca0[j + 3u * G];
            const uint64_t c0 = ca1[j], c1 = ca1[j + G], c2 = ca1[j + 2u * G], c3 = ca1[j + 3u * G];
            rank += ((b0 < a0) || (b0 == a0 && c0 < a1)) ? 1u : 0u;
            rank += ((b1 < a0) || (b1 == a0 && c1 < a1)) ? 1u : 0u;
            rank += ((b2 < a0) || (b2 == a0 && c2 < a1)) ? 1u : 0u;
            rank += ((b3 < a0) || (b3 == a0 && c3 < a1)) ? 1u : 0u;
          }
          for (; j < want; j += G) {
            const uint64_t b0 = ca0[j], c0 = ca1[j];
            rank += ((b0 < a0) || (b0 == a0 && c0 < a1)) ? 1u : 0u;
          }
          if (rank) ctx.atomic_add(&slist[i], rank);
        }
      }
      ctx.sync();
      for (uint32_t i = ctx.tid; i < want; i += ctx.nt) {
        const uint64_t a1 = ca1[i];
        const uint32_t rank = slist[i];
        const uint32_t idx = (uint32_t)(a1 & 0xFFFFFFFFull);
        L.sel[rank] = idx;
        L.keep[rank] = 1u;
        if (with_hist) {
          uint64_t hh, ph;
          uint32_t cc;
          hist_key(idx, &hh, &ph, &cc);
          L.hk_h[rank] = hh;
          L.hk_p[rank] = ph;
          L.hk_c[rank] = cc;
        }
      }
    }
    ctx.sync();
    if (ctx.tid == 0) L.scal[5] = 0;
    return n;
  }

  // keep only the best `beam_width` pool entries (exact: pruning is monotone)
  CTC_HD void prune_pool() {
    tick<6>();
    uint32_t pool_n = L.scal[0];
    double mx = sortable_to_max();
    if (ctx.tid == 0) L.smax[3] = 0;  // (ordered before the atomics below by the barriers inside sort_pool)
    uint32_t n = sort_pool(pool_n, mx + prm.beam_prune_logp, false, false);
    if (n > (uint32_t)prm.beam_width) n = (uint32_t)prm.beam_width;
    // gather the survivors through the temp arrays, then rewrite the pool front; the lowest score kept is the one a later
    // candidate has to beat
    for (uint32_t k0 = 0; k0 < n; k0 += (uint32_t)ctx.nt) {  // (uniform trip count: a wave reduction inside)
      const uint32_t k = k0 + (uint32_t)ctx.tid;
      uint64_t low = 0;
      if (k < n) {
        uint32_t idx = L.sel[k];
        const double sc = L.p_score[idx];
        L.g_score[k] = sc;
        L.g_logit[k] = L.p_logit[idx];
        L.g_arr[k] = L.p_arr[idx];
        L.g_don[k] = L.p_don[idx];
        L.g_wid[k] = L.p_wid[idx];
        L.g_m2[k] = L.p_m2[idx];
        low = score_sort_key(sc);  // (grows as the score falls; never 0 for a finite score)
      }
      low = ctx.wave_max_u64(low);
      if (ctx.is_wave_leader() && low != 0) ctx.atomic_max64(&L.smax[3], low);
    }
    ctx.sync();
    for (uint32_t k = ctx.tid; k < n; k += ctx.nt) {
      L.p_score[k] = L.g_score[k];
      L.p_logit[k] = L.g_logit[k];
      L.p_arr[k] = L.g_arr[k];
      L.p_don[k] = L.g_don[k];
      L.p_wid[k] = L.g_wid[k];
      L.p_m2[k] = L.g_m2[k];
    }
    if (ctx.tid == 0) {
      L.scal[0] = n;
      // from now on only a candidate that beats the current beam_width-th best can still matter: later
      // candidates arrive later, so an equal score ranks behind the beam_width entries kept here
      if (n >= (uint32_t)prm.beam_width) L.smax[2] = ~L.smax[3];  // = asc_key(lowest score kept)
    }
    ctx.sync();
    clear_table();  // the gather temp may overlap the (all-zero between chunks) merge table
    ctx.sync();
    tick<23>();
  }

  CTC_HD double sortable_to_max() const {
    // smax holds max over pushed scores as an ascending-sortable key
    uint64_t u = L.smax[0];
    uint64_t bits = (u >> 63) ? (u & ~(1ull << 63)) : ~u;
    union { double d; uint64_t u; } c;
    c.u = bits;
    return c.d;
  }
  CTC_HD static uint64_t asc_key(double s) {
    if (s == 0.0) s = 0.0;
    union { double d; uint64_t u; } c;
    c.d = s;
    return (c.u >> 63) ? ~c.u : (c.u | (1ull << 63));
  }


  // The partial word a candidate ends up with, seen through the prefix / hot-word tables:
  // length, table flags (meta2 layout), LM word id and partial-word score.
  struct PartView {
    uint32_t pl, m2, wid;
    double ps;
  };
  CTC_HD PartView new_partial(const BeamSoA& b, int i, const TkView& tk, uint32_t br,
                              uint64_t new_part_h, bool have_pre, const PrefixEntry& pre_p,
                              const HotEntry& pre_h) const {
    PartView v;
    if (br == 0) {  // blank / repeat: unchanged
      v.pl = plen(b, i);
      v.m2 = b.meta2[i];
      v.wid = b.word_id[i];
      v.ps = b.pscore[i];
    } else if (br == BR_BOUNDARY && tk.len_clean() > 0) {  // a new word starts with the clean label
      uint32_t hmin = tk.hot_min();
      uint32_t hcomp = tk.hot_complete();
      v.pl = tk.len_clean();
      v.m2 = (tk.start_flags() & (PF_PARTIAL_MASK | PF_ON_TABLE)) | (hmin ? M2_HOT_ON : 0u) | (hcomp ? M2_HOT_COMPLETE : 0u) | (hmin << 8);
      v.wid = tk.start_word_id();
      v.ps = partial_score(tab, prm, tk.start_flags(), hmin, v.pl);
    } else if (br == BR_APPEND) {
      uint32_t m2 = b.meta2[i];
      uint32_t pf = 0, nw = 0, hmin = 0, hcomp = 0;
      // first probe of both tables issued together (one memory round trip instead of two)
      const bool want_p = (m2 & PF_ON_TABLE) && tab.prefixes && new_part_h != 0;
      const bool want_h = (m2 & M2_HOT_ON) && tab.hot && new_part_h != 0;
      const uint64_t hk = table_slot(new_part_h);
      uint64_t sp = hk & tab.prefix_mask, sh = hk & tab.hot_mask;
      PrefixEntry ep = {0, 0, 0};
      HotEntry eh = {0, 0, 0};
      if (want_p) ep = have_pre ? pre_p : tab.prefixes[sp];
      if (want_h) eh = have_pre ? pre_h : tab.hot[sh];
      bool on = false, hon = false;
      if (want_p) {
        while (ep.key != new_part_h && ep.key != 0) {
          sp = (sp + 1) & tab.prefix_mask;
          ep = tab.prefixes[sp];
        }
        on = ep.key == new_part_h;
        nw = ep.word_id;
        pf = ep.flags;
      }
      if (want_h) {
        while (eh.key != new_part_h && eh.key != 0) {
          sh = (sh + 1) & tab.hot_mask;
          eh = tab.hot[sh];
        }
        hon = eh.key == new_part_h;
        hmin = eh.min_len;
        hcomp = eh.complete;
      }
      v.pl = plen(b, i) + tk.len_raw();
      v.m2 = (on ? (PF_ON_TABLE | (pf & PF_PARTIAL_MASK)) : 0u) | (hon ? M2_HOT_ON : 0u) | ((hon && hcomp) ? M2_HOT_COMPLETE : 0u) |
             ((hon ? hmin : 0u) << 8);
      v.wid = on ? nw : 0;
      v.ps = partial_score(tab, prm, on ? pf : 0u, hon ? hmin : 0u, v.pl);
    } else {  // space, or a bare boundary mark: the open word is empty
      v.pl = 0;
      v.m2 = EMPTY_PARTIAL_M2;
      v.wid = 0;
      v.ps = 0.0;
    }
    return v;
  }

  // Candidate generation + merge + scoring for survivors [s0, s1)
  CTC_HD void process_chunk(uint32_t s0, uint32_t s1) {
    const BeamSoA b = beams_at(cur);
    uint32_t Q = (s1 - s0) * (uint32_t)N;
    // conservative running threshold from the chunks already seen (nobody writes smax here)
    double thr_prev = sortable_to_max() + prm.beam_prune_logp;
    const uint64_t kth_key = L.smax[2];  // 0 until a pool prune has fixed a beam_width-th best
    // first table probes of this thread's first candidate (the one it scores in S), issued here so that
    // they are in flight across the merge phase
    PrefixEntry pre_p = {0, 0, 0};
    HotEntry pre_h = {0, 0, 0};
    bool have_pre = false;
    // G1: keys
    for (uint32_t q = ctx.tid; q < Q; q += ctx.nt) {
      uint32_t s = s0 + q / (uint32_t)N;
      int i = (int)(q % (uint32_t)N);
      uint32_t c = L.surv[s].id;
      const TkView tk = tok_of(s);
      uint32_t br = branch_of(b, tk.flags(), L.surv[s].mode, c, i);
      uint64_t kt = b.text_h[i], kp = b.part_h[i];
      if (br == BR_BOUNDARY || br == BR_SPACE) {
        if (plen(b, i) > 0) kt = b.c_text_h[i];
        kp = br == BR_BOUNDARY ? tk.h_clean() : 0;
      } else if (br == BR_APPEND) {
        kp = str_concat(kp, tk.pow_raw(), tk.h_raw());
        if (q == (uint32_t)ctx.tid && kp != 0) {
          const uint32_t m2 = b.meta2[i];
          const uint64_t hk = table_slot(kp);
          if ((m2 & PF_ON_TABLE) && tab.prefixes) pre_p = tab.prefixes[hk & tab.prefix_mask];
          if ((m2 & M2_HOT_ON) && tab.hot) pre_h = tab.hot[hk & tab.hot_mask];
          have_pre = true;
        }
      }
      L.ck_text[q] = kt;
      L.ck_part[q] = kp;
      L.c_logit[q] = b.logit[i] + L.surv[s].lp;
      L.rmin[q] = 0xFFFFFFFFu;
      L.rmax[q] = 0;
      L.rcnt[q] = 0;
    }
    ctx.sync();
    tick<3>();
    // G2: merge table (candidates of one label only ever merge with each other)
    // (a chunk of at most one candidate per thread -- the usual case -- hands the table back clean without
    // a sweep: each thread remembers the slot it filled and zeroes it once the merge phase is over)
    const bool one_pass = Q <= (uint32_t)ctx.nt;
    uint32_t my_slot = 0xFFFFFFFFu;
    for (uint32_t q = ctx.tid; q < Q; q += ctx.nt) {
      uint32_t r = table_insert(q, (uint32_t)N, &my_slot);
      L.crep[q] = r;
      ctx.atomic_min(&L.rmin[r], q);
      ctx.atomic_max(&L.rmax[r], q);
      ctx.atomic_add(&L.rcnt[r], 1u);
    }
    ctx.sync();
    tick<4>();
    // S: owners fold, score, push. The loop keeps every wave converged (uniform trip count) so the
    // running maximum is reduced inside the wave and published with ONE LDS atomic per wave.
    if (one_pass && my_slot != 0xFFFFFFFFu) L.table[my_slot] = 0;
    for (uint32_t base = 0; base < Q; base += (uint32_t)ctx.nt) {
      const uint32_t q = base + (uint32_t)ctx.tid;
      uint64_t my_key = 0;  // below every real key
      bool owner = false;
      if (q < Q) {
        const uint32_t r = L.crep[q];
        owner = L.rmin[r] == q;
      }
      if (owner) {
        const uint32_t r = L.crep[q];
        uint32_t qmax = L.rmax[r], cnt = L.rcnt[r];
        double lg = L.c_logit[q];
        if (cnt == 2) {
          lg = lse2(lg, L.c_logit[qmax]);
        } else if (cnt > 2) {
          for (uint32_t q2 = q + 1; q2 <= qmax; ++q2)
            if (L.crep[q2] == r) lg = lse2(lg, L.c_logit[q2]);
        }
        uint32_t s = s0 + q / (uint32_t)N;
        int i = (int)(q % (uint32_t)N);
        uint32_t c = L.surv[s].id;
        const TkView tk = tok_of(s);
        uint32_t br = branch_of(b, tk.flags(), L.surv[s].mode, c, i);
        double lmhw = b.lm_hw[i];
        if ((br == BR_BOUNDARY || br == BR_SPACE) && plen(b, i) > 0) lmhw = b.c_lm_hw[i];
        if (io.prof) {
          ctx.use(lg + lmhw);
          tick<14>();
        }
        PartView pv = new_partial(b, i, tk, br, L.ck_part[q], have_pre && base == 0, pre_p, pre_h);
        if (io.prof) {
          ctx.use(pv.ps);
          tick<15>();
        }
        double score = total_score(tab, lg, lmhw, pv.ps, pv.pl);
        my_key = asc_key(score);
        if (score >= thr_prev && my_key > kth_key) {
          uint32_t imax = qmax % (uint32_t)N;
          pool_push(score, lg, s * (uint32_t)N + (uint32_t)i, (s << 8) | imax, pv.wid, pv.m2);
        }
        tick<16>();
      }
      const uint64_t wave_key = ctx.wave_max_u64(my_key);
      if (ctx.is_wave_leader() && wave_key != 0) ctx.atomic_max64(&L.smax[0], wave_key);
    }
    if (flush_nodes) {  // (uniform) see step(): completes this frame's TextNode stores
      ctx.sync_mem();
      flush_nodes = false;
    } else {
      ctx.sync();
    }
    tick<5>();
    if (!one_pass) {
      clear_table();
      ctx.sync();
    }
    tick<6>();
  }

  // Build beam `dst` of the next table from pool entry `idx` (payload = the donor, i.e. the
  // last-arriving duplicate: decoder.py:221-223)
  CTC_HD void build_beam(const BeamSoA& nb, int dst, uint32_t idx, int frame) {
    const BeamSoA b = beams_at(cur);
    uint32_t don = L.p_don[idx];
    uint32_t s = don >> 8;
    int i = (int)(don & 0xFFu);
    uint32_t c = L.surv[s].id;
    const TkView tk = tok_of(s);
    uint32_t br = branch_of(b, tk.flags(), L.surv[s].mode, c, i);
    uint32_t pl = plen(b, i);
    uint64_t th = b.text_h[i], ph = b.part_h[i], hh = b.hist_h[i];
    double lmhw = b.lm_hw[i];
    uint32_t tnode = b.text_node[i], cnode = b.comp_node[i];
    int32_t pst = b.pstart[i], pen = b.pend[i];
    uint32_t enode = b.emit_node[i], depth = b.depth[i];
    uint64_t cth = b.c_text_h[i], chh = b.c_hist_h[i];
    double clm = b.c_lm_hw[i];
    // the new partial word as seen through the tables was resolved when the candidate was scored
    uint32_t m2 = L.p_m2[idx], wid = L.p_wid[idx];
    uint32_t npl = pl;
    if (br == 0) {
      if (!(tk.flags() & TK_BLANK)) pen = frame + 1;  // decoder.py:453-461
    } else {
      int32_t wst = pst, wen = pen;
      if (br == BR_BOUNDARY || br == BR_SPACE) {
        if (pl > 0) {  // the open word is completed (decoder.py:483-495, 501-515)
          th = cth;
          hh = chh;
          lmhw = clm;
          tnode = cnode;
        }
        if (br == BR_BOUNDARY) {
          ph = tk.h_clean();
          npl = tk.len_clean();
          pst = frame;
          pen = frame + 1;
        } else {
          ph = 0;
          npl = 0;
          pst = -1;
          pen = -1;
        }
      } else {  // BR_APPEND (decoder.py:518-534)
        ph = str_concat(ph, tk.pow_raw(), tk.h_raw());
        npl = pl + tk.len_raw();
        pst = pst < 0 ? frame : pst;
        pen = frame + 1;
      }
      cnode = 0;
      uint32_t e = ctx.atomic_add(&L.scal[2], 1u);
      if (e >= io.emit_cap) {
        ctx.atomic_or(&L.scal[6], ST_EMIT_OVERFLOW);
        e = io.emit_cap - 1;
      }
      EmitNode en;
      en.parent = enode;
      en.tok_branch = c | (br << 16);
      en.wstart = wst;
      en.wend = wen;
      io.emit_nodes[e] = en;
      enode = e;
      depth += 1;
    }
    double ps = 0.0;
    if (npl > 0) ps = partial_score(tab, prm, m2 & PF_PARTIAL_MASK, (m2 & M2_HOT_ON) ? ((m2 >> 8) & 0xFFFFu) : 0u, npl);
    nb.logit[dst] = L.p_logit[idx];
    nb.lm_hw[dst] = lmhw;
    nb.pscore[dst] = ps;
    nb.c_lm_hw[dst] = clm;
    nb.text_h[dst] = th;
    nb.part_h[dst] = ph;
    nb.hist_h[dst] = hh;
    nb.c_text_h[dst] = cth;
    nb.c_hist_h[dst] = chh;
    nb.text_node[dst] = tnode;
    nb.comp_node[dst] = cnode;
    nb.emit_node[dst] = enode;
    nb.word_id[dst] = wid;
    nb.meta1[dst] = c | (npl << 16);
    nb.meta2[dst] = m2;
    nb.depth[dst] = depth;
    nb.pstart[dst] = pst;
    nb.pend[dst] = pen;
  }

  // history-prune key of pool entry idx (decoder.py:250-254): (last words, partial, last_char)
  CTC_HD void hist_key(uint32_t idx, uint64_t* hh, uint64_t* ph, uint32_t* cc) const {
    const BeamSoA b = beams_at(cur);
    uint32_t don = L.p_don[idx];
    uint32_t s = don >> 8;
    int i = (int)(don & 0xFFu);
    uint32_t c = L.surv[s].id;
    const TkView tk = tok_of(s);
    uint32_t br = branch_of(b, tk.flags(), L.surv[s].mode, c, i);
    uint64_t h = b.hist_h[i], p = b.part_h[i];
    if (br == BR_BOUNDARY || br == BR_SPACE) {
      if (plen(b, i) > 0) h = b.c_hist_h[i];
      p = br == BR_BOUNDARY ? tk.h_clean() : 0;
    } else if (br == BR_APPEND) {
      p = str_concat(p, tk.pow_raw(), tk.h_raw());
    }
    *hh = h;
    *ph = p;
    *cc = c;
  }

  // ---- runs of single-label frames ----------------------------------------------------------------
  // A frame whose only survivor is the label every live beam already ends in (a blank after blanks, a held token)
  // extends every beam in place (decoder.py:452-471): logit += p and, for a token, the end frame of the open word.
  // Nothing merges (the merge and history keys are those of the previous frame, which left them distinct), no word
  // completes, every score moves by the same p -- up to fp rounding, so each frame's new scores are checked to be
  // still sorted and above the threshold; the first frame where they are not, or with another survivor set, ends the
  // run and takes the full path. Same rule as WaveDecoder::label_run (beam_wave.h). Returns the first frame not
  // consumed (t: none). Scratch: c_logit / p_score (new logits / scores), p_logit (look-ahead window),
  // scal[14] (stop flag), scal[15] (window length).
  CTC_HD int label_run(int t) {
    const BeamSoA b = beams_at(cur);
    if (ctx.tid == 0) {
      L.surv[0].id = pf_id;
      L.surv[0].lp = pf_lp;
      L.stok[0].flags = pf_tok.flags;
      L.scal[14] = 0;
    }
    ctx.sync();
    const uint32_t lab = L.surv[0].id;
    for (int i = ctx.tid; i < N; i += ctx.nt)
      if (last_char(b, i) != lab) L.scal[14] = 1u;
    ctx.sync();
    if (L.scal[14]) return t;
    const bool lab_is_blank = (L.stok[0].flags & TK_BLANK) != 0u;
    double p = L.surv[0].lp;
    uint32_t w_n = 0, w_pos = 0;
    int tt = t;
    for (;;) {
      for (int i = ctx.tid; i < N; i += ctx.nt) {
        const double nl = b.logit[i] + p;
        L.c_logit[i] = nl;
        L.p_score[i] = total_score(tab, nl, b.lm_hw[i], b.pscore[i], plen(b, i));
      }
      ctx.sync();
      const double thr = L.p_score[0] + prm.beam_prune_logp;
      for (int i = ctx.tid; i < N; i += ctx.nt) {
        const double sc = L.p_score[i];
        bool bad = !(sc >= thr);
        if (i + 1 < N) bad = bad || !(sc >= L.p_score[i + 1]);
        if (bad) L.scal[14] = 1u;
      }
      ctx.sync();
      if (L.scal[14]) break;
      for (int i = ctx.tid; i < N; i += ctx.nt) b.logit[i] = L.c_logit[i];
      ++tt;
      if (tt >= io.T) break;
      if (w_pos == w_n) {  // look ahead: up to 64 frames, one per thread
        if (ctx.tid == 0) L.scal[15] = 64u;
        ctx.sync();
        for (int k = ctx.tid; k < 64; k += ctx.nt) {
          const int f = tt + k;
          bool q = false;
          if (f < io.T) {
            q = io.surv_cnt[f] == 1u && io.surv_id[(size_t)f * prm.max_surv] == lab;
            L.p_logit[k] = io.surv_lp[(size_t)f * prm.max_surv];
          }
          if (!q) ctx.atomic_min(&L.scal[15], (uint32_t)k);
        }
        ctx.sync();
        w_n = L.scal[15];
        w_pos = 0;
        if (w_n == 0) break;
      }
      p = L.p_logit[w_pos];
      ++w_pos;
    }
    if (tt == t) return t;
    if (!lab_is_blank)
      for (int i = ctx.tid; i < N; i += ctx.nt) b.pend[i] = io.first_frame + tt;  // last held frame + 1
    ctx.sync();
    prefetch(tt);
    prefetch_tok();
    return tt;
  }

  CTC_HD int step(int t) {
    int frame = io.first_frame + t;
    if (run_ok && pf_cnt == 1u && N > 0 && !prm.no_label_runs) {
      const int t2 = label_run(t);
      if (t2 > t) return t2;
    }
    if (ctx.tid == 0) {
      L.scal[0] = 0;
      L.scal[4] = 0;
      L.smax[0] = asc_key(-INFINITY);
      L.smax[2] = 0;
    }
    tick<9>();
    const BeamSoA b = beams_at(cur);
    prefetch_node(b);  // in flight while the first wave works out the branch modes
    const uint32_t ns = pf_cnt;
    if (ns <= (uint32_t)ctx.wave_width() && ns <= (uint32_t)TOK_STAGE) {
      load_survivors_and_modes(ns);
      tick<0>();
    } else {
      load_survivors(t);
      ctx.sync();
      tick<0>();
      compute_modes(ns);
    }
    tick<1>();
    if (L.scal[4]) {
      for (int i = ctx.tid; i < N; i += ctx.nt)
        if (plen(b, i) > 0 && b.comp_node[i] == 0) make_completion(b, i);
      // The TextNodes written here are read by other threads from the next frame on. Their stores have to be
      // complete behind a barrier by then, but nothing in THIS frame reads them (the c_* copies are in LDS):
      // the full barrier that waits for them is the one that ends the scoring phase, a few microseconds
      // from now, when the wait is free -- here it would stall every wave for a store round trip.
      flush_nodes = true;
      tick<22>();
    }
    ctx.sync();
    prefetch(t + 1);  // lands while this frame's candidates are processed
    tick<2>();
    // Labels are taken in chunks of whole labels (<= cand candidates). Before a chunk that might not fit
    // the pool, the pool is compacted to its best beam_width entries; that also fixes the score a later
    // candidate has to beat (smax[2]), so the following chunks add little to the pool.
    // (N == 0 -- no beam left, ST_NO_BEAMS already set -- runs through with empty candidate sets)
    uint32_t per = (uint32_t)shape.cand / (uint32_t)(N > 0 ? N : 1);
    if (per == 0) per = 1;
    for (uint32_t s0 = 0; s0 < ns; s0 += per) {
      uint32_t s1 = s0 + per < ns ? s0 + per : ns;
      if (L.scal[0] + (s1 - s0) * (uint32_t)N > (uint32_t)shape.pool) prune_pool();
      process_chunk(s0, s1);
    }
    if (flush_nodes) {  // (no chunk ran: cannot happen with >= 1 survivor per frame, kept for safety)
      ctx.sync_mem();
      flush_nodes = false;
    }
    prefetch_tok();
    finish_frame(frame, false);
    run_ok = true;
    return t + 1;
  }

  // threshold prune, top-B, history prune, next beam table (decoder.py:545-554)
  CTC_HD void finish_frame(int frame, bool final_stage) {
    uint32_t pool_n = L.scal[0];
    double thr = sortable_to_max() + prm.beam_prune_logp;
    const bool hist = prm.prune_history && !final_stage;
    uint32_t n = sort_pool(pool_n, thr, hist);
    tick<7>();
    const bool big_set = n > 256u;  // ranked by the bucket selection: history keys still to be compared
    if (n > (uint32_t)prm.beam_width) n = (uint32_t)prm.beam_width;
    // nothing passed the threshold: only possible with non-finite scores (NaN rows) or a positive
    // beam_prune_logp; the reference then dies on max([]) (decoder.py:545) -- reported through the status
    if (n == 0 && ctx.tid == 0) {
      L.scal[7] = 0;
      L.scal[6] |= ST_NO_BEAMS;
    }
    const BeamSoA nb = beams_at(cur ^ 1);
    if (final_stage) {
      if (ctx.tid == 0) L.scal[9] = n;
      ctx.sync();
      return;
    }
    if (hist && big_set) {  // (small sets: already done by the ranking sweep)
      // first of each (history, partial, last_char) in sorted order wins (decoder.py:248-257).
      // The r2 < r pair space is tiled 16 x 16 over the threads so that no thread walks a whole row.
      const uint32_t ta = (uint32_t)ctx.tid >> 4, tb = (uint32_t)ctx.tid & 15u;
      const uint32_t step_a = ctx.nt >= 16 ? (uint32_t)ctx.nt >> 4 : 1u;
      const uint32_t step_b = ctx.nt >= 16 ? 16u : 1u;
      for (uint32_t r = (ctx.nt >= 16 ? ta : 0u); r < n; r += step_a) {
        const uint64_t hh = L.hk_h[r], ph = L.hk_p[r];
        const uint32_t cc = L.hk_c[r];
        uint32_t dup = 0;
        for (uint32_t r2 = (ctx.nt >= 16 ? tb : 0u); r2 < r; r2 += step_b)
          dup |= (L.hk_h[r2] == hh && L.hk_p[r2] == ph && L.hk_c[r2] == cc) ? 1u : 0u;
        if (dup) L.keep[r] = 0u;  // keep[] was preset to 1 by the ranking phase
      }
      ctx.sync();
    }
    tick<20>();
    const uint32_t W = (uint32_t)ctx.wave_width();
    const uint32_t lane = (uint32_t)ctx.tid & (W - 1u);
    for (uint32_t base = 0; base < n; base += (uint32_t)ctx.nt) {  // uniform trip count: ballots inside
      const uint32_t r = base + (uint32_t)ctx.tid;
      uint32_t dst = r;
      uint32_t kept = r < n ? 1u : 0u;
      if (hist) {
        // position = kept entries before r: whole 64-blocks by ballot + popcount, own block by lane mask
        const uint32_t wave0 = r - lane;
        dst = 0;
        for (uint32_t j0 = 0; j0 < wave0; j0 += W)
          dst += (uint32_t)ctx.popc64(ctx.ballot(j0 + lane < n && L.keep[j0 + lane] != 0u));
        kept = (r < n && L.keep[r] != 0u) ? 1u : 0u;
        const uint64_t mine = ctx.ballot(kept != 0u);
        dst += (uint32_t)ctx.popc64(mine & ((1ull << lane) - 1ull));
      }
      if (r == n - 1) L.scal[7] = dst + kept;  // size of the next beam table
      if (kept) build_beam(nb, (int)dst, L.sel[r], frame);
    }
    tick<21>();
    ctx.sync();
    N = (int)L.scal[7];
    cur ^= 1;
    tick<8>();
  }

  // ---- init / finalisation -------------------------------------------------------------------
  CTC_HD void init() {
    if (ctx.tid == 0) {
      for (int k = 0; k < 16; ++k) L.scal[k] = 0;
      L.scal[1] = node_span();  // text node 0 = empty text
      L.scal[2] = io.emit_start > 0 ? io.emit_start : 1u;  // emission node 0 = root (a resident stream goes on in its arena)
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
      if (MULTI && io.start_state) {  // model k's start state rides in node k
        for (uint32_t k = 1; k < tab.n_lms; ++k) {
          root.state = io.start_state[k];
          io.text_nodes[k] = root;
        }
      }
      EmitNode er;
      er.parent = 0;
      er.tok_branch = 0;
      er.wstart = -1;
      er.wend = -1;
      io.emit_nodes[0] = er;
      const BeamSoA b = beams_at(0);
      b.logit[0] = 0.0;
      b.lm_hw[0] = 0.0;
      b.pscore[0] = 0.0;
      b.c_lm_hw[0] = 0.0;
      b.text_h[0] = 0;
      b.part_h[0] = 0;
      b.hist_h[0] = root.hist_h;
      b.c_text_h[0] = 0;
      b.c_hist_h[0] = 0;
      b.text_node[0] = 0;
      b.comp_node[0] = 0;
      b.emit_node[0] = 0;
      b.word_id[0] = 0;
      b.meta1[0] = NO_CHAR;  // last_char None, empty partial
      b.meta2[0] = EMPTY_PARTIAL_M2;
      b.depth[0] = 0;
      b.pstart[0] = -1;
      b.pend[0] = -1;
    }
    clear_table();
    ctx.sync_mem();
    cur = 0;
    N = 1;
    if (io.imports && io.n_import > 0) import_beams();
  }

  // streaming: rebuild the beam table from the caller's beams (their order is the rank order). Beams built by the host
  // are rooted in fresh BR_IMPORT emission nodes; beams carried over on the device (resident streams) keep their chains.
  CTC_HD void import_beams() {
    const BeamSoA b = beams_at(0);
    const int n = io.n_import;
    const uint32_t emit_base = L.scal[2];
    const bool host_built = io.imports[0].resident == 0u;
    for (int i = ctx.tid; i < n; i += ctx.nt) {
      const ImportBeam& m = io.imports[i];
      const uint32_t span = node_span();
      const uint32_t node = (1u + (uint32_t)i) * span;  // node 0 (.. span-1) is the empty text
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
      if (MULTI) {
        TextNode* more = &tn;
        for (uint32_t k = 1; k < span; ++k)
          copy_state(&more[k].state, io.import_xstates[(size_t)i * (span - 1) + (k - 1)]);
      }
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
          ctx.atomic_or(&L.scal[6], ST_EMIT_OVERFLOW);
          enode = io.emit_cap - 1;
        }
        io.emit_nodes[enode] = en;
      }
      b.logit[i] = m.logit_score;
      b.lm_hw[i] = lmhw;
      b.pscore[i] = m.plen > 0 ? partial_score(tab, prm, m.m2 & PF_PARTIAL_MASK, (m.m2 & M2_HOT_ON) ? ((m.m2 >> 8) & 0xFFFFu) : 0u, m.plen) : 0.0;
      b.c_lm_hw[i] = 0.0;
      b.text_h[i] = m.text_h;
      b.part_h[i] = m.part_h;
      b.hist_h[i] = hh;
      b.c_text_h[i] = 0;
      b.c_hist_h[i] = 0;
      b.text_node[i] = node;
      b.comp_node[i] = 0;
      b.emit_node[i] = enode;
      b.word_id[i] = m.word_id;
      b.meta1[i] = (m.last_char & 0xFFFFu) | (m.plen << 16);
      b.meta2[i] = m.plen > 0 ? m.m2 : EMPTY_PARTIAL_M2;
      b.depth[i] = depth;
      b.pstart[i] = m.pstart;
      b.pend[i] = m.pend;
    }
    ctx.sync();  // (every thread has read the arena's first free node)
    if (ctx.tid == 0) {
      L.scal[1] = (1 + (uint32_t)n) * node_span();
      if (host_built) L.scal[2] = emit_base + (uint32_t)n;
    }
    ctx.sync_mem();
    N = n;
  }

  // _finalize_beams(force_next_word, is_end) + output records (decoder.py:558-602,653-667).
  // prm.fold: close the open word of every beam and merge equal texts; prm.eos: score end of sentence.
  CTC_HD void finalise() {
    const BeamSoA b = beams_at(cur);
    const bool fold = prm.fold != 0, eos = prm.eos != 0;
    if (ctx.tid == 0) {
      L.scal[0] = 0;
      L.smax[0] = asc_key(-INFINITY);
    }
    ctx.sync_mem();
    pre_valid = false;
    if (fold) {
      for (int i = ctx.tid; i < N; i += ctx.nt)
        if (plen(b, i) > 0 && b.comp_node[i] == 0) make_completion(b, i);
    }
    ctx.sync_mem();
    const int Q = N;  // N <= beam capacity <= candidate chunk
    if (fold) {
      // candidates: one per beam, key (text (+) partial, "", None)
      for (int q = ctx.tid; q < Q; q += ctx.nt) {
        L.ck_text[q] = plen(b, q) > 0 ? b.c_text_h[q] : b.text_h[q];
        L.ck_part[q] = 0;
        L.c_logit[q] = b.logit[q];
        L.rmin[q] = 0xFFFFFFFFu;
        L.rmax[q] = 0;
        L.rcnt[q] = 0;
      }
      ctx.sync();
      for (int q = ctx.tid; q < Q; q += ctx.nt) {
        uint32_t r = table_insert((uint32_t)q, 0x7FFFFFFFu);
        ctx.atomic_min(&L.rmin[r], (uint32_t)q);
        ctx.atomic_max(&L.rmax[r], (uint32_t)q);
        ctx.atomic_add(&L.rcnt[r], 1u);
        L.keep[q] = r;  // keep[] doubles as the representative map here (capacity bw >= N)
      }
      ctx.sync();
    }
    for (int q = ctx.tid; q < Q; q += ctx.nt) {
      double lg, score;
      uint32_t donor = (uint32_t)q;
      if (fold) {
        uint32_t r = L.keep[q];
        if (L.rmin[r] != (uint32_t)q) continue;
        uint32_t qmax = L.rmax[r];
        lg = L.c_logit[q];
        for (uint32_t q2 = (uint32_t)q + 1; q2 <= qmax; ++q2)
          if (L.keep[q2] == r) lg = lse2(lg, L.c_logit[q2]);
        // scored through the donor's (text, next_word) split (decoder.py:387-395)
        const int d = (int)qmax;
        donor = qmax;
        const uint32_t m2 = b.meta2[d];
        const uint32_t pl = plen(b, d);
        double lmhw;
        if (eos) {
          const TextNode& src = io.text_nodes[b.text_node[d]];
          uint32_t cnt = src.hw_cnt + ((pl > 0 && (m2 & M2_HOT_COMPLETE)) ? 1u : 0u);
          if (MULTI) {
            const TextNode* src_nodes = &src;
            const double w = multi_word_score(
                pl > 0 ? b.word_id[d] : 0u, true, [&](int k, LmState* st) { copy_state(st, src_nodes[k].state); },
                [&](int, const LmState&) {});
            lmhw = (src.raw_lm + w) + prm.hot_weight * (double)cnt;
          } else if (tab.has_lm) {
            LmState end;
            uint32_t wid = pl > 0 ? b.word_id[d] : 0u;
            uint32_t wfl = pl > 0 ? m2 : 0u;
            float base_s = lm_base_score(tab, src.state, wid, &end);
            double end_score = 0.0;
            if (prm.score_boundary) {
              LmState tmp;
              end_score = (double)lm_base_score(tab, end, tab.eos_id, &tmp);
            }
            double raw = src.raw_lm + lm_word_score(tab, prm, base_s, wfl, end_score, true);
            lmhw = raw + prm.hot_weight * (double)cnt;
          } else {
            lmhw = prm.hot_weight * (double)cnt;
          }
        } else {
          lmhw = pl > 0 ? b.c_lm_hw[d] : b.lm_hw[d];  // memo entry (text (+) word, False)
        }
        score = tab.has_lm ? lg + lmhw : lg + lmhw + 0.0;
      } else {
        lg = b.logit[q];
        score = total_score(tab, lg, b.lm_hw[q], b.pscore[q], plen(b, q));
      }
      ctx.atomic_max64(&L.smax[0], asc_key(score));
      pool_push(score, lg, (uint32_t)q, donor);
    }
    ctx.sync();
    finish_frame(0, true);
    uint32_t n = L.scal[9];
    if (io.carry_out && !eos) carry_beams(b, n, fold);
    uint32_t n_out = io.want_out ? n : 0u;
    if (prm.n_best > 0 && n_out > (uint32_t)prm.n_best) n_out = (uint32_t)prm.n_best;
    if (prm.texts_only != 0 && n_out > 0) {
      // decode_batch: only the best beam's text is wanted, and a separate launch assembles it (assemble_texts: every
      // utterance's chain walk at once instead of at the tail of this one's life) -- leave it where its chain ends
      if (ctx.tid == 0) {
        const uint32_t idx = L.sel[0];
        const int d = (int)L.p_don[idx];
        OutBeam& ob = io.out[0];
        ob.logit_score = L.p_logit[idx];
        ob.lm_score = L.p_score[idx];
        ob.raw_lm = 0.0;
        ob.tok_off = 0;
        ob.tok_cnt = 0;
        ob.state.len = -1;
        ob.last_char = NO_CHAR;
        ob.pstart = ob.pend = -1;
        ob.pad[0] = 0;
        ob.pad[1] = b.emit_node[d];
        *io.n_out = 1;
        *io.status = L.scal[6];
      }
      return;
    }
    // output records + back-trace of each returned beam's emission chain
    if (ctx.tid == 0) L.scal[8] = 0;
    ctx.sync();
    for (uint32_t r = ctx.tid; r < n_out; r += ctx.nt) {
      uint32_t idx = L.sel[r];
      int d = (int)L.p_don[idx];
      uint32_t len = b.depth[d] + ((fold && plen(b, d) > 0) ? 1u : 0u);
      L.keep[r] = ctx.atomic_add(&L.scal[8], len);  // offset inside this utterance's block
    }
    ctx.sync();
    if (ctx.tid == 0) {
      unsigned long long base = ctx.global_add(io.tok_pool_head, (unsigned long long)L.scal[8]);
      if (base + L.scal[8] > io.tok_pool_cap) {
        L.scal[6] |= ST_TOK_OVERFLOW;
        base = 0;
      }
      L.smax[1] = base;
    }
    ctx.sync();
    bool tok_ok = !(L.scal[6] & ST_TOK_OVERFLOW);
    for (uint32_t r = ctx.tid; r < n_out; r += ctx.nt) {
      uint32_t idx = L.sel[r];
      int d = (int)L.p_don[idx];
      OutBeam& ob = io.out[r];
      ob.logit_score = L.p_logit[idx];
      ob.lm_score = L.p_score[idx];
      const uint32_t pl = plen(b, d);
      const bool closes = fold && pl > 0;
      uint32_t len = b.depth[d] + (closes ? 1u : 0u);
      uint32_t off = (uint32_t)(L.smax[1] + L.keep[r]);
      ob.tok_off = off;
      ob.tok_cnt = tok_ok ? len : 0;
      ob.pad[0] = 0;
      ob.pad[1] = 0;
      ob.last_char = fold ? NO_CHAR : last_char(b, d);
      ob.pstart = fold ? -1 : b.pstart[d];
      ob.pend = fold ? -1 : b.pend[d];
      // the text's memo entry: raw LM sum and the state after its last word
      const TextNode& node = io.text_nodes[closes ? b.comp_node[d] : b.text_node[d]];
      ob.raw_lm = node.raw_lm;
      if (!tab.has_lm) {
        ob.state.len = -1;
CTC_UNROLL
        for (int k = 0; k < MAX_CTX; ++k) {
          ob.state.words[k] = 0;
          ob.state.backoff[k] = 0.f;
        }
      } else if (MULTI) {
        // one state per model: LM 0 in the record, the others in the side array
        LmState* xs = io.out_xstates + (size_t)r * (tab.n_lms - 1);
        if (eos) {
          const TextNode* src_nodes = &io.text_nodes[b.text_node[d]];
          multi_word_score(
              pl > 0 ? b.word_id[d] : 0u, false, [&](int k, LmState* st) { copy_state(st, src_nodes[k].state); },
              [&](int k, const LmState& st) { copy_state(k == 0 ? &ob.state : &xs[k > 0 ? k - 1 : 0], st); });
        } else {
          const TextNode* nodes = &node;
          copy_state(&ob.state, nodes[0].state);
          for (uint32_t k = 1; k < tab.n_lms; ++k) copy_state(&xs[k - 1], nodes[k].state);
        }
      } else if (eos) {
        // last_lm_state: state after the last word, before </s> (language_model.py:357); an empty
        // last word is still scored as a word (decoder.py:387-395)
        const TextNode& src = io.text_nodes[b.text_node[d]];
        lm_base_score(tab, src.state, pl > 0 ? b.word_id[d] : 0u, &ob.state);
      } else {
        ob.state.len = node.state.len;
CTC_UNROLL
        for (int k = 0; k < MAX_CTX; ++k) {
          ob.state.words[k] = node.state.words[k];
          ob.state.backoff[k] = node.state.backoff[k];
        }
      }
      if (tok_ok) {
        uint32_t pos = off + len;
        if (closes) {
          EmitNode fin;
          fin.parent = 0;
          fin.tok_branch = BR_FINAL << 16;
          fin.wstart = b.pstart[d];
          fin.wend = b.pend[d];
          io.tok_pool[--pos] = fin;
        }
        uint32_t e = b.emit_node[d];
        while (e != 0 && pos > off) {
          EmitNode en = io.emit_nodes[e];
          io.tok_pool[--pos] = en;
          e = en.parent;
        }
      }
    }
    ctx.sync();
    if (ctx.tid == 0) {
      *io.n_out = n_out;
      *io.status = L.scal[6];
      if (io.sstate) {
        io.sstate->n_carry = n;
        io.sstate->emit_next = L.scal[2];
        io.sstate->status = L.scal[6];
        io.sstate->pad = 0;
      }
    }
  }

  // Device-resident streams: the ranked beams of this chunk, written where the next chunk's import_beams() reads them
  // (what the reference's caller carries between partial_decode_beams calls, decoder.py:681-728). A beam whose open
  // word the finalisation closed (force_next_word) gets a BR_FINAL emission node for that word.
  CTC_HD void carry_beams(const BeamSoA& b, uint32_t n, bool fold) {
    for (uint32_t r = ctx.tid; r < n; r += ctx.nt) {
      const uint32_t idx = L.sel[r];
      const int d = (int)L.p_don[idx];
      const uint32_t pl = plen(b, d);
      const bool closes = fold && pl > 0;
      uint32_t enode = b.emit_node[d], depth = b.depth[d];
      if (closes) {
        uint32_t e = ctx.atomic_add(&L.scal[2], 1u);
        if (e >= io.emit_cap) {
          ctx.atomic_or(&L.scal[6], ST_EMIT_OVERFLOW);
          e = io.emit_cap - 1;
        }
        EmitNode fin;
        fin.parent = enode;
        fin.tok_branch = BR_FINAL << 16;
        fin.wstart = b.pstart[d];
        fin.wend = b.pend[d];
        io.emit_nodes[e] = fin;
        enode = e;
        depth += 1;
      }
      const uint32_t nidx = closes ? b.comp_node[d] : b.text_node[d];
      const TextNode& node = io.text_nodes[nidx];
      ImportBeam& m = io.carry_out[r];
      m.logit_score = L.p_logit[idx];
      m.raw_lm = node.raw_lm;
      m.text_h = node.text_h;
      m.part_h = fold ? 0ull : b.part_h[d];
CTC_UNROLL
      for (int k = 0; k < MAX_CTX; ++k) m.ring[k] = node.ring[k];
      m.ring_cnt = node.ring_cnt;
      m.hw_cnt = node.hw_cnt;
      m.plen = fold ? 0u : pl;
      m.last_char = fold ? NO_CHAR : last_char(b, d);
      m.m2 = fold ? EMPTY_PARTIAL_M2 : b.meta2[d];
      m.word_id = fold ? 0u : b.word_id[d];
      m.pstart = fold ? -1 : b.pstart[d];
      m.pend = fold ? -1 : b.pend[d];
      copy_state(&m.state, node.state);
      m.enode = enode;
      m.depth = depth;
      m.resident = 1u;
      if (MULTI && io.carry_xstates) {  // the states of model 1.. ride in the nodes behind the text's own
        const TextNode* nodes = &node;
        for (uint32_t k = 1; k < tab.n_lms; ++k) copy_state(&io.carry_xstates[(size_t)r * (tab.n_lms - 1) + (k - 1)], nodes[k].state);
      }
    }
    ctx.sync();
  }

  CTC_HD void run() {
    init();
    if (io.prof && ctx.tid == 0) t_last = ctx.clock();
    prefetch(0);
    prefetch_tok();
    for (int t = 0; t < io.T;) t = step(t);
    tick<9>();
    finalise();
    tick<10>();
    if (io.prof && ctx.tid == 0) {
CTC_UNROLL
      for (int k = 0; k < N_PROF; ++k) io.prof[k] = t_acc[k];
    }
  }
};

}  // namespace ctc
