// common.h -- hashing and table layouts shared by the host builder, the HIP kernels and the CPU
// simulator build (tests/sim).  No torch, no STL in anything the device sees.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) && !defined(CTC_SIM)
#define CTC_HD __host__ __device__ __forceinline__
#else
#define CTC_HD inline
#endif

// loop unrolling request for the device compiler (a plain host compiler does not know the pragma)
#if defined(__HIPCC__) && !defined(CTC_SIM)
#define CTC_UNROLL _Pragma("unroll")
#else
#define CTC_UNROLL
#endif

namespace ctc {

// ---------------------------------------------------------------------------------------------
// String identity.  The reference keys every merge / memo on Python strings
// (decoder.py:215-216, 250-254, 387, 399).  On device a string is a PAIR of polynomial hashes over its UTF-8 bytes
// (+1 so that no byte is zero) in the Mersenne field p = 2^31-1, with two independent bases, packed into one 64-bit word
// (low half: base 1, high half: base 2):
//     H("") = 0,  H(s.c) = H(s)*BASE + (c+1),  H(s.t) = H(s)*BASE^|t| + H(t)   (each half mod p)
// Two distinct strings of <= L bytes collide with probability <= (L/2^31)^2 (random bases): 2^-52 for 30-byte words.
// (Rounds 1-4 used ONE polynomial mod 2^61-1: the same strength, but its 61 x 61-bit modular product is seven multiply-class
// instructions and ~45 vector instructions per appended label on this hardware; two 31 x 31-bit ones are two and ~18.)
// ---------------------------------------------------------------------------------------------
constexpr uint32_t P31 = 0x7FFFFFFFu;
constexpr uint32_t STR_BASE_1 = 0x3A4E6F71u & P31, STR_BASE_2 = 0x1D2F5C8Bu;  // byte polynomial bases (< p)
constexpr uint64_t STR_BASE = ((uint64_t)STR_BASE_2 << 32) | STR_BASE_1;      // ... packed like a hash

// x < 2^32 - 1 -> its canonical residue (x - p wraps to a huge value when x < p: the minimum is x)
CTC_HD uint32_t mod31(uint32_t x) {
  const uint32_t y = x - P31;
  return y < x ? y : x;
}
CTC_HD uint32_t mulmod31(uint32_t a, uint32_t b) {  // a, b < p
  const uint64_t t = (uint64_t)a * b;               // < 2^62
  return mod31(((uint32_t)t & P31) + (uint32_t)(t >> 31));
}
CTC_HD uint32_t addmod31(uint32_t a, uint32_t b) { return mod31(a + b); }  // a, b < p

// H(s . t) from H(s), BASE^|t|, H(t)   (all three packed pairs)
CTC_HD uint64_t str_concat(uint64_t hs, uint64_t pow_t, uint64_t ht) {
  const uint32_t lo = addmod31(mulmod31((uint32_t)hs, (uint32_t)pow_t), (uint32_t)ht);
  const uint32_t hi = addmod31(mulmod31((uint32_t)(hs >> 32), (uint32_t)(pow_t >> 32)), (uint32_t)(ht >> 32));
  return ((uint64_t)hi << 32) | lo;
}
// H(s . c) for one byte c
CTC_HD uint64_t str_push_byte(uint64_t hs, unsigned char c) {
  const uint64_t one = (uint64_t)c + 1u;
  return str_concat(hs, STR_BASE, (one << 32) | one);
}

CTC_HD uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// One step of a 64-bit state (two 32-bit halves) absorbing a 32-bit / 64-bit value: two 32-bit multiply-xorshift rounds, the
// second half keyed by the first. For a fixed input the step is a bijection of the state (each half is a bijection of itself
// given the other), so two different histories can only meet by accident (~2^-64). Written in 32-bit halves because that is
// what the hardware multiplies at full width in one instruction: a 64 x 64 -> 64 multiply is three multiply-class
// instructions and a 61-bit modular one seven; a step here is two (round 4 spent 74 of a word completion's ~310 vector
// instructions hashing n-gram keys with 64-bit finalisers and 50 on the text's modular polynomial).
CTC_HD uint64_t mix_step(uint64_t state, uint32_t x_lo, uint32_t x_hi) {
  uint32_t a = (uint32_t)state, b = (uint32_t)(state >> 32);
  a = (a ^ x_lo) * 0x9E3779B1u;
  a ^= a >> 15;
  b = ((b + x_hi) ^ a) * 0x85EBCA6Bu;
  b ^= b >> 13;
  return ((uint64_t)b << 32) | a;
}

// text' = text (+) word : a position-sensitive chain over word hashes (the words are polynomial string hashes, 2 x 31 bits;
// appending the same word to different texts keeps them different, appending different words to one text gives different
// texts unless the step collides). Only ever compared for equality, and computed by this one function on the host
// (imported streaming beams, api.cpp) and on the device.
CTC_HD uint64_t text_push(uint64_t text_h, uint64_t word_h) {
  const uint64_t s1 = mix_step(text_h, (uint32_t)word_h, (uint32_t)(word_h >> 32));
  // (a second absorption of the word, halves swapped: every bit of the word reaches both halves of the state through a multiply)
  return mix_step(s1, (uint32_t)(word_h >> 32) + 0x7F4A7C15u, (uint32_t)word_h);
}

CTC_HD uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

// merge key (text, partial, last_char) -> table slot hash (equality is checked on all three)
CTC_HD uint32_t key_slot_hash(uint64_t text_h, uint64_t part_h, uint32_t ch) {
  return (uint32_t)(mix64(text_h * 0x9E3779B97F4A7C15ull + part_h * 0xC2B2AE3D27D4EB4Full + ch) >> 32);
}

// n-gram key: a chain over the word ids NEWEST FIRST (the scored word, then its context going back), so the keys of all
// orders of one query share their prefix: key_n = end(push(...push(first(w_n), w_{n-1})..., w_1), n) costs one step per order
// instead of one per word per order. Round 6: the chain IS kenlm's own -- lm/search_hashed.hh: a unigram's node is its word
// index, every further (older) context word is folded in by CombineWordHash, (node * 8978948897894561157) ^ ((1 + word) *
// 17894857484156487943) -- over kenlm's own word indices (<unk> = 0, the others in the order of the ARPA unigram section):
// the middle / longest hash tables of a kenlm PROBING binary can then be adopted entry by entry, key and all, without the
// n-grams' words (which such a file does not hold: host_tables.cpp, HostLM::load_kenlm_binary). The order goes into the top
// byte (kenlm keeps one table per order, here all orders share one); never 0 (0 marks an empty slot); the slot is the key's
// low bits (tools/hash_quality.py on the bench model: 849 152 n-grams + 18 M synthetic tuples without a collision, 1.126
// probes per hit at load <= 1/4 -- the same as rounds 4 and 5's chains). Two 64-bit multiplies per step (~13 vector
// instructions; round 5's mix_step: ~8) on a path that runs once per completed word.
CTC_HD uint64_t ngram_key_first(uint32_t wid) { return (uint64_t)wid; }
CTC_HD uint64_t ngram_key_push(uint64_t k, uint32_t id) {
  return (k * 8978948897894561157ull) ^ ((uint64_t)(1u + id) * 17894857484156487943ull);
}
CTC_HD uint64_t ngram_key_end(uint64_t k, uint32_t n) {
  k ^= (uint64_t)n << 56;
  return k == 0 ? 1 : k;
}

// slot hash of the prefix / hot-word tables (keys are polynomial string hashes, 2 x 31 bits): fold to 32 bits, one
// multiply, one xor-shift so that the low bits used for the slot depend on all of the key
CTC_HD uint64_t table_slot(uint64_t key) {
  uint32_t x = (uint32_t)key ^ (uint32_t)(key >> 32);
  x *= 0x9E3779B1u;
  x ^= x >> 15;
  return (uint64_t)x;
}

// ---------------------------------------------------------------------------------------------
// Device tables (all open addressing, linear probing, power-of-two sizes, empty key = 0)
// ---------------------------------------------------------------------------------------------
struct NgramEntry {  // 16 B: one dwordx4 probe
  uint64_t key;
  float prob;     // log10
  float backoff;  // log10, 0 when ARPA omits it
};

// every code-point-boundary prefix of every LM vocabulary word (the unigram char-trie of
// language_model.py:263 and the kenlm vocabulary of :95,352 in one table)
enum : uint32_t {
  PF_UNI_PREFIX = 1u,  // prefix of a word in unigram_set   (score_partial_token, :331)
  PF_LM_WORD = 2u,     // is itself an LM vocabulary word   (word in kenlm_model, :352)
  PF_UNI_WORD = 4u,    // is itself in unigram_set          (:351)
};
struct PrefixEntry {  // 16 B
  uint64_t key;       // H(prefix bytes); never 0 for a stored prefix
  uint32_t word_id;   // LM vocabulary index when PF_LM_WORD, else 0 (<unk>); several LMs: index into the
                      // union word list (see LmExtra::winfo)
  uint32_t flags;     // PF_* of LM 0; several LMs: PF_UNI_PREFIX of LM k >= 1 in bit PF_X_SHIFT + k
};
// MultiLanguageModel (language_model.py:455-502): up to MAX_LMS n-gram models scored side by side. The
// per-frame state of a beam only needs each model's "is a unigram-trie prefix" bit; word identity and
// the vocabulary flags of each model are looked up by union word index when a word is completed.
constexpr int MAX_LMS = 4;
constexpr uint32_t PF_X_SHIFT = 23u;                   // LM k's prefix bit = 1 << (PF_X_SHIFT + k), k = 1..3
constexpr uint32_t PF_X_MASK = 0x07000000u;            // bits 24..26
constexpr uint32_t PF_PARTIAL_MASK = PF_X_MASK | 7u;   // what a beam keeps of a table entry's flags
constexpr uint32_t WI_LM_WORD = 0x80000000u, WI_UNI_WORD = 0x40000000u, WI_ID_MASK = 0x3FFFFFFFu;

struct HotEntry {  // 16 B; every prefix of every hot-word unigram (language_model.py:133-150)
  uint64_t key;
  uint32_t min_len;   // code points of the shortest hot word with this prefix
  uint32_t complete;  // 1 when the prefix is itself a hot word (language_model.py:137-139)
};

struct UnigramEntry {  // indexed by word id
  float prob;
  float backoff;
};

// per-label constants (normalised alphabet)
enum : uint32_t {
  TK_BLANK = 1u,  // ""                       decoder.py:452
  TK_SPACE = 2u,  // " " in a char vocabulary decoder.py:500
  TK_LEAD = 4u,   // starts with U+2581 (BPE) decoder.py:474,478
  TK_TRAIL = 8u,  // ends with U+2581         decoder.py:480-482
};
struct TokInfo {  // 64 B; the first three 16-byte chunks are, as they stand, what the wave kernel stages per label in LDS
  uint64_t h_raw, pow_raw;      // label as appended by branch D (decoder.py:528)
  uint64_t h_clean;             // label without boundary marks, as started by branch B (:477-481)
  uint32_t len_raw, len_clean;  // code points
  uint32_t flags;
  // prefix-table view of the CLEAN label taken as the start of a new word (static per LM)
  uint32_t start_flags;  // PF_* bits, plus PF_ON_TABLE
  uint32_t start_word_id;
  uint32_t pad0;
  uint64_t pow_clean;
  uint32_t pad[2];
};
static_assert(sizeof(TokInfo) == 64, "TokInfo is read in 16-byte chunks");
constexpr uint32_t PF_ON_TABLE = 8u;  // beam flag: the partial is a stored prefix

// the UTF-8 bytes of a label, for the kernels that assemble decoded texts themselves (decode_batch)
struct TokText {  // 16 B
  uint32_t raw_off, clean_off;  // into DeviceTables::tok_bytes: the label as appended / without boundary marks
  uint16_t raw_len, clean_len;
  uint32_t pad;
};

struct TokHot {  // per call (hot words change per call): hot-word view of the CLEAN label
  uint32_t min_len;  // 0 = not a hot-word prefix
  uint32_t complete;
};

constexpr int MAX_CTX = 5;  // CTCDEC_MAX_LM_ORDER - 1

struct LmState {
  int32_t len;
  uint32_t words[MAX_CTX];  // newest first
  float backoff[MAX_CTX];   // backoff[k]: back-off weight of the newest k+1 words
};

// one completed-words prefix ("text" of the reference Beam): everything that is a pure function
// of the word sequence (the reference memoises it per text: decoder.py:387-396)
struct TextNode {  // 128 B
  uint64_t text_h;
  double raw_lm;      // sum of LanguageModel.score over the words   (decoder.py:393)
  double lm_hw;       // raw_lm + hotword_weight * hw_cnt            (decoder.py:394)
  uint64_t hist_h;    // hash of the last n_hist words               (decoder.py:250-251)
  uint32_t hw_cnt;    // words that are hot words                    (language_model.py:139)
  uint32_t ring_cnt;  // words held in ring (<= n_hist)
  LmState state;      // 44 B
  uint32_t pad0;
  uint64_t ring[MAX_CTX];  // last word hashes, newest first
};

// one non-blank, non-repeat emission on a beam's path; host replays the chain into words+frames
enum : uint32_t { BR_BOUNDARY = 1u, BR_SPACE = 2u, BR_APPEND = 3u, BR_FINAL = 4u, BR_IMPORT = 5u };
struct EmitNode {  // 16 B
  uint32_t parent;
  uint32_t tok_branch;  // token | branch << 16
  int32_t wstart, wend; // frames of the word closed by this emission (BOUNDARY/SPACE/FINAL)
};

// per-beam result record written by the beam kernel
struct OutBeam {  // 104 B
  double logit_score;
  double lm_score;
  double raw_lm;      // LM score sum of the beam's text (memo value a streaming caller carries on)
  uint32_t tok_off;   // offset of this beam's emission list in the token pool
  uint32_t tok_cnt;
  LmState state;      // last_lm_state
  uint32_t last_char; // label of the last frame (NO_CHAR=0xFFFF: None)
  int32_t pstart, pend;  // partial_frames of the still open word
  uint32_t pad[2];
};

// a live beam handed back in by a streaming caller (partial_decode_beams, decoder.py:681-728):
// the host resolves strings to hashes / table views, the kernel rebuilds its LDS row from this
struct ImportBeam {  // 160 B
  double logit_score;
  double raw_lm;
  uint64_t text_h, part_h;
  uint64_t ring[MAX_CTX];
  uint32_t ring_cnt, hw_cnt, plen, last_char, m2, word_id;
  int32_t pstart, pend;
  LmState state;
  // device-resident streams (ctcdec_stream_*): a beam carried over by the previous chunk's kernel (resident = 1) keeps
  // its place in the stream's emission arena; a beam built by the host (0) is rooted in a fresh BR_IMPORT node
  uint32_t enode;
  uint32_t depth;  // emission nodes on its chain
  uint32_t resident;
};
// what a device-resident stream keeps between chunks besides its carried beams and its emission arena
struct StreamState {  // 16 B
  uint32_t n_carry;    // beams the last chunk handed on (rank order)
  uint32_t emit_next;  // first free node of the emission arena
  uint32_t status;
  uint32_t pad;
};

// one additional language model of a MultiLanguageModel (LM 0 lives in DeviceTables / DecodeParams)
struct LmExtra {
  const UnigramEntry* unigrams;
  const NgramEntry* ngrams;
  uint64_t ngram_mask;
  const uint32_t* winfo;  // union word index -> local word id | WI_* flags
  uint32_t lm_order, has_trie, uniset_nonempty, eos_id;
  double alpha, beta, unk;
  int32_t score_boundary;
  int32_t pad;
};

struct DeviceTables {
  const TokInfo* tok;
  const TokHot* tok_hot;
  const TokText* tok_text;    // per label: where its bytes are
  const uint8_t* tok_bytes;
  uint32_t max_label_bytes;   // longest label, in bytes
  uint32_t pad_text;
  const UnigramEntry* unigrams;
  const NgramEntry* ngrams;
  uint64_t ngram_mask;  // table size - 1 (0: no table)
  const PrefixEntry* prefixes;
  uint64_t prefix_mask;
  const HotEntry* hot;
  uint64_t hot_mask;
  uint32_t n_labels;
  uint32_t is_bpe;
  uint32_t has_lm;
  uint32_t lm_order;
  uint32_t has_trie;        // unigrams is not None       (language_model.py:328)
  uint32_t uniset_nonempty; // len(unigram_set) > 0       (language_model.py:350)
  uint32_t eos_id;          // vocabulary index of "</s>" (0 if absent)
  uint32_t n_hist;          // max(1, order-1)            (decoder.py:244)
  uint32_t n_lms;           // 0/1: the single model above; 2..MAX_LMS: MultiLanguageModel
  uint32_t pad_lms;
  const uint32_t* winfo0;   // LM 0's view of the union word list (several LMs only)
  LmExtra x[MAX_LMS - 1];   // LM 1..n_lms-1
};

struct DecodeParams {
  int32_t beam_width;
  int32_t prune_history;
  int32_t n_best;
  int32_t first_frame;
  double beam_prune_logp;
  double token_min_logp;
  double hot_weight;
  double alpha, beta, unk, log_base_change;
  int32_t score_boundary;
  int32_t max_surv;  // stride of the survivor arrays
  int32_t fold;      // finalisation closes the open word (force_next_word or is_end, decoder.py:570)
  int32_t eos;       // finalisation scores end of sentence (is_end, decoder.py:597)
  int32_t no_label_runs;  // diagnostics (CTCDEC_NO_LABEL_RUNS=1): every frame takes the full path
  int32_t texts_only;     // decode_batch: the best beam's text is assembled on the device, nothing else is returned
};

// ---------------------------------------------------------------------------------------------
// table probes (host + device)
// ---------------------------------------------------------------------------------------------
CTC_HD bool prefix_lookup(const PrefixEntry* tab, uint64_t mask, uint64_t key, uint32_t* word_id,
                          uint32_t* flags) {
  if (!tab || key == 0) return false;
  uint64_t s = table_slot(key) & mask;
  for (;;) {
    PrefixEntry e = tab[s];
    if (e.key == key) {
      *word_id = e.word_id;
      *flags = e.flags;
      return true;
    }
    if (e.key == 0) return false;
    s = (s + 1) & mask;
  }
}

CTC_HD bool hot_lookup(const HotEntry* tab, uint64_t mask, uint64_t key, uint32_t* min_len,
                       uint32_t* complete) {
  if (!tab || key == 0) return false;
  uint64_t s = table_slot(key) & mask;
  for (;;) {
    HotEntry e = tab[s];
    if (e.key == key) {
      *min_len = e.min_len;
      *complete = e.complete;
      return true;
    }
    if (e.key == 0) return false;
    s = (s + 1) & mask;
  }
}

CTC_HD bool ngram_lookup(const NgramEntry* tab, uint64_t mask, uint64_t key, float* prob, float* backoff) {
  uint64_t s = key & mask;
  for (;;) {
    NgramEntry e = tab[s];
    if (e.key == key) {
      *prob = e.prob;
      *backoff = e.backoff;
      return true;
    }
    if (e.key == 0) return false;
    s = (s + 1) & mask;
  }
}

// kenlm GenericModel::FullScore restated on the flat hashed trie (see oracle/arpa_lm.py for the
// CPU restatement and DESIGN.md for the state convention).  Returns log10 p as fp32.
// finish the probe of one order: first entry `e` was loaded from slot `s`; walk on collisions (rare)
template <class Tab>
CTC_HD bool lm_resolve(const Tab& t, uint64_t key, uint64_t s, NgramEntry e, float* prob, float* bo) {
  while (e.key != key && e.key != 0) {
    s = (s + 1) & t.ngram_mask;
    e = t.ngrams[s];
  }
  if (e.key != key) return false;
  *prob = e.prob;
  *bo = e.backoff;
  return true;
}

// Tab: DeviceTables (LM 0) or LmExtra (a further model): unigrams, ngrams, ngram_mask, lm_order
// One word scored in two halves so that a caller can put other work between the probes and their use:
// lm_probe_issue computes the n-gram keys and loads the first slot of every order (independent loads overlap
// their latency; kenlm walks the orders one after the other, the longest match is the same), lm_probe_finish
// resolves them. Scalars only: no run-time indexed temporaries (they would live in scratch memory).
struct LmProbe {
  UnigramEntry u;
  int max_n;
  uint64_t k2, k3, k4, k5, k6, s2, s3, s4, s5, s6;
  NgramEntry e2, e3, e4, e5, e6;
};

// MAXORD: the highest n-gram order compiled in (a kernel instantiated for models of order <= 4 carries no registers for
// the keys and entries of orders 5 and 6)
template <int MAXORD = MAX_CTX + 1, class Tab>
CTC_HD void lm_probe_issue(const Tab& t, const LmState& in, uint32_t wid, LmProbe& p) {
  p.u = t.unigrams[wid];
  const int in_len = in.len;
  const int max_n = !t.ngrams ? 1 : ((int)t.lm_order < in_len + 1 ? (int)t.lm_order : in_len + 1);
  p.max_n = max_n;
  const NgramEntry none = {0, 0.f, 0.f};
  p.k2 = p.k3 = p.k4 = p.k5 = p.k6 = 0;
  p.s2 = p.s3 = p.s4 = p.s5 = p.s6 = 0;
  p.e2 = p.e3 = p.e4 = p.e5 = p.e6 = none;
  // one chain, newest word first: the key of order n extends the key of order n-1 by one word
  uint64_t c = ngram_key_first(wid);
  if (max_n >= 2) { c = ngram_key_push(c, in.words[0]); p.k2 = ngram_key_end(c, 2); p.s2 = p.k2 & t.ngram_mask; p.e2 = t.ngrams[p.s2]; }
  if (MAXORD >= 3 && max_n >= 3) { c = ngram_key_push(c, in.words[1]); p.k3 = ngram_key_end(c, 3); p.s3 = p.k3 & t.ngram_mask; p.e3 = t.ngrams[p.s3]; }
  if (MAXORD >= 4 && max_n >= 4) { c = ngram_key_push(c, in.words[2]); p.k4 = ngram_key_end(c, 4); p.s4 = p.k4 & t.ngram_mask; p.e4 = t.ngrams[p.s4]; }
  if (MAXORD >= 5 && max_n >= 5) { c = ngram_key_push(c, in.words[3]); p.k5 = ngram_key_end(c, 5); p.s5 = p.k5 & t.ngram_mask; p.e5 = t.ngrams[p.s5]; }
  if (MAXORD >= 6 && max_n >= 6) { c = ngram_key_push(c, in.words[4]); p.k6 = ngram_key_end(c, 6); p.s6 = p.k6 & t.ngram_mask; p.e6 = t.ngrams[p.s6]; }
}

template <int MAXORD = MAX_CTX + 1, class Tab>
CTC_HD float lm_probe_finish(const Tab& t, const LmState& in, uint32_t wid, const LmProbe& p, LmState* out) {
  const int in_len = in.len;
  const int max_n = p.max_n;
  float prob = p.u.prob;
  float b0 = p.u.backoff, b1 = 0.f, b2 = 0.f, b3 = 0.f, b4 = 0.f, b5 = 0.f;
  int matched = 1;
  if (max_n >= 2 && lm_resolve(t, p.k2, p.s2, p.e2, &prob, &b1)) {
    matched = 2;
    if (MAXORD >= 3 && max_n >= 3 && lm_resolve(t, p.k3, p.s3, p.e3, &prob, &b2)) {
      matched = 3;
      if (MAXORD >= 4 && max_n >= 4 && lm_resolve(t, p.k4, p.s4, p.e4, &prob, &b3)) {
        matched = 4;
        if (MAXORD >= 5 && max_n >= 5 && lm_resolve(t, p.k5, p.s5, p.e5, &prob, &b4)) {
          matched = 5;
          if (MAXORD >= 6 && max_n >= 6 && lm_resolve(t, p.k6, p.s6, p.e6, &prob, &b5)) matched = 6;
        }
      }
    }
  }
  (void)b5;  // the highest order carries no back-off
  // back-offs of the skipped contexts, fp32, shortest context first (kenlm FullScore)
  const float i0 = in.backoff[0], i1 = in.backoff[1], i2 = in.backoff[2], i3 = in.backoff[3], i4 = in.backoff[4];
  if (0 >= matched - 1 && 0 < in_len) prob = prob + i0;
  if (1 >= matched - 1 && 1 < in_len) prob = prob + i1;
  if (2 >= matched - 1 && 2 < in_len) prob = prob + i2;
  if (3 >= matched - 1 && 3 < in_len) prob = prob + i3;
  if (4 >= matched - 1 && 4 < in_len) prob = prob + i4;
  const int keep = matched < (int)t.lm_order - 1 ? matched : (int)t.lm_order - 1;
  // `in` and `out` never alias (out is a fresh node or a local)
  const uint32_t w0 = in.words[0], w1 = in.words[1], w2 = in.words[2], w3 = in.words[3];
  out->len = keep;
  out->words[0] = keep > 0 ? wid : 0u;
  out->words[1] = keep > 1 ? w0 : 0u;
  out->words[2] = keep > 2 ? w1 : 0u;
  out->words[3] = keep > 3 ? w2 : 0u;
  out->words[4] = keep > 4 ? w3 : 0u;
  out->backoff[0] = keep > 0 ? b0 : 0.f;
  out->backoff[1] = keep > 1 ? b1 : 0.f;
  out->backoff[2] = keep > 2 ? b2 : 0.f;
  out->backoff[3] = keep > 3 ? b3 : 0.f;
  out->backoff[4] = keep > 4 ? b4 : 0.f;
  return prob;
}

template <int MAXORD = MAX_CTX + 1, class Tab>
CTC_HD float lm_base_score(const Tab& t, const LmState& in, uint32_t wid, LmState* out) {
  LmProbe p;
  lm_probe_issue<MAXORD>(t, in, wid, p);
  return lm_probe_finish<MAXORD>(t, in, wid, p, out);
}

}  // namespace ctc
