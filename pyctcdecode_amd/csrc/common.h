// common.h -- hashing and table layouts shared by the host builder, the HIP kernels and the CPU
// simulator build (tests/sim).  No torch, no STL in anything the device sees.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) && !defined(CTC_SIM)
#define CTC_HD __host__ __device__ __forceinline__
#else
#define CTC_HD inline
#endif

namespace ctc {

// ---------------------------------------------------------------------------------------------
// String identity.  The reference keys every merge / memo on Python strings
// (decoder.py:215-216, 250-254, 387, 399).  On device a string is its polynomial hash in the
// Mersenne field p = 2^61-1 over UTF-8 bytes (+1 so that no byte is zero):
//     H("") = 0,  H(s.c) = H(s)*BASE + (c+1),  H(s.t) = H(s)*BASE^|t| + H(t)   (all mod p)
// Two distinct strings of <= L bytes collide with probability <= L/2^61 (random BASE).
// ---------------------------------------------------------------------------------------------
constexpr uint64_t M61 = (1ull << 61) - 1;
constexpr uint64_t STR_BASE = 0x1D2F5C8B3A4E6F71ull & M61;   // byte polynomial base
constexpr uint64_t TEXT_BASE = 0x0B7E151628AED2A7ull & M61;  // word-sequence polynomial base

CTC_HD uint64_t mod61(uint64_t x) {
  x = (x & M61) + (x >> 61);
  return x >= M61 ? x - M61 : x;
}

CTC_HD uint64_t mulmod61(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t hi = __umul64hi(a, b);
  uint64_t lo = a * b;
#else
  unsigned __int128 pr = (unsigned __int128)a * b;
  uint64_t hi = (uint64_t)(pr >> 64);
  uint64_t lo = (uint64_t)pr;
#endif
  // a,b < 2^61 -> product < 2^122: hi < 2^58
  uint64_t r = (lo & M61) + ((lo >> 61) | (hi << 3));
  return mod61(r);
}

CTC_HD uint64_t addmod61(uint64_t a, uint64_t b) { return mod61(a + b); }

// H(s . t) from H(s), BASE^|t|, H(t)
CTC_HD uint64_t str_concat(uint64_t hs, uint64_t pow_t, uint64_t ht) {
  return addmod61(mulmod61(hs, pow_t), ht);
}

// text' = text (+) word : position-sensitive polynomial over word hashes (+1: words are non-empty)
CTC_HD uint64_t text_push(uint64_t text_h, uint64_t word_h) {
  return addmod61(mulmod61(text_h, TEXT_BASE), addmod61(word_h, 1));
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

// n-gram key: word ids oldest..newest (n >= 2).  Never 0 (0 marks an empty slot).
CTC_HD uint64_t ngram_key_begin(uint32_t n) { return 0x243F6A8885A308D3ull ^ n; }
CTC_HD uint64_t ngram_key_push(uint64_t k, uint32_t id) { return mix64(k ^ (uint64_t)id) + 0x13198A2E03707344ull; }
CTC_HD uint64_t ngram_key_end(uint64_t k) { return k == 0 ? 1 : k; }

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
  uint32_t word_id;   // LM vocabulary index when PF_LM_WORD, else 0 (<unk>)
  uint32_t flags;
};

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
struct TokInfo {  // 64 B
  uint64_t h_raw, pow_raw;      // label as appended by branch D (decoder.py:528)
  uint64_t h_clean, pow_clean;  // label without boundary marks, as started by branch B (:477-481)
  uint32_t len_raw, len_clean;  // code points
  uint32_t flags;
  // prefix-table view of the CLEAN label taken as the start of a new word (static per LM)
  uint32_t start_flags;  // PF_* bits, plus PF_ON_TABLE
  uint32_t start_word_id;
  uint32_t pad[3];
};
constexpr uint32_t PF_ON_TABLE = 8u;  // beam flag: the partial is a stored prefix

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
enum : uint32_t { BR_BOUNDARY = 1u, BR_SPACE = 2u, BR_APPEND = 3u, BR_FINAL = 4u };
struct EmitNode {  // 16 B
  uint32_t parent;
  uint32_t tok_branch;  // token | branch << 16
  int32_t wstart, wend; // frames of the word closed by this emission (BOUNDARY/SPACE/FINAL)
};

// per-beam result record written by the beam kernel
struct OutBeam {  // 80 B
  double logit_score;
  double lm_score;
  uint32_t tok_off;   // offset of this beam's emission list in the token pool
  uint32_t tok_cnt;
  LmState state;      // last_lm_state
  uint32_t pad;
};

struct DeviceTables {
  const TokInfo* tok;
  const TokHot* tok_hot;
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
};

// ---------------------------------------------------------------------------------------------
// table probes (host + device)
// ---------------------------------------------------------------------------------------------
CTC_HD bool prefix_lookup(const PrefixEntry* tab, uint64_t mask, uint64_t key, uint32_t* word_id,
                          uint32_t* flags) {
  if (!tab || key == 0) return false;
  uint64_t s = mix64(key) & mask;
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
  uint64_t s = mix64(key) & mask;
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
  uint64_t s = mix64(key) & mask;
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
CTC_HD float lm_base_score(const DeviceTables& t, const LmState& in, uint32_t wid, LmState* out) {
  UnigramEntry u = t.unigrams[wid];
  const int in_len = in.len;
  const int max_n = !t.ngrams ? 1 : ((int)t.lm_order < in_len + 1 ? (int)t.lm_order : in_len + 1);
  // First probe of every order is issued up front (independent loads overlap their latency);
  // kenlm walks the orders one after the other, the result is the same longest match.
  uint64_t keys[MAX_CTX + 2];
  uint64_t slots[MAX_CTX + 2];
  NgramEntry ent[MAX_CTX + 2];
#pragma unroll
  for (int n = 2; n <= MAX_CTX + 1; ++n) {
    keys[n] = 0;
    slots[n] = 0;
    ent[n].key = 0;
    ent[n].prob = 0.f;
    ent[n].backoff = 0.f;
    if (n <= max_n) {
      uint64_t k = ngram_key_begin((uint32_t)n);
#pragma unroll
      for (int c = MAX_CTX - 1; c >= 0; --c)
        if (c <= n - 2) k = ngram_key_push(k, in.words[c]);
      k = ngram_key_end(ngram_key_push(k, wid));
      keys[n] = k;
      slots[n] = mix64(k) & t.ngram_mask;
      ent[n] = t.ngrams[slots[n]];
    }
  }
  float prob = u.prob;
  float obo[MAX_CTX + 1];
#pragma unroll
  for (int k = 0; k <= MAX_CTX; ++k) obo[k] = 0.f;
  obo[0] = u.backoff;
  int matched = 1;
  bool go = true;
#pragma unroll
  for (int n = 2; n <= MAX_CTX + 1; ++n) {
    if (go && n <= max_n) {
      NgramEntry e = ent[n];
      uint64_t s = slots[n];
      while (e.key != keys[n] && e.key != 0) {  // linear probing past a collision (rare)
        s = (s + 1) & t.ngram_mask;
        e = t.ngrams[s];
      }
      if (e.key == keys[n]) {
        prob = e.prob;
        obo[n - 1] = e.backoff;
        matched = n;
      } else {
        go = false;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAX_CTX; ++i)
    if (i >= matched - 1 && i < in_len) prob = prob + in.backoff[i];  // fp32, shortest context first
  const int keep = matched < (int)t.lm_order - 1 ? matched : (int)t.lm_order - 1;
  LmState o;
  o.len = keep;
#pragma unroll
  for (int k = 0; k < MAX_CTX; ++k) {
    bool on = k < keep;
    o.words[k] = on ? (k == 0 ? wid : in.words[k > 0 ? k - 1 : 0]) : 0u;
    o.backoff[k] = on ? obo[k] : 0.f;
  }
  *out = o;
  return prob;
}

}  // namespace ctc
