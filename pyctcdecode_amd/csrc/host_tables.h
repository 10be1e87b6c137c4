// host_tables.h -- host-side builders for every read-only table the kernels consume: label
// constants, ARPA -> flat hashed n-gram trie (replaces kenlm.Model, decoder.py:1074), vocabulary
// prefix table (replaces the pygtrie unigram trie, language_model.py:263) and the per-call
// hot-word table (replaces HotwordScorer.build_scorer, language_model.py:152-189).
// Pure C++ (no HIP): shared by the library (api.cpp) and the CPU simulator (tests/sim).
#pragma once
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace ctc {

uint64_t hash_bytes(const char* s, size_t n);                 // H(s)
uint64_t pow_base(size_t nbytes);                             // STR_BASE^nbytes (both halves, packed)
uint32_t utf8_length(const char* s, size_t n);                // code points
// byte offsets of every code-point boundary after the first code point, including n
void utf8_boundaries(const char* s, size_t n, std::vector<size_t>* out);

struct HostAlphabet {
  std::vector<std::string> labels;  // normalised
  bool is_bpe = false;
  std::vector<std::string> clean;   // label without boundary marks
  std::vector<TokInfo> tok;         // start_* fields filled by HostLM::fill_token_starts
  void build(const std::vector<std::string>& labels_, bool is_bpe_);
};

// The hashed n-gram table: by far the largest part of a model (64 MB per million n-grams) and immutable once loaded, so
// a model, its clones (ctcdec_lm_clone: own unigram set and prefix table over the same n-grams) and every decoder that
// holds one of them share ONE host copy and ONE device upload. `device` is owned by whoever uploads (api.cpp: a DevBuf
// released with the last reference).
struct NgramStore {
  std::vector<NgramEntry> table;  // open addressing, empty key 0
  std::shared_ptr<void> device;
};

struct HostLM {
  int order = 0;
  std::vector<std::string> words;  // id -> string, id 0 = <unk>
  std::unordered_map<std::string, uint32_t> vocab;
  std::vector<UnigramEntry> unigrams;
  std::shared_ptr<NgramStore> ngr = std::make_shared<NgramStore>();  // copies of a HostLM share it
  const std::vector<NgramEntry>& ngram_table() const { return ngr->table; }
  uint64_t ngram_mask = 0;
  uint32_t bos_id = 0, eos_id = 0;
  bool has_trie = false;          // unigrams is not None
  size_t uniset_size = 0;         // |unigram_set| after filtering to the LM vocabulary
  std::vector<uint8_t> in_uniset; // by word id
  std::vector<PrefixEntry> prefix_table;
  uint64_t prefix_mask = 0;
  size_t n_ngrams = 0;

  // ARPA -> kenlm binary writer only (kenlm_binary.cpp): the n-grams as listed, with the keys of their two (n-1)-gram halves
  struct RawNgram {
    int order;
    uint64_t key, suffix_key, prefix_key;  // kenlm's chain over w1..wn / w2..wn / w1..w(n-1) (before ngram_key_end)
    float prob, backoff;
  };
  bool keep_raw = false;
  bool unk_listed = false;  // the ARPA file lists <unk> itself
  std::vector<RawNgram> raw_ngrams;

  // returns "" on success, else an error message
  std::string load_arpa(const std::string& path);
  // a kenlm PROBING binary (build_binary probing): kenlm_binary.cpp -- format unpinned against real kenlm, see there
  std::string load_kenlm_binary(const std::string& path);
  // The parsed model as one flat file (vocabulary, unigram array, hashed n-gram table exactly as they
  // are uploaded): loading it is a few freads instead of an ARPA parse.  Same return convention.
  std::string save_cache(const std::string& path) const;
  std::string load_cache(const std::string& path);
  uint32_t index(const std::string& w) const;  // 0 for OOV and for "<unk>"
  void set_unigrams(bool has, const std::vector<std::string>& unigrams);
  void build_prefix_table();
  void fill_token_starts(HostAlphabet* alpha) const;
  void start_state(bool begin_sentence, LmState* out) const;
  void tables(DeviceTables* t) const;  // host pointers (for the host-side query)
};

bool looks_like_kenlm_binary(const std::string& path);  // by its magic bytes
// ARPA -> kenlm probing binary, as far as kenlm's sources say (test infrastructure for the reader + a converter)
std::string arpa_to_kenlm_binary(const std::string& arpa_path, const std::string& out_path, float probing_multiplier);

// TokInfo.start_* of every label, looked up in a vocabulary prefix table
void fill_token_starts_from(const std::vector<PrefixEntry>& table, uint64_t mask, HostAlphabet* alpha);

// MultiLanguageModel (language_model.py:455-502): the union word list of the member models, each
// model's view of it, and one prefix table that carries every model's unigram-trie bit.
struct HostMulti {
  std::vector<std::shared_ptr<HostLM>> lms;  // 2..MAX_LMS
  std::vector<std::string> words;            // union index -> string; 0 = no word
  std::vector<std::vector<uint32_t>> winfo;  // [lm][union index]: local id | WI_* flags
  std::vector<PrefixEntry> prefix_table;
  uint64_t prefix_mask = 0;
  int order = 0;                             // max over the models (language_model.py:467-469)
  void build();
};

struct HostHotwords {
  std::vector<HotEntry> table;
  uint64_t mask = 0;
  std::vector<TokHot> tok_hot;
  // unigrams: already stripped/split hot-word unigrams
  void build(const std::vector<std::string>& unigrams, const HostAlphabet& alpha);
};

}  // namespace ctc
