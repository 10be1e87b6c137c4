// kenlm_binary.cpp -- kenlm's PROBING binary model files (what `build_binary probing x.arpa x.bin` writes and
// `kenlm.Model("x.bin")` maps: /root/reference/pyctcdecode/decoder.py:1074, language_model.py:424 accept `.bin` / `.binary`).
//
// FORMAT UNPINNED AGAINST REAL KENLM. kenlm (github.com/kpu/kenlm) is an external dependency of the reference that is absent
// from this container, and so is any file it wrote: the layout below is restated from the published sources (lm/binary_format.cc,
// lm/vocab.cc, lm/search_hashed.{hh,cc}, lm/weights.hh, util/probing_hash_table.hh, util/murmur_hash.cc) and pinned only
// against this file's own writer (tests/test_kenlm_binary.py: ARPA -> writer -> reader == ARPA loader, every n-gram, every
// score). What protects a user against a misremembered detail: the reader checks everything the format lets it check -- the
// magic and sanity block, the model type, the file size against the sizes the header implies, and EVERY vocabulary string
// against the vocabulary hash table (MurmurHash64A of the word must be found there with the word's index): a layout that is off
// by a byte fails loudly instead of scoring wrongly.
//
//   [0]    Sanity (88 B): magic "mmap lm http://kheafield.com/code format version 5\n\0" in 56 bytes; float 0, 1, -0.5;
//          uint32 1, 0xFFFFFFFF; (4 B padding); uint64 1                                        -- byte order / type sizes
//   [88]   FixedWidthParameters (20 B): u8 order, (3) float probing_multiplier, int32 model_type, u8 has_vocabulary, (3),
//          uint32 search_version
//   [108]  uint64 counts[order]; header padded to a multiple of 8
//   vocab  ProbingVocabulary: {uint32 version, uint32 bound}, then max(count1 + 1, multiplier * count1) buckets of
//          {uint64 MurmurHash64A(word, seed 0), uint32 index} (12 B, key 0 = empty), linear probing from hash % buckets
//   search unigrams: (count1 + 1) x {float prob, float backoff} by word index; then for n = 2 .. order-1 a table of
//          max(count_n + 1, multiplier * count_n) buckets {uint64 key, float prob, float backoff} (16 B); then the longest
//          order's table of {uint64 key, float prob} (12 B). key = kenlm's CombineWordHash chain (common.h: ngram_key_*),
//          bucket = key % buckets, linear probing. A prob's sign bit is a flag (cleared: the n-gram extends to the left; the
//          value is -|prob|), a backoff of -0.0 marks "no extension to the right".
//   words  the vocabulary strings in index order ("<unk>" first), each followed by a 0 byte
//
// The n-grams' WORDS are not in such a file, only their 64-bit keys: that is why the flat trie of this library uses kenlm's own
// key chain since round 6 -- the tables are adopted entry by entry (key ^ order << 56, -|prob|, backoff).
// Other model types (trie, quantised, array-compressed, rest-cost probing) are refused by name.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "host_tables.h"

namespace ctc {

static const char kKenlmMagic[] = "mmap lm http://kheafield.com/code format version 5\n";  // + "\0" + the literal's own terminator
static const size_t kSanityBytes = 88, kFixedBytes = 20;

static uint64_t murmur64a(const void* key, size_t len, uint64_t seed) {
  const uint64_t m = 0xc6a4a7935bd1e995ull;
  const int r = 47;
  uint64_t h = seed ^ (len * m);
  const unsigned char* p = (const unsigned char*)key;
  const unsigned char* end = p + (len / 8) * 8;
  for (; p != end; p += 8) {
    uint64_t k;
    memcpy(&k, p, 8);
    k *= m;
    k ^= k >> r;
    k *= m;
    h ^= k;
    h *= m;
  }
  switch (len & 7) {
    case 7: h ^= (uint64_t)p[6] << 48;  // fall through
    case 6: h ^= (uint64_t)p[5] << 40;  // fall through
    case 5: h ^= (uint64_t)p[4] << 32;  // fall through
    case 4: h ^= (uint64_t)p[3] << 24;  // fall through
    case 3: h ^= (uint64_t)p[2] << 16;  // fall through
    case 2: h ^= (uint64_t)p[1] << 8;   // fall through
    case 1: h ^= (uint64_t)p[0]; h *= m;
  }
  h ^= h >> r;
  h *= m;
  h ^= h >> r;
  return h;
}

static uint64_t probing_buckets(uint64_t entries, float multiplier) {
  return std::max<uint64_t>(entries + 1, (uint64_t)(multiplier * (float)entries));
}
static size_t align8(size_t x) { return (x + 7) & ~(size_t)7; }

static const char* kenlm_type_name(int32_t t) {
  switch (t) {
    case 0: return "probing";
    case 1: return "rest-cost probing";
    case 2: return "trie";
    case 3: return "quantised trie";
    case 4: return "array-compressed trie";
    case 5: return "quantised array-compressed trie";
    default: return "unknown";
  }
}

bool looks_like_kenlm_binary(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char head[32];
  const bool ok = fread(head, 1, sizeof(head), f) == sizeof(head) && memcmp(head, "mmap lm http://kheafield.com/code", 32) == 0;
  fclose(f);
  return ok;
}

std::string HostLM::load_kenlm_binary(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return "cannot open LM file " + path;
  fseek(f, 0, SEEK_END);
  const uint64_t file_size = (uint64_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<unsigned char> buf(file_size);
  const bool read_ok = file_size == 0 || fread(buf.data(), 1, file_size, f) == file_size;
  fclose(f);
  if (!read_ok) return "short read from " + path;
  auto bad = [&](const std::string& why) { return "kenlm binary " + path + ": " + why; };
  if (file_size < kSanityBytes + kFixedBytes + 8) return bad("too short for a header");
  if (memcmp(buf.data(), "mmap lm http://kheafield.com/code format version ", 49) != 0) return bad("not a kenlm binary (magic)");
  if (memcmp(buf.data(), kKenlmMagic, sizeof(kKenlmMagic)) != 0)
    return bad("format version " + std::string((const char*)buf.data() + 49, 1) + " (only version 5 is read)");
  float sf[3];
  uint32_t su[2];
  uint64_t s64;
  memcpy(sf, buf.data() + 56, 12);
  memcpy(su, buf.data() + 68, 8);
  memcpy(&s64, buf.data() + 80, 8);
  if (sf[0] != 0.0f || sf[1] != 1.0f || sf[2] != -0.5f || su[0] != 1u || su[1] != 0xFFFFFFFFu || s64 != 1ull)
    return bad("sanity block mismatch (written on a machine with another byte order or other type sizes)");
  const unsigned char* fx = buf.data() + kSanityBytes;
  const int ord = fx[0];
  float multiplier;
  int32_t model_type;
  uint32_t search_version;
  memcpy(&multiplier, fx + 4, 4);
  memcpy(&model_type, fx + 8, 4);
  const bool has_vocab = fx[12] != 0;
  memcpy(&search_version, fx + 16, 4);
  if (model_type != 0)
    return bad(std::string("model type '") + kenlm_type_name(model_type) + "' is not supported: only the probing hash-table "
               "format is read (rebuild with `build_binary probing`, or load the ARPA file)");
  if (search_version != 0) return bad("probing search version " + std::to_string(search_version) + " (only 0 is read)");
  if (ord < 1 || ord > MAX_CTX + 1) return bad("order " + std::to_string(ord) + " outside 1 .. " + std::to_string(MAX_CTX + 1));
  if (!(multiplier > 1.0f) || multiplier > 64.0f) return bad("probing multiplier out of range");
  if (!has_vocab) return bad("written without its vocabulary strings (build_binary -w ...): the words cannot be recovered from hashes");
  if (file_size < kSanityBytes + kFixedBytes + 8ull * ord) return bad("truncated header");
  std::vector<uint64_t> counts(ord);
  memcpy(counts.data(), fx + kFixedBytes, 8 * (size_t)ord);
  for (uint64_t c : counts)
    if (c > (1ull << 40)) return bad("implausible n-gram count");
  if (counts[0] + 1 > WI_ID_MASK) return bad("vocabulary too large");
  size_t pos = align8(kSanityBytes + kFixedBytes + 8 * (size_t)ord);
  // ---- vocabulary
  const uint64_t vbuckets = probing_buckets(counts[0], multiplier);
  const size_t vocab_bytes = 8 + (size_t)vbuckets * 12;
  if (pos + vocab_bytes > file_size) return bad("truncated vocabulary table");
  uint32_t vhead[2];
  memcpy(vhead, buf.data() + pos, 8);
  const uint32_t bound = vhead[1];
  if (bound != counts[0] && bound != counts[0] + 1) return bad("vocabulary bound does not match the unigram count");
  const unsigned char* vtab = buf.data() + pos + 8;
  pos += vocab_bytes;
  // ---- search: sizes
  const size_t uni_bytes = (size_t)(counts[0] + 1) * 8;
  std::vector<uint64_t> buckets(ord, 0);
  size_t search_bytes = uni_bytes;
  for (int n = 2; n <= ord; ++n) {
    buckets[n - 1] = probing_buckets(counts[n - 1], multiplier);
    search_bytes += (size_t)buckets[n - 1] * (n == ord ? 12 : 16);
  }
  if (pos + search_bytes > file_size) return bad("truncated n-gram tables");
  const unsigned char* uni = buf.data() + pos;
  const size_t words_at = pos + search_bytes;
  // ---- vocabulary strings, each checked against the hash table (a misread layout fails here)
  std::vector<std::string> w;
  w.reserve(bound);
  {
    size_t a = words_at;
    while (a < file_size && w.size() < bound) {
      const void* z = memchr(buf.data() + a, 0, file_size - a);
      if (!z) return bad("unterminated vocabulary string");
      const size_t b = (size_t)((const unsigned char*)z - buf.data());
      w.emplace_back((const char*)buf.data() + a, b - a);
      a = b + 1;
    }
    if (w.size() != bound) return bad("fewer vocabulary strings than words");
    if (a != file_size) return bad("bytes after the last vocabulary string");
  }
  if (w.empty() || w[0] != "<unk>") return bad("the first vocabulary string is not <unk>");
  for (uint32_t id = 1; id < bound; ++id) {
    const uint64_t h = murmur64a(w[id].data(), w[id].size(), 0);
    uint64_t s = h % vbuckets;
    bool found = false;
    for (uint64_t step = 0; step < vbuckets; ++step) {
      uint64_t k;
      uint32_t v;
      memcpy(&k, vtab + s * 12, 8);
      memcpy(&v, vtab + s * 12 + 8, 4);
      if (k == h) {
        found = v == id;
        break;
      }
      if (k == 0) break;
      s = s + 1 == vbuckets ? 0 : s + 1;
    }
    if (!found)
      return bad("vocabulary string " + std::to_string(id) + " ('" + w[id] + "') is not in the vocabulary hash table under its "
                 "index: the file's layout is not what this reader assumes");
  }
  // ---- adopt
  words = w;
  vocab.clear();
  vocab.reserve(words.size() * 2);
  for (uint32_t id = 1; id < bound; ++id) vocab.emplace(words[id], id);
  vocab["<unk>"] = 0;
  unigrams.assign(words.size(), UnigramEntry{0.f, 0.f});
  for (uint32_t id = 0; id < bound; ++id) {
    float pb[2];
    memcpy(pb, uni + (size_t)id * 8, 8);
    unigrams[id] = UnigramEntry{-fabsf(pb[0]), pb[1] + 0.0f};
  }
  uint64_t total = 0;
  for (int n = 2; n <= ord; ++n) total += counts[n - 1];
  uint64_t size = 16;
  while (size < 4 * total + 1) size <<= 1;
  ngr = std::make_shared<NgramStore>();
  std::vector<NgramEntry>& tab = ngr->table;
  tab.assign(size, NgramEntry{0, 0.f, 0.f});
  ngram_mask = size - 1;
  n_ngrams = 0;
  const unsigned char* t = uni + uni_bytes;
  for (int n = 2; n <= ord; ++n) {
    const size_t esz = n == ord ? 12 : 16;
    for (uint64_t bkt = 0; bkt < buckets[n - 1]; ++bkt, t += esz) {
      uint64_t k;
      memcpy(&k, t, 8);
      if (k == 0) continue;  // empty bucket
      float p, b = 0.f;
      memcpy(&p, t + 8, 4);
      if (esz == 16) memcpy(&b, t + 12, 4);
      const uint64_t key = ngram_key_end(k, (uint32_t)n);
      uint64_t s = key & ngram_mask;
      while (tab[s].key != 0 && tab[s].key != key) s = (s + 1) & ngram_mask;
      if (tab[s].key == key) return bad("two n-grams share a 64-bit key");
      tab[s] = NgramEntry{key, -fabsf(p), b + 0.0f};
      if (++n_ngrams > total + (uint64_t)ord) return bad("more n-grams than the header counts");
    }
  }
  order = ord;
  bos_id = index("<s>");
  eos_id = index("</s>");
  in_uniset.assign(words.size(), 0);
  has_trie = false;
  uniset_size = 0;
  build_prefix_table();
  return "";
}

// ARPA -> kenlm probing binary (what `build_binary probing` writes, as far as the sources say: see the header of this file).
std::string arpa_to_kenlm_binary(const std::string& arpa_path, const std::string& out_path, float multiplier) {
  HostLM lm;
  lm.keep_raw = true;
  std::string e = lm.load_arpa(arpa_path);
  if (!e.empty()) return e;
  const int ord = lm.order;
  if (!(multiplier > 1.0f)) multiplier = 1.5f;
  std::vector<uint64_t> counts(ord, 0);
  counts[0] = lm.unk_listed ? lm.words.size() : lm.words.size() - 1;
  for (const HostLM::RawNgram& g : lm.raw_ngrams) counts[g.order - 1] += 1;
  // which n-grams extend to the left / right (the two flag bits of the stored weights)
  std::unordered_set<uint64_t> has_left_ext, has_right_ext;  // keyed by ngram_key_end(key, order)
  for (const HostLM::RawNgram& g : lm.raw_ngrams) {
    has_left_ext.insert(ngram_key_end(g.suffix_key, (uint32_t)(g.order - 1)));   // w2..wn is extended by w1 on the left
    has_right_ext.insert(ngram_key_end(g.prefix_key, (uint32_t)(g.order - 1)));  // w1..w(n-1) is extended by wn on the right
  }
  std::vector<unsigned char> out;
  auto put = [&](const void* p, size_t n) { out.insert(out.end(), (const unsigned char*)p, (const unsigned char*)p + n); };
  auto pad_to = [&](size_t n) { out.resize(n, 0); };
  put(kKenlmMagic, sizeof(kKenlmMagic));  // 52 bytes with the literal's terminator; the explicit "\0" of kenlm's literal follows
  pad_to(56);
  const float sf[3] = {0.0f, 1.0f, -0.5f};
  const uint32_t su[2] = {1u, 0xFFFFFFFFu};
  const uint64_t s64 = 1;
  put(sf, 12);
  put(su, 8);
  pad_to(80);
  put(&s64, 8);
  unsigned char fx[20] = {0};
  fx[0] = (unsigned char)ord;
  memcpy(fx + 4, &multiplier, 4);
  const int32_t mt = 0;
  memcpy(fx + 8, &mt, 4);
  fx[12] = 1;
  const uint32_t sv = 0;
  memcpy(fx + 16, &sv, 4);
  put(fx, 20);
  put(counts.data(), 8 * (size_t)ord);
  pad_to(align8(out.size()));
  // vocabulary
  const uint64_t vb = probing_buckets(counts[0], multiplier);
  const uint32_t vhead[2] = {0u, (uint32_t)lm.words.size()};
  put(vhead, 8);
  {
    std::vector<unsigned char> vt((size_t)vb * 12, 0);
    for (uint32_t id = 1; id < lm.words.size(); ++id) {
      const uint64_t h = murmur64a(lm.words[id].data(), lm.words[id].size(), 0);
      uint64_t s = h % vb;
      for (;;) {
        uint64_t k;
        memcpy(&k, vt.data() + s * 12, 8);
        if (k == 0) break;
        if (k == h) return "two vocabulary words share a MurmurHash64A value";
        s = s + 1 == vb ? 0 : s + 1;
      }
      memcpy(vt.data() + s * 12, &h, 8);
      memcpy(vt.data() + s * 12 + 8, &id, 4);
    }
    put(vt.data(), vt.size());
  }
  // unigrams (count + 1 entries: one to spare for a hallucinated <unk>)
  auto flagged = [&](float p, bool extends_left) {
    uint32_t u;
    float v = -fabsf(p);
    memcpy(&u, &v, 4);
    if (extends_left) u &= 0x7FFFFFFFu;
    memcpy(&v, &u, 4);
    return v;
  };
  for (uint64_t id = 0; id < counts[0] + 1; ++id) {
    float pb[2] = {0.f, 0.f};
    if (id < lm.words.size()) {
      const uint64_t k1 = ngram_key_end(ngram_key_first((uint32_t)id), 1);
      pb[0] = flagged(lm.unigrams[id].prob, has_left_ext.count(k1) != 0);
      pb[1] = lm.unigrams[id].backoff;
      if (pb[1] == 0.0f && !has_right_ext.count(k1)) pb[1] = -0.0f;
    }
    put(pb, 8);
  }
  for (int n = 2; n <= ord; ++n) {
    const uint64_t nb = probing_buckets(counts[n - 1], multiplier);
    const size_t esz = n == ord ? 12 : 16;
    std::vector<unsigned char> tb((size_t)nb * esz, 0);
    for (const HostLM::RawNgram& g : lm.raw_ngrams) {
      if (g.order != n) continue;
      uint64_t s = g.key % nb;
      for (;;) {
        uint64_t k;
        memcpy(&k, tb.data() + s * esz, 8);
        if (k == 0 || k == g.key) break;  // (an n-gram listed twice overwrites itself)
        s = s + 1 == nb ? 0 : s + 1;
      }
      const uint64_t kn = ngram_key_end(g.key, (uint32_t)n);
      const float p = flagged(g.prob, n < ord && has_left_ext.count(kn) != 0);
      float b = g.backoff;
      if (b == 0.0f && !has_right_ext.count(kn)) b = -0.0f;
      memcpy(tb.data() + s * esz, &g.key, 8);
      memcpy(tb.data() + s * esz + 8, &p, 4);
      if (esz == 16) memcpy(tb.data() + s * esz + 12, &b, 4);
    }
    put(tb.data(), tb.size());
  }
  for (const std::string& word : lm.words) put(word.c_str(), word.size() + 1);
  FILE* f = fopen(out_path.c_str(), "wb");
  if (!f) return "cannot write " + out_path;
  const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
  return (fclose(f) == 0 && ok) ? "" : "short write to " + out_path;
}

}  // namespace ctc
