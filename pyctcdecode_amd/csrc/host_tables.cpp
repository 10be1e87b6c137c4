// host_tables.cpp -- see host_tables.h
#include "host_tables.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

namespace ctc {

uint64_t hash_bytes(const char* s, size_t n) {
  uint64_t h = 0;
  for (size_t i = 0; i < n; ++i) h = str_push_byte(h, (unsigned char)s[i]);
  return h;
}

uint64_t pow_base(size_t nbytes) {
  uint64_t r = (1ull << 32) | 1ull;  // BASE^0 in both halves
  for (size_t i = 0; i < nbytes; ++i) r = str_concat(r, STR_BASE, 0);
  return r;
}

uint32_t utf8_length(const char* s, size_t n) {
  uint32_t c = 0;
  for (size_t i = 0; i < n; ++i)
    if (((unsigned char)s[i] & 0xC0) != 0x80) ++c;
  return c;
}

void utf8_boundaries(const char* s, size_t n, std::vector<size_t>* out) {
  out->clear();
  for (size_t i = 1; i < n; ++i)
    if (((unsigned char)s[i] & 0xC0) != 0x80) out->push_back(i);
  if (n > 0) out->push_back(n);
}

static const char kMark[] = "\xE2\x96\x81";  // U+2581

static bool starts_with_mark(const std::string& s) { return s.size() >= 3 && memcmp(s.data(), kMark, 3) == 0; }
static bool ends_with_mark(const std::string& s) {
  return s.size() >= 3 && memcmp(s.data() + s.size() - 3, kMark, 3) == 0;
}

void HostAlphabet::build(const std::vector<std::string>& labels_, bool is_bpe_) {
  labels = labels_;
  is_bpe = is_bpe_;
  clean.assign(labels.size(), std::string());
  tok.assign(labels.size(), TokInfo());
  for (size_t i = 0; i < labels.size(); ++i) {
    const std::string& l = labels[i];
    TokInfo t;
    memset(&t, 0, sizeof(t));
    std::string c = l;
    if (l.empty()) t.flags |= TK_BLANK;
    if (!is_bpe && l == " ") t.flags |= TK_SPACE;
    if (is_bpe) {
      // decoder.py:477-482: strip a leading mark, then a trailing one (tested on the label itself)
      if (starts_with_mark(l)) {
        t.flags |= TK_LEAD;
        c = c.substr(3);
      }
      if (ends_with_mark(l)) {
        t.flags |= TK_TRAIL;
        if (c.size() >= 3) c = c.substr(0, c.size() - 3);
      }
    }
    clean[i] = c;
    t.h_raw = hash_bytes(l.data(), l.size());
    t.pow_raw = pow_base(l.size());
    t.len_raw = utf8_length(l.data(), l.size());
    t.h_clean = hash_bytes(c.data(), c.size());
    t.pow_clean = pow_base(c.size());
    t.len_clean = utf8_length(c.data(), c.size());
    tok[i] = t;
  }
}

// ---------------------------------------------------------------------------------------------
static void split(const std::string& s, char sep, std::vector<std::string>* out) {
  out->clear();
  size_t a = 0;
  for (;;) {
    size_t b = s.find(sep, a);
    if (b == std::string::npos) {
      out->push_back(s.substr(a));
      return;
    }
    out->push_back(s.substr(a, b - a));
    a = b + 1;
  }
}

static void split_ws(const std::string& s, std::vector<std::string>* out) {
  out->clear();
  size_t i = 0, n = s.size();
  while (i < n) {
    while (i < n && (s[i] == ' ' || s[i] == '\t')) ++i;
    size_t a = i;
    while (i < n && s[i] != ' ' && s[i] != '\t') ++i;
    if (i > a) out->push_back(s.substr(a, i - a));
  }
}

static std::string strip(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r' || s[a] == '\n')) ++a;
  while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r' || s[b - 1] == '\n')) --b;
  return s.substr(a, b - a);
}

struct RawGram {
  uint64_t key, chk;
  float prob, backoff;
};

// chk: a second, independent 64-bit hash of the same word-id tuple (kept only while the table is built). Two entries with one
// key and different check values are two DIFFERENT n-grams whose keys collide: the model is refused rather than scored
// wrongly (the key is all the device compares). An n-gram listed twice in the file overwrites itself, as before.
static bool table_put(std::vector<NgramEntry>& tab, std::vector<uint64_t>& chk_tab, uint64_t mask, uint64_t key, uint64_t chk,
                      float p, float b) {
  uint64_t s = key & mask;
  for (;;) {
    if (tab[s].key == 0 || tab[s].key == key) {
      if (tab[s].key == key && chk_tab[s] != chk) return false;
      tab[s].key = key;
      tab[s].prob = p;
      tab[s].backoff = b;
      chk_tab[s] = chk;
      return true;
    }
    s = (s + 1) & mask;
  }
}

std::string HostLM::load_arpa(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return "cannot open LM file " + path;
  words.clear();
  vocab.clear();
  unigrams.clear();
  raw_ngrams.clear();
  unk_listed = false;
  words.push_back("<unk>");
  vocab["<unk>"] = 0;
  unigrams.push_back(UnigramEntry{-100.0f, 0.0f});  // kenlm default when <unk> is missing
  std::vector<RawGram> raw;
  int section = 0, max_order_header = 0;
  std::string line;
  std::vector<std::string> fields, toks;
  char* buf = nullptr;
  size_t cap = 0;
  ssize_t got;
  bool seen_data = false;
  while ((got = getline(&buf, &cap, f)) >= 0) {
    line.assign(buf, (size_t)got);
    line = strip(line);
    if (line.empty()) continue;
    if (line[0] == '\\') {
      if (line == "\\data\\") {
        section = 0;
        seen_data = true;
      } else if (line == "\\end\\") {
        break;
      } else {
        size_t dash = line.find('-');
        if (dash != std::string::npos && line.size() > 7 && line.compare(line.size() - 7, 7, "-grams:") == 0) {
          section = atoi(line.substr(1, dash - 1).c_str());
          if (section < 1 || section > 16) {
            fclose(f);
            free(buf);
            return "bad ARPA section header: " + line;
          }
        }
      }
      continue;
    }
    if (section == 0) {
      if (line.compare(0, 6, "ngram ") == 0) {
        size_t eq = line.find('=');
        if (eq != std::string::npos) max_order_header = std::max(max_order_header, atoi(line.substr(6, eq - 6).c_str()));
      }
      continue;
    }
    float prob, backoff = 0.0f;
    if (line.find('\t') != std::string::npos) {
      split(line, '\t', &fields);
      prob = strtof(fields[0].c_str(), nullptr);
      if (fields.size() < 2) {
        fclose(f);
        free(buf);
        return "malformed ARPA line: " + line;
      }
      split(fields[1], ' ', &toks);
      if (fields.size() > 2 && !fields[2].empty()) backoff = strtof(fields[2].c_str(), nullptr);
    } else {
      split_ws(line, &fields);
      prob = strtof(fields[0].c_str(), nullptr);
      toks.assign(fields.begin() + 1, fields.begin() + std::min(fields.size(), (size_t)1 + section));
      if (fields.size() > (size_t)1 + section) backoff = strtof(fields[1 + section].c_str(), nullptr);
    }
    if ((int)toks.size() != section) {
      fclose(f);
      free(buf);
      return "malformed ARPA line in " + std::to_string(section) + "-gram section: " + line;
    }
    if (prob > 0) {
      fclose(f);
      free(buf);
      return "positive log probability in ARPA: " + line;
    }
    if (section == 1) {
      const std::string& w = toks[0];
      uint32_t id;
      if (w == "<unk>") {
        id = 0;
        unk_listed = true;
      } else {
        auto it = vocab.find(w);
        if (it == vocab.end()) {
          id = (uint32_t)words.size();
          vocab.emplace(w, id);
          words.push_back(w);
          unigrams.push_back(UnigramEntry{0.f, 0.f});
        } else {
          id = it->second;
        }
      }
      unigrams[id] = UnigramEntry{prob, backoff};
    } else {
      uint64_t k = 0, chk = 0x6A09E667F3BCC909ull, k_suffix = 0;  // newest word first (common.h)
      for (int j = section - 1; j >= 0; --j) {
        const uint32_t id = index(toks[j]);
        if (j == 0) k_suffix = k;  // the chain over w2..wn
        k = j == section - 1 ? ngram_key_first(id) : ngram_key_push(k, id);
        chk = mix64(chk ^ (uint64_t)id) + 0x13198A2E03707344ull;
      }
      if (keep_raw) {
        uint64_t k_prefix = 0;  // the chain over w1..w(n-1)
        for (int j = section - 2; j >= 0; --j) k_prefix = j == section - 2 ? ngram_key_first(index(toks[j])) : ngram_key_push(k_prefix, index(toks[j]));
        raw_ngrams.push_back(RawNgram{section, k, k_suffix, k_prefix, prob, backoff});
      }
      raw.push_back(RawGram{ngram_key_end(k, (uint32_t)section), chk ^ (uint64_t)section, prob, backoff});
    }
  }
  free(buf);
  fclose(f);
  if (!seen_data && words.size() <= 1) return "not an ARPA file: " + path;
  order = max_order_header;
  if (order <= 0) return "ARPA header has no ngram counts: " + path;
  if (order > MAX_CTX + 1) return "LM order " + std::to_string(order) + " exceeds the supported maximum";
  n_ngrams = raw.size();
  uint64_t size = 16;
  while (size < 4 * raw.size() + 1) size <<= 1;  // load <= 1/4: a miss (the common case when backing off) costs ~1.2 probes
  ngr = std::make_shared<NgramStore>();  // (never the table another model may be sharing)
  std::vector<NgramEntry>& ngram_table = ngr->table;
  ngram_table.assign(size, NgramEntry{0, 0.f, 0.f});
  ngram_mask = size - 1;
  {
    std::vector<uint64_t> chk_table(size, 0);
    for (const RawGram& g : raw)
      if (!table_put(ngram_table, chk_table, ngram_mask, g.key, g.chk, g.prob, g.backoff))
        return "two different n-grams of " + path + " hash to one 64-bit key: refusing the model (not scoring it wrongly)";
  }
  bos_id = index("<s>");
  eos_id = index("</s>");
  in_uniset.assign(words.size(), 0);
  has_trie = false;
  uniset_size = 0;
  build_prefix_table();
  return "";
}

// ---- flat model file: "CTCDLM01" | order, n_words, bos, eos (u32) | n_ngrams, table_size, blob_bytes (u64)
//      | word end offsets u64[n_words] | word bytes | UnigramEntry[n_words] | NgramEntry[table_size] | "CTCDEND1"
static const char kCacheMagic[8] = {'C', 'T', 'C', 'D', 'L', 'M', '0', '4'};  // 04: n-gram keys by kenlm's CombineWordHash chain (03: mix_step, 02: splitmix), newest-first, slot = key & mask
static const char kCacheEnd[8] = {'C', 'T', 'C', 'D', 'E', 'N', 'D', '1'};

std::string HostLM::save_cache(const std::string& path) const {
  if (order <= 0) return "no language model loaded";
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return "cannot write " + path;
  bool ok = fwrite(kCacheMagic, 1, 8, f) == 8;
  const uint32_t h32[4] = {(uint32_t)order, (uint32_t)words.size(), bos_id, eos_id};
  std::vector<uint64_t> ends(words.size());
  uint64_t blob = 0;
  for (size_t i = 0; i < words.size(); ++i) ends[i] = (blob += words[i].size());
  const std::vector<NgramEntry>& ngram_table = ngr->table;
  const uint64_t h64[3] = {(uint64_t)n_ngrams, (uint64_t)ngram_table.size(), blob};
  ok = ok && fwrite(h32, 4, 4, f) == 4 && fwrite(h64, 8, 3, f) == 3;
  ok = ok && fwrite(ends.data(), 8, ends.size(), f) == ends.size();
  for (const std::string& w : words) ok = ok && (w.empty() || fwrite(w.data(), 1, w.size(), f) == w.size());
  ok = ok && fwrite(unigrams.data(), sizeof(UnigramEntry), unigrams.size(), f) == unigrams.size();
  ok = ok && fwrite(ngram_table.data(), sizeof(NgramEntry), ngram_table.size(), f) == ngram_table.size();
  ok = ok && fwrite(kCacheEnd, 1, 8, f) == 8;
  ok = (fclose(f) == 0) && ok;
  return ok ? "" : "short write to " + path;
}

std::string HostLM::load_cache(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return "cannot open LM file " + path;
  auto bad = [&](const std::string& why) {
    fclose(f);
    return "not a usable ctcdec model file (" + why + "): " + path;
  };
  char magic[8];
  uint32_t h32[4];
  uint64_t h64[3];
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, kCacheMagic, 8) != 0) return bad("magic");
  if (fread(h32, 4, 4, f) != 4 || fread(h64, 8, 3, f) != 3) return bad("header");
  const uint64_t nw = h32[1], tsize = h64[1], blob = h64[2];
  if (h32[0] < 1 || h32[0] > (uint32_t)(MAX_CTX + 1)) return bad("order");
  if (nw < 1 || nw > WI_ID_MASK || h32[2] >= nw || h32[3] >= nw) return bad("vocabulary");
  if (tsize < 16 || (tsize & (tsize - 1)) != 0 || h64[0] >= tsize) return bad("table size");
  std::vector<uint64_t> ends(nw);
  if (fread(ends.data(), 8, nw, f) != nw || ends[nw - 1] != blob) return bad("word offsets");
  std::string bytes(blob, '\0');
  if (blob && fread(&bytes[0], 1, blob, f) != blob) return bad("word bytes");
  std::vector<UnigramEntry> uni(nw);
  std::vector<NgramEntry> tab(tsize);
  if (fread(uni.data(), sizeof(UnigramEntry), nw, f) != nw || fread(tab.data(), sizeof(NgramEntry), tsize, f) != tsize)
    return bad("tables");
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, kCacheEnd, 8) != 0) return bad("truncated");
  fclose(f);
  words.assign(nw, std::string());
  vocab.clear();
  vocab.reserve(nw * 2);
  uint64_t a = 0;
  for (uint64_t i = 0; i < nw; ++i) {
    if (ends[i] < a || ends[i] > blob) return "corrupt word offsets in " + path;
    words[i].assign(bytes, a, ends[i] - a);
    a = ends[i];
    if (i > 0) vocab.emplace(words[i], (uint32_t)i);
  }
  vocab["<unk>"] = 0;
  order = (int)h32[0];
  bos_id = h32[2];
  eos_id = h32[3];
  n_ngrams = h64[0];
  unigrams.swap(uni);
  ngr = std::make_shared<NgramStore>();
  ngr->table.swap(tab);
  ngram_mask = tsize - 1;
  in_uniset.assign(words.size(), 0);
  has_trie = false;
  uniset_size = 0;
  build_prefix_table();
  return "";
}

uint32_t HostLM::index(const std::string& w) const {
  auto it = vocab.find(w);
  return it == vocab.end() ? 0u : it->second;
}

void HostLM::set_unigrams(bool has, const std::vector<std::string>& uni) {
  has_trie = has;
  in_uniset.assign(words.size(), 0);
  uniset_size = 0;
  if (has) {
    for (const std::string& w : uni) {
      uint32_t id = index(w);  // language_model.py:95: keep only words the LM knows
      if (id != 0 && !in_uniset[id]) {
        in_uniset[id] = 1;
        ++uniset_size;
      }
    }
  }
  build_prefix_table();
}

void HostLM::build_prefix_table() {
  std::unordered_map<uint64_t, PrefixEntry> m;
  m.reserve(words.size() * 6);
  std::vector<size_t> bounds;
  for (uint32_t id = 1; id < words.size(); ++id) {
    const std::string& w = words[id];
    utf8_boundaries(w.data(), w.size(), &bounds);
    uint64_t h = 0;
    size_t pos = 0;
    for (size_t b : bounds) {
      for (; pos < b; ++pos) h = str_push_byte(h, (unsigned char)w[pos]);
      PrefixEntry& e = m[h];
      e.key = h;
      if (in_uniset[id]) e.flags |= PF_UNI_PREFIX;
      if (b == w.size()) {
        e.flags |= PF_LM_WORD;
        e.word_id = id;
        if (in_uniset[id]) e.flags |= PF_UNI_WORD;
      }
    }
  }
  uint64_t size = 16;
  while (size < 4 * m.size() + 1) size <<= 1;  // load <= 1/4: most probes are misses, keep them at ~1.2 slots
  prefix_table.assign(size, PrefixEntry{0, 0, 0});
  prefix_mask = size - 1;
  for (auto& kv : m) {
    if (kv.first == 0) continue;
    uint64_t s = table_slot(kv.first) & prefix_mask;
    while (prefix_table[s].key != 0) s = (s + 1) & prefix_mask;
    prefix_table[s] = kv.second;
  }
}

void fill_token_starts_from(const std::vector<PrefixEntry>& table, uint64_t mask, HostAlphabet* alpha) {
  for (size_t i = 0; i < alpha->tok.size(); ++i) {
    TokInfo& t = alpha->tok[i];
    t.start_flags = 0;
    t.start_word_id = 0;
    uint32_t wid = 0, fl = 0;
    if (t.len_clean > 0 && !table.empty() && prefix_lookup(table.data(), mask, t.h_clean, &wid, &fl)) {
      t.start_flags = fl | PF_ON_TABLE;
      t.start_word_id = wid;
    }
  }
}

void HostLM::fill_token_starts(HostAlphabet* alpha) const { fill_token_starts_from(prefix_table, prefix_mask, alpha); }

void HostMulti::build() {
  const size_t K = lms.size();
  std::unordered_map<std::string, uint32_t> uni;
  words.assign(1, std::string());
  order = 0;
  for (const auto& lm : lms) {
    order = std::max(order, lm->order);
    for (uint32_t id = 1; id < lm->words.size(); ++id)
      if (uni.emplace(lm->words[id], (uint32_t)words.size()).second) words.push_back(lm->words[id]);
  }
  winfo.assign(K, std::vector<uint32_t>(words.size(), 0u));
  for (size_t k = 0; k < K; ++k) {
    const HostLM& lm = *lms[k];
    for (uint32_t id = 1; id < lm.words.size(); ++id)
      winfo[k][uni[lm.words[id]]] = id | WI_LM_WORD | (lm.in_uniset[id] ? WI_UNI_WORD : 0u);
  }
  std::unordered_map<uint64_t, PrefixEntry> m;
  m.reserve(words.size() * 6);
  std::vector<size_t> bounds;
  for (uint32_t u = 1; u < words.size(); ++u) {
    const std::string& w = words[u];
    uint32_t pbits = 0;  // models whose unigram set holds this word: their tries hold all its prefixes
    for (size_t k = 0; k < K; ++k)
      if (winfo[k][u] & WI_UNI_WORD) pbits |= k == 0 ? PF_UNI_PREFIX : (1u << (PF_X_SHIFT + (uint32_t)k));
    utf8_boundaries(w.data(), w.size(), &bounds);
    uint64_t h = 0;
    size_t pos = 0;
    for (size_t b : bounds) {
      for (; pos < b; ++pos) h = str_push_byte(h, (unsigned char)w[pos]);
      PrefixEntry& e = m[h];
      e.key = h;
      e.flags |= pbits;
      if (b == w.size()) {
        e.word_id = u;
        if (winfo[0][u] & WI_LM_WORD) e.flags |= PF_LM_WORD;
        if (winfo[0][u] & WI_UNI_WORD) e.flags |= PF_UNI_WORD;
      }
    }
  }
  uint64_t size = 16;
  while (size < 4 * m.size() + 1) size <<= 1;
  prefix_table.assign(size, PrefixEntry{0, 0, 0});
  prefix_mask = size - 1;
  for (auto& kv : m) {
    if (kv.first == 0) continue;
    uint64_t s = table_slot(kv.first) & prefix_mask;
    while (prefix_table[s].key != 0) s = (s + 1) & prefix_mask;
    prefix_table[s] = kv.second;
  }
}

void HostLM::start_state(bool begin_sentence, LmState* out) const {
  memset(out, 0, sizeof(*out));
  if (begin_sentence && order >= 2) {
    out->len = 1;
    out->words[0] = bos_id;
    out->backoff[0] = unigrams[bos_id].backoff;
  }
}

void HostLM::tables(DeviceTables* t) const {
  memset(t, 0, sizeof(*t));
  t->unigrams = unigrams.data();
  t->ngrams = ngr->table.data();
  t->ngram_mask = ngram_mask;
  t->prefixes = prefix_table.data();
  t->prefix_mask = prefix_mask;
  t->has_lm = 1;
  t->lm_order = (uint32_t)order;
  t->has_trie = has_trie ? 1u : 0u;
  t->uniset_nonempty = uniset_size > 0 ? 1u : 0u;
  t->eos_id = eos_id;
  t->n_hist = (uint32_t)std::max(1, order - 1);
}

void HostHotwords::build(const std::vector<std::string>& uni, const HostAlphabet& alpha) {
  std::unordered_map<uint64_t, HotEntry> m;
  std::vector<size_t> bounds;
  for (const std::string& w : uni) {
    if (w.empty()) continue;
    uint32_t wl = utf8_length(w.data(), w.size());
    if (wl > 0xFFFFu) wl = 0xFFFFu;  // 16 bits in a beam's flag word
    utf8_boundaries(w.data(), w.size(), &bounds);
    uint64_t h = 0;
    size_t pos = 0;
    for (size_t b : bounds) {
      for (; pos < b; ++pos) h = str_push_byte(h, (unsigned char)w[pos]);
      auto it = m.find(h);
      if (it == m.end()) {
        m.emplace(h, HotEntry{h, wl, b == w.size() ? 1u : 0u});
      } else {
        if (wl < it->second.min_len) it->second.min_len = wl;
        if (b == w.size()) it->second.complete = 1;
      }
    }
  }
  table.clear();
  mask = 0;
  if (!m.empty()) {
    uint64_t size = 16;
    while (size < 4 * m.size() + 1) size <<= 1;  // load <= 1/4: most probes are misses, keep them at ~1.2 slots
    table.assign(size, HotEntry{0, 0, 0});
    mask = size - 1;
    for (auto& kv : m) {
      if (kv.first == 0) continue;
      uint64_t s = table_slot(kv.first) & mask;
      while (table[s].key != 0) s = (s + 1) & mask;
      table[s] = kv.second;
    }
  }
  tok_hot.assign(alpha.tok.size(), TokHot{0, 0});
  if (!table.empty()) {
    for (size_t i = 0; i < alpha.tok.size(); ++i) {
      const TokInfo& t = alpha.tok[i];
      uint32_t ml = 0, cp = 0;
      if (t.len_clean > 0 && hot_lookup(table.data(), mask, t.h_clean, &ml, &cp)) tok_hot[i] = TokHot{ml, cp};
    }
  }
}

}  // namespace ctc
