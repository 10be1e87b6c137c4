// api.cpp -- the C ABI of include/ctcdec.h on top of backend.h + host_tables.h.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ctcdec.h"
#include "backend.h"
#include "beam_core.h"
#include "beam_wave.h"
#include "host_tables.h"

using namespace ctc;

static thread_local std::string g_err;
// The backend owns one stream and one set of timing events per process: calls that touch the device are
// serialised (several host threads may share decoders; the GIL is released during ctypes calls).
static std::mutex g_device_mu;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes, std::string* err) {
    if (bytes <= cap && p) return 0;
    if (p) be::release(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    p = be::alloc(want, err);
    if (!p) return -1;
    cap = want;
    return 0;
  }
  void drop() {
    if (p) be::release(p);
    p = nullptr;
    cap = 0;
  }
};

struct HostBuf {  // grow-only page-locked staging buffer
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes, std::string* err) {
    if (bytes <= cap && p) return 0;
    if (p) be::release_host(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    p = be::alloc_host(want, err);
    if (!p) return -1;
    cap = want;
    return 0;
  }
  void drop() {
    if (p) be::release_host(p);
    p = nullptr;
    cap = 0;
  }
};

template <class T>
int upload(DevBuf& b, const std::vector<T>& v, std::string* err) {
  size_t bytes = std::max<size_t>(sizeof(T) * v.size(), 16);
  if (b.ensure(bytes, err)) return -1;
  if (!v.empty() && be::h2d(b.p, v.data(), sizeof(T) * v.size(), err)) return -1;
  return 0;
}

struct BeamResult {
  std::string text;
  std::vector<int32_t> word_off, start, end;
  double logit = 0, lm = 0;
  ctcdec_lm_state state;
  std::vector<ctcdec_lm_state> xstates;  // MultiLanguageModel: states of model 1..
  // streaming extras (decoder.py:69-79 fields of the returned LMBeam)
  std::string partial;
  int32_t src = -1, last_char = -1, pstart = -1, pend = -1;
  double raw_lm = 0;
};

// what a streaming call adds to a plain batch decode
struct StreamIn {
  const int32_t* first_frame;
  const ctcdec_beam_in* beams;
  const int64_t* beam_off;
  const char* text_blob;
  int32_t fold, eos;
};

}  // namespace

// A few persistent host threads for the per-utterance replay (spawning them per call cost more than the
// work itself: 8 x ~80 us on the bench box). Jobs are index ranges handed out through an atomic cursor.
class ReplayPool {
 public:
  explicit ReplayPool(int n) {
    for (int i = 0; i < n; ++i) workers_.emplace_back([this] { loop(); });
  }
  ~ReplayPool() {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  int size() const { return (int)workers_.size(); }
  // fn(u0, u1) over [0, n) in chunks; returns when all chunks are done (the caller works too)
  void run(int32_t n, int32_t chunk, const std::function<void(int32_t, int32_t)>& fn) {
    {
      std::lock_guard<std::mutex> g(m_);
      fn_ = &fn;
      n_ = n;
      chunk_ = chunk;
      next_.store(0);
      pending_ = (int)workers_.size();
      ++epoch_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  void work() {
    for (;;) {
      int32_t u0 = next_.fetch_add(chunk_);
      if (u0 >= n_) return;
      (*fn_)(u0, std::min(n_, u0 + chunk_));
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return stop_ || epoch_ != seen; });
        if (stop_) return;
        seen = epoch_;
      }
      work();
      {
        std::lock_guard<std::mutex> g(m_);
        if (--pending_ == 0) done_.notify_one();
      }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int32_t, int32_t)>* fn_ = nullptr;
  std::atomic<int32_t> next_{0};
  int32_t n_ = 0, chunk_ = 1;
  int pending_ = 0;
  uint64_t epoch_ = 0;
  bool stop_ = false;
};

struct ctcdec_decoder {
  std::unique_ptr<ReplayPool> replay_pool;  // created on the first large batch
  HostAlphabet alpha;
  std::shared_ptr<HostLM> lm_ptr = std::make_shared<HostLM>();
  HostLM& lm_ref() { return *lm_ptr; }
  const HostLM& lm_ref() const { return *lm_ptr; }
  bool has_lm = false;
  // MultiLanguageModel: lm_ptr is model 0, multi holds all of them plus the union tables
  std::unique_ptr<HostMulti> multi;
  double x_alpha[MAX_LMS] = {0}, x_beta[MAX_LMS] = {0}, x_unk[MAX_LMS] = {0};
  int32_t x_boundary[MAX_LMS] = {0};
  int n_lms() const { return multi ? (int)multi->lms.size() : 1; }
  int hist_order() const { return multi ? multi->order : lm_ptr->order; }
  HostHotwords hot;
  bool tables_dirty = true, hot_dirty = true;
  DevBuf d_tok, d_tok_hot, d_uni, d_pref, d_hot;  // (the n-gram tables: NgramStore::device, shared between decoders)
  DevBuf d_xuni[MAX_LMS - 1], d_winfo[MAX_LMS], w_xstate, w_impx;
  HostBuf h_xstate;
  // per-call workspace (grow only)
  DevBuf w_logits, w_ptrs, w_row0, w_rowsum, w_isprob, w_scnt, w_sid, w_slp, w_flags, w_text, w_emit, w_toff,
      w_eoff, w_start, w_out, w_nout, w_status, w_tok, w_head, w_prof, w_imp, w_impoff, w_ff, w_cold, w_pay, w_tscr, w_tsoff, w_tpool,
      d_toktext, d_tokbytes, w_slow, w_order, w_side;
  bool slicing = false;  // a time-sliced host ingest is under way (decode_host_sliced): the prune stage notes each slice's side of 1
  uint32_t max_label_bytes = 1;
  bool arenas_worst_case = false;  // a call has outgrown the usual reservation of the node arenas: reserve the worst case from now on
  HostBuf h_tok, h_out, h_small;
  int dense_calls = 0;      // calls left that skip the 64-rows-per-wave prune kernel (most rows of a recent call overflowed it)
  HostBuf h_stage;          // page-locked staging of a call's small uploads (upload_staged): they go over without the host waiting
  size_t stage_used = 0;
  bool profile = false;
  unsigned long long prof[N_PROF] = {0};
  ~ctcdec_decoder() {
    DevBuf* all[] = {&d_tok,  &d_tok_hot, &d_uni,  &d_pref,  &d_hot,  &w_logits, &w_ptrs, &w_row0,
                     &w_rowsum, &w_isprob, &w_scnt, &w_sid,  &w_slp,   &w_flags, &w_text,   &w_emit, &w_toff,
                     &w_eoff,  &w_start,  &w_out,  &w_nout, &w_status, &w_tok,  &w_head, &w_prof, &w_imp, &w_impoff, &w_ff, &w_cold, &w_pay,
                     &w_tscr,  &w_tsoff, &w_tpool, &d_toktext, &d_tokbytes, &w_slow, &w_order, &w_side};
    for (DevBuf* b : all) b->drop();
    for (int k = 0; k < MAX_LMS - 1; ++k) {
      d_xuni[k].drop();
    }
    for (int k = 0; k < MAX_LMS; ++k) d_winfo[k].drop();
    w_xstate.drop();
    w_impx.drop();
    h_xstate.drop();
    h_tok.drop();
    h_out.drop();
    h_small.drop();
    h_stage.drop();
  }
};

// A call's small per-utterance tables (pointers, row offsets, arena offsets, start states): copied into the decoder's page-locked
// staging block and sent on the decode stream WITHOUT waiting -- the kernels that read them are queued behind them on the same
// stream. (upload() waits for every copy: six round trips of ~20 us in front of the first kernel of a call.) The block is reused
// from the start by the next call, which begins after this one's last synchronisation. Falls back to upload() when it is full.
template <class T>
static int upload_staged(ctcdec_decoder* dec, DevBuf& b, const std::vector<T>& v, std::string* err) {
  const size_t bytes = sizeof(T) * v.size();
  const size_t room = dec->h_stage.p ? dec->h_stage.cap - dec->stage_used : 0;
  if (bytes == 0 || bytes > room) return upload(b, v, err);
  if (b.ensure(std::max<size_t>(bytes, 16), err)) return -1;
  char* src = (char*)dec->h_stage.p + dec->stage_used;
  memcpy(src, v.data(), bytes);
  dec->stage_used += (bytes + 63) & ~(size_t)63;
  return be::h2d_async(b.p, src, bytes, err);
}

struct ctcdec_result {
  std::vector<std::vector<BeamResult>> utts;
  double ms[3] = {0, 0, 0};
  int beam_kernel = 0;  // be::last_beam_kernel() of the launch that produced this result
  // texts only (ctcdec_result_texts)
  bool texts_packed = false;
  std::string t_blob;
  std::vector<int64_t> t_off;
  std::string j_blob;  // ctcdec_result_texts_joined
  // params.texts_only: the texts as the device wrote them (one block per utterance somewhere in dev_texts) and the
  // output records; BeamResults are only built when an accessor other than ctcdec_result_texts_joined asks for them
  bool device_texts = false;
  std::string dev_texts;
  std::vector<OutBeam> dev_out;  // [n_utts]
  std::vector<int64_t> blk_off, blk_len;  // ctcdec_result_text_blocks
  // packed view (built on demand by ctcdec_result_pack)
  bool packed = false;
  std::vector<int64_t> beam_off, text_off, word_cnt_off;
  std::string text_blob;
  std::vector<double> logit, lm;
  std::vector<int32_t> word_byte_off, word_start, word_end;
  std::vector<ctcdec_lm_state> states;
  std::string partial_blob;
  std::vector<int64_t> partial_off;
  std::vector<int32_t> src_beam, last_char, pstart, pend;
  std::vector<double> raw_lm;
};

// A batch of device-resident streams (ctcdec_stream_*): what survives between chunks lives in device memory owned by
// the handle -- per stream a row of carried beams (the ImportBeam records the kernels' finalisation writes and the next
// chunk's import_beams() reads), the emission arena (grow-only: the chains reach back to the start of the stream) and
// the counters -- plus host mirrors of the counters, refreshed after every push.
struct ctcdec_stream {
  ctcdec_decoder* dec = nullptr;
  int32_t n = 0;
  int K = 1;
  static constexpr int CAP = CTCDEC_MAX_BEAM_WIDTH;  // carried beams per stream
  DevBuf carry, carry_x, sstate, emit, eoff;
  uint64_t emit_cap = 0;  // emission nodes per stream
  std::vector<StreamState> mirror;
  std::vector<int64_t> frames;  // frames pushed so far
  int64_t pushes = 0;           // chunks pushed since the streams' last start (each may close a word: one more chain entry)
  std::vector<ctcdec_lm_state> start_states;  // n * K, or empty: the models' defaults
  // the caller's beams of the last ctcdec_stream_import: roots (BR_IMPORT) of the chains decoded since
  bool has_import = false;
  std::vector<ctcdec_beam_in> imp_beams;
  std::vector<int64_t> imp_off;
  std::string imp_blob;
  ~ctcdec_stream() {
    carry.drop();
    carry_x.drop();
    sstate.drop();
    emit.drop();
    eoff.drop();
  }
};

// the device copy of a model's n-gram table: uploaded once per NgramStore, shared by every decoder that holds the model
// or a clone of it, released with the last of them
static const NgramEntry* device_ngrams(const HostLM& lm) {
  return lm.ngr->device ? (const NgramEntry*)static_cast<DevBuf*>(lm.ngr->device.get())->p : nullptr;
}
static int upload_ngrams(const HostLM& lm, std::string* err) {
  if (lm.ngr->device) return 0;
  std::shared_ptr<DevBuf> buf(new DevBuf(), [](DevBuf* b) {
    b->drop();
    delete b;
  });
  if (upload(*buf, lm.ngr->table, err)) return -1;
  lm.ngr->device = buf;
  return 0;
}

static int sync_tables(ctcdec_decoder* d, std::string* err) {
  if (d->tables_dirty) {
    if (d->multi) fill_token_starts_from(d->multi->prefix_table, d->multi->prefix_mask, &d->alpha);
    else if (d->has_lm) d->lm_ref().fill_token_starts(&d->alpha);
    if (upload(d->d_tok, d->alpha.tok, err)) return -1;
    {  // the labels' UTF-8 bytes, for the kernels that assemble texts themselves
      std::vector<TokText> tt(d->alpha.labels.size());
      std::string bytes;
      uint32_t longest = 1;
      for (size_t i = 0; i < tt.size(); ++i) {
        const std::string &raw = d->alpha.labels[i], &clean = d->alpha.clean[i];
        tt[i].raw_off = (uint32_t)bytes.size();
        tt[i].raw_len = (uint16_t)raw.size();
        bytes += raw;
        tt[i].clean_off = (uint32_t)bytes.size();
        tt[i].clean_len = (uint16_t)clean.size();
        bytes += clean;
        tt[i].pad = 0;
        longest = std::max<uint32_t>(longest, (uint32_t)std::max(raw.size(), clean.size()));
      }
      std::vector<uint8_t> bv(bytes.begin(), bytes.end());
      if (bv.empty()) bv.push_back(0);
      if (upload(d->d_toktext, tt, err) || upload(d->d_tokbytes, bv, err)) return -1;
      d->max_label_bytes = longest;
    }
    if (d->has_lm) {
      if (upload(d->d_uni, d->lm_ref().unigrams, err)) return -1;
      if (upload_ngrams(d->lm_ref(), err)) return -1;
      if (upload(d->d_pref, d->multi ? d->multi->prefix_table : d->lm_ref().prefix_table, err)) return -1;
    }
    if (d->multi) {
      for (int k = 0; k < d->n_lms(); ++k) {
        if (upload(d->d_winfo[k], d->multi->winfo[(size_t)k], err)) return -1;
        if (k > 0 && (upload(d->d_xuni[k - 1], d->multi->lms[(size_t)k]->unigrams, err) ||
                      upload_ngrams(*d->multi->lms[(size_t)k], err)))
          return -1;
      }
    }
    d->tables_dirty = false;
    d->hot_dirty = true;
  }
  if (d->hot_dirty) {
    if (d->hot.tok_hot.size() != d->alpha.tok.size()) d->hot.build({}, d->alpha);
    if (upload(d->d_tok_hot, d->hot.tok_hot, err)) return -1;
    if (upload(d->d_hot, d->hot.table, err)) return -1;
    d->hot_dirty = false;
  }
  return 0;
}

static void device_tables(const ctcdec_decoder* d, DeviceTables* t) {
  memset(t, 0, sizeof(*t));
  if (d->has_lm) d->lm_ref().tables(t);
  t->tok = (const TokInfo*)d->d_tok.p;
  t->tok_hot = (const TokHot*)d->d_tok_hot.p;
  t->tok_text = (const TokText*)d->d_toktext.p;
  t->tok_bytes = (const uint8_t*)d->d_tokbytes.p;
  t->max_label_bytes = d->max_label_bytes;
  if (d->has_lm) {
    t->unigrams = (const UnigramEntry*)d->d_uni.p;
    t->ngrams = device_ngrams(d->lm_ref());
    t->prefixes = (const PrefixEntry*)d->d_pref.p;
    if (d->multi) {
      t->prefix_mask = d->multi->prefix_mask;
      t->n_lms = (uint32_t)d->n_lms();
      t->n_hist = (uint32_t)std::max(1, d->multi->order - 1);
      t->winfo0 = (const uint32_t*)d->d_winfo[0].p;
      for (int k = 1; k < d->n_lms(); ++k) {
        const HostLM& lm = *d->multi->lms[(size_t)k];
        LmExtra& x = t->x[k - 1];
        x.unigrams = (const UnigramEntry*)d->d_xuni[k - 1].p;
        x.ngrams = device_ngrams(*d->multi->lms[(size_t)k]);
        x.ngram_mask = lm.ngram_mask;
        x.winfo = (const uint32_t*)d->d_winfo[k].p;
        x.lm_order = (uint32_t)lm.order;
        x.has_trie = lm.has_trie ? 1u : 0u;
        x.uniset_nonempty = lm.uniset_size > 0 ? 1u : 0u;
        x.eos_id = lm.eos_id;
        x.alpha = d->x_alpha[k];
        x.beta = d->x_beta[k];
        x.unk = d->x_unk[k];
        x.score_boundary = d->x_boundary[k];
      }
    }
  } else {
    t->n_hist = 1;  // lm_order 1 without an LM (decoder.py:551)
  }
  t->hot = d->hot.table.empty() ? nullptr : (const HotEntry*)d->d_hot.p;
  t->hot_mask = d->hot.mask;
  t->n_labels = (uint32_t)d->alpha.labels.size();
  t->is_bpe = d->alpha.is_bpe ? 1u : 0u;
}

extern "C" {

const char* ctcdec_last_error(void) { return g_err.c_str(); }
const char* ctcdec_version(void) { return "ctcdec 0.1 (gfx950)"; }

int ctcdec_create(const char* labels_blob, const int64_t* labels_off, int32_t n_labels, int32_t is_bpe,
                  int32_t device, ctcdec_decoder** out) {
  if (!labels_blob || !labels_off || !out || n_labels <= 0) return fail(CTCDEC_ERR_ARG, "bad arguments");
  if (n_labels > CTCDEC_MAX_VOCAB) return fail(CTCDEC_ERR_LIMIT, "vocabulary larger than 65535 labels");
  std::string err;
  if (be::init(device, &err)) return fail(CTCDEC_ERR_DEVICE, err);
  std::unique_ptr<ctcdec_decoder> d(new ctcdec_decoder());
  std::vector<std::string> labels;
  for (int32_t i = 0; i < n_labels; ++i)
    labels.emplace_back(labels_blob + labels_off[i], (size_t)(labels_off[i + 1] - labels_off[i]));
  d->alpha.build(labels, is_bpe != 0);
  d->hot.build({}, d->alpha);
  *out = d.release();
  return CTCDEC_OK;
}

void ctcdec_destroy(ctcdec_decoder* dec) { delete dec; }

int ctcdec_lm_load_arpa(ctcdec_decoder* dec, const char* path, int32_t* order_out) {
  if (!dec || !path) return fail(CTCDEC_ERR_ARG, "bad arguments");
  std::string e = dec->lm_ref().load_arpa(path);
  if (!e.empty()) return fail(CTCDEC_ERR_IO, e);
  dec->has_lm = true;
  dec->tables_dirty = true;
  if (order_out) *order_out = dec->lm_ref().order;
  return CTCDEC_OK;
}

int ctcdec_lm_save_flat(const ctcdec_decoder* dec, const char* path) {
  if (!dec || !path || !dec->has_lm) return fail(CTCDEC_ERR_ARG, "no language model loaded");
  std::string e = dec->lm_ref().save_cache(path);
  if (!e.empty()) return fail(CTCDEC_ERR_IO, e);
  return CTCDEC_OK;
}

int ctcdec_lm_load_flat(ctcdec_decoder* dec, const char* path, int32_t* order_out) {
  if (!dec || !path) return fail(CTCDEC_ERR_ARG, "bad arguments");
  if (dec->multi) return fail(CTCDEC_ERR_ARG, "decoder holds several language models");
  std::string e = dec->lm_ref().load_cache(path);
  if (!e.empty()) return fail(CTCDEC_ERR_IO, e);
  dec->has_lm = true;
  dec->tables_dirty = true;
  if (order_out) *order_out = dec->lm_ref().order;
  return CTCDEC_OK;
}

int ctcdec_lm_load_kenlm(ctcdec_decoder* dec, const char* path, int32_t* order_out) {
  if (!dec || !path) return fail(CTCDEC_ERR_ARG, "bad arguments");
  if (dec->multi) return fail(CTCDEC_ERR_ARG, "decoder holds several language models");
  std::string e = dec->lm_ref().load_kenlm_binary(path);
  if (!e.empty()) return fail(CTCDEC_ERR_IO, e);
  dec->has_lm = true;
  dec->tables_dirty = true;
  if (order_out) *order_out = dec->lm_ref().order;
  return CTCDEC_OK;
}

int ctcdec_is_kenlm_binary(const char* path) { return path && looks_like_kenlm_binary(path) ? 1 : 0; }

int ctcdec_arpa_to_kenlm_binary(const char* arpa_path, const char* out_path, float probing_multiplier) {
  if (!arpa_path || !out_path) return fail(CTCDEC_ERR_ARG, "bad arguments");
  std::string e = arpa_to_kenlm_binary(arpa_path, out_path, probing_multiplier);
  if (!e.empty()) return fail(CTCDEC_ERR_IO, e);
  return CTCDEC_OK;
}

int ctcdec_lm_set_unigrams(ctcdec_decoder* dec, int32_t has_unigrams, const char* blob, const int64_t* off,
                           int64_t n_unigrams, int64_t* n_kept_out) {
  if (!dec || !dec->has_lm) return fail(CTCDEC_ERR_ARG, "no language model loaded");
  std::vector<std::string> uni;
  if (has_unigrams)
    for (int64_t i = 0; i < n_unigrams; ++i) uni.emplace_back(blob + off[i], (size_t)(off[i + 1] - off[i]));
  dec->lm_ref().set_unigrams(has_unigrams != 0, uni);
  dec->tables_dirty = true;
  if (n_kept_out) *n_kept_out = (int64_t)dec->lm_ref().uniset_size;
  return CTCDEC_OK;
}

int ctcdec_lm_share(ctcdec_decoder* dst, const ctcdec_decoder* src) {
  if (!dst || !src || !src->has_lm) return fail(CTCDEC_ERR_ARG, "source has no language model");
  if (src->multi) return fail(CTCDEC_ERR_ARG, "source holds several language models");
  dst->lm_ptr = src->lm_ptr;
  dst->multi.reset();
  dst->has_lm = true;
  dst->tables_dirty = true;
  return CTCDEC_OK;
}

int ctcdec_lm_clone(ctcdec_decoder* dst, const ctcdec_decoder* src) {
  if (!dst || !src || !src->has_lm) return fail(CTCDEC_ERR_ARG, "source has no language model");
  if (src->multi) return fail(CTCDEC_ERR_ARG, "source holds several language models");
  dst->lm_ptr = std::make_shared<HostLM>(*src->lm_ptr);  // own tables: own unigram set, own prefix flags
  dst->multi.reset();
  dst->has_lm = true;
  dst->tables_dirty = true;
  return CTCDEC_OK;
}

int ctcdec_lm_share_multi(ctcdec_decoder* dst, const ctcdec_decoder* const* srcs, int32_t n) {
  if (!dst || !srcs || n < 2) return fail(CTCDEC_ERR_ARG, "a MultiLanguageModel holds at least 2 language models");
  if (n > CTCDEC_MAX_LMS) return fail(CTCDEC_ERR_LIMIT, "more language models than CTCDEC_MAX_LMS");
  std::unique_ptr<HostMulti> m(new HostMulti());
  for (int32_t k = 0; k < n; ++k) {
    if (!srcs[k] || !srcs[k]->has_lm || srcs[k]->multi) return fail(CTCDEC_ERR_ARG, "source has no (single) language model");
    m->lms.push_back(srcs[k]->lm_ptr);
  }
  m->build();
  if (m->words.size() > WI_ID_MASK) return fail(CTCDEC_ERR_LIMIT, "union vocabulary too large");
  dst->lm_ptr = m->lms[0];
  dst->multi = std::move(m);
  dst->has_lm = true;
  dst->tables_dirty = true;
  return CTCDEC_OK;
}

int ctcdec_lm_set_params(ctcdec_decoder* dec, int32_t k, double alpha, double beta, double unk_score_offset,
                         int32_t lm_score_boundary) {
  if (!dec || !dec->multi || k < 1 || k >= dec->n_lms()) return fail(CTCDEC_ERR_ARG, "no such additional language model");
  dec->x_alpha[k] = alpha;
  dec->x_beta[k] = beta;
  dec->x_unk[k] = unk_score_offset;
  dec->x_boundary[k] = lm_score_boundary ? 1 : 0;
  return CTCDEC_OK;
}

int ctcdec_lm_count(const ctcdec_decoder* dec, int32_t* n_out) {
  if (!dec || !n_out) return fail(CTCDEC_ERR_ARG, "bad arguments");
  *n_out = dec->has_lm ? dec->n_lms() : 0;
  return CTCDEC_OK;
}

int ctcdec_lm_prefix_flags(const ctcdec_decoder* dec, const char* s, int64_t len, uint32_t* flags_out) {
  if (!dec || !dec->has_lm || !flags_out) return fail(CTCDEC_ERR_ARG, "no language model loaded");
  uint32_t wid = 0, fl = 0;
  const HostLM& lm = dec->lm_ref();
  *flags_out = 0;
  if (len > 0 && prefix_lookup(lm.prefix_table.data(), lm.prefix_mask, hash_bytes(s, (size_t)len), &wid, &fl))
    *flags_out = fl;
  return CTCDEC_OK;
}

int ctcdec_lm_word_index(const ctcdec_decoder* dec, const char* w, int64_t len, uint32_t* index_out) {
  if (!dec || !dec->has_lm || !index_out) return fail(CTCDEC_ERR_ARG, "no language model loaded");
  *index_out = dec->lm_ref().index(std::string(w, (size_t)len));
  return CTCDEC_OK;
}

int ctcdec_lm_word_string(const ctcdec_decoder* dec, uint32_t index, const char** str_out, int64_t* len_out) {
  if (!dec || !dec->has_lm || index >= dec->lm_ref().words.size()) return fail(CTCDEC_ERR_ARG, "bad word index");
  *str_out = dec->lm_ref().words[index].data();
  *len_out = (int64_t)dec->lm_ref().words[index].size();
  return CTCDEC_OK;
}

int ctcdec_lm_start_state(const ctcdec_decoder* dec, int32_t begin_sentence, ctcdec_lm_state* out) {
  if (!dec || !dec->has_lm || !out) return fail(CTCDEC_ERR_ARG, "no language model loaded");
  LmState st;
  dec->lm_ref().start_state(begin_sentence != 0, &st);
  out->length = st.len;
  for (int k = 0; k < MAX_CTX; ++k) {
    out->words[k] = st.words[k];
    out->backoff[k] = st.backoff[k];
  }
  return CTCDEC_OK;
}

int ctcdec_lm_base_score(const ctcdec_decoder* dec, const ctcdec_lm_state* in, uint32_t word_index,
                         ctcdec_lm_state* out, float* log10_prob_out) {
  if (!dec || !dec->has_lm || !in || !out || !log10_prob_out) return fail(CTCDEC_ERR_ARG, "bad arguments");
  if (word_index >= dec->lm_ref().words.size() || in->length < 0 || in->length > MAX_CTX)
    return fail(CTCDEC_ERR_ARG, "bad LM state or word index");
  DeviceTables t;
  dec->lm_ref().tables(&t);
  LmState a, b;
  a.len = in->length;
  for (int k = 0; k < MAX_CTX; ++k) {
    a.words[k] = in->words[k];
    a.backoff[k] = in->backoff[k];
  }
  *log10_prob_out = lm_base_score(t, a, word_index, &b);
  out->length = b.len;
  for (int k = 0; k < MAX_CTX; ++k) {
    out->words[k] = b.words[k];
    out->backoff[k] = b.backoff[k];
  }
  return CTCDEC_OK;
}

int ctcdec_set_hotwords(ctcdec_decoder* dec, const char* blob, const int64_t* off, int64_t n_words) {
  if (!dec) return fail(CTCDEC_ERR_ARG, "bad arguments");
  std::string err;
  std::lock_guard<std::mutex> device_lock(g_device_mu);
  if (be::bind_thread(&err)) return fail(CTCDEC_ERR_DEVICE, err);
  std::vector<std::string> uni;
  for (int64_t i = 0; i < n_words; ++i) uni.emplace_back(blob + off[i], (size_t)(off[i + 1] - off[i]));
  dec->hot.build(uni, dec->alpha);
  dec->hot_dirty = true;
  return CTCDEC_OK;
}

// Rebuild text + word frames of one beam from its emission list (root -> leaf). The text is written
// in place: `open` is where the currently open (partial) word starts. A streaming beam starts from the
// caller's beam named by its BR_IMPORT root (text so far + open partial word).
static void replay(const ctcdec_decoder* d, const EmitNode* toks, uint32_t n, const StreamIn* st, int64_t imp0,
                   BeamResult* r) {
  std::string& text = r->text;
  text.clear();
  text.reserve((size_t)n * 3 + 8);
  size_t open = 0;
  auto close_word = [&](int32_t s, int32_t e, bool more) {
    if (text.size() == open) return;  // empty open word: nothing to close
    r->word_off.push_back((int32_t)open);
    r->start.push_back(s);
    r->end.push_back(e);
    if (more) text.push_back(' ');
    open = text.size();
  };
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t br = toks[k].tok_branch >> 16, tok = toks[k].tok_branch & 0xFFFFu;
    if (br == BR_BOUNDARY) {
      close_word(toks[k].wstart, toks[k].wend, true);
      text += d->alpha.clean[tok];
    } else if (br == BR_SPACE) {
      close_word(toks[k].wstart, toks[k].wend, true);
    } else if (br == BR_APPEND) {
      text += d->alpha.labels[tok];
    } else if (br == BR_FINAL) {
      // (the last entry of a folded beam, or -- resident streams after force_next_word -- an entry in the middle of
      // the chain: the trailing separator of the former is dropped below)
      close_word(toks[k].wstart, toks[k].wend, true);
    } else if (br == BR_IMPORT && st) {
      const ctcdec_beam_in& in = st->beams[imp0 + tok];
      r->src = (int32_t)tok;
      text.assign(st->text_blob + in.text_begin, (size_t)(in.text_end - in.text_begin));
      if (!text.empty()) text.push_back(' ');
      open = text.size();
      text.append(st->text_blob + in.partial_begin, (size_t)(in.partial_end - in.partial_begin));
    }
  }
  // split off the still open word (streaming without force_next_word / is_end)
  r->partial.assign(text, open, std::string::npos);
  text.resize(open);
  if (!text.empty() && text.back() == ' ') text.pop_back();
  r->word_off.push_back((int32_t)text.size());
}

// OutBeam record (+ the states of the further language models) -> the host-side beam; the text follows by replay()
static void fill_result(const OutBeam& ob, const LmState* xs, int K, BeamResult* r) {
  r->logit = ob.logit_score;
  r->lm = ob.lm_score;
  r->state.length = ob.state.len;
  for (int j = 0; j < MAX_CTX; ++j) {
    r->state.words[j] = ob.state.words[j];
    r->state.backoff[j] = ob.state.backoff[j];
  }
  if (xs) {
    r->xstates.resize((size_t)(K - 1));
    for (int x = 0; x < K - 1; ++x) {
      r->xstates[(size_t)x].length = xs[x].len;
      for (int j = 0; j < MAX_CTX; ++j) {
        r->xstates[(size_t)x].words[j] = xs[x].words[j];
        r->xstates[(size_t)x].backoff[j] = xs[x].backoff[j];
      }
    }
  }
  r->last_char = ob.last_char == NO_CHAR ? -1 : (int32_t)ob.last_char;
  r->pstart = ob.pstart;
  r->pend = ob.pend;
  r->raw_lm = ob.raw_lm;
}

// after_launch: called once the kernels of this call are queued and before the host waits for them (time-sliced host ingest:
// the next slice's copy runs under this slice's kernels)
typedef std::function<int(std::string*)> AfterLaunch;
static int decode_impl(ctcdec_decoder* dec, const void* const* utt_logits, const int32_t* utt_frames, int32_t n_utts,
                       int32_t dtype, int32_t is_device, const ctcdec_params* p, const ctcdec_lm_state* start_states,
                       const StreamIn* stream, ctcdec_result** out, ctcdec_stream* rs = nullptr, bool want_result = true,
                       const AfterLaunch* after_launch = nullptr);
static int decode_host_sliced(ctcdec_decoder* dec, const void* const* utt_logits, const int32_t* utt_frames, int32_t n_utts,
                              int32_t dtype, const ctcdec_params* p, const ctcdec_lm_state* start_states, int n_slices,
                              ctcdec_result** out);
static int host_slices_wanted(const ctcdec_decoder* dec, const int32_t* utt_frames, int32_t n_utts, int32_t dtype);

int ctcdec_decode_batch(ctcdec_decoder* dec, const void* const* utt_logits, const int32_t* utt_frames,
                        int32_t n_utts, int32_t dtype, int32_t is_device, const ctcdec_params* p,
                        const ctcdec_lm_state* start_states, ctcdec_result** out) {
  // Large HOST batches (the reference's own calling convention: numpy in) go over in time slices, the copy of slice k + 1
  // under the kernels of slice k (decode_host_sliced); 1 = "decode it in one piece after all" (probability-like rows)
  if (dec && p && out && !is_device && n_utts > 0 && utt_logits && utt_frames && dtype >= CTCDEC_F32 && dtype <= CTCDEC_BF16) {
    const int n_slices = host_slices_wanted(dec, utt_frames, n_utts, dtype);
    if (n_slices >= 2) {
      const int rc = decode_host_sliced(dec, utt_logits, utt_frames, n_utts, dtype, p, start_states, n_slices, out);
      if (rc != 1) return rc;
    }
  }
  return decode_impl(dec, utt_logits, utt_frames, n_utts, dtype, is_device, p, start_states, nullptr, out);
}

int ctcdec_decode_stream_batch(ctcdec_decoder* dec, const void* const* utt_logits, const int32_t* utt_frames,
                               int32_t n_streams, int32_t dtype, int32_t is_device, const ctcdec_params* p,
                               const int32_t* first_frame, const ctcdec_beam_in* beams, const int64_t* beam_off,
                               const char* text_blob, int32_t force_next_word, int32_t is_end, ctcdec_result** out) {
  if (!first_frame || !beams || !beam_off || !text_blob) return fail(CTCDEC_ERR_ARG, "bad arguments");
  StreamIn st;
  st.first_frame = first_frame;
  st.beams = beams;
  st.beam_off = beam_off;
  st.text_blob = text_blob;
  st.fold = (force_next_word || is_end) ? 1 : 0;
  st.eos = is_end ? 1 : 0;
  return decode_impl(dec, utt_logits, utt_frames, n_streams, dtype, is_device, p, nullptr, &st, out);
}

// host side of a streaming import: strings -> hashes, table views, history ring (the kernel rebuilds
// the beam row and its TextNode from this)
static int64_t shape_bw_limit(int beam_width) { return beam_bucket(beam_width); }

static std::string build_import(const ctcdec_decoder* dec, const StreamIn& st, int64_t k, int beam_width, ImportBeam* m,
                                LmState* more /* n_lms - 1 entries, or nullptr */) {
  const ctcdec_beam_in& in = st.beams[k];
  memset(m, 0, sizeof(*m));
  if (in.text_end < in.text_begin || in.partial_end < in.partial_begin) return "bad beam text range";
  const char* t = st.text_blob + in.text_begin;
  const size_t tn = (size_t)(in.text_end - in.text_begin);
  const uint32_t n_hist = dec->has_lm ? (uint32_t)std::max(1, dec->hist_order() - 1) : 1u;
  uint64_t th = 0;
  std::vector<uint64_t> wh;
  uint32_t hw = 0;
  size_t a = 0;
  while (a < tn) {
    while (a < tn && t[a] == ' ') ++a;
    size_t b = a;
    while (b < tn && t[b] != ' ') ++b;
    if (b > a) {
      uint64_t h = hash_bytes(t + a, b - a);
      th = text_push(th, h);
      wh.push_back(h);
      uint32_t ml = 0, cp = 0;
      if (!dec->hot.table.empty() && hot_lookup(dec->hot.table.data(), dec->hot.mask, h, &ml, &cp) && cp) ++hw;
    }
    a = b;
  }
  m->text_h = th;
  m->hw_cnt = hw;
  m->ring_cnt = (uint32_t)std::min<size_t>(n_hist, wh.size());
  for (uint32_t j = 0; j < m->ring_cnt; ++j) m->ring[j] = wh[wh.size() - 1 - j];
  const char* pp = st.text_blob + in.partial_begin;
  const size_t pn = (size_t)(in.partial_end - in.partial_begin);
  m->part_h = hash_bytes(pp, pn);
  m->plen = utf8_length(pp, pn);
  if (m->plen > 0xFFFF) return "partial word too long";
  uint32_t m2 = 0, wid = 0;
  if (pn > 0) {
    uint32_t fl = 0, w = 0;
    const std::vector<PrefixEntry>& ptab = dec->multi ? dec->multi->prefix_table : dec->lm_ref().prefix_table;
    const uint64_t pmask = dec->multi ? dec->multi->prefix_mask : dec->lm_ref().prefix_mask;
    if (dec->has_lm && prefix_lookup(ptab.data(), pmask, m->part_h, &w, &fl)) {
      m2 |= PF_ON_TABLE | (fl & PF_PARTIAL_MASK);
      wid = w;
    }
    uint32_t ml = 0, cp = 0;
    if (!dec->hot.table.empty() && hot_lookup(dec->hot.table.data(), dec->hot.mask, m->part_h, &ml, &cp))
      m2 |= M2_HOT_ON | (cp ? M2_HOT_COMPLETE : 0u) | ((ml & 0xFFFFu) << 8);
  }
  m->m2 = m2;
  m->word_id = wid;
  if (in.last_char >= (int32_t)dec->alpha.labels.size()) return "last_char out of range";
  m->last_char = in.last_char < 0 ? NO_CHAR : (uint32_t)in.last_char;
  m->pstart = in.partial_start;
  m->pend = in.partial_end_frame;
  m->logit_score = in.logit_score;
  m->raw_lm = dec->has_lm ? in.raw_lm_score : 0.0;
  if (dec->has_lm) {
    if (in.lm_state.length < 0 || in.lm_state.length > MAX_CTX) return "bad LM state in beam";
    m->state.len = in.lm_state.length;
    for (int j = 0; j < m->state.len; ++j) {
      if (in.lm_state.words[j] >= dec->lm_ref().words.size()) return "bad LM state word in beam";
      m->state.words[j] = in.lm_state.words[j];
      m->state.backoff[j] = in.lm_state.backoff[j];
    }
  }
  if (dec->multi) {
    if (!in.more_states) return "beam lacks the states of the further language models";
    for (int x = 1; x < dec->n_lms(); ++x) {
      const ctcdec_lm_state& g = in.more_states[x - 1];
      LmState& o = more[x - 1];
      memset(&o, 0, sizeof(o));
      if (g.length < 0 || g.length > MAX_CTX) return "bad LM state in beam";
      o.len = g.length;
      for (int j = 0; j < o.len; ++j) {
        if (g.words[j] >= dec->multi->lms[(size_t)x]->words.size()) return "bad LM state word in beam";
        o.words[j] = g.words[j];
        o.backoff[j] = g.backoff[j];
      }
    }
  }
  (void)beam_width;
  return "";
}

// rs: the streams are device-resident (`stream` then only carries first_frame / fold / eos and, below an import, the
// caller's beams for the replay); want_result: materialise beams at all
static int decode_impl(ctcdec_decoder* dec, const void* const* utt_logits, const int32_t* utt_frames, int32_t n_utts,
                       int32_t dtype, int32_t is_device, const ctcdec_params* p, const ctcdec_lm_state* start_states,
                       const StreamIn* stream, ctcdec_result** out, ctcdec_stream* rs, bool want_result,
                       const AfterLaunch* after_launch) {
  if (!dec || !p || !out || n_utts < 0 || (n_utts > 0 && (!utt_logits || !utt_frames)))
    return fail(CTCDEC_ERR_ARG, "bad arguments");
  if (dtype < CTCDEC_F32 || dtype > CTCDEC_BF16) return fail(CTCDEC_ERR_ARG, "dtype must be f32, f64, f16 or bf16");
  if (p->beam_width < 1) return fail(CTCDEC_ERR_ARG, "beam_width must be >= 1");
  if (p->beam_width > CTCDEC_MAX_BEAM_WIDTH)
    return fail(CTCDEC_ERR_LIMIT, "beam_width above the supported maximum of 256");
  std::string err;
  auto t_begin = std::chrono::steady_clock::now();
  std::unique_ptr<ctcdec_result> res(new ctcdec_result());
  res->utts.resize((size_t)n_utts);
  if (n_utts == 0) {
    *out = res.release();
    return CTCDEC_OK;
  }
  const int K = dec->has_lm ? dec->n_lms() : 1;
  std::lock_guard<std::mutex> device_lock(g_device_mu);
  if (be::bind_thread(&err)) return fail(CTCDEC_ERR_DEVICE, err);
  if (sync_tables(dec, &err)) return fail(CTCDEC_ERR_DEVICE, err);
  const int V = (int)dec->alpha.labels.size();
  const size_t esz = dtype == CTCDEC_F32 ? 4 : dtype == CTCDEC_F64 ? 8 : 2;
  // staging for this call's small uploads (every earlier call has synchronised: nothing is in flight from the block)
  dec->stage_used = 0;
  if (dec->h_stage.ensure((size_t)n_utts * (64 + sizeof(LmState) * (size_t)K) + 4096, &err)) return fail(CTCDEC_ERR_DEVICE, err);

  std::vector<int64_t> row0((size_t)n_utts + 1, 0);
  for (int32_t u = 0; u < n_utts; ++u) {
    if (utt_frames[u] < 0) return fail(CTCDEC_ERR_ARG, "negative frame count");
    row0[(size_t)u + 1] = row0[(size_t)u] + utt_frames[u];
  }
  const int64_t R = row0[(size_t)n_utts];

  // logits: device pointers are used in place, host matrices are staged
  std::vector<const void*> ptrs((size_t)n_utts);
  if (is_device) {
    for (int32_t u = 0; u < n_utts; ++u) ptrs[(size_t)u] = utt_logits[u];
  } else {
    if (dec->w_logits.ensure((size_t)std::max<int64_t>(R, 1) * V * esz, &err)) return fail(CTCDEC_ERR_DEVICE, err);
    // (utterances that follow each other in host memory -- one [B, T, V] array -- go over in one copy)
    for (int32_t u = 0; u < n_utts;) {
      char* dst = (char*)dec->w_logits.p + (size_t)row0[(size_t)u] * V * esz;
      const char* src = (const char*)utt_logits[u];
      size_t bytes = (size_t)utt_frames[u] * V * esz;
      ptrs[(size_t)u] = dst;
      int32_t v = u + 1;
      while (v < n_utts && (const char*)utt_logits[v] == src + bytes) {
        ptrs[(size_t)v] = dst + bytes;
        bytes += (size_t)utt_frames[v] * V * esz;
        ++v;
      }
      if (bytes && be::h2d(dst, src, bytes, &err)) return fail(CTCDEC_ERR_DEVICE, err);
      u = v;
    }
  }
  if (upload_staged(dec, dec->w_ptrs, ptrs, &err) || upload_staged(dec, dec->w_row0, row0, &err)) return fail(CTCDEC_ERR_DEVICE, err);

  // survivor bound: rows are normalised log-probabilities, so at most floor(e^-min) labels pass
  int max_surv = V;
  if (p->token_min_logp > log(1e-15)) {  // frames are clipped at ln(MIN_TOKEN_CLIP_P) (constants.py:17)
    double bound = floor(exp(-p->token_min_logp)) + 2.0;
    if (bound < (double)V) max_surv = (int)bound;
  }
  if (max_surv < 1) max_surv = 1;

  const int B = p->beam_width;

  // Arenas of text nodes (one per completed-words prefix that is scored) and emission nodes (one per non-blank,
  // non-repeat step of a kept beam). The worst case is beam_width of each per frame; what a frame really takes is a
  // handful, so the usual reservation is 16 per frame (+ 2 beam_widths): an utterance that
  // outgrows it reports ST_TEXT_OVERFLOW / ST_EMIT_OVERFLOW and the beam stage is redone with the worst case, which
  // this decoder then keeps reserving. A resident stream's kernel cannot be redone: always the worst case (per chunk).
  std::vector<uint64_t> toff((size_t)n_utts + 1, 0), eoff((size_t)n_utts + 1, 0);
  bool arenas_full = rs != nullptr || dec->arenas_worst_case || getenv("CTCDEC_WORST_CASE_ARENAS") != nullptr;
  auto size_arenas = [&](bool full) {
    const uint64_t per_frame = full ? (uint64_t)B : std::min<uint64_t>((uint64_t)B, 16);
    for (int32_t u = 0; u < n_utts; ++u) {
      uint64_t T = (uint64_t)utt_frames[u];
      uint64_t n_imp = rs ? (uint64_t)rs->mirror[(size_t)u].n_carry
                          : stream ? (uint64_t)(stream->beam_off[u + 1] - stream->beam_off[u]) : 0;
      toff[(size_t)u + 1] = toff[(size_t)u] + ((T + 1) * per_frame + 2 * (uint64_t)B + 2 + n_imp) * (uint64_t)K;
      eoff[(size_t)u + 1] = eoff[(size_t)u] + T * per_frame + 2 * (uint64_t)B + 2 + n_imp;
    }
  };
  size_arenas(arenas_full);
  int n_best = p->n_best > 0 ? std::min(p->n_best, B) : B;
  // emission lists: at most one entry per frame plus the import root and the closing entry
  unsigned long long tok_cap = (unsigned long long)n_best * (unsigned long long)(R + 2 * (int64_t)n_utts);
  if (rs) {
    // A resident stream's lists reach back to its start, but a beam's chain holds at most one entry per frame pushed so
    // far, one per chunk (a word closed by force_next_word) and its root -- NOT the stream's whole emission arena (every
    // beam's nodes: ten minutes of audio on 64 streams would ask for gigabytes here; round-3 advisor finding).
    unsigned long long depth = 0;
    for (int32_t u = 0; u < n_utts; ++u)
      depth += (unsigned long long)(rs->frames[(size_t)u] + utt_frames[u] + rs->pushes + 3);
    tok_cap = want_result ? (unsigned long long)n_best * depth : 1ull;
  }
  // streaming: carried-over beams of every stream
  const ImportBeam* d_imports = nullptr;
  const LmState* d_import_x = nullptr;
  int32_t max_import = 0;
  if (rs) {
    for (int32_t u = 0; u < n_utts; ++u) max_import = std::max<int32_t>(max_import, (int32_t)rs->mirror[(size_t)u].n_carry);
    std::vector<int32_t> ff(stream->first_frame, stream->first_frame + n_utts);
    if (upload(dec->w_ff, ff, &err)) return fail(CTCDEC_ERR_DEVICE, err);
    // the emission arena of a resident stream is its own and reaches back to the start of the stream: room for this chunk
    uint64_t need = 0;
    for (int32_t u = 0; u < n_utts; ++u)
      need = std::max<uint64_t>(need, (uint64_t)rs->mirror[(size_t)u].emit_next + (uint64_t)utt_frames[u] * (uint64_t)B +
                                          (uint64_t)ctcdec_stream::CAP + 2);
    if (need > rs->emit_cap) {
      // (sized for the worst case -- one node per frame and beam --, of which a real stream uses a few per cent: grow
      // in big steps so that a long stream reallocates a handful of times)
      const uint64_t cap = std::max<uint64_t>(std::max<uint64_t>(4 * need, 2 * rs->emit_cap), 4096);
      DevBuf grown;
      if (grown.ensure((size_t)n_utts * cap * sizeof(EmitNode), &err)) return fail(CTCDEC_ERR_DEVICE, err);
      for (int32_t u = 0; u < n_utts && rs->emit.p; ++u) {
        const size_t used = (size_t)rs->mirror[(size_t)u].emit_next * sizeof(EmitNode);
        if (used && be::d2d((char*)grown.p + (size_t)u * cap * sizeof(EmitNode), (const char*)rs->emit.p + (size_t)u * rs->emit_cap * sizeof(EmitNode), used, &err)) {
          grown.drop();  // (the stream keeps its old arena)
          return fail(CTCDEC_ERR_DEVICE, err);
        }
      }
      // the new offsets first: a failure here leaves the stream on its old arena with its old stride
      std::vector<uint64_t> eo((size_t)n_utts + 1);
      for (int32_t u = 0; u <= n_utts; ++u) eo[(size_t)u] = (uint64_t)u * cap;
      if (upload(rs->eoff, eo, &err)) {
        grown.drop();
        return fail(CTCDEC_ERR_DEVICE, err);
      }
      rs->emit.drop();
      rs->emit = grown;
      rs->emit_cap = cap;
    }
  } else if (stream) {
    const int64_t n_imp_total = stream->beam_off[n_utts];
    std::vector<ImportBeam> imps((size_t)std::max<int64_t>(n_imp_total, 1));
    std::vector<LmState> imps_x(K > 1 ? (size_t)std::max<int64_t>(n_imp_total, 1) * (size_t)(K - 1) : 0);
    std::vector<int64_t> ioff(stream->beam_off, stream->beam_off + n_utts + 1);
    std::vector<int32_t> ff(stream->first_frame, stream->first_frame + n_utts);
    for (int32_t u = 0; u < n_utts; ++u) {
      int64_t cnt = ioff[(size_t)u + 1] - ioff[(size_t)u];
      if (cnt < 1 || cnt > shape_bw_limit(B))
        return fail(CTCDEC_ERR_ARG, "a stream must carry between 1 and beam-capacity beams");
      max_import = std::max<int32_t>(max_import, (int32_t)cnt);
      for (int64_t k = ioff[(size_t)u]; k < ioff[(size_t)u + 1]; ++k) {
        std::string e = build_import(dec, *stream, k, B, &imps[(size_t)k],
                                     K > 1 ? &imps_x[(size_t)k * (size_t)(K - 1)] : nullptr);
        if (!e.empty()) return fail(CTCDEC_ERR_ARG, e);
      }
    }
    if (upload(dec->w_imp, imps, &err) || upload(dec->w_impoff, ioff, &err) || upload(dec->w_ff, ff, &err))
      return fail(CTCDEC_ERR_DEVICE, err);
    d_imports = (const ImportBeam*)dec->w_imp.p;
    if (K > 1) {
      if (upload(dec->w_impx, imps_x, &err)) return fail(CTCDEC_ERR_DEVICE, err);
      d_import_x = (const LmState*)dec->w_impx.p;
    }
  }
  if (dec->w_text.ensure(toff[(size_t)n_utts] * sizeof(TextNode), &err) ||
      (!rs && (dec->w_emit.ensure(eoff[(size_t)n_utts] * sizeof(EmitNode), &err) || upload_staged(dec, dec->w_eoff, eoff, &err))) ||
      upload_staged(dec, dec->w_toff, toff, &err) || dec->w_out.ensure((size_t)n_utts * n_best * sizeof(OutBeam), &err) ||
      dec->w_nout.ensure((size_t)n_utts * 4, &err) || dec->w_status.ensure((size_t)n_utts * 4, &err) ||
      dec->w_tok.ensure((size_t)std::max<unsigned long long>(tok_cap, 1) * sizeof(EmitNode), &err) ||
      dec->w_head.ensure(16, &err) || dec->w_cold.ensure((size_t)n_utts * 2 * COLD_STRIDE * sizeof(ColdRec), &err))
    return fail(CTCDEC_ERR_DEVICE, err);
  const LmState* d_start = nullptr;
  if (stream && !rs) {
    d_start = nullptr;  // every imported beam carries its own LM state
  } else if (dec->has_lm) {  // (resident streams: used by the streams that are at their starting state)
    // per utterance one state per model; a negative length (or no array) asks for the model's own default
    std::vector<LmState> st((size_t)n_utts * K);
    for (int32_t u = 0; u < n_utts; ++u) {
      for (int k = 0; k < K; ++k) {
        LmState& s = st[(size_t)u * K + k];
        memset(&s, 0, sizeof(s));
        const HostLM& lm = K > 1 ? *dec->multi->lms[(size_t)k] : dec->lm_ref();
        const ctcdec_lm_state* given = start_states ? &start_states[(size_t)u * K + k] : nullptr;
        if (!given || given->length < 0) {
          const bool boundary = k == 0 ? p->lm_score_boundary != 0 : dec->x_boundary[k] != 0;
          lm.start_state(boundary, &s);
        } else {
          if (given->length > MAX_CTX) return fail(CTCDEC_ERR_ARG, "LM start state too long");
          s.len = given->length;
          for (int j = 0; j < s.len; ++j) {
            if (given->words[j] >= lm.words.size()) return fail(CTCDEC_ERR_ARG, "bad LM state word");
            s.words[j] = given->words[j];
            s.backoff[j] = given->backoff[j];
          }
        }
      }
    }
    if (upload_staged(dec, dec->w_start, st, &err)) return fail(CTCDEC_ERR_DEVICE, err);
    d_start = (const LmState*)dec->w_start.p;
  }
  const size_t xstate_bytes = K > 1 ? (size_t)n_utts * n_best * (size_t)(K - 1) * sizeof(LmState) : 0;
  if (xstate_bytes && dec->w_xstate.ensure(xstate_bytes, &err)) return fail(CTCDEC_ERR_DEVICE, err);

  be::BeamArgs ba;
  device_tables(dec, &ba.tables);
  DecodeParams& dp = ba.params;
  memset(&dp, 0, sizeof(dp));
  dp.beam_width = B;
  dp.prune_history = p->prune_history ? 1 : 0;
  dp.n_best = n_best;
  dp.first_frame = p->first_frame;
  dp.beam_prune_logp = p->beam_prune_logp;
  dp.token_min_logp = p->token_min_logp;
  dp.hot_weight = p->hotword_weight;
  dp.alpha = p->alpha;
  dp.beta = p->beta;
  dp.unk = p->unk_score_offset;
  dp.log_base_change = p->log_base_change;
  dp.score_boundary = p->lm_score_boundary ? 1 : 0;
  dp.fold = stream ? stream->fold : 1;
  dp.eos = stream ? stream->eos : 1;
  dp.no_label_runs = getenv("CTCDEC_NO_LABEL_RUNS") != nullptr ? 1 : 0;
  // decode_batch: the kernels assemble the best beam's text themselves (CTCDEC_HOST_REPLAY=1: the emission lists come
  // back and the host replays them, as for every other call)
  const bool device_texts = p->texts_only != 0 && n_best == 1 && !stream && getenv("CTCDEC_HOST_REPLAY") == nullptr;
  dp.texts_only = device_texts ? 1 : 0;
  ba.n_utts = n_utts;
  ba.utt_row0 = (const int64_t*)dec->w_row0.p;
  ba.surv_cnt = (const uint32_t*)dec->w_scnt.p;
  ba.surv_id = (const uint16_t*)dec->w_sid.p;
  ba.surv_lp = (const double*)dec->w_slp.p;
  ba.text_nodes = (TextNode*)dec->w_text.p;
  ba.emit_nodes = (EmitNode*)dec->w_emit.p;
  ba.text_off = (const uint64_t*)dec->w_toff.p;
  ba.emit_off = (const uint64_t*)dec->w_eoff.p;
  ba.start_states = d_start;
  ba.out_xstates = xstate_bytes ? (LmState*)dec->w_xstate.p : nullptr;
  ba.out = (OutBeam*)dec->w_out.p;
  ba.out_stride = n_best;
  ba.n_out = (uint32_t*)dec->w_nout.p;
  ba.status = (uint32_t*)dec->w_status.p;
  ba.tok_pool = (EmitNode*)dec->w_tok.p;
  ba.tok_pool_head = (unsigned long long*)dec->w_head.p;
  ba.tok_pool_cap = tok_cap;
  ba.prof = nullptr;
  ba.imports = d_imports;
  ba.import_xstates = d_import_x;
  ba.import_off = (stream && !rs) ? (const int64_t*)dec->w_impoff.p : nullptr;
  ba.first_frames = stream ? (const int32_t*)dec->w_ff.p : nullptr;
  ba.cold = (ColdRec*)dec->w_cold.p;
  ba.max_import = max_import;
  ba.text_scratch = nullptr;
  ba.text_soff = nullptr;
  ba.text_pool = nullptr;
  ba.text_pool_cap = 0;
  if (device_texts) {
    // scratch per utterance: at most one emission per frame, each a label and a separator
    std::vector<uint64_t> soff((size_t)n_utts + 1, 0);
    for (int32_t u = 0; u < n_utts; ++u)
      soff[(size_t)u + 1] = soff[(size_t)u] + (((uint64_t)utt_frames[u] + 2) * (uint64_t)(dec->max_label_bytes + 1) + 15) / 16 * 16;
    if (dec->w_tscr.ensure((size_t)soff[(size_t)n_utts] + 16, &err) || dec->w_tpool.ensure((size_t)soff[(size_t)n_utts] + 16, &err) ||
        upload_staged(dec, dec->w_tsoff, soff, &err))
      return fail(CTCDEC_ERR_DEVICE, err);
    ba.text_scratch = (uint8_t*)dec->w_tscr.p;
    ba.text_soff = (const uint64_t*)dec->w_tsoff.p;
    ba.text_pool = (uint8_t*)dec->w_tpool.p;
    ba.text_pool_cap = soff[(size_t)n_utts];
  }
  ba.carry_out = nullptr;
  ba.carry_xstates = nullptr;
  ba.sstate = nullptr;
  ba.carry_stride = 0;
  ba.want_out = 1;
  ba.resident_in = 0;
  // ragged batches of more utterances than fit the device at once: longest first (BeamArgs::order)
  ba.order = nullptr;
  if (n_utts > be::cus() * 2 && !getenv("CTCDEC_NO_LPT_ORDER")) {
    bool ragged = false;
    for (int32_t u = 1; u < n_utts; ++u) ragged = ragged || utt_frames[u] != utt_frames[0];
    if (ragged) {
      std::vector<int32_t> order((size_t)n_utts);
      for (int32_t u = 0; u < n_utts; ++u) order[(size_t)u] = u;
      std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return utt_frames[x] > utt_frames[y]; });
      if (upload(dec->w_order, order, &err)) return fail(CTCDEC_ERR_DEVICE, err);
      ba.order = (const int32_t*)dec->w_order.p;
    }
  }
  if (rs) {
    ba.emit_nodes = (EmitNode*)rs->emit.p;
    ba.emit_off = (const uint64_t*)rs->eoff.p;
    ba.imports = (const ImportBeam*)rs->carry.p;
    ba.import_xstates = K > 1 ? (const LmState*)rs->carry_x.p : nullptr;
    ba.import_off = nullptr;
    ba.resident_in = 1;
    ba.carry_out = (ImportBeam*)rs->carry.p;
    ba.carry_xstates = K > 1 ? (LmState*)rs->carry_x.p : nullptr;
    ba.sstate = (StreamState*)rs->sstate.p;
    ba.carry_stride = ctcdec_stream::CAP;
    ba.want_out = want_result ? 1 : 0;
  }
  if (dec->profile) {
    if (dec->w_prof.ensure(N_PROF * 8, &err) || be::zero(dec->w_prof.p, N_PROF * 8, &err)) return fail(CTCDEC_ERR_DEVICE, err);
    ba.prof = (unsigned long long*)dec->w_prof.p;
  }
  // Everything the beam stage needs is staged BEFORE the prune stage is launched, and the beam kernel is
  // queued right behind it on the same stream: the host never sits between the two kernels. The two
  // rare events the prune stage can report (probability-like input, survivor overflow) are read back
  // afterwards and simply redo the affected stage(s).
  if (dec->w_flags.ensure(32, &err) || dec->w_head.ensure(16, &err)) return fail(CTCDEC_ERR_DEVICE, err);
  ba.surv_x16 = 0;
  ba.total_rows = R;
  // Small batches choose their beam kernel by the input (backend: wave_kernel_chosen): their beam stage is launched when the
  // prune stage has reported -- like a resident stream's --, one small read-back between the two stages.
  auto t_setup = t_begin, t_queued = t_begin, t_flags = t_begin;  // (CTCDEC_HOST_TIMING: where the host side of a call goes)
  const bool by_input = !rs && be::beam_kernel_depends_on_input(ba);
  const bool late_beam = rs != nullptr || by_input;
  for (int attempt = 0; attempt < 2; ++attempt) {
    size_t rows = (size_t)std::max<int64_t>(R, 1);
    if (dec->w_rowsum.ensure(rows * 8, &err) || dec->w_isprob.ensure((size_t)n_utts * 4, &err) ||
        dec->w_scnt.ensure(rows * 4, &err) || dec->w_sid.ensure(rows * max_surv * 2, &err) ||
        dec->w_slp.ensure(rows * max_surv * 8, &err) || dec->w_flags.ensure(32, &err) || dec->w_slow.ensure(rows * 4, &err))
      return fail(CTCDEC_ERR_DEVICE, err);
    if (be::zero(dec->w_flags.p, 32, &err)) return fail(CTCDEC_ERR_DEVICE, err);
    be::PruneArgs pa;
    pa.utt_logits = (const void* const*)dec->w_ptrs.p;
    pa.utt_row0 = (const int64_t*)dec->w_row0.p;
    pa.n_utts = n_utts;
    pa.n_rows = R;
    pa.n_labels = V;
    pa.dtype = dtype;
    pa.token_min_logp = p->token_min_logp;
    pa.max_surv = max_surv;
    pa.row_sum = (double*)dec->w_rowsum.p;
    pa.utt_is_prob = (uint32_t*)dec->w_isprob.p;
    pa.surv_cnt = (uint32_t*)dec->w_scnt.p;
    pa.surv_id = (uint16_t*)dec->w_sid.p;
    pa.surv_lp = (double*)dec->w_slp.p;
    pa.overflow = (uint32_t*)dec->w_flags.p;
    pa.pass = 0;
    pa.row_base = 0;
    pa.slow_rows = (uint32_t*)dec->w_slow.p;
    pa.utt_side = dec->slicing ? (uint32_t*)dec->w_side.p : nullptr;
    pa.utt_sum = dec->slicing ? (double*)((char*)dec->w_side.p + (((size_t)n_utts * 4 + 15) & ~(size_t)15)) : nullptr;
    pa.dense_hint = 0;
    if (dec->dense_calls > 0 && !rs) {
      pa.dense_hint = 1;
      if (attempt == 0) --dec->dense_calls;
    }
    pa.rows_aligned16 = 1;
    pa.rows_aligned4 = 1;
    for (const void* q : ptrs) {
      if (((uintptr_t)q & 15u) != 0) pa.rows_aligned16 = 0;
      if (((uintptr_t)q & 3u) != 0) pa.rows_aligned4 = 0;
    }
    ba.surv_cnt = (const uint32_t*)dec->w_scnt.p;
    ba.surv_id = (const uint16_t*)dec->w_sid.p;
    ba.surv_lp = (const double*)dec->w_slp.p;
    dp.max_surv = max_surv;
    // the wave kernel's payload lines (one per candidate a frame can push into its pool)
    auto reserve_pay = [&]() -> int {
      ba.pay = nullptr;
      ba.pay_stride = 0;
      if (be::wave_kernel_chosen(ba)) {  // reserved only for launches that will use it (2 GB at the bench size)
        ba.pay_stride = (uint64_t)wave_pay_stride(dp);
        if (dec->w_pay.ensure((size_t)n_utts * (size_t)ba.pay_stride * sizeof(PoolPay), &err)) return -1;
        ba.pay = (PoolPay*)dec->w_pay.p;
      }
      return 0;
    };
    if (!by_input && reserve_pay()) return fail(CTCDEC_ERR_DEVICE, err);
    auto run_beam = [&]() -> int {
      if (be::zero(dec->w_head.p, 16, &err)) return -1;
      return be::launch_beam(ba, &err);
    };
    // (a resident stream's beam kernel advances persistent state: it is launched once, when the prune stage has
    // reported -- and so is a small batch's, whose kernel is chosen by what the prune stage counted; everything else
    // launches it right behind the first prune pass and redoes it in the two rare cases)
    t_setup = std::chrono::steady_clock::now();
    if (be::launch_prune(pa, &err) || (!late_beam && run_beam())) return fail(CTCDEC_ERR_DEVICE, err);
    t_queued = std::chrono::steady_clock::now();
    uint32_t flags[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (be::d2h(flags, dec->w_flags.p, 32, &err)) return fail(CTCDEC_ERR_DEVICE, err);
    t_flags = std::chrono::steady_clock::now();
    uint32_t surv_total = flags[4];  // (pass 0 counted every row as logits)
    // small vocabularies fed flat logits: nearly every row overflows the 64-rows-per-wave kernel's sixteen candidates and is done
    // again by the per-row kernel -- the next sixteen calls go there directly (then the fast kernel is tried again)
    if (V <= 128 && R >= 1024 && (uint64_t)flags[3] * 2 > (uint64_t)R) dec->dense_calls = 16;
    if (flags[2]) {  // rows that sum to about 1: the reference's test in its own dtype and summation order (decoder.py:760)
      if (be::launch_sniff_exact(pa, &err) || be::d2h(flags, dec->w_flags.p, 16, &err)) return fail(CTCDEC_ERR_DEVICE, err);
    }
    if (flags[1]) {  // some utterance holds probabilities: redo those rows as log(clip(p)), then the beams
      pa.pass = 1;
      // pass-0 overflows of those rows are void, and so is its survivor count (flags[4]: the pass counts them again)
      if (be::zero(dec->w_flags.p, 4, &err) || be::zero((char*)dec->w_flags.p + 16, 4, &err)) return fail(CTCDEC_ERR_DEVICE, err);
      if (be::launch_prune(pa, &err) || (!late_beam && run_beam())) return fail(CTCDEC_ERR_DEVICE, err);
      if (be::d2h(flags, dec->w_flags.p, 32, &err)) return fail(CTCDEC_ERR_DEVICE, err);
      surv_total = flags[4];
    }
    const uint32_t ovf = flags[0];
    if (!ovf) {
      if (by_input) {
        const double mean = R > 0 ? (double)surv_total / (double)R : 0.0;
        ba.surv_x16 = (int32_t)std::min(1.0e6, std::max(1.0, mean * 16.0 + 0.5));
        if (reserve_pay()) return fail(CTCDEC_ERR_DEVICE, err);
      }
      if (late_beam && run_beam()) return fail(CTCDEC_ERR_DEVICE, err);
      break;
    }
    if (max_surv == V) return fail(CTCDEC_ERR_INTERNAL, "survivor overflow at full vocabulary");
    max_surv = V;  // un-normalised probability rows can exceed the bound: redo at full width
  }

  if (after_launch && *after_launch && (*after_launch)(&err)) return fail(CTCDEC_ERR_DEVICE, err);
  // results back (page-locked staging: the token pool is a few MB per batch)
  const bool host_timing = getenv("CTCDEC_HOST_TIMING") != nullptr;
  auto t_launch = std::chrono::steady_clock::now();
  if (dec->h_small.ensure((size_t)n_utts * 8 + 16, &err) ||
      dec->h_out.ensure((size_t)n_utts * n_best * sizeof(OutBeam), &err))
    return fail(CTCDEC_ERR_DEVICE, err);
  uint32_t* n_out = (uint32_t*)dec->h_small.p;
  uint32_t* status = n_out + n_utts;
  unsigned long long head = 0;
  // (one wait for the three: the targets are page-locked, `heads` rides in the spare 16 bytes behind the status words)
  unsigned long long* heads_pinned = (unsigned long long*)(status + n_utts);
  if (be::d2h_async(n_out, dec->w_nout.p, (size_t)n_utts * 4, &err) ||
      be::d2h_async(status, dec->w_status.p, (size_t)n_utts * 4, &err) || be::d2h_async(heads_pinned, dec->w_head.p, 16, &err) ||
      be::sync(&err))
    return fail(CTCDEC_ERR_DEVICE, err);
  head = heads_pinned[0];
  bool outgrown_redone = false;
  if (!arenas_full) {
    bool outgrown = false;
    for (int32_t u = 0; u < n_utts; ++u) outgrown = outgrown || (status[u] & (ST_TEXT_OVERFLOW | ST_EMIT_OVERFLOW)) != 0;
    if (outgrown) {  // (rare: flat posteriors that complete a word for every beam in every frame) the beam stage again,
      // with the worst case reserved
      if (getenv("CTCDEC_ARENA_TRACE")) fprintf(stderr, "[ctcdec host] node arenas outgrown: beam stage redone with the worst case\n");
      dec->arenas_worst_case = true;
      arenas_full = true;
      outgrown_redone = true;
      size_arenas(true);
      if (dec->w_text.ensure(toff[(size_t)n_utts] * sizeof(TextNode), &err) ||
          dec->w_emit.ensure(eoff[(size_t)n_utts] * sizeof(EmitNode), &err) || upload(dec->w_toff, toff, &err) ||
          upload(dec->w_eoff, eoff, &err))
        return fail(CTCDEC_ERR_DEVICE, err);
      ba.text_nodes = (TextNode*)dec->w_text.p;
      ba.emit_nodes = (EmitNode*)dec->w_emit.p;
      ba.text_off = (const uint64_t*)dec->w_toff.p;
      ba.emit_off = (const uint64_t*)dec->w_eoff.p;
      if (be::zero(dec->w_head.p, 16, &err) || be::launch_beam(ba, &err) ||
          be::d2h(n_out, dec->w_nout.p, (size_t)n_utts * 4, &err) || be::d2h(status, dec->w_status.p, (size_t)n_utts * 4, &err) ||
          be::d2h(&head, dec->w_head.p, 8, &err))
        return fail(CTCDEC_ERR_DEVICE, err);
    }
  }
  auto t_kernel = std::chrono::steady_clock::now();
  if (rs) {  // the streams have moved on, whatever the chunk's outcome: refresh the mirrors first
    if (be::d2h(rs->mirror.data(), rs->sstate.p, (size_t)n_utts * sizeof(StreamState), &err)) return fail(CTCDEC_ERR_DEVICE, err);
    for (int32_t u = 0; u < n_utts; ++u) rs->frames[(size_t)u] += utt_frames[u];
    rs->pushes += 1;
    if (stream->eos) {  // decoder.py:681-728 with is_end: the next chunk starts a new utterance
      for (auto& m : rs->mirror) {
        m.n_carry = 0;
        m.emit_next = 1;
        m.status = 0;
      }
      if (be::h2d(rs->sstate.p, rs->mirror.data(), (size_t)n_utts * sizeof(StreamState), &err)) return fail(CTCDEC_ERR_DEVICE, err);
      rs->has_import = false;
      std::fill(rs->frames.begin(), rs->frames.end(), 0);
      rs->pushes = 0;
    }
  }
  for (int32_t u = 0; u < n_utts; ++u)
    if (status[u] & ST_NO_BEAMS)  // the reference: ValueError from max([]) (decoder.py:545 / :585)
      return fail(CTCDEC_ERR_ARG, "max() arg is an empty sequence (utterance " + std::to_string(u) +
                                      ": no beam survived -- non-finite scores or a positive beam_prune_logp)");
  for (int32_t u = 0; u < n_utts; ++u)
    if (status[u]) return fail(CTCDEC_ERR_INTERNAL, "beam kernel status " + std::to_string(status[u]) +
                                                        " for utterance " + std::to_string(u));
  if (device_texts) {  // one block of text per utterance, written by the kernels
    unsigned long long heads[2] = {heads_pinned[0], heads_pinned[1]};  // (read back with the counters above)
    if (outgrown_redone && be::d2h(heads, dec->w_head.p, 16, &err)) return fail(CTCDEC_ERR_DEVICE, err);
    res->device_texts = true;
    res->dev_out.resize((size_t)n_utts);
    res->dev_texts.resize((size_t)heads[1]);
    if (be::d2h(res->dev_out.data(), dec->w_out.p, (size_t)n_utts * sizeof(OutBeam), &err) ||
        (heads[1] && be::d2h(&res->dev_texts[0], dec->w_tpool.p, (size_t)heads[1], &err)))
      return fail(CTCDEC_ERR_DEVICE, err);
    for (int32_t u = 0; u < n_utts; ++u) {
      const OutBeam& ob = res->dev_out[(size_t)u];
      if (n_out[u] != 1 || (unsigned long long)ob.tok_off + ob.tok_cnt > heads[1]) return fail(CTCDEC_ERR_INTERNAL, "text pool range");
    }
    be::last_timing(&res->ms[0], &res->ms[1]);
    res->beam_kernel = be::last_beam_kernel();
    if (dec->profile && be::d2h(dec->prof, dec->w_prof.p, N_PROF * 8, &err)) return fail(CTCDEC_ERR_DEVICE, err);
    res->ms[2] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (host_timing) {
      auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
      };
      fprintf(stderr, "[ctcdec host] texts from the device: %llu bytes, native call %.3f ms (kernels %.3f + %.3f): setup %.3f, launches %.3f, "
                      "wait for the kernels %.3f, counters back %.3f, records + texts back %.3f\n", heads[1], res->ms[2], res->ms[0], res->ms[1],
              ms(t_begin, t_setup), ms(t_setup, t_queued), ms(t_queued, t_flags), ms(t_flags, t_kernel), ms(t_kernel, std::chrono::steady_clock::now()));
    }
    *out = res.release();
    return CTCDEC_OK;
  }
  if (!want_result) {  // a resident stream between reads: nothing to bring back
    if (host_timing) {
      auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
      };
      double pm = 0, bm = 0;
      be::last_timing(&pm, &bm);
      fprintf(stderr, "[ctcdec host] stream push: setup+prune+launch %.3f ms, wait beam kernel %.3f ms, total %.3f ms (prune kernel %.3f, beam kernel %.3f)\n",
              ms(t_begin, t_launch), ms(t_launch, t_kernel), ms(t_begin, std::chrono::steady_clock::now()), pm, bm);
    }
    be::last_timing(&res->ms[0], &res->ms[1]);
    res->beam_kernel = be::last_beam_kernel();
    res->ms[2] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    *out = res.release();
    return CTCDEC_OK;
  }
  const OutBeam* obs = (const OutBeam*)dec->h_out.p;
  if (be::d2h(dec->h_out.p, dec->w_out.p, (size_t)n_utts * n_best * sizeof(OutBeam), &err))
    return fail(CTCDEC_ERR_DEVICE, err);
  if (dec->h_tok.ensure((size_t)head * sizeof(EmitNode) + 16, &err)) return fail(CTCDEC_ERR_DEVICE, err);
  const EmitNode* toks = (const EmitNode*)dec->h_tok.p;
  if (head && be::d2h(dec->h_tok.p, dec->w_tok.p, (size_t)head * sizeof(EmitNode), &err))
    return fail(CTCDEC_ERR_DEVICE, err);
  const LmState* xst = nullptr;
  if (xstate_bytes) {
    if (dec->h_xstate.ensure(xstate_bytes, &err) || be::d2h(dec->h_xstate.p, dec->w_xstate.p, xstate_bytes, &err))
      return fail(CTCDEC_ERR_DEVICE, err);
    xst = (const LmState*)dec->h_xstate.p;
  }
  auto t_copy = std::chrono::steady_clock::now();
  be::last_timing(&res->ms[0], &res->ms[1]);
  res->beam_kernel = be::last_beam_kernel();
  if (dec->profile && be::d2h(dec->prof, dec->w_prof.p, N_PROF * 8, &err)) return fail(CTCDEC_ERR_DEVICE, err);

  for (int32_t u = 0; u < n_utts; ++u)
    for (uint32_t k = 0; k < n_out[u]; ++k) {
      const OutBeam& ob = obs[(size_t)u * n_best + k];
      if ((unsigned long long)ob.tok_off + ob.tok_cnt > head) return fail(CTCDEC_ERR_INTERNAL, "token pool range");
    }
  auto replay_range = [&](int32_t u0, int32_t u1) {
    for (int32_t u = u0; u < u1; ++u) {
      auto& beams = res->utts[(size_t)u];
      beams.resize(n_out[u]);
      for (uint32_t k = 0; k < n_out[u]; ++k) {
        const OutBeam& ob = obs[(size_t)u * n_best + k];
        BeamResult& r = beams[k];
        fill_result(ob, xst ? &xst[((size_t)u * n_best + k) * (size_t)(K - 1)] : nullptr, K, &r);
        replay(dec, toks + ob.tok_off, ob.tok_cnt, stream, stream ? stream->beam_off[u] : 0, &r);
      }
    }
  };
  // host replay is independent per utterance: a few threads once there is enough of it
  if (head > 50000 && n_utts >= 16) {
    if (!dec->replay_pool) {
      unsigned want = 31u;  // + the calling thread
      if (const char* env = getenv("CTCDEC_REPLAY_THREADS")) want = (unsigned)std::max(0, atoi(env) - 1);
      const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
      dec->replay_pool.reset(new ReplayPool((int)std::min(want, hw > 1 ? hw - 1 : 0u)));
    }
    const std::function<void(int32_t, int32_t)> job = replay_range;
    dec->replay_pool->run(n_utts, 4, job);
  } else {
    replay_range(0, n_utts);
  }
  auto t_end = std::chrono::steady_clock::now();
  res->ms[2] = std::chrono::duration<double, std::milli>(t_end - t_begin).count();
  if (host_timing) {
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::milli>(b - a).count();
    };
    fprintf(stderr, "[ctcdec host] setup+launch %.3f ms, wait kernels %.3f ms, copy back %.3f ms (%llu tokens), replay %.3f ms\n",
            ms(t_begin, t_launch), ms(t_launch, t_kernel), ms(t_kernel, t_copy), head, ms(t_copy, t_end));
  }
  *out = res.release();
  return CTCDEC_OK;
}

// ---- device-resident streams ------------------------------------------------------------------------------------
int ctcdec_stream_open(ctcdec_decoder* dec, int32_t n_streams, const ctcdec_lm_state* start_states, ctcdec_stream** out) {
  if (!dec || !out || n_streams < 1) return fail(CTCDEC_ERR_ARG, "bad arguments");
  std::string err;
  std::lock_guard<std::mutex> device_lock(g_device_mu);
  if (be::bind_thread(&err)) return fail(CTCDEC_ERR_DEVICE, err);
  std::unique_ptr<ctcdec_stream> st(new ctcdec_stream());
  st->dec = dec;
  st->n = n_streams;
  st->K = dec->has_lm ? dec->n_lms() : 1;
  if (start_states && dec->has_lm) st->start_states.assign(start_states, start_states + (size_t)n_streams * st->K);
  st->mirror.assign((size_t)n_streams, StreamState{0u, 1u, 0u, 0u});
  st->frames.assign((size_t)n_streams, 0);
  st->imp_off.assign((size_t)n_streams + 1, 0);
  if (st->carry.ensure((size_t)n_streams * ctcdec_stream::CAP * sizeof(ImportBeam), &err) ||
      (st->K > 1 && st->carry_x.ensure((size_t)n_streams * ctcdec_stream::CAP * (size_t)(st->K - 1) * sizeof(LmState), &err)) ||
      st->sstate.ensure((size_t)n_streams * sizeof(StreamState), &err) ||
      be::h2d(st->sstate.p, st->mirror.data(), (size_t)n_streams * sizeof(StreamState), &err))
    return fail(CTCDEC_ERR_DEVICE, err);
  *out = st.release();
  return CTCDEC_OK;
}

int ctcdec_stream_push(ctcdec_stream* st, const void* const* chunk_logits, const int32_t* chunk_frames, int32_t dtype,
                       int32_t is_device, const ctcdec_params* params, const int32_t* first_frame, int32_t force_next_word,
                       int32_t is_end, int32_t want_result, ctcdec_result** out) {
  if (!st || !params || !chunk_frames) return fail(CTCDEC_ERR_ARG, "bad arguments");
  if ((want_result || is_end) && !out) return fail(CTCDEC_ERR_ARG, "a result is wanted but there is nowhere to put it");
  std::vector<int32_t> ff((size_t)st->n);
  for (int32_t u = 0; u < st->n; ++u) ff[(size_t)u] = first_frame ? first_frame[u] : (int32_t)st->frames[(size_t)u];
  StreamIn sin;
  sin.first_frame = ff.data();
  sin.beams = st->has_import ? st->imp_beams.data() : nullptr;
  sin.beam_off = st->imp_off.data();
  sin.text_blob = st->imp_blob.data();
  sin.fold = (force_next_word || is_end) ? 1 : 0;
  sin.eos = is_end ? 1 : 0;
  ctcdec_result* res = nullptr;
  const int rc = decode_impl(st->dec, chunk_logits, chunk_frames, st->n, dtype, is_device, params,
                             st->start_states.empty() ? nullptr : st->start_states.data(), &sin, &res, st,
                             want_result != 0 || is_end != 0);
  if (rc != CTCDEC_OK) return rc;
  if (out) *out = res;
  else ctcdec_result_free(res);
  return CTCDEC_OK;
}

int ctcdec_stream_read(ctcdec_stream* st, const ctcdec_params* params, ctcdec_result** out) {
  if (!st || !params || !out) return fail(CTCDEC_ERR_ARG, "bad arguments");
  // a chunk of zero frames: the finalisation ranks the carried beams again (same beams, same order) and this time
  // writes output records and back-traces the emission chains
  std::vector<const void*> ptrs((size_t)st->n, nullptr);
  std::vector<int32_t> zero((size_t)st->n, 0);
  return ctcdec_stream_push(st, ptrs.data(), zero.data(), CTCDEC_F32, 0, params, nullptr, 0, 0, 1, out);
}

int ctcdec_stream_import(ctcdec_stream* st, const ctcdec_beam_in* beams, const int64_t* beam_off, const char* text_blob,
                         int64_t text_bytes) {
  if (!st || !beams || !beam_off || !text_blob || text_bytes < 0) return fail(CTCDEC_ERR_ARG, "bad arguments");
  ctcdec_decoder* dec = st->dec;
  std::string err;
  std::lock_guard<std::mutex> device_lock(g_device_mu);
  if (be::bind_thread(&err)) return fail(CTCDEC_ERR_DEVICE, err);
  if (sync_tables(dec, &err)) return fail(CTCDEC_ERR_DEVICE, err);  // (the hot-word view of the partial words)
  const int K = st->K;
  StreamIn sin;
  sin.first_frame = nullptr;
  sin.beams = beams;
  sin.beam_off = beam_off;
  sin.text_blob = text_blob;
  sin.fold = sin.eos = 0;
  std::vector<ImportBeam> imps((size_t)st->n * ctcdec_stream::CAP);
  std::vector<LmState> imps_x(K > 1 ? imps.size() * (size_t)(K - 1) : 0);
  for (int32_t u = 0; u < st->n; ++u) {
    const int64_t cnt = beam_off[u + 1] - beam_off[u];
    if (cnt < 1 || cnt > ctcdec_stream::CAP) return fail(CTCDEC_ERR_ARG, "a stream must carry between 1 and 256 beams");
    for (int64_t k = 0; k < cnt; ++k) {
      const size_t slot = (size_t)u * ctcdec_stream::CAP + (size_t)k;
      std::string e = build_import(dec, sin, beam_off[u] + k, 0, &imps[slot], K > 1 ? &imps_x[slot * (size_t)(K - 1)] : nullptr);
      if (!e.empty()) return fail(CTCDEC_ERR_ARG, e);
    }
  }
  // the new roots go behind what the arena already holds; the old chains are unreachable from now on
  for (int32_t u = 0; u < st->n; ++u) st->mirror[(size_t)u].n_carry = (uint32_t)(beam_off[u + 1] - beam_off[u]);
  if (be::h2d(st->carry.p, imps.data(), imps.size() * sizeof(ImportBeam), &err) ||
      (K > 1 && be::h2d(st->carry_x.p, imps_x.data(), imps_x.size() * sizeof(LmState), &err)) ||
      be::h2d(st->sstate.p, st->mirror.data(), (size_t)st->n * sizeof(StreamState), &err))
    return fail(CTCDEC_ERR_DEVICE, err);
  const int64_t nb = beam_off[st->n];
  st->imp_beams.assign(beams, beams + nb);
  for (auto& b : st->imp_beams) b.more_states = nullptr;  // (copied into the carry rows above)
  st->imp_off.assign(beam_off, beam_off + st->n + 1);
  st->imp_blob.assign(text_blob, (size_t)text_bytes);
  st->has_import = true;
  return CTCDEC_OK;
}

// ---- time-sliced ingest of host batches --------------------------------------------------------------------------------
// The reference is called with numpy matrices (decoder.py:730-775). Copying a large host batch takes longer than decoding it
// (2.1 GB: 37 ms over PCIe, 13 ms of kernels), and an utterance's beam search is ~11 us x T of latency whatever the batch size,
// so overlapping per-utterance chunks would still end with one full-length decode after the last copy. TIME slices do not:
// the batch goes through the device-resident stream machinery (ctcdec_stream_*: the beams stay on the device between
// slices), slice k + 1 is copied -- hipMemcpy2D of [utterances x slice bytes] with the batch's pitch, as fast from pageable
// memory as one big copy: tools/h2d_2d_probe.py -- while slice k's kernels run, and only the last slice's kernels are exposed.
// Chunked == unchunked for logits (what partial_decode_beams guarantees and the suites check); the one thing that is a property
// of the WHOLE utterance is the probability sniff (decoder.py:760): whenever any slice of any utterance is within reach of
// "mean row sum = 1", or an utterance has slices on both sides of 1, the batch is decoded again in one piece (return 1).
static int host_slices_wanted(const ctcdec_decoder* dec, const int32_t* utt_frames, int32_t n_utts, int32_t dtype) {
  const char* env = getenv("CTCDEC_HOST_SLICES");  // 0: never; n >= 2: always, in n slices (tests); unset: by size
  if (env && atoi(env) < 2) return 0;
  const size_t esz = dtype == CTCDEC_F32 ? 4 : dtype == CTCDEC_F64 ? 8 : 2;
  int64_t rows = 0, tmax = 0;
  for (int32_t u = 0; u < n_utts; ++u) {
    if (utt_frames[u] < 0) return 0;
    rows += utt_frames[u];
    tmax = std::max<int64_t>(tmax, utt_frames[u]);
  }
  const double bytes = (double)rows * (double)dec->alpha.labels.size() * (double)esz;
  int n = env ? atoi(env) : (bytes >= 512e6 ? (int)std::min(16.0, std::max(4.0, bytes / 256e6)) : 0);
  if (n > tmax / 2) n = (int)(tmax / 2);
  return n >= 2 ? n : 0;
}

static int decode_host_sliced(ctcdec_decoder* dec, const void* const* utt_logits, const int32_t* utt_frames, int32_t n_utts,
                              int32_t dtype, const ctcdec_params* p, const ctcdec_lm_state* start_states, int n_slices,
                              ctcdec_result** out) {
  if (p->beam_width < 1 || p->beam_width > CTCDEC_MAX_BEAM_WIDTH) return 1;  // (the one-piece path words the refusal)
  const size_t V = dec->alpha.labels.size();
  const size_t esz = dtype == CTCDEC_F32 ? 4 : dtype == CTCDEC_F64 ? 8 : 2;
  int64_t tmax = 0;
  for (int32_t u = 0; u < n_utts; ++u) tmax = std::max<int64_t>(tmax, utt_frames[u]);
  const int64_t C = (tmax + n_slices - 1) / n_slices;  // frames per slice
  const size_t row_bytes = V * esz, slot = (size_t)C * row_bytes;
  std::string err;
  ctcdec_stream* st = nullptr;
  int rc = ctcdec_stream_open(dec, n_utts, start_states, &st);
  if (rc != CTCDEC_OK) return rc;
  // per utterance: "a slice was read as probabilities" (u32), then the sum of all row sums seen so far (f64)
  const size_t side_off = ((size_t)n_utts * 4 + 15) & ~(size_t)15, side_bytes = side_off + (size_t)n_utts * 8;
  struct Closer {
    ctcdec_stream* s;
    ctcdec_decoder* d;
    ~Closer() {
      d->slicing = false;
      ctcdec_stream_close(s);
    }
  } closer{st, dec};
  {
    std::lock_guard<std::mutex> device_lock(g_device_mu);
    if (be::bind_thread(&err) || dec->w_logits.ensure(2 * (size_t)n_utts * slot, &err) || dec->w_side.ensure(side_bytes, &err) ||
        be::zero(dec->w_side.p, side_bytes, &err) || be::sync(&err))
      return fail(CTCDEC_ERR_DEVICE, err);
  }
  dec->slicing = true;
  std::vector<int32_t> frames((size_t)n_utts);
  std::vector<const void*> ptrs((size_t)n_utts);
  auto slice_frames = [&](int k, int32_t u) { return (int32_t)std::max<int64_t>(0, std::min<int64_t>(C, (int64_t)utt_frames[u] - (int64_t)k * C)); };
  // slice k of every utterance -> half (k & 1) of the staging buffer, utterance u at u * slot: runs of utterances that follow
  // each other in host memory with one length (a [B, T, V] array) go over in ONE two-dimensional copy
  auto copy_slice = [&](int k, std::string* e) -> int {
    char* half = (char*)dec->w_logits.p + (size_t)(k & 1) * (size_t)n_utts * slot;
    for (int32_t u = 0; u < n_utts;) {
      const int32_t f = slice_frames(k, u);
      int32_t v = u + 1;
      const size_t pitch = (size_t)utt_frames[u] * row_bytes;
      while (v < n_utts && utt_frames[v] == utt_frames[u] && (const char*)utt_logits[v] == (const char*)utt_logits[u] + (size_t)(v - u) * pitch) ++v;
      if (f > 0 && be::h2d_2d_overlapped(half + (size_t)u * slot, slot, (const char*)utt_logits[u] + (size_t)k * slot, pitch,
                                         (size_t)f * row_bytes, (size_t)(v - u), e))
        return -1;
      u = v;
    }
    return 0;
  };
  const bool trace = getenv("CTCDEC_SLICE_TRACE") != nullptr;
  if (trace) fprintf(stderr, "[ctcdec host] time-sliced ingest: %d utterances, %d slices of %lld frames\n", n_utts, n_slices, (long long)C);
  if (copy_slice(0, &err)) return fail(CTCDEC_ERR_DEVICE, err);
  ctcdec_result* res = nullptr;
  for (int k = 0; k < n_slices; ++k) {
    const bool last = k == n_slices - 1;
    char* half = (char*)dec->w_logits.p + (size_t)(k & 1) * (size_t)n_utts * slot;
    for (int32_t u = 0; u < n_utts; ++u) {
      frames[(size_t)u] = slice_frames(k, u);
      ptrs[(size_t)u] = half + (size_t)u * slot;
    }
    std::vector<int32_t> ff((size_t)n_utts);
    for (int32_t u = 0; u < n_utts; ++u) ff[(size_t)u] = (int32_t)st->frames[(size_t)u];
    StreamIn sin;
    sin.first_frame = ff.data();
    sin.beams = nullptr;
    sin.beam_off = st->imp_off.data();
    sin.text_blob = st->imp_blob.data();
    sin.fold = last ? 1 : 0;
    sin.eos = last ? 1 : 0;
    const AfterLaunch next = [&, k](std::string* e) -> int { return copy_slice(k + 1, e); };
    res = nullptr;
    rc = decode_impl(dec, ptrs.data(), frames.data(), n_utts, dtype, /*is_device=*/1, p,
                     st->start_states.empty() ? nullptr : st->start_states.data(), &sin, &res, st, /*want_result=*/last,
                     last ? nullptr : &next);
    if (rc != CTCDEC_OK) return rc;
    if (!last) ctcdec_result_free(res);
  }
  // The probability test (decoder.py:760) is about the whole utterance: the sliced decode stands when no slice was read as
  // probabilities AND the utterance's own mean row sum is safely not 1 (the slices' sums carry float32 noise of < 1e-2 per row:
  // 0.05 is far outside it and far inside what logits give) -- else the batch is decoded again in one piece.
  std::vector<unsigned char> side(side_bytes, 0);
  {
    std::lock_guard<std::mutex> device_lock(g_device_mu);
    if (be::bind_thread(&err) || be::d2h(side.data(), dec->w_side.p, side_bytes, &err)) {
      ctcdec_result_free(res);
      return fail(CTCDEC_ERR_DEVICE, err);
    }
  }
  const uint32_t* seen = (const uint32_t*)side.data();
  const double* sums = (const double*)(side.data() + side_off);
  for (int32_t u = 0; u < n_utts; ++u) {
    const double mean = utt_frames[u] > 0 ? sums[u] / (double)utt_frames[u] : NAN;
    if (seen[u] == 3u || (std::isfinite(mean) && fabs(mean - 1.0) <= 0.05)) {
      if (trace) fprintf(stderr, "[ctcdec host] time-sliced ingest: probability-like rows, decoding in one piece\n");
      ctcdec_result_free(res);
      return 1;
    }
  }
  *out = res;
  return CTCDEC_OK;
}

int ctcdec_stream_frames(const ctcdec_stream* st, int64_t* frames_out) {
  if (!st || !frames_out) return fail(CTCDEC_ERR_ARG, "bad arguments");
  for (int32_t u = 0; u < st->n; ++u) frames_out[u] = st->frames[(size_t)u];
  return CTCDEC_OK;
}

void ctcdec_stream_close(ctcdec_stream* st) {
  if (!st) return;
  std::string err;
  std::lock_guard<std::mutex> device_lock(g_device_mu);
  be::bind_thread(&err);
  delete st;
}

// Diagnostics: run only the frame-prune stage on one utterance and hand back its survivor lists.
int ctcdec_frame_survivors(ctcdec_decoder* dec, const void* logits, int32_t n_frames, int32_t dtype, int32_t is_device,
                           double token_min_logp, int32_t stride, int32_t* counts, int32_t* ids, double* logps) {
  if (!dec || !counts || !ids || !logps || n_frames < 0 || stride < 1 || (n_frames > 0 && !logits))
    return fail(CTCDEC_ERR_ARG, "bad arguments");
  if (dtype < CTCDEC_F32 || dtype > CTCDEC_BF16) return fail(CTCDEC_ERR_ARG, "dtype must be f32, f64, f16 or bf16");
  if (n_frames == 0) return CTCDEC_OK;
  std::string err;
  std::lock_guard<std::mutex> device_lock(g_device_mu);
  if (be::bind_thread(&err)) return fail(CTCDEC_ERR_DEVICE, err);
  const int V = (int)dec->alpha.labels.size();
  const size_t esz = dtype == CTCDEC_F32 ? 4 : dtype == CTCDEC_F64 ? 8 : 2;
  const size_t rows = (size_t)n_frames;
  std::vector<const void*> ptrs(1, logits);
  if (!is_device) {
    if (dec->w_logits.ensure(rows * V * esz, &err) || be::h2d(dec->w_logits.p, logits, rows * V * esz, &err))
      return fail(CTCDEC_ERR_DEVICE, err);
    ptrs[0] = dec->w_logits.p;
  }
  std::vector<int64_t> row0 = {0, (int64_t)n_frames};
  int max_surv = V;  // the decode path's bound: rows are normalised, at most floor(e^-min) labels pass
  if (token_min_logp > log(1e-15)) {
    double bound = floor(exp(-token_min_logp)) + 2.0;
    if (bound < (double)V) max_surv = (int)bound;
  }
  if (upload(dec->w_ptrs, ptrs, &err) || upload(dec->w_row0, row0, &err) || dec->w_rowsum.ensure(rows * 8, &err) ||
      dec->w_isprob.ensure(4, &err) || dec->w_scnt.ensure(rows * 4, &err) || dec->w_sid.ensure(rows * max_surv * 2, &err) ||
      dec->w_slp.ensure(rows * max_surv * 8, &err) || dec->w_flags.ensure(32, &err) || be::zero(dec->w_flags.p, 32, &err) ||
      dec->w_slow.ensure(rows * 4, &err))
    return fail(CTCDEC_ERR_DEVICE, err);
  be::PruneArgs pa;
  pa.utt_logits = (const void* const*)dec->w_ptrs.p;
  pa.utt_row0 = (const int64_t*)dec->w_row0.p;
  pa.n_utts = 1;
  pa.n_rows = n_frames;
  pa.n_labels = V;
  pa.dtype = dtype;
  pa.token_min_logp = token_min_logp;
  pa.max_surv = max_surv;
  pa.row_sum = (double*)dec->w_rowsum.p;
  pa.utt_is_prob = (uint32_t*)dec->w_isprob.p;
  pa.surv_cnt = (uint32_t*)dec->w_scnt.p;
  pa.surv_id = (uint16_t*)dec->w_sid.p;
  pa.surv_lp = (double*)dec->w_slp.p;
  pa.overflow = (uint32_t*)dec->w_flags.p;
  pa.row_base = 0;
  pa.pass = 0;
  pa.slow_rows = (uint32_t*)dec->w_slow.p;
  pa.utt_side = nullptr;
  pa.utt_sum = nullptr;
  pa.dense_hint = 0;
  pa.rows_aligned16 = (((uintptr_t)ptrs[0]) & 15u) == 0 ? 1 : 0;
  pa.rows_aligned4 = (((uintptr_t)ptrs[0]) & 3u) == 0 ? 1 : 0;
  if (be::launch_prune(pa, &err)) return fail(CTCDEC_ERR_DEVICE, err);
  uint32_t flags[4] = {0, 0, 0, 0};
  if (be::d2h(flags, dec->w_flags.p, 16, &err)) return fail(CTCDEC_ERR_DEVICE, err);
  if (flags[2] && (be::launch_sniff_exact(pa, &err) || be::d2h(flags, dec->w_flags.p, 16, &err))) return fail(CTCDEC_ERR_DEVICE, err);
  if (flags[1]) {
    pa.pass = 1;
    if (be::launch_prune(pa, &err)) return fail(CTCDEC_ERR_DEVICE, err);
  }
  std::vector<uint32_t> cnt(rows);
  std::vector<uint16_t> sid(rows * (size_t)max_surv);
  std::vector<double> slp(rows * (size_t)max_surv);
  if (be::d2h(cnt.data(), dec->w_scnt.p, rows * 4, &err) || be::d2h(sid.data(), dec->w_sid.p, sid.size() * 2, &err) ||
      be::d2h(slp.data(), dec->w_slp.p, slp.size() * 8, &err))
    return fail(CTCDEC_ERR_DEVICE, err);
  for (size_t t = 0; t < rows; ++t) {
    counts[t] = (int32_t)cnt[t];
    for (uint32_t k = 0; k < cnt[t] && k < (uint32_t)stride; ++k) {
      ids[t * (size_t)stride + k] = sid[t * (size_t)max_surv + k];
      logps[t * (size_t)stride + k] = slp[t * (size_t)max_surv + k];
    }
  }
  return CTCDEC_OK;
}

// a texts-only result (params.texts_only) as ordinary beams, for the accessors that want them
static void materialise(const ctcdec_result* cr) {
  ctcdec_result* r = const_cast<ctcdec_result*>(cr);
  if (!r || !r->device_texts || r->dev_out.empty()) return;
  for (size_t u = 0; u < r->dev_out.size(); ++u) {
    const OutBeam& ob = r->dev_out[u];
    r->utts[u].resize(1);
    BeamResult& b = r->utts[u][0];
    b.text.assign(r->dev_texts, ob.tok_off, ob.tok_cnt);
    b.logit = ob.logit_score;
    b.lm = ob.lm_score;
    b.state.length = -1;
    for (int j = 0; j < MAX_CTX; ++j) {
      b.state.words[j] = 0;
      b.state.backoff[j] = 0.f;
    }
    b.word_off.assign(1, (int32_t)b.text.size());
  }
  r->dev_out.clear();
}
int32_t ctcdec_result_num_utts(const ctcdec_result* r) { return r ? (int32_t)r->utts.size() : 0; }
int32_t ctcdec_result_num_beams(const ctcdec_result* r, int32_t utt) {
  materialise(r);
  if (!r || utt < 0 || (size_t)utt >= r->utts.size()) return 0;
  return (int32_t)r->utts[(size_t)utt].size();
}
static const BeamResult* get_beam(const ctcdec_result* r, int32_t utt, int32_t beam) {
  materialise(r);
  if (!r || utt < 0 || (size_t)utt >= r->utts.size()) return nullptr;
  const auto& b = r->utts[(size_t)utt];
  if (beam < 0 || (size_t)beam >= b.size()) return nullptr;
  return &b[(size_t)beam];
}
int ctcdec_result_text(const ctcdec_result* r, int32_t utt, int32_t beam, const char** s, int64_t* len) {
  const BeamResult* b = get_beam(r, utt, beam);
  if (!b) return fail(CTCDEC_ERR_ARG, "no such beam");
  *s = b->text.data();
  *len = (int64_t)b->text.size();
  return CTCDEC_OK;
}
int ctcdec_result_scores(const ctcdec_result* r, int32_t utt, int32_t beam, double* logit, double* lm) {
  const BeamResult* b = get_beam(r, utt, beam);
  if (!b) return fail(CTCDEC_ERR_ARG, "no such beam");
  *logit = b->logit;
  *lm = b->lm;
  return CTCDEC_OK;
}
int ctcdec_result_frames(const ctcdec_result* r, int32_t utt, int32_t beam, int32_t* n_words,
                         const int32_t** word_off, const int32_t** start, const int32_t** end) {
  const BeamResult* b = get_beam(r, utt, beam);
  if (!b) return fail(CTCDEC_ERR_ARG, "no such beam");
  *n_words = (int32_t)b->start.size();
  *word_off = b->word_off.data();
  *start = b->start.data();
  *end = b->end.data();
  return CTCDEC_OK;
}
int ctcdec_result_lm_state(const ctcdec_result* r, int32_t utt, int32_t beam, ctcdec_lm_state* out) {
  const BeamResult* b = get_beam(r, utt, beam);
  if (!b) return fail(CTCDEC_ERR_ARG, "no such beam");
  *out = b->state;
  return CTCDEC_OK;
}
int ctcdec_result_lm_state_of(const ctcdec_result* r, int32_t utt, int32_t beam, int32_t k, ctcdec_lm_state* out) {
  const BeamResult* b = get_beam(r, utt, beam);
  if (!b || !out) return fail(CTCDEC_ERR_ARG, "no such beam");
  if (k == 0) {
    *out = b->state;
    return CTCDEC_OK;
  }
  if (k < 0 || (size_t)k > b->xstates.size()) return fail(CTCDEC_ERR_ARG, "no such language model");
  *out = b->xstates[(size_t)k - 1];
  return CTCDEC_OK;
}
// decode_batch only wants the texts: one blob + offsets, nothing else is touched (packing the word frames of a
// 4096-utterance batch costs more than copying its results back from the device)
int ctcdec_result_texts(ctcdec_result* r, const char** blob_out, const int64_t** off_out, int64_t* n_out) {
  if (!r || !blob_out || !off_out || !n_out) return fail(CTCDEC_ERR_ARG, "no result");
  materialise(r);
  if (!r->texts_packed) {
    size_t nb = 0, bytes = 0;
    for (const auto& beams : r->utts)
      for (const BeamResult& b : beams) {
        ++nb;
        bytes += b.text.size();
      }
    r->t_blob.clear();
    r->t_blob.reserve(bytes);
    r->t_off.clear();
    r->t_off.reserve(nb + 1);
    r->t_off.push_back(0);
    for (const auto& beams : r->utts)
      for (const BeamResult& b : beams) {
        r->t_blob += b.text;
        r->t_off.push_back((int64_t)r->t_blob.size());
      }
    r->texts_packed = true;
  }
  *blob_out = r->t_blob.data();
  *off_out = r->t_off.data();
  *n_out = (int64_t)r->t_off.size() - 1;
  return CTCDEC_OK;
}

int ctcdec_result_texts_joined(ctcdec_result* r, char sep, const char** blob_out, int64_t* bytes_out, int64_t* n_out) {
  if (!r || !blob_out || !bytes_out || !n_out) return fail(CTCDEC_ERR_ARG, "no result");
  if (r->device_texts && !r->dev_out.empty()) {  // straight from the blocks the device wrote
    r->j_blob.clear();
    r->j_blob.reserve(r->dev_texts.size() + r->dev_out.size());
    for (size_t u = 0; u < r->dev_out.size(); ++u) {
      if (u) r->j_blob += sep;
      r->j_blob.append(r->dev_texts, r->dev_out[u].tok_off, r->dev_out[u].tok_cnt);
    }
    *blob_out = r->j_blob.data();
    *bytes_out = (int64_t)r->j_blob.size();
    *n_out = (int64_t)r->dev_out.size();
    return CTCDEC_OK;
  }
  size_t nb = 0, bytes = 0;
  for (const auto& beams : r->utts)
    for (const BeamResult& b : beams) {
      ++nb;
      bytes += b.text.size() + 1;
    }
  r->j_blob.clear();
  r->j_blob.reserve(bytes);
  size_t k = 0;
  for (const auto& beams : r->utts)
    for (const BeamResult& b : beams) {
      if (k++) r->j_blob += sep;
      r->j_blob += b.text;
    }
  *blob_out = r->j_blob.data();
  *bytes_out = (int64_t)r->j_blob.size();
  *n_out = (int64_t)nb;
  return CTCDEC_OK;
}

int ctcdec_result_text_blocks(ctcdec_result* r, const char** pool_out, const int64_t** off_out, const int64_t** len_out,
                              int64_t* n_out) {
  if (!r || !pool_out || !off_out || !len_out || !n_out) return fail(CTCDEC_ERR_ARG, "no result");
  if (r->device_texts && !r->dev_out.empty()) {  // the blocks the device wrote, as they are
    if (r->blk_off.empty()) {
      r->blk_off.reserve(r->dev_out.size());
      r->blk_len.reserve(r->dev_out.size());
      for (const OutBeam& ob : r->dev_out) {
        r->blk_off.push_back((int64_t)ob.tok_off);
        r->blk_len.push_back((int64_t)ob.tok_cnt);
      }
    }
    *pool_out = r->dev_texts.data();
  } else {  // any other result: the packed texts
    const char* blob = nullptr;
    const int64_t* off = nullptr;
    int64_t n = 0;
    int rc = ctcdec_result_texts(r, &blob, &off, &n);
    if (rc != CTCDEC_OK) return rc;
    if (r->blk_off.empty() && n > 0) {
      r->blk_off.assign(off, off + n);
      for (int64_t i = 0; i < n; ++i) r->blk_len.push_back(off[i + 1] - off[i]);
    }
    *pool_out = blob;
  }
  *off_out = r->blk_off.data();
  *len_out = r->blk_len.data();
  *n_out = (int64_t)r->blk_off.size();
  return CTCDEC_OK;
}

int ctcdec_result_pack(ctcdec_result* r, ctcdec_packed* out) {
  if (!r || !out) return fail(CTCDEC_ERR_ARG, "no result");
  materialise(r);
  if (!r->packed) {
    size_t nb = 0, nw = 0, tb = 0;
    for (const auto& beams : r->utts)
      for (const BeamResult& b : beams) {
        ++nb;
        nw += b.start.size();
        tb += b.text.size();
      }
    r->text_blob.reserve(tb);
    r->text_off.reserve(nb + 1);
    r->word_cnt_off.reserve(nb + 1);
    r->partial_off.reserve(nb + 1);
    r->logit.reserve(nb);
    r->lm.reserve(nb);
    r->states.reserve(nb);
    r->src_beam.reserve(nb);
    r->last_char.reserve(nb);
    r->pstart.reserve(nb);
    r->pend.reserve(nb);
    r->raw_lm.reserve(nb);
    r->word_byte_off.reserve(nw);
    r->word_start.reserve(nw);
    r->word_end.reserve(nw);
    r->beam_off.assign(1, 0);
    r->text_off.assign(1, 0);
    r->word_cnt_off.assign(1, 0);
    r->partial_off.assign(1, 0);
    for (const auto& beams : r->utts) {
      for (const BeamResult& b : beams) {
        r->text_blob += b.text;
        r->text_off.push_back((int64_t)r->text_blob.size());
        r->logit.push_back(b.logit);
        r->lm.push_back(b.lm);
        r->states.push_back(b.state);
        r->partial_blob += b.partial;
        r->partial_off.push_back((int64_t)r->partial_blob.size());
        r->src_beam.push_back(b.src);
        r->last_char.push_back(b.last_char);
        r->pstart.push_back(b.pstart);
        r->pend.push_back(b.pend);
        r->raw_lm.push_back(b.raw_lm);
        for (size_t k = 0; k < b.start.size(); ++k) {
          r->word_byte_off.push_back(b.word_off[k]);
          r->word_start.push_back(b.start[k]);
          r->word_end.push_back(b.end[k]);
        }
        r->word_cnt_off.push_back((int64_t)r->word_start.size());
      }
      r->beam_off.push_back((int64_t)r->logit.size());
    }
    r->packed = true;
  }
  out->n_utts = (int64_t)r->utts.size();
  out->n_beams = (int64_t)r->logit.size();
  out->n_words = (int64_t)r->word_start.size();
  out->beam_off = r->beam_off.data();
  out->text_blob = r->text_blob.data();
  out->text_off = r->text_off.data();
  out->logit_score = r->logit.data();
  out->lm_score = r->lm.data();
  out->word_cnt_off = r->word_cnt_off.data();
  out->word_byte_off = r->word_byte_off.data();
  out->word_start = r->word_start.data();
  out->word_end = r->word_end.data();
  out->lm_state = r->states.data();
  out->partial_blob = r->partial_blob.data();
  out->partial_off = r->partial_off.data();
  out->src_beam = r->src_beam.data();
  out->last_char = r->last_char.data();
  out->partial_start = r->pstart.data();
  out->partial_end = r->pend.data();
  out->raw_lm_score = r->raw_lm.data();
  return CTCDEC_OK;
}

int ctcdec_profile_phases(ctcdec_decoder* dec, int32_t enable, uint64_t* ticks_out, int32_t n) {
  if (!dec) return fail(CTCDEC_ERR_ARG, "bad arguments");
  dec->profile = enable != 0;
  if (ticks_out)
    for (int k = 0; k < n && k < N_PROF; ++k) ticks_out[k] = dec->prof[k];
  return CTCDEC_OK;
}

int ctcdec_result_timing(const ctcdec_result* r, double* ms3) {
  if (!r) return fail(CTCDEC_ERR_ARG, "no result");
  ms3[0] = r->ms[0];
  ms3[1] = r->ms[1];
  ms3[2] = r->ms[2];
  return CTCDEC_OK;
}
int ctcdec_result_beam_kernel(const ctcdec_result* r) { return r ? r->beam_kernel : 0; }
int ctcdec_device(void) { return be::current_device(); }
// Tearing down the per-beam strings and vectors of a large batch takes about as long as copying the results back from
// the device did: large results are handed to ONE reclaimer thread (started on first use, joined when the library is
// unloaded or the process exits, so that no free can race static destruction).
namespace {
class Reclaimer {
 public:
  ~Reclaimer() {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
    }
    cv_.notify_all();
    if (worker_.joinable()) worker_.join();
    for (ctcdec_result* r : queue_) delete r;
  }
  bool hand_over(ctcdec_result* r) {
    try {
      std::lock_guard<std::mutex> g(m_);
      if (stop_ || queue_.size() >= 64) return false;  // (a caller that frees faster than we reclaim: do it inline)
      if (!worker_.joinable()) worker_ = std::thread([this] { loop(); });
      queue_.push_back(r);
    } catch (...) {  // no thread / no memory: the caller frees inline (nothing may escape an extern "C" function)
      return false;
    }
    cv_.notify_one();
    return true;
  }

 private:
  void loop() {
    for (;;) {
      ctcdec_result* r = nullptr;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return stop_ || !queue_.empty(); });
        if (queue_.empty()) return;  // (stop requested and nothing left)
        r = queue_.back();
        queue_.pop_back();
      }
      delete r;
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::vector<ctcdec_result*> queue_;
  std::thread worker_;
  bool stop_ = false;
};
Reclaimer g_reclaimer;
}  // namespace

void ctcdec_result_free(ctcdec_result* r) {
  if (!r) return;
  if (r->utts.size() >= 256 && g_reclaimer.hand_over(r)) return;
  delete r;
}

}  // extern "C"
