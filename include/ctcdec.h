/*
 * ctcdec.h -- C ABI of libctcdec.so, the MI355X-native CTC prefix-beam-search decoder.
 *
 * The reference (kensho-technologies/pyctcdecode) is pure Python and has NO FFI: its boundary for
 * this path is the Python surface BeamSearchDecoderCTC.decode / decode_beams / decode_batch /
 * decode_beams_batch (pyctcdecode/decoder.py:730-945) built by build_ctcdecoder
 * (decoder.py:1051-1099).  This header is the C ABI a maintainer would bind underneath that
 * surface (ctypes stub: INTEGRATION.md); every entry point cites the reference code it replaces.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success and a
 * negative ctcdec_status otherwise (never throws); ctcdec_last_error() returns a thread-local
 * message.  The caller owns all input buffers; the library owns result objects until
 * ctcdec_result_free().  One decoder handle may be used from one host thread at a time.
 */
#ifndef CTCDEC_H
#define CTCDEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ctcdec_decoder ctcdec_decoder; /* opaque */
typedef struct ctcdec_result ctcdec_result;   /* opaque */

enum ctcdec_status {
  CTCDEC_OK = 0,
  CTCDEC_ERR_ARG = -1,     /* bad argument (ValueError on the Python side)            */
  CTCDEC_ERR_IO = -2,      /* cannot read / parse the LM file                          */
  CTCDEC_ERR_DEVICE = -3,  /* HIP runtime failure or no gfx950 device                  */
  CTCDEC_ERR_LIMIT = -4,   /* request exceeds a documented limit (beam width, order..) */
  CTCDEC_ERR_INTERNAL = -5
};

enum ctcdec_dtype { CTCDEC_F32 = 0, CTCDEC_F64 = 1, CTCDEC_F16 = 2, CTCDEC_BF16 = 3 };  /* host or device pointers alike (host float16 matrices are staged as they are) */

/* Maximum supported values (checked; CTCDEC_ERR_LIMIT otherwise). */
#define CTCDEC_MAX_BEAM_WIDTH 256
#define CTCDEC_MAX_LM_ORDER 6
#define CTCDEC_MAX_VOCAB 65535
#define CTCDEC_MAX_LMS 4 /* language models in one MultiLanguageModel */

/* Decode-time parameters: the keyword arguments of decode_beams (decoder.py:730-740) plus the
 * LM parameters that reset_params may change between calls (decoder.py:292-313,
 * language_model.py:271-301), which is why they are per-call kernel arguments. */
typedef struct ctcdec_params {
  int32_t beam_width;        /* DEFAULT_BEAM_WIDTH 100       constants.py:8  */
  int32_t prune_history;     /* 0/1; decode() forces 1       decoder.py:888  */
  int32_t n_best;            /* beams to return per utterance (<=0: all, 1 for decode()) */
  int32_t want_lm_state;     /* 0/1: also return last_lm_state per beam      */
  double beam_prune_logp;    /* DEFAULT_PRUNE_LOGP -10       constants.py:10 */
  double token_min_logp;     /* DEFAULT_MIN_TOKEN_LOGP -5    constants.py:12 */
  double hotword_weight;     /* DEFAULT_HOTWORD_WEIGHT 10    constants.py:9  */
  double alpha;              /* language_model.py:266        */
  double beta;               /* language_model.py:267        */
  double unk_score_offset;   /* language_model.py:268        */
  double log_base_change;    /* LOG_BASE_CHANGE_FACTOR       constants.py:18 */
  int32_t lm_score_boundary; /* language_model.py:269        */
  int32_t first_frame;       /* processed_frames offset      decoder.py:443  */
  int32_t texts_only;        /* 0/1 (ctcdec_decode_batch with n_best = 1): only the best beam's TEXT is wanted
                                (decode_batch, decoder.py:895-945) -- it is assembled on the device and the result holds
                                one beam per utterance with its text and scores, no word frames and no LM state */
  int32_t reserved;
} ctcdec_params;

/* LM start state for one utterance (decode_beams(lm_start_state=...), decoder.py:621-625):
 * context words newest first, as vocabulary indices of THIS decoder's LM, with the back-off
 * weight of each context length. length == -1 means "use the LM's own start state". */
typedef struct ctcdec_lm_state {
  int32_t length;
  uint32_t words[CTCDEC_MAX_LM_ORDER - 1];
  float backoff[CTCDEC_MAX_LM_ORDER - 1];
} ctcdec_lm_state;

/* ---- construction: replaces BeamSearchDecoderCTC.__init__ / build_ctcdecoder ------------------
 * labels_blob/labels_off: the NORMALISED alphabet (Alphabet.labels, alphabet.py:139-148) as one
 * UTF-8 blob with n_labels+1 byte offsets; is_bpe = Alphabet.is_bpe.  device: HIP device index. */
int ctcdec_create(const char* labels_blob, const int64_t* labels_off, int32_t n_labels,
                  int32_t is_bpe, int32_t device, ctcdec_decoder** out);
void ctcdec_destroy(ctcdec_decoder* dec);

/* Replaces kenlm.Model(path) (decoder.py:1074): parse an ARPA file into the flat hashed n-gram
 * trie and upload it.  order_out receives the n-gram order (LanguageModel.order). */
int ctcdec_lm_load_arpa(ctcdec_decoder* dec, const char* path, int32_t* order_out);

/* The parsed model as one flat file (vocabulary, unigram array and the hashed n-gram table in upload
 * layout): what a kenlm binary is to an ARPA file (language_model.py:424 accepts .bin/.binary next to
 * .arpa) -- loading is a few reads instead of a parse.  (kenlm's own probing binaries: ctcdec_lm_load_kenlm below.) */
int ctcdec_lm_save_flat(const ctcdec_decoder* dec, const char* path);
int ctcdec_lm_load_flat(ctcdec_decoder* dec, const char* path, int32_t* order_out);

/* Replaces kenlm.Model("x.bin") (decoder.py:1074; language_model.py:424 accepts .bin / .binary): a kenlm PROBING binary
 * (`build_binary probing`) -- header and sanity block, vocabulary strings checked against the vocabulary hash table, the
 * unigram array and the per-order probing tables adopted entry by entry into the flat trie (which uses kenlm's own n-gram key
 * chain for that reason).  Trie / quantised / array-compressed / rest-cost models are refused by name (CTCDEC_ERR_IO).
 * FORMAT UNPINNED AGAINST REAL KENLM: restated from the published sources, pinned against ctcdec_arpa_to_kenlm_binary only
 * (csrc/kenlm_binary.cpp).  ctcdec_is_kenlm_binary: 1 when the file starts with kenlm's magic bytes. */
int ctcdec_lm_load_kenlm(ctcdec_decoder* dec, const char* path, int32_t* order_out);
int ctcdec_is_kenlm_binary(const char* path);
/* ARPA -> kenlm probing binary (what `build_binary probing x.arpa x.bin` writes, as far as kenlm's sources say): a converter
 * and the reader's test vector.  probing_multiplier <= 1: kenlm's default 1.5. */
int ctcdec_arpa_to_kenlm_binary(const char* arpa_path, const char* out_path, float probing_multiplier);

/* Replaces LanguageModel.__init__'s unigram handling (language_model.py:257-265, :87-103):
 * unigrams given as a UTF-8 blob + offsets; has_unigrams=0 means "unigrams is None" (no trie).
 * Words not in the LM vocabulary are dropped (language_model.py:95).  n_kept_out: |unigram_set|. */
int ctcdec_lm_set_unigrams(ctcdec_decoder* dec, int32_t has_unigrams, const char* blob,
                           const int64_t* off, int64_t n_unigrams, int64_t* n_kept_out);

/* Let `dst` use the language model already loaded into `src` (shared, reference counted): a
 * LanguageModel object constructed on its own (language_model.py:237-269) and later handed to
 * BeamSearchDecoderCTC(alphabet, language_model) (decoder.py:275-290) is parsed only once. */
int ctcdec_lm_share(ctcdec_decoder* dst, const ctcdec_decoder* src);

/* Give `dst` a private COPY of the language model loaded into `src`. The reference builds many LanguageModels with
 * different unigram sets on one immutable kenlm.Model (tests/test_decoder.py:188-280); here the unigram set lives
 * in the model's prefix table, so every LanguageModel after the first gets its own copy instead of rewriting the
 * tables under the decoders that share the first one. */
int ctcdec_lm_clone(ctcdec_decoder* dst, const ctcdec_decoder* src);

/* Replaces MultiLanguageModel (language_model.py:455-502): `dst` scores every word with the n
 * (2..CTCDEC_MAX_LMS) language models loaded into srcs[0..n) -- each from its own state, the scores
 * averaged (language_model.py:483-502), partial words by the mean of the models' unigram-trie scores
 * (:477-481), history pruning by the largest order (:467-469). Model 0 takes alpha / beta /
 * unk_score_offset / lm_score_boundary from ctcdec_params like a single model; models 1.. take theirs
 * from ctcdec_lm_set_params (each reference LanguageModel carries its own, language_model.py:266-269).
 * With several models every "LM state" of the decode calls is n consecutive ctcdec_lm_state entries
 * in model order: start_states holds n per utterance, ctcdec_result_lm_state_of reads model k's; a streaming
 * beam (ctcdec_decode_stream_batch) carries the states of models 1.. in ctcdec_beam_in.more_states. */
int ctcdec_lm_share_multi(ctcdec_decoder* dst, const ctcdec_decoder* const* srcs, int32_t n);
int ctcdec_lm_set_params(ctcdec_decoder* dec, int32_t k /* 1..n-1 */, double alpha, double beta,
                         double unk_score_offset, int32_t lm_score_boundary);
/* number of language models behind the decoder (0: none) */
int ctcdec_lm_count(const ctcdec_decoder* dec, int32_t* n_out);

/* Unigram char-trie query: CharTrie.has_node (language_model.py:331) and set membership
 * (language_model.py:351).  flags_out: bit0 prefix of a unigram, bit1 LM vocabulary word,
 * bit2 member of the unigram set. */
int ctcdec_lm_prefix_flags(const ctcdec_decoder* dec, const char* str_utf8, int64_t len,
                           uint32_t* flags_out);

/* kenlm vocabulary queries used by the Python shell: Model.__contains__ (language_model.py:95,
 * 352) and word index / word string for exporting and importing LM states. */
int ctcdec_lm_word_index(const ctcdec_decoder* dec, const char* word_utf8, int64_t len,
                         uint32_t* index_out);
int ctcdec_lm_word_string(const ctcdec_decoder* dec, uint32_t index, const char** str_out,
                          int64_t* len_out);

/* Host-side single-word query = kenlm.Model.BaseScore (language_model.py:347) and the start
 * states BeginSentenceWrite / NullContextWrite (language_model.py:311-314).  Used by the public
 * LanguageModel.score()/get_start_state() methods, NOT by the decode path (which runs on device). */
int ctcdec_lm_start_state(const ctcdec_decoder* dec, int32_t begin_sentence, ctcdec_lm_state* out);
int ctcdec_lm_base_score(const ctcdec_decoder* dec, const ctcdec_lm_state* in, uint32_t word_index,
                         ctcdec_lm_state* out, float* log10_prob_out);

/* Replaces HotwordScorer.build_scorer (language_model.py:152-189) for the next decode calls:
 * hot-word UNIGRAMS (already stripped and split by the caller) as UTF-8 blob + offsets. */
int ctcdec_set_hotwords(ctcdec_decoder* dec, const char* blob, const int64_t* off, int64_t n_words);

/* ---- the hot path: replaces decode_beams / decode_batch / decode_beams_batch ------------------
 * utt_logits[i] points to a row-major [utt_frames[i], n_labels] matrix of `dtype`; pointers may
 * be device (HBM-resident, no copy) or host memory (copied H2D first), is_device tells which.
 * start_states may be NULL. The call returns when results are on the host. */
int ctcdec_decode_batch(ctcdec_decoder* dec, const void* const* utt_logits,
                        const int32_t* utt_frames, int32_t n_utts, int32_t dtype, int32_t is_device,
                        const ctcdec_params* params, const ctcdec_lm_state* start_states,
                        ctcdec_result** out);

/* ---- streaming: replaces partial_decode_beams (decoder.py:681-728) for a batch of streams -------
 * Every stream hands back the beams the previous call returned (rank order). Strings travel as
 * byte ranges of text_blob: `text` = completed words separated by single spaces, `partial` = the
 * open partial_word. raw_lm_score / lm_state are the memo values of the text
 * (cached_lm_scores[(text, False)], decoder.py:121-126); (0, LM start state) for the empty text.
 * force_next_word / is_end as in decoder.py:693-694. Results: the packed view incl. its streaming
 * extras; word frames there are only the words closed during THIS call. */
typedef struct ctcdec_beam_in {
  double logit_score;
  double raw_lm_score;
  ctcdec_lm_state lm_state;
  int32_t last_char;          /* label index, -1 = None */
  int32_t partial_start;      /* partial_frames */
  int32_t partial_end_frame;
  int32_t reserved;
  int64_t text_begin, text_end;        /* byte range in text_blob */
  int64_t partial_begin, partial_end;  /* byte range in text_blob */
  const ctcdec_lm_state* more_states;  /* several language models: the states of model 1.. (n-1 entries,
                                          lm_state above being model 0's); NULL with a single model */
} ctcdec_beam_in;
int ctcdec_decode_stream_batch(ctcdec_decoder* dec, const void* const* utt_logits,
                               const int32_t* utt_frames, int32_t n_streams, int32_t dtype,
                               int32_t is_device, const ctcdec_params* params,
                               const int32_t* first_frame /* [n_streams] processed_frames */,
                               const ctcdec_beam_in* beams, const int64_t* beam_off /* [n_streams+1] */,
                               const char* text_blob, int32_t force_next_word, int32_t is_end,
                               ctcdec_result** out);

/* ---- device-resident streams ---------------------------------------------------------------------
 * The same recursion as ctcdec_decode_stream_batch (get_starting_state / partial_decode_beams,
 * decoder.py:669-728), but what the reference's caller carries between chunks -- the live beams, their LM states and
 * memo values, the words and frames decoded so far -- stays on the device: a chunk costs two kernel launches and a
 * 16-byte read-back per stream, and beams are only materialised when somebody asks for them.
 *   open   n independent streams, each at the reference's starting state (start_states: n * max(1, n_lms) LM
 *          states or NULL for the models' own defaults)
 *   push   one chunk per stream (chunk_frames[u] may be 0). first_frame: processed_frames per stream, NULL = the frames
 *          pushed so far. want_result != 0 (always the case for is_end): *out receives the ranked beams exactly as
 *          ctcdec_decode_stream_batch would return them -- texts / word frames from the START of the stream, src_beam = -1
 *          (or, below a ctcdec_stream_import, relative to the imported beam src_beam). After is_end the streams are back
 *          at the starting state.
 *   read   the current beams without advancing (a push of zero frames)
 *   import replace the carried beams of every stream by beams the caller built or edited (the slow path: the host
 *          resolves their strings like ctcdec_decode_stream_batch does). The arrays must stay untouched until close or the
 *          next import (they are copied).
 * One handle belongs to one decoder; its calls are serialised with that decoder's other calls. */
typedef struct ctcdec_stream ctcdec_stream;
int ctcdec_stream_open(ctcdec_decoder* dec, int32_t n_streams, const ctcdec_lm_state* start_states, ctcdec_stream** out);
int ctcdec_stream_push(ctcdec_stream* st, const void* const* chunk_logits, const int32_t* chunk_frames, int32_t dtype,
                       int32_t is_device, const ctcdec_params* params, const int32_t* first_frame,
                       int32_t force_next_word, int32_t is_end, int32_t want_result, ctcdec_result** out);
int ctcdec_stream_read(ctcdec_stream* st, const ctcdec_params* params, ctcdec_result** out);
int ctcdec_stream_import(ctcdec_stream* st, const ctcdec_beam_in* beams, const int64_t* beam_off /* [n_streams+1] */,
                         const char* text_blob, int64_t text_bytes);
/* frames pushed so far per stream ([n_streams]) */
int ctcdec_stream_frames(const ctcdec_stream* st, int64_t* frames_out);
void ctcdec_stream_close(ctcdec_stream* st);

/* ---- results (OutputBeam fields, decoder.py:102-110, assembled as decoder.py:653-667) --------- */
int32_t ctcdec_result_num_utts(const ctcdec_result* r);
int32_t ctcdec_result_num_beams(const ctcdec_result* r, int32_t utt);
/* text of beam b of utterance u (UTF-8, not NUL-terminated) */
int ctcdec_result_text(const ctcdec_result* r, int32_t utt, int32_t beam, const char** str_out,
                       int64_t* len_out);
int ctcdec_result_scores(const ctcdec_result* r, int32_t utt, int32_t beam, double* logit_score,
                         double* lm_score);
/* word frames: n words; word k spans bytes [word_off[k], word_off[k+1]) of the text and frames
 * [start[k], end[k]). Pointers stay valid until ctcdec_result_free. */
int ctcdec_result_frames(const ctcdec_result* r, int32_t utt, int32_t beam, int32_t* n_words,
                         const int32_t** word_off, const int32_t** start, const int32_t** end);
int ctcdec_result_lm_state(const ctcdec_result* r, int32_t utt, int32_t beam, ctcdec_lm_state* out);
/* several language models: the state of model k (k = 0 is ctcdec_result_lm_state) */
int ctcdec_result_lm_state_of(const ctcdec_result* r, int32_t utt, int32_t beam, int32_t k,
                              ctcdec_lm_state* out);
/* Bulk view of a whole result (one call instead of four per beam): beams of utterance u are
 * [beam_off[u], beam_off[u+1]); beam k's text is text_blob[text_off[k] .. text_off[k+1]); its words
 * are [word_cnt_off[k], word_cnt_off[k+1]) in word_byte_off / word_start / word_end, where
 * word_byte_off is relative to the beam's text and word j ends where word j+1 starts minus one
 * space (or at the end of the text). Pointers stay valid until ctcdec_result_free. */
typedef struct ctcdec_packed {
  int64_t n_utts, n_beams, n_words;
  const int64_t* beam_off;      /* [n_utts + 1]  */
  const char* text_blob;
  const int64_t* text_off;      /* [n_beams + 1] */
  const double* logit_score;    /* [n_beams]     */
  const double* lm_score;       /* [n_beams]     */
  const int64_t* word_cnt_off;  /* [n_beams + 1] */
  const int32_t* word_byte_off; /* [n_words]     */
  const int32_t* word_start;    /* [n_words]     */
  const int32_t* word_end;      /* [n_words]     */
  const ctcdec_lm_state* lm_state; /* [n_beams]  */
  /* streaming extras (LMBeam fields of partial_decode_beams, decoder.py:69-99) */
  const char* partial_blob;        /* still open partial_word of every beam */
  const int64_t* partial_off;      /* [n_beams + 1] */
  const int32_t* src_beam;         /* [n_beams] index of the carried-in beam this beam descends from */
  const int32_t* last_char;        /* [n_beams] label index, -1 = None */
  const int32_t* partial_start;    /* [n_beams] partial_frames */
  const int32_t* partial_end;      /* [n_beams] */
  const double* raw_lm_score;      /* [n_beams] LM score sum of the beam's text (memo value) */
} ctcdec_packed;
int ctcdec_result_pack(ctcdec_result* r, ctcdec_packed* out);
/* Only the texts of all beams (utterance-major, the order of ctcdec_result_pack): UTF-8 blob + n+1 byte offsets. What
 * decode_batch needs (decoder.py:895-945) without packing word frames and states. Owned by the result. */
int ctcdec_result_texts(ctcdec_result* r, const char** blob_out, const int64_t** off_out, int64_t* n_out);
/* The same texts as ONE buffer with `sep` between consecutive texts (n - 1 separators, none at the end): a caller whose
 * language splits strings natively (Python: str.split) gets its n string objects in one call instead of n slices.
 * The caller picks a byte that cannot occur in a text (a label containing it rules this entry point out). */
int ctcdec_result_texts_joined(ctcdec_result* r, char sep, const char** blob_out, int64_t* bytes_out, int64_t* n_out);
/* ... or without any copy: text i is pool[off[i] .. off[i] + len[i]) (in whatever order the pool happens to hold them).
 * Pointers stay valid until ctcdec_result_free. */
int ctcdec_result_text_blocks(ctcdec_result* r, const char** pool_out, const int64_t** off_out, const int64_t** len_out,
                              int64_t* n_out);

/* timing of the last call's device stages in milliseconds (HIP events on the decode stream):
 * [0] frame-prune kernel, [1] beam kernel, [2] total device time incl. result copy */
int ctcdec_result_timing(const ctcdec_result* r, double* ms3);
/* which beam kernel produced the result: 1 = one wavefront per utterance (csrc/beam_wave.h: one language model
 * or none, beam_width <= 128, at most 480 survivors per frame), 2 = one workgroup per utterance
 * (csrc/beam_core.h: everything else), 0 = empty batch. The environment variable CTCDEC_BEAM_KERNEL=group|wave
 * forces one of them (tests and tuning). */
int ctcdec_result_beam_kernel(const ctcdec_result* r);
/* HIP device index every decoder of this process runs on (-1 before the first ctcdec_create). One process drives one
 * GPU: a ctcdec_create for a different device is refused (CTCDEC_ERR_DEVICE). */
int ctcdec_device(void);
void ctcdec_result_free(ctcdec_result* r);

/* Diagnostics: the frame-prune stage alone on one [n_frames, V] matrix -- per frame the labels
 * `set(np.where(logp >= token_min_logp)[0]) | {argmax}` in CPython's set ITERATION order (the order the
 * reference walks them in, decoder.py:444-447) with their log-probabilities. counts[t] entries of row t are
 * written to ids / logps at t*stride (at most `stride` of them). Lets a test pin the order emulation on
 * CPython itself. */
int ctcdec_frame_survivors(ctcdec_decoder* dec, const void* logits, int32_t n_frames, int32_t dtype,
                           int32_t is_device, double token_min_logp, int32_t stride, int32_t* counts,
                           int32_t* ids, double* logps);

/* Diagnostics: enable/disable per-phase tick accumulation (100 MHz wall clock) for utterance 0 of the
 * following decode calls and read the 24 counters of the last one. Workgroup kernel: 0 load, 1 modes,
 * 2 completions, 3 keys, 4 merge, 5 score, 6 clear, 7 sort, 8 rebuild, 9 rest, 10 finalise (11.. sub-phases).
 * Wave kernel: 0 load + modes, 1 completions, 2 candidate keys, 3 match, 4 fold, 5 score, 6 rank, 7 rebuild,
 * 8 finalise, 9 pool compaction. No reference analogue. */
int ctcdec_profile_phases(ctcdec_decoder* dec, int32_t enable, uint64_t* ticks_out, int32_t n);

const char* ctcdec_last_error(void);
const char* ctcdec_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CTCDEC_H */
