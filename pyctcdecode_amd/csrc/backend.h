// backend.h -- the narrow device interface api.cpp is written against.
//   * pyctcdecode_amd/csrc/backend_hip.hip : the product (MI355X / gfx950 HIP kernels)
//   * tests/sim/backend_sim.cpp            : sequential CPU execution of the same beam_core.h,
//                                            test infrastructure only (never built into the product)
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>

#include "beam_core.h"

namespace ctc {
namespace be {

const char* name();
int init(int device, std::string* err);
int bind_thread(std::string* err);  // make the decoder's device current for the calling host thread
int current_device();               // device index all decoders of this process run on (-1: none yet)
void* alloc(size_t bytes, std::string* err);
void release(void* p);
void* alloc_host(size_t bytes, std::string* err);  // page-locked staging memory for result copies
void release_host(void* p);
int h2d(void* dst, const void* src, size_t bytes, std::string* err);
int d2h(void* dst, const void* src, size_t bytes, std::string* err);
int d2d(void* dst, const void* src, size_t bytes, std::string* err);
int zero(void* dst, size_t bytes, std::string* err);
int sync(std::string* err);
// `height` rows of `width` bytes, host (pitch spitch) -> device (pitch dpitch), on the COPY stream (its own non-blocking stream:
// the kernels queued on the decode stream keep running while the host waits here). Returns when the copy has landed.
int h2d_2d_overlapped(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, std::string* err);
// streams and events for the chunked pipeline of large batches (api.cpp: decode_pipelined). Every call above and the
// launches below act on the current stream; stream 0 is the default.
void use_stream(int idx);
int n_events();
int ev_record(int id, std::string* err);
int ev_wait(int id, std::string* err);
int ev_sync(int id, std::string* err);
double ev_elapsed_ms(int a, int b);
int d2h_async(void* dst, const void* src, size_t bytes, std::string* err);
// host -> device on the current stream WITHOUT waiting: `src` is page-locked memory the caller leaves alone until the stream has passed it
int h2d_async(void* dst, const void* src, size_t bytes, std::string* err);
int sync_all(std::string* err);
int cus();  // compute units of the device
void set_last_timing(double prune_ms, double beam_ms);

// Frame-prune stage: input normalisation (decoder.py:759-765), token prune (decoder.py:444-445)
// and CPython-set ordering of the survivors, for every frame of every utterance.
struct PruneArgs {
  const void* const* utt_logits;  // [n_utts] device pointers to [T_u, V] matrices
  const int64_t* utt_row0;        // [n_utts + 1] prefix sums of T_u (device)
  int32_t n_utts;
  int64_t n_rows;
  int32_t n_labels;
  int32_t dtype;  // ctcdec_dtype
  double token_min_logp;
  int32_t max_surv;
  double* row_sum;       // [n_rows] scratch
  uint32_t* utt_is_prob; // [n_utts] out: 1 when the input looks like probabilities
  uint32_t* surv_cnt;    // [n_rows]
  uint16_t* surv_id;     // [n_rows * max_surv]
  double* surv_lp;       // [n_rows * max_surv]
  uint32_t* overflow;    // [8] ([4]: the survivors of all rows, summed -- utt_sniff; the caller decides small batches' beam kernel
                         //      by the mean per frame) [0]: a row had more than max_surv survivors, [1]: a probability-like utterance exists,
                         // [2]: an utterance is marked 2 in utt_is_prob (its rows sum to about 1: launch_sniff_exact decides)
                         // [3]: device-side counter of slow_rows
  int32_t pass;          // 0: all utterances as logits + row sums + sniff; 1: redo the probability-like ones
  int32_t rows_aligned16; // every utterance base pointer is 16-byte aligned
  int32_t rows_aligned4;  // ... 4-byte aligned (float32 rows read one label at a time)
  int64_t row_base;      // first row of this launch (utt_row0 and the row-indexed arrays are absolute); n_rows rows follow
  uint32_t* slow_rows;   // [n_rows] scratch, or nullptr: rows (relative to row_base) the 64-rows-per-wave kernel hands to the
                         // per-row one; their count is kept in overflow[3]
  uint32_t* utt_side;    // [n_utts] or nullptr (time-sliced host ingest, api.cpp): set to 3 when this launch's rows of the utterance
                         // were classified as probabilities
  double* utt_sum;       // [n_utts] or nullptr (likewise): the row sums of this launch's rows are added to it
  int32_t dense_hint;    // 1: the caller has seen most rows of such input overflow the 64-rows-per-wave kernel (small vocabulary, flat
                         // logits: every label survives) -- go straight to one wave per row
  // set by launch_prune itself:
  int32_t f32_np;        // float32 rows in the reference's own float32 arithmetic (np_f32.h + numpy's summation order): the default;
                         // 0 under CTCDEC_PRUNE_EXP=pk (round 5's packed polynomial, fp64 from there on) / =f64
  const uint8_t* np_prog;  // frame_prune_fast, f32_np: the pairwise tree over the leaf sums of one row as (dst, src) pairs
  const uint16_t* np_leaf; // ... and the leaves as (offset, length) pairs; np_n_leaf of them
  int32_t np_n_leaf;
  int32_t np_uniform8;     // 1, 2, 4 or 8 leaves of 128 labels each (V = 128 .. 1024 in powers of two): the tree is walked in registers
};
int launch_prune(const PruneArgs& a, std::string* err);
// decoder.py:760 in the input dtype and numpy's summation order for the utterances pass 0 marked ambiguous
int launch_sniff_exact(const PruneArgs& a, std::string* err);

// Beam stage: one workgroup per utterance.
struct BeamArgs {
  DeviceTables tables;  // device pointers
  DecodeParams params;
  int32_t n_utts;
  const int64_t* utt_row0;   // [n_utts + 1] (device)
  const uint32_t* surv_cnt;
  const uint16_t* surv_id;
  const double* surv_lp;
  TextNode* text_nodes;      // arena for all utterances
  EmitNode* emit_nodes;
  const uint64_t* text_off;  // [n_utts + 1] node offsets (device)
  const uint64_t* emit_off;  // [n_utts + 1]
  const LmState* start_states;  // [n_utts * max(1, tables.n_lms)] or nullptr
  LmState* out_xstates;         // several LMs: [n_utts * out_stride * (n_lms - 1)], else nullptr
  OutBeam* out;                 // [n_utts * out_stride]
  int32_t out_stride;
  uint32_t* n_out;              // [n_utts]
  uint32_t* status;             // [n_utts]
  EmitNode* tok_pool;
  unsigned long long* tok_pool_head;  // [1]
  unsigned long long tok_pool_cap;
  unsigned long long* prof;  // [N_PROF] phase cycle counters of utterance 0, or nullptr
  const ImportBeam* imports;   // streaming: all utterances' carried-over beams, or nullptr
  const LmState* import_xstates;  // several LMs: [total beams * (n_lms - 1)] their states of LM 1.., else nullptr
  const int64_t* import_off;   // [n_utts + 1] (device)
  const int32_t* first_frames; // [n_utts] processed_frames per utterance (device), or nullptr
  ColdRec* cold;               // [n_utts * 2 * COLD_STRIDE] scratch of the wave kernel
  PoolPay* pay;                // [n_utts * pay_stride] scratch of the wave kernel (nullptr: the decode is not eligible for it)
  uint64_t pay_stride;
  int32_t max_import;          // streaming: the largest number of beams any stream carries in
  // device-resident streams (ctcdec_stream_*), else nullptr / 0: where stream u's finalisation leaves its beams for the
  // next chunk (carry_out + u * carry_stride; several LMs: carry_xstates + u * carry_stride * (n_lms - 1)), its counters
  // (sstate + u: beams carried in, first free node of its emission arena)
  ImportBeam* carry_out;
  LmState* carry_xstates;
  StreamState* sstate;
  int32_t carry_stride;
  int32_t want_out;            // 0: no output records / emission lists for this launch (resident streams between reads)
  // texts assembled on the device (params.texts_only), else nullptr: utterance u writes its text backwards from the end of
  // text_scratch[text_soff[u] .. text_soff[u + 1]) and copies it to a block of text_pool taken from tok_pool_head[1]
  uint8_t* text_scratch;
  const uint64_t* text_soff;   // [n_utts + 1] (device)
  uint8_t* text_pool;
  unsigned long long text_pool_cap;
  const int32_t* order;        // [n_utts] (device) or nullptr: workgroup b decodes utterance order[b] -- longest first, so
                               // that a ragged batch that needs several rounds of resident waves ends evenly (the hardware
                               // hands the next workgroup to the first free slot: longest-processing-time-first scheduling)
  int32_t resident_in;         // 1: `imports` is the carry buffer itself (stream u: imports + u * carry_stride, sstate[u].n_carry
                               // beams; import_xstates likewise), import_off is not used
  int32_t surv_x16;            // 16 x the mean number of survivors per frame of this launch's rows, as the prune stage counted
                               // them (0: not known). Small batches choose their kernel by it (wave_kernel_chosen).
  int64_t total_rows;          // frames of all utterances of this launch
  // wave kernel, set by launch_beam itself:
  unsigned long long* wave_clock;  // diagnostics (CTCDEC_WAVE_TIMES=<file>), else nullptr: [n_utts * 4] per workgroup {start, end of
                               // its wave (100 MHz real-time counter), HW_ID | XCC_ID << 32, frames}
  int32_t prio_mode;           // issue priority of a wave among the waves of its SIMD (beam_wave_hip.hip, WaveGpuCtx::frame_done):
                               // 0 left alone; 1 + k: rotated every 2^k frames; 32: by the frames it still has to decode against
                               // the launch's average (`progress`); 33: likewise, each remaining frame weighed by what a frame of
                               // this utterance is expected to cost (`block_weight`: survivors per frame against the launch's mean)
  unsigned long long* progress;  // [1] frames decoded so far by all waves of the launch (prio_mode 32; zeroed by launch_beam)
  unsigned long long total_frames;
  float inv_n_utts;
  const float* block_weight;   // [n_utts] (prio_mode 33) expected cost of a frame of the utterance workgroup b decodes, relative to
                               // the launch's mean (utt_weigh / utt_place, backend_hip.hip); the same kernels write `order`
};
int launch_beam(const BeamArgs& a, std::string* err);
// Will launch_beam run the wave kernel on these arguments (given payload lines)? THE kernel-selection rule, shared by the
// launcher and by the caller that reserves the wave kernel's scratch: eligibility, the carried-in beams, the batch-size
// rule and the CTCDEC_BEAM_KERNEL override. `a.pay` itself is not looked at.
bool wave_kernel_chosen(const BeamArgs& a);
// Is this batch small enough for the choice to depend on the input (then the caller reads the prune stage's survivor count
// before it launches the beam stage -- one small read-back between the two stages)?
bool beam_kernel_depends_on_input(const BeamArgs& a);

// stage timing (ms) of the last launch_prune / launch_beam pair, measured on the decode stream
void last_timing(double* prune_ms, double* beam_ms);
// which beam kernel the last launch_beam used: 1 = one wave per utterance (beam_wave.h), 2 = one workgroup
// per utterance (beam_core.h), 0 = none yet
int last_beam_kernel();

}  // namespace be
}  // namespace ctc
