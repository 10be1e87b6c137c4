// backend_sim.cpp -- TEST INFRASTRUCTURE ONLY.  A sequential CPU implementation of backend.h that
// runs the very same beam_core.h / set_order.h source the HIP kernels compile, one "thread" per
// workgroup.  It exists so the beam logic can be debugged against the oracle in a container without
// a GPU.  It is built into tests/_build/libctcdec_sim.so by tests/sim/build_sim.py, is never
// linked into pyctcdecode_amd/libctcdec.so and the product never loads it.
#include <math.h>
#include <cmath>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../pyctcdecode_amd/csrc/backend.h"
#include "../../pyctcdecode_amd/csrc/beam_core.h"
#include "../../pyctcdecode_amd/csrc/beam_wave.h"
#include "../../pyctcdecode_amd/csrc/text_wave.h"
#include "wave_fibers.h"
#include "group_fibers.h"
#include "../../pyctcdecode_amd/csrc/set_order.h"
#include "../../pyctcdecode_amd/csrc/np_sum.h"
#include "../../pyctcdecode_amd/csrc/np_f32.h"

namespace ctc {
namespace be {

struct SeqCtx {
  int tid = 0, nt = 1;
  void sync() {}
  void sync_mem() {}
  uint32_t atomic_add(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
  void atomic_or(uint32_t* p, uint32_t v) { *p |= v; }
  void atomic_min(uint32_t* p, uint32_t v) { if (v < *p) *p = v; }
  void atomic_max(uint32_t* p, uint32_t v) { if (v > *p) *p = v; }
  void atomic_max64(uint64_t* p, uint64_t v) { if (v > *p) *p = v; }
  uint32_t atomic_cas(uint32_t* p, uint32_t cmp, uint32_t val) { uint32_t o = *p; if (o == cmp) *p = val; return o; }
  unsigned long long clock() { return 0; }
  void use(double) {}
  uint64_t wave_max_u64(uint64_t v) { return v; }
  bool is_wave_leader() { return true; }
  int wave_width() { return 1; }
  uint64_t ballot(bool p) { return p ? 1ull : 0ull; }
  int clz64(uint64_t x) { return __builtin_clzll(x); }
  int popc64(uint64_t x) { return __builtin_popcountll(x); }
  unsigned long long global_add(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
};

const char* name() { return "sim"; }
int init(int, std::string*) { return 0; }
int bind_thread(std::string*) { return 0; }
int current_device() { return 0; }
void* alloc(size_t bytes, std::string* err) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p && err) *err = "out of host memory";
  return p;
}
void release(void* p) { free(p); }
void* alloc_host(size_t bytes, std::string* err) { return alloc(bytes, err); }
void release_host(void* p) { free(p); }
int h2d(void* d, const void* s, size_t n, std::string*) { memcpy(d, s, n); return 0; }
int d2h(void* d, const void* s, size_t n, std::string*) { memcpy(d, s, n); return 0; }
int d2d(void* d, const void* s, size_t n, std::string*) { memcpy(d, s, n); return 0; }
int zero(void* d, size_t n, std::string*) { memset(d, 0, n); return 0; }
int h2d_2d_overlapped(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, std::string*) {
  for (size_t r = 0; r < height; ++r) memcpy((char*)d + r * dpitch, (const char*)s + r * spitch, width);
  return 0;
}
int sync(std::string*) { return 0; }
void use_stream(int) {}
int n_events() { return 80; }
int ev_record(int, std::string*) { return 0; }
int ev_wait(int, std::string*) { return 0; }
int ev_sync(int, std::string*) { return 0; }
double ev_elapsed_ms(int, int) { return 0.0; }
int d2h_async(void* d, const void* s, size_t n, std::string*) { memcpy(d, s, n); return 0; }
int h2d_async(void* d, const void* s, size_t n, std::string*) { memcpy(d, s, n); return 0; }
int sync_all(std::string*) { return 0; }
int cus() { return 4; }  // small on purpose: the chunked pipeline is exercised by a few dozen utterances
void set_last_timing(double, double) {}
void last_timing(double* a, double* b) { *a = 0; *b = 0; }

static double load(const void* base, int dtype, size_t idx) {
  if (dtype == 0) return (double)((const float*)base)[idx];
  if (dtype == 1) return ((const double*)base)[idx];
  uint16_t h = ((const uint16_t*)base)[idx];
  if (dtype == 3) {  // bf16
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return (double)f;
  }
  // fp16 -> double
  int sign = h >> 15, ex = (h >> 10) & 31, man = h & 1023;
  double v = ex == 0 ? ldexp((double)man, -24) : ex == 31 ? (man ? NAN : INFINITY) : ldexp((double)(man | 1024), ex - 25);
  return sign ? -v : v;
}

int launch_prune(const PruneArgs& a, std::string*) {
  if (a.pass != 0) return 0;  // the sequential reference resolves the input kind exactly in pass 0
  const int V = a.n_labels;
  const double clip_lo = log(1e-15);
  std::vector<double> lp((size_t)V);
  std::vector<uint16_t> asc((size_t)V + 1), order((size_t)V + 2);
  uint32_t cap = set_table_cap((uint32_t)V + 1);
  std::vector<uint16_t> ta(cap), tr(cap), sc(cap);
  for (int u = 0; u < a.n_utts; ++u) {
    const void* x = a.utt_logits[u];
    int64_t r0 = a.utt_row0[u], T = a.utt_row0[u + 1] - r0;
    // decoder.py:760 on the input dtype, in numpy's summation order (np_sum.h)
    for (int64_t t = 0; t < T; ++t) a.row_sum[r0 + t] = np_row_sum(x, a.dtype, t, V);
    const double mean = T > 0 ? np_mean_of_sums(a.row_sum + r0, a.dtype, T) : NAN;
    bool is_prob = np_mean_is_one(mean);
    a.utt_is_prob[u] = is_prob ? 1u : 0u;
    if (a.utt_side && is_prob) a.utt_side[u] = 3u;
    if (a.utt_sum)
      for (int64_t t = 0; t < T; ++t) a.utt_sum[u] += a.row_sum[r0 + t];
    // float32 rows: the reference's arithmetic stays float32 (decoder.py:180-197, 762 on a float32 array: numpy's float32
    // exp / log -- its SIMD kernels, restated in np_f32.h --, a float32 pairwise sum, float32 subtractions) and only the clip
    // against ln(1e-15) widens the result: restated step by step, so that a frame's log-probabilities are the reference's bits.
    // CTCDEC_PRUNE_EXP=f64 keeps the round-2 behaviour (everything in fp64 from the exact upcast).
    const char* pe = getenv("CTCDEC_PRUNE_EXP");
    const bool f32_exact = a.dtype == 0 && !(pe && pe[0] == 'f');
    for (int64_t t = 0; t < T; ++t) {
      if (f32_exact) {
        const float* xr = (const float*)x + (size_t)t * V;
        if (is_prob) {
          const float lo = (float)1e-15;
          for (int v = 0; v < V; ++v) {
            float p = xr[v];
            p = p < lo ? lo : (p > 1.0f ? 1.0f : p);  // (NaN stays NaN, like np.clip)
            lp[(size_t)v] = (double)np_log_f32(p);
          }
        } else {
          float mx = -INFINITY;
          for (int v = 0; v < V; ++v) mx = fmaxf(mx, xr[v]);
          if (!std::isfinite(mx)) mx = 0.0f;
          const float se = np_pairwise<float>([&](int64_t v) { return np_exp_f32(xr[v] - mx); }, (int64_t)V);
          const float lse = np_log_f32(se);
          for (int v = 0; v < V; ++v) {
            const double y = (double)((xr[v] - mx) - lse);
            lp[(size_t)v] = y < clip_lo ? clip_lo : (y > 0.0 ? 0.0 : y);
          }
        }
      } else if (is_prob) {
        for (int v = 0; v < V; ++v) {
          double p = load(x, a.dtype, (size_t)t * V + v);
          p = p < 1e-15 ? 1e-15 : (p > 1.0 ? 1.0 : p);
          lp[(size_t)v] = log(p);
        }
      } else {
        double mx = -INFINITY;
        for (int v = 0; v < V; ++v) mx = fmax(mx, load(x, a.dtype, (size_t)t * V + v));
        if (!isfinite(mx)) mx = 0.0;
        // the normaliser in numpy's own summation order (np.sum is pairwise: np_sum.h), so that for float64 input a frame's
        // log-probabilities are the reference's bit for bit wherever exp() rounds like numpy's
        const double se = np_pairwise<double>([&](int64_t v) { return exp(load(x, a.dtype, (size_t)t * V + (size_t)v) - mx); }, (int64_t)V);
        double lse = log(se);
        for (int v = 0; v < V; ++v) {
          double y = (load(x, a.dtype, (size_t)t * V + v) - mx) - lse;
          lp[(size_t)v] = y < clip_lo ? clip_lo : (y > 0.0 ? 0.0 : y);
        }
      }
      uint32_t n = 0, amax = 0;
      for (int v = 0; v < V; ++v) {
        // numpy.argmax: the first NaN wins, else the first maximum
        if (!std::isnan(lp[amax]) && (std::isnan(lp[(size_t)v]) || lp[(size_t)v] > lp[amax])) amax = (uint32_t)v;
        if (lp[(size_t)v] >= a.token_min_logp) asc[n++] = (uint16_t)v;
      }
      uint32_t m = cpython_set_order(asc.data(), n, amax, ta.data(), tr.data(), sc.data(), order.data());
      size_t row = (size_t)(r0 + t);
      if (m > (uint32_t)a.max_surv) {
        *a.overflow = 1;
        m = (uint32_t)a.max_surv;
      }
      a.surv_cnt[row] = m;
      for (uint32_t k = 0; k < m; ++k) {
        a.surv_id[row * a.max_surv + k] = order[k];
        a.surv_lp[row * a.max_surv + k] = lp[order[k]];
      }
    }
  }
  return 0;
}

int launch_sniff_exact(const PruneArgs&, std::string*) { return 0; }  // (pass 0 of the sequential reference is exact)

static void fill_io(const BeamArgs& a, int u, UttIO& io) {
  int64_t r0 = a.utt_row0[u];
  io.surv_cnt = a.surv_cnt + r0;
  io.surv_id = a.surv_id + (size_t)r0 * a.params.max_surv;
  io.surv_lp = a.surv_lp + (size_t)r0 * a.params.max_surv;
  io.T = (int32_t)(a.utt_row0[u + 1] - r0);
  io.text_nodes = a.text_nodes + a.text_off[u];
  io.text_cap = (uint32_t)(a.text_off[u + 1] - a.text_off[u]);
  io.emit_nodes = a.emit_nodes + a.emit_off[u];
  io.emit_cap = (uint32_t)(a.emit_off[u + 1] - a.emit_off[u]);
  const uint32_t n_lms = a.tables.n_lms > 1 ? a.tables.n_lms : 1u;
  io.start_state = a.start_states ? a.start_states + (size_t)u * n_lms : nullptr;
  io.out_xstates = a.out_xstates ? a.out_xstates + (size_t)u * a.out_stride * (n_lms - 1) : nullptr;
  io.out = a.out + (size_t)u * a.out_stride;
  io.n_out = a.n_out + u;
  io.status = a.status + u;
  io.tok_pool = a.tok_pool;
  io.tok_pool_head = a.tok_pool_head;
  io.tok_pool_cap = a.tok_pool_cap;
  io.prof = nullptr;
  io.imports = (a.imports && !a.resident_in) ? a.imports + a.import_off[u] : nullptr;
  io.n_import = (a.imports && !a.resident_in) ? (int32_t)(a.import_off[u + 1] - a.import_off[u]) : 0;
  io.import_xstates = (a.imports && a.import_xstates && !a.resident_in) ? a.import_xstates + (size_t)a.import_off[u] * (n_lms - 1) : nullptr;
  io.first_frame = a.first_frames ? a.first_frames[u] : a.params.first_frame;
  io.cold = a.cold ? a.cold + (size_t)u * 2 * COLD_STRIDE : nullptr;
  io.pay = a.pay ? a.pay + (size_t)u * a.pay_stride : nullptr;
  io.carry_out = a.carry_out ? a.carry_out + (size_t)u * a.carry_stride : nullptr;
  io.carry_xstates = (a.carry_out && a.carry_xstates) ? a.carry_xstates + (size_t)u * a.carry_stride * (n_lms - 1) : nullptr;
  io.sstate = a.sstate ? a.sstate + u : nullptr;
  io.emit_start = a.sstate ? a.sstate[u].emit_next : 0u;
  io.want_out = a.want_out;
  if (a.resident_in) {
    io.imports = a.imports + (size_t)u * a.carry_stride;
    io.n_import = (int32_t)a.sstate[u].n_carry;
    io.import_xstates = a.import_xstates ? a.import_xstates + (size_t)u * a.carry_stride * (n_lms - 1) : nullptr;
  }
}

// one wavefront per utterance (beam_wave.h) on 64 cooperative fibers
template <int BW, int ORD>
static void run_wave_ord(const BeamArgs& a) {
  const size_t bytes = wave_lds_bytes<BW>();
  std::vector<char> lds(bytes + 64);
  char* base = (char*)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15);
  for (int b = 0; b < a.n_utts; ++b) {
    const int u = a.order ? a.order[b] : b;  // (the dispatch order of the HIP launch: longest utterance first)
    memset(base, 0xCD, bytes);  // poison: catch reads of never-written LDS
    WaveLds view;
    wave_lds_carve<BW>(view, base);
    UttIO io;
    fill_io(a, u, io);
    wavesim::Wave wave;
    wave.run([&](int lane) {
      wavesim::SimWaveCtx ctx{lane, &wave, &a.tables, &a.params};
      WaveDecoder<wavesim::SimWaveCtx, BW, ORD> dec(ctx, view, io);
      dec.run();
    });
  }
}
template <int BW>
static void run_wave(const BeamArgs& a) {  // (the instantiations of the HIP launcher: n-gram orders up to 4, or all)
  if (!a.tables.has_lm || a.tables.lm_order <= 4) run_wave_ord<BW, 4>(a);
  else run_wave_ord<BW, MAX_CTX + 1>(a);
}

static int g_last_kernel = 0;
int last_beam_kernel() { return g_last_kernel; }

// the launch behind the beam kernel for params.texts_only (backend_hip.hip: assemble_texts)
static void assemble_texts(const BeamArgs& a) {
  for (int u = 0; u < a.n_utts; ++u) {
    if (a.n_out[u] == 0) continue;
    OutBeam& ob = a.out[(size_t)u * a.out_stride];
    uint8_t* scratch = a.text_scratch + a.text_soff[u];
    const uint32_t cap = (uint32_t)(a.text_soff[u + 1] - a.text_soff[u]);
    const uint32_t n_nodes = (uint32_t)(a.emit_off[u + 1] - a.emit_off[u]);
    // the wave's walk (text_wave.h, what the HIP build launches) on 64 fibers ...
    std::vector<char> lds(TEXT_LDS_BYTES + 16);
    memset(lds.data(), 0xCD, lds.size());
    TextLds tl;
    text_lds_carve(tl, (char*)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15));
    uint32_t wave_pos[wavesim::LANES];
    memset(scratch, 0xEE, cap);
    wavesim::Wave wave;
    wave.run([&](int lane) {
      wavesim::SimWaveCtx ctx{lane, &wave, &a.tables, &a.params};
      wave_pos[lane] = wave_text_backwards(ctx, tl, a.emit_nodes + a.emit_off[u], a.tables, ob.pad[1], scratch, cap, n_nodes);
    });
    // ... checked against the one-thread walk (beam_core.h) on every call
    std::vector<uint8_t> seq(cap ? cap : 1, 0xEE);
    const uint32_t pos = text_backwards(a.emit_nodes + a.emit_off[u], a.tables, ob.pad[1], seq.data(), cap, n_nodes);
    for (int l = 0; l < wavesim::LANES; ++l)
      if (wave_pos[l] != pos) {
        fprintf(stderr, "sim: wave_text_backwards returns %u in lane %d, text_backwards %u (utterance %d)\n", wave_pos[l], l, pos, u);
        abort();
      }
    if (memcmp(scratch + pos, seq.data() + pos, cap - pos) != 0) {
      fprintf(stderr, "sim: wave_text_backwards and text_backwards write different bytes (utterance %d)\n", u);
      abort();
    }
    uint32_t len = cap - pos;
    unsigned long long base = a.tok_pool_head[1];
    if (base + len > a.text_pool_cap) {
      a.status[u] |= ST_TOK_OVERFLOW;
      base = 0;
      len = 0;
    }
    a.tok_pool_head[1] = base + len;
    ob.tok_off = (uint32_t)base;
    ob.tok_cnt = len;
    memcpy(a.text_pool + base, scratch + pos, len);
  }
}

static int launch_beam_kernels(const BeamArgs& a, std::string*);
int launch_beam(const BeamArgs& a, std::string* err) {
  const int rc = launch_beam_kernels(a, err);
  if (rc == 0 && a.n_utts > 0 && a.params.texts_only && a.text_scratch) assemble_texts(a);
  return rc;
}
bool beam_kernel_depends_on_input(const BeamArgs&) { return false; }  // (the simulator runs the wave kernel unless told otherwise)
bool wave_kernel_chosen(const BeamArgs& a) {
  const char* force = getenv("CTCDEC_BEAM_KERNEL");  // "wave" / "group": same switch as the HIP backend (default here: wave)
  const bool want_group = force && force[0] == 'g';
  return a.n_utts > 0 && !want_group && wave_eligible(a.tables, a.params) && a.max_import <= wave_bucket(a.params.beam_width);
}
static int launch_beam_kernels(const BeamArgs& a, std::string*) {
  if (a.pay && wave_kernel_chosen(a)) {
    switch (wave_bucket(a.params.beam_width)) {
      case 64: run_wave<64>(a); break;
      case 100: run_wave<100>(a); break;
      default: run_wave<128>(a); break;
    }
    g_last_kernel = 1;
    return 0;
  }
  if (a.n_utts > 0) g_last_kernel = 2;
  // CTCDEC_SIM_CAND=512|1024|2048: the candidate chunk of the workgroup kernel (the device build takes 1024 for its
  // 512-thread variant, 512 otherwise -- beam_core.h: group_cand)
  int cand = CAND_CHUNK;
  if (const char* e = getenv("CTCDEC_SIM_CAND")) {
    const int v = atoi(e);
    if (v == 512 || v == 1024 || v == 2048) cand = v;
  }
  // CTCDEC_SIM_GROUP_THREADS=64|128|256|512: run the workgroup kernel on that many cooperative fibers instead of one
  // sequential thread (barrier placement and every thread-count-dependent path as on the device; 512 also picks the
  // device's wide candidate chunk unless CTCDEC_SIM_CAND says otherwise)
  int group_threads = 0;
  if (const char* e = getenv("CTCDEC_SIM_GROUP_THREADS")) {
    const int v = atoi(e);
    if (v == 64 || v == 128 || v == 256 || v == 512) group_threads = v;
  }
  if (group_threads == 512 && !getenv("CTCDEC_SIM_CAND") && a.tables.n_lms <= 1)
    cand = group_cand(beam_bucket(a.params.beam_width), true);
  LdsShape shape = make_shape(a.params.beam_width, a.params.max_surv, cand);
  size_t bytes = lds_bytes(shape);
  std::vector<char> lds(bytes + 64);
  char* base = (char*)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15);
  for (int b = 0; b < a.n_utts; ++b) {
    const int u = a.order ? a.order[b] : b;
    memset(base, 0xCD, bytes);  // poison: catch reads of never-written LDS
    LdsView view;
    lds_carve(view, base, shape);
    UttIO io;
    fill_io(a, u, io);
    const uint32_t n_lms = a.tables.n_lms > 1 ? a.tables.n_lms : 1u;
    if (group_threads > 0) {  // as many fibers as the device launch has threads (group_fibers.h)
      static thread_local groupsim::Block block;
      block.run(group_threads, [&](int tid) {
        groupsim::GroupFiberCtx ctx{tid, group_threads, &block};
        if (n_lms > 1) {
          BeamDecoder<groupsim::GroupFiberCtx, true> dec(ctx, view, shape, a.tables, a.params, io);
          dec.run();
        } else {
          BeamDecoder<groupsim::GroupFiberCtx, false> dec(ctx, view, shape, a.tables, a.params, io);
          dec.run();
        }
      });
      continue;
    }
    SeqCtx ctx;
    if (n_lms > 1) {
      BeamDecoder<SeqCtx, true> dec(ctx, view, shape, a.tables, a.params, io);
      dec.run();
    } else {
      BeamDecoder<SeqCtx, false> dec(ctx, view, shape, a.tables, a.params, io);
      dec.run();
    }
  }
  return 0;
}

}  // namespace be
}  // namespace ctc
