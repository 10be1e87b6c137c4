// backend_hip.hip -- the product backend: HIP kernels for gfx950 (MI355X, CDNA4, wave64).
//
//   frame_prune[_f32x4]       log-softmax + clip (decoder.py:180-197,762-765), token prune and
//                             argmax (decoder.py:444-445), CPython-set ordering (set_order.h), row
//                             sums for the input sniff; one wave per frame row, fully parallel over
//                             all frames of the batch -- the stage that streams the [T x V] logits
//                             from HBM exactly once (fp32 rows live in registers)
//   utt_sniff                 decoder.py:760 (mean row sum ~ 1 => probabilities; those utterances
//                             get a second frame_prune pass with log(clip(p)))
//   beam_decode               the sequential prefix-beam recursion (beam_core.h); one workgroup
//                             per utterance, beam table / candidates / merge table in LDS
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <algorithm>
#include <type_traits>
#include <vector>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "backend.h"
#include "beam_core.h"
#include "beam_wave.h"
#include "set_order.h"
#include "np_f32.h"
#include "set_order_small.h"
#include "np_sum.h"
#include "text_wave.h"
#include "wave_ops_hip.h"

namespace ctc {
namespace be {

// Stream 0 is the decode stream of small batches; large batches are cut into chunks whose frame-prune kernels
// (stream 0), beam kernels (stream 1) and result copies (stream 2) overlap (api.cpp: decode_pipelined). Every
// backend call acts on the CURRENT stream (use_stream); calls are serialised by the caller's device mutex.
constexpr int N_STREAMS = 3;
constexpr int N_EVENTS = 80;
static hipStream_t g_streams[N_STREAMS] = {nullptr, nullptr, nullptr};
static hipStream_t g_stream = nullptr;  // the current one
static hipEvent_t g_ev[3] = {nullptr, nullptr, nullptr};
static hipEvent_t g_pool_ev[N_EVENTS] = {};
static double g_timing_override[2] = {-1.0, -1.0};
static int g_device = -1;
static int g_cus = 256;  // compute units of the device (MI355X: 256)
static bool g_timing_valid = false;

#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      if (err) *err = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
      return -1;                                                                         \
    }                                                                                    \
  } while (0)

const char* name() { return "hip-gfx950"; }

// One process drives ONE GPU (the multi-GPU layout is one process per GPU): the stream, the
// timing events and every workspace belong to the device of the first decoder. A later decoder asking for a
// different device is refused instead of silently mixing that device's kernels with this device's stream.
int init(int device, std::string* err) {
  if (g_stream) {
    int n = 0;
    (void)hipGetDeviceCount(&n);
    if (device < 0 || device >= n) device = 0;
    if (device != g_device) {
      if (err) *err = "this process already decodes on device " + std::to_string(g_device) + "; device " + std::to_string(device) +
                      " needs its own process (one process per GPU)";
      return -1;
    }
    return bind_thread(err);
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    if (err) *err = "no HIP device available (libctcdec has no CPU fallback)";
    return -1;
  }
  if (device < 0 || device >= n) device = 0;
  HIP_TRY(hipSetDevice(device));
  if (!g_stream) {
    for (int k = 0; k < N_STREAMS; ++k) HIP_TRY(hipStreamCreateWithFlags(&g_streams[k], hipStreamNonBlocking));
    g_stream = g_streams[0];
    for (int k = 0; k < 3; ++k) HIP_TRY(hipEventCreate(&g_ev[k]));
    for (int k = 0; k < N_EVENTS; ++k) HIP_TRY(hipEventCreate(&g_pool_ev[k]));
  }
  g_device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) g_cus = prop.multiProcessorCount;
  return 0;
}

// Host threads other than the one that created the decoder start with device 0 current: every entry point that
// allocates, copies or launches binds the calling thread to the decoder's device first.
int bind_thread(std::string* err) {
  if (g_device >= 0) HIP_TRY(hipSetDevice(g_device));
  return 0;
}
int current_device() { return g_device; }

void* alloc(size_t bytes, std::string* err) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
  if (e != hipSuccess) {
    if (err) *err = std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e);
    return nullptr;
  }
  return p;
}
void release(void* p) { (void)hipFree(p); }
void* alloc_host(size_t bytes, std::string* err) {
  void* p = nullptr;
  hipError_t e = hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault);
  if (e != hipSuccess) {
    if (err) *err = std::string("hipHostMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e);
    return nullptr;
  }
  return p;
}
void release_host(void* p) { (void)hipHostFree(p); }
int h2d(void* d, const void* s, size_t n, std::string* err) {
  HIP_TRY(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, g_stream));
  HIP_TRY(hipStreamSynchronize(g_stream));  // the source is caller memory that may be reused
  return 0;
}
int h2d_2d_overlapped(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, std::string* err) {
  hipStream_t copy = g_streams[N_STREAMS - 1];
  if (height == 0 || width == 0) return 0;
  if (height == 1 || (dpitch == width && spitch == width)) HIP_TRY(hipMemcpyAsync(d, s, width * height, hipMemcpyHostToDevice, copy));
  else HIP_TRY(hipMemcpy2DAsync(d, dpitch, s, spitch, width, height, hipMemcpyHostToDevice, copy));
  HIP_TRY(hipStreamSynchronize(copy));
  return 0;
}
int d2h(void* d, const void* s, size_t n, std::string* err) {
  HIP_TRY(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, g_stream));
  HIP_TRY(hipStreamSynchronize(g_stream));
  return 0;
}
int d2d(void* d, const void* s, size_t n, std::string* err) {
  HIP_TRY(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, g_stream));
  return 0;
}
int zero(void* d, size_t n, std::string* err) {
  HIP_TRY(hipMemsetAsync(d, 0, n, g_stream));
  return 0;
}
int sync(std::string* err) {
  HIP_TRY(hipStreamSynchronize(g_stream));
  return 0;
}

void use_stream(int idx) { g_stream = g_streams[idx >= 0 && idx < N_STREAMS ? idx : 0]; }
int n_events() { return N_EVENTS; }
int ev_record(int id, std::string* err) {
  HIP_TRY(hipEventRecord(g_pool_ev[id], g_stream));
  return 0;
}
int ev_wait(int id, std::string* err) {  // the current stream waits for the event
  HIP_TRY(hipStreamWaitEvent(g_stream, g_pool_ev[id], 0));
  return 0;
}
int ev_sync(int id, std::string* err) {  // the host waits for the event
  HIP_TRY(hipEventSynchronize(g_pool_ev[id]));
  return 0;
}
double ev_elapsed_ms(int a, int b) {
  float ms = 0.f;
  return hipEventElapsedTime(&ms, g_pool_ev[a], g_pool_ev[b]) == hipSuccess ? (double)ms : 0.0;
}
int d2h_async(void* d, const void* s, size_t n, std::string* err) {
  HIP_TRY(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, g_stream));
  return 0;
}
int h2d_async(void* d, const void* s, size_t n, std::string* err) {
  HIP_TRY(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, g_stream));
  return 0;
}
int sync_all(std::string* err) {
  for (int k = 0; k < N_STREAMS; ++k) HIP_TRY(hipStreamSynchronize(g_streams[k]));
  return 0;
}
int cus() { return g_cus; }
void set_last_timing(double prune_ms, double beam_ms) {
  g_timing_override[0] = prune_ms;
  g_timing_override[1] = beam_ms;
}

void last_timing(double* prune_ms, double* beam_ms) {
  *prune_ms = 0;
  *beam_ms = 0;
  if (g_timing_override[0] >= 0.0) {
    *prune_ms = g_timing_override[0];
    *beam_ms = g_timing_override[1];
    return;
  }
  if (!g_timing_valid) return;
  float a = 0, b = 0;
  if (hipEventSynchronize(g_ev[2]) != hipSuccess) return;
  if (hipEventElapsedTime(&a, g_ev[0], g_ev[1]) == hipSuccess) *prune_ms = a;
  if (hipEventElapsedTime(&b, g_ev[1], g_ev[2]) == hipSuccess) *beam_ms = b;
}

template <typename T>
__device__ __forceinline__ double ld(const T* p, size_t i) {
  return (double)p[i];
}
// 16-bit logits straight from the acoustic model: exact widening to fp64 (no intermediate fp32 copy)
struct half_bits { uint16_t u; };
struct bf16_bits { uint16_t u; };
template <>
__device__ __forceinline__ double ld<half_bits>(const half_bits* p, size_t i) {
  return (double)__half2float(__ushort_as_half(p[i].u));
}
template <>
__device__ __forceinline__ double ld<bf16_bits>(const bf16_bits* p, size_t i) {
  return (double)__uint_as_float(((uint32_t)p[i].u) << 16);
}

// row -> utterance (binary search over the prefix sums)
__device__ __forceinline__ int find_utt(const int64_t* row0, int n_utts, int64_t row) {
  int lo = 0, hi = n_utts - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (row0[mid] <= row) lo = mid; else hi = mid - 1;
  }
  return lo;
}

constexpr int PRUNE_WAVES = 4;  // rows per 256-thread block

// decoder.py:760 in two steps. The reference tests  math.isclose(logits.sum(axis=1).mean(), 1)  in the INPUT dtype, and for
// float32 / float16 that is in effect "does numpy's mean round to exactly 1". The frame-prune pass leaves exact (fp64) row
// sums; an utterance whose fp64 mean is nowhere near 1 is not a probability matrix in any summation order, everything
// else is marked ambiguous (2) and settled by utt_sniff_exact in numpy's own order and precision (np_sum.h).
__global__ __launch_bounds__(64) void utt_sniff(PruneArgs a) {
  const int u = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t r0 = a.utt_row0[u], r1 = a.utt_row0[u + 1];
  double s = 0.0, c = 0.0;
  {  // four rows per lane and trip: the loads of a trip are in flight together (one row per trip: 58 us of load latency for 4096 x 1000 rows)
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int64_t r = r0 + lane; r < r1; r += 256) {
      const bool h1 = r + 64 < r1, h2 = r + 128 < r1, h3 = r + 192 < r1;
      const double x0 = a.row_sum[r], x1 = h1 ? a.row_sum[r + 64] : 0.0, x2 = h2 ? a.row_sum[r + 128] : 0.0, x3 = h3 ? a.row_sum[r + 192] : 0.0;
      const uint32_t n0 = a.surv_cnt[r], n1 = h1 ? a.surv_cnt[r + 64] : 0u, n2 = h2 ? a.surv_cnt[r + 128] : 0u, n3 = h3 ? a.surv_cnt[r + 192] : 0u;
      s += x0; s1 += x1; s2 += x2; s3 += x3;
      c0 += n0; c1 += n1; c2 += n2; c3 += n3;
    }
    s = (s + s1) + (s2 + s3);
    c = (double)c0 + (double)c1 + (double)c2 + (double)c3;
  }
  s = wave_sum(s);
  c = wave_sum(c);
  if (lane == 0) {
    // (the survivors of all rows, summed: what a small batch's beam kernel is chosen by; saturating, only read for small batches)
    // (only small batches read it, and 4096 atomics on one address are 40 us of a 55-us kernel: launches of more than 1024
    //  utterances leave it alone)
    if (a.pass == 0 && a.n_utts <= 1024) atomicAdd(&a.overflow[4], (uint32_t)fmin(c, 1.0e9));
    double mean = r1 > r0 ? s / (double)(r1 - r0) : NAN;
    // (an infinite mean -- rows masked with -inf -- is never close: math.isclose(+-inf, 1) is False.) The window:
    // float32 pairwise sums of rows of ordinary logits are off by < 1e-2; float64 ones by < 1e-12.
    const bool amb = isfinite(mean) && fabs(mean - 1.0) <= (a.dtype == 1 ? 1e-6 : 0.5);
    // (time-sliced ingest: the reference tests the WHOLE utterance's mean row sum -- the slices' sums are added up for the caller)
    if (a.utt_sum && r1 > r0) atomicAdd(&a.utt_sum[u], s);
    a.utt_is_prob[u] = amb ? 2u : 0u;
    if (amb) a.overflow[2] = 1u;  // flags[2]: some utterance needs the exact test
  }
}
// pass 1 redid the rows of the probability utterances as log(clip(p)): the survivors of all rows counted again (pass 0 counted
// those rows as logits -- softmax outputs read as logits are nearly flat, almost every label survives -- and a small batch
// chooses its beam kernel by this number). The caller zeroes overflow[4] before the pass.
__global__ __launch_bounds__(64) void utt_recount(PruneArgs a) {
  const int u = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t r0 = a.utt_row0[u], r1 = a.utt_row0[u + 1];
  double c = 0.0;
  for (int64_t r = r0 + lane; r < r1; r += 64) c += (double)a.surv_cnt[r];
  c = wave_sum(c);
  if (lane == 0) atomicAdd(&a.overflow[4], (uint32_t)fmin(c, 1.0e9));
}
// one workgroup per ambiguous utterance, one thread per row (rare path: probability inputs, or logits whose rows sum to ~1)
__global__ __launch_bounds__(256) void utt_sniff_exact(PruneArgs a) {
  const int u = blockIdx.x;
  if (a.utt_is_prob[u] != 2u) return;
  const int64_t r0 = a.utt_row0[u], T = a.utt_row0[u + 1] - r0;
  const void* x = a.utt_logits[u];
  for (int64_t t = threadIdx.x; t < T; t += blockDim.x) a.row_sum[r0 + t] = np_row_sum(x, a.dtype, t, a.n_labels);
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool is_prob = T > 0 && np_mean_is_one(np_mean_of_sums(a.row_sum + r0, a.dtype, T));
    a.utt_is_prob[u] = is_prob ? 1u : 0u;
    if (is_prob && a.utt_side) a.utt_side[u] = 3u;  // (time-sliced ingest: a slice was read as probabilities)
    if (is_prob) a.overflow[1] = 1u;  // flags[1]: some utterance needs the probability pass
  }
}

// per-wave LDS work area of the prune kernels
constexpr int NP_LEAVES = 64;  // leaves (<= 128 elements each) of numpy's pairwise sum a wave handles: rows up to ~3600 labels
struct PruneLds {
  double* asc_lp;
  uint16_t *asc_id, *order, *tabA, *tabR, *scratch;
  double* np_sum;     // [NP_LEAVES + 16] leaf sums, then the combining stack (float64 rows: np_order_exp_sum)
  uint16_t* np_tab;   // [2 * NP_LEAVES + 48] leaf offsets, leaf lengths, two 16-entry stacks, the leaf count
};
constexpr size_t NP_LDS_BYTES = (NP_LEAVES + 16) * 8 + (((2 * NP_LEAVES + 48) * 2 + 15) & ~(size_t)15);
__host__ __device__ inline size_t prune_lds_bytes(size_t ms, size_t cap) {
  return (((ms + 1) * 8 + 15) & ~(size_t)15) + (((ms + 2) * 2 * 2 + 15) & ~(size_t)15) +
         ((cap * 2 * 3 + 15) & ~(size_t)15) + NP_LDS_BYTES;
}
__device__ __forceinline__ PruneLds prune_lds(char* smem, int wave, uint32_t ms, uint32_t cap) {
  char* base = smem + prune_lds_bytes(ms, cap) * wave;
  PruneLds w;
  w.asc_lp = (double*)base;
  w.asc_id = (uint16_t*)(base + (((size_t)(ms + 1) * 8 + 15) & ~(size_t)15));
  w.order = w.asc_id + (ms + 2);
  w.tabA = (uint16_t*)((char*)w.asc_id + (((size_t)(ms + 2) * 2 * 2 + 15) & ~(size_t)15));
  w.tabR = w.tabA + cap;
  w.scratch = w.tabR + cap;
  w.np_sum = (double*)(base + prune_lds_bytes(ms, cap) - NP_LDS_BYTES);
  w.np_tab = (uint16_t*)(w.np_sum + NP_LEAVES + 16);
  return w;
}

// numpy.argmax semantics on a row that may hold NaN (non-finite input rows): the first NaN wins, else the
// first maximum. (best, best_id) start as (-inf, INT_MAX); ids arrive in ascending order inside a lane.
__device__ __forceinline__ void argmax_take(double y, int id, double& best, int& best_id) {
  const bool bnan = best != best;
  if (!bnan && (y != y || y > best)) {
    best = y;
    best_id = id;
  }
}
__device__ __forceinline__ void argmax_merge(double ob, int oi, double& best, int& best_id) {
  const bool bnan = best != best, onan = ob != ob;
  const bool take = onan ? (!bnan || oi < best_id) : (!bnan && (ob > best || (ob == best && oi < best_id)));
  if (take) {
    best = ob;
    best_id = oi;
  }
}

// Common tail: argmax across the wave (first maximum, like numpy), CPython-set ordering on lane 0,
// coalesced write of the ordered (id, logp) list.
// ---- CPython set order with the table spread over the wave -------------------------------------------
// The same algorithm as set_order.h (Objects/setobject.c: set_add_entry / set_insert_clean /
// set_table_resize / set_merge) for tables of at most 64 slots: slot k lives in lane k, the free-slot map is
// a wave-uniform 64-bit mask, keys are wave-uniform -- so a probe sequence is a handful of SCALAR
// instructions on that mask (the linear 10-slot window is one shift + ctz) instead of a lane-0 walk over LDS.
// Up to 18 keys (+ argmax) never need more than 64 slots: 8 -> 32 at the 5th key, 128 only at the 19th.
constexpr uint32_t WAVE_SET_MAX_KEYS = 18;
struct WTab {
  uint32_t val;    // this lane's slot (SET_EMPTY when free)
  uint64_t empty;  // bit k: slot k is free            (uniform)
  uint32_t mask;   // table size - 1                   (uniform)
  uint32_t used;   //                                  (uniform)
};
__device__ __forceinline__ uint64_t wt_all(uint32_t size) { return size >= 64u ? ~0ull : ((1ull << size) - 1ull); }
__device__ __forceinline__ void wt_clear(WTab& t, uint32_t size) {
  t.val = SET_EMPTY;
  t.mask = size - 1u;
  t.used = 0;
  t.empty = wt_all(size);
}
__device__ __forceinline__ void wt_put(WTab& t, int lane, uint32_t pos, uint32_t key) {
  if ((uint32_t)lane == pos) t.val = key;
  t.empty &= ~(1ull << pos);
}
__device__ __forceinline__ void wt_insert_clean(WTab& t, int lane, uint32_t key) {  // set_insert_clean
  const uint32_t mask = t.mask;
  uint32_t perturb = key, i = key & mask;
  for (;;) {
    const uint64_t win = (t.empty >> i) & ((i + 9u <= mask) ? 0x3FFull : 1ull);  // slot i and the 9 after it
    if (win) {
      wt_put(t, lane, i + (uint32_t)__builtin_ctzll(win), key);
      return;
    }
    perturb >>= 5;
    i = (i * 5u + 1u + perturb) & mask;
  }
}
// every occupied slot of `src`, in slot order, re-inserted into `dst`
__device__ __forceinline__ void wt_reinsert(WTab& dst, int lane, uint32_t src_val, uint64_t src_occ) {
  while (src_occ) {
    const uint32_t k = (uint32_t)__builtin_ctzll(src_occ);
    src_occ &= src_occ - 1ull;
    const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)src_val, (int)k);
    wt_insert_clean(dst, lane, key);
  }
}
__device__ __forceinline__ void wt_resize(WTab& t, int lane, uint32_t minused) {  // set_table_resize
  uint32_t newsize = 8;
  while (newsize <= minused) newsize <<= 1;
  const uint32_t old_val = t.val, used = t.used;
  const uint64_t occ = ~t.empty & wt_all(t.mask + 1u);
  wt_clear(t, newsize);
  t.used = used;
  wt_reinsert(t, lane, old_val, occ);
}
__device__ __forceinline__ void wt_add(WTab& t, int lane, uint32_t key) {  // set_add_entry
  const uint32_t mask = t.mask;
  const uint64_t same = __ballot(t.val == key);
  uint32_t perturb = key, i = key & mask;
  for (;;) {
    const uint64_t winmask = (i + 9u <= mask) ? 0x3FFull : 1ull;
    const uint64_t we = (t.empty >> i) & winmask, ws = (same >> i) & winmask;
    if (we | ws) {
      const uint32_t first = (uint32_t)__builtin_ctzll(we | ws);
      if ((ws >> first) & 1ull) return;  // already a member
      wt_put(t, lane, i + first, key);
      t.used += 1u;
      if (t.used * 5u >= mask * 3u) wt_resize(t, lane, t.used > 50000u ? t.used * 2u : t.used * 4u);
      return;
    }
    perturb >>= 5;
    i = (i * 5u + 1u + perturb) & mask;
  }
}
// set(asc[0..n)) | {argmax} for n <= WAVE_SET_MAX_KEYS: returns the member count; this lane's slot value and
// the occupancy mask (slot order = iteration order) come back through the references.
__device__ __forceinline__ uint32_t wave_set_order(const uint16_t* asc, uint32_t n, uint32_t argmax, int lane,
                                                   uint32_t& my_val, uint64_t& occ) {
  WTab a, r;
  wt_clear(a, 8);
  for (uint32_t k = 0; k < n; ++k) wt_add(a, lane, (uint32_t)__builtin_amdgcn_readfirstlane((int)asc[k]));
  wt_clear(r, 8);
  if (a.used) {  // set_merge (the `|` copies the left operand first)
    if (a.used * 5u >= r.mask * 3u) {
      uint32_t newsize = 8;
      while (newsize <= a.used * 2u) newsize <<= 1;
      wt_clear(r, newsize);
    }
    if (r.mask == a.mask) {
      r.val = a.val;
      r.empty = a.empty;
    } else {
      wt_reinsert(r, lane, a.val, ~a.empty & wt_all(a.mask + 1u));
    }
    r.used = a.used;
  }
  if ((r.used + 1u) * 5u >= r.mask * 3u) wt_resize(r, lane, (r.used + 1u) * 2u);
  wt_add(r, lane, argmax);
  my_val = r.val;
  occ = ~r.empty & wt_all(r.mask + 1u);
  return (uint32_t)__popcll(occ);
}

__device__ __forceinline__ void prune_finish(const PruneArgs& a, int64_t row, int lane, const PruneLds& w, uint32_t n,
                                             double best, int best_id, bool reduced = false) {
  const uint32_t ms = (uint32_t)a.max_surv;
  bool overflow = false;
  if (n > ms) {
    overflow = true;
    n = ms;
  }
  if (!reduced) {  // (best, best_id) are per-lane candidates: reduce across the wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      double ob = __shfl_xor(best, off, 64);
      int oi = __shfl_xor(best_id, off, 64);
      argmax_merge(ob, oi, best, best_id);
    }
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  uint16_t* out_id = a.surv_id + (size_t)row * ms;
  double* out_lp = a.surv_lp + (size_t)row * ms;
  if (n <= WAVE_SET_MAX_KEYS) {  // the usual frame: a handful of survivors, table in registers
    uint32_t my_id = SET_EMPTY;
    uint64_t occ = 0;
    uint32_t m = wave_set_order(w.asc_id, n, (uint32_t)__builtin_amdgcn_readfirstlane(best_id), lane, my_id, occ);
    const uint32_t pos = (uint32_t)__popcll(occ & ((1ull << lane) - 1ull));
    if (m > ms) {
      overflow = true;
      m = ms;
    }
    if (lane == 0) {
      a.surv_cnt[row] = m;
      if (overflow) a.overflow[0] = 1u;
    }
    if (((occ >> lane) & 1ull) && pos < m) {
      double lp = best;  // the argmax may be absent from the list (below the threshold)
      for (uint32_t k = 0; k < n; ++k)
        if (w.asc_id[k] == my_id) lp = w.asc_lp[k];
      out_id[pos] = (uint16_t)my_id;
      out_lp[pos] = lp;
    }
    return;
  }
  uint32_t m = 0;
  if (lane == 0) m = cpython_set_order(w.asc_id, n, (uint32_t)best_id, w.tabA, w.tabR, w.scratch, w.order);
  m = __shfl(m, 0, 64);
  __threadfence_block();
  if (m > ms) {
    overflow = true;
    m = ms;
  }
  if (lane == 0) {
    a.surv_cnt[row] = m;
    if (overflow) a.overflow[0] = 1u;
  }
  for (uint32_t k = lane; k < m; k += 64) {
    uint32_t id = w.order[k];
    // binary search the ascending list for the log-prob (the argmax may be absent: below threshold)
    double lp = best;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
      uint32_t mid = (lo + hi) >> 1;
      if (w.asc_id[mid] < id) lo = mid + 1; else hi = mid;
    }
    if (lo < n && w.asc_id[lo] == id) lp = w.asc_lp[lo];
    out_id[k] = (uint16_t)id;
    out_lp[k] = lp;
  }
}

__device__ __forceinline__ double to_logp(double xv, bool is_prob, double mx, double lse) {
  const double clip_lo = -34.538776394910684;  // ln(1e-15)
  if (is_prob) {
    double p = xv < 1e-15 ? 1e-15 : (xv > 1.0 ? 1.0 : xv);
    return log(p);
  }
  double y = (xv - mx) - lse;
  return y < clip_lo ? clip_lo : (y > 0.0 ? 0.0 : y);
}

// float32 rows in the reference's own arithmetic (decoder.py:180-197 and :762 on a float32 array): (x - max) - log(sum) in
// float32 with numpy's float32 log (np_f32.h), widened by the clip; probability rows: numpy's float32 log of the float32 clip.
__device__ __forceinline__ double to_logp_np32(float xv, bool is_prob, float mf, float l32) {
  const double clip_lo = -34.538776394910684;  // ln(1e-15)
  if (is_prob) {
    const float lo = (float)1e-15;
    const float p = xv < lo ? lo : (xv > 1.0f ? 1.0f : xv);  // (NaN stays NaN, like np.clip)
    return (double)np_log_f32(p);
  }
  const float t = xv - mf;
  const double y = (double)(t - l32);
  return y < clip_lo ? clip_lo : (y > 0.0 ? 0.0 : y);
}

// float64 rows: sum_v exp(x[v] - m) in the order numpy's np.sum adds a contiguous float64 row (pairwise: np_sum.h), by one
// wave. The reference's log-softmax (decoder.py:180-197) is x - max - log(np.sum(np.exp(x - max))): with the normaliser
// summed in this order a frame's log-probabilities are the reference's own bit for bit wherever exp / log round like
// numpy's, and with them the order inside runs of equal scores (tests/test_order_stability.py). Leaves of the recursion
// (<= 128 elements: eight strided accumulators, then the leftovers) go to groups of eight lanes, eight leaves at a time;
// lane 0 lists them first and combines their sums last, both by walking the recursion with a small stack in LDS.
// Returns false for rows of more than NP_LEAVES leaves (the caller then sums in lane order: same value to ~1 ulp).
// T = double: the terms are exp(x - m) in fp64; T = float: numpy's float32 exp of the float32 difference, float32 additions
// (the leaf sums travel through the fp64 LDS words unchanged: every float32 is a double).
template <typename T>
__device__ __forceinline__ T np_term(const T* x, int i, T m);
template <>
__device__ __forceinline__ double np_term<double>(const double* x, int i, double m) { return exp(x[i] - m); }
template <>
__device__ __forceinline__ float np_term<float>(const float* x, int i, float m) { return np_exp_f32(x[i] - m); }
template <typename T>
__device__ __forceinline__ bool np_order_exp_sum(const T* x, int V, T m, int lane, const PruneLds& w, T* out) {
  uint16_t* leaf_off = w.np_tab;
  uint16_t* leaf_len = w.np_tab + NP_LEAVES;
  uint16_t* stk_n = w.np_tab + 2 * NP_LEAVES;
  uint16_t* stk_s = stk_n + 16;
  uint16_t* n_leaf = stk_s + 16;
  if (lane == 0) {  // the leaves, left to right
    int sp = 0, k = 0;
    stk_s[0] = 0;
    stk_n[0] = (uint16_t)V;  // (V <= 65535: label ids are 16 bits wide)
    sp = 1;
    while (sp > 0 && k <= NP_LEAVES) {
      --sp;
      const int off = stk_s[sp], n = stk_n[sp];
      if (n <= 128) {
        if (k < NP_LEAVES) {
          leaf_off[k] = (uint16_t)off;
          leaf_len[k] = (uint16_t)n;
        }
        ++k;
      } else if (sp + 2 <= 16) {
        int n2 = n / 2;
        n2 -= n2 % 8;
        stk_s[sp] = (uint16_t)(off + n2);  // right half below the left one: the left is listed first
        stk_n[sp] = (uint16_t)(n - n2);
        stk_s[sp + 1] = (uint16_t)off;
        stk_n[sp + 1] = (uint16_t)n2;
        sp += 2;
      } else {
        k = NP_LEAVES + 1;
      }
    }
    *n_leaf = (uint16_t)k;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int nl = *n_leaf;
  if (nl > NP_LEAVES) return false;
  const int g = lane >> 3, j = lane & 7;
  for (int base = 0; base < nl; base += 8) {
    const int k = base + g;
    const bool mine = k < nl;
    const int off = mine ? leaf_off[k] : 0, n = mine ? leaf_len[k] : 0;
    T res = (T)0;
    if (n < 8) {
      for (int i = 0; i < n; ++i) res = res + np_term<T>(x, off + i, m);
    } else {
      T r = np_term<T>(x, off + j, m);
      const int body = n - (n % 8);
      for (int i = 8; i < body; i += 8) r = r + np_term<T>(x, off + i + j, m);
      // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)): additions commute, so an xor butterfly inside the group is that very tree
      r = r + __shfl_xor(r, 1, 64);
      r = r + __shfl_xor(r, 2, 64);
      r = r + __shfl_xor(r, 4, 64);
      res = r;
      for (int i = body; i < n; ++i) res = res + np_term<T>(x, off + i, m);
    }
    if (mine && j == 0) w.np_sum[k] = (double)res;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {  // pairwise(a, n2) + pairwise(a + n2, n - n2), all the way up
    double* acc = w.np_sum + NP_LEAVES;
    uint16_t* stk_st = stk_s;
    int sp = 1, k = 0;
    stk_n[0] = (uint16_t)V;
    stk_st[0] = 0;
    T ret = (T)0;
    while (sp > 0) {
      const int n = stk_n[sp - 1], st = stk_st[sp - 1];
      if (n <= 128) {
        ret = (T)w.np_sum[k++];
        --sp;
        continue;
      }
      int n2 = n / 2;
      n2 -= n2 % 8;
      if (st == 0) {
        stk_st[sp - 1] = 1;
        stk_n[sp] = (uint16_t)n2;
        stk_st[sp] = 0;
        ++sp;
      } else if (st == 1) {
        acc[sp - 1] = (double)ret;
        stk_st[sp - 1] = 2;
        stk_n[sp] = (uint16_t)(n - n2);
        stk_st[sp] = 0;
        ++sp;
      } else {
        ret = (T)acc[sp - 1] + ret;
        --sp;
      }
    }
    w.np_sum[0] = (double)ret;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  *out = (T)w.np_sum[0];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return true;
}
template <typename T>
__device__ __forceinline__ bool np_order_exp_sum_t(const T*, int, double, int, const PruneLds&, double*) { return false; }
template <>
__device__ __forceinline__ bool np_order_exp_sum_t<double>(const double* x, int V, double m, int lane, const PruneLds& w, double* out) {
  return np_order_exp_sum<double>(x, V, m, lane, w, out);
}

// Generic frame-prune: one wave per frame row, any V / dtype; the row is swept three times (max and
// row sum, sum of exponentials, selection) and stays in L1/L2 between sweeps.
// pass 0: every utterance is treated as logits (the overwhelmingly common case) and the row sums for the
//         probability sniff are produced on the way; pass 1 (only launched when utt_sniff found
//         probability-like utterances) redoes just those utterances with log(clip(p)).
template <typename T>
__device__ __forceinline__ void prune_row_generic(const PruneArgs& a, int64_t row, const PruneLds& w, int lane) {
  const int V = a.n_labels;
  const int u = find_utt(a.utt_row0, a.n_utts, row);
  const bool is_prob = a.pass == 1;
  if (is_prob && a.utt_is_prob[u] != 1u) return;
  const T* x = (const T*)a.utt_logits[u] + (size_t)(row - a.utt_row0[u]) * V;
  double mx = 0.0, lse = 0.0;
  // float32 rows: the reference's own float32 arithmetic (to_logp_np32); rows of more leaves than a wave lists fall back to fp64
  constexpr bool F32 = std::is_same<T, float>::value;
  bool np32 = F32 && a.f32_np && is_prob;
  float mf = 0.f, l32 = 0.f;
  if (!is_prob) {
    double m = -INFINITY, rs = 0.0;
    for (int v = lane; v < V; v += 64) {
      double xv = ld(x, v);
      m = fmax(m, xv);
      rs += xv;
    }
    m = wave_max(m);
    rs = wave_sum(rs);
    if (lane == 0) a.row_sum[row] = rs;
    if (!isfinite(m)) m = 0.0;  // decoder.py:186-189
    if constexpr (F32) {
      if (a.f32_np) {
        float s32 = 0.f;
        mf = (float)m;  // (exact: the maximum of float32 values)
        if (np_order_exp_sum<float>((const float*)x, V, mf, lane, w, &s32)) {
          l32 = np_log_f32(s32);
          np32 = true;
        }
      }
    }
    if (!np32) {
      double s = 0.0;
      if (!np_order_exp_sum_t<T>(x, V, m, lane, w, &s)) {  // (float64 rows: numpy's own summation order)
        for (int v = lane; v < V; v += 64) s += exp(ld(x, v) - m);
        s = wave_sum(s);
      }
      lse = log(s);
    }
    mx = m;
  }
  uint32_t n = 0;
  double best = -INFINITY;
  int best_id = 0x7FFFFFFF;
  const uint32_t ms = (uint32_t)a.max_surv;
  for (int v0 = 0; v0 < V; v0 += 64) {
    int v = v0 + lane;
    double y = -INFINITY;
    bool in = v < V;
    if (in) {
      if constexpr (F32) y = np32 ? to_logp_np32(((const float*)x)[v], is_prob, mf, l32) : to_logp(ld(x, v), is_prob, mx, lse);
      else y = to_logp(ld(x, v), is_prob, mx, lse);
      argmax_take(y, v, best, best_id);
    }
    bool keep = in && y >= a.token_min_logp;
    unsigned long long mask = __ballot(keep);
    if (keep) {
      uint32_t pos = n + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      if (pos < ms) {
        w.asc_id[pos] = (uint16_t)v;
        w.asc_lp[pos] = y;
      }
    }
    n += (uint32_t)__popcll(mask);
  }
  prune_finish(a, row, lane, w, n, best, best_id);
}
template <typename T>
__global__ __launch_bounds__(PRUNE_WAVES * 64) void frame_prune(PruneArgs a, uint32_t cap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t row = a.row_base + (int64_t)blockIdx.x * PRUNE_WAVES + wave;
  if (row >= a.row_base + a.n_rows) return;
  prune_row_generic<T>(a, row, prune_lds(smem, wave, (uint32_t)a.max_surv, cap), lane);
}
// the rows frame_prune_fast left on its list for 16-bit inputs (a.slow_rows, a.overflow[3] of them), one wave per row
template <typename T>
__global__ __launch_bounds__(PRUNE_WAVES * 64) void frame_prune_listed(PruneArgs a, uint32_t cap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const PruneLds w = prune_lds(smem, wave, (uint32_t)a.max_surv, cap);
  const uint32_t count = a.overflow[3];
  for (uint32_t k = blockIdx.x * PRUNE_WAVES + wave; k < count; k += gridDim.x * PRUNE_WAVES) {
    prune_row_generic<T>(a, a.row_base + (int64_t)a.slow_rows[k], w, lane);
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
  }
}

// exp(d) for d <= 0 (d = logit - row max), fp64, ~1 ulp: 2^k * exp(r) with k = rint(d / ln 2), |r| <= ln2/2,
// exp(r) by its degree-13 Taylor polynomial (next term < 5e-18 relative). No overflow side to handle and the
// underflow side is a clamp, which is what makes it ~2/3 of the general routine's instructions.
__device__ __forceinline__ double exp_nonpos(double d) {
  d = fmax(d, -750.0);  // exp(-750) == 0 in fp64; also maps -inf (masked labels) to 0
  const double kf = rint(d * 1.44269504088896338700e+00);
  double r = fma(kf, -6.93147180369123816490e-01, d);
  r = fma(kf, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;            // 1/13!
  p = fma(p, r, 2.08767569878680990e-09);       // 1/12!
  p = fma(p, r, 2.50521083854417188e-08);       // 1/11!
  p = fma(p, r, 2.75573192239858907e-07);       // 1/10!
  p = fma(p, r, 2.75573192239858907e-06);       // 1/9!
  p = fma(p, r, 2.48015873015873016e-05);       // 1/8!
  p = fma(p, r, 1.98412698412698413e-04);       // 1/7!
  p = fma(p, r, 1.38888888888888889e-03);       // 1/6!
  p = fma(p, r, 8.33333333333333333e-03);       // 1/5!
  p = fma(p, r, 4.16666666666666667e-02);       // 1/4!
  p = fma(p, r, 1.66666666666666667e-01);       // 1/3!
  p = fma(p, r, 5.00000000000000000e-01);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)kf);
}

// The same with the degree-9 polynomial: relative error < 7e-12 (|r|^10 / 10!), for sums that feed a float32 input's
// log-softmax (frame_prune_f32x4); four FMAs less per call.
__device__ __forceinline__ double exp_nonpos9(double d) {
  d = fmax(d, -750.0);
  const double kf = rint(d * 1.44269504088896338700e+00);
  double r = fma(kf, -6.93147180369123816490e-01, d);
  r = fma(kf, -1.90821492927058770002e-10, r);
  double p = 2.75573192239858907e-06;           // 1/9!
  p = fma(p, r, 2.48015873015873016e-05);       // 1/8!
  p = fma(p, r, 1.98412698412698413e-04);       // 1/7!
  p = fma(p, r, 1.38888888888888889e-03);       // 1/6!
  p = fma(p, r, 8.33333333333333333e-03);       // 1/5!
  p = fma(p, r, 4.16666666666666667e-02);       // 1/4!
  p = fma(p, r, 1.66666666666666667e-01);       // 1/3!
  p = fma(p, r, 5.00000000000000000e-01);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)kf);
}

// float32 exp(d), d <= 0, two at a time (v_pk_* math): degree-8 Taylor polynomial after a two-constant argument reduction,
// every step one rounding (fma). Measured on the host with the same operations (tools/exp_f32_study.c, 20 M arguments in
// [-14, 0]): mean relative error 1.5e-11 (libm's expf: 3.5e-11), mean |error| 2.2e-8, maximum 7.4e-8 -- i.e. what numpy's
// float32 exp gives the reference (decoder.py:180-197), and without the one-sided error of v_exp_f32. Arguments below -80
// (incl. -inf masks) are clamped: exp(-80) = 1.8e-35 does not register in a sum >= 1.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 exp_nonpos_f32x2(f32x2 d) {
  d = __builtin_elementwise_max(d, (f32x2)(-80.0f));
  const f32x2 n = __builtin_elementwise_roundeven(d * (f32x2)(1.44269504088896340736f));
  f32x2 r = __builtin_elementwise_fma(n, (f32x2)(-0.693145751953125f), d);
  r = __builtin_elementwise_fma(n, (f32x2)(-1.42860682030941723212e-6f), r);
  f32x2 p = (f32x2)(2.48015873015873016e-05f);                              // 1/8!
  p = __builtin_elementwise_fma(p, r, (f32x2)(1.98412698412698413e-04f));   // 1/7!
  p = __builtin_elementwise_fma(p, r, (f32x2)(1.38888888888888889e-03f));   // 1/6!
  p = __builtin_elementwise_fma(p, r, (f32x2)(8.33333333333333333e-03f));   // 1/5!
  p = __builtin_elementwise_fma(p, r, (f32x2)(4.16666666666666667e-02f));   // 1/4!
  p = __builtin_elementwise_fma(p, r, (f32x2)(1.66666666666666667e-01f));   // 1/3!
  p = __builtin_elementwise_fma(p, r, (f32x2)(0.5f));
  p = __builtin_elementwise_fma(p, r, (f32x2)(1.0f));
  p = __builtin_elementwise_fma(p, r, (f32x2)(1.0f));
  f32x2 out;
  out.x = ldexpf(p.x, (int)n.x);
  out.y = ldexpf(p.y, (int)n.y);
  return out;
}

// log(s) for 1 <= s < 2^24, fp64: one Newton step on the fp32 logarithm -- y0 = logf(s), r = s * exp(-y0) - 1
// (|r| ~ 1e-6), log(s) = y0 + log1p(r) = y0 + r - r^2/2 (next term < 1e-18). A third of ocml's double-double log.
__device__ __forceinline__ double log_ge1(double s) {
  const double y0 = (double)__logf((float)s);
  const double r = fma(s, exp_nonpos(-y0), -1.0);
  return y0 + (r - 0.5 * r * r);
}

// Register-resident frame-prune for fp32 rows with V % 4 == 0 and V <= 1024*... (NC chunks of 256
// labels): each lane pulls its 4*NC logits with 16-byte loads ONCE (1 KiB per wave-instruction, fully
// coalesced) and all three sweeps run out of registers: the logits cross HBM exactly once.
// PK: the exponentials of the clean-row path as packed float32 polynomials (the default; CTCDEC_PRUNE_EXP=f64: fp64)
template <int NC, bool PK>
__device__ __forceinline__ void prune_row_f32x4(const PruneArgs& a, int64_t row, const PruneLds& w, int lane) {
  const int V = a.n_labels;
  const int u = find_utt(a.utt_row0, a.n_utts, row);
  const bool is_prob = a.pass == 1;
  if (is_prob && a.utt_is_prob[u] != 1u) return;
  const float4* x4 = (const float4*)((const float*)a.utt_logits[u] + (size_t)(row - a.utt_row0[u]) * V);
  const int n4 = V >> 2;
  float4 r[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int i4 = k * 64 + lane;
    r[k] = i4 < n4 ? x4[i4] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  }
  double mx = 0.0, lse = 0.0;
  const uint32_t ms = (uint32_t)a.max_surv;
  const unsigned long long lt = (1ull << lane) - 1ull;
  if (!is_prob) {
    float mf = -INFINITY;
    // The row sum only feeds utt_sniff's coarse "could these be probabilities?" test (|mean row sum - 1| <= 0.5; the exact
    // test in numpy's own order is utt_sniff_exact's): each lane adds its <= 16 values in float32 (error < 2e-5 for
    // logits up to +-20, nothing for probabilities), the wave sum is fp64.
    float rsf = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      if (k * 64 + lane < n4) {
        mf = fmaxf(fmaxf(mf, fmaxf(r[k].x, r[k].y)), fmaxf(r[k].z, r[k].w));
        rsf += (r[k].x + r[k].y) + (r[k].z + r[k].w);
      }
    }
    double m = wave_max((double)mf);
    double rs = wave_sum((double)rsf);
    if (lane == 0) a.row_sum[row] = rs;
    // ---- the clean row (finite maximum, no NaN: -inf masks are fine): everything below in its cheapest form
    if (isfinite(m) && rs == rs) {
      // The reference computes the log-softmax of float32 logits IN float32 (decoder.py:180-197: np.exp / np.sum on
      // the float32 array, correctly rounded per term, errors of either sign). Tried and dropped: v_exp_f32 -- a
      // third of the instructions, but the hardware exp2 errs to one side, which shifts lse by ~5e-8 in EVERY frame and
      // the scores by 4.8e-5 over T=1000 (tools/golden_full_gap.py; the bound is 1e-4). Kept: the fp64 routine cut to
      // the degree its purpose needs (1e-11 per term, errors of either sign), fp64 accumulation.
      const float mfw = (float)m;  // exact: m is the maximum of float32 values
      double sl = 0.0;
      // round 6: the reference's float32 arithmetic itself (numpy's float32 exp, its pairwise float32 sum, its float32 log:
      // np_f32.h, np_order_exp_sum<float>; the row is read a second time, from L1 / L2) -- a frame's log-probabilities are the
      // reference's bits. The polynomial variants below stay for CTCDEC_PRUNE_EXP=pk / f64.
      float l32 = 0.f, s32 = 0.f;
      const bool np32 = a.f32_np && np_order_exp_sum<float>((const float*)x4, V, mfw, lane, w, &s32);
      if (np32) l32 = np_log_f32(s32);
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        if (!np32 && k * 64 + lane < n4) {
          if (PK) {
            // (x - m in float32: one rounding, as in the reference's float32 `x - x_max`)
            const f32x2 e01 = exp_nonpos_f32x2((f32x2){r[k].x - mfw, r[k].y - mfw});
            const f32x2 e23 = exp_nonpos_f32x2((f32x2){r[k].z - mfw, r[k].w - mfw});
            sl += ((double)e01.x + (double)e01.y) + ((double)e23.x + (double)e23.y);
          } else {
            sl += (exp_nonpos9((double)r[k].x - m) + exp_nonpos9((double)r[k].y - m)) +
                  (exp_nonpos9((double)r[k].z - m) + exp_nonpos9((double)r[k].w - m));
          }
        }
      }
      if (!np32) {
        const double s = wave_sum(sl);
        lse = log_ge1(s);
      } else {
        lse = (double)l32;
      }
      // argmax of the log-probs = first maximum of the logits (x -> clip(x - m - lse) is monotone, and two
      // different fp32 logits never round to the same fp64 value after the two subtractions)
      int first = 0x7FFFFFFF;
#pragma unroll
      for (int k = NC - 1; k >= 0; --k) {
        const int v0 = (k * 64 + lane) * 4;
        if (k * 64 + lane < n4) {
          first = r[k].w == mfw ? v0 + 3 : first;
          first = r[k].z == mfw ? v0 + 2 : first;
          first = r[k].y == mfw ? v0 + 1 : first;
          first = r[k].x == mfw ? v0 : first;
        }
      }
      first = wave_min_i32(first);
      const double best = np32 ? to_logp_np32(mfw, false, mfw, l32) : to_logp((double)mfw, false, m, lse);
      // survivors: an fp32 screen that cannot miss (threshold rounded down, with a margin far above the fp64
      // rounding of the exact test), then the exact fp64 test only in the 256-label chunks that have a candidate
      const double xthr = m + lse + a.token_min_logp;
      // (a threshold at or below the clip keeps every label: decoder.py:444 tests the clipped values)
      // (float32 arithmetic: two roundings of up to half an ulp of |x - m| and |y| each -- a wider margin)
      const float pre = a.token_min_logp <= -34.538776394910684 ? -INFINITY
                                                                  : __double2float_rd(xthr - (np32 ? 3e-5 * (4.0 + fabs(xthr)) : 1e-6 * (1.0 + fabs(xthr))));
      uint32_t n = 0;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const bool in = k * 64 + lane < n4;
        const bool cand = in && (r[k].x >= pre || r[k].y >= pre || r[k].z >= pre || r[k].w >= pre);
        if (__ballot(cand)) {
          const int v0 = (k * 64 + lane) * 4;
          double y0 = -INFINITY, y1 = -INFINITY, y2 = -INFINITY, y3 = -INFINITY;
          if (cand && np32) {
            y0 = to_logp_np32(r[k].x, false, mfw, l32);
            y1 = to_logp_np32(r[k].y, false, mfw, l32);
            y2 = to_logp_np32(r[k].z, false, mfw, l32);
            y3 = to_logp_np32(r[k].w, false, mfw, l32);
          } else if (cand) {
            y0 = to_logp((double)r[k].x, false, m, lse);
            y1 = to_logp((double)r[k].y, false, m, lse);
            y2 = to_logp((double)r[k].z, false, m, lse);
            y3 = to_logp((double)r[k].w, false, m, lse);
          }
          const bool k0 = cand && y0 >= a.token_min_logp, k1 = cand && y1 >= a.token_min_logp;
          const bool k2 = cand && y2 >= a.token_min_logp, k3 = cand && y3 >= a.token_min_logp;
          const unsigned long long b0 = __ballot(k0), b1 = __ballot(k1), b2 = __ballot(k2), b3 = __ballot(k3);
          uint32_t pos = n + (uint32_t)(__popcll(b0 & lt) + __popcll(b1 & lt) + __popcll(b2 & lt) + __popcll(b3 & lt));
          if (k0) { if (pos < ms) { w.asc_id[pos] = (uint16_t)v0; w.asc_lp[pos] = y0; } ++pos; }
          if (k1) { if (pos < ms) { w.asc_id[pos] = (uint16_t)(v0 + 1); w.asc_lp[pos] = y1; } ++pos; }
          if (k2) { if (pos < ms) { w.asc_id[pos] = (uint16_t)(v0 + 2); w.asc_lp[pos] = y2; } ++pos; }
          if (k3) { if (pos < ms) { w.asc_id[pos] = (uint16_t)(v0 + 3); w.asc_lp[pos] = y3; } ++pos; }
          n += (uint32_t)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
        }
      }
      prune_finish(a, row, lane, w, n, best, first, /*reduced=*/true);
      return;
    }
    if (!isfinite(m)) m = 0.0;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      if (k * 64 + lane < n4)
        s += (exp((double)r[k].x - m) + exp((double)r[k].y - m)) + (exp((double)r[k].z - m) + exp((double)r[k].w - m));
    }
    s = wave_sum(s);
    mx = m;
    lse = log(s);
  }
  uint32_t n = 0;
  double best = -INFINITY;
  int best_id = 0x7FFFFFFF;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const bool in = k * 64 + lane < n4;
    const int v0 = (k * 64 + lane) * 4;
    double y0 = -INFINITY, y1 = -INFINITY, y2 = -INFINITY, y3 = -INFINITY;
    if (in && is_prob && a.f32_np) {  // (probability rows: numpy's float32 log of the float32 clip)
      y0 = to_logp_np32(r[k].x, true, 0.f, 0.f);
      y1 = to_logp_np32(r[k].y, true, 0.f, 0.f);
      y2 = to_logp_np32(r[k].z, true, 0.f, 0.f);
      y3 = to_logp_np32(r[k].w, true, 0.f, 0.f);
    } else if (in) {
      y0 = to_logp((double)r[k].x, is_prob, mx, lse);
      y1 = to_logp((double)r[k].y, is_prob, mx, lse);
      y2 = to_logp((double)r[k].z, is_prob, mx, lse);
      y3 = to_logp((double)r[k].w, is_prob, mx, lse);
    }
    if (in) {
      argmax_take(y0, v0, best, best_id);
      argmax_take(y1, v0 + 1, best, best_id);
      argmax_take(y2, v0 + 2, best, best_id);
      argmax_take(y3, v0 + 3, best, best_id);
    }
    const bool k0 = in && y0 >= a.token_min_logp, k1 = in && y1 >= a.token_min_logp;
    const bool k2 = in && y2 >= a.token_min_logp, k3 = in && y3 >= a.token_min_logp;
    const unsigned long long any = __ballot(k0 || k1 || k2 || k3);
    if (any) {  // ascending id order inside the chunk = (lane, element)
      const unsigned long long b0 = __ballot(k0), b1 = __ballot(k1), b2 = __ballot(k2), b3 = __ballot(k3);
      uint32_t pos = n + (uint32_t)(__popcll(b0 & lt) + __popcll(b1 & lt) + __popcll(b2 & lt) + __popcll(b3 & lt));
      if (k0) { if (pos < ms) { w.asc_id[pos] = (uint16_t)v0; w.asc_lp[pos] = y0; } ++pos; }
      if (k1) { if (pos < ms) { w.asc_id[pos] = (uint16_t)(v0 + 1); w.asc_lp[pos] = y1; } ++pos; }
      if (k2) { if (pos < ms) { w.asc_id[pos] = (uint16_t)(v0 + 2); w.asc_lp[pos] = y2; } ++pos; }
      if (k3) { if (pos < ms) { w.asc_id[pos] = (uint16_t)(v0 + 3); w.asc_lp[pos] = y3; } ++pos; }
      n += (uint32_t)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
    }
  }
  prune_finish(a, row, lane, w, n, best, best_id);
}
template <int NC, bool PK>
__global__ __launch_bounds__(PRUNE_WAVES * 64) void frame_prune_f32x4(PruneArgs a, uint32_t cap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t row = a.row_base + (int64_t)blockIdx.x * PRUNE_WAVES + wave;
  if (row >= a.row_base + a.n_rows) return;
  prune_row_f32x4<NC, PK>(a, row, prune_lds(smem, wave, (uint32_t)a.max_surv, cap), lane);
}
// the rows frame_prune_fast left on its list (a.slow_rows, a.overflow[3] of them), one wave per row
template <int NC>
__global__ __launch_bounds__(PRUNE_WAVES * 64) void frame_prune_f32x4_listed(PruneArgs a, uint32_t cap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const PruneLds w = prune_lds(smem, wave, (uint32_t)a.max_surv, cap);
  const uint32_t count = a.overflow[3];
  for (uint32_t k = blockIdx.x * PRUNE_WAVES + wave; k < count; k += gridDim.x * PRUNE_WAVES) {
    prune_row_f32x4<NC, true>(a, a.row_base + (int64_t)a.slow_rows[k], w, lane);
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
  }
}

// ---------------------------------------------------------------------------------------------
// frame_prune_fast: 64 frames per wavefront, in two phases
// ---------------------------------------------------------------------------------------------
// The per-row kernel above spends most of its instructions on what follows the log-softmax: finding the survivors, ordering
// them the way a CPython set would, writing them out -- a serial tail of a few hundred instructions that 63 of the 64 lanes
// sit through. Here a wave takes 64 consecutive rows and
//   phase A (one row at a time, 64 lanes x 16 logits, the next row's loads in flight): row maximum and sum, the sum of
//     exponentials (packed float32 polynomials, see exp_nonpos_f32x2m), the first maximum, and the labels that pass a float32
//     screen of the threshold, left UNORDERED in a 16-slot LDS column of the row; the row's scalars land in lane `row`;
//   phase B (one row per LANE): the exact fp64 test of the candidates, an insertion sort by label id, the CPython set order
//     (set_order_small.h: 48 table slots per lane) and the output records -- the serial tail, 64 rows at a time.
// Rows this does not cover (non-finite maximum or NaN, more than 16 candidates / 15 survivors, survivor counts at the
// max_surv bound) go onto a list and frame_prune_f32x4_listed runs the per-row code on them right behind this kernel.
#ifndef CTC_PF_ROWS
#define CTC_PF_ROWS 64
#endif
#ifndef CTC_PF_CAND
#define CTC_PF_CAND 16
#endif
constexpr int PF_ROWS = CTC_PF_ROWS;
constexpr int PF_CAND = CTC_PF_CAND;
constexpr size_t PF_LDS_IDS = (size_t)PF_ROWS * PF_CAND * 2;
constexpr size_t PF_LDS_X = (size_t)PF_ROWS * PF_CAND * 4;
constexpr int PF_MAX_LABELS = 4095;  // 64 lanes x 16 groups of four labels, less one: label ids have to fit the 12-bit set tables
constexpr size_t PF_LDS = PF_LDS_IDS + PF_LDS_X + (size_t)PF_ROWS * SMALL_SET_SLOTS * 2;  // 12 KiB: 13 waves per CU

// exp(d), d <= 0, two at a time, as exp_nonpos_f32x2 with the rounding and the scaling done by the 1.5 * 2^23 trick
// (t = d * log2(e) + magic holds round(d * log2 e) in its low mantissa bits: n = t - magic, and bits(t) << 23 is n in the
// exponent field: 2^n * p is one integer add, no v_rndne / v_cvt / v_ldexp) and the degree-7 polynomial. Same operations
// on the host (20 M arguments in [-14, 0]): mean relative error -5.8e-10, mean |error| 2.2e-8, maximum 7.7e-8
// (degree 8: 1.5e-11 / 2.2e-8 / 7.4e-8; libm's expf: 3.5e-11 / 2.1e-8).
__device__ __forceinline__ f32x2 exp_nonpos_f32x2m(f32x2 d) {
  d.x = fmaxf(d.x, -80.0f);  // exp(-80) = 1.8e-35 does not register in a sum >= 1; also maps -inf masks
  d.y = fmaxf(d.y, -80.0f);
  const f32x2 magic = (f32x2)(12582912.0f);
  const f32x2 t = __builtin_elementwise_fma(d, (f32x2)(1.44269504088896340736f), magic);
  const f32x2 n = t - magic;
  f32x2 r = __builtin_elementwise_fma(n, (f32x2)(-0.693145751953125f), d);
  r = __builtin_elementwise_fma(n, (f32x2)(-1.42860682030941723212e-6f), r);
  f32x2 p = (f32x2)(1.98412698412698413e-04f);                              // 1/7!
  p = __builtin_elementwise_fma(p, r, (f32x2)(1.38888888888888889e-03f));   // 1/6!
  p = __builtin_elementwise_fma(p, r, (f32x2)(8.33333333333333333e-03f));   // 1/5!
  p = __builtin_elementwise_fma(p, r, (f32x2)(4.16666666666666667e-02f));   // 1/4!
  p = __builtin_elementwise_fma(p, r, (f32x2)(1.66666666666666667e-01f));   // 1/3!
  p = __builtin_elementwise_fma(p, r, (f32x2)(0.5f));
  p = __builtin_elementwise_fma(p, r, (f32x2)(1.0f));
  p = __builtin_elementwise_fma(p, r, (f32x2)(1.0f));
  f32x2 out;
  out.x = __uint_as_float(__float_as_uint(p.x) + (__float_as_uint(t.x) << 23));
  out.y = __uint_as_float(__float_as_uint(p.y) + (__float_as_uint(t.y) << 23));
  return out;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float ident, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// v_max3_f32 / v_max_f32_dpp by hand: fmaxf() makes the compiler quiet each operand first (v_max_f32 x, x, x), which doubles
// the instructions of a maximum over loaded values. NaNs need no care here: a row that holds one is recognised by its sum
// and handed to the per-row kernel.
__device__ __forceinline__ float min3_raw(float a, float b, float c) {
  float o;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
  return o;
}
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
  float o;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
  return o;
}
// (lanes a DPP step has no source for keep their value: the destination is the second operand; s_nop 1: the two wait
// states a DPP read needs after the VALU write of its source -- the compiler cannot see into the asm)
#define CTC_MAX_DPP(V, CTRL) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " CTRL : "+v"(V))
__device__ __forceinline__ float wave_max_f32(float v) {
  CTC_MAX_DPP(v, "row_shr:1 row_mask:0xf bank_mask:0xf");
  CTC_MAX_DPP(v, "row_shr:2 row_mask:0xf bank_mask:0xf");
  CTC_MAX_DPP(v, "row_shr:4 row_mask:0xf bank_mask:0xf");
  CTC_MAX_DPP(v, "row_shr:8 row_mask:0xf bank_mask:0xf");
  CTC_MAX_DPP(v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
  CTC_MAX_DPP(v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
  asm volatile("s_nop 1" ::: "memory");
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
  v += dpp_f32<0x111, 0xf>(0.f, v);
  v += dpp_f32<0x112, 0xf>(0.f, v);
  v += dpp_f32<0x114, 0xf>(0.f, v);
  v += dpp_f32<0x118, 0xf>(0.f, v);
  v += dpp_f32<0x142, 0xa>(0.f, v);
  v += dpp_f32<0x143, 0xc>(0.f, v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {  // set bits of `mask` below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// one lane's 48 table slots, interleaved with the other lanes' (slot k of lane l at [k * 64 + l])
struct LaneTab {
  uint16_t* t;
  __device__ __forceinline__ uint16_t get(uint32_t k) const { return t[k * PF_ROWS]; }
  __device__ __forceinline__ void put(uint32_t k, uint16_t v) { t[k * PF_ROWS] = v; }
};

// DT: ctcdec_dtype of the rows -- 0 float32 (NC 16-byte loads of four labels per lane), 2 float16 / 3 bfloat16 (NC / 2 loads
// of eight labels, widened exactly to the same NC float4 groups: the arithmetic below never knows the difference)
// (native vector types: what a load returns stays in its registers untouched until the row is worked on -- and, unlike the HIP
//  vector classes, they can be read through a pointer with an address space)
typedef float PfF4 __attribute__((ext_vector_type(4)));
typedef uint32_t PfU4 __attribute__((ext_vector_type(4)));
template <int DT>
struct PfRaw {
  typedef PfF4 type;
};
template <>
struct PfRaw<2> {
  typedef PfU4 type;
};
template <>
struct PfRaw<3> {
  typedef PfU4 type;
};
template <int DT>
__device__ __forceinline__ float pf_widen(uint32_t h16) {
  return DT == 2 ? __half2float(__ushort_as_half((unsigned short)h16)) : __uint_as_float(h16 << 16);
}

// numpy's float32 exp (np_f32.h: np_exp_f32) for the arguments of a clean row's log-softmax, t = x - max <= 0 or -inf, without
// its special-case branches: the same operations in the same order, the underflow rule as a select.
__device__ __forceinline__ float np_exp_nonpos_dev(float t) {
#pragma clang fp contract(off)
  const float tc = fmaxf(t, -104.0f);  // (keeps the polynomial's inputs finite; t <= xmin is answered by the select below)
  float q = tc * 1.442695040888963407359924681001892137f;
  q = (q + 12582912.0f) - 12582912.0f;
  float r = fmaf(q, -6.93145752e-1f, tc);
  r = fmaf(q, -1.42860677e-6f, r);
  float num = fmaf(5.082762527590693718096e-04f, r, 6.757896990527504603057e-03f);
  num = fmaf(num, r, 5.114512081637298353406e-02f);
  num = fmaf(num, r, 2.473615434895520810817e-01f);
  num = fmaf(num, r, 7.257664613233124478488e-01f);
  num = fmaf(num, r, 9.999999999980870924916e-01f);
  float den = fmaf(2.159509375685829852307e-02f, r, -2.742335390411667452936e-01f);
  den = fmaf(den, r, 1.000000000000000000000e+00f);
  const float v = ldexpf(num / den, (int)q);
  return t <= -103.97208404541015625f ? 0.0f : v;
}
// The same for the four labels of a lane's group, two at a time in packed float32 instructions (v_pk_mul / add / fma_f32: each
// component rounds exactly like the scalar instruction; what has no packed form -- the maximum, the reciprocal, the scaling
// and the select -- stays scalar). A wave issues one vector instruction every ~5 cycles whatever it is (tools/micro/
// valu_rates.hip), so half the instructions is what counts. The division num / den is IEEE-exact in both forms:
//   default: the compiler's division (v_div_scale / v_rcp / four fused steps / v_div_fmas / v_div_fixup);
//   CTC_NP_SHORT_DIV: v_rcp + one Newton step + quotient + one fused correction -- den is in [0.8, 1.2] and num in [0.7, 1.5]
//   here, nothing needs scaling or fixing up; correct rounding of that sequence on THIS chip's v_rcp_f32 is checked for every
//   reduced argument by tools/micro/np_div_check.hip (round 6: 0 of 2.1e9 differ) before the macro may be set.
__device__ __forceinline__ f32x2 np_exp_nonpos_pk(f32x2 t) {
#pragma clang fp contract(off)
  f32x2 tc;
  tc.x = fmaxf(t.x, -104.0f);
  tc.y = fmaxf(t.y, -104.0f);
  const f32x2 magic = (f32x2)(12582912.0f);
  f32x2 q = tc * (f32x2)(1.442695040888963407359924681001892137f);
  q = q + magic;
  q = q - magic;
  f32x2 r = __builtin_elementwise_fma(q, (f32x2)(-6.93145752e-1f), tc);
  r = __builtin_elementwise_fma(q, (f32x2)(-1.42860677e-6f), r);
  f32x2 num = __builtin_elementwise_fma((f32x2)(5.082762527590693718096e-04f), r, (f32x2)(6.757896990527504603057e-03f));
  num = __builtin_elementwise_fma(num, r, (f32x2)(5.114512081637298353406e-02f));
  num = __builtin_elementwise_fma(num, r, (f32x2)(2.473615434895520810817e-01f));
  num = __builtin_elementwise_fma(num, r, (f32x2)(7.257664613233124478488e-01f));
  num = __builtin_elementwise_fma(num, r, (f32x2)(9.999999999980870924916e-01f));
  f32x2 den = __builtin_elementwise_fma((f32x2)(2.159509375685829852307e-02f), r, (f32x2)(-2.742335390411667452936e-01f));
  den = __builtin_elementwise_fma(den, r, (f32x2)(1.0f));
  f32x2 p;
#ifdef CTC_NP_SHORT_DIV
  f32x2 y0;
  y0.x = __builtin_amdgcn_rcpf(den.x);
  y0.y = __builtin_amdgcn_rcpf(den.y);
  const f32x2 e = __builtin_elementwise_fma(-den, y0, (f32x2)(1.0f));
  const f32x2 y = __builtin_elementwise_fma(e, y0, y0);
  const f32x2 q0 = num * y;
  const f32x2 rem = __builtin_elementwise_fma(-den, q0, num);
  p = __builtin_elementwise_fma(rem, y, q0);
#else
  p.x = num.x / den.x;
  p.y = num.y / den.y;
#endif
  f32x2 v;
  v.x = ldexpf(p.x, (int)q.x);
  v.y = ldexpf(p.y, (int)q.y);
  v.x = t.x <= -103.97208404541015625f ? 0.0f : v.x;
  v.y = t.y <= -103.97208404541015625f ? 0.0f : v.y;
  return v;
}
// ... and for rows all of whose arguments are in (-87, 0] -- every row of ordinary logits; a row that reaches further down is
// handed to the per-row kernel --: no clamp, no underflow select, and the scaling by 2^k as one integer add into the exponent
// field (the result is a normal number there: exactly what ldexpf gives; k sits in the low mantissa bits of q + 1.5 * 2^23).
__device__ __forceinline__ f32x2 np_exp_nonpos_pk_fast(f32x2 t) {
#pragma clang fp contract(off)
  const f32x2 magic = (f32x2)(12582912.0f);
  f32x2 q = t * (f32x2)(1.442695040888963407359924681001892137f);
  const f32x2 qm = q + magic;
  q = qm - magic;
  f32x2 r = __builtin_elementwise_fma(q, (f32x2)(-6.93145752e-1f), t);
  r = __builtin_elementwise_fma(q, (f32x2)(-1.42860677e-6f), r);
  f32x2 num = __builtin_elementwise_fma((f32x2)(5.082762527590693718096e-04f), r, (f32x2)(6.757896990527504603057e-03f));
  num = __builtin_elementwise_fma(num, r, (f32x2)(5.114512081637298353406e-02f));
  num = __builtin_elementwise_fma(num, r, (f32x2)(2.473615434895520810817e-01f));
  num = __builtin_elementwise_fma(num, r, (f32x2)(7.257664613233124478488e-01f));
  num = __builtin_elementwise_fma(num, r, (f32x2)(9.999999999980870924916e-01f));
  f32x2 den = __builtin_elementwise_fma((f32x2)(2.159509375685829852307e-02f), r, (f32x2)(-2.742335390411667452936e-01f));
  den = __builtin_elementwise_fma(den, r, (f32x2)(1.0f));
  // num / den, IEEE-exact: v_rcp_f32, quotient, one fused correction (den in [0.8, 1.2], num in [0.7, 1.5]: nothing to scale or
  // fix up, and the reciprocal's 1 ulp is enough for the correction term). Checked against the compiler's division for EVERY
  // reduced argument on this chip: tools/micro/np_div_check.hip, profiles/r06_np_div_check.txt -- 0 of 2 104 533 978 differ
  // (with and without a Newton step on the reciprocal).
  f32x2 y0;
  y0.x = __builtin_amdgcn_rcpf(den.x);
  y0.y = __builtin_amdgcn_rcpf(den.y);
  const f32x2 q0 = num * y0;
  const f32x2 rem = __builtin_elementwise_fma(-den, q0, num);
  const f32x2 p = __builtin_elementwise_fma(rem, y0, q0);
  f32x2 v;
  v.x = __uint_as_float(__float_as_uint(p.x) + (__float_as_uint(qm.x) << 23));
  v.y = __uint_as_float(__float_as_uint(p.y) + (__float_as_uint(qm.y) << 23));
  return v;
}
// The same for N pairs at once, stage by stage. Why: on gfx950 a vector instruction that reads the result of the packed-float32
// instruction right before it costs a wait state, and the compiler -- which schedules one exponential's chain after the other --
// pays it with an `s_nop` between nearly every two instructions of the sixteen chains of a row (round 6: 102 of the 340
// instructions of a row's exponentials were s_nop, and a wave issues one instruction per ~5 cycles whatever it is). Written
// stage-major with the scheduler fenced between stages, consecutive instructions belong to different chains and nothing waits.
// Same operations per element, in the same order: the values are np_exp_nonpos_pk_fast's bit for bit.
template <int N>
__device__ __forceinline__ void np_exp_nonpos_pk_fast_n(const f32x2 (&t)[N], f32x2 (&v)[N]) {
#pragma clang fp contract(off)
  const f32x2 magic = (f32x2)(12582912.0f);
  f32x2 q[N], qm[N], r[N], num[N], den[N], y0[N], q0[N];
#define CTC_STAGE(expr)                      \
  _Pragma("unroll") for (int i = 0; i < N; ++i) { expr; } \
  __builtin_amdgcn_sched_barrier(0);
  CTC_STAGE(q[i] = t[i] * (f32x2)(1.442695040888963407359924681001892137f))
  CTC_STAGE(qm[i] = q[i] + magic)
  CTC_STAGE(q[i] = qm[i] - magic)
  CTC_STAGE(r[i] = __builtin_elementwise_fma(q[i], (f32x2)(-6.93145752e-1f), t[i]))
  CTC_STAGE(r[i] = __builtin_elementwise_fma(q[i], (f32x2)(-1.42860677e-6f), r[i]))
  CTC_STAGE(num[i] = __builtin_elementwise_fma((f32x2)(5.082762527590693718096e-04f), r[i], (f32x2)(6.757896990527504603057e-03f));
            den[i] = __builtin_elementwise_fma((f32x2)(2.159509375685829852307e-02f), r[i], (f32x2)(-2.742335390411667452936e-01f)))
  CTC_STAGE(num[i] = __builtin_elementwise_fma(num[i], r[i], (f32x2)(5.114512081637298353406e-02f));
            den[i] = __builtin_elementwise_fma(den[i], r[i], (f32x2)(1.0f)))
  CTC_STAGE(num[i] = __builtin_elementwise_fma(num[i], r[i], (f32x2)(2.473615434895520810817e-01f));
            y0[i].x = __builtin_amdgcn_rcpf(den[i].x))
  CTC_STAGE(num[i] = __builtin_elementwise_fma(num[i], r[i], (f32x2)(7.257664613233124478488e-01f));
            y0[i].y = __builtin_amdgcn_rcpf(den[i].y))
  CTC_STAGE(num[i] = __builtin_elementwise_fma(num[i], r[i], (f32x2)(9.999999999980870924916e-01f)))
  CTC_STAGE(q0[i] = num[i] * y0[i])
  CTC_STAGE(num[i] = __builtin_elementwise_fma(-den[i], q0[i], num[i]))  // (the remainder)
  CTC_STAGE(q0[i] = __builtin_elementwise_fma(num[i], y0[i], q0[i]))
  CTC_STAGE(v[i].x = __uint_as_float(__float_as_uint(q0[i].x) + (__float_as_uint(qm[i].x) << 23));
            v[i].y = __uint_as_float(__float_as_uint(q0[i].y) + (__float_as_uint(qm[i].y) << 23)))
#undef CTC_STAGE
}
// index of element i of a row in the exchange buffer: eight floats of padding per 128 (the accumulator lanes of different
// leaves then read different banks)
__device__ __forceinline__ int np_pad(int i) { return i + ((i >> 7) << 3); }
constexpr size_t PF_TABS_BYTES = (size_t)PF_ROWS * SMALL_SET_SLOTS * 2;
// LDS a block needs beyond PF_LDS in numpy-order mode: the leaf sums of its 64 rows, and the exchange buffer of one row when it
// does not fit the set tables' bytes (which are idle during phase A)
__host__ __device__ inline size_t pf_np_rowbuf_bytes(int nc) { return ((size_t)(nc * 256 + nc * 16 + 16) * 4 + 15) & ~(size_t)15; }
__host__ __device__ inline size_t pf_np_extra_lds(int nc, int n_leaf) {
  const size_t ls = (size_t)n_leaf * PF_ROWS * 4;
  return ls + (pf_np_rowbuf_bytes(nc) <= PF_TABS_BYTES ? 0 : pf_np_rowbuf_bytes(nc));
}

// AL: rows start on 16-byte boundaries and hold a multiple of four (16-bit: eight) labels: 16-byte loads. !AL (float32 only):
// any label count and any 4-byte-aligned rows -- each lane fetches its four consecutive labels one by one (the common
// "1024 BPE pieces + blank = 1025 labels" shape); labels past the row's end read as -inf.
// NP (float32 rows): the sum of exponentials in the reference's own float32 arithmetic -- numpy's float32 exp per label, its
// pairwise summation order (np_sum.h) and, in phase B, its float32 log: a row's exponentials cross LDS once so that lane
// (leaf, j) adds the elements numpy's accumulator j of that leaf adds, in numpy's order; the leaf sums wait in LDS for phase B,
// where every lane combines its row's along the recursion's tree (a.np_prog). A frame's log-probabilities are then the
// reference's bits, and the decode of float32 logits equals the reference's to the last bit of every score.
template <int NC, int DT, bool AL = true, bool NP = false>
#ifndef CTC_PF_WAVES
#define CTC_PF_WAVES 3
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NC <= 4 ? CTC_PF_WAVES : 1, NC <= 4 ? CTC_PF_WAVES : 8))) void frame_prune_fast(PruneArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int64_t row_lo = a.row_base + (int64_t)blockIdx.x * PF_ROWS;
  const int64_t row_end = a.row_base + a.n_rows;
  if (row_lo >= row_end) return;
  const int nrows = (int)(row_end - row_lo < (int64_t)PF_ROWS ? row_end - row_lo : (int64_t)PF_ROWS);
  uint16_t* ar_id = (uint16_t*)smem;                       // [PF_CAND][64 rows]
  float* ar_x = (float*)(smem + PF_LDS_IDS);               // [PF_CAND][64 rows]
  uint16_t* tabs = (uint16_t*)(smem + PF_LDS_IDS + PF_LDS_X);  // [SMALL_SET_SLOTS][64 rows]
  static_assert(!NP || DT == 0, "numpy-order sums: float32 rows");
  float* np_ls = (float*)(smem + PF_LDS);        // NP: [n_leaf][64 rows] leaf sums
  float* np_rb = pf_np_rowbuf_bytes(NC) <= PF_TABS_BYTES ? (float*)tabs : (float*)(smem + PF_LDS + (a.np_uniform8 ? 0 : (size_t)a.np_n_leaf * PF_ROWS * 4));
  constexpr bool WIDE = DT == 0;                 // float32 rows
  constexpr int NL = WIDE ? NC : NC / 2;         // 16-byte loads per lane and row
  constexpr int PER = WIDE ? 4 : 8;              // labels per load
  static_assert(WIDE || NC % 2 == 0, "16-bit rows: two float4 groups per load");
  static_assert(AL || WIDE, "element-wise loads: float32 rows only");
  typedef typename PfRaw<DT>::type Raw;
  const int V = a.n_labels;
  const int n4 = AL ? V / PER : (V + 3) / 4;     // loads (groups of four labels) per row
  const float tminf = (float)a.token_min_logp;
  // label id of element e of group k in this lane; is group k inside the row?
  auto id_of = [&](int k, int e) { return WIDE ? (k * 64 + lane) * 4 + e : ((k >> 1) * 64 + lane) * 8 + (k & 1) * 4 + e; };
  auto in_row = [&](int k) { return (WIDE ? k : (k >> 1)) * 64 + lane < n4; };

  // rows are walked in order: the utterance of the first one by bisection, the rest by stepping.
  // Round 5: the utterance bookkeeping is SCALAR (uniform loads from the constant address space: s_load, counted by lgkmcnt)
  // and the rows are read through a GLOBAL-address-space pointer. Until then the rows' base pointer -- fetched from the
  // per-utterance pointer array by a vector load -- made every row load a flat_load and put an unconditional
  // `s_waitcnt vmcnt(0)` in front of each row's arithmetic: the two rows "in flight" behind the one being worked on were waited
  // for the moment they were requested, and every row paid a full HBM round trip (3.8 TB/s at V = 1024, ~8 500 cycles a row).
  typedef const int64_t __attribute__((address_space(4))) * ConstI64;
  typedef const void* const __attribute__((address_space(4))) * ConstPtrs;
  typedef const Raw __attribute__((address_space(1))) * GlobalRaw;
  const ConstI64 row0_c = (ConstI64)a.utt_row0;
  const ConstPtrs logits_c = (ConstPtrs)a.utt_logits;
  int u = __builtin_amdgcn_readfirstlane(find_utt(a.utt_row0, a.n_utts, row_lo));
  int64_t u_r0 = row0_c[u], u_r1 = row0_c[u + 1];
  const char __attribute__((address_space(1))) * u_base = (const char __attribute__((address_space(1)))*)logits_c[u];
  auto load_row = [&](int64_t row, Raw(&r)[NL]) {
    while (row >= u_r1) {
      ++u;
      u_r0 = u_r1;
      u_r1 = row0_c[u + 1];
      u_base = (const char __attribute__((address_space(1)))*)logits_c[u];
    }
    const GlobalRaw x4 = (GlobalRaw)(u_base + (size_t)(row - u_r0) * V * (WIDE ? 4 : 2));
    // Every load is UNCONDITIONAL (a lane past the row's end re-reads the row's first group; `widen` puts -inf there when the
    // row is worked on): a load behind a lane mask is a branch, and at the join the compiler can no longer count how many loads
    // are outstanding -- it waits for all of them, the prefetched rows included.
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i4 = k * 64 + lane;
      if constexpr (!AL) {
        const float __attribute__((address_space(1)))* xf = (const float __attribute__((address_space(1)))*)x4;
        const int e0 = i4 * 4;
        Raw v;
        v.x = xf[e0 < V ? e0 : 0];
        v.y = xf[e0 + 1 < V ? e0 + 1 : 0];
        v.z = xf[e0 + 2 < V ? e0 + 2 : 0];
        v.w = xf[e0 + 3 < V ? e0 + 3 : 0];
        r[k] = v;
      } else {
        r[k] = x4[i4 < n4 ? i4 : 0];
      }
    }
  };
  auto widen = [&](const Raw(&raw)[NL], float4(&r)[NC]) {
    const float ninf = -INFINITY;
    if constexpr (WIDE) {
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        r[k] = make_float4(raw[k].x, raw[k].y, raw[k].z, raw[k].w);
        if constexpr (!AL) {  // labels past the end of the row
          const int e0 = (k * 64 + lane) * 4;
          if (e0 >= V) r[k].x = ninf;
          if (e0 + 1 >= V) r[k].y = ninf;
          if (e0 + 2 >= V) r[k].z = ninf;
          if (e0 + 3 >= V) r[k].w = ninf;
        } else if (!in_row(k)) {
          r[k] = make_float4(ninf, ninf, ninf, ninf);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        r[2 * k] = make_float4(pf_widen<DT>(raw[k].x & 0xFFFFu), pf_widen<DT>(raw[k].x >> 16), pf_widen<DT>(raw[k].y & 0xFFFFu),
                               pf_widen<DT>(raw[k].y >> 16));
        r[2 * k + 1] = make_float4(pf_widen<DT>(raw[k].z & 0xFFFFu), pf_widen<DT>(raw[k].z >> 16), pf_widen<DT>(raw[k].w & 0xFFFFu),
                                   pf_widen<DT>(raw[k].w >> 16));
        if (!in_row(2 * k)) {
          r[2 * k] = make_float4(ninf, ninf, ninf, ninf);
          r[2 * k + 1] = make_float4(ninf, ninf, ninf, ninf);
        }
      }
    }
  };

  // what phase A leaves for row i, in lane i
  float my_m = 0.f, my_rs = 0.f, my_s32 = 1.0f;
  double my_s = 1.0;
  int my_first = 0;
  uint32_t my_cnt = 0;

  // numpy's accumulators over the exchange buffer of row i (NP): lane (leaf, j) adds the elements accumulator j of that leaf
  // adds, in numpy's order; the leaf sums go to LDS for phase B. The leaves of the first sweep are fetched once per block.
  int np_off0 = 0, np_len0 = 0;
  if constexpr (NP) {
    const int leaf = lane >> 3;
    if (leaf < a.np_n_leaf) {
      np_off0 = a.np_leaf[2 * leaf];
      np_len0 = a.np_leaf[2 * leaf + 1];
    }
  }
  auto np_accumulate = [&](int i) {
#ifdef CTC_NP_DIAG_NOEXCH
    if (lane < a.np_n_leaf) np_ls[lane * PF_ROWS + i] = 1.0f;
    return;
#endif
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int g = lane >> 3, j = lane & 7;
    const int nl = a.np_n_leaf;
    for (int base = 0; base < nl; base += 8) {
      const int leaf = base + g;
      const bool mine = leaf < nl;
      int off = np_off0, len = np_len0;
      if (base > 0) {
        off = len = 0;
        if (mine) {
          off = a.np_leaf[2 * leaf];
          len = a.np_leaf[2 * leaf + 1];
        }
      }
      float res = 0.f;
      if (len == 128 && (off & 127) == 0) {  // (the usual leaf: sixteen terms per accumulator, constant offsets from one address)
        const float* q = np_rb + np_pad(off) + j;
        float t0 = q[0], t1 = q[8], t2 = q[16], t3 = q[24], t4 = q[32], t5 = q[40], t6 = q[48], t7 = q[56];
        float t8 = q[64], t9 = q[72], t10 = q[80], t11 = q[88], t12 = q[96], t13 = q[104], t14 = q[112], t15 = q[120];
        float rr = t0 + t1;
        rr = rr + t2; rr = rr + t3; rr = rr + t4; rr = rr + t5; rr = rr + t6; rr = rr + t7; rr = rr + t8;
        rr = rr + t9; rr = rr + t10; rr = rr + t11; rr = rr + t12; rr = rr + t13; rr = rr + t14; rr = rr + t15;
        res = rr;
      } else if (len >= 8) {
        const int body = len - (len & 7);
        float rr = np_rb[np_pad(off + j)];
        for (int t = 8; t < body; t += 8) rr = rr + np_rb[np_pad(off + t + j)];
        res = rr;
      }
      // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)) inside the group of eight lanes: additions commute, an xor butterfly is that tree
      res = res + dpp_f32<0xB1, 0xf>(0.f, res);   // quad_perm [1, 0, 3, 2]
      res = res + dpp_f32<0x4E, 0xf>(0.f, res);   // quad_perm [2, 3, 0, 1]
      res = res + dpp_f32<0x141, 0xf>(0.f, res);  // row_half_mirror (both quads hold their sums in every lane by now)
      if ((len & 7) != 0 || len < 8) {  // leftovers (and leaves shorter than eight: a plain loop from 0), one by one
        const int body = len >= 8 ? len - (len & 7) : 0;
        if (len < 8) res = 0.f;
        for (int t = body; t < len; ++t) res = res + np_rb[np_pad(off + t)];
      }
      if (a.np_uniform8) {
        // up to eight leaves of 128, one per group of eight lanes (every lane of a group holds its leaf's sum; groups without a
        // leaf hold 0): numpy's tree ((S0+S1)+(S2+S3)) + ((S4+S5)+(S6+S7)) by three cross-lane additions -- partners 8 lanes
        // apart inside a row of sixteen, then rows 0 -> 1 and 2 -> 3, then row 1 -> row 3 (additions commute) -- and the row's
        // sum lands in lane i's register: no leaf sums in LDS, which buys the block's two further waves per CU back
        float t = res + dpp_f32<0x128, 0xf>(0.f, res);  // row_ror:8
        t = t + dpp_f32<0x142, 0xa>(0.f, t);              // row_bcast:15 into rows 1 and 3
        t = t + dpp_f32<0x143, 0xc>(0.f, t);              // row_bcast:31 into rows 2 and 3
        const float total = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 63));
        if (lane == i) my_s32 = total;
      } else if (mine && j == 0) {
        np_ls[leaf * PF_ROWS + i] = res;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();  // (the next row's exponentials go to the same buffer)
  };

  auto phase_a = [&](int i, const Raw(&raw)[NL]) -> uint32_t {
    float4 r[NC];
    widen(raw, r);
    float mf = -INFINITY, rsf;
#pragma unroll
    for (int k = 0; k < NC; ++k) mf = max3_raw(max3_raw(mf, r[k].x, r[k].y), r[k].z, r[k].w);
    if (!AL) {  // (labels past the end of the row hold -inf for the maximum: nothing for the sum)
      rsf = 0.f;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const int e0 = (k * 64 + lane) * 4;
        rsf += ((e0 < V ? r[k].x : 0.f) + (e0 + 1 < V ? r[k].y : 0.f)) + ((e0 + 2 < V ? r[k].z : 0.f) + (e0 + 3 < V ? r[k].w : 0.f));
      }
    } else if (n4 == NL * 64) {  // (see prune_row_f32x4 on this sum)
      f32x2 rs2 = (f32x2)(0.f);
#pragma unroll
      for (int k = 0; k < NC; ++k) rs2 += (f32x2){r[k].x, r[k].y} + (f32x2){r[k].z, r[k].w};
      rsf = rs2.x + rs2.y;
    } else {  // lanes past the row hold -inf for the maximum: nothing for the sum
      rsf = 0.f;
#pragma unroll
      for (int k = 0; k < NC; ++k) rsf += in_row(k) ? (r[k].x + r[k].y) + (r[k].z + r[k].w) : 0.f;
    }
    const float m = wave_max_f32(mf);
    const float rs = wave_sum_f32(rsf);
    uint32_t cnt = 0xFFFFu;  // "not a clean row": the per-row kernel takes it
    int first = 0;
    double s = 1.0;
    bool clean = isfinite(m) && rs == rs;
    if constexpr (NP) {
      // numpy-order sums: a row whose smallest logit is 87 or more below its maximum (exponentials that are denormal or zero in
      // float32; -inf masks; labels past the row's end do not count) takes the per-row kernel, whose exponential has the
      // underflow rules -- the 16 exponentials below then need neither a clamp nor a select, and scale by an integer add
      float lo = INFINITY;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        if constexpr (AL) {
          if (n4 == NL * 64 || in_row(k)) lo = min3_raw(min3_raw(lo, r[k].x, r[k].y), r[k].z, r[k].w);
        } else {
          const int e0 = (k * 64 + lane) * 4;
          lo = fminf(lo, fminf(fminf(e0 < V ? r[k].x : lo, e0 + 1 < V ? r[k].y : lo), fminf(e0 + 2 < V ? r[k].z : lo, e0 + 3 < V ? r[k].w : lo)));
        }
      }
      if (clean && __ballot(!(lo - m > -87.0f)) != 0ull) clean = false;
    }
    if (clean) {
      if constexpr (NP) {
        // numpy's float32 exp of every label, in place, into the exchange buffer; their sum in any order for the screen (the
        // exact one -- numpy's accumulators, np_accumulate below -- is taken AFTER the screen: the wave issues in order, so the
        // screen's work stands between the LDS writes here and the reads there instead of a wait)
        f32x2 approx = (f32x2)(0.f);
        const f32x2 mm2 = (f32x2)(m);
#if !defined(CTC_NP_DIAG_PKEXP) && !defined(CTC_NP_CHAINWISE)
        // (the exponentials of CTC_NP_IL groups of four labels at a time, stage by stage: np_exp_nonpos_pk_fast_n)
#ifndef CTC_NP_IL
#define CTC_NP_IL 2
#endif
        constexpr int IL = NC % CTC_NP_IL == 0 ? CTC_NP_IL : 1;
        f32x2 ex[NC][2];
#pragma unroll
        for (int k0 = 0; k0 < NC; k0 += IL) {
          f32x2 tin[2 * IL], tout[2 * IL];
#pragma unroll
          for (int j = 0; j < IL; ++j) {
            tin[2 * j] = (f32x2){r[k0 + j].x, r[k0 + j].y} - mm2;
            tin[2 * j + 1] = (f32x2){r[k0 + j].z, r[k0 + j].w} - mm2;
          }
          np_exp_nonpos_pk_fast_n<2 * IL>(tin, tout);
#pragma unroll
          for (int j = 0; j < IL; ++j) {
            ex[k0 + j][0] = tout[2 * j];
            ex[k0 + j][1] = tout[2 * j + 1];
          }
        }
#endif
#pragma unroll
        for (int k = 0; k < NC; ++k) {
#ifdef CTC_NP_DIAG_PKEXP  // diagnostics only (what the exact exponentials cost): round 5's polynomial in their place
          const f32x2 e01 = exp_nonpos_f32x2m((f32x2){r[k].x, r[k].y} - mm2), e23 = exp_nonpos_f32x2m((f32x2){r[k].z, r[k].w} - mm2);
#elif defined(CTC_NP_CHAINWISE)
          const f32x2 e01 = np_exp_nonpos_pk_fast((f32x2){r[k].x, r[k].y} - mm2), e23 = np_exp_nonpos_pk_fast((f32x2){r[k].z, r[k].w} - mm2);
#else
          const f32x2 e01 = ex[k][0], e23 = ex[k][1];
#endif
          bool inside = true;  // (labels past the row's end hold -inf: garbage here, never read by the leaves, kept out of the screen's sum)
          if constexpr (AL) inside = n4 == NL * 64 || in_row(k);
          else inside = (k * 64 + lane) * 4 + 3 < V;
          if (inside) approx += e01 + e23;
#ifndef CTC_NP_DIAG_NOEXCH
          *(float4*)(np_rb + np_pad((k * 64 + lane) * 4)) = make_float4(e01.x, e01.y, e23.x, e23.y);
#endif
        }
        s = (double)wave_sum_f32(approx.x + approx.y);
      } else {
      // sum of exponentials: float32 per lane (16 terms), fp64 across the lanes
      f32x2 acc = (f32x2)(0.f);
      const f32x2 mm = (f32x2)(m);
#pragma unroll
      for (int k = 0; k < NC; ++k) {
#ifdef CTC_PF_SKIP_EXP  // diagnostics only (timing the rest of phase A): no exponentials, a made-up sum
        acc += ((f32x2){r[k].x, r[k].y} - mm) * (f32x2)(1e-9f) + ((f32x2){r[k].z, r[k].w} - mm) * (f32x2)(1e-9f);
#else
        acc += exp_nonpos_f32x2m((f32x2){r[k].x, r[k].y} - mm);
        acc += exp_nonpos_f32x2m((f32x2){r[k].z, r[k].w} - mm);
#endif
      }
      s = wave_sum((double)(acc.x + acc.y));
#ifdef CTC_PF_SKIP_EXP
      s = 1.02 + s * 1e-30;
#endif
      }
      // candidates: a float32 screen that cannot miss a survivor -- the threshold on the logits from a float32 logarithm
      // (|error| < 1e-6 (1 + lse)), lowered by a margin two orders above that and above the rounding of the sum
      const float xthr = (m + __logf((float)s)) + tminf;
      const float pre = xthr - 3e-5f * (4.0f + fabsf(xthr));
      cnt = 0;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = e == 0 ? r[k].x : e == 1 ? r[k].y : e == 2 ? r[k].z : r[k].w;
          const bool c = x >= pre;
          const uint64_t mk = __ballot(c);
          if (mk) {
            const uint32_t pos = cnt + lanes_below(mk);
            if (c && pos < (uint32_t)PF_CAND) {
              ar_id[pos * PF_ROWS + i] = (uint16_t)id_of(k, e);
              ar_x[pos * PF_ROWS + i] = x;
            }
            cnt += (uint32_t)__popcll(mk);
          }
        }
      }
      // the first maximum (numpy.argmax): when the maximum passes the screen, every label that ties for it is among the
      // candidates and phase B picks the smallest id; only a row whose maximum is below the threshold is searched here,
      // from the lane masks of `== m`
      if (cnt == 0) {
        first = 0x7FFFFFFF;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          if (first == 0x7FFFFFFF || (!WIDE && (k & 1))) {
            const uint64_t b0 = __ballot(r[k].x == m), b1 = __ballot(r[k].y == m), b2 = __ballot(r[k].z == m),
                           b3 = __ballot(r[k].w == m);
            const uint64_t any = b0 | b1 | b2 | b3;
            if (any) {
              const int l = __builtin_ctzll(any);
              const int e = ((b0 >> l) & 1ull) ? 0 : ((b1 >> l) & 1ull) ? 1 : ((b2 >> l) & 1ull) ? 2 : 3;
              // (16-bit rows: groups 2j and 2j+1 hold labels 8l..8l+3 and 8l+4..8l+7 of lane l -- the smaller id may
              // sit in the odd group of an earlier lane, so both groups of a load are looked at before the answer stands)
              const int cand = WIDE ? (k * 64 + l) * 4 + e : ((k >> 1) * 64 + l) * 8 + (k & 1) * 4 + e;
              if (cand < first) first = cand;
            }
          }
        }
      }
      if constexpr (NP) np_accumulate(i);
    }
    if (lane == i) {
      my_m = m;
      my_rs = rs;
      my_s = s;
      my_first = first;
      my_cnt = cnt;
    }
    return cnt;  // (wave-uniform)
  };

  // Rows in flight behind the one being worked on. EVERY iteration issues the same loads whatever the block's length (past its
  // end the last row is requested again: an L2 hit nobody looks at): the compiler places `s_waitcnt vmcnt(n)` by counting the
  // loads outstanding on every path into a block and keeping the SMALLEST count, so a single path that skips a row's loads
  // (a short block, the loop's last iterations) turned every wait into "all of them" -- the prefetch existed in the source only.
  const int64_t row_last = row_lo + nrows - 1;
  auto row_at = [&](int i) { return row_lo + i < row_last ? row_lo + i : row_last; };
  if constexpr (NC <= 4) {
    // two rows in flight (three register buffers in rotation): at 13 waves per CU (LDS) and 4 KB a row that is ~100 KB per CU
    // on its way, what ~6 TB/s times the loaded memory latency asks for
    Raw ra[NL], rb[NL], rc[NL];
    load_row(row_at(0), ra);
    load_row(row_at(1), rb);
    for (int i = 0; i < nrows; i += 3) {
      load_row(row_at(i + 2), rc);
      phase_a(i, ra);
      load_row(row_at(i + 3), ra);
      if (i + 1 < nrows) phase_a(i + 1, rb);
      load_row(row_at(i + 4), rb);
      if (i + 2 < nrows) phase_a(i + 2, rc);
    }
  } else if constexpr (NC <= 8) {
    Raw ra[NL], rb[NL];
    load_row(row_at(0), ra);
    for (int i = 0; i < nrows; i += 2) {
      load_row(row_at(i + 1), rb);
      phase_a(i, ra);
      load_row(row_at(i + 2), ra);
      if (i + 1 < nrows) phase_a(i + 1, rb);
    }
  } else {
    // 2049 .. 4095 labels (round 5): a row is up to 64 registers per lane, there is no room for a second one in flight --
    // the other waves of the CU cover the wait (round 5 measured what the rows in flight are worth at V = 1024: 4 %)
    Raw ra[NL];
    for (int i = 0; i < nrows; ++i) {
      load_row(row_at(i), ra);
      phase_a(i, ra);
    }
  }
  __syncthreads();  // (one wave: orders phase A's LDS writes before phase B's reads)

  // ---- phase B: lane = row
  const int64_t row = row_lo + lane;
  const uint32_t ms = (uint32_t)a.max_surv;
  bool slow = false;
#ifdef CTC_PF_SKIP_B  // diagnostics only (timing phase A alone): one made-up survivor per row
  if (lane < nrows) {
    a.surv_id[(size_t)row * ms] = 0;
    a.surv_lp[(size_t)row * ms] = -1.0 + 1e-30 * (double)(my_cnt + (uint32_t)my_first) + 1e-30 * my_s;
    a.surv_cnt[row] = 1;
    a.row_sum[row] = (double)my_rs + (double)my_m * 1e-30;
  }
#else
  if (lane < nrows) {
    const uint32_t cnt = my_cnt;
    slow = cnt > (uint32_t)PF_CAND;
    if (!slow) {
      const double s = my_s;
      const double md = (double)my_m;
      double lse = 0.0;
      float l32 = 0.f;
      if constexpr (NP) {
        // this row's leaf sums, combined along numpy's recursion (every lane walks the same (dst, src) list on its own column)
        float s32 = my_s32;  // (up to eight leaves of 128: combined in registers by phase A)
        if (!a.np_uniform8) {
          for (int q = 0; q + 1 < a.np_n_leaf; ++q) {
            const int dst = a.np_prog[2 * q], src = a.np_prog[2 * q + 1];
            np_ls[dst * PF_ROWS + lane] = np_ls[dst * PF_ROWS + lane] + np_ls[src * PF_ROWS + lane];
          }
          s32 = np_ls[lane];
        }
        l32 = np_log_f32(s32);
      } else {
        lse = log_ge1(s);
      }
      auto logp_of = [&](float x) { return NP ? to_logp_np32(x, false, my_m, l32) : to_logp((double)x, false, md, lse); };
      const double best = logp_of(my_m);
      uint16_t* ids = ar_id + lane;
      float* xs = ar_x + lane;
      // the exact test (fp64, as in the per-row kernel) + insertion sort by id, in place
      uint32_t n = 0;
      uint32_t first = cnt > 0 ? 0xFFFFu : (uint32_t)my_first;  // (candidates: the smallest id whose logit is the maximum)
      for (uint32_t j = 0; j < cnt; ++j) {
        const uint16_t id = ids[j * PF_ROWS];
        const float x = xs[j * PF_ROWS];
        if (x == my_m && id < first) first = id;
        if (logp_of(x) >= a.token_min_logp) {
          uint32_t p = n;
          while (p > 0 && ids[(p - 1) * PF_ROWS] > id) {
            ids[p * PF_ROWS] = ids[(p - 1) * PF_ROWS];
            xs[p * PF_ROWS] = xs[(p - 1) * PF_ROWS];
            --p;
          }
          ids[p * PF_ROWS] = id;
          xs[p * PF_ROWS] = x;
          ++n;
        }
      }
      slow = n > SMALL_SET_MAX_KEYS || n >= ms;  // (n == ms with the argmax outside: the per-row kernel's overflow rules)
      if (!slow) {
        LaneTab tab{tabs + lane};
        const SmallSet r = small_set_order(tab, n, [ids](uint32_t k) { return (uint32_t)ids[k * PF_ROWS]; },
                                           first);
        uint16_t* out_id = a.surv_id + (size_t)row * ms;
        double* out_lp = a.surv_lp + (size_t)row * ms;
        uint32_t pos = 0;
        for (uint32_t k = 0; k <= r.mask; ++k) {
          const uint16_t v = tab.get(r.base + k);
          if (v != SMALL_SET_EMPTY) {
            const uint32_t pay = v >> SMALL_SET_ID_BITS;
            out_id[pos] = (uint16_t)(v & SMALL_SET_ID_MASK);
            out_lp[pos] = pay == SMALL_SET_ARGMAX ? best : logp_of(xs[pay * PF_ROWS]);
            ++pos;
          }
        }
        a.surv_cnt[row] = pos;
        a.row_sum[row] = (double)my_rs;
      }
    }
  }
#endif
  const uint64_t sm = __ballot(slow);
  if (sm) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&a.overflow[3], (uint32_t)__popcll(sm));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (slow) a.slow_rows[base + lanes_below(sm)] = (uint32_t)(row - a.row_base);
  }
}

// numpy's pairwise recursion over a row of V elements (np_sum.h): its leaves (offset, length) left to right and the additions
// that combine their sums, as (dst, src) leaf indices -- the result of a node lives where its leftmost leaf's did
struct NpPlan {
  int V = -1, n_leaf = 0, uniform8 = 0;
  uint16_t* d_leaf = nullptr;
  uint8_t* d_prog = nullptr;
};
static int np_plan_node(int off, int n, std::vector<uint16_t>& leaf, std::vector<uint8_t>& prog) {
  if (n <= 128) {
    leaf.push_back((uint16_t)off);
    leaf.push_back((uint16_t)n);
    return (int)leaf.size() / 2 - 1;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  const int l = np_plan_node(off, n2, leaf, prog), r = np_plan_node(off + n2, n - n2, leaf, prog);
  prog.push_back((uint8_t)l);
  prog.push_back((uint8_t)r);
  return l;
}
static int np_plan_for(int V, NpPlan** out, std::string* err) {
  static NpPlan plans[4];
  static int next = 0;
  for (NpPlan& p : plans)
    if (p.V == V) {
      *out = &p;
      return 0;
    }
  NpPlan& p = plans[next];
  next = (next + 1) % 4;
  std::vector<uint16_t> leaf;
  std::vector<uint8_t> prog;
  np_plan_node(0, V, leaf, prog);
  prog.push_back(0);  // (never empty)
  prog.push_back(0);
  HIP_TRY(hipStreamSynchronize(g_stream));  // (a launch in flight may still read the plan this one replaces)
  if (p.d_leaf) (void)hipFree(p.d_leaf);
  if (p.d_prog) (void)hipFree(p.d_prog);
  p.d_leaf = nullptr;
  p.d_prog = nullptr;
  p.V = -1;
  HIP_TRY(hipMalloc((void**)&p.d_leaf, leaf.size() * 2));
  HIP_TRY(hipMalloc((void**)&p.d_prog, prog.size()));
  HIP_TRY(hipMemcpy(p.d_leaf, leaf.data(), leaf.size() * 2, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p.d_prog, prog.data(), prog.size(), hipMemcpyHostToDevice));
  p.n_leaf = (int)leaf.size() / 2;
  p.uniform8 = (p.n_leaf == 1 || p.n_leaf == 2 || p.n_leaf == 4 || p.n_leaf == 8) ? 1 : 0;
  for (size_t k = 0; k < leaf.size(); k += 2)
    if (leaf[k + 1] != 128 || (leaf[k] & 127) != 0) p.uniform8 = 0;
  p.V = V;
  *out = &p;
  return 0;
}

int launch_prune(const PruneArgs& a_in, std::string* err) {
  PruneArgs a = a_in;
  {
    // float32 rows: the reference's own float32 arithmetic unless told otherwise (pk: round 5's packed polynomial; f64: round 2's)
    const char* ex0 = getenv("CTCDEC_PRUNE_EXP");
    a.f32_np = (a.dtype == 0 && !(ex0 && (ex0[0] == 'p' || ex0[0] == 'f'))) ? 1 : 0;
    a.np_prog = nullptr;
    a.np_leaf = nullptr;
    a.np_n_leaf = 0;
    a.np_uniform8 = 0;
  }
  if (a.pass == 0) {
    g_timing_valid = false;
    g_timing_override[0] = g_timing_override[1] = -1.0;
    HIP_TRY(hipEventRecord(g_ev[0], g_stream));
  }
  if (a.n_rows > 0) {
    uint32_t cap = set_table_cap((uint32_t)a.max_surv + 1);
    size_t lds = prune_lds_bytes((size_t)a.max_surv, cap) * PRUNE_WAVES;
    if (lds > 160 * 1024) {
      if (err) *err = "token_min_logp admits too many labels per frame for the LDS set tables";
      return -1;
    }
    dim3 grid((unsigned)((a.n_rows + PRUNE_WAVES - 1) / PRUNE_WAVES)), block(PRUNE_WAVES * 64);
    const bool vec4 = a.dtype == 0 && (a.n_labels % 4) == 0 && a.n_labels <= 1024 && a.rows_aligned16;
    // float32 rows of up to 2048 labels (8 groups of four per lane) take the 64-rows-per-wave kernel whatever their count
    // and alignment (16-byte loads when both allow them)
    static_assert(PF_MAX_LABELS - 1 <= (int)SMALL_SET_MAX_ID, "label ids must fit the small set tables");
    const bool f32_fast = a.dtype == 0 && a.n_labels <= PF_MAX_LABELS && a.rows_aligned4;
    const bool f32_al = (a.n_labels % 4) == 0 && a.rows_aligned16;
#define CTC_LAUNCH_PRUNE(KERN)                                                                                 \
  do {                                                                                                         \
    HIP_TRY(hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
    hipLaunchKernelGGL(KERN, grid, block, lds, g_stream, a, cap);                                              \
  } while (0)
    const char* ex = getenv("CTCDEC_PRUNE_EXP");  // "f64": the fp64 exponential of round 2 (diagnostics)
    const char* pk = getenv("CTCDEC_PRUNE_KERNEL");  // "row": one wave per row for every row (diagnostics)
    // (round 5: vocabularies no larger than the survivor bound -- character models, V ~ 30 -- take this kernel too: a row whose
    //  survivors reach the bound, there "every label survives", is handed to the per-row kernel like any other overflow. They
    //  used to run one row per wave at ~60 GB/s. Peaky posteriors -- what such models emit -- gain 35-40 % of the stage
    //  (V = 29: 1.02 -> 0.67 ms per 512 x 1000 rows); FLAT logits, where every row overflows, pay the screening on top of the
    //  per-row kernel (BASELINE configs[1], the stress input: 1.4 -> 2.5 of its 39 ms). An early exit per block was tried and
    //  changes neither. Round 6: the caller notices -- most rows of a call came back on the list -- and sets dense_hint for the
    //  next calls on that decoder: 3.6 -> 1.8 ms there.)
    const bool rows_ok = a.pass == 0 && a.slow_rows && a.n_rows < (1ll << 32) && !(ex && ex[0] == 'f') && !(pk && pk[0] == 'r') &&
                         !(a.dense_hint && a.dtype == 0 && a.f32_np);  // (the hint only where both kernels give the same bits)
    const bool rows64 = f32_fast && rows_ok;
    // 16-bit rows of a multiple of eight labels (16-byte loads of eight): the same kernel, widened on the fly
    const bool rows64h = (a.dtype == 2 || a.dtype == 3) && (a.n_labels % 8) == 0 && a.n_labels <= 1024 && a.rows_aligned16 && rows_ok;
    const dim3 fgrid((unsigned)((a.n_rows + PF_ROWS - 1) / PF_ROWS)), fblock(64);
    const unsigned rest = (unsigned)std::min<int64_t>((a.n_rows + PRUNE_WAVES - 1) / PRUNE_WAVES, 2048);
    if (rows64) {
      const int nc = ((a.n_labels + 3) / 4 + 63) / 64;  // groups of four labels per lane: 1 .. 16
      size_t flds = PF_LDS;
      if (a.f32_np) {  // numpy's summation plan for rows of this length
        NpPlan* plan = nullptr;
        if (np_plan_for(a.n_labels, &plan, err)) return -1;
        a.np_leaf = plan->d_leaf;
        a.np_prog = plan->d_prog;
        a.np_n_leaf = plan->n_leaf;
        a.np_uniform8 = plan->uniform8;
        flds += pf_np_extra_lds(nc <= 8 ? nc : (nc <= 12 ? 12 : 16), plan->uniform8 ? 0 : plan->n_leaf);
      }
      // rows the fast kernel hands over: the per-row float4 kernel where it applies (<= 1024 aligned labels), else the generic one
#define CTC_LAUNCH_FAST_K(KERN)                                                                                           \
  do {                                                                                                                    \
    HIP_TRY(hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));               \
    hipLaunchKernelGGL(KERN, fgrid, fblock, flds, g_stream, a);                                                           \
  } while (0)
#define CTC_LAUNCH_FAST(NCV, ALV)                                                                                         \
  do {                                                                                                                    \
    if (a.f32_np) CTC_LAUNCH_FAST_K((frame_prune_fast<NCV, 0, ALV, true>));                                               \
    else CTC_LAUNCH_FAST_K((frame_prune_fast<NCV, 0, ALV, false>));                                                       \
    if (ALV && NCV <= 4) {                                                                                                \
      HIP_TRY(hipFuncSetAttribute((const void*)frame_prune_f32x4_listed<(NCV <= 4 ? NCV : 4)>,                            \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                 \
      hipLaunchKernelGGL((frame_prune_f32x4_listed<(NCV <= 4 ? NCV : 4)>), dim3(rest), block, lds, g_stream, a, cap);     \
    } else {                                                                                                              \
      HIP_TRY(hipFuncSetAttribute((const void*)frame_prune_listed<float>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                  (int)lds));                                                                             \
      hipLaunchKernelGGL((frame_prune_listed<float>), dim3(rest), block, lds, g_stream, a, cap);                          \
    }                                                                                                                     \
  } while (0)
#define CTC_LAUNCH_FAST_NC(NCV)            \
  do {                                     \
    if (f32_al) CTC_LAUNCH_FAST(NCV, true); \
    else CTC_LAUNCH_FAST(NCV, false);      \
  } while (0)
      switch (nc) {
        case 1: CTC_LAUNCH_FAST_NC(1); break;
        case 2: CTC_LAUNCH_FAST_NC(2); break;
        case 3: CTC_LAUNCH_FAST_NC(3); break;
        case 4: CTC_LAUNCH_FAST_NC(4); break;
        case 5: CTC_LAUNCH_FAST_NC(5); break;
        case 6: CTC_LAUNCH_FAST_NC(6); break;
        case 7: CTC_LAUNCH_FAST_NC(7); break;
        case 8: CTC_LAUNCH_FAST_NC(8); break;
        case 9: case 10: case 11: case 12: CTC_LAUNCH_FAST_NC(12); break;  // (groups past the row's end are masked)
        default: CTC_LAUNCH_FAST_NC(16); break;
      }
#undef CTC_LAUNCH_FAST_NC
#undef CTC_LAUNCH_FAST
#undef CTC_LAUNCH_FAST_K
    } else if (rows64h) {
      const int nl = (a.n_labels / 8 + 63) / 64;  // 16-byte loads per lane: 1 (up to 512 labels) or 2
#define CTC_LAUNCH_FAST16(NCV, DTV, T)                                                                                    \
  do {                                                                                                                    \
    hipLaunchKernelGGL((frame_prune_fast<NCV, DTV>), fgrid, fblock, PF_LDS, g_stream, a);                                 \
    HIP_TRY(hipFuncSetAttribute((const void*)frame_prune_listed<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((frame_prune_listed<T>), dim3(rest), block, lds, g_stream, a, cap);                                 \
  } while (0)
      if (a.dtype == 2) {
        if (nl <= 1) CTC_LAUNCH_FAST16(2, 2, half_bits);
        else CTC_LAUNCH_FAST16(4, 2, half_bits);
      } else {
        if (nl <= 1) CTC_LAUNCH_FAST16(2, 3, bf16_bits);
        else CTC_LAUNCH_FAST16(4, 3, bf16_bits);
      }
#undef CTC_LAUNCH_FAST16
    } else if (vec4) {
      const int nc = (a.n_labels / 4 + 63) / 64;
      if (!(ex && ex[0] == 'f')) {
        if (nc <= 1) CTC_LAUNCH_PRUNE((frame_prune_f32x4<1, true>));
        else if (nc == 2) CTC_LAUNCH_PRUNE((frame_prune_f32x4<2, true>));
        else if (nc == 3) CTC_LAUNCH_PRUNE((frame_prune_f32x4<3, true>));
        else CTC_LAUNCH_PRUNE((frame_prune_f32x4<4, true>));
      } else {
        if (nc <= 1) CTC_LAUNCH_PRUNE((frame_prune_f32x4<1, false>));
        else if (nc == 2) CTC_LAUNCH_PRUNE((frame_prune_f32x4<2, false>));
        else if (nc == 3) CTC_LAUNCH_PRUNE((frame_prune_f32x4<3, false>));
        else CTC_LAUNCH_PRUNE((frame_prune_f32x4<4, false>));
      }
    } else if (a.dtype == 0) {
      CTC_LAUNCH_PRUNE(frame_prune<float>);
    } else if (a.dtype == 1) {
      CTC_LAUNCH_PRUNE(frame_prune<double>);
    } else if (a.dtype == 2) {
      CTC_LAUNCH_PRUNE(frame_prune<half_bits>);
    } else {
      CTC_LAUNCH_PRUNE(frame_prune<bf16_bits>);
    }
#undef CTC_LAUNCH_PRUNE
    HIP_TRY(hipGetLastError());
  }
  if (a.n_utts > 0) {
    if (a.pass == 0) hipLaunchKernelGGL(utt_sniff, dim3((unsigned)a.n_utts), dim3(64), 0, g_stream, a);
    else hipLaunchKernelGGL(utt_recount, dim3((unsigned)a.n_utts), dim3(64), 0, g_stream, a);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(g_ev[1], g_stream));
  return 0;
}

int launch_sniff_exact(const PruneArgs& a, std::string* err) {
  if (a.n_utts > 0) {
    hipLaunchKernelGGL(utt_sniff_exact, dim3((unsigned)a.n_utts), dim3(256), 0, g_stream, a);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

// workgroup kernel: one workgroup per utterance (beam_core.h), compiled in its own translation unit (beam_group_hip.hip).
// kind: 0 = four waves per utterance, 1 = eight, 2 = several language models (four waves)
int launch_group(const BeamArgs& a, const LdsShape& shape, size_t lds, int kind, hipStream_t stream, std::string* err);
// wave kernel: one wavefront per utterance (beam_wave.h), compiled in its own translation unit (beam_wave_hip.hip)
int launch_wave(const BeamArgs& a, hipStream_t stream, std::string* err);

// decode_batch (params.texts_only): the best beam's text of every utterance, assembled on the device. One wave per
// utterance walks the emission chain leaf to root through LDS windows and writes the UTF-8 bytes into the utterance's
// scratch area (text_wave.h), then copies them to a block of the text pool. A launch of its own, behind the beam kernel:
// all the chain walks run side by side instead of one at the tail of every beam wave.
struct TextGpuCtx {
  int lane;
  __device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
  __device__ __forceinline__ int clz64(uint64_t x) { return __clzll((long long)x); }
  __device__ __forceinline__ uint32_t wave_excl_sum_u32(uint32_t v) {
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
      if (lane >= off) incl += o;
    }
    return incl - v;
  }
  __device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
  }
};

__global__ __launch_bounds__(64) void assemble_texts(BeamArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[TEXT_LDS_BYTES];
  const int u = blockIdx.x;
  const int lane = threadIdx.x;
  if (a.n_out[u] == 0) return;
  OutBeam& ob = a.out[(size_t)u * a.out_stride];
  uint8_t* scratch = a.text_scratch + a.text_soff[u];
  const uint32_t cap = (uint32_t)(a.text_soff[u + 1] - a.text_soff[u]);
  TextLds L;
  text_lds_carve(L, (CTC_LDS char*)smem);
  TextGpuCtx ctx{lane};
  const uint32_t pos = wave_text_backwards(ctx, L, a.emit_nodes + a.emit_off[u], a.tables, ob.pad[1], scratch, cap,
                                           (uint32_t)(a.emit_off[u + 1] - a.emit_off[u]));
  uint32_t len = cap - pos;
  unsigned long long base = 0;
  if (lane == 0) {
    base = atomicAdd(a.tok_pool_head + 1, (unsigned long long)len);
    if (base + len > a.text_pool_cap) {
      a.status[u] |= ST_TOK_OVERFLOW;
      base = 0;
      len = 0;
    }
    ob.tok_off = (uint32_t)base;
    ob.tok_cnt = len;
    ob.pad[0] = (uint32_t)(base >> 32);
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (every lane reads bytes other lanes wrote)
  __builtin_amdgcn_wave_barrier();
  len = (uint32_t)__builtin_amdgcn_readfirstlane((int)len);
  const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
  const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
  base = ((unsigned long long)bhi << 32) | blo;
  for (uint32_t k = (uint32_t)lane; k < len; k += 64u) a.text_pool[base + k] = scratch[pos + k];
}

// ---- what a frame of each utterance will cost, before the wave kernel decodes it (round 6) -------------------------------------
// The wave kernel's launch lasts as long as its slowest wave. What makes a wave slow is its utterance's candidates per frame
// (live beams x surviving labels: r = 0.97 with a wave's natural lifetime, profiles/r06_weigh_predictors.txt). The live beams are
// known only once the utterance is decoded, but the surviving labels are what the prune stage has just counted, and they explain
// a fifth of it (r = 0.47) -- enough to let the launch know its probably-heavy utterances BEFORE it starts:
//   utt_weigh  (one wave per utterance): cost per frame = WEIGH_C0 + survivors per frame, and the utterance's total;
//   utt_place  (sixteen utterances per workgroup): ranks the totals by counting and deals the utterances out -- heaviest first: workgroup
//              b of a launch lands in slot b / 1024 of SIMD b mod 1024, so the first 1024 are the oldest waves of their SIMDs, and
//              age is the arbiter's tie-break -- in a snake over the SIMD count, so that each SIMD's four waves add up to about the
//              same predicted work; every workgroup is left the relative weight of its utterance (BeamArgs::block_weight), by which
//              WaveGpuCtx::frame_done scales the frames it still has to decode.
// Measured (profiles/r06_ab_weigh_place.log, r06_ab_weigh_gain_snake.log): the dispatch order is worth 1.0 ms of the 17-ms launch
// (lightest first: +1.3 ms), the snake 0.2 ms, the weights in the priority rule 0.1 ms.
// Which utterance a workgroup decodes never changes what it computes (tests/test_full_occupancy.py: placement test).
constexpr float WEIGH_C0 = 4.0f;
constexpr float WEIGH_GAIN = 4.0f;  // (the regression of a wave's natural lifetime on this weight has slope ~3: profiles/r06_weigh_*.txt)
constexpr int PLACE_MAX = 8192;  // utterances utt_place ranks (LDS: 4 bytes each); larger launches keep their order
struct WeighArgs {
  const int64_t* utt_row0;
  const uint32_t* surv_cnt;
  int32_t n_utts;
  float* total;        // [n_utts] WEIGH_C0 * frames + survivors
  float* per_frame;    // [n_utts]
  const int32_t* given_order;  // or nullptr: utt_place decides
  int32_t* order;      // [n_utts] out (given_order == nullptr)
  float* block_weight; // [n_utts] out
  int32_t simds;
  float gain;          // block_weight = 1 + gain * (relative cost per frame - 1)
  int32_t snake;       // odd generations dealt out in reverse
};
__global__ __launch_bounds__(64) void utt_weigh(WeighArgs a) {
  const int u = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t r0 = a.utt_row0[u], r1 = a.utt_row0[u + 1];
  double c = 0.0;
  {
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;  // (four loads in flight per trip)
    for (int64_t r = r0 + lane; r < r1; r += 256) {
      const uint32_t n0 = a.surv_cnt[r], n1 = r + 64 < r1 ? a.surv_cnt[r + 64] : 0u, n2 = r + 128 < r1 ? a.surv_cnt[r + 128] : 0u,
                     n3 = r + 192 < r1 ? a.surv_cnt[r + 192] : 0u;
      c0 += n0; c1 += n1; c2 += n2; c3 += n3;
    }
    c = (double)c0 + (double)c1 + (double)c2 + (double)c3;
  }
  c = wave_sum(c);
  if (lane == 0) {
    const float T = (float)(r1 - r0);
    const float tot = WEIGH_C0 * T + (float)c;
    const float pf = T > 0.f ? tot / T : WEIGH_C0;
    a.total[u] = tot;
    a.per_frame[u] = pf;  // (their sum is taken by utt_place's workgroups themselves: thousands of atomics on one address cost 40 us)
  }
}
// 16 utterances per workgroup of 256 threads: sixteen lanes share the count of one utterance's rank (a single workgroup ranking
// 4096 totals took 0.7 ms -- most of what the placement saved)
constexpr int PLACE_PER_BLOCK = 16;
__global__ __launch_bounds__(256) void utt_place(WeighArgs a) {
  __shared__ float w[PLACE_MAX];
  __shared__ float part_sum[4];
  const int n = a.n_utts;
  {  // the launch's mean cost per frame (every workgroup adds the per-utterance values up itself)
    float acc = 0.f;
    for (int v = threadIdx.x; v < n; v += blockDim.x) acc += a.per_frame[v];
    acc = (float)wave_sum((double)acc);
    if ((threadIdx.x & 63) == 0) part_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
  }
  const float scale = (float)n / ((part_sum[0] + part_sum[1]) + (part_sum[2] + part_sum[3]));
  const int u = blockIdx.x * PLACE_PER_BLOCK + (threadIdx.x >> 4), part = threadIdx.x & 15;
  if (a.given_order || n > PLACE_MAX || a.simds < 0) {  // weights only
    if (part == 0 && u < n) {
      const int v = a.given_order ? a.given_order[u] : u;
      a.block_weight[u] = 1.f + a.gain * (a.per_frame[v] * scale - 1.f);
    }
    return;
  }
  for (int v = threadIdx.x; v < n; v += blockDim.x) w[v] = a.total[v];
  __syncthreads();
  const int uu = u < n ? u : 0;
  const float mine = w[uu];
  int rank = 0;
  for (int v = part; v < n; v += 16) {
    const float x = w[v];
    rank += (x > mine || (x == mine && v < uu)) ? 1 : 0;
  }
  rank += __shfl_xor(rank, 1, 64);
  rank += __shfl_xor(rank, 2, 64);
  rank += __shfl_xor(rank, 4, 64);
  rank += __shfl_xor(rank, 8, 64);
  if (part == 0 && u < n) {
    const int S = a.simds;
    const int g = rank / S, k = rank - g * S;
    const int len = n - g * S < S ? n - g * S : S;
    int b = g * S + (((g & 1) && a.snake == 1) ? len - 1 - k : k);
    if (a.snake == 2) b = n - 1 - rank;  // (diagnostics: lightest first)
    a.order[b] = u;
    a.block_weight[b] = 1.f + a.gain * (a.per_frame[u] * scale - 1.f);
  }
}

static int g_last_kernel = 0;  // 1: wave kernel, 2: workgroup kernel
int last_beam_kernel() { return g_last_kernel; }

bool beam_kernel_depends_on_input(const BeamArgs& a) {
  return !getenv("CTCDEC_BEAM_KERNEL") && a.n_utts > 0 && a.n_utts <= g_cus && wave_eligible(a.tables, a.params) &&
         a.max_import <= wave_bucket(a.params.beam_width);
}

bool wave_kernel_chosen(const BeamArgs& a) {
  const char* force = getenv("CTCDEC_BEAM_KERNEL");
  // Small batches (at most one utterance per CU): by the input. Round 5 measured one utterance per call on real-posterior-like
  // input (371 x 29, ~1.3 survivors a frame, most frames consumed as single-label runs): 1.01 ms on one wave against 1.69 ms
  // on the workgroup kernel; on the bench input (~6 survivors a frame) 11.3 against 10.5 ms for one utterance; on flat logits
  // (29 survivors a frame, thousands of candidates) the workgroup kernel is four times faster. Hence: the wave kernel below 3
  // survivors a frame; unknown (surv_x16 == 0: the caller did not ask) the workgroup kernel. Between one and two utterances
  // per CU the density says too little -- 512 bench utterances 11.5 vs 12.1 ms in favour of the wave kernel, 512 utterances of
  // BASELINE configs[2] (32 character labels, a 4-gram: ~5 survivors but many live beams) 31.7 vs 17.9 ms against it: the
  // workgroup kernel stays.
  bool small_group = true;
  if (a.surv_x16 > 0 && a.n_utts <= g_cus) small_group = a.surv_x16 > 3 * 16;
  const bool want_group = force ? force[0] == 'g' : (a.n_utts <= 2 * g_cus && small_group);
  // (streaming: a stream may carry in more beams than this call's beam_width -- up to the workgroup kernel's table)
  return a.n_utts > 0 && !want_group && wave_eligible(a.tables, a.params) && a.max_import <= wave_bucket(a.params.beam_width);
}

int launch_beam(const BeamArgs& a, std::string* err) {
  // Two kernels for the same recursion. A lone utterance's frame takes about the same time in both (measured in round 4 on
  // the bench input: 11.3 us on the workgroup kernel's eight waves, 11.4 us on one wave), but a CU holds two workgroups
  // against sixteen waves and one wave needs an eighth of the issue slots per utterance: the wave kernel wins as soon as
  // there are more utterances than the workgroup kernel can hold at once (512 utterances: 12.6 ms either way;
  // 1024: 14.9 vs 24.4 ms; 4096: 19.3 ms). Dense frames are the exception: ~2 500 candidates a frame (BASELINE configs[1],
  // 256 utterances) take 38 ms on the workgroup kernel and 160 ms on one wave. CTCDEC_BEAM_KERNEL=wave|group overrides the
  // batch-size rule (tests, tuning; `wave` still falls back when the decode is not eligible for it).
  const bool wave = a.pay && wave_kernel_chosen(a);
  if (wave) {
    BeamArgs wa = a;
    // issue priority among the waves that share a SIMD (WaveGpuCtx::frame_done). CTCDEC_WAVE_PRIO=none | rot<k> | dyn
    // Default: by remaining work (measured on the 4096-utterance bench launch, round 6: 17.95 -> 17.1 ms; rot5 the same; the
    // waves of a launch end 12.3 .. 16.1 ms (5th .. 95th percentile) apart without it, 13.6 .. 15.2 ms with it; batches of one
    // wave per SIMD or fewer are unaffected).
    const char* pr = getenv("CTCDEC_WAVE_PRIO");
    // (round 6, second half) "weigh" = dyn with every remaining frame weighed by the utterance's survivors per frame, and the
    // heavy utterances dispatched first / dealt out evenly over the SIMDs (utt_weigh, utt_place): the default for launches of
    // more than one wave per SIMD
    wa.prio_mode = a.n_utts > 4 * g_cus ? 33 : 32;
    if (pr && pr[0] == 'n') wa.prio_mode = 0;
    if (pr && pr[0] == 'r') wa.prio_mode = 1 + (pr[1] && pr[2] && pr[3] ? atoi(pr + 3) & 15 : 5);
    if (pr && pr[0] == 'd') wa.prio_mode = 32;
    if (pr && pr[0] == 'w') wa.prio_mode = 33;
    wa.progress = nullptr;
    wa.total_frames = 0;
    wa.inv_n_utts = 0.f;
    wa.block_weight = nullptr;
    const bool dump_times = getenv("CTCDEC_WAVE_TIMES") != nullptr;
    if (wa.prio_mode >= 32 || dump_times) {
      static unsigned long long* g_progress = nullptr;
      if (!g_progress) HIP_TRY(hipMalloc((void**)&g_progress, 16));
      HIP_TRY(hipMemsetAsync(g_progress, 0, 16, g_stream));
      wa.progress = g_progress;
      wa.total_frames = (unsigned long long)a.total_rows;
      wa.inv_n_utts = 1.0f / (float)a.n_utts;
      if (a.total_rows >= (1ll << 32)) wa.prio_mode = 0;
      if (wa.prio_mode == 33 || dump_times) {  // (a dump carries the weights along)
        static float* g_weigh = nullptr;  // total | per_frame | block_weight | order
        static int g_weigh_cap = 0;
        if (a.n_utts > g_weigh_cap) {
          if (g_weigh) HIP_TRY(hipFree(g_weigh));
          g_weigh = nullptr;
          g_weigh_cap = 0;
          HIP_TRY(hipMalloc((void**)&g_weigh, (size_t)a.n_utts * 16));
          g_weigh_cap = a.n_utts;
        }
        WeighArgs w;
        w.utt_row0 = a.utt_row0;
        w.surv_cnt = a.surv_cnt;
        w.n_utts = a.n_utts;
        w.total = g_weigh;
        w.per_frame = g_weigh + (size_t)a.n_utts;
        w.block_weight = g_weigh + (size_t)a.n_utts * 2;
        w.order = (int32_t*)(g_weigh + (size_t)a.n_utts * 3);
        const bool keep_order = a.order != nullptr || getenv("CTCDEC_NO_PLACE") != nullptr || a.n_utts > PLACE_MAX || wa.prio_mode != 33;
        w.given_order = a.order;               // (a ragged multi-round launch keeps its longest-first order)
        w.simds = keep_order ? -1 : 4 * g_cus;  // -1: weights only
        w.gain = getenv("CTCDEC_WEIGH_GAIN") ? (float)atof(getenv("CTCDEC_WEIGH_GAIN")) : WEIGH_GAIN;
        w.snake = getenv("CTCDEC_PLACE_SNAKE") ? atoi(getenv("CTCDEC_PLACE_SNAKE")) : 1;
        hipLaunchKernelGGL(utt_weigh, dim3((unsigned)a.n_utts), dim3(64), 0, g_stream, w);
        hipLaunchKernelGGL(utt_place, dim3((unsigned)((a.n_utts + PLACE_PER_BLOCK - 1) / PLACE_PER_BLOCK)), dim3(256), 0, g_stream, w);
        HIP_TRY(hipGetLastError());
        wa.block_weight = w.block_weight;
        if (!keep_order) wa.order = w.order;
      }
    }
    // diagnostics: when and where each wave ran -> CTCDEC_WAVE_TIMES=<file> (n_utts x 4 uint64; synchronous)
    const char* wt = getenv("CTCDEC_WAVE_TIMES");
    wa.wave_clock = nullptr;
    if (wt && wt[0]) HIP_TRY(hipMalloc((void**)&wa.wave_clock, (size_t)a.n_utts * 32));
    const int rc = launch_wave(wa, g_stream, err);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    if (wa.wave_clock) {
      std::vector<unsigned long long> h((size_t)a.n_utts * 4);
      HIP_TRY(hipStreamSynchronize(g_stream));
      HIP_TRY(hipMemcpy(h.data(), wa.wave_clock, h.size() * 8, hipMemcpyDeviceToHost));
      HIP_TRY(hipFree(wa.wave_clock));
      if (FILE* f = fopen(wt, "wb")) {
        fwrite(h.data(), 8, h.size(), f);
        fclose(f);
      }
      if (wa.block_weight) {  // <file>.w: the relative weight of every workgroup's utterance (float32)
        std::vector<float> bw((size_t)a.n_utts);
        HIP_TRY(hipMemcpy(bw.data(), wa.block_weight, bw.size() * 4, hipMemcpyDeviceToHost));
        if (FILE* f = fopen((std::string(wt) + ".w").c_str(), "wb")) {
          fwrite(bw.data(), 4, bw.size(), f);
          fclose(f);
        }
      }
    }
    g_last_kernel = 1;
  } else if (a.n_utts > 0) {
    // Eight waves per utterance instead of four when every CU holds at most one utterance (the LDS of a workgroup allows
    // two per CU, the registers 2 x 256 threads or 1 x 512): measured on MI355X, 256 utterances x T=1000 -- BASELINE config 2
    // (~2 500 candidates per frame) 84.2 -> 70.4 ms in round 3, the headline workload (a few dozen candidates) 12.1 -> 11.8 ms.
    // That variant also takes its candidates in chunks of 1024 with a merge table of 4096 slots (it has the CU's LDS to
    // itself): config 2 in 38 ms (round 4). Otherwise four waves (512 utterances, beam 100: 256 threads 13.0 ms, 128 threads
    // 15.2 ms, 64 threads 19.7 ms for this kernel). CTCDEC_GROUP_THREADS=256|512 forces one.
    const char* gt = getenv("CTCDEC_GROUP_THREADS");
    const bool wide = a.tables.n_lms <= 1 && (gt ? gt[0] == '5' : a.n_utts <= g_cus);
    LdsShape shape = make_shape(a.params.beam_width, a.params.max_surv, group_cand(beam_bucket(a.params.beam_width), wide));
    size_t lds = lds_bytes(shape);
    if (lds > 160 * 1024) {
      if (err) *err = "beam table does not fit LDS (" + std::to_string(lds) + " bytes)";
      return -1;
    }
    int rc;
    rc = launch_group(a, shape, lds, a.tables.n_lms > 1 ? 2 : (wide ? 1 : 0), g_stream, err);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    g_last_kernel = 2;
  }
  if (a.n_utts > 0 && a.params.texts_only && a.text_scratch) {
    hipLaunchKernelGGL(assemble_texts, dim3((unsigned)a.n_utts), dim3(64), 0, g_stream, a);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(g_ev[2], g_stream));
  g_timing_valid = true;
  return 0;
}

}  // namespace be
}  // namespace ctc
