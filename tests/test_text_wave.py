"""csrc/text_wave.h (the wavefront's walk of an emission chain that assembles decode_batch's texts on the device) against
csrc/beam_core.h: text_backwards (one thread, the definition) on random chains: long gaps between parent and child, labels
without bytes, boundaries in every position, scratch areas too small for the text, window and list sizes that cut the chain
everywhere. The wave runs on the 64 fibers of tests/sim/wave_fibers.h. Test infrastructure (tests/_build/)."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pyctcdecode_amd", "csrc")
OUT_DIR = os.path.join(ROOT, "tests", "_build")
HELPER = r'''
#define CTC_SIM
#include <vector>
#include <random>
#include "%(csrc)s/beam_core.h"
#include "%(csrc)s/text_wave.h"
#include "%(root)s/tests/sim/wave_fibers.h"
using namespace ctc;
// one random chain per call; returns 0 when both walks agree, a positive code otherwise
extern "C" int text_walk_case_(uint32_t seed, uint32_t n_nodes, uint32_t max_gap, uint32_t cap, uint32_t* out_len) {
  std::mt19937 rng(seed);
  const uint32_t V = 40;
  std::vector<TokText> tt(V);
  std::vector<uint8_t> bytes;
  for (uint32_t t = 0; t < V; ++t) {
    const uint32_t rl = rng() %% 7, cl = rl ? rng() %% (rl + 1) : 0;   // clean form: a suffix-length of the raw one, maybe empty
    tt[t].raw_off = (uint32_t)bytes.size();
    tt[t].raw_len = (uint16_t)rl;
    for (uint32_t k = 0; k < rl; ++k) bytes.push_back((uint8_t)('a' + rng() %% 26));
    tt[t].clean_off = tt[t].raw_off + (rl - cl);
    tt[t].clean_len = (uint16_t)cl;
    tt[t].pad = 0;
  }
  if (bytes.empty()) bytes.push_back('x');
  std::vector<EmitNode> nodes(n_nodes);
  const uint32_t kinds[4] = {BR_APPEND, BR_BOUNDARY, BR_SPACE, BR_FINAL};
  for (uint32_t e = 1; e < n_nodes; ++e) {
    const uint32_t gap = 1 + rng() %% max_gap;
    nodes[e].parent = e > gap ? e - gap : 0;
    const uint32_t kind = (rng() %% 8 < 5) ? BR_APPEND : kinds[rng() %% 4];
    nodes[e].tok_branch = (rng() %% V) | (kind << 16);
    nodes[e].wstart = nodes[e].wend = 0;
  }
  nodes[0].parent = 0;
  nodes[0].tok_branch = 0;
  DeviceTables tab;
  memset(&tab, 0, sizeof(tab));
  tab.tok_text = tt.data();
  tab.tok_bytes = bytes.data();
  DecodeParams prm;
  memset(&prm, 0, sizeof(prm));
  const uint32_t leaf = n_nodes - 1;
  std::vector<uint8_t> a(cap + 1, 0xEE), b(cap + 1, 0xEE);
  const uint32_t pos = text_backwards(nodes.data(), tab, leaf, a.data(), cap, n_nodes);
  std::vector<char> lds(TEXT_LDS_BYTES + 16, (char)0xCD);
  TextLds tl;
  text_lds_carve(tl, (char*)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15));
  uint32_t wpos[wavesim::LANES];
  wavesim::Wave wave;
  wave.run([&](int lane) {
    wavesim::SimWaveCtx ctx{lane, &wave, &tab, &prm};
    wpos[lane] = wave_text_backwards(ctx, tl, nodes.data(), tab, leaf, b.data(), cap, n_nodes);
  });
  for (int l = 0; l < wavesim::LANES; ++l) if (wpos[l] != pos) return 1;
  if (memcmp(a.data() + pos, b.data() + pos, cap - pos) != 0) return 2;
  if (b[cap] != 0xEE) return 3;                                   // nothing past the area
  for (uint32_t k = 0; k < pos; ++k) if (b[k] != 0xEE) return 4;  // nothing before the text
  *out_len = cap - pos;
  return 0;
}
'''


def _lib(win, lst):
    os.makedirs(OUT_DIR, exist_ok=True)
    tag = "%d_%d" % (win, lst)
    cpp, so = os.path.join(OUT_DIR, "text_walk_%s.cpp" % tag), os.path.join(OUT_DIR, "text_walk_%s.so" % tag)
    with open(cpp, "w") as f:
        f.write(HELPER % {"csrc": CSRC, "root": ROOT})
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-DCTC_TEXT_WIN=%d" % win, "-DCTC_TEXT_LIST=%d" % lst,
                           "-o", so, cpp])
    dll = C.CDLL(so)
    dll.text_walk_case_.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    return dll


@pytest.mark.parametrize("win,lst", [(512, 512), (48, 16), (64, 64), (7, 3), (130, 200)])
def test_wave_walk_equals_the_one_thread_walk(win, lst):
    dll = _lib(win, lst)
    lens = []
    for seed in range(120):
        n_nodes = [2, 3, 40, 700, 3000][seed % 5]
        max_gap = [1, 2, 9, 60, 900][(seed // 5) % 5]
        for cap in (0, 1, 5, 64, 20000):
            n = C.c_uint32(0)
            rc = dll.text_walk_case_(seed, n_nodes, max_gap, cap, C.byref(n))
            assert rc == 0, (win, lst, seed, n_nodes, max_gap, cap, rc)
            lens.append(n.value)
    assert max(lens) > 1000 and min(lens) == 0  # long texts and empty ones were both seen
