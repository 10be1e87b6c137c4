import sys, os, time, numpy as np, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench, synth
from pyctcdecode_amd import build_ctcdecoder
lm, labels, hot = bench.build_assets(os.path.join(sys.path[0], 'bench_cache'), 20000, 60000)
xs = np.stack(bench.make_batch(lm, labels, 0, 512, 1000))
dec = build_ctcdecoder(labels, lm.path)
dev = torch.from_numpy(xs).cuda()
for _ in range(2): dec.decode_batch(None, dev, beam_width=100, hotwords=hot)
torch.cuda.synchronize()
ts=[]
for _ in range(5):
    t=time.perf_counter(); out = dec.decode_batch(None, dev, beam_width=100, hotwords=hot); ts.append((time.perf_counter()-t)*1e3 - dec.last_timing_ms[2])
print('python-side ms per call (total - native):', ['%.3f'%v for v in ts])
pr = cProfile.Profile(); pr.enable()
for _ in range(5): dec.decode_batch(None, dev, beam_width=100, hotwords=hot)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
