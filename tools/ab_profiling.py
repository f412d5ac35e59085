"""What the library's own kernel timers (an event pair around every chain kernel: bench.py's roofline needs them) cost the genome step:
the same workload with profiling on and off, same process.   python tools/ab_profiling.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from stripenn_amd import backend as BK

dev = torch.device('cuda', 0)
hb = BK.HipBackend(0)
spec = dict(names=bench.MM10_NAMES, nbins=[-(-s // bench.RESOL) for s in bench.MM10], seeds=list(range(1, 21)), wl='genome')
W = bench._Workload(hb, dev, spec, 1, 0, '', score=True, sigma=2.0)
for _ in range(2):
    W.step()
for rep in range(2):
    for prof in (True, False):
        W.reset_stats(profiling=prof)
        torch.cuda.synchronize(); W.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            W.step()
        torch.cuda.synchronize(); W.synchronize()
        print('profiling %-5s: %.2f ms per step' % (prof, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
