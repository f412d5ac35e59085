"""How many hysteresis sweeps do the images of the benchmark need?  Timing-only library (make ablate: libstp_ablate_stops.so) with
STP_LINES_STOP=99: k_lines ends after the closure and reports its sweep count as the image's record count.
    STP_LIB=$PWD/stripenn_amd/libstp_ablate_stops.so STP_LINES_STOP=99 python tools/probe_hyst_sweeps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from stripenn_amd import backend as BK

torch.cuda.init()
hb = BK.HipBackend(0)
spec = dict(names=['chr16'], nbins=[bench.CHR16_BINS], seeds=[16], wl='probe')
W = bench._Workload(hb, torch.device('cuda', 0), spec, 1, 0, '', score=False, sigma=2.0)
(ci, f0, f1), fr, pend = W._launch(W.my_units[0])
recs = pend.wait()
key = (recs['frame'].astype(np.int64) * 5 + recs['level']) * 6 + recs['b_index']
cnt = np.bincount(key, minlength=len(fr.S) * 30)
cnt = cnt[cnt > 0]
print('%d images with edges: sweeps mean %.1f, median %d, p90 %d, p99 %d, max %d' % (len(cnt), cnt.mean(), np.median(cnt), np.percentile(cnt, 90), np.percentile(cnt, 99), cnt.max()))
print('histogram (sweeps: images):', {int(k): int(v) for k, v in zip(*np.unique(cnt, return_counts=True))})
