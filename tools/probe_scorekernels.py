"""The score kernels alone (nothing else on the device): k_pvalue and k_stripiness on the candidate stripes of three
chromosomes of the benchmark genome, HIP-event time per launch and per stripe.
    python tools/probe_scorekernels.py [repeats]        (STP_LIB + --allow semantics as bench.py: timing builds welcome)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from stripenn_amd import backend as BK

rep = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda', 0)
torch.cuda.init()
hb = BK.HipBackend(0)
names = ['chr1', 'chr10', 'chr19']
spec = dict(names=names, nbins=[-(-s // bench.RESOL) for s in (195471971, 130694993, 61431566)], seeds=[1, 10, 19], wl='probe')
W = bench._Workload(hb, dev, spec, 1, 0, '', score=True, sigma=2.0)
jobs = []
for unit in W.my_units:
    (ci, f0, f1), fr, pend = W._launch(unit)
    recs = pend.wait()
    pv, sc = BK.score_inputs(recs, fr.nz, W.tabs[ci][0][f0:f1], W.nbins[ci], W.bs)
    jobs.append((ci, pv, sc))
    fr.close()
nst = sum(len(j[1]) for j in jobs)
W.ctx.synchronize()
W.reset_stats(True)
t0 = time.perf_counter()
for _ in range(rep):
    for ci, pv, sc in jobs:
        hb.pvalue(W.bands[names[ci]], W.bs, pv)
for _ in range(rep):
    for ci, pv, sc in jobs:
        hb.stripiness(W.bands[names[ci]], W.EV[ci], sc)
for _ in range(rep):
    for ci, pv, sc in jobs:
        hb.score(W.bands[names[ci]], W.bs, W.EV[ci], pv, sc)
h = np.concatenate([j[1]['row1'] - j[1]['row0'] for j in jobs])
print('stripes: h median %d, p90 %d, max %d; %.1f %% above 192 rows (block kernels)' % (np.median(h), np.percentile(h, 90), h.max(), 100.0 * (h > 192).mean()))
st = W.stats()
for k in ('pvalue', 'stripiness', 'score', 'pvalue_block', 'stripiness_block'):
    if k not in st:
        continue
    v = st[k]
    print('%-10s %6.3f ms per launch, %5.2f us per 100 stripes (%d launches, %d stripes per pass of %d units)'
          % (k, v['ms'] / v['launches'], v['ms'] * 1e3 / (rep * nst) * 100, v['launches'], nst, len(jobs)))
W.release(); hb.close()
