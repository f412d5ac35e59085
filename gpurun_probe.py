import sys, time, numpy as np
sys.path.insert(0, '.')
from stripenn_amd import synth, hip
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 19642
ch = synth.SynthChrom(nb, 16); band_h = ch.band(512)
ctx = hip.Context(0); band = ctx.band_upload(band_h)
nfr = -(-nb // 200)
st = np.array([max(0, i * 200 - 100) for i in range(nfr)]); en = np.minimum((np.arange(nfr) + 1) * 200 + 99, nb - 1)
M = np.quantile(band_h[band_h > 0], [0.95, 0.96, 0.97, 0.98, 0.99])
for rep in range(4):
    t0 = time.perf_counter(); fr = band.frames(st, en); t1 = time.perf_counter()
    recs = fr.stripe_search(M); t2 = time.perf_counter(); fr.close(); t3 = time.perf_counter()
    print('frames %.2f ms  search %.2f ms  close %.2f ms  recs %d' % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, len(recs)))
ctx.set_profiling(True); ctx.reset_stats()
fr = band.frames(st, en); recs = fr.stripe_search(M)
for k, v in ctx.stats().items(): print('  %-14s %8.3f ms' % (k, v['ms']))
