import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, '..')
from stripenn_amd import synth, hip
nb = 19642
ch = synth.SynthChrom(nb, 16); band_h = ch.band(512)
ctx = hip.Context(0); band = ctx.band_upload(band_h)
nfr = -(-nb // 200)
st = np.array([max(0, i * 200 - 100) for i in range(nfr)]); en = np.minimum((np.arange(nfr) + 1) * 200 + 99, nb - 1)
M = np.quantile(band_h[band_h > 0], [0.95, 0.96, 0.97, 0.98, 0.99])
fr = band.frames(st, en); recs = fr.stripe_search(M)
ctx.set_profiling(True)
for rep in range(3):
    ctx.reset_stats(); recs = fr.stripe_search(M)
print(' '.join('%s=%.3f' % (k, v['ms']) for k, v in ctx.stats().items()))
