# frame preparation alone (k_frame_prep on an otherwise idle device): ms per call for the chr16-size chromosome's 99 frames
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, '..')
from stripenn_amd import synth, hip
nb = 19642
ch = synth.SynthChrom(nb, 16); band_h = ch.band(512)
ctx = hip.Context(0); band = ctx.band_upload(band_h)
nfr = -(-nb // 200)
st = np.array([max(0, i * 200 - 100) for i in range(nfr)]); en = np.minimum((np.arange(nfr) + 1) * 200 + 99, nb - 1)
fr = band.frames(st, en); fr.close()
ctx.set_profiling(True)
for rep in range(3):
    ctx.reset_stats()
    t0 = time.perf_counter()
    for k in range(10):
        fr = band.frames(st, en); fr.close()
    dt = (time.perf_counter() - t0) / 10
s = ctx.stats()
print('frames() %.3f ms wall; ' % (dt * 1e3) + ' '.join('%s=%.3f' % (k, v['ms'] / 10) for k, v in s.items()))
