"""Host-side timing of one bench step, call by call (profiling aid)."""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, '..')
import bench
from stripenn_amd import synth, backend as BK
nb = 19642
ch = synth.SynthChrom(nb, 16); band_h = ch.band(512)
hb = BK.HipBackend(0); band = hb.ctx.band_upload(band_h)
st, en = bench.frame_table(nb)
Ms = np.quantile(band_h[band_h > 0], bench.MAXPIXEL)
for rep in range(4):
    t = [time.perf_counter()]
    fr = band.frames(st, en); t.append(time.perf_counter())
    recs = fr.stripe_search(Ms); t.append(time.perf_counter())
    fr.close(); t.append(time.perf_counter())
    print('frames %.3f  search %.3f  close %.3f ms' % tuple((b - a) * 1e3 for a, b in zip(t, t[1:])))
hb.ctx.set_profiling(True); hb.ctx.reset_stats()
fr = band.frames(st, en); recs = fr.stripe_search(Ms)
print({k: round(v['ms'], 3) for k, v in hb.ctx.stats().items()})
