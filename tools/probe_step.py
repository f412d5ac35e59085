"""Wall-clock split of one full bench step (frames, search, score marshalling, pvalue, stripiness)."""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, '..')
import bench
from stripenn_amd import synth, backend as BK, getStripe as GS
nb = 19642
ch = synth.SynthChrom(nb, 16); band_h = ch.band(512)
hb = BK.HipBackend(0)
st, en = bench.frame_table(nb)
Ms = np.quantile(band_h[band_h > 0], bench.MAXPIXEL)
sel = synth.SynthSelector({'chr16': ch}, 5000)
obj = GS.getStripe(sel, 5000, 10, 8, 2.0, ['chr16'], ['chr16'], np.array([nb * 5000]), np.array([nb * 5000]), 2, 3,
                   123456789, backend=hb)
obj._bands['chr16'] = hb.ctx.band_upload(band_h)
band = obj._bands['chr16']
EV = np.asarray(obj.mpmean()['chr16']); hb.set_background(*obj.nulldist())
bs = 10


def score_inputs(recs, fr):
    f = recs['frame']; base = st[f].astype(np.int64); nzf = fr.nz
    x0 = base + nzf[f, recs['x']]; x1 = base + nzf[f, recs['x'] + recs['w'] - 1]
    y0 = base + nzf[f, recs['y']]; y1 = base + nzf[f, recs['y'] + recs['h'] - 1]
    n = len(recs)
    pv = np.zeros(n, dtype=BK.PV_STRIPE_DTYPE)
    pv['row0'], pv['row1'] = y0, y1 + 1
    pv['col0'], pv['col1'] = np.maximum(x0 - bs, 0), np.minimum(x1 + 1 + bs, nb)
    pv['mode'] = np.where(x0 == y0, 0, 1); pv['upbase'] = y1 + 1 - y0
    sc = np.zeros(n, dtype=BK.SCORE_STRIPE_DTYPE)
    sc['row0'], sc['row1'] = y0, y1 + 1
    lm = np.minimum(np.maximum(x0 - bs, 1), x0); rm = np.minimum(x1 + 1 + bs, nb - 1)
    sc['col0'][:, 0], sc['col1'][:, 0] = x0, x1 + 1
    sc['col0'][:, 1], sc['col1'][:, 1] = lm, x0
    sc['col0'][:, 2], sc['col1'][:, 2] = x1 + 1, np.maximum(rm, x1 + 1)
    sc['ex0'][:, 0], sc['ex0'][:, 1], sc['ex0'][:, 2] = x0, lm, x1 + 2
    sc['ey0'] = y0; sc['mirror'] = np.where(x0 == y0, 0, 1)
    sc['mcol0'], sc['mcol1'], sc['mrow0'], sc['mrow1'] = 1, 0, 1, 0
    return pv, sc


names = ('frames', 'search', 'marshal', 'pvalue', 'stripiness', 'close')
for rep in range(5):
    t = [time.perf_counter()]
    fr = band.frames(st, en); t.append(time.perf_counter())
    recs = fr.stripe_search(Ms); t.append(time.perf_counter())
    pv, sc = score_inputs(recs, fr); t.append(time.perf_counter())
    p = hb.pvalue(band, bs, pv); t.append(time.perf_counter())
    g = hb.stripiness(band, EV, sc)[0]; t.append(time.perf_counter())
    fr.close(); t.append(time.perf_counter())
    print('  '.join('%s %.3f' % (n, (b - a) * 1e3) for n, a, b in zip(names, t, t[1:])), ' total %.3f ms  recs %d' % ((t[-1] - t[0]) * 1e3, len(recs)))
hb.ctx.set_profiling(True); hb.ctx.reset_stats()
fr = band.frames(st, en); recs = fr.stripe_search(Ms); pv, sc = score_inputs(recs, fr); hb.pvalue(band, bs, pv); hb.stripiness(band, EV, sc)
print({k: round(v['ms'], 3) for k, v in hb.ctx.stats().items()})
