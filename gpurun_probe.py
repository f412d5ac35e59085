import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from stripenn_amd import synth, hip
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
t = time.time(); ch = synth.SynthChrom(nb, 16); band_h = ch.band(512); print('gen', time.time() - t, band_h.shape)
ctx = hip.Context(0); ctx.set_profiling(True)
t = time.time(); band = ctx.band_upload(band_h); print('upload', time.time() - t)
nfr = -(-nb // 200)
st = np.array([max(0, i * 200 - 100) for i in range(nfr)]); en = np.minimum((np.arange(nfr) + 1) * 200 + 99, nb - 1)
t = time.time(); fr = band.frames(st, en); print('frames', time.time() - t, fr.S[:5])
pos = band_h[band_h > 0]; M = np.quantile(pos, [0.95, 0.96, 0.97, 0.98, 0.99]); print('M', M)
for rep in range(2):
    ctx.reset_stats()
    t = time.time(); recs = fr.stripe_search(M); dt = time.time() - t
    px = float((fr.S.astype(np.float64) ** 2).sum()) * 5
    print('search %.3fs recs %d contact Mpx/s %.1f' % (dt, len(recs), px / dt / 1e6))
    for k, v in ctx.stats().items():
        print('  %-14s launches %3d  ms %9.3f  alg GB/s %8.1f' % (k, v['launches'], v['ms'], v['alg_bytes'] / v['ms'] / 1e6 if v['ms'] else 0))
