"""Does the chain gain from two searches running CONCURRENTLY on one GPU (two contexts = two streams and workspaces, two host
threads), now that the Canny kernel waits for loads and barriers most of its time?  chr16-size sweep, 6 searches in all."""
import sys, time, threading, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, '..')
from stripenn_amd import synth, hip
nb = 19642
ch = synth.SynthChrom(nb, 16); band_h = ch.band(512)
nfr = -(-nb // 200)
st = np.array([max(0, i * 200 - 100) for i in range(nfr)]); en = np.minimum((np.arange(nfr) + 1) * 200 + 99, nb - 1)
M = np.quantile(band_h[band_h > 0], [0.95, 0.96, 0.97, 0.98, 0.99])
ctxs = [hip.Context(0), hip.Context(0)]
frs = []
for c in ctxs:
    band = c.band_upload(band_h); fr = band.frames(st, en); fr.stripe_search(M); frs.append((band, fr))
N = 6
def run(k, n):
    for _ in range(n):
        frs[k][1].stripe_search(M)
for rep in range(2):
    t0 = time.perf_counter(); run(0, N); t1 = time.perf_counter() - t0
    th = [threading.Thread(target=run, args=(k, N // 2)) for k in (0, 1)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; t2 = time.perf_counter() - t0
    print('one context, %d searches: %.1f ms each; two contexts concurrently: %.1f ms each' % (N, t1 / N * 1e3, t2 / N * 1e3), flush=True)
