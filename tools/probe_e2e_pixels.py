"""Wall-clock split of the full `compute` driver fed from a pixel table (profiling aid)."""
import sys, time, os, io, contextlib
sys.path.insert(0, '.'); sys.path.insert(0, '..')
import numpy as np
from stripenn_amd import stripenn, getStripe, pixels, synth
names = ['chr%d' % (i + 1) for i in range(6)]
chroms = {n: synth.SynthChrom(6000 - 400 * i, 7 + i) for i, n in enumerate(names)}
t0 = time.time()
t = pixels.PixelTable.from_synth(names, chroms, 5000)
os.makedirs('gpurun_out', exist_ok=True)
t.save('gpurun_out/e2e_pixels.npz')
print('table: %d pixels, built in %.1f s' % (len(t.count), time.time() - t0))
acc = {}
for name in ('_band', 'getQuantile_original', 'mpmean', 'nulldist', 'extract', 'RemoveRedundant', 'scoringstripes', 'pvalue', '_search'):
    f = getattr(getStripe.getStripe, name)
    def mk(f, name):
        def w(self, *a, **k):
            t = time.time(); r = f(self, *a, **k); acc[name] = acc.get(name, 0.0) + time.time() - t; return r
        return w
    setattr(getStripe.getStripe, name, mk(f, name))
t0 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    stripenn.compute('pixels:gpurun_out/e2e_pixels.npz', 'gpurun_out/e2e_px_out', 'weight', 'all', 2.0, 10, 8,
                     '0.95,0.96,0.97,0.98,0.99', 8, 0.1, '0', False, 3, 123456789, force=True)
print('compute total %.2f s' % (time.time() - t0))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print('  %-22s %.2f s (inclusive)' % (k, v))
print(open('gpurun_out/e2e_px_out/result_filtered.tsv').read().count('\n') - 1, 'filtered stripes;', open('gpurun_out/e2e_px_out/result_unfiltered.tsv').read().count('\n') - 1, 'unfiltered')
