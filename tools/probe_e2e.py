import sys, time, os, io, contextlib
sys.path.insert(0, '.'); sys.path.insert(0, '..')
import numpy as np
from stripenn_amd import stripenn, getStripe
spec = 'synth:' + ','.join('chr%d=%d' % (i + 1, (6000 - 400 * i) * 5000 - 1234) for i in range(6)) + ';resol=5000;seed=7'
t0 = time.time()
# time the phases by wrapping the facade methods
orig = {}
acc = {}
for name in ('_band', 'getQuantile_original', 'mpmean', 'nulldist', 'extract', 'RemoveRedundant', 'scoringstripes', 'pvalue', '_search'):
    f = getattr(getStripe.getStripe, name)
    def mk(f, name):
        def w(self, *a, **k):
            t = time.time(); r = f(self, *a, **k); acc[name] = acc.get(name, 0.0) + time.time() - t; return r
        return w
    setattr(getStripe.getStripe, name, mk(f, name))
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    stripenn.compute(spec, 'gpurun_out/e2e_out', 'KR', 'all', 2.0, 10, 8, '0.95,0.96,0.97,0.98,0.99', 8, 0.1, '0', False, 3, 123456789, force=True)
print('total %.1f s' % (time.time() - t0))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print('  %-22s %.2f s (inclusive)' % (k, v))
print(open('gpurun_out/e2e_out/result_filtered.tsv').read().count('\n') - 1, 'filtered stripes;', open('gpurun_out/e2e_out/result_unfiltered.tsv').read().count('\n') - 1, 'unfiltered')
