"""Whole-genome `compute` on one GPU: mm10 chromosome sizes at 5 kb (2 645 frames x 5 levels),
synthetic pixel table held in memory, the unmodified driver stripenn_amd.stripenn.compute.
    python tools/probe_genome.py [scale]      # scale < 1 shrinks every chromosome (quick check)"""
import contextlib, io as _io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stripenn_amd import getStripe, io, pixels, stripenn, synth, synth_device

MM10 = [195471971, 182113224, 160039680, 156508116, 151834684, 149736546, 145441459, 129401213, 124595110, 130694993,
        122082543, 120129022, 120421639, 124902244, 104043685, 98207768, 94987271, 90702639, 61431566, 171031299]
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
names = ['chr%d' % (i + 1) for i in range(19)] + ['chrX']
t0 = time.time()
torch.cuda.init()
chroms = {n: synth_device.DeviceChrom(int(-(-s * scale // 5000)), 1 + i, torch.device('cuda', 0)) for i, (n, s) in enumerate(zip(names, MM10))}
table = synth_device.pixel_table(names, chroms, 5000)          # synth.py's pixel function evaluated on the device
print('genome: %d bins, %d stored pixels, table built in %.0f s' % (table.chrom_offset[-1], len(table.count), time.time() - t0), flush=True)
stripenn.open_matrix = lambda cool: io.pixel_matrix(table)
acc = {}
for name in ('_band', 'getQuantile_original', 'mpmean', 'nulldist', 'extract', 'RemoveRedundant', 'scoringstripes', 'pvalue', '_search'):
    f = getattr(getStripe.getStripe, name)
    def mk(f, name):
        def w(self, *a, **k):
            t = time.time(); r = f(self, *a, **k); acc[name] = acc.get(name, 0.0) + time.time() - t; return r
        return w
    setattr(getStripe.getStripe, name, mk(f, name))
os.makedirs('gpurun_out', exist_ok=True)
def _run():
    with contextlib.redirect_stdout(_io.StringIO()):
        stripenn.compute('pixels:in-memory', 'gpurun_out/genome_out', 'weight', 'all', 2.0, 10, 8, '0.95,0.96,0.97,0.98,0.99', 8,
                         0.1, '0', False, 3, 123456789, force=True)
if os.environ.get('STP_PROBE_PROFILE') == 'first':      # host-side hot spots of the FIRST run of the process
    import cProfile, pstats
    pr = cProfile.Profile(); t0 = time.time(); pr.enable(); _run(); pr.disable(); total = time.time() - t0
    pstats.Stats(pr).sort_stats('tottime').print_stats(25)
elif os.environ.get('STP_PROBE_PROFILE') == '1':          # host-side hot spots of a second run (library / workspaces warm)
    import cProfile, pstats
    _run(); acc.clear()
    pr = cProfile.Profile(); t0 = time.time(); pr.enable(); _run(); pr.disable(); total = time.time() - t0
    pstats.Stats(pr).sort_stats('tottime').print_stats(30)
else:
    t0 = time.time()
    _run()
    print('first compute of the process (library load, workspaces, pinned buffers): %.2f s' % (time.time() - t0))
    acc.clear()
    t0 = time.time()
    _run()
    total = time.time() - t0
nfr = sum(-(-c.nbins // 200) for c in chroms.values())
print('compute: %.2f s for %d frames x 5 levels (%d frame-levels; the reference needs ~0.78 s of one core for each)' % (total, nfr, nfr * 5))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print('  %-22s %.2f s (inclusive)' % (k, v))
print(open('gpurun_out/genome_out/result_filtered.tsv').read().count('\n') - 1, 'filtered stripes;',
      open('gpurun_out/genome_out/result_unfiltered.tsv').read().count('\n') - 1, 'unfiltered')

# configs[3]'s second half: `stripenn score` re-scoring of the called stripes, on the same genome.  Same seed and
# numcores -> same background tables -> the added columns must reproduce the ones compute wrote.
import pandas as pd
from stripenn_amd import score as score_mod
score_mod.open_matrix = lambda cool: io.pixel_matrix(table)
t0 = time.time()
with contextlib.redirect_stdout(_io.StringIO()):
    res = score_mod.getScore('pixels:in-memory', 'gpurun_out/genome_out/result_unfiltered.tsv', 'weight', 8, 123456789,
                             'gpurun_out/genome_out/scores.tsv')
dt = time.time() - t0
ref = pd.read_csv('gpurun_out/genome_out/result_unfiltered.tsv', sep='\t', float_precision='round_trip')   # exactly parsed
print('score: %.1f s for %d stripes; p-values identical to compute: %s; Stripiness identical: %s'
      % (dt, len(res), bool(np.array_equal(res['pvalue_added'].to_numpy(), ref['pvalue'].to_numpy())),
         bool(np.array_equal(res['Stripiness_added'].to_numpy(), ref['Stripiness'].to_numpy(), equal_nan=True))))
