"""Host-side profile of `stripenn score` on the mm10-size genome (after a `compute` run wrote result_unfiltered.tsv):
    python tools/probe_genome.py && python tools/probe_score.py"""
import contextlib, io as _io, os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stripenn_amd import io, score as score_mod, stripenn, synth_device

MM10 = [195471971, 182113224, 160039680, 156508116, 151834684, 149736546, 145441459, 129401213, 124595110, 130694993,
        122082543, 120129022, 120421639, 124902244, 104043685, 98207768, 94987271, 90702639, 61431566, 171031299]
names = ['chr%d' % (i + 1) for i in range(19)] + ['chrX']
torch.cuda.init()
chroms = {n: synth_device.DeviceChrom(int(-(-s // 5000)), 1 + i, torch.device('cuda', 0)) for i, (n, s) in enumerate(zip(names, MM10))}
table = synth_device.pixel_table(names, chroms, 5000)
stripenn.open_matrix = lambda cool: io.pixel_matrix(table)
score_mod.open_matrix = lambda cool: io.pixel_matrix(table)
os.makedirs('gpurun_out', exist_ok=True)
if not os.path.exists('gpurun_out/genome_out/result_unfiltered.tsv'):
    with contextlib.redirect_stdout(_io.StringIO()):
        stripenn.compute('pixels:in-memory', 'gpurun_out/genome_out', 'weight', 'all', 2.0, 10, 8, '0.95,0.96,0.97,0.98,0.99', 8,
                         0.1, '0', False, 3, 123456789, force=True)


def run():
    with contextlib.redirect_stdout(_io.StringIO()):
        return score_mod.getScore('pixels:in-memory', 'gpurun_out/genome_out/result_unfiltered.tsv', 'weight', 8, 123456789,
                                  'gpurun_out/genome_out/scores.tsv')


run()
pr = cProfile.Profile(); t0 = time.time(); pr.enable(); res = run(); pr.disable()
print('score (warm): %.2f s for %d stripes' % (time.time() - t0, len(res)))
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
