"""One-off randomised soak of the whole `compute` driver: HIP backend vs oracle backend on random small genomes
(dense-fetch and pixel-table sources), result TSVs compared byte for byte.
    python tools/soak_pipeline.py [first_seed] [count]"""
import contextlib, io as _io, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
warnings.filterwarnings('ignore')
import numpy as np
from oracle import oracle as O
from oracle_backend import OracleBackend
from stripenn_amd import backend as BK, io, pixels, stripenn, synth

O.build()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
t0 = time.time(); bad = 0; rows = 0; nexc = 0
for seed in range(first, first + count):
    if os.environ.get('STP_SOAK_VERBOSE'):
        print('seed', seed, flush=True)
    rng = np.random.default_rng(seed)
    resol = int(rng.choice([5000, 5000, 10000]))
    nchr = int(rng.integers(1, 4))
    names = ['chr%d' % (i + 1) for i in range(nchr)]
    chroms = {n: synth.SynthChrom(int(rng.integers(520, 1100)), 7000 + 10 * seed + i, stripe_every=int(rng.integers(40, 200)),
                                  stripe_gain=float(rng.uniform(2.0, 4.0)), nan_frac=float(rng.choice([0.0, 0.005, 0.03])))
              for i, n in enumerate(names)}
    use_pixels = bool(rng.integers(0, 2))
    if use_pixels:
        table = pixels.PixelTable.from_synth(names, chroms, resol)
        stripenn.open_matrix = lambda cool: io.pixel_matrix(table)
        norm = 'weight'
    else:
        sel = synth.SynthSelector(chroms, resol)
        sizes = [chroms[n].nbins * resol for n in names]
        stripenn.open_matrix = lambda cool: io.MatrixInfo(names, sizes, resol, ['chrom', 'start', 'end', 'weight', 'KR'], lambda b: sel)
        norm = 'KR'
    levels = ','.join('%.3f' % v for v in np.sort(rng.uniform(0.9, 0.995, int(rng.integers(1, 4)))))
    cores = int(rng.choice([1, 4]))
    sigma = float(rng.choice([2.0, 2.0, 2.5]))
    mask = '0' if rng.random() < 0.7 else '%s:%d-%d' % (names[0], 300 * resol, 305 * resol)
    gw = O.gauss_weights(sigma)[0]
    outs = []
    for tag, be in (('hip', BK.HipBackend(0)), ('oracle', OracleBackend(gauss_w=gw))):
        out = 'gpurun_out/soak_%s' % tag
        try:
            with contextlib.redirect_stdout(_io.StringIO()):
                stripenn.compute('x', out, norm, 'all', sigma, 10, 8, levels, cores, 0.5, mask, False, 3, 1000 + seed, force=True, backend=be)
            outs.append([open(os.path.join(out, f)).read() for f in ('result_unfiltered.tsv', 'result_filtered.tsv')])
        except Exception as e:                      # the reference's own exceptions (masking IndexError ...) must agree too
            outs.append('%s: %s' % (type(e).__name__, str(e)[:60]))
        be.close()
    same = (outs[0] == outs[1]) if not isinstance(outs[0], str) and not isinstance(outs[1], str) else \
           (isinstance(outs[0], str) and isinstance(outs[1], str) and outs[0].split(':')[0] == outs[1].split(':')[0])
    if not same:
        bad += 1
        print('MISMATCH seed', seed, 'pixels' if use_pixels else 'dense', levels, cores, sigma, mask,
              outs[0] if isinstance(outs[0], str) else 'tsv', outs[1] if isinstance(outs[1], str) else 'tsv', flush=True)
    elif isinstance(outs[0], str):
        nexc += 1
    else:
        rows += outs[0][0].count(chr(10)) - 1
print('%d genomes (%d ended in the same exception on both sides), %d unfiltered rows compared, %d mismatches, %.0f s' % (count, nexc, rows, bad, time.time() - t0))
