"""Randomised soak of the round-2 code paths on the GPU box; the summary lines are kept under profiles/.
For every random small genome (dense-fetch or pixel-table source, 1-3 levels, sigma 2.0 / 2.5, numcores 1 / 4,
sometimes a mask):
  * stripenn.compute with the HIP backend            == the same with the oracle backend      (TSVs byte for byte)
  * shard.sharded_compute, 2-4 rank threads (HIP)    == the unsharded HIP run                 (frame spans + halo frames)
  * per chromosome: the stripe search cut into random frame spans, all kept in flight (stp_stripe_search_begin),
    concatenated                                      == the single synchronous search
    python tools/soak_round2.py [first_seed] [count]"""
import contextlib, io as _io, os, sys, threading, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
warnings.filterwarnings('ignore')
import numpy as np
import stripenn_amd.io as iomod
from oracle import oracle as O
from oracle_backend import OracleBackend
from stripenn_amd import backend as BK, hip, pixels, shard, stripenn, synth

O.build()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10


class ThreadComm:
    class Shared:
        def __init__(self, world):
            self.slots = [None] * world
            self.barrier = threading.Barrier(world, timeout=600)

    def __init__(self, rank, world, sh):
        self.rank, self.world, self.sh = rank, world, sh

    def allgather(self, obj):
        self.sh.slots[self.rank] = obj
        self.sh.barrier.wait()
        out = list(self.sh.slots)
        self.sh.barrier.wait()
        return out

    def barrier(self):
        self.sh.barrier.wait()


def tsvs(out):
    return [open(os.path.join(out, f)).read() for f in ('result_unfiltered.tsv', 'result_filtered.tsv')]


t0 = time.time()
bad = rows = nexc = nshard = nspan = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    resol = int(rng.choice([5000, 5000, 10000]))
    nchr = int(rng.integers(1, 5))
    names = ['chr%d' % (i + 1) for i in range(nchr)]
    chroms = {n: synth.SynthChrom(int(rng.integers(520, 1500)), 9000 + 10 * seed + i, stripe_every=int(rng.integers(40, 200)),
                                  stripe_gain=float(rng.uniform(2.0, 4.0)), nan_frac=float(rng.choice([0.0, 0.005, 0.03])))
              for i, n in enumerate(names)}
    use_pixels = bool(rng.integers(0, 2))
    if use_pixels:
        table = pixels.PixelTable.from_synth(names, chroms, resol)
        opener = lambda cool: iomod.pixel_matrix(table)
        norm = 'weight'
    else:
        sel = synth.SynthSelector(chroms, resol)
        sizes = [chroms[n].nbins * resol for n in names]
        opener = lambda cool: iomod.MatrixInfo(names, sizes, resol, ['chrom', 'start', 'end', 'weight', 'KR'], lambda b: sel)
        norm = 'KR'
    stripenn.open_matrix = opener
    levels = ','.join('%.3f' % v for v in np.sort(rng.uniform(0.9, 0.995, int(rng.integers(1, 4)))))
    cores = int(rng.choice([1, 4]))
    sigma = float(rng.choice([2.0, 2.0, 2.5]))
    mask = '0' if rng.random() < 0.7 else '%s:%d-%d' % (names[0], 300 * resol, 305 * resol)
    gw = O.gauss_weights(sigma)[0] if sigma not in (2.0, 2.5) else hip.gauss_weights(sigma)[0]
    outs = []
    for tag, be in (('hip', BK.HipBackend(0)), ('oracle', OracleBackend(gauss_w=gw))):
        out = 'gpurun_out/soak2_%s' % tag
        try:
            with contextlib.redirect_stdout(_io.StringIO()):
                stripenn.compute('x', out, norm, 'all', sigma, 10, 8, levels, cores, 0.5, mask, False, 3, 1000 + seed, force=True, backend=be)
            outs.append(tsvs(out))
        except Exception as e:                      # the reference's own exceptions (masking IndexError ...) must agree too
            outs.append('%s: %s' % (type(e).__name__, str(e)[:60]))
        be.close()
    if isinstance(outs[0], str) or isinstance(outs[1], str):
        same = isinstance(outs[0], str) and isinstance(outs[1], str) and outs[0].split(':')[0] == outs[1].split(':')[0]
        nexc += same
        if not same:
            bad += 1
            print('MISMATCH (exception) seed', seed, outs[0] if isinstance(outs[0], str) else 'tsv', outs[1] if isinstance(outs[1], str) else 'tsv', flush=True)
        continue
    if outs[0] != outs[1]:
        bad += 1
        print('MISMATCH hip/oracle seed', seed, 'pixels' if use_pixels else 'dense', levels, cores, sigma, mask, flush=True)
        continue
    rows += outs[0][0].count('\n') - 1
    # ---- the frame-span driver with rank threads
    world = int(rng.integers(2, 5))
    sh = ThreadComm.Shared(world)
    orig_open, orig_comm = iomod.open_matrix, shard._Comm
    iomod.open_matrix = opener
    shard._Comm = lambda rank, w: ThreadComm(rank, w, sh)
    errs, made = [], []

    def factory(r):
        made.append(BK.HipBackend(0))
        return made[-1]

    def body(rank):
        try:
            shard.sharded_compute(rank, world, 'x', 'gpurun_out/soak2_shard', norm, 'all', sigma, 10, 8, levels, cores, 0.5, mask,
                                  False, 3, 1000 + seed, force=True, backend_factory=factory)
        except BaseException as e:                  # noqa: BLE001
            errs.append((rank, repr(e)))
            sh.barrier.abort()
    try:
        with contextlib.redirect_stdout(_io.StringIO()):
            th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
            [t.start() for t in th]
            [t.join() for t in th]
    finally:
        iomod.open_matrix, shard._Comm = orig_open, orig_comm
        for b in made:
            b.close()
    if errs or tsvs('gpurun_out/soak2_shard') != outs[0]:
        bad += 1
        print('MISMATCH sharded world', world, 'seed', seed, errs[:1], flush=True)
    else:
        nshard += 1
    # ---- searches in flight over random frame spans == one search
    ctx = hip.Context(0)
    for n in names:
        ch = chroms[n]
        band = ctx.band_upload(ch.band(512))
        nfr = -(-ch.nbins // 200)
        st = np.array([max(0, i * 200 - 100) for i in range(nfr)], dtype=np.int32)
        en = np.minimum((np.arange(nfr) + 1) * 200 + 99, ch.nbins - 1).astype(np.int32)
        blk = ch.block(0, min(600, ch.nbins), 0, min(600, ch.nbins))
        Ms = np.quantile(blk[blk > 0], [0.95, 0.98])
        fr = band.frames(st, en)
        whole = fr.stripe_search(Ms, sigma=sigma)
        cuts = sorted(set([0, nfr] + rng.integers(1, max(2, nfr), size=int(rng.integers(1, 4))).tolist()))
        parts = [band.frames(st[a:b], en[a:b]) for a, b in zip(cuts, cuts[1:])]
        pend = [p.stripe_search_begin(Ms, sigma=sigma) for p in parts]
        got = []
        for a, p, q in zip(cuts, parts, pend):
            r = q.wait().copy(); r['frame'] += a; got.append(r); p.close()
        if np.concatenate(got).tobytes() != whole.tobytes():
            bad += 1
            print('MISMATCH spans seed', seed, n, cuts, flush=True)
        else:
            nspan += 1
        fr.close(); band.close()
    ctx.close()
print('%d genomes from seed %d: %d unfiltered rows HIP == oracle backend byte for byte (%d runs ended in the same exception on '
      'both sides), %d sharded runs (2-4 rank threads) == unsharded, %d chromosomes span-invariant; %d mismatches; %.0f s'
      % (count, first, rows, nexc, nshard, nspan, bad, time.time() - t0))
