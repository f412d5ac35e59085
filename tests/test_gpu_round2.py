"""Round-2 GPU parity: the score kernels at configs[1] size on the arrays bench.py times, a whole `compute`
driver on a six-chromosome genome (HIP vs oracle backend, pixel-table source), sampled frames of the
248 957-bin 1 kb-style band, the product's own Gaussian weights against the reference's tables, the
dense-matrix StripeSearch call, the nearest-pixel table of the band packer and the frame-span driver."""
import contextlib
import io
import os

import numpy as np
import pandas as pd
import pytest

from oracle import oracle as O
from oracle_backend import OracleBackend

pytestmark = pytest.mark.gpu
HW = 512


def _frame_table(nbins):
    nfr = -(-nbins // 200)
    st = np.array([max(0, i * 200 - 100) for i in range(nfr)], dtype=np.int32)
    en = np.minimum((np.arange(nfr) + 1) * 200 + 99, nbins - 1).astype(np.int32)
    return st, en


# --------------------------------------------------------------------------------------------- 5a
def test_chr16_scores_of_every_candidate_match_oracle():
    """configs[1]: p-value, Stripiness, O/E mean / total and observed mean / sum of ALL candidate stripes of
    the chr16-size sweep (the arrays bench.py's chr16 workload times), HIP kernels vs oracle.py on the same
    explicit inputs: p-values and Stripiness bit-exact, the sums within 1e-9 relative."""
    from stripenn_amd import synth, getStripe as GS, backend as BK
    nbins = 19642
    ch = synth.SynthChrom(nbins, 16)
    band_h = ch.band(HW)
    st, en = _frame_table(nbins)
    Ms = np.quantile(band_h[band_h > 0], [0.95, 0.96, 0.97, 0.98, 0.99])
    hb = BK.HipBackend(0)
    name, size = 'chr16', nbins * 5000
    sel = synth.SynthSelector({name: ch}, 5000)
    obj = GS.getStripe(sel, 5000, 10, 8, 2.0, [name], [name], np.array([size]), np.array([size]), 2, 3, 123456789, backend=hb)
    sband = obj._bands[name] = hb.ctx.band_upload(band_h)
    EV = np.asarray(obj.mpmean()[name])
    bg = obj.nulldist()
    hb.set_background(*bg)
    fr = sband.frames(st, en)
    recs = fr.stripe_search(Ms)
    assert len(recs) > 10000
    pv, sc = BK.score_inputs(recs, fr.nz, st, nbins, 10)
    p = hb.pvalue(sband, 10, pv)
    g, cm, ct = hb.stripiness(sband, EV, sc)
    rects = np.zeros(len(recs), dtype=BK.RECT_DTYPE)
    rects['row0'], rects['row1'], rects['col0'], rects['col1'] = sc['row0'], sc['row1'], sc['col0'][:, 0], sc['col1'][:, 0]
    om, osum = hb.stripe_mean(sband, rects)
    ob = OracleBackend()
    oband = ob.open_chrom(band_h)
    ob.set_background(*bg)
    # every 3rd candidate through the (per-stripe, Python) oracle keeps the test within a minute; the stride
    # walks all frames, levels and brightness images
    idx = np.arange(0, len(recs), 3)
    ep = ob.pvalue(oband, 10, pv[idx])
    eg, ecm, ect = ob.stripiness(oband, EV, sc[idx])
    em, es = ob.stripe_mean(oband, rects[idx])
    assert np.array_equal(p[idx], ep), 'p-values differ'
    assert np.array_equal(g[idx], eg, equal_nan=True), 'Stripiness differs'
    assert np.allclose(cm[idx], ecm, rtol=1e-9, atol=0, equal_nan=True) and np.allclose(ct[idx], ect, rtol=1e-9, atol=0, equal_nan=True)
    assert np.allclose(om[idx], em, rtol=1e-9, atol=0, equal_nan=True) and np.allclose(osum[idx], es, rtol=1e-9, atol=0, equal_nan=True)
    fr.close()
    hb.close()


# --------------------------------------------------------------------------------------------- 5b
def _six_chrom_table():
    from stripenn_amd import pixels, synth
    nb = [2300, 1900, 1600, 1250, 1100, 700]
    names = ['chr%d' % (i + 1) for i in range(6)]
    chroms = {n: synth.SynthChrom(b, 71 + i) for i, (n, b) in enumerate(zip(names, nb))}
    return names, chroms, pixels.PixelTable.from_synth(names, chroms, 5000)


def _compute(table, out, backend=None, gpus_args=None, **kw):
    from stripenn_amd import io as sio, stripenn
    orig = stripenn.open_matrix
    stripenn.open_matrix = lambda cool: sio.pixel_matrix(table)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            stripenn.compute('pixels:in-memory', out, 'weight', 'all', 2.0, 10, 8, kw.get('maxpixel', '0.96,0.98,0.99'),
                             kw.get('numcores', 4), 0.2, kw.get('mask', '0'), False, 3, 123456789, force=True, backend=backend)
    finally:
        stripenn.open_matrix = orig
    return [open(os.path.join(out, n)).read() for n in ('result_unfiltered.tsv', 'result_filtered.tsv')]


@pytest.mark.parametrize('numcores', [4, 1])
def test_six_chromosome_compute_hip_equals_oracle_backend(tmp_path, numcores):
    """The whole driver (quantiles -> expected values -> background -> candidates -> p-values -> redundancy filters
    -> Stripiness -> TSVs) on a six-chromosome genome handed over as cooler's pixel table: HIP backend vs oracle
    backend, byte-identical TSVs (the soak of round 1, tools/soak_pipeline.py, as a test)."""
    from stripenn_amd import backend as BK
    names, chroms, table = _six_chrom_table()
    hb = BK.HipBackend(0)
    a = _compute(table, str(tmp_path / 'hip'), backend=hb, numcores=numcores)
    b = _compute(table, str(tmp_path / 'ora'), backend=OracleBackend(), numcores=numcores)
    hb.close()
    assert len(a[0].splitlines()) > 100
    assert a[0] == b[0] and a[1] == b[1]


class _ThreadComm:
    """all_gather of host objects between rank THREADS of one process (one HIP context each): the multi-rank data
    path of stripenn_amd/shard.py on a one-GPU box, without a process group."""

    class Shared:
        def __init__(self, world):
            import threading
            self.slots = [None] * world
            self.barrier = threading.Barrier(world, timeout=300)

    def __init__(self, rank, world, shared):
        self.rank, self.world, self.sh = rank, world, shared

    def allgather(self, obj):
        self.sh.slots[self.rank] = obj
        self.sh.barrier.wait()
        out = list(self.sh.slots)
        self.sh.barrier.wait()
        return out

    def barrier(self):
        self.sh.barrier.wait()


@pytest.mark.parametrize('world,numcores', [(1, 4), (3, 4), (2, 1)])
def test_sharded_driver_hip_equals_unsharded(tmp_path, world, numcores):
    """shard.sharded_compute with the HIP backend -- world 1, and worlds 2 / 3 as rank threads with one HIP context
    each (frame spans cut chromosomes in the middle; numcores 1 = one PRNG stream replayed by every rank) -- writes
    the TSVs of stripenn.compute byte for byte."""
    import threading
    import stripenn_amd.io as iomod
    from stripenn_amd import backend as BK, shard
    names, chroms, table = _six_chrom_table()
    hb = BK.HipBackend(0)
    ref = _compute(table, str(tmp_path / 'plain'), backend=hb, numcores=numcores)
    hb.close()
    orig_open, orig_comm = iomod.open_matrix, shard._Comm
    iomod.open_matrix = lambda cool: iomod.pixel_matrix(table)
    shared = _ThreadComm.Shared(world)
    shard._Comm = lambda rank, w: _ThreadComm(rank, w, shared)
    errors, made = [], []

    def factory(r):
        made.append(BK.HipBackend(0))
        return made[-1]

    def body(rank):
        try:
            shard.sharded_compute(rank, world, 'pixels:in-memory', str(tmp_path / 'w'), 'weight', 'all', 2.0, 10, 8,
                                  '0.96,0.98,0.99', numcores, 0.2, '0', False, 3, 123456789, force=True,
                                  backend_factory=factory)
        except BaseException as e:                      # noqa: BLE001 -- reported by the main thread
            errors.append((rank, repr(e)))
            shared.barrier.abort()
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
            for t in th:
                t.start()
            for t in th:
                t.join()
    finally:
        iomod.open_matrix, shard._Comm = orig_open, orig_comm
        for b in made:
            b.close()
    assert not errors, errors
    got = [open(str(tmp_path / 'w' / n)).read() for n in ('result_unfiltered.tsv', 'result_filtered.tsv')]
    assert got == ref


# --------------------------------------------------------------------------------------------- 5c
def test_1kb_band_sampled_frames_match_oracle(hip_ctx):
    """configs[4]: the 248 957-bin band (generated on the device), all 1 245 frames x 5 levels searched in one
    call; 24 sampled frames x 5 levels are compared with the oracle record by record."""
    import torch
    from stripenn_amd import synth_device
    nbins = 248957
    dc = synth_device.DeviceChrom(nbins, 5, torch.device('cuda', 0))
    t = dc.band(HW)
    torch.cuda.synchronize()
    band = hip_ctx.band_wrap(t.data_ptr(), nbins, HW, keepalive=t)
    st, en = _frame_table(nbins)
    v = torch.sort(t[:40000][t[:40000] > 0]).values
    Ms = [float(v[int(q * (v.numel() - 1))]) for q in (0.95, 0.96, 0.97, 0.98, 0.99)]
    fr = band.frames(st, en)
    recs = fr.stripe_search(Ms)
    assert len(recs) > 100000
    rng = np.random.default_rng(7)
    frames = sorted(set([0, 1, len(st) - 1, len(st) - 2] + rng.choice(len(st), 20, replace=False).tolist()))
    for fi in frames:
        s, e = int(st[fi]), int(en[fi])
        rows = t[s:e + 1].cpu().numpy()
        rr = np.arange(s, e + 1)[:, None]
        cc = np.arange(s, e + 1)[None, :]
        D = rows[rr - s, cc - rr + HW]
        D[np.isnan(D)] = 0
        nz = np.where(D.sum(axis=0) != 0)[0]
        assert fr.S[fi] == (len(nz) if len(nz) > 10 else 0)
        assert np.array_equal(fr.nz[fi, :len(nz)], nz)
        Dc = np.ascontiguousarray(D[np.ix_(nz, nz)])
        exp = []
        for li, M in enumerate(Ms):
            r, tot = O.stripe_search(Dc, M)
            exp += [(li,) + tuple(int(x) for x in q) + (float(tt),) for q, tt in zip(r, tot)]
        mine = recs[recs['frame'] == fi]
        got = [(int(r['level']), int(r['b_index']), int(r['ud']), int(r['x']), int(r['y']), int(r['w']), int(r['h']),
                float(r['total'])) for r in mine]
        assert got == exp, 'frame %d' % fi
    fr.close(); band.close()


# --------------------------------------------------------------------------------------------- 6
def test_product_gauss_weights_reproduce_the_reference_stage_tables(hip_ctx, chr7, golden_stages):
    """No weight injection: hip.gauss_weights() under THIS box's numpy must give the reference's edge maps,
    vertical-line maps and stripe rows of the 36 golden images."""
    g = golden_stages
    band = hip_ctx.band_upload(chr7.band(HW))
    n = int(g['ncases'])
    fr = band.frames([int(g['c%d_start' % i]) for i in range(n)], [int(g['c%d_end' % i]) for i in range(n)])
    for ci in range(n):
        p = 'c%d_' % ci
        M = float(g[p + 'M'])
        for bi in range(len(g['bvals'])):
            got = fr.dbg_stages(ci, M, bi)                      # default weights
            assert np.array_equal(np.packbits(got['edges'].astype(bool), axis=1), g[p + 'edges'][bi])
            assert np.array_equal(np.packbits(got['vert'].astype(bool), axis=1), g[p + 'vert'][bi])
        recs = fr.stripe_search([M])
        mine = recs[recs['frame'] == ci]
        assert np.array_equal(np.stack([mine['x'], mine['y'], mine['h'], mine['w']], axis=1).astype(np.int64).reshape(-1, 4),
                              g[p + 'rec_xywh'])
        assert np.array_equal(mine['total'], g[p + 'rec_total'])
    fr.close(); band.close()


def test_product_gauss_weights_end_to_end():
    """The complete compute and score pipelines with the product's own weights reproduce the reference's TSVs."""
    import e2e_common
    from stripenn_amd.backend import HipBackend
    obj, _ = e2e_common.run_compute(lambda gw: HipBackend(0), True)
    obj.backend.close()
    e2e_common.run_score(lambda gw: HipBackend(0), True)


# --------------------------------------------------------------------------------------------- 9
def test_stripe_search_dense_call_matches_reference_rows(chr7, golden_stages):
    """getStripe.StripeSearch(submat, ...) as search_frame calls it (getStripe.py:822): the golden rows were
    produced by exactly this call of the reference (gen_golden.py); kept rows and bp coordinates must match."""
    from stripenn_amd import getStripe as GS, synth
    from stripenn_amd.backend import HipBackend
    g = golden_stages
    resol = int(g['resol'])
    size = int(g['chromsize'])
    sel = synth.SynthSelector({'chr7': chr7}, resol)
    hb = HipBackend(0)
    obj = GS.getStripe(sel, resol, 10, 8, 2.0, ['chr7'], ['chr7'], np.array([size]), np.array([size]), 1, 3, 1, backend=hb)
    best = None
    for ci in range(int(g['ncases'])):
        p = 'c%d_' % ci
        start, end = int(g[p + 'start']), int(g[p + 'end'])
        fs = end - start + 1
        start_array = [(start + j) * resol + 1 for j in range(fs)]
        end_array = [s + resol - 1 for s in start_array]
        if end_array[-1] >= size:
            end_array[-1] = size
        locus = 'chr7:%d-%d' % (start_array[0], end_array[-1])
        D = GS.nantozero(np.array(sel.fetch(locus, locus), dtype=np.float64))
        nz = np.where(np.sum(D, axis=0) != 0)[0]
        D = D[np.ix_(nz, nz)]
        sa = [start_array[s] for s in nz]
        ea = [end_array[s] for s in nz]
        res = obj.StripeSearch(D, int(g[p + 'idx']), start, end, float(g[p + 'M']), 0.99, 'chr7', len(nz), sa, ea)
        kept = g[p + 'kept_xywh']
        assert np.array_equal(res[['x', 'y', 'h', 'w']].to_numpy(dtype=np.int64).reshape(-1, 4), kept), 'case %d' % ci
        # bp coordinates of the kept rows, looked up among the reference's raw rows
        raw_xywh, raw_pos = g[p + 'rec_xywh'], g[p + 'rec_pos']
        lut = {tuple(r): tuple(q) for r, q in zip(raw_xywh.tolist(), raw_pos.tolist())}
        for r, q in zip(res[['x', 'y', 'h', 'w']].to_numpy(dtype=np.int64).tolist(),
                        res[['pos1', 'pos2', 'pos3', 'pos4']].to_numpy(dtype=np.int64).tolist()):
            assert lut[tuple(r)] == tuple(q)
        if len(res):
            assert float(res['medpixel'].iloc[0]) == float(g[p + 'medpixel'])
            assert list(res.columns) == GS.EXTRACT_COLUMNS
        if best is None or len(res) > best[0]:
            best = (len(res), D, sa, ea, float(g[p + 'M']))
    # a matrix WITH empty columns keeps them (no second compaction inside StripeSearch)
    _, D, sa, ea, M = best
    D2 = D.copy()
    D2[:, 5] = 0.0; D2[5, :] = 0.0
    a = obj.StripeSearch(D2, 0, 0, len(D2) - 1, M, 0.99, 'chr7', len(D2), sa, ea)
    r, tot = O.stripe_search(np.ascontiguousarray(D2), M)
    raw = {(int(q[2]), int(q[3]), int(q[4]), int(q[5])) for q in r}                 # (x, y, w, h) of the oracle's raw rows
    mine = {tuple(v) for v in a[['x', 'y', 'w', 'h']].to_numpy(dtype=np.int64).tolist()}
    assert raw and mine and mine <= raw
    hb.close()


# --------------------------------------------------------------------------------------------- packer
def test_band_nearest_table_and_row_queries():
    """stp_band_nearest vs its numpy restatement, on a table with pixels beyond the band's halfwidth, NaN weights
    and zero counts; and the facade's pools from it equal the pools from dense fetches."""
    from stripenn_amd import backend as BK, getStripe as GS, pixels, synth
    names = ['chrA', 'chrB']
    chroms = {n: synth.SynthChrom(nb, 5 + k, nan_frac=0.03) for k, (n, nb) in enumerate(zip(names, (2600, 1400)))}
    t = pixels.PixelTable.from_synth(names, chroms, 5000)
    # sprinkle far-off-diagonal pixels and empty a few rows
    rng = np.random.default_rng(3)
    keep = ~np.isin(t.bin1_id, [40, 41, 42, 900]) & ~np.isin(t.bin2_id, [40, 41, 42, 900])
    far1 = np.array([10, 40, 700, 901, 2700]); far2 = np.array([1900, 1500, 2599, 2400, 3999])
    b1 = np.concatenate([t.bin1_id[keep], far1]); b2 = np.concatenate([t.bin2_id[keep], far2])
    cn = np.concatenate([t.count[keep], np.array([3, 1, 2, 0, 5], dtype=np.int32)])
    o = np.lexsort((b2, b1))
    t2 = pixels.PixelTable(names, t.chromsizes, 5000, t.chrom_offset, b1[o], b2[o], cn[o], t.weights)
    hb = BK.HipBackend(0)
    sel = pixels.PixelSelector(t2, True)
    for nm in names:
        px = sel.chrom_pixels(nm)
        band = hb.pack_chrom(px, HW)
        got = hb.band_nearest(band)
        exp = O.nearest_from_pixels(px['bin1'], px['bin2'], px['count'], px['weight'], px['lo'], px['nrows'])
        assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])
        assert np.array_equal(band.download(), O.band_from_pixels(px['bin1'], px['bin2'], px['count'], px['weight'], px['lo'],
                                                                  px['nrows'], HW), equal_nan=True)
        band.close()
    sizes = t2.chromsizes

    class FetchOnly:
        def __init__(self, s):
            self.fetch = s.fetch
    a = GS.getStripe(sel, 5000, 10, 8, 2.0, names, names, sizes, sizes, 2, 3, 1, backend=hb)
    b = GS.getStripe(FetchOnly(sel), 5000, 10, 8, 2.0, names, names, sizes, sizes, 2, 3, 1, backend=hb)
    for nm in names:
        assert a._nearest(nm) is not None and b._nearest(nm) is None
        assert a.null_available_cols(nm) == b.null_available_cols(nm)
        assert a.null_pools(nm) == b.null_pools(nm)
    hb.close()


# --------------------------------------------------------------------------------------------- seeimage (8f-4)
def test_seeimage_window_plane_is_the_references_arithmetic(tmp_path):
    """stp_window_plane (the image-build arithmetic on the resident band) vs the reference's numpy lines
    (seeimage.py:78-85 = oracle.window_rgb) on windows with NaN bins, on and off the diagonal; and the CLI-level
    function writes one PNG per level whose pixels are that array."""
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.image as mpimg
    from stripenn_amd import backend as BK, pixels, seeimage, synth
    names = ['chrA', 'chrB']
    chroms = {'chrA': synth.SynthChrom(900, 61, nan_frac=0.02), 'chrB': synth.SynthChrom(500, 62)}
    t = pixels.PixelTable.from_synth(names, chroms, 5000)
    sel = pixels.PixelSelector(t, 'weight')
    hb = BK.HipBackend(0)
    band = hb.pack_chrom(sel.chrom_pixels('chrA'), 512)
    D = sel.fetch('chrA')
    M = float(np.quantile(D[D > 0], 0.97))
    for (r0, nr, c0, nc) in ((200, 200, 200, 200), (0, 300, 0, 300), (100, 50, 400, 90), (880, 20, 700, 200)):
        got = hb.window_plane(band, r0, nr, c0, nc, M)
        exp = O.window_rgb(D[r0:r0 + nr, c0:c0 + nc], M)[..., 1]
        assert np.array_equal(got, exp, equal_nan=True)
    band.close()
    p = str(tmp_path / 't.npz'); t.save(p)
    pos = 'chrA:1000001-2000000'
    files = seeimage.seeimage('pixels:' + p, pos, '0.95,0.99', 'weight', str(tmp_path / 'heat'), False, 1, backend=hb)
    assert len(files) == 2 and all(os.path.getsize(f) > 1000 for f in files)
    png = mpimg.imread(files[0])
    assert png.ndim == 3 and png.shape[2] >= 3 and (png[..., 0] > 0.9).mean() > 0.2      # the red plane of the heat map is there
    hb.close()


# --------------------------------------------------------------------------------------------- genome-size properties
def test_chr1_size_search_is_deterministic_and_span_invariant(hip_ctx):
    """configs[2] at full chromosome size (mm10 chr1 at 5 kb: 39 095 bins, 196 frames x 5 levels x 6 brightness, band
    generated on the device): size-independent properties of the search -- the same call twice gives identical records;
    three searches kept in flight over three frame spans give, concatenated, the records of the single call (what the
    multi-GPU driver and bench.py rely on); frame numbers, levels and slots come out in the reference's order."""
    import torch
    from stripenn_amd import synth_device
    nbins = 39095
    dc = synth_device.DeviceChrom(nbins, 1, torch.device('cuda', 0))
    t = dc.band(HW)
    torch.cuda.synchronize()
    band = hip_ctx.band_wrap(t.data_ptr(), nbins, HW, keepalive=t)
    st, en = _frame_table(nbins)
    v = torch.sort(t[:20000][t[:20000] > 0]).values
    Ms = [float(v[int(q * (v.numel() - 1))]) for q in (0.95, 0.96, 0.97, 0.98, 0.99)]
    fr = band.frames(st, en)
    a = fr.stripe_search(Ms)
    b = fr.stripe_search(Ms)
    assert len(a) > 20000 and a.tobytes() == b.tobytes()
    key = a['frame'].astype(np.int64) * 1000 + a['level'] * 10 + a['b_index']
    assert np.all(np.diff(key) >= 0)                       # (frame, level, brightness) order
    cuts = [0, 70, 131, len(st)]
    parts = [band.frames(st[lo:hi], en[lo:hi]) for lo, hi in zip(cuts, cuts[1:])]
    pend = [p.stripe_search_begin(Ms) for p in parts]      # all three in flight on the one stream
    got = []
    for lo, p, q in zip(cuts, parts, pend):
        r = q.wait().copy()
        r['frame'] += lo
        got.append(r)
        p.close()
    got = np.concatenate(got)
    assert got.tobytes() == a.tobytes()
    fr.close(); band.close()


# --------------------------------------------------------------------------------------------- configs[3]: compute, then score
def test_score_rescoring_reproduces_the_compute_columns(tmp_path):
    """configs[3]'s flow: `compute` calls stripes, `score` re-scores the written table.  With the same seed and
    numcores the background tables are the same, so the added p-value and Stripiness columns must equal the ones
    `compute` wrote for the same stripes (a size-independent idempotence property; checked here on the
    six-chromosome genome, and on the whole mm10-size genome by tools/probe_genome.py)."""
    import stripenn_amd.score as score_mod
    from stripenn_amd import backend as BK, io as sio
    names, chroms, table = _six_chrom_table()
    hb = BK.HipBackend(0)
    out = str(tmp_path / 'c')
    _compute(table, out, backend=hb, numcores=4)
    orig = score_mod.open_matrix
    score_mod.open_matrix = lambda cool: sio.pixel_matrix(table)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            res = score_mod.getScore('pixels:in-memory', os.path.join(out, 'result_unfiltered.tsv'), 'weight', 4, 123456789,
                                     str(tmp_path / 'scores.tsv'), backend=hb)
    finally:
        score_mod.open_matrix = orig
    hb.close()
    assert len(res) > 100
    # (score reads the table with pandas' default float parser like the reference, score.py:11, which may be 1 ulp
    #  off; the columns compute wrote are compared as exactly parsed numbers)
    ref = pd.read_csv(os.path.join(out, 'result_unfiltered.tsv'), sep='\t', float_precision='round_trip')
    assert np.array_equal(res['pvalue_added'].to_numpy(), ref['pvalue'].to_numpy())
    assert np.array_equal(res['Stripiness_added'].to_numpy(), ref['Stripiness'].to_numpy(), equal_nan=True)
    assert np.allclose(res['O_Mean_added'].to_numpy(), ref['Mean'].to_numpy(), rtol=1e-9, atol=0)


# --------------------------------------------------------------------------------------------- float-count coolers
def test_float_count_table_through_the_device(tmp_path):
    """pixels/count as float64 (merged / scaled / --count-as-float coolers): the packer, the nearest-pixel table and the
    quantile select take the column as it is (STP_COUNT_F64); the whole driver on such a table: HIP == oracle backend."""
    from stripenn_amd import backend as BK, pixels, synth
    names = ['chr1', 'chr2']
    chroms = {n: synth.SynthChrom(nb, 33 + k) for k, (n, nb) in enumerate(zip(names, (1300, 800)))}
    t = pixels.PixelTable.from_synth(names, chroms, 5000)
    scale = np.where(np.arange(len(t.count)) % 7 == 0, 0.37, 1.6180339887)           # non-integer, not exactly representable products
    tf = pixels.PixelTable(names, t.chromsizes, 5000, t.chrom_offset, t.bin1_id, t.bin2_id, t.count * scale, t.weights)
    assert tf.count.dtype == np.float64
    hb = BK.HipBackend(0)
    sel = pixels.PixelSelector(tf, True)
    for nm in names:
        px = sel.chrom_pixels(nm)
        s = hb.select_open()
        band = hb.pack_chrom(px, HW, s)
        assert np.array_equal(band.download(), O.band_from_pixels(px['bin1'], px['bin2'], px['count'], px['weight'], px['lo'],
                                                                  px['nrows'], HW), equal_nan=True)
        D = sel.fetch(nm)
        pos = np.sort(D[D > 0])
        assert hb.select_count(s) == len(pos)
        ranks = np.array([0, len(pos) // 3, len(pos) - 1])
        assert np.array_equal(hb.select_ranks(s, ranks), pos[ranks])
        hb.select_close(s); band.close()
    a = _compute(tf, str(tmp_path / 'hip'), backend=hb)
    b = _compute(tf, str(tmp_path / 'ora'), backend=OracleBackend())
    hb.close()
    assert len(a[0].splitlines()) > 20 and a == b
