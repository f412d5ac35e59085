"""Whole-genome runs at CONFIG size inside the driver-run GPU suite (missing #3).

configs[2]: the 20 mm10 chromosome sizes at 5 kb (526 765 bins, 2 645 frames, 266 M stored pixels) x 5 maxpixel levels
through the unmodified driver `stripenn_amd.stripenn.compute` from an in-memory pixel table: determinism, table
invariants, and >= 40 frames sampled across ALL chromosomes against the oracle record by record (every level, every
brightness image).  Driver loop: stripenn.py:126-159.

configs[3]: the 23 hg38 chromosome sizes (606 k bins, 3 044 frames) x 3 levels 0.97-0.99, `compute`, then `stripenn
score` re-scoring of the called stripes (score.py:49-60): the added columns reproduce compute's, and 200 sampled
stripes' p-value / Stripiness equal oracle.py's per-stripe restatements bit for bit.

The genomes are generated on the device (stripenn_amd.synth_device: the same pixel function as synth.SynthChrom)."""
import contextlib
import io
import multiprocessing as mp
import os

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu
RESOL = 5000
MM10 = [195471971, 182113224, 160039680, 156508116, 151834684, 149736546, 145441459, 129401213, 124595110, 130694993,
        122082543, 120129022, 120421639, 124902244, 104043685, 98207768, 94987271, 90702639, 61431566, 171031299]
HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
        135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
        46709983, 50818468, 156040895]
_G = {}


def _genome(sizes_bp, seed0, names):
    import torch
    from stripenn_amd import synth_device
    torch.cuda.init()
    dev = torch.device('cuda', 0)
    chroms = {n: synth_device.DeviceChrom(-(-s // RESOL), seed0 + i, dev) for i, (n, s) in enumerate(zip(names, sizes_bp))}
    table = synth_device.pixel_table(names, chroms, RESOL)
    table.chromsizes = np.asarray(sizes_bp, dtype=np.int64)         # true bp sizes: the last bin of a chromosome is partial
    return chroms, table


def _run_compute(table, out, maxpixel, numcores=8, pcut=0.1):
    from stripenn_amd import io as sio, stripenn
    orig = stripenn.open_matrix
    stripenn.open_matrix = lambda cool: sio.pixel_matrix(table)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            stripenn.compute('pixels:in-memory', out, 'weight', 'all', 2.0, 10, 8, maxpixel, numcores, pcut, '0', False, 3,
                             123456789, force=True)
    finally:
        stripenn.open_matrix = orig
    return [open(os.path.join(out, n)).read() for n in ('result_unfiltered.tsv', 'result_filtered.tsv')]


def _oracle_frame(args):
    from oracle import oracle as O
    ci, fi = args
    D, Ms = _G['D'][(ci, fi)], _G['Ms'][ci]
    D = np.where(np.isnan(D), 0.0, D)
    nz = np.where(D.sum(axis=0) != 0)[0]
    if len(nz) <= 10:
        return ci, fi, nz, []
    Dc = np.ascontiguousarray(D[np.ix_(nz, nz)])
    rows = []
    for li, M in enumerate(Ms):
        r, tot = O.stripe_search(Dc, float(M), gw=_G['gw'])
        rows += [(li,) + tuple(int(v) for v in q) + (float(t),) for q, t in zip(r, tot)]
    return ci, fi, nz, rows


def _check_tables(unf, filt, names, sizes_bp, levels, pcut):
    u = pd.read_csv(io.StringIO(unf), sep='\t')
    f = pd.read_csv(io.StringIO(filt), sep='\t')
    assert list(u.columns) == ['chr', 'pos1', 'pos2', 'chr2', 'pos3', 'pos4', 'length', 'width', 'Mean', 'maxpixel', 'pvalue', 'Stripiness']
    size = dict(zip(names, sizes_bp))
    assert set(u['chr']) <= set(names) and (u['chr'] == u['chr2']).all()
    lim = u['chr'].map(size)
    assert (u['pos1'] >= 1).all() and (u['pos2'] <= lim).all() and (u['pos3'] >= 1).all() and (u['pos4'] <= lim).all()
    assert (u['pos1'] < u['pos2']).all() and (u['pos3'] < u['pos4']).all()
    assert (u['length'] == u['pos4'] - u['pos3'] + 1).all() and (u['width'] == u['pos2'] - u['pos1'] + 1).all()
    assert ((u['pvalue'] > 0) & (u['pvalue'] <= 1)).all() and np.isfinite(u['Mean']).all()
    # rows come level by level, chromosomes in genome order inside a level (stripenn.py:134-144)
    lev = {'%s%%' % (p * 100): i for i, p in enumerate(levels)}
    key = u['maxpixel'].map(lev).to_numpy() * 100 + u['chr'].map({n: i for i, n in enumerate(names)}).to_numpy()
    assert not np.isnan(key).any() and (np.diff(key) >= 0).all()
    # every stripe is anchored on the diagonal at one of its ends
    x1, x2, y1, y2 = (u['pos1'] - 1) // RESOL, u['pos2'] // RESOL, (u['pos3'] - 1) // RESOL, u['pos4'] // RESOL
    assert ((x1 == y1) | (x2 == y2)).all()
    assert (f['pvalue'] < pcut).all() and len(f) == int((u['pvalue'] < pcut).sum())
    s = f['Stripiness'].to_numpy()
    assert (np.diff(s[~np.isnan(s)]) <= 0).all()
    return u, f


def test_mm10_genome_compute_determinism_invariants_and_sampled_frames(tmp_path):
    from oracle import oracle as O
    from stripenn_amd import backend as BK, getStripe as GS, hip, pixels
    O.build()
    names = ['chr%d' % (i + 1) for i in range(19)] + ['chrX']
    levels = [0.95, 0.96, 0.97, 0.98, 0.99]
    chroms, table = _genome(MM10, 1, names)
    assert len(table.count) > 250_000_000
    a = _run_compute(table, str(tmp_path / 'a'), '0.95,0.96,0.97,0.98,0.99')
    b = _run_compute(table, str(tmp_path / 'b'), '0.95,0.96,0.97,0.98,0.99')
    assert a == b, 'two runs of the same genome differ'
    # every intermediate in the reference's f64 operations -- k_canny_pipe instead of k_canny_f32 AND k_gray<1> instead of the
    # certified k_gray_c3 -- on all 79 350 images: same TSVs
    os.environ['STP_CANNY'] = 'exact'
    os.environ['STP_GRAY'] = 'exact'
    try:
        c = _run_compute(table, str(tmp_path / 'c'), '0.95,0.96,0.97,0.98,0.99')
    finally:
        os.environ.pop('STP_CANNY', None)
        os.environ.pop('STP_GRAY', None)
    assert a == c, 'the certified kernels (k_canny_f32, k_gray_c3) and the exact ones (k_canny_pipe, k_gray<1>) give different tables'
    # round 6: the shipped kernels with every Canny tile computed (no use of the images' symmetry)
    os.environ['STP_SYM'] = '0'
    try:
        d = _run_compute(table, str(tmp_path / 'd'), '0.95,0.96,0.97,0.98,0.99')
    finally:
        os.environ.pop('STP_SYM', None)
    assert a == d, 'mirrored class words (k_canny_f32 with the images\' symmetry) and computed ones give different tables'
    u, f = _check_tables(a[0], a[1], names, MM10, levels, 0.1)
    assert len(u) > 30000 and len(f) > 2000 and set(u['chr']) == set(names)
    # >= 40 frames across all chromosomes against the oracle: the facade's own quantiles and searches
    hb = BK.HipBackend(0)
    sel = pixels.PixelSelector(table, True)
    sizes = np.asarray(MM10, dtype=np.int64)
    obj = GS.getStripe(sel, RESOL, 10, 8, 2.0, names, names, sizes, sizes, 8, 3, 123456789, backend=hb)

    class Info:
        chromsizes = pd.Series(sizes, index=names)
    MP = obj.getQuantile_original(Info, names, levels)
    rng = np.random.default_rng(11)
    tasks, got = [], {}
    _G.update(D={}, Ms={}, gw=hip.gauss_weights(2.0)[0])
    for ci, nm in enumerate(names):
        nb = chroms[nm].nbins
        nfr = -(-nb // 200)
        recs = obj._search(nm, ci, MP[nm])
        fr, starts, ends, f0 = obj._chrom_frames(nm, ci)
        _G['Ms'][ci] = MP[nm]
        for fi in sorted(set([0, nfr - 1] if ci % 5 == 0 else []) | set(rng.choice(nfr, 2, replace=False).tolist())):
            s, e = int(starts[fi]), int(ends[fi])
            _G['D'][(ci, fi)] = chroms[nm].host.block(s, e + 1, s, e + 1)
            mine = recs[recs['frame'] == fi]
            got[(ci, fi)] = (int(fr.S[fi]), fr.nz[fi].copy(),
                             [(int(r['level']), int(r['b_index']), int(r['ud']), int(r['x']), int(r['y']), int(r['w']), int(r['h']),
                               float(r['total'])) for r in mine])
            tasks.append((ci, fi))
    assert len(tasks) >= 40 and len({t[0] for t in tasks}) == 20
    cores = min(len(os.sched_getaffinity(0)), 32)
    with mp.get_context('fork').Pool(cores) as pool:
        res = pool.map(_oracle_frame, tasks, chunksize=1)
    nrows = 0
    for ci, fi, nz, rows in res:
        S, gnz, grows = got[(ci, fi)]
        assert S == (len(nz) if len(nz) > 10 else 0) and np.array_equal(gnz[:len(nz)], nz), (ci, fi)
        assert grows == rows, 'records of %s frame %d' % (names[ci], fi)
        nrows += len(rows)
    assert nrows > 1000
    hb.close()


def test_hg38_genome_compute_then_score_with_sampled_stripes(tmp_path):
    from oracle_backend import OracleBackend
    from stripenn_amd import backend as BK, getStripe as GS, io as sio, pixels, score as score_mod
    names = ['chr%d' % (i + 1) for i in range(22)] + ['chrX']
    levels = [0.97, 0.98, 0.99]
    chroms, table = _genome(HG38, 101, names)
    out = str(tmp_path / 'c')
    txt = _run_compute(table, out, '0.97,0.98,0.99')
    u, f = _check_tables(txt[0], txt[1], names, HG38, levels, 0.1)
    assert len(u) > 10000 and set(u['chr']) == set(names)
    # `stripenn score` on the called stripes: same seed and numcores -> same background -> the same columns
    orig = score_mod.open_matrix
    score_mod.open_matrix = lambda cool: sio.pixel_matrix(table)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            res = score_mod.getScore('pixels:in-memory', os.path.join(out, 'result_unfiltered.tsv'), 'weight', 8, 123456789,
                                     os.path.join(out, 'scores.tsv'))
    finally:
        score_mod.open_matrix = orig
    ref = pd.read_csv(os.path.join(out, 'result_unfiltered.tsv'), sep='\t', float_precision='round_trip')
    assert len(res) == len(ref)
    assert np.array_equal(res['pvalue_added'].to_numpy(), ref['pvalue'].to_numpy())
    assert np.array_equal(res['Stripiness_added'].to_numpy(), ref['Stripiness'].to_numpy(), equal_nan=True)
    # 200 sampled stripes of five chromosomes through oracle.py's per-stripe restatements (their inputs -- the four
    # background tables and the expected values -- come from the device: both are checked against the reference at
    # the golden and chr16 sizes)
    sizes = np.asarray(HG38, dtype=np.int64)
    sel = pixels.PixelSelector(table, True)
    hb = BK.HipBackend(0)
    objh = GS.getStripe(sel, RESOL, 10, 8, 2.5, names, names, sizes, sizes, 8, 1, 123456789, backend=hb)
    EV = objh.mpmean()
    bg = objh.nulldist()
    hb.close()
    pick = ['chr1', 'chr7', 'chr15', 'chr22', 'chrX']
    rng = np.random.default_rng(3)
    rows = np.concatenate([rng.choice(np.nonzero((ref['chr'] == c).to_numpy())[0], 40, replace=False) for c in pick])
    sample = ref.iloc[np.sort(rows)][['chr', 'pos1', 'pos2', 'chr2', 'pos3', 'pos4']].reset_index(drop=True)
    want = res.iloc[np.sort(rows)].reset_index(drop=True)
    sub = [names.index(c) for c in pick]
    objo = GS.getStripe(sel, RESOL, 10, 8, 2.5, names, pick, sizes, sizes[sub], 8, 1, 123456789, backend=OracleBackend())
    p = objo.pvalue(*bg, sample)
    g, moe, toe = objo.scoringstripes(sample, EV, '0')
    M, SUM = objo.getMean(sample)
    assert np.array_equal(np.asarray(p), want['pvalue_added'].to_numpy()), 'p-values differ from the oracle'
    assert np.array_equal(np.asarray(g), want['Stripiness_added'].to_numpy(), equal_nan=True), 'Stripiness differs from the oracle'
    assert np.allclose(np.asarray(M), want['O_Mean_added'].to_numpy(), rtol=1e-9, atol=0, equal_nan=True)
    assert np.allclose(np.asarray(moe), want['O/E_Mean_added'].to_numpy(), rtol=1e-9, atol=0, equal_nan=True)
