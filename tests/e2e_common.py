"""Shared end-to-end check: the façade (stripenn_amd.getStripe) driven in the call order of the
reference's stripenn.compute (stripenn.py:120-159) and score.getScore (score.py:49-60), compared
with tables produced by the unmodified reference (tests/golden/e2e_seq.npz)."""
import hashlib
import io
import os

import numpy as np
import pandas as pd

from stripenn_amd import synth, getStripe as GS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INTS = ['pos1', 'pos2', 'pos3', 'pos4', 'length', 'width', 'num', 'start', 'end', 'x', 'y', 'h', 'w']
REL_TOL = 1e-4     # north star: Mean / pvalue / Stripiness within 1e-4 relative


def sha(a):
    """sha256 of the array bytes with every NaN replaced by the canonical quiet NaN (0/0 gives -NaN on
    x86 and +NaN on the GPU; the payload/sign of a NaN is not part of the contract)."""
    a = np.asarray(a)
    if a.dtype.kind == 'f':
        a = np.where(np.isnan(a), np.float64('nan').astype(a.dtype), a)
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(tag='seq'):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'e2e_%s.npz' % tag))
    st = np.load(os.path.join(ROOT, 'tests', 'golden', 'stages_chr7.npz'))
    names = [str(x) for x in g['names']]
    sizes = g['sizes']
    nan_frac = float(g['nan_frac']) if 'nan_frac' in g.files else 0.005
    _, _, sel = synth.make_genome(list(sizes), int(g['resol']), seed0=int(g['seed0']), names=names, nan_frac=nan_frac)
    return g, st, names, sizes, sel


def close(a, b, exact):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if exact:
        return np.array_equal(a, b, equal_nan=True)
    return np.allclose(a, b, rtol=REL_TOL, atol=0, equal_nan=True)


def check_table(g, prefix, df, float_exact):
    assert int(g[prefix + 'n']) == len(df), (prefix, int(g[prefix + 'n']), len(df))
    for c in INTS:      # positions / sizes: always bit-exact
        assert np.array_equal(df[c].to_numpy(dtype=np.int64), g[prefix + c]), (prefix, c)
    assert [str(v) for v in df['chr']] == [str(v) for v in g[prefix + 'chr']]
    assert [str(v) for v in df['maxpixel']] == [str(v) for v in g[prefix + 'maxpixel']]
    for c in ('total', 'Mean', 'medpixel', 'pvalue'):
        assert close(df[c], g[prefix + c], float_exact), (prefix, c)


class Info:
    pass


def run_compute(make_backend, float_exact, core=1, tag='seq'):
    """Returns nothing; asserts every intermediate and the final TSVs against the reference."""
    g, st, names, sizes, sel = load(tag)
    resol = int(g['resol'])
    obj = GS.getStripe(sel, resol, 10, 8, 2.0, names, names, sizes, sizes, core, 3, int(g['prng_seed']),
                       backend=make_backend(np.ascontiguousarray(st['gw_2p0'])))
    info = Info(); info.chromsizes = pd.Series(sizes, index=names)
    MP = obj.getQuantile_original(info, names, list(g['maxpixel']))
    for n in names:
        assert np.array_equal(MP[n], g['MP_' + n])
    EV = obj.mpmean()
    for n in names:
        assert np.array_equal(np.array(EV[n]), g['EV_' + n]), 'expected values differ'
    bg = obj.nulldist()
    for t, k in zip(bg, ('lu', 'ru', 'ld', 'rd')):
        assert tuple(g['bg_%s_shape' % k]) == t.shape
        assert sha(t) == str(g['bg_%s_sha' % k]), 'background table %s differs from the reference' % k
    rt = pd.DataFrame(columns=GS.EXTRACT_COLUMNS + ['pvalue'])
    for i, perc in enumerate(g['maxpixel']):
        res = obj.extract(MP, i, float(perc), *bg)
        check_table(g, 'ex%d_' % i, res, float_exact)
        rt = pd.concat([rt, res])
    rt = obj.RemoveRedundant(df=rt, by='pvalue')
    check_table(g, 'rr_', rt, float_exact)
    s = obj.scoringstripes(rt, EV, '0')
    assert close(s[0], g['rr_g'], float_exact), 'Stripiness differs'
    assert close(s[1], g['rr_oe_mean'], False) and close(s[2], g['rr_oe_total'], False)
    out = rt.drop(columns=['total', 'num', 'start', 'end', 'x', 'y', 'h', 'w', 'medpixel'])
    out.insert(out.shape[1], 'Stripiness', s[0], True)
    filt = out[out['pvalue'] < 0.1].sort_values(by=['Stripiness'], ascending=False)
    b1, b2 = io.StringIO(), io.StringIO()
    out.to_csv(b1, sep='\t', header=True, index=False)
    filt.to_csv(b2, sep='\t', header=True, index=False)
    ref1 = pd.read_csv(io.StringIO(str(g['tsv_unfiltered'])), sep='\t')
    ref2 = pd.read_csv(io.StringIO(str(g['tsv_filtered'])), sep='\t')
    got1 = pd.read_csv(io.StringIO(b1.getvalue()), sep='\t')
    got2 = pd.read_csv(io.StringIO(b2.getvalue()), sep='\t')
    for ref, got in ((ref1, got1), (ref2, got2)):
        assert list(ref.columns) == list(got.columns)
        for c in ('chr', 'pos1', 'pos2', 'chr2', 'pos3', 'pos4', 'length', 'width', 'maxpixel'):
            assert ref[c].tolist() == got[c].tolist(), c       # filtered stripe calls identical
        for c in ('Mean', 'pvalue', 'Stripiness'):
            assert close(got[c], ref[c], float_exact), c
    if float_exact:
        assert b1.getvalue() == str(g['tsv_unfiltered']) and b2.getvalue() == str(g['tsv_filtered'])
    mask = str(g['mask'])
    if mask:
        sm = obj.scoringstripes(rt, EV, mask)
        assert close(sm[0], g['rr_g_masked'], float_exact)
    return obj, out


def run_score(make_backend, float_exact, tag='seq'):
    g, st, names, sizes, sel = load(tag)
    resol = int(g['resol'])
    ref = pd.read_csv(io.StringIO(str(g['tsv_unfiltered'])), sep='\t')
    table = ref[['chr', 'pos1', 'pos2', 'chr2', 'pos3', 'pos4']].reset_index(drop=True)
    obj2 = GS.getStripe(sel, resol, 10, 8, 2.5, names, names, sizes, sizes, 1, 1, int(g['prng_seed']),
                        backend=make_backend(np.ascontiguousarray(st['gw_2p5'])))
    EV2 = obj2.mpmean()
    bg2 = obj2.nulldist()
    pval = obj2.pvalue(*bg2, table)
    MEAN, SUM = obj2.getMean(table)
    s2, MOE, TOE = obj2.scoringstripes(table, EV2, '0')
    assert close(pval, g['sc_pvalue'], float_exact)
    assert close(MEAN, g['sc_mean'], False) and close(SUM, g['sc_sum'], False)
    assert close(s2, g['sc_g'], float_exact)
    assert close(MOE, g['sc_oe_mean'], False) and close(TOE, g['sc_oe_total'], False)


def run_par_background(make_backend):
    """numcores > 1: every chromosome restarts the PRNG from the seed (loky pickles self)."""
    g, st, names, sizes, sel = load('par')
    obj = GS.getStripe(sel, int(g['resol']), 10, 8, 2.0, names, names, sizes, sizes, int(g['core']), 3,
                       int(g['prng_seed']), backend=make_backend(None))
    bg = obj.nulldist()
    for t, k in zip(bg, ('lu', 'ru', 'ld', 'rd')):
        assert sha(t) == str(g['bg_%s_sha' % k]), 'background table %s differs (numcores>1 PRNG rule)' % k


def run_nan_flank_indexerror(make_backend):
    """1 kb bins, a NaN bin inside the 50-bin left flank of a 21-row stripe: the reference's
    np.delete(center, rowdel, axis=0) raises IndexError (getStripe.py:735); so must we."""
    import pytest
    resol = 1000
    names, sizes, sel = synth.make_genome([1500 * resol - 5], resol, seed0=3, names=['chr1'], nan_frac=0.02)
    nanb = [int(b) for b in sel.chroms['chr1'].nan_bins if 100 < b < 1300]
    assert nanb
    b = nanb[0]
    x0 = b + 40                                # NaN bin sits at flank column 10 .. inside [x0-50, x0)
    table = pd.DataFrame({'chr': ['chr1'], 'pos1': [x0 * resol + 1], 'pos2': [(x0 + 3) * resol], 'chr2': ['chr1'],
                          'pos3': [x0 * resol + 1], 'pos4': [(x0 + 5) * resol]})     # 5 rows < flank index 10
    obj = GS.getStripe(sel, resol, 10, 8, 2.0, names, names, sizes, sizes, 1, 3, 1, backend=make_backend(None))
    EV = {'chr1': [1.0] * 400}
    with pytest.raises(IndexError):
        obj.scoringstripes(table, EV, '0')
    return obj


def chr16_source(g):
    """The chr16-size synthetic chromosome of tests/golden/e2e_chr16.npz as cooler's pixel table (the reference read
    the same matrix through dense `fetch` calls; the table holds its raw counts and the weight column)."""
    from stripenn_amd import pixels
    names = [str(x) for x in g['names']]
    sizes = g['sizes']
    _, _, dense = synth.make_genome(list(sizes), int(g['resol']), seed0=int(g['seed0']), names=names,
                                    nan_frac=float(g['nan_frac']))
    table = pixels.PixelTable.from_synth(names, dense.chroms, int(g['resol']))
    table.chromsizes = np.asarray(sizes, dtype=np.int64)           # the true bp sizes (the last bin is partial)
    return names, sizes, pixels.PixelSelector(table, balance=True)


def run_chr16(make_backend, float_exact, configs=(1, 0)):
    """BASELINE.json configs[1] (maxpixel 0.95-0.99) and configs[0] (0.99 only) on the chr16-size chromosome against
    the tables the UNMODIFIED reference produced at that size with numcores = 8 (oracle/refharness/gen_golden_e2e.py
    run_chr16): quantiles, expected values, background tables, every extract table row by row, the redundancy filter,
    Stripiness and both TSVs."""
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'e2e_chr16.npz'))
    st = np.load(os.path.join(ROOT, 'tests', 'golden', 'stages_chr7.npz'))
    names, sizes, sel = chr16_source(g)
    obj = GS.getStripe(sel, int(g['resol']), 10, 8, 2.0, names, names, sizes, sizes, int(g['core']), 3,
                       int(g['prng_seed']), backend=make_backend(np.ascontiguousarray(st['gw_2p0'])))
    info = Info(); info.chromsizes = pd.Series(sizes, index=names)
    MP = obj.getQuantile_original(info, names, list(g['maxpixel']))
    assert np.array_equal(MP['chr16'], g['MP_chr16']), 'maxpixel quantiles differ'
    EV = obj.mpmean()
    assert np.array_equal(np.array(EV['chr16']), g['EV_chr16']), 'expected values differ'
    bg = obj.nulldist()
    for t, k in zip(bg, ('lu', 'ru', 'ld', 'rd')):
        assert tuple(g['bg_%s_shape' % k]) == t.shape
        assert sha(t) == str(g['bg_%s_sha' % k]), 'background table %s differs from the reference' % k
    tables = []
    for i, perc in enumerate(g['maxpixel']):
        if 1 not in configs and i != 4:
            tables.append(None)
            continue
        res = obj.extract(MP, i, float(perc), *bg)
        check_table(g, 'ex%d_' % i, res, float_exact)
        tables.append(res)
    rows = {}
    for cfg in configs:
        prefix, parts = ('', tables) if cfg == 1 else ('c0_', tables[4:])
        rt = pd.DataFrame(columns=GS.EXTRACT_COLUMNS + ['pvalue'])
        for r in parts:
            rt = pd.concat([rt, r])
        rt = obj.RemoveRedundant(df=rt, by='pvalue')
        check_table(g, prefix + 'rr_', rt, float_exact)
        s = obj.scoringstripes(rt, EV, '0')
        assert close(s[0], g[prefix + 'rr_g'], float_exact), 'Stripiness differs'
        out = rt.drop(columns=['total', 'num', 'start', 'end', 'x', 'y', 'h', 'w', 'medpixel'])
        out.insert(out.shape[1], 'Stripiness', s[0], True)
        filt = out[out['pvalue'] < 0.1].sort_values(by=['Stripiness'], ascending=False)
        b1, b2 = io.StringIO(), io.StringIO()
        out.to_csv(b1, sep='\t', header=True, index=False)
        filt.to_csv(b2, sep='\t', header=True, index=False)
        for ref_txt, got_txt in ((str(g[prefix + 'tsv_unfiltered']), b1.getvalue()), (str(g[prefix + 'tsv_filtered']), b2.getvalue())):
            ref = pd.read_csv(io.StringIO(ref_txt), sep='\t')
            got = pd.read_csv(io.StringIO(got_txt), sep='\t')
            assert list(ref.columns) == list(got.columns)
            for c in ('chr', 'pos1', 'pos2', 'chr2', 'pos3', 'pos4', 'length', 'width', 'maxpixel'):
                assert ref[c].tolist() == got[c].tolist(), c       # filtered stripe calls identical
            for c in ('Mean', 'pvalue', 'Stripiness'):
                assert close(got[c], ref[c], float_exact), c
            if float_exact:
                assert got_txt == ref_txt
        rows[cfg] = (len(out), len(filt))
    return obj, rows
