"""GPU parity of the StripeSearch chain: HIP kernels (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
HW = 512


@pytest.fixture(scope='module')
def gpu_chr7(hip_ctx, chr7, golden_stages):
    g = golden_stages
    band_host = chr7.band(HW)
    band = hip_ctx.band_upload(band_host)
    starts = [int(g['c%d_start' % i]) for i in range(int(g['ncases']))]
    ends = [int(g['c%d_end' % i]) for i in range(int(g['ncases']))]
    fr = band.frames(starts, ends)
    yield band, fr
    fr.close()
    band.close()


def _oracle_frame(chr7, start, end):
    D, nz = O.frame_dense(chr7.block, start, end)
    return np.ascontiguousarray(D[np.ix_(nz, nz)]), nz


def test_frame_compaction_matches_reference(gpu_chr7, golden_stages):
    g = golden_stages
    _, fr = gpu_chr7
    for ci in range(int(g['ncases'])):
        S = int(g['c%d_S' % ci])
        assert fr.S[ci] == S
        assert np.array_equal(fr.nz[ci, :S], g['c%d_nz' % ci])


@pytest.mark.parametrize('ci', range(6))
def test_stages_bit_exact(gpu_chr7, golden_stages, chr7, ci):
    g = golden_stages
    _, fr = gpu_chr7
    p = 'c%d_' % ci
    M = float(g[p + 'M'])
    gw = np.ascontiguousarray(g['gw_2p0'])
    D, nz = _oracle_frame(chr7, int(g[p + 'start']), int(g[p + 'end']))
    gp = O.gplane(D, M)
    for bi, b in enumerate(g['bvals']):
        got = fr.dbg_stages(ci, M, bi, gauss_w=gw)
        og = O.gray(gp, b)
        assert np.array_equal(got['gray'], og), 'gray differs (case %d b %d)' % (ci, bi)
        oe, dbg = O.canny(og, gw, 8, debug=True)
        assert np.array_equal(got['cls'], dbg['cls']), 'NMS classes differ'
        assert np.array_equal(got['edges'], oe), 'edges differ'
        # and against the reference's own output
        assert np.array_equal(np.packbits(got['edges'].astype(bool), axis=1), g[p + 'edges'][bi])
        ov = O.vertical_line(oe)
        assert np.array_equal(got['vert'], ov)
        assert np.array_equal(np.packbits(got['vert'].astype(bool), axis=1), g[p + 'vert'][bi])
        t, e, ud = O.columns(ov, 10)
        assert np.array_equal(got['col_t'], t) and np.array_equal(got['col_end'], e) and np.array_equal(got['col_ud'], ud)
        assert np.array_equal(np.stack([got['col_t'], got['col_end']], 1), g[p + 'block'][bi])
        tm1, _ = O.join_dbg(oe, ov, 1, 10, 8)
        tm2, _ = O.join_dbg(oe, ov, 2, 10, 8)
        assert np.array_equal(got['testmat1'], tm1) and np.array_equal(got['testmat2'], tm2)


def test_stripe_records_match_reference(gpu_chr7, golden_stages):
    """Each golden case's raw StripeSearch rows (x, y, h, w, total) straight from the reference."""
    g = golden_stages
    _, fr = gpu_chr7
    gw = np.ascontiguousarray(g['gw_2p0'])
    for ci in range(int(g['ncases'])):
        p = 'c%d_' % ci
        recs = fr.stripe_search([float(g[p + 'M'])], gauss_w=gw)
        mine = recs[recs['frame'] == ci]
        got = np.stack([mine['x'], mine['y'], mine['h'], mine['w']], axis=1).astype(np.int64).reshape(-1, 4)
        assert np.array_equal(got, g[p + 'rec_xywh']), 'records differ for case %d' % ci
        assert np.array_equal(mine['total'], g[p + 'rec_total'])


def test_multi_level_batch_matches_oracle(gpu_chr7, golden_stages, chr7):
    """All frames x 2 maxpixel levels in one batched call vs the oracle run frame by frame."""
    g = golden_stages
    _, fr = gpu_chr7
    gw = np.ascontiguousarray(g['gw_2p0'])
    levels = [float(g['MP'][0]), float(g['MP'][1])]
    recs = fr.stripe_search(levels, gauss_w=gw)
    exp = []
    for ci in range(int(g['ncases'])):
        D, nz = _oracle_frame(chr7, int(g['c%d_start' % ci]), int(g['c%d_end' % ci]))
        for li, M in enumerate(levels):
            r, tot = O.stripe_search(D, M, gw=gw)
            for k in range(len(r)):
                exp.append((ci, li) + tuple(int(v) for v in r[k]) + (tot[k],))
    got = [(int(r['frame']), int(r['level']), int(r['b_index']), int(r['ud']), int(r['x']), int(r['y']), int(r['w']),
            int(r['h']), float(r['total'])) for r in recs]
    assert len(got) == len(exp) and len(got) > 50
    assert got == exp


def test_plateaus_and_other_parameters(hip_ctx):
    """Block-structured contact matrices (exact ties in the gradient magnitude -> the certified NMS must
    take its exact fallback), plus canny sigma 2.5 / 3.0 (generic-radius kernel) and bfilter 1 / 5."""
    from stripenn_amd import hip
    n = 900
    rr, cc = np.mgrid[0:n, 0:n]
    dense = np.where((cc // 37 + rr // 53) % 2 == 0, 12.0, 3.0) + np.where(np.abs(cc - rr) < 25, 20.0, 0.0)
    dense = np.where(np.abs(cc - rr) <= 520, dense, 0.0)
    hw = 512
    band_h = np.zeros((n, 2 * hw))
    for i in range(n):
        lo, hi = max(0, i - hw), min(n, i + hw)
        band_h[i, lo - i + hw:hi - i + hw] = dense[i, lo:hi]
    band = hip_ctx.band_upload(band_h)
    fr = band.frames([0, 100, 500], [299, 499, 899])
    for sigma, bf in ((2.0, 3), (2.5, 3), (3.0, 3), (2.0, 1), (2.0, 5)):
        gw, gr = hip.gauss_weights(sigma)
        for f, (s, e) in enumerate(((0, 299), (100, 499), (500, 899))):
            D = np.ascontiguousarray(dense[s:e + 1, s:e + 1])
            for M in (10.0, 30.0):
                gp = O.gplane(D, M)
                for bi in (0, 5):
                    got = fr.dbg_stages(f, M, bi, sigma=sigma, bfilter=bf)
                    og = O.gray(gp, O.brightness_levels()[bi], bf)
                    assert np.array_equal(got['gray'], og)
                    oe, dbg = O.canny(og, gw, gr, debug=True)
                    assert np.array_equal(got['cls'], dbg['cls']), (sigma, bf, f, M, bi)
                    assert np.array_equal(got['edges'], oe)
                recs = fr.stripe_search([M], sigma=sigma, bfilter=bf)
                mine = recs[recs['frame'] == f]
                r, tot = O.stripe_search(D, M, sigma=sigma, bf=bf)
                assert [tuple(int(v) for v in q) for q in r] == [(int(a['b_index']), int(a['ud']), int(a['x']), int(a['y']), int(a['w']), int(a['h'])) for a in mine]
                assert np.array_equal(mine['total'], tot)
    fr.close(); band.close()


@pytest.mark.parametrize('kind', ['continuous', 'counts', 'constant', 'two_values', 'wide_range', 'small'])
def test_medpixel_order_statistics(hip_ctx, kind):
    """medpixel = np.quantile(D[D > 0], 0.5) (getStripe.py:885) on data that exercises every select path:
    continuous values (LDS gather), integer counts / constants (13-bit refinement to the last digit),
    values spanning the whole exponent range, and frames below the gather capacity."""
    rng = np.random.default_rng(5)
    n, hw = 900, 512
    if kind == 'continuous':
        A = rng.gamma(0.7, 3.0, (n, n))
    elif kind == 'counts':
        A = rng.poisson(1.3, (n, n)).astype(np.float64)
    elif kind == 'constant':
        A = np.full((n, n), 2.5)
    elif kind == 'two_values':
        A = rng.choice([0.0, 1.0, 1.0000000000000002], (n, n))
    elif kind == 'wide_range':
        A = np.exp(rng.uniform(-600, 600, (n, n)))
    else:
        A = rng.gamma(0.7, 3.0, (n, n)) * (rng.random((n, n)) < 0.004)
    A = np.triu(A) + np.triu(A, 1).T
    A[np.arange(n), np.arange(n)] += 1.0           # no empty columns
    i = np.arange(n)[:, None]; d = np.arange(-hw, hw)[None, :]     # band[i, d + hw] = A[i, i + d]
    j = i + d
    band_h = np.where((j >= 0) & (j < n) & (np.abs(d) <= 399), A[i, np.clip(j, 0, n - 1)], 0.0)
    band = hip_ctx.band_upload(np.ascontiguousarray(band_h))
    st = np.array([0, 100, 300, 500, 0], dtype=np.int32)
    en = np.array([299, 499, 699, 899, 20], dtype=np.int32)
    fr = band.frames(st, en)
    for f in range(len(st)):
        D = A[st[f]:en[f] + 1, st[f]:en[f] + 1]
        assert fr.medpixel[f] == np.quantile(D[D > 0], 0.5), (kind, f)
    fr.close(); band.close()


def test_dense_stripes_more_than_64_candidate_columns(hip_ctx):
    """Line joining with > 64 candidate columns per image (k_lines then runs the one-lane form of the
    grouping instead of the one-wave form) and hundreds of records per frame, against the oracle."""
    from stripenn_amd import synth
    ch = synth.SynthChrom(900, 21, stripe_every=6, stripe_gain=3.0)
    blk = ch.block(0, 900, 0, 900)
    Ms = np.quantile(blk[blk > 0], [0.9, 0.97])
    band = hip_ctx.band_upload(ch.band(HW))
    st = np.array([100, 300], dtype=np.int32); en = np.array([499, 699], dtype=np.int32)
    fr = band.frames(st, en)
    recs = fr.stripe_search(Ms)
    gw, gr = O.gauss_weights(2.0)
    exp, widest = [], 0
    for fi in range(2):
        D, nz = O.frame_dense(ch.block, int(st[fi]), int(en[fi]))
        D = np.ascontiguousarray(D[np.ix_(nz, nz)])
        for li, M in enumerate(Ms):
            r, tot = O.stripe_search(D, float(M))
            exp += [(fi, li) + tuple(int(v) for v in r[k]) + (float(tot[k]),) for k in range(len(r))]
            if fi == 0:
                g = O.gplane(D, float(M))
                for b in O.brightness_levels():
                    E = O.canny(O.gray(g, b, 3), gw, gr)
                    V = O.vertical_line(E)
                    for ud in (1, 2):
                        tm, _ = O.join_dbg(E, V, ud, 10, 8)
                        widest = max(widest, int((tm.sum(axis=0) >= 3).sum()))
    assert widest > 64, 'input no longer exercises the > 64 column path'
    got = [(int(r['frame']), int(r['level']), int(r['b_index']), int(r['ud']), int(r['x']), int(r['y']), int(r['w']),
            int(r['h']), float(r['total'])) for r in recs]
    assert len(got) > 500 and got == exp
    fr.close(); band.close()


@pytest.mark.parametrize('kind', ['nonneg_asym', 'negative', 'cancelling'])
def test_zero_column_rule_on_general_values(hip_ctx, kind):
    """Zero-column removal keeps column c when sum(axis 0) != 0 after NaN -> 0 (getStripe.py:809-821).  k_frame_prep takes
    the coalesced shortcut "some entry > 0" only for non-negative frames; a frame with a negative value goes through the
    exact column walk: columns whose entries cancel exactly, columns with only negative entries, NaN rows, and an
    asymmetric block (column occupancy != row occupancy) all give numpy's verdict."""
    rng = np.random.default_rng(11)
    n, hw = 700, 512
    A = rng.gamma(0.7, 3.0, (n, n)) * (rng.random((n, n)) < 0.3)
    A = np.triu(A) + np.triu(A, 1).T
    empty = rng.choice(n, 40, replace=False)
    A[empty, :] = 0.0; A[:, empty] = 0.0
    if kind == 'nonneg_asym':
        A[:, empty[:10]] = 0.0
        A[empty[10:20], :] = rng.gamma(1.0, 1.0, (10, n))           # rows filled, their columns still empty
        A[:, empty[10:20]] = 0.0
    elif kind == 'negative':
        A[5, empty[0]] = -2.0                                        # a column with one negative entry only: kept
        A[7, 33] = -1.0
    else:
        c = empty[1]
        A[10, c] = 1.5; A[20, c] = -1.5                              # cancels exactly: removed
        c2 = empty[2]
        A[10, c2] = 1.5; A[20, c2] = -1.0                            # does not cancel: kept
    A[100, :] = np.nan
    i = np.arange(n)[:, None]; d = np.arange(-hw, hw)[None, :]
    j = i + d
    band_h = np.where((j >= 0) & (j < n) & (np.abs(d) <= 399), A[i, np.clip(j, 0, n - 1)], 0.0)
    band = hip_ctx.band_upload(np.ascontiguousarray(band_h))
    st = np.array([0, 100, 300], dtype=np.int32); en = st + 399
    fr = band.frames(st, en)
    for f in range(len(st)):
        D = np.nan_to_num(A[st[f]:en[f] + 1, st[f]:en[f] + 1])
        nz = np.where(D.sum(axis=0) != 0)[0]
        assert fr.S[f] == len(nz), (kind, f)
        assert np.array_equal(fr.nz[f, :len(nz)], nz), (kind, f)
        Dp = D[D > 0]
        assert fr.medpixel[f] == np.quantile(Dp, 0.5), (kind, f)
    fr.close(); band.close()
