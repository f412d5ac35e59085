"""End-to-end parity on the MI355X: the façade with the HIP backend vs the reference's goldens.
Positions/lengths/widths bit-exact; Mean / pvalue / Stripiness within 1e-4 relative (north star) --
and, because the kernels keep numpy's summation order, expected to be exact as well."""
import warnings

import numpy as np
import pytest

import e2e_common as E

pytestmark = pytest.mark.gpu
warnings.filterwarnings('ignore')


class _HipWithWeights:
    """HipBackend whose Gaussian weights are the golden ones (the reference's numpy 1.26 exp)."""

    def __init__(self, gw):
        from stripenn_amd.backend import HipBackend
        self.b = HipBackend(0)
        self.gw = gw

    def __getattr__(self, k):
        return getattr(self.b, k)

    def stripe_search(self, frames, M_levels, sigma, minH, maxW, bfilter):
        return frames.stripe_search(M_levels, sigma=sigma, minH=minH, maxW=maxW, bfilter=bfilter, gauss_w=self.gw)


def test_compute_pipeline_on_gpu():
    obj, out = E.run_compute(_HipWithWeights, float_exact=False)
    obj.backend.close()


def test_compute_pipeline_on_gpu_is_bit_exact():
    obj, out = E.run_compute(_HipWithWeights, float_exact=True)
    obj.backend.close()


def test_score_pipeline_on_gpu():
    E.run_score(_HipWithWeights, float_exact=False)


def test_compute_and_score_at_1kb_on_gpu():
    """1 kb bins (background size 50): k_null_windows<true>, unit-matrix path for wrapped slices."""
    obj, out = E.run_compute(_HipWithWeights, float_exact=True, tag='1kb')
    obj.backend.close()
    E.run_score(_HipWithWeights, float_exact=True, tag='1kb')


def test_nan_flank_raises_indexerror_on_gpu():
    obj = E.run_nan_flank_indexerror(_HipWithWeights)
    obj.backend.close()


def test_background_with_numcores_gt1_on_gpu():
    E.run_par_background(_HipWithWeights)


def test_gpu_quantile_matches_numpy():
    """stp_select_*: exact order statistics of the positive entries over several appended chunks."""
    from stripenn_amd.backend import HipBackend
    from stripenn_amd.getStripe import quantile_linear
    hb = HipBackend(0)
    rng = np.random.default_rng(5)
    chunks = [rng.random(n) * rng.integers(0, 50, n) * (rng.random(n) > 0.3) for n in (1000003, 17, 250000)]
    chunks[1][:3] = [np.nan, -1.0, 0.0]
    allv = np.concatenate(chunks)
    pos = allv[allv > 0]
    sel = hb.select_open()
    for c in chunks:
        hb.select_append(sel, c)
    assert hb.select_count(sel) == len(pos)
    q = [0.95, 0.96, 0.97, 0.98, 0.99, 0.0, 1.0, 0.5]
    got = quantile_linear(lambda r: hb.select_ranks(sel, r), len(pos), q)
    assert np.array_equal(got, np.quantile(pos, q))
    srt = np.sort(pos)
    ranks = np.array([0, 1, len(pos) // 2, len(pos) - 1])
    assert np.array_equal(hb.select_ranks(sel, ranks), srt[ranks])
    hb.select_close(sel)
    hb.close()


def test_gpu_remove_redundant_matches_reference_loops():
    """k_remove_redundant vs the reference's pair loops on random overlapping stripe tables."""
    import pandas as pd
    from stripenn_amd import getStripe as GS
    from stripenn_amd.backend import HipBackend
    from oracle_backend import OracleBackend
    rng = np.random.default_rng(9)
    n = 3000
    num = np.sort(rng.integers(0, 60, n))
    x0 = num * 200 * 5000 + rng.integers(0, 300, n) * 5000 + 1
    wbin = rng.integers(2, 9, n); hbin = rng.integers(11, 200, n)
    df = pd.DataFrame({'chr': np.where(rng.random(n) < 0.5, 'chrA', 'chrB'), 'pos1': x0, 'pos2': x0 + wbin * 5000 - 1,
                       'pos3': x0, 'pos4': x0 + hbin * 5000 - 1, 'h': hbin, 'w': wbin, 'num': num,
                       'pvalue': np.round(rng.random(n), 2), 'Stripiness': np.round(rng.random(n), 1)})
    g = GS.getStripe.__new__(GS.getStripe); g.backend = HipBackend(0)
    o = GS.getStripe.__new__(GS.getStripe); o.backend = OracleBackend()
    for by in ('size', 'pvalue', 'score'):
        a = g.RemoveRedundant(df, by); b = o.RemoveRedundant(df, by)
        assert a.index.tolist() == b.index.tolist() and 0 < len(a) < n, by
        a = g._filter_redundant(df, by, True); b = o._filter_redundant(df, by, True)
        assert a.index.tolist() == b.index.tolist()
    g.backend.close()
