"""Host logic of the façade (frames, pools, PRNG order, region arithmetic, RemoveRedundant, table
assembly, TSV text) against the reference's end-to-end goldens, with the oracle as compute backend.
The same checks run with the HIP backend in test_gpu_e2e.py."""
import warnings

import numpy as np

import e2e_common as E
from oracle_backend import OracleBackend

warnings.filterwarnings('ignore')


def _mk(gw):
    return OracleBackend(gauss_w=gw)


def test_compute_pipeline_matches_reference_exactly():
    E.run_compute(_mk, float_exact=True)


def test_score_pipeline_matches_reference_exactly():
    E.run_score(_mk, float_exact=True)


def test_compute_and_score_at_1kb_match_reference_exactly():
    """resolution 1 kb: background size 50 -> wrapped Python slices in nulldist (the unit-matrix path),
    2500-element window means (numpy's recursive pairwise sum), 50-bin flanks."""
    E.run_compute(_mk, float_exact=True, tag='1kb')
    E.run_score(_mk, float_exact=True, tag='1kb')


def test_nan_flank_raises_indexerror_like_reference():
    E.run_nan_flank_indexerror(_mk)


def test_background_with_numcores_gt1():
    E.run_par_background(_mk)


def test_remove_redundant_semantics():
    import pandas as pd
    from stripenn_amd import getStripe as GS
    obj = GS.getStripe.__new__(GS.getStripe)
    obj.backend = OracleBackend()
    df = pd.DataFrame({'chr': ['a'] * 4, 'pos1': [1, 1, 500001, 1], 'pos2': [20000, 25000, 520000, 20000],
                       'pos3': [1, 1, 500001, 1], 'pos4': [400000, 300000, 900000, 400000], 'h': [80, 60, 80, 80],
                       'w': [4, 5, 4, 4], 'num': [0, 0, 2, 1], 'pvalue': [0.05, 0.01, 0.2, 0.05]})
    out = obj.RemoveRedundant(df, 'size')
    # rows 0/1 overlap: ratios 20 vs 12 -> row 1 dropped; rows 0/3 tie (<=) -> the first (row 0) dropped
    assert out.index.tolist() == [2, 3]
    out = obj.RemoveRedundant(df, 'pvalue')
    assert out.index.tolist() == [1, 2]
    import pytest
    with pytest.raises(ValueError):
        obj.RemoveRedundant(df, 'other')


def test_quantile_linear_equals_numpy():
    """Host interpolation around exact order statistics == np.quantile, bit for bit."""
    from stripenn_amd.getStripe import quantile_linear
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 10, 999, 100003):
        a = rng.random(n) * rng.integers(1, 1000, n)
        srt = np.sort(a)
        for q in ([0.95, 0.96, 0.97, 0.98, 0.99], [0.0, 1.0, 0.5], 0.25, [1e-9, 1 - 1e-12, 0.3333333333333333]):
            got = quantile_linear(lambda r: srt[np.asarray(r)], n, q)
            assert np.array_equal(np.asarray(got), np.asarray(np.quantile(a, q))), (n, q)
