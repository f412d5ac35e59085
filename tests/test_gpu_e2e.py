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


def test_background_with_numcores_gt1_on_gpu():
    E.run_par_background(_HipWithWeights)
