"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/*.h declares
(no compute calls here); creating a context without a GPU must fail loudly."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    from stripenn_amd import hip
    L = hip.load()
    text = open(os.path.join(ROOT, 'include', 'stripenn_hip.h')).read()
    declared = set(re.findall(r'\b(stp_[a-z_0-9]+)\s*\(', text))
    assert len(declared) >= 20
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, 'declared in the header but not exported: %s' % missing
    assert L.stp_version() == 3


def test_no_cpu_fallback_without_gpu():
    import os
    import torch
    if os.path.exists('/dev/kfd') or torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from stripenn_amd import hip
    with pytest.raises(hip.StripennHipError):
        hip.Context(0)
    from stripenn_amd import getStripe as GS
    with pytest.raises(hip.StripennHipError):
        GS.getStripe(None, 5000, 10, 8, 2.0, ['chr1'], ['chr1'], [10 ** 7], [10 ** 7], 1, 3, 1)


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, 'stripenn_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dp, f)).read()
                assert 'oracle' not in src.replace('oracle backend', '').replace('the oracle', '').replace('oracle)', ''), f


def test_product_gauss_weights_are_the_references(golden_stages):
    """hip.gauss_weights() -- what the product hands to the kernels -- equals scipy 1.7.1's _gaussian_kernel1d
    tables stored with the goldens (numpy 1.26.4) for the reference's two sigmas, whatever numpy runs here
    (numpy 2.2's exp moves four taps of each kernel by 1-2 ulp; the two kernels are pinned constants)."""
    import numpy as np
    from stripenn_amd import hip
    for sigma, key, r in ((2.0, 'gw_2p0', 8), (2.5, 'gw_2p5', 10)):
        w, rad = hip.gauss_weights(sigma)
        assert rad == r and w.dtype == np.float64 and np.array_equal(w, golden_stages[key])
    w, rad = hip.gauss_weights(3.0)                     # other sigmas: scipy's recipe with the numpy in use
    assert rad == 12 and len(w) == 25 and abs(w.sum() - 1.0) < 1e-15 and np.array_equal(w, w[::-1])
