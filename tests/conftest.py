import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_sessionstart(session):
    """A GPU test process holds two HIP runtimes: ROCm's (linked by libstripenn_hip.so) and the one bundled with
    torch (used for device-generated test data and torch.distributed).  torch's refuses to initialise after the
    other one has opened the device, so it goes first.  (device_count() does not touch the GPU; on the CPU-only
    build box it is 0 and nothing is initialised.)"""
    try:
        import torch
        if torch.cuda.device_count() > 0:
            torch.cuda.init()
    except Exception:      # noqa: BLE001 -- CPU-only box, or torch absent: the CPU suite does not need it
        pass


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with gpurun / by the driver at round end)')


@pytest.fixture(scope='session')
def golden_stages():
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'stages_chr7.npz'))


@pytest.fixture(scope='session')
def chr7(golden_stages):
    """The synthetic chromosome the stage goldens were generated from (regenerated from its seed)."""
    from stripenn_amd import synth
    g = golden_stages
    names, sizes, sel = synth.make_genome([int(g['chromsize'])], int(g['resol']), seed0=int(g['seed0']), names=['chr7'])
    return sel.chroms['chr7']


@pytest.fixture(scope='session')
def hip_ctx():
    from stripenn_amd import hip
    ctx = hip.Context(0)   # raises loudly if the extension or the GPU is missing
    yield ctx
    ctx.close()
