"""cProfile of `stripenn_amd.stripenn.compute` on the benchmark genome (mm10 sizes, in-memory pixel table), 4th run of the process:
where the host time of the end-to-end figure (`bench.py: e2e_compute`) goes.  Run on the GPU box from the repo root."""
import contextlib
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from stripenn_amd import backend as BK, io as sio, stripenn, synth_device

dev = torch.device('cuda', 0)
names = bench.MM10_NAMES
nbins = [-(-s // bench.RESOL) for s in bench.MM10]
chroms = {n: synth_device.DeviceChrom(nb, i + 1, dev) for i, (n, nb) in enumerate(zip(names, nbins))}
table = synth_device.pixel_table(names, chroms, bench.RESOL)
hb = BK.HipBackend(0)
stripenn.open_matrix = lambda cool: sio.pixel_matrix(table)
out = tempfile.mkdtemp(prefix='stp_e2e_')


def run():
    with contextlib.redirect_stdout(io.StringIO()):
        stripenn.compute('pixels:in-memory', out, 'weight', 'all', 2.0, 10, 8, ','.join(str(m) for m in bench.MAXPIXEL), 8, 0.1,
                         '0', False, 3, 123456789, force=True, backend=hb)


for _ in range(3):
    t0 = time.time(); run(); print('run %.3f s' % (time.time() - t0))
pr = cProfile.Profile(); pr.enable(); t0 = time.time(); run(); dt = time.time() - t0; pr.disable()
print('profiled run %.3f s' % dt)
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
st.sort_stats('tottime').print_stats(30)
