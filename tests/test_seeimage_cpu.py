"""`seeimage` (reference seeimage.py:32-97): saturation quantile through the backend's exact select, the
reference's colour arithmetic, one PNG per level."""
import os

import numpy as np

from oracle import oracle as O
from oracle_backend import OracleBackend
from stripenn_amd import pixels, seeimage, synth


def test_window_image_and_files(tmp_path):
    names = ['chrA', 'chrB']
    chroms = {'chrA': synth.SynthChrom(900, 61), 'chrB': synth.SynthChrom(500, 62)}
    t = pixels.PixelTable.from_synth(names, chroms, 5000)
    p = str(tmp_path / 't.npz'); t.save(p)
    pos = 'chrA:1000001-2000000'
    out = str(tmp_path / 'heat')
    files = seeimage.seeimage('pixels:' + p, pos, '0.95,0.99', 'weight', out, False, 1, backend=OracleBackend())
    assert [os.path.basename(f) for f in files] == ['heat_%s_0.95qt.png' % pos, 'heat_%s_0.99qt.png' % pos]
    assert all(os.path.getsize(f) > 1000 for f in files)
    # the colour planes are StripeSearch's image build: red 1, green = blue = clip((255 (M - A) / M) / 255)
    sel = pixels.PixelSelector(t, 'weight')
    D = sel.fetch('chrA')
    M = np.quantile(D[D > 0], 0.95)
    A = sel.fetch(pos, pos)
    img = O.window_rgb(A, M)
    assert img.shape == (200, 200, 3) and np.all(img[..., 0] == 1.0)
    ok = ~np.isnan(A)
    exp = np.clip(np.where(255 * (M - A) / M < 0, 0, 255 * (M - A) / M) / 255, 0, 1)
    assert np.array_equal(img[..., 1][ok], exp[ok]) and np.array_equal(img[..., 1], img[..., 2], equal_nan=True)
