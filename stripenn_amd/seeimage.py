"""`stripenn seeimage`: heat map of one genomic window at the given saturation quantiles
(reference: seeimage.py:32-97).  Both steps run on the device: the whole-chromosome quantile that sets the
saturation level goes through the backend's exact select like `compute` does, and the window is coloured from the
resident band by the image-build arithmetic of StripeSearch (stp_window_plane: getStripe.py:889-895, red = 1,
green = blue = clip((255 * (M - A) / M) / 255, 0, 1)); matplotlib only draws the array.  (The numpy restatement of the reference's
lines 78-85 that checks the kernel lives with the test infrastructure, not in the product.)"""
import sys

import numpy as np

from . import getStripe
from .io import open_matrix
from .stripenn import resolve_norm


def seeimage(cool, position, maxpixel, norm, out, slow, seed, backend=None, device=0):
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    Lib = open_matrix(cool)
    levels = list(map(float, str(maxpixel).split(',')))
    norm = resolve_norm(Lib, norm, weight_is_true=False)
    chrom = position.split(':')[0]
    names = list(Lib.chromnames)
    if chrom not in names:
        sys.exit('Invalid chromosome name.')
    sizes = Lib.chromsizes
    keep = np.where(np.asarray(sizes) > 500000)[0]
    all_names = [names[i] for i in keep]
    all_sizes = sizes.iloc[keep] if hasattr(sizes, 'iloc') else np.asarray(sizes)[keep]
    if len(all_names) == 0:
        sys.exit('Exit: All chromosomes are shorter than 50kb.')
    sel = Lib.matrix(balance=norm)
    resol = Lib.binsize
    # cooler's extent of the window; the band must hold every pixel of it
    rng = position.split(':')[1].replace(',', '')
    w0, w1 = getStripe._extent(int(rng.split('-')[0]), int(rng.split('-')[1]), resol)
    hw = max(getStripe.HALFWIDTH, -(-(w1 - w0 + 1) // 64) * 64)
    obj = getStripe.getStripe(sel, resol, 10, 8, 2.5, all_names, [chrom], all_sizes, sizes[chrom], 2, 3, seed,
                              backend=backend, device=device, halfwidth=hw)
    written = []
    try:
        MP = (obj.getQuantile_slow if slow else obj.getQuantile_original)(Lib, [chrom], levels)
        band = obj._band(chrom)
        planes = [obj.backend.window_plane(band, w0, w1 - w0, w0, w1 - w0, MP[chrom][i]) for i in range(len(levels))]
    finally:
        if backend is None:
            obj.backend.close()
    for i, q in enumerate(levels):
        img = np.stack([np.ones_like(planes[i]), planes[i], planes[i]], axis=-1)
        ax = plt.subplot(111)
        plt.imshow(img)
        plt.title(position)
        path = out + '_' + position + '_' + str(q) + 'qt' + '.png'
        ax.figure.savefig(path)
        written.append(path)
    plt.close('all')
    return written
