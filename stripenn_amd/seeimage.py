"""`stripenn seeimage`: heat map of one genomic window at the given saturation quantiles
(reference: seeimage.py:32-97).  The only heavy step -- the whole-chromosome quantile that sets the
saturation level -- runs through the backend's exact GPU select like `compute` does; the window itself is a
few hundred bins, coloured with the same arithmetic as StripeSearch's image build (getStripe.py:889-895:
red = 1, green = blue = clip((255 * (M - A) / M) / 255, 0, 1))."""
import sys

import numpy as np

from . import getStripe
from .io import open_matrix
from .stripenn import resolve_norm


def window_rgb(A, M):
    """The float RGB image the reference hands to imshow (values in [0, 1], NaN where A is NaN)."""
    A = np.asarray(A, dtype=np.float64)
    with np.errstate(invalid='ignore', divide='ignore'):
        blue = 255 * (M - A) / M
        blue[np.where(blue < 0)] = 0
        plane = blue / 255
    img = np.stack([np.ones_like(plane), plane, plane], axis=-1)
    return np.clip(img, a_min=0, a_max=1)


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
    obj = getStripe.getStripe(sel, resol, 10, 8, 2.5, all_names, [chrom], all_sizes, sizes[chrom], 2, 3, seed,
                              backend=backend, device=device)
    try:
        MP = (obj.getQuantile_slow if slow else obj.getQuantile_original)(Lib, [chrom], levels)
    finally:
        if backend is None:
            obj.backend.close()
    A = sel.fetch(position, position)
    written = []
    for i, q in enumerate(levels):
        img = window_rgb(A, MP[chrom][i])
        ax = plt.subplot(111)
        plt.imshow(img)
        plt.title(position)
        path = out + '_' + position + '_' + str(q) + 'qt' + '.png'
        ax.figure.savefig(path)
        written.append(path)
    plt.close('all')
    return written
