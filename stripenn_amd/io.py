"""Opening the contact matrix: cooler when importable (host I/O stays cooler's, as in the reference),
a synthetic genome for tests / benchmarks ("synth:" URIs), or cooler's tables as plain arrays
("pixels:" URIs, stripenn_amd.pixels) -- from which the device packs the band without dense fetches.

    synth:chr1=7000000,chr2=4500000;resol=5000;seed=31
    pixels:/path/table.npz            (PixelTable.save)
    pixels:/path/file.mcool::resolutions/5000   (h5py when importable, else stripenn_amd.h5lite)
"""
import numpy as np
import pandas as pd

from . import synth


class MatrixInfo:
    """The few attributes of cooler.Cooler the reference's drivers read (stripenn.py:80-118)."""

    def __init__(self, chromnames, chromsizes, binsize, norm_columns, selector_factory):
        self.chromnames = list(chromnames)
        self.chromsizes = pd.Series(np.asarray(chromsizes, dtype=np.int64), index=self.chromnames)
        self.binsize = int(binsize)
        self._info = {'bin-size': int(binsize)}
        self._cols = norm_columns
        self._sel = selector_factory

    def bins(self):
        return pd.DataFrame(columns=self._cols)

    def matrix(self, balance=True):
        return self._sel(balance)


def pixel_matrix(table):
    """MatrixInfo over a PixelTable: `matrix(balance=...)` returns a PixelSelector."""
    from . import pixels
    cols = ['chrom', 'start', 'end'] + list(table.weights)
    return MatrixInfo(table.chromnames, table.chromsizes, table.binsize, cols,
                      lambda balance: pixels.PixelSelector(table, balance))


def open_matrix(cool):
    if str(cool).startswith('synth:'):
        spec = str(cool)[len('synth:'):]
        parts = spec.split(';')
        chroms = [kv.split('=') for kv in parts[0].split(',')]
        opts = dict(kv.split('=') for kv in parts[1:] if kv)
        resol = int(opts.get('resol', 5000))
        seed = int(opts.get('seed', 1))
        names = [c[0] for c in chroms]
        sizes = [int(c[1]) for c in chroms]
        _, _, sel = synth.make_genome(sizes, resol, seed0=seed, names=names)
        return MatrixInfo(names, sizes, resol, ['chrom', 'start', 'end', 'weight', 'KR', 'VC', 'VC_SQRT'],
                          lambda balance: sel)
    if str(cool).startswith('pixels:'):
        from . import pixels
        spec = str(cool)[len('pixels:'):]
        if spec.endswith('.npz'):
            table = pixels.PixelTable.load(spec)
        else:
            path, _, group = spec.partition('::')
            table = pixels.CoolTable(path, group.lstrip('/') or None)          # lazy: pixel columns stay in the file
        return pixel_matrix(table)
    try:
        import cooler
    except ImportError:
        # no cooler: read the file's tables directly (cooler URI "path::group") -- through h5py when it is there, else
        # through the package's own reader of the HDF5 subset cooler files use (stripenn_amd/h5lite.py)
        from . import pixels
        path, _, group = str(cool).partition('::')
        return pixel_matrix(pixels.CoolTable(path, group.lstrip('/') or None))      # lazy: pixel columns stay in the file
    return cooler.Cooler(cool)
