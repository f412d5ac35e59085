"""Synthetic cooler-schema tables for the real-file tests: the in-memory PixelTable (`table`) and a writer that lays it
out the way cooler does (`write`: an .mcool resolution group with `weight`, a divisive `KR` column, trans pixels,
`indexes/*`, the enumerated `bins/chrom`, variable-length string attributes, gzip + shuffle pixel columns).  `write`
needs h5py (tests/golden/make_cool_fixture.py runs it once for the committed file; tests/test_cool_reader.py per test)."""
import numpy as np

from stripenn_amd import pixels, synth

RESOL = 5000
GROUP = 'resolutions/5000'


def table(float_counts=False, small=False):
    names = ['chrA', 'chrB'] if small else ['chrA', 'chrB', 'chrC']
    nbins = (620, 500) if small else (900, 700, 450)
    chroms = {n: synth.SynthChrom(nb, 51 + k, stripe_every=90, stripe_gain=3.0) for k, (n, nb) in enumerate(zip(names, nbins))}
    t = pixels.PixelTable.from_synth(names, chroms, RESOL)
    # trans pixels (cooler stores them in the same table, sorted by (bin1, bin2)): a few per bin of the leading chromosomes
    rng = np.random.default_rng(5)
    off = t.chrom_offset
    if small:
        tb1 = rng.integers(off[0], off[1], 3000)
        tb2 = rng.integers(off[1], off[2], 3000)
    else:
        tb1 = np.concatenate([rng.integers(off[0], off[1], 4000), rng.integers(off[1], off[2], 2500)])
        tb2 = np.concatenate([rng.integers(off[1], off[3], 4000), rng.integers(off[2], off[3], 2500)])
    key = np.unique(tb1 * (1 << 32) + tb2)
    tb1, tb2 = key >> 32, key & ((1 << 32) - 1)
    b1 = np.concatenate([t.bin1_id, tb1]); b2 = np.concatenate([t.bin2_id, tb2])
    cn = np.concatenate([t.count, rng.integers(1, 4, len(tb1)).astype(np.int32)])
    order = np.lexsort((b2, b1))
    cnt = cn[order] * 0.25 if float_counts else cn[order]
    kr = 1.0 / (t.weights['weight'] * 1.37)                   # a divisive column as hic2cool writes it
    full = pixels.PixelTable(names, t.chromsizes, RESOL, off, b1[order], b2[order], cnt, {'weight': t.weights['weight'], 'KR': kr})
    return names, chroms, full


def write(path, t, h5py, pixel_kw=None, cooler_style=False):
    pixel_kw = dict(chunks=(4096,), compression='gzip') if pixel_kw is None else pixel_kw
    with h5py.File(path, 'w') as f:
        g = f.create_group(GROUP)
        g.attrs['bin-size'] = t.binsize
        nb = int(t.chrom_offset[-1])
        chrom_ids = np.repeat(np.arange(len(t.chromnames)), np.diff(t.chrom_offset))
        if cooler_style:                                      # what `cooler.create` adds around the tables
            g.attrs['format'] = 'HDF5::Cooler'
            g.attrs['format-version'] = 3
            g.attrs['bin-type'] = 'fixed'
            g.attrs['storage-mode'] = 'symmetric-upper'
            g.attrs['nbins'] = nb
            g.attrs['nchroms'] = len(t.chromnames)
            g.attrs['nnz'] = len(t.count)
            g.attrs['genome-assembly'] = 'synthetic'
            f.attrs['format'] = 'HDF5::MCOOL'
            dt = h5py.enum_dtype({n: i for i, n in enumerate(t.chromnames)}, basetype='i4')
            g.create_dataset('bins/chrom', data=chrom_ids.astype('i4'), dtype=dt, chunks=(min(nb, 4096),), compression='gzip', shuffle=True)
            g.create_dataset('chroms/name', data=np.array(t.chromnames, dtype='S'), chunks=(len(t.chromnames),), compression='gzip')
            g.create_dataset('chroms/length', data=np.asarray(t.chromsizes, np.int32), chunks=(len(t.chromnames),), compression='gzip')
        else:
            g.create_dataset('bins/chrom', data=chrom_ids)
            g.create_dataset('chroms/name', data=np.array(t.chromnames, dtype='S'))
            g.create_dataset('chroms/length', data=t.chromsizes)
        start = np.concatenate([np.arange(n) * t.binsize for n in np.diff(t.chrom_offset)])
        g.create_dataset('bins/start', data=start)
        g.create_dataset('bins/end', data=start + t.binsize)
        for k, v in t.weights.items():
            g.create_dataset('bins/' + k, data=v)
        g.create_dataset('pixels/bin1_id', data=t.bin1_id, **pixel_kw)
        g.create_dataset('pixels/bin2_id', data=t.bin2_id, **pixel_kw)
        g.create_dataset('pixels/count', data=t.count, **pixel_kw)
        g.create_dataset('indexes/chrom_offset', data=t.chrom_offset)
        g.create_dataset('indexes/bin1_offset', data=np.searchsorted(t.bin1_id, np.arange(nb + 1), side='left'))
