"""The real-file input route: cooler's HDF5 layout read lazily by stripenn_amd.pixels.CoolTable -- what replaces
`cooler.Cooler(cool).matrix(balance=norm)` (stripenn.py:80, 118) when cooler is not installed.

Needs h5py, which the harness interpreter of this image has (/opt/conda/bin/python3.9 -m pytest tests/test_cool_reader.py)
and /usr/bin/python3 has not: the test is skipped there.  It writes a cooler-schema file (an .mcool resolution group
with `weight`, a divisive `KR` column, trans pixels, `indexes/bin1_offset`), and checks the lazy reader against the
in-memory PixelTable: per-chromosome cis slices, dense fetches, row queries, piece-wise reads (never a whole column),
read-ahead, float counts, and the whole `compute` driver through `open_matrix('file.mcool::resolutions/5000')`."""
import os
import warnings

import numpy as np
import pytest

h5py = pytest.importorskip('h5py')

from oracle import oracle as O                    # noqa: E402
from oracle_backend import OracleBackend          # noqa: E402
from stripenn_amd import io as sio, pixels, stripenn, synth   # noqa: E402

warnings.filterwarnings('ignore')
import cool_fixture as CF                          # noqa: E402
RESOL, GROUP = CF.RESOL, CF.GROUP
_table = CF.table


def _write(path, t):
    CF.write(path, t, h5py)


def test_lazy_reader_equals_the_in_memory_table(tmp_path):
    names, chroms, t = _table()
    path = str(tmp_path / 't.mcool')
    _write(path, t)
    lazy = pixels.CoolTable(path, GROUP, chunk=5000)
    assert lazy.chromnames == names and lazy.binsize == RESOL
    assert np.array_equal(lazy.chromsizes, t.chromsizes) and np.array_equal(lazy.chrom_offset, t.chrom_offset)
    for k in ('weight', 'KR'):
        assert np.array_equal(lazy.weights[k], t.weights[k], equal_nan=True)
    for nm in names:
        lazy.prefetch(nm)                                    # read-ahead on a host thread, collected by chrom_pixels
        a, b = lazy.chrom_pixels(nm), t.chrom_pixels(nm)
        lo, hi = t.chrom_bins(nm)
        assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])) and a[3:] == b[3:]
        assert a[1].max() < hi and a[0].min() >= lo           # cis only: the trans pixels never leave the reader
        assert a[2].dtype == np.int32
    assert 0 < lazy.max_read <= 5000                          # a pixel column is only ever read in pieces
    for balance in ('weight', 'KR', False):
        sl, sm = pixels.PixelSelector(lazy, balance), pixels.PixelSelector(t, balance)
        for reg in (('chrB',), ('chrA:500001-1500000', 'chrA:1-2000000'), ('chrC:1-%d' % (450 * RESOL),)):
            assert np.array_equal(sl.fetch(*reg), sm.fetch(*reg), equal_nan=True), (balance, reg)
        p1, p2 = 'chrA:500001-1500000', 'chrA:1-4500000'
        assert np.array_equal(sl.row_nonzero(p1, p2), sm.row_nonzero(p1, p2))
        assert np.array_equal(sl[100:160, 850:1000], sm[100:160, 850:1000], equal_nan=True)    # global bins, across chromosomes
    # the divisive column follows cooler's rule (bias = 1 / KR), hand-computed on a window
    blk = pixels.PixelSelector(lazy, 'KR').fetch('chrA:500001-1000000')
    bias = 1.0 / t.weights['KR'][100:200]
    with np.errstate(invalid='ignore'):
        assert np.array_equal(blk, chroms['chrA'].counts(100, 200, 100, 200) * np.outer(bias, bias), equal_nan=True)
    lazy.close()


def test_float_counts_survive_the_file(tmp_path):
    names, chroms, t = _table(float_counts=True)
    assert t.count.dtype == np.float64
    path = str(tmp_path / 'f.mcool')
    _write(path, t)
    lazy = pixels.CoolTable(path, GROUP, chunk=3000)
    a, b = lazy.chrom_pixels('chrB'), t.chrom_pixels('chrB')
    assert a[2].dtype == np.float64 and np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0])
    lazy.close()


def test_compute_from_the_mcool_file_equals_compute_from_the_table(tmp_path, monkeypatch):
    """`stripenn compute file.mcool::resolutions/5000` through the h5py route (cooler absent) == the same genome as
    an in-memory table: byte-identical TSVs (oracle backend: this interpreter has no GPU library to load)."""
    import sys
    names, chroms, t = _table()
    path = str(tmp_path / 'g.mcool')
    _write(path, t)
    monkeypatch.setitem(sys.modules, 'cooler', None)           # force open_matrix's h5py branch
    info = sio.open_matrix(path + '::' + GROUP)
    assert list(info.chromnames) == names and info.binsize == RESOL and 'KR' in info.bins().columns
    gw = O.gauss_weights(2.0)[0]
    outs = []
    for src in ('file', 'table'):
        if src == 'table':
            monkeypatch.setattr(stripenn, 'open_matrix', lambda cool: sio.pixel_matrix(t))
        out = str(tmp_path / src)
        stripenn.compute(path + '::' + GROUP, out, 'KR', 'all', 2.0, 10, 8, '0.97,0.99', 2, 0.5, '0', False, 3, 7,
                         force=True, backend=OracleBackend(gauss_w=gw))
        outs.append([open(os.path.join(out, f)).read() for f in ('result_unfiltered.tsv', 'result_filtered.tsv')])
    assert outs[0] == outs[1] and outs[0][0].count('\n') > 5


@pytest.mark.parametrize('layout', ['gzip', 'gzip+shuffle', 'plain', 'lzf', 'gzip+fletcher32'])
def test_chunk_decoder_equals_hdf5s_filter_pipeline(tmp_path, layout):
    """The pixel columns of the usual cooler layouts (chunked 1-d, deflate with or without byte shuffle) are inflated
    outside HDF5 on host threads (pixels._ChunkReader); every other layout goes through h5py's ordinary read.  Both routes
    give the table's columns, whatever the piece boundaries (pieces that start and end inside HDF5 chunks, a short last
    chunk), for int32 and float64 counts."""
    names, chroms, t = _table(float_counts=(layout == 'gzip'))
    path = str(tmp_path / 'l.mcool')
    _write(path, t)
    kw = {'gzip': dict(compression='gzip'), 'gzip+shuffle': dict(compression='gzip', shuffle=True, compression_opts=6),
          'plain': dict(), 'lzf': dict(compression='lzf'), 'gzip+fletcher32': dict(compression='gzip', fletcher32=True)}[layout]
    with h5py.File(path, 'a') as f:
        g = f[GROUP]
        for k, v in (('bin1_id', t.bin1_id), ('bin2_id', t.bin2_id), ('count', t.count)):
            del g['pixels/' + k]
            g.create_dataset('pixels/' + k, data=v, chunks=(1000,), **kw)
    direct = layout in ('gzip', 'gzip+shuffle')
    for threads in (4, 1):
        lazy = pixels.CoolTable(path, GROUP, chunk=2345)
        lazy.threads = threads
        for nm in names:
            a, b = lazy.chrom_pixels(nm), t.chrom_pixels(nm)
            assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])) and a[2].dtype == b[2].dtype, (layout, threads, nm)
            assert a[1].dtype == np.int32        # bin2_id is narrowed while the pieces are copied
        n = len(t.count)
        p = lazy._read_piece(n - 1500, n)                   # the short last chunk
        assert np.array_equal(p[0], t.bin1_id[n - 1500:]) and np.array_equal(p[2], t.count[n - 1500:])
        assert (lazy.direct_reads > 0) == (direct and threads > 1), (layout, threads, lazy.direct_reads)
        lazy.close()
