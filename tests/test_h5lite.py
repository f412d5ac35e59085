"""stripenn_amd.h5lite -- the package's own reader of the HDF5 subset cooler files use -- and the real-file route through
it: what `stripenn compute file.mcool::resolutions/5000` needs where neither cooler nor h5py is installed (this
interpreter; the reference opens the file with `cooler.Cooler`, stripenn.py:80-118).

The committed file tests/golden/cool_tiny.mcool was written by h5py 3.3 / HDF5 1.10.6 from tests/cool_fixture.py's
`table(small=True)` (tests/golden/make_cool_fixture.py), cooler-style: enumerated `bins/chrom`, variable-length string
attributes, gzip + shuffle pixel columns.  The tests regenerate that table in memory and demand the same numbers from
the file.  Where h5py IS importable (the harness interpreter: /opt/conda/bin/python3.9 -m pytest tests/test_h5lite.py)
a second group of tests writes many layouts with h5py and compares dataset by dataset."""
import os
import sys
import warnings

import numpy as np
import pytest

import cool_fixture as CF
from oracle import oracle as O
from oracle_backend import OracleBackend
from stripenn_amd import h5lite, io as sio, pixels, stripenn

warnings.filterwarnings('ignore')
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cool_tiny.mcool')
GROUP = CF.GROUP


@pytest.fixture(scope='module')
def tiny():
    return CF.table(small=True)


def test_tables_of_the_committed_cooler_file(tiny):
    names, chroms, t = tiny
    with h5lite.File(FIXTURE) as f:
        assert f.keys() == ['resolutions'] and f['resolutions'].keys() == ['5000']
        g = f[GROUP]
        assert g.keys() == ['bins', 'chroms', 'indexes', 'pixels']
        assert g['bins'].keys() == ['KR', 'chrom', 'end', 'start', 'weight']
        assert int(g.attrs['bin-size']) == 5000 and int(g.attrs['nbins']) == int(t.chrom_offset[-1]) and int(g.attrs['nnz']) == len(t.count)
        assert 'format' in g.attrs
        with pytest.raises(h5lite.H5LiteUnsupported):          # variable-length strings are not read (and not needed)
            g.attrs['format']
        assert [c.decode() for c in g['chroms/name'][:]] == names
        assert np.array_equal(g['chroms/length'][:], t.chromsizes) and g['chroms/length'].dtype == np.int32
        nb = int(t.chrom_offset[-1])
        assert np.array_equal(g['bins/chrom'][:], np.repeat(np.arange(len(names)), np.diff(t.chrom_offset)))     # enumerated type: its values
        assert np.array_equal(g['bins/weight'][:], t.weights['weight'], equal_nan=True)
        assert np.array_equal(g['bins/KR'][:], t.weights['KR'], equal_nan=True)
        assert np.array_equal(g['indexes/chrom_offset'][:], t.chrom_offset)
        for k, v in (('bin1_id', t.bin1_id), ('bin2_id', t.bin2_id), ('count', t.count)):
            d = g['pixels/' + k]
            assert d.shape == v.shape and d.dtype == v.dtype and d.chunks == (4096,) and d.compression == 'gzip' and d.shuffle
            assert np.array_equal(d[:], v)
            for lo, hi in ((0, 1), (4095, 4097), (100000, 123456), (len(v) - 5, len(v)), (777, 777)):
                assert np.array_equal(d[lo:hi], v[lo:hi]), (k, lo, hi)
            assert d[12345] == v[12345] and d[-1] == v[-1] and len(d) == len(v)
            mask, raw = d.id.read_direct_chunk((8192,))
            assert mask == 0 and 0 < len(raw) < 4096 * v.dtype.itemsize
        assert np.array_equal(g['indexes/bin1_offset'][:], np.searchsorted(t.bin1_id, np.arange(nb + 1), side='left'))
        with pytest.raises(KeyError):
            g['pixels/nothing']


def test_lazy_table_through_the_builtin_reader(tiny):
    """pixels.CoolTable on the built-in reader: per-chromosome cis columns (chunks inflated on host threads), dense fetches
    and row queries equal the in-memory table's."""
    names, chroms, t = tiny
    lazy = pixels.CoolTable(FIXTURE, GROUP, chunk=50000, backend='h5lite')
    assert isinstance(lazy._h5, h5lite.File) and lazy.chromnames == names and lazy.binsize == 5000
    for nm in names:
        lazy.prefetch(nm)
        a, b = lazy.chrom_pixels(nm), t.chrom_pixels(nm)
        assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])) and a[2].dtype == b[2].dtype and a[3:] == b[3:]
        assert a[1].dtype == np.int32            # the file readers narrow bin2_id while they copy the pieces (half the PCIe bytes)
    assert lazy.direct_reads > 0 and 0 < lazy.max_read <= 50000
    for balance in ('weight', 'KR', False):
        sl, sm = pixels.PixelSelector(lazy, balance), pixels.PixelSelector(t, balance)
        for reg in (('chrB',), ('chrA:500001-1500000', 'chrA:1-2000000')):
            assert np.array_equal(sl.fetch(*reg), sm.fetch(*reg), equal_nan=True), (balance, reg)
    lazy.threads = 1                                            # the reader's own slicing instead of the chunk decoder
    a, b = lazy.chrom_pixels('chrB'), t.chrom_pixels('chrB')
    assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3]))
    lazy.close()
    eager = pixels.PixelTable.from_cool(FIXTURE, GROUP)
    assert np.array_equal(eager.bin2_id, t.bin2_id) and np.array_equal(eager.count, t.count) and eager.chromnames == names


def test_compute_from_the_committed_file_equals_compute_from_the_table(tiny, tmp_path, monkeypatch):
    """`stripenn compute tests/golden/cool_tiny.mcool::resolutions/5000` with cooler AND h5py out of the way == the same
    genome as an in-memory table: byte-identical TSVs (oracle backend on the CPU; tests/test_gpu_round2.py runs the same
    file through the HIP backend)."""
    names, chroms, t = tiny
    monkeypatch.setitem(sys.modules, 'cooler', None)
    monkeypatch.setitem(sys.modules, 'h5py', None)
    info = sio.open_matrix(FIXTURE + '::' + GROUP)
    assert list(info.chromnames) == names and info.binsize == 5000 and 'KR' in info.bins().columns
    gw = O.gauss_weights(2.0)[0]
    outs = []
    for src in ('file', 'table'):
        if src == 'table':
            monkeypatch.setattr(stripenn, 'open_matrix', lambda cool: sio.pixel_matrix(t))
        out = str(tmp_path / src)
        stripenn.compute(FIXTURE + '::' + GROUP, out, 'KR', 'all', 2.0, 10, 8, '0.97,0.99', 2, 0.5, '0', False, 3, 7,
                         force=True, backend=OracleBackend(gauss_w=gw))
        outs.append([open(os.path.join(out, f)).read() for f in ('result_unfiltered.tsv', 'result_filtered.tsv')])
    assert outs[0] == outs[1] and outs[0][0].count('\n') > 5


def test_not_an_hdf5_file(tmp_path):
    p = tmp_path / 'x.cool'
    p.write_bytes(b'not hdf5' * 100)
    with pytest.raises(h5lite.H5LiteError):
        h5lite.File(str(p))


# ---------------------------------------------------------------------------------------- against h5py, where there is one
def _compare(g5, gl, h5py, path=''):
    n = 0
    assert sorted(g5.keys()) == gl.keys(), path
    for k in g5.keys():
        a, b = g5[k], gl[k]
        if isinstance(a, h5py.Group):
            n += _compare(a, b, h5py, path + '/' + k)
        else:
            assert a.shape == b.shape and a.dtype == b.dtype, (path, k)
            assert np.array_equal(a[...], b[...], equal_nan=a.dtype.kind == 'f'), (path, k)
            n += 1
    for k, v in g5.attrs.items():
        if isinstance(v, (str, bytes)) or (hasattr(v, 'dtype') and v.dtype.kind in 'OSU'):
            assert k in gl.attrs
            continue
        assert np.array_equal(np.asarray(v), np.asarray(gl.attrs[k])), (path, k)
    return n


def test_layouts_written_by_h5py(tmp_path):
    h5py = pytest.importorskip('h5py')
    rng = np.random.default_rng(1)
    p = str(tmp_path / 'v.h5')
    with h5py.File(p, 'w') as f:
        f.attrs['bin-size'] = 5000; f.attrs['fmt'] = 'HDF5::Cooler'; f.attrs['x'] = np.arange(5.0); f.attrs['i32'] = np.int32(-7)
        g = f.create_group('a/b/c')
        g.create_dataset('contig', data=rng.integers(0, 100, 1000))
        g.create_dataset('f16', data=rng.random(77).astype(np.float16))
        g.create_dataset('be', data=np.arange(10, dtype='>i4'))
        g.create_dataset('u8', data=rng.integers(0, 255, 5000).astype(np.uint8), chunks=(777,), compression='gzip', shuffle=True)
        g.create_dataset('f64c', data=rng.random(100000), chunks=(4096,), compression='gzip', compression_opts=9, shuffle=True, fletcher32=True)
        g.create_dataset('plain_chunks', data=rng.integers(0, 9, 30000), chunks=(1000,))
        g.create_dataset('ext', shape=(0,), maxshape=(None,), dtype=np.int64, chunks=(512,), compression='gzip')
        g['ext'].resize((2000,)); g['ext'][100:1900] = np.arange(1800)
        g.create_dataset('empty', shape=(0,), maxshape=(None,), dtype=np.int32, chunks=(16,))
        g.create_dataset('hole', shape=(10000,), dtype=np.float64, chunks=(1000,)); g['hole'][5000:5100] = 1.5
        g.create_dataset('two_d', data=rng.random((300, 7)), chunks=(64, 7), compression='gzip')
        g.create_dataset('names', data=np.array(['chr1', 'chr2', 'chrX_long'], dtype='S'))
        g.create_dataset('scalar', data=3.25)
        dt = h5py.enum_dtype({'chr%d' % i: i for i in range(25)}, basetype='i1')
        g.create_dataset('chrom_enum', data=rng.integers(0, 25, 4000).astype('i1'), dtype=dt, chunks=(1000,), compression='gzip')
        many = f.create_group('many')                           # a group B-tree of several symbol nodes, attributes over several header blocks
        for i in range(300):
            many.create_dataset('d%03d' % i, data=np.arange(i % 7 + 1))
        for i in range(40):
            many.attrs['a%02d' % i] = i
    with h5py.File(p, 'r') as f5, h5lite.File(p) as fl:
        assert _compare(f5, fl, h5py) == 313
        d = fl['a/b/c/f64c']
        assert (d.chunks, d.compression, d.shuffle, d.fletcher32, d.compression_opts) == ((4096,), 'gzip', True, True, 9)
        assert d.id.read_direct_chunk((4096,)) == f5['a/b/c/f64c'].id.read_direct_chunk((4096,))
        assert np.array_equal(fl['a/b/c/u8'][100:4000:3], f5['a/b/c/u8'][100:4000:3])
    # what the reader does not implement is refused by name, not misread
    p2 = str(tmp_path / 'latest.h5')
    with h5py.File(p2, 'w', libver='latest') as f:
        f.create_dataset('x', data=np.arange(100000), chunks=(1000,), compression='gzip')
    with h5lite.File(p2) as fl:
        assert fl.keys() == ['x']
        with pytest.raises(h5lite.H5LiteUnsupported):
            fl['x']
    p3 = str(tmp_path / 'lzf.h5')
    with h5py.File(p3, 'w') as f:
        f.create_dataset('x', data=np.arange(1000), chunks=(100,), compression='lzf')
    with h5lite.File(p3) as fl, pytest.raises(h5lite.H5LiteUnsupported):
        fl['x']
