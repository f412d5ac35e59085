"""stp_format_tsv (stripenn_amd/csrc/stp_tsv.h; host code of the C-ABI library, no GPU needed): float fields are Python's
repr(float) -- what pandas' to_csv writes for a float64 column (stripenn.py:156-157, score.py:60) -- for millions of values of
every magnitude; whole tables equal pandas' bytes through both writers (library / Python)."""
import ctypes as C
import io
import os

import numpy as np
import pandas as pd
import pytest

from stripenn_amd import hip
from stripenn_amd.stripenn import _write_tsv_native, write_tsv


def _format_floats(v):
    L = hip.load()
    v = np.ascontiguousarray(v, dtype=np.float64)
    vp = C.c_void_p
    kind = (C.c_int32 * 1)(1)
    data = (vp * 1)(v.ctypes.data)
    cap = 27 * len(v) + 1
    buf = np.empty(cap, dtype=np.uint8)
    n = C.c_int64(0)
    L.stp_format_tsv.argtypes = [C.c_int32, vp, vp, vp, vp, vp, C.c_int64, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.stp_format_tsv.restype = C.c_int
    rc = L.stp_format_tsv(1, C.cast(kind, vp), C.cast(data, vp), None, None, None, len(v), buf.ctypes.data, cap, C.byref(n))
    assert rc == 0
    return bytes(memoryview(buf)[:n.value]).decode('ascii').split('\n')[:-1]


def test_float_fields_are_pythons_repr():
    rng = np.random.default_rng(11)
    parts = [rng.integers(-2**63, 2**63 - 1, 1_500_000, dtype=np.int64).view(np.float64),      # random bit patterns: every exponent
             rng.random(500_000), rng.random(500_000) * 10.0 ** rng.integers(-25, 25, 500_000),
             np.round(rng.random(200_000) * 1e6, 3), rng.integers(-10**17, 10**17, 200_000).astype(np.float64),
             10.0 ** np.arange(-330, 310, dtype=np.float64), -(10.0 ** np.arange(-30, 30, dtype=np.float64)),
             np.array([0.0, -0.0, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 1e15, 1e16, 9999999999999998.0, 1e-4, 9.999e-5,
                       0.1 + 0.2, 1 / 3, 123456789012345680.0, np.inf, -np.inf, 2.5, 100.0, 1e22, 1e23, 8.41e21, 9007199254740993.0])]
    v = np.concatenate(parts)
    v = v[~np.isnan(v)]
    got = _format_floats(v)
    want = [repr(x) for x in v.tolist()]
    assert len(got) == len(want)
    bad = [(g, w) for g, w in zip(got, want) if g != w]
    assert not bad, bad[:5]
    assert _format_floats(np.array([np.nan, 1.0, np.nan])) == ['', '1.0', '']


def _pandas_bytes(df):
    b = io.StringIO()
    df.to_csv(b, sep='\t', header=True, index=False)
    return b.getvalue()


@pytest.mark.parametrize('writer', ['native', 'python'])
def test_both_writers_equal_pandas(tmp_path, monkeypatch, writer):
    monkeypatch.setenv('STP_TSV', writer)
    rng = np.random.default_rng(3)
    m = 4000
    df = pd.DataFrame({'chr': ['chr%d' % (i % 21) for i in range(m)], 'pos1': rng.integers(1, 2**40, m), 'pos2': rng.integers(-9, 9, m).astype(np.int32),
                       'Mean': rng.random(m) * 50, 'maxpixel': ['%s%%' % (q * 100) for q in rng.choice([0.95, 0.96, 0.99], m)],
                       'pvalue': rng.random(m) * 10.0 ** rng.integers(-9, 0, m), 'Stripiness': rng.standard_normal(m) * 30})
    df.loc[df.index[::9], 'Stripiness'] = np.nan
    for tag, t in (('table', df), ('rows0', df.iloc[:0]), ('sorted', df.sort_values('Stripiness', ascending=False)),
                   ('objints', pd.DataFrame({'a': pd.Series([1, 2, 3], dtype=object), 'b': pd.Series([0.5, None, 2.0], dtype=object)}))):
        p = str(tmp_path / (tag + '.tsv'))
        write_tsv(t, p)
        assert open(p, newline='').read() == _pandas_bytes(t), tag


def test_duplicate_column_names_go_to_pandas(tmp_path, monkeypatch):
    df = pd.DataFrame([[1, 2.5, 'a'], [3, 4.5, 'b']], columns=['x', 'x', 's'])
    for writer in ('native', 'python'):
        monkeypatch.setenv('STP_TSV', writer)
        p = str(tmp_path / (writer + '.tsv'))
        write_tsv(df, p)
        assert open(p, newline='').read() == _pandas_bytes(df)


def test_native_writer_declines_what_it_cannot_describe(tmp_path):
    p = str(tmp_path / 'x.tsv')
    assert not _write_tsv_native(pd.DataFrame({'n': ['a\tb', 'c'], 'v': [1.0, 2.0]}), p)            # needs quoting
    assert not _write_tsv_native(pd.DataFrame({'m': pd.Series([1, 'x', 2.5], dtype=object)}), p)    # mixed cells
    assert not _write_tsv_native(pd.DataFrame({'b': [True, False]}), p)
    assert not os.path.exists(p)
    assert _write_tsv_native(pd.DataFrame({'a': [1, 2], 's': ['x', 'y']}), p) and open(p).read() == 'a\ts\n1\tx\n2\ty\n'


def test_capacity_and_argument_errors():
    L = hip.load()
    vp = C.c_void_p
    v = np.array([1.5, 2.5]); kind = (C.c_int32 * 1)(1); data = (vp * 1)(v.ctypes.data)
    buf = np.empty(8, dtype=np.uint8); n = C.c_int64(0)
    L.stp_format_tsv.argtypes = [C.c_int32, vp, vp, vp, vp, vp, C.c_int64, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.stp_format_tsv.restype = C.c_int
    assert L.stp_format_tsv(1, C.cast(kind, vp), C.cast(data, vp), None, None, None, 2, buf.ctypes.data, 8, C.byref(n)) == -2   # STP_E_CAPACITY
    kind[0] = 7
    assert L.stp_format_tsv(1, C.cast(kind, vp), C.cast(data, vp), None, None, None, 2, buf.ctypes.data, 8, C.byref(n)) == -1   # STP_E_ARG
    kind[0] = 2                                                                                                                  # strings without a table
    assert L.stp_format_tsv(1, C.cast(kind, vp), C.cast(data, vp), None, None, None, 2, buf.ctypes.data, 8, C.byref(n)) == -1
