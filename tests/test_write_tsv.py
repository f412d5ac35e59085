"""stripenn.write_tsv must produce exactly what pandas' `to_csv(sep='\\t', header=True, index=False)` writes (the
reference's output call, stripenn.py:156-157 / score.py:60): random and extreme floats, NaN, integer, string and
object columns, empty tables, and the golden result tables."""
import io
import os

import numpy as np
import pandas as pd

from stripenn_amd.stripenn import write_tsv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(df, tmp_path, tag):
    p = str(tmp_path / (tag + '.tsv'))
    write_tsv(df, p)
    b = io.StringIO()
    df.to_csv(b, sep='\t', header=True, index=False)
    assert open(p, newline='').read() == b.getvalue(), tag


def test_write_tsv_equals_pandas(tmp_path):
    rng = np.random.default_rng(5)
    n = 5000
    bits = rng.integers(0, 2**63, n, dtype=np.int64).view(np.float64)          # random bit patterns: every magnitude
    bits[~np.isfinite(bits)] = 1.5
    special = np.array([0.0, -0.0, 1.0, 97.0, 1e16, 1e15, 123456789012345680.0, 1e-4, 1e-5, 0.1 + 0.2, 5e-324, 1.7976931348623157e308,
                        np.inf, -np.inf, np.nan, 1 / 3, 2.5e-7, 99999999999999.98, 0.001, 1e22, 1e23])
    f = np.concatenate([bits, special, rng.random(n), rng.random(n) * 10.0 ** rng.integers(-12, 12, n)])
    m = len(f)
    df = pd.DataFrame({'chr': ['chr%d' % (i % 23) for i in range(m)], 'pos1': rng.integers(1, 2**40, m), 'f': f,
                       'g': np.float32(rng.random(m)).astype(np.float64), 'label': ['%s%%' % (q * 100) for q in rng.choice([0.95, 0.96, 0.97], m)],
                       'small': rng.integers(-5, 5, m).astype(np.int32)})
    df.loc[df.index[::7], 'g'] = np.nan
    _same(df, tmp_path, 'random')
    obj = pd.DataFrame({'a': pd.Series([1, 2.5, None, 'x', float('nan'), np.float64(0.1)], dtype=object), 'b': [1, 2, 3, 4, 5, 6]})
    _same(obj, tmp_path, 'object')
    _same(pd.DataFrame(columns=['chr', 'pos1', 'Stripiness']), tmp_path, 'empty')
    _same(pd.DataFrame({'name': ['a\tb', 'c"d', 'e'], 'v': [1.0, 2.0, 3.0]}), tmp_path, 'quoting')   # falls back to pandas
    # mixed frames as the driver builds them (concat of an empty typed frame and a table)
    t = pd.concat([pd.DataFrame(columns=['chr', 'pos1', 'Mean']), pd.DataFrame({'chr': ['chr1'] * 3, 'pos1': [1, 2, 3], 'Mean': [0.5, 1e-9, 3.0]})])
    _same(t, tmp_path, 'concat')


def test_write_tsv_reproduces_the_golden_tables(tmp_path):
    for tag in ('seq', '1kb', 'chr16'):
        g = np.load(os.path.join(ROOT, 'tests', 'golden', 'e2e_%s.npz' % tag))
        for key in ('tsv_unfiltered', 'tsv_filtered'):
            ref = str(g[key])
            df = pd.read_csv(io.StringIO(ref), sep='\t', float_precision='round_trip')
            p = str(tmp_path / ('%s_%s.tsv' % (tag, key)))
            write_tsv(df, p)
            assert open(p, newline='').read() == ref, (tag, key)
