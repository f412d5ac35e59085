"""Writes tests/golden/cool_tiny.mcool: a two-chromosome cooler-schema file (tests/cool_fixture.py: `table(small=True)`
laid out cooler-style -- enumerated bins/chrom, string attributes, gzip + shuffle pixel columns in 4096-row chunks).
It is DATA for the tests of the package's own HDF5 reader (stripenn_amd/h5lite.py), which must work where h5py is absent;
writing it needs h5py:
    /opt/conda/bin/python3.9 tests/golden/make_cool_fixture.py"""
import os
import sys

import h5py

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import cool_fixture as CF      # noqa: E402

names, chroms, t = CF.table(small=True)
out = os.path.join(HERE, 'cool_tiny.mcool')
CF.write(out, t, h5py, pixel_kw=dict(chunks=(4096,), compression='gzip', compression_opts=6, shuffle=True), cooler_style=True)
print(out, os.path.getsize(out), 'bytes;', len(t.count), 'pixels, h5py', h5py.__version__, 'HDF5', h5py.version.hdf5_version)
