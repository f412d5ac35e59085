"""BASELINE.json configs[0] and configs[1] at CONFIG size against the reference itself: the chr16-size chromosome
(19 642 bins of 5 kb), `compute` with maxpixel 0.95-0.99 (configs[1]) and with 0.99 alone (configs[0], the README
example), HIP backend vs the tables the UNMODIFIED reference produced with numcores = 8 (tests/golden/e2e_chr16.npz,
written by oracle/refharness/gen_golden.py chr16): quantiles, expected values, background tables, all five extract
tables row by row (positions bit-exact; total / Mean / medpixel / pvalue), the redundancy filter, Stripiness and
both TSVs byte for byte."""
import warnings

import pytest

import e2e_common as E
from test_gpu_e2e import _HipWithWeights

pytestmark = pytest.mark.gpu
warnings.filterwarnings('ignore')


def test_chr16_compute_equals_the_reference_tables():
    obj, rows = E.run_chr16(_HipWithWeights, float_exact=True)
    assert rows[1] == (1539, 135) and rows[0] == (111, 110)      # unfiltered / filtered rows the reference wrote
    obj.backend.close()
