"""BASELINE.json configs[0] at CONFIG size on the CPU: the façade with the oracle backend against the tables the
unmodified reference produced for the chr16-size chromosome with maxpixel 0.99 (tests/golden/e2e_chr16.npz).  The
five-level sweep (configs[1]) runs with the HIP backend in test_gpu_chr16_reference.py."""
import warnings

import e2e_common as E
from oracle_backend import OracleBackend

warnings.filterwarnings('ignore')


def test_chr16_single_level_matches_the_reference_tables():
    obj, rows = E.run_chr16(lambda gw: OracleBackend(gauss_w=gw), float_exact=True, configs=(0,))
    assert rows[0] == (111, 110)
