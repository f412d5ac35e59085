"""The Typer command line (reference cli.py): same commands and flags; on the GPU the three commands run
end to end from a synthetic input like a user would drive them."""
import os

import pytest
from typer.testing import CliRunner

from stripenn_amd.cli import app

runner = CliRunner()
SPEC = 'synth:chrA=7000000,chrB=5500000;resol=5000;seed=23'


def test_help_lists_the_reference_commands():
    r = runner.invoke(app, ['--help'])
    assert r.exit_code == 0
    for cmd in ('compute', 'score', 'seeimage'):
        assert cmd in r.output
        h = runner.invoke(app, [cmd, '--help'])
        assert h.exit_code == 0 and '--cool' in h.output
    h = runner.invoke(app, ['compute', '--help']).output
    for flag in ('--out', '--norm', '--chrom', '--canny', '--minL', '--maxW', '--maxpixel', '--numcores', '--pvalue',
                 '--mask', '--bfilter', '--seed'):
        assert flag in h


@pytest.mark.gpu
def test_compute_score_seeimage_from_the_command_line(tmp_path):
    out = str(tmp_path / 'res')
    r = runner.invoke(app, ['compute', '--cool', SPEC, '--out', out, '--maxpixel', '0.97,0.99', '--numcores', '1',
                            '--pvalue', '0.5', '--force'])
    assert r.exit_code == 0, r.output
    head = 'chr\tpos1\tpos2\tchr2\tpos3\tpos4\tlength\twidth\tMean\tmaxpixel\tpvalue\tStripiness'
    for f in ('result_unfiltered.tsv', 'result_filtered.tsv'):
        text = open(os.path.join(out, f)).read()
        assert text.startswith(head)
    assert open(os.path.join(out, 'result_unfiltered.tsv')).read().count('\n') > 3
    assert os.path.exists(os.path.join(out, 'stripenn.log'))
    scored = str(tmp_path / 'scores.tsv')
    r = runner.invoke(app, ['score', '--cool', SPEC, '--coord', os.path.join(out, 'result_unfiltered.tsv'), '--numcores', '1',
                            '--out', scored])
    assert r.exit_code == 0, r.output
    cols = open(scored).readline().rstrip('\n').split('\t')
    assert cols[-6:] == ['pvalue_added', 'Stripiness_added', 'O_Mean_added', 'O_Sum_added', 'O/E_Mean_added', 'O/E_Total_added']
    img = str(tmp_path / 'heat')
    r = runner.invoke(app, ['seeimage', '--cool', SPEC, '--position', 'chrA:1000001-2000000', '--maxpixel', '0.98', '--out', img])
    assert r.exit_code == 0, r.output
    assert os.path.getsize(img + '_chrA:1000001-2000000_0.98qt.png') > 1000
