"""bench.py's unit plan (host logic, no GPU): the units of all ranks of an N-rank run cover every frame of the genome exactly
once, whatever the timing mode -- steps pipelined (units of one device chunk) or drained (<= 128 frames per unit in a share)."""
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _nframes():
    # frames of a chromosome of n bins: getStripe.py:794-799 (one frame per 200 bins)
    return [(-(-s // bench.RESOL) + 199) // 200 for s in bench.MM10]


@pytest.mark.parametrize('pipelined', ['1', '0'])
@pytest.mark.parametrize('world', [1, 2, 3, 4, 8])
def test_units_partition_the_frame_grid(world, pipelined, monkeypatch):
    monkeypatch.setenv('STP_BENCH_PIPELINE_STEPS', pipelined)
    for k in ('STP_BENCH_PIECE', 'STP_BENCH_ORDER'):
        monkeypatch.delenv(k, raising=False)
    stub = types.SimpleNamespace(nframes=_nframes())
    seen = {}
    per_rank = []
    for rank in range(world):
        units = bench._Workload.units_for(stub, world, rank)
        per_rank.append(sum(f1 - f0 for _, f0, f1 in units))
        limit = 204 if (world == 1 or pipelined == '1') else 128
        for ci, f0, f1 in units:
            assert 0 <= f0 < f1 <= stub.nframes[ci] and f1 - f0 <= limit
            for f in range(f0, f1):
                assert (ci, f) not in seen, 'frame handed out twice'
                seen[(ci, f)] = rank
    assert len(seen) == sum(stub.nframes)
    # equal frame counts up to the snapping of cuts onto chromosome boundaries (shard.SNAP_FRAMES)
    from stripenn_amd import shard
    assert max(per_rank) - min(per_rank) <= 2 * shard.SNAP_FRAMES + 1


def test_interleaved_order_is_a_permutation(monkeypatch):
    stub = types.SimpleNamespace(nframes=_nframes())
    monkeypatch.delenv('STP_BENCH_ORDER', raising=False)
    a = bench._Workload.units_for(stub, 1, 0)
    monkeypatch.setenv('STP_BENCH_ORDER', 'interleave')
    b = bench._Workload.units_for(stub, 1, 0)
    assert sorted(a) == sorted(b) and a != b
