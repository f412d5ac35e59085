"""Multi-process path on CPU: world_size 2 over gloo (the oracle as compute backend) must write
exactly the TSVs of the single-process run -- same rows, same order, same values."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from stripenn_amd import shard  # noqa: E402


def test_lpt_balance_mm10():
    mm10 = [195471971, 182113224, 160039680, 156508116, 151834684, 149736546, 145441459, 129401213, 124595110,
            130694993, 122082543, 120129022, 120421639, 124902244, 104043685, 98207768, 94987271, 90702639, 61431566,
            171031299]
    costs = shard.chrom_costs(mm10, 5000)
    assert sum(costs) == 2645          # mm10 1-19,X at 5 kb = 2 645 frames
    for n in (2, 4, 8):
        parts = shard.lpt_assign(costs, n)
        assert sorted(sum(parts, [])) == list(range(20))
        loads = [sum(costs[i] for i in p) for p in parts]
        assert max(loads) / (sum(loads) / n) < 1.15     # whole-chromosome units: 1.14 at 8 GPUs (7.0x ideal speed-up)


def test_world2_equals_world1(tmp_path):
    import torch.multiprocessing as mp
    import shard_worker
    out = str(tmp_path)
    shard_worker.run(0, 1, 0, out)
    sys.stdout = sys.__stdout__
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(shard_worker.run, args=(2, port, out), nprocs=2, join=True)
    for name in ('result_unfiltered.tsv', 'result_filtered.tsv'):
        a = open(os.path.join(out, 'w1', name)).read()
        b = open(os.path.join(out, 'w2', name)).read()
        assert len(a.splitlines()) > 10
        assert a == b, name
    log = open(os.path.join(out, 'w2', 'stripenn.log')).read()
    assert 'gpus: 2' in log and 'rank 1' in log
