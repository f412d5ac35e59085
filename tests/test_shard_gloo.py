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


def test_frame_spans_balance_mm10():
    """(chromosome x frame) work queue: contiguous spans, every frame exactly once, imbalance of one frame -- or, with cuts
    snapped onto a chromosome boundary <= 8 frames away (round 6: no slivers), of at most 2 x 8 frames."""
    mm10 = [195471971, 182113224, 160039680, 156508116, 151834684, 149736546, 145441459, 129401213, 124595110,
            130694993, 122082543, 120129022, 120421639, 124902244, 104043685, 98207768, 94987271, 90702639, 61431566,
            171031299]
    nfr = shard.chrom_nframes(mm10, 5000)
    assert sum(nfr) == 2645
    for n in (1, 2, 3, 4, 8):
        spans = shard.frame_spans(nfr, n)
        seen = [[0] * k for k in nfr]
        for sp in spans:
            for ci, lo, hi in sp:
                assert 0 <= lo < hi <= nfr[ci]
                for f in range(lo, hi):
                    seen[ci][f] += 1
        assert all(v == 1 for row in seen for v in row)
        loads = [sum(hi - lo for _, lo, hi in sp) for sp in spans]
        assert max(loads) - min(loads) <= 2 * shard.SNAP_FRAMES + 1 and max(loads) / (sum(loads) / n) < 1.03
        assert all(hi - lo > shard.SNAP_FRAMES or (lo == 0 and hi == nfr[ci]) for sp in spans for ci, lo, hi in sp)     # no slivers
        plain = shard.frame_spans(nfr, n, snap=0)
        loads0 = [sum(hi - lo for _, lo, hi in sp) for sp in plain]
        assert max(loads0) - min(loads0) <= 1


def _spawn(world, out, numcores):
    import torch.multiprocessing as mp
    import shard_worker
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(shard_worker.run, args=(world, port, out, numcores), nprocs=world, join=True)


@pytest.mark.parametrize('numcores,worlds', [(2, (2, 3)), (1, (2,))])
def test_sharded_equals_single_process(tmp_path, numcores, worlds):
    """world_size 2 / 3 over gloo (frame spans cut chr1 / chr2 in the middle: 7 + 5 + 4 frames) write the TSVs of
    the single-process run byte for byte, in both PRNG modes of the background step (numcores 2: one stream per
    chromosome; numcores 1: one stream across chromosomes, replayed by every rank)."""
    import shard_worker
    out = str(tmp_path)
    shard_worker.run(0, 1, 0, out, numcores)
    sys.stdout = sys.__stdout__
    for world in worlds:
        _spawn(world, out, numcores)
        for name in ('result_unfiltered.tsv', 'result_filtered.tsv'):
            a = open(os.path.join(out, 'w1_c%d' % numcores, name)).read()
            b = open(os.path.join(out, 'w%d_c%d' % (world, numcores), name)).read()
            assert len(a.splitlines()) > 10
            assert a == b, (name, world)
        log = open(os.path.join(out, 'w%d_c%d' % (world, numcores), 'stripenn.log')).read()
        assert 'gpus: %d' % world in log and 'rank 1' in log


def test_sharded_world1_equals_unsharded_driver(tmp_path):
    """sharded_compute(world=1) and stripenn.compute are the same computation."""
    import shard_worker
    from oracle_backend import OracleBackend
    from stripenn_amd import stripenn
    out = str(tmp_path)
    shard_worker.run(0, 1, 0, out, 2)
    sys.stdout = sys.__stdout__
    a = shard_worker.ARGS
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        stripenn.compute(shard_worker.COOL, os.path.join(out, 'plain'), a['norm'], a['chrom'], a['canny'], a['minL'], a['maxW'],
                         a['maxpixel'], 2, a['pvalue'], a['mask'], a['slow'], a['bfilter'], a['seed'], force=True,
                         backend=OracleBackend())
    for name in ('result_unfiltered.tsv', 'result_filtered.tsv'):
        assert open(os.path.join(out, 'w1_c2', name)).read() == open(os.path.join(out, 'plain', name)).read()


def test_persistent_pool_serves_several_runs(tmp_path):
    """shard.ComputePool: two rank processes (gloo, oracle backend) stay alive across three `compute` runs with
    different parameters; every run writes the TSVs of the corresponding single-process run, byte for byte, and the
    same two processes served them all (start-up paid once)."""
    import contextlib, io
    import shard_worker
    from oracle_backend import OracleBackend
    from stripenn_amd import stripenn
    a = shard_worker.ARGS
    runs = [dict(maxpixel='0.95,0.98', numcores=2, canny=2.0), dict(maxpixel='0.97', numcores=2, canny=2.5),
            dict(maxpixel='0.95,0.98', numcores=1, canny=2.0)]
    with shard.ComputePool(2, backend_factory=shard_worker._factory) as pool:
        pids = dict(pool.pids)
        assert len(set(pids.values())) == 2 and os.getpid() not in pids.values()
        for k, r in enumerate(runs):
            out = os.path.join(str(tmp_path), 'pool%d' % k)
            secs = pool.compute(shard_worker.COOL, out, a['norm'], a['chrom'], r['canny'], a['minL'], a['maxW'], r['maxpixel'],
                                r['numcores'], a['pvalue'], a['mask'], a['slow'], a['bfilter'], a['seed'])
            assert secs > 0 and pool.pids == pids
    for k, r in enumerate(runs):
        ref = os.path.join(str(tmp_path), 'ref%d' % k)
        with contextlib.redirect_stdout(io.StringIO()):
            stripenn.compute(shard_worker.COOL, ref, a['norm'], a['chrom'], r['canny'], a['minL'], a['maxW'], r['maxpixel'],
                             r['numcores'], a['pvalue'], a['mask'], a['slow'], a['bfilter'], a['seed'], force=True,
                             backend=OracleBackend())
        for name in ('result_unfiltered.tsv', 'result_filtered.tsv'):
            got = open(os.path.join(str(tmp_path), 'pool%d' % k, name)).read()
            assert got == open(os.path.join(ref, name)).read(), (k, name)
            assert name != 'result_unfiltered.tsv' or len(got.splitlines()) > 5
        assert 'gpus: 2' in open(os.path.join(str(tmp_path), 'pool%d' % k, 'stripenn.log')).read()


def test_pool_reports_a_failing_job_and_shuts_down(tmp_path):
    """A job that fails on ONE rank (rank 1's backend raises in the middle of the run, the other rank is left waiting in a
    gloo exchange): the pool reports the failure as an exception naming the rank, terminates every rank process -- nothing
    is re-launched -- and refuses further work; a run before the failure is complete and correct."""
    import shard_worker
    a = shard_worker.ARGS
    pool = shard.ComputePool(2, backend_factory=shard_worker._factory_failing_rank1)
    procs = list(pool.procs)
    pids = dict(pool.pids)
    out0 = os.path.join(str(tmp_path), 'ok')
    pool.compute(shard_worker.COOL, out0, a['norm'], a['chrom'], 2.0, a['minL'], a['maxW'], '0.97', 2, a['pvalue'], a['mask'],
                 a['slow'], a['bfilter'], a['seed'])
    assert len(open(os.path.join(out0, 'result_unfiltered.tsv')).read().splitlines()) > 5
    with pytest.raises(RuntimeError, match='rank 1 failed.*injected failure'):
        pool.compute(shard_worker.COOL, os.path.join(str(tmp_path), 'bad'), a['norm'], a['chrom'], 3.0, a['minL'], a['maxW'], '0.97', 2,
                     a['pvalue'], a['mask'], a['slow'], a['bfilter'], a['seed'], timeout=300)
    assert pool.procs == [] and pool.pids == pids                  # shut down, and no rank was started in the meantime
    for p in procs:
        p.join(20)
        assert not p.is_alive()
    assert not os.path.exists(os.path.join(str(tmp_path), 'bad', 'result_unfiltered.tsv'))
    with pytest.raises(Exception):
        pool.compute(shard_worker.COOL, os.path.join(str(tmp_path), 'after'), a['norm'], a['chrom'], 2.0, a['minL'], a['maxW'], '0.97', 2,
                     a['pvalue'], a['mask'], a['slow'], a['bfilter'], a['seed'], timeout=20)
