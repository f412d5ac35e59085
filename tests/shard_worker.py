"""Worker for tests/test_shard_gloo.py (spawned processes import it by module name)."""
import os
import pickle
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

COOL = 'synth:chr1=6996790,chr2=4499223,chr3=3249000;resol=5000;seed=31'
ARGS = dict(norm='KR', chrom='all', canny=2.0, minL=10, maxW=8, maxpixel='0.95,0.98', numcores=2, pvalue=0.1, mask='0',
            slow=False, bfilter=3, seed=123456789)


def _factory(rank):
    from oracle_backend import OracleBackend
    return OracleBackend()


def _factory_failing_rank1(rank):
    """as _factory, but rank 1's backend raises in the middle of a run whose Canny sigma is 3.0 (ComputePool failure test)"""
    b = _factory(rank)
    if rank == 1:
        for name in ('stripe_search', 'stripe_search_begin'):
            if hasattr(b, name):
                def poisoned(frames, M_levels, sigma, *a, _orig=getattr(b, name), **k):
                    if sigma == 3.0:
                        raise RuntimeError('injected failure on rank 1')
                    return _orig(frames, M_levels, sigma, *a, **k)
                setattr(b, name, poisoned)
    return b


def run(rank, world, port, outdir, numcores=2):
    warnings.filterwarnings('ignore')
    import torch.distributed as dist
    from stripenn_amd import shard
    if world > 1:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        dist.init_process_group('gloo', rank=rank, world_size=world)
    a = dict(ARGS, numcores=numcores)
    sys.stdout = open(os.devnull, 'w')
    res = shard.sharded_compute(rank, world, COOL, os.path.join(outdir, 'w%d_c%d' % (world, numcores)), a['norm'], a['chrom'], a['canny'],
                                a['minL'], a['maxW'], a['maxpixel'], a['numcores'], a['pvalue'], a['mask'], a['slow'],
                                a['bfilter'], a['seed'], force=True, backend_factory=_factory, write=True)
    if world > 1:
        dist.destroy_process_group()


def _factory_hip_device0(rank):
    """every rank on device 0 with the HIP backend (tools/rehearse_pool_hip.py: the multi-process driver on a one-GPU box)"""
    from stripenn_amd.backend import HipBackend
    return HipBackend(0)
