"""The multi-process driver with the HIP backend on ONE GPU: shard.ComputePool with two (then three) rank PROCESSES, both on
device 0 -- spawned by this parent, which never touches the GPU (no HIP call, no torch.cuda call) -- serving several
`compute` runs; every run's TSVs are compared byte for byte with a single-process HIP run of the same arguments (a child
process of its own).  A rehearsal of the process model of `stripenn compute --gpus N` on a one-GPU box, NOT a scaling
measurement: the ranks share one device.
    python tools/rehearse_pool_hip.py [outdir]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

COOL = 'synth:chr1=11500000,chr2=9500000,chr3=8000000,chr4=6250000,chr5=5500000,chr6=3500000;resol=5000;seed=71'
RUNS = [dict(maxpixel='0.96,0.98,0.99', numcores=4, canny=2.0), dict(maxpixel='0.97', numcores=4, canny=2.5),
        dict(maxpixel='0.95,0.98', numcores=1, canny=2.0)]
SINGLE = r'''
import contextlib, io, sys
sys.path.insert(0, %r)
from stripenn_amd import stripenn
with contextlib.redirect_stdout(io.StringIO()):
    stripenn.compute(%r, %r, 'KR', 'all', %r, 10, 8, %r, %r, 0.1, '0', False, 3, 123456789, force=True)
'''


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix='stp_pool_')
    import shard_worker
    from stripenn_amd import shard
    bad = 0
    for world in (2, 3):
        t0 = time.time()
        with shard.ComputePool(world, backend_factory=shard_worker._factory_hip_device0) as pool:
            print('world %d: %d rank processes up in %.1f s (pids %s; parent %d)' % (world, world, time.time() - t0, sorted(pool.pids.values()), os.getpid()), flush=True)
            assert len(set(pool.pids.values())) == world and os.getpid() not in pool.pids.values()
            for k, r in enumerate(RUNS):
                o = os.path.join(out, 'w%d_run%d' % (world, k))
                secs = pool.compute(COOL, o, 'KR', 'all', r['canny'], 10, 8, r['maxpixel'], r['numcores'], 0.1, '0', False, 3, 123456789)
                print('  run %d (%s): slowest rank %.2f s' % (k, r, secs), flush=True)
    for k, r in enumerate(RUNS):
        ref = os.path.join(out, 'single_run%d' % k)
        t0 = time.time()
        subprocess.check_call([sys.executable, '-c', SINGLE % (ROOT, COOL, ref, r['canny'], r['maxpixel'], r['numcores'])])
        print('single process run %d: %.2f s incl. start-up' % (k, time.time() - t0), flush=True)
        for world in (2, 3):
            for name in ('result_unfiltered.tsv', 'result_filtered.tsv'):
                a = open(os.path.join(out, 'w%d_run%d' % (world, k), name)).read()
                b = open(os.path.join(ref, name)).read()
                same = a == b
                bad += not same
                print('  world %d run %d %-22s %s (%d rows)' % (world, k, name, 'identical' if same else 'DIFFERENT', len(b.splitlines()) - 1), flush=True)
    print('rehearsal: %d mismatching files' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
