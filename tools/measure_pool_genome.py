"""What a multi-rank `compute` keeps serial: shard.ComputePool with TWO rank processes (both on device 0 of a one-GPU
box: a measurement of the HOST side of the sharded driver, not of scaling) on configs[2] -- the mm10-size genome as an
in-memory-style pixel table (.npz in /dev/shm) -- three runs; per rank the phase times sharded_compute logs, rank 0's serial tail
(gather + merge + final columns + TSVs), and the same genome through the single-process driver in a child of its own (4 runs).
TSVs are compared byte for byte.   python tools/measure_pool_genome.py"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
NPZ = '/dev/shm/stp_genome_r06.npz'
MAXPIXEL = '0.95,0.96,0.97,0.98,0.99'

BUILD = r'''
import sys, time
sys.path.insert(0, %r)
import torch, bench
from stripenn_amd import synth_device
dev = torch.device('cuda', 0)
names = bench.MM10_NAMES
chroms = {n: synth_device.DeviceChrom(-(-s // bench.RESOL), i + 1, dev) for i, (n, s) in enumerate(zip(names, bench.MM10))}
t = synth_device.pixel_table(names, chroms, bench.RESOL)
t.save(%r)
print('table: %%d pixels' %% len(t.count))
'''
SINGLE = r'''
import contextlib, io, sys, time
sys.path.insert(0, %r)
from stripenn_amd import stripenn, io as sio, pixels
table = pixels.PixelTable.load(%r)
stripenn.open_matrix = lambda cool: sio.pixel_matrix(table)
for k in range(4):
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        stripenn.compute('pixels:in-memory', %r, 'weight', 'all', 2.0, 10, 8, %r, 8, 0.1, '0', False, 3, 123456789, force=True)
    print('single process run %%d: %%.3f s' %% (k, time.time() - t0), flush=True)
'''


def main():
    out = tempfile.mkdtemp(prefix='stp_poolg_')
    try:
        t0 = time.time()
        subprocess.check_call([sys.executable, '-c', BUILD % (ROOT, NPZ)])
        print('genome table built and saved in %.1f s (%.1f GB)' % (time.time() - t0, os.path.getsize(NPZ) / 1e9), flush=True)
        import shard_worker
        from stripenn_amd import shard
        t0 = time.time()
        with shard.ComputePool(2, backend_factory=shard_worker._factory_hip_device0, start_timeout=600) as pool:
            print('2 rank processes up in %.1f s' % (time.time() - t0), flush=True)
            for k in range(3):
                o = os.path.join(out, 'pool_run%d' % k)
                t1 = time.time()
                secs = pool.compute('pixels:' + NPZ, o, 'weight', 'all', 2.0, 10, 8, MAXPIXEL, 8, 0.1, '0', False, 3, 123456789)
                print('pool run %d: slowest rank %.3f s of compute (call incl. loading the table in both ranks: %.1f s)' % (k, secs, time.time() - t1), flush=True)
                for line in open(os.path.join(o, 'stripenn.log')).read().splitlines():
                    if line.startswith('rank') or line.startswith('gpus'):
                        print('    ' + line, flush=True)
        ref = os.path.join(out, 'single')
        subprocess.check_call([sys.executable, '-c', SINGLE % (ROOT, NPZ, ref, MAXPIXEL)])
        bad = 0
        for name in ('result_unfiltered.tsv', 'result_filtered.tsv'):
            a = open(os.path.join(out, 'pool_run2', name)).read()
            b = open(os.path.join(ref, name)).read()
            bad += a != b
            print('%-22s %s (%d rows)' % (name, 'identical' if a == b else 'DIFFERENT', len(b.splitlines()) - 1))
        return 1 if bad else 0
    finally:
        if os.path.exists(NPZ):
            os.remove(NPZ)


if __name__ == '__main__':
    sys.exit(main())
