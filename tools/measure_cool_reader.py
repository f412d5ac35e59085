"""The real-file route at genome size (h5py interpreter, no GPU): writes a cooler-schema .mcool of the benchmark genome
(mm10 sizes at 5 kb, the synthetic pixel function of stripenn_amd/synth.py, weight column, gzip-compressed pixel columns
as cooler writes them), chromosome by chromosome, then reads every chromosome's cis pixels through pixels.CoolTable the
way the driver does (prefetch of the next chromosome on a host thread) and reports read rate and peak RSS.

    /opt/conda/bin/python3.9 tools/measure_cool_reader.py /tmp/mm10_synth.mcool [scale]

`scale` < 1 shrinks every chromosome (a quick run)."""
import os
import resource
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import h5py                                          # noqa: E402
from stripenn_amd import pixels, synth               # noqa: E402

MM10 = [195471971, 182113224, 160039680, 156508116, 151834684, 149736546, 145441459, 129401213, 124595110, 130694993,
        122082543, 120129022, 120421639, 124902244, 104043685, 98207768, 94987271, 90702639, 61431566, 171031299]
NAMES = ['chr%d' % (i + 1) for i in range(19)] + ['chrX']
RESOL, GROUP = 5000, 'resolutions/5000'


def rss_mb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


def write(path, scale):
    nbins = [max(400, int(-(-s // RESOL) * scale)) for s in MM10]
    sizes = np.array([n * RESOL for n in nbins], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(nbins)]).astype(np.int64)
    t0 = time.time()
    with h5py.File(path, 'w') as f:
        g = f.create_group(GROUP)
        g.attrs['bin-size'] = RESOL
        g.create_dataset('chroms/name', data=np.array(NAMES, dtype='S'))
        g.create_dataset('chroms/length', data=sizes)
        g.create_dataset('bins/chrom', data=np.repeat(np.arange(len(NAMES)), nbins))
        start = np.concatenate([np.arange(n) * RESOL for n in nbins])
        g.create_dataset('bins/start', data=start)
        g.create_dataset('bins/end', data=start + RESOL)
        kw = dict(maxshape=(None,), chunks=(65536,), compression='gzip', compression_opts=4, shuffle=True)
        d1 = g.create_dataset('pixels/bin1_id', shape=(0,), dtype=np.int64, **kw)
        d2 = g.create_dataset('pixels/bin2_id', shape=(0,), dtype=np.int64, **kw)
        dc = g.create_dataset('pixels/count', shape=(0,), dtype=np.int32, **kw)
        weights, b1off, n = [], [], 0
        for k, nm in enumerate(NAMES):
            ch = synth.SynthChrom(nbins[k], k + 1)
            t = pixels.PixelTable.from_synth([nm], {nm: ch}, RESOL)
            m = len(t.count)
            for d, v in ((d1, t.bin1_id + off[k]), (d2, t.bin2_id + off[k]), (dc, t.count)):
                d.resize((n + m,))
                d[n:n + m] = v
            b1off.append(np.searchsorted(t.bin1_id, np.arange(nbins[k]), side='left') + n)
            weights.append(t.weights['weight'])
            n += m
            print('  wrote %s: %d bins, %d pixels (%.0f s, rss %.0f MB)' % (nm, nbins[k], m, time.time() - t0, rss_mb()), flush=True)
            del t, ch
        g.create_dataset('bins/weight', data=np.concatenate(weights))
        g.create_dataset('indexes/chrom_offset', data=off)
        g.create_dataset('indexes/bin1_offset', data=np.concatenate(b1off + [[n]]))
    return n, nbins


def read(path):
    t0 = time.time()
    lazy = pixels.CoolTable(path, GROUP)
    base = rss_mb()
    tot, largest = 0, 0
    lazy.prefetch(NAMES[0])
    for k, nm in enumerate(NAMES):
        if k + 1 < len(NAMES):
            lazy.prefetch(NAMES[k + 1])               # as stripenn.compute: the next chromosome is read ahead
        a = lazy.chrom_pixels(nm)
        m = len(a[2])
        tot += m
        largest = max(largest, m)
        del a
    dt = time.time() - t0
    lazy.close()
    return tot, largest, dt, base


if __name__ == '__main__':
    path = sys.argv[1]
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    if not os.path.exists(path):
        n, nbins = write(path, scale)
        print('written: %d pixels, %d bins, file %.2f GB' % (n, sum(nbins), os.path.getsize(path) / 1e9))
    pid = os.fork()                                  # the reader's peak RSS in a process of its own
    if pid == 0:
        tot, largest, dt, base = read(path)
        peak = rss_mb()
        cols = largest * 20 / 1e6                    # bin1 + bin2 (int64) + count (int32) of the largest chromosome, MB
        print('read: %d cis pixels in %.1f s = %.1f Mpixel/s = %.0f MB/s of columns; file %.2f GB; peak RSS %.0f MB '
              '(%.0f MB before the first read); largest chromosome %d pixels = %.0f MB of columns; peak / largest = %.2f'
              % (tot, dt, tot / dt / 1e6, tot * 20 / dt / 1e6, os.path.getsize(path) / 1e9, peak, base, largest, cols,
                 (peak - base) / cols), flush=True)
        os._exit(0)
    os.waitpid(pid, 0)
