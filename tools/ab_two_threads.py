"""Two host threads, each driving its own context (stream, workspaces) over alternate units of the genome step, against the
one-thread pipeline of bench.py -- does one PROCESS reach what two rank processes on one device reach (40.4 vs 44.4 ms)?
    python tools/ab_two_threads.py"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from stripenn_amd import backend as BK

dev = torch.device('cuda', 0)
hb = BK.HipBackend(0)
spec = dict(names=bench.MM10_NAMES, nbins=[-(-s // bench.RESOL) for s in bench.MM10], seeds=list(range(1, 21)), wl='genome')
W = bench._Workload(hb, dev, spec, 1, 0, '', score=True, sigma=2.0)
# a second context with its own view of the same bands
hb2 = BK.HipBackend(0)
hb2.set_background(*[np.zeros((400, 4))] * 4) if False else None
bands2 = {nm: hb2.ctx.band_wrap(W.tens[nm].data_ptr(), W.nbins[W.names.index(nm)], W.hw, keepalive=W.tens[nm]) for nm in W.bands}
# background tables for the second context: recompute through the first workload's objects is costly; reuse by uploading the same arrays
from stripenn_amd import getStripe as GS
sel = bench._DeviceSelector(W.chroms, bench.RESOL)
sizes = np.array([n * bench.RESOL for n in W.nbins], dtype=np.int64)
obj = GS.getStripe(sel, bench.RESOL, 10, 8, 2.0, W.names, W.names, sizes, sizes, 2, 3, 123456789, backend=hb2)
for nm in bands2:
    obj._bands[nm] = bands2[nm]
bg = obj.nulldist()
hb2.set_background(*bg)


def run_units(units, hbk, bands, out):
    nrec = 0
    flight = []
    todo = list(units)[::-1]

    def launch():
        ci, f0, f1 = todo.pop()
        st, en = W.tabs[ci]
        fr = bands[W.names[ci]].frames(st[f0:f1], en[f0:f1])
        flight.append(((ci, f0, f1), fr, fr.stripe_search_begin(W.Ms[ci], sigma=2.0)))
    for _ in range(min(2, len(todo))):
        launch()
    while flight:
        (ci, f0, f1), fr, pend = flight.pop(0)
        recs = pend.wait()
        if todo:
            launch()
        pv, sc = BK.score_inputs(recs, fr.nz, W.tabs[ci][0][f0:f1], W.nbins[ci], W.bs)
        hbk.score(bands[W.names[ci]], W.bs, W.EV[ci], pv, sc)
        nrec += len(recs)
        fr.close()
    out.append(nrec)


def step_two_threads():
    units = W.my_units
    a, b = units[0::2], units[1::2]
    out = []
    t1 = threading.Thread(target=run_units, args=(a, hb, W.bands, out))
    t2 = threading.Thread(target=run_units, args=(b, hb2, bands2, out))
    t1.start(); t2.start(); t1.join(); t2.join()
    return sum(out)


def timed(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize(); W.synchronize(); hb2.ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize(); W.synchronize(); hb2.ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


W.reset_stats(profiling=True); hb2.ctx.set_profiling(True)
only2 = 'only2' in sys.argv            # (for a kernel trace of the two-thread steps alone: tools/overlap_analysis.py)
for rep in range(2):
    ms1, r1 = (0.0, 0) if only2 else timed(lambda: W.step()[0])
    ms2, r2 = timed(step_two_threads)
    print('one thread: %.2f ms per step (%d records); two threads, two contexts: %.2f ms per step (%d records)' % (ms1, r1, ms2, r2), flush=True)
