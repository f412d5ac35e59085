#!/usr/bin/env python3
"""Benchmark of the `stripenn compute` hot path on MI355X (contract: see the task's bench.py section).

A "step" = one pass of the GPU hot path over one chromosome-sized batch already resident in HBM:
frame compaction -> (5 maxpixel levels x 6 brightness levels) image build / Canny / line joining
-> stripe records on the host [-> p-value + Stripiness kernels for the called stripes].
Workload at N=1: BASELINE.json configs[1] (chr16-size 5 kb chromosome, maxpixel sweep 0.95-0.99),
realised synthetically (stripenn_amd/synth.py).  N>1: weak scaling, every rank sweeps its own
chromosome of the same size (chromosome x maxpixel units shard with no collective).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
CHR16_BINS = 19642     # mm10 chr16 (98,207,768 bp) at 5 kb
MAXPIXEL = [0.95, 0.96, 0.97, 0.98, 0.99]
BYTES_PER_IMAGE_PX = {'gray': 12.0, 'canny': 5.0, 'lines': 9.0}  # SURVEY.md 8(d) stages A, B, C-F
SCORE_KERNELS = ('pvalue', 'stripiness')
# HBM bytes per launch of the chain kernels for THIS default workload, from rocprofv3 PMC passes
# (profiles/r01e_pmc.csv: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs, KB units, FETCH_SIZE
# doubled as MI355X_MICROARCH.md prescribes for gfx950).  PMC counters cannot be read inside this script.
PMC_TRAFFIC_BYTES = {'canny': (2 * 922851 + 254623) * 1024.0, 'gray': (2 * 331738 + 1850574) * 1024.0,
                     'lines': (2 * 109698 + 64509) * 1024.0}


def frame_table(nbins):
    nfr = -(-nbins // 200)
    st = np.array([max(0, i * 200 - 100) for i in range(nfr)], dtype=np.int32)
    en = np.minimum((np.arange(nfr) + 1) * 200 + 99, nbins - 1).astype(np.int32)
    return st, en


# ----------------------------------------------------------------------------- CPU baseline
_G = {}


def _cpu_task(args):
    """One (frame, level) through the CPU oracle -- the checker, timed here only as a baseline."""
    from oracle import oracle as O
    fi, M = args
    band, hw, st, en = _G['band'], _G['hw'], _G['st'], _G['en']
    s, e = int(st[fi]), int(en[fi])
    n0 = e - s + 1
    rows = np.arange(s, e + 1)[:, None]
    cols = np.arange(s, e + 1)[None, :]
    D = band[rows, cols - rows + hw].copy()
    D[np.isnan(D)] = 0
    nz = np.where(D.sum(axis=0) != 0)[0]
    if len(nz) <= 10:
        return 0, 0
    D = np.ascontiguousarray(D[np.ix_(nz, nz)])
    recs, tot = O.stripe_search(D, M)
    return len(nz) * len(nz), len(recs)


def _cpu_worker(args):
    """One pool worker: its share of the unit list, one unit after the other, until the deadline."""
    j, cores, deadline = args
    tasks = _G['tasks']
    t0 = time.time()
    px, done, t_last = 0.0, 0, t0
    for k in range(j, len(tasks), cores):
        if time.time() >= deadline:
            break
        r = _cpu_task(tasks[k])
        px += r[0]; done += 1
        t_last = time.time()
    return px, done, t0, t_last


def cpu_baseline(band_h, hw, st, en, Ms, wall_budget_s=12.0):
    """Oracle ("port") on all host cores, time-bounded: every worker walks its share of the units until
    the deadline and returns by itself (no Pool.terminate(), which can dead-lock); the rate is the pixels of
    all completed units over the span from the first start to the last completion."""
    import multiprocessing as mp
    from oracle import oracle as O
    O.build()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:                                    # a container CPU quota below the affinity mask is the real core count
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            cores = max(1, min(cores, -(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    allt = [(fi, M) for fi in range(len(st)) for M in Ms]
    _G.update(band=band_h, hw=hw, st=st, en=en, tasks=allt * 64)
    pool = mp.get_context('fork').Pool(cores)
    try:
        deadline = time.time() + wall_budget_s
        res = pool.map(_cpu_worker, [(j, cores, deadline) for j in range(cores)], chunksize=1)
        pool.close()
        pool.join()
    except BaseException:
        pool.terminate()
        raise
    px = sum(r[0] for r in res); done = sum(r[1] for r in res)
    dt = max(r[3] for r in res) - min(r[2] for r in res)
    return {'value': round(px / dt / 1e6, 2), 'unit': 'contact-Mpx/s', 'cores': cores, 'kind': 'port',
            'sample': '%d (frame,maxpixel) units of the same chromosome (%.2fx one step), StripeSearch chain only, '
                      'oracle/stripe_oracle.c via a fork pool on %d host cores, %.1f s wall'
                      % (done, done / len(allt), cores, dt)}


# ----------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--bins', type=int, default=CHR16_BINS, help='chromosome length in 5 kb bins')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-score', action='store_true', help='StripeSearch chain only (skips the p-value / Stripiness set-up; for very long chromosomes)')
    args = ap.parse_args()

    # the default run takes about a minute; never hang the driver: dump the stacks and exit after 20 min
    import faulthandler
    faulthandler.dump_traceback_later(1200, exit=True)

    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=torch.device('cuda', local_rank))
    torch.cuda.set_device(local_rank)

    from stripenn_amd import synth, hip
    hw = 512
    nb = args.bins
    chrom = synth.SynthChrom(nb, 16 + rank)
    band_h = chrom.band(hw)
    # (HipBackend below raises if the HIP extension / GPU is missing: no CPU fallback; the band is
    #  resident in HBM before the timed region)
    st, en = frame_table(nb)
    # maxpixel quantiles: the reference's getQuantile step stays on the host and is
    # outside the hot path; on band-limited synthetic data the band holds every positive pixel.
    Ms = np.quantile(band_h[band_h > 0], MAXPIXEL)

    # score-path inputs (untimed set-up): expected values and background tables of this chromosome,
    # computed through the same facade the CLI uses
    from stripenn_amd import getStripe as GS, backend as BK
    name = 'chr16'
    sel = synth.SynthSelector({name: chrom}, 5000)
    size = nb * 5000
    hb = BK.HipBackend(local_rank)
    obj = GS.getStripe(sel, 5000, 10, 8, 2.0, [name], [name], np.array([size]), np.array([size]), 2, 3, 123456789,
                       backend=hb)
    obj._bands[name] = hb.ctx.band_upload(band_h)
    sband = obj._bands[name]
    if not args.no_score:
        EV = np.asarray(obj.mpmean()[name])
        bg = obj.nulldist()
        hb.set_background(*bg)
    bs = 10

    def score_inputs(recs, fr):
        """bin rectangles of every candidate stripe (vectorised host arithmetic)"""
        f = recs['frame']
        base = st[f]
        nzr = fr.nz.ravel(); fo = f * fr.nz.shape[1]
        x0 = base + nzr.take(fo + recs['x']); x1 = base + nzr.take(fo + recs['x'] + recs['w'] - 1)
        y0 = base + nzr.take(fo + recs['y']); y1 = base + nzr.take(fo + recs['y'] + recs['h'] - 1)
        n = len(recs)
        pv = np.zeros(n, dtype=BK.PV_STRIPE_DTYPE)
        pv['row0'], pv['row1'] = y0, y1 + 1
        pv['col0'], pv['col1'] = np.maximum(x0 - bs, 0), np.minimum(x1 + 1 + bs, nb)
        pv['mode'] = np.where(x0 == y0, 0, 1)
        pv['upbase'] = y1 + 1 - y0
        sc = np.zeros(n, dtype=BK.SCORE_STRIPE_DTYPE)
        sc['row0'], sc['row1'] = y0, y1 + 1
        lm = np.minimum(np.maximum(x0 - bs, 1), x0); rm = np.minimum(x1 + 1 + bs, nb - 1)
        sc['col0'][:, 0], sc['col1'][:, 0] = x0, x1 + 1
        sc['col0'][:, 1], sc['col1'][:, 1] = lm, x0
        sc['col0'][:, 2], sc['col1'][:, 2] = x1 + 1, np.maximum(rm, x1 + 1)
        sc['ex0'][:, 0], sc['ex0'][:, 1], sc['ex0'][:, 2] = x0, lm, x1 + 2
        sc['ey0'] = y0
        sc['mirror'] = np.where(x0 == y0, 0, 1)
        sc['mcol0'], sc['mcol1'], sc['mrow0'], sc['mrow1'] = 1, 0, 1, 0
        return pv, sc

    def step():
        fr = sband.frames(st, en)
        recs = fr.stripe_search(Ms)
        if not args.no_score:
            pv, sc = score_inputs(recs, fr)
            p = hb.pvalue(sband, bs, pv)
            g = hb.stripiness(sband, EV, sc)[0]
        S = fr.S.copy()
        fr.close()
        return recs, S

    ctx = hb.ctx

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.set_profiling(True)
    ctx.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        recs, S = step()
    barrier()
    dt = time.perf_counter() - t0
    stats = ctx.stats()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    contact_px = float((S.astype(np.float64) ** 2).sum()) * len(Ms)   # this rank, per step
    image_px = contact_px * 6
    total_px = contact_px                                              # all ranks (each has its own chromosome)
    if world > 1:
        tsum = torch.tensor([contact_px], dtype=torch.float64, device='cuda')
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        total_px = float(tsum.item())
    value = total_px * args.steps / dt / 1e6

    out = None
    if rank == 0:
        chain_ms = stats['chain_wall']['ms'] if 'chain_wall' in stats else sum(v['ms'] for k, v in stats.items() if k in BYTES_PER_IMAGE_PX)
        dom = max((k for k in stats if k in BYTES_PER_IMAGE_PX or k in SCORE_KERNELS), key=lambda k: stats[k]['ms'])
        d = stats[dom]
        ach = d['alg_bytes'] / d['launches'] / (d['ms'] / d['launches'] * 1e-3) / 1e9
        roof = {'bound': 'hbm', 'kernel': 'k_' + dom, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(ach / HBM_PEAK_GBS, 4),
                'traffic': (PMC_TRAFFIC_BYTES.get(dom) if nb == CHR16_BINS else None),
                'avg_launch_ms': round(d['ms'] / d['launches'], 4),
                'alg_bytes_per_launch': d['alg_bytes'] / d['launches'],
                'chain': {'kernels_ms_per_step': {k: round(v['ms'] / args.steps, 4) for k, v in stats.items()},
                          'alg_bytes_per_image_px': 26.0,
                          'achieved_GBs': round(26.0 * image_px * args.steps / (chain_ms * 1e-3) / 1e9, 1),
                          'frac': round(26.0 * image_px * args.steps / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
        out = {'metric': 'contact-matrix Mpixels/s through compute path', 'value': round(value, 2),
               'unit': 'contact-Mpx/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
               'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
               'config': {'workload': '%s (%d bins, %d frames), maxpixel sweep 0.95-0.99 x 6 brightness levels: frame '
                                      'compaction + StripeSearch chain%s'
                                      % ('configs[1]: chr16-size 5kb chromosome' if nb == CHR16_BINS else
                                         'configs[4]-like 1kb chr1-size band' if nb > 200000 else 'custom chromosome',
                                         nb, len(st),
                                         '' if args.no_score else ' + p-value and Stripiness of every candidate stripe'),
                          'frames': int(len(st)), 'levels': len(Ms), 'images_per_step': int(len(st) * len(Ms) * 6),
                          'contact_px_per_step': total_px, 'stripe_records': int(len(recs)),
                          'sharding': 'one chromosome per rank, no collective'},
               'roofline': roof}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(band_h, hw, st, en, [float(m) for m in Ms])
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    hb.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
