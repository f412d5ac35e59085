#!/usr/bin/env python3
"""Benchmark of the `stripenn compute` hot path on MI355X (contract: see the task's bench.py section).

Workload (default, `--workload genome`): BASELINE.json's metric configuration -- a 5 kb WHOLE GENOME:
the 20 mm10 chromosome sizes (1-19, X; 526 765 bins, 2 645 frames) x 5 maxpixel levels x 6 brightness
levels = 79 350 images, 2.1 G contact-px per step, realised synthetically (stripenn_amd/synth.py's pixel
function, evaluated on the device by stripenn_amd/synth_device.py).  All bands are resident in HBM before
the timed region.

A "step" = one pass of the GPU hot path over the rank's share of the genome: per chromosome (or frame span
of one) frame compaction + medpixel -> image build / Canny / line joining for every (frame, level,
brightness) -> candidate records on the host -> p-value + Stripiness kernels for every candidate.  The K timed steps run as
ONE pipeline between the two barriers (the first units of step s + 1 are launched while the last units of step s are collected
and scored; every record of every step is collected, every candidate scored inside the timed region); the line also carries the
figure with every step drained before the next one starts (config.drained_ms_per_step; STP_BENCH_PIPELINE_STEPS=0 times that).

N > 1 (`--gpus N`): STRONG scaling of the same genome.  The (chromosome x frame) grid is cut into N
contiguous spans of equal frame count (stripenn_amd.shard.frame_spans; all maxpixel levels of a frame stay
together), one process per GPU, no collective on the data path; the process group only carries the barrier
and the max-over-ranks time.  Launch either through torch.distributed.run (RANK / WORLD_SIZE in the
environment) or plainly as `python bench.py --gpus N`: then this process starts the N ranks itself, as
children, before it has touched the GPU.

Other workloads (extra lines, not the metric): `--workload chr16` (configs[1]: one chr16-size chromosome,
the r01 bench line), `--bins N` (one chromosome of N 5 kb-bins, e.g. 248957 for the configs[4]-like band).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
N_SIMD = 1024          # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9       # max clock
VALU_CYCLES_PER_INST = 4.25   # measured: v_fma_f32 / packed f32 / every f64 op per wave and SIMD (f32 add, sub, mul: 2.2)
MM10 = [195471971, 182113224, 160039680, 156508116, 151834684, 149736546, 145441459, 129401213, 124595110, 130694993,
        122082543, 120129022, 120421639, 124902244, 104043685, 98207768, 94987271, 90702639, 61431566, 171031299]
MM10_NAMES = ['chr%d' % (i + 1) for i in range(19)] + ['chrX']
CHR16_BINS = 19642     # mm10 chr16 (98,207,768 bp) at 5 kb
RESOL = 5000
MAXPIXEL = [0.95, 0.96, 0.97, 0.98, 0.99]
BYTES_PER_IMAGE_PX = {'gray': 12.0, 'canny': 5.0, 'lines': 9.0}  # SURVEY.md 8(d) stages A, B, C-F
SCORE_KERNELS = ('pvalue', 'stripiness', 'score', 'pvalue_block', 'stripiness_block')
PMC_FILE = os.path.join(ROOT, 'profiles', 'pmc_current.json')    # written by tools/summarize_profile.py


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


def cpu_baseline(band_h, hw, st, en, Ms, what, wall_budget_s=12.0):
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
            'sample': '%d (frame,maxpixel) units of %s, StripeSearch chain only (the GPU step also scores every '
                      'candidate stripe), oracle/stripe_oracle.c via a fork pool on %d host cores, %.1f s wall'
                      % (done, what, cores, dt)}


# ----------------------------------------------------------------------------- rank launcher
def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (this process has not
    touched the GPU and never will), relay rank 0's output, return the worst exit code."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # poll: when one rank dies the others would wait for it in the next collective until their own time-out, so the
    # rest is terminated (never re-launched) and the run reports the failure at once
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            rc = max(rc, abs(r))
            if r != 0:
                for q in live:
                    q.terminate()
    return rc


# ----------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', choices=('genome', 'chr16'), default='genome')
    ap.add_argument('--bins', type=int, default=0, help='one chromosome of this many 5 kb bins instead of a named workload')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-e2e', action='store_true', help='skip the end-to-end `compute` run (quantile -> TSVs) at N = 1')
    ap.add_argument('--no-score', action='store_true', help='StripeSearch chain only (for very long chromosomes)')
    ap.add_argument('--canny', type=float, default=2.0, help='Canny sigma (2.0 = the reference default; `score` passes 2.5)')
    ap.add_argument('--no-extras', action='store_true', help='skip the extra measurements of the N = 1 line: the all-f64 kernels on the same workload and the 1 kb chr1-size band')
    ap.add_argument('--allow-stp-lib', action='store_true', help='accept a library named by STP_LIB (profiling builds)')
    ap.add_argument('--emulate-rank', default='', help='R/N: time the share rank R of an N-rank run would get, alone on this GPU (diagnostic: per-rank fixed costs without an N-GPU node; the line is NOT a multi-GPU measurement)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get('STP_LIB') and not args.allow_stp_lib:
        sys.exit('bench.py: STP_LIB=%s is set; the benchmark measures the product library only '
                 '(pass --allow-stp-lib for an ablation build)' % os.environ['STP_LIB'])

    # never hang the driver: dump the stacks and exit after 20 min
    import faulthandler
    faulthandler.dump_traceback_later(1200, exit=True)

    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # STP_BENCH_REHEARSE=1: every rank on device 0 with gloo (a one-GPU box cannot host an RCCL group of 2)
    rehearse = os.environ.get('STP_BENCH_REHEARSE') == '1'
    rehearse_reduce = False
    if rehearse:
        local_rank = 0
    comm, comm_note, pg = 'none', 'single process', None
    if world > 1:
        import torch.distributed as dist
        import datetime
        import threading
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # The path has no data collective; the group only carries the barrier and the max-over-ranks time.  A gloo group
        # comes up first and is the control plane; RCCL is then probed on a group of its own (one all_reduce under a
        # watchdog) and every rank reports its outcome over gloo, so the choice is made COLLECTIVELY: RCCL carries the
        # timing scalars iff it worked on every rank, gloo otherwise -- and the JSON line says which (`config.comm`).
        dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
        ok, note = 0, 'rehearsal on one device: gloo only'
        if not rehearse:
            torch.cuda.set_device(local_rank)
            box = {}

            def _probe():
                try:
                    torch.cuda.set_device(local_rank)          # (the current device is a per-thread setting)
                    g = dist.new_group(backend='nccl', timeout=datetime.timedelta(seconds=120))
                    t = torch.ones(1, device=torch.device('cuda', local_rank))
                    dist.all_reduce(t, group=g)
                    torch.cuda.synchronize()
                    box['pg'], box['sum'] = g, float(t.item())
                except Exception as e:      # noqa: BLE001
                    box['err'] = str(e)[:160]
            th = threading.Thread(target=_probe, daemon=True)
            th.start()
            th.join(90)
            if th.is_alive():
                note = 'RCCL probe timed out on rank %d' % rank
            elif 'err' in box:
                note = 'RCCL probe failed on rank %d: %s' % (rank, box['err'])
            elif int(box.get('sum', 0)) != world:
                note = 'RCCL probe summed %s over %d ranks' % (box.get('sum'), world)
            else:
                ok, note, pg = 1, 'RCCL all_reduce over %d ranks' % world, box['pg']
        flags = [None] * world
        dist.all_gather_object(flags, (ok, note))
        if all(f[0] for f in flags):
            comm, comm_note = 'nccl', note
        else:
            comm, pg = 'gloo', None
            comm_note = '; '.join(sorted({f[1] for f in flags if not f[0]}))
            if rank == 0:
                print('bench.py: timing scalars over gloo (%s)' % comm_note, file=sys.stderr, flush=True)
        rehearse_reduce = comm == 'gloo'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    rdev = 'cpu' if (rehearse or rehearse_reduce) else 'cuda'      # where the timing scalars are reduced

    from stripenn_amd import hip, shard, backend as BK
    if args.bins:
        spec = dict(names=['chr16'], nbins=[args.bins], seeds=[16],
                    wl='configs[4]-like 1kb chr1-size band' if args.bins > 200000 else 'custom chromosome')
    elif args.workload == 'chr16':
        spec = dict(names=['chr16'], nbins=[CHR16_BINS], seeds=[16], wl='configs[1]: chr16-size 5kb chromosome')
    else:
        spec = dict(names=MM10_NAMES, nbins=[-(-s // RESOL) for s in MM10], seeds=list(range(1, 21)),
                    wl='configs[2]: mm10-size whole genome at 5kb (20 chromosomes)')
    t_setup = time.time()
    hb = BK.HipBackend(local_rank)       # raises if the HIP extension / GPU is missing: no CPU fallback
    ctx = hb.ctx
    W = _Workload(hb, dev, spec, world, rank, args.emulate_rank, score=not args.no_score, sigma=args.canny)
    setup_s = time.time() - t_setup
    names, nbins, nframes = W.names, W.nbins, W.nframes

    def barrier():
        if world > 1:
            dist.barrier(group=pg)        # pg: the RCCL group when every rank's probe succeeded, else None = gloo
        torch.cuda.synchronize()
        W.synchronize()

    for _ in range(args.warmup):
        W.step()
    W.reset_stats()
    barrier()
    t0 = time.perf_counter()
    # The K timed steps are ONE pipeline (bracketed by the barriers, as the contract says): the first units of step s + 1 are
    # launched while the last units of step s are collected and scored -- what a training loop's asynchronous launches do, and
    # what a study of several samples through one process does.  Every record of every step is collected and every candidate
    # scored inside the timed region.  STP_BENCH_PIPELINE_STEPS=0 drains every step before the next one starts (rounds 1-5 and
    # the first half of round 6 reported that; the N = 1 line still carries it: config.drained_ms_per_step): 43.7 against 42.5 ms
    # for the whole genome, 6.2-6.6 against 5.3-5.75 ms for a 1/8 share (profiles/r06_ab_pipelined_steps.txt).
    pipelined = os.environ.get('STP_BENCH_PIPELINE_STEPS', '1') != '0'
    if pipelined:
        nrec, contact_px = W.run(args.steps)
    else:
        for _ in range(args.steps):                              # every step drains before the next one starts
            nrec, contact_px = W.step()
    barrier()
    dt_rank = time.perf_counter() - t0
    stats = W.stats()
    host_wait_ms = W.host_wait_s / (args.steps + args.warmup) * 1e3      # (the warm-up steps count too: same work)
    host_call_ms = W.host_call_s / (args.steps + args.warmup) * 1e3
    drained_ms = None
    if world == 1 and pipelined:                    # the same steps, each drained before the next one starts (what earlier rounds reported)
        drained_ms = round(_timed_steps(W, barrier, min(args.steps, 10))[0], 3)
    dt, total_px, total_rec, rank_ms = dt_rank, contact_px, nrec, [dt_rank / args.steps * 1e3]
    if world > 1:
        tmax = torch.tensor([dt_rank], dtype=torch.float64, device=rdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=pg)
        dt = float(tmax.item())
        tsum = torch.tensor([contact_px, float(nrec)], dtype=torch.float64, device=rdev)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM, group=pg)
        total_px, total_rec = float(tsum[0].item()), int(tsum[1].item())
        allt = [torch.zeros(1, dtype=torch.float64, device=rdev) for _ in range(world)]
        dist.all_gather(allt, torch.tensor([dt_rank / args.steps * 1e3], dtype=torch.float64, device=rdev), group=pg)
        rank_ms = [float(t.item()) for t in allt]
    value = total_px * args.steps / dt / 1e6
    devnames = [torch.cuda.get_device_name(local_rank)]
    if world > 1:
        devnames = [None] * world
        dist.all_gather_object(devnames, '%d:%s' % (local_rank, torch.cuda.get_device_name(local_rank)))

    if rank == 0:
        image_px = contact_px * 6                                   # this rank, per step
        chain_ms = stats['chain_wall']['ms'] if 'chain_wall' in stats else sum(v['ms'] for k, v in stats.items() if k in BYTES_PER_IMAGE_PX)
        dom = max((k for k in stats if k in BYTES_PER_IMAGE_PX or k in SCORE_KERNELS), key=lambda k: stats[k]['ms'])
        d = stats[dom]
        ach = d['alg_bytes'] / (d['ms'] * 1e-3) / 1e9
        pmc = _load_pmc('genome' if not args.bins and args.workload == 'genome' else
                        'chr16' if not args.bins else 'bins%d' % args.bins)
        traffic, valu = None, None
        # counter-derived figures only when the committed counters were collected on the sources of THIS library
        pmc_fresh = bool(pmc) and pmc.get('src_sha') == hip.source_hash() and abs(args.canny - 2.0) < 1e-9     # (the counters are those of sigma 2.0: k_canny_f32<8>)
        if pmc_fresh and dom in pmc.get('kernels', {}):
            k = pmc['kernels'][dom]
            if k.get('fetch_kb') is not None and k.get('write_kb') is not None:
                # per launch of the dominant kernel: FETCH_SIZE doubled (gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE
                traffic = (2.0 * k['fetch_kb'] + k['write_kb']) * 1024.0
            if k.get('valu_insts') is not None and k.get('image_px'):
                per_px = k['valu_insts'] / k['image_px']
                step_insts = per_px * image_px                      # wave-instructions of this kernel per step
                # issue ceiling of the kernel's arithmetic, measured on this chip (tools/ubench_valu.hip, profiles/r04_ubench_valu.txt):
                # a v_fma_f32, a packed f32 op and every f64 op hold a SIMD ~4.25 cycles per wave; f32 add / sub / mul ~2.2
                cyc = VALU_CYCLES_PER_INST
                valu = {'kernel': 'k_' + dom, 'wave_insts_per_image_px': round(per_px, 3), 'lane_insts_per_image_px': round(per_px * 64, 1),
                        'cycles_per_inst': cyc, 'peak_wave_insts_per_s': N_SIMD * CLOCK_HZ / cyc,
                        'achieved_wave_insts_per_s': round(step_insts / (d['ms'] / args.steps * 1e-3), 0),
                        'frac': round(step_insts / (d['ms'] / args.steps * 1e-3) / (N_SIMD * CLOCK_HZ / cyc), 4),
                        'source': pmc.get('tag')}
        # every kernel of the step under SURVEY 8(d)'s algorithmic-byte formulas (the library's own byte counters), per launch
        # Beside each contract fraction (`frac`: SURVEY 8(d)'s numerator, which a kernel that keeps its data on chip or
        # searches instead of streaming can undercut -- then `frac` exceeds what the hardware moved and is marked
        # `model_exceeds_work`, not roofline evidence) the bytes the memory system really moved: `traffic_frac` =
        # (2 x FETCH_SIZE + WRITE_SIZE) per launch from the committed counters / the launch time measured here / HBM peak.
        per_kernel = {}
        for k, v in stats.items():
            if k == 'chain_wall' or not v['launches'] or v['ms'] <= 0:
                continue
            g = v['alg_bytes'] / (v['ms'] * 1e-3) / 1e9
            e = {'ms_per_step': round(v['ms'] / args.steps, 3), 'avg_launch_ms': round(v['ms'] / v['launches'], 4),
                 'alg_GB_per_step': round(v['alg_bytes'] / args.steps / 1e9, 3), 'achieved_GBs': round(g, 1),
                 'frac': round(g / HBM_PEAK_GBS, 4)}
            pk = pmc.get('kernels', {}).get(k) if pmc_fresh else None
            if pk and pk.get('fetch_kb') is not None and pk.get('write_kb') is not None:
                tb = (2.0 * pk['fetch_kb'] + pk['write_kb']) * 1024.0           # bytes per launch
                tg = tb / (v['ms'] / v['launches'] * 1e-3) / 1e9
                e.update(traffic_GB_per_launch=round(tb / 1e9, 4), traffic_GBs=round(tg, 1), traffic_frac=round(tg / HBM_PEAK_GBS, 4),
                         alg_over_traffic=round(v['alg_bytes'] / v['launches'] / tb, 2) if tb > 0 else None)
            else:
                e.update(traffic_frac=None)
            if g / HBM_PEAK_GBS > 1.0:
                e['model_exceeds_work'] = True
            per_kernel['k_' + k] = e
        roof = {'bound': 'hbm', 'kernel': 'k_' + dom, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': traffic,
                'traffic_source': (pmc.get('tag') if traffic is not None else
                                   ('stale: %s was collected on other kernel sources' % pmc.get('tag')) if pmc and not pmc_fresh else None),
                # what the dominant kernel runs into: its issue rate or HBM when one of them is above half its peak, else
                # neither (latency / synchronisation: DESIGN.md section 3)
                'limiter': ('valu_issue' if valu and valu['frac'] >= 0.5 and valu['frac'] > ach / HBM_PEAK_GBS else
                            'hbm' if ach / HBM_PEAK_GBS >= 0.5 or not valu else 'latency'),
                'avg_launch_ms': round(d['ms'] / d['launches'], 4), 'launches_per_step': d['launches'] / args.steps,
                'alg_bytes_per_launch': d['alg_bytes'] / d['launches'],
                'valu_issue': valu,
                'kernels': per_kernel,
                'chain': {'kernels_ms_per_step': {k: round(v['ms'] / args.steps, 4) for k, v in stats.items()},
                          'alg_bytes_per_image_px': 26.0,
                          'achieved_GBs': round(26.0 * image_px * args.steps / (chain_ms * 1e-3) / 1e9, 1),
                          'frac': round(26.0 * image_px * args.steps / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          # (26 B per image-px of EVERY image pixel: with the tiles below the diagonal, the shared blocks and the unread
                          #  grey tiles not computed the contract figure approaches and can pass 1 -- it is then marked, like the
                          #  per-kernel ones, and is not roofline evidence; `traffic_frac` is what the memory system moved)
                          'model_exceeds_work': bool(26.0 * image_px * args.steps / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS > 1.0),
                          # the three chain kernels' counter bytes per step over the chain's wall time
                          'traffic_frac': (round(sum(per_kernel['k_' + k]['traffic_GB_per_launch'] * stats[k]['launches'] for k in BYTES_PER_IMAGE_PX)
                                                 / (chain_ms * 1e-3) / HBM_PEAK_GBS, 4)
                                           if all(per_kernel.get('k_' + k, {}).get('traffic_GB_per_launch') is not None for k in BYTES_PER_IMAGE_PX) else None)}}
        canny_exact_env, gray_exact_env = os.environ.get('STP_CANNY') == 'exact', os.environ.get('STP_GRAY') == 'exact'
        exact_env = canny_exact_env or gray_exact_env       # (any non-default kernel selection: no `exact` companion, no stale counters)
        out = {'metric': 'contact-matrix Mpixels/s through compute path', 'value': round(value, 2),
               'unit': 'contact-Mpx/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
               'scaling': 'strong', 'vs_baseline': None,
               'dtype': 'f64 (%s; %s)' % ('Canny: every intermediate f64' if canny_exact_env else 'Canny classes: certified f32 + f64 resolver',
                                          'grey: every operation of the reference' if gray_exact_env else 'grey: certified f64 shortcut + exact redo'),
               'data': 'synthetic',
               'emulated_rank': args.emulate_rank or None,
               'config': {'workload': '%s: %d bins, %d frames x %d maxpixel levels (0.95-0.99) x 6 brightness levels; step = frame '
                                      'compaction + medpixel + StripeSearch chain%s, bands resident in HBM'
                                      % (spec['wl'], sum(nbins), sum(nframes), len(MAXPIXEL),
                                         '' if args.no_score else ' + p-value and Stripiness of every candidate stripe'),
                          'chromosomes': len(names), 'frames': int(sum(nframes)), 'levels': len(MAXPIXEL), 'canny_sigma': args.canny,
                          'images_per_step': int(sum(nframes) * len(MAXPIXEL) * 6),
                          'contact_px_per_step': total_px, 'stripe_records_per_step': int(total_rec),
                          'sharding': 'contiguous (chromosome x frame) spans of equal frame count, one process per GPU, '
                                      'no collective on the data path',
                          'rank_ms_per_step': [round(t, 3) for t in rank_ms], 'setup_s': round(setup_s, 1),
                          'host_wait_ms_per_step': round(host_wait_ms, 2),      # of ms_per_step the search thread spent waiting for searches ...
                          'host_blocked_ms_per_step': round(host_wait_ms + host_call_ms, 2),   # ... and inside every blocking device call (searches, frame preparation, p-value, Stripiness); the rest is Python / numpy work (score inputs, bookkeeping)
                          'score_thread': W._thr is not None,                    # p-value / Stripiness calls on a host thread and context of their own
                          'search_contexts': 2 if W.two_ctx else 1,              # contexts (streams, workspaces) taking alternate units
                          'steps_pipelined': pipelined,                          # the K timed steps run as one pipeline (no drain between two steps)
                          'drained_ms_per_step': drained_ms,                     # ... and the same steps with every step drained
                          'comm': comm, 'comm_note': comm_note, 'devices': devnames,
                          'library': hip.LIB_PATH,
                          'arithmetic': ('line joining, scoring: f64 as the reference; grey images: '
                                         + ('every operation of the reference (STP_GRAY=exact)' if os.environ.get('STP_GRAY') == 'exact' else
                                            'f64 from shared row sums, certified against float rounding boundaries, flagged lanes redone in the '
                                            'reference\'s operations (k_gray_c3) -- bit-identical grey images')
                                         + '; Canny classes: '
                                         + ('f64 throughout (STP_CANNY=exact)' if canny_exact_env else
                                            'f32 with a proven error budget, the undecidable pixels in the reference\'s f64 '
                                            '(k_canny_f32) -- class maps identical to the f64 kernel and the oracle'))},
               'roofline': roof}
        if world == 1 and not args.no_extras and not exact_env:
            # the same workload with every intermediate in the reference's arithmetic (k_gray<1>, k_canny_pipe): 3 steps
            import hashlib
            h_def = hashlib.sha256()
            W.step(digest=h_def)                      # the record buffers of one (untimed) step of the shipped kernels, unit by unit
            saved_env = {k: os.environ.get(k) for k in ('STP_CANNY', 'STP_GRAY')}
            os.environ['STP_CANNY'] = 'exact'; os.environ['STP_GRAY'] = 'exact'
            try:
                h_ex = hashlib.sha256()
                W.step(digest=h_ex)                   # ... and of the all-f64 kernels (also their warm-up)
                W.reset_stats(); barrier()
                t0 = time.perf_counter()
                for _ in range(3):
                    _, px_e = W.step()
                barrier()
                dte = time.perf_counter() - t0
                se = W.stats()
                out['exact'] = {'what': 'the same workload with STP_CANNY=exact STP_GRAY=exact (every intermediate in the reference\'s f64 '
                                        'operations), 3 steps; records_equal: sha256 of every unit\'s record buffer of one step, shipped '
                                        'kernels vs these',
                                'records_equal': h_def.hexdigest() == h_ex.hexdigest(),
                                'records_sha256': {'shipped': h_def.hexdigest()[:16], 'exact': h_ex.hexdigest()[:16]},
                                'value': round(px_e * 3 / dte / 1e6, 2), 'unit': 'contact-Mpx/s', 'ms_per_step': round(dte / 3 * 1e3, 3),
                                'canny_ms_per_launch': round(se['canny']['ms'] / se['canny']['launches'], 4),
                                'gray_ms_per_launch': round(se['gray']['ms'] / se['gray']['launches'], 4),
                                'kernels_ms_per_step': {k: round(v['ms'] / 3, 3) for k, v in se.items()}}
            finally:
                for k, v in saved_env.items():        # (a user-set selection stays in force for the measurements below)
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        if world == 1 and not args.no_extras and not exact_env and not args.emulate_rank:
            out['ab_round6'] = ab_round6(W, barrier)
            if not args.bins and args.workload == 'genome':
                out['emulated_shares_ms'] = emulated_shares(W, barrier, 8, pipelined=pipelined)
                if pipelined:
                    out['emulated_shares_drained_ms'] = emulated_shares(W, barrier, 8, steps=5, pipelined=False)
        if world == 1 and not args.no_extras and not exact_env and not args.emulate_rank and not args.bins and args.workload == 'genome':
            try:
                out['regimes'] = regimes_extra(hb, dev, args.canny, barrier)
            except Exception as e:      # noqa: BLE001 -- an extra line must not cost the metric line
                out['regimes'] = {'error': str(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            ci = min({u[0] for u in W.my_units}, key=lambda c: nbins[c])        # the smallest chromosome held here
            band_h = W.bands[names[ci]].download()
            st, en = W.tabs[ci]
            out['cpu_baseline'] = cpu_baseline(band_h, W.hw, st, en, [float(m) for m in W.Ms[ci]],
                                               '%s (%d bins, %d frames) of the same genome' % (names[ci], nbins[ci], len(st)))
            del band_h
            out['cpu_baseline']['reference_python'] = reference_python_figure()
        else:
            out['cpu_baseline'] = None
        if world == 1 and not args.no_e2e and not args.no_score:
            out['e2e_compute'] = e2e_compute(names, W.chroms, total_px, hb)
        if world == 1 and not args.no_extras and not args.bins and args.workload == 'genome' and not exact_env:
            # configs[4]: the 1 kb chr1-size band (248 957 bins, 1 245 frames x 5 levels x 6 images), chain only, shipped kernels
            W.release()
            try:
                spec4 = dict(names=['chr1_1kb'], nbins=[248957], seeds=[5], wl='configs[4]: 1kb chr1-size band')
                W4 = _Workload(hb, dev, spec4, 1, 0, '', score=False, sigma=args.canny)
                W4.step(); W4.step()                  # (two untimed steps: the genome workload's buffers were released a moment ago;
                ctx.reset_stats(); barrier()          #  one run in round 5 saw the first timed step at 1.5x)
                t0 = time.perf_counter()
                for _ in range(5):
                    _, px4 = W4.step()
                barrier()
                dt4 = time.perf_counter() - t0
                s4 = ctx.stats()
                cw = s4['chain_wall']['ms'] if 'chain_wall' in s4 else sum(v['ms'] for k, v in s4.items() if k in BYTES_PER_IMAGE_PX)
                out['band_1kb'] = {'what': 'configs[4]: synthetic 1 kb chr1-size band (248 957 bins, 1 245 frames x 5 levels x 6 images), '
                                           'frame preparation + StripeSearch chain, shipped kernels, 5 steps',
                                   'value': round(px4 * 5 / dt4 / 1e6, 2), 'unit': 'contact-Mpx/s', 'ms_per_step': round(dt4 / 5 * 1e3, 3),
                                   'contact_px_per_step': px4,
                                   'chain_frac_of_hbm_peak': round(26.0 * px4 * 6 * 5 / (cw * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   # (SURVEY 8(d)'s 26 B per image-px count every pixel of every image; with the tiles below the diagonal,
                                   #  the shared blocks and the unread grey tiles not computed the contract figure can exceed what any
                                   #  hardware could move: then it is marked, as in roofline.kernels, and is not roofline evidence)
                                   'model_exceeds_work': bool(26.0 * px4 * 6 * 5 / (cw * 1e-3) / 1e9 / HBM_PEAK_GBS > 1.0),
                                   'kernels_ms_per_step': {k: round(v['ms'] / 5, 3) for k, v in s4.items()}}
                W4.release()
            except Exception as e:      # noqa: BLE001 -- an extra line must not cost the metric line
                out['band_1kb'] = {'error': str(e)[:200]}
        print(json.dumps(out), flush=True)
    hb.close()
    if world > 1:
        dist.barrier()                    # gloo control group
        if comm == 'gloo' and not rehearse:
            # The RCCL probe failed or timed out on some rank: its outcome is `config.comm_note`, nothing else.  A probe thread
            # that is still alive (joined for 90 s above, daemon) may hold a half-built RCCL communicator whose destructor -- or
            # destroy_process_group() walking over that group -- could block this rank's exit for good, after the JSON line is
            # out and the gloo barrier above has passed on every rank: so the process ends here, with status 0, without
            # running any further teardown.  (The path has never met a second device on the build boxes; the two-rank
            # rehearsal -- STP_BENCH_REHEARSE=1, tests/test_shard_gloo.py -- covers everything around it.)
            os._exit(0)
        dist.destroy_process_group()


def _timed_steps(W, barrier, n, pipelined=False):
    W.reset_stats(); barrier()
    t0 = time.perf_counter()
    if pipelined:
        W.run(n)
    else:
        for _ in range(n):
            W.step()
    barrier()
    return (time.perf_counter() - t0) / n * 1e3, W.stats()


def ab_round6(W, barrier, steps=5):
    """Same-process A/B of the two round-6 measures on the metric's workload: the images' symmetry (STP_SYM: the Canny tiles below
    an image's diagonal take their class words from the transposes of the tiles above) and the frame overlap (STP_REUSE: the
    block a frame shares with its successor is computed once).  ms per step and the Canny kernel's share with both, with one,
    with neither; records_equal: sha256 of every unit's record buffer of one step, each selection against `neither`."""
    import hashlib
    saved = {k: os.environ.get(k) for k in ('STP_SYM', 'STP_REUSE')}
    res, sha = {}, {}
    try:
        for name, sym, reuse in (('shipped', None, None), ('no_overlap', None, '0'), ('no_symmetry', '0', None), ('neither', '0', '0')):
            for k, v in (('STP_SYM', sym), ('STP_REUSE', reuse)):
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            h = hashlib.sha256()
            W.step(digest=h)                          # (also the selection's warm-up)
            sha[name] = h.hexdigest()[:16]
            ms, st = _timed_steps(W, barrier, steps)
            res[name] = {'ms_per_step': round(ms, 3), 'canny_ms_per_step': round(st['canny']['ms'] / steps, 3),
                         'lines_ms_per_step': round(st['lines']['ms'] / steps, 3)}
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return {'what': 'STP_SYM / STP_REUSE switched off one at a time and together, %d steps each, same process and box; '
                    'records_equal: every unit\'s record buffer of one step under each selection == under `neither`' % steps,
            'runs': res, 'records_equal': all(v == sha['neither'] for v in sha.values()), 'records_sha256': sha}


def emulated_shares(W, barrier, n, steps=10, pipelined=True):
    """ms per step of each of the n shares an n-rank run would cut the genome into, one after the other ALONE on this GPU
    (diagnostic: per-rank fixed costs -- pipeline fill and drain, launch tails -- without an n-GPU node), timed as the metric is
    (`pipelined`: the steps as one pipeline) or with every step drained.  NOT a multi-GPU measurement and no scaling claim:
    eight real ranks share host cores, PCIe and power."""
    saved = W.my_units
    out = []
    try:
        for r in range(n):
            W.my_units = W.units_for(n, r)
            W.step(); W.step()
            ms = min(_timed_steps(W, barrier, steps, pipelined)[0] for _ in range(2))   # (the lower of two runs: a share is 6 ms, one stray
            out.append(round(ms, 2))                                                    #  host hiccup is a fifth of it)
    finally:
        W.my_units = saved
    return out


REGIMES = (('balanced', {}), ('raw_counts', {'balanced': False}), ('counts_div8', {'balanced': False, 'count_div': 8}),
           ('counts_div32', {'balanced': False, 'count_div': 32}), ('depth_x10', {'depth': 10.0}))


def regimes_extra(hb, dev, sigma, barrier, steps=5):
    """Other data regimes on the device (not the metric): one chr16-size chromosome (19 642 bins, 99 frames x 5 levels x 6 images) of
    balanced floats (the metric's family), raw integer counts (`--norm None`), shallow libraries (counts // 8, // 32: few-level
    images full of exact ties) and a 10 x deeper matrix; frame preparation + StripeSearch chain, shipped kernels.  Per regime:
    contact-Mpx/s, and k_canny_f32's own counters on 8 sampled frames x 2 brightness images at the first level: fraction of
    the image in tiles skipped as flat, candidates, pixels settled by the f64 resolver and tile-images handed to the exact
    kernel, per image.  tests/test_gpu_regimes.py checks every record of such chromosomes against the oracle."""
    res = {}
    for name, kw in REGIMES:
        spec = dict(names=['chr16'], nbins=[CHR16_BINS], seeds=[16], wl='regime ' + name, kw=kw)
        Wr = _Workload(hb, dev, spec, 1, 0, '', score=False, sigma=sigma)
        try:
            Wr.step(); Wr.step()
            Wr.reset_stats(); barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                nrec, px = Wr.step()
            barrier()
            dt = (time.perf_counter() - t0) / steps
            st = Wr.stats()
            fst, fen = Wr.tabs[0]
            fr = Wr.bands['chr16'].frames(fst, fen)
            flat, cand, reso, flag, nimg = 0.0, 0, 0, 0, 0
            for f in range(5, len(fst), 12):
                if fr.S[f] == 0:
                    continue
                for bi in (0, 5):
                    c = fr.dbg_canny_f32(f, float(Wr.Ms[0][0]), bi, sigma=sigma)
                    flat += float(np.isnan(c['g']).mean()); cand += c['candidates']; reso += c['resolved']; flag += c['flagged']
                    nimg += 1
            fr.close()
            res[name] = {'value': round(px / dt / 1e6, 1), 'unit': 'contact-Mpx/s', 'ms_per_step': round(dt * 1e3, 3),
                         'stripe_records': int(nrec), 'maxpixel_levels': [round(float(m), 4) for m in Wr.Ms[0]],
                         'canny_ms_per_step': round(st['canny']['ms'] / steps, 3), 'gray_ms_per_step': round(st['gray']['ms'] / steps, 3),
                         'lines_ms_per_step': round(st['lines']['ms'] / steps, 3),
                         'sampled_images': nimg, 'flat_tile_fraction': round(flat / max(nimg, 1), 3),
                         'candidates_per_image': round(cand / max(nimg, 1), 1), 'resolver_pixels_per_image': round(reso / max(nimg, 1), 1),
                         'exact_tile_images_per_image': round(flag / max(nimg, 1), 2)}
        finally:
            Wr.release()
    return res


class _Workload:
    """Everything a step needs, resident on the device: the bands of the chromosomes, maxpixel quantiles, expected values,
    background tables, the frame tables and this rank's units (pieces of at most one device chunk)."""

    def __init__(self, hb, dev, spec, world, rank, emulate_rank, score=True, sigma=2.0):
        import torch
        from stripenn_amd import synth_device, shard, getStripe as GS, backend as BK
        self.hb, self.ctx, self.score, self.BK, self.sigma = hb, hb.ctx, score, BK, float(sigma)
        self.names, self.nbins, seeds = spec['names'], spec['nbins'], spec['seeds']
        names, nbins = self.names, self.nbins
        self.hw = hw = 512
        self.bs = int(50000 / RESOL)
        sizes = np.array([n * RESOL for n in nbins], dtype=np.int64)
        self.nframes = [-(-n // 200) for n in nbins]
        if emulate_rank:
            er, en_ = (int(v) for v in emulate_rank.split('/'))
            self.my_units = self.units_for(en_, er)
        else:
            self.my_units = self.units_for(world, rank)
        # ---- untimed set-up: every band in HBM, maxpixel quantiles, expected values, background tables
        self.chroms, self.tens, self.bands = {}, {}, {}
        need_all = score                      # the background tables sample every chromosome of the genome
        for ci, nm in enumerate(names):
            if not need_all and not any(u[0] == ci for u in self.my_units):
                continue
            self.chroms[nm] = synth_device.DeviceChrom(nbins[ci], seeds[ci], dev, **spec.get('kw', {}))
            self.tens[nm] = self.chroms[nm].band(hw)
            torch.cuda.synchronize()
            self.bands[nm] = self.ctx.band_wrap(self.tens[nm].data_ptr(), nbins[ci], hw, keepalive=self.tens[nm])
        self.Ms = {}
        for ci in sorted({u[0] for u in self.my_units}):
            # the reference's getQuantile step (np.quantile(mat[mat > 0], q)) is outside the timed path; on this
            # band-limited synthetic data the band holds every positive pixel, so the order statistics of the band's
            # positive entries are those of the dense matrix (sorted on the device, numpy's interpolation on the host)
            t = self.tens[names[ci]]
            v = torch.sort(t[t > 0]).values
            self.Ms[ci] = GS.quantile_linear(lambda ranks: v[torch.as_tensor(ranks, device=dev)].cpu().numpy(), int(v.numel()),
                                             MAXPIXEL)
            del v
        self.EV = {}
        if score:
            sel = _DeviceSelector(self.chroms, RESOL)
            obj = GS.getStripe(sel, RESOL, 10, 8, 2.0, names, names, sizes, sizes, 2, 3, 123456789, backend=hb)
            for nm in self.bands:
                obj._bands[nm] = self.bands[nm]
            EVall = obj.mpmean()
            self.EV = {ci: np.asarray(EVall[names[ci]]) for ci in {u[0] for u in self.my_units}}
            bg = obj.nulldist()
            hb.set_background(*bg)
        self.tabs = {ci: frame_table(nbins[ci]) for ci in {u[0] for u in self.my_units}}
        self.host_wait_s = 0.0
        self.host_call_s = 0.0          # time inside the other blocking device calls of the step (frames, p-value, Stripiness)
        # Scoring on a host thread of its own, through a second context (the ABI's threading model: one context per host
        # thread; contexts run concurrently): p-value and Stripiness of unit u are two blocking calls whose kernels queue
        # beside the chain, and on the search thread they kept it from collecting unit u + 1 in time.  The step ends when
        # every candidate of every unit is scored (step() joins the queue).  Measured (profiles/r04_ab_score_thread.txt): the
        # search thread then waits 54 of 80 ms instead of 13 -- and the step takes the same 80 ms: it is device-bound either way
        # (chain 75 ms + pipeline fill / drain), so the single thread stays the default (STP_BENCH_SCORE_THREAD=1 selects this).
        self.hb2, self.bands2, self._q, self._thr, self._err = None, {}, None, None, []
        # Two search contexts (two HIP streams, two workspaces) taking alternate units: the kernels of two units share the device,
        # so one unit's launch tails, its latency-bound k_lines and the drain behind its last kernel are filled by the other's chain
        # Measured and NOT the default (round 5, STP_BENCH_CTX=2): 74.8-75.2 against 71.8 ms per step -- kernels of two streams that
        # really run side by side slow each other down by more than the filled tails give back (round 2 had found the same).
        # (end of round 5: 65.2 against 63.3 ms; the two contexts driven by two host THREADS, alternate units each: 66.9 -- although
        #  two rank PROCESSES on the one device finish the genome step in 60.7 ms: profiles/r05_rehearse_2ranks.json)
        self.two_ctx = os.environ.get('STP_BENCH_CTX', '1') == '2' and not emulate_rank
        if self.two_ctx and not (score and os.environ.get('STP_BENCH_SCORE_THREAD', '0') == '1'):
            self.hb2 = BK.HipBackend(dev.index or 0)
            if score:
                self.hb2.set_background(*bg)
            for nm in self.bands:
                self.bands2[nm] = self.hb2.ctx.band_wrap(self.tens[nm].data_ptr(), nbins[names.index(nm)], hw, keepalive=self.tens[nm])
        if score and os.environ.get('STP_BENCH_SCORE_THREAD', '0') == '1':
            self.two_ctx = False
            import queue
            import threading
            prio = os.environ.get('STP_BENCH_SCORE_PRIORITY')      # stream priority of the scoring context (measurement hook)
            if prio:
                os.environ['STP_AUX_PRIORITY'] = prio
            self.hb2 = BK.HipBackend(dev.index or 0)
            os.environ.pop('STP_AUX_PRIORITY', None)
            self.hb2.set_background(*bg)
            for nm in self.bands:
                self.bands2[nm] = self.hb2.ctx.band_wrap(self.tens[nm].data_ptr(), nbins[names.index(nm)], hw, keepalive=self.tens[nm])
            self._q = queue.Queue()
            self._thr = threading.Thread(target=self._score_loop, daemon=True)
            self._thr.start()

    def units_for(self, world, rank):
        """The units rank `rank` of `world` works through in a step: its span of the (chromosome x frame) grid
        (shard.frame_spans) cut into pipeline stages of at most one device chunk (6 144 images = 204 frames at 5 levels x 6
        brightness).  The whole genome (world 1) runs in chromosome-size units, largest first, smallest last -- a step that drains
        ends with the scoring of its LAST unit (nothing left on the device to hide it behind), so that unit should be the short
        one (round 5: 66.8 -> 65.7 ms per genome step); a share of it in equal parts of at most 128 frames (below).  The units
        are independent; STP_BENCH_PIECE / STP_BENCH_ORDER=file override."""
        from stripenn_amd import shard
        spans = shard.frame_spans(self.nframes, world)[rank]          # (chromosome index, first frame, end frame)
        # Round 6 (tools/emulated_shares_matrix.sh, profiles/r06_shares_matrix*.txt): a 1/8 share holds 3-4 chromosome pieces.  Cut into
        # pieces of 83 frames + remainders they made 5-7 units, some of a few frames only, each of which costs the host its full
        # per-unit work (frame preparation, record fetch, score inputs, one blocking score call) for microseconds of device time:
        # 8.8 / 9.0 ms for shares 5 / 7 of 8.  Now: shard.frame_spans leaves no slivers (cuts snap onto chromosome boundaries), and
        # a chromosome piece is cut into the fewest EQUAL parts of at most 128 frames: 3-5 units per share, 7.2-7.6 ms.
        # With the timed steps pipelined (the default since the second half of round 6) nothing drains between two steps and a share is
        # HOST-bound (the search thread waits for the device 0.15 of 5.5 ms): fewer, larger units win -- one device chunk (204 frames)
        # per unit: 5.2-5.65 ms for the eight shares against 5.3-6.1 with <= 128 frames (profiles/r06_shares_matrix_pipelined.txt).
        whole = world == 1
        pipelined = os.environ.get('STP_BENCH_PIPELINE_STEPS', '1') != '0'
        piece = max(1, int(os.environ.get('STP_BENCH_PIECE', '204' if whole or pipelined else '128')))
        units = []
        for ci, f0, f1 in spans:
            n, k = f1 - f0, -(-(f1 - f0) // piece)
            units += [(ci, f0 + (n * j) // k, f0 + (n * (j + 1)) // k) for j in range(k)]
        order = os.environ.get('STP_BENCH_ORDER', 'size')
        if order != 'file':
            units.sort(key=lambda u: -(u[2] - u[1]))
        if order == 'interleave':              # (A/B hook: large and small units alternate -- the host's per-unit work of a small unit hides behind a large one's chain)
            units = [units[i // 2] if i % 2 == 0 else units[len(units) - 1 - i // 2] for i in range(len(units))]
        return units

    def _score_loop(self):
        while True:
            job = self._q.get()
            try:
                if job is None:
                    return
                ci, recs, nz, st = job
                pv, sc = self.BK.score_inputs(recs, nz, st, self.nbins[ci], self.bs)
                sband = self.bands2[self.names[ci]]
                self.hb2.score(sband, self.bs, self.EV[ci], pv, sc)
            except BaseException as e:      # noqa: BLE001 -- reported by step()
                self._err.append(e)
            finally:
                self._q.task_done()

    def release(self):
        import torch
        if self._thr is not None:
            self._q.put(None)
            self._thr.join()
            self._thr = None
        for b in list(self.bands2.values()) + list(self.bands.values()):
            b.close()
        if self.hb2 is not None:
            self.hb2.close()
            self.hb2 = None
        self.bands, self.bands2, self.tens, self.chroms = {}, {}, {}, {}
        torch.cuda.empty_cache()

    def stats(self):
        """per-kernel timers of the step: the search context's and, when scoring runs on its own thread, that context's"""
        st = dict(self.ctx.stats())
        if self.hb2 is not None:
            for k, v in self.hb2.ctx.stats().items():
                if k in st:
                    st[k] = {a: st[k][a] + v[a] for a in v}
                else:
                    st[k] = v
        return st

    def reset_stats(self, profiling=True):
        for c in [self.ctx] + ([self.hb2.ctx] if self.hb2 is not None else []):
            c.set_profiling(profiling)
            c.reset_stats()

    def synchronize(self):
        self.ctx.synchronize()
        if self.hb2 is not None:
            self.hb2.ctx.synchronize()

    def _launch(self, unit, k=0):
        """frame compaction + medpixel of one unit (small kernels on the context's auxiliary stream) and its whole
        StripeSearch chain enqueued on the main stream of context k; returns without waiting for the chain"""
        ci, f0, f1 = unit
        st, en = self.tabs[ci]
        if os.environ.get('STP_BENCH_CACHE_FRAMES') == '1':
            # (diagnostic, VERDICT r05 #4: what frame preparation costs THE STEP -- the frames of a unit prepared once and served
            #  from the previous step's results; the step then runs without k_frame_prep and its blocking round trip.  Not the metric.)
            cache = self.__dict__.setdefault('_frames_cache', {})
            fr = cache.get((unit, k))
            if fr is None:
                fr = cache[(unit, k)] = (self.bands2 if k else self.bands)[self.names[ci]].frames(st[f0:f1], en[f0:f1])
                fr.close = lambda: None                 # kept for the next step (released with the workload's bands)
        else:
            fr = (self.bands2 if k else self.bands)[self.names[ci]].frames(st[f0:f1], en[f0:f1])
        return unit, fr, fr.stripe_search_begin(self.Ms[ci], sigma=self.sigma)

    def step(self, digest=None):
        return self.run(1, digest)

    def run(self, nsteps, digest=None):
        """`nsteps` steps (each: every unit of this rank once).  Two searches are kept in flight: while the device runs the
        chain of unit u+1 (and u+2 is queued behind it), the host collects the records of unit u, builds the score inputs and
        enqueues its p-value / Stripiness kernel -- the single in-order stream never runs dry.  run(1) = one step that drains at
        its end; with nsteps > 1 (the benchmark's timed region) the first units of step s + 1 are launched while the last units
        of step s are collected and scored.  All work is finished when
        run() returns.  Returns the records and contact pixels of ONE step (every step does the same work)."""
        nrec, px = 0, 0.0
        nu = len(self.my_units)
        todo = [(s, u) for s in range(nsteps) for u in self.my_units][::-1]
        nctx = 2 if self.two_ctx else 1
        depth = int(os.environ.get('STP_BENCH_FLIGHT', '2')) * nctx     # searches in flight: two per context
        tw = time.perf_counter()
        flight, nl = [], 0
        def launch():
            nonlocal nl
            sidx, unit = todo.pop()
            flight.append((sidx, nl % nctx) + self._launch(unit, nl % nctx)); nl += 1
        for _ in range(min(depth, len(todo))):
            launch()
        self.host_call_s += time.perf_counter() - tw
        while flight:
            sidx, k, (ci, f0, f1), fr, pend = flight.pop(0)
            tw = time.perf_counter()
            recs = pend.wait()
            self.host_wait_s += time.perf_counter() - tw           # time the host spent waiting for the device
            if todo:
                tw = time.perf_counter()
                launch()
                self.host_call_s += time.perf_counter() - tw       # frame preparation (blocking: S / nz / medpixel come back) + enqueue
            if self.score and self._q is not None:
                self._q.put((ci, recs, np.array(fr.nz), self.tabs[ci][0][f0:f1]))
            elif self.score:
                st = self.tabs[ci][0]
                hbk = self.hb2 if k else self.hb
                sband = (self.bands2 if k else self.bands)[self.names[ci]]
                pv, sc = self.BK.score_inputs(recs, fr.nz, st[f0:f1], self.nbins[ci], self.bs)
                tw = time.perf_counter()
                if os.environ.get('STP_BENCH_SCORE_CALLS') == '2':  # (A/B hook: the two separate calls of rounds 1-4)
                    hbk.pvalue(sband, self.bs, pv)
                    hbk.stripiness(sband, self.EV[ci], sc)
                else:
                    hbk.score(sband, self.bs, self.EV[ci], pv, sc)        # p-value and Stripiness: one upload, one launch, one download
                self.host_call_s += time.perf_counter() - tw       # one blocking call: copy in, kernel, copy out
            if digest is not None:
                digest.update(np.ascontiguousarray(recs[:len(recs)]).tobytes())
            if sidx == nsteps - 1:                                 # (the figures of one step: the last)
                nrec += len(recs)
                px += float((fr.S.astype(np.float64) ** 2).sum())
            fr.close()
        if self._q is not None:
            self._q.join()                      # every candidate of the step is scored
            if self._err:
                raise self._err[0]
        return nrec, px * len(MAXPIXEL)


class _DeviceSelector:
    """`matrix(balance=...)`-like selector over device-generated chromosomes (set-up of the background tables):
    fetch() evaluates the block on the GPU and copies it to the host; row_nonzero() answers the pools' only
    question (which rows of the block have a non-zero sum after NaN -> 0; all values are >= 0) without the copy."""

    def __init__(self, chroms, resol):
        self.chroms, self.resol = chroms, int(resol)

    def _extent(self, region):
        region = str(region)
        if ':' not in region:
            return region, 0, self.chroms[region].nbins
        name, rng = region.rsplit(':', 1)
        s, e = rng.replace(',', '').split('-')
        s, e = int(s), int(e)
        if s < 0 or e > self.chroms[name].nbins * self.resol or s > e:
            raise ValueError('Genomic region out of bounds: %s' % region)
        return name, s // self.resol, -(-e // self.resol)

    def _block(self, region, region2):
        import torch
        n1, r0, r1 = self._extent(region)
        n2, c0, c1 = (n1, r0, r1) if region2 is None else self._extent(region2)
        ch = self.chroms[n1]
        r = torch.arange(r0, r1, device=ch.device, dtype=torch.int64)[:, None].expand(r1 - r0, c1 - c0)
        c = torch.arange(c0, c1, device=ch.device, dtype=torch.int64)[None, :].expand(r1 - r0, c1 - c0)
        val = (ch._counts(r, c) * ch.w[r]) * ch.w[c]
        return torch.where(ch.nanflag[r] | ch.nanflag[c], torch.full_like(val, float('nan')), val)

    def fetch(self, region, region2=None):
        return self._block(region, region2).cpu().numpy()

    def row_nonzero(self, region, region2=None):
        return (self._block(region, region2) > 0).any(dim=1).cpu().numpy()


def reference_python_figure():
    """The reference's own rate, measured in the BUILD container (the reference's files never travel to the GPU box):
    wall times stored with the config-size fixture tests/golden/e2e_chr16.npz (oracle/refharness/gen_golden.py chr16 --
    the unmodified reference, numcores = 8, chr16-size chromosome, maxpixel 0.95-0.99 = BASELINE.json configs[1])."""
    try:
        g = np.load(os.path.join(ROOT, 'tests', 'golden', 'e2e_chr16.npz'))
        T = dict(zip([str(k) for k in g['time_keys']], [float(v) for v in g['time_s']]))
        px = 0.0                                  # contact-px of one level: sum of S^2 over the 99 frames
        from stripenn_amd import synth
        nb = -(-int(g['sizes'][0]) // int(g['resol']))
        nanb = synth.SynthChrom(nb, int(g['seed0'])).nan_bins
        for i in range(-(-nb // 200)):
            a, b = (0 if i == 0 else i * 200 - 100), min((i + 1) * 200 + 99, nb - 1)
            S = (b - a + 1) - int(((nanb >= a) & (nanb <= b)).sum())
            px += float(S) * S
        ext = sum(T['extract%d' % i] for i in range(5))
        return {'value': round(5 * px / ext / 1e6, 3), 'unit': 'contact-Mpx/s', 'cores': int(g['core']), 'kind': 'reference',
                'where': 'build container (%s), NOT this box' % str(g['host']),
                'sample': 'the five extract calls (StripeSearch of 99 frames + RemoveRedundant + pvalue per level) of the '
                          'unmodified reference on the chr16-size chromosome: %.1f s; whole compute incl. quantile, '
                          'expected values and background: %.1f s = %.3f contact-Mpx/s'
                          % (ext, T['config1_total'], 5 * px / T['config1_total'] / 1e6),
                'source': 'tests/golden/e2e_chr16.npz (time_keys / time_s)'}
    except Exception as e:      # noqa: BLE001 -- the figure is informative; its absence must not cost the line
        return {'value': None, 'error': str(e)[:120]}


def _load_pmc(workload):
    try:
        with open(PMC_FILE) as f:
            return json.load(f).get(workload)
    except (OSError, ValueError):
        return None


def e2e_compute(names, chroms, contact_px, hb):
    """The whole `stripenn compute` driver (quantiles -> expected values -> background -> candidates -> p-values ->
    redundancy filter -> Stripiness -> TSVs) on the same genome handed over as cooler's pixel table in host
    memory; reported beside the kernel-path figure, never as `value`."""
    import contextlib
    import io as _io
    import tempfile
    from stripenn_amd import io, stripenn, synth_device
    t0 = time.time()
    table = synth_device.pixel_table(names, chroms, RESOL)
    t_table = time.time() - t0
    orig = stripenn.open_matrix
    stripenn.open_matrix = lambda cool: io.pixel_matrix(table)
    out = tempfile.mkdtemp(prefix='stp_e2e_')
    try:
        runs = []
        for _ in range(4):            # the first runs also pay first-use costs (workspaces, band pool, pinned-buffer cache): all are reported
            t0 = time.time()
            with contextlib.redirect_stdout(_io.StringIO()):
                stripenn.compute('pixels:in-memory', out, 'weight', 'all', 2.0, 10, 8, ','.join(str(m) for m in MAXPIXEL), 8, 0.1,
                                 '0', False, 3, 123456789, force=True, backend=hb)
            runs.append(time.time() - t0)
        dt = runs[-1]
        nu = open(os.path.join(out, 'result_unfiltered.tsv')).read().count('\n') - 1
        nf = open(os.path.join(out, 'result_filtered.tsv')).read().count('\n') - 1
    finally:
        stripenn.open_matrix = orig
    return {'seconds': round(dt, 2), 'first_run_seconds': round(runs[0], 2), 'runs_seconds': [round(r, 3) for r in runs],
            'contact_Mpx_s': round(contact_px / dt / 1e6, 1), 'stored_pixels': int(len(table.count)),
            'stripes_unfiltered': nu, 'stripes_filtered': nf, 'pixel_table_build_s': round(t_table, 1),
            'what': 'stripenn_amd.stripenn.compute on the same genome as an in-memory pixel table (quantile -> TSVs), 1 GPU; `seconds` is the last of four runs in this process (`runs_seconds`: the pools of bands, workspaces and pinned buffers '
                    'reach their steady state in the third)'}


if __name__ == '__main__':
    main()
