"""Multi-GPU `compute`: chromosomes sharded over one process per GPU by a static LPT queue.

Units are independent: a chromosome's band, frames, expected values, candidate
stripes, p-values and Stripiness never need another chromosome's pixels.  The only shared data are
the four 400 x ~1000 background tables (12.8 MB), whose per-chromosome parts are themselves
independent; they are exchanged as host objects through torch.distributed (no RCCL collective on
the data path, nothing travels over xGMI).  Rank 0 merges the per-rank tables back into the
reference's row order (maxpixel, chromosome, frame) and writes the TSVs.
"""
import os
import sys
import time

import numpy as np
import pandas as pd


def lpt_assign(costs, nworkers):
    """Longest-processing-time-first static assignment; returns a list of index lists per worker."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * nworkers
    out = [[] for _ in range(nworkers)]
    for i in order:
        w = min(range(nworkers), key=lambda k: (load[k], k))
        out[w].append(i)
        load[w] += costs[i]
    for o in out:
        o.sort()
    return out


def chrom_costs(chromsizes, resol):
    """Cost of a chromosome ~ its number of 400x400 frames = ceil(bins / 200) (getStripe.py:841)."""
    return [float(-(-int(-(-int(s) // int(resol))) // 200)) for s in chromsizes]


SNAP_FRAMES = 8    # a cut this close to a chromosome boundary moves onto it (frame_spans)


def frame_spans(nframes, world, snap=SNAP_FRAMES):
    """Static work queue of the (chromosome x frame) grid: the frames of all chromosomes, laid end to end in
    chromosome order, are cut into `world` contiguous spans of equal length (+-1 frame), so a rank holds at most
    one partial chromosome at each end (mm10 at 5 kb over 8 GPUs: 331 vs 330.6).  A cut that falls within `snap` frames
    of a chromosome boundary moves onto the boundary (round 6): a sliver of a few frames costs a rank a band, a frame
    preparation, a launch of every kernel and a blocking score call of its own -- ~0.4 ms of a 6 ms share, measured -- while
    the <= 8 frames it shifts are 2.4 % of a 1/8 share of mm10 (the spans stay within `snap` frames of equal).
    Every maxpixel level of a frame stays on the rank that holds the frame (they share the band reads).
    Returns, per rank, a list of (chromosome index, first frame, one past the last frame)."""
    total = int(sum(nframes))
    cuts = [(total * r) // world for r in range(world + 1)]
    bounds, base = [], 0
    for nf in nframes:
        base += int(nf)
        bounds.append(base)
    for r in range(1, world):
        near = min(bounds, key=lambda b: (abs(b - cuts[r]), b)) if bounds else cuts[r]
        if 0 < abs(near - cuts[r]) <= snap and cuts[r - 1] < near < total:
            cuts[r] = near
    for r in range(1, world):                        # (monotone whatever the snaps did: tiny genomes, many ranks)
        cuts[r] = max(cuts[r], cuts[r - 1])
    out = [[] for _ in range(world)]
    base = 0
    for ci, nf in enumerate(nframes):
        nf = int(nf)
        for r in range(world):
            lo, hi = max(cuts[r], base), min(cuts[r + 1], base + nf)
            if lo < hi:
                out[r].append((ci, lo - base, hi - base))
        base += nf
    return out


def chrom_nframes(chromsizes, resol):
    """ceil(ceil(size / resol) / 200) frames per chromosome (getStripe.py:841)."""
    return [int(-(-int(-(-int(s) // int(resol))) // 200)) for s in chromsizes]


class _Comm:
    """all_gather of picklable host objects; trivial when world == 1."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def allgather(self, obj):
        if self.world == 1:
            return [obj]
        import torch.distributed as dist
        out = [None] * self.world
        dist.all_gather_object(out, obj)
        return out

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()


HALO = 2   # frames searched beyond each end of a rank's span (see sharded_compute)
# Two invariants the halo width rests on (the byte-equality tests at world 2 and 3 guard both):
#  * the redundancy filter runs exactly three times (per frame, per chromosome by size, whole table by p-value), each
#    reaching one frame further and collecting its deletions without applying them; a fourth pass, or a filter that
#    applies deletions as it goes, needs a wider halo;
#  * every row extract() hands to pvalue() touches one end of the diagonal (StripeSearch: ud = 1 gives x2 == y2, ud = 2
#    gives x1 == y1), so the reference's "inherited background rows" state (getStripe.py:584-597), which follows TABLE
#    order and would differ on a rank whose table starts in the middle of a chromosome, never fires.  sharded_compute
#    asserts it (pvalue_modes).


def pvalue_modes(df, resol):
    """0 / 1 / 2 per row as getStripe.pvalue classifies it (getStripe.py:583-597): anchored on the diagonal at its
    first bin (down), at its last bin (up), or neither (inherits the previous row's background rows)."""
    p1, p2, p3, p4 = (np.asarray(df[k], dtype=np.int64) for k in ('pos1', 'pos2', 'pos3', 'pos4'))
    x1, x2, y1, y2 = (p1 - 1) // resol, p2 // resol, (p3 - 1) // resol, p4 // resol
    return np.where(x1 == y1, 0, np.where(x2 == y2, 1, 2))


def sharded_compute(rank, world, cool, out, norm, chrom, canny, minL, maxW, maxpixel, numcores, pvalue, mask, slow,
                    bfilter, seed, force=True, backend_factory=None, write=True, prepared=False):
    """Body of one rank.  Returns (result_table, res_filter) on rank 0, (None, None) elsewhere.

    Two static work queues, no collective on the data path (host objects only):
      A. whole-chromosome steps -- maxpixel quantiles (stripenn.py:126-131), expected values (:132), background
         pools / windows (:133) -- by LPT over chromosomes; their small results are all-gathered;
      B. the (chromosome x frame) grid of step 4 / 5 (:134-147) in contiguous spans of equal frame count
         (frame_spans).  RemoveRedundant only ever compares rows of frames n and n + 1 and collects deletions
         without applying them (getStripe.py:1116-1161), and the path runs it three times in a row (per frame,
         per chromosome 'size', whole table 'pvalue'), so a rank searches HALO = 2 extra frames on each side of
         its span: the verdict on every row of its own frames then equals the single-process one, and the halo
         rows are dropped before Stripiness.
    prepared=True: the output directory and log were written by the launcher (launch_compute)."""
    from . import getStripe
    from .io import open_matrix
    from .stripenn import (RESULT_COLUMNS, addlog, finish_tables, makeOutDir, resolve_norm, select_chromosomes, write_tsv)
    np.seterr(divide='ignore', invalid='ignore')
    comm = _Comm(rank, world)
    if out[-1] != '/':
        out += '/'
    if rank == 0 and write and not prepared:
        makeOutDir(out, force)
        addlog(cool, out, norm, chrom, canny, minL, maxW, maxpixel, numcores, pvalue, mask, bfilter)
    levels = list(map(float, maxpixel.split(',')))
    Lib = open_matrix(cool)
    normv = resolve_norm(Lib, norm)
    all_names, all_sizes, names, sizes = select_chromosomes(Lib, chrom)
    names = [str(n) for n in names]
    resol = Lib.binsize
    sel = Lib.matrix(balance=normv)
    backend = backend_factory(rank) if backend_factory is not None else None
    t0 = time.time()
    phase = {}                                   # seconds per phase of this rank (written to stripenn.log by rank 0)

    # ---- A: whole-chromosome steps
    mineA = lpt_assign(chrom_costs(sizes, resol), world)[rank]
    namesA = [names[i] for i in mineA]
    sizesA = np.asarray(sizes)[mineA] if len(mineA) else np.zeros(0, dtype=np.int64)
    objA = getStripe.getStripe(sel, resol, minL, maxW, canny, all_names, namesA, all_sizes, sizesA, numcores, bfilter, seed,
                               backend=backend, device=rank)
    MPa = (objA.getQuantile_slow if slow else objA.getQuantile_original)(Lib, namesA, levels) if namesA else {}
    EVa = objA.mpmean()
    MP, EV = {}, {}
    for mp_part, ev_part in comm.allgather((MPa, EVa)):
        MP.update(mp_part)
        EV.update(ev_part)
    # background: sample sizes need every candidate chromosome's pools; a chromosome's pools and windows are
    # computed where its band lives.  numcores > 1: the PRNG restarts per chromosome (loky pickles `self`);
    # numcores == 1: one stream runs on across chromosomes, so every rank replays all draws (a few thousand
    # random() calls) and forms the windows of its own chromosomes only.
    cand = objA.null_candidates()
    owner = {names[i]: r for r, part in enumerate(lpt_assign(chrom_costs(sizes, resol), world)) for i in part}
    extra = [c for c in cand if c not in owner]                   # sampled, but not among the --chrom selection
    owner.update({c: k % world for k, c in enumerate(extra)})
    share = [c for c in cand if owner[c] == rank]
    avail = {}
    for part in comm.allgather({c: objA.null_available_cols(c) for c in share}):
        avail.update(part)
    cand2, samplesize = objA.null_samplesizes(cand, [avail[c] for c in cand])
    pools = {}
    for part in comm.allgather({c: objA.null_pools(c) for c in share if c in cand2} if numcores == 1 else {}):
        pools.update(part)
    parts = {}
    for c in cand2:
        mine = c in share
        if numcores == 1 or mine:
            r = objA.null_tables(c, cand2, samplesize, pools=pools.get(c), windows=mine)
            if mine:
                parts[c] = r
    allparts = {}
    for part in comm.allgather(parts):
        allparts.update(part)
    bg = objA.null_concat([allparts[c] for c in cand2])
    phase['A quantile + expected + background'] = time.time() - t0

    # ---- B: candidate stripes, p-values, redundancy filters, Stripiness of this rank's frame span
    nfr = chrom_nframes(sizes, resol)
    spans = frame_spans(nfr, world)[rank]
    own = {names[ci]: (lo, hi) for ci, lo, hi in spans}
    search = {names[ci]: (max(0, lo - HALO), min(nfr[ci], hi + HALO)) for ci, lo, hi in spans}
    namesB = [names[ci] for ci, _, _ in spans]
    sizesB = np.asarray(sizes)[[ci for ci, _, _ in spans]] if spans else np.zeros(0, dtype=np.int64)
    objB = getStripe.getStripe(sel, resol, minL, maxW, canny, all_names, namesB, all_sizes, sizesB, numcores, bfilter, seed,
                               backend=objA.backend, frame_span=search)
    for c in list(objA._bands):                  # whole bands phase A built are reused; the others are freed
        if c in search and c not in objA._partial:
            objB._bands[c] = objA._bands.pop(c)
    objA.release()
    table = pd.DataFrame(columns=RESULT_COLUMNS)
    s = []
    if namesB:
        tabs = [objB.extract(MP, i, perc, *bg) for i, perc in enumerate(levels)]
        tabs = [t for t in tabs if len(t)]
        for t in tabs:                            # the halo argument needs every row anchored on the diagonal (see HALO)
            if (pvalue_modes(t, resol) == 2).any():
                raise AssertionError('sharded_compute: a candidate row touches neither end of the diagonal; its p-value '
                                     'would depend on the table order of another rank')
        if tabs:
            table = pd.concat(tabs)
            table = objB.RemoveRedundant(df=table, by='pvalue')
            lo = np.array([own[str(c)][0] for c in table['chr']], dtype=np.int64)
            hi = np.array([own[str(c)][1] for c in table['chr']], dtype=np.int64)
            num = np.asarray(table['num'], dtype=np.int64)
            table = table.iloc[np.nonzero((num >= lo) & (num < hi))[0]]          # drop the halo frames
            s = objB.scoringstripes(table, EV, mask)[0]
    # (the nine helper columns have done their work -- the filters and Stripiness above -- and would only be pickled, sent and
    #  concatenated to be dropped by finish_tables: 21 -> 12 columns through the gather)
    from .stripenn import HELPER_COLUMNS
    table = table.drop(columns=[c for c in HELPER_COLUMNS if c in table.columns])
    table.insert(table.shape[1], '_stripiness', list(s), True)
    elapsed = time.time() - t0
    phase['B search + p-values + filters + Stripiness'] = elapsed - phase['A quantile + expected + background']
    gathered = comm.allgather((rank, table, elapsed, phase))
    t_merge = time.time()
    result = (None, None)
    if rank == 0:
        merged = pd.concat([g[1] for g in sorted(gathered, key=lambda g: g[0]) if len(g[1])] or [gathered[0][1]])
        lev_idx = {str(p * 100) + '%': i for i, p in enumerate(levels)}
        chr_idx = {str(n): i for i, n in enumerate(names)}
        key = [lev_idx[str(m)] * (len(names) + 1) + chr_idx[str(c)] for m, c in zip(merged['maxpixel'], merged['chr'])]
        merged = merged.iloc[np.argsort(np.asarray(key, dtype=np.int64), kind='stable')]
        stri = merged['_stripiness'].tolist()
        merged = merged.drop(columns=['_stripiness'])
        result_table, res_filter = finish_tables(merged, stri, pvalue)
        if write:
            write_tsv(result_table, out + 'result_unfiltered.tsv')
            write_tsv(res_filter, out + 'result_filtered.tsv')
            with open(out + 'stripenn.log', 'a') as f:
                f.write('gpus: %d\n' % world)
                for g in sorted(gathered, key=lambda g: g[0]):
                    f.write('rank %d: %.2f s (%s)\n' % (g[0], g[2], '; '.join('%s %.3f' % kv for kv in g[3].items())))
                # what stays serial on rank 0 behind the ranks' work: gathering the tables, merge into the reference's row
                # order, final columns / filter / sort, both TSVs (stripenn.py:149-159)
                f.write('rank 0 serial tail (gather wait + merge + TSVs): %.3f s\n' % (time.time() - t0 - elapsed))
                f.write('rank 0 merge + TSVs alone: %.3f s\n' % (time.time() - t_merge))
        result = (result_table, res_filter)
    if backend is None:
        objB.backend.close()
    comm.barrier()
    return result


def _worker(rank, world, port, args):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('gloo', rank=rank, world_size=world)   # host objects only
    try:
        sharded_compute(rank, world, *args, prepared=True)
    finally:
        dist.destroy_process_group()


def launch_compute(cool, out, norm, chrom, canny, minL, maxW, maxpixel, numcores, pvalue, mask, slow, bfilter, seed, force,
                   gpus):
    """One process per GPU (torch.multiprocessing spawn); rank r drives HIP device r.  The output directory is
    prepared HERE, in the parent, before any process is spawned: the reference's overwrite prompt
    (stripenn.py:12-42) needs the terminal's stdin, which spawned ranks do not have."""
    import socket
    import torch.multiprocessing as mp
    from .stripenn import addlog, makeOutDir
    if out[-1] != '/':
        out += '/'
    makeOutDir(out, force)
    addlog(cool, out, norm, chrom, canny, minL, maxW, maxpixel, numcores, pvalue, mask, bfilter)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    args = (cool, out, norm, chrom, canny, minL, maxW, maxpixel, numcores, pvalue, mask, slow, bfilter, seed, True)
    mp.spawn(_worker, args=(gpus, port, args), nprocs=gpus, join=True)
    return 0


# ---------------------------------------------------------------------------------------------------------------
# Persistent ranks: start-up (Python, torch, the gloo group, HIP context, workspaces: ~3 s per process) is paid ONCE
# for any number of `compute` runs -- one GPU finishes the mm10-size genome in under a second, so a fresh set of
# processes per run (launch_compute) can never pay off; a pool that stays alive across the samples of a study can.
def _pool_worker(rank, world, port, jobs, results, backend_factory):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('gloo', rank=rank, world_size=world)   # host objects only
    backend = None
    try:
        if backend_factory is not None:
            backend = backend_factory(rank)
        else:
            from .backend import HipBackend
            backend = HipBackend(rank)                             # raises without the HIP library / device: no fallback
        results.put(('ready', rank, os.getpid()))
        while True:
            job = jobs.get()
            if job is None:
                break
            t0 = time.time()
            try:
                sharded_compute(rank, world, *job, prepared=True, backend_factory=lambda r: backend)
                results.put(('done', rank, os.getpid(), time.time() - t0))
            except BaseException as e:      # noqa: BLE001 -- reported to the parent, which closes the pool
                results.put(('error', rank, os.getpid(), repr(e)))
                break
    finally:
        if backend is not None and hasattr(backend, 'close'):
            backend.close()
        try:
            dist.destroy_process_group()
        except Exception:      # noqa: BLE001
            pass


class ComputePool:
    """`gpus` rank processes that stay alive: `pool.compute(...)` (the arguments of stripenn.compute) may be called any
    number of times; every call is one sharded_compute over the same ranks, HIP contexts and workspaces."""

    def __init__(self, gpus, backend_factory=None, start_timeout=300):
        import socket
        import torch.multiprocessing as mp
        ctx = mp.get_context('spawn')
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        self.world = int(gpus)
        self.results = ctx.Queue()
        self.jobs = [ctx.Queue() for _ in range(self.world)]
        self.procs = [ctx.Process(target=_pool_worker, args=(r, self.world, port, self.jobs[r], self.results, backend_factory),
                                  daemon=True) for r in range(self.world)]
        for p in self.procs:
            p.start()
        self.pids = {}
        for msg in self._collect('ready', start_timeout):
            self.pids[msg[1]] = msg[2]

    def _collect(self, kind, timeout):
        import queue
        got, deadline = [], time.time() + timeout
        while len(got) < self.world:
            try:
                msg = self.results.get(timeout=0.5)
            except queue.Empty:
                dead = [r for r, p in enumerate(self.procs) if not p.is_alive()]
                if dead or time.time() > deadline:
                    self.close(kill=True)
                    raise RuntimeError('compute pool: rank(s) %s ended or timed out while waiting for %r' % (dead, kind))
                continue
            if msg[0] == 'error':
                self.close(kill=True)
                raise RuntimeError('compute pool: rank %d failed: %s' % (msg[1], msg[3]))
            if msg[0] == kind:
                got.append(msg)
        return got

    def compute(self, cool, out, norm, chrom, canny, minL, maxW, maxpixel, numcores, pvalue, mask, slow, bfilter, seed, force=True,
                timeout=3600):
        """One `compute` run on the pool's ranks; returns the slowest rank's seconds.  The output directory is prepared
        here, in the parent (the reference's overwrite prompt needs a terminal)."""
        from .stripenn import addlog, makeOutDir
        if out[-1] != '/':
            out += '/'
        makeOutDir(out, force)
        addlog(cool, out, norm, chrom, canny, minL, maxW, maxpixel, numcores, pvalue, mask, bfilter)
        job = (cool, out, norm, chrom, canny, minL, maxW, maxpixel, numcores, pvalue, mask, slow, bfilter, seed, True)
        for q in self.jobs:
            q.put(job)
        msgs = self._collect('done', timeout)
        assert {m[1]: m[2] for m in msgs} == self.pids, 'a rank was replaced between runs'
        return max(m[3] for m in msgs)

    def close(self, kill=False):
        for q in self.jobs:
            try:
                q.put(None)
            except Exception:      # noqa: BLE001
                pass
        for p in self.procs:
            p.join(0.1 if kill else 30)
            if p.is_alive():
                p.terminate()
        self.procs = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
