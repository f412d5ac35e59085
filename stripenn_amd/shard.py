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


def frame_spans(nframes, world):
    """Static work queue of the (chromosome x frame) grid: the frames of all chromosomes, laid end to end in
    chromosome order, are cut into `world` contiguous spans of equal length (+-1 frame), so a rank holds at most
    one partial chromosome at each end and the imbalance is 1 frame (mm10 at 5 kb over 8 GPUs: 331 vs 330.6).
    Every maxpixel level of a frame stays on the rank that holds the frame (they share the band reads).
    Returns, per rank, a list of (chromosome index, first frame, one past the last frame)."""
    total = int(sum(nframes))
    cuts = [(total * r) // world for r in range(world + 1)]
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


def sharded_compute(rank, world, cool, out, norm, chrom, canny, minL, maxW, maxpixel, numcores, pvalue, mask, slow,
                    bfilter, seed, force=True, backend_factory=None, write=True):
    """Body of one rank.  Returns (result_table, res_filter) on rank 0, (None, None) elsewhere."""
    from . import getStripe
    from .io import open_matrix
    from .stripenn import (RESULT_COLUMNS, addlog, finish_tables, makeOutDir, resolve_norm, select_chromosomes)
    np.seterr(divide='ignore', invalid='ignore')
    comm = _Comm(rank, world)
    if out[-1] != '/':
        out += '/'
    if rank == 0 and write:
        makeOutDir(out, force)
        addlog(cool, out, norm, chrom, canny, minL, maxW, maxpixel, numcores, pvalue, mask, bfilter)
    levels = list(map(float, maxpixel.split(',')))
    Lib = open_matrix(cool)
    normv = resolve_norm(Lib, norm)
    all_names, all_sizes, names, sizes = select_chromosomes(Lib, chrom)
    resol = Lib.binsize
    mine = lpt_assign(chrom_costs(sizes, resol), world)[rank]
    my_names = [names[i] for i in mine]
    my_sizes = np.asarray(sizes)[mine] if len(mine) else np.zeros(0, dtype=np.int64)
    backend = backend_factory(rank) if backend_factory is not None else None
    obj = getStripe.getStripe(Lib.matrix(balance=normv), resol, minL, maxW, canny, all_names, my_names, all_sizes,
                              my_sizes, numcores, bfilter, seed, backend=backend, device=rank)
    t0 = time.time()
    MP = (obj.getQuantile_slow if slow else obj.getQuantile_original)(Lib, my_names, levels) if my_names else {}
    EV = obj.mpmean()
    # --- background tables: per-chromosome parts are independent when numcores > 1 (the PRNG restarts
    # per chromosome); with numcores == 1 the stream runs on across chromosomes, so rank 0 does them all.
    if numcores == 1:
        bg = obj.nulldist() if rank == 0 else None
        bg = comm.allgather(bg)[0]
    else:
        cand = obj.null_candidates()
        share = [i for i in range(len(cand)) if i % world == rank]
        avail = dict(sum(comm.allgather([(i, obj.null_available_cols(cand[i])) for i in share]), []))
        cand2, samplesize = obj.null_samplesizes(cand, [avail[i] for i in range(len(cand))])
        share2 = [c for k, c in enumerate(cand2) if k % world == rank]
        parts = dict(sum(comm.allgather([(c, obj.null_tables(c, cand2, samplesize)) for c in share2]), []))
        bg = obj.null_concat([parts[c] for c in cand2])
    # --- candidate stripes, p-values, Stripiness of this rank's chromosomes
    table = pd.DataFrame(columns=RESULT_COLUMNS)
    if my_names:
        for i, perc in enumerate(levels):
            table = pd.concat([table, obj.extract(MP, i, perc, *bg)])
        table = obj.RemoveRedundant(df=table, by='pvalue')
        s = obj.scoringstripes(table, EV, mask)[0]
    else:
        s = []
    table = table.copy()
    table.insert(table.shape[1], '_stripiness', list(s), True)
    elapsed = time.time() - t0
    gathered = comm.allgather((rank, table, elapsed))
    result = (None, None)
    if rank == 0:
        merged = pd.concat([g[1] for g in sorted(gathered, key=lambda g: g[0])])
        lev_idx = {str(p * 100) + '%': i for i, p in enumerate(levels)}
        chr_idx = {str(n): i for i, n in enumerate(names)}
        key = [lev_idx[str(m)] * (len(names) + 1) + chr_idx[str(c)] for m, c in zip(merged['maxpixel'], merged['chr'])]
        merged = merged.iloc[np.argsort(np.asarray(key, dtype=np.int64), kind='stable')]
        stri = merged['_stripiness'].tolist()
        merged = merged.drop(columns=['_stripiness'])
        result_table, res_filter = finish_tables(merged, stri, pvalue)
        if write:
            result_table.to_csv(out + 'result_unfiltered.tsv', sep='\t', header=True, index=False)
            res_filter.to_csv(out + 'result_filtered.tsv', sep='\t', header=True, index=False)
            with open(out + 'stripenn.log', 'a') as f:
                f.write('gpus: %d\n' % world)
                for g in sorted(gathered, key=lambda g: g[0]):
                    f.write('rank %d: %.2f s\n' % (g[0], g[2]))
        result = (result_table, res_filter)
    if backend is None:
        obj.backend.close()
    comm.barrier()
    return result


def _worker(rank, world, port, args):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('gloo', rank=rank, world_size=world)   # host objects only
    try:
        sharded_compute(rank, world, *args)
    finally:
        dist.destroy_process_group()


def launch_compute(cool, out, norm, chrom, canny, minL, maxW, maxpixel, numcores, pvalue, mask, slow, bfilter, seed, force,
                   gpus):
    """One process per GPU (torch.multiprocessing spawn); rank r drives HIP device r."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    args = (cool, out, norm, chrom, canny, minL, maxW, maxpixel, numcores, pvalue, mask, slow, bfilter, seed, force)
    mp.spawn(_worker, args=(gpus, port, args), nprocs=gpus, join=True)
    return 0
