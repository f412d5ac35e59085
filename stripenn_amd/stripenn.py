"""`stripenn compute` driver: same arguments, log file, chromosome filter, step order and TSV outputs
as the reference's stripenn.compute (src/stripenn/stripenn.py:64-163); the five steps run through
the MI355X stripe engine (stripenn_amd/getStripe.py).  Extra keywords (not in the reference):
`force` (non-interactive overwrite of the output directory), `device`, and `gpus` > 1 to shard the
chromosome x maxpixel grid over several GPUs (stripenn_amd/shard.py)."""
import os
import shutil
import sys
import time
import warnings

import numpy as np
import pandas as pd

from . import getStripe
from .io import open_matrix

RESULT_COLUMNS = ['chr', 'pos1', 'pos2', 'chr2', 'pos3', 'pos4', 'length', 'width', 'total', 'Mean', 'maxpixel', 'num',
                  'start', 'end', 'x', 'y', 'h', 'w', 'medpixel', 'pvalue']
HELPER_COLUMNS = ['total', 'num', 'start', 'end', 'x', 'y', 'h', 'w', 'medpixel']


def _empty_directory(path):
    """Remove every entry of `path` (not the directory itself); an entry that cannot be removed is reported and skipped."""
    with os.scandir(path) as entries:
        for entry in entries:
            try:
                if entry.is_dir(follow_symlinks=False):
                    shutil.rmtree(entry.path)
                else:
                    os.unlink(entry.path)
            except OSError as e:
                print('Failed to delete %s with the reason: %s' % (entry.path, e))


def makeOutDir(outdir, force=False):
    """Prepare the output directory with the behaviour of the reference's helper (stripenn.py:12-42): a missing directory is
    created; an existing one is emptied after the user answers Y to the same prompt (force=True answers for them), kept and
    the run ended on n, and the run ended on anything else."""
    outdir = os.path.join(outdir, '')                      # trailing separator, as the messages print it
    if not os.path.exists(outdir):
        os.makedirs(outdir, exist_ok=True)
        return
    answer = 'y'
    if not force:
        print('\n%s exists. Do you want to remove all files and save new results in this folder? [Y/n]' % outdir)
        answer = input()
    if answer in ('Y', 'y'):
        print('All directories and files in %s will be deleted.' % outdir)
        _empty_directory(outdir)
        return
    print('Input another output directory. Exit.' if answer in ('n', 'N') else 'Type Y or n.\nExit.')
    sys.exit()


def addlog(cool, out, norm, chrom, canny, minL, maxW, maxpixel, numcores, pvalue, mask, bfilter):
    """stripenn.py:45-61 (same keys, same order)."""
    if out[-1] != '/':
        out += '/'
    with open(out + 'stripenn.log', 'w') as f:
        for k, v in (('cool', cool), ('out', out), ('norm', norm), ('chrom', chrom), ('canny', canny), ('minL', minL),
                     ('maxW', maxW), ('maxpixel', maxpixel), ('num_cores', numcores), ('pvalue', pvalue), ('mask', mask),
                     ('blur filter', bfilter)):
            f.write('%s: %s\n' % (k, str(v)))


def select_chromosomes(Lib, chrom):
    """stripenn.py:94-116"""
    all_chromnames = list(Lib.chromnames)
    all_chromsizes = np.asarray(Lib.chromsizes)
    keep = [i for i in range(len(all_chromnames))
            if 'JH5' not in all_chromnames[i] and 'GL4' not in all_chromnames[i] and 'RANDOM' not in all_chromnames[i]
            and all_chromnames[i] not in ('M', 'chrM', 'Y', 'chrY')]
    all_chromnames = [all_chromnames[i] for i in keep]
    all_chromsizes = all_chromsizes[keep]
    chromnames, chromsizes = all_chromnames, all_chromsizes
    if len(all_chromnames) == 0:
        sys.exit('Exit: All chromosomes are shorter than 50kb.')
    chroms = chrom.split(',')
    if chroms[0] != 'all':
        idx = []
        warnflag = False
        for item in chroms:
            if item in all_chromnames:
                idx.append(all_chromnames.index(item))
            else:
                warnings.warn('\nThere is no chromosomes called ' + str(item) +
                              ' in the provided .cool file or it is shorter than 50kb.')
                warnflag = True
        if warnflag:
            warnings.warn('\nThe possible chromosomes are: ' + ', '.join(all_chromnames))
        chromnames = chroms
        chromsizes = all_chromsizes[idx]
    return all_chromnames, all_chromsizes, chromnames, chromsizes


def resolve_norm(Lib, norm, weight_is_true=True):
    """stripenn.py:81-92"""
    PossibleNorm = Lib.bins().columns
    if norm == 'None':
        return False
    if weight_is_true and norm == 'weight':
        return True
    if norm not in PossibleNorm:
        print('Possible normalization methods are:')
        print('None')
        for n in range(3, len(PossibleNorm)):
            print(PossibleNorm[n])
        print('Invalid normalization method. Normalization method is forced to None')
        return False
    return norm


def _write_tsv_native(df, path):
    """The table through the library's formatter (stp_format_tsv: host code, no device): int, float and plain string columns
    only.  Returns False -- nothing written -- for anything else, and when the library cannot be loaded."""
    import ctypes as C
    try:
        from . import hip
        L = hip.load()
        fmt = L.stp_format_tsv
    except Exception:      # noqa: BLE001 -- no library on this box: the Python writer below
        return False
    n = len(df)
    kinds, arrs, tabs, offs, nstr, width = [], [], [], [], [], 1
    for name in df.columns:
        if any(ch in str(name) for ch in '\t"\n\r'):
            return False
        a = df[name].to_numpy()
        if a.ndim != 1:                          # (duplicate column names: pandas)
            return False
        if a.dtype.kind == 'O':
            kind = pd.api.types.infer_dtype(a, skipna=False)
            if kind == 'integer':
                try:
                    a = a.astype(np.int64)
                except (OverflowError, TypeError, ValueError):
                    return False
            elif kind == 'floating':
                a = a.astype(np.float64)
            elif kind != 'string':
                return False
        if a.dtype.kind == 'f':
            kinds.append(1); arrs.append(np.ascontiguousarray(a, dtype=np.float64)); tabs.append(None); offs.append(None); nstr.append(0)
            width += 26
        elif a.dtype.kind in 'iu' and not (a.dtype.kind == 'u' and a.dtype.itemsize == 8):
            kinds.append(0); arrs.append(np.ascontiguousarray(a, dtype=np.int64)); tabs.append(None); offs.append(None); nstr.append(0)
            width += 21
        elif a.dtype.kind == 'O':
            codes, uniq = pd.factorize(a)
            blobs = [str(u).encode('utf-8') for u in uniq]
            if any(any(ch in b for ch in (b'\t', b'"', b'\n', b'\r')) for b in blobs):
                return False                     # a field that needs quoting: pandas
            off = np.zeros(len(blobs) + 1, dtype=np.int64)
            if blobs:
                off[1:] = np.cumsum([len(b) for b in blobs])
            kinds.append(2); arrs.append(np.ascontiguousarray(codes, dtype=np.int32)); tabs.append(b''.join(blobs) or b'\0'); offs.append(off)
            nstr.append(len(blobs))
            width += max([len(b) for b in blobs] or [0]) + 1
        else:
            return False
    nc = len(kinds)
    if nc == 0:
        return False
    vp = C.c_void_p
    kind_a = (C.c_int32 * nc)(*kinds)
    data_a = (vp * nc)(*[x.ctypes.data for x in arrs])
    tab_a = (C.c_char_p * nc)(*tabs)
    off_a = (vp * nc)(*[None if o is None else o.ctypes.data for o in offs])
    nstr_a = (C.c_int32 * nc)(*nstr)
    cap = max(1, n * width)
    buf = np.empty(cap, dtype=np.uint8)
    out_len = C.c_int64(0)
    fmt.argtypes = [C.c_int32, vp, vp, vp, vp, vp, C.c_int64, vp, C.c_int64, C.POINTER(C.c_int64)]
    fmt.restype = C.c_int
    rc = fmt(nc, C.cast(kind_a, vp), C.cast(data_a, vp), C.cast(tab_a, vp), C.cast(off_a, vp), C.cast(nstr_a, vp), n,
             buf.ctypes.data, cap, C.byref(out_len))
    if rc != 0:
        return False
    with open(path, 'wb') as f:
        f.write(('\t'.join(str(c) for c in df.columns) + '\n').encode('utf-8'))
        f.write(memoryview(buf)[:out_len.value])
    return True


def write_tsv(df, path):
    """`df.to_csv(path, sep='\t', header=True, index=False)` (stripenn.py:156-157, score.py:60), byte for byte, without
    pandas' per-column `astype(str)` (numpy's fixed-width string arrays cost ~1 us per float: 0.4 s for the 18 x 40 000
    table `score` writes).  Floats are written by Python's shortest round-trip repr -- the digits numpy's `astype(str)`
    produces -- NaN / None as the empty field; anything this writer is not sure about goes through pandas itself.
    Tables of int / float / plain string columns -- every table the drivers write -- are formatted by the library
    (stp_format_tsv, round 6: 0.060 -> 0.012 s for the two tables of a genome); the Python writer below stays for the rest."""
    if not df.columns.is_unique:                  # (never from the drivers; both writers below address columns by name)
        df.to_csv(path, sep='\t', header=True, index=False)
        return
    if os.environ.get('STP_TSV', '') != 'python' and _write_tsv_native(df, path):
        return
    cols = []
    try:
        for name in df.columns:
            a = df[name].to_numpy()
            if a.dtype.kind == 'f':
                v = a.astype(np.float64).tolist()
                cols.append(['' if x != x else repr(x) for x in v])
            elif a.dtype.kind in 'iu':
                cols.append(list(map(str, a.tolist())))
            elif a.dtype.kind == 'O':
                kind = pd.api.types.infer_dtype(a, skipna=False)
                if kind == 'string':
                    out = a.tolist()
                    blob = '\x00'.join(out)
                    if '\t' in blob or '"' in blob or '\n' in blob or '\r' in blob:
                        raise ValueError('field needs quoting')
                    cols.append(out)
                elif kind == 'integer':
                    cols.append(list(map(str, a.tolist())))
                elif kind == 'floating':
                    cols.append(['' if x != x else repr(x) for x in a.astype(np.float64).tolist()])
                else:                                     # mixed cells: one by one
                    out = []
                    for x in a.tolist():
                        if x is None or (isinstance(x, float) and x != x):
                            out.append('')
                        elif isinstance(x, (str, int, float)) and not isinstance(x, bool):
                            if isinstance(x, str) and any(ch in x for ch in '\t"\n\r'):
                                raise ValueError('field needs quoting')
                            out.append(repr(float(x)) if isinstance(x, float) else str(x))
                        else:
                            raise ValueError('unsupported cell type %r' % type(x))
                    cols.append(out)
            else:
                raise ValueError('unsupported column dtype %s' % a.dtype)
        if any(any(ch in str(c) for ch in '\t"\n\r') for c in df.columns):
            raise ValueError('header needs quoting')
    except ValueError:
        df.to_csv(path, sep='\t', header=True, index=False)
        return
    lines = ['\t'.join(str(c) for c in df.columns)]
    lines.extend(map('\t'.join, zip(*cols)) if cols else [])
    with open(path, 'w', newline='') as f:
        f.write('\n'.join(lines))
        f.write('\n')


def finish_tables(result_table, stripiness, pcut):
    """stripenn.py:149-152"""
    result_table = result_table.drop(columns=[c for c in HELPER_COLUMNS if c in result_table.columns])   # (the sharded driver's ranks have dropped them already)
    result_table.insert(result_table.shape[1], 'Stripiness', stripiness, True)
    res_filter = result_table[result_table['pvalue'] < pcut]
    res_filter = res_filter.sort_values(by=['Stripiness'], ascending=False)
    return result_table, res_filter


def compute(cool, out, norm, chrom, canny, minL, maxW, maxpixel, numcores, pvalue, mask, slow, bfilter, seed,
            force=False, device=0, gpus=1, backend=None):
    np.seterr(divide='ignore', invalid='ignore')
    t_start = time.time()
    if out[-1] != '/':
        out += '/'
    if gpus > 1:
        if device not in (0, None):
            raise ValueError('--device selects the GPU of a single-GPU run; with --gpus %d rank r drives device r' % gpus)
        from . import shard
        return shard.launch_compute(cool, out, norm, chrom, canny, minL, maxW, maxpixel, numcores, pvalue, mask, slow,
                                    bfilter, seed, force, gpus)
    makeOutDir(out, force)
    addlog(cool, out, norm, chrom, canny, minL, maxW, maxpixel, numcores, pvalue, mask, bfilter)
    maxpixel = list(map(float, maxpixel.split(',')))
    minH, core, pcut = minL, numcores, pvalue
    print('Result will be stored in %s' % out)

    Lib = open_matrix(cool)
    norm = resolve_norm(Lib, norm)
    all_chromnames, all_chromsizes, chromnames, chromsizes = select_chromosomes(Lib, chrom)
    unbalLib = Lib.matrix(balance=norm)
    resol = Lib.binsize
    obj = getStripe.getStripe(unbalLib, resol, minH, maxW, canny, all_chromnames, chromnames, all_chromsizes, chromsizes,
                              core, bfilter, seed, backend=backend, device=device)
    obj.eager_search = True      # searches start while the quantile loop is still uploading the next chromosomes
    print('1. Maximum pixel value calculation ...')
    if slow:
        print('1.1 Slowly estimating Maximum pixel values...')
        MP = obj.getQuantile_slow(Lib, chromnames, maxpixel)
    else:
        MP = obj.getQuantile_original(Lib, chromnames, maxpixel)
    print('2. Expected value calculation ...')
    EV = obj.mpmean()
    print('3. Background distribution estimation ...')
    bgleft_up, bgright_up, bgleft_down, bgright_down = obj.nulldist()
    print('4. Finding candidate stripes from each chromosome ...')
    levels = []
    for i in range(len(maxpixel)):
        perc = maxpixel[i]
        levels.append(obj.extract(MP, i, perc, bgleft_up, bgright_up, bgleft_down, bgright_down))
    levels = [t for t in levels if len(t)]                    # (one concat; empty levels carry no dtypes)
    result_table = pd.concat(levels) if levels else pd.DataFrame(columns=RESULT_COLUMNS)
    result_table = obj.RemoveRedundant(df=result_table, by='pvalue')
    print('5. Stripiness calculation ...')
    s = obj.scoringstripes(result_table, EV, mask)[0]
    result_table, res_filter = finish_tables(result_table, s, pcut)
    write_tsv(result_table, out + 'result_unfiltered.tsv')
    write_tsv(res_filter, out + 'result_filtered.tsv')
    print('\n' + str(round((time.time() - t_start) / 60, 3)) + 'min taken.')
    print('Check the result stored in %s' % out)
    return 0
