"""`stripenn score` driver: the reference's score.getScore (src/stripenn/score.py:7-61) on the
MI355X stripe engine -- same table sniffing, chromosome filter, fixed parameters (minH 10, maxW 8,
canny 2.5, bfilter 1), call order and the six appended columns."""
import numpy as np
import pandas as pd

from . import getStripe
from .io import open_matrix
from .stripenn import resolve_norm, write_tsv


def _halfwidth_for(table, resol):
    """Band halfwidth that holds every pixel the p-value / Stripiness / mean kernels read for these stripes: a
    stripe's rows against its columns widened by the 50 kb flanks (user tables may hold stripes far longer than
    the 400-bin frames `compute` produces).  None: the default is enough."""
    if len(table) == 0:
        return None
    bs = int(50000 / resol)
    p = [np.trunc(np.asarray(table[k], dtype=np.float64)).astype(np.int64) // resol for k in ('pos1', 'pos2', 'pos3', 'pos4')]
    reach = int(max(np.abs(p[1] + bs + 1 - p[2]).max(), np.abs(p[3] + 1 - (p[0] - bs)).max(),
                    np.abs(p[0] - bs - p[2]).max(), np.abs(p[3] - p[1] - bs).max()))
    need = -(-(reach + 2 * bs + 2) // 64) * 64
    return need if need > getStripe.HALFWIDTH else None


def getScore(cool, coordinates, norm, numcores, seed, out, mask='0', device=0, backend=None, halfwidth=None):
    bfilter = 1
    print('Run score function')
    table = pd.read_csv(coordinates, header=None, sep='\t')
    el = table.iloc[0]
    if type(el[1]) == str or type(el[2]) == str:          # score.py:13-20: a header line is present
        table = pd.read_csv(coordinates, header=0, sep='\t')
    table.columns = ['chr', 'pos1', 'pos2', 'chr2', 'pos3', 'pos4'] + table.columns[6:].tolist()

    Lib = open_matrix(cool)
    norm = resolve_norm(Lib, norm, weight_is_true=False)    # score.py:27-35 has no 'weight' -> True mapping
    names = list(Lib.chromnames)
    sizes = np.asarray(Lib.chromsizes)
    keep = [i for i in range(len(names)) if names[i] != 'Y' and 'JH' not in names[i] and 'RANDOM' not in names[i]]
    all_chromnames = [names[i] for i in keep]
    all_chromsizes = np.array([sizes[i] for i in keep])
    big = np.where(all_chromsizes > 1000000)[0]
    all_chromnames = [all_chromnames[i] for i in big]
    all_chromsizes = all_chromsizes[big]
    unbalLib = Lib.matrix(balance=norm)
    resol = Lib._info['bin-size']
    if halfwidth is None:
        halfwidth = _halfwidth_for(table, resol)
    obj = getStripe.getStripe(unbalLib, resol, 10, 8, 2.5, all_chromnames, all_chromnames, all_chromsizes,
                              all_chromsizes, numcores, bfilter, seed, backend=backend, device=device, halfwidth=halfwidth)
    EV = obj.mpmean()
    bg = obj.nulldist()
    pval = obj.pvalue(*bg, table)
    table.insert(table.shape[1], 'pvalue_added', pval, True)
    MEAN, SUM = obj.getMean(table)
    s, MEANOE, TOTALOE = obj.scoringstripes(table, EV, mask)
    table.insert(table.shape[1], 'Stripiness_added', s, True)
    table.insert(table.shape[1], 'O_Mean_added', MEAN, True)
    table.insert(table.shape[1], 'O_Sum_added', SUM, True)
    table.insert(table.shape[1], 'O/E_Mean_added', MEANOE, True)
    table.insert(table.shape[1], 'O/E_Total_added', TOTALOE, True)
    write_tsv(table, out)
    return table
