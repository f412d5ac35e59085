"""Deterministic synthetic Hi-C contact matrices (banded, Poisson-like, planted stripes).

Real .mcool inputs are not available offline, so every BASELINE.json config is
realised with this generator.  Every pixel is a pure
function of ``(seed, min(i, j), max(i, j))`` built from a 64-bit integer hash and
IEEE basic operations only (+ - * / sqrt), so any platform / numpy version
reproduces the same float64 matrix bit for bit, and any sub-block can be
produced without materialising the chromosome.

Only numpy is used: the golden-vector harness imports this file under
python3.9 / numpy 1.26 and the tests under python3.10 / numpy 2.2.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
BAND_LIMIT = 600  # pixels with |i-j| > BAND_LIMIT are exactly 0


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _u01(h, shift):
    """16-bit field of a hash as an integer-valued float64 in [0, 65535]."""
    return ((h >> np.uint64(shift)) & np.uint64(0xFFFF)).astype(np.float64)


class SynthChrom:
    """One synthetic chromosome: ``nbins`` bins, symmetric float64 contact matrix.

    value(i, j) = count(i, j) * w[i] * w[j]   (w = 1 when ``balanced`` is False)
    count ~ round(lam + sqrt(lam) * z), z ~ approx N(0, 1) (Irwin-Hall of 4 hash fields)
    lam(d) = 240 / (1 + d) + 1 for d = |i - j| <= 600, times ``stripe_gain`` on
    planted stripe pixels.  Bins listed in ``nan_bins`` return NaN rows / columns
    (they emulate bins whose balancing weight is NaN).
    """

    def __init__(self, nbins, seed, balanced=True, stripe_every=170, stripe_gain=2.5,
                 nan_frac=0.005, depth=1.0, count_div=1):
        # data regimes (round 6): depth scales lam (sequencing depth: relative noise ~ 1 / sqrt(depth)), count_div > 1 keeps
        # floor(count / count_div) (a shallow library: mostly 0 / 1 / 2 away from the diagonal); balanced=False gives raw
        # integer counts (`--norm None`).  The defaults leave every matrix of rounds 1-5 bit for bit what it was.
        self.depth, self.count_div = float(depth), int(count_div)
        self.nbins = int(nbins)
        self.seed = int(seed)
        self.balanced = bool(balanced)
        with np.errstate(over='ignore'):
            idx = np.arange(self.nbins, dtype=np.uint64)
            hb = _splitmix64(idx * np.uint64(0x2545F4914F6CDD1D) + np.uint64(self.seed * 7919 + 17))
            if balanced:
                self.w = 0.75 + _u01(hb, 0) / 131072.0  # in [0.75, 1.25)
            else:
                self.w = np.ones(self.nbins)
            nanflag = (_u01(hb, 16) < nan_frac * 65536.0)
            # planted stripes: anchor column a, width 3, length L, direction down/up
            self.stripes = []
            a = 60
            k = 0
            while a + 3 < self.nbins:
                hs = int(_splitmix64(np.array([self.seed * 104729 + k], dtype=np.uint64))[0])
                L = 40 + (hs & 0xFFFF) % 120
                down = ((hs >> 20) & 1) == 1
                self.stripes.append((a, 3, L, down))
                a += stripe_every
                k += 1
        self.nan_bins = np.nonzero(nanflag)[0]
        self.nanflag = nanflag
        self.stripe_gain = float(stripe_gain)
        # stripe lookup: for column c -> (row_lo, row_hi) inclusive of enriched rows (in the
        # orientation "vertical line at column c"); symmetric counterpart handled in block()
        self._s_lo = np.full(self.nbins, 1, dtype=np.int64)
        self._s_hi = np.full(self.nbins, 0, dtype=np.int64)
        for (a, wd, L, down) in self.stripes:
            for c in range(a, min(a + wd, self.nbins)):
                if down:
                    self._s_lo[c], self._s_hi[c] = a, min(a + L, self.nbins - 1)
                else:
                    self._s_lo[c], self._s_hi[c] = max(a - L, 0), a + wd - 1

    def counts(self, r0, r1, c0, c1):
        """Raw (unweighted) counts of rows [r0, r1) x cols [c0, c1): integer-valued float64, symmetric."""
        r = np.arange(r0, r1, dtype=np.int64)[:, None]
        c = np.arange(c0, c1, dtype=np.int64)[None, :]
        lo = np.minimum(r, c)
        hi = np.maximum(r, c)
        d = (hi - lo)
        with np.errstate(over='ignore'):
            h = _splitmix64((lo.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15))
                            ^ _splitmix64(hi.astype(np.uint64) + np.uint64(self.seed) * np.uint64(0xD1B54A32D192ED03)))
        usum = _u01(h, 0) + _u01(h, 16) + _u01(h, 32) + _u01(h, 48)
        z = (usum - 131070.0) / 37837.22  # sd of the 4-field sum = 65536/sqrt(3)
        lam = 240.0 / (1.0 + d.astype(np.float64)) + 1.0
        # stripe enrichment (vertical line at column x, rows y) and its mirror image
        x1, y1 = c, r
        in1 = (y1 >= self._s_lo[np.clip(x1, 0, self.nbins - 1)]) & (y1 <= self._s_hi[np.clip(x1, 0, self.nbins - 1)])
        x2, y2 = r, c
        in2 = (y2 >= self._s_lo[np.clip(x2, 0, self.nbins - 1)]) & (y2 <= self._s_hi[np.clip(x2, 0, self.nbins - 1)])
        lam = np.where(in1 | in2, lam * self.stripe_gain, lam)
        if self.depth != 1.0:
            lam = lam * self.depth
        cnt = np.floor(lam + np.sqrt(lam) * z + 0.5)
        cnt = np.where(cnt < 0.0, 0.0, cnt)
        if self.count_div > 1:
            cnt = np.floor(cnt / float(self.count_div))
        return np.where(d > BAND_LIMIT, 0.0, cnt)

    def block(self, r0, r1, c0, c1):
        """Dense float64 block rows [r0, r1) x cols [c0, c1) (bin indices, must be in range)."""
        cnt = self.counts(r0, r1, c0, c1)
        val = (cnt * self.w[r0:r1][:, None]) * self.w[c0:c1][None, :]
        if self.nan_bins.size:
            val = np.where(self.nanflag[r0:r1][:, None] | self.nanflag[c0:c1][None, :], np.nan, val)
        return val

    def band(self, halfwidth=512, r0=0, r1=None, chunk=4096):
        """Diagonal band rows [r0, r1): out[i - r0, hw + d] = M[i, i + d], d in [-hw, hw).

        Out-of-chromosome entries are 0 (never NaN).  This is the HBM layout the HIP
        library consumes (include/stripenn_hip.h, stp_band_upload).
        """
        hw = int(halfwidth)
        if r1 is None:
            r1 = self.nbins
        out = np.zeros((r1 - r0, 2 * hw), dtype=np.float64)
        for a in range(r0, r1, chunk):
            b = min(a + chunk, r1)
            ca, cb = max(a - hw, 0), min(b + hw, self.nbins)
            blk = self.block(a, b, ca, cb)
            rows = np.arange(a, b)[:, None]
            dd = np.arange(-hw, hw)[None, :]
            cols = rows + dd
            ok = (cols >= 0) & (cols < self.nbins)
            sub = blk[(rows - a), np.clip(cols - ca, 0, cb - ca - 1)]
            out[a - r0:b - r0] = np.where(ok, sub, 0.0)
        return out


class SynthSelector:
    """Stand-in for ``cooler.Cooler(...).matrix(balance=...)`` over synthetic chromosomes.

    Implements the two things the reference uses: ``fetch(region[, region2])`` with
    cooler's extent rule (region strings are 0-based half-open bp intervals;
    bins lo = start // binsize, hi = ceil(end / binsize)) and whole-chromosome fetch.
    """

    def __init__(self, chroms, resol):
        self.chroms = dict(chroms)  # name -> SynthChrom
        self.resol = int(resol)
        self.nfetch = 0

    def _extent(self, region):
        region = str(region)
        if ':' not in region:
            ch = self.chroms[region]
            return region, 0, ch.nbins
        name, rng = region.rsplit(':', 1)
        s, e = rng.replace(',', '').split('-')
        s, e = int(s), int(e)
        ch = self.chroms[name]
        if s < 0 or e > ch.nbins * self.resol or s > e:
            raise ValueError('Genomic region out of bounds: %s' % region)
        lo = s // self.resol
        hi = -(-e // self.resol)
        return name, lo, hi

    def fetch(self, region, region2=None):
        self.nfetch += 1
        n1, r0, r1 = self._extent(region)
        if region2 is None:
            n2, c0, c1 = n1, r0, r1
        else:
            n2, c0, c1 = self._extent(region2)
        if n1 != n2:
            raise ValueError('trans fetch not supported by the synthetic selector')
        ch = self.chroms[n1]
        if (r1 - r0) * (c1 - c0) <= (1 << 24):
            return ch.block(r0, r1, c0, c1)
        # whole-chromosome fetches (getQuantile_original): same array, built in row strips so that the
        # generator's temporaries stay small next to the dense result
        out = np.empty((r1 - r0, c1 - c0), dtype=np.float64)
        step = max(1, (1 << 24) // max(c1 - c0, 1))
        for a in range(r0, r1, step):
            b = min(a + step, r1)
            out[a - r0:b - r0] = ch.block(a, b, c0, c1)
        return out


def make_genome(sizes_bp, resol, seed0=1, names=None, **kw):
    """Build (names, sizes, selector) for a list of chromosome sizes in bp."""
    if names is None:
        names = ['chr%d' % (i + 1) for i in range(len(sizes_bp))]
    chroms = {}
    for k, (nm, sz) in enumerate(zip(names, sizes_bp)):
        nb = -(-int(sz) // int(resol))
        chroms[nm] = SynthChrom(nb, seed0 + k, **kw)
    return list(names), np.array(sizes_bp, dtype=np.int64), SynthSelector(chroms, resol)
