"""Python side of the CPU oracle (TEST INFRASTRUCTURE ONLY -- see oracle/README.md).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It wraps oracle/libstripe_oracle.so (C restatement of the image chain) and restates the
host-side frame logic and the score path of the reference with plain numpy.
All citations are to /root/reference/src/stripenn/.
"""
import ctypes as C
import math
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class SoRec(C.Structure):
    _fields_ = [('b_index', C.c_int32), ('ud', C.c_int32), ('x', C.c_int32), ('y', C.c_int32),
                ('w', C.c_int32), ('h', C.c_int32), ('total', C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, 'libstripe_oracle.so')
    srcs = [os.path.join(_HERE, f) for f in ('stripe_oracle.c', 'score_oracle.c')]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'libstripe_oracle.so'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, 'libstripe_oracle.so')
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        dp = np.ctypeslib.ndpointer(np.float64, flags='C')
        fp = np.ctypeslib.ndpointer(np.float32, flags='C')
        bp = np.ctypeslib.ndpointer(np.uint8, flags='C')
        ip = np.ctypeslib.ndpointer(np.int32, flags='C')
        L.so_gplane.argtypes = [dp, C.c_int64, C.c_double, dp]
        L.so_gray.argtypes = [dp, C.c_int, C.c_double, C.c_int, fp]
        L.so_canny.argtypes = [fp, C.c_int, dp, C.c_int, bp] + [C.c_void_p] * 5
        L.so_vertical_line.argtypes = [bp, C.c_int, bp]
        L.so_columns.argtypes = [bp, C.c_int, C.c_int, ip, ip, ip]
        L.so_join_dbg.argtypes = [bp, bp, C.c_int, C.c_int, C.c_int, C.c_int, bp, ip, ip, ip, ip, C.c_int]
        L.so_join_dbg.restype = C.c_int
        L.so_stripe_search.argtypes = [dp, C.c_int, C.c_double, dp, C.c_int, C.c_int, dp, C.c_int, C.c_int,
                                       C.c_int, C.POINTER(SoRec), C.c_int]
        L.so_stripe_search.restype = C.c_int
        _LIB = L
    return _LIB


def gauss_weights(sigma, truncate=4.0):
    """scipy 1.7.1 ndimage.filters._gaussian_kernel1d(sigma, 0, int(truncate*sigma+0.5))[::-1]
    (filters.py:179-192, gaussian_filter1d :256-260), same numpy calls in the same order."""
    sd = float(sigma)
    lw = int(truncate * sd + 0.5)
    sigma2 = sigma * sigma
    x = np.arange(-lw, lw + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    phi_x = phi_x / phi_x.sum()
    return np.ascontiguousarray(phi_x[::-1]), lw


def brightness_levels():
    """getStripe.py:898"""
    return np.arange(0.5, 1.01, 0.1)


# ------------------------------------------------------------------ stage wrappers
def gplane(D, M):
    D = np.ascontiguousarray(D, dtype=np.float64)
    g = np.empty_like(D)
    lib().so_gplane(D, D.size, float(M), g)
    return g


def gray(g, b, bf=3):
    S = g.shape[0]
    out = np.empty((S, S), dtype=np.float32)
    lib().so_gray(np.ascontiguousarray(g), S, float(b), int(bf), out)
    return out


def gray_alt(g, b, bf=3, box_fma=False, order=0):
    """so_gray_alt: the grey image under the deterministic arithmetic alternatives of a real OpenCV build
    (tools/cv2_ambiguity.py only)."""
    S = g.shape[0]
    out = np.empty((S, S), dtype=np.float32)
    L = lib()
    L.so_gray_alt.argtypes = [np.ctypeslib.ndpointer(np.float64, flags='C'), C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                              np.ctypeslib.ndpointer(np.float32, flags='C')]
    L.so_gray_alt.restype = None
    L.so_gray_alt(np.ascontiguousarray(g), S, float(b), int(bf), int(bool(box_fma)), int(order), out)
    return out


def canny(gray_img, gw, gr, debug=False):
    S = gray_img.shape[0]
    edges = np.zeros((S, S), dtype=np.uint8)
    if not debug:
        lib().so_canny(np.ascontiguousarray(gray_img), S, gw, gr, edges, None, None, None, None, None)
        return edges
    sm = np.empty((S, S)); is_ = np.empty((S, S)); js = np.empty((S, S)); mag = np.empty((S, S))
    cls = np.zeros((S, S), dtype=np.uint8)
    lib().so_canny(np.ascontiguousarray(gray_img), S, gw, gr, edges, sm.ctypes.data, is_.ctypes.data,
                   js.ctypes.data, mag.ctypes.data, cls.ctypes.data)
    return edges, dict(smoothed=sm, isobel=is_, jsobel=js, mag=mag, cls=cls)


def vertical_line(edges):
    S = edges.shape[0]
    v = np.zeros((S, S), dtype=np.uint8)
    lib().so_vertical_line(np.ascontiguousarray(edges, dtype=np.uint8), S, v)
    return v


def columns(vert, minH):
    S = vert.shape[0]
    t = np.zeros(S, np.int32); e = np.zeros(S, np.int32); ud = np.zeros(S, np.int32)
    lib().so_columns(np.ascontiguousarray(vert, dtype=np.uint8), S, int(minH), t, e, ud)
    return t, e, ud


def join_dbg(edges, vert, ud, minH, maxW):
    S = edges.shape[0]
    tm = np.zeros((S, S), np.uint8)
    ox = np.zeros(512, np.int32); oy = np.zeros(512, np.int32); ow = np.zeros(512, np.int32); oh = np.zeros(512, np.int32)
    n = lib().so_join_dbg(np.ascontiguousarray(edges, np.uint8), np.ascontiguousarray(vert, np.uint8), S, ud,
                          minH, maxW, tm, ox, oy, ow, oh, 512)
    return tm, np.stack([ox[:n], oy[:n], ow[:n], oh[:n]], axis=1)


def stripe_search(D, M, sigma=2.0, minH=10, maxW=8, bf=3, bvals=None, gw=None, cap=4096):
    """Records of one compacted frame in reference order: rows of (b_index, ud, x, y, w, h), totals."""
    D = np.ascontiguousarray(D, dtype=np.float64)
    S = D.shape[0]
    if bvals is None:
        bvals = brightness_levels()
    bvals = np.ascontiguousarray(bvals, dtype=np.float64)
    if gw is None:
        gw, gr = gauss_weights(sigma)
    else:
        gr = (len(gw) - 1) // 2
    buf = (SoRec * cap)()
    n = lib().so_stripe_search(D, S, float(M), bvals, len(bvals), int(bf), np.ascontiguousarray(gw), gr,
                               int(minH), int(maxW), buf, cap)
    if n > cap:
        return stripe_search(D, M, sigma, minH, maxW, bf, bvals, gw, cap=n)
    recs = np.array([(r.b_index, r.ud, r.x, r.y, r.w, r.h) for r in buf[:n]], dtype=np.int64).reshape(n, 6)
    tot = np.array([r.total for r in buf[:n]], dtype=np.float64)
    return recs, tot


# ------------------------------------------------------------------ frame logic (getStripe.py:792-825)
def frame_bounds(idx, rowsize):
    start = idx * 200 - 100
    end = (idx + 1) * 200 + 99
    if end >= rowsize:
        end = rowsize - 1
    if idx == 0:
        start = 0
    return start, end


def frame_dense(fetch_block, start, end):
    """D after NaN->0 and zero-column removal.  fetch_block(r0, r1, c0, c1) -> dense f64 (bin indices)."""
    D = np.array(fetch_block(start, end + 1, start, end + 1), dtype=np.float64)
    D[np.isnan(D)] = 0
    colsum = np.sum(D, axis=0)
    nz = np.where(colsum != 0)[0]
    return D, nz


# ====================================================================== score path (numeric restatements)
# These take the same explicit integer inputs as the C ABI (include/stripenn_hip.h) and use numpy
# itself for the reductions the reference performs with numpy, so their summation order is the
# reference's by construction.  `M(r0, r1, c0, c1)` returns the dense block with NaN preserved.

def _lib_score():
    L = lib()
    if not hasattr(L, '_score_ready'):
        L.so_null_windows.argtypes = [np.ctypeslib.ndpointer(np.float64, flags='C'), C.c_int64, C.c_int64,
                                      np.ctypeslib.ndpointer(np.int64, flags='C'), C.c_int64, C.c_int, C.c_int] + \
                                     [np.ctypeslib.ndpointer(np.float64, flags='C')] * 4
        L.so_block_mean.restype = C.c_double
        L.so_block_mean.argtypes = [C.c_void_p] + [C.c_int64] * 6
        L._score_ready = True
    return L


def diag_sums(M, nrows):
    """getStripe.py:198-207 per 400-row frame: row-ordered sums of each diagonal 0..399 and term counts."""
    n400 = -(-nrows // 400)
    ps = np.zeros((n400, 400)); pc = np.zeros((n400, 400), np.int64)
    for f in range(n400):
        r0, r1 = f * 400, min((f + 1) * 400, nrows)
        c1 = min((f + 2) * 400, nrows)
        cfm = np.array(M(r0, r1, r0, c1), dtype=np.float64)
        cfm[np.isnan(cfm)] = 0
        for j in range(400):
            d = np.diagonal(cfm, offset=j)          # cfm[i, i+j] while i+j < cols
            if len(d):
                ps[f, j] = np.cumsum(d)[-1]           # sequential accumulation, like the Python loop
                pc[f, j] = len(d)
    return ps, pc


def null_windows(M, row0, nrow, col0, ncol, xs, yoff, bs):
    """getStripe.py:347-378: the four centre-minus-flank tables (400 x len(xs)) of one unit matrix."""
    mat = np.ascontiguousarray(M(row0, row0 + nrow, col0, col0 + ncol), dtype=np.float64)
    mat[np.isnan(mat)] = 0
    xs = np.ascontiguousarray(xs, dtype=np.int64)
    n = len(xs)
    out = [np.zeros((400, n)) for _ in range(4)]
    if n:
        _lib_score().so_null_windows(mat, nrow, ncol, xs, n, int(bs), int(yoff), *out)
    return out


def pvalue_one(mat, bs, mode, upbase, fixed_row, fixed_tab, bg):
    """getStripe.py:561-605 for one stripe.  mat = fetched rows x (cols incl. flanks), NaN preserved;
    bg = (left_up, right_up, left_down, right_down)."""
    mat = np.array(mat, dtype=np.float64)
    parts = []
    for sl in (slice(bs, -bs), slice(None, bs), slice(-bs, None)):
        v = mat[:, sl]
        v[np.isnan(v)] = 0
        with np.errstate(invalid='ignore', divide='ignore'):
            parts.append(np.mean(v, axis=1))
    center, left, right = parts
    ld, rd = center - left, center - right
    pv = []
    for j in range(len(center)):
        if mode == 0:
            d, tab = min(j, 399), 1
        elif mode == 1:
            d = upbase - j - 1
            d, tab = (399 if d >= 400 else d), 0
        else:
            d, tab = fixed_row, fixed_tab
        bl = bg[2 if tab else 0][d, :]
        br = bg[3 if tab else 1][d, :]
        p1 = np.count_nonzero(bl >= ld[j]) / np.count_nonzero(~np.isnan(bl))
        p2 = np.count_nonzero(br >= rd[j]) / np.count_nonzero(~np.isnan(br))
        p = max(p1, p2)
        if p == 0:
            p = 1 / len(bl)
        pv.append(p)
    return float(np.median(pv))


def _expected(exval, x0, nx, y0, ny):
    """expecMatrix (getStripe.py:641-659): exval[min(|x - y|, 399)]"""
    idx = np.abs(np.arange(x0, x0 + nx)[None, :] - np.arange(y0, y0 + ny)[:, None])
    idx[idx >= 400] = 399
    return np.asarray(exval, dtype=np.float64)[idx]


def stripiness_one(obs, exval, ex0, ey0, mirror, mcol, mrow):
    """getStripe.py:684-759 for one stripe.  obs = [centre, left, right] observed blocks (NaN kept),
    ex0[b] = first x index of block b's expected matrix, mcol[b] = (lo, hi) masked relative columns
    (lo > hi: none), mrow likewise.  Returns (g, centerMean, centerTotal)."""
    blocks = []
    with np.errstate(divide='ignore', invalid='ignore'):
        for b in range(3):
            o = np.array(obs[b], dtype=np.float64)
            e = _expected(exval, ex0[b], o.shape[1], ey0, o.shape[0]) + .00000001
            m = np.divide(o, e)
            if mcol[b][0] <= mcol[b][1]:
                m[:, mcol[b][0]:mcol[b][1] + 1] = np.nan
            if mrow[0] <= mrow[1]:
                m[mrow[0]:mrow[1] + 1, :] = np.nan
            blocks.append(m)
        rowdel = []
        for b in range(3):
            dead = [x for x in range(blocks[b].shape[1]) if np.isnan(blocks[b][:, x]).all()]
            if dead:
                blocks[b] = np.delete(blocks[b], dead, axis=1)
                rowdel += dead if not mirror else [blocks[b].shape[0] - 1 - x for x in dead]
        rowdel = np.unique(rowdel)
        if len(rowdel):
            blocks = [np.delete(m, rowdel.astype(int), axis=0) for m in blocks]
        for m in blocks:
            m[np.isnan(m)] = 0
        cm, lm, rm = [np.mean(m, axis=1) for m in blocks]
        center = blocks[0]
        rows = np.where(~np.isnan(center))[0]
        ctot = np.sum(center[rows])
        cmean = np.mean(center[rows])
        n = len(cm)
        gxl, gxr, gy = [], [], []
        for i in range(1, n - 1):
            t = 0
            t += (-1 * lm[i - 1] + -2 * lm[i]) + -1 * lm[i + 1]
            t += (1 * cm[i - 1] + 2 * cm[i]) + 1 * cm[i + 1]
            gxl.append(t)
            t = 0
            t += (1 * cm[i - 1] + 2 * cm[i]) + 1 * cm[i + 1]
            t += (-1 * rm[i - 1] + -2 * rm[i]) + -1 * rm[i + 1]
            gxr.append(t)
            t = 0
            t += (1 * lm[i - 1] + 0 * lm[i]) + -1 * lm[i + 1]
            t += (2 * cm[i - 1] + 0 * cm[i]) + -2 * cm[i + 1]
            t += (1 * rm[i - 1] + 0 * rm[i]) + -1 * rm[i + 1]
            if t < 0:
                t *= -1
            gy.append(t)
        gx = np.minimum(gxl, gxr)
        diff = [a - b for a, b in zip(gx, gy)]
        diff = [x for x in diff if x >= 0 or x < 0]
        g = float(np.nanmedian(cm) * np.mean(diff))
    return g, cmean, ctot


def stripe_mean_one(block):
    """getStripe.py:518-521"""
    with np.errstate(invalid='ignore'):
        return np.nanmean(block), np.nansum(block)


def medpixel(D):
    """getStripe.py:885"""
    return float(np.quantile(D[D > 0], 0.5))


# ------------------------------------------------------------------ band from cooler's pixel table
def band_from_pixels(bin1, bin2, count, weight, lo, nrows, hw):
    """CPU restatement of stp_band_pack: what `cooler.Cooler(cool).matrix(balance=...)` + dense fetch
    (stripenn.py:80-118, getStripe.py:808) would put into the diagonal band.  value = count * (b[bin1] *
    b[bin2]) (cooler's dense branch: count block times np.outer(bias1, bias2); `weight` here is that multiplicative
    bias, 1 / w for the divisive columns; raw counts when weight is None), written at (i, j) and mirrored at
    (j, i).  A cell no stored pixel names is the count 0 times the same product: 0 where the product is finite,
    NaN along the whole row and column of a bin whose weight is NaN (cooler multiplies the DENSE count block, so an
    unbalanced bin is NaN everywhere, not only where pixels are stored -- the reference's all-NaN column deletion,
    getStripe.py:709-737, and nanmean, :518-521, see exactly that; pinned by tests/golden/e2e_chr16.npz, whose
    reference run read the same matrix through dense fetches); cells outside the chromosome are 0.
    cooler itself is absent here: parity unpinned for this reader."""
    bin1 = np.asarray(bin1, np.int64); bin2 = np.asarray(bin2, np.int64)
    v = np.asarray(count).astype(np.float64)
    if weight is not None:
        w = np.asarray(weight, np.float64)
        v = v * (w[bin1] * w[bin2])
    i = bin1 - lo; j = bin2 - lo
    ok = (i >= 0) & (j >= 0) & (i < nrows) & (j < nrows)
    i, j, v = i[ok], j[ok], v[ok]
    d = j - i
    band = np.zeros((int(nrows), 2 * hw), np.float64)
    if weight is not None:
        wl = np.asarray(weight, np.float64)[lo:lo + int(nrows)]
        for r0 in range(0, int(nrows), 4096):
            r1 = min(r0 + 4096, int(nrows))
            cc = np.arange(r0, r1)[:, None] + np.arange(-hw, hw)[None, :]
            inside = (cc >= 0) & (cc < nrows)
            with np.errstate(invalid='ignore', over='ignore'):
                z = 0.0 * (wl[r0:r1][:, None] * wl[np.clip(cc, 0, int(nrows) - 1)])
            band[r0:r1] = np.where(inside, z, 0.0)
    m = (d >= -hw) & (d < hw)
    band[i[m], d[m] + hw] = v[m]
    m = (-d >= -hw) & (-d < hw)
    band[j[m], hw - d[m]] = v[m]
    return band


def nearest_from_pixels(bin1, bin2, count, weight, lo, nrows):
    """CPU restatement of stp_band_nearest: per bin of [lo, lo + nrows) the distance to the nearest stored pixel
    with a positive balanced value in its row of the symmetric matrix, to the right (column >= row) and to the
    left (column < row); INT32_MAX where there is none.  What getStripe.nulldist's pools ask of a dense fetch
    (`np.sum(mat, axis=1) != 0`, getStripe.py:262-273, 329-331) reduces to these two numbers per row."""
    bin1 = np.asarray(bin1, np.int64); bin2 = np.asarray(bin2, np.int64)
    v = np.asarray(count).astype(np.float64)
    if weight is not None:
        w = np.asarray(weight, np.float64)
        v = v * (w[bin1] * w[bin2])
    i = bin1 - lo; j = bin2 - lo
    ok = (i >= 0) & (j >= 0) & (i < nrows) & (j < nrows) & (v > 0)
    i, j = i[ok], j[ok]
    i, j = np.minimum(i, j), np.maximum(i, j)
    big = np.iinfo(np.int32).max
    right = np.full(int(nrows), big, np.int64)
    left = np.full(int(nrows), big, np.int64)
    np.minimum.at(right, i, j - i)
    off = j > i
    np.minimum.at(left, j[off], (j - i)[off])
    return right, left


# ------------------------------------------------------------------ seeimage
def window_rgb(A, M):
    """seeimage.py:78-85: the float RGB image the reference hands to imshow (values in [0, 1], NaN where A is NaN):
    red = 1, green = blue = clip((255 * (M - A) / M) / 255, 0, 1) with negatives zeroed first."""
    A = np.asarray(A, dtype=np.float64)
    with np.errstate(invalid='ignore', divide='ignore'):
        blue = 255 * (M - A) / M
        blue[np.where(blue < 0)] = 0
        plane = blue / 255
    img = np.stack([np.ones_like(plane), plane, plane], axis=-1)
    return np.clip(img, a_min=0, a_max=1)
