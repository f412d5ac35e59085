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
