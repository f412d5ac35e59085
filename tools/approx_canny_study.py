"""Feasibility study (CPU, numpy): how many pixels of a Canny class map can NOT be decided from an all-f32
evaluation of the Gaussian / Sobel / magnitude chain with rigorous error margins?

The exact chain is the oracle's (oracle.canny(debug=True): smoothed, isobel, jsobel, magnitude, class).  The
approximate chain runs the two Gaussian passes, the bleed-over scaling, the Sobel sums and the magnitude in f32
(multiply and add rounded separately -- a little worse than the device's fma).  A pixel is UNCERTAIN when a decision
of skimage's non-maximum suppression (_canny.py:193-280) could flip inside the error bounds:
  E_s  bound on |S_approx - S_exact|, relative to the local maximum of S (3x3)
  E_g  = 8 E_s + 4 ulp: bound on the Sobel components;  E_m = sqrt(2) E_g: bound on the magnitude.
Prints the measured maximal errors (to compare with the bounds) and the uncertain fractions.
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from oracle import oracle as O
from stripenn_amd import synth

F = np.float32


def conv_f32(img, w32, axis):
    """'constant' mode (zeros) symmetric correlation in f32, centre first then pairs from the outside in."""
    R = (len(w32) - 1) // 2
    S = img.shape[0]
    pad = np.zeros((S + 2 * R, S + 2 * R), F)
    pad[R:R + S, R:R + S] = img
    def sh(k):
        return pad[R + k:R + k + S, R:R + S] if axis == 0 else pad[R:R + S, R + k:R + k + S]
    a = sh(0) * w32[R]
    for k in range(R, 0, -1):
        a = (a + (sh(-k) + sh(k)) * w32[R - k]).astype(F)
    return a


def study(gray, gw, gr, ES_REL):
    S = gray.shape[0]
    edges, d = O.canny(gray, gw, gr, debug=True)
    w32 = gw.astype(F)
    Va = conv_f32(gray, w32, 0)
    Ha = conv_f32(Va, w32, 1)
    ones = np.ones((S, S), F)
    bl = conv_f32(conv_f32(ones, w32, 0), w32, 1)          # f32 bleed-over (the device would tabulate the exact one)
    Sa = (Ha / (bl + F(np.finfo(np.float64).eps))).astype(F)
    err_s = np.abs(Sa.astype(np.float64) - d['smoothed'])
    # local max of |S| (3x3) for the relative bounds
    P = np.pad(np.abs(d['smoothed']), 1, mode='edge')
    loc = np.max([P[1 + a:1 + a + S, 1 + b:1 + b + S] for a in (-1, 0, 1) for b in (-1, 0, 1)], axis=0)
    # f32 sobel (reflect == edge replicate for +-1)
    Pa = np.pad(Sa, 1, mode='edge')
    def at(a, b): return Pa[1 + a:1 + a + S, 1 + b:1 + b + S]
    ja = ((at(0, 1) - at(0, -1)) * F(2) + ((at(-1, 1) - at(-1, -1)) + (at(1, 1) - at(1, -1)))).astype(F)
    ia = ((at(1, 0) - at(-1, 0)) * F(2) + ((at(1, -1) - at(-1, -1)) + (at(1, 1) - at(-1, 1)))).astype(F)
    ma = np.sqrt(ia * ia + ja * ja).astype(F)
    err_g = np.maximum(np.abs(ia - d['isobel']), np.abs(ja - d['jsobel']))
    err_m = np.abs(ma - d['mag'])
    Es = ES_REL * loc
    Eg = 8.0 * Es + 4 * 6e-8 * loc * 4
    Em = 1.5 * Eg
    m = ma.astype(np.float64); gi = ia.astype(np.float64); gj = ja.astype(np.float64)
    interior = np.zeros((S, S), bool); interior[1:-1, 1:-1] = True
    cand = interior & (m >= 0.1 - Em)
    unc_thr = cand & ((np.abs(m - 0.1) <= Em) | (np.abs(m - 0.2) <= Em))
    ai, aj = np.abs(gi), np.abs(gj)
    unc_sec = cand & ((ai <= Eg) | (aj <= Eg) | (np.abs(ai - aj) <= 2 * Eg))
    # interpolation test with the approximate values (sector from the approximate signs)
    Pm = np.pad(m, 1, mode='constant'); PE = np.pad(Em, 1, mode='edge')
    def mat(dy, dx): return Pm[1 + dy:1 + dy + S, 1 + dx:1 + dx + S]
    def eat(dy, dx): return PE[1 + dy:1 + dy + S, 1 + dx:1 + dx + S]
    same = ((gi >= 0) & (gj >= 0)) | ((gi <= 0) & (gj <= 0))
    opp = ~same
    lp = np.zeros((S, S)); lm = np.zeros((S, S)); tol = np.zeros((S, S))
    with np.errstate(divide='ignore', invalid='ignore'):
        for mask, num, den, o1, o2 in (
            (same & (ai >= aj), aj, ai, (1, 0), (1, 1)),
            (same & (ai < aj), ai, aj, (0, 1), (1, 1)),
            (opp & (ai < aj), ai, aj, (0, 1), (-1, 1)),
            (opp & (ai >= aj), aj, ai, (-1, 0), (-1, 1)),
        ):
            w = np.where(den > 0, num / den, 0.0)
            # d w <= (E_num + w E_den) / den <= 2 Eg / den
            dw = np.where(den > 0, 2 * Eg / np.maximum(den, 1e-30), 1.0)
            c1p, c2p = mat(*o1), mat(*o2); c1m, c2m = mat(-o1[0], -o1[1]), mat(-o2[0], -o2[1])
            lp_ = c2p * w + c1p * (1 - w); lm_ = c2m * w + c1m * (1 - w)
            t_ = np.maximum(eat(*o1), eat(*o2)) + Em + dw * np.maximum(np.abs(c2p - c1p), np.abs(c2m - c1m))
            lp = np.where(mask, lp_, lp); lm = np.where(mask, lm_, lm); tol = np.where(mask, t_, tol)
    unc_nms = cand & ((np.abs(lp - m) <= tol) | (np.abs(lm - m) <= tol))
    unc = unc_thr | unc_sec | unc_nms
    return dict(S=S, cand=int(cand.sum()), unc=int(unc.sum()), thr=int(unc_thr.sum()), sec=int(unc_sec.sum()),
                nms=int(unc_nms.sum()), err_s=float((err_s / np.maximum(loc, 1e-30)).max()), err_g=float((err_g / np.maximum(loc, 1e-30)).max()),
                err_m=float((err_m / np.maximum(loc, 1e-30)).max()), smax=float(loc.max()), edges=int(edges.sum()),
                unc_mask=unc)


def main():
    es_rel = float(sys.argv[1]) if len(sys.argv) > 1 else 1.3e-6
    ch = synth.SynthChrom(3000, 16)
    gw, gr = O.gauss_weights(2.0)
    blk = ch.block(0, 3000, 0, 3000)
    Ms = np.quantile(blk[blk > 0], [0.95, 0.99])
    tot = dict(px=0, cand=0, unc=0, thr=0, sec=0, nms=0)
    worst = dict(err_s=0, err_g=0, err_m=0)
    for f0 in (100, 700, 1500, 2300):
        D, nz = O.frame_dense(ch.block, f0, f0 + 399)
        D = np.ascontiguousarray(D[np.ix_(nz, nz)])
        for M in Ms:
            g = O.gplane(D, float(M))
            for b in O.brightness_levels():
                r = study(O.gray(g, b, 3), gw, gr, es_rel)
                tot['px'] += r['S'] ** 2
                for k in ('cand', 'unc', 'thr', 'sec', 'nms'):
                    tot[k] += r[k]
                for k in worst:
                    worst[k] = max(worst[k], r[k])
                # clustering: 8x8 patches holding uncertain pixels
                um = r['unc_mask']; S = r['S']
                pat = sum(um[y:y + 8, x:x + 8].any() for y in range(0, S, 8) for x in range(0, S, 8))
                tot['patches'] = tot.get('patches', 0) + int(pat)
                print(f0, round(float(M), 2), round(float(b), 1), {k: r[k] for k in ('cand', 'unc', 'thr', 'sec', 'nms', 'edges')},
                      'patches', int(pat), 'smax %.3f' % r['smax'], flush=True)
    print('TOTAL', tot, 'uncertain per 160k px: %.1f' % (tot['unc'] / tot['px'] * 160000))
    print('measured max relative errors (vs local max |S|):', worst, 'assumed E_s rel', es_rel)


if __name__ == '__main__':
    main()
