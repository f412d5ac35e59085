"""How much could the unpinned OpenCV arithmetic move the output?  

The three OpenCV calls of the path (cv.merge, cv.filter2D, cv.cvtColor: getStripe.py:894-913) are pinned only by
their documented semantics (oracle/refharness/standins/cv2.py); a real OpenCV build may round the float32 grey
value differently by at most 1 ulp (SIMD / FMA / IPP code paths).  This script bounds the effect without OpenCV:
on the 36 golden images (6 frames x 6 brightness levels of tests/golden/stages_chr7.npz) EVERY grey pixel is moved
by +1 or -1 ulp(float32) at random, the rest of StripeSearch (Canny, verticalLine, block, line joining) is re-run
through the CPU oracle, and the edge pixels / stripe rows that differ from the unperturbed run are counted.

It also evaluates the two DETERMINISTIC orderings real OpenCV builds use: `filter2D` accumulating with a
fused multiply-add (AVX2 / FMA dispatch) and `RGB2GRAY` as nested fused multiply-adds, alone and combined.

    python tools/cv2_ambiguity.py [trials]          (CPU only; ~1 min for 10 trials)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402  (tools/ may use the checker; the product never does)
from stripenn_amd import synth          # noqa: E402


def rows_of(edges, minH=10, maxW=8):
    vert = O.vertical_line(edges)
    out = []
    for ud in (1, 2):
        _, r = O.join_dbg(edges, vert, ud, minH, maxW)
        out += [(ud,) + tuple(int(v) for v in q) for q in r]
    return out


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'stages_chr7.npz'))
    names, sizes, sel = synth.make_genome([int(g['chromsize'])], int(g['resol']), seed0=int(g['seed0']), names=['chr7'])
    ch = sel.chroms['chr7']
    gw = np.ascontiguousarray(g['gw_2p0'])
    rng = np.random.default_rng(20261003)
    tot_px = tot_edge = tot_rows = 0
    d_edge = d_rows = d_imgs_edge = d_imgs_rows = 0
    nimg = 0
    for ci in range(int(g['ncases'])):
        p = 'c%d_' % ci
        D, nz = O.frame_dense(ch.block, int(g[p + 'start']), int(g[p + 'end']))
        D = np.ascontiguousarray(D[np.ix_(nz, nz)])
        gp = O.gplane(D, float(g[p + 'M']))
        for bi, b in enumerate(g['bvals']):
            grey = O.gray(gp, b)
            e0 = O.canny(grey, gw, 8)
            assert np.array_equal(np.packbits(e0.astype(bool), axis=1), g[p + 'edges'][bi])     # the golden image itself
            r0 = rows_of(e0)
            nimg += 1
            for _ in range(trials):
                up = rng.random(grey.shape) < 0.5
                pert = np.where(up, np.nextafter(grey, np.float32(2.0)), np.nextafter(grey, np.float32(-1.0))).astype(np.float32)
                e1 = O.canny(pert, gw, 8)
                r1 = rows_of(e1)
                de = int(np.count_nonzero(e0 != e1))
                dr = len(set(r0) ^ set(r1))
                tot_px += grey.size; tot_edge += int(e0.sum()); tot_rows += len(r0)
                d_edge += de; d_rows += dr
                d_imgs_edge += de > 0; d_imgs_rows += dr > 0
    # ---- the DETERMINISTIC alternatives of a real OpenCV build (they move many pixels in the same direction at once):
    #      filter2D accumulating with a fused multiply-add, RGB2GRAY evaluated as nested fused multiply-adds
    variants = (('filter2D with fma accumulation', True, 0), ('RGB2GRAY fma(B,cb,fma(G,cg,R*cr))', False, 1),
                ('RGB2GRAY fma(R,cr,fma(G,cg,B*cb))', False, 2), ('both: fma box sum + fma(B,..) grey', True, 1),
                ('both: fma box sum + fma(R,..) grey', True, 2))
    print('deterministic OpenCV orderings, the same 36 golden images (grey pixels that change, by how many ulp(f32) at most; '
          'edge pixels that change; raw StripeSearch rows that change):')
    for name, box, order in variants:
        gpx = gmax = dpx = drw = npx = nedge = nrows = 0
        for ci in range(int(g['ncases'])):
            p = 'c%d_' % ci
            D, nz = O.frame_dense(ch.block, int(g[p + 'start']), int(g[p + 'end']))
            D = np.ascontiguousarray(D[np.ix_(nz, nz)])
            gp = O.gplane(D, float(g[p + 'M']))
            for bi, b in enumerate(g['bvals']):
                grey = O.gray(gp, b)
                alt = O.gray_alt(gp, b, 3, box, order)
                diff = grey.view(np.int32).astype(np.int64) - alt.view(np.int32).astype(np.int64)
                gpx += int(np.count_nonzero(diff)); gmax = max(gmax, int(np.abs(diff).max())); npx += grey.size
                e0, e1 = O.canny(grey, gw, 8), O.canny(alt, gw, 8)
                r0, r1 = rows_of(e0), rows_of(e1)
                dpx += int(np.count_nonzero(e0 != e1)); nedge += int(e0.sum())
                drw += len(set(r0) ^ set(r1)); nrows += len(r0)
        print('  %-40s grey %d of %d px (max %d ulp); edges %d of %d; rows %d of %d' % (name, gpx, npx, gmax, dpx, nedge, drw, nrows))
    n = nimg * trials
    print('%d golden images x %d random +-1 ulp(f32) perturbations of EVERY grey pixel' % (nimg, trials))
    print('edge pixels changed : %d of %d edge pixels examined (%.3g per image; %d of %d perturbed images differ at all)'
          % (d_edge, tot_edge, d_edge / n, d_imgs_edge, n))
    print('stripe rows changed : %d of %d raw StripeSearch rows examined (%.3g per image; %d of %d perturbed images differ)'
          % (d_rows, tot_rows, d_rows / n, d_imgs_rows, n))
    print('pixels perturbed    : %d' % tot_px)


if __name__ == '__main__':
    main()
