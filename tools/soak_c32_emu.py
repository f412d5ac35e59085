"""CPU soak of k_canny_f32's phase code (tests/emu replay) against the oracle's class maps: random synthetic
chromosomes, frames, maxpixel quantiles, brightness levels, both default sigmas, plus noisy test images of random
size.  usage: soak_c32_emu.py SEED0 NFRAMES"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from oracle import oracle as O
from stripenn_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
E = C.CDLL(os.path.join(HERE, '..', 'tests', 'emu', 'libstp_emu.so'))


def _p(a): return a.ctypes.data_as(C.c_void_p)


def check(gray, sigma):
    S = gray.shape[0]
    gw, gr = O.gauss_weights(sigma)
    full = np.zeros((400, 400), np.float32); full[:S, :S] = gray
    low = np.zeros(2800, np.uint64); high = np.zeros(2800, np.uint64); cnt = np.zeros(2, np.int64)
    E.emu_canny_f32(_p(full), S, gr, _p(gw), _p(low), _p(high), _p(cnt))
    _, d = O.canny(np.ascontiguousarray(gray), gw, gr, debug=True)
    un = lambda w: np.unpackbits(np.ascontiguousarray(w.reshape(7, 400).T).view(np.uint8).reshape(400, 56), axis=1, bitorder='little')[:S, :S]   # class planes: word-column-major (STP_CLS)
    got = un(low).astype(np.uint8) + un(high)
    return int((got != d['cls']).sum()), cnt


def main():
    seed0, nfr = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(seed0)
    bad = 0; tot = np.zeros(2, np.int64); nimg = 0; t0 = time.time()
    for k in range(nfr):
        nb = int(rng.integers(500, 1500))
        ch = synth.SynthChrom(nb, seed0 * 1000 + k, stripe_every=int(rng.integers(20, 200)), stripe_gain=float(rng.uniform(1.5, 4)),
                              nan_frac=float(rng.choice([0.0, 0.005, 0.05])))
        f0 = int(rng.integers(0, nb - 400))
        D, nz = O.frame_dense(ch.block, f0, f0 + 399)
        if len(nz) < 12:
            continue
        D = np.ascontiguousarray(D[np.ix_(nz, nz)])
        pos = D[D > 0]
        if pos.size == 0:
            continue
        M = float(np.quantile(pos, float(rng.uniform(0.8, 0.999))))
        g = O.gplane(D, M)
        for b in O.brightness_levels():
            sigma = float(rng.choice([2.0, 2.5]))
            nbad, cnt = check(O.gray(g, b, 3), sigma)
            bad += nbad; tot += cnt; nimg += 1
            if nbad:
                print('MISMATCH seed', seed0, 'k', k, 'b', b, 'sigma', sigma, nbad, flush=True)
        if k % 20 == 0:
            print(k, 'images', nimg, 'candidates', int(tot[0]), 'resolved', int(tot[1]), 'mismatches', bad, '%.0fs' % (time.time() - t0), flush=True)
    print('DONE seed', seed0, 'images', nimg, 'candidates', int(tot[0]), 'resolved', int(tot[1]), 'mismatching pixels', bad)


if __name__ == '__main__':
    main()
