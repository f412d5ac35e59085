"""GPU soak of k_canny_f32 against k_canny_pipe (STP_CANNY=exact): class maps and edge maps of random synthetic
frames, image by image.  usage: soak_c32_gpu.py SEED0 NCHROM"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from stripenn_amd import synth, hip


def main():
    seed0, nchrom = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(seed0)
    ctx = hip.Context(0)
    nimg = bad = 0
    t0 = time.time()
    for k in range(nchrom):
        nb = int(rng.integers(900, 2400))
        ch = synth.SynthChrom(nb, seed0 * 1000 + k, stripe_every=int(rng.integers(15, 200)), stripe_gain=float(rng.uniform(1.5, 5)),
                              nan_frac=float(rng.choice([0.0, 0.005, 0.05])), balanced=bool(rng.integers(0, 2)))
        band_h = ch.band(512)
        band = ctx.band_upload(band_h)
        nf = 4
        st = np.sort(rng.integers(0, nb - 400, nf)); en = st + rng.integers(150, 400, nf)
        fr = band.frames(st, np.minimum(en, nb - 1))
        pos = band_h[band_h > 0]
        if pos.size == 0:
            continue
        for f in range(nf):
            if fr.S[f] < 20:
                continue
            M = float(np.quantile(pos, float(rng.uniform(0.7, 0.9995))))
            sigma = float(rng.choice([2.0, 2.0, 2.5, 1.0, 1.5, 3.0]))
            bf = int(rng.choice([3, 3, 3, 1, 5]))
            for bi in range(6):
                os.environ.pop('STP_CANNY', None)
                a = fr.dbg_stages(f, M, bi, sigma=sigma, bfilter=bf)
                os.environ['STP_CANNY'] = 'exact'
                b = fr.dbg_stages(f, M, bi, sigma=sigma, bfilter=bf)
                os.environ.pop('STP_CANNY', None)
                nimg += 1
                if not (np.array_equal(a['cls'], b['cls']) and np.array_equal(a['edges'], b['edges'])):
                    bad += 1
                    print('MISMATCH seed', seed0, 'chrom', k, 'frame', f, 'M', M, 'sigma', sigma, 'bfilter', bf, 'bi', bi,
                          int((a['cls'] != b['cls']).sum()), flush=True)
        fr.close(); band.close()
        if k % 25 == 0:
            print(k, 'images', nimg, 'mismatching images', bad, '%.0fs' % (time.time() - t0), flush=True)
    print('DONE seed', seed0, ':', nimg, 'images, class and edge maps of k_canny_f32 == k_canny_pipe;', bad, 'mismatching images; %.0f s' % (time.time() - t0))


if __name__ == '__main__':
    main()
