"""One-off randomised soak: the GPU-vs-oracle comparison of tests/test_gpu_fuzz.py over many more seeds.
    python tools/soak_fuzz.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import oracle as O
from stripenn_amd import hip, synth
import test_gpu_fuzz as F

O.build()
ctx = hip.Context(0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time(); nrec = 0; bad = []
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(320, 1300))
    ch = synth.SynthChrom(n, seed, stripe_every=int(rng.integers(8, 250)), stripe_gain=float(rng.uniform(1.2, 6.0)),
                          nan_frac=float(rng.choice([0.0, 0.005, 0.02, 0.1, 0.3])), balanced=bool(rng.integers(0, 2)))
    A = ch.block(0, n, 0, n)
    pos = A[A > 0]
    Ms = np.quantile(pos, np.sort(rng.uniform(0.3, 0.9999, 3)))
    if rng.random() < 0.2:
        Ms[0] = Ms[0] * float(rng.choice([1e-3, 1e3]))            # nearly all-black / all-white level
    frames = []
    for _ in range(5):
        s = int(rng.integers(0, n - 12)); e = min(n - 1, s + int(rng.integers(11, 400)))
        frames.append((s, e))
    kw = dict(sigma=float(rng.choice([2.0, 2.0, 2.5, 1.5, 3.0])), bfilter=int(rng.choice([3, 3, 3, 5, 1])),
              minH=int(rng.choice([10, 10, 5, 20])), maxW=int(rng.choice([8, 8, 4, 16])))
    try:
        nrec += F._check(ctx, A, frames, Ms, **kw)
    except AssertionError as ex:
        bad.append((seed, kw, str(ex)[:200]))
        print('MISMATCH seed', seed, kw, flush=True)
print('%d configurations, %d records compared, %d mismatches, %.0f s' % (count, nrec, len(bad), time.time() - t0))
for b in bad:
    print(b)
ctx.close()
