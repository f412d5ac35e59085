"""cProfile of the full `compute` driver fed from a pixel table (host-side hot spots)."""
import sys, os, io, contextlib, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stripenn_amd import stripenn, pixels, synth
names = ['chr%d' % (i + 1) for i in range(6)]
chroms = {n: synth.SynthChrom(6000 - 400 * i, 7 + i) for i, n in enumerate(names)}
os.makedirs('gpurun_out', exist_ok=True)
pixels.PixelTable.from_synth(names, chroms, 5000).save('gpurun_out/e2e_pixels.npz')
def run():
    with contextlib.redirect_stdout(io.StringIO()):
        stripenn.compute('pixels:gpurun_out/e2e_pixels.npz', 'gpurun_out/e2e_px_out', 'weight', 'all', 2.0, 10, 8,
                         '0.95,0.96,0.97,0.98,0.99', 8, 0.1, '0', False, 3, 123456789, force=True)
run()                                  # warm-up (library load, workspace)
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(28)
