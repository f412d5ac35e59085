"""Where does stp_band_pack_csr spend its time?  One chr1-size chromosome (39 095 bins, ~20 M stored pixels) packed five times:
wall time of the call against the library's kernel timers.      python tools/probe_pack.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stripenn_amd import backend as BK, pixels, synth_device

torch.cuda.init()
dev = torch.device('cuda', 0)
ch = synth_device.DeviceChrom(39095, 1, dev)
t = synth_device.pixel_table(['chr1'], {'chr1': ch}, 5000)
hb = BK.HipBackend(0)
px = pixels.PixelSelector(t, 'weight').chrom_pixels('chr1')
n = len(px['count'])
print('%d pixels, bin2 %s, count %s: %.2f GB narrow, %.2f GB as int64 columns' % (n, px['bin2'].dtype, px['count'].dtype, n * 8 / 1e9, n * 20 / 1e9))
for rep in range(5):
    hb.ctx.set_profiling(True); hb.ctx.reset_stats()
    sel = hb.select_open()
    t0 = time.perf_counter()
    band = hb.pack_chrom(px, 512, sel)
    dt = time.perf_counter() - t0
    st = hb.ctx.stats()
    print('pack %.1f ms wall (%.1f GB/s of narrow columns); kernels: %s' % (dt * 1e3, n * 8 / dt / 1e9, {k: round(v['ms'], 2) for k, v in st.items()}))
    hb.select_close(sel); band.close()
px2 = dict(px); px2.pop('off')
px2['bin2'] = px['bin2'].astype(np.int64)
for rep in range(2):
    t0 = time.perf_counter(); band = hb.pack_chrom(px2, 512); dt = time.perf_counter() - t0
    print('column form (int64 ids): %.1f ms wall' % (dt * 1e3)); band.close()
hb.close()
