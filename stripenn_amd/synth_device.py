"""Synthetic chromosomes generated directly in device memory (benchmark / probe data only).

`stripenn_amd.synth.SynthChrom` defines every pixel as a pure function of (seed, min(i, j), max(i, j)):
a 64-bit integer hash plus IEEE + - * / sqrt.  This module evaluates the same function with torch
tensor ops on the GPU (int64 arithmetic wraps modulo 2^64 exactly like numpy's uint64), so a
whole-genome set of diagonal bands (4.3 GB for mm10 at 5 kb) appears in HBM in about a second instead of
ten minutes of host numpy, and cooler-style pixel tables of the same genome can be pulled back for the
end-to-end driver.  torch is used for device memory only; nothing here is on the product path.
"""
import numpy as np

from . import synth

_MASK64 = (1 << 64) - 1


def _s64(v):
    """Python int -> the signed 64-bit value with the same bit pattern."""
    v &= _MASK64
    return v - (1 << 64) if v >= (1 << 63) else v


_C1, _C2, _C3 = _s64(0x9E3779B97F4A7C15), _s64(0xBF58476D1CE4E5B9), _s64(0x94D049BB133111EB)
_K_LO = _s64(0x9E3779B97F4A7C15)
_K_SEED = 0xD1B54A32D192ED03


def _lsr(x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def _splitmix64(x):
    x = x + _C1
    z = x
    z = (z ^ _lsr(z, 30)) * _C2
    z = (z ^ _lsr(z, 27)) * _C3
    return z ^ _lsr(z, 31)


def _field(h, shift):
    import torch
    return ((h >> shift) & 0xFFFF).to(torch.float64)


class DeviceChrom:
    """Device-side twin of synth.SynthChrom (same seed -> same matrix)."""

    def __init__(self, nbins, seed, device, **kw):
        import torch
        self.host = synth.SynthChrom(nbins, seed, **kw)
        self.nbins, self.seed, self.device = int(nbins), int(seed), device
        h = self.host
        self.w = torch.from_numpy(np.ascontiguousarray(h.w, dtype=np.float64)).to(device)
        self.nanflag = torch.from_numpy(np.ascontiguousarray(h.nanflag)).to(device)
        self.s_lo = torch.from_numpy(h._s_lo).to(device)
        self.s_hi = torch.from_numpy(h._s_hi).to(device)

    def _counts(self, r, c):
        """counts of the pixels (r, c) (int64 tensors of equal / broadcastable shape, all indices in range)."""
        import torch
        lo = torch.minimum(r, c)
        hi = torch.maximum(r, c)
        d = hi - lo
        h = _splitmix64((lo * _K_LO) ^ _splitmix64(hi + _s64(self.seed * _K_SEED)))
        usum = _field(h, 0) + _field(h, 16) + _field(h, 32) + _field(h, 48)
        z = (usum - 131070.0) / 37837.22
        lam = 240.0 / (1.0 + d.to(torch.float64)) + 1.0
        in1 = (r >= self.s_lo[c]) & (r <= self.s_hi[c])
        in2 = (c >= self.s_lo[r]) & (c <= self.s_hi[r])
        lam = torch.where(in1 | in2, lam * self.host.stripe_gain, lam)
        if self.host.depth != 1.0:
            lam = lam * self.host.depth
        cnt = torch.floor(lam + torch.sqrt(lam) * z + 0.5)
        cnt = torch.where(cnt < 0.0, torch.zeros_like(cnt), cnt)
        if self.host.count_div > 1:
            cnt = torch.floor(cnt / float(self.host.count_div))
        return torch.where(d > synth.BAND_LIMIT, torch.zeros_like(cnt), cnt)

    def band(self, halfwidth=512, chunk=8192):
        """(nbins, 2*hw) float64 device tensor in the library's band layout (include/stripenn_hip.h)."""
        import torch
        hw = int(halfwidth)
        out = torch.empty((self.nbins, 2 * hw), dtype=torch.float64, device=self.device)
        dd = torch.arange(-hw, hw, device=self.device, dtype=torch.int64)[None, :]
        for a in range(0, self.nbins, chunk):
            b = min(a + chunk, self.nbins)
            r = torch.arange(a, b, device=self.device, dtype=torch.int64)[:, None].expand(b - a, 2 * hw)
            c = r + dd
            ok = (c >= 0) & (c < self.nbins)
            cc = c.clamp(0, self.nbins - 1)
            cnt = self._counts(r, cc)
            val = (cnt * self.w[r]) * self.w[cc]
            val = torch.where(self.nanflag[r] | self.nanflag[cc], torch.full_like(val, float('nan')), val)
            out[a:b] = torch.where(ok, val, torch.zeros_like(val))
        return out

    def pixels(self, bin_offset=0, chunk=8192, narrow=False):
        """Upper-triangle stored pixels (bin1 <= bin2, count > 0) as host arrays (global ids = local + bin_offset),
        sorted by (bin1, bin2) like cooler's pixel table.  narrow: bin2_id as int32 (as a reader that narrows the column
        while it inflates the file's chunks hands it over)."""
        import torch
        lim = synth.BAND_LIMIT
        b1, b2, cn = [], [], []
        dd = torch.arange(0, lim + 1, device=self.device, dtype=torch.int64)[None, :]
        for a in range(0, self.nbins, chunk):
            b = min(a + chunk, self.nbins)
            r = torch.arange(a, b, device=self.device, dtype=torch.int64)[:, None].expand(b - a, lim + 1)
            c = r + dd
            ok = c < self.nbins
            cnt = self._counts(r, c.clamp(0, self.nbins - 1))
            keep = ok & (cnt > 0)
            b1.append((r[keep] + bin_offset).cpu().numpy())
            b2.append((c[keep] + bin_offset).to(torch.int32).cpu().numpy() if narrow else (c[keep] + bin_offset).cpu().numpy())
            cn.append(cnt[keep].to(torch.int32).cpu().numpy())
        return np.concatenate(b1), np.concatenate(b2), np.concatenate(cn)


def pixel_table(names, chroms, resol, narrow=True):
    """stripenn_amd.pixels.PixelTable of device-generated chromosomes (dict name -> DeviceChrom); narrow: bin2_id as int32."""
    from . import pixels
    b1, b2, cn, ws, off, sizes = [], [], [], [], [0], []
    for nm in names:
        ch = chroms[nm]
        p = ch.pixels(bin_offset=off[-1], narrow=narrow)
        b1.append(p[0]); b2.append(p[1]); cn.append(p[2])
        w = ch.host.w.astype(np.float64).copy()
        w[ch.host.nan_bins] = np.nan
        ws.append(w)
        off.append(off[-1] + ch.nbins)
        sizes.append(ch.nbins * int(resol))
    return pixels.PixelTable(names, sizes, resol, off, np.concatenate(b1), np.concatenate(b2), np.concatenate(cn),
                             {'weight': np.concatenate(ws)})
