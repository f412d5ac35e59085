"""Stand-in for the three OpenCV calls the reference makes (cv2 is not installable offline).

TEST INFRASTRUCTURE ONLY: used by oracle/refharness/gen_golden.py so that the unmodified
reference (/root/reference/src/stripenn) can be imported under /opt/conda/bin/python3.9.

The reference pins no OpenCV version (pyproject.toml:31-41, poetry.lock is empty), so the
semantics below ARE the specification for this build ("parity unpinned"
for these three calls; real OpenCV may differ by <= 1 ulp(float32) in the grey image):

* merge(planes)                 -> HxWxC float64 stack               (getStripe.py:894, ImageProcessing.py:39)
* filter2D(src, -1, kernel)     -> correlation, anchor at centre, BORDER_REFLECT_101, float64,
                                   accumulated from 0 over kernel taps in row-major order,
                                   one multiply and one add per tap (no FMA)    (getStripe.py:909)
* cvtColor(f32 RGB, RGB2GRAY)   -> float32  (R*0.299f + G*0.587f) + B*0.114f     (getStripe.py:913)
"""
import numpy as np

COLOR_RGB2GRAY = 7


def merge(planes):
    return np.dstack([np.asarray(p, dtype=np.float64) for p in planes])


def filter2D(src, ddepth, kernel):
    src = np.asarray(src)
    kernel = np.asarray(kernel, dtype=np.float64)
    kh, kw = kernel.shape
    ay, ax = kh // 2, kw // 2
    pads = ((ay, ay), (ax, ax)) + ((0, 0),) * (src.ndim - 2)
    pad = np.pad(src, pads, mode='reflect')  # numpy 'reflect' == BORDER_REFLECT_101
    H, W = src.shape[:2]
    out = np.zeros(src.shape, dtype=np.float64)
    for ky in range(kh):
        for kx in range(kw):
            out = out + kernel[ky, kx] * pad[ky:ky + H, kx:kx + W]
    return out


def cvtColor(src, code):
    if code != COLOR_RGB2GRAY:
        raise NotImplementedError(code)
    src = np.asarray(src)
    if src.dtype != np.float32:
        raise TypeError('stand-in cvtColor expects float32')
    r, g, b = src[..., 0], src[..., 1], src[..., 2]
    return (r * np.float32(0.299) + g * np.float32(0.587)) + b * np.float32(0.114)
