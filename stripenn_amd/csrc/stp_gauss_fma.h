// stp_gauss_fma.h -- certified fused multiply-add for the Gaussian passes of the Canny kernels.
//
// An output of scipy's NI_Correlate1D symmetric branch is the FLOAT rounding of an f64 sum of R+1 products:
//   exact:  o = x0*w[R];  o = o + RN((x[-k] + x[k]) * w[R-k]),  k = R .. 1      (2R+1 roundings)
//   fused:  o = x0*w[R];  o = fma(x[-k] + x[k], w[R-k], o)                        ( R+1 roundings)
// Bound: write p_k for the exact product of step k.  The exact order adds RN(p_k) = p_k + e_k, |e_k| <= 2^-53 p_k,
// and both orders round each addition (<= ulp/2 each).  By induction the two partial sums differ by at most
// sum |e_k| + R ulp(o) <= 2^-53 o + R ulp(o) < (R+1) ulp(o) (all terms are non-negative -- grey values and weights
// are -- so the partial sums never exceed the final one).  Hence the two sums round to the same float unless the
// fused one lies within (R+1) <= 13 ulp(f64) of a float rounding boundary, i.e. unless the 29 mantissa bits the
// conversion drops are within 13 of the half-way pattern 0x10000000.  The test below uses 32 and costs 1.5 32-bit
// integer instructions per output (shift-add, half a three-way min); a run in which some output fails it (about 1e-7 of the outputs)
// is recomputed in the reference's exact order.  R+1 instead of 2R+1 FP64 roundings = 17 instead of 25 FP64
// instructions per output at R = 8, results bit for bit those of the exact order.  Checked on 24 M random windows by
// tests/test_emu_kernels.py::test_fma_certification_random_windows; tests/emu also replays the kernels with EVERY
// output flagged (the exact fallback alone).
#pragma once
#include <string.h>

#ifndef STP_FMA_NEAR     /* (tests/emu builds a second replay with -DSTP_FMA_NEAR=0x10000000: every output flagged) */
#define STP_FMA_NEAR 32  /* power of two; f64 ulps around a float rounding boundary inside which the exact order decides */
#endif

STP_HD unsigned stp_lo32(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (unsigned)__double2loint(v);
#else
    unsigned long long b;
    memcpy(&b, &v, 8);
    return (unsigned)b;
#endif
}
// The 29 mantissa bits the float conversion drops, moved to the top of a word and offset so that a value within
// STP_FMA_NEAR ulp(f64) of a float rounding boundary (dropped bits within STP_FMA_NEAR of 0x10000000) gives a word
// below 16 * STP_FMA_NEAR: ((lo + NEAR - 0x10000000) mod 2^29) << 3, ONE v_lshl_add_u32.  (A binade crossing needs
// no extra case: the pattern of a rounding boundary is the same in every binade, and the binade edge itself is a
// float.)  A run keeps the minimum of these words (v_min3_u32 takes two outputs at a time): 1.5 integer
// instructions per output.
STP_HD unsigned stp_fma_near_word(double v)
{
    return (stp_lo32(v) << 3) + (((unsigned)STP_FMA_NEAR - 0x10000000u) << 3);
}

// N consecutive outputs from a window of N + 2R f64 values (fused order).  Returns 0 when the run must be settled by
// the exact order.
template <int R, int N>
STP_HD unsigned stp_gauss_run_fma(const double* win, const double* w, float* out)
{
    unsigned far = 0xFFFFFFFFu;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < N; q++) {
        double a = win[q + R] * w[R];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = R; k >= 1; k--) a = fma(win[q + R - k] + win[q + R + k], w[R - k], a);
        out[q] = (float)a;
        const unsigned nw = stp_fma_near_word(a);
        far = nw < far ? nw : far;
    }
    if ((unsigned long long)STP_FMA_NEAR >= 0x10000000ull) return 0u;      // the all-flagged replay of tests/emu
    return far >= 16u * (unsigned)STP_FMA_NEAR ? 1u : 0u;
}

// The reference's own order for ONE output, from a strided f32 line (run-time loops, no register arrays, not
// inlined: this is the rare path and must not cost the common one any registers or code).  Element k
// (-R <= k <= R) is taken as 0 unless valid_lo <= k + R <= valid_hi (constant-mode zero padding).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __noinline__ static
#else
static
#endif
float stp_gauss_exact(const float* centre, int stride, int R, const double* w, unsigned valid_lo, unsigned valid_hi)
{
    double a = ((unsigned)R >= valid_lo && (unsigned)R <= valid_hi) ? (double)centre[0] * w[R] : 0.0 * w[R];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int k = R; k >= 1; k--) {
        const unsigned il = (unsigned)(R - k), ih = (unsigned)(R + k);
        const double xl = (il >= valid_lo && il <= valid_hi) ? (double)centre[-k * stride] : 0.0;
        const double xh = (ih >= valid_lo && ih <= valid_hi) ? (double)centre[k * stride] : 0.0;
        a += (xl + xh) * w[R - k];
    }
    return (float)a;
}
