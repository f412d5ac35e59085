// stp_phases.h -- per-phase bodies of the gfx950 kernels of libstripenn_hip.so.
//
// Every kernel in stripenn_hip.hip is a sequence of phases separated by workgroup barriers;
// a phase is a strided loop `for (i = tid; i < N; i += nt)` over LDS / global arrays.  The phase
// bodies live here as plain inline functions so that tests/emu (a g++ build used only by the
// CPU test-suite) can replay one workgroup sequentially (tid = 0, nt = 1) and compare it with
// the oracle before anything is launched on a GPU.  The product never runs this code on the
// CPU: the only shipped entry points are the __global__ kernels.
//
// Arithmetic contract: one IEEE rounding per source-level operation (-ffp-contract=off), the
// same operation order as the reference / pinned third-party routine cited at each function.
#pragma once
#include <stdint.h>
#include <math.h>
#include <float.h>
#include <string.h>

#if defined(__HIPCC__)
#define STP_HD __host__ __device__ __forceinline__
#else
#define STP_HD static inline
#endif

typedef unsigned long long stp_u64;

#define STP_PITCH 400 /* pixel pitch of per-image buffers */
#define STP_NW 7      /* u64 words per bit-row (448 >= 400) */
#define STP_RCAP 128  /* record slots per image in the sweep */
#define STP_RCAP_MAX 400  /* slots of the re-run: neighbouring X values pair at most once per direction -> < 400 records */
/* Class bit-planes of an image in global memory (Canny kernels -> k_lines): word w of row y.  Word-column-major since round 5:
   a Canny tile owns ONE word column over 32 rows, so its 32 words are 256 contiguous bytes (row-major they lay 56 bytes
   apart: one sector write per word, WRITE_SIZE 3.2x the payload); k_lines' loader reads a column with consecutive lanes. */
#define STP_CLS(y, w) ((w) * STP_FRAME_MAX + (y))

// ---------------------------------------------------------------------------------------------
// tile geometry
#define GT_Y 32
#define GT_X 64
#define GT_AMAX 3 /* bfilter <= 7 */
#define CT_Y 32
#define CT_X 64
#define CT_RMAX 12
#define CT_SP (CT_X + 7) /* pitch (doubles) of the smoothed tile: odd -> few LDS write conflicts; >= the 70 columns the
                            horizontal pass's whole runs write (the last run ends in the padding) */
#define CT_VP (CT_Y + 5) /* pitch (floats) of the transposed vertical-pass tile */

struct stp_tile {
    int S;        // compacted frame size
    int ty0, tx0; // tile origin (image coordinates)
};

STP_HD int stp_refl101(int i, int n)
{   // cv BORDER_REFLECT_101, valid for overshoot < n
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    if (i < 0) i = 0;
    return i;
}

STP_HD int stp_refl(int i, int n)
{   // scipy.ndimage mode='reflect' (d c b a | a b c d | d c b a), overshoot 1
    if (i < 0) return -i - 1;
    if (i >= n) return 2 * n - 1 - i;
    return i;
}

// getStripe.py:889-895  blue = 255*(M-D)/M; <0 -> 0; /255; clip[0,1]
STP_HD double stp_gplane_px(double D, double M)
{
    double t = 255.0 * (M - D);
    t = t / M;
    if (t < 0.0) t = 0.0;
    double v = t / 255.0;
    if (v < 0.0) v = 0.0;
    if (v > 1.0) v = 1.0;
    return v;
}

// ImageProcessing.py:15-23 with In=(0,b) Out=(0,1)
STP_HD double stp_bright_px(double v, double b, double k)
{
    if (v <= 0.0) return 0.0;
    if (v > b) return 1.0;
    if (v > 0.0 && v <= b) return k * (v - 0.0) + 0.0;
    return 0.0;
}

// kv * stp_bright_px(v, b, k) with the same bits for every input (NaN included) in two compares:
// kv * 1.0 == kv and kv * 0.0 == 0.0 exactly, and the three cases of stp_bright_px are disjoint.
STP_HD double stp_bright_kv(double v, double b, double k, double kv)
{
    return (v > b) ? kv : ((v > 0.0) ? kv * (k * v) : 0.0);
}

// ---------------------------------------------------------------------------------------------
// kernel A (k_gray): D -> g plane (LDS) -> per brightness: adj (LDS) -> mean blur -> grey f32
// sg / sadj: (GT_Y + 2a) x (GT_X + 2a) doubles, origin (ty0 - a, tx0 - a)
STP_HD void gray_p0(int tid, int nt, const double* __restrict__ band, int W, int hw, int64_t st,
                    const int16_t* __restrict__ nz, stp_tile T, int a, double M, double* sg)
{
    const int hh = GT_Y + 2 * a, ww = GT_X + 2 * a;
    for (int i = tid; i < hh * ww; i += nt) {
        int yy = i / ww, xx = i - yy * ww;
        int y = T.ty0 + yy - a, x = T.tx0 + xx - a;
        double g = 0.0;
        if (y < T.S + a && x < T.S + a) {
            int ry = stp_refl101(y, T.S), rx = stp_refl101(x, T.S);
            int oy = nz[ry], ox = nz[rx];
            double v = band[(st + oy) * (int64_t)W + (ox - oy + hw)];
            if (v != v) v = 0.0;                       // nantozero, getStripe.py:809
            g = stp_gplane_px(v, M);
        }
        sg[i] = g;
    }
}

STP_HD void gray_p1(int tid, int nt, int a, double b, const double* sg, double* sadj)
{
    const int hh = GT_Y + 2 * a, ww = GT_X + 2 * a;
    const double k = (1.0 - 0.0) / (b - 0.0);          // ImageProcessing.py:30
    for (int i = tid; i < hh * ww; i += nt) sadj[i] = stp_bright_px(sg[i], b, k);
}

STP_HD void gray_p2(int tid, int nt, stp_tile T, int a, const double* sadj, float* __restrict__ gray_img)
{
    const int ww = GT_X + 2 * a, bf = 2 * a + 1;
    const double kv = 1.0 / (double)(bf * bf);         // np.ones((bf,bf))/(bf*bf), getStripe.py:908
    double rb = 0.0;                                    // red plane: constant 1 through the same filter
    for (int t = 0; t < bf * bf; t++) rb = rb + kv * 1.0;
    if (rb < 0.0) rb = 0.0;
    if (rb > 1.0) rb = 1.0;
    const float r32 = (float)rb;
    for (int i = tid; i < GT_Y * GT_X; i += nt) {
        int yy = i / GT_X, xx = i - yy * GT_X;
        int y = T.ty0 + yy, x = T.tx0 + xx;
        if (y >= T.S || x >= T.S) continue;
        double acc = 0.0;
        for (int ky = 0; ky < bf; ky++)
            for (int kx = 0; kx < bf; kx++) acc = acc + kv * sadj[(yy + ky) * ww + (xx + kx)];
        if (acc < 0.0) acc = 0.0;
        if (acc > 1.0) acc = 1.0;
        float g32 = (float)acc;                        // np.float32(blur), getStripe.py:913
        float v = r32 * 0.299f;                        // RGB2GRAY, left to right
        v = v + g32 * 0.587f;
        v = v + g32 * 0.114f;
        gray_img[y * STP_PITCH + x] = v;
    }
}

// ---- wave-strip form of the per-brightness part (bfilter = 3): the 32 x 64 tile is cut into four
// 8-row strips, one per wave, lane = column.  A wave writes the adj values of ITS strip (+1 halo row
// above / below, +1 halo column left / right) into its own LDS slice and then blurs them, so the
// brightness loop needs no workgroup barrier; the blur keeps the row-major tap order of cv.filter2D.
// sg: (GT_Y+2) x (GT_X+2) g plane of the whole tile; srow: ONE row of GT_X+2 products owned by this wave.
#define GS_ROWS 8
// A wave owns an 8-row strip and walks its 10 product rows (strip rows -1 .. 8) top to bottom.  Every tap of
// cv.filter2D multiplies by the same kv = 1/9, so the product kv * adj is formed (and rounded) once per pixel
// and the nine outputs that use it only add.  Only the newest product row travels through LDS (a lane needs its
// neighbours' two columns); the two rows above it stay in the lane's registers.  One row per wave instead of ten
// keeps the kernel's LDS at 20 KB (7 workgroups per CU instead of 4).
struct stp_gray_lane {
    double w0[3], w1[3];           // products of the two previous rows, columns lane .. lane+2
    unsigned vmin, vmax;           // bit patterns of the smallest / largest grey value stored (see STP_FLAT_RANGE)
};
STP_HD void gray_wrow_put(int lane, int strip, int r, double b, const double* sg, double* srow)
{
    const int WW = GT_X + 2;
    const double k = (1.0 - 0.0) / (b - 0.0);
    const double kv = 1.0 / 9.0;
    const double* g = sg + (strip * GS_ROWS + r) * WW;       // strip rows -1 .. 8 are sg rows strip*8 .. +9
    srow[lane + 1] = stp_bright_kv(g[lane + 1], b, k, kv);
    if (lane < 2) {                                           // the two halo columns
        const int c = lane ? WW - 1 : 0;
        srow[c] = stp_bright_kv(g[c], b, k, kv);
    }
}
// Flat-window rule of the Canny stage.  Every smoothed value is a convex combination of the in-image grey
// values inside its (2R+1)^2 window (the bleed-over division renormalises the truncated sums at the image
// border), up to 3e-7 relative from the two float roundings; scipy's Sobel is a difference of two (1,2,1)
// sums of smoothed values, so |isobel|, |jsobel| <= 4 * range and magnitude <= 5.657 * (range + 6e-7), where
// range = max - min of the grey values that can reach the pixel.  A tile whose whole input window has
// range < STP_FLAT_RANGE therefore has every magnitude < 0.085 < 0.1: no pixel gets a class, whatever the
// local-maximum test says, and the tile's four phases are skipped.  k_gray records min / max of the grey
// values it writes per cell of GC_CY x GC_CX pixels; k_canny_pipe unites the cells its window touches.
#define STP_FLAT_RANGE 0.015f
#define GC_CY 8
#define GC_CX 16
#define GC_ROWS (STP_FRAME_MAX / GC_CY)                /* 50 */
#define GC_COLS ((STP_FRAME_MAX + 63) / 64 * 64 / GC_CX) /* 28 */

// The lane's three products of row r; from the third row on they complete the 3 x 3 window of grey pixel
// (strip row r - 2, column lane).  The grey values are >= +0, so their bit patterns order like the values and
// one integer min / max instruction each tracks the range.
STP_HD void gray_wrow_get(int lane, int strip, int r, stp_tile T, const double* srow, stp_gray_lane* st,
                          float* __restrict__ gray_img)
{
    double w2[3];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int c = 0; c < 3; c++) w2[c] = srow[lane + c];
    if (r >= 2) {
        const double kv = 1.0 / 9.0;
        double rb = 0.0;
        for (int t = 0; t < 9; t++) rb = rb + kv * 1.0;
        if (rb < 0.0) rb = 0.0;
        if (rb > 1.0) rb = 1.0;
        const float r32 = (float)rb;
        double acc = 0.0;                                     // row-major tap order of cv.filter2D
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int c = 0; c < 3; c++) acc = acc + st->w0[c];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int c = 0; c < 3; c++) acc = acc + st->w1[c];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int c = 0; c < 3; c++) acc = acc + w2[c];
        acc = fmin(acc, 1.0);                                 // np.clip upper bound (the sum is finite and >= +0: one v_min_f64)
        const float g32 = (float)acc;
        float v = r32 * 0.299f;
        v = v + g32 * 0.587f;
        v = v + g32 * 0.114f;
        const int y = T.ty0 + strip * GS_ROWS + (r - 2), x = T.tx0 + lane;
        if (y < T.S && x < T.S) {
            gray_img[y * STP_PITCH + x] = v;
            unsigned bits;
            memcpy(&bits, &v, 4);
            st->vmin = bits < st->vmin ? bits : st->vmin;
            st->vmax = bits > st->vmax ? bits : st->vmax;
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int c = 0; c < 3; c++) { st->w0[c] = st->w1[c]; st->w1[c] = w2[c]; }
}

// ---- certified grey for the 3 x 3 mean filter (round 4, k_gray_c3) ---------------------------------------------------
// Every vector instruction of k_gray costs the same ~4.25 cycles per wave and SIMD -- an f64 add like an f64 division
// step (tools/ubench_valu.hip) -- so what the kernel pays for is the NUMBER of operations per grey pixel: two f64
// divisions per contact pixel, two compares + two products + nine sequential additions per image pixel.  The grey value
// is the FLOAT rounding of the blurred green plane, though, so -- exactly as stp_gauss_fma.h does for the Gaussian
// passes -- a cheaper f64 evaluation decides it whenever it lands far enough from a float rounding boundary:
//   reference (getStripe.py:889-913, ImageProcessing.py:15-31, cv.filter2D):
//       g = clip(((255 (M - D)) / M) / 255, 0, 1)                         three roundings after M - D
//       adj = 0 if g <= 0, 1 if g > b, else k g     (k = 1 / b)            one
//       blur = (((0 + kv adj_00) + kv adj_01) + ... + kv adj_22)          one per product (kv = 1 / 9), one per addition
//       G = float(clip(blur, 0, 1))
//   here:   g~ = clip((M - D) rM, 0, 1)               rM = RN(1 / M): two roundings after the same M - D
//           a = min(g~, b)
//           S = ((a_00 + a_01) + a_02  +  (a_10 + a_11) + a_12)  +  (a_20 + a_21) + a_22      row sums shared by three outputs
//           blur~ = S c_b                             c_b = RN(k kv)
// Bound (u = 2^-53; every quantity is >= 0, so every bound is relative to the sums themselves).  Let
// T = sum_i kv k min(max(g_i, 0), b) in real arithmetic on the machine numbers kv, k, b, g_i.
//   reference: a product is kv k g_i (1 + 2.01 u) when 0 < g_i <= b; when g_i > b it is kv while the term of T is
//     kv k b = kv (1 + d), |d| <= u (k = RN(1 / b)); nine non-negative terms added in sequence: |blur - T| <= 10.1 u T.
//   here: g~_i = g_i (1 + 5.01 u) (the rounding of M - D is common to both; 1 / M, the product against 255 x, / M, / 255);
//     min(., b) is monotone and 1-Lipschitz, so |a_i - min(g_i, b)| <= 5.01 u min(g_i, b) also when g_i and g~_i lie on
//     different sides of b; four levels of additions: 4.01 u; c_b and the final product: 2.01 u:  |blur~ - T| <= 11.1 u T.
//   |blur~ - blur| <= 21.2 u T < 22 ulp(blur~)  (u x < ulp(x)).  The clip at 1 changes nothing: a sum above 1 exceeds it by
//   ulps and converts to 1.0f like the clipped one.  A non-zero g is >= 2^-53 (M - D is a difference of doubles relative
//   to M), so blur never comes near the float denormals, where the conversion drops more than 29 bits.
// So float(blur~) == float(clip(blur)) unless blur~ lies within 22 ulp(f64) of a float rounding boundary; the test uses
// STP_GRAY_NEAR = 64 (1.2e-7 of the outputs) and a flagged lane redoes its strip of outputs from the band in the
// reference's own operations (stp_gray_exact9).  M must be a normal number for RN(1 / M) to be finite: the host sends any
// other level to the exact kernel.  tests/emu: 4e7 random and adversarial windows, and the golden grey images.
#define STP_GRAY_NEAR 64u
STP_HD double stp_gplane_fast(double D, double M, double rM)
{
    double t = (M - D) * rM;
    t = t < 0.0 ? 0.0 : t;
    return t > 1.0 ? 1.0 : t;
}
STP_HD unsigned stp_near_word(double v, unsigned near)      // see stp_fma_near_word: < 16 * near when within `near` ulp of a boundary
{
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned lo = (unsigned)__double2loint(v);
#else
    unsigned long long bb;
    memcpy(&bb, &v, 8);
    const unsigned lo = (unsigned)bb;
#endif
    return (lo << 3) + ((near - 0x10000000u) << 3);
}
// RGB2GRAY of (1, G, G) in float, left to right; the red plane is the constant 1 through the same filter
STP_HD float stp_gray_rgb(float g32)
{
    const double kv = 1.0 / 9.0;
    double rb = 0.0;
    for (int t = 0; t < 9; t++) rb = rb + kv * 1.0;
    if (rb < 0.0) rb = 0.0;
    if (rb > 1.0) rb = 1.0;
    const float r32 = (float)rb;
    float v = r32 * 0.299f;
    v = v + g32 * 0.587f;
    v = v + g32 * 0.114f;
    return v;
}
// the reference's operations for one grey pixel from its nine contact values (row-major, NaN -> 0 and border reflection
// already applied): the rare path of k_gray_c3 and the checker of its certification test
STP_HD double stp_gray_exact9_blur(const double* D9, double M, double b)
{
    const double k = (1.0 - 0.0) / (b - 0.0), kv = 1.0 / 9.0;
    double acc = 0.0;
    for (int i = 0; i < 9; i++) acc = acc + kv * stp_bright_px(stp_gplane_px(D9[i], M), b, k);
    if (acc < 0.0) acc = 0.0;
    if (acc > 1.0) acc = 1.0;
    return acc;
}
STP_HD float stp_gray_exact9(const double* D9, double M, double b) { return stp_gray_rgb((float)stp_gray_exact9_blur(D9, M, b)); }
// the cheap evaluation of the same pixel from the nine g~ values; *near gets the boundary word of blur~
STP_HD double stp_gray_c3_blur(const double* g9, double b, double cb)
{
    double rs[3];
    for (int r = 0; r < 3; r++) {
        const double a0 = fmin(g9[3 * r], b), a1 = fmin(g9[3 * r + 1], b), a2 = fmin(g9[3 * r + 2], b);
        rs[r] = (a0 + a1) + a2;
    }
    return ((rs[0] + rs[1]) + rs[2]) * cb;
}
STP_HD float stp_gray_c3_px(const double* g9, double b, double cb, unsigned* near)
{
    const double blur = stp_gray_c3_blur(g9, b, cb);
    *near = stp_near_word(blur, STP_GRAY_NEAR);
    return stp_gray_rgb((float)blur);
}
STP_HD double stp_gray_cb(double b) { return ((1.0 - 0.0) / (b - 0.0)) * (1.0 / 9.0); }

// ---------------------------------------------------------------------------------------------
// kernel B (k_canny): grey -> Gaussian (f32 rounding after each axis) -> /bleed -> Sobel ->
// hypot -> NMS -> class bit-planes.   skimage 0.18.3 _canny.py:53-280, scipy ni_filters.c.
// LDS arrays (R = gaussian radius):
//   sG  (CT_Y+2R+4) x GW  f32, GW = CT_X+2R+4, origin (ty0-R-2, tx0-R-2), 0 outside the image
//   sV  (CT_Y+4)    x GW  f32, origin (ty0-2, tx0-R-2)          vertical pass
//   sS  (CT_Y+4) x (CT_X+4) f64, origin (ty0-2, tx0-2)           smoothed
//   sM  (CT_Y+2) x (CT_X+2) f64, origin (ty0-1, tx0-1)           magnitude
STP_HD int ct_gw(int R) { return CT_X + 2 * R + 4; }

STP_HD void canny_p0(int tid, int nt, const float* __restrict__ gray_img, stp_tile T, int R, float* sG)
{
    const int GH = CT_Y + 2 * R + 4, GW = ct_gw(R);
    for (int i = tid; i < GH * GW; i += nt) {
        int yy = i / GW, xx = i - yy * GW;
        int y = T.ty0 - R - 2 + yy, x = T.tx0 - R - 2 + xx;
        sG[i] = (y >= 0 && y < T.S && x >= 0 && x < T.S) ? gray_img[y * STP_PITCH + x] : 0.0f;
    }
}

// NI_Correlate1D symmetric branch: o = x[0]*w[c]; for k = R..1: o += (x[-k] + x[k]) * w[c-k]
STP_HD void canny_p1(int tid, int nt, stp_tile T, int R, const double* w, const float* sG, float* sV)
{
    const int VH = CT_Y + 4, GW = ct_gw(R);
    for (int i = tid; i < VH * GW; i += nt) {
        int yy = i / GW, xx = i - yy * GW;
        int y = T.ty0 - 2 + yy, x = T.tx0 - R - 2 + xx;
        float out = 0.0f;
        if (y >= 0 && y < T.S && x >= 0 && x < T.S) {
            const float* c = sG + (yy + R) * GW + xx;
            double o = (double)c[0] * w[R];
            for (int k = R; k >= 1; k--) o += ((double)c[-k * GW] + (double)c[k * GW]) * w[R - k];
            out = (float)o;
        }
        sV[i] = out;
    }
}

// bleed_over = gaussian(ones): column factor, then row pass on the constant row (f64)
STP_HD double stp_bleed_v(int y, int S, int R, const double* w)
{
    double o = 1.0 * w[R];
    for (int k = R; k >= 1; k--) {
        double a = (y - k >= 0) ? 1.0 : 0.0, b = (y + k < S) ? 1.0 : 0.0;
        o += (a + b) * w[R - k];
    }
    return o;
}
STP_HD double stp_bleed_h(double V, int x, int S, int R, const double* w)
{
    double o = V * w[R];
    for (int k = R; k >= 1; k--) {
        double a = (x - k >= 0) ? V : 0.0, b = (x + k < S) ? V : 0.0;
        o += (a + b) * w[R - k];
    }
    return o;
}

// sB[0..VH) = column factor V(y); sB[VH..2VH) = full bleed for an interior x (x-R>=0, x+R<S)
STP_HD void canny_p1b(int tid, int nt, stp_tile T, int R, const double* w, double* sB)
{
    const int VH = CT_Y + 4;
    for (int i = tid; i < VH; i += nt) {
        int y = T.ty0 - 2 + i;
        double V = 0.0, BI = 0.0;
        if (y >= 0 && y < T.S) {
            V = stp_bleed_v(y, T.S, R, w);
            if (T.S >= 2 * R + 1) BI = stp_bleed_h(V, R, T.S, R, w);
        }
        sB[i] = V;
        sB[VH + i] = BI;
    }
}

STP_HD void canny_p2(int tid, int nt, stp_tile T, int R, const double* w, const float* sV, const double* sB,
                     double* sS)
{
    const int VH = CT_Y + 4, GW = ct_gw(R), SW = CT_X + 4;
    for (int i = tid; i < VH * SW; i += nt) {
        int yy = i / SW, xx = i - yy * SW;
        int y = T.ty0 - 2 + yy, x = T.tx0 - 2 + xx;
        double s = 0.0;
        if (y >= 0 && y < T.S && x >= 0 && x < T.S) {
            const float* c = sV + yy * GW + (xx + R);
            double o = (double)c[0] * w[R];
            for (int k = R; k >= 1; k--) o += ((double)c[-k] + (double)c[k]) * w[R - k];
            float f = (float)o;
            double bl = (x >= R && x + R < T.S) ? sB[VH + yy] : stp_bleed_h(sB[yy], x, T.S, R, w);
            s = (double)f / (bl + DBL_EPSILON);       // _canny.py:49
        }
        sS[yy * CT_SP + xx] = s;
    }
}

#include "stp_gauss_fma.h"

// ---- register-blocked Gaussian passes (compile-time radius): each thread produces a run of
// consecutive outputs along the filter direction from a sliding window held in registers, so an
// input is loaded and widened to f64 once instead of 2R+1 times.  Same operation order per output.
// The vertical pass writes its result TRANSPOSED (sVT[x][y], pitch CT_VP) so that both passes read
// LDS with consecutive lanes on consecutive words.
#define STP_GRAY_GUARD 65536   /* bytes of padding before and after the grey images: the last tile row reads up to
                                  ty0 + CT_Y + R + 1 = 429 < 400 + 40 rows of 1600 B (R <= 12); R + 2 rows in front */
#define CT_VRUN 12   /* vertical outputs per thread: (CT_Y + 4) = 3 * 12 */
#define CT_HRUN_R(R) ((R) <= 8 ? 10 : 5)   /* horizontal outputs per thread: 7 runs x 36 rows = 252 items (one round) at
                                             radius <= 8; radius 10 would spill with a 30-double window: 14 runs of 5 */
/* columns of the transposed vertical-pass buffer the horizontal pass may touch (runs are whole: runs x HRUN + 2R) */
#define CT_P2_COLS(R) ((((CT_X + 4) + CT_HRUN_R(R) - 1) / CT_HRUN_R(R)) * CT_HRUN_R(R) + 2 * (R))
// ---- valid extents of a tile (the work items of the three passes cover only what lies inside the image) ----
// A 400-pixel frame is 7 x 13 tiles of 64 x 32, so the last tile column holds 16 image columns and the last tile row
// 16 image rows (fewer after zero-column removal): 19 of the 91 tiles are mostly outside the image.  Their passes used
// to run over the full tile on zeros; here every pass numbers only its in-image items, densely, so that whole waves
// of such a tile have nothing to do.  q = n / d for the few run-time divisors: multiply by ceil(2^22 / d)
// (exact for n * d < 2^22; n < 4096, d <= 128 here).
struct stp_udiv { unsigned d, m; };
STP_HD stp_udiv stp_udiv_make(int d) { stp_udiv u; u.d = (unsigned)d; u.m = d > 0 ? (unsigned)(((1u << 22) + (unsigned)d - 1u) / (unsigned)d) : 0u; return u; }
STP_HD unsigned stp_udiv_q(unsigned n, stp_udiv u) { return (unsigned)(((unsigned long long)n * u.m) >> 22); }
struct stp_cgeo {
    int c_lo, ng;        // vertical pass: first in-image column of the tile's window (sVT index), row groups in use
    stp_udiv ncv, nzc;   //   number of in-image columns / of the columns outside the image (written as zeros)
    int yy_lo, xg_lo, nruns;   // horizontal pass: first in-image row of the smoothed tile, first run, number of runs
    stp_udiv nrv;        //   number of in-image rows
    int my_lo, nmh, mx_lo;     // magnitude tile: in-image region [my_lo, my_lo + nmh) x [mx_lo, mx_lo + nmw)
    stp_udiv nmw;
    int nms_rows;        // rows of the tile that can hold an interior pixel (y <= S-2): the NMS collects only those
};
template <int R>
STP_HD stp_cgeo ct_geo(stp_tile T)
{
    constexpr int VRUN = (R <= 8) ? CT_VRUN : CT_VRUN / 2;
    constexpr int HRUN = CT_HRUN_R(R);
    const int GW = CT_X + 2 * R + 4, VH = CT_Y + 4, SW = CT_X + 4;
    stp_cgeo g;
    g.c_lo = R + 2 - T.tx0 > 0 ? R + 2 - T.tx0 : 0;                       // x = tx0 - R - 2 + xx >= 0
    const int c_hi = T.S - T.tx0 + R + 2 < GW ? T.S - T.tx0 + R + 2 : GW;  // x < S
    g.ncv = stp_udiv_make(c_hi - g.c_lo);
    g.nzc = stp_udiv_make(GW - (c_hi - g.c_lo));
    g.yy_lo = 2 - T.ty0 > 0 ? 2 - T.ty0 : 0;                               // y = ty0 - 2 + yy >= 0
    const int yy_hi = T.S - T.ty0 + 2 < VH ? T.S - T.ty0 + 2 : VH;         // y < S
    g.nrv = stp_udiv_make(yy_hi - g.yy_lo);
    g.ng = (yy_hi + VRUN - 1) / VRUN;
    const int xx_lo = 2 - T.tx0 > 0 ? 2 - T.tx0 : 0, xx_hi = T.S - T.tx0 + 2 < SW ? T.S - T.tx0 + 2 : SW;
    g.xg_lo = xx_lo / HRUN;
    g.nruns = (xx_hi + HRUN - 1) / HRUN - g.xg_lo;
    g.my_lo = 1 - T.ty0 > 0 ? 1 - T.ty0 : 0;
    g.nmh = (T.S - T.ty0 + 1 < CT_Y + 2 ? T.S - T.ty0 + 1 : CT_Y + 2) - g.my_lo;
    g.mx_lo = 1 - T.tx0 > 0 ? 1 - T.tx0 : 0;
    g.nmw = stp_udiv_make((T.S - T.tx0 + 1 < CT_X + 2 ? T.S - T.tx0 + 1 : CT_X + 2) - g.mx_lo);
    g.nms_rows = T.S - 1 - T.ty0 < CT_Y ? T.S - 1 - T.ty0 : CT_Y;
    return g;
}

// Item numbering of the vertical pass: (in-image column, row group) pairs first, then one zero-fill item per
// (outside column, row group) -- the horizontal pass reads those columns as the constant-mode zeros.
// Returns xx | yy0 << 8 | zero << 16, or -1 past the last item.
template <int R>
STP_HD int ct_p1_decode(stp_cgeo G, int i)
{
    constexpr int VRUN = (R <= 8) ? CT_VRUN : CT_VRUN / 2;   // radius 10 (sigma 2.5) would spill with 12 outputs per lane
    const int nval = (int)G.ncv.d * G.ng, nzero = (int)G.nzc.d * G.ng;
    if (i < nval) {
        const int yg = (int)stp_udiv_q((unsigned)i, G.ncv), xx = G.c_lo + (i - yg * (int)G.ncv.d);
        return xx | (yg * VRUN) << 8;
    }
    if (i < nval + nzero) {
        const int j = i - nval;
        const int yg = (int)stp_udiv_q((unsigned)j, G.nzc), zc = j - yg * (int)G.nzc.d;
        const int xx = zc < G.c_lo ? zc : zc + (int)G.ncv.d;
        return xx | (yg * VRUN) << 8 | 1 << 16;
    }
    return -1;
}
template <int R>
STP_HD void canny_p1_zero(int xx, int yy0, float* sVT)
{
    constexpr int VRUN = (R <= 8) ? CT_VRUN : CT_VRUN / 2;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < VRUN; q++) sVT[xx * CT_VP + yy0 + q] = 0.0f;
}
// vertical pass of one item, reading the grey image directly (no LDS copy of the tile): consecutive lanes read
// consecutive columns of one image row (coalesced); rows shared by neighbouring groups come from L1/L2.
// YIN: every row this tile touches (ty0-R-2 .. ty0+CT_Y+R+1) lies inside the image -> no clamping.
template <int R, bool YIN>
STP_HD void canny_p1_item(stp_tile T, int xx, int yy0, const double* w, const float* __restrict__ gimg, float* sVT)
{
    constexpr int VRUN = (R <= 8) ? CT_VRUN : CT_VRUN / 2;
    const int x = T.tx0 - R - 2 + xx;
    float raw[VRUN + 2 * R];
    // Rows outside the image are loaded like any other (the grey buffer carries STP_GRAY_GUARD bytes of
    // padding on both sides, so a row up to R+2 above the first image or CT_Y+R+1 below the start of the last
    // tile row is still inside the allocation) and replaced by the constant-mode 0 after the load: no
    // per-element address clamping in the border tiles.
    const float* col = gimg + (T.ty0 - R - 2 + yy0) * STP_PITCH + x;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < VRUN + 2 * R; k++) raw[k] = col[k * STP_PITCH];        // all loads issued before any use
    if (!YIN) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = 0; k < VRUN + 2 * R; k++) {
            const int y = T.ty0 - R - 2 + yy0 + k;
            if ((unsigned)y >= (unsigned)T.S) raw[k] = 0.0f;
        }
    }
    double win[VRUN + 2 * R];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < VRUN + 2 * R; k++) win[k] = (double)raw[k];
    float outv[VRUN];
    const unsigned far = stp_gauss_run_fma<R, VRUN>(win, w, outv);       // certified fused sums (stp_gauss_fma.h)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < VRUN; q++) {
        float out = outv[q];
        if (!YIN) {
            const int y = T.ty0 - 2 + yy0 + q;
            if (!(y >= 0 && y < T.S)) out = 0.0f;
        }
        sVT[xx * CT_VP + yy0 + q] = out;
    }
    if (far == 0) {                               // some output is near a float rounding boundary: the exact order decides
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int q = 0; q < VRUN; q++) {
            const int y = T.ty0 - 2 + yy0 + q;    // output row; window rows y-R .. y+R (zero outside the image)
            if (y < 0 || y >= T.S) continue;
            const int lo = y - R < 0 ? R - y : 0, hi = y + R >= T.S ? R + (T.S - 1 - y) : 2 * R;
            sVT[xx * CT_VP + yy0 + q] = stp_gauss_exact(gimg + y * STP_PITCH + x, STP_PITCH, R, w, (unsigned)lo, (unsigned)hi);
        }
    }
}
// the whole pass as a strided loop over the item numbers (tests/emu; the kernel decodes each thread's items once
// per workgroup -- the numbering depends on the tile only -- and calls the item functions)
template <int R, bool YIN>
STP_HD void canny_p1_blk_g(int tid, int nt, stp_tile T, stp_cgeo G, const double* w, const float* __restrict__ gimg, float* sVT)
{
    for (int i = tid;; i += nt) {
        const int it = ct_p1_decode<R>(G, i);
        if (it < 0) break;
        if (it >> 16) canny_p1_zero<R>(it & 255, (it >> 8) & 255, sVT);
        else canny_p1_item<R, YIN>(T, it & 255, (it >> 8) & 255, w, gimg, sVT);
    }
}

// Bleed-over of the border columns of the image (x < R or x >= S-R), tabulated once per workgroup for
// the VH rows of the tile: sBB[yy * 2R + q], q = x (left border) or R + x - (S-R) (right border).
// The interior value of each row is sB[VH + yy] (canny_p1b).  Geometry only: shared by all images.
template <int R>
STP_HD void canny_p1c(int tid, int nt, stp_tile T, const double* w, const double* sB, double* sBB)
{
    const int VH = CT_Y + 4;
    for (int i = tid; i < VH * 2 * R; i += nt) {
        const int yy = i / (2 * R), q = i - yy * (2 * R);
        const int x = q < R ? q : T.S - R + (q - R);
        double v = 0.0;
        if (x >= 0 && x < T.S) v = stp_bleed_h(sB[yy], x, T.S, R, w);
        sBB[i] = v;
    }
}

// XIN: every column of the smoothed tile (tx0-2 .. tx0+CT_X+1) is an interior column (x-R >= 0 and
// x+R < S) -> no in-image test and no border bleed-over in the inner loop.
// Division by the bleed-over of an interior pixel (one constant c for every interior pixel of every
// image): q = f * rc, r = fma(-q, c, f), q' = fma(r, rc, q) with rc = RN(1 / c) is the correctly rounded
// quotient (Markstein).  It is used only when the host has verified it against the true division for all
// 2^23 float mantissas (the numerator is always a float; the quotient scales exactly with its exponent).
struct stp_fastdiv {
    double c, rc;
    int ok;
};
STP_HD double stp_div_const(double f, double c, double rc)
{
    const double q = f * rc;
    const double r = fma(-q, c, f);
    return fma(r, rc, q);
}

// Item numbering of the horizontal pass: (in-image row, run) pairs, numbered densely; rows outside the image are
// never read (canny_p3_ring supplies the one-pixel ring, the magnitude pass covers the in-image region only).
// Returns yy | xx0 << 8, or -1 past the last item.
template <int R>
STP_HD int ct_p2_decode(stp_cgeo G, int i)
{
    constexpr int HRUN = CT_HRUN_R(R);
    if (i >= (int)G.nrv.d * G.nruns) return -1;
    const int xq = (int)stp_udiv_q((unsigned)i, G.nrv);
    return (G.yy_lo + (i - xq * (int)G.nrv.d)) | ((G.xg_lo + xq) * HRUN) << 8;
}
template <int R, bool XIN>
STP_HD void canny_p2_item(stp_tile T, int yy, int xx0, const double* w, const float* sVT, const double* sB,
                          const double* sBB, double* sS, stp_fastdiv fd)
{
    constexpr int HRUN = CT_HRUN_R(R);
    static_assert(((CT_X + 4 + HRUN - 1) / HRUN) * HRUN <= CT_SP, "the last horizontal run must fit the row pitch");
    const int VH = CT_Y + 4;
    double win[HRUN + 2 * R];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < HRUN + 2 * R; k++)         // sVT column index = image x - (tx0 - R - 2)
        win[k] = (double)sVT[(xx0 + k) * CT_VP + yy];  // columns >= CT_X+2R+4 (last run only) lie in the
                                                       // buffer's CT_P2_COLS(R) padding and feed only the
                                                       // outputs xx >= SW that are dropped below
    float fv[HRUN];
    if (stp_gauss_run_fma<R, HRUN>(win, w, fv) == 0) {    // rare: the run is settled in the exact order (from LDS;
#if defined(__HIP_DEVICE_COMPILE__)                       // every sVT column of the window exists, 0 outside the image)
#pragma unroll
#endif
        for (int q = 0; q < HRUN; q++) fv[q] = stp_gauss_exact(sVT + (xx0 + q + R) * CT_VP + yy, CT_VP, R, w, 0u, (unsigned)(2 * R));
    }
    const double bint = sB[VH + yy] + DBL_EPSILON;
    // (tile-uniform XIN, or this item's own columns) interior columns: the verified constant
    const int xfirst = T.tx0 - 2 + xx0;
    const bool cin = XIN || (xfirst >= R && xfirst + HRUN - 1 + R < T.S);
    const bool fast = cin && fd.ok && (bint == fd.c);
    double* srow = sS + yy * CT_SP + xx0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < HRUN; q++) {            // (no test for the last run's outputs beyond the tile's CT_X + 4 columns: they land in
        const int xx = xx0 + q;                 //  the row's padding, which nothing reads -- a test here keeps the loop from being unrolled)
        const float f = fv[q];
        double s;
        if (fast) {
            s = stp_div_const((double)f, fd.c, fd.rc);
        } else if (XIN) {
            s = (double)f / bint;
        } else {
            const int x = T.tx0 - 2 + xx;
            s = 0.0;
            if (x >= 0 && x < T.S) {
                double bl = bint;
                if (x < R) bl = sBB[yy * 2 * R + x] + DBL_EPSILON;
                else if (x + R >= T.S) bl = sBB[yy * 2 * R + R + (x - (T.S - R))] + DBL_EPSILON;
                s = (double)f / bl;
            }
        }
        srow[q] = s;
    }
}
template <int R, bool XIN>
STP_HD void canny_p2_blk(int tid, int nt, stp_tile T, stp_cgeo G, const double* w, const float* sVT, const double* sB,
                         const double* sBB, double* sS, stp_fastdiv fd)
{
    for (int i = tid;; i += nt) {
        const int it = ct_p2_decode<R>(G, i);
        if (it < 0) break;
        canny_p2_item<R, XIN>(T, it & 255, it >> 8, w, sVT, sB, sBB, sS, fd);
    }
}

// glibc 2.35 sysdeps/ieee754/dbl-64/e_hypot.c (non-FMA kernel), what numpy's np.hypot calls.
// Straight-line form: glibc's three paths (huge: scale by 2^-600, tiny: by 2^600, normal: none) differ
// only by an exact power-of-two scaling around the same kernel, so one path with a selected scale factor
// (1.0 in the normal range) returns identical bits; the "ax >= ay / EPS -> ax + ay" shortcut and the
// kernel's "h <= 2 ay" alternative are selects.  No branches: independent calls interleave.
STP_HD double stp_hypot(double x, double y)
{
    x = fabs(x); y = fabs(y);
    const double ax = x < y ? y : x, ay = x < y ? x : y;
    const bool huge = ax > 0x1p+511, tiny = ay < 0x1p-459;
    const double sc = huge ? 0x1p-600 : (tiny ? 0x1p+600 : 1.0);
    const double us = huge ? 0x1p+600 : (tiny ? 0x1p-600 : 1.0);
    // huge: ay <= ax * EPS ; otherwise: ax >= ay / EPS  (same predicate up to exact scaling)
    const bool far_apart = huge ? (ay <= ax * 0x1p-54) : (ax >= ay * 0x1p+54);
    const double a = ax * sc, b = ay * sc;
    double h = sqrt(a * a + b * b);
    const double dA = h - b, dB = h - a;
    const double t1A = a * (2.0 * dA - a), t2A = (dA - 2.0 * (a - b)) * dA;
    const double t1B = 2.0 * dB * (a - 2.0 * b), t2B = (4.0 * dB - b) * b + dB * dB;
    const bool selA = h <= 2.0 * b;
    const double t1 = selA ? t1A : t1B, t2 = selA ? t2A : t2B;
    h -= (t1 + t2) / (2.0 * h);
    h = h * us;
    return far_apart ? (ax + ay) : h;
}

// smoothed value at image (y, x) with scipy 'reflect' (only +-1 overshoot is ever requested)
STP_HD double ct_s(const double* sS, stp_tile T, int y, int x)
{
    int ry = stp_refl(y, T.S), rx = stp_refl(x, T.S);
    return sS[(ry - (T.ty0 - 2)) * CT_SP + (rx - (T.tx0 - 2))];
}

// ndi.sobel: antisymmetric pass o = x[0]*0 + (x[-1]-x[1])*(-1) (== x[1]-x[-1]); symmetric pass
// o = x[0]*2 + (x[-1]+x[1])*1.   jsobel = sobel(axis=1), isobel = sobel(axis=0) (_canny.py:183-184).
// `c` points at the pixel inside the smoothed tile; up/dn/lf/rt are the offsets of its neighbours.
STP_HD void ct_sobel_off(const double* c, int up, int dn, int lf, int rt, double* is, double* js)
{
    double s00 = c[up + lf], s01 = c[up], s02 = c[up + rt], s10 = c[lf], s12 = c[rt], s20 = c[dn + lf], s21 = c[dn],
           s22 = c[dn + rt];
    double dm = (s00 - s02) * -1.0, d0 = (s10 - s12) * -1.0, dp = (s20 - s22) * -1.0;
    double j = d0 * 2.0;
    j += (dm + dp) * 1.0;
    double em = (s00 - s20) * -1.0, e0 = (s01 - s21) * -1.0, ep = (s02 - s22) * -1.0;
    double i = e0 * 2.0;
    i += (em + ep) * 1.0;
    *is = i; *js = j;
}
// interior pixel: plain offsets
STP_HD void ct_sobel_in(const double* c, double* is, double* js) { ct_sobel_off(c, -CT_SP, CT_SP, -1, 1, is, js); }
// any in-image pixel: scipy's 'reflect' with an overshoot of one is a clamp to the edge pixel
STP_HD void ct_sobel(const double* sS, stp_tile T, int y, int x, double* is, double* js)
{
    const double* c = sS + (y - (T.ty0 - 2)) * CT_SP + (x - (T.tx0 - 2));
    ct_sobel_off(c, y > 0 ? -CT_SP : 0, y < T.S - 1 ? CT_SP : 0, x > 0 ? -1 : 0, x < T.S - 1 ? 1 : 0, is, js);
}

// approximate quotient of two non-negative floats with num <= den (hardware reciprocal on the device)
STP_HD float stp_fdiv32(float num, float den)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return num * __builtin_amdgcn_rcpf(den);
#else
    return num / den;
#endif
}
// approximate magnitude (float, one hardware sqrt): relative error < 3e-7, see ct_nms
STP_HD float stp_mag32(double is, double js)
{
    const float q = (float)(is * is + js * js);
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(q);
#else
    return sqrtf(q);
#endif
}
STP_HD void canny_p3_in(int tid, int nt, const double* sS, float* sM)
{
    const int MH = CT_Y + 2, MW = CT_X + 2;
    for (int i = tid; i < MH * MW; i += nt) {
        const int yy = i / MW, xx = i - yy * MW;
        double is, js;
        ct_sobel_in(sS + (yy + 1) * CT_SP + (xx + 1), &is, &js);
        sM[i] = stp_mag32(is, js);
    }
}
// ---- magnitudes in 2 x 2 blocks over the in-image region of the magnitude tile ----
// A lane forms the four magnitudes M(yy..yy+1, xx..xx+1) from the 4 x 4 smoothed values around them.  Every output
// is computed by ct_sobel_off's own operations in its own order; what the four outputs share are the eight row
// differences (s[r][c-1] - s[r][c+1]) * -1 and the eight column differences (s[r-1][c] - s[r+1][c]) * -1, each
// formed once: 40 instead of 60 FP64 operations and 16 instead of 32 LDS values per four pixels.
// `c` points at the block's top-left smoothed value, S(yy, xx) = neighbour (-1, -1) of pixel M(yy, xx).
STP_HD void ct_sobel_blk2(const double* c, float* m /* [2][2] row-major */)
{
    double s[4][4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 4; r++) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int q = 0; q < 4; q++) s[r][q] = c[r * CT_SP + q];
    }
    double H[4][2], V[2][4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 4; r++) { H[r][0] = (s[r][0] - s[r][2]) * -1.0; H[r][1] = (s[r][1] - s[r][3]) * -1.0; }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < 4; q++) { V[0][q] = (s[0][q] - s[2][q]) * -1.0; V[1][q] = (s[1][q] - s[3][q]) * -1.0; }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 2; r++) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int q = 0; q < 2; q++) {
            double j = H[r + 1][q] * 2.0;
            j += (H[r][q] + H[r + 2][q]) * 1.0;
            double i = V[r][q + 1] * 2.0;
            i += (V[r][q] + V[r][q + 2]) * 1.0;
            m[r * 2 + q] = stp_mag32(i, j);
        }
    }
}
// Walk over the blocks of the region [my_lo, my_lo + nmh) x [mx_lo, mx_lo + nmw) (both extents >= 2): ceil(nmh / 2) x
// ceil(nmw / 2) blocks numbered densely; the last block row / column of an odd extent is shifted back by one pixel
// (it recomputes one row / column with identical values).  The (row, column) of an item advances by the stride.
struct stp_p3walk { int n, nbw, dr, dc, r0, c0, nmh, nmw, soff, moff, my_lo, mx_lo; };
STP_HD stp_p3walk ct_p3_walk(stp_cgeo G, int tid, int nt)
{
    stp_p3walk W;
    W.nmh = G.nmh; W.nmw = (int)G.nmw.d;
    W.nbw = (W.nmw + 1) / 2;
    W.n = ((W.nmh + 1) / 2) * W.nbw;
    const stp_udiv u = stp_udiv_make(W.nbw);
    W.dr = (int)stp_udiv_q((unsigned)nt, u); W.dc = nt - W.dr * W.nbw;
    W.r0 = (int)stp_udiv_q((unsigned)tid, u); W.c0 = tid - W.r0 * W.nbw;
    W.soff = G.my_lo * CT_SP + G.mx_lo;
    W.moff = G.my_lo * (CT_X + 2) + G.mx_lo;
    W.my_lo = G.my_lo; W.mx_lo = G.mx_lo;
    return W;
}
STP_HD void canny_p3_walk(int tid, int nt, stp_p3walk W, const double* sS, float* sM)
{
    int r = W.r0, c = W.c0;
    for (int i = tid; i < W.n; i += nt) {
        const int y = 2 * r < W.nmh - 2 ? 2 * r : W.nmh - 2, x = 2 * c < W.nmw - 2 ? 2 * c : W.nmw - 2;
        float m[4];
        ct_sobel_blk2(sS + W.soff + y * CT_SP + x, m);
        float* o = sM + W.moff + y * (CT_X + 2) + x;
        o[0] = m[0]; o[1] = m[1]; o[CT_X + 2] = m[2]; o[CT_X + 3] = m[3];
        c += W.dc; r += W.dr;
        if (c >= W.nbw) { c -= W.nbw; r++; }
    }
}
STP_HD void canny_p3_reg(int tid, int nt, stp_cgeo G, const double* sS, float* sM)
{
    canny_p3_walk(tid, nt, ct_p3_walk(G, tid, nt), sS, sM);
}

// Border tiles: scipy's 'reflect' with an overshoot of one pixel is edge replication, so writing the
// one-pixel ring around the image (rows -1 and S, columns -1 and S, where this tile holds them) with its
// clamped in-image source lets canny_p3_in's plain offsets serve every tile.  Sources are in-image
// elements (never written here); a corner is written twice with the same value.  Magnitudes of pixels
// outside the image come out as garbage instead of 0: the NMS never reads them.
STP_HD void canny_p3_ring(int tid, int nt, stp_tile T, double* sS)
{
    const int VH = CT_Y + 4, SW = CT_X + 4;
    const int y0 = T.ty0 - 2, x0 = T.tx0 - 2;
    for (int i = tid; i < 2 * SW + 2 * VH; i += nt) {
        int y, x;
        if (i < 2 * SW) { y = (i < SW) ? -1 : T.S; x = x0 + (i < SW ? i : i - SW); }
        else { const int j = i - 2 * SW; x = (j < VH) ? -1 : T.S; y = y0 + (j < VH ? j : j - VH); }
        const int yy = y - y0, xx = x - x0;
        if (yy < 0 || yy >= VH || xx < 0 || xx >= SW) continue;          // this tile does not hold that ring element
        const int cy = y < 0 ? 0 : (y > T.S - 1 ? T.S - 1 : y), cx = x < 0 ? 0 : (x > T.S - 1 ? T.S - 1 : x);
        const int sy = cy - y0, sx = cx - x0;
        if (sy < 0 || sy >= VH || sx < 0 || sx >= SW) continue;
        sS[yy * CT_SP + xx] = sS[sy * CT_SP + sx];
    }
}

STP_HD void canny_p3(int tid, int nt, stp_tile T, const double* sS, float* sM)
{
    const int MH = CT_Y + 2, MW = CT_X + 2;
    for (int i = tid; i < MH * MW; i += nt) {
        int yy = i / MW, xx = i - yy * MW;
        int y = T.ty0 - 1 + yy, x = T.tx0 - 1 + xx;
        float m = 0.0f;
        if (y >= 0 && y < T.S && x >= 0 && x < T.S) {
            double is, js;
            ct_sobel(sS, T, y, x, &is, &js);
            m = stp_mag32(is, js);
        }
        sM[i] = m;
    }
}

// _canny.py:193-280: interior & magnitude>0, four overlapping sectors (later ones override),
// bilinear interpolation with `<=`, thresholds 0.1 / 0.2 with `>=`.  Returns 0 / 1 (low) / 2 (high).
//
// The magnitude tile sM holds a FLOAT approximation h0 of numpy's hypot(is, js): the f64 sum of squares
// rounded to float and one hardware square root, relative error < 3e-7 (2^-24 for the rounding, at most
// 1 ulp(float) for v_sqrt_f32).  A decision taken from h0 values equals the reference's whenever the
// compared quantities differ by more than the propagated error (< 1e-6 of the largest magnitude
// involved; the float interpolation weight adds < 3e-7 of it).  We accept it only when the gap exceeds 1e-5 of that magnitude and the thresholds 0.1 /
// 0.2 are missed by more than 1e-6; otherwise (ties, plateaus, near-ties) the five magnitudes are
// recomputed with the exact glibc kernel (stp_hypot) and the reference's test is evaluated literally.
// The fallback is taken by ~1e-5 of the candidates on noisy data, by thousands on synthetic plateaus.
STP_HD int ct_nms(const double* sS, const float* sM, stp_tile T, int y, int x)
{
    if (y < 1 || x < 1 || y >= T.S - 1 || x >= T.S - 1) return 0;
    const int MW = CT_X + 2;
    const float* mp = sM + (y - (T.ty0 - 1)) * MW + (x - (T.tx0 - 1));
    const double m0 = (double)mp[0];
    if (m0 < 0.1 - 1e-6) return 0;        // exact m < 0.1: class 0 whatever the local-max test says
    double gi, gj;                          // (y, x) is interior: direct indexing
    ct_sobel_in(sS + (y - (T.ty0 - 2)) * CT_SP + (x - (T.tx0 - 2)), &gi, &gj);
    const double ai = fabs(gi), aj = fabs(gj);
    const bool same = (gi >= 0 && gj >= 0) || (gi <= 0 && gj <= 0);
    const bool opp = (gi <= 0 && gj >= 0) || (gi >= 0 && gj <= 0);
    // the LAST matching sector decides (assignment order in _canny.py); the "minus" side neighbours are
    // the mirror images of the "plus" side ones
    int dy1, dx1, dy2, dx2;
    double num, den;
    if (opp && ai >= aj) { num = aj; den = ai; dy1 = -1; dx1 = 0; dy2 = -1; dx2 = 1; }        // 135-180
    else if (opp && ai <= aj) { num = ai; den = aj; dy1 = 0; dx1 = 1; dy2 = -1; dx2 = 1; }    // 90-135
    else if (same && ai <= aj) { num = ai; den = aj; dy1 = 0; dx1 = 1; dy2 = 1; dx2 = 1; }    // 45-90
    else if (same && ai >= aj) { num = aj; den = ai; dy1 = 1; dx1 = 0; dy2 = 1; dx2 = 1; }    // 0-45
    else return 0;
    const int o1 = dy1 * MW + dx1, o2 = dy2 * MW + dx2;
    // interpolation weight: a float quotient (relative error < 3e-7, |w| <= 1) is enough for the certified
    // test below -- it moves lp / lm by < 3e-7 of the largest magnitude; the exact quotient is formed only
    // when the decision has to be re-evaluated literally
    double wq = (double)stp_fdiv32((float)num, (float)den);
    double omw = 1.0 - wq;
    double m = m0, c1p = (double)mp[o1], c2p = (double)mp[o2], c1m = (double)mp[-o1], c2m = (double)mp[-o2];
    double lp = c2p * wq + c1p * omw, lm = c2m * wq + c1m * omw;
    double big = m0;
    big = c1p > big ? c1p : big; big = c2p > big ? c2p : big; big = c1m > big ? c1m : big; big = c2m > big ? c2m : big;
    const double tol = big * 1e-5;
    const bool certain = fabs(lp - m0) > tol && fabs(lm - m0) > tol && fabs(m0 - 0.1) > 1e-6 && fabs(m0 - 0.2) > 1e-6;
    if (!certain) {                         // exact re-evaluation (glibc hypot of the five pixels)
        double is, js;
        wq = num / den; omw = 1.0 - wq;
        m = stp_hypot(gi, gj);
        ct_sobel(sS, T, y + dy1, x + dx1, &is, &js); c1p = stp_hypot(is, js);
        ct_sobel(sS, T, y + dy2, x + dx2, &is, &js); c2p = stp_hypot(is, js);
        ct_sobel(sS, T, y - dy1, x - dx1, &is, &js); c1m = stp_hypot(is, js);
        ct_sobel(sS, T, y - dy2, x - dx2, &is, &js); c2m = stp_hypot(is, js);
        lp = c2p * wq + c1p * omw; lm = c2m * wq + c1m * omw;
        if (!(m > 0.0)) return 0;
    }
    if (!(lp <= m && lm <= m)) return 0;
    return (m >= 0.2) ? 2 : ((m >= 0.1) ? 1 : 0);
}

// class of every tile pixel into an LDS byte tile (CT_Y x CT_X)
STP_HD void canny_p4(int tid, int nt, stp_tile T, const double* sS, const float* sM, uint8_t* sC)
{
    for (int i = tid; i < CT_Y * CT_X; i += nt) {
        int yy = i / CT_X, xx = i - yy * CT_X;
        int y = T.ty0 + yy, x = T.tx0 + xx;
        sC[i] = (y < T.S && x < T.S) ? (uint8_t)ct_nms(sS, sM, T, y, x) : 0;
    }
}

// pack the byte tile into the two global bit-planes (one u64 word per tile row; CT_X == 64)
STP_HD void canny_p5(int tid, int nt, stp_tile T, const uint8_t* sC, stp_u64* __restrict__ low_img,
                     stp_u64* __restrict__ high_img)
{
    for (int yy = tid; yy < CT_Y; yy += nt) {
        int y = T.ty0 + yy;
        if (y >= T.S) continue;
        stp_u64 lo = 0, hi = 0;
        for (int xx = 0; xx < CT_X; xx++) {
            uint8_t c = sC[yy * CT_X + xx];
            lo |= (stp_u64)(c >= 1) << xx;
            hi |= (stp_u64)(c == 2) << xx;
        }
        low_img[STP_CLS(y, T.tx0 >> 6)] = lo;
        high_img[STP_CLS(y, T.tx0 >> 6)] = hi;
    }
}

// ---------------------------------------------------------------------------------------------
// kernel C (k_lines): one workgroup per image, everything bit-packed in LDS.
// Bit matrices: S rows x STP_NW u64 words, bit x of row r = word[x>>6] >> (x&63); bits >= S are 0.
STP_HD stp_u64 bm_shl1(const stp_u64* row, int w)   // result bit x = row bit x-1
{
    return (row[w] << 1) | (w > 0 ? row[w - 1] >> 63 : 0ull);
}
STP_HD stp_u64 bm_shr1(const stp_u64* row, int w)   // result bit x = row bit x+1
{
    return (row[w] >> 1) | (w + 1 < STP_NW ? row[w + 1] << 63 : 0ull);
}
STP_HD stp_u64 bm_shr2(const stp_u64* row, int w)   // result bit x = row bit x+2
{
    return (row[w] >> 2) | (w + 1 < STP_NW ? row[w + 1] << 62 : 0ull);
}
STP_HD int bm_get(const stp_u64* m, int r, int x) { return (int)((m[r * STP_NW + (x >> 6)] >> (x & 63)) & 1ull); }
STP_HD stp_u64 stp_brev64(stp_u64 v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(v);
#else
    v = ((v >> 1) & 0x5555555555555555ull) | ((v & 0x5555555555555555ull) << 1);
    v = ((v >> 2) & 0x3333333333333333ull) | ((v & 0x3333333333333333ull) << 2);
    v = ((v >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((v & 0x0F0F0F0F0F0F0F0Full) << 4);
    v = ((v >> 8) & 0x00FF00FF00FF00FFull) | ((v & 0x00FF00FF00FF00FFull) << 8);
    v = ((v >> 16) & 0x0000FFFF0000FFFFull) | ((v & 0x0000FFFF0000FFFFull) << 16);
    return (v >> 32) | (v << 32);
#endif
}
STP_HD int stp_ctz64(stp_u64 v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((long long)v) - 1;
#else
    return __builtin_ctzll(v);
#endif
}
// flood `seed` (subset of mask) along the runs of ones of `mask` inside one word
STP_HD stp_u64 stp_runfill(stp_u64 mask, stp_u64 seed)
{
    stp_u64 up = (mask & ~(mask + seed)) | seed;
    stp_u64 rm = stp_brev64(mask), rs = stp_brev64(up);
    stp_u64 dn = (rm & ~(rm + rs)) | rs;
    return stp_brev64(dn);
}

// ---- frame overlap (round 6) -------------------------------------------------------------------------------------------
// Frames advance by 200 bins and are 400 wide (getStripe.py:794-799): the trailing 200 x 200 block of frame f is the leading
// block of frame f + 1.  Blur, Gaussian, Sobel and the local-maximum test are functions of a pixel's neighbourhood alone
// wherever that neighbourhood lies inside both images and the two compacted frames keep the same bins there: the class of
// (r, c) in frame f is then the class of (r - shift, c - shift) in frame f + 1 -- the SAME operations on the SAME numbers,
// bit for bit, whatever the arithmetic.  How far from the block's border: the class of a pixel reads the magnitudes of its 3 x 3
// neighbourhood (1), a magnitude the smoothed values of its 3 x 3 neighbourhood (1), a smoothed value the grey values R rows
// and columns away -- all of them inside the image, or the window is cut and renormalised (R) -- and a grey value the contact
// values one pixel away, inside the image or the 3 x 3 mean reflects (1): R + 3.  (Hysteresis runs per frame, on the whole map.)
// The host marks a frame whose trailing kept bins are exactly the leading kept bins of its successor (shift = their first
// index in this frame; -1 otherwise); k_canny_f32 then skips the tiles that lie inside the square [lo, hi)^2 and k_lines' loader
// takes those class words from the successor's planes.
// ---- grey tiles nobody reads (round 6, with the image symmetry) ----------------------------------------------------------------
// k_canny_f32 with `mirror` computes the tiles (ty, tx) with ty <= 2 tx + 1 only.  A computed tile reads grey values at most
// R + 2 <= 14 pixels from its border and grey cells (8 x 16) that overlap that window: inside the tile grown by 16 rows and 16
// columns, i.e. inside the 3 x 3 tile neighbourhood.  A grey tile (gy, gx) therefore has a reader among the computed tiles iff
// gy - 1 <= 2 (gx + 1) + 1: the 20 tiles with gy >= 2 gx + 5 (a quarter of the image, its far corner below the diagonal) are
// written for nobody -- k_gray_c3 skips them.  The two consumers that may look there anyway: the mirrored resolver reads
// its grey values at the transposed position (c32_res_V_taps<R, true>), and a (frame, level) whose tiles below the diagonal go to
// the exact kernel (list overflow, or an image that differs from its transpose) is marked by k_canny_f32 and gets these tiles
// filled in by k_gray_fill -- every operation of the reference, at the position itself -- before k_canny_pipe_list runs.
STP_HD bool stp_gray_dead_tile(int gy, int gx) { return gy >= 2 * gx + 5; }
// (with the frame overlap the rule is applied tile by tile -- stp_gray_tile_unread below, behind the overlap's definitions: the
//  Canny tiles inside the block shared with the next frame are not computed either, which leaves six more grey tiles without a
//  reader in an ordinary frame)

// the mark of frame i (host and device run the same code): z0 / z1 = the kept local indices of frames i and i + 1
STP_HD int stp_overlap_shift(int s0, int n0, int S0, const int16_t* z0, int s1, int n1, int S1, const int16_t* z1)
{
    const int e0 = s0 + n0 - 1, e1 = s1 + n1 - 1;
    if (!(s1 > s0 && s1 <= e0 && e1 >= e0) || S0 <= 0 || S1 <= 0) return -1;
    int p = 0;
    while (p < S0 && s0 + z0[p] < s1) p++;
    const int q = S0 - p;
    if (q < 1 || q > S1) return -1;
    for (int k = 0; k < q; k++)
        if (s0 + z0[p + k] != s1 + z1[k]) return -1;
    if (q < S1 && s1 + z1[q] <= e0) return -1;                 // frame i + 1 keeps a bin of the block that frame i dropped
    return p;
}
#define STP_REUSE_MARGIN(R) ((R) + 3)
struct stp_reuse { int lo, hi, shift; };           // lo >= hi: nothing is taken from the next frame
STP_HD stp_reuse stp_reuse_of(int shift, int S, int R, bool next_in_launch)
{
    stp_reuse U;
    U.shift = shift;
    U.lo = shift + STP_REUSE_MARGIN(R);
    U.hi = (shift >= 0 && next_in_launch) ? S - STP_REUSE_MARGIN(R) : 0;
    if (shift < 0) U.lo = 1;
    return U;
}
// is tile (ty, tx) -- CT_Y rows x CT_X columns, the grid of the Canny kernels -- wholly inside the square
STP_HD bool stp_reuse_tile(stp_reuse U, int ty, int tx)
{
    return U.lo < U.hi && ty * 32 >= U.lo && ty * 32 + 32 <= U.hi && tx * 64 >= U.lo && tx * 64 + 64 <= U.hi;
}
// is Canny tile (ty, tx) of an S x S image computed by k_canny_f32 (not below the diagonal with `mirror`, not inside the shared block)
STP_HD bool stp_canny_tile_computed(int ty, int tx, int S, int mirror, stp_reuse U)
{
    return ty >= 0 && tx >= 0 && ty * 32 < S && tx * 64 < S && !(mirror && ty >= 2 * tx + 2) && !stp_reuse_tile(U, ty, tx);
}
// grey tile (gy, gx) has no reader among the computed Canny tiles: none in its 3 x 3 tile neighbourhood (see "grey tiles nobody
// reads": a Canny tile reads grey values and cells at most 16 pixels beyond its border).  Without the overlap this is
// stp_gray_dead_tile (gy >= 2 gx + 5); k_gray_c3 does not write such a tile, k_gray_fill fills it in on demand.
// Only tiles on or below the line gy = 2 gx + 1 are ever skipped (42 of the 91): that is where every unread tile of the
// reference's frame geometry lies -- the 20 of the symmetry rule and those around the shared block -- and it is the set
// k_gray_fill's grid walks, so "skipped by k_gray_c3" and "filled in on demand" are the same predicate by construction.
STP_HD bool stp_gray_tile_unread(int gy, int gx, int S, int mirror, stp_reuse U)
{
    if (gy < 2 * gx + 1) return false;
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++)
            if (stp_canny_tile_computed(gy + dy, gx + dx, S, mirror, U)) return false;
    return true;
}
// the columns of word w of row r that no tile of this frame writes.  With `mirror` (k_canny_f32, image symmetry) a tile
// strictly below the diagonal (tile row >= 2 x word + 2) receives its two half words from the tiles (2 w + h, r / 64) above
STP_HD stp_u64 stp_reuse_mask(stp_reuse U, int mirror, int r, int w)
{
    if (U.lo >= U.hi) return 0ull;
    const int ty = r >> 5;
    if (mirror && ty >= 2 * w + 2)
        return (stp_reuse_tile(U, 2 * w, r >> 6) ? 0xFFFFFFFFull : 0ull) | (stp_reuse_tile(U, 2 * w + 1, r >> 6) ? 0xFFFFFFFF00000000ull : 0ull);
    return stp_reuse_tile(U, ty, w) ? ~0ull : 0ull;
}
// bits c0 .. c0 + 63 of row r of a class plane (word-column-major); columns outside the plane read as 0
STP_HD stp_u64 stp_cls_bits(const stp_u64* __restrict__ plane, int r, int c0)
{
    const int w0 = c0 >> 6, sh = c0 & 63;           // (arithmetic shift: floor for a negative column)
    const stp_u64 a = (w0 >= 0 && w0 < STP_NW) ? plane[STP_CLS(r, w0)] : 0ull;
    if (sh == 0) return a;
    const stp_u64 b = (w0 + 1 >= 0 && w0 + 1 < STP_NW) ? plane[STP_CLS(r, w0 + 1)] : 0ull;
    return (a >> sh) | (b << (64 - sh));
}
// k_lines' loader with the overlap: a thread first REQUESTS the words it will patch from the next frame's planes (at most
// STP_REUSE_ITEMS (row, word) items of the square's bounding box), then runs the plain load -- all global loads are in flight
// together: as a second pass behind the plain load the patch cost every image one more memory round trip (+1.5 ms per genome
// step) -- and after a barrier writes the requested bits over whatever the skipped tiles' words held.
#define STP_REUSE_ITEMS 3
struct stp_reuse_fetch { stp_u64 m[STP_REUSE_ITEMS], lo[STP_REUSE_ITEMS], hi[STP_REUSE_ITEMS]; int k[STP_REUSE_ITEMS]; };
STP_HD void lines_reuse_request(int tid, int nt, int S, stp_reuse U, int mirror, const stp_u64* __restrict__ low_next,
                                const stp_u64* __restrict__ high_next, stp_reuse_fetch* F)
{
    const int ty0 = (U.lo + 31) >> 5, ty1 = U.hi >> 5;                // tile rows / word columns wholly inside the square
    const int tx0 = (U.lo + 63) >> 6, tx1 = U.hi >> 6;
    const int r0 = ty0 << 5, nr = (ty1 - ty0) << 5;
    // words: the skipped tiles' own (tx0 .. tx1) and, with `mirror`, the words that hold their transposes' half words
    const int w0 = mirror && (ty0 >> 1) < tx0 ? (ty0 >> 1) : tx0, w1 = mirror && ((ty1 + 1) >> 1) > tx1 ? ((ty1 + 1) >> 1) : tx1;
    const int nwu = (S + 63) >> 6, wl = w0 < 0 ? 0 : w0, wh = w1 > nwu ? nwu : w1, nw = wh - wl;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < STP_REUSE_ITEMS; q++) {
        const int i = tid + q * nt;
        F->m[q] = 0ull; F->k[q] = 0;
        if (nr > 0 && nw > 0 && i < nr * nw) {
            const int w = wl + i / nr, r = r0 + i % nr;
            const stp_u64 m = stp_reuse_mask(U, mirror, r, w);
            if (m) {
                F->m[q] = m; F->k[q] = r * STP_NW + w;
                F->lo[q] = stp_cls_bits(low_next, r - U.shift, 64 * w - U.shift);
                F->hi[q] = stp_cls_bits(high_next, r - U.shift, 64 * w - U.shift);
            }
        }
    }
}
STP_HD void lines_reuse_patch(const stp_reuse_fetch* F, stp_u64* sLow, stp_u64* sE)
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < STP_REUSE_ITEMS; q++)
        if (F->m[q]) {
            const int k = F->k[q];
            sLow[k] = (sLow[k] & ~F->m[q]) | (F->lo[q] & F->m[q]);
            sE[k] = (sE[k] & ~F->m[q]) | (F->hi[q] & F->m[q]);
        }
}
STP_HD void lines_load(int tid, int nt, int S, const stp_u64* __restrict__ low_img,
                       const stp_u64* __restrict__ high_img, stp_u64* sLow, stp_u64* sE)
{
    const int nwu = (S + 63) >> 6;
    for (int i = tid; i < S * STP_NW; i += nt) {         // consecutive lanes: consecutive rows of one word column (STP_CLS)
        const int w = i / S, r = i - w * S;
        stp_u64 lo = 0, hi = 0;
        if (w < nwu) { lo = low_img[STP_CLS(r, w)]; hi = high_img[STP_CLS(r, w)]; }
        sLow[r * STP_NW + w] = lo; sE[r * STP_NW + w] = hi;
    }
}

// one in-place dilation sweep of the hysteresis closure (_canny.py:286-296: 8-connected
// components of low that contain a high pixel).  Returns 1 when this thread changed a word.
STP_HD int lines_hyst_sweep(int tid, int nt, int S, const stp_u64* sLow, stp_u64* sE)
{
    int changed = 0;
    for (int i = tid; i < S * STP_NW; i += nt) {
        stp_u64 lowv = sLow[i];
        if (!lowv) continue;
        int r = i / STP_NW, w = i - r * STP_NW;
        stp_u64 cur = sE[i], n = 0;
        for (int dr = -1; dr <= 1; dr++) {
            int rr = r + dr;
            if (rr < 0 || rr >= S) continue;
            const stp_u64* row = sE + rr * STP_NW;
            n |= row[w] | bm_shl1(row, w) | bm_shr1(row, w);
        }
        stp_u64 seed = n & lowv;
        if (!seed) continue;
        stp_u64 nv = stp_runfill(lowv, seed) | cur;
        if (nv != cur) { sE[i] = nv; changed = 1; }
    }
    return changed;
}

// Strip form of the sweep: one item = one word-column x 8 consecutive rows, walked downwards and then
// upwards in place, so a sweep carries a connection across >= 8 rows (vertical edges -- the stripes
// themselves -- needed one sweep per row with the plain form).  Same fix-point (the closure is unique).
#define STP_HYST_STRIP 8
STP_HD stp_u64 lines_hyst_row(int S, int r, int w, const stp_u64* sLow, stp_u64* sE, int* changed)
{
    const stp_u64 lowv = sLow[r * STP_NW + w];
    const stp_u64 cur = sE[r * STP_NW + w];
    if (!lowv) return cur;
    stp_u64 n = 0;
    for (int dr = -1; dr <= 1; dr++) {
        const int rr = r + dr;
        if (rr < 0 || rr >= S) continue;
        const stp_u64* row = sE + rr * STP_NW;
        n |= row[w] | bm_shl1(row, w) | bm_shr1(row, w);
    }
    const stp_u64 seed = n & lowv;
    if (!seed) return cur;
    const stp_u64 nv = stp_runfill(lowv, seed) | cur;
    if (nv != cur) { sE[r * STP_NW + w] = nv; *changed = 1; }
    return nv;
}
STP_HD int lines_hyst_sweep_strip(int tid, int nt, int S, const stp_u64* sLow, stp_u64* sE)
{
    int changed = 0;
    const int nstrip = (S + STP_HYST_STRIP - 1) / STP_HYST_STRIP;
    for (int i = tid; i < nstrip * STP_NW; i += nt) {
        const int st = i / STP_NW, w = i - st * STP_NW;
        const int r0 = st * STP_HYST_STRIP, r1 = (r0 + STP_HYST_STRIP < S) ? r0 + STP_HYST_STRIP : S;
        for (int r = r0; r < r1; r++) lines_hyst_row(S, r, w, sLow, sE, &changed);
        for (int r = r1 - 2; r >= r0; r--) lines_hyst_row(S, r, w, sLow, sE, &changed);
    }
    return changed;
}

// Register form of the strip sweep (the one k_lines runs): a lane owns one item (8 rows x 1 word) for the
// whole closure and keeps its E rows in registers (the low rows are read from LDS at their step).  A sweep first reads what it needs from the
// other items -- the rows above / below the strip and the edge bits of the words left / right of it, all
// loads issued up front -- then walks down and up entirely in registers (no LDS round trip per row) and
// writes its rows back only if they grew.  Words other lanes are rewriting may be read old or new: E
// only grows, so either is a valid lower bound, and a sweep in which nobody changed has seen final values.
struct stp_hyst_item {
    stp_u64 E[STP_HYST_STRIP];
    int item;                      // strip * STP_NW + word (a sweep derives its rows and its word from it: one register, not two)
};
STP_HD stp_u64 stp_dil1(stp_u64 x) { return x | (x << 1) | (x >> 1); }
// Edge bits of an item (round 5): bit k of the low byte = bit 0 of row k, bit k of the high byte = bit 63 of row k.  Every item
// publishes them in sEdge[item] (16 bits; slot `nitem` stays 0 and stands in for neighbours beyond the image), so a sweep takes
// the carry-in bits of its ten rows from SIX halfwords -- the items left and right of it and of the strips above and below --
// instead of twenty bounds-tested word reads (which were 40 % of the instructions of a sweep that grows nothing).  Like the
// words themselves the halfwords only gain bits, so a reader may see the old or the new value.
STP_HD unsigned hyst_edge_bits(const stp_u64* E)
{
    unsigned l = 0, r = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < STP_HYST_STRIP; k++) {
        l |= ((unsigned)E[k] & 1u) << k;
        r |= (unsigned)(E[k] >> 63) << k;
    }
    return l | r << 8;
}
STP_HD void hyst_item_load(int item, int S, const stp_u64* sLow, const stp_u64* sE, stp_hyst_item* it, uint16_t* sEdge)
{
    const int st = item / STP_NW, w = item - st * STP_NW, r0 = st * STP_HYST_STRIP;
    it->item = item;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < STP_HYST_STRIP; k++) {
        const int r = r0 + k;
        it->E[k] = (r < S) ? sE[r * STP_NW + w] : 0ull;
    }
    sEdge[item] = (uint16_t)hyst_edge_bits(it->E);
}
STP_HD int hyst_item_sweep(int S, stp_hyst_item* it, const stp_u64* sLow, stp_u64* sE, uint16_t* sEdge)
{
    constexpr int N = STP_HYST_STRIP;
    const int item = it->item, st = item / STP_NW, w = item - st * STP_NW, r0 = st * N;
    // carry-in bits of rows r0-1 .. r0+N from the neighbouring words: bit 0 <- left word's bit 63,
    // bit 63 <- right word's bit 0 (bm_shl1 / bm_shr1); bit k of cl / cr belongs to row r0-1+k
    unsigned cl, cr;
    {
        const int nstrip = (S + N - 1) / N, nitem = nstrip * STP_NW;
        const bool hl = w > 0, hr = w < STP_NW - 1, hu = r0 > 0, hd = r0 + N < S;
        const unsigned eL = sEdge[hl ? item - 1 : nitem], eR = sEdge[hr ? item + 1 : nitem];
        const unsigned eUL = sEdge[hl && hu ? item - STP_NW - 1 : nitem], eUR = sEdge[hr && hu ? item - STP_NW + 1 : nitem];
        const unsigned eDL = sEdge[hl && hd ? item + STP_NW - 1 : nitem], eDR = sEdge[hr && hd ? item + STP_NW + 1 : nitem];
        cl = (eUL >> 15) | (eL >> 8) << 1 | ((eDL >> 8) & 1u) << (N + 1);
        cr = ((eUR >> 7) & 1u) | (eR & 0xFFu) << 1 | (eDR & 1u) << (N + 1);
    }
#define STP_HYST_CIN(k) ((stp_u64)((cl >> (k)) & 1u) | ((stp_u64)((cr >> (k)) & 1u) << 63))
#define STP_HYST_D(k) (stp_dil1(it->E[(k) - 1]) | STP_HYST_CIN(k))      /* dilated own row k-1 (1 <= k <= N) */
    const stp_u64 up = (r0 - 1 >= 0) ? sE[(r0 - 1) * STP_NW + w] : 0ull;
    const stp_u64 dn = (r0 + N < S) ? sE[(r0 + N) * STP_NW + w] : 0ull;
    const stp_u64 Dtop = stp_dil1(up) | STP_HYST_CIN(0), Dbot = stp_dil1(dn) | STP_HYST_CIN(N + 1);
    // a three-row window of dilated rows slides down and then up (no per-row array: the register budget of
    // k_lines is 80 VGPRs, and anything spilled here would put scratch-memory latency into every step)
    stp_u64 grown = 0;
    stp_u64 above = Dtop, cur = STP_HYST_D(1);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < N; k++) {                     // downwards: row k sees the already updated row k-1
        const stp_u64 below = (k + 1 < N) ? STP_HYST_D(k + 2) : Dbot;
        const stp_u64 lowv = (r0 + k < S) ? sLow[(r0 + k) * STP_NW + w] : 0ull, e = it->E[k];   // low stays in LDS
        const stp_u64 seed = (above | cur | below) & lowv & ~e;
        if (seed) {
            it->E[k] = stp_runfill(lowv, seed) | e;
            cur = STP_HYST_D(k + 1);
            grown |= 1ull << k;
        }
        above = cur; cur = below;
    }
    // A downward pass that grew nothing has tested every row against its final neighbours: the upward pass would test the
    // same words again.  (round 5: the late sweeps of an image -- five on average -- grow a handful of items; whole waves
    //  take this exit.  Wave-uniform on the device, so the lanes of a wave stay together.)
#if defined(__HIP_DEVICE_COMPILE__)
    if (__ballot(grown != 0) == 0ull) return 0;
#else
    if (!grown) return 0;
#endif
    stp_u64 below2 = STP_HYST_D(N);                   // row N-1 as left by the downward pass
    cur = (N >= 2) ? STP_HYST_D(N - 1) : 0ull;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = N - 2; k >= 0; k--) {                // upwards: row k sees the already updated row k+1
        const stp_u64 abv = (k >= 1) ? STP_HYST_D(k) : Dtop;
        const stp_u64 lowv = (r0 + k < S) ? sLow[(r0 + k) * STP_NW + w] : 0ull, e = it->E[k];
        const stp_u64 seed = (abv | cur | below2) & lowv & ~e;
        if (seed) {
            it->E[k] = stp_runfill(lowv, seed) | e;
            cur = STP_HYST_D(k + 1);
            grown |= 1ull << k;
        }
        below2 = cur; cur = abv;
    }
    if (!grown) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < N; k++)
        if ((grown >> k) & 1ull) sE[(r0 + k) * STP_NW + w] = it->E[k];
    sEdge[item] = (uint16_t)hyst_edge_bits(it->E);
    return 1;
#undef STP_HYST_D
#undef STP_HYST_CIN
}

STP_HD void stp_fa(stp_u64 x, stp_u64 y, stp_u64 c, stp_u64* s, stp_u64* co)
{
    stp_u64 t = x ^ y;
    *s = t ^ c;
    *co = (x & y) | (c & t);
}
// bit-sliced n = p + 2*q + r  (0..4) -> bits n0,n1,n2
STP_HD void stp_w121(stp_u64 p, stp_u64 q, stp_u64 r, stp_u64* n0, stp_u64* n1, stp_u64* n2)
{
    stp_u64 cy = p & r;
    *n0 = p ^ r;
    *n1 = cy ^ q;
    *n2 = cy & q;
}
// bit-sliced d = a - b + 8 for 3-bit a, b (0..4): a + (~b & 7) + 1 -> 4 bits
STP_HD void stp_diff8(stp_u64 a0, stp_u64 a1, stp_u64 a2, stp_u64 b0, stp_u64 b1, stp_u64 b2, stp_u64* d0,
                      stp_u64* d1, stp_u64* d2, stp_u64* d3)
{
    stp_u64 c;
    stp_fa(a0, ~b0, ~0ull, d0, &c);
    stp_fa(a1, ~b1, c, d1, &c);
    stp_fa(a2, ~b2, c, d2, &c);
    *d3 = c;
}

// ImageProcessing.verticalLine(edges, 60, 120) (ImageProcessing.py:61-83), exact integer form:
// hit(i,j) <=> Fx > 0 and 3*Fy^2 < Fx^2 with Fx = (1,2,1)^T.(E[:,j-1]-E[:,j+1]),
// Fy = (1,2,1).(E[i+1,:]-E[i-1,:]); the hit is stored at column j-1 (:78).  Column 0's hit would
// wrap to S-1 but needs Fx>0 with an empty left column, which is impossible, so V[:,S-1] = 0.
STP_HD void lines_vline(int tid, int nt, int S, const stp_u64* sE, stp_u64* sV)
{
    for (int i = tid; i < S * STP_NW; i += nt) {
        int r = i / STP_NW, w = i - r * STP_NW;
        stp_u64 X[3][3];
        for (int dr = 0; dr < 3; dr++) {
            int rr = r + dr - 1;
            if (rr < 0 || rr >= S) { X[dr][0] = X[dr][1] = X[dr][2] = 0; continue; }
            const stp_u64* row = sE + rr * STP_NW;
            X[dr][0] = row[w]; X[dr][1] = bm_shr1(row, w); X[dr][2] = bm_shr2(row, w);
        }
        stp_u64 L0, L1, L2, R0, R1, R2, T0, T1, T2, B0, B1, B2;
        stp_w121(X[0][0], X[1][0], X[2][0], &L0, &L1, &L2);   // left column  a + 2d + g
        stp_w121(X[0][2], X[1][2], X[2][2], &R0, &R1, &R2);   // right column c + 2f + i
        stp_w121(X[0][0], X[0][1], X[0][2], &T0, &T1, &T2);   // top row     a + 2b + c
        stp_w121(X[2][0], X[2][1], X[2][2], &B0, &B1, &B2);   // bottom row  g + 2h + i
        stp_u64 x0, x1, x2, x3, y0, y1, y2, y3;
        stp_diff8(L0, L1, L2, R0, R1, R2, &x0, &x1, &x2, &x3);  // Fx + 8 in [4, 12]
        stp_diff8(B0, B1, B2, T0, T1, T2, &y0, &y1, &y2, &y3);  // Fy + 8
        stp_u64 fx_ge1 = x3 & (x0 | x1 | x2);
        stp_u64 fx_ge2 = x3 & (x1 | x2);
        stp_u64 fx_eq4 = x3 & x2;
        stp_u64 fy_eq0 = y3 & ~y2 & ~y1 & ~y0;
        stp_u64 fy_le1 = (y3 & ~y2 & ~y1) | (~y3 & y2 & y1 & y0);
        stp_u64 hit = (fx_ge1 & fy_eq0) | (fx_ge2 & fy_le1) | fx_eq4;
        // valid output columns j' = 64w + bit with j' + 1 <= S - 1
        int lim = S - 1 - 64 * w;                       // number of valid bits in this word
        stp_u64 vm = lim <= 0 ? 0ull : (lim >= 64 ? ~0ull : ((1ull << lim) - 1ull));
        sV[i] = hit & vm;
    }
}

// ImageProcessing.block (ImageProcessing.py:124-195) + the caller's keep test (getStripe.py:929-940).
// value[i] = V[i][c-1] | V[i][c] | V[i][c+1] (ImageProcessing.py:122-123) is formed on the fly from the
// row word of V (plus the neighbouring word for the two lanes at a word edge).
STP_HD void lines_block(int tid, int nt, int S, int minH, const stp_u64* sV, int16_t* colT, int16_t* colEnd, int16_t* colUd)
{
    for (int c = tid; c < S; c += nt) {
        int count = 0, MAX = 0, END = 0, J = 0, buffer = 0;
        const int wi = c >> 6, b = c & 63;
        const stp_u64 bit = 1ull << b;
        const stp_u64 m3 = (b == 0) ? 3ull : ((b == 63) ? (3ull << 62) : (7ull << (b - 1)));   // own-word part of c-1..c+1
        const int nbw = (b == 0 && wi > 0) ? wi - 1 : ((b == 63 && wi + 1 < STP_NW) ? wi + 1 : -1);
        const stp_u64 nbm = (b == 0) ? (1ull << 63) : 1ull;                                       // bit of the neighbour word
        for (int i0 = 0; i0 < S; i0 += 8) {          // 8 independent LDS reads in flight per step
            stp_u64 wd[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                stp_u64 v = 0;
                if (i0 + k < S) {
                    v = sV[(i0 + k) * STP_NW + wi] & m3;
                    if (nbw >= 0) v |= sV[(i0 + k) * STP_NW + nbw] & nbm;
                }
                wd[k] = v;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = i0 + k;
                if (i >= S) break;
                if (wd[k]) { count++; J = i; }
                else if (buffer < 5) buffer++;
                else {
                    if (count > MAX) { MAX = count; END = J; }
                    count = 0; buffer = 0;
                }
            }
        }
        if (count > MAX) { MAX = count; END = J; }
        int t = MAX;
        if (END < c) END = END - t + 1;
        int above = c < END ? c : END, bottom = c > END ? c : END;
        int any = 0;
        if (above < 0) above = 0;
        if (bottom > S - 1) bottom = S - 1;
        for (int y = above; y <= bottom; y++) any |= (sV[y * STP_NW + wi] & bit) != 0;
        colT[c] = (int16_t)t; colEnd[c] = (int16_t)END;
        colUd[c] = (int16_t)((t > minH && any) ? (END > c ? 2 : 1) : 0);
    }
}

// helpers on one bit-row
STP_HD int row_next_set(const stp_u64* row, int from, int S)
{
    if (from >= S) return S;
    int w = from >> 6;
    stp_u64 v = row[w] & (~0ull << (from & 63));
    while (true) {
        if (v) { int x = (w << 6) + stp_ctz64(v); return x < S ? x : S; }
        if (++w >= STP_NW) return S;
        v = row[w];
    }
}
STP_HD int row_next_clear(const stp_u64* row, int from, int S)
{
    if (from >= S) return S;
    int w = from >> 6;
    stp_u64 v = ~row[w] & (~0ull << (from & 63));
    while (true) {
        if (v) { int x = (w << 6) + stp_ctz64(v); return x < S ? x : S; }
        if (++w >= STP_NW) return S;
        v = ~row[w];
    }
}
STP_HD stp_u64 word_range_mask(int w, int lo, int hi)   // bits of word w inside [lo, hi]
{
    int a = lo - 64 * w, b = hi - 64 * w;
    if (b < 0 || a > 63) return 0ull;
    if (a < 0) a = 0;
    if (b > 63) b = 63;
    stp_u64 m = (b == 63) ? ~0ull : ((1ull << (b + 1)) - 1ull);
    return m & (~0ull << a);
}

// ---- column-word forms ------------------------------------------------------------------------
// A column of a bit matrix as 7 words (bit i of the 448-bit vector = row i; rows >= S are 0).  With the
// column in registers, `block`'s per-row state machine advances run by run instead of row by row:
//   a run of k ones : count += k, J = its last row;
//   a run of z zeros: the first 5-buffer of them only fill the gap buffer; one more closes the run
//                     (count > MAX -> MAX = count, END = J; count = buffer = 0); after that every 6th
//                     zero closes again, which is a no-op because count is 0, so buffer = rest % 6.
// The words are walked in order with compile-time indices (a run-time index into a 7-word register array costs
// a 14-select chain per access): bit j of t = x ^ ((x << 1) | carry) marks a row whose value differs from the row
// above, i.e. the end of the run that started at `prev`.
STP_HD void col_block_scan(const stp_u64* v3, const stp_u64* v, int c, int S, int minH, int16_t* t_out, int16_t* end_out,
                           int16_t* ud_out)
{
    int count = 0, MAX = 0, END = 0, J = 0, buffer = 0;
    int prev = 0;
    stp_u64 bit = v3[0] & 1ull;                 // value of the run that starts at row 0
    stp_u64 carry = bit;
#define STP_RUN_END(p) do { \
        const int len_ = (p) - prev; \
        if (bit) { count += len_; J = (p) - 1; } \
        else if (len_ <= 5 - buffer) buffer += len_; \
        else { \
            if (count > MAX) { MAX = count; END = J; } \
            count = 0; \
            buffer = (len_ - (5 - buffer) - 1) % 6; \
        } \
        prev = (p); bit ^= 1ull; } while (0)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < STP_NW; k++) {
        const stp_u64 x = v3[k];
        stp_u64 t = x ^ ((x << 1) | carry);
        const int left = S - 64 * k;            // rows of this word that exist
        if (left <= 0) t = 0ull;
        else if (left < 64) t &= (1ull << left) - 1ull;
        while (t) {
            const int p = 64 * k + stp_ctz64(t);
            t &= t - 1ull;
            STP_RUN_END(p);
        }
        carry = x >> 63;
    }
    if (S > prev) STP_RUN_END(S);
#undef STP_RUN_END
    if (count > MAX) { MAX = count; END = J; }
    const int t = MAX;
    if (END < c) END = END - t + 1;
    int above = c < END ? c : END, bottom = c > END ? c : END;
    if (above < 0) above = 0;
    if (bottom > S - 1) bottom = S - 1;
    int any = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int w = 0; w < STP_NW; w++) any |= (v[w] & word_range_mask(w, above, bottom)) != 0;
    *t_out = (int16_t)t; *end_out = (int16_t)END;
    *ud_out = (int16_t)((t > minH && any) ? (END > c ? 2 : 1) : 0);
}
STP_HD void col_stat(const stp_u64* tc, int S, int16_t* cnt, int16_t* minr, int16_t* maxr)
{
    int n = 0, mn = S, mx = -1;
    for (int w = 0; w < STP_NW; w++) {
        const stp_u64 x = tc[w];
        if (!x) continue;
#if defined(__HIP_DEVICE_COMPILE__)
        n += __popcll(x);
        if (mn == S) mn = (w << 6) + (__ffsll((long long)x) - 1);
        mx = (w << 6) + 63 - __clzll((long long)x);
#else
        n += __builtin_popcountll(x);
        if (mn == S) mn = (w << 6) + __builtin_ctzll(x);
        mx = (w << 6) + 63 - __builtin_clzll(x);
#endif
    }
    *cnt = (int16_t)n; *minr = (int16_t)mn; *maxr = (int16_t)mx;
}
// CPU-replay forms: gather the column bit by bit (the device transposes 64x64 blocks with wave shuffles)
STP_HD void col_gather(const stp_u64* m, int S, int c, stp_u64* out)
{
    for (int w = 0; w < STP_NW; w++) out[w] = 0;
    for (int r = 0; r < S; r++)
        if ((m[r * STP_NW + (c >> 6)] >> (c & 63)) & 1ull) out[r >> 6] |= 1ull << (r & 63);
}
STP_HD void lines_block_cols(int tid, int nt, int S, int minH, const stp_u64* sV, const stp_u64* sV3, int16_t* colT,
                             int16_t* colEnd, int16_t* colUd)
{
    for (int c = tid; c < S; c += nt) {
        stp_u64 v[STP_NW], v3[STP_NW];
        col_gather(sV, S, c, v);
        col_gather(sV3, S, c, v3);
        col_block_scan(v3, v, c, S, minH, &colT[c], &colEnd[c], &colUd[c]);
    }
}
STP_HD void lines_colstat_cols(int tid, int nt, int S, const stp_u64* sT, int16_t* cnt, int16_t* minr, int16_t* maxr)
{
    for (int c = tid; c < S; c += nt) {
        stp_u64 tc[STP_NW];
        col_gather(sT, S, c, tc);
        col_stat(tc, S, &cnt[c], &minr[c], &maxr[c]);
    }
}
// V3[r][c] = V[r][c-1] | V[r][c] | V[r][c+1]  (ImageProcessing.py:122-123)
STP_HD void lines_v3(int tid, int nt, int S, const stp_u64* sV, stp_u64* sV3)
{
    for (int i = tid; i < S * STP_NW; i += nt) {
        int r = i / STP_NW, w = i - r * STP_NW;
        const stp_u64* row = sV + r * STP_NW;
        sV3[i] = row[w] | bm_shl1(row, w) | bm_shr1(row, w);
    }
}

STP_HD void lines_zero(int tid, int nt, int n, stp_u64* m)
{
    for (int i = tid; i < n; i += nt) m[i] = 0;
}

// getStripe.py:948-955: column c painted on rows [st, en) (thread-per-column; needs LDS atomics
// on the device because columns of one word belong to different threads)
#if defined(__HIP_DEVICE_COMPILE__)
#define STP_ATOMIC_OR(p, v) atomicOr((p), (v))
#else
#define STP_ATOMIC_OR(p, v) (*(p) |= (v))
#endif
STP_HD void lines_paint(int tid, int nt, int S, int ud, const int16_t* colEnd, const int16_t* colUd, stp_u64* sT)
{
    for (int c = tid; c < S; c += nt) {
        if (colUd[c] != ud) continue;
        int st = c, en = colEnd[c];
        if (ud == 1) { int t = st; st = en; en = t; }
        if (st < 0) st = 0;
        if (en > S) en = S;
        const stp_u64 bit = 1ull << (c & 63);
        for (int y = st; y < en; y++) STP_ATOMIC_OR(&sT[y * STP_NW + (c >> 6)], bit);
    }
}

// getStripe.py:957-978 line refinement, thread per row
STP_HD void lines_refine(int tid, int nt, int S, const stp_u64* sE, const stp_u64* sV, stp_u64* sT)
{
    for (int r = tid; r < S; r += nt) {
        stp_u64* trow = sT + r * STP_NW;
        const stp_u64* erow = sE + r * STP_NW;
        const stp_u64* vrow = sV + r * STP_NW;
        // runs are determined from the row as painted (st/en lists are built before the L loop).  The live
        // row serves: a run's rewrite below only touches bits inside [st, en] and the scan resumes at en + 1
        int x = 0;
        while (true) {
            int st = row_next_set(trow, x, S);
            if (st >= S) break;
            int en = row_next_clear(trow, st, S) - 1;
            int any = 0;
            for (int w = st >> 6; w <= en >> 6; w++) any |= (erow[w] & word_range_mask(w, st, en)) != 0;
            if (any) {          // testmat[r, st:en] = vert[r, st:en]   (column en untouched)
                for (int w = st >> 6; w <= (en >> 6); w++) {
                    stp_u64 m = word_range_mask(w, st, en - 1);
                    trow[w] = (trow[w] & ~m) | (vrow[w] & m);
                }
            } else {            // clear [st, en), set MED = round-half-even((st+en)/2)
                for (int w = st >> 6; w <= (en >> 6); w++) trow[w] &= ~word_range_mask(w, st, en - 1);
                int s2 = st + en, k = s2 >> 1;
                int med = (s2 & 1) ? ((k & 1) ? k + 1 : k) : k;
                trow[med >> 6] |= 1ull << (med & 63);
            }
            x = en + 1;
        }
    }
}

// per-column pixel count and first / last row (getStripe.py:981-984, 1060-1065)
STP_HD void lines_colstat(int tid, int nt, int S, const stp_u64* sT, int16_t* cnt, int16_t* minr, int16_t* maxr)
{
    for (int c = tid; c < S; c += nt) {
        const int wi = c >> 6;
        const stp_u64 bit = 1ull << (c & 63);
        int n = 0, mn = S, mx = -1;
        for (int y0 = 0; y0 < S; y0 += 8) {
            stp_u64 wd[8];
#pragma unroll
            for (int k = 0; k < 8; k++) wd[k] = (y0 + k < S) ? sT[(y0 + k) * STP_NW + wi] : 0ull;
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (wd[k] & bit) { const int y = y0 + k; n++; if (mn == S) mn = y; mx = y; }
        }
        cnt[c] = (int16_t)n; minr[c] = (int16_t)mn; maxr[c] = (int16_t)mx;
    }
}

// Ordered compaction of the columns with >= 3 pixels (getStripe.py:994) in two phases around a
// barrier: per-wave counts, then positions.  wcnt: 8 ints.  On the CPU replay (nt == 1) phase 1 does
// nothing and phase 2 is the plain loop.
STP_HD void lines_cols_count(int tid, int nt, int S, const int16_t* cnt, int* wcnt)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const bool f = (tid < S) && (cnt[tid] >= 3);
    const stp_u64 m = __ballot(f);
    if ((tid & 63) == 0 && (tid >> 6) < 8) wcnt[tid >> 6] = __popcll(m);
#else
    (void)tid; (void)nt; (void)S; (void)cnt; (void)wcnt;
#endif
}
STP_HD void lines_cols_place(int tid, int nt, int S, const int16_t* cnt, const int* wcnt, int16_t* cidx, int16_t* clen,
                             int* nrow_out)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const bool f = (tid < S) && (cnt[tid] >= 3);
    const stp_u64 m = __ballot(f);
    const int wv = tid >> 6, lane = tid & 63;
    int base = 0, total = 0;
    for (int i = 0; i < 8; i++) { if (i < wv) base += wcnt[i]; total += wcnt[i]; }
    if (f) {
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        cidx[pos] = (int16_t)tid; clen[pos] = cnt[tid];
    }
    if (tid == 0) *nrow_out = total;
#else
    (void)nt; (void)wcnt;
    if (tid == 0) {
        int nrow = 0;
        for (int c = 0; c < S; c++)
            if (cnt[c] >= 3) { cidx[nrow] = (int16_t)c; clen[nrow] = cnt[c]; nrow++; }
        *nrow_out = nrow;
    }
#endif
}

struct stp_lrec { int16_t ud, x, y, w, h; };

// getStripe.py:994-1078 for one ud: column grouping (incl. the stale-[Current] behaviour of the
// loop at :1019-1028), X = sorted(set(meanX)), neighbour pairing.  Executed by ONE thread.
// cidx/clen[0..nrow): the columns with >= 3 pixels and their counts (lines_cols_place); scratch xs[S+2].
// Returns the new record count.
STP_HD int lines_group_pairs(int S, int ud, int maxW, int nrow, const int16_t* minr, const int16_t* maxr,
                             const int16_t* cidx, const int16_t* clen, int16_t* xs, stp_lrec* recs, int nrec, int cap)
{
    // meanX values are integers in [0, S): mark them in a bitmap (set semantics + sorted order)
    stp_u64 seen[STP_NW];
    for (int w = 0; w < STP_NW; w++) seen[w] = 0;
    int g0 = 0, gn = 0;          // Continuous = cidx[g0 .. g0+gn) or the single stale entry
    bool isContinue = false;
#define STP_FLUSH() do { \
        long ssum = 0; for (int q = 0; q < gn; q++) ssum += clen[g0 + q]; \
        double temp = 0.0; \
        for (int q = 0; q < gn; q++) temp = temp + (double)cidx[g0 + q] * ((double)clen[g0 + q] / (double)ssum); \
        int mv = (int)nearbyint(temp); \
        if (mv >= 0 && mv < S) seen[mv >> 6] |= 1ull << (mv & 63); } while (0)
    for (int c = 0; c + 1 < nrow; c++) {
        int Current = cidx[c], Next = cidx[c + 1];
        if (Next - Current == 1 && isContinue) {
            gn++;                                      // group stays contiguous in cidx
        } else if (Next - Current == 1 && !isContinue) {
            g0 = c; gn = 2; isContinue = true;
        } else if (Next - Current != 1 && !isContinue) {
            g0 = c; gn = 1; isContinue = false;
            STP_FLUSH();
        } else {
            STP_FLUSH();
            g0 = c; gn = 1; isContinue = false;        // stale [Current]: last column of the flushed group
        }
    }
    if (gn == 0) seen[0] |= 1ull;                       // np.round(sum([])) = 0
    else STP_FLUSH();
#undef STP_FLUSH
    int nx = 0;
    for (int w = 0; w < STP_NW; w++) {
        stp_u64 v = seen[w];
        while (v) { int b = stp_ctz64(v); v &= v - 1; xs[nx++] = (int16_t)((w << 6) + b); }
    }
    for (int c = 0; c + 1 < nx; c++) {
        int n = xs[c], m = xs[c + 1];
        int gap = m - n;
        if (gap > 1 && gap <= maxW) {
            int p1 = gap > 4 ? m - 2 : m;
            int MIN = S, MAX = -1;
            for (int q = 0; q < 2; q++) {
                int ctr = q ? m : n;
                int lo = ctr - 1 < 0 ? 0 : ctr - 1, hi = ctr + 2 > S ? S : ctr + 2;
                for (int xx = lo; xx < hi; xx++) {
                    if (minr[xx] < MIN) MIN = minr[xx];
                    if (maxr[xx] > MAX) MAX = maxr[xx];
                }
            }
            if (ud == 1) MAX = p1; else MIN = n;
            if (nrec < cap) {
                recs[nrec].ud = (int16_t)ud; recs[nrec].x = (int16_t)n; recs[nrec].y = (int16_t)MIN;
                recs[nrec].w = (int16_t)(p1 - n + 1); recs[nrec].h = (int16_t)(MAX - MIN + 1);
            }
            nrec++;
        }
    }
    return nrec;
}

// row sums of submat[y:y+h, x:x+w] (numpy adds the w (<8) elements of a row sequentially and
// then the rows in order, getStripe.py:1094)
STP_HD void lines_rowsum(int tid, int nt, int S, const double* __restrict__ band, int W, int hw, int64_t st,
                         const int16_t* nz, stp_lrec rc, double* rs)
{
    // (round 5: the pixels of a row are requested EIGHT at a time and then added in order -- the one-by-one form waited a memory
    //  round trip per pixel: k_lines' listing)
    const int x1 = (rc.x + rc.w < S) ? rc.x + rc.w : S;
    for (int i = tid; i < rc.h; i += nt) {
        int y = rc.y + i;
        double s = 0.0;
        if (y >= 0 && y < S) {
            const int oy = nz[y];
            const double* row = band + (st + oy) * (int64_t)W + (hw - oy);
            for (int x0 = rc.x; x0 < x1; x0 += 8) {
                double v[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int t = 0; t < 8; t++) v[t] = (x0 + t < x1) ? row[nz[x0 + t]] : 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int t = 0; t < 8; t++)
                    if (x0 + t < x1) s += (v[t] != v[t]) ? 0.0 : v[t];
            }
        }
        rs[i] = s;
    }
}
