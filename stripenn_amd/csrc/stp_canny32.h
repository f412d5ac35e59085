// stp_canny32.h -- phases of k_canny_f32: the Canny class map decided from an all-f32 evaluation of the chain
// (Gaussian passes, bleed-over scaling, Sobel, magnitude) wherever the decision is provably the reference's, and
// from the reference's own f64 arithmetic, pixel by pixel, wherever it is not.
//
// Why: k_canny_pipe reproduces every intermediate of skimage's canny (_canny.py:53-280 over scipy's f64-accumulating
// correlate1d) bit for bit -- ~95 FP64 instructions per pixel at 4 cycles per wave-instruction, the kernel's limiter.
// What the caller consumes, though, is only the CLASS of a pixel (none / low / high); the intermediates need to
// be exact only where a comparison is close.  f32 instructions issue at twice the FP64 rate and need no widening.
//
// Error budget (u = 2^-24, the relative error bound of one f32 rounding; g = largest grey value the tile's input
// window holds, taken from k_gray's per-cell maxima; grey values and Gaussian weights are >= 0, so every partial
// sum is bounded by the final one and the bounds below are relative to sums of non-negative terms):
//   vertical pass     Va = w'[R] x0, then Va = fma(RN(x[-k] + x[k]), w'[R-k], Va), w' = RN32(w): a term passes through
//                     at most 10 roundings (weight, pair sum or centre product, <= 8 accumulations):
//                     |Va - Tv| <= 10.01 u Tv  (Tv the real-number sum); the reference's V = RN32(f64 sum) has
//                     |V - Tv| <= 1.001 u Tv, so |Va - V| <= 11.01 u Tv.
//   horizontal pass   |Ha - Th(Va)| <= 10.01 u Th(Va), Th(|Va - V|) <= 11.01 u T2, |H - Th(V)| <= 1.001 u T2
//                     (T2 the real 2-D sum):  |Ha - H| <= 22.1 u T2, and T2 <= g * bleed (the same weights summed
//                     over the in-image taps), so |Ha - H| / bleed <= 22.1 u g at borders too.
//   scaling           Sa = RN32(Ha * rb), rb = RN32(1 / (bleed + eps)): two more roundings; in tiles with border columns
//                     Sa = RN32(RN32(Ha * rv) * rc), 1 / bleed = (1 / column factor)(1 / row factor): four:
//                     E_S = |Sa - S| <= 26.2 u g   (the reference's f64 quotient adds 2^-53).           used: 27 u g
//   Sobel             row terms hd = RN32(s[x+1] - s[x-1]): 2 E_S + u g;  hs = RN32(2 s[x] + RN32(s[x-1] + s[x+1])):
//                     4 E_S + 6 u g;  j = RN32(2 hd1 + RN32(hd0 + hd2)): 8 E_S + 10 u g;  i = RN32(hs2 - hs0):
//                     8 E_S + 16 u g  (|i|, |j| <= 4 g; the reference's f64 Sobel adds ~2^-50).  The class test's own
//                     Sobel sums (c32_sobel: differences first) stay below the same bound.   E_G <= 232 u g, used: 233
//   magnitude         |hypot(ia, ja) - hypot(i, j)| <= sqrt(2) E_G; f32 evaluation: fma + product (2 u of the sum of
//                     squares -> u of the root) + v_sqrt_f32 (1 ulp <= 2 u): 3.1 u m, m <= 5.66 g:
//                     E_M <= (329.6 + 17.6) u g                                                           used: 348 u g
//   interpolation     w = num / den: |wa - w| <= 2 E_G / (den - E_G) + 4 u (v_rcp_f32 1 ulp + product);
//                     l = c2 w + c1 (1 - w): |la - l| <= E_M + |wa - w| (|c2a - c1a| + 2 E_M) + 3 u max(c) (17 u g)
//   A comparison l <= m is taken from the f32 values when |la - ma| > 2 E_M + 17 u g + |wa - w| (...); m >= 0.1 / 0.2
//   when |ma - thr| > E_M; the sector (signs of i, j and |i| vs |j|) when |ia|, |ja| > E_G and ||ia| - |ja|| > 2 E_G.
//   A second budget, absolute in g instead of relative to the local sums, is much tighter for the tile-wide scale: the
//   accumulation roundings of a pass are bounded by u times the PARTIAL sums, and with inputs <= g the partial sum after
//   a step is at most g times the weights added so far -- the centre tap and the far (tiny) taps first, so the
//   partial sums stay near 0.2 g for most of the chain.  Per pass, with W the sum of the in-range weights and P_k
//   the in-range weight added up to step k:  |Va - V| <= u g (3 W + sum_k P_k)  (weights, pair sums and the centre
//   product: 2 W; the reference's own rounding: W).  For the two passes and the scaling by 1 / bleed = 1 / (Wy Wx):
//   E_S <= u g (rho_y + rho_x + 2.1 [4.2 with the two-factor scaling of border tiles]), rho = 3 + sum_k P_k / W:
//   6.2 for a full window at sigma 2 (E_S = 14.4 u g instead of 27), at most ~7.1 for a window cut by the image
//   border on one side (c32_budget computes rho for every cut of the actual weights; a window cut on both sides --
//   an image narrower than 2R+1 -- can reach 11).  The tile-wide look uses this budget (interior tiles: full windows;
//   border tiles: the worst one-sided cut, or the worst cut of all for such small images), the per-pixel second look
//   the relative one above.
//   Everything else ("uncertain": ~1e-4 of the pixels of noisy data, every edge pixel of a synthetic step) is
//   resolved by c32_res_*: the reference's arithmetic on the 5 x 5 smoothed values around the pixel, recomputed
//   from the grey image in the exact order (c32_gauss_exact), glibc's hypot, the literal tests.
//
// Symmetry (round 6).  The contact matrix is symmetric and a frame keeps the same bins as rows and as columns (getStripe.py:
// 812-822), so D[r][c] == D[c][r] bit for bit (the host verifies the band: k_band_symcheck) and every per-pixel step up to the
// brightness image is the same function of the same number at (r, c) and at (c, r).  From the box blur on the reference is
// symmetric only up to the ORDER of its roundings: cv2.filter2D sums the nine window terms row-major, so the mirror pixel sums
// the same nine terms column-major; scipy's gaussian_filter runs axis 0 before axis 1, so the mirror pixel sees the passes in the
// other order; isobel and jsobel change places, the four sectors of the local-maximum test are mirrored onto each other
// (0-45 <-> 45-90, 90-135 <-> 135-180 degrees: _canny.py:204-267 with rows and columns swapped).  k_canny_f32 with `mirror` does
// not compute the tiles strictly below the diagonal and takes their class words from the transposes of the tiles above:
//   * grey.  k_gray_c3's certificate bounds |blur~ - T| and |blur - T| against the real-number sum T of the nine terms, whatever
//     the order (10.1 u T for nine sequential additions in ANY order): a pixel the near-boundary test does not flag has the same
//     float -- hence the same grey value -- as its mirror image.  A flagged lane forms its outputs in both orders
//     (gray_c3_redo) and reports the image if one differs; such an image is not mirrored (its tiles below the diagonal go
//     to the exact kernel).  So below, grey(r, c) == grey(c, r) for every pixel of a mirrored image.
//   * budget.  Every bound above is |f32 value - T| + |T - reference value| with T the real-number value, and the second term
//     counts the reference's roundings (one per pass, the f64 dust, glibc's hypot), not their order: T is the same number at
//     both positions, the reference's value at the mirror position obeys the same bound as at the position itself, and the
//     bleed-over factor there (column factor x row factor, formed in the other order) differs by 2^-52 relative, inside the
//     slack of E_S (27 used, 26.2 needed).  So a verdict the f32 test reaches for (r, c) -- below the threshold, class by
//     the decided sector and the decided comparisons -- is the reference's verdict for (c, r) too: the mirror pixel's test is
//     the same test with the roles of i and j, and of the axis neighbours, exchanged, on values inside the same error bands.
//   * the undecidable pixels are settled per position: (r, c) as before, (c, r) by the same resolver at that position
//     (c32_res_S_any: bleed-over factors straight from the weights), each in the reference's own arithmetic and order there.
//   * a tile-image that overflows the list is flagged together with the two tiles its transpose covers.
// Checked on the CPU replay (tests/test_emu_kernels.py::test_canny_mirrored_tiles*: sizes that end inside a tile / word half,
// five radii, plateaus and exact ties on and across the diagonal) and on the device (tests/test_gpu_sym.py; every GPU test runs
// with `mirror` on, STP_SYM=0 switches it off).
#pragma once

#define C32_SP (CT_X + 6)          /* pitch (floats) of the f32 smoothed tile: even (aligned pairs), 6 mod 64 banks per row */
#define C32_EG_U 234.0f            /* the relative budget (g = local scale), in units of u * g, rounded up */
#define C32_EM_U 349.0f
#define C32_T0_U 715.0f            /* 2 E_M + 17 */

// kernel arguments (SGPRs): RN32 of the Gaussian weights, w[R] the centre; the tile-wide budget (E_G, E_M, T0 in units
// of u * g) for interior tiles [0], for tiles whose windows the image border cuts on ONE side at most (every tile of an
// image of at least 2R+1 pixels) [1], and for any cut [2] (c32_budget)
struct stp_w32 {
    float w[CT_RMAX + 1]; float eu[3][3];
    // the bleed-over factors of a row whose window lies inside the image on both sides, and of an interior column of such a
    // row (stp_bleed_v / stp_bleed_h: the same for every such row), with their f32 reciprocals (c32_rb_tables): tiles that the
    // border does not cut vertically fill their tables from these instead of running the 2R+1-step f64 loops per row
    double vfull, bifull; float rbfull, rvfull;
    // flat-window threshold for tiles whose every window lies inside the image (c32_flat_interior below)
    float flat_int;
};

// Flat-window rule for INTERIOR tiles (round 5).  STP_FLAT_RANGE (stp_phases.h) rests on "every smoothed value lies between
// the extremes of its window": |S(x+1) - S(x-1)| <= range, |sobel| <= 4 range, magnitude <= 5.657 range.  Where every
// window of the tile (its two halo rows / columns included) lies inside the image the weights are the same for
// neighbouring pixels -- the bleed-over is one constant -- and the difference is itself ONE weighted sum of the grey values:
//   S(y, x+1) - S(y, x-1) = sum_ky v_ky sum_k (v_{k-1} - v_{k+1}) g(y + ky, x + k),   v = w / sum(w)  (v = 0 beyond +-R).
// The coefficients d_k = v_{k-1} - v_{k+1} add up to 0, so any constant c may be subtracted from g; with c the middle of the
// window's range, |sum_k d_k (g - c)| <= (sum_k |d_k|) range / 2.  The weights fall away from the centre on both sides, so
// sum_k |d_k| telescopes to 2 (v_0 + v_1) (centre tap and its neighbour), and the outer sum is convex:
//   |S(x+1) - S(x-1)| <= (v_0 + v_1) range,   |jsobel|, |isobel| <= 4 (v_0 + v_1) range  (the rows' / columns' (1, 2, 1)),
//   magnitude <= 5.657 (v_0 + v_1) (range + 6e-7)        [the same two float roundings as in STP_FLAT_RANGE's derivation].
// v_0 + v_1 = 0.3755 at sigma 2 (0.307 at 2.5): magnitudes stay below 0.085 for range < 0.040 instead of 0.015.
// Border tiles keep STP_FLAT_RANGE (their windows are renormalised pixel by pixel).  Monotone weights are checked; weights that
// are not (no Gaussian is) fall back to STP_FLAT_RANGE.
STP_HD float c32_flat_interior(const double* w /* w[R] = centre */, int R)
{
    double W = w[R];
    for (int k = 1; k <= R; k++) {
        if (!(w[R - k] <= w[R - k + 1]) || !(w[R - k] >= 0.0)) return STP_FLAT_RANGE;
        W += 2.0 * w[R - k];
    }
    if (R < 1 || !(W > 0.0)) return STP_FLAT_RANGE;
    const double v01 = (w[R] + w[R - 1]) / W;
    const double thr = 0.0849 / (5.65686 * v01) - 1e-5;                  // 5.65686 > 4 sqrt 2; slack for the roundings
    const float f = (float)(thr * 0.9999);
    return f > STP_FLAT_RANGE ? (f < 1.0f ? f : 1.0f) : STP_FLAT_RANGE;
}

// rho = 3 + sum_k P_k / W of one pass for the window cut to taps lo .. hi (-R <= lo <= 0 <= hi <= R), see above
STP_HD double c32_rho(const double* w /* w[R] = centre */, int R, int lo, int hi)
{
    double W = w[R], P = w[R], sumP = 0.0;
    for (int k = R; k >= 1; k--) {
        const double add = (-k >= lo ? w[R - k] : 0.0) + (k <= hi ? w[R - k] : 0.0);
        P += add; W += add;
        sumP += P;                                     // (a step that adds nothing rounds nothing: counted anyway)
    }
    return 3.0 + sumP / W;
}
STP_HD void c32_budget(const double* w, int R, stp_w32* out)
{
    const double rho_int = c32_rho(w, R, -R, R);
    double rho_one = rho_int, rho_any = rho_int;
    for (int lo = -R; lo <= 0; lo++)
        for (int hi = 0; hi <= R; hi++) {
            const double r = c32_rho(w, R, lo, hi);
            rho_any = r > rho_any ? r : rho_any;
            if (lo == -R || hi == R) rho_one = r > rho_one ? r : rho_one;
        }
    const double es[3] = {(2.0 * rho_int + 2.1) * 1.0001, (2.0 * rho_one + 4.2) * 1.0001, (2.0 * rho_any + 4.2) * 1.0001};   // (1 + 12 u), f64 slack
    for (int t = 0; t < 3; t++) {
        const double eg = 8.0 * es[t] + 16.1, em = 1.41422 * eg + 17.7, t0 = 2.0 * em + 17.1;
        out->eu[t][0] = (float)(eg * 1.000001); out->eu[t][1] = (float)(em * 1.000001); out->eu[t][2] = (float)(t0 * 1.000001);
    }
    const int Sbig = 4 * R + 8;                                        // any size with an interior pixel
    out->vfull = stp_bleed_v(2 * R + 2, Sbig, R, w);
    out->bifull = stp_bleed_h(out->vfull, 2 * R + 2, Sbig, R, w);
    out->rbfull = (float)(1.0 / (out->bifull + DBL_EPSILON));
    out->rvfull = (float)(1.0 / out->vfull);
    out->flat_int = c32_flat_interior(w, R);
}
// which of the three a tile uses
STP_HD int c32_budget_of(bool interior, int S, int R) { return interior ? 0 : (S >= 2 * R + 1 ? 1 : 2); }

struct stp_c32tol { float Eg, Em, T0, thr; };
STP_HD stp_c32tol c32_tol_u(float gmax, float eg_u, float em_u, float t0_u)
{
    stp_c32tol t;
    const float su = gmax * 5.9604644775390625e-08f;      // g * 2^-24 (exact scaling)
    t.Eg = fmaf(eg_u, su, 1e-8f);
    t.Em = fmaf(em_u, su, 2e-8f);                          // the slack also covers 0.1f / 0.2f vs the f64 thresholds (< 3e-9)
    t.T0 = fmaf(t0_u, su, 4e-8f);
    t.thr = 0.1f - t.Em - 1e-8f;                           // below it the exact magnitude is < 0.1: no class
    return t;
}
STP_HD stp_c32tol c32_tol(float gmax) { return c32_tol_u(gmax, C32_EG_U, C32_EM_U, C32_T0_U); }      // the relative budget

// Two fused multiply-adds with a common weight as ONE v_pk_fma_f32: on gfx950 every v_fma_f32 occupies its SIMD for ~4.25
// cycles per wave, as does the packed form that does two (tools/ubench_valu.hip: additions, subtractions and
// multiplications of f32 cost ~2.2).  Each half is the same correctly rounded fma as the scalar form: results unchanged.
STP_HD void c32_fma2(float p0, float p1, float w, float* a0, float* a1)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float stp_f2 __attribute__((ext_vector_type(2)));
    // (the two sums stay scalar additions -- 2.2 cycles each, written straight into the pair the packed fma reads: as ONE
    //  packed addition their operands would have to be moved into aligned register pairs first)
    asm("" : "+v"(p0));
    asm("" : "+v"(p1));
    const stp_f2 r = __builtin_elementwise_fma((stp_f2){p0, p1}, (stp_f2){w, w}, (stp_f2){*a0, *a1});
    *a0 = r.x; *a1 = r.y;
#else
    *a0 = fmaf(p0, w, *a0); *a1 = fmaf(p1, w, *a1);
#endif
}

// ---- vertical pass, one item = column xx of the tile window x C32_VRUN_R(R) output rows from yy0 (as canny_p1_item, with
// its own run length: f32 windows are half the registers of k_canny_pipe's f64 ones).  Numbering as ct_p1_decode:
// in-image columns first, then one zero-fill item per (outside column, row group). ----
// Pitch (floats) of k_canny_f32's transposed vertical-pass tile.  The horizontal pass numbers its items (row, run) with the 36
// rows fastest, so a 32-lane group of an LDS read covers the tail of one run's rows and the head of the next run's, whose
// column is HRUN columns on: with HRUN x pitch = 36 (mod 32 banks) the second part continues the first part's bank sequence
// -- 10 x 42 = 420 = 13 x 32 + 4 -- where CT_VP = 37 put it 18 banks on, on top of the first part (round 5: a third of that
// pass's LDS cycles were conflicts).  The pitch is even, so the vertical pass stores its output pairs as one 8-byte word.
// (radii 10 and 12 -- runs of 5 -- would need a pitch of 52: they keep CT_VP.)
#define C32_VP_R(R) ((R) <= 8 ? CT_Y + 10 : CT_VP)
#define C32_VRUN_R(R) ((R) <= 8 ? 12 : 6)      /* output rows per item: the window of VRUN + 2R rows must fit the registers of
                                                  five waves per SIMD (96): 28 at radius 8, 26 / 30 at radii 10 / 12 */
struct stp_c32geo1 { int c_lo, ncv, nzc, ng, vrun; };
template <int R>
STP_HD stp_c32geo1 c32_geo1(stp_tile T)
{
    const int GW = CT_X + 2 * R + 4, VH = CT_Y + 4;
    stp_c32geo1 g;
    g.c_lo = R + 2 - T.tx0 > 0 ? R + 2 - T.tx0 : 0;
    const int c_hi = T.S - T.tx0 + R + 2 < GW ? T.S - T.tx0 + R + 2 : GW;
    g.ncv = c_hi - g.c_lo;
    g.nzc = GW - g.ncv;
    const int yy_hi = T.S - T.ty0 + 2 < VH ? T.S - T.ty0 + 2 : VH;
    g.vrun = C32_VRUN_R(R);
    g.ng = (yy_hi + g.vrun - 1) / g.vrun;
    return g;
}
// returns xx | yy0 << 8 | zero << 16, or -1 past the last item
STP_HD int c32_p1_decode(stp_c32geo1 G, int i)
{
    const int nval = G.ncv * G.ng, nzero = G.nzc * G.ng;
    if (i < nval) {
        const int yg = i / G.ncv;
        return (G.c_lo + (i - yg * G.ncv)) | (yg * G.vrun) << 8;
    }
    if (i < nval + nzero) {
        const int j = i - nval, yg = j / G.nzc, zc = j - yg * G.nzc;
        return (zc < G.c_lo ? zc : zc + G.ncv) | (yg * G.vrun) << 8 | 1 << 16;
    }
    return -1;
}
template <int R>
STP_HD void c32_p1_zero(int xx, int yy0, float* sVT)
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < C32_VRUN_R(R); q++) sVT[xx * C32_VP_R(R) + yy0 + q] = 0.0f;
}
// (guard bytes around the grey images: see canny_p1_item.)  On the device the element addresses are formed as the image's
// wave-uniform base + a 32-bit byte offset per lane (an image is 640 000 bytes), so the loads take the scalar-base form
// and a new row costs one 32-bit addition instead of a 64-bit one.
template <int R, bool YIN>
STP_HD void c32_p1_item(stp_tile T, int xx, int yy0, const stp_w32& W, const float* __restrict__ gimg, float* sVT)
{
    constexpr int VRUN = C32_VRUN_R(R), N = VRUN + 2 * R;
    float raw[N];
#if defined(__HIP_DEVICE_COMPILE__)
    // (offsets are taken from STP_GRAY_GUARD bytes in front of the image -- inside the buffer's leading guard -- so that they
    //  are non-negative for the rows above the first image row too; three rows share one scalar base: 12-bit immediates)
    const unsigned boff = (unsigned)(STP_GRAY_GUARD + ((T.ty0 - R - 2 + yy0) * STP_PITCH + (T.tx0 - R - 2 + xx)) * 4);
    const char* base = (const char*)gimg - STP_GRAY_GUARD;
#pragma unroll
    for (int k0 = 0; k0 < N; k0 += 2) {
        unsigned long long bk = (unsigned long long)(base + k0 * (STP_PITCH * 4));
        asm("" : "+s"(bk));                            // (kept a scalar base: not folded back into a 64-bit address per lane)
        typedef const __attribute__((address_space(1))) char* stp_gp;
#pragma unroll
        for (int k = k0; k < k0 + 2 && k < N; k++)
#if defined(STP_ABLATE_TD)            /* timing-only build: every STP_ABLATE_TD-th row is loaded, the others repeat it (what the
                                         kernel would gain from fewer load instructions at unchanged arithmetic) */
            if (k % STP_ABLATE_TD) raw[k] = raw[k - k % STP_ABLATE_TD] * 1.0001f; else
#endif
            // (plain loads: neighbouring tiles share their halos through L2 -- read non-temporally the kernel lost 1 ms per step, round 6)
            raw[k] = *(const __attribute__((address_space(1))) float*)((stp_gp)bk + (size_t)boff + (k - k0) * (STP_PITCH * 4));
    }
#else
    const float* col = gimg + (T.ty0 - R - 2 + yy0) * STP_PITCH + (T.tx0 - R - 2 + xx);
    for (int k = 0; k < N; k++) raw[k] = col[k * STP_PITCH];
#endif
    if (!YIN) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = 0; k < N; k++) {
            const int y = T.ty0 - R - 2 + yy0 + k;
            if ((unsigned)y >= (unsigned)T.S) raw[k] = 0.0f;
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < VRUN; q += 2) {                // two outputs at a time (c32_fma2)
        float a0 = raw[q + R] * W.w[R], a1 = raw[q + 1 + R] * W.w[R];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = R; k >= 1; k--)
            c32_fma2(raw[q + R - k] + raw[q + R + k], raw[q + 1 + R - k] + raw[q + 1 + R + k], W.w[R - k], &a0, &a1);
        if (!YIN) {
            const int y = T.ty0 - 2 + yy0 + q;
            if (!(y >= 0 && y < T.S)) a0 = 0.0f;
            if (!(y + 1 >= 0 && y + 1 < T.S)) a1 = 0.0f;
        }
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (C32_VP_R(R) % 2 == 0 && VRUN % 2 == 0) {        // (yy0 is a multiple of VRUN, the tile 16-byte aligned)
            typedef float stp_f2 __attribute__((ext_vector_type(2)));
            *(stp_f2*)__builtin_assume_aligned(sVT + xx * C32_VP_R(R) + yy0 + q, 8) = (stp_f2){a0, a1};
        } else
#endif
        {
            sVT[xx * C32_VP_R(R) + yy0 + q] = a0;
            sVT[xx * C32_VP_R(R) + yy0 + q + 1] = a1;
        }
    }
}
template <int R, bool YIN>
STP_HD void c32_p1_blk(int tid, int nt, stp_tile T, const stp_w32& W, const float* __restrict__ gimg, float* sVT)
{
    const stp_c32geo1 G = c32_geo1<R>(T);
    for (int i = tid;; i += nt) {
        const int it = c32_p1_decode(G, i);
        if (it < 0) break;
        if (it >> 16) c32_p1_zero<R>(it & 255, (it >> 8) & 255, sVT);
        else c32_p1_item<R, YIN>(T, it & 255, (it >> 8) & 255, W, gimg, sVT);
    }
}

// reciprocal bleed-over tables (f32), from the f64 factors the exact path uses; geometry only.
//   sRB[yy]  = 1 / (bleed + eps) of an interior column of tile row yy (tiles without a border column use it alone);
//   sRV[yy]  = 1 / column factor of the row,  sRC[xx] = 1 / row factor of tile column xx (0 outside the image):
// the bleed-over of (row, column) is their product up to f64 roundings (stp_bleed_h scales the row sum by the column
// factor; eps is 2^-52 of it), so border tiles scale by sRV * sRC without a test per output.
template <int R, bool ROWS_DONE = false>          // ROWS_DONE: the row tables are filled already (tiles the border does not cut vertically)
STP_HD void c32_rb_tables(int tid, int nt, stp_tile T, const double* w, const double* sB, float* sRB, float* sRV, float* sRC,
                          bool xin)
{
    const int VH = CT_Y + 4;
    if (!ROWS_DONE)
        for (int i = tid; i < VH; i += nt) {
            sRB[i] = (float)(1.0 / (sB[VH + i] + DBL_EPSILON));
            sRV[i] = (float)(1.0 / sB[i]);                      // (rows outside the image: never used)
        }
    if (xin) return;
    for (int i = tid; i < C32_SP; i += nt) {
        const int x = T.tx0 - 2 + i;
        sRC[i] = (x >= 0 && x < T.S) ? (float)(1.0 / stp_bleed_h(1.0, x, T.S, R, w)) : 0.0f;
    }
}

// ---- horizontal pass + scaling, one item (tile row yy, HRUN outputs from column xx0): as canny_p2_item ----
template <int R, bool XIN>
STP_HD void c32_p2_item(stp_tile T, int yy, int xx0, const stp_w32& W, const float* sVT, const float* sRB, const float* sRV,
                        const float* sRC, float* sS)
{
    constexpr int HRUN = CT_HRUN_R(R);
    static_assert(((CT_X + 4 + HRUN - 1) / HRUN) * HRUN <= C32_SP, "the last horizontal run must fit the row pitch");
    float win[HRUN + 2 * R];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < HRUN + 2 * R; k++) win[k] = sVT[(xx0 + k) * C32_VP_R(R) + yy];
    const float rb = XIN ? sRB[yy] : sRV[yy];
    float* srow = sS + yy * C32_SP + xx0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q + 1 < HRUN; q += 2) {     // (the last run ends at column C32_SP - 1: its outputs beyond the CT_X + 4
                                                //  columns of the tile land in the row's padding, which nothing reads)
        float a0 = win[q + R] * W.w[R], a1 = win[q + 1 + R] * W.w[R];      // two outputs at a time (c32_fma2)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = R; k >= 1; k--)
            c32_fma2(win[q + R - k] + win[q + R + k], win[q + 1 + R - k] + win[q + 1 + R + k], W.w[R - k], &a0, &a1);
        srow[q] = XIN ? a0 * rb : (a0 * rb) * sRC[xx0 + q];     // columns outside the image: sRC = 0 (their sums are finite)
        srow[q + 1] = XIN ? a1 * rb : (a1 * rb) * sRC[xx0 + q + 1];   // (round 5: requesting the pair's factors before its chain -- the
    }                                                                 //  listing shows them read one LDS round trip at a time -- changed nothing: 36.1 / 36.0 ms)
    if (HRUN & 1) {
        const int q = HRUN - 1;
        float a = win[q + R] * W.w[R];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = R; k >= 1; k--) a = fmaf(win[q + R - k] + win[q + R + k], W.w[R - k], a);
        srow[q] = XIN ? a * rb : (a * rb) * sRC[xx0 + q];
    }
}
template <int R, bool XIN>
STP_HD void c32_p2_blk(int tid, int nt, stp_tile T, stp_cgeo G, const stp_w32& W, const float* sVT, const float* sRB,
                       const float* sRV, const float* sRC, float* sS)
{
    for (int i = tid;; i += nt) {
        const int it = ct_p2_decode<R>(G, i);
        if (it < 0) break;
        c32_p2_item<R, XIN>(T, it & 255, it >> 8, W, sVT, sRB, sRV, sRC, sS);
    }
}

// the one-pixel ring around the image (see canny_p3_ring), f32 tile
STP_HD void c32_p3_ring(int tid, int nt, stp_tile T, float* sS)
{
    const int VH = CT_Y + 4, SW = CT_X + 4;
    const int y0 = T.ty0 - 2, x0 = T.tx0 - 2;
    for (int i = tid; i < 2 * SW + 2 * VH; i += nt) {
        int y, x;
        if (i < 2 * SW) { y = (i < SW) ? -1 : T.S; x = x0 + (i < SW ? i : i - SW); }
        else { const int j = i - 2 * SW; x = (j < VH) ? -1 : T.S; y = y0 + (j < VH ? j : j - VH); }
        const int yy = y - y0, xx = x - x0;
        if (yy < 0 || yy >= VH || xx < 0 || xx >= SW) continue;
        const int cy = y < 0 ? 0 : (y > T.S - 1 ? T.S - 1 : y), cx = x < 0 ? 0 : (x > T.S - 1 ? T.S - 1 : x);
        const int sy = cy - y0, sx = cx - x0;
        if (sy < 0 || sy >= VH || sx < 0 || sx >= SW) continue;
        sS[yy * C32_SP + xx] = sS[sy * C32_SP + sx];
    }
}

STP_HD float c32_sqrt(float q)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(q);
#else
    return sqrtf(q);
#endif
}
STP_HD float c32_rcp(float d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(d);
#else
    return 1.0f / d;
#endif
}
// f32 Sobel sums of the pixel at `c` (plain offsets: the ring holds the replicated edge)
STP_HD void c32_sobel(const float* c, float* is, float* js)
{
    const float s00 = c[-C32_SP - 1], s01 = c[-C32_SP], s02 = c[-C32_SP + 1], s10 = c[-1], s12 = c[1], s20 = c[C32_SP - 1],
                s21 = c[C32_SP], s22 = c[C32_SP + 1];
    *js = fmaf(s12 - s10, 2.0f, (s02 - s00) + (s22 - s20));
    *is = fmaf(s21 - s01, 2.0f, (s20 - s00) + (s22 - s02));
}
// Magnitudes by rows.  Per smoothed row r and magnitude column X (smoothed columns X .. X+2) two terms:
//   hd = s[X+2] - s[X]   and   hs = 2 s[X+1] + (s[X] + s[X+2]);
// the magnitude of (Y, X) takes them from rows Y, Y+1, Y+2:  j = 2 hd1 + (hd0 + hd2),  i = hs2 - hs0.
// A lane walking down a column keeps the terms of the two previous rows: 3 LDS reads and 9 operations per pixel.
STP_HD void c32_row_terms(const float* s3, float* hd, float* hs)
{
    const float a = s3[0], b = s3[1], c = s3[2];
    *hd = c - a;
    *hs = fmaf(b, 2.0f, a + c);
}
STP_HD float c32_mag(float hd0, float hd1, float hd2, float hs0, float hs2)
{
    const float j = fmaf(hd1, 2.0f, hd0 + hd2), i = hs2 - hs0;
    return c32_sqrt(fmaf(i, i, j * j));
}
// one pixel of the magnitude tile from scratch (the two halo columns of k_canny_f32; CPU replay of the whole region)
STP_HD float c32_mag_px(const float* sS, int Y, int X)
{
    float hd0, hd1, hd2, hs0, hs1, hs2;
    c32_row_terms(sS + Y * C32_SP + X, &hd0, &hs0);
    c32_row_terms(sS + (Y + 1) * C32_SP + X, &hd1, &hs1);
    c32_row_terms(sS + (Y + 2) * C32_SP + X, &hd2, &hs2);
    return c32_mag(hd0, hd1, hd2, hs0, hs2);
}
// the in-image region of the magnitude tile (CPU replay; the kernel walks it by rows in canny32_mag_rows)
STP_HD void c32_p3_region(stp_cgeo G, const float* sS, float* sM)
{
    for (int Y = G.my_lo; Y < G.my_lo + G.nmh; Y++)
        for (int X = G.mx_lo; X < G.mx_lo + (int)G.nmw.d; X++) sM[Y * (CT_X + 2) + X] = c32_mag_px(sS, Y, X);
}

// Class of tile pixel (y, x) from the f32 tiles: 0 / 1 / 2, or 3 = not decidable inside the error budget.
// Decision structure of _canny.py:193-280 as in ct_nms.
STP_HD int c32_nms_E(const float* sS, const float* sM, stp_tile T, int y, int x, stp_c32tol E)
{
    if (y < 1 || x < 1 || y >= T.S - 1 || x >= T.S - 1) return 0;
    const int MW = CT_X + 2;
    const float* mp = sM + (y - (T.ty0 - 1)) * MW + (x - (T.tx0 - 1));
    const float m0 = mp[0];
    if (m0 < E.thr) return 0;                                            // exact magnitude < 0.1
    if (m0 <= 0.1f + E.Em || fabsf(m0 - 0.2f) <= E.Em) return 3;         // a threshold inside the error band
    float gi, gj;
    c32_sobel(sS + (y - (T.ty0 - 2)) * C32_SP + (x - (T.tx0 - 2)), &gi, &gj);
    const float ai = fabsf(gi), aj = fabsf(gj);
    // (round 5, measured and dropped: where exactly ONE of the three questions is open the reference used one of the two
    //  sectors that meet there, so evaluating both and accepting an agreeing verdict decides a quarter of the pixels that go to
    //  the resolver -- class maps identical, diagonal step edges no longer need the resolver at all -- but the kernel's time did
    //  not move (the resolver costs a tile one pixel's latency whether it settles one pixel or three) and the extra code cost
    //  radii 4 and 12 a wave of occupancy.)
    if (ai <= E.Eg || aj <= E.Eg || fabsf(ai - aj) <= 2.0f * E.Eg) return 3;      // a sign or the octant could differ
    const bool same = (gi > 0.0f) == (gj > 0.0f), ibig = ai > aj;
    const float num = ibig ? aj : ai, den = ibig ? ai : aj;
    // neighbours of the "plus" side: (dy1, dx1) the axis neighbour, (dy2, dx2) the diagonal one
    const int dy2 = same ? 1 : -1;
    const int o1 = ibig ? dy2 * MW : 1, o2 = dy2 * MW + 1;
    const float wq = num * c32_rcp(den), omw = 1.0f - wq;
    const float c1p = mp[o1], c2p = mp[o2], c1m = mp[-o1], c2m = mp[-o2];
    const float lp = fmaf(c2p, wq, c1p * omw), lm = fmaf(c2m, wq, c1m * omw);
    const float dw = fmaf(2.0f * E.Eg, c32_rcp(den - E.Eg) * 1.000001f, 2.4e-7f);     // |wa - w| bound (4 u = 2.4e-7)
    const float tp = fmaf(dw, fabsf(c2p - c1p) + 2.0f * E.Em, E.T0), tm = fmaf(dw, fabsf(c2m - c1m) + 2.0f * E.Em, E.T0);
    if (fabsf(lp - m0) <= tp || fabsf(lm - m0) <= tm) return 3;
    if (!(lp <= m0 && lm <= m0)) return 0;
    return m0 >= 0.2f ? 2 : 1;
}

// The class test proper.  The budget above is relative to the grey values that reach the pixels involved, so a pixel the
// tile-wide budget (g = the largest grey value of the tile's whole input window) leaves undecided gets a second look
// with g = the largest smoothed value of its 5 x 5 neighbourhood -- which holds the 3 x 3 neighbourhoods of the pixel
// and of the four neighbours it is compared with; every bound above is a sum over those pixels, each term relative to
// the value it belongs to (+25 u for taking the f32 values as the scale).  Unwritten cells beyond the image ring are
// not part of any of those sums (NaN stand-ins in the CPU replay are ignored by fmaxf; any other value only widens the budget).
STP_HD int c32_nms(const float* sS, const float* sM, stp_tile T, int y, int x, stp_c32tol E)
{
    const int cls = c32_nms_E(sS, sM, T, y, x, E);
    if (cls != 3) return cls;
    const float* c = sS + (y - (T.ty0 - 2)) * C32_SP + (x - (T.tx0 - 2));
    float sm = 0.0f;
    for (int dy = -2; dy <= 2; dy++)
        for (int dx = -2; dx <= 2; dx++) sm = fmaxf(sm, c[dy * C32_SP + dx]);
    return c32_nms_E(sS, sM, T, y, x, c32_tol(sm * 1.00001f));
}

// The reference's order for one output (as stp_gauss_exact) with a compile-time radius: every tap is loaded before the
// first use, so the 2R+1 loads of a lane are in flight together (the resolver is latency, not throughput).
// Tap k (0 .. 2R) counts as 0 unless lo <= k <= hi (constant-mode zero padding); masked taps are not dereferenced.
// In two halves, so that a caller can request the taps of several outputs before summing the first:
template <int R>
STP_HD void c32_gauss_taps(const float* centre, int stride, int lo, int hi, float* v)
{
    // Every tap is READ -- the rows / columns beyond the image lie in the guard regions around the grey images or in a
    // neighbouring image: valid memory, as for the vertical pass -- and the masked ones are then replaced by 0.  (round 5: as one
    // conditional expression per tap each load sat in its own branch with its conversion right behind it, and the 2R+1
    // loads of a lane went out one memory round trip after the other: the resolver's time was mostly that.)
    float raw[2 * R + 1];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k <= 2 * R; k++) raw[k] = centre[(k - R) * stride];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k <= 2 * R; k++) v[k] = (k >= lo && k <= hi) ? raw[k] : 0.0f;
}
template <int R>
STP_HD float c32_gauss_sum(const float* v, const double* w)
{
    double a = (double)v[R] * w[R];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = R; k >= 1; k--) a += ((double)v[R - k] + (double)v[R + k]) * w[R - k];
    return (float)a;
}
template <int R>
STP_HD float c32_gauss_exact(const float* centre, int stride, const double* w, int lo, int hi)
{
    if (R > 8) {                                   // (25 / 21 taps at once would not fit the registers of five waves per SIMD)
        double a = (R >= lo && R <= hi) ? (double)centre[0] * w[R] : 0.0 * w[R];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 4
#endif
        for (int k = R; k >= 1; k--) {                // (four tap pairs -- eight loads -- in flight at a time)
            const float rl = centre[-k * stride], rh = centre[k * stride];           // (read, then masked: see c32_gauss_taps)
            const double xl = (R - k >= lo && R - k <= hi) ? (double)rl : 0.0;
            const double xh = (R + k >= lo && R + k <= hi) ? (double)rh : 0.0;
            a += (xl + xh) * w[R - k];
        }
        return (float)a;
    }
    float v[2 * R + 1];
    c32_gauss_taps<R>(centre, stride, lo, hi, v);
    return c32_gauss_sum<R>(v, w);
}

// ---- the exact resolver: one pixel (y, x) of the tile, interior of the image ----
// V patch: 5 rows (y-2 .. y+2, clamped into the image: scipy's 'reflect' by one = edge replication) x (2R + 5) columns
// x-2-R .. x+2+R of the vertical pass (0 outside the image: constant mode), element l = r * (2R+5) + c.
// c32_res_V_taps requests the taps of element l (nothing for an element outside the image: all taps 0),
// c32_gauss_sum turns them into the value.
// TR (the mirrored resolver, image symmetry): the grey value of image position (a, b) is READ at (b, a).  The resolver settles
// pixels of tiles below the diagonal, whose grey windows may lie in tiles k_gray_c3 no longer writes (stp_gray_dead_tile); in
// an image that equals its transpose -- any other is reported by k_gray_c3 and its tiles below the diagonal are recomputed
// from grey values filled in for them (k_gray_fill) -- the transposed position holds the same float.  Taps beyond the
// image are read (the neighbouring row, image or guard region: valid memory) and masked, as in the plain form.
template <int R, bool TR = false>
STP_HD void c32_res_V_taps(stp_tile T, int y, int x, int l, const float* __restrict__ gimg, float* v)
{
    constexpr int NC = 2 * R + 5;
    const int r = l / NC, c = l - r * NC;
    int yy = y - 2 + r;
    yy = yy < 0 ? 0 : (yy > T.S - 1 ? T.S - 1 : yy);
    const int xx = x - 2 - R + c;
    const bool in = xx >= 0 && xx < T.S;
    const int lo = !in ? 1 : (yy - R < 0 ? R - yy : 0), hi = !in ? 0 : (yy + R >= T.S ? R + (T.S - 1 - yy) : 2 * R);
    if (TR) c32_gauss_taps<R>(gimg + (in ? xx : 0) * STP_PITCH + yy, 1, lo, hi, v);
    else c32_gauss_taps<R>(gimg + yy * STP_PITCH + (in ? xx : 0), STP_PITCH, lo, hi, v);
}
template <int R, bool TR = false>
STP_HD float c32_res_V(stp_tile T, int y, int x, int l, const double* w, const float* __restrict__ gimg)
{
    if (R > 8) {                                   // (as c32_gauss_exact: one tap pair at a time)
        constexpr int NC = 2 * R + 5;
        const int r = l / NC, c = l - r * NC;
        int yy = y - 2 + r;
        yy = yy < 0 ? 0 : (yy > T.S - 1 ? T.S - 1 : yy);
        const int xx = x - 2 - R + c;
        const bool in = xx >= 0 && xx < T.S;
        const int lo = !in ? 1 : (yy - R < 0 ? R - yy : 0), hi = !in ? 0 : (yy + R >= T.S ? R + (T.S - 1 - yy) : 2 * R);
        if (TR) return c32_gauss_exact<R>(gimg + (in ? xx : 0) * STP_PITCH + yy, 1, w, lo, hi);
        return c32_gauss_exact<R>(gimg + yy * STP_PITCH + (in ? xx : 0), STP_PITCH, w, lo, hi);
    }
    float v[2 * R + 1];
    c32_res_V_taps<R, TR>(T, y, x, l, gimg, v);
    return c32_gauss_sum<R>(v, w);
}
// S patch: element l = r * 5 + c is the smoothed value at (clamp(y-2+r), clamp(x-2+c)), the reference's operations
// (canny_p2: f32 horizontal sum in the exact order, f64 quotient by bleed + eps; bleed-over factors as canny_p1b
// tabulates them in sB: [yy] the column factor of tile row yy, [VH + yy] the full factor for an interior column)
template <int R>
STP_HD double c32_res_S(stp_tile T, int y, int x, int l, const double* w, const double* sB, const float* Vp)
{
    constexpr int NC = 2 * R + 5;
    const int VH = CT_Y + 4;
    const int r = l / 5, c = l - r * 5;
    int yy = y - 2 + r, cx = x - 2 + c;
    yy = yy < 0 ? 0 : (yy > T.S - 1 ? T.S - 1 : yy);
    cx = cx < 0 ? 0 : (cx > T.S - 1 ? T.S - 1 : cx);
    const float f = c32_gauss_exact<R>(Vp + r * NC + (cx - (x - 2 - R)), 1, w, 0, 2 * R);
    const int yyt = yy - (T.ty0 - 2);
    const double bl = (cx >= R && cx + R < T.S) ? sB[VH + yyt] : stp_bleed_h(sB[yyt], cx, T.S, R, w);
    return (double)f / (bl + DBL_EPSILON);       // _canny.py:49
}
// The same element for a pixel ANYWHERE in the image (the mirrored resolver of a tile above the diagonal settles pixels of
// the tiles below it): the bleed-over factors straight from the weights -- stp_bleed_v / stp_bleed_h are what canny_p1b
// tabulates, and for an interior column stp_bleed_h(V, cx) performs the very operations of the tabulated stp_bleed_h(V, R).
template <int R>
STP_HD double c32_res_S_any(int S, int y, int x, int l, const double* w, const float* Vp)
{
    constexpr int NC = 2 * R + 5;
    const int r = l / 5, c = l - r * 5;
    int yy = y - 2 + r, cx = x - 2 + c;
    yy = yy < 0 ? 0 : (yy > S - 1 ? S - 1 : yy);
    cx = cx < 0 ? 0 : (cx > S - 1 ? S - 1 : cx);
    const float f = c32_gauss_exact<R>(Vp + r * NC + (cx - (x - 2 - R)), 1, w, 0, 2 * R);
    const double bl = stp_bleed_h(stp_bleed_v(yy, S, R, w), cx, S, R, w);
    return (double)f / (bl + DBL_EPSILON);       // _canny.py:49
}
// magnitude (glibc hypot of the f64 Sobel sums) of pixel l = 3 * (dy + 1) + (dx + 1) of the 3 x 3 block around the
// centre of the 5 x 5 patch (pitch 5)
STP_HD double c32_res_mag(const double* P, int l)
{
    const int r = l / 3, c = l - 3 * r;
    double is, js;
    ct_sobel_off(P + (r + 1) * 5 + (c + 1), -5, 5, -1, 1, &is, &js);
    return stp_hypot(is, js);
}
// class of the centre pixel from the patch and the nine magnitudes: ct_nms's literal branch
STP_HD int c32_res_class(const double* P, const double* M9)
{
    double gi, gj;
    ct_sobel_off(P + 12, -5, 5, -1, 1, &gi, &gj);
    const double ai = fabs(gi), aj = fabs(gj);
    const bool same = (gi >= 0 && gj >= 0) || (gi <= 0 && gj <= 0);
    const bool opp = (gi <= 0 && gj >= 0) || (gi >= 0 && gj <= 0);
    int dy1, dx1, dy2, dx2;
    double num, den;
    if (opp && ai >= aj) { num = aj; den = ai; dy1 = -1; dx1 = 0; dy2 = -1; dx2 = 1; }
    else if (opp && ai <= aj) { num = ai; den = aj; dy1 = 0; dx1 = 1; dy2 = -1; dx2 = 1; }
    else if (same && ai <= aj) { num = ai; den = aj; dy1 = 0; dx1 = 1; dy2 = 1; dx2 = 1; }
    else if (same && ai >= aj) { num = aj; den = ai; dy1 = 1; dx1 = 0; dy2 = 1; dx2 = 1; }
    else return 0;
    const int o1 = dy1 * 3 + dx1, o2 = dy2 * 3 + dx2;
    const double wq = num / den, omw = 1.0 - wq;
    const double m = M9[4], c1p = M9[4 + o1], c2p = M9[4 + o2], c1m = M9[4 - o1], c2m = M9[4 - o2];
    const double lp = c2p * wq + c1p * omw, lm = c2m * wq + c1m * omw;
    if (!(m > 0.0)) return 0;
    if (!(lp <= m && lm <= m)) return 0;
    return (m >= 0.2) ? 2 : ((m >= 0.1) ? 1 : 0);
}
