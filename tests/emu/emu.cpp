// tests/emu/emu.cpp -- CPU replay of ONE workgroup of each libstripenn_hip kernel (tid = 0, nt = 1).
// TEST-ONLY: lets the CPU test-suite check the kernels' phase code (stripenn_amd/csrc/stp_phases.h)
// against the oracle without a GPU.  Never linked into the product library.
#include <vector>
#include <algorithm>
#include <cmath>
#include <limits>
#include <string.h>
#include "../../include/stripenn_hip.h"
#include "../../stripenn_amd/csrc/stp_phases.h"
#include "../../stripenn_amd/csrc/stp_canny32.h"


// k_canny_f32 of one tile: the f32 phases, the certified class test and the exact per-pixel resolver
static long long g_c32_uncertain = 0, g_c32_candidates = 0;
// optional full-image dumps of the f32 intermediates (tests of the error budget): smoothed value, Sobel sums, magnitude,
// and the scale g the tile used; pitch STP_PITCH
static float g_c32_last_eg = 0.0f;
static float *g_c32_dS = nullptr, *g_c32_dI = nullptr, *g_c32_dJ = nullptr, *g_c32_dM = nullptr, *g_c32_dG = nullptr, *g_c32_dE = nullptr;
// sCm (image symmetry, k_canny_f32's `mirror`): class of the MIRROR image (column, row) of every tile pixel, [tile column][tile
// row] -- the f32 verdict where there is one, the reference's arithmetic at the mirror position where there is not
template <int R>
static void emu_tile_c32(stp_tile T, const double* w, const float* gray, float gmax, const double* sB, float* sV, float* sM,
                         uint8_t* sC, bool xin, bool yin, uint8_t* sCm = nullptr)
{
    const int VH = CT_Y + 4;
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    std::vector<float> sS(VH * C32_SP, qnan), sRB(VH, qnan), sRV(VH, qnan), sRC(C32_SP, qnan);
    stp_w32 W;
    for (int k = 0; k <= R; k++) W.w[k] = (float)w[k];
    c32_budget(w, R, &W);
    const stp_cgeo G = ct_geo<R>(T);
    c32_rb_tables<R>(0, 1, T, w, sB, sRB.data(), sRV.data(), sRC.data(), xin);
    if (yin) c32_p1_blk<R, true>(0, 1, T, W, gray, sV);
    else c32_p1_blk<R, false>(0, 1, T, W, gray, sV);
    if (xin) c32_p2_blk<R, true>(0, 1, T, G, W, sV, sRB.data(), sRV.data(), sRC.data(), sS.data());
    else c32_p2_blk<R, false>(0, 1, T, G, W, sV, sRB.data(), sRV.data(), sRC.data(), sS.data());
    if (!(xin && yin)) c32_p3_ring(0, 1, T, sS.data());
    c32_p3_region(G, sS.data(), sM);
    const int et = c32_budget_of(xin && yin, T.S, R);
    const stp_c32tol E = c32_tol_u(gmax, W.eu[et][0], W.eu[et][1], W.eu[et][2]);
    g_c32_last_eg = W.eu[et][0];
    constexpr int NV = 5 * (2 * R + 5);
    for (int i = 0; i < CT_Y * CT_X; i++) {
        const int yy = i / CT_X, xx = i - yy * CT_X, y = T.ty0 + yy, x = T.tx0 + xx;
        int cls = 0;
        if (g_c32_dS && y < T.S && x < T.S) {
            const float* c = sS.data() + (yy + 2) * C32_SP + (xx + 2);
            float gi, gj;
            c32_sobel(c, &gi, &gj);
            const int o = y * STP_PITCH + x;
            g_c32_dS[o] = c[0]; g_c32_dI[o] = gi; g_c32_dJ[o] = gj; g_c32_dM[o] = sM[(yy + 1) * (CT_X + 2) + xx + 1]; g_c32_dG[o] = gmax;
            if (g_c32_dE) g_c32_dE[o] = W.eu[et][0];
        }
        if (y < T.S && x < T.S) {
            cls = c32_nms(sS.data(), sM, T, y, x, E);
            if (y >= 1 && x >= 1 && y < T.S - 1 && x < T.S - 1 && sM[(yy + 1) * (CT_X + 2) + xx + 1] >= E.thr) g_c32_candidates++;
            if (sCm) sCm[xx * CT_Y + yy] = (uint8_t)cls;
            if (cls == 3) {
                g_c32_uncertain++;
                float Vp[NV];
                double Sp[25];
                for (int l = 0; l < NV; l++) Vp[l] = c32_res_V<R>(T, y, x, l, w, gray);
                for (int l = 0; l < 25; l++) Sp[l] = c32_res_S<R>(T, y, x, l, w, sB, Vp);
                double M9[9];
                for (int l = 0; l < 9; l++) M9[l] = c32_res_mag(Sp, l);
                cls = c32_res_class(Sp, M9);
                if (sCm) {                              // the mirror pixel (x, y): canny32_resolve<R, true>
                    for (int l = 0; l < NV; l++) Vp[l] = c32_res_V<R, true>(T, x, y, l, w, gray);      // (grey values read at the transposed position)
                    for (int l = 0; l < 25; l++) Sp[l] = c32_res_S_any<R>(T.S, x, y, l, w, Vp);
                    for (int l = 0; l < 9; l++) M9[l] = c32_res_mag(Sp, l);
                    sCm[xx * CT_Y + yy] = (uint8_t)c32_res_class(Sp, M9);
                }
            }
        } else if (sCm) sCm[xx * CT_Y + yy] = 0;
        sC[i] = (uint8_t)cls;
    }
}

// k_canny_pipe<R>'s phases of one tile (the device path of the tiled radii)
template <int R>
static void emu_tile_pipe(stp_tile T, const double* w, const float* gray, const double* sB, float* sV, double* sS, float* sM,
                          stp_fastdiv fd, bool xin, bool yin)
{
    const int VH = CT_Y + 4;
    std::vector<double> sBB(VH * 2 * R);
    const stp_cgeo G = ct_geo<R>(T);
    if (!xin) canny_p1c<R>(0, 1, T, w, sB, sBB.data());
    if (yin) canny_p1_blk_g<R, true>(0, 1, T, G, w, gray, sV);
    else canny_p1_blk_g<R, false>(0, 1, T, G, w, gray, sV);
    if (xin) canny_p2_blk<R, true>(0, 1, T, G, w, sV, sB, sBB.data(), sS, fd);
    else canny_p2_blk<R, false>(0, 1, T, G, w, sV, sB, sBB.data(), sS, fd);
    if (!(xin && yin)) canny_p3_ring(0, 1, T, sS);
    canny_p3_reg(0, 1, G, sS, sM);
}
static bool emu_tiled_radius(int R) { return R == 4 || R == 6 || R == 8 || R == 10 || R == 12; }

extern "C" {

// zero-column removal (mirror of k_frame_compact)
int emu_compact(const double* band, int W, int hw, int64_t st, int n0, int16_t* nz)
{
    int S = 0;
    for (int c = 0; c < n0; c++) {
        double sum = 0.0;
        for (int r = 0; r < n0; r++) {
            double v = band[(st + r) * (int64_t)W + (c - r + hw)];
            if (v != v) v = 0.0;
            sum += v;
        }
        if (sum != 0.0) nz[S++] = (int16_t)c;
    }
    return S > 10 ? S : 0;
}

void emu_gray(const double* band, int W, int hw, int64_t st, const int16_t* nz, int S, double M, const double* bvals,
              int nb, int a, float* gray /* nb images, pitch 400 */)
{
    std::vector<double> sg((GT_Y + 2 * GT_AMAX) * (GT_X + 2 * GT_AMAX)), sadj(sg.size());
    for (int ty0 = 0; ty0 < S; ty0 += GT_Y)
        for (int tx0 = 0; tx0 < S; tx0 += GT_X) {
            stp_tile T; T.S = S; T.ty0 = ty0; T.tx0 = tx0;
            gray_p0(0, 1, band, W, hw, st, nz, T, a, M, sg.data());
            for (int bi = 0; bi < nb; bi++) {
                if (a == 1) {       // wave-strip form (the device path for bfilter 3): lanes replayed one by one
                    for (int strip = 0; strip < GT_Y / GS_ROWS; strip++) {
                        double* srow = sadj.data() + strip * (GT_X + 2);
                        stp_gray_lane st[64];
                        for (int lane = 0; lane < 64; lane++) {
                            st[lane].vmin = 0x7F800000u; st[lane].vmax = 0u;
                            for (int c = 0; c < 3; c++) st[lane].w0[c] = st[lane].w1[c] = 0.0;
                        }
                        for (int r = 0; r < GS_ROWS + 2; r++) {
                            for (int lane = 0; lane < 64; lane++) gray_wrow_put(lane, strip, r, bvals[bi], sg.data(), srow);
                            for (int lane = 0; lane < 64; lane++)
                                gray_wrow_get(lane, strip, r, T, srow, &st[lane], gray + (size_t)bi * STP_PITCH * STP_PITCH);
                        }
                    }
                } else {
                    gray_p1(0, 1, a, bvals[bi], sg.data(), sadj.data());
                    gray_p2(0, 1, T, a, sadj.data(), gray + (size_t)bi * STP_PITCH * STP_PITCH);
                }
            }
        }
}

// k_gray_c3 (certified grey, bfilter 3), pixel by pixel: the cheap evaluation from the nine g~ values, the reference's
// operations where the near-boundary test flags it; counts[0] = outputs, counts[1] = flagged ones
void emu_gray_c3(const double* band, int W, int hw, int64_t st, const int16_t* nz, int S, double M, const double* bvals,
                 int nb, float* gray /* nb images, pitch 400 */, long long* counts)
{
    const double rM = 1.0 / M;
    long long nout = 0, nflag = 0;
    for (int bi = 0; bi < nb; bi++) {
        const double b = bvals[bi], cb = stp_gray_cb(b);
        for (int y = 0; y < S; y++)
            for (int x = 0; x < S; x++) {
                double D9[9], g9[9];
                for (int i = 0; i < 9; i++) {
                    const int oy = nz[stp_refl101(y - 1 + i / 3, S)], ox = nz[stp_refl101(x - 1 + i % 3, S)];
                    double d = band[(st + oy) * (int64_t)W + (ox - oy + hw)];
                    if (d != d) d = 0.0;
                    D9[i] = d;
                    g9[i] = stp_gplane_fast(d, M, rM);
                }
                unsigned near;
                float v = stp_gray_c3_px(g9, b, cb, &near);
                nout++;
                if (near < 16u * STP_GRAY_NEAR) { v = stp_gray_exact9(D9, M, b); nflag++; }
                gray[(size_t)bi * STP_PITCH * STP_PITCH + y * STP_PITCH + x] = v;
            }
    }
    if (counts) { counts[0] = nout; counts[1] = nflag; }
}
// The claim behind k_gray_c3 on n random 3 x 3 windows: an output the near-boundary test does not flag is the reference's
// float.  mode 0: contact values spread over [0, 1.2 M] with zeros, saturated and near-saturated pixels mixed in;
// mode 1: all nine values close to each other (flat regions); mode 2: values at M (1 - 2^-k) and M b (1 +- few ulp).
// Returns the number of unflagged outputs that differ; *flagged counts the flagged ones, *worst is the largest
// |blur~ - blur| in ulp(f64) seen.
long long emu_certify_gray(long long n, unsigned long long seed, int mode, long long* flagged, double* worst)
{
    unsigned long long sst = seed * 0x9E3779B97F4A7C15ull + 12345ull;
    auto rnd = [&]() { sst ^= sst << 13; sst ^= sst >> 7; sst ^= sst << 17; return sst; };
    auto uni = [&]() { return (double)(rnd() >> 11) * (1.0 / 9007199254740992.0); };
    const double bl[6] = {0.5, 0.6, 0.7, 0.7999999999999999, 0.8999999999999999, 0.9999999999999999};
    long long bad = 0, nfl = 0;
    double w = 0.0;
    for (long long it = 0; it < n; it++) {
        const double M = ldexp(1.0 + uni(), (int)(rnd() % 20) - 10);
        const double b = (rnd() & 7) ? bl[rnd() % 6] : 0.3 + 0.7 * uni();
        const double rM = 1.0 / M, cb = stp_gray_cb(b);
        double D9[9], g9[9];
        const double centre = uni() * 1.1 * M;
        for (int i = 0; i < 9; i++) {
            double d;
            const unsigned sel = (unsigned)(rnd() % 16);
            if (mode == 1) d = centre * (1.0 + (uni() - 0.5) * 1e-3);
            else if (mode == 2) {
                if (sel < 6) d = M * (1.0 - ldexp(1.0, -(int)(rnd() % 52) - 1));
                else if (sel < 12) d = M * (1.0 - b) * (1.0 + ((double)(rnd() % 9) - 4.0) * 2.220446049250313e-16);
                else d = uni() * M;
            } else {
                if (sel == 0) d = 0.0;
                else if (sel == 1) d = M * (1.0 + uni());            // beyond M: blue 0
                else if (sel == 2) d = M * (1.0 - b) * uni();        // above the brightness cut: saturated
                else d = uni() * 1.2 * M;
            }
            if (d < 0.0) d = 0.0;
            D9[i] = d;
            g9[i] = stp_gplane_fast(d, M, rM);
        }
        const double be = stp_gray_exact9_blur(D9, M, b), ba = stp_gray_c3_blur(g9, b, cb);
        if (be > 0.0 && be < 1.0) {
            const double ulp = ldexp(1.0, ilogb(ba > be ? ba : be) - 52), e = fabs(ba - be) / ulp;
            if (e > w) w = e;
        }
        unsigned near;
        const float va = stp_gray_c3_px(g9, b, cb, &near), ve = stp_gray_exact9(D9, M, b);
        if (near < 16u * STP_GRAY_NEAR) nfl++;
        else if (memcmp(&va, &ve, 4) != 0) bad++;
    }
    *flagged = nfl; *worst = w;
    return bad;
}

void emu_canny2(const float* gray_in /* pitch 400 */, int S, int R, const double* w, stp_u64* low, stp_u64* high, int blocked)
{
    // as in the library: the grey image sits between two guard regions (filled with NaN here: whatever the
    // border tiles load from them must never reach a result)
    const size_t guard = STP_GRAY_GUARD / sizeof(float), npx = (size_t)STP_PITCH * STP_PITCH;
    std::vector<float> gpad(npx + 2 * guard, std::numeric_limits<float>::quiet_NaN());
    memcpy(gpad.data() + guard, gray_in, npx * sizeof(float));
    const float* gray = gpad.data() + guard;
    const int GW = ct_gw(R), GH = CT_Y + 2 * R + 4, VH = CT_Y + 4;
    std::vector<float> sG(GH * GW), sV((VH * GW > GW * (CT_Y + 10) ? VH * GW : GW * (CT_Y + 10)) + 8 * (CT_Y + 10));   // + the horizontal pass's padding columns
    std::vector<double> sB(2 * VH), sS(VH * CT_SP);
    std::vector<float> sM((CT_Y + 2) * (CT_X + 2));
    std::vector<uint8_t> sC(CT_Y * CT_X);
    memset(low, 0, sizeof(stp_u64) * STP_FRAME_MAX * STP_NW);
    memset(high, 0, sizeof(stp_u64) * STP_FRAME_MAX * STP_NW);
    stp_fastdiv fd;                     // as the library does: interior constant, verified on 2^23 mantissas
    {
        const int Sx = 4 * R + 8;
        const double V = stp_bleed_v(2 * R + 2, Sx, R, w);
        fd.c = stp_bleed_h(V, 2 * R + 2, Sx, R, w) + DBL_EPSILON; fd.rc = 1.0 / fd.c; fd.ok = 1;
        for (uint32_t m = 0; m < (1u << 23) && fd.ok; m++) {
            uint32_t bits = 0x3F800000u | m; float f; memcpy(&f, &bits, 4);
            const double q = stp_div_const((double)f, fd.c, fd.rc), t = (double)f / fd.c;
            if (memcmp(&q, &t, 8) != 0) fd.ok = 0;
        }
    }
    // blocked 3: k_canny_f32 with `mirror` -- the tiles strictly below the diagonal are not computed, the tiles whose transpose
    // lies there hand their class words over transposed (the image must equal its transpose)
    const bool mirror = blocked == 3;
    if (mirror) blocked = 2;
    std::vector<uint8_t> sCm(CT_Y * CT_X);
    for (int ty0 = 0; ty0 < S; ty0 += CT_Y)
        for (int tx0 = 0; tx0 < S; tx0 += CT_X) {
            stp_tile T; T.S = S; T.ty0 = ty0; T.tx0 = tx0;
            const int tyi = ty0 / CT_Y, txi = tx0 / CT_X;
            if (mirror && tyi >= 2 * txi + 2) continue;
            const bool mir = mirror && txi > (tyi >> 1);
            float gmax = 1.0f;
            if (blocked && emu_tiled_radius(R)) {
                // k_canny_pipe's flat-window rule, evaluated here straight from the grey image over the same
                // cell-aligned window (the kernel unites k_gray's per-cell min / max): skipped tiles stay all-zero
                const int wy0 = std::max(ty0 - R - 2, 0), wy1 = std::min(ty0 + CT_Y + R + 2, S);
                const int wx0 = std::max(tx0 - R - 2, 0), wx1 = std::min(tx0 + CT_X + R + 2, S);
                const int Y0 = wy0 / GC_CY * GC_CY, Y1 = std::min(((wy1 - 1) / GC_CY + 1) * GC_CY, S);
                const int X0 = wx0 / GC_CX * GC_CX, X1 = std::min(((wx1 - 1) / GC_CX + 1) * GC_CX, S);
                float mn = INFINITY, mx = -INFINITY;
                for (int y = Y0; y < Y1; y++)
                    for (int x = X0; x < X1; x++) { const float v = gray[y * STP_PITCH + x]; mn = std::min(mn, v); mx = std::max(mx, v); }
                // (k_canny_f32 calls a wider range flat in tiles whose every window lies inside the image: c32_flat_interior)
                const bool interior = (ty0 - R - 2 >= 0) && (ty0 + CT_Y + R + 1 < S) && (tx0 - 2 - R >= 0) && (tx0 + CT_X + 1 + R < S);
                if (mx - mn < ((blocked == 2 && interior) ? c32_flat_interior(w, R) : STP_FLAT_RANGE)) continue;
                gmax = mx;
            }
            canny_p0(0, 1, gray, T, R, sG.data());
            canny_p1b(0, 1, T, R, w, sB.data());
            const bool yin = (T.ty0 - R - 2 >= 0) && (T.ty0 + CT_Y + R + 1 < S);
            const bool xin = (T.tx0 - 2 - R >= 0) && (T.tx0 + CT_X + 1 + R < S);
            bool did_p3 = false;
            if (blocked && emu_tiled_radius(R)) {
                // k_canny_pipe writes only what lies inside the image (stp_cgeo): poison the LDS stand-ins so that a
                // read of anything it did not write reaches the result as a NaN
                std::fill(sV.begin(), sV.end(), std::numeric_limits<float>::quiet_NaN());
                std::fill(sS.begin(), sS.end(), std::numeric_limits<double>::quiet_NaN());
                std::fill(sM.begin(), sM.end(), std::numeric_limits<float>::quiet_NaN());
            }
            if (blocked == 2 && emu_tiled_radius(R)) {     // k_canny_f32: f32 phases, certified classes, exact resolver
                uint8_t* cm = mir ? sCm.data() : nullptr;
                switch (R) {
                    case 4: emu_tile_c32<4>(T, w, gray, gmax, sB.data(), sV.data(), sM.data(), sC.data(), xin, yin, cm); break;
                    case 6: emu_tile_c32<6>(T, w, gray, gmax, sB.data(), sV.data(), sM.data(), sC.data(), xin, yin, cm); break;
                    case 8: emu_tile_c32<8>(T, w, gray, gmax, sB.data(), sV.data(), sM.data(), sC.data(), xin, yin, cm); break;
                    case 10: emu_tile_c32<10>(T, w, gray, gmax, sB.data(), sV.data(), sM.data(), sC.data(), xin, yin, cm); break;
                    default: emu_tile_c32<12>(T, w, gray, gmax, sB.data(), sV.data(), sM.data(), sC.data(), xin, yin, cm); break;
                }
                canny_p5(0, 1, T, sC.data(), low, high);
                if (mir)                                   // the transposed words: rows tx0 .., half word (ty0 / 32) & 1 of word ty0 / 64
                    for (int xx = 0; xx < CT_X; xx++)
                        for (int yy = 0; yy < CT_Y; yy++) {
                            const int row = tx0 + xx, col = ty0 + yy, cls = sCm[xx * CT_Y + yy];
                            if (row >= S || col >= S || !cls) continue;
                            low[STP_CLS(row, col >> 6)] |= 1ull << (col & 63);
                            if (cls == 2) high[STP_CLS(row, col >> 6)] |= 1ull << (col & 63);
                        }
                continue;
            }
            if (blocked && emu_tiled_radius(R)) {          // the device path of the tiled radii (k_canny_pipe<R>)
                switch (R) {
                    case 4: emu_tile_pipe<4>(T, w, gray, sB.data(), sV.data(), sS.data(), sM.data(), fd, xin, yin); break;
                    case 6: emu_tile_pipe<6>(T, w, gray, sB.data(), sV.data(), sS.data(), sM.data(), fd, xin, yin); break;
                    case 8: emu_tile_pipe<8>(T, w, gray, sB.data(), sV.data(), sS.data(), sM.data(), fd, xin, yin); break;
                    case 10: emu_tile_pipe<10>(T, w, gray, sB.data(), sV.data(), sS.data(), sM.data(), fd, xin, yin); break;
                    default: emu_tile_pipe<12>(T, w, gray, sB.data(), sV.data(), sS.data(), sM.data(), fd, xin, yin); break;
                }
                did_p3 = true;
            } else {
                canny_p1(0, 1, T, R, w, sG.data(), sV.data());
                canny_p2(0, 1, T, R, w, sV.data(), sB.data(), sS.data());
            }
            if (!did_p3) canny_p3(0, 1, T, sS.data(), sM.data());
            canny_p4(0, 1, T, sS.data(), sM.data(), sC.data());
            canny_p5(0, 1, T, sC.data(), low, high);
        }
}

void emu_canny(const float* gray, int S, int R, const double* w, stp_u64* low, stp_u64* high)
{
    emu_canny2(gray, S, R, w, low, high, 1);
}
// the f32 intermediates of every pixel (five STP_PITCH x STP_PITCH float images: S, isobel, jsobel, magnitude, scale g)
void emu_canny_f32_dump(const float* gray, int S, int R, const double* w, float* dS, float* dI, float* dJ, float* dM, float* dG,
                        float* dE /* the tile's E_G in units of u g */)
{
    std::vector<stp_u64> low(STP_FRAME_MAX * STP_NW), high(STP_FRAME_MAX * STP_NW);
    g_c32_dS = dS; g_c32_dI = dI; g_c32_dJ = dJ; g_c32_dM = dM; g_c32_dG = dG; g_c32_dE = dE;
    emu_canny2(gray, S, R, w, low.data(), high.data(), 2);
    g_c32_dS = g_c32_dI = g_c32_dJ = g_c32_dM = g_c32_dG = g_c32_dE = nullptr;
}
void emu_canny_f32(const float* gray, int S, int R, const double* w, stp_u64* low, stp_u64* high, long long* counts /* candidates, uncertain */)
{
    g_c32_uncertain = 0; g_c32_candidates = 0;
    emu_canny2(gray, S, R, w, low, high, 2);
    if (counts) { counts[0] = g_c32_candidates; counts[1] = g_c32_uncertain; }
}
// ... with `mirror` (the image must be symmetric): the tiles below the diagonal from the transposes of the tiles above it
void emu_canny_f32_sym(const float* gray, int S, int R, const double* w, stp_u64* low, stp_u64* high, long long* counts)
{
    g_c32_uncertain = 0; g_c32_candidates = 0;
    emu_canny2(gray, S, R, w, low, high, 3);
    if (counts) { counts[0] = g_c32_candidates; counts[1] = g_c32_uncertain; }
}

// Round 6 invariant: every class half word k_lines reads (rows < S, words < ceil(S / 64)) is written by exactly one source -- a
// computed tile of k_canny_f32 (directly, or as the transpose of a tile above the diagonal) or the loader's patch from the next
// frame (stp_reuse_mask).  The writers are restated here from the kernel's rules; returns the number of (row, half word) cells that
// are written by nobody or claimed twice.
long long emu_class_word_cover(int S, int shift, int R, int mirror, int next_in_launch)
{
    const stp_reuse U = stp_reuse_of(shift, S, R, next_in_launch != 0);
    const int nwu = (S + 63) >> 6;
    std::vector<int> cnt((size_t)STP_FRAME_MAX * STP_NW * 2, 0);
    auto put = [&](int r, int w, int h) { if (r < S) cnt[((size_t)r * STP_NW + w) * 2 + h]++; };
    for (int ty = 0; ty * CT_Y < S; ty++)
        for (int tx = 0; tx * CT_X < S; tx++) {
            if (mirror && ty >= 2 * tx + 2) continue;                       // strictly below the diagonal: never computed
            if (stp_reuse_tile(U, ty, tx)) continue;                         // inside the block shared with the next frame
            for (int r = ty * CT_Y; r < ty * CT_Y + CT_Y; r++) { put(r, tx, 0); put(r, tx, 1); }
            if (mirror && tx > (ty >> 1)) {                                  // its transpose: rows tx0 .., word ty / 2, half ty & 1
                const bool solo = (ty & 1) == 0 && ty * CT_Y + CT_Y >= S;
                for (int r = tx * CT_X; r < tx * CT_X + CT_X; r++) {
                    put(r, ty >> 1, ty & 1);
                    if (solo) put(r, ty >> 1, 1);
                }
            }
        }
    long long bad = 0;
    for (int r = 0; r < S; r++)
        for (int w = 0; w < nwu; w++) {
            const stp_u64 m = stp_reuse_mask(U, mirror, r, w);
            for (int h = 0; h < 2; h++) {
                const bool patched = ((m >> (32 * h)) & 0xFFFFFFFFull) != 0;
                if (patched && ((m >> (32 * h)) & 0xFFFFFFFFull) != 0xFFFFFFFFull) bad++;        // a half word is patched whole or not at all
                if (cnt[((size_t)r * STP_NW + w) * 2 + h] + (patched ? 1 : 0) != 1) bad++;
                if (patched && !(r - shift >= STP_REUSE_MARGIN(R) && r - shift < S - shift - STP_REUSE_MARGIN(R))) bad++;   // source rows inside the shared block's interior
            }
        }
    return bad;
}

// Round 6 invariant: every grey value and every grey cell a COMPUTED Canny tile reads -- its window grown by R + 2 pixels, the
// 8 x 16 cells overlapping it -- lies in a grey tile k_gray_c3 writes (not stp_gray_tile_unread).  Returns the number of
// (Canny tile, grey tile) pairs that violate it; *skipped = grey tiles inside the image that are not written.
long long emu_gray_reader_cover(int S, int shift, int R, int next_in_launch, int* skipped)
{
    const stp_reuse U = stp_reuse_of(shift, S, R, next_in_launch != 0);
    long long bad = 0;
    int nskip = 0;
    for (int gy = 0; gy * GT_Y < S; gy++)
        for (int gx = 0; gx * GT_X < S; gx++) nskip += stp_gray_tile_unread(gy, gx, S, 1, U) ? 1 : 0;
    for (int ty = 0; ty * CT_Y < S; ty++)
        for (int tx = 0; tx * CT_X < S; tx++) {
            if (!stp_canny_tile_computed(ty, tx, S, 1, U)) continue;
            const int wy0 = std::max(ty * CT_Y - R - 2, 0), wy1 = std::min(ty * CT_Y + CT_Y + R + 2, S);
            const int wx0 = std::max(tx * CT_X - R - 2, 0), wx1 = std::min(tx * CT_X + CT_X + R + 2, S);
            const int y0 = wy0 / GC_CY * GC_CY, y1 = std::min(((wy1 - 1) / GC_CY + 1) * GC_CY, S);       // the cells' extent
            const int x0 = wx0 / GC_CX * GC_CX, x1 = std::min(((wx1 - 1) / GC_CX + 1) * GC_CX, S);
            for (int gy = y0 / GT_Y; gy * GT_Y < y1; gy++)
                for (int gx = x0 / GT_X; gx * GT_X < x1; gx++)
                    if (stp_gray_tile_unread(gy, gx, S, 1, U)) bad++;
        }
    if (skipped) *skipped = nskip;
    return bad;
}

struct emu_rec { int32_t ud, x, y, w, h; double total; };

// mirror of k_lines; dbg: E, V, T1, T2 bit matrices (each 400*7 words); cols: t, end, ud (3 x 400)
int emu_lines(const stp_u64* low, const stp_u64* high, const double* band, int W, int hw, int64_t st,
              const int16_t* nz, int S, int minH, int maxW, stp_u64* dbg, int16_t* cols, emu_rec* recs, int cap,
              int* sweeps_out)
{
    const int BW = STP_FRAME_MAX * STP_NW;
    std::vector<stp_u64> buf0(BW), buf1(BW), buf2(BW);
    std::vector<int16_t> colT(400), colEnd(400), colUd(400), cnt(400), minr(400), maxr(400), cidx(400), clen(400), xs(408);
    std::vector<stp_lrec> lrec(STP_RCAP);
    std::vector<double> rs(400);
    lines_load(0, 1, S, low, high, buf0.data(), buf1.data());
    int sweeps = 0;
    {   // k_lines' register form: every item keeps its rows across sweeps (items run in lane order here)
        const int nitem = ((S + STP_HYST_STRIP - 1) / STP_HYST_STRIP) * STP_NW;
        std::vector<stp_hyst_item> items(nitem);
        std::vector<uint16_t> edge(nitem + 1, 0);       // the items' published edge bits; slot nitem = beyond the image
        for (int i = 0; i < nitem; i++) hyst_item_load(i, S, buf0.data(), buf1.data(), &items[i], edge.data());
        for (;;) {
            int ch = 0;
            for (int i = 0; i < nitem; i++) ch |= hyst_item_sweep(S, &items[i], buf0.data(), buf1.data(), edge.data());
            if (!ch) break;
            sweeps++;
        }
    }
    if (sweeps_out) *sweeps_out = sweeps;
    lines_vline(0, 1, S, buf1.data(), buf2.data());
    lines_v3(0, 1, S, buf2.data(), buf0.data());
    lines_block_cols(0, 1, S, minH, buf2.data(), buf0.data(), colT.data(), colEnd.data(), colUd.data());
    memcpy(dbg, buf1.data(), BW * 8);
    memcpy(dbg + BW, buf2.data(), BW * 8);
    for (int i = 0; i < S; i++) { cols[i] = colT[i]; cols[400 + i] = colEnd[i]; cols[800 + i] = colUd[i]; }
    int nrec = 0;
    for (int ud = 1; ud <= 2; ud++) {
        lines_zero(0, 1, S * STP_NW, buf0.data());
        lines_paint(0, 1, S, ud, colEnd.data(), colUd.data(), buf0.data());
        lines_refine(0, 1, S, buf1.data(), buf2.data(), buf0.data());
        lines_colstat_cols(0, 1, S, buf0.data(), cnt.data(), minr.data(), maxr.data());
        memcpy(dbg + (size_t)(1 + ud) * BW, buf0.data(), BW * 8);
        int nrow = 0, wcnt[8] = {0};
        lines_cols_count(0, 1, S, cnt.data(), wcnt);
        lines_cols_place(0, 1, S, cnt.data(), wcnt, cidx.data(), clen.data(), &nrow);
        nrec = lines_group_pairs(S, ud, maxW, nrow, minr.data(), maxr.data(), cidx.data(), clen.data(), xs.data(),
                                 lrec.data(), nrec, STP_RCAP);
    }
    int nst = nrec < STP_RCAP ? nrec : STP_RCAP;
    for (int k = 0; k < nst && k < cap; k++) {
        stp_lrec rc = lrec[k];
        lines_rowsum(0, 1, S, band, W, hw, st, nz, rc, rs.data());
        double tot = 0.0;
        for (int i = 0; i < rc.h; i++) tot += rs[i];
        recs[k].ud = rc.ud; recs[k].x = rc.x; recs[k].y = rc.y; recs[k].w = rc.w; recs[k].h = rc.h; recs[k].total = tot;
    }
    return nrec;
}

}  // extern "C"

// Certification of the fused Gaussian sum (stp_gauss_fma.h): over `n` random windows of float values in [lo, 1]
// the float rounding of the fused sum must equal that of the reference's order whenever the near-boundary test does
// not flag it.  Returns the number of violations (must be 0); *flagged = how many outputs the test sends to the
// exact order; *max_ulp_diff = largest distance between the two f64 sums.
extern "C" long long emu_certify_fma(const double* w, int R, long long n, unsigned long long seed, double lo, long long* flagged,
                                     double* max_ulp_diff)
{
    long long bad = 0, nf = 0;
    double worst = 0.0;
    unsigned long long s = seed * 0x9E3779B97F4A7C15ull + 1;
    std::vector<double> win(2 * R + 1);
    for (long long it = 0; it < n; it++) {
        for (int k = 0; k <= 2 * R; k++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const float f = (float)(lo + (1.0 - lo) * (double)(s >> 40) / 16777216.0);
            win[k] = ((s & 0xF00) == 0) ? 0.0 : (double)f;          // some zero taps (image border)
        }
        double a = win[R] * w[R], e = win[R] * w[R];
        for (int k = R; k >= 1; k--) {
            a = fma(win[R - k] + win[R + k], w[R - k], a);
            e += (win[R - k] + win[R + k]) * w[R - k];
        }
        long long ia, ie;
        memcpy(&ia, &a, 8); memcpy(&ie, &e, 8);
        worst = std::max(worst, (double)std::llabs(ia - ie));
        if ((unsigned long long)STP_FMA_NEAR >= 0x10000000ull || stp_fma_near_word(a) < 16u * (unsigned)STP_FMA_NEAR) { nf++; continue; }
        if ((float)a != (float)e) bad++;
    }
    if (flagged) *flagged = nf;
    if (max_ulp_diff) *max_ulp_diff = worst;
    return bad;
}

// whether the multiply + 2 FMA division by the interior bleed-over constant is exact for every float
extern "C" int emu_fastdiv_ok(const double* w, int R, double* c_out)
{
    const int Sx = 4 * R + 8;
    const double V = stp_bleed_v(2 * R + 2, Sx, R, w);
    const double c = stp_bleed_h(V, 2 * R + 2, Sx, R, w) + DBL_EPSILON, rc = 1.0 / c;
    if (c_out) *c_out = c;
    for (uint32_t m = 0; m < (1u << 23); m++) {
        uint32_t bits = 0x3F800000u | m; float f; memcpy(&f, &bits, 4);
        const double q = stp_div_const((double)f, c, rc), t = (double)f / c;
        if (memcmp(&q, &t, 8) != 0) return 0;
    }
    return 1;
}

// the tile-wide error budgets of k_canny_f32 for the given weights: out[9] = E_G, E_M, T0 (units of u g) for interior
// tiles, tiles cut on one side, tiles of images narrower than 2R + 1; rho[3] = the per-pass constants behind them
extern "C" void emu_c32_budget(const double* w, int R, float* out, double* rho)
{
    stp_w32 W;
    c32_budget(w, R, &W);
    for (int t = 0; t < 3; t++) for (int k = 0; k < 3; k++) out[3 * t + k] = W.eu[t][k];
    rho[0] = c32_rho(w, R, -R, R);
    double one = rho[0], any = rho[0];
    for (int lo = -R; lo <= 0; lo++)
        for (int hi = 0; hi <= R; hi++) {
            const double r = c32_rho(w, R, lo, hi);
            any = r > any ? r : any;
            if (lo == -R || hi == R) one = r > one ? r : one;
        }
    rho[1] = one; rho[2] = any;
}
