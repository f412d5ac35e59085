// libstripenn_hip.so -- gfx950 kernels + C ABI (include/stripenn_hip.h).
// Build: stripenn_amd/csrc/Makefile (hipcc --offload-arch=gfx950 -O3 -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <new>
#include <mutex>
#include <dlfcn.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include "../../include/stripenn_hip.h"
#include "stp_phases.h"
#include "stp_canny32.h"
#ifndef STP_ABLATE_C32
#define STP_ABLATE_C32 0   /* 1..4: timing-only builds of k_canny_f32 (make ablate32), never shipped */
#endif
#ifndef STP_GRAY_LEVRUNS
#define STP_GRAY_LEVRUNS 1    /* workgroups per tile of k_gray_c3, each walking a run of the maxpixel levels (measured: 1 -> 17.0, 2 -> 17.5, 5 -> 19.9 ms per genome step) */
#endif
#ifndef STP_PV_NT
#define STP_PV_NT 256      /* threads per stripe of k_pvalue / k_stripiness */
#endif
#ifndef STP_SC_NT
#define STP_SC_NT 128
#endif
#include "stp_score.h"
#include "stp_select.h"
#include "stp_tsv.h"

// ============================================================================================
// device kernels
// ============================================================================================

// (frame preparation -- zero-column removal and medpixel of every frame -- is k_frame_prep in stp_score.h)

// K-A: image build + brightness + mean blur + grey for all brightness levels of one tile.
// Lane exchanges inside a row of 16 lanes without LDS traffic (DPP): the value of lane ^ 8, ^ 4, ^ 2, ^ 1.  Every lane of the wave
// must be active.  Used by k_lines' bit transposes and by the 16-lane cell reductions of k_gray_c3.
__device__ __forceinline__ unsigned wt_dpp_xor8(unsigned w) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0x128, 0xF, 0xF, false); }   // row_ror:8
__device__ __forceinline__ unsigned wt_dpp_xor4(unsigned w)
{
    int p = __builtin_amdgcn_update_dpp(0, (int)w, 0x104, 0xF, 0x5, false);         // row_shl:4 -> banks 0, 2 (lane bit 2 clear) read lane + 4
    return (unsigned)__builtin_amdgcn_update_dpp(p, (int)w, 0x114, 0xF, 0xA, false);  // row_shr:4 -> banks 1, 3 read lane - 4
}
__device__ __forceinline__ unsigned wt_dpp_xor2(unsigned w) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0x4E, 0xF, 0xF, false); }    // quad_perm:[2,3,0,1]
__device__ __forceinline__ unsigned wt_dpp_xor1(unsigned w) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xF, 0xF, false); }    // quad_perm:[1,0,3,2]

// one tile of one (frame, level): every brightness image, every operation of the reference (k_gray, k_gray_fill)
template <int AMAX>
__device__ __forceinline__ void gray_tile_exact(const double* __restrict__ band, int W, int hw,
                                               const int32_t* __restrict__ fstart, const int32_t* __restrict__ fS,
                                               const int16_t* __restrict__ fnz, int f0,
                                               const double* __restrict__ Mlev, int nlev,
                                               const double* __restrict__ bvals, int nb, int a,
                                               float* __restrict__ gray, float2* __restrict__ cells, int tile, int lev, int fl)
{
    // AMAX = 1 (bfilter 3): 20 KB of LDS -> 7 workgroups per CU; the generic instance is sized for bfilter <= 7
    constexpr int SG_N = (GT_Y + 2 * AMAX) * (GT_X + 2 * AMAX);
    constexpr int SADJ_W = 4 * (GT_X + 2);               // bfilter 3: one product row per wave
    __shared__ double sg[SG_N];
    __shared__ double sadj[AMAX == 1 ? SADJ_W : SG_N];
    const int f = f0 + fl;
    const int S = fS[f];
    if (S == 0) return;
    const int tpr = (STP_FRAME_MAX + GT_X - 1) / GT_X;
    stp_tile T;
    T.S = S; T.ty0 = (tile / tpr) * GT_Y; T.tx0 = (tile % tpr) * GT_X;
    if (T.ty0 >= S || T.tx0 >= S) return;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (AMAX == 1) {
        // bfilter 3 (default): compaction map of the tile's rows / columns staged in LDS, then all band
        // loads of a lane issued back to back (fixed trip count, no dependent global index loads)
        __shared__ int16_t s_ny[GT_Y + 2], s_nx[GT_X + 2];
        const int16_t* nzf = fnz + (size_t)f * STP_FRAME_MAX;
        if (tid < GT_Y + 2) s_ny[tid] = nzf[stp_refl101(min(T.ty0 + tid - 1, S), S)];
        if (tid >= 64 && tid < 64 + GT_X + 2) s_nx[tid - 64] = nzf[stp_refl101(min(T.tx0 + (tid - 64) - 1, S), S)];
        __syncthreads();
        constexpr int HH = GT_Y + 2, WW = GT_X + 2, N = HH * WW, IT = (N + 255) / 256;
        const int64_t st = fstart[f];
        const double M = Mlev[lev];
        double v[IT];
#pragma unroll
        for (int k = 0; k < IT; k++) {
            const int i = min(tid + k * 256, N - 1);
            const int yy = i / WW, xx = i - yy * WW;
            const int oy = s_ny[yy], ox = s_nx[xx];
            v[k] = band[(st + oy) * (int64_t)W + (ox - oy + hw)];
        }
#pragma unroll
        for (int k = 0; k < IT; k++) {
            const int i = tid + k * 256;
            if (i < N) {
                const int yy = i / WW, xx = i - yy * WW;
                const bool in = (T.ty0 + yy - 1 < S + 1) && (T.tx0 + xx - 1 < S + 1);
                double d = v[k];
                if (d != d) d = 0.0;
                sg[i] = in ? stp_gplane_px(d, M) : 0.0;
            }
        }
    } else {
        gray_p0(tid, nt, band, W, hw, (int64_t)fstart[f], fnz + (size_t)f * STP_FRAME_MAX, T, a, Mlev[lev], sg);
    }
    __syncthreads();
    if (AMAX == 1) {
        // wave-strip form: no workgroup barrier in the brightness loop (LDS ops of one wave are in order)
        const int lane = tid & 63, strip = tid >> 6;
        double* srow = sadj + strip * (GT_X + 2);
        for (int bi = 0; bi < nb; bi++) {
            const size_t img = ((size_t)fl * nlev + lev) * nb + bi;
            float* gimg = gray + img * (size_t)(STP_PITCH * STP_PITCH);
            stp_gray_lane st;
            st.vmin = 0x7F800000u; st.vmax = 0u;                // +inf / +0: a cell without pixels unites to nothing
#pragma unroll
            for (int c = 0; c < 3; c++) st.w0[c] = st.w1[c] = 0.0;
#pragma unroll
            for (int r = 0; r < GS_ROWS + 2; r++) {
                gray_wrow_put(lane, strip, r, bvals[bi], sg, srow);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                gray_wrow_get(lane, strip, r, T, srow, &st, gimg);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the reads above precede the next row's writes
                __builtin_amdgcn_wave_barrier();
            }
            unsigned vmn = st.vmin, vmx = st.vmax;
            // min / max of the strip's grey values per 16-column cell (see STP_FLAT_RANGE)
#pragma unroll
            for (int o = 1; o < GC_CX; o <<= 1) {
                vmn = min(vmn, (unsigned)__shfl_xor((int)vmn, o));
                vmx = max(vmx, (unsigned)__shfl_xor((int)vmx, o));
            }
            const int crow = T.ty0 / GC_CY + strip, ccol = (T.tx0 + lane) / GC_CX;
            if (cells && (lane & (GC_CX - 1)) == 0 && crow < GC_ROWS && ccol < GC_COLS)
                cells[(img * GC_ROWS + crow) * GC_COLS + ccol] = make_float2(__uint_as_float(vmn), __uint_as_float(vmx));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    for (int bi = 0; bi < nb; bi++) {
        gray_p1(tid, nt, a, bvals[bi], sg, sadj);
        __syncthreads();
        const size_t img = ((size_t)fl * nlev + lev) * nb + bi;
        gray_p2(tid, nt, T, a, sadj, gray + img * (size_t)(STP_PITCH * STP_PITCH));
        __syncthreads();
    }
}
template <int AMAX>
__global__ __launch_bounds__(256) void k_gray(const double* __restrict__ band, int W, int hw,
                                               const int32_t* __restrict__ fstart, const int32_t* __restrict__ fS,
                                               const int16_t* __restrict__ fnz, int f0,
                                               const double* __restrict__ Mlev, int nlev,
                                               const double* __restrict__ bvals, int nb, int a,
                                               float* __restrict__ gray, float2* __restrict__ cells)
{
    gray_tile_exact<AMAX>(band, W, hw, fstart, fS, fnz, f0, Mlev, nlev, bvals, nb, a, gray, cells, blockIdx.x, blockIdx.y, blockIdx.z);
}
// The grey tiles k_gray_c3 skipped (stp_gray_tile_unread), for the (frame, level) pairs k_canny_f32 has marked: their tiles below the
// diagonal go to the exact kernel, which reads its grey windows wherever they lie.  Grid (the 42 tiles on or below the line
// gy = 2 gx + 1, where every unread tile lies, pairs); a pair without a mark -- every pair of ordinary data -- costs its
// workgroups one byte load.
__global__ __launch_bounds__(256) void k_gray_fill(const double* __restrict__ band, int W, int hw,
                                                    const int32_t* __restrict__ fstart, const int32_t* __restrict__ fS,
                                                    const int16_t* __restrict__ fnz, int f0,
                                                    const double* __restrict__ Mlev, int nlev,
                                                    const double* __restrict__ bvals, int nb,
                                                    float* __restrict__ gray, const uint8_t* __restrict__ need,
                                                    const int32_t* __restrict__ fshift, int gauss_radius, int nframes)
{
    const int pair = blockIdx.y;
    if (!need[pair]) return;
    constexpr int tpr = (STP_FRAME_MAX + GT_X - 1) / GT_X, tpc = (STP_FRAME_MAX + GT_Y - 1) / GT_Y;
    int k = blockIdx.x, tile = -1;                    // the k-th tile with gy >= 2 gx + 1, row by row
    for (int t = 0; t < tpr * tpc; t++)
        if (t / tpr >= 2 * (t % tpr) + 1 && k-- == 0) { tile = t; break; }
    if (tile < 0) return;
    const int fl = pair / nlev, S = fS[f0 + fl];
    stp_reuse U; U.lo = 1; U.hi = 0; U.shift = -1;
    if (fshift != nullptr) U = stp_reuse_of(fshift[f0 + fl], S, gauss_radius, fl + 1 < nframes);
    if (!stp_gray_tile_unread(tile / tpr, tile % tpr, S, 1, U)) return;               // k_gray_c3 has written it
    gray_tile_exact<1>(band, W, hw, fstart, fS, fnz, f0, Mlev, nlev, bvals, nb, 1, gray, nullptr, tile, pair % nlev, fl);
}
#define STP_GRAY_FILL_TILES 42      /* tiles with gy >= 2 gx + 1 in the 13 x 7 grid (checked by a static_assert on the host side) */

// K-A for the 3 x 3 mean filter, certified (stp_phases.h, "certified grey"): the same tile / strip geometry as k_gray<1> -- one
// workgroup per (tile, level, frame), the g~ plane of the tile in LDS, each wave an 8-row strip with lane = column and no
// workgroup barrier in the brightness loop -- but per image pixel one min, four additions (row sums shared by the three
// outputs below each other), one product and the conversion instead of two compares, two products and nine additions, and
// per contact pixel a subtraction and a product instead of two divisions.  A lane whose strip holds an output within
// STP_GRAY_NEAR ulp of a float rounding boundary recomputes its eight outputs from the band in the reference's operations.
// Grey images and cell minima / maxima are bit for bit those of k_gray<1> (STP_GRAY=exact selects that kernel).
// the rare path: one lane's strip of eight outputs from the band in the reference's operations, stored; returns the
// smallest / largest bit pattern stored (not inlined: it must not cost the common path registers)
// asym_flag (round 6, image symmetry): in a symmetric contact matrix the mirror pixel (column, row) sums the SAME nine terms in
// column-major order; only a sum this close to a float rounding boundary can round differently there.  The strip's outputs
// are therefore also formed in that order, and an image in which one of them differs is reported: k_canny_f32 then does
// not hand the transposed class words of that image to the tiles below the diagonal (see there).
__device__ __noinline__ static uint2 gray_c3_redo(const double* __restrict__ band, int W, int hw, int64_t st, const int16_t* s_ny,
                                                  const int16_t* s_nx, int yy0, int xx, int nrows, double M, double b, float* gcol,
                                                  uint8_t* asym_flag)
{   // (yy0 + j, xx): window coordinates of the output pixels' centres in the (GT_Y + 2) x (GT_X + 2) plane
    unsigned vmn = 0x7F800000u, vmx = 0u;
#pragma unroll 1
    for (int j = 0; j < nrows; j++) {
        const double k = (1.0 - 0.0) / (b - 0.0), kv = 1.0 / 9.0;         // as stp_gray_exact9, one contact value at a time
        double acc = 0.0;
#pragma unroll 1
        for (int i = 0; i < 9; i++) {
            const int oy = s_ny[yy0 + j - 1 + i / 3], ox = s_nx[xx - 1 + i % 3];
            double d = band[(st + oy) * (int64_t)W + (ox - oy + hw)];
            if (d != d) d = 0.0;
            acc = acc + kv * stp_bright_px(stp_gplane_px(d, M), b, k);
        }
        if (acc < 0.0) acc = 0.0;
        if (acc > 1.0) acc = 1.0;
        const float v = stp_gray_rgb((float)acc);
        gcol[j * STP_PITCH] = v;
        const unsigned bits = __float_as_uint(v);
        vmn = min(vmn, bits); vmx = max(vmx, bits);
        if (asym_flag != nullptr) {
            double acct = 0.0;
#pragma unroll 1
            for (int i = 0; i < 9; i++) {                                  // the mirror pixel's order: window columns outermost
                const int oy = s_ny[yy0 + j - 1 + i % 3], ox = s_nx[xx - 1 + i / 3];
                double d = band[(st + oy) * (int64_t)W + (ox - oy + hw)];
                if (d != d) d = 0.0;
                acct = acct + kv * stp_bright_px(stp_gplane_px(d, M), b, k);
            }
            if (acct < 0.0) acct = 0.0;
            if (acct > 1.0) acct = 1.0;
            if (__float_as_uint(stp_gray_rgb((float)acct)) != bits) *asym_flag = 1;
        }
    }
    return make_uint2(vmn, vmx);
}
__global__ __launch_bounds__(256) void k_gray_c3(const double* __restrict__ band, int W, int hw,
                                                  const int32_t* __restrict__ fstart, const int32_t* __restrict__ fS,
                                                  const int16_t* __restrict__ fnz, int f0,
                                                  const double* __restrict__ Mlev, int nlev,
                                                  const double* __restrict__ bvals, int nb,
                                                  float* __restrict__ gray, float2* __restrict__ cells,
                                                  uint8_t* __restrict__ asym /* per image, or null: see gray_c3_redo */, int skip_dead,
                                                  const int32_t* __restrict__ fshift /* frame overlap, or null */, int gauss_radius, int nframes)
{
    constexpr int HH = GT_Y + 2, WW = GT_X + 2, N = HH * WW, IT = (N + 255) / 256;
    __shared__ double sg[N], sd[N];
    __shared__ int16_t s_ny[HH], s_nx[WW];
    constexpr int NCB = 16;
    __shared__ double s_cb[NCB];                 // (1 / b) (1 / 9) of every brightness level: one division per workgroup and level
                                                 // instead of one per image and lane (round 5: the division sequence, with its
                                                 // quarter-rate reciprocal, was a tenth of the image loop's cycles)
    const int fl = blockIdx.z, f = f0 + fl;
    const int S = fS[f];
    if (S == 0) return;
    const int tpr = (STP_FRAME_MAX + GT_X - 1) / GT_X;
    stp_tile T;
    T.S = S; T.ty0 = (blockIdx.x / tpr) * GT_Y; T.tx0 = (blockIdx.x % tpr) * GT_X;
    if (T.ty0 >= S || T.tx0 >= S) return;
    if (skip_dead) {                                 // no reader among the computed Canny tiles (stp_phases.h, "grey tiles nobody reads")
        stp_reuse U; U.lo = 1; U.hi = 0; U.shift = -1;
        if (fshift != nullptr) U = stp_reuse_of(fshift[f], S, gauss_radius, fl + 1 < nframes);
        if (stp_gray_tile_unread(blockIdx.x / tpr, blockIdx.x % tpr, S, 1, U)) return;
#if defined(STP_ABLATE_GRAY_LOWER)      /* timing-only build: what not writing ANY grey tile below the diagonal would save (results wrong) */
        if ((int)(blockIdx.x / tpr) >= 2 * (int)(blockIdx.x % tpr) + 2) return;
#endif
    }
#if defined(STP_ABLATE_GRAY_TX6)            /* timing-only build: what the tiles of the last tile column (16 of 64 columns in the image) cost */
    if ((int)(blockIdx.x % tpr) == 6) return;
#endif
    const int tid = threadIdx.x;
    if (tid >= 128 && tid < 128 + NCB && tid - 128 < nb) s_cb[tid - 128] = stp_gray_cb(bvals[tid - 128]);   // (read after the barriers below)
    const int16_t* nzf = fnz + (size_t)f * STP_FRAME_MAX;
    if (tid < HH) s_ny[tid] = nzf[stp_refl101(min(T.ty0 + tid - 1, S), S)];
    if (tid >= 64 && tid < 64 + WW) s_nx[tid - 64] = nzf[stp_refl101(min(T.tx0 + (tid - 64) - 1, S), S)];
    __syncthreads();
    const int64_t st = fstart[f];
    // the tile's contact values are read ONCE for all maxpixel levels (one workgroup walks the levels: the two dependent
    // global round trips of this prologue are paid once per tile, not once per level)
    // (parked in LDS, each thread its own elements: 36 KB per workgroup -- four per CU, as the registers allow anyway)
    {
        double v[IT];
#pragma unroll
        for (int k = 0; k < IT; k++) {
            const int i = min(tid + k * 256, N - 1);
            const int yy = i / WW, xx = i - yy * WW;
            const int oy = s_ny[yy], ox = s_nx[xx];
            v[k] = __builtin_nontemporal_load(&band[(st + oy) * (int64_t)W + (ox - oy + hw)]);      // (read once per tile: non-temporal, -0.3..0.8 ms per step)
        }
#pragma unroll
        for (int k = 0; k < IT; k++) {
            const int i = tid + k * 256;
            if (i < N) sd[i] = (v[k] != v[k]) ? 0.0 : v[k];                  // nantozero (getStripe.py:809)
        }
    }
    // (round 6, measured and dropped: in the last tile column -- 16 image columns, one cell's width -- the first wave taking all four
    //  strips with 16 lanes each, the other waves idle in the image loop: bit-identical images, 10.7 ms per step against 10.5.  The 13
    //  tiles of that column cost 1.07 ms per step for 4 % of the pixels (timing-only build, profiles/r06_ablate_gray_last_column.txt),
    //  but not through the image loop's issue slots.)
    const int lane = tid & 63, strip = tid >> 6;
    const int x = T.tx0 + lane, y0 = T.ty0 + strip * GS_ROWS;
    const bool xin = x < S;
    const int nrows = min(GS_ROWS, S - y0);                                  // rows of the strip inside the image (wave-uniform)
    const double* g0 = sg + (strip * GS_ROWS) * WW + lane;                   // g~ of (strip row -1, columns x-1 .. x+1)
    // (gridDim.y workgroups may share a tile's levels -- STP_GRAY_LEVRUNS; one workgroup for all of them measured fastest)
    const int lev_lo = (int)((long long)nlev * blockIdx.y / gridDim.y), lev_hi = (int)((long long)nlev * (blockIdx.y + 1) / gridDim.y);
    for (int lev = lev_lo; lev < lev_hi; lev++) {
        const double M = Mlev[lev], rM = 1.0 / M;
        if (lev > lev_lo) __syncthreads();                                   // everybody is done with the previous level's plane
#pragma unroll
        for (int k = 0; k < IT; k++) {
            const int i = tid + k * 256;
            if (i < N) {
                const int yy = i / WW, xx = i - yy * WW;
                const bool in = (T.ty0 + yy - 1 < S + 1) && (T.tx0 + xx - 1 < S + 1);
                sg[i] = in ? stp_gplane_fast(sd[i], M, rM) : 0.0;
            }
        }
        __syncthreads();
        // (round 6, measured and dropped -- profiles/r06_ab_saturated_strips.txt: a wave whose whole 10 x 66 window holds g~ >= b writes
        //  ONE number, ((3 b) + (3 b) + (3 b)) c_b, into its 512 outputs; 15 % of the strip-images of the benchmark data qualify -- the
        //  darker brightness levels away from the diagonal -- but the kernel's time did not move (12.5-12.8 ms per step either way):
        //  those strips still store their 2 KB, and the 29 minima per level that find them cost what the skipped arithmetic saved)
        for (int bi = 0; bi < nb; bi++) {
            const size_t img = ((size_t)fl * nlev + lev) * nb + bi;
            float* gimg = gray + img * (size_t)(STP_PITCH * STP_PITCH) + (size_t)y0 * STP_PITCH + x;
            const double b = bvals[bi], cb = bi < NCB ? s_cb[bi] : stp_gray_cb(b);
            float out[GS_ROWS];
            unsigned far = 0xFFFFFFFFu;
            double rs0 = 0.0, rs1 = 0.0;
#pragma unroll
            for (int r = 0; r < GS_ROWS + 2; r++) {
                const double* g = g0 + r * WW;
                const double a0 = fmin(g[0], b), a1 = fmin(g[1], b), a2 = fmin(g[2], b);
                const double rs2 = (a0 + a1) + a2;
                if (r >= 2) {
                    const double blur = ((rs0 + rs1) + rs2) * cb;
#if !(defined(STP_ABLATE_GRAY) && STP_ABLATE_GRAY == 2)   /* ... no rounding-boundary test */
                    const unsigned nw = stp_near_word(blur, STP_GRAY_NEAR);
                    far = nw < far ? nw : far;
#endif
                    out[r - 2] = stp_gray_rgb((float)blur);
                }
                rs0 = rs1; rs1 = rs2;
            }
            unsigned vmn = 0x7F800000u, vmx = 0u;                            // +inf / +0: a cell without pixels unites to nothing
            if (xin && nrows > 0) {
                if (far < 16u * STP_GRAY_NEAR) {                              // ~1e-6 of the lanes: the reference's own operations
                    const uint2 mm = gray_c3_redo(band, W, hw, st, s_ny, s_nx, strip * GS_ROWS + 1, lane + 1, nrows, M, b, gimg,
                                                  asym ? asym + img : nullptr);
                    vmn = mm.x; vmx = mm.y;
                } else if (nrows == GS_ROWS) {                                // (whole strip inside the image: no test per row)
#pragma unroll
                    for (int j = 0; j < GS_ROWS; j++) {
#if defined(STP_ABLATE_GRAY_ST)        /* timing-only build: a quarter of the store instructions */
                        if (j % 4 == 0)
#endif
#if defined(STP_ABLATE_GRAY) && STP_ABLATE_GRAY == 1      /* timing-only builds (never shipped): no pixel stores ... */
                        if (out[j] == 12345.0f)
#endif
                        // (non-temporal, round 6: a launch writes 1.9 GB of grey values that the Canny kernel reads one launch later,
                        //  from HBM either way -- kept out of the caches they no longer evict what the other kernels share: 12.5 ->
                        //  10.5 ms per step, and 2.5 ms off the step: profiles/r06_ab_nt_stores.txt)
                        __builtin_nontemporal_store(out[j], &gimg[j * STP_PITCH]);
                        const unsigned bits = __float_as_uint(out[j]);       // grey values are >= +0: their bit patterns order like the values
#if !(defined(STP_ABLATE_GRAY) && STP_ABLATE_GRAY == 3)   /* ... no cell minima / maxima ... */
                        vmn = min(vmn, bits); vmx = max(vmx, bits);
#else
                        vmn = bits;
#endif
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < GS_ROWS; j++)
                        if (j < nrows) {
                            __builtin_nontemporal_store(out[j], &gimg[j * STP_PITCH]);
                            const unsigned bits = __float_as_uint(out[j]);
                            vmn = min(vmn, bits); vmx = max(vmx, bits);
                        }
                }
            }
            // min / max of the strip's grey values per 16-column cell (see STP_FLAT_RANGE)
            // (round 5: DPP row exchanges instead of four dependent ds_bpermute round trips per image -- all 64 lanes are active here)
            static_assert(GC_CX == 16, "a cell is one DPP row of 16 lanes");
            vmn = min(vmn, wt_dpp_xor1(vmn)); vmx = max(vmx, wt_dpp_xor1(vmx));
            vmn = min(vmn, wt_dpp_xor2(vmn)); vmx = max(vmx, wt_dpp_xor2(vmx));
            vmn = min(vmn, wt_dpp_xor4(vmn)); vmx = max(vmx, wt_dpp_xor4(vmx));
            vmn = min(vmn, wt_dpp_xor8(vmn)); vmx = max(vmx, wt_dpp_xor8(vmx));
            const int crow = T.ty0 / GC_CY + strip, ccol = (T.tx0 + lane) / GC_CX;
            if (cells && (lane & (GC_CX - 1)) == 0 && crow < GC_ROWS && ccol < GC_COLS)
                cells[(img * GC_ROWS + crow) * GC_COLS + ccol] = make_float2(__uint_as_float(vmn), __uint_as_float(vmx));
        }
    }
}

// NMS class of the tile's pixels and bit-plane packing.  Each wave owns 8 rows (lane = x, CT_X == 64):
// it first collects the pixels whose magnitude reaches the low threshold (the only ones that can get a
// class) into a wave-private LDS queue by ballot + popcount, then runs the interpolation test densely
// over the queue, 64 candidates at a time -- candidates are 7-25 % of the pixels but occur in every row,
// so a per-row test would keep all lanes of every wave busy with the expensive path.
__device__ __forceinline__ void canny_nms_pack(int tid, stp_tile T, const double* sS, const float* sM, uint16_t* sQ,
                                               stp_u64* sBits, stp_u64* __restrict__ low_img, stp_u64* __restrict__ high_img,
                                               int nms_rows = CT_Y /* tile rows that can hold an interior pixel */)
{
    const int lane = tid & 63, wv = tid >> 6;
    uint16_t* q = sQ + wv * 512;
    stp_u64* lowB = sBits;
    stp_u64* highB = sBits + CT_Y;
    if (lane < 8) { lowB[wv * 8 + lane] = 0; highB[wv * 8 + lane] = 0; }
    int n = 0;
    const int rows = min(8, nms_rows - wv * 8);       // wave-uniform: a strip below the image collects nothing
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int yy = wv * 8 + r;
        const int y = T.ty0 + yy, x = T.tx0 + lane;
        bool c = false;
        if (r < rows && y >= 1 && x >= 1 && x < T.S - 1) c = sM[(yy + 1) * (CT_X + 2) + lane + 1] >= (float)(0.1 - 1e-6);
        const stp_u64 m = __ballot(c);
        if (c) q[n + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)((yy << 6) | lane);
        n += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + lane;
        if (k < n) {
            const int e = q[k], yy = e >> 6, xx = e & 63;
            const int cls = ct_nms(sS, sM, T, T.ty0 + yy, T.tx0 + xx);
            if (cls >= 1) atomicOr(&lowB[yy], 1ull << xx);
            if (cls == 2) atomicOr(&highB[yy], 1ull << xx);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 8) {
        const int yy = wv * 8 + lane, y = T.ty0 + yy;
        if (y < T.S) {
            low_img[STP_CLS(y, T.tx0 >> 6)] = lowB[yy];
            high_img[STP_CLS(y, T.tx0 >> 6)] = highB[yy];
        }
    }
}

// k_canny_pipe: magnitudes in 2 x 2 blocks (ct_sobel_blk2) and, in the same walk, collection of the pixels whose
// magnitude reaches the low threshold (the only ones that can get a class) into ONE workgroup-wide LDS queue.  Per
// round a lane holds four magnitudes; which of them may be candidates at all is decided per block ROW and block COLUMN
// (two range tests each against the tile's candidate window, computed once per workgroup: tile interior and image
// interior), the four ballots share ONE LDS atomic of the wave's first lane, and a candidate's slot is the running
// popcount of the ballots before it.  Order in the queue is irrelevant: the classes are OR-ed into the bit rows.
struct stp_cwin { int y0, ny, x0, nx; };          // candidate window in magnitude-tile coordinates
__device__ __forceinline__ stp_cwin canny_cand_window(stp_tile T)
{
    stp_cwin C;                                   // pixel (Y, X) of the magnitude tile = image (ty0 + Y - 1, tx0 + X - 1)
    C.y0 = max(1, 2 - T.ty0); C.ny = min(CT_Y, T.S - 1 - T.ty0) - C.y0 + 1;
    C.x0 = max(1, 2 - T.tx0); C.nx = min(CT_X, T.S - 1 - T.tx0) - C.x0 + 1;
    if (C.ny < 0) C.ny = 0;
    if (C.nx < 0) C.nx = 0;
    return C;
}
__device__ __forceinline__ void canny_p3_collect(int tid, stp_cwin C, stp_p3walk W, const double* sS, float* sM, uint16_t* sQ,
                                                 int* sQn)
{
    const int lane = tid & 63;
    int r = W.r0, c = W.c0;
    const int rounds = (W.n + 255) >> 8;                    // workgroup-uniform
    const float thr = (float)(0.1 - 1e-6);
    for (int k = 0; k < rounds; k++) {
        const bool act = tid + 256 * k < W.n;
        float m[4] = {0.f, 0.f, 0.f, 0.f};
        int y = 0, x = 0;
        if (act) {
            y = 2 * r < W.nmh - 2 ? 2 * r : W.nmh - 2; x = 2 * c < W.nmw - 2 ? 2 * c : W.nmw - 2;
            ct_sobel_blk2(sS + W.soff + y * CT_SP + x, m);
            float* o = sM + W.moff + y * (CT_X + 2) + x;
            o[0] = m[0]; o[1] = m[1]; o[CT_X + 2] = m[2]; o[CT_X + 3] = m[3];
        }
        // a block shifted back at an odd extent repeats one row / column of its neighbour: not collected twice
        const int Y = y + W.my_lo, X = x + W.mx_lo;
        const bool r0 = act && (unsigned)(Y - C.y0) < (unsigned)C.ny && y == 2 * r, r1 = act && (unsigned)(Y + 1 - C.y0) < (unsigned)C.ny;
        const bool c0 = (unsigned)(X - C.x0) < (unsigned)C.nx && x == 2 * c, c1 = (unsigned)(X + 1 - C.x0) < (unsigned)C.nx;
        const bool q0 = r0 && c0 && m[0] >= thr, q1 = r0 && c1 && m[1] >= thr, q2 = r1 && c0 && m[2] >= thr, q3 = r1 && c1 && m[3] >= thr;
        const stp_u64 b0 = __ballot(q0), b1 = __ballot(q1), b2 = __ballot(q2), b3 = __ballot(q3);
        const int n0 = __popcll(b0), n1 = __popcll(b1), n2 = __popcll(b2), n3 = __popcll(b3);
        if (n0 + n1 + n2 + n3) {                             // wave-uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(sQn, n0 + n1 + n2 + n3);
            base = __shfl(base, 0);
            const stp_u64 lt = (1ull << lane) - 1ull;
            const int e = (Y - 1) * 64 + (X - 1);             // (row << 6 | column) of the block's first pixel in tile coordinates
                                                              // (a halo row / column gives -1: only the in-tile neighbours are stored)
            if (q0) sQ[base + __popcll(b0 & lt)] = (uint16_t)e;
            if (q1) sQ[base + n0 + __popcll(b1 & lt)] = (uint16_t)(e + 1);
            if (q2) sQ[base + n0 + n1 + __popcll(b2 & lt)] = (uint16_t)(e + 64);
            if (q3) sQ[base + n0 + n1 + n2 + __popcll(b3 & lt)] = (uint16_t)(e + 65);
        }
        c += W.dc; r += W.dr;
        if (c >= W.nbw) { c -= W.nbw; r++; }
    }
}
// the interpolation test over the queue, all threads of the workgroup, 256 candidates at a time
__device__ __forceinline__ void canny_nms_queue(int tid, stp_tile T, const double* sS, const float* sM, const uint16_t* sQ, int n,
                                                stp_u64* sBits)
{
    stp_u64* lowB = sBits;
    stp_u64* highB = sBits + CT_Y;
    for (int k = tid; k < n; k += 256) {
        const int e = sQ[k], yy = e >> 6, xx = e & 63;
        const int cls = ct_nms(sS, sM, T, T.ty0 + yy, T.tx0 + xx);
        if (cls >= 1) atomicOr(&lowB[yy], 1ull << xx);
        if (cls == 2) atomicOr(&highB[yy], 1ull << xx);
    }
}

#define CANNY_NMS_BYTES (2 * CT_Y * 8 + 4 * 512 * 2)   /* bit-rows + 4 wave queues of 512 u16 */
static __host__ __device__ size_t canny_smem_bytes(int R)
{
    const int GW = CT_X + 2 * R + 4, GH = CT_Y + 2 * R + 4, VH = CT_Y + 4;
    size_t fixed = (32 + 2 * VH + VH * CT_SP) * sizeof(double);
    size_t vsz = (size_t)VH * GW > (size_t)GW * CT_VP ? (size_t)VH * GW : (size_t)GW * CT_VP;
    size_t gv = ((size_t)GH * GW + vsz) * sizeof(float);
    size_t mc = (size_t)(CT_Y + 2) * (CT_X + 2) * sizeof(double) + CT_Y * CT_X;
    return fixed + (gv > mc ? gv : mc);
}

// K-B, generic form (any Gaussian radius <= 12, run-time loops): one tile of one image per workgroup.
// The default radii 8 (sigma 2.0) and 10 (sigma 2.5) use k_canny_pipe below.
__global__ __launch_bounds__(256) void k_canny(const float* __restrict__ gray, const int32_t* __restrict__ fS, int f0,
                                                int imgs_per_frame, int R, const double* __restrict__ gw,
                                                stp_u64* __restrict__ low, stp_u64* __restrict__ high)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int img = blockIdx.y;
    const int f = f0 + img / imgs_per_frame;
    const int S = fS[f];
    if (S == 0) return;
    const int tpr = (STP_FRAME_MAX + CT_X - 1) / CT_X;
    stp_tile T;
    T.S = S; T.ty0 = (blockIdx.x / tpr) * CT_Y; T.tx0 = (blockIdx.x % tpr) * CT_X;
    if (T.ty0 >= S || T.tx0 >= S) return;
    const int GW = ct_gw(R), GH = CT_Y + 2 * R + 4, VH = CT_Y + 4;
    // layout: [sW | sB | sS | sG | sV]; sM aliases sG/sV once those are dead
    double* sW = (double*)smem;                                   // 2*CT_RMAX+1 -> 32 slots
    double* sB = sW + 32;                                         // 2*VH
    double* sS = sB + 2 * VH;                                     // VH*CT_SP
    float* sG = (float*)(sS + VH * CT_SP);                        // GH*GW
    float* sV = sG + GH * GW;                                     // VH*GW
    float* sM = sG;                                               // (CT_Y+2)*(CT_X+2) f32, aliases sG
    stp_u64* sBits = (stp_u64*)(smem + canny_smem_bytes(R));      // class bit-rows + candidate queues
    uint16_t* sQ = (uint16_t*)(sBits + 2 * CT_Y);
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid < 2 * R + 1) sW[tid] = gw[tid];
    canny_p0(tid, nt, gray + (size_t)img * (STP_PITCH * STP_PITCH), T, R, sG);
    canny_p1b(tid, nt, T, R, gw, sB);
    __syncthreads();
    canny_p1(tid, nt, T, R, sW, sG, sV);
    __syncthreads();
    canny_p2(tid, nt, T, R, sW, sV, sB, sS);
    __syncthreads();
    canny_p3(tid, nt, T, sS, sM);
    __syncthreads();
    canny_nms_pack(tid, T, sS, sM, sQ, sBits, low + (size_t)img * (STP_FRAME_MAX * STP_NW),
                   high + (size_t)img * (STP_FRAME_MAX * STP_NW));
}

// K-B, per-tile brightness loop (compile-time radius): one workgroup per (tile, frame, level) walks the
// brightness images of that tile (the bleed-over factors and all index arithmetic are shared).  The
// vertical pass reads the grey image straight from global memory into registers, so LDS holds only the
// transposed vertical-pass tile (aliased later by the magnitude tile) and the smoothed tile: ~43 KB.  Blocks are mapped XCD-aware: blocks b, b+8, b+16, ... share an XCD
// (round-robin dispatch), so they get consecutive tiles of the same (frame, level) and the overlapping
// halos of neighbouring tiles are served by that XCD's L2.
#ifndef STP_CANNY_MINBLK
#define STP_CANNY_MINBLK 4
#endif
// (radius 12 -- sigma 3.0, reached under STP_CANNY=exact or through the redo list only -- needs more than the 128 registers four
//  workgroups per CU leave a lane: three workgroups, no scratch)
#define STP_CANNY_MINBLK_OF(RT) ((RT) >= 12 ? 3 : STP_CANNY_MINBLK)
static __host__ __device__ size_t canny_pipe_smem_bytes(int R)
{
    const int GW = CT_X + 2 * R + 4, VH = CT_Y + 4;
    size_t fixed = (32 + 2 * VH + VH * 2 * R + VH * CT_SP) * sizeof(double);
    size_t v = (size_t)(CT_P2_COLS(R) > GW ? CT_P2_COLS(R) : GW) * CT_VP * sizeof(float);
    // the magnitude tile and the NMS candidate queues share the vertical-pass buffer (sV is dead between
    // the horizontal pass and the next image's vertical pass): 38.9 KB at R = 8 -> 4 workgroups per CU
    size_t m = (size_t)(CT_Y + 2) * (CT_X + 2) * sizeof(float) + 4 * 512 * sizeof(uint16_t);
    return fixed + (v > m ? v : m);
}
#define CANNY_PIPE_BITS_BYTES (2 * CT_Y * 8 + 16)   /* class bit-rows + the candidate count */

// one tile of one (frame, level) pair, brightness images bi_lo .. bi_hi-1
template <int RT>
__device__ __forceinline__ void canny_pipe_tile(const float* __restrict__ gray, const int32_t* __restrict__ fS, int f0,
                                                int nf, int nlev, int nb, const double* __restrict__ gw,
                                                stp_u64* __restrict__ low, stp_u64* __restrict__ high, stp_fastdiv fd,
                                                const float2* __restrict__ cells /* null: no flat-window skip */,
                                                int pair, int tile, int bi_lo, int bi_hi)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int R = RT;
    if (pair >= nf * nlev) return;
    const int fl = pair / nlev, lev = pair - fl * nlev;
    const int S = fS[f0 + fl];
    if (S == 0) return;
    constexpr int tpr = (STP_FRAME_MAX + CT_X - 1) / CT_X;
    stp_tile T;
    T.S = S; T.ty0 = (tile / tpr) * CT_Y; T.tx0 = (tile % tpr) * CT_X;
    if (T.ty0 >= S || T.tx0 >= S) return;
    constexpr int VH = CT_Y + 4;
    double* sW = (double*)smem;
    double* sB = sW + 32;
    double* sBB = sB + 2 * VH;                   // border-column bleed-over table, VH x 2R
    double* sS = sBB + VH * 2 * R;
    float* sV = (float*)(sS + VH * CT_SP);
    float* sM = sV;                               // magnitude tile (f32) over the dead vertical-pass buffer
    uint16_t* sQ = (uint16_t*)(sM + (CT_Y + 2) * (CT_X + 2));      // 4 wave queues of 512 candidates
    stp_u64* sBits = (stp_u64*)(smem + canny_pipe_smem_bytes(R));   // class bit-rows
    int* sQn = (int*)(sBits + 2 * CT_Y);                            // candidates in the queue
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid < 2 * CT_Y) sBits[tid] = 0ull;
    if (tid == 64) *sQn = 0;
    if (tid < 2 * R + 1) sW[tid] = gw[tid];
    canny_p1b(tid, nt, T, R, gw, sB);            // bleed-over factors depend on the tile geometry only
    __syncthreads();
    // tile-uniform fast paths: rows / columns this tile touches all inside the image
    const bool yin = (T.ty0 - R - 2 >= 0) && (T.ty0 + CT_Y + R + 1 < S);
    const bool xin = (T.tx0 - 2 - R >= 0) && (T.tx0 + CT_X + 1 + R < S);
    if (!xin) canny_p1c<R>(tid, nt, T, sW, sB, sBB);
    __syncthreads();
    const size_t img0 = ((size_t)fl * nlev + lev) * nb;
    // in-image extents of the three passes' work items: the same for all images of the tile, so each thread decodes
    // its items once (a 400-pixel frame leaves 19 of its 91 tiles mostly outside the image: their idle waves skip
    // the passes instead of filtering zeros)
    constexpr int GWc = CT_X + 2 * R + 4, NG1 = (CT_Y + 4) / ((R <= 8) ? CT_VRUN : CT_VRUN / 2);
    constexpr int NR1 = (GWc * NG1 + 255) / 256;
    constexpr int NR2 = ((CT_Y + 4) * ((CT_X + 4 + CT_HRUN_R(R) - 1) / CT_HRUN_R(R)) + 255) / 256;
    int it1[NR1], it2[NR2];
    stp_p3walk W3;
    {
        const stp_cgeo G = ct_geo<R>(T);
#pragma unroll
        for (int k = 0; k < NR1; k++) it1[k] = ct_p1_decode<R>(G, tid + 256 * k);
#pragma unroll
        for (int k = 0; k < NR2; k++) it2[k] = ct_p2_decode<R>(G, tid + 256 * k);
        W3 = ct_p3_walk(G, tid, 256);
    }
    const stp_cwin CW = canny_cand_window(T);
    // the min/max cell this lane looks at (the same for all images): the cells overlapping the tile's input
    // window [ty0-R-2, ty0+CT_Y+R+2) x [tx0-R-2, tx0+CT_X+R+2) clipped to the image, at most 8 x 7
    int cell_off = -1;
    bool use_cells = cells != nullptr;
    {
        const int wy0 = max(T.ty0 - R - 2, 0), wy1 = min(T.ty0 + CT_Y + R + 2, S);
        const int wx0 = max(T.tx0 - R - 2, 0), wx1 = min(T.tx0 + CT_X + R + 2, S);
        const int r0 = wy0 / GC_CY, nr = (wy1 - 1) / GC_CY - r0 + 1, c0 = wx0 / GC_CX, nc = (wx1 - 1) / GC_CX - c0 + 1;
        const int lane = tid & 63;
        if (nr * nc > 64) use_cells = false;
        else if (lane < nr * nc) { const int rr = lane / nc; cell_off = (r0 + rr) * GC_COLS + c0 + (lane - rr * nc); }
    }
    for (int bi = bi_lo; bi < bi_hi; bi++) {
        const size_t img = img0 + bi;
        const float* gimg = gray + img * (STP_PITCH * STP_PITCH);
        if (use_cells) {                             // every wave reads the same cells: the verdict is block-uniform
            float mn = INFINITY, mx = -INFINITY;
            if (cell_off >= 0) { const float2 v = cells[img * (GC_ROWS * GC_COLS) + cell_off]; mn = v.x; mx = v.y; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
            if (mx - mn < STP_FLAT_RANGE) {          // flat window: no pixel of this tile can reach the low threshold
                if (tid < CT_Y && T.ty0 + tid < S) {
                    low[img * (STP_FRAME_MAX * STP_NW) + STP_CLS(T.ty0 + tid, T.tx0 >> 6)] = 0ull;
                    high[img * (STP_FRAME_MAX * STP_NW) + STP_CLS(T.ty0 + tid, T.tx0 >> 6)] = 0ull;
                }
                continue;
            }
        }
#pragma unroll
        for (int k = 0; k < NR1; k++) {               // this thread's vertical-pass items (decoded once per workgroup)
            int it = it1[k];
            asm volatile("" : "+v"(it));              // opaque per image: everything derived from the item (row masks, addresses)
                                                      // is recomputed here instead of being hoisted out of the image loop into
                                                      // dozens of spilled scalar registers
            if (it >= 0) {
                if (it >> 16) canny_p1_zero<R>(it & 255, (it >> 8) & 255, sV);
                else if (yin) canny_p1_item<R, true>(T, it & 255, (it >> 8) & 255, sW, gimg, sV);
                else canny_p1_item<R, false>(T, it & 255, (it >> 8) & 255, sW, gimg, sV);
            }
        }
        __syncthreads();
#if defined(STP_ABLATE_CANNY_P1)      /* timing-only build: vertical pass only */
        if (tid == 0) low[img * (STP_FRAME_MAX * STP_NW)] = (stp_u64)sV[70];
        __syncthreads();
        continue;
#endif
#pragma unroll
        for (int k = 0; k < NR2; k++) {
            int it = it2[k];
            asm volatile("" : "+v"(it));
            if (it >= 0) {
                if (xin) canny_p2_item<R, true>(T, it & 255, it >> 8, sW, sV, sB, sBB, sS, fd);
                else canny_p2_item<R, false>(T, it & 255, it >> 8, sW, sV, sB, sBB, sS, fd);
            }
        }
        __syncthreads();
#if defined(STP_ABLATE_CANNY_P12)     /* timing-only build: Gaussian passes only */
        if (tid == 0) low[img * (STP_FRAME_MAX * STP_NW)] = (stp_u64)sS[70];
        __syncthreads();
        continue;
#endif
        if (!(xin && yin)) {                      // border tile: replicate the image edge into the ring first
            canny_p3_ring(tid, nt, T, sS);
            __syncthreads();
        }
        canny_p3_collect(tid, CW, W3, sS, sM, sQ, sQn);
        __syncthreads();
#if defined(STP_ABLATE_CANNY_P123)    /* timing-only build: everything but the NMS over the queue */
        if (tid == 64) *sQn = 0;
        __syncthreads();
        continue;
#endif
        canny_nms_queue(tid, T, sS, sM, sQ, *sQn, sBits);
        __syncthreads();     // sM / sQ alias sV: the NMS must be done before the next vertical pass writes it
        if (tid < CT_Y) {    // the tile's class words of this image; the thread clears the two words it has read (the
                             // next image's candidates arrive two barriers later)
            const int y = T.ty0 + tid;
            const stp_u64 lo = sBits[tid], hi = sBits[CT_Y + tid];
            sBits[tid] = 0ull; sBits[CT_Y + tid] = 0ull;
            if (y < S) {
                low[img * (STP_FRAME_MAX * STP_NW) + STP_CLS(y, T.tx0 >> 6)] = lo;
                high[img * (STP_FRAME_MAX * STP_NW) + STP_CLS(y, T.tx0 >> 6)] = hi;
            }
        }
        if (tid == 64) *sQn = 0;
    }
}


template <int RT>
__global__ __launch_bounds__(256, STP_CANNY_MINBLK_OF(RT)) void k_canny_pipe(const float* __restrict__ gray, const int32_t* __restrict__ fS, int f0,
                                                     int nf, int nlev, int nb, const double* __restrict__ gw,
                                                     stp_u64* __restrict__ low, stp_u64* __restrict__ high, stp_fastdiv fd,
                                                     const float2* __restrict__ cells)
{
    constexpr int TPI = ((STP_FRAME_MAX + CT_X - 1) / CT_X) * ((STP_FRAME_MAX + CT_Y - 1) / CT_Y);   // 91
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    canny_pipe_tile<RT>(gray, fS, f0, nf, nlev, nb, gw, low, high, fd, cells, xcd + 8 * (j / TPI), j % TPI, 0, nb);
}
// The same arithmetic for FLAGGED (image, tile) pairs: the tile-images k_canny_f32 hands over because more of their
// pixels were undecidable in f32 than its lists take (plateaus, exact ties).  xflags[image * 91 + tile] != 0 (image
// counted within the launch); a fixed grid scans the flags.
template <int RT>
__global__ __launch_bounds__(256, STP_CANNY_MINBLK_OF(RT)) void k_canny_pipe_list(const float* __restrict__ gray, const int32_t* __restrict__ fS, int f0,
                                                     int nf, int nlev, int nb, const double* __restrict__ gw,
                                                     stp_u64* __restrict__ low, stp_u64* __restrict__ high, stp_fastdiv fd,
                                                     uint8_t* __restrict__ xflags, uint8_t* __restrict__ asym)
{
    constexpr int TPI = ((STP_FRAME_MAX + CT_X - 1) / CT_X) * ((STP_FRAME_MAX + CT_Y - 1) / CT_Y);   // 91
    const int n = nf * nlev * nb * TPI;
    // k_gray_c3's per-image asymmetry reports have been read by k_canny_f32: cleared here, like the flags served below, so
    // that both buffers are all zero between launches
    if (asym != nullptr)          // (the marks of k_gray_fill's pairs lie behind them: nf x nlev more bytes)
        for (int i = blockIdx.x * 256 + (int)threadIdx.x; i < nf * nlev * nb + nf * nlev; i += gridDim.x * 256) asym[i] = 0;
    __shared__ stp_u64 sAny[4];
    for (int k0 = blockIdx.x * 256; k0 < n; k0 += gridDim.x * 256) {     // 256 flags at a time
        const int kk = k0 + (int)threadIdx.x;
        const stp_u64 any = __ballot(kk < n && xflags[kk] != 0);
        if ((threadIdx.x & 63) == 0) sAny[threadIdx.x >> 6] = any;
        __syncthreads();
        stp_u64 m[4] = {sAny[0], sAny[1], sAny[2], sAny[3]};
        __syncthreads();
        for (int w = 0; w < 4; w++)
            while (m[w]) {                                               // workgroup-uniform
                const int k = k0 + w * 64 + __builtin_ctzll(m[w]);
                m[w] &= m[w] - 1;
                const int img = k / TPI, tile = k - img * TPI;
                canny_pipe_tile<RT>(gray, fS, f0, nf, nlev, nb, gw, low, high, fd, nullptr, img / nb, tile, img % nb, img % nb + 1);
                if (threadIdx.x == 0) xflags[k] = 0;                      // served: the buffer is all zero again after this launch
                __syncthreads();
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_canny_f32 (stp_canny32.h): the same tile / image loop as k_canny_pipe with every phase in f32, the class of a
// candidate taken from the f32 values when the error budget allows it, and the reference's f64 arithmetic on the
// 5 x 5 neighbourhood (one wave per pixel) when it does not.  Same class words as k_canny_pipe, bit for bit.
// Magnitudes and candidate collection of one image: lane = tile column (magnitude columns 1 .. CT_X), each wave a strip
// of the region's rows, walked top to bottom with the row terms of the two previous rows in registers (conflict-free
// LDS reads: consecutive lanes, consecutive words); the two halo columns 0 and CT_X + 1 (neighbours only, never
// candidates) pixel by pixel.  Candidates (magnitude >= thr inside the tile's candidate window) go to the wave's own
// segment of the queue -- its fill count is a wave-uniform register, so a row costs one ballot and no atomic.
__device__ __forceinline__ stp_u64 wave_transpose64(stp_u64 x, int lane);      // (defined with k_lines' helpers below)
#define C32_QSEG (((CT_Y + 2 + 3) / 4) * 64)       /* candidates one wave can find: its rows x 64 columns */
__device__ __forceinline__ void canny32_mag_rows(int tid, stp_cwin C, int my_lo, int nmh, int mx_lo, int nmw, float thr,
                                                 const float* sS, float* sM, uint16_t* sQ, int* sQcnt)
{
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, MW = CT_X + 2;     // (scalar: the row bounds below live in SGPRs)
    if (tid < 2 * nmh) {
        const int col = tid >= nmh, X = col ? CT_X + 1 : 0, Y = my_lo + tid - col * nmh;
        if (X >= mx_lo && X < mx_lo + nmw) sM[Y * MW + X] = c32_mag_px(sS, Y, X);
    }
    constexpr int RMAX = (CT_Y + 2 + 3) / 4;                 // rows of a wave's strip at most
    const int rows = (nmh + 3) >> 2, Y0 = my_lo + wv * rows, Y1 = min(Y0 + rows, my_lo + nmh);
    int cnt = 0;
    if (Y0 < Y1) {                                           // wave-uniform
        const int X = 1 + lane;
        const bool colin = X >= mx_lo && X < mx_lo + nmw;
        const bool xcand = colin && (unsigned)(X - C.x0) < (unsigned)C.nx;
        uint16_t* q = sQ + wv * C32_QSEG;
        // all rows of the strip in one go (round 4): the RMAX + 2 smoothed rows are requested together -- one LDS
        // round trip per image instead of one per row -- and the walk is unrolled (no register rotation, no
        // pointer arithmetic per row); rows beyond the strip repeat its last row and are not used
        float hd[RMAX + 2], hs[RMAX + 2];
#pragma unroll
        for (int r = 0; r < RMAX + 2; r++) {
            const int Yr = min(Y0 + r, Y1 + 1);             // (wave-uniform; Y1 + 1 <= CT_Y + 3, the tile's last smoothed row)
            c32_row_terms(sS + Yr * C32_SP + X, &hd[r], &hs[r]);
        }
#pragma unroll
        for (int r = 0; r < RMAX; r++) {
            const int Y = Y0 + r;
            if (Y < Y1) {                                    // wave-uniform
                const float m = c32_mag(hd[r], hd[r + 1], hd[r + 2], hs[r], hs[r + 2]);
                if (colin) sM[Y * MW + X] = m;
                const bool isq = xcand && (unsigned)(Y - C.y0) < (unsigned)C.ny && m >= thr;
                const stp_u64 bq = __ballot(isq);
                if (isq) q[cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bq >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bq, 0u))] =
                             (uint16_t)((Y - 1) * 64 + (X - 1));
                cnt += __popcll(bq);
            }
        }
    }
    if (lane == 0) sQcnt[wv] = cnt;
}

// Pixels the f32 class test cannot decide: a tile keeps them in a small LDS list (image << 11 | tile pixel) while it
// walks its brightness images, and settles them when it is done -- one wave per pixel, the reference's arithmetic on
// the 5 x 5 neighbourhood (c32_res_*) -- before it writes the class words of all its images.  An image that
// overflows the list (plateaus, exact ties) is flagged instead and redone by k_canny_pipe_list.
#define C32_NBMAX 8                  /* brightness images whose class words a workgroup keeps in LDS until its tile ends */
#define C32_DCAP 192                 /* undecidable pixels a tile keeps (all its images together) */
__device__ __forceinline__ void canny32_nms_queue(int tid, int bi, stp_tile T, stp_c32tol E, const float* sS, const float* sM,
                                                  const uint16_t* sQ, const int* sQcnt, stp_u64* sBits, uint16_t* sD, int* sDn, int* sOv)
{
    stp_u64* lowB = sBits;
    stp_u64* highB = sBits + CT_Y;
    const int n0 = sQcnt[0], n1 = n0 + sQcnt[1], n2 = n1 + sQcnt[2], n = n2 + sQcnt[3];
    // (measured and dropped in round 4: every wave testing its own segment -- no search for the segment, 12 instructions
    //  less per candidate -- took the same time: the strips' candidate counts differ and the slowest wave sets the pace)
    for (int k = tid; k < n; k += 256) {                             // candidate k of the four wave segments, in order
        const int seg = (k >= n0) + (k >= n1) + (k >= n2);
        const int idx = k - (seg == 0 ? 0 : (seg == 1 ? n0 : (seg == 2 ? n1 : n2)));
        const int e = sQ[seg * C32_QSEG + idx], yy = e >> 6, xx = e & 63;
        const int cls = c32_nms(sS, sM, T, T.ty0 + yy, T.tx0 + xx, E);
        if (cls == 3) {
            const int slot = atomicAdd(sDn, 1);
            if (slot < C32_DCAP) sD[slot] = (uint16_t)(bi << 11 | e);
            else *sOv = 1;                                            // (benign race: every writer stores 1)
        } else if (cls >= 1) {
            atomicOr(&lowB[yy], 1ull << xx);
            if (cls == 2) atomicOr(&highB[yy], 1ull << xx);
        }
    }
}
// one pixel, one wave: the 5 x (2R+5) vertical-pass values, the 5 x 5 smoothed values, nine magnitudes, the literal test
// MIR (image symmetry, see k_canny_f32): the pixel settled is the MIRROR image (column, row) of tile pixel e -- a pixel of a
// tile below the diagonal that no workgroup computes -- and its class goes to the tile's transposed words sT[plane][tile column]
template <int R, bool MIR = false>
__device__ __forceinline__ void canny32_resolve(stp_tile T, int e, int lane, const double* sW, const double* sB,
                                                const float* __restrict__ gimg, float* Vp, double* Sp, stp_u64* sBits,
                                                uint32_t* sT = nullptr)
{
    constexpr int NV = 5 * (2 * R + 5);
    const int yy = e >> 6, xx = e & 63, y = MIR ? T.tx0 + xx : T.ty0 + yy, x = MIR ? T.ty0 + yy : T.tx0 + xx;
    if constexpr (R <= 8 && !MIR) {   // the taps of a lane's two elements are requested together (one memory round trip instead of two;
                                      //  the mirror form takes the lean loop: both forms inlined with 34 tap registers each spilled at radius 8)
        float va[2 * R + 1], vb[2 * R + 1];
        const bool hb = lane + 64 < NV;
        c32_res_V_taps<R>(T, y, x, lane, gimg, va);
        if (hb) c32_res_V_taps<R>(T, y, x, lane + 64, gimg, vb);
        Vp[lane] = c32_gauss_sum<R>(va, sW);
        if (hb) Vp[lane + 64] = c32_gauss_sum<R>(vb, sW);
    } else {                  // (two windows of 21 / 25 taps would not fit the registers of five waves per SIMD)
#pragma unroll 1
        for (int l = lane; l < NV; l += 64) Vp[l] = c32_res_V<R, MIR>(T, y, x, l, sW, gimg);     // (MIR: grey values read at the transposed position)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 25) Sp[lane] = MIR ? c32_res_S_any<R>(T.S, y, x, lane, sW, Vp) : c32_res_S<R>(T, y, x, lane, sW, sB, Vp);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double* M9 = (double*)Vp;                    // the vertical-pass values are dead: nine magnitudes in their place
    if (lane < 9) M9[lane] = c32_res_mag(Sp, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        const int cls = c32_res_class(Sp, M9);
        if (MIR) {
            if (cls >= 1) atomicOr(&sT[xx], 1u << yy);
            if (cls == 2) atomicOr(&sT[CT_X + xx], 1u << yy);
        } else {
            if (cls >= 1) atomicOr(&sBits[yy], 1ull << xx);
            if (cls == 2) atomicOr(&sBits[CT_Y + yy], 1ull << xx);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

#ifndef STP_C32_MINBLK
#define STP_C32_MINBLK 5
#endif
#define C32_RES_WAVE_BYTES 832       /* per-wave resolver scratch inside the smoothed tile: 26 doubles, then 5 x (2R+5) floats (<= 145) */
struct stp_c32_layout { size_t sB, sRB, sRV, sRC, sS, sV, sQ, sD, sBits, sQn, sG, total; };
static __host__ __device__ stp_c32_layout canny32_layout(int R)
{
    const int GW = CT_X + 2 * R + 4, VH = CT_Y + 4;
    stp_c32_layout L;
    size_t o = 32 * sizeof(double);                          // sW: the f64 weights (bleed-over tables, resolver)
    L.sB = o; o += 2 * VH * sizeof(double);                  // f64 bleed-over factors (reciprocal tables, resolver)
    L.sRB = o; o += (size_t)((VH + 1) & ~1) * sizeof(float);
    L.sRV = o; o += (size_t)((VH + 1) & ~1) * sizeof(float);
    L.sRC = o; o += (size_t)C32_SP * sizeof(float);
    L.sS = o; o += (size_t)VH * C32_SP * sizeof(float);      // smoothed tile (at the end: the resolver's per-wave scratch)
    L.sV = o;                                                 // vertical-pass tile | magnitude tile + candidate queue
    const size_t v = (size_t)(CT_P2_COLS(R) > GW ? CT_P2_COLS(R) : GW) * C32_VP_R(R) * sizeof(float);
    const size_t mq = (size_t)(CT_Y + 2) * (CT_X + 2) * sizeof(float);
    L.sQ = o + mq;
    const size_t m = mq + 4 * C32_QSEG * sizeof(uint16_t);
    o += ((v > m ? v : m) + 7) & ~(size_t)7;
    L.sD = o; o += C32_DCAP * sizeof(uint16_t);
    L.sBits = o; o += (size_t)C32_NBMAX * 2 * CT_Y * sizeof(stp_u64);
    L.sQn = o; o += 32;                                       // four segment fill counts, list length, overflow flag
    L.sG = o; o += 2 * C32_NBMAX * sizeof(float);             // largest grey value under the tile's window and flat flag, per image
    L.total = o;
    return L;
}

// mirror geometry of a tile (k_canny_f32, image symmetry): does its transpose cover tiles below the diagonal, and which
struct stp_c32mgeo { bool mir, t1_in; int t0, tyi, pair; };
__device__ __forceinline__ stp_c32mgeo c32_mgeo(int mirror, int S)
{
    constexpr int TPI = ((STP_FRAME_MAX + CT_X - 1) / CT_X) * ((STP_FRAME_MAX + CT_Y - 1) / CT_Y);
    constexpr int tpr = (STP_FRAME_MAX + CT_X - 1) / CT_X;
    int b = blockIdx.x;
    asm volatile("" : "+s"(b));                      // (opaque: recomputed at every use)
    const int tile = (b >> 3) % TPI, tyi = tile / tpr, txi = tile - tyi * tpr;
    stp_c32mgeo g;
    g.mir = mirror && txi > (tyi >> 1);
    g.t0 = (2 * txi) * tpr + (tyi >> 1);             // the two tiles the transpose covers (t0, t0 + tpr), one word half each
    g.t1_in = (2 * txi + 1) * CT_Y < S;
    g.tyi = tyi;
    g.pair = (b & 7) + 8 * ((b >> 3) / TPI);         // the (frame, level) pair of this workgroup (the XCD-aware numbering)
    return g;
}
// DBG (stp_dbg_canny_f32 only; the product launches <RT, false>): after the magnitudes of image dbg_bi the tile's own pixels
// are dumped -- f32 smoothed value, Sobel sums, magnitude, the scale g and the E_G budget (u g) the tile used, six planes of
// pitch STP_PITCH -- and dbg_cnt counts the image's candidates, the pixels sent to the resolver and the flagged tile-images.
template <int RT, bool DBG = false>
__global__ __launch_bounds__(256, STP_C32_MINBLK) void k_canny_f32(const float* __restrict__ gray, const int32_t* __restrict__ fS, int f0,
                                                    int nf, int nlev, int nb, const double* __restrict__ gw,
                                                    stp_u64* __restrict__ low, stp_u64* __restrict__ high, stp_w32 W32,
                                                    const float2* __restrict__ cells, uint8_t* __restrict__ xflags,
                                                    int mirror /* the images are symmetric: see below */, const uint8_t* __restrict__ asym,
                                                    const int32_t* __restrict__ fshift /* frame overlap, or null */,
                                                    uint8_t* __restrict__ need /* per (frame, level): tiles below the diagonal go to the exact kernel (k_gray_fill), or null */,
                                                    float* __restrict__ dbg = nullptr, int dbg_bi = -1,
                                                    unsigned long long* __restrict__ dbg_cnt = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int R = RT;
    constexpr int TPI = ((STP_FRAME_MAX + CT_X - 1) / CT_X) * ((STP_FRAME_MAX + CT_Y - 1) / CT_Y);   // 91
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int pair = xcd + 8 * (j / TPI), tile = j % TPI;
    if (pair >= nf * nlev) return;
    const int fl = pair / nlev, lev = pair - fl * nlev;
    const int S = fS[f0 + fl];
    if (S == 0) return;
    constexpr int tpr = (STP_FRAME_MAX + CT_X - 1) / CT_X;
    stp_tile T;
    T.S = S; T.ty0 = (tile / tpr) * CT_Y; T.tx0 = (tile % tpr) * CT_X;
    if (T.ty0 >= S || T.tx0 >= S) return;
    // Image symmetry (round 6).  The contact matrix is symmetric and a frame keeps the same bins as rows and as columns, so
    // every image is the transpose of itself up to the ORDER of the reference's roundings (box sum row-major, Gaussian rows
    // before columns).  With `mirror` set (the host has verified the band's symmetry bit for bit and k_gray_c3 reports, per
    // image, a grey value that differs from its mirror image: `asym`) the tiles that lie strictly below the diagonal
    // (tile row >= 2 x tile column + 2) are not computed: a tile whose transpose lies there (tile column > tile row / 2)
    // hands its class words over transposed -- a verdict of the f32 test holds for the mirror pixel too (stp_canny32.h,
    // "Symmetry"), an undecidable pixel is settled twice, once per position, each in the reference's own arithmetic there.
    // (the mirror geometry is re-derived from the block index where it is needed -- c32_mgeo -- so that nothing of it lives
    //  across the image loop: held there it cost radius 8 three spilled registers)
    if (mirror && tile / tpr >= 2 * (tile % tpr) + 2) return;
#if defined(STP_ABLATE_C32_TX6)        /* timing-only build: what the 13 tiles of the last tile column (16 of their 64 columns inside a 400-pixel image) cost */
    if (tile % tpr == 6) return;
#endif
    // Frame overlap (stp_phases.h): a tile inside the block this frame shares with the next one is not computed -- k_lines takes
    // its class words (and the words of the tiles below the diagonal its transpose would have covered) from the next frame's planes
    if (fshift != nullptr && stp_reuse_tile(stp_reuse_of(fshift[f0 + fl], S, R, fl + 1 < nf), tile / tpr, tile % tpr)) return;
    const stp_c32_layout L = canny32_layout(R);
    double* sW = (double*)smem;
    float* sRB = (float*)(smem + L.sRB);
    float* sRV = (float*)(smem + L.sRV);
    float* sRC = (float*)(smem + L.sRC);
    float* sS = (float*)(smem + L.sS);
    float* sV = (float*)(smem + L.sV);
    float* sM = sV;                               // magnitude tile over the dead vertical-pass buffer
    uint16_t* sQ = (uint16_t*)(smem + L.sQ);
    uint16_t* sD = (uint16_t*)(smem + L.sD);      // undecidable pixels of this tile: image << 11 | tile pixel
    stp_u64* sBits = (stp_u64*)(smem + L.sBits);  // class words of every image of this tile: [image][low | high][row]
    int* sQcnt = (int*)(smem + L.sQn);
    int* sDn = sQcnt + 4;
    int* sOv = sQcnt + 5;
    float* sG = (float*)(smem + L.sG);
    const int tid = threadIdx.x, nt = blockDim.x, wv = tid >> 6, lane = tid & 63;
    const size_t img0 = ((size_t)fl * nlev + lev) * nb;
    // which images of this tile are live (not flat: STP_FLAT_RANGE) and the largest grey value under the tile's window: wave w
    // reduces the cells of images w and w + 4 (both cell loads of a lane in flight together), the verdicts meet in LDS
    int* sFlat = (int*)(sG + C32_NBMAX);
    {
        // (interior tiles -- every window inside the image -- may call a wider range flat: c32_flat_interior)
        const float flat_thr = ((T.ty0 - R - 2 >= 0) && (T.ty0 + CT_Y + R + 1 < S) && (T.tx0 - 2 - R >= 0) && (T.tx0 + CT_X + 1 + R < S))
                                   ? W32.flat_int : STP_FLAT_RANGE;
        const int wy0 = max(T.ty0 - R - 2, 0), wy1 = min(T.ty0 + CT_Y + R + 2, S);
        const int wx0 = max(T.tx0 - R - 2, 0), wx1 = min(T.tx0 + CT_X + R + 2, S);
        const int r0 = wy0 / GC_CY, nr = (wy1 - 1) / GC_CY - r0 + 1, c0 = wx0 / GC_CX, nc = (wx1 - 1) / GC_CX - c0 + 1;
        const bool use_cells = cells != nullptr && nr * nc <= 64;
        if (use_cells) {
            int cell_off = -1;
            if (lane < nr * nc) { const int rr = lane / nc; cell_off = (r0 + rr) * GC_COLS + c0 + (lane - rr * nc); }
            float2 cv[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int bi = wv + 4 * q;
                cv[q] = make_float2(INFINITY, -INFINITY);
                if (bi < nb && cell_off >= 0) cv[q] = cells[(img0 + bi) * (GC_ROWS * GC_COLS) + cell_off];
            }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int bi = wv + 4 * q;
                float mn = cv[q].x, mx = cv[q].y;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
                if (lane == 0 && bi < C32_NBMAX) {
                    sG[bi] = mx;
                    sFlat[bi] = (bi < nb && mx - mn < flat_thr) ? 1 : 0;          // flat window: no pixel of this tile can reach the
                                                                                // low threshold (its class words stay 0)
                }
            }
        } else if (tid < C32_NBMAX) { sG[tid] = 1.0000005f; sFlat[tid] = 0; }   // k_gray's grey values never exceed 0.299 + 0.587 + 0.114 (+ 3 roundings)
    }
    for (int i = tid; i < nb * 2 * CT_Y; i += nt) sBits[i] = 0ull;
    if (tid == 64) { *sDn = 0; *sOv = 0; }
    const bool yin = (T.ty0 - R - 2 >= 0) && (T.ty0 + CT_Y + R + 1 < S);
    const bool xin = (T.tx0 - 2 - R >= 0) && (T.tx0 + CT_X + 1 + R < S);
    double* sB = (double*)(smem + L.sB);
    {   // f64 bleed-over factors of the tile's rows and their f32 reciprocal tables
        constexpr int VH = CT_Y + 4;
        if (tid < 2 * R + 1) sW[tid] = gw[tid];
        if (yin) {                                   // every row's window inside the image: one value for all (stp_w32)
            if (tid < VH) { sB[tid] = W32.vfull; sB[VH + tid] = W32.bifull; sRB[tid] = W32.rbfull; sRV[tid] = W32.rvfull; }
            __syncthreads();
            if (!xin) c32_rb_tables<R, true>(tid, nt, T, sW, sB, sRB, sRV, sRC, xin);
        } else {
            canny_p1b(tid, nt, T, R, gw, sB);
            __syncthreads();
            c32_rb_tables<R, false>(tid, nt, T, sW, sB, sRB, sRV, sRC, xin);
        }
    }
    unsigned live = 0;
    __syncthreads();                                 // the verdicts of all four waves
#pragma unroll
    for (int bi = 0; bi < C32_NBMAX; bi++) live |= (bi < nb && !sFlat[bi]) ? (1u << bi) : 0u;
    live = (unsigned)__builtin_amdgcn_readfirstlane((int)live);
    static_assert((CT_Y + 4) % C32_VRUN_R(R) == 0, "whole row groups");
    constexpr int NR1 = ((CT_X + 2 * R + 4) * ((CT_Y + 4) / C32_VRUN_R(R)) + 255) / 256;
    constexpr int NR2 = ((CT_Y + 4) * ((CT_X + 4 + CT_HRUN_R(R) - 1) / CT_HRUN_R(R)) + 255) / 256;
    int it1[NR1], it2[NR2];
    struct { int my_lo, nmh, mx_lo, nmw; } G3;
    {
        const stp_cgeo G = ct_geo<R>(T);
        const stp_c32geo1 G1 = c32_geo1<R>(T);
#pragma unroll
        for (int k = 0; k < NR1; k++) it1[k] = c32_p1_decode(G1, tid + 256 * k);
#pragma unroll
        for (int k = 0; k < NR2; k++) it2[k] = ct_p2_decode<R>(G, tid + 256 * k);
        G3.my_lo = G.my_lo; G3.nmh = G.nmh; G3.mx_lo = G.mx_lo; G3.nmw = (int)G.nmw.d;
    }
    const stp_cwin CW = canny_cand_window(T);
    const int et = c32_budget_of(xin && yin, S, R);  // tile-wide budget: full windows everywhere, or the worst cut
    // (measured and dropped in round 3, on the dword-load form: prefetching the next image's inputs into registers, 18-row
    //  items, two packed columns per lane, the resolver in a kernel of its own, 6 waves per SIMD -- docs/history/DESIGN_r01-r05.md section 3)
    int prev = -1;                                   // the image whose class test has just run (overflow check below)
    while (live) {                                   // workgroup-uniform
        const int bi = __builtin_ctz(live);
        live &= live - 1;
        const size_t img = img0 + bi;
        const float* gimg = gray + img * (STP_PITCH * STP_PITCH);
        __syncthreads();                             // the class test of the previous image is done (sM / sQ alias sV);
                                                     // first pass: the tables are written
        if (tid == 64 && prev >= 0 && *sOv) {        // the previous image overflowed the tile's list: the whole tile-image is redone exactly
            *sOv = 0;
            *sDn = C32_DCAP;
            xflags[(img0 + prev) * TPI + tile] = 1;
            const stp_c32mgeo mg = c32_mgeo(mirror, S);
            if (mg.mir) {                             // ... and the tiles below the diagonal that would have received its transpose
                xflags[(img0 + prev) * TPI + mg.t0] = 1;
                if (mg.t1_in) xflags[(img0 + prev) * TPI + mg.t0 + tpr] = 1;
                if (need != nullptr) need[mg.pair] = 1;
            }
        }
        prev = bi;
        const stp_c32tol E = c32_tol_u(sG[bi], W32.eu[et][0], W32.eu[et][1], W32.eu[et][2]);
#pragma unroll
        for (int k = 0; k < NR1; k++) {
            int it = it1[k];
            asm volatile("" : "+v"(it));              // see k_canny_pipe
            if (it >= 0) {
                if (it >> 16) c32_p1_zero<R>(it & 255, (it >> 8) & 255, sV);
                else if (yin) c32_p1_item<R, true>(T, it & 255, (it >> 8) & 255, W32, gimg, sV);
                else c32_p1_item<R, false>(T, it & 255, (it >> 8) & 255, W32, gimg, sV);
            }
        }
        __syncthreads();
#if STP_ABLATE_C32 == 1                  /* timing-only builds: stop after the vertical pass ... */
        if (tid == 0) low[img * (STP_FRAME_MAX * STP_NW)] = (stp_u64)sV[70];
        continue;
#endif
#pragma unroll
        for (int k = 0; k < NR2; k++) {
            int it = it2[k];
            asm volatile("" : "+v"(it));
            if (it >= 0) {
                if (xin) c32_p2_item<R, true>(T, it & 255, it >> 8, W32, sV, sRB, sRV, sRC, sS);
                else c32_p2_item<R, false>(T, it & 255, it >> 8, W32, sV, sRB, sRV, sRC, sS);
            }
        }
        __syncthreads();
#if STP_ABLATE_C32 == 2                  /* ... after the horizontal pass ... */
        if (tid == 0) low[img * (STP_FRAME_MAX * STP_NW)] = (stp_u64)sS[70];
        continue;
#endif
        if (!(xin && yin)) {
            c32_p3_ring(tid, nt, T, sS);
            __syncthreads();
        }
        canny32_mag_rows(tid, CW, G3.my_lo, G3.nmh, G3.mx_lo, G3.nmw, E.thr, sS, sM, sQ, sQcnt);
        __syncthreads();
#if STP_ABLATE_C32 == 3                  /* ... after the magnitudes and the candidate collection ... */
        continue;
#endif
        int dn0 = 0;
        if (DBG && bi == dbg_bi) {
            constexpr size_t PL = (size_t)STP_PITCH * STP_PITCH;
            for (int i = tid; i < CT_Y * CT_X; i += nt) {
                const int yy = i / CT_X, xx = i - yy * CT_X, y = T.ty0 + yy, x = T.tx0 + xx;
                if (y < S && x < S) {
                    const float* c = sS + (yy + 2) * C32_SP + (xx + 2);
                    float gi, gj;
                    c32_sobel(c, &gi, &gj);
                    const size_t o = (size_t)y * STP_PITCH + x;
                    dbg[o] = c[0]; dbg[PL + o] = gi; dbg[2 * PL + o] = gj;
                    dbg[3 * PL + o] = sM[(yy + 1) * (CT_X + 2) + xx + 1];
                    dbg[4 * PL + o] = sG[bi]; dbg[5 * PL + o] = W32.eu[et][0];
                }
            }
            dn0 = *sDn;                               // (everybody reads it before the class test appends: barrier below)
            if (tid == 0) atomicAdd(&dbg_cnt[0], (unsigned long long)(sQcnt[0] + sQcnt[1] + sQcnt[2] + sQcnt[3]));
            __syncthreads();
        }
        canny32_nms_queue(tid, bi, T, E, sS, sM, sQ, sQcnt, sBits + bi * 2 * CT_Y, sD, sDn, sOv);
        if (DBG && bi == dbg_bi) {
            __syncthreads();
            if (tid == 0) {
                atomicAdd(&dbg_cnt[1], (unsigned long long)(min(*sDn, C32_DCAP) - min(dn0, C32_DCAP)));
                if (*sOv) atomicAdd(&dbg_cnt[2], 1ull);
            }
        }
    }
    __syncthreads();
    const stp_c32mgeo mg = c32_mgeo(mirror, S);
    const bool mir = mg.mir, mtile1_in = mg.t1_in;
    const int mtile0 = mg.t0, tyi = mg.tyi;
    if (tid == 64 && prev >= 0 && *sOv) {            // (the last image's overflow)
        *sOv = 0;
        *sDn = C32_DCAP;
        xflags[(img0 + prev) * TPI + tile] = 1;
        if (mir) {
            xflags[(img0 + prev) * TPI + mtile0] = 1;
            if (mtile1_in) xflags[(img0 + prev) * TPI + mtile0 + tpr] = 1;
            if (need != nullptr) need[mg.pair] = 1;
        }
    }
    uint32_t* sT = (uint32_t*)sV;                    // transposed class words [image][low | high][tile column] (the pass buffers are dead)
    if (mir) {
        // an image whose grey values are not their own mirror image (k_gray_c3 found a pixel whose two summation orders
        // round differently: ~1e-8 of the pixels) is not mirrored: its tiles below the diagonal go to the exact kernel
        if (tid >= 128 && tid < 128 + nb && asym != nullptr && asym[img0 + (tid - 128)]) {
            xflags[(img0 + (tid - 128)) * TPI + mtile0] = 1;
            if (mtile1_in) xflags[(img0 + (tid - 128)) * TPI + mtile0 + tpr] = 1;
            if (need != nullptr) need[mg.pair] = 1;
        }
        // the verdicts of the f32 test, transposed: 32 rows x 64 columns -> 64 half words (the undecidable pixels are still
        // 0 in both forms; the resolver below sets them per position)
        for (int k = wv; k < nb * 2; k += 4) {       // wave-uniform
            const stp_u64 t = wave_transpose64(lane < CT_Y ? sBits[k * CT_Y + lane] : 0ull, lane);
            sT[k * CT_X + lane] = (uint32_t)t;
        }
    }
    __syncthreads();
#if STP_ABLATE_C32 == 0                  /* (4: ... or with the undecidable pixels left unsettled) */
    {
        const int nd = min(*sDn, C32_DCAP);
        float* sVp = (float*)(smem + L.sS + wv * C32_RES_WAVE_BYTES + 26 * sizeof(double));      // the smoothed tile is dead
        double* sSp = (double*)(smem + L.sS + wv * C32_RES_WAVE_BYTES);
        for (int k = wv; k < nd; k += 4) {           // wave-uniform
            const int d = sD[k], bi = d >> 11;
            const float* gimg = gray + (img0 + bi) * (STP_PITCH * STP_PITCH);
            int ln = lane;
            asm volatile("" : "+v"(ln));             // (opaque per pixel and form: the element geometry a lane derives from its number is
                                                     //  recomputed, not kept live across both inlined forms -- that spilled at radius 8)
            canny32_resolve<R>(T, d & 2047, ln, sW, sB, gimg, sVp, sSp, sBits + bi * 2 * CT_Y);
            if (mir) {
                asm volatile("" : "+v"(ln));
                canny32_resolve<R, true>(T, d & 2047, ln, sW, sB, gimg, sVp, sSp, nullptr, sT + bi * 2 * CT_X);
            }
        }
    }
    __syncthreads();
#endif
    for (int i = tid; i < nb * CT_Y; i += nt) {      // the tile's class words of every image
        const int bi = i / CT_Y, row = i - bi * CT_Y, y = T.ty0 + row;
        if (y < S) {
            const size_t o = (img0 + bi) * (STP_FRAME_MAX * STP_NW) + STP_CLS(y, T.tx0 >> 6);   // 32 rows = 256 contiguous bytes per plane
            low[o] = sBits[bi * 2 * CT_Y + row];                     // (plain stores: non-temporal ones for the class words, or non-temporal
            high[o] = sBits[bi * 2 * CT_Y + CT_Y + row];            //  loads in k_lines' loader, changed nothing -- measured in round 6)
        }
    }
    if (mir) {
        // ... and their transposes: rows tx0 .. tx0 + 63 of word column ty0 / 64, the half word of this tile's 32 rows.  The
        // other half belongs to the tile row's partner (ty0 +- 32, same word); where that partner lies beyond the image the
        // whole word is written, so that k_lines never reads a word nobody wrote.
        const int wcol = tyi >> 1, half = tyi & 1;
        const bool solo = half == 0 && T.ty0 + CT_Y >= S;
        for (int i = tid; i < nb * CT_X; i += nt) {
            const int bi = i / CT_X, xx = i - bi * CT_X, row = T.tx0 + xx;
            if (row < S) {
                const size_t o = (img0 + bi) * (STP_FRAME_MAX * STP_NW) + STP_CLS(row, wcol);
                const uint32_t l32 = sT[(bi * 2) * CT_X + xx], h32 = sT[(bi * 2 + 1) * CT_X + xx];
                if (solo) { low[o] = l32; high[o] = h32; }
                else { ((uint32_t*)low)[2 * o + half] = l32; ((uint32_t*)high)[2 * o + half] = h32; }
            }
        }
    }
}

struct stp_drec {
    int16_t ud, x, y, w, h, pad0, pad1, pad2;
    double total;
};

// 64 x 64 bit-block transpose across a wave: lane L holds row L (bit c = column c); afterwards lane L
// holds column L (bit r = row r).  Recursive block swap: at block size j the off-diagonal j x j
// sub-blocks of every 2j x 2j block are exchanged between lanes L and L ^ j.
// Round 5: no LDS traffic (rounds 1-4 fetched the partner's words with ds_bpermute: 12 per transpose, 252 sites in
// k_lines).  The exchange of stage 32 IS v_permlane32_swap on the (low word, high word) pair -- lanes 0..31 hand their
// high word to lanes 32..63 and take those lanes' low word; stage 16 takes the partner row's word by v_permlane16_swap and
// merges half-words with one v_perm_b32 whose selector depends on the lane's side; stages 8 .. 1 stay inside a row of 16
// lanes: the partner's word arrives by a DPP move (row_ror:8, row_shl:4 / row_shr:4 by bank, quad_perm), stage 8 merges bytes by
// v_perm_b32, stages 4 .. 1 rotate the partner's word into place (v_alignbit_b32) and insert it under a mask (v_bfi_b32).
// one word of a stage j <= 4: lanes with bit j clear keep their bits under m and take the partner's bits under m, moved up
// by j; lanes with the bit set keep their bits under m << j (= ~m) and take the partner's bits under ~m, moved down by j
__device__ __forceinline__ unsigned wt_bits(unsigned w, unsigned p, unsigned rot /* 32 - j | j */, unsigned ins /* ~m | m */)
{
    const unsigned r = __builtin_amdgcn_alignbit(p, p, rot);                         // rotate right by rot
    return (ins & r) | (~ins & w);                                                    // v_bfi_b32
}
__device__ __forceinline__ stp_u64 wave_transpose64(stp_u64 x, int lane)
{
    unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
    {   // j = 32: lanes 0..31: (lo, partner lo); lanes 32..63: (partner hi, hi)
        const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);        // lo's lanes 32..63 <-> hi's lanes 0..31
        lo = r[0]; hi = r[1];
    }
    {   // j = 16: even rows: low half-word own, high half-word = partner's low; odd rows: low = partner's high, high own
        const bool odd = (lane & 16) != 0;
        const unsigned sel = odd ? 0x07060302u : 0x05040100u;
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);        // even rows: (own, partner); odd rows: (partner, own)
        lo = __builtin_amdgcn_perm(a[1], a[0], sel);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        hi = __builtin_amdgcn_perm(b[1], b[0], sel);
    }
    {   // j = 8: bytes.  clear: (own b0, partner b0, own b2, partner b2); set: (partner b1, own b1, partner b3, own b3)
        const unsigned sel = (lane & 8) ? 0x03070105u : 0x06020400u;                  // v_perm_b32(S0 = partner: bytes 4..7, S1 = own: 0..3)
        lo = __builtin_amdgcn_perm(wt_dpp_xor8(lo), lo, sel);
        hi = __builtin_amdgcn_perm(wt_dpp_xor8(hi), hi, sel);
    }
    {
        const bool set = (lane & 4) != 0;
        const unsigned rot = set ? 4u : 28u, ins = set ? 0x0F0F0F0Fu : 0xF0F0F0F0u;
        lo = wt_bits(lo, wt_dpp_xor4(lo), rot, ins); hi = wt_bits(hi, wt_dpp_xor4(hi), rot, ins);
    }
    {
        const bool set = (lane & 2) != 0;
        const unsigned rot = set ? 2u : 30u, ins = set ? 0x33333333u : 0xCCCCCCCCu;
        lo = wt_bits(lo, wt_dpp_xor2(lo), rot, ins); hi = wt_bits(hi, wt_dpp_xor2(hi), rot, ins);
    }
    {
        const bool set = (lane & 1) != 0;
        const unsigned rot = set ? 1u : 31u, ins = set ? 0x55555555u : 0xAAAAAAAAu;
        lo = wt_bits(lo, wt_dpp_xor1(lo), rot, ins); hi = wt_bits(hi, wt_dpp_xor1(hi), rot, ins);
    }
    return ((stp_u64)hi << 32) | lo;
}
// column (64 * cg + lane) of a row-major bit matrix as 7 words, all 64 lanes of the wave cooperating
__device__ __forceinline__ void wave_load_cols(const stp_u64* m, int S, int cg, int lane, stp_u64* col)
{
    // (all seven words are requested before the first transpose: as one conditional read per block each LDS round trip was
    //  waited for in turn.  A row beyond the image reads row S - 1's word -- inside the matrix -- and is then replaced by 0.)
    stp_u64 x[STP_NW];
#pragma unroll
    for (int k = 0; k < STP_NW; k++) {
        const int r = 64 * k + lane;
        x[k] = m[(r < S ? r : S - 1) * STP_NW + cg];
    }
#pragma unroll
    for (int k = 0; k < STP_NW; k++) {
        const int r = 64 * k + lane;
        col[k] = wave_transpose64((r < S) ? x[k] : 0ull, lane);
    }
}

// lines_group_pairs (getStripe.py:994-1078) on one wave for nrow <= 64 candidate columns, lane c <-> cidx[c].
// The reference's grouping loop, restated per index c (a_c: cidx[c+1] == cidx[c] + 1, c <= nrow-2):
//   * a maximal run [s..e] of adjacent columns (e > s) is flushed once: X gets round(weighted mean), the
//     weights taken in order by the run's first lane (same f64 operation order as the serial form);
//   * an index c <= nrow-2 outside any run is flushed as a single: X gets cidx[c];
//   * when a run ends exactly at nrow-2, the stale [Current] left behind is flushed at the end: X gets
//     cidx[nrow-2]; the last index is never flushed on its own; nrow <= 1 gives X = {0}.
// X = sorted(set(...)) is a 448-bit LDS bitmap; ranks by popcount give xs[]; neighbour pairs are tested one
// per lane and appended in order by ballot.  All lanes of the wave must call; returns the new record count.
__device__ int lines_group_pairs_wave(int lane, int S, int ud, int maxW, int nrow, const int16_t* minr, const int16_t* maxr,
                                      const int16_t* cidx, const int16_t* clen, int16_t* xs, stp_u64* seen, stp_lrec* recs,
                                      int nrec, int cap)
{
    if (lane < STP_NW) seen[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int ci = (lane < nrow) ? cidx[lane] : -4;
    const int cn = (lane + 1 < nrow) ? cidx[lane + 1] : -8;
    const bool a = (lane + 1 < nrow) && (cn - ci == 1);
    const stp_u64 A = __ballot(a);
    const bool aprev = lane > 0 && ((A >> (lane - 1)) & 1ull);
    int mv = -1;                                        // the X value this lane contributes
    if (nrow <= 1) { if (lane == 0) mv = 0; }
    else if (a && !aprev) {                             // first lane of a run: [lane .. e]
        const stp_u64 rest = ~A >> lane;                // a is false at nrow-1, so a zero bit exists
        const int e = lane + __ffsll((long long)rest) - 1;
        long ssum = 0;
        for (int q = lane; q <= e; q++) ssum += clen[q];
        double temp = 0.0;
        for (int q = lane; q <= e; q++) temp = temp + (double)cidx[q] * ((double)clen[q] / (double)ssum);
        mv = (int)nearbyint(temp);
    } else if (!a && !aprev && lane <= nrow - 2) {
        mv = ci;                                        // single: cidx * (clen / clen)
    } else if (!a && aprev && lane == nrow - 2) {
        mv = ci;                                        // stale [Current] flushed after the loop
    }
    if (mv >= 0 && mv < S) atomicOr((unsigned long long*)&seen[mv >> 6], 1ull << (mv & 63));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int nx = 0, rank = 0;
#pragma unroll
    for (int w = 0; w < STP_NW; w++) {
        const stp_u64 v = seen[w];
        nx += __popcll(v);
        if (mv >= 0 && mv < S) {
            if (w < (mv >> 6)) rank += __popcll(v);
            else if (w == (mv >> 6)) rank += __popcll(v & ((1ull << (mv & 63)) - 1ull));
        }
    }
    if (mv >= 0 && mv < S) xs[rank] = (int16_t)mv;      // equal values share a rank
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bool emit = false;
    stp_lrec r;
    r.ud = (int16_t)ud; r.x = r.y = r.w = r.h = 0;
    if (lane + 1 < nx) {
        const int n = xs[lane], m = xs[lane + 1];
        const int gap = m - n;
        if (gap > 1 && gap <= maxW) {
            const int p1 = gap > 4 ? m - 2 : m;
            int MIN = S, MAX = -1;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int ctr = q ? m : n;
                const int lo = ctr - 1 < 0 ? 0 : ctr - 1, hi = ctr + 2 > S ? S : ctr + 2;
                for (int xx = lo; xx < hi; xx++) {
                    if (minr[xx] < MIN) MIN = minr[xx];
                    if (maxr[xx] > MAX) MAX = maxr[xx];
                }
            }
            if (ud == 1) MAX = p1; else MIN = n;
            r.x = (int16_t)n; r.y = (int16_t)MIN; r.w = (int16_t)(p1 - n + 1); r.h = (int16_t)(MAX - MIN + 1);
            emit = true;
        }
    }
    const stp_u64 em = __ballot(emit);
    if (emit) {
        // (lanes below this one that emit: mbcnt, not a popcount under a per-lane mask -- the mask, hoisted out of the direction
        //  loop, was what k_lines spilled once the wave-cooperative paint had joined the loop)
        const int pos = nrec + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(em >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)em, 0u));
        if (pos < cap) recs[pos] = r;
    }
    return nrec + __popcll(em);
}

// lines_paint (stp_phases.h: getStripe.py:948-955) by waves (round 6): a wave takes 64 columns; the columns to paint (a ballot)
// are walked one after the other and each one's rows are painted by all 64 lanes side by side -- 64 rows per step on different LDS
// words -- instead of every lane walking its own column alone, one atomic after the other, while the wave waits for its longest
// column (a stripe is up to 400 rows long; an image paints ~20 columns).  Same words, same bits: OR commutes.
__device__ __forceinline__ void lines_paint_wave(int tid, int nt, int S, int ud, const int16_t* colEnd, const int16_t* colUd, stp_u64* sT)
{
    const int lane = tid & 63;
    for (int c0 = (tid >> 6) << 6; c0 < S; c0 += nt) {            // wave-uniform
        const int c = c0 + lane;
        stp_u64 m = __ballot(c < S && colUd[c] == ud);
        while (m) {                                                // wave-uniform
            const int col = c0 + __builtin_ctzll(m);
            m &= m - 1;
            // (the column's bounds again from LDS, one broadcast read: no per-lane copies live across the loop)
            int cs = col, ce = __builtin_amdgcn_readfirstlane((int)colEnd[col]);
            if (ud == 1) { const int t = cs; cs = ce; ce = t; }
            if (cs < 0) cs = 0;
            if (ce > S) ce = S;
            const stp_u64 bit = 1ull << (col & 63);
            for (int y = cs + lane; y < ce; y += 64) atomicOr(&sT[y * STP_NW + (col >> 6)], bit);
        }
    }
}

// K-C: hysteresis + verticalLine + block + line joining + totals, one workgroup per image, every mask
// bit-packed in LDS.  Only TWO 22.4 KB bit matrices are resident (bufA: low -> vert, bufB: edges ->
// testmat): after verticalLine the edge map is parked in global memory (escr, L2-resident) and re-read
// one row per lane during refinement; the 3-column OR of `block` is formed on the fly.  53 KB of LDS
// -> three workgroups per CU (this kernel is latency bound: serial scans, one-lane grouping).
template <int RCAP>   // record slots per image: STP_RCAP for the sweep, STP_RCAP_MAX (every possible pair) for the rare re-run
__global__ __launch_bounds__(512, RCAP == STP_RCAP ? 6 : 4) void k_lines(const stp_u64* __restrict__ low, const stp_u64* __restrict__ high,
                                                const double* __restrict__ band, int W, int hw,
                                                const int32_t* __restrict__ fstart, const int32_t* __restrict__ fS,
                                                const int16_t* __restrict__ fnz, int f0, int imgs_per_frame,
                                                int minH, int maxW, stp_drec* __restrict__ recs,
                                                int32_t* __restrict__ rec_count, stp_u64* __restrict__ escr, int want_dbg,
                                                stp_u64* __restrict__ dbg /* E,V,T1,T2 */, int16_t* __restrict__ dbg_cols,
                                                int dbg_stop, const int32_t* __restrict__ fshift /* frame overlap, or null */,
                                                int gauss_radius, int nframes, int mirror)
{
    __shared__ stp_u64 bufA[STP_FRAME_MAX * STP_NW];   // low -> V (vert)
    __shared__ stp_u64 bufB[STP_FRAME_MAX * STP_NW];   // E (edges) -> testmat -> row sums
    __shared__ int16_t s_nz[STP_FRAME_MAX];
    __shared__ int16_t colEnd[STP_FRAME_MAX], colUd[STP_FRAME_MAX];
    __shared__ int16_t cnt[STP_FRAME_MAX], minr[STP_FRAME_MAX], maxr[STP_FRAME_MAX];
    __shared__ int16_t cidx[STP_FRAME_MAX], clen[STP_FRAME_MAX], xs[STP_FRAME_MAX + 8];
    __shared__ stp_lrec lrec[RCAP];
    __shared__ int s_nrec, s_nrow, s_wcnt[8];
    __shared__ stp_u64 s_seen[STP_NW];
    int16_t* colT = cidx;                              // block lengths: only copied out for the parity tests
    const int img = blockIdx.x;
    const int f = f0 + img / imgs_per_frame;
    const int S = fS[f];
    const int tid = threadIdx.x, nt = blockDim.x;
    if (S == 0) {
        if (tid == 0) rec_count[img] = 0;
        return;
    }
    const stp_u64* limg = low + (size_t)img * (STP_FRAME_MAX * STP_NW);
    const stp_u64* himg = high + (size_t)img * (STP_FRAME_MAX * STP_NW);
    stp_u64* eimg = escr + (size_t)img * (STP_FRAME_MAX * STP_NW);
    for (int i = tid; i < S; i += nt) s_nz[i] = fnz[(size_t)f * STP_FRAME_MAX + i];
    stp_reuse U;                  // the class words of the tiles k_canny_f32 skipped come from the next frame's planes (stp_phases.h)
    U.lo = 1; U.hi = 0; U.shift = -1;
    if (fshift != nullptr) U = stp_reuse_of(fshift[f], S, gauss_radius, img / imgs_per_frame + 1 < nframes);
    if (U.lo < U.hi) {            // workgroup-uniform
        static_assert(512 * STP_REUSE_ITEMS >= 192 * STP_NW, "every (row, word) item of the largest square has a thread");
        const size_t nxt = (size_t)imgs_per_frame * (STP_FRAME_MAX * STP_NW);
        stp_reuse_fetch F;
        lines_reuse_request(tid, nt, S, U, mirror, limg + nxt, himg + nxt, &F);
        lines_load(tid, nt, S, limg, himg, bufA, bufB);
        __syncthreads();          // (the plain load has stored whatever the skipped tiles' words held)
        lines_reuse_patch(&F, bufA, bufB);
    } else
        lines_load(tid, nt, S, limg, himg, bufA, bufB);
    __syncthreads();
    if (dbg_stop == 1) { if (tid == 0) rec_count[img] = (int)(bufB[3] & 0); return; }      // timing-only ablation
    if (!want_dbg) {   // an image without a single strong pixel has no edges (hysteresis keeps only components that hold
                       // one: _canny.py:286-296), hence no vertical lines and no records: nothing left to do
        stp_u64 any = 0ull;
        for (int i = tid; i < S * STP_NW; i += nt) any |= bufB[i];
        if (!__syncthreads_or(any != 0ull)) {
            if (tid == 0) rec_count[img] = 0;
            return;
        }
    }
    {   // hysteresis closure: one (8-row strip x word) item per lane, rows held in registers across sweeps
        const int nitem = ((S + STP_HYST_STRIP - 1) / STP_HYST_STRIP) * STP_NW;       // <= 350 < blockDim
        const bool has = tid < nitem;
        stp_hyst_item hit;
        static_assert((((STP_FRAME_MAX + STP_HYST_STRIP - 1) / STP_HYST_STRIP) * STP_NW + 1) * sizeof(uint16_t) <= sizeof(lrec),
                      "the items' edge bits fit the record slots");
        uint16_t* sEdge = (uint16_t*)lrec;             // the items' edge bits (hyst_edge_bits): no record exists before the grouping.
                                                       // (not cnt[]: &cnt[tid] is formed again for the column statistics, and the compiler
                                                       //  kept the common address alive across the whole kernel -- one spilled register)
        hyst_item_load(has ? tid : 0, S, bufA, bufB, &hit, sEdge);
        if (tid == 0) sEdge[nitem] = 0;                // neighbours beyond the image
        __syncthreads();
        // (measured and dropped in round 5: an item sweeping again only if it or one of its eight neighbours changed in the previous
        //  sweep -- images need 5 sweeps on average, profiles/r05_hyst_sweeps.txt -- 11.2 -> 11.1 ms per step: a wave holds nine strips
        //  over the full width and almost always has a changed item in or next to its band, so whole waves rarely sit a sweep out)
        int sweeps = 0;
        for (int it = 0; it < 4 * STP_FRAME_MAX * STP_FRAME_MAX; it++) {
            const int ch = has ? hyst_item_sweep(S, &hit, bufA, bufB, sEdge) : 0;
            sweeps++;
            if (!__syncthreads_or(ch)) break;
        }
        // (diagnostic of the timing-only build, STP_LINES_STOP=99: the image reports its number of sweeps as its record count)
        if (dbg_stop == 99) { if (tid == 0) rec_count[img] = min(sweeps, RCAP); return; }
    }
    if (dbg_stop == 2) { if (tid == 0) rec_count[img] = (int)(bufB[3] & 0); return; }
    lines_vline(tid, nt, S, bufB, bufA);               // low is dead: vert goes to bufA
    for (int i = tid; i < S * STP_NW; i += nt) eimg[i] = bufB[i];   // park the edge map
    __threadfence_block();
    __syncthreads();
    lines_v3(tid, nt, S, bufA, bufB);                  // 3-column OR into the (now free) edge buffer
    __syncthreads();
    if (dbg_stop == 3) { if (tid == 0) rec_count[img] = (int)(bufA[3] & 0); return; }
    {   // block + keep test, one column per lane held as 7 words (waves 0..6 <-> the 7 word-columns)
        const int lane = tid & 63, cg = tid >> 6;
        if (cg < STP_NW) {
            stp_u64 v3[STP_NW], v[STP_NW];
            wave_load_cols(bufB, S, cg, lane, v3);
            wave_load_cols(bufA, S, cg, lane, v);
            const int c = 64 * cg + lane;
            if (c < S) col_block_scan(v3, v, c, S, minH, &colT[c], &colEnd[c], &colUd[c]);
        }
    }
    if (tid == 0) s_nrec = 0;
    __syncthreads();
    if (dbg_stop == 4) { if (tid == 0) rec_count[img] = colT[3] & 0; return; }
    if (want_dbg) {
        stp_u64* d = dbg + (size_t)img * 4 * (STP_FRAME_MAX * STP_NW);
        for (int i = tid; i < S * STP_NW; i += nt) { d[i] = eimg[i]; d[STP_FRAME_MAX * STP_NW + i] = bufA[i]; }
        int16_t* dc = dbg_cols + (size_t)img * 3 * STP_FRAME_MAX;
        for (int i = tid; i < S; i += nt) { dc[i] = colT[i]; dc[STP_FRAME_MAX + i] = colEnd[i]; dc[2 * STP_FRAME_MAX + i] = colUd[i]; }
        __syncthreads();
    }
    for (int ud = 1; ud <= 2; ud++) {
        lines_zero(tid, nt, S * STP_NW, bufB);
        __syncthreads();
#if defined(STP_EXP_OLDPAINT)           /* A/B build: a lane per column, as in rounds 1-5 */
        lines_paint(tid, nt, S, ud, colEnd, colUd, bufB);
#else
        lines_paint_wave(tid, nt, S, ud, colEnd, colUd, bufB);
#endif
        __syncthreads();
        if (dbg_stop == 7) { if (tid == 0) rec_count[img] = (int)(bufB[3] & 0); return; }
        lines_refine(tid, nt, S, eimg, bufA, bufB);
        __syncthreads();
        if (dbg_stop == 8) { if (tid == 0) rec_count[img] = (int)(bufB[3] & 0); return; }
        {   // column counts / first / last row from the transposed testmat
            const int lane = tid & 63, cg = tid >> 6;
            if (cg < STP_NW) {
                stp_u64 tc[STP_NW];
                wave_load_cols(bufB, S, cg, lane, tc);
                const int c = 64 * cg + lane;
                if (c < S) col_stat(tc, S, &cnt[c], &minr[c], &maxr[c]);
            }
        }
        if (want_dbg) {
            stp_u64* d = dbg + (size_t)img * 4 * (STP_FRAME_MAX * STP_NW) + (size_t)(1 + ud) * (STP_FRAME_MAX * STP_NW);
            for (int i = tid; i < S * STP_NW; i += nt) d[i] = bufB[i];
        }
        __syncthreads();
        if (dbg_stop == 9) { if (tid == 0) rec_count[img] = cnt[3] & 0; return; }
        lines_cols_count(tid, nt, S, cnt, s_wcnt);
        __syncthreads();
        lines_cols_place(tid, nt, S, cnt, s_wcnt, cidx, clen, &s_nrow);
        __syncthreads();
        if (tid < 64 && dbg_stop != 6) {               // wave 0; the serial form only for > 64 candidate columns
            int nr;
            if (s_nrow <= 64) nr = lines_group_pairs_wave(tid, S, ud, maxW, s_nrow, minr, maxr, cidx, clen, xs, s_seen, lrec, s_nrec, RCAP);
            else nr = (tid == 0) ? lines_group_pairs(S, ud, maxW, s_nrow, minr, maxr, cidx, clen, xs, lrec, s_nrec, RCAP) : 0;
            if (tid == 0) s_nrec = nr;
        }
        __syncthreads();
    }
    if (dbg_stop == 5 || dbg_stop == 6) { if (tid == 0) rec_count[img] = 0; return; }
    const int nrec = s_nrec;
    const int nst = nrec < RCAP ? nrec : RCAP;
    stp_drec* out = recs + (size_t)img * RCAP;
    const int64_t st = fstart[f];
    {   // totals (getStripe.py:1094): one record per wave at a time -- row sums by the 64 lanes into the wave's
        // own slice of the (now dead) bit matrices, then lane 0 adds the rows in order
        const int wv = tid >> 6, lane = tid & 63;
        double* rsw = (wv < 4 ? (double*)bufA : (double*)bufB) + (wv & 3) * 448;
        for (int k = wv; k < nst; k += 8) {
            const stp_lrec rc = lrec[k];
            lines_rowsum(lane, 64, S, band, W, hw, st, s_nz, rc, rsw);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                double tot = 0.0;
                int i = 0;
                for (; i + 8 <= rc.h; i += 8) {                  // (eight LDS reads in flight, then the eight additions in order)
                    double v[8];
#pragma unroll
                    for (int t = 0; t < 8; t++) v[t] = rsw[i + t];
#pragma unroll
                    for (int t = 0; t < 8; t++) tot += v[t];
                }
                for (; i < rc.h; i++) tot += rsw[i];
                stp_drec d;
                d.ud = rc.ud; d.x = rc.x; d.y = rc.y; d.w = rc.w; d.h = rc.h; d.pad0 = d.pad1 = d.pad2 = 0;
                d.total = tot;
                out[k] = d;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (tid == 0) rec_count[img] = nrec;
}

// Record compaction: per-image slots -> one dense array in the reference's row order (image index
// = (frame, level, brightness) order; slot order inside an image).  Workgroup g owns 64 images: it sums
// the counts of all earlier images (its base offset), scans its own 64 counts, and copies with 16 lanes
// per image.  The last workgroup also publishes the total and the overflow flag.
#define CR_IPB 64
__global__ __launch_bounds__(1024) void k_compact_recs(const stp_drec* __restrict__ recs, const int32_t* __restrict__ cnt,
                                                        int nimg, int f0, int nlev, int nb, stp_stripe_rec* __restrict__ out,
                                                        long long cap, const long long* __restrict__ tot_in /* records / overflow of the earlier chunks */,
                                                        long long* __restrict__ tot_out /* ... including this chunk */,
                                                        int rcap /* slot stride */, int slots /* slots an image may use: <= rcap */)
{
    __shared__ int s_wsum[16], s_wover[16];
    __shared__ int s_cnt[CR_IPB], s_off[CR_IPB];
    __shared__ long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i0 = blockIdx.x * CR_IPB;
    const bool last = (blockIdx.x == gridDim.x - 1);
    // base = sum of the (clamped) counts of images [0, i0); the last block also looks for overflow anywhere
    int loc = 0, over = 0;
    const int lim = last ? nimg : i0;
    for (int i = tid; i < lim; i += 1024) {
        int c = cnt[i];
        if (c > slots) { c = slots; over = 1; }
        if (i < i0) loc += c;
    }
    for (int o = 32; o > 0; o >>= 1) { loc += __shfl_xor(loc, o); over |= __shfl_xor(over, o); }
    if (lane == 0) { s_wsum[wv] = loc; s_wover[wv] = over; }
    if (tid < CR_IPB) {
        int c = (i0 + tid < nimg) ? cnt[i0 + tid] : 0;
        s_cnt[tid] = c > slots ? slots : c;
    }
    __syncthreads();
    if (tid < 64) {                                   // one wave: base + exclusive scan of the 64 counts
        int b = (lane < 16) ? s_wsum[lane] : 0, ov = (lane < 16) ? s_wover[lane] : 0;
        for (int o = 32; o > 0; o >>= 1) { b += __shfl_xor(b, o); ov |= __shfl_xor(ov, o); }
        const int c = s_cnt[lane];
        int incl = c;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        s_off[lane] = incl - c;
        if (lane == 63) {
            s_base = tot_in[0] + b;                   // the chunks of a search append to one output array
            if (last) { tot_out[0] = tot_in[0] + (long long)b + incl; tot_out[1] = tot_in[1] | ov; }
        }
    }
    __syncthreads();
    const int il = tid >> 4, k0 = tid & 15, i = i0 + il;
    if (i >= nimg) return;
    const int c = s_cnt[il];
    const long long pos0 = s_base + s_off[il];
    const int ipf = nlev * nb;
    const int fl = i / ipf, lev = (i % ipf) / nb, bi = i % nb;
    for (int k = k0; k < c; k += 16) {
        const long long pos = pos0 + k;
        if (pos >= cap) continue;
        const stp_drec d = recs[(size_t)i * rcap + k];
        stp_stripe_rec r;
        r.frame = f0 + fl; r.level = lev; r.b_index = bi; r.ud = d.ud;
        r.x = d.x; r.y = d.y; r.w = d.w; r.h = d.h; r.total = d.total;
        out[pos] = r;
    }
}

// Band packer: one lane per stored pixel of cooler's upper-triangular table; every pixel lands in two band
// cells (itself and its mirror image).  The band was zeroed first; cells no pixel names stay 0.
// value = count * (b[bin1] * b[bin2]): cooler's dense read multiplies the count matrix by np.outer(bias1, bias2).
// near[0..nrows) / near[nrows..2 nrows): distance from each bin to the nearest stored pixel with a positive value
// in its row of the symmetric matrix, to the right (column >= row) / to the left (column < row) -- every cis pixel
// takes part, also those beyond the band's halfwidth.
// Band cells no stored pixel names: cooler multiplies the DENSE count block by np.outer(bias1, bias2), so such a cell is
// 0 * (b[row] * b[col]) -- NaN along the whole row and column of a bin whose weight is NaN (or whose product overflows),
// 0 elsewhere; cells outside the chromosome are 0.  One lane per two cells (16 B stores), rows coalesced.
__global__ __launch_bounds__(256) void k_band_init(const double* __restrict__ wloc, int64_t nrows, int W, int hw,
                                                    double* __restrict__ band)
{
    const int64_t n2 = nrows * (int64_t)(W / 2);
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n2; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = p / (W / 2);
        const int c2 = (int)(p - r * (W / 2)) * 2;
        const double wr = wloc[r];
        double2 v;
        const int64_t ca = r + c2 - hw, cb = ca + 1;
        v.x = (ca >= 0 && ca < nrows) ? 0.0 * (wr * wloc[ca]) : 0.0;
        v.y = (cb >= 0 && cb < nrows) ? 0.0 * (wr * wloc[cb]) : 0.0;
        *(double2*)(band + r * (int64_t)W + c2) = v;
    }
}

// bin1_id of the pixels [p0, p0 + n) of a chunk from cooler's CSR index (indexes/bin1_offset): row r owns the pixels
// off[r] .. off[r + 1] - 1.  One wave per row of [r_lo, r_hi] (the rows that reach into the chunk: found on the host), lanes
// over the row's pixels: coalesced 8-byte stores, no search.  The column itself (8 bytes per pixel) never crosses PCIe.
__global__ __launch_bounds__(256) void k_expand_bin1(const int64_t* __restrict__ off, int64_t r_lo, int64_t r_hi, int64_t lo, int64_t p0,
                                                      int64_t n, int64_t* __restrict__ bin1)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = r_lo + wave; r <= r_hi; r += nwaves) {
        int64_t a = off[r] - p0, b = off[r + 1] - p0;
        a = a < 0 ? 0 : a; b = b > n ? n : b;
        for (int64_t p = a + lane; p < b; p += 64) bin1[p] = lo + r;
    }
}

template <typename CT, typename I2>     // pixels/count as stored: int32 or float64; bin2_id: int64 or int32
__global__ __launch_bounds__(256) void k_band_pack(const int64_t* __restrict__ bin1, const I2* __restrict__ bin2,
                                                    const CT* __restrict__ count, int64_t npix,
                                                    const double* __restrict__ wloc /* bias of [lo, lo+nrows) or null */,
                                                    int64_t lo, int64_t nrows, int W, int hw, double* __restrict__ band,
                                                    int32_t* __restrict__ near)
{
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // (whole waves walk the loop together: the trip count is rounded up per wave, lanes beyond the table are masked)
    for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); p0 < npix; p0 += stride) {
        const int64_t p = p0 + lane;
        bool ok = p < npix;
        int64_t i = 0, j = 0;
        if (ok) {
            i = bin1[p] - lo; j = (int64_t)bin2[p] - lo;
            ok = !(i < 0 || j < 0 || i >= nrows || j >= nrows);
        }
        if (j < i) { const int64_t t = i; i = j; j = t; }
        const int64_t d = j - i;
        double v = 0.0;
        if (ok) {
            v = (double)count[p];
            if (wloc) v = v * (wloc[i] * wloc[j]);
        }
        // nearest positive pixel to the right of row i / to the left of row j.  Round 5: the table is sorted by bin1, so the
        // positive pixels of a wave nearly always share ONE row -- 64 atomics on one address, which the memory system takes one
        // after the other (3.9 of the kernel's 4.2 ms per 20 M pixels).  The wave reduces its distances first and sends one.
        const bool pos = ok && v > 0.0;
#if !defined(STP_ABLATE_PACK) || STP_ABLATE_PACK != 2
        const unsigned long long act = __ballot(pos);
        if (act) {
            const int lead = __ffsll((long long)act) - 1;
            const int ilead = __shfl((int)i, lead);
            if (__ballot(pos && (int)i != ilead) == 0ull) {
                int m = pos ? (int)d : 0x7FFFFFFF;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
                if (lane == lead) atomicMin(&near[ilead], m);
            } else if (pos) {
                atomicMin(&near[i], (int32_t)d);                   // (a wave that straddles a row boundary: one per lane, as before)
            }
        }
        if (pos && d > 0) atomicMin(&near[nrows + j], (int32_t)d);    // row j holds the mirror image d columns to the left (distinct addresses within a wave)
#endif
        if (!ok || d > hw) continue;
        if (d < hw) band[i * (int64_t)W + (d + hw)] = v;
#if !defined(STP_ABLATE_PACK) || STP_ABLATE_PACK != 1
        band[j * (int64_t)W + (hw - d)] = v;
#endif
    }
}

// ============================================================================================
// host side
// ============================================================================================
// Diagnostics.  STP_LOG_ALLOC=1: every device allocation / release and every host range pinned in place is written to stderr
// with the source line that asked for it.  Always: the same events are kept in a small registry, and a handler registered
// with the HSA runtime (hsa_amd_register_system_event_handler, looked up at run time) writes a GPU memory fault's address
// and reason to stderr -- and to the file STP_FAULT_LOG names, which survives a test runner's capture -- together with the
// buffer of this library the address falls into (or the nearest ones, or a recently released one), before the runtime's
// own handler aborts the process.  STP_NO_FAULT_REPORT=1 leaves the runtime's handler alone.
static bool stp_log_on() { static const bool on = getenv("STP_LOG_ALLOC") != nullptr; return on; }
struct stp_alloc_rec { const char* p; size_t n; int line; char kind; };     // kind: 'd' device, 'h' pinned host range
static std::mutex g_reg_mutex;
static std::vector<stp_alloc_rec> g_live, g_dead;                             // g_dead: the last releases (ring of 64)
static void reg_add(void* p, size_t n, int line, char kind)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    g_live.push_back(stp_alloc_rec{(const char*)p, n, line, kind});
}
static void reg_del(void* p, int line)
{
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    for (size_t i = g_live.size(); i-- > 0;)
        if (g_live[i].p == (const char*)p) {
            stp_alloc_rec r = g_live[i];
            r.line = line;                                                    // (where it was released)
            g_live[i] = g_live.back(); g_live.pop_back();
            if (g_dead.size() >= 64) g_dead.erase(g_dead.begin());
            g_dead.push_back(r);
            return;
        }
}
static void fault_report(FILE* f, unsigned long long va, unsigned reason, int type)
{
    fprintf(f, "[stp] GPU %s: address 0x%llx, reason mask 0x%x%s%s%s%s\n", type == 0 ? "memory fault" : (type == 1 ? "hardware exception" : "memory error"),
            va, reason, (reason & 1) ? " (page not present)" : "", (reason & 2) ? " (write to a read-only page)" : "",
            (reason & 8) ? " (host-only page)" : "", (reason & 32) ? " (imprecise address)" : "");
    const stp_alloc_rec *in = nullptr, *below = nullptr, *above = nullptr;
    for (const auto& r : g_live) {
        const unsigned long long a = (unsigned long long)(uintptr_t)r.p, b = a + r.n;
        if (va >= a && va < b) in = &r;
        else if (b <= va && (!below || (unsigned long long)(uintptr_t)below->p + below->n < b)) below = &r;
        else if (a > va && (!above || (unsigned long long)(uintptr_t)above->p > a)) above = &r;
    }
    if (in) fprintf(f, "[stp]   inside the %s buffer of line %d: %p + %llu of %zu bytes\n", in->kind == 'd' ? "device" : "pinned host", in->line, (const void*)in->p,
                    va - (unsigned long long)(uintptr_t)in->p, in->n);
    if (!in && below) fprintf(f, "[stp]   %llu bytes past the end of the %s buffer of line %d (%p, %zu bytes)\n",
                              va - ((unsigned long long)(uintptr_t)below->p + below->n), below->kind == 'd' ? "device" : "pinned host", below->line, (const void*)below->p, below->n);
    if (!in && above) fprintf(f, "[stp]   %llu bytes before the %s buffer of line %d (%p, %zu bytes)\n",
                              (unsigned long long)(uintptr_t)above->p - va, above->kind == 'd' ? "device" : "pinned host", above->line, (const void*)above->p, above->n);
    for (const auto& r : g_dead) {
        const unsigned long long a = (unsigned long long)(uintptr_t)r.p;
        if (va >= a && va < a + r.n) fprintf(f, "[stp]   inside a %s buffer RELEASED at line %d: %p + %llu of %zu bytes\n", r.kind == 'd' ? "device" : "pinned host", r.line,
                                               (const void*)r.p, va - a, r.n);
    }
    fprintf(f, "[stp]   %zu live buffers of this library\n", g_live.size());
    fflush(f);
}
static hsa_status_t stp_hsa_event(const hsa_amd_event_t* ev, void*)
{
    unsigned long long va = 0; unsigned reason = 0;
    if (ev->event_type == HSA_AMD_GPU_MEMORY_FAULT_EVENT) { va = ev->memory_fault.virtual_address; reason = ev->memory_fault.fault_reason_mask; }
    else if (ev->event_type == HSA_AMD_GPU_MEMORY_ERROR_EVENT) { va = ev->memory_error.virtual_address; reason = ev->memory_error.error_reason_mask; }
    else if (ev->event_type == HSA_AMD_GPU_HW_EXCEPTION_EVENT) reason = (unsigned)ev->hw_exception.reset_cause;
    else return HSA_STATUS_ERROR;
    std::unique_lock<std::mutex> lk(g_reg_mutex, std::try_to_lock);           // (a faulting process: report even without the lock)
    fault_report(stderr, va, reason, (int)ev->event_type);
    if (const char* path = getenv("STP_FAULT_LOG"))
        if (FILE* f = fopen(path, "a")) { fault_report(f, va, reason, (int)ev->event_type); fclose(f); }
    return HSA_STATUS_ERROR;                                                  // not handled: the runtime's own report and abort follow
}
static void stp_install_fault_report()
{
    static std::once_flag once;
    std::call_once(once, [] {
        if (getenv("STP_NO_FAULT_REPORT")) return;
        void* h = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_NOLOAD);    // the runtime HIP has loaded already
        if (!h) return;
        typedef hsa_status_t (*reg_t)(hsa_amd_system_event_callback_t, void*);
        reg_t reg = (reg_t)dlsym(h, "hsa_amd_register_system_event_handler");
        const int rc = reg ? (int)reg(&stp_hsa_event, nullptr) : -1;          // (fails when somebody else has one: theirs stays)
        if (stp_log_on()) fprintf(stderr, "[stp] GPU fault report handler: %s (%d)\n", rc == 0 ? "installed" : "not installed", rc);
    });
}
static hipError_t stp_dmalloc(int line, void** p, size_t n)
{
    const hipError_t e = hipMalloc(p, n);
    if (e == hipSuccess) reg_add(*p, n, line, 'd');
    if (stp_log_on()) fprintf(stderr, "[stp] malloc L%d %p .. %p (%zu B) rc %d\n", line, *p, (void*)((char*)*p + n), n, (int)e);
    return e;
}
static hipError_t stp_dfree(int line, void* p)
{
    if (stp_log_on()) fprintf(stderr, "[stp] free   L%d %p\n", line, p);
    reg_del(p, line);
    return hipFree(p);
}
static hipError_t stp_hreg(int line, void* p, size_t n, unsigned flags)
{
    const hipError_t e = hipHostRegister(p, n, flags);
    if (e == hipSuccess) reg_add(p, n, line, 'h');
    if (stp_log_on()) fprintf(stderr, "[stp] hreg   L%d %p .. %p (%zu B) rc %d\n", line, p, (void*)((char*)p + n), n, (int)e);
    return e;
}
static hipError_t stp_hfree(int line, void* p)
{
    reg_del(p, line);
    return hipHostFree(p);
}
static hipError_t stp_hunreg(int line, void* p)
{
    reg_del(p, line);
    const hipError_t e = hipHostUnregister(p);
    if (stp_log_on()) fprintf(stderr, "[stp] hunreg L%d %p rc %d\n", line, p, (int)e);
    return e;
}

struct stp_kstat {
    std::string name;
    int64_t launches = 0;
    double ms = 0.0;
    double bytes = 0.0;
};

struct stp_pending {
    hipEvent_t e0, e1;
    const char* name;
    double bytes;
    bool own = true;          // false: the events are shared marks of a chain (prof_marks), destroyed once through stp_ctx::shared_events
};

enum { WS_GRAY = 0, WS_LOW, WS_HIGH, WS_RECS, WS_CNT, WS_OUT, WS_TOTAL, WS_PARAMS, WS_EDGES, WS_CELLS, WS_C32Q, WS_NSLOTS };

struct stp_ctx {
    int device = 0;
    std::vector<stp_pending> pending;
    std::vector<hipEvent_t> shared_events;      // chain marks several pending intervals refer to (destroyed after they are resolved)
    void* ws[WS_NSLOTS] = {nullptr};       // grow-only device workspace, reused across calls
    size_t ws_bytes[WS_NSLOTS] = {0};
    void* pin = nullptr;                   // pinned host staging buffer (records)
    size_t pin_bytes = 0;
    long long rec_guess = 0;               // records of the previous search chunk (+25 %): speculative D2H size
    // size-bucketed free lists: short-lived per-call device buffers are recycled instead of going through
    // hipMalloc / hipFree (both synchronise the device and cost ~0.1 ms each)
    std::vector<std::pair<size_t, void*>> pool_free;
    size_t pool_bytes = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t io = nullptr;              // band packing and the quantile's order-statistic select: PCIe uploads of the pixel
                                           // table and streaming kernels that must not queue behind the searches in flight
    hipStream_t aux = nullptr;             // frame compaction / medpixel and the per-stripe score kernels: small kernels
    hipStream_t pk = nullptr;              // the band packer's kernels (round 5): they run beside the next chunk's PCIe copies on `io`
                                           // with a host round trip each, which must not queue behind (and thereby
                                           // drain) the searches in flight on `stream`
    std::string err;
    bool profiling = false;
    std::vector<stp_kstat> stats;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int sweep_slots = STP_RCAP;            // record slots per image in the first pass (stp_dbg_set_sweep_slots)
    std::vector<std::pair<size_t, void*>> pin_free;   // pinned staging buffers of finished searches, recycled
    std::vector<std::pair<size_t, void*>> stage_free; // pinned staging buffers of stp_xfer (power-of-two sizes), recycled
    void* c32q_zero = nullptr; size_t c32q_zero_bytes = 0;   // the flag buffer known to be all zero (k_canny_pipe_list clears what it reads)
};

struct stp_band {
    const double* d = nullptr;
    int64_t nrows = 0;
    int hw = 0, W = 0;
    bool owned = false;
    int32_t* near = nullptr;   // 2 x nrows nearest-positive-pixel distances (bands built by stp_band_pack only)
    mutable int sym = -1;      // -1 not checked yet, 0 / 1: k_band_symcheck's verdict (the score kernels' coalesced reads).  A band
                               // wrapped around the caller's device memory must not be modified while its handle lives.
};

struct stp_frames {
    const stp_band* band = nullptr;
    int n = 0;
    int32_t *d_start = nullptr, *d_n0 = nullptr, *d_S = nullptr;
    int32_t* d_shift = nullptr;       // frame overlap (stp_phases.h): where the block shared with the next frame starts, or -1
    int16_t* d_nz = nullptr;
    std::vector<int32_t> h_start, h_n0, h_S, h_shift;
    std::vector<int16_t> h_nz;
    std::vector<double> h_med;
};

// Grow-only workspace shared by the searches of a context.  Several searches may be in flight on the context's stream
// and their queued kernels read these buffers, so a buffer is released only after the stream has drained -- waited
// for explicitly here (not left to hipFree's implicit device synchronisation).  Growth is rare: the chunk size is
// fixed (<= 3 072 images), so after the first full-size search of a context nothing grows again.
static hipError_t ws_get(stp_ctx* ctx, int slot, size_t bytes, void** out)
{
    if (ctx->ws_bytes[slot] < bytes) {
        if (ctx->ws[slot]) {
            hipError_t es = hipStreamSynchronize(ctx->stream);
            if (es != hipSuccess) return es;
            (void)stp_dfree(__LINE__, ctx->ws[slot]);
        }
        ctx->ws[slot] = nullptr; ctx->ws_bytes[slot] = 0;
        hipError_t e = stp_dmalloc(__LINE__, &ctx->ws[slot], bytes);
        if (e != hipSuccess) return e;
        ctx->ws_bytes[slot] = bytes;
    }
    *out = ctx->ws[slot];
    return hipSuccess;
}

static size_t pool_round(size_t n)
{
    size_t r = 256;
    while (r < n) r <<= 1;
    return r;
}
static hipError_t pool_alloc(stp_ctx* ctx, size_t bytes, void** out)
{
    const size_t r = pool_round(bytes ? bytes : 1);
    for (size_t i = 0; i < ctx->pool_free.size(); i++)
        if (ctx->pool_free[i].first == r) {
            *out = ctx->pool_free[i].second;
            ctx->pool_free[i] = ctx->pool_free.back();
            ctx->pool_free.pop_back();
            ctx->pool_bytes -= r;
            return hipSuccess;
        }
    return stp_dmalloc(__LINE__, out, r);
}
static void pool_release(stp_ctx* ctx, void* p, size_t bytes)
{
    if (!p) return;
    const size_t r = pool_round(bytes ? bytes : 1);
    if (ctx->pool_bytes + r > ((size_t)4 << 30)) { (void)stp_dfree(__LINE__, p); return; }   // keep at most 4 GiB idle (of 288)
    ctx->pool_free.push_back(std::make_pair(r, p));
    ctx->pool_bytes += r;
}

static hipError_t pin_get(stp_ctx* ctx, size_t bytes, void** out)
{
    if (ctx->pin_bytes < bytes) {
        if (ctx->pin) (void)stp_hfree(__LINE__, ctx->pin);
        ctx->pin = nullptr; ctx->pin_bytes = 0;
        hipError_t e = hipHostMalloc(&ctx->pin, bytes, hipHostMallocDefault);
        reg_add(ctx->pin, bytes, __LINE__, 'h');
        if (stp_log_on()) fprintf(stderr, "[stp] hostmalloc ctx %p (%zu B)\n", ctx->pin, bytes);
        if (e != hipSuccess) return e;
        ctx->pin_bytes = bytes;
    }
    *out = ctx->pin;
    return hipSuccess;
}

static int set_err(stp_ctx* c, int code, const std::string& m)
{
    if (c) c->err = m;
    return code;
}
#define HIPCHK(call)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return set_err(ctx, e_ == hipErrorOutOfMemory ? STP_E_NOMEM : STP_E_HIP,             \
                           std::string(#call) + ": " + hipGetErrorString(e_));                   \
    } while (0)

static stp_kstat& stat_for(stp_ctx* ctx, const char* name)
{
    for (auto& s : ctx->stats)
        if (s.name == name) return s;
    ctx->stats.push_back(stp_kstat());
    ctx->stats.back().name = name;
    return ctx->stats.back();
}

// Kernel timing without stalling the stream: each profiled launch records an event pair; the
// pairs are resolved (hipEventElapsedTime) when statistics are read.
struct prof_scope {
    stp_ctx* ctx;
    const char* name;
    double bytes;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st;
    prof_scope(stp_ctx* c, const char* n, double b, hipStream_t on = nullptr) : ctx(c), name(n), bytes(b), st(on ? on : c->stream)
    {
        if (!ctx->profiling) return;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
        (void)hipEventRecord(e0, st);
    }
    ~prof_scope()
    {
        if (!ctx->profiling || !e0) return;
        (void)hipEventRecord(e1, st);
        stp_pending p;
        p.e0 = e0; p.e1 = e1; p.name = name; p.bytes = bytes;
        ctx->pending.push_back(p);
    }
};

static void resolve_pending(stp_ctx* ctx)
{
    for (auto& p : ctx->pending) {
        (void)hipEventSynchronize(p.e1);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            stp_kstat& s = stat_for(ctx, p.name);
            s.launches++; s.ms += ms; s.bytes += p.bytes;
        }
        if (p.own) {
            (void)hipEventDestroy(p.e0);
            (void)hipEventDestroy(p.e1);
        }
    }
    ctx->pending.clear();
    for (hipEvent_t e : ctx->shared_events) (void)hipEventDestroy(e);
    ctx->shared_events.clear();
}

// The chain's timers as MARKS between its kernels (round 6): grey = [m0, m1], Canny = [m1, m2], line joining = [m2, m3], the chain =
// [m0, m3], the record compaction behind it = [m3, m4] -- five event records per unit instead of the ten that five separate scopes
// put into the stream (every record is a packet the kernels behind it wait for: the timers cost a genome step 0.8-1.6 ms,
// tools/ab_profiling.py).
struct prof_marks {
    stp_ctx* ctx;
    hipEvent_t m[8] = {};
    int n = 0;
    explicit prof_marks(stp_ctx* c) : ctx(c) {}
    int mark()                                      // returns the mark's index, or -1 (not profiling / no event)
    {
        if (!ctx->profiling || n >= 8) return -1;
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return -1;
        (void)hipEventRecord(e, ctx->stream);
        ctx->shared_events.push_back(e);
        m[n] = e;
        return n++;
    }
    void interval(int a, int b, const char* name, double bytes)
    {
        if (a < 0 || b < 0) return;
        stp_pending p;
        p.e0 = m[a]; p.e1 = m[b]; p.name = name; p.bytes = bytes; p.own = false;
        ctx->pending.push_back(p);
    }
};

extern "C" {
#pragma GCC visibility push(default)

int stp_version(void) { return STP_ABI_VERSION; }

int stp_ctx_create(int device_ordinal, stp_ctx** out)
{
    if (!out) return STP_E_ARG;
    *out = nullptr;
    stp_ctx* ctx = new (std::nothrow) stp_ctx();
    if (!ctx) return STP_E_NOMEM;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0 || device_ordinal < 0 || device_ordinal >= ndev) {
        delete ctx;
        return STP_E_HIP;   // no CPU fallback by design
    }
    ctx->device = device_ordinal;
    if (hipSetDevice(device_ordinal) != hipSuccess) { delete ctx; return STP_E_HIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) { delete ctx; return STP_E_HIP; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "libstripenn_hip: device %d is %s, this library is built for gfx950 only\n", device_ordinal,
                prop.gcnArchName);
        delete ctx;
        return STP_E_UNSUPPORTED;
    }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return STP_E_HIP; }
    ctx->own_stream = true;
    {
        // frame preparation feeds the next search and the score kernels are small: the auxiliary stream runs at the
        // highest priority, so its kernels get compute units as soon as the chain's workgroups retire
        int plo = 0, phi = 0;
        (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
        const char* ap = getenv("STP_AUX_PRIORITY");          // measurement hook: "low" / "normal" instead of the highest
        const int prio = (ap && strcmp(ap, "low") == 0) ? plo : ((ap && strcmp(ap, "normal") == 0) ? (plo + phi) / 2 : phi);
        if (hipStreamCreateWithPriority(&ctx->aux, hipStreamNonBlocking, prio) != hipSuccess) { (void)hipStreamDestroy(ctx->stream); delete ctx; return STP_E_HIP; }
    }
    if (hipStreamCreateWithFlags(&ctx->io, hipStreamNonBlocking) != hipSuccess) {
        (void)hipStreamDestroy(ctx->aux); (void)hipStreamDestroy(ctx->stream); delete ctx; return STP_E_HIP;
    }
    if (hipStreamCreateWithFlags(&ctx->pk, hipStreamNonBlocking) != hipSuccess) {
        (void)hipStreamDestroy(ctx->io); (void)hipStreamDestroy(ctx->aux); (void)hipStreamDestroy(ctx->stream); delete ctx; return STP_E_HIP;
    }
    (void)hipEventCreate(&ctx->ev0);
    (void)hipEventCreate(&ctx->ev1);
    stp_install_fault_report();
    *out = ctx;
    return STP_OK;
}

void stp_ctx_destroy(stp_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    resolve_pending(ctx);
    for (int i = 0; i < WS_NSLOTS; i++) if (ctx->ws[i]) (void)stp_dfree(__LINE__, ctx->ws[i]);
    for (auto& e : ctx->pool_free) (void)stp_dfree(__LINE__, e.second);
    if (ctx->pin) (void)stp_hfree(__LINE__, ctx->pin);
    for (auto& e : ctx->pin_free) (void)stp_hfree(__LINE__, e.second);
    for (auto& e : ctx->stage_free) (void)stp_hfree(__LINE__, e.second);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->aux) { (void)hipStreamSynchronize(ctx->aux); (void)hipStreamDestroy(ctx->aux); }
    if (ctx->io) { (void)hipStreamSynchronize(ctx->io); (void)hipStreamDestroy(ctx->io); }
    if (ctx->pk) { (void)hipStreamSynchronize(ctx->pk); (void)hipStreamDestroy(ctx->pk); }
    delete ctx;
}

const char* stp_last_error(const stp_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int stp_ctx_set_stream(stp_ctx* ctx, void* s)
{
    if (!ctx) return STP_E_ARG;
    if (ctx->own_stream && ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    if (s) { ctx->stream = (hipStream_t)s; ctx->own_stream = false; }
    else {
        HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return STP_OK;
}

int stp_ctx_synchronize(stp_ctx* ctx)
{
    if (!ctx) return STP_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return STP_OK;
}

static int check_hw(stp_ctx* ctx, int64_t nrows, int32_t hw)
{
    if (nrows <= 0 || hw < 448 || (hw % 64) != 0 || hw > 4096)
        return set_err(ctx, STP_E_ARG, "band: nrows must be > 0 and halfwidth a multiple of 64 in [448, 4096]");
    return STP_OK;
}

// The caller's (pageable) host array pinned in place for the duration of one call: the copies out of it then run as DMA at
// PCIe speed (57 GB/s measured on the MI355X box) instead of through the runtime's staging copies (11-20 GB/s);
// registering 2 GiB of touched pages takes 17 ms.  Only the WHOLE PAGES inside the range are registered (round 4): two
// ranges -- the three columns of a small table that sit next to each other in the heap, the slices of consecutive
// chromosomes out of one column, somebody else's buffer behind the array -- then never share a page, so releasing one
// range cannot touch a page another transfer still reads; the ragged head and tail (< 4 KB each) travel as pageable
// copies of their own (copy()).  A range whose pages are pinned already (the caller's own registration) is used as it
// is and left alone.  Failing to register (STP_NO_PIN=1, the runtime refusing) is not an error: the copies fall back to
// the pageable path.  The stream is drained before the range is released.
// (Measured and dropped in round 4: the driver pinning the NEXT chromosome's columns from a second host thread while the
//  current ones cross PCIe -- band packing stayed at 0.15-0.16 s for the 5.3 GB of the mm10-size table: the transfers,
//  not the page pinning between them, set that time.)
#define STP_PIN_PAGE ((uintptr_t)4096)
static bool stp_pin_body(const void* ptr, size_t bytes, char** lo, char** hi)
{
    static const bool off = getenv("STP_NO_PIN") != nullptr;
    if (off || !ptr || bytes < ((size_t)1 << 20)) return false;        // (smaller buffers travel through the library's staging: stp_xfer)
    const uintptr_t a = ((uintptr_t)ptr + STP_PIN_PAGE - 1) & ~(STP_PIN_PAGE - 1), b = ((uintptr_t)ptr + bytes) & ~(STP_PIN_PAGE - 1);
    if (b <= a) return false;
    *lo = (char*)a; *hi = (char*)b;
    return true;
}
static bool stp_is_pinned(const void* p)
{
    hipPointerAttribute_t at;          // (hipHostGetFlags answers for hipHostMalloc memory only)
    if (hipPointerGetAttributes(&at, p) == hipSuccess) return at.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    return false;
}
struct host_pin {
    char *lo = nullptr, *hi = nullptr;    // the pinned pages [lo, hi) inside the caller's range (none: lo == hi)
    bool mine = false;                    // registered here (released by the destructor)
    hipStream_t st = nullptr;
    void pin(const void* ptr, size_t bytes, hipStream_t stream)
    {
        st = stream;
        char *a, *b;
        if (!stp_pin_body(ptr, bytes, &a, &b)) return;
        if (stp_is_pinned(a) && stp_is_pinned(b - 1)) { lo = a; hi = b; return; }      // the caller's registration: nothing to do or undo
        if (stp_hreg(__LINE__, a, (size_t)(b - a), hipHostRegisterDefault) == hipSuccess) { lo = a; hi = b; mine = true; }
        else (void)hipGetLastError();                       // clear the sticky error: pageable copies from here on
    }
    // copy between the range given to pin() and device memory: the pinned pages as one transfer (DMA), what lies before and
    // after them as small pageable copies of their own (the runtime stages those)
    hipError_t copy(void* dst, const void* src, size_t bytes, hipStream_t stream, hipMemcpyKind kind = hipMemcpyHostToDevice) const
    {
        const bool up = kind == hipMemcpyHostToDevice;
        const char *h0 = (const char*)(up ? src : dst), *h1 = h0 + bytes;        // the host side
        const char *b0 = h0 > lo ? h0 : lo, *b1 = h1 < hi ? h1 : hi;
        if (lo == hi || b0 >= b1) return hipMemcpyAsync(dst, src, bytes, kind, stream);
        hipError_t e = hipSuccess;
        const size_t n0 = (size_t)(b0 - h0), n1 = (size_t)(b1 - b0), n2 = (size_t)(h1 - b1);
        if (n0) e = hipMemcpyAsync(dst, src, n0, kind, stream);
        if (e == hipSuccess) e = hipMemcpyAsync((char*)dst + n0, (const char*)src + n0, n1, kind, stream);
        if (e == hipSuccess && n2) e = hipMemcpyAsync((char*)dst + n0 + n1, (const char*)src + n0 + n1, n2, kind, stream);
        return e;
    }
    ~host_pin()
    {
        if (!mine) return;
        (void)hipStreamSynchronize(st);
        if (stp_hunreg(__LINE__, lo) != hipSuccess) (void)hipGetLastError();
    }
};

// Every transfer between the caller's (pageable) host memory and the device goes through here (round 4).  Handing pageable
// memory to hipMemcpyAsync lets the runtime pin it on the fly, and the runtime keeps such pins in a small per-queue cache
// keyed by the HOST ADDRESS: a buffer the caller has freed since -- numpy recycles heap addresses all the time -- still has its
// stale pin there, and a later transfer to a new buffer at the same address takes it.  A pin made for an upload is read-only
// to the device: the first download into the recycled address faulted ("write access to a read-only page", reported
// through STP_FAULT_LOG; three aborted runs in round 4 before the report caught it).  So the runtime never sees the
// caller's pointers unless the library has registered them itself:
//   * < 64 KB: the runtime's own staging path (no pinning at that size);
//   * up to 8 MB: through a pinned staging buffer of the context (one host copy; buffers are recycled);
//   * larger: the caller's pages registered in place for the call (host_pin), DMA straight from / to them.
// finish() drains the stream and completes the staged downloads; the destructor does it on the error paths.
#define STP_XFER_SMALL ((size_t)64 << 10)
#define STP_XFER_STAGED ((size_t)8 << 20)
struct stp_xfer {
    stp_ctx* ctx;
    hipStream_t st;
    struct piece { void* stage; size_t cap; void* user; size_t bytes; };        // user != null: a download to complete
    std::vector<piece> pieces;
    std::vector<host_pin*> pins;
    bool done = false;
    stp_xfer(stp_ctx* c, hipStream_t s) : ctx(c), st(s) {}
    void* stage_get(size_t bytes, size_t* cap)
    {
        const size_t r = pool_round(bytes);
        for (size_t i = 0; i < ctx->stage_free.size(); i++)
            if (ctx->stage_free[i].first == r) {
                void* p = ctx->stage_free[i].second;
                ctx->stage_free[i] = ctx->stage_free.back(); ctx->stage_free.pop_back();
                *cap = r;
                return p;
            }
        void* p = nullptr;
        if (hipHostMalloc(&p, r, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        reg_add(p, r, __LINE__, 'h');
        if (stp_log_on()) fprintf(stderr, "[stp] hostmalloc stage %p (%zu B)\n", p, r);
        *cap = r;
        return p;
    }
    hipError_t h2d(void* dst, const void* src, size_t bytes)
    {
        if (!bytes) return hipSuccess;
        if (bytes < STP_XFER_SMALL) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
        if (bytes <= STP_XFER_STAGED) {
            size_t cap = 0;
            void* p = stage_get(bytes, &cap);
            if (!p) return hipErrorOutOfMemory;
            memcpy(p, src, bytes);
            pieces.push_back(piece{p, cap, nullptr, 0});
            return hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, st);
        }
        host_pin* hp = new host_pin();
        pins.push_back(hp);
        hp->pin(src, bytes, st);
        if (hp->lo != hp->hi) return hp->copy(dst, src, bytes, st, hipMemcpyHostToDevice);
        return piecewise(dst, src, bytes, true);               // the pages could not be registered (STP_NO_PIN, the runtime refusing)
    }
    // a large buffer that cannot be pinned in place: through one staging buffer, piece by piece (slow path, synchronous)
    hipError_t piecewise(void* dst, const void* src, size_t bytes, bool up)
    {
        size_t cap = 0;
        void* p = stage_get(STP_XFER_STAGED, &cap);
        if (!p) return hipErrorOutOfMemory;
        hipError_t e = hipSuccess;
        for (size_t o = 0; o < bytes && e == hipSuccess; o += STP_XFER_STAGED) {
            const size_t n = bytes - o < STP_XFER_STAGED ? bytes - o : STP_XFER_STAGED;
            if (up) {
                memcpy(p, (const char*)src + o, n);
                e = hipMemcpyAsync((char*)dst + o, p, n, hipMemcpyHostToDevice, st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);
            } else {
                e = hipMemcpyAsync(p, (const char*)src + o, n, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);
                if (e == hipSuccess) memcpy((char*)dst + o, p, n);
            }
        }
        pieces.push_back(piece{p, cap, nullptr, 0});
        return e;
    }
    hipError_t d2h(void* dst, const void* src, size_t bytes)
    {
        if (!bytes) return hipSuccess;
        if (bytes < STP_XFER_SMALL) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
        if (bytes <= STP_XFER_STAGED) {
            size_t cap = 0;
            void* p = stage_get(bytes, &cap);
            if (!p) return hipErrorOutOfMemory;
            pieces.push_back(piece{p, cap, dst, bytes});
            return hipMemcpyAsync(p, src, bytes, hipMemcpyDeviceToHost, st);
        }
        host_pin* hp = new host_pin();
        pins.push_back(hp);
        hp->pin(dst, bytes, st);
        if (hp->lo != hp->hi) return hp->copy(dst, src, bytes, st, hipMemcpyDeviceToHost);
        return piecewise(dst, src, bytes, false);
    }
    // finish(): the explicit, successful end of a call -- the stream is drained and the staged downloads are delivered into the
    // caller's buffers.  The destructor alone (an early error return) only drains the stream and recycles the staging buffers:
    // by then a download's destination may be gone (a vector declared after the transfer object, a frames object already
    // deleted), so nothing is copied into `piece.user`.
    hipError_t finish(bool deliver = true)
    {
        if (done) return hipSuccess;
        done = true;
        const hipError_t e = hipStreamSynchronize(st);
        for (auto& pc : pieces) {
            if (deliver && e == hipSuccess && pc.user) memcpy(pc.user, pc.stage, pc.bytes);
            if (ctx->stage_free.size() < 32) ctx->stage_free.push_back(std::make_pair(pc.cap, pc.stage));
            else (void)stp_hfree(__LINE__, pc.stage);
        }
        pieces.clear();
        for (host_pin* hp : pins) delete hp;                                    // (drains the stream -- done -- and releases the pages)
        pins.clear();
        return e;
    }
    ~stp_xfer() { (void)finish(false); }
};

int stp_band_upload(stp_ctx* ctx, const double* band_host, int64_t nrows, int32_t hw, stp_band** out)
{
    if (!ctx || !band_host || !out) return STP_E_ARG;
    int rc = check_hw(ctx, nrows, hw);
    if (rc) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    stp_band* b = new (std::nothrow) stp_band();
    if (!b) return STP_E_NOMEM;
    b->nrows = nrows; b->hw = hw; b->W = 2 * hw; b->owned = true;
    double* d = nullptr;
    size_t bytes = (size_t)nrows * b->W * sizeof(double);
    hipError_t e = stp_dmalloc(__LINE__, (void**)&d, bytes);
    if (e != hipSuccess) { delete b; return set_err(ctx, STP_E_NOMEM, "hipMalloc(band) failed"); }
    {
        stp_xfer x(ctx, ctx->stream);
        e = x.h2d(d, band_host, bytes);
        if (e == hipSuccess) e = x.finish();
    }
    if (e != hipSuccess) { (void)stp_dfree(__LINE__, d); delete b; return set_err(ctx, STP_E_HIP, "band upload failed"); }
    b->d = d;
    *out = b;
    return STP_OK;
}

int stp_band_wrap_device(stp_ctx* ctx, const void* dptr, int64_t nrows, int32_t hw, stp_band** out)
{
    if (!ctx || !dptr || !out) return STP_E_ARG;
    int rc = check_hw(ctx, nrows, hw);
    if (rc) return rc;
    stp_band* b = new (std::nothrow) stp_band();
    if (!b) return STP_E_NOMEM;
    b->d = (const double*)dptr; b->nrows = nrows; b->hw = hw; b->W = 2 * hw; b->owned = false;
    *out = b;
    return STP_OK;
}

void stp_band_free(stp_ctx* ctx, stp_band* b)
{
    if (!b) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (b->owned && b->d) (void)stp_dfree(__LINE__, (void*)b->d);
    if (b->near) (void)stp_dfree(__LINE__, b->near);
    delete b;
}

// Frame overlap (stp_phases.h): the mark of every frame from the compaction k_frame_prep has just written (same stream), one WAVE
// per frame: stp_overlap_shift's tests with the lanes side by side (a lane per frame walked its 400-entry lists alone: 0.25 ms
// per launch, in front of the blocking copy every frame preparation ends with).  Second rule (the block frame i takes from frame
// i + 1 must not reach into the block frame i + 1 takes from frame i + 2): both marks are formed by the wave that needs them.
__device__ __forceinline__ int overlap_shift_wave(int s0, int n0, int S0, const int16_t* __restrict__ z0, int s1, int n1, int S1,
                                                  const int16_t* __restrict__ z1, int lane)
{
    const int e0 = s0 + n0 - 1, e1 = s1 + n1 - 1;
    if (!(s1 > s0 && s1 <= e0 && e1 >= e0) || S0 <= 0 || S1 <= 0) return -1;          // (wave-uniform throughout)
    int p = 0;                                                                          // kept bins of frame i before frame i + 1 starts (z0 ascends)
    for (int k0 = 0; k0 < S0; k0 += 64) {
        const int k = k0 + lane;
        p += __popcll(__ballot(k < S0 && s0 + z0[k] < s1));
    }
    const int q = S0 - p;
    if (q < 1 || q > S1) return -1;
    bool diff = false;
    for (int k0 = 0; k0 < q; k0 += 64) {
        const int k = k0 + lane;
        diff = diff || (k < q && s0 + z0[p + k] != s1 + z1[k]);
    }
    if (__any(diff)) return -1;
    if (q < S1 && s1 + z1[q] <= e0) return -1;
    return p;
}
__global__ __launch_bounds__(64) void k_overlap_shift(const int32_t* __restrict__ fstart, const int32_t* __restrict__ fn0,
                                                      const int32_t* __restrict__ fS, const int16_t* __restrict__ fnz, int n,
                                                      int32_t* __restrict__ shift)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    int p = -1;
    if (i + 1 < n) {
        p = overlap_shift_wave(fstart[i], fn0[i], fS[i], fnz + (size_t)i * STP_FRAME_MAX, fstart[i + 1], fn0[i + 1], fS[i + 1],
                               fnz + (size_t)(i + 1) * STP_FRAME_MAX, lane);
        if (p >= 0 && i + 2 < n) {
            const int p1 = overlap_shift_wave(fstart[i + 1], fn0[i + 1], fS[i + 1], fnz + (size_t)(i + 1) * STP_FRAME_MAX, fstart[i + 2],
                                              fn0[i + 2], fS[i + 2], fnz + (size_t)(i + 2) * STP_FRAME_MAX, lane);
            if (p1 >= 0 && p1 < fS[i] - p) p = -1;
        }
    }
    if (lane == 0) shift[i] = p;
}

void stp_frames_free(stp_ctx* ctx, stp_frames* fr)
{
    if (!fr) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (ctx) {
        pool_release(ctx, fr->d_start, fr->n * sizeof(int32_t));
        pool_release(ctx, fr->d_n0, fr->n * sizeof(int32_t));
        pool_release(ctx, fr->d_S, fr->n * sizeof(int32_t));
        pool_release(ctx, fr->d_shift, fr->n * sizeof(int32_t));
        pool_release(ctx, fr->d_nz, (size_t)fr->n * STP_FRAME_MAX * sizeof(int16_t));
    }
    delete fr;
}

int stp_frames_create(stp_ctx* ctx, const stp_band* band, const int32_t* start, const int32_t* end, int32_t n,
                      stp_frames** out)
{
    return stp_frames_create_ex(ctx, band, start, end, n, 0, out);
}

int stp_frames_create_ex(stp_ctx* ctx, const stp_band* band, const int32_t* start, const int32_t* end, int32_t n,
                         int32_t flags, stp_frames** out)
{
    if (!ctx || !band || !start || !end || !out || n <= 0) return STP_E_ARG;
    if (flags & ~STP_FRAMES_KEEP_ALL) return set_err(ctx, STP_E_ARG, "stp_frames_create_ex: unknown flag");
    for (int i = 0; i < n; i++) {
        int n0 = end[i] - start[i] + 1;
        if (start[i] < 0 || end[i] >= band->nrows || n0 < 1 || n0 > STP_FRAME_MAX)
            return set_err(ctx, STP_E_ARG, "frame " + std::to_string(i) + " out of range or larger than 400");
        if (n0 > band->hw) return set_err(ctx, STP_E_ARG, "frame wider than band halfwidth");
    }
    HIPCHK(hipSetDevice(ctx->device));
    stp_frames* fr = new (std::nothrow) stp_frames();
    if (!fr) return STP_E_NOMEM;
    fr->band = band; fr->n = n;
    fr->h_start.assign(start, start + n);
    fr->h_n0.resize(n);
    for (int i = 0; i < n; i++) fr->h_n0[i] = end[i] - start[i] + 1;
    fr->h_S.resize(n); fr->h_nz.assign((size_t)n * STP_FRAME_MAX, 0); fr->h_med.assign(n, 0.0);
#define FRCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { stp_frames_free(ctx, fr); \
        return set_err(ctx, e_ == hipErrorOutOfMemory ? STP_E_NOMEM : STP_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } } while (0)
    FRCHK(pool_alloc(ctx, n * sizeof(int32_t), (void**)&fr->d_start));
    FRCHK(pool_alloc(ctx, n * sizeof(int32_t), (void**)&fr->d_n0));
    FRCHK(pool_alloc(ctx, n * sizeof(int32_t), (void**)&fr->d_S));
    FRCHK(pool_alloc(ctx, n * sizeof(int32_t), (void**)&fr->d_shift));
    FRCHK(pool_alloc(ctx, (size_t)n * STP_FRAME_MAX * sizeof(int16_t), (void**)&fr->d_nz));
    stp_xfer x(ctx, ctx->aux);
    FRCHK(x.h2d(fr->d_start, fr->h_start.data(), n * sizeof(int32_t)));
    FRCHK(x.h2d(fr->d_n0, fr->h_n0.data(), n * sizeof(int32_t)));
    FRCHK(hipMemsetAsync(fr->d_nz, 0, (size_t)n * STP_FRAME_MAX * sizeof(int16_t), ctx->aux));
    double* d_med = nullptr;
    FRCHK(pool_alloc(ctx, (size_t)n * 3 * sizeof(double), (void**)&d_med));
    {
        // zero-column removal + medpixel (two order statistics per frame by radix select, numpy's lerp on the host) in
        // one kernel: the frame is read twice (walk + gather) instead of four times
        double bytes = 0;
        for (int i = 0; i < n; i++) bytes += 8.0 * fr->h_n0[i] * fr->h_n0[i];
        prof_scope ps(ctx, "frame_prep", bytes, ctx->aux);
        stp_bandref B{band->d, band->nrows, band->W, band->hw};
        hipLaunchKernelGGL(k_frame_prep, dim3(n), dim3(STP_PREP_NT), 0, ctx->aux, B, fr->d_start, fr->d_n0, fr->d_S, fr->d_nz, d_med,
                           (flags & STP_FRAMES_KEEP_ALL) ? 1 : 0);
        hipLaunchKernelGGL(k_overlap_shift, dim3(n), dim3(64), 0, ctx->aux, fr->d_start, fr->d_n0, fr->d_S, fr->d_nz, n, fr->d_shift);
    }
    {
        hipError_t el = hipGetLastError();
        if (el != hipSuccess) pool_release(ctx, d_med, (size_t)n * 3 * sizeof(double));
        FRCHK(el);
    }
    std::vector<double> hm((size_t)n * 3);
    hipError_t e1 = x.d2h(hm.data(), d_med, hm.size() * sizeof(double));
    if (e1 == hipSuccess) e1 = x.d2h(fr->h_S.data(), fr->d_S, n * sizeof(int32_t));
    if (e1 == hipSuccess) e1 = x.d2h(fr->h_nz.data(), fr->d_nz, (size_t)n * STP_FRAME_MAX * sizeof(int16_t));
    if (e1 == hipSuccess) e1 = x.finish();
    pool_release(ctx, d_med, (size_t)n * 3 * sizeof(double));
    FRCHK(e1);
    for (int i = 0; i < n; i++) {
        const double a = hm[3 * i], b = hm[3 * i + 1], N = hm[3 * i + 2];
        if (N < 1) { fr->h_med[i] = NAN; continue; }
        // numpy _lerp with t = frac((N-1)*0.5): t = 0 -> a ; t = 0.5 -> b - (b-a)*(1-t)
        const bool even = (((long long)N) % 2) == 0;
        fr->h_med[i] = even ? (b - (b - a) * (1 - 0.5)) : (a + (b - a) * 0.0);
    }
    // Frame overlap (stp_phases.h, "frame overlap"): frame i is marked when its trailing kept bins are exactly the leading kept
    // bins of frame i + 1 -- the same absolute bins, one contiguous run in both compacted frames -- and the block it would
    // take from frame i + 1 does not reach into the block frame i + 1 takes from frame i + 2.
    // (the device marks its copy itself -- k_overlap_shift behind k_frame_prep, same code -- so the marks cost no second round trip)
    fr->h_shift.assign(n, -1);
    for (int i = 0; i + 1 < n; i++)
        fr->h_shift[i] = stp_overlap_shift(fr->h_start[i], fr->h_n0[i], fr->h_S[i], fr->h_nz.data() + (size_t)i * STP_FRAME_MAX,
                                           fr->h_start[i + 1], fr->h_n0[i + 1], fr->h_S[i + 1], fr->h_nz.data() + (size_t)(i + 1) * STP_FRAME_MAX);
    for (int i = 0; i + 2 < n; i++)                                    // (p of frame i + 1 == q of frame i in the reference's geometry)
        if (fr->h_shift[i] >= 0 && fr->h_shift[i + 1] >= 0 && fr->h_shift[i + 1] < fr->h_S[i] - fr->h_shift[i]) fr->h_shift[i] = -1;
#undef FRCHK
    *out = fr;
    return STP_OK;
}

int stp_frames_info(stp_ctx* ctx, const stp_frames* fr, int32_t* S_out, int16_t* nz_out, double* med_out)
{
    if (!ctx || !fr) return STP_E_ARG;
    if (S_out) memcpy(S_out, fr->h_S.data(), fr->n * sizeof(int32_t));
    if (nz_out) memcpy(nz_out, fr->h_nz.data(), (size_t)fr->n * STP_FRAME_MAX * sizeof(int16_t));
    if (med_out) memcpy(med_out, fr->h_med.data(), fr->n * sizeof(double));
    return STP_OK;
}

int stp_frames_overlap(stp_ctx* ctx, const stp_frames* fr, int32_t* shift_out)
{
    if (!ctx || !fr || !shift_out) return STP_E_ARG;
    // the marks the kernels use are the DEVICE's (k_overlap_shift); the host forms its own from the same rule: they are compared
    // here, so that a caller of this (test / diagnostic) entry point sees a disagreement instead of the host's opinion
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<int32_t> dev(fr->n);
    {
        stp_xfer x(ctx, ctx->aux);
        HIPCHK(x.d2h(dev.data(), fr->d_shift, fr->n * sizeof(int32_t)));
        HIPCHK(x.finish());
    }
    for (int i = 0; i < fr->n; i++)
        if (dev[i] != fr->h_shift[i])
            return set_err(ctx, STP_E_HIP, "frame overlap: device mark of frame " + std::to_string(i) + " = " + std::to_string(dev[i]) +
                                           ", host " + std::to_string(fr->h_shift[i]));
    memcpy(shift_out, dev.data(), fr->n * sizeof(int32_t));
    return STP_OK;
}

static int check_params(stp_ctx* ctx, const stp_search_params* p)
{
    if (!p || !p->bright || !p->gauss_w) return set_err(ctx, STP_E_ARG, "null search params");
    if (p->bfilter < 1 || p->bfilter > 2 * GT_AMAX + 1 || (p->bfilter % 2) == 0)
        return set_err(ctx, STP_E_UNSUPPORTED, "bfilter must be odd and <= 7");
    if (p->n_bright < 1 || p->n_bright > 8) return set_err(ctx, STP_E_ARG, "n_bright must be in 1..8");
    if (p->gauss_radius < 1 || p->gauss_radius > CT_RMAX)
        return set_err(ctx, STP_E_UNSUPPORTED, "gaussian radius must be in 1..12 (sigma <= 3.1)");
    if (p->minH < 0 || p->maxW < 2 || p->maxW > STP_FRAME_MAX) return set_err(ctx, STP_E_ARG, "bad minH/maxW");
    return STP_OK;
}

struct dev_buf {                      // per-call device buffer, recycled through the ctx pool
    void* p = nullptr;
    size_t n = 0;
    stp_ctx* c = nullptr;
    ~dev_buf() { if (p) { if (c) pool_release(c, p, n); else (void)stp_dfree(__LINE__, p); } }
    hipError_t alloc(stp_ctx* ctx, size_t bytes) { c = ctx; n = bytes; return pool_alloc(ctx, bytes, &p); }
};

struct stp_select {
    std::vector<std::pair<double*, long long>> chunks;   // device buffers (from the context's pool), number of values
    long long npos = -1;                                 // cached count of positive values
    stp_sel_state* state = nullptr;
};
static unsigned sel_grid(long long n);

int stp_band_pack(stp_ctx* ctx, const int64_t* bin1, const int64_t* bin2, const int32_t* count, int64_t npix,
                  const double* weight, int64_t nbins_total, int64_t lo, int64_t nrows, int32_t hw, stp_band** out)
{
    return stp_band_pack_select(ctx, bin1, bin2, count, STP_COUNT_I32, npix, weight, nbins_total, lo, nrows, hw, nullptr, out);
}

static int band_pack_impl(stp_ctx* ctx, const int64_t* bin1, const int64_t* off, const void* bin2_v, int32_t id2_type, const void* count_v,
                          int32_t count_type, int64_t npix, const double* weight, int64_t nbins_total, int64_t lo, int64_t nrows, int32_t hw,
                          stp_select* sel, stp_band** out);

int stp_band_pack_select(stp_ctx* ctx, const int64_t* bin1, const int64_t* bin2, const void* count_v, int32_t count_type,
                         int64_t npix, const double* weight, int64_t nbins_total, int64_t lo, int64_t nrows, int32_t hw,
                         stp_select* sel, stp_band** out)
{
    if (!ctx || (npix > 0 && !bin1)) return STP_E_ARG;
    return band_pack_impl(ctx, bin1, nullptr, bin2, STP_ID_I64, count_v, count_type, npix, weight, nbins_total, lo, nrows, hw, sel, out);
}

int stp_band_pack_csr(stp_ctx* ctx, const int64_t* bin1_offset, const void* bin2, int32_t bin2_type, const void* count_v, int32_t count_type,
                      int64_t npix, const double* weight, int64_t nbins_total, int64_t lo, int64_t nrows, int32_t hw, stp_select* sel,
                      stp_band** out)
{
    if (!ctx || !bin1_offset || nrows < 1 || npix < 0) return STP_E_ARG;
    if (bin2_type != STP_ID_I64 && bin2_type != STP_ID_I32) return set_err(ctx, STP_E_ARG, "bin2_type must be STP_ID_I64 or STP_ID_I32");
    if (bin1_offset[0] != 0 || bin1_offset[nrows] != npix) return set_err(ctx, STP_E_ARG, "bin1_offset must run from 0 to npix over nrows + 1 entries");
    for (int64_t r = 0; r < nrows; r++)
        if (bin1_offset[r] > bin1_offset[r + 1]) return set_err(ctx, STP_E_ARG, "bin1_offset must not decrease");
    return band_pack_impl(ctx, nullptr, bin1_offset, bin2, bin2_type, count_v, count_type, npix, weight, nbins_total, lo, nrows, hw, sel, out);
}

static int band_pack_impl(stp_ctx* ctx, const int64_t* bin1, const int64_t* off, const void* bin2_v, int32_t id2_type, const void* count_v,
                          int32_t count_type, int64_t npix, const double* weight, int64_t nbins_total, int64_t lo, int64_t nrows, int32_t hw,
                          stp_select* sel, stp_band** out)
{
    if (count_type != STP_COUNT_I32 && count_type != STP_COUNT_F64) return set_err(ctx, STP_E_ARG, "count_type must be STP_COUNT_I32 or STP_COUNT_F64");
    const size_t csz = count_type == STP_COUNT_F64 ? sizeof(double) : sizeof(int32_t);
    const size_t i2sz = id2_type == STP_ID_I32 ? sizeof(int32_t) : sizeof(int64_t);
    const char* count = (const char*)count_v;
    const char* bin2 = (const char*)bin2_v;
    if (!ctx || !out || npix < 0 || (npix > 0 && (!(bin1 || off) || !bin2 || !count))) return STP_E_ARG;
    if (lo < 0 || (weight && lo + nrows > nbins_total)) return set_err(ctx, STP_E_ARG, "bin range outside the weight column");
    int rc = check_hw(ctx, nrows, hw);
    if (rc) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    stp_band* b = new (std::nothrow) stp_band();
    if (!b) return STP_E_NOMEM;
    b->nrows = nrows; b->hw = hw; b->W = 2 * hw; b->owned = true;
    double* d = nullptr;
    const size_t bytes = (size_t)nrows * b->W * sizeof(double);
    if (stp_dmalloc(__LINE__, (void**)&d, bytes) != hipSuccess) { delete b; return set_err(ctx, STP_E_NOMEM, "hipMalloc(band) failed"); }
    // (measured and dropped in round 3: two pinned staging sets filled by eight host threads, 40 MB pieces, DMA of piece
    //  k beside the host copy of piece k + 1 -- 0.34 s for the 5.3 GB of the mm10-size table against 0.26 s for the
    //  runtime's own pageable path; what does pay is pinning the caller's columns IN PLACE for the call: host_pin)
    const int64_t CH = (int64_t)1 << 23;                 // pixels per staged chunk (160 MB of table columns)
    const int64_t nch = npix < CH ? npix : CH;
    // Round 5: two sets of device staging buffers.  The PCIe copies of chunk k + 1 (stream `io`) run beside the kernels of chunk k
    // (stream `pk`: index expansion, packing, the select's values); events order the two streams and say when a set is free again.
    const int nset = npix > CH ? 2 : 1;
    dev_buf b1s[2], b2s[2], bcs[2], bw, bo;
    struct evpair {
        hipEvent_t c[2] = {nullptr, nullptr}, k[2] = {nullptr, nullptr}, setup = nullptr;
        ~evpair() { for (int q = 0; q < 2; q++) { if (c[q]) (void)hipEventDestroy(c[q]); if (k[q]) (void)hipEventDestroy(k[q]); } if (setup) (void)hipEventDestroy(setup); }
    } ev;
    host_pin pin1, pin2, pinc;                            // (declared after the device buffers: released first)
    if (bin1) pin1.pin(bin1, (size_t)npix * sizeof(int64_t), ctx->io);
    pin2.pin(bin2, (size_t)npix * i2sz, ctx->io);
    pinc.pin(count, (size_t)npix * csz, ctx->io);
    stp_xfer x(ctx, ctx->io);                               // columns too small to pin (< 1 MB), the weights: staged
    auto up = [&](const host_pin& pin, void* dst, const void* src, size_t n) {
        return pin.lo != pin.hi ? pin.copy(dst, src, n, ctx->io) : x.h2d(dst, src, n);
    };
    int32_t* near = nullptr;
    if (stp_dmalloc(__LINE__, (void**)&near, (size_t)nrows * 2 * sizeof(int32_t)) != hipSuccess) {
        (void)stp_dfree(__LINE__, d); delete b;
        return set_err(ctx, STP_E_NOMEM, "hipMalloc(band nearest-pixel table) failed");
    }
    hipError_t e = hipSuccess;
    for (int q = 0; q < nset && e == hipSuccess; q++) {
        e = hipEventCreateWithFlags(&ev.c[q], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev.k[q], hipEventDisableTiming);
        if (e == hipSuccess && nch) e = b1s[q].alloc(ctx, (size_t)nch * sizeof(int64_t));
        if (e == hipSuccess && nch) e = b2s[q].alloc(ctx, (size_t)nch * i2sz);
        if (e == hipSuccess && nch) e = bcs[q].alloc(ctx, (size_t)nch * csz);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev.setup, hipEventDisableTiming);
    if (e == hipSuccess && weight) e = bw.alloc(ctx, (size_t)nrows * sizeof(double));
    if (e == hipSuccess && off) e = bo.alloc(ctx, (size_t)(nrows + 1) * sizeof(int64_t));
    if (e == hipSuccess && off) e = x.h2d(bo.p, off, (size_t)(nrows + 1) * sizeof(int64_t));     // the CSR index: 8 bytes per BIN
    if (e == hipSuccess && weight)
        e = x.h2d(bw.p, weight + lo, (size_t)nrows * sizeof(double));
    if (e == hipSuccess) e = hipEventRecord(ev.setup, ctx->io);          // the small uploads the kernels read
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->pk, ev.setup, 0);
    if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)near, 0x7FFFFFFF, (size_t)nrows * 2, ctx->pk);
    if (e == hipSuccess && weight) {
        prof_scope ps(ctx, "band_init", (double)bytes, ctx->pk);
        hipLaunchKernelGGL(k_band_init, dim3(256 * 16), dim3(256), 0, ctx->pk, (const double*)bw.p, nrows, b->W, hw, d);
        e = hipGetLastError();
    } else if (e == hipSuccess) {
        e = hipMemsetAsync(d, 0, bytes, ctx->pk);
    }
    int64_t kchunk = 0;
    for (int64_t p0 = 0; e == hipSuccess && p0 < npix; p0 += CH, kchunk++) {
        const int64_t n = npix - p0 < CH ? npix - p0 : CH;
        const int q = (int)(kchunk % nset);
        if (kchunk >= nset) e = hipEventSynchronize(ev.k[q]);             // the kernels that read this set two chunks ago are done
        if (e != hipSuccess) break;
        // (measured and dropped in round 4: bin2_id on a second upload stream, i.e. a second copy engine -- the 5.3 GB of the
        //  mm10-size table took the same 0.15-0.16 s: ~35 GB/s is what this host's memory feeds the link)
        if (bin1) e = up(pin1, b1s[q].p, bin1 + p0, (size_t)n * sizeof(int64_t));
        if (e == hipSuccess) e = up(pin2, b2s[q].p, bin2 + (size_t)p0 * i2sz, (size_t)n * i2sz);
        if (e == hipSuccess) e = up(pinc, bcs[q].p, count + (size_t)p0 * csz, (size_t)n * csz);
        if (e == hipSuccess) e = hipEventRecord(ev.c[q], ctx->io);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->pk, ev.c[q], 0);
        if (e != hipSuccess) break;
        if (!bin1) {      // bin1_id of this chunk's pixels from the CSR index, on the device
            const int64_t* hi_it = std::upper_bound(off, off + nrows + 1, p0);                    // first row starting beyond p0
            const int64_t r_lo = std::max<int64_t>(0, (hi_it - off) - 1);
            const int64_t r_hi = std::min<int64_t>(nrows - 1, (std::lower_bound(off, off + nrows + 1, p0 + n) - off) - 1);
            if (r_hi >= r_lo)
                hipLaunchKernelGGL(k_expand_bin1, dim3((unsigned)std::min<int64_t>((r_hi - r_lo + 4) / 4, 4096)), dim3(256), 0, ctx->pk,
                                   (const int64_t*)bo.p, r_lo, r_hi, lo, p0, n, (int64_t*)b1s[q].p);
            e = hipGetLastError();
            if (e != hipSuccess) break;
        }
        {
            prof_scope ps(ctx, "band_pack", (double)n * 36.0, ctx->pk);    // 20 B of table read + two 8 B cells written
            const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 256 * 64);
#define STP_PACK(CT, I2)                                                                                                                    \
            hipLaunchKernelGGL((k_band_pack<CT, I2>), dim3(grid), dim3(256), 0, ctx->pk, (const int64_t*)b1s[q].p, (const I2*)b2s[q].p,    \
                               (const CT*)bcs[q].p, n, weight ? (const double*)bw.p : nullptr, lo, nrows, b->W, hw, d, near)
            if (count_type == STP_COUNT_F64) { if (id2_type == STP_ID_I32) STP_PACK(double, int32_t); else STP_PACK(double, int64_t); }
            else { if (id2_type == STP_ID_I32) STP_PACK(int32_t, int32_t); else STP_PACK(int32_t, int64_t); }
#undef STP_PACK
        }
        e = hipGetLastError();
        if (e == hipSuccess && sel) {
            // the same staged columns also feed the quantile's order-statistic select (values as the dense symmetric
            // matrix holds them: off-diagonal pixels twice), so the table crosses PCIe once
            double* vals = nullptr;
            e = pool_alloc(ctx, (size_t)n * 2 * sizeof(double), (void**)&vals);
            if (e == hipSuccess) {
                prof_scope ps(ctx, "select_pixels", 36.0 * n, ctx->pk);
#define STP_SELV(CT, I2)                                                                                                                    \
                hipLaunchKernelGGL((k_sel_pixel_values<CT, I2>), dim3(sel_grid(n)), dim3(256), 0, ctx->pk, (const int64_t*)b1s[q].p,        \
                                   (const I2*)b2s[q].p, (const CT*)bcs[q].p, (long long)n, weight ? (const double*)bw.p : nullptr,           \
                                   (long long)nrows, vals, (long long)lo)
                if (count_type == STP_COUNT_F64) { if (id2_type == STP_ID_I32) STP_SELV(double, int32_t); else STP_SELV(double, int64_t); }
                else { if (id2_type == STP_ID_I32) STP_SELV(int32_t, int32_t); else STP_SELV(int32_t, int64_t); }
#undef STP_SELV
                e = hipGetLastError();
                sel->chunks.push_back(std::make_pair(vals, (long long)(2 * n)));
                sel->npos = -1;
            }
        }
        if (e == hipSuccess) e = hipEventRecord(ev.k[q], ctx->pk);
    }
    {   // both streams drained before anything is released or handed on (also on the error path: the staging sets go back to the pool)
        const hipError_t e1 = hipStreamSynchronize(ctx->pk), e2 = hipStreamSynchronize(ctx->io);
        if (e == hipSuccess) e = e1 != hipSuccess ? e1 : e2;
    }
    if (e != hipSuccess) {
        (void)stp_dfree(__LINE__, d); (void)stp_dfree(__LINE__, near); delete b;
        return set_err(ctx, e == hipErrorOutOfMemory ? STP_E_NOMEM : STP_E_HIP, std::string("band pack: ") + hipGetErrorString(e));
    }
    b->d = d;
    b->near = near;
    *out = b;
    return STP_OK;
}

int stp_band_nearest(stp_ctx* ctx, const stp_band* band, int32_t* right_out, int32_t* left_out)
{
    if (!ctx || !band || !right_out || !left_out) return STP_E_ARG;
    if (!band->near)
        return set_err(ctx, STP_E_UNSUPPORTED, "nearest-pixel table: only bands built by stp_band_pack carry one");
    HIPCHK(hipSetDevice(ctx->device));
    stp_xfer x(ctx, ctx->stream);
    HIPCHK(x.d2h(right_out, band->near, (size_t)band->nrows * sizeof(int32_t)));
    HIPCHK(x.d2h(left_out, band->near + band->nrows, (size_t)band->nrows * sizeof(int32_t)));
    HIPCHK(x.finish());
    return STP_OK;
}

int stp_band_download(stp_ctx* ctx, const stp_band* band, double* out_host)
{
    if (!ctx || !band || !out_host) return STP_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    stp_xfer x(ctx, ctx->stream);
    HIPCHK(x.d2h(out_host, band->d, (size_t)band->nrows * band->W * sizeof(double)));
    HIPCHK(x.finish());
    return STP_OK;
}

// Interior bleed-over constant of the given Gaussian weights and exhaustive verification of the
// multiply + 2 FMA division by it for every float mantissa (cached per weight vector).
static stp_fastdiv make_fastdiv(stp_ctx* ctx, const double* w, int R)
{
    // cached per process and per weight vector (contexts may be driven from different host threads: one lock around the
    // look-up and the 8 M-iteration verification, ~20 ms, which every new context of a process used to pay again); a few
    // vectors are kept so that alternating sigmas do not re-run it
    static std::mutex fd_mutex;
    static std::vector<std::pair<std::vector<double>, stp_fastdiv>> fd_cache;
    std::lock_guard<std::mutex> fd_lock(fd_mutex);
    (void)ctx;
    for (auto& e : fd_cache)
        if ((int)e.first.size() == 2 * R + 1 && memcmp(e.first.data(), w, (2 * R + 1) * sizeof(double)) == 0) return e.second;
    stp_fastdiv fd;
    const int S = 4 * R + 8;                                       // any size with an interior pixel
    const double V = stp_bleed_v(2 * R + 2, S, R, w);
    fd.c = stp_bleed_h(V, 2 * R + 2, S, R, w) + DBL_EPSILON;
    fd.rc = 1.0 / fd.c;
    fd.ok = 1;
    for (uint32_t m = 0; m < (1u << 23) && fd.ok; m++) {
        uint32_t bits = 0x3F800000u | m;                           // floats in [1, 2)
        float f;
        memcpy(&f, &bits, 4);
        const double q = stp_div_const((double)f, fd.c, fd.rc), t = (double)f / fd.c;
        if (memcmp(&q, &t, 8) != 0) fd.ok = 0;
    }
    if (fd_cache.size() >= 8) fd_cache.erase(fd_cache.begin());
    fd_cache.push_back(std::make_pair(std::vector<double>(w, w + 2 * R + 1), fd));
    return fd;
}

// Gaussian radii the tiled Canny kernels are instantiated for: int(4 sigma + 0.5) of sigma 1.0, 1.5, 2.0 (the reference's
// default), 2.5 and 3.0; any other radius <= 12 runs the generic k_canny
static bool canny_tiled_radius(int R) { return R == 4 || R == 6 || R == 8 || R == 10 || R == 12; }
#define STP_CANNY_RADII(X) X(4) X(6) X(8) X(10) X(12)

static int band_symmetric(stp_ctx* ctx, const stp_band* b, int* out);
// shared by stp_stripe_search and stp_dbg_stages: run the three image kernels on frames
// [f0, f0+nf) for n_levels levels; buffers sized by the caller.
static int run_chain(stp_ctx* ctx, const stp_frames* fr, const stp_search_params* prm, int f0, int nf,
                     const double* h_M /* the levels on the host */, const double* d_M, int nlev, const double* d_b, const double* d_w, float* d_gray, stp_u64* d_low,
                     stp_u64* d_high, stp_drec* d_recs, int32_t* d_cnt, int want_dbg, stp_u64* d_dbg, int16_t* d_dbgc,
                     int rcap = STP_RCAP, bool whole_gray = false /* stp_dbg_canny_f32 reads every grey tile afterwards */,
                     prof_marks* marks_out = nullptr /* the chain's timer marks, for the caller's own interval behind them */)
{
    void* p_edges = nullptr;     // parked edge maps of k_lines (one bit matrix per image)
    HIPCHK(ws_get(ctx, WS_EDGES, (size_t)nf * nlev * prm->n_bright * STP_FRAME_MAX * STP_NW * sizeof(stp_u64), &p_edges));
    const stp_band* band = fr->band;
    const int nb = prm->n_bright, a = prm->bfilter / 2, R = prm->gauss_radius;
    const int ipf = nlev * nb;
    const size_t nimg = (size_t)nf * ipf;
    bool levels_normal = true;                     // the certified grey kernel forms 1 / M and 1 / b once per level / image
    for (int i = 0; i < nlev; i++) levels_normal = levels_normal && h_M[i] > 1e-280 && h_M[i] < 1e280;
    for (int i = 0; i < nb; i++) levels_normal = levels_normal && prm->bright[i] > 1e-6 && prm->bright[i] < 1e6;
    double px = 0;
    for (int i = 0; i < nf; i++) px += (double)fr->h_S[f0 + i] * fr->h_S[f0 + i];
    const double ipx = px * ipf;   // image pixels in this launch
    // (no memset of the class bit-planes: the Canny kernels write every word k_lines reads -- one word per tile
    //  row for each tile that reaches into the image, zeros included)
    void* p_cells = nullptr;     // min / max of the grey values per 8 x 16 cell (flat-window rule, bfilter 3 only)
    if (a == 1 && canny_tiled_radius(R))
        HIPCHK(ws_get(ctx, WS_CELLS, nimg * GC_ROWS * GC_COLS * sizeof(float2), &p_cells));
    // (measured and dropped: running k_lines of one sub-chunk on a second stream beside gray / canny of the
    //  next -- chain wall 5.99 ms alone vs 6.02 / 6.29 / 6.83 ms with 2 / 4 / 8 sub-chunks)
    // Which kernels run (read per call: the tests compare them in one process).  STP_GRAY=exact selects k_gray<1> (every
    // operation of the reference; the certified kernel needs finite 1 / M), STP_CANNY=exact k_canny_pipe (every intermediate
    // in the reference's f64 arithmetic), STP_SYM=0 switches the use of the images' symmetry off.
    const char* gray_env = getenv("STP_GRAY");
    const bool gray_exact = (gray_env && strcmp(gray_env, "exact") == 0) || !levels_normal;
    const char* canny_env = getenv("STP_CANNY");
    const bool canny_exact = canny_env && strcmp(canny_env, "exact") == 0;
    const bool canny_f32 = canny_tiled_radius(R) && nb <= C32_NBMAX && !canny_exact;
    const int ctiles = ((STP_FRAME_MAX + CT_X - 1) / CT_X) * ((STP_FRAME_MAX + CT_Y - 1) / CT_Y);
    // Image symmetry (k_canny_f32): the tiles below the diagonal take their class words from the transposes of the tiles above
    // it.  Needs a band that is symmetric bit for bit (verified once per band: band_symmetric) and grey images whose every
    // pixel equals its mirror image -- k_gray_c3 reports the images where one does not (the flags behind the tile flags).
    int mirror = 0;
    bool sym_all = false;
    if (canny_f32 && a == 1 && !gray_exact) {
        const char* sym_env = getenv("STP_SYM");
        if (!(sym_env && sym_env[0] == '0')) { const int rcs = band_symmetric(ctx, band, &mirror); if (rcs) return rcs; }
        sym_all = sym_env && strcmp(sym_env, "report-all") == 0;
    }
    void* p_x = nullptr;         // k_canny_f32's flags: tile-images for the exact kernel [nimg x tiles], asymmetric images [nimg]
    const size_t nflags = nimg * ctiles;
    if (canny_f32) {
        HIPCHK(ws_get(ctx, WS_C32Q, nflags + nimg + (size_t)nf * nlev, &p_x));
        // the flags are zero between launches: k_canny_pipe_list clears every flag it has served, so the buffer is
        // cleared here only when it is new (no fill kernel -- two launch gaps -- between k_gray and the Canny kernel)
        if (ctx->c32q_zero != p_x || ctx->c32q_zero_bytes < ctx->ws_bytes[WS_C32Q]) {
            HIPCHK(hipMemsetAsync(p_x, 0, ctx->ws_bytes[WS_C32Q], ctx->stream));
            ctx->c32q_zero = p_x; ctx->c32q_zero_bytes = ctx->ws_bytes[WS_C32Q];
        }
    }
    uint8_t* p_asym = mirror ? (uint8_t*)p_x + nflags : nullptr;
    // with the symmetry in use the 20 grey tiles no computed Canny tile reads are not written (stp_gray_dead_tile); the debug entry
    // points return whole grey images and keep them
    const int skip_dead = mirror && !want_dbg && !whole_gray;
    uint8_t* p_need = skip_dead ? p_asym + nimg : nullptr;
    static_assert([] { int n = 0; for (int t = 0; t < 91; t++) n += (t / 7 >= 2 * (t % 7) + 1); return n; }() == STP_GRAY_FILL_TILES
                  && ((STP_FRAME_MAX + GT_X - 1) / GT_X) == 7 && ((STP_FRAME_MAX + GT_Y - 1) / GT_Y) == 13, "k_gray_fill's grid");
    // Frame overlap: k_canny_f32 skips the tiles inside the block a frame shares with its successor, k_lines fetches their class
    // words from the successor's planes (STP_REUSE=0 switches it off)
    const int32_t* p_shift = nullptr;
    if (canny_f32 && nf > 1) {
        const char* reuse_env = getenv("STP_REUSE");
        if (!(reuse_env && reuse_env[0] == '0')) p_shift = fr->d_shift;
    }
    {   // (test hook: STP_TEST_POISON_GRAY=1 fills the grey images and their cell table with NaN patterns before the grey kernel runs,
        //  so that a read of anything this launch did not write -- a skipped tile -- reaches the class maps instead of finding the
        //  previous launch's values there)
        const char* pz = getenv("STP_TEST_POISON_GRAY");
        if (pz && pz[0] == '1') {
            HIPCHK(hipMemsetAsync(d_gray, 0xFF, nimg * (size_t)STP_PITCH * STP_PITCH * sizeof(float), ctx->stream));
            if (p_cells) HIPCHK(hipMemsetAsync(p_cells, 0xFF, nimg * GC_ROWS * GC_COLS * sizeof(float2), ctx->stream));
        }
    }
    // timers (prof_marks): grey = [m0, m1] (stage A of SURVEY 8(d): 8 B read + 4 B written per image px), Canny = [m1, m2] (stage B: 4 B
    // read + 1 B written), line joining = [m2, m3] (stages C-F: 2 + 2 + 1 + 4 B), the chain as one interval = [m0, m3]
    prof_marks local_marks(ctx);
    prof_marks& PM = marks_out ? *marks_out : local_marks;
    const int m0 = PM.mark();
    {
        const int tiles = ((STP_FRAME_MAX + GT_X - 1) / GT_X) * ((STP_FRAME_MAX + GT_Y - 1) / GT_Y);
        if (a == 1 && !gray_exact)
            hipLaunchKernelGGL(k_gray_c3, dim3(tiles, STP_GRAY_LEVRUNS < nlev ? STP_GRAY_LEVRUNS : nlev, nf), dim3(256), 0, ctx->stream, band->d, band->W, band->hw,
                               fr->d_start, fr->d_S, fr->d_nz, f0, d_M, nlev, d_b, nb, d_gray, (float2*)p_cells, p_asym, skip_dead, p_shift, R, nf);
        else if (a == 1)
            hipLaunchKernelGGL(k_gray<1>, dim3(tiles, nlev, nf), dim3(256), 0, ctx->stream, band->d, band->W, band->hw,
                               fr->d_start, fr->d_S, fr->d_nz, f0, d_M, nlev, d_b, nb, a, d_gray, (float2*)p_cells);
        else
            hipLaunchKernelGGL(k_gray<GT_AMAX>, dim3(tiles, nlev, nf), dim3(256), 0, ctx->stream, band->d, band->W, band->hw,
                               fr->d_start, fr->d_S, fr->d_nz, f0, d_M, nlev, d_b, nb, a, d_gray, (float2*)nullptr);
        // (test hook: STP_SYM=report-all marks every image as "has a grey pixel that differs from its mirror image", so the
        //  tests can drive the path such an image takes -- its tiles below the diagonal through the exact kernel -- at will)
        if (p_asym && sym_all) HIPCHK(hipMemsetAsync(p_asym, 1, nimg, ctx->stream));
    }
    HIPCHK(hipGetLastError());
    const int m1 = PM.mark();
    PM.interval(m0, m1, "gray", ipx * 12.0);
    {
        const int tiles = ((STP_FRAME_MAX + CT_X - 1) / CT_X) * ((STP_FRAME_MAX + CT_Y - 1) / CT_Y);
        const dim3 cg(tiles, (unsigned)nimg);
        const unsigned pgrid = (unsigned)(((nf * nlev + 7) / 8) * 8 * tiles);
        if (canny_f32) {    // (without k_gray's cell maxima -- bfilter other than 3 -- g is the largest grey value k_gray can write,
                            //  and no tile is skipped as flat)
            stp_w32 W32;
            for (int k = 0; k <= CT_RMAX; k++) W32.w[k] = k <= R ? (float)prm->gauss_w[k] : 0.0f;
            c32_budget(prm->gauss_w, R, &W32);
            const stp_fastdiv fd = make_fastdiv(ctx, prm->gauss_w, R);
            const size_t smem = canny32_layout(R).total, smem_x = canny_pipe_smem_bytes(R) + CANNY_PIPE_BITS_BYTES;
            const unsigned xgrid = (unsigned)std::min<size_t>(2048, (nflags + 255) / 256);
            switch (R) {
#define STP_X(RR) case RR: \
                hipLaunchKernelGGL(k_canny_f32<RR>, dim3(pgrid), dim3(256), smem, ctx->stream, d_gray, fr->d_S, f0, nf, nlev, nb, d_w, \
                                   d_low, d_high, W32, (const float2*)p_cells, (uint8_t*)p_x, mirror, (const uint8_t*)p_asym, p_shift, p_need); \
                if (p_need) hipLaunchKernelGGL(k_gray_fill, dim3(STP_GRAY_FILL_TILES, (unsigned)(nf * nlev)), dim3(256), 0, ctx->stream, band->d, band->W, \
                                               band->hw, fr->d_start, fr->d_S, fr->d_nz, f0, d_M, nlev, d_b, nb, d_gray, (const uint8_t*)p_need, p_shift, R, nf); \
                hipLaunchKernelGGL(k_canny_pipe_list<RR>, dim3(xgrid), dim3(256), smem_x, ctx->stream, d_gray, fr->d_S, f0, nf, nlev, nb, \
                                   d_w, d_low, d_high, fd, (uint8_t*)p_x, p_asym); \
                break;
                STP_CANNY_RADII(STP_X)
#undef STP_X
            }
            // "the flag buffer is all zero" holds only if k_canny_pipe_list really ran behind k_canny_f32 (it clears what it
            // serves).  When a launch is refused the claim is withdrawn, so the next search clears the buffer again.  (A flag
            // left standing would only make the exact kernel redo a tile whose f32 result is already the exact one --
            // identical bits -- but that safety should not be what the invariant rests on.)
            const hipError_t launch_err = hipGetLastError();
            if (launch_err != hipSuccess) { ctx->c32q_zero = nullptr; ctx->c32q_zero_bytes = 0; }
            HIPCHK(launch_err);
        } else if (canny_tiled_radius(R)) {
            const stp_fastdiv fd = make_fastdiv(ctx, prm->gauss_w, R);
            switch (R) {
#define STP_X(RR) case RR: \
                hipLaunchKernelGGL(k_canny_pipe<RR>, dim3(pgrid), dim3(256), canny_pipe_smem_bytes(R) + CANNY_PIPE_BITS_BYTES, ctx->stream, \
                                   d_gray, fr->d_S, f0, nf, nlev, nb, d_w, d_low, d_high, fd, (const float2*)p_cells); \
                break;
                STP_CANNY_RADII(STP_X)
#undef STP_X
            }
        } else {
            hipLaunchKernelGGL(k_canny, cg, dim3(256), canny_smem_bytes(R) + CANNY_NMS_BYTES, ctx->stream, d_gray, fr->d_S, f0, ipf, R, d_w,
                               d_low, d_high);
        }
    }
    HIPCHK(hipGetLastError());
    const int m2 = PM.mark();
    PM.interval(m1, m2, "canny", ipx * 5.0);
    {
#if defined(STP_ABLATE_LINES_STOPS)   /* timing-only build (make ablate): truncate k_lines after phase N */
        static const int lines_stop = getenv("STP_LINES_STOP") ? atoi(getenv("STP_LINES_STOP")) : 0;
#else
        const int lines_stop = 0;          // the product library has no such switch
#endif
        if (rcap == STP_RCAP)
            hipLaunchKernelGGL(k_lines<STP_RCAP>, dim3((unsigned)nimg), dim3(512), 0, ctx->stream, d_low, d_high, band->d, band->W,
                               band->hw, fr->d_start, fr->d_S, fr->d_nz, f0, ipf, prm->minH, prm->maxW, d_recs, d_cnt,
                               (stp_u64*)p_edges, want_dbg, d_dbg, d_dbgc, lines_stop, p_shift, R, nf, mirror);
        else
            hipLaunchKernelGGL(k_lines<STP_RCAP_MAX>, dim3((unsigned)nimg), dim3(512), 0, ctx->stream, d_low, d_high, band->d, band->W,
                               band->hw, fr->d_start, fr->d_S, fr->d_nz, f0, ipf, prm->minH, prm->maxW, d_recs, d_cnt,
                               (stp_u64*)p_edges, want_dbg, d_dbg, d_dbgc, lines_stop, p_shift, R, nf, mirror);
    }
    HIPCHK(hipGetLastError());
    const int m3 = PM.mark();
    PM.interval(m2, m3, "lines", ipx * 9.0);
    PM.interval(m0, m3, "chain_wall", ipx * 26.0);
    return STP_OK;
}

// One search in flight: every chunk of the batch is enqueued on the ctx stream without a host round trip; the chunks
// append their records to ONE device array (k_compact_recs carries the running total from chunk to chunk), and the
// total travels back together with as many records as the previous search of this context produced (+25 %).
struct stp_search {
    const stp_frames* fr = nullptr;
    std::vector<double> M, bright, gw;
    stp_search_params prm;
    int nlev = 0, rcap = STP_RCAP, nchunks = 0;
    void* d_out = nullptr; size_t out_bytes = 0;
    long long* d_tot = nullptr; size_t tot_bytes = 0;
    void* pin = nullptr; size_t pin_bytes = 0, guess = 0, rec_off = 64;   // pinned: totals | parameters | records
    hipEvent_t done = nullptr;
    long long n = -1;                 // records, once known
};

static void search_release(stp_ctx* ctx, stp_search* s)
{
    if (!s) return;
    pool_release(ctx, s->d_out, s->out_bytes);
    pool_release(ctx, s->d_tot, s->tot_bytes);
    if (s->pin) {
        if (ctx->pin_free.size() < 64) ctx->pin_free.push_back(std::make_pair(s->pin_bytes, s->pin));
        else (void)stp_hfree(__LINE__, s->pin);
    }
    if (s->done) (void)hipEventDestroy(s->done);
    delete s;
}

// enqueue all chunks of the search with s->rcap record slots per image
static int search_enqueue(stp_ctx* ctx, stp_search* s)
{
    const stp_frames* fr = s->fr;
    const stp_search_params* prm = &s->prm;
    const int n_levels = s->nlev, nb = prm->n_bright, ipf = n_levels * nb, rcap = s->rcap;
    // (measured: smaller chunks, so that the grey images of a chunk stay in the 256 MB Infinity Cache between k_gray and the
    //  Canny kernel, lose more to launch tails than they gain: chr16 chain 3.11 ms at 3 072 images, 3.19 / 3.47 / 4.20 at 1 536 / 768 / 384.
    //  Round 5, on the faster kernels: 6 144 images per launch -- a whole chromosome of up to 204 frames at 5 levels x 6 images -- instead
    //  of 3 072: 72.0 -> 70.0 ms per genome step (20 instead of 36 launches of each chain kernel: fewer tails, fewer small stream
    //  operations between units); 4 608: 70.3, 9 216: 69.9.  3.9 GB of grey workspace.  STP_CHUNK_IMAGES: measurement hook.)
    static const int chunk_images = [] {            // (read once per process; anything that is not a number in 64 .. 65 536 keeps the default)
        const char* e = getenv("STP_CHUNK_IMAGES");
        if (!e || !*e) return 6144;
        char* end = nullptr;
        const long v = strtol(e, &end, 10);
        return (end && *end == 0 && v >= 64 && v <= 65536) ? (int)v : 6144;
    }();
    int chunk = chunk_images / ipf;
    if (chunk < 1) chunk = 1;
    if (chunk > fr->n) chunk = fr->n;
    const size_t cimg = (size_t)chunk * ipf;
    s->nchunks = (fr->n + chunk - 1) / chunk;
    void *pGray, *pLow, *pHigh, *pRecs, *pCnt, *pPar;
    HIPCHK(ws_get(ctx, WS_GRAY, cimg * STP_PITCH * STP_PITCH * sizeof(float) + 2 * STP_GRAY_GUARD, &pGray));
    pGray = (char*)pGray + STP_GRAY_GUARD;            // border tiles of the vertical pass read (and discard) up to R+2 rows outside
    HIPCHK(ws_get(ctx, WS_LOW, cimg * STP_FRAME_MAX * STP_NW * sizeof(stp_u64), &pLow));
    HIPCHK(ws_get(ctx, WS_HIGH, cimg * STP_FRAME_MAX * STP_NW * sizeof(stp_u64), &pHigh));
    HIPCHK(ws_get(ctx, WS_RECS, cimg * (size_t)rcap * sizeof(stp_drec), &pRecs));
    HIPCHK(ws_get(ctx, WS_CNT, cimg * sizeof(int32_t), &pCnt));
    const int nwt = 2 * prm->gauss_radius + 1;
    HIPCHK(ws_get(ctx, WS_PARAMS, (size_t)(n_levels + nb + nwt) * sizeof(double), &pPar));
    double* dM = (double*)pPar;
    double* dB = dM + n_levels;
    double* dW = dB + nb;
    // the parameters travel from the search's own pinned buffer (a copy from pageable memory could block the host
    // behind the searches already queued): [64, 64 + phdr) of s->pin, the totals come back into [0, 16)
    const size_t npar = (size_t)(n_levels + nb + nwt);
    const size_t phdr = (npar * sizeof(double) + 63) / 64 * 64;
    s->guess = std::min((size_t)fr->n * ipf * (size_t)rcap, (size_t)ctx->rec_guess);
    {
        const size_t pb = 64 + phdr + std::max(s->guess, (size_t)1024) * sizeof(stp_stripe_rec);
        if (s->pin_bytes < pb) {
            if (s->pin) (void)stp_hfree(__LINE__, s->pin);
            s->pin = nullptr; s->pin_bytes = 0;
            for (size_t i = 0; i < ctx->pin_free.size(); i++)
                if (ctx->pin_free[i].first >= pb) {
                    s->pin = ctx->pin_free[i].second; s->pin_bytes = ctx->pin_free[i].first;
                    ctx->pin_free[i] = ctx->pin_free.back(); ctx->pin_free.pop_back();
                    break;
                }
            if (!s->pin) {
                HIPCHK(hipHostMalloc(&s->pin, pb, hipHostMallocDefault));
                reg_add(s->pin, pb, __LINE__, 'h');
                if (stp_log_on()) fprintf(stderr, "[stp] hostmalloc search %p (%zu B)\n", s->pin, pb);
                s->pin_bytes = pb;
            }
        }
    }
    s->rec_off = 64 + phdr;
    {
        double* hp = (double*)((char*)s->pin + 64);
        memcpy(hp, s->M.data(), n_levels * sizeof(double));
        memcpy(hp + n_levels, prm->bright, nb * sizeof(double));
        memcpy(hp + n_levels + nb, prm->gauss_w, nwt * sizeof(double));
        HIPCHK(hipMemcpyAsync(dM, hp, npar * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    // output of the whole batch: every image may fill its slots
    const size_t ocap = (size_t)fr->n * ipf * (size_t)rcap;
    if (s->out_bytes < ocap * sizeof(stp_stripe_rec)) {
        pool_release(ctx, s->d_out, s->out_bytes);
        s->d_out = nullptr; s->out_bytes = 0;
        HIPCHK(pool_alloc(ctx, ocap * sizeof(stp_stripe_rec), &s->d_out));
        s->out_bytes = ocap * sizeof(stp_stripe_rec);
    }
    const size_t tb = (size_t)(s->nchunks + 1) * 2 * sizeof(long long);
    if (s->tot_bytes < tb) {
        pool_release(ctx, s->d_tot, s->tot_bytes);
        s->d_tot = nullptr; s->tot_bytes = 0;
        HIPCHK(pool_alloc(ctx, tb, (void**)&s->d_tot));
        s->tot_bytes = tb;
    }
    HIPCHK(hipMemsetAsync(s->d_tot, 0, 2 * sizeof(long long), ctx->stream));
    for (int c = 0; c < s->nchunks; c++) {
        const int f0 = c * chunk;
        const int nf = (fr->n - f0 < chunk) ? fr->n - f0 : chunk;
        const size_t nimg = (size_t)nf * ipf;
        prof_marks PM(ctx);
        int rc = run_chain(ctx, fr, prm, f0, nf, s->M.data(), dM, n_levels, dB, dW, (float*)pGray, (stp_u64*)pLow, (stp_u64*)pHigh,
                           (stp_drec*)pRecs, (int32_t*)pCnt, 0, nullptr, nullptr, rcap, false, &PM);
        if (rc) return rc;
        {
            hipLaunchKernelGGL(k_compact_recs, dim3((unsigned)((nimg + CR_IPB - 1) / CR_IPB)), dim3(1024), 0, ctx->stream, (const stp_drec*)pRecs,
                               (const int32_t*)pCnt, (int)nimg, f0, n_levels, nb, (stp_stripe_rec*)s->d_out, (long long)ocap,
                               (const long long*)(s->d_tot + 2 * c), s->d_tot + 2 * (c + 1), rcap,
                               rcap == STP_RCAP ? std::min(rcap, ctx->sweep_slots) : rcap);
            const int m4 = PM.mark();
            PM.interval(PM.n >= 2 ? PM.n - 2 : -1, m4, "compact_recs", (double)nimg * 4.0);      // from the chain's last mark
        }
        HIPCHK(hipGetLastError());
    }
    // the count travels with as many records as the previous search of this context produced (+25 %): one round trip
    // in the common case
    s->guess = std::min(s->guess, (s->pin_bytes - s->rec_off) / sizeof(stp_stripe_rec));
    // (measured and dropped in round 6: the two copies on the transfer stream behind an event, so that they do not stand between this
    //  search's last kernel and the next search's first -- 42.2 ms per genome step either way, profiles/r06_ab_pipeline.txt)
    HIPCHK(hipMemcpyAsync(s->pin, s->d_tot + 2 * s->nchunks, 2 * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    if (s->guess)
        HIPCHK(hipMemcpyAsync((char*)s->pin + s->rec_off, s->d_out, s->guess * sizeof(stp_stripe_rec), hipMemcpyDeviceToHost, ctx->stream));
    if (!s->done) HIPCHK(hipEventCreateWithFlags(&s->done, hipEventDisableTiming));
    HIPCHK(hipEventRecord(s->done, ctx->stream));
    s->n = -1;
    return STP_OK;
}

int stp_stripe_search_begin(stp_ctx* ctx, const stp_frames* fr, const stp_search_params* prm, const double* M_levels,
                            int32_t n_levels, stp_search** out)
{
    if (!ctx || !fr || !M_levels || !out || n_levels < 1) return STP_E_ARG;
    *out = nullptr;
    int rc = check_params(ctx, prm);
    if (rc) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    stp_search* s = new (std::nothrow) stp_search();
    if (!s) return STP_E_NOMEM;
    s->fr = fr; s->nlev = n_levels;
    s->M.assign(M_levels, M_levels + n_levels);
    s->bright.assign(prm->bright, prm->bright + prm->n_bright);
    s->gw.assign(prm->gauss_w, prm->gauss_w + 2 * prm->gauss_radius + 1);
    s->prm = *prm; s->prm.bright = s->bright.data(); s->prm.gauss_w = s->gw.data();
    rc = search_enqueue(ctx, s);
    if (rc) { search_release(ctx, s); return rc; }
    *out = s;
    return STP_OK;
}

int stp_stripe_search_count(stp_ctx* ctx, stp_search* s, int64_t* out_count)
{
    if (!ctx || !s || !out_count) return STP_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    if (s->n < 0) {
        for (;;) {
            HIPCHK(hipEventSynchronize(s->done));
            const long long* h_tot = (const long long*)s->pin;
            if (h_tot[1] && s->rcap == STP_RCAP) {
                // some image needed more than the sweep's record slots (never seen on contact maps): the batch is
                // searched again with one slot per possible column pair -- neighbouring X values pair at most once
                s->rcap = STP_RCAP_MAX;
                int rc = search_enqueue(ctx, s);
                if (rc) return rc;
                continue;
            }
            if (h_tot[1]) return set_err(ctx, STP_E_CAPACITY, "an image produced more candidate stripes than pairs of columns exist");
            s->n = h_tot[0];
            break;
        }
        if ((size_t)s->n * sizeof(stp_stripe_rec) > s->out_bytes) return set_err(ctx, STP_E_CAPACITY, "record compaction overran its buffer");
        // (a slowly decaying maximum: a search that finds more records than the speculative copy brought needs a second
        //  round trip, and units of one genome differ by more than 25 % from one to the next)
        ctx->rec_guess = std::max(s->n + s->n / 4 + 64, ctx->rec_guess - ctx->rec_guess / 16);
    }
    *out_count = s->n;
    return STP_OK;
}

int stp_stripe_search_fetch(stp_ctx* ctx, stp_search* s, stp_stripe_rec* out, int64_t cap)
{
    if (!ctx || !s || (cap > 0 && !out)) return STP_E_ARG;
    int64_t n = 0;
    int rc = stp_stripe_search_count(ctx, s, &n);
    if (rc == STP_OK && n > cap) rc = set_err(ctx, STP_E_CAPACITY, "output capacity too small for " + std::to_string(n) + " records");
    if (rc == STP_OK && n > 0) {
        const size_t have = std::min((size_t)n, s->guess);
        memcpy(out, (char*)s->pin + s->rec_off, have * sizeof(stp_stripe_rec));
        if ((size_t)n > have) {
            // the rest comes over the upload stream: this search is complete (its event), and on ctx->stream the copy would
            // wait behind every later search already queued -- the pipeline would drain for it
            stp_xfer x(ctx, ctx->io);
            hipError_t e = x.d2h(out + have, (stp_stripe_rec*)s->d_out + have, ((size_t)n - have) * sizeof(stp_stripe_rec));
            if (e == hipSuccess) e = x.finish();
            if (e != hipSuccess) rc = set_err(ctx, STP_E_HIP, std::string("record copy: ") + hipGetErrorString(e));
        }
    }
    search_release(ctx, s);
    return rc;
}

void stp_stripe_search_cancel(stp_ctx* ctx, stp_search* s)
{
    if (!ctx || !s) return;
    (void)hipSetDevice(ctx->device);
    if (s->done) (void)hipEventSynchronize(s->done);
    search_release(ctx, s);
}

int stp_stripe_search(stp_ctx* ctx, const stp_frames* fr, const stp_search_params* prm, const double* M_levels,
                      int32_t n_levels, stp_stripe_rec* out, int64_t cap, int64_t* out_count)
{
    if (!ctx || !fr || !M_levels || !out_count || n_levels < 1 || (cap > 0 && !out)) return STP_E_ARG;
    stp_search* s = nullptr;
    int rc = stp_stripe_search_begin(ctx, fr, prm, M_levels, n_levels, &s);
    if (rc) return rc;
    int64_t n = 0;
    rc = stp_stripe_search_count(ctx, s, &n);
    if (rc) { stp_stripe_search_cancel(ctx, s); return rc; }
    *out_count = n;
    if (n > cap) {
        stp_stripe_search_cancel(ctx, s);
        return set_err(ctx, STP_E_CAPACITY, "output capacity too small; out_count holds the needed size");
    }
    return stp_stripe_search_fetch(ctx, s, out, cap);
}

int stp_dbg_set_sweep_slots(stp_ctx* ctx, int32_t slots)
{
    if (!ctx || slots < 1 || slots > STP_RCAP) return STP_E_ARG;
    ctx->sweep_slots = slots;
    return STP_OK;
}

static void unpack_bits(const stp_u64* src, int S, uint8_t* dst)
{
    for (int y = 0; y < S; y++)
        for (int x = 0; x < S; x++) dst[(size_t)y * S + x] = (uint8_t)((src[y * STP_NW + (x >> 6)] >> (x & 63)) & 1ull);
}

int stp_dbg_stages(stp_ctx* ctx, const stp_frames* fr, const stp_search_params* prm, int32_t f, double M, int32_t bi,
                   float* gray, uint8_t* cls, uint8_t* edges, uint8_t* vert, int32_t* col_t, int32_t* col_end,
                   int32_t* col_ud, uint8_t* tm1, uint8_t* tm2)
{
    if (!ctx || !fr || f < 0 || f >= fr->n) return STP_E_ARG;
    int rc = check_params(ctx, prm);
    if (rc) return rc;
    if (bi < 0 || bi >= prm->n_bright) return STP_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    const int S = fr->h_S[f];
    if (S == 0) return set_err(ctx, STP_E_ARG, "frame has <= 10 non-empty columns");
    const int nb = prm->n_bright;
    dev_buf bM, bB, bW, bGray, bLow, bHigh, bRecs, bCnt, bDbg, bDbgc;
    const size_t nimg = nb;
    HIPCHK(bM.alloc(ctx, sizeof(double)));
    HIPCHK(bB.alloc(ctx, nb * sizeof(double)));
    HIPCHK(bW.alloc(ctx, (2 * prm->gauss_radius + 1) * sizeof(double)));
    HIPCHK(bGray.alloc(ctx, nimg * STP_PITCH * STP_PITCH * sizeof(float) + 2 * STP_GRAY_GUARD));
    float* dGray = (float*)((char*)bGray.p + STP_GRAY_GUARD);
    HIPCHK(bLow.alloc(ctx, nimg * STP_FRAME_MAX * STP_NW * sizeof(stp_u64)));
    HIPCHK(bHigh.alloc(ctx, nimg * STP_FRAME_MAX * STP_NW * sizeof(stp_u64)));
    HIPCHK(bRecs.alloc(ctx, nimg * STP_RCAP_MAX * sizeof(stp_drec)));
    HIPCHK(bCnt.alloc(ctx, nimg * sizeof(int32_t)));
    HIPCHK(bDbg.alloc(ctx, nimg * 4 * STP_FRAME_MAX * STP_NW * sizeof(stp_u64)));
    HIPCHK(bDbgc.alloc(ctx, nimg * 3 * STP_FRAME_MAX * sizeof(int16_t)));
    HIPCHK(hipMemsetAsync(bDbg.p, 0, nimg * 4 * STP_FRAME_MAX * STP_NW * sizeof(stp_u64), ctx->stream));
    stp_xfer x(ctx, ctx->stream);
    HIPCHK(x.h2d(bM.p, &M, sizeof(double)));
    HIPCHK(x.h2d(bB.p, prm->bright, nb * sizeof(double)));
    HIPCHK(x.h2d(bW.p, prm->gauss_w, (2 * prm->gauss_radius + 1) * sizeof(double)));
    rc = run_chain(ctx, fr, prm, f, 1, &M, (const double*)bM.p, 1, (const double*)bB.p, (const double*)bW.p, dGray,
                   (stp_u64*)bLow.p, (stp_u64*)bHigh.p, (stp_drec*)bRecs.p, (int32_t*)bCnt.p, 1, (stp_u64*)bDbg.p,
                   (int16_t*)bDbgc.p, STP_RCAP_MAX);
    if (rc) return rc;
    const size_t BW = STP_FRAME_MAX * STP_NW;
    std::vector<float> hg((size_t)STP_PITCH * STP_PITCH);
    std::vector<stp_u64> hl(BW), hh(BW), hd(4 * BW);
    std::vector<int16_t> hc(3 * STP_FRAME_MAX);
    HIPCHK(x.d2h(hg.data(), dGray + (size_t)bi * STP_PITCH * STP_PITCH, hg.size() * sizeof(float)));
    HIPCHK(x.d2h(hl.data(), (stp_u64*)bLow.p + bi * BW, BW * 8));
    HIPCHK(x.d2h(hh.data(), (stp_u64*)bHigh.p + bi * BW, BW * 8));
    HIPCHK(x.d2h(hd.data(), (stp_u64*)bDbg.p + (size_t)bi * 4 * BW, 4 * BW * 8));
    HIPCHK(x.d2h(hc.data(), (int16_t*)bDbgc.p + (size_t)bi * 3 * STP_FRAME_MAX, hc.size() * 2));
    HIPCHK(x.finish());
    if (gray)
        for (int y = 0; y < S; y++) memcpy(gray + (size_t)y * S, hg.data() + (size_t)y * STP_PITCH, S * sizeof(float));
    if (cls) {
        std::vector<uint8_t> lo((size_t)S * S), hi((size_t)S * S);
        std::vector<stp_u64> rl(BW), rh(BW);                         // the class planes are stored word-column-major (STP_CLS)
        for (int y = 0; y < STP_FRAME_MAX; y++)
            for (int w = 0; w < STP_NW; w++) { rl[y * STP_NW + w] = hl[STP_CLS(y, w)]; rh[y * STP_NW + w] = hh[STP_CLS(y, w)]; }
        unpack_bits(rl.data(), S, lo.data());
        unpack_bits(rh.data(), S, hi.data());
        for (size_t i = 0; i < (size_t)S * S; i++) cls[i] = (uint8_t)(lo[i] + hi[i]);
    }
    if (edges) unpack_bits(hd.data(), S, edges);
    if (vert) unpack_bits(hd.data() + BW, S, vert);
    if (tm1) unpack_bits(hd.data() + 2 * BW, S, tm1);
    if (tm2) unpack_bits(hd.data() + 3 * BW, S, tm2);
    for (int c = 0; c < S; c++) {
        if (col_t) col_t[c] = hc[c];
        if (col_end) col_end[c] = hc[STP_FRAME_MAX + c];
        if (col_ud) col_ud[c] = hc[2 * STP_FRAME_MAX + c];
    }
    return STP_OK;
}


// ---------------------------------------------------------------------------------------------
// score path
struct stp_background {
    double* d = nullptr;        // lu | ru | ld | rd, each 400 x ncol
    double* sorted = nullptr;   // the same rows sorted ascending (NaN last)
    double* sorted_t = nullptr; // ... and with the row index fastest (k_score_wave: lanes = consecutive rows)
    int* nvalid = nullptr;      // non-NaN count of each of the 1600 rows
    int ncol = 0;
};

// Device-side evidence for k_canny_f32's error budget (tests): the chain runs as in stp_dbg_stages, then the DBG instance of
// the kernel is launched on the same grey images and cell table and dumps what the tiles of image bi computed in f32.
int stp_dbg_canny_f32(stp_ctx* ctx, const stp_frames* fr, const stp_search_params* prm, int32_t f, double M, int32_t bi,
                      float* planes, int64_t* counts)
{
    if (!ctx || !fr || f < 0 || f >= fr->n || !planes || !counts) return STP_E_ARG;
    int rc = check_params(ctx, prm);
    if (rc) return rc;
    const int nb = prm->n_bright, R = prm->gauss_radius;
    if (bi < 0 || bi >= nb) return STP_E_ARG;
    if (!canny_tiled_radius(R) || nb > C32_NBMAX || prm->bfilter != 3)
        return set_err(ctx, STP_E_UNSUPPORTED, "k_canny_f32 runs for the tiled radii, at most 8 brightness levels (cells: bfilter 3)");
    HIPCHK(hipSetDevice(ctx->device));
    const int S = fr->h_S[f];
    if (S == 0) return set_err(ctx, STP_E_ARG, "frame has <= 10 non-empty columns");
    dev_buf bM, bB, bW, bGray, bLow, bHigh, bRecs, bCnt, bPl, bC, bX;
    const size_t nimg = nb, PL = (size_t)STP_PITCH * STP_PITCH;
    HIPCHK(bM.alloc(ctx, sizeof(double)));
    HIPCHK(bB.alloc(ctx, nb * sizeof(double)));
    HIPCHK(bW.alloc(ctx, (2 * R + 1) * sizeof(double)));
    HIPCHK(bGray.alloc(ctx, nimg * PL * sizeof(float) + 2 * STP_GRAY_GUARD));
    float* dGray = (float*)((char*)bGray.p + STP_GRAY_GUARD);
    HIPCHK(bLow.alloc(ctx, nimg * STP_FRAME_MAX * STP_NW * sizeof(stp_u64)));
    HIPCHK(bHigh.alloc(ctx, nimg * STP_FRAME_MAX * STP_NW * sizeof(stp_u64)));
    HIPCHK(bRecs.alloc(ctx, nimg * STP_RCAP_MAX * sizeof(stp_drec)));
    HIPCHK(bCnt.alloc(ctx, nimg * sizeof(int32_t)));
    HIPCHK(bPl.alloc(ctx, 6 * PL * sizeof(float)));
    HIPCHK(bC.alloc(ctx, 4 * sizeof(unsigned long long)));
    const int tiles = ((STP_FRAME_MAX + CT_X - 1) / CT_X) * ((STP_FRAME_MAX + CT_Y - 1) / CT_Y);
    HIPCHK(bX.alloc(ctx, nimg * tiles));
    stp_xfer x(ctx, ctx->stream);
    HIPCHK(x.h2d(bM.p, &M, sizeof(double)));
    HIPCHK(x.h2d(bB.p, prm->bright, nb * sizeof(double)));
    HIPCHK(x.h2d(bW.p, prm->gauss_w, (2 * R + 1) * sizeof(double)));
    rc = run_chain(ctx, fr, prm, f, 1, &M, (const double*)bM.p, 1, (const double*)bB.p, (const double*)bW.p, dGray,
                   (stp_u64*)bLow.p, (stp_u64*)bHigh.p, (stp_drec*)bRecs.p, (int32_t*)bCnt.p, 0, nullptr, nullptr, STP_RCAP_MAX, true);
    if (rc) return rc;
    void* p_cells = nullptr;                         // the table run_chain has just filled (same workspace slot, same size)
    HIPCHK(ws_get(ctx, WS_CELLS, nimg * GC_ROWS * GC_COLS * sizeof(float2), &p_cells));
    HIPCHK(hipMemsetAsync(bPl.p, 0xFF, 6 * PL * sizeof(float), ctx->stream));        // NaN: pixels of tiles skipped as flat
    HIPCHK(hipMemsetAsync(bC.p, 0, 4 * sizeof(unsigned long long), ctx->stream));
    HIPCHK(hipMemsetAsync(bX.p, 0, nimg * tiles, ctx->stream));
    stp_w32 W32;
    for (int k = 0; k <= CT_RMAX; k++) W32.w[k] = k <= R ? (float)prm->gauss_w[k] : 0.0f;
    c32_budget(prm->gauss_w, R, &W32);
    const size_t smem = canny32_layout(R).total;
    const unsigned pgrid = (unsigned)(8 * tiles);
    switch (R) {
#define STP_X(RR) case RR: \
        hipLaunchKernelGGL((k_canny_f32<RR, true>), dim3(pgrid), dim3(256), smem, ctx->stream, (const float*)dGray, fr->d_S, f, 1, 1, nb, \
                           (const double*)bW.p, (stp_u64*)bLow.p, (stp_u64*)bHigh.p, W32, (const float2*)p_cells, (uint8_t*)bX.p, \
                           0, (const uint8_t*)nullptr, (const int32_t*)nullptr, (uint8_t*)nullptr /* every tile computed: the dump covers the whole image */, \
                           (float*)bPl.p, (int)bi, (unsigned long long*)bC.p); \
        break;
        STP_CANNY_RADII(STP_X)
#undef STP_X
    }
    HIPCHK(hipGetLastError());
    std::vector<float> hp(6 * PL);
    unsigned long long hc[4];
    HIPCHK(x.d2h(hp.data(), bPl.p, hp.size() * sizeof(float)));
    HIPCHK(x.d2h(hc, bC.p, sizeof(hc)));
    HIPCHK(x.finish());
    for (int q = 0; q < 6; q++)
        for (int y = 0; y < S; y++) memcpy(planes + ((size_t)q * S + y) * S, hp.data() + q * PL + (size_t)y * STP_PITCH, S * sizeof(float));
    counts[0] = (int64_t)hc[0]; counts[1] = (int64_t)hc[1]; counts[2] = (int64_t)hc[2];
    return STP_OK;
}

static stp_bandref bref(const stp_band* b) { return stp_bandref{b->d, b->nrows, b->W, b->hw, 0}; }
// the band's symmetry verdict, established on first use (one pass over the band on the auxiliary stream, one word back)
static int band_symmetric(stp_ctx* ctx, const stp_band* b, int* out)
{
    if (b->sym < 0) {
        dev_buf f;
        HIPCHK(f.alloc(ctx, sizeof(int)));
        HIPCHK(hipMemsetAsync(f.p, 0, sizeof(int), ctx->aux));
        if (b->hw > 1)
            hipLaunchKernelGGL(k_band_symcheck, dim3(256 * 16), dim3(256), 0, ctx->aux, b->d, b->nrows, b->W, b->hw, (int*)f.p);
        HIPCHK(hipGetLastError());
        int asym = 1;
        HIPCHK(hipMemcpyAsync(&asym, f.p, sizeof(int), hipMemcpyDeviceToHost, ctx->aux));
        HIPCHK(hipStreamSynchronize(ctx->aux));
        b->sym = asym ? 0 : 1;
    }
    *out = b->sym;
    return STP_OK;
}

int stp_diag_sums(stp_ctx* ctx, const stp_band* band, double* part_sum, int64_t* part_cnt, int32_t n400)
{
    if (!ctx || !band || !part_sum || !part_cnt) return STP_E_ARG;
    if (n400 != (int32_t)((band->nrows + 399) / 400)) return set_err(ctx, STP_E_ARG, "n400 must be ceil(nrows/400)");
    HIPCHK(hipSetDevice(ctx->device));
    dev_buf bs, bc;
    const size_t n = (size_t)n400 * STP_NDIAG;
    HIPCHK(bs.alloc(ctx, n * sizeof(double)));
    HIPCHK(bc.alloc(ctx, n * sizeof(long long)));
    {
        prof_scope ps(ctx, "diag_sums", 8.0 * 400.0 * (double)band->nrows, ctx->aux);
        hipLaunchKernelGGL(k_diag_sums, dim3(n400), dim3(448), 0, ctx->aux, bref(band), (double*)bs.p, (long long*)bc.p);
    }
    HIPCHK(hipGetLastError());
    stp_xfer x(ctx, ctx->aux);
    HIPCHK(x.d2h(part_sum, bs.p, n * sizeof(double)));
    HIPCHK(x.d2h(part_cnt, bc.p, n * sizeof(long long)));
    HIPCHK(x.finish());
    return STP_OK;
}

int stp_null_windows(stp_ctx* ctx, const stp_band* band, const double* unit_matrix, const stp_null_sample* samples,
                     int32_t n, int32_t bs, double* lu, double* ru, double* ld, double* rd)
{
    if (!ctx || !band || !samples || n <= 0 || !lu || !ru || !ld || !rd) return STP_E_ARG;
    if (unit_matrix)
        for (int i = 1; i < n; i++)
            if (samples[i].row0 != samples[0].row0 || samples[i].nrow != samples[0].nrow ||
                samples[i].col0 != samples[0].col0 || samples[i].ncol != samples[0].ncol)
                return set_err(ctx, STP_E_ARG, "with a unit matrix all samples of a call must share its geometry");
    if (bs < 1 || bs > 1000) return set_err(ctx, STP_E_ARG, "background window size must be in 1..1000 bins");
    for (int i = 0; i < n; i++) {
        const stp_null_sample& s = samples[i];
        if (s.nrow <= 0 || s.ncol <= 0 || s.row0 < 0 || s.col0 < 0 || s.row0 + s.nrow > band->nrows ||
            s.col0 + s.ncol > band->nrows)
            return set_err(ctx, STP_E_ARG, "null sample " + std::to_string(i) + ": unit matrix outside the chromosome");
    }
    HIPCHK(hipSetDevice(ctx->device));
    dev_buf bS, bO, bD;
    stp_xfer x(ctx, ctx->aux);
    const size_t tn = (size_t)STP_NDIAG * n;
    if (unit_matrix) {
        const size_t mb = (size_t)samples[0].nrow * samples[0].ncol * sizeof(double);
        HIPCHK(bD.alloc(ctx, mb));
        HIPCHK(x.h2d(bD.p, unit_matrix, mb));
    }
    HIPCHK(bS.alloc(ctx, (size_t)n * sizeof(stp_null_sample)));
    HIPCHK(bO.alloc(ctx, 4 * tn * sizeof(double)));
    HIPCHK(x.h2d(bS.p, samples, (size_t)n * sizeof(stp_null_sample)));
    double* o = (double*)bO.p;
    {
        prof_scope ps(ctx, "null_windows", 8.0 * (3.0 * bs) * (800.0 + bs) * n, ctx->aux);
        if (bs * bs <= 128)
            hipLaunchKernelGGL(k_null_windows<false>, dim3(n), dim3(256), 0, ctx->aux, bref(band),
                               (const double*)(unit_matrix ? bD.p : nullptr), (const stp_null_sample*)bS.p, n, bs, o, o + tn,
                               o + 2 * tn, o + 3 * tn);
        else
            hipLaunchKernelGGL(k_null_windows<true>, dim3(n), dim3(256), 0, ctx->aux, bref(band),
                               (const double*)(unit_matrix ? bD.p : nullptr), (const stp_null_sample*)bS.p, n, bs, o, o + tn,
                               o + 2 * tn, o + 3 * tn);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(x.d2h(lu, o, tn * sizeof(double)));
    HIPCHK(x.d2h(ru, o + tn, tn * sizeof(double)));
    HIPCHK(x.d2h(ld, o + 2 * tn, tn * sizeof(double)));
    HIPCHK(x.d2h(rd, o + 3 * tn, tn * sizeof(double)));
    HIPCHK(x.finish());
    return STP_OK;
}

int stp_background_upload(stp_ctx* ctx, const double* lu, const double* ru, const double* ld, const double* rd,
                          int32_t ncol, stp_background** out)
{
    if (!ctx || !lu || !ru || !ld || !rd || ncol <= 0 || !out) return STP_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    stp_background* bg = new (std::nothrow) stp_background();
    if (!bg) return STP_E_NOMEM;
    const size_t tn = (size_t)STP_NDIAG * ncol;
    if (stp_dmalloc(__LINE__, (void**)&bg->d, 4 * tn * sizeof(double)) != hipSuccess) { delete bg; return set_err(ctx, STP_E_NOMEM, "hipMalloc(background)"); }
    bg->ncol = ncol;
    const double* src[4] = {lu, ru, ld, rd};
    stp_xfer x(ctx, ctx->aux);
    for (int t = 0; t < 4; t++) {
        hipError_t e = x.h2d(bg->d + t * tn, src[t], tn * sizeof(double));
        if (e != hipSuccess) { (void)stp_dfree(__LINE__, bg->d); delete bg; return set_err(ctx, STP_E_HIP, "background upload failed"); }
    }
    if (ncol > STP_BG_MAXCOL) { (void)stp_dfree(__LINE__, bg->d); delete bg; return set_err(ctx, STP_E_UNSUPPORTED, "background tables wider than 2048 columns"); }
    if (stp_dmalloc(__LINE__, (void**)&bg->sorted, 4 * tn * sizeof(double)) != hipSuccess ||
        stp_dmalloc(__LINE__, (void**)&bg->sorted_t, 4 * tn * sizeof(double)) != hipSuccess ||
        stp_dmalloc(__LINE__, (void**)&bg->nvalid, 4 * STP_NDIAG * sizeof(int)) != hipSuccess) {
        stp_background_free(ctx, bg);
        return set_err(ctx, STP_E_NOMEM, "hipMalloc(sorted background)");
    }
    {
        prof_scope ps(ctx, "bg_sort", 16.0 * 4 * tn, ctx->aux);
        hipLaunchKernelGGL(k_bg_sort, dim3(4 * STP_NDIAG), dim3(512), 0, ctx->aux, (const double*)bg->d, ncol, bg->sorted,
                           bg->nvalid, bg->sorted_t);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(x.finish());
    *out = bg;
    return STP_OK;
}

void stp_background_free(stp_ctx* ctx, stp_background* bg)
{
    if (!bg) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (bg->d) (void)stp_dfree(__LINE__, bg->d);
    if (bg->sorted) (void)stp_dfree(__LINE__, bg->sorted);
    if (bg->sorted_t) (void)stp_dfree(__LINE__, bg->sorted_t);
    if (bg->nvalid) (void)stp_dfree(__LINE__, bg->nvalid);
    delete bg;
}

static int check_rect(stp_ctx* ctx, const stp_band* band, int64_t i, int r0, int r1, int c0, int c1, int maxrows, int maxcols)
{
    if (r0 < 0 || c0 < 0 || r1 > band->nrows || c1 > band->nrows || r1 <= r0 || c1 < c0)
        return set_err(ctx, STP_E_ARG, "stripe " + std::to_string(i) + ": region outside the chromosome");
    if (r1 - r0 > maxrows || c1 - c0 > maxcols)
        return set_err(ctx, STP_E_UNSUPPORTED, "stripe " + std::to_string(i) + ": longer than " + std::to_string(STP_SCORE_MAXROWS) +
                       " bins or wider than " + std::to_string(STP_SCORE_MAXCOLS) + " bins");
    // every pixel must lie inside the stored band
    if ((int64_t)c1 - 1 - r0 >= band->hw || (int64_t)r1 - 1 - c0 > band->hw)
        return set_err(ctx, STP_E_ARG, "stripe " + std::to_string(i) + ": region leaves the +-halfwidth band; upload a wider band");
    return STP_OK;
}

// p-value and / or Stripiness of n stripes: ONE upload, one launch of k_score_wave over the stripes its wave form takes
// (plus the block kernels over the rest, through index lists), one download.  pst / sst: either may be null.
static int run_score(stp_ctx* ctx, const stp_band* band, const stp_background* bg, int32_t bs, const double* exval400,
                     const stp_pv_stripe* pst, const stp_score_stripe* sst, int64_t n, double* out_p, double* out_g, double* out_mean,
                     double* out_total, int32_t* out_status)
{
    if (n == 0) return STP_OK;
    if (n > 0x7FFFFFF0ll) return set_err(ctx, STP_E_UNSUPPORTED, "more than 2^31 stripes in one call");
    for (int64_t i = 0; i < n; i++) {
        if (pst) {
            int rc = check_rect(ctx, band, i, pst[i].row0, pst[i].row1, pst[i].col0, pst[i].col1, STP_SCORE_MAXROWS, 1 << 20);
            if (rc) return rc;
            if (pst[i].mode < 0 || pst[i].mode > 2 || (pst[i].mode == 2 && (pst[i].fixed_row < 0 || pst[i].fixed_row >= STP_NDIAG)))
                return set_err(ctx, STP_E_ARG, "stripe " + std::to_string(i) + ": bad direction mode");
        }
        if (sst)
            for (int b = 0; b < 3; b++) {
                int rc = check_rect(ctx, band, i, sst[i].row0, sst[i].row1, sst[i].col0[b], sst[i].col1[b], STP_SCORE_MAXROWS,
                                    STP_SCORE_MAXCOLS);
                if (rc) return rc;
            }
    }
    HIPCHK(hipSetDevice(ctx->device));
    // which stripes the wave form takes (STP_SCORE=block: none -- the tests compare the two forms)
    const char* env = getenv("STP_SCORE");
    const bool all_block = env && strcmp(env, "block") == 0;
    std::vector<int> small, tall, big;                                    // wave form (<= 64 SW_KR_SHORT = 192 rows), wave form (<= 256 rows), block kernels
    double bytes_pv = 0, bytes_sc = 0;
    for (int64_t i = 0; i < n; i++) {
        bool ok = !all_block;
        int h = 0;
        if (pst) {
            const int w = pst[i].col1 - pst[i].col0;
            h = pst[i].row1 - pst[i].row0;
            ok = ok && w <= 128 && bs <= 128;                            // every row sum is one pairwise leaf
            bytes_pv += (8.0 * w + 16000.0) * h;
        }
        if (sst) {
            h = std::max(h, (int)(sst[i].row1 - sst[i].row0));
            for (int b = 0; b < 3; b++) {
                ok = ok && (sst[i].col1[b] - sst[i].col0[b]) <= SW_MAXW;
                bytes_sc += 8.0 * (sst[i].col1[b] - sst[i].col0[b]) * (sst[i].row1 - sst[i].row0);
            }
        }
        (!ok || h > SW_MAXH ? big : (h > 64 * SW_KR_SHORT ? tall : small)).push_back((int)i);
    }
    dev_buf bP, bS, bE, bO, bI;
    if (pst) HIPCHK(bP.alloc(ctx, (size_t)n * sizeof(stp_pv_stripe)));
    if (sst) HIPCHK(bS.alloc(ctx, (size_t)n * sizeof(stp_score_stripe)));
    if (sst) HIPCHK(bE.alloc(ctx, STP_NDIAG * sizeof(double)));
    HIPCHK(bO.alloc(ctx, (size_t)n * 4 * sizeof(double) + (size_t)n * sizeof(int)));
    const bool need_idx = small.size() != (size_t)n;                     // (everything in one wave launch: no list, stripes in order)
    if (need_idx) HIPCHK(bI.alloc(ctx, (size_t)n * sizeof(int)));
    std::vector<int> all;                                                // the index lists' host copy: declared before the transfer object, so it outlives every copy
    stp_xfer x(ctx, ctx->aux);
    if (pst) HIPCHK(x.h2d(bP.p, pst, (size_t)n * sizeof(stp_pv_stripe)));
    if (sst) HIPCHK(x.h2d(bS.p, sst, (size_t)n * sizeof(stp_score_stripe)));
    if (sst) HIPCHK(x.h2d(bE.p, exval400, STP_NDIAG * sizeof(double)));
    int *d_small = nullptr, *d_tall = nullptr, *d_big = nullptr;
    if (need_idx) {
        all = small;
        all.insert(all.end(), tall.begin(), tall.end());
        all.insert(all.end(), big.begin(), big.end());
        d_small = (int*)bI.p; d_tall = d_small + small.size(); d_big = d_tall + tall.size();
        HIPCHK(x.h2d(d_small, all.data(), all.size() * sizeof(int)));
    }
    double* o = (double*)bO.p;                                           // p | g | mean | total | status
    double* o_p = o; double* o_g = o + n; double* o_m = o + 2 * n; double* o_t = o + 3 * n;
    int* o_s = (int*)(o + 4 * n);
    stp_bandref br = bref(band);
    {
        const char* e2 = getenv("STP_SCORE_NOSYM");                      // (measurement hook: row-strided reads even in a symmetric band)
        if (!(e2 && e2[0] == '1')) { int rcs = band_symmetric(ctx, band, &br.sym); if (rcs) return rcs; }
    }
    const double* d_srt = bg ? (const double*)bg->sorted : nullptr;
    const double* d_srt_t = bg ? (const double*)bg->sorted_t : nullptr;      // the wave kernel's layout
    const int* d_nv = bg ? (const int*)bg->nvalid : nullptr;
    const int ncolbg = bg ? bg->ncol : 0;
    const char* scope = (pst && sst) ? "score" : (pst ? "pvalue" : "stripiness");
    for (int cls = 0; cls < 2; cls++) {
        const std::vector<int>& L = cls ? tall : small;
        if (L.empty()) continue;
        const int* d_idx = need_idx ? (cls ? d_tall : d_small) : nullptr;
        const unsigned grid = (unsigned)((L.size() + SW_WAVES - 1) / SW_WAVES);
        const double frac = (double)L.size() / (double)n;
        prof_scope ps(ctx, scope, ((pst ? bytes_pv : 0.0) + (sst ? bytes_sc : 0.0)) * frac, ctx->aux);
#define STP_SW_LAUNCH(PV, SC, KR)                                                                                                      \
        hipLaunchKernelGGL((k_score_wave<PV, SC, KR>), dim3(grid), dim3(64 * SW_WAVES), 0, ctx->aux, br, d_idx, (int)L.size(), bs, d_srt_t, d_nv, ncolbg, \
                           (const stp_pv_stripe*)bP.p, o_p, (const double*)bE.p, (const stp_score_stripe*)bS.p, o_g, o_m, o_t, o_s)
        if (pst && sst) { if (cls) STP_SW_LAUNCH(true, true, SW_KR_TALL); else STP_SW_LAUNCH(true, true, SW_KR_SHORT); }
        else if (pst) { if (cls) STP_SW_LAUNCH(true, false, SW_KR_TALL); else STP_SW_LAUNCH(true, false, SW_KR_SHORT); }
        else { if (cls) STP_SW_LAUNCH(false, true, SW_KR_TALL); else STP_SW_LAUNCH(false, true, SW_KR_SHORT); }
#undef STP_SW_LAUNCH
        HIPCHK(hipGetLastError());
    }
    if (!big.empty()) {
        const double frac = (double)big.size() / (double)n;
        if (pst) {
            prof_scope ps(ctx, (small.empty() && tall.empty()) ? "pvalue" : "pvalue_block", bytes_pv * frac, ctx->aux);
            bool bigsum = bs > 128;
            int hmax = 1;
            for (int i : big) {
                bigsum = bigsum || (pst[i].col1 - pst[i].col0) > 128;
                hmax = std::max(hmax, (int)(pst[i].row1 - pst[i].row0));
            }
            const size_t lds = 4 * sizeof(double) * (size_t)hmax;
            if (lds > 48 * 1024) {      // stripes beyond ~1500 rows: more dynamic LDS than the default launch limit
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pvalue<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pvalue<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            }
            if (!bigsum)
                hipLaunchKernelGGL(k_pvalue<false>, dim3((unsigned)big.size()), dim3(STP_PV_NT), lds, ctx->aux, bref(band), d_srt, d_nv, ncolbg, bs,
                                   (const stp_pv_stripe*)bP.p, o_p, hmax, (const int*)d_big);
            else
                hipLaunchKernelGGL(k_pvalue<true>, dim3((unsigned)big.size()), dim3(STP_PV_NT), lds, ctx->aux, bref(band), d_srt, d_nv, ncolbg, bs,
                                   (const stp_pv_stripe*)bP.p, o_p, hmax, (const int*)d_big);
            HIPCHK(hipGetLastError());
        }
        if (sst) {
            prof_scope ps(ctx, (small.empty() && tall.empty()) ? "stripiness" : "stripiness_block", bytes_sc * frac, ctx->aux);
            bool bigsum = false;
            int hmax = 1, wmax = 1;
            for (int i : big) {
                hmax = std::max(hmax, (int)(sst[i].row1 - sst[i].row0));
                for (int b = 0; b < 3; b++) {
                    wmax = std::max(wmax, (int)(sst[i].col1[b] - sst[i].col0[b]));
                    bigsum = bigsum || (sst[i].col1[b] - sst[i].col0[b]) > 128;
                }
            }
            const int HR = (hmax + 3) & ~3, CW = (wmax + 3) & ~3;     // keeps every sub-array 8-byte aligned
            const size_t lds = sizeof(double) * (STP_NDIAG + 3 * (size_t)HR + std::max(HR, 256)) +
                               sizeof(int16_t) * ((size_t)HR + 3 * (size_t)CW) + (size_t)HR;
            if (lds > 48 * 1024) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stripiness<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stripiness<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            }
            if (!bigsum)
                hipLaunchKernelGGL(k_stripiness<false>, dim3((unsigned)big.size()), dim3(STP_SC_NT), lds, ctx->aux, bref(band), (const double*)bE.p,
                                   (const stp_score_stripe*)bS.p, o_g, o_m, o_t, o_s, HR, CW, (const int*)d_big);
            else
                hipLaunchKernelGGL(k_stripiness<true>, dim3((unsigned)big.size()), dim3(STP_SC_NT), lds, ctx->aux, bref(band), (const double*)bE.p,
                                   (const stp_score_stripe*)bS.p, o_g, o_m, o_t, o_s, HR, CW, (const int*)d_big);
            HIPCHK(hipGetLastError());
        }
    }
    if (pst) HIPCHK(x.d2h(out_p, o_p, (size_t)n * sizeof(double)));
    if (sst) {
        HIPCHK(x.d2h(out_g, o_g, (size_t)n * sizeof(double)));
        HIPCHK(x.d2h(out_mean, o_m, (size_t)n * sizeof(double)));
        HIPCHK(x.d2h(out_total, o_t, (size_t)n * sizeof(double)));
        if (out_status) HIPCHK(x.d2h(out_status, o_s, (size_t)n * sizeof(int)));
    }
    HIPCHK(x.finish());
    return STP_OK;
}

int stp_pvalue(stp_ctx* ctx, const stp_band* band, const stp_background* bg, int32_t bs, const stp_pv_stripe* st, int64_t n,
               double* out_p)
{
    if (!ctx || !band || !bg || !st || !out_p || n < 0 || bs < 1) return STP_E_ARG;
    return run_score(ctx, band, bg, bs, nullptr, st, nullptr, n, out_p, nullptr, nullptr, nullptr, nullptr);
}

int stp_stripiness(stp_ctx* ctx, const stp_band* band, const double* exval400, const stp_score_stripe* st, int64_t n,
                   double* out_g, double* out_mean, double* out_total, int32_t* out_status)
{
    if (!ctx || !band || !exval400 || !st || !out_g || !out_mean || !out_total || n < 0) return STP_E_ARG;
    return run_score(ctx, band, nullptr, 1, exval400, nullptr, st, n, nullptr, out_g, out_mean, out_total, out_status);
}

int stp_score(stp_ctx* ctx, const stp_band* band, const stp_background* bg, int32_t bs, const double* exval400,
              const stp_pv_stripe* pv_stripes, const stp_score_stripe* sc_stripes, int64_t n, double* out_p, double* out_g,
              double* out_mean, double* out_total, int32_t* out_status)
{
    if (!ctx || !band || !bg || !exval400 || !pv_stripes || !sc_stripes || !out_p || !out_g || !out_mean || !out_total || n < 0 || bs < 1)
        return STP_E_ARG;
    // Identical stripes are scored once (round 6).  A driver that scores candidates straight from the search hands over the same
    // rectangle many times -- a stripe is found in several brightness images, at several maxpixel levels and in both frames that
    // hold it: 59 % of the benchmark genome's 375 k candidates per step are byte-for-byte repeats of another row -- and a row's
    // results are a function of its two descriptors alone (inherited background rows arrive resolved, as `mode` / `upbase`).  Rows
    // are compared as bytes (a hash table over the two structs, verified by memcmp), the unique ones scored, the results copied
    // to their repeats: 0.2 ms of host time per 18 k rows for 0.14 ms less kernel time and smaller copies, on a device where
    // the score kernels' time comes 1 : 1 out of the step (they run beside the chain, which fills the device alone).
    if (n >= 256) {
        const size_t ps = sizeof(stp_pv_stripe), ss = sizeof(stp_score_stripe);
        auto row_hash = [&](int64_t i) {
            unsigned long long h = 0x243F6A8885A308D3ull;
            auto mix = [&](const unsigned char* p, size_t len) {
                size_t k = 0;
                for (; k + 8 <= len; k += 8) { unsigned long long w; memcpy(&w, p + k, 8); h = (h ^ w) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; }
                if (k < len) { unsigned long long w = 0; memcpy(&w, p + k, len - k); h = (h ^ w) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; }
            };
            mix((const unsigned char*)(pv_stripes + i), ps);
            mix((const unsigned char*)(sc_stripes + i), ss);
            return h;
        };
        size_t cap = 1;
        while (cap < (size_t)n * 2) cap <<= 1;
        std::vector<int32_t> slot(cap, -1), rep((size_t)n), uniq;
        uniq.reserve((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            size_t k = (size_t)row_hash(i) & (cap - 1);
            for (;;) {
                const int32_t j = slot[k];
                if (j < 0) { slot[k] = (int32_t)uniq.size(); rep[i] = (int32_t)uniq.size(); uniq.push_back((int32_t)i); break; }
                const int64_t r = uniq[j];
                if (memcmp(pv_stripes + r, pv_stripes + i, ps) == 0 && memcmp(sc_stripes + r, sc_stripes + i, ss) == 0) { rep[i] = j; break; }
                k = (k + 1) & (cap - 1);
            }
        }
        const int64_t m = (int64_t)uniq.size();
        if (m * 10 <= n * 9) {                       // (fewer than a tenth repeated: not worth the copies)
            std::vector<stp_pv_stripe> pvu((size_t)m);
            std::vector<stp_score_stripe> scu((size_t)m);
            for (int64_t j = 0; j < m; j++) { pvu[j] = pv_stripes[uniq[j]]; scu[j] = sc_stripes[uniq[j]]; }
            std::vector<double> o((size_t)m * 4);
            std::vector<int32_t> os((size_t)m, 0);
            const int rc = run_score(ctx, band, bg, bs, exval400, pvu.data(), scu.data(), m, o.data(), o.data() + m, o.data() + 2 * m,
                                     o.data() + 3 * m, os.data());
            if (rc == STP_OK) {
                for (int64_t i = 0; i < n; i++) {
                    const int32_t j = rep[i];
                    out_p[i] = o[j]; out_g[i] = o[m + j]; out_mean[i] = o[2 * m + j]; out_total[i] = o[3 * m + j];
                    if (out_status) out_status[i] = os[j];
                }
                return STP_OK;
            }
            // (an error names a stripe by its number: report it against the caller's own list)
        }
    }
    return run_score(ctx, band, bg, bs, exval400, pv_stripes, sc_stripes, n, out_p, out_g, out_mean, out_total, out_status);
}

int stp_stripe_mean(stp_ctx* ctx, const stp_band* band, const stp_rect* rc, int64_t n, double* out_mean, double* out_sum)
{
    if (!ctx || !band || !rc || !out_mean || !out_sum || n < 0) return STP_E_ARG;
    if (n == 0) return STP_OK;
    for (int64_t i = 0; i < n; i++) {
        int r = check_rect(ctx, band, i, rc[i].row0, rc[i].row1, rc[i].col0, rc[i].col1, 1 << 20, 1 << 20);
        if (r) return r;
    }
    HIPCHK(hipSetDevice(ctx->device));
    dev_buf bS, bO;
    HIPCHK(bS.alloc(ctx, (size_t)n * sizeof(stp_rect)));
    HIPCHK(bO.alloc(ctx, (size_t)n * 2 * sizeof(double)));
    stp_xfer x(ctx, ctx->aux);
    HIPCHK(x.h2d(bS.p, rc, (size_t)n * sizeof(stp_rect)));
    double* o = (double*)bO.p;
    {
        double bytes = 0;
        for (int64_t i = 0; i < n; i++) bytes += 8.0 * (rc[i].col1 - rc[i].col0) * (rc[i].row1 - rc[i].row0);
        prof_scope ps(ctx, "stripe_mean", bytes, ctx->aux);
        hipLaunchKernelGGL(k_stripe_mean, dim3((unsigned)n), dim3(256), 0, ctx->aux, bref(band), (const stp_rect*)bS.p, o, o + n);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(x.d2h(out_mean, o, (size_t)n * sizeof(double)));
    HIPCHK(x.d2h(out_sum, o + n, (size_t)n * sizeof(double)));
    HIPCHK(x.finish());
    return STP_OK;
}


int stp_window_plane(stp_ctx* ctx, const stp_band* band, int64_t row0, int32_t nrows, int64_t col0, int32_t ncols, double M,
                     double* out)
{
    if (!ctx || !band || !out || nrows <= 0 || ncols <= 0) return STP_E_ARG;
    if (row0 < 0 || col0 < 0 || row0 + nrows > band->nrows || col0 + ncols > band->nrows)
        return set_err(ctx, STP_E_ARG, "window outside the chromosome");
    if (col0 + ncols - 1 - row0 >= band->hw || row0 + nrows - 1 - col0 > band->hw)
        return set_err(ctx, STP_E_ARG, "window leaves the +-halfwidth band; build a wider band");
    HIPCHK(hipSetDevice(ctx->device));
    dev_buf bO;
    const size_t n = (size_t)nrows * ncols;
    HIPCHK(bO.alloc(ctx, n * sizeof(double)));
    {
        prof_scope ps(ctx, "window_plane", 16.0 * (double)n, ctx->aux);
        hipLaunchKernelGGL(k_window_plane, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535)), dim3(256), 0, ctx->aux, bref(band), row0,
                           nrows, col0, ncols, M, (double*)bO.p);
    }
    HIPCHK(hipGetLastError());
    stp_xfer x(ctx, ctx->aux);
    HIPCHK(x.d2h(out, bO.p, n * sizeof(double)));
    HIPCHK(x.finish());
    return STP_OK;
}

// ---------------------------------------------------------------------------------------------
// order statistics of the positive pixels (getQuantile_original)

int stp_select_create(stp_ctx* ctx, stp_select** out)
{
    if (!ctx || !out) return STP_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    stp_select* s = new (std::nothrow) stp_select();
    if (!s) return STP_E_NOMEM;
    if (stp_dmalloc(__LINE__, (void**)&s->state, sizeof(stp_sel_state)) != hipSuccess) { delete s; return set_err(ctx, STP_E_NOMEM, "hipMalloc(select state)"); }
    *out = s;
    return STP_OK;
}

void stp_select_free(stp_ctx* ctx, stp_select* s)
{
    if (!s) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    for (auto& c : s->chunks) { if (ctx) pool_release(ctx, c.first, (size_t)c.second * sizeof(double)); else (void)stp_dfree(__LINE__, c.first); }
    if (s->state) (void)stp_dfree(__LINE__, s->state);
    delete s;
}

int stp_select_append(stp_ctx* ctx, stp_select* s, const double* values_host, int64_t n)
{
    if (!ctx || !s || n < 0 || (n > 0 && !values_host)) return STP_E_ARG;
    if (n == 0) return STP_OK;
    HIPCHK(hipSetDevice(ctx->device));
    double* d = nullptr;
    HIPCHK(pool_alloc(ctx, (size_t)n * sizeof(double), (void**)&d));
    stp_xfer x(ctx, ctx->io);
    hipError_t e = x.h2d(d, values_host, (size_t)n * sizeof(double));
    if (e == hipSuccess) e = x.finish();
    if (e != hipSuccess) { pool_release(ctx, d, (size_t)n * sizeof(double)); return set_err(ctx, STP_E_HIP, "select append: upload failed"); }
    s->chunks.push_back(std::make_pair(d, (long long)n));
    s->npos = -1;
    return STP_OK;
}

int stp_select_append_pixels(stp_ctx* ctx, stp_select* s, const int64_t* bin1, const int64_t* bin2, const int32_t* count,
                             int64_t npix, const double* weight, int64_t nbins_total)
{
    return stp_select_append_pixels_ex(ctx, s, bin1, bin2, count, STP_COUNT_I32, npix, weight, nbins_total);
}

int stp_select_append_pixels_ex(stp_ctx* ctx, stp_select* s, const int64_t* bin1, const int64_t* bin2, const void* count_v,
                                int32_t count_type, int64_t npix, const double* weight, int64_t nbins_total)
{
    if (count_type != STP_COUNT_I32 && count_type != STP_COUNT_F64) return set_err(ctx, STP_E_ARG, "count_type must be STP_COUNT_I32 or STP_COUNT_F64");
    const size_t csz = count_type == STP_COUNT_F64 ? sizeof(double) : sizeof(int32_t);
    const char* count = (const char*)count_v;
    if (!ctx || !s || npix < 0 || (npix > 0 && (!bin1 || !bin2 || !count)) || (weight && nbins_total <= 0)) return STP_E_ARG;
    if (npix == 0) return STP_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t CH = (int64_t)1 << 23;                 // pixels per staged chunk
    const int64_t nch = npix < CH ? npix : CH;
    dev_buf d1, d2, dc, dw;
    host_pin pin1, pin2, pinc;
    pin1.pin(bin1, (size_t)npix * sizeof(int64_t), ctx->io);
    pin2.pin(bin2, (size_t)npix * sizeof(int64_t), ctx->io);
    pinc.pin(count, (size_t)npix * csz, ctx->io);
    stp_xfer x(ctx, ctx->io);                               // columns too small to pin (< 1 MB), the weights: staged
    auto up = [&](const host_pin& pin, void* dst, const void* src, size_t n) {
        return pin.lo != pin.hi ? pin.copy(dst, src, n, ctx->io) : x.h2d(dst, src, n);
    };
    HIPCHK(d1.alloc(ctx, (size_t)nch * sizeof(int64_t)));
    HIPCHK(d2.alloc(ctx, (size_t)nch * sizeof(int64_t)));
    HIPCHK(dc.alloc(ctx, (size_t)nch * csz));
    if (weight) {
        HIPCHK(dw.alloc(ctx, (size_t)nbins_total * sizeof(double)));
        HIPCHK(x.h2d(dw.p, weight, (size_t)nbins_total * sizeof(double)));
    }
    for (int64_t p0 = 0; p0 < npix; p0 += CH) {
        const int64_t n = npix - p0 < CH ? npix - p0 : CH;
        double* out = nullptr;
        HIPCHK(pool_alloc(ctx, (size_t)n * 2 * sizeof(double), (void**)&out));
        hipError_t e = up(pin1, d1.p, bin1 + p0, (size_t)n * sizeof(int64_t));
        if (e == hipSuccess) e = up(pin2, d2.p, bin2 + p0, (size_t)n * sizeof(int64_t));
        if (e == hipSuccess) e = up(pinc, dc.p, count + (size_t)p0 * csz, (size_t)n * csz);
        if (e == hipSuccess) {
            prof_scope ps(ctx, "select_pixels", 36.0 * n, ctx->io);
            if (count_type == STP_COUNT_F64)
                hipLaunchKernelGGL((k_sel_pixel_values<double, int64_t>), dim3(sel_grid(n)), dim3(256), 0, ctx->io, (const int64_t*)d1.p,
                                   (const int64_t*)d2.p, (const double*)dc.p, (long long)n, weight ? (const double*)dw.p : nullptr,
                                   (long long)nbins_total, out, 0ll);
            else
                hipLaunchKernelGGL((k_sel_pixel_values<int32_t, int64_t>), dim3(sel_grid(n)), dim3(256), 0, ctx->io, (const int64_t*)d1.p,
                                   (const int64_t*)d2.p, (const int32_t*)dc.p, (long long)n, weight ? (const double*)dw.p : nullptr,
                                   (long long)nbins_total, out, 0ll);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->io);      // the staging buffers are reused
        if (e != hipSuccess) { pool_release(ctx, out, (size_t)n * 2 * sizeof(double)); return set_err(ctx, STP_E_HIP, std::string("select append pixels: ") + hipGetErrorString(e)); }
        s->chunks.push_back(std::make_pair(out, (long long)(2 * n)));
        s->npos = -1;
    }
    return STP_OK;
}

static unsigned sel_grid(long long n)
{
    long long g = (n + 256 * 8 - 1) / (256 * 8);
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (unsigned)g;
}

int stp_select_count(stp_ctx* ctx, stp_select* s, int64_t* n_positive)
{
    if (!ctx || !s || !n_positive) return STP_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    if (s->npos < 0) {
        // the first radix pass (top digit of every positive value); its histogram stays on the device for the ranks
        HIPCHK(hipMemsetAsync(s->state, 0, sizeof(stp_sel_state), ctx->io));
        for (auto& c : s->chunks) {
            prof_scope ps(ctx, "select_hist", 8.0 * c.second, ctx->io);
            hipLaunchKernelGGL(k_sel_hist0, dim3(sel_grid(c.second)), dim3(256), 0, ctx->io, (const double*)c.first, c.second, s->state);
        }
        hipLaunchKernelGGL(k_sel_count, dim3(1), dim3(1024), 0, ctx->io, s->state);
        HIPCHK(hipGetLastError());
        unsigned long long k = 0;
        HIPCHK(hipMemcpyAsync(&k, &s->state->count, sizeof(k), hipMemcpyDeviceToHost, ctx->io));
        HIPCHK(hipStreamSynchronize(ctx->io));
        s->npos = (long long)k;
    }
    *n_positive = s->npos;
    return STP_OK;
}

int stp_select_ranks(stp_ctx* ctx, stp_select* s, const int64_t* ranks, int32_t nranks, double* out)
{
    if (!ctx || !s || !ranks || !out || nranks < 0) return STP_E_ARG;
    int64_t np = 0;
    int rc = stp_select_count(ctx, s, &np);
    if (rc) return rc;
    for (int r = 0; r < nranks; r++)
        if (ranks[r] < 0 || ranks[r] >= np) return set_err(ctx, STP_E_ARG, "rank outside [0, n_positive)");
    static bool lds_set = false;
    if (!lds_set) {       // up to 16 histograms of 8 KB in dynamic LDS
        (void)hipFuncSetAttribute((const void*)k_sel_hist, hipFuncAttributeMaxDynamicSharedMemorySize, STP_SEL_MAXR * STP_SEL_BINS * 4);
        lds_set = true;
    }
    // all ranks of a batch descend together: one sweep over the data per pass, one host round trip per batch
    for (int r0 = 0; r0 < nranks; r0 += STP_SEL_MAXR) {
        const int R = nranks - r0 < STP_SEL_MAXR ? nranks - r0 : STP_SEL_MAXR;
        struct { unsigned long long prefix[STP_SEL_MAXR], k[STP_SEL_MAXR]; int group[STP_SEL_MAXR]; int nranks, ngroups;
                 unsigned long long gprefix[STP_SEL_MAXR]; } init;
        memset(&init, 0, sizeof(init));
        for (int r = 0; r < R; r++) init.k[r] = (unsigned long long)ranks[r0 + r];
        init.nranks = R; init.ngroups = 1;
        static_assert(offsetof(stp_sel_state, hist0) == sizeof(init), "select state header layout");
        HIPCHK(hipMemcpyAsync(s->state, &init, sizeof(init), hipMemcpyHostToDevice, ctx->io));
        hipLaunchKernelGGL(k_sel_pick, dim3(1), dim3(1024), 0, ctx->io, s->state, 0);          // first pass: the kept histogram
        for (int pass = 1; pass < STP_SEL_PASSES; pass++) {
            for (auto& c : s->chunks) {
                prof_scope ps(ctx, "select_hist", 8.0 * c.second, ctx->io);
                hipLaunchKernelGGL(k_sel_hist, dim3(sel_grid(c.second)), dim3(512), (size_t)R * STP_SEL_BINS * 4, ctx->io,
                                   (const double*)c.first, c.second, pass, R, s->state);
            }
            hipLaunchKernelGGL(k_sel_pick, dim3(1), dim3(1024), 0, ctx->io, s->state, pass);
        }
        HIPCHK(hipGetLastError());
        unsigned long long keys[STP_SEL_MAXR];
        HIPCHK(hipMemcpyAsync(keys, s->state->prefix, R * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->io));
        HIPCHK(hipStreamSynchronize(ctx->io));
        memcpy(out + r0, keys, R * sizeof(double));
    }
    return STP_OK;
}


int stp_remove_redundant(stp_ctx* ctx, int64_t n, const int64_t* pos1, const int64_t* pos2, const int64_t* pos3,
                         const int64_t* pos4, const int32_t* h, const int32_t* w, const double* key, int32_t by,
                         const int32_t* order, const int32_t* b0, const int32_t* b1, const int32_t* b2, uint8_t* keep)
{
    if (!ctx || n < 0 || by < 0 || by > 2) return STP_E_ARG;
    if (n == 0) return STP_OK;
    if (!pos1 || !pos2 || !pos3 || !pos4 || !h || !w || !order || !b0 || !b1 || !b2 || !keep || (by != 0 && !key)) return STP_E_ARG;
    for (int64_t i = 0; i < n; i++) {
        if (b0[i] < 0 || b0[i] > b1[i] || b1[i] > b2[i] || b2[i] > n || order[i] < 0 || order[i] >= n)
            return set_err(ctx, STP_E_ARG, "remove_redundant: bad bucket table");
        if (pos2[i] == pos1[i] || pos4[i] == pos3[i]) return set_err(ctx, STP_E_ARG, "remove_redundant: zero-length stripe (the reference divides by zero)");
    }
    HIPCHK(hipSetDevice(ctx->device));
    dev_buf bP, bI, bK, bD;
    HIPCHK(bP.alloc(ctx, (size_t)n * 4 * sizeof(long long)));
    HIPCHK(bI.alloc(ctx, (size_t)n * 6 * sizeof(int)));
    HIPCHK(bK.alloc(ctx, (size_t)n * sizeof(double)));
    HIPCHK(bD.alloc(ctx, (size_t)n * sizeof(unsigned int)));
    long long* dp = (long long*)bP.p;
    int* di = (int*)bI.p;
    const int64_t* ps[4] = {pos1, pos2, pos3, pos4};
    stp_xfer x(ctx, ctx->stream);
    for (int k = 0; k < 4; k++) HIPCHK(x.h2d(dp + k * n, ps[k], (size_t)n * 8));
    const int32_t* is[6] = {h, w, order, b0, b1, b2};
    for (int k = 0; k < 6; k++) HIPCHK(x.h2d(di + k * n, is[k], (size_t)n * 4));
    if (key) HIPCHK(x.h2d(bK.p, key, (size_t)n * 8));
    HIPCHK(hipMemsetAsync(bD.p, 0, (size_t)n * sizeof(unsigned int), ctx->stream));
    {
        prof_scope ps2(ctx, "remove_redundant", 48.0 * n);
        hipLaunchKernelGGL(k_remove_redundant, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (long long)n, dp, dp + n,
                           dp + 2 * n, dp + 3 * n, di, di + n, (const double*)bK.p, by, di + 2 * n, di + 3 * n, di + 4 * n,
                           di + 5 * n, (unsigned int*)bD.p);
    }
    HIPCHK(hipGetLastError());
    std::vector<unsigned int> hd((size_t)n);
    HIPCHK(x.d2h(hd.data(), bD.p, (size_t)n * sizeof(unsigned int)));
    HIPCHK(x.finish());
    for (int64_t i = 0; i < n; i++) keep[i] = hd[i] ? 0 : 1;
    return STP_OK;
}

int stp_set_profiling(stp_ctx* ctx, int on)
{
    if (!ctx) return STP_E_ARG;
    ctx->profiling = on != 0;
    return STP_OK;
}

int stp_get_stats(stp_ctx* ctx, stp_kernel_stat* out, int32_t capacity, int32_t* count)
{
    if (!ctx || !count) return STP_E_ARG;
    resolve_pending(ctx);
    *count = (int32_t)ctx->stats.size();
    if ((int32_t)ctx->stats.size() > capacity) return STP_E_CAPACITY;
    for (size_t i = 0; i < ctx->stats.size(); i++) {
        memset(&out[i], 0, sizeof(out[i]));
        strncpy(out[i].name, ctx->stats[i].name.c_str(), sizeof(out[i].name) - 1);
        out[i].launches = ctx->stats[i].launches;
        out[i].ms_total = ctx->stats[i].ms;
        out[i].alg_bytes = ctx->stats[i].bytes;
    }
    return STP_OK;
}

int stp_reset_stats(stp_ctx* ctx)
{
    if (!ctx) return STP_E_ARG;
    resolve_pending(ctx);
    ctx->stats.clear();
    return STP_OK;
}

// Result tables as text (stp_tsv.h): host code only.
int stp_format_tsv(int32_t ncols, const int32_t* kind, const void* const* data, const char* const* strtab,
                   const int64_t* const* stroff, const int32_t* nstr, int64_t nrows, char* out, int64_t cap, int64_t* out_len)
{
    if (ncols < 0 || nrows < 0 || cap < 0 || !out_len || (ncols > 0 && (!kind || !data)) || (cap > 0 && !out)) return STP_E_ARG;
    *out_len = 0;
    // widest a row can get: decides once whether the buffer is large enough (no test per field)
    int64_t roww = 1;
    for (int c = 0; c < ncols; c++) {
        if (!data[c] && nrows > 0) return STP_E_ARG;
        if (kind[c] == STP_COL_I64) roww += 21;
        else if (kind[c] == STP_COL_F64) roww += 26;
        else if (kind[c] == STP_COL_STR) {
            if (!strtab || !stroff || !nstr || !strtab[c] || !stroff[c] || nstr[c] < 0) return STP_E_ARG;
            int64_t w = 0;
            for (int k = 0; k < nstr[c]; k++) {
                const int64_t l = stroff[c][k + 1] - stroff[c][k];
                if (l < 0) return STP_E_ARG;
                w = l > w ? l : w;
            }
            roww += w + 1;
        } else return STP_E_ARG;
    }
    if (ncols == 0) return STP_OK;                       // (pandas writes one empty line per row of a table without columns: left to it)
    if (nrows > 0 && roww > cap / nrows) return STP_E_CAPACITY;
    char* p = out;
    for (int64_t r = 0; r < nrows; r++) {
        for (int c = 0; c < ncols; c++) {
            if (c) *p++ = '\t';
            if (kind[c] == STP_COL_I64) p = stp_tsv_i64(p, ((const int64_t*)data[c])[r]);
            else if (kind[c] == STP_COL_F64) p = stp_tsv_repr(p, ((const double*)data[c])[r]);
            else {
                const int32_t k = ((const int32_t*)data[c])[r];
                if (k < 0 || k >= nstr[c]) return STP_E_ARG;
                const int64_t o = stroff[c][k], l = stroff[c][k + 1] - o;
                memcpy(p, strtab[c] + o, (size_t)l);
                p += l;
            }
        }
        *p++ = '\n';
    }
    *out_len = (int64_t)(p - out);
    return STP_OK;
}

#pragma GCC visibility pop
}  // extern "C"
