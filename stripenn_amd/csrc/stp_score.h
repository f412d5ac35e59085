// stp_score.h -- kernels of the score path: expected values (mpmean), background windows
// (nulldist), per-stripe p-value, Stripiness, observed mean, and the per-frame medpixel.
// Included by stripenn_hip.hip only (device code).
//
// Summation order: the p-value is a RANK against the background tables, so the window / row
// means that feed it reproduce numpy's reduction order exactly (pairwise inner loop of
// numpy/core/src/umath/loops_utils.h.src, rows accumulated in order).  Stripiness and the
// observed mean are plain floating-point statistics (tolerance 1e-4 relative in the contract);
// they use the same helpers where that is free.
#pragma once
#include "../../include/stripenn_hip.h"
#include "stp_phases.h"

#define STP_SCORE_MAXROWS 3072   /* LDS-bound: 4 x 8 B per row (k_pvalue), ~40 B per row (k_stripiness); 15 Mb at 5 kb */
#define STP_SCORE_MAXCOLS 1024

struct stp_bandref {
    const double* d;
    int64_t nrows;
    int W, hw;
    int sym;      // 1: the band has been verified bit for bit symmetric (k_band_symcheck) -- M[r][c] may be read as M[c][r]
};

// Is the band symmetric, M[i][i + d] == M[i + d][i] bit for bit for 0 < d < hw?  Every band of a contact map is (cooler stores
// the upper triangle; the packer writes one value into both cells), but the ABI takes any band, so it is checked -- once per
// band -- before a kernel relies on it.  Why it matters: a stripe is a tall, narrow rectangle.  Its rows lie W - 1 doubles
// apart in the band (lane = row: 64 cache lines per wave load), but by symmetry column c of the matrix IS row c of the band:
// M[r][c] = band[c][r - c + hw], consecutive rows at consecutive addresses (lane = row: one 512-byte run per wave load).
__global__ __launch_bounds__(256) void k_band_symcheck(const double* __restrict__ band, int64_t nrows, int W, int hw, int* __restrict__ asym)
{
    const int64_t n = nrows * (int64_t)(hw - 1);
    int bad = 0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n && !bad; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = p / (hw - 1);
        const int d = (int)(p - i * (hw - 1)) + 1;
        if (i + d >= nrows) continue;
        const unsigned long long a = (unsigned long long)__double_as_longlong(band[i * (int64_t)W + hw + d]);
        const unsigned long long b = (unsigned long long)__double_as_longlong(band[(i + d) * (int64_t)W + hw - d]);
        if (a != b) bad = 1;
    }
    if (bad) atomicOr(asym, 1);
}

// M[r][c] with NaN kept; 0 outside the band / chromosome
__device__ __forceinline__ double band_at(const stp_bandref& B, int64_t r, int64_t c)
{
    int64_t dd = c - r + B.hw;
    if (r < 0 || r >= B.nrows || c < 0 || c >= B.nrows || dd < 0 || dd >= B.W) return 0.0;
    return B.d[r * (int64_t)B.W + dd];
}
__device__ __forceinline__ double band_at0(const stp_bandref& B, int64_t r, int64_t c)
{
    double v = band_at(B, r, c);
    return (v != v) ? 0.0 : v;   // nantozero
}

// numpy pairwise sum of n values get(0..n)
template <class F>
__device__ double stp_pw_leaf(F get, int64_t o, int n)
{
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += get(o + i);
        return res;
    }
    double r0 = get(o), r1 = get(o + 1), r2 = get(o + 2), r3 = get(o + 3), r4 = get(o + 4), r5 = get(o + 5),
           r6 = get(o + 6), r7 = get(o + 7);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += get(o + i); r1 += get(o + i + 1); r2 += get(o + i + 2); r3 += get(o + i + 3);
        r4 += get(o + i + 4); r5 += get(o + i + 5); r6 += get(o + i + 6); r7 += get(o + i + 7);
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += get(o + i);
    return res;
}
// BIG = false: the caller guarantees n <= 128 (one leaf, no stack, no scratch memory).
// BIG = true : numpy's recursion pw(o,n) = pw(o,n2) + pw(o+n2,n-n2), n2 = n/2 - (n/2)%8, evaluated in
// post-order with an explicit stack that the caller provides (LDS or private memory), depth <= 16.
struct stp_pw_frame {
    long long o, n;
    double v;
    int state;
};
template <bool BIG, class F>
__device__ double stp_pw(F get, int64_t o, int64_t n, stp_pw_frame* stk = nullptr)
{
    if (!BIG || n <= 128) return stp_pw_leaf(get, o, (int)n);
    int sp = 0;
    stk[0].o = o; stk[0].n = n; stk[0].state = 0;
    double ret = 0.0;
    while (sp >= 0) {
        const int64_t cn = stk[sp].n, co = stk[sp].o;
        if (cn <= 128) { ret = stp_pw_leaf(get, co, (int)cn); sp--; continue; }
        int64_t n2 = cn / 2; n2 -= n2 % 8;
        if (stk[sp].state == 0) { stk[sp].state = 1; stk[sp + 1].o = co; stk[sp + 1].n = n2; stk[sp + 1].state = 0; sp++; }
        else if (stk[sp].state == 1) { stk[sp].v = ret; stk[sp].state = 2; stk[sp + 1].o = co + n2; stk[sp + 1].n = cn - n2; stk[sp + 1].state = 0; sp++; }
        else { ret = stk[sp].v + ret; sp--; }
    }
    return ret;
}

__device__ __forceinline__ void stp_pyslice(int64_t a, int64_t b, int64_t n, int64_t* lo, int64_t* hi)
{
    if (a < 0) { a += n; if (a < 0) a = 0; }
    if (a > n) a = n;
    if (b < 0) { b += n; if (b < 0) b = 0; }
    if (b > n) b = n;
    if (b < a) b = a;
    *lo = a; *hi = b;
}

// ---------------------------------------------------------------------------------------------
// getStripe.mpmean (getStripe.py:198-207): per 400-row frame, per diagonal j, the row-ordered sum
__global__ __launch_bounds__(448) void k_diag_sums(stp_bandref B, double* __restrict__ psum, long long* __restrict__ pcnt)
{
    const int f = blockIdx.x, j = threadIdx.x;
    if (j >= STP_NDIAG) return;
    const int64_t r0 = (int64_t)f * 400;
    double s = 0.0;
    long long c = 0;
    for (int i = 0; i < 400; i++) {
        int64_t r = r0 + i;
        if (r >= B.nrows || r + j >= B.nrows) break;
        double v = B.d[r * (int64_t)B.W + B.hw + j];
        if (v != v) v = 0.0;
        s += v;
        c++;
    }
    psum[(size_t)f * STP_NDIAG + j] = s;
    pcnt[(size_t)f * STP_NDIAG + j] = c;
}

// ---------------------------------------------------------------------------------------------
// getStripe.nulldist window means (getStripe.py:347-378).  mat = M[row0:row0+nrow, col0:col0+ncol]
// with NaN -> 0; python slice semantics on mat's extents; np.mean order (8192-element buffers).

// `dense` (may be null): the unit matrix itself, nrow x ncol row-major, NaN preserved.  It is supplied
// by the host for batches in which a Python slice wraps around (negative start) and therefore reads
// columns far outside the diagonal band; otherwise the resident band is read.
template <bool BIG>
__device__ double null_block_mean(const stp_bandref& B, const double* __restrict__ dense, const stp_null_sample& s,
                                  int64_t r0, int64_t r1, int64_t c0, int64_t c1)
{
    stp_pw_frame stk_store[BIG ? 16 : 1];
    stp_pw_frame* stk = stk_store;
    int64_t rl, rh, cl, ch;
    stp_pyslice(r0, r1, s.nrow, &rl, &rh);
    stp_pyslice(c0, c1, s.ncol, &cl, &ch);
    const int64_t cnt = (rh - rl) * (ch - cl);
    double acc = 0.0;
    if (cnt > 0) {
        // numpy buffers the strided slice 8192 elements at a time and runs one pairwise loop over each buffer of
        // the row-major flattened block; the partial sums of a block larger than that (bs > 90) add up in order
        const int64_t w = ch - cl;
        for (int64_t o = 0; o < cnt; o += 8192) {
            const int64_t nn = cnt - o < 8192 ? cnt - o : 8192;
            double part;
            if (dense) {
                const double* base = dense + rl * (int64_t)s.ncol + cl;
                const int64_t nc = s.ncol;
                part = stp_pw<BIG>([&](int64_t k) { double v = base[(k / w) * nc + (k % w)]; return (v != v) ? 0.0 : v; }, o, nn, stk);
            } else {
                const int64_t gr0 = s.row0 + rl, gc0 = s.col0 + cl;
                part = stp_pw<BIG>([&](int64_t k) { return band_at0(B, gr0 + k / w, gc0 + k % w); }, o, nn, stk);
            }
            acc = (o == 0) ? part : acc + part;
        }
    }
    return acc / (double)cnt;
}

template <bool BIG>
__global__ __launch_bounds__(256) void k_null_windows(stp_bandref B, const double* __restrict__ dense,
                                                       const stp_null_sample* __restrict__ samp, int n, int bs,
                                                       double* __restrict__ lu, double* __restrict__ ru,
                                                       double* __restrict__ ld, double* __restrict__ rd)
{
    const int i = blockIdx.x;
    const stp_null_sample s = samp[i];
    const int up = bs / 2, down = bs - up;
    const int64_t x = s.x;
    for (int it = threadIdx.x; it < 2 * STP_NDIAG; it += blockDim.x) {
        const int j = it >> 1, dn = it & 1;
        const int64_t y = (dn ? x + j : x - j) + s.yoff;
        double l = null_block_mean<BIG>(B, dense, s, x - up - bs, x - up, y - up, y + down);
        double c = null_block_mean<BIG>(B, dense, s, x - up, x + down, y - up, y + down);
        double r = null_block_mean<BIG>(B, dense, s, x + down, x + down + bs, y - up, y + down);
        double* L = dn ? ld : lu;
        double* R = dn ? rd : ru;
        L[(size_t)j * n + i] = c - l;
        R[(size_t)j * n + i] = c - r;
    }
}

// ---------------------------------------------------------------------------------------------
// getStripe.pvalue (getStripe.py:552-605) for one stripe per workgroup

__device__ __forceinline__ double wave_sum_i(int v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return (double)v;
}

// median of vals[0..n) (np.median: mean of the two middle order statistics)
__device__ double block_median(const double* vals, int n, double* s_res)
{
    const int k0 = (n - 1) / 2, k1 = n / 2;
    if (threadIdx.x == 0) { s_res[0] = 0.0; s_res[1] = 0.0; }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = vals[i];
        int less = 0, leq = 0;
        for (int k = 0; k < n; k++) { less += vals[k] < v; leq += vals[k] <= v; }
        if (less <= k0 && k0 < leq) s_res[0] = v;
        if (less <= k1 && k1 < leq) s_res[1] = v;
    }
    __syncthreads();
    return (k0 == k1) ? s_res[0] : (s_res[0] + s_res[1]) / 2.0;
}

// Background rows sorted once per upload (ascending, NaN last): the rank "#(b >= x)" of :599-600
// becomes nvalid - lower_bound(x).  One workgroup per table row, bitonic sort of <= 2048 values in LDS.
#define STP_BG_MAXCOL 2048
// `sorted_t` (round 5, may be null): the same sorted rows with the ROW index fastest -- element (table, column, row) at
// (table * ncol + column) * STP_NDIAG + row.  k_score_wave's lanes search consecutive rows in lock step: in this layout the
// first steps of all lanes (equal or neighbouring pivots) read one or a few cache lines instead of 64.
__global__ __launch_bounds__(512) void k_bg_sort(const double* __restrict__ bg, int ncol, double* __restrict__ sorted,
                                                  int* __restrict__ nvalid, double* __restrict__ sorted_t)
{
    __shared__ double v[STP_BG_MAXCOL];
    __shared__ int s_nv;
    const size_t row = blockIdx.x;
    if (threadIdx.x == 0) s_nv = 0;
    __syncthreads();
    int loc = 0;
    for (int i = threadIdx.x; i < STP_BG_MAXCOL; i += blockDim.x) {
        double x = (i < ncol) ? bg[row * ncol + i] : NAN;
        if (x == x) loc++;
        v[i] = (x == x) ? x : INFINITY;      // NaN and padding sort to the end (+inf sentinels; counted out by nvalid)
    }
    atomicAdd(&s_nv, loc);
    __syncthreads();
    for (int k = 2; k <= STP_BG_MAXCOL; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < STP_BG_MAXCOL; i += blockDim.x) {
                int ixj = i ^ j;
                if (ixj > i) {
                    double a = v[i], b = v[ixj];
                    bool up = ((i & k) == 0);
                    if ((a > b) == up) { v[i] = b; v[ixj] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < ncol; i += blockDim.x) sorted[row * ncol + i] = v[i];
    if (sorted_t) {
        const size_t tab = row / STP_NDIAG, d = row - tab * STP_NDIAG;
        for (int i = threadIdx.x; i < ncol; i += blockDim.x) sorted_t[(tab * ncol + i) * STP_NDIAG + d] = v[i];
    }
    if (threadIdx.x == 0) nvalid[row] = s_nv;
}

// number of valid background values >= x  (0 when x is NaN, like numpy's comparison)
__device__ __forceinline__ int bg_count_ge(const double* __restrict__ srt, int nv, double x)
{
    if (x != x) return 0;
    int lo = 0, hi = nv;                      // first index with srt[idx] >= x
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (srt[mid] < x) lo = mid + 1; else hi = mid;
    }
    return nv - lo;
}

template <bool BIG>
__global__ __launch_bounds__(256) void k_pvalue(stp_bandref B, const double* __restrict__ srt /* sorted lu,ru,ld,rd */,
                                                 const int* __restrict__ nvalid, int ncolbg, int bs,
                                                 const stp_pv_stripe* __restrict__ st, double* __restrict__ out, int HR,
                                                 const int* __restrict__ idx = nullptr /* stripes of this launch (null: all, in order) */)
{
    // LDS sized by the host for the tallest stripe of the batch: pv[HR] | part[3][HR]
    extern __shared__ double s_dyn[];
    double* const pv = s_dyn;
    double* const part[3] = {s_dyn + HR, s_dyn + 2 * HR, s_dyn + 3 * HR};
    __shared__ double s_res[2];
    const int si = idx ? idx[blockIdx.x] : (int)blockIdx.x;
    const stp_pv_stripe s = st[si];
    const int h = s.row1 - s.row0;
    const int ncol = s.col1 - s.col0;
    int64_t lo[3], hi[3];
    stp_pyslice(bs, -bs, ncol, &lo[0], &hi[0]);        // mat[:, bs:-bs]
    stp_pyslice(0, bs, ncol, &lo[1], &hi[1]);          // mat[:, :bs]
    stp_pyslice(-bs, ncol, ncol, &lo[2], &hi[2]);      // mat[:, -bs:]
    // row means of the three column blocks: np.mean(axis=1) = per-row pairwise sum / n
    for (int it = threadIdx.x; it < 3 * h; it += blockDim.x) {
        const int p = it / h, j = it - p * h;
        const int64_t gr = s.row0 + j;
        const int64_t n = hi[p] - lo[p];
        stp_pw_frame stk[BIG ? 16 : 1];
        part[p][j] = stp_pw<BIG>([&](int64_t k) { return band_at0(B, gr, s.col0 + k); }, lo[p], n, stk) / (double)n;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < h; j += blockDim.x) {
        const double ldv = part[0][j] - part[1][j], rdv = part[0][j] - part[2][j];
        int d, tab;
        if (s.mode == 0) { d = j; tab = 1; }
        else if (s.mode == 1) { d = s.upbase - j - 1; tab = 0; }
        else { d = s.fixed_row; tab = s.fixed_tab; }
        if (s.mode != 2 && d >= 400) d = 399;
        if (d < 0) d += STP_NDIAG;               // python negative index on the table (never for real stripes)
        const size_t rl = (size_t)(tab ? 2 : 0) * STP_NDIAG + d, rr = (size_t)(tab ? 3 : 1) * STP_NDIAG + d;
        const int vL = nvalid[rl], vR = nvalid[rr];
        const double p1 = (double)bg_count_ge(srt + rl * ncolbg, vL, ldv) / (double)vL;
        const double p2 = (double)bg_count_ge(srt + rr * ncolbg, vR, rdv) / (double)vR;
        double p = (p2 > p1) ? p2 : p1;          // python max(p1, p2)
        if (p == 0.0) p = 1.0 / (double)ncolbg;
        pv[j] = p;
    }
    __syncthreads();
    double med = block_median(pv, h, s_res);
    if (threadIdx.x == 0) out[si] = med;
}

// ---------------------------------------------------------------------------------------------
// getStripe.scoringstripes.iterate_idx (getStripe.py:661-759) for one stripe per workgroup

template <bool BIG>
__global__ __launch_bounds__(256) void k_stripiness(stp_bandref B, const double* __restrict__ exval,
                                                     const stp_score_stripe* __restrict__ st, double* __restrict__ out_g,
                                                     double* __restrict__ out_mean, double* __restrict__ out_total,
                                                     int* __restrict__ out_status, int HR, int CW,
                                                     const int* __restrict__ idx = nullptr)
{
    // LDS sized by the host for the tallest / widest stripe of the batch (HR rows, CW columns per block):
    // ex[400] | rowm[3][HR] | diff[max(HR, 256)] | keepr[HR] | keepc[3][CW] | rowdel[HR]
    extern __shared__ double s_dyn[];
    double* const ex = s_dyn;
    double* const rowm_base = ex + STP_NDIAG;
    double* const rowm[3] = {rowm_base, rowm_base + HR, rowm_base + 2 * HR};
    double* const diff = rowm_base + 3 * HR;
    int16_t* const keepr = (int16_t*)(diff + (HR > 256 ? HR : 256));
    int16_t* const keepc_base = keepr + HR;
    int16_t* const keepc[3] = {keepc_base, keepc_base + CW, keepc_base + 2 * CW};
    uint8_t* const rowdel = (uint8_t*)(keepc_base + 3 * CW);
    __shared__ int nkc[3], nkr;
    __shared__ double s_res[2];
    __shared__ double s_tot;
    __shared__ stp_score_stripe s_stripe;            // indexed with run-time block numbers: keep it out of scratch
    __shared__ stp_pw_frame s_stk[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int si = idx ? idx[blockIdx.x] : (int)blockIdx.x;
    if (tid == 0) s_stripe = st[si];
    __syncthreads();
    const stp_score_stripe& s = s_stripe;
    const int h = s.row1 - s.row0;
    for (int i = tid; i < STP_NDIAG; i += nt) ex[i] = exval[i];
    for (int i = tid; i < h; i += nt) rowdel[i] = 0;
    __syncthreads();
    // observed / expected with masking; NaN where the observed pixel is NaN or masked
    auto oe = [&](int b, int r, int c) -> double {
        bool masked = (c >= s.mcol0[b] && c <= s.mcol1[b]) || (r >= s.mrow0 && r <= s.mrow1);
        if (masked) return NAN;
        double o = band_at(B, s.row0 + r, s.col0[b] + c);
        int idx = (s.ex0[b] + c) - (s.ey0 + r);
        if (idx < 0) idx = -idx;
        if (idx >= 400) idx = 399;
        double e = ex[idx] + .00000001;
        return o / e;
    };
    // all-NaN columns (:709-711): a column is alive when any of its pixels is not NaN (all lanes work
    // on pixels; a benign write race sets the flag)
    if (tid < 3) nkc[tid] = 0;
    for (int b = 0; b < 3; b++)
        for (int c = tid; c < CW; c += nt) keepc[b][c] = 0;
    __syncthreads();
    // (whether o / e is NaN needs no division when e is finite and positive: then it is NaN exactly when o is)
    auto oe_alive = [&](int b, int r, int c) -> bool {
        const bool masked = (c >= s.mcol0[b] && c <= s.mcol1[b]) || (r >= s.mrow0 && r <= s.mrow1);
        if (masked) return false;
        const double o = band_at(B, s.row0 + r, s.col0[b] + c);
        int idx = (s.ex0[b] + c) - (s.ey0 + r);
        if (idx < 0) idx = -idx;
        if (idx >= 400) idx = 399;
        const double e = ex[idx] + .00000001;
        if (e > 0.0 && e < INFINITY) return o == o;
        const double v = o / e;
        return v == v;
    };
    for (int b = 0; b < 3; b++) {
        const int w = s.col1[b] - s.col0[b];
        for (int i = tid; i < w * h; i += nt) {
            const int r = i / w, c = i - r * w;
            if (oe_alive(b, r, c)) keepc[b][c] = 1;
        }
    }
    __syncthreads();
    // the rows dead columns delete (:713-733), compaction of kept columns / rows.  Ordered compactions by ballot:
    // wave b takes column block b (a chunk's writes never pass its reads: in place), then wave 0 the rows.
    __shared__ int s_anydel, s_status;
    if (tid == 0) { s_anydel = 0; s_status = 0; s_tot = 0.0; }
    __syncthreads();
    {
        const int lane = tid & 63, wv = tid >> 6, nwv = nt >> 6;
        for (int b = wv; b < 3; b += nwv) {
            const int w = s.col1[b] - s.col0[b];
            int base = 0;
            for (int c0 = 0; c0 < w; c0 += 64) {
                const int c = c0 + lane;
                const bool in = c < w;
                const bool alive = in && keepc[b][c] != 0;
                const unsigned long long bal = __ballot(alive);
                __builtin_amdgcn_wave_barrier();
                if (alive) keepc[b][base + __popcll(bal & ((1ull << lane) - 1ull))] = (int16_t)c;
                if (in && !alive) {
                    int rd = s.mirror ? (h - 1 - c) : c;
                    if (rd < -h || rd >= h) s_status = 1;           // np.delete: index out of bounds (IndexError)
                    if (rd < 0) rd += h;
                    if (rd >= 0 && rd < h) { rowdel[rd] = 1; s_anydel = 1; }
                }
                base += __popcll(bal);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (lane == 0) nkc[b] = base;
        }
    }
    __syncthreads();
    if (tid == 0 && out_status) out_status[si] = s_status;
    if (s_anydel) {
        if (tid < 64) {
            const int lane = tid;
            int base = 0;
            for (int r0 = 0; r0 < h; r0 += 64) {
                const int r = r0 + lane;
                const bool keep = r < h && !rowdel[r];
                const unsigned long long bal = __ballot(keep);
                if (keep) keepr[base + __popcll(bal & ((1ull << lane) - 1ull))] = (int16_t)r;
                base += __popcll(bal);
            }
            if (lane == 0) nkr = base;
        }
    } else {
        for (int r = tid; r < h; r += nt) keepr[r] = (int16_t)r;
        if (tid == 0) nkr = h;
    }
    __syncthreads();
    const int hk = nkr;
    // row means after nantozero (:739-745), numpy pairwise order over the kept columns
    for (int it = tid; it < 3 * hk; it += nt) {
        const int b = it / hk, q = it - b * hk, r = keepr[q];
        const int n = nkc[b];
        stp_pw_frame stk[BIG ? 16 : 1];
        double sum = stp_pw<BIG>([&](int64_t k) { double v = oe(b, r, keepc[b][k]); return (v != v) ? 0.0 : v; }, 0, n, stk);
        rowm[b][q] = sum / (double)n;
    }
    __syncthreads();
    // centerTotal / centerMean (:747-748): sum over center[rows repeated once per kept column]
    // (tolerance statistic: tree reduction of the row sums)
    {
        double loc = 0.0;
        const int n = nkc[0];
        for (int q = tid; q < hk; q += nt) loc += rowm[0][q] * (double)n;
        diff[tid] = loc;             // diff[] is free until the scores below (nt <= MAXROWS)
        __syncthreads();
        for (int o = nt >> 1; o > 0; o >>= 1) {
            if (tid < o) diff[tid] += diff[tid + o];
            __syncthreads();
        }
        if (tid == 0) s_tot = diff[0];
        __syncthreads();
    }
    // Sobel-like scores (:750-756, stats.py:184-199)
    for (int i = tid + 1; i < hk - 1; i += nt) {
        const double* cm = rowm[0]; const double* lm = rowm[1]; const double* rm = rowm[2];
        double gxl = 0.0, gxr = 0.0, gy = 0.0;
        gxl += (-1.0 * lm[i - 1] + -2.0 * lm[i]) + -1.0 * lm[i + 1];
        gxl += (1.0 * cm[i - 1] + 2.0 * cm[i]) + 1.0 * cm[i + 1];
        gxr += (1.0 * cm[i - 1] + 2.0 * cm[i]) + 1.0 * cm[i + 1];
        gxr += (-1.0 * rm[i - 1] + -2.0 * rm[i]) + -1.0 * rm[i + 1];
        gy += (1.0 * lm[i - 1] + 0.0 * lm[i]) + -1.0 * lm[i + 1];
        gy += (2.0 * cm[i - 1] + 0.0 * cm[i]) + -2.0 * cm[i + 1];
        gy += (1.0 * rm[i - 1] + 0.0 * rm[i]) + -1.0 * rm[i + 1];
        if (gy < 0) gy *= -1;
        double gx = (gxl < gxr || gxl != gxl) ? gxl : gxr;   // np.minimum (NaN propagates)
        if (gxr != gxr) gx = gxr;
        diff[i - 1] = gx - gy;
    }
    __syncthreads();
    const int nd = hk - 2 > 0 ? hk - 2 : 0;
    double med = 0.0;
    // np.nanmedian(centerm): the non-NaN values into rowm[1] (left means are no longer needed; the median does not
    // depend on their order)
    __shared__ int s_nm, s_ndk;
    __shared__ double s_acc[8];
    if (tid == 0) s_nm = 0;
    __syncthreads();
    for (int q = tid; q < hk; q += nt) {
        const double v = rowm[0][q];
        if (v == v) rowm[1][atomicAdd(&s_nm, 1)] = v;
    }
    // diff = [x for x in diff if x >= 0 or x < 0] in order, into rowm[2] (right means are no longer needed)
    if (tid < 64) {
        const int lane = tid;
        int base = 0;
        for (int q0 = 0; q0 < nd; q0 += 64) {
            const int q = q0 + lane;
            const double v = q < nd ? diff[q] : NAN;
            const bool keep = v == v;
            const unsigned long long bal = __ballot(keep);
            if (keep) rowm[2][base + __popcll(bal & ((1ull << lane) - 1ull))] = v;
            base += __popcll(bal);
        }
        if (lane == 0) s_ndk = base;
    }
    __syncthreads();
    if (s_nm > 0) med = block_median(rowm[1], s_nm, s_res);
    else med = NAN;
    // np.mean(diff): numpy's pairwise sum; up to 128 values it is eight stride-8 partial sums combined in a fixed tree plus
    // the tail in order -- the partial sums by eight lanes, everything else (and longer lists) by one thread
    const int n = s_ndk;
    const double* dk = rowm[2];
    if (n >= 8 && n <= 128 && tid < 8) {
        double r = dk[tid];
        for (int i = 8; i < n - (n % 8); i += 8) r += dk[i + tid];
        s_acc[tid] = r;
    }
    __syncthreads();
    if (tid == 0) {
        double sum;
        if (n >= 8 && n <= 128) {
            sum = ((s_acc[0] + s_acc[1]) + (s_acc[2] + s_acc[3])) + ((s_acc[4] + s_acc[5]) + (s_acc[6] + s_acc[7]));
            for (int i = n - (n % 8); i < n; i++) sum += dk[i];
        } else {
            sum = stp_pw<true>([&](int64_t k) { return dk[k]; }, 0, n, s_stk);
        }
        const double avg = sum / (double)n;
        out_g[si] = med * avg;
        const int nc = nkc[0];
        // every centre column deleted (a mask covering the whole stripe width): np.sum over the empty block is 0.0,
        // np.mean is NaN (getStripe.py:747-748)
        out_total[si] = (nc == 0) ? 0.0 : s_tot * (double)nc;
        out_mean[si] = (nc == 0) ? NAN : (s_tot * (double)nc) / ((double)hk * nc * nc);
    }
}

// ---------------------------------------------------------------------------------------------
// k_score_wave (round 5): p-value and / or Stripiness of one stripe per WAVE -- four stripes per workgroup, no workgroup
// barrier after the expected-value table is in LDS.  k_pvalue / k_stripiness give a stripe a whole workgroup and walk it in
// ~8 / ~20 barrier-separated phases; a candidate stripe is ~60 rows x ~28 columns, so most of their time is barriers, index
// divisions, bounds tests per element and a quadratic median spread over idle lanes.  Here lane = row (up to SW_KR rows per
// lane), every loop over columns is wave-uniform, and
//   * the host has checked the rectangles against the band (check_rect), so pixels are read without bounds tests;
//   * column occupancy (the all-NaN-column rule, getStripe.py:709-711) is one ballot per column, kept as a 64-bit mask;
//   * the medians are rank selections whose counts are ballots (scalar popcounts), ending as soon as both middle order
//     statistics are known;
//   * np.mean(diff) keeps numpy's pairwise order (eight stride-8 partial sums per leaf, two leaves beyond 128 values).
// Same operations in the same order per output as k_pvalue / k_stripiness (p-value, Stripiness: bit-identical; the centre
// sum is a tolerance statistic in both and is reduced in another order here).  Stripes the wave form does not take --
// more than SW_MAXH rows, a block wider than SW_MAXW columns, row sums of more than 128 terms -- go to the block kernels
// through an index list; STP_SCORE=block sends everything there (tests compare the two).
#define SW_KR_SHORT 3      /* rows per lane of the two instances: stripes of up to 64 x 3 = 192 rows (95 % of the candidates; 28 waves per CU) ... */
#define SW_KR_TALL 4       /* ... and of up to 256 rows (the rest of what a 400-bin frame yields; 16 waves per CU) */
#define SW_MAXH (64 * SW_KR_TALL)
#define SW_MAXW 64
#define SW_WAVES 4

// stp_pw_leaf with its loads in batches of eight (the values of a batch are requested before the first is added: one memory
// round trip per batch instead of one per element); same additions in the same order.  An element comes in two steps --
// `load(k)`: the memory access alone, `val(k, raw)`: what is added -- so that the short leaves and the tails (up to seven
// elements behind wave-uniform tests; most blocks of a candidate stripe are 3 .. 10 columns wide) also have all their loads in
// flight together: as one functor each element's NaN test sat next to its load, inside the element's own branch, and the
// loads of a tail went out one round trip after the other (round 5, seen in the listing).
template <class L, class V>
__device__ __forceinline__ double sw_pw_leaf(L load, V val, int64_t o, int n)
{
    if (n < 8) {
        double v[7];
#pragma unroll
        for (int i = 0; i < 7; i++) v[i] = (i < n) ? load(o + i) : 0.0;
        double res = 0.;
#pragma unroll
        for (int i = 0; i < 7; i++)
            if (i < n) res += val(o + i, v[i]);
        return res;
    }
    double r[8];
#pragma unroll
    for (int t = 0; t < 8; t++) r[t] = load(o + t);
#pragma unroll
    for (int t = 0; t < 8; t++) r[t] = val(o + t, r[t]);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        double v[8];
#pragma unroll
        for (int t = 0; t < 8; t++) v[t] = load(o + i + t);
#pragma unroll
        for (int t = 0; t < 8; t++) r[t] += val(o + i + t, v[t]);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    {
        const int m = n - i;                                    // 0 .. 7 tail values, in order
        double v[7];
#pragma unroll
        for (int t = 0; t < 7; t++) v[t] = (t < m) ? load(o + i + t) : 0.0;
#pragma unroll
        for (int t = 0; t < 7; t++)
            if (t < m) res += val(o + i + t, v[t]);
    }
    return res;
}

__device__ __forceinline__ void sw_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// Order statistics k0 <= k1 of arr[0..n) (this wave's LDS array): *r0 / *r1 = a value v with #(< v) <= k < #(<= v), 0.0 when
// there is none -- block_median's rule; NaN entries are never counted and never chosen.  All lanes must call.
// Round 5: selection by pivoting on wave-uniform masks.  Every lane holds its (up to SW_KR) elements in registers; the pivot is
// the first element still in play (v_readlane, no LDS round trip), two ballots per chunk count the elements below and equal to
// it, and the set in play shrinks to one side: ~2 ln n rounds of ~20 instructions instead of testing the candidates one LDS
// read after the other (n / 2 .. n dependent reads; a fifth of the kernel's time in the round's ablation).  The value of an
// order statistic does not depend on how it is found.
template <int SW_KR>
__device__ __forceinline__ double sw_kth(const double* m, const unsigned long long* act, int k)
{
    unsigned long long a[SW_KR];
#pragma unroll
    for (int q = 0; q < SW_KR; q++) a[q] = act[q];
    for (int round = 0; round < 64 * SW_KR + 1; round++) {      // (every round removes at least the pivot: the bound is never reached)
        double v = 0.0;
        bool have = false;
#pragma unroll
        for (int q = 0; q < SW_KR; q++)
            if (!have && a[q] != 0ull) {                        // wave-uniform
                const int lp = __ffsll((long long)a[q]) - 1;
                v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m[q]), lp), __builtin_amdgcn_readlane(__double2loint(m[q]), lp));
                have = true;
            }
        if (!have) return 0.0;                                  // k beyond the elements that count
        unsigned long long L[SW_KR], E[SW_KR];
        int nl = 0, ne = 0;
#pragma unroll
        for (int q = 0; q < SW_KR; q++) {
            L[q] = __ballot(m[q] < v) & a[q];
            E[q] = __ballot(m[q] == v) & a[q];
            nl += __popcll(L[q]); ne += __popcll(E[q]);
        }
        if (k < nl) {
#pragma unroll
            for (int q = 0; q < SW_KR; q++) a[q] = L[q];
        } else if (k < nl + ne) {
            return v;
        } else {
            k -= nl + ne;
#pragma unroll
            for (int q = 0; q < SW_KR; q++) a[q] &= ~(L[q] | E[q]);
        }
    }
    return 0.0;
}
template <int SW_KR>
__device__ __forceinline__ void sw_select(const double* arr, int n, int k0, int k1, int lane, double* r0, double* r1)
{
    double m[SW_KR];
    unsigned long long act[SW_KR];
    const int nq = (n + 63) >> 6;
#pragma unroll
    for (int q = 0; q < SW_KR; q++) {
        const int j = lane + 64 * q;
        m[q] = (q < nq && j < n) ? arr[j] : NAN;
        act[q] = __ballot(m[q] == m[q]);                        // the elements that count: inside the array and not NaN
    }
#if defined(STP_ABLATE_SCORE) && STP_ABLATE_SCORE == 1      /* timing-only builds (tools: never shipped) */
    if (n > 0) { *r0 = arr[k0 < n ? k0 : 0]; *r1 = arr[k1 < n ? k1 : 0]; return; }
#endif
    const double a = sw_kth<SW_KR>(m, act, k0);
    *r0 = a;
    *r1 = (k1 == k0) ? a : sw_kth<SW_KR>(m, act, k1);
}
// numpy's pairwise sum of v[0..n), n <= 256: stp_pw's recursion pw(o, n) = pw(o, n2) + pw(o + n2, n - n2), n2 = n / 2 - (n / 2) % 8,
// ends in at most three leaves of <= 128 values here -- [0, nA), and [nA, n) either whole or split once more -- whose eight
// stride-8 partial sums run on lanes 0..7 / 8..15 / 16..23; `scr`: 24 doubles of this wave's LDS.  All lanes must call;
// the result is wave-uniform.
__device__ __forceinline__ double sw_pw_sum(const double* v, int n, int lane, double* scr)
{
    int off[3] = {0, 0, 0}, len[3] = {n, 0, 0};
    int nl = 1;
    if (n > 128) {
        int n2 = n / 2; n2 -= n2 % 8;
        len[0] = n2; off[1] = n2; len[1] = n - n2; nl = 2;
        if (len[1] > 128) {
            int m2 = len[1] / 2; m2 -= m2 % 8;
            off[2] = off[1] + m2; len[2] = len[1] - m2; len[1] = m2; nl = 3;
        }
    }
    const int lf = lane >> 3, t = lane & 7;
    if (lf < nl) {
        const int o = lf == 0 ? off[0] : (lf == 1 ? off[1] : off[2]), m = lf == 0 ? len[0] : (lf == 1 ? len[1] : len[2]);
        if (m >= 8) {
            double r = v[o + t];
            for (int i = 8; i < m - (m % 8); i += 8) r += v[o + i + t];
            scr[lf * 8 + t] = r;
        }
    }
    sw_sync();
    double res[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int l = 0; l < 3; l++) {
        if (l >= nl) break;
        const int o = off[l], m = len[l];
        double r;
        int i;
        if (m < 8) { r = 0.; i = 0; }
        else {
            const double* a = scr + l * 8;
            r = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
            i = m - (m % 8);
        }
        for (; i < m; i++) r += v[o + i];
        res[l] = r;
    }
    const double tot = nl == 1 ? res[0] : (nl == 2 ? res[0] + res[1] : res[0] + (res[1] + res[2]));
    sw_sync();
    return tot;
}

template <bool DO_PV, bool DO_SC, int SW_KR>
__global__ __launch_bounds__(64 * SW_WAVES) void k_score_wave(stp_bandref B, const int* __restrict__ idx, int nidx, int bs,
                                                              const double* __restrict__ srt, const int* __restrict__ nvalid, int ncolbg,
                                                              const stp_pv_stripe* __restrict__ pst, double* __restrict__ out_p,
                                                              const double* __restrict__ exval, const stp_score_stripe* __restrict__ sst,
                                                              double* __restrict__ out_g, double* __restrict__ out_mean,
                                                              double* __restrict__ out_total, int* __restrict__ out_status)
{
    __shared__ double exl[STP_NDIAG];                            // expected value + 1e-8 (getStripe.py:655-659)
    __shared__ double s_arr[SW_WAVES][3][64 * SW_KR];
    __shared__ double s_scr[SW_WAVES][24];
    __shared__ int16_t s_keepc[SW_WAVES][3][SW_MAXW];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    bool e_pos = false;                                          // every expected value + 1e-8 is finite and > 0 (the usual table):
    if (DO_SC) {                                                 // then observed / expected is NaN exactly where observed is
        bool ok = true;
        for (int i = tid; i < STP_NDIAG; i += 64 * SW_WAVES) {
            const double e = exval[i] + .00000001;
            exl[i] = e;
            ok = ok && e > 0.0 && e < INFINITY;
        }
        e_pos = __syncthreads_and(ok) != 0;
    }
    const int slot = blockIdx.x * SW_WAVES + wv;
    if (slot >= nidx) return;                                    // (no workgroup barrier below)
    const int si = __builtin_amdgcn_readfirstlane(idx ? idx[slot] : slot);
    double* const A = s_arr[wv][0];
    double* const Bv = s_arr[wv][1];
    double* const Cv = s_arr[wv][2];
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (DO_PV) {
        // ---- getStripe.pvalue (getStripe.py:552-605), as k_pvalue
        const stp_pv_stripe s = pst[si];
        const int h = s.row1 - s.row0, ncol = s.col1 - s.col0, nq = (h + 63) >> 6;
        const bool tr = B.sym && (s.row1 - 1 - s.col0 < B.hw);          // every mirrored pixel lies inside its band row (wave-uniform)
        int64_t lo[3], hi[3];
        stp_pyslice(bs, -bs, ncol, &lo[0], &hi[0]);
        stp_pyslice(0, bs, ncol, &lo[1], &hi[1]);
        stp_pyslice(-bs, ncol, ncol, &lo[2], &hi[2]);
        double dv[2 * SW_KR];                                    // centre - left, centre - right of row lane + 64 q
        int nv[2 * SW_KR];
        const double* sp[2 * SW_KR];
#pragma unroll
        for (int t = 0; t < 2 * SW_KR; t++) { dv[t] = 0.0; nv[t] = 0; sp[t] = srt; }
#pragma unroll
        for (int q = 0; q < SW_KR; q++) {
            if (q >= nq) break;                                  // wave-uniform
            const int j = lane + 64 * q;
            const bool on = j < h;
            const int jj = on ? j : 0;                           // idle lanes repeat row 0 (loads stay inside the rectangle)
            const int64_t gr = (int64_t)s.row0 + jj;
            // mat[jj][k] = M[gr][col0 + k]: read from band row gr (rowp[k]) or, in a symmetric band, as M[col0 + k][gr] from
            // band row col0 + k (colp[k * (W - 1)]: consecutive lanes = consecutive rows = consecutive addresses)
            const double* rowp = B.d + gr * (int64_t)B.W + ((int64_t)s.col0 - gr + B.hw);
            const double* colp = B.d + (int64_t)s.col0 * B.W + (gr - (int64_t)s.col0 + B.hw);
            const int64_t cstep = B.W - 1;
            double part[3];
#pragma unroll
            for (int p = 0; p < 3; p++) {
                const int n = (int)(hi[p] - lo[p]);
                auto nan0 = [](int64_t, double v) { return (v != v) ? 0.0 : v; };
                if (tr) part[p] = sw_pw_leaf([&](int64_t k) { return colp[k * cstep]; }, nan0, lo[p], n) / (double)n;
                else part[p] = sw_pw_leaf([&](int64_t k) { return rowp[k]; }, nan0, lo[p], n) / (double)n;
            }
            dv[2 * q] = part[0] - part[1]; dv[2 * q + 1] = part[0] - part[2];
            int d, tab;
            if (s.mode == 0) { d = jj; tab = 1; }
            else if (s.mode == 1) { d = s.upbase - jj - 1; tab = 0; }
            else { d = s.fixed_row; tab = s.fixed_tab; }
            if (s.mode != 2 && d >= 400) d = 399;
            if (d < 0) d += STP_NDIAG;
#if defined(STP_ABLATE_SCORE) && STP_ABLATE_SCORE == 5      /* timing-only: every lane searches ONE row (what coalesced table reads could gain at most) */
            d = 7;
#endif
            const size_t rl = (size_t)(tab ? 2 : 0) * STP_NDIAG + d, rr = (size_t)(tab ? 3 : 1) * STP_NDIAG + d;
            nv[2 * q] = nvalid[rl]; nv[2 * q + 1] = nvalid[rr];
            // (`srt` is k_bg_sort's row-fastest copy: element `mid` of row d of table t at srt[(t * ncolbg + mid) * STP_NDIAG + d])
            sp[2 * q] = srt + (size_t)(tab ? 2 : 0) * ncolbg * STP_NDIAG + d; sp[2 * q + 1] = srt + (size_t)(tab ? 3 : 1) * ncolbg * STP_NDIAG + d;
        }
        // the lower bounds of all (row, side) pairs of a lane in lock step: 2 nq dependent load chains in flight
        int slo[2 * SW_KR], shi[2 * SW_KR];
#pragma unroll
        for (int t = 0; t < 2 * SW_KR; t++) { slo[t] = 0; shi[t] = (t < 2 * nq) ? nv[t] : 0; }
        for (;;) {
#if defined(STP_ABLATE_SCORE) && STP_ABLATE_SCORE == 2
            break;
#endif
            bool any = false;
            double xv[2 * SW_KR];
            int mid[2 * SW_KR];
#pragma unroll
            for (int t = 0; t < 2 * SW_KR; t++) {
                const bool a = slo[t] < shi[t];
                any = any || a;
                mid[t] = (slo[t] + shi[t]) >> 1;
                xv[t] = a ? sp[t][(size_t)mid[t] * STP_NDIAG] : 0.0;
            }
            if (!any) break;
#pragma unroll
            for (int t = 0; t < 2 * SW_KR; t++)
                if (slo[t] < shi[t]) { if (xv[t] < dv[t]) slo[t] = mid[t] + 1; else shi[t] = mid[t]; }
        }
#pragma unroll
        for (int q = 0; q < SW_KR; q++) {
            if (q >= nq) break;
            const int j = lane + 64 * q;
            const int vL = nv[2 * q], vR = nv[2 * q + 1];
            const int cL = (dv[2 * q] != dv[2 * q]) ? 0 : vL - slo[2 * q], cR = (dv[2 * q + 1] != dv[2 * q + 1]) ? 0 : vR - slo[2 * q + 1];   // bg_count_ge
            const double p1 = (double)cL / (double)vL, p2 = (double)cR / (double)vR;
            double p = (p2 > p1) ? p2 : p1;
            if (p == 0.0) p = 1.0 / (double)ncolbg;
            if (j < h) A[j] = p;
        }
        sw_sync();
        double r0, r1;
        const int k0 = (h - 1) / 2, k1 = h / 2;
        sw_select<SW_KR>(A, h, k0, k1, lane, &r0, &r1);
        if (lane == 0) out_p[si] = (k0 == k1) ? r0 : (r0 + r1) / 2.0;
        sw_sync();                                               // A is reused below
    }
    if (DO_SC) {
        // ---- getStripe.scoringstripes.iterate_idx (getStripe.py:661-759), as k_stripiness
        const stp_score_stripe s = sst[si];
        const int h = s.row1 - s.row0, nq = (h + 63) >> 6;
        int16_t (*keepc)[SW_MAXW] = s_keepc[wv];
        // one pixel of block b: observed / (expected + 1e-8); a masked pixel is NaN (:701-707)
        const int cmin = min(s.col0[0], min(s.col0[1], s.col0[2]));
        const bool tr = B.sym && (s.row1 - 1 - cmin < B.hw);            // as above, for the three blocks
        // Pixel (r, c) of block b = obs_row(b, r)[c * cs]: the row's first pixel and ONE wave-uniform element stride (W - 1 through
        // the symmetry, 1 in the band's own rows).  (round 5: formed per element -- a 64-bit product of a column number the
        // compiler could not know to be wave-uniform and the tr / non-tr choice by selects -- the address took ~28 vector
        // instructions per pixel, three of them quarter-rate integer multiplies; now one scalar product and one 64-bit add.)
        const int64_t cs = tr ? (int64_t)B.W - 1 : 1;
        auto obs_row = [&](int b, int r) -> const double* {
            const int64_t gr = (int64_t)s.row0 + r, c0 = (int64_t)s.col0[b];
            return tr ? B.d + c0 * (int64_t)B.W + (gr - c0 + B.hw) : B.d + gr * (int64_t)B.W + (c0 - gr + B.hw);
        };
        auto exv = [&](int b, int r, int c) -> double {
            int ix = (s.ex0[b] + c) - (s.ey0 + r);
            if (ix < 0) ix = -ix;
            if (ix >= 400) ix = 399;
            return exl[ix];
        };
        // columns that hold a pixel that is not NaN (:709-711)
        unsigned long long alive[3];
        int wb[3];
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const int w = s.col1[b] - s.col0[b];
            wb[b] = w;
            unsigned long long m = 0ull;
#if defined(STP_ABLATE_SCORE) && STP_ABLATE_SCORE == 4
            m = w >= 64 ? ~0ull : ((1ull << w) - 1ull);
            for (int c0 = 0; c0 < 0; c0 += 4) {
#else
            for (int c0 = 0; c0 < w; c0 += 4) {                  // wave-uniform; four columns' pixels requested together
#endif                                                           // (eight per round trip measured slower: 0.189 against 0.165 ms per
                                                                 //  11 k stripes -- blocks are 3 .. 10 columns wide, the idle slots cost)
                unsigned found = 0u;
                for (int q = 0; q < nq && found != 15u; q++) {
                    const int r = lane + 64 * q;
                    const bool ron = r < h && !(r >= s.mrow0 && r <= s.mrow1);
                    const double* orow = obs_row(b, ron ? r : 0);
                    double o[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int c = c0 + u;
                        const bool con = ron && c < w && !(c >= s.mcol0[b] && c <= s.mcol1[b]);
                        o[u] = con ? orow[(int64_t)c * cs] : NAN;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        bool al = (o[u] == o[u]);                                    // o / e is NaN exactly when o is ...
                        if (!e_pos && al) {                                          // ... unless the table holds something else (workgroup-uniform)
                            const double v = o[u] / exv(b, r, c0 + u);               // (o is not NaN: the pixel is inside the block and unmasked)
                            al = (v == v);
                        }
                        if (__ballot(al) != 0ull) found |= 1u << u;
                    }
                }
                m |= (unsigned long long)found << c0;
            }
            alive[b] = m;
        }
        // the rows dead columns delete (:713-733)
        int status = 0;
        bool del[SW_KR];
#pragma unroll
        for (int q = 0; q < SW_KR; q++) del[q] = false;
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const int w = wb[b];
            unsigned long long dead = ~alive[b] & (w >= 64 ? ~0ull : ((1ull << w) - 1ull));
            while (dead) {                                       // wave-uniform, rare
                const int c = __ffsll((long long)dead) - 1;
                dead &= dead - 1ull;
                int rd = s.mirror ? (h - 1 - c) : c;
                if (rd < -h || rd >= h) status = 1;              // np.delete: index out of bounds
                if (rd < 0) rd += h;
                if (rd >= 0 && rd < h) {
#pragma unroll
                    for (int q = 0; q < SW_KR; q++) del[q] = del[q] || (lane + 64 * q == rd);
                }
            }
            if (lane < w && ((alive[b] >> lane) & 1ull)) keepc[b][__popcll(alive[b] & lt)] = (int16_t)lane;
        }
        int nkc[3];
#pragma unroll
        for (int b = 0; b < 3; b++) nkc[b] = __popcll(alive[b]);
        // kept rows in order
        int pos[SW_KR], hk = 0;
        bool keep[SW_KR];
#pragma unroll
        for (int q = 0; q < SW_KR; q++) {
            const int r = lane + 64 * q;
            keep[q] = q < nq && r < h && !del[q];
            const unsigned long long bal = __ballot(keep[q]);
            pos[q] = hk + __popcll(bal & lt);
            hk += __popcll(bal);
        }
        sw_sync();                                               // keepc is written
        // row means after nantozero (:739-745), numpy's pairwise order over the kept columns
        double loc = 0.0;
#pragma unroll
        for (int q = 0; q < SW_KR; q++) {
            if (q >= nq) break;
            if (keep[q]) {
                const int r = lane + 64 * q;
                const bool rmask = r >= s.mrow0 && r <= s.mrow1;
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    const int n = nkc[b];
                    const bool full = n == wb[b];                // no column deleted: kept column k is column k
                    const double* orow = obs_row(b, r);
                    // (the column of kept column k is the same for every lane: a scalar, so that its products are scalar too)
                    auto colof = [&](int64_t k) { return full ? (int)k : __builtin_amdgcn_readfirstlane((int)keepc[b][k]); };
                    const double sum = sw_pw_leaf([&](int64_t k) {                       // the observed pixel (0 under the mask: never read)
                        const int c = colof(k);
                        return (rmask || (c >= s.mcol0[b] && c <= s.mcol1[b])) ? 0.0 : orow[(int64_t)c * cs];
                    }, [&](int64_t k, double o) {
                        const int c = colof(k);
                        if (rmask || (c >= s.mcol0[b] && c <= s.mcol1[b])) return 0.0;
#if defined(STP_ABLATE_SCORE) && STP_ABLATE_SCORE == 3
                        const double v = o * exv(b, r, c);
#else
                        const double v = o / exv(b, r, c);
#endif
                        return (v != v) ? 0.0 : v;
                    }, 0, n);
                    const double mean = sum / (double)n;
                    s_arr[wv][b][pos[q]] = mean;
                    if (b == 0) loc += mean * (double)n;         // centerTotal (:747), a tolerance statistic
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) loc += __shfl_xor(loc, o);
        const double s_tot = loc;
        sw_sync();
        // Sobel-like scores (:750-756, stats.py:184-199) of rows 1 .. hk-2
        const int nd = hk - 2 > 0 ? hk - 2 : 0;
        double dq[SW_KR];
#pragma unroll
        for (int q = 0; q < SW_KR; q++) {
            const int i = 1 + lane + 64 * q;
            dq[q] = NAN;
            if (q < nq && i < hk - 1) {
                const double* cm = A; const double* lm = Bv; const double* rm = Cv;
                double gxl = 0.0, gxr = 0.0, gy = 0.0;
                gxl += (-1.0 * lm[i - 1] + -2.0 * lm[i]) + -1.0 * lm[i + 1];
                gxl += (1.0 * cm[i - 1] + 2.0 * cm[i]) + 1.0 * cm[i + 1];
                gxr += (1.0 * cm[i - 1] + 2.0 * cm[i]) + 1.0 * cm[i + 1];
                gxr += (-1.0 * rm[i - 1] + -2.0 * rm[i]) + -1.0 * rm[i + 1];
                gy += (1.0 * lm[i - 1] + 0.0 * lm[i]) + -1.0 * lm[i + 1];
                gy += (2.0 * cm[i - 1] + 0.0 * cm[i]) + -2.0 * cm[i + 1];
                gy += (1.0 * rm[i - 1] + 0.0 * rm[i]) + -1.0 * rm[i + 1];
                if (gy < 0) gy *= -1;
                double gx = (gxl < gxr || gxl != gxl) ? gxl : gxr;           // np.minimum (NaN propagates)
                if (gxr != gxr) gx = gxr;
                dq[q] = gx - gy;
            }
        }
        sw_sync();                                               // everybody has read the flank means: Bv takes the differences
        // diff = [x for x in diff if x >= 0 or x < 0], in order (:757)
        int ndk = 0;
#pragma unroll
        for (int q = 0; q < SW_KR; q++) {
            const int t = lane + 64 * q;
            const bool kp = q < nq && t < nd && dq[q] == dq[q];
            const unsigned long long bal = __ballot(kp);
            if (kp) Bv[ndk + __popcll(bal & lt)] = dq[q];
            ndk += __popcll(bal);
        }
        // np.nanmedian(centerm): the order statistics of the values that are not NaN
        int nm = 0;
#pragma unroll
        for (int q = 0; q < SW_KR; q++) {
            const int j = lane + 64 * q;
            const double v = (q < nq && j < hk) ? A[j] : NAN;
            nm += __popcll(__ballot(v == v));
        }
        sw_sync();
        double med = NAN;
        if (nm > 0) {
            double r0, r1;
            const int k0 = (nm - 1) / 2, k1 = nm / 2;
            sw_select<SW_KR>(A, hk, k0, k1, lane, &r0, &r1);
            med = (k0 == k1) ? r0 : (r0 + r1) / 2.0;
        }
        const double sum = sw_pw_sum(Bv, ndk, lane, s_scr[wv]);
        if (lane == 0) {
            const double avg = sum / (double)ndk;
            out_g[si] = med * avg;
            const int nc = nkc[0];
            out_total[si] = (nc == 0) ? 0.0 : s_tot * (double)nc;
            out_mean[si] = (nc == 0) ? NAN : (s_tot * (double)nc) / ((double)hk * nc * nc);
            if (out_status) out_status[si] = status;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// seeimage.py:78-85 -- the green = blue plane of the heat map of one window, with StripeSearch's image arithmetic
// (stp_gplane_px, getStripe.py:889-895): clip((255 * (M - A) / M) / 255, 0, 1); a NaN pixel stays NaN (the
// reference's np.where / np.clip leave it alone; imshow draws it blank).  One lane per pixel, rows coalesced.
__global__ __launch_bounds__(256) void k_window_plane(stp_bandref B, int64_t row0, int nrows, int64_t col0, int ncols, double M,
                                                       double* __restrict__ out)
{
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < (int64_t)nrows * ncols; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / ncols, c = i - r * ncols;
        out[i] = stp_gplane_px(band_at(B, row0 + r, col0 + c), M);
    }
}

// ---------------------------------------------------------------------------------------------
// getStripe.getMean (getStripe.py:518-521): nanmean / nansum of the observed stripe pixels

__global__ __launch_bounds__(256) void k_stripe_mean(stp_bandref B, const stp_rect* __restrict__ rc, double* __restrict__ out_mean,
                                                      double* __restrict__ out_sum)
{
    __shared__ double ssum[256];
    __shared__ long long scnt[256];
    const stp_rect r = rc[blockIdx.x];
    const int h = r.row1 - r.row0, w = r.col1 - r.col0;
    double s = 0.0;
    long long c = 0;
    for (int64_t i = threadIdx.x; i < (int64_t)h * w; i += blockDim.x) {
        double v = band_at(B, r.row0 + i / w, r.col0 + i % w);
        if (v == v) { s += v; c++; }
    }
    ssum[threadIdx.x] = s; scnt[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { ssum[threadIdx.x] += ssum[threadIdx.x + o]; scnt[threadIdx.x] += scnt[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out_sum[blockIdx.x] = ssum[0]; out_mean[blockIdx.x] = ssum[0] / (double)scnt[0]; }
}

// ---------------------------------------------------------------------------------------------
// Frame preparation, one workgroup per frame, ONE kernel:
//   * zero-column removal (getStripe.py:809-821).  The reference keeps column c when sum(axis 0) != 0 after NaN -> 0.
//     Balanced contact values are >= 0, and for non-negative columns "sum != 0" is "some entry > 0", whatever the order
//     of the additions.  So the rows are read the way HBM likes them (round 4) -- one wave per row, the whole row of a
//     lane's seven 64-column segments requested before the first use, two rows in flight per wave, sixteen waves -- and a
//     column's occupancy is the OR of the rows' ballots.  A frame that holds a NEGATIVE value (no balanced map does; the
//     ABI does not forbid it) is flagged by the same walk and takes the exact column walk instead: one lane per column,
//     rows in numpy's axis-0 order.
//   * StripeSearch's medpixel = np.quantile(submat[submat > 0], 0.5) (getStripe.py:885): exact order statistics over
//     the frame's positive pixels (positive doubles order like their bit patterns).  The SAME walk counts them, tracks
//     the min / max key and fills an 8192-bin histogram of range-normalised digits; the range is estimated beforehand
//     from eight sampled rows (2 % of the frame) -- it only has to spread the pixels over the bins (no same-address
//     LDS-atomic pile-up), exactness never depends on it: keys outside the estimate fall into the two end bins.
//     A second pass gathers the keys of the bin that holds the rank (continuous data: a few dozen) into LDS, tracks the
//     smallest key of the higher bins, and ranks are counted in LDS.
//     med[f*3 + {0,1,2}] = a[k0], a[k1], N with k0 = (N-1)/2, k1 = N/2 (numpy's lerp is done on the host).
//   An exact select needs the second pass (the candidates of the median's bin are unknown until the histogram is
//   complete), so the frame is read twice.  Heavily duplicated data (integer counts) or a selected bin with more than
//   STP_MED_CAP keys falls back to the general radix select: 13 bits of (key - min) per further pass, then a <= /
//   successor pass.
// Footprint: 1024 threads and 44 KB of LDS (8192 bins, 1024 gathered keys).  These launches run on the auxiliary stream while
// the main stream is busy with the chain; measured on one box (profiles/r04_ab_prep.txt): a SMALL workgroup (512 threads,
// 9.5 KB, 1024 bins) finds room beside k_canny_f32's five workgroups per CU and runs at once -- its own interval is shorter,
// 13.6 instead of 23.9 ms of the step -- but it takes issue slots and LDS bandwidth from the chain (canny 46.1 instead of
// 44.8 ms), and the step is what counts: 80.5 against 79.0 ms.  The large workgroup waits for a CU to drain and then owns it.
#ifndef STP_MED_LOG
#define STP_MED_LOG 13
#endif
#define STP_MED_BINS (1 << STP_MED_LOG)
#ifndef STP_MED_CAP
#define STP_MED_CAP 1024
#endif
#ifndef STP_PREP_NT
#define STP_PREP_NT 1024
#endif
#define STP_PREP_SEG ((STP_FRAME_MAX + 63) / 64)     /* 64-column segments of a frame row: 7 */
__global__ __launch_bounds__(STP_PREP_NT) void k_frame_prep(stp_bandref B, const int32_t* __restrict__ fstart,
                                                            const int32_t* __restrict__ fn0, int32_t* __restrict__ S_out,
                                                            int16_t* __restrict__ nz_out, double* __restrict__ med, int keep_all)
{
    constexpr int NT = STP_PREP_NT, NWV = NT / 64, BPT = STP_MED_BINS / NT, NSEG = STP_PREP_SEG;
    __shared__ unsigned int hist[STP_MED_BINS];
    __shared__ unsigned int part[NT];
    __shared__ unsigned long long cand[STP_MED_CAP];
    __shared__ unsigned long long s_hi, s_key[2], s_n, s_mn, s_mx, s_occ[NSEG];
    __shared__ unsigned int s_k, s_le, s_cnt, s_m, s_bin, s_neg;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t st = fstart[f];
    const int n0 = fn0[f];
    for (int i = tid; i < STP_MED_BINS; i += NT) hist[i] = 0;
    if (tid == 0) { s_le = 0; s_hi = ~0ull; s_m = 0; s_key[0] = s_key[1] = ~0ull; s_n = 0; s_mn = ~0ull; s_mx = 0; s_neg = 0; }
    if (tid < NSEG) s_occ[tid] = 0ull;
    __syncthreads();
    const double* p = B.d + st * (int64_t)B.W + (tid + B.hw);          // column tid of row 0 of the frame
    // ---- range estimate from eight sampled rows
    {
        unsigned long long mn = ~0ull, mx = 0;
        if (tid < n0) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = p[(int64_t)(((2 * q + 1) * n0) >> 4) * (B.W - 1)];
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (v[q] > 0.0) {
                    const unsigned long long key = (unsigned long long)__double_as_longlong(v[q]);
                    mn = key < mn ? key : mn; mx = key > mx ? key : mx;
                }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long a = __shfl_xor(mn, o), b = __shfl_xor(mx, o);
            mn = a < mn ? a : mn; mx = b > mx ? b : mx;
        }
        if (lane == 0 && mx) { atomicMin(&s_mn, mn); atomicMax(&s_mx, mx); }
    }
    __syncthreads();
    const unsigned long long smin = s_mx ? s_mn : 0ull;
    int shift1;
    {
        const unsigned long long range = s_mx ? (s_mx - s_mn) : 0ull;
        const int hb = 64 - __clzll((long long)(range | 1ull));          // bits of the sampled range (>= 1)
        shift1 = hb > STP_MED_LOG ? hb - STP_MED_LOG : 0;
    }
    auto digit1 = [&](unsigned long long key) -> unsigned int {          // monotone in key: bin order = key order
        if (key <= smin) return 0u;
        const unsigned long long d = (key - smin) >> shift1;
        return d < (unsigned long long)(STP_MED_BINS - 1) ? (unsigned int)d : (unsigned int)(STP_MED_BINS - 1);
    };
    __syncthreads();                                                      // everyone has read the estimate
    if (tid == 0) { s_mn = ~0ull; s_mx = 0; }
    // one wave walks rows wave, wave + NWV, ...: two rows -- 2 x NSEG coalesced 512-byte loads per lane -- in flight
    auto scan2 = [&](auto&& fn) {
        for (int r = wave; r < n0; r += 2 * NWV) {
            const double* q0 = B.d + (st + r) * (int64_t)B.W + (B.hw - r);
            const bool two = r + NWV < n0;                                // wave-uniform
            const double* q1 = q0 + (two ? (int64_t)NWV * (B.W - 1) : 0);
            double v[2 * NSEG];
#pragma unroll
            for (int q = 0; q < NSEG; q++) { const int c = lane + 64 * q; v[q] = c < n0 ? q0[c] : 0.0; }
#pragma unroll
            for (int q = 0; q < NSEG; q++) { const int c = lane + 64 * q; v[NSEG + q] = (two && c < n0) ? q1[c] : 0.0; }
#pragma unroll
            for (int q = 0; q < 2 * NSEG; q++) fn(v[q], q % NSEG);
        }
    };
    // ---- pass 1: column occupancy + count / min / max + histogram
    unsigned long long cnt = 0, mn = ~0ull, mx = 0;
    {
        unsigned long long occ[NSEG];
        bool neg = false;
#pragma unroll
        for (int q = 0; q < NSEG; q++) occ[q] = 0ull;
        scan2([&](double x, int q) {
            const bool pos = x > 0.0;
            occ[q] |= __ballot(pos);
            neg = neg || x < 0.0;
            if (pos) {
                const unsigned long long key = (unsigned long long)__double_as_longlong(x);
                cnt++; mn = key < mn ? key : mn; mx = key > mx ? key : mx;
                atomicAdd(&hist[digit1(key)], 1u);
            }
        });
        if (lane < NSEG) {
            unsigned long long o = 0ull;
#pragma unroll
            for (int q = 0; q < NSEG; q++) o = (lane == q) ? occ[q] : o;
            if (o) atomicOr(&s_occ[lane], o);
        }
        if (__ballot(neg) && lane == 0) s_neg = 1u;
    }
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o);
        const unsigned long long a = __shfl_xor(mn, o), b = __shfl_xor(mx, o);
        mn = a < mn ? a : mn; mx = b > mx ? b : mx;
    }
    __syncthreads();                                                      // (also orders the reset of s_mn / s_mx)
    if (lane == 0 && cnt) { atomicAdd(&s_n, cnt); atomicMin(&s_mn, mn); atomicMax(&s_mx, mx); }
    if (s_neg) {
        // ---- a negative value: the reference's own test, sum over the rows in order != 0, one lane per column
        __syncthreads();
        if (tid < NSEG) s_occ[tid] = 0ull;
        __syncthreads();
        double sum = 0.0;
        if (tid < n0) {
            for (int r0 = 0; r0 < n0; r0 += 8) {
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = (r0 + q < n0) ? p[(int64_t)(r0 + q) * (B.W - 1)] : 0.0;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    double x = v[q];
                    if (x != x) x = 0.0;
                    if (r0 + q < n0) sum += x;
                }
            }
        }
        const unsigned long long bal = __ballot(tid < n0 && sum != 0.0);
        if (lane == 0 && wave < NSEG) s_occ[wave] = bal;
    }
    __syncthreads();
    {   // nz = the kept columns in order, S = their number (getStripe.py:812-821)
        int base = 0, total = 0;
        unsigned long long mine = 0ull;
#pragma unroll
        for (int i = 0; i < NSEG; i++) {
            unsigned long long o = s_occ[i];
            if (keep_all) o = ~0ull;
            const int hi = n0 - 64 * i;                                   // columns of this segment inside the frame
            o = hi >= 64 ? o : (hi > 0 ? (o & ((1ull << hi) - 1ull)) : 0ull);
            if (i < wave) base += __popcll(o);
            if (i == wave) mine = o;
            total += __popcll(o);
        }
        if (wave < NSEG && ((mine >> lane) & 1ull))
            nz_out[f * STP_FRAME_MAX + base + __popcll(mine & ((1ull << lane) - 1ull))] = (int16_t)tid;
        if (tid == 0) S_out[f] = (keep_all || total > 10) ? total : 0;   // getStripe.py:818
    }
    __syncthreads();
    const unsigned int N = (unsigned int)s_n;
    if (N == 0) {
        if (tid == 0) { med[f * 3] = NAN; med[f * 3 + 1] = NAN; med[f * 3 + 2] = 0.0; }
        return;
    }
    const unsigned long long kmin = s_mn, kmax = s_mx;
    const unsigned int k0 = (N - 1) / 2, k1 = N / 2;
    auto scan = [&](auto&& fn) {                // the positive keys of the frame (two rows in flight per wave: scan2)
        scan2([&](double x, int) { if (x > 0.0) fn((unsigned long long)__double_as_longlong(x)); });
    };
    // bin of rank k in the current histogram: BPT bins per lane -> partial sums -> inclusive scan -> locate
    auto find_bin = [&](unsigned int k) {
        unsigned int mine = 0;
        for (int b = 0; b < BPT; b++) mine += hist[tid * BPT + b];
        part[tid] = mine;
        __syncthreads();
        for (int o = 1; o < NT; o <<= 1) {
            unsigned int v = (tid >= o) ? part[tid - o] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        const unsigned int incl = part[tid], excl = incl - mine;
        if (excl <= k && k < incl) {              // exactly one lane
            unsigned int acc = excl;
            int d = tid * BPT;
            for (;; d++) { if (acc + hist[d] > k) break; acc += hist[d]; }
            s_k = k - acc; s_cnt = hist[d]; s_bin = (unsigned int)d;
        }
        __syncthreads();
    };
    // ranks k, k + (k1 - k0) among the m gathered keys of the selected bin; s_hi = smallest key of any higher bin
    auto finish_gathered = [&](unsigned int k) {
        const unsigned int m = s_m < STP_MED_CAP ? s_m : STP_MED_CAP;
        for (unsigned int t = tid; t < m; t += NT) {
            const unsigned long long x = cand[t];
            unsigned int lo = 0, le = 0;
            for (unsigned int j = 0; j < m; j++) { const unsigned long long y = cand[j]; lo += (y < x); le += (y <= x); }
            const unsigned int ka = k, kb = k + (k1 - k0);
            if (lo <= ka && ka < le) s_key[0] = x;          // equal keys write the same value
            if (lo <= kb && kb < le) s_key[1] = x;
        }
        __syncthreads();
        if (tid == 0) {
            med[f * 3] = __longlong_as_double((long long)s_key[0]);
            med[f * 3 + 1] = __longlong_as_double((long long)((k + (k1 - k0) < m) ? s_key[1] : s_hi));
            med[f * 3 + 2] = (double)N;
        }
    };
    find_bin(k0);
    if (s_cnt <= STP_MED_CAP) {
        // ---- pass 2 (common case): gather the selected bin
        const unsigned int sel = s_bin, k = s_k;
        unsigned long long hi_mn = ~0ull;
        scan([&](unsigned long long key) {
            const unsigned int b = digit1(key);
            if (b == sel) { const unsigned int slot = atomicAdd(&s_m, 1u); if (slot < STP_MED_CAP) cand[slot] = key; }
            else if (b > sel && key < hi_mn) hi_mn = key;
        });
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long a = __shfl_xor(hi_mn, o); hi_mn = a < hi_mn ? a : hi_mn; }
        if (lane == 0 && hi_mn != ~0ull) atomicMin(&s_hi, hi_mn);
        __syncthreads();
        finish_gathered(k);
        return;
    }
    // ---- general radix select on (key - kmin), STP_MED_LOG bits per pass (heavily duplicated data)
    unsigned int k = k0;
    unsigned long long prefix = 0;              // in the (key - kmin) domain
    const unsigned long long range = kmax - kmin;
    const int hb = range ? 64 - __clzll((long long)range) : 0;
    int width = hb < STP_MED_LOG ? hb : STP_MED_LOG, shift = hb - width;
    bool gathered = false;
    while (width > 0) {
        const int hs = shift + width;           // hs <= 63 here except possibly the very first pass
        __syncthreads();
        for (int i = tid; i < STP_MED_BINS; i += NT) hist[i] = 0;
        __syncthreads();
        const unsigned long long pfx_hi = hs >= 64 ? 0ull : (prefix >> hs);
        const unsigned int dmask = (1u << width) - 1u;
        scan([&](unsigned long long key) {
            const unsigned long long rel = key - kmin;
            const unsigned long long hi = hs >= 64 ? 0ull : (rel >> hs);
            if (hi == pfx_hi) atomicAdd(&hist[(unsigned int)(rel >> shift) & dmask], 1u);
        });
        __syncthreads();
        find_bin(k);
        k = s_k; prefix |= ((unsigned long long)s_bin << shift);
        const unsigned int cntb = s_cnt;
        if (shift == 0) break;
        if (cntb <= STP_MED_CAP) { gathered = true; break; }
        width = shift < STP_MED_LOG ? shift : STP_MED_LOG;
        shift -= width;
    }
    if (gathered) {
        const unsigned long long sel = prefix >> shift;
        unsigned long long hi_mn = ~0ull;
        scan([&](unsigned long long key) {
            const unsigned long long b = (key - kmin) >> shift;
            if (b == sel) { const unsigned int slot = atomicAdd(&s_m, 1u); if (slot < STP_MED_CAP) cand[slot] = key; }
            else if (b > sel && key < hi_mn) hi_mn = key;
        });
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long a = __shfl_xor(hi_mn, o); hi_mn = a < hi_mn ? a : hi_mn; }
        if (lane == 0 && hi_mn != ~0ull) atomicMin(&s_hi, hi_mn);
        __syncthreads();
        finish_gathered(k);
        return;
    }
    const unsigned long long key0 = kmin + prefix;
    unsigned int le = 0;
    unsigned long long mnk = ~0ull;
    scan([&](unsigned long long key) {
        if (key <= key0) le++;
        else if (key < mnk) mnk = key;
    });
    atomicAdd(&s_le, le);
    atomicMin(&s_hi, mnk);
    __syncthreads();
    if (tid == 0) {
        const unsigned long long key1 = (s_le > k1) ? key0 : s_hi;
        med[f * 3] = __longlong_as_double((long long)key0);
        med[f * 3 + 1] = __longlong_as_double((long long)key1);
        med[f * 3 + 2] = (double)N;
    }
}
