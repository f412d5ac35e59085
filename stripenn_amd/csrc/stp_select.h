// stp_select.h -- exact order statistics of the positive values of a (chunked) float64 array:
// the device part of getStripe.getQuantile_original (getStripe.py:160-176, `np.quantile(mat[mat>0], q)`).
// Positive doubles order like their bit patterns, so a rank is found by a 5-pass radix select with
// 13-bit digits over all chunks (grid-stride, LDS-privatised 8192-bin histograms merged with atomics),
// with the running (prefix, rank) state kept on the device between passes.
#pragma once
#include "stp_phases.h"

#define STP_SEL_BINS 8192

struct stp_sel_state {
    unsigned long long prefix;
    unsigned long long k;
    unsigned long long hist[STP_SEL_BINS];
};

__global__ __launch_bounds__(256) void k_sel_count(const double* __restrict__ v, long long n, unsigned long long* __restrict__ cnt)
{
    unsigned long long loc = 0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) loc += (v[i] > 0.0);
    for (int o = 32; o > 0; o >>= 1) loc += __shfl_xor(loc, o);
    if ((threadIdx.x & 63) == 0 && loc) atomicAdd(cnt, loc);
}

// balanced value of every stored pixel, twice for off-diagonal ones (slot 2p+1 stays 0 = ignored on the diagonal)
template <typename CT>     // pixels/count as stored: int32, or float64 (coolers written with --count-as-float, merged / scaled ones)
__global__ __launch_bounds__(256) void k_sel_pixel_values(const int64_t* __restrict__ b1, const int64_t* __restrict__ b2,
                                                           const CT* __restrict__ cnt, long long n,
                                                           const double* __restrict__ w /* bias of bins [lo, lo + nbins) */,
                                                           long long nbins, double* __restrict__ out, long long lo)
{
    for (long long p = blockIdx.x * 256ll + threadIdx.x; p < n; p += (long long)gridDim.x * 256) {
        const long long i = b1[p] - lo, j = b2[p] - lo;
        double v = (double)cnt[p];
        if (w) v = (i >= 0 && j >= 0 && i < nbins && j < nbins) ? v * (w[i] * w[j]) : 0.0;
        out[2 * p] = v;
        out[2 * p + 1] = (i != j) ? v : 0.0;
    }
}

__global__ __launch_bounds__(256) void k_sel_hist(const double* __restrict__ v, long long n, int shift, int width, int pass,
                                                   stp_sel_state* __restrict__ st)
{
    __shared__ unsigned int h[STP_SEL_BINS];
    for (int i = threadIdx.x; i < STP_SEL_BINS; i += 256) h[i] = 0;
    __syncthreads();
    const unsigned long long prefix = st->prefix;
    const int hs = shift + width;
    const unsigned int mask = (1u << width) - 1u;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double x = v[i];
        if (!(x > 0.0)) continue;
        const unsigned long long key = (unsigned long long)__double_as_longlong(x);
        if (pass == 0 || (key >> hs) == (prefix >> hs)) atomicAdd(&h[(unsigned int)(key >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < STP_SEL_BINS; i += 256)
        if (h[i]) atomicAdd(&st->hist[i], (unsigned long long)h[i]);
}

// one workgroup: locate the bin holding rank k, descend, clear the histogram
__global__ __launch_bounds__(1024) void k_sel_pick(stp_sel_state* __restrict__ st, int shift)
{
    __shared__ unsigned long long part[1024];
    const int tid = threadIdx.x;
    unsigned long long mine = 0;
    for (int b = 0; b < 8; b++) mine += st->hist[tid * 8 + b];
    part[tid] = mine;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        unsigned long long x = (tid >= o) ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += x;
        __syncthreads();
    }
    const unsigned long long k = st->k, incl = part[tid], excl = incl - mine;
    __syncthreads();
    if (excl <= k && k < incl) {
        unsigned long long acc = excl;
        int d = tid * 8;
        for (;; d++) { const unsigned long long c = st->hist[d]; if (acc + c > k) break; acc += c; }
        st->k = k - acc;
        st->prefix |= (unsigned long long)d << shift;
    }
    __syncthreads();
    for (int b = 0; b < 8; b++) st->hist[tid * 8 + b] = 0;
}

// ---------------------------------------------------------------------------------------------
// getStripe.RemoveRedundant (getStripe.py:1116-1196): one lane per row, pairs = rows of its own frame
// bucket that come later in the table plus every row of the next frame's bucket.
__global__ __launch_bounds__(256) void k_remove_redundant(long long n, const long long* __restrict__ p1,
                                                          const long long* __restrict__ p2, const long long* __restrict__ p3,
                                                          const long long* __restrict__ p4, const int* __restrict__ h,
                                                          const int* __restrict__ w, const double* __restrict__ key, int by,
                                                          const int* __restrict__ order, const int* __restrict__ b0,
                                                          const int* __restrict__ b1, const int* __restrict__ b2,
                                                          unsigned int* __restrict__ drop)
{
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= n) return;
    const long long ax0 = p1[i], ax1 = p2[i], ay0 = p3[i], ay1 = p4[i];
    for (int q = b0[i]; q < b2[i]; q++) {
        const int j = order[q];
        if (q < b1[i] && j <= i) continue;            // own bucket: each unordered pair once
        const long long bx0 = p1[j], bx1 = p2[j], by0 = p3[j], by1 = p4[j];
        long long ox = (ax1 < bx1 ? ax1 : bx1) - (ax0 > bx0 ? ax0 : bx0) + 1;
        long long oy = (ay1 < by1 ? ay1 : by1) - (ay0 > by0 ? ay0 : by0) + 1;
        if (ox < 0) ox = 0;
        if (oy < 0) oy = 0;
        const long long dx = (ax1 - ax0 < bx1 - bx0) ? ax1 - ax0 : bx1 - bx0;
        const long long dy = (ay1 - ay0 < by1 - by0) ? ay1 - ay0 : by1 - by0;
        const double sx = (double)ox / (double)dx, sy = (double)oy / (double)dy;   // python int / int
        if (sx > 0.2 && sy > 0.2) {
            const long long a = (i < j) ? i : j, b = (i < j) ? j : i;              // A comes first in the table
            bool dropA;
            if (by == 0) dropA = ((double)h[a] / (double)w[a]) <= ((double)h[b] / (double)w[b]);
            else if (by == 1) dropA = key[a] <= key[b];
            else dropA = key[a] > key[b];
            drop[dropA ? a : b] = 1u;
        }
    }
}
