// stp_select.h -- exact order statistics of the positive values of a (chunked) float64 array:
// the device part of getStripe.getQuantile_original (getStripe.py:160-176, `np.quantile(mat[mat>0], q)`).
// Positive doubles order like their bit patterns, so the ranks are found by a radix select with 11-bit digits
// (6 passes over all chunks, grid-stride, LDS-privatised histograms merged with atomics).  ALL ranks of a call
// descend together: ranks that still share a prefix share one histogram ("group"), a pass makes ONE sweep over the data
// for all groups, and the (prefix, rank, group) state stays on the device between passes -- one host round trip per
// call.  The first pass is the same for every rank; it also yields the number of positive values, so
// stp_select_count's sweep is that pass and stp_select_ranks adds five more.
#pragma once
#include "stp_phases.h"

#define STP_SEL_BITS 11
#define STP_SEL_BINS (1 << STP_SEL_BITS)
#define STP_SEL_MAXR 16                     /* ranks per call (numpy's linear interpolation asks for 2 per quantile) */
#define STP_SEL_PASSES 6
__device__ __constant__ const int stp_sel_shift[STP_SEL_PASSES] = {53, 42, 31, 20, 9, 0};
__device__ __constant__ const int stp_sel_width[STP_SEL_PASSES] = {11, 11, 11, 11, 11, 9};

struct stp_sel_state {
    unsigned long long prefix[STP_SEL_MAXR];      // per rank: the key bits fixed so far
    unsigned long long k[STP_SEL_MAXR];           // per rank: rank among the keys that share the prefix
    int group[STP_SEL_MAXR];                      // per rank: its histogram in the current pass
    int nranks, ngroups;
    unsigned long long gprefix[STP_SEL_MAXR];     // per group: the shared prefix
    unsigned long long hist0[STP_SEL_BINS];       // first pass (all positive values), kept: it serves every later call
    unsigned long long count;                     // number of positive values
    unsigned long long hist[STP_SEL_MAXR][STP_SEL_BINS];
};

// balanced value of every stored pixel, twice for off-diagonal ones (slot 2p+1 stays 0 = ignored on the diagonal)
template <typename CT, typename I2>     // pixels/count as stored: int32, or float64 (coolers written with --count-as-float, merged / scaled ones); bin2_id: int64 or int32
__global__ __launch_bounds__(256) void k_sel_pixel_values(const int64_t* __restrict__ b1, const I2* __restrict__ b2,
                                                           const CT* __restrict__ cnt, long long n,
                                                           const double* __restrict__ w /* bias of bins [lo, lo + nbins) */,
                                                           long long nbins, double* __restrict__ out, long long lo)
{
    for (long long p = blockIdx.x * 256ll + threadIdx.x; p < n; p += (long long)gridDim.x * 256) {
        const long long i = b1[p] - lo, j = (long long)b2[p] - lo;
        double v = (double)cnt[p];
        if (w) v = (i >= 0 && j >= 0 && i < nbins && j < nbins) ? v * (w[i] * w[j]) : 0.0;
        out[2 * p] = v;
        out[2 * p + 1] = (i != j) ? v : 0.0;
    }
}

// first pass: histogram of the top digit of every positive value
__global__ __launch_bounds__(256) void k_sel_hist0(const double* __restrict__ v, long long n, stp_sel_state* __restrict__ st)
{
    __shared__ unsigned int h[STP_SEL_BINS];
    for (int i = threadIdx.x; i < STP_SEL_BINS; i += 256) h[i] = 0;
    __syncthreads();
    const long long stride = (long long)gridDim.x * 256;
    for (long long i0 = blockIdx.x * 256ll + threadIdx.x; i0 < n; i0 += 4 * stride) {
        double x[4];
#pragma unroll
        for (int q = 0; q < 4; q++) x[q] = (i0 + q * stride < n) ? v[i0 + q * stride] : 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (x[q] > 0.0) atomicAdd(&h[(unsigned int)((unsigned long long)__double_as_longlong(x[q]) >> 53)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < STP_SEL_BINS; i += 256)
        if (h[i]) atomicAdd(&st->hist0[i], (unsigned long long)h[i]);
}
__global__ __launch_bounds__(1024) void k_sel_count(stp_sel_state* __restrict__ st)
{
    __shared__ unsigned long long part[1024];
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < STP_SEL_BINS; i += 1024) s += st->hist0[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) st->count = part[0];
}

// later passes: one histogram per group of ranks that share a prefix; dynamic LDS = ngroups_max * STP_SEL_BINS * 4
__global__ __launch_bounds__(512) void k_sel_hist(const double* __restrict__ v, long long n, int pass, int maxg,
                                                   stp_sel_state* __restrict__ st)
{
    extern __shared__ unsigned int hsel[];
    __shared__ unsigned long long s_gp[STP_SEL_MAXR];
    const int ng = st->ngroups < maxg ? st->ngroups : maxg;
    for (int i = threadIdx.x; i < ng * STP_SEL_BINS; i += 512) hsel[i] = 0;
    if ((int)threadIdx.x < ng) s_gp[threadIdx.x] = st->gprefix[threadIdx.x];
    __syncthreads();
    const int shift = stp_sel_shift[pass], hs = shift + stp_sel_width[pass];
    const unsigned int mask = (1u << stp_sel_width[pass]) - 1u;
    const long long stride = (long long)gridDim.x * 512;
    for (long long i0 = blockIdx.x * 512ll + threadIdx.x; i0 < n; i0 += 4 * stride) {
        double x[4];                                       // four independent loads in flight per lane
#pragma unroll
        for (int q = 0; q < 4; q++) x[q] = (i0 + q * stride < n) ? v[i0 + q * stride] : 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (!(x[q] > 0.0)) continue;
            const unsigned long long key = (unsigned long long)__double_as_longlong(x[q]);
            const unsigned long long top = key >> hs;
            for (int g = 0; g < ng; g++)                   // the groups' prefixes are distinct: at most one matches
                if (top == (s_gp[g] >> hs)) { atomicAdd(&hsel[g * STP_SEL_BINS + ((unsigned int)(key >> shift) & mask)], 1u); break; }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ng * STP_SEL_BINS; i += 512)
        if (hsel[i]) atomicAdd(&st->hist[i / STP_SEL_BINS][i % STP_SEL_BINS], (unsigned long long)hsel[i]);
}

// one workgroup (one wave per rank): locate each rank's bin in its group's histogram, descend, regroup, clear
__global__ __launch_bounds__(1024) void k_sel_pick(stp_sel_state* __restrict__ st, int pass)
{
    __shared__ unsigned long long s_new[STP_SEL_MAXR];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int R = st->nranks, shift = stp_sel_shift[pass];
    if (wave < R) {
        const unsigned long long* h = pass == 0 ? st->hist0 : st->hist[st->group[wave]];
        const unsigned long long k = st->k[wave];
        // lane L owns bins [32 L, 32 L + 32): partial sums, wave scan, the owning lane walks its bins
        unsigned long long mine = 0;
        for (int b = 0; b < STP_SEL_BINS / 64; b++) mine += h[lane * (STP_SEL_BINS / 64) + b];
        unsigned long long incl = mine;
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long x = __shfl_up(incl, o); if (lane >= o) incl += x; }
        const unsigned long long excl = incl - mine;
        if (excl <= k && k < incl) {
            unsigned long long acc = excl;
            int d = lane * (STP_SEL_BINS / 64);
            for (;; d++) { const unsigned long long c = h[d]; if (acc + c > k) break; acc += c; }
            st->k[wave] = k - acc;
            s_new[wave] = st->prefix[wave] | ((unsigned long long)d << shift);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                   // regroup: ranks with equal prefixes share a histogram
        int ng = 0;
        for (int r = 0; r < R; r++) {
            st->prefix[r] = s_new[r];
            int g = -1;
            for (int q = 0; q < ng; q++) if (st->gprefix[q] == s_new[r]) { g = q; break; }
            if (g < 0) { g = ng++; st->gprefix[g] = s_new[r]; }
            st->group[r] = g;
        }
        st->ngroups = ng;
    }
    __syncthreads();
    if (pass > 0)
        for (int i = threadIdx.x; i < R * STP_SEL_BINS; i += 1024) st->hist[i / STP_SEL_BINS][i % STP_SEL_BINS] = 0;
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
