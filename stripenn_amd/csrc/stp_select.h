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
