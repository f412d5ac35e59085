// v_mfma_f32_16x16x4_f32 on gfx950: numerics, fragment layout and how it shares a SIMD with vector instructions (tools only;
// build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/ubench_mfma tools/ubench_mfma.hip).
//  1. numerics / layout: D = A B + C of one wave against the host's fmaf chain in k order 0..3, bit for bit, with the operand
//     maps A[l & 15][l >> 4], B[l >> 4][l & 15], D[4 (l >> 4) + i][l & 15] -- and D handed on as the next product's B operand
//     (Y = W D, the contraction over D's row index) and as its A operand (Z = D^T W), the two chained forms k_canny_mf uses.
//  2. issue: nM waves per SIMD issuing MFMAs back to back (four independent accumulators) beside nV waves per SIMD issuing
//     v_fma_f32 / v_pk_fma_f32 / v_add_f32 -- cycles per MFMA and per vector instruction, alone and together.
//  3. one wave: an MFMA followed by k independent vector instructions (what a single wave can hide in an MFMA's shadow).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void k_num(const float* A /*16x4*/, const float* B /*4x16*/, const float* C /*16x16*/, const float* W /*16x16*/, float* D,
                      float* Y, float* Z)
{
    const int l = threadIdx.x, n = l & 15, g = l >> 4;
    f4 c;
    for (int i = 0; i < 4; i++) c[i] = C[(4 * g + i) * 16 + n];
    f4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(A[n * 4 + g], B[g * 16 + n], c, 0, 0, 0);
    for (int i = 0; i < 4; i++) D[(4 * g + i) * 16 + n] = d[i];
    // Y = W D: k-step i takes register i of every lane: B[k = g][n] = D[4 g + i][n]; the A operand is W[m][4 g + i]
    f4 y = {0, 0, 0, 0}, z = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++) y = __builtin_amdgcn_mfma_f32_16x16x4f32(W[n * 16 + 4 * g + i], d[i], y, 0, 0, 0);
    for (int i = 0; i < 4; i++) Y[(4 * g + i) * 16 + n] = y[i];
    // Z = D^T W: A[m = n][k = g] = D[4 g + i][n]; the B operand is W[4 g + i][n']
    for (int i = 0; i < 4; i++) z = __builtin_amdgcn_mfma_f32_16x16x4f32(d[i], W[(4 * g + i) * 16 + n], z, 0, 0, 0);
    for (int i = 0; i < 4; i++) Z[(4 * g + i) * 16 + n] = z[i];
}

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
// role of a wave: its index / 4 < nM -> MFMA loop, else vector loop of kind vop.  BF: the MFMA waves issue v_mfma_f32_16x16x32_bf16
// (the real matrix pipe) instead of v_mfma_f32_16x16x4_f32 -- the control experiment for the co-execution counters
template <int VOP, bool BF = false>
__global__ __launch_bounds__(1024) void k_mix(int nM, int iters, float* out, long long* cyc, float a0)
{
    const int w = threadIdx.x >> 6;
    const bool mf = (w >> 2) < nM;
    float r = 0;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (mf) {
        f4 acc[4];
        for (int j = 0; j < 4; j++) acc[j] = f4{a0, a0, a0, a0};
        const float a = a0 + threadIdx.x, b = a0 * 0.5f;
        if (BF) {
            bf8 ab, bb;
            for (int i = 0; i < 8; i++) { ab[i] = (__bf16)(a0 + 0.01f * i); bb[i] = (__bf16)(0.5f * a0); }
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[j], 0, 0, 0);
            }
        } else
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 4; j++) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else {
        float x[8]; f2 p[8];
        for (int i = 0; i < 8; i++) { x[i] = a0 + i + threadIdx.x; p[i] = f2{a0 + i, a0 - i}; }
        const float b = a0 * 0.999f, c = a0 * 0.5f;
        const f2 bp = f2{b, b}, cp = f2{c, c};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (VOP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "v"(c));
                    if (VOP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(bp), "v"(cp));
                    if (VOP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                }
        }
        for (int i = 0; i < 8; i++) r += x[i] + p[i].x + p[i].y;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + w] = t1 - t0;
}

// one wave per SIMD: an MFMA, then k independent v_fma_f32 / v_add_f32
template <int K, int VOP>
__global__ __launch_bounds__(256) void k_shadow(int iters, float* out, long long* cyc, float a0)
{
    f4 acc[2] = {f4{a0, a0, a0, a0}, f4{a0, a0, a0, a0}};
    float x[8];
    for (int i = 0; i < 8; i++) x[i] = a0 + i + threadIdx.x;
    const float a = a0 + threadIdx.x, b = a0 * 0.5f, c = a0 * 0.25f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < K; i++) {
                if (VOP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(b), "v"(c));
                else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i & 7]) : "v"(c));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int j = 0; j < 2; j++) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int i = 0; i < 8; i++) r += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static float frand() { return (float)(rand() / (double)RAND_MAX) * 2.0f - 0.7f; }

static void numerics()
{
    std::vector<float> A(64), B(64), C(256), W(256), D(256), Y(256), Z(256);
    for (auto& v : A) v = frand();
    for (auto& v : B) v = frand();
    for (auto& v : C) v = frand();
    for (auto& v : W) v = frand();
    float *dA, *dB, *dC, *dW, *dD, *dY, *dZ;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dC, 1024); hipMalloc(&dW, 1024); hipMalloc(&dD, 1024); hipMalloc(&dY, 1024); hipMalloc(&dZ, 1024);
    hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_num, dim3(1), dim3(64), 0, 0, dA, dB, dC, dW, dD, dY, dZ);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost); hipMemcpy(Y.data(), dY, 1024, hipMemcpyDeviceToHost); hipMemcpy(Z.data(), dZ, 1024, hipMemcpyDeviceToHost);
    int badD = 0, badDrev = 0, badY = 0, badZ = 0;
    std::vector<float> Dh(256);
    for (int m = 0; m < 16; m++)
        for (int n = 0; n < 16; n++) {
            float f = C[m * 16 + n], r = C[m * 16 + n];
            for (int k = 0; k < 4; k++) f = fmaf(A[m * 4 + k], B[k * 16 + n], f);
            for (int k = 3; k >= 0; k--) r = fmaf(A[m * 4 + k], B[k * 16 + n], r);
            Dh[m * 16 + n] = f;
            badD += memcmp(&f, &D[m * 16 + n], 4) != 0;
            badDrev += memcmp(&r, &D[m * 16 + n], 4) != 0;
        }
    for (int m = 0; m < 16; m++)
        for (int n = 0; n < 16; n++) {
            // the chained forms sum over k = 4 g + i in the order i = 0..3 (the MFMAs), g = 0..3 inside each
            float y = 0, z = 0;
            for (int i = 0; i < 4; i++)
                for (int g = 0; g < 4; g++) {
                    y = fmaf(W[m * 16 + 4 * g + i], Dh[(4 * g + i) * 16 + n], y);
                    z = fmaf(Dh[(4 * g + i) * 16 + m], W[(4 * g + i) * 16 + n], z);
                }
            badY += memcmp(&y, &Y[m * 16 + n], 4) != 0;
            badZ += memcmp(&z, &Z[m * 16 + n], 4) != 0;
        }
    printf("numerics: D vs fmaf chain k=0..3: %d of 256 differ (k=3..0 order: %d differ); Y = W D chained: %d differ; Z = D^T W chained: %d differ\n",
           badD, badDrev, badY, badZ);
}

template <int VOP, bool BF = false>
static void mix(const char* vname, int nM, int nV, float* out, long long* cyc)
{
    const int waves = 4 * (nM + nV), iters = 4096, blocks = 256;
    hipLaunchKernelGGL((k_mix<VOP, BF>), dim3(blocks), dim3(64 * waves), 0, 0, nM, iters, out, cyc, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix<VOP, BF>), dim3(blocks), dim3(64 * waves), 0, 0, nM, iters, out, cyc, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 16);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double cm = 0, cv = 0;
    for (int b = 0; b < blocks; b++)
        for (int w = 0; w < waves; w++) ((w >> 2) < nM ? cm : cv) += h[b * 16 + w];
    if (nM) cm /= (double)blocks * 4 * nM;
    if (nV) cv /= (double)blocks * 4 * nV;
    // per SIMD: nM waves x iters x 16 MFMAs, nV waves x iters x 16 vector instructions
    printf("  %d %s + %d %-12s waves/SIMD: %.3f ms", nM, BF ? "bf16-MFMA" : "MFMA", nV, vname, ms);
    if (nM) printf(" | MFMA wave %.0f cyc = %.1f cyc per MFMA and SIMD", cm, cm / (iters * 16.0 * nM));
    if (nV) printf(" | vector wave %.0f cyc = %.2f cyc per instruction and SIMD", cv, cv / (iters * 16.0 * nV));
    printf("\n");
}

template <int K, int VOP>
static void shadow(float* out, long long* cyc)
{
    const int iters = 4096, blocks = 256;
    hipLaunchKernelGGL((k_shadow<K, VOP>), dim3(blocks), dim3(256), 0, 0, iters, out, cyc, 1.0f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k_shadow<K, VOP>), dim3(blocks), dim3(256), 0, 0, iters, out, cyc, 1.0f);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double c = 0;
    for (auto v : h) c += v;
    c /= h.size();
    printf("  MFMA + %2d %s: %.1f cyc per (MFMA + fillers)\n", K, VOP ? "v_add_f32" : "v_fma_f32", c / (iters * 2.0));
}

int main()
{
    numerics();
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipMalloc(&cyc, 256 * 16 * sizeof(long long));
    printf("issue (wave cycles by s_memtime; one workgroup of 4 (nM + nV) waves per CU):\n");
    for (int nM = 1; nM <= 2; nM++) mix<0>("-", nM, 0, out, cyc);
    mix<0>("v_fma_f32", 0, 1, out, cyc); mix<0>("v_fma_f32", 0, 2, out, cyc); mix<0>("v_fma_f32", 0, 3, out, cyc);
    mix<1>("v_pk_fma_f32", 0, 2, out, cyc); mix<2>("v_add_f32", 0, 2, out, cyc);
    mix<0>("v_fma_f32", 1, 1, out, cyc); mix<0>("v_fma_f32", 1, 2, out, cyc); mix<0>("v_fma_f32", 1, 3, out, cyc); mix<0>("v_fma_f32", 2, 2, out, cyc);
    mix<1>("v_pk_fma_f32", 1, 2, out, cyc); mix<2>("v_add_f32", 1, 2, out, cyc); mix<2>("v_add_f32", 1, 3, out, cyc);
    printf("control: v_mfma_f32_16x16x32_bf16 (the matrix pipe proper) beside the same vector waves:\n");
    mix<0, true>("-", 1, 0, out, cyc); mix<0, true>("v_fma_f32", 1, 1, out, cyc); mix<0, true>("v_fma_f32", 1, 2, out, cyc); mix<0, true>("v_fma_f32", 1, 3, out, cyc);
    mix<1, true>("v_pk_fma_f32", 1, 2, out, cyc);
    printf("one wave per SIMD, fillers in the MFMA's shadow:\n");
    shadow<0, 0>(out, cyc); shadow<2, 0>(out, cyc); shadow<4, 0>(out, cyc); shadow<6, 0>(out, cyc); shadow<8, 0>(out, cyc); shadow<12, 0>(out, cyc);
    shadow<4, 1>(out, cyc); shadow<8, 1>(out, cyc); shadow<12, 1>(out, cyc);
    return 0;
}
