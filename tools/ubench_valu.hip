// Issue cost of the vector instructions the image kernels are made of, on gfx950 (tools only; build:
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/ubench_valu tools/ubench_valu.hip).
// Per instruction: ns and cycles per wave-instruction and SIMD at 1, 2, 4 and 8 waves per SIMD, from the wall clock of a
// launch of 256 x wps workgroups of 256 threads, each lane running ITER x 64 copies of the instruction on 8 independent
// register groups.  Clock = s_memtime ticks / s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#define ITER 2048

typedef float v2 __attribute__((ext_vector_type(2)));
struct regs { float x[8]; double d[8]; v2 p[8]; unsigned u[8]; };

#define OPS(X) \
    X(0, "v_fma_f32", "v_fma_f32 %0, %0, %4, %5", x) \
    X(1, "v_fmac_f32", "v_fmac_f32 %0, %4, %5", x) \
    X(2, "v_add_f32", "v_add_f32 %0, %0, %4", x) \
    X(3, "v_sub_f32", "v_sub_f32 %0, %0, %4", x) \
    X(4, "v_mul_f32", "v_mul_f32 %0, %0, %4", x) \
    X(5, "v_max_f32", "v_max_f32 %0, %0, %4", x) \
    X(6, "v_fma_f32 sgpr", "v_fma_f32 %0, %0, %8, %5", x) \
    X(7, "v_pk_fma_f32", "v_pk_fma_f32 %2, %2, %6, %7", p) \
    X(8, "v_pk_add_f32", "v_pk_add_f32 %2, %2, %6", p) \
    X(9, "v_pk_mul_f32", "v_pk_mul_f32 %2, %2, %6", p) \
    X(10, "v_fma_f64", "v_fma_f64 %1, %1, %9, %10", d) \
    X(11, "v_add_f64", "v_add_f64 %1, %1, %9", d) \
    X(12, "v_mul_f64", "v_mul_f64 %1, %1, %9", d) \
    X(13, "v_min_f64", "v_min_f64 %1, %1, %9", d) \
    X(14, "v_cvt_f32_f64", "v_cvt_f32_f64 %0, %1", x) \
    X(15, "v_cvt_f64_f32", "v_cvt_f64_f32 %1, %0", d) \
    X(16, "v_add_u32", "v_add_u32 %3, %3, %11", u) \
    X(17, "v_min_u32", "v_min_u32 %3, %3, %11", u) \
    X(18, "v_and_b32", "v_and_b32 %3, %3, %11", u) \
    X(19, "v_lshlrev_b32", "v_lshlrev_b32 %3, 1, %3", u) \
    X(20, "v_lshl_add_u32", "v_lshl_add_u32 %3, %3, 1, %11", u) \
    X(21, "v_add3_u32", "v_add3_u32 %3, %3, %11, %11", u) \
    X(22, "v_mad_u32_u24", "v_mad_u32_u24 %3, %3, %11, %11", u) \
    X(23, "v_mul_lo_u32", "v_mul_lo_u32 %3, %3, %11", u) \
    X(24, "v_bfe_u32", "v_bfe_u32 %3, %3, 1, 8", u) \
    X(25, "v_mov_b32", "v_mov_b32 %3, %11", u) \
    X(26, "v_cndmask_b32", "v_cndmask_b32 %3, %3, %11, vcc", u) \
    X(27, "v_cmp_gt_f32", "v_cmp_gt_f32 vcc, %0, %4", x) \
    X(28, "v_cmp_gt_f32 sgpr", "v_cmp_gt_f32 s[40:41], %0, %4", x) \
    X(29, "v_max3_f32", "v_max3_f32 %0, %0, %4, %5", x) \
    X(30, "v_med3_f32", "v_med3_f32 %0, %0, %4, %5", x) \
    X(31, "v_sqrt_f32", "v_sqrt_f32 %0, %0", x) \
    X(32, "v_rcp_f32", "v_rcp_f32 %0, %0", x) \
    X(33, "v_bcnt_u32_b32", "v_bcnt_u32_b32 %3, %3, %11", u) \
    X(34, "v_mbcnt_lo", "v_mbcnt_lo_u32_b32 %3, %11, %3", u) \
    X(35, "v_readlane", "v_readlane_b32 s40, %3, 3", u) \
    X(36, "v_mov_dpp shr1", "v_mov_b32_dpp %3, %11 row_shr:1 row_mask:0xf bank_mask:0xf", u) \
    X(37, "v_add_f32 dpp", "v_add_f32_dpp %0, %4, %0 row_shr:1 row_mask:0xf bank_mask:0xf", x) \
    X(38, "v_perm_b32", "v_perm_b32 %3, %3, %11, %11", u) \
    X(39, "v_xor_b32", "v_xor_b32 %3, %3, %11", u) \
    X(40, "v_lshlrev_b64", "v_lshlrev_b64 %1, 1, %1", d) \
    X(41, "v_cmp_lt_u32", "v_cmp_lt_u32 vcc, %3, %11", u) \
    X(42, "v_max_f64", "v_max_f64 %1, %1, %9", d) \
    X(43, "v_cmp_gt_f64", "v_cmp_gt_f64 vcc, %1, %9", d) \
    X(44, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %3", x) \
    X(45, "v_alignbit_b32", "v_alignbit_b32 %3, %3, %11, 3", u) \
    X(46, "v_pk_fma_f32 sgpr", "v_pk_fma_f32 %2, %2, s[42:43], %7", p) \
    X(47, "v_mul_f32 x2.0", "v_mul_f32 %0, 2.0, %0", x) \
    X(48, "v_add_co_u32", "v_add_co_u32 %3, vcc, %3, %11", u) \
    X(49, "v_ffbl_b32", "v_ffbl_b32 %3, %3", u)

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, long long* rt, float a0, float b0)
{
    regs r;
    for (int i = 0; i < 8; i++) { r.x[i] = a0 + i + threadIdx.x; r.d[i] = a0 + i; r.p[i] = v2{a0 + i, b0 + i}; r.u[i] = threadIdx.x + i; }
    const float b = b0, c = a0 * 0.5f;
    const double bd = b0, cd = a0 * 0.5;
    const v2 bp = v2{b0, b0 + 1.f}, cp = v2{a0, a0 + 2.f};
    const unsigned ub = (unsigned)(a0 * 3.0f);
    const float sb = b0;                 // uniform: lives in an SGPR
    asm volatile("s_mov_b32 s42, 1.0\n\ts_mov_b32 s43, 1.0" ::: "s42", "s43");
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
#define X(N, NAME, ASM, G) \
                if (OP == N) asm volatile(ASM : "+v"(r.x[i]), "+v"(r.d[i]), "+v"(r.p[i]), "+v"(r.u[i]) \
                                          : "v"(b), "v"(c), "v"(bp), "v"(cp), "s"(sb), "v"(bd), "v"(cd), "v"(ub) : "vcc", "s40", "s41");
                OPS(X)
#undef X
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += r.x[i] + (float)r.d[i] + r.p[i].x + r.p[i].y + (float)r.u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int OP>
static void run(const char* name, const char* only)
{
    if (only && !strstr(name, only)) return;
    float* out; long long *cyc, *rt;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipMalloc(&cyc, 256 * 8 * sizeof(long long));
    hipMalloc(&rt, 256 * 8 * sizeof(long long));
    printf("%-18s", name);
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, rt, 1.0f, 1.000001f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, rt, 1.0f, 1.000001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks), hr(blocks);
        hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        hipMemcpy(hr.data(), rt, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        double avg = 0, avr = 0; for (int i = 0; i < blocks; i++) { avg += h[i]; avr += hr[i]; } avg /= blocks; avr /= blocks;
        const double n = (double)ITER * 64, clk = avg / (avr * 10.0), ns = ms * 1e6 / (n * wps);
        printf(" | w%d %.2f ns = %.2f cyc @%.2f GHz", wps, ns, ns * clk, clk);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("\n");
    hipFree(out); hipFree(cyc); hipFree(rt);
}

int main(int argc, char** argv)
{
    const char* only = argc > 1 ? argv[1] : nullptr;
#define X(N, NAME, ASM, G) run<N>(NAME, only);
    OPS(X)
#undef X
    return 0;
}
