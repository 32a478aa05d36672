// How much VALU work fits in the shadow of v_mfma_f32_32x32x16_f16 (8 passes = 32 cycles) when ONE wave per SIMD issues both?
// One workgroup of 4 waves per CU (512 registers per lane), a loop of 4 independent MFMA chains with K VALU instructions of a
// given kind behind every MFMA.  Prints cycles per MFMA (s_memtime) for K = 0..8 and each kind; 32 = the matrix pipe never waits.
//   kinds: 0 v_add_f32 on VGPRs (independent)      1 v_accvgpr_read of another accumulator set
//          2 read + v_cmp (VCC) + v_addc (VCC)     3 read + v_cmp -> SGPR pair, v_addc from the pair written one slot earlier
//          4 kind 2 on VGPR-resident values (no accvgpr_read)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shadow mfma_shadow.hip && ./mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <int KIND, int K, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void shadow_kernel(const float* __restrict__ in, float* __restrict__ out, long long* cyc, int iters) {
    const int lane = threadIdx.x;
    h16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)in[lane + j]; b[j] = (_Float16)in[lane + 8 + j]; }
    f32x16 c[4], prev[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { c[i][r] = 0.f; prev[i][r] = in[(lane + i * 16 + r) & 1023]; }
    float tau = in[lane & 63];
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = in[lane + j];
    unsigned bits = 0;
    unsigned long long m0 = 0, m1 = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t)                      // 16 MFMAs per iteration
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[i], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int e = (t * 4 + i) * K + k;
                    if (KIND == 0) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k % 8]) : "v"(tau)); }
                    else if (KIND == 1) { float x; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(prev[(e >> 4) & 3][e & 15])); asm volatile("" :: "v"(x)); }
                    else if (KIND == 2) {
                        if (k % 3 == 0) { asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[0]) : "a"(prev[(e >> 4) & 3][e & 15])); }
                        else if (k % 3 == 1) { asm volatile("v_cmp_nlt_f32_e32 vcc, %0, %1" :: "v"(v[0]), "v"(tau) : "vcc"); }
                        else { asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(bits) :: "vcc"); }
                    } else if (KIND == 3) {
                        if (k % 3 == 0) { asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[0]) : "a"(prev[(e >> 4) & 3][e & 15])); }
                        else if (k % 3 == 1) { asm volatile("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(bits), "=s"(m1) : "s"(m0)); }
                        else { asm volatile("v_cmp_nlt_f32_e64 %0, %1, %2" : "=s"(m0) : "v"(v[0]), "v"(tau)); }
                    } else if (KIND == 4) {
                        if (k % 2 == 0) { asm volatile("v_cmp_nlt_f32_e32 vcc, %0, %1" :: "v"(v[k % 8]), "v"(tau) : "vcc"); }
                        else { asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(bits) :: "vcc"); }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += c[i][r] + prev[i][r];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + lane] = s + (float)bits + (float)(m0 + m1);
    if (lane == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int K, int W>
static void run(const float* in, float* out, long long* cyc, int iters, int grid) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((shadow_kernel<KIND, K, W>), dim3(grid), dim3(256), 0, 0, in, out, cyc, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((shadow_kernel<KIND, K, W>), dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    const double nm = 16.0 * iters * W;                 // MFMAs per SIMD (W waves share it)
    printf("kind %d  K %d  waves/SIMD %d: %7.1f us, %6.1f ns per MFMA per SIMD = %5.1f cycles at 2.4 GHz   (counter: %.1f ticks per MFMA of one wave)\n",
           KIND, K, W, ms * 1e3, ms * 1e6 / nm, ms * 1e6 / nm * 2.4, (double)h / (16.0 * iters));
    fflush(stdout);
}

int main() {
    float *in, *out; long long* cyc;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 512 * 256 * 4)); CK(hipMalloc(&cyc, 8));
    CK(hipMemset(in, 0, 4096 * 4));
    const int iters = 2000;
#define ROW(KIND, W, G) run<KIND, 0, W>(in, out, cyc, iters, G); run<KIND, 2, W>(in, out, cyc, iters, G); run<KIND, 4, W>(in, out, cyc, iters, G); \
    run<KIND, 6, W>(in, out, cyc, iters, G); run<KIND, 8, W>(in, out, cyc, iters, G);
    ROW(0, 1, 256) ROW(1, 1, 256) ROW(2, 1, 256) ROW(3, 1, 256) ROW(4, 1, 256)
    ROW(0, 2, 512) ROW(2, 2, 512)
    return 0;
}
