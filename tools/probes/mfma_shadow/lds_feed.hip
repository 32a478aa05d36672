// The filter's inner loop in isolation: per k-step ONE ds_read_b128 row fragment (row stride 304 B as in the kernel, or 16-byte
// lane-linear for comparison) feeding NQ MFMAs (v_mfma_f32_32x32x16_f16), requested PF steps ahead, plus E compare elements
// (v_cmp -> SGPR pair, v_addc one slot later; VGPR-resident values) behind every MFMA.  No global memory, no barriers.
// cycles per MFMA per SIMD; 32 = the matrix pipe never waits.
//   hipcc --offload-arch=gfx950 -O3 -o lds_feed lds_feed.hip && ./lds_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <int NQ, int E, int W, int LINEAR, int LDSREAD>
__global__ __launch_bounds__(256, W) void feed_kernel(const float* __restrict__ in, float* __restrict__ out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 4 * 32 * 304];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    for (int e = tid; e < (int)sizeof(lds) / 4; e += 256) reinterpret_cast<float*>(lds)[e] = in[e & 1023];
    __syncthreads();
    h16x8 b[NQ][9];
#pragma unroll
    for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) b[i][t][j] = (_Float16)in[(lane + i * 9 + t + j) & 1023];
    f32x16 c[NQ], prev[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { c[i][r] = 0.f; prev[i][r] = in[(lane + i * 16 + r) & 1023]; }
    const float tau = in[lane & 63];
    unsigned bits = 0;
    unsigned long long m0 = 0, m1 = 0;
    const unsigned char* ar = LINEAR ? lds + lane * 16 : lds + l31 * 304 + lh * 16;
    const int tstride = LINEAR ? 1024 : 32, tilestride = LINEAR ? 9 * 1024 : 32 * 304;
    h16x8 fr[9];
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const unsigned char* a0 = ar + (it & 3) * tilestride + ((it >> 2) & 1) * 4 * 32 * 304 * (LINEAR ? 0 : 1);
        if (LDSREAD) { fr[0] = *reinterpret_cast<const h16x8*>(a0); fr[1] = *reinterpret_cast<const h16x8*>(a0 + tstride); }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (LDSREAD && t + 2 < 9) fr[t + 2] = *reinterpret_cast<const h16x8*>(a0 + (t + 2) * tstride);
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(LDSREAD ? fr[t] : b[0][(t + 1) % 9], b[i][t], c[i], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < E; ++k) {
                    const int e = ((t * NQ + i) * E + k) & 15;
                    asm volatile("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(bits), "=s"(m1) : "s"(m0));
                    asm volatile("v_cmp_nlt_f32_e64 %0, %1, %2" : "=s"(m0) : "v"(prev[i][e]), "v"(tau));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += c[i][r] + prev[i][r];
    out[blockIdx.x * 256 + tid] = s + (float)bits + (float)(m0 + m1);
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NQ, int E, int W, int LINEAR, int LDSREAD>
static void run(const float* in, float* out, long long* cyc, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * W;
    hipLaunchKernelGGL((feed_kernel<NQ, E, W, LINEAR, LDSREAD>), dim3(grid), dim3(256), 0, 0, in, out, cyc, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((feed_kernel<NQ, E, W, LINEAR, LDSREAD>), dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    const double nm = 9.0 * NQ * iters * W;
    printf("query blocks %d  compares/MFMA %d  waves/SIMD %d  %s  %s: %7.1f us, %5.1f cycles per MFMA per SIMD at 2.4 GHz (counter: %.1f ticks per MFMA of one wave)\n",
           NQ, E, W, LDSREAD ? "A from LDS" : "A in regs ", LINEAR ? "lane-linear" : "stride 304 ", ms * 1e3, ms * 1e6 / nm * 2.4, (double)h / (9.0 * NQ * iters));
    fflush(stdout);
}

int main(int argc, char** argv) {
    float *in, *out; long long* cyc;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 512 * 256 * 4)); CK(hipMalloc(&cyc, 8));
    const bool zeros = argc > 2 && atoi(argv[2]) == 0;          // operand data: zeros draw far less power than random values
    { float h[4096]; unsigned long long rs = 88172645463325252ull;
      for (int i = 0; i < 4096; ++i) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; h[i] = zeros ? 0.f : (float)((rs >> 40) * (1.0 / 8388608.0) - 1.0); }
      CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice)); }
    printf("operands: %s\n", zeros ? "zeros" : "random in [-1, 1)");
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    run<2, 0, 2, 0, 0>(in, out, cyc, iters); run<2, 0, 2, 0, 1>(in, out, cyc, iters); run<2, 0, 2, 1, 1>(in, out, cyc, iters);
    run<2, 1, 2, 0, 0>(in, out, cyc, iters); run<2, 1, 2, 0, 1>(in, out, cyc, iters); run<2, 1, 2, 1, 1>(in, out, cyc, iters);
    run<2, 2, 2, 0, 0>(in, out, cyc, iters); run<2, 2, 2, 0, 1>(in, out, cyc, iters); run<2, 2, 2, 1, 1>(in, out, cyc, iters);
    run<2, 2, 1, 0, 0>(in, out, cyc, iters); run<2, 2, 1, 0, 1>(in, out, cyc, iters);
    run<4, 0, 1, 0, 1>(in, out, cyc, iters); run<4, 2, 1, 0, 0>(in, out, cyc, iters); run<4, 2, 1, 0, 1>(in, out, cyc, iters);
    run<1, 0, 2, 0, 1>(in, out, cyc, iters); run<1, 2, 2, 0, 1>(in, out, cyc, iters);
    return 0;
}
