// Round 6: what can a fully fused F(4x4) convolution (raw patches in, B^T d B in registers, all 36 position GEMMs
// in one workgroup, A^T M A out) achieve AT BEST on MI355X?  Its shape is fixed by the register file and the LDS: one 4-wave workgroup
// owns 32 tiles x 32 output channels x 36 positions (9 accumulators of 32x32 per wave) and, per 8 input channels, needs
//   U slab   36 positions x 32 couts x 8 channels x 4 B = 36.9 KB   (re-read by every tile group: 51 x 9.4 MB = 481 MB at 256 -> 256 / 1 620 tiles)
//   input    32 tiles x 36 pixels x 8 channels x 4 B    = 36.9 KB   (re-read by every cout group)
// for 36 MFMAs (v_mfma_f32_32x32x2_f32) per wave: 8 FLOP per operand byte, against 32 for the 128x128 position GEMM tile of today.
// This probe is that kernel WITHOUT its transforms and epilogue - only the operand stream (global -> LDS, double buffered, one barrier
// per step), the fragment reads and the MFMAs - so its rate is an UPPER bound of the fused kernel's.  Swept over the bytes per step
// (intensity 8 / 16 / 32 FLOP/B) and over one / two workgroups per CU where the LDS allows.
//   hipcc --offload-arch=gfx950 -O3 -o feed_sweep feed_sweep.hip && ./feed_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

// NL = 16-byte loads per thread and step from EACH of the two operand streams (bytes per step = 2 * NL * 4 KB); NACC accumulators per wave
template <int NL, int NACC>
__global__ __launch_bounds__(256) void feed_kernel(const float* __restrict__ U, const float* __restrict__ X, float* __restrict__ out,
                                                   int steps, int tile_groups, int cout_groups, size_t u_chunks, size_t x_chunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CH = NL * 1024;                      // floats per operand chunk (NL x 4 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    { const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3); }
    const int cb = bid % cout_groups, tg = bid / cout_groups;
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 ru[NL], rx[NL];
    auto fetch = [&](int k) {
        const float* up = U + ((size_t)(cb * steps + k) % u_chunks) * CH;
        const float* xp = X + ((size_t)(tg * steps + k) % x_chunks) * CH;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            ru[i] = *reinterpret_cast<const f32x4*>(up + (size_t)(i * 256 + tid) * 4);
            rx[i] = *reinterpret_cast<const f32x4*>(xp + (size_t)(i * 256 + tid) * 4);
        }
    };
    auto stash = [&](int buf) {
        float* b = lds + (size_t)buf * 2 * CH;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            *reinterpret_cast<f32x4*>(b + (size_t)(i * 256 + tid) * 4) = ru[i];
            *reinterpret_cast<f32x4*>(b + CH + (size_t)(i * 256 + tid) * 4) = rx[i];
        }
    };
    fetch(0); stash(0);
    __syncthreads();
    for (int k = 0; k < steps; ++k) {
        const int buf = k & 1;
        if (k + 1 < steps) fetch(k + 1);
        const float* b = lds + (size_t)buf * 2 * CH;
        // per accumulator: one A fragment and one B fragment (16 bytes per lane, lane-linear: conflict-free) feed four MFMAs (8 channels)
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(b + CH + ((size_t)((wave * NACC + i) * 64 + lane) * 4) % CH);
            const f32x4 w = *reinterpret_cast<const f32x4*>(b + ((size_t)((wave * NACC + i) * 64 + lane) * 4) % CH);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], w[s], acc[i], 0, 0, 0);
        }
        if (k + 1 < steps) stash(buf ^ 1);
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int NL, int NACC>
static void run(const float* U, const float* X, float* out, int steps, int tg, int cg, size_t ubytes, size_t xbytes, const char* what) {
    const size_t lds = (size_t)2 * 2 * NL * 4096;
    auto kern = feed_kernel<NL, NACC>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const size_t uc = ubytes / (NL * 4096), xc = xbytes / (NL * 4096);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = tg * cg, reps = 20;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, U, X, out, steps, tg, cg, uc, xc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, U, X, out, steps, tg, cg, uc, xc);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    const double flop = (double)grid * steps * 4 * NACC * 4 * 4096.0;          // 4 waves x NACC accumulators x 4 MFMAs x 32*32*2*2
    const double bytes = (double)grid * steps * 2 * NL * 4096.0;
    printf("%-58s grid %4d  steps %3d  LDS %6.1f KB  %7.1f us  %6.1f TF/s  %5.2f TB/s into LDS  %4.1f FLOP/B\n", what, grid, steps, lds / 1024.0, us,
           flop / us * 1e-6, bytes / us * 1e-6, flop / bytes);
    fflush(stdout);
}

int main() {
    const size_t ub = (size_t)36 * 256 * 256 * 4, xb = (size_t)60 << 20;        // U of a 256 -> 256 layer; a V-sized input-side buffer
    float *U, *X, *out;
    CK(hipMalloc(&U, ub)); CK(hipMalloc(&X, xb)); CK(hipMalloc(&out, (size_t)4096 * 256 * 4));
    { size_t n = xb / 4; float* h = (float*)malloc(xb); unsigned long long rs = 88172645463325252ull;
      for (size_t i = 0; i < n; ++i) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; h[i] = (float)((rs >> 40) * (1.0 / 8388608.0) - 1.0); }
      CK(hipMemcpy(X, h, xb, hipMemcpyHostToDevice)); CK(hipMemcpy(U, h, ub, hipMemcpyHostToDevice)); free(h); }
    printf("# 256 -> 256 at 120 x 216 (1 620 tiles): 51 tile groups x 8 cout groups, 32 steps of 8 channels; 7.64 GFLOP executed.\n"
           "# today (plan 23, isolated): input transform 19 + position GEMMs 84 + output transform 17-20 = 120 us = 63 TF/s over all three, 91 TF/s in the GEMM\n");
    run<9, 9>(U, X, out, 32, 51, 8, ub, xb, "fused shape: 32 tiles x 32 couts x 36 positions");          // 2 x 36.9 KB per step, 36 MFMAs per wave: 8 FLOP/B
    run<9, 9>(U, X, out, 8, 203, 2, ub, xb, "  the same at 64 -> 64, batch 4 (6 480 tiles, 8 steps)");
    run<5, 9>(U, X, out, 32, 51, 8, ub, xb, "half the operand bytes per step (~15 FLOP/B)");
    run<2, 9>(U, X, out, 32, 51, 8, ub, xb, "a quarter (~36 FLOP/B: today's 128x128 GEMM tile)");
    run<9, 4>(U, X, out, 32, 51, 8, ub, xb, "fused operand stream with 4 accumulators (MFMA work / 2.25)");
    return 0;
}
