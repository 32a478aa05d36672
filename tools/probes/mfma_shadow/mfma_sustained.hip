// Sustained matrix-pipe rate against operand data: a register-only loop of independent MFMA chains on every SIMD of the chip
// (no memory, no LDS, no other instructions), operands all zero vs random.  s_memtime counts the same ticks per MFMA either
// way; the wall-clock time does not: the chip clocks to its power budget (MI355X_MICROARCH.md, "DVFS give-back"), and dense
// MFMAs on real data sit well below the 2.4 GHz the nominal peaks are quoted at.  Prints TFLOP/s and the fraction of the
// nominal peak: the second is the ceiling any kernel built on that instruction has on this box.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_sustained mfma_sustained.hip && ./mfma_sustained [ms per measurement, default 8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <int KIND>   // 0: v_mfma_f32_32x32x16_f16   1: v_mfma_f32_32x32x2_f32
__global__ __launch_bounds__(256, 2) void sustained_kernel(const float* __restrict__ in, float* __restrict__ out, long long* cyc, int iters) {
    const int lane = threadIdx.x;
    h16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)in[(lane * 8 + j) & 4095]; b[j] = (_Float16)in[(lane * 8 + j + 2048) & 4095]; }
    const float fa = in[lane & 4095], fb = in[(lane + 999) & 4095];
    f32x16 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (KIND == 0) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[i], 0, 0, 0);
                else c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c[i], 0, 0, 0);
            }
        // keep the accumulators bounded (random operands would overflow to inf, which is cheap data again): halve now and then
        if ((it & 63) == 63) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) c[i][r] *= 0.015625f;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += c[i][r];
    out[blockIdx.x * 256 + lane] = s;
    if (lane == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
static void run(const char* name, double flop_per_mfma, double nominal_tf, const float* in, float* out, long long* cyc, double target_ms, const char* data) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int iters = 2000;
    float ms = 0.f;
    for (int pass = 0; pass < 2; ++pass) {               // pass 0 sizes the loop, pass 1 measures
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((sustained_kernel<KIND>), dim3(512), dim3(256), 0, 0, in, out, cyc, iters);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (pass == 0) iters = (int)(iters * target_ms / ms);
    }
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    const double n_mfma = 16.0 * iters * 512 * 4;        // chip-wide
    const double tf = n_mfma * flop_per_mfma / (ms * 1e-3) / 1e12;
    const double cyc_per = (double)h / (16.0 * iters) / 2.0;   // two waves share a SIMD
    printf("%-26s %-7s %8.2f ms  %7.1f TFLOP/s = %.2f of the nominal %.0f   (%.1f s_memtime ticks per MFMA per SIMD)\n",
           name, data, ms, tf, tf / nominal_tf, nominal_tf, cyc_per);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double target_ms = argc > 1 ? atof(argv[1]) : 8.0;
    float *in, *out; long long* cyc;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 512 * 256 * 4)); CK(hipMalloc(&cyc, 8));
    for (int zeros = 1; zeros >= 0; --zeros) {
        float h[4096]; unsigned long long rs = 88172645463325252ull;
        for (int i = 0; i < 4096; ++i) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; h[i] = zeros ? 0.f : (float)((rs >> 40) * (1.0 / 8388608.0) - 1.0); }
        CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; ++rep) {
            run<0>("v_mfma_f32_32x32x16_f16", 32768.0, 2500.0, in, out, cyc, target_ms, zeros ? "zeros" : "random");
            run<1>("v_mfma_f32_32x32x2_f32", 4096.0, 157.3, in, out, cyc, target_ms, zeros ? "zeros" : "random");
        }
    }
    // short launches (the B32 filter is ~30 us): does the clock hold for a kernel that short after an idle gap?
    {
        float h[4096]; unsigned long long rs = 88172645463325252ull;
        for (int i = 0; i < 4096; ++i) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; h[i] = (float)((rs >> 40) * (1.0 / 8388608.0) - 1.0); }
        CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
        run<0>("v_mfma_f32_32x32x16_f16", 32768.0, 2500.0, in, out, cyc, 0.03, "random");
        run<0>("v_mfma_f32_32x32x16_f16", 32768.0, 2500.0, in, out, cyc, 0.3, "random");
        run<1>("v_mfma_f32_32x32x2_f32", 4096.0, 157.3, in, out, cyc, 0.05, "random");
        run<1>("v_mfma_f32_32x32x2_f32", 4096.0, 157.3, in, out, cyc, 0.5, "random");
    }
    return 0;
}
