// Packed-f32 probe, stage 3: a SELF-CONTAINED reproducer (plain hipcc, no torch, no library) shaped like the case that
// tools/probes/pk_hazard/lib_probe.cpp reproduces through the library: the victim (upsample2x_add arithmetic, packed / unpacked
// f32 VALU builds) launched back to back on stream 1 while stream 2 runs MANY SHORT workgroups (36 KB of LDS each, like the
// library's 64x64 GEMM tiles), so that victim waves become co-resident with aggressor waves on the same SIMDs as slots turn over.
// Aggressor ingredients are switched on one by one to find which instruction class it takes:
//   bit 0: v_mfma_f32_32x32x16_f16 (accumulators in AGPRs)      bit 1: fp32 -> fp16 split conversions (v_cvt_pkrtz_f16_f32, v_cvt_f16_f32, v_med3_f32)
//   bit 2: LDS b128 write / read + s_barrier                     bit 3: v_mfma_f32_32x32x2_f32 instead of the fp16 MFMA (control)
//   bit 4: global loads feeding the conversions
// Output: one line per (victim build, aggressor mask): rounds, rounds with wrong elements, wrong elements.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void pk_upsample(const float*, const float*, float*, int, int, int, int);
__global__ void nopk_upsample(const float*, const float*, float*, int, int, int, int);

__device__ __forceinline__ f32x4 split_pack(const f32x4 v) {      // the library's split_pack (csrc/conv_mfma.hip)
    const fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y), h23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
    h16x8 o;
    o[0] = (_Float16)h01.x; o[1] = (_Float16)h01.y; o[2] = (_Float16)h23.x; o[3] = (_Float16)h23.y;
    o[4] = (_Float16)__builtin_amdgcn_fmed3f(v.x - (float)h01.x, -65504.f, 65504.f);
    o[5] = (_Float16)__builtin_amdgcn_fmed3f(v.y - (float)h01.y, -65504.f, 65504.f);
    o[6] = (_Float16)__builtin_amdgcn_fmed3f(v.z - (float)h23.x, -65504.f, 65504.f);
    o[7] = (_Float16)__builtin_amdgcn_fmed3f(v.w - (float)h23.y, -65504.f, 65504.f);
    return __builtin_bit_cast(f32x4, o);
}

template <int MASK>
__global__ __launch_bounds__(256) void aggressor(const float* __restrict__ src, float* __restrict__ sink, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[9216];         // 36 KB: four workgroups per CU, like the 64x64 GEMM tile
    const int tid = threadIdx.x;
    f32x4 x = {0.001f * tid, 0.002f * tid, -0.003f * tid, 0.5f};
    f32x4 y = {1.f, -2.f, 0.25f, 0.125f};
    f32x16 acc = {};
    for (int it = 0; it < iters; ++it) {
        if (MASK & 16) x += reinterpret_cast<const f32x4*>(src)[(size_t)blockIdx.x * 256 * 4 + (size_t)(it & 3) * 256 + tid];
        f32x4 a = x, b = y;
        if (MASK & 2) { a = split_pack(x); b = split_pack(y); }
        if (MASK & 4) {
            reinterpret_cast<f32x4*>(lds)[tid] = a;
            reinterpret_cast<f32x4*>(lds)[256 + tid] = b;
            __syncthreads();
            a = reinterpret_cast<f32x4*>(lds)[(tid + 17) & 255];
            b = reinterpret_cast<f32x4*>(lds)[256 + ((tid + 5) & 255)];
            __syncthreads();
        }
        if (MASK & 1) {
            const h16x8 ha = __builtin_bit_cast(h16x8, a), hb = __builtin_bit_cast(h16x8, b);
            const h16x8 hs = __builtin_shufflevector(hb, hb, 4, 5, 6, 7, 0, 1, 2, 3);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hs, acc, 0, 0, 0);
            }
        }
        if (MASK & 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            }
        }
        x.x += a.w * 1e-9f; y.y += b.z * 1e-9f;
    }
    float s = x.x + y.y;
    for (int r = 0; r < 16; ++r) s += acc[r];
    sink[(size_t)blockIdx.x * 256 + tid] = s;
}

static unsigned long long rs = 0x1234567ull;
static float urand() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (float)((rs >> 40) * (1.0 / 8388608.0) - 1.0); }

template <int MASK>
static void launch_aggr(hipStream_t s, const float* src, float* sink, int grid, int iters) {
    hipLaunchKernelGGL(aggressor<MASK>, dim3(grid), dim3(256), 0, s, src, sink, iters);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 60;
    const int h = 60, w = 108, C = 256;
    const size_t ng = (size_t)h * w * C, nout = ng * 4;
    std::vector<float> hg(ng), hs(nout);
    for (auto& v : hg) v = urand();
    for (auto& v : hs) v = urand();
    float *g, *skip, *out, *sink, *src;
    const int agrid = 4096;
    CK(hipMalloc(&g, ng * 4)); CK(hipMalloc(&skip, nout * 4)); CK(hipMalloc(&out, nout * 4)); CK(hipMalloc(&sink, (size_t)agrid * 256 * 4));
    CK(hipMalloc(&src, (size_t)agrid * 256 * 4 * 16)); CK(hipMemset(src, 0, (size_t)agrid * 256 * 4 * 16));
    CK(hipMemcpy(g, hg.data(), ng * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(skip, hs.data(), nout * 4, hipMemcpyHostToDevice));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t ea; CK(hipEventCreate(&ea));
    std::vector<float> want(nout), got(nout);
    const size_t total = (size_t)1 * 2 * h * 2 * w * (C / 4);
    const int vgrid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    const int masks[] = {0, 1, 2, 4, 3, 5, 6, 7, 23, 8, 14};
    int any = 0;
    for (int packed = 1; packed >= 0; --packed) {
        auto victim = [&](hipStream_t st) {
            if (packed) hipLaunchKernelGGL(pk_upsample, dim3(vgrid), dim3(256), 0, st, g, skip, out, 1, h, w, C);
            else hipLaunchKernelGGL(nopk_upsample, dim3(vgrid), dim3(256), 0, st, g, skip, out, 1, h, w, C);
        };
        CK(hipMemset(out, 0, nout * 4));
        victim(s1); CK(hipStreamSynchronize(s1));
        CK(hipMemcpy(want.data(), out, nout * 4, hipMemcpyDeviceToHost));
        for (int m : masks) {
            long wrong_total = 0; int wrong_rounds = 0, overlapped = 0;
            for (int r = 0; r < rounds; ++r) {
                CK(hipMemsetAsync(out, 0, nout * 4, s1)); CK(hipStreamSynchronize(s1));
                if (m) {
                    for (int k = 0; k < 12; ++k) {
                        const int it = 24;
                        switch (m) {
                            case 1: launch_aggr<1>(s2, src, sink, agrid, it); break;   case 2: launch_aggr<2>(s2, src, sink, agrid, it * 8); break;
                            case 4: launch_aggr<4>(s2, src, sink, agrid, it * 4); break; case 3: launch_aggr<3>(s2, src, sink, agrid, it); break;
                            case 5: launch_aggr<5>(s2, src, sink, agrid, it); break;   case 6: launch_aggr<6>(s2, src, sink, agrid, it * 4); break;
                            case 7: launch_aggr<7>(s2, src, sink, agrid, it); break;   case 23: launch_aggr<23>(s2, src, sink, agrid, it); break;
                            case 8: launch_aggr<8>(s2, src, sink, agrid, it); break;   case 14: launch_aggr<14>(s2, src, sink, agrid, it); break;
                        }
                    }
                    CK(hipEventRecord(ea, s2));
                }
                for (int k = 0; k < 4; ++k) victim(s1);
                CK(hipStreamSynchronize(s1));
                if (m) { overlapped += hipEventQuery(ea) == hipErrorNotReady ? 1 : 0; CK(hipStreamSynchronize(s2)); }
                CK(hipMemcpy(got.data(), out, nout * 4, hipMemcpyDeviceToHost));
                long wrong = 0;
                for (size_t i = 0; i < nout; ++i) { unsigned a, b; memcpy(&a, &got[i], 4); memcpy(&b, &want[i], 4); wrong += a != b; }
                wrong_total += wrong; wrong_rounds += wrong ? 1 : 0;
            }
            printf("victim upsample2x_add [%-8s] | aggressor mask %2d (%s%s%s%s%s): %3d rounds (%3d finished while the aggressor ran), %d with wrong elements, %ld wrong of %zu\n",
                   packed ? "packed" : "unpacked", m, (m & 1) ? "mfma_f16 " : "", (m & 2) ? "cvt_split " : "", (m & 4) ? "lds+barrier " : "", (m & 8) ? "mfma_f32 " : "",
                   (m & 16) ? "global_loads " : "", rounds, overlapped, wrong_rounds, wrong_total, nout);
            fflush(stdout);
            any |= wrong_total ? 1 : 0;
        }
    }
    printf(any ? "RESULT: corruption reproduced standalone\n" : "RESULT: no corruption in any combination\n");
    return 0;
}
