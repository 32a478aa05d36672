// Standalone reproducer attempt: does a kernel whose f32x4 arithmetic is compiled to PACKED f32 VALU
// instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) return wrong values while a kernel issuing v_mfma_f32_32x32x16_f16
// runs on ANOTHER stream?  Plain hipcc, no torch, no library: private buffers on both sides, no shared memory.
//
//   bash tools/probes/pk_hazard/build.sh && tools/probes/pk_hazard/pk_hazard
//
// Victims: the library's upsample2x_add arithmetic and a dense f32x4 fma chain, each built twice (packed / unpacked VALU).
// Aggressors: fp16 MFMA loop, fp32 MFMA loop, a VALU-only loop, none.  Every victim launch that overlaps an aggressor is
// compared bit for bit with the victim's solo output.  Prints one line per (victim, build, aggressor): launches, launches with
// any wrong element, wrong elements in total, and the first few (index, got, want) triples.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void pk_upsample(const float*, const float*, float*, int, int, int, int);
__global__ void nopk_upsample(const float*, const float*, float*, int, int, int, int);
__global__ void pk_chain(const float*, const float*, float*, size_t);
__global__ void nopk_chain(const float*, const float*, float*, size_t);

// ---- aggressors: register-only loops, one result word per lane so that nothing is optimised away --------------------------
__global__ __launch_bounds__(256) void aggr_mfma_f16(float* sink, int iters) {
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, acc3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void aggr_mfma_f32(float* sink, int iters) {
    const float a = 0.001f * threadIdx.x, b = 0.002f * (threadIdx.x ^ 5);
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void aggr_valu(float* sink, int iters) {
    float x = 0.001f * threadIdx.x, y = 1.0001f;
    for (int it = 0; it < iters * 16; ++it) { x = fmaf(x, y, 0.5f); y = fmaf(y, 0.9999f, 1e-4f); }
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}

static unsigned long long rs = 0x1234567ull;
static float urand() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (float)((rs >> 40) * (1.0 / 8388608.0) - 1.0); }

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 60;
    const int B = 1, h = 60, w = 108, C = 256;                       // the decoder's 1/8 -> 1/4 resolution upsample at 480p
    const size_t ng = (size_t)B * h * w * C, nout = ng * 4, n4 = nout / 4;
    std::vector<float> hg(ng), hs(nout);
    for (auto& v : hg) v = urand();
    for (auto& v : hs) v = urand();
    float *g, *skip, *out, *sink;
    CK(hipMalloc(&g, ng * 4)); CK(hipMalloc(&skip, nout * 4)); CK(hipMalloc(&out, nout * 4)); CK(hipMalloc(&sink, 4096 * 256 * 4));
    CK(hipMemcpy(g, hg.data(), ng * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(skip, hs.data(), nout * 4, hipMemcpyHostToDevice));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t ev_a0, ev_a1, ev_v0, ev_v1;
    CK(hipEventCreate(&ev_a0)); CK(hipEventCreate(&ev_a1)); CK(hipEventCreate(&ev_v0)); CK(hipEventCreate(&ev_v1));
    std::vector<float> want(nout), got(nout);
    const int vgrid = 2048;
    struct Victim { const char* name; int kind; bool packed; };
    const Victim victims[] = {{"upsample2x_add", 0, true}, {"upsample2x_add", 0, false}, {"fma-chain x48", 1, true}, {"fma-chain x48", 1, false}};
    const char* aggr_names[] = {"none", "mfma_f32_32x32x16_f16", "mfma_f32_32x32x2_f32", "valu loop"};
    auto launch_victim = [&](const Victim& v, hipStream_t st) {
        if (v.kind == 0) {
            if (v.packed) hipLaunchKernelGGL(pk_upsample, dim3(vgrid), dim3(256), 0, st, g, skip, out, B, h, w, C);
            else hipLaunchKernelGGL(nopk_upsample, dim3(vgrid), dim3(256), 0, st, g, skip, out, B, h, w, C);
        } else {
            if (v.packed) hipLaunchKernelGGL(pk_chain, dim3(vgrid), dim3(256), 0, st, skip, skip + nout / 2, out, n4 / 2);
            else hipLaunchKernelGGL(nopk_chain, dim3(vgrid), dim3(256), 0, st, skip, skip + nout / 2, out, n4 / 2);
        }
    };
    int any_bad = 0;
    for (const Victim& v : victims) {
        const size_t nchk = v.kind == 0 ? nout : nout / 2;
        // solo reference (and run-to-run identity of the solo run)
        CK(hipMemset(out, 0, nout * 4));
        launch_victim(v, s1); CK(hipStreamSynchronize(s1));
        CK(hipMemcpy(want.data(), out, nchk * 4, hipMemcpyDeviceToHost));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(out, 0, nout * 4));
            launch_victim(v, s1); CK(hipStreamSynchronize(s1));
            CK(hipMemcpy(got.data(), out, nchk * 4, hipMemcpyDeviceToHost));
            if (memcmp(got.data(), want.data(), nchk * 4)) { printf("%s [%s]: SOLO runs differ from each other\n", v.name, v.packed ? "packed" : "unpacked"); any_bad = 1; }
        }
        for (int ak = 0; ak < 4; ++ak) {
            // aggressor geometries: partial grids leave CU slots for the victim (co-residence on the same SIMDs), full grids queue it
            for (int agrid : {256, 1024}) {
                if (ak == 0 && agrid != 256) continue;
                long wrong_total = 0; int wrong_launches = 0, overlapped = 0; float a_ms = 0, v_ms = 0;
                long first_idx[4]; float first_got[4], first_want[4]; int nfirst = 0;
                for (int r = 0; r < rounds; ++r) {
                    CK(hipMemsetAsync(out, 0, nout * 4, s1));
                    CK(hipStreamSynchronize(s1));
                    CK(hipEventRecord(ev_a0, s2));
                    const int iters = 60000;
                    if (ak == 1) hipLaunchKernelGGL(aggr_mfma_f16, dim3(agrid), dim3(256), 0, s2, sink, iters);
                    else if (ak == 2) hipLaunchKernelGGL(aggr_mfma_f32, dim3(agrid), dim3(256), 0, s2, sink, iters / 8);
                    else if (ak == 3) hipLaunchKernelGGL(aggr_valu, dim3(agrid), dim3(256), 0, s2, sink, iters);
                    CK(hipEventRecord(ev_a1, s2));
                    CK(hipEventRecord(ev_v0, s1));
                    launch_victim(v, s1);
                    CK(hipEventRecord(ev_v1, s1));
                    CK(hipStreamSynchronize(s1));
                    const bool aggr_still_running = (ak != 0) && hipEventQuery(ev_a1) == hipErrorNotReady;
                    CK(hipStreamSynchronize(s2));
                    overlapped += aggr_still_running ? 1 : 0;
                    float t; CK(hipEventElapsedTime(&t, ev_a0, ev_a1)); a_ms += t; CK(hipEventElapsedTime(&t, ev_v0, ev_v1)); v_ms += t;
                    CK(hipMemcpy(got.data(), out, nchk * 4, hipMemcpyDeviceToHost));
                    long wrong = 0;
                    for (size_t i = 0; i < nchk; ++i) {
                        unsigned a, b; memcpy(&a, &got[i], 4); memcpy(&b, &want[i], 4);
                        if (a != b) { if (nfirst < 4) { first_idx[nfirst] = (long)i; first_got[nfirst] = got[i]; first_want[nfirst] = want[i]; ++nfirst; } ++wrong; }
                    }
                    wrong_total += wrong; wrong_launches += wrong ? 1 : 0;
                }
                printf("%-15s [%-8s] aggressor %-22s grid %4d: %3d launches (%3d finished while the aggressor ran; victim %.3f ms, aggressor %.2f ms avg), "
                       "%d with wrong elements, %ld wrong of %zu per launch", v.name, v.packed ? "packed" : "unpacked", aggr_names[ak], ak ? agrid : 0, rounds, overlapped,
                       v_ms / rounds, a_ms / rounds, wrong_launches, wrong_total, nchk);
                for (int i = 0; i < nfirst; ++i) printf("  [%ld] got %.9g want %.9g", first_idx[i], first_got[i], first_want[i]);
                printf("\n"); fflush(stdout);
                any_bad |= wrong_total ? 1 : 0;
            }
        }
    }
    printf(any_bad ? "RESULT: corruption reproduced\n" : "RESULT: no corruption in any combination\n");
    return 0;
}
