// Standalone driver of xmem_conv2d_nhwc through the C ABI (no torch, no python: starts in milliseconds on a fresh GPU box).
// Times convolution plans per layer shape with HIP events and checks every plan against the direct implicit-GEMM plan.
//
//   hipcc -O2 -std=c++17 tools/conv_bench.cpp -I include -L xmem2_amd/csrc -lxmem_hip -Wl,-rpath,'$ORIGIN/../xmem2_amd/csrc' -o tools/conv_bench
//   tools/conv_bench [-n iters] [-r relu_in,res,relu_out] "B H W Cin Cout" plan[,plan...] ["B H W Cin Cout" plan,... ]
//
// Shapes are 3x3 / stride 1 / pad 1 layers (the Winograd-eligible ones); plan = plan_tile of xmem_conv_desc (3 = direct 64x64,
// 9 = F(2x2) + 64x64 GEMM, 19 = F(4x4) + 64x64 GEMM, ...).  Output: one line per (shape, plan): microseconds per call, the
// DIRECT form's TFLOP/s, executed-MFMA TFLOP/s where it differs, max |y - y_direct| / max |y_direct|.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <algorithm>
#include "xmem_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static inline float urand() {          // uniform [-1, 1)
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (float)((rng_state >> 40) * (1.0 / 8388608.0) - 1.0);
}

static void winograd_weights(const std::vector<float>& w, int Cout, int Cin, const double* G, int T, std::vector<float>& u) {
    // u[(i*T+j)][n][c] = sum_ab G[i][a] w[n][a][b][c] G[j][b]   (formed in fp64, rounded once)
    u.assign((size_t)T * T * Cout * Cin, 0.f);
    for (int n = 0; n < Cout; ++n)
        for (int c = 0; c < Cin; ++c) {
            double g[3][3];
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) g[a][b] = w[(((size_t)n * 3 + a) * 3 + b) * Cin + c];
            for (int i = 0; i < T; ++i)
                for (int j = 0; j < T; ++j) {
                    double s = 0;
                    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) s += G[i * 3 + a] * g[a][b] * G[j * 3 + b];
                    u[((size_t)(i * T + j) * Cout + n) * Cin + c] = (float)s;
                }
        }
}

int main(int argc, char** argv) {
    int iters = 20, relu_in = 0, use_res = 0, relu_out = 0, ref_plan = 3;
    std::vector<std::pair<std::string, std::string>> jobs;
    for (int a = 1; a < argc; ++a) {
        if (!strcmp(argv[a], "-n") && a + 1 < argc) { iters = atoi(argv[++a]); continue; }
        if (!strcmp(argv[a], "-ref") && a + 1 < argc) { ref_plan = atoi(argv[++a]); continue; }
        if (!strcmp(argv[a], "-r") && a + 1 < argc) { sscanf(argv[++a], "%d,%d,%d", &relu_in, &use_res, &relu_out); continue; }
        if (a + 1 < argc) { jobs.push_back({argv[a], argv[a + 1]}); ++a; }
    }
    if (jobs.empty()) { fprintf(stderr, "usage: conv_bench [-n iters] [-r relu_in,res,relu_out] \"B H W Cin Cout\" plan[,plan] ...\n"); return 1; }
    static const double G2[12] = {1, 0, 0, 0.5, 0.5, 0.5, 0.5, -0.5, 0.5, 0, 0, 1};
    // F(4x4) with the points {0, +-3/4, +-3/2, inf} (csrc/conv_mfma.hip): rows (1, p, p^2) / N_p, N_p = prod (p - p_k)
    static const double G4[18] = {64.0 / 81, 0, 0, -128.0 / 243, -32.0 / 81, -8.0 / 27, -128.0 / 243, 32.0 / 81, -8.0 / 27,
                                  32.0 / 243, 16.0 / 81, 8.0 / 27, 32.0 / 243, -16.0 / 81, 8.0 / 27, 0, 0, 1};
    if (xmem_version() != XMEM_ABI_VERSION) { fprintf(stderr, "libxmem_hip.so has ABI version %d, this driver was built for %d: rebuild tools/conv_bench\n", xmem_version(), XMEM_ABI_VERSION); return 2; }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& job : jobs) {
        int B, H, W, Cin, Cout, KS = 3, stride = 1;
        if (sscanf(job.first.c_str(), "%d %d %d %d %d %d %d", &B, &H, &W, &Cin, &Cout, &KS, &stride) < 5) { fprintf(stderr, "bad shape '%s'\n", job.first.c_str()); return 1; }
        const int pad = KS / 2, Ho = (H + 2 * pad - KS) / stride + 1, Wo = (W + 2 * pad - KS) / stride + 1;
        const size_t npix_in = (size_t)B * H * W, npix = (size_t)B * Ho * Wo;
        std::vector<float> x(npix_in * Cin), w((size_t)Cout * KS * KS * Cin), sc(Cout), sh(Cout), res(npix * Cout);
        for (auto& v : x) v = urand();
        const float ws = 1.f / sqrtf((float)(KS * KS) * Cin);
        for (auto& v : w) v = urand() * ws * 1.7f;
        for (auto& v : sc) v = 1.f + 0.25f * urand();
        for (auto& v : sh) v = 0.1f * urand();
        for (auto& v : res) v = urand();
        std::vector<float> u2(4, 0.f), u4(4, 0.f);
        if (KS == 3 && stride == 1) { winograd_weights(w, Cout, Cin, G2, 4, u2); winograd_weights(w, Cout, Cin, G4, 6, u4); }
        float *dx, *dw, *dsc, *dsh, *dres, *du2, *du4, *dout, *dref;
        CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&dsc, Cout * 4)); CK(hipMalloc(&dsh, Cout * 4));
        CK(hipMalloc(&dres, res.size() * 4)); CK(hipMalloc(&du2, u2.size() * 4)); CK(hipMalloc(&du4, u4.size() * 4));
        CK(hipMalloc(&dout, npix * Cout * 4)); CK(hipMalloc(&dref, npix * Cout * 4));
        CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsc, sc.data(), Cout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsh, sh.data(), Cout * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dres, res.data(), res.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(du2, u2.data(), u2.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(du4, u4.data(), u4.size() * 4, hipMemcpyHostToDevice));
        xmem_conv_desc d; memset(&d, 0, sizeof d);
        d.in = dx; d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.ldin = Cin;
        d.w = dw; d.Cout = Cout; d.KH = KS; d.KW = KS; d.stride = stride; d.pad = pad;
        d.scale = dsc; d.shift = dsh; d.res = use_res ? dres : nullptr; d.ldres = Cout; d.ldout = Cout;
        d.relu_in = relu_in; d.relu_out = relu_out; d.plan_splitk = 1;
        if (KS == 3 && stride == 1) { d.w_winograd = du2; d.w_winograd4 = du4; }
        std::vector<float> yref(npix * Cout), y(npix * Cout);
        auto run = [&](int plan, float* out, int n) -> float {
            d.plan_tile = plan; d.out = out;
            const size_t need = xmem_conv2d_workspace_bytes(&d);
            void* wsp = nullptr;
            if (need) CK(hipMalloc(&wsp, need));
            int rc = 0;
            for (int i = 0; i < 3 && !rc; ++i) rc = xmem_conv2d_nhwc(&d, wsp, need, st);
            if (rc) { fprintf(stderr, "plan %d: xmem_conv2d_nhwc -> %d (%s)\n", plan, rc, xmem_last_error_string(rc)); if (wsp) CK(hipFree(wsp)); return -1.f; }
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < n; ++i) xmem_conv2d_nhwc(&d, wsp, need, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (wsp) CK(hipFree(wsp));
            return ms * 1000.f / n;
        };
        run(ref_plan, dref, 1);
        CK(hipMemcpy(yref.data(), dref, yref.size() * 4, hipMemcpyDeviceToHost));
        double amax = 0; for (float v : yref) amax = std::max(amax, (double)fabsf(v));
        const double gflop = 2.0 * npix * Cout * (double)(KS * KS) * Cin * 1e-9;
        char* plans = strdup(job.second.c_str());
        for (char* tok = strtok(plans, ","); tok; tok = strtok(nullptr, ",")) {
            const int plan = atoi(tok);
            CK(hipMemset(dout, 0xff, npix * Cout * 4));
            const float us = run(plan, dout, iters);
            if (us < 0) continue;
            CK(hipMemcpy(y.data(), dout, y.size() * 4, hipMemcpyDeviceToHost));
            double err = 0; size_t bad = 0;
            for (size_t i = 0; i < y.size(); ++i) { const double e = fabs((double)y[i] - yref[i]); if (!(e <= 1e30)) ++bad; else err = std::max(err, e); }
            const double div = (plan >= 35 || KS != 3) ? 1.0 : ((plan >= 17 && plan <= 28) ? 4.0 : (plan >= 7 ? 2.25 : 1.0));
            printf("shape %-22s r%d%d%d plan %2d  %8.1f us  %7.1f TF direct-form  %6.1f TF executed  err %.2e%s\n", job.first.c_str(), relu_in, use_res, relu_out,
                   plan, us, gflop / us * 1e3, gflop / div / us * 1e3, err / amax, bad ? "  NON-FINITE OUTPUT" : "");
            fflush(stdout);
        }
        free(plans);
        CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dsc)); CK(hipFree(dsh)); CK(hipFree(dres)); CK(hipFree(du2)); CK(hipFree(du4)); CK(hipFree(dout)); CK(hipFree(dref));
    }
    return 0;
}
