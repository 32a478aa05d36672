// Second stage of the packed-f32 investigation: the LIBRARY's own kernels through the C ABI, no torch, private hipMalloc buffers.
// Two builds of the library are loaded side by side: libxmem_hip.so (shipped: -packed-fp32-ops) and libxmem_hip_pk.so (the same
// sources with the default gfx950 feature set, i.e. v_pk_*_f32 allowed).  Victim = xmem_upsample2x_add at the decoder's shape on
// stream 1; aggressor = xmem_conv2d_nhwc in split-operand arithmetic (arith = 1: v_mfma_f32_32x32x16_f16) on stream 2 - the pair
// tools/probes/hog_probe.py ran inside the whole pipeline.  2 x 2: {victim build} x {aggressor build}; every victim launch is
// compared bit for bit with its solo result.
//
//   bash tools/probes/pk_hazard/build.sh lib && tools/probes/pk_hazard/lib_probe
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include "xmem_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef int (*conv_fn)(const xmem_conv_desc*, void*, size_t, void*);
typedef size_t (*convws_fn)(const xmem_conv_desc*);
typedef int (*up_fn)(const float*, const float*, float*, int, int, int, int, void*);

struct Lib { void* h; conv_fn conv; convws_fn convws; up_fn up; const char* tag; };

static Lib open_lib(const std::string& path, const char* tag) {
    Lib l; l.tag = tag;
    l.h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!l.h) { fprintf(stderr, "dlopen %s: %s\n", path.c_str(), dlerror()); exit(2); }
    l.conv = (conv_fn)dlsym(l.h, "xmem_conv2d_nhwc"); l.convws = (convws_fn)dlsym(l.h, "xmem_conv2d_workspace_bytes");
    l.up = (up_fn)dlsym(l.h, "xmem_upsample2x_add");
    int (*ver)(void) = (int (*)(void))dlsym(l.h, "xmem_version");
    if (!ver || ver() != XMEM_ABI_VERSION) { fprintf(stderr, "%s: ABI version mismatch (rebuild: bash tools/probes/pk_hazard/build.sh lib)\n", path.c_str()); exit(2); }
    if (!l.conv || !l.convws || !l.up) { fprintf(stderr, "missing symbols in %s\n", path.c_str()); exit(2); }
    return l;
}

static unsigned long long rs = 0x1234567ull;
static float urand() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (float)((rs >> 40) * (1.0 / 8388608.0) - 1.0); }

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 80;
    std::string dir = argv[0]; dir = dir.substr(0, dir.find_last_of('/'));
    Lib libs[2] = {open_lib(dir + "/../../../xmem2_amd/csrc/libxmem_hip.so", "unpacked (shipped)"), open_lib(dir + "/libxmem_hip_pk.so", "packed")};
    // victim: decoder upsample 1/8 -> 1/4 at 480p
    const int h = 60, w = 108, C = 256;
    const size_t ng = (size_t)h * w * C, nout = ng * 4;
    std::vector<float> hg(ng), hs(nout);
    for (auto& v : hg) v = urand();
    for (auto& v : hs) v = urand();
    float *g, *skip, *out;
    CK(hipMalloc(&g, ng * 4)); CK(hipMalloc(&skip, nout * 4)); CK(hipMalloc(&out, nout * 4));
    CK(hipMemcpy(g, hg.data(), ng * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(skip, hs.data(), nout * 4, hipMemcpyHostToDevice));
    // aggressor: pointwise 64 -> 256 at 4 x 120 x 216 in split arithmetic (hog_probe.py's shape), 12 launches per round
    const int B = 4, H = 120, W = 216, Cin = 64, Cout = 256;
    const size_t npix = (size_t)B * H * W;
    std::vector<float> hx(npix * Cin), hw((size_t)Cout * Cin), ones(Cout, 1.f), zeros(Cout, 0.f);
    for (auto& v : hx) v = urand();
    for (auto& v : hw) v = 0.1f * urand();
    std::vector<_Float16> hwsp((size_t)Cout * Cin * 2);            // [Cout][Cin/4][hi x 4 | lo x 4]
    for (int n = 0; n < Cout; ++n)
        for (int c4 = 0; c4 < Cin / 4; ++c4)
            for (int j = 0; j < 4; ++j) {
                const float v = hw[(size_t)n * Cin + c4 * 4 + j];
                const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                hwsp[((size_t)n * (Cin / 4) + c4) * 8 + j] = hi; hwsp[((size_t)n * (Cin / 4) + c4) * 8 + 4 + j] = lo;
            }
    float *dx, *dw, *dsc, *dsh, *dy; void* dwsp;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dsc, Cout * 4)); CK(hipMalloc(&dsh, Cout * 4));
    CK(hipMalloc(&dy, npix * Cout * 4)); CK(hipMalloc(&dwsp, hwsp.size() * 2));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsc, ones.data(), Cout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsh, zeros.data(), Cout * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwsp, hwsp.data(), hwsp.size() * 2, hipMemcpyHostToDevice));
    xmem_conv_desc d; memset(&d, 0, sizeof d);
    d.in = dx; d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.ldin = Cin; d.w = dw; d.Cout = Cout; d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0;
    d.scale = dsc; d.shift = dsh; d.out = dy; d.ldout = Cout; d.plan_tile = 3; d.plan_splitk = 1; d.arith = 1; d.w_split = dwsp;
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t ea; CK(hipEventCreate(&ea));
    std::vector<float> want(nout), got(nout);
    int any = 0;
    for (int vi = 0; vi < 2; ++vi) {
        const Lib& V = libs[vi];
        CK(hipMemset(out, 0, nout * 4));
        V.up(g, skip, out, 1, h, w, C, s1); CK(hipStreamSynchronize(s1));
        CK(hipMemcpy(want.data(), out, nout * 4, hipMemcpyDeviceToHost));
        for (int ai = -1; ai < 2; ++ai) {
            for (int arith = 1; arith >= 0; --arith) {
                if (ai < 0 && arith == 0) continue;
                long wrong_total = 0; int wrong_launches = 0, overlapped = 0;
                for (int r = 0; r < rounds; ++r) {
                    CK(hipMemsetAsync(out, 0, nout * 4, s1)); CK(hipStreamSynchronize(s1));
                    if (ai >= 0) {
                        d.arith = arith;
                        for (int k = 0; k < 12; ++k) { const int rc = libs[ai].conv(&d, nullptr, 0, s2); if (rc) { fprintf(stderr, "aggressor conv rc %d\n", rc); return 2; } }
                        CK(hipEventRecord(ea, s2));
                    }
                    for (int k = 0; k < 4; ++k) V.up(g, skip, out, 1, h, w, C, s1);       // the last launch's output is checked
                    CK(hipStreamSynchronize(s1));
                    if (ai >= 0) { overlapped += hipEventQuery(ea) == hipErrorNotReady ? 1 : 0; CK(hipStreamSynchronize(s2)); }
                    CK(hipMemcpy(got.data(), out, nout * 4, hipMemcpyDeviceToHost));
                    long wrong = 0;
                    for (size_t i = 0; i < nout; ++i) { unsigned a, b; memcpy(&a, &got[i], 4); memcpy(&b, &want[i], 4); wrong += a != b; }
                    wrong_total += wrong; wrong_launches += wrong ? 1 : 0;
                }
                printf("victim upsample2x_add [%-18s] | aggressor %-44s: %3d rounds (%3d finished while the aggressor ran), %d with wrong elements, %ld wrong of %zu\n",
                       V.tag, ai < 0 ? "none" : (std::string(arith ? "split-operand conv fp16 MFMA [" : "fp32 MFMA conv [") + libs[ai].tag + "]").c_str(),
                       rounds, overlapped, wrong_launches, wrong_total, nout);
                fflush(stdout);
                any |= wrong_total ? 1 : 0;
            }
        }
    }
    printf(any ? "RESULT: corruption reproduced through the library\n" : "RESULT: no corruption in any combination\n");
    return 0;
}
