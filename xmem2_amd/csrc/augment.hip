// Deterministic permanent-memory augmentations on the device.
//
// The reference multiplies every annotated frame by a fixed list of (image transform, mask transform) pairs before it enters the
// permanent memory (inference/frame_selection/frame_selection_utils.py:50-218, used at inference/run_on_video.py:231-242 with the
// subset 'best_all').  Its transforms are torchvision objects over PIL (image side) and over the tensor backend (mask side); here
// ONE launch makes all augmented uint8 frames and float masks of an annotation in HBM, from the decoded frame that is already
// there.  Every operation restates the arithmetic of the host path (xmem2_amd/augmentations.py, i.e. PIL 's C kernels and
// torch.grid_sample) step by step in the same precision - this file is compiled with -ffp-contract=off so that no multiply-add
// is fused where the host rounds twice:
//   brightness f   ImageEnhance.Brightness = Image.blend(black, im, f):  t = 0 + f * (p - 0);  <=0 -> 0, >=255 -> 255, else (uint8) t
//   posterize b    ImageOps.posterize: p & ~(2^(8-b) - 1)
//   sharpness f    ImageEnhance.Sharpness = blend(im.filter(SMOOTH), im, f); SMOOTH = 3x3 (1 1 1; 1 5 1; 1 1 1) / 13 in float32,
//                  accumulated from 0.5, truncated and clipped, the one-pixel border copied
//   blur           FT.gaussian_blur(kernel 7): float32 7x7 kernel (outer product of the normalised 1-D pdf, sigma = 0.15 k + 0.35),
//                  reflect padding, round half to even
//   gray           Image.convert('L') (ITU-R 601-2: (19595 R + 38470 G + 7471 B + 0x8000) >> 16) on three channels
//   affine         PIL Image.transform(AFFINE, NEAREST, fill 0): source column / row of output pixel (x, y) = floor of
//                  a0 (x + 0.5) + a1 (y + 0.5) + a2 in float64; PIL accumulates these sums incrementally - for a pure scaling its
//                  per-column / per-row tables are reproduced on the host (exact ties every third pixel at scale 1.5) and passed in.
//                  Masks: torchvision's tensor branch = affine grid in float32 + grid_sample(nearest, zeros, align_corners=False).
#include "common.hpp"
#include <math.h>
#include <vector>

struct AugDev {
    int type;                  // XMEM_AUG_*
    float f;                   // brightness / sharpness factor, posterize bits
    double a[6];               // image side: output pixel centre -> input coordinate (PIL inverse affine matrix)
    float r[6];                // mask side: rescaled theta^T (r00 r10 r20 | r01 r11 r21), see xmem2_amd/augmentations.py affine_tensor
    int table_off;             // >= 0: pure scaling - offset into `tables` of this entry's [W] column and [H] row source indices (-1: out)
};

__device__ __forceinline__ unsigned char blend_clip(float a, float b, float f) {     // Imaging blend with extrapolation
    const float t = a + f * (b - a);
    if (t <= 0.f) return 0;
    if (t >= 255.f) return 255;
    return (unsigned char)t;
}

__device__ __forceinline__ int reflect_idx(int i, int n) {            // F.pad(mode='reflect')
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

__global__ __launch_bounds__(256) void augment_kernel(const unsigned char* __restrict__ img, const float* __restrict__ mask, int H, int W, int K,
                                                      const AugDev* __restrict__ augs, int n_aug, const int* __restrict__ tables,
                                                      const float* __restrict__ gk, unsigned char* __restrict__ out_img,
                                                      float* __restrict__ out_mask) {
    const size_t HW = (size_t)H * W;
    const size_t total = HW * n_aug;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int ai = (int)(e / HW);
        const size_t pix = e - (size_t)ai * HW;
        const int y = (int)(pix / W), x = (int)(pix - (size_t)y * W);
        const AugDev a = augs[ai];
        unsigned char o[3];
        const unsigned char* p = img + pix * 3;
        bool geometric = false;
        int sx = x, sy = y;                                       // source pixel of a geometric transform
        bool inside = true;
        switch (a.type) {
        case XMEM_AUG_BRIGHTNESS:
            for (int c = 0; c < 3; ++c) o[c] = blend_clip(0.f, (float)p[c], a.f);
            break;
        case XMEM_AUG_POSTERIZE: {
            const unsigned char m = (unsigned char)~((1u << (8 - (int)a.f)) - 1u);
            for (int c = 0; c < 3; ++c) o[c] = p[c] & m;
            break;
        }
        case XMEM_AUG_GRAY: {
            const unsigned char L = (unsigned char)((19595u * p[0] + 38470u * p[1] + 7471u * p[2] + 0x8000u) >> 16);
            o[0] = o[1] = o[2] = L;
            break;
        }
        case XMEM_AUG_SHARPNESS:
            for (int c = 0; c < 3; ++c) {
                float deg;
                if (y == 0 || x == 0 || y == H - 1 || x == W - 1) deg = (float)p[c];         // the filter copies the border
                else {
                    float ss = 0.5f;                               // ImagingFilter3x3: rows from the bottom, offset + 0.5 first
                    for (int dy = 1; dy >= -1; --dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            const float kv = ((dy == 0 && dx == 0) ? 5.0f : 1.0f) / 13.0f;
                            ss = ss + (float)img[((size_t)(y + dy) * W + (x + dx)) * 3 + c] * kv;
                        }
                    deg = ss <= 0.f ? 0.f : (ss >= 255.f ? 255.f : (float)(unsigned char)ss);
                }
                o[c] = blend_clip(deg, (float)p[c], a.f);
            }
            break;
        case XMEM_AUG_BLUR7:
            for (int c = 0; c < 3; ++c) {
                float ss = 0.f;
                for (int dy = -3; dy <= 3; ++dy) {
                    const int yy = reflect_idx(y + dy, H);
                    for (int dx = -3; dx <= 3; ++dx)
                        ss = ss + (float)img[((size_t)yy * W + reflect_idx(x + dx, W)) * 3 + c] * gk[(dy + 3) * 7 + (dx + 3)];
                }
                const float r = rintf(ss);                          // torch.round: half to even
                o[c] = r <= 0.f ? 0 : (r >= 255.f ? 255 : (unsigned char)r);
            }
            break;
        default: {                                                  // XMEM_AUG_AFFINE
            geometric = true;
            if (a.table_off >= 0) {
                sx = tables[a.table_off + x];
                sy = tables[a.table_off + W + y];
                inside = sx >= 0 && sy >= 0;
            } else {
                const double xin = (double)x + 0.5, yin = (double)y + 0.5;
                const double fx = a.a[0] * xin + a.a[1] * yin + a.a[2];
                const double fy = a.a[3] * xin + a.a[4] * yin + a.a[5];
                sx = (int)floor(fx); sy = (int)floor(fy);
                inside = fx >= 0.0 && fy >= 0.0 && sx < W && sy < H;
            }
            const unsigned char* q = img + ((size_t)(inside ? sy : 0) * W + (inside ? sx : 0)) * 3;
            for (int c = 0; c < 3; ++c) o[c] = inside ? q[c] : 0;
        }
        }
        unsigned char* op = out_img + ((size_t)ai * HW + pix) * 3;
        op[0] = o[0]; op[1] = o[1]; op[2] = o[2];
        if (out_mask && geometric) {
            // torchvision tensor branch: base grid (x - W/2 + 0.5, y - H/2 + 0.5, 1) . rescaled theta^T in float32, then
            // grid_sample's un-normalisation ((g + 1) * size - 1) / 2 and nearest = round half to even
            const float X = (float)x + (-(float)W * 0.5f + 0.5f), Y = (float)y + (-(float)H * 0.5f + 0.5f);
            const float gx = (X * a.r[0] + Y * a.r[1]) + a.r[2];
            const float gy = (X * a.r[3] + Y * a.r[4]) + a.r[5];
            const float ux = ((gx + 1.f) * (float)W - 1.f) / 2.f, uy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
            const float rx = rintf(ux), ry = rintf(uy);
            const bool ok = rx >= 0.f && ry >= 0.f && rx < (float)W && ry < (float)H;
            const size_t src = ok ? (size_t)(int)ry * W + (int)rx : 0;
            for (int k = 0; k < K; ++k)
                out_mask[((size_t)ai * K + k) * HW + pix] = ok ? mask[(size_t)k * HW + src] : 0.f;
        } else if (out_mask) {
            for (int k = 0; k < K; ++k) out_mask[((size_t)ai * K + k) * HW + pix] = mask[(size_t)k * HW + pix];
        }
    }
}

extern "C" size_t xmem_augment_workspace_bytes(int n_aug, int H, int W) {
    if (n_aug <= 0 || H <= 0 || W <= 0) return 0;
    return align_up((size_t)n_aug * sizeof(AugDev), 256) + align_up((size_t)n_aug * (H + W) * sizeof(int), 256) + 256;
}

extern "C" int xmem_augment_frames(const uint8_t* img, const float* mask, int H, int W, int K, const xmem_aug_desc* descs, int n_aug,
                                   uint8_t* out_img, float* out_mask, void* workspace, size_t workspace_bytes, void* stream) {
    if (!img || !descs || !out_img || n_aug <= 0 || H <= 0 || W <= 0 || K < 0 || (out_mask && (!mask || K == 0))) return XMEM_ERR_BAD_ARG;
    if (n_aug > XMEM_AUG_MAX) return XMEM_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < xmem_augment_workspace_bytes(n_aug, H, W)) return XMEM_ERR_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // host side: descriptors, the Gaussian kernel, PIL's incremental source tables for pure scalings
    static thread_local AugDev h_aug[XMEM_AUG_MAX];
    static thread_local float h_gk[64];
    std::vector<int> h_tab;
    int toff = 0;
    for (int i = 0; i < n_aug; ++i) {
        AugDev& a = h_aug[i];
        a.type = descs[i].type; a.f = descs[i].factor; a.table_off = -1;
        for (int j = 0; j < 6; ++j) { a.a[j] = descs[i].image_matrix[j]; a.r[j] = descs[i].mask_grid[j]; }
        if (a.type < 0 || a.type > XMEM_AUG_AFFINE) return XMEM_ERR_BAD_ARG;
        if (a.type == XMEM_AUG_POSTERIZE && !(a.f >= 1.f && a.f <= 8.f)) return XMEM_ERR_BAD_ARG;
        if (a.type == XMEM_AUG_AFFINE && a.a[1] == 0.0 && a.a[3] == 0.0) {
            // ImagingScaleAffine: xo = a2 + a0 / 2, then += a0 per column (the same for rows): sequential float64 sums
            a.table_off = toff;
            h_tab.resize((size_t)toff + W + H);
            double xo = a.a[2] + a.a[0] * 0.5;
            for (int x = 0; x < W; ++x) { const int xin = xo < 0.0 ? -1 : (int)xo; h_tab[toff + x] = (xin >= 0 && xin < W) ? xin : -1; xo += a.a[0]; }
            double yo = a.a[5] + a.a[4] * 0.5;
            for (int y = 0; y < H; ++y) { const int yin = yo < 0.0 ? -1 : (int)yo; h_tab[toff + W + y] = (yin >= 0 && yin < H) ? yin : -1; yo += a.a[4]; }
            toff += W + H;
        }
    }
    {   // float32 Gaussian, kernel 7: sigma = 0.15 * 7 + 0.35; pdf / sum, outer product (functional_tensor.py _get_gaussian_kernel2d)
        const float sigma = (float)(7 * 0.15 + 0.35);
        float pdf[7], sum = 0.f;
        for (int i = 0; i < 7; ++i) { const float xv = -3.f + (float)i; const float q = xv / sigma; pdf[i] = expf(-0.5f * (q * q)); }
        for (int i = 0; i < 7; ++i) sum = sum + pdf[i];
        for (int i = 0; i < 7; ++i) pdf[i] = pdf[i] / sum;
        for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) h_gk[i * 7 + j] = pdf[i] * pdf[j];
    }
    char* ws = reinterpret_cast<char*>(workspace);
    AugDev* d_aug = reinterpret_cast<AugDev*>(ws);
    int* d_tab = reinterpret_cast<int*>(ws + align_up((size_t)n_aug * sizeof(AugDev), 256));
    float* d_gk = reinterpret_cast<float*>(ws + align_up((size_t)n_aug * sizeof(AugDev), 256) + align_up((size_t)n_aug * (H + W) * sizeof(int), 256));
    // small uploads from pageable host memory, completed before the launch (load-time call: not meant for graph capture)
    if (hipMemcpyAsync(d_aug, h_aug, (size_t)n_aug * sizeof(AugDev), hipMemcpyHostToDevice, s) != hipSuccess) return XMEM_ERR_LAUNCH;
    if (!h_tab.empty() && hipMemcpyAsync(d_tab, h_tab.data(), h_tab.size() * sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess) return XMEM_ERR_LAUNCH;
    if (hipMemcpyAsync(d_gk, h_gk, 49 * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) return XMEM_ERR_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return XMEM_ERR_LAUNCH;      // the host staging arrays are reused / freed after this call
    const size_t total = (size_t)H * W * n_aug;
    int blocks = (int)((total + 255) / 256); if (blocks > 32768) blocks = 32768;
    hipLaunchKernelGGL(augment_kernel, dim3(blocks), dim3(256), 0, s, img, mask, H, W, K, d_aug, n_aug, d_tab, d_gk, out_img, out_mask);
    return xmem_check_launch();
}
