// HBM-bound kernels of the encoders / decoder: pooling, resampling, CBAM, gates, layout packing,
// soft aggregation and argmax.  All NHWC fp32, float4-vectorised over channels (C % 4 == 0), coalesced:
// consecutive lanes walk consecutive channels of one pixel.  Reference call sites: include/xmem_hip.h.
#include "common.hpp"
#include <math.h>

namespace {
inline int grid_for(size_t n, int block = 256, int cap = 8192) {
    size_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g > (size_t)cap ? cap : g);
}
}  // namespace

// Storage types of the fp16 loop (config['precision'] = 'fp16'): activations may live in HBM as IEEE halfs; every kernel below
// computes in fp32 and converts at its loads / stores (one rounding to nearest even per stored value).  The `_t` entry points take
// the storage type of each tensor as a flag (0 = float, 1 = half); the plain entry points are the fp32 instantiations.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld4<_Float16>(const _Float16* p) {
    const h16x4 h = *reinterpret_cast<const h16x4*>(p);
    f32x4 v = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    return v;
}
template <typename T> __device__ __forceinline__ void st4(T* p, const f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st4<_Float16>(_Float16* p, const f32x4 v) {
    h16x4 h; h.x = (_Float16)v.x; h.y = (_Float16)v.y; h.z = (_Float16)v.z; h.w = (_Float16)v.w;
    *reinterpret_cast<h16x4*>(p) = h;
}
// dispatch on two storage flags: F(TI, TO)
#define XMEM_DISPATCH_IO(in_half, out_half, F)                          \
    do {                                                                \
        if (in_half) { if (out_half) { F(_Float16, _Float16); } else { F(_Float16, float); } } \
        else { if (out_half) { F(float, _Float16); } else { F(float, float); } }               \
    } while (0)

// ---------------------------------------------------------------------------------------------
// max pool 3x3 / stride 2 / pad 1  (implicit -inf padding like nn.MaxPool2d)
// ---------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void maxpool3x3s2_kernel(const TI* __restrict__ in, TO* __restrict__ out,
                                    int B, int H, int W, int C, int Ho, int Wo) {
    const int C4 = C >> 2;
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % C4);
        size_t pix = e / C4;
        const int ow = (int)(pix % Wo); pix /= Wo;
        const int oh = (int)(pix % Ho);
        const int b = (int)(pix / Ho);
        // nine UNCONDITIONAL loads from clamped coordinates (a clamped tap repeats an in-range neighbour of the same window: the max
        // is unchanged): a load behind a bounds branch waits for its predecessor - nine serialised round trips (round 6)
        f32x4 v[3][3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int ih = min(max(oh * 2 - 1 + dy, 0), H - 1);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int iw = min(max(ow * 2 - 1 + dx, 0), W - 1);
                v[dy][dx] = ld4(in + (((size_t)b * H + ih) * W + iw) * C + c4 * 4);
            }
        }
        f32x4 m = v[1][1];                          // the centre tap (2 oh, 2 ow) is always inside
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                m.x = fmaxf(m.x, v[dy][dx].x); m.y = fmaxf(m.y, v[dy][dx].y); m.z = fmaxf(m.z, v[dy][dx].z); m.w = fmaxf(m.w, v[dy][dx].w);
            }
        st4(out + (((size_t)b * Ho + oh) * Wo + ow) * C + c4 * 4, m);
    }
}

extern "C" int xmem_maxpool3x3s2_t(const void* in, int in_half, void* out, int out_half, int B, int H, int W, int C, void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0) return XMEM_ERR_BAD_ARG;
    if (C % 4) return XMEM_ERR_UNSUPPORTED;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)B * Ho * Wo * (C / 4);
#define F(TI, TO) hipLaunchKernelGGL((maxpool3x3s2_kernel<TI, TO>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const TI*)in, (TO*)out, B, H, W, C, Ho, Wo)
    XMEM_DISPATCH_IO(in_half, out_half, F);
#undef F
    return xmem_check_launch();
}
extern "C" int xmem_maxpool3x3s2(const float* in, float* out, int B, int H, int W, int C, void* stream) {
    return xmem_maxpool3x3s2_t(in, 0, out, 0, B, H, W, C, stream);
}

// ---------------------------------------------------------------------------------------------
// bilinear x2 (align_corners=False) + broadcast skip add
// source index as ATen's area_pixel_compute_source_index: src = 0.5*(dst+0.5)-0.5, clamped at 0
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bilinear_src(int dst, float scale, int size, int& i0, int& i1, float& l1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > size - 1) i0 = size - 1;
    i1 = i0 + (i0 < size - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

template <typename T>
__global__ void upsample2x_add_kernel(const T* __restrict__ g, const T* __restrict__ skip,
                                      T* __restrict__ out, int B, int h, int w, int C) {
    const int C4 = C >> 2, H = 2 * h, W = 2 * w;
    const size_t total = (size_t)B * H * W * C4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % C4);
        size_t pix = e / C4;
        const int x = (int)(pix % W); pix /= W;
        const int y = (int)(pix % H);
        const int b = (int)(pix / H);
        int y0, y1, x0, x1; float ly, lx;
        bilinear_src(y, 0.5f, h, y0, y1, ly);
        bilinear_src(x, 0.5f, w, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const T* gb = g + (size_t)b * h * w * C + c4 * 4;
        const f32x4 v00 = ld4(gb + ((size_t)y0 * w + x0) * C);
        const f32x4 v01 = ld4(gb + ((size_t)y0 * w + x1) * C);
        const f32x4 v10 = ld4(gb + ((size_t)y1 * w + x0) * C);
        const f32x4 v11 = ld4(gb + ((size_t)y1 * w + x1) * C);
        const f32x4 s = ld4(skip + ((size_t)y * W + x) * C + c4 * 4);
        f32x4 o;
        o.x = s.x + (hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x));
        o.y = s.y + (hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y));
        o.z = s.z + (hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z));
        o.w = s.w + (hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w));
        st4(out + (((size_t)b * H + y) * W + x) * C + c4 * 4, o);
    }
}

extern "C" int xmem_upsample2x_add_t(const void* g, const void* skip, void* out, int half, int B, int h, int w, int C, void* stream) {
    if (!g || !skip || !out || B <= 0 || h <= 0 || w <= 0 || C <= 0) return XMEM_ERR_BAD_ARG;
    if (C % 4) return XMEM_ERR_UNSUPPORTED;
    const size_t total = (size_t)B * 4 * h * w * (C / 4);
    if (half) hipLaunchKernelGGL(upsample2x_add_kernel<_Float16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                                 (const _Float16*)g, (const _Float16*)skip, (_Float16*)out, B, h, w, C);
    else hipLaunchKernelGGL(upsample2x_add_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                            (const float*)g, (const float*)skip, (float*)out, B, h, w, C);
    return xmem_check_launch();
}
extern "C" int xmem_upsample2x_add(const float* g, const float* skip, float* out, int B, int h, int w, int C, void* stream) {
    return xmem_upsample2x_add_t(g, skip, out, 0, B, h, w, C, stream);
}

// ---------------------------------------------------------------------------------------------
// area (average) downsample by integer ratio
// ---------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void area_down_kernel(const TI* __restrict__ in, int ldin, TO* __restrict__ out, int ldout,
                                 int B, int H, int W, int C, int r) {
    const int Ho = H / r, Wo = W / r;
    const size_t total = (size_t)B * Ho * Wo * C;
    const float inv = 1.0f / (float)(r * r);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        size_t pix = e / C;
        const int ow = (int)(pix % Wo); pix /= Wo;
        const int oh = (int)(pix % Ho);
        const int b = (int)(pix / Ho);
        float s = 0.f;
        for (int dy = 0; dy < r; ++dy)
            for (int dx = 0; dx < r; ++dx)
                s += (float)in[(((size_t)b * H + oh * r + dy) * W + ow * r + dx) * ldin + c];
        out[(((size_t)b * Ho + oh) * Wo + ow) * ldout + c] = (TO)(s * inv);
    }
}

extern "C" int xmem_area_downsample_t(const void* in, int in_half, int ldin, void* out, int out_half, int ldout, int B, int H, int W, int C, int r, void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || r <= 0) return XMEM_ERR_BAD_ARG;
    if (H % r || W % r) return XMEM_ERR_UNSUPPORTED;
    const size_t total = (size_t)B * (H / r) * (W / r) * C;
#define F(TI, TO) hipLaunchKernelGGL((area_down_kernel<TI, TO>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const TI*)in, ldin, (TO*)out, ldout, B, H, W, C, r)
    XMEM_DISPATCH_IO(in_half, out_half, F);
#undef F
    return xmem_check_launch();
}
extern "C" int xmem_area_downsample(const float* in, int ldin, float* out, int ldout, int B, int H, int W, int C, int r, void* stream) {
    return xmem_area_downsample_t(in, 0, ldin, out, 0, ldout, B, H, W, C, r, stream);
}

// ---------------------------------------------------------------------------------------------
// channel-slice copy with object broadcast (concat builder)
// ---------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void copy_channels_kernel(const TI* __restrict__ src, int ldsrc, int srcB, TO* __restrict__ dst, int lddst,
                                     int B, int P, int C4) {
    const size_t total = (size_t)B * P * C4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % C4);
        size_t r = e / C4;
        const int pidx = (int)(r % P);
        const int b = (int)(r / P);
        const f32x4 v = ld4(src + ((size_t)(b % srcB) * P + pidx) * ldsrc + c4 * 4);
        st4(dst + ((size_t)b * P + pidx) * lddst + c4 * 4, v);
    }
}

extern "C" int xmem_copy_channels_t(const void* src, int src_half, int ldsrc, int srcB, void* dst, int dst_half, int lddst, int B, int P, int C, void* stream) {
    if (!src || !dst || B <= 0 || P <= 0 || C <= 0 || srcB <= 0) return XMEM_ERR_BAD_ARG;
    if (C % 4 || ldsrc % 4 || lddst % 4 || ((uintptr_t)src & (src_half ? 7 : 15)) || ((uintptr_t)dst & (dst_half ? 7 : 15))) return XMEM_ERR_UNSUPPORTED;
    const size_t total = (size_t)B * P * (C / 4);
#define F(TI, TO) hipLaunchKernelGGL((copy_channels_kernel<TI, TO>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const TI*)src, ldsrc, srcB, (TO*)dst, lddst, B, P, C / 4)
    XMEM_DISPATCH_IO(src_half, dst_half, F);
#undef F
    return xmem_check_launch();
}
extern "C" int xmem_copy_channels(const float* src, int ldsrc, int srcB, float* dst, int lddst, int B, int P, int C, void* stream) {
    return xmem_copy_channels_t(src, 0, ldsrc, srcB, dst, 0, lddst, B, P, C, stream);
}

// ---------------------------------------------------------------------------------------------
// HiddenUpdater input in ONE launch (model/modules.py:49-57, group_modules.py:15-23):
//   out[k][y][x] = [ g16 | area2(g8) | area4(g4) | area4(logits) ]      (channels; the caller keeps what follows them zero)
// the concatenated input of the fused g16 / g8 / g4 pointwise convolution.  It used to be four launches per frame (one channel copy, three
// area downsamples of one float per thread: 9-15 us each in the stream for 36 MB of reads).  One float4 group of the output per thread; the
// window sums run in area_down_kernel's order (rows, then columns, then one multiply by 1 / r^2): the same bits.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hidden_gather_kernel(const float* __restrict__ g16, int c16, const float* __restrict__ g8, int c8,
                                                            const float* __restrict__ g4, int c4, const float* __restrict__ logits,
                                                            float* __restrict__ out, int ldout, int K, int h, int w) {
    const int n16 = c16 >> 2, n8 = c8 >> 2, n4 = c4 >> 2, G = n16 + n8 + n4 + 1;
    const size_t total = (size_t)K * h * w * G;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        int g = (int)(e % G);
        size_t pix = e / G;
        const int x = (int)(pix % w); const size_t ky = pix / w;          // ky = k * h + y
        float* const o = out + pix * ldout;
        if (g < n16) {
            st4(o + 4 * g, ld4(g16 + pix * c16 + 4 * g));
        } else if (g < n16 + n8) {
            g -= n16;
            const float* src = g8 + ((2 * ky) * (size_t)(2 * w) + 2 * x) * c8 + 4 * g;
            const size_t row = (size_t)(2 * w) * c8;
            const f32x4 a = ld4(src), b = ld4(src + c8), c = ld4(src + row), d = ld4(src + row + c8);
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            s += a; s += b; s += c; s += d;
            st4(o + c16 + 4 * g, s * 0.25f);
        } else if (g < n16 + n8 + n4) {
            g -= n16 + n8;
            const float* src = g4 + ((4 * ky) * (size_t)(4 * w) + 4 * x) * c4 + 4 * g;
            const size_t row = (size_t)(4 * w) * c4;
            f32x4 v[16];
#pragma unroll
            for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                for (int dx = 0; dx < 4; ++dx) v[4 * dy + dx] = ld4(src + dy * row + (size_t)dx * c4);
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 16; ++i) s += v[i];
            st4(o + c16 + c8 + 4 * g, s * 0.0625f);
        } else {
            const float* src = logits + (4 * ky) * (size_t)(4 * w) + 4 * x;      // one channel: a pixel row is 4 w floats
            float s = 0.f;
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) {
                const f32x4 r = ld4(src + (size_t)dy * (4 * w));                  // 4 x is a multiple of 4 floats: aligned
                s += r.x; s += r.y; s += r.z; s += r.w;
            }
            o[c16 + c8 + c4] = s * 0.0625f;
        }
    }
}

extern "C" int xmem_hidden_update_gather(const float* g16, int c16, const float* g8, int c8, const float* g4, int c4, const float* logits,
                                         float* out, int ldout, int K, int h, int w, void* stream) {
    if (!g16 || !g8 || !g4 || !logits || !out || K <= 0 || h <= 0 || w <= 0 || c16 <= 0 || c8 <= 0 || c4 <= 0) return XMEM_ERR_BAD_ARG;
    if (c16 % 4 || c8 % 4 || c4 % 4 || ldout % 4 || ldout < c16 + c8 + c4 + 1) return XMEM_ERR_UNSUPPORTED;
    if (((uintptr_t)g16 | (uintptr_t)g8 | (uintptr_t)g4 | (uintptr_t)logits | (uintptr_t)out) & 15) return XMEM_ERR_UNSUPPORTED;
    const size_t total = (size_t)K * h * w * ((c16 + c8 + c4) / 4 + 1);
    hipLaunchKernelGGL(hidden_gather_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, g16, c16, g8, c8, g4, c4, logits, out,
                       ldout, K, h, w);
    return xmem_check_launch();
}

// ---------------------------------------------------------------------------------------------
// CBAM + residual:  out = g + SpatialGate(ChannelGate(g))
//   1. channel_pool:   avg / max over the P pixels per (b, c)            -> pooled partials   (kernel 1: needs all pixels)
//   2. channel_mlp:    cscale = sigmoid(mlp(avg) + mlp(max))
//   3. compress:       per pixel max / mean over c of g*cscale
//   4. spatial+apply:  sg = sigmoid(conv7x7(comp)); out = g + (g*cscale)*sg             (2-3: kernel 2, 4: kernel 3; 16 pixels per workgroup)
// ---------------------------------------------------------------------------------------------
#define CBAM_PSPLIT 16
template <typename T>
__global__ void cbam_channel_pool_kernel(const T* __restrict__ g, float* __restrict__ partial, int P, int C) {
    // block = 256 threads = 64 channels x 4 pixel stripes; grid = (C/64, B, CBAM_PSPLIT); partial [B][PSPLIT][2][C]
    __shared__ float ssum[4][64];
    __shared__ float smax[4][64];
    const int cl = threadIdx.x & 63, stripe = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, b = blockIdx.y, ps = blockIdx.z;
    const int per = (P + CBAM_PSPLIT - 1) / CBAM_PSPLIT;
    const int p0 = ps * per, p1 = min(P, p0 + per);
    float s = 0.f, m = -INFINITY;
    if (c < C) {
        const T* gb = g + (size_t)b * P * C + c;
#pragma unroll 8                                     // eight pixels' loads in flight together (same order of additions)
        for (int pidx = p0 + stripe; pidx < p1; pidx += 4) {
            const float v = (float)gb[(size_t)pidx * C];
            s += v; m = fmaxf(m, v);
        }
    }
    ssum[stripe][cl] = s; smax[stripe][cl] = m;
    __syncthreads();
    if (stripe == 0 && c < C) {
        const float ts = (ssum[0][cl] + ssum[1][cl]) + (ssum[2][cl] + ssum[3][cl]);
        const float tm = fmaxf(fmaxf(smax[0][cl], smax[1][cl]), fmaxf(smax[2][cl], smax[3][cl]));
        partial[(((size_t)b * CBAM_PSPLIT + ps) * 2 + 0) * C + c] = ts;
        partial[(((size_t)b * CBAM_PSPLIT + ps) * 2 + 1) * C + c] = tm;
    }
}

// Steps 2-4 in two kernels of many small workgroups (16 pixels each): every workgroup recomputes the channel MLP from the
// pooled partials (512 -> 32 -> 512: 64 K MACs, L2-resident operands) instead of waiting for a one-workgroup MLP launch
// (41 us of latency for ~1 us of arithmetic), then (a) compresses its pixels over the channels; (b) after the compressed map
// is complete, takes the 7x7 gate of its pixels and applies the gated residual.
#define CBAM_PIX 16
__device__ __forceinline__ void cbam_channel_scale(const float* __restrict__ partial, const float* __restrict__ w1,
                                                   const float* __restrict__ b1, const float* __restrict__ w2,
                                                   const float* __restrict__ b2, int b, int P, int C, int Cr,
                                                   float* pb, float* hid, float* cs) {
    // latency-bound: every phase issues all of its independent loads before reducing
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += 256) {
        float sv[CBAM_PSPLIT], mv[CBAM_PSPLIT];
#pragma unroll
        for (int ps = 0; ps < CBAM_PSPLIT; ++ps) {
            sv[ps] = partial[(((size_t)b * CBAM_PSPLIT + ps) * 2 + 0) * C + c];
            mv[ps] = partial[(((size_t)b * CBAM_PSPLIT + ps) * 2 + 1) * C + c];
        }
        float s = 0.f, m = -INFINITY;
#pragma unroll
        for (int ps = 0; ps < CBAM_PSPLIT; ++ps) { s += sv[ps]; m = fmaxf(m, mv[ps]); }
        pb[c] = s / (float)P; pb[C + c] = m;
    }
    __syncthreads();
    // hidden layer: wave w owns units j = w, w+4, ... for BOTH pooled vectors (the weight row is loaded once); up to 8 units
    // per wave are accumulated side by side so that their weight loads are all in flight together
    for (int j0 = wave; j0 < Cr; j0 += 32) {
        float acc0[8], acc1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc0[u] = 0.f; acc1[u] = 0.f; }
#pragma unroll 4                                     // four iterations' weight loads (32) in flight together, not eight after eight
        for (int c = lane; c < C; c += 64) {
            const float p0 = pb[c], p1 = pb[C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + 4 * u;
                const float wv = j < Cr ? w1[(size_t)j * C + c] : 0.f;
                acc0[u] += p0 * wv; acc1[u] += p1 * wv;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + 4 * u;
            const float s0 = wave_sum(acc0[u]), s1 = wave_sum(acc1[u]);
            if (lane == 0 && j < Cr) { hid[j] = fmaxf(s0 + b1[j], 0.f); hid[Cr + j] = fmaxf(s1 + b1[j], 0.f); }
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float sa = 0.f, smx = 0.f;
        const float* wr = w2 + (size_t)c * Cr;
        if ((Cr & 3) == 0) {
#pragma unroll 8
            for (int j = 0; j < Cr; j += 4) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(wr + j);
                sa += hid[j] * w.x; smx += hid[Cr + j] * w.x;
                sa += hid[j + 1] * w.y; smx += hid[Cr + j + 1] * w.y;
                sa += hid[j + 2] * w.z; smx += hid[Cr + j + 2] * w.z;
                sa += hid[j + 3] * w.w; smx += hid[Cr + j + 3] * w.w;
            }
        } else {
            for (int j = 0; j < Cr; ++j) { sa += hid[j] * wr[j]; smx += hid[Cr + j] * wr[j]; }
        }
        cs[c] = sigmoidf_((sa + b2[c]) + (smx + b2[c]));
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void cbam_compress_kernel(const T* __restrict__ g, const float* __restrict__ partial,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ w2, const float* __restrict__ b2,
                                                            float* __restrict__ cscale, float* __restrict__ comp, int P, int C, int Cr) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* pb = sm; float* cs = pb + 2 * C; float* hid = cs + C;
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    cbam_channel_scale(partial, w1, b1, w2, b2, b, P, C, Cr, pb, hid, cs);
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += 256) cscale[(size_t)b * C + c] = cs[c];
    for (int i = wave; i < CBAM_PIX; i += 4) {                   // one wave per pixel
        const int pix = blockIdx.x * CBAM_PIX + i;
        if (pix >= P) break;
        const T* gp = g + ((size_t)b * P + pix) * C;
        float s = 0.f, m = -INFINITY;
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 v = ld4(gp + c);
            const f32x4 k = *reinterpret_cast<const f32x4*>(cs + c);
            const float a0 = v.x * k.x, a1 = v.y * k.y, a2 = v.z * k.z, a3 = v.w * k.w;
            s += (a0 + a1) + (a2 + a3);
            m = fmaxf(fmaxf(m, fmaxf(a0, a1)), fmaxf(a2, a3));
        }
        s = wave_sum(s); m = wave_max(m);
        if (lane == 0) { comp[((size_t)b * P + pix) * 2 + 0] = m; comp[((size_t)b * P + pix) * 2 + 1] = s / (float)C; }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cbam_gate_apply_kernel(const T* __restrict__ g, const float* __restrict__ cscale,
                                                              const float* __restrict__ comp, const float* __restrict__ sw,
                                                              const float* __restrict__ sb, T* __restrict__ out,
                                                              int H, int W, int C) {
    __shared__ float sg[CBAM_PIX];
    __shared__ float swl[98];
    const int P = H * W, b = blockIdx.y, tid = threadIdx.x;
    if (tid < 98) swl[tid] = sw[tid];
    __syncthreads();
    {   // 16 lanes per pixel over the 98 taps of the 7x7x2 gate
        const int i = tid >> 4, l = tid & 15;
        const int pix = blockIdx.x * CBAM_PIX + i;
        float s = 0.f;
        if (pix < P) {
            const int y = pix / W, x = pix - y * W;
            // the lane's 6-7 taps: unconditional loads from clamped coordinates, requested together; a padding tap is selected away
            float cv[7]; bool okv[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int t = min(l + 16 * u, 97);
                const int ch = t / 49, r = t - ch * 49, dy = r / 7, dx = r - dy * 7;
                const int iy = y + dy - 3, ix = x + dx - 3;
                okv[u] = l + 16 * u < 98 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                cv[u] = comp[(((size_t)b * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)) * 2 + ch];
            }
#pragma unroll
            for (int u = 0; u < 7; ++u)
                if (okv[u]) s += cv[u] * swl[min(l + 16 * u, 97)];
        }
        s += __shfl_xor(s, 8, 16); s += __shfl_xor(s, 4, 16); s += __shfl_xor(s, 2, 16); s += __shfl_xor(s, 1, 16);
        if (l == 0) sg[i] = sigmoidf_(s + sb[0]);
    }
    __syncthreads();
    const int C4 = C >> 2;
#pragma unroll 4
    for (int e = tid; e < CBAM_PIX * C4; e += 256) {
        const int c4 = e % C4, i = e / C4;
        const int pix = blockIdx.x * CBAM_PIX + i;
        if (pix >= P) break;
        const size_t off = ((size_t)b * P + pix) * C + c4 * 4;
        const f32x4 v = ld4(g + off);
        const f32x4 k = *reinterpret_cast<const f32x4*>(cscale + (size_t)b * C + c4 * 4);
        const float gsc = sg[i];
        f32x4 o;
        o.x = v.x + (v.x * k.x) * gsc; o.y = v.y + (v.y * k.y) * gsc;
        o.z = v.z + (v.z * k.z) * gsc; o.w = v.w + (v.w * k.w) * gsc;
        st4(out + off, o);
    }
}

extern "C" size_t xmem_cbam_workspace_bytes(int B, int P, int C) {
    if (B <= 0 || P <= 0 || C <= 0) return 0;
    return align_up((size_t)B * CBAM_PSPLIT * 2 * C * 4, 256) + align_up((size_t)B * C * 4, 256) +
           align_up((size_t)B * P * 2 * 4, 256) + align_up((size_t)B * P * 4, 256);
}

template <typename T>
static void cbam_launch(const T* g, T* out, int B, int H, int W, int C, int P, int Cr, const float* w1, const float* b1, const float* w2,
                        const float* b2, const float* sw, const float* sb, float* pooled, float* cscale, float* comp, hipStream_t s) {
    hipLaunchKernelGGL(cbam_channel_pool_kernel<T>, dim3(cdiv(C, 64), B, CBAM_PSPLIT), dim3(256), 0, s, g, pooled, P, C);
    const size_t lds = ((size_t)3 * C + 2 * Cr) * sizeof(float);
    hipLaunchKernelGGL(cbam_compress_kernel<T>, dim3(cdiv(P, CBAM_PIX), B), dim3(256), lds, s, g, pooled, w1, b1, w2, b2, cscale, comp, P, C, Cr);
    hipLaunchKernelGGL(cbam_gate_apply_kernel<T>, dim3(cdiv(P, CBAM_PIX), B), dim3(256), 0, s, g, cscale, comp, sw, sb, out, H, W, C);
}

extern "C" int xmem_cbam_residual_t(const void* g, void* out, int half, int B, int H, int W, int C,
                                    const float* w1, const float* b1, const float* w2, const float* b2,
                                    const float* sw, const float* sb, void* workspace, size_t workspace_bytes, void* stream) {
    if (!g || !out || !w1 || !b1 || !w2 || !b2 || !sw || !sb || B <= 0 || H <= 0 || W <= 0 || C <= 0) return XMEM_ERR_BAD_ARG;
    if (C % 16) return XMEM_ERR_UNSUPPORTED;
    const int P = H * W, Cr = C / 16;
    if (!workspace || workspace_bytes < xmem_cbam_workspace_bytes(B, P, C)) return XMEM_ERR_WORKSPACE;
    char* ws = (char*)workspace;
    float* pooled = (float*)ws; ws += align_up((size_t)B * CBAM_PSPLIT * 2 * C * 4, 256);
    float* cscale = (float*)ws; ws += align_up((size_t)B * C * 4, 256);
    float* comp = (float*)ws;   ws += align_up((size_t)B * P * 2 * 4, 256);
    float* sgate = (float*)ws;
    hipStream_t s = (hipStream_t)stream;
    (void)sgate;
    if (half) cbam_launch<_Float16>((const _Float16*)g, (_Float16*)out, B, H, W, C, P, Cr, w1, b1, w2, b2, sw, sb, pooled, cscale, comp, s);
    else cbam_launch<float>((const float*)g, (float*)out, B, H, W, C, P, Cr, w1, b1, w2, b2, sw, sb, pooled, cscale, comp, s);
    return xmem_check_launch();
}
extern "C" int xmem_cbam_residual(const float* g, float* out, int B, int H, int W, int C,
                                  const float* w1, const float* b1, const float* w2, const float* b2,
                                  const float* sw, const float* sb, void* workspace, size_t workspace_bytes, void* stream) {
    return xmem_cbam_residual_t(g, out, 0, B, H, W, C, w1, b1, w2, b2, sw, sb, workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------------------------
// GRU-like gate, add3
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void gru_gate_kernel(const T* __restrict__ values, const float* h, float* nh,      // nh may alias h (in-place state update)
                                size_t BP, int Ch) {
    const size_t total = BP * Ch;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = e / Ch; const int c = (int)(e - pix * Ch);
        const T* v = values + pix * 3 * Ch;
        const float f = sigmoidf_((float)v[c]), u = sigmoidf_((float)v[Ch + c]), n = tanhf((float)v[2 * Ch + c]);
        nh[e] = f * h[e] * (1.f - u) + u * n;
    }
}

extern "C" int xmem_gru_gate_t(const void* values, int values_half, const float* h, float* new_h, int B, int P, int Ch, void* stream) {
    if (!values || !h || !new_h || B <= 0 || P <= 0 || Ch <= 0) return XMEM_ERR_BAD_ARG;
    const size_t BP = (size_t)B * P;
    if (values_half) hipLaunchKernelGGL(gru_gate_kernel<_Float16>, dim3(grid_for(BP * Ch)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)values, h, new_h, BP, Ch);
    else hipLaunchKernelGGL(gru_gate_kernel<float>, dim3(grid_for(BP * Ch)), dim3(256), 0, (hipStream_t)stream, (const float*)values, h, new_h, BP, Ch);
    return xmem_check_launch();
}
extern "C" int xmem_gru_gate(const float* values, const float* h, float* new_h, int B, int P, int Ch, void* stream) {
    return xmem_gru_gate_t(values, 0, h, new_h, B, P, Ch, stream);
}

__global__ void add3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                            float* __restrict__ y, size_t n) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
        y[e] = a[e] + b[e] + c[e];
}

extern "C" int xmem_add3(const float* a, const float* b, const float* c, float* y, size_t n, void* stream) {
    if (!a || !b || !c || !y || n == 0) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(add3_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, c, y, n);
    return xmem_check_launch();
}

// ---------------------------------------------------------------------------------------------
// input packing
// ---------------------------------------------------------------------------------------------
__global__ void pack_image_kernel(const float* __restrict__ img, float* __restrict__ out, int H, int W, int Hp, int Wp, int lh, int lw) {
    const int total = Hp * Wp;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int x = e % Wp, y = e / Wp;
        const int sy = y - lh, sx = x - lw;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) {
            const size_t o = (size_t)sy * W + sx, plane = (size_t)H * W;
            v.x = img[o]; v.y = img[plane + o]; v.z = img[2 * plane + o];
        }
        *reinterpret_cast<f32x4*>(out + (size_t)e * 4) = v;
    }
}

extern "C" int xmem_pack_image(const float* img, float* out, int H, int W, int Hp, int Wp, int lh, int lw, void* stream) {
    if (!img || !out || H <= 0 || W <= 0 || Hp < H || Wp < W || lh < 0 || lw < 0) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(pack_image_kernel, dim3(grid_for((size_t)Hp * Wp)), dim3(256), 0, (hipStream_t)stream, img, out, H, W, Hp, Wp, lh, lw);
    return xmem_check_launch();
}

// uint8 H x W x 3 frame as decoded -> ToTensor (x / 255) + Normalize ((x - mean) / std) + pad + NHWC4 in one pass
// (inference/data/video_reader.py:61-76: transforms.ToTensor, im_normalization; dataset/range_transform.py:5-8).
// Same fp32 operation order as torchvision: a correctly rounded division by 255, a subtraction, a division.
__global__ void pack_image_u8_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int H, int W, int Hp, int Wp, int lh, int lw,
                                     float m0, float m1, float m2, float s0, float s1, float s2) {
    const int total = Hp * Wp;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int x = e % Wp, y = e / Wp;
        const int sy = y - lh, sx = x - lw;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) {
            const uint8_t* px = img + ((size_t)sy * W + sx) * 3;
            v.x = __fdiv_rn(__fsub_rn(__fdiv_rn((float)px[0], 255.0f), m0), s0);
            v.y = __fdiv_rn(__fsub_rn(__fdiv_rn((float)px[1], 255.0f), m1), s1);
            v.z = __fdiv_rn(__fsub_rn(__fdiv_rn((float)px[2], 255.0f), m2), s2);
        }
        *reinterpret_cast<f32x4*>(out + (size_t)e * 4) = v;
    }
}

extern "C" int xmem_pack_image_u8(const uint8_t* img, float* out, int H, int W, int Hp, int Wp, int lh, int lw,
                                  const float* mean3_host, const float* std3_host, void* stream) {
    if (!img || !out || !mean3_host || !std3_host || H <= 0 || W <= 0 || Hp < H || Wp < W || lh < 0 || lw < 0) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(pack_image_u8_kernel, dim3(grid_for((size_t)Hp * Wp)), dim3(256), 0, (hipStream_t)stream, img, out, H, W, Hp, Wp,
                       lh, lw, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
    return xmem_check_launch();
}

__global__ void pack_value_input_kernel(const float* __restrict__ image4, const float* __restrict__ masks, float* __restrict__ out,
                                        int K, int P) {
    const size_t total = (size_t)K * P;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int pix = (int)(e % P), k = (int)(e / P);
        const f32x4 im = *reinterpret_cast<const f32x4*>(image4 + (size_t)pix * 4);
        float others = 0.f;
        for (int j = 0; j < K; ++j) if (j != k) others += masks[(size_t)j * P + pix];
        f32x4 lo = {im.x, im.y, im.z, masks[(size_t)k * P + pix]};
        f32x4 hi = {others, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(out + e * 8) = lo;
        *reinterpret_cast<f32x4*>(out + e * 8 + 4) = hi;
    }
}

extern "C" int xmem_pack_value_input(const float* image4, const float* masks, float* out, int K, int Hp, int Wp, void* stream) {
    if (!image4 || !masks || !out || K <= 0 || Hp <= 0 || Wp <= 0) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(pack_value_input_kernel, dim3(grid_for((size_t)K * Hp * Wp)), dim3(256), 0, (hipStream_t)stream,
                       image4, masks, out, K, Hp * Wp);
    return xmem_check_launch();
}

__global__ void key_post_kernel(const float* __restrict__ proj, int ldp, float* __restrict__ key, float* __restrict__ shr,
                                float* __restrict__ sel, int P, int Ck) {
    const int per = 2 * Ck + 1;
    const size_t total = (size_t)P * per;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int pix = (int)(e / per), c = (int)(e - (size_t)pix * per);
        const float v = proj[(size_t)pix * ldp + c];
        if (c < Ck) key[(size_t)pix * Ck + c] = v;
        else if (c == Ck) { if (shr) shr[pix] = v * v + 1.f; }
        else if (sel) sel[(size_t)pix * Ck + (c - Ck - 1)] = sigmoidf_(v);
    }
}

extern "C" int xmem_key_post(const float* proj, int ldp, float* key, float* shrinkage, float* selection, int P, int Ck, void* stream) {
    if (!proj || !key || P <= 0 || Ck <= 0 || ldp < 2 * Ck + 1) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(key_post_kernel, dim3(grid_for((size_t)P * (2 * Ck + 1))), dim3(256), 0, (hipStream_t)stream,
                       proj, ldp, key, shrinkage, selection, P, Ck);
    return xmem_check_launch();
}

// ---------------------------------------------------------------------------------------------
// decoder tail: x4 bilinear -> sigmoid -> soft aggregation -> softmax (-> crop)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float agg_logit(float p) {
    p = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
    return logf(p / (1.f - p));
}

__device__ __forceinline__ float bilinear4(const float* lg, int h4, int w4, int y0, int y1, int x0, int x1, float ly, float lx) {
    const float hy = 1.f - ly, hx = 1.f - lx;
    return hy * (hx * lg[(size_t)y0 * w4 + x0] + lx * lg[(size_t)y0 * w4 + x1]) +
           ly * (hx * lg[(size_t)y1 * w4 + x0] + lx * lg[(size_t)y1 * w4 + x1]);
}

__global__ void logits_to_prob_kernel(const float* __restrict__ logits, float* __restrict__ prob, float* __restrict__ prob_pad,
                                      int K, int h4, int w4, int H, int W, int lh, int lw) {
    const int Hp = 4 * h4, Wp = 4 * w4;
    const int total = Hp * Wp;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int x = e % Wp, y = e / Wp;
        int y0, y1, x0, x1; float ly, lx;
        bilinear_src(y, 0.25f, h4, y0, y1, ly);
        bilinear_src(x, 0.25f, w4, x0, x1, lx);
        // pass 1: background product and max logit.  The per-object logits are computed ONCE and kept in registers (first
        // 8 objects): recomputing them per pass let the compiler contract the three copies differently, and a 1-ulp change
        // of a saturated probability moves its logit by ~1e-4 - numerator and denominator must see the same value.
        constexpr int KREG = 8;
        float lbuf[KREG];
        float bgp = 1.f, mx = -INFINITY;
        for (int k = 0; k < K; ++k) {
            const float pk = sigmoidf_(bilinear4(logits + (size_t)k * h4 * w4, h4, w4, y0, y1, x0, x1, ly, lx));
            bgp *= (1.f - pk);
            const float lk = agg_logit(pk);
            if (k < KREG) lbuf[k] = lk;
            mx = fmaxf(mx, lk);
        }
        const float l0 = agg_logit(bgp);
        mx = fmaxf(mx, l0);
        auto logit_of = [&](int k) -> float {
            if (k < KREG) return lbuf[k];
            return agg_logit(sigmoidf_(bilinear4(logits + (size_t)k * h4 * w4, h4, w4, y0, y1, x0, x1, ly, lx)));
        };
        // pass 2: softmax terms (kept for the first KREG objects) and denominator
        float ebuf[KREG];
        const float e0 = expf(l0 - mx);
        float den = e0;
        for (int k = 0; k < K; ++k) {
            const float ek = expf(logit_of(k) - mx);
            if (k < KREG) ebuf[k] = ek;
            den += ek;
        }
        const int oy = y - lh, ox = x - lw;
        const bool inside = (unsigned)oy < (unsigned)H && (unsigned)ox < (unsigned)W;
        // pass 3: write
        for (int k = -1; k < K; ++k) {
            const float ek = (k < 0) ? e0 : (k < KREG ? ebuf[k] : expf(logit_of(k) - mx));
            const float pr = ek / den;
            if (inside) prob[((size_t)(k + 1) * H + oy) * W + ox] = pr;
            if (prob_pad) prob_pad[(size_t)(k + 1) * total + e] = pr;
        }
    }
}

extern "C" int xmem_logits_to_prob(const float* logits, float* prob, float* prob_padded, int K, int h4, int w4,
                                   int H, int W, int lh, int lw, void* stream) {
    if (!logits || !prob || K <= 0 || h4 <= 0 || w4 <= 0 || H <= 0 || W <= 0 || lh < 0 || lw < 0) return XMEM_ERR_BAD_ARG;
    if (lh + H > 4 * h4 || lw + W > 4 * w4) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(logits_to_prob_kernel, dim3(grid_for((size_t)16 * h4 * w4)), dim3(256), 0, (hipStream_t)stream,
                       logits, prob, prob_padded, K, h4, w4, H, W, lh, lw);
    return xmem_check_launch();
}

__global__ void aggregate_masks_kernel(const float* __restrict__ masks, float* __restrict__ prob, int K, int P) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < P; e += gridDim.x * blockDim.x) {
        float bgp = 1.f, mx = -INFINITY;
        for (int k = 0; k < K; ++k) {
            const float pk = masks[(size_t)k * P + e];
            bgp *= (1.f - pk);
            mx = fmaxf(mx, agg_logit(pk));
        }
        const float l0 = agg_logit(bgp);
        mx = fmaxf(mx, l0);
        float den = expf(l0 - mx);
        for (int k = 0; k < K; ++k) den += expf(agg_logit(masks[(size_t)k * P + e]) - mx);
        prob[e] = expf(l0 - mx) / den;
        for (int k = 0; k < K; ++k) prob[(size_t)(k + 1) * P + e] = expf(agg_logit(masks[(size_t)k * P + e]) - mx) / den;
    }
}

extern "C" int xmem_aggregate_masks(const float* masks, float* prob, int K, int H, int W, void* stream) {
    if (!masks || !prob || K <= 0 || H <= 0 || W <= 0) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(aggregate_masks_kernel, dim3(grid_for((size_t)H * W)), dim3(256), 0, (hipStream_t)stream, masks, prob, K, H * W);
    return xmem_check_launch();
}

// mask given on a frame that was also segmented (inference_core.py:117-127): objects whose label is valid take
// the given mask, the others keep the prediction, zeroed wherever any given mask is set.
__global__ void merge_masks_kernel(const float* __restrict__ pred, const float* __restrict__ mask, unsigned long long valid_bits,
                                   float* __restrict__ out, int K, int P) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < P; e += gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += mask[(size_t)k * P + e];
        const bool region = s > 0.5f;
        for (int k = 0; k < K; ++k) {
            const bool valid = (valid_bits >> k) & 1ull;
            out[(size_t)k * P + e] = valid ? mask[(size_t)k * P + e] : (region ? 0.f : pred[(size_t)k * P + e]);
        }
    }
}

extern "C" int xmem_merge_masks(const float* pred_no_bg, const float* mask, uint64_t valid_bits, float* out, int K, int H, int W, void* stream) {
    if (!pred_no_bg || !mask || !out || K <= 0 || K > 64 || H <= 0 || W <= 0) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(merge_masks_kernel, dim3(grid_for((size_t)H * W)), dim3(256), 0, (hipStream_t)stream, pred_no_bg, mask,
                       (unsigned long long)valid_bits, out, K, H * W);
    return xmem_check_launch();
}

// F.interpolate(prob.unsqueeze(1), shape, mode='bilinear', align_corners=False) (run_on_video.py:166-168)
__global__ void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int Hi, int Wi, int Ho, int Wo) {
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    const size_t total = (size_t)C * Ho * Wo;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % Wo); const int y = (int)((e / Wo) % Ho); const int c = (int)(e / ((size_t)Wo * Ho));
        int y0, y1, x0, x1; float ly, lx;
        bilinear_src(y, sy, Hi, y0, y1, ly);
        bilinear_src(x, sx, Wi, x0, x1, lx);
        const float* p = in + (size_t)c * Hi * Wi;
        const float hy = 1.f - ly, hx = 1.f - lx;
        out[e] = hy * (hx * p[(size_t)y0 * Wi + x0] + lx * p[(size_t)y0 * Wi + x1]) +
                 ly * (hx * p[(size_t)y1 * Wi + x0] + lx * p[(size_t)y1 * Wi + x1]);
    }
}

extern "C" int xmem_resize_bilinear(const float* in, float* out, int C, int Hi, int Wi, int Ho, int Wo, void* stream) {
    if (!in || !out || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for((size_t)C * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, in, out, C, Hi, Wi, Ho, Wo);
    return xmem_check_launch();
}

__global__ void argmax_u8_kernel(const float* __restrict__ prob, uint8_t* __restrict__ out, int C, int P) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < P; e += gridDim.x * blockDim.x) {
        float best = prob[e]; int bi = 0;
        for (int c = 1; c < C; ++c) {
            const float v = prob[(size_t)c * P + e];
            if (v > best) { best = v; bi = c; }     // first maximal index, as torch.argmax
        }
        out[e] = (uint8_t)bi;
    }
}

extern "C" int xmem_argmax_u8(const float* prob, uint8_t* out, int C, int H, int W, void* stream) {
    if (!prob || !out || C <= 0 || C > 255 || H <= 0 || W <= 0) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(argmax_u8_kernel, dim3(grid_for((size_t)H * W)), dim3(256), 0, (hipStream_t)stream, prob, out, C, H * W);
    return xmem_check_launch();
}

// ---------------------------------------------------------------------------------------------
// layout transposes through a padded LDS tile (32 pixels x 32 channels)
// ---------------------------------------------------------------------------------------------
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, int ld, float* __restrict__ out, int P, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int r = ty; r < 32; r += 8) {
        const int pidx = p0 + r, c = c0 + tx;
        tile[r][tx] = (pidx < P && c < C) ? in[((size_t)b * P + pidx) * ld + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, pidx = p0 + tx;
        if (pidx < P && c < C) out[((size_t)b * C + c) * P + pidx] = tile[tx][r];
    }
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int ld, int P, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, pidx = p0 + tx;
        tile[r][tx] = (pidx < P && c < C) ? in[((size_t)b * C + c) * P + pidx] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int pidx = p0 + r, c = c0 + tx;
        if (pidx < P && c < C) out[((size_t)b * P + pidx) * ld + c] = tile[tx][r];
    }
}

extern "C" int xmem_nhwc_to_nchw(const float* in, int ld, float* out, int B, int P, int C, void* stream) {
    if (!in || !out || B <= 0 || P <= 0 || C <= 0 || ld < C) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(cdiv(P, 32), cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, in, ld, out, P, C);
    return xmem_check_launch();
}

extern "C" int xmem_nchw_to_nhwc(const float* in, float* out, int ld, int B, int P, int C, void* stream) {
    if (!in || !out || B <= 0 || P <= 0 || C <= 0 || ld < C) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(P, 32), cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, in, out, ld, P, C);
    return xmem_check_launch();
}
