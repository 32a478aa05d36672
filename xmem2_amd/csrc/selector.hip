// Annotation-candidate selector (inference/frame_selection/frame_selection.py:99-244): the second consumer of the
// anisotropic-L2 similarity.  For a chosen frame A and every candidate frame B the reference materialises two
// HW x HW similarity matrices,
//     fwd[i][j] = S(mem = cA_i, ms = sA_i ; q = cB_j, qe = eB_j)          (frame_selection.py:218)
//     rev[i][j] = S(mem = cB_i, ms = sB_i ; q = cA_j, qe = eA_j)          (frame_selection.py:219)
// (c = mask-weighted composite key, s = shrinkage, e = selection) and scores the pair with
//     relu(fwd - rev).sum() / numel                                          (frame_selection.py:223-226)
// Here nothing is materialised: a prepare kernel expands every frame once into the two K = 2*C_k operands of the
// contraction ([c^2, c] as memory rows, [-e, 2ce] and b_sq = sum e c^2 as query columns, memory_util.py:20-27), and the
// score kernel runs both MFMA chains for a 64x64 block per wave, takes relu(fwd - rev) in registers and reduces.
// One launch scores ALL candidates against one chosen frame; the host keeps the running minimum over chosen
// frames, so a selection of k frames costs (|previous| + k - 1) launches instead of the reference's O(k^2 N) pairs.
#include "common.hpp"
#include <math.h>

#define SEL_TILE 128       // output block per workgroup (2x2 waves of 64x64)

// ---------------------------------------------------------------------------------------------
// prepare: composite key + expanded operands + mask presence count for one frame
// ---------------------------------------------------------------------------------------------
// key/sel: [HW][Ck] rows; mask: [C][H][W] float (or NULL => composite = key); nearest resize to h x w as
// torch.nn.functional.interpolate(mode='nearest') (src = min(floor(dst * (in/out)), in-1), scale in fp32);
// composite = (key * m) * alpha + key * (1 - alpha) with separately rounded products (frame_selection.py:183-184).
__global__ void selector_prepare_kernel(const float* __restrict__ key, const float* __restrict__ sel,
                                        const float* __restrict__ mask, int C, int H, int W, int h, int w, int Ck,
                                        float alpha, float one_minus_alpha,
                                        float* __restrict__ Mexp, float* __restrict__ Qexp, float* __restrict__ bsq) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * (blockDim.x >> 6) + wave;
    if (p >= h * w) return;
    float m = 1.f;
    if (mask) {
        const int y = p / w, x = p % w;
        const float sy = (float)H / (float)h, sx = (float)W / (float)w;
        const int yy = min((int)floorf((float)y * sy), H - 1);
        const int xx = min((int)floorf((float)x * sx), W - 1);
        m = mask[(size_t)yy * W + xx];
        for (int c = 1; c < C; ++c) m = fmaxf(m, mask[((size_t)c * H + yy) * W + xx]);
    }
    float bs = 0.f;
    for (int c = lane; c < Ck; c += 64) {
        const float k = key[(size_t)p * Ck + c];
        float ck = k;
        if (mask) ck = __fadd_rn(__fmul_rn(__fmul_rn(k, m), alpha), __fmul_rn(k, one_minus_alpha));
        const float e = sel[(size_t)p * Ck + c];
        Mexp[(size_t)p * 2 * Ck + c] = ck * ck;
        Mexp[(size_t)p * 2 * Ck + Ck + c] = ck;
        Qexp[(size_t)p * 2 * Ck + c] = -e;
        Qexp[(size_t)p * 2 * Ck + Ck + c] = 2.f * (ck * e);
        bs += e * (ck * ck);
    }
    bs = wave_sum(bs);
    if (lane == 0) bsq[p] = bs;
}

// count of pixels whose max-over-channels mask value exceeds eps (frame_selection.py:161-163)
__global__ void mask_presence_kernel(const float* __restrict__ mask, int C, int HW, float eps, int* __restrict__ count) {
    int local = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        float m = mask[i];
        for (int c = 1; c < C; ++c) m = fmaxf(m, mask[(size_t)c * HW + i]);
        local += (m > eps) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(count, local);
}

extern "C" int xmem_selector_prepare(const float* key, const float* sel, const float* mask, int C, int H, int W,
                                     int h, int w, int Ck, float alpha, float one_minus_alpha, float eps,
                                     float* Mexp, float* Qexp, float* bsq, int32_t* presence, void* stream) {
    if (!key || !sel || !Mexp || !Qexp || !bsq || h <= 0 || w <= 0 || Ck <= 0) return XMEM_ERR_BAD_ARG;
    if (mask && (C <= 0 || H <= 0 || W <= 0)) return XMEM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (presence) {
        if (hipMemsetAsync(presence, 0, sizeof(int32_t), st) != hipSuccess) return XMEM_ERR_LAUNCH;
        if (mask) {
            const int HWf = H * W;
            hipLaunchKernelGGL(mask_presence_kernel, dim3(min(cdiv(HWf, 256), 1024)), dim3(256), 0, st, mask, C, HWf, eps, presence);
        }
    }
    hipLaunchKernelGGL(selector_prepare_kernel, dim3(cdiv(h * w, 4)), dim3(256), 0, st, key, sel, mask, C, H, W, h, w, Ck,
                       alpha, one_minus_alpha, Mexp, Qexp, bsq);
    return xmem_check_launch();
}

// ---------------------------------------------------------------------------------------------
// score: relu(fwd - rev) summed over the HW x HW block grid, all candidates vs one chosen frame
// ---------------------------------------------------------------------------------------------
// MFMA 32x32x2 operands come straight from global/L2 (one frame's operands are < 1 MB): lane (r = l&31, hh = l>>5)
// loads float4 #(2c + hh) of its row per K-chunk c, and k-step t of the chunk uses component t on BOTH operands, so
// every k is contracted exactly once (the contraction order is free).  A wave owns a 64x64 block in both
// directions: 2 dirs x 2x2 accumulators of 16 registers.
template <int K2>
__global__ __launch_bounds__(256) void selector_score_kernel(
        const float* __restrict__ Mexp, const float* __restrict__ Qexp, const float* __restrict__ bsq,
        const float* __restrict__ shr, int HW, int chosen, const uint8_t* __restrict__ valid, float inv_sqrt_ck_is_div,
        double* __restrict__ partial, int tiles) {
    const int f = blockIdx.y;
    const int tile = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ double wsum[4];
    double* my = partial + (size_t)f * tiles * tiles + tile;
    if (valid && !valid[f]) { if (threadIdx.x == 0) *my = 0.0; return; }
    const int ti = tile / tiles, tj = tile % tiles;
    const int i0 = ti * SEL_TILE + (wave >> 1) * 64, j0 = tj * SEL_TILE + (wave & 1) * 64;
    const int r = lane & 31, hh = lane >> 5;
    const size_t fs = (size_t)HW * K2;
    const float* MA = Mexp + (size_t)chosen * fs; const float* QA = Qexp + (size_t)chosen * fs;
    const float* MB = Mexp + (size_t)f * fs;      const float* QB = Qexp + (size_t)f * fs;
    // operand rows of this lane (clamped; out-of-range products are masked in the epilogue)
    const int ia = min(i0 + r, HW - 1), ib = min(i0 + 32 + r, HW - 1);
    const int ja = min(j0 + r, HW - 1), jb = min(j0 + 32 + r, HW - 1);
    const f32x4* pMA0 = reinterpret_cast<const f32x4*>(MA + (size_t)ia * K2) + hh;
    const f32x4* pMA1 = reinterpret_cast<const f32x4*>(MA + (size_t)ib * K2) + hh;
    const f32x4* pMB0 = reinterpret_cast<const f32x4*>(MB + (size_t)ia * K2) + hh;
    const f32x4* pMB1 = reinterpret_cast<const f32x4*>(MB + (size_t)ib * K2) + hh;
    const f32x4* pQA0 = reinterpret_cast<const f32x4*>(QA + (size_t)ja * K2) + hh;
    const f32x4* pQA1 = reinterpret_cast<const f32x4*>(QA + (size_t)jb * K2) + hh;
    const f32x4* pQB0 = reinterpret_cast<const f32x4*>(QB + (size_t)ja * K2) + hh;
    const f32x4* pQB1 = reinterpret_cast<const f32x4*>(QB + (size_t)jb * K2) + hh;

    f32x16 F00 = {0}, F01 = {0}, F10 = {0}, F11 = {0};     // fwd: rows of A (memory) x cols of B (query)
    f32x16 R00 = {0}, R01 = {0}, R10 = {0}, R11 = {0};     // rev: rows of B (memory) x cols of A (query)
    constexpr int CH = K2 / 8;
    f32x4 a0 = pMA0[0], a1 = pMA1[0], b0 = pMB0[0], b1 = pMB1[0];
    f32x4 qa0 = pQA0[0], qa1 = pQA1[0], qb0 = pQB0[0], qb1 = pQB1[0];
#pragma unroll 2
    for (int c = 0; c < CH; ++c) {
        const int cn = (c + 1 < CH) ? 2 * (c + 1) : 2 * c;
        const f32x4 na0 = pMA0[cn], na1 = pMA1[cn], nb0 = pMB0[cn], nb1 = pMB1[cn];
        const f32x4 nqa0 = pQA0[cn], nqa1 = pQA1[cn], nqb0 = pQB0[cn], nqb1 = pQB1[cn];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            F00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], qb0[t], F00, 0, 0, 0);
            F01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], qb1[t], F01, 0, 0, 0);
            F10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], qb0[t], F10, 0, 0, 0);
            F11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], qb1[t], F11, 0, 0, 0);
            R00 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[t], qa0[t], R00, 0, 0, 0);
            R01 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[t], qa1[t], R01, 0, 0, 0);
            R10 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[t], qa0[t], R10, 0, 0, 0);
            R11 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[t], qa1[t], R11, 0, 0, 0);
        }
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; qa0 = nqa0; qa1 = nqa1; qb0 = nqb0; qb1 = nqb1;
    }
    // epilogue: D[row = (q&3) + 8*(q>>2) + 4*hh][col = r]
    const float* bsqA = bsq + (size_t)chosen * HW; const float* bsqB = bsq + (size_t)f * HW;
    const float* sA = shr + (size_t)chosen * HW;   const float* sB = shr + (size_t)f * HW;
    const float sq = inv_sqrt_ck_is_div;           // sqrt(C_k): divided, as `similarity * ms / math.sqrt(CK)`
    float acc = 0.f;
    auto fold = [&](const f32x16& Fv, const f32x16& Rv, int ibase, int jbase) {
        const int j = jbase + r;
        if (j >= HW) return;
        const float bB = bsqB[j], bA = bsqA[j];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = ibase + (q & 3) + 8 * (q >> 2) + 4 * hh;
            if (i < HW) {
                const float vf = ((Fv[q] - bB) * sA[i]) / sq;
                const float vr = ((Rv[q] - bA) * sB[i]) / sq;
                acc += fmaxf(vf - vr, 0.f);
            }
        }
    };
    fold(F00, R00, i0, j0); fold(F01, R01, i0, j0 + 32); fold(F10, R10, i0 + 32, j0); fold(F11, R11, i0 + 32, j0 + 32);
    double d = (double)acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if (lane == 0) wsum[wave] = d;
    __syncthreads();
    if (threadIdx.x == 0) *my = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ void selector_reduce_kernel(const double* __restrict__ partial, int per_frame, double denom, double* __restrict__ out) {
    const int f = blockIdx.x;
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < per_frame; i += 256) s += partial[(size_t)f * per_frame + i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[f] = sh[0] / denom;
}

extern "C" size_t xmem_cycle_dissimilarity_workspace_bytes(int n_frames, int HW) {
    if (n_frames <= 0 || HW <= 0) return 0;
    const size_t tiles = (size_t)cdiv(HW, SEL_TILE);
    return (size_t)n_frames * tiles * tiles * sizeof(double);
}

extern "C" int xmem_cycle_dissimilarity(const float* Mexp, const float* Qexp, const float* bsq, const float* shrinkage,
                                        int n_frames, int HW, int Ck, int chosen, const uint8_t* valid, double* out,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!Mexp || !Qexp || !bsq || !shrinkage || !out || n_frames <= 0 || HW <= 0) return XMEM_ERR_BAD_ARG;
    if (chosen < 0 || chosen >= n_frames) return XMEM_ERR_BAD_ARG;
    if (Ck != 64) return XMEM_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < xmem_cycle_dissimilarity_workspace_bytes(n_frames, HW)) return XMEM_ERR_WORKSPACE;
    const int tiles = cdiv(HW, SEL_TILE);
    hipStream_t st = (hipStream_t)stream;
    double* partial = (double*)workspace;
    hipLaunchKernelGGL(selector_score_kernel<128>, dim3(tiles * tiles, n_frames), dim3(256), 0, st,
                       Mexp, Qexp, bsq, shrinkage, HW, chosen, valid, sqrtf((float)Ck), partial, tiles);
    hipLaunchKernelGGL(selector_reduce_kernel, dim3(n_frames), dim3(256), 0, st, partial, tiles * tiles,
                       (double)HW * (double)HW, out);
    return xmem_check_launch();
}
