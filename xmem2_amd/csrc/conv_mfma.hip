// Implicit-GEMM convolution on the CDNA4 fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// GEMM view:  M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin (Cin fastest).
// NHWC activations and [Cout][KH][KW][Cin] weights make both operands K-contiguous, so one
// 128-byte global segment feeds one padded LDS row and every MFMA operand fragment is a single
// ds_read_b128 (row stride 36 floats -> the 16 lanes of a b128 lane-group hit 16 distinct 16-B slots).
//
// MFMA lane maps (cdna_hip_programming.md section 3): A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
// D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31].  A lane reads 4 consecutive k (float4) at k-offset 4*(l>>5)
// inside an 8-wide group; MFMA step j then contracts k = {j, 4+j} of the group - the same permutation for
// A and B, so the sum is unchanged.  fp32-input MFMA is an exact fmaf chain: results are fp32-roundoff
// class against the reference's CPU convolution.
//
// Replaces the nn.Conv2d/BatchNorm2d/ReLU/residual call sites listed in include/xmem_hip.h.
#include "common.hpp"
#include "gemm_stream.hpp"
#include <stdlib.h>

// BK (k-depth of a staged tile) is a template parameter: 32 or 64.  LDS rows are padded by 4 floats:
// stride 36 (=9x16 B) or 68 (=17x16 B) floats; both are odd multiples of 16 B, so the 16 lanes of a ds_read_b128
// lane-group (16 distinct rows) fall on 16 distinct 16-B slots of the 256-B bank row.

struct ConvArgs {
    const float* in; const float* w; const float* scale; const float* shift; const float* res;
    float* out; float* partial;
    int B, H, W, Cin, ldin;
    int Ho, Wo, Cout, ldout, ldres;
    int KH, KW, stride, pad;
    int K, M, HoWo;
    int relu_in, relu_out;
    int nk, splitk, kt_per_split;
    int tiles_m, tiles_n;
    int raw;                                   // 1: store the bare accumulator (Winograd-domain GEMM)
    int res_mod;                               // > 0: residual is one image broadcast over the batch (pixel index mod Ho*Wo)
    long in_gstride, w_gstride, out_gstride;   // blockIdx.y = group (the 16 Winograd tile positions), floats
    int a_presplit;                            // SPLIT kernels: the A operand already holds [hi4|lo4] groups (Winograd-domain V)
    int dbg;                                   // tools builds only (-DXMEM_TOOLS, env XMEM_CONV_DBG): 1 = no epilogue stores, 2 = every M-tile loads
                                               // tile 0's A rows (L2-resident operands); results are then wrong - they attribute time
};
#ifdef XMEM_TOOLS
#define CDBG(bit) (p.dbg & (bit))
// XMEM_CONV_DBG & 8: phase timestamps of every workgroup of the pointwise kernel (s_memtime, 100 MHz-class constant clock):
// entry, first tiles requested, first tiles in LDS, k-loop done, stores issued; xmem_conv2d_nhwc prints their averages.
#define CTRACE_MAX 16384
__device__ unsigned long long g_conv_trace[CTRACE_MAX][6];
#define CTRACE(slot) do { if (CDBG(8) && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < CTRACE_MAX) \
    g_conv_trace[blockIdx.x][slot] = __builtin_readcyclecounter(); } while (0)
#else
#define CDBG(bit) 0
#define CTRACE(slot) do { } while (0)
#endif

// ---- SPLIT-OPERAND arithmetic (opt-in mode 'fp32x', never the default) ------------------------------------------------
// An fp32 value x is carried as two halfs, x = hi + lo (+ O(2^-21 |x|)): hi = x rounded toward zero to fp16 (saturating:
// finite values never become inf), lo = fp16(x - hi).  Four channels travel as ONE 16-byte group [hi0 hi1 hi2 hi3 | lo0 lo1
// lo2 lo3] - the same bytes as the four floats they replace, so tensor shapes, pixel strides, LDS layout and every loader of
// this file are unchanged.  As an operand of v_mfma_f32_32x32x16_f16 a group fills the 8 k-slots of a lane, and with B the
// same layout   mfma(A, B) = sum hi_a hi_b + lo_a lo_b,   mfma(A, swap halves of B) = sum hi_a lo_b + lo_a hi_b :
// two fp16 MFMAs (2 x 32 cycles) give all four partial products of 8 channels, where the fp32 pipe spends 4 x 64 cycles.
// Products of halfs are exact in the fp32 accumulator; what is lost is the representation error of the two operands.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 split_pack(const f32x4 v) {
    const fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y), h23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
    h16x8 o;
    o[0] = (_Float16)h01.x; o[1] = (_Float16)h01.y; o[2] = (_Float16)h23.x; o[3] = (_Float16)h23.y;
    // the residual of a saturated hi is clamped as well: |x| beyond 131008 saturates instead of turning into inf - inf
    o[4] = (_Float16)__builtin_amdgcn_fmed3f(v.x - (float)h01.x, -65504.f, 65504.f);
    o[5] = (_Float16)__builtin_amdgcn_fmed3f(v.y - (float)h01.y, -65504.f, 65504.f);
    o[6] = (_Float16)__builtin_amdgcn_fmed3f(v.z - (float)h23.x, -65504.f, 65504.f);
    o[7] = (_Float16)__builtin_amdgcn_fmed3f(v.w - (float)h23.y, -65504.f, 65504.f);
    return __builtin_bit_cast(f32x4, o);
}

// ONE: 1x1 / pad 0 (any stride) with Cin % BK == 0 (the pointwise layers and the 16 Winograd-domain GEMMs): operand rows are
// plain matrix rows, so each thread keeps loop-invariant 32-bit byte offsets and the K loop only advances a uniform base -
// no per-tile index arithmetic, bounds tests or exec-masked branches around the loads (rows past M / Cout are clamped:
// their products are never stored).  The address VALU work of the general loader was comparable to the MFMA issue time.
// HALF (the fp16 loop, config['precision'] = 'fp16': the counterpart of the reference's autocast frame loop,
// inference/run_on_video.py:76): activations and weights are IEEE halfs in memory, contracted on v_mfma_f32_32x32x16_f16 with fp32
// accumulation; the epilogue (scale / shift / residual / relu) runs in fp32 and the result is stored as fp32 (HALF = 1: key
// projection, split-K partials) or rounded once to fp16 (HALF = 2: every activation).  Eight halfs occupy the 16 bytes of four
// floats, so with Cin, ldin and K passed in 4-byte units (Cin / 2, ...) every loader of this file is unchanged; a 16-byte
// fragment is ONE fp16 MFMA (k = 16: 8 halfs per lane half) instead of four fp32 ones.
template <int BM, int BN, int TM, int TN, int BK, bool GENERIC, bool ONE = false, bool SPLIT = false, int HALF = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW) void conv_mfma_kernel(ConvArgs p) {
    constexpr int WN = BN / (32 * TN);
    constexpr int WM = BM / (32 * TM);
    static_assert(WM * WN == NW, "NW waves per workgroup (4; 8 for the 256x128 tile of the fp16 loop)");
    constexpr int LDK = BK + 4;
    constexpr int C4 = BK / 4;                 // float4 columns per staged row
    constexpr int RPP = 64 * NW / C4;          // rows staged per pass of the workgroup's threads
    constexpr int RA = BM / RPP, RB = BN / RPP;
    constexpr int BUF = (BM + BN) * LDK;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int lrow = tid / C4, c4 = tid % C4;
    CTRACE(0);

    // XCD-aware tile order: consecutive tile ids (same M-tile, neighbouring N-tiles) stay on one XCD / L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const float* __restrict__ gin = p.in + (size_t)blockIdx.y * p.in_gstride;
    const float* __restrict__ gw = p.w + (size_t)blockIdx.y * p.w_gstride;
    float* __restrict__ gout = p.out + (size_t)blockIdx.y * p.out_gstride;

    int a_ih0[RA], a_iw0[RA], a_pix[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + lrow + RPP * i;
        if (m < p.M) {
            const int b = m / p.HoWo, rem = m - b * p.HoWo;
            const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
            a_ih0[i] = oh * p.stride - p.pad;
            a_iw0[i] = ow * p.stride - p.pad;
            a_pix[i] = b * p.H * p.W;
        } else {
            a_ih0[i] = -(1 << 28); a_iw0[i] = -(1 << 28); a_pix[i] = 0;
        }
    }

    const int kt_begin = blockIdx.z * p.kt_per_split;
    const int kt_end = min(p.nk, kt_begin + p.kt_per_split);

    unsigned a_boff[RA], b_boff[RB];           // ONE: byte offsets of this thread's operand rows (k = 0)
    if (ONE) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = CDBG(2) ? min(lrow + RPP * i, p.M - 1) : min(m0 + lrow + RPP * i, p.M - 1);
            unsigned pix = (unsigned)m;
            if (p.stride != 1) {               // strided pointwise layer (ResNet downsample): input pixel of output pixel m
                const int b = m / p.HoWo, rem = m - b * p.HoWo;
                const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
                pix = (unsigned)((b * p.H + oh * p.stride) * p.W + ow * p.stride);
            }
            a_boff[i] = (pix * (unsigned)p.ldin + (unsigned)(c4 * 4)) * 4u;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int n = min(n0 + lrow + RPP * i, p.Cout - 1);
            b_boff[i] = ((unsigned)n * (unsigned)p.K + (unsigned)(c4 * 4)) * 4u;
        }
    }

    f32x4 ra[RA], rb[RB];
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (ONE) {
            const char* abase = reinterpret_cast<const char*>(gin) + (size_t)k0 * 4;
            const char* bbase = reinterpret_cast<const char*>(gw) + (size_t)k0 * 4;
#pragma unroll
            for (int i = 0; i < RA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(abase + a_boff[i]);
#pragma unroll
            for (int i = 0; i < RB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bbase + b_boff[i]);
            return;
        }
        if (!GENERIC) {
            const int tap = k0 / p.Cin, c0 = k0 - tap * p.Cin;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
                const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const long off = (long)(a_pix[i] + ih * p.W + iw) * p.ldin + c0 + c4 * 4;
                ra[i] = ok ? *reinterpret_cast<const f32x4*>(gin + off) : zero;
            }
        } else {
            const int k = k0 + c4 * 4;
            const bool kok = k < p.K;
            const int tap = k / p.Cin, c0 = k - tap * p.Cin;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
                const bool ok = kok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const long off = (long)(a_pix[i] + ih * p.W + iw) * p.ldin + c0;
                ra[i] = ok ? *reinterpret_cast<const f32x4*>(gin + off) : zero;
            }
        }
        const int kb = k0 + c4 * 4;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int n = n0 + lrow + RPP * i;
            const bool ok = n < p.Cout && kb < p.K;
            rb[i] = ok ? *reinterpret_cast<const f32x4*>(gw + (size_t)n * p.K + kb) : zero;
        }
    };
    auto store_tile = [&](int buf) {
        float* As = smem + buf * BUF;
        float* Bs = As + BM * LDK;
        if (p.relu_in) {                       // relu-on-load, applied when the data is consumed (never right after the
#pragma unroll                                 // global load: that would stall the wave before its MFMAs)
            for (int i = 0; i < RA; ++i) {
                if (HALF) {
                    const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    ra[i] = __builtin_bit_cast(f32x4, __builtin_elementwise_max(__builtin_bit_cast(h16x8, ra[i]), z));
                } else {
                    ra[i].x = fmaxf(ra[i].x, 0.f); ra[i].y = fmaxf(ra[i].y, 0.f);
                    ra[i].z = fmaxf(ra[i].z, 0.f); ra[i].w = fmaxf(ra[i].w, 0.f);
                }
            }
        }
        if (SPLIT && !p.a_presplit) {          // fp32 activations become [hi4|lo4] groups on their way into LDS
#pragma unroll
            for (int i = 0; i < RA; ++i) ra[i] = split_pack(ra[i]);
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(&As[(lrow + RPP * i) * LDK + c4 * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(&Bs[(lrow + RPP * i) * LDK + c4 * 4]) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // epilogue store mode (uniform): 0 raw accumulators, 1 split-K partials, 2 scale / shift (/ relu), 3 the same + residual
    const int mode = p.raw ? 0 : (p.splitk > 1 ? 1 : (p.res ? 3 : 2));
    const bool half_io = (HALF == 2) && mode >= 2;                        // activations (output and residual) are halfs; ldout / ldres count halfs
    // Small tiles request their residual values HERE, ahead of the operand tiles: the short-K pointwise layers (64 -> 256 at
    // 1/4 resolution: two k-tiles) are bound by memory round trips per workgroup, and a residual fetched in the epilogue adds a
    // whole one after the last MFMA (the same lane / row mapping as the epilogue below).
    constexpr bool EARLY_RES = TM * TN <= 2;
    float rve[EARLY_RES ? TM * TN : 1][16];
    // Two-accumulator tiles (128x64) also request the per-channel scale / shift of the epilogue here (round 5): -5 ... -12 % on the
    // batch-4 pointwise layers that take that tile; the 64x64 tile does not gain (+-2 %) and keeps the epilogue fetch
    // (profiles/r05_pointwise_prefetch_ab.txt).
    constexpr bool EARLY_SS = TM * TN == 2;
    float sce[EARLY_SS ? TN : 1], she[EARLY_SS ? TN : 1];
    // Round 6: WHERE they are requested.  Until then they stood between a workgroup's entry and its first operand request (~770
    // instructions with a per-element modulo behind a branch for the batch-broadcast case): the residual cost 13 us of a 51 us layer.  The
    // plain case is now 16 clamped rows off ONE 64-bit base, and the requests are issued AFTER the first operand tiles are in LDS, right
    // before the k-loop: they fly under the first k-tile's MFMAs, and the next tile's operand loads - younger - are waited for as a whole
    // anyway.  (Between the first operand request and its wait they lose: the compiler's load counter is the minimum over all paths to
    // a join, so with a path that loads no residual the operand wait becomes vmcnt(0..3), behind every younger residual load.)
    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        CTRACE(1);
        store_tile(0);
    }

    __syncthreads();
    CTRACE(2);

    if (EARLY_SS) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = min(n0 + wn * 32 * TN + j * 32 + l31, p.Cout - 1);
            sce[j] = mode >= 2 ? p.scale[n] : 1.f;
            she[j] = mode >= 2 ? p.shift[n] : 0.f;
        }
    }
    if (EARLY_RES && mode == 3 && p.res_mod == 0) {
        // plain residual (the rule): 16 unconditional loads from clamped rows, no per-element modulo, no branch
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = min(n0 + wn * 32 * TN + j * 32 + l31, p.Cout - 1);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + wm * 32 * TM + i * 32 + 4 * lh;
                const int left = p.M - 1 - mb;                             // rows below this lane's first one (negative past the end)
                const long rb = (long)mb * p.ldres + n;                     // ONE 64-bit product per block; the rows are 24-bit multiples of ldres
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long t = rb + __mul24(min((r & 3) + 8 * (r >> 2), left), p.ldres);
                    if (half_io) rve[i * TN + j][r] = (float)reinterpret_cast<const _Float16*>(p.res)[t];
                    else rve[i * TN + j][r] = p.res[t];
                }
            }
        }
    } else if (EARLY_RES && mode == 3) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = min(n0 + wn * 32 * TN + j * 32 + l31, p.Cout - 1);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + wm * 32 * TM + i * 32 + 4 * lh;
                const int mr = p.res_mod ? mb % p.res_mod : mb;
                const bool wrap_ok = p.res_mod >= 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = (r & 3) + 8 * (r >> 2);
                    int t = mr + o;
                    if (p.res_mod) { if (wrap_ok) { if (t >= p.res_mod) t -= p.res_mod; } else t = (mb + o) % p.res_mod; }
                    // unconditional loads from a clamped row (a predicated load is a branch, and behind a branch every load waits for
                    // its predecessor: rows past M are fetched from row M - 1 and never stored)
                    if (!p.res_mod) t = min(t, p.M - 1);
                    if (half_io) rve[i * TN + j][r] = (float)reinterpret_cast<const _Float16*>(p.res)[(size_t)t * p.ldres + n];
                    else rve[i * TN + j][r] = p.res[(size_t)t * p.ldres + n];
                }
            }
        }
    }

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int buf = (kt - kt_begin) & 1;
        const bool has_next = kt + 1 < kt_end;
        if (has_next) load_tile(kt + 1);           // global loads in flight under the MFMAs below

        const float* As = smem + buf * BUF + (wm * 32 * TM + l31) * LDK + lh * 4;
        const float* Bs = smem + buf * BUF + BM * LDK + (wn * 32 * TN + l31) * LDK + lh * 4;
        // fragment loads run one k-group ahead of the MFMAs (two register sets): the ds_read latency of group kk+1 hides
        // under the MFMA chain of group kk instead of stalling the wave between groups
        f32x4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDK);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BK / 8) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[nxt][i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDK + (kk + 1) * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[nxt][j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDK + (kk + 1) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);     // keep the prefetch above this k-group's MFMAs
            if (HALF) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, af[cur][i]), __builtin_bit_cast(h16x8, bf[cur][j]),
                                                                           acc[i][j], 0, 0, 0);
            } else if (SPLIT) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const h16x8 b = __builtin_bit_cast(h16x8, bf[cur][j]);
                    const h16x8 bs = __builtin_shufflevector(b, b, 4, 5, 6, 7, 0, 1, 2, 3);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const h16x8 a = __builtin_bit_cast(h16x8, af[cur][i]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bs, acc[i][j], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][s], bf[cur][j][s], acc[i][j], 0, 0, 0);
            }
        }
        if (has_next) store_tile(buf ^ 1);
        __syncthreads();
    }

    CTRACE(3);
    // epilogue: lane owns output channel n (col) and 16 pixels (rows) per 32x32 tile.  The store mode is uniform, so it is
    // decided once and each 32x32 block runs straight-line code: all residual loads of a block are issued together
    // (one memory round trip per block instead of one per element) before the fused scale / shift / add / relu and stores.
    float* const obase = (mode == 1) ? p.partial + (size_t)blockIdx.z * p.M * p.Cout : gout;
    const long ldo = (mode == 1) ? (long)p.Cout : (long)p.ldout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * 32 * TN + j * 32 + l31;
        if (n >= p.Cout) continue;
        float sc = 1.f, sh = 0.f;
        if (EARLY_SS) { sc = sce[j]; sh = she[j]; }
        else if (mode >= 2) { sc = p.scale[n]; sh = p.shift[n]; }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * 32 * TM + i * 32 + 4 * lh;          // this lane's rows: mb + (r & 3) + 8 * (r >> 2)
            float* const orow = obase + (size_t)mb * ldo + n;
            _Float16* const orow_h = reinterpret_cast<_Float16*>(obase) + (size_t)mb * ldo + n;
            float rv[16];
            if (EARLY_RES && mode == 3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = rve[i * TN + j][r];
            } else if (mode == 3) {
                int mr = p.res_mod ? mb % p.res_mod : mb;                 // broadcast residual: row index modulo one image
                const bool wrap_ok = p.res_mod >= 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = (r & 3) + 8 * (r >> 2);
                    int t = mr + o;
                    if (p.res_mod) { if (wrap_ok) { if (t >= p.res_mod) t -= p.res_mod; } else t = (mb + o) % p.res_mod; }
                    if (!p.res_mod) t = min(t, p.M - 1);          // clamped, unconditional (see the early request above)
                    if (half_io) rv[r] = (float)reinterpret_cast<const _Float16*>(p.res)[(size_t)t * p.ldres + n];
                    else rv[r] = p.res[(size_t)t * p.ldres + n];
                }
            }
            // TWO PHASES (round 6).  gfx9 counts loads AND stores in one in-order counter (vmcnt).  With the fused arithmetic and the store
            // of an element in one loop body - behind the row-bounds branch of that element - the compiler cannot know, at the join after a
            // skipped element, whether scale / shift / residual have arrived, and waits vmcnt(0) in EVERY body: each store then waits for the
            // previous store to land - 16 serialised round trips, 9 000 of a 17 500-cycle workgroup at the 64 -> 256 pointwise layer
            // (profiles/r06_pointwise_epilogue.txt).  Phase 1 consumes every loaded operand; phase 2 only stores registers.
            float vout[16];
            if (mode >= 2) {
                const bool add_res = mode == 3, relu = p.relu_out != 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] * sc + sh;
                    if (add_res) v += rv[r];
                    vout[r] = relu ? fmaxf(v, 0.f) : v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) vout[r] = acc[i][j][r];
            }
            if (CDBG(1)) { if (vout[0] == 1.2345e-33f) orow[0] = vout[0]; continue; }
            if (m0 + wm * 32 * TM + i * 32 + 32 <= p.M) {      // (wave-uniform) every row of the block exists: straight-line stores
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = (r & 3) + 8 * (r >> 2);
                    if (half_io) orow_h[(long)o * ldo] = (_Float16)vout[r];
                    else orow[(long)o * ldo] = vout[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = (r & 3) + 8 * (r >> 2);
                    if (mb + o >= p.M) continue;
                    if (half_io) orow_h[(long)o * ldo] = (_Float16)vout[r];
                    else orow[(long)o * ldo] = vout[r];
                }
            }
        }
    }
    CTRACE(4);
    if (CDBG(8)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CTRACE(5); }
}

template <bool HALF_IO>
__global__ void conv_splitk_reduce_kernel(ConvArgs p) {
    const size_t total = (size_t)p.M * p.Cout;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(e / p.Cout), n = (int)(e - (size_t)m * p.Cout);
        float v = 0.f;
#pragma unroll 4                                       // the slabs' loads in flight together (same summation order)
        for (int z = 0; z < p.splitk; ++z) v += p.partial[(size_t)z * total + e];
        v = v * p.scale[n] + p.shift[n];
        const size_t ri = (size_t)(p.res_mod ? m % p.res_mod : m) * p.ldres + n;
        if (p.res) v += HALF_IO ? (float)reinterpret_cast<const _Float16*>(p.res)[ri] : p.res[ri];
        if (p.relu_out) v = fmaxf(v, 0.f);
        if (HALF_IO) reinterpret_cast<_Float16*>(p.out)[(size_t)m * p.ldout + n] = (_Float16)v;
        else p.out[(size_t)m * p.ldout + n] = v;
    }
}

// Cout == 1 (the decoder's mask head, model/modules.py:242): a GEMV per pixel - one wave per output pixel,
// lanes over input channels (float4), wave reduction.  HBM/L2-bound instead of wasting a 64-wide MFMA tile.
template <bool HALF_IN>
__global__ __launch_bounds__(256) void conv_cout1_kernel(ConvArgs p) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= p.M) return;
    const int b = m / p.HoWo, rem = m - b * p.HoWo;
    const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
    float acc = 0.f;
    if (p.KH == 3 && p.KW == 3 && p.Cin <= 256) {
        // 3x3 with one 256-channel chunk per lane group (the mask head: 256 -> 1): the nine taps' loads are requested TOGETHER, from
        // clamped coordinates (a padding tap is selected to zero afterwards).  The general loop below fetches tap after tap behind
        // its bounds branches - nine serialised round trips (round 6).  Same products, same order of additions.
        const int c = lane * 4;
        if (c < p.Cin) {
            f32x4 xv[9], wv[9];
            bool ok[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ih = oh * p.stride - p.pad + t / 3, iw = ow * p.stride - p.pad + t % 3;
                ok[t] = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const int ihc = min(max(ih, 0), p.H - 1), iwc = min(max(iw, 0), p.W - 1);
                xv[t] = *reinterpret_cast<const f32x4*>(p.in + ((size_t)(b * p.H + ihc) * p.W + iwc) * p.ldin + c);
                wv[t] = *reinterpret_cast<const f32x4*>(p.w + (size_t)t * p.Cin + c);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (!ok[t]) continue;                  // (no memory operation below: skipping costs nothing)
                if (HALF_IN) {
                    const h16x8 xh = __builtin_bit_cast(h16x8, xv[t]), wh = __builtin_bit_cast(h16x8, wv[t]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float xe = (float)xh[e];
                        if (p.relu_in) xe = fmaxf(xe, 0.f);
                        acc = fmaf(xe, (float)wh[e], acc);
                    }
                } else {
                    f32x4 x = xv[t];
                    if (p.relu_in) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
                    acc = fmaf(x.x, wv[t].x, acc); acc = fmaf(x.y, wv[t].y, acc); acc = fmaf(x.z, wv[t].z, acc); acc = fmaf(x.w, wv[t].w, acc);
                }
            }
        }
    } else
    for (int kh = 0; kh < p.KH; ++kh) {
        const int ih = oh * p.stride - p.pad + kh;
        if ((unsigned)ih >= (unsigned)p.H) continue;
        for (int kw = 0; kw < p.KW; ++kw) {
            const int iw = ow * p.stride - p.pad + kw;
            if ((unsigned)iw >= (unsigned)p.W) continue;
            const float* x = p.in + ((size_t)(b * p.H + ih) * p.W + iw) * p.ldin;      // HALF_IN: Cin / ldin count 4-byte units (two halfs)
            const float* w = p.w + (size_t)(kh * p.KW + kw) * p.Cin;
            for (int c = lane * 4; c < p.Cin; c += 256) {
                f32x4 xv = *reinterpret_cast<const f32x4*>(x + c);
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c);
                if (HALF_IN) {
                    const h16x8 xh = __builtin_bit_cast(h16x8, xv), wh = __builtin_bit_cast(h16x8, wv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float xe = (float)xh[e];
                        if (p.relu_in) xe = fmaxf(xe, 0.f);
                        acc = fmaf(xe, (float)wh[e], acc);
                    }
                } else {
                    if (p.relu_in) { xv.x = fmaxf(xv.x, 0.f); xv.y = fmaxf(xv.y, 0.f); xv.z = fmaxf(xv.z, 0.f); xv.w = fmaxf(xv.w, 0.f); }
                    acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
                }
            }
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        float v = acc * p.scale[0] + p.shift[0];
        if (p.res) v += p.res[(size_t)(p.res_mod ? m % p.res_mod : m) * p.ldres];
        if (p.relu_out) v = fmaxf(v, 0.f);
        p.out[(size_t)m * p.ldout] = v;
    }
}

// The mask head proper (3x3, stride 1, 256 -> 1 at 1/4 resolution, fp32): one wave per FOUR neighbouring output pixels of a row.  With a
// wave per pixel every wave fetched its own 3 x 3 input vectors and the nine weight vectors: 18 KB through the vector cache for 1 KB of
// new input - 466 MB per launch, 20 us for 60 MFLOP.  Four pixels share a 3 x 6 patch and one set of weights: 27 KB per four pixels
// (3.4x fewer bytes through the cache).  Per pixel the products and the order of the additions are those of conv_cout1_kernel: same bits.
static bool cout1_row4() { static const bool on = !(getenv("XMEM_COUT1_ROW4") && getenv("XMEM_COUT1_ROW4")[0] == '0'); return on; }   // tools: A/B
__global__ __launch_bounds__(256) void conv_cout1_row4_kernel(ConvArgs p) {
    const int lane = threadIdx.x & 63;
    const int gpr = (p.Wo + 3) >> 2;                              // groups of four pixels per output row
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= p.B * p.Ho * gpr) return;
    const int g = wid % gpr, row = wid / gpr;                     // row = b * Ho + oh
    const int oh = row % p.Ho, b = row / p.Ho;
    const int ow0 = 4 * g;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int c = lane * 4;
    if (c < p.Cin) {
        f32x4 xv[3][6], wv[9];
        bool okr[3], okc[6];
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) okc[cc] = (unsigned)(ow0 - p.pad + cc) < (unsigned)p.W;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ih = oh - p.pad + r;
            okr[r] = (unsigned)ih < (unsigned)p.H;
            const int ihc = min(max(ih, 0), p.H - 1);
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) {                      // unconditional loads from clamped coordinates, all requested together
                const int iwc = min(max(ow0 - p.pad + cc, 0), p.W - 1);
                xv[r][cc] = *reinterpret_cast<const f32x4*>(p.in + ((size_t)(b * p.H + ihc) * p.W + iwc) * p.ldin + c);
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(p.w + (size_t)t * p.Cin + c);
        if (p.relu_in) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) {
                    f32x4& x = xv[r][cc];
                    x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f);
                }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int r = t / 3, cc = j + t % 3;
                if (!(okr[r] && okc[cc])) continue;               // a padding tap
                const f32x4 x = xv[r][cc];
                acc[j] = fmaf(x.x, wv[t].x, acc[j]); acc[j] = fmaf(x.y, wv[t].y, acc[j]);
                acc[j] = fmaf(x.z, wv[t].z, acc[j]); acc[j] = fmaf(x.w, wv[t].w, acc[j]);
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = wave_sum(acc[j]);
    if (lane < 4 && ow0 + lane < p.Wo) {
        const float a = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
        const int m = row * p.Wo + ow0 + lane;
        float v = a * p.scale[0] + p.shift[0];
        if (p.res) v += p.res[(size_t)(p.res_mod ? m % p.res_mod : m) * p.ldres];
        if (p.relu_out) v = fmaxf(v, 0.f);
        p.out[(size_t)m * p.ldout] = v;
    }
}

// ----------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) for the 3x3 / stride 1 / pad 1 convolutions (85 % of the network's FLOPs):
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      (Lavin & Gray) - 16 multiplies per 2x2 outputs instead of 36.
// The element-wise product over channels is 16 independent GEMMs [tiles x Cin] x [Cin x Cout] that run on the same
// MFMA kernel (blockIdx.y = tile position), between two HBM-bound transform kernels.  All fp32.
// ----------------------------------------------------------------------------------------------
__global__ void wino_input_kernel(const float* __restrict__ in, int ldin, int B, int H, int W, int C, int th, int tw,
                                  int relu_in, float* __restrict__ V, int split) {
    const int C4 = C >> 2;
    const size_t P = (size_t)B * th * tw;
    const size_t total = P * C4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % C4);
        size_t t = e / C4;
        const int tx = (int)(t % tw); size_t r = t / tw;
        const int ty = (int)(r % th);
        const int b = (int)(r / th);
        f32x4 d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ih = 2 * ty - 1 + i;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iw = 2 * tx - 1 + j;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                    v = *reinterpret_cast<const f32x4*>(in + (((size_t)b * H + ih) * W + iw) * ldin + c4 * 4);
                d[i][j] = v;
            }
        }
        // relu AFTER every load has been requested: a relu inside the load loop sits behind a (uniform) branch of its own, which
        // makes each load wait for its predecessor's max - 16 / 36 serialised round trips (round 6: 17.3 -> 10 us at 30x54x512)
        if (relu_in) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = d[i][j];
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    d[i][j] = v;
                }
        }
        f32x4 u[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {          // B^T d (rows)
            u[0][j] = d[0][j] - d[2][j];
            u[1][j] = d[1][j] + d[2][j];
            u[2][j] = d[2][j] - d[1][j];
            u[3][j] = d[1][j] - d[3][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {          // (B^T d) B (columns)
            f32x4 v0 = u[i][0] - u[i][2], v1 = u[i][1] + u[i][2], v2 = u[i][2] - u[i][1], v3 = u[i][1] - u[i][3];
            if (split) { v0 = split_pack(v0); v1 = split_pack(v1); v2 = split_pack(v2); v3 = split_pack(v3); }   // 'fp32x' mode
            float* o = V + ((size_t)(i * 4) * P + t) * C + c4 * 4;
            *reinterpret_cast<f32x4*>(o) = v0;
            *reinterpret_cast<f32x4*>(o + P * C) = v1;
            *reinterpret_cast<f32x4*>(o + 2 * P * C) = v2;
            *reinterpret_cast<f32x4*>(o + 3 * P * C) = v3;
        }
    }
}

__global__ void wino_output_kernel(const float* __restrict__ Mt, int B, int Ho, int Wo, int Cout, int th, int tw,
                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                   const float* __restrict__ res, int ldres, int res_bcast, int relu_out, float* __restrict__ out, int ldout) {
    const int N4 = Cout >> 2;
    const size_t P = (size_t)B * th * tw;
    const size_t total = P * N4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int n4 = (int)(e % N4);
        size_t t = e / N4;
        const int tx = (int)(t % tw); size_t r = t / tw;
        const int ty = (int)(r % th);
        const int b = (int)(r / th);
        // the residual values are requested FIRST, together with the M planes: fetched inside the store loop each one sits behind the
        // bounds branches of its pixel and costs a round trip of its own (round 6: 13.4 -> ~9 us at 60x108x256 with a residual)
        f32x4 rv[2][2];
        if (res) {
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int oh = min(2 * ty + dy, Ho - 1), ow = min(2 * tx + dx, Wo - 1);      // clamped: never stored when outside
                    const size_t pix = ((size_t)b * Ho + oh) * Wo + ow;
                    rv[dy][dx] = *reinterpret_cast<const f32x4*>(res + (res_bcast ? (size_t)oh * Wo + ow : pix) * ldres + n4 * 4);
                }
        }
        f32x4 m[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                m[i][j] = *reinterpret_cast<const f32x4*>(Mt + ((size_t)(i * 4 + j) * P + t) * Cout + n4 * 4);
        f32x4 s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {          // A^T m
            s0[j] = m[0][j] + m[1][j] + m[2][j];
            s1[j] = m[1][j] - m[2][j] - m[3][j];
        }
        f32x4 y[2][2];
        y[0][0] = s0[0] + s0[1] + s0[2]; y[0][1] = s0[1] - s0[2] - s0[3];
        y[1][0] = s1[0] + s1[1] + s1[2]; y[1][1] = s1[1] - s1[2] - s1[3];
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + n4 * 4);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + n4 * 4);
        // two phases (see the epilogue of conv_mfma_kernel): every loaded operand is consumed before the first store, so no store waits
        // for its predecessor (loads and stores share the in-order vmcnt counter; a wait inside a bounds branch is a vmcnt(0))
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                f32x4 v = y[dy][dx] * sc + sh;
                if (res) v += rv[dy][dx];
                if (relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                y[dy][dx] = v;
            }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int oh = 2 * ty + dy;
            if (oh >= Ho) continue;
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int ow = 2 * tx + dx;
                if (ow >= Wo) continue;
                const size_t pix = ((size_t)b * Ho + oh) * Wo + ow;
                *reinterpret_cast<f32x4*>(out + pix * ldout + n4 * 4) = y[dy][dx];
            }
        }
    }
}

// Fused Winograd-domain GEMM + output transform.  One workgroup owns a (tile-pixel block x Cout block) and walks all 16
// tile positions xi = (i, j) as ONE long K loop of 16 * Cin/BK staged tiles: after each position the 32x32 accumulator
// M_xi is folded into the four output-position accumulators  Y[a][b] += At[a][i] * At[b][j] * M_xi  (At entries are
// 0 / +-1), so the [16][tiles][Cout] intermediate never exists and prologue / epilogue are amortised over a 16x deeper
// loop than the per-position GEMMs.  A = V[xi][m][k] (from wino_input_kernel), B = U[xi][n][k] = (G g G^T).
struct WinoArgs {
    const float* V; const float* U; const float* scale; const float* shift; const float* res; float* out;
    int P, Cin, Cout, B, Ho, Wo, th, tw, ldout, ldres, relu_out, nk, tiles_m, tiles_n, res_bcast;
};

template <int BM, int BN, int TM, int TN, int BK>
__global__ __launch_bounds__(256) void wino_fused_kernel(WinoArgs p) {
    constexpr int WN = BN / (32 * TN);
    constexpr int WM = BM / (32 * TM);
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int LDK = BK + 4;
    constexpr int C4 = BK / 4;
    constexpr int RPP = 256 / C4;
    constexpr int RA = BM / RPP, RB = BN / RPP;
    constexpr int BUF = (BM + BN) * LDK;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int lrow = tid / C4, c4 = tid % C4;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const size_t vstride = (size_t)p.P * p.Cin, ustride = (size_t)p.Cout * p.Cin;
    f32x4 ra[RA], rb[RB];
    auto load_tile = [&](int it) {
        const int xi = it / p.nk, kt = it - xi * p.nk;
        const int k = kt * BK + c4 * 4;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const float* va = p.V + (size_t)xi * vstride + k;
        const float* ub = p.U + (size_t)xi * ustride + k;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = m0 + lrow + RPP * i;
            ra[i] = (m < p.P && k < p.Cin) ? *reinterpret_cast<const f32x4*>(va + (size_t)m * p.Cin) : zero;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int n = n0 + lrow + RPP * i;
            rb[i] = (n < p.Cout && k < p.Cin) ? *reinterpret_cast<const f32x4*>(ub + (size_t)n * p.Cin) : zero;
        }
    };
    auto store_tile = [&](int buf) {
        float* As = smem + buf * BUF;
        float* Bs = As + BM * LDK;
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(&As[(lrow + RPP * i) * LDK + c4 * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(&Bs[(lrow + RPP * i) * LDK + c4 * 4]) = rb[i];
    };

    f32x16 acc[TM][TN];
    f32x16 y[4][TM][TN];                       // output positions (a, b) = (o >> 1, o & 1)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int r = 0; r < 16; ++r) y[o][i][j][r] = 0.f;
        }

    const int total = 16 * p.nk;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int kt_in_xi = 0, xi = 0;
    for (int it = 0; it < total; ++it) {
        const int buf = it & 1;
        const bool has_next = it + 1 < total;
        if (has_next) load_tile(it + 1);
        const float* As = smem + buf * BUF + (wm * 32 * TM + l31) * LDK + lh * 4;
        const float* Bs = smem + buf * BUF + BM * LDK + (wn * 32 * TN + l31) * LDK + lh * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDK + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDK + kk * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        if (++kt_in_xi == p.nk) {              // position finished: Y[a][b] += At[a][i] * At[b][j] * M_xi
            const int wi = xi >> 2, wj = xi & 3;
            // At = [1 1 1 0; 0 1 -1 -1]
            const float a0 = (wi < 3) ? 1.f : 0.f, a1 = (wi == 0) ? 0.f : (wi == 1 ? 1.f : -1.f);
            const float b0 = (wj < 3) ? 1.f : 0.f, b1 = (wj == 0) ? 0.f : (wj == 1 ? 1.f : -1.f);
            const float c00 = a0 * b0, c01 = a0 * b1, c10 = a1 * b0, c11 = a1 * b1;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[i][j][r];
                        y[0][i][j][r] = fmaf(c00, v, y[0][i][j][r]);
                        y[1][i][j][r] = fmaf(c01, v, y[1][i][j][r]);
                        y[2][i][j][r] = fmaf(c10, v, y[2][i][j][r]);
                        y[3][i][j][r] = fmaf(c11, v, y[3][i][j][r]);
                        acc[i][j][r] = 0.f;
                    }
                }
            kt_in_xi = 0; ++xi;
        }
        if (has_next) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue: lane owns output channel n and 16 tile-pixels per 32x32 tile; each tile-pixel is 2x2 outputs
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * 32 * TN + j * 32 + l31;
        if (n >= p.Cout) continue;
        const float sc = p.scale[n], sh = p.shift[n];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= p.P) continue;
                const int tx = m % p.tw; const int t2 = m / p.tw;
                const int ty = t2 % p.th; const int b = t2 / p.th;
                // two phases (see the epilogue of conv_mfma_kernel): residual values first (clamped addresses), then the four stores
                float vv[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int oh = min(2 * ty + (o >> 1), p.Ho - 1), ow = min(2 * tx + (o & 1), p.Wo - 1);
                    const size_t pix = ((size_t)b * p.Ho + oh) * p.Wo + ow;
                    float v = y[o][i][j][r] * sc + sh;
                    if (p.res) v += p.res[(p.res_bcast ? (size_t)oh * p.Wo + ow : pix) * p.ldres + n];
                    vv[o] = p.relu_out ? fmaxf(v, 0.f) : v;
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int oh = 2 * ty + (o >> 1), ow = 2 * tx + (o & 1);
                    if (oh >= p.Ho || ow >= p.Wo) continue;
                    const size_t pix = ((size_t)b * p.Ho + oh) * p.Wo + ow;
                    p.out[pix * p.ldout + n] = vv[o];
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Winograd F(4x4, 3x3) for the LARGE 3x3 / stride 1 layers (plan_tile 17..28): 36 multiplies per 4x4 outputs instead of
// 144 - 4x fewer MFMA FLOPs than the direct form (F(2x2): 2.25x) and 2.25 / 4 of F(2x2)'s transform-domain traffic
// (36 values per 16 outputs instead of 16 per 4).  All fp32.
//   V = B^T d B (6x6 input tile, stride 4), M_xi = V_xi U_xi^T on the same MFMA GEMM (36 groups), Y = A^T M A (4x4).
// INTERPOLATION POINTS {0, +-3/4, +-3/2, inf} (round 5; rounds 2-4: the textbook {0, +-1, +-2, inf}).  The error of a Winograd
// convolution is set by the magnitudes in A, B, G, i.e. by the points (Barabasz et al., "Error analysis and improving the accuracy
// of Winograd convolution for DNNs", 2018): with +-1, +-2 the output transform multiplies by up to 8 and the input transform by up to
// 5; with +-3/4, +-3/2 every coefficient of A^T is <= 3.375 and of B^T <= 2.8125.  Simulated in fp32 against an fp64 convolution
// (relu'd normal inputs, Cin 64 / 512 / 1600): max error / max |y| 2.5e-6 / 5.0e-6 / 8.6e-6 -> 7.4e-7 / 1.4e-6 / 2.3e-6 (3.4-3.7x
// lower, rms 2x lower) - the level of F(2x2) and of the direct form's own fp32 summation.  It matters: on the 3-object 480p clip
// (multi-object conditioning) the textbook points cost 73 argmax pixels against the oracle where the reference disagrees with
// itself on 20 and F(2x2) on 1 (profiles/r05_c3_parity_by_plan.txt).  All powers of 3/4 and 3/2 are exact in fp32, so B^T and A^T
// below are exact constants; the non-dyadic factors (1 / N_i) live in G, applied once at load time in fp64 (ops.winograd4_weights).
// Rows of B^T = coefficients of the monic polynomials prod_{k != i} (x - p_k); rows of A^T = powers of the points:
//   p = (0, a, -a, b, -b, inf), a = 3/4, b = 3/2
// ----------------------------------------------------------------------------------------------
#define W4_A 0.75f
#define W4_B 1.5f
#define W4_A2 0.5625f          // a^2
#define W4_B2 2.25f            // b^2
#define W4_A3 0.421875f        // a^3
#define W4_B3 3.375f           // b^3
#define W4_A2B2 1.265625f      // a^2 b^2
#define W4_SUM 2.8125f         // a^2 + b^2
#define W4_AB2 1.6875f         // a b^2
#define W4_BA2 0.84375f        // b a^2
__device__ __forceinline__ void wino4_bt(const f32x4* d, f32x4* t) {      // t = B^T d for one 6-vector
    t[0] = W4_A2B2 * d[0] - W4_SUM * d[2] + d[4];
    t[1] = -W4_AB2 * d[1] - W4_B2 * d[2] + W4_A * d[3] + d[4];
    t[2] = W4_AB2 * d[1] - W4_B2 * d[2] - W4_A * d[3] + d[4];
    t[3] = -W4_BA2 * d[1] - W4_A2 * d[2] + W4_B * d[3] + d[4];
    t[4] = W4_BA2 * d[1] - W4_A2 * d[2] - W4_B * d[3] + d[4];
    t[5] = W4_A2B2 * d[1] - W4_SUM * d[3] + d[5];
}
__device__ __forceinline__ void wino4_at(const f32x4* m, f32x4* s) {      // s = A^T m for one 6-vector
    s[0] = m[0] + m[1] + m[2] + m[3] + m[4];
    s[1] = W4_A * m[1] - W4_A * m[2] + W4_B * m[3] - W4_B * m[4];
    s[2] = W4_A2 * m[1] + W4_A2 * m[2] + W4_B2 * m[3] + W4_B2 * m[4];
    s[3] = W4_A3 * m[1] - W4_A3 * m[2] + W4_B3 * m[3] - W4_B3 * m[4] + m[5];
}

__global__ __launch_bounds__(256) void wino4_input_kernel(const float* __restrict__ in, int ldin, int B, int H, int W, int C, int th, int tw,
                                                          int relu_in, float* __restrict__ V, int split) {
    const int C4 = C >> 2;
    const size_t P = (size_t)B * th * tw;
    const size_t total = P * C4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % C4);
        size_t t = e / C4;
        const int tx = (int)(t % tw); size_t r = t / tw;
        const int ty = (int)(r % th);
        const int b = (int)(r / th);
        f32x4 d[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int ih = 4 * ty - 1 + i;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int iw = 4 * tx - 1 + j;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                    v = *reinterpret_cast<const f32x4*>(in + (((size_t)b * H + ih) * W + iw) * ldin + c4 * 4);
                d[i][j] = v;
            }
        }
        if (relu_in) {                                 // after the loads, not between them (see wino_input_kernel)
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    f32x4 v = d[i][j];
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    d[i][j] = v;
                }
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {              // columns: d[:, j] <- B^T d[:, j]
            f32x4 col[6], tc[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) col[i] = d[i][j];
            wino4_bt(col, tc);
#pragma unroll
            for (int i = 0; i < 6; ++i) d[i][j] = tc[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {              // rows: (B^T d) B, stored to the 36 position planes
            f32x4 tr[6];
            wino4_bt(d[i], tr);
#pragma unroll
            for (int j = 0; j < 6; ++j)
                *reinterpret_cast<f32x4*>(V + ((size_t)(i * 6 + j) * P + t) * C + c4 * 4) = split ? split_pack(tr[j]) : tr[j];
        }
    }
}

__global__ __launch_bounds__(256) void wino4_output_kernel(const float* __restrict__ Mt, int B, int Ho, int Wo, int Cout, int th, int tw,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ res, int ldres, int res_bcast, int relu_out,
                                                           float* __restrict__ out, int ldout) {
    const int N4 = Cout >> 2;
    const size_t P = (size_t)B * th * tw;
    const size_t total = P * N4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int n4 = (int)(e % N4);
        size_t t = e / N4;
        const int tx = (int)(t % tw); size_t r = t / tw;
        const int ty = (int)(r % th);
        const int b = (int)(r / th);
        f32x4 rv[4][4];                            // residual values, requested before the M planes (see wino_output_kernel)
        if (res) {
#pragma unroll
            for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                for (int dx = 0; dx < 4; ++dx) {
                    const int oh = min(4 * ty + dy, Ho - 1), ow = min(4 * tx + dx, Wo - 1);      // clamped: never stored when outside
                    const size_t pix = ((size_t)b * Ho + oh) * Wo + ow;
                    rv[dy][dx] = *reinterpret_cast<const f32x4*>(res + (res_bcast ? (size_t)oh * Wo + ow : pix) * ldres + n4 * 4);
                }
        }
        f32x4 s[4][6];                             // A^T m, column by column
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            f32x4 col[6], sc[4];
#pragma unroll
            for (int i = 0; i < 6; ++i) col[i] = *reinterpret_cast<const f32x4*>(Mt + ((size_t)(i * 6 + j) * P + t) * Cout + n4 * 4);
            wino4_at(col, sc);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i][j] = sc[i];
        }
        const f32x4 scl = *reinterpret_cast<const f32x4*>(scale + n4 * 4);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + n4 * 4);
        // two phases (see the epilogue of conv_mfma_kernel): the 16 outputs are finished in the residual registers first ...
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            f32x4 y[4];
            wino4_at(s[dy], y);
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                f32x4 v = y[dx] * scl + sh;
                if (res) v += rv[dy][dx];
                if (relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                rv[dy][dx] = v;
            }
        }
        // ... then stored: no store depends on a load any more, none waits for its predecessor
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            const int oh = 4 * ty + dy;
            if (oh >= Ho) continue;
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                const int ow = 4 * tx + dx;
                if (ow >= Wo) continue;
                const size_t pix = ((size_t)b * Ho + oh) * Wo + ow;
                *reinterpret_cast<f32x4*>(out + pix * ldout + n4 * 4) = rv[dy][dx];
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// REDUCED-PRECISION MODE (opt-in, plan_tile 16): the Winograd-domain operands V = B^T d B and U = G g G^T are stored in
// fp16 and contracted on v_mfma_f32_32x32x16_f16 (fp32 accumulation, 16x the fp32 MFMA rate); transforms, epilogue and
// every other kernel stay fp32.  Mirrors the reference's fp16-autocast GPU loop (inference/run_on_video.py:76); it is
// outside the fp32 parity contract and is reported under its own bench key.
// ----------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__global__ void wino_input_f16_kernel(const float* __restrict__ in, int ldin, int B, int H, int W, int C, int th, int tw,
                                      int relu_in, _Float16* __restrict__ V) {
    const int C4 = C >> 2;
    const size_t P = (size_t)B * th * tw;
    const size_t total = P * C4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % C4);
        size_t t = e / C4;
        const int tx = (int)(t % tw); size_t r = t / tw;
        const int ty = (int)(r % th);
        const int b = (int)(r / th);
        f32x4 d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ih = 2 * ty - 1 + i;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iw = 2 * tx - 1 + j;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
                    v = *reinterpret_cast<const f32x4*>(in + (((size_t)b * H + ih) * W + iw) * ldin + c4 * 4);
                    if (relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                }
                d[i][j] = v;
            }
        }
        f32x4 u[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u[0][j] = d[0][j] - d[2][j];
            u[1][j] = d[1][j] + d[2][j];
            u[2][j] = d[2][j] - d[1][j];
            u[3][j] = d[1][j] - d[3][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v[4] = {u[i][0] - u[i][2], u[i][1] + u[i][2], u[i][2] - u[i][1], u[i][1] - u[i][3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f16x4 hv;
                hv.x = (_Float16)v[j].x; hv.y = (_Float16)v[j].y; hv.z = (_Float16)v[j].z; hv.w = (_Float16)v[j].w;
                *reinterpret_cast<f16x4*>(V + ((size_t)(i * 4 + j) * P + t) * C + c4 * 4) = hv;
            }
        }
    }
}

// M[xi][m][n] = sum_k V[xi][m][k] * U[xi][n][k]  (fp16 x fp16 -> fp32).  One workgroup: BM x BN outputs of one tile position,
// 4 waves (2 x 2), wave tile (BM/2) x (BN/2) of 32x32 MFMA blocks, BK = 64 halfs per stage: register-staged global loads one
// stage ahead, single LDS buffer (144-byte rows: an odd multiple of 16 B -> conflict-free ds_read_b128 fragments).
struct WinoF16Args {
    const _Float16* V; const _Float16* U; float* M;
    int P, Cin, Cout, tiles_m, tiles_n;
};

template <int BM, int BN>
__global__ __launch_bounds__(256) void wino_gemm_f16_kernel(WinoF16Args p) {
    constexpr int BK = 64, LDB = 144;                 // bytes per LDS row
    constexpr int TM = BM / 64, TN = BN / 64;         // 32x32 blocks per wave in M / N
    constexpr int CA = BM * 8 / 256, CB = BN * 8 / 256;   // 16-byte chunks per thread per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    unsigned char* As = smem16;
    unsigned char* Bs = smem16 + BM * LDB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int xi = blockIdx.y;
    const unsigned char* gA = reinterpret_cast<const unsigned char*>(p.V + (size_t)xi * p.P * p.Cin);
    const unsigned char* gB = reinterpret_cast<const unsigned char*>(p.U + (size_t)xi * p.Cout * p.Cin);
    const size_t rowb = (size_t)p.Cin * 2;            // bytes per operand row

    size_t a_off[CA], b_off[CB];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
        const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
        a_off[i] = (size_t)min(m0 + row, p.P - 1) * rowb + ch * 16;       // rows past M are clamped (never stored)
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
        const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
        b_off[i] = (size_t)min(n0 + row, p.Cout - 1) * rowb + ch * 16;
    }
    f32x4 ra[CA], rb[CB];                             // raw 16-byte chunks
    auto load_stage = [&](int kt) {
        const size_t kb = (size_t)kt * BK * 2;
#pragma unroll
        for (int i = 0; i < CA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(gA + a_off[i] + kb);
#pragma unroll
        for (int i = 0; i < CB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(gB + b_off[i] + kb);
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int i = 0; i < CA; ++i) { const int c = tid + 256 * i; *reinterpret_cast<f32x4*>(As + (c >> 3) * LDB + (c & 7) * 16) = ra[i]; }
#pragma unroll
        for (int i = 0; i < CB; ++i) { const int c = tid + 256 * i; *reinterpret_cast<f32x4*>(Bs + (c >> 3) * LDB + (c & 7) * 16) = rb[i]; }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.Cin / BK;
    load_stage(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                              // previous stage's fragment reads are done
        store_stage();
        __syncthreads();
        if (kt + 1 < nk) load_stage(kt + 1);          // next stage in flight under this stage's MFMAs
        const unsigned char* a0 = As + (wm * (BM / 2) + l31) * LDB + lh * 16;
        const unsigned char* b0 = Bs + (wn * (BN / 2) + l31) * LDB + lh * 16;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            f16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f16x8*>(a0 + i * 32 * LDB + kk * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f16x8*>(b0 + j * 32 * LDB + kk * 32);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    float* gM = p.M + (size_t)xi * p.P * p.Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + l31;
        if (n >= p.Cout) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * (BM / 2) + i * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m < p.P) gM[(size_t)m * p.Cout + n] = acc[i][j][r];
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------
namespace {

struct Plan { int bm, bn, bk, splitk, kt_per_split, nk; bool generic; bool wino; int fused; bool f16; bool wino4; bool split;
              int stream, sv, sring; int half; };      // stream: 1 = GEMM on the streaming kernel (csrc/gemm_stream.hip), tile variant sv, ring sring

inline bool wino_ok(const xmem_conv_desc* d) {
    return d->w_winograd && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Cin % 32 == 0 && d->Cout % 4 == 0 &&
           d->ldout % 4 == 0 && (!d->res || d->ldres % 4 == 0) && (((uintptr_t)d->out) & 15) == 0 && (!d->res || (((uintptr_t)d->res) & 15) == 0);
}

int validate(const xmem_conv_desc* d) {
    if (!d || !d->in || !d->w || !d->scale || !d->shift || !d->out) return XMEM_ERR_BAD_ARG;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 ||
        d->stride <= 0 || d->pad < 0) return XMEM_ERR_BAD_ARG;
    if (d->Cin % 4 != 0 || d->ldin % 4 != 0 || d->ldin < d->Cin || d->ldout < d->Cout) return XMEM_ERR_UNSUPPORTED;
    if (d->res && d->ldres < d->Cout) return XMEM_ERR_BAD_ARG;
    if ((d->H + 2 * d->pad - d->KH) < 0 || (d->W + 2 * d->pad - d->KW) < 0) return XMEM_ERR_BAD_ARG;
    if (d->plan_tile < 0 || d->plan_tile > 40 || d->plan_splitk < 0) return XMEM_ERR_BAD_ARG;
    return XMEM_OK;
}

inline void out_dims(const xmem_conv_desc* d, int& Ho, int& Wo) {
    Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
    Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
}

// plan_tile: 0 = heuristic, 1..6 = {128x128, 128x64, 64x64} x {BK 32, BK 64}
Plan make_plan(const xmem_conv_desc* d) {
    int Ho, Wo; out_dims(d, Ho, Wo);
    const int M = d->B * Ho * Wo, K = d->KH * d->KW * d->Cin;
    Plan pl;
    pl.bk = 32;
    pl.generic = false;
    pl.wino = false;
    pl.fused = 0;
    pl.f16 = false;
    pl.wino4 = false;
    pl.stream = 0; pl.sv = 0; pl.sring = 3; pl.half = 0;
    // split-operand arithmetic ('fp32x', opt-in): every GEMM-shaped path; the Cout = 1 GEMV stays fp32 (it is HBM-bound)
    pl.split = d->arith == 1 && d->w_split != nullptr;
    if (d->Cout == 1) { pl.split = false; pl.bm = 0; pl.bn = 0; pl.nk = cdiv(K, 32); pl.splitk = 1; pl.kt_per_split = pl.nk; return pl; }   // GEMV path
    auto tiles = [&](int bm, int bn) { return (long)cdiv(M, bm) * cdiv(d->Cout, bn); };
    if (d->plan_tile > 0) {
        static const int cfg[6][3] = {{128, 128, 32}, {128, 64, 32}, {64, 64, 32}, {128, 128, 64}, {128, 64, 64}, {64, 64, 64}};
        int t = d->plan_tile;
        if (t >= 23) {
            // 23..28: F(4x4) with the position GEMMs on the streaming kernel, 29..34: F(2x2) likewise, 35..40: the pointwise
            // (1x1, pad 0) convolution itself on it; within a group: tile {64x64, 128x64, 128x128} x ring {3, 4} stages.
            // Shapes the streaming kernel does not take (split arithmetic, Cin % 32 != 0, 1x1 with padding) fall back to the
            // 64x64 tile of the corresponding classic plan.
            const int grp = (t - 23) / 6, v = (t - 23) % 6;
            const bool ok = !pl.split && d->Cin % 32 == 0 &&
                            (grp == 2 ? (d->KH == 1 && d->KW == 1 && d->pad == 0 &&
                                         (double)d->B * d->H * d->W * d->ldin * 4.0 < 2.0e9 && (double)d->Cout * d->Cin * 4.0 < 2.0e9)
                                      : (wino_ok(d) && (grp == 0 ? d->w_winograd4 != nullptr : true)));
            if (ok) { pl.stream = 1; pl.sv = v % 3; pl.sring = 3 + v / 3; }
            t = grp == 0 ? 19 : (grp == 1 ? 9 : 3);
        }
        if (t >= 17) {                     // F(4x4, 3x3) with the GEMM tile of plan t - 10; F(2x2) when its operand is absent
            if (wino_ok(d) && (pl.split ? d->w_winograd4_split : (const void*)d->w_winograd4)) pl.wino4 = true;
            t -= 10;
        }
        if (pl.split && t == 16) t = 9;    // the fp16-storage mode and the split mode exclude each other
        if (pl.split && t > 12) t = (t == 13) ? 8 : 9;   // fused GEMM + output transform has no split variant: separate transform
        if (t == 16) {                     // reduced-precision Winograd (opt-in); falls back to the fp32 Winograd tile 64x64
            if (wino_ok(d) && d->w_winograd_f16 && d->Cin % 64 == 0) {
                pl.wino = true; pl.f16 = true; pl.bm = 0; pl.bn = 0; pl.nk = d->Cin / 64; pl.splitk = 1; pl.kt_per_split = pl.nk;
                return pl;
            }
            t = 9;
        }
        if (t > 12) {                      // 13..15: fused Winograd GEMM + output transform {128x64, 64x64, 64x128}, BK 32
            if (wino_ok(d)) { pl.wino = true; pl.fused = t - 12; }
            t = (t == 13) ? 2 : 3;         // fall back to a direct tile when Winograd is not applicable
        }
        if (t > 6 && wino_ok(d) && (!pl.split || d->w_winograd_split)) { pl.wino = true; }
        if (t > 6) t -= 6;
        pl.bm = cfg[t - 1][0]; pl.bn = cfg[t - 1][1]; pl.bk = cfg[t - 1][2];
        if (pl.wino) {           // the GEMM runs per tile position: M = tiles, K = Cin, no split-K
            pl.generic = (d->Cin % pl.bk) != 0; pl.nk = cdiv(d->Cin, pl.bk); pl.splitk = 1; pl.kt_per_split = pl.nk;
            return pl;
        }
    } else {
        // 256 CUs; two resident workgroups per CU is the sweet spot for the 128-wide tiles
        if (d->Cout > 64 && tiles(128, 128) >= 384) { pl.bm = 128; pl.bn = 128; }
        else if (tiles(128, 64) >= 384) { pl.bm = 128; pl.bn = 64; }
        else { pl.bm = 64; pl.bn = 64; }
    }
    pl.generic = (d->Cin % pl.bk) != 0;
    pl.nk = cdiv(K, pl.bk);
    pl.splitk = 1;
    if (d->plan_splitk > 0) {
        pl.splitk = d->plan_splitk < pl.nk ? d->plan_splitk : pl.nk;
    } else {
        const long t = tiles(pl.bm, pl.bn);
        if (t < 384) {
            int want = (int)cdiv(512, (int)t);
            int maxs = pl.nk / (256 / pl.bk); if (maxs < 1) maxs = 1;
            pl.splitk = want < maxs ? want : maxs;
            if (pl.splitk > 16) pl.splitk = 16;
            if (pl.splitk < 1) pl.splitk = 1;
        }
    }
    if (pl.stream) pl.splitk = 1;                  // the streaming kernel contracts the whole K of a unit
    pl.kt_per_split = cdiv(pl.nk, pl.splitk);
    pl.splitk = cdiv(pl.nk, pl.kt_per_split);
    return pl;
}

// the Winograd-domain position GEMMs M[xi] = V[xi] U[xi]^T on the streaming kernel: G positions of [P x Cin] x [Cin x Cout]
int launch_stream_positions(const Plan& pl, const float* V, const float* U, float* Mt, int G, int P, int Cin, int Cout, hipStream_t s) {
    GemmStreamArgs g = {};
    g.A = V; g.B = U; g.C = Mt;
    g.a_gstride = (long)P * Cin; g.b_gstride = (long)Cout * Cin; g.c_gstride = (long)P * Cout;
    g.M = P; g.N = Cout; g.K = Cin; g.G = G;
    g.lda = Cin; g.ldb = Cin; g.ldc = Cout;
    g.mode = 0; g.stride = 1;
    return gemm_stream_launch(g, pl.sv, pl.sring, s);
}

// the 1x1 fast path needs 32-bit byte offsets into both operands (per group)
// Transform kernels are grid-stride over (tile, channel quad) items, one item per thread.  A 1/16-resolution layer has ~15 000 items:
// as 256-thread workgroups that is ~60 workgroups on 60 of the 256 CUs, each bound by its own CU's load / store path (measured
// 10-17 us for 12 MB).  Below 512 workgroups of 256 the items are dealt as 64-thread workgroups instead: four times the CUs.
static inline void transform_grid(size_t items, int& blocks, int& threads) {
    threads = items < (size_t)512 * 256 ? 64 : 256;
    size_t b = (items + threads - 1) / threads;
    blocks = (int)(b > 16384 ? 16384 : b);
}

static bool conv_is_one(const ConvArgs& a) {
    static const int off = getenv("XMEM_CONV_ONE") && getenv("XMEM_CONV_ONE")[0] == '0';
    if (off) return false;
    return a.KH == 1 && a.KW == 1 && a.pad == 0 &&
           (double)a.B * a.H * a.W * a.ldin * 4.0 < 4.0e9 && (double)a.Cout * a.K * 4.0 < 4.0e9;
}

template <int BM, int BN, int TM, int TN, int BK, bool G>
int launch_cfg(const ConvArgs& a, hipStream_t s, int groups = 1, bool split = false, int half = 0) {
    // (a workgroup that contracts a single k-tile never touches the second buffer: half the LDS, twice the resident workgroups)
    const size_t lds = (a.kt_per_split == 1 ? 1 : 2) * (size_t)(BM + BN) * (BK + 4) * sizeof(float);
    auto kern = conv_mfma_kernel<BM, BN, TM, TN, BK, G, false, false>;
    if (half && BK == 32) {                        // fp16 loop: half operands (HALF 1: fp32 output, 2: half output + residual)
        const bool one = !G && conv_is_one(a);
        if (half == 2) kern = one ? conv_mfma_kernel<BM, BN, TM, TN, 32, false, true, false, 2> : conv_mfma_kernel<BM, BN, TM, TN, 32, G, false, false, 2>;
        else kern = one ? conv_mfma_kernel<BM, BN, TM, TN, 32, false, true, false, 1> : conv_mfma_kernel<BM, BN, TM, TN, 32, G, false, false, 1>;
    } else if (split) {
        kern = conv_mfma_kernel<BM, BN, TM, TN, BK, G, false, true>;
        if (!G && conv_is_one(a)) kern = conv_mfma_kernel<BM, BN, TM, TN, BK, false, true, true>;
    } else if (!G && conv_is_one(a)) kern = conv_mfma_kernel<BM, BN, TM, TN, BK, false, true, false>;
    if (xmem_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != XMEM_OK) return XMEM_ERR_LAUNCH;
    dim3 grid(a.tiles_m * a.tiles_n, groups, a.splitk);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    return xmem_check_launch();
}

// fp16 loop only: 256x128 tile, 8 waves (4 x 2, each 64x64 = four accumulators): 1/85 byte of operand traffic per FLOP instead of
// 1/64 - the half kernels are bound by the L2 -> LDS operand stream, not by the matrix pipe
template <bool G>
int launch_half_256(const ConvArgs& a, hipStream_t s, int half) {
    const size_t lds = 2 * (size_t)(256 + 128) * 36 * sizeof(float);
    const bool one = !G && conv_is_one(a);
    auto kern = conv_mfma_kernel<256, 128, 2, 2, 32, G, false, false, 2, 8>;
    if (half == 2) { if (one) kern = conv_mfma_kernel<256, 128, 2, 2, 32, false, true, false, 2, 8>; }
    else kern = one ? conv_mfma_kernel<256, 128, 2, 2, 32, false, true, false, 1, 8> : conv_mfma_kernel<256, 128, 2, 2, 32, G, false, false, 1, 8>;
    if (xmem_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != XMEM_OK) return XMEM_ERR_LAUNCH;
    dim3 grid(a.tiles_m * a.tiles_n, 1, a.splitk);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, a);
    return xmem_check_launch();
}

template <int BK, bool G>
int launch_bk(const Plan& pl, const ConvArgs& a, hipStream_t s, int groups = 1) {
    if (pl.bm == 256) return launch_half_256<G>(a, s, pl.half);
    if (pl.bm == 128 && pl.bn == 128) return launch_cfg<128, 128, 2, 2, BK, G>(a, s, groups, pl.split, pl.half);
    if (pl.bm == 128 && pl.bn == 64) return launch_cfg<128, 64, 2, 1, BK, G>(a, s, groups, pl.split, pl.half);
    return launch_cfg<64, 64, 1, 1, BK, G>(a, s, groups, pl.split, pl.half);
}

}  // namespace

// fp16 loop: the descriptor as the fp32 machinery sees it - channels in 4-byte units (two halfs), direct plans only
static bool half_view(const xmem_conv_desc* d, xmem_conv_desc& v) {
    if (!d->w_half || d->Cin % 8 != 0 || d->ldin % 8 != 0 || (((uintptr_t)d->in) & 15) != 0 || (((uintptr_t)d->w_half) & 15) != 0) return false;
    v = *d;
    v.Cin = d->Cin / 2; v.ldin = d->ldin / 2;
    v.w_winograd = nullptr; v.w_winograd4 = nullptr; v.w_winograd_f16 = nullptr; v.arith = 0; v.w_split = nullptr;
    if (v.plan_tile == 4) v.plan_tile = 104;                                          // 256x128 (8 waves), resolved by the caller
    else if (v.plan_tile > 3) v.plan_tile = (v.plan_tile <= 6) ? v.plan_tile - 3 : 0;  // BK = 32 (64 halfs) tiles only
    return true;
}

// plan of a half-typed call: plan_tile 1..3 = {128x128, 128x64, 64x64}, 4 = 256x128 (8 waves), 0 = heuristic
static Plan make_plan_half(xmem_conv_desc& v) {
    const bool big = v.plan_tile == 104;
    if (big) v.plan_tile = 1;
    Plan pl = make_plan(&v);
    if (big && pl.bm != 0) { pl.bm = 256; pl.bn = 128; }
    return pl;
}

extern "C" size_t xmem_conv2d_workspace_bytes(const xmem_conv_desc* d) {
    if (validate(d) != XMEM_OK) return 0;
    xmem_conv_desc hv;
    if (d->in_half) {
        if (!half_view(d, hv)) return 0;
        d = &hv;
    }
    Plan pl = d->in_half ? make_plan_half(hv) : make_plan(d);
    int Ho, Wo; out_dims(d, Ho, Wo);
    if (pl.wino && pl.wino4) return (size_t)36 * d->B * cdiv(Ho, 4) * cdiv(Wo, 4) * (d->Cin + d->Cout) * sizeof(float);
    if (pl.wino && pl.f16) return align_up((size_t)16 * d->B * cdiv(Ho, 2) * cdiv(Wo, 2) * d->Cin * 2, 256) +
                                  (size_t)16 * d->B * cdiv(Ho, 2) * cdiv(Wo, 2) * d->Cout * sizeof(float);
    if (pl.wino && pl.fused) return (size_t)16 * d->B * cdiv(Ho, 2) * cdiv(Wo, 2) * d->Cin * sizeof(float);
    if (pl.wino) return (size_t)16 * d->B * cdiv(Ho, 2) * cdiv(Wo, 2) * (d->Cin + d->Cout) * sizeof(float);
    if (pl.splitk == 1) return 0;
    return (size_t)pl.splitk * d->B * Ho * Wo * d->Cout * sizeof(float);
}

extern "C" int xmem_conv2d_nhwc(const xmem_conv_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = validate(d);
    if (rc != XMEM_OK) return rc;
    xmem_conv_desc hv;
    const int half = d->in_half ? (d->out_half ? 2 : 1) : 0;
    if (half) {
        if (d->out_half && d->Cout == 1) return XMEM_ERR_UNSUPPORTED;      // the mask head writes fp32 logits
        if (!half_view(d, hv)) return XMEM_ERR_UNSUPPORTED;
        d = &hv;
    } else if (d->out_half) return XMEM_ERR_UNSUPPORTED;                    // half output needs half input (the stems stay fp32)
    Plan pl = half ? make_plan_half(hv) : make_plan(d);
    pl.half = half;
    int Ho, Wo; out_dims(d, Ho, Wo);
    ConvArgs a;
    a.in = d->in; a.w = half ? reinterpret_cast<const float*>(d->w_half) : (pl.split ? reinterpret_cast<const float*>(d->w_split) : d->w);
    a.scale = d->scale; a.shift = d->shift; a.res = d->res; a.out = d->out;
    a.a_presplit = 0;
    a.partial = reinterpret_cast<float*>(workspace);
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldin = d->ldin;
    a.Ho = Ho; a.Wo = Wo; a.Cout = d->Cout; a.ldout = d->ldout; a.ldres = d->ldres;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.K = d->KH * d->KW * d->Cin; a.M = d->B * Ho * Wo; a.HoWo = Ho * Wo;
    a.relu_in = d->relu_in; a.relu_out = d->relu_out;
    a.nk = pl.nk; a.splitk = pl.splitk; a.kt_per_split = pl.kt_per_split;
    a.tiles_m = pl.bm ? cdiv(a.M, pl.bm) : 0; a.tiles_n = pl.bn ? cdiv(a.Cout, pl.bn) : 0;
    a.raw = 0; a.res_mod = d->res_broadcast ? Ho * Wo : 0; a.in_gstride = 0; a.w_gstride = 0; a.out_gstride = 0;
    a.dbg = 0;
#ifdef XMEM_TOOLS
    { static const int dbg = getenv("XMEM_CONV_DBG") ? atoi(getenv("XMEM_CONV_DBG")) : 0; a.dbg = dbg; }
#endif
    if (pl.wino && pl.wino4) {
        const int th = cdiv(Ho, 4), tw = cdiv(Wo, 4);
        const size_t P = (size_t)d->B * th * tw;
        const size_t need = (size_t)36 * P * (d->Cin + d->Cout) * sizeof(float);
        if (!workspace || workspace_bytes < need) return XMEM_ERR_WORKSPACE;
        if (P > 0x7fffffff) return XMEM_ERR_UNSUPPORTED;
        float* V = reinterpret_cast<float*>(workspace);
        float* Mt = V + (size_t)36 * P * d->Cin;
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        int blocks, threads;
        transform_grid(P * (d->Cin / 4), blocks, threads);
        hipLaunchKernelGGL(wino4_input_kernel, dim3(blocks), dim3(threads), 0, s, d->in, d->ldin, d->B, d->H, d->W, d->Cin, th, tw,
                           d->relu_in, V, pl.split ? 1 : 0);
        ConvArgs g = a;
        g.in = V; g.w = pl.split ? reinterpret_cast<const float*>(d->w_winograd4_split) : d->w_winograd4;
        g.a_presplit = 1; g.out = Mt; g.res = nullptr; g.partial = nullptr;
        g.B = 1; g.H = 1; g.W = (int)P; g.ldin = d->Cin; g.Ho = 1; g.Wo = (int)P; g.ldout = d->Cout; g.ldres = 0;
        g.KH = 1; g.KW = 1; g.stride = 1; g.pad = 0; g.K = d->Cin; g.M = (int)P; g.HoWo = (int)P;
        g.relu_in = 0; g.relu_out = 0; g.nk = pl.nk; g.splitk = 1; g.kt_per_split = pl.nk;
        g.tiles_m = cdiv(g.M, pl.bm); g.tiles_n = cdiv(g.Cout, pl.bn);
        g.raw = 1; g.in_gstride = (long)P * d->Cin; g.w_gstride = (long)d->Cout * d->Cin; g.out_gstride = (long)P * d->Cout;
        if (pl.stream) rc = launch_stream_positions(pl, V, d->w_winograd4, Mt, 36, (int)P, d->Cin, d->Cout, s);
        else
        rc = (pl.bk == 64) ? (pl.generic ? launch_bk<64, true>(pl, g, s, 36) : launch_bk<64, false>(pl, g, s, 36))
                           : (pl.generic ? launch_bk<32, true>(pl, g, s, 36) : launch_bk<32, false>(pl, g, s, 36));
        if (rc != XMEM_OK) return rc;
        transform_grid(P * (d->Cout / 4), blocks, threads);
        hipLaunchKernelGGL(wino4_output_kernel, dim3(blocks), dim3(threads), 0, s, Mt, d->B, Ho, Wo, d->Cout, th, tw, d->scale, d->shift,
                           d->res, d->ldres, d->res_broadcast ? 1 : 0, d->relu_out, d->out, d->ldout);
        return xmem_check_launch();
    }
    if (pl.wino && pl.f16) {
        const int th = cdiv(Ho, 2), tw = cdiv(Wo, 2);
        const size_t P = (size_t)d->B * th * tw;
        const size_t vbytes = align_up((size_t)16 * P * d->Cin * 2, 256);
        const size_t need = vbytes + (size_t)16 * P * d->Cout * sizeof(float);
        if (!workspace || workspace_bytes < need) return XMEM_ERR_WORKSPACE;
        if (P > 0x7fffffff) return XMEM_ERR_UNSUPPORTED;
        _Float16* V = reinterpret_cast<_Float16*>(workspace);
        float* Mt = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + vbytes);
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        size_t tot = P * (d->Cin / 4);
        int blocks = (int)((tot + 255) / 256); if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(wino_input_f16_kernel, dim3(blocks), dim3(256), 0, s, d->in, d->ldin, d->B, d->H, d->W, d->Cin, th, tw,
                           d->relu_in, V);
        WinoF16Args fa;
        fa.V = V; fa.U = reinterpret_cast<const _Float16*>(d->w_winograd_f16); fa.M = Mt;
        fa.P = (int)P; fa.Cin = d->Cin; fa.Cout = d->Cout;
        // 128x128 tiles once they fill the chip, 64x64 for the small (1/16-resolution, few-channel) layers
        const long big = (long)cdiv((int)P, 128) * cdiv(d->Cout, 128) * 16;
        if (big >= 512 && d->Cout >= 128) {
            fa.tiles_m = cdiv((int)P, 128); fa.tiles_n = cdiv(d->Cout, 128);
            hipLaunchKernelGGL((wino_gemm_f16_kernel<128, 128>), dim3(fa.tiles_m * fa.tiles_n, 16), dim3(256), (size_t)256 * 144, s, fa);
        } else {
            fa.tiles_m = cdiv((int)P, 64); fa.tiles_n = cdiv(d->Cout, 64);
            hipLaunchKernelGGL((wino_gemm_f16_kernel<64, 64>), dim3(fa.tiles_m * fa.tiles_n, 16), dim3(256), (size_t)128 * 144, s, fa);
        }
        if ((rc = xmem_check_launch()) != XMEM_OK) return rc;
        tot = P * (d->Cout / 4);
        blocks = (int)((tot + 255) / 256); if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(wino_output_kernel, dim3(blocks), dim3(256), 0, s, Mt, d->B, Ho, Wo, d->Cout, th, tw, d->scale, d->shift,
                           d->res, d->ldres, d->res_broadcast ? 1 : 0, d->relu_out, d->out, d->ldout);
        return xmem_check_launch();
    }
    if (pl.wino) {
        const int th = cdiv(Ho, 2), tw = cdiv(Wo, 2);
        const size_t P = (size_t)d->B * th * tw;
        const size_t need = (size_t)16 * P * (d->Cin + (pl.fused ? 0 : d->Cout)) * sizeof(float);
        if (!workspace || workspace_bytes < need) return XMEM_ERR_WORKSPACE;
        if (P > 0x7fffffff) return XMEM_ERR_UNSUPPORTED;
        float* V = reinterpret_cast<float*>(workspace);
        float* Mt = V + (size_t)16 * P * d->Cin;
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        int blocks, threads;
        transform_grid(P * (d->Cin / 4), blocks, threads);
        hipLaunchKernelGGL(wino_input_kernel, dim3(blocks), dim3(threads), 0, s, d->in, d->ldin, d->B, d->H, d->W, d->Cin, th, tw,
                           d->relu_in, V, pl.split ? 1 : 0);
        if (pl.fused) {
            WinoArgs wa;
            wa.V = V; wa.U = d->w_winograd; wa.scale = d->scale; wa.shift = d->shift; wa.res = d->res; wa.out = d->out;
            wa.P = (int)P; wa.Cin = d->Cin; wa.Cout = d->Cout; wa.B = d->B; wa.Ho = Ho; wa.Wo = Wo; wa.th = th; wa.tw = tw;
            wa.ldout = d->ldout; wa.ldres = d->ldres; wa.relu_out = d->relu_out; wa.nk = cdiv(d->Cin, 32);
            wa.res_bcast = d->res_broadcast ? 1 : 0;
            const int bm = pl.fused == 1 ? 128 : 64, bn = pl.fused == 3 ? 128 : 64;
            wa.tiles_m = cdiv(wa.P, bm); wa.tiles_n = cdiv(wa.Cout, bn);
            const size_t lds = 2 * (size_t)(bm + bn) * 36 * sizeof(float);
            dim3 grid(wa.tiles_m * wa.tiles_n);
            if (pl.fused == 1) hipLaunchKernelGGL((wino_fused_kernel<128, 64, 2, 1, 32>), grid, dim3(256), lds, s, wa);
            else if (pl.fused == 2) hipLaunchKernelGGL((wino_fused_kernel<64, 64, 1, 1, 32>), grid, dim3(256), lds, s, wa);
            else hipLaunchKernelGGL((wino_fused_kernel<64, 128, 1, 2, 32>), grid, dim3(256), lds, s, wa);
            return xmem_check_launch();
        }
        ConvArgs g = a;
        g.in = V; g.w = pl.split ? reinterpret_cast<const float*>(d->w_winograd_split) : d->w_winograd;
        g.a_presplit = 1; g.out = Mt; g.res = nullptr; g.partial = nullptr;
        g.B = 1; g.H = 1; g.W = (int)P; g.ldin = d->Cin; g.Ho = 1; g.Wo = (int)P; g.ldout = d->Cout; g.ldres = 0;
        g.KH = 1; g.KW = 1; g.stride = 1; g.pad = 0; g.K = d->Cin; g.M = (int)P; g.HoWo = (int)P;
        g.relu_in = 0; g.relu_out = 0; g.nk = pl.nk; g.splitk = 1; g.kt_per_split = pl.nk;
        g.tiles_m = cdiv(g.M, pl.bm); g.tiles_n = cdiv(g.Cout, pl.bn);
        g.raw = 1; g.in_gstride = (long)P * d->Cin; g.w_gstride = (long)d->Cout * d->Cin; g.out_gstride = (long)P * d->Cout;
        if (pl.stream) rc = launch_stream_positions(pl, V, d->w_winograd, Mt, 16, (int)P, d->Cin, d->Cout, s);
        else
        rc = (pl.bk == 64) ? (pl.generic ? launch_bk<64, true>(pl, g, s, 16) : launch_bk<64, false>(pl, g, s, 16))
                           : (pl.generic ? launch_bk<32, true>(pl, g, s, 16) : launch_bk<32, false>(pl, g, s, 16));
        if (rc != XMEM_OK) return rc;
        transform_grid(P * (d->Cout / 4), blocks, threads);
        hipLaunchKernelGGL(wino_output_kernel, dim3(blocks), dim3(threads), 0, s, Mt, d->B, Ho, Wo, d->Cout, th, tw, d->scale, d->shift,
                           d->res, d->ldres, d->res_broadcast ? 1 : 0, d->relu_out, d->out, d->ldout);
        return xmem_check_launch();
    }
    if (pl.splitk > 1) {
        const size_t need = (size_t)pl.splitk * a.M * a.Cout * sizeof(float);
        if (!workspace || workspace_bytes < need) return XMEM_ERR_WORKSPACE;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (pl.stream && pl.bm != 0) {                 // pointwise convolution on the streaming kernel, fused epilogue
        GemmStreamArgs g = {};
        g.A = d->in; g.B = d->w; g.C = d->out; g.scale = d->scale; g.shift = d->shift; g.res = d->res;
        g.M = a.M; g.N = d->Cout; g.K = d->Cin; g.G = 1;
        g.lda = d->ldin; g.ldb = d->Cin; g.ldc = d->ldout; g.ldres = d->ldres;
        g.mode = 1; g.relu_in = d->relu_in; g.relu_out = d->relu_out; g.res_mod = a.res_mod;
        g.stride = d->stride; g.H = d->H; g.W = d->W; g.Wo = Wo; g.HoWo = Ho * Wo;
        return gemm_stream_launch(g, pl.sv, pl.sring, s);
    }
    if (pl.bm == 0) {
        // (round 4: a variant with EIGHT output pixels of a row per wave measured 28.0 us against 24.9 us for the one-pixel-per-wave form at
        // the 480p mask head - its taps were fetched one after the other behind their bounds branches.  Round 6, with every load of a wave
        // requested together: four pixels per wave 26.8 us against 31.2 (host-side events, same bits), +0.6 % on the B32 line; small maps
        // (2 x 37 x 51: 21 against 16 us) keep one pixel per wave - too few waves otherwise)
        if (half) hipLaunchKernelGGL(conv_cout1_kernel<true>, dim3(cdiv(a.M, 4)), dim3(256), 0, s, a);
        else if (cout1_row4() && a.M >= 8192 && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.Cin <= 256 && a.Ho == a.H + 2 * a.pad - 2 && a.Wo == a.W + 2 * a.pad - 2)
            hipLaunchKernelGGL(conv_cout1_row4_kernel, dim3(cdiv(a.B * a.Ho * ((a.Wo + 3) / 4), 4)), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(conv_cout1_kernel<false>, dim3(cdiv(a.M, 4)), dim3(256), 0, s, a);
        return xmem_check_launch();
    }
    if (pl.bk == 64) rc = pl.generic ? launch_bk<64, true>(pl, a, s) : launch_bk<64, false>(pl, a, s);
    else rc = pl.generic ? launch_bk<32, true>(pl, a, s) : launch_bk<32, false>(pl, a, s);
    if (rc != XMEM_OK) return rc;
#ifdef XMEM_TOOLS
    if (a.dbg & 8) {
        static int printed = 0;
        (void)hipStreamSynchronize(s);
        const int n = a.tiles_m * a.tiles_n < CTRACE_MAX ? a.tiles_m * a.tiles_n : CTRACE_MAX;
        static unsigned long long h[CTRACE_MAX][6];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_conv_trace), sizeof(unsigned long long) * 6 * n);
        if (printed++ % 16 == 4) {
            double d[6] = {0, 0, 0, 0, 0, 0}; unsigned long long tmin = ~0ull, tmax = 0;
            for (int i = 0; i < n; ++i) { for (int k = 1; k < 6; ++k) d[k] += (double)(h[i][k] - h[i][0]); if (h[i][0] < tmin) tmin = h[i][0]; if (h[i][5] > tmax) tmax = h[i][5]; }
            fprintf(stderr, "[conv trace] %d workgroups: entry -> first tiles requested %.0f, in LDS %.0f, k-loop done %.0f, stores issued %.0f, stores landed %.0f ticks (mean); "
                            "first entry -> last store landed %.0f ticks; ticks are s_memtime units (100 MHz = 10 ns on gfx950)\n", n, d[1] / n, d[2] / n, d[3] / n, d[4] / n, d[5] / n,
                    (double)(tmax - tmin));
        }
    }
#endif
    if (pl.splitk > 1) {
        const size_t total = (size_t)a.M * a.Cout;
        int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
        if (half == 2) hipLaunchKernelGGL(conv_splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(conv_splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, s, a);
        rc = xmem_check_launch();
    }
    return rc;
}
