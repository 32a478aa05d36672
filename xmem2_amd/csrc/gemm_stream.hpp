// Streaming batched GEMM on the fp32 matrix pipe (csrc/gemm_stream.hip): the Winograd-domain position GEMMs and the pointwise
// convolutions.  Internal to the library (not part of the C ABI).
#pragma once
#include "common.hpp"

struct GemmStreamArgs {
    const float* A;            // [G][M rows][K], row stride lda floats (activations / Winograd-domain V)
    const float* B;            // [G][N rows][K], row stride ldb floats (weights / Winograd-domain U)
    float* C;                  // [G][M][N], row stride ldc floats
    const float* scale;        // fused epilogue (mode 1): per output channel n
    const float* shift;
    const float* res;          // optional residual [M][ldres] (mode 1)
    long a_gstride, b_gstride, c_gstride;   // floats between groups
    int M, N, K, G;
    int lda, ldb, ldc, ldres;
    int nk;                    // K / 32
    int tiles_m, tiles_n, units, units_per_wg;
    int mode;                  // 0: store the bare accumulator; 1: (acc * scale + shift [+ res]) [relu]
    int relu_in, relu_out, res_mod;
    // strided pointwise convolution (mode 1): A row of output pixel m is input pixel ((b*H + oh*stride)*W + ow*stride)
    int stride, H, W, Wo, HoWo;
    int dbg;                   // tools only (XMEM_STREAM_DBG): 1 = every unit loads unit 0's operands, 2 = no stores; results are then wrong
};

// variant: 0 = 64x64 tile, 1 = 128x64, 2 = 128x128 (rows x columns of C per workgroup); ring = LDS stages (3 or 4)
size_t gemm_stream_lds_bytes(int variant, int ring);
int gemm_stream_launch(GemmStreamArgs& a, int variant, int ring, hipStream_t s);
