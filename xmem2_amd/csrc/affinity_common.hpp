// Definitions shared by the memory-readout kernels (affinity.hip: fp32 MFMA select; affinity_filter.hip: fp16-filter + exact refine).
#pragma once
#include "common.hpp"
#include <math.h>

#define AFF_ROWS 32        // memory rows per wave tile
#define AFF_MAX_TOPK 64
#define AFW_GCAP 2048      // per-query global candidate list (all splits of the large-memory select kernels)

typedef unsigned long long u64;

struct SegDev { const float* key; const float* shr; int n; int base; int tile0; int pad; const _Float16* rows16; };   // rows16: fp16 filter only

// Segment lookup with a PER-LANE index: p.seg[i] for a constant i is a scalar kernel-argument load, p.seg[lane_value] would be a
// vector memory load from the argument buffer (one more ~1 us dependent latency) - so select field by field instead.
template <class P> __device__ __forceinline__ SegDev seg_of_row(const P& p, int gi) {       // segment holding global row gi
    SegDev s = p.seg[0];
#pragma unroll
    for (int i = 1; i < XMEM_MAX_SEGMENTS; ++i)
        if (i < p.n_seg && gi >= p.seg[i].base) s = p.seg[i];
    return s;
}
template <class P> __device__ __forceinline__ SegDev seg_of_tile(const P& p, int tile) {    // segment holding 32-row tile `tile`
    SegDev s = p.seg[0];
#pragma unroll
    for (int i = 1; i < XMEM_MAX_SEGMENTS; ++i)
        if (i < p.n_seg && tile >= p.seg[i].tile0) s = p.seg[i];
    return s;
}
template <class P> __device__ __forceinline__ SegDev seg_of_slot(const P& p, int slot) {
    SegDev s = p.seg[0];
#pragma unroll
    for (int i = 1; i < XMEM_MAX_SEGMENTS; ++i)
        if (slot == i) s = p.seg[i];
    return s;
}

// 64-bit candidate key: larger similarity first, then LOWER memory index (keys are unique because the index is part of them)
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ u64 pack_key(float v, int idx) { return ((u64)f2ord(v) << 32) | (u64)(0xffffffffu - (unsigned)idx); }
__device__ __forceinline__ float key_val(u64 k) { return ord2f((unsigned)(k >> 32)); }
__device__ __forceinline__ int key_idx(u64 k) { return (int)(0xffffffffu - (unsigned)k); }

// one term of b_sq = sum_c e_c k_c^2 (memory_util.py:30): ONE rounding of k*k, then a fused multiply-add - spelled out so that
// every kernel that stages a query (select kernels, filter, refine) produces the same bits whatever the contraction setting
__device__ __forceinline__ float bsq_term(float bs, float e, float k) { return fmaf(e, k * k, bs); }

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// ---- fp16 filter operands (affinity_filter.hip): K = 128 contraction terms + 16 augmentation terms -------------------------
// row n   : [ msr x^2 (64) | msr x (64) | msr_hi, msr_hi, msr_lo, kA, kB, |msr|^, zA, 0 x 9 ]      msr = shrinkage / 8
// query q : [   -e    (64) |  2ke  (64) | -bs_hi, -bs_lo, -bs_hi, C^, D^,  mq^, 2^-10, 0 x 9 ]
//   kA = (KAPPA ||x^2|| |msr|)^, kB = (KAPPA ||x|| |msr|)^, C = ||e||, D = ||2ke||, mq = ACC |b_sq| + ABS (C + D),
//   zA = (ABS (||x^2|| + ||x||) |msr| 2^10)^ or +inf when the row leaves the fp16 range;  ^ = rounded up to fp16.
// The fp32-accumulated dot product of the two is a(n,q) + eps(n,q): an UPPER bound of the exact similarity (see the file header).
#define F16_CS 32              // ints between two queries' list counters (one 128-byte line each)
#define F16_K 144              // halfs per operand row (288 B)
#ifndef F16_KAPPA
#define F16_KAPPA 1.07e-3f     // 2^-10 * 1.05 (two fp16 roundings per product) + 4.5e-5 (fp32 accumulation of filter and refine)
#endif
#define F16_ACC 4.5e-5f        // the same accumulation term on |b_sq| (it rides in the accumulator of both chains)
#define F16_ABS 3e-7f          // absolute rounding error of an fp16 SUBNORMAL operand (2^-25), x sqrt(64) via Cauchy-Schwarz
__device__ __forceinline__ _Float16 f16_up(float v) { return (_Float16)(v * 1.001f + 6e-8f); }   // >= v for v >= 0 (RNE is off by <= 2^-11 rel / 3e-8 abs)

// fp16-filter pipeline (affinity_filter.hip), launched by xmem_affinity_topk_hinted when a hint bound is available
struct Filter16Args {
    SegDev seg[XMEM_MAX_SEGMENTS];
    int n_seg, total_tiles;
    const float* qk; const float* qe;
    int HW, top_k;
    int splits, tiles_per_split, qtiles;   // set by aff_filter16_launch (qtiles = query tiles of 128)
    const _Float16* qop16;           // [HW][F16_K] query operand rows          } written by the bound kernel
    const float* qmeta;              // [HW][4]   b_sq (select kernels' arithmetic), 0, 0, 0   }
    _Float16* rows16;                // workspace [N + 32][F16_K] for segments whose caller keeps no operand rows (seg[i].rows16 == 0 on entry)
    float* tau;                      // [HW] valid lower bound of the exact k-th similarity (-inf: none yet); raised by the tighten pass
    int* gcand32; int* gcnt;         // [HW][lstride] candidate indices; [HW][F16_CS] list lengths (one counter per 128-byte line: the filter's
                                     // drains bump them with atomics from every workgroup at once - 1620 counters in 51 lines serialise in a few
                                     // L2 channels), zeroed by the bound kernel
    int* cnt_diag;                   // [HW] final list lengths, written by the refine (diagnostics only)
    int lcap;                        // capacity of a list in THIS pass (pass 1: aff_filter16_list_cap; pass 2: lstride = up to 4x that -
                                     // after a scene cut the similarities are flat and thousands of pairs sit within eps of the k-th)
    int lcap1, lstride;              // pass-1 capacity; allocation stride = pass-2 capacity (aff_filter16_list_stride)
    // Two passes.  Pass 1 (only == nullptr) lists every query; a list that overflows (bound too loose: scene cut, first frames,
    // garbage hint) sets flag1 of its 128-query tile.  The tighten launch turns the partial lists of flagged tiles into a real
    // bound (k-th best EXACT similarity of the listed elements), pass 2 (only == flag1) filters and lists those tiles again.
    // A list that still overflows (exact ties by the thousand) sets flag2: the refine scans that query's tile in full.
    int* flag1; int* flag2;          // [ceil(HW/128)] each, zeroed by the bound kernel
    const int* only;                 // filter / tighten: restrict to tiles whose flag is set (nullptr: all tiles)
    int* flag_out;                   // filter: where an overflow is recorded (flag1 in pass 1, flag2 in pass 2)
    float* out_w; int* out_idx; float* out_sim;
};
size_t aff_filter16_rows_bytes(int n_total);
int aff_filter16_list_cap(int n_total);
int aff_filter16_list_stride(int n_total);
int aff_filter16_launch(Filter16Args a, void* stream);            // rows, filter (lists), tighten, filter, refine
