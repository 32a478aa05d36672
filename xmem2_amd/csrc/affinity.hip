// Fused memory readout for XMem++: anisotropic-L2 similarity (MFMA) + streaming exact top-k + softmax,
// usage accumulation and the sparse value readout.  The N x HW similarity matrix the reference
// materialises (model/memory_util.py:7-65, 336 MB at 480p / 32 memory frames) never leaves the chip.
//
// Contraction (memory_util.py:20-27 folded into one K=2*C_k chain):
//     sim[n][q] = ( sum_c mk[n][c]^2 * (-qe[q][c]) + mk[n][c] * (2*qk[q][c]*qe[q][c])  -  b_sq[q] ) * ms[n] / sqrt(C_k)
// A (memory keys, 32 rows x 64 ch per wave) is loaded straight into registers (each key row is used by exactly
// one wave of the workgroup; squares are formed in-register), B (64 queries x 128 k) is staged once per
// workgroup in LDS with a 132-float row stride (conflict-free ds_read_b128).  D comes out with one query per
// lane (col = lane&31) and 16 memory rows per lane, so the top-k filter is lane-local:
//     v > tau[q]  ->  LDS atomic append to the query's candidate buffer.
// tau[q] is the k-th largest value seen so far; when a buffer passes `limit` entries a wave re-ranks it by
// exact counting (ties -> lower memory index), keeps k and raises tau.  Candidates can never overflow:
// a step adds at most 4 waves x 32 rows = 128 per query and cap = limit + 128.
// N is split across workgroups; a merge kernel ranks the per-split lists and applies the softmax
// exp(v)/sum exp(v) (no max shift, memory_util.py:48-49).
#include "common.hpp"
#include <math.h>

#define AFF_BQ 64          // queries per workgroup
#define AFF_LDB 132        // LDS row stride of the query operand (floats)
#define AFF_ROWS 32        // memory rows per wave tile
#define AFF_STEP_ROWS 128  // rows per step (4 waves)
#define AFF_MAXU 4         // candidate entries per lane during a re-rank (cap <= 256)

struct SegDev { const float* key; const float* shr; int n; int base; int tile0; int pad; };

struct AffArgs {
    SegDev seg[XMEM_MAX_SEGMENTS];
    int n_seg, total_tiles;
    const float* qk; const float* qe;
    int HW, top_k, cap, limit;
    int splits, tiles_per_split;
    float sqrt_ck;
    float* part_v; int* part_i;
};

__device__ __forceinline__ void rerank(float* cv, int* ci, int c, int top_k, float* tau_q, int* cnt_q, int lane) {
    float mv[AFF_MAXU]; int mi[AFF_MAXU]; int rk[AFF_MAXU];
#pragma unroll
    for (int u = 0; u < AFF_MAXU; ++u) {
        const int e = lane + 64 * u;
        mv[u] = e < c ? cv[e] : 0.f;
        mi[u] = e < c ? ci[e] : 0;
        rk[u] = 0;
    }
    for (int f = 0; f < c; ++f) {
        const float vf = cv[f]; const int jf = ci[f];
#pragma unroll
        for (int u = 0; u < AFF_MAXU; ++u)
            rk[u] += (vf > mv[u]) || (vf == mv[u] && jf < mi[u]);
    }
#pragma unroll
    for (int u = 0; u < AFF_MAXU; ++u) {
        const int e = lane + 64 * u;
        if (e < c && rk[u] < top_k) {
            cv[rk[u]] = mv[u]; ci[rk[u]] = mi[u];
            if (rk[u] == top_k - 1) *tau_q = mv[u];
        }
    }
    if (lane == 0) *cnt_q = c < top_k ? c : top_k;
}

template <int CK>
__global__ __launch_bounds__(256) void affinity_topk_kernel(AffArgs p) {
    static_assert(CK == 64, "kernel is specialised for C_k = 64");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Bq = smem;                               // [64][132]
    float* bsq = Bq + AFF_BQ * AFF_LDB;             // [64]
    float* tau = bsq + AFF_BQ;                      // [64]
    int* cnt = reinterpret_cast<int*>(tau + AFF_BQ);  // [64]
    float* cand_v = reinterpret_cast<float*>(cnt + AFF_BQ);   // [64][cap]
    int* cand_i = reinterpret_cast<int*>(cand_v + AFF_BQ * p.cap);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.x * AFF_BQ;
    const int split = blockIdx.y;

    // ---- stage the query operand:  k<64: -qe ; k>=64: 2*qk*qe ; b_sq = sum qe*qk^2 ----
    {
        const int q = tid >> 2, part = tid & 3, qg = q0 + q;
        float bs = 0.f;
        float* row = Bq + q * AFF_LDB;
        if (qg < p.HW) {
            const float* kq = p.qk + (size_t)qg * CK + part * 16;
            const float* eq = p.qe ? p.qe + (size_t)qg * CK + part * 16 : nullptr;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float k = kq[c];
                const float e = eq ? eq[c] : 1.f;
                row[part * 16 + c] = -e;
                row[CK + part * 16 + c] = 2.f * (k * e);
                bs += e * (k * k);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) { row[part * 16 + c] = 0.f; row[CK + part * 16 + c] = 0.f; }
        }
        bs += __shfl_xor(bs, 1, 64);
        bs += __shfl_xor(bs, 2, 64);
        if (part == 0) {
            bsq[q] = p.qe ? bs : 0.f;     // no b_sq term without selection (memory_util.py:28-31)
            tau[q] = -INFINITY;
            cnt[q] = 0;
        }
    }
    __syncthreads();

    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(p.total_tiles, t_begin + p.tiles_per_split);

    // tile -> (segment, row0); returns false for an inactive slot
    auto tile_info = [&](int tile, const float*& key, const float*& shr, int& segn, int& base, int& row0) -> bool {
        if (tile >= t_end) { key = nullptr; shr = nullptr; segn = 0; base = 0; row0 = 0; return false; }
        int s = 0;
#pragma unroll
        for (int i = 1; i < XMEM_MAX_SEGMENTS; ++i)
            if (i < p.n_seg && tile >= p.seg[i].tile0) s = i;
        key = p.seg[s].key; shr = p.seg[s].shr; segn = p.seg[s].n; base = p.seg[s].base;
        row0 = (tile - p.seg[s].tile0) * AFF_ROWS;
        return true;
    };

    f32x4 an[8]; float msn = 1.f;
    const float* n_key; const float* n_shr; int n_segn, n_base, n_row0; bool n_active;
    auto issue_loads = [&](int tile) {
        n_active = tile_info(tile, n_key, n_shr, n_segn, n_base, n_row0);
        const int r = n_row0 + l31;
        const bool ok = n_active && r < n_segn;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const float* src = n_key + (size_t)r * CK + lh * 4;
#pragma unroll
        for (int t = 0; t < 8; ++t) an[t] = ok ? *reinterpret_cast<const f32x4*>(src + t * 8) : zero;
        msn = (ok && n_shr) ? n_shr[r] : 1.f;
    };

    issue_loads(t_begin + wave);
    for (int tb = t_begin; tb < t_end; tb += 4) {
        f32x4 a[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) a[t] = an[t];
        const float ms_mine = msn;
        const bool active = n_active;
        const int segn = n_segn, base = n_base, row0 = n_row0;
        issue_loads(tb + 4 + wave);               // next tile's rows in flight under the MFMAs

        if (active) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const float* bq = Bq + (sub * 32 + l31) * AFF_LDB + lh * 4;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const f32x4 blo = *reinterpret_cast<const f32x4*>(bq + t * 8);
                    const f32x4 bhi = *reinterpret_cast<const f32x4*>(bq + CK + t * 8);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float x = a[t][j];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x * x, blo[j], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, bhi[j], acc, 0, 0, 0);
                    }
                }
                const int q = sub * 32 + l31;
                const float my_tau = tau[q], bs = bsq[q];
                const bool q_ok = (q0 + q) < p.HW;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float msr = __shfl(ms_mine, rl, 64);
                    const float v = ((acc[r] - bs) * msr) / p.sqrt_ck;
                    const int rr = row0 + rl;
                    if (q_ok && rr < segn && v > my_tau) {
                        const int slot = atomicAdd(&cnt[q], 1);
                        if (slot < p.cap) { cand_v[q * p.cap + slot] = v; cand_i[q * p.cap + slot] = base + rr; }
                    }
                }
            }
        }
        __syncthreads();
        for (int q = wave; q < AFF_BQ; q += 4) {
            const int c = cnt[q];
            if (c > p.limit) rerank(cand_v + q * p.cap, cand_i + q * p.cap, c, p.top_k, &tau[q], &cnt[q], lane);
        }
        __syncthreads();
    }

    // ---- final sort of every query's list and hand-off to the merge kernel ----
    for (int q = wave; q < AFF_BQ; q += 4) {
        const int qg = q0 + q;
        if (qg >= p.HW) continue;
        const int c = cnt[q];
        if (c > 0) rerank(cand_v + q * p.cap, cand_i + q * p.cap, c, p.top_k, &tau[q], &cnt[q], lane);
        const int kept = c < p.top_k ? c : p.top_k;
        const size_t o = ((size_t)split * p.HW + qg) * p.top_k;
        for (int s = lane; s < p.top_k; s += 64) {
            p.part_v[o + s] = s < kept ? cand_v[q * p.cap + s] : -INFINITY;
            p.part_i[o + s] = s < kept ? cand_i[q * p.cap + s] : -1;
        }
    }
}

// one wave per query: merge `splits` sorted lists, emit sorted top-k + softmax weights
__global__ void affinity_merge_kernel(const float* __restrict__ part_v, const int* __restrict__ part_i, int splits, int HW, int top_k,
                                      float* __restrict__ out_w, int* __restrict__ out_idx, float* __restrict__ out_sim) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int C = splits * top_k;
    float* lv = sm + (size_t)wv * (2 * C + 2 * top_k);   // candidate values, then filtered values
    int* li = reinterpret_cast<int*>(lv + C);
    const int q = blockIdx.x * wpb + wv;
    if (q >= HW) return;
    float thr = -INFINITY;
    for (int e = lane; e < C; e += 64) {
        const int s = e / top_k, r = e - s * top_k;
        const size_t o = ((size_t)s * HW + q) * top_k + r;
        lv[e] = part_v[o]; li[e] = part_i[o];
    }
    for (int s = lane; s < splits; s += 64) thr = fmaxf(thr, part_v[((size_t)s * HW + q) * top_k + top_k - 1]);
    thr = wave_max(thr);
    // in-place stable filter: keep valid candidates >= thr (the global k-th value is >= every split's k-th)
    int nk = 0;
    for (int b0 = 0; b0 < C; b0 += 64) {
        const int e = b0 + lane;
        const float v = e < C ? lv[e] : 0.f; const int i = e < C ? li[e] : -1;
        const bool keep = e < C && i >= 0 && v >= thr;
        const unsigned long long m = __ballot(keep);
        const int pos = nk + __popcll(m & ((1ull << lane) - 1ull));
        if (keep) { lv[pos] = v; li[pos] = i; }        // pos <= e: never overwrites an unread entry of a later chunk
        nk += __popcll(m);
    }
    // rank by counting; the sorted top-k goes to a private LDS strip first
    float* sv = reinterpret_cast<float*>(li + C);
    int* si = reinterpret_cast<int*>(sv + top_k);
    for (int e = lane; e < nk; e += 64) {
        const float ve = lv[e]; const int ie = li[e];
        int rk = 0;
        for (int f = 0; f < nk; ++f) { const float vf = lv[f]; const int jf = li[f]; rk += (vf > ve) || (vf == ve && jf < ie); }
        if (rk < top_k) { sv[rk] = ve; si[rk] = ie; }
    }
    // softmax without max shift (memory_util.py:48-49); DS ops of one wave execute in order
    float s = 0.f;
    for (int r = lane; r < top_k; r += 64) s += expf(sv[r]);
    s = wave_sum(s);
    for (int r = lane; r < top_k; r += 64) {
        const float v = sv[r];
        out_w[(size_t)q * top_k + r] = expf(v) / s;
        out_idx[(size_t)q * top_k + r] = si[r];
        if (out_sim) out_sim[(size_t)q * top_k + r] = v;
    }
}

namespace {
struct AffPlan { int splits, tiles_per_split, total_tiles, limit, cap; size_t lds; };

AffPlan aff_plan(int total_tiles, int HW, int top_k) {
    AffPlan pl;
    pl.total_tiles = total_tiles;
    const int qtiles = cdiv(HW, AFF_BQ);
    int s = cdiv(512, qtiles);
    int maxs = total_tiles / 16; if (maxs < 1) maxs = 1;
    if (s > maxs) s = maxs;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    pl.tiles_per_split = cdiv(total_tiles, s);
    pl.splits = cdiv(total_tiles, pl.tiles_per_split);
    pl.limit = top_k < 32 ? 32 : (top_k + 7) / 8 * 8;
    pl.cap = pl.limit + AFF_STEP_ROWS;
    pl.lds = ((size_t)AFF_BQ * AFF_LDB + 3 * AFF_BQ + 2 * (size_t)AFF_BQ * pl.cap) * sizeof(float);
    return pl;
}
}  // namespace

extern "C" size_t xmem_affinity_topk_workspace_bytes(int n_total, int HW, int top_k) {
    if (n_total <= 0 || HW <= 0 || top_k <= 0) return 0;
    AffPlan pl = aff_plan(cdiv(n_total, AFF_ROWS) + XMEM_MAX_SEGMENTS, HW, top_k);
    return (size_t)pl.splits * HW * top_k * 8 + 256;
}

extern "C" int xmem_affinity_topk(const xmem_key_segment* segs, int n_seg, const float* qk, const float* qe, int Ck, int HW,
                                  int top_k, float* out_w, int32_t* out_idx, float* out_sim,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    if (!segs || n_seg <= 0 || n_seg > XMEM_MAX_SEGMENTS || !qk || !out_w || !out_idx || HW <= 0) return XMEM_ERR_BAD_ARG;
    if (Ck != 64) return XMEM_ERR_UNSUPPORTED;
    if (top_k < 1 || top_k > 112) return XMEM_ERR_UNSUPPORTED;
    AffArgs a;
    int tiles = 0, base = 0, ns = 0;
    for (int i = 0; i < n_seg; ++i) {
        if (segs[i].n < 0) return XMEM_ERR_BAD_ARG;
        if (segs[i].n == 0) { continue; }
        if (!segs[i].key) return XMEM_ERR_BAD_ARG;
        a.seg[ns].key = segs[i].key; a.seg[ns].shr = segs[i].shrinkage; a.seg[ns].n = segs[i].n;
        a.seg[ns].base = base; a.seg[ns].tile0 = tiles; a.seg[ns].pad = 0;
        tiles += cdiv(segs[i].n, AFF_ROWS); base += segs[i].n; ++ns;
    }
    if (base < top_k) return XMEM_ERR_TOPK;
    for (int i = ns; i < XMEM_MAX_SEGMENTS; ++i) { a.seg[i].key = nullptr; a.seg[i].shr = nullptr; a.seg[i].n = 0; a.seg[i].base = base; a.seg[i].tile0 = tiles; a.seg[i].pad = 0; }
    AffPlan pl = aff_plan(tiles, HW, top_k);
    const size_t need = (size_t)pl.splits * HW * top_k * 8;
    if (!workspace || workspace_bytes < need) return XMEM_ERR_WORKSPACE;
    a.n_seg = ns; a.total_tiles = tiles; a.qk = qk; a.qe = qe; a.HW = HW; a.top_k = top_k; a.cap = pl.cap; a.limit = pl.limit;
    a.splits = pl.splits; a.tiles_per_split = pl.tiles_per_split; a.sqrt_ck = sqrtf((float)Ck);
    a.part_v = reinterpret_cast<float*>(workspace);
    a.part_i = reinterpret_cast<int*>(a.part_v + (size_t)pl.splits * HW * top_k);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    auto kern = affinity_topk_kernel<64>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds) != hipSuccess)
        return XMEM_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(cdiv(HW, AFF_BQ), pl.splits), dim3(256), pl.lds, s, a);
    int rc = xmem_check_launch();
    if (rc != XMEM_OK) return rc;
    const int C = pl.splits * top_k;
    int wpb = 4;
    while (wpb > 1 && (size_t)wpb * (2 * C + 2 * top_k) * 4 > 60 * 1024) wpb >>= 1;
    const size_t mlds = (size_t)wpb * (2 * C + 2 * top_k) * 4;
    if (mlds > 64 * 1024) return XMEM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(affinity_merge_kernel, dim3(cdiv(HW, wpb)), dim3(64 * wpb), mlds, s, a.part_v, a.part_i, pl.splits, HW, top_k,
                       out_w, out_idx, out_sim);
    return xmem_check_launch();
}

// ---------------------------------------------------------------------------------------------
// usage = affinity.sum(dim=2), order-independent (64-bit fixed point, 2^-40 resolution)
// ---------------------------------------------------------------------------------------------
#define USAGE_FX_SCALE 1099511627776.0   /* 2^40 */

__global__ void usage_scatter_kernel(const float* __restrict__ w, const int* __restrict__ idx, size_t total, int first, int count,
                                     unsigned long long* __restrict__ fx) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int i = idx[e] - first;
        if ((unsigned)i < (unsigned)count) {
            const unsigned long long q = (unsigned long long)((double)w[e] * USAGE_FX_SCALE);
            atomicAdd(&fx[i], q);
        }
    }
}

__global__ void usage_apply_kernel(const unsigned long long* __restrict__ fx, float* __restrict__ use, float* __restrict__ life, int count) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        use[i] += (float)((double)fx[i] * (1.0 / USAGE_FX_SCALE));
        life[i] += 1.f;
    }
}

extern "C" int xmem_usage_update(const float* w, const int32_t* idx, int HW, int top_k, int first, int count,
                                 float* use_count, float* life_count, uint64_t* fx_scratch, void* stream) {
    if (!w || !idx || HW <= 0 || top_k <= 0 || first < 0 || count < 0) return XMEM_ERR_BAD_ARG;
    if (count == 0) return XMEM_OK;
    if (!use_count || !life_count || !fx_scratch) return XMEM_ERR_BAD_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(fx_scratch, 0, (size_t)count * 8, s) != hipSuccess) return XMEM_ERR_LAUNCH;
    const size_t total = (size_t)HW * top_k;
    int g = (int)((total + 255) / 256); if (g > 2048) g = 2048;
    hipLaunchKernelGGL(usage_scatter_kernel, dim3(g), dim3(256), 0, s, w, idx, total, first, count,
                       reinterpret_cast<unsigned long long*>(fx_scratch));
    int g2 = (count + 255) / 256; if (g2 > 2048) g2 = 2048;
    hipLaunchKernelGGL(usage_apply_kernel, dim3(g2), dim3(256), 0, s, reinterpret_cast<const unsigned long long*>(fx_scratch),
                       use_count, life_count, count);
    return xmem_check_launch();
}

// ---------------------------------------------------------------------------------------------
// sparse readout: out[obj][q][:] = sum_s w[q][s] * V_obj[idx[q][s]][:]
// ---------------------------------------------------------------------------------------------
#define RO_MAX_ENT 64
struct ReadoutArgs {
    const float* val[RO_MAX_ENT]; int n[XMEM_MAX_SEGMENTS];
    int n_obj, n_seg;
    const float* w; const int* idx; int HW, top_k, Cv;
    float* out; int ldout; size_t obj_stride;
};

__global__ void readout_sparse_kernel(ReadoutArgs p) {
    const int q = blockIdx.x, obj = blockIdx.y;
    const float* wq = p.w + (size_t)q * p.top_k;
    const int* iq = p.idx + (size_t)q * p.top_k;
    for (int c4 = threadIdx.x; c4 * 4 < p.Cv; c4 += blockDim.x) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < p.top_k; ++s) {
            int i = iq[s];
            const float ws = wq[s];
            int sg = 0;
            while (sg < p.n_seg - 1 && i >= p.n[sg]) { i -= p.n[sg]; ++sg; }
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.val[obj * p.n_seg + sg] + (size_t)i * p.Cv + c4 * 4);
            acc.x += ws * v.x; acc.y += ws * v.y; acc.z += ws * v.z; acc.w += ws * v.w;
        }
        *reinterpret_cast<f32x4*>(p.out + (size_t)obj * p.obj_stride + (size_t)q * p.ldout + c4 * 4) = acc;
    }
}

extern "C" int xmem_readout_sparse(const xmem_value_segment* vsegs, int n_obj, int n_seg, const float* w, const int32_t* idx,
                                   int HW, int top_k, int Cv, float* out, int ldout, size_t obj_stride, void* stream) {
    if (!vsegs || n_obj <= 0 || n_seg <= 0 || n_seg > XMEM_MAX_SEGMENTS || !w || !idx || !out || HW <= 0 || top_k <= 0) return XMEM_ERR_BAD_ARG;
    if (Cv % 4 || ldout % 4 || ldout < Cv || obj_stride % 4) return XMEM_ERR_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int per = RO_MAX_ENT / n_seg;
    for (int o0 = 0; o0 < n_obj; o0 += per) {
        const int no = (n_obj - o0) < per ? (n_obj - o0) : per;
        ReadoutArgs a;
        for (int sg = 0; sg < XMEM_MAX_SEGMENTS; ++sg) a.n[sg] = sg < n_seg ? vsegs[(size_t)o0 * n_seg + sg].n : 0;
        for (int o = 0; o < no; ++o)
            for (int sg = 0; sg < n_seg; ++sg) {
                const xmem_value_segment& v = vsegs[(size_t)(o0 + o) * n_seg + sg];
                if (v.n != a.n[sg] || (v.n > 0 && !v.value)) return XMEM_ERR_BAD_ARG;
                a.val[o * n_seg + sg] = v.value;
            }
        a.n_obj = no; a.n_seg = n_seg; a.w = w; a.idx = idx; a.HW = HW; a.top_k = top_k; a.Cv = Cv;
        a.out = out + (size_t)o0 * obj_stride; a.ldout = ldout; a.obj_stride = obj_stride;
        int threads = Cv / 4; if (threads > 256) threads = 256; threads = (threads + 63) / 64 * 64;
        hipLaunchKernelGGL(readout_sparse_kernel, dim3(HW, no), dim3(threads), 0, s, a);
        int rc = xmem_check_launch();
        if (rc != XMEM_OK) return rc;
    }
    return XMEM_OK;
}

// ---------------------------------------------------------------------------------------------
// dense similarity for the consolidation: out[p][n]
// ---------------------------------------------------------------------------------------------
__global__ void similarity_dense_kernel(const float* __restrict__ key, const float* __restrict__ shr, int n,
                                        const float* __restrict__ qk, const float* __restrict__ qe, int Ck, float sqrt_ck,
                                        float* __restrict__ out) {
    extern __shared__ float sb[];          // [2*Ck] operand + b_sq
    const int p = blockIdx.y;
    float* blo = sb; float* bhi = sb + Ck;
    __shared__ float bsq_s;
    if (threadIdx.x < 64) {
        float bs = 0.f;
        for (int c = threadIdx.x; c < Ck; c += 64) {
            const float k = qk[(size_t)p * Ck + c];
            const float e = qe ? qe[(size_t)p * Ck + c] : 1.f;
            blo[c] = -e; bhi[c] = 2.f * (k * e); bs += e * (k * k);
        }
        bs = wave_sum(bs);
        if (threadIdx.x == 0) bsq_s = qe ? bs : 0.f;
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* row = key + (size_t)i * Ck;
    float acc = 0.f;
    for (int c = 0; c < Ck; c += 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(row + c);
        acc = fmaf(x.x * x.x, blo[c], acc);     acc = fmaf(x.x, bhi[c], acc);
        acc = fmaf(x.y * x.y, blo[c + 1], acc); acc = fmaf(x.y, bhi[c + 1], acc);
        acc = fmaf(x.z * x.z, blo[c + 2], acc); acc = fmaf(x.z, bhi[c + 2], acc);
        acc = fmaf(x.w * x.w, blo[c + 3], acc); acc = fmaf(x.w, bhi[c + 3], acc);
    }
    const float ms = shr ? shr[i] : 1.f;
    out[(size_t)p * n + i] = ((acc - bsq_s) * ms) / sqrt_ck;
}

extern "C" int xmem_similarity_dense(const float* key, const float* shrinkage, int n, const float* qk, const float* qe, int P, int Ck,
                                     float* out, void* stream) {
    if (!key || !qk || !out || n <= 0 || P <= 0 || Ck <= 0) return XMEM_ERR_BAD_ARG;
    if (Ck % 4) return XMEM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(similarity_dense_kernel, dim3(cdiv(n, 256), P), dim3(256), 2 * Ck * sizeof(float), (hipStream_t)stream,
                       key, shrinkage, n, qk, qe, Ck, sqrtf((float)Ck), out);
    return xmem_check_launch();
}
