// Fused memory readout for XMem++: anisotropic-L2 similarity (MFMA) + streaming exact top-k + softmax,
// usage accumulation and the sparse value readout.  The N x HW similarity matrix the reference
// materialises (model/memory_util.py:7-65, 336 MB at 480p / 32 memory frames) never leaves the chip.
//
// Contraction (memory_util.py:20-27 folded into one K=2*C_k chain):
//     sim[n][q] = ( sum_c mk[n][c]^2 * (-qe[q][c]) + mk[n][c] * (2*qk[q][c]*qe[q][c])  -  b_sq[q] ) * ms[n] / sqrt(C_k)
// A (memory keys, 32 rows x 64 ch per wave) is loaded straight into registers (each key row is used by exactly
// one wave of the workgroup; squares are formed in-register), B (64 queries x 128 k) is staged once per
// workgroup in LDS with a 132-float row stride (conflict-free ds_read_b128).  D comes out with one query per
// lane (col = lane&31) and 16 memory rows per lane, so the top-k filter is lane-local.
//
// Exact top-k without materialising anything, in two MFMA passes:
//   pass A (bound): every R-th 32-row tile; each lane keeps the two largest values it sees per query in registers
//           (branch-free max/min chain).  The k-th largest of the union of those survivors is a value that at
//           least k memory elements reach, i.e. a LOWER BOUND tau0[q] of the true k-th similarity.
//   pass B (select): all tiles; v >= tau0[q] -> LDS atomic append of a packed 64-bit key
//           (orderable(v) << 32 | ~index) to the query's candidate buffer.  About R*k/splits candidates survive per
//           workgroup, so the exact re-rank (by counting, ties -> lower index) is a rare safety valve:
//           it runs only if a buffer passes `limit` entries and can never overflow (a step adds at most
//           4 waves x 32 rows = 128 per query, cap = limit + 128).
//   merge: one wave per query ranks the surviving candidates of all splits, emits the sorted top-k and the
//           softmax exp(v)/sum exp(v) (no max shift, memory_util.py:48-49).
// Small memories (< 256 tiles) skip pass A (tau0 = -inf) and rely on the re-rank.
#include "affinity_common.hpp"
#include <stdlib.h>

#define AFF_BQ 64          // queries per workgroup
#define AFF_LDB 132        // LDS row stride of the query operand (floats)
#define AFF_STEP_ROWS 128  // rows per step (4 waves)
#define AFF_MAXU 4         // candidate entries per lane during a re-rank (cap <= 256)
#define AFF_OUTCAP 88      // candidates a (split, query) hands to the merge kernel
#define AFF_BOUND_M 1      // survivors per lane and query in the bound pass
#define AFF_BOUND_SLOTS (8 * AFF_BOUND_M)   // per (split, query): 4 waves x 2 half-waves x M
#define AFF_MAX_BOUND_SPLITS 64

struct AffArgs {
    SegDev seg[XMEM_MAX_SEGMENTS];
    int n_seg, total_tiles;
    const float* qk; const float* qe;
    int HW, top_k, cap, limit;
    int splits, tiles_per_split;
    int chunk;                       // > 0: tiles are dealt to the splits in chunks of this many (round-robin); 0: contiguous ranges
    int merge_splits;                // MODE 3: number of per-split lists the merge kernel will read (>= splits)
    int tile_stride;                 // visit tiles 0, R, 2R, ...
    int sub_tiles;                   // ceil(total_tiles / tile_stride)
    const float* tau_init;           // [HW] lower bound of the k-th value (select pass) or NULL
    float sqrt_ck; int sqrt_is_pow2;
    u64* part_key; int* part_cnt;    // select pass output: [splits][HW][AFF_OUTCAP], [splits][HW]
    float* bound_part;               // bound pass output: [splits][HW][AFF_BOUND_SLOTS]
    int* ovf;                        // [query tiles] overflow flags of the optimistic select pass
    u64* cand_spill;                 // MODE 3: [splits][query tiles][64][cap] global candidate buffers
};

// exact re-rank of one query's buffer by counting: larger key first (value, then lower index); keeps top_k sorted
__device__ __forceinline__ void rerank(u64* ck, int c, int top_k, float* tau_q, int* cnt_q, int lane) {
    u64 mk[AFF_MAXU]; int rk[AFF_MAXU];
    const int U = (c + 63) >> 6;
#pragma unroll
    for (int u = 0; u < AFF_MAXU; ++u) {
        const int e = lane + 64 * u;
        mk[u] = e < c ? ck[e] : 0ull;
        rk[u] = 0;
    }
    for (int f = 0; f < c; f += 2) {                          // two broadcast keys per 16-B LDS read
        const ulonglong2 kf = *reinterpret_cast<const ulonglong2*>(ck + f);
        const u64 k1 = (f + 1 < c) ? kf.y : 0ull;
#pragma unroll
        for (int u = 0; u < AFF_MAXU; ++u)
            if (u < U) rk[u] += (int)(kf.x > mk[u]) + (int)(k1 > mk[u]);
    }
#pragma unroll
    for (int u = 0; u < AFF_MAXU; ++u) {
        const int e = lane + 64 * u;
        if (e < c && rk[u] < top_k) {
            ck[rk[u]] = mk[u];
            if (rk[u] == top_k - 1) *tau_q = key_val(mk[u]);
        }
    }
    if (lane == 0) *cnt_q = c < top_k ? c : top_k;
}

// MODE 0: bound pass.  MODE 1: optimistic select (small candidate buffers, no re-rank, no barriers in the tile loop,
// two workgroups per CU; a buffer overflow raises ovf[query tile]).  MODE 2: safe select (worst-case buffers +
// re-rank valve); when p.ovf is set it only re-does query tiles whose optimistic pass overflowed.
// MODE 3: the safe select as the fallback of the optimistic pass: its worst-case candidate buffers live in GLOBAL
// scratch (p.cand_spill), so the launch asks for 34 KB of LDS instead of 148 KB.  Nearly all of its workgroups return
// at the flag test, and with the big allocation they could not even START while another stream's kernels held LDS
// (measured: 90 us per frame of pure scheduling stall under the pipelined key encoder).
template <int CK, int MODE>
__device__ __forceinline__ void affinity_body(const AffArgs& p, const int bx, const int by, const int spill_slot) {
    static_assert(CK == 64, "kernel is specialised for C_k = 64");
    constexpr bool BOUND = (MODE == 0);
    constexpr bool SAFE = (MODE >= 2);
    constexpr bool SPILL = (MODE == 3);
    if (SAFE && p.ovf && p.ovf[bx] == 0) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Bq = smem;                               // [64][132]
    float* bsq = Bq + AFF_BQ * AFF_LDB;             // [64]
    float* tau = bsq + AFF_BQ;                      // [64]
    int* cnt = reinterpret_cast<int*>(tau + AFF_BQ);  // [64]
    u64* cand = reinterpret_cast<u64*>(cnt + AFF_BQ + 4);  // [64][cap]   (select pass only); cnt[64..67]: flag + pad
    if (SPILL) cand = p.cand_spill + (size_t)spill_slot * ((size_t)AFF_BQ * p.cap);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int q0 = bx * AFF_BQ;
    const int split = by;

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
                bs = bsq_term(bs, e, k);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) { row[part * 16 + c] = 0.f; row[CK + part * 16 + c] = 0.f; }
        }
        bs += __shfl_xor(bs, 1, 64);
        bs += __shfl_xor(bs, 2, 64);
        if (part == 0) {
            bsq[q] = p.qe ? bs : 0.f;     // no b_sq term without selection (memory_util.py:28-31)
            // tau0 is reached by >= k elements; elements equal to it must still pass the strict test below
            tau[q] = (!BOUND && p.tau_init && qg < p.HW) ? nextafterf(p.tau_init[qg], -INFINITY) : -INFINITY;
            cnt[q] = 0;
        }
        if (tid == 0) cnt[AFF_BQ] = 0;              // overflow-valve flag
    }
    __syncthreads();

    // Split -> tiles.  Default: split s owns the contiguous range [s*tps, (s+1)*tps).  Large memories (p.chunk > 0) deal
    // chunks of ~half a memory frame round-robin to the splits instead: neighbouring frames are similar, so a query's best
    // matches cluster in time, and with contiguous ranges they piled up in the candidate buffers of one split.
    const int t_begin = p.chunk ? 0 : split * p.tiles_per_split;         // in units of visited (sub-sampled) tiles
    const int t_end = p.chunk ? p.tiles_per_split : min(p.sub_tiles, t_begin + p.tiles_per_split);

    auto tile_info = [&](int tile, const float*& key, const float*& shr, int& segn, int& base, int& row0) -> bool {
        bool off = tile >= t_end;
        if (p.chunk && !off) {
            const int ci = tile / p.chunk, wi = tile - ci * p.chunk;
            tile = (ci * p.splits + split) * p.chunk + wi;
            off = tile >= p.sub_tiles;
        }
        if (off) { key = nullptr; shr = nullptr; segn = 0; base = 0; row0 = 0; return false; }
        tile *= p.tile_stride;
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

    // bound pass state: the AFF_BOUND_M largest values this lane has seen, per query sub-tile
    float top[2][AFF_BOUND_M];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int j = 0; j < AFF_BOUND_M; ++j) top[sub][j] = -INFINITY;

    int* flag = cnt + AFF_BQ;                        // overflow-valve flag lives right after cnt[]
    // sqrt(C_k) = 8 is a power of two, so ((x * ms) / sqrt(C_k)) == x * (ms * 0.125) bit for bit (memory_util.py:34-37)
    static_assert(CK == 64, "scale folding assumes sqrt(C_k) is a power of two");
    constexpr float inv_sqrt = 0.125f;
    const float bs0 = bsq[l31], bs1 = bsq[32 + l31];
    const bool q_ok0 = (q0 + l31) < p.HW, q_ok1 = (q0 + 32 + l31) < p.HW;

    // ---- epilogue of one finished tile ---------------------------------------------------------------------
    // acc already holds (sum - b_sq) (the accumulators start at -b_sq); msr = shrinkage / sqrt(C_k) per owned row.
    // `best*` is the lane's running maximum of the tile (computed next to the following tile's MFMAs);
    // only lanes whose maximum passes the bound walk their 16 values.
    auto append_sub = [&](const f32x16& acc, const float* msr, int sub, float best, bool q_ok, int row0, int segn, int base) {
        const int q = sub * 32 + l31;
        const float my_tau = tau[q];
        if (q_ok && best > my_tau) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float x = acc[r] * msr[r];
                if (rr < segn && x > my_tau) {
                    const int slot = atomicAdd(&cnt[q], 1);
                    if (slot < p.cap) cand[(size_t)q * p.cap + slot] = pack_key(x, base + rr);
                    if (SAFE) { if (slot >= p.limit) *flag = 1; }
                    else if (slot >= p.cap) p.ovf[bx] = 1;   // optimistic buffers exhausted: redo this query tile safely
                }
            }
        }
    };
    auto bound_insert = [&](float v0, float v1) {      // branch-free insertion into the sorted survivors
#pragma unroll
        for (int j = 0; j < AFF_BOUND_M; ++j) {
            const float h0 = fmaxf(top[0][j], v0); v0 = fminf(top[0][j], v0); top[0][j] = h0;
            const float h1 = fmaxf(top[1][j], v1); v1 = fminf(top[1][j], v1); top[1][j] = h1;
        }
    };

    // ---- one tile: MFMAs into (c0, c1) while the PREVIOUS tile (p0, p1) is filtered on the VALU ------------
    struct TileMeta { int row0, segn, base; bool have, full; };
    float msrA[16], msrB[16];
    auto run_tile = [&](f32x16& c0, f32x16& c1, float* cmsr, TileMeta& cm,
                        const f32x16& p0, const f32x16& p1, const float* pmsr, const TileMeta& pm, int tb) {
        f32x4 a[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) a[t] = an[t];
        const float ms_mine = msn;
        cm.have = n_active; cm.row0 = n_row0; cm.segn = n_segn; cm.base = n_base;
        cm.full = n_active && (n_row0 + AFF_ROWS <= n_segn);
        issue_loads(tb + 4 + wave);               // next tile's rows in flight under the MFMAs
        const bool pfast = pm.have && pm.full;    // full previous tile: its filter runs inside the MFMA loop
        float best0 = -INFINITY, best1 = -INFINITY;
        if (cm.have) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { c0[r] = -bs0; c1[r] = -bs1; }
            const float* bq0 = Bq + l31 * AFF_LDB + lh * 4;
            const float* bq1 = bq0 + 32 * AFF_LDB;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f32x4 blo0 = *reinterpret_cast<const f32x4*>(bq0 + t * 8);
                const f32x4 bhi0 = *reinterpret_cast<const f32x4*>(bq0 + CK + t * 8);
                const f32x4 blo1 = *reinterpret_cast<const f32x4*>(bq1 + t * 8);
                const f32x4 bhi1 = *reinterpret_cast<const f32x4*>(bq1 + CK + t * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = a[t][j];
                    const float xx = x * x;
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xx, blo0[j], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xx, blo1[j], c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, bhi0[j], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, bhi1[j], c1, 0, 0, 0);
                }
                if (pfast) {                      // two rows of the previous tile per k-group: pure VALU, no branches
#pragma unroll
                    for (int r = 2 * t; r < 2 * t + 2; ++r) {
                        float x0 = p0[r] * pmsr[r], x1 = p1[r] * pmsr[r];
                        if (BOUND) bound_insert(x0, x1);
                        else { best0 = fmaxf(best0, x0); best1 = fmaxf(best1, x1); }
                    }
                }
            }
            // shrinkage of the 16 rows this lane owns; ((x*ms)/sqrt(Ck)) == x*(ms/sqrt(Ck)) exactly when sqrt(Ck) = 2^j
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m = __shfl(ms_mine, (r & 3) + 8 * (r >> 2) + 4 * lh, 64);
                cmsr[r] = m * inv_sqrt;
            }
        }
        if (pm.have && !(pfast && cm.have)) {     // ragged previous tile, or nothing to hide it under: plain loop
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = pm.row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float x0 = p0[r] * pmsr[r], x1 = p1[r] * pmsr[r];
                x0 = rr < pm.segn ? x0 : -INFINITY; x1 = rr < pm.segn ? x1 : -INFINITY;
                if (BOUND) bound_insert(x0, x1);
                else { best0 = fmaxf(best0, x0); best1 = fmaxf(best1, x1); }
            }
        }
        if (!BOUND && pm.have) {
            append_sub(p0, pmsr, 0, best0, q_ok0, pm.row0, pm.segn, pm.base);
            append_sub(p1, pmsr, 1, best1, q_ok1, pm.row0, pm.segn, pm.base);
        }
        if (SAFE) {
            __syncthreads();
            if (*flag) {                          // overflow valve: a buffer passed `limit` entries (never with a valid bound)
                for (int q = wave; q < AFF_BQ; q += 4) {
                    const int c = cnt[q];
                    if (c > p.limit) rerank(cand + (size_t)q * p.cap, c, p.top_k, &tau[q], &cnt[q], lane);
                }
                __syncthreads();
                if (tid == 0) *flag = 0;
                __syncthreads();
            }
        }
    };

    f32x16 accA0, accA1, accB0, accB1;
    TileMeta mA = {0, 0, 0, false, false}, mB = {0, 0, 0, false, false};
    issue_loads(t_begin + wave);
    // steps are processed in pairs so that the accumulator sets swap roles without register copies; one extra
    // (empty) step flushes the filter of the last tile.
    for (int tb = t_begin; tb < t_end + 4; tb += 8) {
        run_tile(accA0, accA1, msrA, mA, accB0, accB1, msrB, mB, tb);
        run_tile(accB0, accB1, msrB, mB, accA0, accA1, msrA, mA, tb + 4);
    }
    if (!BOUND) __syncthreads();

    if (BOUND) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int qg = q0 + sub * 32 + l31;
            if (qg < p.HW) {
                float* o = p.bound_part + ((size_t)split * p.HW + qg) * AFF_BOUND_SLOTS + (wave * 2 + lh) * AFF_BOUND_M;
#pragma unroll
                for (int j = 0; j < AFF_BOUND_M; ++j) o[j] = top[sub][j];
            }
        }
        return;
    }
    // ---- hand the surviving candidates (unsorted unless re-ranked) to the merge kernel ----
    for (int q = wave; q < AFF_BQ; q += 4) {
        const int qg = q0 + q;
        if (qg >= p.HW) continue;
        int c = cnt[q];
        if (SAFE) { if (c > AFF_OUTCAP) { rerank(cand + (size_t)q * p.cap, c, p.top_k, &tau[q], &cnt[q], lane); c = min(c, p.top_k); } }
        else c = min(c, p.cap);                   // cap <= AFF_OUTCAP in the optimistic pass
        const size_t o = (size_t)split * p.HW + qg;
        if (lane == 0) p.part_cnt[o] = c;
        for (int s = lane; s < c; s += 64) p.part_key[o * AFF_OUTCAP + s] = cand[(size_t)q * p.cap + s];
        if (SPILL && lane == 0)                   // the fallback grid has fewer splits than the optimistic pass it replaces:
            for (int s2 = split + p.splits; s2 < p.merge_splits; s2 += p.splits) p.part_cnt[(size_t)s2 * p.HW + qg] = 0;
    }
}


// MODE 0 / 1 / 2: one (query tile, split) per workgroup.  MODE 3 (the safe pass over tiles whose candidate lists overflowed,
// normally none): a SMALL persistent grid - every workgroup lists the flagged tiles and takes (tile, split) pairs round-robin.
// An unneeded launch therefore costs a handful of workgroups that read the flags and leave (a full-size grid of empty
// workgroups queued for 20-60 us behind the other stream's kernels in the pipelined frame loop).
#define AFF_FB_MAXTILES 2048
template <int CK, int MODE>
__global__ __launch_bounds__(256, (MODE == 2 ? 1 : 2)) void affinity_kernel(AffArgs p) {
    if (MODE != 3) { affinity_body<CK, MODE>(p, blockIdx.x, blockIdx.y, 0); return; }
    __shared__ int s_list[AFF_FB_MAXTILES];
    __shared__ int s_n;
    const int qtiles = (p.HW + AFF_BQ - 1) / AFF_BQ;
    if (threadIdx.x == 0) {
        int n = 0;
        for (int t = 0; t < qtiles && n < AFF_FB_MAXTILES; ++t) if (p.ovf[t]) s_list[n++] = t;
        s_n = n;
    }
    __syncthreads();
    const int pairs = s_n * p.splits;
    for (int it = blockIdx.x; it < pairs; it += gridDim.x) {
        affinity_body<CK, MODE>(p, s_list[it / p.splits], it % p.splits, blockIdx.x);
        __syncthreads();
    }
}

// bound pass reduction: tau0[q] = k-th largest of the survivors of all splits (one wave per query)
__global__ void affinity_bound_kernel(const float* __restrict__ bound_part, int splits, int HW, int top_k, float* __restrict__ tau0,
                                      int* __restrict__ gcnt, int* __restrict__ ovf) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int T = splits * AFF_BOUND_SLOTS;
    float* v = sm + (size_t)wv * T;
    const int q = blockIdx.x * wpb + wv;
    if (q >= HW) return;
    for (int e = lane; e < T; e += 64) {
        const int s = e / AFF_BOUND_SLOTS, r = e - s * AFF_BOUND_SLOTS;
        v[e] = bound_part[((size_t)s * HW + q) * AFF_BOUND_SLOTS + r];
    }
    float res = -INFINITY;      // stays -inf when fewer than k survivors exist
    for (int e = lane; e < T; e += 64) {
        const float ve = v[e];
        int rk = 0;
        for (int f = 0; f < T; f += 4) {                      // T is a multiple of 16
            const f32x4 vf = *reinterpret_cast<const f32x4*>(v + f);
            rk += (int)((vf.x > ve) || (vf.x == ve && f < e)) + (int)((vf.y > ve) || (vf.y == ve && f + 1 < e)) +
                  (int)((vf.z > ve) || (vf.z == ve && f + 2 < e)) + (int)((vf.w > ve) || (vf.w == ve && f + 3 < e));
        }
        if (rk == top_k - 1) res = ve;
    }
    res = wave_max(res);
    if (lane == 0) { tau0[q] = res; gcnt[q] = 0; if ((q & 63) == 0) ovf[q >> 6] = 0; }
}

// one wave per query: rank the candidates of all splits, emit sorted top-k + softmax weights
__global__ void affinity_merge_kernel(const u64* __restrict__ part_key, const int* __restrict__ part_cnt, int splits, int HW, int top_k,
                                      float* __restrict__ out_w, int* __restrict__ out_idx, float* __restrict__ out_sim) {
    extern __shared__ __attribute__((aligned(16))) u64 smk[];
    const int lane = threadIdx.x;
    const int q = blockIdx.x;
    u64* keys = smk;                                         // [splits * AFF_OUTCAP] (+1 pad)
    float* sv = reinterpret_cast<float*>(keys + (size_t)splits * AFF_OUTCAP + 2);
    int* si = reinterpret_cast<int*>(sv + top_k);
    // counts -> exclusive prefix (splits <= 64: one lane per split)
    const int my_c = lane < splits ? part_cnt[(size_t)lane * HW + q] : 0;
    int incl = my_c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    int total = __shfl(incl, 63, 64);
    const int excl = incl - my_c;
    for (int s = 0; s < splits; ++s) {
        const int c = __shfl(my_c, s, 64), off = __shfl(excl, s, 64);
        for (int e = lane; e < c; e += 64) keys[off + e] = part_key[((size_t)s * HW + q) * AFF_OUTCAP + e];
    }
    // pre-filter: the top_k-th largest of the 64 lane-local maxima is reached by >= top_k candidates, so everything
    // below it can be dropped before the O(T^2/64) exact ranking (tames queries with several hundred survivors)
    if (total > 96) {
        u64 lmax = 0ull;
        for (int e = lane; e < total; e += 64) { const u64 k = keys[e]; lmax = k > lmax ? k : lmax; }
        int rk = 0;
        for (int j = 0; j < 64; ++j) {
            const unsigned lo = __shfl((unsigned)lmax, j, 64), hi = __shfl((unsigned)(lmax >> 32), j, 64);
            const u64 o = ((u64)hi << 32) | lo;
            rk += (o > lmax) || (o == lmax && j < lane);
        }
        const unsigned long long sel = __ballot(rk == top_k - 1);
        const int src = sel ? __ffsll((long long)sel) - 1 : 0;
        const unsigned tlo = __shfl((unsigned)lmax, src, 64), thi = __shfl((unsigned)(lmax >> 32), src, 64);
        const u64 thr = sel ? (((u64)thi << 32) | tlo) : 0ull;
        int nk = 0;
        for (int b0 = 0; b0 < total; b0 += 64) {               // stable in-place compaction (pos <= e)
            const int e = b0 + lane;
            const u64 k = e < total ? keys[e] : 0ull;
            const bool keep = e < total && k >= thr;
            const unsigned long long m = __ballot(keep);
            const int pos = nk + __popcll(m & ((1ull << lane) - 1ull));
            if (keep) keys[pos] = k;
            nk += __popcll(m);
        }
        total = nk;
    }
    if (lane == 0) keys[total] = 0ull;                       // pad for the 2-wide reads below
    // rank by counting (keys are unique: the index is part of the key)
    for (int e = lane; e < total; e += 64) {
        const u64 ke = keys[e];
        int rk = 0;
        for (int f = 0; f < total; f += 2) {
            const ulonglong2 kf = *reinterpret_cast<const ulonglong2*>(keys + f);
            rk += (int)(kf.x > ke) + (int)((f + 1 < total) && (kf.y > ke));
        }
        if (rk < top_k) { sv[rk] = key_val(ke); si[rk] = key_idx(ke); }
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


// =================================================================================================================
// Wide select (large memories): 128 queries x 32-row tiles per wave, 8 waves per workgroup, ONE workgroup per CU.
//   * every key row is loaded by exactly one wave of the workgroup and contracted against 128 queries (four independent
//     32x32 accumulator chains): half the L2 -> CU key traffic of the 64-query kernel and no dependent-MFMA stalls;
//   * no accumulator double buffering: the two waves that share a SIMD cover each other's (short) VALU filter phase;
//   * candidates (v > tau0) go to per-query LDS lists and are flushed once, at the end, to ONE global list per query
//     (atomic reservation), so the merge reads a single contiguous list per query instead of one list per split;
//   * a full LDS / global list only raises the 64-query tile's overflow flag: the safe kernel (MODE 3) redoes that tile.
// tau0 comes either from the sampled bound pass (MODE 0) or from `affinity_hint_bound_kernel` below.
// =================================================================================================================
#define AFW_BQ 128
#define AFW_WAVES 8
#define AFW_CAP 80          // per (workgroup, query) LDS candidates

struct WideArgs {
    SegDev seg[XMEM_MAX_SEGMENTS];
    int n_seg, total_tiles;
    const float* qk; const float* qe;
    int HW, top_k;
    int splits, tiles_per_split, chunk, sub_tiles;
    const float* tau_init;           // [HW] valid lower bound of the k-th similarity (-inf: no bound -> tile goes to the safe kernel)
    u64* gcand; int* gcnt;           // [HW][AFW_GCAP], [HW] (zeroed by the bound kernels)
    int* ovf;                        // [ceil(HW/64)]
};

__global__ __launch_bounds__(512, 1) void affinity_wide_kernel(WideArgs p) {
    constexpr int CK = 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Bq = smem;                                 // [128][132]
    float* bsq = Bq + AFW_BQ * AFF_LDB;               // [128]
    float* tau = bsq + AFW_BQ;                        // [128]   current lower bound of the k-th similarity (only ever raised)
    int* cnt = reinterpret_cast<int*>(tau + AFW_BQ);  // [128] + flag + pad
    volatile int* flag = cnt + AFW_BQ;                // "a candidate list is full": every wave joins a tightening episode
    u64* cand = reinterpret_cast<u64*>(cnt + AFW_BQ + 4); // [128][AFW_CAP]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.x * AFW_BQ;
    const int split = blockIdx.y;
    const int keep = p.top_k > 32 ? p.top_k : 32;     // entries a list is cut back to (>= top_k)

    {   // stage the query operand (same arithmetic as affinity_kernel: bit-identical similarities)
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
                bs = bsq_term(bs, e, k);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) { row[part * 16 + c] = 0.f; row[CK + part * 16 + c] = 0.f; }
        }
        bs += __shfl_xor(bs, 1, 64);
        bs += __shfl_xor(bs, 2, 64);
        if (part == 0) {
            bsq[q] = p.qe ? bs : 0.f;
            // tau0 is reached by >= k elements; elements equal to it must still pass the strict test below.
            // -inf (no bound at all) is fine: the lists tighten themselves; queries past HW never collect anything.
            const float t0 = (qg < p.HW) ? (p.tau_init ? p.tau_init[qg] : -INFINITY) : INFINITY;
            tau[q] = (t0 == INFINITY || t0 == -INFINITY) ? t0 : nextafterf(t0, -INFINITY);
            cnt[q] = 0;
        }
        if (tid == 0) *flag = 0;
    }
    __syncthreads();

    const int t_begin = p.chunk ? 0 : split * p.tiles_per_split;
    const int t_end = p.chunk ? p.tiles_per_split : min(p.sub_tiles, t_begin + p.tiles_per_split);

    // next tile's key rows: unconditional loads (row index clamped into the segment: rows past its end are excluded by the
    // filter's row test, never by exec-masked branches around the loads)
    f32x4 an[8]; float msn = 1.f;
    int n_segn = 1, n_base = 0, n_row0 = 0; bool n_active = false;
    auto issue_loads = [&](int tile) {
        bool off = tile >= t_end;
        if (p.chunk && !off) {
            const int ci = tile / p.chunk, wi = tile - ci * p.chunk;
            tile = (ci * p.splits + split) * p.chunk + wi;
            off = tile >= p.sub_tiles;
        }
        n_active = !off;
        if (off) return;                              // wave-uniform
        int sg = 0;
#pragma unroll
        for (int i = 1; i < XMEM_MAX_SEGMENTS; ++i)
            if (i < p.n_seg && tile >= p.seg[i].tile0) sg = i;
        const float* key = p.seg[sg].key; const float* shr = p.seg[sg].shr;
        n_segn = p.seg[sg].n; n_base = p.seg[sg].base;
        n_row0 = (tile - p.seg[sg].tile0) * AFF_ROWS;
        const int r = min(n_row0 + l31, n_segn - 1);
        const float* src = key + (size_t)r * CK + lh * 4;
#pragma unroll
        for (int t = 0; t < 8; ++t) an[t] = *reinterpret_cast<const f32x4*>(src + t * 8);
        msn = shr ? shr[r] : 1.f;
    };

    // ---- self-tightening candidate lists -------------------------------------------------------------------------------
    // A lane appends the values of one 32x32 block all-or-nothing (one reservation).  If a list has no room the lane marks the
    // block for a redo and raises `flag`; every wave joins the episode at its next tile boundary (or from the drain loop at
    // the end): barrier, each list longer than `keep` is cut back to its best `keep` entries and its bound raised to the k-th
    // of them (>= k elements reach it, so it is a valid lower bound of the true k-th), barrier.  The wave then contracts the
    // same tile again and stores only the marked blocks - no vector state lives across an episode.  A list can therefore
    // never lose a candidate, whatever the initial bound was (-inf included): no overflow path, no second kernel.
    auto compact = [&](int q, int c) {                // whole wave, wave-uniform q; all c <= AFW_CAP entries are written
        u64* L = cand + (size_t)q * AFW_CAP;
        const u64 k0 = lane < c ? L[lane] : 0ull;
        const u64 k1 = lane + 64 < c ? L[lane + 64] : 0ull;
        int r0 = 0, r1 = 0;
        for (int f = 0; f < c; ++f) { const u64 kf = L[f]; r0 += (int)(kf > k0); r1 += (int)(kf > k1); }
        __builtin_amdgcn_wave_barrier();
        if (lane < c && r0 < keep) L[r0] = k0;        // keys are unique: the ranks are a permutation
        if (lane + 64 < c && r1 < keep) L[r1] = k1;
        if (lane < c && r0 == p.top_k - 1) tau[q] = fmaxf(tau[q], nextafterf(key_val(k0), -INFINITY));
        if (lane + 64 < c && r1 == p.top_k - 1) tau[q] = fmaxf(tau[q], nextafterf(key_val(k1), -INFINITY));
        if (lane == 0) cnt[q] = c < keep ? c : keep;
    };
    auto episode = [&]() -> bool {                    // every wave of the workgroup calls this the same number of times
        __syncthreads();
        const bool f = *flag != 0;                    // uniform: the flag is only cleared after the second barrier, which no
        if (f) {                                      // wave can pass before every wave has read it here
            for (int q = wave; q < AFW_BQ; q += AFW_WAVES) {
                int c = cnt[q];
                c = c > AFW_CAP ? AFW_CAP : c;
                if (c > keep) compact(q, c);
            }
            __syncthreads();
            if (tid == 0) *flag = 0;                  // a flag raised again this early could be lost: harmless, that wave's next
        }                                             // attempt fails again and raises it again (a retry is >= one full tile away)
        return f;
    };

    float my_tau[4], my_bs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) my_bs[i] = bsq[i * 32 + l31];
    constexpr float inv_sqrt = 0.125f;               // sqrt(C_k) = 8: (x * ms) / 8 == x * (ms / 8) bit for bit

    int cur = t_begin + wave;                         // this wave's current tile (units of its split's tile list)
    unsigned redo = 0;                                // per lane: blocks of tile `cur` that found their list full
    issue_loads(cur);
    for (;;) {
        if (*flag) episode();                         // wave-uniform LDS read
        if (!n_active) break;                         // past the end of this wave's tiles
#pragma unroll
        for (int i = 0; i < 4; ++i) my_tau[i] = tau[i * 32 + l31];
        f32x4 a[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) a[t] = an[t];
        const float ms_mine = msn;
        const int row0 = n_row0, segn = n_segn, base = n_base;
        const bool is_redo = __any(redo != 0);
        issue_loads(cur + AFW_WAVES);                 // next tile's rows in flight under this tile's MFMAs
        f32x16 c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float nb = -my_bs[i];
            asm volatile("" : "+v"(nb));              // re-materialise per tile (64 loop-invariant registers otherwise)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[i][r] = nb;
        }
        const float* bq = Bq + l31 * AFF_LDB + lh * 4;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            f32x4 blo[4], bhi[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                blo[i] = *reinterpret_cast<const f32x4*>(bq + i * 32 * AFF_LDB + t * 8);
                bhi[i] = *reinterpret_cast<const f32x4*>(bq + i * 32 * AFF_LDB + CK + t * 8);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = a[t][j];
                const float xx = x * x;
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(xx, blo[i][j], c[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, bhi[i][j], c[i], 0, 0, 0);
            }
        }
        float msr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) msr[r] = __shfl(ms_mine, (r & 3) + 8 * (r >> 2) + 4 * lh, 64) * inv_sqrt;
        const bool full = row0 + AFF_ROWS <= segn;        // wave-uniform
        unsigned todo = 0;                                // blocks of this lane to store: those that pass the bound
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float best = -INFINITY;
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) best = fmaxf(best, c[i][r] * msr[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    best = fmaxf(best, rr < segn ? c[i][r] * msr[r] : -INFINITY);
                }
            }
            if (best > my_tau[i]) todo |= 1u << i;
        }
        if (is_redo) todo &= redo;                        // a repeated tile stores only what failed before
        redo = 0;
        if (__any(todo != 0)) {                           // rare: some lane of the wave has candidates
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!(todo & (1u << i))) continue;
                const int q = i * 32 + l31;
                const float tq = my_tau[i];
                int n = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    n += (int)(rr < segn && c[i][r] * msr[r] > tq);
                }
                if (n == 0) continue;
                // reserve n slots or nothing (a failed add-then-give-back would corrupt the count if another lane's reservation
                // succeeded in between): compare-and-swap on the list length
                int slot = *reinterpret_cast<volatile int*>(&cnt[q]);
                for (;;) {
                    if (slot + n > AFW_CAP) { slot = -1; break; }
                    const int seen = atomicCAS(&cnt[q], slot, slot + n);
                    if (seen == slot) break;
                    slot = seen;
                }
                if (slot >= 0) {
                    int w2 = slot;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float x = c[i][r] * msr[r];
                        if (rr < segn && x > tq) cand[(size_t)q * AFW_CAP + w2++] = pack_key(x, base + rr);
                    }
                } else {                                   // no room: the list is cut back in the episode, the tile repeated
                    *flag = 1;
                    redo |= 1u << i;
                }
            }
        }
        if (__any(redo != 0)) issue_loads(cur);           // same tile again (after the episode at the loop top)
        else cur += AFW_WAVES;
    }
    while (episode()) {}                                  // drain: other waves may still need partners for an episode
    // ---- flush: every list cut back to <= keep entries, then copied behind the query's global list ----
    for (int q = wave; q < AFW_BQ; q += AFW_WAVES) {
        const int qg = q0 + q;
        if (qg >= p.HW) continue;
        int c = cnt[q];
        if (c == 0) continue;
        if (c > keep) { compact(q, c); __builtin_amdgcn_wave_barrier(); c = keep; }
        int gb = 0;
        if (lane == 0) gb = atomicAdd(&p.gcnt[qg], c);
        gb = __shfl(gb, 0, 64);
        for (int s2 = lane; s2 < c && gb + s2 < AFW_GCAP; s2 += 64) p.gcand[(size_t)qg * AFW_GCAP + gb + s2] = cand[(size_t)q * AFW_CAP + s2];
    }
}

// Bound from a hint: `hint_idx` holds k_h indices per query from an EARLIER call (the previous frame's top-k of the same
// store list).  Any k distinct memory elements give a valid lower bound of the k-th largest similarity: the k-th largest
// of their similarities.  In a video the previous frame's matches of a query and of its four grid neighbours are nearly
// the current ones, so this bound is far tighter than a sampled pass (and costs ~k rows per query instead of N/4).
// The similarities are evaluated on the VALU in a different summation order than the MFMA chain of the select kernels;
// a rigorous fp32 bound on that difference (2 * 128 * 2^-24 * sum|terms|) is subtracted, so "at least k elements pass
// tau0" holds for the values the select kernel computes.  Also zeroes the per-query list counters and overflow flags.
struct HintArgs {
    SegDev seg[XMEM_MAX_SEGMENTS]; int n_seg, n_total;
    int old_n[XMEM_MAX_SEGMENTS]; int new_of_old[XMEM_MAX_SEGMENTS]; int old_seg;   // segment slots of the call that produced hint_idx
    const int* hint_idx; int hint_k; int grid_w;
    const float* qk; const float* qe; int HW, top_k;
    float* tau0; int* gcnt; int* ovf;
    _Float16* qop16; float* qmeta;       // optional: query operands of the fp16 filter (affinity_filter.hip)
    int* flag1; int* flag2;              // optional: per-128-query-tile flags of the two filter passes, zeroed here
    int* fcnt;                           // optional: the filter's list counters [HW][F16_CS], zeroed here
};
#define HINT_MAXC 320       // 5 queries x 64 indices
#define HINT_NB 12          // entries taken from each grid neighbour's list

__global__ __launch_bounds__(256) void affinity_hint_bound_kernel(HintArgs p) {
    constexpr int CK = 64;
    __shared__ __attribute__((aligned(16))) float s_op[4][2 * CK];
    __shared__ __attribute__((aligned(16))) float s_val[4][HINT_MAXC];
    __shared__ int s_tab[4][512];                    // open-addressing set of the candidate indices (duplicates -> -inf)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wv;
    if (q >= p.HW) return;
    float* op = s_op[wv]; float* val = s_val[wv]; int* tab = s_tab[wv];
    float bs = 0.f;
    {
        const float k = p.qk[(size_t)q * CK + lane];
        const float e = p.qe ? p.qe[(size_t)q * CK + lane] : 1.f;
        op[lane] = -e; op[CK + lane] = 2.f * (k * e);
        bs = p.qe ? e * (k * k) : 0.f;
    }
    if (p.qop16) {
        // query operand row of the fp16 filter (layout: affinity_common.hpp), once per call; b_sq exactly as the select kernels
        // stage it (four parts of 16 sequential terms, then the pairwise tree) for the exact refine pass
        const float k = p.qk[(size_t)q * CK + lane];
        const float e = p.qe ? p.qe[(size_t)q * CK + lane] : 1.f;
        const float ke2 = 2.f * (k * e);
        _Float16* orow = p.qop16 + (size_t)q * F16_K;
        orow[lane] = (_Float16)(-e);
        orow[CK + lane] = (_Float16)ke2;
        const float sC = wave_sum(e * e), sD = wave_sum(ke2 * ke2);
        const float mx = wave_max(fmaxf(fabsf(e), fabsf(ke2)));
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float kc = __shfl(k, 16 * (lane & 3) + c, 64), ec = __shfl(e, 16 * (lane & 3) + c, 64);
            part = bsq_term(part, ec, kc);
        }
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        const float bsq = p.qe ? part : 0.f;
        if (lane == 0) {
            f32x4 m; m[0] = bsq; m[1] = 0.f; m[2] = 0.f; m[3] = 0.f;
            *reinterpret_cast<f32x4*>(p.qmeta + (size_t)q * 4) = m;
        }
        if (lane < 16) {
            float C = sqrtf(sC) * 1.0001f, D = sqrtf(sD) * 1.0001f;
            if (!(fmaxf(mx, fabsf(bsq)) < 6.5e4f)) { C = INFINITY; D = INFINITY; }   // beyond the fp16 range (or NaN): no bound, keep every pair
            const _Float16 bh = (_Float16)bsq;
            const _Float16 bl = (_Float16)(bsq - (float)bh);
            _Float16 v = (_Float16)0.f;
            if (lane == 0 || lane == 2) v = -bh;
            else if (lane == 1) v = -bl;
            else if (lane == 3) v = f16_up(C);
            else if (lane == 4) v = f16_up(D);
            else if (lane == 5) v = f16_up(F16_ACC * fabsf(bsq) + F16_ABS * (C + D));
            else if (lane == 6) v = (_Float16)0.0009765625f;          // 2^-10
            orow[2 * CK + lane] = v;
        }
    }
    bs = wave_sum(bs);
#pragma unroll
    for (int i = 0; i < 8; ++i) tab[lane + 64 * i] = -1;
    if (lane == 0) {
        p.gcnt[q] = 0; if ((q & 63) == 0) p.ovf[q >> 6] = 0;
        if (p.flag1 && (q & 127) == 0) { p.flag1[q >> 7] = 0; p.flag2[q >> 7] = 0; }
        if (p.fcnt) p.fcnt[(size_t)q * F16_CS] = 0;
    }
    // candidate lists: this query and its grid neighbours in the hint
    int nq[5]; int nn = 0;
    nq[nn++] = q;
    if (p.grid_w > 0) {
        const int x = q % p.grid_w;
        if (x > 0) nq[nn++] = q - 1;
        if (x + 1 < p.grid_w && q + 1 < p.HW) nq[nn++] = q + 1;
        if (q >= p.grid_w) nq[nn++] = q - p.grid_w;
        if (q + p.grid_w < p.HW) nq[nn++] = q + p.grid_w;
    }
    // the query's own list in full, the best HINT_NB entries of each neighbour's (the lists are sorted by similarity)
    const int nbk = p.hint_k < HINT_NB ? p.hint_k : HINT_NB;
    const int T = p.hint_k + (nn - 1) * nbk;
    const int T4 = (T + 3) & ~3;
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < T4; e += 64) {
        float v = -INFINITY;
        if (e < T) {
            const int li = e < p.hint_k ? 0 : 1 + (e - p.hint_k) / nbk;
            const int pos = e < p.hint_k ? e : (e - p.hint_k) - (li - 1) * nbk;
            int o = p.hint_idx[(size_t)nq[li] * p.hint_k + pos];
            // re-base from the old segment layout to today's
            // (prefix sums: the old slot is the LAST one whose first element is <= o.  Testing o against ob + old_n[i] with
            // ob advanced only on a hit sent an index of an early slot into a later, shorter one - a valid but far looser bound)
            int sgi = 0, ob = 0, cum = 0;
#pragma unroll
            for (int i = 0; i + 1 < XMEM_MAX_SEGMENTS; ++i) {
                if (i + 1 < p.old_seg) {
                    cum += p.old_n[i];
                    if (o >= cum) { ob = cum; sgi = i + 1; }
                }
            }
            o -= ob;
            {   // per-lane index into kernel-argument arrays: select, do not load (see seg_of_slot)
                int mapped = p.new_of_old[0];
#pragma unroll
                for (int i = 1; i < XMEM_MAX_SEGMENTS; ++i)
                    if (sgi == i) mapped = p.new_of_old[i];
                sgi = mapped;
            }
            if (sgi >= p.n_seg) sgi = p.n_seg - 1;
            const SegDev sd = seg_of_slot(p, sgi);
            if (o < 0) o = 0;
            if (o >= sd.n) o = sd.n - 1;
            const int gi = sd.base + o;
            bool uniq = false;
            unsigned h = ((unsigned)gi * 2654435761u) >> 23;
            for (int probe = 0; probe < 512; ++probe) {
                const int old = atomicCAS(&tab[h], -1, gi);
                if (old == -1) { uniq = true; break; }
                if (old == gi) break;
                h = (h + 1) & 511u;
            }
            if (uniq) {
                const float* row = sd.key + (size_t)o * CK;
                f32x4 xr[CK / 4];                                // the whole row in flight at once
#pragma unroll
                for (int c4 = 0; c4 < CK / 4; ++c4) xr[c4] = *reinterpret_cast<const f32x4*>(row + c4 * 4);
                float accv[4] = {0.f, 0.f, 0.f, 0.f}, aaccv[4] = {0.f, 0.f, 0.f, 0.f};     // four independent chains
#pragma unroll
                for (int c4 = 0; c4 < CK / 4; ++c4) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(op + c4 * 4);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(op + CK + c4 * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xv = xr[c4][j], xx = xv * xv;
                        const float t1 = xx * lo[j], t2 = xv * hi[j];
                        accv[j] += t1; accv[j] += t2;
                        aaccv[j] += fabsf(t1) + fabsf(t2);
                    }
                }
                const float acc = (accv[0] + accv[1]) + (accv[2] + accv[3]);
                const float aacc = (aaccv[0] + aaccv[1]) + (aaccv[2] + aaccv[3]);
                const float ms = sd.shr ? sd.shr[o] : 1.f;
                const float est = (acc - bs) * (ms * 0.125f);
                const float margin = (aacc + fabsf(bs)) * (fabsf(ms) * 0.125f) * 3.2e-5f + 1e-30f;   // > 2 * 130 * 2^-24 * sum|terms|
                v = est - margin;
            }
        }
        val[e] = v;
    }
    __builtin_amdgcn_wave_barrier();
    float res = -INFINITY;                           // stays -inf with fewer than k distinct candidates
    for (int e = lane; e < T; e += 64) {
        const float ve = val[e];
        if (ve == -INFINITY) continue;
        int rk = 0;
        for (int f = 0; f < T4; f += 4) {
            const f32x4 vf = *reinterpret_cast<const f32x4*>(val + f);
            rk += (int)((vf.x > ve) || (vf.x == ve && f < e)) + (int)((vf.y > ve) || (vf.y == ve && f + 1 < e)) +
                  (int)((vf.z > ve) || (vf.z == ve && f + 2 < e)) + (int)((vf.w > ve) || (vf.w == ve && f + 3 < e));
        }
        if (rk == p.top_k - 1) res = ve;
    }
    res = wave_max(res);
    if (lane == 0) p.tau0[q] = res;
}

// Merge for the wide select.  Light queries (<= AFM_LIGHT candidates, the normal case): 16 lanes per query, four queries
// per wave, rank by counting (keys are unique: the index is part of the key).  Heavy queries (flat image regions keep
// hundreds of near-tied candidates; tiles redone by the safe kernel): one wave per query with the lane-maxima pre-filter.
// Source: the query's global list, or - when its 64-query tile overflowed - the per-split lists of the safe kernel.
// Output: sorted top-k, softmax without max shift (memory_util.py:48-49).
#define AFM_LIGHT 64
#define AFM_HEAVY 2048      // >= AFW_GCAP
__device__ __forceinline__ int merge_count(const int* gcnt, const int* ovf, const int* part_cnt, int fsplits, int HW, int q, bool& fb,
                                           int heavy_cap) {
    fb = fsplits > 0 && ovf[q >> 6] != 0;
    if (!fb) { const int t = gcnt[q]; return t > heavy_cap ? heavy_cap : t; }
    int t = 0;
    for (int s = 0; s < fsplits; ++s) t += part_cnt[(size_t)s * HW + q];
    return t > heavy_cap ? heavy_cap : t;
}
__global__ __launch_bounds__(256) void affinity_merge16_kernel(const u64* __restrict__ gcand, const int* __restrict__ gcnt,
                                                               const int* __restrict__ ovf, const u64* __restrict__ part_key,
                                                               const int* __restrict__ part_cnt, int fsplits, int HW, int top_k,
                                                               int heavy_cap, float* __restrict__ out_w, int* __restrict__ out_idx,
                                                               float* __restrict__ out_sim) {
    __shared__ __attribute__((aligned(16))) u64 s_keys[16][AFM_LIGHT + 2];
    extern __shared__ __attribute__((aligned(16))) u64 s_heavy_dyn[];        // [4][heavy_cap + 2]: the longest list a query can have
    __shared__ float s_v[16][AFF_MAX_TOPK];
    __shared__ int s_i[16][AFF_MAX_TOPK];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int q = blockIdx.x * 16 + g;
    float* sv = s_v[g]; int* si = s_i[g];
    for (int r = l; r < top_k; r += 16) { sv[r] = -INFINITY; si[r] = 0; }     // never an out-of-range index, whatever happens upstream
    bool fb = false;
    const int total = q < HW ? merge_count(gcnt, ovf, part_cnt, fsplits, HW, q, fb, heavy_cap) : 0;
    const bool light = q < HW && total <= AFM_LIGHT;
    if (light) {
        u64* keys = s_keys[g];
        if (fb) {
            int off = 0;
            for (int s = 0; s < fsplits; ++s) {
                const int c = part_cnt[(size_t)s * HW + q];
                for (int e = l; e < c; e += 16) keys[off + e] = part_key[((size_t)s * HW + q) * AFF_OUTCAP + e];
                off += c;
            }
        } else {
            for (int e = l; e < total; e += 16) keys[e] = gcand[(size_t)q * AFW_GCAP + e];
        }
        if (l == 0) keys[total] = 0ull;
        __builtin_amdgcn_wave_barrier();              // the 16 lanes of a group are one quarter of a wave: LDS ops are in order
        for (int e = l; e < total; e += 16) {
            const u64 ke = keys[e];
            int rk = 0;
            for (int f = 0; f < total; f += 2) {
                const ulonglong2 kf = *reinterpret_cast<const ulonglong2*>(keys + f);
                rk += (int)(kf.x > ke) + (int)((f + 1 < total) && (kf.y > ke));
            }
            if (rk < top_k) { sv[rk] = key_val(ke); si[rk] = key_idx(ke); }
        }
    }
    // heavy queries of this wave (its four 16-lane groups), one at a time on all 64 lanes
    for (int j = 0; j < 4; ++j) {
        const int tj = __shfl(total, j * 16, 64);
        const int qj = blockIdx.x * 16 + wv * 4 + j;
        if (qj >= HW || tj <= AFM_LIGHT) continue;
        const bool fbj = __shfl((int)fb, j * 16, 64) != 0;
        u64* keys = s_heavy_dyn + (size_t)wv * (heavy_cap + 2);
        float* svj = s_v[wv * 4 + j]; int* sij = s_i[wv * 4 + j];
        int T = 0;
        if (fbj) {
            for (int s = 0; s < fsplits && T < heavy_cap; ++s) {
                int c = part_cnt[(size_t)s * HW + qj];
                if (T + c > heavy_cap) c = heavy_cap - T;
                for (int e = lane; e < c; e += 64) keys[T + e] = part_key[((size_t)s * HW + qj) * AFF_OUTCAP + e];
                T += c;
            }
        } else {
            T = tj;
            for (int e = lane; e < T; e += 64) keys[e] = gcand[(size_t)qj * AFW_GCAP + e];
        }
        __builtin_amdgcn_wave_barrier();
        // pre-filter: the top_k-th largest of the 64 lane-local maxima is reached by >= top_k candidates
        u64 lmax = 0ull;
        for (int e = lane; e < T; e += 64) { const u64 k = keys[e]; lmax = k > lmax ? k : lmax; }
        int rk0 = 0;
        for (int jj = 0; jj < 64; ++jj) {
            const unsigned lo = __shfl((unsigned)lmax, jj, 64), hi = __shfl((unsigned)(lmax >> 32), jj, 64);
            const u64 o = ((u64)hi << 32) | lo;
            rk0 += (o > lmax) || (o == lmax && jj < lane);
        }
        const unsigned long long selm = __ballot(rk0 == top_k - 1);
        const int src = selm ? __ffsll((long long)selm) - 1 : 0;
        const unsigned tlo = __shfl((unsigned)lmax, src, 64), thi = __shfl((unsigned)(lmax >> 32), src, 64);
        const u64 thr = selm ? (((u64)thi << 32) | tlo) : 0ull;
        int nk = 0;
        for (int b0 = 0; b0 < T; b0 += 64) {                   // stable in-place compaction (pos <= e)
            const int e = b0 + lane;
            const u64 k = e < T ? keys[e] : 0ull;
            const bool keep = e < T && k >= thr;
            const unsigned long long m = __ballot(keep);
            const int pos = nk + __popcll(m & ((1ull << lane) - 1ull));
            __builtin_amdgcn_wave_barrier();
            if (keep) keys[pos] = k;
            nk += __popcll(m);
        }
        T = nk;
        if (lane == 0) keys[T] = 0ull;
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < T; e += 64) {
            const u64 ke = keys[e];
            int rk = 0;
            for (int f = 0; f < T; f += 2) {
                const ulonglong2 kf = *reinterpret_cast<const ulonglong2*>(keys + f);
                rk += (int)(kf.x > ke) + (int)((f + 1 < T) && (kf.y > ke));
            }
            if (rk < top_k) { svj[rk] = key_val(ke); sij[rk] = key_idx(ke); }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (q >= HW) return;
    __builtin_amdgcn_wave_barrier();
    float s = 0.f;
    for (int r = l; r < top_k; r += 16) s += expf(sv[r]);
    s += __shfl_xor(s, 8, 16); s += __shfl_xor(s, 4, 16); s += __shfl_xor(s, 2, 16); s += __shfl_xor(s, 1, 16);
    for (int r = l; r < top_k; r += 16) {
        const float v = sv[r];
        out_w[(size_t)q * top_k + r] = expf(v) / s;
        out_idx[(size_t)q * top_k + r] = si[r];
        if (out_sim) out_sim[(size_t)q * top_k + r] = v;
    }
}

namespace {
struct AffPlan { int splits, tiles_per_split, sub_tiles, limit, cap; size_t lds; };

AffPlan aff_plan(int sub_tiles, int HW, int top_k, bool bound, int n_rows = 0) {
    AffPlan pl;
    pl.sub_tiles = sub_tiles;
    const int qtiles = cdiv(HW, AFF_BQ);
    int s;
    if (bound) {
        s = 512 / qtiles;                        // small LDS: two workgroups per CU
        if (s > 16) s = 16;                      // keeps the survivor list of the bound reduction short
        // The bound is the k-th largest of one survivor per lane group (8 groups per split), so it is only tight while the
        // best matches of different memory frames fall into different groups: keep ~4 groups per memory frame
        // (32 frames -> 16 splits; 256 frames at 8 splits let ~1000 candidates per query through and overflowed every frame).
        const int frames = n_rows / (HW > 0 ? HW : 1);
        int want = frames / 2;
        if (want > AFF_MAX_BOUND_SPLITS) want = AFF_MAX_BOUND_SPLITS;
        if (want > s) s = want;
    } else {
        // two workgroups per CU: aim for whole rounds of the 512 slots, >= 16 tiles (4 steps) per split
        s = 512 / qtiles;
        if (s < 1) s = 1;
        if (sub_tiles / s < 16) s = 256 / qtiles;
        if (s < 16 && sub_tiles / 16 >= 16) {
            // many query tiles (HW > 2048): the candidates a query keeps (~ R*k, far more in low-texture regions) must
            // still spread over enough splits to fit the optimistic buffers - with 512/qtiles = 8 splits the 720p / 256-frame
            // configuration overflowed on every frame.  Smallest s >= 16 that fills its last round of slots >= 90 %.
            int best = 16; double best_eff = 0.0;
            for (int c = 16; c <= 32 && sub_tiles / c >= 16; ++c) {
                const long wgs = (long)qtiles * c;
                const double eff = (double)wgs / (double)(((wgs + 511) / 512) * 512);
                if (eff > best_eff + 1e-9) { best_eff = eff; best = c; }
                if (eff >= 0.9) { best = c; break; }
            }
            s = best;
        }
    }
    if (s < 1) s = 1;
    int maxs = sub_tiles / 8; if (maxs < 1) maxs = 1;
    if (s > maxs) s = maxs;
    if (s > 64) s = 64;
    pl.tiles_per_split = cdiv(sub_tiles, s);
    pl.splits = cdiv(sub_tiles, pl.tiles_per_split);
    pl.limit = top_k <= 96 ? 96 : (top_k + 7) / 8 * 8;      // re-rank when a buffer holds more than `limit` entries
    pl.cap = pl.limit + AFF_STEP_ROWS;
    pl.lds = ((size_t)AFF_BQ * AFF_LDB + 3 * AFF_BQ + 4) * sizeof(float) + (bound ? 0 : (size_t)AFF_BQ * pl.cap * sizeof(u64));
    return pl;
}

#define AFF_OPT_CAP 88     // optimistic candidate buffer (expected fill ~ R*k/splits = 6..8, observed max 54 on redundant memories):
                           // 45 KB + 33.8 KB of query operand = 78.9 KB -> still two workgroups per 160 KB CU
inline size_t opt_lds() { return ((size_t)AFF_BQ * AFF_LDB + 3 * AFF_BQ + 4) * sizeof(float) + (size_t)AFF_BQ * AFF_OPT_CAP * sizeof(u64); }

inline int bound_stride(int total_tiles) {
    // the bound pass samples every 4th 32-row tile (25 % extra MFMA work) once the memory is large enough
    return total_tiles >= 256 ? 4 : 1;
}

// XMEM_AFFINITY_FILTER16=0 keeps hinted calls on the fp32 select (A/B measurements, tests of both pipelines)
inline bool aff_use_filter16() {
    const char* e = getenv("XMEM_AFFINITY_FILTER16");
    return !(e && e[0] == '0');
}

struct WsLayout { size_t key_off, cnt_off, bound_off, tau_off, ovf_off, gcand_off, gcnt_off, spill_off, qop16_off, qmeta_off, rows16_off, gcand32_off, flag_off, fcnt_off, total; int fsplits; };
// fallback (MODE 3) split count: efficiency is irrelevant on this rare path, its worst-case global candidate buffers are not
#define AFF_FB_GRID 128     // persistent workgroups of the safe fallback pass (a scene cut flags every tile: ~0.7 ms at B32)
inline int fallback_splits(int HW) { (void)HW; return 16; }
WsLayout ws_layout(int HW, int n_total) {
    WsLayout w;
    w.fsplits = fallback_splits(HW);
    w.key_off = 0;                                                     // [<= 64 splits][HW][AFF_OUTCAP] (MODE 2 / MODE 3 lists)
    w.cnt_off = align_up((size_t)64 * HW * AFF_OUTCAP * sizeof(u64), 256);
    w.bound_off = w.cnt_off + align_up((size_t)64 * HW * sizeof(int), 256);
    w.tau_off = w.bound_off + align_up((size_t)AFF_MAX_BOUND_SPLITS * HW * AFF_BOUND_SLOTS * sizeof(float), 256);
    w.ovf_off = w.tau_off + align_up((size_t)HW * sizeof(float), 256);
    w.gcand_off = w.ovf_off + align_up((size_t)cdiv(HW, AFF_BQ) * sizeof(int), 256);
    w.gcnt_off = w.gcand_off + align_up((size_t)HW * AFW_GCAP * sizeof(u64), 256);
    w.spill_off = w.gcnt_off + align_up((size_t)HW * sizeof(int), 256);
    // fallback buffers: one per persistent workgroup, cap = 96 + 128
    w.qop16_off = w.spill_off + align_up((size_t)AFF_FB_GRID * AFF_BQ * (96 + AFF_STEP_ROWS) * sizeof(u64), 256);
    // fp16 filter: query operands, per-query meta, operand rows of segments whose caller keeps none, candidate lists, flags
    w.qmeta_off = w.qop16_off + align_up((size_t)HW * F16_K * sizeof(_Float16), 256);
    w.rows16_off = w.qmeta_off + align_up((size_t)HW * 4 * sizeof(float), 256);
    w.gcand32_off = w.rows16_off + align_up(aff_filter16_rows_bytes(n_total), 256);
    w.flag_off = w.gcand32_off + align_up((size_t)HW * aff_filter16_list_stride(n_total) * sizeof(int), 256);
    w.fcnt_off = w.flag_off + align_up((size_t)2 * cdiv(HW, AFW_BQ) * sizeof(int), 256);
    w.total = w.fcnt_off + align_up((size_t)HW * F16_CS * sizeof(int), 256);
    return w;
}
}  // namespace

extern "C" size_t xmem_affinity_topk_workspace_bytes(int n_total, int HW, int top_k) {
    if (n_total <= 0 || HW <= 0 || top_k <= 0) return 0;
    return ws_layout(HW, n_total).total;
}

extern "C" int xmem_affinity_debug_offsets(int n_total, int HW, size_t* count_off, size_t* flag_off, size_t* bound_off) {
    if (n_total <= 0 || HW <= 0 || !count_off || !flag_off || !bound_off) return XMEM_ERR_BAD_ARG;
    const WsLayout w = ws_layout(HW, n_total);
    *count_off = w.gcnt_off; *flag_off = w.flag_off; *bound_off = w.tau_off;     // flags: [pass 1 | pass 2], ceil(HW/128) each
    return XMEM_OK;
}

extern "C" int xmem_affinity_topk_hinted(const xmem_key_segment* segs, int n_seg, const float* qk, const float* qe, int Ck, int HW,
                                         int top_k, const xmem_affinity_hint* hint, float* out_w, int32_t* out_idx, float* out_sim,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    if (!segs || n_seg <= 0 || n_seg > XMEM_MAX_SEGMENTS || !qk || !out_w || !out_idx || HW <= 0) return XMEM_ERR_BAD_ARG;
    if (Ck != 64) return XMEM_ERR_UNSUPPORTED;
    if (top_k < 1 || top_k > AFF_MAX_TOPK) return XMEM_ERR_UNSUPPORTED;
    AffArgs a;
    int tiles = 0, base = 0, ns = 0;
    int pos_map[XMEM_MAX_SEGMENTS];                     // caller's segment slot -> compacted (non-empty) segment, or -1
    for (int i = 0; i < n_seg; ++i) {
        if (segs[i].n < 0) return XMEM_ERR_BAD_ARG;
        pos_map[i] = -1;
        if (segs[i].n == 0) { continue; }
        if (!segs[i].key) return XMEM_ERR_BAD_ARG;
        pos_map[i] = ns;
        a.seg[ns].key = segs[i].key; a.seg[ns].shr = segs[i].shrinkage; a.seg[ns].n = segs[i].n;
        a.seg[ns].base = base; a.seg[ns].tile0 = tiles; a.seg[ns].pad = 0;
        a.seg[ns].rows16 = reinterpret_cast<const _Float16*>(segs[i].rows16);
        tiles += cdiv(segs[i].n, AFF_ROWS); base += segs[i].n; ++ns;
    }
    if (base < top_k) return XMEM_ERR_TOPK;
    for (int i = ns; i < XMEM_MAX_SEGMENTS; ++i) { a.seg[i].key = nullptr; a.seg[i].shr = nullptr; a.seg[i].n = 0; a.seg[i].base = base; a.seg[i].tile0 = tiles; a.seg[i].pad = 0; a.seg[i].rows16 = nullptr; }
    const WsLayout wl = ws_layout(HW, base);
    if (!workspace || workspace_bytes < wl.total) return XMEM_ERR_WORKSPACE;
    char* ws = reinterpret_cast<char*>(workspace);
    a.n_seg = ns; a.total_tiles = tiles; a.qk = qk; a.qe = qe; a.HW = HW; a.top_k = top_k;
    a.sqrt_ck = sqrtf((float)Ck);
    { int e = 0; const float m = frexpf(a.sqrt_ck, &e); a.sqrt_is_pow2 = (m == 0.5f); }
    a.part_key = reinterpret_cast<u64*>(ws + wl.key_off);
    a.part_cnt = reinterpret_cast<int*>(ws + wl.cnt_off);
    a.bound_part = reinterpret_cast<float*>(ws + wl.bound_off);
    float* tau0 = reinterpret_cast<float*>(ws + wl.tau_off);
    int* ovf = reinterpret_cast<int*>(ws + wl.ovf_off);
    u64* gcand = reinterpret_cast<u64*>(ws + wl.gcand_off);
    int* gcnt = reinterpret_cast<int*>(ws + wl.gcnt_off);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc;
    a.tau_init = nullptr; a.ovf = nullptr; a.cand_spill = nullptr; a.merge_splits = 0;

    if (tiles < 256 || top_k > AFW_CAP) {
        // ---- small memories: one safe pass (worst-case LDS buffers + exact re-rank valve), per-split lists, wave-per-query merge
        AffPlan pl = aff_plan(tiles, HW, top_k, false);
        a.limit = pl.limit; a.cap = pl.cap; a.splits = pl.splits; a.tiles_per_split = pl.tiles_per_split; a.sub_tiles = pl.sub_tiles;
        a.tile_stride = 1; a.chunk = 0;
        auto kern = affinity_kernel<64, 2>;
        if ((rc = xmem_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), pl.lds)) != XMEM_OK) return rc;
        hipLaunchKernelGGL(kern, dim3(cdiv(HW, AFF_BQ), pl.splits), dim3(256), pl.lds, s, a);
        if ((rc = xmem_check_launch()) != XMEM_OK) return rc;
        const size_t mlds = ((size_t)pl.splits * AFF_OUTCAP + 2) * sizeof(u64) + (size_t)2 * top_k * sizeof(float);
        hipLaunchKernelGGL(affinity_merge_kernel, dim3(HW), dim3(64), mlds, s, a.part_key, a.part_cnt, pl.splits, HW, top_k,
                           out_w, out_idx, out_sim);
        return xmem_check_launch();
    }

    // ---- large memories: bound (hint or sampled pass) -> wide select -> safe kernel on overflowed tiles -> merge ----
    const int qt128 = cdiv(HW, AFW_BQ);
    WideArgs w;
    for (int i = 0; i < XMEM_MAX_SEGMENTS; ++i) w.seg[i] = a.seg[i];
    w.n_seg = ns; w.total_tiles = tiles; w.qk = qk; w.qe = qe; w.HW = HW; w.top_k = top_k;
    w.tau_init = tau0; w.gcand = gcand; w.gcnt = gcnt; w.ovf = ovf;
    // one 8-wave workgroup per CU: splits so that query tiles x splits fills (at most) the 256 CUs, >= 8 tiles per wave
    int sp = 256 / qt128; if (sp < 1) sp = 1;
    { int maxs = tiles / (8 * AFW_WAVES); if (maxs < 1) maxs = 1; if (sp > maxs) sp = maxs; }
    if (sp > 32) sp = 32;                              // AFW_GCAP = 32 splits x 64 entries
    w.sub_tiles = tiles;
    w.tiles_per_split = cdiv(tiles, sp);
    w.splits = cdiv(tiles, w.tiles_per_split);
    // chunks of ~half a memory frame dealt round-robin to the splits (temporal clusters of good matches would otherwise pile
    // up in one split's candidate lists), when every split gets >= 16 chunks
    int sel_chunk = cdiv(HW, AFF_ROWS) / 2; if (sel_chunk < AFW_WAVES) sel_chunk = AFW_WAVES;
    sel_chunk = cdiv(sel_chunk, AFW_WAVES) * AFW_WAVES;
    const int n_chunks = cdiv(tiles, sel_chunk);
    if (n_chunks < 16 * w.splits) sel_chunk = 0;
    w.chunk = sel_chunk;
    if (sel_chunk) w.tiles_per_split = cdiv(n_chunks, w.splits) * sel_chunk;

    const bool hinted = hint && hint->idx && hint->top_k >= top_k && hint->top_k <= 64 && hint->n_seg == n_seg;
    if (hinted) {
        HintArgs h;
        for (int i = 0; i < XMEM_MAX_SEGMENTS; ++i) { h.seg[i] = a.seg[i]; h.old_n[i] = 0; }
        h.n_seg = ns; h.n_total = base;
        int last = 0;                                   // old slot -> today's compacted segment of the same slot (or a neighbour)
        for (int i = 0; i < n_seg; ++i) {
            h.old_n[i] = hint->seg_n[i] > 0 ? hint->seg_n[i] : 0;
            if (pos_map[i] >= 0) last = pos_map[i];
            h.new_of_old[i] = pos_map[i] >= 0 ? pos_map[i] : last;
        }
        h.old_seg = n_seg;
        h.hint_idx = hint->idx; h.hint_k = hint->top_k; h.grid_w = hint->grid_w > 0 ? hint->grid_w : 0;
        h.qk = qk; h.qe = qe; h.HW = HW; h.top_k = top_k; h.tau0 = tau0; h.gcnt = gcnt; h.ovf = ovf;
        const bool use16 = aff_use_filter16();
        h.qop16 = use16 ? reinterpret_cast<_Float16*>(ws + wl.qop16_off) : nullptr;
        h.qmeta = use16 ? reinterpret_cast<float*>(ws + wl.qmeta_off) : nullptr;
        h.flag1 = use16 ? reinterpret_cast<int*>(ws + wl.flag_off) : nullptr;
        h.flag2 = use16 ? h.flag1 + cdiv(HW, AFW_BQ) : nullptr;
        h.fcnt = use16 ? reinterpret_cast<int*>(ws + wl.fcnt_off) : nullptr;
        hipLaunchKernelGGL(affinity_hint_bound_kernel, dim3(cdiv(HW, 4)), dim3(256), 0, s, h);
        if ((rc = xmem_check_launch()) != XMEM_OK) return rc;
        if (use16) {
            // fp16 filter + exact fp32 refine (affinity_filter.hip): same outputs as the fp32 select below, bit for bit
            Filter16Args f;
            for (int i = 0; i < XMEM_MAX_SEGMENTS; ++i) f.seg[i] = a.seg[i];
            f.n_seg = ns; f.total_tiles = tiles; f.qk = qk; f.qe = qe; f.HW = HW; f.top_k = top_k;
            f.splits = 0; f.tiles_per_split = 0;
            f.qop16 = h.qop16; f.qmeta = h.qmeta;
            f.rows16 = reinterpret_cast<_Float16*>(ws + wl.rows16_off);
            f.tau = tau0; f.gcand32 = reinterpret_cast<int*>(ws + wl.gcand32_off); f.gcnt = h.fcnt; f.cnt_diag = gcnt; f.lcap = aff_filter16_list_cap(base);
            f.lcap1 = f.lcap; f.lstride = aff_filter16_list_stride(base);
            f.flag1 = h.flag1; f.flag2 = h.flag2; f.only = nullptr; f.flag_out = nullptr;
            f.out_w = out_w; f.out_idx = out_idx; f.out_sim = out_sim;
            return aff_filter16_launch(f, stream);
        }
    } else {
        // sampled bound pass: every 4th 32-row tile (every 8th for very large chunk-dealt memories)
        int R = 4;
        if (sel_chunk && tiles >= 8192) R = 8;
        AffPlan pa = aff_plan(cdiv(tiles, R), HW, top_k, true, base);
        a.cap = 0; a.limit = 0; a.splits = pa.splits; a.tiles_per_split = pa.tiles_per_split; a.sub_tiles = pa.sub_tiles;
        a.tile_stride = R; a.chunk = 0;
        hipLaunchKernelGGL((affinity_kernel<64, 0>), dim3(cdiv(HW, AFF_BQ), pa.splits), dim3(256), pa.lds, s, a);
        if ((rc = xmem_check_launch()) != XMEM_OK) return rc;
        const int T = pa.splits * AFF_BOUND_SLOTS;
        hipLaunchKernelGGL(affinity_bound_kernel, dim3(cdiv(HW, 4)), dim3(256), (size_t)4 * T * sizeof(float), s,
                           a.bound_part, pa.splits, HW, top_k, tau0, gcnt, ovf);
        if ((rc = xmem_check_launch()) != XMEM_OK) return rc;
    }
    {
        const size_t lds = ((size_t)AFW_BQ * AFF_LDB + 3 * AFW_BQ + 4) * sizeof(float) + (size_t)AFW_BQ * AFW_CAP * sizeof(u64);
        if ((rc = xmem_ensure_dynamic_lds(reinterpret_cast<const void*>(affinity_wide_kernel), lds)) != XMEM_OK) return rc;
        hipLaunchKernelGGL(affinity_wide_kernel, dim3(qt128, w.splits), dim3(512), lds, s, w);
        if ((rc = xmem_check_launch()) != XMEM_OK) return rc;
    }
    // no overflow path: the lists of the wide kernel tighten themselves (see its episodes); the merge reads the global lists
    int heavy_cap = w.splits * (top_k > 32 ? top_k : 32);          // every workgroup hands over at most max(top_k, 32) entries
    if (heavy_cap > AFW_GCAP) heavy_cap = AFW_GCAP;
    hipLaunchKernelGGL(affinity_merge16_kernel, dim3(cdiv(HW, 16)), dim3(256), (size_t)4 * (heavy_cap + 2) * sizeof(u64), s, gcand, gcnt,
                       ovf, a.part_key, a.part_cnt, 0, HW, top_k, heavy_cap, out_w, out_idx, out_sim);
    return xmem_check_launch();
}

extern "C" int xmem_affinity_topk(const xmem_key_segment* segs, int n_seg, const float* qk, const float* qe, int Ck, int HW,
                                  int top_k, float* out_w, int32_t* out_idx, float* out_sim,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    return xmem_affinity_topk_hinted(segs, n_seg, qk, qe, Ck, HW, top_k, nullptr, out_w, out_idx, out_sim, workspace, workspace_bytes, stream);
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

template <bool OUT_HALF>
__global__ void readout_sparse_kernel(ReadoutArgs p) {
    // one workgroup per (query, object): the k row pointers are resolved once, then the rows stream in batches of six
    // independent 16-byte loads per thread (the gather is latency-bound: k * C_v * 4 B = 60 KB per query and object)
    __shared__ const float* rows[AFF_MAX_TOPK];
    __shared__ float wsm[AFF_MAX_TOPK];
    const int q = blockIdx.x, obj = blockIdx.y;
    if ((int)threadIdx.x < p.top_k) {
        int i = p.idx[(size_t)q * p.top_k + threadIdx.x];
        int sg = 0;
        while (sg < p.n_seg - 1 && i >= p.n[sg]) { i -= p.n[sg]; ++sg; }
        rows[threadIdx.x] = p.val[obj * p.n_seg + sg] + (size_t)i * p.Cv;
        wsm[threadIdx.x] = p.w[(size_t)q * p.top_k + threadIdx.x];
    }
    __syncthreads();
    for (int c4 = threadIdx.x; c4 * 4 < p.Cv; c4 += blockDim.x) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 6 <= p.top_k; s += 6) {
            f32x4 v[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) v[u] = *reinterpret_cast<const f32x4*>(rows[s + u] + c4 * 4);
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const float ws = wsm[s + u];
                acc.x += ws * v[u].x; acc.y += ws * v[u].y; acc.z += ws * v[u].z; acc.w += ws * v[u].w;
            }
        }
        for (; s < p.top_k; ++s) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rows[s] + c4 * 4);
            const float ws = wsm[s];
            acc.x += ws * v.x; acc.y += ws * v.y; acc.z += ws * v.z; acc.w += ws * v.w;
        }
        if (OUT_HALF) {                                // fp16 loop: the readout lands in the decoder's half-typed input (strides count halfs)
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            h4 h; h.x = (_Float16)acc.x; h.y = (_Float16)acc.y; h.z = (_Float16)acc.z; h.w = (_Float16)acc.w;
            *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(p.out) + (size_t)obj * p.obj_stride + (size_t)q * p.ldout + c4 * 4) = h;
        } else {
            *reinterpret_cast<f32x4*>(p.out + (size_t)obj * p.obj_stride + (size_t)q * p.ldout + c4 * 4) = acc;
        }
    }
}

extern "C" int xmem_readout_sparse_t(const xmem_value_segment* vsegs, int n_obj, int n_seg, const float* w, const int32_t* idx,
                                     int HW, int top_k, int Cv, void* out_v, int out_half, int ldout, size_t obj_stride, void* stream) {
    float* out = reinterpret_cast<float*>(out_v);
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
        a.out = out_half ? reinterpret_cast<float*>(reinterpret_cast<_Float16*>(out_v) + (size_t)o0 * obj_stride) : out + (size_t)o0 * obj_stride;
        a.ldout = ldout; a.obj_stride = obj_stride;
        int threads = Cv / 4; if (threads > 256) threads = 256; threads = (threads + 63) / 64 * 64;
        if (out_half) hipLaunchKernelGGL(readout_sparse_kernel<true>, dim3(HW, no), dim3(threads), 0, s, a);
        else hipLaunchKernelGGL(readout_sparse_kernel<false>, dim3(HW, no), dim3(threads), 0, s, a);
        int rc = xmem_check_launch();
        if (rc != XMEM_OK) return rc;
    }
    return XMEM_OK;
}
extern "C" int xmem_readout_sparse(const xmem_value_segment* vsegs, int n_obj, int n_seg, const float* w, const int32_t* idx,
                                   int HW, int top_k, int Cv, float* out, int ldout, size_t obj_stride, void* stream) {
    return xmem_readout_sparse_t(vsegs, n_obj, n_seg, w, idx, HW, top_k, Cv, out, 0, ldout, obj_stride, stream);
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
            blo[c] = -e; bhi[c] = 2.f * (k * e); bs = bsq_term(bs, e, k);
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
