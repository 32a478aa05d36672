// Exact top-k readout with an fp16 FILTER pass and an exact fp32 REFINE pass (large memories, when a bound hint exists).
//
// The fp32 select of affinity.hip contracts all N x HW pairs on the fp32 matrix pipe (64 FLOP/clk/SIMD) although only a few
// dozen pairs per query can end up in the top-k.  Here the N x HW contraction runs on v_mfma_f32_32x32x16_f16 (16x the rate)
// with operands ROUNDED to fp16, and a rigorous bound eps on |approximate - exact| decides which pairs could still matter:
//
//   filter:  a(n,q) = fp16 contraction (fp32 accumulate, same formula as memory_util.py:20-37)
//            |a - S| <= eps(n)   with S the exact value, eps(n) = kappa * (||mk_n^2|| Cmax + ||mk_n|| Dmax) * ms_n / 8  (+ small terms)
//            [Cauchy-Schwarz over the 2*C_k products, each off by <= 2^-10 relative after two fp16 roundings; Cmax / Dmax are the
//             largest ||qe||, ||2 qk qe|| of the workgroup's queries; kappa also covers the fp32 accumulation of both passes;
//             fp16 subnormal operands are NOT flushed by the matrix pipe (tools/probes/f16_denorm_probe.hip), their absolute
//             rounding error is the F16_ABS term; operands beyond the fp16 range make eps infinite]
//            tau(q) <= exact k-th similarity (the hint bound)  =>  every element of the exact top-k has a + eps >= tau.
//            One bit per (memory row, query) pair says "a + eps >= tau"; no lists, no atomics, no synchronisation.
//   scan:    the bit matrix (N x HW / 8 bytes, ~0.2 % set) is turned into one index list per query.
//   refine:  the listed candidates (~100 per query) are re-evaluated EXACTLY in fp32 with the same fmaf chain the fp32 MFMA
//            select executes (bit-identical values), ranked, and soft-maxed as the merge kernel does - the outputs are
//            bit-identical to the fp32 path's.  A query whose list overflows (or that has no bound) is scanned in full by
//            its refine wave, exactly.
#include "affinity_common.hpp"
#ifndef SC_DBG
#define SC_DBG 0      // timing experiments only (tools/probes/filter_ab.sh)
#endif

#define F16_BQ 128             // queries per filter workgroup (4 blocks of 32)
#define F16_WAVES 4
#define F16_LDB 272            // bytes per query row of the fp16 operand (256 + 16: odd multiple of 16 B)
#ifndef F16_KAPPA
#define F16_KAPPA 1.07e-3f     // 2^-10 * 1.05 (two fp16 roundings per product) + 4.5e-5 (fp32 accumulation of filter and refine)
#endif
#define F16_ACC 4.5e-5f        // the same accumulation term on |b_sq| (it rides in the accumulator of both chains)
#define F16_ABS 3e-7f          // absolute rounding error of an fp16 SUBNORMAL operand (2^-25), x sqrt(64) via Cauchy-Schwarz

// the exact similarity of ONE (row, query) pair: the fmaf chain of the fp32 MFMA select (affinity_wide_kernel):
// accumulator starts at -b_sq; per 8-channel group t and j = 0..3: k-pairs (8t+j, 8t+4+j) of [x^2 * -e] then of [x * 2ke].
__device__ __forceinline__ float exact_sim(const float* __restrict__ row, const float* ne, const float* ke2, float bs, float msr) {
    float acc = -bs;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const f32x4 xa = *reinterpret_cast<const f32x4*>(row + 8 * t);
        const f32x4 xb = *reinterpret_cast<const f32x4*>(row + 8 * t + 4);
        const f32x4 na = *reinterpret_cast<const f32x4*>(ne + 8 * t), nb = *reinterpret_cast<const f32x4*>(ne + 8 * t + 4);
        const f32x4 ka = *reinterpret_cast<const f32x4*>(ke2 + 8 * t), kb = *reinterpret_cast<const f32x4*>(ke2 + 8 * t + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = xa[j], x1 = xb[j];
            const float xx0 = x0 * x0, xx1 = x1 * x1;
            acc = fmaf(xx0, na[j], acc);
            acc = fmaf(xx1, nb[j], acc);
            acc = fmaf(x0, ka[j], acc);
            acc = fmaf(x1, kb[j], acc);
        }
    }
    return acc * msr;
}

// ============================================================ filter ===================================================
// grid (query tiles of 128, splits of the memory), 4 waves; wave w of split s takes tiles t_begin + w, + 4, ...
// Per 32-row tile and 32-query block: 8 x v_mfma_f32_32x32x16_f16, then 16 compares whose lane masks ARE the output words.
__global__ __launch_bounds__(256, 2) void affinity_filter16_kernel(Filter16Args p) {
    constexpr int CK = 64;
    __shared__ __attribute__((aligned(16))) unsigned char Bh[F16_BQ * F16_LDB];   // fp16 (-e | 2ke) per query
    __shared__ float s_bs[F16_BQ], s_tau[F16_BQ];
    __shared__ unsigned s_qmax[3];                                                  // max ||qe||, ||2ke||, |b_sq| (bits of non-negative floats)
    __shared__ __attribute__((aligned(16))) float s_row[F16_WAVES][AFF_ROWS][2];    // per wave: (ms / 8, eps) of the tile's rows

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.x * F16_BQ;
    const int split = blockIdx.y;

    if (tid < 3) s_qmax[tid] = 0u;
    __syncthreads();
    // query operand: 128 rows of 256 B prepared once per call by the bound kernel (coalesced 16-B copies)
    for (int e = tid; e < F16_BQ * 16; e += 256) {
        const int q = e >> 4, part = e & 15, qg = q0 + q;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (qg < p.HW) v = *reinterpret_cast<const uint4*>(p.qop16 + (size_t)qg * 2 * CK + part * 8);
        *reinterpret_cast<uint4*>(Bh + q * F16_LDB + part * 16) = v;
    }
    if (tid < F16_BQ) {
        const int qg = q0 + tid;
        float t0 = INFINITY, bs = 0.f;
        if (qg < p.HW) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(p.qmeta + (size_t)qg * 4);
            bs = m[0];
            atomicMax(&s_qmax[0], __float_as_uint(m[1]));
            atomicMax(&s_qmax[1], __float_as_uint(m[2]));
            atomicMax(&s_qmax[2], __float_as_uint(fabsf(bs)));
            t0 = p.tau_init[qg];
            if (t0 == -INFINITY) {                                   // no bound: the refine scans this query in full
                if (split == 0) p.gcnt[qg] = AFW_GCAP + 1;
                t0 = INFINITY;
            }
        }
        s_bs[tid] = bs; s_tau[tid] = t0;
    }
    __syncthreads();
    float Cmax = __uint_as_float(s_qmax[0]), Dmax = __uint_as_float(s_qmax[1]);
    const float bsmax = __uint_as_float(s_qmax[2]);
    // an operand beyond the fp16 range (|v| <= ||v||) voids the bound: eps = inf keeps every pair for the exact pass
    if (!(Cmax < 6.5e4f)) Cmax = INFINITY;
    if (!(Dmax < 6.5e4f)) Dmax = INFINITY;

    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(p.total_tiles, t_begin + p.tiles_per_split);

    // next tile's key rows (lane: row l31, channels [16t + 8 lh, +8) for t = 0..3), row index clamped into the segment
    // (the duplicated rows of a segment's last tile set spurious bits; the scan drops rows past the segment's end)
    f32x4 an[8]; float msn = 1.f;
    auto issue_loads = [&](int tile) {
        if (tile >= t_end) return;
        int sg = 0;
#pragma unroll
        for (int i = 1; i < XMEM_MAX_SEGMENTS; ++i)
            if (i < p.n_seg && tile >= p.seg[i].tile0) sg = i;
        const int r = min((tile - p.seg[sg].tile0) * AFF_ROWS + l31, p.seg[sg].n - 1);
        const float* src = p.seg[sg].key + (size_t)r * CK + lh * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            an[2 * t] = *reinterpret_cast<const f32x4*>(src + 16 * t);
            an[2 * t + 1] = *reinterpret_cast<const f32x4*>(src + 16 * t + 4);
        }
        msn = p.seg[sg].shr ? p.seg[sg].shr[r] : 1.f;
    };

    float my_tau[4], my_bs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { my_bs[i] = s_bs[i * 32 + l31]; my_tau[i] = s_tau[i * 32 + l31]; }
    const unsigned char* bq = Bh + l31 * F16_LDB + lh * 16;
    float* rowinfo = &s_row[wave][0][0];
    const size_t blk0 = (size_t)blockIdx.x * 4;

    issue_loads(t_begin + wave);
    for (int tile = t_begin + wave; tile < t_end; tile += F16_WAVES) {
        // fp16 operands of this tile's rows: x (k >= 64) and x^2 (k < 64); row norms for eps
        h16x8 xh[4], x2h[4];
        float sA = 0.f, sB = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xa = an[2 * t][j], xb = an[2 * t + 1][j];
                const float qa = xa * xa, qb = xb * xb;
                xh[t][j] = (_Float16)xa; xh[t][4 + j] = (_Float16)xb;
                x2h[t][j] = (_Float16)qa; x2h[t][4 + j] = (_Float16)qb;
                sB += qa + qb; sA += qa * qa + qb * qb;
            }
        }
        const float msr_mine = msn * 0.125f;
        issue_loads(tile + F16_WAVES);                          // next tile's rows in flight under this tile's work
        sA += __shfl_xor(sA, 32, 64); sB += __shfl_xor(sB, 32, 64);
        // |a - S| <= eps for every query of this workgroup (header)
        const float An = sqrtf(sA), Bn = sqrtf(sB);
        float eps_mine = ((An * Cmax + Bn * Dmax) * F16_KAPPA + F16_ABS * (An + Bn + Cmax + Dmax) + F16_ACC * bsmax)
                         * fabsf(msr_mine) * 1.0001f;
        // x^2 <= sqrt(sum x^4) must stay inside the fp16 range (then |x| does too); NaN / inf rows and a NaN bound land here as well
        if (!(sA < 4.0e9f) || !(eps_mine < INFINITY)) eps_mine = INFINITY;
        if (lh == 0) { rowinfo[2 * l31] = msr_mine; rowinfo[2 * l31 + 1] = eps_mine; }

        f32x16 c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float nb = -my_bs[i];
            asm volatile("" : "+v"(nb));
#pragma unroll
            for (int r = 0; r < 16; ++r) c[i][r] = nb;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const h16x8 blo = *reinterpret_cast<const h16x8*>(bq + i * 32 * F16_LDB + t * 32);
                c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x2h[t], blo, c[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const h16x8 bhi = *reinterpret_cast<const h16x8*>(bq + i * 32 * F16_LDB + 128 + t * 32);
                c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[t], bhi, c[i], 0, 0, 0);
            }
        }
        // (ms / 8, eps) of accumulator register r's row: rows 8g + 4 lh + {0..3} for g = r >> 2 (same-wave LDS ops are in order)
        float msr[16], eps[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(rowinfo + 2 * (8 * g + 4 * lh));
            const f32x4 u1 = *reinterpret_cast<const f32x4*>(rowinfo + 2 * (8 * g + 4 * lh) + 4);
            msr[4 * g] = u0[0]; eps[4 * g] = u0[1]; msr[4 * g + 1] = u0[2]; eps[4 * g + 1] = u0[3];
            msr[4 * g + 2] = u1[0]; eps[4 * g + 2] = u1[1]; msr[4 * g + 3] = u1[2]; eps[4 * g + 3] = u1[3];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int mlo = 0, mhi = 0;                                   // lane r < 16 collects word r (v_writelane: 2 instructions per word)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // NaN-safe "upper estimate >= tau": a non-finite estimate keeps the pair for the exact pass
                const u64 m = __ballot(!(fmaf(c[i][r], msr[r], eps[r]) < my_tau[i]));
                // gfx950: a VALU write of an SGPR needs 2 wait states before a VALU read of it; the compiler cannot see into the asm
                asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
                    : "+v"(mlo), "+v"(mhi) : "s"((int)(unsigned)m), "s"((int)(unsigned)(m >> 32)), "n"(r));
            }
            if (lane < 16)
                p.mask[((blk0 + i) * (size_t)p.total_tiles + tile) * 16 + lane] = ((u64)(unsigned)mhi << 32) | (u64)(unsigned)mlo;
        }
    }
}

// ============================================================ scan =====================================================
// bit (word r, lane j) of tile t, query block b  <->  query 32 b + (j & 31), row 32 (t - tile0) + (r & 3) + 8 (r >> 2) + 4 (j >> 5)
#define SCAN_TILES 128
#define SCAN_CAP 192
__global__ __launch_bounds__(256) void affinity_scan_kernel(Filter16Args p) {
    __shared__ int s_cnt[32], s_base[32];
    __shared__ int s_buf[32][SCAN_CAP];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int t0 = blockIdx.y * SCAN_TILES;
    const int nt = min(SCAN_TILES, p.total_tiles - t0);
    if (tid < 32) s_cnt[tid] = 0;
    __syncthreads();
    const u64* words = p.mask + ((size_t)b * p.total_tiles + t0) * 16;
    for (int w0 = tid; w0 < nt * 16; w0 += 4 * 256) {
        u64 m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) m[u] = (w0 + 256 * u < nt * 16) ? words[w0 + 256 * u] : 0ull;
#if SC_DBG == 1
        if ((m[0] ^ m[1] ^ m[2] ^ m[3]) == 0x123456789ull) p.gcnt[0] = 1;
        continue;
#endif
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int w = w0 + 256 * u;
            u64 mm = m[u];
            if (!mm) continue;
            const int tile = t0 + (w >> 4), r = w & 15;
            const SegDev sd = seg_of_tile(p, tile);
            const int rbase = (tile - sd.tile0) * AFF_ROWS + (r & 3) + 8 * (r >> 2);
            const int segn = sd.n, gbase = sd.base;
            while (mm) {
                const int j = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                const int row = rbase + 4 * (j >> 5);
                if (row >= segn) continue;                       // clamped duplicate of the segment's last row
                const int qi = j & 31, gi = gbase + row;
                const int slot = atomicAdd(&s_cnt[qi], 1);
                if (slot < SCAN_CAP) s_buf[qi][slot] = gi;
                else {                                           // local buffer full: straight to the query's global list
                    const int qg = b * 32 + qi;
                    const int gs = atomicAdd(&p.gcnt[qg], 1);
                    if (gs < AFW_GCAP) p.gcand32[(size_t)qg * AFW_GCAP + gs] = gi;
                }
            }
        }
    }
    __syncthreads();
#if SC_DBG == 2
    return;
#endif
    if (tid < 32) {
        const int n = min(s_cnt[tid], SCAN_CAP), qg = b * 32 + tid;
        s_base[tid] = (n > 0 && qg < p.HW) ? atomicAdd(&p.gcnt[qg], n) : 0;
    }
    __syncthreads();
    for (int e = tid; e < 32 * SCAN_CAP; e += 256) {
        const int qi = e / SCAN_CAP, j = e - qi * SCAN_CAP, qg = b * 32 + qi;
        if (qg < p.HW && j < min(s_cnt[qi], SCAN_CAP) && s_base[qi] + j < AFW_GCAP)
            p.gcand32[(size_t)qg * AFW_GCAP + s_base[qi] + j] = s_buf[qi][j];
    }
}

// ============================================================ refine ===================================================
// One workgroup (4 waves) per query: exact similarities of its candidates (or of ALL memory elements when the list overflowed
// or no bound existed), 64 per wave and round, each wave keeping a running list of its best by counting ranks; wave 0 merges
// the four lists and applies the softmax exactly as affinity_merge16_kernel computes it.
#define RF_BUF 256             // running list of a wave: compacted to the best top_k whenever another 64 might not fit
__global__ __launch_bounds__(256) void affinity_refine_kernel(Filter16Args p) {
    constexpr int CK = 64;
    __shared__ __attribute__((aligned(16))) float s_op[2 * CK];
    __shared__ __attribute__((aligned(16))) u64 s_keys[4][RF_BUF + 2];
    __shared__ int s_n[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = blockIdx.x;
    float* ne = s_op; float* ke2 = s_op + CK; u64* keys = s_keys[wv];
    if (wv == 0) {
        const float k = p.qk[(size_t)q * CK + lane];
        const float e = p.qe ? p.qe[(size_t)q * CK + lane] : 1.f;
        ne[lane] = -e; ke2[lane] = 2.f * (k * e);
    }
    const float bs = p.qmeta[(size_t)q * 4];          // b_sq with the select kernels' arithmetic (bound kernel)
    const int T = p.gcnt[q];
    const bool full = T > AFW_GCAP;
    int total = T;
    if (full) { total = 0; for (int i = 0; i < p.n_seg; ++i) total += p.seg[i].n; }
    const int* list = p.gcand32 + (size_t)q * AFW_GCAP;
    int gi_next = 0;
    if (wv * 64 + lane < total) gi_next = full ? wv * 64 + lane : list[wv * 64 + lane];
    __syncthreads();

    // The best min(n, keepn) of keys[0..n) move to the front in descending order; returns the new length.  n <= RF_BUF, keepn <= 64.
    // Selection: the keepn-th largest key by bitwise descent (value half first, the index half only among exact ties), then
    // the survivors are ranked by counting (keys are unique).
    auto compact = [&](int n, int keepn) -> int {
        __builtin_amdgcn_wave_barrier();
        if (n > keepn) {
            const int per = (n + 63) >> 6;                             // uniform
            unsigned hi[4], lo[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = lane + 64 * u;
                const u64 k = e < n ? keys[e] : 0ull;                  // valid keys have a non-zero value half
                hi[u] = (unsigned)(k >> 32); lo[u] = (unsigned)k;
            }
            unsigned pre = 0u;
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned cand = pre | (1u << bit);
                int c = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u < per) c += __popcll(__ballot(hi[u] >= cand));
                if (c >= keepn) pre = cand;
            }
            int cg = 0, ce = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < per) { cg += __popcll(__ballot(hi[u] > pre)); ce += __popcll(__ballot(hi[u] == pre)); }
            const int need = keepn - cg;                                // >= 1 of the ce elements tied at the threshold value
            unsigned lpre = 0u;
            if (ce > need) {                                            // exact ties: the larger index half (= lower memory index) wins
                for (int bit = 31; bit >= 0; --bit) {
                    const unsigned cand = lpre | (1u << bit);
                    int c = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (u < per) c += __popcll(__ballot(hi[u] == pre && lo[u] >= cand));
                    if (c >= need) lpre = cand;
                }
            }
            int pos = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u < per) {
                    const bool kp = hi[u] > pre || (hi[u] == pre && lo[u] >= lpre);
                    const unsigned long long m = __ballot(kp);
                    if (kp) keys[pos + __popcll(m & ((1ull << lane) - 1ull))] = ((u64)hi[u] << 32) | lo[u];
                    pos += __popcll(m);
                }
            }
            n = pos;                                                    // == keepn
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) { keys[n] = 0ull; keys[n + 1] = 0ull; }
        __builtin_amdgcn_wave_barrier();
        const u64 mine = lane < n ? keys[lane] : ~0ull;
        int rk = 0;
#pragma unroll 4
        for (int f = 0; f < n; f += 2) {
            const ulonglong2 kf = *reinterpret_cast<const ulonglong2*>(keys + f);
            rk += (int)(kf.x > mine) + (int)(kf.y > mine);             // the padding keys are 0: never greater
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < n) keys[rk] = mine;
        __builtin_amdgcn_wave_barrier();
        return n;
    };

    int cnt = 0;
    float thr = -INFINITY;                            // raised to the wave's k-th best once it knows k candidates
    for (int b0 = wv * 64; b0 < total; b0 += 256) {
        const int e = b0 + lane;
        const int gi = gi_next;
        if (e + 256 < total) gi_next = full ? e + 256 : list[e + 256];       // next round's index in flight under this round
        bool pass = false; float s = 0.f;
        if (e < total) {
            const SegDev sd = seg_of_row(p, gi);
            const int o = gi - sd.base;
            const float msr = (sd.shr ? sd.shr[o] : 1.f) * 0.125f;
            s = exact_sim(sd.key + (size_t)o * CK, ne, ke2, bs, msr);
            pass = s >= thr;                          // NaN never enters (as in the fp32 select)
        }
        const unsigned long long m = __ballot(pass);
        if (pass) keys[cnt + __popcll(m & ((1ull << lane) - 1ull))] = pack_key(s, gi);
        cnt += __popcll(m);
        if (cnt > RF_BUF - 64) {
            cnt = compact(cnt, p.top_k);
            if (cnt >= p.top_k) thr = key_val(keys[p.top_k - 1]);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (total > 64) {                                 // more than one wave had work: every wave hands over its best top_k
        cnt = compact(cnt, p.top_k);
        if (lane == 0) s_n[wv] = cnt;
        __syncthreads();
        if (wv != 0) return;
        for (int w = 1; w < 4; ++w) {
            const int nw = s_n[w];
            if (lane < nw) keys[cnt + lane] = s_keys[w][lane];
            cnt += nw;
        }
        __builtin_amdgcn_wave_barrier();
    } else if (wv != 0) return;
    cnt = compact(cnt, p.top_k);
    // softmax without max shift (memory_util.py:48-49), summed exactly as affinity_merge16_kernel: 16 lanes, r = l, l+16, ...
    if (lane < 16) {
        float sum = 0.f;
        for (int r = lane; r < p.top_k; r += 16) sum += expf(r < cnt ? key_val(keys[r]) : -INFINITY);
        sum += __shfl_xor(sum, 8, 16); sum += __shfl_xor(sum, 4, 16); sum += __shfl_xor(sum, 2, 16); sum += __shfl_xor(sum, 1, 16);
        for (int r = lane; r < p.top_k; r += 16) {
            const float v = r < cnt ? key_val(keys[r]) : -INFINITY;
            p.out_w[(size_t)q * p.top_k + r] = expf(v) / sum;
            p.out_idx[(size_t)q * p.top_k + r] = r < cnt ? key_idx(keys[r]) : 0;
            if (p.out_sim) p.out_sim[(size_t)q * p.top_k + r] = v;
        }
    }
}

size_t aff_filter16_mask_bytes(int n_total, int HW) {
    const size_t tiles = (size_t)cdiv(n_total, AFF_ROWS) + XMEM_MAX_SEGMENTS;
    return (size_t)cdiv(HW, F16_BQ) * 4 * tiles * 16 * sizeof(u64);
}

int aff_filter16_launch(Filter16Args a, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int qt = cdiv(a.HW, F16_BQ);
    // two 4-wave workgroups per CU: splits so that query tiles x splits ~ 512, >= 4 tiles per wave
    int sp = 512 / qt; if (sp < 1) sp = 1;
    { int maxs = a.total_tiles / (4 * F16_WAVES); if (maxs < 1) maxs = 1; if (sp > maxs) sp = maxs; }
    a.tiles_per_split = cdiv(a.total_tiles, sp);
    a.splits = cdiv(a.total_tiles, a.tiles_per_split);
    hipLaunchKernelGGL(affinity_filter16_kernel, dim3(qt, a.splits), dim3(256), 0, s, a);
    int rc = xmem_check_launch();
    if (rc != XMEM_OK) return rc;
    hipLaunchKernelGGL(affinity_scan_kernel, dim3(qt * 4, cdiv(a.total_tiles, SCAN_TILES)), dim3(256), 0, s, a);
    if ((rc = xmem_check_launch()) != XMEM_OK) return rc;
    hipLaunchKernelGGL(affinity_refine_kernel, dim3(a.HW), dim3(256), 0, s, a);
    return xmem_check_launch();
}
