// Exact top-k readout with an fp16 FILTER pass and an exact fp32 REFINE pass (memories of >= 256 tiles, when a bound hint exists).
//
// The fp32 select of affinity.hip contracts all N x HW pairs on the fp32 matrix pipe (64 FLOP/clk/SIMD) although only a few
// dozen pairs per query can end up in the top-k.  Here the N x HW contraction runs on v_mfma_f32_32x32x16_f16 (16x the rate)
// with operands ROUNDED to fp16, and a rigorous bound eps on |approximate - exact| decides which pairs could still matter.
// With S(n,q) the exact fp32 similarity (memory_util.py:20-37 as the fp32 MFMA select evaluates it) and a(n,q) its fp16 version:
//
//   |a - S| <= eps(n,q) = [ KAPPA (||x_n^2|| ||e_q|| + ||x_n|| ||2 k_q e_q||) + ACC |b_sq(q)| + ABS (norms) ] * |ms_n| / 8
//       Cauchy-Schwarz over the 2 C_k products, each off by <= 2^-10 (1 + 2^-12) relative after two fp16 roundings (KAPPA adds
//       the fp32 accumulation of both evaluations, ACC the same on b_sq which rides in the accumulator); fp16 SUBNORMAL operands
//       are not flushed by the matrix pipe (tools/probes/f16_denorm_probe.hip), their absolute rounding error is the ABS term;
//       an operand beyond the fp16 range, or NaN, makes eps infinite.
//
// rows:    one fp16 operand row per memory element, [ms/8 x^2 | ms/8 x | 16 augmentation terms] (affinity_common.hpp); the
//          query rows [-e | 2ke | ...] come from the bound kernel.  The augmentation terms make the fp32-accumulated dot product
//          of a memory row and a query row equal to  a(n,q) + eps(n,q)  DIRECTLY (b_sq split hi/lo, eps as products of row and
//          query factors): the contraction yields an upper estimate of S, the epilogue is one compare.
// filter:  tau(q) <= exact k-th similarity (the hint bound)  =>  every element of the exact top-k has a + eps >= tau.
//          One bit per (memory row, query) pair says so.  The bits never leave the chip (round 6; rounds 2-5 stored an N x HW / 8
//          byte bit matrix and a scan kernel turned it into lists): a lane's NON-ZERO halfword of 16 pair bits (~3 % of them at
//          the ~0.2 % pair density of a video) is pushed into a small per-WAVE LDS buffer - ballot + popcount + one ds_write, no
//          atomics, nothing waited for - and the buffer is drained into the per-query index lists with ONE global atomic per
//          halfword when it fills up and at the wave's end.  A list that overflows (no usable bound: scene cut, garbage hint,
//          hostile ties) flags its 128-query tile.
// refine:  the listed candidates (~100 per query) are re-evaluated EXACTLY in fp32 with the fmaf chain the fp32 MFMA select
//          executes (bit-identical values), ranked, and soft-maxed as the merge kernel does: outputs bit-identical to the fp32
//          path's.  Flagged tiles are computed by the fp32 select + merge of the same launch (affinity.hip), not here.
#include "affinity_common.hpp"
#include <stdlib.h>

#define F16_BQ 128             // queries per FLAG tile (flag1 / flag2 / `only`: 4 blocks of 32)
// filter workgroup of NW waves (4 | 8): 64 NW queries (2 blocks of 32 per wave), LDS stages of NW 32-row tiles
// (2 stages x NW x 32 x 304 B = 76 | 152 KB; 9 pieces of 16 bytes per thread and stage either way)
#ifndef F16_WG_PER_CU
#define F16_WG_PER_CU 2       // filter workgroups (4 waves) per CU the register budget is set for
#endif
#ifndef RF_WAVES
#define RF_WAVES 4           // waves of a refine workgroup (one query)
#endif
#ifndef F16_PF
#define F16_PF 2               // k-steps the row fragments are requested from LDS ahead of their MFMAs
#endif
#define F16_LDB 304            // bytes per memory operand row in LDS (288 + 16: odd multiple of 16 B, conflict-free 16-byte fragment reads)
#define F16_WB 252             // entries of a wave's candidate buffer: the 16 PAD bytes of 63 of the staged operand rows (no LDS of its own:
                               // 76 KB per 4-wave workgroup as before - with 4 KB more the second workgroup of a CU no longer fits: 61 us instead of 35)
#define F16_DU 4                // entries per lane and drain trip (F16_WB / 64: a full buffer goes in one trip)
#define F16_WB_TSPAN 500       // an entry carries its tile relative to the buffer's base tile in 9 bits: drained at least every 500 tiles

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

// ============================================================ rows ======================================================
// fp16 operand rows of the memory (layout: affinity_common.hpp), 16 lanes per row.  Depends on (key, shrinkage) only.
__global__ __launch_bounds__(256) void affinity_rows16_kernel(const float* __restrict__ key, const float* __restrict__ shr, int n,
                                                             _Float16* __restrict__ rows16) {
    constexpr int CK = 64;
    const int l = threadIdx.x & 15;
    const int o = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (o >= n) return;
    const f32x4 x = *reinterpret_cast<const f32x4*>(key + (size_t)o * CK + 4 * l);
    const float msr = (shr ? shr[o] : 1.f) * 0.125f;
    float sA = 0.f, sB = 0.f, mx = fabsf(msr);
    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
    h16x4 h2, h1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float q = x[j] * x[j];
        const float a2 = msr * q, a1 = msr * x[j];
        sA += q * q; sB += q;
        mx = fmaxf(mx, fmaxf(fabsf(a2), fabsf(a1)));
        h2[j] = (_Float16)a2; h1[j] = (_Float16)a1;
    }
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) {
        sA += __shfl_xor(sA, d, 16); sB += __shfl_xor(sB, d, 16); mx = fmaxf(mx, __shfl_xor(mx, d, 16));
    }
    _Float16* orow = rows16 + (size_t)o * F16_K;
    *reinterpret_cast<h16x4*>(orow + 4 * l) = h2;
    *reinterpret_cast<h16x4*>(orow + CK + 4 * l) = h1;
    const float An = sqrtf(sA) * 1.0001f, Bn = sqrtf(sB) * 1.0001f, am = fabsf(msr);
    const _Float16 mh = (_Float16)msr;
    _Float16 v = (_Float16)0.f;
    if (l == 0 || l == 1) v = mh;
    else if (l == 2) v = (_Float16)(msr - (float)mh);
    else if (l == 3) v = f16_up(F16_KAPPA * An * am);
    else if (l == 4) v = f16_up(F16_KAPPA * Bn * am);
    else if (l == 5) v = f16_up(am);
    else if (l == 6) {
        // every operand of the row must be inside the fp16 range (NaN fails the test too); otherwise no bound: keep every pair
        const float z = F16_ABS * (An + Bn) * am * 1024.f;
        v = (mx < 6.5e4f && z < 6.5e4f) ? f16_up(z) : (_Float16)INFINITY;
    }
    orow[2 * CK + l] = v;
}

// ============================================================ filter ===================================================
// C[rows x queries] = rows16 . qop16^T on v_mfma_f32_32x32x16_f16, K = 144; the output is ONE BIT per pair (estimate >= tau).
// Workgroup = NW waves (4 | 8), 64 NW queries x one split of the memory's 32-row tiles:
//   queries  wave w owns query blocks 2w, 2w + 1 (32 queries each); their operand rows stay in REGISTERS for the whole
//            workgroup (2 x 9 fragments = 72 VGPRs, read from qop16 once)
//   rows     streamed through LDS in stages of NW tiles by all threads: consecutive lanes copy consecutive 16-byte pieces of
//            consecutive 288-byte rows - whole cache lines, 9 pieces per thread and stage, one address add per piece on the
//            regular stage (one segment, no clamped row); double buffered, one barrier per stage; every wave reads every
//            tile's fragments from LDS (ds_read_b128, row stride 304 B: conflict-free), two k-steps ahead of their MFMAs
//   compares software-pipelined inside the wave: two accumulator sets of one tile x 2 query blocks; behind each of the 18 MFMAs
//            of tile j + 1 (32 cycles each in the pipe) two compares of tile j against tau - accumulator array a under MFMAs
//            1 + 8a .. 8 + 8a, then its 64 halfwords of the bit matrix are stored.  A compare (v_cmp_nlt: NaN-safe "upper
//            estimate >= tau", true for a non-finite estimate - the exact pass decides) writes its lane mask to an SGPR pair
//            of its own; the add-with-carry that shifts the bit into the lane's halfword runs one MFMA slot later (r = 0 ends
//            up in bit 15).  The results of a tile's last MFMA are first read after three later MFMAs have been issued behind
//            it (in-order pipe): that covers the MFMA -> VALU read hazard the compiler cannot see through asm statements.
// Everything stays in 228 VGPRs, two waves per SIMD (a v_accvgpr_read beside MFMAs costs 8 cycles, a lone wave hides at most
// ~5 single-issue instructions per MFMA: tools/probes/mfma_shadow).  Round 3 ran each pair of tiles' 36 MFMAs and then its 64
// compares as a block, leaving the overlap to the two waves of a SIMD: 35.9 us / 884 us / 8.26 ms at the three served sizes
// (51 840 x 1 620, 921 600 x 3 600, 4 177 920 x 8 160); this schedule: 33.5 us / 823 us / 7.91 ms = 0.30 / 0.48 / 0.50 of the
// nominal fp16 peak.  The bound is the chip's power budget, not the schedule: a register-only loop of the same MFMA on random
// operands sustains 0.60 - 0.66 of the nominal peak (0.90 - 0.95 on zeros; tools/probes/mfma_shadow/mfma_sustained.hip), and
// this kernel with its row fetch, LDS hand-over and bit stores knocked out runs at 0.61 - 0.63
// (profiles/r04_filter_kernel_ab.txt).  (Round 2 kept 128 queries in LDS and loaded the rows as MFMA fragments straight from
// global memory: 972 us / 9.56 ms.)
// 1-D grid, XCD-aware: workgroup L runs on XCD L % 8 (round-robin dispatch); all query tiles of one split - the workgroups
// that stream the SAME rows - share L % 8 and neighbouring slots.
// PASS2 = the second pass over the flagged 128-query tiles (its own kernel name: a trace tells the working launch from the one
// that normally returns at once).  DBG: knock-outs for tools builds (-DXMEM_TOOLS), 0 in the shipped instantiations.
template <bool PASS2, int NW, int DBG = 0>
__global__ __launch_bounds__(64 * NW, NW == 4 ? F16_WG_PER_CU : 1) void affinity_filter16_kernel(Filter16Args p) {
    constexpr int NQB = 2, F16_STAGE = NW, F16_WGQ = 64 * NW, NTHR = 64 * NW;
    __shared__ __attribute__((aligned(16))) unsigned char Ah[2][F16_STAGE * AFF_ROWS * F16_LDB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int slot = blockIdx.x >> 3;
    const int qtile = slot % p.qtiles;
    const int split = (blockIdx.x & 7) + 8 * (slot / p.qtiles);
    if (split >= p.splits) return;
    const int b0 = qtile * (F16_WGQ / 32) + wave * NQB;          // this wave's first query block
    bool active = b0 * 32 < p.HW;                                // (wave-uniform)
    if (PASS2) {                                                 // pass 2: flagged 128-query tiles only
        const int nflag = (p.HW + F16_BQ - 1) / F16_BQ, f = (F16_WGQ / F16_BQ) * qtile;
        bool any = false;
#pragma unroll
        for (int i = 0; i < F16_WGQ / F16_BQ; ++i) any = any || (f + i < nflag && p.only[f + i] != 0);
        if (!any) return;
        if (active) active = p.only[b0 >> 2] != 0;               // (a wave's two blocks lie in one flag tile)
    }

    h16x8 bq[NQB][9];
    float my_tau[NQB];
    unsigned qopen[NQB];                                         // all ones, or 0 once this lane's query of block i takes no (more) candidates
#pragma unroll
    for (int i = 0; i < NQB; ++i) {
        const int q = (b0 + i) * 32 + l31;
        const bool ok = active && q < p.HW;
        qopen[i] = ok ? 0xffffffffu : 0u;                        // (a padding query's estimate is NaN beside a non-finite row: never listed)
        const _Float16* src = p.qop16 + (size_t)min(q, p.HW - 1) * F16_K + lh * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            h16x8 v = *reinterpret_cast<const h16x8*>(src + 16 * t);
            if (!ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.f;
            }
            bq[i][t] = v;
        }
        // a query without a bound (-inf) keeps every pair: its list fills up, flags the tile, and the tighten pass makes a bound;
        // padding queries (+inf) keep none
        my_tau[i] = ok ? p.tau[q] : INFINITY;
    }

    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(p.total_tiles, t_begin + p.tiles_per_split);
    const int nst = (t_end - t_begin + F16_STAGE - 1) / F16_STAGE;

    // one stage = F16_STAGE tiles x 32 rows x 18 pieces of 16 bytes = 9 pieces per thread.  Row index clamped into the segment
    // (the duplicated rows of a segment's last tile set spurious bits; the scan drops rows past the segment's end); the segment
    // table is copied to scalars once; tiles past the split's end repeat its last tile - branch-free loads, their bits are not stored.
    // FAST stage (the rule; decided per stage with scalar arithmetic): its tiles lie in one segment, none is the segment's clamped
    // last tile and none is past the split - the stage is F16_STAGE x 9216 contiguous bytes, piece e at byte 16 e
    static_assert(XMEM_MAX_SEGMENTS == 4, "segment select below is written out for 4 segments");
    const int gt1 = 1 < p.n_seg ? p.seg[1].tile0 : 0x7fffffff, gt2 = 2 < p.n_seg ? p.seg[2].tile0 : 0x7fffffff,
              gt3 = 3 < p.n_seg ? p.seg[3].tile0 : 0x7fffffff;
    const int gn0 = p.seg[0].n, gn1 = 1 < p.n_seg ? p.seg[1].n : 1, gn2 = 2 < p.n_seg ? p.seg[2].n : 1, gn3 = 3 < p.n_seg ? p.seg[3].n : 1;
    const _Float16 *gr0 = p.seg[0].rows16, *gr1 = 1 < p.n_seg ? p.seg[1].rows16 : gr0, *gr2 = 2 < p.n_seg ? p.seg[2].rows16 : gr0,
                   *gr3 = 3 < p.n_seg ? p.seg[3].rows16 : gr0;
    const int gt0 = p.seg[0].tile0;
    uint4 st0, st1, st2, st3, st4, st5, st6, st7, st8;
    const unsigned char* fbase = nullptr;
    bool ffast = false;
#define F16_STAGE_SETUP(S)                                                                                           \
    {                                                                                                                \
        const int T0 = t_begin + (S) * F16_STAGE, Tl = T0 + F16_STAGE - 1;                                           \
        const bool s1 = T0 >= gt1, s2 = T0 >= gt2, s3 = T0 >= gt3;                                                   \
        const _Float16* base = s3 ? gr3 : s2 ? gr2 : s1 ? gr1 : gr0;                                                 \
        const int tile0 = s3 ? gt3 : s2 ? gt2 : s1 ? gt1 : gt0, n = s3 ? gn3 : s2 ? gn2 : s1 ? gn1 : gn0;            \
        const int next0 = s3 ? 0x7fffffff : s2 ? gt3 : s1 ? gt2 : gt1;                                               \
        ffast = Tl < t_end && Tl < next0 && (Tl - tile0 + 1) * AFF_ROWS <= n;                                        \
        fbase = reinterpret_cast<const unsigned char*>(base) + (size_t)(T0 - tile0) * (AFF_ROWS * F16_K * 2);        \
    }
#define F16_FETCH_ONE(J, DST)                                                                                        \
    if (ffast) {                                                                                                     \
        DST = *reinterpret_cast<const uint4*>(fbase + (size_t)(tido + NTHR * (J)) * 16);                             \
    } else {                                                                                                         \
        const int e = tido + NTHR * (J), rr = e / 18, part = e - rr * 18;                                              \
        const int tile = min(t_begin + fs * F16_STAGE + (rr >> 5), t_end - 1);                                       \
        const bool s1 = tile >= gt1, s2 = tile >= gt2, s3 = tile >= gt3;                                             \
        const _Float16* base = s3 ? gr3 : s2 ? gr2 : s1 ? gr1 : gr0;                                                 \
        const int tile0 = s3 ? gt3 : s2 ? gt2 : s1 ? gt1 : gt0, n = s3 ? gn3 : s2 ? gn2 : s1 ? gn1 : gn0;            \
        const int r = min((tile - tile0) * AFF_ROWS + (rr & 31), n - 1);                                             \
        DST = *reinterpret_cast<const uint4*>(base + (size_t)r * F16_K + part * 8);                                  \
    }
#define F16_FETCH(S)                                                                                                 \
    {                                                                                                                \
        const int fs = (S);                                                                                          \
        F16_STAGE_SETUP(fs)                                                                                          \
        F16_FETCH_ONE(0, st0) F16_FETCH_ONE(1, st1) F16_FETCH_ONE(2, st2) F16_FETCH_ONE(3, st3) F16_FETCH_ONE(4, st4) \
        F16_FETCH_ONE(5, st5) F16_FETCH_ONE(6, st6) F16_FETCH_ONE(7, st7) F16_FETCH_ONE(8, st8)                      \
    }
#define F16_STASH_ONE(J, SRC)                                                                                        \
    {                                                                                                                \
        const int e = tido + NTHR * (J), rr = e / 18, part = e - rr * 18;                                              \
        *reinterpret_cast<uint4*>(sb + rr * F16_LDB + part * 16) = SRC;                                                 \
    }
#define F16_STASH(B)                                                                                                 \
    {                                                                                                                \
        unsigned char* sb = (B);                                                                                       \
        F16_STASH_ONE(0, st0) F16_STASH_ONE(1, st1) F16_STASH_ONE(2, st2) F16_STASH_ONE(3, st3) F16_STASH_ONE(4, st4) \
        F16_STASH_ONE(5, st5) F16_STASH_ONE(6, st6) F16_STASH_ONE(7, st7) F16_STASH_ONE(8, st8)                      \
    }

    f32x16 cA[NQB], cB[NQB];                                     // the two accumulator sets: one TILE x NQB query blocks each
#pragma unroll
    for (int i = 0; i < NQB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { cA[i][r] = 0.f; cB[i][r] = 0.f; }
    unsigned bits[NQB];
#pragma unroll
    for (int a = 0; a < NQB; ++a) bits[a] = 0u;
    h16x8 fr[9];
    // ---- candidates: bits -> per-wave LDS buffer -> per-query global lists --------------------------------------------------
    // entry = halfword of 16 pair bits | filter lane << 16 | query block << 22 | (tile - tbase) << 23.  The buffer belongs to ONE wave:
    // its fill count is a scalar, a push is a ballot, a prefix count and an LDS write that nothing waits for.  drain(): one lane per
    // entry, ONE atomic reserves the halfword's slots in its query's list (bit 15 - r of lane l <-> row 32 (t - tile0) + (r & 3) +
    // 8 (r >> 2) + 4 (l >> 5), query 32 b + (l & 31)); a list that the atomic finds at its capacity closes its query for the wave
    // (after a scene cut nearly every bit is set: a hopeless pass costs about lcap entries per query, not N).
    // The buffer lives in the PAD of the staged rows: bytes 288 .. 303 of a 304-byte LDS row are neither written by the stash nor read
    // by the fragment loads; entry j of wave w is dword j & 3 of the pad of row 64 w + (j >> 2) (rows of both stages, numbered through).
    static_assert(F16_LDB == F16_K * 2 + 16, "the candidate buffer is the 16 pad bytes behind the 288 operand bytes of a staged row");
    static_assert(F16_WB == 63 * 4 && F16_DU * 64 >= F16_WB, "63 rows x 4 dwords of entries (the 64th row's pad is the closed-list table); one drain trip takes a full buffer");
    static_assert(F16_WB_TSPAN + 8 < 512, "9 bits of an entry carry its tile relative to the buffer's base tile");
    unsigned char* const wb = &Ah[0][0] + (size_t)wave * 64 * F16_LDB + 288;
    auto wb_at = [&](int j) -> unsigned* {                       // (24-bit multiply-add: full rate; the 32-bit multiply is a quarter-rate instruction)
        return reinterpret_cast<unsigned*>(wb + __umul24((unsigned)j >> 2, (unsigned)F16_LDB) + ((unsigned)j & 3u) * 4u);
    };
    // pad of the wave's 64th row: the wave's "list is full" table, one bit per query of its two blocks (set by whichever lane's atomic
    // finds a list at its capacity, read by the lane that owns the query after every drain)
    unsigned* const wfull = reinterpret_cast<unsigned*>(wb + 63 * F16_LDB);
    if (lane < NQB) wfull[lane] = 0u;
    int wcnt = 0, tbase = split * p.tiles_per_split;
    auto drain = [&](int next_tile, bool last) {
        if (DBG & 16) { wcnt = 0; tbase = next_tile; return; }   // tools: pushes only
        // F16_DU entries per lane go through decode -> atomic -> stores TOGETHER: one memory round trip per drain, not one per 64
        // entries.  A list found at its capacity closes its query for the whole wave (wfull): after a scene cut, or for a block of
        // queries whose hint is garbage, nearly every bit is set and every split's wave would otherwise keep bumping the SAME counters
        // (same-address atomics serialise in L2: 70 us instead of 36 with 128 such queries, profiles/r06_filter_lists.txt).
        for (int base = 0; base < wcnt; base += 64 * F16_DU) {
            unsigned bm[F16_DU]; int qq[F16_DU], g0[F16_DU], sl[F16_DU];
#pragma unroll
            for (int u = 0; u < F16_DU; ++u) {
                const int i = base + 64 * u + lane;
                const unsigned e = i < wcnt ? *wb_at(i) : 0u;
                bm[u] = e & 0xffffu;
                const int fl = (int)(e >> 16) & 63, tile = tbase + (int)(e >> 23);
                qq[u] = (b0 + (int)((e >> 22) & 1u)) * 32 + (fl & 31);
                const SegDev sd = seg_of_tile(p, tile);
                const int row0 = (tile - sd.tile0) * AFF_ROWS + 4 * (fl >> 5);
                if (row0 + 28 > sd.n) {                          // the segment's clamped last tile: rows past its end are copies of its last row
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (row0 + (r & 3) + 8 * (r >> 2) >= sd.n) bm[u] &= ~(0x8000u >> r);
                }
                g0[u] = sd.base + row0;
            }
#pragma unroll
            for (int u = 0; u < F16_DU; ++u) {
                sl[u] = 0;                                       // (tools, 32: no atomics - every halfword lands on slot 0)
                if (bm[u] != 0u && !(DBG & 32)) sl[u] = atomicAdd(&p.gcnt[(size_t)qq[u] * F16_CS], __popc(bm[u]));
            }
#pragma unroll
            for (int u = 0; u < F16_DU; ++u) {
                if (bm[u] != 0u) {
                    unsigned m = bm[u];
                    int s_ = sl[u];
                    if (s_ + __popc(m) >= p.lcap) {              // a list that reaches its capacity counts as overflowed
                        p.flag_out[qq[u] >> 7] = 1;
                        atomicOr(&wfull[(qq[u] >> 5) - b0], 1u << (qq[u] & 31));
                        if (s_ >= p.lcap) m = 0u;
                    }
                    int* const dst = p.gcand32 + (size_t)qq[u] * p.lstride;
                    while (m) {
                        const int r = __clz((int)m) - 16;        // bit 15 - r
                        m &= ~(0x8000u >> r);
                        if (s_ < p.lcap) dst[s_] = g0[u] + (r & 3) + 8 * (r >> 2);
                        ++s_;
                    }
                }
            }
        }
        wcnt = 0; tbase = next_tile;
        if (last) return;
#pragma unroll
        for (int i = 0; i < NQB; ++i)
            if ((wfull[i] >> l31) & 1u) qopen[i] = 0u;
    };
    unsigned long long pmk0 = 0ull, pmk1 = 0ull;                 // lane masks of the two compares of the previous MFMA slot
    unsigned sink = 0u;                                          // (tools: knock-out 8 keeps the compares alive without the stores)

    // one k-step of the CURRENT tile (accumulators CUR, fragments from AR) with the compares of the PENDING tile (PRV, tile number
    // PT, stored when PTOK) behind its MFMAs; the fragment of step T + F16_PF is requested first - for the last steps of a tile
    // that is a first fragment of the NEXT tile of the same stage (ARN; the stage's last tile has no successor before the barrier)
#define F16P_COMPARE(PRV, M, PT, PTOK)                                                                                \
    {                                                                                                                 \
        /* consume: the two lane masks slot M - 1 produced are shifted into their array's halfword (bit 15 - r <-> r) */   \
        if ((M) >= 2 && (M) < 2 + 8 * NQB) {                                                                          \
            const int a_ = ((M) - 2) >> 3, e_ = (((M) - 2) & 7) * 2;                                                  \
            unsigned long long junk0, junk1;                                                                          \
            asm volatile("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(bits[a_]), "=s"(junk0) : "s"(pmk0));                     \
            asm volatile("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(bits[a_]), "=s"(junk1) : "s"(pmk1));                     \
            if ((DBG & 8) && e_ == 14) sink ^= bits[a_];                                                              \
            else if (e_ == 14 && (PTOK)) {                                                                            \
                const bool nz_ = (bits[a_] & qopen[a_]) != 0u;                                                        \
                const unsigned long long nm_ = __ballot(nz_);                                                         \
                if (nz_) *wb_at(wcnt + __builtin_amdgcn_mbcnt_hi((unsigned)(nm_ >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)nm_, 0u))) = \
                    bits[a_] | ((unsigned)lane << 16) | ((unsigned)a_ << 22) | ((unsigned)((PT) - tbase) << 23);      \
                wcnt += __popcll(nm_);                                                                                \
            }                                                                                                         \
        }                                                                                                             \
        /* produce: two compares of the pending tile into two SGPR pairs of their own (nothing waits for them here) */     \
        if ((M) >= 1 && (M) < 1 + 8 * NQB) {                                                                          \
            const int a_ = ((M) - 1) >> 3, e_ = (((M) - 1) & 7) * 2;                                                  \
            if (e_ == 0) bits[a_] = 0u;                                                                               \
            asm volatile("v_cmp_nlt_f32_e64 %0, %1, %2" : "=s"(pmk0) : "v"(PRV[a_][e_]), "v"(my_tau[a_]));                      \
            asm volatile("v_cmp_nlt_f32_e64 %0, %1, %2" : "=s"(pmk1) : "v"(PRV[a_][e_ + 1]), "v"(my_tau[a_]));                  \
        }                                                                                                             \
    }
#define F16P_KSTEP(T, CUR, PRV, AR, ARN, HASN, PT, PTOK)                                                              \
    {                                                                                                                 \
        if ((T) + F16_PF < 9) fr[((T) + F16_PF) % 9] = *reinterpret_cast<const h16x8*>((AR) + ((T) + F16_PF) * 32);   \
        else if (HASN) fr[((T) + F16_PF) % 9] = *reinterpret_cast<const h16x8*>((ARN) + (((T) + F16_PF) % 9) * 32);   \
        _Pragma("unroll") for (int i = 0; i < NQB; ++i) {                                                             \
            if ((T) == 0) {                                                                                           \
                f32x16 z;                                                                                             \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) z[r] = 0.f;                                            \
                CUR[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[0], bq[i][0], z, 0, 0, 0);                         \
            } else {                                                                                                  \
                CUR[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[T], bq[i][T], CUR[i], 0, 0, 0);                    \
            }                                                                                                         \
            F16P_COMPARE(PRV, (T) * NQB + i, PT, PTOK)                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
        }                                                                                                             \
    }
#define F16P_TILE(CUR, PRV, AR, ARN, HASN, PT, PTOK)                                                                  \
    F16P_KSTEP(0, CUR, PRV, AR, ARN, HASN, PT, PTOK) F16P_KSTEP(1, CUR, PRV, AR, ARN, HASN, PT, PTOK)                 \
    F16P_KSTEP(2, CUR, PRV, AR, ARN, HASN, PT, PTOK) F16P_KSTEP(3, CUR, PRV, AR, ARN, HASN, PT, PTOK)                 \
    F16P_KSTEP(4, CUR, PRV, AR, ARN, HASN, PT, PTOK) F16P_KSTEP(5, CUR, PRV, AR, ARN, HASN, PT, PTOK)                 \
    F16P_KSTEP(6, CUR, PRV, AR, ARN, HASN, PT, PTOK) F16P_KSTEP(7, CUR, PRV, AR, ARN, HASN, PT, PTOK)                 \
    F16P_KSTEP(8, CUR, PRV, AR, ARN, HASN, PT, PTOK)

    int tido = tid;
    F16_FETCH(0)
    F16_STASH(&Ah[0][0])
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        asm volatile("" : "+v"(tido));
        if (!(DBG & 1)) F16_FETCH(min(s + 1, nst - 1))
        if (active) {
            const int tile = t_begin + s * F16_STAGE;
            const unsigned char* ar = &Ah[s & 1][0] + l31 * F16_LDB + lh * 16;
            constexpr int TILEB = AFF_ROWS * F16_LDB;
            // (tiles past the split's end hold copies of its last tile: their MFMAs run, their bits are not stored)
#pragma unroll
            for (int t = 0; t < F16_PF; ++t) fr[t] = *reinterpret_cast<const h16x8*>(ar + t * 32);
#pragma nounroll
            for (int g = 0; g < F16_STAGE; g += 2, ar += 2 * TILEB) {
                // tile g into set A behind the compares of tile g - 1 (set B; for g = 0 the last tile of the previous stage),
                // tile g + 1 into set B behind the compares of tile g
                // (each tile pushes at most 128 entries - 64 lanes x 2 query blocks - of the tile before it)
                if (wcnt > F16_WB - 128 || tile + g - tbase > F16_WB_TSPAN) drain(tile + g - 1, false);
                F16P_TILE(cA, cB, ar, ar + TILEB, true, tile + g - 1, (s > 0 || g > 0) && tile + g - 1 < t_end)
                if (wcnt > F16_WB - 128) drain(tile + g, false);
                F16P_TILE(cB, cA, ar + TILEB, ar + 2 * TILEB, g + 2 < F16_STAGE, tile + g, tile + g < t_end)
            }
        }
        if (!(DBG & 4)) {
            F16_STASH(&Ah[(s + 1) & 1][0])
            __syncthreads();
        }
    }
    if (active) {                                                // the last tile's compares have no MFMAs to hide behind
        const int lt = t_begin + (nst - 1) * F16_STAGE + F16_STAGE - 1;
        if (wcnt > F16_WB - 128) drain(lt, false);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int m = 1; m < 2 + 8 * NQB; ++m) { F16P_COMPARE(cB, m, lt, lt < t_end) }
        drain(lt, true);
    }
    if ((DBG & 8) && sink == 0x12345u) p.cnt_diag[0] = (int)sink;
#undef F16P_TILE
#undef F16P_KSTEP
#undef F16P_COMPARE
#undef F16_FETCH
#undef F16_STAGE_SETUP
#undef F16_FETCH_ONE
#undef F16_STASH
#undef F16_STASH_ONE
}

// ============================================================ refine ===================================================
// One workgroup (4 waves) per query: exact similarities of its candidates, 64 per wave and round, each wave keeping a running
// list of its best (bitwise-descent selection + counting ranks); wave 0 merges the four lists and applies the softmax exactly
// as affinity_merge16_kernel computes it.
#define RF_BUF 256             // running list of a wave: compacted to the best top_k whenever another 64 might not fit
#ifndef RF_MINWG
#define RF_MINWG 1
#endif
// TIGHTEN = true (between the two passes, flagged tiles only): instead of the outputs, the k-th best exact similarity of the
// listed elements becomes the query's bound (valid: k real elements reach it) and its list is emptied for pass 2.
template <bool TIGHTEN>
__global__ __launch_bounds__(64 * RF_WAVES, RF_MINWG) void affinity_refine_kernel(Filter16Args p) {
    constexpr int CK = 64;
    __shared__ __attribute__((aligned(16))) float s_op[2 * CK];
    __shared__ __attribute__((aligned(16))) u64 s_keys[RF_WAVES][RF_BUF + 2];
    __shared__ int s_n[RF_WAVES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = blockIdx.x;
    if (TIGHTEN && p.flag1[q >> 7] == 0) return;
    // a list that overflowed even with the tightened bound (thousands of exact ties): every memory element is evaluated
    // capacity of this query's list: a tile that went through the second pass lists up to lstride entries, the others lcap1
    const int cap = (!TIGHTEN && p.flag1[q >> 7] != 0) ? p.lstride : p.lcap1;
    const int listed = p.gcnt[(size_t)q * F16_CS];
    const bool full = !TIGHTEN && p.flag2[q >> 7] != 0 && listed >= cap;
    float* ne = s_op; float* ke2 = s_op + CK; u64* keys = s_keys[wv];
    if (wv == 0) {
        const float k = p.qk[(size_t)q * CK + lane];
        const float e = p.qe ? p.qe[(size_t)q * CK + lane] : 1.f;
        ne[lane] = -e; ke2[lane] = 2.f * (k * e);
    }
    const float bs = p.qmeta[(size_t)q * 4];          // b_sq with the select kernels' arithmetic (bound kernel)
    int total = min(listed, cap);
    if (!TIGHTEN && threadIdx.x == 0) p.cnt_diag[q] = listed;      // diagnostics (xmem_affinity_debug_offsets): the list length as one dense array
    if (full) { total = 0; for (int i = 0; i < p.n_seg; ++i) total += p.seg[i].n; }
    const int* list = p.gcand32 + (size_t)q * p.lstride;
    // the first round's indices are requested without waiting for the count (the list is lcap >= 2048 long; stale entries are
    // never used: every use is guarded by e < total)
    int gi_next = full ? wv * 64 + lane : list[wv * 64 + lane];
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
    for (int b0 = wv * 64; b0 < total; b0 += 64 * RF_WAVES) {
        const int e = b0 + lane;
        const int gi = gi_next;
        if (e + 64 * RF_WAVES < total) gi_next = full ? e + 64 * RF_WAVES : list[e + 64 * RF_WAVES];       // next round's index in flight under this round
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
        for (int w = 1; w < RF_WAVES; ++w) {
            const int nw = s_n[w];
            if (lane < nw) keys[cnt + lane] = s_keys[w][lane];
            cnt += nw;
        }
        __builtin_amdgcn_wave_barrier();
    } else if (wv != 0) return;
    cnt = compact(cnt, p.top_k);
    if (TIGHTEN) {
        if (lane == 0) {
            if (cnt >= p.top_k) p.tau[q] = fmaxf(p.tau[q], key_val(keys[p.top_k - 1]));
            p.gcnt[(size_t)q * F16_CS] = 0;
        }
        return;
    }
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

// per-query candidate list length: ~N/64 (redundant memories keep thousands of near-ties for a few queries), 2048 .. 16384
int aff_filter16_list_cap(int n_total) {
    static const int cmin = getenv("XMEM_AFF_LCAP_MIN") ? atoi(getenv("XMEM_AFF_LCAP_MIN")) : 2048;     // tools: A/B of the floor
    int c = cmin >= 256 && cmin <= 16384 ? cmin : 2048;
    while (c < n_total / 64 && c < 16384) c *= 2;
    return c;
}

// allocation stride of the lists = capacity of the SECOND pass: 4x the first pass's, at most 16384
int aff_filter16_list_stride(int n_total) {
    const int c = aff_filter16_list_cap(n_total);
    return c >= 4096 ? 16384 : 4 * c;
}

size_t aff_filter16_rows_bytes(int n_total) { return ((size_t)n_total + AFF_ROWS) * F16_K * sizeof(_Float16); }

// Measurement aid (bench.py): two HIP events the next hinted calls record right before / after their pass-1 filter launch, on the
// launch stream.  NULL, NULL turns it off.  The pair belongs to the CALLING HOST THREAD (thread_local): calls issued by other threads,
// on whatever stream or device, never see it - the library keeps no process-wide mutable state.  A tool's hook, not the data path.
static thread_local hipEvent_t g_prof_ev[2] = {nullptr, nullptr};
extern "C" int xmem_affinity_profile_events(void* before_filter, void* after_filter) {
    g_prof_ev[0] = reinterpret_cast<hipEvent_t>(before_filter);
    g_prof_ev[1] = reinterpret_cast<hipEvent_t>(after_filter);
    return XMEM_OK;
}

extern "C" int xmem_affinity_rows16(const float* key, const float* shrinkage, int n, void* rows16, void* stream) {
    if (n < 0 || (n > 0 && (!key || !rows16))) return XMEM_ERR_BAD_ARG;
    if (n == 0) return XMEM_OK;
    hipLaunchKernelGGL(affinity_rows16_kernel, dim3(cdiv(n, 16)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), key, shrinkage, n,
                       reinterpret_cast<_Float16*>(rows16));
    return xmem_check_launch();
}

int aff_filter16_launch(Filter16Args a, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // 8-wave workgroups (512 queries, one per CU) when the query count fills them and the memory is long enough: the rows
    // cross the fabric once per 512 instead of once per 256 queries
    int nw = (a.HW >= 2048 && a.total_tiles >= 8192) ? 8 : 4;
    if (const char* e = getenv("XMEM_F16_WAVES")) nw = atoi(e) == 8 ? 8 : 4;   // tools: A/B
    const int wgq = 64 * nw, stage = nw, per_xcd = nw == 4 ? 32 * F16_WG_PER_CU : 32;
    const int qt = cdiv(a.HW, wgq);
    // all workgroups of a split resident on one XCD at once when the query tiles allow: 8 x floor(per_xcd / query tiles) splits,
    // whole LDS stages, >= 2 stages per split
    int sp = 8 * (per_xcd / qt); if (sp < 8) sp = 8;
    if (const char* e = getenv("XMEM_F16_SPLITS")) { sp = atoi(e); if (sp < 1) sp = 1; }   // tools: A/B of the split count
    { int maxs = a.total_tiles / (2 * stage); if (maxs < 1) maxs = 1; if (sp > maxs) sp = maxs; }
    a.tiles_per_split = cdiv(cdiv(a.total_tiles, sp), stage) * stage;
    a.splits = cdiv(a.total_tiles, a.tiles_per_split);
    int n_total = 0;
    for (int i = 0; i < a.n_seg; ++i) n_total += a.seg[i].n;
    a.qtiles = qt;
    const dim3 fgrid(8 * qt * cdiv(a.splits, 8));
    int rc;
    // operand rows: kept by the caller (xmem_key_segment.rows16), else derived into the workspace for this call
    for (int i = 0; i < a.n_seg; ++i) {
        if (a.seg[i].rows16 || a.seg[i].n == 0) continue;
        _Float16* dst = a.rows16 + (size_t)a.seg[i].base * F16_K;
        hipLaunchKernelGGL(affinity_rows16_kernel, dim3(cdiv(a.seg[i].n, 16)), dim3(256), 0, s, a.seg[i].key, a.seg[i].shr, a.seg[i].n, dst);
        if ((rc = xmem_check_launch()) != XMEM_OK) return rc;
        a.seg[i].rows16 = dst;
    }
    // pass 1: every tile, the caller's bound
    a.only = nullptr; a.flag_out = a.flag1;
    if (g_prof_ev[0]) (void)hipEventRecord(g_prof_ev[0], s);       // tools: bracket the pass-1 filter (xmem_affinity_profile_events)
#ifdef XMEM_TOOLS
    // knock-outs of the pipelined kernel (wrong results, timing only): 1 no row fetch after the first stage, 4 no LDS stage
    // hand-over (stash + barrier), 8 no candidate pushes, 16 pushes but no drain, 32 drain without atomics
    const int dbg = getenv("XMEM_F16_DBG") ? atoi(getenv("XMEM_F16_DBG")) : 0;
#define F16P_DBG_CASE(D)                                                                                              \
    if (dbg == (D)) {                                                                                                 \
        if (nw == 8) hipLaunchKernelGGL((affinity_filter16_kernel<false, 8, D>), fgrid, dim3(512), 0, s, a);         \
        else hipLaunchKernelGGL((affinity_filter16_kernel<false, 4, D>), fgrid, dim3(256), 0, s, a);                 \
    } else
    F16P_DBG_CASE(1) F16P_DBG_CASE(4) F16P_DBG_CASE(5) F16P_DBG_CASE(8) F16P_DBG_CASE(13) F16P_DBG_CASE(16) F16P_DBG_CASE(32)
#undef F16P_DBG_CASE
#endif
    if (nw == 8) hipLaunchKernelGGL((affinity_filter16_kernel<false, 8>), fgrid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((affinity_filter16_kernel<false, 4>), fgrid, dim3(256), 0, s, a);
    if ((rc = xmem_check_launch()) != XMEM_OK) return rc;
    if (g_prof_ev[1]) (void)hipEventRecord(g_prof_ev[1], s);
    // pass 2: tiles with an overflowed list, with the bound their partial lists give (these two launches return at once
    // when nothing is flagged - the normal frame)
    a.only = a.flag1; a.flag_out = a.flag2;
    a.lcap = a.lstride;                                 // the second pass may list up to the allocation stride
    hipLaunchKernelGGL(affinity_refine_kernel<true>, dim3(a.HW), dim3(64 * RF_WAVES), 0, s, a);
    if ((rc = xmem_check_launch()) != XMEM_OK) return rc;
    if (nw == 8) hipLaunchKernelGGL((affinity_filter16_kernel<true, 8>), fgrid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((affinity_filter16_kernel<true, 4>), fgrid, dim3(256), 0, s, a);
    if ((rc = xmem_check_launch()) != XMEM_OK) return rc;
    hipLaunchKernelGGL(affinity_refine_kernel<false>, dim3(a.HW), dim3(64 * RF_WAVES), 0, s, a);
    return xmem_check_launch();
}
