// Streaming batched GEMM on the CDNA4 fp32 matrix pipe (v_mfma_f32_32x32x2_f32, exact fmaf chain).
//
//   C[g][m][n] = sum_k A[g][m][k] * B[g][n][k]        g < G groups, K % 32 == 0
//
// serves the Winograd-domain position GEMMs (G = 36 / 16 positions, M = tiles, K = Cin, N = Cout; the bare accumulator is
// stored) and the pointwise (1x1) convolutions (G = 1, fused scale / shift / residual / relu epilogue).  These GEMMs are SHORT:
// K = 64 ... 512, i.e. 2 ... 16 staged k-tiles per output tile.  A kernel that gives every 64x64 output tile its own workgroup
// spends most of a workgroup's life outside the MFMA loop (dispatch, index arithmetic, the first global -> LDS round trip, the
// store tail): measured 15 000 cycles of wave lifetime for 2 048 cycles of MFMA at K = 64 (profiles/r03_conv64_counters.txt).
//
// Here ONE workgroup walks a contiguous run of output tiles ("units") and the (unit, k-tile) sequence is ONE flat software
// pipeline: operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers, no ds_write) into a ring
// of NS stages, NS - 1 k-tiles ahead of the MFMAs and straight across unit boundaries; one s_barrier per k-tile; the stores of a
// finished unit are issued asynchronously and drain under the next unit's MFMAs.  Waits are counted (s_waitcnt vmcnt(N), never
// a drain to zero in the steady state).
//
// LDS image of a stage: [BM + BN rows][32 floats] with NO padding (LDS-DMA writes 64 lanes x 16 B linearly); bank conflicts of
// the fragment reads are avoided by an XOR swizzle applied on the SOURCE side: the 16-byte slot s of tile row r holds the global
// chunk s ^ ((r >> 1) & 7), so the 16 rows of a ds_read_b128 lane group land on 16 distinct 16-byte slots of the 256-byte bank
// line (rows r and r + 1 are the two halves of a line; (r >> 1) & 7 spreads 8 row pairs over the 8 slots of each half).
//
// MFMA lane maps and the k permutation are those of conv_mfma.hip: a lane reads 4 consecutive k (one 16-byte chunk, chunk index
// 2 kk + (lane >> 5) of k-group kk) and MFMA step j contracts k = {j, 4 + j} of the group for A and B alike.
#include "gemm_stream.hpp"
#include <stdlib.h>

// Tools knobs (XMEM_STREAM_DBG, only in a -DXMEM_TOOLS build: XMEM_HIPCC_FLAGS=-DXMEM_TOOLS python -m xmem2_amd.build --force):
// 1 = every unit loads unit 0's operands, 2 = no stores, 4 = no barrier, 8 = no LDS-DMA in the loop, 16 = MFMAs on constants,
// 32 = contiguous unit runs per workgroup.  Results are wrong with any of 1..16; they exist to attribute time.
#ifdef XMEM_TOOLS
#define DBG(bit) (p.dbg & (bit))
#else
#define DBG(bit) 0
#endif

namespace {

// LDS-DMA: 64 lanes x 16 bytes from (uniform base + per-lane 32-bit byte offset) to LDS bytes [lds_addr, lds_addr + 1024).
// M0 is written in the statement that reads it (the compiler does not preserve it across statements).
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_loads_only() {    // tools (XMEM_STREAM_DBG & 4): the same wait without the barrier - results are then wrong
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_loads_then_barrier() {
    // every LDS-DMA of the step about to be consumed has landed for THIS wave (loads complete in issue order: at most N younger
    // vector-memory operations may still be in flight), then all waves meet: the step's tile is complete and nobody still reads
    // the stage that is refilled next.
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" : : "n"(N) : "memory");
}

template <int TM, int TN, int NS, int MODE>
__global__ __launch_bounds__(256) void gemm_stream_kernel(GemmStreamArgs p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int STAGE = (BM + BN) * 128;             // bytes per ring stage
    constexpr int LA = BM / 32, LB = BN / 32;          // LDS-DMA instructions per wave and step: A rows / B rows (8 rows each)
    constexpr int L = LA + LB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // XCD-aware order of the workgroups' unit runs: consecutive runs (same A rows, neighbouring B rows) stay on one XCD / L2
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // Units are dealt round-robin over the (XCD-ordered) workgroups: at any time the workgroups of one XCD walk a narrow band of
    // consecutive units - one tile position, a few dozen row tiles, all column tiles - so that both operands of the band stay in
    // that XCD's 4 MiB L2 (contiguous runs per workgroup, as first written, spread the resident workgroups over all 36
    // positions: 9.4 MB of weights in flight, L2 hit rate 0.43 instead of 0.65, 4x the algorithmic fetch bytes).
    const int ustep = DBG(32) ? 1 : (int)gridDim.x;
    const int u0 = DBG(32) ? bid * p.units_per_wg : bid;
    const int u1 = DBG(32) ? min(p.units, u0 + p.units_per_wg) : p.units;
    if (u0 >= u1) return;
    const int nunits = (u1 - u0 + ustep - 1) / ustep;
    const int S = nunits * p.nk;                       // pipeline steps of this workgroup
    const unsigned lds_base = (unsigned)(size_t)smem;  // LDS byte address of the ring (low half of the flat address)

    // ---- load side -------------------------------------------------------------------------------------------------
    // instruction i of a wave covers tile rows 8 q .. 8 q + 7, q = 4 i + wave; lane -> (row 8 q + (lane >> 3), slot lane & 7)
    const int lrow = lane >> 3, lslot = lane & 7;
    unsigned a_voff[LA], b_voff[LB];
    int ld_u = u0, ld_kt = 0;
    const char* ld_abase = nullptr;
    const char* ld_bbase = nullptr;
    auto ld_unit_setup = [&]() {                       // operand row offsets of unit ld_u (rows past M / N are clamped: never stored)
        const int lu = DBG(1) ? 0 : ld_u;
        const int tn = lu % p.tiles_n, t2 = lu / p.tiles_n;
        const int tm = t2 % p.tiles_m, g = t2 / p.tiles_m;
        const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int r = 8 * (4 * i + wave) + lrow;
            const int m = min(m0 + r, p.M - 1);
            unsigned row = (unsigned)m;
            if (MODE == 1 && p.stride != 1) {
                const int b = m / p.HoWo, rem = m - b * p.HoWo;
                const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
                row = (unsigned)((b * p.H + oh * p.stride) * p.W + ow * p.stride);
            }
            a_voff[i] = (row * (unsigned)p.lda + (unsigned)((lslot ^ ((r >> 1) & 7)) * 4)) * 4u;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int r = 8 * (4 * i + wave) + lrow;
            const int n = min(n0 + r, p.N - 1);
            b_voff[i] = ((unsigned)n * (unsigned)p.ldb + (unsigned)((lslot ^ ((r >> 1) & 7)) * 4)) * 4u;
        }
        ld_abase = reinterpret_cast<const char*>(p.A + (size_t)g * p.a_gstride);
        ld_bbase = reinterpret_cast<const char*>(p.B + (size_t)g * p.b_gstride);
    };
    auto issue = [&](int stage) {                      // LDS-DMA of step (ld_u, ld_kt) into ring stage `stage`, then advance
        const char* ab = ld_abase + (size_t)ld_kt * 128;
        const char* bb = ld_bbase + (size_t)ld_kt * 128;
        const unsigned dst = lds_base + (unsigned)stage * STAGE + (unsigned)wave * 1024u;
#pragma unroll
        for (int i = 0; i < LA; ++i) glds16(ab, a_voff[i], dst + i * 4096);
#pragma unroll
        for (int i = 0; i < LB; ++i) glds16(bb, b_voff[i], dst + BM * 128 + i * 4096);
        if (++ld_kt == p.nk) {
            ld_kt = 0;
            ld_u += ustep;
            if (ld_u < u1) ld_unit_setup();
        }
    };

    // ---- compute side ----------------------------------------------------------------------------------------------
    // fragment byte offsets inside a stage for the four k-groups: row * 128 + 16 * ((2 kk + lh) ^ swizzle(row))
    const int swz = (l31 >> 1) & 7;
    unsigned rd_a[4], rd_b[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const unsigned ch = (unsigned)(((2 * kk + lh) ^ swz) * 16);
        rd_a[kk] = (unsigned)((wm * 32 * TM + l31) * 128) + ch;
        rd_b[kk] = (unsigned)(BM * 128 + (wn * 32 * TN + l31) * 128) + ch;
    }

    ld_unit_setup();
#pragma unroll
    for (int j = 0; j < NS - 1; ++j)
        if (j < S) issue(j);

    // Stores of a finished unit sit in the same in-order vector-memory queue as the loads.  For the NS - 1 steps that follow a
    // unit whose blocks were all full (exactly ST store instructions issued by this wave), the awaited load group is older than
    // that burst, so the burst may stay in flight as well; otherwise the wait simply covers it (conservative, still correct).
    constexpr int ST = (MODE == 0) ? 16 * TM * TN : 0;
    constexpr int WAIT_STEADY = L * (NS - 2);
    constexpr int WAIT_CREDIT = (WAIT_STEADY + ST > 63) ? 63 : WAIT_STEADY + ST;
    int s = 0, stage = 0, credit = 0;
    for (int cu = u0; cu < u1; cu += ustep) {
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int kt = 0; kt < p.nk; ++kt, ++s) {
            // loads of steps s+1 .. s+NS-2 may stay in flight (L each); the tail of the run has fewer steps behind it
            const int behind = S - 1 - s;
            if (DBG(4)) wait_loads_only<WAIT_STEADY>();
            else if (behind >= NS - 2) {
                if (ST > 0 && credit > 0) wait_loads_then_barrier<WAIT_CREDIT>();
                else wait_loads_then_barrier<WAIT_STEADY>();
            } else if (NS > 3 && behind == 1) wait_loads_then_barrier<L>();
            else wait_loads_then_barrier<0>();
            if (credit > 0) --credit;
            if (s + NS - 1 < S && !DBG(8)) {
                int st = stage + NS - 1; if (st >= NS) st -= NS;
                issue(st);                              // refills the stage every wave finished reading before this barrier
            }

            const unsigned char* sb = smem + (DBG(16) ? 0 : stage * STAGE);
            f32x4 af[2][TM], bf[2][TN];
            if (DBG(16)) {                         // tools: MFMAs on register constants, no fragment reads
                const f32x4 c = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
                for (int i = 0; i < TM; ++i) { af[0][i] = c; af[1][i] = c; }
#pragma unroll
                for (int j = 0; j < TN; ++j) { bf[0][j] = c; bf[1][j] = c; }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][i][t], bf[0][j][t], acc[i][j], 0, 0, 0);
                if (++stage == NS) stage = 0;
                continue;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(sb + rd_a[0] + i * 4096);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(sb + rd_b[0] + j * 4096);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk + 1 < 4) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[nxt][i] = *reinterpret_cast<const f32x4*>(sb + rd_a[kk + 1] + i * 4096);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[nxt][j] = *reinterpret_cast<const f32x4*>(sb + rd_b[kk + 1] + j * 4096);
                }
                __builtin_amdgcn_sched_barrier(0);     // the prefetch stays above this k-group's MFMAs
                if (MODE == 1) {                       // relu-on-load of the pointwise layers: max(x, 0) or max(x, x)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        af[cur][i].x = fmaxf(af[cur][i].x, p.relu_in ? 0.f : af[cur][i].x); af[cur][i].y = fmaxf(af[cur][i].y, p.relu_in ? 0.f : af[cur][i].y);
                        af[cur][i].z = fmaxf(af[cur][i].z, p.relu_in ? 0.f : af[cur][i].z); af[cur][i].w = fmaxf(af[cur][i].w, p.relu_in ? 0.f : af[cur][i].w);
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][t], bf[cur][j][t], acc[i][j], 0, 0, 0);
            }
            if (++stage == NS) stage = 0;
        }

        // ---- unit finished: lane owns output column n and 16 rows per 32x32 block; the stores drain under the next unit ----
        const int tn = cu % p.tiles_n, t2 = cu / p.tiles_n;
        const int tm = t2 % p.tiles_m, g = t2 / p.tiles_m;
        const int m0 = tm * BM, n0 = tn * BN;
        float* const cg = p.C + (size_t)g * p.c_gstride;
        bool all_full = true;
        if (DBG(2)) { credit = 0; continue; }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = n0 + wn * 32 * TN + j * 32;             // wave-uniform first column of the block
            const int n = nb + l31;
            float sc = 1.f, sh = 0.f;
            if (MODE == 1 && n < p.N) { sc = p.scale[n]; sh = p.shift[n]; }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mw = m0 + wm * 32 * TM + i * 32;          // wave-uniform first row of the block
                const int mb = mw + 4 * lh;                         // this lane's rows: mb + (r & 3) + 8 * (r >> 2)
                float* const orow = cg + (size_t)mb * p.ldc + n;
                const bool full = (mw + 32 <= p.M) && (nb + 32 <= p.N);
                all_full = all_full && full;
                if (MODE == 0) {
                    if (full) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) orow[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldc] = acc[i][j][r];
                    } else if (n < p.N) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int o = (r & 3) + 8 * (r >> 2);
                            if (mb + o < p.M) orow[(size_t)o * p.ldc] = acc[i][j][r];
                        }
                    }
                } else if (n < p.N) {
                    float rv[16];
                    if (p.res) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int o = (r & 3) + 8 * (r >> 2);
                            const int t = p.res_mod ? (mb + o) % p.res_mod : min(mb + o, p.M - 1);     // clamped, unconditional load
                            rv[r] = p.res[(size_t)t * p.ldres + n];
                        }
                    }
                    // two phases (csrc/conv_mfma.hip, epilogue): no store between the uses of loaded operands
                    float vout[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[i][j][r] * sc + sh;
                        if (p.res) v += rv[r];
                        vout[r] = p.relu_out ? fmaxf(v, 0.f) : v;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = (r & 3) + 8 * (r >> 2);
                        if (mb + o < p.M) orow[(size_t)o * p.ldc] = vout[r];
                    }
                }
            }
        }
        credit = (ST > 0 && all_full) ? NS - 1 : 0;
    }
}

template <int TM, int TN, int NS>
int launch_variant(const GemmStreamArgs& a, int nwg, hipStream_t s) {
    const size_t lds = (size_t)NS * (64 * TM + 64 * TN) * 128;
    auto kern = a.mode == 1 ? gemm_stream_kernel<TM, TN, NS, 1> : gemm_stream_kernel<TM, TN, NS, 0>;
    if (xmem_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != XMEM_OK) return XMEM_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, a);
    return xmem_check_launch();
}

}  // namespace

size_t gemm_stream_lds_bytes(int variant, int ring) {
    const int bm = variant == 0 ? 64 : 128, bn = variant == 2 ? 128 : 64;
    return (size_t)ring * (bm + bn) * 128;
}

int gemm_stream_launch(GemmStreamArgs& a, int variant, int ring, hipStream_t s) {
    if (a.K % 32 != 0 || a.K <= 0 || a.M <= 0 || a.N <= 0 || a.G <= 0) return XMEM_ERR_UNSUPPORTED;
    if (variant < 0 || variant > 2 || (ring != 3 && ring != 4)) return XMEM_ERR_BAD_ARG;
    const int bm = variant == 0 ? 64 : 128, bn = variant == 2 ? 128 : 64;
    static const int dbg = getenv("XMEM_STREAM_DBG") ? atoi(getenv("XMEM_STREAM_DBG")) : 0;
    a.dbg = dbg;
    a.nk = a.K / 32;
    a.tiles_m = cdiv(a.M, bm); a.tiles_n = cdiv(a.N, bn);
    const long units = (long)a.G * a.tiles_m * a.tiles_n;
    if (units > 0x7fffffff) return XMEM_ERR_UNSUPPORTED;
    a.units = (int)units;
    // resident workgroups: LDS-bound (160 KiB per CU), 256 CUs.  Each workgroup takes a contiguous run of units; the run length
    // is the smallest that lets the grid fit the chip in ONE round, and the grid is then cut to runs of equal length.
    const size_t lds = gemm_stream_lds_bytes(variant, ring);
    int per_cu = (int)((160 * 1024) / lds); if (per_cu > 8) per_cu = 8; if (per_cu < 1) per_cu = 1;
    const int slots = 256 * per_cu;
    a.units_per_wg = cdiv(a.units, slots);
    const int nwg = cdiv(a.units, a.units_per_wg);     // every workgroup gets units_per_wg (the last ones one fewer) units, strided by nwg
    if (variant == 0) return ring == 3 ? launch_variant<1, 1, 3>(a, nwg, s) : launch_variant<1, 1, 4>(a, nwg, s);
    if (variant == 1) return ring == 3 ? launch_variant<2, 1, 3>(a, nwg, s) : launch_variant<2, 1, 4>(a, nwg, s);
    return ring == 3 ? launch_variant<2, 2, 3>(a, nwg, s) : launch_variant<2, 2, 4>(a, nwg, s);
}
