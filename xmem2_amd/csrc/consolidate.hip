// Long-term memory consolidation and eviction (inference/memory_manager.py:316-390,
// inference/kv_memory_store.py:160-189).  These fire once every (T_max - T_min) * mem_every frames
// on ~8k candidates, so they are written as simple, exact, order-preserving kernels.
#include "common.hpp"
#include <math.h>

extern "C" int xmem_version(void) { return XMEM_ABI_VERSION; }

__global__ void xmem_trace_marker_kernel(int tag) { (void)tag; }
extern "C" int xmem_trace_marker(int tag, void* stream) {
    hipLaunchKernelGGL(xmem_trace_marker_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), tag);
    return xmem_check_launch();
}

extern "C" const char* xmem_last_error_string(int code) {
    switch (code) {
        case XMEM_OK: return "ok";
        case XMEM_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size or inconsistent stride)";
        case XMEM_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels (see include/xmem_hip.h)";
        case XMEM_ERR_WORKSPACE: return "workspace missing or too small";
        case XMEM_ERR_LAUNCH: return "HIP launch failed";
        case XMEM_ERR_TOPK: return "selected index k out of range: fewer memory elements than top_k";
        default: return "unknown xmem status";
    }
}

__global__ void usage_ratio_kernel(const float* __restrict__ use, const float* __restrict__ life, float* __restrict__ usage, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) usage[i] = use[i] / life[i];
}

extern "C" int xmem_usage_ratio(const float* use, const float* life, float* usage, int n, void* stream) {
    if (!use || !life || !usage || n <= 0) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(usage_ratio_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, use, life, usage, n);
    return xmem_check_launch();
}

// exact rank of every element (ties -> lower index first); rank < k is the answer, already sorted
__global__ void topk_1d_kernel(const float* __restrict__ v, int n, int k, int largest, int* __restrict__ out_idx, float* __restrict__ out_val) {
    __shared__ float tile[256];
    const int e = blockIdx.x * 256 + threadIdx.x;
    const float ve = e < n ? v[e] : 0.f;
    int rk = 0;
    for (int f0 = 0; f0 < n; f0 += 256) {
        const int f = f0 + threadIdx.x;
        __syncthreads();
        tile[threadIdx.x] = f < n ? v[f] : 0.f;
        __syncthreads();
        const int lim = min(256, n - f0);
        for (int j = 0; j < lim; ++j) {
            const float vf = tile[j];
            const bool before = largest ? (vf > ve) : (vf < ve);
            rk += before || (vf == ve && (f0 + j) < e);
        }
    }
    if (e < n && rk < k) { out_idx[rk] = e; out_val[rk] = ve; }
}

extern "C" int xmem_topk_1d(const float* values, int n, int k, int largest, int32_t* out_idx, float* out_val, void* stream) {
    if (!values || !out_idx || !out_val || n <= 0 || k <= 0) return XMEM_ERR_BAD_ARG;
    if (k > n) return XMEM_ERR_TOPK;
    hipLaunchKernelGGL(topk_1d_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, values, n, k, largest, out_idx, out_val);
    return xmem_check_launch();
}

__global__ void gather_rows_kernel(const float* __restrict__ src, int C, const int* __restrict__ index, int n, float* __restrict__ dst) {
    const size_t total = (size_t)n * C;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / C), c = (int)(e - (size_t)i * C);
        dst[e] = src[(size_t)index[i] * C + c];
    }
}

extern "C" int xmem_gather_rows(const float* src, int C, const int32_t* index, int n, float* dst, void* stream) {
    if (!src || !index || !dst || C <= 0 || n <= 0) return XMEM_ERR_BAD_ARG;
    const size_t total = (size_t)n * C;
    int g = (int)((total + 255) / 256); if (g > 4096) g = 4096;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, src, C, index, n, dst);
    return xmem_check_launch();
}

// stable softmax over the last `count` entries of each row, zeros before (memory_util.py:55-60 on a suffix slice)
__global__ void softmax_rows_suffix_kernel(float* __restrict__ sim, int n, int count) {
    __shared__ float red[4];
    float* row = sim + (size_t)blockIdx.x * n;
    const int start = n - count;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int i = start + threadIdx.x; i < n; i += 256) m = fmaxf(m, row[i]);
    m = wave_max(m);
    if (lane == 0) red[wv] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int i = start + threadIdx.x; i < n; i += 256) { const float e = expf(row[i] - m); row[i] = e; s += e; }
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    for (int i = start + threadIdx.x; i < n; i += 256) row[i] = row[i] / s;
    for (int i = threadIdx.x; i < start; i += 256) row[i] = 0.f;
}

extern "C" int xmem_softmax_rows_suffix(float* sim, int P, int n, int count, void* stream) {
    if (!sim || P <= 0 || n <= 0 || count <= 0 || count > n) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(softmax_rows_suffix_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, sim, n, count);
    return xmem_check_launch();
}

// top-k softmax of each row in place, zeros elsewhere (do_softmax with top_k on a materialised similarity, memory_util.py:41-54:
// topk -> exp WITHOUT max shift -> / sum -> scatter into zeros).  One workgroup per row: the k-th largest value by bitwise
// descent over order-preserving integer keys (32 counting passes over the row, L2 resident), exact ties at the threshold go to
// the lowest indices (a stable sort's choice), then exp / sum / scale.
__device__ __forceinline__ unsigned order_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // NaN sorts above +inf, as torch.topk ranks it
}
__global__ __launch_bounds__(256) void softmax_rows_topk_kernel(float* __restrict__ sim, int n, int k) {
    __shared__ int s_cnt[4];
    __shared__ float s_sum[4];
    float* row = sim + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    auto count_if = [&](auto pred) -> int {                     // number of row elements whose key satisfies pred (uniform result)
        int c = 0;
        for (int i = tid; i < n; i += 256) c += pred(order_key(row[i])) ? 1 : 0;
        c = wave_sum_i(c);
        __syncthreads();
        if (lane == 0) s_cnt[wv] = c;
        __syncthreads();
        return (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
    };
    unsigned thr = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = thr | (1u << bit);
        if (count_if([=](unsigned key) { return key >= cand; }) >= k) thr = cand;
    }
    const int cg = count_if([=](unsigned key) { return key > thr; });
    const int need = k - cg;                                    // >= 1 of the elements tied at the threshold
    // exp of the kept elements in place (0 elsewhere) and their sum; ties are ranked in index order, chunk by chunk
    float acc = 0.f;
    int seen = 0;                                               // ties in earlier chunks (uniform)
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        const float v = i < n ? row[i] : 0.f;
        const unsigned key = i < n ? order_key(v) : 0u;
        const bool tie = i < n && key == thr;
        const unsigned long long m = __ballot(tie);
        __syncthreads();
        if (lane == 0) s_cnt[wv] = __popcll(m);
        __syncthreads();
        int before = seen + __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wv; ++w) before += s_cnt[w];
        seen += (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
        const bool keep = i < n && (key > thr || (tie && before < need));
        const float e = keep ? expf(v) : 0.f;
        if (i < n) row[i] = e;
        acc += e;
    }
    acc = wave_sum(acc);
    if (lane == 0) s_sum[wv] = acc;
    __syncthreads();
    const float total = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
    for (int i = tid; i < n; i += 256) row[i] = row[i] / total;
}

extern "C" int xmem_softmax_rows_topk(float* sim, int P, int n, int k, void* stream) {
    if (!sim || P <= 0 || n <= 0 || k <= 0 || k > n) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(softmax_rows_topk_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, sim, n, k);
    return xmem_check_launch();
}

// out[p][c] = sum_i aff[p][n-count+i] * V[i][c];  block: 64 channels x 4 row stripes, 8 prototypes per block
#define WR_P 8
__global__ void weighted_rows_kernel(const float* __restrict__ aff, int P, int n, int count, const float* __restrict__ V, int C,
                                     float* __restrict__ out) {
    __shared__ float red[4][WR_P][64];
    const int cl = threadIdx.x & 63, stripe = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, p0 = blockIdx.y * WR_P;
    const int start = n - count;
    float acc[WR_P];
#pragma unroll
    for (int j = 0; j < WR_P; ++j) acc[j] = 0.f;
    if (c < C) {
        for (int i = stripe; i < count; i += 4) {
            const float v = V[(size_t)i * C + c];
#pragma unroll
            for (int j = 0; j < WR_P; ++j) {
                const int pp = p0 + j;
                if (pp < P) acc[j] += aff[(size_t)pp * n + start + i] * v;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < WR_P; ++j) red[stripe][j][cl] = acc[j];
    __syncthreads();
    if (stripe == 0 && c < C) {
#pragma unroll
        for (int j = 0; j < WR_P; ++j) {
            const int pp = p0 + j;
            if (pp < P) out[(size_t)pp * C + c] = (red[0][j][cl] + red[1][j][cl]) + (red[2][j][cl] + red[3][j][cl]);
        }
    }
}

extern "C" int xmem_weighted_rows(const float* aff, int P, int n, int count, const float* V, int C, float* out, void* stream) {
    if (!aff || !V || !out || P <= 0 || n <= 0 || count <= 0 || count > n || C <= 0) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(weighted_rows_kernel, dim3(cdiv(C, 64), cdiv(P, WR_P)), dim3(256), 0, (hipStream_t)stream, aff, P, n, count, V, C, out);
    return xmem_check_launch();
}

// order-preserving compaction of {i : usage[i] > *threshold}; single workgroup, 256-wide chunks
__global__ void select_greater_kernel(const float* __restrict__ usage, int n, const float* __restrict__ thr, int* __restrict__ out_index,
                                      int* __restrict__ out_count) {
    __shared__ int wsum[4];
    __shared__ int base_s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float t = thr[0];
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const bool keep = i < n && usage[i] > t;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wsum[wv] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wv; ++w) off += wsum[w];
        if (keep) out_index[off + __popcll(m & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (threadIdx.x == 0) base_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) out_count[0] = base_s;
}

extern "C" int xmem_select_greater(const float* usage, int n, const float* threshold_dev, int32_t* out_index, int32_t* out_count, void* stream) {
    if (!usage || !threshold_dev || !out_index || !out_count || n <= 0) return XMEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(select_greater_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, usage, n, threshold_dev, out_index, out_count);
    return xmem_check_launch();
}
