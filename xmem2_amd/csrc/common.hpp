// Shared helpers for the gfx950 kernels.  Written for CDNA4 only (wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <unordered_map>
#include "../../include/xmem_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define XMEM_WAVE 64

static inline int xmem_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? XMEM_OK : XMEM_ERR_LAUNCH;
}

// Dynamic LDS above the 64 KiB default needs hipFuncAttributeMaxDynamicSharedMemorySize on the kernel.  Set it when a (kernel, device)
// first needs a size (or a larger one than it was given before) - not on every launch: the runtime call costs host time on a
// launch-bound path.  The table is a cache of an idempotent, monotone setting (one per translation unit that includes this header).
static inline int xmem_ensure_dynamic_lds(const void* fn, size_t bytes) {
    if (bytes <= 64 * 1024) return XMEM_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return XMEM_ERR_LAUNCH;
    static std::mutex mu;
    static std::unordered_map<uint64_t, size_t> granted;
    const uint64_t key = (uint64_t)(uintptr_t)fn ^ ((uint64_t)(unsigned)dev << 56);
    std::lock_guard<std::mutex> lock(mu);
    size_t& cur = granted[key];
    if (bytes <= cur) return XMEM_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return XMEM_ERR_LAUNCH;
    cur = bytes;
    return XMEM_OK;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
