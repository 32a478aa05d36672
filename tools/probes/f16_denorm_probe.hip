// Does v_mfma_f32_32x32x16_f16 flush fp16 SUBNORMAL operands?  (the error bound of the fp16 filter pass depends on it)
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/f16_denorm_probe.hip -o gpurun_out/f16_denorm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a, float b, float* o) {
    h16x8 A, B;
    for (int j = 0; j < 8; ++j) { A[j] = (_Float16)a; B[j] = (_Float16)b; }
    f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) { o[0] = c[0]; o[1] = (float)A[0]; o[2] = (float)B[0]; }
}
int main() {
    float* d; hipMalloc(&d, 16);
    const float cases[4][2] = {{9.5367431640625e-07f, 1024.f}, {1024.f, 9.5367431640625e-07f}, {3e-5f, 3e-5f}, {1.f, 1.f}};
    for (auto& cs : cases) {
        k<<<1, 64>>>(cs[0], cs[1], d);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a=%g b=%g  cvt a=%g b=%g  mfma=%.9g expected(no flush)=%.9g\n", cs[0], cs[1], h[1], h[2], h[0], 16.0 * (double)h[1] * (double)h[2]);
    }
    return 0;
}
