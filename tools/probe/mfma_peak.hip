// Peak probe: back-to-back v_mfma_f32_16x16x4_f32 on two accumulator chains, waves per SIMD = argv[1] (1 or 2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_peak(float *out, int iters, float a, float b) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1};
    float x = a + threadIdx.x, y = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, c1, 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c0[2] + c1[3];
}
int main(int argc, char **argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 1;
    const int threads = 256 * wps, grid = 256, iters = 4000;
    float *out; hipMalloc(&out, grid * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_peak, dim3(grid), dim3(threads), 0, 0, out, 100, 1e-9f, 1e-9f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_peak, dim3(grid), dim3(threads), 0, 0, out, iters, 1e-9f, 1e-9f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)grid * (threads / 64) * iters * 32;
    printf("waves/SIMD %d: %.3f ms, %.1f TFLOP/s, %.2f cycles@2.4GHz per MFMA per SIMD\n", wps, ms, mfmas * 2048 / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (mfmas / (grid * 4)));
    return 0;
}
