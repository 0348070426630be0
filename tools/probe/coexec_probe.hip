// Do the matrix pipe (v_mfma_f32_16x16x4_f32) and the VALU (v_pk_fma_f32 / v_fma_f32) of one SIMD run side by side when the two
// instruction streams come from two DIFFERENT waves of that SIMD?  One 512-thread workgroup per CU: waves 0-3 issue MFMAs,
// waves 4-7 issue VALU multiply-adds (mode bits choose which halves run).
//   coexec_probe <mode>   1 = MFMA waves only, 2 = VALU waves only (packed), 3 = both, 4 = plain v_fma only, 5 = MFMA + plain v_fma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void k(float *out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const float s = (float)threadIdx.x * 1e-9f;
    if (wave < 4) {
        if (!(mode & 1)) return;
        f32x4 acc[7];
        for (int i = 0; i < 7; ++i) acc[i] = f32x4{s, s, s, s};
        float a = s + 1.0f, b = s + 0.5f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 7; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        float t = 0.0f;
        for (int i = 0; i < 7; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * 512 + threadIdx.x] = t;
    } else {
        if (!(mode & 2) && !(mode & 4)) return;
        if (mode & 2) {
            f32x2 acc[16];
            for (int i = 0; i < 16; ++i) acc[i] = f32x2{s, s + 1.0f};
            f32x2 a = {s + 1.0f, s + 2.0f}, b = {0.999f, 1.001f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 14; ++u)   // 14 x 16 = 224 packed fmas = 4 x 7 MFMAs' worth of issue time (896 cycles)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(a, b, acc[i]);
            }
            float t = 0.0f;
            for (int i = 0; i < 16; ++i) t += acc[i][0] + acc[i][1];
            out[blockIdx.x * 512 + threadIdx.x] = t;
        } else {
            float acc[16];
            for (int i = 0; i < 16; ++i) acc[i] = s + i;
            float a = s + 1.0f, b = 0.999f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 14; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = fmaf(a, b, acc[i]);
            }
            float t = 0.0f;
            for (int i = 0; i < 16; ++i) t += acc[i];
            out[blockIdx.x * 512 + threadIdx.x] = t;
        }
    }
}
int main(int argc, char **argv) {
    float *out; if (hipMalloc(&out, 256 * 512 * 4) != hipSuccess) return 1;
    const int iters = 2000;
    for (int mode : {1, 2, 3, 4, 5}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfma_cycles = (double)iters * 28 * 32, valu_cycles = (double)iters * 224 * 4;
        printf("mode %d: %.1f us  (MFMA stream alone at 32 clk each: %.0f cycles, VALU stream at 4 clk each: %.0f cycles; at 2.4 GHz %.1f / %.1f us)\n", mode,
               ms * 1e3, mfma_cycles, valu_cycles, mfma_cycles / 2400.0, valu_cycles / 2400.0);
    }
    return 0;
}
