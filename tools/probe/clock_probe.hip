// Shader clock under matrix-core load: every wave of a full-chip launch runs a v_mfma_f32_16x16x4_f32 stream (two accumulator
// chains, as the conv kernels do); one lane per workgroup reads the shader-cycle counter (s_memtime) and the constant 100 MHz
// counter (s_memrealtime) around it.  cycles / (ticks * 10 ns) = the clock the kernel really ran at.
//   hipcc --offload-arch=gfx950 -O3 -o bin/clock_probe clock_probe.hip ; ./bin/clock_probe [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(int iters, unsigned long long *out, float *sink) {
    f32x4 a = {0, 0, 0, 0}, b = {1, 1, 1, 1};
    const float x = threadIdx.x * 0.001f, y = 1.0f + blockIdx.x * 1e-6f;
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            a = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, b, 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = t1 - t0; out[2 * gridDim.x + 2 * blockIdx.x] = t0; out[2 * gridDim.x + 2 * blockIdx.x + 1] = t1; }
    if (a[0] + b[0] == 12345.0f) sink[0] = a[1];
}
int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int grid = argc > 2 ? atoi(argv[2]) : 256;
    unsigned long long *d; float *s;
    hipMalloc(&d, grid * 4 * 8); hipMalloc(&s, 4);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("device %s: %d CUs, clock %d kHz, grid %d\n", pr.name, pr.multiProcessorCount, pr.clockRate, grid);
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int l = 0; l < 20; ++l) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, iters, d, s);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(grid * 4); hipMemcpy(h.data(), d, grid * 4 * 8, hipMemcpyDeviceToHost);
        double cyc = 0, tk = 0; for (int i = 0; i < grid; ++i) { cyc += h[2 * i]; tk += h[2 * i + 1]; }
        cyc *= 256.0 / grid; tk *= 256.0 / grid;
        unsigned long long tmin = ~0ull, tmax = 0, smax = 0; for (int i = 0; i < grid; ++i) { tmin = std::min(tmin, h[2 * grid + 2 * i]); tmax = std::max(tmax, h[2 * grid + 2 * i + 1]); smax = std::max(smax, h[2 * grid + 2 * i]); }
        printf("  last launch: first start -> last end %.1f us, latest start %.1f us after the first\n", (tmax - tmin) / 100.0, (smax - tmin) / 100.0);
        const double mfma = (double)iters * 32 * 2;  // per wave; 2 waves per SIMD
        printf("launch %.1f us; shader cycles/wg %.0f, 100MHz ticks/wg %.0f -> %.3f GHz; cycles per MFMA per SIMD %.1f; %.1f TFLOP/s\n", ms * 1e3 / 20, cyc / 256,
               tk / 256, cyc / tk / 10.0, (cyc / 256) / mfma, (double)grid * 8 * iters * 32 * 2048.0 / (ms * 1e-3 / 20) / 1e12);
    }
    return 0;
}
