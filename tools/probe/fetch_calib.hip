// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the two read patterns of the env kernels (MI355X_MICROARCH.md, HBM
// section: "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern"):
//   k_stream : every lane reads 16 bytes, consecutive lanes consecutive addresses (the guide: counter = 1/2 of the bytes)
//   k_gather8: every lane reads ONE 8-byte word at a random 8-byte aligned address of a 2 GiB buffer (the health gather
//              of k_step<N,true>); how many bytes does the counter report per gather?
// hipcc --offload-arch=gfx950 -O3 -o fetch_calib_bin fetch_calib.hip ; rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- ./fetch_calib_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_stream(const uint4 *src, size_t n16, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) acc ^= src[i].x;
    if (acc == 0x12345u) *sink = acc;
}
__global__ void k_gather8(const double *src, size_t nwords, size_t ngather, double *sink) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ngather; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long h = (i + 1) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        acc += src[h % nwords];
    }
    if (acc == 1.2345) *sink = acc;
}

int main() {
    const size_t bytes = 2ull << 30;
    void *buf, *sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMemset(buf, 0, bytes));
    const size_t ngather = 16ull << 20;
    for (int it = 0; it < 4; ++it) {
        hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (const uint4 *)buf, bytes / 2 / 16, (unsigned *)sink);      // 1 GiB read
        hipLaunchKernelGGL(k_gather8, dim3(2048), dim3(256), 0, 0, (const double *)buf, bytes / 8, ngather, (double *)sink);  // 16 Mi gathers
    }
    CK(hipDeviceSynchronize());
    printf("k_stream reads %zu bytes per launch; k_gather8 makes %zu 8-byte gathers per launch\n", bytes / 2, ngather);
    return 0;
}
