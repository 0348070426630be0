// Four write-only kernels, ten launches each over a 642 MB buffer, for rocprofv3 --pmc passes (tools/pmc_write_gap.sh): what do
// the L2 <-> fabric counters say about a one-store-per-thread fill in address order (~6.9 TB/s) versus the writers that top out
// at ~5.3 TB/s (many concurrent sequential streams)?   bin/write_probe_pmc [MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void fill_in_order(uint4 *dst, long npages) {   // page = block: ONE sequential stream
    dst[(long)blockIdx.x * 256 + threadIdx.x] = make_uint4(blockIdx.x, 2, 3, 4);
}
__global__ __launch_bounds__(256) void fill_512_streams(uint4 *dst, long npages) {  // consecutive blocks round-robin over 512 regions
    const long R = 512, per = npages / R, b = blockIdx.x;
    const long pg = b < per * R ? (b % R) * per + b / R : b;
    dst[pg * 256 + threadIdx.x] = make_uint4((unsigned)b, 2, 3, 4);
}
__global__ __launch_bounds__(256) void persistent_static_32k(uint4 *dst, long npages) {  // 1024 workgroups, 8 pages per tile
    for (long p0 = (long)blockIdx.x * 8; p0 < npages; p0 += (long)gridDim.x * 8)
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (p0 + s < npages) dst[(p0 + s) * 256 + threadIdx.x] = make_uint4(p0, s, 3, 4);
}
__global__ __launch_bounds__(256) void lds_tiles_like_k_observe(uint4 *dst, int tile16, int ntiles) {  // 31 360-byte tiles via LDS
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint4 *out = dst + (size_t)t * tile16;
        __syncthreads();
        for (int i = tid; i < tile16; i += 256) lds[i] = make_uint4(t, i, 3, 4);
        __syncthreads(); __syncthreads(); __syncthreads();
        if (tid < 192) for (int i = tid; i < tile16; i += 192) out[i] = lds[i];
    }
}

int main(int argc, char **argv) {
    const size_t bytes = ((argc > 1 ? atol(argv[1]) : 642) * 1000000ull) & ~(size_t)4095;
    uint4 *d; CK(hipMalloc(&d, bytes + (1 << 20)));
    const long npages = (long)(bytes / 4096);
    const int tile16 = 1960, ntiles = (int)(bytes / 16 / tile16);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char *name, auto f) {
        f(); f();
        CK(hipEventRecord(a));
        for (int i = 0; i < 10; ++i) f();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-28s %8.1f us  %.2f TB/s\n", name, ms * 100.f, bytes / (ms * 100.f) / 1e6);
    };
    run("fill_in_order", [&] { hipLaunchKernelGGL(fill_in_order, dim3((unsigned)npages), dim3(256), 0, 0, d, npages); });
    run("fill_512_streams", [&] { hipLaunchKernelGGL(fill_512_streams, dim3((unsigned)npages), dim3(256), 0, 0, d, npages); });
    run("persistent_static_32k", [&] { hipLaunchKernelGGL(persistent_static_32k, dim3(1024), dim3(256), 0, 0, d, npages); });
    run("lds_tiles_like_k_observe", [&] { hipLaunchKernelGGL(lds_tiles_like_k_observe, dim3(1024), dim3(256), tile16 * 16, 0, d, tile16, ntiles); });
    return 0;
}
