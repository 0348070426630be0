// Write-only bandwidth of several store patterns (why does the LDS-staged tile stream-out of k_observe reach ~75 % of a
// plain fill once the output no longer fits the Infinity Cache?).  hipcc --offload-arch=gfx950 -O3 -o write_probe write_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_fill(uint4 *dst, size_t n16) {  // grid-stride, like a library fill
    const uint4 v = make_uint4(1, 2, 3, 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
// persistent workgroups, tile = `tile16` 16-byte chunks, tiles b, b + grid, ...; mode bits: 1 = stage through LDS,
// 2 = barriers like k_observe (4 per tile), 4 = only 192 of 256 threads store
template <int MODE>
__global__ __launch_bounds__(256) void k_tiles(uint4 *dst, int tile16, int ntiles) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x;
    const int nst = (MODE & 4) ? 192 : 256;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint4 *out = dst + (size_t)t * tile16;
        if (MODE & 2) __syncthreads();
        if (MODE & 1) {
            for (int i = tid; i < tile16; i += 256) lds[i] = make_uint4(t, i, 3, 4);
        }
        if (MODE & 2) { __syncthreads(); __syncthreads(); __syncthreads(); }
        else if (MODE & 1) __syncthreads();
        if (tid < nst) {
            if (MODE & 1) for (int i = tid; i < tile16; i += nst) out[i] = lds[i];
            else for (int i = tid; i < tile16; i += nst) out[i] = make_uint4(t, i, 3, 4);
        }
        if ((MODE & 1) && !(MODE & 2)) __syncthreads();
    }
}

// persistent tiles, but every sweep of the workgroup covers ONE 4-KiB-aligned block of the destination (lane = chunk
// within the block), whatever the tile's own alignment; MODE bit 1: through LDS (+ k_observe's barriers)
template <int MODE>
__global__ __launch_bounds__(256) void k_tiles_al(uint4 *dst, int tile16, int ntiles) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long g0 = (long)t * tile16, g1 = g0 + tile16;
        if (MODE & 1) {
            __syncthreads();
            for (int i = tid; i < tile16 + 256; i += 256) lds[i] = make_uint4(t, i, 3, 4);
            __syncthreads(); __syncthreads(); __syncthreads();
        }
        const int ph = (int)(g0 & 255);  // tile placed in LDS at its phase within the 4-KiB block
        for (long blk = g0 >> 8; (blk << 8) < g1; ++blk) {
            const long g = (blk << 8) + tid;
            if (g >= g0 && g < g1) dst[g] = (MODE & 1) ? lds[(int)(g - g0) + ph] : make_uint4(t, (int)g, 3, 4);
        }
    }
}

// persistent tiles; every WAVE streams through its own contiguous quarter of the tile (consecutive stores of a wave are
// adjacent in memory), MODE bit 1: through LDS + barriers
template <int MODE>
__global__ __launch_bounds__(256) void k_tiles_wc(uint4 *dst, int tile16, int ntiles) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int span = (tile16 + 3) / 4;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint4 *out = dst + (size_t)t * tile16;
        if (MODE & 1) {
            __syncthreads();
            for (int i = tid; i < tile16; i += 256) lds[i] = make_uint4(t, i, 3, 4);
            __syncthreads(); __syncthreads(); __syncthreads();
        }
        const int lo = wave * span, hi = min(lo + span, tile16);
        for (int i = lo + lane; i < hi; i += 64) out[i] = (MODE & 1) ? lds[i] : make_uint4(t, i, 3, 4);
    }
}

template <typename F> float timeit(F f, int iters = 30) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / iters;
}

int main(int argc, char **argv) {
    const size_t bytes = (argc > 1 ? atol(argv[1]) : 642) * 1000000ull;
    const int tile16 = argc > 2 ? atoi(argv[2]) : 1840;  // 29 440 bytes
    const int per_cu = argc > 3 ? atoi(argv[3]) : 4;
    uint4 *d; CK(hipMalloc(&d, bytes + 65536));
    const size_t n16 = bytes / 16;
    const int ntiles = (int)(n16 / tile16);
    const size_t lds = (size_t)tile16 * 16;
    float us = timeit([&] { hipLaunchKernelGGL(k_fill, dim3(256 * 8), dim3(256), 0, 0, d, n16); });
    printf("fill grid-stride          %8.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
#define RUN(M, name) { us = timeit([&] { hipLaunchKernelGGL((k_tiles<M>), dim3(256 * per_cu), dim3(256), (M & 1) ? lds : 0, 0, d, tile16, ntiles); }); \
    printf("%-26s%8.1f us  %.2f TB/s\n", name, us, (double)ntiles * tile16 * 16 / us / 1e6); }
    // one workgroup per tile, in address order (non-persistent)
#define RUN1(M, name) { us = timeit([&] { hipLaunchKernelGGL((k_tiles<M>), dim3(ntiles), dim3(256), (M & 1) ? lds : 0, 0, d, tile16, ntiles); }); \
    printf("%-26s%8.1f us  %.2f TB/s\n", name, us, (double)ntiles * tile16 * 16 / us / 1e6); }
    RUN1(0, "1 WG/tile regs");
    RUN1(3, "1 WG/tile lds+barriers");
    {   // library-style fill: one 256-thread block per 4 KiB, each thread one 16-byte store
        const int nb = (int)(n16 / 256);
        us = timeit([&] { hipLaunchKernelGGL(k_fill, dim3(nb), dim3(256), 0, 0, d, n16); });
        printf("fill 1 block per 4 KiB    %8.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
        us = timeit([&] { hipLaunchKernelGGL(k_fill, dim3(nb / 4), dim3(256), 0, 0, d, n16); });
        printf("fill 1 block per 16 KiB   %8.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    }
#define RUNA(M, name) { us = timeit([&] { hipLaunchKernelGGL((k_tiles_al<M>), dim3(256 * per_cu), dim3(256), (M & 1) ? lds + 8192 : 0, 0, d, tile16, ntiles); }); \
    printf("%-26s%8.1f us  %.2f TB/s\n", name, us, (double)ntiles * tile16 * 16 / us / 1e6); }
    RUNA(0, "tiles regs 4KiB sweeps");
    RUNA(1, "tiles lds 4KiB sweeps");
#define RUNW(M, name) { us = timeit([&] { hipLaunchKernelGGL((k_tiles_wc<M>), dim3(256 * per_cu), dim3(256), (M & 1) ? lds : 0, 0, d, tile16, ntiles); }); \
    printf("%-26s%8.1f us  %.2f TB/s\n", name, us, (double)ntiles * tile16 * 16 / us / 1e6); }
    RUNW(0, "tiles regs wave-contig");
    RUNW(1, "tiles lds wave-contig");
    RUN(0, "tiles regs");
    RUN(4, "tiles regs 192thr");
    RUN(1, "tiles lds");
    RUN(3, "tiles lds+barriers");
    RUN(7, "tiles lds+barriers 192thr");
    return 0;
}
