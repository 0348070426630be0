// Round-3 write probe, part 2: is the 5.3-vs-6.9 TB/s gap a matter of HOW MANY DISTINCT ADDRESS REGIONS the chip writes at the
// same time (DRAM row-buffer locality), i.e. of the width of the in-flight write window?
//   bin/write_probe3 <MB> [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one 16-B store per thread, one 4-KiB page per block; page = f(blockIdx):
//  MODE 0: identity (address order)   MODE 1: R regions, consecutive blocks go round-robin over the regions (each region is
//  written sequentially)              MODE 2: multiplicative scramble (consecutive blocks land far apart, no sequential runs)
template <int MODE>
__global__ __launch_bounds__(256) void k_page_order(uint4 *dst, long npages, long R, long mult) {
    long b = blockIdx.x, pg;
    if (MODE == 0) pg = b;
    else if (MODE == 1) { const long per = npages / R; pg = (b % R) * per + b / R; if (b >= per * R) pg = b; }
    else pg = (b * mult) % npages;
    dst[pg * 256 + threadIdx.x] = make_uint4((unsigned)b, 2, 3, 4);
}
// persistent, static: generation g, workgroup b -> S consecutive pages at (g * G + b) * S
template <int S>
__global__ __launch_bounds__(256) void k_static_pages(uint4 *dst, long npages) {
    for (long p0 = (long)blockIdx.x * S; p0 < npages; p0 += (long)gridDim.x * S)
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (p0 + s < npages) dst[(p0 + s) * 256 + threadIdx.x] = make_uint4(p0, s, 3, 4);
}
// non-persistent, one workgroup per tile of `tile16` 16-byte chunks, assembled in LDS with k_observe's four barriers, streamed
// out by 192 of the 256 threads; tiles in address order
__global__ __launch_bounds__(256) void k_tile_once(uint4 *dst, int tile16, int ntiles) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, t = blockIdx.x;
    uint4 *out = dst + (size_t)t * tile16;
    for (int i = tid; i < tile16; i += 256) lds[i] = make_uint4(t, i, 3, 4);
    __syncthreads(); __syncthreads(); __syncthreads();
    if (tid < 192) for (int i = tid; i < tile16; i += 192) out[i] = lds[i];
}
// persistent LDS tile writer (k_observe's shape), grid and tile free
__global__ __launch_bounds__(256) void k_tile_persist(uint4 *dst, int tile16, int ntiles) {
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
// 1024-thread workgroups, one per CU: 12 of 16 waves stream a tile out
__global__ __launch_bounds__(1024) void k_tile_persist_1024(uint4 *dst, int tile16, int ntiles) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint4 *out = dst + (size_t)t * tile16;
        __syncthreads();
        for (int i = tid; i < tile16; i += 1024) lds[i] = make_uint4(t, i, 3, 4);
        __syncthreads(); __syncthreads(); __syncthreads();
        if (tid < 768) for (int i = tid; i < tile16; i += 768) out[i] = lds[i];
    }
}

static hipEvent_t ev_a, ev_b;
template <typename F> float timeit(F f, int iters) {
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(ev_a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(ev_b)); CK(hipEventSynchronize(ev_b));
    float ms; CK(hipEventElapsedTime(&ms, ev_a, ev_b));
    return ms * 1e3f / iters;
}

int main(int argc, char **argv) {
    const size_t bytes = ((argc > 1 ? atol(argv[1]) : 642) * 1000000ull) & ~(size_t)4095;
    const int iters = argc > 2 ? atoi(argv[2]) : 30;
    uint4 *d; CK(hipMalloc(&d, bytes + (1 << 20)));
    CK(hipEventCreate(&ev_a)); CK(hipEventCreate(&ev_b));
    const long npages = (long)(bytes / 4096);
    printf("# %zu bytes (%ld pages of 4 KiB), %d launches each\n", bytes, npages, iters);
    char name[128];
    auto rep = [&](const char *nm, float us, double b) { printf("%-52s%8.1f us  %.2f TB/s\n", nm, us, b / us / 1e6); fflush(stdout); };
    rep("page order: identity", timeit([&] { hipLaunchKernelGGL((k_page_order<0>), dim3((unsigned)npages), dim3(256), 0, 0, d, npages, 1L, 1L); }, iters), (double)bytes);
    for (long R : {2L, 8L, 32L, 128L, 512L, 2048L, 8192L}) {
        snprintf(name, sizeof name, "page order: %ld sequential regions round-robin", R);
        rep(name, timeit([&] { hipLaunchKernelGGL((k_page_order<1>), dim3((unsigned)npages), dim3(256), 0, 0, d, npages, R, 1L); }, iters), (double)bytes);
    }
    for (long m : {257L, 4099L, 65537L}) {  // coprime with npages for the sizes used is not required: collisions only lower the byte count slightly
        snprintf(name, sizeof name, "page order: scrambled (b * %ld mod npages)", m);
        rep(name, timeit([&] { hipLaunchKernelGGL((k_page_order<2>), dim3((unsigned)npages), dim3(256), 0, 0, d, npages, 1L, m); }, iters), (double)bytes);
    }
#define D_(S, G) { snprintf(name, sizeof name, "persistent static S=%d pages, grid %d (window %.0f MiB)", S, G, S * G / 256.0); \
    rep(name, timeit([&] { hipLaunchKernelGGL((k_static_pages<S>), dim3(G), dim3(256), 0, 0, d, npages); }, iters), (double)bytes); }
    D_(8, 256) D_(8, 512) D_(8, 1024) D_(8, 2048) D_(4, 512) D_(4, 1024) D_(4, 2048) D_(2, 1024) D_(2, 2048) D_(16, 256) D_(16, 512) D_(32, 256)
    for (int kb : {4, 8, 16, 30}) {
        const int tile16 = kb == 30 ? 1840 : kb * 64, ntiles = (int)(bytes / 16 / tile16);
        const double b = (double)ntiles * tile16 * 16;
        snprintf(name, sizeof name, "LDS tile %d B, one workgroup per tile", tile16 * 16);
        rep(name, timeit([&] { hipLaunchKernelGGL(k_tile_once, dim3(ntiles), dim3(256), tile16 * 16, 0, d, tile16, ntiles); }, iters), b);
        for (int G : {256, 512, 1024, 2048}) {
            if ((size_t)tile16 * 16 * (G / 256) > 160 * 1024) continue;
            snprintf(name, sizeof name, "LDS tile %d B, persistent grid %d (window %.0f MiB)", tile16 * 16, G, tile16 * 16.0 * G / 1048576);
            rep(name, timeit([&] { hipLaunchKernelGGL(k_tile_persist, dim3(G), dim3(256), tile16 * 16, 0, d, tile16, ntiles); }, iters), b);
        }
    }
    for (int kb : {30, 60, 120}) {
        const int tile16 = kb * 64 - (kb == 30 ? 80 : 0), ntiles = (int)(bytes / 16 / tile16);
        snprintf(name, sizeof name, "LDS tile %d B, 1024-thread WGs, grid 256 (window %.0f MiB)", tile16 * 16, tile16 * 16.0 * 256 / 1048576);
        rep(name, timeit([&] { hipLaunchKernelGGL(k_tile_persist_1024, dim3(256), dim3(1024), tile16 * 16, 0, d, tile16, ntiles); }, iters), (double)ntiles * tile16 * 16);
    }
    return 0;
}
