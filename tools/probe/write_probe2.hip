// Round-3 write-pattern probe: WHY do multi-store-per-thread writers top out at ~5.3 TB/s for outputs beyond the 256 MiB
// Infinity Cache while a one-store-per-thread fill in address order reaches ~6.9 TB/s (profiles/r02/write_probe_642MB.txt)?
// Every variant is its own kernel name so that rocprofv3 --pmc passes can tell them apart.
//   hipcc --offload-arch=gfx950 -O3 -o bin/write_probe2 write_probe2.hip ;  bin/write_probe2 <MB> [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// A: non-persistent, block b writes S consecutive 4-KiB pages (one 16-B store per thread per page), blocks in address order
template <int S>
__global__ __launch_bounds__(256) void k_block_contig(uint4 *dst, long npages) {
    const long p0 = (long)blockIdx.x * S;
#pragma unroll
    for (int s = 0; s < S; ++s)
        if (p0 + s < npages) dst[(p0 + s) * 256 + threadIdx.x] = make_uint4(blockIdx.x, s, 3, 4);
}
// A': same, but the wave waits for every store's acknowledgement before issuing the next (at most one store in flight per wave)
template <int S>
__global__ __launch_bounds__(256) void k_block_contig_wait(uint4 *dst, long npages) {
    const long p0 = (long)blockIdx.x * S;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (p0 + s < npages) dst[(p0 + s) * 256 + threadIdx.x] = make_uint4(blockIdx.x, s, 3, 4);
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    }
}
// B: non-persistent, block b writes S pages that lie `nblocks` pages apart (sweep s of ALL blocks covers one contiguous region)
template <int S>
__global__ __launch_bounds__(256) void k_block_strided(uint4 *dst, long npages) {
    const long nb = gridDim.x;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const long pg = (long)s * nb + blockIdx.x;
        if (pg < npages) dst[pg * 256 + threadIdx.x] = make_uint4(blockIdx.x, s, 3, 4);
    }
}
// C: persistent workgroups, pages claimed in address order from ONE atomic counter (S pages per claim)
template <int S>
__global__ __launch_bounds__(256) void k_dyn_pages(uint4 *dst, long npages, unsigned *ctr) {
    __shared__ unsigned sh;
    for (;;) {
        if (threadIdx.x == 0) sh = atomicAdd(ctr, 1u);
        __syncthreads();
        const long p0 = (long)sh * S;
        __syncthreads();
        if (p0 >= npages) return;
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (p0 + s < npages) dst[(p0 + s) * 256 + threadIdx.x] = make_uint4(p0, s, 3, 4);
    }
}
// D: persistent, static assignment: generation g, workgroup b -> S consecutive pages at (g * G + b) * S  (k_observe's order)
template <int S, int WAIT>
__global__ __launch_bounds__(256) void k_static_pages(uint4 *dst, long npages) {
    for (long p0 = (long)blockIdx.x * S; p0 < npages; p0 += (long)gridDim.x * S) {
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (p0 + s < npages) dst[(p0 + s) * 256 + threadIdx.x] = make_uint4(p0, s, 3, 4);
        if (WAIT) __builtin_amdgcn_s_waitcnt(0x0f70);  // the tile's stores acknowledged before the next tile starts
    }
}
// E: persistent, static, but sweep-interleaved: in generation g the G workgroups jointly cover G*S consecutive pages and sweep s
// of workgroup b is page g*G*S + s*G + b (at any moment the chip writes one contiguous G-page region)
template <int S>
__global__ __launch_bounds__(256) void k_static_interleaved(uint4 *dst, long npages) {
    const long G = gridDim.x;
    for (long g0 = 0; g0 < npages; g0 += G * S) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const long pg = g0 + (long)s * G + blockIdx.x;
            if (pg < npages) dst[pg * 256 + threadIdx.x] = make_uint4(pg, s, 3, 4);
        }
    }
}
// F: D through LDS with k_observe's barrier pattern and only 192 storing threads (3 of 4 waves), 7.19-page tiles (29 440 B)
__global__ __launch_bounds__(256) void k_observe_like(uint4 *dst, int tile16, int ntiles) {
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
// G: F, but tiles claimed in address order from an atomic counter (one claim ahead, so the claim's latency is hidden)
__global__ __launch_bounds__(256) void k_observe_like_dyn(uint4 *dst, int tile16, int ntiles, unsigned *ctr) {
    extern __shared__ uint4 lds[];
    __shared__ unsigned nxt[2];
    const int tid = threadIdx.x;
    if (tid == 0) nxt[0] = atomicAdd(ctr, 1u);
    __syncthreads();
    for (int it = 0;; ++it) {
        const int t = (int)nxt[it & 1];
        if (t >= ntiles) return;
        if (tid == 255) nxt[(it + 1) & 1] = atomicAdd(ctr, 1u);
        uint4 *out = dst + (size_t)t * tile16;
        for (int i = tid; i < tile16; i += 256) lds[i] = make_uint4(t, i, 3, 4);
        __syncthreads(); __syncthreads(); __syncthreads();
        if (tid < 192) for (int i = tid; i < tile16; i += 192) out[i] = lds[i];
        __syncthreads();
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
    const char *only = argc > 3 ? argv[3] : "";
    uint4 *d; CK(hipMalloc(&d, bytes + (1 << 20)));
    unsigned *ctr; CK(hipMalloc(&ctr, 4));
    CK(hipEventCreate(&ev_a)); CK(hipEventCreate(&ev_b));
    const long npages = (long)(bytes / 4096);
    printf("# %zu bytes (%ld pages of 4 KiB), %d launches each\n", bytes, npages, iters);
    auto rep = [&](const char *name, float us) { printf("%-44s%8.1f us  %.2f TB/s\n", name, us, bytes / us / 1e6); fflush(stdout); };
#define WANT(n) (!*only || strstr(n, only))
#define A(S) if (WANT("A")) rep("A block_contig S=" #S, timeit([&] { hipLaunchKernelGGL((k_block_contig<S>), dim3((unsigned)((npages + S - 1) / S)), dim3(256), 0, 0, d, npages); }, iters));
    A(1) A(2) A(4) A(8)
#define AW(S) if (WANT("A")) rep("A' block_contig wait-per-store S=" #S, timeit([&] { hipLaunchKernelGGL((k_block_contig_wait<S>), dim3((unsigned)((npages + S - 1) / S)), dim3(256), 0, 0, d, npages); }, iters));
    AW(4) AW(8)
#define B(S) if (WANT("B")) rep("B block_strided S=" #S, timeit([&] { hipLaunchKernelGGL((k_block_strided<S>), dim3((unsigned)((npages + S - 1) / S)), dim3(256), 0, 0, d, npages); }, iters));
    B(2) B(4) B(8)
#define C_(S, G) if (WANT("C")) rep("C dyn_pages S=" #S " grid=" #G, timeit([&] { hipMemsetAsync(ctr, 0, 4, 0); hipLaunchKernelGGL((k_dyn_pages<S>), dim3(G), dim3(256), 0, 0, d, npages, ctr); }, iters));
    C_(1, 2048) C_(2, 2048) C_(8, 2048) C_(8, 1024)
#define D_(S, W, G) if (WANT("D")) rep("D static_pages S=" #S " wait=" #W " grid=" #G, timeit([&] { hipLaunchKernelGGL((k_static_pages<S, W>), dim3(G), dim3(256), 0, 0, d, npages); }, iters));
    D_(1, 0, 2048) D_(1, 0, 1024) D_(8, 0, 1024) D_(8, 1, 1024) D_(8, 1, 2048) D_(2, 1, 2048)
#define E_(S, G) if (WANT("E")) rep("E static_interleaved S=" #S " grid=" #G, timeit([&] { hipLaunchKernelGGL((k_static_interleaved<S>), dim3(G), dim3(256), 0, 0, d, npages); }, iters));
    E_(8, 1024) E_(8, 2048) E_(4, 2048)
    {
        const int tile16 = 1840, ntiles = (int)(bytes / 16 / tile16);
        auto rep2 = [&](const char *name, float us) { printf("%-44s%8.1f us  %.2f TB/s\n", name, us, (double)ntiles * tile16 * 16 / us / 1e6); fflush(stdout); };
        if (WANT("F")) rep2("F observe_like static grid=1024", timeit([&] { hipLaunchKernelGGL(k_observe_like, dim3(1024), dim3(256), tile16 * 16, 0, d, tile16, ntiles); }, iters));
        if (WANT("G")) rep2("G observe_like dyn grid=1024", timeit([&] { hipMemsetAsync(ctr, 0, 4, 0); hipLaunchKernelGGL(k_observe_like_dyn, dim3(1024), dim3(256), tile16 * 16, 0, d, tile16, ntiles, ctr); }, iters));
        if (WANT("G")) rep2("G observe_like dyn grid=1280", timeit([&] { hipMemsetAsync(ctr, 0, 4, 0); hipLaunchKernelGGL(k_observe_like_dyn, dim3(1280), dim3(256), tile16 * 16, 0, d, tile16, ntiles, ctr); }, iters));
    }
    return 0;
}
