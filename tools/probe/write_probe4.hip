// Round-3 write probe, part 3: LDS-staged tile writers -- does page (4 KiB) alignment of the tile boundaries matter, and does
// claiming tiles IN ADDRESS ORDER (per-XCD counters, 8 sequential regions) recover the in-order dispatcher's bandwidth?
//   bin/write_probe4 <MB> [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// BAR: k_observe's four barriers per tile; NST: storing threads (192 = three of four waves)
template <int BAR, int NST>
__global__ __launch_bounds__(256) void k_once(uint4 *dst, int tile16, int ntiles) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, t = blockIdx.x;
    uint4 *out = dst + (size_t)t * tile16;
    for (int i = tid; i < tile16; i += 256) lds[i] = make_uint4(t, i, 3, 4);
    __syncthreads();
    if (BAR) { __syncthreads(); __syncthreads(); }
    if (tid < NST) for (int i = tid; i < tile16; i += NST) out[i] = lds[i];
}
template <int NST>
__global__ __launch_bounds__(256) void k_static(uint4 *dst, int tile16, int ntiles) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint4 *out = dst + (size_t)t * tile16;
        __syncthreads();
        for (int i = tid; i < tile16; i += 256) lds[i] = make_uint4(t, i, 3, 4);
        __syncthreads(); __syncthreads(); __syncthreads();
        if (tid < NST) for (int i = tid; i < tile16; i += NST) out[i] = lds[i];
    }
}
// tiles claimed in address order: shard s = blockIdx & (NSH-1) owns the contiguous tile range [s*per, (s+1)*per) and a counter on
// a cache line of its own; a workgroup claims one tile AHEAD (the claim's latency overlaps the current tile)
template <int NST, int NSH>
__global__ __launch_bounds__(256) void k_dynamic(uint4 *dst, int tile16, int ntiles, unsigned *ctr) {
    extern __shared__ uint4 lds[];
    __shared__ unsigned nxt[2];
    const int tid = threadIdx.x;
    const int sh = blockIdx.x & (NSH - 1);
    const int per = (ntiles + NSH - 1) / NSH;
    const int lo = sh * per, hi = min(ntiles, lo + per);
    unsigned *c = ctr + sh * 32;
    if (tid == 255) nxt[0] = atomicAdd(c, 1u);
    __syncthreads();
    for (int it = 0;; ++it) {
        const int t = lo + (int)nxt[it & 1];
        if (t >= hi) return;
        if (tid == 255) nxt[(it + 1) & 1] = atomicAdd(c, 1u);
        uint4 *out = dst + (size_t)t * tile16;
        for (int i = tid; i < tile16; i += 256) lds[i] = make_uint4(t, i, 3, 4);
        __syncthreads(); __syncthreads(); __syncthreads();
        if (tid < NST) for (int i = tid; i < tile16; i += NST) out[i] = lds[i];
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
    uint4 *d; CK(hipMalloc(&d, bytes + (1 << 20)));
    unsigned *ctr; CK(hipMalloc(&ctr, 64 * 128));
    CK(hipEventCreate(&ev_a)); CK(hipEventCreate(&ev_b));
    printf("# %zu bytes, %d launches each\n", bytes, iters);
    char name[160];
    for (int tb : {28672, 29440, 32768, 31360, 15680, 16384, 8192, 7840}) {
        const int tile16 = tb / 16, ntiles = (int)(bytes / tb);
        const double b = (double)ntiles * tb;
        auto rep = [&](const char *nm, float us) { printf("tile %5d B (%s)  %-44s%8.1f us  %.2f TB/s\n", tb, tb % 4096 ? "unaligned" : "page-aligned", nm, us, b / us / 1e6); fflush(stdout); };
        rep("one WG per tile, 256 store, no extra barriers", timeit([&] { hipLaunchKernelGGL((k_once<0, 256>), dim3(ntiles), dim3(256), tb, 0, d, tile16, ntiles); }, iters));
        rep("one WG per tile, 192 store, barriers", timeit([&] { hipLaunchKernelGGL((k_once<1, 192>), dim3(ntiles), dim3(256), tb, 0, d, tile16, ntiles); }, iters));
        for (int G : {1024, 2048}) {
            if ((size_t)tb * (G / 256) > 160 * 1024) continue;
            snprintf(name, sizeof name, "persistent static, grid %d", G);
            rep(name, timeit([&] { hipLaunchKernelGGL((k_static<192>), dim3(G), dim3(256), tb, 0, d, tile16, ntiles); }, iters));
            snprintf(name, sizeof name, "persistent in-order claims, 8 shards, grid %d", G);
            rep(name, timeit([&] { hipMemsetAsync(ctr, 0, 64 * 128, 0); hipLaunchKernelGGL((k_dynamic<192, 8>), dim3(G), dim3(256), tb, 0, d, tile16, ntiles, ctr); }, iters));
            snprintf(name, sizeof name, "persistent in-order claims, 4 shards, grid %d", G);
            rep(name, timeit([&] { hipMemsetAsync(ctr, 0, 64 * 128, 0); hipLaunchKernelGGL((k_dynamic<192, 4>), dim3(G), dim3(256), tb, 0, d, tile16, ntiles, ctr); }, iters));
            snprintf(name, sizeof name, "persistent in-order claims, 16 shards, grid %d", G);
            rep(name, timeit([&] { hipMemsetAsync(ctr, 0, 64 * 128, 0); hipLaunchKernelGGL((k_dynamic<192, 16>), dim3(G), dim3(256), tb, 0, d, tile16, ntiles, ctr); }, iters));
        }
    }
    return 0;
}
