// meda_vec.hip -- C ABI (include/meda_vec.h) of the vectorised MEDA environment, the LDS-staged
// observation kernel, map/task/state accessors and the dispatch to the per-N kernels built from
// meda_vec_n.hip.  Device code of the transition: meda_kernels.h.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "meda_kernels.h"

using namespace medak;

namespace {

// diagnostic build only (-DMEDA_ABLATE, tools/ab_meda_obs.py): phases of the observation kernel switched off by a
// device-side bit mask, to see what each costs.  Never defined for the product library.
#ifdef MEDA_ABLATE
__device__ int g_ablate;
#define ABL(bit) (g_ablate & (bit))
#else
#define ABL(bit) 0
#endif

// ---- MEDAEnv.getOneObs (meda.py:613-674) and MEDAEnv_v0_2.getOneObs (meda.py:850-897), LDS-staged ----------
// Persistent workgroups (the grid covers the CUs a few times over; a workgroup walks tiles blockIdx.x,
// blockIdx.x + gridDim.x, ...).  A tile is T consecutive chips = T*n rows of obs_len bytes, built in LDS at the
// same 16-byte phase as its destination in HBM (shift = global offset & 15), so its body streams out with
// aligned 16-byte stores whatever T, n and fov are.
// Roles, as in dmfbk::k_observe: the LAST wave is the loader -- it alone reads global memory (droplet words, one
// chip per lane; the refresh mask) and never stores; the other waves stream finished tiles out and never load
// (gfx950 counts a wave's loads and stores in one in-order counter).  The droplet words run two tiles ahead:
// while tile k is filled, those of tile k+1 sit in the loader's registers; while tile k streams out they are
// written to the second word buffer and tile k+2 is requested.
// Rows are filled by waves, two rows per wave pass (lanes 0-24 and 32-56 each own one cell of a 5x5 footprint);
// a row is always written by ONE wave, and LDS operations of one wave complete in order, so for overlapping
// footprints the higher droplet index wins exactly as in the reference's sequential loops.
constexpr int kObsBlock = 256;
constexpr int kObsWork = kObsBlock - kWave;  // threads that stream tiles out

__host__ __device__ inline size_t obs_tile_area(int T, int row_bytes) { return ((size_t)T * row_bytes + 16 + 15) & ~(size_t)15; }
__host__ __device__ inline size_t obs_lds_bytes(int T, int n, int row_bytes) {
    return obs_tile_area(T, row_bytes) + 2 * (size_t)T * n * 4 + 2 * (((size_t)T + 15) & ~(size_t)15) + 512;
}

// One 5x5 footprint (value val, centre (x0, y0) in window coordinates) into a window layer [y][x]: every cell is
// clamped into the window.  For a footprint that touches the window this writes exactly its in-window cells (a cell
// outside lands on the window edge, which the footprint then covers); for a goal it is np.clip's smear (meda.py:665,
// 872).  The caller predicates footprints that miss the window entirely.
__device__ __forceinline__ void put5x5(int8_t *layer, int fov, int x0, int y0, int8_t val) {
    int xs[2 * kR + 1], ys[2 * kR + 1];
#pragma unroll
    for (int k = 0; k <= 2 * kR; ++k) {
        xs[k] = min(max(x0 + k - kR, 0), fov - 1);
        ys[k] = min(max(y0 + k - kR, 0), fov - 1) * fov;
    }
#pragma unroll
    for (int ky = 0; ky <= 2 * kR; ++ky)
#pragma unroll
        for (int kx = 0; kx <= 2 * kR; ++kx) layer[ys[ky] + xs[kx]] = val;
}
__device__ __forceinline__ bool touches(int x0, int y0, int fov) {  // some footprint cell inside the window
    return x0 + kR >= 0 && x0 - kR <= fov - 1 && y0 + kR >= 0 && y0 - kR <= fov - 1;
}

// Fill of one zeroed LDS tile (all waves of the workgroup): words = [tv*n] droplet words.  An item is one LAYER of
// one observation row and belongs to one lane, which writes that layer's droplets in the reference's order (later
// writes win, as in its sequential loops); waves are layer-uniform.  The v0_2 boundary bands are a phase of their
// own, one lane per window line, before a barrier.
__device__ __forceinline__ void fill_rows(const MCfg &c, int8_t *tile, const uint32_t *words, const int8_t *zoom, int tv, int tid) {
    const int n = c.n, fov = c.fov, ff = c.ff, hf = fov / 2;
    const int wave = tid / kWave, lane = tid % kWave;
    const int rows = tv * n;
    constexpr int kWaves = kObsBlock / kWave;
    if (c.version == 2) {
        // ---- MEDAEnv_v0_2.getOneObs (meda.py:850-897)
        // layer 2: boundary bands with the reference's axis mix-up (meda.py:880-890).  A bit per column, spread to one
        // byte per bit four columns at a time and OR-ed into the zeroed tile (lines are fov bytes long, so neighbouring
        // lanes share words: atomic OR)
        if (!ABL(2))
        for (int it = tid; it < rows * fov; it += kObsBlock) {
            const int r = it / fov, wr = it - r * fov;
            const uint32_t wa = words[r];
            const int cx = wa & 0xff, cy = (wa >> 8) & 0xff;
            const int left = hf - cx, right = hf - (c.W - 1 - cx);
            const int up = hf - cy, down = hf - (c.L - 1 - cy);
            int c0 = 0, c1 = 0;
            if (up > 0) { c0 = 0; c1 = up < fov ? up : fov; }
            else if (down > 0) { c0 = fov - down < 0 ? 0 : fov - down; c1 = fov; }
            const bool full = left > 0 ? (wr < left) : (right > 0 ? (wr >= fov - right) : false);
            int8_t *dst = tile + (size_t)r * c.obs_len + 2 * ff + wr * fov;
            if (fov > 28) {  // wide windows: plain byte stores
                const int b0 = full ? 0 : c0, b1 = full ? fov : c1;
                for (int cc = b0; cc < b1; ++cc) dst[cc] = 1;
                continue;
            }
            const uint32_t colbits = full ? ((1u << fov) - 1u) : ((1u << c1) - (1u << c0));
            if (colbits) {
                const int ph = (int)((uintptr_t)dst & 3);
                uint32_t *w = (uint32_t *)(dst - ph);
                const unsigned long long bits = (unsigned long long)colbits << ph;
                for (int q = 0; q * 4 < fov + 3; ++q) {
                    const uint32_t v = ((uint32_t)((bits >> (4 * q)) & 15u) * 0x00204081u) & 0x01010101u;
                    if (v) atomicOr(w + q, v);
                }
            }
        }
        __syncthreads();  // the byte stores below may share a word with a band's atomic
        // waves 0,2: layer 0 + direction bytes; waves 1,3: layer 1
        const int layer = wave & 1;
        for (int r = (wave >> 1) * kWave + lane; r < rows; r += (kWaves / 2) * kWave) {
            const int s = r / n, a = r - s * n;
            const uint32_t wa = words[r];
            const int cx = wa & 0xff, cy = (wa >> 8) & 0xff, gxa = (wa >> 16) & 0xff, gya = wa >> 24;
            const int ox = cx - hf, oy = cy - hf;
            int8_t *row = tile + (size_t)r * c.obs_len;
            if (layer == 0) {
                if (!ABL(4))
                for (int j = 0; j < n; ++j) {  // every droplet, ascending index
                    const uint32_t wj = words[s * n + j];
                    const int x0 = (int)(wj & 0xff) - ox, y0 = (int)((wj >> 8) & 0xff) - oy;
                    if (touches(x0, y0, fov)) put5x5(row, fov, x0, y0, (int8_t)(j + 1));
                }
                row[3 * ff] = zoom[gya - cy + 128];
                row[3 * ff + 1] = zoom[256 + gxa - cx + 128];
                continue;
            }
            // members of the `observed` set: droplets with at least one footprint cell inside the window
            uint32_t obs_mask = 0;
            for (int j = 0; j < n; ++j) {
                const uint32_t wj = words[s * n + j];
                obs_mask |= (uint32_t)touches((int)(wj & 0xff) - ox, (int)((wj >> 8) & 0xff) - oy, fov) << j;
            }
            if (ABL(1)) obs_mask = 0;
            // iteration order of the CPython set (see oracle/meda_oracle.c): ascending once it has had 5
            // members (table resized to 32 slots), else slot order of the 8-slot table
            unsigned long long order = 0;  // 4-bit entries
            int cnt = 0;
            if (__popc(obs_mask) >= 5) {
                for (int j = 0; j < n; ++j)
                    if ((obs_mask >> j) & 1) { order |= (unsigned long long)j << (4 * cnt); ++cnt; }
            } else {
                unsigned long long slots = 0;  // 8 x (valid bit | 4-bit value)
                for (int j = 0; j < n; ++j)
                    if ((obs_mask >> j) & 1) {
                        int i = j & 7;
                        while ((slots >> (5 * i)) & 16) i = (i * 5 + 1) & 7;
                        slots |= (unsigned long long)(16 | j) << (5 * i);
                    }
                for (int i = 0; i < 8; ++i)
                    if ((slots >> (5 * i)) & 16) { order |= ((slots >> (5 * i)) & 15) << (4 * cnt); ++cnt; }
            }
            for (int tt = 0; tt < cnt; ++tt) {  // clipped goals of the observed OTHER droplets, in set order
                const int j = (int)((order >> (4 * tt)) & 15);
                if (j == a) continue;
                const uint32_t wj = words[s * n + j];
                put5x5(row + ff, fov, (int)((wj >> 16) & 0xff) - ox, (int)(wj >> 24) - oy, (int8_t)(j + 1));
            }
        }
        return;
    }
    // ---- MEDAEnv.getOneObs (meda.py:613-674): wave = layer
    const int layer = wave;
    for (int r = lane; r < rows; r += kWave) {
        const int s = r / n, a = r - s * n;
        const uint32_t wa = words[r];
        const int cx = wa & 0xff, cy = (wa >> 8) & 0xff, gxa = (wa >> 16) & 0xff, gya = wa >> 24;
        const int ox = cx - hf, oy = cy - hf;
        int8_t *row = tile + (size_t)r * c.obs_len;
        if (layer == 0) {  // own footprint, direction
            put5x5(row, fov, hf, hf, (int8_t)(a + 1));
            row[4 * ff] = (int8_t)(gxa - cx);
            row[4 * ff + 1] = (int8_t)(gya - cy);
        } else if (layer == 1) {  // own goal, in-window cells only
            if (touches(gxa - ox, gya - oy, fov)) put5x5(row + ff, fov, gxa - ox, gya - oy, (int8_t)(a + 1));
        } else {  // 2: the OTHER droplets (in-window cells); 3: their goals, smeared onto the window edge; ascending index
            for (int j = 0; j < n; ++j) {
                if (j == a) continue;
                const uint32_t wj = words[s * n + j];
                if (layer == 2) {
                    const int x0 = (int)(wj & 0xff) - ox, y0 = (int)((wj >> 8) & 0xff) - oy;
                    if (touches(x0, y0, fov)) put5x5(row + 2 * ff, fov, x0, y0, (int8_t)(j + 1));
                } else {
                    put5x5(row + 3 * ff, fov, (int)((wj >> 16) & 0xff) - ox, (int)(wj >> 24) - oy, (int8_t)(j + 1));
                }
            }
        }
    }
}

template <int NW>  // droplet words the loader keeps per chip: n <= NW
__global__ __launch_bounds__(kObsBlock) void k_meda_observe(MCfg c, MPtrs p, const uint8_t *mask, int8_t *gobs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int T = c.T, E = c.E, n = c.n;
    const int tid = threadIdx.x;
    const bool loader = tid >= kObsWork;
    const int ltid = tid - kObsWork;
    const int ntiles = (E + T - 1) / T;
    const int row_bytes = n * c.obs_len;
    uint32_t *const wbuf = (uint32_t *)(smem + obs_tile_area(T, row_bytes));  // [2][T*n] droplet words
    uint8_t *const fbuf = (uint8_t *)(wbuf + 2 * (size_t)T * n);               // [2][align16(T)] refresh flags
    const int fstride = (T + 15) & ~15;
    const int8_t *const zoom = (const int8_t *)(fbuf + 2 * fstride);           // LDS copy of MPtrs::zoom (v0_2 only)
    uint32_t rec[NW];
    auto prefetch = [&](int tile) {  // unconditional loads from clamped (always valid) addresses: nothing to wait for here
        const int base = tile < ntiles ? tile * T : 0;
        const int chip = min(base + ltid, E - 1);
#pragma unroll
        for (int w = 0; w < NW; ++w) rec[w] = p.st[(size_t)min(w, n - 1) * E + chip];
    };
    auto unpack = [&](int tile, uint32_t *dst) {
        if (tile >= ntiles || ltid >= min(T, E - tile * T)) return;
#pragma unroll
        for (int w = 0; w < NW; ++w)
            if (w < n) dst[ltid * n + w] = rec[w];
    };
    int tile = blockIdx.x, buf = 0;
    if (loader) {
        prefetch(tile);
        if (c.version == 2)
            for (int i = ltid; i < 128; i += kWave) ((uint32_t *)zoom)[i] = ((const uint32_t *)p.zoom)[i];
        unpack(tile, wbuf);
        prefetch(tile + gridDim.x);
    }
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const int tile_base = tile * T;
        const int tv = min(T, E - tile_base);
        const size_t g0 = (size_t)tile_base * row_bytes;
        const int shift = (int)(((uintptr_t)gobs + g0) & 15);
        const int bytes = tv * row_bytes;
        int8_t *const tile_lds = (int8_t *)smem + shift;
        const uint32_t *const words = wbuf + (size_t)buf * T * n;
        uint8_t *const flag = fbuf + buf * fstride;
        // this barrier publishes the words written during the previous tile's stream-out and separates that
        // stream-out (LDS reads) from the zero fill below; the loader's lanes flag and count the chips to refresh
        bool refresh = false;
        if (loader && ltid < tv) {
            refresh = !mask || mask[tile_base + ltid] != 0;
            flag[ltid] = (uint8_t)refresh;
        }
        const int cnt = __syncthreads_count(refresh);
        if (cnt != 0) {  // (uniform) something to refresh in this tile
            if (!ABL(16)) {
                uint4 *z = (uint4 *)smem;
                const uint4 zero = make_uint4(0, 0, 0, 0);
                for (int i = tid; i < (shift + bytes + 15) / 16; i += kObsBlock) z[i] = zero;
            }
            __syncthreads();
            fill_rows(c, tile_lds, words, zoom, tv, tid);
            __syncthreads();
        }
        if (loader) {
            unpack(tile + gridDim.x, wbuf + (size_t)(buf ^ 1) * T * n);
            prefetch(tile + 2 * gridDim.x);
        } else if (ABL(8)) {
        } else if (cnt == tv) {
            // head (< 16 bytes) and tail by bytes, body 16 bytes per lane: four LDS reads in flight, then four stores
            const int head = (16 - shift) & 15;
            const int hb = head < bytes ? head : bytes;
            for (int b = tid; b < hb; b += kObsWork) gobs[g0 + b] = tile_lds[b];
            const int n16 = (bytes - hb) >> 4;
            const uint4 *src = (const uint4 *)(tile_lds + hb);
            uint4 *dst = (uint4 *)(gobs + g0 + hb);
            int i = tid;
            for (; i + 3 * kObsWork < n16; i += 4 * kObsWork) {
                const uint4 v0 = src[i], v1 = src[i + kObsWork], v2 = src[i + 2 * kObsWork], v3 = src[i + 3 * kObsWork];
                dst[i] = v0; dst[i + kObsWork] = v1; dst[i + 2 * kObsWork] = v2; dst[i + 3 * kObsWork] = v3;
            }
            for (; i < n16; i += kObsWork) dst[i] = src[i];
            for (int b = hb + (n16 << 4) + tid; b < bytes; b += kObsWork) gobs[g0 + b] = tile_lds[b];
        } else if (cnt != 0) {
            for (int s = 0; s < tv; ++s)
                if (flag[s])
                    for (int b = tid; b < row_bytes; b += kObsWork) gobs[g0 + (size_t)s * row_bytes + b] = tile_lds[(size_t)s * row_bytes + b];
        }
    }
}

// MEDAEnv.updateHealth (meda.py:600-605) for the chips flagged by the last reset / auto-reset
__global__ __launch_bounds__(kBlock) void k_meda_update_health(MCfg c, MPtrs p) {
    const int e = blockIdx.x * (kBlock / kWave) + (int)(threadIdx.x / kWave);
    if (e >= c.E || !p.reset_flag[e]) return;
    const int lane = threadIdx.x % kWave;
    const int cells = c.W * c.L;
    const size_t base = (size_t)e * cells;
    for (int i = lane; i < cells; i += kWave)
        if (p.usage[base + i] > 50) { p.health[base + i] = p.health[base + i] * p.degrade[base + i]; p.usage[base + i] = 0; }
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) p.reset_flag[e] = 0;
}

// MEDAEnv.__init__ maps (meda.py:494-504)
__global__ void k_meda_init_maps(MCfg c, MPtrs p) {
    const size_t total = (size_t)c.E * c.W * c.L;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cells = c.W * c.L;
    const int e = (int)(i / cells), cell = (int)(i % cells);
    p.health[i] = 1.0;
    p.usage[i] = 0;
    double v = 1.0;
    if (c.b_degrade) {
        uint32_t w[4];
        philox(c.k0, c.k1, c.env_id0 + (uint32_t)e, 0u, (uint32_t)cell, STREAM_DEGRADE << 8, w);
        const double d = u53(w[0], w[1]) * 0.4 + 0.6;
        v = (u53(w[2], w[3]) < c.per_healthy) ? 1.0 : d;
    }
    p.degrade[i] = v;
}

__global__ void k_meda_set_task(MCfg c, MPtrs p, const int32_t *starts, const int32_t *ends) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.E) return;
    const int n = c.n, E = c.E;
    for (int i = 0; i < n; ++i) {
        const size_t k = ((size_t)e * n + i) * 2;
        const uint32_t s = (uint32_t)(starts[k] & 0xff) | ((uint32_t)(starts[k + 1] & 0xff) << 8);
        const uint32_t g = (uint32_t)(ends[k] & 0xff) | ((uint32_t)(ends[k + 1] & 0xff) << 8);
        p.starts[(size_t)i * E + e] = s;
        p.st[(size_t)i * E + e] = s | (g << 16);
    }
    p.st[(size_t)n * E + e] = 0;
    p.st[(size_t)(n + 1) * E + e] = 0;
}
__global__ void k_meda_get_task(MCfg c, MPtrs p, int32_t *starts, int32_t *ends) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.E) return;
    const int n = c.n, E = c.E;
    for (int i = 0; i < n; ++i) {
        const size_t k = ((size_t)e * n + i) * 2;
        const uint32_t s = p.starts[(size_t)i * E + e], w = p.st[(size_t)i * E + e];
        if (starts) { starts[k] = s & 0xff; starts[k + 1] = (s >> 8) & 0xff; }
        if (ends) { ends[k] = (w >> 16) & 0xff; ends[k + 1] = w >> 24; }
    }
}
__global__ void k_meda_get_state(MCfg c, MPtrs p, int32_t *pos, uint8_t *status, int32_t *step_count, uint8_t *failed) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.E) return;
    const int n = c.n, E = c.E;
    const uint32_t st = p.st[(size_t)n * E + e], sw = p.st[(size_t)(n + 1) * E + e];
    for (int i = 0; i < n; ++i) {
        const uint32_t w = p.st[(size_t)i * E + e];
        if (pos) { pos[((size_t)e * n + i) * 2] = w & 0xff; pos[((size_t)e * n + i) * 2 + 1] = (w >> 8) & 0xff; }
        if (status) status[(size_t)e * n + i] = (st >> i) & 1;
    }
    if (step_count) step_count[e] = sw & 0xffff;
    if (failed) failed[e] = (sw >> 16) != 0;
}
__global__ void k_meda_get_map(size_t total, const double *health, const double *degrade, const uint16_t *usage, int which,
                               double *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    out[i] = which == MEDA_MAP_HEALTH ? health[i] : which == MEDA_MAP_DEGRADE ? degrade[i] : (double)usage[i];
}
__global__ void k_meda_set_map(size_t total, double *health, double *degrade, uint16_t *usage, int which, const double *in) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (which == MEDA_MAP_HEALTH) health[i] = in[i];
    else if (which == MEDA_MAP_DEGRADE) degrade[i] = in[i];
    else usage[i] = (uint16_t)in[i];
}

thread_local int g_last_hip = 0;
inline int hip_fail(hipError_t e, const char *what, int line) {
    g_last_hip = (int)e;
    if (getenv("DMFB_VEC_DEBUG")) fprintf(stderr, "meda_vec: %s failed at line %d: %s (%d)\n", what, line, hipGetErrorString(e), (int)e);
    return MEDA_ERR_HIP;
}
#define HIP_TRY(expr)                                                 \
    do {                                                              \
        hipError_t _e = (expr);                                       \
        if (_e != hipSuccess) return hip_fail(_e, #expr, __LINE__);   \
    } while (0)
#define LAUNCH(kernel, grid, block, lds, stream, ...)                  \
    do {                                                               \
        (void)hipGetLastError();                                       \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__); \
        HIP_TRY(hipGetLastError());                                    \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) { ok = false; hip_fail(e, "hipGetDevice", __LINE__); prev = -1; return; }
        if (prev != dev) {
            e = hipSetDevice(dev);
            if (e != hipSuccess) { ok = false; hip_fail(e, "hipSetDevice", __LINE__); }
        } else {
            prev = -1;
        }
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

struct meda_vec {
    meda_vec_config cfg;
    MCfg dc;
    MPtrs dp;
    size_t bytes = 0;
    size_t obs_lds = 0;
    int obs_grid = 1;    // persistent grid of the observation kernel
    int n_cu = 256;
    int8_t zoom_host[512];
    int8_t *zoom_dev = nullptr;
    // meda_vec_observe_timing: event pairs that receive the dispatch time stamps of the observation kernel
    static constexpr int kTimed = 256;
    hipEvent_t ev[2 * kTimed] = {};
    int timing = 0, timed = 0;
};

namespace {

template <int N> int step_n(meda_vec *h, const MStepArgs &a, hipStream_t s) { HIP_TRY(launch_meda_step_n<N>(h->dc, h->dp, a, s)); return MEDA_OK; }
template <int N> int reset_n(meda_vec *h, const uint8_t *m, int mode, hipStream_t s) { HIP_TRY(launch_meda_reset_n<N>(h->dc, h->dp, m, mode, s)); return MEDA_OK; }

#define DISPATCH_N(n, FN, ...)                                  \
    switch (n) {                                                \
    case 1: return FN<1>(__VA_ARGS__);  case 2: return FN<2>(__VA_ARGS__);   \
    case 3: return FN<3>(__VA_ARGS__);  case 4: return FN<4>(__VA_ARGS__);   \
    case 5: return FN<5>(__VA_ARGS__);  case 6: return FN<6>(__VA_ARGS__);   \
    case 7: return FN<7>(__VA_ARGS__);  case 8: return FN<8>(__VA_ARGS__);   \
    case 9: return FN<9>(__VA_ARGS__);  case 10: return FN<10>(__VA_ARGS__); \
    case 11: return FN<11>(__VA_ARGS__); case 12: return FN<12>(__VA_ARGS__); \
    case 13: return FN<13>(__VA_ARGS__); case 14: return FN<14>(__VA_ARGS__); \
    case 15: return FN<15>(__VA_ARGS__); case 16: return FN<16>(__VA_ARGS__); \
    default: return MEDA_ERR_UNSUPPORTED;                       \
    }
int launch_step(meda_vec *h, const MStepArgs &a, hipStream_t s) { DISPATCH_N(h->cfg.n_agents, step_n, h, a, s) }
int launch_reset(meda_vec *h, const uint8_t *m, int mode, hipStream_t s) { DISPATCH_N(h->cfg.n_agents, reset_n, h, m, mode, s) }

int launch_observe(const meda_vec *h, const uint8_t *mask, int8_t *obs, hipStream_t s) {
    const dim3 grid(h->obs_grid), block(kObsBlock);
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (h->timing && h->timed < meda_vec::kTimed) {
        meda_vec *hm = const_cast<meda_vec *>(h);
        t0 = hm->ev[2 * h->timed]; t1 = hm->ev[2 * h->timed + 1];
        hm->timed += 1;
    }
    (void)hipGetLastError();
    // hipExtLaunchKernelGGL: the events receive the dispatch's own start/end time stamps (nullptr = plain launch)
    if (h->dc.n <= 4) hipExtLaunchKernelGGL(k_meda_observe<4>, grid, block, h->obs_lds, s, t0, t1, 0, h->dc, h->dp, mask, obs);
    else if (h->dc.n <= 8) hipExtLaunchKernelGGL(k_meda_observe<8>, grid, block, h->obs_lds, s, t0, t1, 0, h->dc, h->dp, mask, obs);
    else hipExtLaunchKernelGGL(k_meda_observe<MEDA_MAX_AGENTS>, grid, block, h->obs_lds, s, t0, t1, 0, h->dc, h->dp, mask, obs);
    HIP_TRY(hipGetLastError());
    return MEDA_OK;
}
int launch_update_health(const meda_vec *h, hipStream_t s) {
    if (!h->dp.health) return MEDA_OK;
    if (!h->cfg.b_degrade) {  // updateHealth returns early (meda.py:601-602): just clear the flags
        HIP_TRY(hipMemsetAsync(h->dp.reset_flag, 0, (size_t)h->cfg.n_envs, s));
        return MEDA_OK;
    }
    const int per = kBlock / kWave;
    LAUNCH(k_meda_update_health, dim3((h->cfg.n_envs + per - 1) / per), dim3(kBlock), 0, s, h->dc, h->dp);
    return MEDA_OK;
}

}  // namespace

extern "C" {

int meda_vec_check_config(const meda_vec_config *c) {
    if (!c) return MEDA_ERR_BAD_ARG;
    if (c->width <= 0 || c->length <= 0) return MEDA_ERR_BAD_SIZE;
    if (c->n_agents <= 0) return MEDA_ERR_NO_AGENTS;
    if (c->n_agents > (c->width / 15) * (c->length / 15)) return MEDA_ERR_TOO_MANY_DROPLETS;
    if (c->n_agents > MEDA_MAX_AGENTS || c->width > MEDA_MAX_DIM || c->length > MEDA_MAX_DIM || c->fov < 1) return MEDA_ERR_UNSUPPORTED;
    if (obs_lds_bytes(1, c->n_agents, c->n_agents * (4 * c->fov * c->fov + 2)) > 60 * 1024) return MEDA_ERR_UNSUPPORTED;  // one chip's rows must fit the LDS tile
    if (c->obs_version != 0 && c->obs_version != 2) return MEDA_ERR_UNSUPPORTED;
    if (c->n_envs <= 0) return MEDA_ERR_BAD_ARG;
    return MEDA_OK;
}

int meda_vec_create(const meda_vec_config *cfg, void *stream, meda_vec **out) {
    if (!out) return MEDA_ERR_BAD_ARG;
    int rc = meda_vec_check_config(cfg);
    if (rc) return rc;
    DeviceGuard g(cfg->device);
    if (!g.ok) return MEDA_ERR_HIP;
    meda_vec *h = new (std::nothrow) meda_vec();
    if (!h) return MEDA_ERR_BAD_ARG;
    h->cfg = *cfg;
    MCfg &d = h->dc;
    d.W = cfg->width; d.L = cfg->length; d.fov = cfg->fov; d.ff = cfg->fov * cfg->fov;
    d.version = cfg->obs_version; d.obs_len = (cfg->obs_version == 2 ? 3 : 4) * d.ff + 2;
    d.max_step = cfg->width + cfg->length; d.b_degrade = cfg->b_degrade != 0; d.E = cfg->n_envs; d.n = cfg->n_agents;
    d.k0 = (uint32_t)cfg->seed; d.k1 = (uint32_t)(cfg->seed >> 32); d.env_id0 = cfg->env_id0;
    d.per_healthy = 1.0 - cfg->per_degrade;
    const int E = cfg->n_envs, n = cfg->n_agents;
    const size_t row = (size_t)n * d.obs_len;
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && v > 0) h->n_cu = v;
    }
    // observation tile, at most 64 chips (the loader wave keeps one chip per lane).  Measured on MI355X at 30x30, 4
    // droplets, fov 19 (tools/probe/meda_tile_sweep.sh, profiles/r02/meda_tile_sweep.txt): the base observation is
    // fastest with the largest tile (64 KB of LDS, two workgroups per CU: its fill is light, fewer barriers per
    // byte), the v0_2 one with ~40 KB (four per CU: bands and the set order make its fill heavier, more overlap)
    size_t cap = (cfg->obs_version == 2 ? 40 : 64) * 1024;
    if (const char *v = getenv("MEDA_VEC_TILE_KB")) cap = (size_t)atoi(v) * 1024;  // tuning knob
    int T = 1;
    while (T < 64 && obs_lds_bytes(T + 1, n, (int)row) <= cap) ++T;
    while (T > 1 && (E + T - 1) / T < 2 * h->n_cu) --T;  // keep the grid wide for small batches
    if (const char *v = getenv("MEDA_VEC_OBS_TILE")) {  // tuning knob: chips per tile
        const int t = atoi(v);
        if (t >= 1 && t <= 64 && obs_lds_bytes(t, n, (int)row) <= 64 * 1024) T = t;
    }
    d.T = T;
    h->obs_lds = obs_lds_bytes(T, n, (int)row);
    {
        int per_cu = (int)((size_t)160 * 1024 / h->obs_lds);
        per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
        const int ntiles = (E + T - 1) / T;
        h->obs_grid = ntiles < h->n_cu * per_cu ? ntiles : h->n_cu * per_cu;
    }
    hipStream_t s = (hipStream_t)stream;
    memset(&h->dp, 0, sizeof(h->dp));
    auto fail = [&](hipError_t e, const char *what, int line) { hip_fail(e, what, line); meda_vec_destroy(h); return MEDA_ERR_HIP; };
#define CREATE_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(_e, #expr, __LINE__); } while (0)
    const size_t st_bytes = (size_t)(n + 4) * E * 4, starts_bytes = (size_t)n * E * 4, cells = (size_t)cfg->width * cfg->length;
    CREATE_TRY(hipMalloc(&h->dp.st, st_bytes));
    CREATE_TRY(hipMalloc(&h->dp.starts, starts_bytes));
    CREATE_TRY(hipMalloc(&h->dp.reset_flag, (size_t)E));
    h->bytes = st_bytes + starts_bytes + E;
    CREATE_TRY(hipMemsetAsync(h->dp.st, 0, st_bytes, s));
    CREATE_TRY(hipMemsetAsync(h->dp.reset_flag, 0, (size_t)E, s));
    for (int dd = -128; dd <= 127; ++dd) {  // python round() = half to even on the double quotient (meda.py:894)
        h->zoom_host[dd + 128] = (int8_t)(int)std::nearbyint((double)dd / ((double)cfg->width / 30.0));
        h->zoom_host[256 + dd + 128] = (int8_t)(int)std::nearbyint((double)dd / ((double)cfg->length / 30.0));
    }
    CREATE_TRY(hipMalloc(&h->zoom_dev, sizeof(h->zoom_host)));
    CREATE_TRY(hipMemcpyAsync(h->zoom_dev, h->zoom_host, sizeof(h->zoom_host), hipMemcpyHostToDevice, s));
    h->dp.zoom = h->zoom_dev;
    if (cfg->b_degrade || cfg->with_maps) {
        CREATE_TRY(hipMalloc(&h->dp.health, cells * E * 8));
        CREATE_TRY(hipMalloc(&h->dp.degrade, cells * E * 8));
        CREATE_TRY(hipMalloc(&h->dp.usage, cells * E * 2));
        h->bytes += cells * E * 18;
        const size_t total = cells * E;
        (void)hipGetLastError();
        hipLaunchKernelGGL(k_meda_init_maps, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, h->dc, h->dp);
        CREATE_TRY(hipGetLastError());
    }
    rc = launch_reset(h, nullptr, 3, s);
    if (rc) { meda_vec_destroy(h); return rc; }
    CREATE_TRY(hipMemsetAsync(h->dp.reset_flag, 0, (size_t)E, s));  // construction does not run updateHealth
    CREATE_TRY(hipStreamSynchronize(s));  // the zoom table upload reads host memory owned by the handle
    *out = h;
    return MEDA_OK;
}

int meda_vec_destroy(meda_vec *h) {
    if (!h) return MEDA_OK;
    DeviceGuard g(h->cfg.device);
    (void)hipFree(h->dp.st); (void)hipFree(h->dp.starts); (void)hipFree(h->dp.reset_flag);
    (void)hipFree(h->dp.health); (void)hipFree(h->dp.degrade); (void)hipFree(h->dp.usage); (void)hipFree(h->zoom_dev);
    for (int i = 0; i < 2 * meda_vec::kTimed; ++i)
        if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    delete h;
    return MEDA_OK;
}

size_t meda_vec_state_bytes(const meda_vec *h) { return h ? h->bytes : 0; }
int meda_vec_obs_len(const meda_vec *h) { return h ? h->dc.obs_len : MEDA_ERR_BAD_ARG; }
int meda_vec_max_step(const meda_vec *h) { return h ? h->dc.max_step : MEDA_ERR_BAD_ARG; }
int meda_vec_n_envs(const meda_vec *h) { return h ? h->cfg.n_envs : MEDA_ERR_BAD_ARG; }
int meda_vec_n_agents(const meda_vec *h) { return h ? h->cfg.n_agents : MEDA_ERR_BAD_ARG; }

int meda_vec_reset(meda_vec *h, const uint8_t *d_mask, int8_t *d_obs, void *stream) {
    if (!h) return MEDA_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_reset(h, d_mask, 0, s);
    if (rc) return rc;
    rc = launch_update_health(h, s);
    if (rc || !d_obs) return rc;
    return launch_observe(h, d_mask, d_obs, s);
}

int meda_vec_restart(meda_vec *h, const uint8_t *d_mask, int8_t *d_obs, void *stream) {
    if (!h) return MEDA_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    int rc = launch_reset(h, d_mask, 2, (hipStream_t)stream);
    if (rc || !d_obs) return rc;
    return launch_observe(h, d_mask, d_obs, (hipStream_t)stream);
}

int meda_vec_set_task(meda_vec *h, const int32_t *d_starts, const int32_t *d_ends, void *stream) {
    if (!h || !d_starts || !d_ends) return MEDA_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    LAUNCH(k_meda_set_task, dim3((h->cfg.n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->dc, h->dp, d_starts, d_ends);
    return MEDA_OK;
}
int meda_vec_get_task(const meda_vec *h, int32_t *d_starts, int32_t *d_ends, void *stream) {
    if (!h) return MEDA_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    LAUNCH(k_meda_get_task, dim3((h->cfg.n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->dc, h->dp, d_starts, d_ends);
    return MEDA_OK;
}

int meda_vec_step(meda_vec *h, const void *d_actions, const double *d_uniforms, const uint8_t *d_active, uint32_t flags,
                  const meda_vec_step_out *out, void *stream) {
    if (!h || !d_actions || !out) return MEDA_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    hipStream_t s = (hipStream_t)stream;
    MStepArgs a;
    a.actions = d_actions; a.uniforms = d_uniforms; a.active = d_active; a.flags = flags; a.out = *out;
    int rc = launch_step(h, a, s);
    if (rc) return rc;
    if (flags & MEDA_STEP_AUTORESET) {
        rc = launch_update_health(h, s);
        if (rc) return rc;
    }
    if (out->d_obs) return launch_observe(h, nullptr, out->d_obs, s);
    return MEDA_OK;
}

int meda_vec_observe(const meda_vec *h, const uint8_t *d_mask, int8_t *d_obs, void *stream) {
    if (!h || !d_obs) return MEDA_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
#ifdef MEDA_ABLATE
    {
        const char *v = getenv("MEDA_ABLATE");
        const int bits = v ? atoi(v) : 0;
        static int last = 0;
        if (bits != last) {
            (void)hipStreamSynchronize((hipStream_t)stream);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ablate), &bits, sizeof(bits));
            last = bits;
        }
    }
#endif
    return launch_observe(h, d_mask, d_obs, (hipStream_t)stream);
}

int meda_vec_get_state(const meda_vec *h, int32_t *d_pos, uint8_t *d_status, int32_t *d_step_count, uint8_t *d_failed,
                       void *stream) {
    if (!h) return MEDA_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    LAUNCH(k_meda_get_state, dim3((h->cfg.n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->dc, h->dp, d_pos,
           d_status, d_step_count, d_failed);
    return MEDA_OK;
}

int meda_vec_get_map(const meda_vec *h, int which, double *d_buf, void *stream) {
    if (!h || !d_buf || which < 0 || which > 2) return MEDA_ERR_BAD_ARG;
    if (!h->dp.health) return MEDA_ERR_NO_MAPS;
    DeviceGuard g(h->cfg.device);
    const size_t total = (size_t)h->cfg.n_envs * h->cfg.width * h->cfg.length;
    LAUNCH(k_meda_get_map, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, total, h->dp.health,
           h->dp.degrade, h->dp.usage, which, d_buf);
    return MEDA_OK;
}
int meda_vec_set_map(meda_vec *h, int which, const double *d_buf, void *stream) {
    if (!h || !d_buf || which < 0 || which > 2) return MEDA_ERR_BAD_ARG;
    if (!h->dp.health) return MEDA_ERR_NO_MAPS;
    DeviceGuard g(h->cfg.device);
    const size_t total = (size_t)h->cfg.n_envs * h->cfg.width * h->cfg.length;
    LAUNCH(k_meda_set_map, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, total, h->dp.health,
           h->dp.degrade, h->dp.usage, which, d_buf);
    return MEDA_OK;
}

int meda_vec_launch_shape(const meda_vec *h, int32_t out[4]) {
    if (!h || !out) return MEDA_ERR_BAD_ARG;
    out[0] = kBlock; out[1] = h->dc.T; out[2] = kObsBlock; out[3] = h->obs_grid;
    return MEDA_OK;
}

int meda_vec_observe_timing(meda_vec *h, int enable) {
    if (!h) return MEDA_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    if (enable && !h->ev[0])
        for (int i = 0; i < 2 * meda_vec::kTimed; ++i) HIP_TRY(hipEventCreate(&h->ev[i]));
    h->timing = enable != 0;
    h->timed = 0;
    return MEDA_OK;
}

int meda_vec_observe_timing_read(meda_vec *h, double *total_us, int *launches) {
    if (!h || !total_us || !launches) return MEDA_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    double sum = 0.0;
    for (int i = 0; i < h->timed; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(h->ev[2 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]));
        sum += (double)ms * 1e3;
    }
    *total_us = sum; *launches = h->timed;
    h->timed = 0;
    return MEDA_OK;
}

const char *meda_vec_strerror(int code) {
    switch (code) {
    case MEDA_OK: return "ok";
    case MEDA_ERR_BAD_ARG: return "bad argument";
    case MEDA_ERR_TOO_MANY_DROPLETS: return "Too many droplets in the MEDA array";
    case MEDA_ERR_BAD_SIZE: return "w > 0 and l > 0 required";
    case MEDA_ERR_NO_AGENTS: return "n_agents > 0 required";
    case MEDA_ERR_UNSUPPORTED: return "configuration outside the build limits";
    case MEDA_ERR_NO_MAPS: return "handle was created without health/usage/degrade maps";
    case MEDA_ERR_HIP: return "HIP runtime error";
    default: return "unknown error";
    }
}
int meda_vec_last_hip_error(void) { return g_last_hip; }

}  // extern "C"
