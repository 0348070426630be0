// dmfb_vec.hip -- C ABI (include/dmfb_vec.h) of the vectorised DMFB environment, the
// N-independent kernels (observation, task/state/map accessors) and the dispatch to the per-N
// transition/reset kernels built from dmfb_vec_n.hip.  Device code: dmfb_kernels.h.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "dmfb_kernels.h"

using namespace dmfbk;

namespace {

// ---- small utility kernels -----------------------------------------------------------------------
__global__ void k_set_task(DevCfg c, DevPtrs p, const int32_t *starts, const int32_t *ends) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.E) return;
    const int n = c.n, np = (n + 1) / 2, E = c.E;
    bool dup = false;
    for (int w = 0; w < np; ++w) {
        uint32_t sw = 0, gw = 0;
        for (int h = 0; h < 2 && 2 * w + h < n; ++h) {
            const size_t k = ((size_t)e * n + 2 * w + h) * 2;
            sw |= ((uint32_t)(starts[k] & 0xff) | ((uint32_t)(starts[k + 1] & 0xff) << 8)) << (16 * h);
            gw |= ((uint32_t)(ends[k] & 0xff) | ((uint32_t)(ends[k + 1] & 0xff) << 8)) << (16 * h);
        }
        p.starts[(size_t)w * E + e] = sw;
        p.st[(size_t)w * E + e] = sw;
        p.st[(size_t)(np + w) * E + e] = gw;
    }
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            const size_t a = ((size_t)e * n + i) * 2, b = ((size_t)e * n + j) * 2;
            dup |= (starts[a] == starts[b]) && (starts[a + 1] == starts[b + 1]);
        }
    // counters to zero; the usage log keeps its length (injecting a task does not touch the maps)
    p.st[(size_t)(2 * np) * E + e] = (p.st[(size_t)(2 * np) * E + e] & (0xfffu << kUlenShift)) | (dup ? (FLAG_DUP << 16) : 0u);
    p.st[(size_t)(2 * np + 1) * E + e] = 0;
}

// Fold every chip's usage log into its usage map (no updateHealth): before the map is read or overwritten from outside.
// One wave per chip, LDS histogram per wave (dynamic LDS = 4 * hist_bytes).
__global__ __launch_bounds__(kBlock) void k_flush_usage(DevCfg c, DevPtrs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int e = blockIdx.x * (kBlock / kWave) + (int)(threadIdx.x / kWave);
    if (e >= c.E) return;
    const int lane = (int)(threadIdx.x & (kWave - 1));
    const size_t wstep = (size_t)(2 * ((c.n + 1) / 2)) * c.E + e;
    const uint32_t s = p.st[wstep];
    const int ulen = (int)(s >> kUlenShift);
    if (ulen == 0) return;  // (wave-uniform)
    uint16_t *hist = c.hist_bytes ? (uint16_t *)(smem + (size_t)(threadIdx.x / kWave) * c.hist_bytes) : nullptr;
    flush_usage(c, p, e, ulen, false, hist, lane);
    if (lane == 0) p.st[wstep] = s & ~(0xfffu << kUlenShift);
}

__global__ void k_get_task(DevCfg c, DevPtrs p, int32_t *starts, int32_t *ends) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.E) return;
    const int n = c.n, np = (n + 1) / 2, E = c.E;
    for (int i = 0; i < n; ++i) {
        const uint32_t s = (p.starts[(size_t)(i >> 1) * E + e] >> (16 * (i & 1))) & 0xffff;
        const uint32_t g = (p.st[(size_t)(np + (i >> 1)) * E + e] >> (16 * (i & 1))) & 0xffff;
        const size_t k = ((size_t)e * n + i) * 2;
        if (starts) { starts[k] = s & 0xff; starts[k + 1] = s >> 8; }
        if (ends) { ends[k] = g & 0xff; ends[k + 1] = g >> 8; }
    }
}

__global__ void k_set_blocks(DevCfg c, DevPtrs p, const int32_t *blocks, int nb) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.E) return;
    for (int b = 0; b < nb; ++b) {
        const int32_t *q = blocks + ((size_t)e * nb + b) * 4;
        p.blocks[(size_t)b * c.E + e] = (uint32_t)(q[0] & 0xff) | ((uint32_t)(q[1] & 0xff) << 8) | ((uint32_t)(q[2] & 0xff) << 16) |
                                         ((uint32_t)(q[3] & 0xff) << 24);
    }
}
__global__ void k_get_blocks(DevCfg c, DevPtrs p, int32_t *blocks, int cap) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.E) return;
    for (int b = 0; b < c.nb; ++b) {
        const uint32_t o = p.blocks[(size_t)b * c.E + e];
        int32_t *q = blocks + ((size_t)e * cap + b) * 4;
        q[0] = o & 0xff; q[1] = (o >> 8) & 0xff; q[2] = (o >> 16) & 0xff; q[3] = o >> 24;
    }
}

__global__ void k_get_state(DevCfg c, DevPtrs p, int32_t *pos, int32_t *dist, int32_t *step_count, int64_t *cons) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.E) return;
    const int n = c.n, np = (n + 1) / 2, E = c.E;
    for (int i = 0; i < n; ++i) {
        const uint32_t s = (p.st[(size_t)(i >> 1) * E + e] >> (16 * (i & 1))) & 0xffff;
        const uint32_t g = (p.st[(size_t)(np + (i >> 1)) * E + e] >> (16 * (i & 1))) & 0xffff;
        const int x = s & 0xff, y = s >> 8, gx = g & 0xff, gy = g >> 8;
        if (pos) { pos[((size_t)e * n + i) * 2] = x; pos[((size_t)e * n + i) * 2 + 1] = y; }
        if (dist) dist[(size_t)e * n + i] = iabs(x - gx) + iabs(y - gy);
    }
    if (step_count) step_count[e] = (int32_t)(p.st[(size_t)(2 * np) * E + e] & 0xffff);
    if (cons) cons[e] = (int64_t)p.st[(size_t)(2 * np + 1) * E + e];
}

__global__ void k_set_word(int *dst, int v) { *dst = v; }  // DevPtrs::dflags, stream-ordered and graph-capturable

__global__ void k_get_map(size_t total, const double *health, const double *degrade, const uint16_t *usage, int which,
                          double *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    out[i] = which == DMFB_MAP_HEALTH ? health[i] : which == DMFB_MAP_DEGRADE ? degrade[i] : (double)usage[i];
}
__global__ void k_set_map(size_t total, double *health, double *degrade, uint16_t *usage, int which, const double *in) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (which == DMFB_MAP_HEALTH) health[i] = in[i];
    else if (which == DMFB_MAP_DEGRADE) degrade[i] = in[i];
    else usage[i] = (uint16_t)in[i];
}

thread_local int g_last_hip = 0;
// DMFB_VEC_DEBUG=1 in the environment prints the failing HIP call to stderr.
inline int hip_fail(hipError_t e, const char *what, int line) {
    g_last_hip = (int)e;
    if (getenv("DMFB_VEC_DEBUG")) fprintf(stderr, "dmfb_vec: %s failed at line %d: %s (%d)\n", what, line, hipGetErrorString(e), (int)e);
    return DMFB_ERR_HIP;
}
#define HIP_TRY(expr)                                                 \
    do {                                                              \
        hipError_t _e = (expr);                                       \
        if (_e != hipSuccess) return hip_fail(_e, #expr, __LINE__);   \
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

// python round(): half-to-even on the double value (dmfb.py:444-453)
int zoom_one(int d, int hf, int size) {
    if (std::abs(d) > hf) {
        const double scale = (double)(size - hf) / (double)(10 - hf);
        if (d > 0) return (int)std::nearbyint((double)(d - hf) / scale) + hf;
        return (int)std::nearbyint((double)(d + hf) / scale) - hf;
    }
    return d;
}

}  // namespace

struct dmfb_vec {
    dmfb_vec_config cfg;
    DevCfg dc;
    DevPtrs dp;
    int8_t zoom_host[2 * 511];
    int8_t *zoom_dev = nullptr;
    unsigned long long *band_dev = nullptr;  // DevPtrs::band
    int *dflags_dev = nullptr;               // DevPtrs::dflags
    int use_lanes = 0;                       // knob: lane-per-droplet transition for n >= 8 (dmfb_step_lanes.h); off: slower
    int obs_oneshot = 0;                     // measurement knob: k_observe with one workgroup per tile
    int obs_per_cu = 0;                      // cap on k_observe's persistent workgroups per CU (0: as many as LDS admits)
    size_t bytes = 0;
    int T_fused = 16;    // chips per workgroup of the fused step+observe launch (<= 64)
    int T_obs = 16;      // chips per workgroup of k_observe
    int split_min = 0;   // batches of at least this many chips use the step-only + observe pair
    int T_min = 16;      // smallest tile pick_tile may choose (DMFB_VEC_MIN_TILE); smaller tiles do not pay off (measured)
    int n_cu = 256;      // compute units of the device (persistent grid of the observation kernel)
    // dmfb_vec_observe_timing: event pairs that receive the dispatch time stamps of the observation kernel
    static constexpr int kTimed = 256;
    hipEvent_t ev[2 * kTimed] = {};
    int timing = 0, timed = 0;
};

namespace {

// LDS of the usage histograms (one per wave) of a kernel that may fold usage logs into the maps
size_t hist_lds(const dmfb_vec *h) { return h->dp.health ? (size_t)(kBlock / kWave) * h->dc.hist_bytes : 0; }

// chips up to this many cells count a usage log through an LDS histogram; larger ones use global atomics
constexpr int kHistMaxCells = 4096;

// 8-byte words per band image: the fov*fov layer bytes, padded so that the kernel reads them in batches of 12
int band_words(int fov) { return ((fov * fov + 7) / 8 + 11) / 12 * 12; }

// Tile of the LDS-staged observation: at most 64 chips (wave 0 owns one chip per lane in the fused launch),
// obs block <= 40 KB so that >= 3-4 workgroups share a CU's 160 KiB, and halved while the grid would not
// cover the 256 CUs `min_groups`/256 times over (any tile size works: the tile is phase-aligned in LDS).
int pick_tile(const dmfb_vec *h, int min_groups) {
    const int row = h->cfg.n_agents * h->dc.obs_len;
    int T = 64;
    size_t cap = 40 * 1024;
    if (const char *v = getenv("DMFB_VEC_TILE_KB")) cap = (size_t)atoi(v) * 1024;  // tuning knob
    while (T > 1 && ((size_t)T * row > cap || tile_lds_bytes(T, h->cfg.n_agents, h->dc.obs_len, true, table_words(h->dc.hf, h->dc.nq)) + hist_lds(h) > 64 * 1024)) T >>= 1;
    while (T > h->T_min && (h->cfg.n_envs + T - 1) / T < min_groups) T >>= 1;
    return T;
}

constexpr int kStepOnlyTile = 256;  // step-only launch: every wave of the workgroup owns 64 chips


template <int N> int observe_n(const dmfb_vec *h, const uint8_t *mask, int8_t *obs, hipStream_t s) {
    const int T = h->T_obs;
    const size_t lds = tile_lds_bytes(T, N, h->dc.obs_len, true, table_words(h->dc.hf, h->dc.nq));
    // persistent grid: as many workgroups as fit the chip at once (LDS-limited, at most 8 per CU), or one per tile
    int per_cu = (int)((size_t)160 * 1024 / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
    if (h->obs_per_cu > 0 && h->obs_per_cu < per_cu) per_cu = h->obs_per_cu;
    const int ntiles = (h->cfg.n_envs + T - 1) / T;
    int grid = ntiles < h->n_cu * per_cu ? ntiles : h->n_cu * per_cu;
    if (h->obs_oneshot) grid = ntiles;  // measurement knob: one workgroup per tile, dispatched in address order
    hipEvent_t t0 = nullptr, t1 = nullptr;
    dmfb_vec *hm = const_cast<dmfb_vec *>(h);
    if (h->timing && h->timed < dmfb_vec::kTimed) {
        t0 = hm->ev[2 * h->timed]; t1 = hm->ev[2 * h->timed + 1];
        hm->timed += 1;
    }
    HIP_TRY(launch_observe_n<N>(h->dc, h->dp, mask, obs, grid, lds, s, t0, t1));
    return DMFB_OK;
}

// One fused launch for small batches (launch latency dominates); for large batches a step-only launch
// (all four waves stepping, no LDS tile, high occupancy) followed by the observation kernel.
// step-only launch (no observation tile): one lane per chip, 256 chips per workgroup.  DMFB_VEC_LANES=1 selects the
// lane-per-droplet kernel for n >= 8 (dmfb_step_lanes.h: 16 lanes per chip) -- bit-identical, measured 1.6-1.9x SLOWER than the
// lane-per-chip kernel, kept as the tested record of that experiment
template <int N> int step_only_n(dmfb_vec *h, const StepArgs &b, hipStream_t s) {
    const int E = h->cfg.n_envs;
    DevCfg c = h->dc;
    c.T = kStepOnlyTile;
    if constexpr (N >= kLanesMinN) {
        if (h->use_lanes) {
            HIP_TRY(launch_step_lanes_n<N>(c, h->dp, b, (E + 15) / 16, hist_lds(h), s));
            return DMFB_OK;
        }
    }
    HIP_TRY(launch_step_n<N>(c, h->dp, b, (E + c.T - 1) / c.T, tile_lds_bytes(c.T, N, c.obs_len, false, 0) + hist_lds(h), s));
    return DMFB_OK;
}

template <int N> int step_n(dmfb_vec *h, const StepArgs &a, hipStream_t s) {
    const int E = h->cfg.n_envs;
    if (a.out.d_obs_terminal && (!a.out.d_obs || E >= h->split_min || !(a.flags & DMFB_STEP_AUTORESET))) return DMFB_ERR_UNSUPPORTED;
    if (a.out.d_obs && E >= h->split_min) {
        StepArgs b = a;
        b.out.d_obs = nullptr;
        const int rc = step_only_n<N>(h, b, s);
        if (rc) return rc;
        return observe_n<N>(h, nullptr, a.out.d_obs, s);
    }
    if (!a.out.d_obs) return step_only_n<N>(h, a, s);
    DevCfg c = h->dc;
    const bool with_obs = a.out.d_obs != nullptr;
    c.T = with_obs ? h->T_fused : kStepOnlyTile;
    HIP_TRY(launch_step_n<N>(c, h->dp, a, (E + c.T - 1) / c.T,
                             tile_lds_bytes(c.T, N, c.obs_len, with_obs, table_words(c.hf, c.nq)) + hist_lds(h), s));
    return DMFB_OK;
}
template <int N> int reset_n(dmfb_vec *h, const uint8_t *mask, int mode, hipStream_t s) {
    HIP_TRY(launch_reset_n<N>(h->dc, h->dp, mask, mode, (h->cfg.n_envs + (kBlock / kWave) - 1) / (kBlock / kWave),
                              (size_t)(kBlock / kWave) * h->dc.hist_bytes, s));
    return DMFB_OK;
}

#ifdef DMFB_STAMPS_ONLY_N  // the diagnostic build instantiates two droplet counts only
#define DISPATCH_N(n, FN, ...)                                  \
    switch (n) {                                                \
    case 4: return FN<4>(__VA_ARGS__);                          \
    case 10: return FN<10>(__VA_ARGS__);                        \
    default: return DMFB_ERR_UNSUPPORTED;                       \
    }
#else
#define DISPATCH_N(n, FN, ...)                                  \
    switch (n) {                                                \
    case 1: return FN<1>(__VA_ARGS__);                          \
    case 2: return FN<2>(__VA_ARGS__);                          \
    case 3: return FN<3>(__VA_ARGS__);                          \
    case 4: return FN<4>(__VA_ARGS__);                          \
    case 5: return FN<5>(__VA_ARGS__);                          \
    case 6: return FN<6>(__VA_ARGS__);                          \
    case 7: return FN<7>(__VA_ARGS__);                          \
    case 8: return FN<8>(__VA_ARGS__);                          \
    case 9: return FN<9>(__VA_ARGS__);                          \
    case 10: return FN<10>(__VA_ARGS__);                        \
    case 11: return FN<11>(__VA_ARGS__);                        \
    case 12: return FN<12>(__VA_ARGS__);                        \
    case 13: return FN<13>(__VA_ARGS__);                        \
    case 14: return FN<14>(__VA_ARGS__);                        \
    case 15: return FN<15>(__VA_ARGS__);                        \
    case 16: return FN<16>(__VA_ARGS__);                        \
    default: return DMFB_ERR_UNSUPPORTED;                       \
    }
#endif

int launch_step(dmfb_vec *h, const StepArgs &a, hipStream_t s) { DISPATCH_N(h->cfg.n_agents, step_n, h, a, s) }
int launch_observe(const dmfb_vec *h, const uint8_t *mask, int8_t *obs, hipStream_t s) {
    DISPATCH_N(h->cfg.n_agents, observe_n, h, mask, obs, s)
}
int launch_reset(dmfb_vec *h, const uint8_t *mask, int mode, hipStream_t s) {
    DISPATCH_N(h->cfg.n_agents, reset_n, h, mask, mode, s)
}

}  // namespace

extern "C" {

int dmfb_vec_check_config(const dmfb_vec_config *c) {
    if (!c) return DMFB_ERR_BAD_ARG;
    if (c->width < 5 || c->length < 5) return DMFB_ERR_CHIP_TOO_SMALL;
    if (c->n_agents <= 0) return DMFB_ERR_NO_AGENTS;
    if (c->fov > (c->width < c->length ? c->width : c->length)) return DMFB_ERR_FOV_TOO_LARGE;
    if (c->n_agents > (int)((c->width + 1) * (c->length + 1) / 9)) return DMFB_ERR_TOO_MANY_DROPLETS;
    if (c->n_agents > DMFB_MAX_AGENTS || c->width > DMFB_MAX_DIM || c->length > DMFB_MAX_DIM || c->fov < 1 ||
        c->n_blocks < 0 || c->n_blocks > DMFB_MAX_BLOCKS)
        return DMFB_ERR_UNSUPPORTED;
    if (c->n_envs <= 0) return DMFB_ERR_BAD_ARG;
    if (tile_lds_bytes(1, c->n_agents, 3 * c->fov * c->fov + 2, true, table_words(c->fov / 2, band_words(c->fov))) > 64 * 1024)
        return DMFB_ERR_UNSUPPORTED;
    return DMFB_OK;
}

int dmfb_vec_create(const dmfb_vec_config *cfg, void *stream, dmfb_vec **out) {
    if (!out) return DMFB_ERR_BAD_ARG;
    int rc = dmfb_vec_check_config(cfg);
    if (rc) return rc;
    DeviceGuard g(cfg->device);
    if (!g.ok) return DMFB_ERR_HIP;
    dmfb_vec *h = new (std::nothrow) dmfb_vec();
    if (!h) return DMFB_ERR_BAD_ARG;
    h->cfg = *cfg;
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && v > 0) h->n_cu = v;
    }
    DevCfg &d = h->dc;
    d.W = cfg->width; d.L = cfg->length; d.fov = cfg->fov; d.hf = cfg->fov / 2; d.ff = cfg->fov * cfg->fov;
    d.obs_len = 3 * d.ff + 2; d.max_step = 2 * (cfg->width + cfg->length);
    d.nq = band_words(cfg->fov);
    d.stall = cfg->stall != 0; d.b_degrade = cfg->b_degrade != 0; d.E = cfg->n_envs; d.n = cfg->n_agents;
    d.k0 = (uint32_t)cfg->seed; d.k1 = (uint32_t)(cfg->seed >> 32); d.env_id0 = cfg->env_id0;
    d.per_healthy = 1.0 - cfg->per_degrade;
    const int E = cfg->n_envs, n = cfg->n_agents;
    const size_t cells = (size_t)cfg->width * cfg->length;
    const size_t st_bytes = (size_t)rec_words(n) * E * 4, starts_bytes = (size_t)((n + 1) / 2) * E * 4;
    hipStream_t s = (hipStream_t)stream;
    auto fail = [&](int code) { dmfb_vec_destroy(h); return code; };
#define CREATE_TRY(expr)                                                            \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) { hip_fail(_e, #expr, __LINE__); return fail(DMFB_ERR_HIP); } \
    } while (0)
    memset(&h->dp, 0, sizeof(h->dp));
    CREATE_TRY(hipMalloc(&h->dp.st, st_bytes));
    CREATE_TRY(hipMalloc(&h->dp.starts, starts_bytes));
    h->bytes = st_bytes + starts_bytes;
    if (cfg->b_degrade || cfg->with_maps) {
        CREATE_TRY(hipMalloc(&h->dp.health, cells * E * 8));
        CREATE_TRY(hipMalloc(&h->dp.degrade, cells * E * 8));
        CREATE_TRY(hipMalloc(&h->dp.usage, cells * E * 2 + 4));  // + 4: the 32-bit atomics of the large-chip path stay in bounds
        d.ucap = d.max_step;
        d.lstride = 16;
        if (const char *v = getenv("DMFB_VEC_LOG_STRIDE")) d.lstride = atoi(v) == 16 ? 16 : n;  // measurement knob: n = packed entries
        CREATE_TRY(hipMalloc(&h->dp.ulog, (size_t)E * d.ucap * d.lstride * 2));
        CREATE_TRY(hipMalloc(&h->dp.kmap, kmap_bytes(cells) * E));
        CREATE_TRY(hipMalloc(&h->dflags_dev, 4));
        h->dp.dflags = h->dflags_dev;
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, s, h->dflags_dev, 1);  // until health or degrade is replaced (set_map)
        CREATE_TRY(hipGetLastError());
        d.hist_bytes = (int)cells <= kHistMaxCells ? (int)((cells * 2 + 15) & ~(size_t)15) : 0;
        h->bytes += cells * E * 18 + kmap_bytes(cells) * E + (size_t)E * d.ucap * d.lstride * 2;
    }
    // GenRandomBlocks guards (dmfb.py:230-234): no blocks on tiny chips or above 20 % coverage
    d.nb = cfg->n_blocks;
    if (cfg->width < 5 || cfg->length < 5 || (double)(cfg->n_blocks * 4) / (double)(cfg->width * cfg->length) > 0.2) d.nb = 0;
    if (cfg->n_blocks > 0) {
        CREATE_TRY(hipMalloc(&h->dp.blocks, (size_t)cfg->n_blocks * E * 4));
        CREATE_TRY(hipMemsetAsync(h->dp.blocks, 0, (size_t)cfg->n_blocks * E * 4, s));
        h->bytes += (size_t)cfg->n_blocks * E * 4;
    }
    for (int dd = -255; dd <= 255; ++dd) {
        int zx = dd, zy = dd;
        if (d.hf != 10) { zx = zoom_one(dd, d.hf, d.W); zy = zoom_one(dd, d.hf, d.L); }
        h->zoom_host[dd + 255] = (int8_t)zx;
        h->zoom_host[511 + dd + 255] = (int8_t)zy;
    }
    CREATE_TRY(hipMalloc(&h->zoom_dev, sizeof(h->zoom_host)));
    h->bytes += sizeof(h->zoom_host);
    CREATE_TRY(hipMemcpyAsync(h->zoom_dev, h->zoom_host, sizeof(h->zoom_host), hipMemcpyHostToDevice, s));
    h->dp.zoom = h->zoom_dev;
    {   // observation tables (DevPtrs::band): band images [axis][pattern][nq], then the zoom table
        const int fov = d.fov, hf = d.hf, ff = d.ff, npat = 2 * hf + 1;
        const size_t words = (size_t)table_words(hf, d.nq);
        unsigned long long *img = new (std::nothrow) unsigned long long[words]();
        if (!img) return fail(DMFB_ERR_BAD_ARG);
        for (int axis = 0; axis < 2; ++axis)
            for (int pat = 1; pat < npat; ++pat) {
                unsigned char *bytes = (unsigned char *)(img + ((size_t)axis * npat + pat) * d.nq);
                for (int b = 0; b < ff; ++b) {
                    const int v = axis == 0 ? b / fov : b % fov;  // window x (first axis) or y
                    // pattern p <= hf: the first p window rows/columns lie outside the chip (obs[2, 0:left, :] = 1,
                    // dmfb.py:430-431); p > hf: the last p - hf ones (obs[2, -right:, :] = 1, dmfb.py:432-433)
                    if (pat <= hf ? v < pat : v >= fov - (pat - hf)) bytes[b] = 1;
                }
            }
        memcpy(img + (size_t)2 * npat * d.nq, h->zoom_host, sizeof(h->zoom_host));
        hipError_t e1 = hipMalloc(&h->band_dev, words * 8);
        if (e1 == hipSuccess) e1 = hipMemcpy(h->band_dev, img, words * 8, hipMemcpyHostToDevice);
        delete[] img;
        if (e1 != hipSuccess) { hip_fail(e1, "observation table upload", __LINE__); return fail(DMFB_ERR_HIP); }
        h->bytes += words * 8;
        h->dp.band = h->band_dev;
    }
    CREATE_TRY(hipMemsetAsync(h->dp.st, 0, st_bytes, s));
    d.fov_magic = d.fov >= 2 ? (uint32_t)((0x100000000ull + (uint64_t)d.fov - 1) / (uint64_t)d.fov) : 0u;
    if (const char *v = getenv("DMFB_VEC_MIN_TILE")) h->T_min = atoi(v) > 0 ? atoi(v) : 1;  // tuning knob
    h->T_fused = pick_tile(h, 1024);
    h->T_obs = pick_tile(h, 2048);
    {   // observation kernel: at most 128 rows per tile, so that the two half-workgroups (layer 0 | layer 1, first | second half
        // of the band image) each cover every row in one pass
        int t = 128 / n;
        t = t < 1 ? 1 : (t > 64 ? 64 : t);
        while (t > 1 && tile_lds_bytes(t, n, d.obs_len, true, table_words(d.hf, d.nq)) > 64 * 1024) --t;
        while (t > h->T_min && (E + t - 1) / t < 1024) t = (t + 1) / 2;
        h->T_obs = t;
    }
    if (const char *v = getenv("DMFB_VEC_OBS_TILE")) {  // tuning knob: chips per workgroup of the observation kernel (1..64)
        const int t = atoi(v);
        if (t >= 1 && t <= 64 && tile_lds_bytes(t, n, d.obs_len, true, table_words(d.hf, d.nq)) <= 64 * 1024) h->T_obs = t;
    }
    d.T = h->T_fused; d.T_obs = h->T_obs;
    for (uint32_t k = 0; k < 64u * DMFB_MAX_AGENTS * (uint32_t)d.fov && d.fov >= 2; ++k)  // the magic must be exact on the range used
        if ((uint32_t)(((uint64_t)k * d.fov_magic) >> 32) != k / (uint32_t)d.fov) return fail(DMFB_ERR_UNSUPPORTED);
    h->split_min = 32768;
    if (const char *v = getenv("DMFB_VEC_SPLIT_MIN_ENVS")) h->split_min = atoi(v);  // tuning / test knob
    if (const char *v = getenv("DMFB_VEC_LANES")) h->use_lanes = atoi(v);             // measurement / test knob: lane-per-droplet transition for n >= 8
    if (const char *v = getenv("DMFB_VEC_OBS_ONESHOT")) h->obs_oneshot = atoi(v);     // measurement knob (see observe_n)
    if (const char *v = getenv("DMFB_VEC_OBS_PER_CU")) h->obs_per_cu = atoi(v);       // tuning knob: persistent workgroups per CU of k_observe
    rc = launch_reset(h, nullptr, 3, s);
    if (rc) return fail(rc);
    // the zoom table upload reads host memory owned by the handle: make it safe to use right away
    CREATE_TRY(hipStreamSynchronize(s));
    *out = h;
    return DMFB_OK;
}

int dmfb_vec_destroy(dmfb_vec *h) {
    if (!h) return DMFB_OK;
    DeviceGuard g(h->cfg.device);
    (void)hipFree(h->dp.st); (void)hipFree(h->dp.starts); (void)hipFree(h->dp.health);
    (void)hipFree(h->dp.degrade); (void)hipFree(h->dp.usage); (void)hipFree(h->dp.ulog); (void)hipFree(h->dp.kmap); (void)hipFree(h->zoom_dev); (void)hipFree(h->dp.blocks);
    (void)hipFree(h->band_dev); (void)hipFree(h->dflags_dev);
    for (int i = 0; i < 2 * dmfb_vec::kTimed; ++i)
        if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    delete h;
    return DMFB_OK;
}

size_t dmfb_vec_state_bytes(const dmfb_vec *h) { return h ? h->bytes : 0; }
int dmfb_vec_obs_len(const dmfb_vec *h) { return h ? h->dc.obs_len : DMFB_ERR_BAD_ARG; }
int dmfb_vec_max_step(const dmfb_vec *h) { return h ? h->dc.max_step : DMFB_ERR_BAD_ARG; }
int dmfb_vec_n_envs(const dmfb_vec *h) { return h ? h->cfg.n_envs : DMFB_ERR_BAD_ARG; }
int dmfb_vec_n_agents(const dmfb_vec *h) { return h ? h->cfg.n_agents : DMFB_ERR_BAD_ARG; }

int dmfb_vec_reset(dmfb_vec *h, const uint8_t *d_mask, int new_flag, int8_t *d_obs, void *stream) {
    if (!h) return DMFB_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    int rc = launch_reset(h, d_mask, new_flag ? 1 : 0, (hipStream_t)stream);
    if (!rc && new_flag && !d_mask && h->dp.health) {  // every map is the generator's own again
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, (hipStream_t)stream, h->dflags_dev, 1);
        HIP_TRY(hipGetLastError());
    }
    if (rc || !d_obs) return rc;
    return launch_observe(h, d_mask, d_obs, (hipStream_t)stream);
}

int dmfb_vec_restart(dmfb_vec *h, const uint8_t *d_mask, int8_t *d_obs, void *stream) {
    if (!h) return DMFB_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    int rc = launch_reset(h, d_mask, 2, (hipStream_t)stream);
    if (rc || !d_obs) return rc;
    return launch_observe(h, d_mask, d_obs, (hipStream_t)stream);
}

int dmfb_vec_set_task(dmfb_vec *h, const int32_t *d_starts, const int32_t *d_ends, void *stream) {
    if (!h || !d_starts || !d_ends) return DMFB_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    (void)hipGetLastError();  // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL(k_set_task, dim3((h->cfg.n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->dc, h->dp,
                       d_starts, d_ends);
    HIP_TRY(hipGetLastError());
    return DMFB_OK;
}

int dmfb_vec_get_task(const dmfb_vec *h, int32_t *d_starts, int32_t *d_ends, void *stream) {
    if (!h) return DMFB_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    (void)hipGetLastError();  // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL(k_get_task, dim3((h->cfg.n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->dc, h->dp,
                       d_starts, d_ends);
    HIP_TRY(hipGetLastError());
    return DMFB_OK;
}

int dmfb_vec_set_blocks(dmfb_vec *h, const int32_t *d_blocks, int nb, void *stream) {
    if (!h || nb < 0 || nb > h->cfg.n_blocks || (nb > 0 && !d_blocks)) return DMFB_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    h->dc.nb = nb;
    if (nb == 0) return DMFB_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_set_blocks, dim3((h->cfg.n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->dc, h->dp,
                       d_blocks, nb);
    HIP_TRY(hipGetLastError());
    return DMFB_OK;
}

int dmfb_vec_get_blocks(const dmfb_vec *h, int32_t *d_blocks, int *nb_out, void *stream) {
    if (!h) return DMFB_ERR_BAD_ARG;
    if (nb_out) *nb_out = h->dc.nb;
    if (!d_blocks || h->dc.nb == 0) return DMFB_OK;
    DeviceGuard g(h->cfg.device);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_get_blocks, dim3((h->cfg.n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->dc, h->dp,
                       d_blocks, h->cfg.n_blocks);
    HIP_TRY(hipGetLastError());
    return DMFB_OK;
}

int dmfb_vec_step(dmfb_vec *h, const void *d_actions, const double *d_uniforms, const uint8_t *d_active,
                  uint32_t flags, const dmfb_vec_step_out *out, void *stream) {
    if (!h || !d_actions || !out) return DMFB_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    StepArgs a;
    a.actions = d_actions; a.uniforms = d_uniforms; a.active = d_active; a.flags = flags; a.out = *out;
    return launch_step(h, a, (hipStream_t)stream);
}

int dmfb_vec_observe(const dmfb_vec *h, const uint8_t *d_mask, int8_t *d_obs, void *stream) {
    if (!h || !d_obs) return DMFB_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    return launch_observe(h, d_mask, d_obs, (hipStream_t)stream);
}

int dmfb_vec_get_state(const dmfb_vec *h, int32_t *d_pos, int32_t *d_dist, int32_t *d_step_count,
                       int64_t *d_constraints, void *stream) {
    if (!h) return DMFB_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    (void)hipGetLastError();  // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL(k_get_state, dim3((h->cfg.n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->dc, h->dp,
                       d_pos, d_dist, d_step_count, d_constraints);
    HIP_TRY(hipGetLastError());
    return DMFB_OK;
}

int dmfb_vec_get_map(const dmfb_vec *h, int which, double *d_buf, void *stream) {
    if (!h || !d_buf || which < 0 || which > 2) return DMFB_ERR_BAD_ARG;
    if (!h->dp.health) return DMFB_ERR_NO_MAPS;
    DeviceGuard g(h->cfg.device);
    const size_t total = (size_t)h->cfg.n_envs * h->cfg.width * h->cfg.length;
    (void)hipGetLastError();  // drop stale errors left by other users of the runtime
    if (which == DMFB_MAP_USAGE) {  // the steps since the last reset are still in the usage log
        hipLaunchKernelGGL(k_flush_usage, dim3((h->cfg.n_envs + (kBlock / kWave) - 1) / (kBlock / kWave)), dim3(kBlock), hist_lds(h),
                           (hipStream_t)stream, h->dc, h->dp);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_get_map, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, total,
                       h->dp.health, h->dp.degrade, h->dp.usage, which, d_buf);
    HIP_TRY(hipGetLastError());
    return DMFB_OK;
}

int dmfb_vec_set_map(dmfb_vec *h, int which, const double *d_buf, void *stream) {
    if (!h || !d_buf || which < 0 || which > 2) return DMFB_ERR_BAD_ARG;
    if (!h->dp.health) return DMFB_ERR_NO_MAPS;
    DeviceGuard g(h->cfg.device);
    const size_t total = (size_t)h->cfg.n_envs * h->cfg.width * h->cfg.length;
    (void)hipGetLastError();  // drop stale errors left by other users of the runtime
    if (which != DMFB_MAP_USAGE) {  // health / degrade no longer follow from the generator: gather the float64 map
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, (hipStream_t)stream, h->dflags_dev, 0);
        HIP_TRY(hipGetLastError());
    }
    if (which == DMFB_MAP_USAGE) {  // pending log entries belong to the map that is being replaced: fold them in first
        hipLaunchKernelGGL(k_flush_usage, dim3((h->cfg.n_envs + (kBlock / kWave) - 1) / (kBlock / kWave)), dim3(kBlock), hist_lds(h),
                           (hipStream_t)stream, h->dc, h->dp);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_set_map, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, total,
                       h->dp.health, h->dp.degrade, h->dp.usage, which, d_buf);
    HIP_TRY(hipGetLastError());
    return DMFB_OK;
}

int dmfb_vec_launch_shape(const dmfb_vec *h, int32_t out[6]) {
    if (!h || !out) return DMFB_ERR_BAD_ARG;
    out[0] = h->T_fused; out[1] = h->T_obs; out[2] = h->split_min; out[3] = kStepOnlyTile;
    const size_t lds = tile_lds_bytes(h->T_obs, h->cfg.n_agents, h->dc.obs_len, true, table_words(h->dc.hf, h->dc.nq));
    int per_cu = (int)((size_t)160 * 1024 / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
    const int ntiles = (h->cfg.n_envs + h->T_obs - 1) / h->T_obs;
    out[4] = ntiles < h->n_cu * per_cu ? ntiles : h->n_cu * per_cu;
    out[5] = kObsBlock;
    return DMFB_OK;
}

int dmfb_vec_observe_timing(dmfb_vec *h, int enable) {
    if (!h) return DMFB_ERR_BAD_ARG;
    DeviceGuard g(h->cfg.device);
    if (enable && !h->ev[0])
        for (int i = 0; i < 2 * dmfb_vec::kTimed; ++i) HIP_TRY(hipEventCreate(&h->ev[i]));
    h->timing = enable != 0;
    h->timed = 0;
    return DMFB_OK;
}

int dmfb_vec_observe_timing_read(dmfb_vec *h, double *total_us, int *launches) {
    if (!h || !total_us || !launches) return DMFB_ERR_BAD_ARG;
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
    return DMFB_OK;
}

int dmfb_vec_zoom_lut(const dmfb_vec *h, int8_t *host_out) {
    if (!h || !host_out) return DMFB_ERR_BAD_ARG;
    memcpy(host_out, h->zoom_host, sizeof(h->zoom_host));
    return DMFB_OK;
}

#ifdef DMFB_STAMPS
// diagnostic build only (not declared in include/dmfb_vec.h): attach a device buffer of 8 uint64 per k_observe workgroup
int dmfb_vec_dbg_stamps(dmfb_vec *h, unsigned long long *d_buf) {
    if (!h) return DMFB_ERR_BAD_ARG;
    h->dp.dbg = d_buf;
    return DMFB_OK;
}
#endif

const char *dmfb_vec_strerror(int code) {
    switch (code) {
    case DMFB_OK: return "ok";
    case DMFB_ERR_BAD_ARG: return "bad argument";
    case DMFB_ERR_FOV_TOO_LARGE: return "Fov is too large";
    case DMFB_ERR_TOO_MANY_DROPLETS: return "Too many droplets for DMFB";
    case DMFB_ERR_CHIP_TOO_SMALL: return "width >= 5 and length >= 5 required";
    case DMFB_ERR_NO_AGENTS: return "n_agents > 0 required";
    case DMFB_ERR_UNSUPPORTED: return "configuration outside the build limits";
    case DMFB_ERR_BAD_ACTION: return "action is illegal";
    case DMFB_ERR_NO_MAPS: return "handle was created without health/usage/degrade maps";
    case DMFB_ERR_HIP: return "HIP runtime error";
    default: return "unknown error";
    }
}

int dmfb_vec_last_hip_error(void) { return g_last_hip; }

}  // extern "C"
