// meda_vec.hip -- C ABI (include/meda_vec.h) of the vectorised MEDA environment, the LDS-staged
// observation kernel, map/task/state accessors and the dispatch to the per-N kernels built from
// meda_vec_n.hip.  Device code of the transition: meda_kernels.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "meda_kernels.h"

using namespace medak;

namespace {

// ---- MEDAEnv.getOneObs (meda.py:613-674), LDS-staged --------------------------------------------------
// A workgroup owns T consecutive chips = T*n rows of 4*fov*fov+2 bytes.  The tile sits in LDS at
// the same 16-byte phase as its destination in HBM (shift = global offset & 15), so the body of the
// tile streams out with aligned 16-byte loads/stores whatever T, n and fov are.  Rows are filled by
// waves: two rows per wave pass (lanes 0-24 and 32-56 each own one cell of a 5x5 footprint); a row is
// always written by ONE wave, and LDS operations of one wave complete in order, so for overlapping
// footprints the higher droplet index wins exactly as in the reference's sequential loops.
__global__ __launch_bounds__(kBlock) void k_meda_observe(MCfg c, MPtrs p, const uint8_t *mask, int8_t *gobs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int T = c.T, E = c.E, n = c.n, fov = c.fov, ff = c.ff;
    const int tid = threadIdx.x;
    const int tile_base = blockIdx.x * T;
    const int tv = min(T, E - tile_base);
    bool all = true, any = true;
    if (mask) {
        int cnt = 0;
        for (int s = 0; s < tv; ++s) cnt += mask[tile_base + s] != 0;
        all = cnt == tv; any = cnt > 0;
    }
    if (!any) return;
    const int row_bytes = n * c.obs_len;
    const size_t g0 = (size_t)tile_base * row_bytes;
    const int shift = (int)(((uintptr_t)gobs + g0) & 15);
    const int bytes = tv * row_bytes;
    int8_t *tile = (int8_t *)smem + shift;
    uint32_t *words = (uint32_t *)(smem + (((size_t)T * row_bytes + 16 + 15) & ~(size_t)15));  // [T*n] droplet words
    for (int it = tid; it < tv * n; it += kBlock) {
        const int s = it / n, i = it - s * n;
        words[it] = p.st[(size_t)i * E + tile_base + s];
    }
    {
        uint4 *z = (uint4 *)smem;
        const uint4 zero = make_uint4(0, 0, 0, 0);
        for (int i = tid; i < (shift + bytes + 15) / 16; i += kBlock) z[i] = zero;
    }
    __syncthreads();
    const int wave = tid / kWave, lane = tid % kWave;
    const int half = lane >> 5, k = lane & 31;
    const int rows = tv * n;
    for (int r0 = wave * 2; r0 < rows; r0 += 2 * (kBlock / kWave)) {
        const int r = r0 + half;
        const bool on = (r < rows) && (k < 25);
        const int rr = r < rows ? r : rows - 1;
        const int s = rr / n, a = rr - s * n;
        const uint32_t wa = words[rr];
        const int cx = wa & 0xff, cy = (wa >> 8) & 0xff, gxa = (wa >> 16) & 0xff, gya = wa >> 24;
        const int ox = cx - fov / 2, oy = cy - fov / 2;
        const int kx = k % 5 - kR, ky = k / 5 - kR;
        int8_t *row = tile + (size_t)rr * c.obs_len;
        if (c.version == 2) {
            // ---- MEDAEnv_v0_2.getOneObs (meda.py:850-897)
            const int hf = fov / 2;
            // members of the `observed` set: droplets with at least one footprint cell inside the window
            uint32_t obs_mask = 0;
            for (int j = 0; j < n; ++j) {
                const uint32_t wj = words[s * n + j];
                const int xj = wj & 0xff, yj = (wj >> 8) & 0xff;
                const bool vis = (xj + kR >= ox) && (xj - kR <= ox + fov - 1) && (yj + kR >= oy) && (yj - kR <= oy + fov - 1);
                obs_mask |= (uint32_t)vis << j;
                if (on) {  // layer 0: every droplet, ascending index
                    const int nx = xj + kx - ox, ny = yj + ky - oy;
                    if (nx >= 0 && nx < fov && ny >= 0 && ny < fov) row[ny * fov + nx] = (int8_t)(j + 1);
                }
            }
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
            for (int tt = 0; tt < cnt; ++tt) {  // layer 1: clipped goals of the observed OTHER droplets
                const int j = (int)((order >> (4 * tt)) & 15);
                if (on && j != a) {
                    const uint32_t wj = words[s * n + j];
                    int mx = (int)((wj >> 16) & 0xff) + kx - ox, my = (int)(wj >> 24) + ky - oy;
                    mx = mx < 0 ? 0 : (mx > fov - 1 ? fov - 1 : mx);
                    my = my < 0 ? 0 : (my > fov - 1 ? fov - 1 : my);
                    row[ff + my * fov + mx] = (int8_t)(j + 1);
                }
            }
            if (r < rows) {  // layer 2: boundary bands with the reference's axis mix-up (meda.py:880-890)
                const int left = hf - cx, right = hf - (c.W - 1 - cx);
                const int up = hf - cy, down = hf - (c.L - 1 - cy);
                int c0 = 0, c1 = 0;
                if (up > 0) { c0 = 0; c1 = up < fov ? up : fov; }
                else if (down > 0) { c0 = fov - down < 0 ? 0 : fov - down; c1 = fov; }
                for (int rr = k; rr < fov; rr += 32) {
                    const bool full = left > 0 ? (rr < left) : (right > 0 ? (rr >= fov - right) : false);
                    const int b0 = full ? 0 : c0, b1 = full ? fov : c1;
                    for (int cc = b0; cc < b1; ++cc) row[2 * ff + rr * fov + cc] = 1;
                }
                if (k == 0) {
                    row[3 * ff] = p.zoom[gya - cy + 128];
                    row[3 * ff + 1] = p.zoom[256 + gxa - cx + 128];
                }
            }
            continue;
        }
        if (on) {
            {   // layer 0: own footprint
                const int nx = cx + kx - ox, ny = cy + ky - oy;
                if (nx >= 0 && nx < fov && ny >= 0 && ny < fov) row[ny * fov + nx] = (int8_t)(a + 1);
            }
            {   // layer 1: own goal
                const int nx = gxa + kx - ox, ny = gya + ky - oy;
                if (nx >= 0 && nx < fov && ny >= 0 && ny < fov) row[ff + ny * fov + nx] = (int8_t)(a + 1);
            }
            if (k == 0) { row[4 * ff] = (int8_t)(gxa - cx); row[4 * ff + 1] = (int8_t)(gya - cy); }
        }
        for (int j = 0; j < n; ++j) {  // layers 2/3: the OTHER droplets and goals, ascending index
            const uint32_t wj = words[s * n + j];
            if (on && j != a) {
                const int nx = (int)(wj & 0xff) + kx - ox, ny = (int)((wj >> 8) & 0xff) + ky - oy;
                if (nx >= 0 && nx < fov && ny >= 0 && ny < fov) row[2 * ff + ny * fov + nx] = (int8_t)(j + 1);
                int mx = (int)((wj >> 16) & 0xff) + kx - ox, my = (int)(wj >> 24) + ky - oy;
                mx = mx < 0 ? 0 : (mx > fov - 1 ? fov - 1 : mx);   // np.clip: goals outside the window smear onto its edge
                my = my < 0 ? 0 : (my > fov - 1 ? fov - 1 : my);
                row[3 * ff + my * fov + mx] = (int8_t)(j + 1);
            }
        }
    }
    __syncthreads();
    if (all) {
        // head (< 16 bytes) and tail by bytes, body by aligned 16-byte vectors
        const int head = (16 - shift) & 15;
        const int hb = head < bytes ? head : bytes;
        for (int b = tid; b < hb; b += kBlock) gobs[g0 + b] = tile[b];
        const int n16 = (bytes - hb) >> 4;
        const uint4 *src = (const uint4 *)(tile + hb);
        uint4 *dst = (uint4 *)(gobs + g0 + hb);
        for (int i = tid; i < n16; i += kBlock) dst[i] = src[i];
        for (int b = hb + (n16 << 4) + tid; b < bytes; b += kBlock) gobs[g0 + b] = tile[b];
    } else {
        for (int s = 0; s < tv; ++s)
            if (mask[tile_base + s])
                for (int b = tid; b < row_bytes; b += kBlock) gobs[g0 + (size_t)s * row_bytes + b] = tile[(size_t)s * row_bytes + b];
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
    int8_t zoom_host[512];
    int8_t *zoom_dev = nullptr;
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
    const int T = h->dc.T;
    LAUNCH(k_meda_observe, dim3((h->cfg.n_envs + T - 1) / T), dim3(kBlock), h->obs_lds, s, h->dc, h->dp, mask, obs);
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
    if ((size_t)c->n_agents * (4 * c->fov * c->fov + 2) + 64 + (size_t)c->n_agents * 4 > 60 * 1024) return MEDA_ERR_UNSUPPORTED;
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
    size_t cap = 40 * 1024;
    if (const char *v = getenv("MEDA_VEC_TILE_KB")) cap = (size_t)atoi(v) * 1024;  // tuning knob
    int T = (int)(cap / row);
    if (T < 1) T = 1;
    if (T > 32) T = 32;
    while (T > 1 && (E + T - 1) / T < 512) --T;  // keep the grid wide for small batches
    d.T = T;
    h->obs_lds = (((size_t)T * row + 16 + 15) & ~(size_t)15) + (size_t)T * n * 4;
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

int meda_vec_launch_shape(const meda_vec *h, int32_t out[2]) {
    if (!h || !out) return MEDA_ERR_BAD_ARG;
    out[0] = kBlock; out[1] = h->dc.T;
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
