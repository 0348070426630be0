// meda_kernels.h -- device code of the vectorised MEDA environment (gfx950).
//
// Replaces, for thousands of chips at once, the reference's per-chip Python:
//   MEDAEnv.step / reset / restart / getObs / addUsage / updateHealth   env/MEDA/meda.py:513-674
//   RoutingTaskManager.moveDroplets / moveOneDroplet / calPunish         env/MEDA/meda.py:241-330
//   RoutingTaskManager.addTask / _genLegalDroplet                        env/MEDA/meda.py:175-233
//
// MEDA differs from DMFB in ways that shape the kernels: moves never depend on the other droplets
// (no revert on overlap, only a proximity punishment afterwards), so the transition is a plain
// thread-per-chip kernel with fully coalesced structure-of-arrays state; the observation is 5.8 KB
// per chip (4 x 19 x 19 x n bytes) and dominates the traffic, so it has its own LDS-staged kernel
// (zero-fill, ordered footprint scatter, 16-byte coalesced stream-out).  All distance tests are on
// integer d^2 (`< 4` <=> d^2 < 16, `< 6` <=> d^2 < 36, `< 9` <=> d^2 < 81); rewards are float64 in
// the reference's operation order.  HBM-bound integer work, no MFMA.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/meda_vec.h"

namespace medak {

constexpr int kBlock = 256;
constexpr int kWave = 64;
constexpr int kR = 2;                        // droplet radius (meda.py:150)
constexpr uint32_t kTaskMaxDraws = 1u << 20; // bound on rejection sampling (same in the oracle)
enum : uint32_t { STREAM_MOVE = 1, STREAM_DEGRADE = 3, STREAM_MEDA_TASK = 5 };

struct MCfg {
    int W, L, fov, ff, obs_len, max_step, b_degrade, E, n, T, version;
    uint32_t k0, k1, env_id0;
    double per_healthy;
};
struct MPtrs {
    uint32_t *st;      // [n+4][E]: droplet words cx|cy<<8|gx<<16|gy<<24, status bits, step|failed<<16, rng_step, rng_ep
    uint32_t *starts;  // [n][E]: sx | sy<<8
    double *health;    // [E][W*L] indexed [y][x], or nullptr
    double *degrade;
    uint16_t *usage;
    uint8_t *reset_flag;  // [E] chips auto-reset by the last step (updateHealth follow-up)
    const int8_t *zoom;   // [2][256] v0_2 direction tables: round(d/(width/30)), round(d/(length/30)), index d+128
};
struct MStepArgs {
    const void *actions;
    const double *uniforms;
    const uint8_t *active;
    uint32_t flags;
    meda_vec_step_out out;
};

__device__ __forceinline__ void philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                       uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (double)(v >> 11) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ int below(uint32_t w, int n) { return (int)__umulhi(w, (uint32_t)n); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int d2(int ax, int ay, int bx, int by) { return (ax - bx) * (ax - bx) + (ay - by) * (ay - by); }

// ---- task generation: addTask x n (meda.py:175-185), serial like the reference ---------------------
template <int N>
__device__ __noinline__ void gen_task(uint32_t k0, uint32_t k1, int W, int L, uint32_t gid, uint32_t ep, int (&sx)[N], int (&sy)[N],
                                      int (&gx)[N], int (&gy)[N]) {
    // (the configuration comes in as scalars: a reference to the kernel's by-value MCfg made every lane copy the struct into scratch
    // memory at kernel entry -- 64 B of stores per chip-step in k_meda_step whether or not a task was generated)
    // Rare path (once per episode): kept as ROLLED loops over private arrays on purpose -- fully
    // unrolling the nested rejection loops makes the compile time explode for N = 9..15.
    int lx[2][N], ly[2][N];  // [0] droplets, [1] destinations
    uint32_t k = 0;
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
#pragma unroll 1
        for (int which = 0; which < 2; ++which) {
#pragma unroll 1
            for (;;) {
                uint32_t w[4];  // getRandomYX (meda.py:224-227)
                philox(k0, k1, gid, ep, k, STREAM_MEDA_TASK << 8, w);
                ++k;
                const int y = kR + below(w[0], W - 2 * kR);
                const int x = kR + below(w[1], L - 2 * kR);
                bool ok = true;  // _genLegalDroplet: not closer than 9 to an earlier box of the same list
#pragma unroll 1
                for (int j = 0; j < i; ++j) ok &= d2(x, y, lx[which][j], ly[which][j]) >= 81;
                // a destination is also redrawn while it overlaps its own droplet (meda.py:180-182)
                if (which == 1 && ok) ok = !(iabs(lx[0][i] - x) <= 2 * kR && iabs(ly[0][i] - y) <= 2 * kR);
                if (ok || k >= kTaskMaxDraws) { lx[which][i] = x; ly[which][i] = y; break; }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { sx[i] = lx[0][i]; sy[i] = ly[0][i]; gx[i] = lx[1][i]; gy[i] = ly[1][i]; }
}

template <int N>
__global__ __launch_bounds__(kBlock) void k_meda_step(MCfg c, MPtrs p, MStepArgs a) {
    const int E = c.E;
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= E) return;
    const bool maps = p.health != nullptr;
    if (a.active && !a.active[e]) {  // frozen chip: report a finished env, touch nothing
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (a.out.d_dones) a.out.d_dones[(size_t)e * N + i] = 1;
            if (a.out.d_rewards) a.out.d_rewards[(size_t)e * N + i] = 0.0;
        }
        if (a.out.d_fail) a.out.d_fail[e] = 0.0;
        if (a.out.d_success) a.out.d_success[e] = 0;
        if (a.out.d_terminated) a.out.d_terminated[e] = 1;
        if (a.out.d_team_reward) a.out.d_team_reward[e] = 0.0;
        return;
    }
    int cx[N], cy[N], gx[N], gy[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t w = p.st[(size_t)i * E + e];
        cx[i] = w & 0xff; cy[i] = (w >> 8) & 0xff; gx[i] = (w >> 16) & 0xff; gy[i] = w >> 24;
    }
    uint32_t status = p.st[(size_t)N * E + e];
    const uint32_t sw = p.st[(size_t)(N + 1) * E + e];
    uint32_t step = (sw & 0xffff) + 1, failed = sw >> 16;
    uint32_t rstep = p.st[(size_t)(N + 2) * E + e], rep = p.st[(size_t)(N + 3) * E + e];
    const int cells = c.W * c.L;
    double rew[N];
    // ---- moveDroplets / moveOneDroplet (meda.py:241-292)
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int act;
        const size_t ai = (size_t)e * N + i;
        if (a.flags & MEDA_ACT_I8) act = ((const int8_t *)a.actions)[ai];
        else if (a.flags & MEDA_ACT_I64) act = (int)((const long long *)a.actions)[ai];
        else act = ((const int32_t *)a.actions)[ai];
        if ((status >> i) & 1) { rew[i] = 0.0; continue; }
        const int old = d2(cx[i], cy[i], gx[i], gy[i]);
        if (old < 16) {  // goal reached last step: snap, status turns True now (meda.py:273-277)
            cx[i] = gx[i]; cy[i] = gy[i];
            rew[i] = 0.0;
            status |= 1u << i;
        } else {
            bool mv = true;
            if (maps || a.uniforms) {
                double prob = 1.0;
                if (maps) {  // getMoveProb (meda.py:302-309): row-major sum of the footprint / 25.0
                    prob = 0.0;
                    const double *hm = p.health + (size_t)e * cells;
                    for (int y = cy[i] - kR; y <= cy[i] + kR; ++y)
                        for (int x = cx[i] - kR; x <= cx[i] + kR; ++x) prob = prob + hm[y * c.L + x];
                    prob = prob / 25.0;
                }
                double u;
                if (a.uniforms) u = a.uniforms[ai];
                else {
                    uint32_t w[4];
                    philox(c.k0, c.k1, c.env_id0 + (uint32_t)e, rstep, (uint32_t)i, STREAM_MOVE << 8, w);
                    u = u53(w[0], w[1]);
                }
                mv = (u <= prob);
            }
            if (mv && act >= 0 && act < 8) {  // Droplet.move (meda.py:106-138); STALL returns before the clamps
                const int r = 3;
                const int dx = (act == 1) * r - (act == 3) * r + ((act == 4) | (act == 5)) * (r - 1) - ((act == 6) | (act == 7)) * (r - 1);
                const int dy = (act == 2) * r - (act == 0) * r + ((act == 5) | (act == 6)) * (r - 1) - ((act == 4) | (act == 7)) * (r - 1);
                int x = cx[i] + dx, y = cy[i] + dy;
                if (x + kR >= c.L) x = c.L - 1 - kR; else if (x - kR < 0) x = kR;
                if (y + kR >= c.W) y = c.W - 1 - kR; else if (y - kR < 0) y = kR;
                cx[i] = x; cy[i] = y;
            }
            const int nd = d2(cx[i], cy[i], gx[i], gy[i]);
            rew[i] = nd < 16 ? 0.0 : (nd == old && act == 8) ? -0.2 : (nd < old) ? -0.08 : -0.4;
        }
    }
    rstep += 1;
    // ---- calPunish (meda.py:321-330): -0.6 for both droplets of every pair closer than 6
    // punish[i] starts at 0 and has 0.6 subtracted once per near pair, sequentially (meda.py:328-329):
    // the value depends only on the pair count, so count in integers and subtract in a rolled loop.
    int near_cnt[N];
    bool any = false;
#pragma unroll
    for (int i = 0; i < N; ++i) near_cnt[i] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = i + 1; j < N; ++j) {
            const bool near = d2(cx[i], cy[i], cx[j], cy[j]) < 36;
            near_cnt[i] += near; near_cnt[j] += near; any |= near;
        }
    double punish[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double pz = 0.0;
#pragma unroll 1
        for (int k = 0; k < near_cnt[i]; ++k) pz = pz - 0.6;
        punish[i] = pz;
    }
    double fail;  // np.sum(punish): numpy pairwise order
    if constexpr (N < 8) {
        fail = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) fail = fail + punish[i];
    } else {
        double q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = punish[j];
        constexpr int M = N - (N % 8);
#pragma unroll
        for (int i = 8; i < M; i += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = q[j] + punish[i + j];
        fail = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
#pragma unroll
        for (int i = M; i < N; ++i) fail = fail + punish[i];
    }
    if (any) failed = 1;
    const bool all = status == ((N == 32) ? 0xffffffffu : ((1u << N) - 1u));
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double v = rew[i] + punish[i];
        if (all) { v = v + 3.0; if (!failed) v = v + 3.0; }  // MEDAEnv.step (meda.py:522-525)
        rew[i] = v;
    }
    const bool in_time = (int)step < c.max_step;
    const bool success = in_time && all && !failed;
    bool term = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const bool d = in_time ? ((status >> i) & 1) : true;
        term &= d;
        if (a.out.d_dones) a.out.d_dones[(size_t)e * N + i] = (uint8_t)d;
        if (a.out.d_rewards) a.out.d_rewards[(size_t)e * N + i] = rew[i];
    }
    if (in_time && maps) {  // addUsage (meda.py:591-598): footprints of the agents that are not done
        uint16_t *um = p.usage + (size_t)e * cells;
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (!((status >> i) & 1))
                for (int y = cy[i] - kR; y <= cy[i] + kR; ++y)
                    for (int x = cx[i] - kR; x <= cx[i] + kR; ++x) um[y * c.L + x] += 1;
    }
    if (a.out.d_fail) a.out.d_fail[e] = fail;
    if (a.out.d_success) a.out.d_success[e] = (uint8_t)success;
    if (a.out.d_terminated) a.out.d_terminated[e] = (uint8_t)term;
    if (a.out.d_team_reward) {
        double s;
        if constexpr (N < 8) {
            s = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) s = s + rew[i];
        } else {
            double q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = rew[j];
            constexpr int M = N - (N % 8);
#pragma unroll
            for (int i = 8; i < M; i += 8)
#pragma unroll
                for (int j = 0; j < 8; ++j) q[j] = q[j] + rew[i + j];
            s = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
#pragma unroll
            for (int i = M; i < N; ++i) s = s + rew[i];
        }
        a.out.d_team_reward[e] = s / (double)N;
    }
    if (term && (a.flags & MEDA_STEP_AUTORESET)) {  // MEDAEnv.reset() inside the launch (meda.py:541-550)
        // gen_task is a real call taking its four arrays by reference: they live in scratch memory.  Handing it gx / gy themselves
        // put the GOALS OF EVERY TRANSITION there (224 B of scratch per lane: 96 B of extra WRITE_SIZE and 36 B of FETCH_SIZE per
        // chip-step, tools/probe/meda_step_writes.py); with arrays of its own only a lane that resets touches scratch.
        int sx[N], sy[N], tgx[N], tgy[N];
        gen_task<N>(c.k0, c.k1, c.W, c.L, c.env_id0 + (uint32_t)e, rep, sx, sy, tgx, tgy);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            gx[i] = tgx[i]; gy[i] = tgy[i];
            cx[i] = sx[i]; cy[i] = sy[i];
            p.starts[(size_t)i * E + e] = (uint32_t)sx[i] | ((uint32_t)sy[i] << 8);
        }
        rep += 1; step = 0; failed = 0; status = 0;
        if (p.reset_flag) p.reset_flag[e] = 1;
    }
    // WHOLE words go back: the compiler sees that a droplet word's goal half (and the `failed` half of the step word) is the value
    // it loaded and would store the changed 16 bits alone (global_store_short).  A wave of 2-byte stores leaves every 32-byte
    // sector half dirty; the L2 then fills and writes back whole lines for them: the record cost 128 B of WRITE_SIZE per chip-step
    // instead of 32 and doubled the kernel's FETCH_SIZE (tools/probe/meda_step_writes.py).  The empty asm makes each word opaque.
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint32_t w = (uint32_t)cx[i] | ((uint32_t)cy[i] << 8) | ((uint32_t)gx[i] << 16) | ((uint32_t)gy[i] << 24);
        asm volatile("" : "+v"(w));
        p.st[(size_t)i * E + e] = w;
    }
    uint32_t sw2 = (step & 0xffff) | (failed << 16);
    asm volatile("" : "+v"(sw2));
    p.st[(size_t)N * E + e] = status;
    p.st[(size_t)(N + 1) * E + e] = sw2;
    p.st[(size_t)(N + 2) * E + e] = rstep;
    if (term && (a.flags & MEDA_STEP_AUTORESET)) p.st[(size_t)(N + 3) * E + e] = rep;   // rng_ep changes only with a new task
}

// mode 0: reset()  2: restart()  3: create
template <int N>
__global__ __launch_bounds__(kBlock) void k_meda_reset(MCfg c, MPtrs p, const uint8_t *mask, int mode) {
    const int E = c.E;
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= E) return;
    if (mask && !mask[e]) return;
    if (mode == 2) {  // RoutingTaskManager.restart (meda.py:170-173) + step_count = 0; `fails` kept
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint32_t s = p.starts[(size_t)i * E + e];
            const uint32_t w = p.st[(size_t)i * E + e];
            p.st[(size_t)i * E + e] = (s & 0xffff) | (w & 0xffff0000u);
        }
        p.st[(size_t)N * E + e] = 0;
        p.st[(size_t)(N + 1) * E + e] &= 0xffff0000u;
        return;
    }
    const uint32_t rep = mode == 3 ? 0u : p.st[(size_t)(N + 3) * E + e];
    int sx[N], sy[N], gx[N], gy[N];
    gen_task<N>(c.k0, c.k1, c.W, c.L, c.env_id0 + (uint32_t)e, rep, sx, sy, gx, gy);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        p.starts[(size_t)i * E + e] = (uint32_t)sx[i] | ((uint32_t)sy[i] << 8);
        p.st[(size_t)i * E + e] = (uint32_t)sx[i] | ((uint32_t)sy[i] << 8) | ((uint32_t)gx[i] << 16) | ((uint32_t)gy[i] << 24);
    }
    p.st[(size_t)N * E + e] = 0;
    p.st[(size_t)(N + 1) * E + e] = 0;
    if (mode == 3) p.st[(size_t)(N + 2) * E + e] = 0;
    p.st[(size_t)(N + 3) * E + e] = rep + 1;
    if (p.reset_flag) p.reset_flag[e] = 1;
}

template <int N>
hipError_t launch_meda_step_n(const MCfg &c, const MPtrs &p, const MStepArgs &a, hipStream_t s);
template <int N>
hipError_t launch_meda_reset_n(const MCfg &c, const MPtrs &p, const uint8_t *mask, int mode, hipStream_t s);

}  // namespace medak
