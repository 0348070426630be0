// dmfb_kernels.h -- device code of the vectorised DMFB environment (gfx950).
//
// Included by dmfb_vec_n.hip (one translation unit per droplet count N, so the 16 instantiations
// build in parallel) and by dmfb_vec.hip (C ABI, N-independent kernels, dispatch).
//
// What runs here replaces, for thousands of chips at once, the reference's per-chip Python:
//   DMFBenv.step / reset / restart / getObs          env/DMFB/dmfb.py:560-626
//   RoutingTaskManager.moveDroplets / moveOneDroplet  env/DMFB/dmfb.py:253-359
//   RoutingTaskManager.getOneObs                      env/DMFB/dmfb.py:395-457
//   RoutingTaskManager.addUsage / updateHealth        env/DMFB/dmfb.py:459-471
//   RoutingTaskManager._Generate_Start_End            env/DMFB/dmfb.py:207-226
//
// Design (DESIGN.md has the long form):
//   * State is structure-of-arrays in HBM: word w of env e lives at st[w*E + e], so a wave that
//     owns 64 consecutive envs reads/writes 256 contiguous bytes per word.
//   * One workgroup (256 threads) owns a tile of T consecutive envs.  Wave 0 runs the
//     transition, one lane per env; droplets are moved serially in index order inside the lane
//     (the reference's semantics are order dependent).  Episode ends are resolved with a wave
//     ballot; new tasks are found by all 64 lanes testing 64 rejection-sampling attempts at once.
//   * Observations (89 % of the bytes) are assembled in LDS: the tile's rows are zero-filled with
//     16-byte LDS stores, the few non-zero cells are scattered with byte stores, then the tile is
//     streamed to HBM with 16-byte-per-lane coalesced stores.
//   * Pure integer work apart from the health compare and the reward sums, which are float64
//     so that results are bit-identical to the reference.  No MFMA.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dmfb_vec.h"

namespace dmfbk {

constexpr int kBlock = 256;
constexpr int kObsBlock = 256;  // threads per workgroup of k_observe (two lanes per observation row, see scatter_rows)
constexpr int kWave = 64;
// Rejection sampling is bounded so that every wave terminates: after 2^22 rejected attempts the
// last attempt's points are kept as they are (same rule in the oracle; never hit by legal configs
// of practical density).
constexpr uint32_t kTaskMaxRounds = 1u << 16;

enum : uint32_t { STREAM_MOVE = 1, STREAM_TASK = 2, STREAM_DEGRADE = 3, STREAM_BLOCK = 4 };

struct DevCfg {
    int W, L, fov, hf, ff, obs_len, max_step, stall, b_degrade, E, n;
    int nb;     // obstacle blocks per chip currently in force (0 when none / skipped by the density rule)
    int T;      // chips per workgroup of the launch being made (k_step)
    int T_obs;  // chips per workgroup of k_observe
    uint32_t k0, k1, env_id0;
    uint32_t fov_magic;  // ceil(2^32 / fov): k / fov == __umulhi(k, fov_magic) for the ranges used (fov >= 2)
    int nq;              // 8-byte words per band image, padded to a multiple of 12 (see DevPtrs::band)
    int ucap;            // steps a chip's usage log holds (= max_step), see DevPtrs::ulog
    int lstride;         // uint16 entries per logged step: 16 (one aligned 32-byte sector per step, unused ones 0xFFFF)
    int hist_bytes;      // LDS bytes of one wave's usage histogram, 0 = chip too large for LDS (global-atomic path)
    double per_healthy;
};

struct DevPtrs {
    uint32_t *st;      // [NW][E] packed env records
    uint32_t *starts;  // [NP][E] packed start cells
    double *health;    // [E][W*L] or nullptr
    double *degrade;   // [E][W*L]
    uint16_t *usage;   // [E][W*L]
    // addUsage (dmfb.py:459-463) as an append-only log: step k of a chip since its last flush stores, per droplet, the cell
    // index x*L+y it occupies (0xFFFF: the droplet is on its goal, or the slot is beyond n) at ulog[(e*ucap + k)*16 + i].
    // A transition writes ONE aligned 32-byte sector instead of n read-modify-writes scattered over a sector each (and no
    // partial-sector write, which HBM3 turns into a read-modify-write); the log is folded into the usage map (flush_usage)
    // when the episode is reset, when it is full, and before the map is read or written from outside.
    uint16_t *ulog;
    // How often updateHealth (dmfb.py:465-471) has degraded each cell since the maps were generated, saturating at 15:
    // 4 bits per cell, [E][kmap_bytes(W*L)] (cell c = nibble c & 1 of byte c >> 1).  m_health[cell] is then 1.0 multiplied
    // `k` times by m_degrade[cell] -- the very products the reference forms -- and m_degrade[cell] is a pure function of
    // the Philox key (gen_degrade_env), so a transition reads one byte per droplet out of a map of W*L/2 bytes (a 20x20
    // chip: 200 bytes, under two 128-byte lines) instead of one float64 from a 128-byte line each.  A saturated count
    // (health <= 0.6^15 of a degrading cell, or 15+ degradations) falls back to the float64 map for that cell.
    uint8_t *kmap;
    // dflags[0] != 0: the maps are the generator's own (never replaced through set_map), so the transition may rebuild a
    // cell's health from `kmap` instead of gathering the float64 map.  A DEVICE word, written by stream-ordered one-thread
    // launches (dmfb_vec_set_map / reset(new)), so that a transition captured into a HIP graph before a map was injected
    // sees the switch when it is replayed (a by-value kernel argument would be frozen at capture time).
    const int *dflags;
    const int8_t *zoom;  // [2][511] direction zoom table
    uint32_t *blocks;    // [n_blocks][E] x_min | x_max<<8 | y_min<<16 | y_max<<24, or nullptr
    // Observation tables, copied into LDS by every workgroup that builds observations (table_words() 8-byte words):
    //   [axis 2][pattern 2*hf+1][nq] out-of-chip band images of layer 2 (dmfb.py:428-439): the fov*fov layer bytes of
    //   one axis' band, zero padded to nq words.  Pattern 0 = window inside the chip on that axis, p in 1..hf = `p`
    //   leading window rows/columns outside, hf+p = `p` trailing ones.  A row ORs (X image | Y image), shifted to the byte
    //   phase of its layer in LDS, into the zeroed tile with aligned 8-byte LDS atomics.
    //   then 128 words = the direction zoom table int8[2][511] (dmfb.py:444-453), padded to 1024 bytes.
    const unsigned long long *band;
    unsigned long long *dbg;  // diagnostic builds (-DDMFB_STAMPS) only: per-workgroup phase time stamps of k_observe
};

// In-kernel time stamps exist only in the diagnostic build (make stamps -> lib/libdmfb_vec_stamps.so, tools/exp_stamps.py);
// the shipped kernels contain none.
#ifdef DMFB_STAMPS
// accumulates, per workgroup, the cycles between consecutive stamps: dbg[wg][k] += now - previous stamp
#define DMFB_STAMP(k)                                                                               \
    do {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        if (threadIdx.x == 0 && p.dbg) {                                                            \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                           \
            p.dbg[(size_t)blockIdx.x * 8 + (k)] += now_ - stamp_prev_;                              \
            stamp_prev_ = now_;                                                                     \
        }                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    } while (0)
#define DMFB_STAMP_INIT() unsigned long long stamp_prev_ = __builtin_amdgcn_s_memtime()
#else
#define DMFB_STAMP(k) do { } while (0)
#define DMFB_STAMP_INIT() do { } while (0)
#endif

// ---- packed record layout -------------------------------------------------------------------
// NP = ceil(N/2) words of positions (agent i: word i>>1, half i&1, value x | y<<8),
// NP words of goals, then: step_count | flags<<16 | usage-log length<<20, cumulative constraints, rng_step, rng_ep, rng_map.
template <int N> struct Rec {
    static constexpr int NP = (N + 1) / 2;
    static constexpr int W_POS = 0, W_GOAL = NP, W_STEP = 2 * NP, W_CUM = 2 * NP + 1,
                         W_RSTEP = 2 * NP + 2, W_REP = 2 * NP + 3, W_RMAP = 2 * NP + 4, NW = 2 * NP + 5;
};
constexpr uint32_t FLAG_DUP = 1u;  // two droplets share a cell (only reachable through set_task)
constexpr uint32_t kFlagMask = 0xfu;   // bits 16..19 of the step word
constexpr int kUlenShift = 20;         // bits 20..31: steps in the chip's usage log (<= max_step <= 1020)

__host__ __device__ inline int rec_words(int n) { return 2 * ((n + 1) / 2) + 5; }
__host__ __device__ inline int table_words(int hf, int nq) { return 2 * (2 * hf + 1) * nq + 128; }

// ---- Philox4x32-10 ---------------------------------------------------------------------------
__device__ __forceinline__ void philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                                       uint32_t c3, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;  // one v_mad_u64_u32 each
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
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
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }

// ---- per-lane env registers -------------------------------------------------------------------
// Cells are kept packed, x | y << 16 (two int16 halves): the pair tests of the transition are packed 16-bit arithmetic, the
// observation tile wants this very format, and a chip's cells take 2N registers instead of 4N.
template <int N> struct EnvR {
    uint32_t pos[N], goal[N];
    uint32_t step, flags, ulen, cum, rstep, rep, rmap;
};
__device__ __forceinline__ uint32_t xy16(uint32_t b) { return (b & 0xffu) | ((b & 0xff00u) << 8); }          // x | y<<8  -> x | y<<16
__device__ __forceinline__ uint32_t xy8(uint32_t c) { return (c & 0xffu) | ((c >> 8) & 0xff00u); }            // x | y<<16 -> x | y<<8
__device__ __forceinline__ int cell_x(uint32_t c) { return (int)(c & 0xffffu); }
__device__ __forceinline__ int cell_y(uint32_t c) { return (int)(c >> 16); }

template <int N>
__device__ __forceinline__ void load_env(const DevPtrs &p, int E, int e, EnvR<N> &r, bool want_rstep = true, bool want_rmap = true,
                                         bool want_rep = false) {
    using R = Rec<N>;
#pragma unroll
    for (int w = 0; w < R::NP; ++w) {
        const uint32_t pw = p.st[(size_t)(R::W_POS + w) * E + e];
        const uint32_t gw = p.st[(size_t)(R::W_GOAL + w) * E + e];
        r.pos[2 * w] = xy16(pw); r.goal[2 * w] = xy16(gw);
        if (2 * w + 1 < N) { r.pos[2 * w + 1] = xy16(pw >> 16); r.goal[2 * w + 1] = xy16(gw >> 16); }
    }
    uint32_t s = p.st[(size_t)R::W_STEP * E + e];
    r.step = s & 0xffff; r.flags = (s >> 16) & kFlagMask; r.ulen = s >> kUlenShift;
    r.cum = p.st[(size_t)R::W_CUM * E + e];
    r.rstep = want_rstep ? p.st[(size_t)R::W_RSTEP * E + e] : 0u;
    r.rep = want_rep ? p.st[(size_t)R::W_REP * E + e] : 0u;
    r.rmap = want_rmap ? p.st[(size_t)R::W_RMAP * E + e] : 0u;
}

// record word w of a packed-cell array: agent 2w in the low half, 2w+1 in the high half, each x | y << 8
template <int N> __device__ __forceinline__ uint32_t pack_pos(const uint32_t (&cells)[N], int w) {
    uint32_t v = xy8(cells[2 * w]);
    if (2 * w + 1 < N) v |= xy8(cells[2 * w + 1]) << 16;
    return v;
}

// with_goal: the lane's episode ended and a new task was drawn (goals and rng_ep change only then); rng_map never changes in a
// transition; with_rstep: see k_step
template <int N>
__device__ __forceinline__ void store_env(const DevPtrs &p, int E, int e, const EnvR<N> &r, bool with_goal, bool with_rstep = true) {
    using R = Rec<N>;
#pragma unroll
    for (int w = 0; w < R::NP; ++w) {
        p.st[(size_t)(R::W_POS + w) * E + e] = pack_pos<N>(r.pos, w);
        if (with_goal) p.st[(size_t)(R::W_GOAL + w) * E + e] = pack_pos<N>(r.goal, w);
    }
    p.st[(size_t)R::W_STEP * E + e] = (r.step & 0xffff) | (r.flags << 16) | (r.ulen << kUlenShift);
    p.st[(size_t)R::W_CUM * E + e] = r.cum;
    if (with_rstep) p.st[(size_t)R::W_RSTEP * E + e] = r.rstep;
    if (with_goal) p.st[(size_t)R::W_REP * E + e] = r.rep;
}

template <int N> __device__ __forceinline__ bool any_dup(const EnvR<N> &r) {
    bool d = false;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = i + 1; j < N; ++j) d |= r.pos[i] == r.pos[j];
    return d;
}

// ---- task generation: _Generate_Start_End (dmfb.py:207-226) -----------------------------------
// Executed by a FULL wave with wave-uniform (env_gid, ep).  Lane l tests attempt 64*round + l;
// the lowest accepted attempt wins, which is exactly the first accepted attempt of a serial loop.
// On return every lane holds the winner's 2N points: pts[b] = point 2b | point (2b+1) << 16.
template <int N>
__device__ __forceinline__ void gen_task_wave(const DevCfg &c, uint32_t env_gid, uint32_t ep, uint32_t (&pts)[N]) {
    const int lane = lane_id();
    for (uint32_t round = 0;; ++round) {
        const bool last = round == kTaskMaxRounds - 1;  // bounded: the final attempt is kept unchecked
        const uint32_t attempt = round * kWave + lane;
        uint32_t mine[N];
#pragma unroll
        for (int b = 0; b < N; ++b) {
            uint32_t w[4];
            philox(c.k0, c.k1, env_gid, ep, attempt, (STREAM_TASK << 8) | (uint32_t)b, w);
            uint32_t p0 = (uint32_t)below(w[1], c.W) | ((uint32_t)below(w[0], c.L) << 8);
            uint32_t p1 = (uint32_t)below(w[3], c.W) | ((uint32_t)below(w[2], c.L) << 8);
            mine[b] = p0 | (p1 << 16);
        }
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 2 * N; ++i) {
            const uint32_t pi = (mine[i >> 1] >> (16 * (i & 1))) & 0xffff;
            const int xi = pi & 0xff, yi = pi >> 8;
#pragma unroll
            for (int j = i + 1; j < 2 * N; ++j) {
                const uint32_t pj = (mine[j >> 1] >> (16 * (j & 1))) & 0xffff;
                const int dx = xi - (int)(pj & 0xff), dy = yi - (int)(pj >> 8);
                ok &= (dx * dx + dy * dy > 2);
            }
        }
        const unsigned long long acc = __ballot(ok) | (last ? (1ull << (kWave - 1)) : 0ull);
        if (acc) {
            const int win = __ffsll((long long)acc) - 1;
#pragma unroll
            for (int b = 0; b < N; ++b) pts[b] = (uint32_t)__shfl((int)mine[b], win, kWave);
            return;
        }
    }
}

// the accepted 2N points (point i = start of droplet i, point N + i = its goal, each x | y << 8) become the chip's cells
template <int N> __device__ __forceinline__ void task_to_env(const uint32_t (&pts)[N], EnvR<N> &r) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int gi = N + i;
        r.pos[i] = xy16((pts[i >> 1] >> (16 * (i & 1))) & 0xffff);
        r.goal[i] = xy16((pts[gi >> 1] >> (16 * (gi & 1))) & 0xffff);
    }
}

// ---- obstacle generation: GenRandomBlocks (dmfb.py:228-251) --------------------------------------
// Executed by a FULL wave after gen_task_wave (pts = the accepted 2N points, same in every lane).
// Blocks are drawn one after another; for block b all 64 lanes test 64 attempts at once and the
// lowest accepted attempt wins (= the serial loop).  Lane b keeps block b; returns this lane's block.
template <int N>
__device__ __forceinline__ uint32_t gen_blocks_wave(const DevCfg &c, uint32_t env_gid, uint32_t ep, const uint32_t (&pts)[N]) {
    const int lane = lane_id();
    uint32_t myblk = 0;
    for (int b = 0; b < c.nb; ++b) {
        for (uint32_t round = 0;; ++round) {
            const bool last = round == kTaskMaxRounds - 1;
            const uint32_t attempt = round * kWave + lane;
            uint32_t w[4];
            philox(c.k0, c.k1, env_gid, ep, attempt, (STREAM_BLOCK << 8) | (uint32_t)b, w);
            const int y0 = below(w[0], c.L - 3), x0 = below(w[1], c.W - 3);
            bool bad = false;
#pragma unroll
            for (int i = 0; i < 2 * N; ++i) {  // Block.isPointInside over starts and ends
                const uint32_t pi = (pts[i >> 1] >> (16 * (i & 1))) & 0xffff;
                const int px = pi & 0xff, py = pi >> 8;
                bad |= (px >= x0) & (px <= x0 + 1) & (py >= y0) & (py <= y0 + 1);
            }
            for (int k = 0; k < b; ++k) {  // Block.isBlockOverlap with the earlier blocks (dmfb.py:56-69)
                const uint32_t o = (uint32_t)__shfl((int)myblk, k, kWave);
                const int ox0 = o & 0xff, ox1 = (o >> 8) & 0xff, oy0 = (o >> 16) & 0xff, oy1 = o >> 24;
                bad |= !(x0 > ox1 || ox0 > x0 + 1) && !(y0 > oy1 || oy0 > y0 + 1);
            }
            const unsigned long long acc = __ballot(!bad) | (last ? (1ull << (kWave - 1)) : 0ull);
            if (acc) {
                const int win = __ffsll((long long)acc) - 1;
                const uint32_t cand = (uint32_t)x0 | ((uint32_t)(x0 + 1) << 8) | ((uint32_t)y0 << 16) | ((uint32_t)(y0 + 1) << 24);
                const uint32_t chosen = (uint32_t)__shfl((int)cand, win, kWave);
                if (lane == b) myblk = chosen;
                break;
            }
        }
    }
    return myblk;
}

template <int N> __device__ __forceinline__ void store_starts(const DevPtrs &p, int E, int e, const uint32_t (&cells)[N]) {
#pragma unroll
    for (int w = 0; w < Rec<N>::NP; ++w) p.starts[(size_t)w * E + e] = pack_pos<N>(cells, w);
}

// ---- health maps ------------------------------------------------------------------------------
// updateHealth (dmfb.py:465-471) for one env, strided over `nthreads` cooperating threads.
__device__ __forceinline__ void update_health_env(const DevPtrs &p, int cells, int e, int tid, int nthreads) {
    const size_t base = (size_t)e * cells;
    for (int cidx = tid; cidx < cells; cidx += nthreads) {
        if (p.usage[base + cidx] > 50) {
            p.health[base + cidx] = p.health[base + cidx] * p.degrade[base + cidx];
            p.usage[base + cidx] = 0;
        }
    }
}
// m_degrade of one cell (the value gen_degrade_env stores) and m_health rebuilt from the degrade count
__device__ __forceinline__ double degrade_of(const DevCfg &c, uint32_t env_gid, uint32_t rmap, int cell) {
    if (!c.b_degrade) return 1.0;
    uint32_t w[4];
    philox(c.k0, c.k1, env_gid, rmap, (uint32_t)cell, STREAM_DEGRADE << 8, w);
    const double d = u53(w[0], w[1]) * 0.4 + 0.6;
    return (u53(w[2], w[3]) < c.per_healthy) ? 1.0 : d;
}
constexpr int kCountMax = 15;  // saturation value of a degrade count (4 bits)
__host__ __device__ inline size_t kmap_bytes(size_t cells) { return ((cells + 1) / 2 + 3) & ~(size_t)3; }  // per chip, whole 32-bit words
__device__ __forceinline__ int kmap_get(const DevPtrs &p, size_t kb, int e, int cell) {
    return (p.kmap[(size_t)e * kb + (cell >> 1)] >> (4 * (cell & 1))) & kCountMax;
}
// count of one cell + 1 (the caller's lane is the only writer of this nibble; the other seven nibbles of the word may
// belong to other lanes of the wave: 32-bit atomic add, no carry because the nibble is below 15)
__device__ __forceinline__ void kmap_bump(const DevPtrs &p, size_t kb, int e, int cell) {
    if (kmap_get(p, kb, e, cell) == kCountMax) return;
    const size_t byte = (size_t)e * kb + (cell >> 1);
    atomicAdd((uint32_t *)p.kmap + (byte >> 2), 1u << (8 * (byte & 3) + 4 * (cell & 1)));
}
__device__ __forceinline__ double health_from_count(double d, int k) {
    double h = 1.0;
    for (int t = 0; t < k; ++t) h = h * d;  // the reference's products, in its order (dmfb.py:469)
    return h;
}

// _random_health_statue (dmfb.py:157-166) for one env
__device__ __forceinline__ void gen_degrade_env(const DevCfg &c, const DevPtrs &p, int cells, int e, uint32_t rmap,
                                                int tid, int nthreads) {
    const size_t base = (size_t)e * cells;
    for (int cidx = tid; cidx < cells; cidx += nthreads) {
        p.degrade[base + cidx] = degrade_of(c, c.env_id0 + (uint32_t)e, rmap, cidx);
    }
}

// Fold a chip's usage log into its usage map and, if `update`, run updateHealth (dmfb.py:465-471) on the result.
// Executed by ONE wave; `hist` = c.hist_bytes of LDS owned by the wave (nullptr when the chip is too large: the counts
// then go straight to the map with 32-bit global atomics on the containing words).  Global traffic: the log read once,
// the map read and written once, both contiguous.
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
__device__ __forceinline__ void flush_usage(const DevCfg &c, const DevPtrs &p, int e, int ulen, bool update, uint16_t *hist, int lane) {
    const int cells = c.W * c.L;
    const size_t base = (size_t)e * cells, kb = kmap_bytes(cells);
    const uint16_t *log = p.ulog + (size_t)e * c.ucap * c.lstride;
    const int entries = ulen * c.lstride;
    if (hist) {
        uint32_t *h32 = (uint32_t *)hist;
        for (int k = lane; k < cells; k += kWave) hist[k] = p.usage[base + k];
        wave_fence();
        for (int k = lane; k < entries; k += kWave) {
            const uint32_t cell = log[k];
            if (cell != 0xffffu) atomicAdd(&h32[cell >> 1], 1u << (16 * (cell & 1)));
        }
        wave_fence();
        for (int k = lane; k < cells; k += kWave) {
            uint16_t u = hist[k];
            if (update && u > 50) {
                p.health[base + k] = p.health[base + k] * p.degrade[base + k];
                kmap_bump(p, kb, e, k);
                u = 0;
            }
            p.usage[base + k] = u;
        }
        wave_fence();
        return;
    }
    uint32_t *u32 = (uint32_t *)p.usage;  // hipMalloc'ed: 4-byte aligned
    for (int k = lane; k < entries; k += kWave) {
        const uint32_t cell = log[k];
        if (cell != 0xffffu) {
            const size_t g = base + cell;
            atomicAdd(&u32[g >> 1], 1u << (16 * (g & 1)));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    if (!update) return;
    for (int k = lane; k < cells; k += kWave) {
        const size_t g = base + k;
        const uint32_t w = __hip_atomic_load(&u32[g >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // not through this CU's L1
        if (((w >> (16 * (g & 1))) & 0xffffu) > 50) {
            p.health[g] = p.health[g] * p.degrade[g];
            kmap_bump(p, kb, e, k);
            atomicAnd(&u32[g >> 1], (g & 1) ? 0x0000ffffu : 0xffff0000u);  // the neighbouring cell may belong to another lane
        }
    }
}

// ---- LDS tile -----------------------------------------------------------------------------------
struct Tile {
    int8_t *obs;       // [T][N][obs_len], 16-byte aligned (absent in the step-only launch)
    uint32_t *pos;     // [T][N] x | y << 16 (two int16 lanes: the window tests use packed 16-bit arithmetic)
    uint32_t *goal;    // [T][N]
    uint8_t *flag;     // [T] per-env flag (ended / masked)
    unsigned long long *tab;  // LDS copy of DevPtrs::band (band images, then the zoom table)
    uint16_t *ulen;           // [T] usage-log length of the chips whose log must be folded in after the step
};
__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
// LDS bytes of a workgroup: obs block (0 for the step-only launch) + pos/goal + flags
// The obs block sits in LDS at the SAME 16-byte phase as its destination in HBM (shift = offset & 15), so
// the body of a tile streams out with aligned 16-byte accesses for ANY tile size (16 bytes of slack).
__host__ __device__ inline size_t tile_lds_bytes(int T, int n, int obs_len, bool with_obs, int tab_words) {
    // observation tile + tables, positions/goals, flags, and (observation kernel) a second positions/goals buffer
    return (with_obs ? align16((size_t)T * n * obs_len + 16) + (size_t)tab_words * 8 + (size_t)T * n * 8 : 0) + (size_t)T * n * 8 +
           align16((size_t)T) + align16((size_t)2 * T);
}
__device__ __forceinline__ Tile carve(unsigned char *smem, int T, int n, int obs_len, bool with_obs, int tab_words, int shift = 0) {
    Tile t;
    t.obs = (int8_t *)smem + shift;
    size_t off = with_obs ? align16((size_t)T * n * obs_len + 16) : 0;
    t.tab = (unsigned long long *)(smem + off);
    if (with_obs) off += (size_t)tab_words * 8;
    t.pos = (uint32_t *)(smem + off);
    t.goal = t.pos + (size_t)T * n;
    t.flag = (uint8_t *)(t.goal + (size_t)T * n);
    t.ulen = (uint16_t *)(t.flag + align16((size_t)T) + (with_obs ? (size_t)T * n * 8 : 0));
    return t;
}
// copy the observation tables into LDS (any subset of the workgroup's threads; a barrier follows before they are read)
__device__ __forceinline__ void load_tables(const DevCfg &c, const DevPtrs &p, const Tile &t, int tid, int nthreads) {
    const int words = table_words(c.hf, c.nq);
    for (int i = tid; i < words; i += nthreads) t.tab[i] = p.band[i];
}

__device__ __forceinline__ void zero_tile(unsigned char *smem, int bytes16, int tid, int nthreads) {
    uint4 *p = (uint4 *)smem;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < bytes16; i += nthreads) p[i] = z;
}

// ---- packed int16 pairs (x in the low half, y in the high half) ------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b));
}
__device__ __forceinline__ uint32_t pk_min_u(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_clamp_i(uint32_t a, uint32_t lo, uint32_t hi) {
    s16x2 v = __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, lo));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(v, __builtin_bit_cast(s16x2, hi)));
}
__device__ __forceinline__ uint32_t pk_abs(uint32_t a) {
    const s16x2 v = __builtin_bit_cast(s16x2, a);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(v, -v));
}
__device__ __forceinline__ uint32_t pk2(int v) { return ((uint32_t)v & 0xffffu) * 0x10001u; }
__device__ __forceinline__ uint32_t to_xy(uint32_t byte_pair) { return (byte_pair & 0xffu) | ((byte_pair & 0xff00u) << 8); }

// getOneObs (dmfb.py:395-457) for the tile, phase 1 of 2 (runs between the zero-fill and the byte stores of
// scatter_rows, a barrier on either side): layer 2, the out-of-chip bands (dmfb.py:428-439).  One lane per row ORs
// the row's band image -- X-axis image | Y-axis image from DevPtrs::band, selected by how far the window hangs over the
// chip border and by the byte phase of the row's layer in LDS -- into the zeroed tile with aligned 8-byte LDS atomics
// (neighbouring rows share the 8-byte words at their seams, hence atomics; no byte store is in flight in this phase).
template <int N>
__device__ __forceinline__ void scatter_bands(const DevCfg &c, const Tile &t, unsigned char *smem, int tv, int tid, int nthreads) {
    const int hf = c.hf, nq = c.nq, npat = 2 * hf + 1;
    const int rows = tv * N;
    const int half = nthreads >> 1;  // lanes [0, half): image words [0, nq/2) of row `lane`; lanes [half, ..): the other half
    const int obs0 = (int)((unsigned char *)t.obs - smem);  // LDS byte offset of the tile's first row (smem is 16-byte aligned)
    const int part = tid >= half ? 1 : 0;
    for (int it = tid - part * half; it < rows; it += half) {
        const uint32_t pc = t.pos[it];
        const int cx = (int)(pc & 0xffff), cy = (int)(pc >> 16);
        const int left = hf - cx, right = hf - (c.W - 1 - cx);
        const int up = hf - cy, down = hf - (c.L - 1 - cy);
        const int xpat = left > 0 ? left : (right > 0 ? hf + right : 0);
        const int ypat = up > 0 ? up : (down > 0 ? hf + down : 0);
        if ((xpat | ypat) == 0) continue;  // window inside the chip: layer 2 stays zero
        const int a = obs0 + it * c.obs_len + 2 * c.ff;  // LDS byte offset of this row's layer 2
        const int sh = (a & 7) * 8;                      // bit shift that moves the image to the layer's byte phase
        const int qa = part * (nq >> 1), qb = qa + (nq >> 1);  // nq is a multiple of 12: six-word batches
        unsigned long long *dst = (unsigned long long *)(smem + (a & ~7));
        const unsigned long long *tx = t.tab + (size_t)xpat * nq, *ty = t.tab + (size_t)(npat + ypat) * nq;
        unsigned long long prev = qa ? (tx[qa - 1] | ty[qa - 1]) : 0ull;  // its top bytes shift into this lane's first word
        for (int q0 = qa; q0 < qb; q0 += 6) {  // twelve LDS reads in flight, then the atomics
            unsigned long long v[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) v[k] = tx[q0 + k] | ty[q0 + k];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const unsigned long long o = sh ? (v[k] << sh) | (prev >> (64 - sh)) : v[k];
                prev = v[k];
                if (o) atomicOr(dst + q0 + k, o);
            }
        }
        if (part && sh && (prev >> (64 - sh))) atomicOr(dst + nq, prev >> (64 - sh));
    }
}

// phase 2 of 2: layers 0 and 1, the direction bytes and the obstacle blocks -- byte stores.  One lane per row walks the
// chip's droplets in ascending index (a later droplet overwrites an earlier one as in dmfb.py:404-420); the window and
// visibility tests run on packed int16 pairs: with D = pos_j - pos_i, droplet j is inside row i's window iff both
// halves of D + hf lie in [0, fov), i.e. min_u16(D + hf, fov - 1) == D + hf.
template <int N>
__device__ __forceinline__ void scatter_rows(const DevCfg &c, const DevPtrs &p, const Tile &t, int tile_base, int tv,
                                             int tid, int nthreads) {
    const int fov = c.fov, hf = c.hf, ff = c.ff;
    const int rows = tv * N;
    const bool odd = (fov & 1) != 0;  // odd fov: "inside the window" and "visible" (2|d| < fov) are the same test
    const int hv = (fov - 1) / 2;     // visible <=> |d| <= hv on both axes
    const uint32_t k_hf = pk2(hf), k_f1 = pk2(fov - 1), k_hv = pk2(hv), k_2hv = pk2(2 * hv);
    // two lanes per row -- lane r of the first half of the workgroup: layer 0 and the direction bytes, lane r of the second
    // half: layer 1 -- so the serial walk over the droplets (ascending index inside each layer, as the reference's loops) is
    // half as long and every wave runs one of the two bodies, not both
    const int half = nthreads >> 1;
    const int part = tid >= half ? 1 : 0;
    for (int it = tid - part * half; it < rows; it += half) {
        const int env = it / N, i = it - env * N;
        int8_t *row = t.obs + (size_t)it * c.obs_len;
        // every read first (the byte stores below could alias them as far as the compiler knows, which would
        // serialise an LDS round trip per droplet)
        uint32_t P[N];
#pragma unroll
        for (int j = 0; j < N; ++j) P[j] = t.pos[env * N + j];
        const uint32_t pi = t.pos[it];
        const uint32_t org = pk_sub(pi, k_hf);  // window origin (ox, oy)
        if (part == 0) {
            const uint32_t d = pk_sub(t.goal[it], pi);  // goal - position, sign-extended halves
            const int8_t *zoom = (const int8_t *)(t.tab + 2 * (2 * hf + 1) * c.nq);
            const int8_t dir_x = zoom[(int)(short)(d & 0xffff) + 255], dir_y = zoom[511 + (int)(short)(d >> 16) + 255];
#pragma unroll
            for (int j = 0; j < N; ++j) {  // layer 0: droplet j inside the window
                const uint32_t w = pk_sub(P[j], org);
                if (pk_min_u(w, k_f1) == w) row[(w & 0xffff) * fov + (w >> 16)] = (int8_t)(j + 1);
            }
            row[3 * ff] = dir_x;
            row[3 * ff + 1] = dir_y;
        } else {
            uint32_t G[N];
#pragma unroll
            for (int j = 0; j < N; ++j) G[j] = t.goal[env * N + j];
#pragma unroll
            for (int j = 0; j < N; ++j) {  // layer 1: the visible droplet's goal clipped into the window
                bool vis;
                if (odd) {
                    const uint32_t w = pk_sub(P[j], org);
                    vis = pk_min_u(w, k_f1) == w;
                } else {
                    const uint32_t v = pk_add(pk_sub(P[j], pi), k_hv);
                    vis = pk_min_u(v, k_2hv) == v;
                }
                if (vis && j != i) {
                    const uint32_t g = pk_clamp_i(pk_sub(G[j], org), 0u, k_f1);
                    row[ff + (g & 0xffff) * fov + (g >> 16)] = (int8_t)(j + 1);
                }
            }
        }
    }
    if (c.nb > 0) {  // layer 2: blocks, GLOBAL coordinates used as window coordinates (reference quirk dmfb.py:422-426)
        for (int k = tid; k < rows * c.nb; k += nthreads) {
            const int r = k / c.nb, b = k - r * c.nb;
            const uint32_t o = p.blocks[(size_t)b * c.E + tile_base + r / N];
            const int x0 = o & 0xff, x1 = (o >> 8) & 0xff, y0 = (o >> 16) & 0xff, y1 = o >> 24;
            int8_t *row = t.obs + (size_t)r * c.obs_len + 2 * ff;
            for (int i = x0; i <= x1 && i < fov; ++i)
                for (int j = y0; j <= y1 && j < fov; ++j) row[i * fov + j] = 1;
        }
    }
}

// stream the finished tile to HBM: head (< 16 bytes) and tail by bytes, body 16 bytes per lane, coalesced
__device__ __forceinline__ void copy_tile_out(const Tile &t, int shift, int8_t *gobs, size_t tile_off, int bytes, int tid,
                                              int nthreads) {
    const int head = (16 - shift) & 15;
    const int hb = head < bytes ? head : bytes;
    for (int b = tid; b < hb; b += nthreads) gobs[tile_off + b] = t.obs[b];
    const int n16 = (bytes - hb) >> 4;
    const uint4 *src = (const uint4 *)(t.obs + hb);
    uint4 *dst = (uint4 *)(gobs + tile_off + hb);
    int i = tid;
    for (; i + 3 * nthreads < n16; i += 4 * nthreads) {  // four LDS reads in flight, then four stores
        const uint4 v0 = src[i], v1 = src[i + nthreads], v2 = src[i + 2 * nthreads], v3 = src[i + 3 * nthreads];
        dst[i] = v0; dst[i + nthreads] = v1; dst[i + 2 * nthreads] = v2; dst[i + 3 * nthreads] = v3;
    }
    for (; i < n16; i += nthreads) dst[i] = src[i];
    for (int b = hb + (n16 << 4) + tid; b < bytes; b += nthreads) gobs[tile_off + b] = t.obs[b];
}

template <typename T> __device__ __forceinline__ int load_action(const void *a, size_t idx) {
    return (int)((const T *)a)[idx];
}

// ---- the fused transition kernel ------------------------------------------------------------------
struct StepArgs {
    const void *actions;
    const double *uniforms;
    const uint8_t *active;
    uint32_t flags;
    dmfb_vec_step_out out;
};

// Minimum waves per SIMD the register allocator must leave room for: the transition is ONE generation of resident waves, so its
// speed follows the occupancy.  n <= 10 fits 128 VGPRs (four waves) with at most 4 spilled registers; larger n would spill dozens.
constexpr int step_min_waves(int n) { return n <= 10 ? 4 : 3; }
// OBS: the launch also builds the observations (fused small-batch launch); false = step-only launch (a.out.d_obs is NULL), which
// then carries none of the observation phases' code or registers
template <int N, bool MAPS, bool OBS>
__global__ __launch_bounds__(kBlock, step_min_waves(N)) void k_step(DevCfg c, DevPtrs p, StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int T = c.T, E = c.E;
    constexpr bool want_obs = OBS;
    const int tid = threadIdx.x;
    const int tile_base = blockIdx.x * T;
    const int shift = want_obs ? (int)(((uintptr_t)a.out.d_obs + (size_t)tile_base * N * c.obs_len) & 15) : 0;
    const Tile t = carve(smem, T, N, c.obs_len, want_obs, table_words(c.hf, c.nq), shift);
    const int tv = min(T, E - tile_base);
    const int cells = c.W * c.L;
    const int wave = tid / kWave, lane = tid % kWave;
    // Waves whose 64 slots start inside the tile run the transition, one lane per chip (T <= 64 in the
    // fused launch: wave 0 only; T = 256 in the step-only launch: all four).  The other waves zero the
    // obs tile meanwhile.
    const int step_waves = (T + kWave - 1) / kWave;

    // Terminal observations (dmfb_vec_step_out::d_obs_terminal): with the reset inside the launch the observation phases below show
    // an ended chip's NEXT episode; the observation the reference hands back for the ending step is built first, from the cells
    // the droplets ended on and the obstacle blocks of the ending episode, and only the ended chips' rows of it are written.
    // Every wave calls this at its own place in the code (the stepping wave between the transition and the reset, the others
    // behind the zero-fill): the same sequence of workgroup barriers on every path.  A tile without an ended chip pays one barrier.
    const bool term_obs = OBS && a.out.d_obs_terminal != nullptr;
    auto terminal_pass = [&](bool mine_ended) {
        if (!__syncthreads_or((int)mine_ended)) return;   // (also publishes the zeroed tile, the tables and the stepping wave's cells)
        scatter_bands<N>(c, t, smem, tv, tid, kBlock);
        __syncthreads();
        scatter_rows<N>(c, p, t, tile_base, tv, tid, kBlock);
        __syncthreads();
        const int row_bytes = N * c.obs_len;
        for (int b = tid; b < tv * row_bytes; b += kBlock)
            if (t.flag[b / row_bytes]) a.out.d_obs_terminal[(size_t)tile_base * row_bytes + b] = t.obs[b];
        __syncthreads();
        zero_tile(smem, (int)(align16((size_t)shift + (size_t)tv * N * c.obs_len) >> 4), tid, kBlock);   // for the phases below
    };
    if (wave >= step_waves) {
        if (want_obs) load_tables(c, p, t, tid - step_waves * kWave, kBlock - step_waves * kWave);
        if (want_obs) zero_tile(smem, (int)(align16((size_t)shift + (size_t)tv * N * c.obs_len) >> 4), tid - step_waves * kWave,
                                kBlock - step_waves * kWave);
        if (term_obs) terminal_pass(false);
    } else {
        const int slot = tid;
        const bool present = slot < tv;
        const int e = tile_base + (present ? slot : 0);
        const bool active = present && (!a.active || a.active[e]);
        EnvR<N> r;
        bool ended = false;
        int flush_len = 0, flush_kind = 0;
        // The record words a transition does not need stay in HBM: rng_ep (tasks generated) is read -- and written back -- only by a
        // lane whose episode ends in this launch; rng_map only where health is rebuilt from the degrade count; rng_step (transitions
        // executed = the counter word of the move draws) only by handles with maps: without maps no draw is ever consumed
        // (health is 1.0 everywhere, dmfb.py:335).  Three words in and out per chip-step less for configs A / D.
        if (present) load_env<N>(p, E, e, r, MAPS, MAPS);
        if (present && !active) {  // episode over, not reset yet: report a finished env, touch nothing
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (a.out.d_dones) a.out.d_dones[(size_t)e * N + i] = 1;
                if (a.out.d_rewards) a.out.d_rewards[(size_t)e * N + i] = 0.0;
            }
            if (a.out.d_constraints) a.out.d_constraints[e] = 0;
            if (a.out.d_success) a.out.d_success[e] = 0;
            if (a.out.d_terminated) a.out.d_terminated[e] = 1;
            if (a.out.d_team_reward) a.out.d_team_reward[e] = 0.0;
        }
        if (active) {
            // Register diet (this kernel is ONE generation of resident waves, so its speed is set by how many waves a SIMD
            // holds): actions are packed 3 bits each, per-droplet booleans are bit masks, cells are packed int16 pairs, the
            // draw and the health of a droplet are reduced to ONE bit ("it moves") as soon as both exist, and rewards are
            // produced, stored and summed one at a time.
            const size_t a0 = (size_t)e * N;
            uint64_t apk = 0;  // actions: 0 STALL, 1..4 moves (dmfb.py:26-31); anything else = 5 (no move, not a STALL)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int v = (a.flags & DMFB_ACT_I8) ? (int)((const int8_t *)a.actions)[a0 + i]
                            : (a.flags & DMFB_ACT_I64) ? (int)((const long long *)a.actions)[a0 + i] : ((const int32_t *)a.actions)[a0 + i];
                apk |= (uint64_t)((unsigned)v > 4u ? 5u : (unsigned)v) << (3 * i);
            }
            auto act_of = [&](int i) { return (int)((apk >> (3 * i)) & 7u); };
            const bool use_draws = MAPS || a.uniforms != nullptr;
            const bool compact = MAPS && p.dflags[0] != 0;
            uint32_t mvm = (1u << N) - 1u;  // bit i: droplet i's draw lets it move (u <= health, dmfb.py:335)
            if (use_draws) {
                uint64_t kpk = 0;  // degrade counts under the droplets, 4 bits each: all the gathers are in flight together
#ifndef DMFB_ABLATE_KMAP  // DMFB_ABLATE_*: timing experiments only (tools/build_variant.sh with EXTRA_FLAGS): wrong results
                if (compact) {
#pragma unroll
                    for (int i = 0; i < N; ++i) kpk |= (uint64_t)kmap_get(p, kmap_bytes(cells), e, cell_x(r.pos[i]) * c.L + cell_y(r.pos[i])) << (4 * i);
                }
#endif
                const uint32_t gen = c.b_degrade ? r.rmap - 1u : 0u;  // counter value the current maps were drawn with
                mvm = 0;
                uint32_t w[4] = {0u, 0u, 0u, 0u};  // one Philox4x32-10 serves TWO droplets' 53-bit draws: (w0 w1) and (w2 w3)
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const int cell = cell_x(r.pos[i]) * c.L + cell_y(r.pos[i]);
                    double prob = 1.0;  // getMoveProb (dmfb.py:361-363): the float64 map, or health rebuilt from the degrade count
                    if (MAPS && !compact) prob = p.health[(size_t)e * cells + cell];
                    if (compact) {  // health = 1.0 * degrade * ... * degrade (count times), degrade from the map's Philox stream
                        const int k = (int)((kpk >> (4 * i)) & 15u);
                        if (k == kCountMax) prob = p.health[(size_t)e * cells + cell];  // saturated count: the map itself
                        else if (k != 0) prob = health_from_count(degrade_of(c, c.env_id0 + (uint32_t)e, gen, cell), k);
                    }
                    double draw;
                    if (a.uniforms) draw = a.uniforms[a0 + i];
                    else {
#ifdef DMFB_ABLATE_PHILOX
                        draw = (double)((r.rstep * 2654435761u + (uint32_t)i * 40503u) >> 8) * (1.0 / 16777216.0);
#else
                        if ((i & 1) == 0) philox(c.k0, c.k1, c.env_id0 + (uint32_t)e, r.rstep, (uint32_t)(i >> 1), STREAM_MOVE << 8, w);
                        draw = (i & 1) ? u53(w[2], w[3]) : u53(w[0], w[1]);
#endif
                    }
                    mvm |= (uint32_t)(draw <= prob) << i;
                }
            }
            const uint32_t k_hi = (uint32_t)(c.W - 1) | ((uint32_t)(c.L - 1) << 16);
            // Droplet.move (dmfb.py:103-124) of a packed cell: +-1 on one axis, clamped to the chip
            auto moved = [&](uint32_t cell, int act) {
                const uint32_t delta = act == 1 ? 0x00000001u : act == 2 ? 0x0000ffffu : act == 3 ? 0xffff0000u : act == 4 ? 0x00010000u : 0u;
                return pk_clamp_i(pk_add(cell, delta), 0u, k_hi);
            };
            uint32_t (&P)[N] = r.pos;
            const uint32_t (&Gp)[N] = r.goal;
            // _isTouchingBlocks (dmfb.py:301-308) of every droplet's TENTATIVE cell: it depends only on the
            // droplet's own position and action, so it is evaluated up front, one pass over the blocks.
            uint32_t blocked = 0;
            if (c.nb > 0) {
                for (int b = 0; b < c.nb; ++b) {
                    const uint32_t o = p.blocks[(size_t)b * E + e];
                    const int x0 = o & 0xff, x1 = (o >> 8) & 0xff, y0 = (o >> 16) & 0xff, y1 = o >> 24;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const uint32_t t = moved(P[i], act_of(i));
                        const int tx = (int)(t & 0xffffu), ty = (int)(t >> 16);
                        blocked |= (uint32_t)((tx >= x0) & (tx <= x1) & (ty >= y0) & (ty <= y1)) << i;
                    }
                }
            }
            // ---- moveDroplets (dmfb.py:253-299).  Packed int16 pairs (x | y << 16): one packed subtract, one packed unsigned
            // min and ONE compare per pair test (v_pk_sub_i16 / v_pk_min_u16) instead of two of each plus a mask AND.
            const uint32_t k_one = 0x00010001u, k_two = 0x00020002u;
            auto pdist = [](uint32_t u, uint32_t v) {  // Manhattan distance of two packed cells
                const uint32_t ad = pk_abs(pk_sub(u, v));
                return (int)((ad & 0xffffu) + (ad >> 16));
            };
            // |ax - bx| <= 1 and |ay - by| <= 1, given a1 = a + (1, 1): both halves of a1 - b lie in {0, 1, 2}
            auto near1p = [&](uint32_t a1, uint32_t bcell) {
                const uint32_t t = pk_sub(a1, bcell);
                return pk_min_u(t, k_two) == t;
            };
            uint32_t past[N];
            uint32_t cpk = 0, wdm = 0;  // reward code of droplet i in bits 2i..2i+1; bit i: it was on its goal before the moves
            r.step += 1;
            const bool dup = (r.flags & FLAG_DUP) != 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {  // moveOneDroplet (dmfb.py:325-359), strictly in index order
                past[i] = P[i];
                const int act = act_of(i);
                const int old = pdist(P[i], Gp[i]);
                wdm |= (uint32_t)(old == 0) << i;
                int code = 0;
                if (!(c.stall && old == 0)) {
                    if ((mvm >> i) & 1u) {
                        uint32_t np = moved(P[i], act);
                        if ((blocked >> i) & 1u) np = P[i];  // revert when touching a block (dmfb.py:338-340)
                        bool clash = false;  // _isinvalidaction (dmfb.py:310-323)
                        if (dup) {
                            P[i] = np;
#pragma unroll
                            for (int u = 0; u < N; ++u)
#pragma unroll
                                for (int v = u + 1; v < N; ++v) clash |= P[u] == P[v];
                        } else {
#pragma unroll
                            for (int j = 0; j < N; ++j)
                                if (j != i) clash |= P[j] == np;
                        }
                        P[i] = clash ? past[i] : np;
                    }
                    const int nd = pdist(P[i], Gp[i]);
                    code = (nd == old && old == 0) ? 1 : (nd == old && act == 0) ? 2 : (nd < old) ? 1 : 3;
                }
                cpk |= (uint32_t)code << (2 * i);
            }
            r.rstep += 1;
            int sta[N], dyn[N];
#pragma unroll
            for (int i = 0; i < N; ++i) { sta[i] = 0; dyn[i] = 0; }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const uint32_t c1 = pk_add(P[i], k_one), p1 = pk_add(past[i], k_one);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    if (j > i) {  // comflic_static (dmfb.py:254-261)
                        const bool near = near1p(c1, P[j]);
                        sta[i] += near; sta[j] += near;
                    }
                    if (j != i) {  // comflic_dynamic (dmfb.py:263-271)
                        const bool near = near1p(p1, P[j]);
                        dyn[i] += near; dyn[j] += near;
                    }
                }
            }
            int constraints = 0;
            bool all_done = true;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                constraints += sta[i] + dyn[i];
                all_done &= P[i] == Gp[i];
            }
            bool log_full = false;
            if (MAPS && (a.flags & DMFB_STEP_RECORD)) {  // addUsage (dmfb.py:459-463): append this step to the chip's usage log
                uint16_t *ul = p.ulog + ((size_t)e * c.ucap + r.ulen) * c.lstride;
#ifndef DMFB_ABLATE_LOG
                auto entry = [&](int i) { return P[i] != Gp[i] ? (uint32_t)(cell_x(P[i]) * c.L + cell_y(P[i])) : 0xffffu; };
                if (c.lstride == 16) {  // one whole, aligned 32-byte sector per step: no partial-sector write reaches HBM
                    uint32_t w[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        uint32_t lo = 0xffffu, hi = 0xffffu;
                        if (2 * k < N) lo = entry(2 * k);
                        if (2 * k + 1 < N) hi = entry(2 * k + 1);
                        w[k] = lo | (hi << 16);
                    }
                    ((uint4 *)ul)[0] = make_uint4(w[0], w[1], w[2], w[3]);
                    ((uint4 *)ul)[1] = make_uint4(w[4], w[5], w[6], w[7]);
                } else {
#pragma unroll
                    for (int i = 0; i < N; ++i) ul[i] = (uint16_t)entry(i);
                }
#endif
                r.ulen += 1;
                log_full = (int)r.ulen == c.ucap;  // folded into the map right after this step: a step always finds room
            }
            r.cum += (uint32_t)constraints;
            const bool in_time = (int)r.step < c.max_step;  // DMFBenv.step (dmfb.py:577-585)
            const bool success = in_time && all_done && r.cum == 0;
            const bool term = in_time ? all_done : true;
            // rewards: ((base - 2*sta) - 2*dy), 0 if it was done, +10, +10  (dmfb.py:288-296); produced one at a time
            auto reward = [&](int i) {
                const int code = (int)((cpk >> (2 * i)) & 3u);
                const double base = code == 0 ? 0.0 : code == 1 ? -0.1 : code == 2 ? -0.25 : -0.4;
                double v = (base - (double)(2 * sta[i])) - (double)(2 * dyn[i]);
                if (c.stall && ((wdm >> i) & 1u)) v = 0.0;
                if (all_done) {
                    v = v + 10.0;
                    if (constraints == 0) v = v + 10.0;
                }
                if (a.out.d_rewards) a.out.d_rewards[(size_t)e * N + i] = v;
                if (a.out.d_dones) a.out.d_dones[(size_t)e * N + i] = (uint8_t)(in_time ? (P[i] == Gp[i]) : true);
                return v;
            };
            double s;  // np.sum(list)/n in NumPy's order (rollout.py:33): sequential below 8 values, else 8 partial sums
            if constexpr (N < 8) {
                s = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) s = s + reward(i);
            } else {
                constexpr int M = N - (N % 8);
                // q[j] = rew[j] + rew[j + 8] + ... (j < 8), then ((q0+q1)+(q2+q3))+((q4+q5)+(q6+q7)), then the rest in index order
                auto q = [&](int j) {
                    double v = reward(j);
#pragma unroll
                    for (int i = 8; i < M; i += 8) v = v + reward(i + j);
                    return v;
                };
                const double q01 = q(0) + q(1), q23 = q(2) + q(3);
                const double lo4 = q01 + q23;
                const double q45 = q(4) + q(5), q67 = q(6) + q(7);
                s = lo4 + (q45 + q67);
#pragma unroll
                for (int i = M; i < N; ++i) s = s + reward(i);
            }
            if (a.out.d_constraints) a.out.d_constraints[e] = constraints;
            if (a.out.d_success) a.out.d_success[e] = (uint8_t)success;
            if (a.out.d_terminated) a.out.d_terminated[e] = (uint8_t)term;
            if (a.out.d_team_reward) a.out.d_team_reward[e] = s / (double)N;
            if (dup) r.flags = any_dup<N>(r) ? (r.flags | FLAG_DUP) : (r.flags & ~FLAG_DUP);
            ended = term && (a.flags & DMFB_STEP_AUTORESET);
            if (MAPS && (ended || log_full)) {  // the waves fold the log in after the barrier (flush_usage)
                flush_len = (int)r.ulen;
                flush_kind = ended ? 1 : 2;  // 1: + updateHealth (reset(new=False), dmfb.py:182-183)   2: log full
                r.ulen = 0;
            }
        }
        if (term_obs) {  // the cells the step ended on, flag = "this chip's episode ended"; both are rewritten after the reset
            if (present) {
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    t.pos[slot * N + i] = r.pos[i];
                    t.goal[slot * N + i] = r.goal[i];
                }
                t.flag[slot] = (uint8_t)ended;
            }
            terminal_pass(ended);
        }
        // ---- episode boundary inside the launch: reset(new=False) for the lanes that ended
        if (ended) r.rep = p.st[(size_t)Rec<N>::W_REP * E + e];
        unsigned long long m = __ballot(ended);
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const uint32_t gid = c.env_id0 + (uint32_t)(tile_base + wave * kWave + src);
            const uint32_t ep = (uint32_t)__shfl((int)r.rep, src, kWave);
            uint32_t pts[N];
            gen_task_wave<N>(c, gid, ep, pts);
            if (c.nb > 0) {
                const uint32_t blk = gen_blocks_wave<N>(c, gid, ep, pts);
                if (lane < c.nb) p.blocks[(size_t)lane * E + (tile_base + wave * kWave + src)] = blk;
            }
            if (lane == src) {
                task_to_env<N>(pts, r);
                store_starts<N>(p, E, e, r.pos);
                r.rep += 1; r.step = 0; r.cum = 0; r.flags = 0;
            }
        }
        if (active) store_env<N>(p, E, e, r, ended, MAPS);

        if (present) {
            if (want_obs) {
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    t.pos[slot * N + i] = r.pos[i];
                    t.goal[slot * N + i] = r.goal[i];
                }
            }
            t.flag[slot] = (uint8_t)flush_kind;
            if (MAPS) t.ulen[slot] = (uint16_t)flush_len;
        }
    }
    const bool may_flush = MAPS && (a.flags & (DMFB_STEP_AUTORESET | DMFB_STEP_RECORD));
    if (!want_obs && !may_flush) return;
    __syncthreads();
    if (may_flush) {  // usage logs of the chips that were reset (+ updateHealth) or whose log is full
        // one flag read per thread + a wave ballot instead of a serial scan of the tile's flags; every wave folds the
        // chips of its own 64 slots through its own LDS histogram
        uint16_t *hist = c.hist_bytes ? (uint16_t *)(smem + tile_lds_bytes(T, N, c.obs_len, want_obs, table_words(c.hf, c.nq)) +
                                                     (size_t)wave * c.hist_bytes) : nullptr;
        for (int base = 0; base < tv; base += kBlock) {
            const int s = base + tid;
            unsigned long long fm = __ballot(s < tv && t.flag[s] != 0);
            while (fm) {
                const int src = __ffsll((long long)fm) - 1;
                fm &= fm - 1;
                const int slot = base + wave * kWave + src;
                flush_usage(c, p, tile_base + slot, t.ulen[slot], t.flag[slot] == 1, hist, lane);
            }
        }
    }
    if (!want_obs) return;
    scatter_bands<N>(c, t, smem, tv, tid, kBlock);
    __syncthreads();
    scatter_rows<N>(c, p, t, tile_base, tv, tid, kBlock);
    __syncthreads();
    copy_tile_out(t, shift, a.out.d_obs, (size_t)tile_base * N * c.obs_len, tv * N * c.obs_len, tid, kBlock);
}

// ---- standalone observation kernel (getObs after reset/restart/set_task, and the second launch of the
// step-only + observe pair used for large batches) ---------------------------------------------------
template <int N>
__global__ __launch_bounds__(kObsBlock) void k_observe(DevCfg c, DevPtrs p, const uint8_t *mask, int8_t *gobs) {
    // Persistent workgroups: the grid covers the CUs a few times over and every workgroup walks tiles blockIdx.x,
    // blockIdx.x + gridDim.x, ...  The observation tables are copied into LDS once per workgroup.
    // Roles: the LAST wave is the loader -- it alone reads global memory (packed records, mask; one chip per lane,
    // T <= 64) and never stores; the other waves stream the finished tiles out and never load.  gfx950 counts a wave's
    // loads and stores in ONE in-order counter, so a wave that did both would have to wait for the acknowledgement of
    // its previous tile's stores before it could use the next records.  With the roles split no wave ever waits for a
    // store: a tile's LDS is reused as soon as it has been READ.
    // The records run two tiles ahead: while tile k is scattered the records of tile k+1 sit in the loader's registers;
    // before tile k is streamed out they are unpacked into the second position buffer and tile k+2 is requested.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int T = c.T_obs, E = c.E;
    const int tid = threadIdx.x;
    constexpr int kWork = kObsBlock - kWave;  // threads that stream tiles out
    const bool loader = tid >= kWork;
    const int ltid = tid - kWork;
    const int ntiles = (E + T - 1) / T;
    const int row_bytes = N * c.obs_len;
    constexpr int NP = Rec<N>::NP;
    uint32_t pw[NP], gw[NP];
    auto prefetch = [&](int tile) {  // lane = chip: for every record word the lanes read consecutive addresses
        const int base = tile * T;
        const bool on = tile < ntiles && ltid < min(T, E - base);
#pragma unroll
        for (int w = 0; w < NP; ++w) {
#ifdef DMFB_ABLATE_LOADS  // timing experiment: no global reads while the tiles stream out (wrong observations)
            pw[w] = on ? (uint32_t)(base + ltid) * 0x01010101u & 0x07070707u : 0u;
            gw[w] = pw[w] ^ 0x01000100u;
#else
            pw[w] = on ? p.st[(size_t)w * E + base + ltid] : 0u;
            gw[w] = on ? p.st[(size_t)(NP + w) * E + base + ltid] : 0u;
#endif
        }
    };
    const Tile t0 = carve(smem, T, N, c.obs_len, true, table_words(c.hf, c.nq), 0);
    // second position/goal buffer behind the tile structure (tile_lds_bytes reserves it)
    uint32_t *const pos2 = (uint32_t *)(t0.flag + align16((size_t)T));
    auto unpack = [&](int tile, uint32_t *pos, uint32_t *goal) {
        if (tile >= ntiles || ltid >= min(T, E - tile * T)) return;
#pragma unroll
        for (int w = 0; w < NP; ++w) {
            pos[ltid * N + 2 * w] = to_xy(pw[w]);
            goal[ltid * N + 2 * w] = to_xy(gw[w]);
            if (2 * w + 1 < N) {
                pos[ltid * N + 2 * w + 1] = to_xy(pw[w] >> 16);
                goal[ltid * N + 2 * w + 1] = to_xy(gw[w] >> 16);
            }
        }
    };
    int tile = blockIdx.x;
    int buf = 0;
    DMFB_STAMP_INIT();
    if (loader) {
        prefetch(tile);
        load_tables(c, p, t0, ltid, kWave);
        unpack(tile, t0.pos, t0.goal);
        prefetch(tile + gridDim.x);
    }
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const int tile_base = tile * T;
        const int shift = (int)(((uintptr_t)gobs + (size_t)tile_base * row_bytes) & 15);
        Tile t = carve(smem, T, N, c.obs_len, true, table_words(c.hf, c.nq), shift);
        uint32_t *const npos = buf ? t0.pos : pos2, *const ngoal = npos + (size_t)T * N;
        if (buf) { t.pos = pos2; t.goal = pos2 + (size_t)T * N; }
        const int tv = min(T, E - tile_base);
        // this barrier publishes the positions unpacked during the previous tile and separates its stream-out (LDS
        // reads) from the zero fill below; the loader's lanes flag and count the chips to refresh
        // (flags double-buffered like the positions: the loader reaches the next tile while the other waves may still
        // read this tile's flags in the partial stream-out; the second buffer is the carve's usage-log slot, unused here)
        uint8_t *const flag = buf ? (uint8_t *)t0.ulen : t0.flag;
        bool refresh = false;
        if (loader && ltid < tv) {
            refresh = !mask || mask[tile_base + ltid] != 0;
            flag[ltid] = (uint8_t)refresh;
        }
        const int cnt = __syncthreads_count(refresh);
        DMFB_STAMP(0);
        if (cnt != 0) {  // (uniform) something to refresh in this tile
            zero_tile(smem, (int)(align16((size_t)shift + (size_t)tv * row_bytes) >> 4), tid, kObsBlock);
            DMFB_STAMP(1);
            __syncthreads();
            DMFB_STAMP(2);
#ifndef DMFB_ABLATE_BANDS  // timing experiments only (tools/build_variant.sh with EXTRA_FLAGS): wrong observations
            scatter_bands<N>(c, t, smem, tv, tid, kObsBlock);
            __syncthreads();
#endif
            DMFB_STAMP(3);
#ifndef DMFB_ABLATE_ROWS
            scatter_rows<N>(c, p, t, tile_base, tv, tid, kObsBlock);
            __syncthreads();
#endif
            DMFB_STAMP(4);
        }
        if (loader) {
            unpack(tile + gridDim.x, npos, ngoal);
            prefetch(tile + 2 * gridDim.x);
        } else if (cnt == tv) {
            copy_tile_out(t, shift, gobs, (size_t)tile_base * row_bytes, tv * row_bytes, tid, kWork);
        } else if (cnt != 0) {
            for (int s = 0; s < tv; ++s)
                if (flag[s])
                    for (int b = tid; b < row_bytes; b += kWork)
                        gobs[(size_t)(tile_base + s) * row_bytes + b] = t.obs[(size_t)s * row_bytes + b];
        }
        DMFB_STAMP(5);
    }
#ifdef DMFB_STAMPS
    __builtin_amdgcn_s_waitcnt(0);  // stores acknowledged
    DMFB_STAMP(6);
#endif
}

// ---- reset / restart / init: one wave per env -------------------------------------------------------
// mode 0: reset(new=False)  1: reset(new=True)  2: restart()  3: create (first task + fresh maps)
template <int N>
__global__ __launch_bounds__(kBlock) void k_reset(DevCfg c, DevPtrs p, const uint8_t *mask, int mode) {
    using R = Rec<N>;
    const int E = c.E;
    const int e = blockIdx.x * (kBlock / kWave) + (int)(threadIdx.x / kWave);
    if (e >= E) return;
    if (mask && !mask[e]) return;
    const int lane = lane_id();
    const int cells = c.W * c.L;
    if (mode == 2) {  // restartforall (dmfb.py:185-190) + counters (dmfb.py:599-605)
        if (lane == 0) {
            bool d = false;
            uint32_t sw[R::NP];
#pragma unroll
            for (int w = 0; w < R::NP; ++w) {
                sw[w] = p.starts[(size_t)w * E + e];
                p.st[(size_t)(R::W_POS + w) * E + e] = sw[w];
            }
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = i + 1; j < N; ++j)
                    d |= ((sw[i >> 1] >> (16 * (i & 1))) & 0xffff) == ((sw[j >> 1] >> (16 * (j & 1))) & 0xffff);
            // counters to zero, the usage log keeps its length (restart does not touch the maps)
            p.st[(size_t)R::W_STEP * E + e] = (p.st[(size_t)R::W_STEP * E + e] & (0xfffu << kUlenShift)) | (d ? (FLAG_DUP << 16) : 0u);
            p.st[(size_t)R::W_CUM * E + e] = 0;
        }
        return;
    }
    // steps waiting in the usage log (read by every lane before lane 0 rewrites the record)
    const int ulen = (mode == 3 || !p.health) ? 0 : (int)(p.st[(size_t)R::W_STEP * E + e] >> kUlenShift);
    uint32_t rep = mode == 3 ? 0u : p.st[(size_t)R::W_REP * E + e];
    uint32_t rmap = mode == 3 ? 0u : p.st[(size_t)R::W_RMAP * E + e];
    uint32_t pts[N];
    gen_task_wave<N>(c, c.env_id0 + (uint32_t)e, rep, pts);
    if (c.nb > 0) {
        const uint32_t blk = gen_blocks_wave<N>(c, c.env_id0 + (uint32_t)e, rep, pts);
        if (lane < c.nb) p.blocks[(size_t)lane * E + e] = blk;
    }
    if (lane == 0) {
        EnvR<N> r;
        task_to_env<N>(pts, r);
        store_starts<N>(p, E, e, r.pos);
        r.step = 0; r.flags = 0; r.cum = 0; r.ulen = 0;
        r.rstep = mode == 3 ? 0u : p.st[(size_t)R::W_RSTEP * E + e];
        r.rep = rep + 1;
        r.rmap = rmap + ((mode == 1 || mode == 3) && p.health && c.b_degrade ? 1u : 0u);
        store_env<N>(p, E, e, r, true);
        p.st[(size_t)R::W_RMAP * E + e] = r.rmap;   // (store_env leaves rng_map alone: a transition never changes it)
    }
    if (p.health) {
        if (mode == 0) {  // refresh(new=False): the episode's usage, then updateHealth (dmfb.py:182-183)
            extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
            uint16_t *hist = c.hist_bytes ? (uint16_t *)(smem + (size_t)(threadIdx.x / kWave) * c.hist_bytes) : nullptr;
            flush_usage(c, p, e, ulen, true, hist, lane);
        } else {  // new maps: whatever the log holds is discarded with the old usage map
            const size_t base = (size_t)e * cells;
            for (int cidx = lane; cidx < cells; cidx += kWave) { p.health[base + cidx] = 1.0; p.usage[base + cidx] = 0; }
            const size_t kb = kmap_bytes(cells);
            for (int w = lane; w < (int)(kb / 4); w += kWave) ((uint32_t *)p.kmap)[((size_t)e * kb) / 4 + w] = 0u;
            gen_degrade_env(c, p, cells, e, rmap, lane, kWave);
        }
    }
}


// ---- per-N launchers: declared here, defined (explicitly specialised) in dmfb_vec_n.hip -------------
template <int N>
hipError_t launch_step_n(const DevCfg &c, const DevPtrs &p, const StepArgs &a, int grid, size_t lds, hipStream_t s);
// the lane-per-droplet transition (dmfb_step_lanes.h), instantiated for n >= kLanesMinN only
constexpr int kLanesMinN = 8;
template <int N>
hipError_t launch_step_lanes_n(const DevCfg &c, const DevPtrs &p, const StepArgs &a, int grid, size_t lds, hipStream_t s);
template <int N>
hipError_t launch_reset_n(const DevCfg &c, const DevPtrs &p, const uint8_t *mask, int mode, int grid, size_t lds, hipStream_t s);
template <int N>
hipError_t launch_observe_n(const DevCfg &c, const DevPtrs &p, const uint8_t *mask, int8_t *obs, int grid, size_t lds,
                            hipStream_t s, hipEvent_t t0, hipEvent_t t1);  // t0/t1 non-null: dispatch start/stop time stamps

}  // namespace dmfbk
