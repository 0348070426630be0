// dmfb_step_lanes.h -- the DMFB transition with ONE LANE PER DROPLET (16 lanes per chip, four chips per wave) for n >= 8.
//
// Same semantics as dmfbk::k_step (dmfb_kernels.h), which keeps one lane per chip: DMFBenv.step + moveDroplets + addUsage
// (env/DMFB/dmfb.py:253-363, 459-463, 560-587).  With ten droplets the lane-per-chip mapping is a serial chain of ten
// Philox4x32-10 draws, ten health rebuilds, an O(n^2) clash test per move and O(n^2) conflict counts in every lane, and a
// 262 144-chip launch is ONE generation of 4096 resident waves -- latency-bound.  Here every droplet
// has its own lane:
//   * the draw, the health under the droplet and the obstacle test run once per lane, in parallel;
//   * the index-ordered move loop (a droplet sees j < i already moved and j > i not yet moved, dmfb.py:279-283, 336-343)
//     is n rounds: the mover's tentative cell is broadcast to its 16-lane row (DPP row_newbcast), every other lane compares
//     it with its own cell, a wave ballot collects the clash bits and the mover reads its row's 16 bits;
//   * static / dynamic conflict counts (dmfb.py:254-271) are n broadcast rounds with O(1) work per lane;
//   * constraints, "all done" and the team reward are row reductions; the team reward is formed by lane 0 in NumPy's
//     pairwise order (rollout.py:33), rewards stay float64 with the reference's operation order.
// A wave owns four chips, so the same batch is 16x as many (short) waves: the latency chains overlap.
// Step-only launches only (the observation follows as k_observe); the fused small-batch launch keeps k_step.
#pragma once

#include "dmfb_kernels.h"

namespace dmfbk {

constexpr int kLaneGroup = 16;                      // lanes per chip = one DPP row
constexpr int kLaneChips = kBlock / kLaneGroup;     // chips per 256-thread workgroup

template <int CTRL> __device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
// value of lane `SRC` of this lane's 16-lane row, in every lane of the row
template <int SRC> __device__ __forceinline__ int row_bcast(int v) { return dpp_mov<0x150 + SRC>(v); }
template <int SRC> __device__ __forceinline__ double row_bcast_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)row_bcast<SRC>((int)(unsigned)b), hi = (unsigned)row_bcast<SRC>((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}
// the value K lanes up the row (row_shl:K; lanes without a source read 0)
template <int K> __device__ __forceinline__ double row_up_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)dpp_mov<0x100 + K>((int)(unsigned)b), hi = (unsigned)dpp_mov<0x100 + K>((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}
// the row's 16 bits of a wave ballot
__device__ __forceinline__ unsigned row_bits(unsigned long long m, int lane) { return (unsigned)(m >> (lane & 48)) & 0xffffu; }
// sum over the 16 lanes of a row, in every lane (rotations within the row)
__device__ __forceinline__ int row_sum(int v) {
    v += dpp_mov<0x120 + 8>(v);
    v += dpp_mov<0x120 + 4>(v);
    v += dpp_mov<0x120 + 2>(v);
    v += dpp_mov<0x120 + 1>(v);
    return v;
}
__device__ __forceinline__ bool near1(int ax, int ay, int bx, int by) { return (iabs(ax - bx) <= 1) & (iabs(ay - by) <= 1); }

// do any two droplets of the row share a cell?  `cell` = x | y << 8 | 1 << 16 for lanes that hold a droplet, 0 otherwise
// (only reachable through set_task: FLAG_DUP); rotations by 1..15 meet every pair
template <int K> struct DupRounds {
    static __device__ __forceinline__ bool run(int cell) {
        const int o = dpp_mov<0x120 + K>(cell);
        return ((cell == o) & ((cell >> 16) != 0)) | DupRounds<K - 1>::run(cell);
    }
};
template <> struct DupRounds<0> { static __device__ __forceinline__ bool run(int) { return false; } };

template <int N, bool MAPS> struct LaneStep {
    // per-lane droplet state; the chip's scalars are replicated in the 16 lanes of its row
    int x, y, gx, gy;
    uint32_t step, flags, ulen, cum, rstep, rep, rmap;

    // ---- the index-ordered move loop, round R (moveOneDroplet, dmfb.py:325-359) ----
    template <int R>
    __device__ __forceinline__ void move_round(const DevCfg &c, int i, int lane, bool has, bool moves, int nx, int ny, bool any_dup_row, bool dup) {
        // the mover's tentative cell, valid only if it does move this step
        const int mine = (i == R && moves) ? (nx | (ny << 8) | (1 << 16)) : 0;
        const int t = row_bcast<R>(mine);
        const bool tv = (t >> 16) != 0;
        const int tx = t & 0xff, ty = (t >> 8) & 0xff;
        bool hit = has & (i != R) & tv & (x == tx) & (y == ty);   // _isinvalidaction (dmfb.py:310-323)
        if (any_dup_row) {  // (wave-uniform) some chip of this wave carries FLAG_DUP: there the test is "any two droplets coincide"
            const int cell = has ? (((i == R && tv) ? (tx | (ty << 8)) : (x | (y << 8))) | (1 << 16)) : 0;
            const bool d = DupRounds<15>::run(cell);
            hit = dup ? d : hit;
        }
        const unsigned clash = row_bits(__ballot(hit), lane);
        if (i == R && tv && clash == 0) { x = tx; y = ty; }
        if constexpr (R + 1 < N) move_round<R + 1>(c, i, lane, has, moves, nx, ny, any_dup_row, dup);
    }
    // ---- comflic_static / comflic_dynamic (dmfb.py:254-271), round R: lane R's current and past cell meet every other lane ----
    template <int R>
    __device__ __forceinline__ void conflict_round(int i, bool has, int px, int py, int &sta, int &dyn) {
        const int t = row_bcast<R>(x | (y << 8) | (px << 16) | (py << 24));
        const int cx = t & 0xff, cy = (t >> 8) & 0xff, qx = (t >> 16) & 0xff, qy = (t >> 24) & 0xff;
        if (has && i != R) {
            sta += near1(x, y, cx, cy);                          // unordered pair {i, R}: counted once from each side
            dyn += near1(px, py, cx, cy) + near1(qx, qy, x, y);  // ordered pairs (i, R) and (R, i): each bumps both ends
        }
        if constexpr (R + 1 < N) conflict_round<R + 1>(i, has, px, py, sta, dyn);
    }
    // sum of the rewards of lanes FIRST.. in index order on top of `s` (the sequential part of np.sum)
    template <int K>
    static __device__ __forceinline__ double add_tail(double s, double rew) {
        if constexpr (K < N) return add_tail<K + 1>(s + row_bcast_f64<K>(rew), rew);
        else return s;
    }
};

template <int N, bool MAPS>
__global__ __launch_bounds__(kBlock) void k_step_lanes(DevCfg c, DevPtrs p, StepArgs a) {
    static_assert(N >= 2 && N <= kLaneGroup, "one DPP row per chip");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using R = Rec<N>;
    const int E = c.E;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int i = lane & (kLaneGroup - 1);                     // droplet index (lanes i >= N idle along)
    const int e_raw = blockIdx.x * kLaneChips + tid / kLaneGroup;
    const bool present = e_raw < E;
    const int e = present ? e_raw : 0;
    const bool active = present && (!a.active || a.active[e]);
    const int cells = c.W * c.L;
    LaneStep<N, MAPS> r;
    {
        const int w = i < N ? (i >> 1) : 0;
        const uint32_t pw = p.st[(size_t)(R::W_POS + w) * E + e] >> (16 * (i & 1));
        const uint32_t gw = p.st[(size_t)(R::W_GOAL + w) * E + e] >> (16 * (i & 1));
        r.x = pw & 0xff; r.y = (pw >> 8) & 0xff; r.gx = gw & 0xff; r.gy = (gw >> 8) & 0xff;
        const uint32_t s = p.st[(size_t)R::W_STEP * E + e];
        r.step = s & 0xffff; r.flags = (s >> 16) & kFlagMask; r.ulen = s >> kUlenShift;
        r.cum = p.st[(size_t)R::W_CUM * E + e];
        r.rstep = p.st[(size_t)R::W_RSTEP * E + e];
        r.rep = p.st[(size_t)R::W_REP * E + e];
        r.rmap = p.st[(size_t)R::W_RMAP * E + e];
    }
    if (present && !active) {  // episode over, not reset yet: report a finished env, touch nothing
        if (i < N) {
            if (a.out.d_dones) a.out.d_dones[(size_t)e * N + i] = 1;
            if (a.out.d_rewards) a.out.d_rewards[(size_t)e * N + i] = 0.0;
        }
        if (i == 0) {
            if (a.out.d_constraints) a.out.d_constraints[e] = 0;
            if (a.out.d_success) a.out.d_success[e] = 0;
            if (a.out.d_terminated) a.out.d_terminated[e] = 1;
            if (a.out.d_team_reward) a.out.d_team_reward[e] = 0.0;
        }
    }
    bool ended = false;
    int flush_len = 0, flush_kind = 0;
    // everything below is executed by whole waves (ballots, row broadcasts): inactive rows carry `act = false` through it
    const bool act = active;
    const bool actl = act && i < N;
    int action = 0;
    if (actl) {
        const size_t ai = (size_t)e * N + i;
        action = (a.flags & DMFB_ACT_I8) ? (int)((const int8_t *)a.actions)[ai]
               : (a.flags & DMFB_ACT_I64) ? (int)((const long long *)a.actions)[ai] : ((const int32_t *)a.actions)[ai];
    }
    const bool use_draws = MAPS || a.uniforms != nullptr;
    double prob = 1.0, draw = 0.0;
    if (use_draws && actl) {
        const int cell = r.x * c.L + r.y;
        const bool compact = MAPS && p.dflags[0] != 0;
        // getMoveProb (dmfb.py:361-363): the float64 map, or health rebuilt from one nibble of the degrade-count map
        if (MAPS && !compact) prob = p.health[(size_t)e * cells + cell];
        if (a.uniforms) draw = a.uniforms[(size_t)e * N + i];
        else {
            uint32_t w[4];
            philox(c.k0, c.k1, c.env_id0 + (uint32_t)e, r.rstep, (uint32_t)(i >> 1), STREAM_MOVE << 8, w);   // droplets 2p, 2p + 1 share a block
            draw = (i & 1) ? u53(w[2], w[3]) : u53(w[0], w[1]);
        }
        if (compact) {
            const int k = kmap_get(p, kmap_bytes(cells), e, cell);
            const uint32_t gen = c.b_degrade ? r.rmap - 1u : 0u;  // counter value the current maps were drawn with
            if (k == kCountMax) prob = p.health[(size_t)e * cells + cell];  // saturated count: the map itself
            else if (k != 0) prob = health_from_count(degrade_of(c, c.env_id0 + (uint32_t)e, gen, cell), k);
        }
    }
    // Droplet.move (dmfb.py:103-124) of this lane's droplet: tentative cell, clamped; _isTouchingBlocks (dmfb.py:301-308) reverts it
    int nx = r.x + (action == 1) - (action == 2), ny = r.y + (action == 4) - (action == 3);
    nx = nx > c.W - 1 ? c.W - 1 : (nx < 0 ? 0 : nx);
    ny = ny > c.L - 1 ? c.L - 1 : (ny < 0 ? 0 : ny);
    if (c.nb > 0 && actl) {
        bool blocked = false;
        for (int b = 0; b < c.nb; ++b) {
            const uint32_t o = p.blocks[(size_t)b * E + e];
            const int x0 = o & 0xff, x1 = (o >> 8) & 0xff, y0 = (o >> 16) & 0xff, y1 = o >> 24;
            blocked |= (nx >= x0) & (nx <= x1) & (ny >= y0) & (ny <= y1);
        }
        if (blocked) { nx = r.x; ny = r.y; }
    }
    // ---- moveDroplets (dmfb.py:253-299)
    const int px = r.x, py = r.y;                                  // past cell
    const int old = iabs(r.x - r.gx) + iabs(r.y - r.gy);
    const bool was_done = old == 0;
    const bool stalled = c.stall && old == 0;                      // finished droplets neither draw nor move (dmfb.py:331-332)
    const bool moves = actl && !stalled && (use_draws ? (draw <= prob) : true);
    const bool dup = act && (r.flags & FLAG_DUP) != 0;
    const bool any_dup_row = __ballot(dup) != 0ull;
    if (act) r.step += 1;
    r.template move_round<0>(c, i, lane, actl, moves, nx, ny, any_dup_row, dup);
    const int nd = iabs(r.x - r.gx) + iabs(r.y - r.gy);
    const int code = stalled ? 0 : (nd == old && old == 0) ? 1 : (nd == old && action == 0) ? 2 : (nd < old) ? 1 : 3;
    if (act) r.rstep += 1;
    int sta = 0, dyn = 0;
    r.template conflict_round<0>(i, actl, px, py, sta, dyn);
    const int constraints = row_sum(actl ? sta + dyn : 0);
    const unsigned full = (1u << N) - 1u;
    const unsigned done_bits = row_bits(__ballot(actl && nd == 0), lane);
    const bool all_done = done_bits == full;
    // rewards: ((base - 2*sta) - 2*dy), 0 if it was done, +10, +10  (dmfb.py:288-296)
    double rew;
    {
        const double base = code == 0 ? 0.0 : code == 1 ? -0.1 : code == 2 ? -0.25 : -0.4;
        double v = (base - (double)(2 * sta)) - (double)(2 * dyn);
        if (c.stall && was_done) v = 0.0;
        if (all_done) {
            v = v + 10.0;
            if (constraints == 0) v = v + 10.0;
        }
        rew = actl ? v : 0.0;
    }
    bool log_full = false;
    if (MAPS && (a.flags & DMFB_STEP_RECORD)) {  // addUsage (dmfb.py:459-463): append this step to the chip's usage log
        const uint32_t ent = (actl && nd != 0) ? (uint32_t)(r.x * c.L + r.y) : 0xffffu;
#ifndef DMFB_ABLATE_LOG
        if (c.lstride == 16) {  // one whole, aligned 32-byte sector per chip-step: even lanes store a pair of entries each
            const uint32_t hi = (uint32_t)dpp_mov<0x100 + 1>((int)ent);  // row_shl:1 = the next lane's entry (lane 15: 0 -> unused)
            if (act && (i & 1) == 0) {
                uint32_t *ul = (uint32_t *)(p.ulog + ((size_t)e * c.ucap + r.ulen) * 16);
                ul[i >> 1] = ent | ((i == 15 ? 0xffffu : hi) << 16);
            }
        } else if (actl) {
            p.ulog[((size_t)e * c.ucap + r.ulen) * c.lstride + i] = (uint16_t)ent;
        }
#endif
        if (act) {
            r.ulen += 1;
            log_full = (int)r.ulen == c.ucap;  // folded into the map right after this step: a step always finds room
        }
    }
    if (act) r.cum += (uint32_t)constraints;
    const bool in_time = (int)r.step < c.max_step;  // DMFBenv.step (dmfb.py:577-585)
    const bool success = in_time && all_done && r.cum == 0;
    const bool d = in_time ? (nd == 0) : true;
    const bool term = in_time ? all_done : true;
    if (actl) {
        if (a.out.d_dones) a.out.d_dones[(size_t)e * N + i] = (uint8_t)d;
        if (a.out.d_rewards) a.out.d_rewards[(size_t)e * N + i] = rew;
    }
    if (a.out.d_team_reward) {  // np.sum(list)/n in NumPy's order (rollout.py:33): sequential below 8 values, else 8 partial sums
        double s;
        if constexpr (N < 8) {
            s = LaneStep<N, MAPS>::template add_tail<1>(row_bcast_f64<0>(rew), rew);
        } else {
            constexpr int M = N - (N % 8);
            double q = rew;                                    // lanes 0..7: q[j] = rew[j] (+ rew[8 + j] when n == 16)
            if constexpr (M == 16) q = rew + row_up_f64<8>(rew);
            // ((q0+q1)+(q2+q3))+((q4+q5)+(q6+q7)): lane 0 ends up with the tree's root
            q = q + row_up_f64<1>(q);
            q = q + row_up_f64<2>(q);
            q = q + row_up_f64<4>(q);
            s = LaneStep<N, MAPS>::template add_tail<M>(q, rew);   // lane 0: + rew[M], rew[M+1], ... in index order
        }
        if (act && i == 0) a.out.d_team_reward[e] = s / (double)N;
    }
    if (act && i == 0) {
        if (a.out.d_constraints) a.out.d_constraints[e] = constraints;
        if (a.out.d_success) a.out.d_success[e] = (uint8_t)success;
        if (a.out.d_terminated) a.out.d_terminated[e] = (uint8_t)term;
    }
    if (any_dup_row) {  // FLAG_DUP follows the positions (k_step does the same after the moves)
        const bool dnow = DupRounds<15>::run(actl ? (r.x | (r.y << 8) | (1 << 16)) : 0);
        const bool row_dup = row_bits(__ballot(dnow), lane) != 0;
        if (dup) r.flags = row_dup ? (r.flags | FLAG_DUP) : (r.flags & ~FLAG_DUP);
    }
    ended = act && term && (a.flags & DMFB_STEP_AUTORESET);
    if (MAPS && act && (ended || log_full)) {  // the wave folds the log in below (flush_usage)
        flush_len = (int)r.ulen;
        flush_kind = ended ? 1 : 2;  // 1: + updateHealth (reset(new=False), dmfb.py:182-183)   2: log full
        r.ulen = 0;
    }
    // ---- episode boundary inside the launch: reset(new=False) for the rows that ended (whole wave per ended chip)
    unsigned long long m = __ballot(ended && i == 0);
    while (m) {
        const int src = __ffsll((long long)m) - 1;   // lane 0 of the ended row
        m &= m - 1;
        const int e_src = blockIdx.x * kLaneChips + (wave * kWave + src) / kLaneGroup;
        const uint32_t gid = c.env_id0 + (uint32_t)e_src;
        const uint32_t ep = (uint32_t)__shfl((int)r.rep, src, kWave);
        uint32_t pts[N];
        gen_task_wave<N>(c, gid, ep, pts);
        if (c.nb > 0) {
            const uint32_t blk = gen_blocks_wave<N>(c, gid, ep, pts);
            if (lane < c.nb) p.blocks[(size_t)lane * E + e_src] = blk;
        }
        if ((lane & ~(kLaneGroup - 1)) == src) {  // the ended row takes its new task: point i = start, point N + i = goal
            uint32_t sw = 0, gw = 0;
#pragma unroll
            for (int b = 0; b < N; ++b) {
                sw = (i >> 1) == b ? pts[b] : sw;
                gw = ((N + i) >> 1) == b ? pts[b] : gw;
            }
            const uint32_t sp = (sw >> (16 * (i & 1))) & 0xffff, gp = (gw >> (16 * ((N + i) & 1))) & 0xffff;
            if (i < N) { r.x = sp & 0xff; r.y = sp >> 8; r.gx = gp & 0xff; r.gy = gp >> 8; }
            r.rep += 1; r.step = 0; r.cum = 0; r.flags = 0;
        }
    }
    // ---- write the records back: even lanes store a packed pair of cells; lanes 0..4 one scalar word each
    {
        const uint32_t me = (uint32_t)r.x | ((uint32_t)r.y << 8), mg = (uint32_t)r.gx | ((uint32_t)r.gy << 8);
        const uint32_t nxt = (uint32_t)dpp_mov<0x100 + 1>((int)me), nxg = (uint32_t)dpp_mov<0x100 + 1>((int)mg);
        if (actl && (i & 1) == 0) {
            const uint32_t wv = me | ((i + 1 < N ? nxt : 0u) << 16), wg = mg | ((i + 1 < N ? nxg : 0u) << 16);
            p.st[(size_t)(R::W_POS + (i >> 1)) * E + e] = wv;
            if (ended) {
                p.st[(size_t)(R::W_GOAL + (i >> 1)) * E + e] = wg;
                p.starts[(size_t)(i >> 1) * E + e] = wv;   // store_starts: the new task's start cells
            }
        }
        if (act && i < 5) {
            const uint32_t v = i == 0 ? ((r.step & 0xffff) | (r.flags << 16) | (r.ulen << kUlenShift)) : i == 1 ? r.cum : i == 2 ? r.rstep : i == 3 ? r.rep : r.rmap;
            p.st[(size_t)(R::W_STEP + i) * E + e] = v;
        }
    }
    if (MAPS && (a.flags & (DMFB_STEP_AUTORESET | DMFB_STEP_RECORD))) {
        // usage logs of the chips that were reset (+ updateHealth) or whose log is full: the wave folds them one after another
        // through its own LDS histogram; the log entries of this step were stored by this wave: drain them first
        unsigned long long fm = __ballot(flush_kind != 0 && i == 0);
        if (fm) {
            __builtin_amdgcn_s_waitcnt(0);
            wave_fence();
            uint16_t *hist = c.hist_bytes ? (uint16_t *)(smem + (size_t)wave * c.hist_bytes) : nullptr;
            while (fm) {
                const int src = __ffsll((long long)fm) - 1;
                fm &= fm - 1;
                const int e_src = blockIdx.x * kLaneChips + (wave * kWave + src) / kLaneGroup;
                const int len = __shfl(flush_len, src, kWave), kind = __shfl(flush_kind, src, kWave);
                flush_usage(c, p, e_src, len, kind == 1, hist, lane);
            }
        }
    }
}

}  // namespace dmfbk
