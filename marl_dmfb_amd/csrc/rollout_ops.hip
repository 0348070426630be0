// rollout_ops.hip -- epsilon-greedy selection and per-step episode book-keeping of the lock-step rollout as two
// launches (include/rollout_ops.h).  Tiny, latency-bound kernels: the point is the launch count, not bandwidth.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/rollout_ops.h"

namespace {

__device__ __forceinline__ void philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__global__ void k_select(const float *__restrict__ q, int rows, int n, int A, const float *__restrict__ eps_p, int evaluate,
                         uint32_t k0, uint32_t k1, const uint32_t *__restrict__ draw_p, int32_t *__restrict__ actions, int8_t *__restrict__ last_onehot,
                         int8_t *__restrict__ ep_u, int8_t *__restrict__ ep_onehot, int T, int t) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float *qr = q + (size_t)r * A;
    int best = 0;
    float bv = qr[0];
    for (int k = 1; k < A; ++k) {
        const float v = qr[k];
        if (v > bv) { bv = v; best = k; }
    }
    int act = best;
    if (!evaluate) {
        uint32_t w[4];
        philox(k0, k1, (uint32_t)r, *draw_p, 0u, 0x600u, w);
        const float u1 = (float)(w[0] >> 8) * (1.0f / 16777216.0f);
        if (u1 < *eps_p) act = (int)__umulhi(w[1], (uint32_t)A);
    }
    actions[r] = act;
    int8_t *lo = last_onehot + (size_t)r * A;
    for (int k = 0; k < A; ++k) lo[k] = (int8_t)(k == act);
    if (ep_u) {
        const int e = r / n, a = r - e * n;
        const size_t slot = ((size_t)e * T + t) * n + a;
        ep_u[slot] = (int8_t)act;
        if (ep_onehot) {
            int8_t *eo = ep_onehot + slot * A;
            for (int k = 0; k < A; ++k) eo[k] = (int8_t)(k == act);
        }
    }
}

// GRU gate math (the formulas of aten's fused cell: hy = n + z (h - n)) + the Q head + the epsilon-greedy pick.  Hidden
// size 128: half a wave per (chip, droplet) row, lane l of the half owns units 4l..4l+3 (16-byte loads of the six gate
// pre-activations, h and the head weights); the head's dot products are 32-lane butterfly reductions.
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void k_gru_head_select(const float *__restrict__ ig, const float *__restrict__ hg, const float *__restrict__ b_ih,
                                                         const float *__restrict__ b_hh, float *__restrict__ h, const float *__restrict__ fc_w,
                                                         const float *__restrict__ fc_b, int rows, int n, int A,
                                                         const float *__restrict__ eps_p, int evaluate, uint32_t k0, uint32_t k1,
                                                         const uint32_t *__restrict__ draw_p, int32_t *__restrict__ actions,
                                                         int8_t *__restrict__ last_onehot, int8_t *__restrict__ ep_u, int8_t *__restrict__ ep_onehot,
                                                         int T, int t, float *__restrict__ q_out,
                                                         const int32_t *__restrict__ live_chips, const int32_t *__restrict__ n_live,
                                                         const int32_t *__restrict__ t_ep) {
    constexpr int H = 128;
    const int l32 = threadIdx.x & 31;
    // live_chips != NULL: only the rows of the listed chips; the x-side gates `ig` are COMPACT (row k*n + a of ig belongs to row
    // live_chips[k]*n + a of everything else), as crnn_front9_forward_live leaves them
    if (live_chips) rows = min(rows, n_live[0] * n);
    if (blockIdx.x * 8 >= rows) return;  // (uniform) nothing live in this workgroup
    const int cr = blockIdx.x * 8 + (threadIdx.x >> 5);
    const bool live = cr < rows;
    const int crc = live ? cr : rows - 1;  // dead half-waves shadow the last row (no stores) so that shuffles stay full-wave
    const int rc = live_chips ? live_chips[crc / n] * n + crc % n : crc;
    const int r = rc;
    const int u = 4 * l32;
    const float4 *igr = (const float4 *)(ig + (size_t)crc * 3 * H + u), *hgr = (const float4 *)(hg + (size_t)rc * 3 * H + u);
    const float4 i_r = igr[0], i_z = igr[H / 4], i_n = igr[2 * H / 4];
    const float4 h_r = hgr[0], h_z = hgr[H / 4], h_n = hgr[2 * H / 4];
    const float4 hx = *(const float4 *)(h + (size_t)rc * H + u);
    const float4 bi_r = *(const float4 *)(b_ih + u), bi_z = *(const float4 *)(b_ih + H + u), bi_n = *(const float4 *)(b_ih + 2 * H + u);
    const float4 bh_r = *(const float4 *)(b_hh + u), bh_z = *(const float4 *)(b_hh + H + u), bh_n = *(const float4 *)(b_hh + 2 * H + u);
    float hy[4];
    {
        const float ir[4] = {i_r.x, i_r.y, i_r.z, i_r.w}, iz[4] = {i_z.x, i_z.y, i_z.z, i_z.w}, in_[4] = {i_n.x, i_n.y, i_n.z, i_n.w};
        const float hr[4] = {h_r.x, h_r.y, h_r.z, h_r.w}, hz[4] = {h_z.x, h_z.y, h_z.z, h_z.w}, hn[4] = {h_n.x, h_n.y, h_n.z, h_n.w};
        const float b1r[4] = {bi_r.x, bi_r.y, bi_r.z, bi_r.w}, b1z[4] = {bi_z.x, bi_z.y, bi_z.z, bi_z.w}, b1n[4] = {bi_n.x, bi_n.y, bi_n.z, bi_n.w};
        const float b2r[4] = {bh_r.x, bh_r.y, bh_r.z, bh_r.w}, b2z[4] = {bh_z.x, bh_z.y, bh_z.z, bh_z.w}, b2n[4] = {bh_n.x, bh_n.y, bh_n.z, bh_n.w};
        const float hp[4] = {hx.x, hx.y, hx.z, hx.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float rg = sigm(ir[k] + hr[k] + b1r[k] + b2r[k]);
            const float zg = sigm(iz[k] + hz[k] + b1z[k] + b2z[k]);
            const float ng = tanhf(in_[k] + b1n[k] + rg * (hn[k] + b2n[k]));
            hy[k] = ng + zg * (hp[k] - ng);
        }
    }
    if (live) *(float4 *)(h + (size_t)r * H + u) = float4{hy[0], hy[1], hy[2], hy[3]};
    int best = 0;
    float bv = 0.0f;
    for (int a = 0; a < A; ++a) {
        const float4 w = *(const float4 *)(fc_w + a * H + u);
        float v = fmaf(hy[3], w.w, fmaf(hy[2], w.z, fmaf(hy[1], w.y, hy[0] * w.x)));
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
        v += fc_b[a];
        if (q_out && live && l32 == 0) q_out[(size_t)r * A + a] = v;
        if (a == 0 || v > bv) { bv = v; best = a; }
    }
    if (l32 != 0 || !live) return;
    int act = best;
    if (!evaluate) {
        uint32_t w[4];
        philox(k0, k1, (uint32_t)r, *draw_p, 0u, 0x600u, w);
        const float u1 = (float)(w[0] >> 8) * (1.0f / 16777216.0f);
        if (u1 < *eps_p) act = (int)__umulhi(w[1], (uint32_t)A);
    }
    actions[r] = act;
    int8_t *lo = last_onehot + (size_t)r * A;
    for (int k = 0; k < A; ++k) lo[k] = (int8_t)(k == act);
    if (ep_u) {
        const int e = r / n, a = r - e * n;
        if (t_ep) t = t_ep[e];   // continuous rollout: every chip is at its own step of its own episode
        const size_t slot = ((size_t)e * T + t) * n + a;
        ep_u[slot] = (int8_t)act;
        if (ep_onehot) {
            int8_t *eo = ep_onehot + slot * A;
            for (int k = 0; k < A; ++k) eo[k] = (int8_t)(k == act);
        }
    }
}

constexpr int kPostBlock = 256;

// ids of the chips with alive[e] != 0, ascending, and their count: ONE workgroup walks the chips in chunks of its size, a
// block-wide exclusive scan per chunk (stable, deterministic).  4096 chips = four chunks.
constexpr int kCompactBlock = 1024;
__global__ __launch_bounds__(kCompactBlock) void k_compact_alive(int E, const uint8_t *__restrict__ alive, int32_t *__restrict__ list,
                                                                 int32_t *__restrict__ count) {
    __shared__ int s_wave[kCompactBlock / 64];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int e0 = 0; e0 < E; e0 += kCompactBlock) {
        const int e = e0 + tid;
        const bool a = e < E && alive[e] != 0;
        const unsigned long long m = __ballot(a);
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (a) list[off + rank] = e;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < kCompactBlock / 64; ++w) tot += s_wave[w];
            s_base += tot;
        }
        __syncthreads();
    }
    if (tid == 0) count[0] = s_base;
}

// Episode observation appends of lock-step t (rollout.py:118-141 incl. the zero padding): with a = alive before the
// step, o_next[e][t] = a ? obs[e] : 0 and o[e][t+1] = (a && !term[e]) ? obs[e] : 0 (o[t+1] of a chip that plays on IS
// o_next[t]).  The episode tensors start zeroed, so masked rows are simply not written.  VEC = bytes per thread.
template <typename V>
__global__ void k_obs_append(const V *__restrict__ obs, int E, int row_v, int T, int t, const uint8_t *__restrict__ alive,
                             const uint8_t *__restrict__ term, V *__restrict__ ep_o, V *__restrict__ ep_o_next) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)E * row_v) return;
    const int e = (int)(idx / row_v), k = (int)(idx - (long)e * row_v);
    if (!alive[e]) return;
    const V v = obs[idx];
    if (ep_o_next) ep_o_next[((size_t)e * T + t) * row_v + k] = v;
    if (ep_o && t + 1 < T && !term[e]) ep_o[((size_t)e * T + t + 1) * row_v + k] = v;
}

// One thread per chip; the two live-chip counts go through a 4-int device workspace (ws[0] = alive after, published by
// the last workgroup; ws[1], ws[2] = running sums; ws[3] = ticket) so that any number of workgroups can take part.
__global__ __launch_bounds__(kPostBlock) void k_post(int E, int T, int t, uint8_t *__restrict__ alive, const uint8_t *__restrict__ term,
                                                     const double *__restrict__ team_reward, const void *__restrict__ constraints, int cons_f64,
                                                     const uint8_t *__restrict__ success, float *__restrict__ ep_r, uint8_t *__restrict__ ep_padded,
                                                     uint8_t *__restrict__ ep_term, double *__restrict__ sum_reward, double *__restrict__ sum_cons,
                                                     int64_t *__restrict__ sum_success, int64_t *__restrict__ steps, float *__restrict__ eps_p,
                                                     float anneal, float min_eps, int32_t *__restrict__ ws, uint32_t *__restrict__ draw_p) {
    __shared__ int s_cnt[2][kPostBlock / 64];
    __shared__ int s_last;
    int was = 0, now = 0;
    const int e = blockIdx.x * kPostBlock + threadIdx.x;
    if (e < E) {
        const int a = alive[e] != 0, tm = term[e] != 0;
        const double tr = team_reward[e];
        if (ep_r) ep_r[(size_t)e * T + t] = (float)tr;
        if (ep_padded) ep_padded[(size_t)e * T + t] = (uint8_t)!a;
        if (ep_term) ep_term[(size_t)e * T + t] = (uint8_t)tm;
        sum_reward[e] += tr;
        sum_cons[e] += cons_f64 ? ((const double *)constraints)[e] : (double)((const int32_t *)constraints)[e];
        sum_success[e] += success[e];
        steps[e] += a;
        const int na = a && !tm;
        alive[e] = (uint8_t)na;
        was = a;
        now = na;
    }
    for (int o = 32; o > 0; o >>= 1) { was += __shfl_down(was, o); now += __shfl_down(now, o); }
    if ((threadIdx.x & 63) == 0) { s_cnt[0][threadIdx.x >> 6] = was; s_cnt[1][threadIdx.x >> 6] = now; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int w = 0, nw = 0;
        for (int i = 0; i < kPostBlock / 64; ++i) { w += s_cnt[0][i]; nw += s_cnt[1][i]; }
        atomicAdd(&ws[1], w);
        atomicAdd(&ws[2], nw);
        __threadfence();
        s_last = atomicAdd(&ws[3], 1) == (int)gridDim.x - 1;
        if (s_last) {  // every workgroup has added its counts
            __threadfence();
            const int wt = atomicAdd(&ws[1], 0), nt = atomicAdd(&ws[2], 0);
            if (anneal > 0.0f) *eps_p = fmaxf(*eps_p - anneal * (float)wt, min_eps);
            ws[0] = nt;
            ws[1] = 0; ws[2] = 0; ws[3] = 0;
            if (draw_p) *draw_p += 1u;
        }
    }
}

// ---- continuous rollout (include/rollout_ops.h, "stream" entry points) --------------------------------------------------------
struct RingPtrs {
    int8_t *o, *o_next, *u, *u_onehot, *avail_u, *avail_u_next;
    float *r;
    uint8_t *padded, *terminated;
    int32_t *len;
    double *stats;
};

// bytes != 0 among the four of a word
__device__ __forceinline__ int nz_bytes(uint32_t w) { return __popc((w | ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u); }

// One workgroup per chip.  (1) the chip's row of this lock-step is appended to its staged episode and its running sums advance.
// (2) EVERY workgroup scans the termination flags of all chips (16 bytes per thread and pass, a block-wide prefix sum over the
// 16-chip groups): the k-th ended chip in ASCENDING chip order closes into ring slot (cursor + k) mod S -- deterministic whatever
// the scheduling (graph replay == eager play, bit for bit), and no single workgroup walks the chips alone.  (3) A chip whose
// episode ended writes the small tensors of its ring slot, its statistics, and clears its recurrent state (policy.init_hidden,
// rollout.py:112).  (4) The bulk of the close -- the 2 x T observation rows of every ended episode, with the padding of
// rollout.py:131-141 -- is cut into chunks of kChunk words and shared by ALL the workgroups of the launch (workgroup e takes chunks
// e, e + E, ...): one workgroup copying its own episode alone took 12 us (DMFB, 40 x 980 B) to 82 us (MEDA, 60 x 4340 B) even with
// sixteen loads in flight per thread, and any lock-step with a single close paid that.  The helpers read only what this launch does
// not write: staged rows of EARLIER steps, this step's row from the env's output, the step index through the double-buffered t_ep.
// The workgroup of the LAST chip publishes the new cursor / fill level / episode count (double-buffered ring state: the cursor is
// read by everybody while it is advanced), epsilon and the draw counter.
constexpr int kStreamBlock = 256;
constexpr int kStreamU = 8;                          // independent loads in flight per thread before the first store
constexpr int kChunk = kStreamBlock * kStreamU;      // words of the (o, o_next) index space per work item
constexpr int kMaxGroups = 2048;
constexpr int kCoopMinWords = 16384;                 // episodes of more words than this are copied by all workgroups together                     // 16-chip groups: E <= 32 768 (the fused-launch range of the env kernels)
template <typename V, bool COOP>
__global__ __launch_bounds__(kStreamBlock) void k_stream_step(int E, int n, int A, int T, int S, int row_v, int H, const V *__restrict__ obs_prev,
                                                     const V *__restrict__ obs_new, const V *__restrict__ obs_term, const uint8_t *__restrict__ term,
                                                     const double *__restrict__ team_reward, const void *__restrict__ constraints, int cons_f64,
                                                     const uint8_t *__restrict__ success, const int32_t *__restrict__ t_in, int32_t *__restrict__ t_out,
                                                     V *__restrict__ stage_o0, V *__restrict__ stage_o_next, const int8_t *__restrict__ stage_u,
                                                     const int8_t *__restrict__ stage_onehot, float *__restrict__ stage_r,
                                                     double *__restrict__ ep_acc, int64_t *__restrict__ chip_acc, int32_t *__restrict__ close_slot,
                                                     RingPtrs ring, const int64_t *__restrict__ state_in, int64_t *__restrict__ state_out,
                                                     float *__restrict__ hidden, int8_t *__restrict__ last_onehot, float *__restrict__ eps_p,
                                                     float anneal, float min_eps, uint32_t *__restrict__ draw_p) {
    __shared__ int s_pref[kMaxGroups + 1];   // ended chips in the groups below group g; [groups] = all of them
    __shared__ int s_wave[kStreamBlock / 64];
    __shared__ int s_base;
    const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = t_in[e];
    const bool tm = term[e] != 0;
    const V *on = (tm && obs_term ? obs_term : obs_new) + (size_t)e * row_v, *op = obs_prev + (size_t)e * row_v;
    V *so = stage_o_next + (size_t)e * T * row_v;
    for (int k = tid; k < row_v; k += kStreamBlock) {
        so[(size_t)t * row_v + k] = on[k];
        if (t == 0) stage_o0[(size_t)e * row_v + k] = op[k];
    }
    double rew = 0.0, cons = 0.0, succ = 0.0;
    if (tid == 0) {
        const double tr = team_reward[e];
        stage_r[(size_t)e * T + t] = (float)tr;
        rew = ep_acc[(size_t)e * 3] + tr;
        cons = ep_acc[(size_t)e * 3 + 1] + (cons_f64 ? ((const double *)constraints)[e] : (double)((const int32_t *)constraints)[e]);
        succ = ep_acc[(size_t)e * 3 + 2] + (double)success[e];
        ep_acc[(size_t)e * 3] = tm ? 0.0 : rew;
        ep_acc[(size_t)e * 3 + 1] = tm ? 0.0 : cons;
        ep_acc[(size_t)e * 3 + 2] = tm ? 0.0 : succ;
        chip_acc[(size_t)e * 4 + 3] += 1;   // env steps played
        t_out[e] = tm ? 0 : t + 1;
        if (!tm) close_slot[e] = -1;
    }
    // everything of a chip's close but the observation rows (called by the chip's own workgroup)
    auto close_small = [&](int slot, int rank) {
        (void)rank;
        const int len = t + 1;
        if (tid == 0) {
            const long infl = succ > 0.0 ? len : T;   // `steps` is forced to episode_limit when not successful (rollout.py:148-149)
            ring.len[slot] = len;
            ring.stats[(size_t)slot * 4] = rew;
            ring.stats[(size_t)slot * 4 + 1] = (double)infl;
            ring.stats[(size_t)slot * 4 + 2] = cons;
            ring.stats[(size_t)slot * 4 + 3] = succ;
            chip_acc[(size_t)e * 4] += 1;
            chip_acc[(size_t)e * 4 + 1] += infl;
            chip_acc[(size_t)e * 4 + 2] += succ > 0.0 ? 1 : 0;
            close_slot[e] = slot;
        }
        for (int i = tid; i < T * n * A; i += kStreamBlock) {
            const int tt = i / (n * A);
            const int8_t live = (int8_t)(tt < len);
            ring.u_onehot[(size_t)slot * T * n * A + i] = live ? stage_onehot[(size_t)e * T * n * A + i] : (int8_t)0;
            ring.avail_u[(size_t)slot * T * n * A + i] = live;
            ring.avail_u_next[(size_t)slot * T * n * A + i] = live;
        }
        for (int i = tid; i < T * n; i += kStreamBlock) ring.u[(size_t)slot * T * n + i] = i / n < len ? stage_u[(size_t)e * T * n + i] : (int8_t)0;
        for (int tt = tid; tt < T; tt += kStreamBlock) {
            ring.r[(size_t)slot * T + tt] = tt < len ? (tt == t ? (float)team_reward[e] : stage_r[(size_t)e * T + tt]) : 0.0f;
            ring.padded[(size_t)slot * T + tt] = (uint8_t)(tt >= len);
            ring.terminated[(size_t)slot * T + tt] = (uint8_t)(tt >= len - 1);
        }
        for (int i = tid; i < n * H; i += kStreamBlock) hidden[(size_t)e * n * H + i] = 0.0f;
        for (int i = tid; i < n * A; i += kStreamBlock) last_onehot[(size_t)e * n * A + i] = 0;
    };
    if constexpr (!COOP) {
        // Small episodes (T x row words <= kCoopMinWords: config A): the chip's own workgroup copies its episode -- ~12 us on a
        // lock-step with a close -- and a lock-step WITHOUT a close costs nothing beyond the staging: only workgroups of ended chips
        // (and the last one, which publishes the totals) look at the other chips' flags.  The shared form below costs every
        // launch ~2.5 us (flag scan + barrier in all 4096 workgroups) and pays off when one episode is tens of thousands of words.
        const bool last = e == E - 1;
        if (!tm && !last) return;   // (uniform)
        int c = 0;
        for (int i = tid * 16; i < e; i += kStreamBlock * 16) {
            if (i + 16 <= e && ((size_t)(term + i) & 15) == 0) {
                const uint4 w = *(const uint4 *)(term + i);
                c += nz_bytes(w.x) + nz_bytes(w.y) + nz_bytes(w.z) + nz_bytes(w.w);
            } else {
                for (int k = i; k < min(i + 16, e); ++k) c += term[k] != 0;
            }
        }
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
        if (lane == 0) s_wave[wave] = c;
        __syncthreads();
        int rank = 0;
        for (int w = 0; w < kStreamBlock / 64; ++w) rank += s_wave[w];
        const long cursor0 = state_in[0];
        if (last && tid == 0) {
            const long closed = rank + (tm ? 1 : 0);
            state_out[0] = (cursor0 + closed) % S;
            state_out[1] = min((long)S, state_in[1] + closed);
            state_out[2] = state_in[2] + closed;
            state_out[3] = state_in[3];
            if (anneal > 0.0f) *eps_p = fmaxf(*eps_p - anneal * (float)E, min_eps);
            if (draw_p) *draw_p += 1u;
        }
        if (!tm) return;
        const int slot = (int)((cursor0 + rank) % S);
        close_small(slot, rank);
        const int len = t + 1, total = T * row_v;
        V *ro = (V *)ring.o + (size_t)slot * T * row_v, *rn = (V *)ring.o_next + (size_t)slot * T * row_v;
        const V *o0 = t == 0 ? op : stage_o0 + (size_t)e * row_v;
        for (int base = tid; base < total; base += kChunk) {
            V vo[kStreamU], vn[kStreamU];
#pragma unroll
            for (int u = 0; u < kStreamU; ++u) {
                const int i = base + u * kStreamBlock;
                const int tt = i / row_v, k = i - tt * row_v;
                vo[u] = 0;
                vn[u] = 0;
                if (i < total && tt < len) {
                    vn[u] = tt == t ? on[k] : so[(size_t)tt * row_v + k];
                    vo[u] = tt == 0 ? o0[k] : so[(size_t)(tt - 1) * row_v + k];
                }
            }
#pragma unroll
            for (int u = 0; u < kStreamU; ++u) {
                const int i = base + u * kStreamBlock;
                if (i < total) { ro[i] = vo[u]; rn[i] = vn[u]; }
            }
        }
        return;
    }
    // ---- ended chips per 16-chip group, exclusive prefix over the groups.  A cheap first look (one 16-byte load per thread and
    // pass, ONE barrier) keeps the lock-steps in which no episode ended -- most of them under a policy that does not finish -- off
    // the scan
    const int groups = (E + 15) / 16;
    auto group_count = [&](int g) {
        int c = 0;
        if (g < groups) {
            const int i = g * 16;
            if (i + 16 <= E && ((size_t)(term + i) & 15) == 0) {
                const uint4 w = *(const uint4 *)(term + i);
                c = nz_bytes(w.x) + nz_bytes(w.y) + nz_bytes(w.z) + nz_bytes(w.w);
            } else {
                for (int k = i; k < min(i + 16, E); ++k) c += term[k] != 0;
            }
        }
        return c;
    };
    int seen = 0;
    for (int g0 = 0; g0 < groups; g0 += kStreamBlock) seen |= group_count(g0 + tid);
    if (tid == 0) s_base = 0;
    if (__syncthreads_or(seen)) {
        for (int g0 = 0; g0 < groups; g0 += kStreamBlock) {
            const int g = g0 + tid;
            const int c = group_count(g);
            int inc = c;   // inclusive prefix inside the wave
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(inc, o);
                if (lane >= o) inc += v;
            }
            if (lane == 63) s_wave[wave] = inc;
            __syncthreads();
            int off = s_base;
            for (int w = 0; w < wave; ++w) off += s_wave[w];
            if (g < groups) s_pref[g] = off + inc - c;
            __syncthreads();
            if (tid == 0) {
                int tot = 0;
                for (int w = 0; w < kStreamBlock / 64; ++w) tot += s_wave[w];
                s_base += tot;
            }
            __syncthreads();
        }
    }
    const int n_close = s_base;
    if (tid == 0) s_pref[groups] = n_close;
    const long cursor0 = state_in[0];
    if (e == E - 1 && tid == 0) {
        state_out[0] = (cursor0 + n_close) % S;
        state_out[1] = min((long)S, state_in[1] + n_close);
        state_out[2] = state_in[2] + n_close;
        state_out[3] = state_in[3];
        if (anneal > 0.0f) *eps_p = fmaxf(*eps_p - anneal * (float)E, min_eps);   // every chip played a step (rollout.py:126-127)
        if (draw_p) *draw_p += 1u;
    }
    if (n_close == 0) return;   // (uniform)
    __syncthreads();            // s_pref[groups]
    // ---- the chip's own close: everything but the observation rows
    if (tm) {
        int rank = s_pref[e >> 4];
        for (int k = e & ~15; k < e; ++k) rank += term[k] != 0;
        close_small((int)((cursor0 + rank) % S), rank);
    }
    // ---- the observation rows of every ended episode, shared by all workgroups.  o[tt] = first observation (tt == 0) or
    // o_next[tt - 1]; o_next[tt] as staged; the row of THIS step comes from the env's output (its staged copy is being written by
    // the chip's own workgroup in this very launch)
    const int total = T * row_v;
    const int chunks = (total + kChunk - 1) / kChunk;
    for (long j = e; j < (long)n_close * chunks; j += E) {
        const int ci = (int)(j / chunks), ch = (int)(j - (long)ci * chunks);
        int lo = 0, hi = groups;   // last group whose prefix is <= ci (broadcast LDS reads)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_pref[mid] <= ci) lo = mid; else hi = mid;
        }
        int c = lo * 16, left = ci - s_pref[lo];
        for (;; ++c) {   // the (left + 1)-th ended chip of the group
            if (term[c] != 0 && left-- == 0) break;
        }
        const int tc = t_in[c], len = tc + 1;
        const int slot = (int)((cursor0 + ci) % S);
        const V *onc = (obs_term ? obs_term : obs_new) + (size_t)c * row_v;
        const V *soc = stage_o_next + (size_t)c * T * row_v;
        const V *o0 = tc == 0 ? obs_prev + (size_t)c * row_v : stage_o0 + (size_t)c * row_v;
        V *ro = (V *)ring.o + (size_t)slot * T * row_v, *rn = (V *)ring.o_next + (size_t)slot * T * row_v;
        V vo[kStreamU], vn[kStreamU];
#pragma unroll
        for (int u = 0; u < kStreamU; ++u) {
            const int i = ch * kChunk + u * kStreamBlock + tid;
            const int tt = i / row_v, k = i - tt * row_v;
            vo[u] = 0;
            vn[u] = 0;
            if (i < total && tt < len) {
                vn[u] = tt == tc ? onc[k] : soc[(size_t)tt * row_v + k];
                vo[u] = tt == 0 ? o0[k] : soc[(size_t)(tt - 1) * row_v + k];
            }
        }
#pragma unroll
        for (int u = 0; u < kStreamU; ++u) {
            const int i = ch * kChunk + u * kStreamBlock + tid;
            if (i < total) { ro[i] = vo[u]; rn[i] = vn[u]; }
        }
    }
}

thread_local int g_last = 0;

int finish() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last = (int)e;
        if (getenv("DMFB_VEC_DEBUG")) fprintf(stderr, "rollout_ops: launch failed: %s\n", hipGetErrorString(e));
        return ROLLOUT_ERR_HIP;
    }
    return ROLLOUT_OK;
}

}  // namespace

extern "C" {

int rollout_select_actions(const float *d_q, int32_t n_envs, int32_t n_agents, int32_t n_actions, const float *d_epsilon,
                           int32_t evaluate, uint64_t seed, const uint32_t *d_draw, int32_t *d_actions, int8_t *d_last_onehot,
                           int8_t *d_ep_u, int8_t *d_ep_onehot, int32_t episode_limit, int32_t t, void *stream) {
    if (!d_q || !d_actions || !d_last_onehot || n_envs < 0 || n_agents < 1 || n_actions < 1 || n_actions > 127 ||
        (!evaluate && (!d_epsilon || !d_draw)) || (d_ep_u && (t < 0 || t >= episode_limit)))
        return ROLLOUT_ERR_BAD_ARG;
    const long rows = (long)n_envs * n_agents;
    if (rows == 0) return ROLLOUT_OK;
    if (rows > 0x7fffffffL) return ROLLOUT_ERR_BAD_ARG;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_select, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_q, (int)rows, n_agents, n_actions,
                       d_epsilon, evaluate, (uint32_t)seed, (uint32_t)(seed >> 32), d_draw, d_actions, d_last_onehot, d_ep_u, d_ep_onehot,
                       episode_limit, t);
    return finish();
}

static int gru_head_select_impl(const float *d_igates, const float *d_hgates, const float *d_b_ih, const float *d_b_hh, float *d_h,
                                const float *d_fc_w, const float *d_fc_b, int32_t n_envs, int32_t n_agents, int32_t hidden,
                                int32_t n_actions, const float *d_epsilon, int32_t evaluate, uint64_t seed, const uint32_t *d_draw,
                                int32_t *d_actions, int8_t *d_last_onehot, int8_t *d_ep_u, int8_t *d_ep_onehot, int32_t episode_limit,
                                int32_t t, float *d_q, const int32_t *d_live_chips, const int32_t *d_n_live, void *stream,
                                const int32_t *d_t_ep = nullptr) {
    if (!d_igates || !d_hgates || !d_b_ih || !d_b_hh || !d_h || !d_fc_w || !d_fc_b || !d_actions || !d_last_onehot || n_envs < 0 ||
        n_agents < 1 || hidden != 128 || n_actions < 1 || n_actions > 16 || (!evaluate && (!d_epsilon || !d_draw)) ||
        (d_ep_u && !d_t_ep && (t < 0 || t >= episode_limit)) || ((d_live_chips == nullptr) != (d_n_live == nullptr)))
        return ROLLOUT_ERR_BAD_ARG;
    const long rows = (long)n_envs * n_agents;
    if (rows == 0) return ROLLOUT_OK;
    if (rows > 0x7fffffffL) return ROLLOUT_ERR_BAD_ARG;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_gru_head_select, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, d_igates, d_hgates, d_b_ih,
                       d_b_hh, d_h, d_fc_w, d_fc_b, (int)rows, n_agents, n_actions, d_epsilon, evaluate, (uint32_t)seed,
                       (uint32_t)(seed >> 32), d_draw, d_actions, d_last_onehot, d_ep_u, d_ep_onehot, episode_limit, t, d_q,
                       d_live_chips, d_n_live, d_t_ep);
    return finish();
}

int rollout_gru_head_select(const float *d_igates, const float *d_hgates, const float *d_b_ih, const float *d_b_hh, float *d_h,
                            const float *d_fc_w, const float *d_fc_b, int32_t n_envs, int32_t n_agents, int32_t hidden,
                            int32_t n_actions, const float *d_epsilon, int32_t evaluate, uint64_t seed, const uint32_t *d_draw,
                            int32_t *d_actions, int8_t *d_last_onehot, int8_t *d_ep_u, int8_t *d_ep_onehot, int32_t episode_limit,
                            int32_t t, float *d_q, void *stream) {
    return gru_head_select_impl(d_igates, d_hgates, d_b_ih, d_b_hh, d_h, d_fc_w, d_fc_b, n_envs, n_agents, hidden, n_actions, d_epsilon,
                                evaluate, seed, d_draw, d_actions, d_last_onehot, d_ep_u, d_ep_onehot, episode_limit, t, d_q, nullptr,
                                nullptr, stream);
}

int rollout_gru_head_select_live(const float *d_igates, const float *d_hgates, const float *d_b_ih, const float *d_b_hh, float *d_h,
                                 const float *d_fc_w, const float *d_fc_b, int32_t n_envs, int32_t n_agents, int32_t hidden,
                                 int32_t n_actions, const float *d_epsilon, int32_t evaluate, uint64_t seed, const uint32_t *d_draw,
                                 int32_t *d_actions, int8_t *d_last_onehot, int8_t *d_ep_u, int8_t *d_ep_onehot, int32_t episode_limit,
                                 int32_t t, float *d_q, const int32_t *d_live_chips, const int32_t *d_n_live, void *stream) {
    if (!d_live_chips || !d_n_live) return ROLLOUT_ERR_BAD_ARG;
    return gru_head_select_impl(d_igates, d_hgates, d_b_ih, d_b_hh, d_h, d_fc_w, d_fc_b, n_envs, n_agents, hidden, n_actions, d_epsilon,
                                evaluate, seed, d_draw, d_actions, d_last_onehot, d_ep_u, d_ep_onehot, episode_limit, t, d_q, d_live_chips,
                                d_n_live, stream);
}

int rollout_compact_alive(int32_t n_envs, const uint8_t *d_alive, int32_t *d_live_chips, int32_t *d_n_live, void *stream) {
    if (n_envs < 0 || !d_alive || !d_live_chips || !d_n_live) return ROLLOUT_ERR_BAD_ARG;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_compact_alive, dim3(1), dim3(kCompactBlock), 0, (hipStream_t)stream, n_envs, d_alive, d_live_chips, d_n_live);
    return finish();
}

int rollout_post_step(int32_t n_envs, int32_t episode_limit, int32_t t, uint8_t *d_alive, const uint8_t *d_term,
                      const double *d_team_reward, const void *d_constraints, int32_t constraints_f64,
                      const uint8_t *d_success, float *d_ep_r, uint8_t *d_ep_padded, uint8_t *d_ep_terminated,
                      double *d_sum_reward, double *d_sum_constraints, int64_t *d_sum_success, int64_t *d_steps,
                      float *d_epsilon, float anneal, float min_epsilon, int32_t *d_n_alive, uint32_t *d_draw,
                      const int8_t *d_obs, int32_t obs_row_bytes, int8_t *d_ep_o, int8_t *d_ep_o_next, void *stream) {
    if (!d_alive || !d_term || !d_team_reward || !d_constraints || !d_success || !d_sum_reward || !d_sum_constraints ||
        !d_sum_success || !d_steps || !d_n_alive || n_envs < 0 || (anneal > 0.0f && !d_epsilon) ||
        ((d_ep_r || d_ep_padded || d_ep_terminated || d_ep_o || d_ep_o_next) && (t < 0 || t >= episode_limit)) ||
        ((d_ep_o || d_ep_o_next) && (!d_obs || obs_row_bytes < 1)))
        return ROLLOUT_ERR_BAD_ARG;
    if (n_envs == 0) return ROLLOUT_OK;
    (void)hipGetLastError();
    if (d_ep_o || d_ep_o_next) {  // reads d_alive BEFORE k_post (same stream) updates it
        const bool dw = obs_row_bytes % 4 == 0 && ((size_t)d_obs | (size_t)d_ep_o | (size_t)d_ep_o_next) % 4 == 0;
        const int row_v = dw ? obs_row_bytes / 4 : obs_row_bytes;
        const long n = (long)n_envs * row_v;
        if (dw)
            hipLaunchKernelGGL((k_obs_append<uint32_t>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)d_obs,
                               n_envs, row_v, episode_limit, t, d_alive, d_term, (uint32_t *)d_ep_o, (uint32_t *)d_ep_o_next);
        else
            hipLaunchKernelGGL((k_obs_append<int8_t>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_obs, n_envs, row_v,
                               episode_limit, t, d_alive, d_term, d_ep_o, d_ep_o_next);
    }
    hipLaunchKernelGGL(k_post, dim3((unsigned)((n_envs + kPostBlock - 1) / kPostBlock)), dim3(kPostBlock), 0, (hipStream_t)stream, n_envs, episode_limit, t, d_alive, d_term, d_team_reward,
                       d_constraints, constraints_f64, d_success, d_ep_r, d_ep_padded, d_ep_terminated, d_sum_reward, d_sum_constraints,
                       d_sum_success, d_steps, d_epsilon, anneal, min_epsilon, d_n_alive, d_draw);
    return finish();
}

int rollout_gru_head_select_stream(const float *d_igates, const float *d_hgates, const float *d_b_ih, const float *d_b_hh, float *d_h,
                                   const float *d_fc_w, const float *d_fc_b, int32_t n_envs, int32_t n_agents, int32_t hidden,
                                   int32_t n_actions, const float *d_epsilon, int32_t evaluate, uint64_t seed, const uint32_t *d_draw,
                                   int32_t *d_actions, int8_t *d_last_onehot, int8_t *d_stage_u, int8_t *d_stage_onehot,
                                   int32_t episode_limit, const int32_t *d_t_ep, float *d_q, void *stream) {
    if (!d_t_ep || !d_stage_u) return ROLLOUT_ERR_BAD_ARG;
    return gru_head_select_impl(d_igates, d_hgates, d_b_ih, d_b_hh, d_h, d_fc_w, d_fc_b, n_envs, n_agents, hidden, n_actions, d_epsilon,
                                evaluate, seed, d_draw, d_actions, d_last_onehot, d_stage_u, d_stage_onehot, episode_limit, 0, d_q, nullptr,
                                nullptr, stream, d_t_ep);
}

int rollout_stream_step(int32_t n_envs, int32_t n_agents, int32_t n_actions, int32_t episode_limit, int32_t obs_row_bytes,
                        int32_t hidden, const int8_t *d_obs_prev, const int8_t *d_obs_new, const int8_t *d_obs_term, const uint8_t *d_term,
                        const double *d_team_reward, const void *d_constraints, int32_t constraints_f64, const uint8_t *d_success,
                        const rollout_stage *stage, const rollout_ring *ring, int32_t parity, float *d_hidden, int8_t *d_last_onehot,
                        float *d_epsilon, float anneal, float min_epsilon, uint32_t *d_draw, void *stream) {
    if (!ring || !stage || !d_obs_prev || !d_obs_new || !d_term || !d_team_reward || !d_constraints || !d_success || !d_hidden ||
        !d_last_onehot || n_envs < 0 || n_agents < 1 || n_actions < 1 || episode_limit < 1 || obs_row_bytes < 1 || hidden < 1 ||
        ring->slots < 1 || n_envs > 16 * kMaxGroups || !stage->d_t_ep || !stage->d_close_slot || !stage->d_o0 || !stage->d_o_next || !stage->d_u || !stage->d_onehot ||
        !stage->d_r || !stage->d_ep_acc || !stage->d_chip_acc || !stage->d_state_alt || !ring->d_o || !ring->d_o_next || !ring->d_u ||
        !ring->d_u_onehot || !ring->d_avail_u || !ring->d_avail_u_next || !ring->d_r || !ring->d_padded || !ring->d_terminated ||
        !ring->d_len || !ring->d_stats || !ring->d_state || (anneal > 0.0f && !d_epsilon))
        return ROLLOUT_ERR_BAD_ARG;
    if (n_envs == 0) return ROLLOUT_OK;
    const RingPtrs rp{ring->d_o, ring->d_o_next, ring->d_u, ring->d_u_onehot, ring->d_avail_u, ring->d_avail_u_next, ring->d_r,
                      ring->d_padded, ring->d_terminated, ring->d_len, ring->d_stats};
    const int64_t *st_in = parity ? stage->d_state_alt : ring->d_state;
    int64_t *st_out = parity ? ring->d_state : stage->d_state_alt;
    const int32_t *t_in = stage->d_t_ep + (parity ? n_envs : 0);   // d_t_ep is int32[2][E]: the step index is double-buffered too
    int32_t *t_out = stage->d_t_ep + (parity ? 0 : n_envs);
    const bool dw = obs_row_bytes % 4 == 0 &&
                    ((size_t)d_obs_prev | (size_t)d_obs_new | (size_t)d_obs_term | (size_t)stage->d_o0 | (size_t)stage->d_o_next | (size_t)ring->d_o | (size_t)ring->d_o_next) % 4 == 0;
    (void)hipGetLastError();
    const bool coop = (long)episode_limit * (dw ? obs_row_bytes / 4 : obs_row_bytes) > kCoopMinWords;
    if (dw && coop)
        hipLaunchKernelGGL((k_stream_step<uint32_t, true>), dim3((unsigned)n_envs), dim3(kStreamBlock), 0, (hipStream_t)stream, n_envs, n_agents, n_actions,
                           episode_limit, ring->slots, obs_row_bytes / 4, hidden, (const uint32_t *)d_obs_prev, (const uint32_t *)d_obs_new,
                           (const uint32_t *)d_obs_term, d_term, d_team_reward, d_constraints, constraints_f64, d_success, t_in, t_out, (uint32_t *)stage->d_o0,
                           (uint32_t *)stage->d_o_next, stage->d_u, stage->d_onehot, stage->d_r, stage->d_ep_acc, stage->d_chip_acc,
                           stage->d_close_slot, rp, st_in, st_out, d_hidden, d_last_onehot, d_epsilon, anneal, min_epsilon, d_draw);
    else if (dw)
        hipLaunchKernelGGL((k_stream_step<uint32_t, false>), dim3((unsigned)n_envs), dim3(kStreamBlock), 0, (hipStream_t)stream, n_envs, n_agents, n_actions,
                           episode_limit, ring->slots, obs_row_bytes / 4, hidden, (const uint32_t *)d_obs_prev, (const uint32_t *)d_obs_new,
                           (const uint32_t *)d_obs_term, d_term, d_team_reward, d_constraints, constraints_f64, d_success, t_in, t_out, (uint32_t *)stage->d_o0,
                           (uint32_t *)stage->d_o_next, stage->d_u, stage->d_onehot, stage->d_r, stage->d_ep_acc, stage->d_chip_acc,
                           stage->d_close_slot, rp, st_in, st_out, d_hidden, d_last_onehot, d_epsilon, anneal, min_epsilon, d_draw);
    else   // (byte rows: the shared form)
        hipLaunchKernelGGL((k_stream_step<int8_t, true>), dim3((unsigned)n_envs), dim3(kStreamBlock), 0, (hipStream_t)stream, n_envs, n_agents, n_actions,
                           episode_limit, ring->slots, obs_row_bytes, hidden, d_obs_prev, d_obs_new, d_obs_term, d_term, d_team_reward, d_constraints,
                           constraints_f64, d_success, t_in, t_out, stage->d_o0, stage->d_o_next, stage->d_u, stage->d_onehot, stage->d_r,
                           stage->d_ep_acc, stage->d_chip_acc, stage->d_close_slot, rp, st_in, st_out, d_hidden, d_last_onehot, d_epsilon,
                           anneal, min_epsilon, d_draw);
    return finish();
}

int rollout_last_hip_error(void) { return g_last; }

}  // extern "C"
