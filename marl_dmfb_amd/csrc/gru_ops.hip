// gru_ops.hip -- the GRU cell of the reference's Q-network (network/base_net.py:56,69: nn.GRUCell(.., 128))
// unrolled over a whole episode inside ONE kernel launch, forward and backward, for gfx950.
//
// Why: VDN.learn unrolls the cell over T = 40..200 time steps (policy/vdn.py:174-191).  Through library
// calls every step is a chain of 3-8 launches of a few microseconds each on a (B*n) x 128 state -- latency,
// not work.  The recurrence is independent per row, so a workgroup can own a block of rows for the whole
// sequence and keep everything it needs on chip:
//   * W_hh (3H x H = 196 KB fp32) does not fit LDS; it is spread over the REGISTERS of the 512 threads:
//     thread (q, u) holds the 3 gate rows of hidden unit u restricted to the k-range [32q, 32q+32) (96 floats);
//   * h of the workgroup's RW rows lives in LDS; per step every thread forms its 3 partial dot products per
//     row from 16-byte broadcast LDS reads, the 4 k-quarters are reduced through LDS, then thread (q, u)
//     finishes the gates of rows 2q, 2q+1;
//   * the backward pass walks t = T-1..0 with the transposed register layout (thread (q, j) holds
//     W_hh[g*H + 32q + k][j]) for dh_{t-1} += W_hh^T d_hgates.
// The input projection x @ W_ih^T (all steps at once) and the weight gradients (from the stacked gate
// gradients) stay large rocBLAS GEMMs outside.  fp32, same formulas as torch's fused GRU cell:
//   r = s(i_r+b_ir+h_r+b_hr)  z = s(i_z+b_iz+h_z+b_hz)  n = tanh(i_n+b_in + r*(h_n+b_hn))  h' = (1-z)*n + z*h
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/crnn_ops.h"

namespace {

constexpr int kBlock = 512;
constexpr int H = 128;   // rnn_hidden_dim of every shipped config
constexpr int RW = 8;    // rows per workgroup: thread (q, u) finishes rows 2q and 2q+1
constexpr int KQ = 32;   // k-range per thread

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// Packed sequences (include/crnn_ops.h: gru_seq_forward_packed): the rows are sorted by sequence length, longest first; step t
// holds only the rows whose sequence is at least t + 1 long, stored back to back: rows [off[t], off[t + 1]) of every per-step
// tensor are rows 0 .. off[t + 1] - off[t] - 1 of the batch.  A kernel argument (scalar loads), no device buffer.
struct SeqOff {
    int32_t packed;
    int32_t off[GRU_SEQ_MAX_STEPS + 1];
};

// The dot products run on TWO rows at a time: (row 2p, row 2p + 1) sit next to each other in LDS, so one weight (broadcast) times
// such a pair is a v_pk_fma_f32 on a natural register pair -- half the instructions of the scalar chains, the same sums in the
// same order per row.  Written as an explicit 2-vector fma (from scalar code the compiler formed no packed instruction here), with
// the weights held as 4-vectors: a broadcast operand costs nothing when it is an element of a register tuple, and a move when
// it is a lone register.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kBlock) void k_gru_seq_fwd(const float *__restrict__ igates, const float *__restrict__ h0,
                                                        const float *__restrict__ w_hh, const float *__restrict__ b_ih,
                                                        const float *__restrict__ b_hh, int T, long R,
                                                        float *__restrict__ hs, float *__restrict__ gates, const SeqOff so) {
    __shared__ __attribute__((aligned(16))) float s_h[RW / 2][H][2];   // h of rows (2p, 2p + 1), interleaved
    __shared__ __attribute__((aligned(16))) float s_part[4][RW][3][H];
    const int tid = threadIdx.x, q = tid / H, u = tid - q * H;
    const long row0 = (long)blockIdx.x * RW;
    const int rv = (int)min((long)RW, R - row0);
    f32x4 w[3][KQ / 4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < KQ / 4; ++k) w[g][k] = *(const f32x4 *)(w_hh + (size_t)(g * H + u) * H + q * KQ + 4 * k);
    const float bir = b_ih[u] + b_hh[u], biz = b_ih[H + u] + b_hh[H + u], bin = b_ih[2 * H + u], bhn = b_hh[2 * H + u];
    for (int i = tid; i < RW * H; i += kBlock) {
        const int rr = i / H, c = i - rr * H;
        s_h[rr >> 1][c][rr & 1] = rr < rv ? h0[(row0 + rr) * H + c] : 0.0f;
    }
    __syncthreads();
    const int ra = 2 * q;  // this thread finishes rows ra and ra + 1
    for (int t = 0; t < T; ++t) {
        // packed: rows still running at step t = the first rt rows of the batch; a workgroup is done once its first row is
        // (lengths only shrink along the rows); row (t, r) of the per-step tensors is row base + r
        const long rt = so.packed ? (long)(so.off[t + 1] - so.off[t]) : R;
        if (row0 >= rt) break;  // uniform
        const int rvt = (int)min((long)rv, rt - row0);
        const size_t base = so.packed ? (size_t)so.off[t] : (size_t)t * R;
        // input gates of my two rows (independent of the recurrence: issued first, consumed after the reduction)
        float ig[2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int rr = ra + s;
#pragma unroll
            for (int g = 0; g < 3; ++g) ig[s][g] = rr < rvt ? igates[(base + row0 + rr) * 3 * H + g * H + u] : 0.0f;
        }
        // partial h @ W_hh^T over my k-quarter, all RW rows, two rows per instruction
#pragma unroll
        for (int rp = 0; rp < RW / 2; ++rp) {
            const f32x4 *ph = (const f32x4 *)(&s_h[rp][q * KQ][0]);
            f32x2 acc[3] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
            for (int j = 0; j < KQ / 4; ++j) {   // k = 4j .. 4j + 3: two 16-byte reads of (h_2p[k], h_2p+1[k]) pairs
                const f32x4 v0 = ph[2 * j], v1 = ph[2 * j + 1];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const f32x4 wv = w[g][j];
                    acc[g] = __builtin_elementwise_fma((f32x2){wv.x, wv.x}, (f32x2){v0.x, v0.y}, acc[g]);
                    acc[g] = __builtin_elementwise_fma((f32x2){wv.y, wv.y}, (f32x2){v0.z, v0.w}, acc[g]);
                    acc[g] = __builtin_elementwise_fma((f32x2){wv.z, wv.z}, (f32x2){v1.x, v1.y}, acc[g]);
                    acc[g] = __builtin_elementwise_fma((f32x2){wv.w, wv.w}, (f32x2){v1.z, v1.w}, acc[g]);
                }
            }
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                s_part[q][2 * rp][g][u] = acc[g].x;
                s_part[q][2 * rp + 1][g][u] = acc[g].y;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int rr = ra + s;
            if (rr < rvt) {
                const float hr = s_part[0][rr][0][u] + s_part[1][rr][0][u] + s_part[2][rr][0][u] + s_part[3][rr][0][u];
                const float hz = s_part[0][rr][1][u] + s_part[1][rr][1][u] + s_part[2][rr][1][u] + s_part[3][rr][1][u];
                const float hn = s_part[0][rr][2][u] + s_part[1][rr][2][u] + s_part[2][rr][2][u] + s_part[3][rr][2][u] + bhn;
                const float rg = sigmoidf_(ig[s][0] + hr + bir);
                const float zg = sigmoidf_(ig[s][1] + hz + biz);
                const float ng = tanhf(ig[s][2] + bin + rg * hn);
                const float hp = s_h[rr >> 1][u][rr & 1];
                const float hnew = (1.0f - zg) * ng + zg * hp;
                const size_t o = base + row0 + rr;
                hs[o * H + u] = hnew;
                if (gates) {
                    gates[o * 4 * H + u] = rg; gates[o * 4 * H + H + u] = zg; gates[o * 4 * H + 2 * H + u] = ng;
                    gates[o * 4 * H + 3 * H + u] = hn;
                }
                s_h[rr >> 1][u][rr & 1] = hnew;  // only this thread reads this element between the two barriers
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kBlock) void k_gru_seq_bwd(const float *__restrict__ grad_hs, const float *__restrict__ gates,
                                                        const float *__restrict__ hs, const float *__restrict__ h0,
                                                        const float *__restrict__ w_hh, int T, long R,
                                                        float *__restrict__ d_ig, float *__restrict__ d_hg,
                                                        float *__restrict__ d_h0, float *__restrict__ bias_part, const SeqOff so,
                                                        float *__restrict__ h_prev_out) {
    __shared__ __attribute__((aligned(16))) float s_dhg[RW / 2][3][H][2];  // gate gradients of rows (2p, 2p + 1), interleaved
    __shared__ __attribute__((aligned(16))) float s_part[4][RW][H];
    const int tid = threadIdx.x, q = tid / H, u = tid - q * H;
    const long row0 = (long)blockIdx.x * RW;
    const int rv = (int)min((long)RW, R - row0);
    f32x4 wt[3][KQ / 4];  // W_hh[g*H + 32q + k][u]: column u of the rows of my k-quarter, four k per register tuple
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < KQ / 4; ++k) {
            const float *col = w_hh + (size_t)(g * H + q * KQ + 4 * k) * H + u;
            wt[g][k] = (f32x4){col[0], col[H], col[2 * H], col[3 * H]};
        }
    const int ra = 2 * q;
    float gh[2] = {0.0f, 0.0f};  // dL/dh_t arriving from the future for my two (row, u) entries
    float bs_r = 0.0f, bs_z = 0.0f, bs_n = 0.0f, bs_hn = 0.0f;  // column sums of the gate gradients (bias gradients)
    // steps this workgroup takes part in: 0 .. t_last (packed: the rows are sorted by length, the running rows only shrink with t)
    int t_last = -1;
    for (int t = 0; t < T; ++t)
        if (row0 < (so.packed ? (long)(so.off[t + 1] - so.off[t]) : R)) t_last = t;
    // The six inputs of a (row, unit) entry do not depend on the recurrence: those of step t - 1 are requested while step t's dot
    // products run, so that no step starts with a global-memory round trip.
    struct In { float g, rg, zg, ng, hn, hp; };
    In cur[2], nx[2];
    auto fetch = [&](int t, In (&in)[2]) __attribute__((always_inline)) {
        const long rt = so.packed ? (long)(so.off[t + 1] - so.off[t]) : R;
        const int rvt = (int)min((long)rv, rt - row0);
        const size_t base = so.packed ? (size_t)so.off[t] : (size_t)t * R;
        const size_t base_prev = so.packed ? (size_t)so.off[t > 0 ? t - 1 : 0] : (size_t)(t > 0 ? t - 1 : 0) * R;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int rr = ra + s;
            const bool on = rr < rvt;
            const size_t o = base + row0 + (on ? rr : 0);
            in[s].g = on ? grad_hs[o * H + u] : 0.0f;
            in[s].rg = on ? gates[o * 4 * H + u] : 0.0f;
            in[s].zg = on ? gates[o * 4 * H + H + u] : 0.0f;
            in[s].ng = on ? gates[o * 4 * H + 2 * H + u] : 0.0f;
            in[s].hn = on ? gates[o * 4 * H + 3 * H + u] : 0.0f;
            in[s].hp = !on ? 0.0f : t > 0 ? hs[(base_prev + row0 + rr) * H + u] : h0[(row0 + rr) * H + u];
        }
    };
    if (t_last >= 0) fetch(t_last, cur);
    for (int t = t_last; t >= 0; --t) {
        const long rt = so.packed ? (long)(so.off[t + 1] - so.off[t]) : R;   // rows running at step t (see k_gru_seq_fwd)
        const int rvt = (int)min((long)rv, rt - row0);
        const size_t base = so.packed ? (size_t)so.off[t] : (size_t)t * R;
        float direct[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int rr = ra + s;
            float dpr = 0.0f, dpz = 0.0f, dpn = 0.0f, dhn = 0.0f;
            direct[s] = 0.0f;
            if (rr < rvt) {
                const size_t o = base + row0 + rr;
                const float g = cur[s].g + gh[s];
                const float rg = cur[s].rg, zg = cur[s].zg, ng = cur[s].ng, hn = cur[s].hn;
                const float hp = cur[s].hp;
                if (h_prev_out) h_prev_out[o * H + u] = hp;  // h_{t-1} in the layout of d_hgates: dW_hh = d_hgates^T h_prev is ONE GEMM
                const float dn = g * (1.0f - zg), dz = g * (hp - ng);
                direct[s] = g * zg;
                dpn = dn * (1.0f - ng * ng);
                dhn = dpn * rg;
                dpr = dpn * hn * rg * (1.0f - rg);
                dpz = dz * zg * (1.0f - zg);
                d_ig[o * 3 * H + u] = dpr; d_ig[o * 3 * H + H + u] = dpz; d_ig[o * 3 * H + 2 * H + u] = dpn;
                d_hg[o * 3 * H + u] = dpr; d_hg[o * 3 * H + H + u] = dpz; d_hg[o * 3 * H + 2 * H + u] = dhn;
                bs_r += dpr; bs_z += dpz; bs_n += dpn; bs_hn += dhn;
            }
            s_dhg[rr >> 1][0][u][rr & 1] = dpr; s_dhg[rr >> 1][1][u][rr & 1] = dpz; s_dhg[rr >> 1][2][u][rr & 1] = dhn;
        }
        if (t > 0) fetch(t - 1, nx);
        __syncthreads();
        // partial W_hh^T d_hgates over my k-quarter of the gate rows, output column u, all RW rows, two rows per instruction
#pragma unroll
        for (int rp = 0; rp < RW / 2; ++rp) {
            f32x2 acc = {0.0f, 0.0f};
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const f32x4 *pd = (const f32x4 *)(&s_dhg[rp][g][q * KQ][0]);
#pragma unroll
                for (int j = 0; j < KQ / 4; ++j) {
                    const f32x4 v0 = pd[2 * j], v1 = pd[2 * j + 1], wv = wt[g][j];
                    acc = __builtin_elementwise_fma((f32x2){wv.x, wv.x}, (f32x2){v0.x, v0.y}, acc);
                    acc = __builtin_elementwise_fma((f32x2){wv.y, wv.y}, (f32x2){v0.z, v0.w}, acc);
                    acc = __builtin_elementwise_fma((f32x2){wv.z, wv.z}, (f32x2){v1.x, v1.y}, acc);
                    acc = __builtin_elementwise_fma((f32x2){wv.w, wv.w}, (f32x2){v1.z, v1.w}, acc);
                }
            }
            s_part[q][2 * rp][u] = acc.x;
            s_part[q][2 * rp + 1][u] = acc.y;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int rr = ra + s;
            gh[s] = direct[s] + s_part[0][rr][u] + s_part[1][rr][u] + s_part[2][rr][u] + s_part[3][rr][u];
            cur[s] = nx[s];
        }
        // the next iteration writes s_dhg only after its first barrier's predecessor: all reads of s_dhg above are
        // complete (second barrier), and s_part is rewritten only after the next first barrier
    }
    if (d_h0)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (ra + s < rv) d_h0[(row0 + ra + s) * H + u] = gh[s];
    if (bias_part) {  // this workgroup's share of db_ih (r|z|n) and db_hh (r|z|hn): the 4 row-pair threads of a unit meet in LDS
        __syncthreads();
        s_part[q][0][u] = bs_r; s_part[q][1][u] = bs_z; s_part[q][2][u] = bs_n; s_part[q][3][u] = bs_hn;
        __syncthreads();
        if (q == 0) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (s_part[0][k][u] + s_part[1][k][u]) + (s_part[2][k][u] + s_part[3][k][u]);
            float *bp = bias_part + (size_t)blockIdx.x * 6 * H;
            bp[u] = v[0]; bp[H + u] = v[1]; bp[2 * H + u] = v[2];
            bp[3 * H + u] = v[0]; bp[4 * H + u] = v[1]; bp[5 * H + u] = v[3];
        }
    }
}

// ---- forward on the matrix cores, eval and target network in ONE launch (the packed learn of VDN.learn) ----------------------
// k_gru_seq_fwd gives 8 rows to a workgroup because its VALU dot products need the k-quarters' partial sums exchanged through LDS;
// 2048 sequences then fill the 256 CUs once, and the eval and the target network (two launches) each pay T steps of ~3.6 µs.
// On the matrix cores a workgroup takes 16 rows (the M of v_mfma_f32_16x16x4_f32): wave w owns hidden units 16 w .. 16 w + 15 of all
// three gates (three column tiles: the r / z / n values of a (row, unit) pair land in the same lane), keeps its 3 x 16 x 128 slice of
// W_hh in 96 registers as the B operands, reads h (16 x 128, LDS) as the A operand with eight 16-byte reads per step (K order: a
// lane quarter holds k = 16 jj + 4 kq + c), and finishes its own gates: no partial sums, ONE barrier per step (h double-buffered).
// 2048 sequences are 128 workgroups; the two networks of a learn make 256 -- one launch, every CU busy, ~the time one network took.
constexpr int kMW = 16, kMHS = 132;   // rows per workgroup; LDS row stride of h (= 4 mod 32: the 16 lanes of a read group cover all banks)
struct GruNetIO {
    const float *igates, *h0, *w_hh, *b_ih, *b_hh;   // h0 NULL = zeros
    float *hs, *gates;                               // gates NULL = not saved
};
__global__ __launch_bounds__(kBlock) void k_gru_seq_fwd_mfma(GruNetIO n0, GruNetIO n1, int blocks_per_net, int T, long R, const SeqOff so) {
    __shared__ __attribute__((aligned(16))) float s_h[2][kMW][kMHS];
    const bool second = (int)blockIdx.x >= blocks_per_net;
    const GruNetIO io = second ? n1 : n0;
    const long row0 = (long)((int)blockIdx.x - (second ? blocks_per_net : 0)) * kMW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
    const int u = 16 * wave + j;   // this lane's hidden unit (B column / D column)
    const int rv = (int)min((long)kMW, R - row0);
    f32x4 bw[3][8];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) bw[g][jj] = *(const f32x4 *)(io.w_hh + (size_t)(g * H + u) * H + 16 * jj + 4 * kq);
    const float bir = io.b_ih[u] + io.b_hh[u], biz = io.b_ih[H + u] + io.b_hh[H + u], bin = io.b_ih[2 * H + u], bhn = io.b_hh[2 * H + u];
    for (int i = tid; i < kMW * H; i += kBlock) {
        const int rr = i / H, c = i - rr * H;
        s_h[0][rr][c] = (rr < rv && io.h0) ? io.h0[(row0 + rr) * H + c] : 0.0f;
        s_h[1][rr][c] = 0.0f;   // rows past the end / ended sequences: finite values on the A side, results never stored
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const long rt = so.packed ? (long)(so.off[t + 1] - so.off[t]) : R;
        if (row0 >= rt) break;  // uniform: lengths only shrink along the rows
        const int rvt = (int)min((long)rv, rt - row0);
        const size_t base = (so.packed ? (size_t)so.off[t] : (size_t)t * R) + (size_t)row0;
        const float (*cur)[kMHS] = s_h[t & 1];
        float (*nxt)[kMHS] = s_h[(t + 1) & 1];
        // x-side gates of my four rows (D rows 4 kq + q): issued first, consumed after the MFMAs
        float ig[4][3];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = 4 * kq + q;
#pragma unroll
            for (int g = 0; g < 3; ++g) ig[q][g] = rr < rvt ? io.igates[(base + rr) * 3 * H + g * H + u] : 0.0f;
        }
        f32x4 a4[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) a4[jj] = *(const f32x4 *)(&cur[j][16 * jj + 4 * kq]);
        // two partial sums per gate (k-halves of 64), added at the end: six independent MFMA chains in flight, and the rounding of a
        // 128-term fp32 sum grows with the length of the chain
        f32x4 acc2[2][3];
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc2[hlf][g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf)
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        acc2[hlf][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[4 * hlf + jj][c], bw[g][4 * hlf + jj][c], acc2[hlf][g], 0, 0, 0);
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = acc2[0][g] + acc2[1][g];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = 4 * kq + q;
            if (rr < rvt) {
                const float hn = acc[2][q] + bhn;
                const float rg = sigmoidf_(ig[q][0] + acc[0][q] + bir);
                const float zg = sigmoidf_(ig[q][1] + acc[1][q] + biz);
                const float ng = tanhf(ig[q][2] + bin + rg * hn);
                const float hp = cur[rr][u];
                const float hnew = (1.0f - zg) * ng + zg * hp;
                const size_t o = base + rr;
                io.hs[o * H + u] = hnew;
                if (io.gates) {
                    io.gates[o * 4 * H + u] = rg; io.gates[o * 4 * H + H + u] = zg; io.gates[o * 4 * H + 2 * H + u] = ng;
                    io.gates[o * 4 * H + 3 * H + u] = hn;
                }
                nxt[rr][u] = hnew;
            }
        }
        __syncthreads();
    }
}

thread_local int g_last = 0;

}  // namespace

extern "C" {

static int make_seq_off(const int32_t *step_rows, int T, int64_t R, SeqOff &so) {
    so.packed = step_rows ? 1 : 0;
    if (!step_rows) return CRNN_OK;
    if (T > GRU_SEQ_MAX_STEPS) return CRNN_ERR_UNSUPPORTED;
    long off = 0, prev = R;
    for (int t = 0; t < T; ++t) {
        if (step_rows[t] < 0 || step_rows[t] > prev) return CRNN_ERR_BAD_ARG;  // counts must not grow, none above R
        so.off[t] = (int32_t)off;
        off += step_rows[t];
        prev = step_rows[t];
        if (off > 0x7fffffffL) return CRNN_ERR_BAD_ARG;
    }
    so.off[T] = (int32_t)off;
    return CRNN_OK;
}

static int seq_forward(const float *d_igates, const float *d_h0, const float *d_w_hh, const float *d_b_ih, const float *d_b_hh,
                       int T, int64_t R, int hidden, const int32_t *step_rows, float *d_hs, float *d_gates, void *stream) {
    if (!d_igates || !d_h0 || !d_w_hh || !d_b_ih || !d_b_hh || !d_hs || T < 0 || R < 0) return CRNN_ERR_BAD_ARG;
    if (hidden != H) return CRNN_ERR_UNSUPPORTED;
    if (T == 0 || R == 0) return CRNN_OK;
    SeqOff so;
    const int rc = make_seq_off(step_rows, T, R, so);
    if (rc != CRNN_OK) return rc;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_gru_seq_fwd, dim3((unsigned)((R + RW - 1) / RW)), dim3(kBlock), 0, (hipStream_t)stream, d_igates, d_h0, d_w_hh,
                       d_b_ih, d_b_hh, T, (long)R, d_hs, d_gates, so);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last = (int)e; return CRNN_ERR_HIP; }
    return CRNN_OK;
}

int gru_seq_forward(const float *d_igates, const float *d_h0, const float *d_w_hh, const float *d_b_ih, const float *d_b_hh,
                    int T, int64_t R, int hidden, float *d_hs, float *d_gates, void *stream) {
    return seq_forward(d_igates, d_h0, d_w_hh, d_b_ih, d_b_hh, T, R, hidden, nullptr, d_hs, d_gates, stream);
}

int gru_seq_forward_packed(const float *d_igates, const float *d_h0, const float *d_w_hh, const float *d_b_ih, const float *d_b_hh,
                           int T, int64_t R, int hidden, const int32_t *step_rows, float *d_hs, float *d_gates, void *stream) {
    if (!step_rows) return CRNN_ERR_BAD_ARG;
    return seq_forward(d_igates, d_h0, d_w_hh, d_b_ih, d_b_hh, T, R, hidden, step_rows, d_hs, d_gates, stream);
}

int gru_seq_forward_packed_pair(const float *d_igates_a, const float *d_h0_a, const float *d_w_hh_a, const float *d_b_ih_a,
                                const float *d_b_hh_a, float *d_hs_a, float *d_gates_a, const float *d_igates_b, const float *d_h0_b,
                                const float *d_w_hh_b, const float *d_b_ih_b, const float *d_b_hh_b, float *d_hs_b, float *d_gates_b,
                                int T, int64_t R, int hidden, const int32_t *step_rows, void *stream) {
    if (!d_igates_a || !d_w_hh_a || !d_b_ih_a || !d_b_hh_a || !d_hs_a || T < 0 || R < 0 || !step_rows) return CRNN_ERR_BAD_ARG;
    const bool two = d_igates_b != nullptr;
    if (two && (!d_w_hh_b || !d_b_ih_b || !d_b_hh_b || !d_hs_b)) return CRNN_ERR_BAD_ARG;
    if ((((size_t)d_w_hh_a | (size_t)d_w_hh_b) & 15) != 0) return CRNN_ERR_BAD_ARG;
    if (hidden != H) return CRNN_ERR_UNSUPPORTED;
    if (T == 0 || R == 0) return CRNN_OK;
    SeqOff so;
    const int rc = make_seq_off(step_rows, T, R, so);
    if (rc != CRNN_OK) return rc;
    const long per_net = (R + kMW - 1) / kMW;
    if (per_net * 2 > 0x7fffffffL) return CRNN_ERR_BAD_ARG;
    const GruNetIO a{d_igates_a, d_h0_a, d_w_hh_a, d_b_ih_a, d_b_hh_a, d_hs_a, d_gates_a};
    const GruNetIO b{d_igates_b, d_h0_b, d_w_hh_b, d_b_ih_b, d_b_hh_b, d_hs_b, d_gates_b};
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_gru_seq_fwd_mfma, dim3((unsigned)(per_net * (two ? 2 : 1))), dim3(kBlock), 0, (hipStream_t)stream, a, two ? b : a,
                       (int)per_net, T, (long)R, so);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last = (int)e; return CRNN_ERR_HIP; }
    return CRNN_OK;
}

static int seq_backward(const float *d_grad_hs, const float *d_gates, const float *d_hs, const float *d_h0, const float *d_w_hh,
                        int T, int64_t R, int hidden, const int32_t *step_rows, float *d_d_igates, float *d_d_hgates, float *d_d_h0,
                        float *d_bias_part, void *stream, float *d_h_prev = nullptr) {
    if (!d_grad_hs || !d_gates || !d_hs || !d_h0 || !d_w_hh || !d_d_igates || !d_d_hgates || T < 0 || R < 0) return CRNN_ERR_BAD_ARG;
    if (hidden != H) return CRNN_ERR_UNSUPPORTED;
    if (T == 0 || R == 0) return CRNN_OK;
    SeqOff so;
    const int rc = make_seq_off(step_rows, T, R, so);
    if (rc != CRNN_OK) return rc;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_gru_seq_bwd, dim3((unsigned)((R + RW - 1) / RW)), dim3(kBlock), 0, (hipStream_t)stream, d_grad_hs, d_gates, d_hs,
                       d_h0, d_w_hh, T, (long)R, d_d_igates, d_d_hgates, d_d_h0, d_bias_part, so, d_h_prev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last = (int)e; return CRNN_ERR_HIP; }
    return CRNN_OK;
}

int gru_seq_backward(const float *d_grad_hs, const float *d_gates, const float *d_hs, const float *d_h0, const float *d_w_hh,
                     int T, int64_t R, int hidden, float *d_d_igates, float *d_d_hgates, float *d_d_h0, float *d_bias_part,
                     void *stream) {
    return seq_backward(d_grad_hs, d_gates, d_hs, d_h0, d_w_hh, T, R, hidden, nullptr, d_d_igates, d_d_hgates, d_d_h0, d_bias_part, stream);
}

int gru_seq_backward_packed(const float *d_grad_hs, const float *d_gates, const float *d_hs, const float *d_h0, const float *d_w_hh,
                            int T, int64_t R, int hidden, const int32_t *step_rows, float *d_d_igates, float *d_d_hgates,
                            float *d_d_h0, float *d_bias_part, float *d_h_prev, void *stream) {
    if (!step_rows) return CRNN_ERR_BAD_ARG;
    return seq_backward(d_grad_hs, d_gates, d_hs, d_h0, d_w_hh, T, R, hidden, step_rows, d_d_igates, d_d_hgates, d_d_h0, d_bias_part, stream,
                        d_h_prev);
}

int64_t gru_seq_row_blocks(int64_t R) { return (R + RW - 1) / RW; }

int gru_last_hip_error(void) { return g_last; }

}  // extern "C"
