// crnn_mfma.h -- the same conv1+ReLU+conv2+ReLU front end as k_conv9 (crnn_ops.hip), on the gfx950 matrix cores
// with f32 operands: v_mfma_f32_16x16x4_f32 is an exact f32 fma chain at 64 FLOP/clk/SIMD, twice the rate of the
// plain (non-packed) v_fma_f32 stream the VALU kernel issues.  Both convolutions are GEMMs whose M dimension is
// (row, output position) flattened, N the output channel (two 16-wide halves; od 24 leaves 8 columns of the second
// half idle) and K the (input channel, tap) pairs:
//   conv1: M = RB*49, K = 27 (+1 zero), A gathered from the float image of the int8 pixel rows in LDS
//   conv2: M = RB*25, K = od*9,          A gathered from the conv1 activations in LDS
// The A operand of the 16x16x4 form is ONE float per lane (lane l: A[i = l & 15][k = l >> 4]) so the im2col gather
// is a single ds_read_b32 per MFMA with a compile-time offset: K is ordered (channel quad, tap) with the lane's
// k = l >> 4 selecting the channel inside the quad, which folds into the per-lane base address.  The B operands
// (the weights of the lane's output channel) stay in registers for the whole kernel: 7 + od*9/4 VGPRs.
// 8 waves per workgroup, two per SIMD; SIMDs 0,1 own channel half 0, SIMDs 2,3 half 1; each wave walks every fourth
// 16-position tile of its half with two tiles (two independent accumulators) in flight.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace crnn_mfma {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlockM = 512;

// tools/probe/conv_probe.hip -DCRNN_PROBE_TS: lane 0 of every wave stamps the shader-cycle counter at the phase boundaries of
// its SECOND row block into g_crnn_ts[workgroup][wave][stamp]; compiled out of the product.
#ifdef CRNN_PROBE_TS
__device__ unsigned long long *g_crnn_ts;
#define CRNN_TS(k) do { if (lane == 0 && ts_it == 1 && g_crnn_ts) g_crnn_ts[((size_t)blockIdx.x * 8 + wave) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define CRNN_TS(k) do { } while (0)
#endif

// RBV: rows per workgroup iteration.  0 = the largest block one workgroup per CU can hold (16 rows at od 24, 12 at od 32:
// 139 / 134 KB of LDS).  Two workgroups per CU (8 rows each) were built twice -- eight waves at 128 registers, and four waves
// owning both channel halves -- and both measured SLOWER (register spills; tools/probe/conv9_mfma_w4.patch).
template <int OD, int RBV = 0, int NW = 8> struct GeoM {
    static constexpr int RB = RBV ? RBV : (OD <= 24 ? 16 : 12);  // rows per iteration (LDS-bound)
    static constexpr int CS = 53;                        // conv1 activation stride per channel: odd, so the epilogue's
                                                         // 16 channel lanes fall on different banks
    // ROWLANE (a block of exactly 16 rows: od 24): the 16 M entries of an MFMA tile are the 16 ROWS of the block at ONE output
    // position, instead of 16 consecutive (row, position) pairs.  Lane (j, kq) of the A operand then reads
    // row j * ROW_STRIDE + kq * (odd channel / tap offset) + a wave-uniform position offset: with the row strides = 2 (mod 32
    // words) the 32 lanes of a ds_read_b32 group (j = 0..15, two kq) hit 32 different banks.  With 16 consecutive positions of one
    // row as the tile (the other geometry) the two kq halves overlap on 6 of 16 banks: 4 LDS cycles per gather instead of 2
    // (SQ_LDS_BANK_CONFLICT = 44 % of the LDS cycles, profiles/r03/pmc_conv_summary.json).
#ifdef CRNN_NO_ROWLANE   // same-box A/B builds only (tools/ab_conv.sh)
    static constexpr bool ROWLANE = false;
#else
    static constexpr bool ROWLANE = RB == 16;
#endif
    static constexpr int ROW_A1 = ROWLANE ? (OD * CS + 31) / 32 * 32 + 2 : OD * CS;
    static constexpr int IN_STRIDE = ROWLANE ? 258 : 244;
    static constexpr int PAD_COLS = (OD * 25 + 10 + 63) / 64 * 64;  // 640 / 832: a row may be written out zero-padded to a
                                                         // multiple of 64 floats (the GRU input GEMM runs 15-25 % faster on K = 640 than on 610)
    static constexpr int OUT_STRIDE = PAD_COLS + 4;      // staged output row: conv features | 10 vector features | zeros
    static constexpr int KQ = OD / 4;                    // channel quads
    static constexpr int NSTEP2 = KQ * 9;                // conv2 k-steps (54 / 72)
    static constexpr int M1 = RB * 49, M2 = RB * 25;
    static constexpr int T2 = (M2 + 15) / 16;
    static constexpr int NSUB = NW / 2;                  // waves per channel half
    static constexpr int BLOCK = 64 * NW;                // NW = 8: one workgroup per CU, two waves per SIMD; NW = 4 (RB 8): two
                                                         // workgroups per CU, one wave each per SIMD, out of phase by themselves
    static_assert(NW == 8 || NW == 4, "wave roles");
    static_assert(RB % NSUB == 0 && RB <= 16, "conv1 tiling: rows split over NSUB waves per half, one position-48 tile");
    static constexpr int VEC = 18;                       // dir_x, dir_y, one-hot (<= 16) per row
    static constexpr int MLP = 10 * VEC + 10;            // the vector branch's weights [10][nin] and biases
    static constexpr size_t LDS_FLOATS = (size_t)RB * IN_STRIDE + (size_t)RB * ROW_A1 + (size_t)RB * OUT_STRIDE + (size_t)RB * VEC + MLP;
};

// conv2 for NT (1 or 2) tiles of 16 output positions: gathers of channel quad cq + 1 are issued before the MFMAs of
// quad cq (sched_barrier keeps that order), so an LDS read has a whole quad of MFMA issue time to land.
template <int OD, int RBV, int NT>
__device__ __forceinline__ void conv2_tiles(const float *s_a1, float *s_out, const float (&bw2)[GeoM<OD, RBV>::NSTEP2], float bias2,
                                            int t0, int t1, int j, int kq, int ch, bool chv) {
    using G = GeoM<OD, RBV>;
    const float *ap[NT];
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        int r, p;
        if constexpr (G::ROWLANE) {   // tile = output position, lane j = row j
            r = j;
            p = n == 0 ? t0 : t1;
        } else {
            int m = (n == 0 ? t0 : t1) * 16 + j;
            m = m < G::M2 ? m : G::M2 - 1;
            r = m / 25;
            p = m - r * 25;
        }
        ap[n] = s_a1 + r * G::ROW_A1 + (p / 5) * 7 + p % 5 + kq * G::CS;
        acc[n] = f32x4{bias2, bias2, bias2, bias2};
    }
    float v[2][NT][9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int n = 0; n < NT; ++n) v[0][n][tap] = ap[n][(tap / 3) * 7 + tap % 3];
#ifdef CRNN_PROBE_NO_GATHER   // timing only: the gathers of channel quads 1.. are replaced by a register
#define CRNN_GATHER(x) (bias2)
#else
#define CRNN_GATHER(x) (x)
#endif
#pragma unroll
    for (int cq = 0; cq < G::KQ; ++cq) {
        if (cq + 1 < G::KQ) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    v[(cq + 1) & 1][n][tap] = CRNN_GATHER(ap[n][(cq + 1) * 4 * G::CS + (tap / 3) * 7 + tap % 3]);  // compile-time offset:imm
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[cq & 1][n][tap], bw2[cq * 9 + tap], acc[n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef CRNN_PROBE_NO_EPI      // timing only: no tile epilogue
    if (chv && acc[0][0] == 12345.678f) {
#else
    if (chv) {
#endif
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (G::ROWLANE) {   // D row 4 kq + q = block row, the tile's position is the column inside the channel
                    s_out[(kq * 4 + q) * G::OUT_STRIDE + ch * 25 + (n == 0 ? t0 : t1)] = fmaxf(acc[n][q], 0.0f);
                } else {
                    const int mm = (n == 0 ? t0 : t1) * 16 + kq * 4 + q;
                    if (mm < G::M2) { const int rr = mm / 25, pp = mm - rr * 25; s_out[rr * G::OUT_STRIDE + ch * 25 + pp] = fmaxf(acc[n][q], 0.0f); }
                }
            }
    }
}

template <int OD, int RBV = 0, int NW = 8>
__global__ __launch_bounds__(64 * NW) void k_conv9_mfma(const int8_t *__restrict__ obs, long obs_stride, long rows,
                                                        const float *__restrict__ w1, const float *__restrict__ b1,
                                                        const float *__restrict__ w2, const float *__restrict__ b2,
                                                        float *__restrict__ out, long out_stride, int out_cols,
                                                        const int8_t *__restrict__ onehot, int n_actions,
                                                        const float *__restrict__ mlp_w, const float *__restrict__ mlp_b,
                                                        const int32_t *__restrict__ live_chips, const int32_t *__restrict__ n_live,
                                                        int rows_per_chip) {
    // live_chips != NULL (rollout with finished chips): only the rows of the *n_live chips listed in live_chips (rows_per_chip
    // consecutive rows each) are processed; input row = live_chips[k] * rows_per_chip + a, OUTPUT row = k * rows_per_chip + a
    // (compact).  `rows` is then the worst case the grid was sized for; the device-side count decides.
    using G = GeoM<OD, RBV, NW>;
    constexpr int kBlockM = G::BLOCK;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *s_in = lds;                              // [RB][244]  float image of the pixel bytes
    float *s_a1 = s_in + G::RB * G::IN_STRIDE;      // [RB][OD][53] conv1 activations
    float *s_out = s_a1 + G::RB * G::ROW_A1;        // [RB][OUT_STRIDE]
    float *s_vec = s_out + G::RB * G::OUT_STRIDE;   // [RB][18] inputs of the vector branch
    float *s_mlp = s_vec + G::RB * G::VEC;          // [10][nin] weights, then [10] biases of the vector branch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // wave w runs on SIMD w & 3: SIMDs 0,1 hold channel half 0, SIMDs 2,3 half 1; the two waves of a SIMD take
    // tiles sub, sub + 4, ... with sub = (w & 1) and (w & 1) + 2, so every SIMD gets 12 or 13 of the 25 conv2 tiles
    const int nh = (wave >> 1) & 1, sub = (wave & 1) + 2 * (wave >> 2);   // NW = 4: sub = wave & 1
    const int j = lane & 15, kq = lane >> 4;
    const int ch = nh * 16 + j;                     // the output channel this lane's B column / D column belongs to
    const bool chv = ch < OD;

    if (live_chips) rows = min(rows, (long)n_live[0] * rows_per_chip);
    const long n_blocks = (rows + G::RB - 1) / G::RB;
    const int nin = 2 + n_actions;
    // the vector branch's parameters live in LDS: read from global memory at the head of every row block, the wait for them
    // was also a wait for the previous block's output stores (one in-order memory counter on gfx9)
    if (mlp_w) {
        for (int i = tid; i < 10 * nin; i += kBlockM) s_mlp[i] = mlp_w[i];
        if (tid < 10) s_mlp[10 * G::VEC + tid] = mlp_b[tid];
    }
    // cr / rows_per_chip as a multiply-high (exact for cr * rows_per_chip < 2^32: the host checks the row count)
    // one row per chip (drop_num = 1): 2^32 / 1 does not fit the 32-bit magic -- the quotient is cr itself
    const uint32_t rpc = (uint32_t)(rows_per_chip > 0 ? rows_per_chip : 1);
    const uint32_t rpc_magic = rpc > 1 ? (uint32_t)(0x100000000ull / rpc) + 1u : 0u;
    auto src_row = [&](long cr) -> long {  // compact row -> row of the input tensors
        if (!live_chips) return cr;
        const uint32_t k = rpc > 1 ? __umulhi((uint32_t)cr, rpc_magic) : (uint32_t)cr;
        return (long)((uint32_t)live_chips[k] * rpc + ((uint32_t)cr - k * rpc));
    };
    // The bytes of block i+1 are fetched into registers while block i is in conv1 and parked in LDS once conv1 is
    // done with s_in: the HBM latency of the int8 rows never sits between two barriers.  A WAVE fetches whole rows (rows
    // wave, wave + 8 of the block): the row pointer is wave-uniform (scalar registers, the live list is read with scalar
    // loads), the lanes add lane + 64 v.  Per-thread element indices (row = i / 243 per lane) made the compiler hoist eight
    // 64-bit row offsets per lane out of the loop, spill them, and wait on every reload -- which, the memory counter being
    // in-order, also waited on the previous row byte: the eight loads of a block ran one after the other.
    constexpr int RPW = (G::RB + NW - 1) / NW, ROWB = 245;  // rows per wave; bytes of a row: 243 pixels, dir_x, dir_y
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int pf[RPW][4], pfo[RPW];
    int pf_rv = 0;
#pragma unroll
    for (int h = 0; h < RPW; ++h) {
        pfo[h] = 0;
#pragma unroll
        for (int v = 0; v < 4; ++v) pf[h][v] = 0;
    }
    const int p3 = min(lane + 192, mlp_w ? ROWB - 1 : 242);  // the two direction bytes exist only with the vector branch
    auto fetch = [&](long b) {
        const long r0 = b * G::RB;
        pf_rv = b < n_blocks ? (int)min((long)G::RB, rows - r0) : 0;
#pragma unroll
        for (int h = 0; h < RPW; ++h) {
            const int rr = wave_u + NW * h;
            if (rr < pf_rv) {  // wave-uniform
                const long sr = src_row(r0 + rr);
                const int8_t *row = obs + sr * obs_stride;
                pf[h][0] = row[lane];
                pf[h][1] = row[lane + 64];
                pf[h][2] = row[lane + 128];
                pf[h][3] = row[p3];
                if (mlp_w && onehot && n_actions > 0) pfo[h] = onehot[sr * n_actions + min(lane, n_actions - 1)];
            }
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int h = 0; h < RPW; ++h) {
            const int rr = wave_u + NW * h;
            if (rr < G::RB) {
                const bool on = rr < pf_rv;   // rows past the end: finite zeros
                float *dst = s_in + rr * G::IN_STRIDE;
#pragma unroll
                for (int v = 0; v < 3; ++v) dst[lane + 64 * v] = on ? (float)pf[h][v] : 0.0f;
                const float last = on ? (float)pf[h][3] : 0.0f;
                if (lane + 192 < 243) dst[lane + 192] = last;
                else if (lane + 192 < ROWB) s_vec[rr * G::VEC + lane + 192 - 243] = last;
                if (lane < 16) s_vec[rr * G::VEC + 2 + lane] = (on && onehot && lane < n_actions) ? (float)pfo[h] : 0.0f;
            }
        }
    };
    fetch(blockIdx.x);  // in flight while the weights are staged; parked below

    // ---- B operands (weights of channel ch) and the lane's conv1 gather offsets.  The weights go through LDS: read straight
    // from global memory every lane of a wave hits a different cache line (channel stride 9 od floats), 61 such loads x 8 waves
    // through the CU's one address unit cost ~15 us per launch; staged with coalesced loads the prologue is ~1 us.
    for (int i = tid; i < OD * OD * 9; i += kBlockM) s_a1[i] = w2[i];   // [c_out][c_in][tap], s_a1 is free until the first conv1
    for (int i = tid; i < OD * 27; i += kBlockM) s_in[i] = w1[i];       // [c_out][27]
    __syncthreads();
    float bw1[7];
    int off1[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int k = 4 * s + kq;
        const bool kv = k < 27;
        bw1[s] = (chv && kv) ? s_in[ch * 27 + k] : 0.0f;
        const int c0 = k / 9, tap = k - c0 * 9;
        off1[s] = kv ? c0 * 81 + (tap / 3) * 9 + tap % 3 : (G::ROWLANE ? 1 : 0);   // (k = 27: zero weight; an odd offset keeps its bank apart from k = 26's)
    }
    int goff[3];
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) { const int p = qt * 16 + j; goff[qt] = (p / 7) * 9 + p % 7; }
    float bw2[G::NSTEP2];
#pragma unroll
    for (int cq = 0; cq < G::KQ; ++cq)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) bw2[cq * 9 + tap] = chv ? s_a1[(ch * OD + 4 * cq + kq) * 9 + tap] : 0.0f;
    __syncthreads();  // the staging areas are reused below (s_in by park(), s_a1 by conv1)
    const float bias1 = chv ? b1[ch] : 0.0f, bias2 = chv ? b2[ch] : 0.0f;
    const int n_feat = OD * 25 + (mlp_w ? 10 : 0);
    const int n_out = out_cols > n_feat ? out_cols : n_feat;  // columns n_feat .. n_out-1 of a row are written as zeros
    const bool wide_out = (out_stride % 2 == 0) && (((size_t)out) % 8 == 0) && (n_out % 2 == 0);
    const bool quad_out = (out_stride % 4 == 0) && (((size_t)out) % 16 == 0) && (n_out % 4 == 0) && (G::OUT_STRIDE % 4 == 0);
    for (int i = tid; i < G::RB * (G::OUT_STRIDE - n_feat); i += kBlockM) {  // the zero tail of every staged row, once
        const int rr = i / (G::OUT_STRIDE - n_feat), k = i - rr * (G::OUT_STRIDE - n_feat);
        s_out[rr * G::OUT_STRIDE + n_feat + k] = 0.0f;
    }

    // Two barriers per row block.  B2 (after conv1) frees s_in / s_vec: the NEXT block's rows, fetched into registers before conv1,
    // are parked right behind it, while conv2 runs.  B3 (after conv2) publishes the staged rows and the parked inputs.  There
    // is NO barrier between a block's stream-out and the next block's conv1: conv1 reads s_in and writes s_a1, the stream-out
    // reads s_out -- a wave that has streamed its rows out starts on the next conv1 while slower waves still store, and s_out is
    // rewritten only behind the next B2, which every wave reaches after its stream-out.
    park();
    __syncthreads();
#ifdef CRNN_PROBE_TS
    int ts_it = 0;
#endif
    for (long blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        CRNN_TS(0);
        const long row0 = blk * G::RB;
        const int rv = (int)min((long)G::RB, rows - row0);
        float mv = 0.0f;  // vector branch: relu(mlp1([dir_x, dir_y, last-action one-hot])) (base_net.py:66); s_vec is re-parked behind B2
        const int mr = tid / 10, mc = tid - mr * 10;
        if (mlp_w && tid < G::RB * 10) {
            mv = s_mlp[10 * G::VEC + mc];
            for (int k = 0; k < nin; ++k) mv = fmaf(s_vec[mr * G::VEC + k], s_mlp[mc * nin + k], mv);
        }
        fetch(blk + gridDim.x);
        CRNN_TS(1);
        // ---- conv1
#ifndef CRNN_PROBE_SKIP_CONV1
        if constexpr (G::ROWLANE) {
            // Tile = ONE of the 49 conv1 output positions for the 16 rows of the block (lane j = row j); wave `sub` of a channel half
            // takes positions sub, sub + 4, ...: 13 or 12 tiles, three accumulator chains at a time, the gathers of the next three
            // in flight while the MFMAs of the current three issue.  Address = lane part (row, k) + wave-uniform position offset.
            constexpr int NT1 = 49, NIT = (NT1 / G::NSUB) / 3;   // 12 tiles per wave in 4 passes of 3; position 48 is wave 0's 13th
            static_assert(G::NSUB == 4 && NIT * 3 * G::NSUB == NT1 - 1, "conv1 row-lane tiling");
            const int sub_u = __builtin_amdgcn_readfirstlane(sub);
            int jb = j * G::IN_STRIDE;
            asm volatile("" : "+v"(jb));   // formed per block: hoisted out of the block loop the seven addresses stay live through conv2
            int lb[7];
#pragma unroll
            for (int s = 0; s < 7; ++s) lb[s] = jb + off1[s];
            auto spos = [&](int m) {  // offset of position sub + 4 m in the 9x9 image
                const int p = sub_u + G::NSUB * m;
                return (p / 7) * 9 + p % 7;
            };
            float cv[2][3][7];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int sp = spos(c);
#pragma unroll
                for (int s = 0; s < 7; ++s) cv[0][c][s] = s_in[lb[s] + sp];
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (it + 1 < NIT) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const int sp = spos(3 * (it + 1) + c);
#pragma unroll
                        for (int s = 0; s < 7; ++s) cv[(it + 1) & 1][c][s] = s_in[lb[s] + sp];
                    }
                } else if (sub_u == 0) {   // the gathers of position 48 (image offset 60), wave 0 of each channel half
#pragma unroll
                    for (int s = 0; s < 7; ++s) cv[(it + 1) & 1][0][s] = s_in[lb[s] + 60];
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4 acc[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = f32x4{bias1, bias1, bias1, bias1};
#pragma unroll
                for (int s = 0; s < 7; ++s)
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[it & 1][c][s], bw1[s], acc[c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (chv) {   // D row 4 kq + q = block row
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float *dst = s_a1 + kq * 4 * G::ROW_A1 + ch * G::CS + sub_u + G::NSUB * (3 * it + c);
#pragma unroll
                        for (int q = 0; q < 4; ++q) dst[q * G::ROW_A1] = fmaxf(acc[c][q], 0.0f);
                    }
                }
            }
            if (sub_u == 0) {
                f32x4 acc = {bias1, bias1, bias1, bias1};
#pragma unroll
                for (int s = 0; s < 7; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[NIT & 1][0][s], bw1[s], acc, 0, 0, 0);
                if (chv) {
                    float *dst = s_a1 + kq * 4 * G::ROW_A1 + ch * G::CS + 48;
#pragma unroll
                    for (int q = 0; q < 4; ++q) dst[q * G::ROW_A1] = fmaxf(acc[q], 0.0f);
                }
            }
        } else {
            // Tiles follow the rows so that no index needs a division: a row's positions 0..47 are three tiles
            // (quarter qt: p = 16 qt + i), position 48 of all RB rows is one more tile.  Wave `sub` takes rows
            // sub, sub + 4, ...: three accumulator chains per row; the gathers of the next row are in flight while
            // the MFMAs of the current one issue.  Addresses are per-lane constants + compile-time offsets.
            // The 21 gather addresses goff[qt] + off1[s] are formed HERE, per row block (the empty asm hides goff from the
            // loop-invariant code motion): hoisted out of the block loop they stayed live through conv2 and the stream-out
            // as well, and the kernel spilled.
            int gq[3];
#pragma unroll
            for (int qt = 0; qt < 3; ++qt) { gq[qt] = goff[qt] + sub * G::IN_STRIDE; asm volatile("" : "+v"(gq[qt])); }
            float cv[2][3][7];
#pragma unroll
            for (int qt = 0; qt < 3; ++qt)
#pragma unroll
                for (int s = 0; s < 7; ++s) cv[0][qt][s] = s_in[gq[qt] + off1[s]];
#pragma unroll
            for (int i = 0; i < G::RB / G::NSUB; ++i) {
                if (i + 1 < G::RB / G::NSUB) {
#pragma unroll
                    for (int qt = 0; qt < 3; ++qt)
#pragma unroll
                        for (int s = 0; s < 7; ++s)
                            cv[(i + 1) & 1][qt][s] = s_in[G::NSUB * (i + 1) * G::IN_STRIDE + gq[qt] + off1[s]];
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4 acc[3];
#pragma unroll
                for (int qt = 0; qt < 3; ++qt) acc[qt] = f32x4{bias1, bias1, bias1, bias1};
#pragma unroll
                for (int s = 0; s < 7; ++s)
#pragma unroll
                    for (int qt = 0; qt < 3; ++qt) acc[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[i & 1][qt][s], bw1[s], acc[qt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (chv) {
                    float *dst = s_a1 + (sub + G::NSUB * i) * G::ROW_A1 + ch * G::CS + kq * 4;
#pragma unroll
                    for (int qt = 0; qt < 3; ++qt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) dst[qt * 16 + q] = fmaxf(acc[qt][q], 0.0f);
                }
            }
            if (sub == 0) {  // position 48 (x = y = 6) of every row: lane i gathers row i
                const int rr = j < G::RB ? j : G::RB - 1;
                f32x4 acc = {bias1, bias1, bias1, bias1};
#pragma unroll
                for (int s = 0; s < 7; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(s_in[rr * G::IN_STRIDE + 60 + off1[s]], bw1[s], acc, 0, 0, 0);
                if (chv) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (kq * 4 + q < G::RB) s_a1[(kq * 4 + q) * G::ROW_A1 + ch * G::CS + 48] = fmaxf(acc[q], 0.0f);
                }
            }
        }
#endif
        CRNN_TS(2);
        __syncthreads();  // B2
        CRNN_TS(3);
        park();           // s_in / s_vec were last read before B2
        CRNN_TS(4);
        // ---- conv2
#ifndef CRNN_PROBE_SKIP_CONV2
        {
            int t = sub;
#ifdef CRNN_PROBE_TS
            int ts_k = 8;
#endif
#pragma unroll 1
            for (; t + G::NSUB < G::T2; t += 2 * G::NSUB) {
                conv2_tiles<OD, RBV, 2>(s_a1, s_out, bw2, bias2, t, t + G::NSUB, j, kq, ch, chv);
#ifdef CRNN_PROBE_TS
                CRNN_TS(ts_k); ++ts_k;
#endif
            }
            if (t < G::T2) conv2_tiles<OD, RBV, 1>(s_a1, s_out, bw2, bias2, t, t, j, kq, ch, chv);
        }
#endif
        if (mlp_w && tid < G::RB * 10) s_out[mr * G::OUT_STRIDE + OD * 25 + mc] = fmaxf(mv, 0.0f);
        CRNN_TS(5);
        __syncthreads();  // B3
        CRNN_TS(6);
        // ---- stream the staged rows out: a wave per row, consecutive lanes on consecutive floats
#ifndef CRNN_PROBE_SKIP_OUT
        if (quad_out) {  // 16-byte stores (the padded 640- / 832-column rows of the rollouts and of VDN.learn)
            for (int rr = wave; rr < rv; rr += kBlockM / 64) {
                float4 *dst = (float4 *)(out + (row0 + rr) * out_stride);
                const float4 *src = (const float4 *)(s_out + rr * G::OUT_STRIDE);
                for (int k = lane; k < n_out / 4; k += 64) dst[k] = src[k];
            }
        } else if (wide_out) {  // 8-byte stores: n_out, OUT_STRIDE and (checked once) out / out_stride are even
            for (int rr = wave; rr < rv; rr += kBlockM / 64) {
                float2 *dst = (float2 *)(out + (row0 + rr) * out_stride);
                const float2 *src = (const float2 *)(s_out + rr * G::OUT_STRIDE);
                for (int k = lane; k < n_out / 2; k += 64) dst[k] = src[k];
            }
        } else {
            for (int rr = wave; rr < rv; rr += kBlockM / 64) {
                float *dst = out + (row0 + rr) * out_stride;
                const float *src = s_out + rr * G::OUT_STRIDE;
                for (int k = lane; k < n_out; k += 64) dst[k] = src[k];
            }
        }
#endif
        CRNN_TS(7);
#ifdef CRNN_PROBE_TS
        ++ts_it;
#endif
    }
}


}  // namespace crnn_mfma
