// crnn_mfma19.h -- the conv front end of the reference's CRNN for fov 19 (network/base_net.py:23-33: conv_str(19) =
// [Conv2d(3, od, 3, stride 2), conv3, conv3] with the LAST TWO entries the same module, i.e. tied weights), each followed
// by ReLU (base_net.py:63-65), on the gfx950 matrix cores with f32 operands (v_mfma_f32_16x16x4_f32: an exact f32 fma
// chain).  Same scheme as crnn_mfma.h (fov 9): every convolution is a GEMM whose M dimension is (row, output position)
// flattened, N the output channel (two 16-wide halves) and K the (input channel, tap) pairs; the A operand is ONE float
// per lane gathered from LDS with a compile-time offset, the B operands (the weights of the lane's output channel) stay
// in registers for the whole kernel -- the tied conv3 weights are loaded ONCE and serve both of its applications.
//   stage 1: 3x19x19 int8 image, stride 2 -> od x 9x9     M = RB*81, K = 27 (+1 zero)
//   stage 2: od x 9x9 -> od x 7x7 (conv3)                  M = RB*49, K = od*9
//   stage 3: od x 7x7 -> od x 5x5 (conv3 again)            M = RB*25, K = od*9
// The staged output rows overlay the stage-1 activations, which are dead once stage 2 is done.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace crnn_mfma19 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlockM = 512;
constexpr int kFov = 19, kPix = 3 * kFov * kFov;  // 1083 pixel bytes per row, then dir_x, dir_y

template <int OD> struct Geo {
    static constexpr int RB = 8;                         // rows per iteration (LDS-bound: 150 KB at od 32)
    static constexpr int IMG = 1088;                     // staged pixel bytes per row (1083 padded to a word multiple)
    static constexpr int CS1 = 85;                       // stage-1 activation stride per channel (81 used): odd, and 85 mod 64 = 21
                                                         // spreads the epilogue's 16 channel lanes over 16 different banks
    static constexpr int CS2 = 53;                       // stage-2 activation stride per channel (49 used), as crnn_mfma.h
    static constexpr int ROW_A1 = OD * CS1, ROW_A2 = OD * CS2;
    static constexpr int PAD_COLS = (OD * 25 + 10 + 63) / 64 * 64;  // a row may be written out zero-padded (see crnn_mfma.h)
    static constexpr int OUT_STRIDE = PAD_COLS + 4;      // staged output row: conv features | 10 vector features | zeros
    static constexpr int KQ = OD / 4;                    // channel quads
    static constexpr int NSTEP = KQ * 9;                 // conv3 k-steps
    static constexpr int M2 = RB * 49, M3 = RB * 25;
    static constexpr int T2 = (M2 + 15) / 16, T3 = (M3 + 15) / 16;
    static constexpr int VEC = 18;                       // dir_x, dir_y, one-hot (<= 16) per row
    static constexpr int RPW = (RB + 7) / 8;             // rows a wave fetches per block
    static constexpr int ROWB = kPix + 2, NLD = (ROWB + 63) / 64;     // bytes of a row, byte loads per lane and row
    static constexpr int MLP = 10 * VEC + 10;            // the vector branch's weights [10][nin] and biases
    static_assert(RB % 4 == 0 && RB <= 16, "stage-1 tiling: rows split over 4 waves per channel half, one position-80 tile");
    static_assert(OUT_STRIDE <= ROW_A1, "the staged output rows overlay the stage-1 activations");
    static constexpr size_t LDS_FLOATS = (size_t)RB * IMG / 4 + (size_t)RB * ROW_A1 + (size_t)RB * ROW_A2 + (size_t)RB * VEC + MLP;
};

// conv3 (3x3, stride 1) for NT (1 or 2) tiles of 16 output positions, input planes IW x IW (channel stride CS_IN, row
// stride ROW_IN floats), output planes OW x OW written as s_o[row * row_o + channel * cs_o + position] after ReLU.
// The gathers of channel quad cq + 1 are issued before the MFMAs of quad cq (sched_barrier keeps that order).
template <int OD, int NT, int IW, int OW, int CS_IN, int ROW_IN, int M>
__device__ __forceinline__ void conv3_tiles(const float *s_in, float *s_o, int row_o, int cs_o, const float (&bw)[Geo<OD>::NSTEP],
                                            float bias, int t0, int t1, int j, int kq, int ch, bool chv) {
    using G = Geo<OD>;
    constexpr int PP = OW * OW;
    const float *ap[NT];
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        int m = (n == 0 ? t0 : t1) * 16 + j;
        m = m < M ? m : M - 1;
        const int r = m / PP, p = m - r * PP;
        ap[n] = s_in + r * ROW_IN + (p / OW) * IW + p % OW + kq * CS_IN;
        acc[n] = f32x4{bias, bias, bias, bias};
    }
    float v[2][NT][9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int n = 0; n < NT; ++n) v[0][n][tap] = ap[n][(tap / 3) * IW + tap % 3];
#pragma unroll
    for (int cq = 0; cq < G::KQ; ++cq) {
        if (cq + 1 < G::KQ) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int n = 0; n < NT; ++n) v[(cq + 1) & 1][n][tap] = ap[n][(cq + 1) * 4 * CS_IN + (tap / 3) * IW + tap % 3];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[cq & 1][n][tap], bw[cq * 9 + tap], acc[n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (chv) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mm = (n == 0 ? t0 : t1) * 16 + kq * 4 + q;
                if (mm < M) { const int rr = mm / PP, pp = mm - rr * PP; s_o[rr * row_o + ch * cs_o + pp] = fmaxf(acc[n][q], 0.0f); }
            }
    }
}

// stage 1 for NQ tiles (16 positions each, tiles Q0 .. Q0+NQ-1 of one row): byte gathers from the staged image, 7 k-steps
template <int Q0, int NQ>
__device__ __forceinline__ void stage1_tiles(const int8_t *img, float *dst, const int (&goff)[5], const int (&off1)[7],
                                             const float (&bw1)[7], float bias1, bool chv) {
    float cv[NQ][7];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt)
#pragma unroll
        for (int s = 0; s < 7; ++s) cv[qt][s] = (float)img[goff[Q0 + qt] + off1[s]];
    f32x4 acc[NQ];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) acc[qt] = f32x4{bias1, bias1, bias1, bias1};
#pragma unroll
    for (int s = 0; s < 7; ++s)
#pragma unroll
        for (int qt = 0; qt < NQ; ++qt) acc[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[qt][s], bw1[s], acc[qt], 0, 0, 0);
    if (chv) {
#pragma unroll
        for (int qt = 0; qt < NQ; ++qt)
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[(Q0 + qt) * 16 + q] = fmaxf(acc[qt][q], 0.0f);
    }
}

template <int OD>
__global__ __launch_bounds__(kBlockM) void k_conv19_mfma(const int8_t *__restrict__ obs, long obs_stride, long rows,
                                                         const float *__restrict__ w1, const float *__restrict__ b1,
                                                         const float *__restrict__ w3, const float *__restrict__ b3,
                                                         float *__restrict__ out, long out_stride, int out_cols,
                                                         const int8_t *__restrict__ onehot, int n_actions,
                                                         const float *__restrict__ mlp_w, const float *__restrict__ mlp_b) {
    using G = Geo<OD>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int8_t *s_img = (int8_t *)lds;                      // [RB][1088] pixel bytes
    float *s_a1 = lds + G::RB * G::IMG / 4;             // [RB][OD][85] stage-1 activations (9x9)
    float *s_a2 = s_a1 + G::RB * G::ROW_A1;             // [RB][OD][53] stage-2 activations (7x7)
    float *s_vec = s_a2 + G::RB * G::ROW_A2;            // [RB][18] inputs of the vector branch
    float *s_mlp = s_vec + G::RB * G::VEC;              // [10][nin] weights, then [10] biases of the vector branch
    float *s_out = s_a1;                                // [RB][OUT_STRIDE], overlays s_a1 (dead after stage 2)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // wave w runs on SIMD w & 3: SIMDs 0,1 hold channel half 0, SIMDs 2,3 half 1; the two waves of a SIMD take tiles
    // sub, sub + 4, ... with sub = (w & 1) and (w & 1) + 2
    const int nh = (wave >> 1) & 1, sub = (wave & 1) + 2 * (wave >> 2);
    const int j = lane & 15, kq = lane >> 4;
    const int ch = nh * 16 + j;                         // the output channel this lane's B column / D column belongs to
    const bool chv = ch < OD;

    const long n_blocks = (rows + G::RB - 1) / G::RB;
    const int nin = 2 + n_actions;
    // The vector branch's parameters live in LDS (crnn_mfma.h: a global load at the head of a row block also waits for the
    // previous block's output stores).
    if (mlp_w) {
        for (int i = tid; i < 10 * nin; i += kBlockM) s_mlp[i] = mlp_w[i];
        if (tid < 10) s_mlp[10 * G::VEC + tid] = mlp_b[tid];
    }
    // The bytes of the next block are fetched right after stage 1 is done with s_img and parked behind stage 2: their latency
    // is covered by stage 2.  A WAVE fetches whole rows with a wave-uniform row pointer and lane + 64 v offsets (crnn_mfma.h:
    // per-thread element indices made the compiler spill hoisted row offsets and wait on every reload, which serialised the
    // 17 byte loads of a block).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int pf[G::RPW][G::NLD], pfo[G::RPW];
    int pf_rv = 0;
#pragma unroll
    for (int h = 0; h < G::RPW; ++h) {
        pfo[h] = 0;
#pragma unroll
        for (int v = 0; v < G::NLD; ++v) pf[h][v] = 0;
    }
    const int p_last = min(lane + 64 * (G::NLD - 1), mlp_w ? G::ROWB - 1 : kPix - 1);  // the direction bytes exist only with the vector branch
    auto fetch = [&](long b) {
        const long r0 = b * G::RB;
        pf_rv = b < n_blocks ? (int)min((long)G::RB, rows - r0) : 0;
#pragma unroll
        for (int h = 0; h < G::RPW; ++h) {
            const int rr = wave_u + 8 * h;
            if (rr < pf_rv) {  // wave-uniform
                const int8_t *row = obs + (r0 + rr) * obs_stride;
#pragma unroll
                for (int v = 0; v + 1 < G::NLD; ++v) pf[h][v] = row[lane + 64 * v];
                pf[h][G::NLD - 1] = row[p_last];
                if (mlp_w && onehot && n_actions > 0) pfo[h] = onehot[(r0 + rr) * n_actions + min(lane, n_actions - 1)];
            }
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int h = 0; h < G::RPW; ++h) {
            const int rr = wave_u + 8 * h;
            if (rr < G::RB) {
                const bool on = rr < pf_rv;   // rows past the end: zeros
                int8_t *dst = s_img + rr * G::IMG;
#pragma unroll
                for (int v = 0; v + 1 < G::NLD; ++v) dst[lane + 64 * v] = on ? (int8_t)pf[h][v] : (int8_t)0;
                const int pl = lane + 64 * (G::NLD - 1);
                const int last = on ? pf[h][G::NLD - 1] : 0;
                if (pl < kPix) dst[pl] = (int8_t)last;
                else if (pl < G::ROWB) s_vec[rr * G::VEC + pl - kPix] = (float)last;
                if (lane < 16) s_vec[rr * G::VEC + 2 + lane] = (on && onehot && lane < n_actions) ? (float)pfo[h] : 0.0f;
            }
        }
    };
    fetch(blockIdx.x);  // in flight while the weights are staged; parked below

    // ---- B operands (weights of channel ch) and the lane's stage-1 gather offsets (bytes into the staged image).  The weights
    // are staged through LDS with coalesced loads (crnn_mfma.h: read straight from global memory, every lane of a wave hits a
    // different cache line and the scattered loads of 8 waves cost tens of microseconds per launch).
    for (int i = tid; i < OD * OD * 9; i += kBlockM) s_a1[i] = w3[i];   // [c_out][c_in][tap]; s_a1 is free until the first stage 1
    for (int i = tid; i < OD * 27; i += kBlockM) s_a2[i] = w1[i];       // [c_out][27]
    __syncthreads();
    float bw1[7];
    int off1[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int k = 4 * s + kq;
        const bool kv = k < 27;
        bw1[s] = (chv && kv) ? s_a2[ch * 27 + k] : 0.0f;
        const int c0 = k / 9, tap = k - c0 * 9;
        off1[s] = kv ? c0 * kFov * kFov + (tap / 3) * kFov + tap % 3 : 0;
    }
    int goff[5];
#pragma unroll
    for (int qt = 0; qt < 5; ++qt) { const int p = qt * 16 + j; goff[qt] = 2 * (p / 9) * kFov + 2 * (p % 9); }
    float bw3[G::NSTEP];
#pragma unroll
    for (int cq = 0; cq < G::KQ; ++cq)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) bw3[cq * 9 + tap] = chv ? s_a1[(ch * OD + 4 * cq + kq) * 9 + tap] : 0.0f;
    __syncthreads();  // the staging areas are reused by the stages below
    const float bias1 = chv ? b1[ch] : 0.0f, bias3 = chv ? b3[ch] : 0.0f;
    const int n_feat = OD * 25 + (mlp_w ? 10 : 0);
    const int n_out = out_cols > n_feat ? out_cols : n_feat;  // columns n_feat .. n_out-1 of a row are written as zeros
    const bool wide_out = (out_stride % 2 == 0) && (((size_t)out) % 8 == 0) && (n_out % 2 == 0);
    const bool quad_out = (out_stride % 4 == 0) && (((size_t)out) % 16 == 0) && (n_out % 4 == 0) && (G::OUT_STRIDE % 4 == 0);

    park();
    for (long blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const long row0 = blk * G::RB;
        const int rv = (int)min((long)G::RB, rows - row0);
        __syncthreads();  // the parked block is visible; the previous block's staged rows (= s_a1) have been streamed out
        float mv = 0.0f;  // vector branch: relu(mlp1([dir_x, dir_y, last-action one-hot])) (base_net.py:66)
        const int mr = tid / 10, mc = tid - mr * 10;
        if (mlp_w && tid < G::RB * 10) {
            mv = s_mlp[10 * G::VEC + mc];
            for (int k = 0; k < nin; ++k) mv = fmaf(s_vec[mr * G::VEC + k], s_mlp[mc * nin + k], mv);
        }
        // ---- stage 1: stride-2 conv of the image.  A row's positions 0..79 are five tiles (p = 16 qt + i), position
        // 80 of all RB rows is one more tile.  Wave `sub` takes rows sub, sub + 4, ...
        {
            // the 35 gather addresses goff[qt] + off1[s] are formed per row block (the empty asm hides goff from the
            // loop-invariant code motion; hoisted, they stayed live through stages 2 and 3 and the kernel spilled)
            int gq[5];
#pragma unroll
            for (int qt = 0; qt < 5; ++qt) { gq[qt] = goff[qt]; asm volatile("" : "+v"(gq[qt])); }
#pragma unroll 1
            for (int i = 0; i < G::RB / 4; ++i) {
                const int8_t *img = s_img + (sub + 4 * i) * G::IMG;
                float *dst = s_a1 + (sub + 4 * i) * G::ROW_A1 + ch * G::CS1 + kq * 4;
                stage1_tiles<0, 3>(img, dst, gq, off1, bw1, bias1, chv);
                stage1_tiles<3, 2>(img, dst, gq, off1, bw1, bias1, chv);
            }
            if (sub == 0) {  // position 80 (x = y = 8) of every row: lane i gathers row i
                const int rr = j < G::RB ? j : G::RB - 1;
                f32x4 acc = {bias1, bias1, bias1, bias1};
#pragma unroll
                for (int s = 0; s < 7; ++s)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((float)s_img[rr * G::IMG + 16 * kFov + 16 + off1[s]], bw1[s], acc, 0, 0, 0);
                if (chv) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (kq * 4 + q < G::RB) s_a1[(kq * 4 + q) * G::ROW_A1 + ch * G::CS1 + 80] = fmaxf(acc[q], 0.0f);
                }
            }
        }
        __syncthreads();
        fetch(blk + gridDim.x);  // s_img / s_vec were last read before this barrier
        // ---- stage 2: conv3 on the 9x9 planes
        {
            int t = sub;
#pragma unroll 1
            for (; t + 4 < G::T2; t += 8)
                conv3_tiles<OD, 2, 9, 7, G::CS1, G::ROW_A1, G::M2>(s_a1, s_a2, G::ROW_A2, G::CS2, bw3, bias3, t, t + 4, j, kq, ch, chv);
            if (t < G::T2) conv3_tiles<OD, 1, 9, 7, G::CS1, G::ROW_A1, G::M2>(s_a1, s_a2, G::ROW_A2, G::CS2, bw3, bias3, t, t, j, kq, ch, chv);
        }
        park();
        __syncthreads();
        // ---- stage 3: the same conv3 on the 7x7 planes, into the staged output rows
        {
            int t = sub;
#pragma unroll 1
            for (; t + 4 < G::T3; t += 8)
                conv3_tiles<OD, 2, 7, 5, G::CS2, G::ROW_A2, G::M3>(s_a2, s_out, G::OUT_STRIDE, 25, bw3, bias3, t, t + 4, j, kq, ch, chv);
            if (t < G::T3) conv3_tiles<OD, 1, 7, 5, G::CS2, G::ROW_A2, G::M3>(s_a2, s_out, G::OUT_STRIDE, 25, bw3, bias3, t, t, j, kq, ch, chv);
        }
        if (mlp_w && tid < G::RB * 10) s_out[mr * G::OUT_STRIDE + OD * 25 + mc] = fmaxf(mv, 0.0f);
        if (n_out > n_feat)  // the zero tail of the staged rows (their LDS held stage-1 activations until stage 2 was done)
            for (int i = tid; i < G::RB * (n_out - n_feat); i += kBlockM) {
                const int rr = i / (n_out - n_feat), k = i - rr * (n_out - n_feat);
                s_out[rr * G::OUT_STRIDE + n_feat + k] = 0.0f;
            }
        __syncthreads();
        // ---- stream the staged rows out: a wave per row, consecutive lanes on consecutive floats
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
        // next iteration: s_a1 (= s_out) is rewritten after its first barrier
    }
}

}  // namespace crnn_mfma19
