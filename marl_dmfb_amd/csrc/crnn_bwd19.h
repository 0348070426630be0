// crnn_bwd19.h -- gradients of the fov-19 conv stack of the reference's CRNN (network/base_net.py:23-33: Conv2d(3, od, 3,
// stride 2) + ReLU, then the SAME conv3 = Conv2d(od, od, 3) + ReLU applied twice: tied weights) w.r.t. its four parameter
// tensors, for the eval network inside VDN.learn (policy/vdn.py:123-128) on MEDA (env/MEDA/meda.py:846-897 observation).
//
// Nothing is saved by the forward (crnn_mfma19.h): a workgroup recomputes a1 = relu(conv1(img)) and a2 = relu(conv3(a1)) of
// its row block on the matrix cores with the forward's own tile routines, then runs the backward of the three layers, all
// GEMM-shaped phases as v_mfma_f32_16x16x4_f32 (exact f32 fma chains, f32 operands):
//   dz3 = g * (a3 > 0)                          a3 = the forward's output (its sign is the ReLU mask)
//   dz2 = convT(dz3, W3) * (a2 > 0)             "gather form": dz3 zero-padded by 2 is convolved with the FLIPPED, TRANSPOSED
//   dz1 = convT(dz2, W3) * (a1 > 0)             weights: the forward's conv3 tile routine again (M = positions, N = c_in, K = (c_out, tap))
//   dW3 = sum dz3 (x) a2-windows + sum dz2 (x) a1-windows   (both applications of the tied module add into the same gradient)
//   dW1 = sum dz1 (x) stride-2 image windows;  db3 = sum dz3 + sum dz2;  db1 = sum dz1
// Weight gradients are GEMMs with M = c_out, N = (c_in, tap), K = (row, position): each wave owns a fixed set of 16x16 output
// tiles whose accumulators stay in registers over ALL row blocks of the workgroup; the workgroup writes ONE partial vector at the
// end and a second kernel adds the <= 256 partial vectors in a fixed order (deterministic, no atomics).
// LDS per row: image bytes, a1 (9x9), a2 (7x7), a 9x9 plane set (dz3 padded, later dz1) and an 11x11 plane set (dz2 padded):
// 45.6 KB at od 32 (2 rows per workgroup), 34.5 KB at od 24 (4 rows); next to them the conv3 weights stay in LDS for the whole
// launch (37 KB at od 32).  A lane's 72 B operands of a conv3-shaped pass are re-read from there at the head of the pass (the
// forward's set before S2, the flipped + transposed set before D2): holding both sets in registers next to the dW3
// accumulators cost 98 spilled registers per lane and a scratch load in front of most matrix instructions.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "crnn_mfma19.h"

namespace crnn_bwd19 {

using crnn_mfma19::f32x4;
using crnn_mfma19::kFov;
using crnn_mfma19::kPix;
constexpr int kBlockB = 512;

template <int OD> struct GeoB19 {
    static constexpr int RBB = OD <= 24 ? 4 : 2;          // rows per iteration (LDS-bound)
    static constexpr int IMG = 1088;                       // staged pixel bytes per row
    static constexpr int CS1 = 85, CS2 = 53, CS4 = 125;    // channel strides of 9x9, 7x7 and 11x11 planes (odd: bank spread)
    static constexpr int ROW_A = OD * CS1, ROW_B = OD * CS2, ROW_C = OD * CS1, ROW_D = OD * CS4;
    static constexpr int KQ = OD / 4, NSTEP = KQ * 9;
    static constexpr int NCOL3 = OD * 9, NT3 = (NCOL3 + 15) / 16;  // dW3 columns (c_in, tap) and their 16-wide tiles
    static constexpr int NTW = (NT3 + 3) / 4;              // dW3 column tiles per wave (a wave owns ONE 16-row half of c_out)
    static constexpr int WS = OD * 9 + 1;                  // c_out stride of the LDS copy of W3 (odd: the 16 channels of a wave hit 16 banks)
    static constexpr int W_FLOATS = OD * WS;
    static constexpr int PFR = (OD * 25 + kBlockB - 1) / kBlockB;       // prefetched (a3, g) pairs per thread and row
    static constexpr int PFI = (kPix + kBlockB - 1) / kBlockB;         // prefetched image bytes per thread and row
    static_assert(PFI <= 4, "packed into one register");
    static constexpr int N_W3 = OD * OD * 9, N_W1 = OD * 27;
    // partial vector of a workgroup: dW3 | dW1 | bias sums per thread: [kBlockB] dz3+dz2, [kBlockB] dz1
    static constexpr int PART = N_W3 + N_W1 + 2 * kBlockB;
    static constexpr int GRADS = N_W3 + OD + N_W1 + OD;   // dW3 | db3 | dW1 | db1
    // per K index of the two dW3 passes (k = (row, position)): offset of dz inside s_C / s_D and of the window origin inside s_B /
    // s_A, as int2; one table read per k-step instead of two divisions and their multiplies
    static constexpr int KA = (RBB * 25 + 3) / 4 * 4, KB = (RBB * 49 + 3) / 4 * 4;
    static constexpr int LUT_INTS = 2 * (KA + KB);
    static constexpr size_t LDS_FLOATS = (size_t)RBB * (IMG / 4 + ROW_A + ROW_B + ROW_C + ROW_D) + W_FLOATS + LUT_INTS;
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "one workgroup per CU");
    static_assert(OD * 16 <= kBlockB, "bias sums: one thread per (channel, 1/16 of the positions)");
};

// conv3-shaped tile pass (3x3, stride 1, valid) for NT tiles of 16 output positions, exactly the forward's gather scheme
// (crnn_mfma19::conv3_tiles), with the epilogue left to the caller: epi(tile position index mm, accumulator value) for the lane's
// channel.  Used for the forward recompute (ReLU store) and for both transposed convolutions (mask + store).
template <int OD, int NT, int IW, int OW, int CS_IN, int ROW_IN, int M, typename Epi>
__device__ __forceinline__ void conv_tiles(const float *s_in, const float (&bw)[GeoB19<OD>::NSTEP], float bias, int t0, int t1, int j,
                                           int kq, Epi epi) {
    using G = GeoB19<OD>;
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
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int mm = (n == 0 ? t0 : t1) * 16 + kq * 4 + q;
            if (mm < M) epi(mm, acc[n][q]);
        }
}

template <int OD>
__global__ __launch_bounds__(kBlockB) void k_conv19_bwd(const int8_t *__restrict__ obs, long obs_stride, long rows,
                                                        const float *__restrict__ a3, long a3_stride, const float *__restrict__ g, long g_stride,
                                                        const float *__restrict__ w1, const float *__restrict__ b1,
                                                        const float *__restrict__ w3, const float *__restrict__ b3, float *__restrict__ part) {
    using G = GeoB19<OD>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int8_t *s_img = (int8_t *)lds;                           // [RBB][1088]
    float *s_A = lds + G::RBB * G::IMG / 4;                  // [RBB][OD][85]  a1 (9x9)
    float *s_B = s_A + G::RBB * G::ROW_A;                    // [RBB][OD][53]  a2 (7x7)
    float *s_C = s_B + G::RBB * G::ROW_B;                    // [RBB][OD][85]  dz3 zero-padded to 9x9, later dz1 (9x9)
    float *s_D = s_C + G::RBB * G::ROW_C;                    // [RBB][OD][125] dz2 zero-padded to 11x11
    float *s_W = s_D + G::RBB * G::ROW_D;                    // [OD][WS]       conv3 weights [c_out][c_in][tap], c_out stride WS
    int2 *s_lutA = (int2 *)(s_W + G::W_FLOATS);              // [KA] dW3 pass a: (dz3 offset in s_C, a2 window origin in s_B) of k
    int2 *s_lutB = s_lutA + G::KA;                           // [KB] dW3 pass b: (dz2 offset in s_D, a1 window origin in s_A) of k
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nh = (wave >> 1) & 1, sub = (wave & 1) + 2 * (wave >> 2);
    const int j = lane & 15, kq = lane >> 4;
    const int ch = nh * 16 + j;
    const bool chv = ch < OD;
    const int chc = chv ? ch : 0;

    // ---- weights into LDS with coalesced loads (scattered per-lane loads straight from global memory cost tens of microseconds
    // per launch: crnn_mfma.h); conv1's 7 B operands per lane stay in registers.
    for (int i = tid; i < OD * OD * 9; i += kBlockB) { const int co = i / (OD * 9); s_W[co * G::WS + (i - co * OD * 9)] = w3[i]; }
    for (int i = tid; i < OD * 27; i += kBlockB) s_B[i] = w1[i];       // [c_out][27]
    for (int k = tid; k < G::KA; k += kBlockB) {
        const int kc = k < G::RBB * 25 ? k : 0, rr = kc / 25, p = kc - rr * 25;
        s_lutA[k] = make_int2(rr * G::ROW_C + (p / 5 + 2) * 9 + p % 5 + 2, rr * G::ROW_B + (p / 5) * 7 + p % 5);
    }
    for (int k = tid; k < G::KB; k += kBlockB) {
        const int kc = k < G::RBB * 49 ? k : 0, rr = kc / 49, p = kc - rr * 49;
        s_lutB[k] = make_int2(rr * G::ROW_D + (p / 7 + 2) * 11 + p % 7 + 2, rr * G::ROW_A + (p / 7) * 9 + p % 7);
    }
    __syncthreads();
    float bw1[7];
    int off1[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int k = 4 * s + kq;
        const bool kv = k < 27;
        bw1[s] = (chv && kv) ? s_B[ch * 27 + k] : 0.0f;
        const int c0 = k / 9, tap = k - c0 * 9;
        off1[s] = kv ? c0 * kFov * kFov + (tap / 3) * kFov + tap % 3 : 0;
    }
    int goff[5];
#pragma unroll
    for (int qt = 0; qt < 5; ++qt) { const int p = qt * 16 + j; goff[qt] = 2 * (p / 9) * kFov + 2 * (p % 9); }
    __syncthreads();  // the staging area is zeroed / reused below
    const float bias1 = chv ? b1[ch] : 0.0f, bias3 = chv ? b3[ch] : 0.0f;
    const float *wf = s_W + chc * G::WS + kq * 9;             // forward set:  W[c_out = ch][c_in = 4cq+kq][tap]         at wf[36 cq + tap]
    const float *wt = s_W + kq * G::WS + chc * 9 + 8;         // backward set: W[c_out = 4cq+kq][c_in = ch][8 - tap]     at wt[4 WS cq - tap]
    float bw[G::NSTEP];

    // ---- weight-gradient roles.  dW3: wave w owns the 16-row half (w & 1) of c_out for column tiles nt = (w >> 1), (w >> 1) + 4,
    // ... (columns n = 16 nt + j = c_in * 9 + tap).  dW1: waves 0..3 own tile (half = w & 1, column tile = w >> 1) of the 27
    // (c0, tap) columns.
    const int wh = wave & 1, wt0 = wave >> 1;
    const int a_ch = 16 * wh + j;
    const bool a_chv = a_ch < OD;
    const int a_chc = a_chv ? a_ch : 0;
    f32x4 acc3[G::NTW];
    int colB_a2[G::NTW], colB_a1[G::NTW];   // per-lane offset of column n inside a row's a2 / a1 planes (channel + tap shift)
    bool colv[G::NTW];
#pragma unroll
    for (int u = 0; u < G::NTW; ++u) {
        const int n = 16 * (wt0 + 4 * u) + j;
        colv[u] = wt0 + 4 * u < G::NT3 && n < G::NCOL3;
        const int ci = colv[u] ? n / 9 : 0, tap = colv[u] ? n % 9 : 0;
        colB_a2[u] = ci * G::CS2 + (tap / 3) * 7 + tap % 3;
        colB_a1[u] = ci * G::CS1 + (tap / 3) * 9 + tap % 3;
        acc3[u] = f32x4{0, 0, 0, 0};
    }
    f32x4 acc1 = {0, 0, 0, 0};
    const int w1_n = 16 * wt0 + j;
    const bool w1_on = wave < 4, w1_colv = w1_on && w1_n < 27;
    const int w1_off = w1_colv ? (w1_n / 9) * kFov * kFov + ((w1_n % 9) / 3) * kFov + (w1_n % 9) % 3 : 0;
    float accb3 = 0.0f, accb1 = 0.0f;        // bias sums: thread = (channel tid % OD, position slice tid / OD), tid < 16 OD
    const int bc = tid % OD, bs = tid / OD;
    const bool b_on = tid < 16 * OD;

    for (int i = tid; i < G::RBB * (G::ROW_C + G::ROW_D); i += kBlockB) s_C[i] = 0.0f;  // the zero borders of the padded planes (s_D follows s_C)
    for (int i = tid; i < G::RBB * G::IMG / 4; i += kBlockB) ((int *)s_img)[i] = 0;

    // ---- the row block's inputs travel one iteration ahead in registers: dz3's two sources (a3, g) of the 5x5 interior and the
    // image bytes.  Slot s of a thread is element tid + s * kBlockB of [RBB][OD][25] resp. [RBB][361].
    float pf_a[G::RBB][G::PFR], pf_g[G::RBB][G::PFR];
    uint32_t pf_i[G::RBB];                    // PFI (3) image bytes of a row packed into one register
    int pf_dst[G::PFR];                       // where feature tid + h * kBlockB of a row lands in its padded planes
#pragma unroll
    for (int h = 0; h < G::PFR; ++h) {
        const int f = tid + h * kBlockB, c = f / 25, q = f - c * 25;
        pf_dst[h] = f < OD * 25 ? c * G::CS1 + (q / 5 + 2) * 9 + q % 5 + 2 : -1;
    }
    auto prefetch = [&](long row0, int rv) {
#pragma unroll
        for (int rr = 0; rr < G::RBB; ++rr) {
            const bool on = rr < rv;
            const float *pa = a3 + (row0 + rr) * a3_stride + tid, *pg = g + (row0 + rr) * g_stride + tid;
#pragma unroll
            for (int h = 0; h < G::PFR; ++h) {
                const bool in = on && pf_dst[h] >= 0;
                pf_a[rr][h] = in ? pa[h * kBlockB] : 0.0f;
                pf_g[rr][h] = in ? pg[h * kBlockB] : 0.0f;
            }
            uint32_t pk = 0;
#pragma unroll
            for (int h = 0; h < G::PFI; ++h) {
                const uint32_t b = (on && tid + h * kBlockB < kPix) ? (uint8_t)obs[(row0 + rr) * obs_stride + tid + h * kBlockB] : 0u;
                pk |= b << (8 * h);
            }
            pf_i[rr] = pk;
        }
    };

    const long n_blocks = (rows + G::RBB - 1) / G::RBB;
    const long per = (n_blocks + gridDim.x - 1) / gridDim.x;
    const long blk0 = (long)blockIdx.x * per, blk1 = min(n_blocks, blk0 + per);
    if (blk0 < blk1) prefetch(blk0 * G::RBB, (int)min((long)G::RBB, rows - blk0 * G::RBB));
    for (long blk = blk0; blk < blk1; ++blk) {
        __syncthreads();  // the previous block's phases are done with every buffer
        // ---- stage in: image bytes, and dz3 = g * (a3 > 0) into the interior of the 9x9 planes; the border is zeroed again (the
        // planes held dz1 of the previous block)
#pragma unroll
        for (int rr = 0; rr < G::RBB; ++rr)
#pragma unroll
            for (int h = 0; h < G::PFI; ++h)
                if (tid + h * kBlockB < kPix) s_img[rr * G::IMG + tid + h * kBlockB] = (int8_t)(pf_i[rr] >> (8 * h));
#pragma unroll 1
        for (int i = tid; i < G::RBB * OD * 56; i += kBlockB) {   // 56 border cells of a 9x9 plane around its 5x5 interior
            const int pl = i / 56, b = i - pl * 56;
            // b < 18: rows 0, 1;  b < 36: rows 7, 8;  else rows 2..6, columns 0, 1, 7, 8
            const int p = b < 18 ? b : b < 36 ? 45 + b : (2 + (b - 36) / 4) * 9 + ((b - 36) % 4 < 2 ? (b - 36) % 4 : (b - 36) % 4 + 5);
            const int rr = pl / OD, c = pl - rr * OD;
            s_C[rr * G::ROW_C + c * G::CS1 + p] = 0.0f;
        }
#pragma unroll
        for (int rr = 0; rr < G::RBB; ++rr)
#pragma unroll
            for (int h = 0; h < G::PFR; ++h)
                if (pf_dst[h] >= 0) s_C[rr * G::ROW_C + pf_dst[h]] = pf_a[rr][h] > 0.0f ? pf_g[rr][h] : 0.0f;
        if (blk + 1 < blk1) prefetch((blk + 1) * G::RBB, (int)min((long)G::RBB, rows - (blk + 1) * G::RBB));
        __syncthreads();
        // ---- S1: a1 = relu(conv1(img)), the forward's stage 1 (rows split over the four waves of a channel half)
#pragma unroll 1
        for (int rr = sub; rr < G::RBB; rr += 4) {
            const int8_t *img = s_img + rr * G::IMG;
            float *dst = s_A + rr * G::ROW_A + ch * G::CS1 + kq * 4;
            crnn_mfma19::stage1_tiles<0, 3>(img, dst, goff, off1, bw1, bias1, chv);
            crnn_mfma19::stage1_tiles<3, 2>(img, dst, goff, off1, bw1, bias1, chv);
        }
        if (sub == (G::RBB < 4 ? 3 : 0)) {  // position 80 of every row: lane i gathers row i
            const int rr = j < G::RBB ? j : G::RBB - 1;
            f32x4 acc = {bias1, bias1, bias1, bias1};
#pragma unroll
            for (int s = 0; s < 7; ++s)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32((float)s_img[rr * G::IMG + 16 * kFov + 16 + off1[s]], bw1[s], acc, 0, 0, 0);
            if (chv) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (kq * 4 + q < G::RBB) s_A[(kq * 4 + q) * G::ROW_A + ch * G::CS1 + 80] = fmaxf(acc[q], 0.0f);
            }
        }
#pragma unroll
        for (int cq = 0; cq < G::KQ; ++cq)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) { const float w = wf[36 * cq + tap]; bw[cq * 9 + tap] = chv ? w : 0.0f; }
        __syncthreads();
        // ---- S2: a2 = relu(conv3(a1))
        {
            constexpr int M2 = G::RBB * 49, T2 = (M2 + 15) / 16;
#pragma unroll 1
            for (int t = sub; t < T2; t += 4)
                conv_tiles<OD, 1, 9, 7, G::CS1, G::ROW_A, M2>(s_A, bw, bias3, t, t, j, kq, [&](int mm, float v) {
                    if (chv) { const int rr = mm / 49, pp = mm - rr * 49; s_B[rr * G::ROW_B + ch * G::CS2 + pp] = fmaxf(v, 0.0f); }
                });
        }
#pragma unroll
        for (int cq = 0; cq < G::KQ; ++cq)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) { const float w = wt[4 * G::WS * cq - tap]; bw[cq * 9 + tap] = chv ? w : 0.0f; }
        __syncthreads();
        // ---- D2: dz2 = convT(dz3) * (a2 > 0) into the interior of the 11x11 planes;  W3a: dW3 += dz3 (x) a2;  db3 += sum dz3
        {
            constexpr int M2 = G::RBB * 49, T2 = (M2 + 15) / 16;
#pragma unroll 1
            for (int t = sub; t < T2; t += 4)
                conv_tiles<OD, 1, 9, 7, G::CS1, G::ROW_C, M2>(s_C, bw, 0.0f, t, t, j, kq, [&](int mm, float v) {
                    if (chv) {
                        const int rr = mm / 49, pp = mm - rr * 49;
                        const float act = s_B[rr * G::ROW_B + ch * G::CS2 + pp];
                        s_D[rr * G::ROW_D + ch * G::CS4 + (pp / 7 + 2) * 11 + pp % 7 + 2] = act > 0.0f ? v : 0.0f;
                    }
                });
            constexpr int K3 = G::RBB * 25;
#pragma unroll 1
            for (int k0 = 0; k0 < K3; k0 += 4) {
                const int k = k0 + kq;
                const bool kv = k < K3;
                const int2 lo = s_lutA[k];   // (k < KA always: K3 rounded up to the step)
                const float *za = s_C + lo.x;                                               // dz3[rr][.][p] in its padded plane
                const float *zb = s_B + lo.y;                                               // a2 window origin of position p
                const float a_raw = za[a_chc * G::CS1];
                const float av = (kv && a_chv) ? a_raw : 0.0f;
#pragma unroll
                for (int u = 0; u < G::NTW; ++u)
                    if (wt0 + 4 * u < G::NT3) {
                        const float b_raw = zb[colB_a2[u]];
                        acc3[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, colv[u] ? b_raw : 0.0f, acc3[u], 0, 0, 0);
                    }
            }
            if (b_on)
#pragma unroll 1
                for (int q = bs; q < G::RBB * 25; q += 16) {
                    const int rr = q / 25, p = q - rr * 25;
                    accb3 += s_C[rr * G::ROW_C + bc * G::CS1 + (p / 5 + 2) * 9 + p % 5 + 2];
                }
        }
        __syncthreads();
        // ---- D1: dz1 = convT(dz2) * (a1 > 0) into the 9x9 planes (dz3 is dead);  W3b: dW3 += dz2 (x) a1;  db3 += sum dz2
        {
            constexpr int M1 = G::RBB * 81, T1 = (M1 + 15) / 16;
#pragma unroll 1
            for (int t = sub; t < T1; t += 4)
                conv_tiles<OD, 1, 11, 9, G::CS4, G::ROW_D, M1>(s_D, bw, 0.0f, t, t, j, kq, [&](int mm, float v) {
                    if (chv) {
                        const int rr = mm / 81, pp = mm - rr * 81;
                        const float act = s_A[rr * G::ROW_A + ch * G::CS1 + pp];
                        s_C[rr * G::ROW_C + ch * G::CS1 + pp] = act > 0.0f ? v : 0.0f;
                    }
                });
            constexpr int K2 = G::RBB * 49;
#pragma unroll 1
            for (int k0 = 0; k0 < K2; k0 += 4) {
                const int k = k0 + kq;
                const bool kv = k < K2;
                const int2 lo = s_lutB[k];
                const float *za = s_D + lo.x;
                const float *zb = s_A + lo.y;
                const float a_raw = za[a_chc * G::CS4];
                const float av = (kv && a_chv) ? a_raw : 0.0f;
#pragma unroll
                for (int u = 0; u < G::NTW; ++u)
                    if (wt0 + 4 * u < G::NT3) {
                        const float b_raw = zb[colB_a1[u]];
                        acc3[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, colv[u] ? b_raw : 0.0f, acc3[u], 0, 0, 0);
                    }
            }
            if (b_on)
#pragma unroll 1
                for (int q = bs; q < G::RBB * 49; q += 16) {
                    const int rr = q / 49, p = q - rr * 49;
                    accb3 += s_D[rr * G::ROW_D + bc * G::CS4 + (p / 7 + 2) * 11 + p % 7 + 2];
                }
        }
        __syncthreads();
        // ---- W1: dW1 += dz1 (x) stride-2 image windows;  db1 += sum dz1
        if (w1_on) {
            constexpr int K1 = G::RBB * 81;
#pragma unroll 1
            for (int k0 = 0; k0 < K1; k0 += 4) {
                const int k = k0 + kq;
                const bool kv = k < K1;
                const int rr = kv ? k / 81 : 0, p = kv ? k - rr * 81 : 0;
                const float av = (kv && a_chv) ? s_C[rr * G::ROW_C + a_chc * G::CS1 + p] : 0.0f;
                const float bv = w1_colv ? (float)s_img[rr * G::IMG + 2 * (p / 9) * kFov + 2 * (p % 9) + w1_off] : 0.0f;
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc1, 0, 0, 0);
            }
        }
        if (b_on)
#pragma unroll 1
            for (int q = bs; q < G::RBB * 81; q += 16) {
                const int rr = q / 81, p = q - rr * 81;
                accb1 += s_C[rr * G::ROW_C + bc * G::CS1 + p];
            }
    }
    // ---- the workgroup's partial vector: D[i = 4 kq + q][j] of a tile is row c_out = 16 half + i, column n = 16 tile + j
    float *mine = part + (size_t)blockIdx.x * G::PART;
#pragma unroll
    for (int u = 0; u < G::NTW; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co = 16 * wh + 4 * kq + q, n = 16 * (wt0 + 4 * u) + j;
            if (colv[u] && co < OD) mine[(size_t)co * G::NCOL3 + n] = acc3[u][q];  // (c_out * OD + c_in) * 9 + tap = c_out * 9 OD + n
        }
    if (w1_on) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c1 = 16 * wh + 4 * kq + q;
            if (w1_colv && c1 < OD) mine[G::N_W3 + c1 * 27 + w1_n] = acc1[q];
        }
    }
    mine[G::N_W3 + G::N_W1 + tid] = b_on ? accb3 : 0.0f;
    mine[G::N_W3 + G::N_W1 + kBlockB + tid] = b_on ? accb1 : 0.0f;
}

// grads = dW3 | db3 | dW1 | db1 from the n_part partial vectors.  64 outputs per 1024-thread workgroup: thread (ty, tx) adds every
// 16th partial vector of output tx (independent loads in flight), the 16 sub-sums meet in LDS in a fixed order (deterministic).
constexpr int kRedY19 = 16;
template <int OD>
__global__ __launch_bounds__(64 * kRedY19) void k_conv19_bwd_reduce(const float *__restrict__ part, int n_part, float *__restrict__ grads) {
    using G = GeoB19<OD>;
    __shared__ float s_sum[kRedY19][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    float s = 0.0f;
    if (i < G::GRADS) {
        if (i < G::N_W3) {
            for (int b = ty; b < n_part; b += kRedY19) s += part[(size_t)b * G::PART + i];
        } else if (i < G::N_W3 + OD) {                               // db3[c]: the 16 position slices of every workgroup
            const int c = i - G::N_W3;
            for (int b = ty; b < n_part; b += kRedY19)
                for (int sl = 0; sl < 16; ++sl) s += part[(size_t)b * G::PART + G::N_W3 + G::N_W1 + sl * OD + c];
        } else if (i < G::N_W3 + OD + G::N_W1) {
            const int k = i - G::N_W3 - OD;
            for (int b = ty; b < n_part; b += kRedY19) s += part[(size_t)b * G::PART + G::N_W3 + k];
        } else {
            const int c = i - G::N_W3 - OD - G::N_W1;
            for (int b = ty; b < n_part; b += kRedY19)
                for (int sl = 0; sl < 16; ++sl) s += part[(size_t)b * G::PART + G::N_W3 + G::N_W1 + kBlockB + sl * OD + c];
        }
    }
    s_sum[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < G::GRADS) {
        float t = 0.0f;
#pragma unroll
        for (int y = 0; y < kRedY19; ++y) t += s_sum[y][tx];
        grads[i] = t;
    }
}

}  // namespace crnn_bwd19
