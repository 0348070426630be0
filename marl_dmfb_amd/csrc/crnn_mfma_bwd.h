// crnn_mfma_bwd.h -- gradients of conv1+ReLU+conv2+ReLU w.r.t. the four parameter tensors on the gfx950 matrix
// cores (v_mfma_f32_16x16x4_f32: exact f32 fma chains), the training partner of crnn_mfma.h.
//
// One persistent 512-thread workgroup per CU walks blocks of RB = 8 rows.  Per block, everything lives in LDS:
//   in   [r][244]       float image of the int8 pixels
//   a1   [r][c1][53]    conv1 activations, RECOMPUTED here with the forward's MFMA tiling (nothing is saved by the
//                       forward: that would be 2 x 400 MB of HBM traffic per learn step for 80 us of MFMA work)
//   dz2  [r][c2][27]    g * (a2 > 0), entry 25 is a permanent zero that K-padding points at
//   da1  [r][c1][53]    zero, then the scatter target of phase B, then dz1 = da1 * (a1 > 0)
// and three GEMM-shaped phases run on MFMA:
//   A  dW2[c2][(c1,tap)] += sum_{r,pos} dz2[r][c2][pos] * a1[r][c1][pos+tap]       M = c2, N = (c1,tap)+ones, K = (r,pos)
//      (the extra all-ones N column yields db2); accumulators stay in registers for the whole kernel
//   B  tmp[(r,pos)][(c1,tap)] = sum_c2 dz2[r][c2][pos] * W2[c2][c1][tap]            M = (r,pos), N = (c1,tap), K = c2
//      scattered with LDS float atomics into da1[r][c1][pos+tap] (col2im): exactly the useful MACs, no zero padding
//   C  dW1[c1][(c0,tap)] += sum_{r,p} dz1[r][c1][p] * in[r][c0][p+tap]              M = c1, N = (c0,tap)+ones, K = (r,p)
// Each workgroup writes ONE partial gradient vector in the layout of the result (dW2 | db2 | dW1 | db1); a second
// kernel adds the <= 256 partials in a fixed order (deterministic up to the LDS atomic order inside phase B).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

namespace crnn_mfma {

constexpr int kBlockB = 512;

template <int OD> struct GeoMB {
    static constexpr int RB = 8;
    static constexpr int CS = 53;                  // a1 / da1 channel stride (49 + 4 zero pads)
    static constexpr int DS = 27;                  // dz2 channel stride (25 + zero + pad)
    static constexpr int A1_ROW = OD * CS, DZ_ROW = OD * DS, IN_STRIDE = 244;
    static constexpr int N2 = OD * 9;              // (c1, tap) columns
    static constexpr int NT_A = (N2 + 1 + 15) / 16;  // phase A N tiles incl. the ones column (14 / 19)
    static constexpr int NT_B = (N2 + 15) / 16;      // phase B N tiles (14 / 18)
    static constexpr int TA = (NT_A + 3) / 4;        // phase A tiles per wave (4 / 5)
    static constexpr int KQ = OD / 4;
    static constexpr int MT_B = (RB * 25 + 15) / 16; // phase B M tiles (13)
    static constexpr int PART = OD * OD * 9 + OD + OD * 27 + OD;
    static constexpr size_t LDS_FLOATS = (size_t)RB * IN_STRIDE + 2 * (size_t)RB * A1_ROW + (size_t)RB * DZ_ROW + 16 * 64;
    static constexpr int NPF_DZ = (RB * OD * 25 + kBlockB - 1) / kBlockB;
    static constexpr int NPF_IN = (RB * 243 + kBlockB - 1) / kBlockB;
};

template <int OD>
__global__ __launch_bounds__(kBlockB) void k_conv9_bwd_mfma(const int8_t *__restrict__ obs, long obs_stride, long rows,
                                                            const float *__restrict__ a2, long a2_stride,
                                                            const float *__restrict__ g, long g_stride,
                                                            const float *__restrict__ w1, const float *__restrict__ b1,
                                                            const float *__restrict__ w2, float *__restrict__ part) {
    using G = GeoMB<OD>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *s_in = lds;
    float *s_a1 = s_in + G::RB * G::IN_STRIDE;
    float *s_da1 = s_a1 + G::RB * G::A1_ROW;
    float *s_dz2 = s_da1 + G::RB * G::A1_ROW;
    float *s_red = s_dz2 + G::RB * G::DZ_ROW;  // [4 tiles][4 regs][64 lanes] phase C combine
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;

    // ---- conv1 recompute: same roles as the forward (crnn_mfma.h)
    const int nh = (wave >> 1) & 1, sub = (wave & 1) + 2 * (wave >> 2);
    const int ch = nh * 16 + j;
    const bool chv = ch < OD;
    float bw1[7];
    int off1[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int k = 4 * s + kq;
        const bool kv = k < 27;
        bw1[s] = (chv && kv) ? w1[ch * 27 + k] : 0.0f;
        const int c0 = k / 9, tap = k - c0 * 9;
        off1[s] = kv ? c0 * 81 + (tap / 3) * 9 + tap % 3 : 0;
    }
    int goff[3];
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) { const int p = qt * 16 + j; goff[qt] = (p / 7) * 9 + p % 7; }
    const float bias1 = chv ? b1[ch] : 0.0f;

    // ---- phase A roles: M half mhA (c2), N tiles ntg + 4 i
    const int mhA = wave & 1, ntg = wave >> 1;
    const int c2A = min(mhA * 16 + j, OD - 1);
    int bnA[G::TA];
    float oneA[G::TA];
#pragma unroll
    for (int i = 0; i < G::TA; ++i) {
        const int n = (ntg + 4 * i) * 16 + j, nc = min(n, G::N2 - 1);
        const int c1 = nc / 9, tap = nc - c1 * 9;
        bnA[i] = c1 * G::CS + (tap / 3) * 7 + tap % 3;
        oneA[i] = n == G::N2 ? 1.0f : 0.0f;
    }
    int aposA[7], bposA[7];  // K index pos = 4 s + kq: dz2 cell (25 = the zero) and a1 offset of the position
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int pos = 4 * s + kq;
        aposA[s] = pos < 25 ? pos : 25;
        bposA[s] = pos < 25 ? (pos / 5) * 7 + pos % 5 : 0;
    }
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 accA[G::TA];
#pragma unroll
    for (int i = 0; i < G::TA; ++i) accA[i] = f32x4{0, 0, 0, 0};

    // ---- phase B roles: wave owns N tiles wave, wave + 8, ... of (c1, tap) and walks all M tiles of (r, pos);
    // B operand = W2[c2 = 4 s + kq][n] of the owned tiles in registers
    constexpr int TB = (G::NT_B + 7) / 8;
    float bw2[TB][G::KQ];
    int bnB[TB];
#pragma unroll
    for (int i = 0; i < TB; ++i) {
        const int n = (wave + 8 * i) * 16 + j;
        const bool nv = wave + 8 * i < G::NT_B && n < G::N2;
#pragma unroll
        for (int s = 0; s < G::KQ; ++s) bw2[i][s] = nv ? w2[(size_t)(4 * s + kq) * G::N2 + n] : 0.0f;
        const int nc = min(n, G::N2 - 1), c1 = nc / 9, tap = nc - c1 * 9;
        bnB[i] = nv ? c1 * G::CS + (tap / 3) * 7 + tap % 3 : -1;
    }

    // ---- phase C roles: tile (mhC, ntC), rows of parity rpC
    const int mhC = wave & 1, ntC = (wave >> 1) & 1, rpC = wave >> 2;
    const int c1C = min(mhC * 16 + j, OD - 1);
    const int nC = ntC * 16 + j, nCc = min(nC, 26);
    const int binC = (nCc / 9) * 81 + ((nCc % 9) / 3) * 9 + nCc % 3;
    const float oneC = nC == 27 ? 1.0f : 0.0f;
    int inposC[13];
#pragma unroll
    for (int s = 0; s < 13; ++s) { const int p = 4 * s + kq; inposC[s] = p < 49 ? (p / 7) * 9 + p % 7 : 0; }
    f32x4 accC = {0, 0, 0, 0};

    // permanent zeros: a1 / da1 pads (49..52) and the dz2 zero cells (25, 26)
    for (int i = tid; i < G::RB * OD; i += kBlockB) {
#pragma unroll
        for (int k = 49; k < G::CS; ++k) s_a1[i * G::CS + k] = 0.0f;
        s_dz2[i * G::DS + 25] = 0.0f;
        s_dz2[i * G::DS + 26] = 0.0f;
    }

    const long n_blocks = (rows + G::RB - 1) / G::RB;
    const long per = (n_blocks + gridDim.x - 1) / gridDim.x;
    const long blk0 = (long)blockIdx.x * per, blk1 = min(n_blocks, blk0 + per);

    float pf_dz[G::NPF_DZ], pf_in[G::NPF_IN];
    auto fetch = [&](long b) {
        const long r0 = b * G::RB;
        const int rvb = b < blk1 ? (int)min((long)G::RB, rows - r0) : 0;
#pragma unroll
        for (int u = 0; u < G::NPF_DZ; ++u) {
            const int i = tid + u * kBlockB, rr = i / (OD * 25), rem = i - rr * (OD * 25);
            float v = 0.0f;
            if (i < G::RB * OD * 25 && rr < rvb) {
                const float act = a2[(r0 + rr) * a2_stride + rem];
                const float gv = g[(r0 + rr) * g_stride + rem];
                v = act > 0.0f ? gv : 0.0f;
            }
            pf_dz[u] = v;
        }
#pragma unroll
        for (int u = 0; u < G::NPF_IN; ++u) {
            const int i = tid + u * kBlockB, rr = i / 243, p = i - rr * 243;
            pf_in[u] = (i < G::RB * 243 && rr < rvb) ? (float)obs[(r0 + rr) * obs_stride + p] : 0.0f;
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int u = 0; u < G::NPF_DZ; ++u) {
            const int i = tid + u * kBlockB, rc = i / 25, pos = i - rc * 25;  // rc = r * OD + c2
            if (i < G::RB * OD * 25) s_dz2[rc * G::DS + pos] = pf_dz[u];
        }
#pragma unroll
        for (int u = 0; u < G::NPF_IN; ++u) {
            const int i = tid + u * kBlockB, rr = i / 243, p = i - rr * 243;
            if (i < G::RB * 243) s_in[rr * G::IN_STRIDE + p] = pf_in[u];
        }
    };
    fetch(blk0);
    for (long blk = blk0; blk < blk1; ++blk) {
        __syncthreads();  // phase C of the previous block is done with s_in / s_da1
        park();
        for (int i = tid; i < G::RB * G::A1_ROW; i += kBlockB) s_da1[i] = 0.0f;
        __syncthreads();
        fetch(blk + 1);
        // ---- conv1 recompute -> s_a1 (see crnn_mfma.h; RB = 8: rows sub and sub + 4)
        {
#pragma unroll
            for (int i = 0; i < G::RB / 4; ++i) {
                const int rr = sub + 4 * i;
                f32x4 acc[3];
#pragma unroll
                for (int qt = 0; qt < 3; ++qt) acc[qt] = f32x4{bias1, bias1, bias1, bias1};
                float cv[3][7];
#pragma unroll
                for (int qt = 0; qt < 3; ++qt)
#pragma unroll
                    for (int s = 0; s < 7; ++s) cv[qt][s] = s_in[rr * G::IN_STRIDE + goff[qt] + off1[s]];
#pragma unroll
                for (int s = 0; s < 7; ++s)
#pragma unroll
                    for (int qt = 0; qt < 3; ++qt) acc[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[qt][s], bw1[s], acc[qt], 0, 0, 0);
                if (chv) {
                    float *dst = s_a1 + rr * G::A1_ROW + ch * G::CS + kq * 4;
#pragma unroll
                    for (int qt = 0; qt < 3; ++qt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) dst[qt * 16 + q] = fmaxf(acc[qt][q], 0.0f);
                }
            }
            if (sub == 0) {
                const int rr = j < G::RB ? j : G::RB - 1;
                f32x4 acc = {bias1, bias1, bias1, bias1};
#pragma unroll
                for (int s = 0; s < 7; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(s_in[rr * G::IN_STRIDE + 60 + off1[s]], bw1[s], acc, 0, 0, 0);
                if (chv) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (kq * 4 + q < G::RB) s_a1[(kq * 4 + q) * G::A1_ROW + ch * G::CS + 48] = fmaxf(acc[q], 0.0f);
                }
            }
        }
        __syncthreads();
        // ---- phase A: dW2 / db2 accumulate over the K = (r, pos) of this block
#ifndef CRNN_PROBE_SKIP_A
        {
            auto phaseA = [&](auto nTilesC) {
                constexpr int NTL = decltype(nTilesC)::value;
                for (int rr = 0; rr < G::RB; ++rr) {
                    const float *pa = s_dz2 + rr * G::DZ_ROW + c2A * G::DS;
                    const float *pb = s_a1 + rr * G::A1_ROW;
                    float av[7], bv[NTL][7];
#pragma unroll
                    for (int s = 0; s < 7; ++s) {
                        av[s] = pa[aposA[s]];
#pragma unroll
                        for (int i = 0; i < NTL; ++i) bv[i][s] = oneA[i] != 0.0f ? 1.0f : pb[bnA[i] + bposA[s]];
                    }
#pragma unroll
                    for (int s = 0; s < 7; ++s)
#pragma unroll
                        for (int i = 0; i < NTL; ++i) accA[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[i][s], accA[i], 0, 0, 0);
                }
            };
            if (ntg + 4 * (G::TA - 1) < G::NT_A) phaseA(std::integral_constant<int, G::TA>{});
            else phaseA(std::integral_constant<int, G::TA - 1>{});
        }
#endif
        // ---- phase B: tmp = dz2 x W2 per (M tile, N tile), scattered into da1
#ifndef CRNN_PROBE_SKIP_B
        for (int mt = 0; mt < G::MT_B; ++mt) {
            int m = mt * 16 + j;
            m = m < G::RB * 25 ? m : G::RB * 25 - 1;
            const int ra = m / 25, pa_ = m - ra * 25;
            const float *pa = s_dz2 + ra * G::DZ_ROW + kq * G::DS + pa_;
            float av[G::KQ];
#pragma unroll
            for (int s = 0; s < G::KQ; ++s) av[s] = pa[s * 4 * G::DS];
            // scatter base of my 4 output rows m' = mt*16 + kq*4 + q (consecutive positions): r*A1_ROW + x*7 + y, or -1
            int dsto[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mm = mt * 16 + kq * 4 + q;
                const int r2 = mm / 25, p2 = mm - r2 * 25;
                dsto[q] = mm < G::RB * 25 ? r2 * G::A1_ROW + (p2 / 5) * 7 + p2 % 5 : -1;
            }
#pragma unroll
            for (int i = 0; i < TB; ++i) {
                if (wave + 8 * i < G::NT_B) {
                    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < G::KQ; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bw2[i][s], acc, 0, 0, 0);
                    if (bnB[i] >= 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (dsto[q] >= 0) atomicAdd(&s_da1[dsto[q] + bnB[i]], acc[q]);
                    }
                }
            }
        }
#endif
        __syncthreads();
        // ---- dz1 = da1 * (a1 > 0) in place (pads stay zero)
        for (int i = tid; i < G::RB * OD * 49; i += kBlockB) {
            const int rc = i / 49, p = i - rc * 49;
            if (!(s_a1[rc * G::CS + p] > 0.0f)) s_da1[rc * G::CS + p] = 0.0f;
        }
        __syncthreads();
        // ---- phase C: dW1 / db1 over K = (r, p), rows of my parity
#ifndef CRNN_PROBE_SKIP_C
        for (int rr = rpC; rr < G::RB; rr += 2) {
            const float *pa = s_da1 + rr * G::A1_ROW + c1C * G::CS + kq;
            const float *pb = s_in + rr * G::IN_STRIDE + binC;
            float av[13], bv[13];
#pragma unroll
            for (int s = 0; s < 13; ++s) { av[s] = pa[4 * s]; bv[s] = pb[inposC[s]]; }
#pragma unroll
            for (int s = 0; s < 13; ++s) accC = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], oneC != 0.0f ? 1.0f : bv[s], accC, 0, 0, 0);
        }
#endif
    }
    // ---- write this workgroup's partial gradient vector
    float *pp = part + (size_t)blockIdx.x * G::PART;
#pragma unroll
    for (int i = 0; i < G::TA; ++i) {
        const int nt = ntg + 4 * i;
        if (nt < G::NT_A) {
            const int n = nt * 16 + j;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c2 = mhA * 16 + kq * 4 + q;
                if (c2 < OD) {
                    if (n < G::N2) pp[c2 * G::N2 + n] = accA[i][q];
                    else if (n == G::N2) pp[OD * G::N2 + c2] = accA[i][q];
                }
            }
        }
    }
    __syncthreads();
    if (rpC == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s_red[((wave & 3) * 4 + q) * 64 + lane] = accC[q];
    }
    __syncthreads();
    if (rpC == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v = accC[q] + s_red[((wave & 3) * 4 + q) * 64 + lane];
            const int c1 = mhC * 16 + kq * 4 + q;
            if (c1 < OD) {
                if (nC < 27) pp[OD * G::N2 + OD + c1 * 27 + nC] = v;
                else if (nC == 27) pp[OD * G::N2 + OD + OD * 27 + c1] = v;
            }
        }
    }
}

template <int OD>
__global__ void k_conv9_bwd_mfma_reduce(const float *__restrict__ part, int n_part, float *__restrict__ grads) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= GeoMB<OD>::PART) return;
    float acc = 0.0f;
    for (int b = 0; b < n_part; ++b) acc += part[(size_t)b * GeoMB<OD>::PART + i];
    grads[i] = acc;
}

}  // namespace crnn_mfma
