// crnn_ops.hip -- fused conv1+ReLU+conv2+ReLU forward of the reference's CRNN front end for fov 9
// (network/base_net.py:23-33,63-65), hand-written for gfx950.  See include/crnn_ops.h.
//
// Why a kernel: in the rollout the network sees (envs x agents) = 16 384 rows of 3x9x9 int8 pixels
// per lock-step.  Through library GEMMs this is two im2col copies plus two GEMMs with N = 24 output
// columns (17 TFLOP/s, ~0.55 ms).  Here one workgroup keeps both weight tensors in LDS and pushes
// blocks of RB rows (21 for od 24) through conv1 (LDS -> registers -> LDS) and conv2 (LDS -> registers -> HBM):
//   thread (row r, channel c) owns one output channel of one row; per input channel it pulls the
//   7x7 conv1 activations of row r into registers with 16-byte LDS reads (the same address for all
//   channels of a row = broadcast) and applies its 9 weights: 225 FMAs per 58 LDS words.  conv2
//   weights sit in LDS as (c1, tap, c2) so the 24/32 channel lanes of a row hit consecutive banks.
// fp32 VALU FMAs (v_fma_f32), no MFMA: the fp32 MFMA rate equals the VALU rate on gfx950 and the
// tiles (24 x 25) are far below an MFMA-friendly shape.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/crnn_ops.h"
#include "crnn_mfma.h"
#include "crnn_mfma19.h"
#include "crnn_bwd19.h"

namespace {

constexpr int kBlock = 512;  // 8 waves: one workgroup per CU keeps ONE copy of the weights in LDS for 2 waves per SIMD
constexpr int kA1Stride = 52;  // 49 conv1 activations per (row, channel), padded to a 16-byte multiple
constexpr int kDz1Unit = 108;  // dz1 of a (row, channel pair): 2 x 49 floats interleaved, padded (16-byte multiple, 108 mod 32 = 12)

constexpr int kLdsBudget = 158 * 1024;  // of the CU's 160 KiB

// ---- backward of conv1+ReLU+conv2+ReLU w.r.t. the four parameter tensors (the int8 observation needs no
// gradient).  One persistent 512-thread workgroup per CU walks blocks of RBB rows; every thread keeps
// its share of the weight-gradient sums in registers over ALL the rows of the workgroup and writes
// them once to part[blockIdx][...]; the host adds the <= 256 partial vectors (deterministic, no atomics).
//   P0  stage a1 (saved by the forward), dz2 = g * (a2 > 0) and the float input rows in LDS
//   P1  dW2[c2][c1][tap] += sum_pos dz2[c2][pos] * a1[c1][pos + tap]         thread = (c2, c1) pair(s)
//   P2  da1[c1][p] = sum_c2,tap dz2[c2][p - tap] * W2[c2][c1][tap]; dz1 = da1 * (a1 > 0)   lane quad = (row, channel pair), lane = quarter of c2
//   P3  dW1[c1][c0][tap] += sum_pos dz1[c1][pos] * in[c0][pos + tap]          thread = (c1, c0, tap) item(s)
template <int OD> struct GeoB {
    static constexpr int DZ2 = 28;  // 25 padded to a 16-byte multiple
    static constexpr int DZ_ROW = OD * DZ2 + 4;  // +4 words: OD*28 is a multiple of the 32 LDS banks, and P2's lanes span rows
    static constexpr int ROW_FLOATS = OD * kA1Stride + OD * DZ2 + 4 + 244 + (OD / 2) * kDz1Unit;  // a1, dz2 (+4 pad), in, dz1
    static constexpr int FIXED_FLOATS = OD * OD * 9;
    static constexpr int RBB = ((kLdsBudget / 4 - FIXED_FLOATS) / ROW_FLOATS) < (kBlock / (2 * OD)) ? ((kLdsBudget / 4 - FIXED_FLOATS) / ROW_FLOATS)
                                                                                                   : (kBlock / (2 * OD));
    static constexpr size_t LDS_FLOATS = (size_t)FIXED_FLOATS + (size_t)RBB * ROW_FLOATS + OD * 28;  // + conv1 weights and biases
    // dW2 work units (c2, channel PAIR of c1): the first NA units one thread each over all rows; the remainder (od 24: 32 units)
    // spread over the upper half of the workgroup, each (unit, row slice) a different thread (32 units x 8 slices)
    static constexpr int UNITS = OD * (OD / 2);
    static constexpr int NA = UNITS >= kBlock ? kBlock : kBlock / 2;
    static constexpr int NB = UNITS - NA;
    static constexpr int XSLICES = NB > 0 ? (kBlock - NA) / NB : 1;
    static_assert(NB >= 0 && (NB == 0 || (NA + NB * XSLICES <= kBlock && XSLICES <= 8)), "dW2 roles");
    // dW1 work items (c1, c0, kx) x row slices
    static constexpr int ITEMS3 = (OD / 2) * 9;  // (channel PAIR, c0, kx)
    static constexpr int RS3 = kBlock / ITEMS3;  // 4 (od 24) or 3 (od 32)
    // partial vector: dW2 [kBlock][9 taps][2 channels] (thread slot) | db2 [OD] | dW1 [kBlock][2 channels][3] | db1 [OD]
    static constexpr int PART = kBlock * 9 + kBlock * 9 + OD + kBlock * 6 + OD;
};

// Nothing is saved by the forward: the conv1 activations of each row block are recomputed on the matrix cores (the
// forward's f32 MFMA tiling, crnn_mfma.h) and the next block's dz2 / pixel rows are fetched into registers while phase
// P3 runs, so no HBM latency sits between two barriers.
template <int OD>
__global__ __launch_bounds__(kBlock) void k_conv9_bwd(const int8_t *__restrict__ obs, long obs_stride, long rows,
                                                      const float *__restrict__ a2, long a2_stride, const float *__restrict__ g, long g_stride,
                                                      const float *__restrict__ w2, float *__restrict__ part,
                                                      const float *__restrict__ w1, const float *__restrict__ b1) {
    using G = GeoB<OD>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *s_w2 = lds;                                  // [c2][tap][c1]
    float *s_a1 = s_w2 + OD * OD * 9;                   // [RBB][OD][52]
    float *s_dz2 = s_a1 + G::RBB * OD * kA1Stride;      // [RBB][OD][28]
    float *s_in = s_dz2 + G::RBB * G::DZ_ROW;            // [RBB][244]
    float *s_dz1 = s_in + G::RBB * 244;                 // [RBB][OD/2][108]: dz1 of a channel pair, interleaved
    float *s_w1 = s_dz1 + G::RBB * (OD / 2) * kDz1Unit; // [OD][27] conv1 weights, then [OD] biases
    const int tid = threadIdx.x;
    for (int i = tid; i < OD * 27; i += kBlock) s_w1[i] = w1[i];
    if (tid < OD) s_w1[OD * 27 + tid] = b1[tid];
    for (int i = tid; i < OD * OD * 9; i += kBlock) {   // global (c2, c1, tap) -> LDS (c2, tap, c1)
        const int c2 = i / (OD * 9), rem = i - c2 * OD * 9, c1 = rem / 9, tap = rem - c1 * 9;
        s_w2[(c2 * 9 + tap) * OD + c1] = w2[i];
    }
    float accB2 = 0.0f, accB1 = 0.0f, accB1y = 0.0f;
    float2 accW2[9], accW1[3];  // persistent over all rows of the workgroup
#pragma unroll
    for (int k = 0; k < 9; ++k) accW2[k] = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < 3; ++k) accW1[k] = make_float2(0.0f, 0.0f);
    // dW2 role: unit (c2, channel pair), first row and row step; dW1 (item, slice)
    const bool pa_full = tid < G::NA;
    const int pa_unit = pa_full ? tid : G::NA + (G::NB > 0 ? (tid - G::NA) % (G::NB > 0 ? G::NB : 1) : 0);
    const int pa_slice = pa_full ? 0 : (G::NB > 0 ? (tid - G::NA) / (G::NB > 0 ? G::NB : 1) : 0);
    const bool pa_on = pa_full || (G::NB > 0 && pa_slice < G::XSLICES);
    const int pa_step = pa_full ? 1 : G::XSLICES;
    const int pa_c2 = pa_unit / (OD / 2), pa_cp = pa_unit - pa_c2 * (OD / 2);
    const int i3 = tid % G::ITEMS3, s3 = tid / G::ITEMS3;
    const bool p3_on = s3 < G::RS3;
    const int cp_3 = i3 / 9, c0_3 = (i3 - cp_3 * 9) / 3, kx_3 = i3 - cp_3 * 9 - c0_3 * 3;

    // P2 role: kSplit2 neighbouring lanes = (row r2, channel PAIR cp), lane q of them = its share of the c2 range.  db2 role: thread (row br,
    // channel bc) of the first RBB * OD threads.
    // (od 24: the dW2 phase keeps waves 0-3 busy over all rows and waves 4-7 only for their row slices, so P2 lives on waves 4-7
    // alone, two lanes per unit: every SIMD then carries one long P1 wave and one P2 wave instead of two waves that both run
    // P1 then P2 with very different P1 lengths.  od 32: P1 is even over the waves, P2 takes four lanes per unit on waves 0-5.)
    constexpr int kSplit2 = G::NB > 0 ? 2 : 4, kBase2 = G::NB > 0 ? kBlock / 2 : 0;
    const int q2 = (tid - kBase2) & (kSplit2 - 1), unit2 = tid >= kBase2 ? (tid - kBase2) / kSplit2 : G::RBB * (OD / 2);
    const int r2 = unit2 / (OD / 2), cp2 = unit2 - r2 * (OD / 2);
    const int br = tid / OD, bc = tid - br * OD;

    const long n_blocks = (rows + G::RBB - 1) / G::RBB;
    const long per = (n_blocks + gridDim.x - 1) / gridDim.x;
    const long blk0 = (long)blockIdx.x * per, blk1 = min(n_blocks, blk0 + per);

    // ---- conv1 roles (wave = (channel half, row residue)), B operands, prefetch registers
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = tid & 63, wave = tid >> 6, j16 = lane & 15, kq = lane >> 4;
    const int c_nh = (wave >> 1) & 1, c_sub = (wave & 1) + 2 * (wave >> 2), c_ch = c_nh * 16 + j16;
    const bool c_chv = c_ch < OD;
    constexpr int NPD = (G::RBB * OD * 25 + kBlock - 1) / kBlock, NPI = (G::RBB * 243 + kBlock - 1) / kBlock;
    // The prefetch is branch-free: clamped (always valid) addresses, RAW values in the registers; dz2 = g * (a2 > 0) and the
    // int -> float conversion happen when the block is parked.  With the select at the load, every slot waited for its own
    // pair of loads before the next slot's were issued: 17 global-memory latencies in a row per block.
    float pfa[NPD], pfg[NPD];
    int pfi[NPI];
#pragma unroll
    for (int u = 0; u < NPD; ++u) { pfa[u] = 0.0f; pfg[u] = 0.0f; }
#pragma unroll
    for (int u = 0; u < NPI; ++u) pfi[u] = 0;
    auto fetch = [&](long b) {
        const long r0 = b * G::RBB;
        const int rvb = b < blk1 ? (int)min((long)G::RBB, rows - r0) : 0;
        if (rvb <= 0) return;  // uniform: nothing follows this block
        int t_ = tid;
        asm volatile("" : "+v"(t_));  // opaque: keeps the index arithmetic below out of the loop-invariant registers
#pragma unroll
        for (int u = 0; u < NPD; ++u) {
            const int i = min(t_ + u * kBlock, G::RBB * OD * 25 - 1), rr = i / (OD * 25), rem = i - rr * OD * 25;
            const long row = r0 + min(rr, rvb - 1);
            pfa[u] = a2[row * a2_stride + rem];
            pfg[u] = g[row * g_stride + rem];
        }
#pragma unroll
        for (int u = 0; u < NPI; ++u) {
            const int i = min(t_ + u * kBlock, G::RBB * 243 - 1), rr = i / 243, pp = i - rr * 243;
            pfi[u] = obs[(r0 + min(rr, rvb - 1)) * obs_stride + pp];
        }
    };
    fetch(blk0);
    for (long blk = blk0; blk < blk1; ++blk) {
        const long row0 = blk * G::RBB;
        const int rv = (int)min((long)G::RBB, rows - row0);
        __syncthreads();
        // ---- P0: park the prefetched rows, then conv1 + ReLU on MFMA into s_a1 ([row][c][52])
        int t_ = tid;
        asm volatile("" : "+v"(t_));
#pragma unroll
        for (int u = 0; u < NPD; ++u) {
            const int i = t_ + u * kBlock, rc = i / 25, k = i - rc * 25;
            if (i < G::RBB * OD * 25) s_dz2[(rc / OD) * G::DZ_ROW + (rc % OD) * G::DZ2 + k] = (rc / OD < rv && pfa[u] > 0.0f) ? pfg[u] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < NPI; ++u) {
            const int i = t_ + u * kBlock, rr = i / 243, pp = i - rr * 243;
            if (i < G::RBB * 243) s_in[rr * 244 + pp] = rr < rv ? (float)pfi[u] : 0.0f;  // rows past the end: finite zeros
        }
        __syncthreads();
        // the conv1 B operands are re-read from LDS every block: nothing of this phase stays in registers
        // while the gradient phases run at the register limit
        int off1[7], goff[3];
        float bw1[7];
        int kq_ = kq, j_ = j16;
        asm volatile("" : "+v"(kq_), "+v"(j_));
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const int k = 4 * s + kq_, c0 = k / 9, tap = k - c0 * 9;
            off1[s] = k < 27 ? c0 * 81 + (tap / 3) * 9 + tap % 3 : 0;
            const float w = s_w1[(c_chv ? c_ch : 0) * 27 + (k < 27 ? k : 0)];
            bw1[s] = (c_chv && k < 27) ? w : 0.0f;
        }
        const float c_bias = c_chv ? s_w1[OD * 27 + c_ch] : 0.0f;
#pragma unroll
        for (int qt = 0; qt < 3; ++qt) { const int p = qt * 16 + j_; goff[qt] = (p / 7) * 9 + p % 7; }
        for (int rr = c_sub; rr < G::RBB; rr += 4) {
            float cv[3][7];
#pragma unroll
            for (int qt = 0; qt < 3; ++qt)
#pragma unroll
                for (int s = 0; s < 7; ++s) cv[qt][s] = s_in[rr * 244 + goff[qt] + off1[s]];
            f32x4 acc[3];
#pragma unroll
            for (int qt = 0; qt < 3; ++qt) acc[qt] = f32x4{c_bias, c_bias, c_bias, c_bias};
#pragma unroll
            for (int s = 0; s < 7; ++s)
#pragma unroll
                for (int qt = 0; qt < 3; ++qt) acc[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[qt][s], bw1[s], acc[qt], 0, 0, 0);
            if (c_chv) {
                float *dst = s_a1 + ((rr * (OD / 2) + (c_ch >> 1)) * kA1Stride + kq * 4) * 2 + (c_ch & 1);  // [row][pair][position][2]
#pragma unroll
                for (int qt = 0; qt < 3; ++qt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) dst[(qt * 16 + q) * 2] = fmaxf(acc[qt][q], 0.0f);
            }
        }
        if (c_sub == 0) {  // position 48 of every row: lane i gathers row i
            const int rr = j16 < G::RBB ? j16 : G::RBB - 1;
            f32x4 acc = {c_bias, c_bias, c_bias, c_bias};
#pragma unroll
            for (int s = 0; s < 7; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(s_in[rr * 244 + 60 + off1[s]], bw1[s], acc, 0, 0, 0);
            if (c_chv) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (kq * 4 + q < G::RBB) s_a1[(((kq * 4 + q) * (OD / 2) + (c_ch >> 1)) * kA1Stride + 48) * 2 + (c_ch & 1)] = fmaxf(acc[q], 0.0f);
            }
        }
        __syncthreads();
        // ---- P1: dW2 (and db2)
        // ---- P1: dW2[c2][c1][tap] += sum_pos dz2[c2][pos] * a1[c1][pos + tap] for BOTH channels of the unit's pair: every
        // multiply-add a v_pk_fma_f32 on a natural register pair -- (sums of the two channels) += dz2 (broadcast) * (a1 of the two
        // channels, interleaved in LDS).  One channel per thread, the compiler packed neighbouring taps and spent as many
        // register moves as it saved multiply-adds; a thread also re-read dz2 for every single channel.
        auto unit_rows = [&](int c2, int cp, int r_begin, int r_step) {
#pragma unroll 1
            for (int rr = r_begin; rr < rv; rr += r_step) {
                float dz[G::DZ2];
                float2 a[kA1Stride];
                const float4 *pd = (const float4 *)__builtin_assume_aligned(s_dz2 + rr * G::DZ_ROW + c2 * G::DZ2, 16);
                const float4 *pa = (const float4 *)__builtin_assume_aligned(s_a1 + (size_t)(rr * (OD / 2) + cp) * 2 * kA1Stride, 16);
#pragma unroll
                for (int j = 0; j < G::DZ2 / 4; ++j) { const float4 t = pd[j]; dz[4 * j] = t.x; dz[4 * j + 1] = t.y; dz[4 * j + 2] = t.z; dz[4 * j + 3] = t.w; }
#pragma unroll
                for (int j = 0; j < kA1Stride / 2; ++j) { const float4 t = pa[j]; a[2 * j] = make_float2(t.x, t.y); a[2 * j + 1] = make_float2(t.z, t.w); }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int x = 0; x < 5; ++x)
#pragma unroll
                            for (int y = 0; y < 5; ++y) {
                                const float z = dz[x * 5 + y];
                                const float2 v = a[(x + kx) * 7 + y + ky];
                                accW2[kx * 3 + ky].x = fmaf(z, v.x, accW2[kx * 3 + ky].x);
                                accW2[kx * 3 + ky].y = fmaf(z, v.y, accW2[kx * 3 + ky].y);
                            }
            }
        };
        // db2[c2] += sum of dz2[row][c2][.]: thread (row br, channel bc).  Inside pair_rows (the thread of pair (c2, 0) has the
        // row's dz2 in registers) the 25 adds ran in every wave for one lane in 24.
        if (tid < G::RBB * OD && br < rv) {
            const float4 *pd = (const float4 *)__builtin_assume_aligned(s_dz2 + br * G::DZ_ROW + bc * G::DZ2, 16);
            float t = 0.0f;
#pragma unroll
            for (int j = 0; j < 6; ++j) { const float4 v = pd[j]; t += (v.x + v.y) + (v.z + v.w); }
            accB2 += t + pd[6].x;
        }
#ifndef CRNN_PROBE_SKIP_P1
        if (pa_on) unit_rows(pa_c2, pa_cp, pa_slice, pa_step);
#endif
        // ---- P2: da1[c1][p] = sum over c2, tap of dz2[c2][p - tap] * W2[c2][c1][tap];  dz1 = da1 * (a1 > 0).
        // A thread owns TWO channels (c1 = 2 cp, 2 cp + 1) of one row and a quarter (od 24: half) of the c2 range: every multiply-add is a
        // v_pk_fma_f32 on a natural register pair -- (da[p] of both channels) += dz2 (broadcast) * (W2 of both channels, adjacent
        // in the [c2][tap][c1] LDS copy).  With one channel per thread the compiler packed neighbouring positions instead and
        // spent 155 register moves per 135 multiply-adds on lining the pairs up.  The partial sums of a unit's lanes meet through
        // DPP (no LDS, no barrier); lane 0 of the unit applies the ReLU mask and writes dz1, pair-interleaved:
        // s_dz1[(row * OD/2 + cp) * kDz1Unit + 2 * position + (c1 & 1)]  (kDz1Unit = 108: P3's eight channel pairs per wave on
        // eight different bank groups).
#ifndef CRNN_PROBE_SKIP_P2
        const bool p2 = unit2 < G::RBB * (OD / 2) && r2 < rv;
#else
        const bool p2 = false;
#endif
        if (p2) {
            float2 da[49];
#pragma unroll
            for (int k = 0; k < 49; ++k) da[k] = make_float2(0.0f, 0.0f);
            const int cbeg = q2 * (OD / kSplit2), cend = cbeg + OD / kSplit2;
#pragma unroll 1
            for (int c2 = cbeg; c2 < cend; ++c2) {
                float dz[G::DZ2];
                float2 w[9];
                const float4 *pd = (const float4 *)__builtin_assume_aligned(s_dz2 + r2 * G::DZ_ROW + c2 * G::DZ2, 16);
#pragma unroll
                for (int j = 0; j < G::DZ2 / 4; ++j) { const float4 t = pd[j]; dz[4 * j] = t.x; dz[4 * j + 1] = t.y; dz[4 * j + 2] = t.z; dz[4 * j + 3] = t.w; }
#pragma unroll
                for (int k = 0; k < 9; ++k) w[k] = *(const float2 *)(s_w2 + (c2 * 9 + k) * OD + 2 * cp2);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int x = 0; x < 5; ++x)
#pragma unroll
                            for (int y = 0; y < 5; ++y) {
                                float2 &d = da[(x + kx) * 7 + y + ky];
                                const float z = dz[x * 5 + y];
                                d.x = fmaf(z, w[kx * 3 + ky].x, d.x);
                                d.y = fmaf(z, w[kx * 3 + ky].y, d.y);
                            }
            }
            // quad sum: lanes q ^ 1, then q ^ 2 (quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E); whole quads are active or not
#ifndef CRNN_PROBE_P2_NO_DPP
            // one v_add_f32_dpp per value and step (the builtin became v_mov_dpp x 2 + v_pk_add through two shared temporaries:
            // a serial chain).  All of step 1 first, then all of step 2: a DPP operand must not have been written by the
            // instruction just before it, and the compiler does not look into the asm for that.
#pragma unroll
            for (int k = 0; k < 49; ++k) {
                asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(da[k].x));
                asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(da[k].y));
            }
            if (kSplit2 == 4) {
                asm volatile("s_nop 1");
#pragma unroll
                for (int k = 0; k < 49; ++k) {
                    asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(da[k].x));
                    asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(da[k].y));
                }
            }
#endif
#ifdef CRNN_PROBE_P2_NO_MASK
            if (q2 == 0 && da[0].x == 12345.678f) {
#else
            if (q2 == 0) {
#endif
                // ReLU mask of both channels as bits (vector loads first), then selects and 16-byte stores of two positions each
                const float4 *act = (const float4 *)__builtin_assume_aligned(s_a1 + (size_t)unit2 * 2 * kA1Stride, 16);  // (x, y) of 2 positions
                unsigned long long lx = 0, ly = 0;
#pragma unroll
                for (int j = 0; j < 25; ++j) {
                    const float4 a = act[j];
                    lx |= (unsigned long long)((a.x > 0.0f ? 1u : 0u) | (a.z > 0.0f ? 2u : 0u)) << (2 * j);
                    ly |= (unsigned long long)((a.y > 0.0f ? 1u : 0u) | (a.w > 0.0f ? 2u : 0u)) << (2 * j);
                }
                float4 *dst4 = (float4 *)__builtin_assume_aligned(s_dz1 + (size_t)unit2 * kDz1Unit, 16);
                float sx = 0.0f, sy = 0.0f;
#pragma unroll
                for (int j = 0; j < 25; ++j) {  // positions 2j, 2j + 1 (position 49 does not exist: zeros)
                    float4 o;
                    o.x = (lx >> (2 * j)) & 1 ? da[2 * j].x : 0.0f;
                    o.y = (ly >> (2 * j)) & 1 ? da[2 * j].y : 0.0f;
                    o.z = (2 * j + 1 < 49 && ((lx >> (2 * j + 1)) & 1)) ? da[2 * j + 1 < 49 ? 2 * j + 1 : 0].x : 0.0f;
                    o.w = (2 * j + 1 < 49 && ((ly >> (2 * j + 1)) & 1)) ? da[2 * j + 1 < 49 ? 2 * j + 1 : 0].y : 0.0f;
                    dst4[j] = o;
                    sx += o.x + o.z;
                    sy += o.y + o.w;
                }
                accB1 += sx;   // db1[2 cp]     += sum of dz1[r2][2 cp][.]
                accB1y += sy;  // db1[2 cp + 1]
            }
        }
        __syncthreads();
        fetch(blk + 1);  // lands while P3 runs; parked after the next barrier
        // ---- P3: dW1[c1][c0][kx][0..2] over this thread's row slice (and db1)
#ifdef CRNN_PROBE_SKIP_P3
        if (false) {
#else
        if (p3_on) {
#endif
            for (int rr = s3; rr < rv; rr += G::RS3) {
                // one 7-wide line of dz1 at a time (the whole plane in registers next to the prefetched block went over the
                // register file); BOTH channels of the pair as P2 left them interleaved: packed FMAs on natural pairs again
                const float2 *pd = (const float2 *)(s_dz1 + (size_t)(rr * (OD / 2) + cp_3) * kDz1Unit);
                const float *in = s_in + rr * 244 + c0_3 * 81 + kx_3 * 9;
#pragma unroll
                for (int x = 0; x < 7; ++x) {
                    float2 dz[7];
                    float v[9];
#pragma unroll
                    for (int y = 0; y < 7; ++y) dz[y] = pd[x * 7 + y];
#pragma unroll
                    for (int y = 0; y < 9; ++y) v[y] = in[x * 9 + y];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int y = 0; y < 7; ++y) {
                            // an explicit 2-vector fma: left to itself the vectoriser pairs neighbouring taps here, not the two
                            // channels, and pays one register move per multiply-add for it
                            typedef float f32x2 __attribute__((ext_vector_type(2)));
                            const f32x2 r = __builtin_elementwise_fma((f32x2){dz[y].x, dz[y].y}, (f32x2){v[y + ky], v[y + ky]},
                                                                      (f32x2){accW1[ky].x, accW1[ky].y});
                            accW1[ky] = make_float2(r.x, r.y);
                        }
                }
            }
        }
    }
    // bias gradients: thread (br, bc) holds db2's sum of its row slot, lane 0 of quad (r2, cp) db1's of its two channels; the RBB
    // slots of a channel meet in LDS and are added in a fixed order
    __syncthreads();
    if (tid < G::RBB * OD) s_dz2[OD * 16 + bc * 16 + br] = accB2;
    if (q2 == 0 && unit2 < G::RBB * (OD / 2)) {
        s_dz2[(2 * cp2) * 16 + r2] = accB1;
        s_dz2[(2 * cp2 + 1) * 16 + r2] = accB1y;
    }
    __syncthreads();
    float *pp = part + (size_t)blockIdx.x * G::PART;
#pragma unroll
    for (int k = 0; k < 9; ++k) { pp[tid * 18 + 2 * k] = accW2[k].x; pp[tid * 18 + 2 * k + 1] = accW2[k].y; }
    if (tid < OD) {
        float t = 0.0f;
        for (int k = 0; k < G::RBB; ++k) t += s_dz2[OD * 16 + tid * 16 + k];
        pp[kBlock * 18 + tid] = t;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { pp[kBlock * 18 + OD + tid * 6 + k] = accW1[k].x; pp[kBlock * 18 + OD + tid * 6 + 3 + k] = accW1[k].y; }
    if (tid < OD) {
        float t = 0.0f;
        for (int k = 0; k < G::RBB; ++k) t += s_dz2[tid * 16 + k];
        pp[kBlock * 18 + OD + kBlock * 6 + tid] = t;
    }
}

// Sum of the partial vectors -> the four gradient tensors (<= 256 x ~11k floats).  64 outputs per 1024-thread
// workgroup: thread (ty, tx) adds every 16th partial vector of output tx (independent loads in flight), the 16
// sub-sums meet in LDS in a fixed order (deterministic).
constexpr int kRedY = 16;
template <int OD>
__global__ __launch_bounds__(64 * kRedY) void k_conv9_bwd_reduce(const float *__restrict__ part, int n_part, float *__restrict__ grads) {
    using G = GeoB<OD>;
    __shared__ float s_sum[kRedY][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    const int n2 = OD * OD * 9, nb = OD, n1 = OD * 27;
    float acc = 0.0f;
    if (i < n2 + nb + n1 + nb) {
        // up to 8 source slots of output i inside one partial vector (offsets), all partial vectors share them
        int off[8], cnt = 0;
        if (i < n2) {                                        // dW2[c2][c1][tap]: unit (c2, c1 / 2), component c1 & 1
            const int pair = i / 9, tap = i - pair * 9, c2 = pair / OD, c1 = pair - c2 * OD;
            const int unit = c2 * (OD / 2) + (c1 >> 1), comp = 2 * tap + (c1 & 1);
            if (unit < G::NA) off[cnt++] = unit * 18 + comp;
            else
                for (int sl = 0; sl < G::XSLICES && sl < 8; ++sl) off[cnt++] = (G::NA + sl * G::NB + (unit - G::NA)) * 18 + comp;
        } else if (i < n2 + nb) {                            // db2
            off[cnt++] = kBlock * 18 + (i - n2);
        } else if (i < n2 + nb + n1) {                       // dW1[c1][c0][kx][ky]: thread item = ((c1 / 2) * 3 + c0) * 3 + kx
            const int jx = i - n2 - nb, c1 = jx / 27, r = jx - c1 * 27, ky = r % 3;   // r = (c0 * 3 + kx) * 3 + ky
            const int item = (c1 >> 1) * 9 + r / 3;
            for (int sl = 0; sl < G::RS3; ++sl) off[cnt++] = kBlock * 18 + OD + (sl * G::ITEMS3 + item) * 6 + (c1 & 1) * 3 + ky;
        } else {                                             // db1
            off[cnt++] = kBlock * 18 + OD + kBlock * 6 + (i - n2 - nb - n1);
        }
        for (int k = 0; k < cnt; ++k) {
            float a0 = 0.0f, a1 = 0.0f;
            int b = ty;
            for (; b + kRedY < n_part; b += 2 * kRedY) { a0 += part[(size_t)b * G::PART + off[k]]; a1 += part[(size_t)(b + kRedY) * G::PART + off[k]]; }
            if (b < n_part) a0 += part[(size_t)b * G::PART + off[k]];
            acc += a0 + a1;
        }
    }
    s_sum[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && i < n2 + nb + n1 + nb) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < kRedY; ++k) t += s_sum[k][tx];
        grads[i] = t;
    }
}

// ---- backward of the vector branch relu(mlp1([dir_x, dir_y, last-action one-hot])) (network/base_net.py:66): dW [10][nin], db [10]
// from the gradient / the forward's output columns of the branch and the int8 inputs.  Thread = row (grid-stride), 10 * nin + 10
// sums in registers, added over the workgroup through LDS in a fixed order, one partial vector per workgroup; the last launch adds
// the partial vectors (fixed order: deterministic).  81 920 rows: 7 MB of reads, two launches, against eleven torch launches.
constexpr int kMlpBlock = 256, kMlpOut = 10, kMlpMaxIn = 18, kMlpMaxParts = 256;
__global__ __launch_bounds__(kMlpBlock) void k_mlp_bwd(const int8_t *__restrict__ obs, long obs_stride, int dir_off,
                                                       const int8_t *__restrict__ onehot, int n_actions, long rows,
                                                       const float *__restrict__ x, long x_stride, const float *__restrict__ g, long g_stride,
                                                       int col0, float *__restrict__ part) {
    __shared__ float s_red[kMlpBlock / 64][kMlpOut * (kMlpMaxIn + 1)];
    float acc[kMlpOut][kMlpMaxIn + 1];   // [o][k < nin] = dW, [o][kMlpMaxIn] = db
#pragma unroll
    for (int o = 0; o < kMlpOut; ++o)
#pragma unroll
        for (int k = 0; k <= kMlpMaxIn; ++k) acc[o][k] = 0.0f;
    for (long r = (long)blockIdx.x * kMlpBlock + threadIdx.x; r < rows; r += (long)gridDim.x * kMlpBlock) {
        float v[kMlpMaxIn];
        v[0] = (float)obs[r * obs_stride + dir_off];
        v[1] = (float)obs[r * obs_stride + dir_off + 1];
#pragma unroll
        for (int k = 0; k < kMlpMaxIn - 2; ++k) v[2 + k] = k < n_actions ? (float)onehot[r * n_actions + k] : 0.0f;
#pragma unroll
        for (int o = 0; o < kMlpOut; ++o) {
            const float gz = x[r * x_stride + col0 + o] > 0.0f ? g[r * g_stride + col0 + o] : 0.0f;
#pragma unroll
            for (int k = 0; k < kMlpMaxIn; ++k) acc[o][k] = fmaf(gz, v[k], acc[o][k]);
            acc[o][kMlpMaxIn] += gz;
        }
    }
    // wave sums (shuffles), then the four waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 0; o < kMlpOut; ++o)
#pragma unroll
        for (int k = 0; k <= kMlpMaxIn; ++k) {
            float t = acc[o][k];
            for (int d = 32; d > 0; d >>= 1) t += __shfl_xor(t, d);
            if (lane == 0) s_red[wave][o * (kMlpMaxIn + 1) + k] = t;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < kMlpOut * (kMlpMaxIn + 1); i += kMlpBlock) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < kMlpBlock / 64; ++w) t += s_red[w][i];
        part[(size_t)blockIdx.x * kMlpOut * (kMlpMaxIn + 1) + i] = t;
    }
}
__global__ __launch_bounds__(256) void k_mlp_bwd_reduce(const float *__restrict__ part, int n_part, int nin, float *__restrict__ dw, float *__restrict__ db) {
    const int i = threadIdx.x;   // one thread per output: 10 * nin weights, then 10 biases
    if (i >= kMlpOut * nin + kMlpOut) return;
    const int o = i < kMlpOut * nin ? i / nin : i - kMlpOut * nin, k = i < kMlpOut * nin ? i - o * nin : kMlpMaxIn;
    float t = 0.0f;
    for (int b = 0; b < n_part; ++b) t += part[(size_t)b * kMlpOut * (kMlpMaxIn + 1) + o * (kMlpMaxIn + 1) + k];
    if (i < kMlpOut * nin) dw[i] = t;
    else db[o] = t;
}

thread_local int g_last_hip = 0;

// The dynamic-LDS limit is an attribute of the function ON ONE DEVICE: remember per device whether it has been raised
// (a process may drive several GPUs; one rank per GPU is the normal case).  Thread-compatible like the rest of the ABI.
struct PerDeviceOnce {
    bool done[64] = {};
    static int device() {
        int dev = 0;
        (void)hipGetDevice(&dev);
        return dev;
    }
    bool need() const {
        const int dev = device();
        return dev < 0 || dev >= 64 || !done[dev];
    }
    void mark() {
        const int dev = device();
        if (dev >= 0 && dev < 64) done[dev] = true;
    }
};

struct LiveRows { const int32_t *chips; const int32_t *n; int rows_per_chip; };

template <int OD, int RBV>
int launch_rb(const int8_t *obs, long obs_stride, long rows, const float *w1, const float *b1, const float *w2, const float *b2,
              float *out, long out_stride, int out_cols, const int8_t *onehot, int n_actions, const float *mlp_w, const float *mlp_b, hipStream_t s,
              LiveRows live = LiveRows{nullptr, nullptr, 1}) {
    using GM = crnn_mfma::GeoM<OD, RBV>;
    const size_t lds = GM::LDS_FLOATS * sizeof(float);
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void *)crnn_mfma::k_conv9_mfma<OD, RBV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { g_last_hip = (int)e; return CRNN_ERR_HIP; }
        attr_set.mark();
    }
    const long n_blocks = (rows + GM::RB - 1) / GM::RB;
    // persistent: as many workgroups as the CUs hold at once (LDS-limited: one at 16 / 12 rows, two at 8), weights stay in registers
    const long resident = 256L * (long)((size_t)160 * 1024 / lds);
    const int grid = (int)(n_blocks < resident ? n_blocks : resident);
    (void)hipGetLastError();
    hipLaunchKernelGGL((crnn_mfma::k_conv9_mfma<OD, RBV>), dim3(grid), dim3(crnn_mfma::kBlockM), lds, s, obs, obs_stride, rows, w1, b1, w2, b2,
                       out, out_stride, out_cols, onehot, n_actions, mlp_w, mlp_b, live.chips, live.n, live.rows_per_chip);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip = (int)e; return CRNN_ERR_HIP; }
    return CRNN_OK;
}

template <int OD>
int launch(const int8_t *obs, long obs_stride, long rows, const float *w1, const float *b1, const float *w2, const float *b2,
           float *out, long out_stride, int out_cols, const int8_t *onehot, int n_actions, const float *mlp_w, const float *mlp_b, hipStream_t s,
           LiveRows live = LiveRows{nullptr, nullptr, 1}) {
    return launch_rb<OD, 0>(obs, obs_stride, rows, w1, b1, w2, b2, out, out_stride, out_cols, onehot, n_actions, mlp_w, mlp_b, s, live);
}

template <int OD>
int launch19(const int8_t *obs, long obs_stride, long rows, const float *w1, const float *b1, const float *w3, const float *b3,
             float *out, long out_stride, int out_cols, const int8_t *onehot, int n_actions, const float *mlp_w, const float *mlp_b, hipStream_t s) {
    using GM = crnn_mfma19::Geo<OD>;
    const size_t lds = GM::LDS_FLOATS * sizeof(float);
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void *)crnn_mfma19::k_conv19_mfma<OD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { g_last_hip = (int)e; return CRNN_ERR_HIP; }
        attr_set.mark();
    }
    const long n_blocks = (rows + GM::RB - 1) / GM::RB;
    const int grid = (int)(n_blocks < 256 ? n_blocks : 256);  // persistent: one workgroup per CU keeps the weights in registers
    (void)hipGetLastError();
    hipLaunchKernelGGL((crnn_mfma19::k_conv19_mfma<OD>), dim3(grid), dim3(crnn_mfma19::kBlockM), lds, s, obs, obs_stride, rows, w1, b1, w3,
                       b3, out, out_stride, out_cols, onehot, n_actions, mlp_w, mlp_b);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip = (int)e; return CRNN_ERR_HIP; }
    return CRNN_OK;
}

template <int OD>
int launch_bwd(const int8_t *obs, long obs_stride, long rows, const float *a2, long a2_stride,
               const float *g, long g_stride, const float *w2, float *part, int grid, float *grads, const float *w1, const float *b1,
               hipStream_t s) {
    using G = GeoB<OD>;
    const size_t lds = G::LDS_FLOATS * sizeof(float);
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void *)k_conv9_bwd<OD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { g_last_hip = (int)e; return CRNN_ERR_HIP; }
        attr_set.mark();
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL((k_conv9_bwd<OD>), dim3(grid), dim3(kBlock), lds, s, obs, obs_stride, rows, a2, a2_stride, g, g_stride,
                       w2, part, w1, b1);
    const int n_out = OD * OD * 9 + OD + OD * 27 + OD;
    hipLaunchKernelGGL((k_conv9_bwd_reduce<OD>), dim3((n_out + 63) / 64), dim3(64 * kRedY), 0, s, part, grid, grads);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip = (int)e; return CRNN_ERR_HIP; }
    return CRNN_OK;
}

template <int OD>
int launch_bwd19(const int8_t *obs, long obs_stride, long rows, const float *a3, long a3_stride, const float *g, long g_stride,
                 const float *w1, const float *b1, const float *w3, const float *b3, float *part, int n_part, float *grads, hipStream_t s) {
    using G = crnn_bwd19::GeoB19<OD>;
    const size_t lds = G::LDS_FLOATS * sizeof(float);
    static PerDeviceOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void *)crnn_bwd19::k_conv19_bwd<OD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { g_last_hip = (int)e; return CRNN_ERR_HIP; }
        attr_set.mark();
    }
    const long n_blocks = (rows + G::RBB - 1) / G::RBB;
    const int grid = (int)(n_blocks < n_part ? n_blocks : n_part);  // one persistent workgroup per partial vector (<= 256: one per CU)
    (void)hipGetLastError();
    hipLaunchKernelGGL((crnn_bwd19::k_conv19_bwd<OD>), dim3(grid), dim3(crnn_bwd19::kBlockB), lds, s, obs, obs_stride, rows, a3, a3_stride, g,
                       g_stride, w1, b1, w3, b3, part);
    hipLaunchKernelGGL((crnn_bwd19::k_conv19_bwd_reduce<OD>), dim3((G::GRADS + 63) / 64), dim3(64 * crnn_bwd19::kRedY19), 0, s, part, grid, grads);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip = (int)e; return CRNN_ERR_HIP; }
    return CRNN_OK;
}

}  // namespace

extern "C" {

int crnn_conv9_forward(const int8_t *d_obs, int64_t obs_stride, int64_t rows, const float *d_w1, const float *d_b1,
                       const float *d_w2, const float *d_b2, int od, float *d_out, int64_t out_stride, void *stream) {
    if (!d_obs || !d_w1 || !d_b1 || !d_w2 || !d_b2 || !d_out || rows < 0 || obs_stride < 243 || out_stride < od * 25)
        return CRNN_ERR_BAD_ARG;
    if (rows == 0) return CRNN_OK;
    if (od == 24) return launch<24>(d_obs, obs_stride, rows, d_w1, d_b1, d_w2, d_b2, d_out, out_stride, 0, nullptr, 0, nullptr, nullptr, (hipStream_t)stream);
    if (od == 32) return launch<32>(d_obs, obs_stride, rows, d_w1, d_b1, d_w2, d_b2, d_out, out_stride, 0, nullptr, 0, nullptr, nullptr, (hipStream_t)stream);
    return CRNN_ERR_UNSUPPORTED;
}

int crnn_front9_forward(const int8_t *d_obs, int64_t obs_stride, const int8_t *d_onehot, int n_actions, int64_t rows,
                        const float *d_w1, const float *d_b1, const float *d_w2, const float *d_b2, const float *d_mlp_w,
                        const float *d_mlp_b, int od, float *d_out, int64_t out_stride, int out_cols, void *stream) {
    if (!d_obs || !d_w1 || !d_b1 || !d_w2 || !d_b2 || !d_mlp_w || !d_mlp_b || !d_out || rows < 0 || obs_stride < 245 ||
        out_stride < od * 25 + 10 || n_actions < 0 || n_actions > 16)
        return CRNN_ERR_BAD_ARG;
    if (od != 24 && od != 32) return CRNN_ERR_UNSUPPORTED;
    if (out_cols != 0 && (out_cols < od * 25 + 10 || out_cols > crnn_front_padded_cols(od) || out_cols > out_stride)) return CRNN_ERR_BAD_ARG;
    if (rows == 0) return CRNN_OK;
    if (od == 24) return launch<24>(d_obs, obs_stride, rows, d_w1, d_b1, d_w2, d_b2, d_out, out_stride, out_cols, d_onehot, n_actions, d_mlp_w, d_mlp_b, (hipStream_t)stream);
    return launch<32>(d_obs, obs_stride, rows, d_w1, d_b1, d_w2, d_b2, d_out, out_stride, out_cols, d_onehot, n_actions, d_mlp_w, d_mlp_b, (hipStream_t)stream);
}

int crnn_front9_forward_live(const int8_t *d_obs, int64_t obs_stride, const int8_t *d_onehot, int n_actions, int64_t rows,
                             const float *d_w1, const float *d_b1, const float *d_w2, const float *d_b2, const float *d_mlp_w,
                             const float *d_mlp_b, int od, float *d_out, int64_t out_stride, int out_cols, const int32_t *d_live_chips,
                             const int32_t *d_n_live, int rows_per_chip, void *stream) {
    if (!d_obs || !d_w1 || !d_b1 || !d_w2 || !d_b2 || !d_mlp_w || !d_mlp_b || !d_out || rows < 0 || obs_stride < 245 ||
        out_stride < od * 25 + 10 || n_actions < 0 || n_actions > 16 || !d_live_chips || !d_n_live || rows_per_chip < 1 ||
        rows_per_chip > 64 || rows % rows_per_chip != 0 || rows >= (1 << 25))
        return CRNN_ERR_BAD_ARG;
    if (od != 24 && od != 32) return CRNN_ERR_UNSUPPORTED;
    if (out_cols != 0 && (out_cols < od * 25 + 10 || out_cols > crnn_front_padded_cols(od) || out_cols > out_stride)) return CRNN_ERR_BAD_ARG;
    if (rows == 0) return CRNN_OK;
    const LiveRows live{d_live_chips, d_n_live, rows_per_chip};
    if (od == 24) return launch<24>(d_obs, obs_stride, rows, d_w1, d_b1, d_w2, d_b2, d_out, out_stride, out_cols, d_onehot, n_actions, d_mlp_w, d_mlp_b, (hipStream_t)stream, live);
    return launch<32>(d_obs, obs_stride, rows, d_w1, d_b1, d_w2, d_b2, d_out, out_stride, out_cols, d_onehot, n_actions, d_mlp_w, d_mlp_b, (hipStream_t)stream, live);
}

int crnn_front_padded_cols(int od) { return od == 24 ? crnn_mfma::GeoM<24>::PAD_COLS : od == 32 ? crnn_mfma::GeoM<32>::PAD_COLS : CRNN_ERR_UNSUPPORTED; }

int crnn_front19_forward(const int8_t *d_obs, int64_t obs_stride, const int8_t *d_onehot, int n_actions, int64_t rows,
                         const float *d_w1, const float *d_b1, const float *d_w3, const float *d_b3, const float *d_mlp_w,
                         const float *d_mlp_b, int od, float *d_out, int64_t out_stride, int out_cols, void *stream) {
    const bool vec = d_mlp_w != nullptr;
    const int n_feat = od * 25 + (vec ? 10 : 0);
    if (!d_obs || !d_w1 || !d_b1 || !d_w3 || !d_b3 || !d_out || rows < 0 || obs_stride < 3 * 19 * 19 + (vec ? 2 : 0) ||
        out_stride < n_feat || n_actions < 0 || n_actions > 16 || (vec && !d_mlp_b))
        return CRNN_ERR_BAD_ARG;
    if (od != 24 && od != 32) return CRNN_ERR_UNSUPPORTED;
    if (out_cols != 0 && (out_cols < n_feat || out_cols > crnn_front_padded_cols(od) || out_cols > out_stride)) return CRNN_ERR_BAD_ARG;
    if (rows == 0) return CRNN_OK;
    if (od == 24) return launch19<24>(d_obs, obs_stride, rows, d_w1, d_b1, d_w3, d_b3, d_out, out_stride, out_cols, d_onehot, n_actions, d_mlp_w, d_mlp_b, (hipStream_t)stream);
    return launch19<32>(d_obs, obs_stride, rows, d_w1, d_b1, d_w3, d_b3, d_out, out_stride, out_cols, d_onehot, n_actions, d_mlp_w, d_mlp_b, (hipStream_t)stream);
}

int crnn_conv9_backward_parts(int od) { return od == 24 ? GeoB<24>::PART : od == 32 ? GeoB<32>::PART : CRNN_ERR_UNSUPPORTED; }

int crnn_conv9_backward(const int8_t *d_obs, int64_t obs_stride, int64_t rows, const float *d_out, int64_t out_stride,
                        const float *d_grad_out, int64_t grad_stride, const float *d_w1, const float *d_b1, const float *d_w2,
                        int od, float *d_part, int n_part, float *d_grads, void *stream) {
    if (!d_obs || !d_out || !d_grad_out || !d_w1 || !d_b1 || !d_w2 || !d_part || !d_grads || rows <= 0 || n_part < 1 || n_part > 256)
        return CRNN_ERR_BAD_ARG;
    if (od == 24) return launch_bwd<24>(d_obs, obs_stride, rows, d_out, out_stride, d_grad_out, grad_stride, d_w2, d_part, n_part, d_grads, d_w1, d_b1, (hipStream_t)stream);
    if (od == 32) return launch_bwd<32>(d_obs, obs_stride, rows, d_out, out_stride, d_grad_out, grad_stride, d_w2, d_part, n_part, d_grads, d_w1, d_b1, (hipStream_t)stream);
    return CRNN_ERR_UNSUPPORTED;
}

int crnn_mlp_backward_parts(void) { return kMlpMaxParts * kMlpOut * (kMlpMaxIn + 1); }

int crnn_mlp_backward(const int8_t *d_obs, int64_t obs_stride, int dir_offset, const int8_t *d_onehot, int n_actions, int64_t rows,
                      const float *d_out, int64_t out_stride, const float *d_grad_out, int64_t grad_stride, int col0, float *d_part,
                      float *d_grad_w, float *d_grad_b, void *stream) {
    if (!d_obs || !d_onehot || !d_out || !d_grad_out || !d_part || !d_grad_w || !d_grad_b || rows <= 0 || n_actions < 0 ||
        n_actions > kMlpMaxIn - 2 || dir_offset < 0 || obs_stride < dir_offset + 2 || col0 < 0 || out_stride < col0 + kMlpOut ||
        grad_stride < col0 + kMlpOut)
        return CRNN_ERR_BAD_ARG;
    const long want = (rows + kMlpBlock - 1) / kMlpBlock;
    const int grid = (int)(want < kMlpMaxParts ? want : kMlpMaxParts);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_mlp_bwd, dim3(grid), dim3(kMlpBlock), 0, (hipStream_t)stream, d_obs, (long)obs_stride, dir_offset, d_onehot, n_actions,
                       (long)rows, d_out, (long)out_stride, d_grad_out, (long)grad_stride, col0, d_part);
    hipLaunchKernelGGL(k_mlp_bwd_reduce, dim3(1), dim3(256), 0, (hipStream_t)stream, d_part, grid, 2 + n_actions, d_grad_w, d_grad_b);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip = (int)e; return CRNN_ERR_HIP; }
    return CRNN_OK;
}

int crnn_conv19_backward_parts(int od) { return od == 24 ? crnn_bwd19::GeoB19<24>::PART : od == 32 ? crnn_bwd19::GeoB19<32>::PART : CRNN_ERR_UNSUPPORTED; }

int crnn_conv19_backward(const int8_t *d_obs, int64_t obs_stride, int64_t rows, const float *d_out, int64_t out_stride,
                         const float *d_grad_out, int64_t grad_stride, const float *d_w1, const float *d_b1, const float *d_w3,
                         const float *d_b3, int od, float *d_part, int n_part, float *d_grads, void *stream) {
    if (!d_obs || !d_out || !d_grad_out || !d_w1 || !d_b1 || !d_w3 || !d_b3 || !d_part || !d_grads || rows <= 0 || n_part < 1 ||
        n_part > 256 || obs_stride < 3 * 19 * 19 || out_stride < od * 25 || grad_stride < od * 25)
        return CRNN_ERR_BAD_ARG;
    if (od == 24) return launch_bwd19<24>(d_obs, obs_stride, rows, d_out, out_stride, d_grad_out, grad_stride, d_w1, d_b1, d_w3, d_b3, d_part, n_part, d_grads, (hipStream_t)stream);
    if (od == 32) return launch_bwd19<32>(d_obs, obs_stride, rows, d_out, out_stride, d_grad_out, grad_stride, d_w1, d_b1, d_w3, d_b3, d_part, n_part, d_grads, (hipStream_t)stream);
    return CRNN_ERR_UNSUPPORTED;
}

int crnn_last_hip_error(void) { return g_last_hip; }

}  // extern "C"
