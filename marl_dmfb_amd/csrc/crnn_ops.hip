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

namespace {

constexpr int kBlock = 512;  // 8 waves: one workgroup per CU keeps ONE copy of the weights in LDS for 2 waves per SIMD
constexpr int kA1Stride = 52;  // 49 conv1 activations per (row, channel), padded to a 16-byte multiple

constexpr int kLdsBudget = 158 * 1024;  // of the CU's 160 KiB

template <int OD> struct Geo {
    static constexpr int ROW_BYTES = (OD * kA1Stride + 4 + 244) * 4;               // a1 row + input row
    static constexpr int FIXED_BYTES = (OD * OD * 9 + OD * 27 + 2 * OD) * 4;       // weights + biases
    static constexpr int RB_LDS = (kLdsBudget - FIXED_BYTES) / ROW_BYTES;
    static constexpr int RB = (kBlock / OD) < RB_LDS ? (kBlock / OD) : RB_LDS;      // rows per iteration (24 -> 21, 32 -> 15)
    static constexpr int ROW_A1 = OD * kA1Stride + 4;   // +4 floats: rows start on different banks
    static constexpr int IN_STRIDE = 244;               // 243 pixels (+1)
    static constexpr size_t LDS_FLOATS = (size_t)OD * OD * 9 + OD * 27 + 2 * OD + (size_t)RB * IN_STRIDE + (size_t)RB * ROW_A1;
};

template <int OD>
__global__ __launch_bounds__(kBlock) void k_conv9(const int8_t *__restrict__ obs, long obs_stride, long rows,
                                                  const float *__restrict__ w1, const float *__restrict__ b1,
                                                  const float *__restrict__ w2, const float *__restrict__ b2,
                                                  float *__restrict__ out, long out_stride,
                                                  const int8_t *__restrict__ onehot, int n_actions,
                                                  const float *__restrict__ mlp_w, const float *__restrict__ mlp_b) {
    using G = Geo<OD>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *s_w2 = lds;                         // [OD c1][9 tap][OD c2]: lanes (= c2) read consecutive banks
    float *s_w1 = s_w2 + OD * OD * 9;          // [OD][27]
    float *s_b1 = s_w1 + OD * 27;              // [OD]
    float *s_b2 = s_b1 + OD;                   // [OD]
    float *s_in = s_b2 + OD;                   // [RB][244]
    float *s_a1 = s_in + G::RB * G::IN_STRIDE; // [RB][ROW_A1], 16-byte aligned by construction
    const int tid = threadIdx.x;
    for (int i = tid; i < OD * OD * 9; i += kBlock) {  // global (c2, c1, tap) -> LDS (c1, tap, c2)
        const int c2 = i / (OD * 9), rem = i - c2 * OD * 9;
        s_w2[rem * OD + c2] = w2[i];
    }
    for (int i = tid; i < OD * 27; i += kBlock) s_w1[i] = w1[i];
    if (tid < OD) { s_b1[tid] = b1[tid]; s_b2[tid] = b2[tid]; }
    const int r = tid / OD, c = tid - r * OD;  // row in block, output channel
    const bool worker = r < G::RB;
    const long n_blocks = (rows + G::RB - 1) / G::RB;
    for (long blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const long row0 = blk * G::RB;
        const int rv = (int)min((long)G::RB, rows - row0);
        __syncthreads();  // previous iteration's readers of s_in / s_a1 are done (and weights are loaded)
        for (int i = tid; i < rv * 243; i += kBlock) {
            const int rr = i / 243, p = i - rr * 243;
            s_in[rr * G::IN_STRIDE + p] = (float)obs[(row0 + rr) * obs_stride + p];
        }
        __syncthreads();
        if (worker && r < rv) {
            // ---- conv1 + ReLU: out (7,7) for channel c of row r; input (3,9,9) in (c0,x,y) order
            const float *in = s_in + r * G::IN_STRIDE;
            float *a1 = s_a1 + r * G::ROW_A1 + c * kA1Stride;
            float wv[27];
#pragma unroll
            for (int k = 0; k < 27; ++k) wv[k] = s_w1[c * 27 + k];
            const float bias = s_b1[c];
#pragma unroll 1
            for (int x = 0; x < 7; ++x) {
                float acc[7];
#pragma unroll
                for (int y = 0; y < 7; ++y) acc[y] = bias;
#pragma unroll
                for (int c0 = 0; c0 < 3; ++c0)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float v[9];
#pragma unroll
                        for (int y = 0; y < 9; ++y) v[y] = in[c0 * 81 + (x + kx) * 9 + y];
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int y = 0; y < 7; ++y) acc[y] = fmaf(v[y + ky], wv[c0 * 9 + kx * 3 + ky], acc[y]);
                    }
#pragma unroll
                for (int y = 0; y < 7; ++y) a1[x * 7 + y] = fmaxf(acc[y], 0.0f);
            }
        }
        __syncthreads();
        if (worker && r < rv) {
            // ---- conv2 + ReLU: out (5,5) for channel c of row r over OD input channels of (7,7)
            float acc[25];
            const float bias = s_b2[c];
#pragma unroll
            for (int k = 0; k < 25; ++k) acc[k] = bias;
            const float *a1row = s_a1 + r * G::ROW_A1;
            const float *wrow = s_w2 + c;
#pragma unroll 2
            for (int c1 = 0; c1 < OD; ++c1) {
                float a[kA1Stride];
                const float4 *src = (const float4 *)__builtin_assume_aligned(a1row + c1 * kA1Stride, 16);  // ds_read_b128
#pragma unroll
                for (int q = 0; q < kA1Stride / 4; ++q) {
                    const float4 t = src[q];
                    a[4 * q] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w;
                }
                float w[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) w[k] = wrow[(c1 * 9 + k) * OD];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int x = 0; x < 5; ++x)
#pragma unroll
                            for (int y = 0; y < 5; ++y)
                                acc[x * 5 + y] = fmaf(a[(x + kx) * 7 + y + ky], w[kx * 3 + ky], acc[x * 5 + y]);
            }
            float *o = out + (row0 + r) * out_stride + c * 25;
#pragma unroll
            for (int k = 0; k < 25; ++k) o[k] = fmaxf(acc[k], 0.0f);
            if (mlp_w && c < 10) {  // vector branch: relu(mlp1([dir_x, dir_y, last-action one-hot])) (base_net.py:66)
                const int nin = 2 + n_actions;
                const int8_t *ob = obs + (row0 + r) * obs_stride;
                float v = mlp_b[c];
                v = fmaf((float)ob[243], mlp_w[c * nin], v);
                v = fmaf((float)ob[244], mlp_w[c * nin + 1], v);
                if (onehot)
                    for (int k = 0; k < n_actions; ++k) v = fmaf((float)onehot[(row0 + r) * n_actions + k], mlp_w[c * nin + 2 + k], v);
                out[(row0 + r) * out_stride + OD * 25 + c] = fmaxf(v, 0.0f);
            }
        }
    }
}

thread_local int g_last_hip = 0;

template <int OD>
int launch(const int8_t *obs, long obs_stride, long rows, const float *w1, const float *b1, const float *w2,
           const float *b2, float *out, long out_stride, const int8_t *onehot, int n_actions, const float *mlp_w,
           const float *mlp_b, hipStream_t s) {
    using G = Geo<OD>;
    const size_t lds = G::LDS_FLOATS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_conv9<OD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { g_last_hip = (int)e; return CRNN_ERR_HIP; }
        attr_set = true;
    }
    const long n_blocks = (rows + G::RB - 1) / G::RB;
    const int grid = (int)(n_blocks < 256 ? n_blocks : 256);  // persistent: one 8-wave workgroup per CU keeps the weights resident
    (void)hipGetLastError();
    hipLaunchKernelGGL((k_conv9<OD>), dim3(grid), dim3(kBlock), lds, s, obs, obs_stride, rows, w1, b1, w2, b2, out, out_stride, onehot,
                       n_actions, mlp_w, mlp_b);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last_hip = (int)e;
        if (getenv("DMFB_VEC_DEBUG")) fprintf(stderr, "crnn_ops: launch failed: %s\n", hipGetErrorString(e));
        return CRNN_ERR_HIP;
    }
    return CRNN_OK;
}

}  // namespace

extern "C" {

int crnn_conv9_forward(const int8_t *d_obs, int64_t obs_stride, int64_t rows, const float *d_w1, const float *d_b1,
                       const float *d_w2, const float *d_b2, int od, float *d_out, int64_t out_stride, void *stream) {
    if (!d_obs || !d_w1 || !d_b1 || !d_w2 || !d_b2 || !d_out || rows < 0 || obs_stride < 243 || out_stride < od * 25)
        return CRNN_ERR_BAD_ARG;
    if (rows == 0) return CRNN_OK;
    if (od == 24) return launch<24>(d_obs, obs_stride, rows, d_w1, d_b1, d_w2, d_b2, d_out, out_stride, nullptr, 0, nullptr, nullptr, (hipStream_t)stream);
    if (od == 32) return launch<32>(d_obs, obs_stride, rows, d_w1, d_b1, d_w2, d_b2, d_out, out_stride, nullptr, 0, nullptr, nullptr, (hipStream_t)stream);
    return CRNN_ERR_UNSUPPORTED;
}

int crnn_front9_forward(const int8_t *d_obs, int64_t obs_stride, const int8_t *d_onehot, int n_actions, int64_t rows,
                        const float *d_w1, const float *d_b1, const float *d_w2, const float *d_b2, const float *d_mlp_w,
                        const float *d_mlp_b, int od, float *d_out, int64_t out_stride, void *stream) {
    if (!d_obs || !d_w1 || !d_b1 || !d_w2 || !d_b2 || !d_mlp_w || !d_mlp_b || !d_out || rows < 0 || obs_stride < 245 ||
        out_stride < od * 25 + 10 || n_actions < 0 || n_actions > 16)
        return CRNN_ERR_BAD_ARG;
    if (rows == 0) return CRNN_OK;
    if (od == 24) return launch<24>(d_obs, obs_stride, rows, d_w1, d_b1, d_w2, d_b2, d_out, out_stride, d_onehot, n_actions, d_mlp_w, d_mlp_b, (hipStream_t)stream);
    if (od == 32) return launch<32>(d_obs, obs_stride, rows, d_w1, d_b1, d_w2, d_b2, d_out, out_stride, d_onehot, n_actions, d_mlp_w, d_mlp_b, (hipStream_t)stream);
    return CRNN_ERR_UNSUPPORTED;
}

int crnn_last_hip_error(void) { return g_last_hip; }

}  // extern "C"
