// vdn_ops.hip -- the TD-error block of VDN.learn (policy/vdn.py:104-123) as one kernel each way.  See include/vdn_ops.h.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/vdn_ops.h"

namespace {

thread_local int g_last = 0;

// one thread per (episode b, step t)
__global__ __launch_bounds__(256) void k_td_forward(const float *__restrict__ qe, const float *__restrict__ qt, const int8_t *__restrict__ u,
                                                    const float *__restrict__ r, const int8_t *__restrict__ avail,
                                                    const uint8_t *__restrict__ term, const uint8_t *__restrict__ padded, int B, int T,
                                                    int Tl, int n, int A, float gamma, float *__restrict__ mtd, float *__restrict__ maskf,
                                                    int32_t *__restrict__ bad) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * T) return;
    const int b = idx / T, t = idx - b * T;
    const size_t ep = (size_t)b * Tl + t;            // slot in the chip-major episode tensors
    const size_t q0 = ((size_t)t * B + b) * n * A;   // row block in the time-major Q tensors
    float qe_tot = 0.0f, qt_tot = 0.0f;
    bool ok = true;
    for (int i = 0; i < n; ++i) {
        int a_taken = (int)u[ep * n + i];
        if ((unsigned)a_taken >= (unsigned)A) { ok = false; a_taken = 0; }  // torch.gather raises here; never read out of bounds
        qe_tot = qe_tot + qe[q0 + (size_t)i * A + a_taken];
        float m = -3.4e38f;
        for (int a = 0; a < A; ++a) {
            const float v = avail[(ep * n + i) * A + a] == 0 ? -9999999.0f : qt[q0 + (size_t)i * A + a];
            m = v > m ? v : m;
        }
        qt_tot = qt_tot + m;
    }
    const float not_term = 1.0f - (term[ep] ? 1.0f : 0.0f);
    const float target = r[ep] + (gamma * qt_tot) * not_term;
    const float mk = 1.0f - (padded[ep] ? 1.0f : 0.0f);
    mtd[idx] = ok ? mk * (target - qe_tot) : __builtin_nanf("");  // an action outside [0, A) poisons the loss (loud) ...
    maskf[idx] = mk;
    if (!ok && bad) atomicAdd(bad, 1);                            // ... and is counted for vdn_td_forward's caller
}

// one thread per (t, b, i) row of the time-major gradient
__global__ __launch_bounds__(256) void k_td_backward(const float *__restrict__ mtd, const float *__restrict__ maskf, const int8_t *__restrict__ u,
                                                     const float *__restrict__ g, int B, int T, int Tl, int n, int A,
                                                     float *__restrict__ gq) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= (long)T * B * n) return;
    const int i = (int)(row % n);
    const long tb = row / n;
    const int b = (int)(tb % B), t = (int)(tb / B);
    const int idx = b * T + t;
    const float d = -((2.0f * mtd[idx]) * maskf[idx]) * g[0];
    const int a_taken = (int)u[((size_t)b * Tl + t) * n + i];  // outside [0, A): mtd is NaN for the slot, so d is NaN: keep it visible
    const bool bad_a = (unsigned)a_taken >= (unsigned)A;
    for (int a = 0; a < A; ++a) gq[row * A + a] = (bad_a || a == a_taken) ? d : 0.0f;
}

// The same block on PACKED (episode, step) units (include/vdn_ops.h: vdn_td_forward_packed): unit j is step units[j] % t_limit of
// the episode in slot units[j] / t_limit of the replay tensors; its n rows of the Q tensors are rows j*n .. j*n + n - 1.
__global__ __launch_bounds__(256) void k_td_forward_packed(const float *__restrict__ qe, const float *__restrict__ qt, const int32_t *__restrict__ units,
                                                           int U, const int8_t *__restrict__ u, const float *__restrict__ r,
                                                           const int8_t *__restrict__ avail, const uint8_t *__restrict__ term,
                                                           const uint8_t *__restrict__ padded, int n, int A, float gamma,
                                                           float *__restrict__ mtd, float *__restrict__ maskf, int32_t *__restrict__ bad) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= U) return;
    const size_t ep = (size_t)units[j];
    const size_t q0 = (size_t)j * n * A;
    float qe_tot = 0.0f, qt_tot = 0.0f;
    bool ok = true;
    for (int i = 0; i < n; ++i) {
        int a_taken = (int)u[ep * n + i];
        if ((unsigned)a_taken >= (unsigned)A) { ok = false; a_taken = 0; }
        qe_tot = qe_tot + qe[q0 + (size_t)i * A + a_taken];
        float m = -3.4e38f;
        for (int a = 0; a < A; ++a) {
            const float v = avail[(ep * n + i) * A + a] == 0 ? -9999999.0f : qt[q0 + (size_t)i * A + a];
            m = v > m ? v : m;
        }
        qt_tot = qt_tot + m;
    }
    const float not_term = 1.0f - (term[ep] ? 1.0f : 0.0f);
    const float target = r[ep] + (gamma * qt_tot) * not_term;
    const float mk = 1.0f - (padded[ep] ? 1.0f : 0.0f);
    mtd[j] = ok ? mk * (target - qe_tot) : __builtin_nanf("");
    maskf[j] = mk;
    if (!ok && bad) atomicAdd(bad, 1);
}

__global__ __launch_bounds__(256) void k_td_backward_packed(const float *__restrict__ mtd, const float *__restrict__ maskf, const int32_t *__restrict__ units,
                                                            const int8_t *__restrict__ u, const float *__restrict__ g, long rows, int n, int A,
                                                            float *__restrict__ gq) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const int i = (int)(row % n);
    const long j = row / n;
    const float d = -((2.0f * mtd[j]) * maskf[j]) * g[0];
    const int a_taken = (int)u[(size_t)units[j] * n + i];
    const bool bad_a = (unsigned)a_taken >= (unsigned)A;
    for (int a = 0; a < A; ++a) gq[row * A + a] = (bad_a || a == a_taken) ? d : 0.0f;
}

// dst unit j = src unit units[j] + shift (zeros for j < zero_below): coalesced copies of whole units (n rows of the replay tensors)
template <typename V>
__global__ __launch_bounds__(256) void k_gather_units(const V *__restrict__ src, int unit_v, const int32_t *__restrict__ units, int U, int shift,
                                                      int zero_below, V *__restrict__ dst) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)U * unit_v) return;
    const int j = (int)(idx / unit_v), k = (int)(idx - (long)j * unit_v);
    dst[idx] = j < zero_below ? (V)0 : src[((size_t)units[j] + shift) * unit_v + k];
}

}  // namespace

namespace {

// ---- clip_grad_norm_ + Adam over a list of tensors (include/vdn_ops.h)
struct TensorList {
    float *p[VDN_MAX_TENSORS];
    float *g[VDN_MAX_TENSORS];
    float *m[VDN_MAX_TENSORS];
    float *v[VDN_MAX_TENSORS];
    long off[VDN_MAX_TENSORS + 1];  // prefix sums of the element counts
    int n;
};

// element i of the concatenation -> (tensor k, index inside it); i only grows along a thread's walk, so k is carried along
__device__ __forceinline__ void locate(const TensorList &tl, long i, int &k) {
    while (i >= tl.off[k + 1]) ++k;
}

__global__ __launch_bounds__(256) void k_sqnorm_partials(TensorList tl, float *__restrict__ partials, const float *__restrict__ grad_div) {
    __shared__ float s_w[4];
    const long total = tl.off[tl.n];
    const long per = (total + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = min(total, lo + per);
    float acc = 0.0f;
    int k = 0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        locate(tl, i, k);
        float g = tl.g[k][i - tl.off[k]];
        if (grad_div) g = g / grad_div[0];
        acc = fmaf(g, g, acc);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ __launch_bounds__(256) void k_clip_adam(TensorList tl, const float *__restrict__ partials, int n_partials, float max_norm,
                                                   float step_size, float omb1, float beta2, float omb2, float eps, float bc2_sqrt,
                                                   float *__restrict__ total_norm, const float *__restrict__ grad_div) {
    __shared__ float s_coef;
    if (threadIdx.x < 64) {  // every workgroup adds the partial sums in the same fixed order
        float t = 0.0f;
        for (int b = threadIdx.x; b < n_partials; b += 64) t += partials[b];
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (threadIdx.x == 0) {
            const float norm = sqrtf(t);
            // clamp(max_norm / (norm + 1e-6), max = 1) of clip_grad_norm_; a NaN norm stays NaN in every gradient, as in torch
            s_coef = norm != norm ? norm : fminf(max_norm / (norm + 1e-6f), 1.0f);
            if (blockIdx.x == 0) total_norm[0] = norm;
        }
    }
    __syncthreads();
    const float coef = s_coef;
    const long total = tl.off[tl.n];
    const long per = (total + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = min(total, lo + per);
    int k = 0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        locate(tl, i, k);
        const long j = i - tl.off[k];
        float g = tl.g[k][j];
        if (grad_div) g = g / grad_div[0];  // the gradient of the un-normalised loss / the (global) mask count
        g = g * coef;
        if (grad_div || coef != 1.0f) tl.g[k][j] = g;  // clip_grad_norm_ leaves the rescaled gradient in p.grad
        float m = tl.m[k][j], v = tl.v[k][j];
        m = m + omb1 * (g - m);
        v = beta2 * v + omb2 * g * g;
        tl.m[k][j] = m;
        tl.v[k][j] = v;
        // torch's element arithmetic (foreach and fused Adam alike): float operations, the Python-double scalars rounded to float
        // where they enter: denom = sqrt(v) / sqrt(bias_correction2) + eps;  p += -step_size * (m / denom)   (addcdiv)
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        tl.p[k][j] -= step_size * (m / denom);
    }
}

}  // namespace

extern "C" {

int vdn_clip_adam_step(int32_t n_tensors, float *const *params, float *const *grads, float *const *exp_avg,
                       float *const *exp_avg_sq, const int64_t *numel, float max_norm, double lr, double beta1, double beta2,
                       double eps, double bias_correction1, double bias_correction2, float *d_partials, float *d_total_norm,
                       const float *d_grad_div, void *stream) {
    if (n_tensors < 1 || n_tensors > VDN_MAX_TENSORS || !params || !grads || !exp_avg || !exp_avg_sq || !numel || !d_partials ||
        !d_total_norm || !(bias_correction1 > 0.0) || !(bias_correction2 > 0.0))
        return VDN_ERR_BAD_ARG;
    TensorList tl;
    tl.n = n_tensors;
    tl.off[0] = 0;
    for (int k = 0; k < VDN_MAX_TENSORS; ++k) {
        const bool on = k < n_tensors;
        if (on && (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k] || numel[k] < 1)) return VDN_ERR_BAD_ARG;
        tl.p[k] = on ? params[k] : nullptr;
        tl.g[k] = on ? grads[k] : nullptr;
        tl.m[k] = on ? exp_avg[k] : nullptr;
        tl.v[k] = on ? exp_avg_sq[k] : nullptr;
        tl.off[k + 1] = tl.off[k] + (on ? (long)numel[k] : 0);
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_sqnorm_partials, dim3(VDN_NORM_BLOCKS), dim3(256), 0, (hipStream_t)stream, tl, d_partials, d_grad_div);
    const long total = tl.off[n_tensors];
    const int blocks = (int)((total + 1023) / 1024 < 1024 ? (total + 1023) / 1024 : 1024);
    hipLaunchKernelGGL(k_clip_adam, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tl, d_partials, VDN_NORM_BLOCKS, max_norm,
                       (float)((double)lr / bias_correction1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)sqrt(bias_correction2), d_total_norm, d_grad_div);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last = (int)e; return VDN_ERR_HIP; }
    return VDN_OK;
}


int vdn_td_forward(const float *d_q_eval, const float *d_q_target, const int8_t *d_u, const float *d_r,
                   const int8_t *d_avail_next, const uint8_t *d_terminated, const uint8_t *d_padded, int32_t B, int32_t T,
                   int32_t t_limit, int32_t n_agents, int32_t n_actions, float gamma, float *d_mtd, float *d_mask, int32_t *d_bad_actions,
                   void *stream) {
    if (!d_q_eval || !d_q_target || !d_u || !d_r || !d_avail_next || !d_terminated || !d_padded || !d_mtd || !d_mask || B < 0 ||
        T < 1 || t_limit < T || n_agents < 1 || n_actions < 1 || n_actions > 127)
        return VDN_ERR_BAD_ARG;
    if (B == 0) return VDN_OK;
    (void)hipGetLastError();
    const int total = B * T;
    hipLaunchKernelGGL(k_td_forward, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_q_eval, d_q_target, d_u, d_r,
                       d_avail_next, d_terminated, d_padded, B, T, t_limit, n_agents, n_actions, gamma, d_mtd, d_mask, d_bad_actions);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last = (int)e; return VDN_ERR_HIP; }
    return VDN_OK;
}

int vdn_td_backward(const float *d_mtd, const float *d_mask, const int8_t *d_u, const float *d_grad_num, int32_t B, int32_t T,
                    int32_t t_limit, int32_t n_agents, int32_t n_actions, float *d_grad_q, void *stream) {
    if (!d_mtd || !d_mask || !d_u || !d_grad_num || !d_grad_q || B < 0 || T < 1 || t_limit < T || n_agents < 1 || n_actions < 1 ||
        n_actions > 127)
        return VDN_ERR_BAD_ARG;
    if (B == 0) return VDN_OK;
    (void)hipGetLastError();
    const long rows = (long)T * B * n_agents;
    hipLaunchKernelGGL(k_td_backward, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_mtd, d_mask, d_u,
                       d_grad_num, B, T, t_limit, n_agents, n_actions, d_grad_q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last = (int)e; return VDN_ERR_HIP; }
    return VDN_OK;
}

int vdn_td_forward_packed(const float *d_q_eval, const float *d_q_target, const int32_t *d_units, int32_t n_units, const int8_t *d_u,
                          const float *d_r, const int8_t *d_avail_next, const uint8_t *d_terminated, const uint8_t *d_padded,
                          int32_t n_agents, int32_t n_actions, float gamma, float *d_mtd, float *d_mask, int32_t *d_bad_actions,
                          void *stream) {
    if (!d_q_eval || !d_q_target || !d_units || !d_u || !d_r || !d_avail_next || !d_terminated || !d_padded || !d_mtd || !d_mask ||
        n_units < 0 || n_agents < 1 || n_actions < 1 || n_actions > 127)
        return VDN_ERR_BAD_ARG;
    if (n_units == 0) return VDN_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_td_forward_packed, dim3((unsigned)((n_units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_q_eval, d_q_target,
                       d_units, n_units, d_u, d_r, d_avail_next, d_terminated, d_padded, n_agents, n_actions, gamma, d_mtd, d_mask,
                       d_bad_actions);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last = (int)e; return VDN_ERR_HIP; }
    return VDN_OK;
}

int vdn_td_backward_packed(const float *d_mtd, const float *d_mask, const int32_t *d_units, int32_t n_units, const int8_t *d_u,
                           const float *d_grad_num, int32_t n_agents, int32_t n_actions, float *d_grad_q, void *stream) {
    if (!d_mtd || !d_mask || !d_units || !d_u || !d_grad_num || !d_grad_q || n_units < 0 || n_agents < 1 || n_actions < 1 || n_actions > 127)
        return VDN_ERR_BAD_ARG;
    if (n_units == 0) return VDN_OK;
    const long rows = (long)n_units * n_agents;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_td_backward_packed, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_mtd, d_mask, d_units,
                       d_u, d_grad_num, rows, n_agents, n_actions, d_grad_q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last = (int)e; return VDN_ERR_HIP; }
    return VDN_OK;
}

int vdn_gather_units(const void *d_src, int32_t unit_bytes, const int32_t *d_units, int32_t n_units, int32_t unit_shift,
                     int32_t zero_below, void *d_dst, void *stream) {
    if (!d_src || !d_units || !d_dst || unit_bytes < 1 || n_units < 0 || zero_below < 0) return VDN_ERR_BAD_ARG;
    if (n_units == 0) return VDN_OK;
    (void)hipGetLastError();
    const bool dw = unit_bytes % 4 == 0 && ((size_t)d_src | (size_t)d_dst) % 4 == 0;
    const int uv = dw ? unit_bytes / 4 : unit_bytes;
    const long total = (long)n_units * uv;
    if (dw)
        hipLaunchKernelGGL((k_gather_units<uint32_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const uint32_t *)d_src, uv, d_units, n_units, unit_shift, zero_below, (uint32_t *)d_dst);
    else
        hipLaunchKernelGGL((k_gather_units<uint8_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const uint8_t *)d_src, uv, d_units, n_units, unit_shift, zero_below, (uint8_t *)d_dst);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last = (int)e; return VDN_ERR_HIP; }
    return VDN_OK;
}

int vdn_last_hip_error(void) { return g_last; }

}  // extern "C"
