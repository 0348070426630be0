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

}  // namespace

extern "C" {

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

int vdn_last_hip_error(void) { return g_last; }

}  // extern "C"
