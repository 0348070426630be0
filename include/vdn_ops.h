/*
 * vdn_ops.h -- C ABI of the TD-error block of VDN.learn (reference policy/vdn.py:104-123), fused.
 *
 * Between the Q-networks and the backward pass the reference does, on (episodes B) x (steps T) x (agents n) x
 * (actions A) tensors:  q_evals = gather(q_evals, u);  q_targets[avail_u_next == 0] = -9999999;  q_targets = max_a;
 * q_total_* = VDNNet = sum over agents (network/vdn_net.py:5-10);  targets = r + gamma * q_total_target * (1 - terminated);
 * td_error = targets.detach() - q_total_eval;  mask = 1 - padded;  masked_td_error = mask * td_error;
 * loss = (masked_td_error ** 2).sum() / mask.sum().  As tensor ops that is ~30 launches of a few microseconds forward and
 * backward; here it is one launch each way (the two scalar sums stay with the caller).
 *
 * Conventions as dmfb_vec.h: plain C types, caller-owned DEVICE buffers, `stream` = hipStream_t as void*, asynchronous,
 * negative int error codes.  The Q tensors are TIME-MAJOR float32 [T][B][n][A] (what the GRU sequence kernels produce);
 * the episode tensors are the replay buffer's, chip-major with `t_limit` slots per episode:
 * u int8[B][t_limit][n][1], r float32[B][t_limit][1], avail_u_next int8[B][t_limit][n][A],
 * terminated / padded uint8 (bool) [B][t_limit][1].  Only steps t < T are read.
 */
#ifndef VDN_OPS_H
#define VDN_OPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDN_OK 0
#define VDN_ERR_BAD_ARG (-1)
#define VDN_ERR_HIP (-100)

/* d_mtd[b*T + t] = masked_td_error, d_mask[b*T + t] = 1 - padded (float32 [B*T] each); the float operations and their
 * order are the reference's: gamma * q_total_target, then * (1 - terminated), then r + ...; agents summed in index order.
 * An action outside [0, n_actions) (torch.gather raises on it, policy/vdn.py:106) is never used as an index: the slot's
 * d_mtd becomes NaN (so the loss and every gradient of that learn are NaN) and, when d_bad_actions is not NULL, the int32
 * device counter it points to is incremented once per such (episode, step) -- the call is asynchronous, so the caller reads
 * the counter when it next synchronises. */
int vdn_td_forward(const float *d_q_eval, const float *d_q_target, const int8_t *d_u, const float *d_r,
                   const int8_t *d_avail_next, const uint8_t *d_terminated, const uint8_t *d_padded, int32_t B, int32_t T,
                   int32_t t_limit, int32_t n_agents, int32_t n_actions, float gamma, float *d_mtd, float *d_mask,
                   int32_t *d_bad_actions, void *stream);

/* Gradient of  num = sum(masked_td_error ** 2)  w.r.t. the eval network's Q values, scaled by the upstream gradient
 * *d_grad_num (device scalar):  d_grad_q[t][b][i][a] = -(2 * mtd * mask) * *d_grad_num  if a == u[b][t][i], else 0
 * (float32 [T][B][n][A], every element written). */
int vdn_td_backward(const float *d_mtd, const float *d_mask, const int8_t *d_u, const float *d_grad_num, int32_t B, int32_t T,
                    int32_t t_limit, int32_t n_agents, int32_t n_actions, float *d_grad_q, void *stream);

/* The same block for a batch whose padded steps were left out (episodes sorted by length, the valid (episode, step) UNITS of
 * step 0 first, then those of step 1, ...: the layout of gru_seq_forward_packed in crnn_ops.h).  d_units int32[n_units]:
 * unit j is the step `slot * t_limit + t` of the replay tensors, which are indexed directly (no gathered copies):
 * d_u int8[slots*t_limit][n], d_r float32[slots*t_limit], d_avail_next int8[slots*t_limit][n][A], d_terminated / d_padded
 * uint8[slots*t_limit].  Rows j*n .. j*n+n-1 of d_q_eval / d_q_target (float32[n_units*n][A]) belong to unit j.
 * d_mtd / d_mask float32[n_units]; arithmetic and error behaviour as vdn_td_forward. */
int vdn_td_forward_packed(const float *d_q_eval, const float *d_q_target, const int32_t *d_units, int32_t n_units, const int8_t *d_u,
                          const float *d_r, const int8_t *d_avail_next, const uint8_t *d_terminated, const uint8_t *d_padded,
                          int32_t n_agents, int32_t n_actions, float gamma, float *d_mtd, float *d_mask, int32_t *d_bad_actions,
                          void *stream);
/* d_grad_q float32[n_units*n][A] = the gradient of vdn_td_backward in that layout (every element written). */
int vdn_td_backward_packed(const float *d_mtd, const float *d_mask, const int32_t *d_units, int32_t n_units, const int8_t *d_u,
                           const float *d_grad_num, int32_t n_agents, int32_t n_actions, float *d_grad_q, void *stream);

/* The inputs of the packed learn (policy/vdn.py:134-165: o / o_next of a step and the previous step's action one-hot) copied out
 * of the replay tensors unit by unit: d_dst unit j (unit_bytes bytes) = d_src unit d_units[j] + unit_shift, or zeros for
 * j < zero_below (the last action of step 0).  A unit = the n_agents consecutive rows of one (slot, step). */
int vdn_gather_units(const void *d_src, int32_t unit_bytes, const int32_t *d_units, int32_t n_units, int32_t unit_shift,
                     int32_t zero_below, void *d_dst, void *stream);

/* The tail of VDN.learn (policy/vdn.py:125-127): torch.nn.utils.clip_grad_norm_(eval_parameters, max_norm) followed by
 * optimizer.step() of torch.optim.Adam(lr, betas) (policy/vdn.py:67-68), for up to VDN_MAX_TENSORS float32 parameter tensors,
 * in two launches instead of the ~11 of the foreach / fused torch path:
 *   total = sqrt(sum over all tensors of sum(g^2));  c = min(1, max_norm / (total + 1e-6));  g' = g * c
 *   m = m + (1 - beta1) * (g' - m);  v = beta2 * v + (1 - beta2) * g' * g';
 *   p -= (lr / bias_correction1) * m / (sqrt(v) / sqrt(bias_correction2) + eps)
 * (torch's formulas and element arithmetic: bias_correction = 1 - beta^step is computed by the caller in double; 1 - beta,
 * lr / bias_correction1, sqrt(bias_correction2) and eps are rounded to float once and every element operation is a float operation
 * in torch's order: m.lerp, v = beta2 v + ((1 - beta2) g) g, p += -step (m / denom)).  A NaN norm makes every gradient NaN, as
 * clip_grad_norm_ does.  d_grad_div (device float, may be NULL): every gradient is first divided by *d_grad_div -- VDN.learn
 * differentiates the UN-normalised loss sum((mask td)^2) and divides by the mask count here (policy/vdn.py:122), the same
 * arithmetic with and without data parallelism (there the count is the all-reduced one).  The gradients are rescaled in memory as well
 * (clip_grad_norm_ leaves g' in p.grad).  d_partials: VDN_NORM_BLOCKS floats of scratch;
 * *d_total_norm receives `total` (the value clip_grad_norm_ returns).  The pointer arrays are HOST arrays of DEVICE pointers. */
#define VDN_MAX_TENSORS 32
#define VDN_NORM_BLOCKS 128
int vdn_clip_adam_step(int32_t n_tensors, float *const *params, float *const *grads, float *const *exp_avg,
                       float *const *exp_avg_sq, const int64_t *numel, float max_norm, double lr, double beta1, double beta2,
                       double eps, double bias_correction1, double bias_correction2, float *d_partials, float *d_total_norm,
                       const float *d_grad_div, void *stream);

int vdn_last_hip_error(void);

#ifdef __cplusplus
}
#endif
#endif /* VDN_OPS_H */
