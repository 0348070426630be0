/*
 * crnn_ops.h -- C ABI of the hand-written HIP inference kernel for the convolutional front end of
 * the reference's per-agent Q-network `CRNN` (network/base_net.py:35-71): for fov 9 the stack is
 * Conv2d(3->od,k3,s1)+ReLU -> Conv2d(od->od,k3,s1)+ReLU -> flatten (c,h,w) (base_net.py:23-33,63-65).
 *
 * Used by the vectorised rollout (agent/agent.py:22-48 batched over envs x agents): the observation
 * rows are the int8 rows written by the env kernels, the output is the float32 feature block the GRU
 * consumes.  Forward only (rollouts run under no_grad); training uses the autograd path.
 * fp32 arithmetic, same products as torch.nn.functional.conv2d, different summation order.
 */
#ifndef CRNN_OPS_H
#define CRNN_OPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRNN_OK 0
#define CRNN_ERR_BAD_ARG (-1)
#define CRNN_ERR_UNSUPPORTED (-6) /* only fov 9 with od 24 or 32 (the reference's 4d / other yaml values) */
#define CRNN_ERR_HIP (-100)

/* d_obs:  int8 [rows][obs_stride], first 243 bytes of a row = pixel block (3,9,9) in (c,x,y) order
 * d_w1:   float32 [od][3][3][3]   d_b1: [od]      (conv1.weight / conv1.bias)
 * d_w2:   float32 [od][od][3][3]  d_b2: [od]      (conv2.weight / conv2.bias)
 * d_out:  float32 [rows][out_stride], first od*25 entries of a row are written, index c*25 + h*5 + w */
int crnn_conv9_forward(const int8_t *d_obs, int64_t obs_stride, int64_t rows, const float *d_w1, const float *d_b1,
                       const float *d_w2, const float *d_b2, int od, float *d_out, int64_t out_stride, void *stream);
/* The whole non-recurrent front end of CRNN.forward (network/base_net.py:59-68) for fov 9: the pixel
 * block as above AND the vector branch relu(mlp1([dir(2), last_action_onehot(n_actions)])), written
 * side by side so that a row of d_out IS the GRU input `x = cat([pixel, vec], dim=1)`:
 *   d_out[r][0 .. od*25)            conv features
 *   d_out[r][od*25 .. od*25+10)     relu(mlp1(vec)),  vec = [obs[r][243], obs[r][244], onehot[r][0..n_actions)]
 * d_onehot: int8 [rows][n_actions] (may be NULL = all zeros: first step of an episode);
 * d_mlp_w: float32 [10][2+n_actions] (mlp1.weight), d_mlp_b: [10]; n_actions <= 16.
 * out_cols: 0, or od*25+10 <= out_cols <= crnn_front_padded_cols(od) (and <= out_stride): the columns od*25+10 ..
 * out_cols-1 of every row are written as ZEROS, so that the row can feed the GRU input projection as a GEMM operand with
 * K = 640 (od 24) / 832 (od 32) against a weight zero-padded alike -- same sums, and rocBLAS runs K = 640 15-25 % faster
 * than K = 610 on gfx950 (tools/probe/gemm_align_probe.py). */
int crnn_front9_forward(const int8_t *d_obs, int64_t obs_stride, const int8_t *d_onehot, int n_actions, int64_t rows,
                        const float *d_w1, const float *d_b1, const float *d_w2, const float *d_b2, const float *d_mlp_w,
                        const float *d_mlp_b, int od, float *d_out, int64_t out_stride, int out_cols, void *stream);
/* The same for the LIVE chips of a lock-step rollout only (common/rollout.py: a chip whose episode is over is frozen until the
 * round ends; the reference simply leaves its loop, rollout.py:108).  d_live_chips int32[<= rows / rows_per_chip]: ascending ids
 * of the chips still playing, *d_n_live their count (both on the device: rollout_compact_alive of rollout_ops.h writes them, so a
 * captured HIP graph follows the count without host involvement).  Input row of compact row k * rows_per_chip + a is
 * d_live_chips[k] * rows_per_chip + a (obs and one-hot alike); d_out rows are COMPACT: row k * rows_per_chip + a.  `rows` =
 * the worst case (all chips live) the grid is sized for; workgroups beyond the live rows exit at once; rows of d_out beyond
 * *d_n_live * rows_per_chip are not written. */
int crnn_front9_forward_live(const int8_t *d_obs, int64_t obs_stride, const int8_t *d_onehot, int n_actions, int64_t rows,
                             const float *d_w1, const float *d_b1, const float *d_w2, const float *d_b2, const float *d_mlp_w,
                             const float *d_mlp_b, int od, float *d_out, int64_t out_stride, int out_cols, const int32_t *d_live_chips,
                             const int32_t *d_n_live, int rows_per_chip, void *stream);
/* od*25+10 rounded up to a multiple of 64 (640 / 832), or CRNN_ERR_UNSUPPORTED. */
int crnn_front_padded_cols(int od);
/* The same front end for fov 19 (the MEDA v0_2 observation): conv_str(19) of network/base_net.py:23-33 =
 * Conv2d(3, od, 3, stride 2) + ReLU, then conv3 = Conv2d(od, od, 3) + ReLU applied TWICE (the two list entries are the
 * same module: tied weights), 19x19 -> 9x9 -> 7x7 -> 5x5, plus the vector branch as in crnn_front9_forward.
 * d_obs: int8 [rows][obs_stride >= 1085]: 3*19*19 pixel bytes (index c*361 + a*19 + b, the layout the MEDA observation
 *        kernel writes), then dir_x, dir_y;  d_w1 float32 [od][3][3][3], d_w3 float32 [od][od][3][3], biases [od];
 * d_out[r][0 .. od*25) conv features (index c*25 + h*5 + w), d_out[r][od*25 .. od*25+10) relu(mlp1(vec)).
 * d_mlp_w == NULL: pixel features only (obs_stride >= 1083, out_stride >= od*25).  od 24 or 32, n_actions <= 16.
 * out_cols as in crnn_front9_forward. */
int crnn_front19_forward(const int8_t *d_obs, int64_t obs_stride, const int8_t *d_onehot, int n_actions, int64_t rows,
                         const float *d_w1, const float *d_b1, const float *d_w3, const float *d_b3, const float *d_mlp_w,
                         const float *d_mlp_b, int od, float *d_out, int64_t out_stride, int out_cols, void *stream);
/* Gradients of the four conv tensors for the eval network of VDN.learn (policy/vdn.py:123-128 backward through
 * network/base_net.py:63-65); the observation needs none.  Nothing is saved by the forward: the conv1 activations of each
 * row block are recomputed inside the kernel (f32 MFMA, as the forward computes them) and the next block's inputs are
 * prefetched while the current one is reduced.  d_out is the forward's output (crnn_conv9_forward or crnn_front9_forward;
 * its sign is the ReLU mask), d_grad_out the gradient w.r.t. it.  One workgroup per partial vector accumulates over its
 * rows into d_part float32[n_part][crnn_conv9_backward_parts(od)] (scratch, n_part <= 256), then a second small kernel
 * adds the partial vectors (fixed order: deterministic) into
 * d_grads float32[od*od*9 + od + od*27 + od] = dW2[od][od][3][3] | db2[od] | dW1[od][3][3][3] | db1[od]. */
int crnn_conv9_backward_parts(int od);
int crnn_conv9_backward(const int8_t *d_obs, int64_t obs_stride, int64_t rows, const float *d_out, int64_t out_stride,
                        const float *d_grad_out, int64_t grad_stride, const float *d_w1, const float *d_b1, const float *d_w2,
                        int od, float *d_part, int n_part, float *d_grads, void *stream);
/* Gradients of the vector branch relu(mlp1([dir_x, dir_y, last-action one-hot])) of the GRU input row (network/base_net.py:66,
 * backward of policy/vdn.py:123-128): d_grad_w float32[10][2 + n_actions], d_grad_b float32[10].  d_out / d_grad_out are the
 * forward's output rows (crnn_front9_forward / crnn_front19_forward) and the gradient w.r.t. them; the branch's ten columns
 * start at col0 (od*25).  The direction bytes are d_obs[row*obs_stride + dir_offset .. +1] (243 for fov 9, 1083 for fov 19).
 * d_part: scratch float32[crnn_mlp_backward_parts()].  Two launches; sums in a fixed order (deterministic). */
int crnn_mlp_backward_parts(void);
int crnn_mlp_backward(const int8_t *d_obs, int64_t obs_stride, int dir_offset, const int8_t *d_onehot, int n_actions, int64_t rows,
                      const float *d_out, int64_t out_stride, const float *d_grad_out, int64_t grad_stride, int col0, float *d_part,
                      float *d_grad_w, float *d_grad_b, void *stream);
/* Gradients of the fov-19 conv stack (conv_str(19) of network/base_net.py:23-33: stride-2 conv1, then the tied conv3 twice) for
 * the eval network of VDN.learn on MEDA: d_out = crnn_front19_forward's output (its sign is the last ReLU's mask), d_grad_out the
 * gradient w.r.t. it (only the first od*25 columns of a row are read).  Nothing is saved by the forward: a1 and a2 of each row
 * block are recomputed on the matrix cores; conv3's two applications add into ONE weight gradient.  One persistent workgroup
 * per partial vector (n_part <= 256) accumulates over its rows into d_part float32[n_part][crnn_conv19_backward_parts(od)]
 * (scratch); a second small kernel adds the partial vectors in a fixed order (deterministic) into
 * d_grads float32[od*od*9 + od + od*27 + od] = dW3[od][od][3][3] | db3[od] | dW1[od][3][3][3] | db1[od]. */
int crnn_conv19_backward_parts(int od);
int crnn_conv19_backward(const int8_t *d_obs, int64_t obs_stride, int64_t rows, const float *d_out, int64_t out_stride,
                         const float *d_grad_out, int64_t grad_stride, const float *d_w1, const float *d_b1, const float *d_w3,
                         const float *d_b3, int od, float *d_part, int n_part, float *d_grads, void *stream);
int crnn_last_hip_error(void);

/* ---- GRU cell unrolled over an episode (network/base_net.py:56,69 nn.GRUCell; policy/vdn.py:174-191 time loop) ----
 * One launch runs all T steps of h_t = GRUCell(x_t, h_{t-1}) for R independent rows (hidden must be 128):
 *   d_igates float32[T][R][3H] = x_t @ W_ih^T WITHOUT bias (one GEMM over all steps, done by the caller)
 *   d_h0     float32[R][H]       d_w_hh float32[3H][H]   d_b_ih, d_b_hh float32[3H]   gate order r|z|n
 *   d_hs     float32[T][R][H]    every h_t
 *   d_gates  float32[T][R][4H]   saved r | z | n | (W_hn h + b_hn) for the backward; NULL for inference */
int gru_seq_forward(const float *d_igates, const float *d_h0, const float *d_w_hh, const float *d_b_ih, const float *d_b_hh,
                    int T, int64_t R, int hidden, float *d_hs, float *d_gates, void *stream);
/* Backward through time of the above: d_grad_hs float32[T][R][H] = dL/dh_t from the consumers of each step.
 * Writes d_d_igates float32[T][R][3H] (gradient w.r.t. d_igates = w.r.t. the pre-activations on the input side, so
 * db_ih is its column sum) and d_d_hgates float32[T][R][3H] (w.r.t. W_hh h + b_hh: dW_hh = d_hgates^T @ h_{t-1}
 * stacked over t, db_hh its column sum), optionally d_d_h0 float32[R][H].  When d_bias_part is non-NULL it receives
 * float32[gru_seq_row_blocks(R)][6H]: per row block the column sums db_ih (3H) | db_hh (3H) over its rows and all steps;
 * the caller adds the blocks (fixed order: deterministic). */
int64_t gru_seq_row_blocks(int64_t R);
int gru_seq_backward(const float *d_grad_hs, const float *d_gates, const float *d_hs, const float *d_h0, const float *d_w_hh,
                     int T, int64_t R, int hidden, float *d_d_igates, float *d_d_hgates, float *d_d_h0, float *d_bias_part,
                     void *stream);
/* The same two for PACKED sequences of different lengths (VDN.learn on episodes that ended early: the reference trims a batch
 * to its longest episode, agent/agent.py:51-61, and masks the padded steps, policy/vdn.py:115-122; here the padded steps are
 * not computed at all).  The R rows are sorted by sequence length, longest first; step_rows (HOST array int32[T], non-increasing,
 * step_rows[0] <= R) = rows still running at step t, i.e. the first step_rows[t] rows of the batch.  Every per-step tensor
 * (d_igates, d_hs, d_gates, d_grad_hs, d_d_igates, d_d_hgates) holds the running rows of step 0, then those of step 1, ...
 * back to back: sum(step_rows) rows in all, row (t, r) at offset sum(step_rows[0..t)) + r.  Rows not running at a step are
 * neither read nor written; a sequence's gradient chain starts at its own last step.  T <= GRU_SEQ_MAX_STEPS.
 * d_h_prev (backward, optional) float32[sum(step_rows)][H] receives h_{t-1} of every running (t, row) in that same layout, so that
 * dW_hh = d_d_hgates^T @ d_h_prev is one GEMM although consecutive steps hold different numbers of rows. */
#define GRU_SEQ_MAX_STEPS 255
int gru_seq_forward_packed(const float *d_igates, const float *d_h0, const float *d_w_hh, const float *d_b_ih, const float *d_b_hh,
                           int T, int64_t R, int hidden, const int32_t *step_rows, float *d_hs, float *d_gates, void *stream);
/* gru_seq_forward_packed for ONE or TWO networks over the same packed batch in one launch (the eval and the target network of
 * VDN.learn, policy/vdn.py:174-191): the recurrence h W_hh^T runs on the matrix cores, 16 rows per workgroup, so that the two
 * networks' 2 x R/16 workgroups fill the GPU where one network's R/8 did (csrc/gru_ops.hip).  Network b is skipped when
 * d_igates_b is NULL.  d_h0_* NULL = zero initial state; d_gates_* NULL = gates not saved (no backward).  Outputs as
 * gru_seq_forward_packed (same layouts: gru_seq_backward_packed takes them); sums in another order: values agree to rounding. */
int gru_seq_forward_packed_pair(const float *d_igates_a, const float *d_h0_a, const float *d_w_hh_a, const float *d_b_ih_a,
                                const float *d_b_hh_a, float *d_hs_a, float *d_gates_a, const float *d_igates_b, const float *d_h0_b,
                                const float *d_w_hh_b, const float *d_b_ih_b, const float *d_b_hh_b, float *d_hs_b, float *d_gates_b,
                                int T, int64_t R, int hidden, const int32_t *step_rows, void *stream);
int gru_seq_backward_packed(const float *d_grad_hs, const float *d_gates, const float *d_hs, const float *d_h0, const float *d_w_hh,
                            int T, int64_t R, int hidden, const int32_t *step_rows, float *d_d_igates, float *d_d_hgates,
                            float *d_d_h0, float *d_bias_part, float *d_h_prev, void *stream);
int gru_last_hip_error(void);

#ifdef __cplusplus
}
#endif
#endif
