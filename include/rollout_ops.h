/*
 * rollout_ops.h -- C ABI of the two book-keeping kernels of the vectorised rollout
 * (reference common/rollout.py:101-150 generate_episode, agent/agent.py:22-48 choose_action), for E chips
 * x n droplets played in lock-step on one GPU.  Between the Q-network and the env transition the
 * reference does per agent: epsilon-greedy pick, one-hot of the action, append to the episode lists;
 * after the transition: reward / padded / terminated appends and the running sums.  Done op by op on
 * tensors that is ~25 launches of a few microseconds per lock-step; here it is one launch before
 * and one after dmfb_vec_step / meda_vec_step.
 *
 * Conventions as dmfb_vec.h: plain C types, caller-owned DEVICE buffers, `stream` = hipStream_t as void*,
 * asynchronous, negative int error codes.  Episode tensors are chip-major with the reference's shapes:
 * u int8[E][T][n][1], u_onehot int8[E][T][n][A], r float32[E][T][1], padded / terminated uint8[E][T][1].
 */
#ifndef ROLLOUT_OPS_H
#define ROLLOUT_OPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ROLLOUT_OK 0
#define ROLLOUT_ERR_BAD_ARG (-1)
#define ROLLOUT_ERR_HIP (-100)

/* agent/agent.py:41-45 for every (chip, droplet) row r = e*n + a (all actions available, as in both envs):
 *   greedy = argmax_k q[r][k] (first maximum);  if (!evaluate && U1 < *d_epsilon) action = floor(U2 * A) else greedy.
 * U1, U2: Philox4x32-10, key = seed, counter = (r, *d_draw, 0, 0x600): words 0 and 1; U1 = (w0 >> 8) * 2^-24.
 * The draw counter lives on the device (rollout_post_step advances it) so that a captured HIP graph of the
 * episode draws fresh numbers on every replay.
 * Writes d_actions int32[E*n] (the env's input), d_last_onehot int8[E*n][A] (next step's last-action input) and,
 * when d_ep_u / d_ep_onehot are non-NULL, slot t of the episode tensors. */
int rollout_select_actions(const float *d_q, int32_t n_envs, int32_t n_agents, int32_t n_actions, const float *d_epsilon,
                           int32_t evaluate, uint64_t seed, const uint32_t *d_draw, int32_t *d_actions, int8_t *d_last_onehot,
                           int8_t *d_ep_u, int8_t *d_ep_onehot, int32_t episode_limit, int32_t t, void *stream);

/* The tail of the per-step Q-network (network/base_net.py:69-70: GRUCell gate math + fc1) fused with the pick above: given
 * d_igates = x W_ih^T and d_hgates = h W_hh^T (float32[E*n][3H], no bias, gate order r|z|n: the two GEMMs stay in the BLAS
 * library) it computes h' = GRUCell(...) in place in d_h (float32[E*n][H]), q = fc1(h') (d_fc_w float32[A][H], d_fc_b [A];
 * A <= 16; optionally written to d_q float32[E*n][A]) and then exactly rollout_select_actions on q.  hidden must be 128
 * (ROLLOUT_ERR_BAD_ARG otherwise: use the BLAS + rollout_select_actions path). */
int rollout_gru_head_select(const float *d_igates, const float *d_hgates, const float *d_b_ih, const float *d_b_hh, float *d_h,
                            const float *d_fc_w, const float *d_fc_b, int32_t n_envs, int32_t n_agents, int32_t hidden,
                            int32_t n_actions, const float *d_epsilon, int32_t evaluate, uint64_t seed, const uint32_t *d_draw,
                            int32_t *d_actions, int8_t *d_last_onehot, int8_t *d_ep_u, int8_t *d_ep_onehot, int32_t episode_limit,
                            int32_t t, float *d_q, void *stream);

/* The chips still playing: d_live_chips int32[n_envs] receives the ids e with d_alive[e] != 0 in ASCENDING order, *d_n_live
 * their number (device memory: the next lock-step's kernels read both there, so a captured HIP graph of the episode follows
 * the shrinking batch without host involvement).  The reference's loop simply ends when an episode is over (rollout.py:108);
 * in the lock-step batch a finished chip would otherwise ride through the Q-network until the slowest chip is done. */
int rollout_compact_alive(int32_t n_envs, const uint8_t *d_alive, int32_t *d_live_chips, int32_t *d_n_live, void *stream);

/* rollout_gru_head_select for the chips listed by rollout_compact_alive only: row k * n_agents + a of d_igates (COMPACT, as
 * crnn_front9_forward_live of crnn_ops.h leaves the x-side operand) belongs to row d_live_chips[k] * n_agents + a of every other
 * tensor (d_hgates, d_h, d_actions, d_last_onehot, the episode tensors, d_q: all in chip order).  Rows of finished chips are not
 * touched; the Philox draw of a row is keyed by its chip-order index, so the picks do not depend on who else is still playing. */
int rollout_gru_head_select_live(const float *d_igates, const float *d_hgates, const float *d_b_ih, const float *d_b_hh, float *d_h,
                                 const float *d_fc_w, const float *d_fc_b, int32_t n_envs, int32_t n_agents, int32_t hidden,
                                 int32_t n_actions, const float *d_epsilon, int32_t evaluate, uint64_t seed, const uint32_t *d_draw,
                                 int32_t *d_actions, int8_t *d_last_onehot, int8_t *d_ep_u, int8_t *d_ep_onehot, int32_t episode_limit,
                                 int32_t t, float *d_q, const int32_t *d_live_chips, const int32_t *d_n_live, void *stream);

/* After the transition of lock-step t (rollout.py:118-129) for every chip e, with a = d_alive[e] BEFORE the step:
 *   r[e][t] = team_reward[e]; padded[e][t] = !a; terminated[e][t] = term[e]       (episode pointers may be NULL)
 *   sum_reward[e] += team_reward[e]; sum_constraints[e] += constraints[e]; sum_success[e] += success[e];
 *   steps[e] += a; d_alive[e] = a && !term[e]
 *   *d_epsilon = max(*d_epsilon - anneal * #{e: a}, min_epsilon) when anneal > 0 (epsilon_anneal_scale == 'step')
 *   d_n_alive[0] = #{e: d_alive[e]} after the update (the host polls it to stop an all-finished round early); d_n_alive
 *                 is an int32[4] workspace that must be ZERO before the first call ([1..3] are returned to zero by every call)
 *   *d_draw += 1 (the Philox draw counter of rollout_select_actions; may be NULL)
 * and, when d_ep_o / d_ep_o_next are non-NULL (int8[E][T][obs_row_bytes], ZERO-initialised by the caller; d_obs =
 * int8[E][obs_row_bytes], the observation after this step), the observation appends with the padding rule applied:
 *   o_next[e][t] = obs[e] if a;   o[e][t+1] = obs[e] if a && !term[e] && t + 1 < T   (masked rows stay zero)
 * d_constraints is int32[E] (constraints_f64 == 0, DMFB) or float64[E] (== 1, MEDA). Frozen chips report
 * team_reward 0, constraints 0, success 0, term 1 from the env kernels. */
int rollout_post_step(int32_t n_envs, int32_t episode_limit, int32_t t, uint8_t *d_alive, const uint8_t *d_term,
                      const double *d_team_reward, const void *d_constraints, int32_t constraints_f64,
                      const uint8_t *d_success, float *d_ep_r, uint8_t *d_ep_padded, uint8_t *d_ep_terminated,
                      double *d_sum_reward, double *d_sum_constraints, int64_t *d_sum_success, int64_t *d_steps,
                      float *d_epsilon, float anneal, float min_epsilon, int32_t *d_n_alive, uint32_t *d_draw,
                      const int8_t *d_obs, int32_t obs_row_bytes, int8_t *d_ep_o, int8_t *d_ep_o_next, void *stream);

int rollout_last_hip_error(void);

#ifdef __cplusplus
}
#endif
#endif
