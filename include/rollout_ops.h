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

/* ---- continuous rollout ("stream" mode) -------------------------------------------------------------------------------------
 * The reference plays one chip: when its episode ends, generate_episode returns, the caller stores the episode
 * (train.py:62-73) and calls generate_episode again, which resets the chip (common/rollout.py:101-112).  In the lock-step
 * batch every chip does exactly that on its own clock: a chip whose episode ended in lock-step s starts its next episode in
 * lock-step s + 1, and the finished episode is written into the replay ring (common/replay_buffer.py:33-46) on the device, so a
 * round is a fixed number of lock-steps in which EVERY row is live, however short the policy's episodes are.
 *
 * rollout_stage: per-chip state of the episode being played (DEVICE pointers, caller-owned; zero-initialised):
 *   d_t_ep int32[2][E]        step index inside the running episode, double-buffered: the call with parity p reads row p and
 *                             writes row 1 - p (rollout_gru_head_select_stream of the same lock-step is given row p)
 *   d_o0 int8[E][row]         its first observation;   d_o_next int8[E][T][row]  observation after each step (row = n * obs bytes)
 *   d_u int8[E][T][n]         d_onehot int8[E][T][n][A]      d_r float32[E][T]
 *   d_ep_acc float64[E][3]    running (reward, constraints, success) of the episode (rollout.py:122-124)
 *   d_chip_acc int64[E][4]    += (episodes closed, their steps with the failure inflation of rollout.py:148-149, successes,
 *                                 env steps played); the caller sums / clears them between rounds
 *   d_close_slot int32[E]     written by rollout_stream_step: ring slot the chip's episode closed into in this lock-step, or -1
 *   d_state_alt int64[4]      the second buffer of the ring state (see rollout_stream_step)
 * rollout_ring: the replay buffer's tensors (common/replay_buffer.py:10-28, `slots` episodes of T steps) plus
 *   d_len int32[slots]        valid steps of the episode in each slot (0 = never written)
 *   d_stats float64[slots][4] (reward, steps incl. failure inflation, constraints, success) = generate_episode's return values
 *   d_state int64[4]          [0] next slot to write (ring cursor), [1] slots filled (current_size), [2] episodes closed so far
 */
typedef struct {
    int32_t *d_t_ep;
    int8_t *d_o0, *d_o_next, *d_u, *d_onehot;
    float *d_r;
    double *d_ep_acc;
    int64_t *d_chip_acc;
    int32_t *d_close_slot;
    int64_t *d_state_alt;
} rollout_stage;

typedef struct {
    int32_t slots;
    int8_t *d_o, *d_o_next, *d_u, *d_u_onehot, *d_avail_u, *d_avail_u_next;
    float *d_r;
    uint8_t *d_padded, *d_terminated;
    int32_t *d_len;
    double *d_stats;
    int64_t *d_state;
} rollout_ring;

/* rollout_gru_head_select with every chip at its own step: the pick of row (e, a) is staged at d_stage_u[e][d_t_ep[e]][a]
 * (and its one-hot alike) instead of one common slot t. */
int rollout_gru_head_select_stream(const float *d_igates, const float *d_hgates, const float *d_b_ih, const float *d_b_hh, float *d_h,
                                   const float *d_fc_w, const float *d_fc_b, int32_t n_envs, int32_t n_agents, int32_t hidden,
                                   int32_t n_actions, const float *d_epsilon, int32_t evaluate, uint64_t seed, const uint32_t *d_draw,
                                   int32_t *d_actions, int8_t *d_last_onehot, int8_t *d_stage_u, int8_t *d_stage_onehot,
                                   int32_t episode_limit, const int32_t *d_t_ep, float *d_q, void *stream);

/* After the transition of a lock-step (rollout.py:118-129), for every chip e at step t = d_t_ep[e] of its episode:
 *   stage r[e][t] = team_reward[e], o_next[e][t] = d_obs_new[e] (and o0[e] = d_obs_prev[e] when t == 0: the observation the step
 *   was chosen from); the episode's running sums += this step's reward / constraints / success;
 *   if d_term[e]: the episode closes into ring slot (cursor + k) % slots, k = number of LOWER-numbered chips closing in this
 *   lock-step (deterministic, chip order): the whole episode is written into the slot with the reference's padding
 *   (rollout.py:131-141: rows behind the end are zeros, padded = 1, terminated = 1, avail_u = avail_u_next = 0; o[t] = o0 for
 *   t == 0, o_next[t - 1] after), ring d_len / d_stats of the slot and the chip's counters are written, the running sums, the
 *   chip's rows of d_hidden (float32[E*n][hidden]; policy.init_hidden, rollout.py:112) and d_last_onehot (rollout.py:110) are
 *   cleared, d_close_slot[e] = that slot and the chip's step index becomes 0; every other chip gets d_close_slot[e] = -1 and its
 *   step index + 1.
 *   Then ring cursor / size / total advance, *d_epsilon = max(*d_epsilon - anneal * n_envs, min_epsilon) (every chip played a
 *   step; epsilon_anneal_scale == 'step', rollout.py:126-127), *d_draw += 1.
 * The ring state is double-buffered inside one call (the cursor is read by every closing chip while the new one is published):
 * parity 0 reads ring->d_state and writes stage->d_state_alt, parity 1 the other way round (stage->d_t_ep likewise, rows 0 / 1: the
 * observation rows of an ended episode are copied by ALL the workgroups of the launch, which read its step index while the chip's
 * own workgroup resets it); the caller alternates and, after an odd number of calls, copies the second buffers back.  n_envs <= 32 768.  The caller then resets the closed chips' env (dmfb_vec_reset / meda_vec_reset with
 * d_term as mask), which also rewrites their rows of d_obs_new with the first observation of the next episode.
 * d_obs_term (may be NULL): when the env resets ended chips INSIDE its transition launch (DMFB_STEP_AUTORESET with
 * dmfb_vec_step_out::d_obs_terminal) d_obs_new already holds an ended chip's next first observation and its terminal observation
 * is taken from d_obs_term[e] instead; no reset call follows then.
 * obs_row_bytes = n_agents * observation bytes. */
int rollout_stream_step(int32_t n_envs, int32_t n_agents, int32_t n_actions, int32_t episode_limit, int32_t obs_row_bytes,
                        int32_t hidden, const int8_t *d_obs_prev, const int8_t *d_obs_new, const int8_t *d_obs_term, const uint8_t *d_term,
                        const double *d_team_reward, const void *d_constraints, int32_t constraints_f64, const uint8_t *d_success,
                        const rollout_stage *stage, const rollout_ring *ring, int32_t parity, float *d_hidden, int8_t *d_last_onehot,
                        float *d_epsilon, float anneal, float min_epsilon, uint32_t *d_draw, void *stream);

int rollout_last_hip_error(void);

#ifdef __cplusplus
}
#endif
#endif
