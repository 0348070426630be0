"""Vectorised rollout with the reference's API surface (common/rollout.py:10-150).

The reference plays ONE chip: per step it calls the Q-net once per agent with batch size 1,
steps the env, and appends the transition; `generate_episode` pads to `episode_limit`.  Here an
`Evaluator`/`RolloutWorker` owns a `VecDMFB` batch of E chips and plays one episode on EVERY chip
in lock-step: one Q-net forward over (E x n) rows, epsilon-greedy on the device, one fused HIP
transition for all chips.  Everything stays in HBM; the episode batch has the reference's keys,
shapes and padding rules with leading dimension E.

Book-keeping kept from the reference:
  * team reward = np.sum(rewards)/n (computed inside the transition kernel in numpy's order)
  * `terminated` = all(dones); an env that terminated is frozen (active mask) until the next reset
  * padding: zeros, padded = 1, terminated = 1 after the end of an episode (rollout.py:131-141)
  * `steps` is forced to `episode_limit` for unsuccessful episodes (rollout.py:60-61,148-149)
  * epsilon anneals per env-step: one lock-step of k live chips anneals k steps (rollout.py:126-127)

Continuous mode (`RolloutWorker.generate_steps`, the Trainer's default on the GPU): every chip plays on its own clock -- a chip
whose episode ended starts its next one in the following lock-step, and the finished episode is written into the replay ring
on the device (include/rollout_ops.h, "stream" entry points).  A round is then a fixed number of lock-steps in which every
(chip, droplet) row is live, whatever the length of the policy's episodes; the episodes in the ring are, one by one, what
`generate_episode` of the reference returns (tests/test_gpu_rollout_stream.py replays them through the CPU oracle).
"""
import types

import torch


class Evaluator:
    def __init__(self, env, agents, episode_limit):
        self.agents = agents
        self.env = env
        self.n_agents = agents.n_agents
        self.n_actions = agents.n_actions
        self.episode_limit = episode_limit
        self.device = env.device
        self.n_envs = env.n_envs
        self.generator = None
        self.sync_every = 4  # lock-steps between host checks for "every chip has terminated"
        self.reset_fn = None  # tests replace env.reset() (e.g. by restart() on an injected task)
        self.uniforms_fn = None  # tests inject the move draws: uniforms_fn(t) -> float64 [E, n] for lock-step t (else Philox)
        # HIP-graph mode: the whole lock-step episode (reset, T x [Q-net, epsilon-greedy, fused env
        # transition, book-keeping]) is captured once and replayed, which removes the per-op host
        # launch cost that otherwise dominates a lock-step of a few thousand chips.
        self.use_graph = False
        self._graphs = {}
        self._rollout_lib = None
        self.fuse_tail = True  # GRU gate math + fc1 + epsilon-greedy as one kernel when the Q-net is the fov-9 CRNN
        # Finished chips stay out of the Q-network: every `compact_every` lock-steps the ids of the chips still playing are
        # listed on the device (include/rollout_ops.h: rollout_compact_alive) and the conv front end and the GRU-head kernel
        # walk that list (worst-case grids, device-side count: graph-capturable).  Between two compactions the list is a
        # superset of the live chips, which is harmless: a finished chip's rows are computed and ignored, as without the list.
        # The two GRU projections stay full-size library GEMMs.  0 switches it off.
        # The list costs ~2 % of a lock-step when every chip plays to the end (one more launch every `compact_every` steps, an
        # indirection in two kernels), so it is used only while episodes do end early: `note_played` keeps the share of
        # (chip, lock-step) slots that were live in the last round, and the list is on when that share is below `live_threshold` (break-even measured near 0.75: tools/bench_front_live.py).
        self.compact_every = 4
        self.live_threshold = 0.75
        self.live_share = 1.0
        # key of the epsilon-greedy Philox stream: the env seed, shifted per shard so that ranks draw different numbers
        self.rng_seed = (int(getattr(env, 'seed', 0)) * 0x9E3779B97F4A7C15 + int(getattr(env, 'env_id0', 0)) + 0x600) & 0xFFFFFFFFFFFFFFFF

    def note_played(self, played):
        """env steps played in the round just finished (a host number the caller has anyway) -> share of live slots."""
        self.live_share = float(played) / float(max(1, self.n_envs * self.episode_limit))

    def _skip_finished(self):
        return self.compact_every > 0 and self.live_share < self.live_threshold

    def _new_round(self, new=False):
        obs = self.reset_fn() if self.reset_fn is not None else self.env.reset(new=new)
        E, n = self.n_envs, self.n_agents
        hidden = torch.zeros((E * n, self.agents.args.rnn_hidden_dim), device=self.device)
        last_action = torch.zeros((E, n, self.n_actions), dtype=torch.int8, device=self.device)
        return obs, hidden, last_action

    _capturing = False

    @torch.no_grad()
    def _play_graphed(self, epsilon, evaluate, record):
        """_play through a captured HIP graph (one graph per (evaluate, record) mode).  epsilon lives in
        a static device tensor that the graph reads and (when annealing) updates in place."""
        key = (bool(evaluate), bool(record), self._skip_finished())
        g = self._graphs.get(key)
        if g is None:
            eps_in = torch.zeros((), device=self.device)
            eps_in.copy_(torch.as_tensor(epsilon, device=self.device, dtype=torch.float32))
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):  # warm-up outside capture (lazy inits, rocBLAS handles)
                self._play(eps_in.clone(), evaluate, record)
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            self._capturing = True
            try:
                with torch.cuda.graph(graph):
                    out = self._play(eps_in, evaluate, record)
            finally:
                self._capturing = False
            g = {'graph': graph, 'eps_in': eps_in, 'out': out, 'last_played': self.last_played}
            self._graphs[key] = g
        g['eps_in'].copy_(torch.as_tensor(epsilon, device=self.device, dtype=torch.float32))
        g['graph'].replay()
        self.last_played = g['last_played']  # the static tensor of THIS graph (several graphs exist: evaluate / record / live list)
        return g['out']

    def _ops(self):
        if self._rollout_lib is None:
            from .. import _lib
            self._rollout_lib = _lib.rollout_ops()
            self._draw = torch.zeros(1, dtype=torch.int32, device=self.device)     # Philox draw counter (device side)
            self._n_alive = torch.zeros(4, dtype=torch.int32, device=self.device)  # [0] = live chips; [1..3] kernel workspace
        return self._rollout_lib

    @torch.no_grad()
    def _play(self, epsilon, evaluate, record):
        """One episode on every chip.  Returns per-chip stats and (if record) the episode batch.
        Per lock-step: Q-net forward, rollout_select_actions, the fused env transition, rollout_post_step
        (include/rollout_ops.h) -- the episode tensors are written in place by those kernels."""
        import ctypes as C
        E, n, A, T = self.n_envs, self.n_agents, self.n_actions, self.episode_limit
        dev = self.device
        lib = self._ops()
        vp = C.c_void_p
        stream = vp(torch.cuda.current_stream(dev).cuda_stream)
        obs, hidden, last_action = self._new_round()
        alive = torch.ones(E, dtype=torch.uint8, device=dev)
        reward = torch.zeros(E, dtype=torch.float64, device=dev)
        steps = torch.zeros(E, dtype=torch.int64, device=dev)
        constraints = torch.zeros(E, dtype=torch.float64, device=dev)  # MEDA reports a float (sum of punishments)
        success = torch.zeros(E, dtype=torch.int64, device=dev)
        actions = torch.empty((E, n), dtype=torch.int32, device=dev)
        eps = torch.as_tensor(epsilon, dtype=torch.float32, device=dev).reshape(1).clone()
        anneal = 0.0
        if not evaluate and self.agents.args.epsilon_anneal_scale == 'step':
            anneal = float(self.anneal_epsilon)
        min_eps = float(getattr(self, 'min_epsilon', 0.0))
        ep = None
        null = vp(None)
        p_u = p_oh = p_r = p_pad = p_term = p_o = p_on = null
        if record:
            O = self.env.obs_len
            ep = {'o': torch.zeros((E, T, n, O), dtype=torch.int8, device=dev),
                  'u': torch.zeros((E, T, n, 1), dtype=torch.int8, device=dev),
                  'r': torch.zeros((E, T, 1), dtype=torch.float32, device=dev),
                  'o_next': torch.zeros((E, T, n, O), dtype=torch.int8, device=dev),
                  'avail_u': torch.zeros((E, T, n, A), dtype=torch.int8, device=dev),
                  'avail_u_next': torch.zeros((E, T, n, A), dtype=torch.int8, device=dev),
                  'u_onehot': torch.zeros((E, T, n, A), dtype=torch.int8, device=dev),
                  'padded': torch.ones((E, T, 1), dtype=torch.bool, device=dev),
                  'terminated': torch.ones((E, T, 1), dtype=torch.bool, device=dev)}
            p_u, p_oh, p_r = vp(ep['u'].data_ptr()), vp(ep['u_onehot'].data_ptr()), vp(ep['r'].data_ptr())
            p_pad, p_term = vp(ep['padded'].data_ptr()), vp(ep['terminated'].data_ptr())
            # o / o_next are appended by rollout_post_step with the padding rule applied (frozen chips keep zero rows)
            p_o, p_on = vp(ep['o'].data_ptr()), vp(ep['o_next'].data_ptr())
            ep['o'][:, 0] = obs
        net = self.agents.policy.eval_rnn
        fused_tail = (self.fuse_tail and hasattr(net, 'act_ok') and net.act_ok(obs.reshape(E * n, -1)) and hidden.is_contiguous()
                      and hidden.dtype == torch.float32 and hidden.shape[1] == 128 and net.fc1.weight.is_contiguous())
        t_played = 0
        # the GRU input projection runs against rnn.weight_ih zero-padded to K = 640 / 832: one in-place copy per episode
        w_ih_pad = net.refresh_padded() if fused_tail else None
        live = (fused_tail and self._skip_finished() and hasattr(net, 'front_features_live') and net._hip_geometry() == 9
                and last_action.dtype == torch.int8)
        if live:
            live_chips = torch.empty(E, dtype=torch.int32, device=dev)
            n_live = torch.zeros(1, dtype=torch.int32, device=dev)
            x_live = torch.empty((E * n, net.padded_cols()), dtype=torch.float32, device=dev)  # compact rows (rows beyond the live ones: unused)
            if lib.rollout_compact_alive(E, vp(alive.data_ptr()), vp(live_chips.data_ptr()), vp(n_live.data_ptr()), stream) != 0:
                raise RuntimeError('rollout_compact_alive failed (hip %d)' % lib.rollout_last_hip_error())
        for t in range(T):
            obs2, la2 = obs.reshape(E * n, -1), last_action.reshape(E * n, -1)
            if live:
                # the same, for the listed chips only: x and x W_ih^T in compact row order, everything else in chip order
                net.front_features_live(obs2, la2, live_chips, n_live, n, x_live)
                ig, hg = torch.matmul(x_live, w_ih_pad.t()), torch.matmul(hidden, net.rnn.weight_hh.t())
                rc = lib.rollout_gru_head_select_live(vp(ig.data_ptr()), vp(hg.data_ptr()), vp(net.rnn.bias_ih.data_ptr()),
                                                      vp(net.rnn.bias_hh.data_ptr()), vp(hidden.data_ptr()), vp(net.fc1.weight.data_ptr()),
                                                      vp(net.fc1.bias.data_ptr()), E, n, hidden.shape[1], A, vp(eps.data_ptr()),
                                                      int(bool(evaluate)), self.rng_seed, vp(self._draw.data_ptr()),
                                                      vp(actions.data_ptr()), vp(last_action.data_ptr()), p_u, p_oh, T, t, null,
                                                      vp(live_chips.data_ptr()), vp(n_live.data_ptr()), stream)
            elif fused_tail:
                # front end + the two GRU GEMMs, then gate math + fc1 + epsilon-greedy in one launch (h updated in place)
                ig, hg = net.act_gates(obs2, la2, hidden, w_ih_pad)
                rc = lib.rollout_gru_head_select(vp(ig.data_ptr()), vp(hg.data_ptr()), vp(net.rnn.bias_ih.data_ptr()),
                                                 vp(net.rnn.bias_hh.data_ptr()), vp(hidden.data_ptr()), vp(net.fc1.weight.data_ptr()),
                                                 vp(net.fc1.bias.data_ptr()), E, n, hidden.shape[1], A, vp(eps.data_ptr()),
                                                 int(bool(evaluate)), self.rng_seed, vp(self._draw.data_ptr()),
                                                 vp(actions.data_ptr()), vp(last_action.data_ptr()), p_u, p_oh, T, t, null, stream)
            else:
                q, hidden = net.forward_obs(obs2, la2, hidden)
                q = q.contiguous()
                rc = lib.rollout_select_actions(vp(q.data_ptr()), E, n, A, vp(eps.data_ptr()), int(bool(evaluate)), self.rng_seed,
                                                vp(self._draw.data_ptr()), vp(actions.data_ptr()), vp(last_action.data_ptr()),
                                                p_u, p_oh, T, t, stream)
            if rc != 0:
                raise RuntimeError('rollout action selection failed: %d (hip %d)' % (rc, lib.rollout_last_hip_error()))
            # frozen chips are not stepped: the kernel reports reward 0 / constraints 0 / success 0 / terminated 1
            u = self.uniforms_fn(t) if self.uniforms_fn is not None else None
            obs, _, _, info = self.env.step(actions, uniforms=u, active=alive, record=True)
            cons = info['constraints']
            rc = lib.rollout_post_step(E, T, t, vp(alive.data_ptr()), vp(info['terminated'].data_ptr()),
                                       vp(info['team_reward'].data_ptr()), vp(cons.data_ptr()), int(cons.dtype == torch.float64),
                                       vp(info['success'].data_ptr()), p_r, p_pad, p_term, vp(reward.data_ptr()),
                                       vp(constraints.data_ptr()), vp(success.data_ptr()), vp(steps.data_ptr()),
                                       vp(eps.data_ptr()), anneal, min_eps, vp(self._n_alive.data_ptr()),
                                       vp(self._draw.data_ptr()), vp(obs.data_ptr()), n * self.env.obs_len if record else 0,
                                       p_o, p_on, stream)
            if rc != 0:
                raise RuntimeError('rollout_post_step failed: %d (hip %d)' % (rc, lib.rollout_last_hip_error()))
            if live and (t + 1) % self.compact_every == 0 and t + 1 < T:
                if lib.rollout_compact_alive(E, vp(alive.data_ptr()), vp(live_chips.data_ptr()), vp(n_live.data_ptr()), stream) != 0:
                    raise RuntimeError('rollout_compact_alive failed (hip %d)' % lib.rollout_last_hip_error())
            t_played = t + 1
            if not self._capturing and (t + 1) % self.sync_every == 0 and int(self._n_alive[0].item()) == 0:
                break
        if record:  # padding rules of rollout.py:131-141 applied once: zeros, avail 0 where padded
            valid = ~ep['padded']                                   # (E, T, 1)
            v4 = valid.unsqueeze(-1)
            for key in ('u', 'u_onehot'):
                ep[key] *= v4
            ep['r'] *= valid
            ep['avail_u'][:] = v4
            ep['avail_u_next'][:] = v4
        self.last_played = steps.sum()  # env steps actually played this round (before the failure inflation below)
        steps = torch.where(success > 0, steps, torch.full_like(steps, self.episode_limit))
        return reward, steps, constraints, success, ep, eps.reshape(())

    def _generate_episode(self):
        """Greedy episode on every chip (rollout.py:41-67): per-chip reward, steps, constraints, success."""
        self.agents.policy.init_hidden(1)
        play = self._play_graphed if self.use_graph else self._play
        reward, steps, constraints, success, _, _ = play(0.0, evaluate=True, record=False)
        return reward, steps, constraints, success

    def evaluate(self, task_num):
        """`task_num` consecutive greedy episodes on every chip (the chips keep ageing between
        episodes, as the single reference chip does); means over all chips and episodes."""
        tot = [0.0, 0.0, 0.0, 0.0]
        keep = self.live_share  # the greedy episodes have their own live share; the training rollouts' statistic is put back
        self.live_share = getattr(self, '_eval_live_share', 1.0)
        for _ in range(task_num):
            out = self._generate_episode()
            self.note_played(int(self.last_played.item()))
            for k in range(4):
                tot[k] += float(out[k].double().mean().item())
        self._eval_live_share, self.live_share = self.live_share, keep
        if hasattr(self, 'stream_restart'):
            self.stream_restart()
        return tuple(v / task_num for v in tot)


class RolloutWorker(Evaluator):
    def __init__(self, env, agents, args):
        super().__init__(env, agents, args.episode_limit)
        self.n_actions = args.n_actions
        self.obs_shape = args.obs_shape[-1]
        self.epsilon_anneal_scale = args.epsilon_anneal_scale
        self.min_epsilon = args.min_epsilon
        self.anneal_epsilon = (args.epsilon - args.min_epsilon) / args.anneal_steps
        self.epsilon = torch.tensor(float(args.epsilon), device=self.device)

    def generate_episode(self):
        """One episode per chip.  Returns (reward[E], step[E], constraints[E], success[E], episode)
        where `episode` has the reference's keys with leading dimension E (rollout.py:101-150)."""
        self.agents.policy.init_hidden(1)
        epsilon = self.epsilon
        if self.epsilon_anneal_scale == 'episode':
            epsilon = torch.clamp(epsilon - self.anneal_epsilon * self.n_envs, min=self.min_epsilon)
        play = self._play_graphed if self.use_graph else self._play
        reward, steps, constraints, success, episode, epsilon = play(epsilon, evaluate=False, record=True)
        self.epsilon = epsilon.clone() if self.use_graph else epsilon
        return reward, steps, constraints, success, episode

    # ------------------------------------------------------------------ continuous rollout (every row live)
    def stream_ok(self):
        """The continuous rollout needs the fused lock-step tail (HIP conv front end + rollout_gru_head_select: hidden 128, the
        reference's CRNN) and a GPU env; anything else keeps the episode-per-round form."""
        net = self.agents.policy.eval_rnn
        probe = torch.zeros((1, self.env.obs_len), dtype=torch.int8, device=self.device)
        with torch.no_grad():
            return bool(self.fuse_tail and self.device.type == 'cuda' and hasattr(net, 'act_ok') and net.act_ok(probe)
                        and self.agents.args.rnn_hidden_dim == 128 and net.fc1.weight.is_contiguous()
                        and self.epsilon_anneal_scale == 'step')

    def _stream_state(self, buffer):
        st = getattr(self, '_stream', None)
        if st is not None and st.buffer is buffer:
            return st
        import ctypes as C
        from .. import _lib
        E, n, A, T, O, dev = self.n_envs, self.n_agents, self.n_actions, self.episode_limit, self.env.obs_len, self.device
        if buffer.episode_limit != T or buffer.obs_shape != O or buffer.device != dev:
            raise ValueError('replay buffer does not match the env (episode_limit / obs / device)')
        st = types.SimpleNamespace(buffer=buffer, started=False, graphs={})
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        # the observation is double-buffered: the transition of lock-step s reads obs[s & 1] (through the Q-network) and writes
        # obs[(s + 1) & 1], so that the first observation of an episode is still there when its first step is staged
        st.obs = [z((E, n, O), torch.int8), z((E, n, O), torch.int8)]
        # Reset inside the transition launch where the env offers the terminal observation as a second output (DMFB fused
        # launch, include/dmfb_vec.h: d_obs_terminal): two launches per lock-step less than step + reset + observe
        st.fused_reset = bool(getattr(self, 'stream_fused_reset', True) and hasattr(self.env._out, 'd_obs_terminal')
                              and hasattr(self.env, 'launch_shape') and E < self.env.launch_shape().get('split_min_envs', 0))
        st.obs_term = z((E, n, O), torch.int8) if st.fused_reset else None
        st.out = []
        for k in range(2):
            o = type(self.env._out)()
            C.memmove(C.byref(o), C.byref(self.env._out), C.sizeof(o))
            o.d_obs = st.obs[k].data_ptr()
            if st.fused_reset:
                o.d_obs_terminal = st.obs_term.data_ptr()
            st.out.append(o)
        st.hidden = z((E * n, self.agents.args.rnn_hidden_dim), torch.float32)
        st.last_action = z((E, n, A), torch.int8)
        st.actions = z((E, n), torch.int32)
        st.t_ep = z((2, E), torch.int32)   # double-buffered (include/rollout_ops.h: rollout_stream_step)
        st.o0, st.o_next = z((E, n * O), torch.int8), z((E, T, n * O), torch.int8)
        st.u, st.onehot, st.r = z((E, T, n), torch.int8), z((E, T, n, A), torch.int8), z((E, T), torch.float32)
        st.ep_acc, st.chip_acc = z((E, 3), torch.float64), z((E, 4), torch.int64)
        st.close_slot = torch.full((E,), -1, dtype=torch.int32, device=dev)
        st.eps = z((1,), torch.float32)
        st.state_alt = z((4,), torch.int64)
        st.stage = _lib.RolloutStage(st.t_ep.data_ptr(), st.o0.data_ptr(), st.o_next.data_ptr(), st.u.data_ptr(), st.onehot.data_ptr(),
                                     st.r.data_ptr(), st.ep_acc.data_ptr(), st.chip_acc.data_ptr(), st.close_slot.data_ptr(),
                                     st.state_alt.data_ptr())
        st.ring = buffer.ring_struct()
        self._stream = st
        return st

    def stream_restart(self):
        """Forget the episodes in flight: the next generate_steps resets every chip first (called after anything else -- a greedy
        evaluation, generate_episode -- has used the env)."""
        st = getattr(self, '_stream', None)
        if st is not None:
            st.started = False

    @torch.no_grad()
    def _play_stream(self, st, K):
        """K lock-steps of every chip: Q-network (front end, the two GRU GEMMs, gate math + fc1 + epsilon-greedy), the env
        transition, rollout_stream_step (staging + episode close), reset of the chips whose episode ended."""
        import ctypes as C
        E, n, A, T = self.n_envs, self.n_agents, self.n_actions, self.episode_limit
        lib = self._ops()
        vp = C.c_void_p
        dev = self.device
        stream = vp(torch.cuda.current_stream(dev).cuda_stream)
        net = self.agents.policy.eval_rnn
        env = self.env
        if not st.started:   # (outside the captured graph: the warm-up call comes first)
            if self.reset_fn is None:
                env.reset(obs=st.obs[0])
            else:
                st.obs[0].copy_(self.reset_fn())
            for t_ in (st.hidden, st.last_action, st.t_ep, st.ep_acc, st.chip_acc):
                t_.zero_()
            st.started = True
        anneal = float(self.anneal_epsilon)
        w_ih_pad = net.refresh_padded()
        cons_f64 = None
        for s in range(K):
            cur, nxt = st.obs[s & 1], st.obs[(s + 1) & 1]
            ig, hg = net.act_gates(cur.view(E * n, -1), st.last_action.view(E * n, -1), st.hidden, w_ih_pad)
            rc = lib.rollout_gru_head_select_stream(vp(ig.data_ptr()), vp(hg.data_ptr()), vp(net.rnn.bias_ih.data_ptr()),
                                                    vp(net.rnn.bias_hh.data_ptr()), vp(st.hidden.data_ptr()), vp(net.fc1.weight.data_ptr()),
                                                    vp(net.fc1.bias.data_ptr()), E, n, st.hidden.shape[1], A, vp(st.eps.data_ptr()), 0,
                                                    self.rng_seed, vp(self._draw.data_ptr()), vp(st.actions.data_ptr()),
                                                    vp(st.last_action.data_ptr()), vp(st.u.data_ptr()), vp(st.onehot.data_ptr()), T,
                                                    vp(st.t_ep[s & 1].data_ptr()), None, stream)
            if rc != 0:
                raise RuntimeError('rollout_gru_head_select_stream failed: %d (hip %d)' % (rc, lib.rollout_last_hip_error()))
            u = self.uniforms_fn(s) if self.uniforms_fn is not None else None
            _, _, _, info = env.step(st.actions, uniforms=u, record=True, autoreset=st.fused_reset, out=st.out[(s + 1) & 1])
            cons = info['constraints']
            cons_f64 = int(cons.dtype == torch.float64)
            rc = lib.rollout_stream_step(E, n, A, T, n * env.obs_len, st.hidden.shape[1], vp(cur.data_ptr()), vp(nxt.data_ptr()),
                                         vp(st.obs_term.data_ptr()) if st.fused_reset else None, vp(info['terminated'].data_ptr()), vp(info['team_reward'].data_ptr()), vp(cons.data_ptr()),
                                         cons_f64, vp(info['success'].data_ptr()), C.byref(st.stage), C.byref(st.ring), s & 1,
                                         vp(st.hidden.data_ptr()), vp(st.last_action.data_ptr()), vp(st.eps.data_ptr()), anneal,
                                         float(self.min_epsilon), vp(self._draw.data_ptr()), stream)
            if rc != 0:
                raise RuntimeError('rollout_stream_step failed: %d (hip %d)' % (rc, lib.rollout_last_hip_error()))
            if self.stream_step_hook is not None:   # tests: (lock-step, actions, terminated) before the chips are reset
                self.stream_step_hook(s, st.actions, info['terminated'])
            if not st.fused_reset:
                env.reset(mask=info['terminated'], obs=nxt)   # reset(new=False) of the chips whose episode ended (rollout.py:103)
        if K & 1:   # the double-buffered observation and ring state end in their second buffers
            st.obs[0].copy_(st.obs[1])
            st.buffer.ring_state.copy_(st.state_alt)
            st.t_ep[0].copy_(st.t_ep[1])

    stream_step_hook = None

    def generate_steps(self, buffer, n_steps=None):
        """`n_steps` lock-steps (default: episode_limit) of every chip, episodes that end on the way written into `buffer`'s ring
        (include/rollout_ops.h).  Returns a device tensor int64[4]: episodes closed in this call, their step count with the
        failure inflation of rollout.py:148-149 (what train.py:65 adds to time_steps), successes, env steps played."""
        K = int(n_steps or self.episode_limit)
        st = self._stream_state(buffer)
        self._ops()
        if not st.started:
            st.eps.copy_(torch.as_tensor(self.epsilon, dtype=torch.float32, device=self.device).reshape(1))
        if self.use_graph and self.uniforms_fn is None and self.stream_step_hook is None:
            g = st.graphs.get(K)
            if g is None:
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):  # warm-up outside capture; these K lock-steps count like any others
                    self._play_stream(st, K)
                torch.cuda.current_stream(self.device).wait_stream(side)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                self._capturing = True
                try:
                    with torch.cuda.graph(g):
                        self._play_stream(st, K)
                finally:
                    self._capturing = False
                st.graphs[K] = g
            else:
                g.replay()
        else:
            self._play_stream(st, K)
        out = st.chip_acc.sum(0)
        st.chip_acc.zero_()
        self.epsilon = st.eps[0].clone()
        return out
