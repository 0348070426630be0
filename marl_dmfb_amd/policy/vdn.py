"""VDN learner with the reference's API surface (policy/vdn.py:8-218): VDN(args) with
.learn(batch, max_episode_len, train_step, epsilon=None), .get_q_values, .init_hidden,
.eval_hidden/.target_hidden, .eval_rnn/.target_rnn, .save_model(train_step=None).

Differences in HOW (not in what is computed):
  * the batch stays on the device; there is no per-time-step host->device copy
    (reference: policy/vdn.py:134-165 builds and uploads the inputs every t);
  * the non-recurrent part of the Q-net (convs + vector MLP) runs once over all T steps, only the
    GRU cell + head are unrolled over time (reference: whole net T times, vdn.py:174-191);
  * the target net runs under no_grad (the reference detaches afterwards, vdn.py:117);
  * data parallel: when torch.distributed is initialised and args.dist is true the gradient of the
    un-normalised loss and the mask count travel in ONE flat all-reduce over RCCL, then every rank
    divides, clips and steps identically.
"""
import os

import torch

from ..network.base_net import CRNN, RNN
from ..network.vdn_net import VDNNet


def _t(x, device, dtype):
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(x)
    return x.to(device=device, dtype=dtype)


def _episode_slots(t):
    """Slots per episode of a (B, T', ...) tensor that is a [:, :T'] view of a contiguous (B, slots, ...) one; None if it
    is laid out any other way."""
    inner = 1
    for k in range(t.dim() - 1, 0, -1):
        if t.stride(k) != inner and t.shape[k] != 1:
            return None
        inner *= t.shape[k]
    per_step = inner // t.shape[1]
    if t.shape[0] == 1:
        return t.shape[1]
    if per_step == 0 or t.stride(0) % per_step or t.stride(0) // per_step < t.shape[1]:
        return None
    return t.stride(0) // per_step


class _TDLoss(torch.autograd.Function):
    """The TD-error block of VDN.learn (reference policy/vdn.py:104-123) as one HIP launch each way (include/vdn_ops.h):
    (time-major Q values of both nets, the sampled episode tensors as the replay buffer stores them) ->
    num = sum((mask * td_error) ** 2), mask.sum().  Gradient for the eval net's Q values only."""

    @staticmethod
    def forward(ctx, q_eval_tm, q_target_tm, u, r, avail_next, terminated, padded, T, gamma, bad=None):
        import ctypes as C
        from .. import _lib
        lib = _lib.vdn_ops()
        vp = C.c_void_p
        B, n, A = u.shape[0], u.shape[2], avail_next.shape[3]
        t_limit = _episode_slots(u)  # the tensors may be [:, :T] views of the sampled (B, episode_limit, ...) tensors
        q_eval_tm, q_target_tm = q_eval_tm.contiguous(), q_target_tm.contiguous()
        mtd = torch.empty(B * T, dtype=torch.float32, device=u.device)
        mask = torch.empty(B * T, dtype=torch.float32, device=u.device)
        stream = vp(torch.cuda.current_stream(u.device).cuda_stream)
        rc = lib.vdn_td_forward(vp(q_eval_tm.data_ptr()), vp(q_target_tm.data_ptr()), vp(u.data_ptr()), vp(r.data_ptr()),
                                vp(avail_next.data_ptr()), vp(terminated.data_ptr()), vp(padded.data_ptr()), B, T, t_limit, n, A,
                                float(gamma), vp(mtd.data_ptr()), vp(mask.data_ptr()),
                                None if bad is None else vp(bad.data_ptr()), stream)
        if rc != 0:
            raise RuntimeError('vdn_td_forward failed: %d (hip %d)' % (rc, lib.vdn_last_hip_error()))
        ctx.save_for_backward(mtd, mask, u)
        ctx.dims = (B, T, t_limit, n, A)
        num, mask_sum = (mtd * mtd).sum(), mask.sum()
        ctx.mark_non_differentiable(mask_sum)
        return num, mask_sum

    @staticmethod
    def backward(ctx, g_num, _g_mask):
        import ctypes as C
        from .. import _lib
        lib = _lib.vdn_ops()
        vp = C.c_void_p
        mtd, mask, u = ctx.saved_tensors
        B, T, t_limit, n, A = ctx.dims
        gq = torch.empty((T, B * n, A), dtype=torch.float32, device=u.device)
        g = g_num.reshape(1).to(torch.float32).contiguous()
        rc = lib.vdn_td_backward(vp(mtd.data_ptr()), vp(mask.data_ptr()), vp(u.data_ptr()), vp(g.data_ptr()), B, T, t_limit, n, A,
                                 vp(gq.data_ptr()), vp(torch.cuda.current_stream(u.device).cuda_stream))
        if rc != 0:
            raise RuntimeError('vdn_td_backward failed: %d (hip %d)' % (rc, lib.vdn_last_hip_error()))
        return gq, None, None, None, None, None, None, None, None, None


class _TDLossPacked(torch.autograd.Function):
    """`_TDLoss` on packed (episode, step) units (include/vdn_ops.h: vdn_td_forward_packed): the Q tensors hold only the valid steps
    of the length-sorted batch, the replay tensors are indexed in place through `units` (= slot * T + t)."""

    @staticmethod
    def forward(ctx, q_eval, q_target, units, n_units, u, r, avail_next, terminated, padded, n, A, gamma, bad=None):
        import ctypes as C
        from .. import _lib
        lib = _lib.vdn_ops()
        vp = C.c_void_p
        q_eval, q_target = q_eval.contiguous(), q_target.contiguous()
        mtd = torch.empty(n_units, dtype=torch.float32, device=u.device)
        mask = torch.empty(n_units, dtype=torch.float32, device=u.device)
        rc = lib.vdn_td_forward_packed(vp(q_eval.data_ptr()), vp(q_target.data_ptr()), vp(units.data_ptr()), n_units, vp(u.data_ptr()),
                                       vp(r.data_ptr()), vp(avail_next.data_ptr()), vp(terminated.data_ptr()), vp(padded.data_ptr()),
                                       n, A, float(gamma), vp(mtd.data_ptr()), vp(mask.data_ptr()),
                                       None if bad is None else vp(bad.data_ptr()), vp(torch.cuda.current_stream(u.device).cuda_stream))
        if rc != 0:
            raise RuntimeError('vdn_td_forward_packed failed: %d (hip %d)' % (rc, lib.vdn_last_hip_error()))
        ctx.save_for_backward(mtd, mask, units, u)
        ctx.dims = (n_units, n, A, q_eval.shape[0])
        num, mask_sum = (mtd * mtd).sum(), mask.sum()
        ctx.mark_non_differentiable(mask_sum)
        return num, mask_sum

    @staticmethod
    def backward(ctx, g_num, _g_mask):
        import ctypes as C
        from .. import _lib
        lib = _lib.vdn_ops()
        vp = C.c_void_p
        mtd, mask, units, u = ctx.saved_tensors
        n_units, n, A, rows_pad = ctx.dims
        gq = torch.empty((rows_pad, A), dtype=torch.float32, device=u.device)
        gq[n_units * n:].zero_()
        g = g_num.reshape(1).to(torch.float32).contiguous()
        rc = lib.vdn_td_backward_packed(vp(mtd.data_ptr()), vp(mask.data_ptr()), vp(units.data_ptr()), n_units, vp(u.data_ptr()),
                                        vp(g.data_ptr()), n, A, vp(gq.data_ptr()), vp(torch.cuda.current_stream(u.device).cuda_stream))
        if rc != 0:
            raise RuntimeError('vdn_td_backward_packed failed: %d (hip %d)' % (rc, lib.vdn_last_hip_error()))
        return (gq,) + (None,) * 12


PACK_ROWS = 2048  # packed row counts of 32 768 rows and more (where the split-K weight gradients apply) are rounded up to a multiple
                  # of this (zero rows): few distinct GEMM shapes, every one divisible by the 64 split-K chunks; smaller ones to 64


class VDN:
    def __init__(self, args):
        self.args = args
        self.n_actions = args.n_actions
        self.n_agents = args.n_agents
        input_shape = args.obs_shape[-1]
        if args.last_action:
            input_shape += self.n_actions
        if args.reuse_network:
            input_shape += self.n_agents
        if args.net == 'rnn':
            self.eval_rnn = RNN(args.obs_shape[-1] + (self.n_actions if args.last_action else 0), args)
            self.target_rnn = RNN(args.obs_shape[-1] + (self.n_actions if args.last_action else 0), args)
        elif args.net == 'crnn':
            self.eval_rnn = CRNN(args)
            self.target_rnn = CRNN(args)
        else:
            raise Exception('No such net')
        self.eval_vdn_net = VDNNet()
        self.target_vdn_net = VDNNet()
        if getattr(args, 'device', None) is not None:
            self.device = torch.device(args.device)
        else:
            self.device = torch.device('cuda', torch.cuda.current_device()) if args.cuda else torch.device('cpu')
        if self.device.type == 'cuda':  # (a CPU learner -- tests, bench.py's cpu_baseline workers -- never touches the GPU runtime)
            from ..common import gemm_tuning
            gemm_tuning.enable()  # before the first GEMM: shipped rocBLAS / hipBLASLt solution choices (no on-line tuning)
        for m in (self.eval_rnn, self.target_rnn, self.eval_vdn_net, self.target_vdn_net):
            m.to(self.device)

        self.model_dir = args.model_dir + '/' + args.alg + '/fov{}/'.format(args.fov)
        if args.load_model:
            path_rnn = self.model_dir + args.load_model_name + 'rnn_net_params.pkl'
            path_vdn = self.model_dir + args.load_model_name + 'vdn_net_params.pkl'
            if os.path.exists(path_rnn):
                self.eval_rnn.load_state_dict(torch.load(path_rnn, map_location=self.device, weights_only=True))
                if os.path.exists(path_vdn):
                    self.eval_vdn_net.load_state_dict(torch.load(path_vdn, map_location=self.device, weights_only=True))
                print('Successfully load the model: {} and {}'.format(path_rnn, path_vdn))
            else:
                raise Exception('No model!')

        self.target_rnn.load_state_dict(self.eval_rnn.state_dict())
        self.target_vdn_net.load_state_dict(self.eval_vdn_net.state_dict())
        for p in self.target_rnn.parameters():
            p.requires_grad_(False)

        self.eval_parameters = list(self.eval_vdn_net.parameters()) + list(self.eval_rnn.parameters())
        if args.optimizer == 'RMS':
            self.optimizer = torch.optim.RMSprop(self.eval_parameters, lr=args.lr)
        elif args.optimizer == 'SGD':
            self.optimizer = torch.optim.SGD(self.eval_parameters, lr=args.lr)
        elif args.optimizer == 'ADAM':
            # same update rule; on the GPU all parameter tensors are stepped by ONE fused kernel instead of ~8 foreach launches
            fused = {'fused': True} if self.device.type == 'cuda' else {}
            self.optimizer = torch.optim.Adam(self.eval_parameters, lr=args.lr, betas=(0.9, 0.99), **fused)
        elif args.optimizer == 'ASGD':
            self.optimizer = torch.optim.Adam(self.eval_parameters, lr=args.lr)
        else:
            raise Exception('No such optimizer')

        self.eval_hidden = None
        self.target_hidden = None
        self.last_loss = None
        self.last_grad_norm = None
        self._flat = None
        self._adam = None  # moments and step count of the two-launch clip + Adam step (_fused_step)
        self._td_bad = None  # device counter of (episode, step) slots whose action was outside [0, n_actions) (include/vdn_ops.h)
        # scalars a caller wants summed over the ranks without a collective of their own (Trainer: the env-step count of the
        # round): float32 tensor set before learn(); the next gradient all-reduce carries it and leaves the sums here
        self.ride_along = None
        self.ride_along_sum = None
        self.overlap_hook = None       # callable queued while the next gradient all-reduce is in flight (one shot)
        self.allreduce_events = None   # list -> (start, end) HIP event pair of every gradient all-reduce (diagnostics)
        self.dist = bool(getattr(args, 'dist', False)) and torch.distributed.is_available() \
            and torch.distributed.is_initialized() \
            and (torch.distributed.get_world_size() > 1 or bool(getattr(args, 'force_dist', False)))
        if self.dist:
            self.broadcast_parameters()

    # ------------------------------------------------------------------ data parallel
    def broadcast_parameters(self, src=0):
        """Same initial weights on every rank (one flat broadcast)."""
        params = list(self.eval_rnn.parameters())
        flat = torch.cat([p.data.reshape(-1) for p in params])
        if flat.is_cuda and torch.distributed.get_backend() == 'gloo':
            host = flat.cpu()
            torch.distributed.broadcast(host, src)
            flat = host.to(flat.device)
        else:
            torch.distributed.broadcast(flat, src)
        off = 0
        for p in params:
            p.data.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.target_rnn.load_state_dict(self.eval_rnn.state_dict())

    def all_reduce_sum(self, flat, overlap=None):
        """In-place SUM all-reduce of a flat tensor; returns it.  The collective is started asynchronously (RCCL runs it on its
        own stream behind the work already queued on this one), `overlap()` -- work that does not depend on the result -- is
        queued meanwhile, then this stream waits for the collective (no host block with RCCL)."""
        dist = torch.distributed
        if flat.is_cuda and dist.get_backend() == 'gloo':
            # rehearsal only (several ranks sharing one GPU over gloo): stage through the host
            host = flat.cpu()
            work = dist.all_reduce(host, op=dist.ReduceOp.SUM, async_op=True)
            if overlap is not None:
                overlap()
            work.wait()
            flat.copy_(host)
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
            if overlap is not None:
                overlap()
            work.wait()
        return flat

    def _allreduce_grads(self, mask_sum):
        """ONE collective per learn step: [all gradients of the un-normalised loss, mask count, ride-along scalars].  Leaves the
        SUMMED gradients in p.grad and returns the summed mask count (device scalar).
        Flatten = one concatenation, un-flatten = one multi-tensor copy, so a rank adds ~3 launches to the
        all-reduce.  `overlap_hook` (set by the Trainer: the NEXT learn's replay sample, which does not depend on this learn)
        is queued while the collective is in flight; with `allreduce_events` set to a list every collective is bracketed by a
        HIP event pair on the compute stream (bench.py: allreduce_ms_per_learn)."""
        params = [p for p in self.eval_parameters if p.grad is not None]
        n = sum(p.numel() for p in params)
        parts = [p.grad.reshape(-1) for p in params] + [mask_sum.reshape(1).to(torch.float32)]
        extra = self.ride_along
        if extra is not None:
            parts.append(extra.reshape(-1).to(device=mask_sum.device, dtype=torch.float32))
        flat = torch.cat(parts)
        timed = self.allreduce_events is not None and flat.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        hook, self.overlap_hook = self.overlap_hook, None
        flat = self.all_reduce_sum(flat, hook)
        if timed:
            e1.record()
            self.allreduce_events.append((e0, e1))
        if extra is not None:
            self.ride_along_sum = flat[n + 1:].clone()
            self.ride_along = None
        total = flat[n].clone()   # the global mask count: the caller divides by it (in the clip + Adam kernel: _fused_step)
        views, off = [], 0
        for p in params:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        torch._foreach_copy_([p.grad for p in params], views)
        return total

    # ------------------------------------------------------------------ learn (policy/vdn.py:79-132)
    def learn(self, batch, max_episode_len, train_step, epsilon=None):
        dev, T, n = self.device, max_episode_len, self.n_agents
        episode_num = batch['o'].shape[0]
        self.init_hidden(episode_num)
        if self._td_fused_ok(batch):
            # replay-buffer tensors on the GPU: Q values stay time-major, the TD block is one launch each way
            q_e, q_t = self.get_q_values(batch, T, time_major=True)
            if self._td_bad is None:
                self._td_bad = torch.zeros(1, dtype=torch.int32, device=dev)
            num, mask_sum = _TDLoss.apply(q_e, q_t, batch['u'], batch['r'], batch['avail_u_next'], batch['terminated'],
                                          batch['padded'], T, self.args.gamma, self._td_bad)
            return self._backward_and_step(num, mask_sum, train_step)
        u = _t(batch['u'], dev, torch.long)[:, :T]
        r = _t(batch['r'], dev, torch.float32)[:, :T]
        avail_u_next = _t(batch['avail_u_next'], dev, torch.float32)[:, :T]
        terminated = _t(batch['terminated'], dev, torch.float32)[:, :T]
        mask = 1 - _t(batch['padded'], dev, torch.float32)[:, :T]

        q_evals, q_targets = self.get_q_values(batch, T)

        q_evals = torch.gather(q_evals, dim=3, index=u).squeeze(3)
        q_targets = q_targets.masked_fill(avail_u_next == 0.0, -9999999)
        q_targets = q_targets.max(dim=3)[0]

        q_total_eval = self.eval_vdn_net(q_evals)
        q_total_target = self.target_vdn_net(q_targets)
        targets = r + self.args.gamma * q_total_target * (1 - terminated)
        td_error = targets.detach() - q_total_eval
        masked_td_error = mask * td_error

        self.optimizer.zero_grad()
        if self.dist:
            num = (masked_td_error ** 2).sum()
            num.backward()
            total = self._allreduce_grads(mask.sum())
            return self._step_and_sync(num.detach() / total, train_step, grad_div=total)
        loss = (masked_td_error ** 2).sum() / mask.sum()
        loss.backward()
        return self._step_and_sync(loss, train_step)

    def _backward_and_step(self, num, mask_sum, train_step):
        """loss = num / mask_sum (policy/vdn.py:122), backward, clip, step.  The un-normalised num is differentiated and the
        division by the mask count happens inside the clip + Adam kernel (include/vdn_ops.h: d_grad_div), so that one rank and
        many ranks (where the count is the all-reduced one) run the same arithmetic: a one-rank data-parallel run reproduces the
        plain run bit for bit (tests/test_gpu_dist_learn.py)."""
        self.optimizer.zero_grad()
        num.backward()
        total = self._allreduce_grads(mask_sum) if self.dist else mask_sum
        return self._step_and_sync(num.detach() / total, train_step, grad_div=total)

    def packed_ok(self, buffers):
        """learn_packed applies to the replay ring on the GPU (int8 / float32 / bool episode tensors), the CRNN with one of the two
        HIP front ends (fov 9; fov 19 = MEDA) and the GRU sequence kernels, and the parameter-free VDN mixer."""
        import ctypes  # noqa: F401
        net = self.eval_rnn
        o = buffers.get('o')
        if not (isinstance(o, torch.Tensor) and o.is_cuda and o.dtype == torch.int8 and o.is_contiguous() and self.args.alg == 'vdn'
                and self.args.last_action and hasattr(net, 'recurrent_seq_packed') and getattr(net, 'rnn_hidden_dim', 0) == 128
                and len(list(self.eval_vdn_net.parameters())) == 0 and o.shape[1] <= 255):
            return False
        want = {'u': torch.int8, 'r': torch.float32, 'avail_u_next': torch.int8, 'terminated': torch.bool, 'padded': torch.bool,
                'u_onehot': torch.int8, 'o_next': torch.int8}
        if any(not (isinstance(buffers.get(k), torch.Tensor) and buffers[k].dtype == dt and buffers[k].is_contiguous()) for k, dt in want.items()):
            return False
        probe = o.view(-1, o.shape[-1])[:1]
        with torch.no_grad():
            return bool(hasattr(net, '_hip_conv_ok') and net._hip_conv_ok(probe) and net._hip_geometry() in (9, 19)
                        and hasattr(net, '_hip_train_ok'))

    @staticmethod
    def pack_units(idx, lens, t_ring):
        """Host side of learn_packed: (counts per step, unit list) of a length-sorted draw.  counts[t] = episodes longer than t;
        units = for t = 0, 1, ...: slot * t_ring + t of those episodes (int32)."""
        import numpy as np
        idx, lens = np.asarray(idx, np.int64), np.asarray(lens, np.int64)
        counts = (lens[None, :] > np.arange(int(lens[0]))[:, None]).sum(1)
        units = np.concatenate([idx[:c] * t_ring + t for t, c in enumerate(counts)]).astype(np.int32)
        return counts, units

    def learn_packed(self, buffers, idx, lens, train_step, plan=None):
        """VDN.learn (policy/vdn.py:79-132) on the episodes in slots `idx` of the replay tensors `buffers`, WITHOUT their padded
        steps.  idx / lens: host integer arrays sorted by episode length, longest first (ReplayBuffer.draw).  The reference trims
        the batch to its longest episode (agent/agent.py:51-70) and multiplies the TD error of a padded step by 0
        (policy/vdn.py:118-122); here those steps are never computed: the valid (episode, step) units are laid out step after step
        (step t: the episodes longer than t), the conv front end, the GRU input projection and the head run on exactly those rows,
        the GRU sequence kernels stop every row at its own length, and the TD block indexes the replay tensors in place."""
        import numpy as np
        dev, n, A = self.device, self.n_agents, self.n_actions
        B, T_ring, O = len(idx), buffers['o'].shape[1], buffers['o'].shape[-1]
        if plan is None:   # (counts, device unit list) may come from the caller, who uploads the lists of a round's learns in one go
            counts, units_np = self.pack_units(idx, lens, T_ring)
            units = torch.from_numpy(units_np).to(dev, non_blocking=True)
        else:
            counts, units = plan
        U = int(units.shape[0])
        V = U * n
        pad = PACK_ROWS if V >= 32768 else 64
        Vp = -(-V // pad) * pad
        import ctypes as C
        from .. import _lib
        lib = _lib.vdn_ops()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

        def packed(src, shift=0, zero_below=0):   # (Vp, row) copy of the units' rows; rows V .. Vp-1 are zeros
            row = src.shape[-1]
            out = torch.empty((Vp, row), dtype=src.dtype, device=dev)
            out[V:].zero_()
            rc = lib.vdn_gather_units(C.c_void_p(src.data_ptr()), n * row * src.element_size(), C.c_void_p(units.data_ptr()), U, shift,
                                      zero_below, C.c_void_p(out.data_ptr()), stream)
            if rc != 0:
                raise RuntimeError('vdn_gather_units failed: %d (hip %d)' % (rc, lib.vdn_last_hip_error()))
            return out
        obs_e, obs_t, oh_t = packed(buffers['o']), packed(buffers['o_next']), packed(buffers['u_onehot'])
        oh_e = packed(buffers['u_onehot'], shift=-1, zero_below=B)   # last action of step t = u_onehot[t - 1]; zeros at t == 0 (vdn.py:150-160)
        x_e = self._features(self.eval_rnn, obs_e, oh_e)
        with torch.no_grad():
            x_t = self._features(self.target_rnn, obs_t, oh_t)
        step_rows = [int(c) * n for c in counts]
        if os.environ.get('MARL_DMFB_GRU_PAIR', '1') != '0' and hasattr(self.eval_rnn, 'recurrent_seq_packed_pair'):
            # the two networks' recurrences in one launch on the matrix cores (include/crnn_ops.h: gru_seq_forward_packed_pair)
            q_e, q_t = self.eval_rnn.recurrent_seq_packed_pair(self.eval_rnn, x_e, self.target_rnn, x_t, step_rows, B * n)
        else:
            q_e = self.eval_rnn.recurrent_seq_packed(x_e, step_rows, B * n)
            with torch.no_grad():
                q_t = self.target_rnn.recurrent_seq_packed(x_t, step_rows, B * n)
        if self._td_bad is None:
            self._td_bad = torch.zeros(1, dtype=torch.int32, device=dev)
        num, mask_sum = _TDLossPacked.apply(q_e, q_t, units, U, buffers['u'], buffers['r'], buffers['avail_u_next'],
                                            buffers['terminated'], buffers['padded'], n, A, self.args.gamma, self._td_bad)
        return self._backward_and_step(num, mask_sum, train_step)

    def check_td_inputs(self):
        """Raises if any learn since the last call met an action outside [0, n_actions) -- the input torch.gather raises on in
        the reference (policy/vdn.py:106).  The fused TD kernel never indexes with such a value; it poisons that learn's loss
        with NaN and counts the slot.  Reading the counter synchronises, so this runs at checkpoints (save_model), not per learn."""
        if self._td_bad is not None:
            bad = int(self._td_bad.item())
            if bad:
                self._td_bad.zero_()
                raise RuntimeError('VDN.learn: %d (episode, step) slots with an action outside [0, %d)' % (bad, self.n_actions))

    def _fused_step(self, grad_div=None):
        """clip_grad_norm_ + Adam.step as two launches (include/vdn_ops.h: vdn_clip_adam_step) instead of torch's ~11 small
        ones; same formulas.  Returns False (torch path) when it does not apply: not the GPU Adam of policy/vdn.py:67-68, a
        non-float32 / non-contiguous tensor, more tensors than the C ABI takes.  The moments live here, not in
        self.optimizer.state (the reference never saves optimizer state: policy/vdn.py:167-174)."""
        a = self.args
        params = [p for p in self.eval_parameters if p.grad is not None]
        ok = (self.device.type == 'cuda' and a.optimizer == 'ADAM' and getattr(a, 'fused_clip_adam', True) and 0 < len(params) <= 32
              and all(p.dtype == torch.float32 and p.grad.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                      for p in params))
        # the Adam moments live either here or in self.optimizer.state: a learner that changed sides mid-run would silently
        # restart them, so the choice of the first step is binding
        if getattr(self, '_fused_choice', None) is None:
            self._fused_choice = ok
        elif self._fused_choice != ok:
            raise RuntimeError('VDN: the clip + Adam step changed from %s to %s between two learns (gradient layout / dtype / set of '
                               'tensors changed); the two keep separate Adam moments' % (('torch', 'fused')[self._fused_choice], ('torch', 'fused')[ok]))
        if not ok:
            return False
        import ctypes as C
        from .. import _lib
        lib = _lib.vdn_ops()
        st = self._adam
        if st is None:
            st = self._adam = {'step': 0, 'm': {}, 'v': {}, 'partials': torch.empty(128, dtype=torch.float32, device=self.device),
                               'norm': torch.zeros(1, dtype=torch.float32, device=self.device)}
        for p in params:
            if id(p) not in st['m']:
                st['m'][id(p)] = torch.zeros_like(p)
                st['v'][id(p)] = torch.zeros_like(p)
        st['step'] += 1
        group = self.optimizer.param_groups[0]
        b1, b2 = group['betas']
        n = len(params)
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        numel = (C.c_int64 * n)(*[p.numel() for p in params])
        rc = lib.vdn_clip_adam_step(n, arr(params), arr([p.grad for p in params]), arr([st['m'][id(p)] for p in params]),
                                    arr([st['v'][id(p)] for p in params]), numel, float(a.grad_norm_clip), float(group['lr']),
                                    float(b1), float(b2), float(group['eps']), 1.0 - b1 ** st['step'], 1.0 - b2 ** st['step'],
                                    C.c_void_p(st['partials'].data_ptr()), C.c_void_p(st['norm'].data_ptr()),
                                    None if grad_div is None else C.c_void_p(grad_div.data_ptr()),
                                    C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise RuntimeError('vdn_clip_adam_step failed: %d (hip %d)' % (rc, lib.vdn_last_hip_error()))
        self.last_grad_norm = st['norm'][0]
        return True

    def _step_and_sync(self, loss, train_step, grad_div=None):
        """clip_grad_norm_, optimizer step, hard target sync every target_update_cycle learns (policy/vdn.py:125-132).  grad_div
        (device scalar): p.grad holds the gradient of the un-normalised loss and is divided by it first."""
        if grad_div is not None:
            grad_div = grad_div.reshape(1).to(torch.float32)
        if not self._fused_step(grad_div):
            if grad_div is not None:
                torch._foreach_div_([p.grad for p in self.eval_parameters if p.grad is not None], grad_div.reshape(()))
            self.last_grad_norm = torch.nn.utils.clip_grad_norm_(self.eval_parameters, self.args.grad_norm_clip)
            self.optimizer.step()
        self.last_loss = loss.detach()

        if train_step > 0 and train_step % self.args.target_update_cycle == 0:
            self.target_rnn.load_state_dict(self.eval_rnn.state_dict())
            self.target_vdn_net.load_state_dict(self.eval_vdn_net.state_dict())
        return self.last_loss

    def _td_fused_ok(self, batch):
        """The fused TD block (include/vdn_ops.h) applies to what ReplayBuffer.sample hands over on the GPU: device tensors
        in the buffer's dtypes, and networks with the time-major sequence path; the mixer must be the parameter-free VDN sum."""
        want = {'u': torch.int8, 'r': torch.float32, 'avail_u_next': torch.int8, 'terminated': torch.bool, 'padded': torch.bool}
        slots = set()
        for key, dt in want.items():
            t = batch.get(key)
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dt and t.dim() >= 3):
                return False
            slots.add(_episode_slots(t))
        if len(slots) != 1 or None in slots:
            return False
        return (hasattr(self.eval_rnn, 'recurrent_seq') and self.args.alg == 'vdn' and batch['avail_u_next'].shape[3] <= 127
                and len(list(self.eval_vdn_net.parameters())) == 0)

    def _sequence(self, batch, T):
        """Inputs for t = 0..T (policy/vdn.py:134-165): obs_seq[t] = o[:,0] if t == 0 else
        o_next[:,t-1]; last-action one-hot is zero at t == 0, else u_onehot[:,t-1].  Returned
        time-major: obs (T+1, B*n, obs) int8/float, last action (T+1, B*n, A)."""
        dev = self.device
        o = _t(batch['o'], dev, batch['o'].dtype if isinstance(batch['o'], torch.Tensor) else torch.float32)
        o_next = _t(batch['o_next'], dev, o.dtype)
        B = o.shape[0]
        # written time-major directly (one transposing copy; no intermediate (B, T+1, ...) concatenation)
        obs_seq = torch.empty((T + 1, B) + tuple(o.shape[2:]), dtype=o.dtype, device=dev)
        obs_seq[0].copy_(o[:, 0])
        obs_seq[1:].copy_(o_next[:, :T].permute(1, 0, 2, 3))
        obs_seq = obs_seq.view(T + 1, B * self.n_agents, -1)
        la = None
        if self.args.last_action:
            src = batch['u_onehot']
            keep = isinstance(src, torch.Tensor) and src.dtype == torch.int8 and o.dtype == torch.int8
            uo = _t(src, dev, torch.int8 if keep else torch.float32)   # int8 stays int8 for the HIP front end
            la = torch.empty((T + 1, B) + tuple(uo.shape[2:]), dtype=uo.dtype, device=dev)
            la[0].zero_()
            la[1:].copy_(uo[:, :T].permute(1, 0, 2, 3))
            la = la.view(T + 1, B * self.n_agents, -1)
        return obs_seq, la

    def _features(self, net, obs_rows, la_rows):
        if la_rows is not None and hasattr(net, '_hip_conv_ok') and net._hip_conv_ok(obs_rows):
            # no-grad pass (target net) over int8 rows: hand-written HIP conv front end
            return net._front_features_hip(obs_rows, la_rows, padded=hasattr(net, 'recurrent_seq'))
        if la_rows is not None and hasattr(net, '_hip_train_ok') and net._hip_train_ok(obs_rows):
            return net.features_obs_train(obs_rows, la_rows)  # eval net: HIP conv forward + backward
        x = obs_rows.float()
        if la_rows is not None:
            x = torch.cat([x, la_rows.float()], dim=1)
        return net.features(x)

    def get_q_values(self, batch, max_episode_len, time_major=False):
        """(q_evals, q_targets) as (B, T, n, A) like the reference (policy/vdn.py:167-203); time_major=True returns the
        (T, B*n, A) tensors the GRU sequence kernels produce, without the transposing views (None if that path is not taken)."""
        T, n = max_episode_len, self.n_agents
        B = batch['o'].shape[0]
        obs_seq, la = self._sequence(batch, T)
        R = B * n
        # eval net sees inputs 0..T-1, target net inputs 1..T, each from a zero hidden state
        x_eval = self._features(self.eval_rnn, obs_seq[:T].reshape(T * R, -1),
                                None if la is None else la[:T].reshape(T * R, -1)).view(T, R, -1)
        with torch.no_grad():
            x_tgt = self._features(self.target_rnn, obs_seq[1:T + 1].reshape(T * R, -1),
                                   None if la is None else la[1:T + 1].reshape(T * R, -1)).view(T, R, -1)
        self.eval_hidden = self.eval_hidden.to(self.device).reshape(R, -1)
        self.target_hidden = self.target_hidden.to(self.device).reshape(R, -1)
        if hasattr(self.eval_rnn, 'recurrent_seq'):
            q_e, self.eval_hidden = self.eval_rnn.recurrent_seq(x_eval, self.eval_hidden)
            with torch.no_grad():
                q_t, self.target_hidden = self.target_rnn.recurrent_seq(x_tgt, self.target_hidden)
            # (T, B*n, A) -> (B, T, n, A)
            if time_major:
                return q_e, q_t
            return (q_e.view(T, B, n, -1).permute(1, 0, 2, 3), q_t.view(T, B, n, -1).permute(1, 0, 2, 3))
        if time_major:
            return None
        q_evals, q_targets = [], []
        for t in range(T):
            q_eval, self.eval_hidden = self.eval_rnn.recurrent(x_eval[t], self.eval_hidden)
            with torch.no_grad():
                q_target, self.target_hidden = self.target_rnn.recurrent(x_tgt[t], self.target_hidden)
            q_evals.append(q_eval.view(B, n, -1))
            q_targets.append(q_target.view(B, n, -1))
        return torch.stack(q_evals, dim=1), torch.stack(q_targets, dim=1)

    def init_hidden(self, episode_num):
        shape = (episode_num, self.n_agents, self.args.rnn_hidden_dim)
        self.eval_hidden = torch.zeros(shape, device=self.device)
        self.target_hidden = torch.zeros(shape, device=self.device)

    def save_model(self, train_step=None):
        """File names of the reference (policy/vdn.py:205-218): {i}_[{k}_]rnn_net_params.pkl and the
        (empty) mixer state dict."""
        self.check_td_inputs()
        if not os.path.exists(self.model_dir):
            os.makedirs(self.model_dir)
        i = self.args.ith_run
        tag = str(i) + '_' if train_step is None else str(i) + '_' + str(train_step) + '_'
        torch.save(self.eval_vdn_net.state_dict(), self.model_dir + tag + 'vdn_net_params.pkl')
        torch.save(self.eval_rnn.state_dict(), self.model_dir + tag + 'rnn_net_params.pkl')
