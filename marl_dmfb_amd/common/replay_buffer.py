"""Episode replay buffer of the reference (common/replay_buffer.py:5-75) kept RESIDENT IN HBM.

Same keys, shapes and integer dtypes; `r` is float32 (the reference stores float64 and casts to
float32 at learn time, policy/vdn.py:92, so the learner sees identical values).  Ring insertion
and uniform sampling WITH replacement follow the reference; sampling indices are drawn on the
device so `sample` never synchronises the host.

The continuous rollout (common/rollout.py: generate_steps) writes finished episodes into the ring ON THE DEVICE
(include/rollout_ops.h: rollout_stream_step); cursor, fill level and every slot's episode length then
live in device tensors (`ring_state`, `ring_len`, `ring_stats`).  `sync_host` brings them to the host in one transfer per round;
`draw` then picks the episodes of a learn on the HOST (uniform with replacement over the filled slots, as
common/replay_buffer.py:53) and knows their lengths, so the learn is sized exactly (agent/agent.py:51-61) without a read-back."""
import threading

import numpy as np
import torch


class ReplayBuffer:
    def __init__(self, args, device=None):
        self.args = args
        self.n_actions = args.n_actions
        self.n_agents = args.n_agents
        self.obs_shape = args.obs_shape[-1]
        self.size = args.buffer_size
        self.episode_limit = args.episode_limit
        if device is None:
            device = getattr(args, 'device', None) or ('cuda' if args.cuda else 'cpu')
        self.device = torch.device(device)
        self.current_idx = 0
        self.current_size = 0
        S, T, n, O, A, dev = self.size, self.episode_limit, self.n_agents, self.obs_shape, self.n_actions, self.device
        self.buffers = {
            'o': torch.empty((S, T, n, O), dtype=torch.int8, device=dev),
            'u': torch.empty((S, T, n, 1), dtype=torch.int8, device=dev),
            'r': torch.empty((S, T, 1), dtype=torch.float32, device=dev),
            'o_next': torch.empty((S, T, n, O), dtype=torch.int8, device=dev),
            'avail_u': torch.empty((S, T, n, A), dtype=torch.int8, device=dev),
            'avail_u_next': torch.empty((S, T, n, A), dtype=torch.int8, device=dev),
            'u_onehot': torch.empty((S, T, n, A), dtype=torch.int8, device=dev),
            'padded': torch.empty((S, T, 1), dtype=torch.bool, device=dev),
            'terminated': torch.empty((S, T, 1), dtype=torch.bool, device=dev),
        }
        self.lock = threading.Lock()
        self.generator = None
        # device-side ring bookkeeping (include/rollout_ops.h: rollout_ring) and its host mirror
        self.ring_len = torch.zeros(S, dtype=torch.int32, device=dev)          # valid steps per slot, 0 = never written
        self.ring_stats = torch.zeros((S, 4), dtype=torch.float64, device=dev)  # reward, steps (inflated), constraints, success
        self.ring_state = torch.zeros(4, dtype=torch.int64, device=dev)         # cursor, filled slots, episodes closed
        self.host_len = np.zeros(S, np.int32)
        self.host_closed = 0
        rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
        self.host_rng = np.random.default_rng([int(getattr(args, 'seed', 0) or 0), rank, 0x5A3])

    def ring_struct(self):
        from .. import _lib
        b = self.buffers
        return _lib.RolloutRing(self.size, b['o'].data_ptr(), b['o_next'].data_ptr(), b['u'].data_ptr(), b['u_onehot'].data_ptr(),
                                b['avail_u'].data_ptr(), b['avail_u_next'].data_ptr(), b['r'].data_ptr(), b['padded'].data_ptr(),
                                b['terminated'].data_ptr(), self.ring_len.data_ptr(), self.ring_stats.data_ptr(),
                                self.ring_state.data_ptr())

    def sync_host(self, extra=None):
        """ONE device -> host transfer: ring cursor / fill level / episode count, every slot's length, and `extra` (an int64
        device tensor the caller wants in the same trip).  Waits for the work queued so far (the rollout).  Returns `extra` as a
        list of ints."""
        parts = [self.ring_state, self.ring_len.to(torch.int64)]
        n_extra = 0
        if extra is not None:
            extra = extra.reshape(-1).to(torch.int64)
            n_extra = extra.numel()
            parts.append(extra)
        host = torch.cat(parts).cpu().numpy()
        self.current_idx, self.current_size, self.host_closed = int(host[0]), int(host[1]), int(host[2])
        self.host_len = host[4:4 + self.size].astype(np.int32)
        return [int(v) for v in host[4 + self.size:4 + self.size + n_extra]]

    def draw(self, batch_size):
        """Episodes of one learn, picked on the host from the mirror `sync_host` left: (slot indices, their lengths), sorted by
        length, longest first (stable).  Uniform with replacement over the filled slots (common/replay_buffer.py:53)."""
        if self.current_size < 1:
            raise RuntimeError('empty replay buffer')
        idx = self.host_rng.integers(0, self.current_size, int(batch_size))
        lens = self.host_len[idx]
        order = np.argsort(-lens, kind='stable')
        return idx[order], lens[order]

    def gather(self, idx):
        """The episode tensors of the slots `idx` (host integers), as `sample` returns them."""
        t = torch.as_tensor(np.ascontiguousarray(idx, dtype=np.int64)).to(self.device, non_blocking=True)
        return {key: buf[t] for key, buf in self.buffers.items()}

    def store_episode(self, episode_batch):
        batch_size = episode_batch['o'].shape[0]
        with self.lock:
            start = self.current_idx
            idxs = self._get_storage_idx(inc=batch_size)
            # the common case is one contiguous range of the ring: a plain slice copy (3 TB/s) instead of an indexed put
            if start + batch_size > self.size and start >= self.size:
                start = 0                                   # third branch of the wrap rule: restart at slot 0
            contiguous = start + batch_size <= self.size
            for key, buf in self.buffers.items():
                src = episode_batch[key]
                if not isinstance(src, torch.Tensor):
                    src = torch.as_tensor(src)
                if contiguous:
                    buf[start:start + batch_size].copy_(src)
                else:
                    buf[idxs] = src.to(device=self.device, dtype=buf.dtype)
            # the device-side bookkeeping of the ring follows (episode lengths from the padding flags)
            pad = episode_batch['padded']
            if not isinstance(pad, torch.Tensor):
                pad = torch.as_tensor(pad)
            lens = (pad.to(self.device).reshape(batch_size, -1) == 0).sum(1).to(torch.int32)
            if contiguous:
                self.ring_len[start:start + batch_size] = lens
            else:
                self.ring_len[idxs] = lens
            self.ring_state[0] = self.current_idx % self.size
            self.ring_state[1] = self.current_size

    def sample(self, batch_size):
        idx = torch.randint(0, self.current_size, (batch_size,), device=self.device, generator=self.generator)
        return {key: buf[idx] for key, buf in self.buffers.items()}

    def _get_storage_idx(self, inc=None):
        """Ring allocation with the reference's wrap rule (common/replay_buffer.py:58-75)."""
        inc = inc or 1
        if inc > self.size:
            raise ValueError('episode batch (%d) larger than the buffer (%d)' % (inc, self.size))
        if self.current_idx + inc <= self.size:
            idx = torch.arange(self.current_idx, self.current_idx + inc, device=self.device)
            self.current_idx += inc
        elif self.current_idx < self.size:
            overflow = inc - (self.size - self.current_idx)
            idx = torch.cat([torch.arange(self.current_idx, self.size, device=self.device),
                             torch.arange(0, overflow, device=self.device)])
            self.current_idx = overflow
        else:
            idx = torch.arange(0, inc, device=self.device)
            self.current_idx = inc
        self.current_size = min(self.size, self.current_size + inc)
        return idx
