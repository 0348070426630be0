"""Episode replay buffer of the reference (common/replay_buffer.py:5-75) kept RESIDENT IN HBM.

Same keys, shapes and integer dtypes; `r` is float32 (the reference stores float64 and casts to
float32 at learn time, policy/vdn.py:92, so the learner sees identical values).  Ring insertion
and uniform sampling WITH replacement follow the reference; sampling indices are drawn on the
device so `sample` never synchronises the host."""
import threading

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
