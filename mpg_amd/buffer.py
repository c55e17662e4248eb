"""ReplayBuffer - device-resident mirror of buffer.py:21-91: same method names (`add_batch`, `replay`, `sample`,
`__len__`), ring semantics and uniform sampling with replacement; storage is a set of device arrays and the
ring/gather/index kernels are HIP (mpg_replay_*).  Replay shards are per GPU (no exchange)."""
import torch

from . import _lib as L


class ReplayBuffer(object):
    def __init__(self, args, buffer_id, device='cuda', obs_dim=None, act_dim=None):
        self.args = args
        self.buffer_id = buffer_id
        self.device = torch.device(device)
        self._maxsize = int(args.max_buffer_size)
        self.replay_starts = int(args.replay_starts)
        self.replay_batch_size = int(args.replay_batch_size)
        od = obs_dim if obs_dim is not None else args.obs_dim
        ad = act_dim if act_dim is not None else args.act_dim
        self.obs_dim, self.act_dim = od, ad
        n = self._maxsize
        f = dict(dtype=torch.float32, device=self.device)
        self.obs, self.obs2 = torch.zeros(n, od, **f), torch.zeros(n, od, **f)
        self.act, self.rew = torch.zeros(n, ad, **f), torch.zeros(n, **f)
        self.done = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self._size = 0
        self._next_idx = 0
        self.seed = int(getattr(args, 'seed', 0)) * 7919 + int(buffer_id)
        self.replay_times = 0
        self.stats = {}

    def get_stats(self):
        self.stats.update(dict(storage=self._size))
        return self.stats

    def __len__(self):
        return self._size

    def add_batch(self, batch):
        """batch = (obs [n,od], act [n,ad], rew [n], obs2 [n,od], done [n] uint8) device tensors (buffer.py:80-82)."""
        obs, act, rew, obs2, done = batch
        n = obs.shape[0]
        if done.dtype != torch.uint8:
            done = done.to(torch.uint8)
        L.call('mpg_replay_add', L.c_int(self._maxsize), L.c_int(self._next_idx), L.c_int(n), L.c_int(self.obs_dim),
               L.c_int(self.act_dim), L.ptr(obs), L.ptr(act), L.ptr(rew),
               L.ptr(obs2), L.ptr(done), L.ptr(self.obs), L.ptr(self.act), L.ptr(self.rew),
               L.ptr(self.obs2), L.ptr(self.done), L.stream())
        self._next_idx = (self._next_idx + n) % self._maxsize
        self._size = min(self._size + n, self._maxsize)

    def sample_idxes(self, batch_size):
        idx = torch.empty(batch_size, dtype=torch.int32, device=self.device)
        L.call('mpg_uniform_indices', L.c_int(self._size), L.c_int(batch_size), L.c_u64(self.seed),
               L.c_u64(self.replay_times), L.ptr(idx), L.stream())
        return idx

    def _encode_sample(self, idxes):
        n = idxes.shape[0]
        f = dict(dtype=torch.float32, device=self.device)
        o, o2 = torch.empty(n, self.obs_dim, **f), torch.empty(n, self.obs_dim, **f)
        a, r, d = torch.empty(n, self.act_dim, **f), torch.empty(n, **f), torch.empty(n, **f)
        L.call('mpg_replay_gather', L.c_int(n), L.ptr(idxes), L.c_int(self.obs_dim), L.c_int(self.act_dim),
               L.ptr(self.obs), L.ptr(self.act), L.ptr(self.rew), L.ptr(self.obs2), L.ptr(self.done), L.ptr(o), L.ptr(a),
               L.ptr(r), L.ptr(o2), L.ptr(d), L.stream())
        return o, a, r, o2, d

    def sample_with_idxes(self, idxes):
        return list(self._encode_sample(idxes)) + [idxes]

    def sample(self, batch_size):
        return self.sample_with_idxes(self.sample_idxes(batch_size))

    def replay(self):
        """[obs, act, rew, obs', done, idx] or None before `replay_starts` transitions (buffer.py:84-91)."""
        if self._size < self.replay_starts:
            return None
        self.replay_times += 1
        return self.sample(self.replay_batch_size)


class PrioritizedReplayBuffer(ReplayBuffer):
    """buffer.py:94-189 on device (sum / min segment trees in the reference's heap layout, float64).  The shipped
    constructor is dead code (asserts args.alpha > 0 with alpha None and reads args.size); this is the
    canonical PER it evidently intends: new transitions enter at max priority, p = (|td| + eps)^alpha with
    replay_alpha 0.6, IS weights with replay_beta 0.4 (train_script.py:239-240)."""

    def __init__(self, args, buffer_id, device='cuda', obs_dim=None, act_dim=None, eps=1e-6):
        super().__init__(args, buffer_id, device, obs_dim, act_dim)
        self._alpha, self._beta, self._eps = float(args.replay_alpha), float(args.replay_beta), float(eps)
        cap = 1
        while cap < self._maxsize:                                       # buffer.py:119-121
            cap *= 2
        self._cap = cap
        self._it_sum = torch.empty(2 * cap, dtype=torch.float64, device=self.device)
        self._it_min = torch.empty(2 * cap, dtype=torch.float64, device=self.device)
        self._stamp = torch.empty(cap, dtype=torch.int32, device=self.device)
        self._max_priority = torch.ones(1, dtype=torch.float64, device=self.device)          # buffer.py:125 (a python float there: float64)
        L.call('mpg_per_init', L.ptr(self._it_sum), L.ptr(self._it_min), L.ptr(self._stamp), L.c_int(cap), L.stream())

    def _set(self, idx, prio, eps):
        L.call('mpg_per_update', L.ptr(self._it_sum), L.ptr(self._it_min), L.ptr(self._stamp), L.c_int(self._cap),
               L.c_int(idx.shape[0]), L.ptr(idx), L.ptr(prio), L.c_double(self._alpha), L.c_double(eps),
               L.ptr(self._max_priority), L.stream())

    def add_batch(self, batch):
        n = batch[0].shape[0]
        idx = torch.empty(n, dtype=torch.int32, device=self.device)          # scratch: the slots (start + i) % capacity
        start = self._next_idx
        super().add_batch(batch)
        # weight = max priority (buffer.py:133-136): leaves (max_priority ** alpha) in float64
        L.call('mpg_per_add', L.ptr(self._it_sum), L.ptr(self._it_min), L.ptr(self._stamp), L.c_int(self._cap), L.c_int(self._maxsize),
               L.c_int(start), L.c_int(n), L.c_double(self._alpha), L.ptr(self._max_priority), L.ptr(idx), L.stream())

    def sample_idxes(self, batch_size, u=None, want_weights=True):
        idx = torch.empty(batch_size, dtype=torch.int32, device=self.device)
        w = torch.empty(batch_size, dtype=torch.float32, device=self.device) if want_weights else None
        L.call('mpg_per_sample', L.ptr(self._it_sum), L.ptr(self._it_min), L.c_int(self._cap), L.c_int(self._size),
               L.c_int(batch_size), L.ptr(u), L.c_u64(self.seed), L.c_u64(self.replay_times), L.c_double(self._beta),
               L.ptr(idx), L.ptr(w), L.stream())
        self._last_weights = w
        return idx

    def sample(self, batch_size):
        """[obs, act, rew, obs', done, weights, idx] (buffer.py:146-164): the proportional draw, the IS weights and the gather of the
        drawn rows in ONE launch (mpg_per_sample_gather; the same results as sample_idxes + _encode_sample)."""
        n = batch_size
        f = dict(dtype=torch.float32, device=self.device)
        idx = torch.empty(n, dtype=torch.int32, device=self.device)
        w = torch.empty(n, **f)
        o, o2 = torch.empty(n, self.obs_dim, **f), torch.empty(n, self.obs_dim, **f)
        a, r, d = torch.empty(n, self.act_dim, **f), torch.empty(n, **f), torch.empty(n, **f)
        L.call('mpg_per_sample_gather', L.ptr(self._it_sum), L.ptr(self._it_min), L.c_int(self._cap), L.c_int(self._size), L.c_int(n),
               L.ptr(None), L.c_u64(self.seed), L.c_u64(self.replay_times), L.c_double(self._beta), L.ptr(idx), L.ptr(w),
               L.c_int(self.obs_dim), L.c_int(self.act_dim), L.ptr(self.obs), L.ptr(self.act), L.ptr(self.rew), L.ptr(self.obs2),
               L.ptr(self.done), L.ptr(o), L.ptr(a), L.ptr(r), L.ptr(o2), L.ptr(d), L.stream())
        self._last_weights = w
        return [o, a, r, o2, d, w, idx]

    def update_priorities(self, idxes, priorities):
        """buffer.py:166-189; priorities may be signed td errors (|.| + eps is taken on the device)."""
        self._set(idxes.to(torch.int32).contiguous(), priorities.to(torch.float32).contiguous(), self._eps)
