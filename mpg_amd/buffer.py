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
        done = done.to(torch.uint8).contiguous()
        L.call('mpg_replay_add', L.c_int(self._maxsize), L.c_int(self._next_idx), L.c_int(n), L.c_int(self.obs_dim),
               L.c_int(self.act_dim), L.ptr(obs.contiguous()), L.ptr(act.contiguous()), L.ptr(rew.contiguous()),
               L.ptr(obs2.contiguous()), L.ptr(done), L.ptr(self.obs), L.ptr(self.act), L.ptr(self.rew),
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
