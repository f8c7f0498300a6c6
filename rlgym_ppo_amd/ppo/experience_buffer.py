"""ExperienceBuffer -- drop-in for rlgym_ppo/ppo/experience_buffer.py:16-118, resident in HBM.

Same nine FIFO fields, same constructor, same `submit_experience` / `get_all_batches_shuffled` / `clear`, same
shuffle stream (numpy legacy RandomState(seed).permutation per epoch, remainder dropped).  Differences are layout
only: the buffer lives on the GPU (the reference keeps it on the CPU and re-uploads every minibatch,
ppo_learner.py:139-143), `states` rows are zero-padded to the kernels' leading dimension, PPOLearner reads
the buffer through index vectors (`epoch_indices`) instead of materialised gathers, and the FIFO is a RING: the reference's
`_cat` re-allocates all nine tensors on every submit (experience_buffer.py:18-37; 2 x 268 MB + 7 vectors at 524,288 rows);
here new rows overwrite the oldest ones in place and a rotating base index maps logical row i (0 = oldest, the reference's
order) to physical row (base + i) mod capacity -- the kernels apply that map to the permutation's entries
(rlppo_minibatch_args.ring_base / ring_cap), the reference-shaped accessors materialise the logical order on demand.
"""
import os

import numpy as np
import torch

from .. import _native as N
from ..engine import DeviceIndexRing, LegacyPermutation, ptr, require_gpu, stream_ptr

_FIELDS = ("states", "actions", "log_probs", "rewards", "next_states", "dones", "truncated", "values", "advantages")


class ExperienceBuffer(object):
    def __init__(self, max_size, seed, device):
        # the reference's Learner passes device="cpu" here (learner.py:124-126); the data still belongs in HBM
        self.device = device
        dev = torch.device(device)
        self._dev = dev if dev.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
        require_gpu(self._dev)
        self.seed = seed
        self.max_size = max_size
        self.rng = np.random.RandomState(seed)
        # epochs drawn ahead of the request by the shuffle pipeline (speculative, transparent: engine.LegacyPermutation)
        lookahead = max(0, int(os.environ.get("RLPPO_SHUFFLE_LOOKAHEAD", 3)))
        ring = getattr(self, "_ring", None)  # clear() re-runs __init__: the pinned / device index vectors are kept
        if ring is None or ring.slots != lookahead + 2:
            ring = DeviceIndexRing(self._dev, lookahead + 2)
        self._ring = ring
        self._perm = LegacyPermutation(self.rng, lookahead=lookahead, ring=ring)
        self._store = {k: None for k in _FIELDS}   # physical ring storage [capacity, ...] per field
        self._cap = 0     # rows allocated (grows geometrically up to max_size, then stays)
        self._base = 0    # physical row of logical row 0 (the oldest sample)
        self._count = 0   # valid rows
        self._d = None    # logical observation width

    # ------------------------------------------------------------------------------------------- FIFO
    def _grow(self, need, new):
        """(Re-)allocate the ring for at least `need` rows (at most max_size), keeping the current rows in logical order."""
        cap = min(self.max_size, max(need, 2 * self._cap))
        fresh = {}
        for k in _FIELDS:
            t = torch.empty((cap,) + tuple(new[k].shape[1:]), dtype=torch.float32, device=self._dev)
            if self._count:
                t[:self._count].copy_(self._logical(k))
            fresh[k] = t
        self._store, self._cap, self._base = fresh, cap, 0

    def _logical(self, k):
        """Rows of field k in the reference's order (oldest first): a view when the ring has not wrapped, else a copy."""
        t, b, n = self._store[k], self._base, self._count
        if b + n <= self._cap:
            return t[b:b + n]
        return torch.cat((t[b:], t[:b + n - self._cap]), 0)

    def _append(self, new):
        """Keep the newest max_size rows of old ++ new (experience_buffer.py:18-37: its four cases are this one rule)."""
        n_new = new["rewards"].shape[0]
        size = self.max_size
        if n_new >= size:                       # the new chunk alone fills the buffer: its last max_size rows
            if self._cap < size or any(self._store[k].shape[1:] != new[k].shape[1:] for k in _FIELDS):
                self._count = 0
                self._cap = 0
                self._grow(size, new)
            for k in _FIELDS:
                self._store[k].copy_(new[k][n_new - size:])
            self._base, self._count = 0, size
            return
        if self._count + n_new > self._cap and self._cap < size:
            self._grow(self._count + n_new, new)
        cap = self._cap
        w0 = (self._base + self._count) % cap   # first physical row to write
        first = min(n_new, cap - w0)
        for k in _FIELDS:
            self._store[k][w0:w0 + first].copy_(new[k][:first])
            if first < n_new:
                self._store[k][:n_new - first].copy_(new[k][first:])
        self._count += n_new
        if self._count > cap:                   # the oldest rows were overwritten
            self._base = (self._base + self._count - cap) % cap
            self._count = cap

    def _to_dev(self, x):
        if isinstance(x, torch.Tensor):
            return x.detach().to(self._dev, dtype=torch.float32)
        return torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).to(self._dev)

    def _pad_states(self, x):
        if isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 2 and self._d is not None \
                and x.shape[1] == N.lib().rlppo_padded_width(self._d):
            return x  # already padded device rows (Learner.add_new_experience hands them over as such)
        t = self._to_dev(x)
        if t.dim() == 1:
            t = t.view(-1, 1)
        t = t.reshape(t.shape[0], -1).contiguous()
        n, d = t.shape
        self._d = d
        ld = int(N.lib().rlppo_padded_width(d))
        out = torch.empty((n, ld), dtype=torch.float32, device=self._dev)
        N.check(N.lib().rlppo_pad_rows(stream_ptr(), ptr(t), 0, n, d, d, ptr(out), ld, 0, 0.0, 1.0))
        return out

    def submit_experience(self, states, actions, log_probs, rewards, next_states, dones, truncated, values, advantages):
        new = dict(states=self._pad_states(states), actions=self._to_dev(actions), log_probs=self._to_dev(log_probs),
                   rewards=self._to_dev(rewards), next_states=self._pad_states(next_states), dones=self._to_dev(dones),
                   truncated=self._to_dev(truncated), values=self._to_dev(values), advantages=self._to_dev(advantages))
        n = new["rewards"].shape[0]
        for k in _FIELDS:
            if new[k].shape[0] != n:
                raise ValueError(f"submit_experience: field '{k}' has {new[k].shape[0]} rows, 'rewards' has {n}")
        if self._cap and any(self._store[k].shape[1:] != new[k].shape[1:] for k in _FIELDS) and n < self.max_size:
            raise ValueError("submit_experience: row shapes differ from the rows already in the buffer")
        if n:
            self._append(new)

    def ring(self):
        """(physical storage dict, base, capacity) for the kernels: logical row i = physical row (i + base) mod capacity."""
        return self._store, self._base, self._cap

    # ------------------------------------------------------------------------------- reference-shaped views
    def _get(self, k):
        if self._count == 0:
            return torch.empty(0, dtype=torch.float32, device=self._dev)
        t = self._logical(k)
        if k in ("states", "next_states"):
            return t[:, :self._d]
        return t

    states = property(lambda s: s._get("states"))
    actions = property(lambda s: s._get("actions"))
    log_probs = property(lambda s: s._get("log_probs"))
    rewards = property(lambda s: s._get("rewards"))
    next_states = property(lambda s: s._get("next_states"))
    dones = property(lambda s: s._get("dones"))
    truncated = property(lambda s: s._get("truncated"))
    values = property(lambda s: s._get("values"))
    advantages = property(lambda s: s._get("advantages"))

    def __len__(self):
        return self._count

    # ------------------------------------------------------------------------------------------ shuffle
    def epoch_indices(self):
        """The permutation of one epoch (host int64 array): RandomState.permutation(total_samples), consumed once
        per epoch from the persistent generator (experience_buffer.py:97-98)."""
        return self._perm.permutation(len(self)).copy()  # the pipeline's own vector is recycled two requests later

    def epoch_indices_device(self, refill=True):
        """The same permutation as a device int64 vector, ordered on the current stream (uploaded by the shuffle pipeline
        on its own stream, normally long before it is asked for).  It stays valid until the next call.  refill=False: the caller
        tops the pipeline's look-ahead up itself (refill_shuffle) once its own launches are out."""
        self._ring.release_held()
        return self._ring.take(self._perm.take(len(self), refill=refill))

    def refill_shuffle(self):
        self._perm.refill()

    def _get_samples(self, indices):
        idx = torch.as_tensor(np.asarray(indices), device=self._dev)
        return (self.actions[idx], self.log_probs[idx], self.states[idx], self.values[idx], self.advantages[idx])

    def get_all_batches_shuffled(self, batch_size):
        """Reference-compatible generator of (actions, log_probs, states, values, advantages) gathers; batches that do
        not fill batch_size are dropped (quirk Q7).  PPOLearner uses epoch_indices() instead."""
        total = len(self)
        indices = self.epoch_indices()
        start = 0
        while start + batch_size <= total:
            yield self._get_samples(indices[start:start + batch_size])
            start += batch_size

    def clear(self):
        self._perm.close()
        self.__init__(self.max_size, self.seed, self.device)
