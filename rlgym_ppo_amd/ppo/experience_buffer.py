"""ExperienceBuffer -- drop-in for rlgym_ppo/ppo/experience_buffer.py:16-118, resident in HBM.

Same nine FIFO fields, same constructor, same `submit_experience` / `get_all_batches_shuffled` / `clear`, same
shuffle stream (numpy legacy RandomState(seed).permutation per epoch, remainder dropped).  Differences are layout
only: the buffer lives on the GPU (the reference keeps it on the CPU and re-uploads every minibatch,
ppo_learner.py:139-143), `states` rows are zero-padded to the kernels' leading dimension, and PPOLearner reads
the buffer through index vectors (`epoch_indices`) instead of materialised gathers.
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
        self._store = {k: None for k in _FIELDS}
        self._d = None  # logical observation width

    # ------------------------------------------------------------------------------------------- FIFO
    def _fifo(self, old, new):
        """Keep the newest max_size rows of old ++ new (experience_buffer.py:18-37), on the device."""
        size = self.max_size
        if old is None or new.shape[0] >= size:
            return new[new.shape[0] - size:].clone() if new.shape[0] > size else new
        keep = min(old.shape[0], size - new.shape[0])
        return torch.cat((old[old.shape[0] - keep:], new), 0)

    def _to_dev(self, x):
        if isinstance(x, torch.Tensor):
            return x.detach().to(self._dev, dtype=torch.float32)
        return torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).to(self._dev)

    def _pad_states(self, x):
        if isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 2 and x.shape[1] % 32 == 0 and self._d is not None \
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
        for k in _FIELDS:
            self._store[k] = self._fifo(self._store[k], new[k])

    # ------------------------------------------------------------------------------- reference-shaped views
    def _get(self, k):
        t = self._store[k]
        if t is None:
            return torch.empty(0, dtype=torch.float32, device=self._dev)
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
        t = self._store["rewards"]
        return 0 if t is None else t.shape[0]

    # ------------------------------------------------------------------------------------------ shuffle
    def epoch_indices(self):
        """The permutation of one epoch (host int64 array): RandomState.permutation(total_samples), consumed once
        per epoch from the persistent generator (experience_buffer.py:97-98)."""
        return self._perm.permutation(len(self)).copy()  # the pipeline's own vector is recycled two requests later

    def epoch_indices_device(self):
        """The same permutation as a device int64 vector, ordered on the current stream (uploaded by the shuffle pipeline
        on its own stream, normally long before it is asked for).  It stays valid until the next call."""
        self._ring.release_held()
        return self._ring.take(self._perm.take(len(self)))

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
