"""DiscreteFF -- drop-in for rlgym_ppo/ppo/discrete_policy.py:16-80.

get_action runs librlppo's fused forward + softmax + clamp + argmax(p/q) kernel.  The Exp(1) noise `q` is drawn
on the host with torch's CPU generator, because that is exactly what torch.multinomial(probs, 1, True) consumes
in the reference's CPU path; a seeded run therefore picks the reference's action indices.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import _native as N
from ..engine import host_exponential, ptr, stream_ptr
from ._mlp import ArenaModule, build_body


class DiscreteFF(ArenaModule):
    def __init__(self, input_shape, n_actions, layer_sizes, device):
        super().__init__()
        self.model = build_body(input_shape, layer_sizes, n_actions, nn.Softmax(dim=-1))
        self.n_actions = int(n_actions)
        self._finish(device)

    @torch.no_grad()
    def get_output(self, obs):
        """Softmax probabilities [n, n_actions] on the device (discrete_policy.py:34-42)."""
        rows = self.arena.stage_obs(obs)
        logits = self.arena.forward(rows)[:, :self.n_actions]
        # softmax of 90 numbers per row is not worth a kernel of its own for this compatibility accessor;
        # get_action() below uses the fused kernel.
        return torch.softmax(logits, dim=-1)

    @torch.no_grad()
    def get_action(self, obs, deterministic=False, noise=None, standardize=None):
        """-> (actions int64 CPU [n], log_probs fp32 CPU [n]) like discrete_policy.py:44-62.
        `noise`: optional [n, n_actions] Exp(1) draws (default: torch.empty(n, A).exponential_(1) from the CPU
        generator, the reference's stream).  `standardize`: optional (mean0, std0) scalars fused into staging."""
        a = self.arena
        if not deterministic:
            out = self._graph_act(obs, noise, standardize)  # small host batches: one hipGraph replay (ppo/_mlp.py)
            if out is not None:
                return out
        rows = a.stage_obs(obs, standardize)
        n = rows.shape[0]
        if deterministic:
            probs = torch.clamp(torch.softmax(a.forward(rows)[:, :self.n_actions], dim=-1), min=1e-11, max=1)
            return probs.cpu().numpy().argmax(), 0  # quirk Q11: flat argmax over the whole batch
        actions, logp = self.act_padded(rows, noise)
        return actions.cpu(), logp.cpu()

    # ---- hooks of the graph-replayed rollout step (ppo/_mlp.py::ActGraph)
    def _noise_shape(self, n):
        return (n, self.n_actions)

    def _draw_noise(self, n):
        return host_exponential((n, self.n_actions))  # == torch.empty(n, A).exponential_(1), drawn ahead (engine.py)

    def _action_buffer(self, cap):
        return torch.zeros(cap, dtype=torch.int64)

    def _act_launch(self, rows, n, noise, actions, logp, ws):
        a = self.arena
        N.check(N.lib().rlppo_discrete_act(stream_ptr(), a.dims_c, a.n_layers, ptr(a.packed), ptr(rows), rows.shape[1], n,
                                           ptr(noise), ptr(actions), ptr(logp), None, ptr(ws), ws.numel()))

    def act_padded(self, rows, noise=None):
        """Padded device rows [n, ld_in] -> (actions int64 [n], log_probs fp32 [n]) ON THE DEVICE: the part of get_action
        after staging, for callers that keep the rollout on the GPU (VectorAgentManager)."""
        a = self.arena
        n = rows.shape[0]
        if noise is None and self.noise_mode == "device":
            noise = torch.empty(n, self.n_actions, device=a.device).exponential_(1)  # fast mode: torch's HIP generator, not the reference's CPU stream
        elif noise is None:
            noise = host_exponential((n, self.n_actions))
        q = torch.as_tensor(noise, dtype=torch.float32).to(a.device, non_blocking=True).contiguous()
        a.ensure_packed()
        actions = torch.empty(n, dtype=torch.int64, device=a.device)
        logp = torch.empty(n, dtype=torch.float32, device=a.device)
        ws = a.forward_ws(n)
        N.check(N.lib().rlppo_discrete_act(stream_ptr(), a.dims_c, a.n_layers, ptr(a.packed), ptr(rows), rows.shape[1], n,
                                           ptr(q), ptr(actions), ptr(logp), None, ptr(ws), ws.numel()))
        return actions, logp

    def get_backprop_data(self, obs, acts):
        """Compatibility accessor with an autograd graph (discrete_policy.py:64-80), evaluated by stock PyTorch
        on the same parameters.  PPOLearner.learn does NOT use it: the update runs in rlppo_ppo_minibatch."""
        acts = acts.long()
        if not isinstance(obs, torch.Tensor):
            obs = torch.as_tensor(np.asarray(obs), dtype=torch.float32, device=self.arena.device)
        probs = torch.clamp(self.model(obs).view(-1, self.n_actions), min=1e-11, max=1)
        log_probs = torch.log(probs)
        return log_probs.gather(-1, acts), -(log_probs * probs).sum(dim=-1).mean()
