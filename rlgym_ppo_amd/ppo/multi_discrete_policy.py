"""MultiDiscreteFF -- drop-in for rlgym_ppo/ppo/multi_discrete_policy.py:16-89 (+ MultiDiscreteRolv,
util/torch_functions.py:81-122) on librlppo's fused forward + 8-way categorical sampling kernel."""
import numpy as np
import torch

from .. import _native as N
from ..engine import host_exponential, ptr, stream_ptr
from ..util import torch_functions
from ._mlp import ArenaModule, build_body


class MultiDiscreteFF(ArenaModule):
    def __init__(self, input_shape, layer_sizes, device):
        super().__init__()
        bins = [3, 3, 3, 3, 3, 2, 2, 2]
        self.model = build_body(input_shape, layer_sizes, sum(bins))
        self.splits = bins
        self.multi_discrete = torch_functions.MultiDiscreteRolv(bins)
        self._finish(device)

    @torch.no_grad()
    def get_output(self, obs):
        rows = self.arena.stage_obs(obs)
        return self.arena.forward(rows)[:, :21]

    @torch.no_grad()
    def get_action(self, obs, deterministic=False, noise=None, standardize=None):
        a = self.arena
        if deterministic:
            logits = self.get_output(obs)
            action, start = [], 0
            for split in self.splits:
                action.append(logits[..., start:start + split].argmax(dim=-1))
                start += split
            return torch.stack(action).cpu().numpy(), 0
        out = self._graph_act(obs, noise, standardize)  # small host batches: one hipGraph replay (ppo/_mlp.py)
        if out is not None:
            return out
        rows = a.stage_obs(obs, standardize)
        actions, logp = self.act_padded(rows, noise)
        return actions.cpu(), logp.cpu()

    # ---- hooks of the graph-replayed rollout step (ppo/_mlp.py::ActGraph)
    def _noise_shape(self, n):
        return (n * 8, 3)

    def _draw_noise(self, n):
        return host_exponential((n * 8, 3))  # Categorical.sample -> multinomial on [n*8, 3]: torch.empty(n*8, 3).exponential_(1)

    def _action_buffer(self, cap):
        return torch.zeros((cap, 8), dtype=torch.int64)

    def _act_launch(self, rows, n, noise, actions, logp, ws, opts=None):
        a = self.arena
        N.check(N.lib().rlppo_multidiscrete_act(stream_ptr(), a.dims_c, a.n_layers, ptr(a.packed), ptr(rows), rows.shape[1],
                                                n, ptr(noise), ptr(actions), ptr(logp), ptr(ws), ws.numel(), opts))

    def act_padded(self, rows, noise=None):
        """Padded device rows -> (actions int64 [n, 8], log_probs fp32 [n]) on the device (see DiscreteFF.act_padded)."""
        a = self.arena
        n = rows.shape[0]
        if noise is None and self.noise_mode == "device":
            noise = torch.empty(n * 8, 3, device=a.device).exponential_(1)  # fast mode: torch's HIP generator, not the reference's CPU stream
        elif noise is None:
            noise = host_exponential((n * 8, 3), device=a.device)  # Categorical.sample -> multinomial on [n*8, 3]
        q = torch.as_tensor(noise, dtype=torch.float32).to(a.device, non_blocking=True).contiguous()
        a.ensure_packed()
        actions = torch.empty((n, 8), dtype=torch.int64, device=a.device)
        logp = torch.empty(n, dtype=torch.float32, device=a.device)
        ws = a.forward_ws(n)
        N.check(N.lib().rlppo_multidiscrete_act(stream_ptr(), a.dims_c, a.n_layers, ptr(a.packed), ptr(rows), rows.shape[1],
                                                n, ptr(q), ptr(actions), ptr(logp), ptr(ws), ws.numel(), None))
        return actions, logp

    def get_backprop_data(self, obs, acts):
        """Compatibility accessor with an autograd graph (multi_discrete_policy.py:76-89); unused by PPOLearner."""
        if not isinstance(obs, torch.Tensor):
            obs = torch.as_tensor(np.asarray(obs), dtype=torch.float32, device=self.arena.device)
        dist = self.multi_discrete
        dist.make_distribution(self.model(obs))
        return dist.log_prob(acts), dist.entropy().mean()
