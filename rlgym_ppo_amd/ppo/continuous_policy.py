"""ContinuousPolicy -- drop-in for rlgym_ppo/ppo/continuous_policy.py:23-121 (+ MapContinuousToAction,
util/torch_functions.py:15-33) on librlppo's fused forward + Gaussian sampling kernel."""
import numpy as np
import torch
import torch.nn as nn

from .. import _native as N
from ..engine import ptr, stream_ptr
from ..util import torch_functions
from ._mlp import ArenaModule, build_body


class ContinuousPolicy(ArenaModule):
    def __init__(self, input_shape, output_shape, layer_sizes, device, var_min=0.1, var_max=1.0):
        super().__init__()
        self.affine_map = torch_functions.MapContinuousToAction(range_min=var_min, range_max=var_max)
        self.model = build_body(input_shape, layer_sizes, output_shape, nn.Tanh())
        self.n_out = int(output_shape)
        self._finish(device)

    @torch.no_grad()
    def get_output(self, obs):
        rows = self.arena.stage_obs(obs)
        y = self.arena.forward(rows, out_tanh=True)[:, :self.n_out]
        return self.affine_map(y)

    @torch.no_grad()
    def get_action(self, obs, summed_probs=True, deterministic=False, noise=None, standardize=None):
        a = self.arena
        if deterministic or not summed_probs:
            mean, std = self.get_output(obs)
            if deterministic:
                return mean, 0
            eps = torch.empty(mean.shape).normal_(0, 1).to(a.device) if noise is None else torch.as_tensor(noise).to(a.device)
            action = (eps * std + mean).clamp(min=-1, max=1)
            return action.cpu(), self.logpdf(action, mean, std).cpu()
        out = self._graph_act(obs, noise, standardize)  # small host batches: one hipGraph replay (ppo/_mlp.py)
        if out is not None:
            return out
        rows = a.stage_obs(obs, standardize)
        actions, logp = self.act_padded(rows, noise)
        return actions.cpu(), logp.cpu()

    # ---- hooks of the graph-replayed rollout step (ppo/_mlp.py::ActGraph)
    def _noise_shape(self, n):
        return (n, self.n_out // 2)

    def _draw_noise(self, n):
        return torch.empty(n, self.n_out // 2).normal_(0, 1)  # what Normal.sample() draws on the reference's CPU path

    def _action_buffer(self, cap):
        return torch.zeros((cap, self.n_out // 2), dtype=torch.float32)

    def _act_launch(self, rows, n, noise, actions, logp, ws, opts=None):
        a = self.arena
        N.check(N.lib().rlppo_gaussian_act(stream_ptr(), a.dims_c, a.n_layers, ptr(a.packed), ptr(rows), rows.shape[1], n,
                                           ptr(noise), float(self.affine_map.m), float(self.affine_map.b), ptr(actions),
                                           ptr(logp), ptr(ws), ws.numel(), opts))

    def act_padded(self, rows, noise=None):
        """Padded device rows -> (actions fp32 [n, k], summed log_probs fp32 [n]) on the device (see DiscreteFF.act_padded)."""
        a = self.arena
        n, k = rows.shape[0], self.n_out // 2
        if noise is None and self.noise_mode == "device":
            noise = torch.empty(n, k, device=a.device).normal_(0, 1)  # fast mode: torch's HIP generator, not the reference's CPU stream
        elif noise is None:
            noise = torch.empty(n, k).normal_(0, 1)  # what Normal.sample() draws on the reference's CPU path
        eps = torch.as_tensor(noise, dtype=torch.float32).to(a.device, non_blocking=True).contiguous()
        a.ensure_packed()
        actions = torch.empty((n, k), dtype=torch.float32, device=a.device)
        logp = torch.empty(n, dtype=torch.float32, device=a.device)
        ws = a.forward_ws(n)
        N.check(N.lib().rlppo_gaussian_act(stream_ptr(), a.dims_c, a.n_layers, ptr(a.packed), ptr(rows), rows.shape[1], n,
                                           ptr(eps), float(self.affine_map.m), float(self.affine_map.b), ptr(actions),
                                           ptr(logp), ptr(ws), ws.numel(), None))
        return actions, logp

    @staticmethod
    def logpdf(x, mean, std):
        msq, ssq, xsq = mean * mean, std * std, x * x
        return (-torch.divide(msq, 2 * ssq) + torch.divide(mean * x, ssq) - torch.divide(xsq, 2 * ssq)
                + torch.log(1 / torch.sqrt(2 * np.pi * ssq)))

    def get_backprop_data(self, obs, acts, summed_probs=True):
        """Compatibility accessor with an autograd graph (continuous_policy.py:100-121); unused by PPOLearner."""
        if not isinstance(obs, torch.Tensor):
            obs = torch.as_tensor(np.asarray(obs), dtype=torch.float32, device=self.arena.device)
        mean, std = self.affine_map(self.model(obs))
        prob = self.logpdf(acts, mean, std)
        log_probs = prob.sum(dim=1) if summed_probs else prob
        entropy = (0.5 + 0.5 * np.log(2 * np.pi) + torch.log(std)).mean()
        return log_probs, entropy
