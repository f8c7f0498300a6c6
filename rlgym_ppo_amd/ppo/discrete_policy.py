"""DiscreteFF -- drop-in for rlgym_ppo/ppo/discrete_policy.py:16-80.

get_action runs librlppo's fused forward + softmax + clamp + argmax(p/q) kernel.  The Exp(1) noise `q` is drawn
on the host with torch's CPU generator, because that is exactly what torch.multinomial(probs, 1, True) consumes
in the reference's CPU path; a seeded run therefore picks the reference's action indices.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from .. import _native as N
from ..engine import host_exponential, host_exponential_prefetch, ptr, stream_ptr
from ._mlp import ArenaModule, build_body


class DiscreteFF(ArenaModule):
    fused_step = True  # VectorAgentManager: collect through step() (one launch per environment step, time-major storage)

    def __init__(self, input_shape, n_actions, layer_sizes, device):
        super().__init__()
        self.model = build_body(input_shape, layer_sizes, n_actions, nn.Softmax(dim=-1))
        self.n_actions = int(n_actions)
        self._host_out = None  # pinned (actions, log-probs) the fused step writes into (grown on demand)
        self._finish(device)

    def _probs(self, rows, clamp, want_probs=True, want_argmax=False):
        """rlppo_discrete_probs on padded device rows: softmax (or clamp(softmax)) [n, n_actions] and/or the flat arg-max."""
        a = self.arena
        n = rows.shape[0]
        a.ensure_packed()
        probs = torch.empty(n, self.n_actions, dtype=torch.float32, device=a.device) if want_probs else None
        best = torch.empty(1, dtype=torch.int64, device=a.device) if want_argmax else None
        ws = a.forward_ws(n)
        N.check(N.lib().rlppo_discrete_probs(stream_ptr(), a.dims_c, a.n_layers, ptr(a.packed), ptr(rows), rows.shape[1], n, int(clamp),
                                             ptr(probs) if want_probs else None, self.n_actions, ptr(best) if want_argmax else None,
                                             ptr(ws), ws.numel(), None))
        return probs, best

    @torch.no_grad()
    def get_output(self, obs):
        """Softmax probabilities [n, n_actions] on the device (discrete_policy.py:34-42)."""
        return self._probs(self.arena.stage_obs(obs), clamp=False)[0]

    def get_action(self, obs, deterministic=False, noise=None, standardize=None):
        """-> (actions int64 CPU [n], log_probs fp32 CPU [n]) like discrete_policy.py:44-62.
        `noise`: optional [n, n_actions] Exp(1) draws (default: torch.empty(n, A).exponential_(1) from the CPU
        generator, the reference's stream).  `standardize`: optional (mean0, std0) scalars fused into staging."""
        if not deterministic:
            # small host batches: one hipGraph replay (ppo/_mlp.py).  (Nothing in there touches autograd: the no_grad scope -- 2 us
            # of a 45 us call -- is entered below, where torch operators run.)
            out = self._graph_act(obs, noise, standardize)
            if out is not None:
                return out
        return self._get_action_general(obs, deterministic, noise, standardize)

    @torch.no_grad()
    def _get_action_general(self, obs, deterministic, noise, standardize):
        a = self.arena
        if not deterministic:
            return self.step(obs, noise, standardize)       # [r3] the whole step in one launch (rlppo_discrete_step)
        rows = a.stage_obs(obs, standardize)
        n = rows.shape[0]
        if deterministic:  # quirk Q11: numpy's argmax over the flattened clamped [n, A] array -- one index for the whole batch
            return np.int64(self._probs(rows, clamp=True, want_probs=False, want_argmax=True)[1].item()), 0
        actions, logp = self.act_padded(rows, noise)
        return actions.cpu(), logp.cpu()

    def step(self, obs, noise=None, standardize=None, rows_out=None, actions_f32=None, logp_out=None, to_host=True):
        """One rollout step through rlppo_discrete_step [r3]: raw observations (numpy / tensor, fp32 or fp64, host or device) ->
        standardise + pad -> MLP -> softmax -> clamp -> argmax(p / q) -> log p, one launch (csrc/fused_act.hip).
        to_host=True: the kernel stores actions (int64) and log-probabilities straight into pinned host buffers; after ONE stream
        synchronisation they are returned as CPU tensors (no device-to-host copies).  to_host=False: device tensors.
        rows_out / actions_f32 / logp_out: optional device destinations of a device-resident rollout (VectorAgentManager): the
        padded policy-input rows [n, ld_in], the actions as floats [n] and the log-probabilities [n]."""
        a = self.arena
        if isinstance(obs, torch.Tensor):
            t = obs.detach()
            if t.dtype not in (torch.float32, torch.float64):
                t = t.float()
            t = t.to(a.device, non_blocking=True)
        else:
            arr = np.asarray(obs)
            if arr.dtype not in (np.float32, np.float64):
                arr = arr.astype(np.float32)
            t = torch.from_numpy(np.ascontiguousarray(arr)).to(a.device, non_blocking=True)
        if t.dim() == 1:
            t = t.view(1, -1)
        t = t.reshape(-1, t.shape[-1]).contiguous()
        n, d = t.shape
        if d != a.d_in:
            raise ValueError(f"observation width {d} != network input {a.d_in}")
        if noise is None and self.noise_mode == "device":
            q = torch.empty(n, self.n_actions, device=a.device).exponential_(1)  # fast mode: not the reference's CPU stream
        elif noise is None:
            q = host_exponential((n, self.n_actions), device=a.device)
        else:
            q = torch.as_tensor(noise, dtype=torch.float32).to(a.device, non_blocking=True).contiguous()
        if tuple(q.shape) != (n, self.n_actions):
            raise ValueError(f"noise shape {tuple(q.shape)} != {(n, self.n_actions)}")
        mode, mean0, std0, mean_v, std_v = 0, 0.0, 1.0, None, None
        if standardize is not None:
            if isinstance(standardize[0], torch.Tensor):
                mean_v, std_v = (x.to(a.device, dtype=torch.float32).contiguous() for x in standardize)
                if mean_v.numel() != d or std_v.numel() != d:
                    raise ValueError("per-feature statistics must have one entry per observation feature")
                mode = 2
            else:
                mode, mean0, std0 = 1, float(standardize[0]), float(standardize[1])
        a.ensure_packed()
        L = N.lib()
        if to_host:
            if self._host_out is None or self._host_out[0].numel() < n:
                cap = max(n, 64)
                self._host_out = (torch.empty(cap, dtype=torch.int64).pin_memory(), torch.empty(cap, dtype=torch.float32).pin_memory())
            actions = self._host_out[0][:n]
            logp = self._host_out[1][:n] if to_host is True else None   # to_host="actions": only the indices go to the host
        else:
            actions = torch.empty(n, dtype=torch.int64, device=a.device)
            logp = None
        if logp is None:
            logp = logp_out if logp_out is not None else torch.empty(n, dtype=torch.float32, device=a.device)
        elif logp_out is not None:
            raise ValueError("step: logp_out needs to_host=False or to_host='actions'")
        ws = a.ws.get(L.rlppo_discrete_step_workspace_bytes(a.dims_c, a.n_layers, n))
        N.check(L.rlppo_discrete_step(stream_ptr(), a.dims_c, a.n_layers, ptr(a.packed), ptr(t), int(t.dtype == torch.float64), d, n,
                                      mode, mean0, std0, ptr(mean_v), ptr(std_v), ptr(q), ptr(actions), ptr(actions_f32), ptr(logp),
                                      ptr(rows_out), rows_out.stride(0) if rows_out is not None else 0, ptr(ws), ws.numel(), None))
        if to_host:
            torch.cuda.current_stream(a.device).synchronize()
            return actions.clone(), (logp.clone() if to_host is True else logp)
        return actions, logp

    def prefetch_noise(self, n, count):
        """The next `count` calls of step() / get_action() will each act on n observations with the reference's CPU noise stream:
        have those draws produced ahead on the helper threads (engine.HostExponential.prefetch; transparent speculation)."""
        return host_exponential_prefetch((int(n), self.n_actions), int(count), device=self.arena.device)

    # ---- hooks of the graph-replayed rollout step (ppo/_mlp.py::ActGraph)
    def _noise_shape(self, n):
        return (n, self.n_actions)

    def _draw_noise(self, n):
        return host_exponential((n, self.n_actions))  # == torch.empty(n, A).exponential_(1), drawn ahead (engine.py)

    def _action_buffer(self, cap):
        return torch.zeros(cap, dtype=torch.int64)

    def _act_launch(self, rows, n, noise, actions, logp, ws, opts=None):
        a = self.arena
        N.check(N.lib().rlppo_discrete_act(stream_ptr(), a.dims_c, a.n_layers, ptr(a.packed), ptr(rows), rows.shape[1], n,
                                           ptr(noise), ptr(actions), ptr(logp), None, ptr(ws), ws.numel(), opts))

    def _act_launch_raw(self, g, cap, opts=None):
        """ActGraph's body as one node: rlppo_discrete_step on the graph's pinned observations / noise / outputs (g=None: the
        workspace bytes that takes)."""
        a = self.arena
        L = N.lib()
        if g is None:
            return int(L.rlppo_discrete_step_workspace_bytes(a.dims_c, a.n_layers, cap))
        args = getattr(g, "_raw_args", None)
        if args is None or args[0] is not opts:  # every pointer of the call is fixed for the graph's lifetime: built once
            args = g._raw_args = (opts, (a.dims_c, a.n_layers, ptr(a.packed), g.obs_arg, 0, a.d_in, cap, 0, 0.0, 1.0, None, None, g.q_arg,
                                         ptr(g.act_pin), None, ptr(g.logp_pin), None, 0, ptr(g.ws), g.ws.numel(),
                                         ctypes.byref(opts) if opts is not None else None))
        N.check(L.rlppo_discrete_step(stream_ptr(), *args[1]))

    def act_padded(self, rows, noise=None):
        """Padded device rows [n, ld_in] -> (actions int64 [n], log_probs fp32 [n]) ON THE DEVICE: the part of get_action
        after staging, for callers that keep the rollout on the GPU (VectorAgentManager)."""
        a = self.arena
        n = rows.shape[0]
        if noise is None and self.noise_mode == "device":
            noise = torch.empty(n, self.n_actions, device=a.device).exponential_(1)  # fast mode: torch's HIP generator, not the reference's CPU stream
        elif noise is None:  # the reference's CPU stream, uploaded asynchronously (the pinned ring slot is event-protected)
            noise = host_exponential((n, self.n_actions), device=a.device)
        q = torch.as_tensor(noise, dtype=torch.float32).to(a.device, non_blocking=True).contiguous()
        a.ensure_packed()
        actions = torch.empty(n, dtype=torch.int64, device=a.device)
        logp = torch.empty(n, dtype=torch.float32, device=a.device)
        ws = a.forward_ws(n)
        N.check(N.lib().rlppo_discrete_act(stream_ptr(), a.dims_c, a.n_layers, ptr(a.packed), ptr(rows), rows.shape[1], n,
                                           ptr(q), ptr(actions), ptr(logp), None, ptr(ws), ws.numel(), None))
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
