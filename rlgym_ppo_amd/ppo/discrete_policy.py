"""DiscreteFF -- drop-in for rlgym_ppo/ppo/discrete_policy.py:16-80.

get_action runs librlppo's fused forward + softmax + clamp + argmax(p/q) kernel.  The Exp(1) noise `q` is drawn
on the host with torch's CPU generator, because that is exactly what torch.multinomial(probs, 1, True) consumes
in the reference's CPU path; a seeded run therefore picks the reference's action indices.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _native as N
from ..engine import ptr, stream_ptr
from ._mlp import ArenaModule, build_body


class _ActGraph:
    """One hipGraph of the whole rollout step for up to `cap` observations, with no copy in it: rlppo_pad_rows reads the
    observations and rlppo_discrete_act (forward + softmax + clamp + argmax(p/q)) reads the Exp(1) noise straight from pinned
    host memory (hipHostMalloc memory is mapped into the GPU's address space), and the actions / log-probabilities are written
    straight into pinned host memory.  At the reference's rollout scale (8-80 observations per call,
    batched_agent_manager.py:202-204) a call is nothing but latency -- ~7 launches, three copies and two blocking read-backs,
    ~140-250 us; one replay + one synchronisation does the same work, with no copy node at all.  Same kernels, same arguments: results are those of the eager path bit for bit.  Rows past
    the caller's n hold stale data and are ignored."""

    def __init__(self, pol, cap):
        a = pol.arena
        dev, d, A = a.device, a.d_in, pol.n_actions
        self.cap = cap
        self.obs_pin = torch.zeros(cap, d).pin_memory()
        self.q_pin = torch.ones(cap, A).pin_memory()
        self.rows = torch.zeros(cap, a.ld_in, device=dev)
        self.act_pin = torch.zeros(cap, dtype=torch.int64).pin_memory()
        self.logp_pin = torch.zeros(cap, dtype=torch.float32).pin_memory()
        self.ws = torch.empty(int(N.lib().rlppo_forward_workspace_bytes(a.dims_c, a.n_layers, cap)), dtype=torch.uint8, device=dev)
        L = N.lib()

        def body():
            N.check(L.rlppo_pad_rows(stream_ptr(), ptr(self.obs_pin), 0, cap, d, d, ptr(self.rows), a.ld_in, 0, 0.0, 1.0))
            N.check(L.rlppo_discrete_act(stream_ptr(), a.dims_c, a.n_layers, ptr(a.packed), ptr(self.rows), a.ld_in, cap,
                                         ptr(self.q_pin), ptr(self.act_pin), ptr(self.logp_pin), None, ptr(self.ws), self.ws.numel()))

        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            body()
        side.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            body()

    def run(self, obs, q, n):
        self.obs_pin[:n].numpy()[...] = obs
        self.q_pin[:n].numpy()[...] = q.numpy()  # plain memcpy: Tensor.copy_ fans out to an OpenMP team above 32k elements (10 ms on a 256-thread host)
        self.graph.replay()
        torch.cuda.current_stream().synchronize()
        return self.act_pin[:n].clone(), self.logp_pin[:n].clone()


def _bucket(n):
    """Capacity of the graph that serves n observations: multiples of 16 up to 128, then powers of two."""
    if n <= 128:
        return (n + 15) // 16 * 16
    c = 256
    while c < n:
        c *= 2
    return c


class DiscreteFF(ArenaModule):
    def __init__(self, input_shape, n_actions, layer_sizes, device):
        super().__init__()
        self.model = build_body(input_shape, layer_sizes, n_actions, nn.Softmax(dim=-1))
        self.n_actions = int(n_actions)
        self._finish(device)
        self.act_graphs = os.environ.get("RLPPO_ACT_GRAPH", "1") != "0"  # get_action through a hipGraph per batch-size bucket
        self.act_graph_max = 1024  # beyond that the explicit copies of the eager path are the better transport for the noise
        self._graphs = {}

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
        # host observations, host (or default) noise, no fused standardisation: the whole step as one graph replay
        if (self.act_graphs and not deterministic and standardize is None and self.noise_mode == "host"
                and not (isinstance(obs, torch.Tensor) and obs.is_cuda) and not (isinstance(noise, torch.Tensor) and noise.is_cuda)):
            o = np.asarray(obs.detach().numpy() if isinstance(obs, torch.Tensor) else obs)
            if o.ndim == 1:
                o = o.reshape(1, -1)
            if o.ndim == 2 and o.shape[1] == a.d_in and 0 < o.shape[0] <= self.act_graph_max:
                n = o.shape[0]
                q = torch.empty(n, self.n_actions).exponential_(1) if noise is None else torch.as_tensor(noise, dtype=torch.float32)
                if q.shape == (n, self.n_actions):
                    g = self._graphs.get(_bucket(n))
                    if g is None:
                        g = self._graphs[_bucket(n)] = _ActGraph(self, _bucket(n))
                    a.ensure_packed()
                    return g.run(o.astype(np.float32, copy=False), q, n)
        rows = a.stage_obs(obs, standardize)
        n = rows.shape[0]
        if deterministic:
            probs = torch.clamp(torch.softmax(a.forward(rows)[:, :self.n_actions], dim=-1), min=1e-11, max=1)
            return probs.cpu().numpy().argmax(), 0  # quirk Q11: flat argmax over the whole batch
        actions, logp = self.act_padded(rows, noise)
        return actions.cpu(), logp.cpu()

    def act_padded(self, rows, noise=None):
        """Padded device rows [n, ld_in] -> (actions int64 [n], log_probs fp32 [n]) ON THE DEVICE: the part of get_action
        after staging, for callers that keep the rollout on the GPU (VectorAgentManager)."""
        a = self.arena
        n = rows.shape[0]
        if noise is None and self.noise_mode == "device":
            noise = torch.empty(n, self.n_actions, device=a.device).exponential_(1)  # fast mode: torch's HIP generator, not the reference's CPU stream
        elif noise is None:
            noise = torch.empty(n, self.n_actions).exponential_(1)
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
