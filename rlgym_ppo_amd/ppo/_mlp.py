"""Shared construction of the MLP bodies.

Layer order, nn.Linear default initialisation and CPU-generator consumption are those of the reference's
constructors (rlgym_ppo/ppo/discrete_policy.py:21-31, continuous_policy.py:29-41,
multi_discrete_policy.py:22-32, value_estimator.py:18-28): layers are built on the CPU first (so a seeded run
starts from the reference's weights bit for bit) and only then moved to the GPU arena.
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _native as N
from ..engine import NetArena, linears_of, ptr, require_gpu, selection_epoch, stream_ptr


def build_body(input_shape, layer_sizes, n_out, final_activation=None):
    assert len(layer_sizes) != 0, "AT LEAST ONE LAYER MUST BE SPECIFIED TO BUILD THE NEURAL NETWORK!"
    layers = [nn.Linear(int(input_shape), int(layer_sizes[0])), nn.ReLU()]
    prev = int(layer_sizes[0])
    for size in layer_sizes[1:]:
        layers.append(nn.Linear(prev, int(size)))
        layers.append(nn.ReLU())
        prev = int(size)
    layers.append(nn.Linear(prev, int(n_out)))
    if final_activation is not None:
        layers.append(final_activation)
    return nn.Sequential(*layers)


class ActGraph:
    """One hipGraph of the whole rollout step of a policy head for up to `cap` observations, with no copy in it:
    rlppo_pad_rows reads the observations and the head's act entry point (forward + sampling) reads its noise straight from
    pinned host memory (the discrete head: ONE node, rlppo_discrete_step on the raw observations [r3]) (hipHostMalloc memory is mapped into the GPU's address space), and the actions / log-probabilities are
    written straight into pinned host memory.  At the reference's rollout scale (8-80 observations per call,
    batched_agent_manager.py:202-204) a call is nothing but latency -- ~7 launches, three copies and two blocking read-backs,
    ~140-250 us; one replay + one synchronisation does the same work, with no copy node at all.  Same kernels, same
    arguments: results are those of the eager path bit for bit.  Rows past the caller's n hold stale data and are ignored."""

    def __init__(self, pol, cap):
        a = pol.arena
        dev, d = a.device, a.d_in
        self.cap = cap
        self.dev = dev
        self.obs_pin = torch.zeros(cap, d).pin_memory()
        self.q_pin = torch.ones(pol._noise_shape(cap)).pin_memory()
        self.rows = torch.zeros(cap, a.ld_in, device=dev)
        self.act_pin = pol._action_buffer(cap).pin_memory()
        self.logp_pin = torch.zeros(cap, dtype=torch.float32).pin_memory()
        L = N.lib()
        raw = getattr(pol, "_act_launch_raw", None)  # [r3] a head whose whole step is one launch on raw observations
        ws_bytes = max(int(L.rlppo_forward_workspace_bytes(a.dims_c, a.n_layers, cap)), raw(None, cap) if raw is not None else 0)
        self.ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)

        # [r5] completion by words the call's last kernel stores into pinned memory behind the results (rlppo_act_opts): run()
        # clears them, launches and polls them in C (rlppo_host_wait_words) instead of hipStreamSynchronize, which cost ~20 us of
        # a 66 us call; a poll that times out falls back to the synchronisation.  RLPPO_ACT_POLL=0: always synchronise.
        self.n_done = int(L.rlppo_act_done_words(cap))
        self.done_pin = torch.zeros(max(self.n_done, 1), dtype=torch.int32).pin_memory()
        self.poll = os.environ.get("RLPPO_ACT_POLL", "1") != "0"
        self.opts = N.ActOpts(N.PRECISION_DEFAULT, 1, self.done_pin.data_ptr()) if self.poll else None
        self.polled = self.poll_timeouts = 0
        self.seq = self.graph_value = 1   # (the warm-up launch and the capture below store 1)

        def body():
            if raw is not None:
                raw(self, cap, self.opts)
                return
            N.check(L.rlppo_pad_rows(stream_ptr(), ptr(self.obs_pin), 0, cap, d, d, ptr(self.rows), a.ld_in, 0, 0.0, 1.0))
            pol._act_launch(self.rows, cap, self.q_pin, self.act_pin, self.logp_pin, self.ws, self.opts)

        self.body = body
        # The body is replayed as a hipGraph.  (Rounds 3-4 issued a ONE-launch body -- the discrete head -- eagerly: the replay of a
        # one-node graph then cost 2-3 us more than the launch.  [r5] With completion polled instead of synchronised the replay is
        # the cheaper of the two by 2-3 us -- one hipGraphLaunch against a 22-argument ctypes call + hipLaunchKernel --
        # tools/small_batch_latency.py, profiles/r05_small_batch_latency.txt.)  RLPPO_ACT_EAGER=0/1 forces either form.
        self.eager = os.environ.get("RLPPO_ACT_EAGER", "0") == "1"
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            body()
        side.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: the shuffle pipeline's helper threads may be issuing copies / events on their own stream right now
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            body()
        self.obs_np, self.q_np = self.obs_pin.numpy(), self.q_pin.view(-1).numpy()
        self.act_np, self.logp_np = self.act_pin.numpy(), self.logp_pin.numpy()
        self.done_np = self.done_pin.numpy()
        self._done_ptr = ctypes.c_void_p(self.done_pin.data_ptr())
        self._wait = L.rlppo_host_wait_words

    def run(self, obs, q, n):
        # plain memcpy through numpy views made once: Tensor.copy_ fans out to an OpenMP team above 32k elements (10 ms on a
        # 256-thread host), and slicing tensors costs more than these copies at 8-80 rows
        self.obs_np[:n] = obs
        self.q_np[:q.numel()] = q.reshape(-1).numpy()
        if self.eager:
            # a fresh completion value per call: a workgroup of an EARLIER launch that finishes late (rows past that call's n, which
            # nobody waited for) stores the earlier value and cannot be mistaken for this call's; only the words of the rows the
            # caller asked for are awaited
            self.seq = self.seq % 0x7FFFFFFF + 1
            if self.poll:
                self.opts.done_value = self.seq
            self.body()
            value, count = self.seq, (n + 15) // 16
        else:
            # a replayed graph stores the value it was captured with: clear the words, wait for ALL of them (nothing of this
            # launch is then still running when the next call clears them again)
            if self.poll:
                self.done_np[:] = 0
            self.graph.replay()
            value, count = self.graph_value, self.n_done
        if self.poll and self._wait(self._done_ptr, count, value, 2000) == 0:
            self.polled += 1
        else:
            self.poll_timeouts += int(self.poll)
            torch.cuda.current_stream(self.dev).synchronize()
        return torch.from_numpy(self.act_np[:n].copy()), torch.from_numpy(self.logp_np[:n].copy())


def _bucket(n):
    """Capacity of the graph that serves n observations: multiples of 16 up to 128, then powers of two."""
    if n <= 128:
        return (n + 15) // 16 * 16
    c = 256
    while c < n:
        c *= 2
    return c


class ArenaModule(nn.Module):
    """nn.Module whose `self.model` Linear parameters live in a NetArena on the GPU."""

    # "host": sampling noise is drawn with torch's CPU generator -- the stream the reference's CPU path consumes, so a
    #         seeded run picks the reference's actions (costs ~1 us per drawn number on the host);
    # "device": noise drawn on the GPU (what the reference does when it runs on cuda); no host work per step.
    noise_mode = "host"

    def _finish(self, device):
        self.device = device
        dev = require_gpu(device)
        self.model = self.model.to(dev)
        self.arena = NetArena(linears_of(self.model), dev)
        self.act_graphs = os.environ.get("RLPPO_ACT_GRAPH", "1") != "0"  # get_action through a hipGraph per batch-size bucket
        self.act_graph_max = 1024  # beyond that the explicit copies of the eager path are the better transport for the noise
        self._graphs = {}
        self._graph_epoch = -1

    def _graph_act(self, obs, noise, standardize):
        """get_action for a small HOST batch as one graph replay, or None when that form does not apply (device inputs, fused
        standardisation, device-drawn noise, more than act_graph_max rows): the caller then takes the eager path."""
        if not self.act_graphs or standardize is not None or self.noise_mode != "host":
            return None
        if (isinstance(obs, torch.Tensor) and obs.is_cuda) or (isinstance(noise, torch.Tensor) and noise.is_cuda):
            return None
        a = self.arena
        if type(obs) is np.ndarray and obs.ndim == 2:  # (the per-environment-step call: no conversions)
            o = obs
        else:
            o = np.asarray(obs.detach().numpy() if isinstance(obs, torch.Tensor) else obs)
            if o.ndim == 1:
                o = o.reshape(1, -1)
        if o.ndim != 2 or o.shape[1] != a.d_in or not (0 < o.shape[0] <= self.act_graph_max):
            return None
        n = o.shape[0]
        q = self._draw_noise(n) if noise is None else torch.as_tensor(noise, dtype=torch.float32)
        if tuple(q.shape) != tuple(self._noise_shape(n)):
            return None
        # a captured graph replays the kernels that were selected at capture time: the library's selection epoch (bumped by
        # rlppo_set_inference_precision and the A/B switches) is part of the cache key, stale graphs are dropped
        epoch = selection_epoch()
        if epoch != self._graph_epoch:
            self._graphs.clear()
            self._graph_epoch = epoch
        g = self._graphs.get(_bucket(n))
        if g is None:
            g = self._graphs[_bucket(n)] = ActGraph(self, _bucket(n))
        a.ensure_packed()
        return g.run(o if o.dtype == np.float32 else o.astype(np.float32), q if q.is_contiguous() else q.contiguous(), n)

    def _apply(self, fn, *a, **k):  # .to()/.float()/... : re-bind afterwards so the kernels keep seeing the params
        out = super()._apply(fn, *a, **k)
        if hasattr(self, "arena"):
            self.arena.bind()
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.arena.bind()
        self.arena.native_epoch += 1
        return out
