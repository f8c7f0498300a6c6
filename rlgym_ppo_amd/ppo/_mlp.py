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


class _WindowRows:
    """The padded observation rows of a host window as the launchers see a tensor: a shape and an address."""

    def __init__(self, address, n, ld):
        self.shape, self._address = (n, ld), address

    def data_ptr(self):
        return self._address


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
        # a 66 us call; a poll that times out falls back to the synchronisation.
        # [r6] ONE configuration + ONE fallback: host window, late noise and polled completion words are what every call uses;
        # RLPPO_ACT_PUSH=0 (or a device that refuses a host window) keeps inputs and noise in pinned host memory.  Round 5's other
        # switches (RLPPO_ACT_POLL / _EAGER / _LATE_NOISE / _HDP_FLUSH / _GRAPH) were configurations nobody ran and are gone.
        self.n_done = int(L.rlppo_act_done_words(cap))
        self.done_pin = torch.zeros(max(self.n_done, 1), dtype=torch.int32).pin_memory()
        self.opts = N.ActOpts(N.PRECISION_DEFAULT, 1, self.done_pin.data_ptr())
        self.calls = self.polled = self.poll_timeouts = self.late_retries = self.stale_relaunches = 0
        self.graph_value = 1   # (the warm-up launch and the capture below store 1)
        # [r5] host window (rlppo_host_window_alloc): the one-launch step reads its observations, noise and control words from DEVICE
        # memory that run() writes directly through the PCIe aperture (posted writes: 1.5-2.5 us for 8-80 observations) instead of
        # reading pinned host memory itself (GPU-initiated PCIe reads: 11-20 us of the kernel).  RLPPO_ACT_PUSH=0: pinned memory.
        self.window = None
        self.obs_arg, self.q_arg = self.obs_pin.data_ptr(), self.q_pin.data_ptr()   # what _act_launch_raw hands to the kernel
        self.push = os.environ.get("RLPPO_ACT_PUSH", "1") != "0"   # (the layer chains of the other heads read the window too)
        if self.push:
            r256 = lambda x: (x + 255) // 256 * 256
            # (the layer chains read their first layer's input straight from the window: rows of ld_in floats, zero beyond d -- the
            # window is zeroed once and the host only ever writes the first d floats of a row: no pad launch)
            self.padded = raw is None
            obs_bytes, q_bytes = r256(cap * (a.ld_in if self.padded else d) * 4), r256(int(self.q_pin.numel()) * 4)
            win = ctypes.c_void_p()
            with torch.cuda.device(dev):   # (the window belongs to the device that is current when it is made)
                rc_win = L.rlppo_host_window_alloc(256 + obs_bytes + q_bytes, ctypes.byref(win))
            if rc_win == 0:
                self.window = win.value
                self.ctl_arg, self.obs_arg, self.q_arg = win.value, win.value + 256, win.value + 256 + obs_bytes
                self._push, self._stage, self._stage_rows = L.rlppo_host_push, L.rlppo_host_stage_call, L.rlppo_host_stage_rows
                # the device's host data path is flushed between the staged bytes and the launch (by the book: 0.9 us of the call;
                # round 5's "flush behind the launch" ordered data by launch latency -- a timing argument, not a guarantee -- and is gone)
                self._flush = L.rlppo_host_window_flush
                if self.padded:
                    self.rows = _WindowRows(self.obs_arg, cap, a.ld_in)
            else:
                self.push = False   # (a device that does not expose its memory to the host)
        self.padded = self.push and raw is None
        # [r5] late noise (rlppo_act_opts.noise_ctl): run() launches FIRST and draws the Exp(1) numbers afterwards -- the bit-exact
        # draw (5-11 us at 8-80 rows) then costs the call nothing, it hides behind the launch latency and the layers; the kernel
        # looks for control word 2 when its head layer starts.  Up to 256 rows: beyond that the draw outlasts the kernel.
        self.late = bool(self.push and raw is not None and cap <= 256)
        if self.late:
            self.opts.noise_ctl = self.ctl_arg
            self.late = L.rlppo_discrete_step_one_launch(a.dims_c, a.n_layers, cap, ctypes.byref(self.opts)) == 1
            if not self.late:
                self.opts.noise_ctl = None
        if self.late:
            self.noise_seq = 0   # (control words {sequence, live rows}: 0 rows while capturing)
        self.q_per_row = int(self.q_pin.numel()) // cap

        # (no closure over self: an ActGraph dropped from the policy's cache dies there and then, by reference count -- a cyclic
        # collection that destroyed its hipGraph later, at a random allocation, could wait for the GPU while a kernel waits for us)
        self._raw, self._pol = raw, pol
        body = self.body
        # The body is replayed as a hipGraph.  (Rounds 3-4 issued a ONE-launch body -- the discrete head -- eagerly: the replay of a
        # one-node graph then cost 2-3 us more than the launch.  [r5] With completion polled instead of synchronised the replay is
        # the cheaper of the two by 2-3 us -- one hipGraphLaunch against a 22-argument ctypes call + hipLaunchKernel --
        # tools/small_batch_latency.py, profiles/r05_small_batch_latency.txt.)
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

    def __del__(self):
        win, self.window = getattr(self, "window", None), None
        if win:
            try:
                torch.cuda.synchronize(self.dev)   # (its graph may still be running)
                with torch.cuda.device(self.dev):
                    N.lib().rlppo_host_window_free(ctypes.c_void_p(win))
            except Exception:  # noqa: BLE001 -- interpreter shutdown
                pass

    def read_control_words(self):
        """The 32 control / statistics words of the late noise (a slow uncached read of device memory: tools only)."""
        out = np.zeros(32, dtype=np.uint32)
        torch.cuda.synchronize(self.dev)
        ctypes.memmove(out.ctypes.data, self.ctl_arg, 128)
        return out

    def body(self):
        pol, a, cap = self._pol, self._pol.arena, self.cap
        if self._raw is not None:
            self._raw(self, cap, self.opts)
            return
        if not self.padded:
            N.check(N.lib().rlppo_pad_rows(stream_ptr(), ptr(self.obs_arg), 0, cap, a.d_in, a.d_in, ptr(self.rows), a.ld_in, 0, 0.0, 1.0))
        pol._act_launch(self.rows, cap, self.q_arg, self.act_pin, self.logp_pin, self.ws, self.opts)

    def run(self, obs, q, n, draw=None, verify=None):
        """obs [n, d] float32 numpy; q: the call's noise (CPU tensor) or None with draw(): called for it -- AFTER the launch when
        the graph takes late noise.  verify (optional): () -> bool, asked after the launch whether what the launch read (the packed
        weights) was current; on False redo() -- its second return value -- is run and the call made again."""
        # plain memcpy through numpy views made once: Tensor.copy_ fans out to an OpenMP team above 32k elements (10 ms on a
        # 256-thread host), and slicing tensors costs more than these copies at 8-80 rows
        m = n * self.q_per_row
        if self.push:
            if not obs.flags.c_contiguous:
                obs = np.ascontiguousarray(obs)
            if self.late:
                self.noise_seq = self.noise_seq % 0x7FFFFFFF + 1
                rc = self._stage(self.ctl_arg, self.noise_seq, n, self.obs_arg, obs.ctypes.data, obs.nbytes)
            else:
                if q is None:
                    q = draw()
                q = q if q.is_contiguous() else q.contiguous()
                if q.numel() != m or q.dtype != torch.float32:
                    raise ValueError("noise: expected %d float32 numbers" % m)
                rc = self._push(self.q_arg, q.data_ptr(), 4 * m, None, 0)
                if self.padded:
                    d4 = 4 * obs.shape[1]
                    rc = rc or self._stage_rows(None, 0, 0, self.obs_arg, 4 * self.rows.shape[1], obs.ctypes.data, d4, d4, n)
                else:
                    rc = rc or self._stage(None, 0, 0, self.obs_arg, obs.ctypes.data, obs.nbytes)
            if rc:
                N.check(rc)
        else:
            self.obs_np[:n] = obs
            if q is None:
                q = draw()
            self.q_np[:m] = q.reshape(-1).numpy()
        if self.push:
            self._flush(self.window)
        self.calls += 1
        value, count = self._launch(n)
        if self.late:
            # the kernel is on its way: now the noise.  Whatever happens here control word 2 gets this call's sequence (a kernel left
            # waiting sits on the GPU until it gives up, 20 ms).  Nothing in here may wait for the GPU: the kernel waits for us.
            try:
                if q is None:
                    q = draw()
                q = q if q.is_contiguous() else q.contiguous()
                if q.numel() != m or q.dtype != torch.float32:
                    raise ValueError("noise: expected %d float32 numbers" % m)
                N.check(self._push(self.q_arg, q.data_ptr(), 4 * m, self.ctl_arg + 8, self.noise_seq))
            except BaseException:
                self._push(None, None, 0, self.ctl_arg + 8, self.noise_seq)
                self._wait(self._done_ptr, count, value, 100000)
                raise
        rc = None
        if verify is not None:
            ok, redo = verify()
            if not ok:
                # the launch read a stale copy of the weights (somebody wrote the parameters since the last call): let it finish,
                # bring the copy up to date, make the call again -- the noise is in place
                rc = self._finish(value, count)
                redo()
                self.stale_relaunches += 1
                value, count = self._launch(n)
                rc = None
        if (self._finish(value, count) if rc is None else rc) == 2:
            # the noise came later than the kernel's patience (the host was held up between the launch and the publish: a
            # collector pause that destroys device objects, a descheduled thread): the pairs are there now, launch again
            self.late_retries += 1
            value, count = self._launch(n)
            if self._finish(value, count) != 0:
                raise RuntimeError("rollout step: the kernel gave up on noise that is there (rlppo_act_opts.noise_ctl): sequence %d, "
                                   "completion words %s" % (self.noise_seq, self.done_np.astype(np.uint32).tolist()))
        return torch.from_numpy(self.act_np[:n].copy()), torch.from_numpy(self.logp_np[:n].copy())

    def _launch(self, n):
        """-> (value, count): the completion words to wait for.  A replayed graph stores the value it was captured with: clear the
        words, wait for ALL of them (nothing of this launch is then still running when the next call clears them again)."""
        self.done_np[:] = 0
        self.graph.replay()
        return self.graph_value, self.n_done

    def _finish(self, value, count):
        """Waits for the launch: 0 = results are there, 2 = the kernel gave up on its late noise."""
        rc = self._wait(self._done_ptr, count, value, 2000)
        if rc == 0:
            self.polled += 1
        elif rc == 1:
            self.poll_timeouts += 1
            torch.cuda.current_stream(self.dev).synchronize()
            rc = self._wait(self._done_ptr, count, value, 0) if self.late else 0   # (every word is stored by now: 0 or 2)
        return rc


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
        self.act_graphs = True  # get_action through a hipGraph per batch-size bucket (False: every call takes the general path)
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
        q = None  # drawn by the graph: after its launch when it takes late noise (ActGraph.run)
        if noise is not None:
            q = torch.as_tensor(noise, dtype=torch.float32)
            if tuple(q.shape) != tuple(self._noise_shape(n)):
                return None
            if not q.is_contiguous():
                q = q.contiguous()
        # a captured graph replays the kernels that were selected at capture time: the library's selection epoch (bumped by
        # rlppo_set_inference_precision and the A/B switches) is part of the cache key, stale graphs are dropped
        epoch = selection_epoch()
        if epoch != self._graph_epoch:
            self._graphs.clear()
            self._graph_epoch = epoch
        g = self._graphs.get(_bucket(n))
        if g is None:
            g = self._graphs[_bucket(n)] = ActGraph(self, _bucket(n))
        if o.dtype != np.float32:
            o = o.astype(np.float32)
        if a._packed_key is not None:
            # launch on the packed copy as it is and check that it was current WHILE the GPU works (8 Parameters' versions and
            # addresses: 5 us of the call's critical path otherwise); a stale one -- rare: a stock optimiser stepped, the module
            # moved -- costs a second launch
            return g.run(o, q, n, self._draw_bound(n), self._verify)
        a.ensure_packed()
        return g.run(o, q, n, self._draw_bound(n))

    def _draw_bound(self, n):
        return lambda: self._draw_noise(n)

    def _verify(self):
        a = self.arena
        return a.packed_is_current(), a.ensure_packed

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
