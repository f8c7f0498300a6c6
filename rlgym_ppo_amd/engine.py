"""Host-side plumbing between the reference-shaped Python classes and librlppo.so.

PyTorch is used for what the task allows it for -- device memory, streams, H2D/D2H copies and
torch.distributed -- never for the arithmetic of the hot path.  Everything numeric is a call into the C ABI
(include/rlppo.h) on torch's current HIP stream.
"""
import collections
import ctypes
import threading

import numpy as np
import torch

from . import _native as N


def require_gpu(device):
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(
            f"rlgym_ppo_amd runs its hot path on an AMD GPU through librlppo.so; device '{device}' is not supported "
            f"(there is deliberately no CPU fallback).")
    if not torch.cuda.is_available():
        raise RuntimeError("rlgym_ppo_amd: no HIP device visible to PyTorch")
    return dev


def _resolve_device(device):
    """torch.device with its index filled in ("cuda" -> the current device): what a tensor's .device compares equal to."""
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Tensor (or None, or a raw address: memory of a host window, ppo/_mlp.py::ActGraph) -> void *."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t if isinstance(t, int) else t.data_ptr())


class Workspace:
    """A grow-only device scratch buffer (the library never allocates)."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, nbytes):
        nbytes = int(nbytes)
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = None
            self.buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=self.device)
        return self.buf


class NetArena:
    """Device state of one MLP: flat parameter arena (parameters_to_vector order), flat gradient arena, and the
    tile-padded packed copy the kernels read.  The nn.Module's Parameters are re-pointed at views of `flat`, so
    state_dict()/load_state_dict() stay stock (keys model.{0,2,..}.{weight,bias})."""

    def __init__(self, linears, device):
        self.device = device
        self.linears = list(linears)
        self.dims = [self.linears[0].in_features] + [l.out_features for l in self.linears]
        self.n_layers = len(self.linears)
        self.dims_c = N.dims_array(self.dims)
        L = N.lib()
        self.n_flat = int(L.rlppo_flat_floats(self.dims_c, self.n_layers))
        self.n_packed = int(L.rlppo_packed_floats(self.dims_c, self.n_layers))
        self.d_in = self.dims[0]
        self.d_out = self.dims[-1]
        self.ld_in = int(L.rlppo_padded_width(self.d_in))
        self.ld_out = int(L.rlppo_padded_out(self.d_out))
        self.flat = torch.zeros(self.n_flat, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.n_flat, dtype=torch.float32, device=device)
        self.packed = torch.zeros(self.n_packed, dtype=torch.float32, device=device)
        self.ws = Workspace(device)
        self.packed_r = self.wb16 = None  # bf16 update precision: rounded images (rlppo_net_pack_bf16), built on first use
        self._packed_bf16_key = None
        self.x3 = None  # split-bf16 update precision: the three-plane images of the covered layers (rlppo_net_pack_x3)
        self._packed_x3_key = None
        self._packed_key = None
        self.native_epoch = 0  # bumped whenever a kernel rewrites `flat` behind torch's back (Adam)
        self._plan = None      # byte offsets of the parameters in `flat` (is_bound)
        self.bind()

    def params(self):
        for lin in self.linears:
            yield lin.weight
            yield lin.bias

    def bind(self):
        """(Re-)point every Parameter (and its .grad) at its slice of the arenas, keeping current values."""
        o = 0
        with torch.no_grad():
            for p in self.params():
                n = p.numel()
                view = self.flat[o:o + n].view(p.shape)
                if p.data.data_ptr() != view.data_ptr():
                    view.copy_(p.data.to(self.device))
                    p.data = view
                p.grad = self.grad[o:o + n].view(p.shape)
                o += n
        assert o == self.n_flat

    def is_bound(self):
        # (on the per-call path of get_action: Parameter.data_ptr() against cached byte offsets -- `p.data` would build an alias
        # tensor per parameter, 6-8 us per call for 8 parameters)
        plan = self._plan
        if plan is None or len(plan) != 2 * len(self.linears):
            o, plan = 0, []
            for p in self.params():
                plan.append(4 * o)
                o += p.numel()
            self._plan = plan
        base = self.flat.data_ptr()
        i = 0
        for lin in self.linears:
            if lin.weight.data_ptr() != base + plan[i] or lin.bias.data_ptr() != base + plan[i + 1]:
                return False
            i += 2
        return True

    def packed_is_current(self):
        """Would ensure_packed() have nothing to do?  (ActGraph.run asks AFTER its launch, while the GPU works: a call on a stale
        copy is repeated, and the 5 us of these checks leave the call's critical path.)"""
        return self.is_bound() and self._pack_key() == self._packed_key

    def ensure_packed(self):
        if not self.is_bound():  # e.g. the user called module.to(...) or replaced .data
            self.bind()
        key = self._pack_key()
        if key != self._packed_key:
            N.check(N.lib().rlppo_net_pack(stream_ptr(), self.dims_c, self.n_layers, ptr(self.flat), ptr(self.packed)))
            self._packed_key = key

    def ensure_packed_bf16(self):
        """The rounded weight images of the bf16 update precision (packed layout with bf16-rounded values as fp32 + the W
        blocks as bf16), rebuilt whenever the master weights changed.  Returns (packed_r, wb16)."""
        if not self.is_bound():
            self.bind()
        if self.packed_r is None:
            self.packed_r = torch.zeros(self.n_packed, dtype=torch.float32, device=self.device)
            self.wb16 = torch.zeros(int(N.lib().rlppo_wb16_elems(self.dims_c, self.n_layers)), dtype=torch.bfloat16, device=self.device)
        key = self._pack_key()
        if key != self._packed_bf16_key:
            N.check(N.lib().rlppo_net_pack_bf16(stream_ptr(), self.dims_c, self.n_layers, ptr(self.flat), ptr(self.packed_r),
                                                ptr(self.wb16)))
            self._packed_bf16_key = key
        return self.packed_r, self.wb16

    def ensure_packed_x3(self):
        """[r4] The weight image of the split-bf16 update precision (set_update_precision("x3")): for every hidden forward / dX product
        the split kernels cover, the layer's weights as three bf16 planes in stage-major order (include/rlppo.h, rlppo_net_pack_x3),
        derived from the PACKED copy and rebuilt whenever the master weights changed.  Returns the image (a bf16 tensor)."""
        if not self.is_bound():
            self.bind()
        self.ensure_packed()
        if self.x3 is None:
            self.x3 = torch.zeros(max(int(N.lib().rlppo_x3_elems(self.dims_c, self.n_layers)), 8), dtype=torch.bfloat16, device=self.device)
        key = self._pack_key()
        if key != self._packed_x3_key:
            N.check(N.lib().rlppo_net_pack_x3(stream_ptr(), self.dims_c, self.n_layers, ptr(self.packed), ptr(self.x3)))
            self._packed_x3_key = key
        return self.x3

    def _pack_key(self):
        # bind() re-points every Parameter with `p.data = view`, which gives it a version counter of its own: in-place updates
        # made THROUGH the Parameters (a stock torch optimiser on get_backprop_data's graph, vector_to_parameters, p.data.mul_())
        # bump p._version, not flat._version -- both are part of the key; native_epoch covers the kernels that rewrite `flat`
        # behind torch's back (Adam)
        v = 0
        for lin in self.linears:
            v += lin.weight._version + lin.bias._version
        return (self.flat._version, self.native_epoch, v)

    def mark_repacked(self):
        """A kernel has just updated `flat` AND written the new values into `packed` (rlppo_clip_adam_pack2)."""
        self.native_epoch += 1
        self._packed_key = self._pack_key()

    def invalidate(self):
        """Force a re-pack before the next kernel reads the weights (for writers that bypass both version counters, e.g.
        a raw pointer write into the arena)."""
        self._packed_key = self._packed_bf16_key = self._packed_x3_key = None  # (the bf16 / split-bf16 images derive from the same weights)

    # ------------------------------------------------------------------------------------------ inference
    def stage_obs(self, obs, standardize=None, out=None):
        """numpy / tensor observations of any float dtype -> zero-padded fp32 device rows [n, ld_in]
        (written into `out` when given: the rollout storage's slot of the current step).  `standardize`: None, the
        reference's scalar pair (mean0, std0) or a pair of per-feature tensors (mean[d], std[d])."""
        if isinstance(obs, torch.Tensor):
            t = obs.detach()
            if t.dtype not in (torch.float32, torch.float64):
                t = t.float()
            t = t.to(self.device).contiguous()
        else:
            a = np.asarray(obs)
            if a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float32)
            t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device, non_blocking=False)
        if t.dim() == 1:
            t = t.view(1, -1)
        t = t.reshape(-1, t.shape[-1]) if t.dim() > 2 else t
        n, d = t.shape
        if d != self.d_in:
            raise ValueError(f"observation width {d} != network input {self.d_in}")
        if out is None:
            out = torch.empty((n, self.ld_in), dtype=torch.float32, device=self.device)
        elif out.shape != (n, self.ld_in) or not out.is_contiguous() or out.dtype != torch.float32:
            raise ValueError("stage_obs: out must be a contiguous fp32 [n, ld_in] device tensor")
        mean0, std0, flag = 0.0, 1.0, 0
        if standardize is not None:
            if isinstance(standardize[0], torch.Tensor):  # per-feature statistics: device vectors (mean[d], std[d])
                mean_v, std_v = (x.to(self.device, dtype=torch.float32).contiguous() for x in standardize)
                if mean_v.numel() != d or std_v.numel() != d:
                    raise ValueError("stage_obs: per-feature statistics must have one entry per observation feature")
                N.check(N.lib().rlppo_pad_rows_per_feature(stream_ptr(), ptr(t), int(t.dtype == torch.float64), n, d, d, ptr(out),
                                                           self.ld_in, ptr(mean_v), ptr(std_v)))
                return out
            mean0, std0, flag = float(standardize[0]), float(standardize[1]), 1
        N.check(N.lib().rlppo_pad_rows(stream_ptr(), ptr(t), int(t.dtype == torch.float64), n, d, d, ptr(out), self.ld_in,
                                       flag, mean0, std0))
        return out

    def forward_ws(self, n):
        return self.ws.get(N.lib().rlppo_forward_workspace_bytes(self.dims_c, self.n_layers, n))

    def forward(self, obs_padded, out_tanh=False):
        """[n, ld_in] padded device rows -> [n, ld_out] raw outputs (padded)."""
        self.ensure_packed()
        n = obs_padded.shape[0]
        out = torch.empty((n, self.ld_out), dtype=torch.float32, device=self.device)
        ws = self.forward_ws(n)
        N.check(N.lib().rlppo_mlp_forward(stream_ptr(), self.dims_c, self.n_layers, ptr(self.packed), ptr(obs_padded),
                                          obs_padded.shape[1], n, int(out_tanh), ptr(out), self.ld_out, ptr(ws), ws.numel(), None))
        return out


def linears_of(sequential):
    return [m for m in sequential if isinstance(m, torch.nn.Linear)]


# ----------------------------------------------------------------------------------------------- shuffle
_POOLS = {}


def _pool(name, workers):
    """Process-wide helper threads of the shuffle pipeline (the C calls release the GIL)."""
    if name not in _POOLS:
        from concurrent.futures import ThreadPoolExecutor
        _POOLS[name] = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="rlppo-" + name)
    return _POOLS[name]


class _ShuffleEntry:
    __slots__ = ("n", "after", "out", "slot", "done", "error")

    def __init__(self, n):
        self.n, self.after, self.out, self.slot, self.error = n, None, None, None, None
        self.done = threading.Event()


class LegacyPermutation:
    """numpy.random.RandomState(seed).permutation(n) through the library's host implementation (bit-identical stream),
    keeping the numpy generator object in sync so that user code that touches `buffer.rng` still sees the reference's
    state (experience_buffer.py:52,97-98).

    The permutation is produced in two phases on helper threads: the serial stream phase (rlppo_mt19937_draw_targets,
    one thread, epochs in order) and the swap phase (rlppo_apply_swap_targets, any thread).  With `lookahead` > 0 the
    pipeline also draws that many permutations of the same size AHEAD of the request (an epoch loop asks for the same n
    again and again); they are speculative: a request is served from the queue only if n matches and the numpy generator
    is still in exactly the state the queue was drawn from -- otherwise the queue is dropped and the draw restarts from the
    generator's current state, so the observable stream never differs from the reference's.

    `ring` (optional, see DeviceIndexRing) supplies the output arrays and is told when one is complete."""

    def __init__(self, rng, lookahead=0, ring=None):
        self.rng = rng
        self.lookahead = int(lookahead)
        self._ring = ring
        self._q = collections.deque()
        self._chain = None      # {"st": uint32[625] owned by the draw thread, "dead": bool}
        self._expected = None   # generator state the head of the queue was drawn from
        self._seq = 0
        # [r6] a live uint32[625] view of the generator's MT19937 state (key[624], pos) through numpy's documented BitGenerator
        # interface (`.ctypes.state_address`): comparing / advancing the state costs 3 us instead of the 50 + 50 us of
        # get_state() / set_state(), which sat in front of every learn()'s first launch.  Checked once against get_state();
        # anything unexpected leaves the legacy accessors in charge.
        self._view = None
        try:
            bg = rng._bit_generator
            view = np.ctypeslib.as_array((ctypes.c_uint32 * 625).from_address(bg.ctypes.state_address))
            kind, key, pos = rng.get_state()[:3]
            if kind == "MT19937" and np.array_equal(view[:624], key) and int(view[624]) == int(pos):
                self._view, self._bg = view, bg   # (the view borrows the bit generator's memory: keep it alive)
        except Exception:  # noqa: BLE001 -- another numpy: get_state / set_state
            self._view = None

    def _rng_state(self):
        kind, key, pos, has_gauss, cached = self.rng.get_state()
        assert kind == "MT19937"
        st = np.empty(625, dtype=np.uint32)
        st[:624] = key
        st[624] = pos
        return st, (kind, has_gauss, cached)

    def _enqueue(self, n):
        e = _ShuffleEntry(n)
        ring, chain = self._ring, self._chain
        if ring is not None:
            e.slot = self._seq % ring.slots
            e.out = ring.host_array(e.slot, n)
        else:
            e.out = np.empty(n, dtype=np.int64)
        self._seq += 1
        targets = np.empty(max(n - 1, 0) + 8, dtype=np.uint32)
        L = N.lib()

        def apply():
            try:
                N.check(L.rlppo_apply_swap_targets(n, ctypes.c_void_p(targets.ctypes.data), ctypes.c_void_p(e.out.ctypes.data)))
                if ring is not None:
                    ring.after_write(e.slot, n)
            except BaseException as ex:  # noqa: BLE001 -- handed to the consumer
                e.error = ex
            finally:
                e.done.set()

        def draw():
            try:
                if chain.get("dead"):
                    raise RuntimeError("shuffle pipeline: entry dropped")
                if ring is not None:
                    ring.before_write(e.slot)
                st = chain["st"]
                N.check(L.rlppo_mt19937_draw_targets(st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n,
                                                     ctypes.c_void_p(targets.ctypes.data)))
                e.after = st.copy()
                _pool("swap", 2).submit(apply)
            except BaseException as ex:  # noqa: BLE001
                e.error = ex
                e.done.set()

        self._q.append(e)
        _pool("draw", 1).submit(draw)

    def _flush(self):
        if self._chain is not None:
            self._chain["dead"] = True
        for e in self._q:
            e.done.wait()
        self._q.clear()
        self._chain = None

    close = _flush

    def take(self, n, refill=True):
        """The next permutation of arange(n) as a completed pipeline entry (`.out`: int64 host array, `.slot`).  refill=False: the
        queue is not topped up before returning (allocating the next entry and waking a helper thread costs ~50 us; a caller on a
        latency-critical path calls refill() itself once its launches are out)."""
        n = int(n)
        view = self._view
        if view is not None:
            st, extra = view, None
        else:
            st, extra = self._rng_state()
        if not (self._q and self._q[0].n == n and self._expected is not None and np.array_equal(st, self._expected)):
            self._flush()
            if self._ring is not None:
                self._ring.ensure(n)
            self._chain = {"st": st.copy()}
        while len(self._q) < 1 + self.lookahead:
            self._enqueue(n)
        e = self._q.popleft()
        e.done.wait()
        if e.error is not None:
            self._flush()
            raise e.error
        if view is not None:
            view[:] = e.after   # (key and position: what set_state would write; the legacy Gaussian cache is not touched by either)
        else:
            self.rng.set_state((extra[0], e.after[:624].copy(), int(e.after[624]), extra[1], extra[2]))
        self._expected = e.after
        self._refill_n = n
        if refill:
            self.refill()
        return e

    def refill(self):
        """Top the look-ahead queue up after a take(n, refill=False)."""
        n = getattr(self, "_refill_n", None)
        if n is not None and self._chain is not None and not self._chain.get("dead"):
            # (1 + lookahead is what the next take() wants to find: it then enqueues nothing before its wait; lookahead 0 = nothing speculative)
            while len(self._q) < (1 + self.lookahead if self.lookahead > 0 else 0):
                self._enqueue(n)

    def permutation(self, n):
        return self.take(n).out


class DeviceIndexRing:
    """Output side of the shuffle pipeline for the PPO update: every permutation is written by the swap thread straight into
    a pinned host vector and copied to HBM from that thread on a dedicated upload stream, so an epoch's index vector is
    already resident when the epoch starts and the learner thread only orders its stream after the copy's event.

    Slot life cycle (ring of `slots` >= lookahead + 2 vectors): swap thread writes pinned[slot] -> H2D on the upload stream,
    `copied[slot]` recorded -> learner takes it (its stream waits for `copied`) -> before the learner asks for the NEXT vector
    `released[slot]` is recorded on its stream -> the slot's next H2D waits for `released` (the kernels of that epoch are done
    reading) and the next host write waits for `copied` (the previous H2D no longer reads the pinned vector).  A host view
    returned by LegacyPermutation.permutation() therefore stays valid until the next-but-one request."""

    def __init__(self, device, slots):
        self.device = torch.device(device)
        self.slots = int(slots)
        self.cap = 0
        self.pinned = self.dev = None
        self.stream = None
        self.copied = [torch.cuda.Event() for _ in range(self.slots)]
        self.released = [torch.cuda.Event() for _ in range(self.slots)]
        self._held = None

    def ensure(self, n):
        """Called with no pipeline entry outstanding (the queue was just dropped)."""
        if self.stream is None:
            self.stream = torch.cuda.Stream(self.device)
        n = max(int(n), 1)
        if n > self.cap:
            torch.cuda.synchronize(self.device)  # rare: the buffer grew; nothing may still read the old vectors
            self.cap = int(n)
            self.pinned = [torch.empty(self.cap, dtype=torch.int64).pin_memory() for _ in range(self.slots)]
            self.dev = [torch.empty(self.cap, dtype=torch.int64, device=self.device) for _ in range(self.slots)]

    def host_array(self, slot, n):
        return self.pinned[slot][:n].numpy()

    def before_write(self, slot):
        self.copied[slot].synchronize()

    def after_write(self, slot, n):
        with torch.cuda.device(self.device):
            self.stream.wait_event(self.released[slot])
            with torch.cuda.stream(self.stream):
                self.dev[slot][:n].copy_(self.pinned[slot][:n], non_blocking=True)
            self.copied[slot].record(self.stream)

    def release_held(self):
        """Learner thread, before asking the pipeline for the next vector: everything that reads the vector handed out last
        has been enqueued on the current stream; its slot may be refilled once that work is done."""
        if self._held is not None:
            self.released[self._held].record(torch.cuda.current_stream(self.device))
            self._held = None

    def take(self, entry):
        """Learner thread: device view of a completed entry, ordered on the current stream."""
        self.release_held()
        torch.cuda.current_stream(self.device).wait_event(self.copied[entry.slot])
        self._held = entry.slot
        return self.dev[entry.slot][:entry.n]


# ------------------------------------------------------------------------------------------- sampling noise
class HostExponential:
    """torch.empty(shape).exponential_(1) of the global CPU generator -- the stream torch.multinomial(probs, 1, True) consumes
    in the reference's CPU rollout (discrete_policy.py:59), which is what makes a seeded run pick the reference's action
    indices -- produced by librlppo's host implementation (rlppo_torch_cpu_exponential: bit-identical values and generator
    advance, pinned by tests against torch itself; AVX-512 / AVX2 stream phase + a vectorised logarithm certified element by
    element against the float32 rounding, libm where it cannot be, instead of torch's serial 12-26 ns per number) and drawn
    AHEAD on helper threads.

    The look-ahead is speculative and transparent, the contract of LegacyPermutation: after a draw of `shape` the helpers draw
    the same shape again -- `depth` requests deep [r3: a chain of three, was one] -- from the states the generator WILL be in if
    nobody else uses it, into pinned buffers; a request is served from the head of the chain only if the shape matches and
    torch's global generator is in exactly the predicted state, and then the generator is advanced to the state after that
    draw -- otherwise the chain is dropped and the draw happens on the spot.  The observable stream never differs from torch's.
    [r3] Each speculative draw is ONE call of rlppo_torch_cpu_exponential_chained on one of two helper threads: the call waits
    (in C, spinning on a link block) until its predecessor's serial MT19937 stream phase is done, takes the state that phase
    left behind, runs its own stream phase, publishes its link block and only then transforms -- so the transform of draw k
    overlaps the whole of draw k + 1, and a 4096 x 90 draw leaves the pair every ~0.2 ms instead of every 0.36 ms, with no
    Python (no GIL) between the phases.  Buffers are pinned (a ring of depth + 3): the caller uploads them with an
    asynchronous copy and must have consumed a buffer before the ring comes round.
    [r4] prefetch(shape, count): a caller that knows its next `count` draws (the rollout of the NEXT iteration: 128 draws of
    [4096, 90]) extends the chain to that length in one go -- two C calls (rlppo_torch_cpu_exponential_burst), one per helper
    thread, that walk the burst's draws alternately into one pinned block -- so that the draws happen while PPOLearner.learn
    keeps the GPU busy (learn() does not touch torch's CPU generator) and the collect's critical path only uploads.  Same
    contract: every entry is served only from exactly the predicted generator state, anything else drops the chain."""

    DEPTH = 3  # with two helper threads: a third request always waits in their queue, so neither goes to sleep between draws (tools/host_noise_pipeline.py)
    LOOKAHEAD_MIN = 65536  # elements: below that the hand-over to the helper thread (~50 us) costs more than the draw itself (~1 ns per number)

    def __init__(self):
        import os
        look = os.environ.get("RLPPO_NOISE_LOOKAHEAD", "")
        self.depth = self.DEPTH if look == "" else max(0, int(look))  # requests drawn ahead (0: on demand)
        self.RING = self.depth + 3
        self.workers = max(1, int(os.environ.get("RLPPO_NOISE_THREADS", "2")))  # helper threads the chain's draws alternate over
        self.threads = max(1, min(8, (os.cpu_count() or 2) // 2))  # an upper bound: the library uses one thread per 2^20 elements (0.35 ms per rollout-step draw)
        self._buf = {}       # numel -> dict(ring=[pinned float32 vectors], uploaded=[event or None], turn)
        self._chain = []      # speculative requests in stream order: dict(shape, numel, state_in, link, future, buf, rec, slot)
        self._burst = {}      # numel -> dict(out=[cap, numel] pinned, links, cancel, uploaded, futures): storage of prefetch()
        self._side = {}       # device -> the stream a finished burst is uploaded on
        self._burst_lock = threading.Lock()
        self.resident_hits = 0  # draws served from a burst that was already in HBM
        self.burst_bytes = int(os.environ.get("RLPPO_NOISE_PREFETCH_MB", "256")) << 20  # byte budget of one prefetched burst
        self.hits = self.misses = 0
        self.checked = False  # the one-time self-check against torch's own exponential_ (see _self_check)
        self.trusted = True

    def _buffer(self, numel):
        """Next pinned vector of the ring for this size, as (ring record, slot).  A slot whose last hand-out was uploaded with
        an asynchronous copy (upload()) is reused only after that copy has run: the helper thread writes into it next."""
        rec = self._buf.get(numel)
        if rec is None:
            pin = torch.cuda.is_available()
            rec = self._buf[numel] = dict(ring=[torch.empty(numel, dtype=torch.float32, pin_memory=pin) for _ in range(self.RING)],
                                          scratch=[None] * self.RING, uploaded=[None] * self.RING, turn=0)
            if len(self._buf) > 8:  # shapes come and go (worker counts change): keep the cache small -- but never a ring the
                busy = {e["numel"] for e in self._chain}  # pending chain is writing into
                for key in list(self._buf):
                    if key != numel and key not in busy:
                        self._buf.pop(key)
                        break
        rec["turn"] = slot = (rec["turn"] + 1) % self.RING
        ev = rec["uploaded"][slot]
        if ev is not None:
            ev.synchronize()
            rec["uploaded"][slot] = None
        return rec, slot

    def _draw_into(self, state, buf, numel):
        """state: uint8 tensor (a private copy of a generator state), advanced in place."""
        N.check(N.lib().rlppo_torch_cpu_exponential(ctypes.c_void_p(state.data_ptr()), state.numel(), numel, 1.0,
                                                    ctypes.c_void_p(buf.data_ptr()), self.threads))
        return state

    def _self_check(self):
        """Once per process: 4096 values drawn both ways from the same generator state must agree bit for bit, values AND the
        state the generator is left in.  rlppo_torch_cpu_exponential restates the CPU path of THIS torch build; on
        another build (a changed transform, a vectorised CPU kernel) the stream could differ silently and a seeded run would no
        longer pick the reference's actions -- then torch's own exponential_ is used (slower, always right) and a warning says so."""
        self.checked = True
        keep = torch.get_rng_state()
        try:
            want = torch.empty(4096).exponential_(1)
            after_torch = torch.get_rng_state()
            got = torch.empty(4096)
            after_lib = self._draw_into(keep.clone(), got, 4096)
            self.trusted = bool(torch.equal(want, got) and torch.equal(after_torch, after_lib))
        except Exception:  # noqa: BLE001 -- e.g. an unexpected generator state size
            self.trusted = False
        finally:
            torch.set_rng_state(keep)
        if not self.trusted:
            import warnings
            warnings.warn("rlgym_ppo_amd: librlppo's host exponential_ does not reproduce this torch build's CPU stream; "
                          "falling back to torch.empty(shape).exponential_(1) for the rollout noise (slower, same results)")

    @property
    def lookahead(self):
        return self.depth > 0

    @lookahead.setter
    def lookahead(self, on):
        self.depth = (self.depth or self.DEPTH) if on else 0

    def _chained_draw(self, e, prev_link):
        """Helper thread: one C call (waits for the predecessor's stream phase inside the library)."""
        N.check(N.lib().rlppo_torch_cpu_exponential_chained(
            ctypes.c_void_p(e["state_in"].data_ptr()) if prev_link is None else None, e["state_bytes"], e["numel"], 1.0,
            ctypes.c_void_p(e["buf"].data_ptr()), ctypes.c_void_p(e["words"].ctypes.data),
            ctypes.c_void_p(prev_link.ctypes.data) if prev_link is not None else None, ctypes.c_void_p(e["link"].ctypes.data)))

    def _guarded_draw(self, e, prev_link):
        """Whatever goes wrong on the helper thread, the successor (spinning inside the library on this request's link block) is
        released: a failed link passes the failure down the chain and draw() falls back to drawing on the spot."""
        try:
            self._chained_draw(e, prev_link)
        except BaseException:
            e["link"][:4] = np.frombuffer(np.int32(-1).tobytes(), np.uint8)
            raise

    def _speculate(self, shape, numel, state_now):
        """Top the chain up to `depth` requests of `shape`; the first one starts from state_now (the generator's state)."""
        if self._chain and self._chain[-1]["shape"] != shape:
            self._drain()
        nbytes = state_now.numel()
        while len(self._chain) < self.depth:
            rec, slot = self._buffer(numel)
            if rec["scratch"][slot] is None:  # (words, link block) of this ring slot
                rec["scratch"][slot] = (np.empty(2 * numel + 8, np.uint32), np.zeros(N.EXP_LINK_HEADER + nbytes, np.uint8))
            words, link = rec["scratch"][slot]
            link[:4] = 0
            prev = self._chain[-1] if self._chain else None
            e = dict(shape=shape, numel=numel, state_bytes=nbytes, state_in=state_now if prev is None else None, prev=prev, link=link,
                     words=words, buf=rec["ring"][slot], rec=rec, slot=slot)
            e["future"] = _pool("noise", self.workers).submit(self._guarded_draw, e, prev["link"] if prev is not None else None)
            self._chain.append(e)

    def _drain(self):
        """Drop the chain: wait for everything in flight first (its buffers go back into rotation)."""
        for b in self._burst.values():  # a burst in flight stops at its next draw (the remaining ones are marked failed)
            b["cancel"][0] = 1
        for e in self._chain:
            try:
                e["future"].result()
            except Exception:  # noqa: BLE001 -- a failed speculation is simply not used
                pass
        for b in self._burst.values():
            for f in b["futures"]:
                try:
                    f.result()
                except Exception:  # noqa: BLE001
                    pass
            b["futures"] = []
        self._chain = []

    @staticmethod
    def _link_state(e):
        return torch.from_numpy(e["link"][N.EXP_LINK_HEADER:])

    class _BurstSlot:
        """Stands in for the future of one draw of a burst: result() returns once the draw's values are complete."""

        TIMEOUT_S = 10.0  # the bound of the chained C path's own waits

        def __init__(self, link, rec=None, run=0):
            self.link = link.view(np.int32)
            self.rec, self.run = rec, run   # the burst record and the index of the helper-thread run that produces this draw

        def done(self):
            return self.link[1] != 0

        def result(self):
            import time
            spins, t0 = 0, None
            while self.link[1] == 0:   # normally long done (the burst ran during learn())
                spins += 1
                if spins >= 200:
                    # [r5, advisor] never an unbounded wait: a run that died before marking its links (an early argument-check
                    # return of the C call, a helper thread that never reached it) leaves them 0 -- its pool future then says so
                    fut = self.rec["futures"][self.run] if self.rec is not None and self.run < len(self.rec["futures"]) else None
                    if fut is not None and fut.done() and self.link[1] == 0:
                        raise RuntimeError("burst draw failed: its helper-thread run ended without producing it (%r)" % (fut.exception(),))
                    t0 = t0 or time.monotonic()
                    if time.monotonic() - t0 > self.TIMEOUT_S:
                        raise RuntimeError("burst draw timed out after %.0f s" % self.TIMEOUT_S)
                    time.sleep(0.0001)
                else:
                    time.sleep(0)
            if self.link[1] < 0:
                raise RuntimeError("burst draw failed or was cancelled")

    def _burst_run(self, b, state_in, link_in0, nbytes, numel, first, count):
        try:
            self._burst_run_inner(b, state_in, link_in0, nbytes, numel, first, count)
        except BaseException:
            # [r5, advisor] whatever went wrong, nobody may wait for this run's draws: mark every link it owns that is not complete
            # as failed (ready = -1 lets a successor chained to it stop too) before the exception goes to the pool future
            links = b["links"]
            for i in range(first, count, self.workers):
                w = links[i].view(np.int32)
                if w[1] == 0:
                    w[0], w[1] = -1, -1
            raise

    def _burst_run_inner(self, b, state_in, link_in0, nbytes, numel, first, count):
        N.check(N.lib().rlppo_torch_cpu_exponential_burst(
            ctypes.c_void_p(state_in.data_ptr()) if state_in is not None else None,
            ctypes.c_void_p(link_in0.ctypes.data) if link_in0 is not None else None, nbytes, numel, 1.0,
            ctypes.c_void_p(b["out"].data_ptr()), numel, ctypes.c_void_p(b["links"].ctypes.data), b["links"].shape[1],
            first, self.workers, count, ctypes.c_void_p(b["cancel"].ctypes.data)))
        pend = b.get("dev_pending")
        if pend is not None:
            with self._burst_lock:
                pend[0] -= 1
                last = pend[0] == 0
            if last and b["cancel"][0] == 0:  # every run of the burst is complete: ONE copy of the whole block, on the side stream
                n_rows, dev = pend[1], pend[2]
                with torch.cuda.device(dev):
                    side = self._side.get(str(dev))
                    if side is None:
                        side = self._side[str(dev)] = torch.cuda.Stream(device=dev)
                    side.wait_event(b["dev_free"])
                    with torch.cuda.stream(side):
                        b["dev"][:n_rows].copy_(b["out"][:n_rows], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(side)
                b["dev_event"], b["dev_waited"] = ev, False
                b["uploaded"][0] = ev      # the pinned block is not redrawn into before the copy has run (prefetch waits on it)
                b["dev_ok"] = True

    def prefetch(self, shape, count, device=None):
        """The caller's next `count` draws will be of `shape`, and nothing else will use torch's CPU generator before them (if
        something does, the chain is dropped at the next draw as ever: transparent).  Extends the speculative chain to `count`
        requests -- capped by `burst_bytes` -- and returns the number of requests added.  Returns at once: the draws run on the
        helper threads (two C calls, no Python between draws).  With `device` (a CUDA device) the finished burst is copied to HBM in
        ONE asynchronous copy on a side stream, enqueued by the helper thread that finishes last: the draws it serves then cost
        neither a copy nor an event per step.  (Speculative too: values that are never served are simply never read.)"""
        shape = tuple(int(x) for x in shape)
        numel = 1
        for x in shape:
            numel *= x
        if self.depth == 0 or numel < self.LOOKAHEAD_MIN or count <= 0:
            return 0
        if not self.checked:
            self._self_check()
        if not self.trusted:
            return 0
        if self._chain and self._chain[-1]["shape"] != shape:
            self._drain()
        count = min(int(count), self.burst_bytes // (4 * numel))
        need = count - len(self._chain)
        if need <= 0:
            return 0
        state_now = torch.get_rng_state()
        nbytes = state_now.numel()
        b = self._burst.get(numel)
        if b is not None and (b["futures"] or any(e.get("rec") is b for e in self._chain)):
            if any(e.get("rec") is b for e in self._chain) or not all(f.done() for f in b["futures"]):
                return 0  # the previous burst on this storage is still being produced or served
            b["futures"] = []
        if b is None or b["out"].shape[0] < need:
            pin = torch.cuda.is_available()
            b = self._burst[numel] = dict(out=torch.empty((need, numel), dtype=torch.float32, pin_memory=pin),
                                          links=np.zeros((need, N.EXP_LINK_HEADER + nbytes), np.uint8), cancel=np.zeros(1, np.int32),
                                          uploaded=[None] * need, futures=[])
        for i, ev in enumerate(b["uploaded"]):  # uploads of the previous burst's slots (stream-ordered: normally long complete)
            if ev is not None:
                ev.synchronize()
                b["uploaded"][i] = None
        b["links"][:need, :8] = 0
        b["cancel"][0] = 0
        prev = self._chain[-1] if self._chain else None
        pool = _pool("noise", self.workers)
        b["dev_ok"] = False
        if device is not None and torch.device(device).type == "cuda" and torch.cuda.is_available():
            dev = _resolve_device(device)  # ("cuda" and "cuda:0" must compare equal with a tensor's device: advisor)
            if b.get("dev") is None or b["dev"].device != dev or b["dev"].shape[0] < b["out"].shape[0]:
                b["dev"] = torch.empty(b["out"].shape, dtype=torch.float32, device=dev)
            # the block is rewritten only behind everything launched so far (the kernels that read the previous burst)
            b["dev_free"] = torch.cuda.Event()
            b["dev_free"].record(torch.cuda.current_stream(dev))
            b["dev_pending"] = [min(self.workers, need), need, dev]
        else:
            b["dev_pending"] = None
        b["futures"] = [pool.submit(self._burst_run, b, state_now if prev is None else None, prev["link"] if prev is not None else None,
                                    nbytes, numel, j, need) for j in range(min(self.workers, need))]
        for i in range(need):
            link = b["links"][i]
            e = dict(shape=shape, numel=numel, state_bytes=nbytes, state_in=state_now if (prev is None and i == 0) else None, prev=prev,
                     link=link, words=None, buf=b["out"][i], rec=b, slot=i,
                     future=HostExponential._BurstSlot(link, b, i % min(self.workers, need)))
            self._chain.append(e)
            prev = e
        return need

    def draw(self, shape, device=None):
        """Exp(1) noise of `shape`: a view of a pinned ring buffer, or -- with `device` -- its asynchronous upload on the current
        stream (the ring slot is then protected by an event until the copy has run)."""
        shape = tuple(int(x) for x in shape)
        numel = 1
        for x in shape:
            numel *= x
        if numel == 0:
            return torch.empty(shape, device=device)
        if not self.checked:
            self._self_check()
        if not self.trusted:
            out = torch.empty(shape).exponential_(1)
            return out if device is None else out.to(device)
        state = torch.get_rng_state()
        head = self._chain[0] if self._chain else None
        served = False
        pre = None  # (burst record, index) when the served request's values are already in HBM
        if head is not None and head["shape"] == shape:
            # the state this request was drawn from: given for the first of a chain, else what its predecessor (already served:
            # the generator was set to exactly that) left behind
            drawn_from = head["state_in"] if head["state_in"] is not None else head.get("prev_state")
            ok = drawn_from is not None and torch.equal(drawn_from, state)
            if ok:
                try:
                    head["future"].result()
                    served = True
                except Exception:  # noqa: BLE001 -- fall back to the draw on the spot
                    served = False
            if served:
                self._chain.pop(0)
                after = self._link_state(head).clone()
                buf, rec, slot = head["buf"], head["rec"], head["slot"]
                pre = (rec, slot) if rec.get("dev") is not None and rec.get("dev_ok") else None
                if self._chain:
                    self._chain[0]["prev_state"] = after  # what the new head must find the generator in
                    self._chain[0]["prev"] = None
                self.hits += 1
        if not served:
            if self._chain:
                self._drain()  # somebody else used the generator, or the shape changed: nothing drawn ahead is valid
            rec, slot = self._buffer(numel)
            buf = rec["ring"][slot]
            after = self._draw_into(state.clone(), buf, numel)
            self.misses += 1
        torch.set_rng_state(after)
        out = buf.view(shape)
        on_gpu = device is not None and torch.device(device).type == "cuda"
        if on_gpu and pre is not None and pre[0]["dev"].device == _resolve_device(device):
            # [r4] the whole burst went to HBM when its last draw finished (during learn()): no copy, no event per step
            b_, i_ = pre
            if not b_["dev_waited"]:
                torch.cuda.current_stream(b_["dev"].device).wait_event(b_["dev_event"])
                b_["dev_waited"] = True
            out = b_["dev"][i_].view(shape)
            self.resident_hits += 1
        elif on_gpu:
            out = out.to(device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(out.device))
            rec["uploaded"][slot] = ev
        if self.depth > 0 and numel >= self.LOOKAHEAD_MIN:
            self._speculate(shape, numel, after)
        return out


_HOST_EXP = None


def host_exponential_prefetch(shape, count, device=None):
    """HostExponential.prefetch on the process-wide instance (see there): `count` upcoming draws of `shape` are produced ahead
    (and, with `device`, parked in HBM as soon as they are complete)."""
    global _HOST_EXP
    if _HOST_EXP is None:
        _HOST_EXP = HostExponential()
    return _HOST_EXP.prefetch(shape, count, device)


def host_exponential(shape, device=None):
    """Exp(1) noise of `shape` from torch's global CPU generator (values and generator advance identical to
    torch.empty(shape).exponential_(1)).  device=None: a view of a recycled pinned buffer -- consume it (copy it synchronously)
    before the next-but-one call.  device=cuda: the noise uploaded on the current stream with an asynchronous copy; the pinned
    slot is not rewritten until that copy has run (an event per ring slot), so callers may keep the rollout on the device without a
    host synchronisation per step."""
    global _HOST_EXP
    if _HOST_EXP is None:
        _HOST_EXP = HostExponential()
    return _HOST_EXP.draw(shape, device)


_EPOCH = None


def selection_epoch():
    """Counter the library bumps whenever a call changes which kernels later launches select (precision setters, A/B
    switches): captured graphs of library calls are keyed on it (ppo/_mlp.py::ActGraph).  (Read in place: it sits on the
    per-call path of get_action.)"""
    global _EPOCH
    if _EPOCH is None:
        import ctypes
        _EPOCH = ctypes.c_int64.from_address(N.lib().rlppo_selection_epoch_ptr())
    return _EPOCH.value


def set_update_precision(mode):
    """Precision of the PPO update (PPOLearner.learn): "fp32" (default: the reference's arithmetic, 1e-5 parity) or "bf16" =
    BASELINE configs[4] "bf16 fwd / fp32 master weights", i.e. mixed-precision training: forward AND backward products on bf16
    operands (bf16 activations and activation gradients, bf16 MFMA, fp32 accumulate), fp32 losses / dW, db accumulation / clip /
    Adam on the fp32 master arena (include/rlppo.h)."""
    N.check(N.lib().rlppo_set_update_precision({"fp32": 0, "bf16": 1, "x3": 2}[mode]))


def update_precision():
    return {0: "fp32", 1: "bf16", 2: "x3"}[int(N.lib().rlppo_get_update_precision())]


def set_inference_precision(mode):
    """Precision of the rollout forward passes (get_action / value predictions): "fp32" (default: the reference's
    arithmetic, bit-exact action indices) or "bf16" (activations and master weights rounded to bf16 as MFMA operands,
    fp32 accumulation; BASELINE configs[4]).  The PPO update always runs in fp32."""
    from . import _native as N
    N.check(N.lib().rlppo_set_inference_precision({"fp32": 0, "bf16": 1}[mode]))
