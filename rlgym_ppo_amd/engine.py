"""Host-side plumbing between the reference-shaped Python classes and librlppo.so.

PyTorch is used for what the task allows it for -- device memory, streams, H2D/D2H copies and
torch.distributed -- never for the arithmetic of the hot path.  Everything numeric is a call into the C ABI
(include/rlppo.h) on torch's current HIP stream.
"""
import ctypes

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


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


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
        self._packed_key = None
        self.native_epoch = 0  # bumped whenever a kernel rewrites `flat` behind torch's back (Adam)
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
        o = 0
        for p in self.params():
            if p.data.data_ptr() != self.flat.data_ptr() + 4 * o:
                return False
            o += p.numel()
        return True

    def ensure_packed(self):
        if not self.is_bound():  # e.g. the user called module.to(...) or replaced .data
            self.bind()
        key = (self.flat._version, self.native_epoch)
        if key != self._packed_key:
            N.check(N.lib().rlppo_net_pack(stream_ptr(), self.dims_c, self.n_layers, ptr(self.flat), ptr(self.packed)))
            self._packed_key = key

    # ------------------------------------------------------------------------------------------ inference
    def stage_obs(self, obs, standardize=None, out=None):
        """numpy / tensor observations of any float dtype -> zero-padded fp32 device rows [n, ld_in]
        (written into `out` when given: the rollout storage's slot of the current step)."""
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
                                          obs_padded.shape[1], n, int(out_tanh), ptr(out), self.ld_out, ptr(ws), ws.numel()))
        return out


def linears_of(sequential):
    return [m for m in sequential if isinstance(m, torch.nn.Linear)]


# ----------------------------------------------------------------------------------------------- shuffle
class LegacyPermutation:
    """numpy.random.RandomState(seed).permutation(n) through the library's host implementation (bit-identical
    stream, ~4x faster, releases the GIL), keeping the numpy generator object in sync so that user code that
    touches `buffer.rng` still sees the reference's state."""

    def __init__(self, rng):
        self.rng = rng

    def permutation(self, n):
        kind, key, pos, has_gauss, cached = self.rng.get_state()
        assert kind == "MT19937"
        st = np.empty(625, dtype=np.uint32)
        st[:624] = key
        st[624] = pos
        out = np.empty(int(n), dtype=np.int64)
        N.check(N.lib().rlppo_mt19937_permutation(st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), int(n),
                                                  ctypes.c_void_p(out.ctypes.data)))
        self.rng.set_state((kind, st[:624].copy(), int(st[624]), has_gauss, cached))
        return out


def set_inference_precision(mode):
    """Precision of the rollout forward passes (get_action / value predictions): "fp32" (default: the reference's
    arithmetic, bit-exact action indices) or "bf16" (activations and master weights rounded to bf16 as MFMA operands,
    fp32 accumulation; BASELINE configs[4]).  The PPO update always runs in fp32."""
    from . import _native as N
    N.check(N.lib().rlppo_set_inference_precision({"fp32": 0, "bf16": 1}[mode]))
