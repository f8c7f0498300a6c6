"""Shared construction of the MLP bodies.

Layer order, nn.Linear default initialisation and CPU-generator consumption are those of the reference's
constructors (rlgym_ppo/ppo/discrete_policy.py:21-31, continuous_policy.py:29-41,
multi_discrete_policy.py:22-32, value_estimator.py:18-28): layers are built on the CPU first (so a seeded run
starts from the reference's weights bit for bit) and only then moved to the GPU arena.
"""
import torch.nn as nn

from ..engine import NetArena, linears_of, require_gpu


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
