"""ValueEstimator -- drop-in for rlgym_ppo/ppo/value_estimator.py:13-36 on librlppo's fused MLP forward."""
import torch

from ._mlp import ArenaModule, build_body


class ValueEstimator(ArenaModule):
    def __init__(self, input_shape, layer_sizes, device):
        super().__init__()
        self.model = build_body(input_shape, layer_sizes, 1)
        self._finish(device)

    @torch.no_grad()
    def forward(self, x):
        """numpy/tensor [n, d] of any float dtype -> device tensor [n, 1] (value_estimator.py:30-36)."""
        rows = self.arena.stage_obs(x)
        return self.arena.forward(rows)[:, :1]

    def forward_padded(self, rows):
        """[n, ld] already-padded device rows -> contiguous [n] values (used by Learner.add_new_experience)."""
        return self.arena.forward(rows)[:, 0]
