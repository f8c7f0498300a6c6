"""MetricsLogger -- user-subclassable env-metrics (de)serialiser; wire format of rlgym_ppo/util/metrics_logger.py:
each array is flattened as [ndim, *shape, *values] into one float32 vector (out of the hot path)."""
from abc import ABC

import numpy as np


class MetricsLogger(ABC):
    def collect_metrics(self, game_state) -> np.ndarray:
        flat = []
        for arr in self._collect_metrics(game_state):
            shape = np.shape(arr)
            flat.append(len(shape))
            flat.extend(shape)
            flat.extend(np.ravel(arr).tolist())
        return np.asarray(flat).astype(np.float32)

    def report_metrics(self, collected_metrics, wandb_run, cumulative_timesteps):
        if wandb_run is None:
            return
        reports = []
        for ser in collected_metrics:
            arrays, i = [], 0
            while i < len(ser):
                ndim = int(ser[i])
                shape = [int(s) for s in ser[i + 1:i + 1 + ndim]]
                count = int(np.prod(shape)) if ndim else 1
                arrays.append(ser[i + 1 + ndim:i + 1 + ndim + count])
                i += 1 + ndim + count
            reports.append(arrays)
        self._report_metrics(reports, wandb_run, cumulative_timesteps)

    def _collect_metrics(self, game_state):
        raise NotImplementedError

    def _report_metrics(self, collected_metrics, wandb_run, cumulative_timesteps):
        raise NotImplementedError
