from .running_stats import WelfordRunningStat
from .metrics_logger import MetricsLogger
from .kbhit import KBHit
