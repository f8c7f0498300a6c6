from .running_stats import WelfordRunningStat
from .metrics_logger import MetricsLogger
from .rlgym_v2_gym_wrapper import RLGymV2GymWrapper
from .kbhit import KBHit
