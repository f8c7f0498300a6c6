"""rlgym_ppo_amd -- MI355X-native hot path of rlgym-ppo (rollout inference -> GAE -> PPO update) behind the
reference's Python API:  `from rlgym_ppo_amd import Learner` mirrors `from rlgym_ppo import Learner`
(reference: rlgym_ppo/__init__.py:1).  Sub-packages mirror the reference's: .ppo, .util, .batched_agents."""

__version__ = "0.1.0"


def __getattr__(name):  # lazy: importing the package must not spawn anything or touch the GPU
    if name == "Learner":
        from .learner import Learner
        return Learner
    raise AttributeError(name)
