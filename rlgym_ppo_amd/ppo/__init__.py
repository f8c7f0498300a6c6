from .continuous_policy import ContinuousPolicy
from .multi_discrete_policy import MultiDiscreteFF
from .discrete_policy import DiscreteFF
from .value_estimator import ValueEstimator
from .ppo_learner import PPOLearner
from .experience_buffer import ExperienceBuffer
