from .loss_strategies import FlowMatchingLoss, ImprovedMeanFlowLoss, LossStrategy, MeanFlowLoss  # noqa: F401
from .noise_schedules import LinearNoiseSchedule, NoiseSchedule, UniformNoiseSchedule  # noqa: F401
from .time_sampling import (LogitNormalTimeSampling, MeanFlowTimeSampling, PRNGKey,  # noqa: F401
                            TimeSamplingStrategy, UniformTimeSampling)
from .training_steps import train_step  # noqa: F401
from .train import create_flow_model, create_loss_strategy, train_flow  # noqa: F401
