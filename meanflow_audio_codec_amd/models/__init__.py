from .conv_flow import ConditionalConvFlow  # noqa: F401
from .mlp_flow import ConditionalFlow  # noqa: F401
from .mlp_mixer import ConditionalMLPMixerFlow  # noqa: F401
from .train_state import AdamW, TrainState, adamw  # noqa: F401
