from .sampling import GraphedDecoder, heun_integrate, one_step_decode, sample  # noqa: F401
from .audio_metrics import spectral_distance  # noqa: F401,E402
