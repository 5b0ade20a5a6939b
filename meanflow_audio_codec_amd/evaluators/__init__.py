from .sampling import GraphedDecoder, heun_integrate, one_step_decode, sample  # noqa: F401
