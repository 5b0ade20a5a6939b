from .mdct import MDCTConfig, imdct, mdct  # noqa: F401
