from .audio import (audio_to_frames, batch, buffer_shuffle, build_audio_pipeline, design_lowpass,  # noqa: F401
                    glob_audio_files, load_audio, load_audio_files, resample)
from .mnist import load_mnist  # noqa: F401

__all__ = ["build_audio_pipeline", "load_audio", "load_audio_files", "audio_to_frames", "glob_audio_files", "batch",
           "buffer_shuffle", "load_mnist", "resample", "design_lowpass"]
