"""meanflow_audio_codec_amd -- MI355X-native hot path of gabrieldernbach/meanflow_audio_codec.

MDCT tokenizer, ConvNeXt / MLP velocity nets, flow-matching / MeanFlow / improved-MeanFlow loss steps
(forward, forward-mode tangent, reverse pass), AdamW and the low-NFE sampler as hand-written HIP
kernels for gfx950 behind a C ABI (``include/mfc.h``); this package is the host-side mirror of the
reference's Python interfaces for that path.  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"
