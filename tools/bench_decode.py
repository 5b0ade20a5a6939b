"""1-NFE decode of the literal config on its own (what bench.py reports as decode_audio_s_per_s): hipGraph of
noise -> u(eps, [1, 1]) -> x0 = eps - u -> IMDCT.  usage: python tools/bench_decode.py [batch] [reps]"""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd.evaluators import GraphedDecoder
from meanflow_audio_codec_amd.models import ConditionalConvFlow
from meanflow_audio_codec_amd.preprocessing import MDCTConfig, MDCTTokenization

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
T, sr = 196608, 24000
tok = MDCTTokenization(config=MDCTConfig(window_size=512, hop_size=256))
n_tok, tok_dim = tok.token_shape(T)
model = ConditionalConvFlow(n_tok * tok_dim, 128, 8, 256, dtype=torch.bfloat16)
params = model.init(seed=42, device="cuda")
work = {k: (p.to(model.compute_dtype_of(k)) if model.compute_dtype_of(k) != torch.float32 else p) for k, p in params.items()}
del params
lat = torch.zeros(B, 256, device="cuda")
dec = GraphedDecoder(model, work, B, lat, n_steps=0, token_shape=(n_tok, tok_dim), mdct_config=tok.config, seed=42)
for _ in range(2):
    dec()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    audio = dec()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"decode batch {B}: {dt * 1e3:.2f} ms -> {B * T / sr / dt:.0f} audio-s/s; finite={bool(torch.isfinite(audio).all())}; out {tuple(audio.shape)}")
