"""Micro-benchmark of the fused channel MLP of the Mixer at the config-#3 size (128 + 64 tangent samples x 1024 tokens, H = 2048).
usage: python tools/bench_chanmlp.py [f32|bf16]"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd import ops

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
dev, H, nt, R, ntan = "cuda", 2048, 1024, 128, 64
rows, act = (R + ntan) * nt, R * nt
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn(rows, 16, device=dev, generator=g).to(dt)
res = torch.randn(rows, 16, device=dev, generator=g).to(dt)
W1 = (torch.randn(16, H, device=dev, generator=g) / 4).to(dt)
W2 = (torch.randn(H, 16, device=dev, generator=g) / 45).to(dt)
b1, b2 = torch.zeros(H, device=dev), torch.zeros(16, device=dev)
dy = torch.randn(act, 16, device=dev, generator=g).to(dt)
dW1, dW2, db1 = torch.empty_like(W1), torch.empty_like(W2), torch.empty(H, device=dev)
out = torch.empty_like(a)
da = torch.empty(act, 16, device=dev, dtype=dt)
ap = a[:act].contiguous()


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


peak = 157.3e12 if dt == torch.float32 else 2.5e15
f = t(lambda: ops.chanmlp_fwd(a, W1, b1, W2, b2, act_rows=act, residual=res, out=out))
fp = t(lambda: ops.chanmlp_fwd(ap, W1, b1, W2, b2, residual=res[:act], out=out[:act]))
b = t(lambda: ops.chanmlp_bwd(ap, dy, W1, b1, W2, dW1, db1, dW2, da=da))
ff, fb = 2.0 * rows * 2 * 16 * H, 2.0 * act * 5 * 16 * H
print(f"{dt}: fwd (128 + 64 tangent samples) {f:.3f} ms = {ff / f / 1e9:.1f} TFLOP/s ({ff / (f * 1e-3) / peak:.3f}) | fwd primal only {fp:.3f} ms "
      f"({2.0 * act * 2 * 16 * H / (fp * 1e-3) / peak:.3f}) | bwd {b:.3f} ms = {fb / b / 1e9:.1f} TFLOP/s ({fb / (b * 1e-3) / peak:.3f})")
