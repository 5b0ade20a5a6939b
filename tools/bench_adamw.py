"""Micro-benchmark of mfc_adamw on one literal-config weight (128 x 6270016, bf16 gradient). usage: python tools/bench_adamw.py"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd import ops

n = 128 * 6270016
dev = "cuda"
p = torch.randn(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
g = (torch.randn(n, device=dev) * 1e-2).to(torch.bfloat16); pw = torch.empty(n, device=dev, dtype=torch.bfloat16)
for it in range(2):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(3):
        ops.adamw(p, g, m, v, lr=1e-4, step=k + 1, wd=1e-4, p_bf16=pw)
    e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 3
print(f"adamw n={n}: {ms:.3f} ms  {n * 28 / ms / 1e9:.2f} TB/s")
