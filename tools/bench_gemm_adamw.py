"""Micro-benchmark: fused weight-gradient + AdamW (mfc_gemm_adamw) vs mfc_gemm -> mfc_adamw on the two big ConvFlow kernels."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd import ops

S, R, dev = 6270016, 128, "cuda"
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.05).bfloat16()
for name, (A, dY) in {"dW3 [S,128]": (rnd(R, S), rnd(R, 128)), "dW2 [128,S]": (rnd(R, 128), rnd(R, S))}.items():
    M, N = A.shape[1], dY.shape[1]
    p = torch.randn(M, N, device=dev); m = torch.zeros_like(p); v = torch.zeros_like(p); pw = p.bfloat16(); g = torch.empty_like(pw)
    def sep():
        ops.gemm(A, dY, trans_a=True, out=g); ops.adamw(p, g, m, v, lr=1e-4, wd=1e-4, step=1, p_bf16=pw)
    def fused():
        ops.gemm_adamw(A, dY, trans_a=True, p=p, m=m, v=v, p_bf16=pw, lr=1e-4, wd=1e-4, step=1)
    for label, fn in (("gemm + adamw", sep), ("gemm_adamw  ", fused)):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): fn()
        e.record(); torch.cuda.synchronize()
        print(f"{name} {label} {s.elapsed_time(e) / 3:7.3f} ms")
    del p, m, v, pw, g
