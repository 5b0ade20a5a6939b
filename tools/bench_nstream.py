"""N-streaming GEMM as the ConvFlow block launches it: [R,128] @ [128,S] with bias on the primal rows and the fused first
LayerNorm (+ its tangent for the rows behind them).  usage: python tools/bench_nstream.py"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd import ops

S, dev = 6270016, "cuda"
W = (torch.randn(128, S, device=dev) * 0.05).bfloat16()
b = torch.zeros(S, device=dev)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
out = []
for M, R, tan in ((192, 128, True), (128, 128, False), (64, 64, False), (256, 128, True)):
    X = (torch.randn(M, 128, device=dev) * 0.3).bfloat16()
    rho = torch.zeros(R, S // 16, device=dev)
    C = torch.empty(M, S, device=dev, dtype=torch.bfloat16)
    plain = t(lambda: ops.gemm(X, W, bias=b, bias_rows=R, out=C))
    ln = t(lambda: ops.gemm(X, W, bias=b, bias_rows=R, ln_rstd=rho, ln_tangent=tan, out=C))
    out.append(f"M={M}: plain {plain:.3f} ln{'+tan' if tan else ''} {ln:.3f}")
    del X, rho, C
print(" | ".join(out))
# the NT form (dX against a [S, 128] kernel): [R, 128] @ [S, 128]^T
Wt = (torch.randn(S, 128, device=dev) * 0.05).bfloat16()
out = []
for M in (128, 64, 192):
    X = (torch.randn(M, 128, device=dev) * 0.3).bfloat16()
    C = torch.empty(M, S, device=dev, dtype=torch.bfloat16)
    out.append(f"NT M={M}: {t(lambda: ops.gemm(X, Wt, trans_b=True, out=C)):.3f}")
    del X, C
print(" | ".join(out))
