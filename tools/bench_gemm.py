"""Micro-benchmark of the ConvFlow GEMM shapes (literal config). usage: python tools/bench_gemm.py [R]"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd import _lib, ops
from meanflow_audio_codec_amd.models.common import dense, dense_dw, dense_dx

R = int(sys.argv[1]) if len(sys.argv) > 1 else 192
D, S, T = 392704, 6270016, torch.bfloat16
dev = "cuda"
def rnd(*shape): return (torch.randn(*shape, device=dev) * 0.05).to(T)
X, W1, W2, W3, W4 = rnd(R, D), rnd(D, 128), rnd(128, S), rnd(S, 128), rnd(128, D)
A1, H0 = rnd(R, 128), rnd(R, S)
b128, bS, bD = torch.zeros(128, device=dev), torch.zeros(S, device=dev), torch.zeros(D, device=dev)
cases = [
    ("fwd1  X[R,D]@W1[D,128]      ", lambda: dense(X, W1, b128)),
    ("fwd2  A[R,128]@W2[128,S]    ", lambda: dense(A1, W2, bS)),
    ("fwd3  O[R,S]@W3[S,128]      ", lambda: dense(H0, W3, b128)),
    ("fwd4  A[R,128]@W4[128,D]+res", lambda: dense(A1, W4, bD, residual=X, alpha=0.125)),
    ("dx4   dX[R,D]@W4^T          ", lambda: dense_dx(X, W4)),
    ("dw4   A^T[128,R]@dX[R,D]    ", lambda: dense_dw(A1, X)),
    ("dx3   dA[R,128]@W3^T -> dO  ", lambda: dense_dx(A1, W3)),
    ("dw3   O^T[S,R]@dA[R,128]    ", lambda: dense_dw(H0, A1)),
    ("dw2   A^T[128,R]@dH0[R,S]   ", lambda: dense_dw(A1, H0)),
    ("dx2   dH0[R,S]@W2^T         ", lambda: dense_dx(H0, W2)),
    ("dw1   X^T[D,R]@dA[R,128]    ", lambda: dense_dw(X, A1)),
    ("dx1   dA[R,128]@W1^T+res    ", lambda: dense_dx(A1, W1, residual=X)),
]
for name, fn in cases:
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name} {s.elapsed_time(e) / 3:8.3f} ms")
