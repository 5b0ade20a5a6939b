"""Split-K sweep of the K = D and K = S products of the literal config (which slab count moves the least bytes per
unit time).  usage: python tools/sweep_splitk.py"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd import ops

D, S, T, dev = 392704, 6270016, torch.bfloat16, "cuda"
def rnd(*shape): return (torch.randn(*shape, device=dev) * 0.05).to(T)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for K, name in ((D, "D"), (S, "S")):
    W = rnd(K, 128); Wt = rnd(128, K)
    for M in (64, 128, 192):
        X = rnd(M, K)
        row = []
        for sk in ((64, 128, 192, 256, 384, 512, 768, 1024) if K == D else (256, 512, 768, 1024, 1536, 2048)):
            a = t(lambda: ops.gemm(X, W, splitk=sk))                      # X[M,K] @ W[K,128]
            b = t(lambda: ops.gemm(X, Wt, trans_b=True, splitk=sk))       # X[M,K] @ Wt[128,K]^T
            row.append(f"{sk}: {a:.3f}/{b:.3f}")
        print(f"K={name} M={M}  NN/NT ms  " + "  ".join(row), flush=True)
    del W, Wt, X
