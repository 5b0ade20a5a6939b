"""Bitwise repeatability of the N-streaming GEMM launches of the ConvFlow block (same inputs, same launch, many times):
usage: python tools/probe_determinism.py [reps]"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
S, D, dev = 6270016, 392704, "cuda"
g = torch.Generator(device=dev).manual_seed(1)
bad = 0


def check(name, fn):
    global bad
    first = [t.clone() for t in fn()]
    n_bad = 0
    for _ in range(reps):
        out = fn()
        if not all(torch.equal(a, b) for a, b in zip(first, out)):
            n_bad += 1
    print(f"{name}: {n_bad} of {reps} repeats differ", flush=True)
    bad += n_bad


for N in (S, D):
    W = (torch.randn(128, N, device=dev, generator=g) * 0.05).bfloat16()
    Wt = (torch.randn(N, 128, device=dev, generator=g) * 0.05).bfloat16()
    b = torch.randn(N, device=dev, generator=g) * 0.1
    for M, R, tan in ((6, 4, True), (64, 64, False), (128, 128, False), (192, 128, True)):
        X = (torch.randn(M, 128, device=dev, generator=g) * 0.3).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        rho = torch.zeros(R, N // 16, device=dev)
        Rs = (torch.randn(M, N, device=dev, generator=g) * 0.1).bfloat16()
        check(f"NN N={N} M={M} bias", lambda: (ops.gemm(X, W, bias=b, bias_rows=R, out=C),))
        check(f"NN N={N} M={M} ln{'+tan' if tan else ''}", lambda: (ops.gemm(X, W, bias=b, bias_rows=R, ln_rstd=rho, ln_tangent=tan, out=C), rho))
        check(f"NN N={N} M={M} alpha+residual", lambda: (ops.gemm(X, W, bias=b, bias_rows=R, alpha=0.125, residual=Rs, beta=1.0, out=C),))
        check(f"NT N={N} M={M}", lambda: (ops.gemm(X, Wt, trans_b=True, out=C),))
        check(f"NT N={N} M={M} residual", lambda: (ops.gemm(X, Wt, trans_b=True, residual=Rs, beta=1.0, out=C),))
        del X, C, rho, Rs
    del W, Wt, b
print("TOTAL differing repeats:", bad)
sys.exit(1 if bad else 0)
