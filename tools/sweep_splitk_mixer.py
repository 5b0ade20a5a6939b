"""Split-K sweep of the fp32 GEMM shapes of BASELINE config #3 (MLP-Mixer, B = 128): which slice count gives the most
TFLOP/s per shape, against what `models/common.auto_splitk` picks.  usage: python tools/sweep_splitk_mixer.py [bf16]"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd import ops
from meanflow_audio_codec_amd.models.common import auto_splitk

T = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
dev = "cuda"


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


# (M, N, K, trans_a, trans_b): token mixing (fwd, dX, dW), input / output projections, encoder
SHAPES = [(3072, 2048, 1024, 0, 0), (3072, 1024, 2048, 0, 0), (2048, 1024, 2048, 0, 1), (2048, 2048, 1024, 0, 1),
          (1024, 2048, 2048, 1, 0), (2048, 1024, 2048, 1, 0), (192, 16384, 1024, 0, 0), (192, 1024, 16384, 0, 0),
          (128, 1024, 16384, 0, 1), (128, 16384, 1024, 0, 1), (1024, 16384, 128, 1, 0), (16384, 1024, 128, 1, 0),
          (32768, 544, 2048, 0, 0), (69632, 2048, 256, 0, 0), (69632, 256, 2048, 0, 0), (544, 2048, 32768, 1, 0), (256, 2048, 69632, 1, 0)]
for M, N, K, ta, tb in SHAPES:
    A = (torch.randn((K, M) if ta else (M, K), device=dev) * 0.05).to(T)
    B = (torch.randn((N, K) if tb else (K, N), device=dev) * 0.05).to(T)
    C = torch.empty(M, N, device=dev, dtype=T)
    auto = auto_splitk(M, N, K)
    row = []
    for sk in (1, 2, 3, 4, 6, 8, 16, 32):
        if sk > 1 and K // sk < 64:
            continue
        ms = t(lambda: ops.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=C, splitk=sk))
        row.append(f"{sk}{'*' if sk == auto else ''}: {2.0 * M * N * K / ms / 1e9:.0f}")
    print(f"M={M} N={N} K={K} ta={ta} tb={tb} (auto {auto})  TFLOP/s by split-K  " + "  ".join(row), flush=True)
    del A, B, C
