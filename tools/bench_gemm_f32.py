"""fp32 tiled GEMM on three Mixer shapes (whole K, one slice): TFLOP/s.  usage: python tools/bench_gemm_f32.py"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd import ops
dev = "cuda"
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
out = []
for M, N, K in ((69632, 2048, 256), (32768, 512, 2048), (4096, 4096, 4096)):
    A = torch.randn(M, K, device=dev) * 0.05
    B = torch.randn(K, N, device=dev) * 0.05
    C = torch.empty(M, N, device=dev)
    ms = t(lambda: ops.gemm(A, B, out=C))
    out.append(f"{M}x{N}x{K}: {2.0 * M * N * K / ms / 1e9:.0f}")
    del A, B, C
print(" | ".join(out))
