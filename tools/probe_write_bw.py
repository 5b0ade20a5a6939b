"""What does HBM deliver for pure writes, pure reads and a 40 / 60 read / write mix of the N-streaming GEMM's size
(1.6 GB in, 2.4 GB out)?  PyTorch fill / sum / copy kernels on contiguous tensors: an upper bound for the write side of
`gemm_nstream_kernel` (M = 192: 4.0 GB in 0.96 ms = 4.2 TB/s).  usage: python tools/probe_write_bw.py"""
import torch

dev = "cuda"
S = 6270016


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for rows in (64, 128, 192, 256):
    C = torch.empty(rows, S, device=dev, dtype=torch.bfloat16)
    W = torch.empty(128, S, device=dev, dtype=torch.bfloat16).normal_()
    gb = C.numel() * 2 / 1e9
    ms_fill = t(lambda: C.fill_(1.0))
    ms_copy = t(lambda: C[:128 if rows >= 128 else rows].copy_(W[:128 if rows >= 128 else rows]))
    ms_read = t(lambda: W.sum())
    n = min(rows, 128)
    print(f"rows={rows}: fill {gb:.2f} GB in {ms_fill:.3f} ms = {gb / ms_fill:.2f} TB/s | copy {n} rows (read+write {2 * n * S * 2 / 1e9:.2f} GB) "
          f"{ms_copy:.3f} ms = {2 * n * S * 2 / 1e9 / ms_copy:.2f} TB/s | read 1.6 GB (sum) {ms_read:.3f} ms = {1.605 / ms_read:.2f} TB/s", flush=True)
    del C, W
