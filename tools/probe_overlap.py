"""Do a VALU-bound ConvNeXt kernel and the HBM-bound AdamW stream share the GPU when launched on two streams?
Times (a) N x (cnx stats + apply) on stream A, (b) M x mfc_adamw of a 0.8 B-parameter leaf on stream B, (c) both at once.
usage: python tools/probe_overlap.py [R] [N] [M]   (env MFC_CNX_MAX_BLOCKS to cap the persistent grids)"""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd import ops

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
M = int(sys.argv[3]) if len(sys.argv) > 3 else 2
s, dev = 626, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
h0 = torch.randn(R, s, s, 16, device=dev, generator=g).bfloat16()
sc = 0.1 * torch.randn(R, 16, device=dev, generator=g); sh = 0.1 * torch.randn(R, 16, device=dev, generator=g)
w = {"conv_w": (torch.randn(3, 3, 16, 16, device=dev, generator=g) / 12).bfloat16(), "conv_b": torch.zeros(16, device=dev),
     "exp_w": (torch.randn(16, 32, device=dev, generator=g) / 4).bfloat16(), "exp_b": torch.zeros(32, device=dev),
     "grn_gamma": torch.zeros(32, device=dev), "grn_beta": torch.zeros(32, device=dev),
     "con_w": (torch.randn(32, 16, device=dev, generator=g) / 5.6).bfloat16(), "con_b": torch.zeros(16, device=dev),
     "ls": torch.full((16,), 0.5, device=dev)}
h0, _ = ops.ln16(h0)
n = 128 * 6270016
p = torch.randn(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
gr = (0.01 * torch.randn(n, device=dev)).bfloat16(); pw = p.bfloat16()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def run_a():
    with torch.cuda.stream(sa):
        for _ in range(N):
            ops.cnx_forward(h0, sc, sh, w, s)


def run_b():
    with torch.cuda.stream(sb):
        for i in range(M):
            ops.adamw(p, gr, m, v, lr=1e-4, wd=1e-4, step=i + 1, p_bf16=pw)


def timed(fns):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in fns:
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


run_a(); run_b(); torch.cuda.synchronize()
for rep in range(2):
    ta, tb = timed([run_a]), timed([run_b])
    tab, tba = timed([run_a, run_b]), timed([run_b, run_a])
    print(f"cnx alone {ta:.2f} ms | adamw alone {tb:.2f} ms | sum {ta + tb:.2f} | both (cnx first) {tab:.2f} | both (adamw first) {tba:.2f}")
