"""Micro-benchmark of the ConvNeXt-interior kernels at the literal spatial size (s=626).
usage: python tools/bench_cnx.py [R] [dtype]"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd import _lib, ops

R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dtype = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.bfloat16
s = 626
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
h0 = torch.randn(R, s, s, 16, device=dev, generator=g).to(dtype)
h0d = torch.randn(R, s, s, 16, device=dev, generator=g).to(dtype)
dout = torch.randn(R, s, s, 16, device=dev, generator=g).to(dtype)
sc = 0.1 * torch.randn(R, 16, device=dev, generator=g); sh = 0.1 * torch.randn(R, 16, device=dev, generator=g)
w = {"conv_w": (torch.randn(3, 3, 16, 16, device=dev, generator=g) / 12).to(dtype), "conv_b": torch.zeros(16, device=dev),
     "exp_w": (torch.randn(16, 32, device=dev, generator=g) / 4).to(dtype), "exp_b": torch.zeros(32, device=dev),
     "grn_gamma": torch.zeros(32, device=dev), "grn_beta": torch.zeros(32, device=dev),
     "con_w": (torch.randn(32, 16, device=dev, generator=g) / 5.6).to(dtype), "con_b": torch.zeros(16, device=dev),
     "ls": torch.full((16,), 0.5, device=dev)}
grads = {k: torch.zeros(v.shape, dtype=torch.float32, device=dev) for k, v in w.items()}
h0, rho = ops.ln16(h0)
o, _, G, q = ops.cnx_forward(h0, sc, sh, w, s)
NREP = int(sys.argv[3]) if len(sys.argv) > 3 else 8
n1 = torch.empty_like(h0)
rho1 = torch.empty(R, s, s, dtype=torch.float32, device=dev)
best = {}
order = []
for rep in range(NREP):
    _lib.enable_timing()
    ops.cnx_forward(h0, sc, sh, w, s)                                                   # h1-based kernels
    ops.cnx_forward(h0, sc, sh, w, s, h0dot=h0d, scaledot=sc, shiftdot=sh)
    ops.cnx_backward(h0, sc, sh, w, s, G, q, dout, grads, rho0=rho)
    ops.cnx_forward(h0, sc, sh, w, s, keep=(n1, rho1))                                  # statistics keep n1; the rest starts from it
    ops.cnx_forward(h0, sc, sh, w, s, h0dot=h0d, scaledot=sc, shiftdot=sh, keep=(n1, rho1))
    ops.cnx_backward(h0, sc, sh, w, s, G, q, dout, grads, rho0=rho, n1=n1, rho1=rho1)
    torch.cuda.synchronize()
    seen = {}
    for name, ints, nn, a, b in _lib.disable_timing():
        if not name.startswith("mfc_cnx"):
            continue
        jvp = nn[1] if name in ("mfc_cnx_stats", "mfc_cnx_apply", "mfc_cnx_stats_save") else False
        k0 = (name, int(jvp))
        seen[k0] = seen.get(k0, 0) + 1
        k = k0 + (seen[k0],)                   # the old-path bwd_conv and the n1-path bwd_conv are the same kernel: #1, #2
        if k not in best:
            order.append(k)
            best[k] = []
        best[k].append(a.elapsed_time(b))
px = R * s * s
for k in order:
    v = sorted(best[k])
    print(f"{k[0]:22s} jvp={k[1]} #{k[2]} min {v[0]:7.3f} ms  median {v[len(v) // 2]:7.3f} ms  {v[0] * 1e6 / px:6.2f} ns/pixel")
