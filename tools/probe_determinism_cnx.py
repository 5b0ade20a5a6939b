"""Bitwise repeatability of the ConvNeXt-interior kernels at the literal spatial size and of the tiled K = S / weight-gradient
products, launch family by launch family (same inputs, many repeats).  usage: python tools/probe_determinism_cnx.py [reps] [R]"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd import ops
from meanflow_audio_codec_amd.models.common import dense, dense_dx, dense_dw

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
s, dev, dtype = 626, "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
h0 = torch.randn(R, s, s, 16, device=dev, generator=g).to(dtype)
h0d = torch.randn(R, s, s, 16, device=dev, generator=g).to(dtype)
dout = torch.randn(R, s, s, 16, device=dev, generator=g).to(dtype)
sc = 0.1 * torch.randn(R, 16, device=dev, generator=g); sh = 0.1 * torch.randn(R, 16, device=dev, generator=g)
w = {"conv_w": (torch.randn(3, 3, 16, 16, device=dev, generator=g) / 12).to(dtype), "conv_b": torch.zeros(16, device=dev),
     "exp_w": (torch.randn(16, 32, device=dev, generator=g) / 4).to(dtype), "exp_b": torch.zeros(32, device=dev),
     "grn_gamma": torch.full((32,), 0.1, device=dev), "grn_beta": torch.zeros(32, device=dev),
     "con_w": (torch.randn(32, 16, device=dev, generator=g) / 5.6).to(dtype), "con_b": torch.zeros(16, device=dev),
     "ls": torch.full((16,), 0.5, device=dev)}
h0, rho = ops.ln16(h0)
n1 = torch.empty_like(h0); rho1 = torch.empty(R, s, s, dtype=torch.float32, device=dev); n1d = torch.empty_like(h0)
total = 0


def check(name, fn):
    global total
    first = [t.clone() for t in fn() if t is not None]
    bad = 0
    for _ in range(reps):
        out = [t for t in fn() if t is not None]
        if not all(torch.equal(a, b) for a, b in zip(first, out)):
            bad += 1
    print(f"{name}: {bad} of {reps} repeats differ", flush=True)
    total += bad


def fwd(jvp, keep):
    kw = dict(h0dot=h0d, scaledot=sc, shiftdot=sh) if jvp else {}
    if keep:
        kw["keep"] = (n1, rho1, n1d) if jvp else (n1, rho1)
    o, od, G, q = ops.cnx_forward(h0, sc, sh, w, s, **kw)
    return (o, od, G, q) + ((n1, rho1) if keep else ())


def bwd(from_n1):
    grads = {k: torch.zeros(v.shape, dtype=torch.float32, device=dev) for k, v in w.items()}
    o, _, G, q = ops.cnx_forward(h0, sc, sh, w, s, keep=(n1, rho1))
    kw = dict(n1=n1, rho1=rho1) if from_n1 else {}
    dh0, dsc, dsh = ops.cnx_backward(h0, sc, sh, w, s, G, q, dout, grads, rho0=rho, **kw)
    return (dh0, dsc, dsh) + tuple(grads[k] for k in sorted(grads))


check(f"cnx forward R={R}", lambda: fwd(False, False))
check(f"cnx forward + tangent R={R}", lambda: fwd(True, False))
check(f"cnx forward keep n1 R={R}", lambda: fwd(False, True))
check(f"cnx forward + tangent keep n1 R={R}", lambda: fwd(True, True))
check(f"cnx backward R={R}", lambda: bwd(False))
check(f"cnx backward from n1 R={R}", lambda: bwd(True))
del h0d, dout, n1, n1d

S = 16 * s * s
for M in (R, R + R // 2):
    X = (torch.randn(M, S, device=dev, generator=g) * 0.3).bfloat16()
    Wd = (torch.randn(S, 128, device=dev, generator=g) * 0.01).bfloat16()
    Wu = (torch.randn(128, S, device=dev, generator=g) * 0.05).bfloat16()
    b = torch.randn(128, device=dev, generator=g) * 0.1
    dy = (torch.randn(M, 128, device=dev, generator=g) * 0.3).bfloat16()
    check(f"K = S product M={M}", lambda: (dense(X, Wd, b, bias_rows=R),))
    check(f"K = S NT product (dg1) M={M}", lambda: (dense_dx(X, Wu),))
    check(f"weight gradient [S,128] M={M}", lambda: (dense_dw(X, dy),))
    check(f"weight gradient [128,S] M={M}", lambda: (dense_dw(dy, X),))
    del X, Wd, Wu, dy
print("TOTAL differing repeats:", total)
sys.exit(1 if total else 0)
