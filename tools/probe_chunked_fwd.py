"""Does running the ConvNeXt forward passes (statistics -> finalize -> apply from n1) over row chunks small enough for the
256 MiB Infinity Cache beat one launch per pass over all rows?  (n1 / 1-sigma written by the statistics pass and h1 are
re-read by the apply pass: 64 of its 96 bytes per pixel.)  Every variant is captured in a hipGraph so the host cost of
the extra launches does not enter.  usage: python tools/probe_chunked_fwd.py [R] [jvp]"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd import ops

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
JVP = len(sys.argv) > 2 and sys.argv[2] == "jvp"
dtype, s, dev = torch.bfloat16, 626, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
h0 = torch.randn(R, s, s, 16, device=dev, generator=g).to(dtype)
h0d = torch.randn(R, s, s, 16, device=dev, generator=g).to(dtype) if JVP else None
sc = 0.1 * torch.randn(R, 16, device=dev, generator=g); sh = 0.1 * torch.randn(R, 16, device=dev, generator=g)
w = {"conv_w": (torch.randn(3, 3, 16, 16, device=dev, generator=g) / 12).to(dtype), "conv_b": torch.zeros(16, device=dev),
     "exp_w": (torch.randn(16, 32, device=dev, generator=g) / 4).to(dtype), "exp_b": torch.zeros(32, device=dev),
     "grn_gamma": torch.zeros(32, device=dev), "grn_beta": torch.zeros(32, device=dev),
     "con_w": (torch.randn(32, 16, device=dev, generator=g) / 5.6).to(dtype), "con_b": torch.zeros(16, device=dev),
     "ls": torch.full((16,), 0.5, device=dev)}
h0, _ = ops.ln16(h0)
n1 = torch.empty_like(h0); rho1 = torch.empty(R, s, s, dtype=torch.float32, device=dev)
n1d = torch.empty_like(h0) if JVP else None
o = torch.empty_like(h0); od = torch.empty_like(h0) if JVP else None
ops.cnx_workspace(R, s, dev)


def run(chunk):
    for i in range(0, R, chunk):
        j = min(R, i + chunk)
        keep = (n1[i:j], rho1[i:j]) + ((n1d[i:j],) if JVP else ())
        ops.cnx_forward(h0[i:j], sc[i:j], sh[i:j], w, s, h0dot=h0d[i:j] if JVP else None, scaledot=sc[i:j] if JVP else None,
                        shiftdot=sh[i:j] if JVP else None, out=o[i:j], outdot=od[i:j] if JVP else None, keep=keep)


ref = None
for chunk in (R, 32, 16, 8, 6, 4, 2):
    if chunk > R:
        continue
    run(chunk); torch.cuda.synchronize()
    st = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        run(chunk)
        st.synchronize()
        with torch.cuda.graph(gr, stream=st):
            run(chunk)
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    same = "" if ref is None else f" same bits as one launch: {torch.equal(o, ref)}"
    if ref is None:
        ref = o.clone()
    print(f"R={R} jvp={JVP} chunk {chunk:3d}: min {min(ts):.3f} ms median {sorted(ts)[3]:.3f} ms{same}", flush=True)
