"""Localise a non-repeatable evaluation of the literal-size iMF loss + reverse pass: every `ops.*` call is followed by an exact
digest of every tensor it was given or returned; evaluations are compared call by call against the first one and the first
call whose digest differs is printed.  (The digests synchronise after every call: if no evaluation differs in this mode but
tools/probe_step_determinism.py sees differences, the cause needs two launches in flight.)
usage: python tools/probe_step_trace.py [reps]"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd import ops
from meanflow_audio_codec_amd.models import ConditionalConvFlow, TrainState, adamw
from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, PRNGKey

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
D, CD, LAT, NB = 392704, 128, 256, 8
trace = []


def tensors_of(x, out):
    if isinstance(x, torch.Tensor):
        out.append(x)
    elif isinstance(x, (tuple, list)):
        for y in x:
            tensors_of(y, out)
    elif isinstance(x, dict):
        for y in x.values():
            tensors_of(y, out)


def dig(t):
    if not t.is_cuda or t.numel() == 0:
        return 0
    c = t if t.is_contiguous() else t.contiguous()
    v = c.reshape(-1).view(torch.uint8) if c.element_size() == 1 else c.reshape(-1).view({2: torch.int16, 4: torch.int32, 8: torch.int64}[c.element_size()])
    return int(v.sum(dtype=torch.int64).item())


def wrap(name, fn):
    def inner(*a, **k):
        r = fn(*a, **k)
        ts = []
        tensors_of(a, ts); tensors_of(k, ts); tensors_of(r, ts)
        trace.append((name, tuple(tuple(t.shape) for t in ts), tuple(dig(t) for t in ts)))
        return r
    return inner


for n in dir(ops):
    f = getattr(ops, n)
    if callable(f) and not n.startswith("_") and getattr(f, "__module__", "") == ops.__name__:
        setattr(ops, n, wrap(n, f))

model = ConditionalConvFlow(D, CD, NB, LAT, dtype=torch.bfloat16)
params = model.init(seed=1, device="cuda")
for k, p in params.items():
    if k.endswith("layer_scale_gamma"):
        p.fill_(0.3)
    elif k.endswith("GlobalResponseNormalization_0/gamma"):
        p.fill_(0.1)
state = TrainState.create(apply_fn=model.apply, params=params, tx=adamw(1e-4, 1e-4), model=model)
g = torch.Generator(device="cuda").manual_seed(5)
B = 4
x = 0.1 * torch.randn(B, D, generator=g, device="cuda")
e = torch.randn(B, D, generator=g, device="cuda")
t = torch.tensor([[0.9], [0.6], [0.5], [0.3]], device="cuda")
r = torch.tensor([[0.4], [0.1], [0.5], [0.3]], device="cuda")
strat = ImprovedMeanFlowLoss()

strat.compute_loss(state, PRNGKey(0), x, e=e, t=t, r=r)     # allocates every workspace
trace.clear()
strat.compute_loss(state, PRNGKey(0), x, e=e, t=t, r=r)
ref = list(trace)
print(f"{len(ref)} ops calls per evaluation", flush=True)
bad = 0
for i in range(reps):
    trace.clear()
    strat.compute_loss(state, PRNGKey(0), x, e=e, t=t, r=r)
    if len(trace) != len(ref):
        print(f"repeat {i}: {len(trace)} calls"); bad += 1; continue
    for j, (a, b) in enumerate(zip(ref, trace)):
        if a != b:
            bad += 1
            which = [n for n, (p, q) in enumerate(zip(a[2], b[2])) if p != q]
            print(f"repeat {i}: first differing call #{j} of {len(ref)}: {a[0]} shapes {a[1]} -- tensors {which} differ; previous call {ref[j - 1][0]} {ref[j - 1][1]}", flush=True)
            break
print(f"{bad} of {reps} evaluations differ from the reference evaluation", flush=True)
