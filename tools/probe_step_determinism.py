"""Bitwise repeatability of the whole iMF loss + reverse pass at the literal size (the property
tests/test_literal_size_gpu.py::test_literal_shape_shards_sum_to_global_batch asserts once), many times in one process:
per repeat, which leaves' gradients differ from the first evaluation.   usage: python tools/probe_step_determinism.py [reps]"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd.models import ConditionalConvFlow, TrainState, adamw
from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, PRNGKey

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
D, CD, LAT, NB = 392704, 128, 256, 8
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


def digest(grads):
    # exact: a 64-bit sum of the raw bit patterns of every leaf
    out = {}
    for k, v in grads.items():
        raw = v.reshape(-1).view(torch.int16 if v.dtype == torch.bfloat16 else torch.int32)
        out[k] = int(raw.to(torch.int64).sum().item()) ^ int((raw.to(torch.int64) * 3 + 1)[::7].sum().item())
    return out


def poison(bits, gib=48):
    """fill `gib` GiB of the caching allocator's free pool with a 16-bit pattern and hand it back: the workspaces allocated
    next are carved from these blocks, so any dependence on never-written scratch shows up as a changed result"""
    chunks = []
    try:
        for _ in range(gib // 8):
            c = torch.empty(8 * 2 ** 30 // 2, dtype=torch.int16, device="cuda")
            c.fill_(bits)
            chunks.append(c)
    except RuntimeError:
        pass
    torch.cuda.synchronize()
    del chunks


poison(-1)                                   # 0xFFFF: NaN as bf16, NaN pairs as fp32
loss0, grads = strat.compute_loss(state, PRNGKey(0), x, e=e, t=t, r=r)
d0 = digest(grads)
small0 = {k: v.clone() for k, v in grads.items() if v.numel() < (1 << 24)}
state._grads = None
model.release_workspace()
del grads
poison(0x3F80)                               # 1.0 as bf16
loss1, grads = strat.compute_loss(state, PRNGKey(0), x, e=e, t=t, r=r)
d1 = digest(grads)
diff = sorted(k for k in d0 if d1[k] != d0[k])
print(f"fresh workspaces over 0xFFFF vs over 0x3F80 scratch: loss equal {loss1.item() == loss0.item()}; {len(diff)} leaves differ: {diff[:16]}", flush=True)
poisoned_bad = 1 if (diff or loss1.item() != loss0.item()) else 0
d0 = d1
loss0 = loss1
bad = 0
for i in range(reps):
    loss, grads = strat.compute_loss(state, PRNGKey(0), x, e=e, t=t, r=r)
    d = digest(grads)
    diff = sorted(k for k in d0 if d[k] != d0[k])
    if diff or loss.item() != loss0.item():
        bad += 1
        print(f"repeat {i}: loss equal {loss.item() == loss0.item()}; {len(diff)} leaves differ: {diff[:12]}", flush=True)
print(f"{bad} of {reps} repeats differ from the first evaluation", flush=True)
sys.exit(1 if (bad or poisoned_bad) else 0)
