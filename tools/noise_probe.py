"""Run-to-run noise of the literal-size gradient (same batch twice) measured on +-1 linear functionals."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from meanflow_audio_codec_amd.models import ConditionalConvFlow, TrainState, adamw
from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, PRNGKey
D, CD, LAT, NB = 392704, 128, 256, 8
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] != "f32") else torch.float32
if dt == torch.float32:
    NB = 2
model = ConditionalConvFlow(D, CD, NB, LAT, dtype=dt)
params = model.init(seed=1, device="cuda")
for k, p in params.items():
    if k.endswith("layer_scale_gamma"): p.fill_(0.3)
    elif k.endswith("GlobalResponseNormalization_0/gamma"): p.fill_(0.1)
state = TrainState.create(apply_fn=model.apply, params=params, tx=adamw(1e-4, 1e-4), model=model)
g = torch.Generator(device="cuda").manual_seed(5)
B = 4
x = 0.1 * torch.randn(B, D, generator=g, device="cuda"); e = torch.randn(B, D, generator=g, device="cuda")
t = torch.tensor([[0.9], [0.6], [0.5], [0.3]], device="cuda"); r = torch.tensor([[0.4], [0.1], [0.5], [0.3]], device="cuda")
strat = ImprovedMeanFlowLoss()
big = [k for k, p in state.params.items() if p.numel() > (1 << 24)]
probes = {}
def probe_of(p):
    if p.numel() not in probes:
        probes[p.numel()] = torch.empty(p.numel(), dtype=torch.bfloat16, device="cuda").bernoulli_(0.5, generator=g).mul_(2).sub_(1)
    return probes[p.numel()]
def fp(grads):
    return {k: ((grads[k].reshape(-1).float() * probe_of(grads[k]).float()).sum().double().item(), grads[k].float().norm().item()) for k in big}
runs = []
for i in range(3):
    aux = {}
    loss, grads = strat.compute_loss(state, PRNGKey(0), x, e=e, t=t, r=r, aux=aux)
    runs.append((fp(grads), aux["u"].float().clone(), None if aux["dudt"] is None else aux["dudt"].float().clone()))
for i in (1, 2):
    worst = max(abs(runs[i][0][k][0] - runs[0][0][k][0]) / runs[0][0][k][1] for k in big if runs[0][0][k][1] > 0)
    du = (runs[i][1] - runs[0][1]).norm().item() / runs[0][1].norm().item()
    dd = (runs[i][2] - runs[0][2]).norm().item() / runs[0][2].norm().item()
    print(f"run {i} vs 0: worst functional deviation {worst:.3e} |g|; u rel diff {du:.3e}; dudt rel diff {dd:.3e}")
