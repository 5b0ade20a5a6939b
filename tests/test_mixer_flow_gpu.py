"""GPU parity: ConditionalMLPMixerFlow + MLPMixerEncoder (BASELINE config #3 family) vs the fp64 oracle."""
import pytest
import torch

from oracle import flow_oracle as fo

pytestmark = pytest.mark.gpu

KW = dict(token_mix_dim=48, channel_mix_dim=40, num_channels=16, num_latent_tokens=4, num_context_tokens=8)


def _make(D=64, CD=32, LAT=16, NB=2, dtype=torch.float32, seed=0):
    from meanflow_audio_codec_amd.models import ConditionalMLPMixerFlow, TrainState, adamw
    model = ConditionalMLPMixerFlow(D, CD, NB, LAT, dtype=dtype, **KW)
    shapes = fo.mixer_flow_shapes(D, CD, LAT, NB, C=16, tmd=48, cmd=40, n_lat=4, n_ctx=8)
    p64 = fo.init_params(shapes, seed=seed, special=False)
    flat = {k: v.float().cuda().contiguous() for k, v in fo.flatten(p64).items()}
    assert {k: tuple(v.shape) for k, v in flat.items()} == {k: tuple(v) for k, v in model.param_shapes().items()}
    state = TrainState.create(apply_fn=model.apply, params=flat, tx=adamw(1e-3, 1e-2), model=model)
    pq = fo.unflatten({k: state.work[k].double().cpu() for k in flat})
    return model, state, pq


def _rel(a, b):
    return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 6e-2)])
def test_apply_and_encode(dtype, tol):
    model, state, pq = _make(dtype=dtype)
    g = torch.Generator().manual_seed(1)
    x, time = torch.randn(5, 64, generator=g), torch.rand(5, 2, generator=g)
    xq = x.to(dtype).double()
    lat = model.apply({"params": state.work}, x.cuda(), method="encode")
    lat_ref = fo.mixer_encode(pq, xq)
    assert lat.shape == (5, 4, 16) and _rel(lat, lat_ref) < tol
    out = model.apply({"params": state.work}, x.cuda(), time.cuda(), lat)
    assert _rel(out, fo.mixer_flow_apply(pq, xq, time.double(), lat.double().cpu())) < tol
    out0 = model.apply({"params": state.work}, x.cuda(), time.cuda(), None)
    assert _rel(out0, fo.mixer_flow_apply(pq, xq, time.double(), None)) < tol


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 2e-4, 3e-3), (torch.bfloat16, 5e-2, 0.25)])
def test_mean_flow_and_imf_losses(dtype, tol, gtol):
    """config #3 is method=mean_flow + mlp_mixer + mdct: JVP loss through the mixer and its encoder."""
    from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, MeanFlowLoss, PRNGKey
    model, state, pq = _make(dtype=dtype, seed=3)
    g = torch.Generator().manual_seed(5)
    B = 6
    x, e = torch.randn(B, 64, generator=g), torch.randn(B, 64, generator=g)
    t, r = fo.sample_tr_from_normals(torch.randn(B, 1, generator=g, dtype=torch.float64),
                                     torch.randn(B, 1, generator=g, dtype=torch.float64))
    t, r = t.float(), r.float()
    for strat, ref in ((MeanFlowLoss(), fo.mf_loss), (ImprovedMeanFlowLoss(), fo.imf_loss)):
        loss_ref, g_ref, _ = ref(fo.mixer_flow_apply, fo.mixer_encode, pq, x.double(), e.double(), t.double(), r.double())
        loss, grads = strat.compute_loss(state, PRNGKey(0), x.cuda(), e=e.cuda(), t=t.cuda(), r=r.cuda())
        assert abs(loss.item() - loss_ref.item()) < tol * max(1.0, abs(loss_ref.item())), type(strat).__name__
        gr = fo.flatten(g_ref)
        bad = {k: _rel(grads[k], gr[k]) for k in gr if gr[k].abs().max() > 0 and not _rel(grads[k], gr[k]) < gtol}
        assert not bad, (type(strat).__name__, bad)


def test_mnist_mdct_shape_runs():
    """BASELINE config #3 literal shapes: MNIST 784 -> MDCT(512, 256) -> D = 1024, nt = 1024, C = 16."""
    from meanflow_audio_codec_amd.models import ConditionalMLPMixerFlow, TrainState, adamw
    from meanflow_audio_codec_amd.preprocessing import MDCTTokenization
    from meanflow_audio_codec_amd.trainers import MeanFlowLoss, PRNGKey, train_step
    tok = MDCTTokenization(512, 256)
    assert tok.token_shape(784) == (2, 512)
    model = ConditionalMLPMixerFlow(1024, 128, 2, 256, num_context_tokens=16, dtype=torch.bfloat16)
    state = TrainState.create(apply_fn=model.apply, params=model.init(0), tx=adamw(1e-4, 1e-4), model=model)
    x = torch.rand(8, 784).cuda()
    tokens = tok.tokenize(x).reshape(8, -1)
    state, loss, key = train_step(state, PRNGKey(1), tokens, MeanFlowLoss())
    assert torch.isfinite(loss).item() and state.step == 1


def test_config3_real_dimensions_vs_oracle():
    """BASELINE config #3 at its REAL dimensions (configs/method=mean_flow--architecture=mlp_mixer--dataset=mnist--
    tokenization=mdct.json; reference models/mlp_mixer.py:171-323 defaults): D = 1024 -> nt = 1024 tokens x 16 channels,
    token / channel mixing dims 2048 / 2048, 8 blocks, cond 128, latent 256, 32 latent + 512 context tokens
    (303.8 M parameters), fp32, B = 4: MeanFlow loss (JVP through the mixer and its encoder) and the
    gradients of every leaf against the fp64 oracle."""
    from meanflow_audio_codec_amd.models import ConditionalMLPMixerFlow, TrainState, adamw
    from meanflow_audio_codec_amd.preprocessing import MDCTTokenization
    from meanflow_audio_codec_amd.trainers import MeanFlowLoss, PRNGKey
    D, CD, LAT, NB, B = 1024, 128, 256, 8, 4
    model = ConditionalMLPMixerFlow(D, CD, NB, LAT, dtype=torch.float32)            # every other argument: the defaults
    shapes = fo.mixer_flow_shapes(D, CD, LAT, NB)                                    # oracle defaults = reference defaults
    p64 = fo.init_params(shapes, seed=11, special=False)
    flat = {k: v.float().cuda().contiguous() for k, v in fo.flatten(p64).items()}
    assert {k: tuple(v.shape) for k, v in flat.items()} == {k: tuple(v) for k, v in model.param_shapes().items()}
    n_params = sum(v.numel() for v in flat.values())
    # SURVEY 8(d) quotes 303.8 M for the flow itself; the MLPMixerEncoder wired to `encode` adds D x 512 x 256 + its block
    assert n_params > 303e6
    state = TrainState.create(apply_fn=model.apply, params=flat, tx=adamw(1e-4, 1e-4), model=model)
    pq = fo.unflatten({k: state.work[k].double().cpu() for k in flat})
    g = torch.Generator().manual_seed(21)
    img = torch.rand(B, 784, generator=g)
    x = MDCTTokenization(512, 256).tokenize(img.cuda()).reshape(B, -1)                # the config's own tokens: [B, 2, 512]
    assert x.shape == (B, D)
    e = torch.randn(B, D, generator=g)
    t, r = fo.sample_tr_from_normals(torch.randn(B, 1, generator=g, dtype=torch.float64),
                                     torch.randn(B, 1, generator=g, dtype=torch.float64))
    loss_ref, g_ref, _ = fo.mf_loss(fo.mixer_flow_apply, fo.mixer_encode, pq, x.double().cpu(), e.double(), t, r)
    loss, grads = MeanFlowLoss().compute_loss(state, PRNGKey(0), x, e=e.cuda(), t=t.float().cuda(), r=r.float().cuda())
    assert abs(loss.item() - loss_ref.item()) < 2e-4 * max(1.0, abs(loss_ref.item()))
    gr = fo.flatten(g_ref)
    errs = {k: _rel(grads[k], gr[k]) for k in gr if gr[k].abs().max() > 0}
    bad = {k: v for k, v in errs.items() if not v < 3e-3}
    assert len(errs) > 100 and not bad, bad
