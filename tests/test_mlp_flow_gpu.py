"""GPU parity: MLP ConditionalFlow (BASELINE config #2 family) vs the fp64 oracle, incl. the reference's
two iMF property tests (test/test_improved_mean_flow.py) on the very model they use."""
import pytest
import torch

from oracle import flow_oracle as fo

pytestmark = pytest.mark.gpu


def _make(D, CD, LAT, NB, dtype=torch.float32, seed=0):
    from meanflow_audio_codec_amd.models import ConditionalFlow, TrainState, adamw
    model = ConditionalFlow(D, CD, NB, LAT, dtype=dtype)
    p64 = fo.init_params(fo.mlp_flow_shapes(D, CD, LAT, NB), seed=seed, special=False)
    flat = {k: v.float().cuda().contiguous() for k, v in fo.flatten(p64).items()}
    assert {k: tuple(v.shape) for k, v in flat.items()} == {k: tuple(v) for k, v in model.param_shapes().items()}
    state = TrainState.create(apply_fn=model.apply, params=flat, tx=adamw(1e-3, 1e-2), model=model)
    pq = fo.unflatten({k: state.work[k].double().cpu() for k in flat})
    return model, state, pq


def _rel(a, b):
    return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 5e-2)])
def test_apply_and_encode(dtype, tol):
    model, state, pq = _make(784, 128, 256, 3, dtype)
    g = torch.Generator().manual_seed(1)
    x, time, lat = torch.randn(5, 784, generator=g), torch.rand(5, 2, generator=g), torch.randn(5, 256, generator=g)
    xq = x.to(dtype).double()
    out = model.apply({"params": state.work}, x.cuda(), time.cuda(), lat.cuda())
    assert _rel(out, fo.mlp_flow_apply(pq, xq, time.double(), lat.to(dtype).double())) < tol
    out0 = model.apply({"params": state.work}, x.cuda(), time.cuda(), None)
    assert _rel(out0, fo.mlp_flow_apply(pq, xq, time.double(), None)) < tol
    enc = model.apply({"params": state.work}, x.cuda(), method="encode")
    assert _rel(enc, fo.mlp_flow_encode(pq, xq)) < tol


def test_reference_property_boundary_condition():
    """test/test_improved_mean_flow.py:31-54: ConditionalFlow(noise=8, cond=32, latent=64, blocks=2), B=4,
    no latents, t == r  =>  v_pred == u (1e-6)."""
    from meanflow_audio_codec_amd import ops
    model, state, pq = _make(8, 32, 64, 2)
    w = state.work
    g = torch.Generator().manual_seed(0)
    z = torch.randn(4, 8, generator=g).cuda()
    t = torch.rand(4, 1, generator=g).cuda()
    cond_v, _ = model.conditioning(w, t, torch.zeros_like(t), None)
    v, _, _ = model.forward(w, z, cond_v)
    cond_u, cdot = model.conditioning(w, t, t - t, None, want_dot=True)
    u, dudt, _ = model.forward(w, z, cond_u, xdot=v, cond_dot=cdot)
    v_pred = u + (t - t) * dudt
    assert (v_pred - u).abs().max().item() <= 1e-6
    assert (u - v).abs().max().item() <= 1e-6


def test_reference_property_jvp_matches_reverse_mode():
    """test/test_improved_mean_flow.py:57-100: (noise=6, cond=32, latent=64, blocks=2), B=3, r = 0.5 t,
    tangent v/||v||: sum(dudt) == <grad_z sum(u), v> + sum(grad_t sum(u))  (1e-4)."""
    model, state, pq = _make(6, 32, 64, 2, seed=2)
    w = state.work
    g = torch.Generator().manual_seed(2)
    z = torch.randn(3, 6, generator=g).cuda()
    t = torch.rand(3, 1, generator=g).cuda()
    r = 0.5 * t
    v = torch.randn(3, 6, generator=g)
    v = (v / v.norm()).cuda()
    cond, cdot = model.conditioning(w, t, t - r, None, want_dot=True)
    u, dudt, ctx = model.forward(w, z, cond, xdot=v, cond_dot=cdot, save=True)
    lhs = dudt.double().sum().item()
    dz, dcond, _ = model.backward(w, ctx, torch.ones_like(u), state.grad_buffers())
    rhs = (dz.double() * v.double()).sum().item() + (dcond.double() * cdot.double()).sum().item()
    assert abs(lhs - rhs) < 1e-4, (lhs, rhs)
    # and against the oracle's jvp
    def u_fn(z_, t_, r_):
        return fo.mlp_flow_apply(pq, z_, torch.cat([t_, t_ - r_], -1), None)
    _, dref = torch.func.jvp(u_fn, (z.double().cpu(), t.double().cpu(), r.double().cpu()),
                             (v.double().cpu(), torch.ones(3, 1, dtype=torch.float64), torch.zeros(3, 1, dtype=torch.float64)))
    assert _rel(dudt, dref) < 1e-4


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 2e-4, 2e-3), (torch.bfloat16, 5e-2, 0.2)])
def test_losses_and_grads_mnist_shape(dtype, tol, gtol):
    """BASELINE config #2 shape (D=784, cond 128, latent 256), 2 blocks: FM, MF and iMF steps vs oracle."""
    from meanflow_audio_codec_amd.trainers import FlowMatchingLoss, ImprovedMeanFlowLoss, MeanFlowLoss, PRNGKey
    model, state, pq = _make(784, 128, 256, 2, dtype, seed=3)
    g = torch.Generator().manual_seed(5)
    B = 6
    x, e = torch.rand(B, 784, generator=g), torch.randn(B, 784, generator=g)
    t, r = fo.sample_tr_from_normals(torch.randn(B, 1, generator=g, dtype=torch.float64),
                                     torch.randn(B, 1, generator=g, dtype=torch.float64))
    t, r = t.float(), r.float()
    cases = [
        (ImprovedMeanFlowLoss(), dict(r=r.cuda()),
         lambda: fo.imf_loss(fo.mlp_flow_apply, fo.mlp_flow_encode, pq, x.double(), e.double(), t.double(), r.double())),
        (MeanFlowLoss(), dict(r=r.cuda()),
         lambda: fo.mf_loss(fo.mlp_flow_apply, fo.mlp_flow_encode, pq, x.double(), e.double(), t.double(), r.double())),
        (FlowMatchingLoss(), dict(),
         lambda: fo.fm_loss(fo.mlp_flow_apply, fo.mlp_flow_encode, pq, x.double(), e.double(), t.double())),
    ]
    for strat, kw, ref in cases:
        loss_ref, g_ref, _ = ref()
        loss, grads = strat.compute_loss(state, PRNGKey(0), x.cuda(), e=e.cuda(), t=t.cuda(), **kw)
        assert abs(loss.item() - loss_ref.item()) < tol * max(1.0, abs(loss_ref.item())), type(strat).__name__
        gr = fo.flatten(g_ref)
        bad = {k: _rel(grads[k], gr[k]) for k in gr if gr[k].abs().max() > 0 and not _rel(grads[k], gr[k]) < gtol}
        assert not bad, (type(strat).__name__, bad)


def test_sampler_matches_oracle_heun():
    from meanflow_audio_codec_amd.evaluators import heun_integrate, one_step_decode
    model, state, pq = _make(64, 32, 16, 2, seed=7)
    g = torch.Generator().manual_seed(9)
    x0, lat = torch.randn(3, 64, generator=g), torch.randn(3, 16, generator=g)
    ref = fo.heun_sample(fo.mlp_flow_apply, pq, x0.double(), lat.double(), n_steps=3)
    out = heun_integrate(model, state.work, x0.cuda(), lat.cuda(), 3)
    assert _rel(out, ref) < 1e-4
    ref_cfg = fo.heun_sample(fo.mlp_flow_apply, pq, x0.double(), lat.double(), n_steps=2, guidance_scale=1.5)
    out_cfg = heun_integrate(model, state.work, x0.cuda(), lat.cuda(), 2, guidance_scale=1.5)
    assert _rel(out_cfg, ref_cfg) < 1e-4
    ref1 = fo.one_step_decode(fo.mlp_flow_apply, pq, x0.double(), lat.double())
    assert _rel(one_step_decode(model, state.work, x0.cuda(), lat.cuda()), ref1) < 1e-4
