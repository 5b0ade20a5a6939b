"""GPU parity of the fused channel-mixing MLP of the Mixer block (mfc_chanmlp_fwd / mfc_chanmlp_bwd; reference
models/mlp_mixer.py:66-94) against the fp64 oracle's dense / gelu (oracle/flow_oracle.py), against the two-GEMM
formulation it replaces, and through the whole Mixer flow."""
import pytest
import torch

from oracle import flow_oracle as fo

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _case(rows, act, H, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(rows, 16, generator=g)
    W1 = torch.randn(16, H, generator=g) / 4
    W2 = torch.randn(H, 16, generator=g) / (H ** 0.5)
    b1 = torch.randn(H, generator=g) * 0.3
    b2 = torch.randn(16, generator=g) * 0.3
    res = torch.randn(rows, 16, generator=g)
    q = lambda t: t.to(dtype).double()          # both sides see the same rounded operands
    return (a, W1, b1, W2, b2, res), (q(a), q(W1), b1.double(), q(W2), b2.double(), q(res))


def _ref_fwd(a, W1, b1, W2, b2, res, act):
    p1, p2 = {"kernel": W1, "bias": b1}, {"kernel": W2, "bias": b2}
    f = lambda x: fo.dense(p2, fo.gelu(fo.dense(p1, x)))
    out = f(a[:act]) + res[:act]
    n_tan = a.shape[0] - act
    if n_tan == 0:
        return out
    _, jv = torch.func.jvp(f, (a[:n_tan],), (a[act:],))
    return torch.cat([out, jv + res[act:]], 0)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("rows,act,H", [(64, 64, 128), (50, 37, 256), (1000, 600, 512), (4096 + 24, 4096, 2048), (33, 33, 48)])
def test_forward_and_tangent_vs_oracle(dtype, tol, rows, act, H):
    from meanflow_audio_codec_amd import ops
    raw, ref = _case(rows, act, H, dtype)
    a, W1, b1, W2, b2, res = (t.cuda() for t in raw)
    out = ops.chanmlp_fwd(a.to(dtype), W1.to(dtype), b1, W2.to(dtype), b2, act_rows=act, residual=res.to(dtype))
    want = _ref_fwd(*ref, act)
    assert _rel(out[:act], want[:act]) < tol
    if rows > act:
        assert _rel(out[act:], want[act:]) < tol
    out0 = ops.chanmlp_fwd(a.to(dtype)[:act].contiguous(), W1.to(dtype), b1, W2.to(dtype), b2)
    assert _rel(out0, want[:act] - ref[5][:act]) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-5), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize("rows,H", [(64, 128), (37, 256), (1000, 512), (5000, 1024), (4100, 2048)])
def test_reverse_vs_oracle(dtype, tol, rows, H):
    from meanflow_audio_codec_amd import ops
    raw, ref = _case(rows, rows, H, dtype, seed=3)
    a, W1, b1, W2, b2, _ = (t.cuda() for t in raw)
    g = torch.Generator().manual_seed(9)
    dy = torch.randn(rows, 16, generator=g)
    a64, W164, b164, W264, b264, _ = (t.clone().requires_grad_(True) for t in ref)
    y = fo.dense({"kernel": W264, "bias": b264}, fo.gelu(fo.dense({"kernel": W164, "bias": b164}, a64)))
    (y * dy.to(dtype).double()).sum().backward()
    dW1 = torch.full((16, H), 7.0, device="cuda", dtype=dtype)          # overwritten, not accumulated
    dW2 = torch.full((H, 16), 7.0, device="cuda", dtype=dtype)
    db1 = torch.full((H,), 7.0, device="cuda")
    da = ops.chanmlp_bwd(a.to(dtype), dy.cuda().to(dtype), W1.to(dtype), b1, W2.to(dtype), dW1, db1, dW2)
    assert _rel(da, a64.grad) < tol
    assert _rel(dW1, W164.grad) < tol and _rel(dW2, W264.grad) < tol and _rel(db1, b164.grad) < tol
    # fixed-order reductions: a second launch gives the same bits
    dW1b, dW2b, db1b = torch.empty_like(dW1), torch.empty_like(dW2), torch.empty_like(db1)
    dab = ops.chanmlp_bwd(a.to(dtype), dy.cuda().to(dtype), W1.to(dtype), b1, W2.to(dtype), dW1b, db1b, dW2b)
    assert torch.equal(da, dab) and torch.equal(dW1, dW1b) and torch.equal(dW2, dW2b) and torch.equal(db1, db1b)


def test_argument_checks():
    from meanflow_audio_codec_amd import _lib, ops
    assert ops.chanmlp_ok(16, 2048) and ops.chanmlp_ok(16, 128) and not ops.chanmlp_ok(16, 40) and not ops.chanmlp_ok(32, 2048)
    a = torch.zeros(32, 16, device="cuda")
    W1, W2 = torch.zeros(16, 40, device="cuda"), torch.zeros(40, 16, device="cuda")
    with pytest.raises(_lib.MfcError):          # H % 16 != 0
        ops.chanmlp_fwd(a, W1, torch.zeros(40, device="cuda"), W2, torch.zeros(16, device="cuda"))
    W1, W2 = torch.zeros(16, 48, device="cuda"), torch.zeros(48, 16, device="cuda")
    with pytest.raises(_lib.MfcError):          # the reverse pass takes H in {128, 256, 512, k * 1024}
        ops.chanmlp_bwd(a, a, W1, torch.zeros(48, device="cuda"), W2, torch.empty_like(W1), torch.zeros(48, device="cuda"),
                        torch.empty_like(W2))
    assert _lib.lib().mfc_chanmlp_ws_elems(0, 128) < 0


KW = dict(token_mix_dim=48, channel_mix_dim=128, num_channels=16, num_latent_tokens=4, num_context_tokens=8)


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 2e-4, 3e-3), (torch.bfloat16, 5e-2, 0.25)])
def test_mixer_flow_with_fused_channel_mlp(dtype, tol, gtol):
    """The Mixer flow with a channel_mix_dim the fused kernels take (the default 2048 does; the other Mixer tests use 40,
    which goes through the GEMM formulation): forward, MeanFlow loss (JVP through the fused tangent) and every gradient."""
    from meanflow_audio_codec_amd.models import ConditionalMLPMixerFlow, TrainState, adamw
    from meanflow_audio_codec_amd.trainers import MeanFlowLoss, PRNGKey
    D, CD, LAT, NB = 64, 32, 16, 2
    model = ConditionalMLPMixerFlow(D, CD, NB, LAT, dtype=dtype, **KW)
    assert model.mix[0].fused_channel_mlp(torch.empty(16, 128, dtype=dtype))
    shapes = fo.mixer_flow_shapes(D, CD, LAT, NB, C=16, tmd=48, cmd=128, n_lat=4, n_ctx=8)
    p64 = fo.init_params(shapes, seed=2, special=False)
    flat = {k: v.float().cuda().contiguous() for k, v in fo.flatten(p64).items()}
    state = TrainState.create(apply_fn=model.apply, params=flat, tx=adamw(1e-3, 1e-2), model=model)
    pq = fo.unflatten({k: state.work[k].double().cpu() for k in flat})
    g = torch.Generator().manual_seed(5)
    B = 6
    x, e = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
    time = torch.rand(B, 2, generator=g)
    out = model.apply({"params": state.work}, x.cuda(), time.cuda(), None)
    assert _rel(out, fo.mixer_flow_apply(pq, x.to(dtype).double(), time.double(), None)) < tol
    t, r = fo.sample_tr_from_normals(torch.randn(B, 1, generator=g, dtype=torch.float64),
                                     torch.randn(B, 1, generator=g, dtype=torch.float64))
    loss_ref, g_ref, _ = fo.mf_loss(fo.mixer_flow_apply, fo.mixer_encode, pq, x.double(), e.double(), t, r)
    loss, grads = MeanFlowLoss().compute_loss(state, PRNGKey(0), x.cuda(), e=e.cuda(), t=t.float().cuda(), r=r.float().cuda())
    assert abs(loss.item() - loss_ref.item()) < tol * max(1.0, abs(loss_ref.item()))
    gr = fo.flatten(g_ref)
    bad = {k: _rel(grads[k], gr[k]) for k in gr if gr[k].abs().max() > 0 and not _rel(grads[k], gr[k]) < gtol}
    assert not bad, bad
