"""CPU: pin the flow oracle with the reference's own property tests, restated.

* test/test_improved_mean_flow.py:31-54  -- t == r  =>  v_pred == u   (1e-6)
* test/test_improved_mean_flow.py:57-100 -- forward-mode JVP == reverse-mode
  <grad_z sum(u), v> + sum(grad_t sum(u))   (1e-4)
both on ConditionalFlow without latents (zero-latent branch), tangent normalised
v/||v||; here additionally on the ConvNeXt flow.
"""
import math

import pytest
import torch

from oracle import flow_oracle as fo



@pytest.fixture(autouse=True)
def _float64_default():
    """the oracle runs in float64; restore the process default afterwards (other test modules rely on fp32)"""
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(old)


def _mlp_params(noise, cond=32, latent=64, blocks=2, seed=0):
    return fo.init_params(fo.mlp_flow_shapes(noise, cond, latent, blocks), seed=seed, special=False)


def _conv_params(D=64, cond=32, blocks=2, seed=0, latent_in=0):
    return fo.init_params(fo.conv_flow_shapes(D, cond, latent_in, blocks), seed=seed, special=False)


@pytest.mark.parametrize("kind", ["mlp", "conv"])
def test_boundary_condition_t_equals_r(kind):
    g = torch.Generator().manual_seed(0)
    B = 4
    if kind == "mlp":
        params, apply, D = _mlp_params(8), fo.mlp_flow_apply, 8
    else:
        params, apply, D = _conv_params(64), fo.conv_flow_apply, 64
    x = torch.zeros(B, D) if kind == "mlp" else torch.randn(B, D, generator=g)   # the reference's x: zeros (:44)
    e = torch.randn(B, D, generator=g)
    t = torch.rand(B, 1, generator=g)
    r = t.clone()
    v, u, dudt, v_pred, target = fo.imf_parts(apply, None, params, x, e, t, r)
    assert torch.allclose(v_pred, u, atol=1e-6)
    # the reference's helper normalises the tangent (v / (||v|| + 1e-6) per row, :23): same property
    z = fo.linear_interpolate(x, e, t)
    v_dir = v / (v.norm(dim=-1, keepdim=True) + 1e-6)
    u2, _, v_pred2 = fo.imf_core(lambda z_, t_, r_: apply(params, z_, torch.cat([t_, t_ - r_], -1), None), v_dir, z, t, r)
    assert torch.allclose(v_pred2, u2, rtol=1e-6, atol=1e-6) and torch.allclose(u2, u, atol=1e-12)
    # and with h = 0 the u-pass equals the v-pass
    assert torch.allclose(u, v, atol=1e-12)


@pytest.mark.parametrize("kind", ["mlp", "conv"])
def test_jvp_matches_reverse_mode(kind):
    g = torch.Generator().manual_seed(2)
    B = 3
    if kind == "mlp":
        params, apply, D = _mlp_params(6), fo.mlp_flow_apply, 6
    else:
        params, apply, D = _conv_params(49), fo.conv_flow_apply, 49
    x = torch.randn(B, D, generator=g)
    e = torch.randn(B, D, generator=g)
    t = torch.rand(B, 1, generator=g)
    r = 0.5 * t
    z = fo.linear_interpolate(x, e, t)                       # noised = (1-t) x + (0.001 + 0.999 t) noise (:77)
    v = apply(params, z, torch.cat([t, torch.zeros_like(t)], -1), None)
    v = (v / (v.norm(dim=-1, keepdim=True) + 1e-6)).detach()  # v_dir (:86-87)

    def u_fn(z_, t_, r_):
        return apply(params, z_, torch.cat([t_, t_ - r_], -1), None)

    _, dudt = torch.func.jvp(u_fn, (z, t, r), (v, torch.ones_like(t), torch.zeros_like(r)))
    zz = z.clone().requires_grad_(True)
    tt = t.clone().requires_grad_(True)
    s = u_fn(zz, tt, r).sum()
    gz, gt = torch.autograd.grad(s, [zz, tt])
    lhs = dudt.sum()
    rhs = (gz * v).sum() + gt.sum()
    assert abs(lhs.item() - rhs.item()) < 1e-4


def test_layer_norm_and_gelu_closed_forms():
    x = torch.randn(5, 16)
    ref = torch.nn.functional.layer_norm(x, (16,), eps=1e-6)
    assert torch.allclose(fo.layer_norm(x), ref, atol=1e-10)
    assert torch.allclose(fo.gelu(x), torch.nn.functional.gelu(x, approximate="tanh"), atol=1e-12)


def test_weighted_l2_is_pe_over_pe_plus_c():
    p, t = torch.randn(4, 10), torch.randn(4, 10)
    pe = ((p - t) ** 2).sum(1)
    assert torch.allclose(fo.weighted_l2_loss(p, t), (pe / (pe + 1e-3)).mean())


def test_sample_tr_rule():
    nt, nr = torch.randn(8, 1), torch.randn(8, 1)
    t, r = fo.sample_tr_from_normals(nt, nr)
    assert (t >= r).all()
    assert torch.equal(t[:4], r[:4])          # first int(B*0.5) rows: r = t (utils.py:41-44)
    assert (t[4:] >= r[4:]).all()


def test_adamw_matches_torch_optim():
    p = torch.randn(7, 3)
    w = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([w], lr=1e-2, weight_decay=0.1, betas=(0.9, 0.999), eps=1e-8)
    m, v, q = torch.zeros_like(p), torch.zeros_like(p), p.clone()
    for step in range(1, 4):
        g = torch.randn(7, 3)
        w.grad = g.clone()
        opt.step()
        q, m, v = fo.adamw_step(q, g, m, v, step, 1e-2, 0.1)
    # torch decays with p*(1-lr*wd) before the Adam term, optax adds wd*p to the update:
    # identical to first order in lr*wd; exact formula checked separately below
    assert torch.allclose(q, w.detach(), atol=5e-5)
    p1, m1, v1 = fo.adamw_step(p, torch.ones_like(p), torch.zeros_like(p), torch.zeros_like(p), 1, 0.1, 0.0)
    assert torch.allclose(p1, p - 0.1 * (1.0 / (1.0 + 1e-8)))


def test_imf_loss_grads_finite_and_param_count():
    shapes = fo.conv_flow_shapes(392704, 128, 256, 8)
    n = sum(math.prod(s) for s in fo.flatten(shapes).values())
    assert abs(n / 1e9 - 13.70) < 0.02, n      # 13.70 B parameters
    params = _conv_params(64, latent_in=8)
    g = torch.Generator().manual_seed(1)
    x, e = torch.randn(4, 64, generator=g), torch.randn(4, 64, generator=g)
    t, r = fo.sample_tr_from_normals(torch.randn(4, 1, generator=g), torch.randn(4, 1, generator=g))
    lat = torch.randn(4, 8, generator=g)
    loss, grads, aux = fo.imf_loss(fo.conv_flow_apply, lambda p, xx: lat, params, x, e, t, r)
    assert torch.isfinite(loss)
    for k, v in fo.flatten(grads).items():
        assert torch.isfinite(v).all(), k
    assert fo.flatten(grads)["blocks_0/input_proj2/kernel"].abs().sum() > 0
    # rows with r == t contribute no dudt term
    assert torch.allclose(aux["v_pred"][:2], aux["u"][:2])


def test_heun_and_one_step_shapes():
    params = _conv_params(64)
    x = torch.randn(2, 64)
    out = fo.heun_sample(fo.conv_flow_apply, params, x, None, n_steps=2)
    assert out.shape == (2, 64)
    assert fo.one_step_decode(fo.conv_flow_apply, params, x, None).shape == (2, 64)
