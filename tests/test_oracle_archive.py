"""The oracle's loss / sampler cores against numbers the REFERENCE ITSELF produced: its PyTorch implementations
``references/archive/{flow,mflow,imflow}.py`` were run in the build container by
``tests/golden/gen_archive_flow_golden.py`` (fixtures: ``tests/golden/archive_flow_golden.npz``).

``oracle.flow_oracle.{fm_core, mf_core, imf_core, heun_integrate}`` are the single implementation behind both the
JAX-path restatements used by every GPU parity test (``fm_loss``, ``mf_loss``, ``imf_loss``, ``heun_sample``) and the
archive variants checked here, so a wrong interpolation, target, JVP tangent, stop-gradient, weight or integrator
fails this file.  Everything runs in float64; agreement is to rounding."""
import numpy as np
import pytest
import torch

from oracle import flow_oracle as fo


@pytest.fixture(scope="module")
def gold(golden_dir):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(golden_dir / "archive_flow_golden.npz").items()}


def _params(gold, prefix):
    pre = f"{prefix}/param/"
    return {k[len(pre):]: v for k, v in gold.items() if k.startswith(pre)}


def _check_grads(gold, prefix, grads, rtol=1e-9):
    pre = f"{prefix}/grad/"
    names = [k[len(pre):] for k in gold if k.startswith(pre)]
    assert names and set(names) == set(grads)
    for n in names:
        ref = gold[pre + n]
        scale = max(ref.abs().max().item(), 1e-30)
        assert (grads[n] - ref).abs().max().item() <= rtol * scale + 1e-300, n
    # at least the class embedding rows of unused classes are exactly zero in both
    return len(names)


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()


def test_improved_mean_flow_core_matches_reference_archive(gold):
    p = _params(gold, "imf")
    x0, cls = gold["x0"], gold["cls"].long()
    t, r, e = gold["imf/t"][:, None], gold["imf/r"][:, None], gold["imf/e"]
    loss, grads, aux = fo.archive_imf_loss(p, x0, cls, e, t, r)
    assert abs(loss.item() - gold["imf/loss"].item()) <= 1e-10 * abs(gold["imf/loss"].item())
    assert _rel(aux["v"], gold["imf/v"]) < 1e-10 and _rel(aux["u"], gold["imf/u"]) < 1e-10
    assert _rel(aux["dudt"], gold["imf/dudt"]) < 1e-9
    assert _check_grads(gold, "imf", grads) == len(p) - 0
    # rows with r == t: du/dt is multiplied by exactly zero, so V == u there (test_improved_mean_flow.py:31-54)
    same = (t == r)[:, 0]
    assert same.any() and (~same).any()
    # the JAX-path tangent (v, 1, 0) is a DIFFERENT directional derivative: the core's parameter matters
    u2, d2, _ = fo.imf_core(lambda z_, t_, r_: fo.archive_net(p, z_, t_, r_, cls), aux["v"],
                            fo.linear_interpolate(x0, e, t, 0.0, 1.0), t, r, tangent="t")
    assert _rel(u2, gold["imf/u"]) < 1e-10 and _rel(d2, gold["imf/dudt"]) > 1e-3
    # stop-gradient placement: letting the gradient flow through du/dt changes the gradients (so the fixture would
    # catch a missing .detach())
    pr = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    z = fo.linear_interpolate(x0, e, t, 0.0, 1.0)
    v = fo.archive_net(pr, z, t, t, cls)
    u, dudt = torch.func.jvp(lambda z_, t_, r_: fo.archive_net(pr, z_, t_, r_, cls), (z, t, r),
                             (v, torch.zeros_like(t), torch.ones_like(t)))
    bad = ((u + (t - r) * dudt - (e - x0)) ** 2).mean()
    gb = torch.autograd.grad(bad, pr["blocks.0.mlp.0.weight"])[0]
    assert _rel(gb, gold["imf/grad/blocks.0.mlp.0.weight"]) > 1e-3


def test_mean_flow_core_matches_reference_archive(gold):
    p = _params(gold, "mf")
    x0, cls = gold["x0"], gold["cls"].long()
    t, r, e = gold["mf/t"][:, None], gold["mf/r"][:, None], gold["mf/e"]
    loss, grads, _ = fo.archive_mf_loss(p, x0, cls, e, t, r, gamma=0.5, c=1e-3)
    assert abs(loss.item() - gold["mf/loss"].item()) <= 1e-10 * abs(gold["mf/loss"].item())
    _check_grads(gold, "mf", grads)


def test_flow_matching_core_matches_reference_archive(gold):
    p = _params(gold, "fm")
    x0, cls = gold["x0"], gold["cls"].long()
    loss, grads, _ = fo.archive_fm_loss(p, x0, cls, gold["fm/e"], gold["fm/t"], 0.001, 0.999)
    assert abs(loss.item() - gold["fm/loss"].item()) <= 1e-12 * abs(gold["fm/loss"].item())
    _check_grads(gold, "fm", grads, rtol=1e-10)


def test_heun_samplers_match_reference_archive(gold):
    cls = gold["cls"].long()
    B = cls.shape[0]
    # evaluators/sampling.py's integrator == archive/flow.py:115-124 (one time input, k2 at t - dt)
    p = _params(gold, "fm")
    with torch.no_grad():
        out = fo.heun_integrate(lambda x, tv: fo.archive_net(p, x, torch.full((B, 1), float(tv), dtype=x.dtype), None, cls),
                                gold["fm/sample_x0"], 4)
    assert _rel(out, gold["fm/sample_n4"]) < 1e-10
    # the two-time sampler of archive/imflow.py:170-182 and mflow.py:154-166
    for pre, n, key in (("imf", 3, "imf/sample_n3"), ("mf", 2, "mf/sample_n2")):
        p = _params(gold, pre)
        f = lambda x, tv, rv: fo.archive_net(p, x, torch.full((B, 1), float(tv), dtype=x.dtype),
                                             torch.full((B, 1), float(rv), dtype=x.dtype), cls)
        with torch.no_grad():
            out = fo.heun_two_time(f, gold[f"{pre}/sample_x0"], n)
        assert _rel(out, gold[key]) < 1e-9, pre


def test_jax_path_losses_run_on_the_same_cores():
    """fm_loss / mf_loss / imf_loss / heun_sample (what the GPU parity tests compare with) call the pinned cores:
    evaluate them once through the public entry points and once by hand through the cores."""
    g = torch.Generator().manual_seed(0)
    D, CD, L, NB, B = 12, 8, 6, 2, 4
    params = fo.init_params(fo.mlp_flow_shapes(D, CD, L, NB), seed=1, special=False)
    x, e = torch.randn(B, D, generator=g, dtype=torch.float64), torch.randn(B, D, generator=g, dtype=torch.float64)
    t, r = fo.sample_tr_from_normals(torch.randn(B, 1, generator=g, dtype=torch.float64),
                                     torch.randn(B, 1, generator=g, dtype=torch.float64))
    loss, _, aux = fo.imf_loss(fo.mlp_flow_apply, fo.mlp_flow_encode, params, x, e, t, r, use_weighted_loss=False)
    lat = fo.mlp_flow_encode(params, x)
    z = fo.linear_interpolate(x, e, t)
    v = fo.mlp_flow_apply(params, z, torch.cat([t, torch.zeros_like(t)], -1), lat)
    u, dudt, V = fo.imf_core(lambda z_, t_, r_: fo.mlp_flow_apply(params, z_, torch.cat([t_, t_ - r_], -1), lat), v, z, t, r)
    assert torch.allclose(aux["u"], u) and torch.allclose(aux["dudt"], dudt)
    assert abs(loss.item() - ((V - fo.linear_target(x, e)) ** 2).mean().item()) < 1e-14
    x1 = fo.heun_sample(fo.mlp_flow_apply, params, e, lat, 3)
    f = lambda xx, tv: fo.mlp_flow_apply(params, xx, torch.cat([torch.full((B, 1), float(tv), dtype=xx.dtype),
                                                                torch.zeros(B, 1, dtype=xx.dtype)], -1), lat)
    assert torch.equal(x1, fo.heun_integrate(f, e, 3))
