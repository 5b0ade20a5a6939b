"""GPU parity: ConvNeXt block interior kernels (fwd, JVP, bwd) vs the fp64 oracle."""
import pytest
import torch

from oracle import flow_oracle as fo

pytestmark = pytest.mark.gpu


def _setup(R, s, seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = fo.conv_flow_shapes(s * s, 128, 0, 1)["blocks_0"]["conv_block"]
    p = fo.init_params(shapes, seed=seed + 1, special=False)
    h0 = torch.randn(R, s, s, 16, generator=g, dtype=torch.float64) * 1.5 + 0.3
    sc = 0.3 * torch.randn(R, 16, generator=g, dtype=torch.float64)
    sh = 0.3 * torch.randn(R, 16, generator=g, dtype=torch.float64)
    return p, h0, sc, sh, g


def _oracle(p, h0, sc, sh):
    h = fo.layer_norm(h0)
    h = (1.0 + sc[:, None, None, :]) * h + sh[:, None, None, :]
    return fo.convnext_block(p, h)


def _weights(p, dtype):
    f = lambda t, dt: t.to(dt).contiguous().cuda()
    return {"conv_w": f(p["Conv_0"]["kernel"], dtype), "conv_b": f(p["Conv_0"]["bias"], torch.float32),
            "exp_w": f(p["Conv_1"]["kernel"].reshape(16, 32), dtype), "exp_b": f(p["Conv_1"]["bias"], torch.float32),
            "grn_gamma": f(p["GlobalResponseNormalization_0"]["gamma"], torch.float32),
            "grn_beta": f(p["GlobalResponseNormalization_0"]["beta"], torch.float32),
            "con_w": f(p["Conv_2"]["kernel"].reshape(32, 16), dtype), "con_b": f(p["Conv_2"]["bias"], torch.float32),
            "ls": f(p["layer_scale_gamma"], torch.float32)}


def _rel(a, b):
    return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


# (2, 520): 33 x 33 tiles, a workgroup spans two rows r.  (1, 626), (2, 626): the LITERAL spatial size of BASELINE config #4
# (s = floor(sqrt(392704)); 40 x 40 = 1600 tiles per image, the geometry the persistent grids are tuned on; the last tile
# row / column holds 2 of 16 image rows / columns, so the below-the-image row skip and the border DMA path both run)
CASES = [(3, 20), (1, 16), (2, 37), (2, 530), (2, 520), (1, 626), (2, 626)]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize("R,s", CASES)
def test_forward_and_jvp(dtype, tol, R, s):
    from meanflow_audio_codec_amd import ops
    p, h0, sc, sh, g = _setup(R, s)
    h0d = torch.randn(h0.shape, generator=g, dtype=torch.float64)
    scd = 0.5 * torch.randn(sc.shape, generator=g, dtype=torch.float64)
    shd = 0.5 * torch.randn(sh.shape, generator=g, dtype=torch.float64)
    # bf16: the oracle sees the same rounded inputs
    h0q = h0.to(dtype).double()
    h0dq = h0d.to(dtype).double()
    pq = fo.tree_map(lambda t: t, p)
    if dtype == torch.bfloat16:
        for k in ("Conv_0", "Conv_1", "Conv_2"):
            pq[k] = dict(pq[k], kernel=p[k]["kernel"].to(dtype).double())
    o_ref, od_ref = torch.func.jvp(lambda a, b, c: _oracle(pq, a, b, c), (h0q, sc, sh), (h0dq, scd, shd))
    w = _weights(p, dtype)
    f32 = lambda t: t.float().contiguous().cuda()
    h1, rho = ops.ln16(h0.to(dtype).cuda())          # the first LayerNorm is a separate (GEMM-fused) step
    h1d = ops.ln16_jvp(h1, rho, h0d.to(dtype).cuda())  # ... and so is its tangent
    o, od, G, q = ops.cnx_forward(h1, f32(sc), f32(sh), w, s, h0dot=h1d, scaledot=f32(scd), shiftdot=f32(shd))
    assert _rel(o, o_ref) < tol, ("primal", _rel(o, o_ref))
    assert _rel(od, od_ref) < tol, ("tangent", _rel(od, od_ref))
    # primal-only entry gives the same primal
    o2, od2, _, _ = ops.cnx_forward(h1, f32(sc), f32(sh), w, s)
    assert od2 is None
    assert _rel(o2, o_ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-5), (torch.bfloat16, 6e-2)])
@pytest.mark.parametrize("R,s", CASES)
def test_backward(dtype, tol, R, s):
    from meanflow_audio_codec_amd import ops
    p, h0, sc, sh, g = _setup(R, s, seed=5)
    dout = torch.randn(h0.shape, generator=g, dtype=torch.float64)
    h0q = h0.to(dtype).double().requires_grad_(True)
    doutq = dout.to(dtype).double()
    pq = fo.tree_map(lambda t: t.clone(), p)
    if dtype == torch.bfloat16:
        for k in ("Conv_0", "Conv_1", "Conv_2"):
            pq[k]["kernel"] = p[k]["kernel"].to(dtype).double()
    pq = fo.tree_map(lambda t: t.requires_grad_(True), pq)
    scq, shq = sc.clone().requires_grad_(True), sh.clone().requires_grad_(True)
    o_ref = _oracle(pq, h0q, scq, shq)
    flat = fo.flatten(pq)
    names = list(flat)
    grads = torch.autograd.grad((o_ref * doutq).sum(), [h0q, scq, shq] + [flat[n] for n in names])
    gref = dict(zip(["h0", "sc", "sh"] + names, grads))

    w = _weights(p, dtype)
    f32 = lambda t: t.float().contiguous().cuda()
    h0g, rho = ops.ln16(h0.to(dtype).cuda())
    o, _, G, q = ops.cnx_forward(h0g, f32(sc), f32(sh), w, s)
    gacc = {k: torch.zeros(v.shape, dtype=torch.float32, device="cuda") for k, v in w.items()}
    dh0, dsc, dsh = ops.cnx_backward(h0g, f32(sc), f32(sh), w, s, G, q, dout.to(dtype).cuda(), gacc, rho0=rho)
    checks = {
        "h0": (dh0, gref["h0"]), "sc": (dsc, gref["sc"]), "sh": (dsh, gref["sh"]),
        "conv_w": (gacc["conv_w"], gref["Conv_0/kernel"]), "conv_b": (gacc["conv_b"], gref["Conv_0/bias"]),
        "exp_w": (gacc["exp_w"], gref["Conv_1/kernel"].reshape(16, 32)), "exp_b": (gacc["exp_b"], gref["Conv_1/bias"]),
        "gamma": (gacc["grn_gamma"], gref["GlobalResponseNormalization_0/gamma"]),
        "beta": (gacc["grn_beta"], gref["GlobalResponseNormalization_0/beta"]),
        "con_w": (gacc["con_w"], gref["Conv_2/kernel"].reshape(32, 16)), "con_b": (gacc["con_b"], gref["Conv_2/bias"]),
        "ls": (gacc["ls"], gref["layer_scale_gamma"]),
    }
    errs = {k: _rel(a, b) for k, (a, b) in checks.items()}
    bad = {k: v for k, v in errs.items() if not v < tol}
    assert not bad, errs


@pytest.mark.parametrize("dtype,tol,btol", [(torch.float32, 2e-5, 5e-5), (torch.bfloat16, 4e-2, 6e-2)])
@pytest.mark.parametrize("R,s", CASES)
def test_kept_n1_paths(dtype, tol, btol, R, s):
    """The statistics pass keeps n1 = LN(conv(FiLM(h1))) and its 1/sigma (mfc_cnx_stats_save); the apply pass
    (mfc_cnx_apply_n1) and the first two reverse passes (mfc_cnx_bwd_stats_n1 / mfc_cnx_bwd_main_n1) start from them:
    the forward result is BIT-identical to the h1-based kernels (the expansion consumed exactly this n1), every
    gradient matches the fp64 oracle at the tolerance of the h1-based path."""
    from meanflow_audio_codec_amd import ops
    p, h0, sc, sh, g = _setup(R, s, seed=7)
    dout = torch.randn(h0.shape, generator=g, dtype=torch.float64)
    h0q = h0.to(dtype).double().requires_grad_(True)
    doutq = dout.to(dtype).double()
    pq = fo.tree_map(lambda t: t.clone(), p)
    if dtype == torch.bfloat16:
        for k in ("Conv_0", "Conv_1", "Conv_2"):
            pq[k]["kernel"] = p[k]["kernel"].to(dtype).double()
    pq = fo.tree_map(lambda t: t.requires_grad_(True), pq)
    scq, shq = sc.clone().requires_grad_(True), sh.clone().requires_grad_(True)
    o_ref = _oracle(pq, h0q, scq, shq)
    flat = fo.flatten(pq)
    names = list(flat)
    grads = torch.autograd.grad((o_ref * doutq).sum(), [h0q, scq, shq] + [flat[n] for n in names])
    gref = dict(zip(["h0", "sc", "sh"] + names, grads))

    w = _weights(p, dtype)
    f32 = lambda t: t.float().contiguous().cuda()
    h1, rho = ops.ln16(h0.to(dtype).cuda())
    o_old, _, G_old, q_old = ops.cnx_forward(h1, f32(sc), f32(sh), w, s)
    n1 = torch.full_like(h1, float("nan"))
    rho1 = torch.full((R, s, s), float("nan"), dtype=torch.float32, device="cuda")
    o, _, G, q = ops.cnx_forward(h1, f32(sc), f32(sh), w, s, keep=(n1, rho1))
    assert torch.isfinite(n1.float()).all() and torch.isfinite(rho1).all() and (rho1 > 0).all()
    assert torch.equal(G, G_old) and torch.equal(q, q_old)
    assert torch.equal(o, o_old), (o.float() - o_old.float()).abs().max().item()
    assert _rel(o, o_ref.detach()) < tol
    # the tangent statistics pass keeps the same n1 (primal rows)
    h0d = torch.randn(h0.shape, generator=g, dtype=torch.float64)
    h1d = ops.ln16_jvp(h1, rho, h0d.to(dtype).cuda())
    n1b, rho1b = torch.zeros_like(n1), torch.zeros_like(rho1)
    jkw = dict(h0dot=h1d, scaledot=f32(0.1 * sc), shiftdot=f32(0.1 * sh))
    oj, odj, Gj, qj = ops.cnx_forward(h1, f32(sc), f32(sh), w, s, keep=(n1b, rho1b), **jkw)
    assert torch.equal(n1b, n1) and torch.equal(rho1b, rho1)
    # ... and the tangent of n1: the tangent apply pass from (n1, n1dot) is bit-identical to the tile kernel as well
    oj_old, odj_old, Gj_old, qj_old = ops.cnx_forward(h1, f32(sc), f32(sh), w, s, **jkw)
    assert torch.equal(Gj, Gj_old) and torch.equal(qj, qj_old) and torch.equal(oj, oj_old)
    assert torch.equal(odj, odj_old), (odj.float() - odj_old.float()).abs().max().item()
    # reverse pass from n1
    gacc = {k: torch.zeros(v.shape, dtype=torch.float32, device="cuda") for k, v in w.items()}
    dh0, dsc, dsh = ops.cnx_backward(h1, f32(sc), f32(sh), w, s, G, q, dout.to(dtype).cuda(), gacc, rho0=rho, n1=n1, rho1=rho1)
    checks = {
        "h0": (dh0, gref["h0"]), "sc": (dsc, gref["sc"]), "sh": (dsh, gref["sh"]),
        "conv_w": (gacc["conv_w"], gref["Conv_0/kernel"]), "conv_b": (gacc["conv_b"], gref["Conv_0/bias"]),
        "exp_w": (gacc["exp_w"], gref["Conv_1/kernel"].reshape(16, 32)), "exp_b": (gacc["exp_b"], gref["Conv_1/bias"]),
        "gamma": (gacc["grn_gamma"], gref["GlobalResponseNormalization_0/gamma"]),
        "beta": (gacc["grn_beta"], gref["GlobalResponseNormalization_0/beta"]),
        "con_w": (gacc["con_w"], gref["Conv_2/kernel"].reshape(32, 16)), "con_b": (gacc["con_b"], gref["Conv_2/bias"]),
        "ls": (gacc["ls"], gref["layer_scale_gamma"]),
    }
    errs = {k: _rel(a, b) for k, (a, b) in checks.items()}
    bad = {k: v for k, v in errs.items() if not v < btol}
    assert not bad, errs


@pytest.mark.parametrize("dtype,tol,btol", [(torch.float32, 2e-5, 5e-5), (torch.bfloat16, 4e-2, 6e-2)])
def test_few_persistent_workgroups(dtype, tol, btol):
    """7 workgroups walk all 3 x 9 tiles: every one crosses tile rows, image rows r, both DMA buffers and flushes its
    per-r accumulators several times -- same results as the oracle."""
    from meanflow_audio_codec_amd import _lib
    old = _lib.lib().mfc_cnx_max_blocks(7)
    try:
        test_forward_and_jvp(dtype, tol, 3, 37)
        test_backward(dtype, btol, 3, 37)
        test_kept_n1_paths(dtype, tol, btol, 3, 37)
    finally:
        _lib.lib().mfc_cnx_max_blocks(old)
