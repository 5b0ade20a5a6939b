"""GPU parity: element-wise loss-step kernels vs the oracle."""
import math

import pytest
import torch

from oracle import flow_oracle as fo

pytestmark = pytest.mark.gpu


def test_time_embed_and_tangent():
    from meanflow_audio_codec_amd import ops
    t = torch.rand(7, dtype=torch.float64)
    h = torch.rand(7, dtype=torch.float64)
    add = torch.randn(7, 128, dtype=torch.float64)
    ref, dref = torch.func.jvp(lambda a, b: fo.sinusoidal_embedding(a, 128) + fo.sinusoidal_embedding(b, 128),
                               (t, h), (torch.ones_like(t), torch.ones_like(h)))
    c, cd = ops.time_embed(t.float().cuda(), h.float().cuda(), 128, add=add.float().cuda(), want_dot=True)
    assert (c.double().cpu() - (ref + add)).abs().max() < 2e-6
    assert (cd.double().cpu() - dref).abs().max() < 2e-6


def test_sample_tr_rule_and_sharding():
    from meanflow_audio_codec_amd import ops
    t, r = ops.sample_tr(42, 3, 0, 64, 64, -0.4, 1.0, 0.5)
    assert torch.equal(t[:32], r[:32]) and (t[32:] >= r[32:]).all() and (t > 0).all() and (t < 1).all()
    assert not torch.equal(t[32:], r[32:])
    # two shards reproduce the global batch (per-global-row Philox + global r=t prefix)
    ta, ra = ops.sample_tr(42, 3, 0, 32, 64, -0.4, 1.0, 0.5)
    tb, rb = ops.sample_tr(42, 3, 32, 32, 64, -0.4, 1.0, 0.5)
    assert torch.equal(torch.cat([ta, tb]), t) and torch.equal(torch.cat([ra, rb]), r)
    t2, _ = ops.sample_tr(42, 4, 0, 64, 64, -0.4, 1.0, 0.5)
    assert not torch.equal(t, t2)          # the step advances the stream (reference defect 4 fixed)
    # interleaved shards (distributed.shard_rows: rank k owns rows k, k+G, ...) reproduce it too
    for G in (2, 4, 8):
        for k in range(G):
            tk, rk = ops.sample_tr(42, 3, k, 64 // G, 64, -0.4, 1.0, 0.5, row_stride=G)
            assert torch.equal(tk, t[k::G]) and torch.equal(rk, r[k::G])
            assert int((tk == rk).sum().item()) == 32 // G       # every rank gets the same share of r == t rows
    # data_size = int(B * p) in host double arithmetic (utils.py:41) -- cases where float32(p) * B lands below the
    # integer (the kernel used to recompute it: 69 instead of 70 for B=100, p=0.7)
    for B, prop in ((100, 0.7), (10, 0.7), (10, 0.9), (50, 0.9), (200, 0.7), (64, 0.5), (7, 0.0), (7, 1.0)):
        tt, rr = ops.sample_tr(5, 1, 0, B, B, -0.4, 1.0, prop)
        ds = int(B * prop)
        assert ops.data_size_of(B, prop) == ds
        assert int((tt == rr).sum().item()) == ds and torch.equal(tt[:ds], rr[:ds]) and (tt[ds:] > rr[ds:]).all()
    with pytest.raises(ValueError):
        ops.sample_tr(1, 0, 60, 8, 64, -0.4, 1.0, 0.5)          # rows 60..67 exceed the global batch
    # logit-normal(-0.4, 1): median sigmoid(-0.4)
    tl, _ = ops.sample_tr(1, 0, 0, 20000, 20000, -0.4, 1.0, 0.5, pair=False)
    assert abs(tl.median().item() - 1 / (1 + math.exp(0.4))) < 0.01


def test_randn_moments_and_flow_prepare():
    from meanflow_audio_codec_amd import ops
    e = ops.randn(7, 1, 0, 64, 4096)
    assert abs(e.mean().item()) < 0.01 and abs(e.std().item() - 1) < 0.01
    assert abs((e ** 4).mean().item() - 3.0) < 0.1
    e2 = torch.cat([ops.randn(7, 1, 0, 32, 4096), ops.randn(7, 1, 32, 32, 4096)])
    assert torch.equal(e, e2)
    x = torch.randn(5, 1001).cuda()
    t = torch.rand(5, 1).cuda()
    ee = torch.randn(5, 1001).cuda()
    z, tgt, _ = ops.flow_prepare(x, t, torch.float32, 0.001, 0.999, e=ee)
    zr = fo.linear_interpolate(x.double(), ee.double(), t.double())
    assert (z.double() - zr).abs().max() < 1e-6
    assert (tgt.double() - fo.linear_target(x.double(), ee.double())).abs().max() < 1e-6
    z2, tgt2, e_used = ops.flow_prepare(x, t, torch.bfloat16, 0.001, 0.999, seed=3, step=9, want_e=True)
    zr2 = fo.linear_interpolate(x.double(), e_used.double(), t.double())
    assert (z2.double() - zr2).abs().max() < 3e-2
    assert abs(e_used.std().item() - 1) < 0.05


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
def test_flow_loss_modes(dtype, tol):
    from meanflow_audio_codec_amd import ops
    B, D, nt = 6, 3000, 4
    g = torch.Generator().manual_seed(0)
    u = torch.randn(B, D, generator=g).to(dtype)
    dudt = torch.randn(nt, D, generator=g).to(dtype)
    tgt = torch.randn(B, D, generator=g)
    t = torch.rand(B, 1, generator=g)
    r = t * torch.rand(B, 1, generator=g)
    ud = u.double().requires_grad_(True)
    coef = torch.zeros(B, 1, dtype=torch.float64)
    coef[:nt] = (t - r)[:nt].double()
    dd = torch.zeros(B, D, dtype=torch.float64)
    dd[:nt] = dudt.double()
    # iMF weighted (loss_strategies.py:270-274)
    vp = ud + coef * dd
    ref = fo.weighted_l2_loss(vp, tgt.double())
    gref, = torch.autograd.grad(ref, ud)
    loss, du, pe = ops.flow_loss(u.cuda(), tgt.cuda(), dudt=dudt.cuda(), n_tan=nt, t=t.cuda(), r=r.cuda())
    assert abs(loss.item() - ref.item()) < tol * max(1, abs(ref.item()))
    assert (du.double().cpu() - gref).abs().max() < tol * gref.abs().max() * 3
    # plain MSE
    ref2 = ((vp - tgt.double()) ** 2).mean()
    g2, = torch.autograd.grad(ref2, ud)
    loss2, du2, _ = ops.flow_loss(u.cuda(), tgt.cuda(), dudt=dudt.cuda(), n_tan=nt, t=t.cuda(), r=r.cuda(), mode=1)
    assert abs(loss2.item() - ref2.item()) < tol * max(1, abs(ref2.item()))
    assert (du2.double().cpu() - g2).abs().max() < tol * g2.abs().max() * 3
    # MeanFlow adaptive (loss_strategies.py:184-196), gamma = 0.5
    u_tgt = tgt.double() - torch.clamp(coef, 0, 1) * dd
    dsq = ((ud - u_tgt) ** 2).mean(1)
    w = (1.0 / (dsq + 1e-3) ** 0.5).detach()
    ref3 = (w * dsq).mean()
    g3, = torch.autograd.grad(ref3, ud)
    loss3, du3, _ = ops.flow_loss(u.cuda(), tgt.cuda(), dudt=dudt.cuda(), n_tan=nt, t=t.cuda(), r=r.cuda(),
                                  kind=1, mode=2, p=0.5, c=1e-3)
    assert abs(loss3.item() - ref3.item()) < tol * max(1, abs(ref3.item()))
    assert (du3.double().cpu() - g3).abs().max() < tol * g3.abs().max() * 3
    # flow matching: no tangent
    ref4 = fo.weighted_l2_loss(ud, tgt.double())
    loss4, _, _ = ops.flow_loss(u.cuda(), tgt.cuda(), want_grad=False)
    assert abs(loss4.item() - ref4.item()) < tol


def test_gelu_colsum_axpby_cast():
    from meanflow_audio_codec_amd import ops
    pre = torch.randn(10, 128, dtype=torch.float64)
    prim, tan = torch.func.jvp(fo.gelu, (pre[:5],), (pre[5:],))
    out = ops.gelu_fwd(pre.float().cuda(), act_rows=5)
    assert (out.double().cpu() - torch.cat([prim, tan])).abs().max() < 1e-5
    dout = torch.randn(5, 128, dtype=torch.float64)
    pg = pre[:5].clone().requires_grad_(True)
    gref, = torch.autograd.grad((fo.gelu(pg) * dout).sum(), pg)
    din = ops.gelu_bwd(pre[:5].float().contiguous().cuda(), dout.float().cuda())
    assert (din.double().cpu() - gref).abs().max() < 1e-5
    X = torch.randn(37, 1000).cuda()
    assert (ops.colsum(X, scale=0.5).cpu() - 0.5 * X.cpu().sum(0)).abs().max() < 1e-4
    acc = torch.ones(1000).cuda()
    ops.colsum(X.bfloat16(), out=acc, accumulate=True)
    assert (acc.cpu() - (1 + X.bfloat16().float().cpu().sum(0))).abs().max() < 1e-3
    a, b = torch.randn(1000).cuda(), torch.randn(1000).cuda()
    assert (ops.axpby(2.0, a, -0.5, b) - (2 * a - 0.5 * b)).abs().max() < 1e-6
    assert torch.equal(ops.cast(a, torch.bfloat16), a.bfloat16())
    assert torch.equal(ops.cast(a.bfloat16(), torch.float32), a.bfloat16().float())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_axpby_vector_path_equals_scalar_path(dtype):
    """mfc_axpby takes 16 bytes per thread when n is a multiple of the vector width and the pointers are 16-byte aligned, the
    one-element kernel otherwise (a view one element into the buffer): same arithmetic per element, so the same bits."""
    from meanflow_audio_codec_amd import ops
    g = torch.Generator().manual_seed(3)
    n = 8 * 12345
    xa = torch.randn(n + 8, generator=g).to(dtype).cuda()
    ya = torch.randn(n + 8, generator=g).to(dtype).cuda()
    vec = ops.axpby(1.5, xa[:n], -0.25, ya[:n])                    # aligned, n % 8 == 0: vector kernel
    sca = ops.axpby(1.5, xa[1:n], -0.25, ya[1:n])                  # unaligned views of n - 1 elements: scalar kernel
    ref = 1.5 * xa[:n].float() - 0.25 * ya[:n].float()
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    assert (vec.float() - ref).abs().max() < tol
    assert torch.equal(vec[1:], sca)
    only_x = ops.axpby(-2.0, xa[:n])
    assert torch.equal(only_x, (-2.0 * xa[:n].float()).to(dtype))


@pytest.mark.parametrize("n", [5000, 5003, 3])
def test_adamw_matches_oracle(n):
    """4-wide vector path, its < 4 element scalar tail, and the all-scalar path."""
    from meanflow_audio_codec_amd import ops
    g = torch.Generator().manual_seed(0)
    p = torch.randn(n, generator=g)
    m, v = torch.zeros(n), torch.zeros(n)
    pd, md, vd = p.double(), m.double(), v.double()
    pg, mg, vg = p.cuda(), m.cuda(), v.cuda()
    pw = pg.bfloat16()
    for step in range(1, 5):
        gr = torch.randn(n, generator=g)
        pd, md, vd = fo.adamw_step(pd, gr.double(), md, vd, step, 1e-2, 1e-2)
        ops.adamw(pg, gr.cuda(), mg, vg, lr=1e-2, wd=1e-2, step=step, p_bf16=pw)
    assert (pg.double().cpu() - pd).abs().max() < 1e-5
    assert (mg.double().cpu() - md).abs().max() < 1e-6 and (vg.double().cpu() - vd).abs().max() < 1e-6
    assert torch.equal(pw, pg.bfloat16())
    # bf16 gradient input
    ops.adamw(pg, torch.randn(n).cuda().bfloat16(), mg, vg, lr=1e-2, wd=0.0, step=5)


def test_adamw_unaligned_views():
    """Pointers that are not 16-byte aligned take the scalar kernel and give the same update."""
    from meanflow_audio_codec_amd import ops
    g = torch.Generator().manual_seed(1)
    n = 1024
    p, gr = torch.randn(n + 1, generator=g).cuda(), torch.randn(n + 1, generator=g).cuda()
    m1, v1, m2, v2 = (torch.zeros(n + 1).cuda() for _ in range(4))
    pa, pb = p.clone(), p.clone()
    ops.adamw(pa[1:], gr[1:], m1[1:], v1[1:], lr=1e-2, wd=1e-2, step=1)          # offset by 4 bytes
    pc, gc, mc, vc = pb[1:].clone(), gr[1:].clone(), m2[1:].clone(), v2[1:].clone()  # aligned copies
    ops.adamw(pc, gc, mc, vc, lr=1e-2, wd=1e-2, step=1)
    assert torch.equal(pa[1:], pc) and torch.equal(m1[1:], mc) and torch.equal(v1[1:], vc)


def test_adamw_multi_equals_one_launch_per_leaf():
    """mfc_adamw_multi (one launch per 48 leaves, descriptors in the kernel arguments) == mfc_adamw leaf by leaf, bit
    for bit: odd sizes, fp32 and bf16 gradients, with and without a bf16 working copy, more than one chunk of leaves,
    several steps on the same table (the table holds raw pointers of tensors that are updated in place)."""
    import torch
    from meanflow_audio_codec_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    sizes = [1, 3, 16, 255, 256, 1000, 1024, 4099, 65537] + [7 * (i + 1) for i in range(50)]
    a, b = [], []
    for i, n in enumerate(sizes):
        p = torch.randn(n, generator=g, device="cuda")
        gdt = torch.bfloat16 if i % 3 == 0 else torch.float32
        w = p.bfloat16() if i % 2 == 0 else None
        a.append([p.clone(), None, torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), None if w is None else w.clone(), gdt])
        b.append([p.clone(), None, torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), None if w is None else w.clone(), gdt])
    grads = [torch.empty(n, device="cuda", dtype=l[5]) for n, l in zip(sizes, a)]
    items, cnt = ops.adamw_multi_items([(l[0], gr, l[2], l[3], l[4]) for l, gr in zip(b, grads)])
    assert cnt == len(sizes) > 48
    for step in range(1, 4):
        for gr in grads:
            gr.copy_(torch.randn(gr.shape, generator=g, device="cuda") * 0.1)
        for l, gr in zip(a, grads):
            ops.adamw(l[0], gr, l[2], l[3], lr=1e-2, wd=1e-2, step=step, p_bf16=l[4], grad_scale=0.5)
        ops.adamw_multi(items, cnt, lr=1e-2, wd=1e-2, step=step, grad_scale=0.5)
        for la, lb in zip(a, b):
            assert torch.equal(la[0], lb[0]) and torch.equal(la[2], lb[2]) and torch.equal(la[3], lb[3])
            if la[4] is not None:
                assert torch.equal(la[4], lb[4])
