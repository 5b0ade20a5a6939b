"""GPU: the HIP training step followed over SEVERAL optimizer updates against the fp64 oracle, at the small shape and
at SURVEY 8(d)'s CI shape (D = 32256, s = 179, S = 512656, 1.12 B parameters), and the run-to-run determinism the
fixed-order reductions buy.

VERDICT r1 weak #2/#4: one-update comparisons at D = 400 could not say whether the literal-size run's growing
unweighted error (BENCH_r01: 6.6 -> 2e7 in 25 updates while the weighted loss prints 1.0) is arithmetic or a defect.
Here the unweighted mean squared error, the loss and sampled parameters are followed update by update.

How an AdamW trajectory is compared.  The first updates move every element by ~lr * sign(g) (m_hat / sqrt(v_hat)
is +-1 until the moments have history), so an element whose gradient is within rounding noise of zero can differ by up to
2 lr per update between two correct implementations.  The assertions therefore bound (a) the bulk: >= 99 % of the
sampled elements within 2 % of the distance travelled, and (b) every element within the distance an Adam step can
travel."""
import time

import pytest
import torch

from oracle import flow_oracle as fo

pytestmark = pytest.mark.gpu


def _oracle_state(flat64):
    return ({k: v.clone() for k, v in flat64.items()}, {k: torch.zeros_like(v) for k, v in flat64.items()},
            {k: torch.zeros_like(v) for k, v in flat64.items()})


def _oracle_update(p, m, v, x, e, t, r, step, lr, wd, pre=None):
    loss, grads, aux = pre if pre is not None else fo.imf_loss(fo.conv_flow_apply, fo.conv_flow_encode,
                                                                   fo.unflatten(p), x, e, t, r)
    gf = fo.flatten(grads)
    for k in p:
        p[k], m[k], v[k] = fo.adamw_step(p[k], gf[k], m[k], v[k], step, lr, wd)
    target = fo.linear_target(x, e)
    mse = ((aux["v_pred"] - target) ** 2).mean().item()
    return loss.item(), mse, gf


def _sample_idx(n, k, gen):
    return torch.arange(n) if n <= k else torch.randperm(n, generator=gen)[:k]


def _compare_params(state, p_ref, p0, lr, steps, gen, label):
    worst = 0.0
    for k, ref in p_ref.items():
        idx = _sample_idx(ref.numel(), 20000, gen)
        hip = state.params[k].reshape(-1)[idx.to(state.params[k].device)].double().cpu()
        rf = ref.reshape(-1)[idx]
        d = (hip - rf).abs()
        travelled = lr * steps
        frac_off = (d > 0.02 * travelled + 1e-9).double().mean().item()
        assert frac_off < 0.01, (label, k, frac_off, d.max().item())
        bound = 2.02 * travelled * (1.0 + 1e-2) + 1e-7
        assert d.max().item() <= bound, (label, k, d.max().item(), bound)
        # and the element really moved like one Adam step per update
        moved = (hip - p0[k].reshape(-1)[idx]).abs().max().item()
        assert moved <= 1.05 * travelled * (1 + 0.01 * steps) + 1e-7, (label, k, moved)
        worst = max(worst, d.max().item())
    return worst


def test_small_shape_five_updates_follow_the_oracle():
    from meanflow_audio_codec_amd.models import ConditionalConvFlow, TrainState, adamw
    from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, PRNGKey, train_step
    D, CD, LAT, NB, B, lr, wd, steps = 400, 128, 24, 2, 6, 1e-3, 1e-2, 5
    model = ConditionalConvFlow(D, CD, NB, LAT, dtype=torch.float32)
    p64 = fo.flatten(fo.init_params(fo.conv_flow_shapes(D, CD, LAT, NB, latent_dim=LAT), seed=41, special=False))
    p64 = {k: v.float().double() for k, v in p64.items()}            # both sides start from the same fp32 values
    state = TrainState.create(apply_fn=model.apply, params={k: v.float().cuda().contiguous() for k, v in p64.items()},
                              tx=adamw(lr, wd), model=model)
    p0 = {k: v.clone() for k, v in p64.items()}
    p, m, v = _oracle_state(p64)
    g = torch.Generator().manual_seed(43)
    x = torch.randn(B, D, generator=g, dtype=torch.float64)
    gen = torch.Generator().manual_seed(1)
    key = PRNGKey(0)
    mse_hip, mse_ref = [], []
    for step in range(1, steps + 1):
        e = torch.randn(B, D, generator=g, dtype=torch.float64)
        t, r = fo.sample_tr_from_normals(torch.randn(B, 1, generator=g, dtype=torch.float64),
                                         torch.randn(B, 1, generator=g, dtype=torch.float64))
        loss_ref, mse_r, _ = _oracle_update(p, m, v, x, e, t, r, step, lr, wd)
        ef, tf, rf = e.float().cuda(), t.float().cuda(), r.float().cuda()
        aux = {}

        class Fixed(ImprovedMeanFlowLoss):
            def compute_loss(self, st, k_, xx, **kw):
                kw.pop("row0", None); kw.pop("global_batch", None); kw.pop("row_stride", None)
                return super().compute_loss(st, k_, xx, e=ef, t=tf, r=rf, aux=aux, **kw)

        state, loss, key = train_step(state, key, x.float().cuda(), Fixed())
        mse_h = aux["per_example"].double().mean().item() / D
        mse_hip.append(mse_h); mse_ref.append(mse_r)
        assert abs(loss.item() - loss_ref) < 2e-4 * max(1.0, abs(loss_ref)), (step, loss.item(), loss_ref)
        assert abs(mse_h - mse_r) < 2e-3 * mse_r, (step, mse_h, mse_r)
        _compare_params(state, p, p0, lr, step, gen, f"step {step}")
    assert state.step == steps
    print("small-shape trajectory, unweighted MSE  hip:", [f"{a:.5f}" for a in mse_hip], " oracle:", [f"{a:.5f}" for a in mse_ref])


@pytest.mark.parametrize("B", [5, 40])
def test_steps_are_bitwise_reproducible(B):
    """Fixed-order reductions (split-K slabs, GRN statistic / weight-gradient records, per-example loss sums, the bias
    column sums inside the fused weight-gradient GEMM): the same three bf16 steps from the same state give bit-identical
    parameters, moments and losses.  B = 40: more than 32 rows, so the fused schedule takes its bias gradients from the
    GEMM (mfc_gemm_adamw's colsum output) -- reproducible as well, and equal to the sequential schedule's mfc_colsum to
    rounding only, hence no cross-schedule bitwise check at that batch."""
    from meanflow_audio_codec_amd.models import ConditionalConvFlow, TrainState, adamw
    from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, PRNGKey, train_step
    D, CD, LAT, NB = 2496, 128, 24, 2               # s = 49: ragged tiles, several tiles per workgroup range
    runs = []
    for rep in range(2):
        for fuse in (True, False):
            model = ConditionalConvFlow(D, CD, NB, LAT, dtype=torch.bfloat16)
            state = TrainState.create(apply_fn=model.apply, params=model.init(seed=5), tx=adamw(1e-3, 1e-2), model=model)
            x = torch.randn(B, D, generator=torch.Generator().manual_seed(2)).cuda()
            key = PRNGKey(11)
            losses = []
            for _ in range(3):
                state, loss, key = train_step(state, key, x, ImprovedMeanFlowLoss(), overlap=False, fuse=fuse)
                losses.append(loss.item())
            torch.cuda.synchronize()
            runs.append((fuse, losses, {k: v.clone() for k, v in state.params.items()},
                         {k: v.clone() for k, v in state.opt_state["nu"].items()}))
    for fuse in (True, False):
        a, b = [r for r in runs if r[0] == fuse]
        assert a[1] == b[1], (fuse, a[1], b[1])
        for k in a[2]:
            assert torch.equal(a[2][k], b[2][k]) and torch.equal(a[3][k], b[3][k]), (fuse, k)
    # the fused schedule (AdamW in the weight-gradient GEMM's epilogue) is bit-identical to gemm -> adamw as well
    f, u = [r for r in runs if r[0]][0], [r for r in runs if not r[0]][0]
    if B <= 32:
        assert f[1] == u[1]
        for k in f[2]:
            assert torch.equal(f[2][k], u[2][k]), k
    else:
        assert max(abs(a - b) for a, b in zip(f[1], u[1])) < 1e-3


def test_ci_shape_oracle_parity_and_trajectory():
    """CI shape, fp32 storage, B = 2: u, du/dt, loss and gradients of the first update against the fp64 oracle (the
    N-streaming / split-K GEMM paths with N, K in the 10^5..10^6 range, ragged 179 x 179 images, persistent ConvNeXt
    workgroups), then one more update followed with the unweighted error and sampled parameters (each fp64 oracle step
    costs ~100 s of host time; the five-update trajectory runs at the small shape above)."""
    from meanflow_audio_codec_amd.models import ConditionalConvFlow, TrainState, adamw
    from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, PRNGKey, train_step
    D, CD, LAT, NB, B, lr, wd, steps = 32256, 128, 256, 8, 2, 1e-4, 1e-4, 2
    t_start = time.time()
    model = ConditionalConvFlow(D, CD, NB, LAT, dtype=torch.float32)
    params = model.init(seed=42, device="cuda")
    for k, pv in params.items():        # give the block interior weight (layer scale 1e-6 / GRN affine 0 at init)
        if k.endswith("layer_scale_gamma"):
            pv.fill_(0.3)
        elif k.endswith("GlobalResponseNormalization_0/gamma"):
            pv.fill_(0.1)
    state = TrainState.create(apply_fn=model.apply, params=params, tx=adamw(lr, wd), model=model)
    p64 = {k: v.double().cpu() for k, v in state.params.items()}
    p0 = {k: v.clone() for k, v in p64.items()}
    p, m, v = _oracle_state(p64)
    g = torch.Generator().manual_seed(7)
    x = 0.1 * torch.randn(B, D, generator=g, dtype=torch.float64)
    gen = torch.Generator().manual_seed(3)
    key = PRNGKey(0)
    mse_hip, mse_ref = [], []
    for step in range(1, steps + 1):
        e = torch.randn(B, D, generator=g, dtype=torch.float64)
        t = torch.tensor([[0.8], [0.45]], dtype=torch.float64)
        r = torch.tensor([[0.3], [0.45]], dtype=torch.float64)          # row 1: r == t (no tangent pass)
        ef, tf, rf = e.float().cuda(), t.float().cuda(), r.float().cuda()
        aux = {}

        class Fixed(ImprovedMeanFlowLoss):
            def compute_loss(self, st, k_, xx, **kw):
                kw.pop("row0", None); kw.pop("global_batch", None); kw.pop("row_stride", None)
                return super().compute_loss(st, k_, xx, e=ef, t=tf, r=rf, aux=aux, **kw)

        if step == 1:
            # first update: everything the step computes, before any parameter moves
            loss_h, grads_h = Fixed().compute_loss(state, key, x.float().cuda())
            ref = fo.imf_loss(fo.conv_flow_apply, fo.conv_flow_encode, fo.unflatten(p), x, e, t, r)
            loss_r, g_r, a_r = ref[0], fo.flatten(ref[1]), ref[2]
            rel = lambda a, b: ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()
            assert abs(loss_h.item() - loss_r.item()) < 2e-4 * max(1.0, abs(loss_r.item()))
            assert rel(aux["u"], a_r["u"]) < 5e-4, rel(aux["u"], a_r["u"])
            assert rel(aux["dudt"], a_r["dudt"][:1]) < 2e-3, rel(aux["dudt"], a_r["dudt"][:1])
            worst = 0.0
            for k, gr in g_r.items():
                scale = gr.abs().max().item()
                if scale == 0:
                    continue
                idx = _sample_idx(gr.numel(), 200000, gen)
                gh = grads_h[k].reshape(-1)[idx.cuda()].double().cpu()
                err = (gh - gr.reshape(-1)[idx]).abs().max().item() / scale
                worst = max(worst, err)
                assert err < 5e-3, (k, err)
            print(f"CI shape: loss hip {loss_h.item():.6f} oracle {loss_r.item():.6f}; worst sampled gradient error {worst:.2e}"
                  f" ({time.time() - t_start:.0f} s)")
            del g_r, a_r, grads_h
        else:
            ref = None
        loss_ref, mse_r, _ = _oracle_update(p, m, v, x, e, t, r, step, lr, wd, pre=ref)
        ref = None
        state, loss, key = train_step(state, key, x.float().cuda(), Fixed())
        mse_h = aux["per_example"].double().mean().item() / D
        mse_hip.append(mse_h); mse_ref.append(mse_r)
        assert abs(loss.item() - loss_ref) < 5e-4 * max(1.0, abs(loss_ref)), (step, loss.item(), loss_ref)
        assert abs(mse_h - mse_r) < 5e-3 * mse_r, (step, mse_h, mse_r)
        _compare_params(state, p, p0, lr, step, gen, f"CI step {step}")
    print("CI-shape trajectory, unweighted MSE  hip:", [f"{a:.5f}" for a in mse_hip], " oracle:",
          [f"{a:.5f}" for a in mse_ref], f"({time.time() - t_start:.0f} s)")
