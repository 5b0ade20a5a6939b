"""GPU parity: ConditionalConvFlow passes and the loss strategies vs the fp64 oracle."""
import pytest
import torch

from oracle import flow_oracle as fo

pytestmark = pytest.mark.gpu

D, CD, LAT, NB = 400, 128, 24, 2      # s = 20, C = 16, S = 6400


def _make(dtype, seed=0, special=False):
    from meanflow_audio_codec_amd.models import ConditionalConvFlow, TrainState, adamw
    model = ConditionalConvFlow(D, CD, NB, LAT, dtype=dtype)
    shapes = fo.conv_flow_shapes(D, CD, LAT, NB, latent_dim=LAT)
    p64 = fo.init_params(shapes, seed=seed, special=special)
    flat = {k: v.float().cuda().contiguous() for k, v in fo.flatten(p64).items()}
    assert set(flat) == set(model.param_shapes()), set(flat) ^ set(model.param_shapes())
    for k, shp in model.param_shapes().items():
        assert tuple(flat[k].shape) == tuple(shp), k
    state = TrainState.create(apply_fn=model.apply, params=flat, tx=adamw(1e-3, 1e-2), model=model)
    # the oracle sees the parameters the kernels see (bf16-rounded big kernels in bf16 mode)
    pq = fo.unflatten({k: state.work[k].double().cpu() for k in flat})
    return model, state, pq


def _rel(a, b):
    return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 5e-2)])
def test_apply_matches_oracle(dtype, tol):
    model, state, pq = _make(dtype)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(5, D, generator=g)
    time = torch.rand(5, 2, generator=g)
    lat = torch.randn(5, LAT, generator=g)
    ref = fo.conv_flow_apply(pq, x.to(dtype).double(), time.double(), lat.double())
    out = model.apply({"params": state.work}, x.cuda(), time.cuda(), lat.cuda())
    assert _rel(out, ref) < tol
    ref0 = fo.conv_flow_apply(pq, x.to(dtype).double(), time.double(), None)
    out0 = model.apply({"params": state.work}, x.cuda(), time.cuda(), None)
    assert _rel(out0, ref0) < tol
    enc = model.apply({"params": state.work}, x.cuda(), method="encode")
    assert _rel(enc, fo.conv_flow_encode(pq, x.to(dtype).double())) < tol


def _draws(B, seed=3):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, D, generator=g)
    e = torch.randn(B, D, generator=g)
    t, r = fo.sample_tr_from_normals(torch.randn(B, 1, generator=g, dtype=torch.float64),
                                     torch.randn(B, 1, generator=g, dtype=torch.float64))
    return x, e, t.float(), r.float()


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 2e-4, 2e-3), (torch.bfloat16, 5e-2, 0.15)])
def test_improved_mean_flow_loss_and_grads(dtype, tol, gtol):
    from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, PRNGKey
    model, state, pq = _make(dtype)
    x, e, t, r = _draws(6)
    loss_ref, g_ref, aux_ref = fo.imf_loss(fo.conv_flow_apply, fo.conv_flow_encode, pq, x.double(), e.double(),
                                           t.double(), r.double())
    aux = {}
    loss, grads = ImprovedMeanFlowLoss().compute_loss(state, PRNGKey(0), x.cuda(), e=e.cuda(), t=t.cuda(),
                                                      r=r.cuda(), aux=aux)
    assert aux["n_tan"] == 3
    assert _rel(aux["u"], aux_ref["u"]) < tol
    assert abs(loss.item() - loss_ref.item()) < tol * max(1.0, abs(loss_ref.item()))
    gr = fo.flatten(g_ref)
    errs = {k: _rel(grads[k], gr[k]) for k in gr if gr[k].abs().max() > 0}
    bad = {k: v for k, v in errs.items() if not v < gtol}
    assert not bad, bad
    # reference property test/test_improved_mean_flow.py:31-54: t == r  =>  v_pred == u  (no tangent rows)
    aux2 = {}
    loss2, _ = ImprovedMeanFlowLoss().compute_loss(state, PRNGKey(0), x.cuda(), e=e.cuda(), t=t.cuda(), r=t.cuda(), aux=aux2)
    assert aux2["n_tan"] == 0 and aux2["dudt"] is None
    loss2_ref, _, aux2_ref = fo.imf_loss(fo.conv_flow_apply, fo.conv_flow_encode, pq, x.double(), e.double(), t.double(),
                                         t.double())
    assert torch.equal(aux2_ref["v_pred"], aux2_ref["u"])                       # the oracle has the property
    assert _rel(aux2["u"], aux2_ref["v_pred"]) < tol                            # ... and the HIP path's v_pred IS its u
    assert abs(loss2.item() - loss2_ref.item()) < tol * max(1.0, abs(loss2_ref.item()))


def test_merged_two_pass_schedule_equals_the_plain_one(monkeypatch):
    """``model.forward_imf`` (the r == t rows ride along with the boundary velocity pass; the tangent pass runs on the
    tangent rows alone) against the row-stacked schedule of ``_run``: the same arithmetic per row, other launch shapes
    -- loss, u, du/dt, v and every gradient agree to fp32 rounding; rows in any order (explicit t, r), and the sampled
    prefix rule."""
    from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, PRNGKey
    model, state, pq = _make(torch.float32, seed=6)
    x, e, t, r = _draws(7, seed=15)
    r = r.clone()
    r[4] = t[4]; r[5] = t[5]                       # r == t rows scattered through the batch (besides _draws' prefix)
    r[0] = 0.5 * t[0]                              # ... and a tangent row in front
    outs = {}
    for merge in ("1", "0"):
        monkeypatch.setenv("MFC_IMF_MERGE", merge)
        aux = {}
        loss, grads = ImprovedMeanFlowLoss().compute_loss(state, PRNGKey(0), x.cuda(), e=e.cuda(), t=t.cuda(), r=r.cuda(), aux=aux)
        outs[merge] = (loss.item(), {k: v.clone() for k, v in grads.items()}, {k: (v.clone() if torch.is_tensor(v) else v)
                                                                               for k, v in aux.items()})
        assert bool(aux.get("tangent_rows_last", False)) == (merge == "1")
    (l1, g1, a1), (l0, g0, a0) = outs["1"], outs["0"]
    assert a1["n_tan"] == a0["n_tan"] == int((t != r).sum().item()) and 0 < a1["n_tan"] < 7
    assert abs(l1 - l0) < 1e-6 * max(1.0, abs(l0))
    for k in ("u", "dudt", "v", "per_example"):
        assert _rel(a1[k], a0[k].double().cpu()) < 2e-5, k
    bad = {k: _rel(g1[k], g0[k].double().cpu()) for k in g0 if g0[k].abs().max() > 0 and not _rel(g1[k], g0[k].double().cpu()) < 2e-4}
    assert not bad, bad
    # sampled times: the r == t rows are a prefix, no permutation in the merged schedule; both schedules draw the same noise
    outs = {}
    for merge in ("1", "0"):
        monkeypatch.setenv("MFC_IMF_MERGE", merge)
        aux = {}
        loss, grads = ImprovedMeanFlowLoss().compute_loss(state, PRNGKey(3), x.cuda(), aux=aux)
        outs[merge] = (loss.item(), {k: v.clone() for k, v in grads.items()}, aux)
    assert outs["1"][2]["perm"] is None and outs["1"][2]["n_tan"] == outs["0"][2]["n_tan"] == 7 - int(7 * 0.5)
    assert abs(outs["1"][0] - outs["0"][0]) < 1e-6 * max(1.0, abs(outs["0"][0]))
    assert _rel(outs["1"][2]["u"], outs["0"][2]["u"].double().cpu()) < 2e-5
    bad = {k: _rel(outs["1"][1][k], outs["0"][1][k].double().cpu()) for k in outs["0"][1]
           if outs["0"][1][k].abs().max() > 0 and not _rel(outs["1"][1][k], outs["0"][1][k].double().cpu()) < 2e-4}
    assert not bad, bad


def test_jvp_matches_reverse_mode_property():
    """test/test_improved_mean_flow.py:57-100 restated on the HIP passes (fp32, 1e-4 relative):
    sum(dudt) for tangent (v, 1, 0) == <grad_z sum(u), v> + sum(grad_t sum(u))."""
    from meanflow_audio_codec_amd import ops
    model, state, pq = _make(torch.float32, seed=2)
    w = state.work
    g = torch.Generator().manual_seed(2)
    B = 3
    z = torch.randn(B, D, generator=g).cuda()
    t = torch.rand(B, 1, generator=g).cuda()
    r = 0.5 * t
    v = torch.randn(B, D, generator=g)
    v = (v / v.norm()).cuda()
    cond, cdot = model.conditioning(w, t, t - r, None, want_dot=True)
    u, dudt, ctx = model.forward(w, z, cond, xdot=v, cond_dot=cdot, save=True)
    lhs = dudt.double().sum().item()
    grads = state.grad_buffers()
    dz, dcond, _ = model.backward(w, ctx, torch.ones_like(u), grads)
    rhs = (dz.double() * v.double()).sum().item() + (dcond.double() * cdot.double()).sum().item()
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 2e-4, 2e-3)])
def test_flow_matching_and_mean_flow_losses(dtype, tol, gtol):
    from meanflow_audio_codec_amd.trainers import FlowMatchingLoss, MeanFlowLoss, PRNGKey
    model, state, pq = _make(dtype, seed=4)
    x, e, t, r = _draws(4, seed=8)
    loss_ref, g_ref, _ = fo.fm_loss(fo.conv_flow_apply, fo.conv_flow_encode, pq, x.double(), e.double(), t.double())
    loss, grads = FlowMatchingLoss().compute_loss(state, PRNGKey(0), x.cuda(), e=e.cuda(), t=t.cuda())
    assert abs(loss.item() - loss_ref.item()) < tol
    gr = fo.flatten(g_ref)
    bad = {k: _rel(grads[k], gr[k]) for k in gr if gr[k].abs().max() > 0 and not _rel(grads[k], gr[k]) < gtol}
    assert not bad, bad
    loss_ref, g_ref, _ = fo.mf_loss(fo.conv_flow_apply, fo.conv_flow_encode, pq, x.double(), e.double(), t.double(),
                                    r.double())
    loss, grads = MeanFlowLoss().compute_loss(state, PRNGKey(0), x.cuda(), e=e.cuda(), t=t.cuda(), r=r.cuda())
    assert abs(loss.item() - loss_ref.item()) < tol
    gr = fo.flatten(g_ref)
    bad = {k: _rel(grads[k], gr[k]) for k in gr if gr[k].abs().max() > 0 and not _rel(grads[k], gr[k]) < gtol}
    assert not bad, bad


def test_train_step_updates_like_oracle():
    from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, PRNGKey, train_step
    model, state, pq = _make(torch.float32, seed=6)
    x, e, t, r = _draws(4, seed=9)
    _, g_ref, _ = fo.imf_loss(fo.conv_flow_apply, fo.conv_flow_encode, pq, x.double(), e.double(), t.double(),
                              r.double())
    strat = ImprovedMeanFlowLoss()
    before = {k: v.clone() for k, v in state.params.items()}

    class Fixed(ImprovedMeanFlowLoss):
        def compute_loss(self, state, key, xx, **kw):
            return super().compute_loss(state, key, xx, e=e.cuda(), t=t.cuda(), r=r.cuda())
    key = PRNGKey(0)
    state, loss, key2 = train_step(state, key, x.cuda(), Fixed())
    assert key2.counter == key.counter + 1 and state.step == 1
    gr = fo.flatten(g_ref)
    for k in ("blocks_0/input_proj2/kernel", "blocks_1/conv_block/Conv_0/kernel", "blocks_1/output_proj2/bias",
              "latent_proj/kernel", "encoder/dense1/kernel"):
        p0 = before[k].double().cpu()
        pn, _, _ = fo.adamw_step(p0, gr[k], torch.zeros_like(p0), torch.zeros_like(p0), 1, 1e-3, 1e-2)
        # step 1 of Adam moves every weight by ~lr*sign(g): compare where |g| is not tiny
        mask = gr[k].abs() > 1e-3 * gr[k].abs().max()
        assert ((state.params[k].double().cpu() - pn)[mask].abs().max() < 2e-5), k


def test_data_parallel_shards_sum_to_global_batch():
    """Two shards (row0 / global_batch) give gradients and losses that SUM to the full-batch step:
    what the RCCL all-reduce of distributed.GradReducer relies on."""
    from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, PRNGKey
    model, state, pq = _make(torch.float32, seed=11)
    x, e, t, r = _draws(8, seed=12)
    strat = ImprovedMeanFlowLoss()
    loss_full, grads = strat.compute_loss(state, PRNGKey(0), x.cuda(), e=e.cuda(), t=t.cuda(), r=r.cuda())
    full = {k: v.double().clone() for k, v in grads.items()}
    acc = {k: torch.zeros_like(v) for k, v in full.items()}
    loss_sum = 0.0
    for sl in (slice(0, 4), slice(4, 8)):
        l, g = strat.compute_loss(state, PRNGKey(0), x[sl].cuda(), e=e[sl].cuda(), t=t[sl].cuda(), r=r[sl].cuda(),
                                  row0=sl.start, global_batch=8)
        loss_sum += l.item()
        for k in acc:
            acc[k] += g[k].double()
    assert abs(loss_sum - loss_full.item()) < 1e-5
    for k in full:
        if full[k].abs().max() > 0:
            assert ((acc[k] - full[k]).abs().max() / full[k].abs().max()).item() < 2e-3, k


@pytest.mark.parametrize("world,prop", [(2, 0.5), (4, 0.5), (2, 0.7)])
def test_interleaved_shards_sum_to_global_batch_sampled(world, prop):
    """The SAMPLED path ((e, t, r) drawn from Philox inside the step) under the interleaved ownership of
    distributed.shard_rows (rank k owns global rows k, k+G, ...): shard losses and gradients sum to the
    full-batch step, every shard sees the same number (+-1) of r == t rows, and (t != r).sum() == n_tan with
    data_size = int(B * p) computed once on the host (B*p integer with p not exact in f32)."""
    from meanflow_audio_codec_amd.distributed import shard_of, shard_rows
    from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, MeanFlowTimeSampling, PRNGKey
    model, state, pq = _make(torch.float32, seed=11)
    Bg = 10 if prop == 0.7 else 8
    g = torch.Generator().manual_seed(31)
    x = torch.randn(Bg, D, generator=g).cuda()
    strat = ImprovedMeanFlowLoss(time_sampling=MeanFlowTimeSampling(-0.4, 1.0, prop))
    key = PRNGKey(7, 3)
    aux = {}
    loss_full, grads = strat.compute_loss(state, key, x, aux=aux)
    assert aux["n_tan"] == Bg - int(Bg * prop) == int((aux["t"] != aux["r"]).sum().item())
    full = {k: v.double().clone() for k, v in grads.items()}
    acc = {k: torch.zeros_like(v) for k, v in full.items()}
    loss_sum, n_tans = 0.0, []
    if Bg % world:
        pytest.skip("global batch not divisible")
    for rank in range(world):
        a = {}
        l, gr = strat.compute_loss(state, key, shard_of(x, rank, world), aux=a, **shard_rows(rank, world, Bg // world))
        # the shard drew exactly the (t, r) of the global rows it owns
        assert torch.equal(a["t"], aux["t"][rank::world]) and torch.equal(a["r"], aux["r"][rank::world])
        assert a["n_tan"] == int((a["t"] != a["r"]).sum().item())
        n_tans.append(a["n_tan"])
        loss_sum += l.item()
        for k in acc:
            acc[k] += gr[k].double()
    assert max(n_tans) - min(n_tans) <= 1 and sum(n_tans) == aux["n_tan"], n_tans
    assert abs(loss_sum - loss_full.item()) < 1e-5
    for k in full:
        if full[k].abs().max() > 0:
            assert ((acc[k] - full[k]).abs().max() / full[k].abs().max()).item() < 2e-3, k


def test_overlapped_train_step_matches_sequential():
    """overlap=True (per-block AdamW on a side stream during the reverse pass) == overlap=False, bit for bit: every
    reduction is fixed-order, so the schedule cannot change a result."""
    from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, PRNGKey, train_step
    results = []
    for overlap in (False, True):
        model, state, pq = _make(torch.float32, seed=21)
        x, e, t, r = _draws(4, seed=22)
        key = PRNGKey(5)
        for _ in range(3):
            state, loss, key = train_step(state, key, x.cuda(), ImprovedMeanFlowLoss(), overlap=overlap, fuse=False)
        torch.cuda.synchronize()
        results.append(({k: v.clone() for k, v in state.params.items()}, loss.item(), state.step))
    (pa, la, sa), (pb, lb, sb) = results
    assert sa == sb == 3 and la == lb
    for k in pa:
        assert torch.equal(pa[k], pb[k]), (k, (pa[k] - pb[k]).abs().max().item())


def test_fused_single_gpu_step_matches_sequential():
    """The default single-GPU schedule (big kernels updated inside their weight-gradient GEMM, mfc_gemm_adamw) against
    compute_loss -> apply_gradients from the same state.  The fused kernel is bit-identical to gemm -> adamw
    (tests/test_gemm_gpu.py, and whole steps in tests/test_trajectory_gpu.py::test_steps_are_bitwise_reproducible); at
    this batch size the bias gradients of the big layers may come from the fused kernel's column sums, which add in
    another order than mfc_colsum.  So: same losses, same first moments of every leaf to rounding, bf16 working copies
    consistent with their masters, runs stay together."""
    from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, PRNGKey, train_step
    runs = []
    for fuse in (False, True):
        model, state, pq = _make(torch.bfloat16, seed=31)
        x, e, t, r = _draws(4, seed=32)
        key = PRNGKey(9)
        state, loss, key = train_step(state, key, x.cuda(), ImprovedMeanFlowLoss(), overlap=False, fuse=fuse)
        torch.cuda.synchronize()
        mu1 = {k: v.clone() for k, v in state.opt_state["mu"].items()}
        losses = [loss.item()]
        for _ in range(2):
            state, loss, key = train_step(state, key, x.cuda(), ImprovedMeanFlowLoss(), overlap=False, fuse=fuse)
            losses.append(loss.item())
        torch.cuda.synchronize()
        runs.append((mu1, losses, state))
    (ma, la, sa), (mb, lb, sb) = runs
    assert sa.step == sb.step == 3 and max(abs(u - v) for u, v in zip(la, lb)) < 1e-4
    big = [k for k in ma if k.endswith("_proj1/kernel") or k.endswith("_proj2/kernel")]
    assert len(big) == 4 * model.num_blocks
    for k in ma:
        scale = ma[k].abs().max().item()
        if scale > 0:
            assert (ma[k] - mb[k]).abs().max().item() <= 1e-5 * scale, (k, (ma[k] - mb[k]).abs().max().item(), scale)
    for k in big:
        assert sb.work[k].dtype == torch.bfloat16 and torch.equal(sb.work[k], sb.params[k].bfloat16()), k
    for k in sa.params:     # lr = 1e-3, three steps: an element whose tiny gradient flips sign moves by 2 lr per step
        assert (sa.params[k] - sb.params[k]).abs().max().item() < 6.6e-3, k
        assert ((sa.params[k] - sb.params[k]).abs() > 1e-4).float().mean().item() < 0.05, k


def test_ci_shape_schedules_agree():
    """SURVEY 8(d)'s CI shape (T = 16384 -> D = 32256, s = 179, S = 512656, 1.12 B parameters, bf16, B = 8): a size
    where tiles are ragged, workgroups are persistent over many tiles, the K = D / K = S products are split-K (slabs
    summed in a fixed order) and the streamed operands exceed the caches.  Size-independent property:
    the fused single-GPU schedule and compute_loss -> apply_gradients see the same loss and, after one step from the
    same state, the same first moments (= 0.1 x the gradient) of all 32 big kernels up to the bf16 rounding of a
    gradient element, and every parameter moved by at most one Adam step."""
    from meanflow_audio_codec_amd.models import ConditionalConvFlow, TrainState, adamw
    from meanflow_audio_codec_amd.preprocessing.tokenization import MDCTTokenization
    from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, PRNGKey, train_step
    T, B, lr = 16384, 8, 1e-4
    tok = MDCTTokenization(window_size=512, hop_size=256)
    g = torch.Generator(device="cuda").manual_seed(42)
    clips = 0.1 * torch.randn(B, T, generator=g, device="cuda")
    x = tok.tokenize(clips).reshape(B, -1)
    assert x.shape[1] == 32256
    snaps, losses = [], []
    for fuse in (False, True):
        model = ConditionalConvFlow(32256, 128, 8, 256, dtype=torch.bfloat16)
        state = TrainState.create(apply_fn=model.apply, params=model.init(seed=3), tx=adamw(lr, 1e-4), model=model)
        p0 = {k: v.clone() for k, v in state.params.items() if k.endswith("_proj1/kernel") or k.endswith("_proj2/kernel")}
        state, loss, _ = train_step(state, PRNGKey(42), x, ImprovedMeanFlowLoss(), overlap=False, fuse=fuse)
        torch.cuda.synchronize()
        losses.append(loss.item())
        snaps.append({k: (state.opt_state["mu"][k], state.params[k], state.work[k]) for k in p0})
        for k, v in p0.items():
            step = (state.params[k] - v).abs().max().item()
            assert 0 < step <= lr * 1.01 + 1e-4 * lr * v.abs().max().item() + 1e-7, (k, step)
            assert torch.equal(state.work[k], state.params[k].bfloat16()), k
        del state, model, p0
    assert len(snaps[0]) == 32 and all(l == l and abs(l) < 1e6 for l in losses)
    assert abs(losses[0] - losses[1]) <= 1e-5 * abs(losses[0])
    for k in snaps[0]:
        ma, mb = snaps[0][k][0], snaps[1][k][0]
        scale = ma.abs().max().item()
        assert scale > 0 and (ma - mb).abs().max().item() <= 3e-2 * scale, (k, (ma - mb).abs().max().item(), scale)


def test_use_grn_false_matches_oracle():
    """ConvNeXtBlock(use_grn=False) (models/conv_flow.py:91-92): no GlobalResponseNormalization parameters, no
    statistics pass -- forward, iMF loss and every gradient against the oracle without the GRN."""
    from meanflow_audio_codec_amd.models import ConditionalConvFlow, TrainState, adamw
    from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, PRNGKey
    model = ConditionalConvFlow(D, CD, NB, LAT, use_grn=False, dtype=torch.float32)
    shapes = fo.conv_flow_shapes(D, CD, LAT, NB, latent_dim=LAT, use_grn=False)
    flat64 = fo.flatten(fo.init_params(shapes, seed=3, special=False))
    assert set(flat64) == set(model.param_shapes()) and not any("GlobalResponseNormalization" in k for k in flat64)
    flat = {k: v.float().cuda().contiguous() for k, v in flat64.items()}
    state = TrainState.create(apply_fn=model.apply, params=flat, tx=adamw(1e-3, 1e-2), model=model)
    pq = fo.unflatten({k: state.work[k].double().cpu() for k in flat})
    x, e, t, r = _draws(5, seed=8)
    g = torch.Generator().manual_seed(2)
    time = torch.rand(5, 2, generator=g)
    lat = torch.randn(5, LAT, generator=g)
    ref = fo.conv_flow_apply(pq, x.double(), time.double(), lat.double())
    out = model.apply({"params": state.work}, x.cuda(), time.cuda(), lat.cuda())
    assert _rel(out, ref) < 1e-4
    loss_ref, g_ref, aux_ref = fo.imf_loss(fo.conv_flow_apply, fo.conv_flow_encode, pq, x.double(), e.double(),
                                           t.double(), r.double())
    loss, grads = ImprovedMeanFlowLoss().compute_loss(state, PRNGKey(0), x.cuda(), e=e.cuda(), t=t.cuda(), r=r.cuda())
    assert abs(loss.item() - loss_ref.item()) < 2e-4 * max(1.0, abs(loss_ref.item()))
    for k, gr in fo.flatten(g_ref).items():
        scale = gr.abs().max().item()
        if scale > 0:
            assert ((grads[k].double().cpu() - gr).abs().max().item() / scale) < 2e-3, k


def test_cfg_sampling_matches_oracle():
    """evaluators/sampling.py:50-96 with classifier-free guidance (guidance_scale != 1: conditional and unconditional
    velocity blended in both Heun stages) on the ConvNeXt flow, against the oracle's heun_sample."""
    from meanflow_audio_codec_amd.evaluators.sampling import heun_integrate
    model, state, pq = _make(torch.float32, seed=13, special=False)
    g = torch.Generator().manual_seed(14)
    x0 = torch.randn(4, D, generator=g)
    lat = torch.randn(4, LAT, generator=g)
    for gs, n_steps in ((2.0, 2), (0.5, 3), (1.0, 2)):
        ref = fo.heun_sample(fo.conv_flow_apply, pq, x0.double(), lat.double(), n_steps, guidance_scale=gs)
        out = heun_integrate(model, state.work, x0.cuda(), lat.cuda(), n_steps, guidance_scale=gs)
        assert _rel(out, ref) < 2e-4, (gs, n_steps, _rel(out, ref))
