"""GPU, BASELINE config #4 at its full size (D = 392704, s = 626, S = 6 270 016, 8 blocks, 13.75 B parameters, bf16):
size-independent properties of the hot path, since no oracle finishes at this size.

* forward-mode tangent == reverse mode (the reference's own property, test/test_improved_mean_flow.py:57-100):
  <w, J v> == <J^T w, v> through all eight blocks -- exercises the N-streaming GEMMs, the fused LayerNorm epilogues, the
  ConvNeXt stats/apply tangent kernels and every reverse kernel at the literal spatial size.
* data-parallel additivity: the gradients and losses of two batch shards (row0 / global_batch) sum to the
  full-batch step, checked through random +-1 linear functionals of every big kernel's gradient.

One process holds the whole train state (179 GiB), as bench.py does; small batches keep the rest short (~1 min)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

D, CD, LAT, NB = 392704, 128, 256, 8


@pytest.fixture(scope="module")
def literal_state():
    from meanflow_audio_codec_amd.models import ConditionalConvFlow, TrainState, adamw
    import gc
    gc.collect()
    torch.cuda.empty_cache()          # blocks cached by earlier tests are not "free" to mem_get_info
    free, total = torch.cuda.mem_get_info()
    if free < 230 * 2 ** 30:
        pytest.skip(f"needs ~230 GiB of free HBM, found {free / 2**30:.0f} GiB")
    model = ConditionalConvFlow(D, CD, NB, LAT, dtype=torch.bfloat16)
    params = model.init(seed=1, device="cuda")
    # layer scale is initialised to 1e-6 and the GRN affine to 0 (models/conv_flow.py:56-63,109): every block would be
    # an identity around its ConvNeXt interior.  Give the interior weight so the properties test it.
    for k, p in params.items():
        if k.endswith("layer_scale_gamma"):
            p.fill_(0.3)
        elif k.endswith("GlobalResponseNormalization_0/gamma"):
            p.fill_(0.1)
    state = TrainState.create(apply_fn=model.apply, params=params, tx=adamw(1e-4, 1e-4), model=model)
    yield model, state
    del state, params
    model.release_workspace()
    torch.cuda.empty_cache()


def test_literal_shape_tangent_equals_reverse_mode(literal_state):
    model, state = literal_state
    w = state.work
    g = torch.Generator(device="cuda").manual_seed(2)
    B = 3
    z = 0.5 * torch.randn(B, D, generator=g, device="cuda")
    t = torch.rand(B, 1, generator=g, device="cuda")
    r = 0.5 * t
    v = torch.randn(B, D, generator=g, device="cuda")
    v = v / v.norm()
    lat = torch.randn(B, LAT, generator=g, device="cuda")
    zb, vb = z.bfloat16(), v.bfloat16()
    cond, cdot = model.conditioning(w, t, t - r, lat, want_dot=True)
    u, dudt, ctx = model.forward(w, zb, cond, xdot=vb, cond_dot=cdot, latents=lat, save=True)
    assert u.shape == (B, D) and torch.isfinite(u.float()).all() and torch.isfinite(dudt.float()).all()
    norm = dudt.double().norm().item()
    assert norm > 0
    wgt = (dudt.float() / norm).to(u.dtype)                       # a cotangent aligned with the tangent: <w, Jv> = |Jv|
    lhs = (wgt.double() * dudt.double()).sum().item()
    grads = state.grad_buffers()
    dz, dcond, _ = model.backward(w, ctx, wgt.contiguous(), grads)
    rhs = (dz.double() * vb.double()).sum().item() + (dcond.double() * cdot.double()).sum().item()
    # Why 3 % and not fp32-tight: the two sides are DIFFERENT computations (forward-mode through the tangent kernels,
    # reverse-mode through the gradient kernels), each storing every activation between its ~80 kernels in bf16
    # (2^-9 relative per rounding); the pairing is a sum of 1.2 M positive-ish terms whose rounding errors do not cancel
    # between the two paths.  Bitwise reproducibility bounds run-to-run noise (zero), not this algorithmic difference;
    # the fp32-storage version of the same property holds to 1e-4 (tests/test_conv_flow_gpu.py, test_mlp_flow_gpu.py).
    print(f"literal tangent/reverse pairing: lhs={lhs:.6g} rhs={rhs:.6g} rel={(lhs - rhs) / lhs:.3e}")
    assert lhs > 0.5 * norm
    assert abs(lhs - rhs) < 3e-2 * lhs, (lhs, rhs)


def test_literal_shape_shards_sum_to_global_batch(literal_state):
    from meanflow_audio_codec_amd.trainers import ImprovedMeanFlowLoss, PRNGKey
    model, state = literal_state
    g = torch.Generator(device="cuda").manual_seed(5)
    B = 4
    x = 0.1 * torch.randn(B, D, generator=g, device="cuda")
    e = torch.randn(B, D, generator=g, device="cuda")
    t = torch.tensor([[0.9], [0.6], [0.5], [0.3]], device="cuda")
    r = torch.tensor([[0.4], [0.1], [0.5], [0.3]], device="cuda")        # rows 2, 3: r == t (no tangent pass)
    strat = ImprovedMeanFlowLoss()
    big = [k for k, p in state.params.items() if p.numel() > (1 << 24)]
    assert len(big) == 4 * NB + 0 or len(big) >= 4 * NB                  # four big kernels per block (+ encoder)
    probes = {}

    def probe_of(p):
        key = p.numel()
        if key not in probes:
            probes[key] = torch.empty(key, dtype=torch.bfloat16, device="cuda").bernoulli_(0.5, generator=g).mul_(2).sub_(1)
        return probes[key]

    def fingerprint(grads):
        out = {}
        for k in big:
            gk = grads[k].reshape(-1)
            out[k] = ((gk.float() * probe_of(gk).float()).sum().double().item(), gk.float().norm().item())
        small = {k: v.double().clone() for k, v in grads.items() if k not in set(big)}
        return out, small

    loss_full, grads = strat.compute_loss(state, PRNGKey(0), x, e=e, t=t, r=r)
    fp_full, small_full = fingerprint(grads)
    # What bitwise-reproducible kernels DO allow at this size: the same launch sequence twice gives identical bits -- loss,
    # every functional of every big gradient, every small leaf (no tolerance).  The tolerances further down compare
    # DIFFERENT launch sequences (other row counts -> other tile variants and split-K slice counts) and are explained there.
    loss_again, grads_again = strat.compute_loss(state, PRNGKey(0), x, e=e, t=t, r=r)
    fp_again, small_again = fingerprint(grads_again)
    assert loss_again.item() == loss_full.item()
    assert fp_again == fp_full
    assert all(torch.equal(small_again[k], small_full[k]) for k in small_full)
    # ... and six more times: the one non-repeatable evaluation ever seen here (round 3: a counted LDS-DMA wait in the ConvNeXt
    # tile kernels) showed up in about one evaluation of fifty; tools/probe_step_determinism.py is the long screen
    for _ in range(6):
        loss_k, grads_k = strat.compute_loss(state, PRNGKey(0), x, e=e, t=t, r=r)
        fp_k, small_k = fingerprint(grads_k)
        assert loss_k.item() == loss_full.item() and fp_k == fp_full
        assert all(torch.equal(small_k[k], small_full[k]) for k in small_full)
    acc = {k: 0.0 for k in big}
    small_acc = {k: torch.zeros_like(v) for k, v in small_full.items()}
    loss_sum = 0.0
    for sl in (slice(0, 2), slice(2, 4)):
        l, gs = strat.compute_loss(state, PRNGKey(0), x[sl].contiguous(), e=e[sl].contiguous(), t=t[sl].contiguous(),
                                   r=r[sl].contiguous(), row0=sl.start, global_batch=B)
        loss_sum += l.item()
        fp, sm = fingerprint(gs)
        for k in big:
            acc[k] += fp[k][0]
        for k in small_acc:
            small_acc[k] += sm[k]
    assert abs(loss_sum - loss_full.item()) < 1e-3 * max(1.0, abs(loss_full.item()))
    nonzero = 0
    worst = max(abs(acc[k] - fp_full[k][0]) / fp_full[k][1] for k in big if fp_full[k][1] > 0)
    print(f"literal shard additivity: loss {loss_sum:.6f} vs {loss_full.item():.6f}; worst functional deviation {worst:.3e} |g|")
    for k in big:
        dot, norm = fp_full[k]
        if norm == 0:
            continue
        nonzero += 1
        # bf16 end to end.  A run is bitwise reproducible (fixed-order reductions), but the full batch and its shards are
        # different launches: other M -> other tile variants and split-K slice counts -> fp32 sums in another order -> a
        # few bf16 roundings flip, and after a few of the ~80 rounding stages of the eight blocks the difference saturates
        # at one bf16 ulp per element (u differs by ~0.6 % in norm, these +-1 functionals by 2-3 % of |g|_2; the same probe
        # in fp32 storage: 6e-6, tools/noise_probe.py).  A missing or doubled shard would show as ~50 %.  The tight version
        # of this property (2e-3, fp32) runs at the small shape in tests/test_conv_flow_gpu.py.
        assert abs(acc[k] - dot) < 6e-2 * norm, (k, acc[k], dot, norm)
    assert nonzero >= 4 * NB
    for k, v in small_full.items():
        scale = v.abs().max().item()
        if scale > 0:
            assert ((small_acc[k] - v).abs().max().item() / scale) < 5e-2, k
