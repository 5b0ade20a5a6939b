"""GPU: the hipGraph-captured decoder (``evaluators.GraphedDecoder``: noise -> 1-NFE decode or n-step Heun -> IMDCT).

north_star: "the low-NFE iMF sampler captured as a hipGraph".  Reference: ``evaluators/sampling.py:50-96`` (Heun, h = 0)
and the 1-NFE formula x0 = eps - u(eps, r=0, t=1) of
``documentation/research/improved_meanflow/improved_meanflow_key_eqn.md:313-316``.

* graph replay == the eager launch sequence for the same noise, BITWISE (every reduction on this path is
  fixed-order: split-K slabs and the GRN statistic partials are summed in a fixed order);
* replay vs the fp64 oracle (fp32 storage), IMDCT included, for the 1-NFE decode and for Heun n_steps = 2;
* two replays with fresh noise differ; a replay with the same noise reproduces itself.
"""
import numpy as np
import pytest
import torch

from oracle import flow_oracle as fo
from oracle import mdct_oracle as mo

pytestmark = pytest.mark.gpu

B, T, N, HOP = 3, 1280, 64, 32           # -> 39 frames x 64 = D 2496 (s = 49, S = 38416)
NF = (T - N) // HOP + 1
D, CD, LAT, NB = NF * N, 128, 16, 2


def _setup(dtype=torch.float32, seed=4):
    from meanflow_audio_codec_amd.models import ConditionalConvFlow, TrainState, adamw
    from meanflow_audio_codec_amd.preprocessing import MDCTConfig
    model = ConditionalConvFlow(D, CD, NB, LAT, dtype=dtype)
    p64 = fo.init_params(fo.conv_flow_shapes(D, CD, LAT, NB, latent_dim=LAT), seed=seed, special=False)
    flat = {k: v.float().cuda().contiguous() for k, v in fo.flatten(p64).items()}
    state = TrainState.create(apply_fn=model.apply, params=flat, tx=adamw(1e-3, 1e-2), model=model)
    pq = fo.unflatten({k: state.work[k].double().cpu() for k in flat})
    lat = torch.randn(B, LAT, generator=torch.Generator().manual_seed(seed + 1))
    return model, state, pq, lat, MDCTConfig(window_size=N, hop_size=HOP)


def _oracle_audio(pq, eps, lat, n_steps):
    e64, l64 = eps.double().cpu(), lat.double()
    if n_steps == 0:
        x0 = fo.one_step_decode(fo.conv_flow_apply, pq, e64, l64)
    else:
        x0 = fo.heun_sample(fo.conv_flow_apply, pq, e64, l64, n_steps)
    return x0, mo.imdct_f64(x0.reshape(B, NF, N).numpy(), N, HOP)


@pytest.mark.parametrize("n_steps", [0, 2])
def test_graph_replay_equals_eager_and_oracle(n_steps):
    from meanflow_audio_codec_amd.evaluators import GraphedDecoder, one_step_decode
    from meanflow_audio_codec_amd.evaluators.sampling import heun_integrate
    from meanflow_audio_codec_amd.preprocessing.mdct import imdct
    model, state, pq, lat, cfg = _setup()
    latd = lat.cuda()
    dec = GraphedDecoder(model, state.work, B, latd, n_steps=n_steps, token_shape=(NF, N), mdct_config=cfg, seed=9)
    a1 = dec(fresh_noise=True).clone()
    eps = dec.eps.clone()
    assert a1.shape == (B, (NF - 1) * HOP + 2 * N) and torch.isfinite(a1).all()
    # same noise -> the replay reproduces itself bit for bit
    a1b = dec(fresh_noise=False).clone()
    assert torch.equal(a1, a1b)
    # eager launch sequence on the same noise == the captured one, bitwise
    x0 = one_step_decode(model, state.work, eps, latd) if n_steps == 0 else heun_integrate(model, state.work, eps, latd, n_steps)
    eager = imdct(x0.reshape(B, NF, N).contiguous(), config=cfg)
    assert torch.equal(eager, a1), (eager - a1).abs().max().item()
    # vs the fp64 oracle, IMDCT included
    x0_ref, audio_ref = _oracle_audio(pq, eps, lat, n_steps)
    rel_x = ((x0.double().cpu() - x0_ref).abs().max() / x0_ref.abs().max()).item()
    rel_a = np.abs(a1.double().cpu().numpy() - audio_ref).max() / np.abs(audio_ref).max()
    assert rel_x < 2e-4 and rel_a < 2e-4, (rel_x, rel_a)
    # fresh noise: a different clip, and the noise stream advances by B rows per call
    a2 = dec(fresh_noise=True).clone()
    assert not torch.equal(dec.eps, eps) and (a2 - a1).abs().max().item() > 1e-3
    # the draw is a node of the graph: replay k decodes rows [k B, (k + 1) B) of the Philox stream (a1b re-used call 0)
    assert dec.noise_in_graph and dec.calls == 2
    assert torch.equal(eps, dec.noise_of_call(0)) and torch.equal(dec.eps, dec.noise_of_call(1))
    _, audio_ref2 = _oracle_audio(pq, dec.eps, lat, n_steps)
    assert np.abs(a2.double().cpu().numpy() - audio_ref2).max() / np.abs(audio_ref2).max() < 2e-4


def test_graphed_decoder_bf16_and_token_output():
    """bf16 storage (the benchmarked configuration) against the oracle evaluated on the bf16-rounded weights, and
    ``token_shape=None`` (no IMDCT: tokens out)."""
    from meanflow_audio_codec_amd.evaluators import GraphedDecoder
    model, state, pq, lat, cfg = _setup(torch.bfloat16, seed=6)
    dec = GraphedDecoder(model, state.work, B, lat.cuda(), n_steps=0, token_shape=None, seed=2)
    x0 = dec().clone()
    assert x0.shape == (B, D) and x0.dtype == torch.float32
    ref = fo.one_step_decode(fo.conv_flow_apply, pq, dec.eps.bfloat16().double().cpu(), lat.double())
    assert ((x0.double().cpu() - ref).abs().max() / ref.abs().max()).item() < 5e-2
    assert torch.equal(dec(fresh_noise=False), x0)


def test_capture_survives_pending_garbage_that_owns_device_resources():
    """Regression for the abort of round 2 ("Fatal Python error: Aborted ... Garbage-collecting" inside
    ``torch.cuda.graph``): the cyclic collector ran during stream capture and destroyed an unrelated object that owned
    HIP resources.  Here the trigger is built deterministically -- an UNREACHABLE reference cycle that owns an older,
    instantiated hipGraph and an event, still uncollected, with the collector's threshold at 1 so that the very next
    allocations would start a collection -- and fifty more such cycles become unreachable INSIDE the capture (at the
    first kernel call of the captured body).  ``GraphedDecoder`` collects before the capture and keeps the collector off
    during it, so the cycles die outside the capture: before it (the old one) and after it (the new ones)."""
    import gc
    import weakref

    from meanflow_audio_codec_amd import ops
    from meanflow_audio_codec_amd.evaluators import GraphedDecoder
    model, state, pq, lat, cfg = _setup()

    class Holder:
        pass

    def make_cycle():
        h = Holder()
        h.me = h                                   # reference cycle: only the cyclic collector can free it
        h.event = torch.cuda.Event()
        h.event.record()
        return h

    old = make_cycle()
    old.graph = torch.cuda.CUDAGraph()
    buf = torch.zeros(16, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.graph(old.graph):
        buf.add_(1.0)
    old.graph.replay()
    torch.cuda.synchronize()
    alive = weakref.ref(old)
    gc.collect()
    was = gc.get_threshold()
    inside = []
    real = ops.randn_dev
    state_in_capture = {}

    # cycles with recorded events (and one more instantiated graph), kept alive until the capture is under way
    pool = [make_cycle() for _ in range(50)]
    pool[0].graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(pool[0].graph):
        buf.add_(2.0)
    torch.cuda.synchronize()
    inside.extend(weakref.ref(h) for h in pool)

    def spy(*a, **kw):
        if torch.cuda.is_current_stream_capturing() and pool:
            state_in_capture["gc_enabled"] = gc.isenabled()
            pool.clear()                           # 50 unreachable cycles owning device resources, DURING the capture
            junk = [[i] for i in range(2000)]      # allocations: far beyond a threshold of 1, were the collector on
            del junk
            state_in_capture["collected_inside"] = sum(r() is None for r in inside)
        return real(*a, **kw)

    try:
        gc.disable()
        del old                                    # now unreachable and uncollected
        assert alive() is not None
        gc.set_threshold(1, 1, 1)
        gc.enable()
        ops.randn_dev = spy
        dec = GraphedDecoder(model, state.work, B, lat.cuda(), n_steps=0, token_shape=(NF, N), mdct_config=cfg, seed=1)
    finally:
        ops.randn_dev = real
        gc.set_threshold(*was)
        gc.enable()
    assert alive() is None                                         # collected by the constructor, before the capture
    assert state_in_capture == {"gc_enabled": False, "collected_inside": 0}
    gc.collect()
    assert all(r() is None for r in inside)                        # ... and the capture-time garbage afterwards
    a = dec().clone()
    assert torch.isfinite(a).all() and torch.equal(dec(fresh_noise=False), a)
