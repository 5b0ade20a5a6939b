#!/usr/bin/env python3
"""Golden vectors for the flow / loss / sampler ALGEBRA, produced by RUNNING the reference's own PyTorch
implementations ``meanflow_audio_codec/references/archive/{flow,mflow,imflow}.py`` in the build container.

These three files are the only part of the reference's flow path that imports here (torch only; the JAX path needs
jax/flax/optax).  They are loaded by file path with ``meanflow_audio_codec.datasets.mnist.load_mnist``
stubbed in ``sys.modules`` (the loader is only used by their ``init_training``).  Nothing of the reference is copied:
the script calls its classes and stores inputs and outputs.

What the vectors pin (``tests/test_oracle_archive.py`` checks ``oracle/flow_oracle.py`` against them):
  * flow matching (``flow.py:107-113``): interpolation z = (1-t) x + (nmin + nmax t) e, target nmax e - x, MSE;
    Heun sampler (``flow.py:115-124``) -- the same integrator as ``evaluators/sampling.py:50-96``;
  * MeanFlow (``mflow.py:128-157``): JVP tangent (e - x, 1, 0), u_tgt = v - clip(t-r,0,1) dudt, stop-gradient on the
    target, per-example MEAN squared error, adaptive weight 1/(d+c)^(1-gamma) -- the algebra of
    ``trainers/loss_strategies.py:141-201``;
  * improved MeanFlow (``imflow.py:125-168``): boundary pass v = u(z, t, t), JVP with tangent (v, 0, 1) on (z, t, r)
    [the JAX path uses (v, 1, 0), ``loss_strategies.py:263-267`` -- the oracle core takes the tangent as a parameter],
    V = u + (t-r) sg(dudt), MSE; two-time Heun sampler (``imflow.py:170-182``).
Still unpinned afterwards (no runnable reference): Flax initialisers, optax.adamw, JAX PRNG streams, and the ConvNeXt /
Mixer nets themselves (only their JVP == reverse-mode property and t = r boundary property are restated).

The archive nets differ from the JAX nets (SiLU, LayerNorm eps 1e-5, class embedding, 2 pi logspace frequencies,
[sin, cos] order): ``oracle/flow_oracle.py::archive_net`` restates THAT net so that the loss cores can be compared
number for number.  The random draws inside the reference's loss functions are recovered by replaying the same
``torch.manual_seed`` stream in the same order, and asserted to reproduce the reference's loss value.

    python tests/golden/gen_archive_flow_golden.py   ->  tests/golden/archive_flow_golden.npz
"""
from __future__ import annotations

import importlib.util
import pathlib
import sys
import types

import numpy as np
import torch

REF = pathlib.Path("/root/reference/meanflow_audio_codec/references/archive")
OUT = pathlib.Path(__file__).resolve().parent / "archive_flow_golden.npz"


def load(name):
    for mod in ("meanflow_audio_codec", "meanflow_audio_codec.datasets", "meanflow_audio_codec.datasets.mnist"):
        if mod not in sys.modules:
            m = types.ModuleType(mod)
            m.__path__ = []
            sys.modules[mod] = m
    sys.modules["meanflow_audio_codec.datasets.mnist"].load_mnist = lambda *a, **k: iter(())
    import matplotlib
    matplotlib.use("Agg")
    spec = importlib.util.spec_from_file_location(f"_ref_archive_{name}", REF / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def small_cfg(mod):
    return mod.Config(noise_dim=16, cond_dim=8, latent_dim=12, n_blocks=2, n_classes=3, batch_size=5, device="cpu")


def sd(model, out, prefix):
    for k, v in model.state_dict().items():
        out[f"{prefix}/param/{k}"] = v.detach().numpy().copy()


def grads(model, out, prefix):
    for k, p in model.named_parameters():
        out[f"{prefix}/grad/{k}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().copy()


def main():
    torch.set_default_dtype(torch.float64)      # the reference's code, evaluated in double
    out = {}
    B = 5
    g = torch.Generator().manual_seed(123)
    x0 = torch.randn(B, 16, generator=g)
    cls = torch.tensor([0, 2, 1, 1, 0])
    out["x0"] = x0.numpy()
    out["cls"] = cls.numpy()

    # ---- improved MeanFlow --------------------------------------------------------------------
    im = load("imflow")
    torch.manual_seed(7)
    model = im.ConditionalFlow(small_cfg(im))
    with torch.no_grad():       # default init leaves biases small; perturb so every term matters
        for p in model.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
    sd(model, out, "imf")
    torch.manual_seed(11)
    loss, mse = model.improved_mean_flow_loss(x0, cls, flow_ratio=0.5)
    loss.backward()
    # replay the draws of imflow.py:137-144 (t, r, mask, e) from the same stream
    torch.manual_seed(11)
    t = torch.rand(B); r = torch.rand(B)
    t, r = torch.maximum(t, r), torch.minimum(t, r)
    r = torch.where(torch.rand(B) < 0.5, t, r)
    e = torch.randn_like(x0)
    with torch.no_grad():       # recompute the loss from the replayed draws with the reference's own forward
        z = (1 - t)[:, None] * x0 + t[:, None] * e
    v = model.forward(z, t[:, None], t[:, None], cls)
    u, dudt = torch.autograd.functional.jvp(lambda z_, t_, r_: model.forward(z_, t_[:, None], r_[:, None], cls),
                                            (z, t, r), (v, torch.zeros_like(t), torch.ones_like(t)))
    chk = ((u + (t - r)[:, None] * dudt - (e - x0)) ** 2).mean()
    assert abs(chk.item() - loss.item()) < 1e-12, (chk.item(), loss.item())
    out.update({"imf/t": t.numpy(), "imf/r": r.numpy(), "imf/e": e.numpy(), "imf/loss": np.float64(loss.item()),
                "imf/v": v.detach().numpy(), "imf/u": u.detach().numpy(), "imf/dudt": dudt.detach().numpy()})
    grads(model, out, "imf")
    assert (t == r).any() and (t != r).any()
    torch.manual_seed(13)
    smp = model.sample(cls, n_steps=3)
    torch.manual_seed(13)
    out["imf/sample_x0"] = torch.randn(B, 16).numpy()
    out["imf/sample_n3"] = smp.numpy()

    # ---- MeanFlow -----------------------------------------------------------------------------
    mf = load("mflow")
    torch.manual_seed(17)
    model = mf.ConditionalFlow(small_cfg(mf))
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
    sd(model, out, "mf")
    torch.manual_seed(19)
    res = model.mean_flow_loss(x0, cls, flow_ratio=0.5, gamma=0.5, c=1e-3)
    loss = res[0] if isinstance(res, tuple) else res
    loss.backward()
    torch.manual_seed(19)
    t = torch.rand(B); r = torch.rand(B)
    t, r = torch.maximum(t, r), torch.minimum(t, r)
    r = torch.where(torch.rand(B) < 0.5, t, r)
    e = torch.randn_like(x0)
    out.update({"mf/t": t.numpy(), "mf/r": r.numpy(), "mf/e": e.numpy(), "mf/loss": np.float64(loss.item())})
    grads(model, out, "mf")
    torch.manual_seed(23)
    smp = model.sample(cls, n_steps=2)
    torch.manual_seed(23)
    out["mf/sample_x0"] = torch.randn(B, 16).numpy()
    out["mf/sample_n2"] = smp.numpy()

    # ---- flow matching ------------------------------------------------------------------------
    fl = load("flow")
    torch.manual_seed(29)
    model = fl.ConditionalFlow(small_cfg(fl))
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
    sd(model, out, "fm")
    torch.manual_seed(31)
    loss = model.flow_matching_loss(x0, cls, 0.001, 0.999)
    loss.backward()
    torch.manual_seed(31)
    e = torch.randn_like(x0)                       # flow.py:109-110: noise first, then time
    t = torch.rand(size=(B, 1)).sigmoid()
    out.update({"fm/t": t.numpy(), "fm/e": e.numpy(), "fm/loss": np.float64(loss.item())})
    grads(model, out, "fm")
    fl.tqdm = lambda it, *a, **k: it               # silence the progress bar of flow.py:118
    torch.manual_seed(37)
    smp = model.sample(cls, n_steps=4)
    torch.manual_seed(37)
    out["fm/sample_x0"] = torch.randn(B, 16).numpy()
    out["fm/sample_n4"] = smp.numpy()

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays,", OUT.stat().st_size, "bytes")
    for k in ("imf/loss", "mf/loss", "fm/loss"):
        print(k, out[k])


if __name__ == "__main__":
    main()
