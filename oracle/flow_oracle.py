"""CPU oracle for the velocity nets, loss strategies, optimizer and sampler.
TEST INFRASTRUCTURE ONLY -- only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module.

A PyTorch-CPU (float64 by default) restatement of the reference's JAX/Flax
arithmetic; every function cites the reference lines it follows (paths relative
to the reference root).  The JAX reference itself cannot be imported in the
build container (jax/flax/optax absent), so this restatement is
pinned by

* the reference's own property tests, restated in ``tests/test_oracle_flow.py``
  (``test/test_improved_mean_flow.py:31-54`` boundary condition t=r => v_pred==u,
  ``:57-100`` forward-mode JVP == reverse-mode directional derivative),
* numbers the reference itself produced: its PyTorch implementations
  ``references/archive/{flow,mflow,imflow}.py`` DO import here and were run by
  ``tests/golden/gen_archive_flow_golden.py``; the loss / sampler CORES below
  (``fm_core``, ``mf_core``, ``imf_core``, ``heun_integrate``, ``heun_two_time``) are the single
  implementation used both by the JAX-path restatements (``fm_loss``, ``mf_loss``, ``imf_loss``,
  ``heun_sample``) and by the archive-net variants that ``tests/test_oracle_archive.py`` compares with those
  fixtures (loss, every gradient, u, du/dt, samples) -- so interpolation, targets, JVP tangents, stop-gradient
  placement, the adaptive weight and the Heun integrator are PINNED, and
* hand-derived closed forms for the small ops (LayerNorm, GRN, GELU, AdamW).

Initialiser, optimizer-update and PRNG parity with Flax/optax/JAX are UNPINNED
by any reference test: they follow the documented semantics only.

Parameter trees are nested dicts of tensors with Flax's names/layouts
(SURVEY Appendix B): ``Dense.kernel [in,out]``, ``Conv.kernel [kh,kw,in,out]``.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# small ops
# --------------------------------------------------------------------------


def gelu(x):
    """jax.nn.gelu(approximate=True) -- models/mlp_flow.py:29, conv_flow.py:88,175,201."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def layer_norm(x, eps=1e-6):
    """flax.linen.LayerNorm(use_scale=False, use_bias=False): last axis, eps 1e-6,
    variance E[x^2]-E[x]^2 (models/conv_flow.py:84,160; mlp_flow.py:76)."""
    mu = x.mean(-1, keepdim=True)
    var = (x * x).mean(-1, keepdim=True) - mu * mu
    return (x - mu) * torch.rsqrt(var + eps)


def sinusoidal_embedding(x, dim, max_period=10000.0):
    """meanflow_audio_codec/utils.py:5-13 -- [cos(x f), sin(x f)], f_j = exp(-ln(1e4) j/half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=x.dtype) / half)
    args = x[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def dense(p, x):
    return x @ p["kernel"] + p["bias"]


def weighted_l2_loss(pred, target, p=1.0, c=1e-3):
    """meanflow_audio_codec/utils.py:16-25."""
    delta = pred - target
    per = (delta ** 2).flatten(1).sum(1)
    w = (1.0 / (per + c) ** p).detach()
    return (w * per).mean()


# --------------------------------------------------------------------------
# ConvNeXt flow (models/conv_flow.py)
# --------------------------------------------------------------------------


def grn(p, x, eps=1e-6):
    """GlobalResponseNormalization, models/conv_flow.py:22-45 (x: [B,H,W,C])."""
    gx = torch.sqrt((x ** 2).sum(dim=(1, 2), keepdim=True))
    n = gx.mean(-1, keepdim=True)
    gx = gx / (n + eps)
    return x * (p["gamma"] + gx) + p["beta"]


def conv_nhwc(x, kernel, bias, pad):
    """flax.linen.Conv on NHWC, kernel [kh,kw,in,out], padding SAME."""
    w = kernel.permute(3, 2, 0, 1)  # OIHW
    y = F.conv2d(x.permute(0, 3, 1, 2), w, bias, padding=pad)
    return y.permute(0, 2, 3, 1)


def convnext_block(p, x):
    """ConvNeXtBlock.__call__, models/conv_flow.py:65-115 (drop_path 0)."""
    res = x
    x = conv_nhwc(x, p["Conv_0"]["kernel"], p["Conv_0"]["bias"], 1)
    x = layer_norm(x)
    x = conv_nhwc(x, p["Conv_1"]["kernel"], p["Conv_1"]["bias"], 0)
    x = gelu(x)
    if "GlobalResponseNormalization_0" in p:          # use_grn (conv_flow.py:91-92); absent: no GRN parameters
        x = grn(p["GlobalResponseNormalization_0"], x)
    x = conv_nhwc(x, p["Conv_2"]["kernel"], p["Conv_2"]["bias"], 0)
    x = x * p["layer_scale_gamma"]
    return x + res


def cond_convnext_block(p, x, cond, num_blocks):
    """ConditionalConvNeXtBlock.__call__, models/conv_flow.py:162-205."""
    res = x
    B = x.shape[0]
    C = p["conditioning_layer"]["kernel"].shape[1] // 2
    S = p["input_proj2"]["kernel"].shape[1]
    s = int(round(math.sqrt(S // C)))
    h = dense(p["input_proj2"], gelu(dense(p["input_proj1"], x)))
    h = h.reshape(B, s, s, C)
    h = layer_norm(h)
    cp = dense(p["conditioning_layer"], cond)
    scale, shift = cp[:, :C], cp[:, C:]
    h = (1.0 + scale[:, None, None, :]) * h + shift[:, None, None, :]
    h = convnext_block(p["conv_block"], h)
    o = dense(p["output_proj2"], gelu(dense(p["output_proj1"], h.reshape(B, -1))))
    return o / num_blocks + res


def conv_flow_apply(params, x, time, latents=None):
    """ConditionalConvFlow.__call__, models/conv_flow.py:242-271."""
    cd = params["blocks_0"]["conditioning_layer"]["kernel"].shape[0]
    cond = sinusoidal_embedding(time[:, 0], cd) + sinusoidal_embedding(time[:, 1], cd)
    if latents is not None:
        cond = cond + dense(params["latent_proj"], latents.reshape(latents.shape[0], -1))
    nb = sum(1 for k in params if k.startswith("blocks_"))
    for i in range(nb):
        x = cond_convnext_block(params[f"blocks_{i}"], x, cond, nb)
    return x


def conv_flow_encode(params, x):
    """BUILD DECISION (reference defect 2: ConditionalConvFlow has no ``encode`` although
    every loss strategy calls it, trainers/loss_strategies.py:99,167,250): a bottleneck
    encoder Dense(D->128) -> GELU -> Dense(128->latent) shaped like the block's own input
    projection (models/conv_flow.py:142-146).  Unpinned by the reference."""
    p = params["encoder"]
    return dense(p["dense2"], gelu(dense(p["dense1"], x)))


def conv_flow_shapes(D, cond_dim, latent_in, num_blocks, latent_dim=None, use_grn=True):
    """name -> shape for ConditionalConvFlow."""
    s = int(math.sqrt(D))
    C = min(16, cond_dim // 4)
    S = s * s * C
    tree = {}
    for i in range(num_blocks):
        tree[f"blocks_{i}"] = {
            "input_proj1": {"kernel": (D, 128), "bias": (128,)},
            "input_proj2": {"kernel": (128, S), "bias": (S,)},
            "conditioning_layer": {"kernel": (cond_dim, 2 * C), "bias": (2 * C,)},
            "conv_block": {
                "Conv_0": {"kernel": (3, 3, C, C), "bias": (C,)},
                "Conv_1": {"kernel": (1, 1, C, 2 * C), "bias": (2 * C,)},
                "GlobalResponseNormalization_0": {"gamma": (2 * C,), "beta": (2 * C,)},
                "Conv_2": {"kernel": (1, 1, 2 * C, C), "bias": (C,)},
                "layer_scale_gamma": (C,),
            },
            "output_proj1": {"kernel": (S, 128), "bias": (128,)},
            "output_proj2": {"kernel": (128, D), "bias": (D,)},
        }
    if not use_grn:
        for i in range(num_blocks):
            del tree[f"blocks_{i}"]["conv_block"]["GlobalResponseNormalization_0"]
    if latent_in:
        tree["latent_proj"] = {"kernel": (latent_in, cond_dim), "bias": (cond_dim,)}
    if latent_dim:
        tree["encoder"] = {"dense1": {"kernel": (D, 128), "bias": (128,)},
                           "dense2": {"kernel": (128, latent_dim), "bias": (latent_dim,)}}
    return tree


# --------------------------------------------------------------------------
# MLP flow (models/mlp_flow.py)
# --------------------------------------------------------------------------


def mlp(p, x):
    """MLP, models/mlp_flow.py:12-31."""
    return dense(p["dense2"], gelu(dense(p["dense1"], x)))


def mlp_flow_encode(params, x):
    """ConditionalFlow.encode -> MLPEncoder, models/mlp_flow.py:39-55,153-162."""
    return mlp(params["encoder"]["encoder_mlp"], x)


def mlp_flow_apply(params, x, time, latents=None):
    """ConditionalFlow.__call__/_decode + ConditionalResidualBlock, models/mlp_flow.py:83-117,164-230."""
    nb = sum(1 for k in params if k.startswith("blocks_"))
    b0 = params["blocks_0"]
    cd = b0["conditioning_layer"]["dense1"]["kernel"].shape[0]
    D = b0["mlp"]["dense2"]["kernel"].shape[1]
    I = b0["mlp"]["dense1"]["kernel"].shape[0]
    L = I - D
    if latents is None:
        latents = torch.zeros(x.shape[0], L, dtype=x.dtype)
    cond = sinusoidal_embedding(time[:, 0], cd) + sinusoidal_embedding(time[:, 1], cd)
    for i in range(nb):
        p = params[f"blocks_{i}"]
        xc = torch.cat([latents, x], dim=-1)
        res = xc[:, -D:]
        h = layer_norm(xc)
        sss = mlp(p["conditioning_layer"], cond)
        s1, sh, s2 = sss[:, :I], sss[:, I:2 * I], sss[:, 2 * I:]
        o = mlp(p["mlp"], (1.0 + s1) * h + sh)
        x = o * (1.0 + s2) / nb + res
    return x


def mlp_flow_shapes(D, cond_dim, latent_dim, num_blocks):
    I = latent_dim + D
    H = (D + latent_dim) // 2
    tree = {"encoder": {"encoder_mlp": {"dense1": {"kernel": (D, H), "bias": (H,)},
                                        "dense2": {"kernel": (H, latent_dim), "bias": (latent_dim,)}}}}
    for i in range(num_blocks):
        tree[f"blocks_{i}"] = {
            "conditioning_layer": {"dense1": {"kernel": (cond_dim, cond_dim), "bias": (cond_dim,)},
                                   "dense2": {"kernel": (cond_dim, 2 * I + D), "bias": (2 * I + D,)}},
            "mlp": {"dense1": {"kernel": (I, I), "bias": (I,)}, "dense2": {"kernel": (I, D), "bias": (D,)}},
        }
    return tree


# --------------------------------------------------------------------------
# MLP-Mixer flow and encoder (models/mlp_mixer.py)
# --------------------------------------------------------------------------


def mixer_block(p, x, cond):
    """MLPMixerBlock.__call__, models/mlp_mixer.py:66-94 (x: [B, tokens, channels]); AdaLN :27-45."""
    C = x.shape[-1]

    def adaln(v, d):
        ss = dense(d, cond)
        return (1.0 + ss[:, None, :C]) * layer_norm(v) + ss[:, None, C:]

    res = x
    a = adaln(x, p["Dense_0"]).transpose(1, 2)
    a = dense(p["Dense_2"], gelu(dense(p["Dense_1"], a))).transpose(1, 2)
    x = a + res
    res = x
    a = adaln(x, p["Dense_3"])
    a = dense(p["Dense_5"], gelu(dense(p["Dense_4"], a)))
    return a + res


def mixer_flow_apply(params, x, time, latents=None):
    """ConditionalMLPMixerFlow.__call__ / ConditionalMLPMixerBlock, models/mlp_mixer.py:137-163,205-235."""
    cd = params["blocks_0"]["mixer_block"]["Dense_0"]["kernel"].shape[0]
    cond = sinusoidal_embedding(time[:, 0], cd) + sinusoidal_embedding(time[:, 1], cd)
    if latents is not None:
        cond = cond + dense(params["latent_proj"], latents.reshape(latents.shape[0], -1))
    nb = sum(1 for k in params if k.startswith("blocks_"))
    B = x.shape[0]
    for i in range(nb):
        p = params[f"blocks_{i}"]
        C = p["mixer_block"]["Dense_0"]["kernel"].shape[1] // 2
        h = dense(p["input_proj"], x).reshape(B, -1, C)
        h = mixer_block(p["mixer_block"], h, cond)
        x = dense(p["output_proj"], h.reshape(B, -1)) / nb + x
    return x


def mixer_encode(params, x):
    """MLPMixerEncoder.__call__, models/mlp_mixer.py:281-323 -> [B, num_latent_tokens, latent_dim]."""
    p = params["encoder"]
    n_lat, L = p["latent_queries"].shape
    B = x.shape[0]
    ctx = dense(p["input_proj"], x).reshape(B, -1, L)
    n_ctx = ctx.shape[1]
    allt = torch.cat([ctx, p["latent_queries"][None].expand(B, n_lat, L)], dim=1)
    cond = p["condition_emb"][None].expand(B, L)
    return mixer_block(p["mixer_block"], allt, cond)[:, n_ctx:, :]


def _mixer_block_shapes(nt, C, cond_dim, tmd, cmd):
    return {"Dense_0": {"kernel": (cond_dim, 2 * C), "bias": (2 * C,)},
            "Dense_1": {"kernel": (nt, tmd), "bias": (tmd,)}, "Dense_2": {"kernel": (tmd, nt), "bias": (nt,)},
            "Dense_3": {"kernel": (cond_dim, 2 * C), "bias": (2 * C,)},
            "Dense_4": {"kernel": (C, cmd), "bias": (cmd,)}, "Dense_5": {"kernel": (cmd, C), "bias": (C,)}}


def mixer_flow_shapes(D, cond_dim, latent_dim, num_blocks, C=16, tmd=2048, cmd=2048, n_lat=32, n_ctx=512):
    s = int(math.sqrt(D))
    nt = s * s
    tree = {}
    for i in range(num_blocks):
        tree[f"blocks_{i}"] = {"input_proj": {"kernel": (D, nt * C), "bias": (nt * C,)},
                               "mixer_block": _mixer_block_shapes(nt, C, cond_dim, tmd, cmd),
                               "output_proj": {"kernel": (nt * C, D), "bias": (D,)}}
    tree["latent_proj"] = {"kernel": (n_lat * latent_dim, cond_dim), "bias": (cond_dim,)}
    tree["encoder"] = {"input_proj": {"kernel": (D, n_ctx * latent_dim), "bias": (n_ctx * latent_dim,)},
                       "latent_queries": (n_lat, latent_dim), "condition_emb": (latent_dim,),
                       "mixer_block": _mixer_block_shapes(n_ctx + n_lat, latent_dim, latent_dim, tmd, cmd)}
    return tree


# --------------------------------------------------------------------------
# parameter helpers
# --------------------------------------------------------------------------


def init_params(shapes, seed=0, dtype=torch.float64, special=True):
    """lecun-normal kernels (variance 1/fan_in; the reference uses the truncated
    variant -- unpinned), zero biases, GRN gamma/beta zeros, layer-scale 1e-6
    (models/conv_flow.py:41-42,99-103).  ``special=False`` draws every leaf
    N(0, .) so tests exercise all terms."""
    g = torch.Generator().manual_seed(seed)

    def rec(t, name):
        if isinstance(t, dict):
            return {k: rec(v, k) for k, v in t.items()}
        shape = t
        if name == "kernel":
            fan_in = math.prod(shape[:-1])
            return torch.randn(shape, generator=g, dtype=dtype) / math.sqrt(fan_in)
        if special:
            if name == "layer_scale_gamma":
                return torch.full(shape, 1e-6, dtype=dtype)
            if name in ("latent_queries", "condition_emb"):     # nn.initializers.normal(0.02), mlp_mixer.py:261-273
                return 0.02 * torch.randn(shape, generator=g, dtype=dtype)
            return torch.zeros(shape, dtype=dtype)
        scale = {"bias": 0.1, "gamma": 0.5, "beta": 0.1, "layer_scale_gamma": 0.5, "latent_queries": 0.5,
                 "condition_emb": 0.5}[name]
        return scale * torch.randn(shape, generator=g, dtype=dtype)

    return rec(shapes, "")


def flatten(tree, prefix=""):
    out = {}
    for k, v in tree.items():
        name = f"{prefix}/{k}" if prefix else k
        if isinstance(v, dict):
            out.update(flatten(v, name))
        else:
            out[name] = v
    return out


def unflatten(flat):
    tree = {}
    for name, v in flat.items():
        parts = name.split("/")
        d = tree
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = v
    return tree


def tree_map(f, tree):
    return {k: tree_map(f, v) if isinstance(v, dict) else f(v) for k, v in tree.items()}


# --------------------------------------------------------------------------
# noise schedules / time sampling (trainers/noise_schedules.py, utils.py)
# --------------------------------------------------------------------------


def linear_interpolate(x0, x1, t, noise_min=0.001, noise_max=0.999):
    """LinearNoiseSchedule.interpolate, trainers/noise_schedules.py:69-80 (t: [B,1])."""
    return (1.0 - t) * x0 + (noise_min + noise_max * t) * x1


def linear_target(x0, x1, noise_max=0.999):
    """LinearNoiseSchedule.compute_target, trainers/noise_schedules.py:82-88."""
    return noise_max * x1 - x0


def sample_tr_from_normals(nt, nr, mean=-0.4, std=1.0, data_proportion=0.5):
    """meanflow_audio_codec/utils.py:32-45 with the two normal draws passed in
    (JAX PRNG streams are not reproducible here): t=max, r=min of two
    logit-normals; the FIRST int(B*data_proportion) rows get r = t."""
    t = torch.sigmoid(nt * std + mean)
    r = torch.sigmoid(nr * std + mean)
    t, r = torch.maximum(t, r), torch.minimum(t, r)
    B = t.shape[0]
    mask = (torch.arange(B) < int(B * data_proportion))[:, None]
    r = torch.where(mask, t, r)
    return t, r


# --------------------------------------------------------------------------
# loss strategies (trainers/loss_strategies.py) with explicit (e, t, r)
# --------------------------------------------------------------------------


def _grads(loss, params):
    flat = flatten(params)
    names = list(flat)
    gs = torch.autograd.grad(loss, [flat[n] for n in names], allow_unused=True)
    return unflatten({n: (g if g is not None else torch.zeros_like(flat[n])) for n, g in zip(names, gs)})


def _req(params):
    return tree_map(lambda v: v.detach().clone().requires_grad_(True), params)


def imf_core(u_fn, v, z, t, r, tangent="t"):
    """The improved-MeanFlow compound prediction, shared by the JAX-path restatement and the archive variant.
    ``u_fn(z, t, r) -> u``; JVP of u along (v, tdot, rdot): ``tangent="t"`` is (v, 1, 0)
    (trainers/loss_strategies.py:263-267), ``tangent="r"`` is (v, 0, 1) (references/archive/imflow.py:152-155).
    du/dt is stop-gradient (loss_strategies.py:270 / imflow.py:158 ``dudt.detach()``);
    V = u + (t - r) sg(du/dt).  Returns (u, dudt, V)."""
    one, zero = torch.ones_like(t), torch.zeros_like(t)
    tang = (v, one, zero) if tangent == "t" else (v, zero, one)
    u, dudt = torch.func.jvp(u_fn, (z, t, r), tang)
    dudt = dudt.detach()
    return u, dudt, u + (t - r) * dudt


def mf_core(u_fn, z, t, r, v, gamma=0.5, c=1e-3):
    """MeanFlow loss core (trainers/loss_strategies.py:177-198 == references/archive/mflow.py:143-150):
    JVP tangent (v, 1, 0) with v = e - x, u_tgt = v - clip(t-r, 0, 1) du/dt, stop-gradient on the whole target,
    per-example MEAN squared error, weight sg(1/(d + c)^(1-gamma)).  Returns (loss, u, dudt)."""
    u, dudt = torch.func.jvp(u_fn, (z, t, r), (v, torch.ones_like(t), torch.zeros_like(r)))
    u_tgt = v - torch.clamp(t - r, 0.0, 1.0) * dudt
    err = u - u_tgt.detach()
    dsq = (err ** 2).flatten(1).mean(1)
    w = (1.0 / (dsq + c) ** (1.0 - gamma)).detach()
    return (w * dsq).mean(), u, dudt


def fm_core(f, x, e, t, noise_min=0.001, noise_max=0.999):
    """Flow-matching pieces (trainers/loss_strategies.py:98-106 with LinearNoiseSchedule ==
    references/archive/flow.py:107-113): (pred, target) with pred = f(z, t)."""
    return f(linear_interpolate(x, e, t, noise_min, noise_max), t), linear_target(x, e, noise_max)


def imf_parts(apply, encode, params, x, e, t, r, noise_min=0.001, noise_max=0.999):
    """ImprovedMeanFlowLoss.compute_loss pieces, trainers/loss_strategies.py:227-277.
    Returns (v, u, dudt, v_pred, target) -- dudt already detached."""
    z = linear_interpolate(x, e, t, noise_min, noise_max)
    target = linear_target(x, e, noise_max)
    latents = encode(params, x) if encode is not None else None
    v = apply(params, z, torch.cat([t, torch.zeros_like(t)], -1), latents)

    def u_fn(z_, t_, r_):
        return apply(params, z_, torch.cat([t_, t_ - r_], -1), latents)

    u, dudt, v_pred = imf_core(u_fn, v, z, t, r, tangent="t")
    return v, u, dudt, v_pred, target


def imf_loss(apply, encode, params, x, e, t, r, use_weighted_loss=True, noise_min=0.001, noise_max=0.999):
    """ImprovedMeanFlowLoss.compute_loss -> (loss, grads, aux)."""
    params = _req(params)
    v, u, dudt, v_pred, target = imf_parts(apply, encode, params, x, e, t, r, noise_min, noise_max)
    loss = weighted_l2_loss(v_pred, target) if use_weighted_loss else ((v_pred - target) ** 2).mean()
    return loss.detach(), _grads(loss, params), dict(v=v.detach(), u=u.detach(), dudt=dudt, v_pred=v_pred.detach())


def fm_loss(apply, encode, params, x, e, t, use_weighted_loss=True, noise_min=0.001, noise_max=0.999,
            schedule="linear"):
    """FlowMatchingLoss.compute_loss, trainers/loss_strategies.py:73-112."""
    params = _req(params)
    latents = encode(params, x) if encode is not None else None
    f = lambda z_, t_: apply(params, z_, torch.cat([t_, torch.zeros_like(t_)], -1), latents)
    if schedule == "linear":
        pred, target = fm_core(f, x, e, t, noise_min, noise_max)
    else:  # UniformNoiseSchedule, trainers/noise_schedules.py:91-115
        pred, target = fm_core(f, x, e, t, 0.0, 1.0)
    loss = weighted_l2_loss(pred, target) if use_weighted_loss else ((pred - target) ** 2).mean()
    return loss.detach(), _grads(loss, params), dict(pred=pred.detach())


def mf_loss(apply, encode, params, x, e, t, r, gamma=0.5, c=1e-3):
    """MeanFlowLoss.compute_loss, trainers/loss_strategies.py:141-201."""
    params = _req(params)
    z = (1.0 - t) * x + t * e
    target = e - x
    latents = encode(params, x) if encode is not None else None

    def u_fn(z_, t_, r_):
        return apply(params, z_, torch.cat([t_, t_ - r_], -1), latents)

    loss, u, dudt = mf_core(u_fn, z, t, r, target, gamma, c)
    return loss.detach(), _grads(loss, params), dict(u=u.detach(), dudt=dudt.detach())


# --------------------------------------------------------------------------
# optimizer: optax.adamw (trainers/train.py:236)
# --------------------------------------------------------------------------


def adamw_step(p, g, m, v, step, lr, wd, b1=0.9, b2=0.999, eps=1e-8):
    """One optax.adamw update; ``step`` is the 1-based count after this update.
    m<-b1 m+(1-b1) g; v<-b2 v+(1-b2) g^2; p <- p - lr (mhat/(sqrt(vhat)+eps) + wd p)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mh = m / (1 - b1 ** step)
    vh = v / (1 - b2 ** step)
    p = p - lr * (mh / (torch.sqrt(vh) + eps) + wd * p)
    return p, m, v


# --------------------------------------------------------------------------
# sampling (evaluators/sampling.py) and the 1-NFE decode
# --------------------------------------------------------------------------


def heun_integrate(f, x, n_steps):
    """The integrator of evaluators/sampling.py:50-96 == references/archive/flow.py:115-124: ``n_steps`` Heun steps
    over ts = linspace(1, 0, n_steps) with dt = 1/n_steps; ``f(x, t_scalar) -> k``."""
    dt = 1.0 / float(n_steps)
    for t in torch.linspace(1.0, 0.0, n_steps, dtype=x.dtype):
        k1 = f(x, t)
        k2 = f(x - dt * k1, t - dt)
        x = x - (dt / 2.0) * (k1 + k2)
    return x


def heun_two_time(f, x, n_steps):
    """The two-time sampler of references/archive/imflow.py:170-182 (== mflow.py:154-166): t_vals = linspace(1, 0,
    n_steps + 1); per step k1 = f(x, t, r), k2 = f(x - dt k1, r, r), x -= dt/2 (k1 + k2).  (The JAX sampler has no
    counterpart: it always passes h = 0.)"""
    tv = torch.linspace(1.0, 0.0, n_steps + 1, dtype=x.dtype)
    for i in range(n_steps):
        t, r = tv[i], tv[i + 1]
        dt = t - r
        k1 = f(x, t, r)
        k2 = f(x - dt * k1, r, r)
        x = x - (dt / 2.0) * (k1 + k2)
    return x


def heun_sample(apply, params, x, latents, n_steps, guidance_scale=1.0):
    """sample(), evaluators/sampling.py:50-96, from a given initial noise x."""
    B = x.shape[0]

    def f(xx, tval):
        tp = torch.cat([torch.full((B, 1), float(tval), dtype=x.dtype), torch.zeros(B, 1, dtype=x.dtype)], -1)
        k = apply(params, xx, tp, latents)
        if guidance_scale != 1.0:
            k = guidance_scale * k + (1.0 - guidance_scale) * apply(params, xx, tp, None)
        return k

    return heun_integrate(f, x, n_steps)


def one_step_decode(apply, params, eps, latents):
    """True 1-NFE MeanFlow decode x0 = eps - u(eps, r=0, t=1): model time input [t=1, h=1]
    (documentation/research/improved_meanflow/improved_meanflow_key_eqn.md:313-316)."""
    B = eps.shape[0]
    tp = torch.ones(B, 2, dtype=eps.dtype)
    return eps - apply(params, eps, tp, latents)


# --------------------------------------------------------------------------
# The archive variant: the net of references/archive/{flow,mflow,imflow}.py restated, so that the cores above can be
# compared with numbers those files produced (tests/golden/gen_archive_flow_golden.py).  It differs from the JAX
# MLP flow on purpose: SiLU, LayerNorm eps 1e-5 (torch default), class embedding instead of encoder
# latents, emb(t) + emb(r) with 2 pi logspace(0, 3) frequencies in [sin, cos] order.
# --------------------------------------------------------------------------


def archive_embedding(x, dim):
    """references/archive/imflow.py:79-83.  ``x``: [B].  (The frequencies are built from a float32 scalar there,
    ``torch.log10(torch.tensor(1_000.))``; reproduced literally.)"""
    freqs = torch.logspace(0, torch.log10(torch.tensor(1_000., dtype=torch.float32)), dim // 2, dtype=x.dtype)
    ang = 2 * torch.pi * freqs[:, None] * x[None, :]
    return torch.cat((ang.sin(), ang.cos()), 0).T


def _lin(p, pre, x):
    return x @ p[f"{pre}.weight"].T + p[f"{pre}.bias"]          # torch.nn.Linear: weight [out, in]


def archive_net(p, x, t, r, cls_idx):
    """ConditionalFlow.forward of references/archive/imflow.py:107-123 (``r=None``: flow.py:101-105, one time).
    ``p``: the module's state_dict as {name: tensor}; ``t``, ``r``: [B, 1]."""
    emb = p["cls_emb.0.weight"][cls_idx]
    cls = _lin(p, "cls_emb.2", F.silu(emb))
    cond = cls + archive_embedding(t[:, 0], cls.shape[-1])
    if r is not None:
        cond = cond + archive_embedding(r[:, 0], cls.shape[-1])
    nb = 1 + max(int(k.split(".")[1]) for k in p if k.startswith("blocks."))
    for i in range(nb):
        b = f"blocks.{i}"
        h = F.layer_norm(x, [x.shape[-1]])                      # eps 1e-5, no affine (imflow.py:99)
        mod = _lin(p, f"{b}.cond.2", F.silu(_lin(p, f"{b}.cond.0", cond)))
        s1, sh, s2 = mod.chunk(3, dim=-1)
        h = _lin(p, f"{b}.mlp.2", F.silu(_lin(p, f"{b}.mlp.0", (1 + s1) * h + sh))) * (1 + s2)
        x = x + h / nb
    return x


def _archive_grads(loss, p):
    names = list(p)
    gs = torch.autograd.grad(loss, [p[n] for n in names], allow_unused=True)
    return {n: (g if g is not None else torch.zeros_like(p[n])) for n, g in zip(names, gs)}


def archive_imf_loss(p, x0, cls_idx, e, t, r):
    """improved_mean_flow_loss of references/archive/imflow.py:125-168 with the draws passed in (t, r: [B, 1]):
    z = (1-t) x + t e, v = u(z, t, t), ``imf_core`` with the r-tangent, plain MSE against e - x."""
    p = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    z = linear_interpolate(x0, e, t, 0.0, 1.0)
    v = archive_net(p, z, t, t, cls_idx)
    u, dudt, V = imf_core(lambda z_, t_, r_: archive_net(p, z_, t_, r_, cls_idx), v, z, t, r, tangent="r")
    loss = ((V - linear_target(x0, e, 1.0)) ** 2).mean()
    return loss.detach(), _archive_grads(loss, p), dict(v=v.detach(), u=u.detach(), dudt=dudt)


def archive_mf_loss(p, x0, cls_idx, e, t, r, gamma=0.5, c=1e-3):
    """mean_flow_loss of references/archive/mflow.py:128-152 through ``mf_core``."""
    p = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    z = linear_interpolate(x0, e, t, 0.0, 1.0)
    loss, u, dudt = mf_core(lambda z_, t_, r_: archive_net(p, z_, t_, r_, cls_idx), z, t, r, e - x0, gamma, c)
    return loss.detach(), _archive_grads(loss, p), dict(u=u.detach(), dudt=dudt.detach())


def archive_fm_loss(p, x0, cls_idx, e, t, noise_min=0.001, noise_max=0.999):
    """flow_matching_loss of references/archive/flow.py:107-113 through ``fm_core`` (plain MSE)."""
    p = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    pred, target = fm_core(lambda z_, t_: archive_net(p, z_, t_, None, cls_idx), x0, e, t, noise_min, noise_max)
    loss = ((pred - target) ** 2).mean()
    return loss.detach(), _archive_grads(loss, p), dict(pred=pred.detach())
