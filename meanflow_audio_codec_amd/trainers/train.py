"""``create_loss_strategy`` / model factory / ``train_flow`` -- host mirror of ``trainers/train.py``.

* ``create_loss_strategy(config)`` follows trainers/train.py:52-153 key by key, including the
  fallback "loss_strategy absent -> improved_mean_flow if use_improved_mean_flow else flow_matching"
  (:62-67) and the falsy-``or`` defaults at :124-126.
* ``create_flow_model(config, D)`` honours ``config.architecture`` (the reference's train_flow always
  builds the MLP flow, defect 1).
* ``train_flow`` is the hot loop of trainers/train.py:330-345 only (tokenise -> train_step -> log);
  workdir layout, checkpoints and plots are the "next" rows N1/N2 of SURVEY 8(f), not built yet.
"""
from __future__ import annotations

import json
import time
from pathlib import Path

import torch

from ..models import TrainState, adamw
from ..preprocessing.tokenization_utils import (compute_token_shape, compute_tokenized_dimension,
                                                create_tokenization_strategy)
from .loss_strategies import FlowMatchingLoss, ImprovedMeanFlowLoss, LossStrategy, MeanFlowLoss
from .noise_schedules import LinearNoiseSchedule, UniformNoiseSchedule
from .time_sampling import LogitNormalTimeSampling, MeanFlowTimeSampling, PRNGKey, UniformTimeSampling
from .training_steps import train_step


def create_loss_strategy(config) -> LossStrategy:
    name = config.loss_strategy
    if name is None:
        name = "improved_mean_flow" if config.use_improved_mean_flow else "flow_matching"
    sched = config.noise_schedule or "linear"
    if sched == "linear":
        noise_schedule = LinearNoiseSchedule(
            noise_min=config.noise_min if config.noise_min is not None else 0.001,
            noise_max=config.noise_max if config.noise_max is not None else 0.999)
    elif sched == "uniform":
        noise_schedule = UniformNoiseSchedule()
    else:
        raise ValueError(f"Unknown noise_schedule: {sched}. Must be one of: 'linear', 'uniform'")
    ts = config.time_sampling or "logit_normal"
    mean = config.time_sampling_mean if config.time_sampling_mean is not None else -0.4
    std = config.time_sampling_std if config.time_sampling_std is not None else 1.0
    if ts == "uniform":
        time_sampling = UniformTimeSampling()
    elif ts == "logit_normal":
        time_sampling = LogitNormalTimeSampling(mean=mean, std=std)
    elif ts == "mean_flow":
        prop = config.time_sampling_data_proportion if config.time_sampling_data_proportion is not None else 0.5
        time_sampling = MeanFlowTimeSampling(mean=mean, std=std, data_proportion=prop)
    else:
        raise ValueError(f"Unknown time_sampling: {ts}. Must be one of: 'uniform', 'logit_normal', 'mean_flow'")
    weighted = config.use_weighted_loss if config.use_weighted_loss is not None else True

    def two_time():
        if isinstance(time_sampling, MeanFlowTimeSampling):
            return time_sampling
        # trainers/train.py:124-126 uses `x or default` (0 / 0.0 fall back to the default)
        return MeanFlowTimeSampling(mean=config.time_sampling_mean or -0.4, std=config.time_sampling_std or 1.0,
                                    data_proportion=config.time_sampling_data_proportion or 0.5)

    if name == "flow_matching":
        return FlowMatchingLoss(noise_schedule=noise_schedule, time_sampling=time_sampling, use_weighted_loss=weighted)
    if name == "mean_flow":
        return MeanFlowLoss(noise_schedule=noise_schedule, time_sampling=two_time(),
                            gamma=config.gamma if config.gamma is not None else 0.5,
                            c=config.c if config.c is not None else 1e-3)
    if name == "improved_mean_flow":
        return ImprovedMeanFlowLoss(noise_schedule=noise_schedule, time_sampling=two_time(), use_weighted_loss=weighted)
    raise ValueError(f"Unknown loss_strategy: {name}. Must be one of: 'flow_matching', 'mean_flow', "
                     "'improved_mean_flow'")


def create_flow_model(config, noise_dimension: int, dtype=torch.float32):
    """models/factories.py:106-148 (create_flow_model), honoured here."""
    arch = config.architecture or "mlp"
    kw = dict(noise_dimension=noise_dimension, condition_dimension=config.condition_dimension,
              num_blocks=config.num_blocks, latent_dimension=config.latent_dimension)
    if arch == "convnet":
        from ..models.conv_flow import ConditionalConvFlow
        return ConditionalConvFlow(**kw, dtype=dtype)
    if arch == "mlp":
        from ..models.mlp_flow import ConditionalFlow
        return ConditionalFlow(**kw, dtype=dtype)
    if arch == "mlp_mixer":
        from ..models.mlp_mixer import ConditionalMLPMixerFlow
        return ConditionalMLPMixerFlow(**kw, dtype=dtype)
    raise ValueError(f"Unknown architecture: {arch}")


def train_flow(config, data_iterator, *, n_steps: int | None = None, dtype=torch.float32, device="cuda",
               log_path: str | Path | None = None, reducer=None, rank: int = 0, world: int = 1):
    """Hot loop of trainers/train.py:330-358 on this backend; ``data_iterator`` yields float32
    ``[B, noise_dimension]`` host or device batches (the reference's iterator contract, :283-306)."""
    if config.condition_dimension % 2:
        raise ValueError(f"condition_dimension must be even, got {config.condition_dimension}")
    tokenization = create_tokenization_strategy(config)
    dataset = config.dataset or "mnist"
    if tokenization is not None:
        D = compute_tokenized_dimension(tokenization, config.noise_dimension, dataset)
        token_shape = compute_token_shape(tokenization, config.noise_dimension, dataset)
    else:
        D, token_shape = config.noise_dimension, None
    model = create_flow_model(config, D, dtype=dtype)
    params = model.init(seed=config.seed, device=device)
    state = TrainState.create(apply_fn=model.apply, params=params,
                              tx=adamw(config.base_lr, config.weight_decay), model=model)
    strategy = create_loss_strategy(config)
    key = PRNGKey(config.seed)
    logf = open(log_path, "a") if log_path else None
    loss_avg = None
    steps = n_steps if n_steps is not None else config.n_steps
    for step in range(steps):
        t0 = time.time()
        x = torch.as_tensor(next(data_iterator), dtype=torch.float32).to(device)
        B = x.shape[0]
        if tokenization is not None:
            x = tokenization.tokenize(x).reshape(B, -1)
        state, loss, key = train_step(state, key, x, strategy, reducer=reducer, row0=rank * B,
                                      global_batch=world * B)
        loss_val = float(loss)                         # device sync, as trainers/train.py:347
        loss_avg = loss_val if loss_avg is None else 0.99 * loss_avg + 0.01 * loss_val   # utils.ema :28-29
        if logf:
            logf.write(json.dumps({"step": step, "loss": loss_val, "loss_avg": loss_avg, "lr": config.base_lr,
                                   "step_time": time.time() - t0}) + "\n")
    if logf:
        logf.close()
    return state, token_shape
