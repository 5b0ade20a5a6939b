"""``create_loss_strategy`` / model factory / ``train_flow`` -- host mirror of ``trainers/train.py``.

* ``create_loss_strategy(config)`` follows trainers/train.py:52-153 key by key, including the
  fallback "loss_strategy absent -> improved_mean_flow if use_improved_mean_flow else flow_matching"
  (:62-67) and the falsy-``or`` defaults at :124-126.
* ``create_flow_model(config, D)`` honours ``config.architecture`` (the reference's train_flow always
  builds the MLP flow, defect 1).
* ``train_flow`` is trainers/train.py:156-507: the hot loop (tokenise -> train_step -> log) plus, when a work
  directory is given, the reference's workdir layout, checkpoints (trainers/checkpoint.py) and ``--resume``
  (SURVEY 8(f) rows N1/N2).  The dataset front end (N4) is not built: batches come from ``data_iterator``.
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


def load_flow_state(config, checkpoint_path, batch_size: int | None = None, *, dtype=torch.float32, device="cuda"):
    """``load_flow_state`` of trainers/utils.py:439-470: build the model of ``config`` (a ``TrainFlowConfig`` or its
    ``to_dict()``), wrap it in a TrainState and fill it from the checkpoint.  Unlike the reference -- which always
    builds the MLP at ``noise_dimension`` -- the architecture and the tokenised dimension of the config are honoured,
    as in ``train_flow``.  ``batch_size`` is accepted for signature compatibility (nothing here is traced).
    Returns ``(model, state)``."""
    from . import checkpoint as ck
    if isinstance(config, dict):
        from ..configs import TrainFlowConfig
        config = TrainFlowConfig.from_dict(config)
    tokenization = create_tokenization_strategy(config)
    D = (compute_tokenized_dimension(tokenization, config.noise_dimension, config.dataset or "mnist")
         if tokenization is not None else config.noise_dimension)
    model = create_flow_model(config, D, dtype=dtype)
    params = model.init(seed=0, device=device)
    state = TrainState.create(apply_fn=model.apply, params=params,
                              tx=adamw(config.base_lr, config.weight_decay), model=model)
    return model, ck.load_checkpoint(Path(checkpoint_path), state)


def dataset_iterator(config, device="cuda", target_sr: int | None = None, rank: int = 0, world: int = 1):
    """The data side of trainers/train.py:281-306: ``[B, noise_dimension]`` float32 batches from ``config.data_dir``.

    * ``mnist``: ``load_mnist(split="train")`` images (labels dropped), :283-289.
    * ``audio``: ``build_audio_pipeline(frame_sz=noise_dimension, seed, batch_size)``, frames kept in HBM; ``.mp3`` and
      ``.wav`` files; ``target_sr`` converts each file on the device.  The reference flattens the
      ``[B, frame_sz, n_channels]`` batch to ``[B, frame_sz * n_channels]`` (:298-303), which no longer has the
      model's ``noise_dimension`` for the two-channel frames its own loader always produces; here the channels are
      averaged to one so that the configured dimension holds.  (The two-channel token layout of ``MDCTLayer`` --
      L/R concatenated on the coefficient axis -- is what ``MDCTTokenization.tokenize`` does with a ``[B, T, 2]`` batch.)
    * data parallel: rank ``r`` of ``world`` seeds its pipeline with ``seed + r``, i.e. draws its own stream."""
    from ..datasets import build_audio_pipeline, load_mnist
    name = config.dataset or "mnist"
    seed = int(config.seed) + int(rank)
    if name == "mnist":
        for img, _ in load_mnist(data_dir=str(config.data_dir), split="train", batch_size=config.batch_size, seed=42 + rank):
            yield img
    elif name == "audio":
        it = build_audio_pipeline(data_dir=str(config.data_dir), seed=seed, frame_sz=config.noise_dimension,
                                  batch_size=config.batch_size, device=device, target_sr=target_sr,
                                  extensions=(".mp3", ".wav"))
        for b in it:
            yield b.mean(dim=2) if b.ndim == 3 else b
    else:
        raise ValueError(f"Unknown dataset: {name}")


def synthetic_iterator(config, device="cuda", scale: float = 0.1):
    """Endless ``[batch_size, noise_dimension]`` float32 batches ~ scale * N(0,1), seeded by ``config.seed`` (the
    measurement input of SURVEY 8(d); the dataset front end is row N4 and not built)."""
    g = torch.Generator(device=device).manual_seed(int(config.seed))
    while True:
        yield scale * torch.randn(config.batch_size, config.noise_dimension, generator=g, device=device)


def _save_samples(samples_dir: Path, step: int, smps: torch.Tensor, config, original_dim: int) -> None:
    """``samples/step_%04d.*`` (trainers/train.py:358-404): the first <= 16 samples as .npy and, when matplotlib is
    importable, the reference's PNG grid (square images for MNIST, waveforms otherwise)."""
    import numpy as np
    n_show = min(16, len(smps))
    arr = smps[:n_show].float().cpu().numpy()
    np.save(samples_dir / f"step_{step:04d}.npy", arr)
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:   # noqa: BLE001
        return
    side = int(original_dim ** 0.5)
    fig, axes = plt.subplots(4, 4, figsize=(8, 8))
    for i, ax in enumerate(axes.ravel()):
        ax.axis("off")
        if i >= n_show:
            continue
        if (config.dataset or "mnist") == "mnist" and side * side == arr.shape[1]:
            ax.imshow(arr[i].reshape(side, side), cmap="gray")
        else:
            ax.plot(arr[i][:4096], linewidth=0.5)
    fig.savefig(samples_dir / f"step_{step:04d}.png", dpi=80)
    plt.close(fig)


def train_flow(config, data_iterator=None, *, resume: bool = False, n_steps: int | None = None,
               dtype=torch.float32, device="cuda", log_path: str | Path | None = None, reducer=None,
               rank: int = 0, world: int = 1, workdir: str | Path | None = None, target_sr: int | None = None):
    """``train_flow`` of trainers/train.py:156-507 on this backend.

    ``data_iterator`` yields float32 ``[B, noise_dimension]`` host or device batches (the reference's iterator
    contract, :283-306); without one the batches come from ``config.data_dir`` through ``dataset_iterator`` (the
    reference's MNIST / audio pipelines; ``target_sr`` resamples the audio files on the device).

    With a work directory (``workdir`` argument or ``config.workdir``) rank 0 writes the reference's layout:
    ``config.json``, ``metadata.json``, ``config_diff.json`` (on resume), ``logs/train_log.jsonl``,
    ``samples/step_%04d.*`` every ``sample_every`` steps and at the end, ``checkpoints/step_%05d.msgpack`` + sidecar
    at ``checkpoint_step`` (default: the last step) with ``max_checkpoints_to_keep`` cleanup, ``summary.json``.
    ``resume=True`` continues from the newest checkpoint that loads (``load_checkpoint_and_resume``), else from
    scratch, as the reference does (:259-270).  Returns ``(state, token_shape)``."""
    from . import checkpoint as ck
    if config.condition_dimension % 2:
        raise ValueError(f"condition_dimension must be even, got {config.condition_dimension}")
    if data_iterator is None:
        if config.data_dir is None:
            raise ValueError("config.data_dir must be provided. It cannot be None. "
                             "Please specify a valid data directory path in your configuration.")
        data_iterator = dataset_iterator(config, device=device, target_sr=target_sr, rank=rank, world=world)
    tokenization = create_tokenization_strategy(config)
    dataset = config.dataset or "mnist"
    if tokenization is not None:
        D = compute_tokenized_dimension(tokenization, config.noise_dimension, dataset)
        token_shape = compute_token_shape(tokenization, config.noise_dimension, dataset)
    else:
        D, token_shape = config.noise_dimension, None

    wd = Path(workdir) if workdir is not None else (Path(config.workdir) if config.workdir is not None else None)
    write = wd is not None and rank == 0
    if write:
        for sub in ("samples", "checkpoints", "logs"):
            (wd / sub).mkdir(parents=True, exist_ok=True)
        cfg_dict = json.loads(json.dumps(config.to_dict(), default=str))
        if resume and (wd / "config.json").exists():
            try:       # trainers/utils.py:1142-1163: a diff that cannot be computed is skipped, not fatal
                from ..configs import diff_configs, load_config_from_json
                diff = json.loads(json.dumps(diff_configs(load_config_from_json(wd / "config.json"), config), default=str))
                for k in ("workdir",):
                    diff["changed"].pop(k, None)
                if diff["changed"] or diff["added"] or diff["removed"]:
                    ck.save_json(wd / "config_diff.json", diff)
            except (OSError, KeyError, TypeError, ValueError):
                pass
        commit, short = ck.get_git_commit_hash()
        ck.save_json(wd / "metadata.json", {
            "timestamp": __import__("datetime").datetime.now().isoformat(), "git_commit": commit,
            "git_commit_short": short, "config_hash": ck.compute_config_hash(cfg_dict),
            "python_version": __import__("sys").version.split()[0], "jax_version": None,
            "backend": f"meanflow_audio_codec_amd (torch {torch.__version__})",
            "platform": __import__("platform").platform(), "cpu_count": __import__("os").cpu_count(),
            "device_info": {"device": torch.cuda.get_device_name(0) if torch.cuda.is_available() else "cpu",
                            "world_size": world, "dtype": str(dtype)}})
        ck.save_json(wd / "config.json", cfg_dict)

    model = create_flow_model(config, D, dtype=dtype)
    params = model.init(seed=config.seed, device=device)
    state = TrainState.create(apply_fn=model.apply, params=params,
                              tx=adamw(config.base_lr, config.weight_decay), model=model)
    start_step = 0
    if resume and wd is not None:
        try:
            state, start_step = ck.load_checkpoint_and_resume(wd, state, config)
            print(f"Resuming from checkpoint at step {start_step}")
        except (FileNotFoundError, ValueError) as e:
            print(f"Failed to resume from checkpoint: {e}\nStarting from scratch")
            start_step = 0
    strategy = create_loss_strategy(config)
    key = PRNGKey(config.seed, start_step)     # the key is advanced once per completed step (training_steps.py)
    if log_path is None and write:
        log_path = wd / "logs" / "train_log.jsonl"
    logger = ck.LogWriter(Path(log_path)) if (log_path and rank == 0) else None
    loss_avg = None
    steps = n_steps if n_steps is not None else config.n_steps
    checkpoint_step = config.checkpoint_step if config.checkpoint_step is not None else steps
    saved_checkpoint = False
    t_begin = time.time()
    step_times = []

    def draw_samples(step):
        from ..evaluators.sampling import sample
        B = config.batch_size
        # train.py:360-364 feeds zeros [B, latent_dimension]; the Mixer's latent_proj is sized for its encoder's
        # [B, 32, latent] output (the reference would fail Flax's shape check there), so the zeros take the model's shape
        latents = torch.zeros((B,) + tuple(model.latent_shape), dtype=torch.float32, device=device)
        smps = sample(state.apply_fn, D, state.work, PRNGKey(config.sample_seed), latents=latents,
                      n_steps=config.sample_steps, use_improved_mean_flow=config.use_improved_mean_flow,
                      guidance_scale=1.0)
        if tokenization is not None and token_shape is not None:
            smps = tokenization.detokenize(smps.float().reshape(B, token_shape[0], token_shape[1]))
        _save_samples(wd / "samples", step, smps, config, config.noise_dimension)

    for step in range(start_step, steps):
        t0 = time.time()
        x = torch.as_tensor(next(data_iterator), dtype=torch.float32).to(device)
        B = x.shape[0]
        if tokenization is not None:
            x = tokenization.tokenize(x).reshape(B, -1)
        # data parallel: this rank's batch holds global rows rank, rank + world, ... (distributed.shard_rows)
        state, loss, key = train_step(state, key, x, strategy, reducer=reducer, row0=rank, row_stride=world,
                                      global_batch=world * B)
        loss_val = float(loss)                         # device sync, as trainers/train.py:347
        loss_avg = loss_val if loss_avg is None else 0.99 * loss_avg + 0.01 * loss_val   # utils.ema :28-29
        step_times.append(time.time() - t0)
        if logger:
            logger.write_step(step, {"loss": loss_val, "loss_avg": loss_avg, "lr": config.base_lr,
                                     "step_time": step_times[-1]})
        if step % 50 == 0 and rank == 0 and write:
            print(f"step={step:04d} loss={loss_val:.9f} loss_avg={loss_avg:.9f}")
        if write and step % config.sample_every == 0:
            draw_samples(step)
        if step + 1 == checkpoint_step and getattr(reducer, "shard_optimizer", False):
            reducer.gather_master(state)      # collective: every rank (sharded masters / moments -> complete tensors)
        if write and step + 1 == checkpoint_step:
            ck.save_checkpoint_with_metadata(wd / "checkpoints" / f"step_{step + 1:05d}.msgpack", state, step + 1, config)
            saved_checkpoint = True
            if config.max_checkpoints_to_keep is not None:
                ck.cleanup_old_checkpoints(wd, config.max_checkpoints_to_keep, keep_final=False, final_step=None)
    if getattr(reducer, "shard_optimizer", False):
        reducer.gather_master(state)          # the final checkpoint below (and the caller) get the complete state
    if write:
        draw_samples(steps)
        if not saved_checkpoint:
            ck.save_checkpoint_with_metadata(wd / "checkpoints" / f"step_{steps:05d}.msgpack", state, steps, config)
        if config.max_checkpoints_to_keep is not None:
            ck.cleanup_old_checkpoints(wd, config.max_checkpoints_to_keep, keep_final=True, final_step=steps)
    if logger:
        logger.close()
    if write:
        total = time.time() - t_begin
        n = max(1, len(step_times))
        ck.save_json(wd / "summary.json", {
            "profiling": {"total_training_time_seconds": total, "total_training_time_hours": total / 3600.0,
                          "steps_per_second": len(step_times) / total if total > 0 else 0.0,
                          "avg_step_time": sum(step_times) / n,
                          "param_count": int(sum(v.numel() for v in state.params.values()))},
            "metrics": ck.generate_training_summary(wd / "logs" / "train_log.jsonl")})
    return state, token_shape
