"""Command line of the reference's ``train.py`` (repo root there): ``--config``, ``--workdir``, ``--resume`` and the
flow-model overrides, on this backend.  Batches come from ``config.data_dir`` through the dataset front end
(``datasets/``: MNIST IDX files, or ``.mp3`` / ``.wav`` audio, ``--target-sr`` resampling on the device) as in the
reference (a config without ``data_dir`` is an error there and here); ``--synthetic`` trains on the benchmark's
``0.1 * N(0,1)`` clips instead.

    python -m meanflow_audio_codec_amd.train_cli --config configs/<name>.json --workdir runs/x [--resume]
"""
from __future__ import annotations

import argparse
from pathlib import Path

_FLOW_ARGS = [("batch-size", int), ("n-steps", int), ("sample-every", int), ("sample-seed", int), ("sample-steps", int),
              ("base-lr", float), ("weight-decay", float), ("seed", int), ("checkpoint-step", int), ("data-dir", str),
              ("noise-dimension", int), ("condition-dimension", int), ("latent-dimension", int), ("num-blocks", int)]
_REQUIRED_WITHOUT_CONFIG = ["batch_size", "n_steps", "sample_every", "sample_seed", "sample_steps", "base_lr",
                            "weight_decay", "seed", "noise_dimension", "condition_dimension", "latent_dimension",
                            "num_blocks"]


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Train flow models (MI355X backend)",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--config", type=Path, help="Path to JSON config file (overrides other arguments)")
    p.add_argument("--workdir", type=Path, required=True, help="Working directory for outputs (samples, checkpoints, logs)")
    p.add_argument("--resume", action="store_true", help="Resume training from latest checkpoint in workdir")
    g = p.add_argument_group("Flow model arguments")
    for name, typ in _FLOW_ARGS:
        g.add_argument(f"--{name}", type=typ, default=None)
    g.add_argument("--use-improved-mean-flow", action="store_true", help="Use improved mean flow objective")
    b = p.add_argument_group("Backend")
    b.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32", help="storage dtype of the big kernels")
    b.add_argument("--steps", type=int, default=None, help="stop after this many steps (default: config.n_steps)")
    b.add_argument("--synthetic", action="store_true", help="train on synthetic clips instead of config.data_dir")
    b.add_argument("--target-sr", type=int, default=None,
                   help="resample every audio file to this rate on the device (the reference keeps 44.1 kHz files)")
    return p


def config_from_args(args):
    from .configs import TrainFlowConfig, load_config_from_json, merge_configs
    if args.config:
        # (the reference assigns config.workdir, which its read-only property rejects; the override is merged here)
        override = {"workdir": str(args.workdir)}
        if args.data_dir is not None:       # the shipped configs carry "data_dir": null
            override["data_dir"] = args.data_dir
        return merge_configs(load_config_from_json(args.config), override)
    missing = [a for a in _REQUIRED_WITHOUT_CONFIG if getattr(args, a) is None]
    if missing:
        raise SystemExit("Missing required arguments (or pass --config): " + ", ".join("--" + m.replace("_", "-") for m in missing))
    kw = {name.replace("-", "_"): getattr(args, name.replace("-", "_")) for name, _ in _FLOW_ARGS}
    kw = {k: v for k, v in kw.items() if v is not None}
    return TrainFlowConfig(workdir=args.workdir, use_improved_mean_flow=args.use_improved_mean_flow, **kw)


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
    config = config_from_args(args)
    import torch
    from .trainers.train import synthetic_iterator, train_flow
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    data = synthetic_iterator(config) if args.synthetic else None
    train_flow(config, data, resume=args.resume, n_steps=args.steps, dtype=dtype, target_sr=args.target_sr)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
