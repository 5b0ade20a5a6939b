"""Experiment configuration -- stdlib-only mirror of the reference's ``configs/config.py`` surface
(``TrainFlowConfig`` with ``base/model/dataset/method/training`` groups :348-705, flat-v1 ->
hierarchical-v2 migration :713-816, ``load_config_from_json`` :1103-1117, merge/diff :824-1022), so the
reference's ``configs/*.json`` load unchanged.

Implemented table-first: ``_FIELDS`` lists every key once with its group, default and validator; the
five group objects, flat attribute access, (de)serialisation and migration are all derived from it.
``tests/test_config.py`` checks ``to_dict()`` equality with the reference for all 74 shipped JSONs.
"""
from __future__ import annotations

import json
from pathlib import Path
from types import SimpleNamespace

_REQUIRED = object()


def _gt0(name):
    def f(v):
        if v is not None and v <= 0:
            raise ValueError(f"{name} must be > 0, got {v}")
    return f


def _ge0(name):
    def f(v):
        if v is not None and v < 0:
            raise ValueError(f"{name} must be >= 0, got {v}")
    return f


def _one_of(name, choices):
    def f(v):
        if v is not None and v not in choices:
            raise ValueError(f"{name} must be one of {list(choices)}, got {v}")
    return f


def _even_pos(name):
    def f(v):
        _gt0(name)(v)
        if v % 2 != 0:
            raise ValueError(f"{name} must be even, got {v}")
    return f


def _rng(name, lo, hi, lo_open, hi_open):
    def f(v):
        if v is None:
            return
        bad = (v <= lo if lo_open else v < lo) or (v >= hi if hi_open else v > hi)
        if bad:
            raise ValueError(f"{name} must be in {'(' if lo_open else '['}{lo}, {hi}{')' if hi_open else ']'}, got {v}")
    return f


# key: (group, default, validator)
_FIELDS = {
    "batch_size": ("base", _REQUIRED, _gt0("batch_size")),
    "n_steps": ("base", _REQUIRED, _gt0("n_steps")),
    "base_lr": ("base", _REQUIRED, _gt0("base_lr")),
    "weight_decay": ("base", _REQUIRED, _ge0("weight_decay")),
    "seed": ("base", _REQUIRED, None),
    "noise_dimension": ("model", _REQUIRED, _gt0("noise_dimension")),
    "condition_dimension": ("model", _REQUIRED, _even_pos("condition_dimension")),
    "latent_dimension": ("model", _REQUIRED, _gt0("latent_dimension")),
    "num_blocks": ("model", _REQUIRED, _gt0("num_blocks")),
    "architecture": ("model", None, _one_of("architecture", ("mlp", "mlp_mixer", "convnet"))),
    "dataset": ("dataset", None, _one_of("dataset", ("mnist", "audio"))),
    "data_dir": ("dataset", None, None),
    "tokenization_strategy": ("dataset", None, _one_of("tokenization_strategy", ("mdct", "reshape"))),
    "tokenization_config": ("dataset", None, None),
    "method": ("method", None, _one_of("method", ("autoencoder", "flow_matching", "mean_flow", "improved_mean_flow"))),
    "use_improved_mean_flow": ("method", False, None),
    "gamma": ("method", None, _gt0("gamma")),
    "flow_ratio": ("method", None, _gt0("flow_ratio")),
    "c": ("method", None, _gt0("c")),
    "use_stop_gradient": ("method", None, None),
    "loss_weighting": ("method", None, _one_of("loss_weighting", ("uniform", "time_dependent", "learned"))),
    "loss_strategy": ("method", None, _one_of("loss_strategy", ("flow_matching", "mean_flow", "improved_mean_flow"))),
    "noise_schedule": ("method", None, _one_of("noise_schedule", ("linear", "uniform"))),
    "noise_min": ("method", None, _rng("noise_min", 0, 1, False, True)),
    "noise_max": ("method", None, _rng("noise_max", 0, 1, True, False)),
    "time_sampling": ("method", None, _one_of("time_sampling", ("uniform", "logit_normal", "mean_flow"))),
    "time_sampling_mean": ("method", None, None),
    "time_sampling_std": ("method", None, _gt0("time_sampling_std")),
    "time_sampling_data_proportion": ("method", None, _rng("time_sampling_data_proportion", 0, 1, False, False)),
    "use_weighted_loss": ("method", None, None),
    "sample_every": ("training", _REQUIRED, _gt0("sample_every")),
    "sample_seed": ("training", _REQUIRED, None),
    "sample_steps": ("training", _REQUIRED, _gt0("sample_steps")),
    "workdir": ("training", None, None),
    "checkpoint_step": ("training", None, _gt0("checkpoint_step")),
    "max_checkpoints_to_keep": ("training", None, _gt0("max_checkpoints_to_keep")),
}
GROUPS = ("base", "model", "dataset", "method", "training")
# keys the reference always serialises even when they hold their default
_ALWAYS = {"use_improved_mean_flow"}


class _Group(SimpleNamespace):
    def to_dict(self) -> dict:
        out = {}
        for k, v in vars(self).items():
            if v is None and k not in _ALWAYS:
                continue
            out[k] = str(v) if isinstance(v, Path) else v
        return out


def migrate_config_v1_to_v2(data: dict) -> dict:
    """Flat v1 dict -> {"config_version": "2.0", group: {...}} (configs/config.py:713-816)."""
    out = {g: {} for g in GROUPS}
    for key, (group, default, _) in _FIELDS.items():
        if key in data:
            out[group][key] = data[key]
        elif default is _REQUIRED:
            raise KeyError(key)
    out["method"].setdefault("use_improved_mean_flow", False)
    out["config_version"] = "2.0"
    for k in ("output_dir", "run_name"):           # deprecated, carried but never serialised
        if k in data:
            out[k] = data[k]
    return out


class TrainFlowConfig:
    """Five groups + read-only flat access (``cfg.batch_size`` == ``cfg.base.batch_size``)."""

    def __init__(self, base=None, model=None, dataset=None, method=None, training=None, output_dir=None,
                 run_name=None, config_version="2.0", **flat):
        # "dataset" and "method" name both a group and a flat key: strings mean the flat key
        if isinstance(dataset, str):
            flat["dataset"], dataset = dataset, None
        if isinstance(method, str):
            flat["method"], method = method, None
        if flat:
            if any(g is not None for g in (base, model, dataset, method, training)):
                raise TypeError("pass either group objects/dicts or flat keyword arguments, not both")
            unknown = sorted(set(flat) - set(_FIELDS))
            if unknown:
                raise TypeError(f"unknown config field(s) {unknown}")
            nested = migrate_config_v1_to_v2(flat)
            base, model, dataset, method, training = (nested[g] for g in GROUPS)
        groups = dict(base=base, model=model, dataset=dataset, method=method, training=training)
        for g, val in groups.items():
            val = {} if val is None else (dict(vars(val)) if not isinstance(val, dict) else dict(val))
            full = {}
            for key, (grp, default, _) in _FIELDS.items():
                if grp != g:
                    continue
                if key in val:
                    full[key] = val.pop(key)
                elif default is _REQUIRED:
                    raise TypeError(f"missing required config field '{key}' in group '{g}'")
                else:
                    full[key] = default
            if val:
                raise TypeError(f"unknown field(s) {sorted(val)} in config group '{g}'")
            if g == "training" and full["workdir"] is not None:
                full["workdir"] = Path(full["workdir"])
            object.__setattr__(self, "_" + g, _Group(**full))
        object.__setattr__(self, "output_dir", None if output_dir is None else Path(output_dir))
        object.__setattr__(self, "run_name", run_name)
        object.__setattr__(self, "config_version", config_version)
        if self._training.workdir is None and self.output_dir is not None and run_name is not None:
            self._training.workdir = self.output_dir / run_name      # deprecated pair -> workdir
        self.validate()

    # ---- groups / flat access
    base = property(lambda self: self._base)
    model = property(lambda self: self._model)
    training = property(lambda self: self._training)
    dataset_config = property(lambda self: self._dataset)
    method_config = property(lambda self: self._method)

    def __getattr__(self, name):
        spec = _FIELDS.get(name)
        if spec is None:
            raise AttributeError(name)
        return getattr(object.__getattribute__(self, "_" + spec[0]), name)

    def __setattr__(self, name, value):
        raise AttributeError(f"TrainFlowConfig is read-only (tried to set '{name}'); use merge_configs()")

    # ---- validation
    def validate(self) -> None:
        for key, (group, _, check) in _FIELDS.items():
            if check is not None:
                check(getattr(getattr(self, "_" + group), key))
        m = self._method
        if m.noise_min is not None and m.noise_max is not None and m.noise_min >= m.noise_max:
            raise ValueError(f"noise_min ({m.noise_min}) must be < noise_max ({m.noise_max})")
        if m.method == "improved_mean_flow" and not m.use_improved_mean_flow:
            raise ValueError("method='improved_mean_flow' requires use_improved_mean_flow=True")

    # ---- (de)serialisation
    def to_dict(self) -> dict:
        out = {"config_version": self.config_version}
        for g in GROUPS:
            out[g] = getattr(self, "_" + g).to_dict()
        return out

    def to_flat_dict(self) -> dict:
        return {k: (str(v) if isinstance(v, Path) else v) for g in GROUPS
                for k, v in vars(getattr(self, "_" + g)).items()}

    @classmethod
    def from_dict(cls, data: dict) -> "TrainFlowConfig":
        flat = "base" not in data and any(k in data for k in ("batch_size", "n_steps", "base_lr"))
        if flat or data.get("config_version", "1.0") == "1.0":
            data = migrate_config_v1_to_v2(data)
        if "base" not in data:
            raise ValueError("Invalid config format: expected hierarchical structure")
        return cls(**{g: data.get(g) for g in GROUPS}, output_dir=data.get("output_dir"),
                   run_name=data.get("run_name"), config_version=data.get("config_version", "2.0"))

    def get_schema(self) -> dict:
        return {g: {k: type(v).__name__ for k, v in vars(getattr(self, "_" + g)).items()} for g in GROUPS}

    def get_documentation(self) -> str:
        lines = ["TrainFlowConfig"]
        for g in GROUPS:
            lines.append(f"[{g}]")
            lines += [f"  {k} = {v!r}" for k, v in vars(getattr(self, '_' + g)).items()]
        return "\n".join(lines)

    def __repr__(self):
        return f"TrainFlowConfig({self.to_dict()!r})"


def load_config_from_json(path) -> TrainFlowConfig:
    with open(path) as f:
        return TrainFlowConfig.from_dict(json.load(f))


def _deep_merge(a: dict, b: dict) -> dict:
    out = dict(a)
    for k, v in b.items():
        out[k] = _deep_merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


def merge_configs(base: TrainFlowConfig, override: dict) -> TrainFlowConfig:
    """Nested override dict; flat keys are routed to their group."""
    routed = {}
    for k, v in override.items():
        if k in GROUPS:
            routed.setdefault(k, {}).update(v)
        elif k in _FIELDS:
            routed.setdefault(_FIELDS[k][0], {})[k] = v
        else:
            raise KeyError(f"unknown config key '{k}'")
    return TrainFlowConfig.from_dict(_deep_merge(base.to_dict(), routed))


def diff_configs(a: TrainFlowConfig, b: TrainFlowConfig) -> dict:
    fa, fb = a.to_flat_dict(), b.to_flat_dict()
    changed = {k: {"old": fa[k], "new": fb[k]} for k in fa if fa[k] != fb.get(k) and fa[k] is not None and fb.get(k) is not None}
    added = sorted(k for k in fb if fa.get(k) is None and fb[k] is not None)
    removed = sorted(k for k in fa if fb.get(k) is None and fa[k] is not None)
    return {"changed": changed, "added": added, "removed": removed}


def create_mnist_config(**overrides) -> TrainFlowConfig:
    d = dict(batch_size=128, n_steps=5000, base_lr=1e-4, weight_decay=1e-4, seed=42, noise_dimension=784,
             condition_dimension=128, latent_dimension=256, num_blocks=8, dataset="mnist", sample_every=500,
             sample_seed=42, sample_steps=50)
    d.update(overrides)
    return TrainFlowConfig(**d)


def create_audio_config(**overrides) -> TrainFlowConfig:
    d = dict(batch_size=128, n_steps=5000, base_lr=1e-4, weight_decay=1e-4, seed=42, noise_dimension=196608,
             condition_dimension=128, latent_dimension=256, num_blocks=8, dataset="audio",
             tokenization_strategy="mdct", tokenization_config={"window_size": 512, "hop_size": 256},
             sample_every=500, sample_seed=42, sample_steps=50)
    d.update(overrides)
    return TrainFlowConfig(**d)
