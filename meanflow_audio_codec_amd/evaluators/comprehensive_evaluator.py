"""NFE-sweep evaluator -- host mirror of ``evaluators/comprehensive_evaluator.py:26-265``.

``ComprehensiveEvaluator(checkpoint_path, config_path, dataset).evaluate(real_data, num_samples, n_steps_list,
batch_size, seed)`` returns the reference's result dictionary: ``config``, ``parameters``, ``memory_before/after`` and,
per NFE setting, ``inference_time`` (batch of one, 5 warm-up + 50 timed calls), ``mse`` and the dataset's metrics.

What runs where: generation is the hot path's decode (``sample`` = Heun integration through the HIP kernels, zero
latents as in the reference :114-119); with a tokenisation the generated tokens go back to the data domain through the
IMDCT kernel before they are compared; ``spectral_distance`` is two MDCT launches + one reduction on the device; MSE /
PSNR / SSIM are float64 host reductions (``metrics.py``).  PESQ / STOI need third-party packages and are recorded as
``None`` + the error text when absent, exactly as the reference does (:200-220).

Beyond the reference: ``one_step=True`` adds the key ``"1nfe"`` -- the true 1-NFE MeanFlow decode
(``sampling.one_step_decode``) that the reference only documents; ``timing_runs``/``timing_warmup`` expose the 50/5
constants (a 250-step sweep of the 13.75 B-parameter config is 500 forward passes per call).
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any

import numpy as np
import torch

from .. import ops
from ..configs import load_config_from_json
from ..preprocessing.tokenization_utils import compute_token_shape, create_tokenization_strategy
from ..trainers.time_sampling import PRNGKey
from . import audio_metrics, metrics
from .performance import count_parameters, inference_time, memory_usage
from .sampling import one_step_decode, sample


def _jsonable(obj):
    if isinstance(obj, dict):
        return {str(k): _jsonable(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_jsonable(v) for v in obj]
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu().tolist()
    if isinstance(obj, np.generic):
        return obj.item()
    if isinstance(obj, Path):
        return str(obj)
    return obj


class ComprehensiveEvaluator:
    def __init__(self, checkpoint_path: Path, config_path: Path | None = None, dataset: str = "mnist", *,
                 dtype=torch.float32, device: str = "cuda"):
        from ..trainers.train import load_flow_state
        self.checkpoint_path = Path(checkpoint_path)
        self.config_path = config_path
        self.dataset = dataset
        self.device = device
        if config_path is None:                       # <workdir>/config.json beside <workdir>/checkpoints/ (:50-60)
            config_path = self.checkpoint_path.parent.parent / "config.json"
            if not config_path.exists():
                raise ValueError("config_path must be provided or config.json must exist in checkpoint directory")
        self.config = load_config_from_json(config_path)
        self.model, self.state = load_flow_state(self.config, self.checkpoint_path, batch_size=self.config.batch_size,
                                                 dtype=dtype, device=device)
        self.tokenization = create_tokenization_strategy(self.config)
        self.token_shape = (compute_token_shape(self.tokenization, self.config.noise_dimension,
                                                self.config.dataset or "mnist")
                            if self.tokenization is not None else None)
        self.param_count = count_parameters(self.state.params)

    # -- generation -------------------------------------------------------------------------------------------
    def _to_data_domain(self, tokens: torch.Tensor) -> torch.Tensor:
        if self.tokenization is None:
            return tokens
        B = tokens.shape[0]
        return self.tokenization.detokenize(tokens.float().reshape(B, self.token_shape[0], self.token_shape[1]))

    def _generate(self, key: PRNGKey, latents: torch.Tensor, n_steps: int) -> torch.Tensor:
        """One batch of samples in model space; ``n_steps == 0`` selects the 1-NFE decode."""
        model, w = self.model, self.state.work
        if n_steps > 0:
            return sample(self.state.apply_fn, model.noise_dimension, w, key, latents=latents, n_steps=n_steps,
                          use_improved_mean_flow=self.config.use_improved_mean_flow, guidance_scale=1.0)
        B = latents.shape[0]
        eps = ops.randn(key.seed, 0x5a00 + (key.counter & 0xFF), 0, B, model.noise_dimension, device=latents.device)
        eps = eps if model.dtype == torch.float32 else ops.cast(eps, model.dtype)
        x0 = one_step_decode(model, w, eps, latents)
        return x0 if x0.dtype == torch.float32 else ops.cast(x0.contiguous(), torch.float32)

    # -- metrics ----------------------------------------------------------------------------------------------
    def _metrics(self, real: np.ndarray, gen: np.ndarray) -> dict[str, Any]:
        out: dict[str, Any] = {"mse": float(np.mean((real.astype(np.float64) - gen.astype(np.float64)) ** 2))}
        n = real.shape[0]
        if self.dataset == "mnist":
            if real.ndim == 2 and real.shape[1] == 784:
                real, gen = real.reshape(n, 28, 28), gen.reshape(n, 28, 28)
            out["psnr"] = metrics.psnr(gen, real)
            out["ssim"] = metrics.ssim(gen, real)
        elif self.dataset == "audio":
            for name, fn in (("pesq", audio_metrics.pesq_score), ("stoi", audio_metrics.stoi_score)):
                try:
                    out[name] = fn(real, gen, sample_rate=16000)
                except (ImportError, ValueError) as e:
                    out[name] = None
                    out[f"{name}_error"] = str(e)
            try:
                out["spectral_distance"] = audio_metrics.spectral_distance(real, gen, domain="mdct",
                                                                           device=self.device)
            except Exception as e:          # the reference records any failure here as None + text (:223-230)
                out["spectral_distance"] = None
                out["spectral_distance_error"] = str(e)
        return out

    # -- the sweep ----------------------------------------------------------------------------------------------
    def evaluate(self, real_data: np.ndarray, num_samples: int = 1000, n_steps_list: list[int] = [1, 10, 50, 250],
                 batch_size: int = 32, seed: int = 42, *, one_step: bool = False, timing_warmup: int = 5,
                 timing_runs: int = 50) -> dict[str, Any]:
        cfg = self.config
        results: dict[str, Any] = {
            "config": {"method": cfg.method, "architecture": cfg.architecture, "dataset": cfg.dataset,
                       "tokenization": cfg.tokenization_strategy},
            "parameters": self.param_count,
            "memory_before": memory_usage(),
            "nfe_results": {},
        }
        real_data = np.asarray(real_data.detach().cpu() if isinstance(real_data, torch.Tensor) else real_data)
        shape = tuple(getattr(self.state.model, "latent_shape", (cfg.latent_dimension,)))
        latents = torch.zeros((batch_size,) + shape, dtype=torch.float32, device=self.device)
        sweep = [(str(n), int(n)) for n in n_steps_list] + ([("1nfe", 0)] if one_step else [])
        key = PRNGKey(seed)
        for label, n_steps in sweep:
            print(f"Generating {num_samples} samples with {label} steps...")
            chunks = []
            for lo in range(0, num_samples, batch_size):
                nb = min(batch_size, num_samples - lo)
                key = key.next()
                toks = self._generate(key, latents[:nb], n_steps)
                chunks.append(self._to_data_domain(toks).float().cpu().numpy())
            generated = np.concatenate(chunks, axis=0)[:num_samples]
            generated = generated.reshape(generated.shape[0], -1) if real_data.ndim == 2 else generated

            tkey = key
            one = latents[:1]
            res: dict[str, Any] = {"inference_time": inference_time(lambda: self._generate(tkey, one, n_steps),
                                                                    num_warmup=timing_warmup, num_runs=timing_runs)}
            n_eval = min(len(real_data), len(generated))
            real_eval = real_data[:n_eval]
            gen_eval = generated[:n_eval]
            if gen_eval.shape != real_eval.shape:
                if gen_eval.ndim == real_eval.ndim == 2 and gen_eval.shape[1] > real_eval.shape[1]:
                    gen_eval = gen_eval[:, :real_eval.shape[1]]     # detokenised clips carry the MDCT right padding
                else:
                    raise ValueError(f"generated samples {gen_eval.shape} do not match real_data {real_eval.shape}")
            res.update(self._metrics(real_eval, gen_eval))
            results["nfe_results"][label] = res
        results["memory_after"] = memory_usage()
        return results

    def save_results(self, results: dict[str, Any], output_path: Path) -> None:
        """JSON, ``indent=2, sort_keys=True`` (:235-265)."""
        output_path = Path(output_path)
        output_path.parent.mkdir(parents=True, exist_ok=True)
        with output_path.open("w", encoding="utf-8") as f:
            json.dump(_jsonable(results), f, indent=2, sort_keys=True)
