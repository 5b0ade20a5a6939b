"""GPU: the NFE-sweep evaluator end to end -- checkpoint written by train_flow, sweep through the HIP
sampler, metrics in the reference's result layout."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _train(workdir, **kw):
    from meanflow_audio_codec_amd.configs import TrainFlowConfig
    from meanflow_audio_codec_amd.trainers.train import synthetic_iterator, train_flow
    base = dict(batch_size=8, n_steps=2, sample_every=100, sample_seed=3, sample_steps=1, base_lr=1e-3,
                weight_decay=1e-4, seed=0, noise_dimension=784, condition_dimension=16, latent_dimension=8,
                num_blocks=1, dataset="mnist", architecture="mlp", use_improved_mean_flow=True,
                method="improved_mean_flow", loss_strategy="improved_mean_flow", workdir=workdir)
    base.update(kw)
    cfg = TrainFlowConfig(**base)
    state, _ = train_flow(cfg, synthetic_iterator(cfg))
    return cfg, state, workdir / "checkpoints" / "step_00002.msgpack"


def test_mnist_sweep_layout_and_metrics(tmp_path):
    from meanflow_audio_codec_amd.evaluators import metrics
    from meanflow_audio_codec_amd.evaluators.comprehensive_evaluator import ComprehensiveEvaluator
    from meanflow_audio_codec_amd.evaluators.sampling import sample
    from meanflow_audio_codec_amd.trainers.time_sampling import PRNGKey
    cfg, state, ckpt = _train(tmp_path / "run")
    ev = ComprehensiveEvaluator(ckpt, dataset="mnist")            # config.json found beside checkpoints/
    assert ev.param_count["total"] == sum(v.numel() for v in state.params.values())
    for k, v in state.params.items():
        assert torch.equal(ev.state.params[k], v), k
    real = np.random.default_rng(0).uniform(-1, 1, size=(12, 784)).astype(np.float32)
    res = ev.evaluate(real, num_samples=12, n_steps_list=[1, 2], batch_size=8, seed=7, one_step=True,
                      timing_warmup=1, timing_runs=2)
    assert res["config"] == {"method": "improved_mean_flow", "architecture": "mlp", "dataset": "mnist",
                             "tokenization": None}
    assert set(res["nfe_results"]) == {"1", "2", "1nfe"}
    assert "gpu_memory_used_mb" in res["memory_before"] and "gpu_memory_total_mb" in res["memory_after"]
    for r in res["nfe_results"].values():
        assert set(r) == {"inference_time", "mse", "psnr", "ssim"}
        assert r["inference_time"]["min"] > 0 and np.isfinite([r["mse"], r["psnr"], r["ssim"]]).all()

    # the sweep's numbers are those of sample() with the evaluator's key sequence (seed -> next() per batch)
    key = PRNGKey(7)
    chunks = []
    for nb in (8, 4):
        key = key.next()
        lat = torch.zeros(nb, cfg.latent_dimension, device="cuda")
        chunks.append(sample(ev.state.apply_fn, 784, ev.state.work, key, latents=lat, n_steps=1,
                             use_improved_mean_flow=True).cpu().numpy())
    gen = np.concatenate(chunks)
    r1 = res["nfe_results"]["1"]
    assert r1["mse"] == pytest.approx(float(np.mean((real.astype(np.float64) - gen) ** 2)), rel=1e-6)
    assert r1["psnr"] == pytest.approx(metrics.psnr(gen.reshape(12, 28, 28), real.reshape(12, 28, 28)), rel=1e-6)
    assert r1["ssim"] == pytest.approx(metrics.ssim(gen.reshape(12, 28, 28), real.reshape(12, 28, 28)), rel=1e-6)

    out = tmp_path / "eval" / "results.json"
    ev.save_results(res, out)
    back = json.loads(out.read_text())
    assert back["nfe_results"]["2"]["mse"] == res["nfe_results"]["2"]["mse"]
    assert back["parameters"]["total"] == ev.param_count["total"]

    with pytest.raises(ValueError, match="config_path must be provided"):
        (tmp_path / "lonely" / "checkpoints").mkdir(parents=True)
        lone = tmp_path / "lonely" / "checkpoints" / "step_00002.msgpack"
        lone.write_bytes(ckpt.read_bytes())
        ComprehensiveEvaluator(lone)


def test_audio_sweep_goes_through_imdct_and_spectral_distance(tmp_path):
    from meanflow_audio_codec_amd.evaluators import audio_metrics
    from meanflow_audio_codec_amd.evaluators.comprehensive_evaluator import ComprehensiveEvaluator
    T = 1024
    cfg, state, ckpt = _train(tmp_path / "arun", dataset="audio", noise_dimension=T, tokenization_strategy="mdct",
                              tokenization_config={"window_size": 64, "hop_size": 32})
    ev = ComprehensiveEvaluator(ckpt, config_path=tmp_path / "arun" / "config.json", dataset="audio")
    assert ev.token_shape == (31, 64) and ev.model.noise_dimension == 31 * 64
    real = (0.1 * np.random.default_rng(1).standard_normal((6, T))).astype(np.float32)
    res = ev.evaluate(real, num_samples=6, n_steps_list=[2], batch_size=4, seed=1, timing_warmup=1, timing_runs=1)
    r = res["nfe_results"]["2"]
    # pesq / pystoi are not installed in this image: recorded as None + message, as the reference does
    assert r["pesq"] is None and "pesq" in r["pesq_error"]
    assert r["stoi"] is None and "pystoi" in r["stoi_error"]
    assert r["spectral_distance"] is not None and np.isfinite(r["spectral_distance"]) and r["spectral_distance"] > 0
    assert res["config"]["tokenization"] == "mdct"
    # detokenised length: (n_frames-1)*hop + 2N = 1088 >= T; the comparison is over the first T samples
    from meanflow_audio_codec_amd.trainers.time_sampling import PRNGKey
    toks = ev._generate(PRNGKey(1).next(),
                        torch.zeros(4, cfg.latent_dimension, device="cuda"), 2)
    audio = ev._to_data_domain(toks)
    assert audio.shape == (4, 1088)
    gen0 = audio[:, :T].cpu().numpy()
    assert audio_metrics.spectral_distance(real[:4], gen0) > 0
