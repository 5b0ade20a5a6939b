"""CPU: evaluator metrics against vectors produced by the reference's own ``evaluators/metrics.py``
(``tests/golden/gen_metrics_golden.py``), plus properties of the reference-compatible quirks documented in the module."""
import pathlib

import numpy as np
import pytest

from meanflow_audio_codec_amd.evaluators import metrics

G = np.load(pathlib.Path(__file__).parent / "golden" / "eval_metrics_golden.npz")
RTOL = 1e-9      # float64 on both sides; differences are summation order only


@pytest.mark.parametrize("name", ["pm1", "u01", "big", "rgb", "single"])
def test_psnr_ssim_match_reference(name):
    pred, target = G[f"img_{name}_pred"], G[f"img_{name}_target"]
    assert metrics.psnr(pred, target) == pytest.approx(float(G[f"psnr_{name}"]), rel=RTOL)
    assert metrics.ssim(pred, target) == pytest.approx(float(G[f"ssim_{name}"]), rel=RTOL)


def test_psnr_ssim_options_match_reference():
    pred, target = G["img_pm1_pred"], G["img_pm1_target"]
    assert metrics.psnr(pred, target, data_range=1.0) == pytest.approx(float(G["psnr_pm1_range1"]), rel=RTOL)
    assert metrics.ssim(pred, target, win_size=7, gaussian_weights=False) == pytest.approx(
        float(G["ssim_pm1_uniform_w7"]), rel=RTOL)
    # even window -> 9; sigma has no effect in the reference (window = filtered constant)
    assert metrics.ssim(pred, target, win_size=8, sigma=3.0) == pytest.approx(float(G["ssim_pm1_w8_sigma3"]), rel=1e-8)
    assert metrics.ssim(pred, target, win_size=8, sigma=3.0) == metrics.ssim(pred, target, win_size=9, sigma=0.5)
    p01, t01 = G["img_u01_pred"], G["img_u01_target"]
    assert metrics.ssim(p01, t01, data_range=1.0, k1=0.02, k2=0.05) == pytest.approx(float(G["ssim_u01_range1_k"]),
                                                                                     rel=RTOL)
    assert metrics.psnr(target, target) == float("inf") == float(G["psnr_identical"])


def test_psnr_ssim_errors():
    a = np.zeros((2, 28, 28))
    with pytest.raises(ValueError, match="Shape mismatch"):
        metrics.psnr(a, a[:1])
    with pytest.raises(ValueError, match="Shape mismatch"):
        metrics.ssim(a, a[:, :20])
    with pytest.raises(ValueError, match="Invalid data_range"):
        metrics.psnr(a + 1.0, a, data_range=0.0)
    with pytest.raises(ValueError, match="smaller than"):
        metrics.ssim(np.ones((1, 8, 8)), np.zeros((1, 8, 8)))


def test_frechet_distance_matches_reference():
    fd = metrics.frechet_distance(G["fd_mu1"], G["fd_sigma1"], G["fd_mu2"], G["fd_sigma2"])
    assert fd == pytest.approx(float(G["fd"]), rel=1e-8)
    same = metrics.frechet_distance(G["fd_mu1"], G["fd_sigma1"], G["fd_mu1"], G["fd_sigma1"])
    assert same == pytest.approx(float(G["fd_same"]), rel=1e-7)
    rd = metrics.frechet_distance(G["fd_mu1"], G["fd_sigma1"], G["fd_mu3"], G["fd_sigma3"])
    assert rd == pytest.approx(float(G["fd_rankdef"]), rel=1e-8)


def test_frechet_distance_true_form():
    # the non-compatible mode is the actual Frechet distance: 0 for identical Gaussians, closed form for isotropic ones
    mu, S = G["fd_mu1"], G["fd_sigma1"]
    assert abs(metrics.frechet_distance(mu, S, mu, S, reference_compatible=False)) < 1e-8
    d = mu.shape[0]
    got = metrics.frechet_distance(np.zeros(d), 4.0 * np.eye(d), np.ones(d), 9.0 * np.eye(d), reference_compatible=False)
    assert got == pytest.approx(d * 1.0 + d * (2.0 - 3.0) ** 2, rel=1e-5)
    # and the reference's value for identical inputs is not 0 -- the documented defect this build reproduces
    assert float(G["fd_same"]) > 0.5


def test_kid_matches_reference():
    real, fake = G["kid_real"], G["kid_fake"]
    assert metrics.kid_score(real, fake) == pytest.approx(float(G["kid_default"]), rel=1e-9)
    assert metrics.kid_score(real, fake, subset_size=32, num_subsets=7, seed=5) == pytest.approx(
        float(G["kid_s32_n7_seed5"]), rel=1e-9)
    assert metrics.kid_score(real[:40], fake[:25], subset_size=100, num_subsets=3, seed=1) == pytest.approx(
        float(G["kid_small"]), rel=1e-9)
    assert metrics.kid_score(real, real, subset_size=50, num_subsets=4, seed=2) == pytest.approx(
        float(G["kid_same"]), rel=1e-9)
    with pytest.raises(ValueError, match="subset_size must be >= 2"):
        metrics.kid_score(real[:1], fake)


def test_performance_helpers_on_host():
    import torch
    from meanflow_audio_codec_amd.evaluators import performance as perf
    calls = []
    out = perf.inference_time(lambda a, b=0: calls.append(a + b), 1, b=2, num_warmup=2, num_runs=5)
    assert len(calls) == 7 and set(out) == {"mean", "std", "min", "max", "total"}
    assert out["min"] <= out["mean"] <= out["max"] and out["total"] == pytest.approx(5 * out["mean"])
    params = {"blocks_0/w/kernel": torch.zeros(3, 4), "blocks_0/w/bias": torch.zeros(4),
              "nested": {"a": np.zeros((2, 2)), "b": {"c": torch.zeros(5)}}}
    cnt = perf.count_parameters(params)
    assert cnt["total"] == 12 + 4 + 4 + 5 == cnt["trainable"] and cnt["total_millions"] == pytest.approx(25e-6)
    assert cnt["by_module"]["nested/b/c"] == 5 and cnt["by_module"]["blocks_0/w/kernel"] == 12
    t = perf.TrainingTimer()
    with pytest.raises(RuntimeError):
        t.elapsed()
    with t:
        pass
    assert t.elapsed() >= 0
    with perf.memory_profiler() as p:
        pass
    assert set(p) == {"before", "after", "delta"}
    assert all(k.endswith("_delta_mb") or k.endswith("_delta_percent") for k in p["delta"])
