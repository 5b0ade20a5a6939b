"""Generate golden vectors for the evaluator metrics.

Run ONCE in the build container (needs /root/reference, numpy + scipy only):

    python tests/golden/gen_metrics_golden.py

Loads ``/root/reference/meanflow_audio_codec/evaluators/metrics.py`` by file path (the module imports only numpy and
scipy; the package ``__init__`` is not executed) and stores seeded inputs + the reference's outputs.  Only the
resulting ``.npz`` data file is committed; no reference source travels.
"""
import importlib.util
import pathlib

import numpy as np

REF = pathlib.Path("/root/reference/meanflow_audio_codec/evaluators/metrics.py")
OUT = pathlib.Path(__file__).parent


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_eval_metrics", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_ref()
    rng = np.random.default_rng(2024)
    out = {}

    # PSNR / SSIM: MNIST-like images in [-1, 1], [0, 1] and an unnormalised range; batched, single, multi-channel
    img_pm1 = np.tanh(rng.standard_normal((6, 28, 28)))
    deg_pm1 = np.clip(img_pm1 + 0.1 * rng.standard_normal(img_pm1.shape), -1, 1)
    img_01 = rng.random((4, 28, 28))
    deg_01 = np.clip(img_01 + 0.05 * rng.standard_normal(img_01.shape), 0, 1)
    img_big = 40.0 * rng.standard_normal((3, 32, 20)) + 7.0
    deg_big = img_big + 3.0 * rng.standard_normal(img_big.shape)
    img_rgb = rng.random((2, 24, 24, 3)) * 2 - 1
    deg_rgb = img_rgb + 0.2 * rng.standard_normal(img_rgb.shape)
    single = rng.random((28, 28))
    single_deg = single + 0.1 * rng.standard_normal(single.shape)
    for name, (a, b) in dict(pm1=(img_pm1, deg_pm1), u01=(img_01, deg_01), big=(img_big, deg_big),
                             rgb=(img_rgb, deg_rgb), single=(single, single_deg)).items():
        out[f"img_{name}_target"] = a
        out[f"img_{name}_pred"] = b
        out[f"psnr_{name}"] = np.float64(ref.psnr(b, a))
        out[f"ssim_{name}"] = np.float64(ref.ssim(b, a))
    out["psnr_pm1_range1"] = np.float64(ref.psnr(deg_pm1, img_pm1, data_range=1.0))
    out["ssim_pm1_uniform_w7"] = np.float64(ref.ssim(deg_pm1, img_pm1, win_size=7, gaussian_weights=False))
    out["ssim_pm1_w8_sigma3"] = np.float64(ref.ssim(deg_pm1, img_pm1, win_size=8, sigma=3.0))
    out["ssim_u01_range1_k"] = np.float64(ref.ssim(deg_01, img_01, data_range=1.0, k1=0.02, k2=0.05))
    out["psnr_identical"] = np.float64(ref.psnr(img_pm1, img_pm1))

    # Frechet distance: two Gaussians fitted to embeddings
    ea = rng.standard_normal((300, 24))
    eb = 1.3 * rng.standard_normal((280, 24)) @ (np.eye(24) + 0.1 * rng.standard_normal((24, 24))) + 0.4
    mu1, mu2 = ea.mean(0), eb.mean(0)
    s1, s2 = np.cov(ea, rowvar=False), np.cov(eb, rowvar=False)
    out.update(fd_mu1=mu1, fd_mu2=mu2, fd_sigma1=s1, fd_sigma2=s2)
    out["fd"] = np.float64(ref.frechet_distance(mu1, s1, mu2, s2))
    out["fd_same"] = np.float64(ref.frechet_distance(mu1, s1, mu1, s1))
    # rank-deficient covariance (fewer samples than dimensions): exercises the eigenvalue floor
    ec = rng.standard_normal((10, 24))
    s3 = np.cov(ec, rowvar=False)
    out.update(fd_sigma3=s3, fd_mu3=ec.mean(0))
    out["fd_rankdef"] = np.float64(ref.frechet_distance(mu1, s1, ec.mean(0), s3))

    # KID (the subset draws come from numpy's default_rng(seed): part of the contract)
    out.update(kid_real=ea, kid_fake=eb)
    out["kid_default"] = np.float64(ref.kid_score(ea, eb))
    out["kid_s32_n7_seed5"] = np.float64(ref.kid_score(ea, eb, subset_size=32, num_subsets=7, seed=5))
    out["kid_small"] = np.float64(ref.kid_score(ea[:40], eb[:25], subset_size=100, num_subsets=3, seed=1))
    out["kid_same"] = np.float64(ref.kid_score(ea, ea, subset_size=50, num_subsets=4, seed=2))

    np.savez_compressed(OUT / "eval_metrics_golden.npz", **out)
    print("wrote", OUT / "eval_metrics_golden.npz", {k: float(v) for k, v in out.items() if np.ndim(v) == 0})


if __name__ == "__main__":
    main()
