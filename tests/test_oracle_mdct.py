"""CPU: pin the MDCT oracle against the reference's own golden vectors.

Fixtures come from the reference's numpy baseline (test/test_mdct_utils.py), the
exact case of test/test_mdct.py:13-56 included; tolerances are the reference's
own (rtol 1e-4 / atol 1e-3) for the float32 restatement and the measured
float32-oracle error envelope for the float64 truth.
"""
import numpy as np
import pytest

from oracle import mdct_oracle as o

CASES = [
    "mdct_n256_h128_t1024_s42", "mdct_n512_h256_t8192_s42", "mdct_n512_h512_t4096_s7",
    "mdct_n64_h32_b3_t1000_s3", "mdct_n256_h128_t100_s5", "mdct_n512_h256_b2_t784_s11",
    "mdct_n128_h32_t2048_s9",
]


def _load(golden_dir, name):
    d = np.load(golden_dir / f"{name}.npz")
    return d["x"], d["X_ref"], d["xr_ref"], int(d["N"]), int(d["hop"])


@pytest.mark.parametrize("name", CASES)
def test_f32_restatement_matches_reference(golden_dir, name):
    x, X_ref, xr_ref, N, hop = _load(golden_dir, name)
    X = o.mdct_f32(x, N, hop).reshape(X_ref.shape)
    np.testing.assert_allclose(X, X_ref, rtol=1e-4, atol=1e-3)  # test_mdct.py:34-36
    xr = o.imdct_f32(X_ref, N, hop).reshape(xr_ref.shape)
    np.testing.assert_allclose(xr, xr_ref, rtol=1e-4, atol=1e-3)  # test_mdct.py:51-56
    # stronger: same arithmetic -> identical up to BLAS summation order
    assert np.abs(X - X_ref).max() < 1e-4


@pytest.mark.parametrize("name", CASES)
def test_f64_truth_within_reference_error_envelope(golden_dir, name):
    x, X_ref, xr_ref, N, hop = _load(golden_dir, name)
    X = o.mdct_f64(x, N, hop).reshape(X_ref.shape)
    atol = 2e-3 if N <= 256 else 1e-2  # float32 basis angle error grows with N
    assert np.abs(X - X_ref).max() < atol
    xr = o.imdct_f64(X_ref.astype(np.float64), N, hop).reshape(xr_ref.shape)
    np.testing.assert_allclose(xr, xr_ref, rtol=1e-4, atol=1e-3)


def test_full_clip_checksums(golden_dir):
    d = np.load(golden_dir / "mdct_n512_h256_t196608_s42_checksums.npz")
    np.random.seed(int(d["seed"]))
    x = np.random.randn(int(d["T"])).astype(np.float32)
    X = o.mdct_f64(x, 512, 256)
    assert X.shape == (767, 512)
    # row / column sums of the reference's float32 output; envelope ~ sqrt(N)*1e-3
    assert np.abs(X.sum(-1) - d["X_row_sum"]).max() < 0.2
    assert np.abs(X.sum(-2) - d["X_col_sum"]).max() < 0.3
    assert np.abs((X ** 2).sum(-1) / d["X_row_sumsq"] - 1).max() < 1e-3


@pytest.mark.parametrize("N,hop,T", [(64, 32, 1000), (512, 256, 4096), (128, 128, 1024), (128, 32, 2048)])
def test_round_trip_gain(N, hop, T):
    """imdct(mdct(x)) == (N/hop) x on the interior."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal(T)
    xr = o.imdct_f64(o.mdct_f64(x, N, hop), N, hop)
    nf = o.num_frames(T, N, hop)
    lo, hi = 2 * N, (nf - 1) * hop
    assert hi > lo
    np.testing.assert_allclose(xr[lo:hi], (N / hop) * x[lo:hi], atol=1e-9)


def test_tokenize_shapes_and_errors():
    x = np.zeros((2, 1000, 2))
    tok = o.mdct_tokenize(x, 64, 32)
    assert tok.shape == (2, 30, 128)
    back = o.mdct_detokenize(tok, 64, 32)
    assert back.shape == (2, 29 * 32 + 128, 2)
    with pytest.raises(ValueError):
        o.mdct_tokenize(np.zeros(5), 64)
    with pytest.raises(ValueError):
        o.mdct_detokenize(np.zeros((2, 3, 65)), 64)
