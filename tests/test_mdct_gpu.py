"""GPU parity: HIP MDCT/IMDCT (through the C ABI) vs the oracle and the reference fixtures."""
import numpy as np
import pytest
import torch

from oracle import mdct_oracle as o

pytestmark = pytest.mark.gpu

CASES = [
    "mdct_n256_h128_t1024_s42", "mdct_n512_h256_t8192_s42", "mdct_n512_h512_t4096_s7",
    "mdct_n64_h32_b3_t1000_s3", "mdct_n256_h128_t100_s5", "mdct_n512_h256_b2_t784_s11",
    "mdct_n128_h32_t2048_s9",
]


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("name", CASES)
def test_against_reference_fixture_and_f64_truth(golden_dir, name):
    from meanflow_audio_codec_amd.preprocessing import imdct, mdct
    d = np.load(golden_dir / f"{name}.npz")
    x, X_ref, xr_ref, N, hop = d["x"], d["X_ref"], d["xr_ref"], int(d["N"]), int(d["hop"])
    X = mdct(_dev(x), N, hop).cpu().numpy()
    assert X.shape == o.mdct_f64(x, N, hop).shape  # leading dims preserved (mdct.py:287)
    X64 = o.mdct_f64(x, N, hop)
    # (i) float64 truth: fp32 FFT accuracy
    assert np.abs(X - X64).max() <= 2e-5 * max(1.0, np.abs(X64).max())
    # (ii) reference float32 fixture within ITS error envelope
    atol = 2e-3 if N <= 256 else 1e-2
    assert np.abs(X.reshape(X_ref.shape) - X_ref).max() < atol
    # (iv) inverse alone vs the reference fixture at the reference's own tolerance
    xr = imdct(_dev(X_ref.reshape(X.shape)), N, hop).cpu().numpy()
    np.testing.assert_allclose(xr.reshape(xr_ref.shape), xr_ref, rtol=1e-4, atol=1e-3)
    xr64 = o.imdct_f64(X_ref.astype(np.float64).reshape(X.shape), N, hop)
    assert np.abs(xr - xr64).max() <= 2e-5 * max(1.0, np.abs(xr64).max())


@pytest.mark.parametrize("N,hop,T,B", [(8, 4, 64, 2), (16, 16, 200, 1), (32, 8, 333, 3), (1024, 512, 9000, 2),
                                       (2048, 1024, 20000, 1), (4096, 2048, 30000, 1), (128, 200, 3000, 2),
                                       (64, 1, 300, 1), (512, 256, 16384, 4)])
def test_pow2_windows_vs_oracle(N, hop, T, B):
    from meanflow_audio_codec_amd.preprocessing import imdct, mdct
    rng = np.random.default_rng(N + hop + T)
    x = rng.standard_normal((B, T)).astype(np.float32)
    X = mdct(_dev(x), N, hop).cpu().numpy()
    X64 = o.mdct_f64(x, N, hop)
    assert X.shape == X64.shape
    assert np.abs(X - X64).max() <= 3e-5 * np.abs(X64).max()
    xr = imdct(_dev(X64.astype(np.float32)), N, hop).cpu().numpy()
    xr64 = o.imdct_f64(X64.astype(np.float32), N, hop)
    assert xr.shape == xr64.shape
    assert np.abs(xr - xr64).max() <= 3e-5 * max(1.0, np.abs(xr64).max())


@pytest.mark.parametrize("hop,T,B", [(256, 196608, 3), (256, 100, 2), (256, 784, 5), (256, 40001, 2), (512, 30000, 2),
                                     (128, 9000, 3), (384, 12345, 2), (64, 5000, 1), (4, 3000, 1), (260, 7000, 2),
                                     (256, 16 * 256 + 512, 1), (256, 17 * 256 + 512, 2), (256, 96 * 256 + 512, 2),
                                     (256, 97 * 256 + 512, 1)])
def test_n512_kernels_vs_oracle(hop, T, B):
    """The N = 512 kernels (csrc/mdct512.hip: register-resident 16 x 16 FFT, LDS overlap-add with carry) against the
    float64 oracle: power-of-two and other hops, hop == N, clips shorter than a window, frame counts around the
    16-frame iteration and the 96-frame segment boundaries, and rows that are NOT 16-byte aligned (views into a wider
    buffer with an odd leading dimension: the scalar-load variant of the forward kernel)."""
    from meanflow_audio_codec_amd import _lib
    from meanflow_audio_codec_amd.preprocessing import imdct, mdct
    rng = np.random.default_rng(hop * 7 + T)
    x = rng.standard_normal((B, T)).astype(np.float32)
    X64 = o.mdct_f64(x, 512, hop)
    X = mdct(_dev(x), 512, hop).cpu().numpy()
    assert X.shape == X64.shape
    assert np.abs(X - X64).max() <= 2e-5 * max(1.0, np.abs(X64).max())
    xr64 = o.imdct_f64(X64.astype(np.float32), 512, hop)
    xr = imdct(_dev(X64.astype(np.float32)), 512, hop).cpu().numpy()
    assert xr.shape == xr64.shape
    assert np.abs(xr - xr64).max() <= 2e-5 * max(1.0, np.abs(xr64).max())
    # unaligned rows through the C ABI: ldx = T + 3 (forward), ldy = out_len + 1 (inverse)
    L = _lib.lib()
    wide = torch.zeros(B, T + 3, device="cuda")
    wide[:, :T] = _dev(x)
    nf = X64.shape[1]
    Xo = torch.empty(B, nf, 512, device="cuda")
    _lib.check(L.mfc_mdct_fwd(wide.data_ptr() + 0, B, T, T + 3, 512, hop, Xo.data_ptr(), _lib.stream_ptr()), "fwd")
    assert np.abs(Xo.cpu().numpy() - X64).max() <= 2e-5 * max(1.0, np.abs(X64).max())
    out_len = xr64.shape[1]
    yo = torch.full((B, out_len + 1), 7.0, device="cuda")
    Xin = _dev(X64.astype(np.float32))
    _lib.check(L.mfc_mdct_inv(Xin.data_ptr(), B, nf, 512, hop, yo.data_ptr(), out_len + 1, _lib.stream_ptr()), "inv")
    assert np.abs(yo[:, :out_len].cpu().numpy() - xr64).max() <= 2e-5 * max(1.0, np.abs(xr64).max())
    assert bool((yo[:, out_len] == 7.0).all())           # nothing written past a row


@pytest.mark.parametrize("N,hop,T", [(576, 288, 5000), (12, 6, 100), (7, 3, 50), (100, 50, 777)])
def test_non_pow2_windows_direct_kernel(N, hop, T):
    """The reference's default window is 576 (mdct.py DEFAULT_WINDOW_SIZE)."""
    from meanflow_audio_codec_amd.preprocessing import imdct, mdct
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, T)).astype(np.float32)
    X = mdct(_dev(x), N, hop).cpu().numpy()
    X64 = o.mdct_f64(x, N, hop)
    assert np.abs(X - X64).max() <= 1e-4 * np.abs(X64).max()
    xr = imdct(_dev(X64.astype(np.float32)), N, hop).cpu().numpy()
    xr64 = o.imdct_f64(X64.astype(np.float32), N, hop)
    assert np.abs(xr - xr64).max() <= 1e-4 * max(1.0, np.abs(xr64).max())


def test_full_size_round_trip_and_checksums(golden_dir):
    """BASELINE config #4 shape: B clips of T=196608, N=512, hop=256 -> [B,767,512].
    Size-independent properties: round trip == 2x on the interior,
    linearity, and the reference's row checksums for clip 0."""
    from meanflow_audio_codec_amd.preprocessing import imdct, mdct
    d = np.load(golden_dir / "mdct_n512_h256_t196608_s42_checksums.npz")
    np.random.seed(42)
    x0 = np.random.randn(196608).astype(np.float32)
    B = 16
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 196608, generator=g)
    x[0] = torch.from_numpy(x0)
    xd = x.cuda()
    X = mdct(xd, 512, 256)
    assert X.shape == (B, 767, 512)
    Xn = X[0].double().cpu().numpy()
    assert np.abs(Xn.sum(-1) - d["X_row_sum"]).max() < 0.2
    assert np.abs((Xn ** 2).sum(-1) / d["X_row_sumsq"] - 1).max() < 1e-3
    xr = imdct(X, 512, 256)
    assert xr.shape == (B, 197120)
    lo, hi = 1024, 766 * 256
    err = (xr[:, lo:hi] - 2.0 * xd[:, lo:hi]).abs().max().item()
    assert err <= 1e-5 * xd.abs().max().item() * 2 * 4, err
    # linearity
    y = torch.randn(B, 196608, generator=g).cuda()
    lhs = mdct(0.5 * xd - 2.0 * y, 512, 256)
    rhs = 0.5 * X - 2.0 * mdct(y, 512, 256)
    assert (lhs - rhs).abs().max().item() < 1e-3


def test_error_behaviour():
    from meanflow_audio_codec_amd.preprocessing import MDCTConfig, imdct, mdct
    with pytest.raises(TypeError):
        mdct(np.zeros(10), 8)
    with pytest.raises(ValueError):
        mdct(torch.tensor(1.0).cuda(), 8)
    with pytest.raises(ValueError):
        mdct(torch.zeros(10).cuda(), 0)
    with pytest.raises(ValueError):
        imdct(torch.zeros(8).cuda(), 8)
    with pytest.raises(ValueError):
        MDCTConfig(window_size=8, hop_size=0)
    cfg = MDCTConfig(window_size=16)
    assert cfg.hop_size == 8
    X = mdct(torch.zeros(3, 2, 100).cuda(), config=cfg)
    assert X.shape == (3, 2, 11, 16)
    assert imdct(X, config=cfg).shape == (3, 2, 10 * 8 + 32)


def test_spectral_distance_mdct_domain():
    """evaluators/audio_metrics.py:112-170 (float64 on the host there): per-sample RMS difference of the MDCT
    coefficients, batch mean -- GPU fp32 vs the float64 oracle transform."""
    from meanflow_audio_codec_amd.evaluators import spectral_distance
    rng = np.random.default_rng(7)
    ref = rng.standard_normal((3, 8192)).astype(np.float32)
    deg = (ref + 0.05 * rng.standard_normal(ref.shape)).astype(np.float32)
    want = float(np.mean([np.sqrt(np.mean((o.mdct_f64(ref[i:i + 1], 512, 256).ravel()
                                           - o.mdct_f64(deg[i:i + 1], 512, 256).ravel()) ** 2)) for i in range(3)]))
    got = spectral_distance(ref, deg)
    assert abs(got - want) <= 1e-4 * want, (got, want)
    one = spectral_distance(ref[0], deg[0], window_size=256, hop_size=64)
    w1 = float(np.sqrt(np.mean((o.mdct_f64(ref[:1], 256, 64).ravel() - o.mdct_f64(deg[:1], 256, 64).ravel()) ** 2)))
    assert abs(one - w1) <= 1e-4 * w1
    assert spectral_distance(ref, ref) == 0.0
    with pytest.raises(ValueError, match="Shape mismatch"):
        spectral_distance(ref, deg[:, :100])
    with pytest.raises(ValueError, match="Invalid domain"):
        spectral_distance(ref, deg, domain="stft")
