"""Generate MDCT golden vectors from the reference's own numpy baseline.

Run ONCE in the build container (needs /root/reference, numpy only):

    python tests/golden/gen_mdct_golden.py

Loads ``/root/reference/test/test_mdct_utils.py`` by file path (the module is
numpy-only) and stores inputs + the reference's float32 outputs.  Only the
resulting ``.npz`` data files are committed; no reference source travels.
"""
import importlib.util
import pathlib

import numpy as np

REF = pathlib.Path("/root/reference/test/test_mdct_utils.py")
OUT = pathlib.Path(__file__).parent


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_mdct_utils", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_ref()
    cases = {
        # the exact case of test/test_mdct.py:13-56
        "mdct_n256_h128_t1024_s42": dict(N=256, hop=128, shape=(1024,), seed=42),
        # shipped tokenization_config (window 512, hop 256)
        "mdct_n512_h256_t8192_s42": dict(N=512, hop=256, shape=(8192,), seed=42),
        # hop == window: reconstruction gain 1
        "mdct_n512_h512_t4096_s7": dict(N=512, hop=512, shape=(4096,), seed=7),
        # batched, ragged length (T not a multiple of hop), small window
        "mdct_n64_h32_b3_t1000_s3": dict(N=64, hop=32, shape=(3, 1000), seed=3),
        # T < N  -> a single zero-padded frame
        "mdct_n256_h128_t100_s5": dict(N=256, hop=128, shape=(100,), seed=5),
        # MNIST row of BASELINE config #3: T=784, N=512, hop=256 -> 2 frames
        "mdct_n512_h256_b2_t784_s11": dict(N=512, hop=256, shape=(2, 784), seed=11),
        # hop = N/4
        "mdct_n128_h32_t2048_s9": dict(N=128, hop=32, shape=(2048,), seed=9),
    }
    for name, c in cases.items():
        np.random.seed(c["seed"])
        x = np.random.randn(*c["shape"]).astype(np.float32)
        X = ref.mdct_baseline(x, c["N"], c["hop"])
        xr = ref.imdct_baseline(X, c["N"], c["hop"])
        np.savez_compressed(OUT / f"{name}.npz", x=x, X_ref=X, xr_ref=xr,
                            N=np.int64(c["N"]), hop=np.int64(c["hop"]))
        print(name, x.shape, X.shape, xr.shape)

    # full-size clip of BASELINE config #4 (T=196608): store checksums only
    np.random.seed(42)
    x = np.random.randn(196608).astype(np.float32)
    X = ref.mdct_baseline(x, 512, 256)
    xr = ref.imdct_baseline(X, 512, 256)
    Xd = X.astype(np.float64)
    np.savez_compressed(
        OUT / "mdct_n512_h256_t196608_s42_checksums.npz",
        N=np.int64(512), hop=np.int64(256), T=np.int64(196608), seed=np.int64(42),
        X_shape=np.array(X.shape), xr_shape=np.array(xr.shape),
        X_row_sum=Xd.sum(axis=-1).ravel(), X_row_sumsq=(Xd ** 2).sum(axis=-1).ravel(),
        X_col_sum=Xd.sum(axis=-2).ravel(),
        xr_block_sum=xr.astype(np.float64).reshape(-1, 256).sum(axis=-1),
    )
    print("checksums", X.shape, xr.shape)


if __name__ == "__main__":
    main()
