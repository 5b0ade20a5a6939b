"""CPU: the data front end's host stages against the straight-line restatement in
``oracle/datasets_oracle.py``, the filter design against scipy, WAV / IDX decoding, and the C-ABI helpers."""
import gzip
import struct
import wave

import numpy as np
import pytest

from meanflow_audio_codec_amd import _lib
from meanflow_audio_codec_amd.datasets import audio as A
from meanflow_audio_codec_amd.datasets import load_mnist
from oracle import datasets_oracle as O


def _files(seed=0, n=5, C=2, lo=50, hi=400):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal((C, int(rng.integers(lo, hi)))).astype(np.float32) for _ in range(n)]


def _write_wav(path, data, sr=44100, width=2):
    """data [n, C] float in [-1, 1)"""
    with wave.open(str(path), "wb") as w:
        w.setnchannels(data.shape[1])
        w.setsampwidth(width)
        w.setframerate(sr)
        if width == 2:
            w.writeframes((np.clip(data, -1, 1 - 1 / 32768) * 32768).astype("<i2").tobytes())
        elif width == 1:
            w.writeframes((np.clip(data, -1, 1 - 1 / 128) * 128 + 128).astype(np.uint8).tobytes())
        elif width == 3:
            v = (np.clip(data, -1, 1 - 2 ** -23) * 8388608).astype(np.int32)
            b = np.stack([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF], axis=-1).astype(np.uint8)
            w.writeframes(b.tobytes())
        else:
            w.writeframes((np.clip(data, -1, 1 - 2 ** -31).astype(np.float64) * 2147483648).astype("<i4").tobytes())


def test_filter_design_matches_scipy():
    import scipy.signal as ss
    for up, down in [(80, 147), (147, 80), (1, 2), (3, 2)]:
        rate = max(up, down)
        ref = ss.firwin(20 * rate + 1, 1.0 / rate, window=("kaiser", 5.0)) * up
        assert np.abs(A.design_lowpass(up, down) - ref).max() < 1e-14
        assert np.abs(O.design_filter(up, down) - ref).max() < 1e-14


def test_oracle_resampler_matches_scipy():
    import scipy.signal as ss
    rng = np.random.default_rng(3)
    for up, down, T in [(80, 147, 2000), (3, 2, 50), (1, 2, 101), (2, 3, 7), (80, 147, 1)]:
        x = rng.standard_normal((2, T))
        got = O.resample_poly_f64(x, up, down, O.design_filter(up, down))
        ref = ss.resample_poly(x, up, down, axis=-1)
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-13


@pytest.mark.parametrize("frame_sz", [64, 100, 1000])
def test_frames_match_oracle(frame_sz):
    files = _files(seed=frame_sz)
    got = list(A.audio_to_frames(iter(files), frame_sz=frame_sz, seed=11))
    ref = O.frames_of(files, frame_sz, 11)
    assert len(got) == len(ref) > 0
    for g, r in zip(got, ref):
        assert g.shape == (frame_sz, 2) and np.array_equal(g, r)
    # every sample of every file survives exactly once, in order, behind its random offset
    flat = np.concatenate([g[:, 0] for g in got])
    assert np.isclose(np.abs(flat).sum(), sum(np.abs(f[0]).sum() for f in files), rtol=1e-6)


@pytest.mark.parametrize("n,buf", [(1, 4), (10, 1), (10, 4), (37, 8), (5, 100)])
def test_buffer_shuffle_matches_oracle(n, buf):
    got = list(A.buffer_shuffle(iter(range(n)), buffer_size=buf, seed=5))
    assert got == O.shuffle_order(n, buf, 5)
    assert sorted(got) == list(range(n))


def test_batch_tail_and_currying():
    items = [np.full((3, 2), k, np.float32) for k in range(7)]
    got = list(A.batch(iter(items), batch_size=3))
    assert [b.shape for b in got] == [(3, 3, 2), (3, 3, 2), (1, 3, 2)]
    assert [b[:, 0, 0].tolist() for b in got] == [[float(v) for v in grp] for grp in
                                                   O.batches_of(list(range(7)), 3, False)]
    assert len(list(A.batch(iter(items), batch_size=3, drop_last=True))) == 2
    assert list(A.batch(iter([]), batch_size=3)) == []
    # keyword-only call returns the stage (the reference composes curried stages)
    stage = A.batch(batch_size=2, drop_last=True)
    assert [b.shape[0] for b in stage(iter(items))] == [2, 2, 2]
    fr = A.audio_to_frames(frame_sz=64, seed=1)
    assert len(list(fr(iter(_files())))) == len(O.frames_of(_files(), 64, 1))


def test_wav_decoding_rates_and_mono(tmp_path):
    rng = np.random.default_rng(0)
    st = rng.uniform(-0.9, 0.9, size=(500, 2))
    for width, tol in [(1, 1 / 128), (2, 1 / 32768), (3, 2 ** -23), (4, 1e-7)]:
        _write_wav(tmp_path / f"s{width}.wav", st, width=width)
        a = A._load_audio(tmp_path / f"s{width}.wav")
        assert a.shape == (2, 500) and a.dtype == np.float32 and np.abs(a.T - st).max() <= tol * 1.01
    _write_wav(tmp_path / "mono.wav", st[:, :1])
    m = A._load_audio(tmp_path / "mono.wav")
    assert m.shape == (2, 500) and np.array_equal(m[0], m[1])                 # mono -> two identical channels
    _write_wav(tmp_path / "r48.wav", st, sr=48000)
    assert A._load_audio(tmp_path / "r48.wav") is None                        # not 44.1 kHz: dropped
    pair = A._load_audio_with_rate(tmp_path / "r48.wav", expected_sr=None)
    assert pair[1] == 48000 and pair[0].shape == (2, 500)
    (tmp_path / "x.mp3").write_bytes(b"\x00" * 16)
    if not A.MINIMP3PY_AVAILABLE:
        with pytest.raises(ImportError, match="minimp3py"):
            A._load_audio(tmp_path / "x.mp3")
        with pytest.raises(ImportError, match="minimp3py"):                   # also through the prefetch thread
            list(A.load_audio_files(iter([tmp_path / "x.mp3"]), prefetch=2))


def test_host_pipeline_end_to_end(tmp_path):
    rng = np.random.default_rng(1)
    raw = {}
    for k in range(6):
        d = rng.uniform(-0.5, 0.5, size=(int(rng.integers(300, 900)), 2))
        _write_wav(tmp_path / f"f{k}.wav", d)
        raw[f"f{k}.wav"] = d
    _write_wav(tmp_path / "other_rate.wav", rng.uniform(-0.5, 0.5, size=(400, 2)), sr=22050)
    (tmp_path / "notes.txt").write_text("not audio")
    (tmp_path / "broken.wav").write_bytes(b"RIFFxxxx")
    assert A.glob_audio_files(str(tmp_path), seed=3) == []                    # reference default: .mp3 only
    files = A.glob_audio_files(str(tmp_path), seed=3, extensions=(".wav",))
    assert sorted(f.name for f in files) == sorted(list(raw) + ["other_rate.wav", "broken.wav"])
    assert files == A.glob_audio_files(str(tmp_path), seed=3, extensions=(".wav",))
    assert files != sorted(files)

    kw = dict(seed=3, frame_sz=128, buffer_size=4, batch_size=5, extensions=(".wav",))
    got = list(A.build_audio_pipeline(str(tmp_path), prefetch=2, **kw))
    same = list(A.build_audio_pipeline(str(tmp_path), prefetch=0, **kw))
    assert len(got) == len(same) and all(np.array_equal(a, b) for a, b in zip(got, same))
    # oracle composition on the decoded files in glob order (wrong-rate and broken files are skipped)
    dec = [A._load_audio(f) for f in files if f.name in raw]
    frames = O.frames_of(dec, 128, 3)
    order = O.shuffle_order(len(frames), 4, 3)
    ref = [np.stack(grp) for grp in O.batches_of([frames[i] for i in order], 5, False)]
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert g.dtype == np.float32 and g.shape[1:] == (128, 2) and np.array_equal(g, r)
    assert all(b.shape[0] == 5 for b in A.build_audio_pipeline(str(tmp_path), drop_last=True, **kw))
    with pytest.raises(ValueError, match="target_sr needs device"):
        A.build_audio_pipeline(str(tmp_path), target_sr=24000, **kw)
    with pytest.raises(TypeError, match="device tensor"):
        A.resample(np.zeros(10, np.float32), 44100, 24000)


def _write_idx(path, arr, gz=False):
    hdr = struct.pack(">HBB", 0, 8, arr.ndim) + struct.pack(">" + "I" * arr.ndim, *arr.shape)
    (gzip.open if gz else open)(path, "wb").write(hdr + arr.astype(np.uint8).tobytes())


def test_mnist_idx_loader(tmp_path):
    rng = np.random.default_rng(0)
    imgs = rng.integers(0, 256, size=(50, 28, 28), dtype=np.uint8)
    labs = rng.integers(0, 10, size=(50,), dtype=np.uint8)
    _write_idx(tmp_path / "train-images-idx3-ubyte.gz", imgs, gz=True)
    _write_idx(tmp_path / "train-labels-idx1-ubyte.gz", labs, gz=True)
    _write_idx(tmp_path / "t10k-images-idx3-ubyte", imgs[:20])
    _write_idx(tmp_path / "t10k-labels-idx1-ubyte", labs[:20])
    it = load_mnist(str(tmp_path), split="train", batch_size=8, seed=42)
    x, y = next(it)
    pick = np.random.default_rng(42).integers(0, 50, size=8)
    assert x.shape == (8, 784) and x.dtype == np.float32 and np.array_equal(y, labs[pick])
    assert np.allclose(x, (imgs[pick].reshape(8, -1).astype(np.float32) / 255.0 - 0.5) / 0.5)
    assert x.min() >= -1 and x.max() <= 1
    x2, _ = next(it)
    assert x2.shape == (8, 784)                                           # endless stream
    test = list(load_mnist(str(tmp_path), split="test", batch_size=8, format="2d", normalize=False))
    assert [b[0].shape[0] for b in test] == [8, 8, 4] and test[0][0].shape[1:] == (28, 28)
    assert np.allclose(test[0][0], imgs[:8] / 255.0)
    with pytest.raises(ValueError, match="Invalid split"):
        next(load_mnist(str(tmp_path), split="val"))
    with pytest.raises(ValueError, match="Invalid format"):
        next(load_mnist(str(tmp_path), format="3d"))
    with pytest.raises(FileNotFoundError):
        next(load_mnist(str(tmp_path / "nowhere")))


def test_resampler_c_abi_without_gpu():
    l = _lib.lib()
    assert l.mfc_resample_out_len(44100, 80, 147) == 24000
    assert l.mfc_resample_out_len(361268, 80, 147) == 196609      # ceil
    assert l.mfc_resample_out_len(1, 80, 147) == 1
    assert l.mfc_resample_out_len(0, 80, 147) == 0
    assert l.mfc_resample_poly(None, 1, 10, 10, 1, 2, None, 41, None, 5, None) == -14
