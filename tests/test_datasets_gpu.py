"""GPU: the device side of the data front end -- the polyphase resampler against
``scipy.signal.resample_poly`` (the published definition ``oracle/datasets_oracle.py`` restates), the HBM-resident
pipeline against the host one, and ``train_flow`` fed from a directory of audio files."""
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_wav(path, data, sr=44100):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(data.shape[1])
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes((np.clip(data, -1, 1 - 1 / 32768) * 32768).astype("<i2").tobytes())


@pytest.mark.parametrize("sr_in,sr_out,shape", [
    (44100, 24000, (2, 44100)),        # the front end's conversion, 80/147
    (44100, 24000, (3, 2, 5001)),      # leading dimensions, ragged length
    (24000, 44100, (1, 3000)),         # 147/80
    (48000, 24000, (4, 1001)),         # 1/2
    (16000, 24000, (2, 777)),          # 3/2
    (44100, 24000, (2, 1)),            # a single sample
    (44100, 24000, (300, 700)),        # more rows than one grid pass of tiles
])
def test_resample_matches_scipy(sr_in, sr_out, shape):
    import scipy.signal as ss
    from meanflow_audio_codec_amd.datasets import resample
    x = np.random.default_rng(sum(shape)).standard_normal(shape).astype(np.float32)
    y = resample(torch.from_numpy(x).cuda(), sr_in, sr_out).cpu().numpy()
    ref = ss.resample_poly(x.astype(np.float64), sr_out, sr_in, axis=-1)
    assert y.shape == ref.shape and y.dtype == np.float32
    # fp32 accumulation of <= ~40 taps against float64: tolerance 2e-5 of the signal scale
    assert np.abs(y - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_resample_matches_oracle_loops_and_properties():
    from meanflow_audio_codec_amd.datasets import design_lowpass, resample
    from oracle.datasets_oracle import design_filter, resample_poly_f64
    x = np.random.default_rng(0).standard_normal((2, 1500)).astype(np.float32)
    y = resample(torch.from_numpy(x).cuda(), 44100, 24000).cpu().numpy()
    ref = resample_poly_f64(x, 80, 147, design_filter(80, 147))
    assert np.abs(y - ref).max() <= 2e-5 * np.abs(ref).max()
    assert np.array_equal(design_lowpass(80, 147), design_lowpass(80, 147)) and len(design_lowpass(80, 147)) == 2941
    # size-independent properties at the literal clip length: unit DC gain and a preserved in-band tone
    T = 361267                                   # -> 196608 samples at 24 kHz (ceil(361267*80/147))
    dc = resample(torch.ones(1, T, device="cuda"), 44100, 24000)
    assert dc.shape == (1, 196608)
    assert (dc[0, 100:-100] - 1.0).abs().max().item() < 2e-4
    t = torch.arange(T, device="cuda", dtype=torch.float64) / 44100.0
    tone = torch.sin(2 * np.pi * 1000.0 * t).float()[None]
    out = resample(tone, 44100, 24000)[0]
    t2 = torch.arange(out.numel(), device="cuda", dtype=torch.float64) / 24000.0
    assert (out[200:-200] - torch.sin(2 * np.pi * 1000.0 * t2).float()[200:-200]).abs().max().item() < 2e-3
    # an out-of-band tone (15 kHz > the 12 kHz Nyquist of the output) is removed
    hi = torch.sin(2 * np.pi * 15000.0 * t).float()[None]
    assert resample(hi, 44100, 24000)[0, 200:-200].abs().max().item() < 2e-2
    # same rate: returned as is
    same = torch.ones(2, 5, device="cuda")
    assert resample(same, 24000, 24000) is same


def test_device_pipeline_matches_host_pipeline(tmp_path):
    import scipy.signal as ss
    from meanflow_audio_codec_amd.datasets import audio as A
    from oracle import datasets_oracle as O
    rng = np.random.default_rng(2)
    for k in range(5):
        _write_wav(tmp_path / f"f{k}.wav", rng.uniform(-0.5, 0.5, size=(int(rng.integers(2000, 6000)), 2)))
    kw = dict(seed=9, frame_sz=512, buffer_size=3, batch_size=4, extensions=(".wav",))
    host = list(A.build_audio_pipeline(str(tmp_path), **kw))
    dev = list(A.build_audio_pipeline(str(tmp_path), device="cuda", **kw))
    assert len(host) == len(dev) > 1
    for h, d in zip(host, dev):
        assert d.is_cuda and d.dtype == torch.float32 and np.array_equal(h, d.cpu().numpy())     # no arithmetic: exact

    # with target_sr: decoded file -> 24 kHz on the device -> the same padding / framing / shuffle arithmetic
    res = list(A.build_audio_pipeline(str(tmp_path), device="cuda", target_sr=24000, **kw))
    files = A.glob_audio_files(str(tmp_path), seed=9, extensions=(".wav",))
    dec = [ss.resample_poly(A._load_audio(f).astype(np.float64), 80, 147, axis=-1).astype(np.float32) for f in files]
    frames = O.frames_of(dec, 512, 9)
    order = O.shuffle_order(len(frames), 3, 9)
    ref = [np.stack(g) for g in O.batches_of([frames[i] for i in order], 4, False)]
    assert len(res) == len(ref)
    for g, r in zip(res, ref):
        assert g.shape == r.shape and np.abs(g.cpu().numpy() - r).max() < 2e-5


def test_train_flow_from_an_audio_directory(tmp_path):
    from meanflow_audio_codec_amd.configs import TrainFlowConfig
    from meanflow_audio_codec_amd.trainers.train import dataset_iterator, train_flow
    data = tmp_path / "audio"
    data.mkdir()
    rng = np.random.default_rng(4)
    for k in range(4):
        _write_wav(data / f"clip{k}.wav", 0.3 * rng.standard_normal((9000, 2)))
    cfg = TrainFlowConfig(batch_size=4, n_steps=3, sample_every=100, sample_seed=1, sample_steps=1, base_lr=1e-3,
                          weight_decay=1e-4, seed=0, noise_dimension=1024, condition_dimension=16, latent_dimension=8,
                          num_blocks=1, dataset="audio", architecture="mlp", use_improved_mean_flow=True,
                          loss_strategy="improved_mean_flow", tokenization_strategy="mdct",
                          tokenization_config={"window_size": 64, "hop_size": 32}, data_dir=str(data),
                          workdir=tmp_path / "run")
    b = next(dataset_iterator(cfg, target_sr=24000))
    assert b.is_cuda and b.shape == (4, 1024)                       # stereo frames averaged to the configured dimension
    state, token_shape = train_flow(cfg, target_sr=24000)
    assert state.step == 3 and token_shape == (31, 64)
    assert (tmp_path / "run" / "checkpoints" / "step_00003.msgpack").exists()
    # stereo token layout of MDCTLayer: L/R concatenated on the coefficient axis
    from meanflow_audio_codec_amd.preprocessing.mdct import mdct
    from meanflow_audio_codec_amd.preprocessing.tokenization import MDCTTokenization
    tok = MDCTTokenization(window_size=64, hop_size=32)
    st = torch.randn(2, 1024, 2, device="cuda")
    tk = tok.tokenize(st)
    assert tk.shape == (2, 31, 128)
    assert torch.equal(tk[..., :64], mdct(st[:, :, 0].contiguous(), window_size=64, hop_size=32))
    assert torch.equal(tk[..., 64:], mdct(st[:, :, 1].contiguous(), window_size=64, hop_size=32))
    back = tok.detokenize(tk)
    assert back.shape == (2, 1088, 2)
    assert (back[:, 128:900] - 2.0 * st[:, 128:900]).abs().max().item() < 1e-4      # gain N/hop = 2 on the interior
