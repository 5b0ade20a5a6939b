"""Audio data front end -- host mirror of ``datasets/audio.py:35-296``.

Same composable stages and names as the reference -- ``glob_audio_files`` -> ``load_audio_files`` ->
``audio_to_frames`` -> ``buffer_shuffle`` -> ``batch``, assembled by ``build_audio_pipeline`` -- and the same
iterator contract: batches ``[batch_size, frame_sz, n_channels]`` of float32, every file left-padded by a random
offset drawn once per file from ``numpy.random.default_rng(seed).integers(0, frame_sz + 1)`` and right-padded to a
frame multiple (:264-277), frames shuffled through a swap-and-pop buffer (:165-183, 213-223), mono files duplicated
to two channels (:254-257).  Each stage can be called with its stream, or with keyword arguments only to get the
stage as a function (the reference does this with ``toolz.curry``).

MI355X-first differences (all opt-in; with the defaults the stages are pure host code yielding numpy arrays):

* ``device="cuda"``: a decoded file is uploaded once; padding, framing, the shuffle buffer and batching then work on
  HBM-resident tensors (1000 frames of 8.192 s stereo are 1.5 GB of the 288 GB), and batches come out as device
  tensors ready for the tokeniser.
* ``target_sr``: the reference drops every file that is not 44.1 kHz and trains on the file rate (:244-251); the
  audio configuration of BASELINE is 24 kHz, so with ``target_sr`` set a file of any rate is converted on the device by
  the polyphase kernel ``mfc_resample_poly`` (``resample``; 44.1 -> 24 kHz is 80/147).  There is no host fallback:
  ``target_sr`` without a device raises.
* ``.wav`` (PCM 8/16/24/32-bit, via the standard library) is decoded besides ``.mp3`` when asked for by
  ``extensions``; MP3 needs ``minimp3py`` exactly as in the reference (:226-234) and raises the same ``ImportError``.
* the prefetch thread hands files over through a bounded blocking queue.  (The reference's ``deque(maxlen=2*prefetch)``
  silently discards the oldest decoded file whenever the consumer is slower than the decoder, :120, 289-294.)
* ``glob_audio_files`` sorts the directory listing before the seeded shuffle, so the order does not depend on the
  file system.
"""
from __future__ import annotations

import functools
import logging
import math
import queue
import random
import threading
import wave
from pathlib import Path
from typing import Callable, Iterable, Iterator

import numpy as np

logger = logging.getLogger(__name__)

try:
    import minimp3py
    MINIMP3PY_AVAILABLE = True
except ImportError:
    minimp3py = None
    MINIMP3PY_AVAILABLE = False

NATIVE_SR = 44100


def _stage(fn: Callable) -> Callable:
    """``fn(stream, **kw)`` -> result; ``fn(**kw)`` -> the stage with those keywords bound."""
    @functools.wraps(fn)
    def wrapper(*args, **kw):
        if not args:
            return functools.partial(fn, **kw)
        return fn(*args, **kw)
    return wrapper


# ---- resampling on the device ---------------------------------------------------------------------------------

_FILTERS: dict = {}


def design_lowpass(up: int, down: int, beta: float = 5.0) -> np.ndarray:
    """Kaiser-windowed sinc at rate ``up * fs_in``: cut-off ``1/max(up, down)`` of Nyquist, ``20 max(up, down) + 1``
    taps, unit DC gain, times ``up`` -- the design ``scipy.signal.resample_poly`` publishes (float64)."""
    rate = max(up, down)
    half = 10 * rate
    m = np.arange(-half, half + 1, dtype=np.float64)
    h = np.sinc(m / rate) / rate * np.kaiser(2 * half + 1, beta)
    return h * (up / h.sum())


def resample(x, sr_in: int, sr_out: int):
    """``[..., T]`` float32 device tensor at ``sr_in`` -> ``[..., ceil(T * sr_out / sr_in)]`` at ``sr_out`` through
    ``mfc_resample_poly``."""
    import torch

    from .. import _lib
    if not isinstance(x, torch.Tensor):
        raise TypeError("resample() takes a device tensor (the resampler is a HIP kernel; there is no host path)")
    _lib.require_cuda(x)
    g = math.gcd(int(sr_in), int(sr_out))
    up, down = int(sr_out) // g, int(sr_in) // g
    if up == down:
        return x
    key = (up, down, x.device)
    h = _FILTERS.get(key)
    if h is None:
        h = _FILTERS[key] = torch.from_numpy(design_lowpass(up, down).astype(np.float32)).to(x.device)
    lead, T = x.shape[:-1], x.shape[-1]
    x2 = x.reshape(-1, T).to(torch.float32).contiguous()
    rows = x2.shape[0]
    L = _lib.lib().mfc_resample_out_len(T, up, down)
    y = torch.empty((rows, L), dtype=torch.float32, device=x.device)
    if rows > 0 and T > 0:
        rc = _lib.lib().mfc_resample_poly(x2.data_ptr(), rows, T, T, up, down, h.data_ptr(), h.numel(), y.data_ptr(),
                                          L, _lib.stream_ptr())
        _lib.check(rc, "mfc_resample_poly")
    return y.reshape(*lead, L)


# ---- public entry points ----------------------------------------------------------------------------------------

def build_audio_pipeline(data_dir: str, seed: int, frame_sz: int = 256 * 256 * 3, prefetch: int = 4,
                         buffer_size: int = 1000, batch_size: int = 32, drop_last: bool = False, *,
                         device: str | None = None, target_sr: int | None = None,
                         extensions: tuple[str, ...] = (".mp3",)) -> Iterator:
    """``datasets/audio.py:35-67``: batches ``[batch_size, frame_sz, n_channels]``."""
    if target_sr is not None and device is None:
        raise ValueError("target_sr needs device=...: resampling runs in the HIP kernel, there is no host path")
    files = glob_audio_files(data_dir, seed=seed, extensions=extensions)
    audio = load_audio_files(iter(files), prefetch=prefetch, device=device, target_sr=target_sr)
    frames = audio_to_frames(audio, frame_sz=frame_sz, seed=seed)
    return batch(buffer_shuffle(frames, buffer_size=buffer_size, seed=seed), batch_size=batch_size,
                 drop_last=drop_last)


def load_audio(files: list[Path], seed: int, frame_sz: int = 256 * 256 * 3, prefetch: int = 4, **kw) -> Iterator:
    """``datasets/audio.py:70-78``: frames of the given files."""
    return audio_to_frames(load_audio_files(iter(files), prefetch=prefetch, **kw), frame_sz=frame_sz, seed=seed)


def glob_audio_files(data_dir: str, seed: int, extensions: tuple[str, ...] = (".mp3",)) -> list[Path]:
    """Files of ``data_dir`` (not recursive) with one of ``extensions``, shuffled by ``random.Random(seed)``
    (:85-92)."""
    found = sorted(p for p in Path(data_dir).iterdir() if p.is_file() and p.suffix.lower() in extensions)
    random.Random(seed).shuffle(found)
    return found


# ---- stages -----------------------------------------------------------------------------------------------------

_DONE = object()


def _put(out: "queue.Queue", item, stop: threading.Event) -> None:
    while not stop.is_set():
        try:
            out.put(item, timeout=0.05)
            return
        except queue.Full:
            pass


def _decode_into(files: Iterable[Path], out: "queue.Queue", stop: threading.Event, decode: Callable) -> None:
    try:
        for f in files:
            if stop.is_set():
                return
            try:
                a = decode(f)
            except ImportError:
                raise
            except Exception as e:          # undecodable file: skipped, as the reference does (:107-109, 293-294)
                logger.warning("skipping %s: %s", f, e)
                continue
            if a is not None:
                _put(out, a, stop)
    except BaseException as e:              # surfaced in the consumer
        _put(out, e, stop)
    finally:
        _put(out, _DONE, stop)


@_stage
def load_audio_files(files: Iterator[Path], prefetch: int = 4, *, device: str | None = None,
                     target_sr: int | None = None) -> Iterator:
    """Decode files to ``[n_channels, n_samples]`` float32 (:95-132), ``prefetch`` files ahead on a worker thread;
    with ``device`` the array is uploaded (and resampled when ``target_sr`` is set) before it is yielded."""
    def finish(pair):
        a, sr = pair
        if device is None:
            return a
        import torch
        t = torch.from_numpy(a).to(device)
        return resample(t, sr, target_sr) if target_sr is not None else t

    decode = functools.partial(_load_audio_with_rate, expected_sr=None if target_sr is not None else NATIVE_SR)
    if prefetch <= 0:
        for f in files:
            try:
                pair = decode(f)
            except ImportError:
                raise
            except Exception as e:
                logger.warning("skipping %s: %s", f, e)
                continue
            if pair is not None:
                yield finish(pair)
        return
    q: "queue.Queue" = queue.Queue(maxsize=max(1, prefetch * 2))
    stop = threading.Event()
    worker = threading.Thread(target=_decode_into, args=(files, q, stop, decode), daemon=True)
    worker.start()
    try:
        while True:
            item = q.get()
            if item is _DONE:
                return
            if isinstance(item, BaseException):
                raise item
            yield finish(item)
    finally:
        stop.set()
        worker.join(timeout=1.0)


def _zeros_like_cols(a, n: int):
    if isinstance(a, np.ndarray):
        return np.zeros((a.shape[0], n), dtype=a.dtype)
    return a.new_zeros((a.shape[0], n))


def _cat_cols(parts):
    parts = [p for p in parts if p.shape[1] > 0]
    if len(parts) == 1:
        return parts[0]
    if isinstance(parts[0], np.ndarray):
        return np.concatenate(parts, axis=1)
    import torch
    return torch.cat(parts, dim=1)


def _prepend_and_pad_audio(audio, frame_sz: int, rng: np.random.Generator):
    """Random left pad in ``[0, frame_sz]`` then right pad to a multiple of ``frame_sz`` (:264-277)."""
    pre = int(rng.integers(0, frame_sz + 1))
    post = (-(audio.shape[-1] + pre)) % frame_sz
    if pre == 0 and post == 0:
        return audio
    return _cat_cols([_zeros_like_cols(audio, pre), audio, _zeros_like_cols(audio, post)])


@_stage
def audio_to_frames(audio_files: Iterator, frame_sz: int, seed: int) -> Iterator:
    """``[C, n]`` files -> ``[frame_sz, C]`` frames (:135-162); one generator draw per file."""
    rng = np.random.default_rng(seed)
    for audio in audio_files:
        padded = _prepend_and_pad_audio(audio, frame_sz, rng)
        C, n = padded.shape
        frames = padded.T.reshape(n // frame_sz, frame_sz, C)
        for k in range(frames.shape[0]):
            yield frames[k]


def _swap_and_pop(buf: list, rng: np.random.Generator):
    """Remove a random element in O(1) (:213-223).  ``integers(0, len - 1)`` excludes the last slot, as there."""
    if len(buf) == 1:
        return buf.pop()
    i = int(rng.integers(0, len(buf) - 1))
    buf[i], buf[-1] = buf[-1], buf[i]
    return buf.pop()


@_stage
def buffer_shuffle(generator: Iterator, buffer_size: int, seed: int) -> Iterator:
    """Shuffle through a buffer of ``buffer_size`` items (:165-183)."""
    rng = np.random.default_rng(seed)
    buf: list = []
    for item in generator:
        buf.append(item)
        if len(buf) >= buffer_size:
            yield _swap_and_pop(buf, rng)
    while buf:
        yield _swap_and_pop(buf, rng)


def _stack(items):
    if isinstance(items[0], np.ndarray):
        return np.stack(items)
    import torch
    return torch.stack(items)


@_stage
def batch(generator: Iterator, batch_size: int, drop_last: bool = False) -> Iterator:
    """Stack ``batch_size`` items; the short tail is kept unless ``drop_last`` (:186-206)."""
    pending: list = []
    for item in generator:
        pending.append(item)
        if len(pending) == batch_size:
            yield _stack(pending)
            pending = []
    if pending and not drop_last:
        yield _stack(pending)


# ---- decoding ---------------------------------------------------------------------------------------------------

def _read_wav(file: Path) -> tuple[np.ndarray, int]:
    with wave.open(str(file), "rb") as w:
        C, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 1:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        a = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif width == 4:
        a = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    else:
        raise ValueError(f"unsupported WAV sample width {width}")
    return a.reshape(-1, C), sr


def _load_audio_with_rate(file: Path, expected_sr: int | None = NATIVE_SR) -> tuple[np.ndarray, int] | None:
    file = Path(file)
    if file.suffix.lower() == ".wav":
        wav, sr = _read_wav(file)
    else:
        if not MINIMP3PY_AVAILABLE:
            raise ImportError("minimp3py is required for audio loading. Install with: uv sync --extra audio or "
                              "CFLAGS='-O3 -march=native' pip install git+https://github.com/f0k/minimp3py.git")
        wav, sr = minimp3py.read(str(file))
    if expected_sr is not None and sr != expected_sr:
        logger.warning("Dropping audio file %s: sample rate is %dHz, expected %dHz", file.name, sr, expected_sr)
        return None
    wav = np.asarray(wav)
    audio = np.stack([wav, wav], axis=0) if wav.ndim == 1 else wav.T      # mono -> two identical channels (:254-257)
    if audio.shape[0] == 1:                  # a one-channel [n, 1] decode is mono as well
        audio = np.concatenate([audio, audio], axis=0)
    return np.ascontiguousarray(audio, dtype=np.float32), int(sr)


def _load_audio(file: Path) -> np.ndarray | None:
    """``[n_channels, n_samples]`` float32, ``None`` for a file that is not 44.1 kHz (:226-262)."""
    pair = _load_audio_with_rate(file, expected_sr=NATIVE_SR)
    return None if pair is None else pair[0]
