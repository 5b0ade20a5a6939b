"""MNIST loader -- host mirror of ``datasets/mnist.py:44-78``.

Same iterator contract: ``load_mnist(data_dir, split, batch_size, format, normalize, seed)`` yields
``(images, labels)``; ``split="train"`` is an endless stream of batches drawn with replacement by
``numpy.random.default_rng(seed).integers(0, n, size=batch_size)``, ``split="test"`` walks the set once in order;
images are ``float32 / 255`` mapped to ``[-1, 1]`` by ``(x - 0.5) / 0.5`` and flattened to ``[B, 784]`` for
``format="1d"`` (:12-35).

The reference reads the set through ``tensorflow_datasets`` (not installed here, and it downloads).  This build reads
the four original IDX files (``train-images-idx3-ubyte`` ..., optionally ``.gz``) from ``data_dir``; the sample order
therefore follows the IDX files, not the TFDS shards.
"""
from __future__ import annotations

import gzip
import struct
from pathlib import Path
from typing import Iterator

import numpy as np

_FILES = {"train": ("train-images-idx3-ubyte", "train-labels-idx1-ubyte"),
          "test": ("t10k-images-idx3-ubyte", "t10k-labels-idx1-ubyte")}


def _read_idx(path: Path) -> np.ndarray:
    opener = gzip.open if path.suffix == ".gz" else open
    with opener(path, "rb") as f:
        zero, dtype_code, ndim = struct.unpack(">HBB", f.read(4))
        if zero != 0 or dtype_code != 0x08:
            raise ValueError(f"{path} is not an unsigned-byte IDX file")
        shape = struct.unpack(">" + "I" * ndim, f.read(4 * ndim))
        data = np.frombuffer(f.read(), dtype=np.uint8)
    if data.size != int(np.prod(shape)):
        raise ValueError(f"{path}: expected {int(np.prod(shape))} bytes of data, found {data.size}")
    return data.reshape(shape)


def _find(data_dir: Path, stem: str) -> Path:
    for cand in (data_dir / stem, data_dir / (stem + ".gz"), data_dir / stem.replace("-idx", ".idx")):
        if cand.exists():
            return cand
    raise FileNotFoundError(f"{stem}[.gz] not found in {data_dir}")


def _preprocess_mnist_images(images: np.ndarray, format: str = "1d", normalize: bool = True) -> np.ndarray:
    if format not in ("1d", "2d"):
        raise ValueError(f"Invalid format: {format}. Must be '1d' or '2d'")
    x = images.astype(np.float32) / 255.0
    if normalize:
        x = (x - 0.5) / 0.5
    return x.reshape(x.shape[0], -1) if format == "1d" else x


def load_mnist(data_dir: str = str(Path.home() / "datasets" / "mnist"), split: str = "train", batch_size: int = 512,
               format: str = "1d", normalize: bool = True, seed: int = 42) -> Iterator[tuple[np.ndarray, np.ndarray]]:
    if split not in ("train", "test"):
        raise ValueError(f"Invalid split: {split}. Must be 'train' or 'test'")
    img_stem, lab_stem = _FILES[split]
    root = Path(data_dir)
    images = _preprocess_mnist_images(_read_idx(_find(root, img_stem)), format=format, normalize=normalize)
    labels = _read_idx(_find(root, lab_stem)).astype(np.int64)
    n = len(images)
    if split == "train":
        rng = np.random.default_rng(seed)
        while True:
            pick = rng.integers(0, n, size=batch_size)
            yield images[pick], labels[pick]
    else:
        for lo in range(0, n, batch_size):
            yield images[lo:lo + batch_size], labels[lo:lo + batch_size]
