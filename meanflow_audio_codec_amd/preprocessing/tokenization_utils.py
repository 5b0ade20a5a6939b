"""``create_tokenization_strategy`` / ``compute_token*`` -- mirror of ``preprocessing/tokenization_utils.py``.

The reference tokenises a dummy zeros batch to learn the shapes (:63-135); here the shapes are
computed in closed form (same numbers, no launch, works without a GPU).
"""
from __future__ import annotations

from .mdct import MDCTConfig
from .tokenization import MDCTTokenization, ReshapeTokenization, TokenizationStrategy


def create_tokenization_strategy(config) -> TokenizationStrategy | None:
    name = config.tokenization_strategy
    if name is None:
        return None
    tc = config.tokenization_config or {}
    if name == "mdct":
        return MDCTTokenization(config=MDCTConfig(window_size=tc.get("window_size", 512), hop_size=tc.get("hop_size")))
    if name == "reshape":
        patch_size, image_size = tc.get("patch_size"), tc.get("image_size")
        if isinstance(image_size, list):
            image_size = tuple(image_size)
        if isinstance(patch_size, list):
            patch_size = tuple(patch_size)
        return ReshapeTokenization(patch_size=patch_size, patch_length=tc.get("patch_length"), image_size=image_size)
    raise ValueError(f"Unknown tokenization_strategy: {name}. Must be one of: 'mdct', 'reshape'")


def _check_dataset(dataset):
    if dataset not in ("mnist", "audio"):
        raise ValueError(f"Unknown dataset: {dataset}")


def compute_tokenized_dimension(tokenization: TokenizationStrategy, original_dimension: int, dataset: str) -> int:
    _check_dataset(dataset)
    n, d = tokenization.token_shape(original_dimension)
    return int(n * d)


def compute_token_shape(tokenization: TokenizationStrategy, original_dimension: int, dataset: str) -> tuple[int, int]:
    _check_dataset(dataset)
    n, d = tokenization.token_shape(original_dimension)
    return int(n), int(d)
