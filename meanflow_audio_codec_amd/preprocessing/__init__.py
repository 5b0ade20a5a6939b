from .mdct import MDCTConfig, imdct, mdct  # noqa: F401
from .tokenization import MDCTTokenization, ReshapeTokenization, TokenizationStrategy  # noqa: F401
from .tokenization_utils import (compute_token_shape, compute_tokenized_dimension,  # noqa: F401
                                 create_tokenization_strategy)
