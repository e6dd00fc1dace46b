"""tokenizers_amd -- MI355X-native ``encode_batch`` for huggingface/tokenizers.

One hot path, nothing else: ByteLevel / Whitespace / Bert pre-tokenization ->
BPE / WordPiece / WordLevel -> token-id CSR arrays, as hand-written HIP kernels
for gfx950 behind a C ABI (``include/tokenizers_amd.h``).  See DESIGN.md.
"""
from ._lib import DeviceError, TokenizersAmdError, UnsupportedError  # noqa: F401
from .tokenizer import BatchEncoding, DeviceBatch, Encoding, Tokenizer, pack_documents, pinned_copy, pinned_empty, read_lines  # noqa: F401

__version__ = "0.1.0"
