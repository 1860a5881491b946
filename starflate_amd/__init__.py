"""starflate_amd -- MI355X-native DEFLATE compressor (hand-written HIP, gfx950).

Only what the hot path needs: csrc/ (HIP kernels + C-ABI), the ctypes binding,
torch plumbing for device buffers, the multi-GPU shard/concat helper and the
synthetic corpora used by tests and bench.py.
"""
from .compressor import (CHUNK_BYTES, Compressor, StarflateError, checksum_combine, compress, compress_multi,  # noqa: F401
                         wrapper_bytes)

__all__ = ["Compressor", "StarflateError", "compress", "CHUNK_BYTES", "checksum_combine", "wrapper_bytes", "compress_multi"]
