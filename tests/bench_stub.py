"""Stand-in compressor for `bench.py --backend gloo` (CPU rehearsal of the N > 1 launch / rounds / gather path).

Test infrastructure, never the product: bench.py loads it only with --backend gloo, and its line then says
"rehearsal".  The streams come from zlib (raw DEFLATE; a non-final piece ends with Z_FULL_FLUSH, i.e. a byte-aligned
empty stored block without BFINAL -- the same framing the GPU compressor gives a non-final piece,
/root/reference/src/decompress.cpp:178 and :416-436)."""
import zlib

import torch


class Compressor:
    name = "tests/bench_stub.py (zlib level 1)"

    def set_profiling(self, on):
        pass

    @staticmethod
    def compress_bound(n):
        return int(n) + (int(n) >> 10) * 5 + 1024

    def compress_tensor_async(self, piece, out, size_dev, final_stream=True, block_bytes=0, effort="default"):
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        s = co.compress(piece.numpy().tobytes()) + co.flush(zlib.Z_FINISH if final_stream else zlib.Z_FULL_FLUSH)
        out[: len(s)] = torch.frombuffer(bytearray(s), dtype=torch.uint8)
        size_dev[0] = len(s)

    def checksum_tensor(self, piece, container):
        f = zlib.crc32 if container == "gzip" else zlib.adler32
        return f(piece.numpy().tobytes())
