"""Host plumbing over the C-ABI: device memory and streams come from torch, the
compression itself happens only in libstarflate_hip.so (HIP kernels, gfx950).

`compress()` is the sibling of the reference's
`starflate::decompress(src, dst) -> status` (/root/reference/src/decompress.hpp:63-71):
raw RFC 1951 out (or, with container="zlib"/"gzip", the RFC 1950 / RFC 1952 wrapper the
reference's fixture tool strips, tools/deflate_compress.py:8-13), caller-owned buffers,
integer status turned into an exception.
"""
import ctypes as C

import numpy as np

from . import _capi

CHUNK_BYTES = 32768


class StarflateError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"starflate_hip error {code}: {msg}")
        self.code = code


class Compressor:
    """One sfh_ctx (one GPU). Not thread-safe; use one per thread / per rank."""

    def __init__(self, device=0):
        self._lib = _capi.lib()
        n = self._lib.sfh_device_count()
        if n <= 0:
            raise StarflateError(-3, "no HIP device visible; the compressor has no CPU fallback")
        h = C.c_void_p()
        rc = self._lib.sfh_create(C.byref(h), int(device))
        if rc:
            raise StarflateError(rc, f"sfh_create(device={device}) failed")
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sfh_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc:
            raise StarflateError(rc, self._lib.sfh_last_error(self._h).decode())

    @staticmethod
    def compress_bound(n):
        return _capi.lib().sfh_compress_bound(int(n), 0)

    # ---- host buffers (PCIe inclusive) ----
    def compress(self, data, strategy="auto", final_stream=True, lazy=True, stored_fast_path=True, container="raw",
                 block_bytes=0, effort="default"):
        src = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        cap = self.compress_bound(src.size)
        dst = np.empty(cap, dtype=np.uint8)
        out_n = C.c_size_t(0)
        opt = _capi.make_options(strategy, final_stream, lazy, stored_fast_path, container, block_bytes, effort)
        self._check(self._lib.sfh_compress(self._h, src.ctypes.data if src.size else None, src.size,
                                           dst.ctypes.data, cap, C.byref(out_n), C.byref(opt)))
        return dst[: out_n.value].tobytes()

    # ---- device buffers (torch uint8 CUDA tensors) ----
    def compress_tensor(self, src, out=None, strategy="auto", final_stream=True, lazy=True, stream=None,
                        stored_fast_path=True, container="raw", block_bytes=0, effort="default"):
        """src: 1-D uint8 tensor on this device. Returns (out tensor, stream byte count)."""
        import torch

        self._check_tensor(src)
        n = src.numel()
        cap = self.compress_bound(n)
        if out is None:
            out = torch.empty(cap, dtype=torch.uint8, device=src.device)
        self._check_tensor(out)
        out_n = C.c_size_t(0)
        opt = _capi.make_options(strategy, final_stream, lazy, stored_fast_path, container, block_bytes, effort)
        s = torch.cuda.current_stream(src.device).cuda_stream if stream is None else stream
        self._check(self._lib.sfh_compress_device(self._h, src.data_ptr() if n else None, n, out.data_ptr(),
                                                  out.numel(), C.byref(out_n), C.byref(opt), C.c_void_p(s)))
        return out, out_n.value

    def compress_tensor_async(self, src, out, size_out, strategy="auto", final_stream=True, lazy=True, stream=None,
                              container="raw", block_bytes=0, effort="default"):
        """Enqueue only. size_out: 1-element int64 CUDA tensor receiving the stream size."""
        import torch

        self._check_tensor(src)
        self._check_tensor(out)
        if size_out.dtype not in (torch.int64, torch.uint64) or not size_out.is_cuda:
            raise ValueError("size_out must be a 1-element int64 CUDA tensor")
        opt = _capi.make_options(strategy, final_stream, lazy, container=container, block_bytes=block_bytes, effort=effort)
        s = torch.cuda.current_stream(src.device).cuda_stream if stream is None else stream
        n = src.numel()
        self._check(self._lib.sfh_compress_device_async(self._h, src.data_ptr() if n else None, n, out.data_ptr(),
                                                        out.numel(), size_out.data_ptr(), C.byref(opt), C.c_void_p(s)))

    # ---- block index + GPU decompress (the reference's decompress(), /root/reference/src/decompress.hpp:63-71,
    #      for streams whose independently decodable 32 KiB segments are known) ----
    def last_block_bytes(self):
        """Strip size the last compress call used (block_bytes after defaulting)."""
        return int(self._lib.sfh_last_block_bytes(self._h))

    def last_error(self):
        """sfh_last_error: what the last failing call (or the last decode with a non-zero status) said."""
        e = self._lib.sfh_last_error(self._h)
        return e.decode() if e else ""

    def last_decode_scratch_bytes(self):
        """Bytes of token scratch the last decompress call used: one batch of whole strips (at most 4 GiB), whatever its size."""
        return int(self._lib.sfh_last_decode_scratch_bytes(self._h))

    def last_index(self, device=None):
        """Index of the last compress call: segments + 1 stream offsets.  numpy uint64 array, or (device given)
        an int64 tensor on that CUDA device."""
        n = self._lib.sfh_index_entries(self._h)
        if n == 0:
            raise StarflateError(-1, "no compress call on this context yet")
        if device is None:
            idx = np.empty(n, dtype=np.uint64)
            self._check(self._lib.sfh_copy_index(self._h, idx.ctypes.data, n, 0, None))
            return idx
        import torch

        idx = torch.empty(n, dtype=torch.int64, device=device)
        s = torch.cuda.current_stream(idx.device).cuda_stream
        self._check(self._lib.sfh_copy_index(self._h, idx.data_ptr(), n, 1, C.c_void_p(s)))
        return idx

    def last_subindex(self, device=None):
        """Sub-index of the last compress call (32 x {bit offset, tokens before} per segment): numpy uint32
        array [segments, 32, 2], or (device given) an int32 tensor of that shape on the CUDA device."""
        nseg = self._lib.sfh_index_entries(self._h) - 1
        if nseg < 0:
            raise StarflateError(-1, "no compress call on this context yet")
        words = nseg * _capi.SUBINDEX_WORDS
        if device is None:
            sub = np.empty((nseg, 32, 2), dtype=np.uint32)
            self._check(self._lib.sfh_copy_subindex(self._h, sub.ctypes.data, words, 0, None))
            return sub
        import torch

        sub = torch.empty((nseg, 32, 2), dtype=torch.int32, device=device)
        s = torch.cuda.current_stream(sub.device).cuda_stream
        self._check(self._lib.sfh_copy_subindex(self._h, sub.data_ptr(), words, 1, C.c_void_p(s)))
        return sub

    def decompress_tensor(self, stream, index, out_n, out=None, hip_stream=None, subindex=None, *, block_bytes):
        """stream: 1-D uint8 CUDA tensor (exactly the compressed bytes); index: int64 CUDA tensor of segments + 1
        offsets; subindex: optional int32 CUDA tensor [segments, 32, 2] (last_subindex); out_n: decompressed size;
        block_bytes (required): the strip size the stream was written with (last_block_bytes() of the compressing call --
        compress*() defaults to strips of up to 256 KiB, so there is no safe default here; 32768 = independent blocks).
        Returns (out tensor, DecompressStatus int, 0 = Success)."""
        import torch

        self._check_tensor(stream)
        if not (isinstance(index, torch.Tensor) and index.is_cuda and index.dtype == torch.int64 and index.is_contiguous()):
            raise ValueError("index must be a contiguous int64 CUDA tensor")
        nseg = index.numel() - 1
        if out is None:
            out = torch.empty(max(int(out_n), 1), dtype=torch.uint8, device=stream.device)
        self._check_tensor(out)
        if out.numel() < out_n:
            raise ValueError("out is smaller than out_n")
        st = C.c_uint32(0)
        s = torch.cuda.current_stream(stream.device).cuda_stream if hip_stream is None else hip_stream
        if subindex is not None and not (subindex.is_cuda and subindex.dtype == torch.int32 and subindex.is_contiguous()
                                         and subindex.numel() == nseg * _capi.SUBINDEX_WORDS):
            raise ValueError("subindex must be a contiguous int32 CUDA tensor of segments * 64 words")
        self._check(self._lib.sfh_decompress_device(self._h, stream.data_ptr(), stream.numel(), index.data_ptr(),
                                                    subindex.data_ptr() if subindex is not None else None, nseg,
                                                    out.data_ptr() if out_n else None, int(out_n), int(block_bytes),
                                                    C.byref(st), C.c_void_p(s)))
        return out[:out_n], st.value

    def decompress(self, data, index, out_n, subindex=None, *, block_bytes):
        """Host buffers: bytes-like stream + numpy uint64 index [+ numpy uint32 sub-index] -> (bytes, DecompressStatus int)."""
        src = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        idx = np.ascontiguousarray(index, dtype=np.uint64)
        dst = np.empty(max(int(out_n), 1), dtype=np.uint8)
        st = C.c_uint32(0)
        sub = None if subindex is None else np.ascontiguousarray(subindex, dtype=np.uint32)
        if sub is not None and sub.size != (idx.size - 1) * _capi.SUBINDEX_WORDS:
            raise ValueError("subindex must hold segments * 64 words")
        self._check(self._lib.sfh_decompress(self._h, src.ctypes.data, src.size, idx.ctypes.data,
                                             sub.ctypes.data if sub is not None else None, idx.size - 1,
                                             dst.ctypes.data if out_n else None, int(out_n), int(block_bytes), C.byref(st)))
        return (dst[:out_n].tobytes() if st.value == 0 else b""), st.value

    def inflate_ms(self):
        ms = (C.c_float * _capi.INFLATE_NSTAGES)()
        self._check(self._lib.sfh_last_inflate_ms(self._h, C.byref(ms)))
        return {self._lib.sfh_inflate_stage_name(k).decode(): float(ms[k]) for k in range(_capi.INFLATE_NSTAGES)}

    def checksum_tensor(self, src, kind, stream=None):
        """kind "zlib" -> Adler-32, "gzip" -> CRC-32 of a 1-D uint8 tensor on this device (GPU kernels)."""
        import torch

        self._check_tensor(src)
        out = C.c_uint32(0)
        s = torch.cuda.current_stream(src.device).cuda_stream if stream is None else stream
        n = src.numel()
        self._check(self._lib.sfh_checksum_device(self._h, src.data_ptr() if n else None, n, _capi.CONTAINER[kind],
                                                  C.byref(out), C.c_void_p(s)))
        return out.value

    def _check_tensor(self, t):
        import torch

        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.dim() == 1 and t.is_contiguous()):
            raise ValueError("expected a contiguous 1-D uint8 CUDA tensor")
        if t.device.index != self.device:
            raise ValueError(f"tensor is on cuda:{t.device.index}, compressor on cuda:{self.device}")

    # ---- measurement / inspection ----
    def lds_order_check(self, op, blocks=256, iters=50):
        """sfh_lds_order_check: does the LDS execute a returning atomic's lanes in ascending order on this device (op 0:
        ds_wrxchg_rtn_b32, the chain efforts; op 1: ds_mskor_rtn_b32, effort "recent")? -> (mismatches, positions checked)"""
        bad, n = C.c_uint64(0), C.c_uint64(0)
        rc = self._lib.sfh_lds_order_check(self._h, int(op), int(blocks), int(iters), C.byref(bad), C.byref(n))
        if rc:
            raise StarflateError(rc, self._lib.sfh_last_error(self._h).decode())
        return bad.value, n.value

    def set_profiling(self, on=True):
        self._lib.sfh_set_profiling(self._h, int(bool(on)))

    def stage_ms(self):
        ms = (C.c_float * _capi.NSTAGES)()
        self._check(self._lib.sfh_last_stage_ms(self._h, C.byref(ms)))
        return {self._lib.sfh_stage_name(k).decode(): float(ms[k]) for k in range(_capi.NSTAGES)}

    def debug(self, what, nchunks):
        shapes = {
            _capi.DBG_NTOK: ((nchunks,), np.uint32),
            _capi.DBG_TOKENS: ((nchunks, CHUNK_BYTES), np.uint32),
            _capi.DBG_HIST: ((nchunks, 576), np.uint32),  # ll at 0, d at 288, raw len-3 counts at 320
            _capi.DBG_PLAN: ((nchunks, 4), np.uint32),
            _capi.DBG_LENS: ((nchunks, 320), np.uint8),
            _capi.DBG_OFFSETS: ((nchunks,), np.uint64),
            _capi.DBG_STAMPS: ((2, nchunks, 8), np.uint64),  # [0] k_lz77 phases, [1] k_plan phases
            _capi.DBG_SUBINDEX: ((nchunks, 32, 2), np.uint32),
            _capi.DBG_ITEMS: ((nchunks, CHUNK_BYTES), np.uint16),
            _capi.DBG_NITEMS: ((nchunks,), np.uint32),
            _capi.DBG_SEGINFO: ((nchunks, 6), np.uint32),  # decoder: status, tokens, raw | serial << 1, bytes, raw offset (u64)
        }
        shape, dt = shapes[what]
        a = np.empty(shape, dtype=dt)
        self._check(self._lib.sfh_debug_read(self._h, what, a.ctypes.data, a.nbytes))
        return a

    def debug_tokens(self, nchunks):
        """The compressor's 16-bit items of the last call as per-chunk uint32 token arrays in the oracle's format
        (bit 31 match, 16..23 len-3, 0..14 dist-1; literal = byte) -> (list of token arrays, list of (token
        index, region) pairs naming the flagged first tokens of parse regions)."""
        items = self.debug(_capi.DBG_ITEMS, nchunks)
        nit = self.debug(_capi.DBG_NITEMS, nchunks)
        toks, flags = [], []
        for c in range(nchunks):
            if nit[c] & 0x80000000:  # kItemsSkipped: stored fast path, the items behind the first 8 KiB were never written
                raise StarflateError(-1, f"chunk {c} took the stored fast path: its items are not materialised (stored_fast_path=False shows them)")
            it = items[c, : nit[c]].astype(np.uint32)
            start = (it & 0x8000) != 0  # kItemTok: a literal or a match head; clear: the distance behind a head
            head = start & ((it & 0x0100) != 0)  # kItemHead
            assert not np.any(head[:-1] & start[1:]) and (it.size == 0 or not head[-1]), "a head without its distance"
            nxt = np.zeros(it.size, dtype=np.uint32)
            nxt[:-1] = it[1:]
            tok = np.where(head, np.uint32(0x80000000) | ((it & 0xFF) << 16) | (nxt & 0x7FFF), it & 0xFF)[start]
            fl = ((it & 0x4000) != 0)[start]
            reg = ((it >> 9) & 31)[start]  # kItemRegionShift
            toks.append(tok.astype(np.uint32))
            flags.append([(int(k), int(reg[k])) for k in np.flatnonzero(fl)])
        return toks, flags


def compress_multi(compressors, data, strategy="auto", final_stream=True, lazy=True, stored_fast_path=True, container="raw",
                   block_bytes=0):
    """One process, several contexts (normally one per GPU): contiguous shards compressed concurrently, one stream
    out -- bit-identical to a single Compressor.compress call (sfh_compress_multi)."""
    L = _capi.lib()
    src = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    cap = L.sfh_compress_bound(src.size, 0)
    dst = np.empty(cap, dtype=np.uint8)
    out_n = C.c_size_t(0)
    opt = _capi.make_options(strategy, final_stream, lazy, stored_fast_path, container, block_bytes)
    handles = (C.c_void_p * len(compressors))(*[c._h for c in compressors])
    rc = L.sfh_compress_multi(handles, len(compressors), src.ctypes.data if src.size else None, src.size, dst.ctypes.data, cap,
                              C.byref(out_n), C.byref(opt))
    if rc:
        raise StarflateError(rc, "; ".join(L.sfh_last_error(c._h).decode() for c in compressors))
    return dst[: out_n.value].tobytes()


def checksum_combine(kind, a, b, len_b):
    """Checksum of A||B from checksum(A), checksum(B), len(B); kind "zlib" (Adler-32) or "gzip" (CRC-32)."""
    L = _capi.lib()
    return (L.sfh_adler32_combine if kind == "zlib" else L.sfh_crc32_combine)(a, b, int(len_b))


def wrapper_bytes(kind, checksum, n):
    """(header, trailer) of the zlib / gzip wrapper exactly as the kernels write them (for a multi-GPU job,
    whose rank 0 wraps the concatenated raw shard streams with the combined checksum)."""
    if kind == "zlib":
        return b"\x78\x9c", int(checksum).to_bytes(4, "big")
    if kind == "gzip":
        return b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff", int(checksum).to_bytes(4, "little") + (n & 0xFFFFFFFF).to_bytes(4, "little")
    return b"", b""


_DEFAULT = {}


def compress(data, device=0, **kw):
    """bytes-like -> raw DEFLATE bytes, through the GPU (host buffers, PCIe inclusive)."""
    c = _DEFAULT.get(device)
    if c is None:
        c = _DEFAULT[device] = Compressor(device)
    return c.compress(data, **kw)
