"""Experiment: token-stage ms of the index-only decoder on 256 MiB of the bench text, own streams (strips of 32 KiB / 256 KiB) against
a zlib -6 stream with Z_FULL_FLUSH every 32 KiB; SFH_INFLATE_SERIAL=1 for the lane-serial kernel."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from starflate_amd import Compressor, synth, _capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 << 20
data = synth.gen_text_torch(n, seed=3, device="cuda")
c = Compressor(0)
c.set_profiling(True)
def run(name, stream, idx, bb):
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        _, st = c.decompress_tensor(stream, idx, n, out=back, block_bytes=bb)
    nseg = idx.numel() - 1
    info = c.debug(_capi.DBG_SEGINFO, nseg)
    print(f"{name:12s} status {st} equal {bool(torch.equal(back, data))} tokens/segment {info[:,1].mean():8.1f} by lane-serial kernel {int(((info[:,2]>>1)&1).sum())}",
          {k: round(v, 3) for k, v in c.inflate_ms().items()}, flush=True)
for bb in (32768, 262144):
    out, nb = c.compress_tensor(data, block_bytes=bb)
    run(f"own/{bb}", out[:nb].clone(), c.last_index(device="cuda"), bb)
host = data.cpu().numpy()
nseg = n // 32768
for level in (6, 1):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    parts = [co.compress(host[k * 32768:(k + 1) * 32768].tobytes()) + co.flush(zlib.Z_FINISH if k == nseg - 1 else zlib.Z_FULL_FLUSH) for k in range(nseg)]
    idx = torch.from_numpy(np.concatenate([[0], np.cumsum([len(q) for q in parts])]).astype(np.int64)).cuda()
    stream = torch.from_numpy(np.frombuffer(b"".join(parts), np.uint8).copy()).cuda()
    run(f"zlib -{level}", stream, idx, 32768)
