import sys, os, zlib, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from starflate_amd import Compressor
c = Compressor(0); c.set_profiling(True)
n = 256 << 20
data, wl = bench.make_input("runs", n, 0, torch.device("cuda", 0))
host = data[:32 << 20].cpu().numpy().tobytes()
zl = bench.zlib6_size(host)
for eff in ("default", "thorough", "max", "chain2", "chain4", "best"):
    out, nb = c.compress_tensor(data, effort=eff)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): out, nb = c.compress_tensor(data, effort=eff)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    from starflate_amd import _capi
    offs = c.debug(_capi.DBG_OFFSETS, n // 32768)
    ours = int(offs[(32 << 20) // 32768])
    ok = zlib.decompress(out[:nb].cpu().numpy().tobytes(), -15) == data.cpu().numpy().tobytes()
    print(eff, "MiB/s", round(n / dt / 2**20), "ratio", round(n / nb, 1), "vs zlib6", round(zl / ours, 4), "rt", ok, flush=True)
