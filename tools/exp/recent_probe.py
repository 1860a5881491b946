"""Quick parity probe of SFH_EFFORT_RECENT against the oracle's `recent` specification, naming the first differing token
(development aid; GPU box)."""
import sys, os, zlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle_lib as O
from starflate_amd import Compressor, _capi, synth, realbytes
c = Compressor(0)
text = synth.gen_text(700000, seed=2)
rng = np.random.default_rng(1)
ok = True
EFF = {"recent": dict(recent=1, near_depth=1, link_steps=1), "recent_all": dict(recent=1, near_depth=1, link_steps=1, stride2=0, step=512)}
for op in (0, 1):
    bad, n = c.lds_order_check(op, 1024, 60)
    print("lds order check op", op, "mismatches", bad, "of", n, flush=True)
    ok = ok and bad == 0 and n > 0
cases = [("text64k", text[:65536]), ("text", text), ("zeros", np.zeros(70000, np.uint8)), ("tiny", np.frombuffer(b"abcabcabcabcabcabc", np.uint8)),
         ("mixed", synth.gen_mixed(1 << 20, seed=4, stripe=1 << 15)), ("period7", np.tile(np.arange(7, dtype=np.uint8), 30000)),
         ("random", rng.integers(0, 256, 3 * 32768 + 5, dtype=np.uint8)), ("low", rng.integers(0, 3, 2 * 32768 + 99, dtype=np.uint8)),
         ("source", realbytes.source(4 << 20)[: 3 << 20]), ("binary", realbytes.binary(32 << 20)[16 << 20:(16 << 20) + (2 << 20)]),
         ("empty", np.zeros(0, np.uint8)), ("ragged", text[:8192 * 3 + 1024])]
for eff, RP in EFF.items():
  for name, data in cases:
    for bb in (0, 32768, 524288):
        p = O.default_params(strip_bytes=bb, **RP)
        got = np.frombuffer(c.compress(data, effort=eff, block_bytes=bb), np.uint8)
        want = O.compress(data, p)
        same = got.size == want.size and np.array_equal(got, want)
        rt = zlib.decompress(bytes(got), -15) == data.tobytes()
        print(eff, name, bb, got.size, want.size, "OK" if same else "DIFF", "rt", rt, flush=True)
        if not same:
            ok = False
            nch = max(1, (data.size + 32767) // 32768)
            toks, flags = c.debug_tokens(nch)
            ref = O.chunk_tokens(data, O.default_params(strip_bytes=c.last_block_bytes(), **RP))
            for ch in range(nch):
                flat, nt, tarr = ref[ch]
                if toks[ch].size != flat.size or not np.array_equal(toks[ch], flat):
                    m = min(toks[ch].size, flat.size)
                    d = np.flatnonzero(toks[ch][:m] != flat[:m]); k = int(d[0]) if d.size else m
                    print(f"  chunk {ch}: ntok gpu {toks[ch].size} oracle {flat.size}; first diff at token {k}: gpu {[hex(int(x)) for x in toks[ch][k:k+4]]} oracle {[hex(int(x)) for x in flat[k:k+4]]}")
                    pos = 0
                    for x in flat[:k]:
                        x = int(x); pos += ((x >> 16) & 0xFF) + 3 if x & 0x80000000 else 1
                    print("   at input position", ch * 32768 + pos)
                    break
sys.exit(0 if ok else 1)
