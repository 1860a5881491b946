import sys, os, zlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle_lib as O
from starflate_amd import Compressor, _capi, synth
c = Compressor(0)
text = synth.gen_text(300000, seed=2)
ok = True
for name, data in (("text64k", text[:65536]), ("text", text), ("zeros", np.zeros(70000, np.uint8)), ("tiny", np.frombuffer(b"abcabcabcabcabcabc", np.uint8))):
    for eff, cd in (("best", 8), ("ultra", 16)):
        p = O.default_params(chain_depth=cd)
        got = np.frombuffer(c.compress(data, effort=eff), np.uint8)
        want = O.compress(data, p)
        same = got.size == want.size and np.array_equal(got, want)
        rt = zlib.decompress(bytes(got), -15) == data.tobytes()
        print(name, eff, got.size, want.size, "OK" if same else "DIFF", "rt", rt, flush=True)
        if not same:
            ok = False
            nch = max(1, (data.size + 32767) // 32768)
            toks, flags = c.debug_tokens(nch)
            ref = O.chunk_tokens(data, p)
            for ch in range(nch):
                flat, nt, tarr = ref[ch]
                if toks[ch].size != flat.size or not np.array_equal(toks[ch], flat):
                    m = min(toks[ch].size, flat.size)
                    d = np.flatnonzero(toks[ch][:m] != flat[:m]); k = int(d[0]) if d.size else m
                    print(f"  chunk {ch}: ntok gpu {toks[ch].size} oracle {flat.size}; first diff at token {k}: gpu {[hex(int(x)) for x in toks[ch][k:k+4]]} oracle {[hex(int(x)) for x in flat[k:k+4]]}")
                    # position of token k
                    pos = 0
                    for x in flat[:k]:
                        x = int(x); pos += ((x >> 16) & 0xFF) + 3 if x & 0x80000000 else 1
                    print("   at input position", ch * 32768 + pos)
                    break
sys.exit(0 if ok else 1)
