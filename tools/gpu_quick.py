"""Quick GPU parity probe (development aid): several inputs x strip sizes through the C-ABI against the oracle,
reporting the first stage that differs (tokens -> histogram -> plan -> stream)."""
import os, sys, zlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from starflate_amd import Compressor, _capi, synth

CH = 32768
comp = Compressor(0)
rng = np.random.default_rng(7)
text = synth.gen_text(3_000_000, seed=2)
inputs = {
    "empty": np.zeros(0, np.uint8),
    "one": np.array([65], np.uint8),
    "tiny_rep": np.frombuffer(b"abcabcabcabcabcabcabcabcabc", np.uint8),
    "zeros_ragged": np.zeros(CH * 2 + 777, np.uint8),
    "text_1chunk": text[:CH],
    "text_ragged": text[: CH * 5 + 1234],
    "text_9chunks": text[: CH * 9 + 5],
    "text_big": text,
    "random": rng.integers(0, 256, CH * 3 + 5, dtype=np.uint8),
    "period7": np.tile(np.arange(7, dtype=np.uint8), 9000),
    "mixed": synth.gen_mixed(3 << 20, seed=4, stripe=1 << 16)[: (2 << 20) + 13],
    "rand_then_text": np.concatenate([rng.integers(0, 256, CH * 2, dtype=np.uint8), text[: CH * 3]]),
}
bad = 0
for bb in (0, 32768, 65536, 262144, 1 << 20):
    for name, data in inputs.items():
        p = O.default_params(strip_bytes=bb)
        got = np.frombuffer(comp.compress(data, block_bytes=bb), np.uint8)
        want = O.compress(data, p)
        ok = got.size == want.size and np.array_equal(got, want)
        rt = False
        try:
            rt = zlib.decompress(bytes(got), -15) == data.tobytes()
        except Exception as e:  # noqa
            rt = False
        print(f"bb={bb:8d} {name:16s} n={data.size:8d} got={got.size:8d} want={want.size:8d} {'OK' if ok else 'DIFF'} roundtrip={'ok' if rt else 'FAIL'}", flush=True)
        if ok:
            continue
        bad += 1
        nch = max(1, (data.size + CH - 1) // CH)
        toks, flags = comp.debug_tokens(nch)
        ntok = comp.debug(_capi.DBG_NTOK, nch)
        hist = comp.debug(_capi.DBG_HIST, nch)
        plan = comp.debug(_capi.DBG_PLAN, nch)
        ref = O.chunk_tokens(data, p)
        for c in range(nch):
            flat, nt, tarr = ref[c]
            if toks[c].size != flat.size or not np.array_equal(toks[c], flat):
                m = min(toks[c].size, flat.size)
                d = np.flatnonzero(toks[c][:m] != flat[:m])
                k = int(d[0]) if d.size else m
                print(f"   chunk {c}: tokens differ: ntok gpu {toks[c].size} (ntok_out {ntok[c]}) oracle {flat.size}; first diff at token {k}: "
                      f"gpu {[hex(int(x)) for x in toks[c][k:k+4]]} oracle {[hex(int(x)) for x in flat[k:k+4]]}")
                break
            ll, dd = O.histogram(tarr, nt, p.region_bytes)
            if not (np.array_equal(hist[c, :286], ll) and np.array_equal(hist[c, 288:318], dd)):
                print(f"   chunk {c}: histogram differs")
                break
            pl = O.plan_chunk(ll, dd, min(CH, data.size - c * CH), c + 1 == nch, p)
            if plan[c, 0] != pl.btype or plan[c, 1] != pl.out_bytes:
                print(f"   chunk {c}: plan differs gpu {plan[c]} oracle {pl.btype},{pl.out_bytes}")
                break
        else:
            d = np.flatnonzero(got[: min(got.size, want.size)] != want[: min(got.size, want.size)])
            print(f"   stages equal; stream first diff at byte {d[:4]}")
        if bad >= 6:
            sys.exit(1)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
