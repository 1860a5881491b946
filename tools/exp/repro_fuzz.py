#!/usr/bin/env python3
"""Development aid: re-runs the seeded stored-fast-path fuzz (tests/test_gpu_parity.py::test_fuzz_stored_fast_path_switching) up to
iteration SF_IT and says, for the failing input, which chunks differ from the specification and how (tokens per chunk, first
differing token)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from starflate_amd import Compressor, _capi, synth

CHUNK = 32768
rng = np.random.default_rng(20261005)
text = synth.gen_text(600_000, seed=91)

def piece(n):
    kind = int(rng.integers(0, 7))
    if kind in (0, 1):
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == 2:
        return rng.integers(0, 64, n, dtype=np.uint8)
    if kind == 3:
        return np.zeros(n, np.uint8)
    if kind == 4:
        a = rng.integers(0, 256, n, dtype=np.uint8)
        k = int(min(n // 3, rng.integers(100, 3000)))
        if k:
            a[n - k:] = a[:k]
        return a
    o = int(rng.integers(0, text.size - n)) if n < text.size else 0
    return text[o:o + n]

efforts = [("default", {}), ("thorough", dict(stride2=0, step=512)), ("recent_all", dict(recent=1, near_depth=1, link_steps=1, stride2=0, step=512)),
           ("best", dict(chain_depth=8)), ("fastest", dict(depth=1, use_near=0)), ("max", dict(stride2=0, step=512, hash_bits=12, long_hash_bytes=7))]
c = Compressor(0)
want_it = int(os.environ.get("SF_IT", "72"))
for it in range(want_it + 1):
    parts = []
    for _ in range(int(rng.integers(1, 9))):
        n = int(rng.choice([CHUNK // 4, CHUNK // 2, CHUNK, CHUNK, 2 * CHUNK, 3 * CHUNK, int(rng.integers(1, 3 * CHUNK))]))
        parts.append(np.ascontiguousarray(piece(n)[:n]))
    data = np.concatenate(parts)
    if it % 3 == 0:
        data = data[: data.size - int(rng.integers(0, min(data.size, CHUNK)))]
    fast = it % 5 != 4
    bb = [0, CHUNK, 2 * CHUNK, 4 * CHUNK, 8 * CHUNK][it % 5]
    lazy = [3, 3, 0, 2][it % 4]
    effort, ekw = efforts[it % len(efforts)]
    if it < want_it:
        continue
    p = O.default_params(lazy=lazy, fast_skip=int(fast), strip_bytes=bb, **ekw)
    got = np.frombuffer(c.compress(data, lazy=lazy, stored_fast_path=fast, block_bytes=bb, effort=effort), np.uint8)
    want, windex, _ = O.compress_indexed(data, p)
    nch = (data.size + CHUNK - 1) // CHUNK
    gindex = c.last_index()
    print("it", it, "size", data.size, "effort", effort, "bb", bb, "lazy", lazy, "fast", fast, "equal", np.array_equal(got, want), got.size, want.size)
    gs, ws = np.diff(gindex.astype(np.int64)), np.diff(windex.astype(np.int64))
    ntok = c.debug(_capi.DBG_NTOK, nch); nit = c.debug(_capi.DBG_NITEMS, nch); plan = c.debug(_capi.DBG_PLAN, nch)
    ot = O.chunk_tokens(data, p)
    for k in range(nch):
        flag = "" if gs[k] == ws[k] else "  <-- size differs"
        print(f"chunk {k}: gpu bytes {gs[k]} spec {ws[k]}  gpu ntok {ntok[k]} nitems {nit[k] & 0x7FFFFFFF} skipped {bool(nit[k] >> 31)} btype {plan[k][0]}  spec ntok {ot[k][0].size}{flag}")
    bad = [k for k in range(nch) if gs[k] != ws[k] or ntok[k] != ot[k][0].size]
    if bad:
        k = bad[0]
        items = c.debug(_capi.DBG_ITEMS, nch)[k, : nit[k] & 0x7FFFFFFF].astype(np.uint32)
        if not (nit[k] >> 31):
            start = (items & 0x8000) != 0
            head = start & ((items & 0x100) != 0)
            nxt = np.zeros(items.size, np.uint32); nxt[:-1] = items[1:]
            tok = np.where(head, np.uint32(0x80000000) | ((items & 0xFF) << 16) | (nxt & 0x7FFF), items & 0xFF)[start]
            w = ot[k][0]
            m = min(tok.size, w.size)
            d = np.flatnonzero(tok[:m] != w[:m])
            print("first bad chunk", k, "tokens gpu", tok.size, "spec", w.size, "first differing token", d[:1])
            if d.size:
                i = int(d[0])
                def pos_of(t, i):
                    return int(sum((((x >> 16) & 0xFF) + 3) if x & 0x80000000 else 1 for x in t[:i]))
                print("  at chunk position", pos_of(w, i), "gpu", [hex(int(x)) for x in tok[i:i + 4]], "spec", [hex(int(x)) for x in w[i:i + 4]])
