"""Experiment (CPU): how many reconcile rounds does k_lz77's parse need per 512-byte region -- lanes of eight positions, every
lane walking from a speculative entry, the true entry the exclusive prefix maximum of the exits before it -- on the bench text
and on real source text / machine code?  Per-position matches from the oracle (sfo_match_chunk on 32 KiB chunks), the take
rule and the extension of capped matches restated here.  Also tries variants of the scheme (see VARIANTS)."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np
import oracle_lib as O
from starflate_amd import synth, realbytes

R, LANE = 512, 8


def successors(data, ln, ds, P, r0, end):
    """succ[p] for p in [r0, end): where the chain goes from p if it arrives there (region-relative)"""
    MM, lazy, cap = P.min_match, P.lazy, P.cap
    n = end - r0
    succ = np.empty(n, np.int32)
    for i in range(n):
        pos = r0 + i
        l = int(ln[pos])
        take = l >= MM
        k2 = 1
        while take and k2 <= lazy:
            if pos + k2 < end and int(ln[pos + k2]) > l + (k2 - 1):
                take = False
            k2 += 1
        if take:
            if cap and l >= cap:
                maxlen = min(end - pos, 258)
                c = pos - int(ds[pos])
                while l < maxlen and data[pos + l] == data[c + l]:
                    l += 1
                # (the distance-1 run rule can only lengthen a match that is already long: ignored here)
            succ[i] = i + l
        else:
            succ[i] = i + 1
    return succ


def rounds_for_region(succ, variant):
    n = len(succ)
    nl = (n + LANE - 1) // LANE
    # lane transfer: exit position (region-relative) for entry offset e
    def walk(lane, e):
        p = lane * LANE + e
        hi = min(n, (lane + 1) * LANE)
        while p < hi:
            p = int(succ[p])
        return p
    nv = [min(LANE, n - l * LANE) for l in range(nl)]
    entry = [0] * nl
    exit_abs = [walk(l, 0) for l in range(nl)]
    rounds = 0
    while True:
        hops = 2 if variant == "two_scans" else 1
        any_changed = False
        for _ in range(hops):
            pm = 0
            changed = []
            for l in range(nl):
                ne = pm - l * LANE if pm > l * LANE else 0
                if ne != entry[l]:
                    changed.append((l, ne))
                pm = max(pm, exit_abs[l])
            for l, ne in changed:
                entry[l] = ne
                exit_abs[l] = 0 if ne >= nv[l] else walk(l, ne)
            any_changed = any_changed or bool(changed)
            if not changed:
                break
        if not any_changed:
            break
        rounds += 1
    return rounds


def run(name, data, P, nchunks=24, variant="base"):
    rs = []
    toks = 0
    for c in range(nchunks):
        chunk = np.ascontiguousarray(data[c * 32768:(c + 1) * 32768])
        if chunk.size < 32768:
            break
        ln, ds = O.match_chunk(chunk, P)
        for r0 in range(0, 32768, R):
            succ = successors(chunk, ln, ds, P, r0, r0 + R)
            rs.append(rounds_for_region(succ, variant))
            p = 0
            while p < R:
                p = int(succ[p]); toks += 1
    rs = np.array(rs)
    # a workgroup's round = 16 regions side by side: it waits for the slowest
    g = rs[: len(rs) // 16 * 16].reshape(-1, 16)
    print(f"{name:8s} {variant:10s} regions {len(rs)} tokens/region {toks / len(rs):6.1f}  rounds mean {rs.mean():5.1f} p90 {np.percentile(rs, 90):4.0f} max {rs.max():3d}"
          f"   per workgroup round: mean of max over 16 {g.max(axis=1).mean():5.1f}", flush=True)


if __name__ == "__main__":
    P = O.default_params(strip_bytes=32768)
    sets = [("text", synth.gen_text(1 << 20, seed=3))]
    src = realbytes.source(8 << 20)
    if src is not None:
        sets.append(("source", np.frombuffer(src, np.uint8)[2 << 20:]))
    b = realbytes.binary(8 << 20)
    if b is not None:
        sets.append(("binary", np.frombuffer(b, np.uint8)[2 << 20:]))
    for variant in sys.argv[1:] or ["base"]:
        for name, d in sets:
            run(name, d, P, variant=variant)
