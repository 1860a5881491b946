"""How many steps would a run-batched (whole-wave) two-queue Huffman merge take on real histograms?  Simulates, on the literal/length and
distance histograms of real chunks (oracle tokens), the merge that pairs whole runs of leaves / of internal nodes per step and falls
back to one merge where a leaf meets a node -- bit-exact with the serial two-queue rule -- and prints steps against symbols (CPU)."""
import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, oracle_lib as O
from starflate_amd import synth, realbytes
INF = 1 << 40
def sim(freqs):
    w = sorted((int(f), s) for s, f in enumerate(freqs) if f)
    m = len(w)
    if m < 2: return m, 0, 0, 0, 0
    W = [x[0] for x in w] + [0] * (m - 1)
    i, j, k = 0, m, m
    it = ll = nn = mx = 0
    while k < 2 * m - 1:
        it += 1
        a0 = W[i] if i < m else INF; a1 = W[i + 1] if i + 1 < m else INF
        b0 = W[j] if j < k else INF; b1 = W[j + 1] if j + 1 < k else INF
        if a1 != INF and a1 <= b0:
            T = 0
            for s in range(64):
                if i + 2 * s + 1 >= m: break
                H = b0 if j < k else (INF if s == 0 else a0 + a1)
                if W[i + 2 * s + 1] <= H: T += 1
                else: break
            for s in range(T): W[k + s] = W[i + 2 * s] + W[i + 2 * s + 1]
            i += 2 * T; k += T; ll += 1
        elif b1 != INF and b1 < a0:
            T = 0
            k0 = k
            for s in range(64):
                if j + 2 * s + 1 >= k0: break
                if W[j + 2 * s + 1] < a0: T += 1
                else: break
            for s in range(T): W[k + s] = W[j + 2 * s] + W[j + 2 * s + 1]
            j += 2 * T; k += T; nn += 1
        else:
            W[k] = a0 + b0; i += 1; j += 1; k += 1; mx += 1
    return m, it, ll, nn, mx
def chunks(data, n=24):
    p = O.default_params(strip_bytes=262144)
    out = []
    for tok, nt, tarr in O.chunk_tokens(data[: n * 32768], p):
        ll, d = O.histogram(tarr, nt, p.region_bytes)
        out.append((ll, d))
    return out
for name, data in (("text", synth.gen_text(1 << 20, seed=3)), ("source", realbytes.source(8 << 20)[4 << 20:]), ("binary", realbytes.binary(20 << 20)[16 << 20:]), ("mixed", synth.gen_mixed(1 << 20, seed=4))):
    rows = [sim(ll) for ll, d in chunks(data)]
    rd = [sim(d) for ll, d in chunks(data)]
    a = np.array(rows, float); b = np.array(rd, float)
    print(f"{name:7s} ll: symbols {a[:,0].mean():.0f}  iterations {a[:,1].mean():.0f} (leaf runs {a[:,2].mean():.0f}, node runs {a[:,3].mean():.0f}, mixed {a[:,4].mean():.0f})   merges/iter {((a[:,0]-1)/a[:,1]).mean():.2f} | d: symbols {b[:,0].mean():.0f} iterations {b[:,1].mean():.0f}")
