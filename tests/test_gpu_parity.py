"""GPU parity: the HIP pipeline (through the C-ABI) must be bit-exact equal to the
CPU oracle's encoder specification, and every stream must round-trip through the
oracle's restatement of the reference decoder (src/decompress.cpp:402-461) and
through zlib inflate.  Integer/byte work: the bar is bit-exact."""
import zlib

import numpy as np
import pytest

import oracle_lib as O
from starflate_amd import _capi, synth

pytestmark = pytest.mark.gpu

CHUNK = 32768


def _params(strategy="auto", final_stream=True, lazy=True, block_bytes=0):
    """The oracle parameters that stand for the library's options (block_bytes 0 = the default rule of both)."""
    return O.default_params(strategy=_capi.STRATEGY[strategy], final_stream=int(final_stream), lazy=3 if lazy is True else int(lazy),
                            strip_bytes=block_bytes)


def _inputs(starfleet):
    rng = np.random.default_rng(7)
    text = synth.gen_text(200_000, seed=2)
    return {
        "empty": np.zeros(0, np.uint8),
        "one": np.array([65], np.uint8),
        "three": np.frombuffer(b"abc", np.uint8),
        "tiny_rep": np.frombuffer(b"abcabcabcabcabcabcabcabcabc", np.uint8),
        "zeros_1chunk": np.zeros(CHUNK, np.uint8),
        "zeros_ragged": np.zeros(CHUNK * 2 + 777, np.uint8),
        "text_1chunk": text[:CHUNK],
        "text_ragged": text[: CHUNK * 5 + 1234],
        "text_chunk_minus1": text[: CHUNK - 1],
        "text_chunk_plus1": text[: CHUNK + 1],
        "starfleet": np.frombuffer(starfleet, np.uint8),
        "random": rng.integers(0, 256, CHUNK * 3 + 5, dtype=np.uint8),
        "low_entropy": rng.integers(0, 4, CHUNK + 99, dtype=np.uint8),
        "period7": np.tile(np.arange(7, dtype=np.uint8), 9000),
        "mixed": synth.gen_mixed(3 << 20, seed=4, stripe=1 << 16)[: (1 << 20) + 13],
    }


def _roundtrip(stream, data):
    st, w, out = O.decompress(stream, data.size)
    assert st == 0, f"oracle decompress status {st}"
    assert w == data.size and np.array_equal(out, data)
    assert zlib.decompress(bytes(stream), -15) == data.tobytes()


@pytest.mark.parametrize("strategy", ["auto", "stored", "fixed", "dynamic"])
def test_bit_exact_vs_oracle(compressor, starfleet, strategy):
    for name, data in _inputs(starfleet).items():
        got = np.frombuffer(compressor.compress(data, strategy=strategy), np.uint8)
        want = O.compress(data, _params(strategy))
        _roundtrip(got, data)
        assert got.size == want.size, f"{name}/{strategy}: size {got.size} != {want.size}"
        assert np.array_equal(got, want), f"{name}/{strategy}: first diff at {np.flatnonzero(got != want)[:4]}"


@pytest.mark.parametrize("block_bytes", [32768, 65536, 262144, 1 << 20])
def test_block_bytes_bit_exact_vs_oracle(compressor, starfleet, block_bytes):
    """sfh_options.block_bytes (the strip that is coded independently): the window slides across the strip's
    DEFLATE blocks, the stream still equals the specification's and round-trips; larger strips are never worse."""
    text = synth.gen_text(40 * CHUNK + 321, seed=12)
    cases = {"text": text, "mixed": synth.gen_mixed(3 << 20, seed=4, stripe=1 << 16)[: (2 << 20) + 13],
             "starfleet": np.frombuffer(starfleet, np.uint8), "zeros": np.zeros(9 * CHUNK + 5, np.uint8),
             "rand_then_text": np.concatenate([np.random.default_rng(3).integers(0, 256, 2 * CHUNK, dtype=np.uint8), text[: 3 * CHUNK]])}
    for name, data in cases.items():
        for strategy in ("auto", "fixed"):
            got = np.frombuffer(compressor.compress(data, strategy=strategy, block_bytes=block_bytes), np.uint8)
            assert compressor.last_block_bytes() == block_bytes
            want = O.compress(data, _params(strategy, block_bytes=block_bytes))
            assert np.array_equal(got, want), f"{name}/{strategy}/{block_bytes}"
            _roundtrip(got, data)
    small = len(compressor.compress(text, block_bytes=32768))
    assert len(compressor.compress(text, block_bytes=block_bytes)) <= small
    with pytest.raises(Exception):
        compressor.compress(text, block_bytes=32768 + 512)  # not a multiple of 32 KiB


def test_effort_fast_bit_exact_vs_oracle(compressor, starfleet):
    """sfh_options.effort = SFH_EFFORT_FAST: one history level per hash bucket (the specification's depth = 1)."""
    text = synth.gen_text(20 * CHUNK + 77, seed=14)
    for data in (text, np.frombuffer(starfleet, np.uint8), synth.gen_mixed(1 << 20, seed=4, stripe=1 << 15)):
        for bb in (0, 131072):
            got = np.frombuffer(compressor.compress(data, effort="fast", block_bytes=bb), np.uint8)
            want = O.compress(data, O.default_params(depth=1, strip_bytes=bb))
            assert np.array_equal(got, want)
            _roundtrip(got, data)
            assert got.size >= len(compressor.compress(data, block_bytes=bb))
            # SFH_EFFORT_FASTEST: that, and no step-local candidate (the specification's use_near = 0)
            got2 = np.frombuffer(compressor.compress(data, effort="fastest", block_bytes=bb), np.uint8)
            want2 = O.compress(data, O.default_params(depth=1, use_near=0, strip_bytes=bb))
            assert np.array_equal(got2, want2)
            _roundtrip(got2, data)
            assert got2.size >= got.size
            # SFH_EFFORT_THOROUGH: every position is searched, in steps of 512 (the specification's stride2 = 0, step = 512)
            got3 = np.frombuffer(compressor.compress(data, effort="thorough", block_bytes=bb), np.uint8)
            want3 = O.compress(data, O.default_params(stride2=0, step=512, strip_bytes=bb))
            assert np.array_equal(got3, want3)
            _roundtrip(got3, data)
            # SFH_EFFORT_MAX: that, with two tables of 4096 buckets keyed by four and by seven bytes
            got4 = np.frombuffer(compressor.compress(data, effort="max", block_bytes=bb), np.uint8)
            want4 = O.compress(data, O.default_params(stride2=0, step=512, hash_bits=12, long_hash_bytes=7, strip_bytes=bb))
            assert np.array_equal(got4, want4)
            _roundtrip(got4, data)


@pytest.mark.parametrize("effort", ["best", "ultra", "extreme"])
def test_effort_chains_bit_exact_vs_oracle(compressor, starfleet, effort):
    """SFH_EFFORT_BEST / _ULTRA / _EXTREME: exact hash chains of depth 8 / 16 / 32 (the specification's chain_depth: every position inserted
    and searched, most recent candidate first) -- bit-exact like every other effort, over inputs whose chains look very
    different (text, the reference's HTML file, mixed stripes, one value, short periods, noise), strips that reach past one
    batch of the links' ring, ragged tails, every strategy and lazy level; never larger than the default effort's stream
    on the compressible ones."""
    rng = np.random.default_rng(5)
    text = synth.gen_text(44 * CHUNK + 77, seed=14)
    cases = {"text": text, "starfleet": np.frombuffer(starfleet, np.uint8), "mixed": synth.gen_mixed(1 << 20, seed=4, stripe=1 << 15),
             "zeros": np.zeros(5 * CHUNK + 333, np.uint8), "period7": np.tile(np.arange(7, dtype=np.uint8), 30000),
             "period300": np.tile(rng.integers(0, 256, 300, dtype=np.uint8), 700), "random": rng.integers(0, 256, 3 * CHUNK + 5, dtype=np.uint8),
             "low_entropy": rng.integers(0, 3, 2 * CHUNK + 99, dtype=np.uint8), "tiny": np.frombuffer(b"abcabcabcabcabcabc", np.uint8),
             "empty": np.zeros(0, np.uint8), "three": np.frombuffer(b"xyz", np.uint8)}
    for name, data in cases.items():
        for bb in (0, 32768, 131072, 1 << 20):
            got = np.frombuffer(compressor.compress(data, effort=effort, block_bytes=bb), np.uint8)
            want = O.compress(data, O.default_params(strip_bytes=bb, **EFFORT_PARAMS[effort]))
            assert got.size == want.size and np.array_equal(got, want), (name, bb, np.flatnonzero(got[:min(got.size, want.size)] != want[:min(got.size, want.size)])[:3])
            _roundtrip(got, data)
        if name in ("text", "starfleet", "mixed"):
            assert len(compressor.compress(data, effort=effort)) < len(compressor.compress(data))
    for strategy in ("fixed", "dynamic", "stored"):
        for lazy in (0, 1, 3):
            got = np.frombuffer(compressor.compress(text, effort=effort, strategy=strategy, lazy=lazy, stored_fast_path=False), np.uint8)
            want = O.compress(text, O.default_params(strategy=_capi.STRATEGY[strategy], lazy=lazy, fast_skip=0, **EFFORT_PARAMS[effort]))
            assert np.array_equal(got, want), (strategy, lazy)
    if effort == "best":
        # sfh_options.chain_depth: any depth with a chain effort ("chain4" = SFH_EFFORT_BEST with chain_depth 4); not with a table effort
        for depth in (1, 2, 4, 5, 33):
            got = np.frombuffer(compressor.compress(text, effort=f"chain{depth}"), np.uint8)
            assert np.array_equal(got, O.compress(text, O.default_params(chain_depth=depth))), depth
        o = _capi.make_options(effort="thorough")
        o.chain_depth = 4
        import ctypes as C
        n_out = C.c_size_t(0)
        buf = np.empty(compressor.compress_bound(text.size), np.uint8)
        assert compressor._lib.sfh_compress(compressor._h, text.ctypes.data, text.size, buf.ctypes.data, buf.size, C.byref(n_out), C.byref(o)) == -1
    # the GPU decoder reads these streams like any other (index + sub-index)
    got = compressor.compress(text, effort=effort)
    back, st = compressor.decompress(got, compressor.last_index(), text.size, subindex=compressor.last_subindex(), block_bytes=compressor.last_block_bytes())
    assert st == 0 and back == text.tobytes()


def test_batches_give_the_stream_of_one_launch(monkeypatch):
    """Inputs beyond one batch (1 GiB) run the four kernels batch after batch with bounded scratch; a small batch
    size (SFH_BATCH_CHUNKS, read at sfh_create) exercises the loop: same stream, same index, wrapped or not."""
    from starflate_amd import Compressor

    data = synth.gen_text(37 * CHUNK + 999, seed=15)
    for bc, bb in ((4, 32768), (8, 65536), (3, 131072), (16, 0)):
        monkeypatch.setenv("SFH_BATCH_CHUNKS", str(bc))
        c = Compressor(0)
        for container in ("raw", "gzip"):
            got = np.frombuffer(c.compress(data, block_bytes=bb, container=container), np.uint8)
            want, widx, wsub = O.compress_indexed(data, O.default_params(strip_bytes=bb, container=_capi.CONTAINER[container]))
            assert np.array_equal(got, want), (bc, bb, container)
            assert np.array_equal(c.last_index(), widx) and np.array_equal(c.last_subindex(), wsub)
            back, st = c.decompress(got, c.last_index(), data.size, subindex=c.last_subindex(), block_bytes=c.last_block_bytes())
            assert st == 0 and back == data.tobytes()
        # a chain effort through the same loop, wrapped and as a non-final shard
        got = np.frombuffer(c.compress(data, block_bytes=bb, container="zlib", effort="best"), np.uint8)
        want = O.compress(data, O.default_params(strip_bytes=bb, container=_capi.CONTAINER["zlib"], chain_depth=8))
        assert np.array_equal(got, want), (bc, bb, "best/zlib")
        got = np.frombuffer(c.compress(data, block_bytes=bb, final_stream=False, effort="chain3"), np.uint8)
        assert np.array_equal(got, O.compress(data, O.default_params(strip_bytes=bb, final_stream=0, chain_depth=3))), (bc, bb, "chain3/shard")
        c.close()


def test_plan_in_three_launches_and_in_one(monkeypatch):
    """k_plan runs as sort / merge (a chunk per LANE) / finish since round 6; SFH_PLAN_FUSED=1 (read at sfh_create) is the one
    launch of rounds 1-5.  Both give the specification's code lengths: alphabets of every width (one symbol, two, all 256
    literals + every length + every distance), chunk counts that do not fill the merge kernel's waves (1, 63, 64, 65, 130),
    stored and fixed chunks in between (their lanes leave early), and more than one batch."""
    from starflate_amd import Compressor

    rng = np.random.default_rng(66)
    pieces = [synth.gen_text(3 * CHUNK + 17, seed=61), rng.integers(0, 256, CHUNK + 5, dtype=np.uint8), np.zeros(CHUNK, np.uint8),
              np.full(40, 7, np.uint8), rng.integers(0, 2, 2 * CHUNK, dtype=np.uint8), synth.gen_mixed(4 * CHUNK, seed=62, stripe=1 << 14),
              np.frombuffer(rng.bytes(9000) * 11, np.uint8), rng.integers(0, 64, CHUNK - 1, dtype=np.uint8)]
    wide = np.concatenate(pieces)
    cases = [wide[:1], wide[:CHUNK], wide[: 63 * CHUNK // 4], np.concatenate([wide] * 5)[: 130 * CHUNK - 3], synth.gen_text(65 * CHUNK, seed=63)]
    for fused, bc in (("0", None), ("1", None), ("0", "3")):
        monkeypatch.setenv("SFH_PLAN_FUSED", fused)
        if bc:
            monkeypatch.setenv("SFH_BATCH_CHUNKS", bc)
        c = Compressor(0)
        for k, data in enumerate(cases):
            for effort, ekw in (("default", {}), ("best", dict(chain_depth=8))):
                got = np.frombuffer(c.compress(data, effort=effort), np.uint8)
                assert np.array_equal(got, O.compress(data, O.default_params(**ekw))), (fused, bc, k, effort)
        c.close()


def test_largest_and_odd_strips(compressor):
    """block_bytes at its maximum (16 MiB: the table's step codes are aged a thousand times, the decoder walks 512
    segments per strip) and at a non-power-of-two multiple of 32 KiB: bit-exact, and decodable on the GPU."""
    data = np.concatenate([synth.gen_text(20 << 20, seed=71), synth.gen_mixed(13 << 20, seed=72, stripe=1 << 18)[: (13 << 20) - 77]])
    for bb, effort, ekw in ((16 << 20, "default", {}), (3 * CHUNK, "default", {}), (16 << 20, "thorough", dict(stride2=0, step=512)), (16 << 20, "max", dict(stride2=0, step=512, hash_bits=12, long_hash_bytes=7)),
                            (5 * CHUNK, "fastest", dict(depth=1, use_near=0)),
                            # chains: the heads' step codes aged a thousand times, the links' ring wrapped four hundred times
                            (16 << 20, "best", dict(chain_depth=8)), (7 * CHUNK, "ultra", dict(chain_depth=16))):
        got = np.frombuffer(compressor.compress(data, block_bytes=bb, effort=effort), np.uint8)
        assert np.array_equal(got, O.compress(data, O.default_params(strip_bytes=bb, **ekw))), (bb, effort)
        back, st = compressor.decompress(got, compressor.last_index(), data.size, subindex=compressor.last_subindex(), block_bytes=bb)
        assert st == 0 and back == data.tobytes(), (bb, effort)
    with pytest.raises(Exception):
        compressor.compress(data[:CHUNK], block_bytes=(16 << 20) + CHUNK)  # beyond the maximum


def test_default_block_bytes_rule(compressor):
    """block_bytes = 0 stands for a function of the input size alone (the oracle and the library share the rule)."""
    for n in (0, 1, CHUNK, 50 * CHUNK, (8 << 20) + 5, 16 << 20, (64 << 20) + 1):
        data = np.zeros(n, np.uint8)
        compressor.compress(data, strategy="stored")
        assert compressor.last_block_bytes() == _capi.resolve_block_bytes(0, n) == O.resolve_strip_bytes(O.default_params(), n), n


def test_stage_parity(compressor, starfleet):
    """tokens / histogram / code lengths / plan of every chunk against the oracle stages (strips of two chunks:
    the second chunk of a strip matches into the first)."""
    data = np.concatenate([np.frombuffer(starfleet, np.uint8), synth.gen_text(3 * CHUNK + 99, seed=13)])
    for block_bytes in (2 * CHUNK, 0):
        p = _params(block_bytes=block_bytes)
        compressor.compress(data, block_bytes=block_bytes)
        nchunks = (data.size + CHUNK - 1) // CHUNK
        ntok = compressor.debug(_capi.DBG_NTOK, nchunks)
        toks, flags = compressor.debug_tokens(nchunks)
        hist = compressor.debug(_capi.DBG_HIST, nchunks)
        lens = compressor.debug(_capi.DBG_LENS, nchunks)
        plan = compressor.debug(_capi.DBG_PLAN, nchunks)
        ref = O.chunk_tokens(data, p)
        assert len(ref) == nchunks
        per_sub = 1024 // p.region_bytes  # parse regions per sub-index entry
        for c in range(nchunks):
            flat, nt, tarr = ref[c]
            assert ntok[c] == flat.size == toks[c].size, f"chunk {c}: ntok"
            assert np.array_equal(toks[c], flat), f"chunk {c}: tokens"
            # the first token of every 1024 bytes is flagged with its index (sub-index transport, k_lz77 -> k_emit)
            starts = np.concatenate([[0], np.cumsum(nt)[:-1]]).astype(np.int64)[::per_sub]
            assert flags[c] == [(int(k), r) for r, k in enumerate(starts)], f"chunk {c}: region flags"
            ll, dd = O.histogram(tarr, nt, p.region_bytes)
            assert np.array_equal(hist[c, :286], ll) and np.array_equal(hist[c, 288:318], dd), f"chunk {c}: hist"
            pl = O.plan_chunk(ll, dd, min(CHUNK, data.size - c * CHUNK), c + 1 == nchunks, p)
            assert np.array_equal(lens[c, :288], np.frombuffer(pl.ll_lens, np.uint8)), f"chunk {c}: ll lens"
            assert np.array_equal(lens[c, 288:320], np.frombuffer(pl.d_lens, np.uint8)), f"chunk {c}: d lens"
            assert plan[c, 0] == pl.btype and plan[c, 1] == pl.out_bytes, f"chunk {c}: plan {plan[c]} vs {pl.btype},{pl.out_bytes}"


def test_lazy_levels(compressor, starfleet):
    """Every look-ahead depth of the lazy rule (0 = greedy .. 3 = default) against the oracle."""
    for data in (synth.gen_text(3 * CHUNK + 17, seed=6), np.frombuffer(starfleet, np.uint8)):
        sizes = []
        for lazy in (0, 1, 2, 3):
            got = np.frombuffer(compressor.compress(data, lazy=lazy), np.uint8)
            assert np.array_equal(got, O.compress(data, _params(lazy=lazy))), f"lazy={lazy}"
            _roundtrip(got, data)
            sizes.append(got.size)
        assert sizes[3] < sizes[0]


def test_stored_fast_path(compressor):
    """High-entropy chunks skip the search after their first 8 KiB; the rule is part of the
    specification, so the stream still equals the oracle's, with the switch on and off."""
    rng = np.random.default_rng(21)
    rnd = rng.integers(0, 256, 5 * CHUNK + 999, dtype=np.uint8)
    head_then_zeros = np.concatenate([rnd[:9000], np.zeros(3 * CHUNK, np.uint8)])
    # six bits of entropy per byte and no matches: the fast path fires, yet a dynamic block wins -- k_emit then takes the
    # items behind the first 8 KiB from the input itself (k_lz77 never wrote them); with a compressible stretch in between
    base64ish = rng.integers(0, 64, 7 * CHUNK + 4321, dtype=np.uint8)
    turns = np.concatenate([base64ish[: 3 * CHUNK], synth.gen_text(2 * CHUNK + 100, seed=31), base64ish[3 * CHUNK:], rnd[: 2 * CHUNK]])
    for data in (rnd, head_then_zeros, synth.gen_mixed(1 << 20, seed=4, stripe=1 << 15), base64ish, turns):
        for on in (True, False):
            for bb in (0, 4 * CHUNK):
                got = np.frombuffer(compressor.compress(data, stored_fast_path=on, block_bytes=bb), np.uint8)
                want = O.compress(data, O.default_params(fast_skip=int(on), strip_bytes=bb))
                assert np.array_equal(got, want)
                _roundtrip(got, data)
    got = compressor.compress(turns, strategy="dynamic")  # (every chunk a Huffman block: the fast path's chunks included)
    assert np.array_equal(np.frombuffer(got, np.uint8), O.compress(turns, O.default_params(strategy=3)))
    back, st = compressor.decompress(got, compressor.last_index(), turns.size, subindex=compressor.last_subindex(),
                                     block_bytes=compressor.last_block_bytes())
    assert st == 0 and back == turns.tobytes()
    assert len(compressor.compress(base64ish)) < 0.8 * base64ish.size
    # every match finder goes through the same fast path (short probe behind a skipped chunk, nothing written for one): the
    # chains with their ring of links, the step tables with steps of 512, the ordered insertion -- wrapped, too
    for effort in ("thorough", "max", "best", "recent", "recent_all", "fastest"):
        for bb in (0, 8 * CHUNK):
            got = np.frombuffer(compressor.compress(turns, effort=effort, block_bytes=bb), np.uint8)
            assert np.array_equal(got, O.compress(turns, O.default_params(strip_bytes=bb, **EFFORT_PARAMS[effort]))), (effort, bb)
    import zlib as _z
    assert _z.decompress(compressor.compress(turns, container="gzip", effort="recent_all"), 31) == turns.tobytes()
    assert len(compressor.compress(rnd)) == rnd.size + 5 * 6  # six stored blocks
    # a skipped block is not inserted into the hash tables either: the next block of the strip sees none of it
    two = np.concatenate([rnd[:CHUNK], rnd[:CHUNK]])
    assert np.array_equal(np.frombuffer(compressor.compress(two, block_bytes=2 * CHUNK), np.uint8),
                          O.compress(two, O.default_params(strip_bytes=2 * CHUNK)))


def test_stored_by_the_probe(compressor):
    """Round 6: a full chunk that takes the stored fast path and whose probe span's BYTES are as good as uniform is stored by
    k_lz77 itself -- no tokens, no byte counts, its other 24 KiB never fetched -- when the block type is the encoder's choice.
    Same rule in the specification (probe_span_is_noise): streams, token counts and histograms agree; a noise head in front of
    six-bit bytes stores the whole chunk, 7.9 bits of entropy per byte do not, a forced block type keeps the literals, a strip's
    short last chunk is counted as before."""
    rng = np.random.default_rng(77)
    rnd = rng.integers(0, 256, 6 * CHUNK, dtype=np.uint8)
    six = rng.integers(0, 64, 6 * CHUNK, dtype=np.uint8)
    pw = np.ones(256)
    pw[:64] = 0.3
    skew = rng.choice(256, size=4 * CHUNK, p=pw / pw.sum()).astype(np.uint8)  # 7.88 bits per byte: a Huffman block of literals
    head = np.concatenate([rnd[:8192], six[: CHUNK - 8192], six[:CHUNK], rnd[: CHUNK + 9000], six[:5000]])
    tail = np.concatenate([six[:8192], rnd[: CHUNK - 8192], rnd[: 2 * CHUNK + 8200]])  # (the last chunk: 8200 bytes, not a full one)
    for k, data in enumerate((rnd, head, tail, skew, np.concatenate([rnd[: 2 * CHUNK], synth.gen_text(3 * CHUNK, seed=5), rnd[: 3 * CHUNK]]))):
        nch = (data.size + CHUNK - 1) // CHUNK
        for effort in ("default", "thorough", "best", "recent_all"):
            for bb in (0, 4 * CHUNK):
                par = O.default_params(strip_bytes=bb, **EFFORT_PARAMS[effort])
                got = np.frombuffer(compressor.compress(data, effort=effort, block_bytes=bb), np.uint8)
                want, widx, wsub = O.compress_indexed(data, par)
                assert np.array_equal(got, want), (k, effort, bb)
                assert np.array_equal(compressor.last_subindex(), wsub), (k, effort, bb)
                _roundtrip(got, data)
        # token counts per chunk: none where the probe stored the chunk
        compressor.compress(data, block_bytes=4 * CHUNK)
        ntok = compressor.debug(_capi.DBG_NTOK, nch)
        want_ntok = []
        for s0 in range(0, data.size, 4 * CHUNK):
            _, nt = O.strip_tokens(data[s0: s0 + 4 * CHUNK], O.default_params(strip_bytes=4 * CHUNK))
            R = 512
            sn = min(4 * CHUNK, data.size - s0)
            want_ntok += [int(nt[c0 // R: (min(c0 + CHUNK, sn) + R - 1) // R].sum()) for c0 in range(0, sn, CHUNK)]
        assert [int(x) for x in ntok] == want_ntok, (k, ntok, want_ntok)
        plan = compressor.debug(_capi.DBG_PLAN, nch)  # (k_plan goes by the token count: such a chunk has no histogram)
        for c in range(nch):
            if want_ntok[c] == 0:
                assert plan[c, 0] == 0 and plan[c, 1] == CHUNK + 5, (k, c, plan[c])
        for strategy in ("dynamic", "fixed"):  # a forced block type: the chunk's literals are coded
            got = np.frombuffer(compressor.compress(data, strategy=strategy), np.uint8)
            assert np.array_equal(got, O.compress(data, O.default_params(strategy=_capi.STRATEGY[strategy]))), (k, strategy)
            _roundtrip(got, data)
    compressor.compress(rnd, block_bytes=4 * CHUNK)
    assert not compressor.debug(_capi.DBG_NTOK, 6).any()
    compressor.compress(head, block_bytes=4 * CHUNK)
    nt = compressor.debug(_capi.DBG_NTOK, 4)
    assert nt[0] == 0 and nt[1] == CHUNK and nt[2] == 0 and nt[3] > 0, nt  # noise head: stored whole; six-bit chunk: literals
    compressor.compress(skew, block_bytes=4 * CHUNK)
    assert compressor.debug(_capi.DBG_NTOK, 4).all()
    assert len(compressor.compress(six)) < 0.8 * six.size


EFFORT_PARAMS = {"default": {}, "fast": {"depth": 1}, "fastest": {"depth": 1, "use_near": 0}, "thorough": {"stride2": 0, "step": 512},
                 "max": {"stride2": 0, "step": 512, "hash_bits": 12, "long_hash_bytes": 7},
                 "best": {"chain_depth": 8}, "ultra": {"chain_depth": 16}, "extreme": {"chain_depth": 32},
                 "recent": {"recent": 1, "near_depth": 1, "link_steps": 1},
                 "recent_all": {"recent": 1, "near_depth": 1, "link_steps": 1, "stride2": 0, "step": 512}}


@pytest.mark.parametrize("effort", sorted(EFFORT_PARAMS))
def test_all_literal_chunks_every_effort(compressor, effort):
    """items == positions: the round's distances are staged in the chunk's own item array (k_lz77), which only works
    because a chunk never holds more items than positions.  The tight case is a chunk of literals only with the stored
    fast path off -- every slot of the staging area is then overwritten by an item -- for every effort, forced to a
    Huffman block so the items are what reaches the stream; and next to it chunks of minimum-length matches."""
    rng = np.random.default_rng(99)
    rnd = rng.integers(0, 256, 3 * CHUNK + 517, dtype=np.uint8)
    # every four bytes a match of exactly four bytes (two items per four positions) between unmatched literals
    quad = np.concatenate([rng.integers(0, 256, 4096, dtype=np.uint8).reshape(-1, 4).repeat(2, axis=0).reshape(-1)] * 5)
    for data in (rnd, quad, np.concatenate([rnd[:CHUNK], quad[:CHUNK + 3]])):
        for strategy in ("dynamic", "auto"):
            for bb in (32768, 131072):
                got = np.frombuffer(compressor.compress(data, strategy=strategy, stored_fast_path=False, block_bytes=bb, effort=effort), np.uint8)
                want = O.compress(data, O.default_params(strategy=_capi.STRATEGY[strategy], fast_skip=0, strip_bytes=bb, **EFFORT_PARAMS[effort]))
                assert np.array_equal(got, want), f"{effort}/{strategy}/{bb}"
                _roundtrip(got, data)
    nch = (rnd.size + CHUNK - 1) // CHUNK
    compressor.compress(rnd, strategy="dynamic", stored_fast_path=False, block_bytes=32768, effort=effort)
    nitems = compressor.debug(_capi.DBG_NITEMS, nch)
    assert int(nitems[0]) >= CHUNK - 64, nitems  # (random bytes: a stray 4-byte match at most here and there)


def test_device_tensor_path_and_shard_concat(compressor):
    """Device-buffer entry point; two non-final/final shards concatenate into one valid stream."""
    import torch

    data = synth.gen_text(CHUNK * 6 + 321, seed=9)
    src = torch.from_numpy(data).cuda()
    out, n = compressor.compress_tensor(src)
    whole = out[:n].cpu().numpy()
    assert np.array_equal(whole, O.compress(data, _params()))
    cut = CHUNK * 3
    a, na = compressor.compress_tensor(src[:cut].clone(), final_stream=False)
    a = a[:na].cpu().numpy()
    b, nb = compressor.compress_tensor(src[cut:].clone(), final_stream=True)
    b = b[:nb].cpu().numpy()
    assert np.array_equal(a, O.compress(data[:cut], _params(final_stream=False)))
    _roundtrip(np.concatenate([a, b]), data)


def test_size_independent_properties_large(compressor):
    """64 MiB: round trip through zlib inflate + determinism (same bytes on a second run)."""
    import torch

    data = synth.gen_text(64 << 20, seed=3)
    src = torch.from_numpy(data).cuda()
    out, n = compressor.compress_tensor(src)
    s1 = out[:n].cpu().numpy().copy()
    out2, n2 = compressor.compress_tensor(src)
    assert n2 == n and torch.equal(out2[:n2], out[:n])
    assert zlib.decompress(bytes(s1), -15) == data.tobytes()
    offs = compressor.debug(_capi.DBG_OFFSETS, (data.size + CHUNK - 1) // CHUNK)
    assert np.all(np.diff(offs.astype(np.int64)) > 0) and offs[0] == 0


@pytest.mark.parametrize("container", ["zlib", "gzip"])
def test_container_wrappers(compressor, starfleet, container):
    """RFC 1950 / RFC 1952 wrappers written on the GPU (checksum kernels): bit-exact with the oracle's
    wrapper, body identical to the raw stream, and accepted by zlib's wrapper-checking inflate."""
    import gzip

    kind = _capi.CONTAINER[container]
    h, t = (2, 4) if container == "zlib" else (10, 8)
    for name, data in _inputs(starfleet).items():
        got = np.frombuffer(compressor.compress(data, container=container), np.uint8)
        want = O.compress(data, O.default_params(container=kind))
        assert np.array_equal(got, want), f"{name}: first diff at {np.flatnonzero(got[:min(got.size, want.size)] != want[:min(got.size, want.size)])[:4]}"
        raw = np.frombuffer(compressor.compress(data), np.uint8)
        assert np.array_equal(got[h:-t], raw), name
        assert zlib.decompress(bytes(got), 15 if container == "zlib" else 31) == data.tobytes(), name
        if container == "gzip":
            assert gzip.decompress(bytes(got)) == data.tobytes()
        _roundtrip(got[h:-t], data)
    with pytest.raises(Exception):  # a non-final shard has no trailer
        compressor.compress(b"abc", container=container, final_stream=False)


def test_checksum_kernels_match_zlib(compressor):
    """sfh_checksum_device against zlib.crc32 / zlib.adler32 (and the oracle's bit-serial definitions) on
    empty, sub-chunk, ragged, all-0xFF (Adler worst case) and > 1024-chunk inputs; combine rules."""
    import torch

    from starflate_amd import checksum_combine

    rng = np.random.default_rng(17)
    cases = [np.zeros(0, np.uint8), np.array([1], np.uint8), rng.integers(0, 256, 127, dtype=np.uint8),
             rng.integers(0, 256, CHUNK, dtype=np.uint8), rng.integers(0, 256, CHUNK + 1, dtype=np.uint8),
             np.full(3 * CHUNK + 4097, 255, np.uint8), rng.integers(0, 256, 1500 * CHUNK + 12345, dtype=np.uint8),
             synth.gen_text(2050 * CHUNK - 1, seed=8)]
    for data in cases:
        src = torch.from_numpy(data).cuda() if data.size else torch.empty(0, dtype=torch.uint8, device="cuda")
        b = data.tobytes()
        assert compressor.checksum_tensor(src, "gzip") == zlib.crc32(b), data.size
        assert compressor.checksum_tensor(src, "zlib") == zlib.adler32(b), data.size
        if data.size <= 4 * CHUNK:
            assert O.crc32(data) == zlib.crc32(b) and O.adler32(data) == zlib.adler32(b)
        cut = data.size // 3
        for kind, f in (("gzip", zlib.crc32), ("zlib", zlib.adler32)):
            assert checksum_combine(kind, f(b[:cut]), f(b[cut:]), len(b) - cut) == f(b)


def test_pipelined_rounds_over_rccl_single_rank(compressor):
    """The N > 1 bench path (block-cyclic rounds + gather) on the one GPU we have: a 1-rank
    `nccl` (= RCCL) group exercises the real process group, the real compressor and the
    BFINAL / alignment logic across rounds; the 2- and 3-rank exchange is covered over gloo."""
    import socket

    import torch
    import torch.distributed as dist

    from starflate_amd import multigpu

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        data = synth.gen_text(4 * 5 * CHUNK, seed=31)
        src = torch.from_numpy(data).cuda()
        pieces = list(src.chunk(4))
        bufs = [torch.empty(compressor.compress_bound(p.numel()), dtype=torch.uint8, device="cuda") for p in pieces]

        def compress_fn(piece, final, k):
            return compressor.compress_tensor(piece, out=bufs[k], final_stream=final)

        out, total = multigpu.compress_pipelined(compress_fn, pieces)
        stream = out[:total].cpu().numpy()
        _roundtrip(stream, data)
        want = np.concatenate([O.compress(data[k * 5 * CHUNK:(k + 1) * 5 * CHUNK], _params(final_stream=(k == 3))) for k in range(4)])
        assert np.array_equal(stream, want)
    finally:
        dist.destroy_process_group()


def test_fixture_tool_matches_reference_cli(starfleet, tmp_path):
    """tools/deflate_compress_gpu.py has the switches of the reference's tools/deflate_compress.py (--src, --fixed)
    and writes the same kind of fixture: raw DEFLATE on stdout, here bit-exact with the specification."""
    import os
    import subprocess
    import sys

    from conftest import GOLDEN, ROOT

    src = os.path.join(GOLDEN, "starfleet.html")
    data = np.frombuffer(starfleet, np.uint8)
    for extra, strategy in (([], "auto"), (["--fixed"], "fixed")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "deflate_compress_gpu.py"), "--src", src] + extra,
                             capture_output=True, timeout=300)
        assert out.returncode == 0, out.stderr.decode()[-500:]
        got = np.frombuffer(out.stdout, np.uint8)
        assert np.array_equal(got, O.compress(data, _params(strategy)))
        assert zlib.decompress(out.stdout, -15) == starfleet
    npz = tmp_path / "ix.npz"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "deflate_compress_gpu.py"), "--src", src, "--gzip",
                          "--index", str(npz)], capture_output=True, timeout=300)
    assert out.returncode == 0 and zlib.decompress(out.stdout, 31) == starfleet
    ix = np.load(npz)
    assert int(ix["size"]) == data.size and ix["offsets"].size == 6 and ix["regions"].shape == (5, 32, 2)


def test_async_entry_point_and_independent_contexts():
    """sfh_compress_device_async only enqueues (size left in device memory); two contexts on two streams run
    side by side and give the bytes of the synchronous call."""
    import torch

    from starflate_amd import Compressor

    a, b = Compressor(0), Compressor(0)
    da = torch.from_numpy(synth.gen_text(9 * CHUNK + 5, seed=41)).cuda()
    db = torch.from_numpy(synth.gen_mixed(1 << 20, seed=42, stripe=1 << 15)).cuda()
    want_a, na = a.compress_tensor(da, container="gzip")
    want_a = want_a[:na].clone()
    want_b, nb = b.compress_tensor(db)
    want_b = want_b[:nb].clone()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    oa = torch.zeros(a.compress_bound(da.numel()), dtype=torch.uint8, device="cuda")
    ob = torch.zeros(b.compress_bound(db.numel()), dtype=torch.uint8, device="cuda")
    za = torch.zeros(1, dtype=torch.int64, device="cuda")
    zb = torch.zeros(1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):  # interleaved enqueues, no host synchronisation in between
        a.compress_tensor_async(da, oa, za, stream=sa.cuda_stream, container="gzip")
        b.compress_tensor_async(db, ob, zb, stream=sb.cuda_stream)
    sa.synchronize()
    sb.synchronize()
    assert int(za.item()) == na and torch.equal(oa[:na], want_a)
    assert int(zb.item()) == nb and torch.equal(ob[:nb], want_b)
    assert zlib.decompress(oa[:na].cpu().numpy().tobytes(), 31) == da.cpu().numpy().tobytes()
    a.close()
    b.close()


def test_one_context_on_two_streams_is_ordered(compressor):
    """Calls on ONE context share its device scratch; the library orders a call behind the previous one even when
    they are enqueued on different streams (an event between them), so interleaved enqueues on two streams give
    the bytes of synchronous calls."""
    import torch

    da = torch.from_numpy(synth.gen_text(40 * CHUNK + 5, seed=51)).cuda()
    db = torch.from_numpy(synth.gen_mixed(2 << 20, seed=52, stripe=1 << 15)).cuda()
    want_a, na = compressor.compress_tensor(da)
    want_a = want_a[:na].clone()
    want_b, nb = compressor.compress_tensor(db, container="zlib")
    want_b = want_b[:nb].clone()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    oa = [torch.zeros(compressor.compress_bound(da.numel()), dtype=torch.uint8, device="cuda") for _ in range(3)]
    ob = [torch.zeros(compressor.compress_bound(db.numel()), dtype=torch.uint8, device="cuda") for _ in range(3)]
    za = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(3)]
    zb = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    for k in range(3):  # no host synchronisation in between
        compressor.compress_tensor_async(da, oa[k], za[k], stream=sa.cuda_stream)
        compressor.compress_tensor_async(db, ob[k], zb[k], stream=sb.cuda_stream, container="zlib")
    sa.synchronize()
    sb.synchronize()
    for k in range(3):
        assert int(za[k].item()) == na and torch.equal(oa[k][:na], want_a), k
        assert int(zb[k].item()) == nb and torch.equal(ob[k][:nb], want_b), k


def test_index_read_on_another_stream_is_ordered(compressor):
    """The index and the sub-index are written by the last call's k_scan / k_emit; a copy of them on ANOTHER stream than
    the one that call ran on waits for it (sfh_copy_index / sfh_copy_subindex order themselves behind the call)."""
    import torch

    d = torch.from_numpy(synth.gen_text(64 << 20, seed=53)).cuda()
    out = torch.zeros(compressor.compress_bound(d.numel()), dtype=torch.uint8, device="cuda")
    _, n = compressor.compress_tensor(d, out=out)
    want_idx = compressor.last_index().copy()
    want_sub = compressor.last_subindex().copy()
    small = torch.from_numpy(synth.gen_text(3 * CHUNK, seed=54)).cuda()
    side, z = torch.cuda.Stream(), torch.zeros(1, dtype=torch.int64, device="cuda")
    for _ in range(3):
        compressor.compress_tensor(small)  # other offsets in the scratch
        torch.cuda.synchronize()
        compressor.compress_tensor_async(d, out, z, stream=side.cuda_stream)  # 64 MiB: still running when the copies are issued
        idx = compressor.last_index(device="cuda")  # torch's current stream, not `side`
        sub = compressor.last_subindex(device="cuda")
        torch.cuda.synchronize()
        assert np.array_equal(idx.cpu().numpy().astype(np.uint64), want_idx)
        assert np.array_equal(sub.cpu().numpy().view(np.uint32).reshape(want_sub.shape), want_sub)
        compressor.compress_tensor(small)
        torch.cuda.synchronize()
        compressor.compress_tensor_async(d, out, z, stream=side.cuda_stream)
        offs = compressor.debug(_capi.DBG_OFFSETS, want_idx.size - 1)  # the debug read waits for the call, too
        assert np.array_equal(offs, want_idx[:-1])


def test_stage_ms_covers_every_batch(monkeypatch):
    """sfh_last_stage_ms sums the per-kernel events of ALL batches of the call (a host-buffer call runs 64 MiB batches,
    a device call beyond 1 GiB several): the sum of the stages of a 3-batch device call is its device time."""
    import torch
    from starflate_amd import Compressor

    monkeypatch.setenv("SFH_BATCH_CHUNKS", "1024")  # 32 MiB batches
    c = Compressor(0)
    c.set_profiling(True)
    d = torch.from_numpy(synth.gen_text(96 << 20, seed=55)).cuda()
    out = torch.zeros(c.compress_bound(d.numel()), dtype=torch.uint8, device="cuda")
    c.compress_tensor(d, out=out)
    one = torch.from_numpy(synth.gen_text(32 << 20, seed=55)).cuda()
    c.compress_tensor(one, out=out, block_bytes=c.last_block_bytes())
    ms_one = sum(c.stage_ms().values())
    # structural, not wall-clock (event jitter, clock ramps and co-tenants must not fail the suite): every stage of the
    # three-batch call is positive, and the median of several calls is well above ONE batch's time -- all three counted
    totals = []
    for _ in range(5):
        c.compress_tensor(d, out=out)
        ms = c.stage_ms()
        assert all(ms[k] > 0 for k in ("k_lz77", "k_plan", "k_scan", "k_emit")), ms
        totals.append(sum(ms.values()))
    total = sorted(totals)[2]
    assert total > 1.8 * ms_one, (totals, ms_one)  # three batches, not the first one only (3.0 x expected)
    # the host-buffer entry point (64 MiB batches at most; 32 MiB here): the same kernels, every batch counted (they run
    # beside the copies of their neighbours and share HBM with them, so they take longer than on resident input)
    host = d.cpu().numpy()
    c.compress(host)
    hs = sum(c.stage_ms().values())
    assert hs > 1.8 * ms_one, (hs, ms_one)  # every batch of the host-buffer call, too
    c.close()


def test_fuzz_bit_exact_vs_oracle(compressor):
    """Seeded fuzz: 300 inputs stitched from generators with very different match structure (runs, short and long
    periods, text, noise, low-entropy noise, counters, sparse bytes), sizes 0..160 KiB with ragged chunk tails."""
    rng = np.random.default_rng(20260101)
    text = synth.gen_text(400_000, seed=77)

    def piece(n):
        kind = int(rng.integers(0, 9))
        if kind == 0:
            return np.full(n, int(rng.integers(0, 256)), np.uint8)
        if kind == 1:
            p = int(rng.integers(1, 40))
            return np.tile(rng.integers(0, 256, p, dtype=np.uint8), n // p + 1)[:n]
        if kind == 2:
            p = int(rng.integers(200, 5000))
            return np.tile(rng.integers(0, 256, p, dtype=np.uint8), n // p + 1)[:n]
        if kind == 3:
            o = int(rng.integers(0, text.size - n)) if n < text.size else 0
            return text[o:o + n]
        if kind == 4:
            return rng.integers(0, 256, n, dtype=np.uint8)
        if kind == 5:
            return rng.integers(0, int(rng.integers(2, 9)), n, dtype=np.uint8)
        if kind == 6:
            return (np.arange(n, dtype=np.uint32) // int(rng.integers(1, 5))).astype(np.uint32).view(np.uint8)[:n]
        if kind == 7:
            a = np.zeros(n, np.uint8)
            idx = rng.integers(0, max(n, 1), max(n // 50, 1))
            a[idx % max(n, 1)] = rng.integers(1, 256, idx.size, dtype=np.uint8)
            return a
        o = int(rng.integers(0, text.size - n)) if n < text.size else 0
        t = text[o:o + n].copy()  # text with point mutations: long matches that break
        t[rng.integers(0, max(n, 1), max(n // 97, 1)) % max(n, 1)] ^= 1
        return t

    import os

    for it in range(int(os.environ.get("SF_FUZZ_N", "300"))):  # SF_FUZZ_N=5000 for a soak run
        total = int(rng.choice([0, 1, 2, 3, 5, 100, 511, 512, 513, 1023, 1024, 1025, 8191, 8192, 8193, CHUNK - 1, CHUNK, CHUNK + 1,
                                int(rng.integers(0, 5 * CHUNK)), int(rng.integers(0, 9 * CHUNK))]))
        parts, left = [], total
        while left > 0:
            n = int(min(left, rng.integers(1, 20000)))
            parts.append(np.ascontiguousarray(piece(n)[:n]))
            left -= n
        data = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
        assert data.size == total
        strategy = ["auto", "auto", "dynamic", "fixed"][it % 4]
        lazy = [3, 0, 1, 2, 3][it % 5]
        fast = it % 7 != 0
        bb = [0, 32768, 65536, 131072][it % 3 if it % 11 else 3]
        # every effort: default (even positions searched), thorough (all, steps of 512), fast (one level), fastest (no near)
        effort, ekw = [("default", {}), ("thorough", dict(stride2=0, step=512)), ("best", dict(chain_depth=8)), ("fast", dict(depth=1)),
                       ("fastest", dict(depth=1, use_near=0)), ("max", dict(stride2=0, step=512, hash_bits=12, long_hash_bytes=7)),
                       ("ultra", dict(chain_depth=16)), ("extreme", dict(chain_depth=32)),
                       ("recent", dict(recent=1, near_depth=1, link_steps=1)),
                       ("recent_all", dict(recent=1, near_depth=1, link_steps=1, stride2=0, step=512))][it % 10 if it % 13 else 5]
        got = np.frombuffer(compressor.compress(data, strategy=strategy, lazy=lazy, stored_fast_path=fast, block_bytes=bb, effort=effort), np.uint8)
        want = O.compress(data, O.default_params(strategy=_capi.STRATEGY[strategy], lazy=lazy, fast_skip=int(fast), strip_bytes=bb, **ekw))
        assert np.array_equal(got, want), (it, total, strategy, lazy, fast, bb, effort, np.flatnonzero(got[:min(got.size, want.size)] != want[:min(got.size, want.size)])[:3])
        if it % 10 == 0:
            _roundtrip(got, data)
            idx, sub = compressor.last_index(), compressor.last_subindex()
            back, st = compressor.decompress(got, idx, data.size, subindex=sub, block_bytes=compressor.last_block_bytes())
            assert st == 0 and back == data.tobytes(), it


def test_fuzz_stored_fast_path_switching(compressor):
    """Seeded fuzz of the stored fast path's switching (round 6: a skipped chunk's later rounds in one go, the window reloaded
    from the input behind it, bank-private byte counts; a chunk stored by its probe): inputs stitched from CHUNK-SCALE pieces -- noise
    (stored), bytes a shade off uniform (either side of the probe's threshold), six-bit
    noise (the fast path taken, the chunk NOT stored: k_emit takes the items from the input), text, zeros, text right behind
    noise with matches into the noise's probe span -- so that the path switches on and off at every offset of a strip, with
    ragged tails; every effort family, the switch on and off, strips of 1..8 chunks.  Bit-exact against the specification,
    round trips through the oracle decoder and the GPU decoder.  SF_FUZZ_N scales it (default 60)."""
    import os

    rng = np.random.default_rng(20261005)
    text = synth.gen_text(600_000, seed=91)

    def piece(n):
        kind = int(rng.integers(0, 9))
        if kind in (0, 1):
            return rng.integers(0, 256, n, dtype=np.uint8)
        if kind in (7, 8):  # all but uniform: 7.8 .. 7.999 bits per byte, around the threshold of "stored by the probe"
            pw = np.ones(256)
            pw[rng.permutation(256)[: int(rng.integers(1, 128))]] = rng.uniform(0.05, 0.95)  # (the rule fires for a good half of these)
            return rng.choice(256, size=n, p=pw / pw.sum()).astype(np.uint8)
        if kind == 2:
            return rng.integers(0, 64, n, dtype=np.uint8)  # high entropy for the probe, compressible for the plan
        if kind == 3:
            return np.zeros(n, np.uint8)
        if kind == 4:  # noise whose head repeats a little later: matches that reach back into a skipped chunk's probe span
            a = rng.integers(0, 256, n, dtype=np.uint8)
            k = int(min(n // 3, rng.integers(100, 3000)))
            if k:
                a[n - k:] = a[:k]
            return a
        o = int(rng.integers(0, text.size - n)) if n < text.size else 0
        return text[o:o + n]

    efforts = [("default", {}), ("thorough", dict(stride2=0, step=512)), ("recent_all", dict(recent=1, near_depth=1, link_steps=1, stride2=0, step=512)),
               ("best", dict(chain_depth=8)), ("fastest", dict(depth=1, use_near=0)), ("max", dict(stride2=0, step=512, hash_bits=12, long_hash_bytes=7))]
    for it in range(int(os.environ.get("SF_FUZZ_N", "60"))):
        parts = []
        for _ in range(int(rng.integers(1, 9))):
            n = int(rng.choice([CHUNK // 4, CHUNK // 2, CHUNK, CHUNK, 2 * CHUNK, 3 * CHUNK, int(rng.integers(1, 3 * CHUNK))]))
            parts.append(np.ascontiguousarray(piece(n)[:n]))
        data = np.concatenate(parts)
        if it % 3 == 0:
            data = data[: data.size - int(rng.integers(0, min(data.size, CHUNK)))]  # a ragged tail inside the last chunk
        fast = it % 5 != 4
        bb = [0, CHUNK, 2 * CHUNK, 4 * CHUNK, 8 * CHUNK][it % 5]
        lazy = [3, 3, 0, 2][it % 4]
        effort, ekw = efforts[it % len(efforts)]
        got = np.frombuffer(compressor.compress(data, lazy=lazy, stored_fast_path=fast, block_bytes=bb, effort=effort), np.uint8)
        want = O.compress(data, O.default_params(lazy=lazy, fast_skip=int(fast), strip_bytes=bb, **ekw))
        assert np.array_equal(got, want), (it, data.size, lazy, fast, bb, effort, np.flatnonzero(got[:min(got.size, want.size)] != want[:min(got.size, want.size)])[:3])
        if it % 4 == 0:
            _roundtrip(got, data)
            idx, sub = compressor.last_index(), compressor.last_subindex()
            back, st = compressor.decompress(got, idx, data.size, subindex=sub, block_bytes=compressor.last_block_bytes())
            assert st == 0 and back == data.tobytes(), it


def test_repeated_calls_do_not_leak_or_drift():
    """200 calls of varying size on one context (growing and shrinking inputs, all entry points): device memory in
    use by the library stays bounded by its largest call, results stay identical."""
    import torch

    from starflate_amd import Compressor

    c = Compressor(0)
    rng = np.random.default_rng(77)
    text = synth.gen_text(8 << 20, seed=5)
    ref = {}
    sizes = [int(x) for x in rng.integers(0, 8 << 20, 40)] + [8 << 20, 0, 1, CHUNK]
    torch.cuda.synchronize()
    big = torch.from_numpy(text).cuda()
    c.compress_tensor(big)  # the largest call sizes the workspace
    c.compress(text)        # ... and the host staging buffers
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for it in range(200):
        n = sizes[it % len(sizes)]
        if it % 3 == 0:
            got = c.compress(text[:n], container=["raw", "zlib", "gzip"][it % 9 // 3])
            key = (n, it % 9 // 3, "h")
        else:
            out, nb = c.compress_tensor(big[:n].clone() if n else torch.empty(0, dtype=torch.uint8, device="cuda"))
            got = out[:nb].cpu().numpy().tobytes()
            key = (n, 0, "d")
            if it % 5 == 0:
                idx, sub = c.last_index(device="cuda"), c.last_subindex(device="cuda")
                back, st = c.decompress_tensor(out[:nb].clone(), idx, n, subindex=sub, block_bytes=c.last_block_bytes())
                assert st == 0 and torch.equal(back, big[:n])
        assert ref.setdefault(key, got) == got
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (64 << 20), (free0, free1)  # nothing accumulates (torch's own cache aside)
    c.close()


def test_compress_multi_is_bit_identical_to_one_call(compressor, starfleet):
    """sfh_compress_multi (one process, N contexts, one host thread each): contiguous shards, one stream, exactly the
    bytes of a single call -- raw and wrapped, more contexts than chunks, empty input."""
    from starflate_amd import Compressor, compress_multi

    ctxs = [Compressor(0) for _ in range(3)]
    text = synth.gen_text(11 * CHUNK + 4321, seed=61)
    cases = {"text": text, "html": np.frombuffer(starfleet, np.uint8), "one_chunk": text[:1000], "empty": np.zeros(0, np.uint8),
             "mixed": synth.gen_mixed(1 << 20, seed=62, stripe=1 << 15)}
    for name, data in cases.items():
        for container in ("raw", "zlib", "gzip"):
            want = compressor.compress(data, container=container)
            for k in (1, 2, 3):
                got = compress_multi(ctxs[:k], data, container=container)
                assert got == want, (name, container, k)
        want = compressor.compress(data, strategy="fixed", final_stream=False)
        assert compress_multi(ctxs, data, strategy="fixed", final_stream=False) == want, name
    with pytest.raises(Exception):
        compress_multi([ctxs[0], ctxs[0]], text)  # the same context twice
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("effort", ["recent", "recent_all"])
def test_effort_recent_bit_exact_vs_oracle(compressor, starfleet, effort):
    """SFH_EFFORT_RECENT / _RECENT_ALL: the step tables with exact recency (the specification's `recent`: buckets {lo, hi}
    filled in position order by ds_mskor_rtn_b32, the exact predecessor as the near candidate) -- bit-exact over inputs
    whose buckets look very different (text, the reference's HTML file, mixed stripes, one value, short periods, noise,
    real source text and machine code), strips from one chunk to 1 MiB, ragged tails, every strategy and lazy level, with
    and without the stored fast path, and through the GPU decoder."""
    from starflate_amd import realbytes

    rng = np.random.default_rng(5)
    text = synth.gen_text(44 * CHUNK + 77, seed=14)
    cases = {"text": text, "starfleet": np.frombuffer(starfleet, np.uint8), "mixed": synth.gen_mixed(1 << 20, seed=4, stripe=1 << 15),
             "zeros": np.zeros(5 * CHUNK + 333, np.uint8), "period7": np.tile(np.arange(7, dtype=np.uint8), 30000),
             "period300": np.tile(rng.integers(0, 256, 300, dtype=np.uint8), 700), "random": rng.integers(0, 256, 3 * CHUNK + 5, dtype=np.uint8),
             "low_entropy": rng.integers(0, 3, 2 * CHUNK + 99, dtype=np.uint8), "tiny": np.frombuffer(b"abcabcabcabcabcabc", np.uint8),
             "empty": np.zeros(0, np.uint8), "three": np.frombuffer(b"xyz", np.uint8),
             "noise_then_text": np.concatenate([rng.integers(0, 256, 3 * CHUNK, dtype=np.uint8), text[: 2 * CHUNK + 9]])}
    src, binb = realbytes.source(3 << 20), realbytes.binary(18 << 20)
    if src.size >= (2 << 20):
        cases["source"] = src[: 2 << 20]
    if binb.size >= (18 << 20):
        cases["binary"] = binb[16 << 20: (16 << 20) + (1 << 20)]
    for name, data in cases.items():
        for bb in (0, 32768, 131072, 1 << 20):
            got = np.frombuffer(compressor.compress(data, effort=effort, block_bytes=bb), np.uint8)
            want = O.compress(data, O.default_params(strip_bytes=bb, **EFFORT_PARAMS[effort]))
            assert got.size == want.size and np.array_equal(got, want), (name, bb, np.flatnonzero(got[:min(got.size, want.size)] != want[:min(got.size, want.size)])[:3])
            _roundtrip(got, data)
        if name in ("text", "starfleet", "mixed", "source", "binary"):
            assert len(compressor.compress(data, effort=effort)) < len(compressor.compress(data)), name
    for strategy in ("fixed", "dynamic", "stored"):
        for lazy in (0, 1, 3):
            got = np.frombuffer(compressor.compress(text, effort=effort, strategy=strategy, lazy=lazy, stored_fast_path=False), np.uint8)
            want = O.compress(text, O.default_params(strategy=_capi.STRATEGY[strategy], lazy=lazy, fast_skip=0, **EFFORT_PARAMS[effort]))
            assert np.array_equal(got, want), (strategy, lazy)
    got = compressor.compress(text, effort=effort)
    back, st = compressor.decompress(got, compressor.last_index(), text.size, subindex=compressor.last_subindex(),
                                     block_bytes=compressor.last_block_bytes())
    assert st == 0 and back == text.tobytes()
    o = _capi.make_options(effort=effort)
    o.chain_depth = 4  # chain_depth belongs to the chain efforts
    import ctypes as C
    n_out = C.c_size_t(0)
    buf = np.zeros(compressor.compress_bound(text.size), np.uint8)
    assert compressor._lib.sfh_compress(compressor._h, text.ctypes.data, text.size, buf.ctypes.data, buf.size, C.byref(n_out), C.byref(o)) == -1


def test_lds_atomics_execute_lanes_in_ascending_order(compressor):
    """The chain efforts insert 64 positions with one ds_wrxchg_rtn_b32, SFH_EFFORT_RECENT with one ds_mskor_rtn_b32, and both
    rely on the LDS executing the lanes of that instruction in ascending lane order where they meet at one address (each
    lane gets the nearest lower lane with its hash), and one wave's instructions in issue order.  That is measured behaviour
    of gfx950, not an architectural promise.  sfh_lds_order_check runs the kernel the library itself runs before the first
    such call -- the match kernel's own pattern: partial exec masks, sixteen back-to-back instructions by one wave on shared
    buckets, the other waves reading the table (and, op 1, storing into the buckets' other halves) meanwhile -- here at a
    few hundred million positions on the device the suite runs on: a part or a driver that orders differently fails HERE."""
    for op in (0, 1):
        bad, n = compressor.lds_order_check(op, blocks=1024, iters=40)
        assert bad == 0 and n > 150_000_000, (op, bad, n)


def test_order_guard_refuses_the_efforts_that_need_it(monkeypatch):
    """SFH_FORCE_ORDER_FAIL=1 makes the library's own check (run once per context, in sfh_create) report failure: those efforts then return SFH_E_UNSUPPORTED with a message that says why, every
    other effort works as ever, and nothing is written."""
    from starflate_amd import Compressor
    from starflate_amd.compressor import StarflateError

    text = synth.gen_text(3 * CHUNK, seed=3)
    with monkeypatch.context() as m:
        m.setenv("SFH_FORCE_ORDER_FAIL", "1")
        c = Compressor(0)
    try:
        for effort in ("best", "ultra", "extreme", "chain3", "recent", "recent_all"):
            with pytest.raises(StarflateError) as e:
                c.compress(text, effort=effort)
            assert e.value.code == -7 and "ascending order" in str(e.value), (effort, str(e.value))
        for effort in ("default", "fast", "thorough", "max"):
            got = np.frombuffer(c.compress(text, effort=effort), np.uint8)
            assert np.array_equal(got, O.compress(text, O.default_params(**EFFORT_PARAMS[effort])))
        assert c.lds_order_check(0, 8, 2)[0] == 0  # (the exported check itself reports what the device does)
    finally:
        c.close()
