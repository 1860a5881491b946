import zlib

import numpy as np

from starflate_amd import synth


def test_generators_are_deterministic_and_shaped():
    a, b = synth.gen_text(300_000, seed=3), synth.gen_text(300_000, seed=3)
    assert np.array_equal(a, b) and a.dtype == np.uint8 and a.size == 300_000
    assert not np.array_equal(a, synth.gen_text(300_000, seed=4))
    r = synth.gen_random(100_000, seed=5)
    assert len(zlib.compress(r.tobytes(), 6)) > 0.99 * r.size
    m = synth.gen_mixed(1 << 20, seed=4, stripe=1 << 16)
    assert m.size == 1 << 20


def test_text_is_enwik_like():
    t = synth.gen_text(4 << 20, seed=3).tobytes()
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    ratio = len(t) / (len(co.compress(t)) + len(co.flush()))
    assert 2.6 < ratio < 3.4, ratio  # zlib-6 ratio target 2.7-3.3
