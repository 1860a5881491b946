"""N>1 path on CPU: world_size-2 `gloo` processes run the same shard/concat code the GPU ranks
run over RCCL (starflate_amd/multigpu.py); the per-rank streams are made by the oracle encoder
(non-final for rank 0, final for rank 1) and the concatenation must decode to the whole input."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_lib as O
    from starflate_amd import multigpu, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = synth.gen_text(n, seed=8)
    lo, hi = multigpu.shard_bounds(n, world)[rank]
    stream = O.compress(data[lo:hi], O.default_params(final_stream=int(rank == world - 1)))
    local = torch.zeros(stream.size + 100, dtype=torch.uint8)
    local[: stream.size] = torch.from_numpy(stream)
    out, total = multigpu.concat_streams(local, stream.size)
    if rank == 0:
        st, w, back = O.decompress(out[:total].numpy(), n)
        q.put((st, w, bool(np.array_equal(back, data)), total))
    dist.barrier()
    dist.destroy_process_group()


def _worker_pipelined(rank, world, port, piece_bytes, K, q, container="raw"):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_lib as O
    from starflate_amd import multigpu, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    whole = synth.gen_text(piece_bytes * K * world, seed=12)
    # global piece g = k*world + rank
    pieces = [torch.from_numpy(whole[(k * world + rank) * piece_bytes:(k * world + rank + 1) * piece_bytes].copy()) for k in range(K)]

    def compress_fn(piece, final, k):
        s = O.compress(piece.numpy(), O.default_params(final_stream=int(final)))
        buf = torch.zeros(s.size + 64, dtype=torch.uint8)
        buf[: s.size] = torch.from_numpy(s)
        return buf, s.size

    if container != "raw":
        import zlib

        f = zlib.crc32 if container == "gzip" else zlib.adler32
        prealloc = torch.zeros(whole.size * 2 + 64, dtype=torch.uint8) if rank == 0 and K > 1 else None
        out, total = multigpu.compress_pipelined(compress_fn, pieces, container=container, out=prealloc,
                                                 checksum_fn=lambda piece, k: f(piece.numpy().tobytes()))
        if rank == 0:  # zlib's wrapper-checking inflate is the judge of header, checksum and ISIZE
            back = zlib.decompress(out[:total].numpy().tobytes(), 31 if container == "gzip" else 15)
            h, t = (10, 8) if container == "gzip" else (2, 4)
            st, w, body = O.decompress(out[h:total - t].numpy(), whole.size)
            q.put((st, w, back == whole.tobytes() and bool(np.array_equal(body, whole)), total))
        dist.barrier()
        dist.destroy_process_group()
        return
    timing = {}
    out, total = multigpu.compress_pipelined(compress_fn, pieces, timing=timing)
    # every rank gets one gather time per round (what bench.py reports as multi_gpu.gather_ms_per_round)
    assert len(timing["gather_ms"]) == K and all(t >= 0 for t in timing["gather_ms"]), timing
    if rank == 0:
        st, w, back = O.decompress(out[:total].numpy(), whole.size)
        q.put((st, w, bool(np.array_equal(back, whole)), total))
    dist.barrier()
    dist.destroy_process_group()


def _worker_bench_logic(rank, world, port, piece_bytes, K, q):
    """bench.py's own N > 1 step and verification (run_steps, pipelined_step, verify_pieces, verify_concatenation)
    over gloo, the compressor replaced by a stub that returns the oracle's stream -- what the driver's
    `bench.py --gpus N` does, minus the GPU."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    import oracle_lib as O
    from starflate_amd import synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = torch.from_numpy(synth.gen_text(piece_bytes * K, seed=40 + rank))  # this rank's shard, as bench.py makes it
    pieces = list(data.chunk(K))
    bound = O.lib().sfo_compress_bound(piece_bytes, O.default_params())
    scratch = [torch.zeros(bound, dtype=torch.uint8) for _ in range(K)]
    gathered = torch.zeros(bound * K * world + 64, dtype=torch.uint8) if rank == 0 else None
    sizes, result = [0] * K, {}

    def compress_fn(piece, final, k):
        s = O.compress(piece.numpy(), O.default_params(final_stream=int(final), strip_bytes=65536))
        scratch[k][: s.size] = torch.from_numpy(s)
        sizes[k] = s.size
        return scratch[k], torch.tensor([s.size], dtype=torch.int64)  # the enqueue-only form: size as a tensor

    def step():  # as in bench.py: the first step has its arguments judged, the repeats skip that
        result["out"], result["total"] = bench.pipelined_step(compress_fn, pieces, gathered,
                                                              validate=not result.get("validated", False))
        result["validated"] = True

    dt = bench.run_steps(step, dist.barrier, 2, 1)
    ok, crcs = bench.verify_pieces(pieces, scratch, sizes, -15)
    crc = torch.tensor(crcs, dtype=torch.int64)
    allc = [torch.zeros(K, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allc, crc)
    if rank == 0:
        whole_ok = bench.verify_concatenation(result["out"][: result["total"]].numpy().tobytes(), -15, piece_bytes * K * world,
                                              piece_bytes, [c.tolist() for c in allc])
        q.put((ok, whole_ok, dt > 0, int(result["total"])))
    dist.barrier()
    dist.destroy_process_group()


def _worker_bad_args(rank, world, port, q):
    """A check that fails on one rank only (rank 0's `out` is too small) raises on EVERY rank, before any
    stream moves -- nobody is left waiting in a collective."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from starflate_amd import multigpu

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pieces = [torch.zeros(32768, dtype=torch.uint8)]
    raised = False
    try:
        multigpu.compress_pipelined(lambda p, f, k: (torch.zeros(10, dtype=torch.uint8), 5), pieces,
                                    out=torch.zeros(16, dtype=torch.uint8) if rank == 0 else None, bound_fn=lambda n: n + 100)
    except ValueError:
        raised = True
    try:
        multigpu.compress_pipelined(lambda p, f, k: (torch.zeros(10, dtype=torch.uint8), 5), pieces, container="gzip")
    except ValueError:
        raised = raised and True
    else:
        raised = False
    q.put((rank, raised))
    dist.barrier()
    dist.destroy_process_group()


def _worker_mismatched_rounds(rank, world, port, q):
    """Ranks that pass different numbers of pieces (also: none at all) all get the documented ValueError; nobody
    aborts inside a collective of mismatched size."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from starflate_amd import multigpu

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    got = []
    for ks in ((1, 2), (0, 2), (3, 1)):
        pieces = [torch.zeros(32768, dtype=torch.uint8) for _ in range(ks[rank])]
        try:
            multigpu.compress_pipelined(lambda p, f, k: (torch.zeros(10, dtype=torch.uint8), 5), pieces, bound_fn=lambda n: n + 100)
            got.append("no error")
        except ValueError as e:
            got.append("rounds" in str(e))
    q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


def test_mismatched_rounds_raise_on_every_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mismatched_rounds, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got == [(0, [True, True, True]), (1, [True, True, True])]


def test_bench_step_and_verify_logic_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world, piece, K = 2, 2 * 65536, 2
    procs = [ctx.Process(target=_worker_bench_logic, args=(r, world, port, piece, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok, whole_ok, timed, total = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok and whole_ok and timed and total > 0


def test_argument_errors_raise_on_every_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bad_args, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got == [(0, True), (1, True)]


@pytest.mark.parametrize("world,K,container", [(2, 3, "gzip"), (2, 1, "zlib"), (3, 2, "zlib")])
def test_pipelined_wrapped_gloo(world, K, container):
    """Shard checksums travel with the sizes and are combined in global piece order on every rank;
    rank 0 wraps the concatenation once (SURVEY.md 8(f)1 on the N > 1 path)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    piece = 2 * 32768
    procs = [ctx.Process(target=_worker_pipelined, args=(r, world, port, piece, K, q, container)) for r in range(world)]
    for p in procs:
        p.start()
    st, w, same, total = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert st == 0 and w == piece * K * world and same and total > 0


@pytest.mark.parametrize("world,piece,K", [(2, 2 * 32768, 3), (3, 32768 + 32768, 2), (2, 32768, 1)])
def test_pipelined_block_cyclic_gloo(world, piece, K):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipelined, args=(r, world, port, piece, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    st, w, same, total = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert st == 0 and w == piece * K * world and same and total > 0


@pytest.mark.parametrize("world,n", [(2, 5 * 32768 + 777), (2, 1000), (3, 7 * 32768)])
def test_shard_concat_gloo(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    st, w, same, total = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert st == 0 and w == n and same and total > 0


def test_shard_bounds():
    from starflate_amd.multigpu import shard_bounds

    for n in (0, 1, 32768, 32769, 10 * 32768 + 5, 1 << 30):
        for world in (1, 2, 3, 8):
            b = shard_bounds(n, world)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(x[1] == y[0] for x, y in zip(b, b[1:]))
            assert all(lo % 32768 == 0 for lo, _ in b if lo < n)
