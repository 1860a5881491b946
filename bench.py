#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X DEFLATE compressor.

Workload (BASELINE.json configs[2]): 1 GiB synthetic enwik-like text per GPU, compressed in strips of
sfh_options.block_bytes (default at this size 512 KiB: a 32 KiB window sliding over sixteen 32 KiB DEFLATE blocks), input
resident in HBM when the timed region starts.  A "step" = one pass of the whole hot path (k_lz77 -> k_plan ->
k_scan -> k_emit) over that input; with N > 1 every rank compresses its own 1 GiB (weak scaling) and the
byte-aligned streams are concatenated on rank 0 over RCCL.  For N > 1 the sharding is block-cyclic in --rounds
rounds (global piece g = k*N + rank), so the gather of round k (sizes all_gather + one point-to-point send per
rank, straight to the final offset) overlaps the compression of round k+1 (starflate_amd/multigpu.py).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bytes B] [--workload text|random|mixed]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (k_lz77), its launch time measured live
with HIP events on the launch stream; `cpu_baseline` times the oracle's restatement of the reference
decompress() on this host; `e2e` is the same call from pinned host buffers (H2D + kernels + D2H); `workloads`
carries the other two BASELINE workloads at a smaller size (N = 1 only).
"""
import argparse
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SEG = 32768


def make_input(workload, n, rank, dev):
    """The bench bytes: generated on the device (text, random) or on the host (mixed)."""
    import torch

    from starflate_amd import synth

    if workload == "text":
        return synth.gen_text_torch(n, seed=3 + 17 * rank, device=dev), f"{n / 2**30:g} GiB synthetic enwik-like text per GPU (gen_text_torch seed 3)"
    if workload == "random":
        g = torch.Generator(device=dev)
        g.manual_seed(5 + rank)
        return (torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=g),
                f"{n / 2**30:g} GiB high-entropy bytes per GPU (stored-block path)")
    if workload == "runs":
        # degenerate but common (sparse files, zero pages, repeated records): half zeros, half one 61-byte line repeated
        g = torch.Generator(device=dev)
        g.manual_seed(9 + rank)
        line = torch.randint(32, 127, (61,), dtype=torch.uint8, device=dev, generator=g)
        d = torch.zeros(n, dtype=torch.uint8, device=dev)
        d[n // 2:] = line.repeat((n - n // 2) // 61 + 1)[: n - n // 2]
        return d, f"{n / 2**30:g} GiB per GPU: half zeros, half one 61-byte line repeated"
    return torch.from_numpy(synth.gen_mixed(n, seed=4 + rank)).to(dev), f"{n / 2**30:g} GiB mixed Silesia-like stripes per GPU"


CORPUS_NAMES = ("enwik9", "enwik8", "dickens", "alice29.txt")


def find_corpus(workload):
    """SURVEY.md 8(d)(1) / BASELINE.md: a real corpus file under $STARFLATE_CORPUS_DIR replaces the text generator."""
    d = os.environ.get("STARFLATE_CORPUS_DIR")
    if not d or workload != "text":
        return None
    for name in CORPUS_NAMES:
        p = os.path.join(d, name)
        if os.path.isfile(p):
            return p
    return None


def file_input(path, n, rank, dev):
    """n bytes of the file for this rank: rank r starts at r*n (mod the file size); a short file is tiled."""
    import numpy as np
    import torch

    size = os.path.getsize(path)
    if size == 0:
        raise SystemExit(f"{path} is empty")
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    start = (rank * n) % size
    parts, left = [], n
    while left:
        take = min(left, size - start)
        parts.append(np.asarray(mm[start:start + take]))
        left -= take
        start = 0
    buf = parts[0] if len(parts) == 1 else np.concatenate(parts)
    return torch.from_numpy(np.ascontiguousarray(buf)).to(dev), f"{n / 2**30:g} GiB per GPU from {os.path.basename(path)} ({size} bytes{', tiled' if size < n else ''})"


def run_steps(step, fence, steps, warmup):
    """W untimed warm-up steps, then exactly K steps between two fences (barrier + device synchronise)."""
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    return time.perf_counter() - t0


def pipelined_step(compress_fn, pieces, gathered, container="raw", checksum_fn=None, group=None, validate=True, bound_fn=None, timing=None):
    """One N > 1 step: block-cyclic rounds with the gather overlapped (what the driver's --gpus N runs).  validate: the
    argument checks of compress_pipelined (two small collectives); a repeated step passes False after the first."""
    from starflate_amd import multigpu

    return multigpu.compress_pipelined(compress_fn, pieces, out=gathered, container=container, checksum_fn=checksum_fn, group=group,
                                       validate=validate, bound_fn=bound_fn, timing=timing)


def verify_pieces(pieces, streams, sizes, wbits):
    """Every rank inflates its own piece streams with zlib (a non-final piece is still inflatable)."""
    ok, crcs = True, []
    for k, piece in enumerate(pieces):
        host_piece = piece.cpu().numpy().tobytes()
        stream = streams[k][: sizes[k]].cpu().numpy().tobytes()
        ok = ok and zlib.decompressobj(wbits).decompress(stream) == host_piece
        crcs.append(zlib.crc32(host_piece))
    return ok, crcs


def verify_concatenation(whole_stream, wbits, total_in, piece_bytes, crcs_by_rank):
    """Rank 0: the concatenation is ONE valid stream of all pieces in the order g = k*N + rank."""
    whole = zlib.decompress(whole_stream, wbits)
    world, K = len(crcs_by_rank), len(crcs_by_rank[0])
    return len(whole) == total_in and all(
        zlib.crc32(whole[(k * world + r) * piece_bytes:(k * world + r + 1) * piece_bytes]) == int(crcs_by_rank[r][k])
        for k in range(K) for r in range(world))


XGMI_LINK_GBS = 77.0      # one xGMI link, one direction (MI355X_MICROARCH.md: ~153 GB/s both ways per pair of GPUs)
XGMI_P2P_EFFICIENCY = 0.82  # what a point-to-point RCCL transfer reaches of that


def predict_scaling(world, rounds, compute_ms, stream_bytes_per_rank):
    """DESIGN.md section 4 in numbers, so that a SCALE record can be read against the model without the document: every
    rank but the root sends its stream over ITS OWN link, the links run side by side, so the gather of a step takes
    T_x = (a rank's stream bytes) / (link rate) whatever N >= 2 is; with K block-cyclic rounds only the last round's
    transfer is exposed: step ~ max(T_c, T_x) + T_x / K, where T_c is this rank's kernel time for the step as measured."""
    t_x = 0.0 if world < 2 else stream_bytes_per_rank / (XGMI_LINK_GBS * XGMI_P2P_EFFICIENCY * 1e9) * 1e3
    step = max(compute_ms, t_x) + (t_x / max(rounds, 1) if world > 1 else 0.0)
    return {"model": "step_ms = max(T_c, T_x) + T_x / rounds;  T_x = stream bytes per rank / (77 GB/s x 0.82), the same for any N >= 2 "
                     "(one xGMI link per peer into the root, side by side)",
            "T_c_ms": round(compute_ms, 3), "T_x_ms": round(t_x, 3), "predicted_step_ms": round(step, 3),
            "predicted_efficiency": round(compute_ms / step, 3) if step > 0 else None,
            "bound": "gather (xGMI into the root)" if t_x > compute_ms else "compression"}


def zlib6_size(host_bytes):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    return len(co.compress(host_bytes)) + len(co.flush())


def traffic_from_profile(kernel, n, now):
    """HBM bytes per launch of `kernel` (FETCH_SIZE x 2 + WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes) from the
    committed rocprofv3 --pmc passes of this very command (tools/prof_round.sh -> profiles/pmc_traffic.json), scaled to this
    run's bytes.  The profile carries the commit and a SHA-256 of the kernel sources it measured; `current` says whether
    those are the sources that are running now (a number from other sources is still reported, flagged)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        e, meta = d.get(kernel), d.get("_meta", {})
        if not e:
            return None, None
        scale = n / float(meta.get("bytes_per_launch", 1 << 30))
        info = {"source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; not collected in this run)",
                "commit": meta.get("commit", "unknown"), "csrc_sha256": meta.get("csrc_sha256"),
                "running": now, "current": meta.get("csrc_sha256") == now["csrc_sha256"],
                "read_bytes": int(e.get("read_bytes", 0) * scale), "write_bytes": int(e.get("write_bytes", 0) * scale)}
        return int(e["hbm_bytes_per_launch"] * scale), info
    except Exception:  # noqa: BLE001
        return None, None


def issue_from_profile(kernel, kernel_ms, n, props, now):
    """The dominant kernel is nowhere near the HBM roofline; what it spends instead: vector
    instructions per launch from the committed PMC profile (SQ_INSTS_VALU of the same workload -- NOT counted in this run) over
    this run's kernel time, as cycles per wave-instruction per SIMD, and the share of the CU's cycles its LDS is busy.
    Dropped (None) when the profile was taken from other kernel sources than the ones running."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            meta = json.load(f).get("_meta", {})
        if meta.get("csrc_sha256") != now["csrc_sha256"] or not meta.get("pmc_summary"):
            return None
        with open(os.path.join(ROOT, "profiles", meta["pmc_summary"])) as f:
            d = json.load(f)
        for name, v in d.items():
            if kernel in name and "SQ_INSTS_VALU" in v:
                nd = max(v.get("pmc_dispatches", 1), 1)
                insts = v["SQ_INSTS_VALU"] / nd * (n / 2**30)
                cus = int(props["compute_units"])
                simds, ghz = 4 * cus, props["clock_khz"] / 1e6
                cycles = kernel_ms * 1e-3 * ghz * 1e9
                lds, conf = v.get("SQ_LDS_IDX_ACTIVE"), v.get("SQ_LDS_BANK_CONFLICT")
                return {"bound": "mixed: LDS round trips between barriers (match phase), vector issue (parse, emit); not HBM",
                        "valu_wave_instructions_per_launch": int(insts), "simds": simds, "clock_ghz": round(ghz, 3),
                        "cycles_per_instruction_per_simd": round(cycles / (insts / simds), 2),
                        "lds_busy_frac": round(lds / nd * (n / 2**30) / cus / cycles, 3) if lds else None,
                        "lds_bank_conflict_frac": round(conf / lds, 3) if lds and conf is not None else None,
                        "commit": meta.get("commit"),
                        "source": f"profiles/{meta['pmc_summary']} (instruction and LDS-cycle counts; not collected live) / this run's kernel time"}
    except Exception:  # noqa: BLE001
        pass
    return None


def secondary_workload(comp, workload, n, dev, block_bytes, steps=3, effort="default", data=None, wl=None):
    """MiB/s + ratio of another BASELINE workload (configs[3] / [4] shapes), or of the main one at another effort."""
    import torch

    if data is None:
        data, wl = make_input(workload, n, 0, dev)
    out = torch.empty(comp.compress_bound(n), dtype=torch.uint8, device=dev)
    nb = 0
    for _ in range(2):
        _, nb = comp.compress_tensor(data, out=out, block_bytes=block_bytes, effort=effort)
    torch.cuda.synchronize()
    # a step of well under a millisecond (the stored path at 256 MiB) is timed over enough repetitions for the two
    # synchronisations to vanish: at least `steps`, at most 20, about 20 ms in all
    t0 = time.perf_counter()
    _, nb = comp.compress_tensor(data, out=out, block_bytes=block_bytes, effort=effort)
    torch.cuda.synchronize()
    one = time.perf_counter() - t0
    steps = int(min(20, max(steps, 0.02 / max(one, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(steps):
        _, nb = comp.compress_tensor(data, out=out, block_bytes=block_bytes, effort=effort)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    zs = min(n, 32 << 20)
    zl = zlib6_size(data[:zs].cpu().numpy().tobytes())
    from starflate_amd import _capi
    offs = comp.debug(_capi.DBG_OFFSETS, (n + SEG - 1) // SEG)
    ours = int(offs[zs // SEG]) if zs < n else nb
    ok = zlib.decompress(out[:nb].cpu().numpy().tobytes(), -15) == data.cpu().numpy().tobytes()
    res = {"workload": wl, "value": round(n / dt / 2**20, 1), "unit": "MiB/s", "ms": round(dt * 1e3, 3), "timed_steps": steps, "ratio": round(n / nb, 4),
           "ratio_vs_zlib6": round((zs / ours) / (zs / zl), 4), "roundtrip_ok": ok,
           "kernel_ms": {k: round(v, 4) for k, v in comp.stage_ms().items()}}
    if workload == "runs":
        res["note"] = ("a throughput probe: matches stop at 512-byte regions and their distances come from the hash table (256 and "
                       "more on a run, 7-8 extra bits each), every 32 KiB block carries its own header -- zlib codes a run as "
                       "258-byte matches at distance 1")
    return res


def self_launch(gpus):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) and relay rank 0's JSON
    line.  Runs before anything in this process has touched torch or HIP, and the parent never does: it only waits.
    Children are new processes (no fork of GPU state, no exec from a GPU process)."""
    import socket
    import subprocess

    import signal

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    limit = float(os.environ.get("STARFLATE_BENCH_LAUNCH_TIMEOUT", "1500"))  # the parent never waits longer than this
    grace = float(os.environ.get("STARFLATE_BENCH_KILL_GRACE", "5"))         # SIGTERM -> SIGKILL

    def stop_all():
        """terminate() exactly the rank processes started here, kill() what ignores it, reap everything."""
        alive = [p for p in procs if p.poll() is None]
        for p in alive:
            p.terminate()
        t_end = time.monotonic() + grace
        for p in alive:
            try:
                p.wait(max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
        for p in alive:
            try:
                p.wait(grace)
            except subprocess.TimeoutExpired:  # unkillable (stuck in the driver): nothing more a parent can do
                print(f"bench.py: rank process {p.pid} did not exit after SIGKILL", file=sys.stderr)

    class _Stop(Exception):
        pass

    def on_signal(signum, _frame):
        raise _Stop(signum)

    old = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        for r in range(gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), LOCAL_WORLD_SIZE=str(gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # only rank 0 prints the line; the other ranks' stdout goes to stderr so stdout stays one JSON line
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=None if r == 0 else sys.stderr))
        live, t_end = set(range(gpus)), time.monotonic() + limit
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with {code}; stopping the others", file=sys.stderr)
                    stop_all()
                    live.clear()
                    break
            if live and time.monotonic() > t_end:
                print(f"bench.py: ranks {sorted(live)} still running after {limit:.0f} s; stopping them", file=sys.stderr)
                rc = 124
                break
            if live:
                time.sleep(0.05)
    except _Stop as e:
        print(f"bench.py: signal {e.args[0]}; stopping the rank processes", file=sys.stderr)
        rc = 128 + int(e.args[0])
    finally:
        for sig, h in old.items():
            signal.signal(sig, signal.SIG_IGN)  # a second signal must not interrupt the clean-up
        stop_all()
        for sig, h in old.items():
            signal.signal(sig, h)
    return rc


LINE_LIMIT = 8192  # the driver keeps the last 8 KB of a bench line: the whole line stays below that (tests/test_bench_launch.py)


def compact_workload(res, n_bytes=None, basis=None):
    """A secondary workload as the line carries it: throughput, ratio against zlib -6, round trip, per-kernel times and its
    own roofline fraction -- `basis` "N+C over k_lz77" (a coded workload: the dominant kernel, like the headline) or
    "2N+5/chunk over the path" (the stored path: a copy, every kernel of the step)."""
    if res is None or "skipped" in res:
        return res
    out = {"workload": res["workload"].split(" (")[0][:60], "value": res["value"], "unit": "MiB/s", "ms": res["ms"], "ratio": res["ratio"],
           "ratio_vs_zlib6": res["ratio_vs_zlib6"], "roundtrip_ok": res["roundtrip_ok"], "kernel_ms": res["kernel_ms"]}
    if n_bytes and basis:
        km = res["kernel_ms"]
        if basis.startswith("2N"):
            alg, t_ms = 2 * n_bytes + 5 * ((n_bytes + SEG - 1) // SEG), sum(km.values())
        else:
            alg, t_ms = n_bytes + int(n_bytes / res["ratio"]), km.get("k_lz77", 0.0)
        if t_ms > 0:
            out["roofline"] = {"basis": basis, "bytes": alg, "ms": round(t_ms, 4), "frac": round(alg / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    return out


def assemble_line(head, kernel_ms, roofline, cpu, e2e, decomp, real_bytes, multi_gpu, configs, extra=None):
    """The ONE JSON line, in the order a reader -- and the driver's 8 KB tail -- needs it: the contract's keys, the dominant
    kernel's roofline, the CPU baseline, the short legs (e2e, decompress, real bytes, multi_gpu), and LAST the BASELINE
    configs at their default effort (`configs`: config[2] = the headline again in brief, config[3]'s bytes on one GPU,
    config[4])."""
    line = dict(head)
    line["kernel_ms"] = {k: round(v, 4) for k, v in kernel_ms.items()}
    line["kernels_total_ms"] = round(sum(kernel_ms.values()), 4)
    line["roofline"] = roofline
    line["cpu_baseline"] = cpu
    line["e2e"] = e2e
    line["decompress"] = decomp
    line["real_bytes"] = real_bytes
    if multi_gpu is not None:
        line["multi_gpu"] = multi_gpu
    if extra:
        line.update(extra)
    line["configs"] = configs
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bytes", type=int, default=1 << 30, help="input bytes per GPU")
    ap.add_argument("--workload", default="text", choices=["text", "random", "mixed", "runs"])
    ap.add_argument("--effort", default="default", choices=["default", "fast", "fastest", "thorough", "max", "best", "ultra", "extreme", "recent", "recent_all"], help="sfh_options.effort of the timed steps")
    ap.add_argument("--block-bytes", type=int, default=0, help="sfh_options.block_bytes (0 = the library's default: 512 KiB at 1 GiB, 1 MiB with the chain efforts)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the e2e leg, the other BASELINE workloads and the real-bytes workloads")
    ap.add_argument("--sweep", nargs="?", const="gpurun_out/sweep.json", default=None, metavar="PATH",
                    help="also run every effort level on the text, mixed and real-bytes workloads (minutes) and write the table to PATH "
                         "(default gpurun_out/sweep.json; the committed copy is profiles/rNN_sweep.json); the JSON line itself stays compact")
    ap.add_argument("--cpu-sample-bytes", type=int, default=512 << 20)
    ap.add_argument("--secondary-bytes", type=int, default=256 << 20)
    ap.add_argument("--rounds", type=int, default=4, help="N > 1: block-cyclic rounds per rank (gather/compute overlap)")
    ap.add_argument("--container", default="raw", choices=["raw", "zlib", "gzip"],
                    help="wrap the stream (RFC 1950 / 1952); the checksum kernels are then inside the timed step")
    ap.add_argument("--no-decompress", action="store_true", help="skip the GPU decompress leg (N = 1 only)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: run the N > 1 code path (RCCL group, rounds, gather) even with one rank")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = CPU rehearsal of the N > 1 launch / rounds / gather / verification with the stand-in compressor "
                         "of tests/bench_stub.py (zlib); its line says so and is not a measurement")
    ap.add_argument("--input-file", default=None,
                    help="compress this file's bytes (tiled / cut to --bytes) instead of a generator; also: $STARFLATE_CORPUS_DIR "
                         "(enwik9 / enwik8 / dickens / alice29.txt, first one found) when --workload text")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    # what ties this run to kernel sources (commit + SHA-256): taken ONCE, here -- before torch or anything else has touched
    # the GPU, so the `git` children it starts are not children of a GPU process (nor of a profiler's preloaded library)
    from starflate_amd.build import source_stamp

    stamp = source_stamp()

    import numpy as np
    import torch
    import torch.distributed as dist

    from starflate_amd import Compressor, _capi

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    rehearsal = args.backend == "gloo"
    multi = world > 1 or args.force_dist
    if rehearsal and not multi:
        raise SystemExit("--backend gloo rehearses the N > 1 path: use --gpus N > 1 or --force-dist")
    if rehearsal:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL (and gloo) print a banner on stdout when the communicator comes up: keep stdout for the one JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearsal:
                dist.init_process_group("gloo", rank=rank, world_size=world)
                dist.all_reduce(torch.zeros(1))
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                warm = torch.zeros(1, device=dev)
                dist.all_reduce(warm)
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    n = args.bytes
    corpus = args.input_file or find_corpus(args.workload)
    if corpus:
        data, wl = file_input(corpus, n, rank, dev)
    else:
        data, wl = make_input(args.workload, n, rank, dev)
    if rehearsal:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import bench_stub

        comp = bench_stub.Compressor()
    else:
        comp = Compressor(local_rank)
    comp.set_profiling(True)
    K = max(1, args.rounds) if multi else 1
    # every piece is compressed with the strip size of the whole shard, so rounds do not change the stream's ratio
    bb = _capi.resolve_block_bytes(args.block_bytes, n, args.effort)
    if n % (K * bb):
        raise SystemExit("--bytes must be a multiple of rounds * block_bytes")
    pieces = list(data.chunk(K))
    bound = comp.compress_bound(n // K)
    scratch = [torch.empty(bound, dtype=torch.uint8, device=dev) for _ in range(K)]
    size_dev = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(K)]
    # rank 0 receives every rank's streams of one step: sized by the bound (what compress_pipelined checks)
    gathered = torch.empty(bound * K * world + 64, dtype=torch.uint8, device=dev) if (multi and rank == 0) else None
    wbits = {"raw": -15, "zlib": 15, "gzip": 31}[args.container]
    stage_acc = {}
    result = {}

    def step():
        ms = {}

        def account():
            for name, v in comp.stage_ms().items():
                ms[name] = ms.get(name, 0.0) + v

        if not multi:
            out, total = comp.compress_tensor(data, out=scratch[0], container=args.container, block_bytes=bb, effort=args.effort)
            account()
            result["sizes"] = [total]
        else:
            def compress_fn(piece, final, k):  # enqueue only: the size stays on the device until the gather reads it
                comp.compress_tensor_async(piece, scratch[k], size_dev[k], final_stream=final, block_bytes=bb, effort=args.effort)
                return scratch[k], size_dev[k]

            # the arguments are the same in every step: the first one has them judged (by all ranks alike), the rest skip that
            out, total = pipelined_step(
                compress_fn, pieces, gathered, container=args.container,
                checksum_fn=(lambda piece, k: comp.checksum_tensor(piece, args.container)) if args.container != "raw" else None,
                validate=not result.get("validated", False), bound_fn=comp.compress_bound)
            result["validated"] = True
            result["sizes"] = torch.cat(size_dev).tolist()  # one read-back
        result["local_n"] = sum(result["sizes"])
        result["out"], result["total"] = out, total
        for name, v in ms.items():
            stage_acc.setdefault(name, []).append(v)

    def compress_fn_for_timing():
        def compress_fn(piece, final, k):
            comp.compress_tensor_async(piece, scratch[k], size_dev[k], final_stream=final, block_bytes=bb, effort=args.effort)
            return scratch[k], size_dev[k]

        return compress_fn

    def fence():
        if multi:
            dist.barrier()
        if not rehearsal:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    stage_acc.clear()
    dt = run_steps(step, fence, args.steps, 0)
    if multi:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    ms_per_step = dt / args.steps * 1e3
    # ---- N > 1: what the record needs to be read without DESIGN.md -- who took part, what the gather cost, what was expected ----
    multi_info = None
    if multi:
        if rehearsal:
            me = f"cpu rank {rank} pid {os.getpid()}"
        else:
            pr = torch.cuda.get_device_properties(dev)
            me = (f"{pr.name} pci {getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', 0):02x}:{getattr(pr, 'pci_device_id', 0):02x}"
                  f" uuid {getattr(pr, 'uuid', 'n/a')}")
        seen = [None] * world
        dist.all_gather_object(seen, me)
        # one more step, untimed, with events around each round's transfers (rank 0: the receives; the others: their send)
        timing = {}
        pipelined_step(compress_fn_for_timing(), pieces, gathered, container=args.container,
                       checksum_fn=(lambda piece, k: comp.checksum_tensor(piece, args.container)) if args.container != "raw" else None,
                       validate=False, bound_fn=comp.compress_bound, timing=timing)
        fence()
        multi_info = {"ranks_seen": dist.get_world_size(), "distinct_devices": len(set(seen)), "devices": seen,
                      "gather_ms_per_round": timing.get("gather_ms"), "rounds": K}
    total_in = n * world
    value = total_in * args.steps / dt / 2**20
    total_out = int(result["total"])
    local_n = int(result["local_n"])

    # ---- verification (untimed): every rank inflates its own piece streams with zlib ----
    ok = None
    if not args.no_verify:
        ok, my_crcs = verify_pieces(pieces, scratch, result["sizes"], -15 if multi else wbits)
        if multi:
            flag = torch.tensor([1 if ok else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())
            crc = torch.tensor(my_crcs, dtype=torch.int64, device=dev)
            crcs = [torch.zeros(K, dtype=torch.int64, device=dev) for _ in range(world)]
            dist.all_gather(crcs, crc)
            if rank == 0:
                ok = ok and verify_concatenation(result["out"][:total_out].cpu().numpy().tobytes(), wbits, total_in, n // K,
                                                 [c.tolist() for c in crcs])

    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return
    if rehearsal:
        print(json.dumps({"metric": "REHEARSAL of the N > 1 path on CPU (gloo, stand-in compressor) -- not a measurement",
                          "value": round(value, 1), "unit": "MiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(ms_per_step, 3), "rehearsal": True, "backend": "gloo", "compressor": comp.name,
                          "config": {"workload": wl, "parallelism": f"shard{world} block-cyclic x{K}"},
                          "compressed_bytes": total_out, "roundtrip_ok": ok,
                          "multi_gpu": dict(multi_info, gather_ms_per_step=round(sum(multi_info["gather_ms_per_round"] or [0.0]), 4),
                                            prediction=predict_scaling(world, K, ms_per_step, local_n))}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    # ---- ratio vs zlib -6 on a bounded sample of the same bytes ----
    stage_ms = {k: sum(v) / len(v) for k, v in stage_acc.items()}
    if multi:  # the async entry point keeps no per-call events: time one synchronous call of the first piece
        _, first_n = comp.compress_tensor(pieces[0], out=scratch[0], block_bytes=bb, final_stream=False)
        stage_ms = {k: v * K for k, v in comp.stage_ms().items()}
        last, last_n = pieces[0], int(first_n)  # (a piece no longer than the zlib sample is measured whole)
    else:
        last, last_n = pieces[-1], result["sizes"][-1]
    zs = min(last.numel(), 64 << 20)
    host_sample = last[:zs].cpu().numpy().tobytes()
    tz = time.perf_counter()
    zlen = zlib6_size(host_sample)
    tz = time.perf_counter() - tz
    nchunks = (last.numel() + SEG - 1) // SEG
    offs = comp.debug(_capi.DBG_OFFSETS, nchunks)  # chunk offsets of the last call (no extra launch)
    ours_sample = int(offs[zs // SEG]) if zs < last.numel() else last_n
    ratio = n / max(local_n, 1)
    ratio_zlib6 = zs / zlen
    ratio_ours_sample = zs / ours_sample

    # ---- roofline of the dominant kernel ----
    dom = max(stage_ms, key=stage_ms.get)
    alg_bytes = n + local_n  # read N + write C per launch of the path
    dom_ms = stage_ms[dom]
    if args.workload == "random" and not corpus:
        # the stored path is a copy made by the whole path (k_lz77 decides and counts, k_emit copies): 2N + 5 per chunk over
        # every kernel of the step -- N + C over the match kernel alone would credit it with bytes it does not write
        alg_bytes, dom, dom_ms = 2 * n + 5 * ((n + SEG - 1) // SEG), "k_lz77+k_plan+k_scan+k_emit", sum(stage_ms.values())
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
    try:  # what hipDeviceProp_t says; `peak` stays the guide's figure
        dp = _capi.device_props(local_rank)
    except Exception as e:  # noqa: BLE001
        dp = {"error": str(e)}
    traffic, traffic_info = traffic_from_profile(dom, n, stamp)
    text_default = args.effort == "default" and args.workload == "text" and not corpus and args.container == "raw" and not multi
    if not text_default:  # the profile is of the default command: another workload's traffic is not in it
        traffic, traffic_info = None, None
    issue = issue_from_profile(dom, dom_ms, n, dp, stamp) if text_default and "error" not in dp else None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_over_algorithmic": round(traffic / alg_bytes, 3) if traffic else None,
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(dom_ms, 4),
                "read_frac": round(n / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                # where `traffic` comes from: the committed counter passes of this command, and whether they are of the running sources
                "traffic_info": None if not traffic_info else {k: traffic_info[k] for k in ("commit", "csrc_sha256", "current", "read_bytes", "write_bytes")},
                "issue": None if not issue else {k: issue[k] for k in ("valu_wave_instructions_per_launch", "cycles_per_instruction_per_simd",
                                                                         "lds_busy_frac", "lds_bank_conflict_frac")}}
    if "error" in dp:
        roofline["device"] = dp
    else:
        roofline["device"] = {"name": dp["name"], "arch": dp["arch"], "compute_units": dp["compute_units"], "clock_khz": dp.get("clock_khz"),
                              # HBM3E moves 8 Gb/s per pin = 4 transfers per reported 2 GHz memory clock
                              "hbm_peak_from_props_GBs": round(4 * dp["memory_clock_khz"] * 1e3 * dp["memory_bus_bits"] / 8 / 1e9, 1)}
    kern_total_ms = sum(stage_ms.values())

    # ---- GPU decompress of the stream just made: the reference's own function, on the GPU ----
    decomp = None
    if not multi and not args.no_decompress:
        out_t, nb = comp.compress_tensor(data, out=scratch[0], container=args.container, block_bytes=bb)
        index = comp.last_index(device=dev)
        subindex = comp.last_subindex(device=dev)
        stream_t = out_t[:nb].clone()
        back = torch.empty(n, dtype=torch.uint8, device=dev)
        decomp = {}
        for label, sub in (("sub_indexed", subindex), ("segment_indexed", None)):
            back.zero_()
            comp.decompress_tensor(stream_t, index, n, out=back, subindex=sub, block_bytes=bb)  # warm-up
            torch.cuda.synchronize()
            reps = 3
            td = time.perf_counter()
            for _ in range(reps):
                _, dstatus = comp.decompress_tensor(stream_t, index, n, out=back, subindex=sub, block_bytes=bb)
            torch.cuda.synchronize()
            td = (time.perf_counter() - td) / reps
            decomp[label] = {"value": round(n / td / 2**20, 1), "unit": "MiB/s of output", "ms": round(td * 1e3, 3),
                             "status": dstatus, "equal_to_input": bool(torch.equal(back, data)),
                             "kernel_ms": {k: round(v, 4) for k, v in comp.inflate_ms().items()}}
        # (sub_indexed: chunk offsets + 32 sub-index entries per chunk, sfh_copy_index / sfh_copy_subindex; segment_indexed: chunk offsets
        # only -- 32 lanes per segment find their token boundaries speculatively, k_inflate_tokens_spec)
        del back, stream_t
        # a FOREIGN indexed stream: zlib -6 with Z_FULL_FLUSH every 32 KiB over the first 256 MiB of the same input -- what the
        # reference's own flushed fixtures are (tools/deflate_compress.py --flush); segment index only, every segment independent
        if not args.no_secondary:
            import zlib
            import numpy as np
            zn = min(n, 256 << 20)
            host = data[:zn].cpu().numpy()
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            nseg = (zn + 32767) // 32768
            parts = [co.compress(host[c * 32768:(c + 1) * 32768].tobytes()) + co.flush(zlib.Z_FINISH if c == nseg - 1 else zlib.Z_FULL_FLUSH)
                     for c in range(nseg)]
            zidx = torch.from_numpy(np.concatenate([[0], np.cumsum([len(q) for q in parts])]).astype(np.int64)).to(dev)
            zraw = np.frombuffer(b"".join(parts), np.uint8)
            zstream = torch.from_numpy(zraw.copy()).to(dev)
            zback = torch.empty(zn, dtype=torch.uint8, device=dev)
            comp.decompress_tensor(zstream, zidx, zn, out=zback, block_bytes=32768)  # warm-up
            torch.cuda.synchronize()
            reps = 5
            td = time.perf_counter()
            for _ in range(reps):
                _, zstatus = comp.decompress_tensor(zstream, zidx, zn, out=zback, block_bytes=32768)
            torch.cuda.synchronize()
            td = (time.perf_counter() - td) / reps
            from starflate_amd import _capi
            by_serial = int(((comp.debug(_capi.DBG_SEGINFO, nseg)[:, 2] >> 1) & 1).sum())
            # (made by zlib.compressobj(6, DEFLATED, -15) with Z_FULL_FLUSH every 32768 bytes of input)
            decomp["zlib_made_segment_indexed"] = {
                "segments": nseg, "by_lane_serial_kernel": by_serial,
                "value": round(zn / td / 2**20, 1), "unit": "MiB/s of output", "ms": round(td * 1e3, 3), "bytes": zn,
                "status": zstatus, "equal_to_input": bool(torch.equal(zback, data[:zn])),
                "kernel_ms": {k: round(v, 4) for k, v in comp.inflate_ms().items()}}
            del zback, zstream

    # ---- end to end from pinned host memory (H2D + kernels + D2H), and the other two workloads ----
    e2e, others = None, None
    if not multi and not args.no_secondary:
        hin = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        hin.copy_(data)
        hout = torch.empty(comp.compress_bound(n), dtype=torch.uint8, pin_memory=True)
        src_np, dst_np = hin.numpy(), hout.numpy()
        import ctypes as C
        opt = _capi.make_options(container=args.container, block_bytes=bb)
        out_n = C.c_size_t(0)
        call = lambda: comp._check(comp._lib.sfh_compress(comp._h, src_np.ctypes.data, n, dst_np.ctypes.data, dst_np.size,  # noqa: E731
                                                          C.byref(out_n), C.byref(opt)))
        call()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            call()
        te = (time.perf_counter() - t0) / reps
        # (sfh_compress from / to pinned host buffers, the call as a whole: inside it 64 MiB batches are pipelined -- H2D of batch b
        # beside the kernels of batch b-1 beside the D2H of batch b-2's stream bytes; PCIe-bound, never `value`)
        e2e = {"value": round(n / te / 2**20, 1), "unit": "MiB/s", "ms": round(te * 1e3, 3), "bytes_out": int(out_n.value), "what": "sfh_compress, pinned host buffers"}
        del hin, hout
        others = {w: secondary_workload(comp, w, args.secondary_bytes, dev, 0) for w in ("text", "mixed", "random", "runs") if w != args.workload}
        # REAL bytes (starflate_amd/realbytes.py): source text and x86-64 machine code from files of this image -- the
        # generators above have no long-range structure, these do (ratio_vs_zlib6 on real data is what a user will see).
        # The line carries the default effort and the one INTEGRATION.md recommends for such data (recent_all); --sweep: all
        from starflate_amd import realbytes

        sweep = {} if args.sweep else None
        for key, buf, what in (("real_source", realbytes.source(96 << 20), "Python / C++ source text (stdlib, torch, ROCm headers)"),
                               ("real_binary", realbytes.binary(min(args.secondary_bytes, 256 << 20)), "x86-64 code + data (head of libtorch_cpu.so)")):
            nb = buf.size // (1 << 20) * (1 << 20)  # whole MiB: a multiple of every strip size the default rule picks here
            if nb < (8 << 20):
                others[key] = {"skipped": f"{what}: not in this image"}
                continue
            desc = realbytes.describe(buf[:nb], what)
            t = torch.from_numpy(buf[:nb]).to(dev)
            if nb < (256 << 20):
                # fewer strips than workgroup slots would time ONE strip, not the chip: repeat the corpus.  Strips are coded
                # independently and a match reaches back 32 KiB, so a period of ~100 MiB changes neither parse nor ratio
                # (ratio_vs_zlib6 is taken on the first 32 MiB, where zlib sees no repetition either)
                reps = -(-(256 << 20) // nb)
                t = t.repeat(reps)
                desc += f", repeated x{reps} for throughput"
                nb *= reps
            for eff in ("default", "recent_all") + (("thorough", "max", "chain2", "chain4", "best", "ultra", "extreme") if args.sweep else ()):
                res = secondary_workload(comp, key, nb, dev, 0, effort=eff, data=t, wl=desc)
                if eff in ("default", "recent_all"):
                    others[key if eff == "default" else f"{key}_effort_{eff}"] = res
                if sweep is not None:
                    sweep[f"{key}_effort_{eff}"] = res
            del t
        if sweep is not None:
            # every effort on the headline bytes and on the mixed workload (block_bytes 0 for the chain efforts: their own default strip)
            for eff in ("fastest", "fast", "default", "thorough", "max", "recent_all", "chain4", "best", "ultra", "extreme"):
                chain = eff in ("chain4", "best", "ultra", "extreme")
                sweep[f"{args.workload}_effort_{eff}"] = secondary_workload(comp, args.workload, n, dev, 0 if chain else bb, effort=eff, data=data, wl=wl)
                sweep[f"mixed_effort_{eff}"] = secondary_workload(comp, "mixed", args.secondary_bytes, dev, 0, effort=eff)

    # ---- CPU baseline: oracle restatement of the reference decompress(), 1 thread ----
    cpu = None
    if not args.no_cpu_baseline and world == 1 and not args.force_dist:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O

        cs = min(n, args.cpu_sample_bytes)
        sample = data[:cs].clone()
        sout, sn = comp.compress_tensor(sample, block_bytes=bb)
        stream = sout[:sn].cpu().numpy()
        tc = time.perf_counter()
        st, w, back = O.decompress(stream, cs)
        tc = time.perf_counter() - tc
        good = st == 0 and w == cs and np.array_equal(back, sample.cpu().numpy())
        cpu = {"value": round(cs / tc / 2**20, 2), "unit": "MiB/s", "cores": 1, "kind": "port",
               "sample": f"oracle sfo_decompress (restates reference src/decompress.cpp:402-461) on the GPU-made stream of the first "
                         f"{cs >> 20} MiB of the workload; output MiB/s; equal={good}; host has {os.cpu_count()} logical cores",
               "zlib6_compress_MiBps_1core": round(zs / tz / 2**20, 2)}
        # block-parallel zlib -6 on this host's cores (1 MiB independent slices; zlib releases the GIL)
        from concurrent.futures import ThreadPoolExecutor

        nthreads = min(16, os.cpu_count() or 1)
        sl = [host_sample[i:i + (1 << 20)] for i in range(0, zs, 1 << 20)]
        tp = time.perf_counter()
        with ThreadPoolExecutor(nthreads) as ex:
            zpar = sum(ex.map(zlib6_size, sl))
        tp = time.perf_counter() - tp
        cpu["zlib6_compress_MiBps_block_parallel"] = {"value": round(zs / tp / 2**20, 1), "threads": nthreads,
                                                      "ratio": round(zs / zpar, 4)}
        ok = good if ok is None else (ok and good)

    metric = "compress MiB/s + ratio vs zlib -6, 1 GiB synthetic; 1/2/4/8 GPU"
    if corpus:  # not BASELINE.json's workload any more: the line says so where a reader looks first
        tiled = os.path.getsize(corpus) < n
        metric = (f"compress MiB/s + ratio vs zlib -6 on {n / 2**30:g} GiB of FILE {os.path.basename(corpus)}"
                  + (" (shorter than that: repeated; the ratio is taken on the first 64 MiB at most)" if tiled else ""))
    head = {
        "metric": metric,
        "value": round(value, 1), "unit": "MiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": f"file:{os.path.basename(corpus)}" if corpus else "synthetic",
        "config": {"workload": wl, "effort": args.effort, "block_bytes": bb, "deflate_block_bytes": SEG, "window_bytes": 32768, "strategy": "auto",
                   "container": args.container,
                   "parallelism": f"shard{world}" + (f" block-cyclic x{K}, gather overlapped" if multi else "")},
        "ratio": round(ratio, 4), "ratio_zlib6": round(ratio_zlib6, 4),
        "ratio_vs_zlib6": round(ratio_ours_sample / ratio_zlib6, 4),
        "compressed_bytes": total_out, "roundtrip_ok": ok,
    }
    mg = None
    if multi_info is not None:
        # RCCL saw these devices; the gather of one step as measured (events around each round's transfers on rank 0) beside
        # what the xGMI arithmetic of DESIGN.md section 4 predicts for this step's own numbers
        mg = dict(multi_info, gather_ms_per_step=round(sum(multi_info["gather_ms_per_round"] or [0.0]), 4),
                  prediction=predict_scaling(world, K, kern_total_ms, local_n))
    # the BASELINE configs at the default effort, LAST on the line (the driver keeps a line's tail): config[2] = the timed workload
    # in brief, config[3]'s bytes on this one GPU (the 8-rank run is `--gpus 8 --workload mixed`), config[4] = the stored path
    this = {"workload": wl.split(" (")[0][:60], "value": round(value, 1), "unit": "MiB/s", "ms": round(ms_per_step, 3), "n_gpus": world,
            "ratio_vs_zlib6": round(ratio_ours_sample / ratio_zlib6, 4), "roundtrip_ok": ok, "roofline_frac": roofline["frac"],
            "roofline_read_frac": roofline["read_frac"]}
    oth = others or {}
    key_of = {"text": "config2_text", "mixed": "config3_mixed_1gpu" if world == 1 else f"config3_mixed_{world}gpu", "random": "config4_random", "runs": "runs"}
    configs = {key_of[args.workload] if not corpus else "file": this}
    sb = args.secondary_bytes
    for w, basis in (("text", "N+C over k_lz77"), ("mixed", "N+C over k_lz77"), ("random", "2N+5/chunk over the path")):
        if w in oth:
            configs[key_of[w]] = compact_workload(oth[w], sb, basis)
    real = {k: compact_workload(v) for k, v in oth.items() if k.startswith("real_")} or None
    extra = {"runs": compact_workload(oth["runs"])} if "runs" in oth else None
    line = assemble_line(head, stage_ms, roofline, cpu, e2e, decomp, real, mg, configs, extra)
    if args.sweep and others is not None:
        os.makedirs(os.path.dirname(os.path.abspath(args.sweep)), exist_ok=True)
        with open(args.sweep, "w") as f:
            json.dump({"_meta": {"source": stamp, "bytes": n, "secondary_bytes": sb, "what": "every effort level on the timed workload, the mixed workload and "
                                 "the real-bytes workloads (bench.py --sweep); MiB/s of the whole path, ratio against zlib -6 on the first 32 MiB"},
                       "headline": this, "sweep": sweep}, f, indent=1)
        line["sweep_file"] = args.sweep
        line["configs"] = line.pop("configs")  # stays last
    text = json.dumps(line)
    if len(text) >= LINE_LIMIT:
        print(f"bench.py: the line is {len(text)} bytes, over the {LINE_LIMIT} the driver keeps", file=sys.stderr)
    print(text, flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
