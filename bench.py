#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X DEFLATE compressor.

Workload (BASELINE.json configs[2]): 1 GiB synthetic enwik-like text per GPU,
32,768 independent 32 KiB DEFLATE blocks, input resident in HBM when the timed
region starts.  A "step" = one pass of the whole hot path (k_lz77 -> k_plan ->
k_scan -> k_emit) over that input; with N > 1 every rank compresses its own 1 GiB
shard (weak scaling) and the byte-aligned streams are concatenated on rank 0 over
RCCL.  For N > 1 the sharding is block-cyclic in --rounds rounds (global piece g = k*N + rank),
so the gather of round k (sizes all_gather + one point-to-point send per rank, straight to
the final offset) overlaps the compression of round k+1 (starflate_amd/multigpu.py).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bytes B] [--workload text|random|mixed]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (k_lz77),
its launch time measured live with HIP events on the launch stream; `cpu_baseline`
times the oracle's restatement of the reference decompress() on this host.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bytes", type=int, default=1 << 30, help="input bytes per GPU")
    ap.add_argument("--workload", default="text", choices=["text", "random", "mixed"])
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-bytes", type=int, default=512 << 20)
    ap.add_argument("--rounds", type=int, default=4, help="N > 1: block-cyclic rounds per rank (gather/compute overlap)")
    ap.add_argument("--container", default="raw", choices=["raw", "zlib", "gzip"],
                    help="wrap the stream (RFC 1950 / 1952); the checksum kernels are then inside the timed step")
    ap.add_argument("--no-decompress", action="store_true", help="skip the GPU decompress leg (N = 1 only)")
    ap.add_argument("--block-bytes", type=int, default=0, help="sfh_options.block_bytes (0 = the library's default)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: run the N > 1 code path (RCCL group, rounds, gather) even with one rank")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from starflate_amd import Compressor, multigpu, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n = args.bytes
    if args.workload == "text":
        data = synth.gen_text_torch(n, seed=3 + 17 * rank, device=dev)
        wl = f"{n / 2**30:g} GiB synthetic enwik-like text per GPU (gen_text_torch seed 3), 32 KiB blocks"
    elif args.workload == "random":
        g = torch.Generator(device=dev)
        g.manual_seed(5 + rank)
        data = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=g)
        wl = f"{n / 2**30:g} GiB high-entropy bytes per GPU (stored-block path)"
    else:
        data = torch.from_numpy(synth.gen_mixed(n, seed=4 + rank)).to(dev)
        wl = f"{n / 2**30:g} GiB mixed Silesia-like stripes per GPU"

    comp = Compressor(local_rank)
    comp.set_profiling(True)
    K = max(1, args.rounds) if multi else 1
    if n % (K * 32768):
        raise SystemExit("--bytes must be a multiple of rounds * 32768")
    pieces = list(data.chunk(K))
    bound = comp.compress_bound(n // K)
    scratch = [torch.empty(bound, dtype=torch.uint8, device=dev) for _ in range(K)]
    gathered = torch.empty(bound * K * world + 32, dtype=torch.uint8, device=dev) if (multi and rank == 0) else None
    wbits = {"raw": -15, "zlib": 15, "gzip": 31}[args.container]
    stage_acc = {}
    result = {}

    def step():
        ms = {}
        sizes = [0] * K

        def compress_fn(piece, final, k):
            out, nb = comp.compress_tensor(piece, out=scratch[k], final_stream=final,
                                           container="raw" if multi else args.container, block_bytes=args.block_bytes)
            sizes[k] = nb
            for name, v in comp.stage_ms().items():
                ms[name] = ms.get(name, 0.0) + v
            return out, nb

        if not multi:
            out, total = compress_fn(data, True, 0)
        else:
            out, total = multigpu.compress_pipelined(
                compress_fn, pieces, out=gathered, container=args.container,
                checksum_fn=(lambda piece, k: comp.checksum_tensor(piece, args.container)) if args.container != "raw" else None)
        result["sizes"], result["local_n"] = sizes, sum(sizes)
        result["out"], result["total"] = out, total
        for name, v in ms.items():
            stage_acc.setdefault(name, []).append(v)

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    stage_acc.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if multi:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    ms_per_step = dt / args.steps * 1e3
    total_in = n * world
    value = total_in * args.steps / dt / 2**20
    total_out = int(result["total"])
    local_n = int(result["local_n"])

    # ---- verification (untimed): every rank inflates its own shard stream with zlib ----
    ok = None
    if not args.no_verify:
        ok = True
        my_crcs = []
        for k in range(K):
            host_piece = pieces[k].cpu().numpy().tobytes()
            stream = scratch[k][: result["sizes"][k]].cpu().numpy().tobytes()
            back = zlib.decompressobj(-15 if multi else wbits).decompress(stream)  # a non-final piece is still inflatable
            ok = ok and back == host_piece
            my_crcs.append(zlib.crc32(host_piece))
            del back, host_piece
        if multi:
            flag = torch.tensor([1 if ok else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())
            crc = torch.tensor(my_crcs, dtype=torch.int64, device=dev)
            crcs = [torch.zeros(K, dtype=torch.int64, device=dev) for _ in range(world)]
            dist.all_gather(crcs, crc)
            if rank == 0:  # the concatenation is ONE valid stream of all pieces in the order g = k*N + rank
                whole = zlib.decompress(result["out"][:total_out].cpu().numpy().tobytes(), wbits)
                pb = n // K
                ok = ok and len(whole) == total_in and all(
                    zlib.crc32(whole[(k * world + r) * pb:(k * world + r + 1) * pb]) == int(crcs[r][k].item())
                    for k in range(K) for r in range(world))
                del whole

    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return

    # ---- ratio vs zlib -6 on a bounded sample of the same bytes ----
    last = pieces[-1]  # the piece of the last compress call (its chunk offsets are still in the ctx)
    zs = min(last.numel(), 64 << 20)
    host_sample = last[:zs].cpu().numpy().tobytes()
    tz = time.perf_counter()
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    zlen = len(co.compress(host_sample)) + len(co.flush())
    tz = time.perf_counter() - tz
    # our bytes for the same prefix: chunk offsets of the last timed call (no extra launch)
    from starflate_amd import _capi
    nchunks = (last.numel() + 32767) // 32768
    offs = comp.debug(_capi.DBG_OFFSETS, nchunks)
    ours_sample = int(offs[zs // 32768]) if zs < last.numel() else result["sizes"][-1]
    ratio = n / max(local_n, 1)
    ratio_zlib6 = zs / zlen
    ratio_ours_sample = zs / ours_sample

    # ---- roofline of the dominant kernel ----
    stage_ms = {k: sum(v) / len(v) for k, v in stage_acc.items()}
    dom = max(stage_ms, key=stage_ms.get)
    alg_bytes = n + local_n  # read N + write C per launch of the path
    achieved = alg_bytes / (stage_ms[dom] * 1e-3) / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            with open(pmc_path) as f:
                traffic = json.load(f).get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(stage_ms[dom], 4),
                "read_frac": round(n / (stage_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
    try:  # what hipDeviceProp_t implies; `peak` above stays the guide's figure
        from starflate_amd import _capi as _c
        dp = _c.device_props(local_rank)
        roofline["device"] = {"name": dp["name"], "arch": dp["arch"], "compute_units": dp["compute_units"],
                              "memory_clock_khz": dp["memory_clock_khz"], "memory_bus_bits": dp["memory_bus_bits"],
                              # HBM3E moves 8 Gb/s per pin = 4 transfers per reported 2 GHz memory clock
                              "hbm_peak_from_props_GBs": round(4 * dp["memory_clock_khz"] * 1e3 * dp["memory_bus_bits"] / 8 / 1e9, 1)}
    except Exception as e:  # noqa: BLE001
        roofline["device"] = {"error": str(e)}
    kern_total_ms = sum(stage_ms.values())

    # ---- GPU decompress of the stream just made: the reference's own function, on the GPU ----
    decomp = None
    if not multi and not args.no_decompress:
        out_t, nb = comp.compress_tensor(data, out=scratch[0], container=args.container)
        index = comp.last_index(device=dev)
        subindex = comp.last_subindex(device=dev)
        stream_t = out_t[:nb].clone()
        back = torch.empty(n, dtype=torch.uint8, device=dev)
        decomp = {}
        for label, sub in (("sub_indexed", subindex), ("segment_indexed", None)):
            back.zero_()
            comp.decompress_tensor(stream_t, index, n, out=back, subindex=sub)  # warm-up
            torch.cuda.synchronize()
            reps = 3
            td = time.perf_counter()
            for _ in range(reps):
                _, dstatus = comp.decompress_tensor(stream_t, index, n, out=back, subindex=sub)
            torch.cuda.synchronize()
            td = (time.perf_counter() - td) / reps
            decomp[label] = {"value": round(n / td / 2**20, 1), "unit": "MiB/s of output", "ms": round(td * 1e3, 3),
                             "status": dstatus, "equal_to_input": bool(torch.equal(back, data)),
                             "kernel_ms": {k: round(v, 4) for k, v in comp.inflate_ms().items()}}
        decomp["note"] = ("sub_indexed: chunk offsets + 32 region entries per chunk (sfh_copy_index, sfh_copy_subindex), "
                          "32 lanes per segment; segment_indexed: chunk offsets only, one lane per segment (any indexed stream)")
        del back, stream_t

    # ---- CPU baseline: oracle restatement of the reference decompress(), 1 thread ----
    cpu = None
    if not args.no_cpu_baseline and world == 1 and not args.force_dist:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O

        cs = min(n, args.cpu_sample_bytes)
        sample = data[:cs].clone()
        sout, sn = comp.compress_tensor(sample)
        stream = sout[:sn].cpu().numpy()
        tc = time.perf_counter()
        st, w, back = O.decompress(stream, cs)
        tc = time.perf_counter() - tc
        good = st == 0 and w == cs and np.array_equal(back, sample.cpu().numpy())
        cpu = {"value": round(cs / tc / 2**20, 2), "unit": "MiB/s", "cores": 1, "kind": "port",
               "sample": f"oracle sfo_decompress (restates reference src/decompress.cpp:402-461) of the GPU-made stream "
                         f"of the first {cs >> 20} MiB of the workload; output MiB/s; round-trip equal={good}; "
                         f"host has {os.cpu_count()} logical cores",
               "zlib6_compress_MiBps_1core": round(zs / tz / 2**20, 2)}
        # block-parallel zlib -6 on this host's cores (1 MiB independent slices; zlib releases the GIL)
        from concurrent.futures import ThreadPoolExecutor

        nthreads = min(16, os.cpu_count() or 1)
        sl = [host_sample[i:i + (1 << 20)] for i in range(0, zs, 1 << 20)]

        def _z(b):
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            return len(co.compress(b)) + len(co.flush())

        tp = time.perf_counter()
        with ThreadPoolExecutor(nthreads) as ex:
            zpar = sum(ex.map(_z, sl))
        tp = time.perf_counter() - tp
        cpu["zlib6_compress_MiBps_block_parallel"] = {"value": round(zs / tp / 2**20, 1), "threads": nthreads,
                                                      "ratio": round(zs / zpar, 4)}
        ok = good if ok is None else (ok and good)

    line = {
        "metric": "compress MiB/s + ratio vs zlib -6, 1 GiB synthetic; 1/2/4/8 GPU",
        "value": round(value, 1), "unit": "MiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": wl, "block_bytes": 32768, "strategy": "auto", "container": args.container,
                   "parallelism": f"shard{world}" + (f" block-cyclic x{K}, gather overlapped" if multi else "")},
        "ratio": round(ratio, 4), "ratio_zlib6": round(ratio_zlib6, 4),
        "ratio_vs_zlib6": round(ratio_ours_sample / ratio_zlib6, 4),
        "compressed_bytes": total_out, "roundtrip_ok": ok,
        "kernel_ms": {k: round(v, 4) for k, v in stage_ms.items()}, "kernels_total_ms": round(kern_total_ms, 4),
        "roofline": roofline, "cpu_baseline": cpu, "decompress": decomp,
    }
    print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
