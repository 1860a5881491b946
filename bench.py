#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X DEFLATE compressor.

Workload (BASELINE.json configs[2]): 1 GiB synthetic enwik-like text per GPU,
32,768 independent 32 KiB DEFLATE blocks, input resident in HBM when the timed
region starts.  A "step" = one pass of the whole hot path (k_lz77 -> k_plan ->
k_scan -> k_emit) over that input; with N > 1 every rank compresses its own 1 GiB
shard (weak scaling) and the byte-aligned streams are concatenated on rank 0 over
RCCL (sizes all_gather + one point-to-point send per rank).

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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
    bound = comp.compress_bound(n)
    scratch = torch.empty(bound, dtype=torch.uint8, device=dev)
    gathered = torch.empty(bound * world, dtype=torch.uint8, device=dev) if (world > 1 and rank == 0) else None
    stage_acc = {}
    result = {}

    def step():
        if world == 1:
            out, total = comp.compress_tensor(data, out=scratch)
            result["local"], result["local_n"] = out, total
        else:
            local, ln = comp.compress_tensor(data, out=scratch, final_stream=(rank == world - 1))
            out, total = multigpu.concat_streams(local, ln, out=gathered)
            result["local"], result["local_n"] = local, ln
        result["out"], result["total"] = out, total
        for k, v in comp.stage_ms().items():
            stage_acc.setdefault(k, []).append(v)

    def fence():
        if world > 1:
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
    if world > 1:
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
        host_in = data.cpu().numpy()
        stream = result["local"][:local_n].cpu().numpy().tobytes()
        d = zlib.decompressobj(-15)
        back = d.decompress(stream)
        ok = len(back) == n and back == host_in.tobytes()
        del back
        if world > 1:
            flag = torch.tensor([1 if ok else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())
            crc = torch.tensor([zlib.crc32(host_in.tobytes())], dtype=torch.int64, device=dev)
            crcs = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
            dist.all_gather(crcs, crc)
            if rank == 0:
                whole = zlib.decompress(result["out"][:total_out].cpu().numpy().tobytes(), -15)
                ok = ok and len(whole) == total_in and all(
                    zlib.crc32(whole[r * n:(r + 1) * n]) == int(crcs[r].item()) for r in range(world))
                del whole

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- ratio vs zlib -6 on a bounded sample of the same bytes ----
    zs = min(n, 64 << 20)
    host_sample = data[:zs].cpu().numpy().tobytes()
    tz = time.perf_counter()
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    zlen = len(co.compress(host_sample)) + len(co.flush())
    tz = time.perf_counter() - tz
    # our bytes for the same prefix: chunk offsets of the last timed call (no extra launch)
    from starflate_amd import _capi
    nchunks = (n + 32767) // 32768
    offs = comp.debug(_capi.DBG_OFFSETS, nchunks)
    ours_sample = int(offs[zs // 32768]) if zs < n else local_n
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
    kern_total_ms = sum(stage_ms.values())

    # ---- CPU baseline: oracle restatement of the reference decompress(), 1 thread ----
    cpu = None
    if not args.no_cpu_baseline and world == 1:
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
        ok = good if ok is None else (ok and good)

    line = {
        "metric": "compress MiB/s + ratio vs zlib -6, 1 GiB synthetic; 1/2/4/8 GPU",
        "value": round(value, 1), "unit": "MiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": wl, "block_bytes": 32768, "strategy": "auto", "parallelism": f"shard{world}"},
        "ratio": round(ratio, 4), "ratio_zlib6": round(ratio_zlib6, 4),
        "ratio_vs_zlib6": round(ratio_ours_sample / ratio_zlib6, 4),
        "compressed_bytes": total_out, "roundtrip_ok": ok,
        "kernel_ms": {k: round(v, 4) for k, v in stage_ms.items()}, "kernels_total_ms": round(kern_total_ms, 4),
        "roofline": roofline, "cpu_baseline": cpu,
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
