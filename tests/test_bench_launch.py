"""`python bench.py --gpus N` starts its N ranks by itself (the driver's command line) -- rehearsed here on CPU over
gloo with the stand-in compressor of tests/bench_stub.py: launch, block-cyclic rounds, gather, verification, one
JSON line on stdout, non-zero exit when a rank fails."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(*extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *extra], capture_output=True, text=True, timeout=300, env=e)


@pytest.mark.parametrize("gpus,container", [(2, "raw"), (3, "gzip")])
def test_bench_self_launch_gloo(gpus, container):
    r = _run("--gpus", str(gpus), "--backend", "gloo", "--bytes", str(1 << 20), "--rounds", "2", "--steps", "2", "--warmup", "1",
             "--container", container)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["rehearsal"] is True and d["roundtrip_ok"] is True
    assert d["steps"] == 2 and d["compressed_bytes"] > 0 and d["value"] > 0
    # the record explains itself: who took part, what the gather cost per round, what the xGMI arithmetic expects
    m = d["multi_gpu"]
    assert m["ranks_seen"] == gpus and m["distinct_devices"] == gpus and len(m["devices"]) == gpus and m["rounds"] == 2
    assert len(m["gather_ms_per_round"]) == 2 and all(t >= 0 for t in m["gather_ms_per_round"]) and m["gather_ms_per_step"] >= 0
    pr = m["prediction"]
    assert {"model", "T_c_ms", "T_x_ms", "predicted_step_ms", "predicted_efficiency", "bound"} <= set(pr)
    assert 0 < pr["predicted_efficiency"] <= 1 and pr["T_x_ms"] > 0


def test_bench_self_launch_propagates_failure():
    # --bytes not a multiple of rounds * strip size: every rank exits with an error, and so must the parent
    r = _run("--gpus", "2", "--backend", "gloo", "--bytes", str((1 << 20) + 1), "--rounds", "2")
    assert r.returncode != 0
    assert not r.stdout.strip()


def test_bench_input_file(tmp_path):
    # a "corpus" shorter than --bytes is tiled; found through $STARFLATE_CORPUS_DIR
    p = tmp_path / "alice29.txt"
    p.write_bytes(b"Alice was beginning to get very tired of sitting by her sister on the bank. " * 700)
    r = _run("--gpus", "2", "--backend", "gloo", "--bytes", str(1 << 18), "--rounds", "1", "--steps", "1", "--warmup", "0",
             env={"STARFLATE_CORPUS_DIR": str(tmp_path)})
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip())
    assert "alice29.txt" in d["config"]["workload"] and d["roundtrip_ok"] is True


def _children(pid):
    import psutil

    try:
        return psutil.Process(pid).children(recursive=True)
    except psutil.NoSuchProcess:
        return []


def test_bench_self_launch_stops_its_ranks_when_signalled():
    """The launcher must not leave rank processes behind (on a GPU box they would sit on the GPUs, possibly inside an RCCL
    collective): SIGTERM to the parent terminates every rank it started, and it exits non-zero."""
    import signal
    import time

    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--bytes", str(1 << 20), "--rounds", "2",
                          "--steps", "1000000", "--warmup", "0"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        t_end = time.monotonic() + 60
        kids = []
        while time.monotonic() < t_end and len(kids) < 2:
            time.sleep(0.2)
            kids = _children(p.pid)
        assert len(kids) >= 2, "the launcher did not start its ranks"
        time.sleep(2.0)  # let them get into the step loop
        p.send_signal(signal.SIGTERM)
        rc = p.wait(timeout=60)
        assert rc == 128 + signal.SIGTERM
        t_end = time.monotonic() + 10
        while time.monotonic() < t_end and any(k.is_running() and k.status() != "zombie" for k in kids):
            time.sleep(0.1)
        assert not any(k.is_running() and k.status() != "zombie" for k in kids), "rank processes survived the launcher"
    finally:
        for k in _children(p.pid):
            k.kill()
        if p.poll() is None:
            p.kill()


def test_bench_self_launch_has_a_deadline():
    """A rank that never finishes (stuck in the driver, say) must not make the parent spin for ever."""
    r = _run("--gpus", "2", "--backend", "gloo", "--bytes", str(1 << 20), "--rounds", "2", "--steps", "1000000", "--warmup", "0",
             env={"STARFLATE_BENCH_LAUNCH_TIMEOUT": "8", "STARFLATE_BENCH_KILL_GRACE": "2"})
    assert r.returncode == 124, (r.returncode, r.stderr[-1500:])
    assert "still running" in r.stderr


def _mock_workload(name, kernels=("k_lz77", "k_plan", "k_scan", "k_emit", "k_checksum")):
    return {"workload": f"0.25 GiB {name} per GPU (generator seed 4, stripes of text / binary / noise)", "value": 1234567.8, "unit": "MiB/s", "ms": 1.234,
            "timed_steps": 20, "ratio": 2.6294, "ratio_vs_zlib6": 0.9275, "roundtrip_ok": True, "kernel_ms": {k: 1.2345 for k in kernels}}


def test_bench_line_layout_and_size():
    """The driver keeps the last 8 KB of bench.py's line: the whole line stays below that, the BASELINE configs come LAST, and the
    keys the contract names are there (round 6: the effort sweep moved behind --sweep, into a file)."""
    import bench

    head = {"metric": "compress MiB/s + ratio vs zlib -6, 1 GiB synthetic; 1/2/4/8 GPU", "value": 210000.1, "unit": "MiB/s", "n_gpus": 8, "steps": 20,
            "warmup": 3, "ms_per_step": 4.876, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "1 GiB synthetic enwik-like text per GPU (gen_text_torch seed 3)", "effort": "default", "block_bytes": 524288,
                       "deflate_block_bytes": 32768, "window_bytes": 32768, "strategy": "auto", "container": "raw",
                       "parallelism": "shard8 block-cyclic x4, gather overlapped"},
            "ratio": 2.6294, "ratio_zlib6": 2.8349, "ratio_vs_zlib6": 0.9275, "compressed_bytes": 3266918912, "roundtrip_ok": True}
    kernel_ms = {k: 3.7612 for k in ("k_lz77", "k_plan", "k_scan", "k_emit", "k_checksum")}
    roofline = {"bound": "hbm", "kernel": "k_lz77", "achieved": 394.05, "peak": 8000.0, "unit": "GB/s", "frac": 0.04926, "traffic": 2328207360,
                "traffic_over_algorithmic": 1.571, "algorithmic_bytes_per_launch": 1482106740, "kernel_ms": 3.7612, "read_frac": 0.03568,
                "traffic_info": {"commit": "616a83f", "csrc_sha256": "4be98d6392ec4dae", "current": True, "read_bytes": 1225961472, "write_bytes": 1101819904},
                "issue": {"valu_wave_instructions_per_launch": 2262057677, "cycles_per_instruction_per_simd": 4.09, "lds_busy_frac": 0.562, "lds_bank_conflict_frac": 0.457},
                "device": {"name": "AMD Instinct MI355X", "arch": "gfx950:sramecc+:xnack-", "compute_units": 256, "clock_khz": 2400000, "hbm_peak_from_props_GBs": 8192.0}}
    cpu = {"value": 145.75, "unit": "MiB/s", "cores": 1, "kind": "port",
           "sample": "oracle sfo_decompress (restates reference src/decompress.cpp:402-461) on the GPU-made stream of the first 512 MiB of the workload; "
                     "output MiB/s; equal=True; host has 256 logical cores", "zlib6_compress_MiBps_1core": 30.61,
           "zlib6_compress_MiBps_block_parallel": {"value": 327.1, "threads": 16, "ratio": 2.8311}}
    dk = {k: 2.6123 for k in ("k_inflate_tokens", "k_inflate_tokens_retry", "k_inflate_bytes", "k_inflate_status")}
    decomp = {lab: {"value": 187000.1, "unit": "MiB/s of output", "ms": 5.471, "status": 0, "equal_to_input": True, "kernel_ms": dk}
              for lab in ("sub_indexed", "segment_indexed")}
    decomp["zlib_made_segment_indexed"] = {"segments": 8192, "by_lane_serial_kernel": 0, "value": 139000.1, "unit": "MiB/s of output", "ms": 1.851, "bytes": 268435456,
                                           "status": 0, "equal_to_input": True, "kernel_ms": dk}
    e2e = {"value": 49900.1, "unit": "MiB/s", "ms": 20.512, "bytes_out": 408364916, "what": "sfh_compress, pinned host buffers"}
    real = {k: bench.compact_workload(_mock_workload(k)) for k in ("real_source", "real_source_effort_recent_all", "real_binary", "real_binary_effort_recent_all")}
    devices = [f"AMD Instinct MI355X pci 0000:{b:02x}:00 uuid GPU-0123456789abcdef0123456789abcdef" for b in range(8)]
    mg = {"ranks_seen": 8, "distinct_devices": 8, "devices": devices, "gather_ms_per_round": [1.2345] * 4, "rounds": 4, "gather_ms_per_step": 4.938,
          "prediction": bench.predict_scaling(8, 4, 4.9, 408364916)}
    configs = {"config2_text": {"workload": "1 GiB synthetic enwik-like text per GPU", "value": 210000.1, "unit": "MiB/s", "ms": 4.876, "n_gpus": 8,
                                "ratio_vs_zlib6": 0.9275, "roundtrip_ok": True, "roofline_frac": 0.04926, "roofline_read_frac": 0.03568},
               "config3_mixed_1gpu": bench.compact_workload(_mock_workload("mixed"), 256 << 20, "N+C over k_lz77"),
               "config4_random": bench.compact_workload(_mock_workload("high-entropy"), 256 << 20, "2N+5/chunk over the path")}
    line = bench.assemble_line(head, kernel_ms, roofline, cpu, e2e, decomp, real, mg, configs, {"runs": bench.compact_workload(_mock_workload("runs"))})
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT, len(text)
    assert list(line)[-1] == "configs" and set(line["configs"]) == {"config2_text", "config3_mixed_1gpu", "config4_random"}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "decompress", "multi_gpu"):
        assert k in line, k
    c4 = line["configs"]["config4_random"]
    assert {"value", "ratio_vs_zlib6", "roundtrip_ok", "roofline", "kernel_ms"} <= set(c4) and c4["roofline"]["basis"].startswith("2N")
    assert c4["roofline"]["bytes"] == 2 * (256 << 20) + 5 * 8192 and 0 < c4["roofline"]["frac"] < 1
    assert line["configs"]["config3_mixed_1gpu"]["roofline"]["basis"].startswith("N+C")


@pytest.mark.gpu
def test_bench_line_on_the_gpu_is_compact(tmp_path):
    """The real command at a reduced size: one line under the driver's 8 KB, `configs` last with config[3]'s bytes and config[4], and --sweep
    writes its table to a file instead of the line."""
    sweep = tmp_path / "sweep.json"
    r = _run("--bytes", str(64 << 20), "--secondary-bytes", str(32 << 20), "--cpu-sample-bytes", str(16 << 20), "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 8192, len(lines[0])
    d = json.loads(lines[0])
    assert list(d)[-1] == "configs" and {"config2_text", "config3_mixed_1gpu", "config4_random"} <= set(d["configs"])
    assert d["roundtrip_ok"] is True and all(c["roundtrip_ok"] for c in d["configs"].values())
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0 and d["decompress"]["sub_indexed"]["equal_to_input"]
    assert d["configs"]["config4_random"]["roofline"]["frac"] > 0.05 and d["configs"]["config4_random"]["ratio"] < 1.001
    assert "real_source" in d["real_bytes"] and "real_binary_effort_recent_all" in d["real_bytes"]
    r = _run("--bytes", str(32 << 20), "--secondary-bytes", str(32 << 20), "--no-cpu-baseline", "--no-decompress", "--steps", "1", "--warmup", "1", "--sweep", str(sweep))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip())
    assert len(r.stdout.strip()) < 8192 and d["sweep_file"] == str(sweep) and list(d)[-1] == "configs"
    sw = json.load(open(sweep))
    assert {"text_effort_best", "mixed_effort_recent_all", "real_source_effort_extreme", "real_binary_effort_max"} <= set(sw["sweep"])
