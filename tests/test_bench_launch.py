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
