"""Development tooling that guards the GPU box: the --pmc runner refuses counter sets that do not fit one pass (round 3: two
TCC-derived counters made rocprofv3 abort and hang), and measurements are stamped with the code they measured."""
import os
import subprocess

from conftest import ROOT


def _pmc(counters, *prog):
    return subprocess.run(["bash", os.path.join(ROOT, "tools", "pmc_run.sh"), "/tmp/sf_pmc_guard_test", "5", counters, "--", *prog],
                          capture_output=True, text=True, timeout=60)


def test_pmc_runner_refuses_what_hangs_the_profiler():
    r = _pmc("FETCH_SIZE WRITE_SIZE", "python", "-c", "1")
    assert r.returncode == 64 and "TCC slots 5/4" in r.stderr
    r = _pmc(" ".join(f"SQ_X{k}" for k in range(9)), "python", "-c", "1")
    assert r.returncode == 64 and "SQ slots 9/8" in r.stderr
    for hop in ("env", "bash", "taskset"):
        r = _pmc("FETCH_SIZE", hop, "python")
        assert r.returncode == 64 and "must follow -- directly" in r.stderr


def test_every_pmc_script_goes_through_the_guard():
    """No script under tools/ starts `rocprofv3 --pmc` by itself."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "tools")):
        for f in files:
            p = os.path.join(d, f)
            if not f.endswith((".sh", ".py")) or f == "pmc_run.sh":
                continue
            with open(p, errors="replace") as fh:
                if "rocprofv3 --pmc" in fh.read():
                    bad.append(os.path.relpath(p, ROOT))
    assert not bad, bad


def test_source_stamp():
    from starflate_amd.build import source_stamp

    a, b = source_stamp(), source_stamp()
    assert a == b and len(a["csrc_sha256"]) == 16 and a["commit"]
