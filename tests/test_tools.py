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


def test_isa_table_of_the_match_kernel(tmp_path):
    """tools/isa_hist.py (round 6): the per-phase instruction table of the default k_lz77 -- the line tables it compiles with
    leave the code as it is, every phase of the source's `@phase` markers gets instructions, and the dynamic total per wave-round
    reproduces the counter pass it is quoted beside (SQ_INSTS_VALU = 1,078 per wave-round, profiles/r06_pmc_summary.json) to 15 %."""
    import json
    import sys

    out = tmp_path / "isa.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hist.py"), "--pmc-valu-per-wave-round", "1078", "--json", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "** DIFFER **" not in r.stdout
    d = json.load(open(out))
    rows = {x["phase"]: x for x in d["rows"]}
    for ph in ("stage", "match.first", "match.second", "match.insert", "parse.take", "parse.transfer", "parse.reconcile", "parse.counts",
               "emit.literals", "emit.matches"):
        assert rows[ph]["static_valu"] > 0 and rows[ph]["dyn_valu"] > 0, ph
    tot = d["total"]
    assert 0.9 < tot["dyn_valu"] / 1078 < 1.15, tot
    assert 2.8 < tot["dyn_valu_cycles"] / tot["dyn_valu"] < 3.8  # the mix: half the instructions in the 2.3-cycle class
    match = sum(v["dyn_valu_cycles"] for k, v in rows.items() if k.startswith("match"))
    assert 0.4 < match / tot["dyn_valu_cycles"] < 0.6  # the match phase is about half of the kernel's vector issue
