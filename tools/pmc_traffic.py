#!/usr/bin/env python3
"""profiles/<tag>_pmc_summary.json -> profiles/pmc_traffic.json (HBM bytes per launch per kernel).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide
(16 B/lane) coalesced reads (MI355X_MICROARCH.md, HBM section), so reads are doubled.
Every dispatch of the profiled command processes the same 1 GiB, so per launch = sum / dispatches."""
import json
import sys

import subprocess

src, dst = sys.argv[1], sys.argv[2]
d = json.load(open(src))
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
except Exception:  # noqa: BLE001  (the GPU box has no .git: the caller fills it in)
    commit = "unknown"
out = {"_meta": {"commit": commit, "source": src, "command": "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-secondary"}}
for name, v in d.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    k = name.replace("void ", "").split("<")[0]
    n = v.get("pmc_dispatches", 1)
    rd = 2 * v["FETCH_SIZE"] * 1024 / n
    wr = v["WRITE_SIZE"] * 1024 / n
    out[k] = {"hbm_bytes_per_launch": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr), "dispatches": n,
              "note": "reads = 2 x FETCH_SIZE KiB (gfx950 wide-read correction), writes = WRITE_SIZE KiB"}
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
