#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection / kernel_trace CSVs per sf:: kernel.
usage: pmc_summary.py <dir> [<dir> ...]   (searches for *_counter_collection.csv / *_kernel_trace.csv)"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("sf::", "")
    return name.split("(")[0]


def main():
    out = {}
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(float))
            disp = collections.defaultdict(set)
            for r in csv.DictReader(open(f)):
                if "sf::" not in r["Kernel_Name"]:
                    continue
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k].add(r["Dispatch_Id"])
            for k, v in acc.items():
                e = out.setdefault(k, {})
                e.setdefault("pmc_dispatches", len(disp[k]))
                for c, x in v.items():
                    e[c] = x
        for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
            dur = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "sf::" not in r["Kernel_Name"]:
                    continue
                dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            for k, v in dur.items():
                e = out.setdefault(k, {})
                e["trace_calls"] = len(v)
                e["trace_avg_us"] = sum(v) / len(v) / 1e3
                e["trace_max_us"] = max(v) / 1e3
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
