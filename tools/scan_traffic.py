#!/usr/bin/env python3
"""tools/scan_traffic.py <FETCH_SIZE summary csv> <WRITE_SIZE summary csv> <config>_b<batch> <profile tag> [kernel substring]
-- HBM bytes per launch of the scan kernel from the two PMC passes of tools/prof_round.sh (summaries written by
profiles/summarize_pmc.py), merged into profiles/scan_traffic.json with the sha of the sources the profile was taken from
(what bench.py's roofline.traffic quotes, and marks stale when the tree has moved on).

bytes = FETCH_SIZE * 1024 * 2 (gfx950 reports wide streaming reads at half their bytes: MI355X_MICROARCH.md, HBM / rocprofv3)
      + WRITE_SIZE * 1024"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sketchy_amd.build import source_sha  # noqa: E402

fetch_csv, write_csv, key, tag = sys.argv[1:5]
pat = sys.argv[5] if len(sys.argv) > 5 else "scan_lean_kernel"


def mean_of(path, counter):
    best = None
    with open(path) as f:
        for r in csv.reader(f):
            if len(r) >= 4 and pat in r[0] and r[1] == counter:
                cand = (int(r[2]), float(r[3]), r[0])
                if best is None or cand[0] > best[0]:
                    best = cand
    if best is None:
        raise SystemExit(f"no {counter} row for {pat} in {path}")
    return best


nf, fetch, kname = mean_of(fetch_csv, "FETCH_SIZE")
nw, write, _ = mean_of(write_csv, "WRITE_SIZE")
entry = {"FETCH_SIZE_mean": fetch, "WRITE_SIZE_mean": write, "bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
         "kernel": kname.replace("void ", "").replace("skx::", "").strip('"').split("(")[0], "launches_profiled": min(nf, nw),
         "source_sha": source_sha(),
         "profile": f"profiles/{tag}_pmc_FETCH_SIZE.csv + {tag}_pmc_WRITE_SIZE.csv (tools/prof_round.sh {tag}; rocprofv3 --pmc, separate "
                    f"passes, mean of {min(nf, nw)} launches -- a launch serves up to eight enqueued batches)",
         "note": "bytes = FETCH_SIZE*1024*2 (gfx950 reports wide streaming reads at half their bytes) + WRITE_SIZE*1024"}
path = os.path.join(ROOT, "profiles", "scan_traffic.json")
with open(path) as f:
    allv = json.load(f)
allv[key] = entry
with open(path, "w") as f:
    json.dump(allv, f, indent=1)
print(key, json.dumps(entry))
