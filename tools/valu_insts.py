#!/usr/bin/env python3
"""tools/valu_insts.py <pmc_all_summary.csv> <config>_b<batch> <profile tag>  -- VALU wave instructions one bench step
issues, per kernel of this library, from the per-kernel SQ_INSTS_VALU means of tools/pmc_all.sh; merges the entry into
profiles/valu_insts.json (what bench.py's roofline_valu block quotes).

A step launches the main sketch kernel (sketch_wave_kernel<k, HCAP, true> with the smallest HCAP) exactly once, so a kernel's launches per step = its dispatches / that
kernel's dispatches (set-up kernels -- ref_tile, band_bounds, filter_build, the rare-hash index's -- are left out: they run once per reference)."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sketchy_amd.build import source_sha  # noqa: E402
SETUP = ("ref_tile_kernel", "band_bounds_kernel", "filter_build_kernel", "rare_count_kernel", "rare_fill_kernel", "mlong_build_kernel",
         "mlong_transpose_kernel", "list_sig_kernel", "pat_exceptions_kernel", "pat_matrix_kernel",
         "collect_dense_kernel")  # (the rare-hash index, the patterns and the static dictionary are built with the reference)
src, key, tag = sys.argv[1], sys.argv[2], sys.argv[3]
sha = sys.argv[4] if len(sys.argv) > 4 else source_sha()   # (argv[4]: re-deriving the JSON from a CSV of an earlier tree)
rows = {}
with open(src) as f:
    for r in csv.DictReader(f):
        if r["counter"] != "SQ_INSTS_VALU" or "skx::" not in r["kernel"]:
            continue
        name = re.sub(r".*skx::", "", r["kernel"]).strip().strip('"')
        if any(name.startswith(s) for s in SETUP):
            continue
        rows[name] = (int(r["dispatches"]), float(r["total_value"]))
# the main sketch kernel = the INRANGE instance with the smallest hash buffer (HCAP 128 since round 4, 256 before)
main = min((name for name in rows if re.match(r"sketch_wave_kernel<\d+, \d+, true>", name)),
           key=lambda name: int(name.split(",")[1]))
steps = rows[main][0]
per_kernel = {name: round(tot / steps) for name, (n, tot) in sorted(rows.items(), key=lambda kv: -kv[1][1])}
entry = {"wave_insts_per_step": int(sum(per_kernel.values())), "steps_profiled": steps, "per_kernel": per_kernel,
         "source_sha": sha,  # (of the tree the profile was taken from: run this on the box, right behind the PMC pass)
         "profile": f"profiles/{tag}_insts_per_kernel.csv (tools/pmc_all.sh: rocprofv3 --pmc SQ_INSTS_VALU ..., per-kernel totals / steps)"}
path = os.path.join(ROOT, "profiles", "valu_insts.json")
try:
    with open(path) as f:
        allv = json.load(f)
except (OSError, ValueError):
    allv = {}
allv[key] = entry
with open(path, "w") as f:
    json.dump(allv, f, indent=1)
print(key, entry["wave_insts_per_step"], json.dumps(per_kernel))
