#!/usr/bin/env python3
"""Condense a bench.py JSON line (stdin) to one short line: tools/bench_line.py [label]"""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get("roofline", {})
st = {k: round(v, 3) for k, v in d.get("stage_ms_per_step", {}).items()}
print(" ".join(sys.argv[1:]), "B=%d" % d["config"]["reads_per_step"], "reads/s=%d" % d["value"],
      "ms/step=%.3f" % d["ms_per_step"], "scan_ms=%.4f" % r.get("avg_launch_ms", 0), "frac=%.3f" % r.get("frac", 0), "alone=%.3f" % r.get("isolated", {}).get("frac", 0),
      "launches/step=%.2f" % r.get("launches_per_step", 0), "cold=%d" % d.get("value_cold", {}).get("value", 0),
      "steady=%d" % d.get("value_steady_state", {}).get("value", 0), "x2=%d" % d.get("value_batch_x2", {}).get("value", 0), "1pass=%d" % d.get("value_one_pass_per_batch", {}).get("value", 0), "hostfed=%d" % d.get("value_host_fed", {}).get("value", 0), "packed=%d" % d.get("value_host_fed_packed", {}).get("value", 0), "e2e=%d" % d.get("value_end_to_end", {}).get("value", 0), "other_workload=%d" % (d.get("value_truth_strain") or d.get("value_ancestor") or {}).get("value", 0),
      "oracle=%s" % {k: (d.get("oracle_whole_steps") or {}).get(k) for k in ("timed_rows_match_oracle", "final_table_matches_oracle", "seconds")},
      "parity=%s" % d.get("parity"), d.get("parity_error", ""), st)
