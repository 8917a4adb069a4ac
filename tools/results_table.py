#!/usr/bin/env python3
"""tools/results_table.py <bench_driver.json> <bench_c4.json>  -- the round's results as a markdown table from the
bench lines themselves: no number of that table is typed by hand."""
import json
import sys

d = json.load(open(sys.argv[1]))
c = json.load(open(sys.argv[2]))
M = lambda v: "n/a" if v is None else f"{v / 1e6:.1f} M"
g = lambda o, *ks: (lambda x: x)(__import__("functools").reduce(lambda a, k: (a or {}).get(k) if isinstance(a, dict) else None, ks, o))
r, a = d["roofline"], d.get("value_ancestor") or {}
rows = [
    ("C2", "**`value`** -- SURVEY §8(d)'s truth-strain stream: 20 distinct batches of 98 304 reads from a fresh table, `skx_stream_enqueue_device`, groups of 6 + 8 + 6 "
           "batches share a scan; median of 7 repetitions",
     f"**{M(d['value'])}** ({' / '.join(f'{v / 1e6:.1f}' for v in d['values_all'])})", f"{d['ms_per_step']:.3f}",
     f"`scan_lean_kernel<0, 6, false>` {r['avg_launch_ms']:.3f} ms beside the sketches = **{r['frac']:.3f} of 8 TB/s**; alone {g(r, 'isolated', 'avg_launch_ms'):.3f} ms = "
     f"**{g(r, 'isolated', 'frac'):.3f}**; traffic {('%.2f x' % (r['traffic'] / r['algorithmic_bytes_per_launch'])) if r.get('traffic') else 'n/a'}; "
     f"`roofline_valu.frac` {g(d, 'roofline_valu', 'frac') and round(g(d, 'roofline_valu', 'frac'), 2)}",
     f"{g(d, 'cpu_baseline', 'value'):.1f} reads/s (1 thread), {g(d, 'cpu_baseline_all_cores', 'value') or 0:.0f} ({g(d, 'cpu_baseline_all_cores', 'cores')} threads)",
     f"every row of the first and last step + table = oracle: {g(d, 'oracle_whole_steps', 'timed_rows_match_oracle')} / {g(d, 'oracle_whole_steps', 'final_table_matches_oracle')}"),
    ("C2", "**`value_ancestor`** (rounds 1-4's near-tie stream, its own child run)", f"**{M(a.get('value'))}**", f"{a.get('ms_per_step', 0):.3f}",
     f"{g(a, 'scan', 'avg_launch_ms') or 0:.3f} ms in the pipeline = {g(a, 'scan', 'frac') or 0:.3f}", "", f"oracle: {g(a, 'oracle_whole_steps', 'timed_rows_match_oracle')}"),
    ("C2", "`value_steady_state` / `value_cold` / `value_batch_x2` / `value_one_pass_per_batch` / `value_other_api` (truth strain)",
     " / ".join(M(g(d, k, "value")) for k in ("value_steady_state", "value_cold", "value_batch_x2", "value_one_pass_per_batch", "value_other_api")), "", "", "", "rows = 4 096-read cuts / the timed run's"),
    ("C2", "`value_membership_reused` (policy `reuse_membership`: the static dense rows scanned for once and kept -- a side leg), 20 batches / a lone batch",
     f"{M(g(d, 'value_membership_reused', 'value'))} / {M(g(d, 'value_membership_reused', 'value_cold'))}", f"{g(d, 'value_membership_reused', 'ms_per_step') or 0:.3f}", "no scan in the timed region", "",
     f"rows = the timed run's: {g(d, 'value_membership_reused', 'rows_match_timed_run')}"),
    ("C2", "`value_top16` (sixteen rows after every read, `-t 16`, truth strain; all K and both workloads: `profiles/r06_topk.txt`), 20 batches from a fresh table",
     f"{M(g(d, 'value_top16', 'value'))}", f"{g(d, 'value_top16', 'ms_per_step') or 0:.3f}", "", "",
     f"first row = the timed run's: {g(d, 'value_top16', 'first_row_matches_timed_run')}"),
    ("C2", "`value_host_fed` / `value_host_fed_packed` (PCIe-bound)", f"{M(g(d, 'value_host_fed', 'value'))} / {M(g(d, 'value_host_fed_packed', 'value'))}", "", "", "", ""),
    ("C2", "**`value_end_to_end`**: `sketchy-hip predict -s` on a 4.7 GB FASTQ file (host-bound: 16 usable CPUs of a shared box)",
     f"**{M(g(d, 'value_end_to_end', 'value'))}** (runs {' / '.join(f'{v / 1e6:.1f}' for v in (g(d, 'value_end_to_end', 'runs_reads_per_s') or []))})", "", "", "",
     f"rows = the device path's: {g(d, 'value_end_to_end', 'rows_match_device_path')}"),
    ("C2", "`value_end_to_end_gz`: the file's first four batches as BGZF (all threads inflate) / plain gzip (one thread); with the stream's set-up counted",
     f"{M(g(d, 'value_end_to_end_gz', 'bgzf', 'value'))} / {M(g(d, 'value_end_to_end_gz', 'plain_gzip', 'value'))}; {M(g(d, 'value_end_to_end_gz', 'bgzf', 'value_with_setup'))} / "
     f"{M(g(d, 'value_end_to_end_gz', 'plain_gzip', 'value_with_setup'))}", "", "", "", f"rows = the device path's: {g(d, 'value_end_to_end_gz', 'bgzf', 'rows_match_device_path')}"),
]
rc, ac = c["roofline"], c.get("value_ancestor") or {}
rows += [
    ("C4", "5 species resident (150 000 genomes, 12.1 GB per scan), log-normal reads mixed over the five species' TRUTH strains, 8 batches from a fresh table (`value` of the c4 line)",
     f"**{M(c['value'])}** (steady {M(g(c, 'value_steady_state', 'value'))}, cold {M(g(c, 'value_cold', 'value'))})", f"{c['ms_per_step']:.2f}",
     f"{rc.get('kernel')} {rc['avg_launch_ms']:.2f} ms = {rc['frac']:.3f}, alone {g(rc, 'isolated', 'frac') or 0:.3f}; {rc['launches_per_step']:.2f} scans per step", "", f"oracle: {g(c, 'oracle_whole_steps', 'timed_rows_match_oracle')}"),
    ("C4", "the same mixed over the five species' ANCESTORS (`value_ancestor` of the c4 line: round 5's C4 stream)", f"**{M(ac.get('value'))}**", f"{ac.get('ms_per_step', 0):.2f}",
     f"{g(ac, 'scan', 'avg_launch_ms') or 0:.2f} ms = {g(ac, 'scan', 'frac') or 0:.3f}", "", f"oracle: {g(ac, 'oracle_whole_steps', 'timed_rows_match_oracle')}"),
]
print("| cfg | what | reads/s | ms/step | scan kernel | CPU port (same box) | parity |\n|---|---|---|---|---|---|---|")
for row in rows:
    print("| " + " | ".join(str(x) for x in row) + " |")
