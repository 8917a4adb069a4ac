"""The bench line's contract, checked on the committed lines of the last profile round (no GPU needed): the keys the driver and
the judge read, the roofline / cpu_baseline objects, and that the quoted PMC figures belong to the committed sources."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_line(path):
    with open(path) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def _latest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    assert files, pattern
    return files[-1]


@pytest.mark.parametrize("pattern", ["r0*_bench_driver.json", "r0*_bench_c4.json"])
def test_committed_bench_line_has_the_contract_keys(pattern):
    d = _last_line(_latest(pattern))
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d.get("parity_error") is None and all(v is True for v in d["parity"].values()), d["parity"]
    # value = whole-job reads / the median repetition's time: reads per step / ms_per_step
    assert abs(d["value"] - d["config"]["reads_per_step"] * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert sorted(d["values_all"])[len(d["values_all"]) // 2] == d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch / the kernel's average launch time, measured in the run
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["timed_rows_match_oracle"] is True and c["sample"]


def test_quoted_profile_figures_belong_to_the_committed_sources():
    """profiles/scan_traffic.json and valu_insts.json (what bench.py quotes as roofline.traffic / roofline_valu) carry the sha of the
    sources they were measured on: the round's last profile run must be of the committed tree."""
    sys.path.insert(0, ROOT)
    from sketchy_amd.build import source_sha
    sha = source_sha()
    with open(os.path.join(ROOT, "profiles", "scan_traffic.json")) as f:
        traffic = json.load(f)
    with open(os.path.join(ROOT, "profiles", "valu_insts.json")) as f:
        insts = json.load(f)
    stale = [key for key in ("c2_b98304", "c4_b98304") if traffic[key].get("source_sha") != sha or insts[key].get("source_sha") != sha]
    if stale:  # (mid-round state: bench.py flags the same figures as stale in its line; not a failure of the code under test)
        pytest.skip(f"profile figures of {stale} were measured on other sources: re-run tools/prof_round.sh / tools/pmc_round.sh")
    for key in ("c2_b98304", "c4_b98304"):
        assert len(traffic[key]["source_sha"]) == 16 and traffic[key]["bytes_per_launch"] > 0 and insts[key]["wave_insts_per_step"] > 0


def test_valu_insts_tool_follows_the_main_sketch_kernel(tmp_path):
    """tools/valu_insts.py takes its per-step divisor from the main sketch kernel = the INRANGE instance with the smallest hash
    buffer, whatever that buffer's size is in this round (256 until round 3, 128 since: the tool once looked for '256')."""
    csv = tmp_path / "pmc.csv"
    csv.write_text('kernel,counter,dispatches,mean_value,total_value\n'
                   '"void skx::sketch_wave_kernel<16, 64, true>",SQ_INSTS_VALU,10,100.0,1000.0\n'
                   '"void skx::sketch_wave_kernel<16, 2048, true>",SQ_INSTS_VALU,10,1.0,10.0\n'
                   '"void skx::scan_lean_kernel<0, 6, false>",SQ_INSTS_VALU,2,500.0,1000.0\n'
                   '"skx::ref_tile_kernel",SQ_INSTS_VALU,3,5.0,15.0\n'
                   '"void at::native::something",SQ_INSTS_VALU,7,5.0,35.0\n')
    keep = os.path.join(ROOT, "profiles", "valu_insts.json")
    with open(keep) as f:
        before = f.read()
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "valu_insts.py"), str(csv), "test_key", "test_tag", "feedbeef"],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        with open(keep) as f:
            e = json.load(f)["test_key"]
        assert e["steps_profiled"] == 10 and e["source_sha"] == "feedbeef"
        assert e["per_kernel"] == {"sketch_wave_kernel<16, 64, true>": 100, "scan_lean_kernel<0, 6, false>": 100, "sketch_wave_kernel<16, 2048, true>": 1}
        assert e["wave_insts_per_step"] == 201
    finally:
        with open(keep, "w") as f:
            f.write(before)
