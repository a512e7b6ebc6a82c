"""The driver parses ONE JSON line from bench.py's stdout; round 4's grew to 29.7 KB and was not parsed (BENCH_r04.parsed = null).
bench_legs.compact_line keeps the contract's keys + roofline + cpu_baseline + summary under 4 KB whatever the secondary legs
add; the full document goes to bench_detail.json.  CPU-only: the committed full document of the end-of-round run is the input."""
import json
import os

import pytest

import bench_legs as legs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline", "summary")


def _detail():
    path = os.path.join(ROOT, "profiles", "r05_z_bench_default_detail.json")
    if not os.path.exists(path):
        pytest.skip("no committed bench detail document")
    return json.load(open(path))


def test_compact_line_is_small_and_complete():
    d = _detail()
    assert len(json.dumps(d)) > 20000  # (the document really is too long for one line)
    line = legs.compact_line(d)
    s = json.dumps(line)
    assert len(s) < legs.LINE_LIMIT
    assert list(line)[-1] == "summary"
    for k in REQUIRED:
        assert k in line, k
    assert line["metric"] == d["metric"] and line["unit"] == "queries/s" and line["config"]["workload"] == d["config"]["workload"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "us_per_launch", "algorithmic_bytes_per_launch"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and "sample" in c and c["parity"]["rowids_bit_exact"] is True
    assert "restatement" in c["parity_caveat"] and "lance-index" in c["parity_caveat"]  # the caveat travels with every "bit-exact"


def test_compact_line_sheds_summary_tables_before_it_grows_past_the_limit():
    d = _detail()
    d["summary"] = dict(d["summary"])
    d["summary"]["qps_vs_batch"] = {str(i): i * 1000 for i in range(400)}  # a leg that reports far too much
    d["config"] = dict(d["config"], workload=d["config"]["workload"] + "_x" * 200)
    s = json.dumps(legs.compact_line(d))
    assert len(s) < legs.LINE_LIMIT
    assert "c3_qps" in json.loads(s)["summary"]
