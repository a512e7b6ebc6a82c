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
    for name in ("r06_z_bench_default_detail.json", "r05_z_bench_default_detail.json"):  # (the newest committed full document)
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            return json.load(open(path))
    pytest.skip("no committed bench detail document")


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


def test_emit_always_prints_a_line_under_the_limit(tmp_path, capsys):
    """emit() asserted on the line's length AFTER the whole benchmark had run — parts compact_line never trims
    (config, roofline, cpu_baseline, multi_gpu) could cost the one line the driver parses.  Now it sheds them instead."""
    d = _detail()
    d["config"] = dict(d["config"], **{f"extra_{i}": "x" * 90 for i in range(40)})
    d["roofline"] = dict(d["roofline"], lds_gather={k: 1.0 for k in ("achieved", "peak", "frac")}, stage_us_per_step={f"s{i}": 1.0 for i in range(200)})
    d["multi_gpu"] = {"rccl_ranks": 8, "coarse": "y" * 3000}
    legs.emit(d, root=str(tmp_path))
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and len(out[0]) < legs.LINE_LIMIT
    line = json.loads(out[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in line, k
    assert os.path.exists(tmp_path / "bench_detail.json")


def test_bench_with_gpus_2_without_a_launcher_says_what_to_run():
    """`python bench.py --gpus 2` outside torch.distributed.run must stop at once with the launch line, not hang in a rendezvous
    or fail somewhere inside torch (SURVEY.md section 8e preflight; runs without a GPU: the check comes before any device work)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    msg = r.stderr + r.stdout
    assert "torch.distributed.run" in msg and "--nproc-per-node 2" in msg and "WORLD_SIZE" in msg, msg[-1500:]
