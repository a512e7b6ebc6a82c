"""bench.py end to end at toy sizes: every leg of the default line runs, the parity flags of every leg are true and the line ends
with the compact `summary` — so that a leg cannot rot unnoticed between the rounds' full-size runs (the driver's bench run is the
only other place the whole file executes).  One subprocess (bench.py uses torch for device memory and the RNG)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_at_toy_sizes():
    pytest.importorskip("torch")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--n-rows", "4000000", "--nlist", "512", "--batch", "256", "--steps", "4", "--warmup", "1",
           "--c4-rows", "30000000", "--c5-rows", "2000000", "--gist-rows", "150000", "--recall-rows", "150000", "--recall-queries", "500",
           "--recall2-rows", "100000", "--recall2-queries", "300", "--recall-iters", "4", "--flat-rows", "600000", "--flat-batch", "256",
           "--loopback-world", "4", "--cpu-seconds", "3"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]  # ONE line on stdout, nothing else
    line = lines[0]
    assert len(line) < 4096, len(line)  # round 4's 29.7 KB line was not parsed by the driver
    head = json.loads(line)
    assert list(head)[-1] == "summary"
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in head, key
    assert head["roofline"]["frac"] > 0 and head["roofline"]["bound"] == "hbm" and head["roofline"]["us_per_launch"] > 0
    assert head["cpu_baseline"]["kind"] == "port" and head["cpu_baseline"]["parity"]["rowids_bit_exact"] and "sample" in head["cpu_baseline"]
    with open(os.path.join(ROOT, "bench_detail.json")) as f:
        d = json.load(f)  # the full document
    assert d["value"] == pytest.approx(head["value"], rel=1e-6) and d["summary"]["c3_qps"] == head["summary"]["c3_qps"]
    assert d["unit"] == "queries/s" and d["value"] > 0 and d["roofline"]["frac"] > 0 and d["config"]["scan_variant"] == 2
    assert d["cpu_baseline"]["parity"]["rowids_bit_exact"] and d["cpu_baseline"]["cores"] >= 1
    sec = d["secondary"]
    c4 = sec["c4"]
    assert c4["cpu_baseline"]["parity"]["rowids_bit_exact"] and c4["cpu_baseline"]["parity"]["distances_equal"]
    assert c4["cpu_baseline"]["parity"]["oracle_probed_only_copied_partitions"]
    assert c4["loopback_world4"]["every_rank_equals_unsharded"]
    assert sec["loopback_world4"]["overlapped"]["every_rank_equals_unsharded"] and sec["loopback_world4"]["serial"]["every_rank_equals_unsharded"]
    for key in ("c3_shape_dim384_m24", "c3_shape_dim3072_m192"):
        assert sec[key]["scan_variant"] == 2 and sec[key]["value"] > 0
    assert sec["c3_shape_dim768_m96_pq4"]["scan_variant"] == 2 and sec["c3_shape_dim768_m96_pq4"]["value"] > 0
    assert [p["batch"] for p in sec["qps_vs_batch"]] == [1, 8, 64, 256, 512, 1024, 2048]
    assert sec["concurrent_callers_c3"]["coalesced_64_threads"]["queries_per_s"] > 0
    assert sec["gist_like"]["scan_variant"] == 2 and len(sec["gist_like"]["points"]) == 8
    assert sec["gist_like"]["nprobes50_refine30_vs_oracle_256_queries"]["rowids_bit_exact"]
    assert sec["c5_refine10"]["cpu_baseline"]["parity"]["rowids_bit_exact"]
    assert sec["c1_flat"]["cpu_baseline"]["parity"]["rowids_bit_exact"] and sec["c1_flat"]["cpu_baseline"]["parity"]["distances_equal"]
    for key in ("flat_c2_l2", "flat_c2_cosine"):
        assert sec[key]["cpu_baseline"]["parity"]["rowids_bit_exact"]
    rec = d["recall_at_10"]
    assert all(v for k2, v in rec.items() if k2.endswith("_rowids_bit_exact"))
    assert d["summary"]["c4_parity_ids"] is True and d["summary"]["c3_parity_ids"] is True


def test_sharded_bench_path_in_a_world_of_one():
    """Preflight of the driver's N > 1 run on the one GPU a test box has: `bench.py --gpus 1 --force-sharded-path`
    under torch.distributed.run goes through everything an N-rank run goes through — process group, RCCL communicator behind the
    C ABI, mi355_search_sharded with its packed all-gather and k-way merge, teardown — and must print ONE JSON line whose
    sharded result equals the plain search.  The first 8-GPU run cannot then fail on plumbing."""
    pytest.importorskip("torch")
    for extra in ([], ["--shard-coarse", "--batch-per-gpu", "128"]):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29581", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-sharded-path", "--n-rows", "3000000",
               "--nlist", "512", "--batch", "256", "--steps", "3", "--warmup", "1"] + extra
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]  # (RCCL's banner goes to stderr: stdout carries the one line)
        d = json.loads(lines[0])
        assert len(lines[0]) < 4096 and d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "strong"
        mg = d["multi_gpu"]
        assert mg["rccl_ranks"] == 1 and mg["sharded_equals_unsharded"] is True and mg["all_ranks_returned_the_same_results"] is True
        assert mg["coarse"].startswith("sharded" if extra else "replicated")
