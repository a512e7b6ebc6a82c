"""Host logic of lancedb_amd/build.py (the mirror of the reference's IvfPqIndexBuilder,
rust/lancedb/src/index/vector.rs:61-119, :306-319) on a CPU-only box: the three device
entry points are stood in for by the CPU oracle (test infrastructure; the GPU runs of the
same builder are in tests/test_gpu_train.py), so what is under test is the sampling, the
seeding, the shapes handed to the C ABI and the quality of the resulting index."""
import os

import numpy as np
import pytest

import lancedb_amd
from lancedb_amd import _abi
from lancedb_amd import build as build_mod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def oracle_backend(oracle, monkeypatch):
    calls = []

    def kmeans_train(vectors, init, metric="l2", iters=50, cols=None, device=0):
        calls.append(("kmeans", vectors.shape, init.shape, metric, iters, cols))
        return oracle.kmeans_train(vectors, init, metric, iters, cols=cols)

    def ivf_residuals(vectors, centroids, metric="l2", device=0):
        calls.append(("residuals", vectors.shape, centroids.shape, metric))
        return oracle.ivf_residuals(vectors, centroids, metric)

    def ivfpq_encode(vectors, centroids, codebook, metric="l2", device=0, return_assign=False, nbits=8):
        calls.append(("encode", vectors.shape, metric))
        po, codes, order, assign = oracle.ivfpq_encode(vectors, centroids, codebook, metric, nbits=nbits)
        return (po, codes, order, assign) if return_assign else (po, codes, order)

    def pq_train(residuals, init_codebook, metric="l2", iters=50, nbits=8, device=0):
        # the definition of mi355_pq_train: sub-quantiser j = kmeans_train on its column range
        m, ks, dsub = init_codebook.shape
        calls.append(("pq", residuals.shape, init_codebook.shape, metric, iters, nbits))
        out = np.empty_like(init_codebook)
        for j in range(m):
            out[j], _ = oracle.kmeans_train(residuals, init_codebook[j], "dot" if metric == "dot" else "l2", iters,
                                            cols=(j * dsub, (j + 1) * dsub))
        return out

    monkeypatch.setattr(build_mod, "pq_train", pq_train)
    monkeypatch.setattr(build_mod, "kmeans_train", kmeans_train)
    monkeypatch.setattr(build_mod, "ivf_residuals", ivf_residuals)
    monkeypatch.setattr(build_mod, "ivfpq_encode", ivfpq_encode)
    return calls


def _data(n=6000, dim=32, nc=24, seed=3):
    rng = np.random.default_rng(seed)
    cent = rng.normal(size=(nc, dim)).astype(np.float32) * 3
    return (cent[rng.integers(0, nc, size=n)] + rng.normal(size=(n, dim))).astype(np.float32)


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
def test_train_calls_and_shapes(oracle_backend, metric):
    x = _data()
    b = lancedb_amd.IvfPqBuilder(distance_type=metric, num_partitions=16, num_sub_vectors=8, sample_rate=8,
                                 max_iterations=3, seed=5)
    cent, cb = b.train(x)
    assert cent.shape == (16, 32) and cb.shape == (8, 256, 4) and cent.dtype == cb.dtype == np.float32
    kinds = [c[0] for c in oracle_backend]
    assert kinds == ["kmeans", "residuals", "pq"]
    # IVF: sample_rate * num_partitions rows; PQ: sample_rate * 256 rows, all 8 sub-quantisers in one call
    assert oracle_backend[0][1] == (8 * 16, 32) and oracle_backend[0][2] == (16, 32) and oracle_backend[0][4] == 3
    assert oracle_backend[1][1] == (8 * 256, 32)
    assert oracle_backend[2][1] == (8 * 256, 32) and oracle_backend[2][2] == (8, 256, 4)
    assert oracle_backend[2][3:] == (metric, 3, 8)  # the index metric: dot -> dot, else plain l2 on residuals
    if metric == "cosine":  # seeded with unit rows
        assert np.isfinite(cent).all()
    # same seed -> same index; another seed -> another sample
    cent2, cb2 = lancedb_amd.IvfPqBuilder(distance_type=metric, num_partitions=16, num_sub_vectors=8, sample_rate=8,
                                          max_iterations=3, seed=5).train(x)
    assert (cent2 == cent).all() and (cb2 == cb).all()
    cent3, _ = lancedb_amd.IvfPqBuilder(distance_type=metric, num_partitions=16, num_sub_vectors=8, sample_rate=8,
                                        max_iterations=3, seed=6).train(x)
    assert not (cent3 == cent).all()


def test_defaults_and_errors(oracle_backend):
    x = _data(n=3000, dim=16)
    b = lancedb_amd.IvfPqBuilder(max_iterations=2, sample_rate=4)
    cent, cb = b.train(x)
    assert cent.shape == (1, 16)                        # num_partitions = rows / 8192, at least 1
    cent_t, _ = lancedb_amd.IvfPqBuilder(max_iterations=2, sample_rate=4, target_partition_size=500).train(x)
    assert cent_t.shape == (6, 16)                      # target_partition_size = 500 -> 3000 / 500
    assert cb.shape == (1, 256, 16)                     # 16 % 16 == 0 -> dim / 16 sub-vectors
    with pytest.raises(ValueError, match="does not divide"):
        lancedb_amd.IvfPqBuilder(num_partitions=4, num_sub_vectors=5).train(x)
    with pytest.raises(ValueError, match="not enough rows"):
        lancedb_amd.IvfPqBuilder(num_partitions=4).train(x[:100])


def test_built_index_is_searchable_and_accurate(oracle, oracle_backend):
    """The arrays the builder produces open as an index whose refined top-10 matches exact search."""
    x = _data(n=8000, dim=32, nc=32, seed=11)
    b = lancedb_amd.IvfPqBuilder(num_partitions=32, num_sub_vectors=8, sample_rate=16, max_iterations=6)
    cent, cb = b.train(x)
    po, codes, order = build_mod.ivfpq_encode(x, cent, cb, metric="l2")
    order = order.astype(np.int64)
    assert sorted(order.tolist()) == list(range(8000)) and int(po[-1]) == 8000
    ox = oracle.OracleIndex(cent, cb, po, codes, order.astype(np.uint64), raw_vectors=x[order])
    q = x[:40] + np.float32(0.01)
    ids, _, cnt, st = ox.search(q, k=10, nprobe_min=8, nprobe_max=8, refine_factor=10)
    truth = oracle.flat_search(x, q, k=10, metric=_abi.METRIC_L2)[0]
    assert st == 0 and (cnt == 10).all()
    recall = np.mean([len(set(ids[i].tolist()) & set(truth[i].tolist())) / 10 for i in range(40)])
    assert recall > 0.9, recall
    assert (ids[:, 0] == np.arange(40)).all()  # every query's own row comes back first


def test_partition_defaults_follow_the_reference():
    """build_ivf_params (table/create_index.rs:66-84): num_partitions > target_partition_size >
    the default partition size, pinned at rows / 8192 by create_index.rs:733-795; 4-bit codes make
    the suggested num_sub_vectors even (create_index.rs:86-102)."""
    assert build_mod.num_partitions_for(2 * 8192) == 2          # the reference's own test
    assert build_mod.num_partitions_for(100_000_000) == 12207
    assert build_mod.num_partitions_for(5000) == 1
    assert build_mod.num_partitions_for(100_000, target_partition_size=1000) == 100
    assert build_mod.num_partitions_for(100_000, num_partitions=7, target_partition_size=1000) == 7
    assert build_mod.get_num_sub_vectors(None, 768) == 48 and build_mod.get_num_sub_vectors(None, 24) == 3
    assert build_mod.get_num_sub_vectors(None, 24, num_bits=4) == 4
    assert build_mod.get_num_sub_vectors(5, 40, num_bits=4) == 5  # an explicit value is passed through
    with pytest.raises(ValueError, match="even when num_bits is 4"):
        lancedb_amd.IvfPqBuilder(num_bits=4, num_sub_vectors=3)
    with pytest.raises(ValueError, match="num_bits"):
        lancedb_amd.IvfPqBuilder(num_bits=6)


def test_four_bit_builder_shapes(oracle_backend):
    x = _data(n=3000)
    b = lancedb_amd.IvfPqBuilder(num_partitions=8, num_sub_vectors=8, num_bits=4, sample_rate=16, max_iterations=2)
    cent, cb = b.train(x)
    assert cb.shape == (8, 16, 4)
    assert oracle_backend[1][1] == (16 * 16, 32)  # PQ sample: sample_rate * 2^num_bits rows
    assert oracle_backend[2][2] == (8, 16, 4) and oracle_backend[2][5] == 4


def test_weighted_shard_plan_balances_the_rows_that_are_scanned():
    """mi355_shard_plan_weighted (host code): cost = rows x probe frequency.  Without weights it is mi355_shard_plan;
    with a skewed probe histogram the heaviest shard's scanned rows drop to within a few percent of the mean."""
    rng = np.random.default_rng(3)
    nlist, shards = 4096, 8
    lens = rng.multinomial(10_000_000, (lambda w: w / w.sum())(np.exp(rng.normal(0, 0.5, nlist))))
    po = np.zeros(nlist + 1, np.uint64)
    po[1:] = np.cumsum(lens)
    plain = lancedb_amd.shard_plan(po, shards)
    assert (lancedb_amd.shard_plan(po, shards, weights=np.ones(nlist)) == plain).all()
    hits = rng.gamma(1.5, 20.0, nlist)  # popular and unpopular partitions, independent of their size
    hits[rng.choice(nlist, 50, replace=False)] = 0.0  # never probed
    weighted = lancedb_amd.shard_plan(po, shards, weights=hits)
    assert weighted.max() < shards and len(np.unique(weighted)) == shards

    def imbalance(owner):
        cost = np.array([(lens[owner == r] * hits[owner == r]).sum() for r in range(shards)])
        return cost.max() / cost.mean()
    assert imbalance(weighted) < 1.01 < imbalance(plain)
    rows = np.array([lens[weighted == r].sum() for r in range(shards)])
    assert rows.max() / rows.mean() < 1.25  # rows held stay reasonable too
    # half of the partitions unseen by the calibration sample (C4 in round 4: 65536 partitions, a 2048-query sample): they are
    # spread by the rows they hold, not piled onto the shard the cost balance happened to leave lightest
    hits2 = hits.copy()
    hits2[rng.choice(nlist, nlist // 2, replace=False)] = 0.0
    w2 = lancedb_amd.shard_plan(po, shards, weights=hits2)
    rows2 = np.array([lens[w2 == r].sum() for r in range(shards)])
    cost2 = np.array([(lens[w2 == r] * hits2[w2 == r]).sum() for r in range(shards)])
    assert rows2.max() / rows2.mean() < 1.15 and cost2.max() / cost2.mean() < 1.01


def test_no_compiler_copy_of_a_register_with_a_load_in_flight():
    """The multi-slab scan hands partial row sums from slab to slab through a register that inline asm loads one tile
    ahead (csrc/kernels_skew.h, SLABBED): hipcc must never copy that register between the load and the counted wait
    (round 4 saw exactly that: a v_mov of the in-flight register, then its late-landing data overwrote an address).
    scripts/check_inflight_regs.py compiles the translation unit to assembly and checks every instantiation."""
    import shutil
    import subprocess
    import sys
    from lancedb_amd import _lib
    if shutil.which(_lib._hipcc()) is None and not os.path.exists(_lib._hipcc()):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_inflight_regs.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 copies of a register in flight" in r.stdout


def test_bench_helpers_host_cores_and_summary():
    """bench_legs.host_cores: what the CPU legs are printed with — the affinity mask capped by the cgroup CPU quota (round 3 printed
    os.cpu_count() = 256 on a box that grants the container 16 CPUs); summary_of: the compact trailer of the bench line."""
    import bench_legs as legs
    hc = legs.host_cores()
    assert 1 <= hc["usable"] <= hc["affinity"] <= hc["box"]
    if hc["cgroup_cpu_quota"] is not None:
        assert hc["usable"] <= max(1, int(hc["cgroup_cpu_quota"] + 0.5))
    line = {"value": 65000.0, "roofline": {"frac": 1.23, "lds_gather": {"frac": 0.5}},
            "secondary": {"c4": {"value": 41000.0, "roofline": {"frac": 1.05}, "config": {"n_rows": 10 ** 9}},
                          "c3_shape_dim384_m24": {"value": 110000.0, "roofline": {"frac": 0.53}},
                          "qps_vs_batch": [{"batch": 1, "queries_per_s": 4800.0}, {"batch": 2048, "queries_per_s": 65000.0}],
                          "concurrent_callers_c3": {"coalesced_64_threads": {"queries_per_s": 40000.0}}}}
    s = legs.summary_of(line)
    assert s["c3_qps"] == 65000.0 and s["c4_rows"] == 10 ** 9 and s["c3_shape_dim384_m24_frac"] == 0.53
    assert s["qps_vs_batch"] == {"1": 4800, "2048": 65000} and s["callers_qps"] == {"coalesced_64": 40000}
    assert s["c5_qps"] is None  # absent legs stay visible as null
    import json
    assert len(json.dumps(s)) < 2000  # the driver keeps the last 2000 characters of the line


def test_coalescing_queue_under_thread_sanitizer(tmp_path):
    """The handle's coalescing queue (csrc/call_queue.h: futex-word wake-ups, direct hand-over of the device, batching
    window) compiled alone with ThreadSanitizer and driven by 48 CPU threads — a mutex-free stand-in for the device checks
    exclusive ownership, every call runs in exactly one batch, failed batches reach every caller they carried, and a lost
    wake-up would hang (timeout).  tests/test_gpu_concurrency.py runs the same header inside the library on the GPU."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "queue_stress")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-I", os.path.join(root, "lancedb_amd", "csrc"),
           os.path.join(root, "tests", "tools", "queue_stress.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("this toolchain has no ThreadSanitizer runtime")
    assert b.returncode == 0, b.stderr[-2000:]
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "ASAN_OPTIONS", "UBSAN_OPTIONS")}  # (scripts/asan_abi_tests.sh preloads ASan)
    r = subprocess.run([exe, "48", "300"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.stdout + r.stderr)[-3000:]
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]


def test_latency_path_kernels_use_no_scratch():
    """Round 5: register arrays of HIP `float4` / `uint4` structs (a dozen elements or more), and arrays indexed in a loop with an
    early exit, are placed in SCRATCH by hipcc and every global load that fills them is then waited for on its own — the
    latency path's coarse kernel spent 16 of its 27 us in 24 serial HBM round trips that way.  scripts/check_scratch.py
    compiles a translation unit's device code and lists the kernels with a private segment; the search pipeline's unit
    (coarse, select, plan, merge, refine kernels) must have none."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_scratch.py"),
                        os.path.join(root, "lancedb_amd", "csrc", "ann_index.hip")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert "outside the allow-list: 0" in r.stdout
