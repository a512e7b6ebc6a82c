"""The C-ABI library loads, exports every symbol include/mi355_ann.h declares,
and rejects bad input before touching a device.  No GPU needed."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import lancedb_amd
from lancedb_amd import _abi, _lib

HEADER = os.path.join(os.path.dirname(os.path.dirname(__file__)), "include", "mi355_ann.h")


@pytest.fixture(scope="module")
def L():
    _lib.build()
    return _lib.lib()


def test_exports_match_header(L):
    src = open(HEADER).read()
    declared = set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", src))
    assert declared == set(_abi.EXPORTED_SYMBOLS)
    for name in declared:
        assert getattr(L, name) is not None
    assert L.mi355_abi_version() == _abi.ABI_VERSION


def test_struct_sizes_match_c_layout(tmp_path):
    """Compile a tiny C program against the header and compare sizeof()s."""
    import subprocess
    c = tmp_path / "sz.c"
    pairs = [("mi355_index_desc", _abi.IndexDesc), ("mi355_search_params", _abi.SearchParams),
             ("mi355_flat_desc", _abi.FlatDesc), ("mi355_stats", _abi.Stats), ("mi355_flat_stats", _abi.FlatStats),
             ("mi355_comm_stats", _abi.CommStats), ("mi355_encode_desc", _abi.EncodeDesc),
             ("mi355_kmeans_desc", _abi.KmeansDesc), ("mi355_pq_train_desc", _abi.PqTrainDesc)]
    c.write_text('#include "mi355_ann.h"\n#include <stdio.h>\nint main(){printf("' + " ".join(["%zu"] * len(pairs)) + '\\n",'
                 + ",".join(f"sizeof({n})" for n, _ in pairs) + ');return 0;}')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.dirname(HEADER), str(c), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert [int(x) for x in out] == [C.sizeof(t) for _, t in pairs]


def _desc(**over):
    cen = np.zeros((4, 8), np.float32)
    cb = np.zeros((2, 256, 4), np.float32)
    po = np.array([0, 1, 2, 3, 4], np.uint64)
    codes = np.zeros((4, 2), np.uint8)
    d = _abi.IndexDesc()
    d.struct_size = C.sizeof(_abi.IndexDesc)
    d.dim, d.nlist, d.m, d.nbits, d.metric, d.n_rows = 8, 4, 2, 8, 0, 4
    d.centroids = cen.ctypes.data_as(C.c_void_p)
    d.codebook = cb.ctypes.data_as(C.c_void_p)
    d.part_offsets = po.ctypes.data_as(C.c_void_p)
    d.codes = codes.ctypes.data_as(C.c_void_p)
    d.shard_count = 1
    for k, v in over.items():
        setattr(d, k, v)
    return d, (cen, cb, po, codes)


@pytest.mark.parametrize("over,status,needle", [
    (dict(struct_size=8), _abi.ERR_INVALID_INPUT, "struct_size"),
    (dict(nbits=4, m=1), _abi.ERR_INVALID_INPUT, "even when num_bits is 4"),  # table/create_index.rs:96-101
    (dict(flags=64), _abi.ERR_INVALID_INPUT, "flags"),
    (dict(nbits=7), _abi.ERR_INVALID_INPUT, "num_bits"),
    (dict(m=3), _abi.ERR_INVALID_INPUT, "divisible"),
    (dict(metric=9), _abi.ERR_INVALID_INPUT, "metric"),
    (dict(n_rows=5), _abi.ERR_INVALID_INPUT, "part_offsets"),
    (dict(shard_count=2, shard_rank=2), _abi.ERR_INVALID_INPUT, "shard_rank"),
    (dict(m=2, dim=40000), _abi.ERR_NOT_SUPPORTED, "LDS"),  # (m = 200 x 256 entries spill to global memory: fine)
])
def test_index_open_rejects_bad_descriptors(L, over, status, needle):
    d, keep = _desc(**over)
    h = C.c_void_p()
    assert L.mi355_index_open(C.byref(d), C.byref(h)) == status
    assert needle in _lib.last_error()
    assert not h.value


def test_open_without_gpu_fails_loudly(L):
    """No CPU fallback: on a box without a device the open is a Runtime error."""
    if lancedb_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    d, keep = _desc()
    h = C.c_void_p()
    assert L.mi355_index_open(C.byref(d), C.byref(h)) == _abi.ERR_RUNTIME
    assert "no HIP device" in _lib.last_error() and "no CPU fallback" in _lib.last_error()
    with pytest.raises(lancedb_amd.EngineError):
        lancedb_amd.FlatIndex(np.zeros((4, 4), np.float32))


def test_shard_plan_matches_oracle(L, oracle):
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 500, size=300)
    po = np.zeros(301, np.uint64)
    po[1:] = np.cumsum(lens)
    for shards in (1, 2, 3, 8):
        got = lancedb_amd.shard_plan(po, shards)
        assert (got == oracle.shard_plan(po, shards)).all()
        loads = np.array([lens[got == s].sum() for s in range(shards)])
        assert loads.max() - loads.min() <= lens.max()


def test_missing_library_is_an_import_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.lib()


def test_encode_validates_before_touching_a_device(L):
    cent = np.zeros((4, 8), np.float32)
    cb = np.zeros((2, 256, 4), np.float32)
    po = np.zeros(5, np.uint64)

    def call(**over):
        f = dict(struct_size=C.sizeof(_abi.EncodeDesc), dim=8, nlist=4, m=2, nbits=8, metric=0, mem=0, device=0,
                 centroids=cent.ctypes.data, codebook=cb.ctypes.data)
        f.update(over)
        d = _abi.EncodeDesc(**f)
        return L.mi355_ivfpq_encode(C.byref(d), None, C.c_uint64(0), C.c_void_p(po.ctypes.data), None, None, None)

    assert call(struct_size=8) == _abi.ERR_INVALID_INPUT and "ABI mismatch" in _lib.last_error()
    assert call(m=3) == _abi.ERR_INVALID_INPUT
    assert call(nbits=5) == _abi.ERR_INVALID_INPUT and "num_bits" in _lib.last_error()
    assert call(nbits=4, m=1, dim=8) == _abi.ERR_INVALID_INPUT and "even" in _lib.last_error()
    assert call(metric=7) == _abi.ERR_INVALID_INPUT
    assert call(centroids=None) == _abi.ERR_INVALID_INPUT
    with pytest.raises(ValueError):
        lancedb_amd.ivfpq_encode(np.zeros((3, 8), np.float32), cent, np.zeros((3, 256, 4), np.float32))
    with pytest.raises(ValueError):
        lancedb_amd.ivfpq_encode(np.zeros((3, 8), np.float32), np.zeros((4, 6), np.float32), cb)


def test_kmeans_validates_before_touching_a_device(L):
    cen = np.zeros((4, 8), np.float32)

    def call(fn, **over):
        f = dict(struct_size=C.sizeof(_abi.KmeansDesc), dim=8, k=4, metric=0, iters=1, mem=0, device=0, reserved0=0, ld=0)
        f.update(over)
        d = _abi.KmeansDesc(**f)
        if fn == "train":
            return L.mi355_kmeans_train(C.byref(d), None, C.c_uint64(0), C.c_void_p(cen.ctypes.data), None)
        return L.mi355_ivf_residuals(C.byref(d), None, C.c_uint64(0), C.c_void_p(cen.ctypes.data), None, None)

    for fn in ("train", "resid"):
        assert call(fn, struct_size=4) == _abi.ERR_INVALID_INPUT and "ABI mismatch" in _lib.last_error()
        assert call(fn, k=0) == _abi.ERR_INVALID_INPUT
        assert call(fn, metric=9) == _abi.ERR_INVALID_INPUT
        assert call(fn, ld=4) == _abi.ERR_INVALID_INPUT
    with pytest.raises(ValueError):
        lancedb_amd.kmeans_train(np.zeros((10, 8), np.float32), np.zeros((4, 6), np.float32))
    with pytest.raises(ValueError):
        lancedb_amd.kmeans_train(np.zeros((10, 8), np.float32), np.zeros((4, 4), np.float32), cols=(6, 10))


def test_builder_parameter_defaults_follow_the_reference():
    """index/vector.rs:306-319 (sub-vectors); partitions = rows / 8192, the default partition size
    pinned by table/create_index.rs:733-795."""
    assert [lancedb_amd.suggested_num_sub_vectors(d) for d in (768, 1536, 24, 7, 16, 8)] == [48, 96, 3, 1, 1, 1]
    assert lancedb_amd.suggested_num_partitions(1_000_000) == 122
    assert lancedb_amd.IvfPqBuilder(num_bits=4).num_bits == 4
    b = lancedb_amd.IvfPqBuilder()
    assert (b.sample_rate, b.max_iterations, b.distance_type) == (256, 50, "l2")
    with pytest.raises(ValueError):
        b.train(np.zeros((10, 16), np.float32))


def test_flat_gemm_tile_walk_visits_every_tile_once():
    """Model of k_flat_gemm's virtual-block walk (csrc/kernels_flat_mfma.h: vb = blockIdx.x +
    i * gridDim.x; xcd = vb & 7, slot = vb >> 3, qt = slot % n_qtiles, rt = (slot / n_qtiles) * 8
    + xcd; virtual blocks with rt >= n_rtiles are skipped): for any grid that is a multiple of 8
    every (row tile, query tile) is visited exactly once, by a workgroup whose blockIdx % 8 is the
    tile's XCD label, and the query tiles of one row tile are adjacent slots on that XCD."""
    for n_rtiles, n_qtiles in ((1, 1), (7, 3), (8, 4), (36, 2), (61, 5), (200, 8)):
        total_vb = (n_rtiles + 7) // 8 * 8 * n_qtiles
        for grid in (8, 16, 64, 256, total_vb):
            grid = min(grid, total_vb)
            seen = {}
            for b in range(grid):
                vb = b
                while vb < total_vb:
                    slot = vb >> 3
                    qt, rt = slot % n_qtiles, (slot // n_qtiles) * 8 + (vb & 7)
                    if rt < n_rtiles:
                        assert (rt, qt) not in seen
                        seen[(rt, qt)] = b
                        assert b % 8 == rt % 8
                    vb += grid
            assert len(seen) == n_rtiles * n_qtiles


_OOM_CHILD = r"""
import ctypes as C, resource, sys
L = C.CDLL(sys.argv[1])          # (libamdhip64 is mapped before the limit goes down)
L.mi355_shard_plan.restype = C.c_int32
L.mi355_last_error.restype = C.c_int32
po = (C.c_uint64 * 8)()
out = (C.c_uint32 * 8)()
soft = 1 << 30
used = int(open('/proc/self/statm').read().split()[0]) * resource.getpagesize()
resource.setrlimit(resource.RLIMIT_AS, (used + soft, used + soft))
# nlist = 2^30 partitions: the plan's first container is 4 GiB > the limit -> std::bad_alloc inside the library,
# thrown before any of the (too short) arrays is read
st = L.mi355_shard_plan(po, C.c_uint32(1 << 30), C.c_uint32(2), out)
buf = C.create_string_buffer(512)
L.mi355_last_error(buf, C.c_size_t(512))
print(st, buf.value.decode())
"""


def test_out_of_host_memory_is_a_status_not_a_crash(L):
    """include/mi355_ann.h: "no exception or abort crosses the ABI" — every entry point is a function-try-block
    (MI355_ABI_GUARD, csrc/ann_internal.h).  A host allocation that fails inside the library must come back as
    MI355_ERR_RUNTIME (rust/lancedb/src/error.rs: Runtime) with a message, not as std::terminate."""
    import subprocess
    import sys
    if "asan" in os.environ.get("LD_PRELOAD", ""):
        pytest.skip("AddressSanitizer reserves terabytes of address space and aborts on allocation failure: RLIMIT_AS cannot be used under it")
    r = subprocess.run([sys.executable, "-c", _OOM_CHILD, _lib.LIB_PATH], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    status, _, msg = r.stdout.strip().partition(" ")
    assert int(status) == 2 and "mi355_shard_plan" in msg and "out of host memory" in msg, r.stdout


def test_every_entry_point_has_the_exception_barrier():
    """Source check: each `extern "C"` definition that can allocate is a function-try-block closed by MI355_ABI_GUARD
    naming itself (mi355_abi_version returns a constant)."""
    csrc = os.path.join(os.path.dirname(os.path.dirname(__file__)), "lancedb_amd", "csrc")
    seen = set()
    for f in sorted(os.listdir(csrc)):
        if not f.endswith(".hip"):
            continue
        src = open(os.path.join(csrc, f)).read()
        defs = re.findall(r'extern "C" (?:int32_t|uint32_t) (mi355_\w+)\(', src)
        guards = re.findall(r'MI355_ABI_GUARD\("(mi355_\w+)"\)', src)
        assert sorted(d for d in defs if d != "mi355_abi_version") == sorted(guards), f
        for d in defs:
            if d != "mi355_abi_version":
                assert re.search(r'extern "C" (?:int32_t|uint32_t) ' + d + r"\([^{;]*\) try \{", src), (f, d)
        seen.update(defs)
    assert seen - {"mi355_dev_counters", "mi355_dev_timeline"} == set(_abi.EXPORTED_SYMBOLS)  # (dev_*: developer builds only, not in the header)
