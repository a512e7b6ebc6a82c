"""The Arrow-level IVF_PQ loader (lancedb_amd/lance_loader.py; SURVEY.md §8f rank 1, the part that
does not need lance's file reader): Arrow arrays in the storage layout the loader assumes ([EXT-1..5]
in its docstring) -> the arrays mi355_index_open takes.  CPU: the produced arrays are checked by
searching them with the oracle against an oracle index built from the plain numpy arrays; the GPU
test opens them on the device."""
import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")

from lancedb_amd import _abi, lance_loader  # noqa: E402
from oracle import train  # noqa: E402


def _arrow_index(s, nbits=8, gap=0):
    """The synthetic index `s` as the Arrow pieces of a lance IVF_PQ index (transposed storage,
    code-major codebook), optionally with `gap` unused storage rows between partitions."""
    nlist, dim = s["centroids"].shape
    m, ks, dsub = s["codebook"].shape
    mb = s["codes"].shape[1]
    po = s["part_offsets"].astype(np.int64)
    cen = pa.FixedSizeListArray.from_arrays(pa.array(s["centroids"].reshape(-1)), dim)
    cb_code_major = np.ascontiguousarray(s["codebook"].transpose(1, 0, 2)).reshape(ks, dim)
    cb = pa.FixedSizeListArray.from_arrays(pa.array(cb_code_major.reshape(-1)), dim)
    blocks, rids, offsets, lengths, at = [], [], [], [], 0
    for p in range(nlist):
        rows = s["codes"][po[p]:po[p + 1]]
        blocks.append(np.ascontiguousarray(rows.T).reshape(-1))  # [mb, len_p]: sub-quantiser major
        rids.append(s["row_ids"][po[p]:po[p + 1]])
        offsets.append(at)
        lengths.append(len(rows))
        at += len(rows)
        if gap:
            blocks.append(np.full(gap * mb, 0xEE, np.uint8))
            rids.append(np.full(gap, 2 ** 63, np.uint64))
            at += gap
    codes = pa.FixedSizeListArray.from_arrays(pa.array(np.concatenate(blocks)), mb)
    rid = pa.array(np.concatenate(rids), type=pa.uint64())
    return cen, cb, codes, rid, offsets, lengths


@pytest.mark.parametrize("nbits,gap", [(8, 0), (8, 3), (4, 0)])
def test_engine_arrays_reproduce_the_index(oracle, nbits, gap):
    s = train.synthetic_index(6000, 32, 12, 8, seed=3, empty_parts=2, nbits=nbits)
    a = lance_loader.engine_arrays(*_arrow_index(s, nbits, gap), nbits=nbits)
    assert a["m"] == 8 and a["codes_layout"] == _abi.CODES_PART_TRANSPOSED
    assert (a["part_offsets"] == s["part_offsets"]).all() and (a["row_ids"] == s["row_ids"]).all()
    assert (a["centroids"] == s["centroids"]).all() and (a["codebook"] == s["codebook"]).all()
    assert (a["codes"] == train.to_part_transposed(s["codes"], s["part_offsets"])).all()
    ox = oracle.OracleIndex(a["centroids"], a["codebook"], a["part_offsets"], a["codes"], a["row_ids"],
                            codes_layout=a["codes_layout"], nbits=nbits)
    ref = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], nbits=nbits)
    q = np.random.default_rng(1).normal(size=(7, 32)).astype(np.float32)
    got, exp = ox.search(q, k=10, nprobe_min=5, nprobe_max=5), ref.search(q, k=10, nprobe_min=5, nprobe_max=5)
    assert (got[0] == exp[0]).all() and (got[1] == exp[1]).all()


def test_loader_rejects_inconsistent_pieces():
    s = train.synthetic_index(500, 16, 4, 4, seed=1)
    cen, cb, codes, rid, off, ln = _arrow_index(s)
    with pytest.raises(Exception, match="one entry per partition"):
        lance_loader.engine_arrays(cen, cb, codes, rid, off[:-1], ln[:-1])
    with pytest.raises(Exception, match="outside the storage"):
        lance_loader.engine_arrays(cen, cb, codes, rid, off, [x + 400 for x in ln])
    with pytest.raises(Exception, match="codebook is"):
        lance_loader.engine_arrays(cen, cen, codes, rid, off, ln)
    with pytest.raises(Exception, match="FixedSizeList"):
        lance_loader.engine_arrays(pa.array([1.0, 2.0]), cb, codes, rid, off, ln)


@pytest.mark.gpu
def test_open_ivf_pq_on_the_device(oracle):
    s = train.synthetic_index(30000, 128, 24, 32, seed=9, skew=0.8, empty_parts=2)
    ix = lance_loader.open_ivf_pq(*_arrow_index(s, gap=5), metric="l2")
    ref = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    q = np.random.default_rng(2).normal(size=(9, 128)).astype(np.float32)
    got = ix.search(q, k=10, nprobe_min=8, nprobe_max=8)
    ids, dist, cnt, _ = ref.search(q, k=10, nprobe_min=8, nprobe_max=8)
    assert (got.rowids == ids).all() and (got.distances == dist).all() and (got.counts == cnt).all()
    assert ix.stats()["scan_variant"] == _abi.SCAN_SKEW


@pytest.mark.parametrize("nbits", [8, 4])
def test_self_check_catches_a_wrong_layout_guess(oracle, nbits):
    """verify_against_raw re-encodes sampled raw rows with the index's own centroids and codebook and compares partitions and code
    bytes with what the index stores (here with the ORACLE's encoder: the check itself needs no GPU).  A trained index passes; each
    wrong [EXT] guess — row-major storage read as transposed, a codebook read in the other layout, swapped nibbles — fails."""
    rng = np.random.default_rng(4)
    n, dim, nlist, m = 3000, 32, 8, 8
    x = rng.normal(size=(n, dim)).astype(np.float32)
    cen = x[rng.choice(n, nlist, replace=False)].copy()
    cb = (rng.normal(size=(m, 1 << nbits, dim // m)) * 0.7).astype(np.float32)
    po, codes, order, _ = oracle.ivfpq_encode(x, cen, cb, nbits=nbits)
    s = {"centroids": cen, "codebook": cb, "part_offsets": po, "codes": codes, "row_ids": order.astype(np.uint64)}
    pieces = _arrow_index(s, nbits)
    enc = lambda v, c, b, metric="l2", nbits=8, return_assign=True: oracle.ivfpq_encode(v, c, b, metric=metric, nbits=nbits)  # noqa: E731
    pick = rng.choice(n, 500, replace=False)
    good = lance_loader.engine_arrays(*pieces, nbits=nbits)
    rep = lance_loader.verify_against_raw(good, x[pick], pick.astype(np.uint64), encode=enc)
    assert rep["ok"] and rep["found"] == 500 and rep["partition_match"] == 1.0 and rep["code_match"] == 1.0
    # [EXT-3] wrong: the storage is transposed per partition, read as row-major
    bad = lance_loader.engine_arrays(*pieces, nbits=nbits, transposed=False)
    rep = lance_loader.verify_against_raw(bad, x[pick], pick.astype(np.uint64), encode=enc, strict=False)
    assert not rep["ok"] and rep["byte_match"] < 0.5
    with pytest.raises(Exception, match="does not decode to its own raw rows"):
        lance_loader.verify_against_raw(bad, x[pick], pick.astype(np.uint64), encode=enc)
    # [EXT-2] wrong: a sub-vector-major codebook handed over as code-major
    cen_a, cb_a, codes_a, rid_a, off, ln = pieces
    cb_sv = pa.FixedSizeListArray.from_arrays(pa.array(np.ascontiguousarray(cb).reshape(-1)), dim)  # [m * ks * dsub] cut into rows of dim
    bad = lance_loader.engine_arrays(cen_a, cb_sv, codes_a, rid_a, off, ln, nbits=nbits)
    rep = lance_loader.verify_against_raw(bad, x[pick], pick.astype(np.uint64), encode=enc, strict=False)
    assert not rep["ok"]
    if nbits == 4:  # [EXT-4] wrong: sub-quantiser 2t in the HIGH nibble
        sw = dict(good)
        sw["codes"] = ((good["codes"] >> 4) | (good["codes"] << 4)).astype(np.uint8)
        rep = lance_loader.verify_against_raw(sw, x[pick], pick.astype(np.uint64), encode=enc, strict=False)
        assert not rep["ok"] and rep["byte_match"] < 0.5
    # rows the index does not hold
    rep = lance_loader.verify_against_raw(good, x[pick[:10]], (pick[:10] + 10 ** 9).astype(np.uint64), encode=enc, strict=False)
    assert not rep["ok"] and rep["found"] == 0
