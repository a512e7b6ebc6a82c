"""Arrow-level loader for a Lance IVF_PQ index (the feasible part).

The reference opens an index through lance (`VectorIndex::new_from_format`,
rust/lancedb/src/index/vector.rs:20-40; `load_indices`, table/create_index.rs:38-46; the files
live under `<table>.lance/_indices/<uuid>/`, nodejs/__test__/table.test.ts:907-908).  The lance
file reader and the protobuf index metadata are NOT in this container (lance is an un-vendored
git dependency), so this module starts one step later: it takes the pieces of an
IVF_PQ index as the ARROW ARRAYS a reader would hand over and turns them, zero-copy wherever
Arrow allows, into what `mi355_index_open` takes.

Every assumption about lance v11.0.0-beta.19's storage is marked [EXT] and must be re-checked
against that revision before this is trusted on real files:

  [EXT-1] IVF model: `centroids` FixedSizeList<float32, dim> of length nlist, plus per-partition
          `offsets` / `lengths` (rows of the storage file), in partition order.
  [EXT-2] PQ codebook: FixedSizeList<float32, dim> of length 2^nbits — row c is the
          concatenation over sub-vectors j of centroid c of sub-quantiser j
          (`codebook_layout="code_major"`); the engine wants [m, 2^nbits, dim/m].
  [EXT-3] Storage: column `__pq_code` FixedSizeList<uint8, m * nbits / 8> and column `_rowid`
          uint64, rows grouped by partition; when the storage metadata says `transposed`, the
          bytes of ONE partition are sub-quantiser major ([m * nbits / 8, len_p]) — exactly
          MI355_CODES_PART_TRANSPOSED, so the values buffer is handed to the engine as is.
  [EXT-4] 4-bit codes: sub-quantiser 2t in the low nibble of byte t.
  [EXT-5] metric: the index's distance type string ("l2" | "cosine" | "dot").
"""
import numpy as np

from . import _abi
from ._lib import InvalidInput
from .index import IvfPqIndex


def _fsl_to_numpy(arr, dtype, what):
    """FixedSizeListArray -> [len, list_size] numpy view (zero-copy when the child has no nulls/offset)."""
    import pyarrow as pa
    if isinstance(arr, pa.ChunkedArray):
        arr = arr.combine_chunks()
    if not pa.types.is_fixed_size_list(arr.type):
        raise InvalidInput(1, f"{what} must be a FixedSizeList array, got {arr.type}")
    if arr.null_count:
        raise InvalidInput(1, f"{what} has nulls")
    width = arr.type.list_size
    values = arr.flatten()  # honours the parent's offset
    a = values.to_numpy(zero_copy_only=False)
    if a.dtype != dtype:
        a = a.astype(dtype)
    return np.ascontiguousarray(a).reshape(len(arr), width)


def engine_arrays(centroids, codebook, pq_codes, row_ids, offsets, lengths, nbits=8, transposed=True,
                  codebook_layout="code_major"):
    """Arrow pieces of an IVF_PQ index -> dict(centroids [nlist, dim] f32, codebook [m, 2^nbits, dsub] f32,
    part_offsets [nlist + 1] u64, codes u8 (flat), codes_layout, row_ids [n] u64, nbits, m).

    `pq_codes` / `row_ids` are the storage columns (rows grouped by partition, possibly with gaps:
    partition p = rows offsets[p] .. offsets[p] + lengths[p]); partitions are compacted in order."""
    cen = _fsl_to_numpy(centroids, np.float32, "centroids")
    nlist, dim = cen.shape
    codes2d = _fsl_to_numpy(pq_codes, np.uint8, "__pq_code")
    mb = codes2d.shape[1]
    if (mb * 8) % nbits:
        raise InvalidInput(1, f"{mb} code bytes per row do not hold whole {nbits}-bit codes")
    m = mb * 8 // nbits
    if dim % m:
        raise InvalidInput(1, f"dim {dim} is not divisible by num_sub_vectors {m}")
    dsub, ks = dim // m, 1 << nbits
    cb = _fsl_to_numpy(codebook, np.float32, "codebook")
    if codebook_layout == "code_major":  # [EXT-2]: [2^nbits, dim] -> [m, 2^nbits, dsub]
        if cb.shape != (ks, dim):
            raise InvalidInput(1, f"codebook is {cb.shape}, expected ({ks}, {dim})")
        cb = np.ascontiguousarray(cb.reshape(ks, m, dsub).transpose(1, 0, 2))
    elif codebook_layout == "sub_vector_major":  # [m * 2^nbits, dsub]
        if cb.shape != (m * ks, dsub):
            raise InvalidInput(1, f"codebook is {cb.shape}, expected ({m * ks}, {dsub})")
        cb = cb.reshape(m, ks, dsub)
    else:
        raise InvalidInput(1, f"unknown codebook_layout {codebook_layout!r}")
    import pyarrow as pa
    rid = row_ids.combine_chunks() if isinstance(row_ids, pa.ChunkedArray) else row_ids
    rid = np.ascontiguousarray(rid.to_numpy(zero_copy_only=False), dtype=np.uint64)
    off = np.asarray(offsets, dtype=np.int64)
    ln = np.asarray(lengths, dtype=np.int64)
    if off.shape != (nlist,) or ln.shape != (nlist,):
        raise InvalidInput(1, "offsets / lengths must have one entry per partition")
    if (ln < 0).any() or (off < 0).any() or ((off + ln) > len(rid)).any():
        raise InvalidInput(1, "a partition runs outside the storage")
    po = np.zeros(nlist + 1, dtype=np.uint64)
    po[1:] = np.cumsum(ln)
    compact = bool((off == po[:-1].astype(np.int64)).all()) and int(po[-1]) == len(rid)
    flat_codes = codes2d.reshape(-1)
    if not compact:  # gaps / reordering: gather the partitions (each partition's block stays intact)
        pieces_c, pieces_r = [], []
        for p in range(nlist):
            pieces_c.append(flat_codes[off[p] * mb:(off[p] + ln[p]) * mb])
            pieces_r.append(rid[off[p]:off[p] + ln[p]])
        flat_codes = np.concatenate(pieces_c) if pieces_c else flat_codes[:0]
        rid = np.concatenate(pieces_r) if pieces_r else rid[:0]
    return dict(centroids=cen, codebook=cb, part_offsets=po, codes=flat_codes, row_ids=rid, nbits=nbits, m=m,
                codes_layout=_abi.CODES_PART_TRANSPOSED if transposed else _abi.CODES_ROW_MAJOR)


def open_ivf_pq(centroids, codebook, pq_codes, row_ids, offsets, lengths, metric="l2", nbits=8, transposed=True,
                codebook_layout="code_major", raw_vectors=None, raw_dtype=_abi.DTYPE_F32, device=0, **open_kw):
    """-> IvfPqIndex on the GPU from the Arrow pieces (see engine_arrays); `raw_vectors` (index
    order) enables refine."""
    a = engine_arrays(centroids, codebook, pq_codes, row_ids, offsets, lengths, nbits, transposed, codebook_layout)
    codes = a["codes"] if transposed else a["codes"].reshape(-1, a["m"] * nbits // 8)
    return IvfPqIndex(a["centroids"], a["codebook"], a["part_offsets"], codes, a["row_ids"], raw_vectors=raw_vectors,
                      metric=metric, codes_layout=a["codes_layout"], raw_dtype=raw_dtype, device=device, nbits=nbits, **open_kw)


def verify_against_raw(arrays, raw_rows, raw_row_ids, metric="l2", encode=None, min_partition_match=0.98, min_byte_match=0.95,
                       strict=True):
    """Self-check of the [EXT] guesses above, for the shim to call once per opened index (a wrong
    guess about the codebook layout, the transposed storage or the nibble order returns garbage SILENTLY).

    `arrays` = engine_arrays(...); `raw_rows` [s, dim] f32 are the raw vectors of s sampled rows and `raw_row_ids` [s] their
    `_rowid` values (the shim takes them from the table it indexes: `take` of a few thousand rows).  The sample is re-encoded
    with the index's OWN centroids and codebook — `mi355_ivfpq_encode` by default, or any callable with
    `lancedb_amd.ivfpq_encode`'s signature and return value — and compared with what the index stores: the partition every
    row sits in, and its code bytes.  lance's trainer / encoder differ from the engine's in rounding and tie-breaks [EXT],
    so a few rows may land in a neighbouring partition or pick another codeword: the thresholds are fractions, not
    equality.  A wrong layout guess leaves ~1 / 2^nbits of the bytes matching.

    -> dict(rows, found, partition_match, byte_match, code_match, ok); raises InvalidInput when `strict` and not ok."""
    cen, cb, po, codes, rid = (arrays[k] for k in ("centroids", "codebook", "part_offsets", "codes", "row_ids"))
    nbits, m = int(arrays["nbits"]), int(arrays["m"])
    mb = m * nbits // 8
    raw = np.ascontiguousarray(raw_rows, dtype=np.float32)
    ids = np.ascontiguousarray(raw_row_ids, dtype=np.uint64)
    if raw.ndim != 2 or raw.shape[1] != cen.shape[1] or ids.shape != (raw.shape[0],):
        raise InvalidInput(1, "raw_rows must be [s, dim] and raw_row_ids [s]")
    if encode is None:
        from .index import ivfpq_encode as encode
    _, enc_codes, enc_order, enc_assign = encode(raw, cen, cb, metric=metric, nbits=nbits, return_assign=True)
    enc_codes = np.asarray(enc_codes).reshape(-1, mb)
    enc_order = np.asarray(enc_order).astype(np.int64)
    by_row = np.empty_like(enc_codes)
    by_row[enc_order] = enc_codes            # codes of input row i (the encoder returns them in index order)
    assign = np.asarray(enc_assign).astype(np.int64)
    # where the index stores each sampled row
    sorter = np.argsort(rid, kind="stable")
    at = np.searchsorted(rid, ids, sorter=sorter)
    at = np.minimum(at, len(rid) - 1) if len(rid) else at
    pos = sorter[at] if len(rid) else at
    found = (rid[pos] == ids) if len(rid) else np.zeros(len(ids), bool)
    po_i = po.astype(np.int64)
    part = np.searchsorted(po_i, pos, side="right") - 1
    ln = po_i[part + 1] - po_i[part]
    local = pos - po_i[part]
    j = np.arange(mb, dtype=np.int64)[None, :]
    if arrays["codes_layout"] == _abi.CODES_PART_TRANSPOSED:
        idx = po_i[part][:, None] * mb + j * ln[:, None] + local[:, None]
    else:
        idx = pos[:, None] * mb + j
    stored = np.asarray(codes).reshape(-1)[np.where(found[:, None], idx, 0)]
    n_found = int(found.sum())
    same_part = found & (part == assign)
    byte_eq = (stored == by_row) & same_part[:, None]
    denom = max(int(same_part.sum()), 1)
    out = {"rows": int(len(ids)), "found": n_found,
           "partition_match": float(same_part.sum()) / max(n_found, 1),
           "byte_match": float(byte_eq.sum()) / (denom * mb),
           "code_match": float(byte_eq.all(axis=1).sum()) / denom}
    out["ok"] = bool(n_found == len(ids) and out["partition_match"] >= min_partition_match and out["byte_match"] >= min_byte_match)
    if strict and not out["ok"]:
        raise InvalidInput(1, "the index does not decode to its own raw rows under the assumed lance layout "
                              f"([EXT-2/3/4], lance_loader.py): {out}")
    return out
