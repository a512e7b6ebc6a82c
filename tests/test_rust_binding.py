"""integration/mi355_sys.rs — the Rust `extern "C"` module a maintainer adds to put the engine behind
`BaseTable::{create_plan, query}` (rust/lancedb/src/table.rs:549-576) — is generated from include/mi355_ann.h and checked
here mechanically, because no Rust toolchain exists in the image: (1) the committed file is what the generator produces
from the current header; (2) an independent parse of the .rs text agrees with a C program compiled against the header on
every struct's size and every field's offset and width (so a reordered, missing or mistyped field fails), and with the
header's prototypes on the function set and argument counts; (3) the function set is the library's export list."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
RS = os.path.join(ROOT, "integration", "mi355_sys.rs")
HEADER = os.path.join(ROOT, "include", "mi355_ann.h")

SIZES = {"u8": 1, "i8": 1, "c_char": 1, "u32": 4, "i32": 4, "f32": 4, "u64": 8, "i64": 8, "f64": 8, "usize": 8}


def _rs_structs(text):
    consts = {n: int(v) for n, v in re.findall(r"pub const (\w+): \w+ = (\d+);", text)}
    out = {}
    for name, body in re.findall(r"#\[repr\(C\)\]\n#\[derive\(Clone, Copy\)\]\npub struct (\w+) \{\n(.*?)\n\}", text, flags=re.S):
        fields = []
        for fname, ty in re.findall(r"pub (\w+): ([^,\n]+),", body):
            arr = re.match(r"\[(\w+); (\w+)\]", ty)
            if arr:
                elem, n = arr.group(1), consts[arr.group(2)] if not arr.group(2).isdigit() else int(arr.group(2))
                size, align = SIZES[elem] * n, SIZES[elem]
            elif ty.startswith("*"):
                size = align = 8
            else:
                size = align = SIZES[ty]
            fields.append((fname, size, align))
        out[name] = fields
    return out


def _repr_c_layout(fields):
    off, max_align, res = 0, 1, []
    for name, size, align in fields:
        off = (off + align - 1) // align * align
        res.append((name, off, size))
        off += size
        max_align = max(max_align, align)
    return res, (off + max_align - 1) // max_align * max_align


def test_committed_file_is_what_the_generator_produces():
    import gen_rust_sys
    assert open(RS).read() == gen_rust_sys.generate(), "stale: run python scripts/gen_rust_sys.py"


def test_struct_layouts_match_a_c_probe_of_the_header(tmp_path):
    structs = _rs_structs(open(RS).read())
    assert {"mi355_index_desc", "mi355_search_params", "mi355_flat_desc", "mi355_stats", "mi355_flat_stats", "mi355_comm_stats",
            "mi355_encode_desc", "mi355_kmeans_desc", "mi355_pq_train_desc"} == set(structs)
    lines = ['#include "mi355_ann.h"', "#include <stddef.h>", "#include <stdio.h>", "int main(void) {"]
    for s, fields in structs.items():
        lines.append(f'  printf("{s} %zu\\n", sizeof({s}));')
        for f, _, _ in fields:
            lines.append(f'  printf("{s}.{f} %zu %zu\\n", offsetof({s}, {f}), sizeof((({s}*)0)->{f}));')
    lines += ["  return 0;", "}"]
    c = tmp_path / "probe.c"
    c.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", os.path.dirname(HEADER), str(c), "-o", str(exe)], check=True)
    got = {}
    for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        k, *v = line.split()
        got[k] = tuple(int(x) for x in v)
    # the C side has no field the Rust side lacks: count the header's declarators per struct
    hdr = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    for s, fields in structs.items():
        layout, size = _repr_c_layout(fields)
        assert got[s] == (size,), (s, got[s], size)
        for f, off, sz in layout:
            assert got[f"{s}.{f}"] == (off, sz), (s, f, got[f"{s}.{f}"], off, sz)
        body = re.search(r"typedef\s+struct\s+" + s + r"\s*\{(.*?)\}\s*" + s + r"\s*;", hdr, flags=re.S).group(1)
        n_decl = sum(len(d.split(",")) for d in body.split(";") if d.strip())
        assert n_decl == len(fields), (s, n_decl, len(fields))


def test_function_set_and_argument_counts_match_the_header_and_the_library():
    from lancedb_amd import _abi
    text = open(RS).read()
    rs = {n: (0 if not a.strip() else len(a.split(","))) for n, a in re.findall(r"pub fn (mi355_\w+)\(([^)]*)\) -> \w+;", text)}
    hdr = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    h = {}
    for n, a in re.findall(r"\bu?int32_t\s+(mi355_\w+)\s*\(([^)]*)\)\s*;", hdr, flags=re.S):
        a = " ".join(a.split())
        h[n] = 0 if a == "void" else len(a.split(","))
    assert rs == h
    assert set(rs) == set(_abi.EXPORTED_SYMBOLS) and len(rs) == 40
    # every pointer the header marks const is `*const` on the Rust side (spot check of the constness mapping)
    assert "pub fn mi355_search(index: *mut mi355_index, queries: *const f32, n_queries: u32, params: *const mi355_search_params, " \
           "out_rowids: *mut u64, out_dist: *mut f32, out_counts: *mut u32) -> i32;" in text
    assert "pub fn mi355_index_open(desc: *const mi355_index_desc, out: *mut *mut mi355_index) -> i32;" in text


def test_a_reordered_field_is_caught(tmp_path):
    """The check has teeth: swap two fields of one struct in a copy of the .rs and the layout comparison must fail."""
    text = open(RS).read()
    bad = text.replace("    pub n_rows: u64,\n    pub mem: u32,\n    pub codes_layout: u32,", "    pub mem: u32,\n    pub n_rows: u64,\n    pub codes_layout: u32,", 1)
    assert bad != text
    layout_ok, _ = _repr_c_layout(_rs_structs(text)["mi355_index_desc"])
    layout_bad, _ = _repr_c_layout(_rs_structs(bad)["mi355_index_desc"])
    assert layout_ok != layout_bad
