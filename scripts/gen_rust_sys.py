#!/usr/bin/env python3
"""Generate integration/mi355_sys.rs — the Rust `extern "C"` bindings of include/mi355_ann.h — from the header itself.

The reference has no FFI on this path (rust/lancedb/src/table.rs:549-576 is a trait object); a maintainer who puts the
engine behind `BaseTable` / the ANN plan nodes adds exactly this `-sys` module plus the shim of INTEGRATION.md §3.  No
Rust toolchain exists in this image, so the file cannot be compiled here; instead it is GENERATED (this script) and
CHECKED (tests/test_rust_binding.py: regenerated text == committed text, and an independent parse of the .rs against a
compiled C probe of the header: same functions, same argument counts, same struct field order, offsets and sizes).

usage: python scripts/gen_rust_sys.py [--check]      (--check: exit 1 if the committed file is stale)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mi355_ann.h")
OUT = os.path.join(ROOT, "integration", "mi355_sys.rs")

PRIM = {"uint8_t": "u8", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "int64_t": "i64", "float": "f32", "double": "f64",
        "size_t": "usize", "char": "c_char", "void": "c_void"}


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def rust_type(ctype, structs):
    """C declarator type (without the name) -> Rust."""
    t = " ".join(ctype.replace("*", " * ").split())
    toks = t.split()
    const = False
    base = None
    ptrs = []  # constness of each pointer level, innermost first
    for tok in toks:
        if tok == "const":
            const = True
        elif tok == "struct":
            continue
        elif tok == "*":
            ptrs.append(const)
            const = False
        else:
            base = tok
    r = PRIM.get(base, base)
    assert r is not None, ctype
    if base not in PRIM:
        assert base in structs, f"unknown type {base!r} in {ctype!r}"
    for is_const in ptrs:
        r = ("*const " if is_const else "*mut ") + r
    return r


def parse(src):
    src = strip_comments(src)
    consts, structs, opaque, funcs = [], {}, [], []
    for name, val in re.findall(r"#define\s+(MI355_[A-Z0-9_]+)\s+([0-9xXa-fA-F]+)u?\s*$", src, flags=re.M):
        consts.append((name, int(val, 0)))
    for body in re.findall(r"\benum\s*\{(.*?)\}\s*;", src, flags=re.S):
        nxt = 0
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, val = (x.strip() for x in item.split("="))
                nxt = int(val.rstrip("uU"), 0)
            else:
                name = item
            consts.append((name, nxt))
            nxt += 1
    for name in re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", src):
        opaque.append(name)
    known = set(opaque)
    for name, body in re.findall(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\1\s*;", src, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(.*?)(\w+(?:\s*\[\s*\w+\s*\])?(?:\s*,\s*\w+)*)$", decl)
            ctype, names = m.group(1).strip(), m.group(2)
            for nm in names.split(","):
                nm = nm.strip()
                arr = re.match(r"(\w+)\s*\[\s*(\w+)\s*\]", nm)
                if arr:
                    fields.append((arr.group(1), ctype, arr.group(2)))
                else:
                    fields.append((nm, ctype, None))
        structs[name] = fields
        known.add(name)
    proto = re.compile(r"\b(u?int32_t)\s+(mi355_\w+)\s*\(([^)]*)\)\s*;", flags=re.S)
    for ret, name, args in proto.findall(src):
        params = []
        args = " ".join(args.split())
        if args != "void":
            for a in args.split(","):
                a = a.strip()
                m = re.match(r"(.*?)(\w+)$", a)
                params.append((m.group(2), m.group(1).strip()))
        funcs.append((name, ret, params))
    return consts, structs, opaque, funcs, known


def generate():
    consts, structs, opaque, funcs, known = parse(open(HEADER).read())
    cmap = dict(consts)
    o = []
    w = o.append
    w("// mi355_sys.rs — GENERATED from include/mi355_ann.h by scripts/gen_rust_sys.py; do not edit by hand.")
    w("// Raw `extern \"C\"` bindings of libmi355_ann.so for the Rust shim of INTEGRATION.md (the `BaseTable` /")
    w("// `ExecutionPlan` side of rust/lancedb/src/table.rs:549-576 and table/query.rs:131-328).  Never compiled in this")
    w("// repository (no Rust toolchain in the image); tests/test_rust_binding.py checks it against the header instead:")
    w("// identical function set, argument counts, struct field order, field offsets and sizes.")
    w("#![allow(non_camel_case_types, non_upper_case_globals, dead_code)]")
    w("use core::ffi::{c_char, c_void};")
    w("")
    w(f"// ---- constants ({len(consts)})")
    for name, val in consts:
        ty = ("usize" if name in ("MI355_COMM_ID_BYTES", "MI355_MAX_RANKS") else
              "i32" if name == "MI355_OK" or name.startswith("MI355_ERR_") else "u32")  # statuses compare with the i32 returns
        w(f"pub const {name}: {ty} = {val};")
    w("")
    w("// ---- opaque handles")
    for name in opaque:
        w("#[repr(C)]")
        w(f"pub struct {name} {{")
        w("    _private: [u8; 0],")
        w("}")
    w("")
    w("// ---- descriptors and statistics (plain old data, `struct_size` = size_of::<Self>() as u32)")
    for name, fields in structs.items():
        w("#[repr(C)]")
        w("#[derive(Clone, Copy)]")
        w(f"pub struct {name} {{")
        for fname, ctype, arr in fields:
            rt = rust_type(ctype, known)
            if arr:
                rt = f"[{rt}; {arr}]"
            w(f"    pub {fname}: {rt},")
        w("}")
    w("")
    w(f"// ---- entry points ({len(funcs)}); every status is 0 = ok, 1 InvalidInput, 2 Runtime, 3 Timeout, 4 NotSupported")
    w("// (rust/lancedb/src/error.rs:55-145); the message of a failing call: mi355_last_error on the same thread")
    w('#[link(name = "mi355_ann")]')
    w('extern "C" {')
    for name, ret, params in funcs:
        ps = ", ".join(f"{pn}: {rust_type(pt, known)}" for pn, pt in params)
        w(f"    pub fn {name}({ps}) -> {PRIM[ret]};")
    w("}")
    w("")
    w("/// `lancedb::Error` of a non-zero status (error.rs:55-145), with the library's message for this thread.")
    w("pub fn status_to_error(status: i32) -> Option<(i32, String)> {")
    w("    if status == 0 {")
    w("        return None;")
    w("    }")
    w("    let mut buf = [0 as c_char; 512];")
    w("    unsafe { mi355_last_error(buf.as_mut_ptr(), buf.len()) };")
    w("    let msg = unsafe { std::ffi::CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned();")
    w("    Some((status, msg))")
    w("}")
    assert cmap["MI355_ANN_ABI_VERSION"] >= 5
    return "\n".join(o) + "\n"


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("integration/mi355_sys.rs is stale: run python scripts/gen_rust_sys.py", file=sys.stderr)
            sys.exit(1)
        sys.exit(0)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT, len(text.splitlines()), "lines")
