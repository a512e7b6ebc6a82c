#!/usr/bin/env python3
"""Generates lancedb_amd/csrc/skew_chunks.inc: the hand-scheduled gfx950 inner
blocks of the skewed ADC scan (kernels_skew.h).

One block = 16 steps of one wave (one global_load_dwordx4 worth of code bytes
per lane).  All 16 LUT gathers are issued first, then the 16 f32 adds retire in
step order behind counted lgkmcnt waits, so the LDS pipe always has >= 16
gathers of this wave in flight and the adds keep the j-ascending order of the
arithmetic contract.  hipcc does not model anything inside an asm statement, so
every wait is counted here (cdna_hip_programming.md §5.7).

Three block kinds:
  SPLIT0  steps  0..15 of a tile: every step adds into X on the lanes that are
          already on the new row and into Y on the others (compile-time EXEC mask)
  SPLIT1  steps 16..31: 16..30 split as above, step 31 is X only
  PLAIN   steps >= 32: X only

Knobs (the defaults are what measured fastest with scripts/ubench_skew_loop.py;
the numbers are in DESIGN.md):
  addr  how the gather address (code << 9 | lane column origin) is formed
        "bfe"   v_bfe_u32 + v_mad_u32_u24            (2 VOP3)
        "vop2"  shift + and + or                     (3 VOP2)
  il    issue-phase interleave: the address ops of `il` consecutive steps are
        emitted stage by stage, so no VALU op directly follows its producer
  drop  timing ablations (results are garbage): "lds" no gathers, "addr" no
        address math, "add" no adds
  wait  adds retire in groups of `wait` behind one s_waitcnt (1, 2, 4, 8, 16)
  split how EXEC is set for the X/Y split
        "lit"   two s_mov_b32 literals, lanes lm <= t in both 32-lane halves
        "bfm"   one s_bfm_b64: needs the mirrored lane->phase map
                (phase = l for l < 32, 63 - l above), Y lanes are contiguous
"""
import os
import sys

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lancedb_amd", "csrc",
                   "skew_chunks.inc")

DEFAULTS = dict(addr="bfe", wait=2, split="bfm", il=16, drop="none")


def body(first_step, split_until, addr="bfe", wait=1, split="lit", il=1, drop="none"):
    """asm text for steps first_step .. first_step+15; steps < split_until split X/Y."""
    lines = []
    for e0 in range(0, 16, il):  # `il` independent steps are interleaved: no back-to-back dependent VALU
        stages = [[], [], [], []]
        for e in range(e0, e0 + il):
            w, b = e // 4, e % 4
            if addr == "bfe":
                if b == 3:
                    stages[0].append(f"v_lshrrev_b32 %[t{e}], 24, %[w{w}]")
                else:
                    stages[0].append(f"v_bfe_u32 %[t{e}], %[w{w}], {8 * b}, 8")
                stages[1].append(f"v_mad_u32_u24 %[t{e}], %[t{e}], %[pb], %[lb]")
            elif addr == "vop2":
                sh = 9 - 8 * b
                if sh > 0:
                    stages[0].append(f"v_lshlrev_b32 %[t{e}], {sh}, %[w{w}]")
                else:
                    stages[0].append(f"v_lshrrev_b32 %[t{e}], {-sh}, %[w{w}]")
                stages[1].append(f"v_and_b32 %[t{e}], %[pb], %[t{e}]")  # pb holds the mask 0x1fe00 here
                stages[2].append(f"v_or_b32 %[t{e}], %[t{e}], %[lb]")
            else:
                raise ValueError(addr)
            stages[3].append(f"ds_read_b32 %[t{e}], %[t{e}] offset:%c[ob]+{4 * e}")
            if drop == "addr":  # ablation: gather from the lane's column origin, no address math
                stages[0], stages[1], stages[2] = stages[0][:-1], stages[1][:-1], stages[2][: max(0, len(stages[2]) - 1)]
                stages[3][-1] = f"ds_read_b32 %[t{e}], %[lb] offset:%c[ob]+{4 * e}"
            if drop == "lds":  # ablation: no gather (the add consumes the address)
                stages[3].pop()
        for st in stages:
            lines += st
    in_split = False
    for e in range(16):
        t = first_step + e
        if e % wait == 0 and drop != "lds":
            lines.append(f"s_waitcnt lgkmcnt({16 - e - wait})")
        if drop == "add":  # ablation: gathers only
            continue
        if t < split_until:
            if split == "lit":
                mask = (2 << t) - 1
                lines.append(f"s_mov_b32 exec_lo, 0x{mask:x}")
                lines.append(f"s_mov_b32 exec_hi, 0x{mask:x}")
                lines.append(f"v_add_f32 %[x], %[x], %[t{e}]")
                lines.append("s_not_b64 exec, exec")
                lines.append(f"v_add_f32 %[y], %[y], %[t{e}]")
            else:  # mirrored map: lanes t+1 .. 62-t are still on the old row
                lines.append(f"s_bfm_b64 exec, {62 - 2 * t}, {t + 1}")
                lines.append(f"v_add_f32 %[y], %[y], %[t{e}]")
                lines.append("s_not_b64 exec, exec")
                lines.append(f"v_add_f32 %[x], %[x], %[t{e}]")
            in_split = True
        else:
            if in_split:
                lines.append("s_mov_b64 exec, -1")
                in_split = False
            lines.append(f"v_add_f32 %[x], %[x], %[t{e}]")
    if in_split:
        lines.append("s_mov_b64 exec, -1")
    return lines


def emit(name, first_step, split_until, doc, **kw):
    lines = body(first_step, split_until, **kw)
    txt = "\n".join(f'      "{l}\\n\\t"' for l in lines)
    temps_decl = ", ".join(f"t{e}" for e in range(16))
    outs = ", ".join(f'[t{e}] "=&v"(t{e})' for e in range(16))
    return f"""
// {doc}
template <int OB>
__device__ __forceinline__ void {name}(const uint4& cw, uint32_t lb, uint32_t pb, float& x, float& y) {{
  uint32_t {temps_decl};
  asm volatile(
{txt}
      : [x] "+v"(x), [y] "+v"(y), {outs}
      : [w0] "v"(cw.x), [w1] "v"(cw.y), [w2] "v"(cw.z), [w3] "v"(cw.w), [lb] "v"(lb), [pb] "s"(pb), [ob] "i"(OB)
      : "scc");
}}
"""


# --------------------------------------------------------------------------
# dual mode: two rows per lane (chains A and B = two streams of the partition)
# --------------------------------------------------------------------------
# One block = 8 steps of BOTH chains: 16 gathers in flight, the adds of the two
# chains pair up in v_pk_add_f32 (half the add instructions, and consecutive adds
# no longer depend on each other).  The gather address is built by ONE SDWA
# byte insert into a per-lane address register R:
#     R = [bit 16: slab][byte 1: code][byte 0: 4 * (32 - phase)],  ds_read R offset:4t
# i.e. address = slab*65536 + code*256 + 4u with u = t + 32 - phase the table
# column.  The table is two 64 KiB slabs of 256-B rows: columns u < 64 in slab 0,
# columns u >= 64 in slab 1, where the carry of 4u into the code byte simply lands
# one row further (the LUT builder stores slab 1 with that formula).  A lane
# crosses from slab 0 to slab 1 at step 32 + phase, so steps 32..63 set bit 16
# under an EXEC mask for the lanes that have crossed; the bit is cleared once per
# tile.  Temps are the fixed registers v[112:127] (pairs: even = A, odd = B).
T0 = 112


DUAL = dict(nreg=1, wait=1, nomask=0)  # nomask: timing ablation (wrong results): no EXEC games at all


def dblock(k, sdwa=True):
    """asm lines for steps 8k .. 8k+7 of both chains."""
    nreg, wait = DUAL["nreg"], DUAL["wait"]
    regs = ["%[r]", "%[r]"] if nreg == 1 else ["%[r]", "%[r2]"]
    lines = []
    for e in range(8):
        t = 8 * k + e
        w, b = e // 4, e % 4
        if 32 <= t < 64 and not DUAL["nomask"]:  # lanes with phase <= t - 32 are in slab 1 from this step on
            p = t - 32
            if p < 31:
                lines.append(f"s_bfm_b64 exec, {62 - 2 * p}, {p + 1}")  # lanes that have NOT crossed (mirrored map)
                lines.append("s_not_b64 exec, exec")
            lines.append("v_or_b32 %[r], %[slab], %[r]")
            if nreg == 2:
                lines.append("v_or_b32 %[r2], %[slab], %[r2]")
            lines.append("s_mov_b64 exec, -1")
        if nreg == 1:
            for ch, wn in ((0, "a"), (1, "b")):
                lines.append(f"v_mov_b32_sdwa %[r], %[w{wn}{w}] dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_{b}")
                lines.append(f"ds_read_b32 v{T0 + 2 * e + ch}, %[r] offset:{4 * t}")
        else:  # both inserts first: neither gather directly follows its producer
            for ch, wn in ((0, "a"), (1, "b")):
                lines.append(f"v_mov_b32_sdwa {regs[ch]}, %[w{wn}{w}] dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_{b}")
            for ch in (0, 1):
                lines.append(f"ds_read_b32 v{T0 + 2 * e + ch}, {regs[ch]} offset:{4 * t}")
    in_split = False
    for e in range(8):
        t = 8 * k + e
        if e % wait == 0:
            lines.append(f"s_waitcnt lgkmcnt({16 - 2 * (e + wait)})")
        pair = f"v[{T0 + 2 * e}:{T0 + 2 * e + 1}]"
        if t < 31 and not DUAL["nomask"]:  # lanes t+1 .. 62-t (mirrored map) are still on the old row
            lines.append(f"s_bfm_b64 exec, {62 - 2 * t}, {t + 1}")
            lines.append(f"v_pk_add_f32 %[y], %[y], {pair}")
            lines.append("s_not_b64 exec, exec")
            lines.append(f"v_pk_add_f32 %[x], %[x], {pair}")
            in_split = True
        else:
            if in_split:
                lines.append("s_mov_b64 exec, -1")
                in_split = False
            lines.append(f"v_pk_add_f32 %[x], %[x], {pair}")
    if in_split:
        lines.append("s_mov_b64 exec, -1")
    return lines


def emit_dblock(k):
    lines = dblock(k)
    txt = "\n".join(f'      "{l}\\n\\t"' for l in lines)
    clob = ", ".join(f'"v{T0 + i}"' for i in range(16))
    return f"""
// steps {8 * k}..{8 * k + 7} of both chains
__device__ __forceinline__ void skew_dblock_{k}(uint32_t wa0, uint32_t wa1, uint32_t wb0, uint32_t wb1, uint32_t& r,
                                               uint32_t& r2, uint32_t slab, sk_f32x2& x, sk_f32x2& y) {{
  asm volatile(
{txt}
      : [x] "+v"(x), [y] "+v"(y), [r] "+v"(r), [r2] "+v"(r2)
      : [wa0] "v"(wa0), [wa1] "v"(wa1), [wb0] "v"(wb0), [wb1] "v"(wb1), [slab] "v"(slab)
      : "scc", {clob});
}}
"""


def render_dual():
    src = ["// GENERATED by scripts/gen_skew_chunks.py dual=1 - do not edit by hand.\n"
           "// 8-step x 2-chain blocks of the skewed ADC scan; see the generator's docstring.\n"
           "#pragma once\n#define SK_DUAL 1\n#define SK_SPLIT_BFM 1\n"
           "typedef __attribute__((ext_vector_type(2))) float sk_f32x2;\n"
           "typedef __attribute__((ext_vector_type(4))) unsigned int sk_u32x4;  // a native vector: valid asm operand\n"]
    for k in range(12):
        src.append(emit_dblock(k))
    src.append("""
// The code stream is loaded by asm (hipcc would drain every prefetch at the loop head:
// it cannot count them across the selection's rare row-id loads).  sk_load2 issues the
// two 16-B loads of one dual chunk; sk_wait_codes<N> waits until at most N of the
// loads issued after this chunk's are outstanding and names the registers "+v" so that
// no compiler copy of them can be scheduled above the wait (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void sk_load2(sk_u32x4& a, sk_u32x4& b, const void* p) {
  asm volatile("global_load_dwordx4 %0, %2, off\\n\\tglobal_load_dwordx4 %1, %2, off offset:1024"
               : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void sk_wait_codes(sk_u32x4& a, sk_u32x4& b) {
  asm volatile("s_waitcnt vmcnt(%c2)" : "+v"(a), "+v"(b) : "i"(N) : "memory");
}

// blocks of chunk G (steps 16G .. 16G+15) of both chains; ca / cb = the 16 code bytes of each chain
template <int G>
__device__ __forceinline__ void skew_dchunk(const sk_u32x4& ca, const sk_u32x4& cb, uint32_t& r, uint32_t& r2, uint32_t slab,
                                            sk_f32x2& x, sk_f32x2& y) {
""")
    for g in range(6):
        src.append(f"  if constexpr (G == {g}) {{\n    skew_dblock_{2 * g}(ca.x, ca.y, cb.x, cb.y, r, r2, slab, x, y);\n"
                   f"    skew_dblock_{2 * g + 1}(ca.z, ca.w, cb.z, cb.w, r, r2, slab, x, y);\n  }}\n")
    src.append("}\n")
    return "".join(src)


def render(**kw):
    cfg = dict(DEFAULTS)
    cfg.update(kw)
    pb_note = ("the table pitch in bytes (512)" if cfg["addr"] == "bfe" else "the code-field mask 0x1fe00")
    src = ["// GENERATED by scripts/gen_skew_chunks.py - do not edit by hand.\n"
           "// 16-step blocks of the skewed ADC scan; see the generator's docstring.\n"
           f"// knobs: addr={cfg['addr']} wait={cfg['wait']} split={cfg['split']} il={cfg['il']};  `pb` is {pb_note}.\n"
           "#pragma once\n"
           f"#define SK_ADDR_{cfg['addr'].upper()} 1\n"
           f"#define SK_SPLIT_{cfg['split'].upper()} 1\n"]
    src.append(emit("skew_chunk_split0", 0, 16, "steps 0..15 of a tile: X on the lanes already on the new row, Y on the others", **cfg))
    src.append(emit("skew_chunk_split1", 16, 31, "steps 16..31 of a tile: 16..30 split, 31 into X", **cfg))
    src.append(emit("skew_chunk_plain", 32, 0, "steps >= 32 of a tile: X only", **cfg))
    return "".join(src)


def main():
    kw = {}
    for a in sys.argv[1:]:
        k, v = a.split("=")
        kw[k] = int(v) if v.isdigit() else v
    dual = kw.pop("dual", 1)  # the committed skew_chunks.inc is the dual form; dual=0 gives the one-row blocks
    for kk in ("nreg", "dwait", "nomask"):
        if kk in kw:
            DUAL["wait" if kk == "dwait" else kk] = kw.pop(kk)
    with open(OUT, "w") as f:
        f.write(render_dual() if dual else render(**kw))
    print("wrote", OUT, kw or DEFAULTS)


if __name__ == "__main__":
    main()
