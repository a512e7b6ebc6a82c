#!/usr/bin/env python3
"""Dev microbenchmark (run on the GPU box): issue rate of the VALU / LDS
instructions the skewed ADC scan is made of.  Generates a .hip, builds it with
hipcc and runs it: `python scripts/ubench_valu_rates.py`."""
import os, subprocess, sys, tempfile

def rep(lines, n=8):
    return "".join(l + "\\n " for l in lines) * 1, n

OPS = []  # (name, wave-instructions per asm block, asm lines for ONE block (repeated 8x), extra operands)
def op(name, n_instr, lines, extra='"v"(b0)', clob=""):
    OPS.append((name, n_instr, lines, extra, clob))

R = range(8)
op("v_fma_f32", 64, [f"v_fma_f32 %{i}, %{i}, %8, %8" for i in R])
op("v_add_f32", 64, [f"v_add_f32 %{i}, %{i}, %8" for i in R])
op("v_bfe_u32", 64, [f"v_bfe_u32 %{i}, %8, {8*(i%3)}, 8" for i in R])
op("v_mad_u32_u24 (sgpr)", 64, [f"v_mad_u32_u24 %{i}, %{i}, %9, %8" for i in R], '"v"(b0), "s"(seed)')
op("v_lshl_add_u32", 64, [f"v_lshl_add_u32 %{i}, %{i}, 9, %8" for i in R])
op("v_mov_b32_sdwa byte insert", 64, [f"v_mov_b32_sdwa %{i}, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_{i%4}" for i in R])
op("v_lshlrev_b32_sdwa byte", 64, [f"v_lshlrev_b32_sdwa %{i}, %9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_{i%4}" for i in R], '"v"(b0), "v"(9u)')
op("v_and_b32", 64, [f"v_and_b32 %{i}, %{i}, %8" for i in R])
op("v_or_b32", 64, [f"v_or_b32 %{i}, %{i}, %8" for i in R])
op("v_lshrrev_b32", 64, [f"v_lshrrev_b32 %{i}, 24, %8" for i in R])
op("v_pk_add_f32 (2 adds)", 64, [f"v_pk_add_f32 %[p{i%2}], %[p{i%2}], %[q{i%2}]" for i in R], "PK")
op("ds_read_b32 conflict-free x8", 64, [f"ds_read_b32 %{i}, %8 offset:{256*i}" for i in R] + ["s_waitcnt lgkmcnt(0)"], '"v"(ad)')
op("ds_read_b32 x16 in flight", 128, [f"ds_read_b32 %{i}, %8 offset:{256*i}" for i in R] + [f"ds_read_b32 %{i}, %8 offset:{256*i+2048}" for i in R] + ["s_waitcnt lgkmcnt(0)"], '"v"(ad)')
op("ds_read_b64 conflict-free x8", 64, [f"ds_read_b64 %[p{i%2}], %[ad] offset:{512*i}" for i in R] + ["s_waitcnt lgkmcnt(0)"], "PKAD")
split = []
for m in (0xff, 0xfff, 0xffff, 0xfffff):
    split += [f"s_mov_b32 exec_lo, 0x{m:x}", f"s_mov_b32 exec_hi, 0x{m:x}", "v_add_f32 %0, %0, %8", "s_not_b64 exec, exec", "v_add_f32 %1, %1, %8"]
split.append("s_mov_b64 exec, -1")
op("split add x4 (8 v_add + 13 salu)", 8 * 4, split, '"v"(b0)', '"scc"')
g = []
for e in range(4):
    g += [f"v_bfe_u32 %{e}, %8, {5*e}, 5", f"v_mad_u32_u24 %{e}, %{e}, %9, %10", f"ds_read_b32 %{e}, %{e} offset:{4*e}"]
for e in range(4):
    g += [f"s_waitcnt lgkmcnt({3-e})", f"v_add_f32 %7, %7, %{e}"]
op("gather step x4 (bfe,mad,ds,add)", 8 * 4, g, '"v"(b0), "s"(512u), "v"(lb)')
g2 = []
for e in range(4):
    g2 += [f"v_mov_b32_sdwa %{e}, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_{e}", f"ds_read_b32 %{4+e}, %{e} offset:{4*e}"]
for e in range(4):
    g2 += [f"s_waitcnt lgkmcnt({3-e})", f"v_add_f32 %7, %7, %{4+e}"]
op("gather step x4 (sdwa,ds,add)", 8 * 4, g2, '"v"(b0)')


# ---- realistic 16-in-flight gather blocks (per-lookup cost is what matters) ----
blk = []
for e in range(16):
    blk += [f"v_bfe_u32 %{e%8}, %8, {(8*(e%3))}, 5" if e < 8 else f"v_bfe_u32 v{100+e}, %8, {(8*(e%3))}, 5"]
# U1: current kernel block: bfe + mad + ds per step, then wait + dependent add per step (temps v100..v115)
u1 = []
for e in range(16):
    u1 += [f"v_bfe_u32 v{100+e}, %8, {8*(e%4) if e%4<3 else 24}, 5", f"v_mad_u32_u24 v{100+e}, v{100+e}, %9, %10", f"ds_read_b32 v{100+e}, v{100+e} offset:{4*e}"]
for e in range(16):
    u1 += [f"s_waitcnt lgkmcnt({15-e})", f"v_add_f32 %0, %0, v{100+e}"]
CL16 = ", ".join(f'"v{100+e}"' for e in range(16))
op("U1 block16: bfe+mad+ds, dep add", 16, u1, '"v"(b0), "s"(512u), "v"(lb)', CL16)
# U2: dual chain, SDWA insert into ONE shared address register, pk_add of (A,B) pairs (v100..v115 = pairs)
u2 = []
for e in range(8):
    u2 += [f"v_mov_b32_sdwa %1, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_{e%4}", f"ds_read_b32 v{100+2*e}, %1 offset:{4*e}",
           f"v_mov_b32_sdwa %1, %9 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_{e%4}", f"ds_read_b32 v{101+2*e}, %1 offset:{4*e}"]
for e in range(8):
    u2 += [f"s_waitcnt lgkmcnt({14-2*e})", f"v_pk_add_f32 %[p0], %[p0], v[{100+2*e}:{101+2*e}]"]
op("U2 dual16: sdwa(shared R)+ds, pk_add", 16, u2, 'U2', CL16)
# U3: dual chain with two address registers
u3 = []
for e in range(8):
    u3 += [f"v_mov_b32_sdwa %1, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_{e%4}", f"v_mov_b32_sdwa %2, %9 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_{e%4}",
           f"ds_read_b32 v{100+2*e}, %1 offset:{4*e}", f"ds_read_b32 v{101+2*e}, %2 offset:{4*e}"]
for e in range(8):
    u3 += [f"s_waitcnt lgkmcnt({14-2*e})", f"v_pk_add_f32 %[p0], %[p0], v[{100+2*e}:{101+2*e}]"]
op("U3 dual16: sdwa(RA,RB)+ds, pk_add", 16, u3, 'U2', CL16)
# U4: U2 plus an exec-masked v_or per step pair (the slab-bit update of the mixed steps)
u4 = []
for e in range(8):
    u4 += ["s_mov_b32 exec_lo, 0x100", "s_mov_b32 exec_hi, 0x800000", "v_or_b32 %1, 0x10000, %1", "s_mov_b64 exec, -1",
           f"v_mov_b32_sdwa %1, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_{e%4}", f"ds_read_b32 v{100+2*e}, %1 offset:{4*e}",
           f"v_mov_b32_sdwa %1, %9 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_{e%4}", f"ds_read_b32 v{101+2*e}, %1 offset:{4*e}"]
for e in range(8):
    u4 += [f"s_waitcnt lgkmcnt({14-2*e})", f"v_pk_add_f32 %[p0], %[p0], v[{100+2*e}:{101+2*e}]"]
op("U4 dual16: U2 + masked slab-bit or", 16, u4, 'U2', CL16)
# U5: dual chain, bfe+mad addressing, pk_add
u5 = []
for e in range(8):
    u5 += [f"v_bfe_u32 v{100+2*e}, %8, {8*(e%4) if e%4<3 else 24}, 5", f"v_bfe_u32 v{101+2*e}, %9, {8*(e%4) if e%4<3 else 24}, 5",
           f"v_mad_u32_u24 v{100+2*e}, v{100+2*e}, %10, %11", f"v_mad_u32_u24 v{101+2*e}, v{101+2*e}, %10, %11",
           f"ds_read_b32 v{100+2*e}, v{100+2*e} offset:{4*e}", f"ds_read_b32 v{101+2*e}, v{101+2*e} offset:{4*e}"]
for e in range(8):
    u5 += [f"s_waitcnt lgkmcnt({14-2*e})", f"v_pk_add_f32 %[p0], %[p0], v[{100+2*e}:{101+2*e}]"]
op("U5 dual16: bfe+mad+ds, pk_add", 16, u5, 'U5', CL16)
# U6: single chain with SDWA (one address register), dependent adds
u6 = []
for e in range(16):
    u6 += [f"v_mov_b32_sdwa %1, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_{e%4}", f"ds_read_b32 v{100+e}, %1 offset:{4*e}"]
for e in range(16):
    u6 += [f"s_waitcnt lgkmcnt({15-e})", f"v_add_f32 %0, %0, v{100+e}"]
op("U6 block16: sdwa+ds, dep add", 16, u6, '"v"(b0)', CL16)

SRC = r'''
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %%s at %%d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t* out, int iters, uint32_t seed) {
  __shared__ float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 1024) lds[i] = i;
  __syncthreads();
  uint32_t a0 = (threadIdx.x & 31) * 4, a1 = a0, a2 = a0, a3 = a0, a4 = 9, a5 = 11, a6 = 13, a7 = 15;
  uint32_t b0 = (seed ^ 0x12345678u) & 0x1f1f1f1fu;
  uint32_t ad = (threadIdx.x & 63) * 4, lb = (threadIdx.x & 31) * 4;
  uint64_t p0 = a0, p1 = a1, q0 = seed, q1 = seed * 3ull;
  (void)ad; (void)lb; (void)p0; (void)p1; (void)q0; (void)q1;
  for (int it = 0; it < iters; ++it) {
%s
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)p0 + (uint32_t)p1 == 0x31337) out[threadIdx.x] = a0;
}
template <int OP>
static int run(const char* name, int instr_per_iter, uint32_t* d_out) {
  const int iters = 2000, blocks = 256;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, d_out, 10, 12345u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, d_out, iters, 12345u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double n = 4.0 * iters * instr_per_iter;  // wave-instructions per SIMD (16 waves per CU)
  double ns = ms * 1e6 / n;
  printf("%%-34s %%8.3f ms  %%6.3f ns / wave-instr / SIMD  (%%5.2f cyc @2.4GHz; x4 SIMDs -> %%5.2f cyc/CU)\n", name, ms, ns, ns * 2.4, ns * 2.4 / 4);
  return 0;
}
int main() {
  uint32_t* d_out; CHECK(hipMalloc(&d_out, 4096));
%s
  return 0;
}
'''
bodies, calls = [], []
for i, (name, n_instr, lines, extra, clob) in enumerate(OPS):
    txt = "".join(l + "\\n " for l in lines)
    outs = '"+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)'
    if extra == "PK":
        outs, extra = '[p0] "+v"(p0), [p1] "+v"(p1)', '[q0] "v"(q0), [q1] "v"(q1)'
    elif extra == "U2":
        outs, extra = '"+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "v"(b0), "v"(b0 ^ 0x0f0f0f0fu) , [p0] "+v"(p0)', None
    elif extra == "U5":
        outs, extra = '"+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "v"(b0), "v"(b0 ^ 0x0f0f0f0fu), "s"(512u), "v"(lb), [p0] "+v"(p0)', None
    elif extra == "PKAD":
        outs, extra = '[p0] "+v"(p0), [p1] "+v"(p1)', '[ad] "v"(ad * 2)'
    cl = f" : {clob}" if clob else ""
    if extra is None:  # outs holds "outputs, inputs" with the first 8 read-write then inputs then p0
        parts = outs.split(", ")
        rw = [x for x in parts if x.startswith('"+v"') or x.startswith('[p0]')]
        ins = [x for x in parts if not (x.startswith('"+v"') or x.startswith('[p0]'))]
        # operand numbering: 8 rw a-regs (%0-%7), then p0 LAST among outputs would shift inputs; keep p0 named only
        bodies.append(f'    if (OP == {i}) {{ asm volatile(' + " ".join([f'"{txt}"'] * 8) + ' : ' + ", ".join(rw[:8]) + ' : ' + ", ".join(ins) + ', [p0] "v"(p0)' + (f' : {clob}' if clob else '') + '); }')
        calls.append(f'  run<{i}>("{name}", {n_instr}, d_out);')
        continue
    bodies.append(f'    if (OP == {i}) {{ asm volatile(' + " ".join([f'"{txt}"'] * 8) + f' : {outs} : {extra}{cl}); }}')
    calls.append(f'  run<{i}>("{name}", {n_instr}, d_out);')
src = SRC % ("\n".join(bodies), "\n".join(calls))
d = tempfile.mkdtemp()
open(os.path.join(d, "u.hip"), "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", os.path.join(d, "u.hip"), "-o", os.path.join(d, "u")])
if "--build-only" not in sys.argv:
    subprocess.check_call([os.path.join(d, "u")])
