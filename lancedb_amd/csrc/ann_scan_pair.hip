// ann_scan_pair.hip — launcher of the generic ADC scan kernel (k_scan_pair: one workgroup per
// (query, probed partition, slice), [sub-quantiser][code] table, any m, 8- or 4-bit codes).
// Its own translation unit so that the kernel families compile in parallel.
#include "ann_internal.h"
#include "kernels_ivfpq.h"

// LDS of one work item: distance table + residual + per-wave candidate lists + wave counters + pass floor
size_t scan_pair_lds(uint32_t m, uint32_t nbits, uint32_t dim, uint32_t lr, uint32_t nt) {
  return (size_t)m * (1u << nbits) * 4 + (((size_t)dim * 4 + 15) & ~(size_t)15) + (size_t)(nt / 64) * lr * 64 * 8 +
         (size_t)(nt / 64) * 4 + 48;
}

// sub-quantisers whose 8-bit tables fit the LDS next to everything else (m itself when all do);
// the spill launch always uses the long lists / 256 threads
uint32_t scan_pair_m_lds(uint32_t m, uint32_t nbits, uint32_t dim) {
  if (scan_pair_lds(m, nbits, dim, 5, 256) <= 160u * 1024) return m;
  if (nbits != 8) return 0;  // 4-bit tables are 64 B per sub-quantiser: not the problem then
  const size_t rest = scan_pair_lds(0, 8, dim, 5, 256);
  return rest + 1024 <= 160u * 1024 ? (uint32_t)((160u * 1024 - rest) / 1024) : 0u;
}

template <int VPT, int LR, int NT, int NBITS, bool MULTI, bool SPILL = false>
static int32_t launch_one(const ScanArgs& sa, dim3 grid, hipStream_t st) {
  auto kern = k_scan_pair<VPT, LR, NT, NBITS, MULTI, SPILL>;
  const size_t lds = scan_pair_lds(SPILL ? sa.m_lds : sa.ix.m, NBITS, sa.ix.dim, LR, NT);
  if (lds > 160u * 1024)
    return fail(MI355_ERR_NOT_SUPPORTED, "scan work item needs %zu B of LDS (> 160 KiB)", lds);
  HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, sa);
  HIP_TRY(hipGetLastError());
  return MI355_OK;
}

// kk <= 64: short lists (LR 2), any block size; kk <= 256: LR 5, 256 threads; beyond: passes of 256
template <int VPT, int NBITS>
static int32_t launch_vpt(const ScanArgs& sa, dim3 grid, hipStream_t st, uint32_t nt) {
  if (sa.kk > SCAN_PASS_ROWS) return launch_one<VPT, 5, 256, NBITS, true>(sa, grid, st);
  if (sa.kk > 64) return launch_one<VPT, 5, 256, NBITS, false>(sa, grid, st);
  if (nt == 256) return launch_one<VPT, 2, 256, NBITS, false>(sa, grid, st);
  if (nt == 512) return launch_one<VPT, 2, 512, NBITS, false>(sa, grid, st);
  if (nt == 1024) return launch_one<VPT, 2, 1024, NBITS, false>(sa, grid, st);
  return fail(MI355_ERR_INVALID_INPUT, "unsupported scan tuning threads=%u", nt);
}

int32_t launch_scan_pair(const ScanArgs& sa, dim3 grid, hipStream_t st, uint32_t vpt, uint32_t nt) {
  if (sa.ix.nbits == 8 && sa.m_lds < sa.ix.m) {  // table tail in global memory
    if (!sa.lut_spill || !sa.m_lds) return fail(MI355_ERR_RUNTIME, "spilled distance table without scratch");
    if (vpt == 4)
      return sa.kk > SCAN_PASS_ROWS ? launch_one<4, 5, 256, 8, true, true>(sa, grid, st)
                                    : launch_one<4, 5, 256, 8, false, true>(sa, grid, st);
    return sa.kk > SCAN_PASS_ROWS ? launch_one<16, 5, 256, 8, true, true>(sa, grid, st)
                                  : launch_one<16, 5, 256, 8, false, true>(sa, grid, st);
  }
  if (sa.ix.nbits == 8) {
    if (vpt == 4) return launch_vpt<4, 8>(sa, grid, st, nt);
    if (vpt == 16) return launch_vpt<16, 8>(sa, grid, st, nt);
  } else if (sa.ix.nbits == 4) {
    if (vpt == 4) return launch_vpt<4, 4>(sa, grid, st, nt);
    if (vpt == 16) return launch_vpt<16, 4>(sa, grid, st, nt);
  }
  return fail(MI355_ERR_INVALID_INPUT, "unsupported scan tuning vpt=%u nbits=%u", vpt, sa.ix.nbits);
}
