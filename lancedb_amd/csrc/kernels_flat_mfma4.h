// kernels_flat_mfma4.h — the flat filter GEMM, persistent FOUR-SLOT schedule (two 32-MFMA phases
// per k-tile, two staggered wave groups).  Same tile, operands, MFMA k-order and epilogue
// arithmetic as k_flat_gemm8 (kernels_flat_mfma8.h); flat KNN = SURVEY.md §8a row a17.
//
// Why (profiles/r02_c_flat_gemm_ablations.json, 4 M x 768 rows, 1024 queries, GEMM us): the 8-phase
// kernel 6152; without its MFMAs 5981; without the epilogue 5505; without LDS-DMA 5336; without
// barriers 5922; barriers + loop + epilogue alone 2775.  The matrix work is completely hidden —
// what is exposed is the synchronisation skeleton: 8 slots per k-tile at ~170 cycles of barrier
// hand-shake and loop bookkeeping each (34 % of the kernel), the epilogue's `s_waitcnt vmcnt(0)`
// (its ordinary loads drain the prefetched LDS-DMA: 10 %), and the DMA issue in the L sections (13 %).
// This schedule halves the slots, issues every DMA piece among the MFMAs and feeds the epilogue from
// LDS (its inputs arrive by LDS-DMA early in the tile, so the epilogue waits on lgkmcnt only).
//
// Geometry as before: 256 x 256 tile, BK = 64, 8 waves = 2 (rows, wr) x 4 (queries, wc), a wave owns
// 128 x 64 outputs (acc 128 VGPRs); 2 LDS buffers x (A 32 KiB + B 32 KiB), 128-B rows, chunk index
// XOR (row & 7) on the global source; + 6 KiB for two slots of epilogue inputs.
//
// Phases of k-tile g (buffer g & 1), the same code for both groups:
//   A:  L: reads A (all 8 row tiles x 2 k-halves = 16) + B0 (query tiles 0-1, 4 reads); prepares the
//          addresses of B-h1(g+1)
//       barrier; M: 32 MFMAs (8 x 2 x 2) with the 2 pieces of B-h1(g+1) issued after the 2nd and 4th
//       barrier
//   B:  L: s_waitcnt vmcnt(0)  (k-tile g+1 has landed: this wave's pieces); reads B1 (4) into B0's
//          registers; prepares A(g+2) (4 pieces) and B-h0(g+2) (2)
//       barrier; M: 32 MFMAs with those 6 pieces issued after every 4th
//       barrier
// Regions: A = all 256 A rows; B-h0 = B rows {wc*64 + 0..31}, B-h1 = {wc*64 + 32..63}.
// Barrier instances: group 0 (wr = 0) Bm_A(g) = 4g+1, Be_A = 4g+2, Bm_B = 4g+3, Be_B = 4g+4; group 1
// runs one instance behind.  Slot s = between instances s and s+1:
//   group 0: L_A 4g, M_A 4g+1, L_B 4g+2, M_B 4g+3;   group 1: L_A 4g+1, M_A 4g+2, L_B 4g+3, M_B 4g+4.
// Reads complete at the head of the M section (lgkmcnt(0)): A and B-h0 of k-tile g are dead from
// instance 4g+3, B-h1 from 4g+5.
// WAR  B-h1(g+1) -> other buffer's B-h1, dead from 4(g-1)+5 = 4g+1; issued in M_A(g), slot >= 4g+1.
//      A(g+2), B-h0(g+2) -> this buffer, dead from 4g+3; issued in M_B(g), slot >= 4g+3.
// RAW  k-tile g+1 is first read in L_A(g+1) (group 0: slot 4g+4).  Every wave waits vmcnt(0) for its
//      pieces in L_B(g) (slots 4g+2 / 4g+3), i.e. before instance 4g+4, which all waves pass before the
//      first read: wait and first read are in different phases with a barrier in between (the rule for
//      staggered groups).  A and B-h0 of g+1 were issued 4 slots before their wait (HBM latency);
//      B-h1(g+1) only one slot before — it is query data, hot in the XCD's L2 (250-400 cycles).
// Past the end of the walk the pieces re-read valid memory into dead regions (as in k_flat_gemm8's
// DMA_IN_M form) and are drained before the kernel ends.
//
// Epilogue inputs: at the first L_A of a tile wave 0 issues vv[row0 .. row0+255] (1 KiB), waves 1 / 2
// the tile's qa / qg (256 floats each) as LDS-DMA pieces into the input slot of the tile's parity.  They are
// older than every piece issued afterwards, so the L_B wait covers them, and they are visible to all
// waves behind that phase's barrier — long before the tile's epilogue (KT >= 2).  The slot written for
// tile t was last read by tile t-2's epilogue.
#pragma once
#include "kernels_flat_mfma8.h"

template <int METRIC, int EPI>
__global__ __launch_bounds__(512, 2) void k_flat_gemm4(FlatGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BM = 256, BN = 256, MI = 8, NI = 4, WN = 4;
  constexpr int A_BYTES = BM * FG_BK * 2, B_BYTES = BN * FG_BK * 2, BUF = A_BYTES + B_BYTES;
  constexpr int EPI_OFF = 2 * BUF, EPI_SLOT = 3072;  // two slots of {vv [256], qa [256], qg [256]} f32
  static_assert(FG_BK == 64, "128-B LDS rows");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / WN, wc = wid % WN;
  const uint32_t KT = a.dimp / FG_BK;  // >= 2 (checked by the launcher)
  const uint32_t pitch = a.dimp * 2u;
  const uint32_t total_vb = ((a.n_rtiles + 7u) / 8u) * 8u * a.n_qtiles;

  struct TileRef {
    const unsigned char* baseA;
    const unsigned char* baseB;
    uint64_t row0;
    uint32_t q0, rt, lim;
  };
  auto decode = [&](uint32_t v, TileRef& t) -> uint32_t {
    while (v < total_vb) {
      const uint32_t slot = v >> 3;
      const uint32_t qt = slot % a.n_qtiles, rt = (slot / a.n_qtiles) * 8u + (v & 7u);
      if (rt < a.n_rtiles) {
        t.rt = rt;
        t.row0 = (uint64_t)rt * BM;
        t.q0 = qt * BN;
        t.baseA = (const unsigned char*)a.v + t.row0 * pitch;
        t.baseB = (const unsigned char*)a.qb + (size_t)t.q0 * pitch;
        const uint64_t left = a.n_rows - t.row0;
        t.lim = left >= (uint64_t)BM ? (uint32_t)(BM - 1) : (uint32_t)(left - 1);
        break;
      }
      v += gridDim.x;
    }
    return v;
  };
  TileRef cur, nxt;
  uint32_t vb = decode(blockIdx.x, cur);
  if (vb >= total_vb) return;
  nxt = cur;
  uint32_t nvb = decode(vb + gridDim.x, nxt);
  bool has_next = nvb < total_vb;

  const uint32_t l8 = (uint32_t)lane >> 3;
  const uint32_t swz = (((uint32_t)lane & 7u) ^ l8) * 16u;
  const uint32_t voffB = l8 * pitch + swz;
  const uint32_t voffA = (wid * 8 + l8) * pitch + swz;

  // ---- pieces: A piece i (0..3) of a wave = rows i*64 + wid*8 + lane/8; B-h piece i (0..1) as in gemm8.
  // Every piece is a wave-uniform base + ONE per-lane 32-bit offset (no per-piece address registers).
  // Rows past the end of a ragged last tile are NOT clamped: the column (and its bf16 shadow) is
  // allocated with 256 rows of padding (mi355_flat_open), their scores are masked in the epilogue.
  auto a_src = [&](const TileRef& t, uint32_t koff, int i) -> const unsigned char* {
    return t.baseA + (koff + (uint32_t)i * 64u * pitch) + (size_t)voffA;
  };
  auto a_lds = [&](uint32_t buf, int i) -> uint32_t { return buf * BUF + (uint32_t)(i * 64 + wid * 8) * 128; };
  auto b_row0 = [&](int h, int i) -> uint32_t {
    const uint32_t g2 = 2 * wid + i;
    return (g2 >> 2) * 64 + h * 32 + (g2 & 3u) * 8;
  };
  auto b_src = [&](const TileRef& t, uint32_t koff, int h, int i) -> const unsigned char* {
    return t.baseB + (koff + b_row0(h, i) * pitch) + (size_t)voffB;
  };
  auto b_lds = [&](uint32_t buf, int h, int i) -> uint32_t { return buf * BUF + A_BYTES + b_row0(h, i) * 128; };
  // k-tile `ahead` (1 | 2) after (tile cur, k-tile u): which tile and byte offset along k; past the end
  // of the walk: k-tile 0 of the last tile (valid memory, dead destination)
  auto kt_of = [&](uint32_t u, uint32_t ahead, const TileRef*& t) -> uint32_t {
    const uint32_t kt0 = u + ahead;
    t = kt0 < KT ? &cur : &nxt;
    return (kt0 < KT ? kt0 : has_next ? kt0 - KT : 0u) * (FG_BK * 2);
  };

  fg_f32x4 acc[MI][NI];
  auto zero_acc = [&]() {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = fg_f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  const uint32_t fr = lane & 15, fk = lane >> 4;
  const uint32_t offA0 = (wr * MI * 16 + fr) * 128, offB0 = A_BYTES + (wc * NI * 16 + fr) * 128;
  const uint32_t sw = fr & 7u;
  const uint32_t ch0 = (fk ^ sw) << 4, ch1 = ((4 + fk) ^ sw) << 4;

  // ---- epilogue of one finished tile; its inputs come from LDS (slot `ep`)
  auto epilogue = [&](const TileRef& t, uint32_t ep) {
    const float* vvs = (const float*)(smem + EPI_OFF + ep * EPI_SLOT);
    const float* qas = vvs + 256;
    const float* qgs = vvs + 512;
    float qa[NI], qg[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      qa[ni] = qas[wc * NI * 16 + ni * 16 + fr];
      qg[ni] = qgs[wc * NI * 16 + ni * 16 + fr];
    }
    const bool full = t.lim == (uint32_t)(BM - 1);  // wave-uniform
#pragma unroll
    for (int g = 0; g < MI / 2; ++g) {
      const uint32_t lr = wr * MI * 16 + g * 32 + fk * 4;  // row inside the tile
      const uint64_t rg = t.row0 + lr;
      const float4 va = *(const float4*)(vvs + lr), vb4 = *(const float4*)(vvs + lr + 16);
      const float vvr[2][4] = {{va.x, va.y, va.z, va.w}, {vb4.x, vb4.y, vb4.z, vb4.w}};
      float outv[NI];
      if (EPI == 0 || !full) {
        float gmin[NI], chk[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          gmin[ni] = __builtin_huge_valf();
          chk[ni] = 0.f;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int mi = g * 2 + h;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const bool live = rg + h * 16 + reg < a.n_rows;
            const float vv = vvr[h][reg];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              const float s = acc[mi][ni][reg];
              float lo;
              if (METRIC == MI355_METRIC_L2)
                lo = qa[ni] + a.omc * vv + qg[ni] * s;
              else if (METRIC == MI355_METRIC_COSINE)
                lo = qa[ni] + qg[ni] * s * (1.0f / sqrtf(vv));
              else
                lo = qa[ni] - s - qg[ni] * sqrtf(vv);
              lo = live ? lo : __builtin_huge_valf();
              gmin[ni] = fminf(gmin[ni], lo);
              chk[ni] = __fmaf_rn(live ? lo : 0.f, 0.f, chk[ni]);
            }
          }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) outv[ni] = chk[ni] == chk[ni] ? gmin[ni] : -__builtin_huge_valf();
      } else {
        float w[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const float vv = vvr[h][reg];
            w[h][reg] = METRIC == MI355_METRIC_L2 ? a.omc * vv : METRIC == MI355_METRIC_COSINE ? 1.0f / sqrtf(vv) : sqrtf(vv);
          }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          float ext = METRIC == MI355_METRIC_L2 ? __builtin_huge_valf() : -__builtin_huge_valf();
          float sum = 0.f;
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              const float s = acc[g * 2 + h][ni][reg];
              float tt;
              if (METRIC == MI355_METRIC_L2) {
                tt = __fmaf_rn(-2.0f, s, w[h][reg]);
                ext = fminf(ext, tt);
              } else if (METRIC == MI355_METRIC_COSINE) {
                tt = s * w[h][reg];
                ext = fmaxf(ext, tt);
              } else {
                tt = __fmaf_rn(qg[ni], w[h][reg], s);
                ext = fmaxf(ext, tt);
              }
              sum += tt;
            }
          float v;
          if (METRIC == MI355_METRIC_L2)
            v = qa[ni] + ext;
          else if (METRIC == MI355_METRIC_COSINE)
            v = qa[ni] + qg[ni] * ext;
          else
            v = qa[ni] - ext;
          outv[ni] = ((sum - sum) == 0.f && (v - v) == 0.f) ? v : -__builtin_huge_valf();
        }
      }
      const uint32_t grp = t.rt * (BM / 32) + wr * (MI / 2) + g;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float v = outv[ni];
        v = fminf(v, __shfl_xor(v, 16));
        v = fminf(v, __shfl_xor(v, 32));
        if (fk == 0) a.gm[(size_t)grp * a.nq_pad + t.q0 + wc * NI * 16 + ni * 16 + fr] = v;
      }
    }
  };
  // the epilogue inputs of tile `t` into slot `ep`: one piece each from waves 0 (vv), 1 (qa), 2 (qg)
  auto stage_epi_inputs = [&](const TileRef& t, uint32_t ep) {
    if (wid == 0)
      fg_glds16((const unsigned char*)(a.vv + t.row0) + lane * 16, smem + EPI_OFF + ep * EPI_SLOT);
    else if (wid == 1)
      fg_glds16((const unsigned char*)(a.qa + t.q0) + lane * 16, smem + EPI_OFF + ep * EPI_SLOT + 1024);
    else if (wid == 2)
      fg_glds16((const unsigned char*)(a.qg + t.q0) + lane * 16, smem + EPI_OFF + ep * EPI_SLOT + 2048);
  };

  // ---- prologue: k-tile 0 entirely, then A(1) and B-h0(1) (what M_B(-1) would have issued)
  stage_epi_inputs(cur, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) fg_glds16(a_src(cur, 0, i), smem + a_lds(0, i));
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) fg_glds16(b_src(cur, 0, h, i), smem + b_lds(0, h, i));
#pragma unroll
  for (int i = 0; i < 4; ++i) fg_glds16(a_src(cur, FG_BK * 2, i), smem + a_lds(1, i));
#pragma unroll
  for (int i = 0; i < 2; ++i) fg_glds16(b_src(cur, FG_BK * 2, 0, i), smem + b_lds(1, 0, i));
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // k-tile 0 (and the epilogue inputs) landed
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind

  fg_bf16x8 fa[2][MI], fb[2][2];  // A: all row tiles; B: the phase's two query tiles
  uint32_t u = 0, par = 0, ep = 0;  // k-tile inside the tile; parity of the global k-tile; epilogue-input slot
  bool pending = false, first_tile = true;
  TileRef done = cur;
  uint32_t done_ep = 0;
  // 32 MFMAs: all row tiles x query tiles ni0, ni0+1 x both k-halves; `issue(j)` after MFMA 2j+2 (first)
  // or 4j+4 (spread), j < n_pieces
  auto mfma_block = [&](int ni0, auto issue, int n_pieces, bool early) {
    int n = 0, j = 0;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          acc[mi][ni0 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][mi], fb[kk][ni], acc[mi][ni0 + ni], 0, 0, 0);
          ++n;
          if (j < n_pieces && n == (early ? 2 * j + 2 : 4 * j + 4)) {
            __builtin_amdgcn_sched_barrier(0);
            issue(j);
            __builtin_amdgcn_sched_barrier(0);
            ++j;
          }
        }
    // pin the block (hipcc sinks trailing MFMAs below the closing barrier otherwise)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) asm volatile("" ::"v"(acc[mi][ni0 + ni]));
  };

  while (true) {
    const unsigned char* sb = smem + par * BUF;
    if (pending) {  // the finished tile's epilogue, under the other group's MFMAs
      epilogue(done, done_ep);
      zero_acc();
      pending = false;
    }
    // ---------------- phase A: reads A (16) + B0 (4); M: x B0, issuing B-h1(g+1)
    if (u == 0 && !first_tile) stage_epi_inputs(cur, ep);  // (the first tile's came with the prologue)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fb[0][i] = *(const fg_bf16x8*)(sb + offB0 + i * 2048 + ch0);
      fb[1][i] = *(const fg_bf16x8*)(sb + offB0 + i * 2048 + ch1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      fa[0][i] = *(const fg_bf16x8*)(sb + offA0 + i * 2048 + ch0);
      fa[1][i] = *(const fg_bf16x8*)(sb + offA0 + i * 2048 + ch1);
    }
    __builtin_amdgcn_sched_barrier(0);
    const TileRef* t1;
    const uint32_t koff1 = kt_of(u, 1, t1);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
    mfma_block(0, [&](int j) { fg_glds16(b_src(*t1, koff1, 1, j), smem + b_lds(par ^ 1u, 1, j)); }, 2, true);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase B: k-tile g+1 has landed; reads B1 (4); M: x B1, issuing A(g+2), B-h0(g+2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fb[0][i] = *(const fg_bf16x8*)(sb + offB0 + (2 + i) * 2048 + ch0);
      fb[1][i] = *(const fg_bf16x8*)(sb + offB0 + (2 + i) * 2048 + ch1);
    }
    __builtin_amdgcn_sched_barrier(0);
    const TileRef* t2;
    const uint32_t koff2 = kt_of(u, 2, t2);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
    mfma_block(2, [&](int j) {
      if (j < 4)
        fg_glds16(a_src(*t2, koff2, j), smem + a_lds(par, j));
      else
        fg_glds16(b_src(*t2, koff2, 0, j - 4), smem + b_lds(par, 0, j - 4));
    }, 6, false);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    par ^= 1u;
    if (++u == KT) {  // tile finished: its epilogue runs at the head of the next phase (or below)
      done = cur;
      done_ep = ep;
      pending = true;
      if (!has_next) break;
      u = 0;
      ep ^= 1u;
      first_tile = false;
      cur = nxt;
      vb = nvb;
      nvb = decode(vb + gridDim.x, nxt);
      has_next = nvb < total_vb;
    }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();  // group 0's extra barrier: every wave executed the same count
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the walk's last (unused) pieces
  epilogue(done, done_ep);
}
