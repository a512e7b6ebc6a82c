// kernels_flat_mfma8.h — the flat filter GEMM as a PERSISTENT 8-phase schedule
// (four phases per k-tile, counted vmcnt once per k-tile, two wave groups staggered by a
// barrier), the structure the CDNA guide gives for its 256^2 bf16 template.  Same operands,
// same MFMA k-order as k_flat_gemm<.., 2, 4, 8, 4, 2>; flat KNN = SURVEY.md §8a row a17
// (/root/reference/python/python/lancedb/query.py:1365-1370).
//
// Why: PMC on the two-barrier kernel (profiles/r01_k_*) — a 256 x 256 x 768 tile costs ~61 k
// cycles for 24.6 k of MFMA issue: the LDS-DMA issue (8 pieces per wave per k-step, 60-185
// cycles each), the fragment reads and the barrier skew all run with the matrix pipe idle
// because both waves of a SIMD do the same thing at the same time.  Here the two waves of a
// SIMD belong to different groups that run one barrier apart: one group's reads / DMA issue /
// epilogue run under the other group's 16 MFMAs.
//
// Geometry: 256 x 256 tile, BK = 64, 512 threads = 8 waves as 2 (rows, wr) x 4 (queries, wc);
// a wave owns 128 x 64 outputs = 8 x 4 MFMA tiles (acc: 128 VGPRs).  LDS: 2 buffers x (A 32 KiB
// + B 32 KiB), 128-B rows, 16-B chunk index XOR (row & 7) applied on the global SOURCE address
// (LDS-DMA writes lane-linear) — the layout of k_flat_gemm, conflict-free for ds_read_b128.
//
// Persistent walk: a workgroup walks the virtual block ids vb = blockIdx.x, + gridDim.x, ...
// (k_flat_gemm's XCD-aware order) and treats ALL its tiles as one long k loop over the global
// k-tile index g = tile * KT + u: the stages of g + 1 / g + 2 belong to the next tile when
// they run off the current one, so a tile's first k-tiles fly under the previous tile's last
// MFMAs and epilogue, and the hazard analysis below is uniform in g.  KT >= 2 is required
// (g + 2 may not skip a tile); dim < 128 keeps k_flat_gemm.
//
// Quadrants of a wave's output, in phase order q = 1..4 (A0 = MFMA row tiles 0-3, A1 = 4-7,
// B0 = query tiles 0-1, B1 = 2-3):   (A0,B0)  (A0,B1)  (A1,B1)  (A1,B0)
// fragment reads:  q1: B0 (4 reads) + A0 (8)   q2: B1 (4)   q3: A1 (8)   q4: none
// A half-tile is the set of LDS rows read in one phase by ALL waves:
//   A-h0 = rows {0-63, 128-191} (read at q1)   A-h1 = rows {64-127, 192-255} (q3)
//   B-h0 = rows {wc*64 + 0..31}  (q1)          B-h1 = rows {wc*64 + 32..63}   (q2)
// each 128 rows x 128 B = 16 KiB = 2 LDS-DMA pieces per thread.
//
// Phase p (k-tile g = (p-1)/4, q = (p-1)%4 + 1), the same code for both groups:
//     L_p: fragment reads of q;  stage ONE half-tile:  q1: B-h1(g+1)  q2: A-h1(g+1)
//                                                      q3: A-h0(g+2)  q4: B-h0(g+2)
//          q4 only: s_waitcnt vmcnt(4 | 0)   (retires k-tile g+1; the two half-tiles of g+2 fly on)
//     s_barrier                      (Bm)
//     s_waitcnt lgkmcnt(0); setprio 1; 16 MFMAs; setprio 0          (M_p)
//     s_barrier                      (Be)
// Group 1 (wr = 1) runs one barrier behind group 0 (it executes one extra barrier before the
// loop, group 0 one after it).  Number the barrier instances 1, 2, ...: group 0 has Bm_p = 2p-1,
// Be_p = 2p; group 1 has Bm_p = 2p, Be_p = 2p+1.  "Slot s" = between instances s and s+1:
//     group 0: L_p in slot 2p-2, M_p in slot 2p-1;   group 1: L_p in slot 2p-1, M_p in slot 2p.
// Reads of k-tile g are complete (lgkmcnt(0) at the head of M) — group 0: q1 in slot 8g+1,
// q2 8g+3, q3 8g+5; group 1 one slot later.  Hence the regions of k-tile g are dead from
//     A-h0, B-h0: instance 8g+3     B-h1: 8g+5     A-h1: 8g+7
// WAR: a piece staged in L_p is issued no earlier than slot 2p-2 (group 0):
//     q3 of g -> A-h0(g+2): slot 8g+4 >= 8g+3      q4 of g -> B-h0(g+2): slot 8g+6 >= 8g+3
//     q1 of g+1 -> B-h1(g+2): slot 8g+8 >= 8g+5    q2 of g+1 -> A-h1(g+2): slot 8g+10 >= 8g+7
// RAW: k-tile g+1 is first read in L_{4g+5} (group 0: slot 8g+8).  Its last piece is staged in
// L_{4g+2}; every wave waits for its own pieces in L_{4g+4}, i.e. before instance 8g+7 (group 0)
// / 8g+8 (group 1): all pieces have landed when instance 8g+8 releases the first reader.  The
// wait (phase 4g+4) and the first read (phase 4g+5) are in different phases with a barrier that
// every wave has passed in between, as the guide's rule for staggered groups demands.
// vmcnt arithmetic: pieces newer than k-tile g+1's at the wait are A-h0(g+2) (q3) and B-h0(g+2)
// (q4, issued before the wait) = 4 per thread when k-tile g+2 exists, else none.  Other VMEM
// ops of the wave (the epilogue's loads and stores) are OLDER than those four wherever they
// sit, so the count can only wait for more than it needs, never less.
//
// Epilogue (EPI = 1, the default): the tile's lower bounds in ~1.5 VALU per output instead of
// ~7 — per row one term w_r, per output one fma + one min/max + one add:
//     L2      lo = qa + min_r fma(-2, s, omc * vv_r)
//     cosine  lo = qa + qg * max_r (s / sqrt(vv_r))          (qg = -1/|q| < 0)
//     dot     lo = qa - max_r fma(qg, sqrt(vv_r), s)
// and a group is marked "never filter" (-inf) when the plain SUM of its terms is not finite
// (any NaN / inf score, or an overflowing sum: conservative).  The few extra or fewer roundings
// against EPI = 0 are covered by the dim * 2^-22 accumulation allowance in c_err (4x the
// worst-case f32 accumulation error).  It runs at the head of the NEXT phase's L section, i.e.
// under the other wave group's MFMAs.  EPI = 0 keeps k_flat_gemm's arithmetic bit for bit
// (dev A/B: equal group-minimum checksums).
#pragma once
#include "kernels_flat_mfma.h"

// Dev ablations (scripts/build_flat_ablations.sh builds one side-by-side library per mask; never in
// the product build): 1 = no LDS-DMA after the prologue, 2 = no fragment reads in the loop,
// 4 = no MFMAs, 8 = no epilogue, 16 = no barriers.  Results are wrong by construction; only the
// time is read (cdna guide §5.4 rule 17: values are kept live with empty asm so nothing is DCE'd).
#ifndef MI355_FLAT_ABLATE
#define MI355_FLAT_ABLATE 0
#endif
#define FG_ABL(bit) ((MI355_FLAT_ABLATE & (bit)) != 0)
#define FG_BARRIER()                                     \
  do {                                                   \
    if (!FG_ABL(16)) __builtin_amdgcn_s_barrier();       \
  } while (0)

template <int METRIC, int EPI>
__global__ __launch_bounds__(512, 2) void k_flat_gemm8(FlatGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BM = 256, BN = 256, MI = 8, NI = 4, WN = 4;
  constexpr int A_BYTES = BM * FG_BK * 2, B_BYTES = BN * FG_BK * 2, BUF = A_BYTES + B_BYTES;
  // behind the two stage buffers: the epilogue's operands of the current tile, |v|^2 [256] | qa [256] | qg [256] f32
  constexpr int EPI_OFF = 2 * BUF;
  static_assert(FG_BK == 64, "128-B LDS rows");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / WN, wc = wid % WN;
  const uint32_t KT = a.dimp / FG_BK;  // >= 2 (checked by the launcher)
  const uint32_t pitch = a.dimp * 2u;  // bytes per row of v / qb
  const uint32_t total_vb = ((a.n_rtiles + 7u) / 8u) * 8u * a.n_qtiles;

  // ---- tile walk (k_flat_gemm's XCD-aware order)
  struct TileRef {
    const unsigned char* baseA;
    const unsigned char* baseB;
    uint64_t row0;
    uint32_t q0, rt, lim;  // lim: last valid row of the tile relative to row0 (source rows are clamped)
  };
  auto decode = [&](uint32_t v, TileRef& t) -> uint32_t {  // first virtual block >= v that maps to a row tile
    while (v < total_vb) {
      const uint32_t slot = v >> 3;
      const uint32_t qt = slot % a.n_qtiles, rt = (slot / a.n_qtiles) * 8u + (v & 7u);
      if (rt < a.n_rtiles) {
        t.rt = rt;
        t.row0 = (uint64_t)rt * BM;
        t.q0 = qt * BN;
        t.baseA = (const unsigned char*)a.v + t.row0 * pitch;
        t.baseB = (const unsigned char*)a.qb + (size_t)t.q0 * pitch;
        const uint64_t left = a.n_rows - t.row0;  // >= 1
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

  // ---- staging: half-tile = 16 wave pieces of 8 rows; wave w issues pieces i = 0, 1
  // LDS row of (half h, wave w, piece i, lane l):
  //   A: i * 128 + h * 64 + w * 8 + l / 8          (rows {0-63,128-191} / {64-127,192-255})
  //   B: (2 w + i) / 4 * 64 + h * 32 + (2 w + i) % 4 * 8 + l / 8      (rows {wc*64 + h*32 + 0..31})
  // every piece starts at a multiple of 8 rows, so (row & 7) = l / 8 and the source swizzle is
  // one per-lane constant
  const uint32_t l8 = (uint32_t)lane >> 3;
  const uint32_t swz = (((uint32_t)lane & 7u) ^ l8) * 16u;
  const uint32_t voffB = l8 * pitch + swz;  // B rows are never clamped (queries are padded)
  // kt: k-tile inside tile `t`; buf: LDS buffer (parity of the GLOBAL k-tile index)
  bool abl_dma_off = false;  // ablation 1: set after the prologue
  auto stage_half = [&](bool is_a, int h, const TileRef& t, uint32_t kt, uint32_t buf) {
    if (FG_ABL(1) && abl_dma_off) return;
    unsigned char* dst = smem + buf * BUF;
    const uint32_t koff = kt * (FG_BK * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (is_a) {
        const uint32_t ra0 = i * 128 + h * 64 + wid * 8;
        uint32_t rs = ra0 + l8;
        rs = rs < t.lim ? rs : t.lim;  // rows past the column re-read its last row (masked in the epilogue)
        fg_glds16(t.baseA + koff + (size_t)(rs * pitch + swz), dst + ra0 * 128);
      } else {
        const uint32_t g2 = 2 * wid + i;
        const uint32_t rb0 = (g2 >> 2) * 64 + h * 32 + (g2 & 3u) * 8;
        fg_glds16(t.baseB + (koff + rb0 * pitch) + (size_t)voffB, dst + A_BYTES + rb0 * 128);
      }
    }
  };
  // stage half-tile of the k-tile `ahead` (1 or 2) after (tile cur, k-tile u); returns false when it does not exist
  auto stage_ahead = [&](bool is_a, int h, uint32_t u, uint32_t ahead, uint32_t buf) -> bool {
    const uint32_t kt = u + ahead;
    if (kt < KT) {
      stage_half(is_a, h, cur, kt, buf);
      return true;
    }
    if (!has_next) return false;
    stage_half(is_a, h, nxt, kt - KT, buf);
    return true;
  };

  // The epilogue's operands travel the same way as the tiles: three 1-KiB LDS-DMA pieces (waves 0-2,
  // one each) issued in the tile's LAST k-tile at q3, i.e. BEFORE the two half-tiles that q4's counted
  // wait leaves in flight — that wait retires them for free and the epilogue runs on ds_reads alone.
  // (With plain loads the epilogue's first use forced s_waitcnt vmcnt(0): vmcnt retires in order, so
  // it drained the next tile's prefetched stages, and every 32-row group paid an L2 round trip behind
  // the previous group's stores — four serial round trips per tile with the matrix pipe idle.)
  auto stage_epi = [&](const TileRef& t) {
    if (FG_ABL(1) && abl_dma_off) return;
    if (wid == 0) {
      // cosine / dot with the fast epilogue on a whole tile: the precomputed row factor instead of |v|^2
      const float* rowv = (EPI == 1 && METRIC != MI355_METRIC_L2 && t.lim == (uint32_t)(BM - 1) && a.vw) ? a.vw : a.vv;
      fg_glds16(rowv + t.row0 + (uint32_t)lane * 4u, smem + EPI_OFF);
    }
    else if (wid == 1)
      fg_glds16(a.qa + t.q0 + (uint32_t)lane * 4u, smem + EPI_OFF + 1024);
    else if (wid == 2)
      fg_glds16(a.qg + t.q0 + (uint32_t)lane * 4u, smem + EPI_OFF + 2048);
  };

  fg_f32x4 acc[MI][NI];
  // one phase's 16 MFMAs (rows mi0..mi0+3, queries ni0..ni0+1, both k-halves)
  auto mfma_block = [&](const fg_bf16x8 (&fa)[2][MI], const fg_bf16x8 (&fb)[2][NI], int mi0, int ni0) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          if (FG_ABL(4))
            asm volatile("" ::"v"(fa[kk][mi0 + mi]), "v"(fb[kk][ni0 + ni]));
          else
            acc[mi0 + mi][ni0 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][mi0 + mi], fb[kk][ni0 + ni], acc[mi0 + mi][ni0 + ni], 0, 0, 0);
        }
    // pin the block: hipcc otherwise sinks MFMAs (register-only, no memory semantics) below the
    // closing barrier into the next phase's fragment reads, i.e. into the OTHER group's MFMA slot
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) asm volatile("" ::"v"(acc[mi0 + mi][ni0 + ni]));
  };

  auto zero_acc = [&]() {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = fg_f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  // fragment addresses (bytes inside a buffer): row * 128 + ((chunk ^ (row & 7)) << 4)
  const uint32_t fr = lane & 15, fk = lane >> 4;
  const uint32_t offA0 = (wr * MI * 16 + fr) * 128, offB0 = A_BYTES + (wc * NI * 16 + fr) * 128;
  const uint32_t sw = fr & 7u;
  const uint32_t ch0 = (fk ^ sw) << 4, ch1 = ((4 + fk) ^ sw) << 4;  // k-half 0 / 1

  // ---- epilogue of one finished tile (reads acc)
  auto epilogue = [&](const TileRef& t) {
    // operands from the EPI region by inline-asm ds_reads: hipcc orders a plain LDS read behind every
    // LDS-DMA in flight with s_waitcnt vmcnt(0) (it cannot know the DMA targets another region), which
    // would drain the next tile's stages here; the pieces that filled the region were retired by q4's
    // counted wait a barrier ago
    const uint32_t ebase = (uint32_t)(size_t)smem + EPI_OFF;
    float qa[NI], qg[NI];
    fg_f32x4 vvq[MI / 2][2];
    {
      const uint32_t aq = ebase + 1024u + (wc * NI * 16 + fr) * 4u;
      const uint32_t av = ebase + (wr * MI * 16 + fk * 4) * 4u;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(qa[ni]) : "v"(aq), "n"(ni * 64));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(qg[ni]) : "v"(aq), "n"(1024 + ni * 64));
      }
#pragma unroll
      for (int g = 0; g < MI / 2; ++g) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vvq[g][0]) : "v"(av), "n"(g * 128));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vvq[g][1]) : "v"(av), "n"(g * 128 + 64));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    const bool full = t.lim == (uint32_t)(BM - 1);  // wave-uniform
#pragma unroll
    for (int g = 0; g < MI / 2; ++g) {  // the wave's groups of 32 rows (two MFMA row tiles each)
      const uint64_t rg = t.row0 + wr * MI * 16 + g * 32 + fk * 4;
      const float vvr[2][4] = {{vvq[g][0][0], vvq[g][0][1], vvq[g][0][2], vvq[g][0][3]},
                               {vvq[g][1][0], vvq[g][1][1], vvq[g][1][2], vvq[g][1][3]}};
      float outv[NI];
      if (EPI == 0 || !full) {
        // k_flat_gemm's arithmetic, element by element (also the ragged last row tile of EPI = 1)
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
            w[h][reg] = METRIC == MI355_METRIC_L2 ? a.omc * vv
                        : a.vw ? vv  // (the staged value IS the factor: k_flat_row_factor)
                        : METRIC == MI355_METRIC_COSINE ? 1.0f / sqrtf(vv) : sqrtf(vv);
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
          // a non-finite term (or bound) means "score not representable": never filter this group
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

  // ---- prologue: k-tile 0 entirely, A-h0 and B-h0 of k-tile 1 (KT >= 2); k-tile 0 must have landed
  // (A-h0(1) before B-h0(1): the order the steady state issues them in)
  stage_half(true, 0, cur, 0, 0);
  stage_half(false, 0, cur, 0, 0);
  stage_half(false, 1, cur, 0, 0);
  stage_half(true, 1, cur, 0, 0);
  stage_half(true, 0, cur, 1, 1);
  stage_half(false, 0, cur, 1, 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  abl_dma_off = true;
  FG_BARRIER();
  if (wr == 1) FG_BARRIER();  // group 1 runs one barrier behind

  fg_bf16x8 fa[2][MI], fb[2][NI];  // [k-half][tile]; A0 = fa[.][0..3], A1 = fa[.][4..7]
  uint32_t u = 0, par = 0;         // k-tile inside the current tile; parity of the global k-tile index
  bool pending = false;            // the previous tile's epilogue is still to run (its TileRef is `done`)
  TileRef done = cur;
  bool first = true;  // (ablation 2 reads the fragments once)
  while (true) {
    const unsigned char* sb = smem + par * BUF;
    // the finished tile's epilogue, under the other wave group's MFMAs
    if (pending) {
      if (!FG_ABL(8)) {
        epilogue(done);
        zero_acc();
      }
      pending = false;
    }
    // ---------------- q1: (A0, B0); stage B-h1(g+1)
    if (!FG_ABL(2) || first) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fb[0][i] = *(const fg_bf16x8*)(sb + offB0 + i * 2048 + ch0);
      fb[1][i] = *(const fg_bf16x8*)(sb + offB0 + i * 2048 + ch1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[0][i] = *(const fg_bf16x8*)(sb + offA0 + i * 2048 + ch0);
      fa[1][i] = *(const fg_bf16x8*)(sb + offA0 + i * 2048 + ch1);
    }
    }
    __builtin_amdgcn_sched_barrier(0);
    (void)stage_ahead(false, 1, u, 1, par ^ 1u);
    asm volatile("" ::: "memory");
    FG_BARRIER();
    __builtin_amdgcn_s_setprio(1);
    mfma_block(fa, fb, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    FG_BARRIER();
    // ---------------- q2: (A0, B1); stage A-h1(g+1)
    if (!FG_ABL(2) || first) {
#pragma unroll
    for (int i = 2; i < 4; ++i) {
      fb[0][i] = *(const fg_bf16x8*)(sb + offB0 + i * 2048 + ch0);
      fb[1][i] = *(const fg_bf16x8*)(sb + offB0 + i * 2048 + ch1);
    }
    }
    __builtin_amdgcn_sched_barrier(0);
    (void)stage_ahead(true, 1, u, 1, par ^ 1u);
    asm volatile("" ::: "memory");
    FG_BARRIER();
    __builtin_amdgcn_s_setprio(1);
    mfma_block(fa, fb, 0, 2);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    FG_BARRIER();
    // ---------------- q3: (A1, B1); stage A-h0(g+2)
    if (!FG_ABL(2) || first) {
#pragma unroll
    for (int i = 4; i < 8; ++i) {
      fa[0][i] = *(const fg_bf16x8*)(sb + offA0 + i * 2048 + ch0);
      fa[1][i] = *(const fg_bf16x8*)(sb + offA0 + i * 2048 + ch1);
    }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (u + 1 == KT) stage_epi(cur);  // older than everything q4's wait leaves in flight
    (void)stage_ahead(true, 0, u, 2, par);
    asm volatile("" ::: "memory");
    FG_BARRIER();
    __builtin_amdgcn_s_setprio(1);
    mfma_block(fa, fb, 4, 2);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    FG_BARRIER();
    // ---------------- q4: (A1, B0); stage B-h0(g+2); retire k-tile g+1
    if (stage_ahead(false, 0, u, 2, par))
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // A-h0(g+2), B-h0(g+2) fly on
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FG_BARRIER();
    __builtin_amdgcn_s_setprio(1);
    mfma_block(fa, fb, 4, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    FG_BARRIER();
    par ^= 1u;
    first = false;
    if (++u == KT) {  // tile finished: its epilogue runs at the head of the next phase (or below)
      done = cur;
      pending = true;
      if (!has_next) break;
      u = 0;
      cur = nxt;
      vb = nvb;
      nvb = decode(vb + gridDim.x, nxt);
      has_next = nvb < total_vb;
    }
  }
  if (wr == 0) FG_BARRIER();  // group 0's extra barrier: every wave executed the same count
  if (!FG_ABL(8)) epilogue(done);
}
