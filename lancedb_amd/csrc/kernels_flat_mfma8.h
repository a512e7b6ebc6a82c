// kernels_flat_mfma8.h — EXPERIMENTAL 8-phase schedule of the flat filter GEMM.
//
// STATUS: compiles for gfx950, NOT yet run on hardware (round 1 ended without GPU minutes).
// It is reachable only through the dev knob MI355_FLAT_TILE=8 and must be screened with
// MI355_FLAT_SYNC=1 (group-minimum checksum == that of tiles 128 / 256, scripts/ab_flat.sh)
// before it may become a default.  Same operands, same MFMA k-order, same epilogue as
// k_flat_gemm<.., 2, 4, 8, 4, 2>: the group minima must be bit-identical.
//
// Why: PMC on the two-barrier kernel (profiles/r01_k_*) — a 256 x 256 x 768 tile costs ~61 k
// cycles for 24.6 k of MFMA issue, waves parked 36 % of the time; re-ordering inside the
// two-barrier loop, a one-barrier stagger with vmcnt(0) drains and an L2 prefetch all measured
// ~0.  What is left is the structure the CDNA guide describes for its 256^2 template: four
// phases per k-tile (one C quadrant x K = 64 = 16 MFMAs each), fragment reads and ONE half-tile
// of LDS-DMA per phase, a counted vmcnt once per k-tile, the two wave groups staggered by a
// barrier so that one group's reads / DMA issue run under the other group's MFMAs.
//
// Geometry: 256 x 256 tile, BK = 64, 512 threads = 8 waves as 2 (rows, wr) x 4 (queries, wc);
// a wave owns 128 x 64 outputs = 8 x 4 MFMA tiles (acc: 128 VGPRs).  LDS: 2 buffers x (A 32 KiB
// + B 32 KiB), 128-B rows, 16-B chunk index XOR (row & 7) applied on the global SOURCE address
// (LDS-DMA writes lane-linear) — the layout of k_flat_gemm, conflict-free for ds_read_b128.
//
// Quadrants of a wave's output, in phase order q = 1..4 (A0 = MFMA row tiles 0-3, A1 = 4-7,
// B0 = query tiles 0-1, B1 = 2-3):   (A0,B0)  (A0,B1)  (A1,B1)  (A1,B0)
// fragment reads:  q1: B0 (4 reads) + A0 (8)   q2: B1 (4)   q3: A1 (8)   q4: none
// A half-tile is the set of LDS rows read in one phase by ALL waves:
//   A-h0 = rows {0-63, 128-191} (read at q1)   A-h1 = rows {64-127, 192-255} (q3)
//   B-h0 = rows {wc*64 + 0..31}  (q1)          B-h1 = rows {wc*64 + 32..63}   (q2)
// each 128 rows x 128 B = 16 KiB = 2 LDS-DMA pieces per thread.
//
// Phase p (k-tile u = (p-1)/4, q = (p-1)%4 + 1), the same code for both groups:
//     L_p: fragment reads of q;  stage ONE half-tile:  q1: B-h1(u+1)  q2: A-h1(u+1)
//                                                      q3: A-h0(u+2)  q4: B-h0(u+2)
//          q4 only: s_waitcnt vmcnt(4 | 0)   (retires k-tile u+1; the two half-tiles of u+2 fly on)
//     s_barrier                      (Bm)
//     s_waitcnt lgkmcnt(0); setprio 1; 16 MFMAs; setprio 0          (M_p)
//     s_barrier                      (Be)
// Group 1 (wr = 1) runs one barrier behind group 0 (it executes one extra barrier before the
// loop, group 0 one after it).  Number the barrier instances 1, 2, ...: group 0 has Bm_p = 2p-1,
// Be_p = 2p; group 1 has Bm_p = 2p, Be_p = 2p+1.  "Slot s" = between instances s and s+1:
//     group 0: L_p in slot 2p-2, M_p in slot 2p-1;   group 1: L_p in slot 2p-1, M_p in slot 2p.
// Reads of k-tile u are complete (lgkmcnt(0) at the head of M) — group 0: q1 in slot 8u+1,
// q2 8u+3, q3 8u+5; group 1 one slot later.  Hence the regions of k-tile u are dead from
//     A-h0, B-h0: instance 8u+3     B-h1: 8u+5     A-h1: 8u+7
// WAR: a piece staged in L_p is issued no earlier than slot 2p-2 (group 0):
//     q3 of u -> A-h0(u+2): slot 8u+4 >= 8u+3      q4 of u -> B-h0(u+2): slot 8u+6 >= 8u+3
//     q1 of u+1 -> B-h1(u+2): slot 8u+8 >= 8u+5    q2 of u+1 -> A-h1(u+2): slot 8u+10 >= 8u+7
// RAW: k-tile u+1 is first read in L_{4u+5} (group 0: slot 8u+8).  Its last piece is staged in
// L_{4u+2}; every wave waits for its own pieces in L_{4u+4}, i.e. before instance 8u+7 (group 0)
// / 8u+8 (group 1): all pieces have landed when instance 8u+8 releases the first reader.  The
// wait (phase 4u+4) and the first read (phase 4u+5) are in different phases with a barrier that
// every wave has passed in between, as the guide's rule for staggered groups demands.
// vmcnt arithmetic: pieces newer than k-tile u+1's at the wait are A-h0(u+2) (q3) and B-h0(u+2)
// (q4, issued before the wait) = 4 per thread when u+2 < KT, else none.
#pragma once
#include "kernels_flat_mfma.h"

template <int METRIC>
__global__ __launch_bounds__(512, 2) void k_flat_gemm8(FlatGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NT = 512, BM = 256, BN = 256, MI = 8, NI = 4, WN = 4;
  constexpr int A_BYTES = BM * FG_BK * 2, B_BYTES = BN * FG_BK * 2, BUF = A_BYTES + B_BYTES;
  static_assert(FG_BK == 64, "128-B LDS rows");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / WN, wc = wid % WN;
  const uint32_t b = blockIdx.x;
  const uint32_t xcd = b & 7u, slot = b >> 3;
  const uint32_t qt = slot % a.n_qtiles;
  const uint32_t rt = (slot / a.n_qtiles) * 8u + xcd;
  if (rt >= a.n_rtiles) return;
  const uint64_t row0 = (uint64_t)rt * BM;
  const uint32_t q0 = qt * BN;
  const uint32_t KT = a.dimp / FG_BK;
  const size_t pitch = (size_t)a.dimp * 2;
  const unsigned char* baseA = (const unsigned char*)a.v + row0 * pitch;
  const unsigned char* baseB = (const unsigned char*)a.qb + (size_t)q0 * pitch;

  // ---- staging: half-tile = 16 wave pieces of 8 rows; wave w issues pieces i = 0, 1
  // LDS row of (half h, wave w, piece i, lane l):
  //   A: i * 128 + h * 64 + w * 8 + l / 8          (rows {0-63,128-191} / {64-127,192-255})
  //   B: (2 w + i) / 4 * 64 + h * 32 + (2 w + i) % 4 * 8 + l / 8      (rows {wc*64 + h*32 + 0..31})
  uint32_t srcA[2][2], srcB[2][2];  // [half][piece]: byte offset from baseA / baseB (k-tile 0)
  uint32_t ldsA[2][2], ldsB[2][2];  // wave-uniform LDS byte offset of the piece inside a buffer
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t c = lane & 7u;
      const uint32_t ra0 = i * 128 + h * 64 + wid * 8;  // first row of the piece
      const uint32_t ra = ra0 + (lane >> 3);
      uint64_t vr = row0 + ra;
      if (vr >= a.n_rows) vr = a.n_rows - 1;  // clamped; masked in the epilogue
      srcA[h][i] = (uint32_t)((vr - row0) * pitch) + (c ^ (ra & 7u)) * 16u;
      ldsA[h][i] = ra0 * 128;
      const uint32_t g = 2 * wid + i;
      const uint32_t rb0 = (g >> 2) * 64 + h * 32 + (g & 3u) * 8;
      const uint32_t rb = rb0 + (lane >> 3);
      srcB[h][i] = (uint32_t)(rb * pitch) + (c ^ (rb & 7u)) * 16u;
      ldsB[h][i] = A_BYTES + rb0 * 128;
    }
  auto stage_half = [&](bool is_a, int h, uint32_t kt) {  // 2 pieces
    unsigned char* dst = smem + (kt & 1u) * BUF;
    const size_t koff = (size_t)kt * (FG_BK * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (is_a)
        fg_glds16(baseA + koff + srcA[h][i], dst + ldsA[h][i]);
      else
        fg_glds16(baseB + koff + srcB[h][i], dst + ldsB[h][i]);
    }
  };

  fg_f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = fg_f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses (bytes inside a buffer): row * 128 + ((chunk ^ (row & 7)) << 4)
  const uint32_t fr = lane & 15, fk = lane >> 4;
  const uint32_t offA0 = (wr * MI * 16 + fr) * 128, offB0 = A_BYTES + (wc * NI * 16 + fr) * 128;
  const uint32_t sw = fr & 7u;
  const uint32_t ch0 = (fk ^ sw) << 4, ch1 = ((4 + fk) ^ sw) << 4;  // k-half 0 / 1

  // ---- prologue: k-tile 0 entirely, A-h0 and B-h0 of k-tile 1; k-tile 0 must have landed
  stage_half(true, 0, 0);
  stage_half(false, 0, 0);
  stage_half(false, 1, 0);
  stage_half(true, 1, 0);
  if (KT > 1) {
    stage_half(true, 0, 1);
    stage_half(false, 0, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind

  fg_bf16x8 fa[2][MI], fb[2][NI];  // [k-half][tile]; A0 = fa[.][0..3], A1 = fa[.][4..7]
  for (uint32_t u = 0; u < KT; ++u) {
    const unsigned char* sb = smem + (u & 1u) * BUF;
    // ---------------- q1: (A0, B0); stage B-h1(u+1)
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
    __builtin_amdgcn_sched_barrier(0);
    if (u + 1 < KT) stage_half(false, 1, u + 1);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][mi], fb[kk][ni], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- q2: (A0, B1); stage A-h1(u+1)
#pragma unroll
    for (int i = 2; i < 4; ++i) {
      fb[0][i] = *(const fg_bf16x8*)(sb + offB0 + i * 2048 + ch0);
      fb[1][i] = *(const fg_bf16x8*)(sb + offB0 + i * 2048 + ch1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (u + 1 < KT) stage_half(true, 1, u + 1);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 2; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][mi], fb[kk][ni], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- q3: (A1, B1); stage A-h0(u+2)
#pragma unroll
    for (int i = 4; i < 8; ++i) {
      fa[0][i] = *(const fg_bf16x8*)(sb + offA0 + i * 2048 + ch0);
      fa[1][i] = *(const fg_bf16x8*)(sb + offA0 + i * 2048 + ch1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (u + 2 < KT) stage_half(true, 0, u + 2);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mi = 4; mi < 8; ++mi)
#pragma unroll
        for (int ni = 2; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][mi], fb[kk][ni], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- q4: (A1, B0); stage B-h0(u+2); retire k-tile u+1
    if (u + 2 < KT) {
      stage_half(false, 0, u + 2);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // A-h0(u+2), B-h0(u+2) fly on
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mi = 4; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][mi], fb[kk][ni], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();  // group 0's extra barrier: every wave executed 2 + 8 KT

  // ---- epilogue: identical to k_flat_gemm's (lower bound, minimum per 32-row group)
  float qa[NI], qg[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const uint32_t n = q0 + wc * NI * 16 + ni * 16 + fr;
    qa[ni] = a.qa[n];
    qg[ni] = a.qg[n];
  }
  float4 vv4[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) vv4[mi] = *(const float4*)(a.vv + row0 + wr * MI * 16 + mi * 16 + fk * 4);
#pragma unroll
  for (int g = 0; g < MI / 2; ++g) {
    float gmin[NI], chk[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      gmin[ni] = __builtin_huge_valf();
      chk[ni] = 0.f;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int mi = g * 2 + h;
      const uint64_t r0 = row0 + wr * MI * 16 + mi * 16 + fk * 4;
      const float vvr[4] = {vv4[mi].x, vv4[mi].y, vv4[mi].z, vv4[mi].w};
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const bool live = r0 + reg < a.n_rows;
        const float vv = vvr[reg];
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
    const uint32_t grp = rt * (BM / 32) + wr * (MI / 2) + g;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float v = chk[ni] == chk[ni] ? gmin[ni] : -__builtin_huge_valf();
      v = fminf(v, __shfl_xor(v, 16));
      v = fminf(v, __shfl_xor(v, 32));
      if (fk == 0) a.gm[(size_t)grp * a.nq_pad + q0 + wc * NI * 16 + ni * 16 + fr] = v;
    }
  }
}
