// LDS-tiled fp32 GEMM main loop on the CDNA4 fp32-input matrix instruction
// v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain per output).
//
//   C[BM x BN] = sum_k A[row, k] * B[col, k]        (both operands K-contiguous,
//                                                   i.e. activations [M, K] and
//                                                   PyTorch-layout weights [N, K])
//
// One workgroup = 4 MFMA waves arranged WM x WN x WK (+ 4 loader waves, see below); every
// MFMA wave owns one 32x32 accumulator (16 VGPRs).  BM = 32*WM, BN = 32*WN; each main-loop iteration
// consumes KT = 32*WK of K, wave (.,.,wk) taking the wk-th 32-wide slice (intra-
// workgroup split-K, summed through LDS in the epilogue).
//
// Staging: direct-to-LDS loads (global_load_lds_dwordx4, 16 B per lane, 1 KiB per wave
// instruction) into a ring of S stages, S-1 K-tiles in flight.  At ~1 workgroup per CU
// the loads in flight are the only latency hiding there is (Little: ~2 us x ~35 GB/s per
// CU = ~64 KiB), so the ring is deep and never drained: each iteration waits with a
// COUNTED s_waitcnt vmcnt(N) for the oldest tile only, crosses one raw s_barrier, refills
// the stage freed by the previous iteration and runs its 16 MFMAs.  (__syncthreads() would
// drain the ring: it waits vmcnt(0) while an LDS-DMA is pending.)
//
// An LDS-DMA writes wave-uniform base + lane*16, i.e. the LDS image is linear (unpadded
// rows of KT floats).  Bank conflicts of the ds_read_b128 fragment reads are removed by an
// XOR swizzle of the 16-byte column index applied to the per-lane SOURCE address and again
// on the read:  LDS(row, c4) = global(row, c4 ^ swz(row)),  swz = (row>>1)&7 for 128-B rows
// (two rows per 256-B bank row), row&15 for 256/512-B rows.  The 16 lanes of a ds_read_b128
// group then hit 16 distinct 16-B slots.
//
// Out-of-range tile elements (row >= M, k >= K, conv padding) take their source address
// from a 16-byte zero block, so every lane always issues its load and the per-tile
// instruction stream - hence the vmcnt arithmetic - is the same for every tile.
//
// K permutation: within a 32-wide slice lane half h = lane>>5 supplies k = 16h + 4q + e
// for MFMA (q, e); A and B use the same map, so each MFMA's two k-slots pair up
// correctly and every k is consumed exactly once.
#pragma once
#include "common.h"

namespace ttsdec {

template <int WM, int WN, int WK, int S>
struct TileCfg {
  static_assert(WM * WN * WK == 4, "4 MFMA waves per workgroup");
  static_assert(S >= 4, "ring needs at least 4 stages (fragment reads run one tile ahead)");
  static constexpr int BM = 32 * WM;
  static constexpr int BN = 32 * WN;
  static constexpr int KT = 32 * WK;
  static constexpr int C4 = KT / 4;                 // 16-byte columns per tile row
  static constexpr int ROWS_PER_INST = 64 / C4;     // tile rows one wave instruction covers
  static constexpr int NA = BM * KT / 1024;         // glds instructions per wave per A tile
  static constexpr int NB = BN * KT / 1024;
  static constexpr int STAGES = S;
  static constexpr int kStageFloats = (BM + BN) * KT;
  static constexpr int LDO = BN + 1;
  static constexpr int kOutFloats = WK * BM * LDO;
  static constexpr int kRingFloats = S * kStageFloats;
  static constexpr int kLdsFloats = kRingFloats > kOutFloats ? kRingFloats : kOutFloats;
  // loads left in flight when the tile whose fragments are read NEXT (one ahead of the MFMAs) has landed
  static constexpr int kWaitCnt = (S - 3) * (NA + NB);
  static_assert(kWaitCnt <= 63, "vmcnt field");
  __device__ static __forceinline__ int swz(int row) { return KT == 32 ? ((row >> 1) & 7) : (row & 15); }
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_void;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Operand loaders (LoaderA / LoaderB) describe rows of a K-segmented operand:
//   int   nseg() / seglen(s)      uniform: number of K segments and their lengths (the
//                                 torch.cat pieces of the reference); A and B agree on them
//   bool  row_ok(r)               tile row r (0 <= r < BM or BN) exists
//   gfloat* row_ptr(r, s)         address of element k = 0 of segment s in tile row r
//   kRange / k_lo(r) / k_hi(r)    optional per-row valid k window (conv zero padding)
// The K loop walks the segments tile by tile (each segment zero-padded up to a multiple of
// KT), so per tile a lane's source address is just "previous + KT": one 64-bit add per
// load.  A lane whose row or k is out of range reads the 16-byte zero block instead.
//
// Wave roles (512-thread workgroup, two waves per SIMD):
//   waves 0-3  MFMA waves: fragment reads (one tile ahead, double-buffered registers) and
//              the dependent MFMA chain, nothing else in their instruction stream;
//   waves 4-7  loader waves: address updates and LDS-DMA issue.  An LDS-DMA costs its
//              issuing wave ~60-185 cycles (one wave sustains only ~25 GB/s), which would
//              come straight out of the MFMA chain if the MFMA waves issued it; on a
//              co-resident wave the VMEM issue overlaps the other wave's MFMAs.
// Per K tile there is one workgroup barrier: loaders arrive once their part of tile t+1
// has landed (counted vmcnt), MFMA waves once tile t-1's MFMAs are issued; after it the
// loaders refill the stage tile t-1 occupied and the MFMA waves read tile t+1 / run tile t.
//
// After the call `smem` holds the BM x BN result, row-major with leading dimension
// Cfg::LDO, summed over the WK slices, visible to all threads.
template <class Cfg, class LoaderA, class LoaderB>
__device__ __forceinline__ void gemm_tile_f32(const LoaderA& la, const LoaderB& lb, float* smem, int dbg = 0) {
  constexpr int BM = Cfg::BM, BN = Cfg::BN, KT = Cfg::KT, C4 = Cfg::C4, RPI = Cfg::ROWS_PER_INST;
  constexpr int NA = Cfg::NA, NB = Cfg::NB, LDO = Cfg::LDO, S = Cfg::STAGES;
  constexpr int WN_ = BN / 32, WK_ = KT / 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = wave8 >= 4;
  const int wave = wave8 & 3;

  // uniform tile sequencing
  const int nseg = la.nseg();
  const int len0 = la.seglen(0), len1 = la.seglen(1), len2 = la.seglen(2);
  const int nt0 = (len0 + KT - 1) / KT, nt1 = (len1 + KT - 1) / KT, nt2 = (len2 + KT - 1) / KT;
  const int nk = nt0 + (nseg > 1 ? nt1 : 0) + (nseg > 2 ? nt2 : 0);

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int wk = wave % WK_;
  const int wn = (wave / WK_) % WN_;
  const int wm = wave / (WK_ * WN_);
  const int half = lane >> 5;
  const int l32 = lane & 31;

  if (is_loader) {
    // =========================== loader waves ===========================
    // loader w issues instructions i = 0..NA-1 covering A-tile rows (w*NA + i)*RPI + lane/C4;
    // this lane's 16-byte column within the row is (lane % C4) ^ swz(row).
    gfloat *qa0[NA], *qa1[NA], *qa2[NA], *cura[NA];
    gfloat *qb0[NB], *qb1[NB], *qb2[NB], *curb[NB];
    int ca[NA], cb[NB], inca[NA], incb[NB];
    int aklo[NA], akhi[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = (wave * NA + i) * RPI + lane / C4;
      const bool ok = la.row_ok(row) && dbg != 1;
      ca[i] = ((lane % C4) ^ Cfg::swz(row)) * 4;
      qa0[i] = ok ? la.row_ptr(row, 0) + ca[i] : zero_addr();
      qa1[i] = ok ? la.row_ptr(row, 1) + ca[i] : zero_addr();
      qa2[i] = ok ? la.row_ptr(row, 2) + ca[i] : zero_addr();
      inca[i] = ok ? KT : 0;
      cura[i] = qa0[i];
      if (LoaderA::kRange) {
        aklo[i] = ok ? la.k_lo(row) : 0;
        akhi[i] = ok ? la.k_hi(row) : 0;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int row = (wave * NB + i) * RPI + lane / C4;
      const bool ok = lb.row_ok(row) && dbg != 1;
      cb[i] = ((lane % C4) ^ Cfg::swz(row)) * 4;
      qb0[i] = ok ? lb.row_ptr(row, 0) + cb[i] : zero_addr();
      qb1[i] = ok ? lb.row_ptr(row, 1) + cb[i] : zero_addr();
      qb2[i] = ok ? lb.row_ptr(row, 2) + cb[i] : zero_addr();
      incb[i] = ok ? KT : 0;
      curb[i] = qb0[i];
    }
    int seg = 0, left = nt0, seg_len = len0, kpos = 0, istage = 0;

    auto issue_tile = [&]() {
      float* st = smem + istage * Cfg::kStageFloats;
      const bool partial = (kpos + KT > seg_len);  // uniform: last, zero-padded tile of a segment (or past the end)
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        gfloat* p = cura[i];
        if (LoaderA::kRange)  // the seg_len bound also covers the zero-block tiles past the last segment
          p = (kpos + ca[i] >= aklo[i] && kpos + ca[i] < akhi[i] && kpos + ca[i] < seg_len) ? p : zero_addr();
        else
          p = (partial && kpos + ca[i] >= seg_len) ? zero_addr() : p;
        __builtin_amdgcn_global_load_lds((global_void*)p, (lds_void*)(st + (wave * NA + i) * 256), 16, 0, 0);
        cura[i] += inca[i];
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        gfloat* p = (partial && kpos + cb[i] >= seg_len) ? zero_addr() : curb[i];
        __builtin_amdgcn_global_load_lds((global_void*)p, (lds_void*)(st + BM * KT + (wave * NB + i) * 256), 16, 0, 0);
        curb[i] += incb[i];
      }
      istage = (istage + 1 == S) ? 0 : istage + 1;
      kpos += KT;
      if (--left == 0) {  // next segment (or, past the last one, zero-block loads that keep vmcnt uniform)
        ++seg;
        kpos = 0;
        if (seg == 1 && nseg > 1) {
          left = nt1; seg_len = len1;
#pragma unroll
          for (int i = 0; i < NA; ++i) cura[i] = qa1[i];
#pragma unroll
          for (int i = 0; i < NB; ++i) curb[i] = qb1[i];
        } else if (seg == 2 && nseg > 2) {
          left = nt2; seg_len = len2;
#pragma unroll
          for (int i = 0; i < NA; ++i) cura[i] = qa2[i];
#pragma unroll
          for (int i = 0; i < NB; ++i) curb[i] = qb2[i];
        } else {
          left = 0x7fffffff; seg_len = 0;  // every further tile is "partial" with nothing valid
        }
      }
    };

    // tiles 0 .. S-2 in flight
#pragma unroll
    for (int t = 0; t < S - 1; ++t) issue_tile();
    wait_vmcnt<(S - 2) * (NA + NB)>();  // tile 0 landed
    __builtin_amdgcn_s_barrier();       // B0
    for (int t = 0; t < nk; ++t) {
      wait_vmcnt<Cfg::kWaitCnt>();      // this wave's part of tile t+1 has landed
      __builtin_amdgcn_s_barrier();     // B(t+1): the MFMA waves have issued tile t-1's MFMAs, its stage is free
      if (dbg != 3) issue_tile();       // tile t+S-1 into that stage (dbg 3: measurement ablation)
    }
    wait_vmcnt<0>();  // trailing zero-block loads must land before the ring is reused
  } else {
    // ============================ MFMA waves ============================
    const int arow = wm * 32 + l32, brow = wn * 32 + l32;
    int aoff[4], boff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c4 = wk * 8 + half * 4 + q;
      aoff[q] = arow * KT + ((c4 ^ Cfg::swz(arow)) << 2);
      boff[q] = BM * KT + brow * KT + ((c4 ^ Cfg::swz(brow)) << 2);
    }
    // Fragment registers are double-buffered: the reads of tile t+1 are issued right after
    // the barrier that makes it visible and complete underneath tile t's MFMA chain.
    f32x4 fa[2][4], fb[2][4];
    int rstage = 0;
    auto read_frags = [&](auto buf_c) {
      constexpr int buf = decltype(buf_c)::value;
      const float* st = smem + rstage * Cfg::kStageFloats;
      rstage = (rstage + 1 == S) ? 0 : rstage + 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        fa[buf][q] = *reinterpret_cast<const f32x4*>(st + aoff[q]);
        fb[buf][q] = *reinterpret_cast<const f32x4*>(st + boff[q]);
      }
    };
    auto tile_step = [&](auto cur_c) {
      constexpr int cur = decltype(cur_c)::value;
      __builtin_amdgcn_s_barrier();        // B(t+1): tile t+1 is in LDS
      // tile t's fragments were requested a whole tile ago: retire them here (no stall), in a
      // form the compiler's wait-count model sees, so it does not later drain the next reads
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only
      read_frags(std::integral_constant<int, cur ^ 1>{});
      __builtin_amdgcn_sched_barrier(0);
      if (dbg != 4) {  // dbg 4: measurement ablation (no MFMAs)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][q][e], fb[cur][q][e], acc, 0, 0, 0);
        }
      }
    };
    __builtin_amdgcn_s_barrier();  // B0: tile 0 is in LDS
    read_frags(std::integral_constant<int, 0>{});
    for (int t = 0; t < nk; t += 2) {
      tile_step(std::integral_constant<int, 0>{});
      if (t + 1 < nk) tile_step(std::integral_constant<int, 1>{});
    }
  }
  __syncthreads();

  // accumulators -> LDS out tile (aliases the ring).
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
  if (!is_loader) {
    float* out = smem + wk * BM * LDO;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      out[row * LDO + wn * 32 + l32] = acc[r];
    }
  }
  __syncthreads();
  if (WK_ > 1) {
    for (int e = tid; e < BM * BN; e += kGemmThreads) {
      const int row = e / BN, col = e % BN;
      float v = smem[row * LDO + col];
#pragma unroll
      for (int s = 1; s < WK_; ++s) v = add_rn(v, smem[s * BM * LDO + row * LDO + col]);
      smem[row * LDO + col] = v;
    }
    __syncthreads();
  }
}

}  // namespace ttsdec
