// LDS-tiled GEMM main loop for gfx950, three arithmetic modes behind one data path:
//
//   PREC_F32   exact fp32 on v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain per output);
//   PREC_F16S  split-fp16: every operand is two fp16 planes (hi, lo*2^11, see common.h) and
//              a*b = ah*bh + (ah*bl + al*bh)*2^-11 on v_mfma_f32_32x32x16_f16 with fp32
//              accumulation - 3 MFMAs of 32 cycles per 16 k instead of 8 of 64 cycles,
//              ~2^-22 relative per product;
//   PREC_BF16  one bf16 plane per operand on v_mfma_f32_32x32x16_bf16, fp32 accumulation
//              (8 significand bits per operand: for the Postnet's "bf16 MFMA" configuration).
//
//   C[BM x BN] = sum_k A[row, k] * B[col, k]        (both operands K-contiguous,
//                                                   i.e. activations [M, K] and
//                                                   PyTorch-layout weights [N, K])
//
// One workgroup = 4 MFMA waves arranged WM x WN x WK (+ 4 loader waves, below); every MFMA
// wave owns one 32x32 output block.  BM = 32*WM, BN = 32*WN; each main-loop iteration
// consumes one tile row of ROWB = 128*WK bytes per plane (32*WK fp32 or 64*WK fp16 of K),
// wave (.,.,wk) taking the wk-th 128-byte slice (intra-workgroup split-K, summed through
// LDS in the epilogue).
//
// Staging: direct-to-LDS loads (global_load_lds_dwordx4, 16 B per lane, 1 KiB per wave
// instruction) into a ring of S stages.  The ring is never drained: each iteration waits
// with a COUNTED s_waitcnt vmcnt(N) for the oldest tile only and crosses one raw s_barrier.
// (__syncthreads() would drain it: it waits vmcnt(0) while an LDS-DMA is pending.)
//
// An LDS-DMA writes wave-uniform base + lane*16, i.e. the LDS image is linear (unpadded
// rows of ROWB bytes).  Bank conflicts of the ds_read_b128 fragment reads are removed by an
// XOR swizzle of the 16-byte column index applied to the per-lane SOURCE address and again
// on the read:  LDS(row, c) = global(row, c ^ swz(row)),  swz = (row>>1)&7 for 128-B rows
// (two rows per 256-B bank row), row&15 for 256/512-B rows.  The 16 lanes of a ds_read_b128
// group then hit 16 distinct 16-B slots.
//
// Out-of-range tile elements (row >= M, k >= K, conv padding) take their source address
// from a 16-byte zero block, so every lane always issues its load and the per-tile
// instruction stream - hence the vmcnt arithmetic - is the same for every tile.
//
// K order inside a 128-byte slice.  fp32: lane half h = lane>>5 supplies k = 16h + 4q + e
// for MFMA (q, e) (64-byte slices of the lean fp32 tile: k = 8h + 4q + e, q < 2).  fp16: k16-step s uses the 16-byte column 2s + h, i.e. k = 16s + 8h + j.
// A and B use the same map, so every k is consumed exactly once.
#pragma once
#include "common.h"

namespace ttsdec {


// TM x TN > 1 ("big" tiles, 16-bit modes only): every MFMA wave owns TM x TN 32x32 accumulators, so a
// 2x2 arrangement of waves covers 128x128 outputs and each byte staged through LDS feeds twice the
// MFMAs of the 64x64 tile (the large GEMMs are bound by the per-CU ingest rate, not by the matrix
// pipe).  HK = 1 halves the K depth of a stage (64-byte rows) so that four such stages still fit LDS.
// Big tiles read their fragments single-buffered, right after the barrier that publishes the tile.
// HK = 1 with TM = TN = 1 is the "lean" 64x64 tile (16 KiB stages of 32 k, fragments still read one tile
// ahead: <= 80 KiB of LDS and <= 128 VGPRs), built so that TWO workgroups fit a CU - the tile of the LSTMs
// that run beside another role's workgroups in one launch (fused_kernels.hip).
template <int WM, int WN, int WK, int S, int PREC = PREC_F32, int AUXB = 0, int TM = 1, int TN = 1, int HK = 0>
struct TileCfg {
  static constexpr int kAuxB = AUXB;  // cache-policy bits of the B (weight) operand's LDS-DMA: 2 = nt (streamed once)
  static constexpr int NMW = WM * WN * WK;  // active MFMA waves (of the 4 in the workgroup)
  static constexpr int kWM = WM, kWN = WN, kWK = WK, kTM = TM, kTN = TN;
  static constexpr bool kBig = TM * TN > 1;
  // waves per SIMD the register budget is set for: the lean tile is meant to run two 8-wave workgroups per CU (128 VGPRs)
  static constexpr int kWavesPerSimd = (HK && TM * TN == 1) ? 4 : 2;
  static_assert(NMW == 4 || NMW == 2, "2 or 4 active MFMA waves per workgroup");
  static_assert(kBig || S >= 3, "ring needs at least 3 stages (fragment reads run one tile ahead of the MFMAs)");
  static_assert(!kBig || (PREC != PREC_F32 && WK == 1 && NMW == 4 && S >= 2), "big tiles: 16-bit modes, no intra-workgroup split-K");
  // HK = 1: a wave's K slice of a tile is 64 bytes instead of 128.  16-bit modes: no K slices (WK = 1), 32 k per stage.  Exact
  // fp32 (round 4): 16 k per wave and stage, so that FOUR MFMA waves (WK = 4) share a 32 x 32 output block inside 16-KiB stages -
  // the lean fp32 tile of batches that have too few 64-row blocks to give every CU a workgroup (fused_kernels.hip).
  static_assert(!HK || PREC == PREC_F32 || WK == 1, "half-depth stages of the 16-bit modes have no K slices");
  static constexpr int kPrec = PREC;
  static constexpr int EB = (PREC == PREC_F32) ? 4 : 2;      // element bytes
  static constexpr int NP = (PREC == PREC_F16S) ? 2 : 1;     // planes per operand
  static constexpr int BM = 32 * WM * TM;
  static constexpr int BN = 32 * WN * TN;
  static constexpr int ROWB = HK ? 64 * WK : 128 * WK;       // bytes per tile row per plane
  static constexpr int NS16 = HK ? 2 : 4;                    // k16 steps of a 16-bit tile (per K slice)
  static constexpr int NQ32 = HK ? 2 : 4;                    // 16-byte columns of an fp32 tile per lane half (per K slice): 4 k each
  static constexpr int KT = ROWB / EB;                       // K elements per tile
  static constexpr int C16 = ROWB / 16;                      // 16-byte columns per tile row
  static constexpr int ROWS_PER_INST = 64 / C16;             // tile rows one wave instruction covers
  // LDS-DMA instructions per loader wave per A / B plane tile.  A plane tile smaller than 4 KiB (2 KiB: 32 rows of
  // 64 bytes) still costs every loader wave one instruction - the vmcnt arithmetic wants the same count in every
  // wave - but the waves beyond the tile read the zero block into a dummy slot behind the ring.
  static constexpr int NA = BM * ROWB >= 4096 ? BM * ROWB / 4096 : 1;
  static constexpr int NB = BN * ROWB >= 4096 ? BN * ROWB / 4096 : 1;
  static constexpr bool kDummy = BM * ROWB < 4096 || BN * ROWB < 4096;
  static constexpr int NLOADS = NP * (NA + NB);              // per loader wave per tile
  static constexpr int STAGES = S;
  static constexpr int kPlaneABytes = BM * ROWB;
  static constexpr int kPlaneBBytes = BN * ROWB;
  static constexpr int kStageBytes = NP * (kPlaneABytes + kPlaneBBytes);
  static constexpr int LDO = BN + 1;
  static constexpr int kOutBytes = WK * BM * LDO * 4;
  static constexpr int kRingBytes = S * kStageBytes;
  static constexpr int kDummyOff = kRingBytes;  // 4 x 1 KiB dummy slots (one per loader wave) when kDummy
  static constexpr int kRingAll = kRingBytes + (kDummy ? 4096 : 0);
  static constexpr int kLdsBytes = kRingAll > kOutBytes ? kRingAll : kOutBytes;
  static constexpr int kLdsFloats = kLdsBytes / 4;
  // loads left in flight when the tile whose fragments are read NEXT (one ahead of the MFMAs) has landed
  static constexpr int kWaitCnt = (kBig ? S - 2 : S - 3) * NLOADS;
  static_assert((S - 2) * NLOADS <= 63, "vmcnt field");
  static_assert(kLdsBytes <= 160 * 1024, "LDS per workgroup");
  // 16-byte column swizzle: rows are packed 256/ROWB to an LDS bank row; the 16 lanes of a ds_read_b128
  // group (16 consecutive rows, same column) must land in 16 distinct 16-byte slots
  __device__ static __forceinline__ int swz(int row) {
    return ROWB == 64 ? ((row >> 2) & 3) : (ROWB == 128 ? ((row >> 1) & 7) : (row & 15));
  }
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_void;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Operand loaders (LoaderA / LoaderB) describe rows of a K-segmented operand:
//   int   nseg() / seglen(s)       uniform: number of K segments and their lengths in elements
//                                  (the torch.cat pieces of the reference); A and B agree
//   bool  row_ok(r)                tile row r (0 <= r < BM or BN) exists
//   gbyte* row_ptr(r, s, plane)    address of element k = 0 of segment s in tile row r
//                                  (plane 0 = fp32 or fp16 hi, plane 1 = fp16 lo)
//   long  col_off(c16)             byte offset of the row's 16-byte column c16 (c16 * 16 for row-major operands)
//   long  tile_inc(rowb)           pointer advance from one K tile to the next (rowb for row-major operands)
//   kRange / k_lo(r) / k_hi(r)     optional per-row valid k window in elements (conv padding)
// The K loop walks the segments tile by tile (each segment zero-padded up to a multiple of
// KT), so per tile a lane's source address is just "previous + ROWB": one 64-bit add per
// load.  A lane whose row or k is out of range reads the 16-byte zero block instead.
//
// Wave roles (512-thread workgroup, two waves per SIMD):
//   waves 0-3  MFMA waves: fragment reads (one tile ahead, double-buffered registers) and
//              the dependent MFMA chain, nothing else in their instruction stream;
//   waves 4-7  loader waves: address updates and LDS-DMA issue (an LDS-DMA costs its
//              issuing wave ~60-185 cycles; on a co-resident wave that overlaps the MFMAs).
// Per K tile there is one workgroup barrier: loaders arrive once their part of tile t+1
// has landed (counted vmcnt), MFMA waves once tile t-1's MFMAs are issued; after it the
// loaders refill the stage tile t-1 occupied and the MFMA waves read tile t+1 / run tile t.
//
// After the call `smem` holds the BM x BN result, row-major with leading dimension
// Cfg::LDO, summed over the WK slices, visible to all threads.
// Gate (optional): gate.seg is the index of a K segment whose A operand is produced by OTHER workgroups of the
// same launch.  Right before the first tile of that segment is issued, ONE wave (the first loader wave) calls
// gate.wait() - the poll; with kAuxA = 0 also an agent-scope acquire, which invalidates this CU's vector L1 for
// everybody - and the whole workgroup crosses one extra barrier, so no wave loads the segment earlier.  (One poller per workgroup: 1024 waves
// polling one counter saturated its memory channel and slowed the producers - 110 vs 88 us per step.)
// gate.seg = -1: no gate.
// kAuxA: cache policy of the A operand's LDS-DMA (16 = sc1: past this CU's vector L1, for operands handed over inside the launch).
// gate.seg2 (< gate.seg, or -1): a second gated segment, waited for with gate.wait(1) (the one-launch step's decoder LSTM).
struct NoGate {
  static constexpr int kAuxA = 0;
  int seg = -1, seg2 = -1;
  __device__ __forceinline__ void wait(int) const {}
  __device__ __forceinline__ void mark(int) const {}
};

// kBufDma (round 4): the loader waves issue their LDS-DMA as buffer_load_dwordx4 ... lds - a wave-uniform descriptor of the
// K segment's base, a per-lane byte offset that is computed ONCE per segment (row offset + swizzled column) and a SCALAR K
// offset that advances per tile - instead of global_load_lds with a 64-bit per-lane address that every instruction updates
// (two 64-bit adds and a select in the vector ALU per DMA).  Why it matters: a loader wave shares its SIMD with an MFMA wave,
// and beside a dense MFMA chain a partner wave's vector-ALU instructions issue at 8-21 cycles each (tools/
// ubench_f32_partner.hip, ubench_f32_barrier_phase.hip: 64 v_add_u32 beside 16 dependent fp32 MFMAs take 1344 cycles, the chain
// itself 968), so it was the LOADERS' per-tile instruction stream - ~40 VALU + 4 DMA - that set the pace of the fp32 K loop:
// 1725 cycles per 32-k tile for 1024 cycles of MFMA, 640 for the loader waves alone.  With scalar bookkeeping only:
// ~1100 in the micro-benchmark.  Out-of-range lanes (rows past the matrix, the zero padding of a segment's last tile, the
// trailing dummy tiles) carry an offset beyond the descriptor's range: such a lane reads nothing and the DMA writes ZEROS to
// its LDS slot (tools/ubench_buffer_lds_oob.hip; the scalar offset is part of the range check on gfx950, so a valid lane needs
// offset + K advance < kBufRange: operands below 2 GiB - the callers that set kBufDma are the LSTM kernels).
// Loaders used with kBufDma provide  seg_base(s, plane)  (uniform pointer) and  row_off(r, s)  (bytes from it).
// kBufDma = 1: ONE descriptor per operand plane for all K segments (the LSTM kernels: segments of one allocation, their distances
// ride in the scalar offset); kBufDma = 2: a descriptor per K segment, rebuilt where the segment is entered (the generic row /
// conv GEMMs: segments may come from different allocations; based at the workgroup's own first row, so any operand size).
constexpr unsigned kBufRange = 0x7FFFF000u;  // num_records of every descriptor = the offset that marks a lane out of range
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
  // (the pointer is wave-uniform by construction - kernel arguments - but say so: a descriptor the compiler believes to be
  // lane-varying becomes a waterfall loop around every load)
  const unsigned long long a = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, (int)kBufRange, 0x00020000);
}

template <class Cfg, class LoaderA, class LoaderB, class Gate = NoGate, int kBufDma = 0>
__device__ __forceinline__ void gemm_tile(const LoaderA& la, const LoaderB& lb, float* smem, bool live = true,
                                          int dbg = 0, const Gate gate = Gate()) {
  // `live` (does this launch have anything to do?) typically comes from a control-block load
  // that is still in flight: it is first looked at AFTER the loader waves have issued their
  // prologue DMAs, so its latency hides under theirs.  A dead launch drains and falls through.
  constexpr int BM = Cfg::BM, BN = Cfg::BN, KT = Cfg::KT, C16 = Cfg::C16, RPI = Cfg::ROWS_PER_INST;
  constexpr int NA = Cfg::NA, NB = Cfg::NB, NP = Cfg::NP, EB = Cfg::EB, ROWB = Cfg::ROWB;
  constexpr int LDO = Cfg::LDO, S = Cfg::STAGES;
  constexpr int WN_ = Cfg::kWN, WK_ = Cfg::kWK;
  constexpr int EPC = 16 / EB;  // elements per 16-byte column
  char* lds = reinterpret_cast<char*>(smem);
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
  // first tile of the gated segment (uniform), or -1
  auto first_tile = [&](int seg) { return (!live || seg < 0 || seg >= nseg) ? -1 : (seg == 0 ? 0 : (seg == 1 ? nt0 : nt0 + nt1)); };
  const int gate_tile = first_tile(gate.seg), gate_tile2 = first_tile(gate.seg2);
  auto gate_sync1 = [&](int which) {  // every wave of the workgroup, at the same point of the tile sequence
    if (wave8 == 4) gate.wait(which);
    __builtin_amdgcn_s_barrier();
  };
  auto gate_at = [&](int tile) {  // in front of the issue of K tile `tile` (uniform)
    if (tile == gate_tile2) gate_sync1(1);
    if (tile == gate_tile) gate_sync1(0);
  };
  // (segments that start inside the prologue tiles)
  if (gate_tile2 >= 0 && gate_tile2 < S - 1) gate_sync1(1);
  if (gate_tile >= 0 && gate_tile < S - 1) gate_sync1(0);

  f32x16 acc, acc2;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc[i] = 0.f;
    acc2[i] = 0.f;
  }
  const int wk = wave % WK_;
  const int wn = (wave / WK_) % WN_;
  const int wm = wave / (WK_ * WN_);
  const int half = lane >> 5;
  const int l32 = lane & 31;

  if (is_loader && kBufDma) {
    // ====================== loader waves, buffer-descriptor form ======================
    if constexpr (kBufDma) {
    // ONE descriptor per operand plane, based at the lowest of the operand's segment pointers; a segment's distance from it is
    // folded into the per-lane offsets below.  (The segments of an LSTM operand are slices of one allocation - the decoder
    // workspace, the packed weight blob - so the distances are far below the 2-GiB range; api.hip refuses larger ones.)  A
    // descriptor per segment kept 3 x planes x 2 base pointers alive in scalar registers across the K loop: spills.
    auto lowest = [&](const auto& ld_, int p) {
      const char* b0 = static_cast<const char*>(ld_.seg_base(0, p));
      const char* b1 = nseg > 1 ? static_cast<const char*>(ld_.seg_base(1, p)) : b0;
      const char* b2 = nseg > 2 ? static_cast<const char*>(ld_.seg_base(2, p)) : b0;
      const char* m = b1 < b0 ? b1 : b0;
      return b2 < m ? b2 : m;
    };
    // distance of plane p's segment sg from that plane's base: uniform, added to the SCALAR offset (the planes of an operand
    // need not be laid out alike, so it cannot ride in the per-lane offsets that the planes share)
    auto seg_delta = [&](const auto& ld_, int sg, int p) {
      if constexpr (kBufDma == 2) return 0u;  // (a descriptor per segment: no distances)
      const char* b = static_cast<const char*>(ld_.seg_base(sg < nseg ? sg : 0, p));
      return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b - lowest(ld_, p)));
    };
    unsigned dla[NP][3], dlb[NP][3];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int sg = 0; sg < 3; ++sg) {
        dla[p][sg] = seg_delta(la, sg, p);
        dlb[p][sg] = seg_delta(lb, sg, p);
      }
    unsigned va[3][NA], vb[3][NB];  // per-lane byte offset inside a segment (row + swizzled column), per K segment; planes share it
    int ca[NA], cb[NB];             // element offset of the lane's column inside a tile
    // Per-row valid k windows (LoaderA::kRange: conv padding).  The per-lane test - two compares and a select in front of every
    // DMA - is what the buffer form exists to avoid, so each instruction also gets the k range in which ALL its lanes are
    // valid (wave-uniform, computed once): tiles inside it take the test-free path; only the tiles at an utterance's first
    // and last frames take the per-lane one.
    int aklo[NA], akhi[NA], alo_all[NA], ahi_all[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = (wave * NA + i) * RPI + lane / C16;
      const bool ok = row < BM && la.row_ok(row) && dbg != 1 && dbg != 5;  // (dbg 1 / 5 / 6: measurement ablations - all / A / B loads read nothing)
      const int c16 = (lane % C16) ^ Cfg::swz(row);
      ca[i] = c16 * EPC;
#pragma unroll
      for (int sg = 0; sg < 3; ++sg) va[sg][i] = ok ? la.row_off(row, sg) + (unsigned)la.col_off(c16) : kBufRange;
      if constexpr (LoaderA::kRange) {
        aklo[i] = ok ? la.k_lo(row) : 0;
        akhi[i] = ok ? la.k_hi(row) : 0x7fffff;
        // (k < 2^24: exact in fp32, so the wave-wide maximum / minimum can use the DPP float reductions of common.h)
        alo_all[i] = __builtin_amdgcn_readfirstlane((int)wave_max((float)aklo[i]));
        ahi_all[i] = __builtin_amdgcn_readfirstlane((int)-wave_max(-(float)akhi[i]));
        if (!ok) akhi[i] = 0;
      } else {
        aklo[i] = akhi[i] = alo_all[i] = ahi_all[i] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int row = (wave * NB + i) * RPI + lane / C16;
      const bool ok = row < BN && lb.row_ok(row) && dbg != 1 && dbg != 6;
      const int c16 = (lane % C16) ^ Cfg::swz(row);
      cb[i] = c16 * EPC;
#pragma unroll
      for (int sg = 0; sg < 3; ++sg) vb[sg][i] = ok ? lb.row_off(row, sg) + (unsigned)lb.col_off(c16) : kBufRange;
    }
#ifdef TTSDEC_CHECK_DMA
    // Debug build (tools/build_check_dma.sh): every lane's first address of every segment and plane against the per-lane
    // pointer form; a lane that disagrees is taken out of range (reads nothing) and reported - never dereferenced.
#pragma unroll
    for (int sg = 0; sg < 3; ++sg) {
      if (sg >= nseg) continue;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const int row = (wave * NA + i) * RPI + lane / C16;
          const int c16 = (lane % C16) ^ Cfg::swz(row);
          if (va[sg][i] != kBufRange && (gbyte*)(kBufDma == 2 ? la.seg_base(sg, p) : lowest(la, p)) + dla[p][sg] + va[sg][i] != la.row_ptr(row, sg, p) + la.col_off(c16)) {
            printf("DMA address mismatch A seg %d plane %d row %d c16 %d\n", sg, p, row, c16);
            va[sg][i] = kBufRange;
          }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int row = (wave * NB + i) * RPI + lane / C16;
          const int c16 = (lane % C16) ^ Cfg::swz(row);
          if (vb[sg][i] != kBufRange && (gbyte*)(kBufDma == 2 ? lb.seg_base(sg, p) : lowest(lb, p)) + dlb[p][sg] + vb[sg][i] != lb.row_ptr(row, sg, p) + lb.col_off(c16)) {
            printf("DMA address mismatch B seg %d plane %d row %d c16 %d\n", sg, p, row, c16);
            vb[sg][i] = kBufRange;
          }
        }
      }
    }
#endif
    const unsigned inca = (unsigned)__builtin_amdgcn_readfirstlane((int)la.tile_inc(ROWB));
    const unsigned incb = (unsigned)__builtin_amdgcn_readfirstlane((int)lb.tile_inc(ROWB));
    __amdgpu_buffer_rsrc_t ra[NP], rb[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      ra[p] = make_rsrc(kBufDma == 2 ? la.seg_base(0, p) : lowest(la, p));
      rb[p] = make_rsrc(kBufDma == 2 ? lb.seg_base(0, p) : lowest(lb, p));
    }
    unsigned cva[NA], cvb[NB];
    unsigned cda[NP], cdb[NP];  // the current segment's distance from the base, per plane (uniform)
    auto enter_seg = [&](int sg) {  // uniform sg
      if constexpr (kBufDma == 2) {
        if (sg > 0) {
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            ra[p] = make_rsrc(la.seg_base(sg, p));
            rb[p] = make_rsrc(lb.seg_base(sg, p));
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) cva[i] = sg == 0 ? va[0][i] : (sg == 1 ? va[1][i] : va[2][i]);
#pragma unroll
      for (int i = 0; i < NB; ++i) cvb[i] = sg == 0 ? vb[0][i] : (sg == 1 ? vb[1][i] : vb[2][i]);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        cda[p] = sg == 0 ? dla[p][0] : (sg == 1 ? dla[p][1] : dla[p][2]);
        cdb[p] = sg == 0 ? dlb[p][0] : (sg == 1 ? dlb[p][1] : dlb[p][2]);
      }
    };
    enter_seg(0);
    int seg = 0, left = nt0, seg_len = len0, kpos = 0, istage = 0;
    unsigned soa = 0, sob = 0;  // scalar byte offsets of the current tile inside the segment

    auto dma_a = [&](const __amdgpu_buffer_rsrc_t& r, char* dst, unsigned voff, unsigned soff) {
      // (size, immediate offset and cache-policy arguments of the builtin must be literals)
      if constexpr (Gate::kAuxA == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, (int)voff, (int)soff, 0, 16);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, (int)voff, (int)soff, 0, 0);
    };
    auto dma_b = [&](const __amdgpu_buffer_rsrc_t& r, char* dst, unsigned voff, unsigned soff) {
      if constexpr (Cfg::kAuxB == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, (int)voff, (int)soff, 0, 2);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, (int)voff, (int)soff, 0, 0);
    };
    auto issue_tile = [&]() {
      char* st = lds + istage * Cfg::kStageBytes;
      const bool partial = (kpos + KT > seg_len);  // uniform: last, zero-padded tile of a segment (or past the end)
      auto dst_a = [&](int p, int i) {
        char* dst = st + p * Cfg::kPlaneABytes + (wave * NA + i) * 1024;
        if (Cfg::kDummy && (wave * NA + i) * 1024 >= Cfg::kPlaneABytes) dst = lds + Cfg::kDummyOff + wave * 1024;
        return dst;
      };
      auto dst_b = [&](int p, int i) {
        char* dst = st + NP * Cfg::kPlaneABytes + p * Cfg::kPlaneBBytes + (wave * NB + i) * 1024;
        if (Cfg::kDummy && (wave * NB + i) * 1024 >= Cfg::kPlaneBBytes) dst = lds + Cfg::kDummyOff + wave * 1024;
        return dst;
      };
      bool slow = partial;
      if constexpr (LoaderA::kRange) {
#pragma unroll
        for (int i = 0; i < NA; ++i) slow = slow || kpos < alo_all[i] || kpos + KT > ahi_all[i];
      }
      if (slow) {
        // (its own block, kept apart from the common one by the asm statement: merged into per-lane selects, the test would put
        // three vector-ALU instructions in front of EVERY tile's DMAs)
        asm volatile("; zero-padded tile" ::: "memory");
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int i = 0; i < NA; ++i) {
            bool in = kpos + ca[i] < seg_len;
            if constexpr (LoaderA::kRange) in = in && kpos + ca[i] >= aklo[i] && kpos + ca[i] < akhi[i];
            dma_a(ra[p], dst_a(p, i), in ? cva[i] : kBufRange, soa + cda[p]);
          }
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int i = 0; i < NB; ++i) dma_b(rb[p], dst_b(p, i), kpos + cb[i] >= seg_len ? kBufRange : cvb[i], sob + cdb[p]);
      } else {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int i = 0; i < NA; ++i) dma_a(ra[p], dst_a(p, i), cva[i], soa + cda[p]);
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int i = 0; i < NB; ++i) dma_b(rb[p], dst_b(p, i), cvb[i], sob + cdb[p]);
      }
      istage = (istage + 1 == S) ? 0 : istage + 1;
      kpos += KT;
      soa += inca;
      sob += incb;
      if (--left == 0) {  // next segment (or, past the last one, out-of-range loads that keep vmcnt uniform)
        ++seg;
        kpos = 0;
        soa = sob = 0;
        if (seg == 1 && nseg > 1) { left = nt1; seg_len = len1; enter_seg(1); }
        else if (seg == 2 && nseg > 2) { left = nt2; seg_len = len2; enter_seg(2); }
        else { left = 0x7fffffff; seg_len = 0; }  // every further tile is "partial" with nothing valid
      }
    };
#pragma unroll
    for (int t = 0; t < S - 1; ++t) issue_tile();
    const int nk_run = live ? nk : 0;
    if (live && !Cfg::kBig) {
      wait_vmcnt<(S - 2) * Cfg::NLOADS>();  // tile 0 landed
      __builtin_amdgcn_s_barrier();         // B0
    }
    for (int t = 0; t < nk_run; ++t) {
      wait_vmcnt<Cfg::kWaitCnt>();
      __builtin_amdgcn_s_barrier();
      if (wave8 == 4) gate.mark(t);
      gate_at(t + S - 1);
      if (dbg != 3) issue_tile();
    }
    wait_vmcnt<0>();
    }
  } else if (is_loader) {
    // =========================== loader waves ===========================
    // loader w issues, per plane, instructions i = 0..NA-1 covering A-tile rows
    // (w*NA + i)*RPI + lane/C16; this lane's 16-byte column is (lane % C16) ^ swz(row).
    gbyte *qa0[NP][NA], *qa1[NP][NA], *qa2[NP][NA], *cura[NP][NA];
    gbyte *qb0[NP][NB], *qb1[NP][NB], *qb2[NP][NB], *curb[NP][NB];
    int ca[NA], cb[NB];      // element offset of the lane's column
    long inca[NA], incb[NB];  // byte increment per tile
    int aklo[NA], akhi[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = (wave * NA + i) * RPI + lane / C16;
      const bool ok = row < BM && la.row_ok(row) && dbg != 1;
      const int c16 = (lane % C16) ^ Cfg::swz(row);
      ca[i] = c16 * EPC;
      inca[i] = ok ? la.tile_inc(ROWB) : 0;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        qa0[p][i] = ok ? la.row_ptr(row, 0, p) + la.col_off(c16) : zero_addr();
        qa1[p][i] = ok ? la.row_ptr(row, 1, p) + la.col_off(c16) : zero_addr();
        qa2[p][i] = ok ? la.row_ptr(row, 2, p) + la.col_off(c16) : zero_addr();
        cura[p][i] = qa0[p][i];
      }
      if (LoaderA::kRange) {
        aklo[i] = ok ? la.k_lo(row) : 0;
        akhi[i] = ok ? la.k_hi(row) : 0;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int row = (wave * NB + i) * RPI + lane / C16;
      const bool ok = row < BN && lb.row_ok(row) && dbg != 1;
      const int c16 = (lane % C16) ^ Cfg::swz(row);
      cb[i] = c16 * EPC;
      incb[i] = ok ? lb.tile_inc(ROWB) : 0;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        qb0[p][i] = ok ? lb.row_ptr(row, 0, p) + lb.col_off(c16) : zero_addr();
        qb1[p][i] = ok ? lb.row_ptr(row, 1, p) + lb.col_off(c16) : zero_addr();
        qb2[p][i] = ok ? lb.row_ptr(row, 2, p) + lb.col_off(c16) : zero_addr();
        curb[p][i] = qb0[p][i];
      }
    }
    int seg = 0, left = nt0, seg_len = len0, kpos = 0, istage = 0;

    auto issue_tile = [&]() {
      char* st = lds + istage * Cfg::kStageBytes;
      const bool partial = (kpos + KT > seg_len);  // uniform: last, zero-padded tile of a segment (or past the end)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          gbyte* ptr = cura[p][i];
          if (LoaderA::kRange)  // the seg_len bound also covers the zero-block tiles past the last segment
            ptr = (kpos + ca[i] >= aklo[i] && kpos + ca[i] < akhi[i] && kpos + ca[i] < seg_len) ? ptr : zero_addr();
          else
            ptr = (partial && kpos + ca[i] >= seg_len) ? zero_addr() : ptr;
          char* dst = st + p * Cfg::kPlaneABytes + (wave * NA + i) * 1024;
          if (Cfg::kDummy && (wave * NA + i) * 1024 >= Cfg::kPlaneABytes) dst = lds + Cfg::kDummyOff + wave * 1024;
          // (size and cache-policy arguments of the builtin must be literals)
          if constexpr (Gate::kAuxA == 16) __builtin_amdgcn_global_load_lds((global_void*)ptr, (lds_void*)dst, 16, 0, 16);
          else __builtin_amdgcn_global_load_lds((global_void*)ptr, (lds_void*)dst, 16, 0, 0);
          cura[p][i] += inca[i];
        }
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          gbyte* ptr = (partial && kpos + cb[i] >= seg_len) ? zero_addr() : curb[p][i];
          char* dstc = st + NP * Cfg::kPlaneABytes + p * Cfg::kPlaneBBytes + (wave * NB + i) * 1024;
          if (Cfg::kDummy && (wave * NB + i) * 1024 >= Cfg::kPlaneBBytes) dstc = lds + Cfg::kDummyOff + wave * 1024;
          lds_void* dst = (lds_void*)dstc;
          // (size and cache-policy arguments of the builtin must be literals)
          if constexpr (Cfg::kAuxB == 2) __builtin_amdgcn_global_load_lds((global_void*)ptr, dst, 16, 0, 2);
          else __builtin_amdgcn_global_load_lds((global_void*)ptr, dst, 16, 0, 0);
          curb[p][i] += incb[i];
        }
      }
      istage = (istage + 1 == S) ? 0 : istage + 1;
      kpos += KT;
      if (--left == 0) {  // next segment (or, past the last one, zero-block loads that keep vmcnt uniform)
        ++seg;
        kpos = 0;
        if (seg == 1 && nseg > 1) {
          left = nt1; seg_len = len1;
#pragma unroll
          for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < NA; ++i) cura[p][i] = qa1[p][i];
#pragma unroll
            for (int i = 0; i < NB; ++i) curb[p][i] = qb1[p][i];
          }
        } else if (seg == 2 && nseg > 2) {
          left = nt2; seg_len = len2;
#pragma unroll
          for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < NA; ++i) cura[p][i] = qa2[p][i];
#pragma unroll
            for (int i = 0; i < NB; ++i) curb[p][i] = qb2[p][i];
          }
        } else {
          left = 0x7fffffff; seg_len = 0;  // every further tile is "partial" with nothing valid
        }
      }
    };

    // tiles 0 .. S-2 in flight
#pragma unroll
    for (int t = 0; t < S - 1; ++t) issue_tile();
    const int nk_run = live ? nk : 0;
    if (live && !Cfg::kBig) {
      wait_vmcnt<(S - 2) * Cfg::NLOADS>();  // tile 0 landed
      __builtin_amdgcn_s_barrier();         // B0
    }
    for (int t = 0; t < nk_run; ++t) {
      // small tiles: this wave's part of tile t+1 has landed (fragments are read one tile ahead);
      // big tiles: tile t has landed (fragments are read right after this barrier)
      wait_vmcnt<Cfg::kWaitCnt>();
      __builtin_amdgcn_s_barrier();       // the MFMA waves are done reading the stage tile t-1 occupied
      if (wave8 == 4) gate.mark(t);       // (measurement only: time stamps of the K loop's progress, TTSDEC_STAMPS)
      gate_at(t + S - 1);
      if (dbg != 3) issue_tile();         // tile t+S-1 into that stage (dbg 3: measurement ablation)
    }
    wait_vmcnt<0>();  // trailing zero-block loads must land before the ring is reused
  } else {
    // ============================ MFMA waves ============================
    const int arow = wm * 32 + l32, brow = wn * 32 + l32;
    int rstage = 0;
    if (wave >= Cfg::NMW) {
      // spare wave of a 2-MFMA-wave tile: only keeps the workgroup barriers balanced
      if (live) {
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nk; ++t) {
          __builtin_amdgcn_s_barrier();
          gate_at(t + S - 1);
        }
      }
    } else if constexpr (Cfg::kBig) {
      // TM x TN accumulators per wave, 16-bit planes; fragments single-buffered
      constexpr int TM = Cfg::kTM, TN = Cfg::kTN, NS = Cfg::NS16;
      f32x16 accb[TM][TN], accb2[TM][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) { accb[i][j][e] = 0.f; accb2[i][j][e] = 0.f; }
      int aoff[TM][NS], boff[TN][NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int c16 = s * 2 + half;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int r = (wm * TM + i) * 32 + l32;
          aoff[i][s] = r * ROWB + ((c16 ^ Cfg::swz(r)) << 4);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int r = (wn * TN + j) * 32 + l32;
          boff[j][s] = NP * Cfg::kPlaneABytes + r * ROWB + ((c16 ^ Cfg::swz(r)) << 4);
        }
      }
      const int nk_run = live ? nk : 0;
      for (int t = 0; t < nk_run; ++t) {
        __builtin_amdgcn_s_barrier();  // tile t is in LDS
        gate_at(t + S - 1);
        const char* st = lds + rstage * Cfg::kStageBytes;
        rstage = (rstage + 1 == S) ? 0 : rstage + 1;
        f16x8 ah[TM][NS], al[TM][NS], bh[TN][NS], bl[TN][NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            ah[i][s] = *reinterpret_cast<const f16x8*>(st + aoff[i][s]);
            if constexpr (NP == 2) al[i][s] = *reinterpret_cast<const f16x8*>(st + Cfg::kPlaneABytes + aoff[i][s]);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            bh[j][s] = *reinterpret_cast<const f16x8*>(st + boff[j][s]);
            if constexpr (NP == 2) bl[j][s] = *reinterpret_cast<const f16x8*>(st + Cfg::kPlaneBBytes + boff[j][s]);
          }
        }
        // k16 step 0's fragments first, then step 1's reads ride in the gaps of step 0's MFMAs (one per gap)
        {
          constexpr int kReads = (TM + TN) * NP, kMfma = TM * TN * (Cfg::kPrec == PREC_F16S ? 3 : 1);
          __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
          if constexpr (NS > 1) {
            static_for<(kReads < kMfma ? kReads : kMfma)>([&](auto) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            });
          }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if constexpr (Cfg::kPrec == PREC_F16S) {
                accb[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i][s], bh[j][s], accb[i][j], 0, 0, 0);
                accb2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i][s], bl[j][s], accb2[i][j], 0, 0, 0);
                accb2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i][s], bh[j][s], accb2[i][j], 0, 0, 0);
              } else {
                accb[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i][s]),
                                                                     __builtin_bit_cast(bf16x8, bh[j][s]), accb[i][j], 0, 0, 0);
              }
            }
      }
      __syncthreads();  // every wave is done with the ring: the out tile may overwrite it
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = accb[i][j][r];
            if constexpr (Cfg::kPrec == PREC_F16S) v = fmaf(accb2[i][j][r], 1.0f / kSplitScale, v);
            const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            smem[row * LDO + (wn * TN + j) * 32 + l32] = v;
          }
    } else if constexpr (Cfg::kPrec == PREC_F32) {
      constexpr int NQ = Cfg::NQ32;
      int aoff[NQ], boff[NQ];  // byte offsets inside a stage
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int c16 = wk * (2 * NQ) + half * NQ + q;
        aoff[q] = arow * ROWB + ((c16 ^ Cfg::swz(arow)) << 4);
        boff[q] = Cfg::kPlaneABytes + brow * ROWB + ((c16 ^ Cfg::swz(brow)) << 4);
      }
      // Fragment registers are double-buffered: the reads of tile t+1 are issued right after
      // the barrier that makes it visible and complete underneath tile t's MFMA chain.
      f32x4 fa[2][NQ], fb[2][NQ];
      auto read_frags = [&](auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
        const char* st = lds + rstage * Cfg::kStageBytes;
        rstage = (rstage + 1 == S) ? 0 : rstage + 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          fa[buf][q] = *reinterpret_cast<const f32x4*>(st + aoff[q]);
          fb[buf][q] = *reinterpret_cast<const f32x4*>(st + boff[q]);
        }
      };
      auto tile_step = [&](auto cur_c, int t) {
        constexpr int cur = decltype(cur_c)::value;
        __builtin_amdgcn_s_barrier();        // B(t+1): tile t+1 is in LDS
        gate_at(t + S - 1);
        // tile t's fragments were requested a whole tile ago: retire them here (no stall), in a
        // form the compiler's wait-count model sees, so it does not later drain the next reads
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only
        read_frags(std::integral_constant<int, cur ^ 1>{});
        // The next tile's fragment reads ride in the gaps of this tile's MFMA chain, one per two MFMAs (round 4).  In
        // front of the chain they delayed its first MFMA by their issue time on every tile - the four MFMA waves leave the
        // barrier together and 32 ds_read_b128 queue up at the LDS - which cost the fp32 K loop ~50-300 cycles per 1024-cycle
        // tile (tools/ubench_f32_partner.hip: 1147 -> 1097 per tile beside an idle partner, more with a busy one; in the
        // library, together with the loaders' scalar bookkeeping: 1725 -> 1270 cycles per tile of the decoder LSTM).
        // (No runtime branch between the reads and the MFMAs - they must sit in ONE basic block to be interleaved - so the
        // "no MFMAs" measurement ablation, dbg 4, no longer exists for this arithmetic mode.)
#pragma unroll
        for (int i = 0; i < 2 * NQ; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][q][e], fb[cur][q][e], acc, 0, 0, 0);
        }
      };
      const int nk_run = live ? nk : 0;
      if (live) {
        __builtin_amdgcn_s_barrier();  // B0: tile 0 is in LDS
        read_frags(std::integral_constant<int, 0>{});
      }
      for (int t = 0; t < nk_run; t += 2) {
        tile_step(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nk_run) tile_step(std::integral_constant<int, 1>{}, t + 1);
      }
    } else {
      // 16-bit planes: per k16-step s one 16-byte fragment of A_hi (, A_lo), B_hi (, B_lo)
      constexpr int NS = Cfg::NS16;  // k16 steps per tile: 4, or 2 for half-depth stages
      int aoff[NS], boff[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int c16 = wk * 8 + s * 2 + half;
        aoff[s] = arow * ROWB + ((c16 ^ Cfg::swz(arow)) << 4);
        boff[s] = NP * Cfg::kPlaneABytes + brow * ROWB + ((c16 ^ Cfg::swz(brow)) << 4);
      }
      f16x8 ah[2][NS], al[2][NS], bh[2][NS], bl[2][NS];  // (bf16 data travels in the same 16-byte registers)
      auto read_frags = [&](auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
        const char* st = lds + rstage * Cfg::kStageBytes;
        rstage = (rstage + 1 == S) ? 0 : rstage + 1;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          ah[buf][s] = *reinterpret_cast<const f16x8*>(st + aoff[s]);
          bh[buf][s] = *reinterpret_cast<const f16x8*>(st + boff[s]);
          if constexpr (NP == 2) {
            al[buf][s] = *reinterpret_cast<const f16x8*>(st + Cfg::kPlaneABytes + aoff[s]);
            bl[buf][s] = *reinterpret_cast<const f16x8*>(st + Cfg::kPlaneBBytes + boff[s]);
          }
        }
      };
      auto tile_step = [&](auto cur_c, int t) {
        constexpr int cur = decltype(cur_c)::value;
        __builtin_amdgcn_s_barrier();
        gate_at(t + S - 1);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only (see the fp32 path)
        read_frags(std::integral_constant<int, cur ^ 1>{});
        // Interleave the next tile's fragment reads with this tile's MFMAs (two ds_read_b128 per MFMA gap are
        // nearly free, MI355X_MICROARCH.md LDS section).  With all reads pinned in front of the chain the
        // MFMA waves spent 16 read-issue slots + 12 MFMAs in series per tile: 0.31 us per tile with no DMA at
        // all (round 1 ablation), more than the operand stream itself needs.
        if constexpr (Cfg::kPrec == PREC_F16S) {
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
        } else {
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          }
        }
        {  // (no runtime branch here: reads and MFMAs must sit in ONE basic block to be interleaved)
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            if constexpr (Cfg::kPrec == PREC_F16S) {
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][s], bh[cur][s], acc, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][s], bl[cur][s], acc2, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][s], bh[cur][s], acc2, 0, 0, 0);
            } else {
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[cur][s]),
                                                            __builtin_bit_cast(bf16x8, bh[cur][s]), acc, 0, 0, 0);
            }
          }
        }
      };
      const int nk_run = live ? nk : 0;
      if (live) {
        __builtin_amdgcn_s_barrier();
        read_frags(std::integral_constant<int, 0>{});
      }
      for (int t = 0; t < nk_run; t += 2) {
        tile_step(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nk_run) tile_step(std::integral_constant<int, 1>{}, t + 1);
      }
      if constexpr (Cfg::kPrec == PREC_F16S) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaf(acc2[i], 1.0f / kSplitScale, acc[i]);
      }
    }
  }
  if (!(Cfg::kBig && !is_loader)) __syncthreads();  // (big-tile MFMA waves crossed this barrier before their stores)

  // accumulators -> LDS out tile (aliases the ring).
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
  if (!Cfg::kBig && !is_loader && wave < Cfg::NMW) {
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
