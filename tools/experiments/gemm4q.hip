// RETIRED EXPERIMENT (round 4): no longer part of libezclip_hip.so.  Round 3 staged this four-wave 128 x 128-wave-tile variant of the
// NT GEMM to test whether fewer LDS reads per MFMA lower the energy per flop under the socket power cap: they do not (its random /
// zero-operand ratio equals the 8-phase kernel's).  Round 4 found where the energy goes -- the MFMA shape
// (v_mfma_f32_16x16x32_bf16 vs 32x32x16: tools/mfma_power_probe.hip) -- and moved the product kernels to it; this file is kept as the
// record of the experiment (it builds against csrc/ headers of commit 2adffaf).
// EXPERIMENT (round 3, late; staged for round 4 -- NOT dispatched by gemm_nt unless ezclip_debug_set(0, 4) selects it).
//
// bf16 MFMA GEMM, 256 x 256 x 64 tile, FOUR waves of 128 x 128 (gfx950 / CDNA4 only):
//
//   C[M,N] = epilogue( A[M,K] . B[N,K]^T )        A, B, C bf16; fp32 accumulate        (bias / activation epilogues only)
//
// Why: under the 1 400 W socket cap the 8-phase kernel of gemm8p.hip is not limited by its schedule but by its energy per flop
// (with all-zero operands it leads hipBLASLt by 9-12 % on the ViT shapes at 2.39 GHz, with random operands
// the vendor kernel leads).  A 128 x 64 wave tile reads 24 operand fragments from LDS per 32 MFMAs (0.75 ds_read_b128 per
// v_mfma_f32_32x32x16_bf16) and crosses two barriers per eight MFMAs; a 128 x 128 wave tile reads 32 per 64 (0.5) and crosses
// one barrier per sixteen.  The price: 256 accumulator registers per lane, i.e. one wave per SIMD and nobody to cover its LDS
// latency -- the reads of phase p + 1 are therefore issued BEFORE the sixteen MFMAs of phase p (software pipeline inside the
// wave; 160 fragment registers: A-lo, A-hi, B-hi and two generations of B-lo).
//
// Everything else is the 8-phase design: half-tile ring of 8 x 16 KiB (half-tile n = 4 * tile + {A-lo, B-lo, B-hi, A-hi} in slot
// n & 7), hand-issued LDS-DMA with counted vmcnt, swizzled lane-linear images, persistent workgroups in XCD-aware tile order, the
// next tile's first six half-tiles DMA'd under the epilogue.  Phase k (k = 4 * tile + P):
//     reads (for phase k + 1): half-tile k + 2        P0 -> B-hi(t)   P1 -> A-hi(t)   P2 -> A-lo(t+1)   P3 -> B-lo(t+1)
//     DMA: half-tile k + 6 (four 1 KiB pieces per wave)
//     16 MFMAs:  P0 A-lo x B-lo   P1 A-lo x B-hi   P2 A-hi x B-hi   P3 A-hi x B-lo
//     s_waitcnt vmcnt(12): half-tile k + 3 has landed (k + 4 .. k + 6 stay in flight);  s_barrier
// The epilogue is gemm_pipe.h's epilogue_rows run on the two 64-column strips of the wave tile one after the other (the second
// strip's bias wait drains the queue: a known cost of this first version; a 128-column epilogue is the follow-up).
#include "ezclip_common.h"
#include "kernels.h"
#include "gemm_pipe.h"

namespace ezclip {

namespace {

constexpr int kThreadsQ = 256;
constexpr int kStageQ = 6 * kSlot;     // epilogue staging: ring slots 6, 7 (4 waves x 8 KiB)
constexpr int kLdsQ = 8 * kSlot;       // 128 KiB

struct CtxQ {
  const char* smem;
  uint32_t lds_base;
  i32x4_t srdA, srdB;
  uint32_t voffA[4], voffB[4];
  uint32_t hiA, hiB;          // byte offset of the "hi" rows (64 * lda, 64 * ldb)
  uint32_t dma_dst;           // wave * 4096 (plus slot base, plus i * 1024)
  uint32_t rdA[4], rdB[4];    // per-lane LDS byte offsets of the 4 k-step chunks (swizzled), row included; second block + 4096
};

struct FragsQ {
  uint4 al[2][4], ah[2][4];   // A-lo / A-hi of the current K-tile: [32-row block][k-step]
  uint4 bl[2][2][4];          // B-lo, two generations (K-tile parity): [generation][32-column block][k-step]
  uint4 bh[2][4];             // B-hi
};

template <int P>
__device__ __forceinline__ void q_read(const CtxQ& c, uint4 (&dst)[2][4], int slot_byte) {
  const uint32_t* rd = (P == 1 || P == 2) ? c.rdA : c.rdB;      // P0: B-hi, P1: A-hi, P2: A-lo, P3: B-lo
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    dst[0][s] = *reinterpret_cast<const uint4*>(c.smem + slot_byte + rd[s]);
    dst[1][s] = *reinterpret_cast<const uint4*>(c.smem + slot_byte + 4096 + rd[s]);
  }
}

// One phase.  P: quadrant; PAR: K-tile parity (static slot bases, B-lo generation); READ: issue the reads of the next phase;
// ISSUE: issue half-tile k + 6; VM: vmcnt to wait for afterwards (-1: none); relaxed: first K-tile of a tile (VMR instead of VM).
template <int P, int PAR, bool READ, bool ISSUE, int VM, int VMR = VM>
__device__ __forceinline__ void q_phase(const CtxQ& c, FragsQ& f, f32x16_t (&accL)[4][2], f32x16_t (&accR)[4][2],
                                        uint32_t kbyte_next1, uint32_t kbyte_next2, bool relaxed = false) {
  constexpr int k8 = 4 * PAR + P;
  // Issue order inside a phase (EZ_Q_INTERLEAVE, default 2):
  //   0: DMA x4, reads x8, then the 16 MFMAs            (first version: 992 TF on vit.qkv, the 8-phase kernel 1 100)
  //   1: DMA x4, then reads and MFMAs interleaved by the scheduler (one ds_read_b128 after every second MFMA: 1 022 TF)
  //   2: everything hand-placed: per k-step  MFMA, read, MFMA, DMA piece, MFMA, read, MFMA -- with one wave per SIMD every
  //      instruction issued ahead of the first MFMA is matrix-pipe idle time, and one LDS-DMA issue (M0 save / set / restore)
  //      fits under one 32-cycle MFMA
#ifndef EZ_Q_INTERLEAVE
#define EZ_Q_INTERLEAVE 2
#endif
  constexpr int rslot = ((k8 + 2) & 7) * kSlot;
  constexpr int dslot = ((k8 + 6) & 7) * kSlot;
  const uint32_t dst = c.lds_base + dslot + c.dma_dst;
  auto dma_piece = [&](int i) {
    if constexpr (ISSUE) {
      if constexpr (P == 0) dma16(dst + i * 1024, c.voffB[i], c.srdB, kbyte_next1 + c.hiB);
      else if constexpr (P == 1) dma16(dst + i * 1024, c.voffA[i], c.srdA, kbyte_next1 + c.hiA);
      else if constexpr (P == 2) dma16(dst + i * 1024, c.voffA[i], c.srdA, kbyte_next2);
      else dma16(dst + i * 1024, c.voffB[i], c.srdB, kbyte_next2);
    }
  };
  uint4 (&rdst)[2][4] = (P == 0) ? f.bh : (P == 1) ? f.ah : (P == 2) ? f.al : f.bl[PAR ^ 1];
  const uint32_t* rd = (P == 1 || P == 2) ? c.rdA : c.rdB;
  auto read_piece = [&](int blk, int s) {
    if constexpr (READ) rdst[blk][s] = *reinterpret_cast<const uint4*>(c.smem + rslot + blk * 4096 + rd[s]);
  };
  uint4 (&a)[2][4] = (P < 2) ? f.al : f.ah;
  uint4 (&b)[2][4] = (P == 0 || P == 3) ? f.bl[PAR] : f.bh;
  f32x16_t (&acc)[4][2] = (P == 0 || P == 3) ? accL : accR;
  constexpr int i0 = (P >= 2) ? 2 : 0;
#if EZ_Q_INTERLEAVE == 2
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    mma32(acc[i0][0], b[0][s], a[0][s], bf16_t());
    __builtin_amdgcn_sched_barrier(0);
    read_piece(0, s);
    __builtin_amdgcn_sched_barrier(0);
    mma32(acc[i0 + 1][0], b[0][s], a[1][s], bf16_t());
    __builtin_amdgcn_sched_barrier(0);
    dma_piece(s);
    __builtin_amdgcn_sched_barrier(0);
    mma32(acc[i0][1], b[1][s], a[0][s], bf16_t());
    __builtin_amdgcn_sched_barrier(0);
    read_piece(1, s);
    __builtin_amdgcn_sched_barrier(0);
    mma32(acc[i0 + 1][1], b[1][s], a[1][s], bf16_t());
    __builtin_amdgcn_sched_barrier(0);
  }
#else
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_piece(i);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < 4; ++s) { read_piece(0, s); read_piece(1, s); }
#if EZ_Q_INTERLEAVE == 0
  __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    mma32(acc[i0][0], b[0][s], a[0][s], bf16_t());
    mma32(acc[i0 + 1][0], b[0][s], a[1][s], bf16_t());
    mma32(acc[i0][1], b[1][s], a[0][s], bf16_t());
    mma32(acc[i0 + 1][1], b[1][s], a[1][s], bf16_t());
  }
#if EZ_Q_INTERLEAVE == 1
  if constexpr (READ) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);     // two MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one LDS read
    }
  }
#endif
  __builtin_amdgcn_sched_barrier(0);
#endif
  if constexpr (VMR != VM) {
    if (relaxed) wait_vm<VMR>(); else wait_vm<VM>();
  } else {
    wait_vm<VM>();
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// One K-tile = 4 phases.  TAIL: 0 = steady state; 1 = second-to-last K-tile (P0, P1 issue; waits 12, 12, 8, 4);
// 2 = last K-tile (no issue, no reads for a next K-tile; waits 0, -, -, -).
template <int PAR, int TAIL, int VMR = 12>
__device__ __forceinline__ void q_ktile(const CtxQ& c, FragsQ& f, f32x16_t (&accL)[4][2], f32x16_t (&accR)[4][2], uint32_t kb1,
                                        uint32_t kb2, bool relaxed = false) {
  if constexpr (TAIL == 0) {
    q_phase<0, PAR, true, true, 12, VMR>(c, f, accL, accR, kb1, kb2, relaxed);
    q_phase<1, PAR, true, true, 12, VMR>(c, f, accL, accR, kb1, kb2, relaxed);
    q_phase<2, PAR, true, true, 12, VMR>(c, f, accL, accR, kb1, kb2, relaxed);
    // (phase 3 waits for half-tile k + 3 = 6, which is issued AFTER the previous tile's stores: no relaxed count there)
    q_phase<3, PAR, true, true, 12>(c, f, accL, accR, kb1, kb2);
  } else if constexpr (TAIL == 1) {
    q_phase<0, PAR, true, true, 12>(c, f, accL, accR, kb1, kb2);
    q_phase<1, PAR, true, true, 12>(c, f, accL, accR, kb1, kb2);
    q_phase<2, PAR, true, false, 8>(c, f, accL, accR, kb1, kb2);
    q_phase<3, PAR, true, false, 4>(c, f, accL, accR, kb1, kb2);
  } else {
    q_phase<0, PAR, true, false, 0>(c, f, accL, accR, kb1, kb2);
    q_phase<1, PAR, true, false, -1>(c, f, accL, accR, kb1, kb2);
    q_phase<2, PAR, false, false, -1>(c, f, accL, accR, kb1, kb2);
    q_phase<3, PAR, false, false, -1>(c, f, accL, accR, kb1, kb2);
  }
}

template <bool HAS_C2>
__global__ __launch_bounds__(kThreadsQ) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_nt_4q_kernel(GemmArgs p, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int tiles_n = p.N >> 8;

  CtxQ c;
  c.smem = smem;
  c.lds_base = (uint32_t)(size_t)smem;
  const uint32_t lda_b = (uint32_t)p.lda * 2u, ldb_b = (uint32_t)p.ldb * 2u;
  c.srdA = make_srd(p.A, (uint32_t)(p.M - 1) * lda_b + (uint32_t)p.K * 2u);
  c.srdB = make_srd(p.B, (uint32_t)(p.N - 1) * ldb_b + (uint32_t)p.K * 2u);
  c.hiA = 64u * lda_b;
  c.hiB = 64u * ldb_b;
  c.dma_dst = wave * 4096;
  {
    const int sw = (l31 >> 1) & 7;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint32_t ch = (uint32_t)((2 * s + h) ^ sw) << 4;
      c.rdA[s] = (uint32_t)(wm * 64 + l31) * 128 + ch;
      c.rdB[s] = (uint32_t)(wn * 64 + l31) * 128 + ch;
    }
  }
  auto tile_origin = [&](int v, int& m0, int& n0) {
    const int t = xcd_remap(v, ntiles);
    const int tm = t / tiles_n;
    m0 = tm << 8;
    n0 = (t - tm * tiles_n) << 8;
  };
  auto set_tile = [&](int m0, int n0) {
    const int ln = lane_id_now();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lr = (wave * 4 + i) * 8 + (ln >> 3);             // row of the half-tile image
      const uint32_t chk = (uint32_t)((ln & 7) ^ ((lr >> 1) & 7)) << 4;
      c.voffA[i] = (uint32_t)(m0 + (lr >> 6) * 128 + (lr & 63)) * lda_b + chk;
      c.voffB[i] = (uint32_t)(n0 + (lr >> 6) * 128 + (lr & 63)) * ldb_b + chk;
    }
  };
  auto issue_prologue = [&]() {      // half-tiles 0..5 of the tile described by c.voff*: A-lo B-lo B-hi A-hi (k 0), A-lo B-lo (k 64)
    const uint32_t d = c.lds_base + c.dma_dst;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(d + 0 * kSlot + i * 1024, c.voffA[i], c.srdA, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(d + 1 * kSlot + i * 1024, c.voffB[i], c.srdB, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(d + 2 * kSlot + i * 1024, c.voffB[i], c.srdB, c.hiB);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(d + 3 * kSlot + i * 1024, c.voffA[i], c.srdA, c.hiA);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(d + 4 * kSlot + i * 1024, c.voffA[i], c.srdA, 128);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(d + 5 * kSlot + i * 1024, c.voffB[i], c.srdB, 128);
  };

  const EpiCtx ep = make_epi_ctx<false, false, HAS_C2, false, false>(p);
  constexpr int NS = kStoresPerBlock * (1 + (HAS_C2 ? 1 : 0));       // stores per 32-row block of a 64-column strip
  constexpr int NST = (8 * NS > 51) ? 51 : 8 * NS;                    // stores of one tile that may still fly (vmcnt is 6 bits)
  int v = blockIdx.x, m0, n0;
  tile_origin(v, m0, n0);
  set_tile(m0, n0);
  issue_prologue();
  wait_vm<12>();                        // half-tiles 0, 1, 2 (this wave's pieces)
  bool first = true;

  f32x16_t accL[4][2], accR[4][2];      // columns 0..63 / 64..127 of the wave tile (re-zeroed block by block in the epilogue)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accL[i][j][r] = 0.f; accR[i][j][r] = 0.f; }

  for (;;) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();       // half-tiles 0..2 are complete; every wave is out of the previous epilogue
    __builtin_amdgcn_sched_barrier(0);

    FragsQ f;
    q_read<2>(c, f.al, 0 * kSlot);      // A-lo(0), B-lo(0): the only reads that are not one phase ahead
    q_read<3>(c, f.bl[0], 1 * kSlot);
    const int nk = p.K >> 6;            // even, >= 4 (checked by the launcher)
    uint32_t kb = 0;
    q_ktile<0, 0, 12 + NST>(c, f, accL, accR, kb + 128, kb + 256, !first);
    q_ktile<1, 0>(c, f, accL, accR, kb + 256, kb + 384);
    kb += 256;
    for (int kt = 2; kt < nk - 2; kt += 2) {
      q_ktile<0, 0>(c, f, accL, accR, kb + 128, kb + 256);
      q_ktile<1, 0>(c, f, accL, accR, kb + 256, kb + 384);
      kb += 256;
    }
    q_ktile<0, 1>(c, f, accL, accR, kb + 128, kb + 256);
    q_ktile<1, 2>(c, f, accL, accR, 0, 0);
    // every wave is past its last LDS read and no DMA is in flight: the ring is free

    const int vn = v + (int)gridDim.x;
    const bool has_next = vn < ntiles;
    int m0n = 0, n0n = 0;
    if (has_next) tile_origin(vn, m0n, n0n);
    EpiLoads eld;
    char* W = smem + kStageQ + wave * 8192;
    // right strip first: its epilogue issues the next tile's 24 DMAs; the left strip's bias wait then drains the queue
    epilogue_rows<true, false, false, HAS_C2, 24, false, false>(ep, accR, m0 + wm * 128, n0 + wn * 128 + 64, W, p.act, eld, [&]() {
      if (has_next) set_tile(m0n, n0n);
      issue_prologue();
    });
    epilogue_rows<true, false, false, HAS_C2, 0, false, false>(ep, accL, m0 + wm * 128, n0 + wn * 128, W, p.act, eld, [&]() {});
    if (!has_next) break;
    wait_vm<12 + NST>();     // half-tiles 0..2 of the next tile have landed; 3..5 and this tile's stores may still fly
    first = false;
    v = vn; m0 = m0n; n0 = n0n;
  }
}

}  // namespace

bool gemm_nt_4q_eligible(const GemmArgs& p, int dtype) {
  if (!gemm_nt_8p_eligible(p, dtype)) return false;
  return !p.R && !p.U && !p.ln_stats && !p.rowstat_part && !p.colsum &&
         (p.act == ACT_NONE || p.act == ACT_QUICKGELU || p.act == ACT_GELU_ERF || p.act == ACT_RELU);
}

int gemm_nt_4q(const GemmArgs& p_in, hipStream_t stream) {
  GemmArgs p = p_in;
  static int num_cus = 0;
  if (num_cus == 0) {
    int dev = 0;
    EZ_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    EZ_HIP(hipGetDeviceProperties(&prop, dev));
    num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int tiles = ((p.M + 255) >> 8) * (p.N >> 8);
  int grid = tiles;
  if (tiles > num_cus) grid = num_cus >= 8 ? (num_cus & ~7) : num_cus;
  static LdsOptIn lds_opt[2];
  ProfScope ps(PROF_GEMM, 2.0 * p.M * (double)p.N * p.K, stream);
  if (p.C2) {
    auto* kern = &gemm_nt_4q_kernel<true>;
    EZ_ENSURE_LDS(kern, lds_opt[1], kLdsQ);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreadsQ), kLdsQ, stream, p, tiles);
  } else {
    auto* kern = &gemm_nt_4q_kernel<false>;
    EZ_ENSURE_LDS(kern, lds_opt[0], kLdsQ);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreadsQ), kLdsQ, stream, p, tiles);
  }
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace ezclip
