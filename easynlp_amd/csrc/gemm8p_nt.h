// bf16 MFMA GEMM, 256 x 256 x 64 tile, 8-phase software pipeline (gfx950 / CDNA4 only): the NT kernel template.
// Included by gemm8p.hip (dispatcher, eligibility, weight-gradient kernel) and by gemm8p_nt_{a,b,c}.hip, which hold the explicit
// instantiations (one translation unit would compile 16 variants of a 250-register kernel back to back).
//
//   C[M,N] = epilogue( A[M,K] . B[N,K]^T )        A, B, C bf16; fp32 accumulate
//
// This is the kernel that carries ~94 % of the path's flops at the bench sizes: the QKV / out-proj /
// FFN products of every ViT and BERT block (reference: nn.MultiheadAttention in_proj/out_proj, mlp.c_fc /
// c_proj -- modeling_chineseclip.py:188-205; BertSelfAttention/BertSelfOutput/BertIntermediate/BertOutput
// -- modeling_bert.py:145-147,260,324,338) and, in the backward pass, every input-gradient product.
// gemm.hip keeps the 128x128 kernel for f32, ragged N, tiny problems and f32 outputs.
//
// Structure (all sizes in bytes are per workgroup = per CU, 512 threads = 8 waves, 1 workgroup / CU):
//   * waves 2 (M) x 4 (N); wave tile 128 x 64 = 4 x 2 MFMA 32x32 accumulators (128 fp32 registers);
//     v_mfma_f32_32x32x16_bf16 issued with swapped operands, so a lane owns one M row and runs of 4
//     consecutive N.
//   * A K-tile (64 k = 128 B per row) is split into four 16 KiB HALF-TILES of 128 rows each:
//       A-lo / A-hi : rows the two wave rows read for their first / second 64 M rows,
//       B-lo / B-hi : rows the four wave columns read for their first / second 32 N rows.
//     The four quadrants of a wave tile are the four PHASES of a K-tile, 8 MFMAs each:
//       P0 (A-lo,B-lo)  P1 (A-lo,B-hi)  P2 (A-hi,B-hi)  P3 (A-hi,B-lo)
//     so LDS is read 12 / 4 / 8 / 0 fragments per phase and every half-tile's buffer is dead again
//     after at most one phase.
//   * LDS is a ring of 8 half-tile slots (128 KiB).  Half-tile number n = 4*tile + {A-lo,B-lo,B-hi,A-hi}
//     lives in slot n & 7.  Phase k (k = 4*tile + P) reads its fragments, issues the LDS-DMA for
//     half-tile k+6 and waits (counted vmcnt, never 0 in steady state) until half-tile k+2 has landed:
//     four half-tiles = 8 loads per wave stay in flight across the barriers, ~4 phases (~1 us) of
//     prefetch distance.  The DMA is hand-issued (buffer_load_dwordx4 ... lds in inline asm, bounds-checked
//     by the buffer descriptor so ragged M needs no clamping): hipcc's own LDS-DMA builtin makes it wait
//     vmcnt(0) before every later ds_read, which serialises load and compute.
//   * The LDS image of a half-tile is lane-linear (that is what LDS-DMA writes): 128 B per row, with the
//     16-byte chunk index XORed by (row >> 1) & 7 on the SOURCE address and again on the ds_read_b128
//     (conflict-free for the 32-row fragment groups).
//   * The two wave rows run one barrier apart (wave row 1 executes one extra s_barrier up front, wave
//     row 0 one at the end): while one wave of a SIMD is in its MFMA segment the other is in its
//     read/DMA segment, so the matrix pipe always has a feeder.
//   * Epilogue: the fp32 accumulators go through a wave-private 8 KiB LDS image (outside ring slots 0..5),
//     one 32-row block at a time, so that the epilogue math runs on row-coalesced registers and every global
//     access of C / residual / pre-activation is 16 B per lane = 8 full 128-byte rows per wave instruction.
//   * The kernel is PERSISTENT (one workgroup per CU walks its tiles): the first six half-tiles of the next
//     tile are DMA'd while the current epilogue runs, bias / residual loads are issued one block ahead, and
//     there is no workgroup launch, kernarg load or cold pipeline between tiles.
#pragma once
#include <type_traits>

#include "ezclip_common.h"
#include "kernels.h"
#include "gemm_pipe.h"

namespace ezclip {
namespace {

struct Ctx {
  const char* smem;
  uint32_t lds_base;          // LDS byte address of smem
  i32x4_t srdA, srdB;
  uint32_t voffA[2], voffB[2];
  uint32_t hiA, hiB;          // byte offset of the "hi" rows (64*lda, 32*ldb)
  uint32_t dma_dst;           // wave*2048 (plus slot base, plus i*1024)
  uint32_t rdA[4], rdB[4];    // per-lane LDS byte offsets of the 4 k-step chunks (swizzled), row included
};

// One phase.  P: quadrant; PAR: tile parity (static slot bases); ISSUE: issue half-tile k+6;
// VM: vmcnt to wait for afterwards (-1: none).
// FR / ACC: Frags + f32x16_t[4][2] (v_mfma_f32_32x32x16_bf16: 8 per phase) or Frags16 + Acc16 (v_mfma_f32_16x16x32_bf16: 16 per
// phase, EZ_MI16) -- same bytes out of LDS either way (8 / 4 ds_read_b128 per A / B half), same 256 matrix-pipe cycles.
// FIRST (MI16 only): first K-tile of an output tile -- the first k-step's MFMAs take a literal zero as their C operand, so the
// accumulators are never zeroed by VALU moves (128 v_mov per wave and tile in the 32x32x16 build) and carry no value from one
// tile of the persistent loop into the next (each tile defines them afresh: no loop-carried register tuples to keep coalesced).
template <int P, int PAR, bool ISSUE, int VM, int VMR = VM, int XLD = 0, bool FIRST = false, typename FR, typename ACC>
__device__ __forceinline__ void phase(const Ctx& c, FR& f, ACC& acc, uint32_t kbyte_next1,
                                      uint32_t kbyte_next2, bool relaxed = false) {
  constexpr int k8 = 4 * PAR + P;                 // phase number mod 8
  constexpr bool MI16 = std::is_same<FR, Frags16>::value;
  // ---- read segment -----------------------------------------------------------------------
#ifdef EZ_ABL_NOLDSREAD     // timing ablation (results wrong by construction): fragments are read in the first K-tile of a tile only
  constexpr bool kRead = FIRST;
#else
  constexpr bool kRead = true;
#endif
  if constexpr (!kRead) {
  } else if constexpr (MI16) {
    if constexpr (P == 0) {
      constexpr int sB = ((k8 + 1) & 7) * kSlot, sA = (k8 & 7) * kSlot;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int v = 0; v < 2; ++v) f.bl[v][s] = *reinterpret_cast<const uint4*>(c.smem + sB + v * 2048 + c.rdB[s]);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u) f.a[u][s] = *reinterpret_cast<const uint4*>(c.smem + sA + u * 2048 + c.rdA[s]);
    } else if constexpr (P == 1) {
      constexpr int sB = ((k8 + 1) & 7) * kSlot;    // B-hi is half-tile 4t+2
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int v = 0; v < 2; ++v) f.bh[v][s] = *reinterpret_cast<const uint4*>(c.smem + sB + v * 2048 + c.rdB[s]);
    } else if constexpr (P == 2) {
      constexpr int sA = ((k8 + 1) & 7) * kSlot;    // A-hi is half-tile 4t+3
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u) f.a[u][s] = *reinterpret_cast<const uint4*>(c.smem + sA + u * 2048 + c.rdA[s]);
    }
  } else {
  if constexpr (P == 0) {
    constexpr int sB = ((k8 + 1) & 7) * kSlot, sA = (k8 & 7) * kSlot;
#pragma unroll
    for (int s = 0; s < 4; ++s) f.bl[s] = *reinterpret_cast<const uint4*>(c.smem + sB + c.rdB[s]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f.a[0][s] = *reinterpret_cast<const uint4*>(c.smem + sA + c.rdA[s]);
      f.a[1][s] = *reinterpret_cast<const uint4*>(c.smem + sA + 4096 + c.rdA[s]);
    }
  } else if constexpr (P == 1) {
    constexpr int sB = ((k8 + 1) & 7) * kSlot;    // B-hi is half-tile 4t+2
#pragma unroll
    for (int s = 0; s < 4; ++s) f.bh[s] = *reinterpret_cast<const uint4*>(c.smem + sB + c.rdB[s]);
  } else if constexpr (P == 2) {
    constexpr int sA = ((k8 + 1) & 7) * kSlot;    // A-hi is half-tile 4t+3
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f.a[0][s] = *reinterpret_cast<const uint4*>(c.smem + sA + c.rdA[s]);
      f.a[1][s] = *reinterpret_cast<const uint4*>(c.smem + sA + 4096 + c.rdA[s]);
    }
  }
  }
  // ---- DMA for half-tile k+6 (kind (P+2)&3: P0 -> B-hi(t+1), P1 -> A-hi(t+1), P2 -> A-lo(t+2), P3 -> B-lo(t+2))
  if constexpr (ISSUE) {
    constexpr int slot = ((k8 + 6) & 7) * kSlot;
    const uint32_t dst = c.lds_base + slot + c.dma_dst;
    if constexpr (P == 0) {
      dma16<1>(dst, c.voffB[0], c.srdB, kbyte_next1 + c.hiB);
      dma16<1>(dst + 1024, c.voffB[1], c.srdB, kbyte_next1 + c.hiB);
    } else if constexpr (P == 1) {
      dma16(dst, c.voffA[0], c.srdA, kbyte_next1 + c.hiA);
      dma16(dst + 1024, c.voffA[1], c.srdA, kbyte_next1 + c.hiA);
    } else if constexpr (P == 2) {
      dma16(dst, c.voffA[0], c.srdA, kbyte_next2);
      dma16(dst + 1024, c.voffA[1], c.srdA, kbyte_next2);
    } else {
      dma16<1>(dst, c.voffB[0], c.srdB, kbyte_next2);
      dma16<1>(dst + 1024, c.voffB[1], c.srdB, kbyte_next2);
    }
  }
  if constexpr (VMR != VM) {   // first K-tile of a tile: the previous tile's stores may still be queued (see ktile)
    if (relaxed) wait_role<VM, VMR - VM, 0, true>(); else wait_role<VM, 0, 0, true>();
  } else if constexpr (VM >= 0) {
    wait_role<VM - XLD, 0, XLD, true>();
  }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  // ---- MFMA segment -------------------------------------------------------------------------
  __builtin_amdgcn_s_setprio(1);
  constexpr int j = (P == 1 || P == 2) ? 1 : 0;
  if constexpr (MI16) {
    constexpr int rb0 = (P >= 2) ? 4 : 0, cb0 = 2 * j;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const uint4& b = (j == 0) ? f.bl[v][s] : f.bh[v][s];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (FIRST && s == 0) {
            f32x4_t z = {0.f, 0.f, 0.f, 0.f};
            mma16(z, b, f.a[u][s]);
            acc.t[rb0 + u][cb0 + v] = z;
          } else {
            mma16(acc.t[rb0 + u][cb0 + v], b, f.a[u][s]);
          }
        }
      }
  } else {
    constexpr int i0 = (P >= 2) ? 2 : 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint4& b = (j == 0) ? f.bl[s] : f.bh[s];
      mma32(acc[i0][j], b, f.a[0][s], bf16_t());
      mma32(acc[i0 + 1][j], b, f.a[1][s], bf16_t());
    }
  }
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// One K-tile = 4 phases.  TAIL: 0 = steady state (all issue, vmcnt 8);
// 1 = second-to-last tile (P0,P1 issue; then 6, 4); 2 = last tile (2, 0, -, -).
// VMR (with relaxed = true): the count for the first K-tile of a tile whose predecessor's stores are still in the
// queue (they sit between this tile's half-tiles 0..5 and 6.. in issue order).
// XL (TAIL == 2 only): loads of the epilogue's block 0 issued right before this last K-tile (newer than every DMA).
template <int PAR, int TAIL, int VMR = 8, int XL = 0, typename FR, typename ACC>
__device__ __forceinline__ void ktile(const Ctx& c, FR& f, ACC& acc, uint32_t kb1, uint32_t kb2,
                                      bool relaxed = false) {
  if constexpr (TAIL == 0) {
    constexpr bool FIRST = VMR != 8 && std::is_same<FR, Frags16>::value;     // (VMR != 8 marks the first K-tile of a tile)
    phase<0, PAR, true, 8, VMR, 0, FIRST>(c, f, acc, kb1, kb2, relaxed);
    phase<1, PAR, true, 8, VMR, 0, FIRST>(c, f, acc, kb1, kb2, relaxed);
    phase<2, PAR, true, 8, VMR, 0, FIRST>(c, f, acc, kb1, kb2, relaxed);
    phase<3, PAR, true, 8, VMR, 0, FIRST>(c, f, acc, kb1, kb2, relaxed);
  } else if constexpr (TAIL == 1) {
    phase<0, PAR, true, 8>(c, f, acc, kb1, kb2);
    phase<1, PAR, true, 8>(c, f, acc, kb1, kb2);
    phase<2, PAR, false, 6>(c, f, acc, kb1, kb2);
    phase<3, PAR, false, 4>(c, f, acc, kb1, kb2);
  } else {
    phase<0, PAR, false, 2 + XL, 2 + XL, XL>(c, f, acc, kb1, kb2);
    phase<1, PAR, false, 0 + XL, 0 + XL, XL>(c, f, acc, kb1, kb2);
    phase<2, PAR, false, -1>(c, f, acc, kb1, kb2);
    phase<3, PAR, false, -1>(c, f, acc, kb1, kb2);
  }
}

constexpr int kStage = 6 * kSlot;      // epilogue staging lives in [96 KiB, 160 KiB): ring slots 6, 7 + 32 KiB
constexpr int kLds = 160 * 1024;

// Persistent kernel: workgroup b walks tiles b, b + grid, b + 2*grid, ... (XCD-aware order).
// ACT: the activation as a compile-time constant (ACT_NONE / ACT_QUICKGELU / ACT_GELU_ERF: straight-line epilogue code), or
// kActRuntime: read GemmArgs::act (ReLU flavours of the ModifiedResNet tower, anything else).
constexpr int kActRuntime = -1;
template <bool FAST, bool HAS_R, bool HAS_U, bool HAS_C2, bool HAS_LN, bool HAS_PS, int ACT>
__global__ __launch_bounds__(kThreads8, 2) void gemm_nt_8p_kernel(GemmArgs p, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int h = lane >> 5, l31 = lane & 31;
  const int tiles_n = p.N >> 8;

  Ctx c;
  c.smem = smem;
  c.lds_base = (uint32_t)(size_t)smem;
  const uint32_t lda_b = (uint32_t)p.lda * 2u, ldb_b = (uint32_t)p.ldb * 2u;
  c.srdA = make_srd(p.A, (uint32_t)(p.M - 1) * lda_b + (uint32_t)p.K * 2u);
  c.srdB = make_srd(p.B, (uint32_t)(p.N - 1) * ldb_b + (uint32_t)p.K * 2u);
  c.hiA = 64u * lda_b;
  c.hiB = 32u * ldb_b;
  c.dma_dst = wave * 2048;
#if EZ_MI16
  {   // 16x16x32 fragments: lane (row l15 of a 16-row block, k-quarter q4 of a 32-k step): chunk 4 s + q4; the swizzle (row >> 1) & 7
      // of row 16 u + l15 is (l15 >> 1) & 7 for every block u, so the blocks are plain + u * 2048 offsets
    const int l15 = lane & 15, q4 = lane >> 4;
    const int sw = (l15 >> 1) & 7;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint32_t ch = (uint32_t)((4 * s + q4) ^ sw) << 4;
      c.rdA[s] = (uint32_t)(wm * 64 + l15) * 128 + ch;
      c.rdB[s] = (uint32_t)(wn * 32 + l15) * 128 + ch;
    }
    c.rdA[2] = c.rdA[3] = c.rdB[2] = c.rdB[3] = 0;
    (void)h; (void)l31;
  }
#else
  {
    const int sw = (l31 >> 1) & 7;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint32_t ch = (uint32_t)((2 * s + h) ^ sw) << 4;
      c.rdA[s] = (uint32_t)(wm * 64 + l31) * 128 + ch;
      c.rdB[s] = (uint32_t)(wn * 32 + l31) * 128 + ch;
    }
  }
#endif
  // Tile order.  An XCD works on 32 consecutive logical tiles at a time (xcd_remap), and what its L2 has to fetch per
  // round is one A row panel per distinct tile row + one B panel per distinct tile column.  n-fastest order makes that
  // 32 / tiles_n rows + tiles_n columns (N = 3072: 2.7 + 12); with raster_gm = g the tiles of g consecutive tile rows are
  // walked column by column, so a round covers g rows x 32 / g columns (g = 4: 4 + 8) -- fewer panels through the fabric.
  const int tiles_m = (p.M + 255) >> 8;
  const int raster_gm = p.raster_gm % 10000;        // (the launcher packs the de-phasing request above it)
  auto tile_origin = [&](int v, int& m0, int& n0) {
    const int t = xcd_remap(v, ntiles);
    int tm, tn;
    if (raster_gm >= 100) {
      // SUPER-COLUMN order (round 5, the default for N >= 2304: gemm8p.hip): the logical order is [super-column of w tile columns]
      // [tile row][column inside it], and xcd_remap hands every XCD one contiguous eighth of it -- an XCD stays inside ONE super-column
      // for (almost) its whole walk, so the w B panels it needs (w x 256 x K x 2 bytes: 2.4 MB at w = 6, K = 768) stay resident in its
      // 4 MiB L2 while the A row panels stream through once per super-column.  n-fastest order needs ALL tiles_n B panels per 32-tile
      // round (4.7 MB at N = 3072: they do not fit, and are re-fetched through the fabric every round -- the 1.9-2.4 x traffic of the
      // N >= 2304 products in profiles/r4_gemm_traffic.json).  w = 6: vit.qkv 0.612 -> 0.603 ms, vit.fc 0.888 -> 0.871 ms, forward step
      // +0.7 % (profiles/r5_gemm_supercolumn.log); w = 3 and w = 4 lose (A is then fetched by 3-4 groups of XCDs).
      const int w = raster_gm - 100;
      const int per = tiles_m * w;
      const int sc = t / per, u = t - sc * per;
      const int wsz = min(w, tiles_n - sc * w);
      tm = u / wsz;
      tn = sc * w + (u - tm * wsz);
    } else if (raster_gm > 0) {
      const int per = raster_gm * tiles_n;
      const int grp = t / per, u = t - grp * per;
      const int first = grp * raster_gm;
      const int gsz = min(raster_gm, tiles_m - first);
      tn = u / gsz;
      tm = first + (u - tn * gsz);
    } else {
      tm = t / tiles_n;
      tn = t - tm * tiles_n;
    }
    m0 = tm << 8;
    n0 = tn << 8;
  };
  // per-lane DMA source offsets of a tile; recomputed from the lane id every time (a handful of integer ops):
  // kept live across the main loop they get spilled, and the reload's compiler-counted vmcnt drains the queue
  auto set_tile = [&](int m0, int n0) {
    const int ln = lane_id_now();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = (wave * 2 + i) * 8 + (ln >> 3);             // row of the half-tile image
      const uint32_t chk = (uint32_t)((ln & 7) ^ ((lr >> 1) & 7)) << 4;
      c.voffA[i] = (uint32_t)(m0 + (lr >> 6) * 128 + (lr & 63)) * lda_b + chk;
      c.voffB[i] = (uint32_t)(n0 + (lr >> 5) * 64 + (lr & 31)) * ldb_b + chk;
    }
  };
  auto issue_prologue = [&]() {      // half-tiles 0..5 of the tile described by c.voff*
    const uint32_t d = c.lds_base + c.dma_dst;
    dma16(d + 0 * kSlot, c.voffA[0], c.srdA, 0);
    dma16(d + 0 * kSlot + 1024, c.voffA[1], c.srdA, 0);
    dma16<1>(d + 1 * kSlot, c.voffB[0], c.srdB, 0);
    dma16<1>(d + 1 * kSlot + 1024, c.voffB[1], c.srdB, 0);
    dma16<1>(d + 2 * kSlot, c.voffB[0], c.srdB, c.hiB);
    dma16<1>(d + 2 * kSlot + 1024, c.voffB[1], c.srdB, c.hiB);
    dma16(d + 3 * kSlot, c.voffA[0], c.srdA, c.hiA);
    dma16(d + 3 * kSlot + 1024, c.voffA[1], c.srdA, c.hiA);
    dma16(d + 4 * kSlot, c.voffA[0], c.srdA, 128);
    dma16(d + 4 * kSlot + 1024, c.voffA[1], c.srdA, 128);
    dma16<1>(d + 5 * kSlot, c.voffB[0], c.srdB, 128);
    dma16<1>(d + 5 * kSlot + 1024, c.voffB[1], c.srdB, 128);
  };

  // De-phasing (gemm8p.hip set_gemm_dephase): the CUs of an XCD start their first tile ((b >> 3) % period) * steps * 1024 clocks apart
  // and -- every tile taking the same time -- stay that far apart: their epilogues (the C stream, matrix pipe idle) no longer coincide.
  {
    const int dp = p.raster_gm / 10000;
    if (dp > 0) {
      const int code = dp / 100, steps = dp - code * 100;
      const unsigned period = code == 0 ? 32u : (1u << code);
      for (int d = (int)((blockIdx.x >> 3) % period) * steps; d > 0; --d) __builtin_amdgcn_s_sleep(16);
    }
  }
  const EpiCtx ep = make_epi_ctx<HAS_R, HAS_U, HAS_C2, HAS_LN, HAS_PS>(p);
  constexpr int NS = kStoresPerBlock * (1 + (HAS_C2 ? 1 : 0) + (HAS_PS ? 1 : 0));   // stores per 32-row block
  int v = blockIdx.x, m0, n0;
  tile_origin(v, m0, n0);
  set_tile(m0, n0);
  issue_prologue();
  wait_role<8, 0, 0, true>();           // half-tiles 0 and 1 (this wave's pieces)
  bool first = true;

#if EZ_MI16
  using FragsT = Frags16;
#else
  f32x16_t acc[4][2];                   // (re-zeroed block by block in the epilogue)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  using FragsT = Frags;
#endif

  for (;;) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();       // ... and everyone else's; every wave is out of the previous epilogue
    if (wm == 1) __builtin_amdgcn_s_barrier();   // wave row 1 runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);

    FragsT f;
#if EZ_MI16
    Acc16 acc;                            // defined by the first K-tile's MFMAs (zero C operand): nothing carried between tiles
#endif
    const int nk = p.K >> 6;              // even, >= 4 (checked by the launcher)
    uint32_t kb = 0;                      // byte offset of the current tile's k range
    // first K-tile: the previous tile's 4*NS stores may still be in flight between half-tiles 0..5 and 6..
    ktile<0, 0, 8 + 4 * NS>(c, f, acc, kb + 128, kb + 256, !first);
    ktile<1, 0>(c, f, acc, kb + 256, kb + 384);
    kb += 256;
    for (int kt = 2; kt < nk - 2; kt += 2) {
      ktile<0, 0>(c, f, acc, kb + 128, kb + 256);
      ktile<1, 0>(c, f, acc, kb + 256, kb + 384);
      kb += 256;
    }
    ktile<0, 1>(c, f, acc, kb + 128, kb + 256);
    // residual / u / row-stat loads of the epilogue's first 32-row block: one K-tile of MFMAs to hide their latency
    EpiLoads eld;
    constexpr bool kLate0 = HAS_PS || (HAS_U && ACT == ACT_GELU_ERF);      // (register pressure: see epilogue_rows ISSUE0)
    constexpr int NL0 = kLate0 ? 0 : 4 * ((HAS_R ? 1 : 0) + (HAS_U ? 1 : 0) + (HAS_LN ? 1 : 0));
    if constexpr (NL0 > 0) epilogue_issue_block<HAS_R, HAS_U, HAS_LN, 0, 0>(ep, m0 + wm * 128, n0 + wn * 64, eld);
    ktile<1, 2, 8, NL0>(c, f, acc, 0, 0);
    if (wm == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // every wave is past its last LDS read and no DMA is in flight: the ring is free

    // ---- epilogue --------------------------------------------------------------------------------
    // One 32-row block of the wave tile at a time: its 32 x 64 fp32 accumulators go through a wave-private
    // 8 KiB LDS image (16-byte chunk index XORed with row & 7: conflict-free both ways) and come back
    // row-coalesced: lane (crow = lane >> 3, g = lane & 7) holds 8 consecutive columns of row it*8 + crow.
    // All epilogue math (alpha, bias, activation or act'(U), residual) runs in that layout in fp32 with
    // one rounding; every global access is 16 B per lane = 8 full 128-byte rows per wave instruction.
    // VMEM stream of one epilogue (all counts static):
    //   bias(2) L0 L1 L2 [D = 12 DMA of the next tile] | S0 L3 | S1 | S2 | S3       (Lb: NL loads, Sb: NS stores)
    // (three of the four residual / u blocks are requested up front -- the fragment registers are dead by now --
    // and the fourth as soon as block 0's accumulators are gone, so that only block 0 can see HBM latency)
    const int vn = v + (int)gridDim.x;
    const bool has_next = vn < ntiles;
    int m0n = 0, n0n = 0;
    if (has_next) tile_origin(vn, m0n, n0n);
    // The next tile's first six half-tiles land in ring slots 0..5 while this epilogue runs.  After the last tile the
    // same twelve DMAs are issued anyway (re-reading this tile's first half-tiles into the dead ring): every counted
    // wait of the epilogue is then a single unconditional statement -- a branch around two asm waits made hipcc copy
    // load destinations before the wait that guards them.
    epilogue_rows<FAST, HAS_R, HAS_U, HAS_C2, 12, HAS_LN, HAS_PS, ACT, kLate0>(ep, acc, m0 + wm * 128, n0 + wn * 64, smem + kStage + wave * 8192, p.act, eld,
                                                  [&]() {
                                                    if (has_next) set_tile(m0n, n0n);
                                                    issue_prologue();
                                                  });
    if (!has_next) break;
    wait_role<6, 4 * NS, 0, true>();   // half-tiles 0..2 of the next tile have landed; 3..5 and this tile's stores may still fly
    first = false;
    v = vn; m0 = m0n; n0 = n0n;
  }
}



}  // namespace

namespace nt8p {
constexpr int kActRuntime = ::ezclip::kActRuntime;

template <bool R, bool U, bool C2, bool LN, bool PS, int ACT>
int launch_8p(const GemmArgs& p, int tiles, int grid, hipStream_t stream) {
  static LdsOptIn lds_opt;
  auto* kern = &gemm_nt_8p_kernel<true, R, U, C2, LN, PS, ACT>;
  EZ_ENSURE_LDS(kern, lds_opt, kLds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads8), kLds, stream, p, tiles);
  return EZ_OK;
}

// the instantiated combinations (gemm8p_nt_{a,b,c}.hip define them, gemm8p.hip dispatches to them)
#define EZ_8P_INSTANCES_A(X)                                                     \
  X(false, false, false, false, false, ACT_NONE)                                 \
  X(false, false, false, false, false, ACT_QUICKGELU)                            \
  X(false, false, false, false, false, ACT_GELU_ERF)                             \
  X(false, false, false, false, false, kActRuntime)                              \
  X(true, false, false, false, true, ACT_NONE)
#define EZ_8P_INSTANCES_B(X)                                                     \
  X(true, false, false, false, false, ACT_NONE)                                  \
  X(true, false, false, false, false, kActRuntime)                               \
  X(false, true, false, false, false, ACT_QUICKGELU)                             \
  X(false, true, false, false, false, ACT_GELU_ERF)                              \
  X(false, true, false, false, false, kActRuntime)
#define EZ_8P_INSTANCES_C(X)                                                     \
  X(false, false, true, false, false, ACT_QUICKGELU)                             \
  X(false, false, true, false, false, ACT_GELU_ERF)                              \
  X(false, false, true, false, false, kActRuntime)                               \
  X(false, false, false, true, false, ACT_NONE)                                  \
  X(false, false, false, true, false, ACT_QUICKGELU)                             \
  X(false, false, false, true, false, kActRuntime)
#define EZ_8P_DECLARE(R, U, C2, LN, PS, ACT) \
  extern template int launch_8p<R, U, C2, LN, PS, ACT>(const GemmArgs&, int, int, hipStream_t);
#define EZ_8P_DEFINE(R, U, C2, LN, PS, ACT) \
  template int launch_8p<R, U, C2, LN, PS, ACT>(const GemmArgs&, int, int, hipStream_t);

}  // namespace nt8p
}  // namespace ezclip
