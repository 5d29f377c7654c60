// Shared pieces of the 8-phase GEMM pipelines (gemm8p.hip): hand-issued LDS-DMA and counted waits,
// buffer descriptors, the row-coalesced fp32 epilogue.  gfx950 only.
#pragma once
#include <type_traits>

#include "ezclip_common.h"
#include "kernels.h"

namespace ezclip {
namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4_t;

constexpr int kSlot = 16384;           // one half-tile: 128 rows x 128 B
constexpr int kRing = 8 * kSlot;       // 128 KiB
constexpr int kThreads8 = 512;

// LDS-DMA: 64 lanes x 16 B land at lds_dst + lane*16 (wave-uniform destination).
// EZ_ROLE_X / EZ_ROLE_Y (tools/build_variants.py; timing only, results are wrong by construction): is the store tail of the
// epilogue the in-order vmcnt of the wave that also waits for the LDS-DMA, or the CU's memory pipeline?  In both variants wave
// row 0 issues every DMA of the workgroup (its own pieces twice: same volume into LDS) and wave row 1 none; the C stores are
// issued by wave row 1 only (X: the DMA waves never have a store in their queue) or by wave row 0 only (Y: they do).  Same
// traffic, same instruction counts per CU -- X faster than Y means a role split would hide the tail.
#if defined(EZ_ROLE_X) || defined(EZ_ROLE_Y)
#define EZ_ROLES 1
__device__ __forceinline__ bool ez_dma_role() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0; }
#ifdef EZ_ROLE_X
__device__ __forceinline__ bool ez_store_role() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 1; }
#else
__device__ __forceinline__ bool ez_store_role() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0; }
#endif
#else
#define EZ_ROLES 0
#endif

// Cache-policy bits of the LDS-DMA loads, per operand (experiment hooks of tools/build_variants.py: -DEZ_DMA_A_MOD='" nt"' etc.; default none).
#ifndef EZ_DMA_A_MOD
#define EZ_DMA_A_MOD ""
#endif
#ifndef EZ_DMA_B_MOD
#define EZ_DMA_B_MOD ""
#endif
template <int OPERAND>   // 0: A (activations: streamed once per column tile), 1: B (weights: re-read by every row tile)
__device__ __forceinline__ void dma16_one(uint32_t lds_dst, uint32_t voff, const i32x4_t& srd, uint32_t soff) {
  uint32_t keep;
  if constexpr (OPERAND == 0) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, %4 offen" EZ_DMA_A_MOD " lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(srd), "s"(soff)
        : "memory");
  } else {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, %4 offen" EZ_DMA_B_MOD " lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(srd), "s"(soff)
        : "memory");
  }
}
template <int OPERAND = 0>
__device__ __forceinline__ void dma16(uint32_t lds_dst, uint32_t voff, const i32x4_t& srd, uint32_t soff) {
#ifdef EZ_ABL_NODMA         // timing ablation (results wrong by construction): no global -> LDS traffic at all; the counted waits see an empty queue
  asm volatile("" ::"s"(lds_dst), "v"(voff), "s"(soff));
  return;
#endif
#ifdef EZ_ABL_DMASAME       // timing ablation: the same instruction stream, every DMA aimed at the first KiB of the operand (L1 / L2 hits:
  voff &= 0x3f0u; soff = 0;  // issue cost and LDS writes stay, L2 / fabric / HBM traffic goes)
#endif
#if EZ_ROLES
  if (!ez_dma_role()) return;
  dma16_one<OPERAND>(lds_dst, voff, srd, soff);
  dma16_one<OPERAND>(lds_dst + 8192u, voff, srd, soff);     // the partner wave's region of the half-tile image
#else
  dma16_one<OPERAND>(lds_dst, voff, srd, soff);
#endif
}

// Lane id recomputed on the spot (v_mbcnt): an asm volatile cannot be hoisted out of the tile loop, so values
// derived from it are not kept live (and spilled) across the main loop.
__device__ __forceinline__ int lane_id_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  if constexpr (N >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Counted wait written as (DMAs, stores, other loads) that may stay in flight.  Without the role experiment: their sum.
// MAIN: a wait of the main loop (for LDS-DMA only): waves that issue no DMA skip it.
template <int ND, int NST, int NLD, bool MAIN>
__device__ __forceinline__ void wait_role() {
#if EZ_ROLES
  if (ez_dma_role()) {
    if (ez_store_role()) wait_vm<2 * ND + NST + NLD>(); else wait_vm<2 * ND + NLD>();
  } else if (!MAIN) {
    if (ez_store_role()) wait_vm<NST + NLD>(); else wait_vm<NLD>();
  }
#else
  wait_vm<ND + NST + NLD>();
#endif
}

__device__ __forceinline__ i32x4_t make_srd(const void* base, uint32_t bytes) {
  const uint64_t a = (uint64_t)base;
  i32x4_t r;
  r.x = (int)(uint32_t)a;
  r.y = (int)(uint32_t)((a >> 32) & 0xffffu);
  r.z = (int)bytes;
  r.w = 0x00020000;
  return r;
}

struct Frags {
  uint4 a[2][4];   // current A half: [i'][k-step]
  uint4 bl[4];     // B-lo, kept from P0 to P3
  uint4 bh[4];     // B-hi, kept from P1 to P2
};

// 16x16x32 layout of the same 128 x 64 wave tile (EZ_MI16): 8 x 4 accumulators of 16 x 16, fragments per 16 rows and 32 k
struct Frags16 {
  uint4 a[4][2];   // current A half: [16-row block][k-step of 32]
  uint4 bl[2][2];  // B-lo, kept from P0 to P3: [16-column block][k-step]
  uint4 bh[2][2];  // B-hi, kept from P1 to P2
};
struct Acc16 {
  f32x4_t t[8][4]; // [16-row block][16-column block]: lane (m = lane & 15, q = lane >> 4) holds columns 4 q .. 4 q + 3 of row m
};
struct Acc32 {
  f32x16_t t[4][2];
};

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// ---- epilogue helpers ---------------------------------------------------------------------------
// Every global LOAD of the epilogue is hand-issued too (buffer_load in inline asm, bounds-checked): with
// LDS-DMA of the NEXT tile in flight, a compiler-counted vmcnt would drain the DMA queue at every use.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__device__ __forceinline__ void ldg16(u32x4_t& dst, uint32_t voff, const i32x4_t& srd, uint32_t soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
// counted wait that also pins the destinations: nothing may read them above this statement
template <int N>
__device__ __forceinline__ void wait_vm4(u32x4_t& a, u32x4_t& b, u32x4_t& c, u32x4_t& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm2(u32x4_t& a, u32x4_t& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

__device__ __forceinline__ void unpack8(const u32x4_t& c, float (&v)[8]) {
  v[0] = __uint_as_float(c.x << 16); v[1] = __uint_as_float(c.x & 0xffff0000u);
  v[2] = __uint_as_float(c.y << 16); v[3] = __uint_as_float(c.y & 0xffff0000u);
  v[4] = __uint_as_float(c.z << 16); v[5] = __uint_as_float(c.z & 0xffff0000u);
  v[6] = __uint_as_float(c.w << 16); v[7] = __uint_as_float(c.w & 0xffff0000u);
}


// ---- row-coalesced epilogue -------------------------------------------------------------------------------------
struct EpiCtx {
  uint32_t ldc_b, ldr_b, ldu_b;
  i32x4_t srdR, srdU, srdBias;
  i32x4_t srdStats, srdLn1, srdLn2;   // folded LayerNorm: row stats, c1, c2
  i32x4_t srdC, srdC2;
  i32x4_t srdPS;                      // row-stat partials [M][N/64][2] f32 (HAS_PS)
  uint32_t ps_row_b;                  // bytes per row of it
  float* colsum;
  i32x4_t srdColsum;                  // [N] f32: fused column sums of the act'(U) epilogue (buffer atomics: no 64-bit lane address)
  float scale;
  int M;
};

template <bool HAS_R, bool HAS_U, bool HAS_C2, bool HAS_LN = false, bool HAS_PS = false>
__device__ __forceinline__ EpiCtx make_epi_ctx(const GemmArgs& p) {
  EpiCtx e;
  e.ps_row_b = (uint32_t)(p.N >> 6) * 8u;
  e.srdPS = make_srd(p.rowstat_part, HAS_PS ? (uint32_t)p.M * e.ps_row_b : 0u);
  e.ldc_b = (uint32_t)p.ldc * 2u; e.ldr_b = (uint32_t)p.ldr * 2u; e.ldu_b = (uint32_t)p.ldu * 2u;
  e.srdR = make_srd(p.R, HAS_R ? (uint32_t)p.M * e.ldr_b : 0u);
  e.srdU = make_srd(p.U, HAS_U ? (uint32_t)p.M * e.ldu_b : 0u);
  e.srdBias = make_srd(p.bias, p.bias ? (uint32_t)p.N * 4u : 0u);   // no bias: reads return 0
  e.srdStats = make_srd(p.ln_stats, HAS_LN ? (uint32_t)p.M * 8u : 0u);
  e.srdLn1 = make_srd(p.ln_c1, HAS_LN ? (uint32_t)p.N * 4u : 0u);
  e.srdLn2 = make_srd(p.ln_c2, HAS_LN ? (uint32_t)p.N * 4u : 0u);
  e.srdC = make_srd(p.C, (uint32_t)p.M * e.ldc_b);
  e.srdC2 = make_srd(HAS_C2 ? p.C2 : p.C, (uint32_t)p.M * e.ldc_b);
  e.scale = p.alpha;
  e.M = p.M;
  e.colsum = p.colsum;
  e.srdColsum = make_srd(p.colsum, (HAS_U && p.colsum) ? (uint32_t)p.N * 4u : 0u);
  return e;
}

// Epilogue of one 128 x 64 wave tile (4 x 2 MFMA 32x32 accumulators, lane = row l31, 4 consecutive columns per
// register quad).  One 32-row block at a time: its 32 x 64 fp32 accumulators go through the wave-private 8 KiB LDS
// image W (16-byte chunk index XORed with row & 7: conflict-free both ways) and come back row-coalesced: lane
// (crow = lane >> 3, g = lane & 7) holds 8 consecutive columns of row it*8 + crow.  All epilogue math (alpha, bias,
// activation or act'(U), residual) runs in that layout in fp32 with one rounding; every global access is 16 B per
// lane = 8 full 128-byte rows per wave instruction.  The accumulators are re-zeroed on the way out.
// Every global LOAD is hand-issued (buffer_load asm) with counted waits, because the caller may have D LDS-DMA
// loads of its next tile in flight (issued by issue_dma() right after the epilogue's own loads): a compiler-counted
// vmcnt would drain them.  VMEM stream (all counts static; Lb: NL loads, Sb: NS stores):
//   L0 (by the caller, one K-tile earlier) ... bias(2) L1 L2 [D x DMA] | S0 L3 | S1 | S2 | S3
// Blocks 1 and 2 are requested up front and block 3 as soon as block 0's accumulators are gone.  On return 4*NS stores (and the D DMAs) may still be in flight.
// 16-byte buffer store, hand-issued WITH its wait states: the data registers of a dwordx3/x4 store must not be
// written by the next VALU instruction(s).  hipcc pads that hazard for flat/global stores but exempts MUBUF stores
// that carry an SGPR soffset -- on gfx950 the exemption does not hold (seen as corrupted second dwords in the last
// lanes of each 16 when a v_pk_fma_f32 reused the data registers right after the builtin's store).
// EZ_STG_MOD / EZ_STG_SAMEADDR: experiment hooks of tools/build_variants.py (cache-policy bits on the store; all stores
// of a wave aimed at one 1 KiB region = same instruction stream without the DRAM write traffic).
// EZ_ABL_NOSTORE / EZ_ABL_NOEPI: timing ablations (results are wrong by construction).
// Default " nt": C is a write-once stream; the non-temporal hint measured +2 % on the K=768 products and +0.9 % on the
// forward step (profiles/README.md, store experiments); sc1 / sc0 sc1 made no difference.
#ifndef EZ_STG_MOD
#define EZ_STG_MOD " nt"
#endif
#if defined(EZ_ABL_NOSTORE) || defined(EZ_ABL_NOEPI)
constexpr int kStoresPerBlock = 0;
#else
constexpr int kStoresPerBlock = 4;
#endif
__device__ __forceinline__ void stg16(const u32x4_t& data, uint32_t voff, const i32x4_t& srd, uint32_t soff) {
#if defined(EZ_ABL_NOSTORE) || defined(EZ_ABL_NOEPI)
  asm volatile("" ::"v"(data), "v"(voff), "s"(soff));
  return;
#endif
#if EZ_ROLES
  if (!ez_store_role()) return;
#endif
#ifdef EZ_STG_SAMEADDR
  voff = (voff & 0x70u) | ((threadIdx.x & 0x1f8u) << 4); soff = 0;
#endif
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" EZ_STG_MOD "\n\ts_nop 1" ::"v"(data), "v"(voff), "s"(srd), "s"(soff) : "memory");
}

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
// 8-byte store; lanes that must not store pass voff = 0xFFFFFFF0 (out of range: dropped by the descriptor)
__device__ __forceinline__ void stg8(const u32x2_t& data, uint32_t voff, const i32x4_t& srd, uint32_t soff) {
#if defined(EZ_ABL_NOSTORE) || defined(EZ_ABL_NOEPI)
  asm volatile("" ::"v"(data), "v"(voff), "s"(soff));
  return;
#endif
  asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen\n\ts_nop 0" ::"v"(data), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void ldg8(u32x2_t& dst, uint32_t voff, const i32x4_t& srd, uint32_t soff) {
  asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm4s(u32x2_t& a, u32x2_t& b, u32x2_t& c, u32x2_t& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}

// Destination registers of the epilogue's per-block loads (residual / pre-activation / LayerNorm row stats).  Owned by
// the kernel so that block 0 can be requested BEFORE the last K-tile of the main loop (epilogue_issue_block0): its
// HBM latency then hides under the last 32 MFMAs instead of stalling the first block of the epilogue.
struct EpiLoads {
  u32x4_t r[4][4], u[4][4];
  u32x2_t s[4][4];
};

template <bool HAS_R, bool HAS_U, bool HAS_LN, int B, int I>
__device__ __forceinline__ void epilogue_issue_block(const EpiCtx& ec, int mw, int nw, EpiLoads& ld) {
  const int eln = lane_id_now();
  const int crow = eln >> 3, g = eln & 7;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const uint32_t row = (uint32_t)(mw + I * 32 + it * 8);
    if constexpr (HAS_R) ldg16(ld.r[B][it], (uint32_t)crow * ec.ldr_b + (uint32_t)g * 16u, ec.srdR, row * ec.ldr_b + (uint32_t)nw * 2u);
    if constexpr (HAS_U) ldg16(ld.u[B][it], (uint32_t)crow * ec.ldu_b + (uint32_t)g * 16u, ec.srdU, row * ec.ldu_b + (uint32_t)nw * 2u);
    if constexpr (HAS_LN) ldg8(ld.s[B][it], (uint32_t)crow * 8u, ec.srdStats, row * 8u);
  }
}

// HAS_LN: folded LayerNorm -- y = acc * rstd_m + (-mean_m rstd_m) * c1[n] + c2[n]  (see GemmArgs::ln_stats)
// HAS_PS: also emit the per-row (sum, sum of squares) of this wave's 64 rounded output columns (GemmArgs::rowstat_part)
// 32-row block I of a wave tile's accumulators -> the wave's LDS image W (row r at W + 256 r, 16-byte chunk index = column / 4
// XORed with r & 7); the 32x32 layout zeroes them for the next tile, the 16x16 layout's first K-tile starts from a literal zero.  Two accumulator layouts: 4 x 2 MFMA 32x32 tiles (lane = row l31, columns
// j*32 + q*8 + h*4 .. +3) and 8 x 4 MFMA 16x16 tiles (lane = row l15 of a 16-row block, columns cb*16 + (lane >> 4)*4 .. +3).
template <int I>
__device__ __forceinline__ void acc_block_to_lds(f32x16_t (&acc)[4][2], char* W, int eln) {
  const int eh = eln >> 5, el31 = eln & 31;
  const uint32_t wr_row = (uint32_t)el31 * 256u, wr_sw = (uint32_t)(el31 & 7);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t ch = (uint32_t)(j * 8 + q * 2 + eh) ^ wr_sw;
      *reinterpret_cast<float4*>(W + wr_row + (ch << 4)) =
          make_float4(acc[I][j][q * 4], acc[I][j][q * 4 + 1], acc[I][j][q * 4 + 2], acc[I][j][q * 4 + 3]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[I][j][r] = 0.f;
  }
}
template <int I>
__device__ __forceinline__ void acc_block_to_lds(Acc16& acc, char* W, int eln) {
  // row 16 u + l15, chunk (4 cb + q4) ^ (row & 7): row & 7 = l15 & 7 for both u, and 4 cb / q4 occupy disjoint bits, so the byte
  // address is ((l15 * 256 + ((q4 ^ (l15 & 7)) << 4)) ^ (cb << 6)) + 4096 u -- ONE live register, an xor per column block and
  // the 16-row block as the instruction's immediate offset
  const uint32_t a0 = (uint32_t)(eln & 15) * 256u + ((uint32_t)((eln >> 4) ^ (eln & 7)) << 4);
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    char* wp = W + (a0 ^ (uint32_t)(cb << 6));
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x4_t& t = acc.t[2 * I + u][cb];
      *reinterpret_cast<float4*>(wp + u * 4096) = make_float4(t[0], t[1], t[2], t[3]);
    }
  }
}
__device__ __forceinline__ void acc_keep_and_zero(f32x16_t (&acc)[4][2]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      asm volatile("" ::"v"(acc[i][j]));       // keep the MFMAs alive
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
}
__device__ __forceinline__ void acc_keep_and_zero(Acc16& acc) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      asm volatile("" ::"v"(acc.t[i][j]));
#pragma unroll
      for (int r = 0; r < 4; ++r) acc.t[i][j][r] = 0.f;
    }
}

// ACT >= 0: the activation is this compile-time constant (the `act` argument is ignored and every test on it folds away: straight-line
// code, no phi copies of the eight outputs per row); ACT < 0: `act` is read at run time.
// ISSUE0: request block 0's loads here, first thing (same queue order), instead of one K-tile earlier by the caller -- for the two
// instantiations whose registers do not hold 16 more values across the last K-tile (R + row-stat partials, U + erf-GELU').
template <bool FAST, bool HAS_R, bool HAS_U, bool HAS_C2, int D, bool HAS_LN = false, bool HAS_PS = false, int ACT = -1, bool ISSUE0 = false,
          typename ACC, typename IssueDma>
__device__ __forceinline__ void epilogue_rows(const EpiCtx& ec, ACC& acc, int mw, int nw, char* W, int act_rt,
                                              EpiLoads& ld, IssueDma&& issue_dma) {
  const int act = ACT >= 0 ? ACT : act_rt;
  constexpr int NL = 4 * ((HAS_R ? 1 : 0) + (HAS_U ? 1 : 0) + (HAS_LN ? 1 : 0));   // loads per 32-row block
  constexpr int NS = kStoresPerBlock * (1 + (HAS_C2 ? 1 : 0) + (HAS_PS ? 1 : 0));   // stores per 32-row block
#ifdef EZ_ABL_NOEPI
  issue_dma();
  acc_keep_and_zero(acc);
  return;
#endif
  const int eln = lane_id_now();
  const int crow = eln >> 3, g = eln & 7;
  const uint32_t lane_c = (uint32_t)crow * ec.ldc_b + (uint32_t)g * 16u;
  u32x4_t bq[2], c1q[2], c2q[2];
  if constexpr (ISSUE0 && (HAS_R || HAS_U || HAS_LN)) epilogue_issue_block<HAS_R, HAS_U, HAS_LN, 0, 0>(ec, mw, nw, ld);
  ldg16(bq[0], (uint32_t)g * 32u, ec.srdBias, (uint32_t)nw * 4u);
  ldg16(bq[1], (uint32_t)g * 32u + 16u, ec.srdBias, (uint32_t)nw * 4u);
  if constexpr (HAS_LN) {
    ldg16(c1q[0], (uint32_t)g * 32u, ec.srdLn1, (uint32_t)nw * 4u);
    ldg16(c1q[1], (uint32_t)g * 32u + 16u, ec.srdLn1, (uint32_t)nw * 4u);
    ldg16(c2q[0], (uint32_t)g * 32u, ec.srdLn2, (uint32_t)nw * 4u);
    ldg16(c2q[1], (uint32_t)g * 32u + 16u, ec.srdLn2, (uint32_t)nw * 4u);
  }
  if constexpr (NL > 0) {     // (block 0 was requested by the caller, before its last K-tile)
    epilogue_issue_block<HAS_R, HAS_U, HAS_LN, 1, 1>(ec, mw, nw, ld);
    epilogue_issue_block<HAS_R, HAS_U, HAS_LN, 2, 2>(ec, mw, nw, ld);
  }
  issue_dma();
  float bv[8], c1v[8];
  float cs[8];                  // HAS_U: column sums of the output (bias gradient)
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = 0.f;

  static_for<4>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int b = i;
    // accumulators -> LDS (MFMA layout); zeroed for the next tile
    acc_block_to_lds<i>(acc, W, eln);
    float x[4][8];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rr = it * 8 + crow;
      const uint32_t sw = (uint32_t)(rr & 7);
      const float4 x0 = *reinterpret_cast<const float4*>(W + rr * 256 + (((uint32_t)(2 * g) ^ sw) << 4));
      const float4 x1 = *reinterpret_cast<const float4*>(W + rr * 256 + (((uint32_t)(2 * g + 1) ^ sw) << 4));
      x[it][0] = x0.x; x[it][1] = x0.y; x[it][2] = x0.z; x[it][3] = x0.w;
      x[it][4] = x1.x; x[it][5] = x1.y; x[it][6] = x1.z; x[it][7] = x1.w;
    }
    // wait for this block's loads (block 0: also the bias)
    if constexpr (i == 0) {
#if EZ_ROLES
      wait_role<D, 0, 2 * NL, false>();
      wait_vm2<63>(bq[0], bq[1]);
#else
      wait_vm2<2 * NL + D>(bq[0], bq[1]);
#endif
      bv[0] = __uint_as_float(bq[0].x); bv[1] = __uint_as_float(bq[0].y);
      bv[2] = __uint_as_float(bq[0].z); bv[3] = __uint_as_float(bq[0].w);
      bv[4] = __uint_as_float(bq[1].x); bv[5] = __uint_as_float(bq[1].y);
      bv[6] = __uint_as_float(bq[1].z); bv[7] = __uint_as_float(bq[1].w);
      if constexpr (HAS_LN) {     // (bias is null here: bv = 0 + c2)
#if EZ_ROLES
        wait_vm4<63>(c1q[0], c1q[1], c2q[0], c2q[1]);
#else
        wait_vm4<2 * NL + D>(c1q[0], c1q[1], c2q[0], c2q[1]);
#endif
        c1v[0] = __uint_as_float(c1q[0].x); c1v[1] = __uint_as_float(c1q[0].y);
        c1v[2] = __uint_as_float(c1q[0].z); c1v[3] = __uint_as_float(c1q[0].w);
        c1v[4] = __uint_as_float(c1q[1].x); c1v[5] = __uint_as_float(c1q[1].y);
        c1v[6] = __uint_as_float(c1q[1].z); c1v[7] = __uint_as_float(c1q[1].w);
        bv[0] += __uint_as_float(c2q[0].x); bv[1] += __uint_as_float(c2q[0].y);
        bv[2] += __uint_as_float(c2q[0].z); bv[3] += __uint_as_float(c2q[0].w);
        bv[4] += __uint_as_float(c2q[1].x); bv[5] += __uint_as_float(c2q[1].y);
        bv[6] += __uint_as_float(c2q[1].z); bv[7] += __uint_as_float(c2q[1].w);
      }
    }
    if constexpr (NL > 0) {
      // newer than block i's loads:  0: (bias) L1 L2 D   1: L2 D S0 L3   2: D S0 L3 S1   3: S1 S2
#if EZ_ROLES
      if constexpr (i == 0) wait_role<D, 0, 2 * NL, false>();
      else if constexpr (i == 1) wait_role<D, NS, 2 * NL, false>();
      else if constexpr (i == 2) wait_role<D, 2 * NS, NL, false>();
      else wait_role<0, 2 * NS, 0, false>();
      constexpr int cnt = 63;
#else
      constexpr int cnt = (i == 0) ? 2 * NL + D : (i == 1) ? 2 * NL + D + NS : (i == 2) ? NL + D + 2 * NS : 2 * NS;
#endif
      if constexpr (HAS_R) wait_vm4<cnt>(ld.r[b][0], ld.r[b][1], ld.r[b][2], ld.r[b][3]);
      if constexpr (HAS_U) wait_vm4<cnt>(ld.u[b][0], ld.u[b][1], ld.u[b][2], ld.u[b][3]);
      if constexpr (HAS_LN) wait_vm4s<cnt>(ld.s[b][0], ld.s[b][1], ld.s[b][2], ld.s[b][3]);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float (&y)[8] = x[it];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if constexpr (HAS_LN) {
          const float ra = __uint_as_float(ld.s[b][it].x), rb = __uint_as_float(ld.s[b][it].y);
          y[e] = fmaf(y[e], ra, fmaf(rb, c1v[e], bv[e]));
        } else {
          y[e] = y[e] * ec.scale + bv[e];
        }
      }
      const uint32_t soff = (uint32_t)(mw + i * 32 + it * 8) * ec.ldc_b + (uint32_t)nw * 2u;
      if constexpr (HAS_C2) {
        const u32x4_t o = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]),
                           pack_bf16x2(y[6], y[7])};
        stg16(o, lane_c, ec.srdC2, soff);
      }
      if constexpr (HAS_U) {
        float uf[8];
        unpack8(ld.u[b][it], uf);
        if (act == ACT_QUICKGELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] *= act_grad<FAST>(uf[e], ACT_QUICKGELU);
        } else if (act == ACT_GELU_ERF) {
          if constexpr (FAST) {       // packed polynomial (ezclip_common.h): two elements per v_pk_fma_f32
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              const f32x2_t gq = gelu_grad_fast<f32x2_t>(f32x2_t{uf[e], uf[e + 1]});
              y[e] *= gq.x; y[e + 1] *= gq.y;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] *= act_grad<FAST>(uf[e], ACT_GELU_ERF);
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] += y[e];     // rows >= M contribute exact zeros (A rows read as 0, no bias)
      } else if (act == ACT_QUICKGELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = act_apply<FAST>(y[e], ACT_QUICKGELU);
      } else if (act == ACT_GELU_ERF) {
        if constexpr (FAST) {
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const f32x2_t gq = gelu_fast<f32x2_t>(f32x2_t{y[e], y[e + 1]});
            y[e] = gq.x; y[e + 1] = gq.y;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = act_apply<FAST>(y[e], ACT_GELU_ERF);
        }
      }
      if constexpr (!HAS_U) {
        if (act == ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = fmaxf(y[e], 0.f);
        }
      }
      if constexpr (HAS_R) {
        float rf[8];
        unpack8(ld.r[b][it], rf);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] += rf[e];
        if (act == ACT_RELU_POST) {
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = fmaxf(y[e], 0.f);
        }
      }
      const u32x4_t o = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]),
                         pack_bf16x2(y[6], y[7])};
      stg16(o, lane_c, ec.srdC, soff);
      if constexpr (HAS_PS) {
        float v[8];
        unpack8(o, v);             // statistics of what the consumer will read (the rounded values)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1 += v[e]; s2 = fmaf(v[e], v[e], s2); }
        s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
        s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
        s1 += __shfl_xor(s1, 4, 64); s2 += __shfl_xor(s2, 4, 64);
        const u32x2_t pv = {__float_as_uint(s1), __float_as_uint(s2)};
        stg8(pv, g == 0 ? (uint32_t)crow * ec.ps_row_b : 0xFFFFFFF0u, ec.srdPS,
             (uint32_t)(mw + i * 32 + it * 8) * ec.ps_row_b + (uint32_t)(nw >> 6) * 8u);
      }
    }
    if constexpr (NL > 0 && i == 0) epilogue_issue_block<HAS_R, HAS_U, HAS_LN, 3, 3>(ec, mw, nw, ld);
  });
  if constexpr (HAS_U) {
    if (ec.colsum != nullptr) {      // 8 lanes share a column group: combine over crow, then 64 atomics per wave tile
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = cs[e];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        // f32 add without return on a buffer address (descriptor + 32-bit lane offset; lanes with crow != 0 aim out of range and
        // are dropped by the bounds check): a flat `unsafeAtomicAdd(ptr + idx)` keeps a 64-bit index pair live across the whole
        // tile loop -- the two registers this instantiation then spilled (with a compiler-placed vmcnt(0) in the main loop)
        const uint32_t off = crow == 0 ? (uint32_t)(nw + g * 8 + e) * 4u : 0xFFFFFFF0u;
        asm volatile("buffer_atomic_add_f32 %0, %1, %2, 0 offen" ::"v"(v), "v"(off), "s"(ec.srdColsum) : "memory");
      }
    }
  }
}

}  // namespace
}  // namespace ezclip
