// bf16 MFMA GEMM, 256 x 256 x 64 tile, 8-phase software pipeline (gfx950 / CDNA4 only).
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
#include <map>
#include <mutex>
#include <type_traits>

#include "ezclip_common.h"
#include "kernels.h"
#include "gemm_pipe.h"
#include "gemm8p_nt.h"

namespace ezclip {

namespace {

// =====================================================================================================
// Weight-gradient GEMM on the same pipeline:  C[P, Q] (+)= A[M, P]^T . B[M, Q]   (contraction over the ROWS)
//
// dW = dY^T X for every nn.Linear of the path (autograd of easynlp/core/trainer.py:658-661).  Both operands are
// row-major with the contraction index m as the slow dimension, so a K-tile is 64 consecutive rows and its
// half-tiles are [64 m][128 columns] images (256 B per image row, full 128-byte lines from HBM): A-lo / A-hi =
// columns p0 + 0..127 / 128..255, B-lo / B-hi likewise.  The MFMA wants 8 consecutive m per lane for one
// column: ds_read_b64_tr_b16 (the CDNA4 LDS transpose read) delivers exactly that from the row-major image --
// 16 lanes read a 4(m) x 16(col) block, lane c receives column c -- two reads per operand fragment.  The
// 64-byte granule index of an image row is XORed with m & 3 (on the DMA source address and on the read), which
// puts the four rows of a transpose block on four different bank quarters: conflict-free.
// The wave tile is 4 x 2 MFMA blocks as in the NT kernel, but the blocks sit at columns
//   p: wm*64 + i*32 (i = 0,1), 128 + wm*64 + (i-2)*32 (i = 2,3);   q: wn*32 (j = 0), 128 + wn*32 (j = 1)
// so that "lo" and "hi" are contiguous 128-column halves.  Phases, ring, counted waits and the barrier stagger
// are those of the NT kernel.
// The contraction (M ~ 2e5) is split over workgroups: split s of tile t writes its 256 x 256 fp32 partial to
// scratch[s][P][Q] (row-coalesced through LDS), and tn_reduce_kernel adds the partials in a fixed order:
// gradients are bit-reproducible (no atomics).

struct CtxTN {
  const char* smem;
  uint32_t lds_base;
  i32x4_t srdA, srdB;
  uint32_t voffA[2], voffB[2];
  uint32_t dma_dst;
  uint32_t rdA[2], rdB;        // per-lane LDS byte offsets of the transpose reads (A: i' = 0, 1)
};

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;

// one operand fragment (8 consecutive m for column l31): two transpose reads
__device__ __forceinline__ uint4 read_tr(const char* base) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 1024));
  const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
  return make_uint4(a.x, a.y, b.x, b.y);
}

// FR / ACC as in the NT kernel's phase(): Frags + f32x16_t[4][2] (32x32x16) or Frags16 + Acc16 (16x16x32, EZ_MI16).  MI16 fragments:
// lane (column l15 of a 16-column block, k-quarter q4) wants m = 32 s + 8 q4 + 0..7 of its column: the two transpose reads of
// read_tr at image row 32 s + 8 q4 (+ 4); 16-column block u of an A half sits 32 u bytes further along the image row -- bit 5 is
// an immediate, bit 6 meets the row swizzle, hence two lane addresses (u < 2, u >= 2).
template <int P, int PAR, bool ISSUE, int VM, typename FR, typename ACC>
__device__ __forceinline__ void phase_tn(const CtxTN& c, FR& f, ACC& acc, uint32_t soff1,
                                         uint32_t soff2) {
  constexpr int k8 = 4 * PAR + P;
  constexpr bool MI16 = std::is_same<FR, Frags16>::value;
  if constexpr (MI16) {
    if constexpr (P == 0) {
      constexpr int sB = ((k8 + 1) & 7) * kSlot, sA = (k8 & 7) * kSlot;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int v = 0; v < 2; ++v) f.bl[v][s] = read_tr(c.smem + sB + (c.rdB ^ (uint32_t)(v << 5)) + s * 8192);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u) f.a[u][s] = read_tr(c.smem + sA + (c.rdA[0] ^ (uint32_t)(u << 5)) + s * 8192);
    } else if constexpr (P == 1) {
      constexpr int sB = ((k8 + 1) & 7) * kSlot;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int v = 0; v < 2; ++v) f.bh[v][s] = read_tr(c.smem + sB + (c.rdB ^ (uint32_t)(v << 5)) + s * 8192);
    } else if constexpr (P == 2) {
      constexpr int sA = ((k8 + 1) & 7) * kSlot;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u) f.a[u][s] = read_tr(c.smem + sA + (c.rdA[0] ^ (uint32_t)(u << 5)) + s * 8192);
    }
  } else {
  if constexpr (P == 0) {
    constexpr int sB = ((k8 + 1) & 7) * kSlot, sA = (k8 & 7) * kSlot;
#pragma unroll
    for (int s = 0; s < 4; ++s) f.bl[s] = read_tr(c.smem + sB + c.rdB + s * 4096);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f.a[0][s] = read_tr(c.smem + sA + c.rdA[0] + s * 4096);
      f.a[1][s] = read_tr(c.smem + sA + c.rdA[1] + s * 4096);
    }
  } else if constexpr (P == 1) {
    constexpr int sB = ((k8 + 1) & 7) * kSlot;
#pragma unroll
    for (int s = 0; s < 4; ++s) f.bh[s] = read_tr(c.smem + sB + c.rdB + s * 4096);
  } else if constexpr (P == 2) {
    constexpr int sA = ((k8 + 1) & 7) * kSlot;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f.a[0][s] = read_tr(c.smem + sA + c.rdA[0] + s * 4096);
      f.a[1][s] = read_tr(c.smem + sA + c.rdA[1] + s * 4096);
    }
  }
  }
  // soff1 / soff2: byte offset of the first row of K-tile t+1 / t+2 (per operand: A uses .x, B .y -- see caller)
  if constexpr (ISSUE) {
    constexpr int slot = ((k8 + 6) & 7) * kSlot;
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(c.lds_base + slot + c.dma_dst));   // (wave-uniform by construction; the allocator otherwise parks it in a VGPR here)
    if constexpr (P == 0) {          // B-hi(t+1)
      dma16(dst, c.voffB[0], c.srdB, soff1 + 256);
      dma16(dst + 1024, c.voffB[1], c.srdB, soff1 + 256);
    } else if constexpr (P == 1) {   // A-hi(t+1)
      dma16(dst, c.voffA[0], c.srdA, soff1 + 256);
      dma16(dst + 1024, c.voffA[1], c.srdA, soff1 + 256);
    } else if constexpr (P == 2) {   // A-lo(t+2)
      dma16(dst, c.voffA[0], c.srdA, soff2);
      dma16(dst + 1024, c.voffA[1], c.srdA, soff2);
    } else {                         // B-lo(t+2)
      dma16(dst, c.voffB[0], c.srdB, soff2);
      dma16(dst + 1024, c.voffB[1], c.srdB, soff2);
    }
  }
  wait_vm<VM>();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
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
        for (int u = 0; u < 4; ++u) mma16(acc.t[rb0 + u][cb0 + v], b, f.a[u][s]);
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

// The A and B operands have different row strides, so the K-tile byte offsets differ: the caller passes the
// A offsets and the phases derive B's from the ratio-free pair below.
struct KOff { uint32_t a1, a2, b1, b2; };

template <int PAR, int TAIL, typename FR, typename ACC>
__device__ __forceinline__ void ktile_tn(const CtxTN& c, FR& f, ACC& acc, const KOff& k) {
  // P0 issues B (t+1), P1 issues A (t+1), P2 issues A (t+2), P3 issues B (t+2)
  if constexpr (TAIL == 0) {
    phase_tn<0, PAR, true, 8>(c, f, acc, k.b1, 0);
    phase_tn<1, PAR, true, 8>(c, f, acc, k.a1, 0);
    phase_tn<2, PAR, true, 8>(c, f, acc, 0, k.a2);
    phase_tn<3, PAR, true, 8>(c, f, acc, 0, k.b2);
  } else if constexpr (TAIL == 1) {
    phase_tn<0, PAR, true, 8>(c, f, acc, k.b1, 0);
    phase_tn<1, PAR, true, 8>(c, f, acc, k.a1, 0);
    phase_tn<2, PAR, false, 6>(c, f, acc, 0, 0);
    phase_tn<3, PAR, false, 4>(c, f, acc, 0, 0);
  } else {
    phase_tn<0, PAR, false, 2>(c, f, acc, 0, 0);
    phase_tn<1, PAR, false, 0>(c, f, acc, 0, 0);
    phase_tn<2, PAR, false, -1>(c, f, acc, 0, 0);
    phase_tn<3, PAR, false, -1>(c, f, acc, 0, 0);
  }
}

__global__ __launch_bounds__(kThreads8, 2) void gemm_tn_8p_kernel(GemmTNArgs p, float* part, int ntiles, int tiles_q,
                                                                  int kt_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int h = lane >> 5, l31 = lane & 31;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int split = id / ntiles, tile = id - split * ntiles;
  const int tp = tile / tiles_q, tq = tile - tp * tiles_q;
  const int p0 = tp << 8, q0 = tq << 8;          // output tile origin: rows = columns of A, cols = columns of B
  const uint32_t m_begin = (uint32_t)split * (uint32_t)kt_per_split * 64u;

  CtxTN c;
  c.smem = smem;
  c.lds_base = (uint32_t)(size_t)smem;
  const uint32_t lda_b = (uint32_t)p.lda * 2u, ldb_b = (uint32_t)p.ldb * 2u;
  c.srdA = make_srd(p.A, (uint32_t)(p.M - 1) * lda_b + (uint32_t)p.N * 2u);    // rows >= M read as zeros
  c.srdB = make_srd(p.B, (uint32_t)(p.M - 1) * ldb_b + (uint32_t)p.K * 2u);
  c.dma_dst = wave * 2048;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 4 + (lane >> 4);                 // image row (m within the K-tile)
#if EZ_MI16
    // 16x16x32 fragments: a 32-lane pass of the transpose read covers rows 8 q + {0..3} for TWO k-quarters q at one 32-byte column
    // block -- eight rows that must land on eight different 32-byte slots of the 256-byte bank row: the slot index is XORed with
    // f(m) = (m & 3) | ((m >> 3) & 1) << 2  (the 64-byte-granule swizzle of the 32x32x16 layout left rows m and m + 8 on the same banks:
    // two-way conflicts on every fragment read, 22-87 M conflict cycles per launch in profiles/r4_gemm_pmc.md before this)
    const uint32_t ch = (uint32_t)((lane & 15) ^ ((((r & 3) | (((r >> 3) & 1) << 2))) << 1)) << 4;
#else
    const uint32_t ch = (uint32_t)((lane & 15) ^ ((r & 3) << 2)) << 4;   // logical 16-byte chunk of the 256-byte row
#endif
    c.voffA[i] = (m_begin + (uint32_t)r) * lda_b + (uint32_t)p0 * 2u + ch;
    c.voffB[i] = (m_begin + (uint32_t)r) * ldb_b + (uint32_t)q0 * 2u + ch;
  }
#if EZ_MI16
  {
    const int t = lane & 15, q4 = lane >> 4;
    const int row = 8 * q4 + (t >> 2);                               // + 32*s + 4*kk via immediates (neither touches bits 0, 1, 3 of m)
    const uint32_t swz = (uint32_t)((t >> 2) | ((q4 & 1) << 2)) << 5;    // f(m) << 5: the 32-byte slot swizzle of the DMA above
    const uint32_t colA = (uint32_t)(wm * 64 + (t & 3) * 4) * 2u, colB = (uint32_t)(wn * 32 + (t & 3) * 4) * 2u;
    // 16-column block u (v) of the half is 32 u bytes along the row: bits 5-6 (5), which the swizzle also moves -- xor, not add:
    // address(u) = rdA ^ (u << 5), a compile-time constant per fragment (one v_xor in a loop whose vector ALU is idle)
    c.rdA[0] = (uint32_t)row * 256u + (colA ^ swz);
    c.rdA[1] = 0;
    c.rdB = (uint32_t)row * 256u + (colB ^ swz);
    (void)h;
  }
  Acc16 acc;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc.t[i][j][r] = 0.f;
  using FragsT = Frags16;
#else
  {
    const int t = lane & 15, sub = (lane >> 4) & 1, g = h;
    const int row = 8 * g + (t >> 2);                                // + 16*s + 4*kk via immediates
    const uint32_t swz = (uint32_t)(t >> 2) << 6;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t colb = (uint32_t)(wm * 64 + i * 32 + sub * 16 + (t & 3) * 4) * 2u;
      c.rdA[i] = (uint32_t)row * 256u + (colb ^ swz);
    }
    const uint32_t colb = (uint32_t)(wn * 32 + sub * 16 + (t & 3) * 4) * 2u;
    c.rdB = (uint32_t)row * 256u + (colb ^ swz);
  }

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  using FragsT = Frags;
#endif

  const uint32_t stepA = 64u * lda_b, stepB = 64u * ldb_b;           // one K-tile further down
  {
    const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)(c.lds_base + c.dma_dst));
    dma16(d + 0 * kSlot, c.voffA[0], c.srdA, 0);
    dma16(d + 0 * kSlot + 1024, c.voffA[1], c.srdA, 0);
    dma16(d + 1 * kSlot, c.voffB[0], c.srdB, 0);
    dma16(d + 1 * kSlot + 1024, c.voffB[1], c.srdB, 0);
    dma16(d + 2 * kSlot, c.voffB[0], c.srdB, 256);
    dma16(d + 2 * kSlot + 1024, c.voffB[1], c.srdB, 256);
    dma16(d + 3 * kSlot, c.voffA[0], c.srdA, 256);
    dma16(d + 3 * kSlot + 1024, c.voffA[1], c.srdA, 256);
    dma16(d + 4 * kSlot, c.voffA[0], c.srdA, stepA);
    dma16(d + 4 * kSlot + 1024, c.voffA[1], c.srdA, stepA);
    dma16(d + 5 * kSlot, c.voffB[0], c.srdB, stepB);
    dma16(d + 5 * kSlot + 1024, c.voffB[1], c.srdB, stepB);
    wait_vm<8>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }

  FragsT f;
  const int nk = kt_per_split;          // even, >= 4
  KOff k;
  k.a1 = stepA; k.a2 = 2 * stepA; k.b1 = stepB; k.b2 = 2 * stepB;
  for (int kt = 0; kt < nk - 2; kt += 2) {
    ktile_tn<0, 0>(c, f, acc, k);
    k.a1 += stepA; k.a2 += stepA; k.b1 += stepB; k.b2 += stepB;
    ktile_tn<1, 0>(c, f, acc, k);
    k.a1 += stepA; k.a2 += stepA; k.b1 += stepB; k.b2 += stepB;
  }
  ktile_tn<0, 1>(c, f, acc, k);
  ktile_tn<1, 2>(c, f, acc, k);
  if (wm == 0) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue: fp32 partial tile, row-coalesced through LDS (16 lanes x 16 B = one 256-byte row segment) ----
  char* W = smem + wave * 8192;
  float* out = part + (size_t)split * (size_t)p.N * (size_t)p.K;
  const int r4 = lane >> 4, c16 = lane & 15;
  static_for<4>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    acc_block_to_lds<i>(acc, W, lane);          // (the NT epilogue's image: row r at 256 r, chunk = column / 4 ^ (r & 7))
    const int prow0 = p0 + ((i >> 1) << 7) + wm * 64 + (i & 1) * 32;     // first output row of this block
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int rr = it * 4 + r4;
      const float4 v = *reinterpret_cast<const float4*>(W + rr * 256 + (((uint32_t)c16 ^ (uint32_t)(rr & 7)) << 4));
      // image columns 0..31 are the j = 0 block (q0 + wn*32 ..), 32..63 the j = 1 block (q0 + 128 + wn*32 ..)
      const int col = q0 + ((c16 >> 3) << 7) + wn * 32 + (c16 & 7) * 4;
      *reinterpret_cast<float4*>(out + (size_t)(prow0 + rr) * (size_t)p.K + col) = v;
    }
  });
  (void)l31;
}

// C = (accumulate ? C : 0) + sum_s part[s]   (fixed order)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* part, int splits, int P, int Q, float* C, int64_t ldc,
                                                        int accumulate) {
  const size_t n4 = (size_t)P * Q / 4;
  const size_t stride = (size_t)P * Q;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i * 4;
    const int r = (int)(e / Q), cidx = (int)(e - (size_t)r * Q);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < splits; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * stride + e);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* dst = C + (size_t)r * ldc + cidx;
    if (accumulate) { s.x += dst[0]; s.y += dst[1]; s.z += dst[2]; s.w += dst[3]; }
    dst[0] = s.x; dst[1] = s.y; dst[2] = s.z; dst[3] = s.w;
  }
}

}  // namespace

bool gemm_nt_8p_eligible(const GemmArgs& p, int dtype) {
  if (dtype != EZCLIP_BF16 || p.out_f32 || p.scale_log != nullptr || p.act == ACT_TANH || p.conv_H > 0) return false;
  if (p.act == ACT_RELU_POST && !p.R) return false;
  if ((p.act == ACT_RELU || p.act == ACT_RELU_POST) && p.U) return false;
  if (p.M < 256 || (p.N & 255) || (p.K & 127) || p.K < 256) return false;
  if ((p.lda & 7) || (p.ldb & 7) || (p.ldc & 7) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15) ||
      ((uintptr_t)p.C & 15))
    return false;
  if (p.C2 && ((uintptr_t)p.C2 & 15)) return false;
  if (p.R && ((p.ldr & 7) || ((uintptr_t)p.R & 15))) return false;
  if (p.U && ((p.ldu & 7) || ((uintptr_t)p.U & 15))) return false;
  if (p.bias && ((uintptr_t)p.bias & 15)) return false;
  // instantiated epilogue combinations: plain, +R, +C2, +U, folded LN (alone)
  const int combo = (p.R ? 1 : 0) + (p.U ? 2 : 0) + (p.C2 ? 4 : 0) + (p.ln_stats ? 8 : 0);
  if (!(combo == 0 || combo == 1 || combo == 2 || combo == 4 || combo == 8)) return false;
  if (p.ln_stats && (p.bias || !p.ln_c1 || !p.ln_c2 || ((uintptr_t)p.ln_c1 & 15) || ((uintptr_t)p.ln_c2 & 15) ||
                     ((uintptr_t)p.ln_stats & 7)))
    return false;
  // 32-bit buffer offsets
  const uint64_t lim = 0xffff0000ull;
  if ((uint64_t)p.M * (uint64_t)p.lda * 2u >= lim || (uint64_t)p.N * (uint64_t)p.ldb * 2u >= lim ||
      (uint64_t)p.M * (uint64_t)p.ldc * 2u >= lim)
    return false;
  if (p.R && (uint64_t)p.M * (uint64_t)p.ldr * 2u >= lim) return false;
  if (p.U && (uint64_t)p.M * (uint64_t)p.ldu * 2u >= lim) return false;
  if (p.colsum && (!p.U || p.bias)) return false;     // fused column sums: the act'(U) epilogue, no bias
  if (p.rowstat_part && (combo != 1 || p.act != ACT_NONE || ((uintptr_t)p.rowstat_part & 7) || (uint64_t)p.M * (uint64_t)(p.N >> 6) * 8u >= lim))
    return false;                                     // row-stat partials: the residual epilogue only
  return true;
}

int g_gemm8p_ablate = 0;
void set_gemm8p_ablate(int v) { g_gemm8p_ablate = v; }
#ifndef EZ_RASTER_GM_DEFAULT
#define EZ_RASTER_GM_DEFAULT 106      // super-columns of 6 tile columns (gemm8p_nt.h tile_origin); identical to n-fastest order for N <= 1536
#endif
int g_gemm8p_raster = EZ_RASTER_GM_DEFAULT;
void set_gemm_raster(int gm) { g_gemm8p_raster = gm < 0 ? EZ_RASTER_GM_DEFAULT : gm; }
// De-phasing of the persistent workgroups (round 6; gemm8p_nt.h): workgroup b sleeps ((b >> 3) % period) * steps * 1024 shader clocks before
// its first tile, so that the 32 CUs of an XCD enter their epilogues -- the C stream, with the matrix pipe idle -- at different times.
// value = steps + 100 * period_code (period 32 / 2 / 4 / 8 / 16 for code 0..4); 0 = off; -1 = the built-in default (per shape, below).
#ifndef EZ_DEPHASE_DEFAULT
#define EZ_DEPHASE_DEFAULT 0
#endif
int g_gemm8p_dephase = -1;
void set_gemm_dephase(int v) { g_gemm8p_dephase = v; }
static int dephase_for(const GemmArgs& p, int tiles, int grid) {
  if (tiles <= grid) return 0;                                  // a single round of tiles: nothing to stagger
  if (g_gemm8p_dephase >= 0) return g_gemm8p_dephase;
  // Measured (round 6, profiles/r6_gemm_dephase.md; gemm_bench, two repetitions, same box): stand-alone, the N = K = 768 products with many
  // rounds of tiles gain 4 % from one step per CU of an XCD (vit.out+res 0.265 -> 0.253 ms, patch 0.228 -> 0.219 ms: their epilogue is the
  // largest share of a tile), vit.qkv / vit.proj+res do not move, the three-round BERT products lose 2-3 %.  INSIDE the forward step the
  // gain is gone: staggering every product costs 1.2 % (39.09 -> 39.55 ms), staggering only the N = K = 768 ones reads 39.30 vs 39.41 ms
  // over three interleaved repetitions (the kernels of the two towers overlap on two streams and the staggered start is also a
  // staggered end).  Off by default; ezclip_debug_set(12, v) / EZCLIP_GEMM_DEPHASE switch it on for A/B runs.
  return EZ_DEPHASE_DEFAULT;
}

namespace nt8p {
EZ_8P_INSTANCES_A(EZ_8P_DECLARE)
EZ_8P_INSTANCES_B(EZ_8P_DECLARE)
EZ_8P_INSTANCES_C(EZ_8P_DECLARE)
}  // namespace nt8p
namespace {
int g_num_cus = 0;

// the activation as a template argument where an instantiation exists (nt8p::launch_8p), GemmArgs::act at run time otherwise
template <bool R, bool U, bool C2, bool LN, bool PS>
int launch_8p_act(const GemmArgs& p, int tiles, int grid, hipStream_t stream) {
  using namespace nt8p;
  constexpr bool plain = !R && !U && !C2 && !LN, has_none = plain || R || LN, has_gelu = plain || U || C2, has_qgelu = plain || U || C2 || LN;
  if constexpr (PS) return launch_8p<R, U, C2, LN, PS, ACT_NONE>(p, tiles, grid, stream);    // (row-stat partials: act none only, see eligibility)
  else {
#ifdef EZ_8P_ACT_RUNTIME      // A/B build (tools/build_variants.py): every launch through the run-time-activation instantiation
    return launch_8p<R, U, C2, LN, PS, kActRuntime>(p, tiles, grid, stream);
#endif
    if (has_none && p.act == ACT_NONE) { if constexpr (has_none) return launch_8p<R, U, C2, LN, PS, ACT_NONE>(p, tiles, grid, stream); }
    if (has_qgelu && p.act == ACT_QUICKGELU) { if constexpr (has_qgelu) return launch_8p<R, U, C2, LN, PS, ACT_QUICKGELU>(p, tiles, grid, stream); }
    if (has_gelu && p.act == ACT_GELU_ERF) { if constexpr (has_gelu) return launch_8p<R, U, C2, LN, PS, ACT_GELU_ERF>(p, tiles, grid, stream); }
    return launch_8p<R, U, C2, LN, PS, kActRuntime>(p, tiles, grid, stream);
  }
}
}  // namespace

int gemm_nt_8p(const GemmArgs& p_in, hipStream_t stream) {
  GemmArgs p = p_in;
  p.vec_ok = g_gemm8p_ablate;
  p.raster_gm = (p.N >> 8) > 1 ? g_gemm8p_raster : 0;
  if (g_num_cus == 0) {
    int dev = 0;
    EZ_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    EZ_HIP(hipGetDeviceProperties(&prop, dev));
    g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int tiles = ((p.M + 255) >> 8) * (p.N >> 8);
  int grid = tiles;                                         // one workgroup per CU (160 KiB LDS each)
  if (tiles > g_num_cus) grid = g_num_cus >= 8 ? (g_num_cus & ~7) : g_num_cus;   // multiple of 8: tile -> XCD affinity across rounds
  if (g_gemm8p_ablate == 4) grid = tiles;                   // debugging: one tile per workgroup
  p.raster_gm += 10000 * dephase_for(p, tiles, grid);        // (decoded by the kernel: GemmArgs has no room for another field, see kernels.h)
  int rc;
  {
    ProfScope ps(PROF_GEMM, 2.0 * p.M * (double)p.N * p.K, stream);
    if (p.ln_stats) rc = launch_8p_act<false, false, false, true, false>(p, tiles, grid, stream);
    else if (p.U) rc = launch_8p_act<false, true, false, false, false>(p, tiles, grid, stream);
    else if (p.C2) rc = launch_8p_act<false, false, true, false, false>(p, tiles, grid, stream);
    else if (p.R && p.rowstat_part) rc = launch_8p_act<true, false, false, false, true>(p, tiles, grid, stream);
    else if (p.R) rc = launch_8p_act<true, false, false, false, false>(p, tiles, grid, stream);
    else rc = launch_8p_act<false, false, false, false, false>(p, tiles, grid, stream);
  }
  if (rc != EZ_OK) return rc;
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

// ---- weight-gradient launcher ---------------------------------------------------------------------------
namespace {
// Library-owned scratch for the split partials, ONE PER STREAM (two handles driven on two streams must not share it),
// grown on first use under a mutex; work on a stream is ordered, so reuse within a stream needs no further sync.
struct TnScratch { float* ptr = nullptr; size_t bytes = 0; };
std::mutex g_tn_mutex;
std::map<hipStream_t, TnScratch> g_tn_scratch;
}  // namespace

bool gemm_tn_8p_eligible(const GemmTNArgs& p, int dtype) {
  if (dtype != EZCLIP_BF16) return false;
  if ((p.N & 255) || (p.K & 255) || p.M < 2048) return false;
  if ((p.lda & 7) || (p.ldb & 7) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15) || ((uintptr_t)p.C & 15) || (p.ldc & 3))
    return false;
  const uint64_t lim = 0xffff0000ull;
  if (((uint64_t)p.M + 65536) * (uint64_t)p.lda * 2u >= lim || ((uint64_t)p.M + 65536) * (uint64_t)p.ldb * 2u >= lim) return false;
  return true;
}

int gemm_tn_8p(const GemmTNArgs& p, hipStream_t stream) {
  if (g_num_cus == 0) {
    int dev = 0;
    EZ_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    EZ_HIP(hipGetDeviceProperties(&prop, dev));
    g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int tiles_p = p.N >> 8, tiles_q = p.K >> 8, ntiles = tiles_p * tiles_q;
  const int kt_total = (p.M + 63) / 64;
  int splits = g_num_cus / ntiles;
  if (splits < 1) splits = 1;
  int kt = (kt_total + splits - 1) / splits;
  if (kt < 4) kt = 4;
  kt = (kt + 1) & ~1;                                  // even, >= 4
  splits = (kt_total + kt - 1) / kt;                   // drop splits that would start past M
  const size_t need = (size_t)splits * (size_t)p.N * (size_t)p.K * sizeof(float);
  float* scratch = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_tn_mutex);
    TnScratch& sc = g_tn_scratch[stream];
    if (need > sc.bytes) {
      if (sc.ptr) { EZ_HIP(hipStreamSynchronize(stream)); EZ_HIP(hipFree(sc.ptr)); sc.ptr = nullptr; sc.bytes = 0; }
      EZ_HIP(hipMalloc(reinterpret_cast<void**>(&sc.ptr), need));
      sc.bytes = need;
    }
    scratch = sc.ptr;
  }
  static LdsOptIn lds_opt;
  EZ_ENSURE_LDS(&gemm_tn_8p_kernel, lds_opt, kRing);
  {
    ProfScope ps(PROF_GEMM, 2.0 * p.M * (double)p.N * p.K, stream);
    hipLaunchKernelGGL(gemm_tn_8p_kernel, dim3(ntiles * splits), dim3(kThreads8), kRing, stream, p, scratch, ntiles,
                       tiles_q, kt);
    const size_t n4 = (size_t)p.N * p.K / 4;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(blocks), dim3(256), 0, stream, scratch, splits, p.N, p.K, p.C, p.ldc,
                       p.accumulate);
  }
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace ezclip
