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
//   * Epilogue: the accumulators go through a wave-private 16 KiB LDS region (the ring is dead by then)
//     so that every global access of C / residual / pre-activation is a 16-byte-per-lane access covering
//     8 full 128-byte rows per wave instruction.
#include "ezclip_common.h"
#include "kernels.h"

namespace ezclip {

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4_t;

constexpr int kSlot = 16384;           // one half-tile: 128 rows x 128 B
constexpr int kRing = 8 * kSlot;       // 128 KiB
constexpr int kThreads8 = 512;

// LDS-DMA: 64 lanes x 16 B land at lds_dst + lane*16 (wave-uniform destination).
__device__ __forceinline__ void dma16(uint32_t lds_dst, uint32_t voff, const i32x4_t& srd, uint32_t soff) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_dst), "v"(voff), "s"(srd), "s"(soff)
      : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  if constexpr (N >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
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
template <int P, int PAR, bool ISSUE, int VM>
__device__ __forceinline__ void phase(const Ctx& c, Frags& f, f32x16_t (&acc)[4][2], uint32_t kbyte_next1,
                                      uint32_t kbyte_next2) {
  constexpr int k8 = 4 * PAR + P;                 // phase number mod 8
  // ---- read segment -----------------------------------------------------------------------
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
  // ---- DMA for half-tile k+6 (kind (P+2)&3: P0 -> B-hi(t+1), P1 -> A-hi(t+1), P2 -> A-lo(t+2), P3 -> B-lo(t+2))
  if constexpr (ISSUE) {
    constexpr int slot = ((k8 + 6) & 7) * kSlot;
    const uint32_t dst = c.lds_base + slot + c.dma_dst;
    if constexpr (P == 0) {
      dma16(dst, c.voffB[0], c.srdB, kbyte_next1 + c.hiB);
      dma16(dst + 1024, c.voffB[1], c.srdB, kbyte_next1 + c.hiB);
    } else if constexpr (P == 1) {
      dma16(dst, c.voffA[0], c.srdA, kbyte_next1 + c.hiA);
      dma16(dst + 1024, c.voffA[1], c.srdA, kbyte_next1 + c.hiA);
    } else if constexpr (P == 2) {
      dma16(dst, c.voffA[0], c.srdA, kbyte_next2);
      dma16(dst + 1024, c.voffA[1], c.srdA, kbyte_next2);
    } else {
      dma16(dst, c.voffB[0], c.srdB, kbyte_next2);
      dma16(dst + 1024, c.voffB[1], c.srdB, kbyte_next2);
    }
  }
  wait_vm<VM>();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  // ---- MFMA segment -------------------------------------------------------------------------
  __builtin_amdgcn_s_setprio(1);
  constexpr int i0 = (P >= 2) ? 2 : 0;
  constexpr int j = (P == 1 || P == 2) ? 1 : 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint4& b = (j == 0) ? f.bl[s] : f.bh[s];
    mma32(acc[i0][j], b, f.a[0][s], bf16_t());
    mma32(acc[i0 + 1][j], b, f.a[1][s], bf16_t());
  }
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// One K-tile = 4 phases.  TAIL: 0 = steady state (all issue, vmcnt 8);
// 1 = second-to-last tile (P0,P1 issue; then 6, 4); 2 = last tile (2, 0, -, -).
template <int PAR, int TAIL>
__device__ __forceinline__ void ktile(const Ctx& c, Frags& f, f32x16_t (&acc)[4][2], uint32_t kb1, uint32_t kb2) {
  if constexpr (TAIL == 0) {
    phase<0, PAR, true, 8>(c, f, acc, kb1, kb2);
    phase<1, PAR, true, 8>(c, f, acc, kb1, kb2);
    phase<2, PAR, true, 8>(c, f, acc, kb1, kb2);
    phase<3, PAR, true, 8>(c, f, acc, kb1, kb2);
  } else if constexpr (TAIL == 1) {
    phase<0, PAR, true, 8>(c, f, acc, kb1, kb2);
    phase<1, PAR, true, 8>(c, f, acc, kb1, kb2);
    phase<2, PAR, false, 6>(c, f, acc, kb1, kb2);
    phase<3, PAR, false, 4>(c, f, acc, kb1, kb2);
  } else {
    phase<0, PAR, false, 2>(c, f, acc, kb1, kb2);
    phase<1, PAR, false, 0>(c, f, acc, kb1, kb2);
    phase<2, PAR, false, -1>(c, f, acc, kb1, kb2);
    phase<3, PAR, false, -1>(c, f, acc, kb1, kb2);
  }
}

// wave-private epilogue staging image: [128 rows][128 B], 16-byte chunk index XORed with row & 7
__device__ __forceinline__ uint32_t stage_off(int row, int c16) { return row * 128 + ((c16 ^ (row & 7)) << 4); }

template <bool FAST>
__global__ __launch_bounds__(kThreads8, 2) void gemm_nt_8p_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int h = lane >> 5, l31 = lane & 31;
  const int tiles_n = p.N >> 8;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (t / tiles_n) << 8;
  const int n0 = (t % tiles_n) << 8;

  Ctx c;
  c.smem = smem;
  c.lds_base = (uint32_t)(size_t)smem;
  const uint32_t lda_b = (uint32_t)p.lda * 2u, ldb_b = (uint32_t)p.ldb * 2u;
  c.srdA = make_srd(p.A, (uint32_t)(p.M - 1) * lda_b + (uint32_t)p.K * 2u);
  c.srdB = make_srd(p.B, (uint32_t)(p.N - 1) * ldb_b + (uint32_t)p.K * 2u);
  c.hiA = 64u * lda_b;
  c.hiB = 32u * ldb_b;
  c.dma_dst = wave * 2048;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lr = (wave * 2 + i) * 8 + (lane >> 3);             // row of the half-tile image
    const uint32_t ch = (uint32_t)((lane & 7) ^ ((lr >> 1) & 7)) << 4;
    const uint32_t ra = (uint32_t)(m0 + (lr >> 6) * 128 + (lr & 63));
    const uint32_t rb = (uint32_t)(n0 + (lr >> 5) * 64 + (lr & 31));
    c.voffA[i] = ra * lda_b + ch;
    c.voffB[i] = rb * ldb_b + ch;
  }
  {
    const int sw = (l31 >> 1) & 7;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint32_t ch = (uint32_t)((2 * s + h) ^ sw) << 4;
      c.rdA[s] = (uint32_t)(wm * 64 + l31) * 128 + ch;
      c.rdB[s] = (uint32_t)(wn * 32 + l31) * 128 + ch;
    }
  }

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: half-tiles 0..5 (tile 0 complete, A-lo / B-lo of tile 1) ---------------------
  {
    const uint32_t d = c.lds_base + c.dma_dst;
    dma16(d + 0 * kSlot, c.voffA[0], c.srdA, 0);
    dma16(d + 0 * kSlot + 1024, c.voffA[1], c.srdA, 0);
    dma16(d + 1 * kSlot, c.voffB[0], c.srdB, 0);
    dma16(d + 1 * kSlot + 1024, c.voffB[1], c.srdB, 0);
    dma16(d + 2 * kSlot, c.voffB[0], c.srdB, c.hiB);
    dma16(d + 2 * kSlot + 1024, c.voffB[1], c.srdB, c.hiB);
    dma16(d + 3 * kSlot, c.voffA[0], c.srdA, c.hiA);
    dma16(d + 3 * kSlot + 1024, c.voffA[1], c.srdA, c.hiA);
    dma16(d + 4 * kSlot, c.voffA[0], c.srdA, 128);
    dma16(d + 4 * kSlot + 1024, c.voffA[1], c.srdA, 128);
    dma16(d + 5 * kSlot, c.voffB[0], c.srdB, 128);
    dma16(d + 5 * kSlot + 1024, c.voffB[1], c.srdB, 128);
    wait_vm<8>();                       // half-tiles 0 and 1 (this wave's pieces)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();       // ... and everyone else's
    if (wm == 1) __builtin_amdgcn_s_barrier();   // wave row 1 runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);
  }

  Frags f;
  const int nk = p.K >> 6;              // even, >= 4 (checked by the launcher)
  uint32_t kb = 0;                      // byte offset of the current tile's k range
  for (int kt = 0; kt < nk - 2; kt += 2) {
    ktile<0, 0>(c, f, acc, kb + 128, kb + 256);
    ktile<1, 0>(c, f, acc, kb + 256, kb + 384);
    kb += 256;
  }
  ktile<0, 1>(c, f, acc, kb + 128, kb + 256);
  ktile<1, 2>(c, f, acc, 0, 0);
  if (wm == 0) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  // every wave is past its last LDS read and no DMA is in flight: the ring is free

  // ---- epilogue -----------------------------------------------------------------------------
  char* W = smem + wave * kSlot;                       // wave-private [128][128 B]
  const int mw = m0 + wm * 128, nw = n0 + wn * 64;     // wave tile origin
  const float scale = p.alpha;
  bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
  bf16_t* C2 = reinterpret_cast<bf16_t*>(p.C2);
  const bf16_t* R = reinterpret_cast<const bf16_t*>(p.R);
  const bf16_t* U = reinterpret_cast<const bf16_t*>(p.U);
  const int crow = lane >> 3, cc = lane & 7;           // coalesced view: 8 rows x 8 chunks per instruction

  // global [128 x 64] bf16 block (row stride ld) -> staging image
  auto stage_in = [&](const bf16_t* src, int64_t ld) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      uint4 v[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = (g * 8 + it) * 8 + crow;
        v[it] = (mw + row < p.M) ? *reinterpret_cast<const uint4*>(src + (int64_t)(mw + row) * ld + nw + cc * 8)
                                 : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = (g * 8 + it) * 8 + crow;
        *reinterpret_cast<uint4*>(W + stage_off(row, cc)) = v[it];
      }
    }
  };
  // staging image -> global
  auto stage_out = [&](bf16_t* dst, int64_t ld) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      uint4 v[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = (g * 8 + it) * 8 + crow;
        v[it] = *reinterpret_cast<const uint4*>(W + stage_off(row, cc));
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = (g * 8 + it) * 8 + crow;
        if (mw + row < p.M) *reinterpret_cast<uint4*>(dst + (int64_t)(mw + row) * ld + nw + cc * 8) = v[it];
      }
    }
  };
  // MFMA-layout view of the image: lane (row l31 of block i) holds n = j*32 + q*8 + h*4 + e
  auto frag_ptr = [&](int i, int j, int q) -> char* {
    const int row = i * 32 + l31;
    return W + stage_off(row, j * 4 + q) + h * 8;
  };

  // 1. acc = alpha * acc + bias
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + nw + j * 32 + q * 8 + h * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i][j][q * 4 + 0] = acc[i][j][q * 4 + 0] * scale + bv.x;
        acc[i][j][q * 4 + 1] = acc[i][j][q * 4 + 1] * scale + bv.y;
        acc[i][j][q * 4 + 2] = acc[i][j][q * 4 + 2] * scale + bv.z;
        acc[i][j][q * 4 + 3] = acc[i][j][q * 4 + 3] * scale + bv.w;
      }
    }
  // 2. optional second output: the pre-activation value
  if (C2) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<uint2*>(frag_ptr(i, j, q)) =
              make_uint2(pack_bf16x2(acc[i][j][q * 4], acc[i][j][q * 4 + 1]),
                         pack_bf16x2(acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]));
    stage_out(C2, p.ldc);
  }
  // 3. activation, or (backward) multiply by act'(U)
  if (U) {
    stage_in(U, p.ldu);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint2 u = *reinterpret_cast<const uint2*>(frag_ptr(i, j, q));
          acc[i][j][q * 4 + 0] *= act_grad<FAST>(__uint_as_float(u.x << 16), p.act);
          acc[i][j][q * 4 + 1] *= act_grad<FAST>(__uint_as_float(u.x & 0xffff0000u), p.act);
          acc[i][j][q * 4 + 2] *= act_grad<FAST>(__uint_as_float(u.y << 16), p.act);
          acc[i][j][q * 4 + 3] *= act_grad<FAST>(__uint_as_float(u.y & 0xffff0000u), p.act);
        }
  } else if (p.act != ACT_NONE) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = act_apply<FAST>(acc[i][j][r], p.act);
  }
  // 4. residual, round once, store
  if (R) stage_in(R, p.ldr);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        char* ptr = frag_ptr(i, j, q);
        float v0 = acc[i][j][q * 4], v1 = acc[i][j][q * 4 + 1], v2 = acc[i][j][q * 4 + 2], v3 = acc[i][j][q * 4 + 3];
        if (R) {
          const uint2 r = *reinterpret_cast<const uint2*>(ptr);
          v0 += __uint_as_float(r.x << 16);
          v1 += __uint_as_float(r.x & 0xffff0000u);
          v2 += __uint_as_float(r.y << 16);
          v3 += __uint_as_float(r.y & 0xffff0000u);
        }
        *reinterpret_cast<uint2*>(ptr) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      }
  stage_out(C, p.ldc);
}

}  // namespace

bool gemm_nt_8p_eligible(const GemmArgs& p, int dtype) {
  if (dtype != EZCLIP_BF16 || p.out_f32 || p.scale_log != nullptr) return false;
  if (p.M < 256 || (p.N & 255) || (p.K & 127) || p.K < 256) return false;
  if ((p.lda & 7) || (p.ldb & 7) || (p.ldc & 7) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15) ||
      ((uintptr_t)p.C & 15))
    return false;
  if (p.C2 && ((uintptr_t)p.C2 & 15)) return false;
  if (p.R && ((p.ldr & 7) || ((uintptr_t)p.R & 15))) return false;
  if (p.U && ((p.ldu & 7) || ((uintptr_t)p.U & 15))) return false;
  if (p.bias && ((uintptr_t)p.bias & 15)) return false;
  // 32-bit buffer offsets
  if ((uint64_t)p.M * (uint64_t)p.lda * 2u >= 0xffff0000ull || (uint64_t)p.N * (uint64_t)p.ldb * 2u >= 0xffff0000ull)
    return false;
  return true;
}

int gemm_nt_8p(const GemmArgs& p, hipStream_t stream) {
  const int tiles = ((p.M + 255) >> 8) * (p.N >> 8);
  static bool attr_set = false;
  if (!attr_set) {
    EZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_8p_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kRing));
    attr_set = true;
  }
  {
    ProfScope ps(PROF_GEMM, 2.0 * p.M * (double)p.N * p.K, stream);
    hipLaunchKernelGGL((gemm_nt_8p_kernel<true>), dim3(tiles), dim3(kThreads8), kRing, stream, p);
  }
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace ezclip
