// bf16 MFMA GEMM, 256 x 128 x 64 tile, FOUR waves per workgroup, TWO workgroups per CU (gfx950 / CDNA4 only).
//
//   C[M,N] = epilogue( A[M,K] . B[N,K]^T )        A, B, C bf16; fp32 accumulate
//
// Same wave tile (128 x 64 = 4 x 2 MFMA 32x32 accumulators), same four quadrant phases per K-tile, same swizzled
// lane-linear LDS images, same hand-issued LDS-DMA with counted waits and the same row-coalesced epilogue as the
// 8-wave kernel of gemm8p.hip.  What differs is WHERE the overlap comes from: the 8-wave kernel owns the CU and
// pairs its two wave rows on each SIMD, which leaves the matrix pipe idle while all eight waves run the epilogue
// (128 KiB of C per 100 MFLOP tile when K = 768: a third of the tile time).  Here a workgroup is one wave per SIMD
// and 80 KiB of LDS, so two independent workgroups share a CU: while one is in its epilogue, prologue or read/DMA
// segment the other feeds the matrix pipe.  The price is 1.5x the LDS-DMA bytes per flop (a 256 x 128 tile) --
// worth it for the short-K products, not for K = 3072 (the dispatcher in gemm.hip chooses).
//
// LDS ring: 10 granules of 8 KiB (64 rows x 128 B).  A K-tile is six granules in need order:
//   A-lo piece 0, A-lo piece 1 (the two wave rows' first 64 M rows), B-lo, B-hi (2 wave columns x 32 N rows),
//   A-hi piece 0, A-hi piece 1;   granule n = 6*tile + j lives in slot n % 10.
// Phase schedule of K-tile t (one barrier per phase; r = the ring position of A-lo piece 0):
//   P0  read A-lo(piece wm), B-lo      issue B-lo(t+1)        wait until B-hi(t) has landed
//   P1  read B-hi                      issue B-hi(t+1)        wait until A-hi(t) has landed
//   P2  read A-hi(piece wm)            issue A-hi(t+1) x2     --
//   P3  --                             issue A-lo(t+2) x2     wait until A-lo(t+1), B-lo(t+1) have landed
// Every granule is overwritten at least two phases after its last read and waited for one phase before its first
// read; 8 to 12 loads per wave stay in flight across the barriers.
#include "gemm_pipe.h"

namespace ezclip {

namespace {

constexpr int kGran = 8192;
constexpr int kRing4 = 10 * kGran;      // 80 KiB: two workgroups per CU
constexpr int kThreads4 = 256;

struct Ctx4 {
  const char* smem;
  uint32_t lds_base;
  i32x4_t srdA, srdB;
  uint32_t voffA[2], voffB[2];
  uint32_t pieceA, hiA, hiB;   // byte offsets: 128 rows of A, 64 rows of A, 32 rows of B
  uint32_t dma_dst;            // wave * 2048
  uint32_t rdA[4], rdB[4];     // per-lane LDS byte offsets of the 4 k-step chunks (swizzled), row included
};

__device__ __forceinline__ int ring(int r, int j) {   // (r + j) % 10 for r < 10, j < 20
  int x = r + j;
  if (x >= 10) x -= 10;
  if (x >= 10) x -= 10;
  return x;
}

__device__ __forceinline__ void dma_gran(const Ctx4& c, int slot, const uint32_t (&voff)[2], const i32x4_t& srd, uint32_t soff) {
  const uint32_t dst = c.lds_base + (uint32_t)slot * kGran + c.dma_dst;
  dma16(dst, voff[0], srd, soff);
  dma16(dst + 1024, voff[1], srd, soff);
}

template <int A, int B>
__device__ __forceinline__ void wait_sel(bool first) {
  if (first) wait_vm<A>(); else wait_vm<B>();
}

template <bool FAST, bool HAS_R, bool HAS_U, bool HAS_C2>
__global__ __launch_bounds__(kThreads4, 2) void gemm_nt_4w_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int tiles_n = p.N >> 7;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = t / tiles_n;
  const int m0 = tm << 8, n0 = (t - tm * tiles_n) << 7;

  Ctx4 c;
  c.smem = smem;
  c.lds_base = (uint32_t)(size_t)smem;
  const uint32_t lda_b = (uint32_t)p.lda * 2u, ldb_b = (uint32_t)p.ldb * 2u;
  c.srdA = make_srd(p.A, (uint32_t)(p.M - 1) * lda_b + (uint32_t)p.K * 2u);
  c.srdB = make_srd(p.B, (uint32_t)(p.N - 1) * ldb_b + (uint32_t)p.K * 2u);
  c.pieceA = 128u * lda_b;
  c.hiA = 64u * lda_b;
  c.hiB = 32u * ldb_b;
  c.dma_dst = wave * 2048;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lr = (wave * 2 + i) * 8 + (lane >> 3);             // row of the granule image (0..63)
    const uint32_t ch = (uint32_t)((lane & 7) ^ ((lr >> 1) & 7)) << 4;
    c.voffA[i] = (uint32_t)(m0 + lr) * lda_b + ch;
    c.voffB[i] = (uint32_t)(n0 + (lr >> 5) * 64 + (lr & 31)) * ldb_b + ch;
  }
  {
    const int sw = (l31 >> 1) & 7;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint32_t ch = (uint32_t)((2 * s + h) ^ sw) << 4;
      c.rdA[s] = (uint32_t)l31 * 128u + ch;
      c.rdB[s] = (uint32_t)(wn * 32 + l31) * 128u + ch;
    }
  }

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K >> 6;               // >= 2 (checked by the launcher)
  // ---- prologue: A-lo(0) x2, B-lo(0), B-hi(0), A-hi(0) x2, A-lo(1) x2  = granules 0..7 -------------------------
  dma_gran(c, 0, c.voffA, c.srdA, 0);
  dma_gran(c, 1, c.voffA, c.srdA, c.pieceA);
  dma_gran(c, 2, c.voffB, c.srdB, 0);
  dma_gran(c, 3, c.voffB, c.srdB, c.hiB);
  dma_gran(c, 4, c.voffA, c.srdA, c.hiA);
  dma_gran(c, 5, c.voffA, c.srdA, c.hiA + c.pieceA);
  dma_gran(c, 6, c.voffA, c.srdA, 128);
  dma_gran(c, 7, c.voffA, c.srdA, 128 + c.pieceA);
  wait_vm<10>();                         // A-lo(0), B-lo(0) (this wave's pieces)
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  Frags f;
  int r = 0;                             // ring position of A-lo piece 0 of the current K-tile
  uint32_t kb = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool has1 = kt + 1 < nk, has2 = kt + 2 < nk;
    // ---- P0: (A-lo, B-lo) -------------------------------------------------------------------------------------
    {
      const char* pa = smem + ring(r, wm) * kGran;
      const char* pb = smem + ring(r, 2) * kGran;
#pragma unroll
      for (int s = 0; s < 4; ++s) f.bl[s] = *reinterpret_cast<const uint4*>(pb + c.rdB[s]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        f.a[0][s] = *reinterpret_cast<const uint4*>(pa + c.rdA[s]);
        f.a[1][s] = *reinterpret_cast<const uint4*>(pa + 4096 + c.rdA[s]);
      }
      if (has1) dma_gran(c, ring(r, 8), c.voffB, c.srdB, kb + 128);                 // B-lo(t+1)
      wait_sel<10, 4>(has1);                                                      // B-hi(t) has landed
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        mma32(acc[0][0], f.bl[s], f.a[0][s], bf16_t());
        mma32(acc[1][0], f.bl[s], f.a[1][s], bf16_t());
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- P1: (A-lo, B-hi) -------------------------------------------------------------------------------------
    {
      const char* pb = smem + ring(r, 3) * kGran;
#pragma unroll
      for (int s = 0; s < 4; ++s) f.bh[s] = *reinterpret_cast<const uint4*>(pb + c.rdB[s]);
      if (has1) dma_gran(c, ring(r, 9), c.voffB, c.srdB, kb + 128 + c.hiB);         // B-hi(t+1)
      wait_sel<8, 0>(has1);                                                       // A-hi(t) has landed
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        mma32(acc[0][1], f.bh[s], f.a[0][s], bf16_t());
        mma32(acc[1][1], f.bh[s], f.a[1][s], bf16_t());
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- P2: (A-hi, B-hi) -------------------------------------------------------------------------------------
    {
      const char* pa = smem + ring(r, 4 + wm) * kGran;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        f.a[0][s] = *reinterpret_cast<const uint4*>(pa + c.rdA[s]);
        f.a[1][s] = *reinterpret_cast<const uint4*>(pa + 4096 + c.rdA[s]);
      }
      if (has1) {                                                                  // A-hi(t+1), both pieces
        dma_gran(c, ring(r, 10), c.voffA, c.srdA, kb + 128 + c.hiA);
        dma_gran(c, ring(r, 11), c.voffA, c.srdA, kb + 128 + c.hiA + c.pieceA);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        mma32(acc[2][1], f.bh[s], f.a[0][s], bf16_t());
        mma32(acc[3][1], f.bh[s], f.a[1][s], bf16_t());
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- P3: (A-hi, B-lo) -------------------------------------------------------------------------------------
    {
      if (has2) {                                                                  // A-lo(t+2), both pieces
        dma_gran(c, ring(r, 12), c.voffA, c.srdA, kb + 256);
        dma_gran(c, ring(r, 13), c.voffA, c.srdA, kb + 256 + c.pieceA);
      }
      if (has1) wait_sel<10, 6>(has2);                                             // A-lo(t+1), B-lo(t+1) have landed
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        mma32(acc[2][0], f.bl[s], f.a[0][s], bf16_t());
        mma32(acc[3][0], f.bl[s], f.a[1][s], bf16_t());
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    r = ring(r, 6);
    kb += 128;
  }
  __builtin_amdgcn_s_barrier();          // every wave is past its last LDS read; no DMA is in flight
  __builtin_amdgcn_sched_barrier(0);

  const EpiCtx ep = make_epi_ctx<HAS_R, HAS_U, HAS_C2>(p);
  EpiLoads eld;
  if constexpr (HAS_R || HAS_U) epilogue_issue_block<HAS_R, HAS_U, false, 0, 0>(ep, m0 + wm * 128, n0 + wn * 64, eld);
  epilogue_rows<FAST, HAS_R, HAS_U, HAS_C2, 0>(ep, acc, m0 + wm * 128, n0 + wn * 64, smem + wave * 8192, p.act, eld, []() {});
}

int g_num_cus4 = 0;

template <bool R, bool U, bool C2>
int launch_4w(const GemmArgs& p, int tiles, hipStream_t stream) {
  static bool attr_set = false;
  auto* kern = &gemm_nt_4w_kernel<true, R, U, C2>;
  if (!attr_set) {
    EZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kRing4));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(kThreads4), kRing4, stream, p);
  return EZ_OK;
}

}  // namespace

bool gemm_nt_4w_eligible(const GemmArgs& p, int dtype) {
  if (!gemm_nt_8p_eligible(p, dtype)) return false;   // same operand / epilogue constraints (N % 256 included)
  return p.K >= 128 && p.ln_stats == nullptr && p.rowstat_part == nullptr && p.act <= ACT_GELU_ERF;
}

int gemm_nt_4w(const GemmArgs& p, hipStream_t stream) {
  const int tiles = ((p.M + 255) >> 8) * (p.N >> 7);
  int rc;
  {
    ProfScope ps(PROF_GEMM, 2.0 * p.M * (double)p.N * p.K, stream);
    if (p.U) rc = launch_4w<false, true, false>(p, tiles, stream);
    else if (p.C2) rc = launch_4w<false, false, true>(p, tiles, stream);
    else if (p.R) rc = launch_4w<true, false, false>(p, tiles, stream);
    else rc = launch_4w<false, false, false>(p, tiles, stream);
  }
  if (rc != EZ_OK) return rc;
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace ezclip
