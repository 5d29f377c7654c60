// MFMA GEMM for gfx950:  C[M,N] = epilogue( A[M,K] . B[N,K]^T )
//
// Both operands are K-contiguous ("NT"), which is how every forward product of
// the path is laid out: activations [tokens, features] times nn.Linear weights
// [out, in] (reference: y = x W^T + b -- bert/modeling_bert.py:145-147,260,
// 324,338; nn.MultiheadAttention in_proj/out_proj and mlp.c_fc/c_proj,
// modeling_chineseclip.py:188-195), and the cross-modal similarity T . I^T
// (appzoo/clip/model.py:148).
//
// Geometry (same bytes for both dtypes; BK = 128 B / sizeof(T)):
//   workgroup 256 threads = 4 waves (2 x 2), tile 128 x 128 x BK
//   wave tile 64 x 64 = 2 x 2 MFMA 32x32 accumulators (64 fp32 regs)
//   LDS: double-buffered A and B tiles, 128 rows x 128 B each, filled by
//        global_load_lds_dwordx4 (no VGPR round trip).  The LDS image is
//        lane-linear, so the bank swizzle chunk ^= (row>>1)&7 is applied to the
//        per-lane *source* address and again on the ds_read_b128 fragment reads
//        (conflict-free for the 32-row fragment groups).
//   MFMA is issued with the operands swapped -- mfma(Bfrag, Afrag) -- so each
//   lane ends up with 4 consecutive N for one M row: vector stores along N,
//   vector bias / residual loads.
//   bf16: v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//   f32 : v_mfma_f32_32x32x2_f32 (exact f32 fmaf chain).
//   1-D grid with an XCD-aware remap: consecutive tiles (same A panel) share an L2.
#include "ezclip_common.h"
#include "kernels.h"

namespace ezclip {

namespace {

constexpr int BM = 128, BN = 128;
constexpr int kTileBytes = 128 * 128;  // one operand tile
constexpr int kThreads = 256;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// Stage a (<=128)-row x 128-byte tile into LDS with LDS-DMA.  16 wave-instructions
// of 1 KiB (8 rows); wave w issues instructions 4w..4w+3.
__device__ __forceinline__ void stage_tile(const char* __restrict__ g, int64_t ld_bytes, int row0,
                                           int row_max, int64_t kbyte, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int inst = wave * 4 + i;
    const int r = inst * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int gr = row0 + r;
    gr = gr < row_max ? gr : row_max - 1;
    const char* src = g + (int64_t)gr * ld_bytes + kbyte + c * 16;
    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(lds_tile + inst * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ uint4 read_frag(const char* lds_tile, int row, int chunk) {
  return *reinterpret_cast<const uint4*>(lds_tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

template <typename T, typename TO>
__global__ __launch_bounds__(kThreads, 2) void gemm_nt_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x (A 16K + B 16K)
  constexpr int BK = 128 / (int)sizeof(T);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  const int t = xcd_remap(blockIdx.x, nwg);
  const int m0 = (t / tiles_n) * BM;
  const int n0 = (t % tiles_n) * BN;
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;

  const char* gA = reinterpret_cast<const char*>(p.A);
  const char* gB = reinterpret_cast<const char*>(p.B);
  const int64_t lda_b = p.lda * (int64_t)sizeof(T), ldb_b = p.ldb * (int64_t)sizeof(T);

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  stage_tile(gA, lda_b, m0, p.M, 0, smem, wave, lane);
  stage_tile(gB, ldb_b, n0, p.N, 0, smem + kTileBytes, wave, lane);

  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * 2 * kTileBytes;
    char* nxt = smem + ((kt + 1) & 1) * 2 * kTileBytes;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt landed everywhere; everyone is done reading `nxt`
    if (kt + 1 < nk) {
      const int64_t kb = (int64_t)(kt + 1) * 128;
      stage_tile(gA, lda_b, m0, p.M, kb, nxt, wave, lane);
      stage_tile(gB, ldb_b, n0, p.N, kb, nxt + kTileBytes, wave, lane);
    }
    const char* tA = cur;
    const char* tB = cur + kTileBytes;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = 2 * s + h;
      uint4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = read_frag(tA, wm * 64 + i * 32 + l31, c);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = read_frag(tB, wn * 64 + j * 32 + l31, c);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma32(acc[i][j], b[j], a[i], T());
    }
  }

  // ---- epilogue: lane owns row m, 4 consecutive n per (j, q) ----
  float scale = p.alpha;
  if (p.scale_log != nullptr) scale *= expf(*p.scale_log);
  TO* C = reinterpret_cast<TO*>(p.C);
  TO* C2 = reinterpret_cast<TO*>(p.C2);
  const T* R = reinterpret_cast<const T*>(p.R);
  constexpr bool kFast = IsFast<TO>::value;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + l31;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + j * 32 + q * 8 + h * 4;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e] * scale;
        if (p.vec_ok) {
          if (p.bias) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
          }
          if (C2) st4(C2 + (int64_t)m * p.ldc + n, v);
          if (p.act != ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_apply<kFast>(v[e], p.act);
          }
          if (R) {
            float rv[4];
            ld4(R + (int64_t)m * p.ldr + n, rv);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rv[e];
          }
          st4(C + (int64_t)m * p.ldc + n, v);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (n + e >= p.N) break;
            float x = v[e];
            if (p.bias) x += p.bias[n + e];
            if (C2) Elem<TO>::st(C2 + (int64_t)m * p.ldc + n + e, x);
            x = act_apply<kFast>(x, p.act);
            if (R) x += Elem<T>::ld(R + (int64_t)m * p.ldr + n + e);
            Elem<TO>::st(C + (int64_t)m * p.ldc + n + e, x);
          }
        }
      }
    }
  }
}

template <typename T, typename TO>
int launch_nt(const GemmArgs& p, hipStream_t stream) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const size_t lds = 4 * kTileBytes;
  static bool attr_set = false;
  if (!attr_set) {
    EZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<T, TO>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  {
    ProfScope ps(PROF_GEMM, 2.0 * p.M * (double)p.N * p.K, stream);
    hipLaunchKernelGGL((gemm_nt_kernel<T, TO>), dim3(tiles), dim3(kThreads), lds, stream, p);
  }
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace

int gemm_nt(GemmArgs p, int dtype, hipStream_t stream) {
  EZ_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm_nt: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  const int esz = dtype == EZCLIP_BF16 ? 2 : 4;
  EZ_REQUIRE((p.K * esz) % 128 == 0, "gemm_nt: K=%d must be a multiple of %d", p.K, 128 / esz);
  EZ_REQUIRE((p.lda * esz) % 16 == 0 && (p.ldb * esz) % 16 == 0, "gemm_nt: lda/ldb must be 16-byte multiples");
  EZ_REQUIRE(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0, "gemm_nt: A/B must be 16-byte aligned");
  const int osz = (dtype == EZCLIP_BF16 && !p.out_f32) ? 2 : 4;
  bool vec = (p.N % 4 == 0) && (p.ldc % 4 == 0) && ((uintptr_t)p.C % 16 == 0) &&
             (p.C2 == nullptr || (uintptr_t)p.C2 % 16 == 0) &&
             (p.bias == nullptr || (uintptr_t)p.bias % 16 == 0) &&
             (p.R == nullptr || (p.ldr % 4 == 0 && (uintptr_t)p.R % 16 == 0));
  (void)osz;
  p.vec_ok = vec ? 1 : 0;
  if (dtype == EZCLIP_F32) return launch_nt<float, float>(p, stream);
  if (dtype == EZCLIP_BF16) {
    if (p.out_f32) return launch_nt<bf16_t, float>(p, stream);
    return launch_nt<bf16_t, bf16_t>(p, stream);
  }
  set_error("gemm_nt: bad dtype %d", dtype);
  return EZ_ERR_INVALID;
}

int gemm_tn(GemmTNArgs, int, hipStream_t) {
  set_error("gemm_tn: not implemented yet");
  return EZ_ERR_UNSUPPORTED;
}

}  // namespace ezclip
