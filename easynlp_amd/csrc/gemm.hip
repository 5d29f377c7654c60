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
#include <type_traits>

#include "ezclip_common.h"
#include "kernels.h"

namespace ezclip {

namespace {

constexpr int BM = 128, BN = 128;
constexpr int kTileBytes = 128 * 128;  // one operand tile
constexpr int kThreads = 256;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ uint4 read_frag(const char* lds_tile, int row, int chunk) {
  return *reinterpret_cast<const uint4*>(lds_tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// Tile-shape template: WM x WN waves, each owning TI x TJ MFMA 32x32 accumulators.
//   <2,2,2,2>: 128x128 tile, 4 waves, 64 KB LDS, 2 workgroups / CU   (small / ragged problems)
//   <2,4,4,2>: 256x256 tile, 8 waves of 128x64, 128 KB LDS, 1 workgroup / CU: 0.75 ds_read_b128 per
//              MFMA instead of 1.0 and half the LDS-DMA bytes per MFMA -- the 128x128 shape is LDS-bound
//              at ~50% MFMA utilisation (4 LDS cycles per read, 8 CU cycles per MFMA).
// Fragments are double-buffered in registers across the four 16-byte K chunks of a tile so the LDS
// latency of chunk s+1 hides under the MFMAs of chunk s.
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int WM, int WN, int TI, int TJ>
struct Shape {
  static constexpr int kBM = WM * TI * 32, kBN = WN * TJ * 32, kWaves = WM * WN, kThreadsS = kWaves * 64;
  static constexpr int kABytes = kBM * 128, kBBytes = kBN * 128, kStage = kABytes + kBBytes;
  static constexpr int kLds = 2 * kStage;
};

template <int ROWS, int NWAVES>
__device__ __forceinline__ void stage_rows(const char* __restrict__ g, int64_t ld_bytes, int row0, int row_max,
                                           int64_t kbyte, char* lds_tile, int wave, int lane) {
  constexpr int kInst = ROWS / 8;            // 1 KiB wave-instructions
  constexpr int kPer = kInst / NWAVES;
  static_assert(kInst % NWAVES == 0, "tile rows must split evenly over the waves");
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const int inst = wave * kPer + i;
    const int r = inst * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int gr = row0 + r;
    gr = gr < row_max ? gr : row_max - 1;
    const char* src = g + (int64_t)gr * ld_bytes + kbyte + c * 16;
    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(lds_tile + inst * 1024), 16, 0, 0);
  }
}

// A tile of an implicit 3x3 convolution (GemmArgs::conv_*): row r of the tile is output pixel m0 + r, the K-tile selects a tap
// and a 128-byte channel slice; the rows a lane stages are fixed for the whole K loop, so their (y, x) are computed once.
template <int ROWS, int NWAVES>
struct ConvRows {
  static constexpr int kPer = ROWS / 8 / NWAVES;
  int m[kPer], y[kPer], x[kPer];
  __device__ __forceinline__ void init(const GemmArgs& p, int m0, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int r = (wave * kPer + i) * 8 + (lane >> 3);
      int mm = m0 + r;
      mm = mm < p.M ? mm : p.M - 1;
      const int hw = p.conv_H * p.conv_W, rem = mm % hw;
      m[i] = mm; y[i] = rem / p.conv_W; x[i] = rem - y[i] * p.conv_W;
    }
  }
  __device__ __forceinline__ void stage(const GemmArgs& p, const char* __restrict__ g, int64_t ld_bytes, int k0_elems, int esz,
                                        char* lds_tile, int wave, int lane) const {
    const int tap = k0_elems / p.conv_C, c0 = k0_elems - tap * p.conv_C;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int inst = wave * kPer + i;
      const int r = inst * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      const bool in = (unsigned)(y[i] + dy) < (unsigned)p.conv_H && (unsigned)(x[i] + dx) < (unsigned)p.conv_W;
      const char* src = in ? g + (int64_t)(m[i] + dy * p.conv_W + dx) * ld_bytes + (int64_t)c0 * esz + c * 16
                           : reinterpret_cast<const char*>(p.conv_zero) + c * 16;
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(lds_tile + inst * 1024), 16, 0, 0);
    }
  }
};

// RANK (f32 only): 0 = the GEMM; 1 / 2 = GemmArgs::rank_mode (paired scores / rank counts instead of storing C)
template <typename T, typename TO, int WM, int WN, int TI, int TJ, bool CONV = false, int RANK = 0>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN * 64 == 256) ? 2 : 2) void gemm_nt_kernel(std::conditional_t<RANK != 0, GemmRankArgs, GemmArgs> p) {
  using SH = Shape<WM, WN, TI, TJ>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BK = 128 / (int)sizeof(T);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_n = (p.N + SH::kBN - 1) / SH::kBN;
  const int nwg = gridDim.x;
  const int t = xcd_remap(blockIdx.x, nwg);
  const int m0 = (t / tiles_n) * SH::kBM;
  const int n0 = (t % tiles_n) * SH::kBN;
  if constexpr (RANK == 1) {       // columns n0 .. n0 + kBN - 1 against paired columns rank_row0 + m0 .. + kBM - 1
    if (n0 + SH::kBN <= p.rank_row0 + m0 || n0 >= p.rank_row0 + m0 + SH::kBM) return;
  }
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, l31 = lane & 31;

  const char* gA = reinterpret_cast<const char*>(p.A);
  const char* gB = reinterpret_cast<const char*>(p.B);
  const int64_t lda_b = p.lda * (int64_t)sizeof(T), ldb_b = p.ldb * (int64_t)sizeof(T);

  // bf16: v_mfma_f32_16x16x32_bf16 (EZ_MI16; the accumulation order of the 8-phase kernel, which stays bit-identical to this one);
  // f32: v_mfma_f32_32x32x2_f32
  constexpr bool MI16 = (EZ_MI16 != 0) && std::is_same<T, bf16_t>::value;
  f32x16_t acc[MI16 ? 1 : TI][MI16 ? 1 : TJ];
  f32x4_t acc16[MI16 ? 2 * TI : 1][MI16 ? 2 * TJ : 1];
  if constexpr (MI16) {
#pragma unroll
    for (int i = 0; i < 2 * TI; ++i)
#pragma unroll
      for (int j = 0; j < 2 * TJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  const int nk = p.K / BK;
  ConvRows<SH::kBM, SH::kWaves> cr;
  if constexpr (CONV) {
    cr.init(p, m0, wave, lane);
    cr.stage(p, gA, lda_b, 0, (int)sizeof(T), smem, wave, lane);
  } else {
    stage_rows<SH::kBM, SH::kWaves>(gA, lda_b, m0, p.M, 0, smem, wave, lane);
  }
  stage_rows<SH::kBN, SH::kWaves>(gB, ldb_b, n0, p.N, 0, smem + SH::kABytes, wave, lane);

  const int arow = wm * TI * 32 + l31, brow = wn * TJ * 32 + l31;
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * SH::kStage;
    char* nxt = smem + ((kt + 1) & 1) * SH::kStage;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt landed everywhere; everyone is done reading `nxt`
    if (kt + 1 < nk) {
      const int64_t kb = (int64_t)(kt + 1) * 128;
      if constexpr (CONV) cr.stage(p, gA, lda_b, (kt + 1) * BK, (int)sizeof(T), nxt, wave, lane);
      else stage_rows<SH::kBM, SH::kWaves>(gA, lda_b, m0, p.M, kb, nxt, wave, lane);
      stage_rows<SH::kBN, SH::kWaves>(gB, ldb_b, n0, p.N, kb, nxt + SH::kABytes, wave, lane);
    }
    const char* tA = cur;
    const char* tB = cur + SH::kABytes;
    if constexpr (MI16) {
      // 16x16x32: two k-steps of 32 per tile; lane (row l15 of a 16-row block, k-quarter q4): chunk 4 s + q4
      const int l15 = lane & 15, q4 = lane >> 4;
      const int arow16 = wm * TI * 32 + l15, brow16 = wn * TJ * 32 + l15;
      uint4 a[2][2 * TI], b[2][2 * TJ];
#pragma unroll
      for (int i = 0; i < 2 * TI; ++i) a[0][i] = read_frag(tA, arow16 + i * 16, q4);
#pragma unroll
      for (int j = 0; j < 2 * TJ; ++j) b[0][j] = read_frag(tB, brow16 + j * 16, q4);
#pragma unroll
      for (int i = 0; i < 2 * TI; ++i) a[1][i] = read_frag(tA, arow16 + i * 16, 4 + q4);
#pragma unroll
      for (int j = 0; j < 2 * TJ; ++j) b[1][j] = read_frag(tB, brow16 + j * 16, 4 + q4);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2 * TI; ++i)
#pragma unroll
          for (int j = 0; j < 2 * TJ; ++j) mma16(acc16[i][j], b[s][j], a[s][i]);
    } else {
    uint4 a[2][TI], b[2][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) a[0][i] = read_frag(tA, arow + i * 32, h);
#pragma unroll
    for (int j = 0; j < TJ; ++j) b[0][j] = read_frag(tB, brow + j * 32, h);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < 3) {
        const int c = 2 * (s + 1) + h;
#pragma unroll
        for (int i = 0; i < TI; ++i) a[(s + 1) & 1][i] = read_frag(tA, arow + i * 32, c);
#pragma unroll
        for (int j = 0; j < TJ; ++j) b[(s + 1) & 1][j] = read_frag(tB, brow + j * 32, c);
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) mma32(acc[i][j], b[s & 1][j], a[s & 1][i], T());
    }
    }
  }

  // ---- epilogue: lane owns row m, 4 consecutive n per (j, q) ----
  float scale = p.alpha;
  if (p.scale_log != nullptr) scale *= expf(*p.scale_log);
  TO* C = reinterpret_cast<TO*>(p.C);
  TO* C2 = reinterpret_cast<TO*>(p.C2);
  const T* R = reinterpret_cast<const T*>(p.R);
  const T* U = reinterpret_cast<const T*>(p.U);
  constexpr bool kFast = IsFast<TO>::value;
  // (static_for: compile-time i/j keep the accumulators in registers whatever the unroller decides)
  // MI16: 2 TI x 2 TJ blocks of 16 x 16, lane = row l15, columns 4 (lane >> 4) .. + 3 (one quad per block);
  // otherwise TI x TJ blocks of 32 x 32, lane = row l31, columns 8 q + 4 h .. + 3 for q = 0..3
  constexpr int NI = MI16 ? 2 * TI : TI, NJ = MI16 ? 2 * TJ : TJ, NQ = MI16 ? 1 : 4, RB = MI16 ? 16 : 32;
  if constexpr (RANK != 0) {
    static_assert(RANK == 0 || (!MI16 && std::is_same<T, float>::value), "rank epilogues: f32 only");
    // Lane = row l31 of every 32-row block i; its columns are n0 + wn*TJ*32 + j*32 + q*8 + h*4 + e.  The values compared are the
    // ones ezclip_similarity would have stored (same kernel, same accumulation order); counts are integers, so the atomics that
    // combine waves and workgroups are order-independent.
    int colc[TJ][4][4];
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) colc[j][q][e] = 0;
    static_for<TI>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int m = m0 + wm * TI * 32 + i * 32 + l31;
      const bool mok = m < p.M;
      const int ig = p.rank_row0 + m;                       // this query's index in the whole set = its paired column
      float d = 0.f;
      if constexpr (RANK == 2) d = mok ? p.rank_diag[ig] : 0.f;
      int cr = 0;
      static_for<TJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int n = n0 + wn * TJ * 32 + j * 32 + q * 8 + h * 4 + e;
            const float v = acc[i][j][q * 4 + e] * scale;
            if (!mok || n >= p.N) continue;
            if constexpr (RANK == 1) {
              if (n == ig) p.rank_diag_out[ig] = v;
            } else {
              cr += (v > d || (v == d && n < ig)) ? 1 : 0;
              if (p.rank_cols != nullptr) {
                const float dc = p.rank_diag[n];
                colc[j][q][e] += (v > dc || (v == dc && ig < n)) ? 1 : 0;
              }
            }
          }
      });
      if constexpr (RANK == 2) {
        cr += __shfl_xor(cr, 32, 64);                       // the two half-waves hold the two column halves of a row
        if (h == 0 && mok && cr != 0) atomicAdd(p.rank_rows + m, cr);
      }
    });
    if constexpr (RANK == 2) {
      if (p.rank_cols != nullptr) {
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              int c = colc[j][q][e];                        // sum over the 32 rows (lanes l31) of the wave's row blocks
              c += __shfl_xor(c, 1, 64); c += __shfl_xor(c, 2, 64); c += __shfl_xor(c, 4, 64);
              c += __shfl_xor(c, 8, 64); c += __shfl_xor(c, 16, 64);
              const int n = n0 + wn * TJ * 32 + j * 32 + q * 8 + h * 4 + e;
              if (l31 == 0 && n < p.N && c != 0) atomicAdd(p.rank_cols + n, c);
            }
      }
    }
    return;
  }
  static_for<NI>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const int m = m0 + wm * TI * 32 + i * RB + (MI16 ? (lane & 15) : l31);
    if (m >= p.M) return;
    static_for<NJ>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int n = n0 + wn * TJ * 32 + j * RB + (MI16 ? (lane >> 4) * 4 : q * 8 + h * 4);
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (MI16) v[e] = acc16[i][j][e] * scale;
          else v[e] = acc[i][j][q * 4 + e] * scale;
        }
        if (p.vec_ok) {
          if (p.bias) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
          }
          if (C2) st4(C2 + (int64_t)m * p.ldc + n, v);
          if (U) {   // dX = (dY . W) * act'(u)
            float uv[4];
            ld4(U + (int64_t)m * p.ldu + n, uv);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= act_grad<kFast>(uv[e], p.act);
          } else if (p.act != ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_apply<kFast>(v[e], p.act);
          }
          if (R) {
            float rv[4];
            ld4(R + (int64_t)m * p.ldr + n, rv);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rv[e];
          }
          if (p.act == ACT_RELU_POST) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          st4(C + (int64_t)m * p.ldc + n, v);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (n + e >= p.N) break;
            float x = v[e];
            if (p.bias) x += p.bias[n + e];
            if (C2) Elem<TO>::st(C2 + (int64_t)m * p.ldc + n + e, x);
            if (U) x *= act_grad<kFast>(Elem<T>::ld(U + (int64_t)m * p.ldu + n + e), p.act);
            else x = act_apply<kFast>(x, p.act);
            if (R) x += Elem<T>::ld(R + (int64_t)m * p.ldr + n + e);
            if (p.act == ACT_RELU_POST) x = fmaxf(x, 0.f);
            Elem<TO>::st(C + (int64_t)m * p.ldc + n + e, x);
          }
        }
      }
    });
  });
}

template <typename T, typename TO, int WM, int WN, int TI, int TJ, bool CONV = false, int RANK = 0>
int launch_nt_shape(const std::conditional_t<RANK != 0, GemmRankArgs, GemmArgs>& p, hipStream_t stream) {
  using SH = Shape<WM, WN, TI, TJ>;
  const int tiles = ((p.M + SH::kBM - 1) / SH::kBM) * ((p.N + SH::kBN - 1) / SH::kBN);
  static LdsOptIn lds_opt;
  EZ_ENSURE_LDS((&gemm_nt_kernel<T, TO, WM, WN, TI, TJ, CONV, RANK>), lds_opt, SH::kLds);
  {
    ProfScope ps(PROF_GEMM, 2.0 * p.M * (double)p.N * p.K, stream);
    hipLaunchKernelGGL((gemm_nt_kernel<T, TO, WM, WN, TI, TJ, CONV, RANK>), dim3(tiles), dim3(SH::kThreadsS), SH::kLds, stream, p);
  }
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

// implicit 3x3 convolution: 256 x 256 tiles when they fill the chip, else 128 x 128
template <typename T>
int launch_nt_conv(const GemmArgs& p, hipStream_t stream) {
  if (p.N % 256 == 0 && (int64_t)((p.M + 255) / 256) * (p.N / 256) >= 192) return launch_nt_shape<T, T, 2, 4, 4, 2, true>(p, stream);
  return launch_nt_shape<T, T, 2, 2, 2, 2, true>(p, stream);
}

int g_gemm_variant = -1;   // -1: heuristic; 0: 128x128; 1: 256x256 (2-phase); 2: 256x256 8-phase (gemm8p.hip); 3: 64x64

template <typename T, typename TO>
int launch_nt(const GemmArgs& p, hipStream_t stream) {
  int v = g_gemm_variant;
  // 256 x 256 tiles only when they fill the chip: the B-row products of the CLS-only last blocks, the projections and the
  // similarity (M = 1024: 8..16 such tiles on 256 CUs, each walking the whole K) run four times as many 128 x 128 workgroups
  if (v != 0 && v != 1 && v != 3) {
    v = (p.N % 256 == 0 && (int64_t)((p.M + 255) / 256) * (p.N / 256) >= 192) ? 1 : 0;
    // ... and 64 x 64 tiles (four waves of one 32 x 32 accumulator: two LDS reads per MFMA, but these products are short of
    // workgroups, not of LDS bandwidth) when even the 128 x 128 tiles leave half the CUs idle: [1024, 768] is 48 of them
    if (v == 0 && (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128) < 128) v = 3;
  }
  if (v == 1) return launch_nt_shape<T, TO, 2, 4, 4, 2>(p, stream);
  if (v == 3) return launch_nt_shape<T, TO, 2, 2, 1, 1>(p, stream);
  return launch_nt_shape<T, TO, 2, 2, 2, 2>(p, stream);
}

}  // namespace

void set_gemm8p_ablate(int v);
void set_gemm_variant(int v) {   // 2x = 8-phase kernel with ablation code x (timing experiments)
  if (v >= 20 && v < 30) { set_gemm8p_ablate(v - 20); v = 2; } else set_gemm8p_ablate(0);
  g_gemm_variant = v;
}

// true when gemm_nt would run this product on the 8-phase kernel (callers that want its optional extras -- folded
// LayerNorm, row-stat partials -- ask first)
bool gemm_nt_uses_8p(const GemmArgs& p, int dtype) {
  if (!((g_gemm_variant < 0 || g_gemm_variant == 2) && gemm_nt_8p_eligible(p, dtype))) return false;
  if (g_gemm_variant == 2) return true;       // forced (tests, sweeps)
  // A dozen 256 x 256 tiles on 256 CUs (the B-row products of the CLS-only last blocks at N = 768; with K = 3072 each tile is a
  // full 86 us): the 128 x 128 kernel spreads the same work over four times as many workgroups.  (Below 512 rows both kernels
  // are launch-bound; those stay on the 8-phase kernel so that small batches run the code large ones do.)
  const int64_t tiles = (int64_t)((p.M + 255) >> 8) * (p.N >> 8);
  return tiles >= 16 || p.M < 512;
}

int gemm_nt_rank(GemmRankArgs p, hipStream_t stream) {
  EZ_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && (p.K * 4) % 128 == 0 && (p.lda * 4) % 16 == 0 && (p.ldb * 4) % 16 == 0 &&
                 ((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0,
             "gemm_nt_rank: bad operands (M %d N %d K %d: K must be a multiple of 32, rows 16-byte aligned)", p.M, p.N, p.K);
  EZ_REQUIRE(p.conv_H == 0 && !p.colsum && !p.rowstat_part && !p.ln_stats && !p.bias && !p.R && !p.U && !p.scale_log &&
                 p.act == ACT_NONE && p.rank_row0 >= 0 && p.rank_row0 + p.M <= p.N &&
                 ((p.rank_mode == 1 && p.rank_diag_out) || (p.rank_mode == 2 && p.rank_diag && p.rank_rows)),
             "gemm_nt_rank: bad fused-rank problem (mode %d, row0 %d, M %d, N %d)", p.rank_mode, p.rank_row0, p.M, p.N);
  return p.rank_mode == 1 ? launch_nt_shape<float, float, 2, 2, 2, 2, false, 1>(p, stream)
                          : launch_nt_shape<float, float, 2, 2, 2, 2, false, 2>(p, stream);
}

int gemm_nt(GemmArgs p, int dtype, hipStream_t stream) {
  EZ_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm_nt: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  const int esz = dtype == EZCLIP_BF16 ? 2 : 4;
  EZ_REQUIRE((p.K * esz) % 128 == 0, "gemm_nt: K=%d must be a multiple of %d", p.K, 128 / esz);
  EZ_REQUIRE((p.lda * esz) % 16 == 0 && (p.ldb * esz) % 16 == 0, "gemm_nt: lda/ldb must be 16-byte multiples");
  EZ_REQUIRE(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0, "gemm_nt: A/B must be 16-byte aligned");
  bool vec = (p.N % 4 == 0) && (p.ldc % 4 == 0) && ((uintptr_t)p.C % 16 == 0) &&
             (p.C2 == nullptr || (uintptr_t)p.C2 % 16 == 0) &&
             (p.bias == nullptr || (uintptr_t)p.bias % 16 == 0) &&
             (p.R == nullptr || (p.ldr % 4 == 0 && (uintptr_t)p.R % 16 == 0)) &&
             (p.U == nullptr || (p.ldu % 4 == 0 && (uintptr_t)p.U % 16 == 0));
  p.vec_ok = vec ? 1 : 0;
  if (p.conv_H > 0) {
    EZ_REQUIRE(p.conv_W > 0 && p.conv_C > 0 && p.conv_zero != nullptr && p.K == 9 * p.conv_C && (p.conv_C * esz) % 128 == 0 &&
                   p.M % (p.conv_H * p.conv_W) == 0 && p.lda >= p.conv_C && !p.out_f32 && !p.colsum && !p.rowstat_part && !p.ln_stats,
               "gemm_nt: bad implicit-convolution problem (H %d W %d C %d K %d M %d)", p.conv_H, p.conv_W, p.conv_C, p.K, p.M);
    return dtype == EZCLIP_F32 ? launch_nt_conv<float>(p, stream) : launch_nt_conv<bf16_t>(p, stream);
  }
  if (gemm_nt_uses_8p(p, dtype)) return gemm_nt_8p(p, stream);
  if (p.colsum != nullptr) {   // not the 8-phase kernel: separate column-sum pass after the GEMM
    float* cs = p.colsum;
    p.colsum = nullptr;
    int rc = gemm_nt(p, dtype, stream);
    if (rc != EZ_OK) return rc;
    return colsum_add(p.C, p.ldc, p.M, p.N, cs, p.out_f32 ? EZCLIP_F32 : dtype, stream);
  }
  EZ_REQUIRE(p.rowstat_part == nullptr, "gemm_nt: row-stat partials need the 8-phase bf16 kernel with a residual");
  EZ_REQUIRE(p.ln_stats == nullptr, "gemm_nt: the folded-LayerNorm epilogue needs the 8-phase bf16 kernel (M >= 256, N %% 256 == 0, K %% 128 == 0)");
  if (dtype == EZCLIP_F32) return launch_nt<float, float>(p, stream);
  if (dtype == EZCLIP_BF16) {
    if (p.out_f32) return launch_nt<bf16_t, float>(p, stream);
    return launch_nt<bf16_t, bf16_t>(p, stream);
  }
  set_error("gemm_nt: bad dtype %d", dtype);
  return EZ_ERR_INVALID;
}

// ---------------------------------------------------------------------------------
// Weight-gradient GEMM:  C[N,K] (+)= A[M,N]^T . B[M,K]   (contraction over the rows)
//
// dW = dY^T X for every nn.Linear of the path (autograd of easynlp/core/trainer.py:658-661).
// Both operands are row-major with the contraction index m as the *slow* dimension,
// but the bf16 MFMA wants 8 consecutive contraction elements per lane.  bf16: each
// thread pulls an 8(m) x 8(n) block with eight coalesced 16-byte loads, transposes it
// in registers (32 byte-permutes) and writes eight 16-byte chunks into the same
// swizzled [row][m] LDS image the NT kernel uses -- the inner loop is then identical.
// f32: the 32x32x2 MFMA takes one element per lane, so the row-major tile is used as is.
// The huge M is split across workgroups (f32 atomics into C: gradients accumulate anyway).
namespace {

constexpr int TN_BM_BF16 = 64;   // contraction rows per LDS tile (128 B of bf16 per image row)
constexpr int TN_BM_F32 = 32;

__device__ __forceinline__ void transpose8x8_bf16(const uint4 (&r)[8], uint4 (&o)[8]) {
  const uint32_t* w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = reinterpret_cast<const uint32_t*>(&r[i]);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    uint32_t d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t lo = w[2 * j][c >> 1], hi = w[2 * j + 1][c >> 1];
      d[j] = (c & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
    }
    o[c] = make_uint4(d[0], d[1], d[2], d[3]);
  }
}

// CONV: B is not a matrix in memory but the 3 x 3 neighbourhoods of an NHWC activation x [M = images * H * W, conv_C]:
//   B[m][tap * conv_C + c] = x[pixel m shifted by (tap / 3 - 1, tap % 3 - 1)][c], zero outside the image
// (the column order of rn_im2col3x3 -- resnet_train.hip -- so C comes out exactly as from the explicit im2col, bit for bit: the
// same tile values meet the same MFMA sequence).  A staging thread's column chunk is fixed, so its tap is a per-thread constant;
// per tile it divides once (first row -> y, x) and steps the pixel position along its eight rows.
template <typename T, bool CONV>
__global__ __launch_bounds__(kThreads, 2) void gemm_tn_kernel(GemmTNArgs p, int tiles_k, int ntiles, int rows_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (A image 16K + B image 16K)
  constexpr bool kBf16 = sizeof(T) == 2;
  constexpr int BMc = kBf16 ? TN_BM_BF16 : TN_BM_F32;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int tile = blockIdx.x % ntiles, split = blockIdx.x / ntiles;
  const int n0 = (tile / tiles_k) * BM;     // C rows  (columns of A)
  const int k0 = (tile % tiles_k) * BN;     // C cols  (columns of B)
  const int m_begin = split * rows_per_split;
  const int m_end = min(p.M, m_begin + rows_per_split);
  const int nsteps = (m_end - m_begin + BMc - 1) / BMc;

  // staging role: threads 0..127 -> A, 128..255 -> B
  const bool isB = tid >= 128;
  const int st = tid & 127;
  const T* g = reinterpret_cast<const T*>(isB ? p.B : p.A);
  const int64_t ld = isB ? p.ldb : p.lda;
  const int c0 = isB ? k0 : n0;
  const int cmax = isB ? p.K : p.N;
  uint4 regs[8];

  // CONV, B staging threads: column chunk -> (tap, channel), the tap's pixel shift
  int cv_col = 0, cv_dy = 0, cv_dx = 0, cv_shift = 0;
  bool cv_ok = false;
  if constexpr (CONV) {
    if (isB) {
      const int col = k0 + (kBf16 ? (st >> 3) * 8 : (st & 31) * 4);
      cv_ok = col < p.K;
      const int tap = cv_ok ? col / p.conv_C : 0;
      cv_col = col - tap * p.conv_C;
      cv_dy = tap / 3 - 1; cv_dx = tap - (tap / 3) * 3 - 1;
      cv_shift = cv_dy * p.conv_W + cv_dx;
    }
  }
  auto load_tile_conv = [&](int mt) {
    constexpr int stride = kBf16 ? 1 : 4;              // row step between a thread's eight loads
    const int m0 = mt + (kBf16 ? (st & 7) * 8 : (st >> 5));
    const int t = m0 / p.conv_W;
    int x = m0 - t * p.conv_W, y = t % p.conv_H;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = m0 + r * stride;
      const bool ok = cv_ok && m < m_end && (unsigned)(y + cv_dy) < (unsigned)p.conv_H && (unsigned)(x + cv_dx) < (unsigned)p.conv_W;
      regs[r] = ok ? *reinterpret_cast<const uint4*>(g + (int64_t)(m + cv_shift) * ld + cv_col) : make_uint4(0, 0, 0, 0);
      x += stride;
      if (x >= p.conv_W) { x -= p.conv_W; if (++y >= p.conv_H) y = 0; }      // (conv_W >= 4 = the largest stride: one wrap at most)
    }
  };
  auto load_tile = [&](int step) {
    const int mt = m_begin + step * BMc;
    if constexpr (CONV) {
      if (isB) { load_tile_conv(mt); return; }
    }
    if constexpr (kBf16) {
      const int mb = st & 7, nb = st >> 3;           // 8 x 16 blocks of 8(m) x 8(cols)
      const int col = c0 + nb * 8;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int m = mt + mb * 8 + r;
        regs[r] = (m < m_end && col < cmax) ? *reinterpret_cast<const uint4*>(g + (int64_t)m * ld + col)
                                            : make_uint4(0, 0, 0, 0);
      }
    } else {
      // [32 m][128 cols] f32: 1024 float4 chunks, 8 per staging thread; chunk = r*128 + st
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int ch = r * 128 + st;
        const int m = mt + (ch >> 5), col = c0 + (ch & 31) * 4;
        regs[r] = (m < m_end && col < cmax) ? *reinterpret_cast<const uint4*>(g + (int64_t)m * ld + col)
                                            : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto store_tile = [&](char* buf) {
    char* img = buf + (isB ? kTileBytes : 0);
    if constexpr (kBf16) {
      const int mb = st & 7, nb = st >> 3;
      uint4 o[8];
      transpose8x8_bf16(regs, o);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int row = nb * 8 + c;
        *reinterpret_cast<uint4*>(img + row * 128 + ((mb ^ ((row >> 1) & 7)) << 4)) = o[c];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) *reinterpret_cast<uint4*>(img + (r * 128 + st) * 16) = regs[r];
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nsteps > 0) {
    load_tile(0);
    store_tile(smem);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    char* cur = smem + (s & 1) * 2 * kTileBytes;
    char* nxt = smem + ((s + 1) & 1) * 2 * kTileBytes;
    if (s + 1 < nsteps) load_tile(s + 1);          // global loads fly during the MFMAs
    const char* tA = cur;
    const char* tB = cur + kTileBytes;
    if constexpr (kBf16) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = 2 * ks + h;
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
    } else {
      const float* fA = reinterpret_cast<const float*>(tA);
      const float* fB = reinterpret_cast<const float*>(tB);
#pragma unroll 4
      for (int kk = 0; kk < 16; ++kk) {
        const int m = 2 * kk + h;
        float a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = fA[m * 128 + wm * 64 + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = fB[m * 128 + wn * 64 + j * 32 + l31];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a[i], acc[i][j], 0, 0, 0);
      }
    }
    if (s + 1 < nsteps) store_tile(nxt);
    __syncthreads();
  }

  // lane owns C row n, 4 consecutive k per (j, q)
  const bool atomic = gridDim.x > (unsigned)ntiles;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + wm * 64 + i * 32 + l31;
    if (n >= p.N) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = k0 + wn * 64 + j * 32 + q * 8 + h * 4;
        float* c = p.C + (int64_t)n * p.ldc + k;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (k + e >= p.K) break;
          const float v = acc[i][j][q * 4 + e];
          if (atomic) atomicAdd(c + e, v);
          else c[e] = (p.accumulate ? c[e] : 0.f) + v;
        }
      }
  }
}

template <typename T, bool CONV = false>
int launch_tn(const GemmTNArgs& p, hipStream_t stream) {
  const int tiles_n = (p.N + BM - 1) / BM, tiles_k = (p.K + BN - 1) / BN;
  const int ntiles = tiles_n * tiles_k;
  constexpr int BMc = sizeof(T) == 2 ? TN_BM_BF16 : TN_BM_F32;
  // enough workgroups to fill 256 CUs x 2, but at least 8 contraction tiles each
  int splits = (1024 + ntiles - 1) / ntiles;
  const int max_splits = (p.M + 8 * BMc - 1) / (8 * BMc);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rows = (p.M + splits - 1) / splits;
  rows = (rows + BMc - 1) / BMc * BMc;
  splits = (p.M + rows - 1) / rows;
  const size_t lds = 4 * kTileBytes;
  static LdsOptIn lds_opt;
  EZ_ENSURE_LDS((&gemm_tn_kernel<T, CONV>), lds_opt, lds);
  if (splits > 1 && !p.accumulate) EZ_HIP(hipMemset2DAsync(p.C, p.ldc * 4, 0, (size_t)p.K * 4, p.N, stream));
  {
    ProfScope ps(PROF_GEMM, 2.0 * p.M * (double)p.N * p.K, stream);
    hipLaunchKernelGGL((gemm_tn_kernel<T, CONV>), dim3(ntiles * splits), dim3(kThreads), lds, stream, p, tiles_k, ntiles, rows);
  }
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace

int gemm_tn(GemmTNArgs p, int dtype, hipStream_t stream) {
  EZ_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm_tn: empty problem");
  const int esz = dtype_size(dtype);
  const int g = 16 / esz;
  EZ_REQUIRE(p.N % g == 0 && p.K % g == 0 && p.lda % g == 0 && p.ldb % g == 0, "gemm_tn: N, K, lda, ldb must be multiples of %d", g);
  EZ_REQUIRE(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0, "gemm_tn: A/B must be 16-byte aligned");
  if (p.conv_H > 0) {        // B = the 3 x 3 neighbourhoods of an NHWC activation (see the kernel): never the 8-phase kernel
    EZ_REQUIRE(p.conv_W >= 4 && p.conv_C > 0 && p.conv_C % g == 0 && p.ldb == p.conv_C && p.K == 9 * p.conv_C &&
                   p.M % (p.conv_H * p.conv_W) == 0,
               "gemm_tn (3x3 neighbourhoods): H %d W %d C %d ldb %lld K %d M %d (want W >= 4, ldb == C, K == 9 C, M a whole number of images)",
               p.conv_H, p.conv_W, p.conv_C, (long long)p.ldb, p.K, p.M);
    if (dtype == EZCLIP_F32) return launch_tn<float, true>(p, stream);
    if (dtype == EZCLIP_BF16) return launch_tn<bf16_t, true>(p, stream);
    set_error("gemm_tn: bad dtype %d", dtype);
    return EZ_ERR_INVALID;
  }
  if ((g_gemm_variant < 0 || g_gemm_variant == 2) && gemm_tn_8p_eligible(p, dtype)) return gemm_tn_8p(p, stream);
  if (dtype == EZCLIP_F32) return launch_tn<float>(p, stream);
  if (dtype == EZCLIP_BF16) return launch_tn<bf16_t>(p, stream);
  set_error("gemm_tn: bad dtype %d", dtype);
  return EZ_ERR_INVALID;
}

}  // namespace ezclip
