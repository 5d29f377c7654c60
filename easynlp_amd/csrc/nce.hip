// Tiled contrastive loss: InfoNCE forward and backward on embeddings without the [n, N] logit blocks.
//
// Reference math (easynlp/appzoo/clip/model.py:148,154-164): logits_per_text = exp(logit_scale) T I^T,
// loss = (CE(S, arange) + CE(S^T, arange)) / 2.  In the north star's global mode a rank owns n rows (offset `off`) of the
// all-gathered [N, e] text / image embeddings and evaluates ITS rows of both directions,
//   S_t = s T_loc I_all^T,  S_i = s I_loc T_all^T   (both [n, N]),
// so both log-sum-exps are rank-local.  Round 2 materialised the two blocks and their gradients in float32
// (4 x 32 MB at n = 1024, N = 8192, six f32-MFMA products, six transposes).  Here nothing of size n x N exists:
//
//   forward   workgroup = 64 owned rows x one chunk of the other modality's rows, streamed in tiles of 128: the score tile is an
//             MFMA product over e, every lane keeps a running (max, sum) of ITS query over the keys it sees (S^T layout: lane =
//             query), combined across lanes / waves / chunks at the end -> lse [n] per direction, and the diagonal logit.
//   backward  d T_all[j] = s c sum_k w_jk I_k,   w_jk = [j local] (exp(S_jk - lse_t[j]) - d_jk) + [k local] (exp(S_jk - lse_i[k]) - d_jk)
//             with S_jk = s T_j . I_k  (and the mirror image for d I_all): ONE pass per output matrix -- owned row tile, streamed
//             rows -- recomputes the score tile, forms w in registers, rounds it to bf16 into an LDS image and multiplies it
//             with the streamed rows again (second MFMA product, operands from a pre-transposed copy).  Rows that are not this
//             rank's only meet this rank's n rows; rows that are meet all N, split in chunks whose partial sums are added in a
//             fixed order (bit-reproducible; no float atomics).  d logit_scale = sum w S comes out of the same pass.
//
// Precision: the MFMA operands are bf16.  SEGS = 1 rounds the embeddings once (the bf16 pipeline: its towers' own noise on the
// embeddings is two orders above bf16 rounding of a unit vector's components); SEGS = 3 splits every operand x = hi + lo
// (lo = bf16(x - hi)) and accumulates hi.hi + hi.lo + lo.hi in float32 -- 2^-16 relative per product, which holds the f32
// pipeline's bounds (loss 1e-5, gradients 1e-5 absolute).  Work at n = 1024, N = 8192, e = 512: 17 GFLOP forward + 69 GFLOP
// backward per rank in bf16 (x 3 split) against 52 GFLOP of float32 MFMA (157 TF peak) + ~1 GB of logit traffic before.
#include "gemm_pipe.h"

namespace ezclip {
namespace {

constexpr int kNceThreads = 256;
constexpr int kSBuf = 24576;                 // one k-tile stage: owned tile 64 x 128 B + streamed tile 128 x 128 B
constexpr int kOffW = 2 * kSBuf;             // w images: [part 2][k half 2] x (64 rows x 128 B)
constexpr int kOffRing = kOffW + 32768;      // transposed-operand ring: [wave 4][slot 4] x (32 rows x 128 B)
constexpr int kOffLse = kOffRing + 65536;    // (lse', valid) of the 128 streamed rows of a step
constexpr int kOffRed = kOffLse + 1024;
constexpr int kNceLds = kOffRed + 2048;      // 150 528 B: one workgroup per CU
constexpr float kLog2e = 1.4426950408889634f;

struct NcePass {
  const bf16_t* Xhl = nullptr;    // owned rows   [N][2e]  (hi | lo)
  const bf16_t* Yhl = nullptr;    // streamed rows [N][2e]
  const bf16_t* Yt = nullptr;     // streamed rows transposed [2e][ldt]
  const float* lse_own = nullptr; // [n] log-sum-exp of the owned side's local rows   (backward)
  const float* lse_str = nullptr; // [n] ... of the streamed side's local rows
  float* out = nullptr;           // [N][e] gradient of the owned rows
  float* out_part = nullptr;      // [nchunk][n_pad][e] partial sums of the local rows' tiles (nchunk > 1)
  float* part = nullptr;          // forward: [nchunk][n_pad][2] (max, sum) in base 2
  float* diag = nullptr;          // forward: [n] s x_j . y_j
};
struct NceArgs {
  NcePass pass[2];
  const float* ls = nullptr;      // log of the logit scale (device)
  int N = 0, n = 0, off = 0, e = 0, ldt = 0;
  int chunk = 0, nchunk = 1;      // local tiles: streamed rows per workgroup (multiple of 128), number of chunks
  int n_wide = 0, n_lo = 0, n_hi = 0, n_pad = 0;   // owned tiles of 64 rows: local rows / rows below / rows above the local range
  float coef = 0.f;               // grad_scale * 0.5 / n
  float* dls_part = nullptr;      // backward: one partial of sum w S per workgroup of pass 0
};

__device__ __forceinline__ uint4 rd_frag(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const uint4*>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// f32 [N][e] -> hl [N][2e] bf16 (hi | lo) and, transposed, t [2e][ldt] (columns >= N zeroed)
__global__ __launch_bounds__(256) void nce_split_kernel(const float* __restrict__ x, int N, int e, int ldt, bf16_t* __restrict__ hl,
                                                        bf16_t* __restrict__ t) {
  __shared__ uint32_t tile[64][65];     // (hi | lo << 16) of x[r0 + r][c0 + c]
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    uint32_t v = 0;
    if (r0 + r < N) {
      const float f = x[(int64_t)(r0 + r) * e + c0 + c];
      const bf16_t hi = f32_to_bf16(f);
      const bf16_t lo = f32_to_bf16(f - bf16_to_f32(hi));
      v = (uint32_t)hi | ((uint32_t)lo << 16);
      hl[(int64_t)(r0 + r) * 2 * e + c0 + c] = hi;
      hl[(int64_t)(r0 + r) * 2 * e + e + c0 + c] = lo;
    }
    tile[r][c] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < ldt) {
      const uint32_t v = tile[r][c];
      t[(int64_t)(c0 + c) * ldt + r0 + r] = (bf16_t)(v & 0xffffu);
      t[(int64_t)(e + c0 + c) * ldt + r0 + r] = (bf16_t)(v >> 16);
    }
  }
}

// SEGS: 1 = bf16 operands, 3 = split operands (hi.hi + hi.lo + lo.hi).  FWD: running log-sum-exp only.  CB: e / 128 (backward).
template <int SEGS, bool FWD, int CB>
__global__ __launch_bounds__(kNceThreads, 1) void nce_tile_kernel(NceArgs p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const NcePass& ps = p.pass[blockIdx.y];
  const int e = FWD ? p.e : CB * 128;
  const int N = p.N, n = p.n, off = p.off;

  // ---- which rows this workgroup owns, which it streams
  int id = blockIdx.x, j0, j_end, y0, y_end, cidx = 0;
  const bool wide = id < p.n_wide * p.nchunk;
  if (wide) {
    const int t = id / p.nchunk;
    cidx = id - t * p.nchunk;
    j0 = off + 64 * t; j_end = min(off + n, j0 + 64);
    y0 = cidx * p.chunk; y_end = min(N, y0 + p.chunk);
  } else {
    int t = id - p.n_wide * p.nchunk;
    if (t < p.n_lo) { j0 = 64 * t; j_end = min(off, j0 + 64); }
    else { t -= p.n_lo; j0 = off + n + 64 * t; j_end = min(N, j0 + 64); }
    y0 = off & ~127; y_end = off + n;
  }
  const int steps = (y_end - y0 + 127) >> 7;
  const float s = expf(*p.ls), sc2 = s * kLog2e;

  const uint32_t lds_base = (uint32_t)(size_t)smem;
  const uint32_t row_b = (uint32_t)e * 4u;                     // bytes per row of the hl arrays
  const i32x4_t srdX = make_srd(ps.Xhl, (uint32_t)N * row_b);
  const i32x4_t srdY = make_srd(ps.Yhl, (uint32_t)N * row_b);
  const int nk1 = e >> 6, nk = SEGS * nk1;                     // k-tiles of 64 per product, per score tile

  // lane-constant pieces of the DMA addresses: a 1 KiB wave instruction covers 8 tile rows, lane = (row, 16-byte position);
  // the position holds source chunk (position ^ swizzle(row)), rd_frag undoes it
  const int dr = lane >> 3, dp = lane & 7;
  auto stage = [&](int yb, int kt, int b) {
    const int seg = SEGS == 1 ? 0 : kt / nk1, kk = kt - seg * nk1;
    const uint32_t xcol = (uint32_t)((seg == 2 ? e : 0) + kk * 64) * 2u, ycol = (uint32_t)((seg == 1 ? e : 0) + kk * 64) * 2u;
    const uint32_t dst = lds_base + (uint32_t)b * kSBuf;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int inst = wave * 2 + i, r = inst * 8 + dr;
      dma16(dst + (uint32_t)inst * 1024u, (uint32_t)(j0 + r) * row_b + xcol + (uint32_t)((dp ^ ((r >> 1) & 7)) << 4), srdX, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int inst = wave * 4 + i, r = inst * 8 + dr;
      dma16(dst + 8192u + (uint32_t)inst * 1024u, (uint32_t)(yb + r) * row_b + ycol + (uint32_t)((dp ^ ((r >> 1) & 7)) << 4), srdY, 0);
    }
  };

  // owned rows of this lane (S^T layout: lane = owned row of block i, registers = streamed rows)
  int jg[2];
  bool jv[2];
  float lo2[2];                   // backward: log2e * lse of the owned row, +inf when it is not a local row
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    jg[i] = j0 + 32 * i + l31;
    jv[i] = jg[i] < j_end;
    lo2[i] = INFINITY;
    if constexpr (!FWD) if (wide && jv[i]) lo2[i] = ps.lse_own[jg[i] - off] * kLog2e;
  }
  float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};     // forward
  float dls_acc = 0.f;                                            // backward
  f32x16_t oacc[2][CB > 0 ? CB : 1];
  if constexpr (!FWD) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][cb][r] = 0.f;
  }
  float2* lse_s = reinterpret_cast<float2*>(smem + kOffLse);

  stage(y0, 0, 0);
  for (int st = 0; st < steps; ++st) {
    const int yb = y0 + 128 * st;
    if constexpr (!FWD) {
      if (tid < 128) {            // (visible after the first barrier of the k loop; the previous step's readers are past B1)
        const int kg = yb + tid;
        const bool loc = kg >= off && kg < off + n;
        lse_s[tid] = make_float2(loc ? ps.lse_str[kg - off] * kLog2e : INFINITY, kg < y_end ? 1.0f : 0.0f);
      }
    }
    f32x16_t sacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[i][r] = 0.f;
    // ---- score tile: S^T[streamed 32 w .. 32 w + 31][owned 0 .. 63] of this wave, contraction over SEGS * e
    for (int kt = 0; kt < nk; ++kt) {
      wait_vm<0>();
      __syncthreads();
      if (kt + 1 < nk) stage(yb, kt + 1, (kt + 1) & 1);
      const char* tX = smem + (kt & 1) * kSBuf;
      const char* tY = tX + 8192;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 2 * q + h;
        const uint4 y = rd_frag(tY, wave * 32 + l31, c);
        const uint4 x0 = rd_frag(tX, l31, c), x1 = rd_frag(tX, 32 + l31, c);
        mma32(sacc[0], y, x0, bf16_t());
        mma32(sacc[1], y, x1, bf16_t());
      }
    }
    if (st + 1 < steps) stage(yb + 128, 0, 0);      // (buffer 0 was last read at k-tile nk - 2: nk is even)

    if constexpr (FWD) {
      // ---- running (max, sum) per lane over the keys it holds; the diagonal logit
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float x[16], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kg = yb + wave * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
          x[r] = kg < y_end ? sacc[i][r] * sc2 : -INFINITY;
          mx = fmaxf(mx, x[r]);
          if (kg == jg[i] && jv[i]) ps.diag[jg[i] - off] = sacc[i][r] * s;
        }
        const float mn = fmaxf(m_run[i], mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(x[r] - mn);
        l_run[i] = l_run[i] * __builtin_amdgcn_exp2f(m_run[i] - mn) + sum;
        m_run[i] = mn;
      }
    } else {
      // ---- w tile -> LDS images (bf16 hi [and lo]), row = owned, 64 streamed rows per image
      constexpr int PARTS = SEGS == 3 ? 2 : 1;
      const int kh = wave >> 1;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int jr = 32 * i + l31;
        float dsum = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float w[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int kl = wave * 32 + 8 * g + 4 * h + t;
            const float2 lm = lse_s[kl];
            const float a = sacc[i][4 * g + t];
            const float x = a * sc2;
            float v = __builtin_amdgcn_exp2f(x - lo2[i]) + __builtin_amdgcn_exp2f(x - lm.x);
            if (wide && yb + kl == jg[i]) v -= 2.0f;      // (the diagonal exists among the local rows only)
            v *= lm.y;
            w[t] = v;
            dsum = fmaf(v, a, dsum);
          }
          const uint32_t chunk = (uint32_t)(4 * (wave & 1) + g) ^ (uint32_t)((jr >> 1) & 7);
          char* dst = smem + kOffW + kh * 8192 + jr * 128 + (chunk << 4) + 8 * h;
          const uint32_t h01 = pack_bf16x2(w[0], w[1]), h23 = pack_bf16x2(w[2], w[3]);
          *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h23);
          if constexpr (PARTS == 2) {
            const float r0 = w[0] - __uint_as_float(h01 << 16), r1 = w[1] - __uint_as_float(h01 & 0xffff0000u);
            const float r2 = w[2] - __uint_as_float(h23 << 16), r3 = w[3] - __uint_as_float(h23 & 0xffff0000u);
            *reinterpret_cast<uint2*>(dst + 16384) = make_uint2(pack_bf16x2(r0, r1), pack_bf16x2(r2, r3));
          }
        }
        if (jv[i]) dls_acc += dsum;
      }
      __syncthreads();            // B1: the w images are complete
      // ---- out[owned 64][this wave's e / 4 columns] += w[64][128] . Y[128][columns]: operand tiles of 32 columns x 64 streamed
      // rows from the transposed copy through a wave-private ring (4 slots, 3 tiles ahead, counted waits)
      constexpr int NT = PARTS * 2 * CB;
      const i32x4_t srdT = make_srd(ps.Yt, (uint32_t)(2 * e) * (uint32_t)p.ldt * 2u);
      const uint32_t ldt_b = (uint32_t)p.ldt * 2u;
      const uint32_t ring = lds_base + kOffRing + (uint32_t)wave * 16384u;
      auto issue = [&](int ti) {
        const int part = ti / (2 * CB), khh = (ti / CB) & 1, cb = ti % CB;
        const uint32_t rowc = (uint32_t)(part * e + wave * (e >> 2) + 32 * cb);
        const uint32_t kbyte = (uint32_t)(yb + 64 * khh) * 2u;
#pragma unroll
        for (int inst = 0; inst < 4; ++inst) {
          const int r = inst * 8 + dr;
          dma16(ring + (uint32_t)(ti & 3) * 4096u + (uint32_t)inst * 1024u,
                (rowc + (uint32_t)r) * ldt_b + kbyte + (uint32_t)((dp ^ ((r >> 1) & 7)) << 4), srdT, 0);
        }
      };
      issue(0);
      if (NT > 1) issue(1);
      if (NT > 2) issue(2);
      uint4 wf[PARTS][2][4];
      static_for<NT>([&](auto tic) {
        constexpr int ti = decltype(tic)::value;
        constexpr int part = ti / (2 * CB), khh = (ti / CB) & 1, cb = ti % CB;
        if constexpr (ti + 3 < NT) issue(ti + 3);
        constexpr int newer = (NT - 1 - ti) < 3 ? (NT - 1 - ti) : 3;
        wait_vm<4 * newer>();
        if constexpr (cb == 0) {          // w fragments of this k half: hi always; lo only against the hi operand rows
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              wf[0][i][q] = rd_frag(smem + kOffW + khh * 8192, 32 * i + l31, 2 * q + h);
              if constexpr (PARTS == 2) if (part == 0) wf[1][i][q] = rd_frag(smem + kOffW + 16384 + khh * 8192, 32 * i + l31, 2 * q + h);
            }
        }
        const char* tT = smem + kOffRing + wave * 16384 + (ti & 3) * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint4 yt = rd_frag(tT, l31, 2 * q + h);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            mma32(oacc[i][cb], yt, wf[0][i][q], bf16_t());
            if constexpr (PARTS == 2) if (part == 0) mma32(oacc[i][cb], yt, wf[1][i][q], bf16_t());
          }
        }
      });
    }
  }

  float* red = reinterpret_cast<float*>(smem + kOffRed);
  if constexpr (FWD) {
    // lanes h = 0 / 1 hold disjoint keys of the same query; then the four waves; one (max, sum) per row and chunk
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float m2 = __shfl_xor(m_run[i], 32, 64), l2 = __shfl_xor(l_run[i], 32, 64);
      const float M = fmaxf(m_run[i], m2);
      const float Lq = l_run[i] * __builtin_amdgcn_exp2f(m_run[i] - M) + l2 * __builtin_amdgcn_exp2f(m2 - M);
      if (h == 0) { red[(wave * 64 + 32 * i + l31) * 2] = M; red[(wave * 64 + 32 * i + l31) * 2 + 1] = Lq; }
    }
    __syncthreads();
    if (tid < 64 && j0 + tid < j_end) {
      float M = -1e30f;
#pragma unroll
      for (int w = 0; w < 4; ++w) M = fmaxf(M, red[(w * 64 + tid) * 2]);
      float Lq = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) Lq += red[(w * 64 + tid) * 2 + 1] * __builtin_amdgcn_exp2f(red[(w * 64 + tid) * 2] - M);
      float* dst = ps.part + ((int64_t)cidx * p.n_pad + (j0 + tid - off)) * 2;
      dst[0] = M; dst[1] = Lq;
    }
  } else {
    const bool direct = !wide || p.nchunk == 1;
    const float osc = direct ? s * p.coef : 1.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!jv[i]) continue;
      float* row = direct ? ps.out + (int64_t)jg[i] * e : ps.out_part + ((int64_t)cidx * p.n_pad + (jg[i] - off)) * e;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(row + wave * (e >> 2) + 32 * cb + 8 * g + 4 * h) =
              make_float4(oacc[i][cb][4 * g] * osc, oacc[i][cb][4 * g + 1] * osc, oacc[i][cb][4 * g + 2] * osc, oacc[i][cb][4 * g + 3] * osc);
    }
    if (blockIdx.y == 0) {        // d logit_scale = sum dS . S = s coef sum w (x . y); every (j, k) pair is met once in pass 0
      const float v = wave_sum(dls_acc);
      __syncthreads();
      if (lane == 0) red[wave] = v;
      __syncthreads();
      if (tid == 0) p.dls_part[blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) * s * p.coef;
    }
  }
}

// lse = ln2 (M + log2 L) over the chunks of a row; row_loss = lse - diag
__global__ __launch_bounds__(256) void nce_lse_combine_kernel(const float* part0, const float* part1, const float* diag0, const float* diag1,
                                                              int nchunk, int n_pad, int n, float* lse0, float* lse1, float* rl0,
                                                              float* rl1) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const float* part = d ? part1 : part0;
    float M = -1e30f;
    for (int c = 0; c < nchunk; ++c) M = fmaxf(M, part[((int64_t)c * n_pad + r) * 2]);
    float Lq = 0.f;
    for (int c = 0; c < nchunk; ++c) Lq += part[((int64_t)c * n_pad + r) * 2 + 1] * exp2f(part[((int64_t)c * n_pad + r) * 2] - M);
    const float lse = (M + log2f(Lq)) * 0.6931471805599453f;
    (d ? lse1 : lse0)[r] = lse;
    (d ? rl1 : rl0)[r] = lse - (d ? diag1 : diag0)[r];
  }
}

// out[off + r][c] = s coef sum_chunk part[chunk][r][c]   (fixed order)
__global__ __launch_bounds__(256) void nce_out_combine_kernel(const float* part, int nchunk, int n_pad, int n, int off, int e,
                                                              const float* ls, float coef, float* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // float4 index inside [n][e]
  if (i >= (int64_t)n * (e >> 2)) return;
  const float sc = expf(*ls) * coef;
  const float4* p4 = reinterpret_cast<const float4*>(part);
  float4 a = p4[i];
  for (int c = 1; c < nchunk; ++c) {
    const float4 b = p4[(int64_t)c * n_pad * (e >> 2) + i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  reinterpret_cast<float4*>(out + (int64_t)off * e)[i] = make_float4(a.x * sc, a.y * sc, a.z * sc, a.w * sc);
}

struct NceGeom {
  int ldt, chunk, nchunk, n_wide, n_lo, n_hi, n_pad, items_f, items_b;
};
NceGeom nce_geometry(int n, int N, int off) {
  NceGeom g;
  g.ldt = (N + 127) / 128 * 128;
  g.n_wide = (n + 63) / 64;
  g.n_lo = (off + 63) / 64;
  g.n_hi = (N - off - n + 63) / 64;
  g.n_pad = g.n_wide * 64;
  // local tiles meet all N rows: split them so that a launch has a few hundred workgroups and a local tile's chunk is about as
  // long as what the other tiles stream (this rank's n rows); at most 8 chunks (their partial sums are 8 n e floats)
  int want = (256 + g.n_wide - 1) / g.n_wide;
  const int by_len = (N + n - 1) / n;
  if (want < by_len) want = by_len;
  if (want > 8) want = 8;
  if (want < 1) want = 1;
  g.chunk = ((N + want - 1) / want + 127) / 128 * 128;
  g.nchunk = (N + g.chunk - 1) / g.chunk;
  g.items_f = g.n_wide * g.nchunk;
  g.items_b = g.items_f + g.n_lo + g.n_hi;
  return g;
}

struct NceTiledWS {
  bf16_t *Thl, *Ihl, *Tt, *It;
  float *part[2], *diag[2], *lse[2], *rl[2], *out_part[2], *dls_part;
};
size_t nce_tiled_layout(int n, int N, int off_unused, int e, void* base, NceTiledWS* out, const NceGeom& g) {
  char* b = (char*)base;
  size_t o = 0;
  auto take = [&](size_t bytes) { void* r = b ? b + o : nullptr; o += (bytes + 255) / 256 * 256; return r; };
  NceTiledWS w;
  w.Thl = (bf16_t*)take((size_t)N * 2 * e * 2); w.Ihl = (bf16_t*)take((size_t)N * 2 * e * 2);
  w.Tt = (bf16_t*)take((size_t)2 * e * g.ldt * 2); w.It = (bf16_t*)take((size_t)2 * e * g.ldt * 2);
  for (int d = 0; d < 2; ++d) {
    w.part[d] = (float*)take((size_t)g.nchunk * g.n_pad * 2 * 4);
    w.diag[d] = (float*)take((size_t)n * 4); w.lse[d] = (float*)take((size_t)n * 4); w.rl[d] = (float*)take((size_t)n * 4);
    w.out_part[d] = g.nchunk > 1 ? (float*)take((size_t)g.nchunk * g.n_pad * e * 4) : nullptr;
  }
  w.dls_part = (float*)take((size_t)g.items_b * 4);
  if (out) *out = w;
  return o + 256;
}

template <int SEGS, int CB>
int nce_launch_bwd(const NceArgs& a, int items, hipStream_t st) {
  static LdsOptIn lds_opt;
  EZ_ENSURE_LDS((&nce_tile_kernel<SEGS, false, CB>), lds_opt, kNceLds);
  hipLaunchKernelGGL((nce_tile_kernel<SEGS, false, CB>), dim3(items, 2), dim3(kNceThreads), kNceLds, st, a);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}
template <int SEGS>
int nce_launch_fwd(const NceArgs& a, int items, hipStream_t st) {
  static LdsOptIn lds_opt;
  EZ_ENSURE_LDS((&nce_tile_kernel<SEGS, true, 0>), lds_opt, kNceLds);
  hipLaunchKernelGGL((nce_tile_kernel<SEGS, true, 0>), dim3(items, 2), dim3(kNceThreads), kNceLds, st, a);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}
template <int SEGS>
int nce_dispatch_bwd(const NceArgs& a, int items, hipStream_t st) {
  switch (a.e / 128) {
    case 1: return nce_launch_bwd<SEGS, 1>(a, items, st);
    case 2: return nce_launch_bwd<SEGS, 2>(a, items, st);
    case 4: return nce_launch_bwd<SEGS, 4>(a, items, st);
    case 6: return nce_launch_bwd<SEGS, 6>(a, items, st);
    case 8: return nce_launch_bwd<SEGS, 8>(a, items, st);
    default: break;
  }
  set_error("infonce_tiled: embed dim %d not instantiated", a.e);
  return EZ_ERR_UNSUPPORTED;
}

}  // namespace

bool infonce_tiled_eligible(int e) { return e == 128 || e == 256 || e == 512 || e == 768 || e == 1024; }

size_t infonce_tiled_workspace_bytes(int n, int N, int e) {
  // (the geometry depends on the offset only through the number of 64-row tiles below / above the local rows: take the worst case)
  NceGeom g = nce_geometry(n, N, 0);
  g.items_b += 2;
  return nce_tiled_layout(n, N, 0, e, nullptr, nullptr, g);
}

int infonce_tiled(const float* T, const float* I, int n, int N, int off, int e, const float* ls, float grad_scale, int split,
                  float* loss, float* dT, float* dI, float* dls, void* wsp, size_t ws_bytes, hipStream_t st) {
  EZ_REQUIRE(infonce_tiled_eligible(e), "infonce_tiled: embed dim %d (128, 256, 512, 768 or 1024)", e);
  EZ_REQUIRE((int64_t)N * e * 4 < (int64_t)1 << 31, "infonce_tiled: N = %d rows of %d exceed a 2 GiB buffer descriptor", N, e);
  NceGeom g = nce_geometry(n, N, off);
  NceTiledWS w;
  NceGeom gl = g;
  gl.items_b = nce_geometry(n, N, 0).items_b + 2;       // (layout as sized by infonce_tiled_workspace_bytes)
  const size_t need = nce_tiled_layout(n, N, off, e, wsp, &w, gl);
  EZ_REQUIRE(ws_bytes >= need, "infonce_tiled: workspace too small (%zu < %zu)", ws_bytes, need);
  const bool bwd = dT != nullptr;
  hipLaunchKernelGGL(nce_split_kernel, dim3(g.ldt / 64, e / 64), dim3(256), 0, st, T, N, e, g.ldt, w.Thl, w.Tt);
  hipLaunchKernelGGL(nce_split_kernel, dim3(g.ldt / 64, e / 64), dim3(256), 0, st, I, N, e, g.ldt, w.Ihl, w.It);
  EZ_LAUNCH_CHECK();
  NceArgs a;
  a.ls = ls; a.N = N; a.n = n; a.off = off; a.e = e; a.ldt = g.ldt; a.chunk = g.chunk; a.nchunk = g.nchunk;
  a.n_wide = g.n_wide; a.n_lo = g.n_lo; a.n_hi = g.n_hi; a.n_pad = g.n_pad;
  a.coef = grad_scale * 0.5f / n;
  a.dls_part = w.dls_part;
  // pass 0: owned text rows against image rows (lse_t, d text); pass 1: owned image rows against text rows (lse_i, d image)
  a.pass[0].Xhl = w.Thl; a.pass[0].Yhl = w.Ihl; a.pass[0].Yt = w.It;
  a.pass[1].Xhl = w.Ihl; a.pass[1].Yhl = w.Thl; a.pass[1].Yt = w.Tt;
  for (int d = 0; d < 2; ++d) {
    a.pass[d].part = w.part[d]; a.pass[d].diag = w.diag[d];
    a.pass[d].lse_own = w.lse[d]; a.pass[d].lse_str = w.lse[1 - d];
    a.pass[d].out = d ? dI : dT; a.pass[d].out_part = w.out_part[d];
  }
  {
    NceArgs f = a;
    f.n_lo = f.n_hi = 0;
    if (split) { int rc = nce_launch_fwd<3>(f, g.items_f, st); if (rc != EZ_OK) return rc; }
    else { int rc = nce_launch_fwd<1>(f, g.items_f, st); if (rc != EZ_OK) return rc; }
  }
  hipLaunchKernelGGL(nce_lse_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w.part[0], w.part[1], w.diag[0], w.diag[1],
                     g.nchunk, g.n_pad, n, w.lse[0], w.lse[1], w.rl[0], w.rl[1]);
  EZ_LAUNCH_CHECK();
  int rc = sum_scaled(w.rl[0], n, 0.5f / n, loss, 0, st);
  if (rc == EZ_OK) rc = sum_scaled(w.rl[1], n, 0.5f / n, loss, 1, st);
  if (rc != EZ_OK || !bwd) return rc;
  rc = split ? nce_dispatch_bwd<3>(a, g.items_b, st) : nce_dispatch_bwd<1>(a, g.items_b, st);
  if (rc != EZ_OK) return rc;
  if (g.nchunk > 1) {
    const int64_t q = (int64_t)n * (e / 4);
    for (int d = 0; d < 2; ++d)
      hipLaunchKernelGGL(nce_out_combine_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, st, w.out_part[d], g.nchunk, g.n_pad,
                         n, off, e, ls, a.coef, d ? dI : dT);
    EZ_LAUNCH_CHECK();
  }
  return sum_scaled(w.dls_part, g.items_b, 1.0f, dls, 0, st);
}

}  // namespace ezclip
