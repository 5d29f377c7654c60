// Fused multi-head self-attention for short sequences (head_dim 64) on gfx950.
//
// Reference semantics:
//   ViT : nn.MultiheadAttention via ResidualAttentionBlock.attention
//         (modeling_chineseclip.py:188,198-200): softmax((q*hd^-1/2) k^T) v, no mask.
//   BERT: BertSelfAttention.forward (bert/modeling_bert.py:210-244):
//         softmax(q k^T / sqrt(hd) + (1-mask)*-10000) v.
// hd^-1/2 = 0.125 is a power of two, so scaling q first or the scores afterwards
// is the same number in both f32 and bf16.
//
// One workgroup per (batch, head).  The whole K (row-major, bank-swizzled) and
// V^T (key-contiguous) of the head live in LDS (L <= 288), so the softmax is a
// plain full-row softmax -- the score matrix never leaves registers:
//   S^T tile = mfma(Kfrag, Qfrag)  -> lane holds one query column q = lane&31 and
//              16 keys per 32-key tile: the row max/sum is an in-lane reduction
//              plus ONE cross-lane exchange (lane ^ 32);
//   O^T      = mfma(V^T frag, P)   -> the P fragment is exactly the lane's own
//              registers (no LDS round trip, no shuffles); the lane ends up with 4
//              consecutive d for its query: vector stores.
// The same code path serves f32 (v_mfma_f32_32x32x2_f32) and bf16
// (v_mfma_f32_32x32x16_bf16); only chunk geometry differs.
#include <type_traits>

#include "ezclip_common.h"
#include "kernels.h"

namespace ezclip {
namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <typename T> struct Geo {
  static constexpr int SZ = (int)sizeof(T);
  static constexpr int RB = 64 * SZ;      // bytes per K row (head_dim 64)
  static constexpr int CPR = RB / 16;     // 16-byte chunks per row: 8 (bf16) / 16 (f32)
  static constexpr int NS = CPR / 2;      // MFMA chunk steps over d
  static constexpr int RPI = 1024 / RB;   // K rows per LDS-DMA wave-instruction
  __device__ static __forceinline__ int swz(int row) { return SZ == 2 ? ((row >> 1) & 7) : (row & 15); }
};

template <int N, typename F>
__device__ __forceinline__ void static_for_q(F&& f) {
  if constexpr (N > 0) {
    static_for_q<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// LDS carve-up for nt 32-key tiles (runtime): K rows | V^T [64][LP] | key bias
template <typename T>
struct Smem {
  int LKP, LP, vtOff, kbOff, bytes;
  __host__ __device__ explicit Smem(int nt) {
    LKP = 32 * nt;
    LP = LKP + 4;   // V^T row length in keys: 4*odd -> conflict-free b64/b128 fragment reads
    vtOff = LKP * Geo<T>::RB;
    kbOff = vtOff + 64 * LP * Geo<T>::SZ;
    bytes = kbOff + LKP * 4;
  }
};

template <typename T>
__device__ __forceinline__ uint4 read_k(const char* kt, int row, int chunk) {
  return *reinterpret_cast<const uint4*>(kt + row * Geo<T>::RB + ((chunk ^ Geo<T>::swz(row)) << 4));
}

// Fill K (LDS-DMA, swizzled source), V^T (register transpose) and the key bias row.
template <typename T>
__device__ __forceinline__ void stage_kv(const AttnArgs& a, const Smem<T>& S, int b, int head, char* smem, int tid,
                                         int nthreads, int wave, int nwaves, int lane, int key0 = 0) {
  // key0: first key of the block of S.LKP keys staged (0: the whole sequence; the chunked forward walks blocks)
  using G = Geo<T>;
  const int L = a.L;
  const int64_t rs = a.row_stride * G::SZ;
  const char* kbase = reinterpret_cast<const char*>(a.k) + ((int64_t)b * L * a.row_stride + head * 64) * G::SZ;
  const char* vbase = reinterpret_cast<const char*>(a.v) + ((int64_t)b * L * a.row_stride + head * 64) * G::SZ;
  const int ninst = S.LKP * G::RB / 1024;
  for (int inst = wave; inst < ninst; inst += nwaves) {
    const int r = inst * G::RPI + lane / G::CPR;
    const int c = (lane % G::CPR) ^ G::swz(r);
    const int gr = key0 + r < L ? key0 + r : L - 1;  // clamp: pad keys get a finite (masked) score
    __builtin_amdgcn_global_load_lds((glb_void*)(kbase + gr * rs + c * 16), (lds_void*)(smem + inst * 1024), 16, 0, 0);
  }
  // V^T: item = (d-chunk dc, key quad kq); consecutive lanes -> consecutive kq (conflict-free writes)
  const int nkq = S.LKP / 4;
  char* vt = smem + S.vtOff;
  for (int idx = tid; idx < nkq * G::CPR; idx += nthreads) {
    const int dc = idx / nkq, kq = idx % nkq;
    uint4 w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = key0 + 4 * kq + r;
      if (key < L) w[r] = *reinterpret_cast<const uint4*>(vbase + key * rs + dc * 16);
      else w[r] = make_uint4(0, 0, 0, 0);
    }
    const uint32_t w0[4] = {w[0].x, w[0].y, w[0].z, w[0].w};
    const uint32_t w1[4] = {w[1].x, w[1].y, w[1].z, w[1].w};
    const uint32_t w2[4] = {w[2].x, w[2].y, w[2].z, w[2].w};
    const uint32_t w3[4] = {w[3].x, w[3].y, w[3].z, w[3].w};
    if constexpr (G::SZ == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int sh = (e & 1) * 16;
        const uint32_t v0 = (w0[e >> 1] >> sh) & 0xffffu, v1 = (w1[e >> 1] >> sh) & 0xffffu;
        const uint32_t v2 = (w2[e >> 1] >> sh) & 0xffffu, v3 = (w3[e >> 1] >> sh) & 0xffffu;
        *reinterpret_cast<uint2*>(vt + ((8 * dc + e) * S.LP + 4 * kq) * 2) = make_uint2(v0 | (v1 << 16), v2 | (v3 << 16));
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        *reinterpret_cast<uint4*>(vt + ((4 * dc + e) * S.LP + 4 * kq) * 4) = make_uint4(w0[e], w1[e], w2[e], w3[e]);
    }
  }
  float* kb = reinterpret_cast<float*>(smem + S.kbOff);
  for (int key = tid; key < S.LKP; key += nthreads)
    kb[key] = key0 + key < L ? (a.key_bias ? a.key_bias[(int64_t)b * L + key0 + key] : 0.f) : -INFINITY;
}

// scaled + biased scores of one 32-key tile for this lane's query: x[r], key = 32t + (r&3) + 8(r>>2) + 4h
template <typename T>
__device__ __forceinline__ void score_tile(const char* kt, const float* kb, const uint4 (&qf)[Geo<T>::NS], int t, int l31,
                                           int h, float scale, float (&x)[16], int causal = 0, int qrow = 0, int key0 = 0) {
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < Geo<T>::NS; ++s) {
    const uint4 kf = read_k<T>(kt, 32 * t + l31, 2 * s + h);
    mma32(acc, kf, qf[s], T());   // D[key][q]
  }
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const float4 kb4 = *reinterpret_cast<const float4*>(kb + 32 * t + 8 * qd + 4 * h);
    x[4 * qd + 0] = fmaf(acc[4 * qd + 0], scale, kb4.x);
    x[4 * qd + 1] = fmaf(acc[4 * qd + 1], scale, kb4.y);
    x[4 * qd + 2] = fmaf(acc[4 * qd + 2], scale, kb4.z);
    x[4 * qd + 3] = fmaf(acc[4 * qd + 3], scale, kb4.w);
  }
  if (causal) {      // key0 = index of this tile's first key; keys after the query are masked
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (key0 + (r & 3) + 8 * (r >> 2) + 4 * h > qrow) x[r] = -INFINITY;
  }
}

// Dropout on a 32-key tile held query-major (lane = one query `row`, element r <-> key 32t + (r&3) + 8(r>>2) + 4h):
// v[r] <- keep ? v[r] / (1 - p) : 0.  One Philox call per aligned key quad.
__device__ __forceinline__ void drop_tile_qmajor(const DropCfg& d, uint32_t row, int t, int h, float (&v)[16]) {
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const uint4 w = drop_words(d, row, (uint32_t)(8 * t + 2 * qd + h));
    v[4 * qd + 0] = w.x >= d.thr ? v[4 * qd + 0] * d.scale : 0.f;
    v[4 * qd + 1] = w.y >= d.thr ? v[4 * qd + 1] * d.scale : 0.f;
    v[4 * qd + 2] = w.z >= d.thr ? v[4 * qd + 2] * d.scale : 0.f;
    v[4 * qd + 3] = w.w >= d.thr ? v[4 * qd + 3] * d.scale : 0.f;
  }
}
// The same decisions seen key-major (lane = one key `col`, element r <-> query row0 + (r&3) + 8(r>>2) + 4h):
// m[r] = 1 / (1 - p) or 0.
__device__ __forceinline__ void drop_factors_kmajor(const DropCfg& d, uint32_t row_base, int q0, int h, uint32_t col,
                                                    float (&m)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const uint32_t q = (uint32_t)(q0 + (r & 3) + 8 * (r >> 2) + 4 * h);
    m[r] = drop_word(d, row_base + q, col) >= d.thr ? d.scale : 0.f;
  }
}

template <typename T>
__global__ __launch_bounds__(512) void attn_fwd_kernel(AttnArgs a, int nt) {
  using G = Geo<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Smem<T> S(nt);
  const int head = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = nthreads >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int L = a.L;
  constexpr bool kFast = IsFast<T>::value;

  stage_kv<T>(a, S, b, head, smem, tid, nthreads, wave, nwaves, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const char* kt = smem;
  const char* vt = smem + S.vtOff;
  const float* kb = reinterpret_cast<const float*>(smem + S.kbOff);
  const int LP = S.LP;
  const int nqb = (L + 31) / 32;
  for (int qb = wave; qb < nqb; qb += nwaves) {
    const int q = qb * 32 + l31;
    const int qc = q < L ? q : L - 1;
    const char* qp = reinterpret_cast<const char*>(a.q) + (((int64_t)b * L + qc) * a.row_stride + head * 64) * G::SZ;
    uint4 qf[G::NS];
#pragma unroll
    for (int s = 0; s < G::NS; ++s) qf[s] = *reinterpret_cast<const uint4*>(qp + (2 * s + h) * 16);

    // ---- pass 1: row max (lane: one q, 16 keys per tile; partner lane^32 holds the other 16) ----
    float mx = -INFINITY;
#pragma unroll 1
    for (int t = 0; t < nt; ++t) {
      float x[16];
      score_tile<T>(kt, kb, qf, t, l31, h, a.scale, x, a.causal, q, 32 * t);
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, x[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));

    // ---- pass 2: p = exp(x - max) (scores recomputed bit-identically), O^T += V^T . P^T ----
    f32x16_t o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float sum = 0.f;
#pragma unroll 1
    for (int t = 0; t < nt; ++t) {
      float x[16];
      score_tile<T>(kt, kb, qf, t, l31, h, a.scale, x, a.causal, q, 32 * t);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        x[r] = kFast ? __expf(x[r] - mx) : expf(x[r] - mx);
        sum += x[r];
      }
      if (a.drop.thr != 0) drop_tile_qmajor(a.drop, ((uint32_t)b * a.H + head) * L + qc, t, h, x);   // bert :238
      if constexpr (G::SZ == 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          uint4 pc;
          pc.x = pack_bf16x2(x[8 * u + 0], x[8 * u + 1]);
          pc.y = pack_bf16x2(x[8 * u + 2], x[8 * u + 3]);
          pc.z = pack_bf16x2(x[8 * u + 4], x[8 * u + 5]);
          pc.w = pack_bf16x2(x[8 * u + 6], x[8 * u + 7]);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const char* vp = vt + ((dt * 32 + l31) * LP + 32 * t + 16 * u + 4 * h) * 2;
            const uint2 lo = *reinterpret_cast<const uint2*>(vp);
            const uint2 hi = *reinterpret_cast<const uint2*>(vp + 16);
            mma32(o[dt], make_uint4(lo.x, lo.y, hi.x, hi.y), pc, T());   // D[d][q]
          }
        }
      } else {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const uint4 pc = make_uint4(__float_as_uint(x[4 * qd + 0]), __float_as_uint(x[4 * qd + 1]),
                                      __float_as_uint(x[4 * qd + 2]), __float_as_uint(x[4 * qd + 3]));
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const uint4 vf = *reinterpret_cast<const uint4*>(vt + ((dt * 32 + l31) * LP + 32 * t + 8 * qd + 4 * h) * 4);
            mma32(o[dt], vf, pc, T());
          }
        }
      }
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (q < L) {
      T* cp = reinterpret_cast<T*>(a.ctx) + ((int64_t)b * L + q) * a.ctx_stride + head * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = o[dt][4 * qd + e] * inv;
          st4(cp + dt * 32 + 8 * qd + 4 * h, v);
        }
      if (a.lse != nullptr && h == 0) a.lse[((int64_t)b * a.H + head) * L + q] = mx + logf(sum);
    }
  }
}

// Sequences whose K and V^T images do not fit the 160 KiB of LDS (f32: L > 288 -- BERT goes to 512 positions,
// modeling_bert.py max_position_embeddings; bf16: L > 576): the keys are walked in blocks of `ch` tiles with an online
// softmax ACROSS blocks -- every wave keeps the running maximum / sum / output of its (up to QBW) query blocks in registers
// while the workgroup re-stages K / V^T for the next key block.  Inside a block the arithmetic is the two-pass one of
// attn_fwd_kernel (block maximum first, then p = exp(x - m)), so a single block reproduces it bit for bit.
template <typename T, int QBW>
__global__ __launch_bounds__(512) void attn_fwd_chunked_kernel(AttnArgs a, int nt, int ch) {
  using G = Geo<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Smem<T> S(ch);
  const int head = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = nthreads >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int L = a.L;
  constexpr bool kFast = IsFast<T>::value;
  const char* kt = smem;
  const char* vt = smem + S.vtOff;
  const float* kb = reinterpret_cast<const float*>(smem + S.kbOff);
  const int LP = S.LP;
  const int nqb = (L + 31) / 32;

  f32x16_t o[QBW][2];
  float m[QBW], l[QBW];
#pragma unroll
  for (int i = 0; i < QBW; ++i) {
    m[i] = -INFINITY; l[i] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][dt][r] = 0.f;
  }
  for (int c0 = 0; c0 < nt; c0 += ch) {
    const int ntc = nt - c0 < ch ? nt - c0 : ch;
    __syncthreads();                                   // every wave is done with the previous key block
    stage_kv<T>(a, S, b, head, smem, tid, nthreads, wave, nwaves, lane, 32 * c0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    static_for_q<QBW>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int qb = wave + i * nwaves;
      if (qb < nqb) {
        const int q = qb * 32 + l31;
        const int qc = q < L ? q : L - 1;
        const char* qp = reinterpret_cast<const char*>(a.q) + (((int64_t)b * L + qc) * a.row_stride + head * 64) * G::SZ;
        uint4 qf[G::NS];
#pragma unroll
        for (int s = 0; s < G::NS; ++s) qf[s] = *reinterpret_cast<const uint4*>(qp + (2 * s + h) * 16);
        float mx = -INFINITY;
#pragma unroll 1
        for (int t = 0; t < ntc; ++t) {
          float x[16];
          score_tile<T>(kt, kb, qf, t, l31, h, a.scale, x, a.causal, q, 32 * (c0 + t));
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, x[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // (no branch here: MFMAs ignore EXEC.  A causal query that precedes the whole key block has mx = -inf, then mn = m[i]
        // -- finite, block 0 always holds key 0 <= q -- alpha = 1 and every p = exp(-inf) = 0: nothing is added)
        {
          const float mn = fmaxf(m[i], mx);
          const float alpha = kFast ? __expf(m[i] - mn) : expf(m[i] - mn);     // first block: exp(-inf) = 0 on o = 0
          m[i] = mn;
          l[i] *= alpha;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][dt][r] *= alpha;
          float sum = 0.f;
#pragma unroll 1
          for (int t = 0; t < ntc; ++t) {
            float x[16];
            score_tile<T>(kt, kb, qf, t, l31, h, a.scale, x, a.causal, q, 32 * (c0 + t));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              x[r] = kFast ? __expf(x[r] - mn) : expf(x[r] - mn);
              sum += x[r];
            }
            if (a.drop.thr != 0) drop_tile_qmajor(a.drop, ((uint32_t)b * a.H + head) * L + qc, c0 + t, h, x);   // bert :238
            if constexpr (G::SZ == 2) {
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                uint4 pc;
                pc.x = pack_bf16x2(x[8 * u + 0], x[8 * u + 1]);
                pc.y = pack_bf16x2(x[8 * u + 2], x[8 * u + 3]);
                pc.z = pack_bf16x2(x[8 * u + 4], x[8 * u + 5]);
                pc.w = pack_bf16x2(x[8 * u + 6], x[8 * u + 7]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                  const char* vp = vt + ((dt * 32 + l31) * LP + 32 * t + 16 * u + 4 * h) * 2;
                  const uint2 lo = *reinterpret_cast<const uint2*>(vp);
                  const uint2 hi = *reinterpret_cast<const uint2*>(vp + 16);
                  mma32(o[i][dt], make_uint4(lo.x, lo.y, hi.x, hi.y), pc, T());   // D[d][q]
                }
              }
            } else {
#pragma unroll
              for (int qd = 0; qd < 4; ++qd) {
                const uint4 pc = make_uint4(__float_as_uint(x[4 * qd + 0]), __float_as_uint(x[4 * qd + 1]),
                                            __float_as_uint(x[4 * qd + 2]), __float_as_uint(x[4 * qd + 3]));
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                  const uint4 vf = *reinterpret_cast<const uint4*>(vt + ((dt * 32 + l31) * LP + 32 * t + 8 * qd + 4 * h) * 4);
                  mma32(o[i][dt], vf, pc, T());
                }
              }
            }
          }
          l[i] += sum;
        }
      }
    });
  }
  static_for_q<QBW>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const int qb = wave + i * nwaves;
    if (qb < nqb) {
      const int q = qb * 32 + l31;
      float sum = l[i];
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.0f / sum;
      if (q < L) {
        T* cp = reinterpret_cast<T*>(a.ctx) + ((int64_t)b * L + q) * a.ctx_stride + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = o[i][dt][4 * qd + e] * inv;
            st4(cp + dt * 32 + 8 * qd + 4 * h, v);
          }
        if (a.lse != nullptr && h == 0) a.lse[((int64_t)b * a.H + head) * L + q] = m[i] + logf(sum);
      }
    }
  });
}

template <typename T, int QBW>
int launch_fwd_chunked(const AttnArgs& a, int nt, int ch, hipStream_t stream) {
  const Smem<T> S(ch);
  static LdsOptIn lds_opt;
  EZ_ENSURE_LDS((&attn_fwd_chunked_kernel<T, QBW>), lds_opt, S.bytes);
  {
    ProfScope ps(PROF_ATTN, 4.0 * a.B * a.H * (double)a.L * a.L * 64, stream);
    hipLaunchKernelGGL((attn_fwd_chunked_kernel<T, QBW>), dim3(a.H, a.B), dim3(512), S.bytes, stream, a, nt, ch);
  }
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

template <typename T>
int launch_fwd(const AttnArgs& a, hipStream_t stream) {
  const int nt = (a.L + 31) / 32;
  const Smem<T> S(nt);
  if (S.bytes > 160 * 1024) {
    int ch = nt;
    while (ch > 1 && Smem<T>(ch).bytes > 160 * 1024) --ch;
    if (nt <= 8) return launch_fwd_chunked<T, 1>(a, nt, ch, stream);
    if (nt <= 16) return launch_fwd_chunked<T, 2>(a, nt, ch, stream);      // L <= 512: two query blocks per wave
    set_error("attention_fwd: sequence length %d > 512 is not supported by the general kernels", a.L);
    return EZ_ERR_UNSUPPORTED;
  }
  static LdsOptIn lds_opt;
  EZ_ENSURE_LDS((&attn_fwd_kernel<T>), lds_opt, S.bytes);
  const int nw = nt < 8 ? nt : 8;
  {
    ProfScope ps(PROF_ATTN, 4.0 * a.B * a.H * (double)a.L * a.L * 64, stream);   // QK^T + PV, unpadded
    hipLaunchKernelGGL((attn_fwd_kernel<T>), dim3(a.H, a.B), dim3(nw * 64), S.bytes, stream, a, nt);
  }
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------
// Attention for ONE query per sample (the CLS token): the last block of either tower only feeds x[:, 0] to the head
// (ln_post(x[:, 0, :]) @ proj, modeling_chineseclip.py:248-251; bert(...)[0][:, 0, :] @ text_projection :349-350), so its
// attention output is needed for that row alone.  One wave per (sample, head): lanes split the keys for q.k (a K row is
// one 128-byte line per lane), a wave softmax, then lane = d accumulates sum_key p_key V[key][d] (V rows read coalesced).
// HBM traffic: K and V once -- the same bytes the full kernel reads for them; no Q block, no ctx block.
namespace {
// 8 of the 64 head dimensions per lane, chunk c = lane & 7: bf16 -> elements 8c .. 8c+7 (one 16-byte load);
// f32 -> 4c .. 4c+3 and 32+4c .. 32+4c+3 (two 16-byte loads).  A group of 8 lanes covers a whole row in full lines.
__device__ __forceinline__ void load8(const bf16_t* row, int c, float (&v)[8]) {
  unpack_chunk(*reinterpret_cast<const uint4*>(row + 8 * c), v, bf16_t());
}
__device__ __forceinline__ void load8(const float* row, int c, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(row + 4 * c), b = *reinterpret_cast<const float4*>(row + 32 + 4 * c);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(bf16_t* row, int c, const float (&v)[8]) {
  *reinterpret_cast<uint4*>(row + 8 * c) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                                      pack_bf16x2(v[6], v[7]));
}
__device__ __forceinline__ void store8(float* row, int c, const float (&v)[8]) {
  *reinterpret_cast<float4*>(row + 4 * c) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(row + 32 + 4 * c) = make_float4(v[4], v[5], v[6], v[7]);
}

// One wave per (sample, head); lane = (key group g = lane >> 3, chunk c = lane & 7): every load instruction of the wave
// covers 8 consecutive keys in full 128-byte (bf16) lines.  Scores go through a per-wave LDS row (512 floats).
template <typename T>
__global__ __launch_bounds__(256) void attn_cls_fwd_kernel(AttnArgs a, const T* q_cls, int64_t q_stride, T* ctx_cls,
                                                           int64_t ctx_stride) {
  __shared__ float sc_all[4][512];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int bh = blockIdx.x * 4 + w;
  if (bh >= a.B * a.H) return;
  float* sc = sc_all[w];
  const int b = bh / a.H, head = bh - b * a.H;
  const int L = a.lens ? a.lens[b] : a.L, g = lane >> 3, c = lane & 7;               // (packed batches: AttnArgs::cu / lens)
  const int64_t row0 = a.cu ? (int64_t)a.cu[b] : (int64_t)b * a.L;
  const T* kbase = reinterpret_cast<const T*>(a.k) + row0 * a.row_stride + head * 64;
  const T* vbase = reinterpret_cast<const T*>(a.v) + row0 * a.row_stride + head * 64;
  float qv[8];
  load8(q_cls + (int64_t)b * q_stride + head * 64, c, qv);
  const int nit = (L + 7) >> 3;
  float mx = -INFINITY;
#pragma unroll 4
  for (int i = 0; i < nit; ++i) {
    const int key = g + 8 * i;
    float dot = 0.f;
    if (key < L) {
      float kv[8];
      load8(kbase + (int64_t)key * a.row_stride, c, kv);
#pragma unroll
      for (int e = 0; e < 8; ++e) dot = fmaf(qv[e], kv[e], dot);
    }
    dot += __shfl_xor(dot, 1, 64);
    dot += __shfl_xor(dot, 2, 64);
    dot += __shfl_xor(dot, 4, 64);
    if (key < L) {
      const float x = fmaf(dot, a.scale, a.key_bias ? a.key_bias[row0 + key] : 0.f);
      mx = fmaxf(mx, x);
      if (c == 0) sc[key] = x;
    }
  }
  mx = wave_max(mx);
  __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the scores are in LDS (same wave wrote them)
  float sum = 0.f;
  for (int key = lane; key < L; key += 64) {
    const float p = expf(sc[key] - mx);
    sc[key] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll 4
  for (int i = 0; i < nit; ++i) {
    const int key = g + 8 * i;
    if (key < L) {
      float vv[8];
      load8(vbase + (int64_t)key * a.row_stride, c, vv);
      const float p = sc[key];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, vv[e], acc[e]);
    }
  }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = acc[e];
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    acc[e] = v * inv;
  }
  if (g == 0) store8(ctx_cls + (int64_t)b * ctx_stride + head * 64, c, acc);
}
}  // namespace

// q_cls: [B, q_stride] rows holding the CLS queries (heads side by side); k / v of `a` as usual; ctx_cls [B, ctx_stride]
int attention_cls_fwd(const AttnArgs& a, const void* q_cls, int64_t q_stride, void* ctx_cls, int64_t ctx_stride, int dtype,
                      hipStream_t stream) {
  EZ_REQUIRE(a.B > 0 && a.L > 0 && a.L <= 512 && a.H > 0 && a.causal == 0 && a.drop.thr == 0,
             "attention_cls_fwd: unsupported problem (L=%d)", a.L);
  const int esz = dtype_size(dtype);
  EZ_REQUIRE((a.row_stride * esz) % 16 == 0 && (q_stride * esz) % 16 == 0 && (ctx_stride * esz) % 16 == 0,
             "attention_cls_fwd: strides must be 16-byte multiples");
  EZ_REQUIRE(((uintptr_t)a.k % 16) == 0 && ((uintptr_t)a.v % 16) == 0 && ((uintptr_t)q_cls % 16) == 0 && ((uintptr_t)ctx_cls % 16) == 0,
             "attention_cls_fwd: pointers must be 16-byte aligned");
  const int waves = a.B * a.H;
  ProfScope ps(PROF_ATTN, 4.0 * a.B * a.H * (double)a.L * 64, stream);
  if (dtype == EZCLIP_F32)
    hipLaunchKernelGGL((attn_cls_fwd_kernel<float>), dim3((waves + 3) / 4), dim3(256), 0, stream, a, (const float*)q_cls, q_stride,
                       (float*)ctx_cls, ctx_stride);
  else
    hipLaunchKernelGGL((attn_cls_fwd_kernel<bf16_t>), dim3((waves + 3) / 4), dim3(256), 0, stream, a, (const bf16_t*)q_cls, q_stride,
                       (bf16_t*)ctx_cls, ctx_stride);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

namespace {
// Backward of attn_cls_fwd_kernel.  One wave per (sample, head), same lane = (key group, chunk) layout.  Writes the FULL
// dq / dk / dv blocks of the head: dk, dv for every key, dq for the query row (row 0 of the sample) and zeros for the
// other rows (they have no path to the loss) -- the in_proj gradients then run over all tokens as usual.
//   p = softmax(scale q.K^T + bias),  D = <dO, O>,  dP_k = <dO, V_k>,  dS_k = p_k (dP_k - D)
//   dV_k = p_k dO,   dK_k = scale dS_k q,   dq = scale sum_k dS_k K_k
template <typename T>
__global__ __launch_bounds__(256) void attn_cls_bwd_kernel(AttnBwdArgs a, const T* q_cls, int64_t q_stride, const T* ctx_cls,
                                                           const T* dctx_cls, int64_t ctx_stride, T* dq_cls, int64_t dq_stride) {
  __shared__ float sc_all[4][512];
  const AttnArgs& f = a.f;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int bh = blockIdx.x * 4 + w;
  if (bh >= f.B * f.H) return;
  float* sc = sc_all[w];
  const int b = bh / f.H, head = bh - b * f.H;
  const int L = f.lens ? f.lens[b] : f.L, g = lane >> 3, c = lane & 7;             // (packed batches: AttnArgs::cu / lens)
  const int64_t row0 = f.cu ? (int64_t)f.cu[b] : (int64_t)b * f.L;
  const int64_t hoff = row0 * f.row_stride + head * 64;
  const T* kbase = reinterpret_cast<const T*>(f.k) + hoff;
  const T* vbase = reinterpret_cast<const T*>(f.v) + hoff;
  T* dqb = a.dq ? reinterpret_cast<T*>(a.dq) + hoff : nullptr;
  T* dkb = reinterpret_cast<T*>(a.dk) + hoff;
  T* dvb = reinterpret_cast<T*>(a.dv) + hoff;
  float qv[8], ov[8], gv[8];
  load8(q_cls + (int64_t)b * q_stride + head * 64, c, qv);
  load8(ctx_cls + (int64_t)b * ctx_stride + head * 64, c, ov);
  load8(dctx_cls + (int64_t)b * ctx_stride + head * 64, c, gv);
  float D = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) D = fmaf(gv[e], ov[e], D);
  D += __shfl_xor(D, 1, 64);
  D += __shfl_xor(D, 2, 64);
  D += __shfl_xor(D, 4, 64);
  const int nit = (L + 7) >> 3;
  float mx = -INFINITY;
#pragma unroll 4
  for (int i = 0; i < nit; ++i) {
    const int key = g + 8 * i;
    float dot = 0.f;
    if (key < L) {
      float kv[8];
      load8(kbase + (int64_t)key * f.row_stride, c, kv);
#pragma unroll
      for (int e = 0; e < 8; ++e) dot = fmaf(qv[e], kv[e], dot);
    }
    dot += __shfl_xor(dot, 1, 64);
    dot += __shfl_xor(dot, 2, 64);
    dot += __shfl_xor(dot, 4, 64);
    if (key < L) {
      const float x = fmaf(dot, f.scale, f.key_bias ? f.key_bias[row0 + key] : 0.f);
      mx = fmaxf(mx, x);
      if (c == 0) sc[key] = x;
    }
  }
  mx = wave_max(mx);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  float sum = 0.f;
  for (int key = lane; key < L; key += 64) {
    const float p = expf(sc[key] - mx);
    sc[key] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  const float inv = 1.0f / sum;
  float dq[8], zero[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { dq[e] = 0.f; zero[e] = 0.f; }
#pragma unroll 2
  for (int i = 0; i < nit; ++i) {
    const int key = g + 8 * i;
    float dp = 0.f;
    float kv[8], vv[8];
    if (key < L) {
      load8(vbase + (int64_t)key * f.row_stride, c, vv);
      load8(kbase + (int64_t)key * f.row_stride, c, kv);
#pragma unroll
      for (int e = 0; e < 8; ++e) dp = fmaf(gv[e], vv[e], dp);
    }
    dp += __shfl_xor(dp, 1, 64);
    dp += __shfl_xor(dp, 2, 64);
    dp += __shfl_xor(dp, 4, 64);
    if (key < L) {
      const float p = sc[key] * inv;
      const float ds = p * (dp - D) * f.scale;
      float dk[8], dv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        dv[e] = p * gv[e];
        dk[e] = ds * qv[e];
        dq[e] = fmaf(ds, kv[e], dq[e]);
      }
      store8(dvb + (int64_t)key * f.row_stride, c, dv);
      store8(dkb + (int64_t)key * f.row_stride, c, dk);
      if (key > 0 && dq_cls == nullptr) store8(dqb + (int64_t)key * f.row_stride, c, zero);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = dq[e];
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    dq[e] = v;
  }
  // dq_cls given (BERT: the query projection itself ran on the CLS rows): compact [B, dq_stride] output, no zero fill
  if (g == 0) store8(dq_cls ? dq_cls + (int64_t)b * dq_stride + head * 64 : dqb, c, dq);
}
}  // namespace

int attention_cls_bwd(const AttnBwdArgs& a, const void* q_cls, int64_t q_stride, const void* ctx_cls, const void* dctx_cls,
                      int64_t ctx_stride, int dtype, hipStream_t stream, void* dq_cls, int64_t dq_stride) {
  const AttnArgs& f = a.f;
  EZ_REQUIRE(f.B > 0 && f.L > 0 && f.L <= 512 && f.H > 0 && f.causal == 0 && f.drop.thr == 0 && (a.dq || dq_cls) && a.dk && a.dv,
             "attention_cls_bwd: unsupported problem (L=%d)", f.L);
  const int esz = dtype_size(dtype);
  EZ_REQUIRE((f.row_stride * esz) % 16 == 0 && (q_stride * esz) % 16 == 0 && (ctx_stride * esz) % 16 == 0,
             "attention_cls_bwd: strides must be 16-byte multiples");
  const int waves = f.B * f.H;
  ProfScope ps(PROF_ATTN, 10.0 * f.B * f.H * (double)f.L * 64, stream);
  if (dtype == EZCLIP_F32)
    hipLaunchKernelGGL((attn_cls_bwd_kernel<float>), dim3((waves + 3) / 4), dim3(256), 0, stream, a, (const float*)q_cls, q_stride,
                       (const float*)ctx_cls, (const float*)dctx_cls, ctx_stride, (float*)dq_cls, dq_stride);
  else
    hipLaunchKernelGGL((attn_cls_bwd_kernel<bf16_t>), dim3((waves + 3) / 4), dim3(256), 0, stream, a, (const bf16_t*)q_cls, q_stride,
                       (const bf16_t*)ctx_cls, (const bf16_t*)dctx_cls, ctx_stride, (bf16_t*)dq_cls, dq_stride);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

static int g_attn_variant = -1;
void set_attention_variant(int v) { g_attn_variant = v; }

int attention_fwd(const AttnArgs& a, int dtype, hipStream_t stream) {
  EZ_REQUIRE(a.B > 0 && a.L > 0 && a.H > 0, "attention_fwd: empty problem");
  const int esz = dtype_size(dtype);
  EZ_REQUIRE((a.row_stride * esz) % 16 == 0 && (a.ctx_stride * esz) % 8 == 0, "attention_fwd: strides must be 16-byte multiples");
  EZ_REQUIRE(((uintptr_t)a.q % 16) == 0 && ((uintptr_t)a.k % 16) == 0 && ((uintptr_t)a.v % 16) == 0 &&
             ((uintptr_t)a.ctx % 16) == 0, "attention_fwd: pointers must be 16-byte aligned");
  EZ_REQUIRE(a.B <= 65535, "attention_fwd: batch %d > 65535", a.B);
  EZ_REQUIRE((a.cu == nullptr) == (a.lens == nullptr), "attention_fwd: cu and lens go together");
  // (the short forward kernel takes dropout with or without the keep-bit buffer; a fused backward needs the bits -- without
  // them, e.g. through the op-level entry points, the backward runs the general kernels, which regenerate the same decisions)
  if (a.cu != nullptr) {      // packed batches: the short forward kernel only (bf16, longest sample <= 288)
    EZ_REQUIRE(attention_short_fwd_eligible(a, dtype), "attention_fwd: packed batches need the bf16 short kernel (longest sample <= 288)");
    return attention_fwd_short(a, stream);
  }
  if (g_attn_variant != 0 && attention_short_fwd_eligible(a, dtype)) return attention_fwd_short(a, stream);
  if (dtype == EZCLIP_F32) return launch_fwd<float>(a, stream);
  if (dtype == EZCLIP_BF16) return launch_fwd<bf16_t>(a, stream);
  set_error("attention_fwd: bad dtype %d", dtype);
  return EZ_ERR_INVALID;
}

// ---------------------------------------------------------------------------------
// Backward (flash-style recompute from the saved log-sum-exp; nothing of size L x L is
// ever stored).  With S = scale*Q K^T + bias, P = exp(S - lse), D_q = <dO_q, O_q>:
//   dV = P^T dO      dP = dO V^T      dS = P o (dP - D)      dQ = scale dS K      dK = scale dS^T Q
// Two kernels, both built on the forward's "the lane already holds its own fragment" trick:
//   dQ  kernel: waves own 32-query blocks, sweep over key tiles; S^T and dP^T tiles leave the
//               MFMA with one query per lane, so dS^T feeds mfma(K^T frag, dS) directly.
//   dKV kernel: waves own 32-key blocks, sweep over query tiles; S and dP tiles leave the MFMA
//               with one key per lane, so P / dS feed mfma(dO^T frag, P) and mfma(Q^T frag, dS).
// The swept operand lives in LDS (row-major swizzled image by LDS-DMA + a transposed image),
// in chunks of up to CH tiles when L is long.
namespace {

template <typename T>
__device__ __forceinline__ void stage_rows_dma(char* dst, const char* gbase, int64_t rs, int row0, int nrows, int L,
                                               int wave, int nwaves, int lane) {
  using G = Geo<T>;
  const int ninst = nrows * G::RB / 1024;
  for (int inst = wave; inst < ninst; inst += nwaves) {
    const int r = inst * G::RPI + lane / G::CPR;
    const int c = (lane % G::CPR) ^ G::swz(r);
    int gr = row0 + r;
    gr = gr < L ? gr : L - 1;   // clamp: finite values, always multiplied by an exact zero downstream
    __builtin_amdgcn_global_load_lds((glb_void*)(gbase + gr * rs + c * 16), (lds_void*)(dst + inst * 1024), 16, 0, 0);
  }
}

// dst[d][LP] (row-contiguous) <- rows row0 .. row0+nrows of a [*, 64] matrix; rows >= L are zero
template <typename T>
__device__ __forceinline__ void stage_rows_T(char* dst, const char* gbase, int64_t rs, int row0, int nrows, int L,
                                             int LP, int tid, int nthreads) {
  using G = Geo<T>;
  const int nkq = nrows / 4;
  for (int idx = tid; idx < nkq * G::CPR; idx += nthreads) {
    const int dc = idx / nkq, kq = idx % nkq;
    uint4 w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 4 * kq + r;
      if (row < L) w[r] = *reinterpret_cast<const uint4*>(gbase + row * rs + dc * 16);
      else w[r] = make_uint4(0, 0, 0, 0);
    }
    const uint32_t w0[4] = {w[0].x, w[0].y, w[0].z, w[0].w};
    const uint32_t w1[4] = {w[1].x, w[1].y, w[1].z, w[1].w};
    const uint32_t w2[4] = {w[2].x, w[2].y, w[2].z, w[2].w};
    const uint32_t w3[4] = {w[3].x, w[3].y, w[3].z, w[3].w};
    if constexpr (G::SZ == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int sh = (e & 1) * 16;
        const uint32_t v0 = (w0[e >> 1] >> sh) & 0xffffu, v1 = (w1[e >> 1] >> sh) & 0xffffu;
        const uint32_t v2 = (w2[e >> 1] >> sh) & 0xffffu, v3 = (w3[e >> 1] >> sh) & 0xffffu;
        *reinterpret_cast<uint2*>(dst + ((8 * dc + e) * LP + 4 * kq) * 2) = make_uint2(v0 | (v1 << 16), v2 | (v3 << 16));
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        *reinterpret_cast<uint4*>(dst + ((4 * dc + e) * LP + 4 * kq) * 4) = make_uint4(w0[e], w1[e], w2[e], w3[e]);
    }
  }
}

// acc[2] (D[d][lane column]) += Xt-fragment . regs, where `v[16]` are this lane's 16 values of tile t
// (element r <-> inner index 32t + (r&3) + 8(r>>2) + 4h) and Xt is a [64][LP] transposed image.
template <typename T>
__device__ __forceinline__ void mma_T(f32x16_t (&acc)[2], const char* xt, int LP, int t, int l31, int h, const float (&v)[16]) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      uint4 pc;
      pc.x = pack_bf16x2(v[8 * u + 0], v[8 * u + 1]);
      pc.y = pack_bf16x2(v[8 * u + 2], v[8 * u + 3]);
      pc.z = pack_bf16x2(v[8 * u + 4], v[8 * u + 5]);
      pc.w = pack_bf16x2(v[8 * u + 6], v[8 * u + 7]);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const char* vp = xt + ((dt * 32 + l31) * LP + 32 * t + 16 * u + 4 * h) * 2;
        const uint2 lo = *reinterpret_cast<const uint2*>(vp);
        const uint2 hi = *reinterpret_cast<const uint2*>(vp + 16);
        mma32(acc[dt], make_uint4(lo.x, lo.y, hi.x, hi.y), pc, T());
      }
    }
  } else {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const uint4 pc = make_uint4(__float_as_uint(v[4 * qd + 0]), __float_as_uint(v[4 * qd + 1]),
                                  __float_as_uint(v[4 * qd + 2]), __float_as_uint(v[4 * qd + 3]));
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const uint4 vf = *reinterpret_cast<const uint4*>(xt + ((dt * 32 + l31) * LP + 32 * t + 8 * qd + 4 * h) * 4);
        mma32(acc[dt], vf, pc, T());
      }
    }
  }
}

// one 32x32 tile: D[image row][lane's fragment]  (image rows 32t.., lane fragment `f`)
template <typename T>
__device__ __forceinline__ void tile_rows_x_frag(const char* img, const uint4 (&f)[Geo<T>::NS], int t, int l31, int h,
                                                 f32x16_t& acc) {
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < Geo<T>::NS; ++s) mma32(acc, read_k<T>(img, 32 * t + l31, 2 * s + h), f[s], T());
}

template <typename T>
__device__ __forceinline__ void load_frag_row(const void* base, int64_t row_elems_off, int h, uint4 (&f)[Geo<T>::NS]) {
  const char* p = reinterpret_cast<const char*>(base) + row_elems_off * Geo<T>::SZ;
#pragma unroll
  for (int s = 0; s < Geo<T>::NS; ++s) f[s] = *reinterpret_cast<const uint4*>(p + (2 * s + h) * 16);
}

template <typename T>
__device__ __forceinline__ float frag_dot(const uint4 (&a)[Geo<T>::NS], const uint4 (&b)[Geo<T>::NS]) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < Geo<T>::NS; ++i) {
    float x[Elem<T>::kPerChunk], y[Elem<T>::kPerChunk];
    unpack_chunk(a[i], x, T());
    unpack_chunk(b[i], y, T());
#pragma unroll
    for (int e = 0; e < Elem<T>::kPerChunk; ++e) s += x[e] * y[e];
  }
  return s;
}

template <typename T> struct BwdSmemA {   // dQ kernel: K rm | V rm | K^T | key bias
  int rows, LP, vOff, ktOff, kbOff, bytes;
  __host__ __device__ explicit BwdSmemA(int ch) {
    rows = 32 * ch; LP = rows + 4;
    vOff = rows * Geo<T>::RB; ktOff = 2 * vOff; kbOff = ktOff + 64 * LP * Geo<T>::SZ; bytes = kbOff + rows * 4;
  }
};
template <typename T> struct BwdSmemB {   // dKV kernel: Q rm | dO rm | Q^T | dO^T | lse | delta
  int rows, LP, doOff, qtOff, dotOff, lseOff, dlOff, bytes;
  __host__ __device__ explicit BwdSmemB(int ch) {
    rows = 32 * ch; LP = rows + 4;
    doOff = rows * Geo<T>::RB; qtOff = 2 * doOff; dotOff = qtOff + 64 * LP * Geo<T>::SZ;
    lseOff = dotOff + 64 * LP * Geo<T>::SZ; dlOff = lseOff + rows * 4; bytes = dlOff + rows * 4;
  }
};

template <typename T>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(AttnBwdArgs a, int nt, int ch) {
  using G = Geo<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const BwdSmemA<T> S(ch);
  const AttnArgs& f = a.f;
  const int head = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthreads >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int L = f.L;
  constexpr bool kFast = IsFast<T>::value;
  const int64_t rs = f.row_stride * G::SZ;
  const int64_t hoff = ((int64_t)b * L * f.row_stride + head * 64) * G::SZ;
  const char* kbase = reinterpret_cast<const char*>(f.k) + hoff;
  const char* vbase = reinterpret_cast<const char*>(f.v) + hoff;

  const int qb = blockIdx.z * nwaves + wave;          // this wave's query block
  const int q = qb * 32 + l31;
  const bool active = qb * 32 < L;
  const int qc = q < L ? q : L - 1;
  uint4 qf[G::NS], dof[G::NS];
  float lse_q = 0.f, dq_delta = 0.f;
  if (active) {
    uint4 of[G::NS];
    load_frag_row<T>(f.q, ((int64_t)b * L + qc) * f.row_stride + head * 64, h, qf);
    load_frag_row<T>(a.dctx, ((int64_t)b * L + qc) * f.ctx_stride + head * 64, h, dof);
    load_frag_row<T>(f.ctx, ((int64_t)b * L + qc) * f.ctx_stride + head * 64, h, of);
    dq_delta = frag_dot<T>(dof, of);
    dq_delta += __shfl_xor(dq_delta, 32, 64);
    lse_q = f.lse[((int64_t)b * f.H + head) * L + qc];
  }
  f32x16_t acc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;

  float* kb = reinterpret_cast<float*>(smem + S.kbOff);
  for (int c0 = 0; c0 < nt; c0 += ch) {
    const int cht = min(ch, nt - c0);
    const int row0 = 32 * c0;
    __syncthreads();   // previous chunk fully consumed
    stage_rows_dma<T>(smem, kbase, rs, row0, 32 * cht, L, wave, nwaves, lane);
    stage_rows_dma<T>(smem + S.vOff, vbase, rs, row0, 32 * cht, L, wave, nwaves, lane);
    stage_rows_T<T>(smem + S.ktOff, kbase, rs, row0, 32 * cht, L, S.LP, tid, nthreads);
    for (int key = tid; key < 32 * cht; key += nthreads) {
      const int gk = row0 + key;
      kb[key] = gk < L ? (f.key_bias ? f.key_bias[(int64_t)b * L + gk] : 0.f) : -INFINITY;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (active) {
#pragma unroll 1
      for (int t = 0; t < cht; ++t) {
        float x[16];
        score_tile<T>(smem, kb, qf, t, l31, h, f.scale, x, f.causal, q, row0 + 32 * t);   // S^T[key][q]
        f32x16_t dp;
        tile_rows_x_frag<T>(smem + S.vOff, dof, t, l31, h, dp);              // dP^T[key][q] = V . dO^T
        float dpm[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dpm[r] = dp[r];
        // dropout: dP = mask/(1-p) o (dO V^T)   (D = <dO, O> already contains the mask through O)
        if (f.drop.thr != 0) drop_tile_qmajor(f.drop, ((uint32_t)b * f.H + head) * L + qc, c0 + t, h, dpm);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = kFast ? __expf(x[r] - lse_q) : expf(x[r] - lse_q);
          x[r] = p * (dpm[r] - dq_delta);                                     // dS^T
        }
        mma_T<T>(acc, smem + S.ktOff, S.LP, t, l31, h, x);                    // dQ^T[d][q] += K^T . dS
      }
    }
  }
  if (active && q < L) {
    T* dqp = reinterpret_cast<T*>(a.dq) + ((int64_t)b * L + q) * f.row_stride + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[dt][4 * qd + e] * f.scale;
        st4(dqp + dt * 32 + 8 * qd + 4 * h, v);
      }
  }
}

template <typename T>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(AttnBwdArgs a, int nt, int ch) {
  using G = Geo<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const BwdSmemB<T> S(ch);
  const AttnArgs& f = a.f;
  const int head = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthreads >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int L = f.L;
  constexpr bool kFast = IsFast<T>::value;
  const int64_t rs = f.row_stride * G::SZ;
  const int64_t cs = f.ctx_stride * G::SZ;
  const int64_t hoff = ((int64_t)b * L * f.row_stride + head * 64) * G::SZ;
  const int64_t coff = ((int64_t)b * L * f.ctx_stride + head * 64) * G::SZ;
  const char* qbase = reinterpret_cast<const char*>(f.q) + hoff;
  const char* dobase = reinterpret_cast<const char*>(a.dctx) + coff;
  const char* obase = reinterpret_cast<const char*>(f.ctx) + coff;

  const int kblk = blockIdx.z * nwaves + wave;        // this wave's key block
  const int key = kblk * 32 + l31;
  const bool active = kblk * 32 < L;
  const int kc = key < L ? key : L - 1;
  uint4 kf[G::NS], vf[G::NS];
  float kbias = -INFINITY;
  if (active) {
    load_frag_row<T>(f.k, ((int64_t)b * L + kc) * f.row_stride + head * 64, h, kf);
    load_frag_row<T>(f.v, ((int64_t)b * L + kc) * f.row_stride + head * 64, h, vf);
    if (key < L) kbias = f.key_bias ? f.key_bias[(int64_t)b * L + key] : 0.f;
  }
  f32x16_t accv[2], acck[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { accv[dt][r] = 0.f; acck[dt][r] = 0.f; }

  float* lse_s = reinterpret_cast<float*>(smem + S.lseOff);
  float* dl_s = reinterpret_cast<float*>(smem + S.dlOff);
  for (int c0 = 0; c0 < nt; c0 += ch) {
    const int cht = min(ch, nt - c0);
    const int row0 = 32 * c0;
    __syncthreads();
    stage_rows_dma<T>(smem, qbase, rs, row0, 32 * cht, L, wave, nwaves, lane);
    stage_rows_dma<T>(smem + S.doOff, dobase, cs, row0, 32 * cht, L, wave, nwaves, lane);
    stage_rows_T<T>(smem + S.qtOff, qbase, rs, row0, 32 * cht, L, S.LP, tid, nthreads);
    stage_rows_T<T>(smem + S.dotOff, dobase, cs, row0, 32 * cht, L, S.LP, tid, nthreads);
    for (int r = tid; r < 32 * cht; r += nthreads) {
      const int gq = row0 + r;
      float l = INFINITY, d = 0.f;
      if (gq < L) {
        l = f.lse[((int64_t)b * f.H + head) * L + gq];
        const char* po = obase + gq * cs;
        const char* pd = dobase + gq * cs;
#pragma unroll
        for (int c = 0; c < G::CPR; ++c) {
          float x[Elem<T>::kPerChunk], y[Elem<T>::kPerChunk];
          unpack_chunk(*reinterpret_cast<const uint4*>(po + c * 16), x, T());
          unpack_chunk(*reinterpret_cast<const uint4*>(pd + c * 16), y, T());
#pragma unroll
          for (int e = 0; e < Elem<T>::kPerChunk; ++e) d += x[e] * y[e];
        }
      }
      lse_s[r] = l;
      dl_s[r] = d;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (active) {
#pragma unroll 1
      for (int t = 0; t < cht; ++t) {
        f32x16_t sacc, dp;
        tile_rows_x_frag<T>(smem, kf, t, l31, h, sacc);                      // S[q][key]   = Q . K^T
        tile_rows_x_frag<T>(smem + S.doOff, vf, t, l31, h, dp);              // dP[q][key]  = dO . V^T
        float p[16], ds[16], dm[16];
        const bool dropping = f.drop.thr != 0;
        if (dropping) drop_factors_kmajor(f.drop, ((uint32_t)b * f.H + head) * L, row0 + 32 * t, h, (uint32_t)kc, dm);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 32 * t + 8 * qd + 4 * h);
          const float4 d4 = *reinterpret_cast<const float4*>(dl_s + 32 * t + 8 * qd + 4 * h);
          const float lv[4] = {l4.x, l4.y, l4.z, l4.w};
          const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * qd + e;
            const float x = fmaf(sacc[r], f.scale, kbias) - lv[e];
            p[r] = kFast ? __expf(x) : expf(x);
            if (f.causal && key > row0 + 32 * t + 8 * qd + 4 * h + e) p[r] = 0.f;       // (query before this key)
            if (dropping) {
              ds[r] = p[r] * (dp[r] * dm[r] - dv[e]);
              p[r] *= dm[r];                                                   // dV takes the dropped probabilities
            } else {
              ds[r] = p[r] * (dp[r] - dv[e]);
            }
          }
        }
        mma_T<T>(accv, smem + S.dotOff, S.LP, t, l31, h, p);                  // dV^T[d][key] += dO^T . P
        mma_T<T>(acck, smem + S.qtOff, S.LP, t, l31, h, ds);                  // dK^T[d][key] += Q^T . dS
      }
    }
  }
  if (active && key < L) {
    T* dkp = reinterpret_cast<T*>(a.dk) + ((int64_t)b * L + key) * f.row_stride + head * 64;
    T* dvp = reinterpret_cast<T*>(a.dv) + ((int64_t)b * L + key) * f.row_stride + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        float v[4], k[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = accv[dt][4 * qd + e]; k[e] = acck[dt][4 * qd + e] * f.scale; }
        st4(dvp + dt * 32 + 8 * qd + 4 * h, v);
        st4(dkp + dt * 32 + 8 * qd + 4 * h, k);
      }
  }
}

template <typename T>
int launch_bwd(const AttnBwdArgs& a, hipStream_t stream) {
  const int nt = (a.f.L + 31) / 32;
  const int budget = 150 * 1024;
  int chA = nt, chB = nt;
  while (chA > 1 && BwdSmemA<T>(chA).bytes > budget) --chA;
  while (chB > 1 && BwdSmemB<T>(chB).bytes > budget) --chB;
  const int ldsA = BwdSmemA<T>(chA).bytes, ldsB = BwdSmemB<T>(chB).bytes;
  static LdsOptIn optA, optB;
  EZ_ENSURE_LDS((&attn_bwd_dq_kernel<T>), optA, ldsA);
  EZ_ENSURE_LDS((&attn_bwd_dkv_kernel<T>), optB, ldsB);
  const int nw = nt < 8 ? nt : 8;
  const int gz = (nt + nw - 1) / nw;
  ProfScope ps(PROF_ATTN, 10.0 * a.f.B * a.f.H * (double)a.f.L * a.f.L * 64, stream);   // 5 L x L x 64 products
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T>), dim3(a.f.H, a.f.B, gz), dim3(nw * 64), ldsA, stream, a, nt, chA);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<T>), dim3(a.f.H, a.f.B, gz), dim3(nw * 64), ldsB, stream, a, nt, chB);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace

int attention_bwd(const AttnBwdArgs& a, int dtype, hipStream_t stream) {
  const AttnArgs& f = a.f;
  EZ_REQUIRE(f.B > 0 && f.L > 0 && f.H > 0, "attention_bwd: empty problem");
  EZ_REQUIRE(f.q && f.k && f.v && f.ctx && f.lse && a.dctx && a.dq && a.dk && a.dv, "attention_bwd: null tensor");
  const int esz = dtype_size(dtype);
  EZ_REQUIRE((f.row_stride * esz) % 16 == 0 && (f.ctx_stride * esz) % 16 == 0, "attention_bwd: strides must be 16-byte multiples");
  EZ_REQUIRE(f.B <= 65535, "attention_bwd: batch %d > 65535", f.B);
  EZ_REQUIRE((a.dbq == nullptr) == (a.dbk == nullptr) && (a.dbq == nullptr) == (a.dbv == nullptr), "attention_bwd: dbq/dbk/dbv must be given together");
  const bool short_drop_ok = f.drop.thr == 0 || (f.keep_bits != nullptr && f.L <= 256);     // (see attention_fwd)
  if (f.cu != nullptr) {      // packed batches: the fused short kernel only
    EZ_REQUIRE(f.lens != nullptr && short_drop_ok && attention_short_eligible(f, dtype) && (a.dbq == nullptr || a.db_part != nullptr),
               "attention_bwd: packed batches need the fused bf16 kernel (longest sample <= 272; with dropout <= 256 and keep_bits)");
    return attention_bwd_short(a, stream);
  }
  if (g_attn_variant != 0 && short_drop_ok && attention_short_eligible(f, dtype) && (a.dbq == nullptr || a.db_part != nullptr))
    return attention_bwd_short(a, stream);
  int rc;
  if (dtype == EZCLIP_F32) rc = launch_bwd<float>(a, stream);
  else if (dtype == EZCLIP_BF16) rc = launch_bwd<bf16_t>(a, stream);
  else { set_error("attention_bwd: bad dtype %d", dtype); return EZ_ERR_INVALID; }
  if (rc != EZ_OK || a.dbq == nullptr) return rc;
  // general kernels: the bias gradients are a separate column-sum pass over dq / dk / dv
  const int D = f.H * 64, M = f.B * f.L;
  if ((rc = colsum_add(a.dq, f.row_stride, M, D, a.dbq, dtype, stream)) != EZ_OK) return rc;
  if ((rc = colsum_add(a.dk, f.row_stride, M, D, a.dbk, dtype, stream)) != EZ_OK) return rc;
  return colsum_add(a.dv, f.row_stride, M, D, a.dbv, dtype, stream);
}

}  // namespace ezclip
