// Short-sequence (L <= 256), head_dim 64, bf16 self-attention for gfx950 -- the shapes of this path:
// ViT-B/16 L = 197, ViT-L/14 L = 257 (handled by attention.hip), BERT L <= 64..256.
//
// Reference semantics (same as attention.hip): ViT nn.MultiheadAttention without mask
// (modeling_chineseclip.py:188,198-200); BertSelfAttention with the additive (1-mask)*-10000 key bias
// (bert/modeling_bert.py:210-244).
//
// At these lengths attention is HBM-bound (98 flop per byte of q/k/v/ctx against a machine balance of ~400), so
// the kernels are built to stream: one workgroup per (sample, head), K and V of the head brought into LDS by
// LDS-DMA in their natural row-major layout (128-byte rows, full lines from the packed qkv buffer), ONE pass over
// the keys with an online softmax, and no register staging at all:
//   S^T tile = mfma(K frag, Q frag)      K image swizzled (chunk ^ (row>>1)&7) for ds_read_b128
//   O^T     += mfma(V^T frag, P)         V^T fragments come straight out of the ROW-MAJOR V image through
//                                        ds_read_b64_tr_b16 (the CDNA4 LDS transpose read): the 64-byte half of an
//                                        image row is XORed with (row>>1)&1, which puts the four rows of a
//                                        transpose block on four different bank quarters.
// The lane that owns query q after mfma(K, Q) holds 16 of each tile's 32 scores (its partner lane^32 the others):
// row max / sum are in-lane reductions plus one cross-lane exchange, and the bf16 P fragment of the PV product is
// the lane's own registers.  ~58 KiB of LDS per workgroup: two workgroups per CU overlap one head's load with the
// other's math.
#include "ezclip_common.h"
#include "kernels.h"

namespace ezclip {
namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;

__device__ __forceinline__ uint2 tr4(const char* p) {
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p));
}

// rows [0, 32*nt) of a [L, 64] bf16 matrix (row stride rs bytes) -> LDS image of 128-byte rows; 16-byte chunk c of
// row r lands at chunk position c ^ swz(r).  Rows >= L repeat row L-1 (finite; always multiplied by an exact 0).
template <int MODE>   // 0: K image (chunk ^ (r>>1)&7)   1: V image (64-byte half ^ (r>>1)&1)
__device__ __forceinline__ void dma_rows(char* dst, const char* gbase, int64_t rs, int nrows, int L, int wave, int nwaves,
                                         int lane) {
  const int ninst = nrows / 8;
  for (int inst = wave; inst < ninst; inst += nwaves) {
    const int r = inst * 8 + (lane >> 3);
    const int c = MODE == 0 ? ((lane & 7) ^ ((r >> 1) & 7)) : ((lane & 7) ^ (((r >> 1) & 1) << 2));
    const int gr = r < L ? r : L - 1;
    __builtin_amdgcn_global_load_lds((glb_void*)(gbase + gr * rs + c * 16), (lds_void*)(dst + inst * 1024), 16, 0, 0);
  }
}

// max / sum of a value with the one held by lane ^ 32 (the other half of the same query's scores): one
// v_permlane32_swap instead of an LDS round trip (ds_bpermute)
__device__ __forceinline__ void swap_halves(float v, float& x0, float& x1) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  x0 = __uint_as_float(r[0]);      // lanes < 32: own value / lanes >= 32: the partner's
  x1 = __uint_as_float(r[1]);      // lanes < 32: the partner's / lanes >= 32: own value
}

// (up to 9 waves = 288 tokens: ViT-L/14 has 257; 104 VGPRs leave room for four waves on a SIMD)
//
// The kernel is VALU-bound, not HBM-bound: streaming the K / V tiles under the math (round 2 experiment) did not move
// it, the per-score instruction count does.  So the softmax runs in the base-2 domain with the scale folded into the
// first (packed) multiply (x = s c, c = scale log2 e; p = 2^(x - m): sub + v_exp_f32 instead of sub, mul, v_exp_f32 -- and
// every consumer of the MFMA result is an ordinary instruction: an inline-asm v_max3 there slipped past hipcc's MFMA
// read-after-write wait states and read the accumulators early), the row sums
// use four independent partial sums, the running output is only rescaled when some lane's maximum moved
// (multiplying by exactly 1 otherwise), the cross-half exchanges are v_permlane32_swap, and every wave has queries
// (64 * nt threads).  HAS_KB: additive key bias (BERT); without it the scores never touch LDS.
// DROP: train-mode dropout on the probabilities (dropout.h).  The lane that owns query q holds keys 8 qd + 4 h + {0..3} of a
// tile in four consecutive registers = the four words of ONE Philox call (colquad = 8 t + 2 qd + h): four calls per tile and
// lane.  The row sum l runs over the undropped exponentials, the P V product over the kept ones scaled by 1 / (1 - p); the
// keep bits of the tile (32 keys, both half-waves) are written out for the backward kernel (AttnArgs::keep_bits).
template <bool HAS_KB, bool CAUSAL, bool DROP>
__global__ __launch_bounds__(576) void attn_fwd_short_kernel(AttnArgs a, int nt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int head = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  // packed batches (AttnArgs::cu / lens): this sample's rows start at cu[b] and there are lens[b] of them; nt (the launch's
  // tile count, LDS sizing, waves) belongs to the longest sample, this workgroup walks its own ceil(L / 32) tiles
  const int L = a.lens ? a.lens[b] : a.L, LKP = 32 * nt;
  const int64_t row0 = a.cu ? (int64_t)a.cu[b] : (int64_t)b * a.L;
  const int keep_words = nt;      // (of the launch)
  nt = (L + 31) >> 5;
  char* kimg = smem;
  char* vimg = smem + LKP * 128;
  float* kb = reinterpret_cast<float*>(smem + 2 * LKP * 128);
  const int64_t rs = a.row_stride * 2;
  const int64_t base = (row0 * a.row_stride + head * 64) * 2;
  constexpr float kLog2e = 1.4426950408889634f;

  const int nwaves = (int)(blockDim.x >> 6);
  dma_rows<0>(kimg, reinterpret_cast<const char*>(a.k) + base, rs, 32 * nt, L, wave, nwaves, lane);
  dma_rows<1>(vimg, reinterpret_cast<const char*>(a.v) + base, rs, 32 * nt, L, wave, nwaves, lane);
  if constexpr (HAS_KB) {      // key bias in base-2 units; keys >= L: -inf
    for (int key = tid; key < 32 * nt; key += (int)blockDim.x)
      kb[key] = key < L ? a.key_bias[row0 + key] * kLog2e : -INFINITY;
  }

  // this wave's 32 queries
  const int qb = wave;
  const bool active = qb * 32 < L;
  const int q = qb * 32 + l31;
  const int qc = q < L ? q : L - 1;
  uint4 qf[4];
  if (active) {
    const char* qp = reinterpret_cast<const char*>(a.q) + base + (int64_t)qc * rs;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const uint4*>(qp + (2 * s + h) * 16);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (!active) return;

  // per-lane LDS offsets
  const int sw = (l31 >> 1) & 7;
  uint32_t koff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = (uint32_t)l31 * 128u + ((uint32_t)((2 * s + h) ^ sw) << 4);
  const int t16 = lane & 15, sub = (lane >> 4) & 1;
  const uint32_t vsw = (uint32_t)((t16 >> 3) & 1);
  const uint32_t voff0 = (uint32_t)(4 * h + (t16 >> 2)) * 128u + (vsw << 6) + (uint32_t)sub * 32u + (uint32_t)(t16 & 3) * 8u;
  const uint32_t voff1 = voff0 ^ 64u;     // d tile 1

  f32x16_t o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m = -INFINITY, l = 0.f;      // running maximum of the base-2 scores s c (+ kb log2 e), running sum of 2^(x - m)
  const float c = a.scale * kLog2e;
  // (round 3) plain instantiation: the row sums come out of the MATRIX pipe -- P^T . 1 as one more product per 16 keys against a
  // fragment of ones (every row of the result is the row sum of this lane's query over the keys of BOTH half-waves) -- instead
  // of 16 dependent-chain adds per tile on the vector pipe, which is what bounds this kernel; the matrix pipe is ~85 % idle.
  // The sum then runs over the bf16-rounded probabilities, i.e. exactly the weights that multiply V.
  const bool mfma_sum = (!HAS_KB && !CAUSAL && !DROP) && a.mfma_rowsum;
  f32x16_t lacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
  const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  // A last tile with at most 8 keys (ViT-B/16: 197 = 6 * 32 + 5) keeps them in registers 0..3 of both half-waves (keys
  // 4 h + r): it gets its own, short epilogue below -- a quarter of the softmax instructions and half of the P V products of a
  // full tile (round 3; the generic loop would run a full tile's 80 vector instructions for five keys).
  constexpr bool kPlain = !HAS_KB && !CAUSAL && !DROP;
  const int rem = L & 31;
  const bool short_tail = kPlain && rem >= 1 && rem <= 8 && a.short_tail;
  const int nt_loop = short_tail ? nt - 1 : nt;
#pragma unroll 1
  for (int t = 0; t < nt_loop; ++t) {
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const char* kt = kimg + t * 4096;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint4 kf = *reinterpret_cast<const uint4*>(kt + koff[s]);
      mma32(acc, kf, qf[s], bf16_t());                       // D[key][q]
    }
    float x[16];
    if constexpr (HAS_KB) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 kb4 = *reinterpret_cast<const float4*>(kb + 32 * t + 8 * qd + 4 * h);
        x[4 * qd + 0] = fmaf(acc[4 * qd + 0], c, kb4.x);
        x[4 * qd + 1] = fmaf(acc[4 * qd + 1], c, kb4.y);
        x[4 * qd + 2] = fmaf(acc[4 * qd + 2], c, kb4.z);
        x[4 * qd + 3] = fmaf(acc[4 * qd + 3], c, kb4.w);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] = acc[r] * c;
      if (32 * (t + 1) > L) {                                 // the ragged last tile: keys >= L do not exist
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (32 * t + (r & 3) + 8 * (r >> 2) + 4 * h >= L) x[r] = -INFINITY;
      }
    }
    if constexpr (CAUSAL) {     // (a template parameter: the index arithmetic is otherwise hoisted into the common path)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (32 * t + (r & 3) + 8 * (r >> 2) + 4 * h > q) x[r] = -INFINITY;
    }
    float tmax = x[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, x[r]);
    float t0, t1;
    swap_halves(tmax, t0, t1);
    const float mn = fmaxf(m, fmaxf(t0, t1));                 // finite from the first tile on (key 0 < L)
    const float alpha = __builtin_amdgcn_exp2f(m - mn);       // (m = -inf on the first tile: 2^-inf = 0)
    if (mfma_sum) {
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_exp2f(x[r] - mn);
    } else {
      float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        x[r] = __builtin_amdgcn_exp2f(x[r] - mn);
        ps[r & 3] += x[r];
      }
      l = fmaf(l, alpha, (ps[0] + ps[1]) + (ps[2] + ps[3]));
    }
    m = mn;
    if (t > 0 && __builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {     // (multiplying by exactly 1 changes nothing)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      lacc[0] *= alpha;             // (the other rows of lacc hold the same sum and are never read)
    }
    if constexpr (DROP) {
      const uint32_t drow = (uint32_t)((b * a.H + head) * (a.drop_L > 0 ? a.drop_L : a.L) + q);
      uint32_t bits = 0;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const uint4 w = drop_words(a.drop, drow, (uint32_t)(8 * t + 2 * qd + h));
        const uint32_t wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool keep = wv[e] >= a.drop.thr;
          x[4 * qd + e] = keep ? x[4 * qd + e] * a.drop.scale : 0.f;
          bits |= (keep ? 1u : 0u) << (8 * qd + 4 * h + e);
        }
      }
      if (a.keep_bits != nullptr) {
        const auto r2 = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);     // own | the other half-wave's
        bits = r2[0] | r2[1];
        if (h == 0 && q < L) a.keep_bits[((row0 + q) * a.H + head) * keep_words + t] = bits;
      }
    }
    const char* vt = vimg + t * 4096;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      uint4 pc;
      pc.x = pack_bf16x2(x[8 * u + 0], x[8 * u + 1]);
      pc.y = pack_bf16x2(x[8 * u + 2], x[8 * u + 3]);
      pc.z = pack_bf16x2(x[8 * u + 4], x[8 * u + 5]);
      pc.w = pack_bf16x2(x[8 * u + 6], x[8 * u + 7]);
      // V^T fragments: keys 32t + 16u + 4h + {0..3} and + 8 + {0..3} for d = dt*32 + l31
      const uint2 a0 = tr4(vt + u * 2048 + voff0), a1 = tr4(vt + u * 2048 + 1024 + voff0);
      const uint2 b0 = tr4(vt + u * 2048 + voff1), b1 = tr4(vt + u * 2048 + 1024 + voff1);
      mma32(o[0], make_uint4(a0.x, a0.y, a1.x, a1.y), pc, bf16_t());   // D[d][q]
      mma32(o[1], make_uint4(b0.x, b0.y, b1.x, b1.y), pc, bf16_t());
      if (mfma_sum) mma32(lacc, ones, pc, bf16_t());                   // D[.][q] += sum over the 16 keys of p
    }
  }
  if constexpr (kPlain) {
    if (short_tail) {
      const int t = nt - 1;
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const char* kt = kimg + t * 4096;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint4 kf = *reinterpret_cast<const uint4*>(kt + koff[s]);
        mma32(acc, kf, qf[s], bf16_t());
      }
      float x[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = (4 * h + r < rem) ? acc[r] * c : -INFINITY;
      const float tmax = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
      float t0, t1;
      swap_halves(tmax, t0, t1);
      const float mn = fmaxf(m, fmaxf(t0, t1));                 // (key 32 t exists: finite)
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = __builtin_amdgcn_exp2f(x[r] - mn);
      m = mn;
      if (!mfma_sum) l = fmaf(l, alpha, (x[0] + x[1]) + (x[2] + x[3]));
      if (nt > 1 && __builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        lacc[0] *= alpha;
      }
      const char* vt = vimg + t * 4096;
      const uint4 pc = make_uint4(pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), 0u, 0u);   // keys 8 .. 31 of the tile: p = 0
      const uint2 a0 = tr4(vt + voff0), a1 = tr4(vt + 1024 + voff0);
      const uint2 b0 = tr4(vt + voff1), b1 = tr4(vt + 1024 + voff1);
      mma32(o[0], make_uint4(a0.x, a0.y, a1.x, a1.y), pc, bf16_t());
      mma32(o[1], make_uint4(b0.x, b0.y, b1.x, b1.y), pc, bf16_t());
      if (mfma_sum) mma32(lacc, ones, pc, bf16_t());
    }
  }
  if (mfma_sum) {
    l = lacc[0];                     // already the sum over both half-waves' keys
  } else {
    float l0, l1;
    swap_halves(l, l0, l1);
    l = l0 + l1;
  }
  const float inv = __builtin_amdgcn_rcpf(l);
  // lane (q, h) holds columns dt*32 + 8 qd + 4 h + {0..3}; after the swap of (qd, qd + 1) pairs it holds 8 consecutive
  // columns dt*32 + 16 j + 8 h + {0..7}: one 16-byte store (4 instead of 16 store instructions per wave)
  uint32_t pk[2][2][4];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      uint32_t lo[2], hi[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        lo[e] = pack_bf16x2(o[dt][4 * (2 * j) + 2 * e] * inv, o[dt][4 * (2 * j) + 2 * e + 1] * inv);
        hi[e] = pack_bf16x2(o[dt][4 * (2 * j + 1) + 2 * e] * inv, o[dt][4 * (2 * j + 1) + 2 * e + 1] * inv);
        const auto r2 = __builtin_amdgcn_permlane32_swap(lo[e], hi[e], false, false);
        lo[e] = r2[0]; hi[e] = r2[1];
      }
      pk[dt][j][0] = lo[0]; pk[dt][j][1] = lo[1]; pk[dt][j][2] = hi[0]; pk[dt][j][3] = hi[1];
    }
  // Round 5 (profiles/r5_attention_fullline_ab.log): the register-layout stores below cover 32 rows x 32 bytes per wave instruction -- every
  // 128-byte line of the output in FOUR pieces, from lanes that are not neighbours, and the write path merges neighbouring lanes only.
  // With a.fullline_store the wave's 32 x 64 tile goes through its own 4 KiB of the (dead) K image -- after a barrier: every wave reads
  // every K tile -- and leaves as 8 full rows per store instruction: ViT-B/16 0.346 -> 0.333 ms, BERT (64 tokens) 0.090 -> 0.082 ms per
  // layer.  The barrier costs more than the lines win once nine waves wait for each other (257 tokens: 0.400 -> 0.45 ms), so the
  // launcher sets the flag for <= 7 key tiles only.
  if (a.fullline_store) {               // (uniform: a property of the launch; waves that returned early do not count at the barrier)
    __syncthreads();
    char* tile = kimg + qb * 4096;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t cidx = (uint32_t)(4 * dt + 2 * j + h);
        *reinterpret_cast<uint4*>(tile + l31 * 128 + ((cidx ^ (uint32_t)(l31 & 7)) << 4)) =
            make_uint4(pk[dt][j][0], pk[dt][j][1], pk[dt][j][2], pk[dt][j][3]);
      }
    // (wave-private tile: the wave's own ds_write / ds_read are ordered by lgkmcnt, no barrier)
    const int cr = lane >> 3, cc = lane & 7;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int r = ps * 8 + cr;
      const uint4 v = *reinterpret_cast<const uint4*>(tile + r * 128 + (((uint32_t)cc ^ (uint32_t)(r & 7)) << 4));
      const int qq = qb * 32 + r;
      if (qq < L) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.ctx) + (row0 + qq) * a.ctx_stride + head * 64 + cc * 8) = v;
    }
    if (q < L && a.lse != nullptr && h == 0) a.lse[((int64_t)b * a.H + head) * a.L + q] = m * 0.6931471805599453f + logf(l);
    return;
  }
  if (q < L) {
    bf16_t* cp = reinterpret_cast<bf16_t*>(a.ctx) + (row0 + q) * a.ctx_stride + head * 64;
    if ((a.ctx_stride & 7) == 0 && ((uintptr_t)a.ctx & 15) == 0) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          *reinterpret_cast<uint4*>(cp + dt * 32 + 16 * j + 8 * h) = make_uint4(pk[dt][j][0], pk[dt][j][1], pk[dt][j][2], pk[dt][j][3]);
    } else {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          *reinterpret_cast<uint2*>(cp + dt * 32 + 16 * j + 8 * h) = make_uint2(pk[dt][j][0], pk[dt][j][1]);
          *reinterpret_cast<uint2*>(cp + dt * 32 + 16 * j + 8 * h + 4) = make_uint2(pk[dt][j][2], pk[dt][j][3]);
        }
    }
    // log-sum-exp of the scaled scores (natural log), as the backward kernels expect it
    if (a.lse != nullptr && h == 0) a.lse[((int64_t)b * a.H + head) * a.L + q] = m * 0.6931471805599453f + logf(l);
  }
}


}  // namespace

bool attention_short_fwd_eligible(const AttnArgs& a, int dtype) {   // forward: two images; one wave per 32 queries, <= 9 waves
  return dtype == EZCLIP_BF16 && a.L <= 288 && a.B <= 65535;
}

static int g_attn_short_tail = 3;      // ezclip_debug_set(9, v): bit 0 short last tile, bit 1 row sums on the matrix pipe, bit 2 NO full-line stores (A/B)
void set_attention_short_tail(int v) { g_attn_short_tail = v & 7; }

int attention_fwd_short(const AttnArgs& a_in, hipStream_t stream) {
  AttnArgs a = a_in;
  a.short_tail = g_attn_short_tail & 1;
  a.mfma_rowsum = (g_attn_short_tail >> 1) & 1;
  const int nt = (a.L + 31) / 32;
  a.fullline_store = (g_attn_short_tail >> 2 & 1) == 0 && nt <= 7 && (a.ctx_stride & 7) == 0 && ((uintptr_t)a.ctx & 15) == 0;
  const int bytes = nt * (2 * 32 * 128 + 32 * 4);
  static LdsOptIn lds_opt[8];
  const int kbi = (a.key_bias != nullptr ? 1 : 0) + (a.causal ? 2 : 0) + (a.drop.thr != 0 ? 4 : 0);
  EZ_REQUIRE(a.drop.thr == 0 || a.keep_bits == nullptr || a.keep_words == nt, "attention_fwd_short: keep_words must be ceil(L / 32)");
  using K = void (*)(AttnArgs, int);
  static const K kerns[8] = {&attn_fwd_short_kernel<false, false, false>, &attn_fwd_short_kernel<true, false, false>,
                             &attn_fwd_short_kernel<false, true, false>,  &attn_fwd_short_kernel<true, true, false>,
                             &attn_fwd_short_kernel<false, false, true>,  &attn_fwd_short_kernel<true, false, true>,
                             &attn_fwd_short_kernel<false, true, true>,   &attn_fwd_short_kernel<true, true, true>};
  const K kern = kerns[kbi];
  EZ_ENSURE_LDS(kern, lds_opt[kbi], bytes);
  {
    ProfScope ps(PROF_ATTN, 4.0 * a.B * a.H * (double)a.L * a.L * 64, stream);   // QK^T + PV, unpadded
    hipLaunchKernelGGL(kern, dim3(a.H, a.B), dim3(64 * nt), bytes, stream, a, nt);
  }
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace ezclip
