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

__global__ __launch_bounds__(512) void attn_fwd_short_kernel(AttnArgs a, int nt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int head = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int L = a.L, LKP = 32 * nt;
  char* kimg = smem;
  char* vimg = smem + LKP * 128;
  float* kb = reinterpret_cast<float*>(smem + 2 * LKP * 128);
  const int64_t rs = a.row_stride * 2;
  const int64_t base = ((int64_t)b * L * a.row_stride + head * 64) * 2;

  dma_rows<0>(kimg, reinterpret_cast<const char*>(a.k) + base, rs, LKP, L, wave, 8, lane);
  dma_rows<1>(vimg, reinterpret_cast<const char*>(a.v) + base, rs, LKP, L, wave, 8, lane);
  for (int key = tid; key < LKP; key += 512)
    kb[key] = key < L ? (a.key_bias ? a.key_bias[(int64_t)b * L + key] : 0.f) : -INFINITY;

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
  float m = -INFINITY, l = 0.f;
  const float scale = a.scale;
#pragma unroll 1
  for (int t = 0; t < nt; ++t) {
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
    float tmax = -INFINITY;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const float4 kb4 = *reinterpret_cast<const float4*>(kb + 32 * t + 8 * qd + 4 * h);
      x[4 * qd + 0] = fmaf(acc[4 * qd + 0], scale, kb4.x);
      x[4 * qd + 1] = fmaf(acc[4 * qd + 1], scale, kb4.y);
      x[4 * qd + 2] = fmaf(acc[4 * qd + 2], scale, kb4.z);
      x[4 * qd + 3] = fmaf(acc[4 * qd + 3], scale, kb4.w);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, x[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mn = fmaxf(m, tmax);                          // finite from the first tile on (key 0 < L)
    const float alpha = __expf(m - mn);
    m = mn;
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      x[r] = __expf(x[r] - mn);
      ps += x[r];
    }
    l = l * alpha + ps;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
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
    }
  }
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  if (q < L) {
    bf16_t* cp = reinterpret_cast<bf16_t*>(a.ctx) + ((int64_t)b * L + q) * a.ctx_stride + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o[dt][4 * qd + e] * inv;
        st4(cp + dt * 32 + 8 * qd + 4 * h, v);
      }
    if (a.lse != nullptr && h == 0) a.lse[((int64_t)b * a.H + head) * L + q] = m + logf(l);
  }
}

}  // namespace

bool attention_short_eligible(const AttnArgs& a, int dtype) {
  return dtype == EZCLIP_BF16 && a.L <= 256 && a.B <= 65535;
}

int attention_fwd_short(const AttnArgs& a, hipStream_t stream) {
  const int nt = (a.L + 31) / 32;
  const int bytes = nt * (2 * 32 * 128 + 32 * 4);
  static int attr_max = 0;
  if (bytes > attr_max) {
    EZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_short_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    attr_max = bytes;
  }
  {
    ProfScope ps(PROF_ATTN, 4.0 * a.B * a.H * (double)a.L * a.L * 64, stream);   // QK^T + PV, unpadded
    hipLaunchKernelGGL(attn_fwd_short_kernel, dim3(a.H, a.B), dim3(512), bytes, stream, a, nt);
  }
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace ezclip
