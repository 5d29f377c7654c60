// STAGED EXPERIMENT (round 2, written without GPU access; never run -- see tools/experiments/README.md).  NOT part of the library.
//
// Fused short-sequence attention backward with TWO workgroups per CU.  What the counters say about the shipped kernel
// (profiles/r2_attention_pmc.md): a wave is parked a third of its cycles (mostly the prologue: five tensors of
// the head land before the first MFMA) and nothing else runs on the CU meanwhile, because four images = 130 KB of LDS allow
// one workgroup per CU.  Here a workgroup keeps TWO images at a time and its waves take two row blocks each:
//
//   phase A   K and V images in LDS.  For each own block: the Q block through a 4-KiB per-wave staging image (DMA), the dO
//             and O rows straight from global memory; D and lse; pass A (dQ); r_q = sum_k dS for the key-bias gradient.
//   barrier   then the Q and dO images replace K and V.
//   phase B   for each own block: the K block through the staging image, the V rows from global memory; pass B (dK, dV);
//             the three bias-gradient products (K^T c from the staging image, Q^T r and dO^T 1 from the images).
//
// LDS at 197 tokens: 2 x 25.6 KB images + 16 KB staging + 4.5 KB row arrays + 7 KB bias areas = 79 KB -> two workgroups per CU
// (4 waves each, two per SIMD as before, but in different phases: one's loads and store tail under the other's math).
// Same arithmetic and the same fragment layouts as attention_short_bwd.hip (no dropout, no causal mask in this experiment).
//
// Harness: `attn_bwd_2wg [batch=1024]` runs the library's forward and fused backward, then this kernel on the same inputs,
// prints max |difference| of dq / dk / dv and of the bias gradients, and the times of both.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "../../easynlp_amd/csrc/ezclip_common.h"
#include "../../easynlp_amd/csrc/kernels.h"

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
  } while (0)

namespace x2 {
using namespace ezclip;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;

__device__ __forceinline__ uint2 tr4(const char* p) {
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p));
}
__device__ __forceinline__ int swz_f(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

// rows [row_first, row_first + nrows) of a [L, 64] bf16 matrix -> LDS image rows 0 .. nrows - 1 (image row i = matrix row
// row_first + i; the swizzle of row i is that of row_first + i because row_first is a multiple of 32)
__device__ __forceinline__ void dma_rows(char* dst, const char* gbase, int64_t rs, int row_first, int nrows, int L, int wave,
                                         int nwaves, int lane) {
  const int ninst = nrows / 8;
  for (int inst = wave; inst < ninst; inst += nwaves) {
    const int r = inst * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz_f(r);
    const int gr = row_first + r < L ? row_first + r : L - 1;
    __builtin_amdgcn_global_load_lds((glb_void*)(gbase + gr * rs + c * 16), (lds_void*)(dst + inst * 1024), 16, 0, 0);
  }
}

constexpr int kRedWave = 448;
__device__ __forceinline__ float bf16_round(float v) { return __uint_as_float(pack_bf16x2(v, 0.f) << 16); }
__device__ __forceinline__ uint4 vec_frag(const float* x, int u, int h, int l31) {
  const float4 a = *reinterpret_cast<const float4*>(x + 16 * u + 4 * h), b = *reinterpret_cast<const float4*>(x + 16 * u + 8 + 4 * h);
  float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  if (l31 == 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] -= bf16_round(v[e]);
  }
  if (l31 > 1) return make_uint4(0u, 0u, 0u, 0u);
  return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}

template <bool HAS_KB>
__global__ __launch_bounds__(256) void attn_bwd_2wg_kernel(AttnBwdArgs a, int nt, int ra, int nw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnArgs& f = a.f;
  const int head = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int L = f.lens ? f.lens[b] : f.L, LKP = 32 * nt;
  const int64_t row0 = f.cu ? (int64_t)f.cu[b] : (int64_t)b * f.L;
  nt = (L + 31) >> 5;
  const int IMG = ra * 128, padb = (LKP - ra) * 128;
  char* img0 = smem;                             // K, then Q
  char* img1 = smem + IMG;                       // V, then dO
  char* stage = smem + 2 * IMG + padb + wave * 4096;     // this wave's own block: Q (phase A), K (phase B)
  float* lseA = reinterpret_cast<float*>(smem + 2 * IMG + padb + nw * 4096);
  float* dA = lseA + LKP;
  float* kb = dA + LKP;
  float* rA = kb + LKP;                          // r_q = sum_k dS[q][k] of every query (phase A -> phase B)
  float* red = rA + LKP;
  float* red_w = red + wave * kRedWave;
  const bool want_db = a.db_part != nullptr;
  const int64_t rs = f.row_stride * 2, cs = f.ctx_stride * 2;
  const int64_t base = (row0 * f.row_stride + head * 64) * 2;
  const int64_t cbase = (row0 * f.ctx_stride + head * 64) * 2;
  constexpr float kLog2e = 1.4426950408889634f;
  const int nload = 32 * nt < ra ? 32 * nt : ra;

  // ---- phase A images: K, V
  dma_rows(img0, reinterpret_cast<const char*>(f.k) + base, rs, 0, nload, L, wave, nw, lane);
  dma_rows(img1, reinterpret_cast<const char*>(f.v) + base, rs, 0, nload, L, wave, nw, lane);
  for (int i = tid * 16; i < padb; i += 64 * nw * 16) *reinterpret_cast<uint4*>(smem + 2 * IMG + i) = make_uint4(0u, 0u, 0u, 0u);
  for (int key = tid; key < 32 * nt; key += 64 * nw)
    kb[key] = key < L ? (HAS_KB ? f.key_bias[row0 + key] * kLog2e : 0.f) : -INFINITY;
  for (int i = tid; i < kRedWave * nw; i += 64 * nw) red[i] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int fl = swz_f(l31);
  uint32_t roff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) roff[s] = (uint32_t)l31 * 128u + ((uint32_t)((2 * s + h) ^ fl) << 4);
  const int t16 = lane & 15, sub = (lane >> 4) & 1;
  const uint32_t tch = (uint32_t)((((t16 >> 3) & 1) << 2) | (sub << 1) | (((t16 & 3) >> 1) ^ h));
  const uint32_t trow = (uint32_t)(4 * h + (t16 >> 2)) * 128u + (uint32_t)(t16 & 1) * 8u;
  auto tr_frag = [&](const char* tile, int u, int dt) -> uint4 {
    const uint2 lo = tr4(tile + u * 2048 + trow + ((tch ^ (uint32_t)(dt << 2)) << 4));
    const uint2 hi = tr4(tile + u * 2048 + 1024 + trow + ((tch ^ (uint32_t)((dt << 2) | 2)) << 4));
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
  // out[col][d] += mul * sum over the 32 rows of img_blk of img_blk[row][d] x[row]     (col 0 / 1: vec_frag)
  auto bias_vec = [&](const char* img_blk, const float* x, float mul, float* out) {
    f32x16_t acc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint4 bf = vec_frag(x, u, h, l31);
      mma32(acc[0], tr_frag(img_blk, u, 0), bf, bf16_t());
      mma32(acc[1], tr_frag(img_blk, u, 1), bf, bf16_t());
    }
    if (l31 < 2) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          float4* o = reinterpret_cast<float4*>(out + l31 * 64 + dt * 32 + 8 * qd + 4 * h);
          float4 v = *o;
          v.x += acc[dt][4 * qd] * mul; v.y += acc[dt][4 * qd + 1] * mul; v.z += acc[dt][4 * qd + 2] * mul; v.w += acc[dt][4 * qd + 3] * mul;
          *o = v;
        }
    }
  };
  const float scale = f.scale;
  const float c = scale * kLog2e;

  // ------------------------------------------------ phase A: dQ of this wave's query blocks ---------------------
  for (int blk = wave; blk < nt; blk += nw) {
    const int row = blk * 32 + l31;
    const int rowc = row < L ? row : L - 1;
    // own rows: the Q block through the staging image, dO and O straight into registers
    dma_rows(stage, reinterpret_cast<const char*>(f.q) + base, rs, blk * 32, 32, L, 0, 1, lane);
    uint4 of[4], gf[4], xf[4];
    {
      const char* op = reinterpret_cast<const char*>(f.ctx) + cbase + (int64_t)rowc * cs;
      const char* gp = reinterpret_cast<const char*>(a.dctx) + cbase + (int64_t)rowc * cs;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        of[s] = *reinterpret_cast<const uint4*>(op + (2 * s + h) * 16);
        gf[s] = *reinterpret_cast<const uint4*>(gp + (2 * s + h) * 16);
      }
    }
    const float lse_q = row < L ? f.lse[((int64_t)b * f.H + head) * f.L + row] : INFINITY;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    float d_q = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      xf[s] = *reinterpret_cast<const uint4*>(stage + roff[s]);
      float gv[8], ov[8];
      unpack_chunk(gf[s], gv, bf16_t());
      unpack_chunk(of[s], ov, bf16_t());
#pragma unroll
      for (int e = 0; e < 8; ++e) d_q += gv[e] * ov[e];
    }
    d_q += __shfl_xor(d_q, 32, 64);
    if (h == 0) { lseA[row] = -lse_q * kLog2e; dA[row] = row < L ? -d_q : 0.f; }
    const float nlse_q = -lse_q * kLog2e;

    f32x16_t dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    float rs4[4] = {0.f, 0.f, 0.f, 0.f};
    auto tile_a = [&](int t, auto with_kb) {
      constexpr bool WITH_KB = decltype(with_kb)::value;
      f32x16_t sacc, pacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = pacc[r] = 0.f;
      const char* kt = img0 + t * 4096;
      const char* vt = img1 + t * 4096;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        mma32(sacc, *reinterpret_cast<const uint4*>(kt + roff[s]), xf[s], bf16_t());
        mma32(pacc, *reinterpret_cast<const uint4*>(vt + roff[s]), gf[s], bf16_t());
      }
      float ds[16];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        float kbv[4] = {0.f, 0.f, 0.f, 0.f};
        if (WITH_KB) {
          const float4 kb4 = *reinterpret_cast<const float4*>(kb + 32 * t + 8 * qd + 4 * h);
          kbv[0] = kb4.x; kbv[1] = kb4.y; kbv[2] = kb4.z; kbv[3] = kb4.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = __builtin_amdgcn_exp2f(fmaf(sacc[4 * qd + e], c, WITH_KB ? kbv[e] + nlse_q : nlse_q));
          ds[4 * qd + e] = p * (pacc[4 * qd + e] - d_q);
          rs4[e] += ds[4 * qd + e];
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        uint4 dc;
        dc.x = pack_bf16x2(ds[8 * u + 0], ds[8 * u + 1]);
        dc.y = pack_bf16x2(ds[8 * u + 2], ds[8 * u + 3]);
        dc.z = pack_bf16x2(ds[8 * u + 4], ds[8 * u + 5]);
        dc.w = pack_bf16x2(ds[8 * u + 6], ds[8 * u + 7]);
        mma32(dq[0], tr_frag(kt, u, 0), dc, bf16_t());
        mma32(dq[1], tr_frag(kt, u, 1), dc, bf16_t());
      }
    };
    if (HAS_KB) {
#pragma unroll 1
      for (int t = 0; t < nt; ++t) tile_a(t, std::true_type());
    } else {
#pragma unroll 1
      for (int t = 0; t < nt - 1; ++t) tile_a(t, std::false_type());
      tile_a(nt - 1, std::true_type());
    }
    if (row < L) {
      bf16_t* dqp = reinterpret_cast<bf16_t*>(a.dq) + (row0 + row) * f.row_stride + head * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = dq[dt][4 * qd + e] * scale;
          st4(dqp + dt * 32 + 8 * qd + 4 * h, v);
        }
    }
    float r = (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
    r += __shfl_xor(r, 32, 64);
    if (h == 0) rA[row] = r;                       // (rows >= L: exact zeros)
  }
  // every wave is done with the K / V images; lseA / dA / rA are complete.  (Waited for before the dq stores could matter:
  // only loads are outstanding-critical here -- the DMA below is issued after the barrier.)
  __syncthreads();

  // ------------------------------------------------ phase B: dK, dV of this wave's key blocks -------------------
  dma_rows(img0, reinterpret_cast<const char*>(f.q) + base, rs, 0, nload, L, wave, nw, lane);
  dma_rows(img1, reinterpret_cast<const char*>(a.dctx) + cbase, cs, 0, nload, L, wave, nw, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (also drains the dq stores: the price of the counted wait here)
  __syncthreads();
  for (int blk = wave; blk < nt; blk += nw) {
    const int row = blk * 32 + l31;
    const int rowc = row < L ? row : L - 1;
    dma_rows(stage, reinterpret_cast<const char*>(f.k) + base, rs, blk * 32, 32, L, 0, 1, lane);
    uint4 gf[4], xf[4];
    {
      const char* vp = reinterpret_cast<const char*>(f.v) + base + (int64_t)rowc * rs;
#pragma unroll
      for (int s = 0; s < 4; ++s) gf[s] = *reinterpret_cast<const uint4*>(vp + (2 * s + h) * 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 4; ++s) xf[s] = *reinterpret_cast<const uint4*>(stage + roff[s]);
    const float kb_key = HAS_KB ? kb[row] : 0.f;
    const float ek = HAS_KB ? 1.f : __builtin_amdgcn_exp2f(kb[row]);
    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;
    float cs4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int t = 0; t < nt; ++t) {
      f32x16_t sacc, pacc;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 d4 = *reinterpret_cast<const float4*>(dA + 32 * t + 8 * qd + 4 * h);
        pacc[4 * qd] = d4.x; pacc[4 * qd + 1] = d4.y; pacc[4 * qd + 2] = d4.z; pacc[4 * qd + 3] = d4.w;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
      const char* qt = img0 + t * 4096;
      const char* gt = img1 + t * 4096;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        mma32(sacc, *reinterpret_cast<const uint4*>(qt + roff[s]), xf[s], bf16_t());
        mma32(pacc, *reinterpret_cast<const uint4*>(gt + roff[s]), gf[s], bf16_t());
      }
      float p[16], ds[16];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 l4 = *reinterpret_cast<const float4*>(lseA + 32 * t + 8 * qd + 4 * h);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pe = __builtin_amdgcn_exp2f(fmaf(sacc[4 * qd + e], c, HAS_KB ? kb_key + lv[e] : lv[e]));
          p[4 * qd + e] = pe;
          ds[4 * qd + e] = pe * pacc[4 * qd + e];
          cs4[e] += ds[4 * qd + e];
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        uint4 pc, dc;
        pc.x = pack_bf16x2(p[8 * u + 0], p[8 * u + 1]);   dc.x = pack_bf16x2(ds[8 * u + 0], ds[8 * u + 1]);
        pc.y = pack_bf16x2(p[8 * u + 2], p[8 * u + 3]);   dc.y = pack_bf16x2(ds[8 * u + 2], ds[8 * u + 3]);
        pc.z = pack_bf16x2(p[8 * u + 4], p[8 * u + 5]);   dc.z = pack_bf16x2(ds[8 * u + 4], ds[8 * u + 5]);
        pc.w = pack_bf16x2(p[8 * u + 6], p[8 * u + 7]);   dc.w = pack_bf16x2(ds[8 * u + 6], ds[8 * u + 7]);
        mma32(dv[0], tr_frag(gt, u, 0), pc, bf16_t());
        mma32(dv[1], tr_frag(gt, u, 1), pc, bf16_t());
        mma32(dk[0], tr_frag(qt, u, 0), dc, bf16_t());
        mma32(dk[1], tr_frag(qt, u, 1), dc, bf16_t());
      }
    }
    if (row < L) {
      bf16_t* dkp = reinterpret_cast<bf16_t*>(a.dk) + (row0 + row) * f.row_stride + head * 64;
      bf16_t* dvp = reinterpret_cast<bf16_t*>(a.dv) + (row0 + row) * f.row_stride + head * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          float vk[4], vv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) { vk[e] = dk[dt][4 * qd + e] * (scale * ek); vv[e] = dv[dt][4 * qd + e] * ek; }
          st4(dkp + dt * 32 + 8 * qd + 4 * h, vk);
          st4(dvp + dt * 32 + 8 * qd + 4 * h, vv);
        }
    }
    if (want_db) {
      float cc = ((cs4[0] + cs4[1]) + (cs4[2] + cs4[3])) * ek;
      cc += __shfl_xor(cc, 32, 64);
      if (h == 0) { red_w[384 + l31] = cc; red_w[416 + l31] = row < L ? 1.f : 0.f; }
      __builtin_amdgcn_wave_barrier();
      bias_vec(stage, red_w + 384, scale, red_w);                      // dbq share: scale K^T c (own K block: the staging image)
      bias_vec(img0 + blk * 4096, rA + 32 * blk, scale, red_w + 128);  // dbk share: scale Q^T r
      bias_vec(img1 + blk * 4096, red_w + 416, 1.0f, red_w + 256);     // dbv share: dO^T 1
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (want_db) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int i = tid; i < 192; i += 64 * nw) {
      float s = 0.f;
      const int o = (i >> 6) * 128 + (i & 63);
      for (int w = 0; w < nw; ++w) s += red[w * kRedWave + o] + red[w * kRedWave + o + 64];
      a.db_part[((int64_t)b * 3 + (i >> 6)) * (f.H * 64) + head * 64 + (i & 63)] = s;
    }
  }
}

int lds_bytes(int L, int nw) {
  const int nt = (L + 31) / 32, ra = (L + 7) / 8 * 8;
  return 2 * ra * 128 + (32 * nt - ra) * 128 + nw * 4096 + 4 * 32 * nt * 4 + nw * kRedWave * 4;
}

int launch(const AttnBwdArgs& a, hipStream_t stream) {
  const int nt = (a.f.L + 31) / 32, ra = (a.f.L + 7) / 8 * 8;
  const int nw = nt < 4 ? nt : 4;
  const int bytes = lds_bytes(a.f.L, nw);
  auto* kern = a.f.key_bias != nullptr ? &attn_bwd_2wg_kernel<true> : &attn_bwd_2wg_kernel<false>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  hipLaunchKernelGGL(kern, dim3(a.f.H, a.f.B), dim3(64 * nw), bytes, stream, a, nt, ra, nw);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace x2

// ------------------------------------------------------------------------------------------------ harness
__global__ void fill_bf16(uint16_t* p, size_t n, unsigned seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed * 40503u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    const float v = ((x & 0xffff) / 65535.0f - 0.5f) * 2.0f * scale;
    p[i] = (uint16_t)(__float_as_uint(v) >> 16);
  }
}
__global__ void maxdiff_bf16(const uint16_t* a, const uint16_t* b, size_t n, float* out) {
  float m = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(__uint_as_float((unsigned)a[i] << 16) - __uint_as_float((unsigned)b[i] << 16)));
  atomicMax(reinterpret_cast<int*>(out), __float_as_int(m));      // (non-negative floats order as ints)
}
__global__ void maxdiff_f32(const float* a, const float* b, size_t n, float* out) {
  float m = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(a[i] - b[i]));
  atomicMax(reinterpret_cast<int*>(out), __float_as_int(m));
}

int main(int argc, char** argv) {
  const int batch = argc > 1 ? atoi(argv[1]) : 1024;
  const int iters = argc > 2 ? atoi(argv[2]) : 20;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  struct AT { const char* name; int B, L, H; bool kb; };
  const AT ats[] = {{"attn.vit", batch, 197, 12, false}, {"attn.bert", batch, 64, 12, true}, {"attn.ragged", 64, 171, 3, false}};
  float* d_md;
  CK(hipMalloc(&d_md, 4));
  for (const AT& t : ats) {
    const size_t rows = (size_t)t.B * t.L, W = (size_t)t.H * 64;
    uint16_t *qkv, *ctx, *dctx, *dq0, *dq1;
    float *lse, *kbias = nullptr, *dbp, *db0, *db1;
    CK(hipMalloc(&qkv, rows * 3 * W * 2)); CK(hipMalloc(&ctx, rows * W * 2)); CK(hipMalloc(&dctx, rows * W * 2));
    CK(hipMalloc(&dq0, rows * 3 * W * 2)); CK(hipMalloc(&dq1, rows * 3 * W * 2));
    CK(hipMalloc(&lse, (size_t)t.B * t.H * t.L * 4));
    CK(hipMalloc(&dbp, (size_t)t.B * 3 * W * 4)); CK(hipMalloc(&db0, 3 * W * 4)); CK(hipMalloc(&db1, 3 * W * 4));
    fill_bf16<<<2048, 256, 0, st>>>(qkv, rows * 3 * W, 21u, 1.5f);
    fill_bf16<<<2048, 256, 0, st>>>(dctx, rows * W, 22u, 1.0f);
    if (t.kb) {       // BERT-style key mask: the last third of every sentence masked
      std::vector<float> hk(rows);
      for (size_t i = 0; i < rows; ++i) hk[i] = (int)(i % t.L) >= 2 * t.L / 3 ? -10000.f : 0.f;
      CK(hipMalloc(&kbias, rows * 4));
      CK(hipMemcpy(kbias, hk.data(), rows * 4, hipMemcpyHostToDevice));
    }
    ezclip::AttnArgs fa;
    fa.q = qkv; fa.k = qkv + W; fa.v = qkv + 2 * W; fa.row_stride = 3 * W; fa.ctx = ctx; fa.ctx_stride = W; fa.lse = lse;
    fa.key_bias = kbias; fa.B = t.B; fa.L = t.L; fa.H = t.H; fa.scale = 0.125f;
    if (ezclip::attention_fwd(fa, EZCLIP_BF16, st) != 0) { printf("fwd ERROR %s\n", ezclip::last_error()); return 1; }
    ezclip::AttnBwdArgs ab;
    ab.f = fa; ab.dctx = dctx; ab.db_part = dbp;
    float ms[2] = {0, 0};
    for (int v = 0; v < 2; ++v) {
      uint16_t* dq = v == 0 ? dq0 : dq1;
      float* db = v == 0 ? db0 : db1;
      ab.dq = dq; ab.dk = dq + W; ab.dv = dq + 2 * W;
      ab.dbq = db; ab.dbk = db + W; ab.dbv = db + 2 * W;
      CK(hipMemsetAsync(db, 0, 3 * W * 4, st));
      auto run = [&]() -> int {
        if (v == 0) return ezclip::attention_bwd(ab, EZCLIP_BF16, st);
        const int rc = x2::launch(ab, st);
        if (rc) return rc;
        return ezclip::colsum3_add(dbp, 3 * W, t.B, 3 * W, ab.dbq, ab.dbk, ab.dbv, (int)W, EZCLIP_F32, st);
      };
      if (run() != 0) { printf("bwd v%d ERROR %s\n", v, ezclip::last_error()); return 1; }
      CK(hipStreamSynchronize(st));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      std::vector<float> keep(3 * W);
      CK(hipMemcpy(keep.data(), db, 3 * W * 4, hipMemcpyDeviceToHost));      // (the timed repetitions accumulate on top)
      CK(hipEventRecord(e0, st));
      for (int it = 0; it < iters; ++it) run();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms[v], e0, e1));
      ms[v] /= iters;
      CK(hipMemcpy(db, keep.data(), 3 * W * 4, hipMemcpyHostToDevice));
    }
    float md = 0, mdb = 0;
    CK(hipMemsetAsync(d_md, 0, 4, st));
    maxdiff_bf16<<<1024, 256, 0, st>>>(dq0, dq1, rows * 3 * W, d_md);
    CK(hipMemcpyAsync(&md, d_md, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    CK(hipMemsetAsync(d_md, 0, 4, st));
    maxdiff_f32<<<64, 256, 0, st>>>(db0, db1, 3 * W, d_md);
    CK(hipMemcpyAsync(&mdb, d_md, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    printf("%-12s B=%5d L=%4d H=%3d : library fused %.3f ms   two-workgroup %.3f ms (%d B LDS)   max |dqkv diff| %.4g   max |bias-grad diff| %.4g\n",
           t.name, t.B, t.L, t.H, ms[0], ms[1], x2::lds_bytes(t.L, t.L > 96 ? 4 : (t.L + 31) / 32), md, mdb);
    hipFree(qkv); hipFree(ctx); hipFree(dctx); hipFree(dq0); hipFree(dq1); hipFree(lse); hipFree(dbp); hipFree(db0); hipFree(db1);
    if (kbias) hipFree(kbias);
  }
  return 0;
}
