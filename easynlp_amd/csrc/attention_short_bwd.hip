// Fused short-sequence attention backward (bf16, L <= 256, head_dim 64) for gfx950; the forward lives in attention_short.hip.
// A translation unit of its own because it is compiled with -mllvm -disable-lsr (build.py): loop strength reduction gave
// every one of the ~36 LDS reads of a key tile its own induction register (237 VGPRs, 36 address increments per tile);
// without it the loops recompute 20 addresses from loop-invariant lane offsets (170 VGPRs, 135 -> 108 vector
// instructions per tile of pass B before the other changes described below).
#include <type_traits>

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

// =====================================================================================================
// Fused backward for the same shapes (bf16, L <= 256): ONE kernel, one workgroup per (sample, head).  Q, K, V and dO
// of the head are brought into LDS once (four row-major 128-byte-row images by LDS-DMA) and serve both halves:
//   pass A (wave = 32-query block, sweep over key tiles):   S^T = mfma(K, Q), dP^T = mfma(V, dO)   (lane = query)
//        P = exp(S - lse), dS = P o (dP - D);   dQ^T += mfma(K^T, dS)
//   pass B (wave = 32-key block, sweep over query tiles):   S = mfma(Q, K), dP = mfma(dO, V)       (lane = key)
//        dV^T += mfma(dO^T, P),  dK^T += mfma(Q^T, dS)
// with D_q = <dO_q, O_q> and the log-sum-exp saved by the forward.  Every image is read BOTH as row fragments
// (ds_read_b128) and as transposed fragments (ds_read_b64_tr_b16); one swizzle serves both: the 16-byte chunk index of
// row r is XORed with f(r) = ((r>>1)&1)<<2 | (r>>2)&3 -- a bijection of (r>>1)&7, so the 32-row b128 fragment groups
// stay conflict-free, and its bit 2 flips between rows r and r+2, so the four rows of a transpose block fall on four
// different bank quarters.  The two-kernel version read q, k, v, dO from HBM twice and staged transposed copies
// through registers.

__device__ __forceinline__ int swz_f(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

// -DEZ_ATTN_BWD_TRACE (tools/build_variants.py; never in the shipped library): every wave of attn_bwd_short_kernel writes the shader-clock
// time of its phase boundaries -- entry, images landed, D / lse published, end of pass A, end of pass B, exit -- plus its HW_ID and the
// 100 MHz real-time counter at entry and exit into a device array; tools/attn_bwd_trace.py reads it back (ezclip_dbg_attn_trace).
#ifdef EZ_ATTN_BWD_TRACE
constexpr int kTraceWords = 10, kTraceWgs = 16384, kTraceWaves = 9;
__device__ unsigned long long g_attn_trace[(size_t)kTraceWgs * kTraceWaves * kTraceWords];
#define EZ_TRACE(slot)                                                                                             \
  do {                                                                                                             \
    const int wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                                           \
    if (lane == 0 && wg_ < kTraceWgs) g_attn_trace[((size_t)wg_ * kTraceWaves + wave) * kTraceWords + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define EZ_TRACE_AUX(slot, value)                                                                                  \
  do {                                                                                                             \
    const int wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                                           \
    if (lane == 0 && wg_ < kTraceWgs) g_attn_trace[((size_t)wg_ * kTraceWaves + wave) * kTraceWords + (slot)] = (value); \
  } while (0)
#else
#define EZ_TRACE(slot) do { } while (0)
#define EZ_TRACE_AUX(slot, value) do { } while (0)
#endif

__device__ __forceinline__ void dma_rows_f(char* dst, const char* gbase, int64_t rs, int nrows, int L, int wave, int nwaves, int lane) {
  const int ninst = nrows / 8;
  for (int inst = wave; inst < ninst; inst += nwaves) {
    const int r = inst * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz_f(r);
    const int gr = r < L ? r : L - 1;
    __builtin_amdgcn_global_load_lds((glb_void*)(gbase + gr * rs + c * 16), (lds_void*)(dst + inst * 1024), 16, 0, 0);
  }
}

// Bias gradients of the q / k / v projections (column sums of dQ, dK, dV over all tokens) without reducing the
// accumulators across lanes.  Summing dQ^T[d][q] over q is a cross-lane reduction of 32 + 64 + 64 accumulator registers
// (it was 480 DPP adds per wave: +10 % on the ViT kernel, +40 % on BERT's two-tile one).  Instead:
//   sum_q dQ[q]  = scale K^T c,  c_k = sum_q dS[q][k]   -- in-lane in pass B (one key per lane, queries in registers)
//   sum_k dK[k]  = scale Q^T r,  r_q = sum_k dS[q][k]   -- in-lane in pass A (one query per lane, keys in registers)
//   sum_k dV[k]  = dO^T 1                               -- rows of P sum to one
// each a [64 x 32] x [32] product over the wave's own 32 rows: two MFMAs per 32 output features with the vector as the
// B operand -- column 0 carries bf16(x), column 1 bf16(x - bf16(x)) (a 16-bit mantissa in all), the other columns zero.
// Per wave: kRedWave floats of LDS = [3 vectors][2 columns][64 features] results, then two 32-float gather areas.
constexpr int kRedWave = 448;
__device__ __forceinline__ float bf16_round(float v) { return __uint_as_float(pack_bf16x2(v, 0.f) << 16); }
// B fragment (u = 0, 1: which 16 of the 32 rows): rows 16u + 4h + {0..3} and 16u + 8 + 4h + {0..3} of x, as tr_frag orders them
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

// HAS_KB / CAUSAL are template parameters: as run-time flags their index arithmetic and selects were executed for every
// score of every tile whatever the flags said (80 of ~200 VALU instructions per tile and pass).  The probabilities are
// recomputed in the base-2 domain, p = 2^(s c + kb log2 e - lse log2 e) with c = scale log2 e: one fma + v_exp_f32 per score.
// DROP: dropout on the probabilities (the forward wrote the keep bits, AttnArgs::keep_bits): O = (P o M s) V, so
//   dV = (P o M s)^T dO,   dP = (dO V^T) o M s,   dS = P o (dP - D) with the same D = <dO, O>,
// pass A reads one 32-key word per tile for its query, pass B the words of a tile's 32 queries through a wave-private LDS
// area (bit = its key), and the rows of P o M s no longer sum to one: sum_k dV[k] = dO^T rowsum(P o M s).
#ifndef EZ_ATTN_BWD_LB
#define EZ_ATTN_BWD_LB 576
#endif
#ifndef EZ_ATTN_BWD_UNROLL
#define EZ_ATTN_BWD_UNROLL 1
#endif
#ifdef EZ_ATTN_BWD_PRIO
#define EZ_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define EZ_PRIO(x) do { } while (0)
#endif
template <bool HAS_KB, bool CAUSAL, bool DROP>
__global__ __launch_bounds__(DROP ? 512 : EZ_ATTN_BWD_LB) void attn_bwd_short_kernel(AttnBwdArgs a, int nt, int ra) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnArgs& f = a.f;
  const int head = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int nwaves = nt;      // 64 * nt threads (nt of the launch: the longest sample's tiles): no idle waves holding registers
  const int keep_words = nt;  // (words per (row, head) of AttnArgs::keep_bits)
  EZ_TRACE(0);
  EZ_TRACE_AUX(6, (unsigned long long)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | 4) |       // HW_ID (hwreg 4), XCC_ID (hwreg 20) above it
                      ((unsigned long long)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | 20) << 32));
  EZ_TRACE_AUX(7, __builtin_amdgcn_s_memrealtime());
  // packed batches (AttnArgs::cu / lens): rows cu[b] .. cu[b] + lens[b] - 1; nt (launch, LDS layout) is the longest sample's
  const int L = f.lens ? f.lens[b] : f.L, LKP = 32 * nt;
  const int64_t row0 = f.cu ? (int64_t)f.cu[b] : (int64_t)b * f.L;
  nt = (L + 31) >> 5;
  // Four images of `ra` rows (the longest sample's length rounded up to the 8 rows of a DMA piece) -- not of 32 nt rows: at
  // 257 tokens (ViT-L/14: nine tiles) four 288-row images do not fit 160 KiB, four 264-row ones do.  The last tile's rows
  // beyond `ra` fall on the first rows of the NEXT image (finite data of this head) or, behind the last image, on a zeroed
  // pad: whatever they hold is multiplied by an exact zero (keys >= L: p = 0; queries >= L: lse = inf).
  const int IMG = ra * 128, padb = (LKP - ra) * 128;
  char* imgQ = smem;
  char* imgK = smem + IMG;
  char* imgV = smem + 2 * IMG;
  char* imgG = smem + 3 * IMG;                 // dO
  for (int i = tid * 16; i < padb; i += 64 * nwaves * 16) *reinterpret_cast<uint4*>(smem + 4 * IMG + i) = make_uint4(0u, 0u, 0u, 0u);
  float* lseA = reinterpret_cast<float*>(smem + 4 * IMG + padb);
  float* dA = lseA + LKP;
  float* kb = dA + LKP;
  float* red = kb + LKP;                       // [waves][kRedWave]: bias gradients (vec_frag above)
  const bool want_db = a.db_part != nullptr;
  const int64_t rs = f.row_stride * 2, cs = f.ctx_stride * 2;
  const int64_t base = (row0 * f.row_stride + head * 64) * 2;
  const int64_t cbase = (row0 * f.ctx_stride + head * 64) * 2;

  const int nload = 32 * nt < ra ? 32 * nt : ra;      // rows >= L repeat row L - 1
  dma_rows_f(imgQ, reinterpret_cast<const char*>(f.q) + base, rs, nload, L, wave, nwaves, lane);
  dma_rows_f(imgK, reinterpret_cast<const char*>(f.k) + base, rs, nload, L, wave, nwaves, lane);
  dma_rows_f(imgV, reinterpret_cast<const char*>(f.v) + base, rs, nload, L, wave, nwaves, lane);
  dma_rows_f(imgG, reinterpret_cast<const char*>(a.dctx) + cbase, cs, nload, L, wave, nwaves, lane);
  constexpr float kLog2e = 1.4426950408889634f;
  for (int key = tid; key < 32 * nt; key += 64 * nwaves)   // key bias in base-2 units; keys >= L: -inf (p = 0)
    kb[key] = key < L ? (HAS_KB ? f.key_bias[row0 + key] * kLog2e : 0.f) : -INFINITY;

  // this wave's 32 rows (queries in pass A, keys in pass B)
  const int blk = wave;
  const bool active = blk * 32 < L;
  const int row = blk * 32 + l31;
  const int rowc = row < L ? row : L - 1;
  // D_q = <dO_q, O_q>: each lane of the pair (h = 0, 1) takes half of the 64 columns of O straight from global memory
  uint4 of[4];
  float lse_q = INFINITY;
  if (active) {
    const char* op = reinterpret_cast<const char*>(f.ctx) + cbase + (int64_t)rowc * cs;
#pragma unroll
    for (int s = 0; s < 4; ++s) of[s] = *reinterpret_cast<const uint4*>(op + (2 * s + h) * 16);
    if (row < L) lse_q = f.lse[((int64_t)b * f.H + head) * f.L + row];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  EZ_TRACE(9);                                   // (this wave's own loads have landed; the barrier waits for the others')
  __syncthreads();
  EZ_TRACE(1);

  const int fl = swz_f(l31);                     // (32t + l31 has the same f as l31)
  uint32_t roff[4];                              // row fragments: row l31 of a 32-row tile, chunk (2s+h) ^ f
#pragma unroll
  for (int s = 0; s < 4; ++s) roff[s] = (uint32_t)l31 * 128u + ((uint32_t)((2 * s + h) ^ fl) << 4);
  // transposed fragments: lane supplies row 4h + (t16>>2) (+ 8*part + 16u + 32t), 4 columns at dt*32 + sub*16 + 4*(t16&3)
  const int t16 = lane & 15, sub = (lane >> 4) & 1;
  const uint32_t tch = (uint32_t)((((t16 >> 3) & 1) << 2) | (sub << 1) | (((t16 & 3) >> 1) ^ h));   // chunk for dt = 0, part = 0
  const uint32_t trow = (uint32_t)(4 * h + (t16 >> 2)) * 128u + (uint32_t)(t16 & 1) * 8u;
  // address(dt, part) = trow + 1024*part + ((tch ^ (dt<<2 | part<<1)) << 4)
  auto tr_frag = [&](const char* tile, int u, int dt) -> uint4 {     // keys/queries 16u + 4h + {0..3}, + 8 + {0..3}
    const uint2 lo = tr4(tile + u * 2048 + trow + ((tch ^ (uint32_t)(dt << 2)) << 4));
    const uint2 hi = tr4(tile + u * 2048 + 1024 + trow + ((tch ^ (uint32_t)((dt << 2) | 2)) << 4));
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  };

  float* red_w = red + wave * kRedWave;
  // the wave's share of a bias gradient: out[col][d] = mul * sum over its 32 rows of img[row][d] x[row]  (col 0 / 1: vec_frag)
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
        for (int qd = 0; qd < 4; ++qd)
          *reinterpret_cast<float4*>(out + l31 * 64 + dt * 32 + 8 * qd + 4 * h) =
              make_float4(acc[dt][4 * qd] * mul, acc[dt][4 * qd + 1] * mul, acc[dt][4 * qd + 2] * mul, acc[dt][4 * qd + 3] * mul);
    }
  };

  float d_q = 0.f;
  uint4 gf[4];        // dO row fragments of this wave's rows (pass A); reused as V fragments in pass B
  uint4 xf[4];        // Q row fragments (pass A); K row fragments (pass B)
  if (active) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      gf[s] = *reinterpret_cast<const uint4*>(imgG + blk * 4096 + roff[s]);
      xf[s] = *reinterpret_cast<const uint4*>(imgQ + blk * 4096 + roff[s]);
      float gv[8], ov[8];
      unpack_chunk(gf[s], gv, bf16_t());
      unpack_chunk(of[s], ov, bf16_t());
#pragma unroll
      for (int e = 0; e < 8; ++e) d_q += gv[e] * ov[e];
    }
    d_q += __shfl_xor(d_q, 32, 64);
    // lseA holds -lse log2 e (rows >= L: -inf -> P = 0 in pass B)
    if (h == 0) { lseA[row] = -lse_q * kLog2e; dA[row] = row < L ? -d_q : 0.f; }     // dA holds -D: pass B starts dP from it
  }
  __syncthreads();
  EZ_TRACE(2);
  const float scale = f.scale;
  const float c = scale * kLog2e;
  const float nlse_q = -lse_q * kLog2e;

  // ------------------------------------------------ pass A: dQ for queries 32*blk + l31 ------------------------
  if (active) {
    f32x16_t dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    float rs4[4] = {0.f, 0.f, 0.f, 0.f};       // r_q = sum over keys of dS[q][key] (this half-wave's keys), four chains
    float rd4[4] = {0.f, 0.f, 0.f, 0.f};       // DROP: sum over keys of (P o M s)[q][key]
    const uint32_t sbits = __float_as_uint(f.drop.scale);
    const uint32_t* kwp = DROP ? f.keep_bits + ((row0 + rowc) * f.H + head) * keep_words : nullptr;
    uint32_t kw_next = DROP ? kwp[0] : 0u;     // this query's keep word of the next tile (one tile ahead of its use)
    // one key tile; WITH_KB = false (no key bias and not the last tile: every key exists) drops the bias read and its add
    auto tile_a = [&](int t, auto with_kb) {
      constexpr bool WITH_KB = decltype(with_kb)::value;
      f32x16_t sacc, pacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = pacc[r] = 0.f;
      const char* kt = imgK + t * 4096;
      const char* vt = imgV + t * 4096;
      EZ_PRIO(1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        mma32(sacc, *reinterpret_cast<const uint4*>(kt + roff[s]), xf[s], bf16_t());     // S^T[key][q]
        mma32(pacc, *reinterpret_cast<const uint4*>(vt + roff[s]), gf[s], bf16_t());     // dP^T[key][q]
      }
      EZ_PRIO(0);
      uint32_t kwh = 0;
      if (DROP) {
        kwh = kw_next >> (4 * h);                       // bit 8 qd + e: key 8 qd + 4 h + e of this tile
        if (t + 1 < nt) kw_next = kwp[t + 1];
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
          // (kb: 0 / -inf for the keys >= L without a key bias)
          float p = __builtin_amdgcn_exp2f(fmaf(sacc[4 * qd + e], c, WITH_KB ? kbv[e] + nlse_q : nlse_q));
          if (CAUSAL && 32 * t + 8 * qd + 4 * h + e > row) p = 0.f;
          if (DROP) {
            // s where the key was kept, 0 where it was dropped
            const float mk = __uint_as_float((uint32_t)__builtin_amdgcn_sbfe((int)kwh, 8 * qd + e, 1) & sbits);
            ds[4 * qd + e] = p * fmaf(pacc[4 * qd + e], mk, -d_q);
            rd4[e] = fmaf(p, mk, rd4[e]);
          } else {
            ds[4 * qd + e] = p * (pacc[4 * qd + e] - d_q);
          }
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
        EZ_PRIO(1);
        mma32(dq[0], tr_frag(kt, u, 0), dc, bf16_t());      // dQ^T[d][q] += K^T . dS^T
        mma32(dq[1], tr_frag(kt, u, 1), dc, bf16_t());
        EZ_PRIO(0);
      }
    };
    if (HAS_KB || DROP) {
#pragma unroll EZ_ATTN_BWD_UNROLL
      for (int t = 0; t < nt; ++t) tile_a(t, std::true_type());
    } else {
#pragma unroll EZ_ATTN_BWD_UNROLL
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
    if (want_db) {      // dbk share = scale Q^T r and dbv share = dO^T 1 over this wave's queries (rows >= L: r = 0, mask 0)
      float r = (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
      r += __shfl_xor(r, 32, 64);
      float rd = 1.f;
      if (DROP) {
        rd = (rd4[0] + rd4[1]) + (rd4[2] + rd4[3]);
        rd += __shfl_xor(rd, 32, 64);
      }
      if (h == 0) { red_w[384 + l31] = r; red_w[416 + l31] = row < L ? rd : 0.f; }
      __builtin_amdgcn_wave_barrier();
      bias_vec(imgQ + blk * 4096, red_w + 384, scale, red_w + 128);
      bias_vec(imgG + blk * 4096, red_w + 416, 1.0f, red_w + 256);
      __builtin_amdgcn_wave_barrier();
    }
  }
  EZ_TRACE(3);
  // ------------------------------------------------ pass B: dK, dV for keys 32*blk + l31 -----------------------
  if (active) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      xf[s] = *reinterpret_cast<const uint4*>(imgK + blk * 4096 + roff[s]);
      gf[s] = *reinterpret_cast<const uint4*>(imgV + blk * 4096 + roff[s]);
    }
    // Without a key bias, kb is 0 or (keys >= L) -inf: a factor 2^kb = 1 / 0 of every probability of this lane's key -- of a
    // whole column of dK^T / dV^T -- applied to the accumulators after the loop; inside it p' = 2^(s c - lse) <= 1 (the rows
    // of a key >= L repeat key L - 1).  A real key bias stays in the exponent: where every key of a sample is masked the
    // -10000 is part of lse as well, and 2^(s c - lse) 2^kb would be inf * 0.
    const float kb_key = HAS_KB ? kb[row] : 0.f;
    const float ek = HAS_KB ? 1.f : __builtin_amdgcn_exp2f(kb[row]);
    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;
    float cs4[4] = {0.f, 0.f, 0.f, 0.f};       // c_k = sum over queries of dS[q][k] (this half-wave's queries)
    // DROP: the keep word (this wave's 32 keys) of query 32 t + l31, fetched one tile ahead, handed to the lanes that need
    // it -- 16 queries per lane -- through 32 words of the wave's LDS area
    const uint32_t sbits_b = __float_as_uint(f.drop.scale);
    uint32_t* kwl = reinterpret_cast<uint32_t*>(red_w + 384);
    auto keep_word_of = [&](int t) -> uint32_t {
      const int qq = 32 * t + l31;
      return f.keep_bits[((row0 + (qq < L ? qq : L - 1)) * f.H + head) * keep_words + blk];
    };
    uint32_t kwq_next = DROP ? keep_word_of(0) : 0u;
#pragma unroll EZ_ATTN_BWD_UNROLL
    for (int t = 0; t < nt; ++t) {
      f32x16_t sacc, pacc;
      if (DROP) {
        if (h == 0) kwl[l31] = kwq_next;
        if (t + 1 < nt) kwq_next = keep_word_of(t + 1);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) pacc[r] = 0.f;       // (dP o M s) - D: D joins after the mask
      } else {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {      // dP - D: the accumulator starts from -D of its query rows
          const float4 d4 = *reinterpret_cast<const float4*>(dA + 32 * t + 8 * qd + 4 * h);
          pacc[4 * qd] = d4.x; pacc[4 * qd + 1] = d4.y; pacc[4 * qd + 2] = d4.z; pacc[4 * qd + 3] = d4.w;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
      const char* qt = imgQ + t * 4096;
      const char* gt = imgG + t * 4096;
      EZ_PRIO(1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        mma32(sacc, *reinterpret_cast<const uint4*>(qt + roff[s]), xf[s], bf16_t());     // S[q][key]
        mma32(pacc, *reinterpret_cast<const uint4*>(gt + roff[s]), gf[s], bf16_t());     // dP[q][key] - D[q]
      }
      EZ_PRIO(0);
      float p[16], ds[16];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 l4 = *reinterpret_cast<const float4*>(lseA + 32 * t + 8 * qd + 4 * h);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w};
        float nd[4] = {0.f, 0.f, 0.f, 0.f};
        uint32_t kwv[4] = {0u, 0u, 0u, 0u};
        if (DROP) {
          const float4 d4 = *reinterpret_cast<const float4*>(dA + 32 * t + 8 * qd + 4 * h);
          nd[0] = d4.x; nd[1] = d4.y; nd[2] = d4.z; nd[3] = d4.w;
          const uint4 w4 = *reinterpret_cast<const uint4*>(kwl + 8 * qd + 4 * h);
          kwv[0] = w4.x; kwv[1] = w4.y; kwv[2] = w4.z; kwv[3] = w4.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float pe = __builtin_amdgcn_exp2f(fmaf(sacc[4 * qd + e], c, HAS_KB ? kb_key + lv[e] : lv[e]));
          if (CAUSAL && row > 32 * t + 8 * qd + 4 * h + e) pe = 0.f;      // this key lies after that query
          if (DROP) {
            const float mk = __uint_as_float((uint32_t)__builtin_amdgcn_sbfe((int)kwv[e], l31, 1) & sbits_b);
            p[4 * qd + e] = pe * mk;
            ds[4 * qd + e] = pe * fmaf(pacc[4 * qd + e], mk, nd[e]);
          } else {
            p[4 * qd + e] = pe;
            ds[4 * qd + e] = pe * pacc[4 * qd + e];
          }
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
        EZ_PRIO(1);
        mma32(dv[0], tr_frag(gt, u, 0), pc, bf16_t());      // dV^T[d][key] += dO^T . P
        mma32(dv[1], tr_frag(gt, u, 1), pc, bf16_t());
        mma32(dk[0], tr_frag(qt, u, 0), dc, bf16_t());      // dK^T[d][key] += Q^T . dS
        mma32(dk[1], tr_frag(qt, u, 1), dc, bf16_t());
        EZ_PRIO(0);
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
    if (want_db) {      // dbq share = scale K^T c over this wave's keys (keys >= L: c = 0)
      float cc = ((cs4[0] + cs4[1]) + (cs4[2] + cs4[3])) * ek;
      cc += __shfl_xor(cc, 32, 64);
      if (h == 0) red_w[384 + l31] = cc;
      __builtin_amdgcn_wave_barrier();
      bias_vec(imgK + blk * 4096, red_w + 384, scale, red_w);
    }
  }
  EZ_TRACE(4);
  if (want_db) {      // combine the waves in a fixed order; per-sample partials (12k workgroups hammering 2304 addresses
    // with atomics cost more than the pass over dqkv this replaces), summed over the batch afterwards.  The barrier orders
    // LDS only: __syncthreads() would also wait out the dk / dv stores (vmcnt(0)).
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int i = tid; i < 192; i += 64 * nwaves) {
      float s = 0.f;
      const int o = (i >> 6) * 128 + (i & 63);
      for (int w = 0; w < nt; ++w) s += red[w * kRedWave + o] + red[w * kRedWave + o + 64];
      a.db_part[((int64_t)b * 3 + (i >> 6)) * (f.H * 64) + head * 64 + (i & 63)] = s;
    }
  }
  EZ_TRACE(5);
  EZ_TRACE_AUX(8, __builtin_amdgcn_s_memrealtime());
}


// =====================================================================================================
// Round 4: every score tile ONCE.  The two-pass kernel above forms each 32 x 32 tile of S, P and dS twice -- once with the
// queries in the lanes (dQ), once with the keys (dK, dV): 28 MFMAs and 32 exponentials per lane and tile pair.  Here a wave owns a
// KEY block for the whole kernel (K / V row fragments and the K^T fragments of its 32 keys in registers, dK^T / dV^T in its
// accumulators) and sweeps the query tiles:
//     S = mfma(Q_t, K_w), dP = mfma(dO_t, V_w) - D      (lane = key, 16 queries in registers: the layout of pass B above)
//     P = 2^(S c - lse), dS = P o dP;   dV^T += mfma(dO_t^T, P),  dK^T += mfma(Q_t^T, dS)
// and dQ, the one product that contracts over the KEYS, takes dS through a 2-KiB wave-private LDS tile ([key][query] bf16, written
// as four 8-byte pieces per lane, read back with the transpose read as the B operand): dQ_t^T += mfma(K_w^T, dS^T).  20 MFMAs and
// 16 exponentials per lane and tile pair.  dQ_t is summed over the key-block waves in an LDS float image (it takes the place of
// the K / V images once their fragments sit in registers): in step s wave w works on query tile (w + s) mod nt, so no two waves
// touch the same rows in a step, a workgroup barrier separates the steps, and the order in which a tile receives its
// contributions is fixed -- the gradients stay bit-reproducible.  (Summation order of dQ differs from the two-pass kernel: equal
// within float32 rounding of the partial sums, not bit for bit.)
// The key-projection bias gradient sum_k dK[k] = scale Q^T r with r_q = sum_k dS[q][k] = sum_k P (dP - D) = D - D: exactly zero
// in exact arithmetic (softmax is shift-invariant; 1e-9 of rounding noise in the reference) -- written as zeros here.
// L <= 256 (at most eight waves: 2 per SIMD, 256 registers); longer sequences stay on the two-pass kernel.
constexpr int kOnceWave = 2304;     // per-wave LDS: 2048 B dS^T tile + 256 B keep words; afterwards the bias-gradient area (kRedWave floats)

template <bool HAS_KB, bool CAUSAL, bool DROP>
__global__ __launch_bounds__(512) void attn_bwd_once_kernel(AttnBwdArgs a, int nt, int ra) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnArgs& f = a.f;
  const int head = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int nwaves = nt;
  const int keep_words = nt;
  const int L = f.lens ? f.lens[b] : f.L, LKP = 32 * nt;
  const int64_t row0 = f.cu ? (int64_t)f.cu[b] : (int64_t)b * f.L;
  nt = (L + 31) >> 5;
  // LDS: Q image | dO image | zeroed pad | region X | zeroed pad | per-row floats | per-wave areas
  // region X: first the K and V images (2 ra rows of 128 B), then -- their fragments loaded -- the float32 dQ image (ra rows of 256 B).
  // The pads take the last tile's rows beyond `ra` of the dO and of the V image (the Q and K images run over into the image behind
  // them: finite bf16 data, always multiplied by an exact zero; float words read as bf16 pairs would not be: -inf, NaN patterns).
  const int IMG = ra * 128, padb = (LKP - ra) * 128;
  char* imgQ = smem;
  char* imgG = smem + IMG;
  char* pad = smem + 2 * IMG;
  char* imgK = pad + padb;
  char* imgV = imgK + IMG;
  char* dqimg = imgK;
  char* pad2 = imgK + 2 * IMG;
  float* lseA = reinterpret_cast<float*>(pad2 + padb);
  float* dA = lseA + LKP;
  float* kb = dA + LKP;
  char* wv = reinterpret_cast<char*>(kb + LKP) + wave * kOnceWave;
  float* red = reinterpret_cast<float*>(reinterpret_cast<char*>(kb + LKP));      // [waves][kOnceWave / 4]: used as [kRedWave] after the loop
  for (int i = tid * 16; i < padb; i += 64 * nwaves * 16) {
    *reinterpret_cast<uint4*>(pad + i) = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(pad2 + i) = make_uint4(0u, 0u, 0u, 0u);
  }
  const bool want_db = a.db_part != nullptr;
  const int64_t rs = f.row_stride * 2, cs = f.ctx_stride * 2;
  const int64_t base = (row0 * f.row_stride + head * 64) * 2;
  const int64_t cbase = (row0 * f.ctx_stride + head * 64) * 2;

  const int nload = 32 * nt < ra ? 32 * nt : ra;
  dma_rows_f(imgQ, reinterpret_cast<const char*>(f.q) + base, rs, nload, L, wave, nwaves, lane);
  dma_rows_f(imgK, reinterpret_cast<const char*>(f.k) + base, rs, nload, L, wave, nwaves, lane);
  dma_rows_f(imgV, reinterpret_cast<const char*>(f.v) + base, rs, nload, L, wave, nwaves, lane);
  dma_rows_f(imgG, reinterpret_cast<const char*>(a.dctx) + cbase, cs, nload, L, wave, nwaves, lane);
  constexpr float kLog2e = 1.4426950408889634f;
  for (int key = tid; key < 32 * nt; key += 64 * nwaves)
    kb[key] = key < L ? (HAS_KB ? f.key_bias[row0 + key] * kLog2e : 0.f) : -INFINITY;

  const int blk = wave;
  const bool active = blk * 32 < L;
  const int row = blk * 32 + l31;                 // this lane's QUERY in the prologue / epilogue, its KEY in the sweep
  const int rowc = row < L ? row : L - 1;
  uint4 of[4];
  float lse_q = INFINITY;
  if (active) {
    const char* op = reinterpret_cast<const char*>(f.ctx) + cbase + (int64_t)rowc * cs;
#pragma unroll
    for (int s = 0; s < 4; ++s) of[s] = *reinterpret_cast<const uint4*>(op + (2 * s + h) * 16);
    if (row < L) lse_q = f.lse[((int64_t)b * f.H + head) * f.L + row];
  }
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

  // D_q = <dO_q, O_q> of this wave's 32 queries, -lse log2 e: the per-row floats every key-block wave reads in the sweep
  if (active) {
    float d_q = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint4 g4 = *reinterpret_cast<const uint4*>(imgG + blk * 4096 + roff[s]);
      float gv[8], ov[8];
      unpack_chunk(g4, gv, bf16_t());
      unpack_chunk(of[s], ov, bf16_t());
#pragma unroll
      for (int e = 0; e < 8; ++e) d_q += gv[e] * ov[e];
    }
    d_q += __shfl_xor(d_q, 32, 64);
    if (h == 0) { lseA[row] = -lse_q * kLog2e; dA[row] = row < L ? -d_q : 0.f; }
  }
  // this wave's KEY block: K / V row fragments, K^T fragments (the A operand of the dQ product)
  uint4 xf[4], gf[4], ktf[2][2];
  if (active) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      xf[s] = *reinterpret_cast<const uint4*>(imgK + blk * 4096 + roff[s]);
      gf[s] = *reinterpret_cast<const uint4*>(imgV + blk * 4096 + roff[s]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) ktf[u][dt] = tr_frag(imgK + blk * 4096, u, dt);
  }
  // (the bias gradient of the query projection needs the K block once more at the very end: keep its 4 KiB in the wave's registers?
  //  no -- sum_q dQ = scale K^T c is formed from ktf: see below)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                                 // every wave holds its fragments: region X becomes the dQ image

  const float scale = f.scale;
  const float c = scale * kLog2e;
  const float kb_key = (row < LKP) ? kb[row] : -INFINITY;        // this lane's key: real bias (HAS_KB), 0, or -inf for keys >= L
  const bool ragged_blk = blk * 32 + 32 > L;                      // the wave whose block holds keys >= L masks them in the exponent
  f32x16_t dk[2], dv[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;
  float cs4[4] = {0.f, 0.f, 0.f, 0.f};            // c_k = sum over queries of dS[q][k]
  const uint32_t sbits_b = __float_as_uint(f.drop.scale);
  uint32_t* kwl = reinterpret_cast<uint32_t*>(wv + 2048);
  auto keep_word_of = [&](int t) -> uint32_t {
    const int qq = 32 * t + l31;
    return f.keep_bits[((row0 + (qq < L ? qq : L - 1)) * f.H + head) * keep_words + blk];
  };
  // dS^T tile of the wave: [key 32][query 32] bf16, 64-byte rows, 8-byte slot index XORed with (key >> 2) & 7
  const uint32_t ws_row = (uint32_t)l31 * 64u, ws_x = (uint32_t)((l31 >> 2) & 7);
  // ... read back as B-operand fragments (lane = query l31, keys 16 u + 4 h + {0..3} and + 8): transpose reads of 4 keys x 16 queries
  const uint32_t rs_slot = (uint32_t)(sub * 4 + (t16 & 3));
  uint32_t rs_off[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      const uint32_t r0 = (uint32_t)(16 * u + 8 * part + 4 * h);                  // first of the four key rows (multiple of 4)
      rs_off[u][part] = (r0 + (uint32_t)(t16 >> 2)) * 64u + ((rs_slot ^ ((r0 >> 2) & 7u)) << 3);
    }
  // dQ image: row q = 256 bytes, 16-byte chunk index (= d / 4) XORed with q & 15
  auto dq_addr = [&](int q, int chunk) -> char* { return dqimg + (uint32_t)q * 256u + ((uint32_t)(chunk ^ (q & 15)) << 4); };

  uint32_t kwq_next = 0u;
  int t = blk;
  if (DROP && active) kwq_next = keep_word_of(t);
#pragma unroll 1
  for (int step = 0; step < nt; ++step) {
    if (active) {
      f32x16_t sacc, pacc;
      if (DROP) {
        if (h == 0) kwl[l31] = kwq_next;
        int tn = t + 1; if (tn >= nt) tn -= nt;
        if (step + 1 < nt) kwq_next = keep_word_of(tn);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
      } else {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const float4 d4 = *reinterpret_cast<const float4*>(dA + 32 * t + 8 * qd + 4 * h);
          pacc[4 * qd] = d4.x; pacc[4 * qd + 1] = d4.y; pacc[4 * qd + 2] = d4.z; pacc[4 * qd + 3] = d4.w;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
      const char* qt = imgQ + t * 4096;
      const char* gt = imgG + t * 4096;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        mma32(sacc, *reinterpret_cast<const uint4*>(qt + roff[s]), xf[s], bf16_t());     // S[q][key]
        mma32(pacc, *reinterpret_cast<const uint4*>(gt + roff[s]), gf[s], bf16_t());     // dP[q][key] - D[q]
      }
      float p[16], ds[16];
      const bool with_kb = HAS_KB || ragged_blk;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 l4 = *reinterpret_cast<const float4*>(lseA + 32 * t + 8 * qd + 4 * h);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w};
        float nd[4] = {0.f, 0.f, 0.f, 0.f};
        uint32_t kwv[4] = {0u, 0u, 0u, 0u};
        if (DROP) {
          const float4 d4 = *reinterpret_cast<const float4*>(dA + 32 * t + 8 * qd + 4 * h);
          nd[0] = d4.x; nd[1] = d4.y; nd[2] = d4.z; nd[3] = d4.w;
          const uint4 w4 = *reinterpret_cast<const uint4*>(kwl + 8 * qd + 4 * h);
          kwv[0] = w4.x; kwv[1] = w4.y; kwv[2] = w4.z; kwv[3] = w4.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // (keys >= L: kb_key = -inf -> p = 0 exactly, so that neither dK / dV nor dQ sees them; queries >= L: lse = inf -> 0)
          float pe = __builtin_amdgcn_exp2f(fmaf(sacc[4 * qd + e], c, with_kb ? kb_key + lv[e] : lv[e]));
          if (CAUSAL && row > 32 * t + 8 * qd + 4 * h + e) pe = 0.f;
          if (DROP) {
            const float mk = __uint_as_float((uint32_t)__builtin_amdgcn_sbfe((int)kwv[e], l31, 1) & sbits_b);
            p[4 * qd + e] = pe * mk;
            ds[4 * qd + e] = pe * fmaf(pacc[4 * qd + e], mk, nd[e]);
          } else {
            p[4 * qd + e] = pe;
            ds[4 * qd + e] = pe * pacc[4 * qd + e];
          }
          cs4[e] += ds[4 * qd + e];
        }
      }
      uint4 pc[2], dc[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        pc[u].x = pack_bf16x2(p[8 * u + 0], p[8 * u + 1]);   dc[u].x = pack_bf16x2(ds[8 * u + 0], ds[8 * u + 1]);
        pc[u].y = pack_bf16x2(p[8 * u + 2], p[8 * u + 3]);   dc[u].y = pack_bf16x2(ds[8 * u + 2], ds[8 * u + 3]);
        pc[u].z = pack_bf16x2(p[8 * u + 4], p[8 * u + 5]);   dc[u].z = pack_bf16x2(ds[8 * u + 4], ds[8 * u + 5]);
        pc[u].w = pack_bf16x2(p[8 * u + 6], p[8 * u + 7]);   dc[u].w = pack_bf16x2(ds[8 * u + 6], ds[8 * u + 7]);
      }
      // dS^T -> the wave's LDS tile: lane (key, h) holds queries 8 qd + 4 h + {0..3} in dc[qd >> 1].{xy | zw}: slot 2 qd + h
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const uint2 v2 = (qd & 1) ? make_uint2(dc[qd >> 1].z, dc[qd >> 1].w) : make_uint2(dc[qd >> 1].x, dc[qd >> 1].y);
        *reinterpret_cast<uint2*>(wv + ws_row + (((uint32_t)(2 * qd + h) ^ ws_x) << 3)) = v2;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        mma32(dv[0], tr_frag(gt, u, 0), pc[u], bf16_t());      // dV^T[d][key] += dO^T . P
        mma32(dv[1], tr_frag(gt, u, 1), pc[u], bf16_t());
        mma32(dk[0], tr_frag(qt, u, 0), dc[u], bf16_t());      // dK^T[d][key] += Q^T . dS
        mma32(dk[1], tr_frag(qt, u, 1), dc[u], bf16_t());
      }
      // dQ_t^T[d][q] += K_w^T . dS^T: the tile's previous partial sum comes out of the LDS image (step 0: this wave is the first)
      const int q = 32 * t + l31;
      const bool qin = q < ra;
      f32x16_t dq[2];
      if (step > 0 && qin) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const float4 v4 = *reinterpret_cast<const float4*>(dq_addr(q, dt * 8 + 2 * qd + h));
            dq[dt][4 * qd] = v4.x; dq[dt][4 * qd + 1] = v4.y; dq[dt][4 * qd + 2] = v4.z; dq[dt][4 * qd + 3] = v4.w;
          }
      } else {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
      }
      __builtin_amdgcn_wave_barrier();              // (the dS^T tile is written by this wave only; LDS executes a wave's accesses in order)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint2 lo = tr4(wv + rs_off[u][0]), hi = tr4(wv + rs_off[u][1]);
        const uint4 dsf = make_uint4(lo.x, lo.y, hi.x, hi.y);
        mma32(dq[0], ktf[u][0], dsf, bf16_t());
        mma32(dq[1], ktf[u][1], dsf, bf16_t());
      }
      if (qin) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd)
            *reinterpret_cast<float4*>(dq_addr(q, dt * 8 + 2 * qd + h)) =
                make_float4(dq[dt][4 * qd], dq[dt][4 * qd + 1], dq[dt][4 * qd + 2], dq[dt][4 * qd + 3]);
      }
      ++t; if (t >= nt) t -= nt;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // the tiles rotate: next step another wave owns these rows
  }

  // ------------------------------------------------ outputs -------------------------------------------------------
  float* red_w = red + wave * (kOnceWave / 4);
  if (active) {
    // dK / dV of this wave's keys (lane = key)
    if (row < L) {
      bf16_t* dkp = reinterpret_cast<bf16_t*>(a.dk) + (row0 + row) * f.row_stride + head * 64;
      bf16_t* dvp = reinterpret_cast<bf16_t*>(a.dv) + (row0 + row) * f.row_stride + head * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          float vk[4], vv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) { vk[e] = dk[dt][4 * qd + e] * scale; vv[e] = dv[dt][4 * qd + e]; }
          st4(dkp + dt * 32 + 8 * qd + 4 * h, vk);
          st4(dvp + dt * 32 + 8 * qd + 4 * h, vv);
        }
    }
    // dQ of this wave's QUERY block out of the LDS image, row-coalesced: lane (row r8 = lane >> 3, c8 = lane & 7) takes 8 columns
    {
      const int r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int q = blk * 32 + it * 8 + r8;
        if (q < L) {
          const float4 x0 = *reinterpret_cast<const float4*>(dq_addr(q, 2 * c8));
          const float4 x1 = *reinterpret_cast<const float4*>(dq_addr(q, 2 * c8 + 1));
          const uint4 o = make_uint4(pack_bf16x2(x0.x * scale, x0.y * scale), pack_bf16x2(x0.z * scale, x0.w * scale),
                                     pack_bf16x2(x1.x * scale, x1.y * scale), pack_bf16x2(x1.z * scale, x1.w * scale));
          bf16_t* dqp = reinterpret_cast<bf16_t*>(a.dq) + (row0 + q) * f.row_stride + head * 64 + c8 * 8;
          if ((f.row_stride & 7) == 0 && ((uintptr_t)a.dq & 15) == 0) {
            *reinterpret_cast<uint4*>(dqp) = o;
          } else {
            *reinterpret_cast<uint2*>(dqp) = make_uint2(o.x, o.y);
            *reinterpret_cast<uint2*>(dqp + 4) = make_uint2(o.z, o.w);
          }
        }
      }
    }
  }
  if (want_db) {
    // per-wave shares of the three bias gradients into red_w[0..63] (q), [128..191] (k), [256..319] (v), column 1 at +64 as above:
    //   sum_q dQ = scale K^T c over this wave's KEYS; sum_k dK = 0 (see the header); sum_k dV = column sums of this wave's dV^T
    if (active) {      // (red_w is this wave's own LDS area -- its dS^T tile until now)
      float cc = (cs4[0] + cs4[1]) + (cs4[2] + cs4[3]);
      cc += __shfl_xor(cc, 32, 64);
      if (h == 0) red_w[384 + l31] = cc;
      __builtin_amdgcn_wave_barrier();
      // K^T c: the K block's image is gone (region X holds dQ) -- the product runs on the K^T fragments kept in registers
      {
        f32x16_t acc[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint4 bf = vec_frag(red_w + 384, u, h, l31);
          mma32(acc[0], ktf[u][0], bf, bf16_t());
          mma32(acc[1], ktf[u][1], bf, bf16_t());
        }
        if (l31 < 2) {
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
              *reinterpret_cast<float4*>(red_w + l31 * 64 + dt * 32 + 8 * qd + 4 * h) =
                  make_float4(acc[dt][4 * qd] * scale, acc[dt][4 * qd + 1] * scale, acc[dt][4 * qd + 2] * scale, acc[dt][4 * qd + 3] * scale);
        }
      }
      if (DROP) {
        // sum_k dV[k] = dO^T rowsum(P o M s): the row sums run over the keys = over the lanes here, so take the column sums of this
        // wave's dV^T accumulators instead (16 lanes by DPP, the other 16 by one exchange) -- once per kernel, dropout builds only
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = row16_sum(row < L ? dv[dt][r] : 0.f);
            v += __shfl_xor(v, 16, 64);
            if (l31 == 0) red_w[256 + dt * 32 + 8 * (r >> 2) + 4 * h + (r & 3)] = v;
          }
        if (lane < 64) red_w[320 + lane] = 0.f;
      } else {
        // rows of P sum to one: sum_k dV[k] = dO^T 1 over this wave's QUERY block (the dO image is still in LDS)
        if (h == 0) red_w[416 + l31] = row < L ? 1.f : 0.f;
        __builtin_amdgcn_wave_barrier();
        f32x16_t acc[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint4 bf = vec_frag(red_w + 416, u, h, l31);
          mma32(acc[0], tr_frag(imgG + blk * 4096, u, 0), bf, bf16_t());
          mma32(acc[1], tr_frag(imgG + blk * 4096, u, 1), bf, bf16_t());
        }
        if (l31 < 2) {
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
              *reinterpret_cast<float4*>(red_w + 256 + l31 * 64 + dt * 32 + 8 * qd + 4 * h) =
                  make_float4(acc[dt][4 * qd], acc[dt][4 * qd + 1], acc[dt][4 * qd + 2], acc[dt][4 * qd + 3]);
        }
      }
      if (lane < 64) { red_w[128 + lane] = 0.f; red_w[192 + lane] = 0.f; }     // dbk = 0 (both columns)
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int i = tid; i < 192; i += 64 * nwaves) {
      float s = 0.f;
      const int o = (i >> 6) * 128 + (i & 63);
      for (int w = 0; w < nt; ++w) s += red[w * (kOnceWave / 4) + o] + red[w * (kOnceWave / 4) + o + 64];
      a.db_part[((int64_t)b * 3 + (i >> 6)) * (f.H * 64) + head * 64 + (i & 63)] = s;
    }
  }
}

}  // namespace

// LDS of the fused backward for sequences of at most L tokens: four images of round_up(L, 8) rows, the pad behind the last
// one, three per-row float arrays, the per-wave bias-gradient areas
static int bwd_short_lds_bytes(int L) {
  const int nt = (L + 31) / 32, ra = (L + 7) / 8 * 8;
  return 4 * ra * 128 + (32 * nt - ra) * 128 + nt * (3 * 32 * 4 + kRedWave * 4);
}

bool attention_short_eligible(const AttnArgs& a, int dtype) {       // fused backward: four images of L rows in LDS, <= 9 waves
  return dtype == EZCLIP_BF16 && a.L <= 288 && a.L >= 1 && bwd_short_lds_bytes(a.L) <= 160 * 1024 && a.B <= 65535;
}

static int bwd_once_lds_bytes(int L) {
  const int nt = (L + 31) / 32, ra = (L + 7) / 8 * 8;
  return 2 * ra * 128 + 2 * (32 * nt - ra) * 128 + 2 * ra * 128 + nt * (3 * 32 * 4) + nt * kOnceWave;
}

// ezclip_debug_set(11, v): 1 (default) the score-tile-once kernel where it is the faster one -- up to 128 tokens (BERT's and the packed
// text tower's lengths: 0.222 vs 0.248 ms at 64 tokens, 0.422 vs 0.457 at 128, B = 1024 x 12 heads, same box; at 197 tokens it
// measures 1.11-1.15 against 1.03-1.12 ms and at 256 tokens 0.69 against 0.65: its LDS traffic per tile pair is 44 KB against the
// two-pass kernel's 36 KB -- the dQ read-modify-write and the dS^T round trip cost more than the second S / dP pass they replace --
// and its waves move in lockstep, one barrier per step; profiles/r4_attention_bwd_once_ab.log), 2 wherever it is eligible (tests,
// A/B), 0 never.
static int g_attn_bwd_once = 1;
void set_attention_bwd_once(int v) { g_attn_bwd_once = v; }

int attention_bwd_short(const AttnBwdArgs& a, hipStream_t stream) {
  const int nt = (a.f.L + 31) / 32, ra = (a.f.L + 7) / 8 * 8;
  const bool once = g_attn_bwd_once != 0 && nt <= (g_attn_bwd_once == 2 ? 8 : 4) && bwd_once_lds_bytes(a.f.L) <= 160 * 1024;
  const int bytes = once ? bwd_once_lds_bytes(a.f.L) : bwd_short_lds_bytes(a.f.L);
  static LdsOptIn lds_opt[16];
  const int vi = (a.f.key_bias != nullptr ? 1 : 0) + (a.f.causal ? 2 : 0) + (a.f.drop.thr != 0 ? 4 : 0);
  EZ_REQUIRE(a.f.drop.thr == 0 || (a.f.keep_bits != nullptr && a.f.keep_words == nt && nt <= 8),
             "attention_bwd_short: dropout needs the keep bits of the forward (keep_words = ceil(L / 32)) and L <= 256");
  using K = void (*)(AttnBwdArgs, int, int);
  static const K kerns[16] = {&attn_bwd_short_kernel<false, false, false>, &attn_bwd_short_kernel<true, false, false>,
                              &attn_bwd_short_kernel<false, true, false>,  &attn_bwd_short_kernel<true, true, false>,
                              &attn_bwd_short_kernel<false, false, true>,  &attn_bwd_short_kernel<true, false, true>,
                              &attn_bwd_short_kernel<false, true, true>,   &attn_bwd_short_kernel<true, true, true>,
                              &attn_bwd_once_kernel<false, false, false>,  &attn_bwd_once_kernel<true, false, false>,
                              &attn_bwd_once_kernel<false, true, false>,   &attn_bwd_once_kernel<true, true, false>,
                              &attn_bwd_once_kernel<false, false, true>,   &attn_bwd_once_kernel<true, false, true>,
                              &attn_bwd_once_kernel<false, true, true>,    &attn_bwd_once_kernel<true, true, true>};
  const K kern = kerns[vi + (once ? 8 : 0)];
  EZ_ENSURE_LDS(kern, lds_opt[vi + (once ? 8 : 0)], bytes);
  {
    ProfScope ps(PROF_ATTN, 10.0 * a.f.B * a.f.H * (double)a.f.L * a.f.L * 64, stream);   // 5 L x L x 64 products
    hipLaunchKernelGGL(kern, dim3(a.f.H, a.f.B), dim3(64 * nt), bytes, stream, a, nt, ra);
  }
  EZ_LAUNCH_CHECK();
  if (a.dbq != nullptr) {      // batch sum of the per-sample partials [B][3][D] -> the three bias gradients
    const int D = a.f.H * 64;
    const int rc = colsum3_add(a.db_part, 3 * D, a.f.B, 3 * D, a.dbq, a.dbk, a.dbv, D, EZCLIP_F32, stream);
    if (rc != EZ_OK) return rc;
  }
  return EZ_OK;
}

}  // namespace ezclip

#ifdef EZ_ATTN_BWD_TRACE
extern "C" int ezclip_dbg_attn_trace(void* host, size_t bytes) {
  const size_t all = sizeof(ezclip::g_attn_trace);
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ezclip::g_attn_trace), bytes < all ? bytes : all, 0, hipMemcpyDeviceToHost);
}
#endif
