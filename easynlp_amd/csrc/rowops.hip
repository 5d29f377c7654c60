// HBM-bound row-wise kernels: LayerNorm (fwd/bwd), embedding assembly for both
// towers, L2-normalise, dtype casts / weight re-packing, im2col for the
// stride==kernel patch-embedding conv.  One wavefront (64 lanes) per row, rows
// held in registers, wave-shuffle reductions, no LDS.
#include "ezclip_common.h"
#include "kernels.h"

namespace ezclip {
namespace {

constexpr int kMaxChunks = 4;  // 4-element chunks per lane: D <= 1024
constexpr int kRowsPerBlock = 4;

// Load a row of D (multiple of 4) elements into registers: lane owns chunks lane, lane+64, ...
template <typename T>
__device__ __forceinline__ void load_row(const T* x, int D, int lane, float (&v)[kMaxChunks][4]) {
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int col = (lane + c * 64) * 4;
    if (col < D) ld4(x + col, v[c]);
    else { v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f; }
  }
}

__device__ __forceinline__ void row_stats(const float (&v)[kMaxChunks][4], int D, int lane, float eps,
                                          float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
  mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    if ((lane + c * 64) * 4 < D) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[c][e] - mean; q += d * d; }
    }
  }
  const float var = wave_sum(q) / (float)D;
  rstd = rsqrtf(var + eps);
  // one Newton step: rsqrtf is ~1 ulp-ish approximate on AMD; keeps f32 parity tight
  rstd = rstd * (1.5f - 0.5f * (var + eps) * rstd * rstd);
}

template <typename T>
__device__ __forceinline__ void norm_store(const float (&v)[kMaxChunks][4], T* y, const float* g, const float* b,
                                           int D, int lane, float mean, float rstd) {
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int col = (lane + c * 64) * 4;
    if (col < D) {
      const float4 gv = *reinterpret_cast<const float4*>(g + col);
      const float4 bv = *reinterpret_cast<const float4*>(b + col);
      float o[4];
      o[0] = (v[c][0] - mean) * rstd * gv.x + bv.x;
      o[1] = (v[c][1] - mean) * rstd * gv.y + bv.y;
      o[2] = (v[c][2] - mean) * rstd * gv.z + bv.z;
      o[3] = (v[c][3] - mean) * rstd * gv.w + bv.w;
      st4(y + col, o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* x, int64_t xs, T* y, int64_t ys, const float* g,
                                                      const float* b, float eps, int rows, int D, float* mean_o,
                                                      float* rstd_o) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[kMaxChunks][4];
  load_row(x + (int64_t)row * xs, D, lane, v);
  float mean, rstd;
  row_stats(v, D, lane, eps, mean, rstd);
  norm_store(v, y + (int64_t)row * ys, g, b, D, lane, mean, rstd);
  if (mean_o != nullptr && lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
}

// bf16 rows whose width is a multiple of 8: a half-wave per row (32 lanes x 16-byte chunks, NC chunks per lane), each wave
// walks row pairs with the next pair's loads in flight under the current pair's arithmetic, gain / bias held in registers.
// (The one-row-per-wave kernel above: 8-byte loads, twelve bpermute steps per row, a workgroup launch per four rows --
// 2.7 TB/s on the [201 728 x 768] rows of the ViT; same arithmetic here, reductions over 32 lanes.)
template <int NC>
__global__ __launch_bounds__(256) void ln_fwd_pair_kernel(const bf16_t* x, int64_t xs, bf16_t* y, int64_t ys, const float* g,
                                                           const float* b, float eps, int rows, int D, float* mean_o,
                                                           float* rstd_o) {
  const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
  const int nw = gridDim.x * 4, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  float gv[NC][8], bv[NC][8];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = (hl + 32 * c) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { gv[c][e] = col < D ? g[col + e] : 0.f; bv[c][e] = col < D ? b[col + e] : 0.f; }
  }
  auto load = [&](int r, uint4 (&dst)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = (hl + 32 * c) * 8;
      dst[c] = (col < D && r < rows) ? *reinterpret_cast<const uint4*>(x + (int64_t)r * xs + col) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  uint4 cur[NC], nxt[NC];
  int row = gw * 2 + half;
  load(row, cur);
  const float inv_d = 1.0f / (float)D;
  for (; row - half < rows; row += 2 * nw) {
    load(row + 2 * nw, nxt);
    float v[NC][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      unpack_chunk(cur[c], v[c], bf16_t());
      s += ((v[c][0] + v[c][1]) + (v[c][2] + v[c][3])) + ((v[c][4] + v[c][5]) + (v[c][6] + v[c][7]));
    }
    const float mean = half_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if ((hl + 32 * c) * 8 < D) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; q += d * d; }
      }
    }
    const float var = half_sum(q) * inv_d;
    float rstd = rsqrtf(var + eps);
    rstd = rstd * (1.5f - 0.5f * (var + eps) * rstd * rstd);
    if (row < rows) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int col = (hl + 32 * c) * 8;
        if (col < D) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (v[c][e] - mean) * rstd * gv[c][e] + bv[c][e];
          *reinterpret_cast<uint4*>(y + (int64_t)row * ys + col) =
              make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
        }
      }
      if (mean_o != nullptr && hl == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) cur[c] = nxt[c];
  }
}

// stats[row] = (rstd, -mean * rstd): what the GEMM's folded-LayerNorm epilogue needs (GemmArgs::ln_stats)
template <typename T>
__global__ __launch_bounds__(256) void ln_stats_kernel(const T* x, int64_t xs, float eps, int rows, int D, float* stats) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[kMaxChunks][4];
  load_row(x + (int64_t)row * xs, D, lane, v);
  float mean, rstd;
  row_stats(v, D, lane, eps, mean, rstd);
  if (lane == 0) *reinterpret_cast<float2*>(stats + 2 * (int64_t)row) = make_float2(rstd, -mean * rstd);
}

// stats[row] = (rstd, -mean * rstd) from the per-slab (sum, sum of squares) partials a producing GEMM left behind
// (GemmArgs::rowstat_part, [rows][slabs][2]); summed in a fixed order, in double (E[x^2] - mean^2 cancels).
__global__ void ln_stats_finalize_kernel(const float* part, int slabs, int D, float eps, int rows, float* stats) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const float2* p = reinterpret_cast<const float2*>(part) + (int64_t)row * slabs;
  double s1 = 0.0, s2 = 0.0;
  for (int s = 0; s < slabs; ++s) {
    const float2 v = p[s];
    s1 += (double)v.x;
    s2 += (double)v.y;
  }
  const double mean = s1 / D;
  double var = s2 / D - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  *reinterpret_cast<float2*>(stats + 2 * (int64_t)row) = make_float2(rstd, (float)(-mean) * rstd);
}

// One wave per output row n:  Wf[n][k] = W[n][k] * g[k] (rounded to T),  c1[n] = sum_k Wf[n][k] (rounded values),
// c2[n] = sum_k b[k] * W[n][k] + bias[n].  Columns K..ldk-1 of Wf are zero.
template <typename T>
__global__ __launch_bounds__(256) void fold_ln_weight_kernel(const float* W, const float* g, const float* b,
                                                              const float* bias, int N, int K, T* Wf, int64_t ldk,
                                                              float* c1, float* c2) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float s1 = 0.f, s2 = 0.f;
  for (int k = lane; k < (int)ldk; k += 64) {
    float wf = 0.f;
    if (k < K) {
      const float w = W[(int64_t)n * K + k];
      wf = w * g[k];
      s2 += b[k] * w;
    }
    Elem<T>::st(Wf + (int64_t)n * ldk + k, wf);
    s1 += Elem<T>::ld(Wf + (int64_t)n * ldk + k);     // the value the MFMA will see
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) { c1[n] = s1; c2[n] = s2 + (bias ? bias[n] : 0.f); }
}

// LayerNorm backward, one row per wave at a time, two rows in flight per wave (loads of the second row are issued
// before the first is reduced), rows interleaved across ~16 waves per CU:
//   xhat = (x - mean) * rstd;  gy = dy * g
//   dx = rstd * (gy - mean(gy) - xhat * mean(gy * xhat))   [+ dres]
//   dg += dy * xhat; db += dy; optional dsum += dx_out (the bias gradient of the Linear that produced x's branch)
// Column sums live in registers, are combined across the block's four waves in LDS and leave as one hardware
// fp32 atomic per column per block.
template <typename T, int NC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* x, int64_t xs, const T* dy, int64_t dys, const float* g,
                                                      const float* mean_i, const float* rstd_i, T* dx, int64_t dxs,
                                                      const T* dres, int64_t drs, float* dg, float* db, float* dsum,
                                                      int rows, int D) {
  __shared__ float red[3][4][NC * 256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nwaves = gridDim.x * 4;
  const int gw = blockIdx.x * 4 + w;
  float ag[NC][4], ab[NC][4], ac[NC][4], gv[NC][4];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) ag[c][e] = ab[c][e] = ac[c][e] = 0.f;
    const int col = (lane + c * 64) * 4;
    if (col < D) ld4(g + col, gv[c]);
    else gv[c][0] = gv[c][1] = gv[c][2] = gv[c][3] = 0.f;
  }
  auto load = [&](int row, float (&xv)[NC][4], float (&dv)[NC][4], float (&rv)[NC][4], float& mean, float& rstd) {
    const bool ok = row < rows;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = (lane + c * 64) * 4;
      if (ok && col < D) {
        ld4(x + (int64_t)row * xs + col, xv[c]);
        ld4(dy + (int64_t)row * dys + col, dv[c]);
        if (dres != nullptr) ld4(dres + (int64_t)row * drs + col, rv[c]);
        else rv[c][0] = rv[c][1] = rv[c][2] = rv[c][3] = 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[c][e] = dv[c][e] = rv[c][e] = 0.f;
      }
    }
    mean = ok ? mean_i[row] : 0.f;
    rstd = ok ? rstd_i[row] : 0.f;
  };
  auto process = [&](int row, float (&xv)[NC][4], float (&dv)[NC][4], float (&rv)[NC][4], float mean, float rstd) {
    if (row >= rows) return;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {       // columns >= D hold zeros: they add nothing
        const float xh = (xv[c][e] - mean) * rstd;
        const float gy = dv[c][e] * gv[c][e];
        s1 += gy; s2 += gy * xh;
        ag[c][e] += dv[c][e] * xh; ab[c][e] += dv[c][e];
        xv[c][e] = xh; dv[c][e] = gy;
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = (lane + c * 64) * 4;
      if (col < D) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = rstd * (dv[c][e] - s1 - xv[c][e] * s2) + rv[c][e];
          ac[c][e] += o[e];
        }
        st4(dx + (int64_t)row * dxs + col, o);
      }
    }
  };
#ifndef EZ_LNBWD_UNPACKED
  // bf16 rows stay PACKED while in flight (2 VGPRs per 4 elements instead of 4), so four rows per wave fit where two
  // unpacked ones did -- twice the bytes in flight at the same occupancy.  Same per-row arithmetic in the same order:
  // results are bit-identical to the unpacked path (-DEZ_LNBWD_UNPACKED keeps it for A/B).  Measured (round 2, 201728
  // x 768 bf16): 302.6 -> 193.2 us per call, 3.07 -> 4.81 TB/s.
  if constexpr (sizeof(T) == 2) {
    constexpr int R = 4;
    auto load_raw = [&](int row, uint2 (&xr)[NC], uint2 (&dr)[NC], uint2 (&rr)[NC], float& mean, float& rstd) {
      const bool ok = row < rows;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int col = (lane + c * 64) * 4;
        xr[c] = dr[c] = rr[c] = make_uint2(0u, 0u);
        if (ok && col < D) {
          xr[c] = *reinterpret_cast<const uint2*>(x + (int64_t)row * xs + col);
          dr[c] = *reinterpret_cast<const uint2*>(dy + (int64_t)row * dys + col);
          if (dres != nullptr) rr[c] = *reinterpret_cast<const uint2*>(dres + (int64_t)row * drs + col);
        }
      }
      mean = ok ? mean_i[row] : 0.f;
      rstd = ok ? rstd_i[row] : 0.f;
    };
    auto unpack4 = [](const uint2& t, float (&v)[4]) {
      v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
      v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    };
    for (int row = gw; row < rows; row += R * nwaves) {
      uint2 xr[R][NC], dr[R][NC], rr[R][NC];
      float mr[R], sr[R];
#pragma unroll
      for (int k = 0; k < R; ++k) load_raw(row + k * nwaves, xr[k], dr[k], rr[k], mr[k], sr[k]);
#pragma unroll
      for (int k = 0; k < R; ++k) {
        float xv[NC][4], dv[NC][4], rv[NC][4];
#pragma unroll
        for (int c = 0; c < NC; ++c) { unpack4(xr[k][c], xv[c]); unpack4(dr[k][c], dv[c]); unpack4(rr[k][c], rv[c]); }
        process(row + k * nwaves, xv, dv, rv, mr[k], sr[k]);
      }
    }
  } else
#endif
  for (int row = gw; row < rows; row += 2 * nwaves) {
    float xa[NC][4], da[NC][4], ra[NC][4], xb[NC][4], dbb[NC][4], rb[NC][4];
    float ma, sa, mb, sb;
    load(row, xa, da, ra, ma, sa);
    load(row + nwaves, xb, dbb, rb, mb, sb);
    process(row, xa, da, ra, ma, sa);
    process(row + nwaves, xb, dbb, rb, mb, sb);
  }
  if (dg == nullptr && dsum == nullptr) return;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = c * 256 + lane * 4 + e;
      red[0][w][idx] = ag[c][e]; red[1][w][idx] = ab[c][e]; red[2][w][idx] = ac[c][e];
    }
  __syncthreads();
  for (int idx = threadIdx.x; idx < NC * 256; idx += 256) {
    if (idx >= D) break;
    if (dg != nullptr) {
      unsafeAtomicAdd(dg + idx, (red[0][0][idx] + red[0][1][idx]) + (red[0][2][idx] + red[0][3][idx]));
      unsafeAtomicAdd(db + idx, (red[1][0][idx] + red[1][1][idx]) + (red[1][2][idx] + red[1][3][idx]));
    }
    if (dsum != nullptr) unsafeAtomicAdd(dsum + idx, (red[2][0][idx] + red[2][1][idx]) + (red[2][2][idx] + red[2][3][idx]));
  }
}

template <typename T>
__global__ void cast_from_f32_kernel(const float* src, T* dst, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float v[4];
    ld4(src + i * 4, v);
    st4(dst + i * 4, v);
  }
}
template <typename T>
__global__ void cast_from_f32_tail(const float* src, T* dst, int64_t start, int64_t n) {
  const int64_t i = start + threadIdx.x;
  if (i < n) Elem<T>::st(dst + i, src[i]);
}
template <typename T>
__global__ void cast_to_f32_kernel(const T* src, float* dst, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = Elem<T>::ld(src + i);
}

// dst[c*ld + r] = src[r*C + c]; 32x32 tile through LDS (+1 pad)
template <typename T>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* src, int64_t sld, int R, int C, T* dst, int64_t ld) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? src[(int64_t)r * sld + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < ld) Elem<T>::st(dst + (int64_t)c * ld + r, r < R ? tile[tx][i] : 0.f);
  }
}

template <typename T>
__global__ void pad_cast_kernel(const float* src, int R, int C, T* dst, int64_t ld) {
  const int64_t total = (int64_t)R * ld;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld), c = (int)(i % ld);
    Elem<T>::st(dst + i, c < C ? src[(int64_t)r * C + c] : 0.f);
  }
}

// every weight's packed copies in one launch (CastJob, kernels.h): workgroup = one 64 x 64 tile of one job's source
template <typename T>
__global__ __launch_bounds__(256) void multi_cast_kernel(const CastJob* jobs, int n) {
  __shared__ float tile[64][65];
  int lo = 0, hi = n - 1;      // the job whose tile range holds blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const CastJob j = jobs[lo];
  const int t = blockIdx.x - j.tile0;
  if (j.kind == 1) {           // f32 vector copy, 4096 elements per tile
    float* d = static_cast<float*>(j.dst_s);
    for (int i = t * 4096 + threadIdx.x; i < j.C && i < (t + 1) * 4096; i += 256) d[i] = j.src[i];
    return;
  }
  const int r0 = (t / j.tiles_c) * 64, c0 = (t % j.tiles_c) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 64 columns x 4 rows per pass
  T* ds = static_cast<T*>(j.dst_s);
  T* dt = static_cast<T*>(j.dst_t);
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    const float v = (r < j.R && c < j.C) ? j.src[(int64_t)r * j.C + c] : 0.f;
    tile[i][tx] = v;
    if (ds != nullptr && r < j.R && c < j.ld_s) Elem<T>::st(ds + (int64_t)r * j.ld_s + c, v);
  }
  if (dt == nullptr) return;
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < j.C && r < j.ld_t) Elem<T>::st(dt + (int64_t)c * j.ld_t + r, tile[tx][i]);
  }
}

// out[(b*G*G + gy*G + gx) * Kpad + c*P*P + ky*P + kx] = px[b][c][gy*P+ky][gx*P+kx]
template <typename T>
__global__ void im2col_kernel(const float* px, T* out, int B, int R, int P, int G, int Kpad) {
  const int K = 3 * P * P;
  const int kq = Kpad / 4;
  const int64_t total = (int64_t)B * G * G * kq;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t rowi = i / kq;
    const int k0 = (int)(i % kq) * 4;
    const int b = (int)(rowi / (G * G));
    const int p = (int)(rowi % (G * G));
    const int gy = p / G, gx = p % G;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + e;
      if (k < K) {
        const int c = k / (P * P), rem = k % (P * P);
        const int ky = rem / P, kx = rem % P;
        v[e] = px[(((int64_t)b * 3 + c) * R + gy * P + ky) * R + gx * P + kx];
      } else v[e] = 0.f;
    }
    st4(out + rowi * Kpad + k0, v);
  }
}

// The same for patch sizes that are multiples of 8 without K padding (ViT-B/16: P = 16, K = 768) in 16-byte pieces: one thread
// = 8 consecutive kx of one (c, ky) patch row = two float4 loads (32 contiguous bytes of an image row) and one 16-byte bf16
// store; a wave writes 1 KiB of consecutive patch-matrix bytes and reads 32-byte runs of which four complete an image line
// across the neighbouring patch's wave.  (The scalar kernel above issued four 4-byte loads per thread.)
__global__ __launch_bounds__(256) void im2col8_bf16_kernel(const float* px, bf16_t* out, int B, int R, int P, int G) {
  const int K = 3 * P * P, k8 = K / 8, P8 = P / 8;
  const int64_t total = (int64_t)B * G * G * k8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t rowi = i / k8;
    const int j = (int)(i - rowi * k8);                 // 8-element piece of the patch row: (c, ky, kx / 8)
    const int b = (int)(rowi / (G * G)), p = (int)(rowi - (int64_t)b * G * G);
    const int gy = p / G, gx = p - gy * G;
    const int cky = j / P8, kx = (j - cky * P8) * 8;
    const int c = cky / P, ky = cky - c * P;
    const float* src = px + (((int64_t)b * 3 + c) * R + gy * P + ky) * R + gx * P + kx;
    const float4 a = *reinterpret_cast<const float4*>(src), d = *reinterpret_cast<const float4*>(src + 4);
    *reinterpret_cast<uint4*>(out + rowi * K + (int64_t)j * 8) =
        make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(d.x, d.y), pack_bf16x2(d.z, d.w));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void vit_assemble_ln_kernel(const T* patch, const float* cls, const float* pos,
                                                               const float* g, const float* b, float eps, T* x0, T* y,
                                                               float* mean_o, float* rstd_o, int B, int Lv, int W) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (row >= (int64_t)B * Lv) return;
  const int bi = (int)(row / Lv), t = (int)(row % Lv);
  float v[kMaxChunks][4];
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int col = (lane + c * 64) * 4;
    if (col < W) {
      float pv[4], sv[4];
      ld4(pos + (int64_t)t * W + col, pv);
      if (t == 0) ld4(cls + col, sv);
      else ld4(patch + ((int64_t)bi * (Lv - 1) + (t - 1)) * W + col, sv);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[c][e] = sv[e] + pv[e];
      if (x0 != nullptr) st4(x0 + row * W + col, v[c]);
    } else { v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f; }
  }
  float mean, rstd;
  row_stats(v, W, lane, eps, mean, rstd);
  norm_store(v, y + row * W, g, b, W, lane, mean, rstd);
  if (mean_o != nullptr && lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
}

// CLIP text transformer input (OPEN_CLIP.encode_text, modeling_openclip.py:355-357): x[b, t] = token_embedding[ids[b, t]]
// + positional_embedding[t], no LayerNorm.  One wave per row.
template <typename T>
__global__ __launch_bounds__(256) void clip_text_embed_kernel(const int64_t* ids, const float* tok, const float* pos, T* x,
                                                              int B, int L, int W, int vocab) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (row >= (int64_t)B * L) return;
  const int t = (int)(row % L);
  int64_t id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int col = (lane + c * 64) * 4;
    if (col < W) {
      float tv[4], pv[4], v[4];
      ld4(tok + id * W + col, tv);
      ld4(pos + (int64_t)t * W + col, pv);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = tv[e] + pv[e];
      st4(x + row * W + col, v);
    }
  }
}

// idx[b] = argmax_t ids[b, t] (first maximum, as torch.argmax): the EOT position (modeling_openclip.py:364-366);
// eot_id >= 0: first t with ids[b, t] == eot_id instead (wukong TextTransformer: x[(text == 102).nonzero()],
// modeling_wukong.py:349,359 -- one such token per row is the caller's contract; 0 when the row has none)
__global__ void argmax_rows_kernel(const int64_t* ids, int* idx, int B, int L, int64_t eot_id) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int64_t best = ids[(int64_t)b * L];
  if (eot_id >= 0) best = best == eot_id;
  int bi = 0;
  for (int t = 1; t < L; ++t) {
    int64_t v = ids[(int64_t)b * L + t];
    if (eot_id >= 0) v = v == eot_id;
    if (v > best) { best = v; bi = t; }
  }
  idx[b] = bi;
}

// mode 0 gather: dst[b] = src[b, idx[b]];  1 scatter: dst[b, idx[b]] = src[b];  2 scatter-add: dst[b, idx[b]] += src[b]
// (rows of W elements; idx == nullptr: row 0 of every sample)
template <typename T>
__global__ void gather_rows_kernel(const T* src, const int* idx, T* dst, int B, int L, int W, int mode) {
  const int64_t n = (int64_t)B * (W >> 2);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / (W >> 2)), c = (int)(i - (int64_t)b * (W >> 2)) * 4;
    const int64_t big = ((int64_t)b * L + (idx ? idx[b] : 0)) * W + c, small = (int64_t)b * W + c;
    float v[4];
    ld4(src + (mode ? small : big), v);
    if (mode == 2) {
      float o[4];
      ld4(dst + big, o);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += o[e];
    }
    st4(dst + (mode ? big : small), v);
  }
}

// pos_ids / type_ids / attn_mask: optional [B, L] int64 (RobertaEmbeddings: pad-aware position ids, roberta/modeling_roberta.py:
// 1497-1510; explicit token types and mask as the huggingface_clip branch passes them, appzoo/clip/model.py:131-133).
// Defaults: position t, type 0, mask = ids != 0 (chinese_clip, modeling_chineseclip.py:347).
template <typename T>
__global__ __launch_bounds__(256) void bert_embed_ln_kernel(const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids,
                                                             const int64_t* attn_mask, const float* word, const float* pos,
                                                             const float* type, const float* g, const float* b,
                                                             float eps, T* x0, T* y, float* mean_o, float* rstd_o,
                                                             float* key_bias, int B, int L, int Hd, int vocab, int max_pos,
                                                             int type_vocab, const int* rowmap, int packed_rows) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);      // output row
  if (row >= (rowmap ? (int64_t)packed_rows : (int64_t)B * L)) return;
  const int64_t src = rowmap ? (int64_t)rowmap[row] : row;                            // token b * L + t it comes from
  int64_t t = pos_ids ? pos_ids[src] : src % L;
  t = t < 0 ? 0 : (t >= max_pos ? max_pos - 1 : t);
  int64_t ty = type_ids ? type_ids[src] : 0;
  ty = ty < 0 ? 0 : (ty >= type_vocab ? type_vocab - 1 : ty);
  int64_t id = ids[src];
  const bool masked = attn_mask ? attn_mask[src] == 0 : id == 0;
  if (lane == 0) key_bias[row] = masked ? -10000.0f : 0.0f;
  if (id < 0) id = 0;
  if (id >= vocab) id = vocab - 1;
  float v[kMaxChunks][4];
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int col = (lane + c * 64) * 4;
    if (col < Hd) {
      float wv[4], pv[4], tv[4];
      ld4(word + id * Hd + col, wv);
      ld4(type + ty * Hd + col, tv);
      ld4(pos + t * Hd + col, pv);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[c][e] = (wv[e] + tv[e]) + pv[e];
      if (x0 != nullptr) st4(x0 + row * Hd + col, v[c]);
    } else { v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f; }
  }
  float mean, rstd;
  row_stats(v, Hd, lane, eps, mean, rstd);
  norm_store(v, y + row * Hd, g, b, Hd, lane, mean, rstd);
  if (mean_o != nullptr && lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
}

__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* x, float* out, float* inv_o, int B, int E) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (row >= B) return;
  float v[kMaxChunks][4];
  load_row(x + (int64_t)row * E, E, lane, v);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) s += v[c][e] * v[c][e];
  s = wave_sum(s);
  const float inv = 1.0f / sqrtf(s);
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int col = (lane + c * 64) * 4;
    if (col < E) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = v[c][e] * inv;
      st4(out + (int64_t)row * E + col, o);
    }
  }
  if (inv_o != nullptr && lane == 0) inv_o[row] = inv;
}

__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* y, const float* dy, const float* inv_i,
                                                          float* dx, int B, int E) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (row >= B) return;
  float yv[kMaxChunks][4], dv[kMaxChunks][4];
  load_row(y + (int64_t)row * E, E, lane, yv);
  load_row(dy + (int64_t)row * E, E, lane, dv);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) s += yv[c][e] * dv[c][e];
  s = wave_sum(s);
  const float inv = inv_i[row];
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int col = (lane + c * 64) * 4;
    if (col < E) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (dv[c][e] - yv[c][e] * s) * inv;
      st4(dx + (int64_t)row * E + col, o);
    }
  }
}


// out[c] += sum_r x[r][c]  (bias gradients).  Block = 4 waves x 256 columns (4 per lane); each wave strides
// the block's row range, LDS-combined, one atomic per column per block.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, int64_t ld, int rows, int cols, int rows_per_block,
                                                      float* out, float* out1, float* out2, int seg) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < cols) {
    for (int r = r0 + w; r < r1; r += 4) {
      float v[4];
      ld4(x + (int64_t)r * ld + c, v);
      a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[w][lane * 4 + e] = a[e];
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < cols) {      // seg > 0: three destinations, columns [0, seg) / [seg, 2 seg) / [2 seg, 3 seg)
    float* o = seg <= 0 || cc < seg ? out + cc : cc < 2 * seg ? out1 + (cc - seg) : out2 + (cc - 2 * seg);
    atomicAdd(o, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
  }
}

// out[t][c] += sum_b x[(b*T + t)*W + c]   (positional-embedding gradients; fixed order of additions: deterministic).
// Workgroup = one position x 256 columns x 8 waves; wave w takes the samples b = w (mod 8), four loads in flight, and the
// eight partial sums meet in LDS.  (One thread per (t, 4 columns) walking all B samples alone left the chip at 0.6 waves
// per SIMD: 0.86 TB/s on the ViT's [1024, 197, 768] block.)
template <typename T>
__global__ __launch_bounds__(512) void batch_sum_kernel(const T* x, int B, int Tn, int W, float* out) {
  __shared__ float red[8][256];
  const int t = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.y * 256 + lane * 4;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < W) {
    const T* p = x + (int64_t)t * W + c;
    const int64_t bs = (int64_t)Tn * W;
    int b = w;
    for (; b + 24 < B; b += 32) {
      float v0[4], v1[4], v2[4], v3[4];
      ld4(p + b * bs, v0); ld4(p + (b + 8) * bs, v1); ld4(p + (b + 16) * bs, v2); ld4(p + (b + 24) * bs, v3);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
    }
    for (; b < B; b += 8) {
      float v[4];
      ld4(p + b * bs, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[w][lane * 4 + e] = a[e];
  __syncthreads();
  const int cc = blockIdx.y * 256 + threadIdx.x;
  if (threadIdx.x < 256 && cc < W) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][threadIdx.x];
    out[(int64_t)t * W + cc] += s;
  }
}

// compact the patch rows of d_x0 [B, Lv, W] into d_pemb [B*(Lv-1), W]
template <typename T>
__global__ void vit_gather_patch_rows_kernel(const T* dx0, T* dpemb, int B, int Lv, int W) {
  const int64_t n4 = (int64_t)B * (Lv - 1) * (W / 4);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / (W / 4);
    const int c = (int)(i % (W / 4)) * 4;
    const int64_t b = row / (Lv - 1), p = row % (Lv - 1);
    float v[4];
    ld4(dx0 + ((b * Lv) + 1 + p) * W + c, v);
    st4(dpemb + row * W + c, v);
  }
}

// word-embedding gradient: dword[ids[r]] += dx0[r]; rows with id == padding_idx (0) get no gradient
// (nn.Embedding(padding_idx=0), bert/modeling_bert.py:77)
template <typename T>
__global__ __launch_bounds__(256) void bert_word_grad_kernel(const int64_t* ids, const T* dx0, float* dword, int64_t rows,
                                                              int Hd, int vocab, int64_t pad_id, const int* rowmap, int seq_len) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (row >= rows) return;
  // packed rows: row r holds token rowmap[r] = b * seq_len + t; ids == nullptr: the table index is the position t itself
  const int64_t src = rowmap ? (int64_t)rowmap[row] : row;
  const int64_t id = ids ? ids[src] : src % seq_len;
  if (id < 0 || id >= vocab || id == pad_id) return;     // nn.Embedding(padding_idx=pad_id): no gradient for that row
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int col = (lane + c * 64) * 4;
    if (col < Hd) {
      float v[4];
      ld4(dx0 + row * Hd + col, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(dword + id * Hd + col + e, v[e]);
    }
  }
}

__global__ void add_inplace_kernel(float* dst, const float* src, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

inline int grid_for(int64_t work, int block) {
  int64_t g = (work + block - 1) / block;
  const int64_t cap = 256 * 8;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

#define EZ_DISPATCH_T(dtype, ...)                                  \
  do {                                                             \
    if ((dtype) == EZCLIP_F32) { using T = float; __VA_ARGS__; }   \
    else if ((dtype) == EZCLIP_BF16) { using T = bf16_t; __VA_ARGS__; } \
    else { set_error("bad dtype %d", (int)(dtype)); return EZ_ERR_INVALID; } \
  } while (0)

int layernorm_fwd(const void* x, int64_t xs, void* y, int64_t ys, const float* g, const float* b, float eps,
                  int rows, int D, int dtype, float* mean, float* rstd, hipStream_t stream) {
  EZ_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 256 * kMaxChunks, "layernorm_fwd: D=%d must be a multiple of 4 and <= 1024", D);
  EZ_REQUIRE(xs % 4 == 0 && ys % 4 == 0, "layernorm_fwd: row strides must be multiples of 4 elements");
  const int blocks = (rows + kRowsPerBlock - 1) / kRowsPerBlock;
  ProfScope ps(PROF_ROWOP, 2.0 * rows * (double)D * dtype_size(dtype), stream);   // bytes: read + write
  if (dtype == EZCLIP_BF16 && D % 8 == 0 && xs % 8 == 0 && ys % 8 == 0 && rows >= 64 && ((uintptr_t)x % 16) == 0 &&
      ((uintptr_t)y % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)b % 16) == 0) {
    // (rows >= 64: the small CLS-row calls keep the one-wave-per-row kernel)   4 waves x 2 rows per workgroup and trip
    const int nc = (D + 255) / 256;
    const int resident = 256 * (nc >= 3 ? 4 : 8);      // workgroups the chip holds at this variant's register count
    int wgs = (rows + 7) / 8;
    if (wgs > resident) wgs = resident;
    auto launch = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, stream, (const bf16_t*)x, xs, (bf16_t*)y, ys, g, b, eps, rows, D, mean, rstd);
    };
    if (nc == 1) launch(ln_fwd_pair_kernel<1>);
    else if (nc == 2) launch(ln_fwd_pair_kernel<2>);
    else if (nc == 3) launch(ln_fwd_pair_kernel<3>);
    else launch(ln_fwd_pair_kernel<4>);
    EZ_LAUNCH_CHECK();
    return EZ_OK;
  }
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((ln_fwd_kernel<T>), dim3(blocks), dim3(256), 0, stream, (const T*)x, xs,
                                          (T*)y, ys, g, b, eps, rows, D, mean, rstd));
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int layernorm_bwd(const void* x, int64_t xs, const void* dy, int64_t dys, const float* g, const float* mean,
                  const float* rstd, void* dx, int64_t dxs, const void* dres, int64_t drs, float* dg, float* db,
                  int rows, int D, int dtype, hipStream_t stream, float* dsum) {
  EZ_REQUIRE(rows > 0 && D % 4 == 0 && D <= 256 * kMaxChunks, "layernorm_bwd: D=%d unsupported", D);
  EZ_REQUIRE(xs % 4 == 0 && dys % 4 == 0 && dxs % 4 == 0 && drs % 4 == 0, "layernorm_bwd: row strides must be multiples of 4");
  // ~16 waves per CU, rows interleaved across them
  int blocks = (rows + 3) / 4;
  if (blocks > 1024) blocks = 1024;
  const int nc = (D + 255) / 256;
#define EZ_LNB(NC)                                                                                                  \
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((ln_bwd_kernel<T, NC>), dim3(blocks), dim3(256), 0, stream, (const T*)x, xs, \
                                          (const T*)dy, dys, g, mean, rstd, (T*)dx, dxs, (const T*)dres, drs, dg, db, \
                                          dsum, rows, D))
  if (nc == 1) EZ_LNB(1); else if (nc == 2) EZ_LNB(2); else if (nc == 3) EZ_LNB(3); else EZ_LNB(4);
#undef EZ_LNB
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int layernorm_row_stats(const void* x, int64_t xs, float eps, int rows, int D, int dtype, float* stats,
                        hipStream_t stream) {
  EZ_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 256 * kMaxChunks && xs % 4 == 0, "layernorm_row_stats: D=%d unsupported", D);
  const int blocks = (rows + kRowsPerBlock - 1) / kRowsPerBlock;
  ProfScope ps(PROF_ROWOP, 1.0 * rows * (double)D * dtype_size(dtype), stream);   // bytes: one read
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((ln_stats_kernel<T>), dim3(blocks), dim3(256), 0, stream, (const T*)x, xs, eps,
                                          rows, D, stats));
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int layernorm_stats_finalize(const float* part, int slabs, int D, float eps, int rows, float* stats, hipStream_t stream) {
  EZ_REQUIRE(rows > 0 && slabs > 0 && D == 64 * slabs, "layernorm_stats_finalize: D=%d must be 64 * slabs (%d)", D, slabs);
  ProfScope ps(PROF_ROWOP, 8.0 * rows * (double)slabs, stream);
  // (a 16-lanes-per-row variant with coalesced partial reads and a butterfly in double measured 15.6 us against 13.2)
  hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, part, slabs, D, eps, rows, stats);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int fold_ln_weight(const float* W, const float* g, const float* b, const float* bias, int N, int K, void* Wf, int64_t ldk,
                   float* c1, float* c2, int dtype, hipStream_t stream) {
  EZ_REQUIRE(N > 0 && K > 0 && ldk >= K, "fold_ln_weight: bad shape");
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((fold_ln_weight_kernel<T>), dim3((N + 3) / 4), dim3(256), 0, stream, W, g, b, bias,
                                          N, K, (T*)Wf, ldk, c1, c2));
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int cast_from_f32(const float* src, void* dst, int64_t n, int dtype, hipStream_t stream) {
  if (n <= 0) return EZ_OK;
  const int64_t n4 = n / 4;
  EZ_DISPATCH_T(dtype, {
    if (n4 > 0) hipLaunchKernelGGL((cast_from_f32_kernel<T>), dim3(grid_for(n4, 256)), dim3(256), 0, stream, src, (T*)dst, n4);
    if (n4 * 4 < n) hipLaunchKernelGGL((cast_from_f32_tail<T>), dim3(1), dim3(64), 0, stream, src, (T*)dst, n4 * 4, n);
  });
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int cast_to_f32(const void* src, float* dst, int64_t n, int dtype, hipStream_t stream) {
  if (n <= 0) return EZ_OK;
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((cast_to_f32_kernel<T>), dim3(grid_for(n, 256)), dim3(256), 0, stream,
                                          (const T*)src, dst, n));
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int transpose_cast(const float* src, int64_t src_ld, int R, int C, void* dst, int64_t ld, int dtype, hipStream_t stream) {
  EZ_REQUIRE(R > 0 && C > 0 && ld >= R && src_ld >= C, "transpose_cast: bad shape");
  dim3 grid((C + 31) / 32, (int)((ld + 31) / 32));
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((transpose_cast_kernel<T>), grid, dim3(256), 0, stream, src, src_ld, R, C, (T*)dst, ld));
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int cast_jobs_finalize(CastJob* jobs, int n) {
  int total = 0;
  for (int i = 0; i < n; ++i) {
    CastJob& j = jobs[i];
    j.tile0 = total;
    if (j.kind == 1) { j.tiles_c = 1; total += (j.C + 4095) / 4096; continue; }
    // the tile grid covers the pad columns of both copies too (they are zero-filled)
    const int cols = j.dst_s != nullptr && j.ld_s > j.C ? j.ld_s : j.C;
    const int rows = j.dst_t != nullptr && j.ld_t > j.R ? j.ld_t : j.R;
    j.tiles_c = (cols + 63) / 64;
    total += j.tiles_c * ((rows + 63) / 64);
  }
  return total;
}

int cast_jobs_run(const CastJob* jobs_dev, int n, int total_tiles, int dtype, hipStream_t stream) {
  if (n <= 0 || total_tiles <= 0) return EZ_OK;
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((multi_cast_kernel<T>), dim3(total_tiles), dim3(256), 0, stream, jobs_dev, n));
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int pad_cast(const float* src, int R, int C, void* dst, int64_t ld, int dtype, hipStream_t stream) {
  EZ_REQUIRE(R > 0 && C > 0 && ld >= C, "pad_cast: bad shape");
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((pad_cast_kernel<T>), dim3(grid_for((int64_t)R * ld, 256)), dim3(256), 0,
                                          stream, src, R, C, (T*)dst, ld));
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int im2col_patches(const float* pixels, void* out, int B, int R, int P, int Kpad, int dtype, hipStream_t stream) {
  const int G = R / P;
  EZ_REQUIRE(B > 0 && G > 0 && Kpad % 4 == 0 && Kpad >= 3 * P * P, "im2col: bad shape");
  if (dtype == EZCLIP_BF16 && P % 8 == 0 && Kpad == 3 * P * P && R % 4 == 0 && ((uintptr_t)pixels % 16) == 0 && ((uintptr_t)out % 16) == 0) {
    const int64_t work8 = (int64_t)B * G * G * (Kpad / 8);
    hipLaunchKernelGGL(im2col8_bf16_kernel, dim3(grid_for(work8, 256)), dim3(256), 0, stream, pixels, (bf16_t*)out, B, R, P, G);
    EZ_LAUNCH_CHECK();
    return EZ_OK;
  }
  const int64_t work = (int64_t)B * G * G * (Kpad / 4);
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((im2col_kernel<T>), dim3(grid_for(work, 256)), dim3(256), 0, stream, pixels,
                                          (T*)out, B, R, P, G, Kpad));
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int vit_assemble_ln(const void* patch, const float* cls, const float* pos, const float* g, const float* b, float eps,
                    void* x0, void* y, float* mean, float* rstd, int B, int Lv, int W, int dtype,
                    hipStream_t stream) {
  EZ_REQUIRE(W % 4 == 0 && W <= 256 * kMaxChunks, "vit_assemble_ln: width %d unsupported", W);
  const int64_t rows = (int64_t)B * Lv;
  const int blocks = (int)((rows + kRowsPerBlock - 1) / kRowsPerBlock);
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((vit_assemble_ln_kernel<T>), dim3(blocks), dim3(256), 0, stream,
                                          (const T*)patch, cls, pos, g, b, eps, (T*)x0, (T*)y, mean, rstd, B, Lv, W));
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int clip_text_embed(const int64_t* ids, const float* tok, const float* pos, void* x, int* eot_idx, int B, int L, int W, int vocab,
                    int64_t eot_id, int dtype, hipStream_t stream) {
  EZ_REQUIRE(W % 4 == 0 && W <= 256 * kMaxChunks, "clip_text_embed: width %d unsupported", W);
  const int64_t rows = (int64_t)B * L;
  const int blocks = (int)((rows + kRowsPerBlock - 1) / kRowsPerBlock);
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((clip_text_embed_kernel<T>), dim3(blocks), dim3(256), 0, stream, ids, tok, pos, (T*)x, B, L,
                                          W, vocab));
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, ids, eot_idx, B, L, eot_id);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int gather_rows(const void* src, const int* idx, void* dst, int B, int L, int W, int mode, int dtype, hipStream_t stream) {
  EZ_REQUIRE(W % 4 == 0, "gather_rows: width %d must be a multiple of 4", W);
  const int64_t n = (int64_t)B * (W >> 2);
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((gather_rows_kernel<T>), dim3(blocks), dim3(256), 0, stream, (const T*)src, idx, (T*)dst, B,
                                          L, W, mode));
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int bert_embed_ln(const int64_t* ids, const float* word, const float* pos, const float* type, const float* g,
                  const float* b, float eps, void* x0, void* y, float* mean, float* rstd, float* key_bias, int B,
                  int L, int Hd, int vocab, int dtype, hipStream_t stream, const int64_t* pos_ids, const int64_t* type_ids,
                  const int64_t* attn_mask, int max_pos, int type_vocab, const int* rowmap, int packed_rows) {
  EZ_REQUIRE(Hd % 4 == 0 && Hd <= 256 * kMaxChunks, "bert_embed_ln: hidden %d unsupported", Hd);
  EZ_REQUIRE(rowmap == nullptr || packed_rows > 0, "bert_embed_ln: empty packed batch");
  const int64_t rows = rowmap ? (int64_t)packed_rows : (int64_t)B * L;
  const int blocks = (int)((rows + kRowsPerBlock - 1) / kRowsPerBlock);
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((bert_embed_ln_kernel<T>), dim3(blocks), dim3(256), 0, stream, ids, pos_ids, type_ids,
                                          attn_mask, word, pos, type, g, b, eps, (T*)x0, (T*)y, mean, rstd, key_bias, B, L, Hd,
                                          vocab, max_pos, type_vocab, rowmap, packed_rows));
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int l2_normalize_fwd(const float* x, float* out, float* inv_norm, int B, int E, hipStream_t stream) {
  EZ_REQUIRE(E % 4 == 0 && E <= 256 * kMaxChunks, "l2_normalize: E=%d unsupported", E);
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((B + kRowsPerBlock - 1) / kRowsPerBlock), dim3(256), 0, stream, x, out,
                     inv_norm, B, E);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int l2_normalize_bwd(const float* y, const float* dy, const float* inv_norm, float* dx, int B, int E,
                     hipStream_t stream) {
  EZ_REQUIRE(E % 4 == 0 && E <= 256 * kMaxChunks, "l2_normalize: E=%d unsupported", E);
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((B + kRowsPerBlock - 1) / kRowsPerBlock), dim3(256), 0, stream, y, dy,
                     inv_norm, dx, B, E);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int colsum_add(const void* x, int64_t ld, int rows, int cols, float* out, int dtype, hipStream_t stream) {
  return colsum3_add(x, ld, rows, cols, out, nullptr, nullptr, 0, dtype, stream);
}

// columns [0, seg), [seg, 2 seg), [2 seg, 3 seg) of x summed into three separate vectors (seg = 0: one vector of `cols`)
int colsum3_add(const void* x, int64_t ld, int rows, int cols, float* out, float* out1, float* out2, int seg, int dtype,
                hipStream_t stream) {
  EZ_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0, "colsum_add: cols/ld must be multiples of 4");
  EZ_REQUIRE(seg == 0 || (cols == 3 * seg && out1 != nullptr && out2 != nullptr), "colsum3_add: cols must be 3 x seg");
  const int bx = (cols + 255) / 256;
  int by = (1024 + bx - 1) / bx;
  if (by > (rows + 63) / 64) by = (rows + 63) / 64;
  if (by < 1) by = 1;
  const int rpb = (rows + by - 1) / by;
  by = (rows + rpb - 1) / rpb;
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T>), dim3(bx, by), dim3(256), 0, stream, (const T*)x, ld, rows,
                                          cols, rpb, out, out1, out2, seg));
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int batch_sum_add(const void* x, int B, int Tn, int t_count, int W, float* out, int dtype, hipStream_t stream) {
  EZ_REQUIRE(B > 0 && Tn > 0 && t_count > 0 && t_count <= Tn && W % 4 == 0, "batch_sum_add: bad shape");
  dim3 grid(t_count, (W + 255) / 256);
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((batch_sum_kernel<T>), grid, dim3(512), 0, stream, (const T*)x, B, Tn, W, out));
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int vit_gather_patch_rows(const void* dx0, void* dpemb, int B, int Lv, int W, int dtype, hipStream_t stream) {
  const int64_t n4 = (int64_t)B * (Lv - 1) * (W / 4);
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((vit_gather_patch_rows_kernel<T>), dim3(grid_for(n4, 256)), dim3(256), 0, stream,
                                          (const T*)dx0, (T*)dpemb, B, Lv, W));
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int bert_word_grad(const int64_t* ids, const void* dx0, float* dword, int64_t rows, int Hd, int vocab, int dtype,
                   hipStream_t stream, int64_t pad_id, const int* rowmap, int seq_len) {
  EZ_REQUIRE(ids != nullptr || (rowmap != nullptr && seq_len > 0), "bert_word_grad: position mode needs the row map");
  const int blocks = (int)((rows + kRowsPerBlock - 1) / kRowsPerBlock);
  EZ_DISPATCH_T(dtype, hipLaunchKernelGGL((bert_word_grad_kernel<T>), dim3(blocks), dim3(256), 0, stream, ids,
                                          (const T*)dx0, dword, rows, Hd, vocab, pad_id, rowmap, seq_len));
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

__global__ void add_cols_kernel(float* dst, int64_t ldd, const float* src, int64_t lds, int rows, int cols) {
  const int64_t n = (int64_t)rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    dst[(int64_t)r * ldd + c] += src[(int64_t)r * lds + c];
  }
}

int add_cols_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int rows, int cols, hipStream_t stream) {
  hipLaunchKernelGGL(add_cols_kernel, dim3(grid_for((int64_t)rows * cols, 256)), dim3(256), 0, stream, dst, ldd, src, lds, rows,
                     cols);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

// ------------------------------------------------------------------------------------------------
// y = dropout(x) [+ res]   (BertSelfOutput / BertOutput: LayerNorm(dropout(dense(.)) + input), modeling_bert.py:264-267,
// 342-345; embeddings: dropout(LayerNorm(.)) :127-128; the same kernel masks the gradients in the backward pass).
// One thread = 4 consecutive columns of one row = one Philox call (dropout.h).  In-place use (y == x) is fine.
template <typename T>
__global__ void dropout_rows_kernel(const T* x, int64_t xs, const T* res, int64_t rs, T* y, int64_t ys, int rows, int D,
                                    DropCfg d, const int* rowmap) {
  const int dq = D >> 2;
  const int64_t n = (int64_t)rows * dq;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / dq), cq = (int)(i - (int64_t)r * dq);
    float v[4];
    ld4(x + (int64_t)r * xs + 4 * cq, v);
    // packed batches: the mask row is the token's row in the PADDED batch (the same decisions as the padded run)
    const uint4 w = drop_words(d, (uint32_t)(rowmap ? rowmap[r] : r), (uint32_t)cq);
    v[0] = w.x >= d.thr ? v[0] * d.scale : 0.f;
    v[1] = w.y >= d.thr ? v[1] * d.scale : 0.f;
    v[2] = w.z >= d.thr ? v[2] * d.scale : 0.f;
    v[3] = w.w >= d.thr ? v[3] * d.scale : 0.f;
    if (res != nullptr) {
      float q[4];
      ld4(res + (int64_t)r * rs + 4 * cq, q);
      // (the reference rounds dropout's output to the tensor dtype before the residual add)
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = bf16_to_f32(f32_to_bf16(v[e]));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += q[e];
    }
    st4(y + (int64_t)r * ys + 4 * cq, v);
  }
}

int dropout_rows(const void* x, int64_t xs, const void* res, int64_t rs, void* y, int64_t ys, int rows, int D,
                 const DropCfg& d, int dtype, hipStream_t stream, const int* rowmap) {
  EZ_REQUIRE(D % 4 == 0 && xs % 4 == 0 && ys % 4 == 0 && (res == nullptr || rs % 4 == 0), "dropout_rows: D and strides must be multiples of 4");
  EZ_REQUIRE(d.thr != 0, "dropout_rows: called with dropout off");
  const int64_t n = (int64_t)rows * (D >> 2);
  if (dtype == EZCLIP_F32)
    hipLaunchKernelGGL((dropout_rows_kernel<float>), dim3(grid_for(n, 256)), dim3(256), 0, stream, (const float*)x, xs,
                       (const float*)res, rs, (float*)y, ys, rows, D, d, rowmap);
  else
    hipLaunchKernelGGL((dropout_rows_kernel<bf16_t>), dim3(grid_for(n, 256)), dim3(256), 0, stream, (const bf16_t*)x, xs,
                       (const bf16_t*)res, rs, (bf16_t*)y, ys, rows, D, d, rowmap);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

// keep[r][c] = 1 / 0 for an arbitrary [rows, cols] site (tests feed these masks to the oracle); words (optional):
// the raw 32-bit Philox outputs (known-answer check of the generator).
__global__ void dropout_mask_kernel(uint8_t* keep, uint32_t* words, int rows, int cols, DropCfg d) {
  const int64_t n = (int64_t)rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    const uint32_t w = drop_word(d, (uint32_t)r, (uint32_t)c);
    if (keep) keep[i] = w >= d.thr ? 1 : 0;
    if (words) words[i] = w;
  }
}

int dropout_mask(uint8_t* keep, uint32_t* words, int rows, int cols, const DropCfg& d, hipStream_t stream) {
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for((int64_t)rows * cols, 256)), dim3(256), 0, stream, keep, words, rows,
                     cols, d);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int add_inplace_f32(float* dst, const float* src, int64_t n, hipStream_t stream) {
  hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, dst, src, n);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace ezclip
