// ModifiedResNet image tower, TRAINING path: the row-matrix kernels around the convolutions (the convolutions themselves are the
// GEMMs of gemm.hip / gemm8p.hip).
//
// Reference: easynlp/modelzoo/models/clip/modeling_chineseclip.py -- Bottleneck :27-74, ModifiedResNet :110-167 with nn.BatchNorm2d in
// train() mode (batch statistics, running statistics moved by momentum 0.1), and torch autograd through it (core/trainer.py:658-661).
// Restated step by step by the tests' CPU restatement of the tower (train_step_grads_by_steps; the data-layout conventions of the packing
// kernels in pack_conv3x3_dgrad / im2col3x3_nhwc / unpack_wgrad3x3): every kernel below is one of those steps.
//
// Layout as in resnet.hip: activations NHWC, [rows = B * H * W, cp] row-major in the compute dtype, channels padded to a multiple of 64
// with EXACT zeros (every kernel here keeps them zero: padded channels get scale = shift = 0 and gradient 0).
//
//   bn_train_fwd     z -> (sum z, sum z^2 per channel: two-stage, fixed order, bit-reproducible) -> mean, biased var, rstd;
//                    running statistics moved with the UNBIASED variance; y = [relu](z * scale + shift [+ residual])
//   bn_train_bwd     g = dy * [y > 0];  (sum g, sum g * xhat) -> dgamma, dbeta;  dz = gamma * rstd * (g - sum g / n - xhat * sum(g xhat) / n);
//                    optionally dres = g (the gradient of the residual input)
//   avgpool2_bwd     dx[b, y, x, :] = dy[b, y / 2, x / 2, :] / 4
//   im2col3x3        col[m][(ky * 3 + kx) * cp + c] = x[pixel m shifted by (ky - 1, kx - 1)][c], zeros outside the image
//   pack_conv_dgrad  Wd[c][(ky * k + kx) * opad + o] = W[o][c][k - 1 - ky][k - 1 - kx]     (k = 1: the transpose)
//   unpack_wgrad     dW[o][c][ky][kx] (+)= dWp[o][(ky * k + kx) * cp + c]                  (dWp = dz^T . im2col(x): gemm_tn's output)
#include "../../include/ezclip.h"
#include "ezclip_common.h"
#include "kernels.h"

namespace ezclip {
namespace {

constexpr int kMomentSlabRows = 64;     // rows per slab at least; at most kMaxSlabs slabs
constexpr int kMaxSlabs = 1024;

// one 16-byte chunk of a row (4 floats / 8 bf16) <-> floats; per-channel float vectors for the same channels
__device__ __forceinline__ uint4 pack_chunk16(const float (&v)[4], float) {
  return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}
__device__ __forceinline__ uint4 pack_chunk16(const float (&v)[8], bf16_t) {
  return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}
template <int V>
__device__ __forceinline__ void ld_vec(const float* __restrict__ p, float (&v)[V]) {
#pragma unroll
  for (int k = 0; k < V; k += 4) {
    const float4 t = *reinterpret_cast<const float4*>(p + k);
    v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
  }
}

// ---- per-channel moments -----------------------------------------------------------------------------------------------------
// grid (slabs, ceil(chunks per row / 64)): a block owns one slab of rows and up to 64 16-byte channel chunks (4 floats / 8 bf16 each);
// its 256 threads are `cl` chunk lanes x `rl` row lanes, cl = min(chunks per row, 64), rl = 256 / cl -- at 64 channels (the stem and
// layer1, the largest row counts) that is 8 x 32 instead of the first version's 16 x 4 with three quarters of the block idle.
// MODE 0: (u, v) = (z, z^2).   MODE 1: g = dy * [y > 0] (y == nullptr: no mask), xhat = (z - mean) * rstd: (u, v) = (g, g * xhat).
// part[slab][2][cp] float; the row lanes are combined in a fixed order (bit-reproducible).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void rn_moments_kernel(const T* __restrict__ z, const T* __restrict__ dy, const T* __restrict__ y,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd, int64_t rows,
                                                          int cp, int rows_per_slab, float* __restrict__ part) {
  constexpr int V = Elem<T>::kPerChunk;
  __shared__ float red[2][256 * V];                 // [u | v][row lane][chunk lane][V]
  const int nch = cp / V;                           // chunks per row
  const int cl = nch < 64 ? nch : 64;               // chunk lanes of this block
  const int rl = 256 / cl;                          // row lanes
  const int t = threadIdx.x;
  const int my_c = t % cl, my_r = t / cl;
  const int chunk = blockIdx.y * 64 + my_c;
  const int c = chunk * V;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_slab;
  int64_t r1 = r0 + rows_per_slab;
  if (r1 > rows) r1 = rows;
  float su[V], sv[V];
#pragma unroll
  for (int e = 0; e < V; ++e) su[e] = sv[e] = 0.f;
  if (my_r < rl && chunk < nch) {
    float mu[V], rs[V];
#pragma unroll
    for (int e = 0; e < V; ++e) mu[e] = rs[e] = 0.f;
    if (MODE == 1) { ld_vec<V>(mean + c, mu); ld_vec<V>(rstd + c, rs); }
    for (int64_t r = r0 + my_r; r < r1; r += rl) {
      float zv[V];
      unpack_chunk(*reinterpret_cast<const uint4*>(z + r * cp + c), zv, T());
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < V; ++e) { su[e] += zv[e]; sv[e] = fmaf(zv[e], zv[e], sv[e]); }
      } else {
        float gv[V], yv[V];
        unpack_chunk(*reinterpret_cast<const uint4*>(dy + r * cp + c), gv, T());
        if (y != nullptr) unpack_chunk(*reinterpret_cast<const uint4*>(y + r * cp + c), yv, T());
        else {
#pragma unroll
          for (int e = 0; e < V; ++e) yv[e] = 1.f;
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float g = yv[e] > 0.f ? gv[e] : 0.f;
          su[e] += g;
          sv[e] = fmaf(g, (zv[e] - mu[e]) * rs[e], sv[e]);
        }
      }
    }
  }
  if (my_r < rl) {
#pragma unroll
    for (int e = 0; e < V; ++e) { red[0][(my_r * cl + my_c) * V + e] = su[e]; red[1][(my_r * cl + my_c) * V + e] = sv[e]; }
  }
  __syncthreads();
  // fixed order over the row lanes: thread -> one channel of this block's cl * V
  for (int ch = t; ch < cl * V; ch += 256) {
    const int gc = blockIdx.y * 64 * V + ch;
    if (gc >= cp) continue;
    float a = 0.f, b = 0.f;
    for (int r = 0; r < rl; ++r) { a += red[0][r * cl * V + ch]; b += red[1][r * cl * V + ch]; }
    float* p = part + (int64_t)blockIdx.x * 2 * cp + gc;
    p[0] = a;
    p[cp] = b;
  }
}

// Sum of the slab partials of ONE channel by one wave: lane l adds slabs l, l + 64, ... in double, then a fixed shuffle tree -- the order
// depends on nothing but the slab count (bit-reproducible).  Round 5: one THREAD per channel walking up to 1024 slabs serially made the two
// finalise kernels 23 % of the RN50 training step (210-220 us each at one to eight workgroups: profiles/r5_rn_train_kernel_stats.md).
__device__ __forceinline__ void slab_sums(const float* __restrict__ part, int slabs, int cp, int c, int lane, double& s1, double& s2) {
  double a = 0.0, b = 0.0;
  for (int s = lane; s < slabs; s += 64) {
    a += (double)part[(int64_t)s * 2 * cp + c];
    b += (double)part[(int64_t)s * 2 * cp + cp + c];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_xor(a, off, 64);
    b += __shfl_xor(b, off, 64);
  }
  s1 = a; s2 = b;
}

// forward statistics: one wave per channel sums the slabs (double), lane 0 writes mean / rstd / scale / shift and moves the
// running statistics.  Channels c >= C (padding): everything 0.
__global__ __launch_bounds__(256) void rn_bn_finalize_fwd_kernel(const float* __restrict__ part, int slabs, int cp, int C, int64_t n,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float* running_mean, float* running_var, float momentum, float eps,
                                                                  float* __restrict__ mean, float* __restrict__ rstd,
                                                                  float* __restrict__ scale, float* __restrict__ shift) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= cp) return;
  if (c >= C) { if (lane == 0) { mean[c] = 0.f; rstd[c] = 0.f; scale[c] = 0.f; shift[c] = 0.f; } return; }
  double s1, s2;
  slab_sums(part, slabs, cp, c, lane, s1, s2);
  if (lane != 0) return;
  const double m = s1 / (double)n;
  double var = s2 / (double)n - m * m;
  if (var < 0.0) var = 0.0;
  const float r = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = (float)m;
  rstd[c] = r;
  const float sc = gamma[c] * r;
  scale[c] = sc;
  shift[c] = beta[c] - (float)m * sc;
  if (running_mean != nullptr) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    const double unbiased = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// y = [relu](z * scale + shift [+ residual]); one thread per 16-byte chunk of one row (4 floats / 8 bf16: round 5 -- the 8-byte accesses of
// the first version ran at ~1.6 TB/s on the stem's 400 MB activations)
template <typename T>
__global__ __launch_bounds__(256) void rn_bn_apply_kernel(const T* __restrict__ z, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const T* __restrict__ residual, int relu,
                                                           int64_t chunks, int cp, T* __restrict__ y) {
  constexpr int V = Elem<T>::kPerChunk;
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= chunks) return;
  const int cq = cp / V;
  const int c = (int)(q % cq) * V;
  const int64_t off = (q / cq) * cp + c;
  float v[V], r[V], sc[V], sh[V];
  unpack_chunk(*reinterpret_cast<const uint4*>(z + off), v, T());
  if (residual != nullptr) unpack_chunk(*reinterpret_cast<const uint4*>(residual + off), r, T());
  else {
#pragma unroll
    for (int e = 0; e < V; ++e) r[e] = 0.f;
  }
  ld_vec<V>(scale + c, sc);
  ld_vec<V>(shift + c, sh);
#pragma unroll
  for (int e = 0; e < V; ++e) {
    float t = fmaf(v[e], sc[e], sh[e]) + r[e];
    if (relu) t = fmaxf(t, 0.f);
    v[e] = t;
  }
  *reinterpret_cast<uint4*>(y + off) = pack_chunk16(v, T());
}

// backward statistics: dgamma = sum g xhat, dbeta = sum g (written, or added when `accumulate`), and the two per-channel
// coefficients of the apply kernel: k1 = sum g / n, k2 = sum(g xhat) / n
__global__ __launch_bounds__(256) void rn_bn_finalize_bwd_kernel(const float* __restrict__ part, int slabs, int cp, int C, int64_t n,
                                                                  float* dgamma, float* dbeta, int accumulate,
                                                                  float* __restrict__ k1, float* __restrict__ k2) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);            // one wave per channel (slab_sums)
  if (c >= cp) return;
  if (c >= C) { if (lane == 0) { k1[c] = 0.f; k2[c] = 0.f; } return; }
  double s1, s2;
  slab_sums(part, slabs, cp, c, lane, s1, s2);
  if (lane != 0) return;
  if (accumulate) { dbeta[c] += (float)s1; dgamma[c] += (float)s2; }
  else { dbeta[c] = (float)s1; dgamma[c] = (float)s2; }
  k1[c] = (float)(s1 / (double)n);
  k2[c] = (float)(s2 / (double)n);
}

// dz = gamma * rstd * (g - k1 - xhat * k2), g = dy * [y > 0]; dres = g (optional).  Padded channels: gamma is not read (c >= C -> 0).
// One thread per 16-byte chunk of one row; the five per-channel vectors come in as float4 loads.
template <typename T>
__global__ __launch_bounds__(256) void rn_bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ z,
                                                               const float* __restrict__ gamma, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ k1,
                                                               const float* __restrict__ k2, int64_t chunks, int cp, int C,
                                                               T* __restrict__ dz, T* __restrict__ dres) {
  constexpr int V = Elem<T>::kPerChunk;
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= chunks) return;
  const int cq = cp / V;
  const int c = (int)(q % cq) * V;
  const int64_t off = (q / cq) * cp + c;
  float gv[V], zv[V], yv[V], o[V], g4[V], mu[V], rs[V], a1[V], a2[V], ga[V];
  unpack_chunk(*reinterpret_cast<const uint4*>(dy + off), gv, T());
  unpack_chunk(*reinterpret_cast<const uint4*>(z + off), zv, T());
  if (y != nullptr) unpack_chunk(*reinterpret_cast<const uint4*>(y + off), yv, T());
  else {
#pragma unroll
    for (int e = 0; e < V; ++e) yv[e] = 1.f;
  }
  ld_vec<V>(mean + c, mu);
  ld_vec<V>(rstd + c, rs);
  ld_vec<V>(k1 + c, a1);
  ld_vec<V>(k2 + c, a2);
  if (c + V <= C) ld_vec<V>(gamma + c, ga);          // (gamma has C entries: a chunk that straddles C reads it element by element)
  else {
#pragma unroll
    for (int e = 0; e < V; ++e) ga[e] = (c + e < C) ? gamma[c + e] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const float g = yv[e] > 0.f ? gv[e] : 0.f;
    g4[e] = g;
    const float a = (c + e < C) ? ga[e] * rs[e] : 0.f;
    const float xh = (zv[e] - mu[e]) * rs[e];
    o[e] = a * (g - a1[e] - xh * a2[e]);
  }
  *reinterpret_cast<uint4*>(dz + off) = pack_chunk16(o, T());
  if (dres != nullptr) *reinterpret_cast<uint4*>(dres + off) = pack_chunk16(g4, T());
}

// AvgPool2d(2) backward on NHWC: dx [B, H, W, cp] from dy [B, H/2, W/2, cp]; a thread owns 4 channels of one INPUT pixel
template <typename T>
__global__ __launch_bounds__(256) void rn_avgpool2_bwd_kernel(const T* __restrict__ dy, int64_t quads, int H, int W, int cp,
                                                               T* __restrict__ dx) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= quads) return;
  const int cq = cp >> 2;
  const int c = (int)(q % cq) * 4;
  const int64_t pix = q / cq;
  const int x = (int)(pix % W), yy = (int)((pix / W) % H);
  const int64_t b = pix / ((int64_t)W * H);
  const int Ho = H >> 1, Wo = W >> 1;
  float v[4];
  ld4(dy + ((b * Ho + (yy >> 1)) * Wo + (x >> 1)) * (int64_t)cp + c, v);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] *= 0.25f;
  st4(dx + pix * cp + c, v);
}

// a += b (gradient fan-in at the residual add), 4 elements per thread
template <typename T>
__global__ __launch_bounds__(256) void rn_add_kernel(T* __restrict__ a, const T* __restrict__ b, int64_t quads) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= quads) return;
  float x[4], y[4];
  ld4(a + q * 4, x);
  ld4(b + q * 4, y);
#pragma unroll
  for (int e = 0; e < 4; ++e) x[e] += y[e];
  st4(a + q * 4, x);
}

// explicit im2col of a 3x3, pad 1, stride 1 convolution on NHWC: col [B*H*W, 9 * cp]; a thread owns 4 channels of one (pixel, tap)
template <typename T>
__global__ __launch_bounds__(256) void rn_im2col3x3_kernel(const T* __restrict__ x, int64_t quads, int H, int W, int cp, T* __restrict__ col) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= quads) return;
  const int cq = cp >> 2;
  const int c = (int)(q % cq) * 4;
  const int tap = (int)((q / cq) % 9);
  const int64_t pix = q / ((int64_t)cq * 9);
  const int xx = (int)(pix % W), yy = (int)((pix / W) % H);
  const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) ld4(x + (pix + (int64_t)(sy - yy) * W + (sx - xx)) * cp + c, v);
  st4(col + pix * 9 * (int64_t)cp + (int64_t)tap * cp + c, v);
}

// ---- 3 x 3 weight gradient at 64 (padded) channels in and out: the stem's conv2 / conv3 and layer1's conv2 ------------------------------
// dWp[o][(ky * 3 + kx) * 64 + c] = sum over pixels m of dz[m][o] * x[m shifted by (ky - 1, kx - 1)][c]  (zero outside the image)
// These are the convolutions with the most pixels (256 images: 3.2 M at 112 x 112, 0.8 M at 56 x 56) and the smallest output (64 x 576):
// as a matrix product (gemm_tn) the 128 x 128 tiles re-read dz five times and x nine times through L2 and run at 2.4 TB/s of that; the
// problem itself reads x and dz ONCE.  Here a workgroup owns a contiguous range of strips (R image rows of one image), keeps the WHOLE
// 64 x 576 result in registers (wave w: channels 16 w .. 16 w + 15 of all nine taps and all 64 outputs = 36 16 x 16 accumulators) and per
// strip brings the rows into LDS once:
//   x image : rows y0 - 1 .. y0 + R as they lie in memory ([pixel][64 channels], 128 B per pixel), every image row preceded by two zero
//             pixels, zero rows where the image ends -- so a tap is a CONSTANT pixel offset (1 + dy) * (W + 2) + dx + 1 and needs no mask;
//   dz image: rows y0 .. y0 + R - 1 with the same row pitch, zero pixels at the pad positions (a pad pixel of dz times anything adds 0).
// The contraction runs over the padded pixel index in steps of 32 (v_mfma_f32_16x16x32_bf16); both operands want 8 consecutive pixels
// of one channel per lane = the LDS transpose read (ds_read_b64_tr_b16: 16 lanes read a 4 (pixel) x 16 (channel) block, lane c receives
// channel c) from the row-major images.  The 32-byte slot of a pixel row is XORed with (bit 1, bit 3) of the LDS row index: the eight
// rows one 32-lane pass touches (r .. r + 3, r + 8 .. r + 11, any r) then fall on eight different bank groups.
// Partials [workgroup][64][576] f32 are summed in a fixed order by rn_wgrad64_reduce_kernel: bit-reproducible.
typedef __attribute__((ext_vector_type(4))) short wg_s16x4_t;
typedef __attribute__((address_space(3))) wg_s16x4_t wg_lds_s16x4;

struct RnWgrad64Args {
  const bf16_t* x; const bf16_t* dz; float* part;
  int H, W, R, strips_per_image, strips, lz_r;
  int cp, opad;                // channels per pixel of x / dz in memory (multiples of 64): blockIdx.y = (64-output block) * (cp / 64) + (64-channel block)
};

__device__ __forceinline__ uint32_t wg64_slot(int row) { return (uint32_t)(((row >> 1) & 1) | (((row >> 3) & 1) << 1)); }
// byte offset of 16-byte chunk `ch` (0..7) of LDS pixel row `row`
__device__ __forceinline__ uint32_t wg64_chunk(int row, int ch) {
  return (uint32_t)row * 128u + ((((uint32_t)ch >> 1) ^ wg64_slot(row)) << 5) + (((uint32_t)ch & 1u) << 4);
}
// 8 consecutive pixel rows (row .. row + 3, row + 4 .. row + 7 by the lane's quarter) of one channel: lane (t = lane & 15, q4 = lane >> 4)
// passes row = first + 8 q4 + (t >> 2); cb = the 16-channel block
__device__ __forceinline__ uint4 wg64_frag(const char* img, int row, int cb, int t) {
  const uint32_t a0 = (uint32_t)row * 128u + (((uint32_t)cb ^ wg64_slot(row)) << 5) + (uint32_t)(t & 3) * 8u;
  const uint32_t a1 = (uint32_t)(row + 4) * 128u + (((uint32_t)cb ^ wg64_slot(row + 4)) << 5) + (uint32_t)(t & 3) * 8u;
  const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s16x4*)(img + a0)));
  const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s16x4*)(img + a1)));
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

__global__ __launch_bounds__(256, 2) void rn_wgrad3x3_c64_kernel(RnWgrad64Args a) {
  extern __shared__ __attribute__((aligned(16))) char wg_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15, q4 = lane >> 4;
  const int Wp = a.W + 2;
  const int nx = a.lz_r + 2 * Wp + 2;                 // pixel rows of the x image (LDS row 0 = the slack before the first padded pixel)
  char* xi = wg_smem;
  char* zi = wg_smem + (size_t)nx * 128;
  for (int q = tid; q < (nx + a.lz_r) * 8; q += 256) *reinterpret_cast<uint4*>(wg_smem + (size_t)q * 16) = make_uint4(0u, 0u, 0u, 0u);

  f32x4_t acc[9][4];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) acc[tap][ob] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int cblocks = a.cp >> 6;
  const int o_off = ((int)blockIdx.y / cblocks) * 64, c_off = ((int)blockIdx.y % cblocks) * 64;   // this workgroup's 64 outputs x 64 channels
  const int s_begin = (int)((int64_t)a.strips * blockIdx.x / gridDim.x), s_end = (int)((int64_t)a.strips * (blockIdx.x + 1) / gridDim.x);
  const int row_chunks = a.W * 8;                     // 16-byte chunks of one image row
  const int nxc = (a.R + 2) * row_chunks, nzc = a.R * row_chunks;
  for (int s = s_begin; s < s_end; ++s) {
    const int b = s / a.strips_per_image, y0 = (s - b * a.strips_per_image) * a.R;
    const bf16_t* xb = a.x + (int64_t)b * a.H * a.W * a.cp + c_off;
    const bf16_t* zb = a.dz + (int64_t)b * a.H * a.W * a.opad + o_off;
    __syncthreads();                                  // the previous strip's fragment reads (first strip: the zero fill) are done
    // eight chunks per thread in flight, then their LDS writes; the x rows first, the dz rows after them in one chunk numbering
    for (int q0 = tid; q0 < nxc + nzc; q0 += 256 * 8) {
      uint4 v[8];
      uint32_t dst[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + u * 256;
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        dst[u] = 0xffffffffu;
        if (q < nxc) {
          const int ry = q / row_chunks, rem = q - ry * row_chunks;
          const int xx = rem >> 3, ch = rem & 7, y = y0 - 1 + ry;
          if ((unsigned)y < (unsigned)a.H) v[u] = *reinterpret_cast<const uint4*>(xb + ((int64_t)y * a.W + xx) * a.cp + ch * 8);
          dst[u] = wg64_chunk(ry * Wp + xx + 2, ch);
        } else if (q < nxc + nzc) {
          const int qz = q - nxc;
          const int rz = qz / row_chunks, rem = qz - rz * row_chunks;
          const int xx = rem >> 3, ch = rem & 7, y = y0 + rz;
          if (y < a.H) v[u] = *reinterpret_cast<const uint4*>(zb + ((int64_t)y * a.W + xx) * a.opad + ch * 8);
          dst[u] = (uint32_t)nx * 128u + wg64_chunk(rz * Wp + xx + 1, ch);
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (dst[u] != 0xffffffffu) *reinterpret_cast<uint4*>(wg_smem + dst[u]) = v[u];
    }
    __syncthreads();
    for (int k = 0; k < a.lz_r; k += 32) {
      const int rz = k + 8 * q4 + (t >> 2);
      uint4 af[4];
#pragma unroll
      for (int ob = 0; ob < 4; ++ob) af[ob] = wg64_frag(zi, rz, ob, t);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int off = (tap / 3) * Wp + (tap % 3);   // (1 + dy) (W + 2) + dx + 1 with dy = tap / 3 - 1, dx = tap % 3 - 1
        const uint4 bf = wg64_frag(xi, rz + off, wave, t);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) mma16(acc[tap][ob], af[ob], bf);
      }
    }
  }
  // D column = lane & 15 = channel, rows 4 (lane >> 4) + r = output channel within the block
  float* pw = a.part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 * 576;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[(size_t)(ob * 16 + 4 * q4 + r) * 576 + tap * 64 + wave * 16 + t] = acc[tap][ob][r];
}

// out[o_off + o][tap * cp + c_off + c] (row stride ldo) = (accumulate ? out : 0) + sum_g part[y][g][o][tap * 64 + c]: block (i, y) owns 64
// float4s of sub-problem y; its 256 threads are 64 outputs x 4 lanes over g (lane l sums g = l, l + 4, ... ascending), the four lane sums
// are added in lane order -- a fixed order, bit-reproducible
__global__ __launch_bounds__(256) void rn_wgrad64_reduce_kernel(const float* __restrict__ part, int G, int cp, float* __restrict__ out, int64_t ldo,
                                                                 int accumulate) {
  __shared__ float4 red[4][64];
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), gl = threadIdx.x >> 6;      // float4 index into [64][576]; 64 * 576 / 4 = 144 * 64
  const int y = blockIdx.y;
  const float* p = part + (size_t)y * G * 64 * 576 + (size_t)i * 4;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int g = gl; g < G; g += 4) {
    const float4 v = *reinterpret_cast<const float4*>(p + (size_t)g * 64 * 576);
    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
  }
  red[gl][threadIdx.x & 63] = sum;
  __syncthreads();
  if (gl != 0) return;
#pragma unroll
  for (int l = 1; l < 4; ++l) {
    const float4 v = red[l][threadIdx.x];
    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
  }
  const int cblocks = cp >> 6;
  const int o = (i * 4) / 576, col = i * 4 - o * 576;
  const int tap = col >> 6, c = col & 63;
  float* d = out + (int64_t)((y / cblocks) * 64 + o) * ldo + tap * cp + (y % cblocks) * 64 + c;
  if (accumulate) { sum.x += d[0]; sum.y += d[1]; sum.z += d[2]; sum.w += d[3]; }
  d[0] = sum.x; d[1] = sum.y; d[2] = sum.z; d[3] = sum.w;
}

// ---- weight gradient of a 1 x 1 convolution with few channels on one side: C[N][K] (+)= A[M][N]^T . B[M][K], M huge, N x K small ------------
// layer1 / layer2 of the ModifiedResNet (64 ... 512 channels at 0.2 - 0.8 M pixels) and the stem's first convolution: as 128 x 128 tiles
// (gemm_tn) these run at 1 - 1.5 TB/s; the 8-phase kernel wants 256-multiples on both sides.  Same structure as the 3 x 3 kernel above
// without the taps: a workgroup owns 64 rows of C (blockIdx.y / k-blocks) x KB = 64 NSUB columns (blockIdx.y % k-blocks) and a contiguous
// range of P-pixel strips; per strip A's 64 columns go to one [pixel][64] LDS image and B's KB columns to NSUB such images (the 64-column
// groups side by side: wave w owns 16-column block w of every group), fragments by the LDS transpose read, the 64 x KB result in
// registers (4 x NSUB accumulators per wave), partials [blockIdx.y][workgroup][64][KB] summed in a fixed order.
struct RnTnSkinnyArgs {
  const bf16_t* A; const bf16_t* B; float* part;
  int64_t lda, ldb;
  int M, P, strips, kblocks;
};

template <int NSUB>
__global__ __launch_bounds__(256, 2) void rn_tn_skinny_kernel(RnTnSkinnyArgs a) {
  extern __shared__ __attribute__((aligned(16))) char wg_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15, q4 = lane >> 4;
  constexpr int KB = 64 * NSUB;
  const int n_off = ((int)blockIdx.y / a.kblocks) * 64, k_off = ((int)blockIdx.y % a.kblocks) * KB;
  const uint32_t img = (uint32_t)a.P * 128u;          // bytes of one [P][64] image; image 0 = A, 1 + i = B's column group i
  f32x4_t acc[NSUB][4];
#pragma unroll
  for (int i = 0; i < NSUB; ++i)
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) acc[i][ob] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int s_begin = (int)((int64_t)a.strips * blockIdx.x / gridDim.x), s_end = (int)((int64_t)a.strips * (blockIdx.x + 1) / gridDim.x);
  const int nac = a.P * 8, nbc = a.P * 8 * NSUB;      // 16-byte chunks of the A image / of the B images per strip
  for (int s = s_begin; s < s_end; ++s) {
    const int64_t m0 = (int64_t)s * a.P;
    __syncthreads();                                  // the previous strip's fragment reads are done
    // kFly chunks per thread in flight, then their LDS writes: two to three rounds per strip.  The counters read 70-76 % of the wave cycles
    // waiting (profiles/r5_rn_wgrad_pmc.md); 16 in flight (one round) spills -- 16 data + 16 address registers more than the 256 allow.
    constexpr int kFly = 8;
    for (int q0 = tid; q0 < nac + nbc; q0 += 256 * kFly) {
      uint4 v[kFly];
      uint32_t dst[kFly];
#pragma unroll
      for (int u = 0; u < kFly; ++u) {
        const int q = q0 + u * 256;
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        dst[u] = 0xffffffffu;
        if (q < nac) {
          const int px = q >> 3, ch = q & 7;
          if (m0 + px < a.M) v[u] = *reinterpret_cast<const uint4*>(a.A + (m0 + px) * a.lda + n_off + ch * 8);
          dst[u] = wg64_chunk(px, ch);
        } else if (q < nac + nbc) {
          const int qb = q - nac;
          const int px = qb / (8 * NSUB), ch = qb % (8 * NSUB);
          if (m0 + px < a.M) v[u] = *reinterpret_cast<const uint4*>(a.B + (m0 + px) * a.ldb + k_off + ch * 8);
          dst[u] = img * (uint32_t)(1 + (ch >> 3)) + wg64_chunk(px, ch & 7);
        }
      }
#pragma unroll
      for (int u = 0; u < kFly; ++u)
        if (dst[u] != 0xffffffffu) *reinterpret_cast<uint4*>(wg_smem + dst[u]) = v[u];
    }
    __syncthreads();
    for (int k = 0; k < a.P; k += 32) {
      const int row = k + 8 * q4 + (t >> 2);
      uint4 af[4];
#pragma unroll
      for (int ob = 0; ob < 4; ++ob) af[ob] = wg64_frag(wg_smem, row, ob, t);
#pragma unroll
      for (int i = 0; i < NSUB; ++i) {
        const uint4 bf = wg64_frag(wg_smem + img * (uint32_t)(1 + i), row, wave, t);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) mma16(acc[i][ob], af[ob], bf);
      }
    }
  }
  float* pw = a.part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 * KB;
#pragma unroll
  for (int i = 0; i < NSUB; ++i)
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[(size_t)(ob * 16 + 4 * q4 + r) * KB + i * 64 + wave * 16 + t] = acc[i][ob][r];
}

// C[n_off + o][k_off + c] (row stride ldc) = (accumulate ? C : 0) + sum_g part[y][g][o][c]  (64 x kb per sub-problem y): a block owns 16
// float4s; its 256 threads are 16 outputs x 16 lanes over g (lane l sums g = l, l + 16, ... ascending), the sixteen lane sums are added in
// lane order -- a fixed order, bit-reproducible.  (Four lanes, the first version, took 34 us for a 64 x 64 result: 128 dependent loads.)
__global__ __launch_bounds__(256) void rn_tn_skinny_reduce_kernel(const float* __restrict__ part, int G, int kb, int kblocks, float* __restrict__ C,
                                                                   int64_t ldc, int accumulate) {
  __shared__ float4 red[16][16];
  const int j = threadIdx.x & 15, gl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + j;                  // float4 index into [64][kb]; 16 kb of them: a multiple of 16
  const int y = blockIdx.y;
  const float* p = part + (size_t)y * G * 64 * kb + (size_t)i * 4;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int g = gl; g < G; g += 16) {
    const float4 v = *reinterpret_cast<const float4*>(p + (size_t)g * 64 * kb);
    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
  }
  red[gl][j] = sum;
  __syncthreads();
  if (gl != 0) return;
#pragma unroll
  for (int l = 1; l < 16; ++l) {
    const float4 v = red[l][j];
    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
  }
  const int o = (i * 4) / kb, c = i * 4 - o * kb;
  float* d = C + (int64_t)((y / kblocks) * 64 + o) * ldc + (y % kblocks) * kb + c;
  if (accumulate) { sum.x += d[0]; sum.y += d[1]; sum.z += d[2]; sum.w += d[3]; }
  d[0] = sum.x; d[1] = sum.y; d[2] = sum.z; d[3] = sum.w;
}

// weights of the input-gradient product: dst [ipad][k*k*opad], dst[c][(ky*k + kx)*opad + o] = W[o][c][k-1-ky][k-1-kx]; zeros elsewhere
template <typename T>
__global__ __launch_bounds__(256) void rn_pack_conv_dgrad_kernel(const float* __restrict__ W, int O, int I, int k, int opad, int ipad,
                                                                  T* __restrict__ dst) {
  const int c = blockIdx.x;                   // row of dst: input channel
  const int ld = k * k * opad;
  for (int j = threadIdx.x; j < ld; j += blockDim.x) {
    const int tap = j / opad, o = j - tap * opad;
    float v = 0.f;
    if (c < I && o < O) {
      const int ky = tap / k, kx = tap - ky * k;
      v = W[(((int64_t)o * I + c) * k + (k - 1 - ky)) * k + (k - 1 - kx)];
    }
    Elem<T>::st(dst + (int64_t)c * ld + j, v);
  }
  (void)ipad;
}

// dW[o][c][ky][kx] (+)= dWp[o][(ky*k + kx)*cp + c]
__global__ __launch_bounds__(256) void rn_unpack_wgrad_kernel(const float* __restrict__ dwp, int64_t ldp, int O, int I, int k, int cp,
                                                               int accumulate, float* __restrict__ dw) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int taps = k * k;
  if (idx >= (int64_t)O * I * taps) return;
  const int t = (int)(idx % taps);
  const int c = (int)((idx / taps) % I);
  const int64_t o = idx / ((int64_t)taps * I);
  const float v = dwp[o * ldp + (int64_t)t * cp + c];
  if (accumulate) dw[idx] += v; else dw[idx] = v;
}

template <typename K, typename... A>
int launch_quads(K kernel, int64_t quads, hipStream_t st, A... a) {
  if (quads <= 0) return EZ_OK;
  hipLaunchKernelGGL(kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, a...);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int moment_slabs(int64_t rows, int* rows_per_slab) {
  int64_t slabs = (rows + kMomentSlabRows - 1) / kMomentSlabRows;
  if (slabs > kMaxSlabs) slabs = kMaxSlabs;
  const int64_t per = (rows + slabs - 1) / slabs;
  *rows_per_slab = (int)per;
  return (int)((rows + per - 1) / per);
}

}  // namespace

#define RN_DISPATCH_T(dtype, ...)                                                    \
  do {                                                                               \
    if ((dtype) == EZCLIP_F32) { using T = float; __VA_ARGS__; }                     \
    else if ((dtype) == EZCLIP_BF16) { using T = bf16_t; __VA_ARGS__; }              \
    else { set_error("bad dtype %d", (int)(dtype)); return EZ_ERR_INVALID; }         \
  } while (0)

// scratch: float [slabs][2][cp] partial moments + 4 * cp per-channel vectors (mean / rstd are outputs of their own)
size_t rn_bn_scratch_bytes(int64_t rows, int cp) {
  int per;
  const int slabs = moment_slabs(rows, &per);
  return ((size_t)slabs * 2 * cp + (size_t)2 * cp) * sizeof(float);
}

int rn_bn_train_fwd(const void* z, int64_t rows, int C, int cp, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, const void* residual, int relu, void* y, float* mean, float* rstd,
                    float* scratch, int dtype, hipStream_t st) {
  EZ_REQUIRE(z && y && gamma && beta && mean && rstd && scratch && rows > 0 && C > 0 && C <= cp && cp % 64 == 0,
             "rn_bn_train_fwd: bad arguments (rows %lld, C %d, cp %d)", (long long)rows, C, cp);
  int per;
  const int slabs = moment_slabs(rows, &per);
  float* part = scratch;
  float* scale = scratch + (size_t)slabs * 2 * cp;
  float* shift = scale + cp;
  RN_DISPATCH_T(dtype, hipLaunchKernelGGL((rn_moments_kernel<T, 0>), dim3(slabs, (cp / Elem<T>::kPerChunk + 63) / 64), dim3(256), 0, st, (const T*)z,
                                           (const T*)nullptr, (const T*)nullptr, (const float*)nullptr, (const float*)nullptr, rows, cp, per,
                                           part));
  EZ_LAUNCH_CHECK();
  hipLaunchKernelGGL(rn_bn_finalize_fwd_kernel, dim3((cp + 3) / 4), dim3(256), 0, st, part, slabs, cp, C, rows, gamma, beta,
                     running_mean, running_var, momentum, eps, mean, rstd, scale, shift);
  EZ_LAUNCH_CHECK();
  RN_DISPATCH_T(dtype, { const int64_t chunks = rows * (cp / Elem<T>::kPerChunk);
                         return launch_quads(rn_bn_apply_kernel<T>, chunks, st, (const T*)z, (const float*)scale, (const float*)shift,
                                             (const T*)residual, relu, chunks, cp, (T*)y); });
  return EZ_OK;
}

int rn_bn_train_bwd(const void* dy, const void* y, const void* z, int64_t rows, int C, int cp, const float* gamma, const float* mean,
                    const float* rstd, void* dz, void* dres, float* dgamma, float* dbeta, int accumulate, float* scratch, int dtype,
                    hipStream_t st) {
  EZ_REQUIRE(dy && z && dz && gamma && mean && rstd && dgamma && dbeta && scratch && rows > 0 && C > 0 && C <= cp && cp % 64 == 0,
             "rn_bn_train_bwd: bad arguments (rows %lld, C %d, cp %d)", (long long)rows, C, cp);
  int per;
  const int slabs = moment_slabs(rows, &per);
  float* part = scratch;
  float* k1 = scratch + (size_t)slabs * 2 * cp;
  float* k2 = k1 + cp;
  RN_DISPATCH_T(dtype, hipLaunchKernelGGL((rn_moments_kernel<T, 1>), dim3(slabs, (cp / Elem<T>::kPerChunk + 63) / 64), dim3(256), 0, st, (const T*)z,
                                           (const T*)dy, (const T*)y, mean, rstd, rows, cp, per, part));
  EZ_LAUNCH_CHECK();
  hipLaunchKernelGGL(rn_bn_finalize_bwd_kernel, dim3((cp + 3) / 4), dim3(256), 0, st, part, slabs, cp, C, rows, dgamma, dbeta,
                     accumulate, k1, k2);
  EZ_LAUNCH_CHECK();
  RN_DISPATCH_T(dtype, { const int64_t chunks = rows * (cp / Elem<T>::kPerChunk);
                         return launch_quads(rn_bn_bwd_apply_kernel<T>, chunks, st, (const T*)dy, (const T*)y, (const T*)z, gamma, mean, rstd,
                                             (const float*)k1, (const float*)k2, chunks, cp, C, (T*)dz, (T*)dres); });
  return EZ_OK;
}

int rn_avgpool2_bwd(const void* dy, int B, int H, int W, int cp, void* dx, int dtype, hipStream_t st) {
  EZ_REQUIRE(dy && dx && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && cp % 4 == 0, "rn_avgpool2_bwd: bad shape %d x %d x %d x %d", B, H,
             W, cp);
  const int64_t quads = (int64_t)B * H * W * (cp / 4);
  RN_DISPATCH_T(dtype, return launch_quads(rn_avgpool2_bwd_kernel<T>, quads, st, (const T*)dy, quads, H, W, cp, (T*)dx));
  return EZ_OK;
}

int rn_add_inplace(void* a, const void* b, int64_t n, int dtype, hipStream_t st) {
  EZ_REQUIRE(a && b && n >= 0 && n % 4 == 0, "rn_add_inplace: n = %lld must be a multiple of 4", (long long)n);
  RN_DISPATCH_T(dtype, return launch_quads(rn_add_kernel<T>, n / 4, st, (T*)a, (const T*)b, n / 4));
  return EZ_OK;
}

int rn_im2col3x3(const void* x, int B, int H, int W, int cp, void* col, int dtype, hipStream_t st) {
  EZ_REQUIRE(x && col && B > 0 && H > 0 && W > 0 && cp % 4 == 0, "rn_im2col3x3: bad shape %d x %d x %d x %d", B, H, W, cp);
  const int64_t quads = (int64_t)B * H * W * 9 * (cp / 4);
  RN_DISPATCH_T(dtype, return launch_quads(rn_im2col3x3_kernel<T>, quads, st, (const T*)x, quads, H, W, cp, (T*)col));
  return EZ_OK;
}

// the strip height of rn_wgrad3x3_c64: R image rows whose two LDS images fit twice into a CU, chosen by the work per image -- strips x
// (padded pixels of the contraction, rounded up to 32, + half a unit per pixel loaded: the halo rows are loaded again by the next strip)
static int wgrad64_strip_rows(int H, int W, int* lz_r, size_t* lds) {
  const int Wp = W + 2;
  int best = 0;
  double best_cost = 0.0;
  for (int R = 1; R <= H && R <= 16; ++R) {
    const int lr = (R * Wp + 31) / 32 * 32;
    const size_t bytes = (size_t)(lr + 2 * Wp + 2 + lr) * 128;
    if (bytes > 78 * 1024) break;
    const double cost = (double)((H + R - 1) / R) * (lr + 0.5 * (R + 2) * Wp);
    if (best == 0 || cost <= best_cost) { best = R; best_cost = cost; *lz_r = lr; *lds = bytes; }
  }
  return best;
}

bool rn_wgrad3x3_c64_eligible(int B, int H, int W, int cp, int opad, int dtype, size_t scratch_bytes) {
  int lr = 0; size_t lds = 0;
  return dtype == EZCLIP_BF16 && cp >= 64 && opad >= 64 && cp % 64 == 0 && opad % 64 == 0 && (cp / 64) * (opad / 64) <= 4 && B > 0 && H > 0 &&
         W >= 2 && wgrad64_strip_rows(H, W, &lr, &lds) > 0 && scratch_bytes >= (size_t)(cp / 64) * (opad / 64) * 64 * 576 * 4;
}

int rn_wgrad3x3_c64(const void* x, const void* dz, int B, int H, int W, int cp, int opad, void* scratch, size_t scratch_bytes, float* out,
                    int64_t ldo, int accumulate, hipStream_t st) {
  EZ_REQUIRE(x && dz && scratch && out && ldo >= 9 * (int64_t)cp && ldo % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dz % 16) == 0 &&
                 ((uintptr_t)out % 16) == 0 && ((uintptr_t)scratch % 16) == 0 && rn_wgrad3x3_c64_eligible(B, H, W, cp, opad, EZCLIP_BF16, scratch_bytes),
             "rn_wgrad3x3_c64: %d images of %d x %d, %d -> %d padded channels (bf16; 64 or 128 each), scratch %zu bytes, ldo %lld (a multiple of 4; 16-byte aligned buffers): not a shape of this kernel",
             B, H, W, cp, opad, scratch_bytes, (long long)ldo);
  RnWgrad64Args a;
  size_t lds = 0;
  a.x = (const bf16_t*)x; a.dz = (const bf16_t*)dz; a.part = (float*)scratch;
  a.H = H; a.W = W; a.cp = cp; a.opad = opad;
  a.R = wgrad64_strip_rows(H, W, &a.lz_r, &lds);
  a.strips_per_image = (H + a.R - 1) / a.R;
  a.strips = B * a.strips_per_image;
  const int sub = (cp / 64) * (opad / 64);            // independent 64-output x 64-channel sub-problems (blockIdx.y)
  int G = 512 / sub;                                  // two workgroups per CU in all
  if (G > a.strips) G = a.strips;
  const size_t fit = scratch_bytes / ((size_t)sub * 64 * 576 * 4);
  if ((size_t)G > fit) G = (int)fit;
  static LdsOptIn lds_opt;
  EZ_ENSURE_LDS(rn_wgrad3x3_c64_kernel, lds_opt, lds);
  hipLaunchKernelGGL(rn_wgrad3x3_c64_kernel, dim3(G, sub), dim3(256), lds, st, a);
  EZ_LAUNCH_CHECK();
  hipLaunchKernelGGL(rn_wgrad64_reduce_kernel, dim3(64 * 576 / 4 / 64, sub), dim3(256), 0, st, (const float*)scratch, G, cp, out, ldo, accumulate);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

// column block of rn_tn_skinny for K columns: the widest of 256 / 128 / 64 that divides K (512 = 32 accumulators per wave spills)
static int tn_skinny_kb(int K) { return K % 256 == 0 ? 256 : K % 128 == 0 ? 128 : 64; }

bool rn_tn_skinny_eligible(int64_t M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int dtype, size_t scratch_bytes) {
  if (dtype != EZCLIP_BF16 || M < 4096 || M > 0x7fffffff || N < 64 || K < 64 || N % 64 || K % 64 || lda % 8 || ldb % 8 || ldc % 4) return false;
  const int kb = tn_skinny_kb(K);
  const int sub = (N / 64) * (K / kb);
  return sub <= 16 && scratch_bytes >= (size_t)sub * 64 * kb * 4;
}

int rn_tn_skinny(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K, int accumulate,
                 void* scratch, size_t scratch_bytes, hipStream_t st) {
  EZ_REQUIRE(A && B && C && scratch && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0 && ((uintptr_t)scratch % 16) == 0 &&
                 rn_tn_skinny_eligible(M, N, K, lda, ldb, ldc, EZCLIP_BF16, scratch_bytes),
             "rn_tn_skinny: M %lld N %d K %d (lda %lld ldb %lld ldc %lld), scratch %zu bytes: not a shape of this kernel (bf16, M >= 4096, N and K multiples "
             "of 64, at most 16 blocks of 64 x 256)", (long long)M, N, K, (long long)lda, (long long)ldb, (long long)ldc, scratch_bytes);
  const int kb = tn_skinny_kb(K), nsub = kb / 64;
  RnTnSkinnyArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.part = (float*)scratch; a.lda = lda; a.ldb = ldb; a.M = (int)M;
  a.kblocks = K / kb;
  a.P = (int)(78 * 1024 / (128 * (1 + nsub))) / 32 * 32;             // pixels per strip: the 1 + NSUB images twice into a CU's LDS
  if (a.P > 256) a.P = 256;
  static const int p_cap = getenv("EZCLIP_RN_SKINNY_P") ? atoi(getenv("EZCLIP_RN_SKINNY_P")) / 32 * 32 : 0;      // A-B switch: shorter strips (no effect at 128 / 64, -2 % at 32: profiles/r5_rn_skinny_strip_length.log)
  if (p_cap >= 32 && a.P > p_cap) a.P = p_cap;
  a.strips = (int)((M + a.P - 1) / a.P);
  const int sub = (N / 64) * a.kblocks;
  int G = 512 / sub;
  if (G > a.strips) G = a.strips;
  const size_t fit = scratch_bytes / ((size_t)sub * 64 * kb * 4);
  if ((size_t)G > fit) G = (int)fit;
  const size_t lds = (size_t)a.P * 128 * (1 + nsub);
  static LdsOptIn opt1, opt2, opt4;
  switch (nsub) {
    case 1: EZ_ENSURE_LDS(rn_tn_skinny_kernel<1>, opt1, lds); hipLaunchKernelGGL(rn_tn_skinny_kernel<1>, dim3(G, sub), dim3(256), lds, st, a); break;
    case 2: EZ_ENSURE_LDS(rn_tn_skinny_kernel<2>, opt2, lds); hipLaunchKernelGGL(rn_tn_skinny_kernel<2>, dim3(G, sub), dim3(256), lds, st, a); break;
    default: EZ_ENSURE_LDS(rn_tn_skinny_kernel<4>, opt4, lds); hipLaunchKernelGGL(rn_tn_skinny_kernel<4>, dim3(G, sub), dim3(256), lds, st, a); break;
  }
  EZ_LAUNCH_CHECK();
  hipLaunchKernelGGL(rn_tn_skinny_reduce_kernel, dim3(16 * kb / 16, sub), dim3(256), 0, st, (const float*)scratch, G, kb, a.kblocks, C, ldc, accumulate);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int rn_pack_conv_dgrad(const float* W, int O, int I, int k, int opad, int ipad, void* dst, int dtype, hipStream_t st) {
  EZ_REQUIRE(W && dst && O > 0 && I > 0 && (k == 1 || k == 3) && opad >= O && ipad >= I, "rn_pack_conv_dgrad: bad shape O %d I %d k %d", O, I, k);
  RN_DISPATCH_T(dtype, hipLaunchKernelGGL((rn_pack_conv_dgrad_kernel<T>), dim3(ipad), dim3(256), 0, st, W, O, I, k, opad, ipad, (T*)dst));
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int rn_unpack_wgrad(const float* dwp, int64_t ldp, int O, int I, int k, int cp, int accumulate, float* dw, hipStream_t st) {
  EZ_REQUIRE(dwp && dw && O > 0 && I > 0 && (k == 1 || k == 3) && cp >= I && ldp >= (int64_t)k * k * cp, "rn_unpack_wgrad: bad shape O %d I %d k %d cp %d",
             O, I, k, cp);
  const int64_t n = (int64_t)O * I * k * k;
  hipLaunchKernelGGL(rn_unpack_wgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dwp, ldp, O, I, k, cp, accumulate, dw);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace ezclip
