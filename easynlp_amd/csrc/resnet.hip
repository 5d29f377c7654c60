// ModifiedResNet image tower (CLIP RN50 family), inference / frozen tower.
//
// Reference: easynlp/modelzoo/models/clip/modeling_chineseclip.py -- Bottleneck :27-74, AttentionPool2d :77-108,
// ModifiedResNet :110-167, built by CHINESE_CLIP when `vision_layers` is a tuple (:279-287).  EVAL mode: BatchNorm with its
// running statistics (what evaluation and prediction run; as a frozen image tower also what a LiT-style fine-tune runs).
//
// Layout: activations NHWC, [B * H * W, Cp] row-major in the compute dtype with the channel count padded to a multiple of 64
// (pad channels are exact zeros: the packed weights carry zero rows / columns there).  Every convolution is an MFMA GEMM:
//   1x1 conv + BN (+ ReLU)         C = act(A . W'^T + shift),  W' = W o (gamma / sqrt(var + eps)) folded at refresh time
//   3x3 conv + BN + ReLU           the same with A read IMPLICITLY (GemmArgs::conv_*: the K-tile picks a tap and a channel slice of
//                                  the shifted pixel; no im2col buffer -- at 64 channels an explicit one would cost 8x the GEMM's time)
//   conv3 + BN + identity + ReLU   residual epilogue with the ReLU after the add (ACT_RELU_POST)
//   stem conv1 (3 channels, stride 2)   explicit im2col of 27 (-> 64) columns, as the ViT's patch embedding
// AvgPool2d(2) (the anti-aliased stride) is one row kernel; AttentionPool2d assembles [mean; positions] + positional embedding,
// projects keys / values for all tokens and the query for the mean token only, and runs the one-query attention kernel of the
// towers' CLS-only last blocks (head dim 64: heads = width * 32 / 64).  Output: L2-normalised [B, output_dim] float32, as
// ezclip_encode_image returns it (CHINESE_CLIP.forward :360).
#include <string>
#include <vector>

#include "../../include/ezclip.h"
#include "ezclip_common.h"
#include "kernels.h"

namespace ezclip {
namespace {

inline int rup(int v, int m) { return (v + m - 1) / m * m; }
constexpr float kBnEps = 1e-5f;

// ---- kernels ---------------------------------------------------------------------------------------------------------------
// Packed copy of a convolution weight with the following BatchNorm folded in:
//   dst[o][(ky * kw + kx) * Cp + c] = W[o][c][ky][kx] * scale[o]   (c < I, o < O; zero elsewhere: o < Opad, c < Cp, + ldk padding)
//   bias[o] = beta[o] - mean[o] * scale[o],  scale = gamma / sqrt(var + eps)          (bn == null: scale 1, bias = lin_bias or 0)
// order_cfirst: K index = c * kh * kw + ky * kw + kx (the stem's explicit im2col, inner index (c, ky, kx) like the ViT's).
template <typename T>
__global__ __launch_bounds__(256) void rn_pack_conv_kernel(const float* __restrict__ W, const float* gamma, const float* beta,
                                                           const float* mean, const float* var, const float* lin_bias, int O, int I,
                                                           int kh, int kw, int Opad, int Cp, int ldk, int order_cfirst,
                                                           T* __restrict__ dst, float* __restrict__ bias) {
  const int o = blockIdx.x;
  float scale = 1.f, shift = 0.f;
  if (o < O) {
    if (gamma) {
      scale = gamma[o] / sqrtf(var[o] + kBnEps);
      shift = beta[o] - mean[o] * scale;
    } else if (lin_bias) {
      shift = lin_bias[o];
    }
  }
  if (threadIdx.x == 0) bias[o] = o < O ? shift : 0.f;
  const int taps = kh * kw;
  for (int k = threadIdx.x; k < ldk; k += blockDim.x) {
    float v = 0.f;
    if (o < O) {
      int c, t;
      if (order_cfirst) { c = k / taps; t = k - c * taps; }
      else { t = k / Cp; c = k - t * Cp; }
      if (c < I && t < taps && (order_cfirst || k < taps * Cp)) v = W[((int64_t)o * I + c) * taps + t] * scale;
    }
    Elem<T>::st(dst + (int64_t)o * ldk + k, v);
  }
}

// stem conv1: pixels [B, 3, R, R] f32 -> col [B * Ho * Ho, ldk] with K index (c, ky, kx), 3x3, stride 2, pad 1
template <typename T>
__global__ __launch_bounds__(256) void rn_stem_im2col_kernel(const float* __restrict__ px, int B, int R, int Ho, int ldk, T* __restrict__ col) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (row, k)
  const int64_t rows = (int64_t)B * Ho * Ho;
  if (idx >= rows * ldk) return;
  const int k = (int)(idx % ldk);
  const int64_t row = idx / ldk;
  float v = 0.f;
  if (k < 27) {
    const int c = k / 9, ky = (k % 9) / 3, kx = k % 3;
    const int xo = (int)(row % Ho), yo = (int)((row / Ho) % Ho);
    const int64_t b = row / ((int64_t)Ho * Ho);
    const int y = yo * 2 + ky - 1, x = xo * 2 + kx - 1;
    if ((unsigned)y < (unsigned)R && (unsigned)x < (unsigned)R) v = px[((b * 3 + c) * R + y) * (int64_t)R + x];
  }
  Elem<T>::st(col + idx, v);
}

// AvgPool2d(2) on NHWC: in [B, H, W, Cp] -> out [B, H/2, W/2, Cp]; a thread owns 4 channels of one output pixel
template <typename T>
__global__ __launch_bounds__(256) void rn_avgpool2_kernel(const T* __restrict__ in, int64_t n_out_quads, int H, int W, int Cp, T* __restrict__ out) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= n_out_quads) return;
  const int cq = Cp >> 2, Ho = H >> 1, Wo = W >> 1;
  const int c = (int)(q % cq) * 4;
  const int64_t pix = q / cq;
  const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho);
  const int64_t b = pix / ((int64_t)Wo * Ho);
  const T* src = in + (((b * H + 2 * yo) * W + 2 * xo) * (int64_t)Cp + c);
  float a[4], t[4];
  ld4(src, a);
  ld4(src + Cp, t);
#pragma unroll
  for (int e = 0; e < 4; ++e) a[e] += t[e];
  ld4(src + (int64_t)W * Cp, t);
#pragma unroll
  for (int e = 0; e < 4; ++e) a[e] += t[e];
  ld4(src + (int64_t)W * Cp + Cp, t);
#pragma unroll
  for (int e = 0; e < 4; ++e) a[e] = (a[e] + t[e]) * 0.25f;
  st4(out + pix * Cp + c, a);
}

// AttentionPool2d input: tok[b][0] = mean_p x[b][p] + pos[0]; tok[b][1 + p] = x[b][p] + pos[1 + p]   (x [B, P, C], pos f32 [P + 1, C])
template <typename T>
__global__ __launch_bounds__(256) void rn_attnpool_tokens_kernel(const T* __restrict__ x, const float* __restrict__ pos, int P, int C,
                                                                 T* __restrict__ tok) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float sum = 0.f;
    for (int p = 0; p < P; ++p) {
      const float v = Elem<T>::ld(x + ((int64_t)b * P + p) * C + c);
      sum += v;
      Elem<T>::st(tok + ((int64_t)b * (P + 1) + 1 + p) * C + c, v + pos[(int64_t)(1 + p) * C + c]);
    }
    Elem<T>::st(tok + (int64_t)b * (P + 1) * C + c, sum / (float)P + pos[c]);
  }
}

template <typename K, typename... A>
int launch1d(K kernel, int64_t threads, hipStream_t st, A... a) {
  if (threads <= 0) return EZ_OK;
  hipLaunchKernelGGL(kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, a...);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace
}  // namespace ezclip

// ---- host side ---------------------------------------------------------------------------------------------------------------
using namespace ezclip;

struct ezclip_rn {
  ezclip_rn_config cfg;
  int dtype = 0, esz = 2;
  struct Param { std::string name; std::vector<int64_t> shape; const float* w = nullptr; };
  std::vector<Param> params;
  struct Conv {             // one convolution (+ BatchNorm) or Linear, with its packed copy
    int w = -1, bn = -1, lin_b = -1;     // param indices: weight; bn.weight (bias, running_mean, running_var follow); Linear bias
    int O = 0, I = 0, k = 1;
    int Opad = 0, Cp = 0, ldk = 0;       // packed [Opad][ldk]; Cp = padded input channels (per tap)
    int cfirst = 0;
    void* s = nullptr;
    float* bias = nullptr;
  };
  Conv stem[3];
  struct Block { Conv c1, c2, c3, down; bool has_down = false; int stride = 1; };
  std::vector<Block> blocks[4];
  Conv kproj, vproj, qproj, cproj;
  int pos_p = -1;
  int heads = 0, embed = 0, sp = 0;      // attention pool: heads, channel count (width * 32), spatial side (resolution / 32)
  void* shadow = nullptr;
  size_t shadow_bytes = 0;
  bool fresh = false;
  void* zero = nullptr;                  // 256 zero bytes inside the shadow (padding taps of the implicit convolutions)
};

namespace {

int rn_add(ezclip_rn* m, const std::string& name, std::vector<int64_t> shape) {
  ezclip_rn::Param p;
  p.name = name; p.shape = shape;
  m->params.push_back(p);
  return (int)m->params.size() - 1;
}
int rn_add_bn(ezclip_rn* m, const std::string& name, int c) {
  const int first = rn_add(m, name + ".weight", {c});
  rn_add(m, name + ".bias", {c});
  rn_add(m, name + ".running_mean", {c});
  rn_add(m, name + ".running_var", {c});
  return first;
}
ezclip_rn::Conv rn_conv(ezclip_rn* m, const std::string& wname, const std::string& bnname, int O, int I, int k, bool cfirst = false) {
  ezclip_rn::Conv c;
  c.O = O; c.I = I; c.k = k; c.cfirst = cfirst ? 1 : 0;
  c.w = rn_add(m, wname, {O, I, k, k});
  c.bn = rn_add_bn(m, bnname, O);
  c.Opad = rup(O, 64);
  c.Cp = cfirst ? I : rup(I, 64);
  c.ldk = cfirst ? rup(I * k * k, 64) : k * k * c.Cp;
  return c;
}
ezclip_rn::Conv rn_linear(ezclip_rn* m, const std::string& name, int O, int I) {
  ezclip_rn::Conv c;
  c.O = O; c.I = I; c.k = 1;
  c.w = rn_add(m, name + ".weight", {O, I});
  c.lin_b = rn_add(m, name + ".bias", {O});
  c.Opad = O;                             // (I = width * 32 is a multiple of 64; O = output_dim may be ragged: the general kernel)
  c.Cp = rup(I, 64);
  c.ldk = c.Cp;
  return c;
}

template <typename F>
void rn_for_each(ezclip_rn* m, F&& f) {
  for (auto& c : m->stem) f(c);
  for (auto& L : m->blocks)
    for (auto& b : L) { f(b.c1); f(b.c2); f(b.c3); if (b.has_down) f(b.down); }
  f(m->kproj); f(m->vproj); f(m->qproj); f(m->cproj);
}

size_t rn_shadow_layout(ezclip_rn* m, char* base) {
  size_t off = 0;
  auto take = [&](size_t bytes) { void* p = base ? base + off : nullptr; off += (bytes + 255) / 256 * 256; return p; };
  void* z = take(256);
  if (base) m->zero = z;
  rn_for_each(m, [&](ezclip_rn::Conv& c) {
    void* s = take((size_t)c.Opad * c.ldk * m->esz);
    float* b = (float*)take((size_t)c.Opad * 4);
    if (base) { c.s = s; c.bias = b; }
  });
  return off + 256;
}

// activation buffers of one chunk of `bc` images: the largest tensor is [bc * (R/2)^2, 64] (stem) = [bc * (R/4)^2, 256 * width/64]
size_t rn_act_bytes(const ezclip_rn* m, int bc) {
  const int R = m->cfg.image_resolution, w = m->cfg.width;
  const size_t stem = (size_t)bc * (R / 2) * (R / 2) * rup(w, 64);
  const size_t l1 = (size_t)bc * (R / 4) * (R / 4) * rup(4 * w, 64);
  const size_t tok = (size_t)bc * (m->sp * m->sp + 1) * m->embed;
  size_t mx = stem > l1 ? stem : l1;
  if (tok > mx) mx = tok;
  return (mx * m->esz + 255) / 256 * 256;
}
constexpr int kRnBufs = 5;
size_t g_rn_buf_bound = (size_t)256 << 20;      // ezclip_debug_set(10, MiB): tests walk a small batch in several chunks
int rn_chunk(const ezclip_rn* m, int B) {
  // bound the workspace: at most ~256 MB per activation buffer
  const size_t per_image = rn_act_bytes(m, 1);
  int bc = (int)(g_rn_buf_bound / per_image);
  if (bc < 1) bc = 1;
  return bc < B ? bc : B;
}

int rn_gemm(const ezclip_rn* m, const void* A, int64_t lda, int M, const ezclip_rn::Conv& c, void* C, int64_t ldc, int act, const void* R,
            int64_t ldr, int convH, int convW, bool out_f32, hipStream_t st) {
  GemmArgs g;
  g.A = A; g.lda = lda; g.B = c.s; g.ldb = c.ldk; g.C = C; g.ldc = ldc;
  g.bias = c.bias; g.R = R; g.ldr = ldr;
  g.M = M; g.N = c.Opad; g.K = c.ldk; g.act = act;
  g.out_f32 = (out_f32 && m->dtype == EZCLIP_BF16) ? 1 : 0;
  if (convH > 0) { g.conv_H = convH; g.conv_W = convW; g.conv_C = c.Cp; g.conv_zero = m->zero; }
  return gemm_nt(g, m->dtype, st);
}

#define RN_TRY(expr)                 \
  do {                               \
    int _rc = (expr);                \
    if (_rc != EZ_OK) return _rc;    \
  } while (0)

template <typename T>
int rn_forward_chunk(ezclip_rn* m, const float* px, int bc, float* out, char* ws, hipStream_t st) {
  const int R = m->cfg.image_resolution, w = m->cfg.width;
  const size_t ab = rn_act_bytes(m, bc);
  T* buf[kRnBufs];
  for (int i = 0; i < kRnBufs; ++i) buf[i] = reinterpret_cast<T*>(ws + i * ab);
  // ---- stem: conv1 (stride 2) through an explicit im2col, conv2 / conv3 implicit, AvgPool2d(2)
  int H = R / 2;
  int64_t M = (int64_t)bc * H * H;
  RN_TRY(launch1d(rn_stem_im2col_kernel<T>, M * m->stem[0].ldk, st, px, bc, R, H, m->stem[0].ldk, buf[0]));
  RN_TRY(rn_gemm(m, buf[0], m->stem[0].ldk, (int)M, m->stem[0], buf[1], m->stem[0].Opad, ACT_RELU, nullptr, 0, 0, 0, false, st));
  RN_TRY(rn_gemm(m, buf[1], m->stem[0].Opad, (int)M, m->stem[1], buf[2], m->stem[1].Opad, ACT_RELU, nullptr, 0, H, H, false, st));
  RN_TRY(rn_gemm(m, buf[2], m->stem[1].Opad, (int)M, m->stem[2], buf[1], m->stem[2].Opad, ACT_RELU, nullptr, 0, H, H, false, st));
  int C = m->stem[2].Opad;
  RN_TRY(launch1d(rn_avgpool2_kernel<T>, M / 4 * (C / 4), st, (const T*)buf[1], M / 4 * (C / 4), H, H, C, buf[0]));
  H /= 2;
  M /= 4;
  T* x = buf[0];                       // current activation [M, C]
  int xi = 0;
  (void)w;
  for (auto& L : m->blocks) {
    for (auto& b : L) {
      // free buffers: every index but xi
      int f[4], nf = 0;
      for (int i = 0; i < kRnBufs; ++i) if (i != xi) f[nf++] = i;
      T *t1 = buf[f[0]], *t2 = buf[f[1]], *t3 = buf[f[2]], *t4 = buf[f[3]];
      RN_TRY(rn_gemm(m, x, C, (int)M, b.c1, t1, b.c1.Opad, ACT_RELU, nullptr, 0, 0, 0, false, st));
      RN_TRY(rn_gemm(m, t1, b.c1.Opad, (int)M, b.c2, t2, b.c2.Opad, ACT_RELU, nullptr, 0, H, H, false, st));
      const T* o2 = t2;
      const T* idn = x;
      int64_t Mo = M;
      int Ho = H;
      if (b.stride > 1) {
        const int Cm = b.c2.Opad;
        RN_TRY(launch1d(rn_avgpool2_kernel<T>, M / 4 * (Cm / 4), st, (const T*)t2, M / 4 * (Cm / 4), H, H, Cm, t1));
        o2 = t1;
        RN_TRY(launch1d(rn_avgpool2_kernel<T>, M / 4 * (C / 4), st, (const T*)x, M / 4 * (C / 4), H, H, C, t3));
        idn = t3;
        Mo = M / 4;
        Ho = H / 2;
      }
      const T* identity = idn;
      int64_t ld_id = C;
      if (b.has_down) {
        RN_TRY(rn_gemm(m, idn, C, (int)Mo, b.down, t4, b.down.Opad, ACT_NONE, nullptr, 0, 0, 0, false, st));
        identity = t4;
        ld_id = b.down.Opad;
      }
      // relu(bn3(conv3(o2)) + identity) -> the buffer neither o2 nor identity nor x lives in: t2 when pooled (o2 = t1), else t3
      T* dst = (b.stride > 1) ? t2 : t3;
      RN_TRY(rn_gemm(m, o2, b.c2.Opad, (int)Mo, b.c3, dst, b.c3.Opad, ACT_RELU_POST, identity, ld_id, 0, 0, false, st));
      x = dst;
      for (int i = 0; i < kRnBufs; ++i) if (buf[i] == dst) xi = i;
      C = b.c3.Opad;
      M = Mo;
      H = Ho;
    }
  }
  // ---- AttentionPool2d
  EZ_REQUIRE(H == m->sp && C == m->embed, "rn_forward: tower ends at %d x %d x %d, attention pool expects %d x %d x %d", H, H, C, m->sp,
             m->sp, m->embed);
  const int P = H * H, Lt = P + 1;
  int f[4], nf = 0;
  for (int i = 0; i < kRnBufs; ++i) if (i != xi) f[nf++] = i;
  T *tok = buf[f[0]], *kk = buf[f[1]], *vv = buf[f[2]], *qc = buf[f[3]];
  hipLaunchKernelGGL(rn_attnpool_tokens_kernel<T>, dim3(bc), dim3(256), 0, st, (const T*)x, m->params[m->pos_p].w, P, C, tok);
  EZ_LAUNCH_CHECK();
  RN_TRY(rn_gemm(m, tok, C, bc * Lt, m->kproj, kk, C, ACT_NONE, nullptr, 0, 0, 0, false, st));
  RN_TRY(rn_gemm(m, tok, C, bc * Lt, m->vproj, vv, C, ACT_NONE, nullptr, 0, 0, 0, false, st));
  RN_TRY(rn_gemm(m, tok, (int64_t)Lt * C, bc, m->qproj, qc, C, ACT_NONE, nullptr, 0, 0, 0, false, st));   // the mean token of every sample
  T* ctx = qc + (size_t)bc * C;        // (the q buffer holds bc rows of C; the rest of it is free)
  AttnArgs a;
  a.k = kk; a.v = vv; a.row_stride = C; a.B = bc; a.L = Lt; a.H = m->heads; a.scale = 0.125f;
  RN_TRY(attention_cls_fwd(a, qc, C, ctx, C, m->dtype, st));
  float* feat = reinterpret_cast<float*>(tok);      // [bc, output_dim] f32 (tok is dead)
  RN_TRY(rn_gemm(m, ctx, C, bc, m->cproj, feat, m->cfg.output_dim, ACT_NONE, nullptr, 0, 0, 0, true, st));
  return l2_normalize_fwd(feat, out, nullptr, bc, m->cfg.output_dim, st);
}

}  // namespace

namespace ezclip {
void set_rn_buffer_bound_mib(int mib) { g_rn_buf_bound = (size_t)(mib > 0 ? mib : 256) << 20; }
}  // namespace ezclip

extern "C" {

int ezclip_rn_create(const ezclip_rn_config* c, ezclip_rn_handle* out) {
  EZ_REQUIRE(c && out, "ezclip_rn_create: null argument");
  EZ_REQUIRE(c->compute_dtype == EZCLIP_F32 || c->compute_dtype == EZCLIP_BF16, "ezclip_rn_create: bad compute_dtype %d", c->compute_dtype);
  EZ_REQUIRE(c->width >= 2 && c->width % 2 == 0 && (c->width * 32) % 64 == 0 && c->width * 32 <= 2048,
             "ezclip_rn_create: width %d (even, width * 32 a multiple of 64 and <= 2048)", c->width);
  EZ_REQUIRE(c->image_resolution >= 32 && c->image_resolution % 32 == 0, "ezclip_rn_create: image_resolution %d must be a multiple of 32",
             c->image_resolution);
  EZ_REQUIRE(c->output_dim > 0 && c->output_dim <= 2048, "ezclip_rn_create: output_dim %d", c->output_dim);
  for (int i = 0; i < 4; ++i) EZ_REQUIRE(c->layers[i] > 0 && c->layers[i] <= 64, "ezclip_rn_create: layers[%d] = %d", i, c->layers[i]);
  ezclip_rn* m = new ezclip_rn();
  m->cfg = *c;
  m->dtype = c->compute_dtype;
  m->esz = dtype_size(m->dtype);
  const int w = c->width;
  m->stem[0] = rn_conv(m, "visual.conv1.weight", "visual.bn1", w / 2, 3, 3, true);
  m->stem[1] = rn_conv(m, "visual.conv2.weight", "visual.bn2", w / 2, w / 2, 3);
  m->stem[2] = rn_conv(m, "visual.conv3.weight", "visual.bn3", w, w / 2, 3);
  int inplanes = w;
  for (int li = 0; li < 4; ++li) {
    const int planes = w << li;
    for (int bi = 0; bi < c->layers[li]; ++bi) {
      ezclip_rn::Block b;
      b.stride = (li > 0 && bi == 0) ? 2 : 1;
      const std::string p = "visual.layer" + std::to_string(li + 1) + "." + std::to_string(bi);
      b.c1 = rn_conv(m, p + ".conv1.weight", p + ".bn1", planes, inplanes, 1);
      b.c2 = rn_conv(m, p + ".conv2.weight", p + ".bn2", planes, planes, 3);
      b.c3 = rn_conv(m, p + ".conv3.weight", p + ".bn3", planes * 4, planes, 1);
      b.has_down = b.stride > 1 || inplanes != planes * 4;
      if (b.has_down) b.down = rn_conv(m, p + ".downsample.0.weight", p + ".downsample.1", planes * 4, inplanes, 1);
      inplanes = planes * 4;
      m->blocks[li].push_back(b);
    }
  }
  m->embed = w * 32;
  m->heads = m->embed / 64;
  m->sp = c->image_resolution / 32;
  m->pos_p = rn_add(m, "visual.attnpool.positional_embedding", {(int64_t)m->sp * m->sp + 1, m->embed});
  m->kproj = rn_linear(m, "visual.attnpool.k_proj", m->embed, m->embed);
  m->qproj = rn_linear(m, "visual.attnpool.q_proj", m->embed, m->embed);
  m->vproj = rn_linear(m, "visual.attnpool.v_proj", m->embed, m->embed);
  m->cproj = rn_linear(m, "visual.attnpool.c_proj", c->output_dim, m->embed);
  *out = m;
  return EZ_OK;
}

void ezclip_rn_destroy(ezclip_rn_handle h) { delete h; }
int ezclip_rn_num_params(ezclip_rn_handle h) { return h ? (int)h->params.size() : 0; }
int ezclip_rn_param_info(ezclip_rn_handle h, int index, const char** name, int64_t* shape, int* ndim) {
  EZ_REQUIRE(h && index >= 0 && index < (int)h->params.size(), "ezclip_rn_param_info: bad index %d", index);
  const auto& p = h->params[index];
  if (name) *name = p.name.c_str();
  if (ndim) *ndim = (int)p.shape.size();
  if (shape) for (size_t i = 0; i < p.shape.size(); ++i) shape[i] = p.shape[i];
  return EZ_OK;
}
int ezclip_rn_bind_param(ezclip_rn_handle h, const char* name, const float* dev) {
  EZ_REQUIRE(h && name && dev, "ezclip_rn_bind_param: null argument");
  for (auto& p : h->params)
    if (p.name == name) {
      if (p.w != dev) h->fresh = false;
      p.w = dev;
      return EZ_OK;
    }
  set_error("ezclip_rn_bind_param: unknown parameter %s", name);
  return EZ_ERR_INVALID;
}
size_t ezclip_rn_shadow_bytes(ezclip_rn_handle h) { return h ? rn_shadow_layout(h, nullptr) : 0; }
int ezclip_rn_set_shadow(ezclip_rn_handle h, void* shadow, size_t bytes) {
  EZ_REQUIRE(h && shadow && ((uintptr_t)shadow % 256) == 0, "ezclip_rn_set_shadow: null / unaligned buffer");
  EZ_REQUIRE(bytes >= rn_shadow_layout(h, nullptr), "ezclip_rn_set_shadow: buffer too small");
  rn_shadow_layout(h, (char*)shadow);
  h->shadow = shadow; h->shadow_bytes = bytes; h->fresh = false;
  return EZ_OK;
}
int ezclip_rn_refresh_weights(ezclip_rn_handle h, void* stream) {
  EZ_REQUIRE(h && h->shadow, "ezclip_rn_refresh_weights: no shadow buffer (ezclip_rn_set_shadow)");
  for (auto& p : h->params) EZ_REQUIRE(p.w != nullptr, "ezclip_rn_refresh_weights: parameter %s is not bound", p.name.c_str());
  hipStream_t st = (hipStream_t)stream;
  EZ_HIP(hipMemsetAsync(h->zero, 0, 256, st));
  int rc = EZ_OK;
  rn_for_each(h, [&](ezclip_rn::Conv& c) {
    if (rc != EZ_OK) return;
    const float* W = h->params[c.w].w;
    const float *g = nullptr, *b = nullptr, *mu = nullptr, *var = nullptr, *lb = nullptr;
    if (c.bn >= 0) { g = h->params[c.bn].w; b = h->params[c.bn + 1].w; mu = h->params[c.bn + 2].w; var = h->params[c.bn + 3].w; }
    if (c.lin_b >= 0) lb = h->params[c.lin_b].w;
    if (h->dtype == EZCLIP_BF16)
      hipLaunchKernelGGL(rn_pack_conv_kernel<bf16_t>, dim3(c.Opad), dim3(256), 0, st, W, g, b, mu, var, lb, c.O, c.I, c.k, c.k, c.Opad, c.Cp,
                         c.ldk, c.cfirst, (bf16_t*)c.s, c.bias);
    else
      hipLaunchKernelGGL(rn_pack_conv_kernel<float>, dim3(c.Opad), dim3(256), 0, st, W, g, b, mu, var, lb, c.O, c.I, c.k, c.k, c.Opad, c.Cp,
                         c.ldk, c.cfirst, (float*)c.s, c.bias);
    rc = check_hip(hipGetLastError(), "rn_pack_conv_kernel");
  });
  if (rc == EZ_OK) h->fresh = true;
  return rc;
}
size_t ezclip_rn_workspace_bytes(ezclip_rn_handle h, int batch) {
  if (!h || batch <= 0) return 0;
  return kRnBufs * rn_act_bytes(h, rn_chunk(h, batch)) + 256;
}
int ezclip_rn_encode_image(ezclip_rn_handle h, const float* pixels, int batch, float* out, void* ws, size_t ws_bytes, void* stream) {
  EZ_REQUIRE(h && pixels && out && ws && batch > 0, "ezclip_rn_encode_image: null / empty argument");
  EZ_REQUIRE(h->fresh, "ezclip_rn_encode_image: weights not packed (ezclip_rn_refresh_weights)");
  EZ_REQUIRE(((uintptr_t)ws % 256) == 0 && ws_bytes >= ezclip_rn_workspace_bytes(h, batch), "ezclip_rn_encode_image: workspace too small / unaligned");
  const int bc = rn_chunk(h, batch);
  const int R = h->cfg.image_resolution;
  for (int b0 = 0; b0 < batch; b0 += bc) {
    const int nb = batch - b0 < bc ? batch - b0 : bc;
    const float* px = pixels + (size_t)b0 * 3 * R * R;
    float* o = out + (size_t)b0 * h->cfg.output_dim;
    int rc = h->dtype == EZCLIP_BF16 ? rn_forward_chunk<bf16_t>(h, px, nb, o, (char*)ws, (hipStream_t)stream)
                                     : rn_forward_chunk<float>(h, px, nb, o, (char*)ws, (hipStream_t)stream);
    if (rc != EZ_OK) return rc;
  }
  return EZ_OK;
}

}  // extern "C"
