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
#include <cstring>
#include <string>
#include <algorithm>
#include <unordered_map>
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
  struct Param { std::string name; std::vector<int64_t> shape; const float* w = nullptr; float* g = nullptr; };   // g: gradient buffer (training path)
  std::vector<Param> params;
  struct Conv {             // one convolution (+ BatchNorm) or Linear, with its packed copy
    int w = -1, bn = -1, lin_b = -1;     // param indices: weight; bn.weight (bias, running_mean, running_var follow); Linear bias
    int O = 0, I = 0, k = 1;
    int Opad = 0, Cp = 0, ldk = 0;       // packed [Opad][ldk]; Cp = padded input channels (per tap)
    int cfirst = 0;
    void* s = nullptr;
    float* bias = nullptr;
    // ---- training path (ezclip_rn_train_*): unfolded weight [Opad][ldk] (no BatchNorm in it), input-gradient weight
    // [Cp][k*k*Opad] (null: the stem's first convolution, whose input is the pixels), and what the training forward saved
    void* st = nullptr;
    void* sd = nullptr;
    const void* in = nullptr;            // input activation [Min, Cp] (NHWC rows) -- or the explicit im2col for the stem's first conv
    void* z = nullptr;                   // convolution output [M, Opad]
    void* y = nullptr;                   // after BatchNorm (+ identity) (+ ReLU) [M, Opad]
    float* mean = nullptr;               // [Opad] batch statistics
    float* rstd = nullptr;
    int64_t M = 0;                       // rows of z / y
    int H = 0;                           // spatial side of the grid the convolution runs on
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
  // ---- training path
  void* tshadow = nullptr;
  size_t tshadow_bytes = 0;
  bool tfresh = false;
  float* junk_bias = nullptr;            // [2048] floats the packing kernel writes its (unused) bias row to
  struct Saved {                         // of the last training forward
    int B = 0;
    void* col0 = nullptr;                // stem: explicit im2col of the pixels
    void* pool0 = nullptr;               // stem: after AvgPool2d(2)
    std::vector<void*> o2p, idp;         // per block (flattened order): pooled conv2 output / pooled block input (stride-2 blocks)
    void *tok = nullptr, *kk = nullptr, *vv = nullptr, *qc = nullptr, *ctx = nullptr;
    float *feat = nullptr, *inv_norm = nullptr;
    const void* last = nullptr;          // the tower's last activation (input of the attention pool)
  } saved;
  // Every saved workspace a training forward has filled: base pointer -> (batch, serial number).  A backward pass names the workspace
  // of ITS forward and every pointer is re-derived from that base (rn_train_bind), so two forwards before one backward (micro-batches
  // summed into one loss, a feature call in between) each keep their own activations; a workspace nobody filled is refused.
  struct Filled { int B; uint64_t serial; };
  std::unordered_map<const void*, Filled> filled;
  uint64_t serial = 0;
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


// ==== training path =================================================================================================================
// BatchNorm in training mode and the tower's backward pass (reference: nn.BatchNorm2d.train() + autograd through Bottleneck / ModifiedResNet /
// AttentionPool2d, modeling_chineseclip.py:27-167; the steps are those of the tests' CPU restatement, train_step_grads_by_steps).
// The whole batch runs at once (the statistics are the batch's); every convolution output z and every activation y is kept in the SAVED
// workspace until the backward pass has used it.  Convolutions run on unfolded packed weights, then rn_bn_train_fwd (resnet_train.hip).

size_t rn_train_shadow_layout(ezclip_rn* m, char* base) {
  size_t off = 0;
  auto take = [&](size_t bytes) { void* p = base ? base + off : nullptr; off += (bytes + 255) / 256 * 256; return p; };
  float* junk = (float*)take(2048 * 4);
  if (base) m->junk_bias = junk;
  bool first = true;
  rn_for_each(m, [&](ezclip_rn::Conv& c) {
    const bool conv = c.bn >= 0;
    void* st = conv ? take((size_t)c.Opad * c.ldk * m->esz) : nullptr;                       // (Linear layers use the eval copy: no BatchNorm to unfold)
    void* sd = (first && conv) ? nullptr : take((size_t)c.Cp * c.k * c.k * rup(c.Opad, 64) * m->esz);
    if (base) { c.st = st; c.sd = sd; }
    first = false;
  });
  return off + 256;
}

struct RnBump {                          // bump allocator over a workspace; null base: size query
  char* base; size_t off = 0;
  explicit RnBump(void* b) : base((char*)b) {}
  void* take(size_t bytes) { void* p = base ? base + off : nullptr; off += (bytes + 255) / 256 * 256; return p; }
};

// The saved workspace of one training forward over B images: ONE walk defines its layout.  `on_conv(conv, M, H, in, lda)` is called for
// every convolution in forward order with its input (a pointer inside the workspace, null in a size query) and must take z, y, mean, rstd
// from the allocator in that order (rn_train_take_conv).  Used by the size query, by the forward pass (which then only launches) and by
// the backward pass (which re-derives every pointer from the base of the workspace ITS forward filled).
inline void rn_train_take_conv(ezclip_rn* m, RnBump& a, ezclip_rn::Conv& c, int64_t M, int H, const void* in, bool assign) {
  void* z = a.take((size_t)M * c.Opad * m->esz);
  void* y = a.take((size_t)M * c.Opad * m->esz);
  float* mean = (float*)a.take((size_t)c.Opad * 4);
  float* rstd = (float*)a.take((size_t)c.Opad * 4);
  if (assign) { c.in = in; c.M = M; c.H = H; c.z = z; c.y = y; c.mean = mean; c.rstd = rstd; }
}
void rn_train_plan(ezclip_rn* m, int B, RnBump& a, bool assign) {
  const int R = m->cfg.image_resolution;
  const size_t e = m->esz;
  ezclip_rn::Saved s;
  s.B = B;
  int H = R / 2;
  int64_t M = (int64_t)B * H * H;
  s.col0 = a.take((size_t)M * m->stem[0].ldk * e);
  rn_train_take_conv(m, a, m->stem[0], M, H, s.col0, assign);
  rn_train_take_conv(m, a, m->stem[1], M, H, m->stem[0].y, assign);
  rn_train_take_conv(m, a, m->stem[2], M, H, m->stem[1].y, assign);
  s.pool0 = a.take((size_t)(M / 4) * m->stem[2].Opad * e);
  H /= 2; M /= 4;
  const void* x = s.pool0;
  for (auto& L : m->blocks)
    for (auto& b : L) {
      rn_train_take_conv(m, a, b.c1, M, H, x, assign);
      rn_train_take_conv(m, a, b.c2, M, H, b.c1.y, assign);
      int64_t Mo = M; int Ho = H;
      void *o2p = nullptr, *idp = nullptr;
      const void* o2 = b.c2.y;
      const void* idn = x;
      if (b.stride > 1) {
        Mo = M / 4; Ho = H / 2;
        o2p = a.take((size_t)Mo * b.c2.Opad * e);
        idp = a.take((size_t)Mo * b.c1.Cp * e);
        o2 = o2p; idn = idp;
      }
      s.o2p.push_back(o2p); s.idp.push_back(idp);
      if (b.has_down) rn_train_take_conv(m, a, b.down, Mo, Ho, idn, assign);
      rn_train_take_conv(m, a, b.c3, Mo, Ho, o2, assign);
      x = b.c3.y; M = Mo; H = Ho;
    }
  s.last = x;
  const int Lt = m->sp * m->sp + 1, C = m->embed;
  s.tok = a.take((size_t)B * Lt * C * e);
  s.kk = a.take((size_t)B * Lt * C * e);
  s.vv = a.take((size_t)B * Lt * C * e);
  s.qc = a.take((size_t)B * C * e);
  s.ctx = a.take((size_t)B * C * e);
  s.feat = (float*)a.take((size_t)B * m->cfg.output_dim * 4);
  s.inv_norm = (float*)a.take((size_t)B * 4);
  if (assign) m->saved = s;
}

size_t rn_train_saved_bytes(ezclip_rn* m, int B) {
  RnBump a(nullptr);
  rn_train_plan(m, B, a, false);             // (a size query between a forward and its backward touches nothing that was saved)
  return a.off + 256;
}
// point the handle at the activations of the training forward that filled `saved_ws`
void rn_train_bind(ezclip_rn* m, int B, char* saved_ws) {
  RnBump a(saved_ws);
  rn_train_plan(m, B, a, true);
}

// largest NHWC activation / gradient of a batch, and the largest explicit im2col of a 3x3 convolution's input (the `col` scratch of the
// backward pass: sized for the explicit route of every convolution, also the home of the operands-once kernels' partial sums)
size_t rn_train_max_act_bytes(const ezclip_rn* m, int B) { return rn_act_bytes(m, B); }
size_t rn_train_max_col_bytes(const ezclip_rn* m, int B) {
  const int R = m->cfg.image_resolution;
  size_t mx = (size_t)B * (R / 2) * (R / 2) * 9 * m->stem[1].Cp;                 // stem conv2 / conv3 inputs
  int H = R / 4;
  for (const auto& L : m->blocks)
    for (const auto& b : L) {
      const size_t v = (size_t)B * H * H * 9 * b.c2.Cp;
      if (v > mx) mx = v;
      if (b.stride > 1) H /= 2;
    }
  return (mx * m->esz + 255) / 256 * 256;
}
constexpr int kRnTrainBufs = 6;
// largest packed weight gradient [Opad][ldk] f32 of a CONVOLUTION (the attention pool's Linear layers write their gradients
// straight into the bound buffers).  One function for the size query and for the backward pass: round 5's first GPU run of this
// path faulted because the backward pass also counted the Linear layers (2048 x 2048 x 4 B at width 64) and carved more than
// rn_train_scratch_bytes had reserved -- the BatchNorm scratch behind it then lay outside the buffer.
size_t rn_train_wgrad_bytes(const ezclip_rn* m) {
  size_t wg = 0;
  auto upd = [&](const ezclip_rn::Conv& c) { const size_t v = (size_t)rup(c.Opad, 64) * c.ldk * 4; if (v > wg) wg = v; };
  for (const auto& c : m->stem) upd(c);
  for (const auto& L : m->blocks) for (const auto& b : L) { upd(b.c1); upd(b.c2); upd(b.c3); if (b.has_down) upd(b.down); }
  return (wg + 255) / 256 * 256;
}
size_t rn_train_scratch_bytes(const ezclip_rn* m, int B) {
  const int Lt = m->sp * m->sp + 1;
  const size_t wg = rn_train_wgrad_bytes(m);
  const size_t act = rn_train_max_act_bytes(m, B);
  const size_t tokb = ((size_t)B * Lt * m->embed * m->esz + 255) / 256 * 256;
  const size_t bn = rn_bn_scratch_bytes((int64_t)B * (m->cfg.image_resolution / 2) * (m->cfg.image_resolution / 2), 2048) + 256;
  return kRnTrainBufs * (act > tokb ? act : tokb) + rn_train_max_col_bytes(m, B) + wg + bn + 4096;
}

// convolution on the unfolded weight: z = conv(x) (no bias, no activation)
int rn_gemm_t(const ezclip_rn* m, const void* A, int64_t lda, int64_t M, const ezclip_rn::Conv& c, void* C, int convH, hipStream_t st) {
  GemmArgs g;
  g.A = A; g.lda = lda; g.B = c.st; g.ldb = c.ldk; g.C = C; g.ldc = c.Opad;
  g.M = (int)M; g.N = c.Opad; g.K = c.ldk; g.act = ACT_NONE;
  if (convH > 0) { g.conv_H = convH; g.conv_W = convH; g.conv_C = c.Cp; g.conv_zero = m->zero; }
  return gemm_nt(g, m->dtype, st);
}

template <typename T>
int rn_train_forward(ezclip_rn* m, const float* px, int B, float* out, char* saved_ws, size_t saved_bytes, float* bn_scratch, hipStream_t st) {
  const int R = m->cfg.image_resolution;
  {
    RnBump probe(nullptr);
    rn_train_plan(m, B, probe, false);
    EZ_REQUIRE(probe.off <= saved_bytes, "rn_train_forward: the pass saves %zu bytes into a workspace of %zu", probe.off, saved_bytes);
  }
  m->filled.erase(saved_ws);                 // (until every launch below is enqueued the workspace holds no complete pass)
  rn_train_bind(m, B, saved_ws);             // every pointer of the pass: Conv::in / z / y / mean / rstd, m->saved.*
  int rc = EZ_OK;
  // one convolution + BatchNorm(train) [+ identity] [+ ReLU] on the bound pointers
  auto conv_bn = [&](ezclip_rn::Conv& c, int64_t lda, int convH, const void* identity, int relu) {
    if (rc != EZ_OK) return;
    rc = rn_gemm_t(m, c.in, lda, c.M, c, c.z, convH, st);
    if (rc != EZ_OK) return;
    rc = rn_bn_train_fwd(c.z, c.M, c.O, c.Opad, m->params[c.bn].w, m->params[c.bn + 1].w, const_cast<float*>(m->params[c.bn + 2].w),
                         const_cast<float*>(m->params[c.bn + 3].w), 0.1f, kBnEps, identity, relu, c.y, c.mean, c.rstd, bn_scratch, m->dtype, st);
  };
  int H = R / 2;
  int64_t M = (int64_t)B * H * H;
  RN_TRY(launch1d(rn_stem_im2col_kernel<T>, M * m->stem[0].ldk, st, px, B, R, H, m->stem[0].ldk, (T*)m->saved.col0));
  conv_bn(m->stem[0], m->stem[0].ldk, 0, nullptr, 1);
  conv_bn(m->stem[1], m->stem[0].Opad, H, nullptr, 1);
  conv_bn(m->stem[2], m->stem[1].Opad, H, nullptr, 1);
  RN_TRY(rc);
  int C = m->stem[2].Opad;
  RN_TRY(launch1d(rn_avgpool2_kernel<T>, M / 4 * (C / 4), st, (const T*)m->stem[2].y, M / 4 * (C / 4), H, H, C, (T*)m->saved.pool0));
  H /= 2; M /= 4;
  const void* x = m->saved.pool0;
  size_t bi = 0;
  for (auto& L : m->blocks)
    for (auto& b : L) {
      conv_bn(b.c1, C, 0, nullptr, 1);
      conv_bn(b.c2, b.c1.Opad, H, nullptr, 1);
      RN_TRY(rc);
      int64_t Mo = M; int Ho = H;
      if (b.stride > 1) {
        Mo = M / 4; Ho = H / 2;
        const int Cm = b.c2.Opad;
        RN_TRY(launch1d(rn_avgpool2_kernel<T>, Mo * (Cm / 4), st, (const T*)b.c2.y, Mo * (Cm / 4), H, H, Cm, (T*)m->saved.o2p[bi]));
        RN_TRY(launch1d(rn_avgpool2_kernel<T>, Mo * (C / 4), st, (const T*)x, Mo * (C / 4), H, H, C, (T*)m->saved.idp[bi]));
      }
      const void* identity = b.stride > 1 ? m->saved.idp[bi] : x;
      if (b.has_down) {
        conv_bn(b.down, C, 0, nullptr, 0);
        identity = b.down.y;
      }
      conv_bn(b.c3, b.c2.Opad, 0, identity, 1);
      RN_TRY(rc);
      x = b.c3.y; C = b.c3.Opad; M = Mo; H = Ho;
      ++bi;
    }
  // ---- AttentionPool2d (as rn_forward_chunk, everything kept)
  EZ_REQUIRE(H == m->sp && C == m->embed, "rn_train_forward: tower ends at %d x %d x %d, attention pool expects %d x %d x %d", H, H, C, m->sp,
             m->sp, m->embed);
  const int P = H * H, Lt = P + 1;
  hipLaunchKernelGGL(rn_attnpool_tokens_kernel<T>, dim3(B), dim3(256), 0, st, (const T*)x, m->params[m->pos_p].w, P, C, (T*)m->saved.tok);
  EZ_LAUNCH_CHECK();
  RN_TRY(rn_gemm(m, m->saved.tok, C, B * Lt, m->kproj, m->saved.kk, C, ACT_NONE, nullptr, 0, 0, 0, false, st));
  RN_TRY(rn_gemm(m, m->saved.tok, C, B * Lt, m->vproj, m->saved.vv, C, ACT_NONE, nullptr, 0, 0, 0, false, st));
  RN_TRY(rn_gemm(m, m->saved.tok, (int64_t)Lt * C, B, m->qproj, m->saved.qc, C, ACT_NONE, nullptr, 0, 0, 0, false, st));
  AttnArgs at;
  at.k = m->saved.kk; at.v = m->saved.vv; at.row_stride = C; at.B = B; at.L = Lt; at.H = m->heads; at.scale = 0.125f;
  RN_TRY(attention_cls_fwd(at, m->saved.qc, C, m->saved.ctx, C, m->dtype, st));
  RN_TRY(rn_gemm(m, m->saved.ctx, C, B, m->cproj, m->saved.feat, m->cfg.output_dim, ACT_NONE, nullptr, 0, 0, 0, true, st));
  RN_TRY(l2_normalize_fwd(m->saved.feat, out, m->saved.inv_norm, B, m->cfg.output_dim, st));
  if (m->filled.size() >= 64) {              // (workspaces whose owners are long gone: forget the oldest half)
    std::vector<std::pair<uint64_t, const void*>> v;
    for (auto& kv : m->filled) v.push_back({kv.second.serial, kv.first});
    std::sort(v.begin(), v.end());
    for (size_t i = 0; i < v.size() / 2; ++i) m->filled.erase(v[i].second);
  }
  m->filled[saved_ws] = {B, ++m->serial};
  return EZ_OK;
}

// d tok[b][0] += dq[b] . Wq  is a GEMM with a residual; d x[b][p] = d tok[b][1 + p] + d tok[b][0] / P
template <typename T>
__global__ __launch_bounds__(256) void rn_attnpool_tokens_bwd_kernel(const T* __restrict__ dtok, int P, int C, T* __restrict__ dx) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float m0 = Elem<T>::ld(dtok + (int64_t)b * (P + 1) * C + c) / (float)P;
    for (int p = 0; p < P; ++p)
      Elem<T>::st(dx + ((int64_t)b * P + p) * C + c, Elem<T>::ld(dtok + ((int64_t)b * (P + 1) + 1 + p) * C + c) + m0);
  }
}

// f32 [rows, cols] -> T [rows, ld] with the columns cols .. ld - 1 zeroed (the contraction dimension of a product must be a multiple of 64)
template <typename T>
__global__ __launch_bounds__(256) void rn_pad_cast_rows_kernel(const float* __restrict__ src, int64_t n, int cols, int ld, T* __restrict__ dst) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int c = (int)(idx % ld);
  const int64_t r = idx / ld;
  Elem<T>::st(dst + idx, c < cols ? src[r * cols + c] : 0.f);
}

// input gradient of a convolution: dx = dz (*) Wd  (1x1: plain product on the transposed weight; 3x3: the implicit convolution)
int rn_dgrad(const ezclip_rn* m, const ezclip_rn::Conv& c, const void* dz, int64_t M, int H, void* dx, const void* add, hipStream_t st) {
  GemmArgs g;
  const int opad64 = rup(c.Opad, 64);
  g.A = dz; g.lda = c.Opad; g.B = c.sd; g.ldb = (int64_t)c.k * c.k * opad64; g.C = dx; g.ldc = c.Cp;
  g.M = (int)M; g.N = c.Cp; g.K = c.k * c.k * opad64;
  g.R = add; g.ldr = c.Cp;
  if (c.k == 3) { g.conv_H = H; g.conv_W = H; g.conv_C = opad64; g.conv_zero = m->zero; }
  return gemm_nt(g, m->dtype, st);
}

// weight gradient of a convolution into its bound gradient buffer: dWp = dz^T . x (1 x 1) / dz^T . (3 x 3 neighbourhoods of x), then unpacked.
// Which product (bf16; fp32 always takes the generic kernel):
//   3 x 3, 64 / 128 padded channels in and out (stem conv2 / conv3, layer1, layer2)  rn_wgrad3x3_c64: operands once, result in registers
//   3 x 3, 256-multiples (layer3 / layer4)                                           explicit column matrix (small there) + 8-phase TN kernel
//   3 x 3, anything else                                                             generic TN kernel gathering the neighbourhoods itself
//   1 x 1 / the stem's first convolution, 256-multiples                              8-phase TN kernel
//   1 x 1 / the stem's first convolution, many pixels and a small result             rn_tn_skinny
//   anything else                                                                    generic TN kernel
// `col` holds the explicit column matrix or the per-workgroup partials of the operands-once kernels.
// EZCLIP_RN_EXPLICIT_IM2COL (A/B switch): 1 = explicit column matrix + generic / 8-phase kernels everywhere (the first version), 2 = no
// rn_wgrad3x3_c64 and no rn_tn_skinny, 3 = no rn_tn_skinny.
static int rn_wgrad_mode() {       // read per backward pass (tests switch it inside one process: same operands, the other summation order)
  const char* e = getenv("EZCLIP_RN_EXPLICIT_IM2COL");
  return e ? atoi(e) : 0;
}

template <typename T>
int rn_wgrad(const ezclip_rn* m, const ezclip_rn::Conv& c, const void* dz, const void* x, int64_t M, int H, int B, void* col, size_t col_bytes,
             float* dwp, hipStream_t st) {
  float* gw = m->params[c.w].g;
  const int mode = rn_wgrad_mode();
  GemmTNArgs t;
  t.A = dz; t.lda = c.Opad; t.C = dwp; t.ldc = c.ldk;
  t.M = (int)M; t.N = c.Opad; t.K = c.ldk; t.accumulate = 0;
  bool done = false;
  if (!c.cfirst && c.k == 3) {
    if ((mode == 0 || mode == 3) && rn_wgrad3x3_c64_eligible(B, H, H, c.Cp, c.Opad, m->dtype, col_bytes)) {
      RN_TRY(rn_wgrad3x3_c64(x, dz, B, H, H, c.Cp, c.Opad, col, col_bytes, dwp, c.ldk, 0, st));
      done = true;
    } else {
      t.B = col; t.ldb = 9 * (int64_t)c.Cp;
      if (mode != 1 && H >= 4 && !gemm_tn_8p_eligible(t, m->dtype)) {
        t.B = x; t.ldb = c.Cp; t.conv_H = H; t.conv_W = H; t.conv_C = c.Cp;      // bit-identical to the explicit route
      } else {
        RN_TRY(rn_im2col3x3(x, B, H, H, c.Cp, col, m->dtype, st));
      }
    }
  } else {
    t.B = x; t.ldb = c.cfirst ? c.ldk : c.Cp;               // the stem's first convolution: x IS its explicit column matrix [M, ldk]
    if (mode == 0 && !gemm_tn_8p_eligible(t, m->dtype) && rn_tn_skinny_eligible(M, t.N, t.K, t.lda, t.ldb, t.ldc, m->dtype, col_bytes)) {
      RN_TRY(rn_tn_skinny(t.A, t.lda, t.B, t.ldb, t.C, t.ldc, M, t.N, t.K, 0, col, col_bytes, st));
      done = true;
    }
  }
  if (!done) RN_TRY(gemm_tn(t, m->dtype, st));
  if (c.cfirst) {
    // K index (c, ky, kx) = the weight's own [I][3][3] order: a strided copy of the first I * 9 columns
    EZ_HIP(hipMemcpy2DAsync(gw, (size_t)c.I * 9 * 4, dwp, (size_t)c.ldk * 4, (size_t)c.I * 9 * 4, c.O, hipMemcpyDeviceToDevice, st));
    return EZ_OK;
  }
  return rn_unpack_wgrad(dwp, c.ldk, c.O, c.I, c.k, c.Cp, 0, gw, st);
}

template <typename T>
int rn_train_backward(ezclip_rn* m, const float* features, const float* d_features, int B, char* scratch, size_t scratch_bytes, hipStream_t st) {
  const int E = m->cfg.output_dim, C = m->embed, P = m->sp * m->sp, Lt = P + 1;
  RnBump a(scratch);
  const size_t act = rn_train_max_act_bytes(m, B);
  const size_t tokb = ((size_t)B * Lt * C * m->esz + 255) / 256 * 256;
  const size_t bufb = act > tokb ? act : tokb;
  void* buf[kRnTrainBufs];
  for (auto& b : buf) b = a.take(bufb);
  const size_t col_bytes = rn_train_max_col_bytes(m, B);
  void* col = a.take(col_bytes);
  float* dwp = (float*)a.take(rn_train_wgrad_bytes(m));                           // (the SAME bytes rn_train_scratch_bytes reserved)
  float* bn_scratch = (float*)a.take(rn_bn_scratch_bytes((int64_t)B * (m->cfg.image_resolution / 2) * (m->cfg.image_resolution / 2), 2048));
  EZ_REQUIRE(a.off <= scratch_bytes, "rn_train_backward: the pass carves %zu bytes out of a scratch of %zu (rn_train_scratch_bytes is out of step)", a.off, scratch_bytes);
  auto G = [&](int pi) { return m->params[pi].g; };

  // ---- attention pool --------------------------------------------------------------------------------------------------------
  float* dfeat = (float*)buf[0];                                                   // [B, E] f32
  RN_TRY(l2_normalize_bwd(features, d_features, m->saved.inv_norm, dfeat, B, E, st));
  const int E64 = rup(E, 64);
  T* dfeat_t = (T*)buf[1];                                                         // [B, E64], zero beyond E
  RN_TRY(launch1d(rn_pad_cast_rows_kernel<T>, (int64_t)B * E64, st, (const float*)dfeat, (int64_t)B * E64, E, E64, dfeat_t));
  // c_proj: feat = ctx . Wc^T + bc
  if (G(m->cproj.lin_b)) { EZ_HIP(hipMemsetAsync(G(m->cproj.lin_b), 0, (size_t)E * 4, st)); RN_TRY(colsum_add(dfeat, E, B, E, G(m->cproj.lin_b), EZCLIP_F32, st)); }
  if (G(m->cproj.w)) {
    GemmTNArgs t; t.A = dfeat_t; t.lda = E64; t.B = m->saved.ctx; t.ldb = C; t.C = G(m->cproj.w); t.ldc = C; t.M = B; t.N = E; t.K = C;
    RN_TRY(gemm_tn(t, m->dtype, st));
  }
  T* dctx = (T*)buf[2];                                                            // [B, C]
  { GemmArgs g; g.A = dfeat_t; g.lda = E64; g.B = m->cproj.sd; g.ldb = E64; g.C = dctx; g.ldc = C; g.M = B; g.N = C; g.K = E64;
    RN_TRY(gemm_nt(g, m->dtype, st)); }
  // one-query attention backward
  T* dk = (T*)buf[3]; T* dv = (T*)buf[4]; T* dq = (T*)buf[5];
  AttnBwdArgs ab;
  ab.f.k = m->saved.kk; ab.f.v = m->saved.vv; ab.f.row_stride = C; ab.f.B = B; ab.f.L = Lt; ab.f.H = m->heads; ab.f.scale = 0.125f;
  ab.dk = dk; ab.dv = dv;
  RN_TRY(attention_cls_bwd(ab, m->saved.qc, C, m->saved.ctx, dctx, C, m->dtype, st, dq, C));
  // projections: bias gradients = column sums, weight gradients = d^T . tok, token gradient = dk . Wk + dv . Wv (+ dq . Wq on row 0)
  auto lin_grads = [&](const ezclip_rn::Conv& c, const T* d, int64_t rows, const void* x, int64_t ldx) -> int {
    if (G(c.lin_b)) { EZ_HIP(hipMemsetAsync(G(c.lin_b), 0, (size_t)c.O * 4, st)); RN_TRY(colsum_add(d, C, (int)rows, c.O, G(c.lin_b), m->dtype, st)); }
    if (G(c.w)) { GemmTNArgs t; t.A = d; t.lda = C; t.B = x; t.ldb = ldx; t.C = G(c.w); t.ldc = c.I; t.M = (int)rows; t.N = c.O; t.K = c.I; RN_TRY(gemm_tn(t, m->dtype, st)); }
    return EZ_OK;
  };
  RN_TRY(lin_grads(m->kproj, dk, (int64_t)B * Lt, m->saved.tok, C));
  RN_TRY(lin_grads(m->vproj, dv, (int64_t)B * Lt, m->saved.tok, C));
  RN_TRY(lin_grads(m->qproj, dq, B, m->saved.tok, (int64_t)Lt * C));
  T* dtok = (T*)buf[0];                                                            // (dfeat is dead) [B * Lt, C]
  { GemmArgs g; g.A = dk; g.lda = C; g.B = m->kproj.sd; g.ldb = C; g.C = dtok; g.ldc = C; g.M = B * Lt; g.N = C; g.K = C; RN_TRY(gemm_nt(g, m->dtype, st)); }
  { GemmArgs g; g.A = dv; g.lda = C; g.B = m->vproj.sd; g.ldb = C; g.C = dtok; g.ldc = C; g.M = B * Lt; g.N = C; g.K = C; g.R = dtok; g.ldr = C;
    RN_TRY(gemm_nt(g, m->dtype, st)); }
  { GemmArgs g; g.A = dq; g.lda = C; g.B = m->qproj.sd; g.ldb = C; g.C = dtok; g.ldc = (int64_t)Lt * C; g.M = B; g.N = C; g.K = C; g.R = dtok;
    g.ldr = (int64_t)Lt * C; RN_TRY(gemm_nt(g, m->dtype, st)); }
  if (G(m->pos_p)) {
    EZ_HIP(hipMemsetAsync(G(m->pos_p), 0, (size_t)Lt * C * 4, st));
    RN_TRY(batch_sum_add(dtok, B, Lt, Lt, C, G(m->pos_p), m->dtype, st));
  }
  T* dx = (T*)buf[1];                                                              // gradient of the tower's last activation [B * P, C]
  hipLaunchKernelGGL(rn_attnpool_tokens_bwd_kernel<T>, dim3(B), dim3(256), 0, st, (const T*)dtok, P, C, dx);
  EZ_LAUNCH_CHECK();
  int dxi = 1;

  // ---- one convolution + BatchNorm backwards: dy (gradient of y; masked by [y > 0] when `relu`) -> dz, BatchNorm / weight gradients,
  //      input gradient into `dst` (+ `add`); dres (optional): the masked dy, i.e. the gradient of the identity input
  // EZCLIP_RN_DEBUG=1: print the sum of squares of every intermediate of the pass (host copy, synchronous): two runs on the same
  // saved state must print the same numbers -- the first line that differs names a stage that is not deterministic
  static const bool dbg = getenv("EZCLIP_RN_DEBUG") != nullptr;
  int dbg_no = 0;
  auto dsum = [&](const char* what, const void* p, int64_t n) {
    if (!dbg || p == nullptr) return;
    std::vector<T> h((size_t)n);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), p, (size_t)n * sizeof(T), hipMemcpyDeviceToHost);
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) {
      double v;
      if constexpr (sizeof(T) == 4) v = (double)*reinterpret_cast<const float*>(&h[(size_t)i]);
      else { const uint32_t w = (uint32_t)*reinterpret_cast<const uint16_t*>(&h[(size_t)i]) << 16; v = (double)__builtin_bit_cast(float, w); }
      s += v * v;
    }
    fprintf(stderr, "[rn-dbg] %3d %-24s n %9lld  sumsq %.17g\n", dbg_no, what, (long long)n, s);
    // EZCLIP_RN_DEBUG_DUMP=<n>: the intermediates of lines 0 .. n as raw elements into /tmp/rn_dbg_<line>.bin
    static const int dump = getenv("EZCLIP_RN_DEBUG_DUMP") ? atoi(getenv("EZCLIP_RN_DEBUG_DUMP")) : -1;
    if (dbg_no <= dump) {
      char path[64];
      snprintf(path, sizeof(path), "/tmp/rn_dbg_%d.bin", dbg_no);
      if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), sizeof(T), (size_t)n, f); fclose(f); }
    }
    ++dbg_no;
  };
  auto conv_bn_bwd = [&](ezclip_rn::Conv& c, const void* dy, bool relu, void* dz, void* dres, void* dst, const void* add) -> int {
    dsum("dy", dy, c.M * c.Opad);
    RN_TRY(rn_bn_train_bwd(dy, relu ? c.y : nullptr, c.z, c.M, c.O, c.Opad, m->params[c.bn].w, c.mean, c.rstd, dz, dres, G(c.bn), G(c.bn + 1), 0,
                           bn_scratch, m->dtype, st));
    dsum("dz", dz, c.M * c.Opad);
    dsum("dres", dres, c.M * c.Opad);
    RN_TRY(rn_wgrad<T>(m, c, dz, c.in, c.M, c.H, B, col, col_bytes, dwp, st));
    if (dst != nullptr && c.sd != nullptr) {
      RN_TRY(rn_dgrad(m, c, dz, c.M, c.H, dst, add, st));
      dsum("dx", dst, c.M * c.Cp);
    }
    return EZ_OK;
  };
  // free buffer picker: any index not in `used`
  auto pick = [&](std::initializer_list<int> used) { for (int i = 0; i < kRnTrainBufs; ++i) { bool u = false; for (int v : used) u |= (v == i); if (!u) return i; } return -1; };

  // ---- blocks, last to first -------------------------------------------------------------------------------------------------
  std::vector<ezclip_rn::Block*> order;
  for (auto& L : m->blocks) for (auto& b : L) order.push_back(&b);
  for (int bi = (int)order.size() - 1; bi >= 0; --bi) {
    ezclip_rn::Block& b = *order[bi];
    const int i_dz = pick({dxi}), i_g = pick({dxi, i_dz}), i_t = pick({dxi, i_dz, i_g});
    // conv3 + bn3 (+ identity, ReLU): g = dy * [y3 > 0] is also the identity branch's gradient
    void* d_o2 = buf[i_t];                                                         // gradient of conv3's input [Mo, C2p]
    RN_TRY(conv_bn_bwd(b.c3, buf[dxi], true, buf[i_dz], buf[i_g], d_o2, nullptr));
    const int64_t Min = b.c1.M;
    const int Hin = b.c1.H;
    // identity branch FIRST (round 6): its gradient then joins the main branch's inside the epilogue of conv1's input-gradient product
    // (GemmArgs::R) -- one rounding of the f32 sum instead of a separate pass over both (rn_add_kernel was 3.2 % of the RN50 step)
    int i_add = i_g;
    if (b.has_down) {
      const int i_e = pick({i_g, i_t}), i_f = pick({i_g, i_t, i_e});
      RN_TRY(conv_bn_bwd(b.down, buf[i_g], false, buf[i_e], nullptr, buf[i_f], nullptr));   // -> d (pooled) x
      i_add = i_f;
      if (b.stride > 1) {
        const int i_h = pick({i_t, i_f});
        RN_TRY(rn_avgpool2_bwd(buf[i_f], B, Hin, Hin, b.c1.Cp, buf[i_h], m->dtype, st));
        i_add = i_h;
      }
    }
    // main branch: (pool) conv2, conv1
    void* d_y2 = d_o2;
    int i_y2 = i_t;
    if (b.stride > 1) {
      const int i_u = pick({i_add, i_t});
      RN_TRY(rn_avgpool2_bwd(d_o2, B, Hin, Hin, b.c2.Opad, buf[i_u], m->dtype, st));
      d_y2 = buf[i_u]; i_y2 = i_u;
    }
    const int i_a = pick({i_add, i_y2}), i_b = pick({i_add, i_y2, i_a});
    RN_TRY(conv_bn_bwd(b.c2, d_y2, true, buf[i_a], nullptr, buf[i_b], nullptr));   // -> d y1 in buf[i_b]
    const int i_c = pick({i_add, i_b}), i_d = pick({i_add, i_b, i_c});
    RN_TRY(conv_bn_bwd(b.c1, buf[i_b], true, buf[i_c], nullptr, buf[i_d], buf[i_add]));   // -> d x = main branch + identity branch
    const int i_out = i_d;
    dxi = i_out;
    dsum("block dx (sum)", buf[dxi], Min * b.c1.Cp);
  }
  // ---- stem -------------------------------------------------------------------------------------------------------------------
  {
    const int R = m->cfg.image_resolution, H = R / 2;
    const int i_p = pick({dxi});
    RN_TRY(rn_avgpool2_bwd(buf[dxi], B, H, H, m->stem[2].Opad, buf[i_p], m->dtype, st));
    const int i_a = pick({i_p}), i_b = pick({i_p, i_a});
    RN_TRY(conv_bn_bwd(m->stem[2], buf[i_p], true, buf[i_a], nullptr, buf[i_b], nullptr));
    const int i_c = pick({i_b}), i_d = pick({i_b, i_c});
    RN_TRY(conv_bn_bwd(m->stem[1], buf[i_b], true, buf[i_c], nullptr, buf[i_d], nullptr));
    const int i_e = pick({i_d});
    RN_TRY(conv_bn_bwd(m->stem[0], buf[i_d], true, buf[i_e], nullptr, nullptr, nullptr));
  }
  return EZ_OK;
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
      if (p.w != dev) { h->fresh = false; h->tfresh = false; }
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
// ---- training path ------------------------------------------------------------------------------------------------------------
size_t ezclip_rn_train_shadow_bytes(ezclip_rn_handle h) { return h ? rn_train_shadow_layout(h, nullptr) : 0; }
int ezclip_rn_set_train_shadow(ezclip_rn_handle h, void* shadow, size_t bytes) {
  EZ_REQUIRE(h && shadow && ((uintptr_t)shadow % 256) == 0, "ezclip_rn_set_train_shadow: null / unaligned buffer");
  EZ_REQUIRE(bytes >= rn_train_shadow_layout(h, nullptr), "ezclip_rn_set_train_shadow: buffer too small");
  rn_train_shadow_layout(h, (char*)shadow);
  h->tshadow = shadow; h->tshadow_bytes = bytes; h->tfresh = false;
  return EZ_OK;
}
int ezclip_rn_refresh_train_weights(ezclip_rn_handle h, void* stream) {
  EZ_REQUIRE(h && h->tshadow, "ezclip_rn_refresh_train_weights: no training shadow buffer (ezclip_rn_set_train_shadow)");
  for (auto& p : h->params) EZ_REQUIRE(p.w != nullptr, "ezclip_rn_refresh_train_weights: parameter %s is not bound", p.name.c_str());
  hipStream_t st = (hipStream_t)stream;
  int rc = EZ_OK;
  rn_for_each(h, [&](ezclip_rn::Conv& c) {
    if (rc != EZ_OK) return;
    const float* W = h->params[c.w].w;
    if (c.st != nullptr) {        // the convolution's own weight, packed as for inference but with no BatchNorm folded in
      if (h->dtype == EZCLIP_BF16)
        hipLaunchKernelGGL(rn_pack_conv_kernel<bf16_t>, dim3(c.Opad), dim3(256), 0, st, W, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, c.O, c.I, c.k, c.k, c.Opad, c.Cp, c.ldk, c.cfirst,
                           (bf16_t*)c.st, h->junk_bias);
      else
        hipLaunchKernelGGL(rn_pack_conv_kernel<float>, dim3(c.Opad), dim3(256), 0, st, W, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, c.O, c.I, c.k, c.k, c.Opad, c.Cp, c.ldk, c.cfirst,
                           (float*)c.st, h->junk_bias);
      rc = check_hip(hipGetLastError(), "rn_pack_conv_kernel (training)");
      if (rc != EZ_OK) return;
    }
    if (c.sd != nullptr) rc = rn_pack_conv_dgrad(W, c.O, c.I, c.k, rup(c.Opad, 64), c.Cp, c.sd, h->dtype, st);
  });
  if (rc == EZ_OK) h->tfresh = true;
  return rc;
}
int ezclip_rn_bind_grad(ezclip_rn_handle h, const char* name, float* dev) {
  EZ_REQUIRE(h && name, "ezclip_rn_bind_grad: null argument");
  for (auto& p : h->params)
    if (p.name == name) { p.g = dev; return EZ_OK; }
  set_error("ezclip_rn_bind_grad: unknown parameter %s", name);
  return EZ_ERR_INVALID;
}
size_t ezclip_rn_train_saved_bytes(ezclip_rn_handle h, int batch) { return (h && batch > 0) ? rn_train_saved_bytes(h, batch) : 0; }
size_t ezclip_rn_train_scratch_bytes(ezclip_rn_handle h, int batch) { return (h && batch > 0) ? rn_train_scratch_bytes(h, batch) : 0; }
int ezclip_rn_encode_image_train(ezclip_rn_handle h, const float* pixels, int batch, float* out, void* saved_ws, size_t saved_bytes,
                                 void* scratch, size_t scratch_bytes, void* stream) {
  EZ_REQUIRE(h && pixels && out && saved_ws && scratch && batch > 0, "ezclip_rn_encode_image_train: null / empty argument");
  // nn.BatchNorm2d.train() needs more than one value per channel: the smallest map of the tower is sp x sp (torch raises the same way)
  EZ_REQUIRE((int64_t)batch * h->sp * h->sp > 1, "ezclip_rn_encode_image_train: batch %d at a %d x %d final map: BatchNorm in training mode needs more "
             "than 1 value per channel", batch, h->sp, h->sp);
  // the weight gradient of c_proj is a TN product with N = output_dim: 16-byte rows (the inference path accepts a ragged output_dim)
  EZ_REQUIRE(h->cfg.output_dim % (16 / (int)h->esz) == 0, "ezclip_rn_encode_image_train: training needs output_dim %% %d == 0 (got %d)",
             16 / (int)h->esz, h->cfg.output_dim);
  EZ_REQUIRE(h->fresh && h->tfresh, "ezclip_rn_encode_image_train: weights not packed (ezclip_rn_refresh_weights + ezclip_rn_refresh_train_weights)");
  EZ_REQUIRE(((uintptr_t)saved_ws % 256) == 0 && saved_bytes >= rn_train_saved_bytes(h, batch) && ((uintptr_t)scratch % 256) == 0 &&
                 scratch_bytes >= rn_train_scratch_bytes(h, batch), "ezclip_rn_encode_image_train: workspace too small / unaligned");
  return h->dtype == EZCLIP_BF16 ? rn_train_forward<bf16_t>(h, pixels, batch, out, (char*)saved_ws, saved_bytes, (float*)scratch, (hipStream_t)stream)
                                 : rn_train_forward<float>(h, pixels, batch, out, (char*)saved_ws, saved_bytes, (float*)scratch, (hipStream_t)stream);
}
int ezclip_rn_backward(ezclip_rn_handle h, const float* features, const float* d_features, int batch, const void* saved_ws, size_t saved_bytes,
                       void* scratch, size_t scratch_bytes, void* stream) {
  EZ_REQUIRE(h && features && d_features && saved_ws && scratch && batch > 0, "ezclip_rn_backward: null argument");
  {
    auto it = h->filled.find(saved_ws);
    EZ_REQUIRE(it != h->filled.end(), "ezclip_rn_backward: no training forward of this handle filled the saved workspace %p", saved_ws);
    EZ_REQUIRE(it->second.B == batch, "ezclip_rn_backward: the saved workspace holds a forward over %d images, the gradient is for %d", it->second.B, batch);
    EZ_REQUIRE(saved_bytes >= rn_train_saved_bytes(h, batch), "ezclip_rn_backward: saved workspace too small");
  }
  rn_train_bind(h, batch, (char*)const_cast<void*>(saved_ws));        // every pointer from the base of THIS forward's workspace
  EZ_REQUIRE(((uintptr_t)scratch % 256) == 0 && scratch_bytes >= rn_train_scratch_bytes(h, batch), "ezclip_rn_backward: scratch too small / unaligned");
  for (auto& p : h->params) {
    const bool stat = p.name.size() > 12 && (p.name.rfind(".running_mean") == p.name.size() - 13 || p.name.rfind(".running_var") == p.name.size() - 12);
    EZ_REQUIRE(stat || p.g != nullptr, "ezclip_rn_backward: no gradient buffer bound for %s (ezclip_rn_bind_grad)", p.name.c_str());
  }
  return h->dtype == EZCLIP_BF16 ? rn_train_backward<bf16_t>(h, features, d_features, batch, (char*)scratch, scratch_bytes, (hipStream_t)stream)
                                 : rn_train_backward<float>(h, features, d_features, batch, (char*)scratch, scratch_bytes, (hipStream_t)stream);
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
