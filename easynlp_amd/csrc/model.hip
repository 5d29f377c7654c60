// Dual-encoder forward/backward orchestration (host code; enqueues HIP kernels).
//
// Reference call tree being replaced (all stock torch ops there):
//   CLIPApp.forward            easynlp/appzoo/clip/model.py:106-150
//   CHINESE_CLIP.forward       easynlp/modelzoo/models/clip/modeling_chineseclip.py:352-365
//   VisualTransformer.forward  modeling_chineseclip.py:236-253
//   ResidualAttentionBlock     modeling_chineseclip.py:184-205 (pre-LN, QuickGELU)
//   BertModel.forward          easynlp/modelzoo/models/bert/modeling_bert.py:792-920
//   BertLayer / BertSelfOutput / BertOutput   modeling_bert.py:257-268,320-346,363-429 (post-LN, erf-GELU)
//
// HBM layout: activations are [tokens, features] row-major in the compute dtype
// (batch-first; the reference's seq-first permute for the ViT is immaterial,
// attention is per sample).  With save_for_backward every layer keeps its own
// set of buffers in the caller's workspace (no recompute; ~32 bytes/token/feature
// per ViT layer in bf16 -- 60 GB at B=1024, far inside 288 GB of HBM3E).
#include "model.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>

namespace ezclip {

// ------------------------------------------------------------------ errors --
static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return EZ_OK;
  set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return EZ_ERR_HIP;
}

// EZ_LAUNCH_CHECK(): hipGetLastError() after every kernel launch of the library.  EZCLIP_SYNC_LAUNCHES=1 in the environment turns
// each of them into a device-wide synchronise as well, so that an asynchronous failure is reported by the launch that caused it
// (file:line in the message) instead of by some later call; =2 also prints the site to stderr BEFORE waiting -- a memory access
// fault aborts the process inside the runtime, and the last line printed names the kernel.  Debugging only (serialises everything).
int launch_check(const char* file, int line) {
  static const int mode = [] { const char* e = getenv("EZCLIP_SYNC_LAUNCHES"); return e ? atoi(e) : 0; }();
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && mode > 0) {
    if (mode > 1) { fprintf(stderr, "[ezclip] launch %s:%d\n", file, line); fflush(stderr); }
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) return EZ_OK;
  set_error("HIP error %d (%s) after the kernel launch at %s:%d", (int)e, hipGetErrorString(e), file, line);
  return EZ_ERR_HIP;
}

// ------------------------------------------------------------------ arena ---
namespace {

struct Arena {
  char* base;
  size_t off = 0;
  explicit Arena(void* b) : base(reinterpret_cast<char*>(b)) {}
  void* take(size_t bytes) {
    off = (off + 255) & ~size_t(255);
    void* p = base ? base + off : nullptr;
    off += bytes;
    return p;
  }
  float* takef(size_t n) { return reinterpret_cast<float*>(take(n * 4)); }
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

// ------------------------------------------------------------------ create --
static int add_param(ezclip_model* m, const std::string& name, std::vector<int64_t> shape) {
  ezclip_model::Param p;
  p.name = name;
  p.shape = shape;
  p.numel = 1;
  for (auto d : shape) p.numel *= d;
  m->params.push_back(p);
  m->index[name] = (int)m->params.size() - 1;
  return (int)m->params.size() - 1;
}

static ezclip_model::Weight make_weight(ezclip_model* m, int p, int N, int K, bool transposed_src = false) {
  ezclip_model::Weight w;
  w.p = p; w.N = N; w.K = K; w.transposed_src = transposed_src;
  const int kmult = 128 / dtype_size(m->dtype);
  w.ldk = round_up(K, kmult);
  w.ldn = N;
  return w;
}

int model_create(const ezclip_config* c, ezclip_model** out, int text_arch) {
  EZ_REQUIRE(c != nullptr && out != nullptr, "ezclip_create: null argument");
  EZ_REQUIRE(c->compute_dtype == EZCLIP_F32 || c->compute_dtype == EZCLIP_BF16, "ezclip_create: bad compute_dtype %d", c->compute_dtype);
  const int kmult = 128 / dtype_size(c->compute_dtype);
  const int W = c->vision_width, H = c->text_hidden_size, F = c->text_intermediate_size, E = c->embed_dim;
  // vision_layers == 0: a text-only handle (the image tower is not a ViT: ModifiedResNet runs behind its own handle, ezclip_rn_*)
  const bool vis = c->vision_layers > 0;
  EZ_REQUIRE(!vis || (W > 0 && W % 64 == 0 && W <= 1024), "vision_width %d must be a multiple of 64 (head_dim 64) and <= 1024", W);
  EZ_REQUIRE(c->text_num_attention_heads > 0 && H == 64 * c->text_num_attention_heads && H <= 1024,
             "text_hidden_size %d must be 64 * heads (head_dim 64) and <= 1024", H);
  EZ_REQUIRE(F % kmult == 0 && F % 4 == 0, "text_intermediate_size %d must be a multiple of %d", F, kmult);
  EZ_REQUIRE(E % 32 == 0 && E <= 1024, "embed_dim %d must be a multiple of 32 and <= 1024", E);
  EZ_REQUIRE(!vis || (c->vision_patch_size > 0 && c->image_resolution >= c->vision_patch_size), "bad patch/resolution");
  EZ_REQUIRE(c->vision_layers >= 0 && c->text_num_hidden_layers > 0, "layer counts must be positive");
  EZ_REQUIRE(c->vocab_size > 0 && c->text_max_position_embeddings > 0 && (c->text_type_vocab_size > 0 || text_arch == 1),
             "bad text table sizes");
  EZ_REQUIRE(text_arch == 0 || text_arch == 1, "text_arch %d: 0 = BERT / RoBERTa, 1 = CLIP text transformer", text_arch);
  EZ_REQUIRE(text_arch == 0 || F == 4 * H, "CLIP text transformer: intermediate size must be 4 * width");

  ezclip_model* m = new ezclip_model();
  m->cfg = *c;
  m->dtype = c->compute_dtype;
  m->theads = c->text_num_attention_heads;
  m->text_arch = text_arch;
  m->cls_p = m->pos_p = m->lnpre_w = m->lnpre_b = m->lnpost_w = m->lnpost_b = -1;
  const int P = vis ? c->vision_patch_size : 1;
  if (vis) {
  m->G = c->image_resolution / c->vision_patch_size;
  m->Lv = m->G * m->G + 1;
  m->Kpatch = 3 * c->vision_patch_size * c->vision_patch_size;
  m->Kpad = round_up(m->Kpatch, kmult);
  m->vheads = W / 64;

  m->cls_p = add_param(m, "visual.class_embedding", {W});
  m->pos_p = add_param(m, "visual.positional_embedding", {m->Lv, W});
  int vproj = add_param(m, "visual.proj", {W, E});
  int conv = add_param(m, "visual.conv1.weight", {W, 3, P, P});
  m->lnpre_w = add_param(m, "visual.ln_pre.weight", {W});
  m->lnpre_b = add_param(m, "visual.ln_pre.bias", {W});
  m->conv_w = make_weight(m, conv, W, m->Kpatch);
  m->vproj_w = make_weight(m, vproj, E, W, true);
  // optional (huggingface_clip: vision_projection is an nn.Linear with a bias, appzoo/clip/model.py:96-101)
  m->vproj_b = add_param(m, "visual.proj_bias", {E});
  for (int i = 0; i < c->vision_layers; ++i) {
    const std::string p = "visual.transformer.resblocks." + std::to_string(i) + ".";
    ezclip_model::VitLayer L;
    L.in_w = make_weight(m, add_param(m, p + "attn.in_proj_weight", {3 * W, W}), 3 * W, W);
    L.in_b = add_param(m, p + "attn.in_proj_bias", {3 * W});
    L.out_w = make_weight(m, add_param(m, p + "attn.out_proj.weight", {W, W}), W, W);
    L.out_b = add_param(m, p + "attn.out_proj.bias", {W});
    L.ln1_w = add_param(m, p + "ln_1.weight", {W});
    L.ln1_b = add_param(m, p + "ln_1.bias", {W});
    L.fc_w = make_weight(m, add_param(m, p + "mlp.c_fc.weight", {4 * W, W}), 4 * W, W);
    L.fc_b = add_param(m, p + "mlp.c_fc.bias", {4 * W});
    L.proj_w = make_weight(m, add_param(m, p + "mlp.c_proj.weight", {W, 4 * W}), W, 4 * W);
    L.proj_b = add_param(m, p + "mlp.c_proj.bias", {W});
    L.ln2_w = add_param(m, p + "ln_2.weight", {W});
    L.ln2_b = add_param(m, p + "ln_2.bias", {W});
    // bf16 inference: ln_1 / ln_2 are folded into the in_proj / c_fc products (GemmArgs::ln_stats)
    L.in_w.fold_g = L.ln1_w; L.in_w.fold_b = L.ln1_b; L.in_w.fold_bias = L.in_b;
    L.fc_w.fold_g = L.ln2_w; L.fc_w.fold_b = L.ln2_b; L.fc_w.fold_bias = L.fc_b;
    m->vit.push_back(L);
  }
  m->lnpost_w = add_param(m, "visual.ln_post.weight", {W});
  m->lnpost_b = add_param(m, "visual.ln_post.bias", {W});
  }   // vis
  if (text_arch == 1) {
    // OPEN_CLIP state_dict names (modeling_openclip.py:296-311)
    m->tok_p = add_param(m, "token_embedding.weight", {c->vocab_size, H});
    m->tpos2_p = add_param(m, "positional_embedding", {c->text_max_position_embeddings, H});
    for (int i = 0; i < c->text_num_hidden_layers; ++i) {
      const std::string p = "transformer.resblocks." + std::to_string(i) + ".";
      ezclip_model::VitLayer L;
      L.in_w = make_weight(m, add_param(m, p + "attn.in_proj_weight", {3 * H, H}), 3 * H, H);
      L.in_b = add_param(m, p + "attn.in_proj_bias", {3 * H});
      L.out_w = make_weight(m, add_param(m, p + "attn.out_proj.weight", {H, H}), H, H);
      L.out_b = add_param(m, p + "attn.out_proj.bias", {H});
      L.ln1_w = add_param(m, p + "ln_1.weight", {H});
      L.ln1_b = add_param(m, p + "ln_1.bias", {H});
      L.fc_w = make_weight(m, add_param(m, p + "mlp.c_fc.weight", {4 * H, H}), 4 * H, H);
      L.fc_b = add_param(m, p + "mlp.c_fc.bias", {4 * H});
      L.proj_w = make_weight(m, add_param(m, p + "mlp.c_proj.weight", {H, 4 * H}), H, 4 * H);
      L.proj_b = add_param(m, p + "mlp.c_proj.bias", {H});
      L.ln2_w = add_param(m, p + "ln_2.weight", {H});
      L.ln2_b = add_param(m, p + "ln_2.bias", {H});
      L.in_w.fold_g = L.ln1_w; L.in_w.fold_b = L.ln1_b; L.in_w.fold_bias = L.in_b;
      L.fc_w.fold_g = L.ln2_w; L.fc_w.fold_b = L.ln2_b; L.fc_w.fold_bias = L.fc_b;
      m->ttx.push_back(L);
    }
    m->lnf_w = add_param(m, "ln_final.weight", {H});
    m->lnf_b = add_param(m, "ln_final.bias", {H});
    m->tproj_w = make_weight(m, add_param(m, "text_projection", {H, E}), E, H, true);
  } else {
    m->word_p = add_param(m, "bert.embeddings.word_embeddings.weight", {c->vocab_size, H});
    m->tpos_p = add_param(m, "bert.embeddings.position_embeddings.weight", {c->text_max_position_embeddings, H});
    m->type_p = add_param(m, "bert.embeddings.token_type_embeddings.weight", {c->text_type_vocab_size, H});
    m->eln_w = add_param(m, "bert.embeddings.LayerNorm.weight", {H});
    m->eln_b = add_param(m, "bert.embeddings.LayerNorm.bias", {H});
    for (int i = 0; i < c->text_num_hidden_layers; ++i) {
      const std::string p = "bert.encoder.layer." + std::to_string(i) + ".";
      ezclip_model::BertLayer L;
      L.q_w = make_weight(m, add_param(m, p + "attention.self.query.weight", {H, H}), H, H);
      L.q_b = add_param(m, p + "attention.self.query.bias", {H});
      L.k_w = make_weight(m, add_param(m, p + "attention.self.key.weight", {H, H}), H, H);
      L.k_b = add_param(m, p + "attention.self.key.bias", {H});
      L.v_w = make_weight(m, add_param(m, p + "attention.self.value.weight", {H, H}), H, H);
      L.v_b = add_param(m, p + "attention.self.value.bias", {H});
      L.o_w = make_weight(m, add_param(m, p + "attention.output.dense.weight", {H, H}), H, H);
      L.o_b = add_param(m, p + "attention.output.dense.bias", {H});
      L.ln1_w = add_param(m, p + "attention.output.LayerNorm.weight", {H});
      L.ln1_b = add_param(m, p + "attention.output.LayerNorm.bias", {H});
      L.i_w = make_weight(m, add_param(m, p + "intermediate.dense.weight", {F, H}), F, H);
      L.i_b = add_param(m, p + "intermediate.dense.bias", {F});
      L.d_w = make_weight(m, add_param(m, p + "output.dense.weight", {H, F}), H, F);
      L.d_b = add_param(m, p + "output.dense.bias", {H});
      L.ln2_w = add_param(m, p + "output.LayerNorm.weight", {H});
      L.ln2_b = add_param(m, p + "output.LayerNorm.bias", {H});
      if (m->dtype == EZCLIP_BF16) L.q_w.s_external = L.k_w.s_external = L.v_w.s_external = true;
      m->bert.push_back(L);
    }
    // chinese_clip: computed by the reference but unused (kept for the checkpoint contract).  huggingface_clip: the text
    // feature IS the pooled output (text_outputs[1], appzoo/clip/model.py:134) -- opt_text_pooler
    m->pool_w = make_weight(m, add_param(m, "bert.pooler.dense.weight", {H, H}), H, H);
    m->pool_b = add_param(m, "bert.pooler.dense.bias", {H});
    m->tproj_w = make_weight(m, add_param(m, "text_projection", {H, E}), E, H, true);
    m->tproj_b = add_param(m, "text_projection_bias", {E});
  }
  m->logit_scale_p = add_param(m, "logit_scale", {});
  *out = m;
  return EZ_OK;
}

// -------------------------------------------------------- weight shadows ---
static void for_each_weight(ezclip_model* m, const std::function<void(ezclip_model::Weight&)>& f) {
  if (m->conv_w.p >= 0) f(m->conv_w);          // (text-only handles have no image tower)
  if (m->vproj_w.p >= 0) f(m->vproj_w);
  f(m->tproj_w);
  if (m->text_arch == 0) f(m->pool_w);
  for (auto& L : m->vit) { f(L.in_w); f(L.out_w); f(L.fc_w); f(L.proj_w); }
  for (auto& L : m->bert) { f(L.q_w); f(L.k_w); f(L.v_w); f(L.o_w); f(L.i_w); f(L.d_w); }
  for (auto& L : m->ttx) { f(L.in_w); f(L.out_w); f(L.fc_w); f(L.proj_w); }
}

static bool needs_pack(const ezclip_model* m, const ezclip_model::Weight& w) {
  return m->dtype != EZCLIP_F32 || w.ldk != w.K || w.transposed_src;
}

size_t model_shadow_layout(ezclip_model* m, char* base, bool with_backward) {
  Arena a(base);
  const size_t esz = dtype_size(m->dtype);
  for (auto& L : m->bert) {
    if (!L.q_w.s_external) continue;
    const size_t one = (size_t)L.q_w.N * L.q_w.ldk * esz;         // (a multiple of 256 bytes: N and ldk are multiples of 64)
    char* blk = static_cast<char*>(a.take(3 * one));
    float* fb = a.takef(3 * (size_t)L.q_w.N);
    if (base) { L.qkv_s = blk; L.qkv_bias = fb; L.q_w.s = blk; L.k_w.s = blk + one; L.v_w.s = blk + 2 * one; }
  }
  for_each_weight(m, [&](ezclip_model::Weight& w) {
    void* s = w.s_external ? w.s : (needs_pack(m, w) ? a.take((size_t)w.N * w.ldk * esz) : nullptr);
    void* st = nullptr;
    // conv1 needs no input gradient (pixels are data)
    if (with_backward && &w != &m->conv_w) st = a.take((size_t)w.K * w.ldn * esz);
    void* sf = nullptr; float *c1 = nullptr, *c2 = nullptr;
    if (w.fold_g >= 0 && m->dtype == EZCLIP_BF16) {
      sf = a.take((size_t)w.N * w.ldk * esz);
      c1 = a.takef(w.N);
      c2 = a.takef(w.N);
    }
    if (base) { w.s = s; w.st = st; w.sf = sf; w.c1 = c1; w.c2 = c2; }
  });
  // device table of the re-packing jobs: one per weight plus the three gathered biases of every BERT layer
  size_t nweights = 0;
  for_each_weight(m, [&](ezclip_model::Weight&) { ++nweights; });
  void* jobs = a.take((nweights + 3 * m->bert.size() + 1) * sizeof(CastJob));
  if (base) { m->cast_jobs_dev = jobs; m->cast_jobs_host.clear(); }
  a.take(0);
  return a.off + 256;
}

static bool is_optional_param(const std::string& n) {
  return n.find("pooler") != std::string::npos || n == "visual.proj_bias" || n == "text_projection_bias";
}

int model_refresh_weights(ezclip_model* m, hipStream_t stream) {
  for (auto& p : m->params)
    EZ_REQUIRE(p.w != nullptr || is_optional_param(p.name), "parameter %s is not bound", p.name.c_str());
  int rc = EZ_OK;
  // One launch for every packed copy (CastJob, kernels.h).  The table only changes when a tensor moved; the folded-LayerNorm
  // copies (bf16 inference path only) are brought up to date by the first product that uses them.
  std::vector<CastJob> jobs;
  for_each_weight(m, [&](ezclip_model::Weight& w) {
    if (rc != EZ_OK) return;
    const float* src = m->P(w.p);
    if (src == nullptr) return;      // unbound optional weight (pooler): never used
    w.sf_fresh = false;
    CastJob j;
    j.src = src;
    // master [N, K]: straight copy = s [N, ldk], transposed = st [K, ldn];  master [K, N]: straight = st, transposed = s
    j.R = w.transposed_src ? w.K : w.N;
    j.C = w.transposed_src ? w.N : w.K;
    void* s_dst = nullptr;
    if (!needs_pack(m, w)) { w.s = const_cast<float*>(src); }
    else {
      if (w.s == nullptr) { set_error("weight shadow not set (call ezclip_set_shadow first)"); rc = EZ_ERR_STATE; return; }
      s_dst = w.s;
    }
    if (w.transposed_src) { j.dst_t = s_dst; j.ld_t = w.ldk; j.dst_s = w.st; j.ld_s = w.ldn; }
    else { j.dst_s = s_dst; j.ld_s = w.ldk; j.dst_t = w.st; j.ld_t = w.ldn; }
    if (j.dst_s != nullptr || j.dst_t != nullptr) jobs.push_back(j);
  });
  if (rc != EZ_OK) return rc;
  for (auto& L : m->bert) {
    if (L.qkv_bias == nullptr) continue;
    const int H = L.q_w.N;
    const int bp[3] = {L.q_b, L.k_b, L.v_b};
    for (int i = 0; i < 3; ++i) {
      CastJob j;
      j.kind = 1; j.src = m->P(bp[i]); j.dst_s = L.qkv_bias + (size_t)i * H; j.C = H; j.R = 1;
      jobs.push_back(j);
    }
  }
  if (!jobs.empty()) {
    EZ_REQUIRE(m->cast_jobs_dev != nullptr, "weight shadow not set (call ezclip_set_shadow first)");
    const int total = cast_jobs_finalize(jobs.data(), (int)jobs.size());
    const size_t bytes = jobs.size() * sizeof(CastJob);
    if (m->cast_jobs_host.size() != bytes || std::memcmp(m->cast_jobs_host.data(), jobs.data(), bytes) != 0) {
      m->cast_jobs_host.assign(reinterpret_cast<const unsigned char*>(jobs.data()), reinterpret_cast<const unsigned char*>(jobs.data()) + bytes);
      rc = check_hip(hipMemcpyAsync(m->cast_jobs_dev, m->cast_jobs_host.data(), bytes, hipMemcpyHostToDevice, stream),
                     "hipMemcpyAsync(cast jobs)");
      // (pageable source: wait for the copy before the host mirror can change again -- only when a tensor moved)
      if (rc == EZ_OK) rc = check_hip(hipStreamSynchronize(stream), "hipStreamSynchronize(cast jobs)");
      if (rc != EZ_OK) { m->cast_jobs_host.clear(); return rc; }
      m->cast_tiles = total;
    }
    rc = cast_jobs_run(static_cast<const CastJob*>(m->cast_jobs_dev), (int)jobs.size(), m->cast_tiles, m->dtype, stream);
    if (rc != EZ_OK) return rc;
  }
  m->weights_fresh = true;
  return EZ_OK;
}

// --------------------------------------------------------------- helpers ---
static GemmArgs linear_args(const ezclip_model* m, const void* A, int64_t lda, const ezclip_model::Weight& w, int bias_p,
                            void* C, int64_t ldc, int M, int act, const void* R, int64_t ldr, void* C2, bool out_f32) {
  GemmArgs g;
  g.A = A; g.lda = lda;
  g.B = w.s; g.ldb = w.ldk;
  g.C = C; g.ldc = ldc; g.C2 = C2;
  g.bias = bias_p >= 0 ? m->P(bias_p) : nullptr;
  g.R = R; g.ldr = ldr;
  g.M = M; g.N = w.N; g.K = w.ldk;
  g.act = act;
  g.out_f32 = (out_f32 && m->dtype == EZCLIP_BF16) ? 1 : 0;
  return g;
}

// rowstat_part: also leave the per-row (sum, sum of squares) partials of the output behind (GemmArgs::rowstat_part;
// the caller checked can_emit_rowstats)
static int linear(const ezclip_model* m, const void* A, int64_t lda, const ezclip_model::Weight& w, int bias_p,
                  void* C, int64_t ldc, int M, int act, const void* R, int64_t ldr, void* C2, bool out_f32,
                  hipStream_t stream, float* rowstat_part = nullptr) {
  GemmArgs g = linear_args(m, A, lda, w, bias_p, C, ldc, M, act, R, ldr, C2, out_f32);
  g.rowstat_part = rowstat_part;
  return gemm_nt(g, m->dtype, stream);
}

#define EZ_TRY(expr)                 \
  do {                               \
    int _rc = (expr);                \
    if (_rc != EZ_OK) return _rc;    \
  } while (0)

// the same against a raw packed [N, ldb] weight block and bias pointer (fused BERT q | k | v projections)
static int linear_ptr(const ezclip_model* m, const void* A, int64_t lda, const void* Bw, int64_t ldb, int N, const float* bias,
                      void* C, int64_t ldc, int M, hipStream_t stream) {
  GemmArgs g;
  g.A = A; g.lda = lda; g.B = Bw; g.ldb = ldb; g.C = C; g.ldc = ldc; g.bias = bias;
  g.M = M; g.N = N; g.K = (int)ldb;
  return gemm_nt(g, m->dtype, stream);
}

static bool g_fuse_bert_qkv = true;
void set_fuse_bert_qkv(int on) { g_fuse_bert_qkv = on != 0; }

// BertSelfAttention's query / key / value Linear (modeling_bert.py:172-200) into the packed [M, 3H] buffer: one product
// against the stacked weights where the library holds them back to back (bf16), three otherwise.  first = 1: key and value only.
static int bert_qkv_proj(const ezclip_model* m, const ezclip_model::BertLayer& Lw, const void* x, int64_t ldx, char* qkv, int M,
                         int first, hipStream_t stream) {
  const int H = m->cfg.text_hidden_size;
  const size_t esz = dtype_size(m->dtype);
  if (g_fuse_bert_qkv && Lw.qkv_s != nullptr) {
    const char* w = static_cast<const char*>(Lw.qkv_s) + (size_t)first * H * Lw.q_w.ldk * esz;
    return linear_ptr(m, x, ldx, w, Lw.q_w.ldk, (3 - first) * H, Lw.qkv_bias + (size_t)first * H, qkv + (size_t)first * H * esz,
                      3 * H, M, stream);
  }
  if (first == 0) EZ_TRY(linear(m, x, ldx, Lw.q_w, Lw.q_b, qkv, 3 * H, M, ACT_NONE, nullptr, 0, nullptr, false, stream));
  EZ_TRY(linear(m, x, ldx, Lw.k_w, Lw.k_b, qkv + H * esz, 3 * H, M, ACT_NONE, nullptr, 0, nullptr, false, stream));
  return linear(m, x, ldx, Lw.v_w, Lw.v_b, qkv + 2 * H * esz, 3 * H, M, ACT_NONE, nullptr, 0, nullptr, false, stream);
}

// C = act(LayerNorm(X) . W^T + bias) with the LayerNorm folded into the product (bf16 inference): X is read raw.
// stats_ready: `stats` already holds (rstd, -mean rstd) of X's rows (left behind by the GEMM that produced X).
static int linear_folded_ln(const ezclip_model* m, const void* X, int64_t ldx, const ezclip_model::Weight& w, float eps,
                            float* stats, void* C, int64_t ldc, int M, int act, hipStream_t stream,
                            bool stats_ready = false) {
  if (!w.sf_fresh) {      // first use since the master weights were re-packed (ezclip_refresh_weights): W o g, c1, c2
    int rc = fold_ln_weight(m->P(w.p), m->P(w.fold_g), m->P(w.fold_b), w.fold_bias >= 0 ? m->P(w.fold_bias) : nullptr, w.N, w.K,
                            w.sf, w.ldk, w.c1, w.c2, m->dtype, stream);
    if (rc != EZ_OK) return rc;
    w.sf_fresh = true;
  }
  if (!stats_ready) {
    int rc = layernorm_row_stats(X, ldx, eps, M, w.K, m->dtype, stats, stream);
    if (rc != EZ_OK) return rc;
  }
  GemmArgs g;
  g.A = X; g.lda = ldx;
  g.B = w.sf; g.ldb = w.ldk;
  g.C = C; g.ldc = ldc;
  g.M = M; g.N = w.N; g.K = w.ldk;
  g.act = act;
  g.ln_stats = stats; g.ln_c1 = w.c1; g.ln_c2 = w.c2;
  return gemm_nt(g, m->dtype, stream);
}

// 0: separate LayerNorm kernels; 1: folded, row statistics produced by the epilogue of the GEMM that writes the
// residual stream; 2: folded, statistics by a separate pass over the stream (ezclip_debug_set(2, v))
static int g_fold_ln = 1;
void set_fold_layernorm(int mode) { g_fold_ln = mode; }

static bool can_fold_ln(const ezclip_model* m, const ezclip_model::Weight& w, int M) {
  if (g_fold_ln == 0 || m->dtype != EZCLIP_BF16 || w.sf == nullptr) return false;
  GemmArgs g;   // the shape constraints of the 8-phase kernel
  g.M = M; g.N = w.N; g.K = w.ldk; g.lda = w.ldk; g.ldb = w.ldk; g.ldc = w.N;
  g.A = w.sf; g.B = w.sf; g.C = w.sf; g.ln_stats = w.c1; g.ln_c1 = w.c1; g.ln_c2 = w.c2;
  return gemm_nt_uses_8p(g, m->dtype);
}

// can the residual GEMM  C = A W^T + b + R  leave row-stat partials of C behind?
static bool can_emit_rowstats(const ezclip_model* m, const void* A, int64_t lda, const ezclip_model::Weight& w, int bias_p,
                              void* C, int M, const void* R, float* part) {
  if (g_fold_ln != 1 || part == nullptr) return false;
  GemmArgs g = linear_args(m, A, lda, w, bias_p, C, w.N, M, ACT_NONE, R, w.N, nullptr, false);
  g.rowstat_part = part;
  return gemm_nt_uses_8p(g, m->dtype);
}

// ------------------------------------------------------- image workspace ---
namespace {

struct VitBufs {
  void *x_in, *ln1, *qkv, *ctx, *x_mid, *ln2, *u, *h, *x_out;
  float *m1, *r1, *m2, *r2, *lse;
  float* stat;   // inference only: (rstd, -mean rstd) per row for the folded-LayerNorm products
  float* part;   // inference only: per-row (sum, sum sq) partials left by the residual GEMMs (GemmArgs::rowstat_part)
};
struct ImgWS {
  void *patches, *pemb, *x0;
  float *m0, *r0;
  std::vector<VitBufs> layers;
  void* cls_ln;
  float *mpost, *rpost, *feat, *emb, *inv_norm;
  // backward scratch
  void *gx, *gx2, *gqkv, *gbig, *gtmp, *gfeatT, *gcls;
  float* gbpart;   // [B][3][W] per-sample partial sums of the qkv bias gradient (fused attention backward)
  float *gfeat;
  float* gconv;   // [W, Kpad] scratch for the patch-embedding weight gradient when K is padded (ViT-L/14: 588 -> 640)
};

// save: keep what the backward pass needs.  With a frozen tower (opt_vision_frozen: the huggingface_clip branch detaches the
// vision output) that is only the projection's input and the embedding scratch: the blocks run on the inference
// buffer set (one set for all layers, LayerNorms folded).
size_t layout_image(const ezclip_model* m, int B, bool save_arg, void* base, ImgWS* ws) {
  const bool save = save_arg && !m->opt_vision_frozen;
  Arena a(base);
  const size_t esz = dtype_size(m->dtype);
  const int W = m->cfg.vision_width, E = m->cfg.embed_dim;
  const size_t Mp = (size_t)B * (m->Lv - 1), M = (size_t)B * m->Lv;
  ImgWS w;
  w.patches = a.take(Mp * m->Kpad * esz);
  w.pemb = a.take(Mp * W * esz);
  w.x0 = save ? a.take(M * W * esz) : nullptr;
  w.m0 = save ? a.takef(M) : nullptr;
  w.r0 = save ? a.takef(M) : nullptr;
  const int nl = m->cfg.vision_layers;
  w.layers.resize(nl);
  if (!save) {
    VitBufs b;
    b.x_in = b.x_mid = b.x_out = a.take(M * W * esz);
    b.ln1 = b.ln2 = a.take(M * W * esz);
    b.qkv = a.take(M * 3 * W * esz);
    b.ctx = a.take(M * W * esz);
    b.u = nullptr;
    b.h = a.take(M * 4 * W * esz);
    b.stat = a.takef(2 * M);
    b.part = a.takef(2 * M * (size_t)(W / 64));
    b.m1 = b.r1 = b.m2 = b.r2 = b.lse = nullptr;
    for (int i = 0; i < nl; ++i) w.layers[i] = b;
  } else {
    void* x = a.take(M * W * esz);
    for (int i = 0; i < nl; ++i) {
      VitBufs& b = w.layers[i];
      b.x_in = x;
      b.ln1 = a.take(M * W * esz);
      b.qkv = a.take(M * 3 * W * esz);
      b.ctx = a.take(M * W * esz);
      b.x_mid = a.take(M * W * esz);
      b.ln2 = a.take(M * W * esz);
      b.u = a.take(M * 4 * W * esz);
      b.h = a.take(M * 4 * W * esz);
      b.x_out = a.take(M * W * esz);
      b.m1 = a.takef(M); b.r1 = a.takef(M); b.m2 = a.takef(M); b.r2 = a.takef(M);
      b.lse = a.takef((size_t)B * m->vheads * m->Lv);
      b.stat = nullptr; b.part = nullptr;
      x = b.x_out;
    }
  }
  w.cls_ln = a.take((size_t)B * W * esz);
  w.mpost = a.takef(B); w.rpost = a.takef(B);
  w.feat = a.takef((size_t)B * E);
  w.emb = a.takef((size_t)B * E);
  w.inv_norm = a.takef(B);
  if (save) {
    w.gx = a.take(M * W * esz);
    w.gx2 = a.take(M * W * esz);
    w.gtmp = a.take(M * W * esz);
    w.gqkv = a.take(M * 3 * W * esz);
    w.gbpart = a.takef((size_t)B * 3 * W);
    w.gbig = a.take(M * 4 * W * esz);
    w.gfeat = a.takef((size_t)B * E);
    w.gfeatT = a.take((size_t)B * E * esz);
    w.gcls = a.take((size_t)B * W * esz);
    w.gconv = m->Kpad != m->Kpatch ? a.takef((size_t)W * m->Kpad) : nullptr;
  } else {
    w.gx = w.gx2 = w.gtmp = w.gqkv = w.gbig = w.gfeatT = w.gcls = nullptr; w.gfeat = nullptr; w.gconv = nullptr; w.gbpart = nullptr;
    if (save_arg) {      // frozen tower: the projection's gradients only
      w.gfeat = a.takef((size_t)B * E);
      w.gfeatT = a.take((size_t)B * E * esz);
    }
  }
  if (ws) *ws = w;
  return a.off + 256;
}

struct BertBufs {
  void *x_in, *qkv, *ctx, *y, *a, *u, *hh, *z, *x_out;
  float *m1, *r1, *m2, *r2, *lse;
  uint32_t* keep;      // keep bits of the attention-probability dropout (AttnArgs::keep_bits), [rows][heads][ceil(L / 32)]
};
struct TxtWS {
  void* x0;
  float *m0, *r0, *key_bias;
  std::vector<BertBufs> layers;
  float *feat, *emb, *inv_norm;
  void *pool, *pool_u, *gpool;     // pooled output tanh(u), its pre-activation u, d u   [B, H] (opt_text_pooler)
  void *cls_rows, *gcls;           // packed batches whose last layer ran on all rows: its CLS rows [B, H] and their gradient
  void *gx, *gx2, *gx3, *gtmp, *gqkv, *gbig, *gfeatT;
  float *gfeat, *gbpart;
};

size_t layout_text(const ezclip_model* m, int B, int L, bool save, void* base, TxtWS* ws) {
  Arena a(base);
  const size_t esz = dtype_size(m->dtype);
  const int H = m->cfg.text_hidden_size, F = m->cfg.text_intermediate_size, E = m->cfg.embed_dim;
  const size_t M = (size_t)B * L;
  TxtWS w;
  w.x0 = save ? a.take(M * H * esz) : nullptr;
  w.m0 = save ? a.takef(M) : nullptr;
  w.r0 = save ? a.takef(M) : nullptr;
  w.key_bias = a.takef(M);
  const int nl = m->cfg.text_num_hidden_layers;
  w.layers.resize(nl);
  if (!save) {
    BertBufs b;
    b.x_in = b.x_out = a.take(M * H * esz);
    b.qkv = a.take(M * 3 * H * esz);
    b.ctx = a.take(M * H * esz);
    b.y = b.a = a.take(M * H * esz);
    b.u = nullptr;
    b.hh = a.take(M * F * esz);
    b.z = b.x_in;
    b.m1 = b.r1 = b.m2 = b.r2 = b.lse = nullptr;
    b.keep = nullptr;
    for (int i = 0; i < nl; ++i) w.layers[i] = b;
  } else {
    void* x = a.take(M * H * esz);
    for (int i = 0; i < nl; ++i) {
      BertBufs& b = w.layers[i];
      b.x_in = x;
      b.keep = static_cast<uint32_t*>(a.take(M * (size_t)m->theads * ((L + 31) / 32) * 4));
      b.qkv = a.take(M * 3 * H * esz);
      b.ctx = a.take(M * H * esz);
      b.y = a.take(M * H * esz);
      b.a = a.take(M * H * esz);
      b.u = a.take(M * F * esz);
      b.hh = a.take(M * F * esz);
      b.z = a.take(M * H * esz);
      b.x_out = a.take(M * H * esz);
      b.m1 = a.takef(M); b.r1 = a.takef(M); b.m2 = a.takef(M); b.r2 = a.takef(M);
      b.lse = a.takef((size_t)B * m->theads * L);
      x = b.x_out;
    }
  }
  w.feat = a.takef((size_t)B * E);
  w.emb = a.takef((size_t)B * E);
  w.inv_norm = a.takef(B);
  w.pool = a.take((size_t)B * H * esz);
  w.pool_u = a.take((size_t)B * H * esz);
  w.gpool = a.take((size_t)B * H * esz);
  w.cls_rows = a.take((size_t)B * H * esz);
  w.gcls = a.take((size_t)B * H * esz);
  if (save) {
    w.gx = a.take(M * H * esz);
    w.gx2 = a.take(M * H * esz);
    w.gx3 = a.take(M * H * esz);     // dropout-masked copy of a residual-branch gradient (train mode with p > 0)
    w.gtmp = a.take(M * H * esz);
    w.gqkv = a.take(M * 3 * H * esz);
    w.gbpart = a.takef((size_t)B * 3 * H);
    w.gbig = a.take(M * F * esz);
    w.gfeat = a.takef((size_t)B * E);
    w.gfeatT = a.take((size_t)B * E * esz);
  } else {
    w.gx = w.gx2 = w.gx3 = w.gtmp = w.gqkv = w.gbig = w.gfeatT = nullptr; w.gfeat = nullptr; w.gbpart = nullptr;
  }
  if (ws) *ws = w;
  return a.off + 256;
}

}  // namespace

size_t image_workspace_bytes(const ezclip_model* m, int B, bool save) {
  return m->vit.empty() ? 0 : layout_image(m, B, save, nullptr, nullptr);
}


// One pre-LN residual attention block (ResidualAttentionBlock, modeling_chineseclip.py:184-205 == modeling_openclip.py's
// text / vision blocks): x_mid = x_in + out_proj(attn(in_proj(ln_1(x_in)))), x_out = x_mid + c_proj(QuickGELU(c_fc(ln_2(x_mid)))).
// Shared by the ViT tower and the CLIP text transformer (causal = 1).  stats_ready / next_stat: the folded-LayerNorm
// statistics hand-over between consecutive blocks of the bf16 inference path.
struct BlockDims { int M, W, B, L, heads, causal; };

static int resblock_forward(ezclip_model* m, const ezclip_model::VitLayer& Lw, const VitBufs& b, const BlockDims& d, bool save,
                            bool& stats_ready, float* next_stat, hipStream_t stream) {
  const int M = d.M, W = d.W, dt = m->dtype;
  const float eps = m->block_ln_eps;  // nn.LayerNorm default 1e-5 (modeling_chineseclip.py:170); EZCLIP_OPT_BLOCK_LN_EPS
  {
    // x = x + attn(ln_1(x))                                          :203
    const bool fold = !save && can_fold_ln(m, Lw.in_w, M) && can_fold_ln(m, Lw.fc_w, M);
    if (fold) {
      EZ_TRY(linear_folded_ln(m, b.x_in, W, Lw.in_w, eps, b.stat, b.qkv, 3 * W, M, ACT_NONE, stream, stats_ready));
    } else {
      EZ_TRY(layernorm_fwd(b.x_in, W, b.ln1, W, m->P(Lw.ln1_w), m->P(Lw.ln1_b), eps, M, W, dt, b.m1, b.r1, stream));
      EZ_TRY(linear(m, b.ln1, W, Lw.in_w, Lw.in_b, b.qkv, 3 * W, M, ACT_NONE, nullptr, 0, nullptr, false, stream));
    }
    stats_ready = false;
    AttnArgs at;
    at.q = b.qkv;
    at.k = (const char*)b.qkv + (size_t)W * dtype_size(dt);
    at.v = (const char*)b.qkv + (size_t)2 * W * dtype_size(dt);
    at.row_stride = 3 * W;
    at.ctx = b.ctx; at.ctx_stride = W;
    at.key_bias = nullptr; at.lse = b.lse;
    at.B = d.B; at.L = d.L; at.H = d.heads; at.scale = 0.125f;
    at.causal = d.causal;
    EZ_TRY(attention_fwd(at, dt, stream));
    // (when the next product folds its LayerNorm, the residual GEMM's epilogue leaves that LayerNorm's row sums behind)
    float* part = fold && can_emit_rowstats(m, b.ctx, W, Lw.out_w, Lw.out_b, b.x_mid, M, b.x_in, b.part) ? b.part : nullptr;
    EZ_TRY(linear(m, b.ctx, W, Lw.out_w, Lw.out_b, b.x_mid, W, M, ACT_NONE, b.x_in, W, nullptr, false, stream, part));
    if (part) EZ_TRY(layernorm_stats_finalize(part, W / 64, W, eps, M, b.stat, stream));
    // x = x + c_proj(QuickGELU(c_fc(ln_2(x))))                      :204
    if (fold) {
      EZ_TRY(linear_folded_ln(m, b.x_mid, W, Lw.fc_w, eps, b.stat, b.h, 4 * W, M, ACT_QUICKGELU, stream, part != nullptr));
    } else {
      EZ_TRY(layernorm_fwd(b.x_mid, W, b.ln2, W, m->P(Lw.ln2_w), m->P(Lw.ln2_b), eps, M, W, dt, b.m2, b.r2, stream));
      EZ_TRY(linear(m, b.ln2, W, Lw.fc_w, Lw.fc_b, b.h, 4 * W, M, ACT_QUICKGELU, nullptr, 0, b.u, false, stream));
    }
    part = fold && next_stat != nullptr &&
                   can_emit_rowstats(m, b.h, 4 * W, Lw.proj_w, Lw.proj_b, b.x_out, M, b.x_mid, b.part) ? b.part : nullptr;
    EZ_TRY(linear(m, b.h, 4 * W, Lw.proj_w, Lw.proj_b, b.x_out, W, M, ACT_NONE, b.x_mid, W, nullptr, false, stream, part));
    if (part) {    // statistics of the next block's ln_1 (inference: every block shares one buffer set)
      EZ_TRY(layernorm_stats_finalize(part, W / 64, W, eps, M, next_stat, stream));
      stats_ready = true;
    }
  }
  return EZ_OK;
}

// The LAST block of a tower on the inference path: only x[:, 0] is read afterwards (ln_post(x[:, 0, :]) @ proj,
// modeling_chineseclip.py:248-251), so after the full-width qkv product everything runs on B rows: attention for the CLS
// query (attention_cls_fwd), out_proj + residual, ln_2, c_fc, QuickGELU, c_proj + residual.  Same arithmetic per row as
// resblock_forward (a GEMM row does not depend on M); 72 % of the block's FLOPs are not spent on rows nobody reads.
// scratch: >= 3 * B * W elements.  Output: x_out_cls = scratch + B * W elements, row stride W.
static bool g_cls_last = true;
void set_cls_last(int on) { g_cls_last = on != 0; }

static bool g_cls_q_only = true;      // ezclip_debug_set(8, 0): project the last block's queries for every token (A/B, cross-check)
void set_cls_q_only(int on) { g_cls_q_only = on != 0; }

static int resblock_forward_cls(ezclip_model* m, const ezclip_model::VitLayer& Lw, const VitBufs& b, const BlockDims& d,
                                bool stats_ready, void* scratch, hipStream_t stream) {
  const int M = d.M, W = d.W, B = d.B, dt = m->dtype;
  const size_t esz = dtype_size(dt);
  const float eps = m->block_ln_eps;
  char* x_mid = static_cast<char*>(scratch);
  char* x_out = x_mid + (size_t)B * W * esz;
  char* ln2 = x_out + (size_t)B * W * esz;
  const void* q_cls = b.qkv;                     // the queries of the CLS rows: inside the packed qkv buffer, or compact [B, W]
  int64_t q_stride = (int64_t)d.L * 3 * W;
  if (can_fold_ln(m, Lw.in_w, M) && g_cls_q_only && d.L >= 5) {      // (the scratch holds 5 B W elements)
    // (round 3) only the CLS rows query: keys | values for all tokens (N = 2W: rows W.. of the folded in_proj copy), the query
    // projection for the B CLS rows alone (LayerNorm of those rows + the plain packed copy) -- a third of this product's rows of
    // output are never read otherwise
    const ezclip_model::Weight& w = Lw.in_w;
    if (!w.sf_fresh) {
      EZ_TRY(fold_ln_weight(m->P(w.p), m->P(w.fold_g), m->P(w.fold_b), w.fold_bias >= 0 ? m->P(w.fold_bias) : nullptr, w.N, w.K,
                            w.sf, w.ldk, w.c1, w.c2, dt, stream));
      w.sf_fresh = true;
    }
    if (!stats_ready) EZ_TRY(layernorm_row_stats(b.x_in, W, eps, M, w.K, dt, b.stat, stream));
    GemmArgs g;
    g.A = b.x_in; g.lda = W;
    g.B = static_cast<const char*>(w.sf) + (size_t)W * w.ldk * esz; g.ldb = w.ldk;
    g.C = static_cast<char*>(b.qkv) + (size_t)W * esz; g.ldc = 3 * W;
    g.M = M; g.N = 2 * W; g.K = w.ldk;
    g.ln_stats = b.stat; g.ln_c1 = w.c1 + W; g.ln_c2 = w.c2 + W;
    EZ_TRY(gemm_nt(g, dt, stream));
    char* ln_cls = ln2 + (size_t)B * W * esz;
    char* qc = ln_cls + (size_t)B * W * esz;
    EZ_TRY(layernorm_fwd(b.x_in, (int64_t)d.L * W, ln_cls, W, m->P(Lw.ln1_w), m->P(Lw.ln1_b), eps, B, W, dt, nullptr, nullptr, stream));
    EZ_TRY(linear_ptr(m, ln_cls, W, w.s, w.ldk, W, m->P(Lw.in_b), qc, W, B, stream));
    q_cls = qc;
    q_stride = W;
  } else if (can_fold_ln(m, Lw.in_w, M)) {
    EZ_TRY(linear_folded_ln(m, b.x_in, W, Lw.in_w, eps, b.stat, b.qkv, 3 * W, M, ACT_NONE, stream, stats_ready));
  } else {
    EZ_TRY(layernorm_fwd(b.x_in, W, b.ln1, W, m->P(Lw.ln1_w), m->P(Lw.ln1_b), eps, M, W, dt, nullptr, nullptr, stream));
    EZ_TRY(linear(m, b.ln1, W, Lw.in_w, Lw.in_b, b.qkv, 3 * W, M, ACT_NONE, nullptr, 0, nullptr, false, stream));
  }
  AttnArgs at;
  at.k = (const char*)b.qkv + (size_t)W * esz;
  at.v = (const char*)b.qkv + (size_t)2 * W * esz;
  at.row_stride = 3 * W;
  at.B = B; at.L = d.L; at.H = d.heads; at.scale = 0.125f;
  EZ_TRY(attention_cls_fwd(at, q_cls, q_stride, b.ctx, W, dt, stream));                 // q of token 0
  EZ_TRY(linear(m, b.ctx, W, Lw.out_w, Lw.out_b, x_mid, W, B, ACT_NONE, b.x_in, (int64_t)d.L * W, nullptr, false, stream));
  if (can_fold_ln(m, Lw.fc_w, B)) {
    EZ_TRY(linear_folded_ln(m, x_mid, W, Lw.fc_w, eps, b.stat, b.h, 4 * W, B, ACT_QUICKGELU, stream));
  } else {
    EZ_TRY(layernorm_fwd(x_mid, W, ln2, W, m->P(Lw.ln2_w), m->P(Lw.ln2_b), eps, B, W, dt, nullptr, nullptr, stream));
    EZ_TRY(linear(m, ln2, W, Lw.fc_w, Lw.fc_b, b.h, 4 * W, B, ACT_QUICKGELU, nullptr, 0, nullptr, false, stream));
  }
  return linear(m, b.h, 4 * W, Lw.proj_w, Lw.proj_b, x_out, W, B, ACT_NONE, x_mid, W, nullptr, false, stream);
}

// Training variant of resblock_forward_cls: the same evaluation (full-width ln_1 + in_proj, everything after it on the CLS
// rows), every intermediate kept for the backward pass in the block's own buffers (the first B rows of each).
static bool g_cls_last_train = true;
void set_cls_last_train(int on) { g_cls_last_train = on != 0; }

static int resblock_forward_cls_save(ezclip_model* m, const ezclip_model::VitLayer& Lw, const VitBufs& b, const BlockDims& d,
                                     hipStream_t stream) {
  const int M = d.M, W = d.W, B = d.B, dt = m->dtype;
  const size_t esz = dtype_size(dt);
  const float eps = m->block_ln_eps;
  EZ_TRY(layernorm_fwd(b.x_in, W, b.ln1, W, m->P(Lw.ln1_w), m->P(Lw.ln1_b), eps, M, W, dt, b.m1, b.r1, stream));
  EZ_TRY(linear(m, b.ln1, W, Lw.in_w, Lw.in_b, b.qkv, 3 * W, M, ACT_NONE, nullptr, 0, nullptr, false, stream));
  AttnArgs at;
  at.k = (const char*)b.qkv + (size_t)W * esz;
  at.v = (const char*)b.qkv + (size_t)2 * W * esz;
  at.row_stride = 3 * W;
  at.B = B; at.L = d.L; at.H = d.heads; at.scale = 0.125f;
  EZ_TRY(attention_cls_fwd(at, b.qkv, (int64_t)d.L * 3 * W, b.ctx, W, dt, stream));
  EZ_TRY(linear(m, b.ctx, W, Lw.out_w, Lw.out_b, b.x_mid, W, B, ACT_NONE, b.x_in, (int64_t)d.L * W, nullptr, false, stream));
  EZ_TRY(layernorm_fwd(b.x_mid, W, b.ln2, W, m->P(Lw.ln2_w), m->P(Lw.ln2_b), eps, B, W, dt, b.m2, b.r2, stream));
  EZ_TRY(linear(m, b.ln2, W, Lw.fc_w, Lw.fc_b, b.h, 4 * W, B, ACT_QUICKGELU, nullptr, 0, b.u, false, stream));
  return linear(m, b.h, 4 * W, Lw.proj_w, Lw.proj_b, b.x_out, W, B, ACT_NONE, b.x_mid, W, nullptr, false, stream);
}

// ------------------------------------------------------------ image fwd ----
int encode_image(ezclip_model* m, const float* pixels, int B, float* out, void* wsp, size_t ws_bytes, bool save,
                 hipStream_t stream) {
  EZ_REQUIRE(B > 0 && pixels && out && wsp, "encode_image: null/empty argument");
  EZ_REQUIRE(!m->vit.empty(), "encode_image: this handle has no image tower (created with vision_layers = 0)");
  EZ_REQUIRE(m->weights_fresh, "encode_image: call ezclip_refresh_weights after binding/updating parameters");
  EZ_REQUIRE(((uintptr_t)wsp % 256) == 0, "encode_image: workspace must be 256-byte aligned");
  ImgWS ws;
  const size_t need = layout_image(m, B, save, wsp, &ws);
  EZ_REQUIRE(ws_bytes >= need, "encode_image: workspace too small (%zu < %zu)", ws_bytes, need);
  const int W = m->cfg.vision_width, E = m->cfg.embed_dim, Lv = m->Lv;
  const int Mp = B * (Lv - 1), M = B * Lv;
  const int dt = m->dtype;
  const float eps = m->block_ln_eps;  // nn.LayerNorm default 1e-5 (modeling_chineseclip.py:170); EZCLIP_OPT_BLOCK_LN_EPS

  // conv1 (stride == kernel, no bias) = im2col + GEMM              :237-239
  EZ_TRY(im2col_patches(pixels, ws.patches, B, m->cfg.image_resolution, m->cfg.vision_patch_size, m->Kpad, dt, stream));
  EZ_TRY(linear(m, ws.patches, m->Kpad, m->conv_w, -1, ws.pemb, W, Mp, ACT_NONE, nullptr, 0, nullptr, false, stream));
  // cls token, + positional embedding, ln_pre                       :240-242
  EZ_TRY(vit_assemble_ln(ws.pemb, m->P(m->cls_p), m->P(m->pos_p), m->P(m->lnpre_w), m->P(m->lnpre_b), eps, ws.x0,
                         ws.layers[0].x_in, ws.m0, ws.r0, B, Lv, W, dt, stream));
  bool stats_ready = false;   // b.stat already holds the row statistics of this block's input
  const BlockDims bd{M, W, B, Lv, m->vheads, 0};
  const bool save_blocks = save && !m->opt_vision_frozen;     // (see layout_image)
  const int nl = m->cfg.vision_layers;
  const bool cls_infer = !save_blocks && g_cls_last && Lv >= 4;     // the last block only feeds x[:, 0]
  const bool cls_train = save_blocks && g_cls_last_train && Lv >= 4;
  const bool cls_last = cls_infer || cls_train;
  for (int i = 0; i < nl - (cls_last ? 1 : 0); ++i)
    EZ_TRY(resblock_forward(m, m->vit[i], ws.layers[i], bd, save_blocks, stats_ready,
                            i + 1 < nl ? ws.layers[i + 1].stat : nullptr, stream));
  // ln_post(x[:, 0, :]) @ proj                                        :248-251
  const void* xl = ws.layers[nl - 1].x_out;
  int64_t xl_stride = (int64_t)Lv * W;
  if (cls_infer) {
    const VitBufs& b = ws.layers[nl - 1];
    EZ_TRY(resblock_forward_cls(m, m->vit[nl - 1], b, bd, stats_ready, b.ln1, stream));     // (ln1: M * W >= 3 * B * W elements)
    xl = static_cast<const char*>(b.ln1) + (size_t)B * W * dtype_size(dt);
    xl_stride = W;
  } else if (cls_train) {
    EZ_TRY(resblock_forward_cls_save(m, m->vit[nl - 1], ws.layers[nl - 1], bd, stream));
    xl_stride = W;                                                   // x_out of the last block: [B, W] (CLS rows)
  }
  EZ_TRY(layernorm_fwd(xl, xl_stride, ws.cls_ln, W, m->P(m->lnpost_w), m->P(m->lnpost_b), eps, B, W, dt,
                       ws.mpost, ws.rpost, stream));
  EZ_TRY(linear(m, ws.cls_ln, W, m->vproj_w, m->vproj_b, ws.feat, E, B, ACT_NONE, nullptr, 0, nullptr, true, stream));
  // image_features / image_features.norm(dim=-1, keepdim=True)       :360
  EZ_TRY(l2_normalize_fwd(ws.feat, ws.emb, ws.inv_norm, B, E, stream));
  EZ_HIP(hipMemcpyAsync(out, ws.emb, (size_t)B * E * 4, hipMemcpyDeviceToDevice, stream));
  return EZ_OK;
}

// ---------------------------------------------- CLIP text transformer (open_clip) ----
namespace {

struct TxtClipWS {
  std::vector<VitBufs> layers;
  int* eot;                        // [B] argmax_t ids
  void *eot_rows, *eot_ln;         // [B, W] gathered EOT rows, ln_final of them
  float *mpost, *rpost, *feat, *emb, *inv_norm;
  void *gx, *gx2, *gtmp, *gqkv, *gbig, *gfeatT, *gcls, *geot;
  float *gfeat, *gbpart;
};

size_t layout_text_clip(const ezclip_model* m, int B, int L, bool save, void* base, TxtClipWS* ws) {
  Arena a(base);
  const size_t esz = dtype_size(m->dtype);
  const int W = m->cfg.text_hidden_size, E = m->cfg.embed_dim;
  const size_t M = (size_t)B * L;
  TxtClipWS w;
  const int nl = m->cfg.text_num_hidden_layers;
  w.layers.resize(nl);
  if (!save) {
    VitBufs b;
    b.x_in = b.x_mid = b.x_out = a.take(M * W * esz);
    b.ln1 = b.ln2 = a.take(M * W * esz);
    b.qkv = a.take(M * 3 * W * esz);
    b.ctx = a.take(M * W * esz);
    b.u = nullptr;
    b.h = a.take(M * 4 * W * esz);
    b.stat = a.takef(2 * M);
    b.part = a.takef(2 * M * (size_t)(W / 64));
    b.m1 = b.r1 = b.m2 = b.r2 = b.lse = nullptr;
    for (int i = 0; i < nl; ++i) w.layers[i] = b;
  } else {
    void* x = a.take(M * W * esz);
    for (int i = 0; i < nl; ++i) {
      VitBufs& b = w.layers[i];
      b.x_in = x;
      b.ln1 = a.take(M * W * esz);
      b.qkv = a.take(M * 3 * W * esz);
      b.ctx = a.take(M * W * esz);
      b.x_mid = a.take(M * W * esz);
      b.ln2 = a.take(M * W * esz);
      b.u = a.take(M * 4 * W * esz);
      b.h = a.take(M * 4 * W * esz);
      b.x_out = a.take(M * W * esz);
      b.m1 = a.takef(M); b.r1 = a.takef(M); b.m2 = a.takef(M); b.r2 = a.takef(M);
      b.lse = a.takef((size_t)B * m->theads * L);
      b.stat = nullptr; b.part = nullptr;
      x = b.x_out;
    }
  }
  w.eot = reinterpret_cast<int*>(a.takef(B));
  w.eot_rows = a.take((size_t)B * W * esz);
  w.eot_ln = a.take((size_t)B * W * esz);
  w.mpost = a.takef(B); w.rpost = a.takef(B);
  w.feat = a.takef((size_t)B * E);
  w.emb = a.takef((size_t)B * E);
  w.inv_norm = a.takef(B);
  if (save) {
    w.gx = a.take(M * W * esz);
    w.gx2 = a.take(M * W * esz);
    w.gtmp = a.take(M * W * esz);
    w.gqkv = a.take(M * 3 * W * esz);
    w.gbpart = a.takef((size_t)B * 3 * W);
    w.gbig = a.take(M * 4 * W * esz);
    w.gfeat = a.takef((size_t)B * E);
    w.gfeatT = a.take((size_t)B * E * esz);
    w.gcls = a.take((size_t)B * W * esz);
    w.geot = a.take((size_t)B * W * esz);
  } else {
    w.gx = w.gx2 = w.gtmp = w.gqkv = w.gbig = w.gfeatT = w.gcls = w.geot = nullptr;
    w.gfeat = nullptr; w.gbpart = nullptr;
  }
  if (ws) *ws = w;
  return a.off + 256;
}

// OPEN_CLIP.encode_text (modeling_openclip.py:354-368) + L2 normalise (:383)
int encode_text_clip(ezclip_model* m, const int64_t* ids, int B, int L, float* out, void* wsp, size_t ws_bytes, bool save,
                     hipStream_t stream) {
  TxtClipWS ws;
  const size_t need = layout_text_clip(m, B, L, save, wsp, &ws);
  EZ_REQUIRE(ws_bytes >= need, "encode_text: workspace too small (%zu < %zu)", ws_bytes, need);
  const int W = m->cfg.text_hidden_size, E = m->cfg.embed_dim, M = B * L, dt = m->dtype;
  const float eps = m->block_ln_eps;
  // x = token_embedding(text) + positional_embedding                :355-357
  EZ_TRY(clip_text_embed(ids, m->P(m->tok_p), m->P(m->tpos2_p), ws.layers[0].x_in, ws.eot, B, L, W, m->cfg.vocab_size,
                         m->text_eot_id, dt, stream));
  bool stats_ready = false;
  const BlockDims bd{M, W, B, L, m->theads, 1};                     // causal: build_attention_mask :343-349
  const int nl = m->cfg.text_num_hidden_layers;
  for (int i = 0; i < nl; ++i)
    EZ_TRY(resblock_forward(m, m->ttx[i], ws.layers[i], bd, save, stats_ready, i + 1 < nl ? ws.layers[i + 1].stat : nullptr, stream));
  // ln_final(x)[arange(B), text.argmax(-1)] @ text_projection       :361-366  (LayerNorm is per row: gather first)
  EZ_TRY(gather_rows(ws.layers[nl - 1].x_out, ws.eot, ws.eot_rows, B, L, W, 0, dt, stream));
  EZ_TRY(layernorm_fwd(ws.eot_rows, W, ws.eot_ln, W, m->P(m->lnf_w), m->P(m->lnf_b), eps, B, W, dt, ws.mpost, ws.rpost, stream));
  EZ_TRY(linear(m, ws.eot_ln, W, m->tproj_w, -1, ws.feat, E, B, ACT_NONE, nullptr, 0, nullptr, true, stream));
  EZ_TRY(l2_normalize_fwd(ws.feat, ws.emb, ws.inv_norm, B, E, stream));
  EZ_HIP(hipMemcpyAsync(out, ws.emb, (size_t)B * E * 4, hipMemcpyDeviceToDevice, stream));
  return EZ_OK;
}

}  // namespace

size_t text_workspace_bytes(const ezclip_model* m, int B, int L, bool save) {
  return m->text_arch == 1 ? layout_text_clip(m, B, L, save, nullptr, nullptr) : layout_text(m, B, L, save, nullptr, nullptr);
}

// ------------------------------------------------------------- text fwd ----
// Last BERT layer on the inference path, CLS rows only (modeling_chineseclip.py:349-350 reads bert(...)[0][:, 0, :]): key and
// value projections over all tokens; query, attention, BertSelfOutput, BertIntermediate, BertOutput for token 0 of each
// sentence.  Scratch: q | ctx in b.ctx, y | a | z | x_out in b.y ([B, H] each; L >= 4).  x_out_cls = b.y + 3 * B * H.
// ex (packed batches): the rows of b.x_in are the packed tokens; the CLS row of sample n is row ex->cu[n].
static int bert_last_layer_cls(ezclip_model* m, const ezclip_model::BertLayer& Lw, const BertBufs& b, const float* key_bias, int B,
                               int L, float eps, hipStream_t stream, const TextExtras* ex = nullptr) {
  const bool packed = ex != nullptr && ex->rowmap != nullptr;
  const int H = m->cfg.text_hidden_size, F = m->cfg.text_intermediate_size, M = packed ? ex->packed_rows : B * L, dt = m->dtype;
  const size_t esz = dtype_size(dt), blk = (size_t)B * H * esz;
  char* qkv = (char*)b.qkv;
  char* q_cls = (char*)b.ctx;
  char* ctx_cls = q_cls + blk;
  char* y = (char*)b.y;
  char* a = y + blk;
  char* z = a + blk;
  char* x_out = z + blk;
  const void* x_cls = b.x_in;                    // the CLS rows of the layer input, row stride x_cls_ld
  int64_t x_cls_ld = (int64_t)L * H;
  if (packed) {                                  // irregular row positions: gather them next to the other [B, H] blocks
    char* xc = x_out + blk;                      // (the workspace holds B * L rows whatever was packed; L >= 8)
    EZ_TRY(gather_rows(b.x_in, ex->cu, xc, B, 0, H, 0, dt, stream));
    x_cls = xc;
    x_cls_ld = H;
  }
  EZ_TRY(linear(m, x_cls, x_cls_ld, Lw.q_w, Lw.q_b, q_cls, H, B, ACT_NONE, nullptr, 0, nullptr, false, stream));
  EZ_TRY(bert_qkv_proj(m, Lw, b.x_in, H, qkv, M, 1, stream));
  AttnArgs at;
  at.k = qkv + H * esz; at.v = qkv + 2 * H * esz;
  at.row_stride = 3 * H;
  at.key_bias = key_bias;
  at.B = B; at.L = packed ? ex->max_len : L; at.H = m->theads; at.scale = 0.125f;
  if (packed) { at.cu = ex->cu; at.lens = ex->lens; }
  EZ_TRY(attention_cls_fwd(at, q_cls, H, ctx_cls, H, dt, stream));
  EZ_TRY(linear(m, ctx_cls, H, Lw.o_w, Lw.o_b, y, H, B, ACT_NONE, x_cls, x_cls_ld, nullptr, false, stream));
  EZ_TRY(layernorm_fwd(y, H, a, H, m->P(Lw.ln1_w), m->P(Lw.ln1_b), eps, B, H, dt, nullptr, nullptr, stream));
  EZ_TRY(linear(m, a, H, Lw.i_w, Lw.i_b, b.hh, F, B, ACT_GELU_ERF, nullptr, 0, nullptr, false, stream));
  EZ_TRY(linear(m, b.hh, F, Lw.d_w, Lw.d_b, z, H, B, ACT_NONE, a, H, nullptr, false, stream));
  return layernorm_fwd(z, H, x_out, H, m->P(Lw.ln2_w), m->P(Lw.ln2_b), eps, B, H, dt, nullptr, nullptr, stream);
}

// Training variant of bert_last_layer_cls: intermediates kept in the layer's own buffers (first B rows of y, a, u, hh, z,
// x_out, the LayerNorm statistics; ctx rows [0, B) = the CLS context, rows [B, 2B) = the CLS queries).
// Packed batches: the CLS rows of x_in sit at rows ex->cu[n]; they are gathered into ctx rows [2B, 3B) (kept for the backward).
static int bert_last_layer_cls_save(ezclip_model* m, const ezclip_model::BertLayer& Lw, const BertBufs& b, const float* key_bias,
                                    int B, int L, float eps, hipStream_t stream, const TextExtras* ex = nullptr) {
  const bool packed = ex != nullptr && ex->rowmap != nullptr;
  const int H = m->cfg.text_hidden_size, F = m->cfg.text_intermediate_size, M = packed ? ex->packed_rows : B * L, dt = m->dtype;
  const size_t esz = dtype_size(dt), blk = (size_t)B * H * esz;
  char* qkv = (char*)b.qkv;
  char* ctx_cls = (char*)b.ctx;
  char* q_cls = ctx_cls + blk;
  const void* x_cls = b.x_in;
  int64_t x_cls_ld = (int64_t)L * H;
  if (packed) {
    char* xc = q_cls + blk;
    EZ_TRY(gather_rows(b.x_in, ex->cu, xc, B, 0, H, 0, dt, stream));
    x_cls = xc;
    x_cls_ld = H;
  }
  EZ_TRY(linear(m, x_cls, x_cls_ld, Lw.q_w, Lw.q_b, q_cls, H, B, ACT_NONE, nullptr, 0, nullptr, false, stream));
  EZ_TRY(bert_qkv_proj(m, Lw, b.x_in, H, qkv, M, 1, stream));
  AttnArgs at;
  at.k = qkv + H * esz; at.v = qkv + 2 * H * esz;
  at.row_stride = 3 * H;
  at.key_bias = key_bias;
  at.B = B; at.L = packed ? ex->max_len : L; at.H = m->theads; at.scale = 0.125f;
  if (packed) { at.cu = ex->cu; at.lens = ex->lens; }
  EZ_TRY(attention_cls_fwd(at, q_cls, H, ctx_cls, H, dt, stream));
  EZ_TRY(linear(m, ctx_cls, H, Lw.o_w, Lw.o_b, b.y, H, B, ACT_NONE, x_cls, x_cls_ld, nullptr, false, stream));
  EZ_TRY(layernorm_fwd(b.y, H, b.a, H, m->P(Lw.ln1_w), m->P(Lw.ln1_b), eps, B, H, dt, b.m1, b.r1, stream));
  EZ_TRY(linear(m, b.a, H, Lw.i_w, Lw.i_b, b.hh, F, B, ACT_GELU_ERF, nullptr, 0, b.u, false, stream));
  EZ_TRY(linear(m, b.hh, F, Lw.d_w, Lw.d_b, b.z, H, B, ACT_NONE, b.a, H, nullptr, false, stream));
  return layernorm_fwd(b.z, H, b.x_out, H, m->P(Lw.ln2_w), m->P(Lw.ln2_b), eps, B, H, dt, b.m2, b.r2, stream);
}

int encode_text(ezclip_model* m, const int64_t* ids, int B, int L, float* out, void* wsp, size_t ws_bytes, bool save,
                hipStream_t stream, const TextExtras* ex) {
  EZ_REQUIRE(B > 0 && L > 0 && ids && out && wsp, "encode_text: null/empty argument");
  EZ_REQUIRE(L <= m->cfg.text_max_position_embeddings, "encode_text: seq_len %d > max_position_embeddings %d", L,
             m->cfg.text_max_position_embeddings);
  EZ_REQUIRE(m->weights_fresh, "encode_text: call ezclip_refresh_weights after binding/updating parameters");
  EZ_REQUIRE(((uintptr_t)wsp % 256) == 0, "encode_text: workspace must be 256-byte aligned");
  if (m->text_arch == 1) return encode_text_clip(m, ids, B, L, out, wsp, ws_bytes, save, stream);
  TxtWS ws;
  const size_t need = layout_text(m, B, L, save, wsp, &ws);
  EZ_REQUIRE(ws_bytes >= need, "encode_text: workspace too small (%zu < %zu)", ws_bytes, need);
  const int H = m->cfg.text_hidden_size, F = m->cfg.text_intermediate_size, E = m->cfg.embed_dim;
  const int dt = m->dtype;
  const size_t esz = dtype_size(dt);
  const float eps = m->text_ln_eps;  // layer_norm_eps (1e-12: modeling_chineseclip.py:311, CLIPTextConfig default)
  const TextExtras none;
  if (ex == nullptr) ex = &none;
  // train-mode dropout (set per call by ezclip_set_text_dropout; masks are regenerated from the seed in the backward)
  const float hp = m->drop_hidden, ap = m->drop_attn;
  const uint64_t seed = m->drop_seed;
  // Packed batches: padded positions never reach the CLS feature -- their keys carry the -10000 bias, exp(-10000 + s - max)
  // is exactly 0 in f32 next to any unmasked key, and nothing reads their rows (modeling_chineseclip.py:347-350) -- so the
  // tower runs on the kept tokens only: M = packed_rows rows through every GEMM / LayerNorm, per-sample row ranges in the
  // attention kernels.  (The caller keeps whole sentences that have no unmasked key at all, and every CLS row.)
  const bool packed = ex->rowmap != nullptr;
  if (packed) {
    // (with train-mode dropout the caller packs only batches whose kept tokens are a PREFIX of every sentence: a packed
    // position is then the padded one, and the masks -- numbered by padded rows and positions -- are those of the padded run)
    EZ_REQUIRE(dt == EZCLIP_BF16, "encode_text: packed batches are bf16");
    EZ_REQUIRE(ex->cu && ex->lens && ex->packed_rows >= B && ex->packed_rows <= B * L && ex->max_len >= 1 && ex->max_len <= L &&
               ex->max_len <= (save ? 256 : 288), "encode_text: bad packing (rows %d of %d x %d, longest %d)", ex->packed_rows, B, L, ex->max_len);
  }
  const int M = packed ? ex->packed_rows : B * L;

  // BertEmbeddings.forward modeling_bert.py:95-129 / RobertaEmbeddings.forward roberta/modeling_roberta.py:98-134
  EZ_TRY(bert_embed_ln(ids, m->P(m->word_p), m->P(m->tpos_p), m->P(m->type_p), m->P(m->eln_w), m->P(m->eln_b), eps,
                       ws.x0, ws.layers[0].x_in, ws.m0, ws.r0, ws.key_bias, B, L, H, m->cfg.vocab_size, dt, stream,
                       ex->pos_ids, ex->type_ids, ex->attn_mask, m->cfg.text_max_position_embeddings,
                       m->cfg.text_type_vocab_size, ex->rowmap, ex->packed_rows));
  if (hp > 0.f)                                                                                    // :128
    EZ_TRY(dropout_rows(ws.layers[0].x_in, H, nullptr, 0, ws.layers[0].x_in, H, M, H, make_drop(hp, seed, drop_sid_embed()), dt, stream,
                        ex->rowmap));
  // inference: the last layer only feeds x[:, 0] to the pooler / projection -- see bert_last_layer_cls
  const int nlayers = m->cfg.text_num_hidden_layers;
  const bool cls_infer = !save && g_cls_last && hp == 0.f && ap == 0.f && L >= 8;
  const bool cls_train = save && g_cls_last_train && hp == 0.f && ap == 0.f && L >= 8;
  const bool cls_last = cls_infer || cls_train;
  for (int i = 0; i < nlayers - (cls_last ? 1 : 0); ++i) {
    const auto& Lw = m->bert[i];
    const BertBufs& b = ws.layers[i];
    // BertSelfAttention: separate q/k/v Linear                        :172-200
    char* qkv = (char*)b.qkv;
    EZ_TRY(bert_qkv_proj(m, Lw, b.x_in, H, qkv, M, 0, stream));
    AttnArgs at;
    at.q = qkv; at.k = qkv + H * esz; at.v = qkv + 2 * H * esz;
    at.row_stride = 3 * H;
    at.ctx = b.ctx; at.ctx_stride = H;
    at.key_bias = ws.key_bias; at.lse = b.lse;
    at.B = B; at.L = packed ? ex->max_len : L; at.H = m->theads; at.scale = 0.125f;
    if (packed) { at.cu = ex->cu; at.lens = ex->lens; }
    at.drop = make_drop(ap, seed, drop_sid_attn(i));                // :238
    at.drop_L = L;                                                  // mask rows are numbered with the padded length
    if (ap > 0.f && b.keep != nullptr && at.L <= 256) {             // short kernels with dropout: keep bits for the backward
      at.keep_bits = b.keep; at.keep_words = (at.L + 31) / 32;
    }
    EZ_TRY(attention_fwd(at, dt, stream));                          // :210-248
    // BertSelfOutput: LN(dropout(dense(ctx)) + x)                     :264-267
    if (hp > 0.f) {
      EZ_TRY(linear(m, b.ctx, H, Lw.o_w, Lw.o_b, b.y, H, M, ACT_NONE, nullptr, 0, nullptr, false, stream));
      EZ_TRY(dropout_rows(b.y, H, b.x_in, H, b.y, H, M, H, make_drop(hp, seed, drop_sid_self_out(i)), dt, stream, ex->rowmap));
    } else {
      EZ_TRY(linear(m, b.ctx, H, Lw.o_w, Lw.o_b, b.y, H, M, ACT_NONE, b.x_in, H, nullptr, false, stream));
    }
    EZ_TRY(layernorm_fwd(b.y, H, b.a, H, m->P(Lw.ln1_w), m->P(Lw.ln1_b), eps, M, H, dt, b.m1, b.r1, stream));
    // BertIntermediate (erf GELU) + BertOutput: LN(dropout(dense(h)) + a)   :330-345
    EZ_TRY(linear(m, b.a, H, Lw.i_w, Lw.i_b, b.hh, F, M, ACT_GELU_ERF, nullptr, 0, b.u, false, stream));
    if (hp > 0.f) {
      EZ_TRY(linear(m, b.hh, F, Lw.d_w, Lw.d_b, b.z, H, M, ACT_NONE, nullptr, 0, nullptr, false, stream));
      EZ_TRY(dropout_rows(b.z, H, b.a, H, b.z, H, M, H, make_drop(hp, seed, drop_sid_out(i)), dt, stream, ex->rowmap));
    } else {
      EZ_TRY(linear(m, b.hh, F, Lw.d_w, Lw.d_b, b.z, H, M, ACT_NONE, b.a, H, nullptr, false, stream));
    }
    EZ_TRY(layernorm_fwd(b.z, H, b.x_out, H, m->P(Lw.ln2_w), m->P(Lw.ln2_b), eps, M, H, dt, b.m2, b.r2, stream));
  }
  // chinese_clip: x[:, 0, :] @ text_projection (pooler unused)  modeling_chineseclip.py:349-350
  // huggingface_clip: text_projection(tanh(pooler.dense(x[:, 0])))  appzoo/clip/model.py:134-135, RobertaPooler :550-562
  const void* xl = ws.layers[nlayers - 1].x_out;
  int64_t xl_ld = (int64_t)L * H;
  if (cls_infer) {
    const BertBufs& b = ws.layers[nlayers - 1];
    EZ_TRY(bert_last_layer_cls(m, m->bert[nlayers - 1], b, ws.key_bias, B, L, eps, stream, ex));
    xl = static_cast<const char*>(b.y) + (size_t)3 * B * H * esz;      // x_out of the CLS rows, [B, H]
    xl_ld = H;
  } else if (cls_train) {
    EZ_TRY(bert_last_layer_cls_save(m, m->bert[nlayers - 1], ws.layers[nlayers - 1], ws.key_bias, B, L, eps, stream, ex));
    xl_ld = H;                                                         // x_out of the last layer: [B, H] (CLS rows)
  } else if (packed) {      // CLS rows of the packed last hidden state (the CLS-only last layer is off: dropout, or the switch)
    EZ_TRY(gather_rows(xl, ex->cu, ws.cls_rows, B, 0, H, 0, dt, stream));
    xl = ws.cls_rows;
    xl_ld = H;
  }
  const void* fa = xl;
  int64_t fa_ld = xl_ld;
  if (m->opt_text_pooler) {
    EZ_REQUIRE(m->P(m->pool_w.p) && m->P(m->pool_b), "encode_text: the pooler is enabled but bert.pooler.dense.* is not bound");
    EZ_TRY(linear(m, xl, xl_ld, m->pool_w, m->pool_b, ws.pool, H, B, ACT_TANH, nullptr, 0, ws.pool_u, false, stream));
    fa = ws.pool;
    fa_ld = H;
  }
  EZ_TRY(linear(m, fa, fa_ld, m->tproj_w, m->tproj_b, ws.feat, E, B, ACT_NONE, nullptr, 0, nullptr, true, stream));
  EZ_TRY(l2_normalize_fwd(ws.feat, ws.emb, ws.inv_norm, B, E, stream));   // :363
  EZ_HIP(hipMemcpyAsync(out, ws.emb, (size_t)B * E * 4, hipMemcpyDeviceToDevice, stream));
  return EZ_OK;
}

// ------------------------------------------------------------- backward ----
// dX[M, w.K] = (dY[M, w.N] . W) [* act'(U)] [+ R]      (NT GEMM against the packed W^T copy)
// bias_p >= 0: dX is itself the output gradient of a Linear with that bias: its column sums are accumulated too
static int dgrad(const ezclip_model* m, const void* dY, int64_t ldy, const ezclip_model::Weight& w, void* dX,
                 int64_t ldx, int M, const void* U, int64_t ldu, int act, const void* R, int64_t ldr,
                 hipStream_t stream, int bias_p = -1) {
  EZ_REQUIRE(w.st != nullptr, "backward needs ezclip_set_shadow(..., with_backward=1)");
  GemmArgs g;
  g.A = dY; g.lda = ldy;
  g.B = w.st; g.ldb = w.ldn;
  g.C = dX; g.ldc = ldx;
  g.M = M; g.N = w.K; g.K = w.N;
  g.U = U; g.ldu = ldu; g.act = act;
  g.R = R; g.ldr = ldr;
  g.colsum = bias_p >= 0 ? m->Gp(bias_p) : nullptr;
  return gemm_nt(g, m->dtype, stream);
}

// grad(W) += dY^T . X  (W is [N, K]);  for [K, N]-stored projections: grad += X^T . dY
static int wgrad(const ezclip_model* m, const void* dY, int64_t ldy, const void* X, int64_t ldx,
                 const ezclip_model::Weight& w, int M, hipStream_t stream) {
  float* G = m->Gp(w.p);
  if (G == nullptr) return EZ_OK;
  GemmTNArgs g;
  g.M = M; g.accumulate = 1; g.C = G;
  if (!w.transposed_src) { g.A = dY; g.lda = ldy; g.B = X; g.ldb = ldx; g.N = w.N; g.K = w.K; g.ldc = w.K; }
  else { g.A = X; g.lda = ldx; g.B = dY; g.ldb = ldy; g.N = w.K; g.K = w.N; g.ldc = w.N; }
  return gemm_tn(g, m->dtype, stream);
}

static int bgrad(const ezclip_model* m, const void* dY, int64_t ldy, int M, int N, int bias_p, hipStream_t stream) {
  float* G = m->Gp(bias_p);
  if (G == nullptr) return EZ_OK;
  return colsum_add(dY, ldy, M, N, G, m->dtype, stream);
}

// bias_p >= 0: the dx written is the output gradient of a Linear with that bias; its column sums (= the bias
// gradient) are accumulated by the same kernel instead of a separate pass over dx.
static int ln_bwd(const ezclip_model* m, const void* x, int64_t xs, const void* dy, int64_t dys, int gamma_p, int beta_p,
                  const float* mean, const float* rstd, void* dx, int64_t dxs, const void* dres, int64_t drs, int rows,
                  int D, hipStream_t stream, int bias_p = -1) {
  return layernorm_bwd(x, xs, dy, dys, m->P(gamma_p), mean, rstd, dx, dxs, dres, drs, m->Gp(gamma_p), m->Gp(beta_p), rows,
                       D, m->dtype, stream, bias_p >= 0 ? m->Gp(bias_p) : nullptr);
}

// backward of resblock_forward.  On entry g.gx = d x_out; on exit g.gx = d x_in.  prev_proj_b: bias parameter of the
// PREVIOUS block's c_proj (its gradient = column sums of d x_in, fused into the last LayerNorm backward), or -1.
struct BlockGrads { void *gx, *gx2, *gtmp, *gqkv, *gbig; float* gbpart; };

static int resblock_backward(ezclip_model* m, const ezclip_model::VitLayer& Lw, const VitBufs& b, const BlockDims& d,
                             const BlockGrads& g, int prev_proj_b, hipStream_t stream) {
  const int M = d.M, W = d.W, dt = m->dtype;
  const size_t esz = dtype_size(dt);
  {
    // x_out = x_mid + c_proj(h);  h = QuickGELU(u);  u = c_fc(ln_2(x_mid))        :204
    EZ_TRY(dgrad(m, g.gx, W, Lw.proj_w, g.gbig, 4 * W, M, b.u, 4 * W, ACT_QUICKGELU, nullptr, 0, stream,
                 Lw.fc_b));   // d u (+ c_fc bias gradient)
    EZ_TRY(wgrad(m, g.gx, W, b.h, 4 * W, Lw.proj_w, M, stream));
    // (c_proj bias gradient: accumulated by the ln_bwd that produced gx)
    EZ_TRY(dgrad(m, g.gbig, 4 * W, Lw.fc_w, g.gtmp, W, M, nullptr, 0, ACT_NONE, nullptr, 0, stream));      // d ln_2
    EZ_TRY(wgrad(m, g.gbig, 4 * W, b.ln2, W, Lw.fc_w, M, stream));
    EZ_TRY(ln_bwd(m, b.x_mid, W, g.gtmp, W, Lw.ln2_w, Lw.ln2_b, b.m2, b.r2, g.gx2, W, g.gx, W, M, W, stream,
                  Lw.out_b));   // d x_mid (+ out_proj bias gradient)
    // x_mid = x_in + out_proj(attn(in_proj(ln_1(x_in))))                           :203
    EZ_TRY(dgrad(m, g.gx2, W, Lw.out_w, g.gtmp, W, M, nullptr, 0, ACT_NONE, nullptr, 0, stream));          // d ctx
    EZ_TRY(wgrad(m, g.gx2, W, b.ctx, W, Lw.out_w, M, stream));
    AttnBwdArgs ab;
    ab.f.q = b.qkv;
    ab.f.k = (const char*)b.qkv + (size_t)W * esz;
    ab.f.v = (const char*)b.qkv + (size_t)2 * W * esz;
    ab.f.row_stride = 3 * W;
    ab.f.ctx = b.ctx; ab.f.ctx_stride = W; ab.f.key_bias = nullptr; ab.f.lse = b.lse;
    ab.f.B = d.B; ab.f.L = d.L; ab.f.H = d.heads; ab.f.scale = 0.125f;
    ab.f.causal = d.causal;
    ab.dctx = g.gtmp;
    ab.dq = g.gqkv;
    ab.dk = (char*)g.gqkv + (size_t)W * esz;
    ab.dv = (char*)g.gqkv + (size_t)2 * W * esz;
    if (float* gb = m->Gp(Lw.in_b)) { ab.dbq = gb; ab.dbk = gb + W; ab.dbv = gb + 2 * W; ab.db_part = g.gbpart; }   // in_proj_bias gradient
    EZ_TRY(attention_bwd(ab, dt, stream));
    EZ_TRY(dgrad(m, g.gqkv, 3 * W, Lw.in_w, g.gtmp, W, M, nullptr, 0, ACT_NONE, nullptr, 0, stream));      // d ln_1
    EZ_TRY(wgrad(m, g.gqkv, 3 * W, b.ln1, W, Lw.in_w, M, stream));
    EZ_TRY(ln_bwd(m, b.x_in, W, g.gtmp, W, Lw.ln1_w, Lw.ln1_b, b.m1, b.r1, g.gx, W, g.gx2, W, M, W, stream,
                  prev_proj_b));    // d x_in (+ the previous block's c_proj bias gradient)
  }
  return EZ_OK;
}

// backward of resblock_forward_cls_save.  On entry g.gx = d x_out of the CLS rows, compact [B, W]; on exit g.gx = d x_in
// for every row [M, W] (only the CLS query, the keys and the values carry gradient into the rows of ln_1).
static int resblock_backward_cls(ezclip_model* m, const ezclip_model::VitLayer& Lw, const VitBufs& b, const BlockDims& d,
                                 const BlockGrads& g, int prev_proj_b, hipStream_t stream) {
  const int M = d.M, W = d.W, B = d.B, dt = m->dtype;
  const size_t esz = dtype_size(dt);
  EZ_TRY(dgrad(m, g.gx, W, Lw.proj_w, g.gbig, 4 * W, B, b.u, 4 * W, ACT_QUICKGELU, nullptr, 0, stream, Lw.fc_b));      // d u
  EZ_TRY(wgrad(m, g.gx, W, b.h, 4 * W, Lw.proj_w, B, stream));
  EZ_TRY(dgrad(m, g.gbig, 4 * W, Lw.fc_w, g.gtmp, W, B, nullptr, 0, ACT_NONE, nullptr, 0, stream));                     // d ln_2
  EZ_TRY(wgrad(m, g.gbig, 4 * W, b.ln2, W, Lw.fc_w, B, stream));
  EZ_TRY(ln_bwd(m, b.x_mid, W, g.gtmp, W, Lw.ln2_w, Lw.ln2_b, b.m2, b.r2, g.gx2, W, g.gx, W, B, W, stream, Lw.out_b)); // d x_mid
  EZ_TRY(dgrad(m, g.gx2, W, Lw.out_w, g.gtmp, W, B, nullptr, 0, ACT_NONE, nullptr, 0, stream));                         // d ctx
  EZ_TRY(wgrad(m, g.gx2, W, b.ctx, W, Lw.out_w, B, stream));
  AttnBwdArgs ab;
  ab.f.k = (const char*)b.qkv + (size_t)W * esz;
  ab.f.v = (const char*)b.qkv + (size_t)2 * W * esz;
  ab.f.row_stride = 3 * W;
  ab.f.B = B; ab.f.L = d.L; ab.f.H = d.heads; ab.f.scale = 0.125f;
  ab.dq = g.gqkv;
  ab.dk = (char*)g.gqkv + (size_t)W * esz;
  ab.dv = (char*)g.gqkv + (size_t)2 * W * esz;
  EZ_TRY(attention_cls_bwd(ab, b.qkv, (int64_t)d.L * 3 * W, b.ctx, g.gtmp, W, dt, stream));
  EZ_TRY(bgrad(m, g.gqkv, 3 * W, M, 3 * W, Lw.in_b, stream));
  // the residual x_mid = x_in + ... reaches x_in at the CLS rows only: d x_mid scattered into a zeroed block (gbig is free)
  EZ_HIP(hipMemsetAsync(g.gbig, 0, (size_t)M * W * esz, stream));
  EZ_TRY(gather_rows(g.gx2, nullptr, g.gbig, B, d.L, W, 1, dt, stream));
  EZ_TRY(dgrad(m, g.gqkv, 3 * W, Lw.in_w, g.gtmp, W, M, nullptr, 0, ACT_NONE, nullptr, 0, stream));                     // d ln_1
  EZ_TRY(wgrad(m, g.gqkv, 3 * W, b.ln1, W, Lw.in_w, M, stream));
  return ln_bwd(m, b.x_in, W, g.gtmp, W, Lw.ln1_w, Lw.ln1_b, b.m1, b.r1, g.gx, W, g.gbig, W, M, W, stream, prev_proj_b);
}

int backward_image(ezclip_model* m, const float* pixels, int B, const float* d_emb, void* wsp, size_t ws_bytes,
                   hipStream_t stream) {
  EZ_REQUIRE(B > 0 && d_emb && wsp, "backward_image: null/empty argument");
  EZ_REQUIRE(!m->vit.empty(), "backward_image: this handle has no image tower (created with vision_layers = 0)");
  EZ_REQUIRE(m->weights_fresh && m->shadow_backward, "backward_image: weights not packed for backward");
  ImgWS ws;
  const size_t need = layout_image(m, B, true, wsp, &ws);
  EZ_REQUIRE(ws_bytes >= need, "backward_image: workspace too small (%zu < %zu): was the forward run with save_for_backward?", ws_bytes, need);
  const int W = m->cfg.vision_width, E = m->cfg.embed_dim, Lv = m->Lv;
  const int Mp = B * (Lv - 1), M = B * Lv;
  const int dt = m->dtype;
  const size_t esz = dtype_size(dt);
  (void)pixels;

  // emb = feat / ||feat||  ->  d feat                                   modeling_chineseclip.py:360
  EZ_TRY(l2_normalize_bwd(ws.emb, d_emb, ws.inv_norm, ws.gfeat, B, E, stream));
  const void* gfeatT = ws.gfeat;
  if (dt != EZCLIP_F32) { EZ_TRY(cast_from_f32(ws.gfeat, ws.gfeatT, (int64_t)B * E, dt, stream)); gfeatT = ws.gfeatT; }
  // feat = ln_post(x[:,0]) @ proj                                         :248-251
  EZ_TRY(wgrad(m, gfeatT, E, ws.cls_ln, W, m->vproj_w, B, stream));
  if (m->Gp(m->vproj_b)) EZ_TRY(colsum_add(ws.gfeat, E, B, E, m->Gp(m->vproj_b), EZCLIP_F32, stream));
  if (m->opt_vision_frozen) {                  // image_embeds = vision_outputs[1].detach()   appzoo/clip/model.py:140
    m->progress(0, EZCLIP_STAGE_HEAD, stream);
    return EZ_OK;
  }
  EZ_TRY(dgrad(m, gfeatT, E, m->vproj_w, ws.gcls, W, B, nullptr, 0, ACT_NONE, nullptr, 0, stream));
  const int nl = m->cfg.vision_layers;
  const void* xl = ws.layers[nl - 1].x_out;
  const BlockDims bd{M, W, B, Lv, m->vheads, 0};
  const BlockGrads bg{ws.gx, ws.gx2, ws.gtmp, ws.gqkv, ws.gbig, ws.gbpart};
  const bool cls_train = g_cls_last_train && Lv >= 4;        // (the forward's choice: encode_image)
  if (cls_train) {
    // x_out of the last block is [B, W]; d x_out lands compact in gx (+ the block's c_proj bias gradient)
    EZ_TRY(ln_bwd(m, xl, W, ws.gcls, W, m->lnpost_w, m->lnpost_b, ws.mpost, ws.rpost, ws.gx, W, nullptr, 0, B, W, stream,
                  m->vit[nl - 1].proj_b));
    m->progress(0, EZCLIP_STAGE_HEAD, stream);         // visual.proj, ln_post
    EZ_TRY(resblock_backward_cls(m, m->vit[nl - 1], ws.layers[nl - 1], bd, bg, nl > 1 ? m->vit[nl - 2].proj_b : -1, stream));
    m->progress(0, nl - 1, stream);
  } else {
    EZ_HIP(hipMemsetAsync(ws.gx, 0, (size_t)M * W * esz, stream));
    // (gx is zero outside the CLS rows: its column sums are the last block's c_proj bias gradient)
    EZ_TRY(ln_bwd(m, xl, (int64_t)Lv * W, ws.gcls, W, m->lnpost_w, m->lnpost_b, ws.mpost, ws.rpost, ws.gx, (int64_t)Lv * W,
                  nullptr, 0, B, W, stream, m->vit[nl - 1].proj_b));
    m->progress(0, EZCLIP_STAGE_HEAD, stream);
  }
  for (int i = nl - (cls_train ? 2 : 1); i >= 0; --i) {
    // (block i's c_proj bias gradient was written by block i + 1's last LayerNorm backward: everything of block i is final)
    EZ_TRY(resblock_backward(m, m->vit[i], ws.layers[i], bd, bg, i > 0 ? m->vit[i - 1].proj_b : -1, stream));
    m->progress(0, i, stream);
  }
  // x = ln_pre(cat(cls, conv(patches)) + pos)                                          :237-242
  EZ_TRY(ln_bwd(m, ws.x0, W, ws.gx, W, m->lnpre_w, m->lnpre_b, ws.m0, ws.r0, ws.gx2, W, nullptr, 0, M, W, stream));
  if (m->Gp(m->pos_p)) EZ_TRY(batch_sum_add(ws.gx2, B, Lv, Lv, W, m->Gp(m->pos_p), dt, stream));
  if (m->Gp(m->cls_p)) EZ_TRY(batch_sum_add(ws.gx2, B, Lv, 1, W, m->Gp(m->cls_p), dt, stream));
  if (m->Gp(m->conv_w.p)) {
    EZ_TRY(vit_gather_patch_rows(ws.gx2, ws.gtmp, B, Lv, W, dt, stream));
    GemmTNArgs g;
    g.A = ws.gtmp; g.lda = W; g.B = ws.patches; g.ldb = m->Kpad;
    g.M = Mp; g.N = W;
    if (m->Kpad == m->Kpatch) {
      g.C = m->Gp(m->conv_w.p); g.ldc = m->Kpatch; g.K = m->Kpatch; g.accumulate = 1;
      EZ_TRY(gemm_tn(g, dt, stream));
    } else {   // K padded to the tile multiple (zero columns in `patches`): gradient into scratch, then un-pad
      g.C = ws.gconv; g.ldc = m->Kpad; g.K = m->Kpad; g.accumulate = 0;
      EZ_TRY(gemm_tn(g, dt, stream));
      EZ_TRY(add_cols_f32(m->Gp(m->conv_w.p), m->Kpatch, ws.gconv, m->Kpad, W, m->Kpatch, stream));
    }
  }
  m->progress(0, EZCLIP_STAGE_EMBED, stream);          // class / positional embedding, conv1, ln_pre
  return EZ_OK;
}

// backward of bert_last_layer_cls_save.  On entry ws.gx = d x_out of the CLS rows, compact [B, H]; on exit ws.gx = d x_in
// for every row [M, H] (the key / value projections over all tokens; the query projection and the residual at the CLS rows).
static int bert_last_layer_cls_backward(ezclip_model* m, const ezclip_model::BertLayer& Lw, const BertBufs& b, const TxtWS& ws,
                                        int B, int L, hipStream_t stream, const TextExtras* ex = nullptr) {
  const bool packed = ex != nullptr && ex->rowmap != nullptr;
  const int H = m->cfg.text_hidden_size, F = m->cfg.text_intermediate_size, M = packed ? ex->packed_rows : B * L, dt = m->dtype;
  const size_t esz = dtype_size(dt), blk = (size_t)B * H * esz;
  char* qkv = (char*)b.qkv;
  char* gq = (char*)ws.gqkv;
  const char* ctx_cls = (const char*)b.ctx;
  const char* q_cls = ctx_cls + blk;
  const void* x_cls = packed ? (const void*)(q_cls + blk) : b.x_in;        // CLS rows of x_in (gathered by the forward when packed)
  const int64_t x_cls_ld = packed ? H : (int64_t)L * H;
  // x_out = LN(z);  z = dense(hh) + a;  hh = gelu(u);  u = dense(a)          (B rows)
  EZ_TRY(ln_bwd(m, b.z, H, ws.gx, H, Lw.ln2_w, Lw.ln2_b, b.m2, b.r2, ws.gx2, H, nullptr, 0, B, H, stream, Lw.d_b));        // d z
  EZ_TRY(dgrad(m, ws.gx2, H, Lw.d_w, ws.gbig, F, B, b.u, F, ACT_GELU_ERF, nullptr, 0, stream, Lw.i_b));                  // d u
  EZ_TRY(wgrad(m, ws.gx2, H, b.hh, F, Lw.d_w, B, stream));
  EZ_TRY(dgrad(m, ws.gbig, F, Lw.i_w, ws.gtmp, H, B, nullptr, 0, ACT_NONE, ws.gx2, H, stream));                           // d a
  EZ_TRY(wgrad(m, ws.gbig, F, b.a, H, Lw.i_w, B, stream));
  // a = LN(y);  y = dense(ctx) + x_in[:, 0]
  EZ_TRY(ln_bwd(m, b.y, H, ws.gtmp, H, Lw.ln1_w, Lw.ln1_b, b.m1, b.r1, ws.gx2, H, nullptr, 0, B, H, stream, Lw.o_b));    // d y
  EZ_TRY(dgrad(m, ws.gx2, H, Lw.o_w, ws.gtmp, H, B, nullptr, 0, ACT_NONE, nullptr, 0, stream));                           // d ctx
  EZ_TRY(wgrad(m, ws.gx2, H, ctx_cls, H, Lw.o_w, B, stream));
  AttnBwdArgs ab;
  ab.f.k = qkv + H * esz; ab.f.v = qkv + 2 * H * esz;
  ab.f.row_stride = 3 * H;
  ab.f.key_bias = ws.key_bias;
  ab.f.B = B; ab.f.L = packed ? ex->max_len : L; ab.f.H = m->theads; ab.f.scale = 0.125f;
  if (packed) { ab.f.cu = ex->cu; ab.f.lens = ex->lens; }
  ab.dq = nullptr; ab.dk = gq + H * esz; ab.dv = gq + 2 * H * esz;
  EZ_TRY(attention_cls_bwd(ab, q_cls, H, ctx_cls, ws.gtmp, H, dt, stream, ws.gx3, H));            // d q of the CLS rows -> gx3 [B, H]
  // d x_in: keys and values over all rows ...
  EZ_TRY(dgrad(m, gq + H * esz, 3 * H, Lw.k_w, ws.gx, H, M, nullptr, 0, ACT_NONE, nullptr, 0, stream));
  EZ_TRY(dgrad(m, gq + 2 * H * esz, 3 * H, Lw.v_w, ws.gx, H, M, nullptr, 0, ACT_NONE, ws.gx, H, stream));
  // ... plus, at the CLS rows, the query projection and the residual of y
  EZ_TRY(dgrad(m, ws.gx3, H, Lw.q_w, ws.gtmp, H, B, nullptr, 0, ACT_NONE, ws.gx2, H, stream));
  if (packed) EZ_TRY(gather_rows(ws.gtmp, ex->cu, ws.gx, B, 0, H, 2, dt, stream));
  else EZ_TRY(gather_rows(ws.gtmp, nullptr, ws.gx, B, L, H, 2, dt, stream));
  EZ_TRY(wgrad(m, ws.gx3, H, x_cls, x_cls_ld, Lw.q_w, B, stream));
  EZ_TRY(wgrad(m, gq + H * esz, 3 * H, b.x_in, H, Lw.k_w, M, stream));
  EZ_TRY(wgrad(m, gq + 2 * H * esz, 3 * H, b.x_in, H, Lw.v_w, M, stream));
  EZ_TRY(bgrad(m, ws.gx3, H, B, H, Lw.q_b, stream));
  EZ_TRY(bgrad(m, gq + H * esz, 3 * H, M, H, Lw.k_b, stream));
  return bgrad(m, gq + 2 * H * esz, 3 * H, M, H, Lw.v_b, stream);
}

// backward of encode_text_clip
static int backward_text_clip(ezclip_model* m, const int64_t* ids, int B, int L, const float* d_emb, void* wsp, size_t ws_bytes,
                              hipStream_t stream) {
  TxtClipWS ws;
  const size_t need = layout_text_clip(m, B, L, true, wsp, &ws);
  EZ_REQUIRE(ws_bytes >= need, "backward_text: workspace too small (%zu < %zu): was the forward run with save_for_backward?", ws_bytes, need);
  const int W = m->cfg.text_hidden_size, E = m->cfg.embed_dim, M = B * L, dt = m->dtype;
  const size_t esz = dtype_size(dt);
  const int nl = m->cfg.text_num_hidden_layers;
  EZ_TRY(l2_normalize_bwd(ws.emb, d_emb, ws.inv_norm, ws.gfeat, B, E, stream));
  const void* gfeatT = ws.gfeat;
  if (dt != EZCLIP_F32) { EZ_TRY(cast_from_f32(ws.gfeat, ws.gfeatT, (int64_t)B * E, dt, stream)); gfeatT = ws.gfeatT; }
  // feat = ln_final(x_eot) @ text_projection
  EZ_TRY(dgrad(m, gfeatT, E, m->tproj_w, ws.gcls, W, B, nullptr, 0, ACT_NONE, nullptr, 0, stream));
  EZ_TRY(wgrad(m, gfeatT, E, ws.eot_ln, W, m->tproj_w, B, stream));
  // d x_eot (+ the last block's c_proj bias gradient = its column sums: every other row of d x is zero)
  EZ_TRY(ln_bwd(m, ws.eot_rows, W, ws.gcls, W, m->lnf_w, m->lnf_b, ws.mpost, ws.rpost, ws.geot, W, nullptr, 0, B, W, stream,
                m->ttx[nl - 1].proj_b));
  m->progress(1, EZCLIP_STAGE_HEAD, stream);           // text_projection, ln_final
  EZ_HIP(hipMemsetAsync(ws.gx, 0, (size_t)M * W * esz, stream));
  EZ_TRY(gather_rows(ws.geot, ws.eot, ws.gx, B, L, W, 1, dt, stream));
  const BlockDims bd{M, W, B, L, m->theads, 1};
  const BlockGrads bg{ws.gx, ws.gx2, ws.gtmp, ws.gqkv, ws.gbig, ws.gbpart};
  for (int i = nl - 1; i >= 0; --i) {
    EZ_TRY(resblock_backward(m, m->ttx[i], ws.layers[i], bd, bg, i > 0 ? m->ttx[i - 1].proj_b : -1, stream));
    m->progress(1, i, stream);
  }
  // x = token_embedding[ids] + positional_embedding: index-add and batch sum of d x
  if (m->Gp(m->tok_p)) EZ_TRY(bert_word_grad(ids, ws.gx, m->Gp(m->tok_p), M, W, m->cfg.vocab_size, dt, stream, -1));
  if (m->Gp(m->tpos2_p)) EZ_TRY(batch_sum_add(ws.gx, B, L, L, W, m->Gp(m->tpos2_p), dt, stream));
  m->progress(1, EZCLIP_STAGE_EMBED, stream);
  return EZ_OK;
}

int backward_text(ezclip_model* m, const int64_t* ids, int B, int L, const float* d_emb, void* wsp, size_t ws_bytes,
                  hipStream_t stream, const TextExtras* ex) {
  EZ_REQUIRE(B > 0 && L > 0 && ids && d_emb && wsp, "backward_text: null/empty argument");
  EZ_REQUIRE(m->weights_fresh && m->shadow_backward, "backward_text: weights not packed for backward");
  if (m->text_arch == 1) return backward_text_clip(m, ids, B, L, d_emb, wsp, ws_bytes, stream);
  TxtWS ws;
  const size_t need = layout_text(m, B, L, true, wsp, &ws);
  EZ_REQUIRE(ws_bytes >= need, "backward_text: workspace too small (%zu < %zu): was the forward run with save_for_backward?", ws_bytes, need);
  const int H = m->cfg.text_hidden_size, F = m->cfg.text_intermediate_size, E = m->cfg.embed_dim;
  const int dt = m->dtype;
  const size_t esz = dtype_size(dt);
  // the dropout state of the matching forward (the caller re-arms ezclip_set_text_dropout with the same seed)
  const float hp = m->drop_hidden, ap = m->drop_attn;
  const uint64_t seed = m->drop_seed;
  const TextExtras none0;
  if (ex == nullptr) ex = &none0;
  const bool packed = ex->rowmap != nullptr;       // the matching forward ran on these packed rows (encode_text)
  if (packed)
    EZ_REQUIRE(dt == EZCLIP_BF16 && ex->cu && ex->lens && ex->packed_rows >= B && ex->packed_rows <= B * L &&
               ex->max_len >= 1 && ex->max_len <= 256, "backward_text: bad packing");
  const int M = packed ? ex->packed_rows : B * L;

  EZ_TRY(l2_normalize_bwd(ws.emb, d_emb, ws.inv_norm, ws.gfeat, B, E, stream));                     // chineseclip:363
  const void* gfeatT = ws.gfeat;
  if (dt != EZCLIP_F32) { EZ_TRY(cast_from_f32(ws.gfeat, ws.gfeatT, (int64_t)B * E, dt, stream)); gfeatT = ws.gfeatT; }
  // feat = x[:, 0, :] @ text_projection                                                               :349-350
  const void* xl = ws.layers[m->cfg.text_num_hidden_layers - 1].x_out;
  const int nlayers = m->cfg.text_num_hidden_layers;
  // (the forward's choice, encode_text: last layer on the CLS rows -> its x_out and the gradient entering it are [B, H])
  const bool cls_train = g_cls_last_train && hp == 0.f && ap == 0.f && L >= 8;
  // packed rows without the CLS-only last layer: the head works on the gathered CLS rows [B, H] and their gradient is
  // scattered to the rows cu[b] of an otherwise zero d x_out
  const bool cls_gather = packed && !cls_train;
  if (cls_gather) xl = ws.cls_rows;
  const int64_t xl_ld = (cls_train || cls_gather) ? H : (int64_t)L * H;
  void* gxl = cls_gather ? ws.gcls : ws.gx;      // where the head leaves d x[:, 0]
  if (!cls_train) EZ_HIP(hipMemsetAsync(ws.gx, 0, (size_t)M * H * esz, stream));
  if (m->Gp(m->tproj_b)) EZ_TRY(colsum_add(ws.gfeat, E, B, E, m->Gp(m->tproj_b), EZCLIP_F32, stream));
  if (m->opt_text_pooler) {
    // feat = proj(tanh(u)), u = pooler.dense(x[:, 0]):  d u = (d feat . W_proj) o tanh'(u)  (+ pooler bias gradient)
    EZ_TRY(dgrad(m, gfeatT, E, m->tproj_w, ws.gpool, H, B, ws.pool_u, H, ACT_TANH, nullptr, 0, stream, m->pool_b));
    EZ_TRY(wgrad(m, gfeatT, E, ws.pool, H, m->tproj_w, B, stream));
    EZ_TRY(dgrad(m, ws.gpool, H, m->pool_w, gxl, xl_ld, B, nullptr, 0, ACT_NONE, nullptr, 0, stream));
    EZ_TRY(wgrad(m, ws.gpool, H, xl, xl_ld, m->pool_w, B, stream));
  } else {
    EZ_TRY(dgrad(m, gfeatT, E, m->tproj_w, gxl, xl_ld, B, nullptr, 0, ACT_NONE, nullptr, 0, stream));
    EZ_TRY(wgrad(m, gfeatT, E, xl, xl_ld, m->tproj_w, B, stream));
  }
  if (cls_gather) EZ_TRY(gather_rows(ws.gcls, ex->cu, ws.gx, B, 0, H, 1, dt, stream));      // d x_out[cu[b]] = d cls[b]
  m->progress(1, EZCLIP_STAGE_HEAD, stream);           // text_projection (+ bias), pooler
  if (cls_train) {
    EZ_TRY(bert_last_layer_cls_backward(m, m->bert[nlayers - 1], ws.layers[nlayers - 1], ws, B, L, stream, ex));
    m->progress(1, nlayers - 1, stream);
  }
  for (int i = nlayers - (cls_train ? 2 : 1); i >= 0; --i) {
    const auto& Lw = m->bert[i];
    const BertBufs& b = ws.layers[i];
    // x_out = LN(z);  z = dense(hh) + a;  hh = gelu(u);  u = dense(a)        modeling_bert.py:330-345
    const void* gd = ws.gx2;      // gradient of the dense output (= d z, or its dropout-masked copy)
    if (hp > 0.f) {
      EZ_TRY(ln_bwd(m, b.z, H, ws.gx, H, Lw.ln2_w, Lw.ln2_b, b.m2, b.r2, ws.gx2, H, nullptr, 0, M, H, stream));          // d z
      EZ_TRY(dropout_rows(ws.gx2, H, nullptr, 0, ws.gx3, H, M, H, make_drop(hp, seed, drop_sid_out(i)), dt, stream, ex->rowmap));
      EZ_TRY(bgrad(m, ws.gx3, H, M, H, Lw.d_b, stream));
      gd = ws.gx3;
    } else {
      EZ_TRY(ln_bwd(m, b.z, H, ws.gx, H, Lw.ln2_w, Lw.ln2_b, b.m2, b.r2, ws.gx2, H, nullptr, 0, M, H, stream, Lw.d_b));  // d z
    }
    EZ_TRY(dgrad(m, gd, H, Lw.d_w, ws.gbig, F, M, b.u, F, ACT_GELU_ERF, nullptr, 0, stream, Lw.i_b));                // d u
    EZ_TRY(wgrad(m, gd, H, b.hh, F, Lw.d_w, M, stream));
    EZ_TRY(dgrad(m, ws.gbig, F, Lw.i_w, ws.gtmp, H, M, nullptr, 0, ACT_NONE, ws.gx2, H, stream));                     // d a = d z + d u W_i
    EZ_TRY(wgrad(m, ws.gbig, F, b.a, H, Lw.i_w, M, stream));
    // a = LN(y);  y = dense(ctx) + x_in                                           :264-267
    gd = ws.gx2;
    if (hp > 0.f) {
      EZ_TRY(ln_bwd(m, b.y, H, ws.gtmp, H, Lw.ln1_w, Lw.ln1_b, b.m1, b.r1, ws.gx2, H, nullptr, 0, M, H, stream));       // d y
      EZ_TRY(dropout_rows(ws.gx2, H, nullptr, 0, ws.gx3, H, M, H, make_drop(hp, seed, drop_sid_self_out(i)), dt, stream, ex->rowmap));
      EZ_TRY(bgrad(m, ws.gx3, H, M, H, Lw.o_b, stream));
      gd = ws.gx3;
    } else {
      EZ_TRY(ln_bwd(m, b.y, H, ws.gtmp, H, Lw.ln1_w, Lw.ln1_b, b.m1, b.r1, ws.gx2, H, nullptr, 0, M, H, stream, Lw.o_b)); // d y
    }
    EZ_TRY(dgrad(m, gd, H, Lw.o_w, ws.gtmp, H, M, nullptr, 0, ACT_NONE, nullptr, 0, stream));                         // d ctx
    EZ_TRY(wgrad(m, gd, H, b.ctx, H, Lw.o_w, M, stream));
    char* qkv = (char*)b.qkv;
    char* gq = (char*)ws.gqkv;
    AttnBwdArgs ab;
    ab.f.q = qkv; ab.f.k = qkv + H * esz; ab.f.v = qkv + 2 * H * esz;
    ab.f.row_stride = 3 * H;
    ab.f.ctx = b.ctx; ab.f.ctx_stride = H; ab.f.key_bias = ws.key_bias; ab.f.lse = b.lse;
    ab.f.B = B; ab.f.L = packed ? ex->max_len : L; ab.f.H = m->theads; ab.f.scale = 0.125f;
    if (packed) { ab.f.cu = ex->cu; ab.f.lens = ex->lens; }
    ab.f.drop = make_drop(ap, seed, drop_sid_attn(i));
    ab.f.drop_L = L;
    if (ap > 0.f && b.keep != nullptr && ab.f.L <= 256) {         // (what the forward chose: encode_text)
      ab.f.keep_bits = b.keep; ab.f.keep_words = (ab.f.L + 31) / 32;
    }
    ab.dctx = ws.gtmp;
    ab.dq = gq; ab.dk = gq + H * esz; ab.dv = gq + 2 * H * esz;
    const bool qkv_bias_grads = m->Gp(Lw.q_b) && m->Gp(Lw.k_b) && m->Gp(Lw.v_b);
    if (qkv_bias_grads) { ab.dbq = m->Gp(Lw.q_b); ab.dbk = m->Gp(Lw.k_b); ab.dbv = m->Gp(Lw.v_b); ab.db_part = ws.gbpart; }
    EZ_TRY(attention_bwd(ab, dt, stream));                                                                           // :210-248
    // d x_in = d y + d q W_q + d k W_k + d v W_v                                  :172-200
    EZ_TRY(dgrad(m, gq, 3 * H, Lw.q_w, ws.gx, H, M, nullptr, 0, ACT_NONE, ws.gx2, H, stream));
    EZ_TRY(dgrad(m, gq + H * esz, 3 * H, Lw.k_w, ws.gx, H, M, nullptr, 0, ACT_NONE, ws.gx, H, stream));
    EZ_TRY(dgrad(m, gq + 2 * H * esz, 3 * H, Lw.v_w, ws.gx, H, M, nullptr, 0, ACT_NONE, ws.gx, H, stream));
    EZ_TRY(wgrad(m, gq, 3 * H, b.x_in, H, Lw.q_w, M, stream));
    EZ_TRY(wgrad(m, gq + H * esz, 3 * H, b.x_in, H, Lw.k_w, M, stream));
    EZ_TRY(wgrad(m, gq + 2 * H * esz, 3 * H, b.x_in, H, Lw.v_w, M, stream));
    if (!qkv_bias_grads) {
      EZ_TRY(bgrad(m, gq, 3 * H, M, H, Lw.q_b, stream));
      EZ_TRY(bgrad(m, gq + H * esz, 3 * H, M, H, Lw.k_b, stream));
      EZ_TRY(bgrad(m, gq + 2 * H * esz, 3 * H, M, H, Lw.v_b, stream));
    }
    m->progress(1, i, stream);
  }
  // embeddings: dropout(LN(word[ids] + type[0] + pos[t]))                       modeling_bert.py:117-128
  if (hp > 0.f) EZ_TRY(dropout_rows(ws.gx, H, nullptr, 0, ws.gx, H, M, H, make_drop(hp, seed, drop_sid_embed()), dt, stream, ex->rowmap));
  // token-type gradient: with the default all-zero types it is the column sum of gx2 into row 0 (fused into ln_bwd)
  EZ_TRY(ln_bwd(m, ws.x0, H, ws.gx, H, m->eln_w, m->eln_b, ws.m0, ws.r0, ws.gx2, H, nullptr, 0, M, H, stream,
                ex->type_ids ? -1 : m->type_p));
  // (packed rows: the index tensors stay [B, L] and are read at rowmap[r]; dropped tokens have no gradient in the reference
  // either -- nothing of them reaches the loss)
  if (ex->type_ids && m->Gp(m->type_p))
    EZ_TRY(bert_word_grad(ex->type_ids, ws.gx2, m->Gp(m->type_p), M, H, m->cfg.text_type_vocab_size, dt, stream, -1, ex->rowmap, L));
  if (m->Gp(m->word_p))
    EZ_TRY(bert_word_grad(ids, ws.gx2, m->Gp(m->word_p), M, H, m->cfg.vocab_size, dt, stream, m->text_pad_id, ex->rowmap, L));
  if (m->Gp(m->tpos_p)) {
    if (ex->pos_ids)    // RobertaEmbeddings: nn.Embedding(max_pos, H, padding_idx=pad_token_id)
      EZ_TRY(bert_word_grad(ex->pos_ids, ws.gx2, m->Gp(m->tpos_p), M, H, m->cfg.text_max_position_embeddings, dt, stream,
                            m->text_pad_id, ex->rowmap, L));
    else if (packed)    // position t = rowmap[r] % L
      EZ_TRY(bert_word_grad(nullptr, ws.gx2, m->Gp(m->tpos_p), M, H, m->cfg.text_max_position_embeddings, dt, stream, -1, ex->rowmap, L));
    else
      EZ_TRY(batch_sum_add(ws.gx2, B, L, L, H, m->Gp(m->tpos_p), dt, stream));
  }
  m->progress(1, EZCLIP_STAGE_EMBED, stream);
  return EZ_OK;
}

}  // namespace ezclip
