// Host-side model state for the ezclip library: parameter table (reference
// names -> caller-owned device pointers), packed weight shadows, workspace
// layouts, and the layer loops that enqueue the kernels.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/ezclip.h"
#include "ezclip_common.h"
#include "kernels.h"

struct ezclip_model {
  ezclip_config cfg;
  int dtype = 0;
  // derived
  int Lv = 0, G = 0, Kpatch = 0, Kpad = 0, vheads = 0, theads = 0;

  struct Param {
    std::string name;
    std::vector<int64_t> shape;
    int64_t numel = 0;
    float* w = nullptr;   // caller-owned f32 master weight
    float* g = nullptr;   // caller-owned f32 grad buffer (may be null)
  };
  std::vector<Param> params;
  std::map<std::string, int> index;

  // A GEMM weight [N, K] (nn.Linear layout) and its packed copies
  struct Weight {
    int p = -1;             // index into params (weight)
    int N = 0, K = 0;
    int ldk = 0;            // leading dim of the packed [N, ldk] copy (K padded to the tile multiple)
    bool transposed_src = false;  // master stored [K, N] (visual.proj / text_projection)
    void* s = nullptr;      // packed [N, ldk] in compute dtype (== master for plain f32)
    void* st = nullptr;     // packed transposed [K(ld N), N] for input gradients (with_backward only)
    int ldn = 0;            // leading dim of st ([K, ldn])
    // folded-LayerNorm copy (bf16 inference path): W o g, c1 = row sums of it, c2 = W b + bias
    int fold_g = -1, fold_b = -1, fold_bias = -1;   // param indices of the LN gain / shift and the Linear bias
    void* sf = nullptr;
    float *c1 = nullptr, *c2 = nullptr;
    bool s_external = false;  // `s` is a slice of a block laid out elsewhere (BertLayer::qkv_s)
    mutable bool sf_fresh = false;   // the folded copy follows the master weights lazily (first folded product after a refresh)
  };
  struct VitLayer {
    Weight in_w, out_w, fc_w, proj_w;
    int in_b, out_b, fc_b, proj_b, ln1_w, ln1_b, ln2_w, ln2_b;
  };
  struct BertLayer {
    Weight q_w, k_w, v_w, o_w, i_w, d_w;
    int q_b, k_b, v_b, o_b, i_b, d_b, ln1_w, ln1_b, ln2_w, ln2_b;
    // bf16: the packed query / key / value weights sit back to back ([3H, ldk]: q | k | v rows) next to a packed copy of
    // the three biases, so the projections run as ONE N = 3H product (or N = 2H for key | value) instead of three N = H ones
    void* qkv_s = nullptr;
    float* qkv_bias = nullptr;
  };
  Weight conv_w, vproj_w, tproj_w, pool_w;
  int vproj_b = -1, tproj_b = -1, pool_b = -1;    // optional parameters (huggingface_clip branch): projection biases, pooler
  int cls_p, pos_p, lnpre_w, lnpre_b, lnpost_w, lnpost_b;
  int word_p, tpos_p, type_p, eln_w, eln_b, logit_scale_p;
  std::vector<VitLayer> vit;
  std::vector<BertLayer> bert;
  // text_arch 1 (open_clip, OPEN_CLIP.encode_text modeling_openclip.py:354-368): token + positional embedding, pre-LN
  // causal residual attention blocks (the ViT's block type), ln_final, feature of the EOT token, text_projection
  int text_arch = 0;
  std::vector<VitLayer> ttx;
  int tok_p = -1, tpos2_p = -1, lnf_w = -1, lnf_b = -1;

  // huggingface_clip branch of CLIPApp (appzoo/clip/model.py:73-104,128-144): ezclip_set_option
  bool opt_text_pooler = false;      // text feature = projection(tanh(pooler.dense(x[:, 0])))  (RobertaModel pooled output)
  bool opt_vision_frozen = false;    // image_embeds = vision_outputs[1].detach(): only the projection gets gradients
  float text_ln_eps = 1e-12f;        // BertConfig / CLIPTextConfig layer_norm_eps
  int64_t text_pad_id = 0;           // padding_idx of the word / position embeddings (no gradient for that row)
  // wukong_clip flavour of the pre-LN towers (modelzoo/models/wukong/modeling_wukong.py:238-361)
  float block_ln_eps = 1e-5f;        // eps of every LayerNorm of the residual-attention-block towers (wukong: 1e-7)
  int64_t text_eot_id = -1;          // text_arch 1: pooled row = first position holding this id (wukong: 102); -1: argmax of the ids

  // BERT train-mode dropout (ezclip_set_text_dropout): probabilities + the seed of the next forward / backward pair
  float drop_hidden = 0.f, drop_attn = 0.f;
  uint64_t drop_seed = 0;

  // ezclip_set_backward_progress: host callback after each parameter group's gradient kernels are enqueued
  ezclip_progress_fn progress_fn = nullptr;
  void* progress_user = nullptr;
  // ezclip_backward_progress_events (round 4): instead of calling into the host from inside the backward call, record an event
  // on the producing stream per finished group and let the caller drain (tower, stage, event) after the call has returned
  struct ProgressItem { int tower, stage; hipEvent_t ev; };
  bool progress_log = false;
  mutable std::vector<ProgressItem> progress_items;       // logged since the last drain, in completion order
  mutable std::vector<hipEvent_t> progress_pool;          // events owned by the handle, reused round-robin
  mutable size_t progress_next = 0;
  void progress(int tower, int stage, hipStream_t stream) const;

  // ezclip_pack_text_meta: a ring of 8 x 4 ints of pinned, device-mapped host memory (rows, longest, prefix, ticket)
  int* pm_host = nullptr;
  int* pm_dev = nullptr;
  int pm_ticket = 0;

  void* shadow = nullptr;
  size_t shadow_bytes = 0;
  bool shadow_backward = false;
  bool weights_fresh = false;
  // the re-packing jobs of ezclip_refresh_weights (kernels.h CastJob): device table inside the shadow buffer, host mirror
  void* cast_jobs_dev = nullptr;
  std::vector<unsigned char> cast_jobs_host;
  int cast_tiles = 0;

  float* P(int i) const { return params[i].w; }
  float* Gp(int i) const { return params[i].g; }
};

namespace ezclip {

int model_create(const ezclip_config* cfg, ezclip_model** out, int text_arch = 0);
size_t model_shadow_layout(ezclip_model* m, char* base, bool with_backward);  // returns bytes; assigns when base != null
int model_refresh_weights(ezclip_model* m, hipStream_t stream);
void set_cls_last_train(int on);     // 1 (default): the same on the training path (forward with save + backward)
void set_cls_q_only(int on);         // 1 (default): the CLS-only last ViT block projects its queries for the CLS rows only
void set_cls_last(int on);           // 1 (default): on the inference path the last block of a tower runs on the CLS rows only
void set_fuse_bert_qkv(int on);      // 1 (default): BERT q / k / v projections as one product on the bf16 path
void set_fold_layernorm(int mode);   // 0: separate LayerNorm kernels; 1: folded + row statistics from the producing GEMM; 2: folded + separate statistics pass

size_t image_workspace_bytes(const ezclip_model* m, int B, bool save);
size_t text_workspace_bytes(const ezclip_model* m, int B, int L, bool save);
int encode_image(ezclip_model* m, const float* pixels, int B, float* out, void* ws, size_t ws_bytes, bool save,
                 hipStream_t stream);
// optional per-token inputs of the text tower (RobertaModel(input_ids, token_type_ids, attention_mask); position ids are
// RobertaEmbeddings' pad-aware ones, computed by the caller); any may be null: position t, type 0, mask = ids != 0
struct TextExtras {
  const int64_t* pos_ids = nullptr;
  const int64_t* type_ids = nullptr;
  const int64_t* attn_mask = nullptr;
  // Packed text batches (ezclip_encode_text_packed): only the tokens rowmap[r] = b * L + t, r < packed_rows, go through the
  // tower -- sample b owns packed rows cu[b] .. cu[b] + lens[b] - 1, its first one is the CLS token.  Inference, bf16.
  const int* rowmap = nullptr;
  const int* cu = nullptr;
  const int* lens = nullptr;
  int packed_rows = 0, max_len = 0;
};
int encode_text(ezclip_model* m, const int64_t* ids, int B, int L, float* out, void* ws, size_t ws_bytes, bool save,
                hipStream_t stream, const TextExtras* ex = nullptr);
int backward_image(ezclip_model* m, const float* pixels, int B, const float* d_emb, void* ws, size_t ws_bytes,
                   hipStream_t stream);
int backward_text(ezclip_model* m, const int64_t* ids, int B, int L, const float* d_emb, void* ws, size_t ws_bytes,
                  hipStream_t stream, const TextExtras* ex = nullptr);

}  // namespace ezclip
