/* ezclip -- MI355X (gfx950) native CLIP text-image retrieval hot path.
 *
 * C ABI of libezclip_hip.so.  Plain pointers and sizes only; every pointer
 * named `*_dev` / documented as "device" is a HIP device pointer owned by the
 * caller (e.g. a PyTorch-ROCm tensor's data_ptr()).  `stream` is a hipStream_t
 * passed as void* (torch.cuda.current_stream().cuda_stream); every call only
 * enqueues work on that stream -- no hidden synchronisation.
 *
 * What each entry point replaces in alibaba/EasyNLP (paths relative to the
 * reference checkout).  The reference has no native code on this path: each
 * symbol below stands in for a composition of stock torch ops.
 *
 *   ezclip_encode_image   CHINESE_CLIP.encode_image + L2 normalise
 *                         easynlp/modelzoo/models/clip/modeling_chineseclip.py:343-344,358-360
 *                         (VisualTransformer.forward :236-253, ResidualAttentionBlock :184-205)
 *   ezclip_encode_text    CHINESE_CLIP.encode_text + L2 normalise
 *                         modeling_chineseclip.py:346-350,361-363
 *                         (BertModel.forward easynlp/modelzoo/models/bert/modeling_bert.py:792-920)
 *   ezclip_similarity     logits_per_text = T @ I^T * exp(logit_scale)
 *                         easynlp/appzoo/clip/model.py:148
 *   ezclip_infonce_*      CLIPApp.compute_loss / clip_loss / contrastive_loss
 *                         easynlp/appzoo/clip/model.py:154-164
 *   ezclip_backward_*     autograd backward of the above (easynlp/core/trainer.py:658-661)
 *   ezclip_recall_ranks   CLIPEvaluator.evaluate's sort loop
 *                         easynlp/appzoo/clip/evaluator.py:50-61
 *   ezclip_bind_param     names/shapes of CHINESE_CLIP.state_dict()
 *                         (checkpoint contract, easynlp/appzoo/clip/model.py:63-72)
 *
 * Error handling: every function returns 0 on success, non-zero otherwise;
 * ezclip_last_error() returns a thread-local message.  Nothing throws across
 * the ABI.  A handle is not thread-safe; use one handle per (device, stream).
 */
#ifndef EZCLIP_H_
#define EZCLIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EZCLIP_OK 0
#define EZCLIP_ERR_INVALID 1
#define EZCLIP_ERR_HIP 2
#define EZCLIP_ERR_UNSUPPORTED 3
#define EZCLIP_ERR_STATE 4

#define EZCLIP_DTYPE_F32 0
#define EZCLIP_DTYPE_BF16 1

#define EZCLIP_ACT_NONE 0
#define EZCLIP_ACT_QUICKGELU 1
#define EZCLIP_ACT_GELU_ERF 2
#define EZCLIP_ACT_TANH 3

/* CHINESE_CLIP constructor kwargs (modeling_chineseclip.py:256-276) + compute dtype. */
typedef struct ezclip_config {
  int32_t embed_dim;
  int32_t image_resolution;
  int32_t vision_layers;
  int32_t vision_width;
  int32_t vision_patch_size;
  int32_t vocab_size;
  int32_t text_hidden_size;
  int32_t text_intermediate_size;
  int32_t text_max_position_embeddings;
  int32_t text_num_attention_heads;
  int32_t text_num_hidden_layers;
  int32_t text_type_vocab_size;
  int32_t compute_dtype; /* EZCLIP_DTYPE_* : dtype of activations / GEMM operands; accumulation is f32 */
} ezclip_config;

typedef struct ezclip_model* ezclip_handle;

const char* ezclip_last_error(void);
const char* ezclip_version(void);

/* ---- model lifetime / parameters ------------------------------------------ */
int ezclip_create(const ezclip_config* cfg, ezclip_handle* out);
/* text_arch: EZCLIP_TEXT_BERT (ezclip_create; BertModel / RobertaModel) or EZCLIP_TEXT_CLIP -- the open_clip branch's text
 * tower (OPEN_CLIP.encode_text, easynlp/modelzoo/models/clip/modeling_openclip.py:296-311,354-368): token_embedding +
 * positional_embedding, pre-LN residual attention blocks with the causal mask of build_attention_mask (:343-349),
 * ln_final, the feature of the EOT token (argmax of the ids) times text_projection.  Parameter names are OPEN_CLIP's
 * (token_embedding.weight, positional_embedding, transformer.resblocks.*, ln_final.*, text_projection); config fields:
 * text_hidden_size = transformer_width, text_num_attention_heads = transformer_heads (width / 64),
 * text_num_hidden_layers = transformer_layers, text_max_position_embeddings = context_length,
 * text_intermediate_size = 4 * width, text_type_vocab_size ignored. */
#define EZCLIP_TEXT_BERT 0
#define EZCLIP_TEXT_CLIP 1
int ezclip_create_ex(const ezclip_config* cfg, int text_arch, ezclip_handle* out);
void ezclip_destroy(ezclip_handle h);
/* Number of parameters the model expects and the i-th reference name (state_dict key without
 * the "chinese_clip." prefix); shape is written to shape[0..*ndim). */
int ezclip_num_params(ezclip_handle h);
int ezclip_param_info(ezclip_handle h, int index, const char** name, int64_t* shape, int* ndim);
/* Bind a float32 device tensor (and optionally its float32 gradient buffer) to a reference
 * parameter name.  The library reads weights through these pointers; it never owns them. */
int ezclip_bind_param(ezclip_handle h, const char* name, void* weight_dev, void* grad_dev,
                      const int64_t* shape, int ndim);
/* Packed weight copies (bf16 casts, transposes for the projections / input gradients).  The
 * caller allocates `ezclip_shadow_bytes` bytes of device memory, hands it over once, and calls
 * ezclip_refresh_weights after every update of the bound parameters (optimizer step, load). */
size_t ezclip_shadow_bytes(ezclip_handle h, int with_backward);
int ezclip_set_shadow(ezclip_handle h, void* shadow_dev, size_t bytes, int with_backward);
int ezclip_refresh_weights(ezclip_handle h, void* stream);

/* ---- huggingface_clip branch of CLIPApp (easynlp/appzoo/clip/model.py:73-104,128-144) -------------------------
 * The same towers under other parameter names (the host layer maps them) plus four differences handled here:
 *   EZCLIP_OPT_TEXT_POOLER   1: text feature = text_projection(tanh(bert.pooler.dense(x[:, 0]))) -- RobertaModel's pooled
 *                               output text_outputs[1] (model.py:134; roberta/modeling_roberta.py:550-562)
 *   EZCLIP_OPT_VISION_FROZEN 1: image_embeds = vision_outputs[1].detach() (model.py:140): ezclip_backward_image stops
 *                               after the projection (weight / bias gradients only)
 *   EZCLIP_OPT_TEXT_LN_EPS      layer_norm_eps of the text tower (default 1e-12)
 *   EZCLIP_OPT_TEXT_PAD_ID      padding_idx of the word / position embeddings (no gradient for that row; default 0)
 * and two optional parameters "visual.proj_bias" / "text_projection_bias" [embed_dim] (vision_projection /
 * text_projection are nn.Linear with bias there); unbound = no bias.
 * ---- wukong_clip (easynlp/appzoo/wukong_clip/model.py:44-73; modelzoo/models/wukong/modeling_wukong.py:238-361) -------
 * the EZCLIP_TEXT_CLIP towers with
 *   EZCLIP_OPT_BLOCK_LN_EPS     eps of every LayerNorm of the residual-attention-block towers: ViT incl. ln_pre / ln_post,
 *                               EZCLIP_TEXT_CLIP text tower incl. ln_final (default 1e-5; wukong builds them with 1e-7)
 *   EZCLIP_OPT_TEXT_EOT_ID      EZCLIP_TEXT_CLIP only: the text feature is taken at the first position holding this token
 *                               id (wukong: 102, `x[(text == 102).nonzero()]`) instead of argmax(ids); -1 = argmax (default) */
#define EZCLIP_OPT_TEXT_POOLER 1
#define EZCLIP_OPT_VISION_FROZEN 2
#define EZCLIP_OPT_TEXT_LN_EPS 3
#define EZCLIP_OPT_TEXT_PAD_ID 4
#define EZCLIP_OPT_BLOCK_LN_EPS 5
#define EZCLIP_OPT_TEXT_EOT_ID 6
int ezclip_set_option(ezclip_handle h, int key, double value);
/* ezclip_encode_text / ezclip_backward_text with the per-token inputs RobertaModel takes (model.py:131-133): device
 * int64 [batch, seq_len] each, any may be NULL: position_ids (default 0..L-1; RobertaEmbeddings' pad-aware ids
 * cumsum(ids != pad) * (ids != pad) + pad are computed by the caller, roberta/modeling_roberta.py:1497-1510),
 * token_type_ids (default 0), attention_mask (default ids != 0). */
int ezclip_encode_text_ex(ezclip_handle h, const int64_t* input_ids_dev, const int64_t* position_ids_dev,
                          const int64_t* token_type_ids_dev, const int64_t* attention_mask_dev, int batch, int seq_len,
                          float* out_embeds_dev, void* workspace_dev, size_t workspace_bytes, int save_for_backward,
                          void* stream);
int ezclip_backward_text_ex(ezclip_handle h, const int64_t* input_ids_dev, const int64_t* position_ids_dev,
                            const int64_t* token_type_ids_dev, const int64_t* attention_mask_dev, int batch, int seq_len,
                            const float* d_embeds_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* PACKED text batches (bf16; forward with or without save_for_backward, and the matching backward).  Padded positions never reach the text feature: their keys carry the
 * (1 - mask) * -10000 bias (modeling_utils.py:427,438-439), exp(-10000 + s - max) is exactly 0 in float32 next to any unmasked
 * key, and only x[:, 0] is read afterwards (modeling_chineseclip.py:349-350) -- so the tower needs to see the kept tokens only.
 * The caller decides which tokens are kept and passes (device int32):
 *   rowmap [packed_rows]  flat index b * seq_len + t of every kept token, sample by sample, t ascending
 *   cu [batch], lens [batch]  first packed row and number of packed rows of each sample (lens >= 1: row cu[b] must be the CLS
 *                          token t = 0; a sentence without any unmasked key must be kept whole -- its softmax is uniform over
 *                          ALL positions in the reference)
 *   max_len                the largest lens (<= 288)
 * input_ids / position_ids / token_type_ids / attention_mask stay [batch, seq_len] (the last three may be NULL) and are read
 * at rowmap[r].  Every GEMM / LayerNorm of the tower then runs over packed_rows instead of batch * seq_len rows.
 * Workspace: ezclip_text_workspace_bytes(h, batch, seq_len, save_for_backward).  Same embeddings as ezclip_encode_text(_ex),
 * and the same gradients: a dropped token has no path to the loss in the reference either (its key is masked in every layer,
 * its own outputs feed nothing), so its embedding rows get exactly zero there too.  Training: longest sample <= 256.
 * With train-mode dropout armed (ezclip_set_text_dropout) the caller must only pack batches in which the kept tokens of every
 * sentence are a PREFIX (t = 0 .. lens[b] - 1): the dropout decisions are numbered by padded rows and positions, and only then is
 * a packed position the padded one -- the packed run then regenerates exactly the padded run's masks (bit-identical embeddings). */
int ezclip_encode_text_packed(ezclip_handle h, const int64_t* input_ids_dev, const int64_t* position_ids_dev,
                              const int64_t* token_type_ids_dev, const int64_t* attention_mask_dev, const int32_t* rowmap_dev,
                              const int32_t* cu_dev, const int32_t* lens_dev, int batch, int seq_len, int packed_rows,
                              int max_len, float* out_embeds_dev, void* workspace_dev, size_t workspace_bytes,
                              int save_for_backward, void* stream);
int ezclip_backward_text_packed(ezclip_handle h, const int64_t* input_ids_dev, const int64_t* position_ids_dev,
                                const int64_t* token_type_ids_dev, const int64_t* attention_mask_dev, const int32_t* rowmap_dev,
                                const int32_t* cu_dev, const int32_t* lens_dev, int batch, int seq_len, int packed_rows,
                                int max_len, const float* d_embeds_dev, void* workspace_dev, size_t workspace_bytes,
                                void* stream);

/* The packing metadata itself, built on the device in ONE launch and handed to the host without a stream synchronisation
 * (round 3, csrc/packmeta.hip): keep[b][t] = mask[b][t] != 0 (mask = attention_mask_dev, or input_ids_dev when it is NULL:
 * modeling_chineseclip.py:347) or t == 0 or sentence b has no unmasked token.  rowmap [batch * seq_len] (first `rows` entries
 * written), cu [batch], lens [batch]: int32 device buffers of the caller.  The launch is asynchronous on `stream`; *ticket
 * identifies it.  ezclip_pack_text_meta_result blocks the HOST until that one launch has written its result words into pinned
 * host memory owned by the handle (a polled flag -- no hipStreamSynchronize, no hipMemcpy; work queued behind the launch keeps
 * the device busy) and returns rows (= packed_rows), longest (= max_len) and prefix (1: the kept tokens of every sentence are a
 * prefix -- the condition for packing under dropout).  At most 8 tickets may be outstanding.  seq_len <= 512.
 * NOT usable while `stream` is being captured into a hipGraph: a captured launch does not execute, so the poll would only return
 * after its 120 s timeout (EZ_ERR_HIP).  Under capture pass host-computed metadata to ezclip_encode_text_packed instead. */
int ezclip_pack_text_meta(ezclip_handle h, const int64_t* input_ids_dev, const int64_t* attention_mask_dev, int batch,
                          int seq_len, int32_t* rowmap_dev, int32_t* cu_dev, int32_t* lens_dev, int* ticket, void* stream);
int ezclip_pack_text_meta_result(ezclip_handle h, int ticket, int* rows, int* longest, int* prefix);


/* Train-mode dropout of the BERT text tower: nn.Dropout(hidden_dropout_prob) after the embedding LayerNorm, after
 * BertSelfOutput.dense and BertOutput.dense, nn.Dropout(attention_probs_dropout_prob) on the attention probabilities
 * (easynlp/modelzoo/models/bert/modeling_bert.py:128,238,266,344; probabilities from CHINESE_CLIP's
 * text_hidden_dropout_prob / text_attention_probs_dropout_prob, modeling_chineseclip.py:266-268,306-307).
 * The setting applies to every following ezclip_encode_text / ezclip_backward_text on this handle; masks are
 * counter-based (Philox-4x32-10 keyed by `seed`, never stored), so a backward call must see the same seed as its
 * forward.  Eval mode = (0, 0, any).  The ViT tower has no dropout (nn.MultiheadAttention dropout 0). */
int ezclip_set_text_dropout(ezclip_handle h, float hidden_p, float attention_p, uint64_t seed);

/* ---- forward ------------------------------------------------------------------ */
/* Workspace (activations; with save_for_backward != 0 also everything backward needs). */
size_t ezclip_image_workspace_bytes(ezclip_handle h, int batch, int save_for_backward);
size_t ezclip_text_workspace_bytes(ezclip_handle h, int batch, int seq_len, int save_for_backward);
/* pixels: device float32 [batch, 3, R, R] (NCHW, as CLIPDataset.batch_fn produces);
 * out_embeds: device float32 [batch, embed_dim], L2-normalised. */
int ezclip_encode_image(ezclip_handle h, const float* pixels_dev, int batch, float* out_embeds_dev,
                        void* workspace_dev, size_t workspace_bytes, int save_for_backward, void* stream);
/* input_ids: device int64 [batch, seq_len]; attention mask = ids != 0 (modeling_chineseclip.py:347). */
int ezclip_encode_text(ezclip_handle h, const int64_t* input_ids_dev, int batch, int seq_len,
                       float* out_embeds_dev, void* workspace_dev, size_t workspace_bytes,
                       int save_for_backward, void* stream);
/* out[i][j] = exp(*logit_scale) * <a[i], b[j]>;  a [na, e], b [nb, e], out [na, nb] float32.
 * logit_scale_dev may be NULL (scale 1: CLIPEvaluator's `agreement`). */
int ezclip_similarity(const float* a_dev, const float* b_dev, int na, int nb, int e,
                      const float* logit_scale_dev, float* out_dev, void* stream);

/* ---- InfoNCE ------------------------------------------------------------------- */
/* Loss of the reference's compute_loss on a materialised logits_per_text [n, n]:
 *   0.5 * (CE(S, arange) + CE(S^T, arange)).  scratch: >= 4*n floats.  */
int ezclip_infonce_from_logits(const float* logits_dev, int n, float* loss_dev, float* scratch_dev,
                               void* stream);
/* d loss / d logits for the above, scaled by *grad_out_dev (scalar, device). */
int ezclip_infonce_from_logits_bwd(const float* logits_dev, int n, const float* grad_out_dev,
                                   float* dlogits_dev, float* scratch_dev, void* stream);
/* One direction only -- CLIPApp.contrastive_loss(logits) = F.cross_entropy(logits, arange(len(logits)))
 * (appzoo/clip/model.py:154-155): mean over the rows of (logsumexp(logits[i, :]) - logits[i, i]).  logits [rows, cols] float32,
 * row-major with leading dimension ld >= cols, rows <= cols.  scratch: >= 2*rows floats (holds the row log-sum-exps for the
 * backward call: pass the same buffer).  The backward writes d loss / d logits * (*grad_out_dev) into dlogits [rows, cols]
 * (leading dimension cols). */
int ezclip_cross_entropy_diag(const float* logits_dev, int rows, int cols, int64_t ld, float* loss_dev, float* scratch_dev,
                              void* stream);
int ezclip_cross_entropy_diag_bwd(const float* logits_dev, int rows, int cols, int64_t ld, const float* grad_out_dev,
                                  const float* scratch_dev, float* dlogits_dev, void* stream);
/* Contrastive step on embeddings (local or all-gathered global batch): loss, d embeddings and d logit_scale of THIS rank's
 * n_local rows of both directions against all n_global columns.
 *   text_all/image_all: float32 [n_global, e] (rows rank_offset..rank_offset+n_local are this rank's)
 *   loss = 0.5 * (mean_i CE(s * T_loc I_all^T)_i + mean_i CE(s * I_loc T_all^T)_i),  s = exp(*logit_scale)
 *   outputs (all three may be NULL to skip the backward part):
 *     d_text_all / d_image_all  float32 [n_global, e] (overwritten): gradient contributions of THIS rank's loss
 *        w.r.t. every row (sum over ranks = reduce-scatter / all-reduce by the caller)
 *     d_logit_scale float32 scalar (overwritten)
 *   grad_scale multiplies every gradient (1/world_size for DDP-style averaging).
 *
 * ezclip_infonce_tiled (round 3, csrc/nce.hip; e in {128, 256, 512, 768, 1024}): NOTHING of size n_local x n_global is
 * materialised.  Score tiles are bf16 MFMA products; the forward keeps a running log-sum-exp per row, the backward recomputes
 * the tiles, forms d(logits) in registers and multiplies it with the streamed rows in the same kernel; partial sums over key
 * chunks are combined in a fixed order (bit-reproducible).  split_operands = 1 splits every operand into bf16 hi + lo parts
 * (three products, float32-class results: loss 1e-5, gradients 1e-5 -- the f32 pipeline); 0 rounds the embeddings to bf16 once
 * (the bf16 pipeline).  Workspace O((n_local + n_global) * e): ezclip_infonce_tiled_workspace_bytes (0 = e not supported).
 *
 * ezclip_infonce_fused: the same contract on the materialising path of rounds 1-2 (exact float32: two [n_local, n_global] f32
 * logit blocks + their gradients in the workspace, f32-MFMA products, row kernels of loss.hip) -- any e % 32 == 0; the f32
 * pipeline's choice while the blocks are small.  workspace: ezclip_infonce_workspace_bytes(n_local, n_global, e). */
size_t ezclip_infonce_workspace_bytes(int n_local, int n_global, int e);
int ezclip_infonce_fused(const float* text_all_dev, const float* image_all_dev, int n_local, int n_global,
                         int rank_offset, int e, const float* logit_scale_dev, float grad_scale,
                         float* loss_dev, float* d_text_all_dev, float* d_image_all_dev,
                         float* d_logit_scale_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
size_t ezclip_infonce_tiled_workspace_bytes(int n_local, int n_global, int e);
int ezclip_infonce_tiled(const float* text_all_dev, const float* image_all_dev, int n_local, int n_global,
                         int rank_offset, int e, const float* logit_scale_dev, float grad_scale, int split_operands,
                         float* loss_dev, float* d_text_all_dev, float* d_image_all_dev,
                         float* d_logit_scale_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- ModifiedResNet image tower (csrc/resnet.hip) ------------------------------------------------
 * CHINESE_CLIP builds `ModifiedResNet(vision_layers, embed_dim, heads, image_resolution, vision_width)` when `vision_layers` is a
 * tuple (modeling_chineseclip.py:279-287; Bottleneck :27-74, AttentionPool2d :77-108).  It runs behind its own handle, in EVAL
 * mode (BatchNorm with running statistics): a frozen image tower -- evaluation, prediction, LiT-style fine-tuning of the text
 * tower (the text side then comes from a handle created with vision_layers = 0).  Convolutions are MFMA GEMMs over NHWC
 * activations (3x3: read implicitly, no im2col buffer), BatchNorm is folded into the packed weights by
 * ezclip_rn_refresh_weights, the attention pool runs the one-query attention kernel.
 * Parameter names = the reference checkpoint's ("visual.conv1.weight", "visual.bn1.running_var", "visual.layer1.0.downsample.1.bias",
 * "visual.attnpool.k_proj.weight", ...; `num_batches_tracked` counters are not read).  All float32, caller-owned. */
typedef struct ezclip_rn* ezclip_rn_handle;
typedef struct ezclip_rn_config {
  int32_t layers[4];         /* blocks per stage, e.g. {3, 4, 6, 3} for RN50 */
  int32_t width;             /* 64 for RN50: stem 32/32/64, stages 64/128/256/512 -> x4 */
  int32_t output_dim;        /* embed_dim */
  int32_t image_resolution;  /* multiple of 32 */
  int32_t compute_dtype;     /* EZCLIP_DTYPE_F32 | EZCLIP_DTYPE_BF16 */
} ezclip_rn_config;
int ezclip_rn_create(const ezclip_rn_config* cfg, ezclip_rn_handle* out);
void ezclip_rn_destroy(ezclip_rn_handle h);
int ezclip_rn_num_params(ezclip_rn_handle h);
int ezclip_rn_param_info(ezclip_rn_handle h, int index, const char** name, int64_t* shape8, int* ndim);
int ezclip_rn_bind_param(ezclip_rn_handle h, const char* name, const float* value_dev);
size_t ezclip_rn_shadow_bytes(ezclip_rn_handle h);                       /* packed, BatchNorm-folded weight copies */
int ezclip_rn_set_shadow(ezclip_rn_handle h, void* shadow_dev, size_t bytes);
int ezclip_rn_refresh_weights(ezclip_rn_handle h, void* stream);          /* after binding / changing parameters */
size_t ezclip_rn_workspace_bytes(ezclip_rn_handle h, int batch);         /* bounded: the batch is walked in chunks */
/* pixels [batch, 3, R, R] float32 NCHW -> out [batch, output_dim] float32, L2-normalised (as ezclip_encode_image) */
int ezclip_rn_encode_image(ezclip_rn_handle h, const float* pixels_dev, int batch, float* out_dev, void* workspace_dev,
                           size_t workspace_bytes, void* stream);

/* Training path of the tower (BatchNorm in training mode, backward pass: modeling_chineseclip.py:27-167 under module.train() and
 * torch autograd).  The whole batch runs at once (batch statistics): no chunking; batch * (R / 32)^2 > 1 (more than one value per
 * channel in every BatchNorm, as torch requires); output_dim a multiple of 16 bytes of the compute dtype.
 *   ezclip_rn_set_train_shadow / ezclip_rn_refresh_train_weights   unfolded packed convolution weights and the packed weights of the
 *                                  input-gradient products; refresh after every parameter update (besides ezclip_rn_refresh_weights)
 *   ezclip_rn_bind_grad            float32 gradient buffer of a parameter (every parameter but the running statistics; WRITTEN, not
 *                                  added to, by ezclip_rn_backward)
 *   ezclip_rn_encode_image_train   forward with batch statistics; moves the bound running_mean / running_var buffers (momentum 0.1);
 *                                  keeps every activation in saved_ws (ezclip_rn_train_saved_bytes) until the backward has run.
 *                                  After it the inference copies are stale: call ezclip_rn_refresh_weights before ezclip_rn_encode_image.
 *   ezclip_rn_backward             features_dev = the forward's output, d_features_dev [batch, output_dim] f32 its gradient, saved_dev =
 *                                  the workspace THAT forward filled (any number of forwards may be pending, each in its own
 *                                  workspace: every pointer is re-derived from the base; a workspace no forward filled is refused) */
size_t ezclip_rn_train_shadow_bytes(ezclip_rn_handle h);
int ezclip_rn_set_train_shadow(ezclip_rn_handle h, void* shadow_dev, size_t bytes);
int ezclip_rn_refresh_train_weights(ezclip_rn_handle h, void* stream);
int ezclip_rn_bind_grad(ezclip_rn_handle h, const char* name, float* grad_dev);
size_t ezclip_rn_train_saved_bytes(ezclip_rn_handle h, int batch);
size_t ezclip_rn_train_scratch_bytes(ezclip_rn_handle h, int batch);
int ezclip_rn_encode_image_train(ezclip_rn_handle h, const float* pixels_dev, int batch, float* out_dev, void* saved_dev, size_t saved_bytes,
                                 void* scratch_dev, size_t scratch_bytes, void* stream);
int ezclip_rn_backward(ezclip_rn_handle h, const float* features_dev, const float* d_features_dev, int batch, const void* saved_dev,
                       size_t saved_bytes, void* scratch_dev, size_t scratch_bytes, void* stream);

/* ---- backward ------------------------------------------------------------------- */
/* Gradients are ACCUMULATED (+=) into the grad buffers bound with ezclip_bind_param (float32),
 * like autograd does; parameters bound without a grad buffer are skipped.
 * workspace must be the one the matching forward (save_for_backward=1) filled. */
int ezclip_backward_image(ezclip_handle h, const float* pixels_dev, int batch, const float* d_embeds_dev,
                          void* workspace_dev, size_t workspace_bytes, void* stream);
int ezclip_backward_text(ezclip_handle h, const int64_t* input_ids_dev, int batch, int seq_len,
                         const float* d_embeds_dev, void* workspace_dev, size_t workspace_bytes,
                         void* stream);

/* Progress of a running ezclip_backward_image / ezclip_backward_text: `fn(user, tower, stage)` is called ON THE HOST,
 * from inside the backward call, each time every kernel that writes the gradients of one parameter group has been
 * ENQUEUED on the stream (nothing is synchronised: the caller records an event / starts a collective ordered behind the
 * stream).  tower: 0 image, 1 text.  stage: EZCLIP_STAGE_HEAD (projection, ln_post / ln_final, pooler), then the layer
 * index L-1 ... 0 (all parameters of that block), then EZCLIP_STAGE_EMBED (embedding tables, ln_pre / embedding
 * LayerNorm, conv1) -- the order in which the groups' gradients become final.  This is the hook a data-parallel caller
 * uses to overlap the gradient all-reduce with the rest of the backward pass, as DistributedDataParallel's bucket
 * hooks do for the reference (easynlp/core/trainer.py:101-108).  fn = NULL removes the hook. */
#define EZCLIP_STAGE_HEAD 1000000
#define EZCLIP_STAGE_EMBED (-1)
typedef void (*ezclip_progress_fn)(void* user, int tower, int stage);
int ezclip_set_backward_progress(ezclip_handle h, ezclip_progress_fn fn, void* user);
/* The same progress WITHOUT host code inside the backward call (round 4; what CLIPApp.contrastive_step(reduce_gradients=True)
 * uses): while the log is enabled, ezclip_backward_image / _text record one event (hipEventDisableTiming, owned by the handle)
 * on the producing stream per finished group and append (tower, stage, event) to a list.  After the backward call has RETURNED
 * -- its kernels are merely enqueued; the host runs tens of milliseconds ahead of the device -- the caller drains the list and
 * orders its collective behind each event (ezclip_stream_wait_event, or hipStreamWaitEvent on the handle it was given: the
 * events are plain hipEvent_t).  ezclip_backward_progress_events(h, 1) (re)arms the log and recycles the events: call it once
 * per training step, before the backward calls; (h, 0) switches it off.  The events come from a ring of 256 owned by the
 * handle: a drained event stays un-re-recorded until 256 later groups have been logged (several passes), whether or not the
 * log is re-armed in between, and a log that is never drained stops at 256 items (further groups are dropped, with the
 * thread's error message set; so is a group whose event could not be created or recorded).
 * max_items: room in the three arrays (a pass logs at most layers + 2 groups per tower). */
int ezclip_backward_progress_events(ezclip_handle h, int enable);
int ezclip_backward_progress_drain(ezclip_handle h, int* towers, int* stages, void** events, int max_items, int* n_items);
/* hipStreamWaitEvent(stream, event): lets a host that has no HIP binding of its own order a stream behind a drained event. */
int ezclip_stream_wait_event(void* stream, void* event);

/* ---- input pipeline (image half) ---------------------------------------------------- */
/* CLIPDataset.convert_single_row_to_example's image branch on the GPU (easynlp/appzoo/clip/data.py:256-262):
 * _resize (:52-72, PIL BICUBIC, shorter side -> size) -> _center_crop (:29-50) -> _normalize (:101-135: /255,
 * (x - mean) / std in float32, CHW).  Bit-identical to the reference running on Pillow (Resample.c 8-bit path).
 *   packed_dev : decoded RGB8 pixels (HWC, rows tightly packed) of all n images in one device buffer
 *   desc_host  : per image byte offset into packed_dev, width, height (HOST array; read during the call)
 *   out_dev    : float32 [n, 3, crop, crop] -- the pixel_values of ezclip_encode_image
 * Resampling windows are computed on the host inside the call (double precision, as Pillow does) and uploaded through
 * a library-owned pinned buffer; the device work is enqueued on `stream`.  Images whose resized size is smaller than
 * the crop (never the case with size >= crop) are rejected. */
typedef struct ezclip_image_desc {
  uint64_t offset;
  int32_t width, height;
} ezclip_image_desc;
size_t ezclip_preprocess_workspace_bytes(const ezclip_image_desc* desc_host, int n, int size, int crop);
int ezclip_preprocess_images(const uint8_t* packed_dev, const ezclip_image_desc* desc_host, int n, int size, int crop,
                             const float* mean3_host, const float* std3_host, float* out_dev, void* workspace_dev,
                             size_t workspace_bytes, void* stream);

/* Host-only: the resampling windows of one axis for outputs [first, first + count) -- ksize taps per output,
 * bounds[o] = (first source index, taps), kk[o][ksize] int32 with 22 fractional bits (Pillow Resample.c
 * precompute_coeffs + normalize_coeffs_8bpc, BICUBIC).  No device work: lets CPU-only tests pin the tables. */
int ezclip_op_resample_table(int in_size, int out_size, int first, int count, int* ksize, int* bounds_host, int* kk_host,
                             int kk_capacity);

/* The same table as built by the device kernel ezclip_preprocess_images uses by default (ezclip_debug_set(5, 0) selects the
 * host tables): device int32 buffers bounds [count][2], kk [count][ksize]; ksize as reported by ezclip_op_resample_table. */
int ezclip_op_resample_table_device(int in_size, int out_size, int first, int count, int* bounds_dev, int* kk_dev, void* stream);

/* ---- retrieval metric -------------------------------------------------------------- */
/* rank_out[i] = number of images j with sim(text i, image j) > sim(text i, image i)
 * (+ ties with j < i, matching a stable descending sort); text/image: float32 [n, e]. */
int ezclip_recall_ranks(const float* text_dev, const float* image_dev, int n, int e, int32_t* rank_out_dev,
                        float* scratch_dev /* n*n floats */, void* stream);

/* The same for a block of query rows (queries row0 .. row0 + rows - 1 of the n, against all n images): the similarity
 * block [rows, n] is the only scratch -- a 50 000-pair validation set evaluated 4096 queries at a time needs 0.8 GB instead
 * of 10 GB, and the blocks are what ranks shard when the evaluation is distributed.
 * text_rows_dev: [rows, e] (the block's queries); image_dev: [n, e]; rank_out_dev: [rows]; scratch_dev: rows * n floats. */
int ezclip_recall_ranks_rows(const float* text_rows_dev, const float* image_dev, int rows, int row0, int n, int e,
                             int32_t* rank_out_dev, float* scratch_dev, void* stream);

/* Fused form: the similarity tile is compared in the registers of the f32 MFMA kernel that computes it --
 * no similarity block is written or read back, and both retrieval directions come out of one sweep.  Two steps:
 *   ezclip_recall_paired_scores   paired_dev[i] = <text i, image i> for all n pairs, by the tiles on the diagonal only
 *                                 (the values are the ones the sweep recomputes: same kernel, same accumulation order);
 *   ezclip_recall_ranks_fused     queries row0 .. row0 + rows - 1 against all n images:
 *        rank_t2i_dev[r]   = #{j : s(i, j) > s(i, i) or (== and j < i)},  i = row0 + r      (overwritten; CLIPEvaluator's sort loop,
 *                                                                          reference appzoo/clip/evaluator.py:53-61)
 *        rank_i2t_dev[j]  += #{i in the block : s(i, j) > s(j, j) or (== and i < j)}         (optional, may be NULL: image -> text;
 *                            int32 [n], ZEROED BY THE CALLER, complete after every block of queries has been swept -- by one
 *                            process, or summed over the ranks that shared the blocks)
 * text_rows_dev: [rows, e] f32; image_dev: [n, e] f32; paired_dev: [n] f32.  e * 4 must be a multiple of 128 bytes. */
int ezclip_recall_paired_scores(const float* text_dev, const float* image_dev, int n, int e, float* paired_dev, void* stream);
int ezclip_recall_ranks_fused(const float* text_rows_dev, const float* image_dev, int rows, int row0, int n, int e,
                              const float* paired_dev, int32_t* rank_t2i_dev, int32_t* rank_i2t_dev, void* stream);

/* ---- measurement hooks ---------------------------------------------------------------- */
/* Between begin and end every launch of the hot kernels is bracketed by HIP events on its
 * own stream.  kernel_class: 0 = MFMA GEMM (work = algorithmic FLOPs), 1 = fused attention
 * (FLOPs), 2 = LayerNorm (bytes).  ezclip_profile_end synchronises on the recorded events. */
#define EZCLIP_PROF_GEMM 0
#define EZCLIP_PROF_ATTN 1
#define EZCLIP_PROF_ROWOP 2
/* Tuning / A-B switches for sweeps and tests.  key 0: GEMM kernel (-1 heuristic, 0 128x128, 1 256x256 two-phase,
 * 2 8-phase);  key 1: attention kernels (-1 heuristic, 0 general two-pass kernels only);
 * key 2: LayerNorm folding on the bf16 inference path (0 off, 1 folded + statistics from the producing GEMM, 2 folded +
 * separate statistics pass);  key 3: last-block CLS-only evaluation on the inference path (1 on, 0 off);  key 4: the same on the training path;  key 5: resampling window
 * tables of ezclip_preprocess_images built on the device (1, default) or on the host (0);  key 6: tile order of the persistent
 * GEMM (0 column-fastest, g > 0 super-rows of g row tiles walked column by column, -1 the built-in default);  key 7: BERT
 * query / key / value projections as one N = 3 * hidden product on the bf16 path (1, default) or three products (0);
 * key 8: the CLS-only last ViT block projects its queries for the CLS rows only (1, default);  key 9: short attention forward:
 * bit 0 short last tile, bit 1 row sums on the matrix pipe (3, default);  key 10: ModifiedResNet tower: bound of one activation
 * buffer in MiB (256; tests lower it to walk a small batch in chunks);  key 11: fused attention backward for sequences up to
 * 256 tokens: 1 (default) the score-tile-once kernel (round 4) up to 128 tokens -- where it measures faster -- and the two-pass
 * kernel of rounds 2-3 beyond, 2 the score-tile-once kernel wherever it is eligible (<= 256 tokens), 0 the two-pass kernel only;
 * key 12: de-phasing of the persistent GEMM's workgroups (staggered first tiles: steps + 100 * period code; 0 off, -1 built-in default). */
int ezclip_debug_set(int key, int value);
int ezclip_profile_begin(void);
int ezclip_profile_end(int kernel_class, double* total_ms, double* total_work, int* launches);

/* ---- operator-level entry points (unit parity tests, micro-benchmarks) ------------- */
int ezclip_op_gemm_nt(const void* a_dev, int64_t lda, const void* b_dev, int64_t ldb, void* c_dev, int64_t ldc,
                      const float* bias_dev, const void* residual_dev, int64_t ldr, int m, int n, int k,
                      int act, int dtype, int out_f32, void* stream);
/* Every epilogue of the NT GEMM (what the towers use internally), for parity tests at full size:
 *   C = act(alpha * A.B^T + bias) + R                         (c2_dev: also the value before act / residual)
 *   C = (alpha * A.B^T) * act'(U)                              (u_dev; colsum_dev[n] += sum_m C[m][n], f32 [n])
 *   C = act(rstd_m * (A.B^T) - rstd_m mean_m c1[n] + c2[n])   (ln_stats_dev [m][2] = (rstd, -mean rstd), ln_c1 / ln_c2 [n])
 *   rowstat_part_dev [m][n / 64][2]: per row and 64-column slab (sum, sum of squares) of the rounded outputs (with R only)
 * force_kernel: -1 heuristic, 0 the 128x128 kernel, 2 the persistent 8-phase kernel, 24 the 8-phase kernel launched with
 * one workgroup per tile (no persistent loop).  Unsupported combinations return an error (no silent fallback). */
typedef struct ezclip_gemm_desc {
  const void* a_dev; int64_t lda;       /* [m, k] */
  const void* b_dev; int64_t ldb;       /* [n, k] */
  void* c_dev; int64_t ldc;             /* [m, n] */
  void* c2_dev;                         /* optional [m, n] (ldc) */
  const float* bias_dev;                /* optional [n] */
  const void* residual_dev; int64_t ldr;
  const void* u_dev; int64_t ldu;
  const float* ln_stats_dev; const float* ln_c1_dev; const float* ln_c2_dev;
  float* rowstat_part_dev;
  float* colsum_dev;
  float alpha;
  int32_t m, n, k, act, dtype, out_f32, force_kernel;
} ezclip_gemm_desc;
int ezclip_op_gemm_nt_ex(const ezclip_gemm_desc* d, void* stream);
/* stats[row] = (rstd, -mean * rstd) of LayerNorm(x[row]) -- the operand of the folded-LayerNorm epilogue */
int ezclip_op_layernorm_stats(const void* x_dev, int64_t x_stride, float eps, int rows, int d, int dtype, float* stats_dev,
                              void* stream);
int ezclip_op_gemm_tn(const void* a_dev, int64_t lda, const void* b_dev, int64_t ldb, float* c_dev, int64_t ldc,
                      int m, int n, int k, int accumulate, int dtype, void* stream);
/* c [n, 9 * cp] (+)= a[images * h * w, n]^T . (the 3 x 3 neighbourhoods of the NHWC activation x [images * h * w, cp]): column
 * (ky * 3 + kx) * cp + ch = the pixel shifted by (ky - 1, kx - 1), zero outside the image -- ezclip_op_gemm_tn on
 * ezclip_op_rn_im2col3x3(x) without the column matrix, bit for bit (the weight gradient of torch's conv2d backward for a 3 x 3 /
 * padding 1 convolution: modeling_chineseclip.py:34, :121-125 under core/trainer.py:658-661).  w >= 4. */
int ezclip_op_gemm_tn_conv3x3(const void* a_dev, int64_t lda, const void* x_dev, int images, int h, int w, int cp, float* c_dev,
                              int64_t ldc, int n, int accumulate, int dtype, void* stream);
/* The same product for 64 or 128 (padded) channels in and out, bf16 -- the stem's conv2 / conv3 and the conv2 of layer1 / layer2 of the
 * ModifiedResNet (modeling_chineseclip.py:121-125, :34), the convolutions with the most pixels: every 64-output x 64-channel block reads
 * its halves of x [images * h * w, cp] and dz [images * h * w, opad] once, out [opad, ldo >= 9 * cp] f32 (+)=, scratch >= (cp / 64) *
 * (opad / 64) * 147 456 bytes (one [64][576] f32 partial per workgroup, up to 512 are used), summed in a fixed order
 * (bit-reproducible).  Other shapes / dtypes: refused (ezclip_op_gemm_tn_conv3x3 takes them). */
int ezclip_op_rn_wgrad3x3_c64(const void* x_dev, const void* dz_dev, int images, int h, int w, int cp, int opad, void* scratch_dev,
                              size_t scratch_bytes, float* out_dev, int64_t ldo, int accumulate, void* stream);
/* c [n, k] f32 (+)= a[m, n]^T . b[m, k], bf16, for MANY rows and a small result (m >= 4096; n, k multiples of 64; at most 16 blocks of
 * 64 x 256): the weight gradients of the 1 x 1 convolutions of layer1 / layer2 and of the stem's first convolution
 * (modeling_chineseclip.py:30-46, :117-119 under core/trainer.py:658-661).  Every 64-row block of c reads its columns of a once;
 * scratch >= (n / 64) * (k / kb) * 64 * kb * 4 bytes (kb = the widest of 256 / 128 / 64 dividing k; one partial per workgroup, up to
 * 512 in all), summed in a fixed order (bit-reproducible).  Other shapes / dtypes: refused (ezclip_op_gemm_tn takes them). */
int ezclip_op_rn_tn_skinny(const void* a_dev, int64_t lda, const void* b_dev, int64_t ldb, float* c_dev, int64_t ldc, int64_t m, int n, int k,
                           int accumulate, void* scratch_dev, size_t scratch_bytes, void* stream);
int ezclip_op_layernorm(const void* x_dev, int64_t x_stride, void* y_dev, int64_t y_stride, const float* g_dev,
                        const float* b_dev, float eps, int rows, int d, int dtype, float* mean_dev, float* rstd_dev,
                        void* stream);
int ezclip_op_layernorm_bwd(const void* x_dev, const void* dy_dev, const float* g_dev, const float* mean_dev,
                            const float* rstd_dev, void* dx_dev, float* dg_dev, float* db_dev, int rows, int d,
                            int dtype, void* stream);
/* Options of ONE op-level attention call (round 3: arguments, no longer process-wide switches -- two users of the library in one
 * process cannot disturb each other).  NULL = plain softmax(QK^T / 8 + key_bias) V.
 *   causal        key index > query index -> -inf (OPEN_CLIP.build_attention_mask, modeling_openclip.py:343-349)
 *   dropout_p/_seed/_site   dropout on the probabilities with the library's counter-based mask (see ezclip_op_dropout_mask) */
typedef struct ezclip_attention_opts {
  int32_t causal;
  float dropout_p;
  uint64_t dropout_seed;
  uint32_t dropout_site;
  uint32_t reserved_;
} ezclip_attention_opts;
int ezclip_op_attention(const void* q_dev, const void* k_dev, const void* v_dev, int64_t row_stride, void* ctx_dev,
                        int64_t ctx_stride, const float* key_bias_dev, float* lse_dev, int batch, int seq_len,
                        int heads, int dtype, const ezclip_attention_opts* opts, void* stream);
int ezclip_op_attention_bwd(const void* q_dev, const void* k_dev, const void* v_dev, int64_t row_stride,
                            const void* ctx_dev, const void* dctx_dev, int64_t ctx_stride, const float* key_bias_dev,
                            const float* lse_dev, void* dq_dev, void* dk_dev, void* dv_dev, int batch, int seq_len,
                            int heads, int dtype, const ezclip_attention_opts* opts, void* stream);
/* The same with the gradients of the q / k / v projection biases: db*[h*64 + d] += sum over (sample, token) of dq / dk / dv
 * (f32 [heads*64] each, ACCUMULATED into).  db_scratch_dev: batch * 3 * heads * 64 floats (per-sample partial sums of the fused
 * short-sequence kernel; unused by the general kernels, may then be NULL).  The towers' backward passes call
 * exactly this (nn.MultiheadAttention in_proj_bias: modeling_chineseclip.py:226; BertSelfAttention q/k/v biases:
 * modeling_bert.py:176-178). */
int ezclip_op_attention_bwd_bias(const void* q_dev, const void* k_dev, const void* v_dev, int64_t row_stride,
                                 const void* ctx_dev, const void* dctx_dev, int64_t ctx_stride, const float* key_bias_dev,
                                 const float* lse_dev, void* dq_dev, void* dk_dev, void* dv_dev, float* dbq_dev, float* dbk_dev,
                                 float* dbv_dev, float* db_scratch_dev, int batch, int seq_len, int heads, int dtype,
                                 const ezclip_attention_opts* opts, void* stream);
/* Dropout building blocks (parity tests feed the library's own masks to the oracle).
 * Element (row, col) of site `site` is kept iff philox4x32_10(ctr = (col>>2, row, site, 0), key = seed)[col&3] >=
 * round(p * 2^32); survivors are scaled by 1/(1-p).  Text-tower sites: 0 = embeddings; layer i: 1+3i attention
 * probabilities (row = (b*heads + h)*L + query, col = key), 2+3i BertSelfOutput, 3+3i BertOutput (row = token). */
int ezclip_op_dropout(const void* x_dev, const void* residual_dev, void* y_dev, int rows, int d, float p, uint64_t seed,
                      uint32_t site, int dtype, void* stream);                    /* y = dropout(x) [+ residual] */
int ezclip_op_dropout_mask(float p, uint64_t seed, uint32_t site, int rows, int cols, uint8_t* keep_dev,
                           uint32_t* words_dev, void* stream);                    /* either output may be NULL */
/* One query per sample (the CLS row of a tower's last block): q_cls [batch, q_stride] holds the queries (heads
 * side by side), k / v as in ezclip_op_attention, ctx_cls / dctx_cls [batch, ctx_stride].  The backward writes dk / dv for every
 * key and dq either as row 0 of each sample in the full block `dq_dev` (other rows zeroed) or, when dq_cls_dev is given, into
 * that compact [batch, dq_stride] buffer. */
int ezclip_op_attention_cls(const void* q_cls_dev, int64_t q_stride, const void* k_dev, const void* v_dev, int64_t row_stride,
                            const float* key_bias_dev, void* ctx_cls_dev, int64_t ctx_stride, int batch, int seq_len, int heads,
                            int dtype, void* stream);
int ezclip_op_attention_cls_bwd(const void* q_cls_dev, int64_t q_stride, const void* k_dev, const void* v_dev, int64_t row_stride,
                                const float* key_bias_dev, const void* ctx_cls_dev, const void* dctx_cls_dev, int64_t ctx_stride,
                                void* dq_dev, void* dk_dev, void* dv_dev, void* dq_cls_dev, int64_t dq_stride, int batch,
                                int seq_len, int heads, int dtype, void* stream);
/* ---- ModifiedResNet tower, training path: the row-matrix steps around the convolutions (reference: nn.BatchNorm2d in train() mode and
 * autograd through Bottleneck / ModifiedResNet, modelzoo/models/clip/modeling_chineseclip.py:27-74,110-167; the tests' CPU restatement
 * walks the same steps: train_step_grads_by_steps).  Activations are NHWC rows [rows = B*H*W, cp] in `dtype`, channels padded to a
 * multiple of 64 with exact zeros (kept zero by every call).  scratch_dev: ezclip_op_rn_bn_scratch_bytes(rows, cp) bytes.
 *   bn_train_fwd   mean / rstd [cp] of z over the rows (biased variance), running statistics moved by `momentum` with the unbiased
 *                  variance (NULL: not touched), y = [relu](z * gamma * rstd + (beta - mean * gamma * rstd) [+ residual])
 *   bn_train_bwd   g = dy * [y > 0] (y NULL: no mask); dgamma = sum g xhat, dbeta = sum g (written or added); dz = gamma rstd (g - mean(g)
 *                  - xhat mean(g xhat)); dres (optional) = g
 *   avgpool2_bwd   dx [b, h, w, cp] = dy [b, h/2, w/2, cp] / 4 broadcast
 *   im2col3x3      col [rows, 9 * cp], column block (ky*3 + kx) = the pixel shifted by (ky - 1, kx - 1), zeros outside
 *   pack_conv_dgrad  dst [ipad][k*k*opad]: dst[c][(ky*k + kx)*opad + o] = w[o][c][k-1-ky][k-1-kx] -- the weights whose (implicit) convolution
 *                  with dz is the input gradient
 *   unpack_wgrad   dw [o, i, k, k] (+)= dwp[o][(ky*k + kx)*cp + c]   (dwp = dz^T . im2col(x) from ezclip_op_gemm_tn, row stride ldp)
 *   conv3x3_nhwc   the tower's implicit 3x3 convolution as an operator: out [rows, n] = conv(x; wp [n][9*cp]); zero256_dev: 256 zero bytes */
size_t ezclip_op_rn_bn_scratch_bytes(int64_t rows, int cp);
int ezclip_op_rn_bn_train_fwd(const void* z_dev, int64_t rows, int c, int cp, const float* gamma_dev, const float* beta_dev,
                              float* running_mean_dev, float* running_var_dev, float momentum, float eps, const void* residual_dev,
                              int relu, void* y_dev, float* mean_dev, float* rstd_dev, float* scratch_dev, int dtype, void* stream);
int ezclip_op_rn_bn_train_bwd(const void* dy_dev, const void* y_dev, const void* z_dev, int64_t rows, int c, int cp,
                              const float* gamma_dev, const float* mean_dev, const float* rstd_dev, void* dz_dev, void* dres_dev,
                              float* dgamma_dev, float* dbeta_dev, int accumulate, float* scratch_dev, int dtype, void* stream);
int ezclip_op_rn_avgpool2_bwd(const void* dy_dev, int b, int h, int w, int cp, void* dx_dev, int dtype, void* stream);
int ezclip_op_rn_im2col3x3(const void* x_dev, int b, int h, int w, int cp, void* col_dev, int dtype, void* stream);
int ezclip_op_rn_pack_conv_dgrad(const float* w_dev, int o, int i, int k, int opad, int ipad, void* dst_dev, int dtype, void* stream);
int ezclip_op_rn_unpack_wgrad(const float* dwp_dev, int64_t ldp, int o, int i, int k, int cp, int accumulate, float* dw_dev, void* stream);
int ezclip_op_conv3x3_nhwc(const void* x_dev, int b, int h, int w, int cp, const void* wp_dev, int n, void* out_dev,
                           const void* zero256_dev, int dtype, void* stream);
int ezclip_op_cast_from_f32(const float* src_dev, void* dst_dev, int64_t n, int dtype, void* stream);
int ezclip_op_cast_to_f32(const void* src_dev, float* dst_dev, int64_t n, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EZCLIP_H_ */
