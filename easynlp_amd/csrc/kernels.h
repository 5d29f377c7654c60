// Internal launcher declarations for the ezclip HIP kernels (not the C ABI;
// see include/ezclip.h for that).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dropout.h"

#define EZCLIP_F32 0
#define EZCLIP_BF16 1

namespace ezclip {

inline int dtype_size(int dtype) { return dtype == EZCLIP_BF16 ? 2 : 4; }

// ---- GEMM (gemm.hip) --------------------------------------------------------
struct GemmArgs {
  const void* A = nullptr; int64_t lda = 0;   // [M, K], dtype T
  const void* B = nullptr; int64_t ldb = 0;   // [N, K], dtype T
  void* C = nullptr; int64_t ldc = 0;         // [M, N], dtype T (or f32 if out_f32)
  void* C2 = nullptr;                         // optional second output: value before act/residual (ldc)
  const float* bias = nullptr;                // [N] f32
  const void* R = nullptr; int64_t ldr = 0;   // optional residual [M, N], dtype T
  const void* U = nullptr; int64_t ldu = 0;   // optional pre-activation [M, N] (T): result *= act'(U) (backward)
  const float* scale_log = nullptr;           // optional device scalar s: acc *= exp(s)
  // Folded LayerNorm (8-phase bf16 kernel only): A holds the RAW rows x, B the weights pre-multiplied by the LN gain
  // (W' = W o g), and the epilogue finishes  LN(x) W^T + bias = rstd (x W'^T) - rstd mean c1 + c2  with
  //   ln_stats[m] = (rstd_m, -mean_m rstd_m),  c1[n] = sum_k W'[n][k],  c2[n] = sum_k b[k] W[n][k] + bias[n].
  const float* ln_stats = nullptr;            // [M][2] f32
  const float* ln_c1 = nullptr;               // [N] f32
  const float* ln_c2 = nullptr;               // [N] f32 (bias folded in; `bias` must be null)
  // Producer side of the same trick (8-phase kernel, with R): the epilogue also writes, for every row and every
  // 64-column slab s = n / 64, rowstat_part[m][s] = (sum, sum of squares) of the ROUNDED outputs C[m][64 s .. 64 s + 63]
  // -- N / 64 partials per row that ln_stats_finalize turns into the next LayerNorm's (rstd, -mean rstd) without
  // re-reading C.  Layout [M][N / 64][2] f32.
  float* rowstat_part = nullptr;
  // Backward only (with U): colsum[n] += sum_m C[m][n] -- the bias gradient of the Linear whose output gradient C is
  // (8-phase kernel: accumulated in the epilogue, one hardware atomic per column per wave tile; else a separate pass)
  float* colsum = nullptr;
  // Implicit 3x3 convolution, pad 1, stride 1, over an NHWC image batch (128x128 / 256x256 kernels of gemm.hip; ModifiedResNet's
  // convs, modeling_chineseclip.py:41-43,115-121): A = x [B * H * W, lda] with conv_C (<= lda) channels per pixel, output row
  // m = (b, y, x) of the same H x W grid, K = 9 * conv_C ordered k = (ky * 3 + kx) * conv_C + c -- the A tile of a K-tile is the
  // 128-byte channel slice of the pixel shifted by (ky - 1, kx - 1); taps outside the image read conv_zero (>= 128 zero bytes).
  int conv_H = 0, conv_W = 0, conv_C = 0;
  const void* conv_zero = nullptr;
  float alpha = 1.0f;                         // acc *= alpha
  int M = 0, N = 0, K = 0;
  int act = 0;                                // ezclip::Act
  int out_f32 = 0;                            // bf16 inputs, f32 output
  int vec_ok = 0;                             // (set by the launcher)
  int raster_gm = 0;                          // (set by the 8-phase launcher) tile order: 0 n-fastest; 0 < g < 100: super-rows of g row tiles, m-fastest inside; 100 + w: super-columns of w tile columns
};
// Fused retrieval ranks (f32, 128x128 kernel; CLIPEvaluator's sort loop, appzoo/clip/evaluator.py:47-67): the similarity tile is
// compared in registers, nothing is written to C.  Query row m of this call is query i = rank_row0 + m of the whole set, its paired
// gallery item is column i.  (A struct of its own: GemmArgs is the by-value kernel argument of the 8-phase kernels, whose register
// allocation sits at the limit -- two of their instantiations spilled when these fields lived there.)
//   rank_mode 1: rank_diag_out[i] = S[m][i]                        (workgroups whose tile misses the diagonal exit at once)
//   rank_mode 2: rank_rows[m] += #{n : S[m][n] > d_i or (S[m][n] == d_i and n < i)},  d = rank_diag      (text -> image)
//                rank_cols[n] += #{m : S[m][n] > d_n or (S[m][n] == d_n and i < n)}   (optional: image -> text, summed over calls)
struct GemmRankArgs : GemmArgs {
  int rank_mode = 0, rank_row0 = 0;
  const float* rank_diag = nullptr;
  float* rank_diag_out = nullptr;
  int32_t* rank_rows = nullptr;
  int32_t* rank_cols = nullptr;
};
int gemm_nt_rank(GemmRankArgs p, hipStream_t stream);
// ---- ModifiedResNet training path (resnet_train.hip): row-matrix kernels on NHWC activations [rows, cp]
size_t rn_bn_scratch_bytes(int64_t rows, int cp);
int rn_bn_train_fwd(const void* z, int64_t rows, int C, int cp, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, const void* residual, int relu, void* y, float* mean, float* rstd,
                    float* scratch, int dtype, hipStream_t st);
int rn_bn_train_bwd(const void* dy, const void* y, const void* z, int64_t rows, int C, int cp, const float* gamma, const float* mean,
                    const float* rstd, void* dz, void* dres, float* dgamma, float* dbeta, int accumulate, float* scratch, int dtype,
                    hipStream_t st);
int rn_avgpool2_bwd(const void* dy, int B, int H, int W, int cp, void* dx, int dtype, hipStream_t st);
int rn_add_inplace(void* a, const void* b, int64_t n, int dtype, hipStream_t st);
int rn_im2col3x3(const void* x, int B, int H, int W, int cp, void* col, int dtype, hipStream_t st);
int rn_pack_conv_dgrad(const float* W, int O, int I, int k, int opad, int ipad, void* dst, int dtype, hipStream_t st);
int rn_unpack_wgrad(const float* dwp, int64_t ldp, int O, int I, int k, int cp, int accumulate, float* dw, hipStream_t st);
// 3 x 3 weight gradient at 64 or 128 padded channels in / out, bf16 (the stem's conv2 / conv3, layer1's and layer2's conv2): out
// [opad][ldo >= 9 cp] f32 in the column order of rn_im2col3x3; every 64-output x 64-channel sub-problem reads its halves of x and dz once;
// scratch holds one [64][576] f32 partial per workgroup (up to 512 in all)
// C[N][K] f32 (+)= A[M][N]^T . B[M][K], bf16, M >= 4096 and N x K small (N, K multiples of 64, at most 16 blocks of 64 x 256): the 1 x 1
// weight gradients of layer1 / layer2 and the stem's first convolution; operands read once per block row / column, fixed-order partial sums
bool rn_tn_skinny_eligible(int64_t M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int dtype, size_t scratch_bytes);
int rn_tn_skinny(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K, int accumulate,
                 void* scratch, size_t scratch_bytes, hipStream_t st);
bool rn_wgrad3x3_c64_eligible(int B, int H, int W, int cp, int opad, int dtype, size_t scratch_bytes);
int rn_wgrad3x3_c64(const void* x, const void* dz, int B, int H, int W, int cp, int opad, void* scratch, size_t scratch_bytes, float* out,
                    int64_t ldo, int accumulate, hipStream_t st);
// C = act(alpha * exp(scale) * A.B^T + bias) + R
int gemm_nt(GemmArgs p, int dtype, hipStream_t stream);
// 256x256x64 8-phase bf16 kernel (gemm8p.hip): large M, N % 256 == 0, K % 128 == 0, bf16 output
bool gemm_nt_uses_8p(const GemmArgs& p, int dtype);
bool gemm_nt_8p_eligible(const GemmArgs& p, int dtype);
int gemm_nt_8p(const GemmArgs& p, hipStream_t stream);
void set_gemm_variant(int v);   // debugging / sweeps: -1 heuristic, 0 = 128x128 tile, 1 = 256x256 tile
void set_gemm_raster(int gm);   // tile order of the persistent 8-phase kernel (GemmArgs::raster_gm); -1: built-in default
void set_gemm_dephase(int v);   // staggered first tiles of the persistent 8-phase kernel (gemm8p.hip); 0 off, -1 built-in default

// C[N,K] (+)= A[M,N]^T . B[M,K]   (weight gradients; contraction over rows)
struct GemmTNArgs {
  const void* A = nullptr; int64_t lda = 0;   // [M, N], dtype T
  const void* B = nullptr; int64_t ldb = 0;   // [M, K], dtype T
  float* C = nullptr; int64_t ldc = 0;        // [N, K] f32
  int M = 0, N = 0, K = 0;
  int accumulate = 0;                         // C += instead of C =
  // conv_H > 0: B is the NHWC activation x [M, conv_C] (ldb == conv_C) and the product runs over its 3 x 3 neighbourhoods,
  // K == 9 * conv_C, column (ky * 3 + kx) * conv_C + c = the pixel shifted by (ky - 1, kx - 1), zero outside the image --
  // the weight gradient of a 3 x 3 convolution without the explicit im2col (generic kernel only)
  int conv_H = 0, conv_W = 0, conv_C = 0;
};
int gemm_tn(GemmTNArgs p, int dtype, hipStream_t stream);
// 256x256 8-phase bf16 weight-gradient kernel (gemm8p.hip): N % 256 == 0, K % 256 == 0, large M; split partials +
// fixed-order reduction (bit-reproducible)
bool gemm_tn_8p_eligible(const GemmTNArgs& p, int dtype);
int gemm_tn_8p(const GemmTNArgs& p, hipStream_t stream);

// ---- attention (attention.hip) ---------------------------------------------
struct AttnArgs {
  const void* q = nullptr;        // element (b, t, h, d) at q + ((b*L + t)*row_stride + h*64 + d)
  const void* k = nullptr;
  const void* v = nullptr;
  int64_t row_stride = 0;         // in elements
  void* ctx = nullptr;            // [B*L, ctx_stride] heads merged
  int64_t ctx_stride = 0;
  const float* key_bias = nullptr;  // optional [B*L] additive key bias (BERT: 0 / -10000)
  float* lse = nullptr;           // optional [B, H, L] log-sum-exp of the scaled scores
  int B = 0, L = 0, H = 0;
  float scale = 0.125f;
  // Packed (variable-length) batches: sample b owns rows cu[b] .. cu[b] + lens[b] - 1 of q / k / v / ctx / key_bias instead
  // of rows b*L .. b*L + L - 1; L is then the LONGEST sample (LDS sizing).  Short forward kernel and the CLS kernel only.
  const int* cu = nullptr;
  const int* lens = nullptr;
  int causal = 0;                 // 1: key j > query i is masked (-inf): CLIP text transformer, build_attention_mask (modeling_openclip.py:343-349)
  DropCfg drop;                   // dropout on the probabilities (BERT train mode); row = (b*H + h)*L + query, col = key
  // Short-sequence kernels with dropout (train mode, also on packed batches whose samples are kept PREFIXES, so that a packed
  // position is the padded one): the forward writes the keep bits, 32 keys per word, keep_bits[(row * H + h) * keep_words + tile]
  // with row = the query's row in q (packed or padded), and the fused backward reads them back instead of running Philox again in
  // its key-major pass.  drop_L: the padded sequence length the mask rows are numbered with (0: L).
  uint32_t* keep_bits = nullptr;
  int keep_words = 0;
  int drop_L = 0;
  int short_tail = 1;             // (set by attention_fwd_short) short epilogue for a last key tile of <= 8 keys
  int mfma_rowsum = 1;            // (set by attention_fwd_short) plain kernel: row sums of P as an MFMA product with ones
  int fullline_store = 0;         // (set by attention_fwd_short) output rows leave as full 128-byte lines through the dead K image
};
int attention_fwd(const AttnArgs& a, int dtype, hipStream_t stream);
// one query per sample (the CLS row of the last block): q_cls [B, q_stride], ctx_cls [B, ctx_stride]; k / v / key_bias of `a`
int attention_cls_fwd(const AttnArgs& a, const void* q_cls, int64_t q_stride, void* ctx_cls, int64_t ctx_stride, int dtype,
                      hipStream_t stream);
// short-sequence bf16 kernels (attention_short.hip): L <= 256, one-pass online softmax, LDS transpose reads
bool attention_short_eligible(const AttnArgs& a, int dtype);
bool attention_short_fwd_eligible(const AttnArgs& a, int dtype);
int attention_fwd_short(const AttnArgs& a, hipStream_t stream);
void set_attention_variant(int v);   // -1 auto, 0: attention.hip kernels only
void set_attention_short_tail(int on);
void set_attention_bwd_once(int mode);  // fused backward: 1 (default) score-tile-once up to 128 tokens, 2 wherever eligible (<= 256), 0 two-pass only
void set_rn_buffer_bound_mib(int mib);   // ModifiedResNet tower: bound of one activation buffer (resnet.hip)

struct AttnBwdArgs {
  AttnArgs f;                     // forward tensors (q, k, v, key_bias, lse; ctx = forward output)
  const void* dctx = nullptr;     // [B*L, ctx_stride]
  void* dq = nullptr;             // same addressing as q/k/v (row_stride)
  void* dk = nullptr;
  void* dv = nullptr;
  // optional: gradients of the q / k / v projection biases, [H*64] f32 each, ACCUMULATED: db[h*64 + d] += sum over
  // (batch, token) of dq/dk/dv -- the fused kernel reduces them on the way out instead of a separate pass over dqkv
  float* dbq = nullptr;
  float* dbk = nullptr;
  float* dbv = nullptr;
  float* db_part = nullptr;       // scratch [B][3][H*64] f32 for the fused kernel's per-sample partial sums (required with dbq)
};
int attention_bwd(const AttnBwdArgs& a, int dtype, hipStream_t stream);
// backward of attention_cls_fwd: writes the full dq / dk / dv blocks (dq: the query row, zeros elsewhere)
// dq_cls != null: dq goes to that compact [B, dq_stride] buffer instead of (row 0 + zero fill of) a.dq
int attention_cls_bwd(const AttnBwdArgs& a, const void* q_cls, int64_t q_stride, const void* ctx_cls, const void* dctx_cls,
                      int64_t ctx_stride, int dtype, hipStream_t stream, void* dq_cls = nullptr, int64_t dq_stride = 0);
int attention_bwd_short(const AttnBwdArgs& a, hipStream_t stream);   // fused single kernel, bf16, L <= 256

// ---- row-wise / elementwise kernels (rowops.hip) ------------------------------
// y = LN(x) * g + b over the last dim D; rows may be strided (x_stride/y_stride in elements).
// Optional: mean/rstd [rows] for the backward pass.
int layernorm_fwd(const void* x, int64_t x_stride, void* y, int64_t y_stride, const float* g,
                  const float* b, float eps, int rows, int D, int dtype, float* mean, float* rstd,
                  hipStream_t stream);
// dx = LN backward; dg/db accumulate (+=) into f32 [D] via atomics.  If dres != nullptr the result is
// dx = dres + LNbwd(dy) (residual branch merge).  dsum (optional, [D] f32) += column sums of the dx written
// (= the bias gradient of the Linear whose output gradient dx is).
int layernorm_bwd(const void* x, int64_t x_stride, const void* dy, int64_t dy_stride, const float* g,
                  const float* mean, const float* rstd, void* dx, int64_t dx_stride, const void* dres,
                  int64_t dres_stride, float* dg, float* db, int rows, int D, int dtype, hipStream_t stream,
                  float* dsum = nullptr);

// stats[row] = (rstd, -mean * rstd) of LayerNorm over the last dim (consumed by the GEMM's folded-LN epilogue)
int layernorm_row_stats(const void* x, int64_t x_stride, float eps, int rows, int D, int dtype, float* stats,
                        hipStream_t stream);
// W [N, K] f32, LN gain g / shift b [K], bias [N] (may be null)  ->  Wf [N, ldk] (compute dtype) = W o g,
// c1[n] = sum_k Wf[n][k] (of the ROUNDED values), c2[n] = sum_k b[k] W[n][k] + bias[n]
int fold_ln_weight(const float* W, const float* g, const float* b, const float* bias, int N, int K, void* Wf, int64_t ldk,
                   float* c1, float* c2, int dtype, hipStream_t stream);
int cast_from_f32(const float* src, void* dst, int64_t n, int dtype, hipStream_t stream);
int cast_to_f32(const void* src, float* dst, int64_t n, int dtype, hipStream_t stream);
// dst[c][r] = src[r][c]; src [R, C] f32 (row stride src_ld) -> dst [C, ld] T (ld >= R, pad zero-filled)
int transpose_cast(const float* src, int64_t src_ld, int R, int C, void* dst, int64_t ld, int dtype, hipStream_t stream);
// src [R, C] f32 -> dst [R, ld] T with zero padding of columns C..ld-1
int pad_cast(const float* src, int R, int C, void* dst, int64_t ld, int dtype, hipStream_t stream);
// Weight re-packing in one launch (ezclip_refresh_weights runs after every optimizer step: ~250 separate pad_cast /
// transpose_cast launches of 5 us each were launch-bound, 1.3 ms per step).  A job reads one f32 matrix src [R][C] and
// writes a straight copy dst_s [R][ld_s] and / or a transposed copy dst_t [C][ld_t] in the compute dtype, pad columns zeroed;
// kind 1: dst_s = src as f32, C elements (bias vectors gathered next to packed weights).  `tile0`: first 64 x 64 tile of the
// job in the launch (prefix sum, filled by cast_jobs_finalize); jobs live in device memory.
struct CastJob {
  const float* src = nullptr;
  void* dst_s = nullptr;
  void* dst_t = nullptr;
  int R = 0, C = 0;
  int ld_s = 0, ld_t = 0;
  int kind = 0;
  int tiles_c = 0;
  int tile0 = 0;
  int pad_ = 0;
};
int cast_jobs_finalize(CastJob* jobs_host, int n);                        // fills tiles_c / tile0; returns the total tile count
int cast_jobs_run(const CastJob* jobs_dev, int n, int total_tiles, int dtype, hipStream_t stream);

// pixels [B,3,R,R] f32 NCHW -> patches [B*G*G, Kpad] T, inner index (c, ky, kx); cols >= 3*P*P zero.
int im2col_patches(const float* pixels, void* out, int B, int R, int P, int Kpad, int dtype, hipStream_t stream);
// x0[b, 0] = cls + pos[0]; x0[b, 1+p] = patch[b*G2 + p] + pos[1+p]; y = LN(x0).  x0 optional (saved for bwd).
int vit_assemble_ln(const void* patch, const float* cls, const float* pos, const float* g, const float* b,
                    float eps, void* x0, void* y, float* mean, float* rstd, int B, int Lv, int W, int dtype,
                    hipStream_t stream);
// BERT embeddings (word + type[0] + pos[t]) -> LN; also key_bias[b*L+t] = ids==0 ? -10000 : 0.
int bert_embed_ln(const int64_t* ids, const float* word, const float* pos, const float* type, const float* g,
                  const float* b, float eps, void* x0, void* y, float* mean, float* rstd, float* key_bias,
                  int B, int L, int Hd, int vocab, int dtype, hipStream_t stream, const int64_t* pos_ids = nullptr,
                  const int64_t* type_ids = nullptr, const int64_t* attn_mask = nullptr, int max_pos = 1 << 30,
                  int type_vocab = 1, const int* rowmap = nullptr, int packed_rows = 0);
// (rowmap != null: only the packed_rows tokens rowmap[r] = b * L + t are embedded, into rows r of y / key_bias)
// CLIP text transformer (OPEN_CLIP.encode_text): x = token_embedding[ids] + positional_embedding; eot_idx[b] = argmax_t ids[b, t]
int clip_text_embed(const int64_t* ids, const float* tok, const float* pos, void* x, int* eot_idx, int B, int L, int W, int vocab,
                    int64_t eot_id, int dtype, hipStream_t stream);
// mode 0: dst[b] = src[b, idx[b]];  1: dst[b, idx[b]] = src[b];  2: dst[b, idx[b]] += src[b]  (other rows untouched; idx null: row 0)
int gather_rows(const void* src, const int* idx, void* dst, int B, int L, int W, int mode, int dtype, hipStream_t stream);
// out[b] = x[b] / ||x[b]||_2 (no eps: reference modeling_chineseclip.py:360,363); inv_norm optional.
int l2_normalize_fwd(const float* x, float* out, float* inv_norm, int B, int E, hipStream_t stream);
// dx = (dy - y * <dy, y>) * inv_norm
int l2_normalize_bwd(const float* y, const float* dy, const float* inv_norm, float* dx, int B, int E,
                     hipStream_t stream);

// backward helpers
int colsum_add(const void* x, int64_t ld, int rows, int cols, float* out, int dtype, hipStream_t stream);      // out[c] += sum_r x[r][c]
int colsum3_add(const void* x, int64_t ld, int rows, int cols, float* out, float* out1, float* out2, int seg, int dtype,
                hipStream_t stream);      // the same into three vectors of `seg` columns each (cols = 3 seg)
// out[t][c] += sum_b x[b][t][c] for t < t_count (x is [B, Tn, W])
int batch_sum_add(const void* x, int B, int Tn, int t_count, int W, float* out, int dtype, hipStream_t stream);
int vit_gather_patch_rows(const void* dx0, void* dpemb, int B, int Lv, int W, int dtype, hipStream_t stream);
// table[ids[r]] += dx0[r] (embedding index-add); rows whose id == pad_id (nn.Embedding padding_idx) or out of range are skipped
// rowmap (packed rows): row r holds token rowmap[r] = b * seq_len + t, ids is read there; ids == nullptr with a row map: the
// table index is the position t (position-embedding gradient of a packed batch)
int bert_word_grad(const int64_t* ids, const void* dx0, float* dword, int64_t rows, int Hd, int vocab, int dtype,
                   hipStream_t stream, int64_t pad_id = 0, const int* rowmap = nullptr, int seq_len = 0);
int add_inplace_f32(float* dst, const float* src, int64_t n, hipStream_t stream);
// dst[r][c] += src[r][c] for c < cols (row strides ldd / lds): un-pads a K-padded weight gradient
int dropout_rows(const void* x, int64_t xs, const void* res, int64_t rs, void* y, int64_t ys, int rows, int D,
                 const DropCfg& d, int dtype, hipStream_t stream, const int* rowmap = nullptr);
int dropout_mask(uint8_t* keep, uint32_t* words, int rows, int cols, const DropCfg& d, hipStream_t stream);
int layernorm_stats_finalize(const float* part, int slabs, int D, float eps, int rows, float* stats, hipStream_t stream);
int add_cols_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int rows, int cols, hipStream_t stream);

// ---- InfoNCE (loss.hip) -----------------------------------------------------------
// Row-wise cross entropy against the diagonal: for local row i (global column diag0 + i):
//   lse[i] = logsumexp_j S[i][j];  row_loss[i] = lse[i] - S[i][diag0 + i]
int ce_rows_fwd(const float* S, int64_t ld, int rows, int cols, int diag0, float* lse, float* row_loss,
                hipStream_t stream);
// dS[i][j] (+)= coef * (*coef_dev) * (exp(S[i][j] - lse[i]) - [j == diag0 + i])   (coef_dev optional)
int ce_rows_bwd(const float* S, int64_t ld, int rows, int cols, int diag0, const float* lse, const float* coef_dev,
                float coef, float* dS, int64_t ldd, int accumulate, hipStream_t stream);
// out (+)= scale * sum(x), fixed summation order
int sum_scaled(const float* x, int n, float scale, float* out, int accumulate, hipStream_t stream);
// out (+)= scale * <a, b>; partial: scratch of >= 256 floats
int dot_scaled(const float* a, const float* b, int64_t n, float scale, float* partial, float* out, int accumulate,
               hipStream_t stream);

int ce_cols_fwd(const float* S, int64_t ld, int n, float* lse, float* col_loss, hipStream_t stream);
int infonce_dlogits(const float* S, int n, const float* lse_r, const float* lse_c, const float* g, float coef, float* dS,
                    hipStream_t stream);
// sim [rows, n] = queries row0 .. row0 + rows - 1 against all n gallery items; rank[r] for query row0 + r
int recall_ranks(const float* sim, int rows, int n, int row0, int32_t* rank, hipStream_t stream);
// packing metadata of a text batch in one launch; (rows, longest, prefix, ticket) land in pinned host memory (packmeta.hip)
int pack_text_meta(const int64_t* ids, const int64_t* mask, int B, int S, int* rowmap, int* cu, int* lens, int* host_out_dev,
                   int ticket, hipStream_t stream);
// Tiled InfoNCE on embeddings (nce.hip): no [n, N] buffer; split = 1: operands as bf16 hi + lo (float32-class), 0: bf16
bool infonce_tiled_eligible(int e);
size_t infonce_tiled_workspace_bytes(int n, int N, int e);
int infonce_tiled(const float* T, const float* I, int n, int N, int off, int e, const float* ls, float grad_scale, int split,
                  float* loss, float* dT, float* dI, float* dls, void* ws, size_t ws_bytes, hipStream_t stream);
const char* last_error();

// ---- optional per-launch timing (profile.hip) --------------------------------------
enum ProfClass { PROF_GEMM = 0, PROF_ATTN = 1, PROF_ROWOP = 2 };
// GPU image pre-processing (preprocess.hip); ezclip_image_desc is declared in include/ezclip.h
}  // namespace ezclip
struct ezclip_image_desc;
namespace ezclip {
size_t preprocess_workspace_bytes(const ::ezclip_image_desc* desc, int n, int size, int crop);
int resample_table(int in_size, int out_size, int first, int count, int* ksize, int* bounds, int* kk, int kk_capacity);
int resample_table_device(int in_size, int out_size, int first, int count, int* bounds_dev, int* kk_dev, hipStream_t stream);
void set_device_resample_tables(int on);
int preprocess_images(const uint8_t* packed, const ::ezclip_image_desc* desc, int n, int size, int crop, const float* mean,
                      const float* stdv, float* out, void* ws, size_t ws_bytes, hipStream_t stream);

struct ProfScope {   // RAII: records HIP events around the launches issued in its lifetime
  ProfScope(int cls, double work, hipStream_t stream);
  ~ProfScope();
  hipStream_t stream_;
  int idx_;
};
int profile_begin();
int profile_end(int cls, double* ms, double* work, int* launches);

}  // namespace ezclip
