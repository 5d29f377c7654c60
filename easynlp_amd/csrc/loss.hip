// InfoNCE pieces (reference: CLIPApp.contrastive_loss / clip_loss,
// easynlp/appzoo/clip/model.py:154-160 -- F.cross_entropy(S, arange(N)) on S and
// S^T, averaged).  Rows of the logits are HBM-streamed once: one 256-thread
// workgroup per row, wave-shuffle + LDS two-level reductions, fixed-order final
// sum (no float atomics: the loss is bit-reproducible run to run).
#include "ezclip_common.h"
#include "kernels.h"

namespace ezclip {
namespace {

__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r += red[i];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void ce_rows_fwd_kernel(const float* S, int64_t ld, int rows, int cols, int diag0,
                                                           float* lse, float* row_loss) {
  __shared__ float red[4];
  const int i = blockIdx.x;
  const float* s = S + (int64_t)i * ld;
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < cols; j += blockDim.x) mx = fmaxf(mx, s[j]);
  mx = block_max(mx, red);
  float sum = 0.f;
  for (int j = threadIdx.x; j < cols; j += blockDim.x) sum += expf(s[j] - mx);
  sum = block_sum(sum, red);
  if (threadIdx.x == 0) {
    const float l = mx + logf(sum);
    lse[i] = l;
    row_loss[i] = l - s[diag0 + i];
  }
}

__global__ __launch_bounds__(256) void ce_rows_bwd_kernel(const float* S, int64_t ld, int rows, int cols, int diag0,
                                                           const float* lse, const float* coef_p, float coef,
                                                           float* dS, int64_t ldd, int accumulate) {
  const int i = blockIdx.x;
  const float* s = S + (int64_t)i * ld;
  float* d = dS + (int64_t)i * ldd;
  const float l = lse[i];
  const float c = coef * (coef_p ? *coef_p : 1.0f);
  for (int j = threadIdx.x; j < cols; j += blockDim.x) {
    float g = c * (expf(s[j] - l) - (j == diag0 + i ? 1.0f : 0.0f));
    if (accumulate) g += d[j];
    d[j] = g;
  }
}

// out (+)= scale * sum(x[0..n)) in a fixed order
__global__ __launch_bounds__(256) void sum_scaled_kernel(const float* x, int n, float scale, float* out, int accumulate) {
  __shared__ float red[4];
  float s = 0.f;
  for (int j = threadIdx.x; j < n; j += blockDim.x) s += x[j];
  s = block_sum(s, red);
  if (threadIdx.x == 0) *out = (accumulate ? *out : 0.f) + scale * s;
}

// out (+)= scale * sum_i a[i] * b[i]   (n up to a few 10^7: d(logit_scale) = sum dS .* S)
__global__ __launch_bounds__(256) void dot_partial_kernel(const float* a, const float* b, int64_t n, float* partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
    s += a[j] * b[j];
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

}  // namespace

int ce_rows_fwd(const float* S, int64_t ld, int rows, int cols, int diag0, float* lse, float* row_loss,
                hipStream_t stream) {
  EZ_REQUIRE(rows > 0 && cols > 0 && diag0 >= 0 && diag0 + rows <= cols, "ce_rows_fwd: bad shape rows=%d cols=%d diag0=%d", rows, cols, diag0);
  hipLaunchKernelGGL(ce_rows_fwd_kernel, dim3(rows), dim3(256), 0, stream, S, ld, rows, cols, diag0, lse, row_loss);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int ce_rows_bwd(const float* S, int64_t ld, int rows, int cols, int diag0, const float* lse, const float* coef_dev,
                float coef, float* dS, int64_t ldd, int accumulate, hipStream_t stream) {
  EZ_REQUIRE(rows > 0 && cols > 0, "ce_rows_bwd: bad shape");
  hipLaunchKernelGGL(ce_rows_bwd_kernel, dim3(rows), dim3(256), 0, stream, S, ld, rows, cols, diag0, lse, coef_dev,
                     coef, dS, ldd, accumulate);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int sum_scaled(const float* x, int n, float scale, float* out, int accumulate, hipStream_t stream) {
  hipLaunchKernelGGL(sum_scaled_kernel, dim3(1), dim3(256), 0, stream, x, n, scale, out, accumulate);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

int dot_scaled(const float* a, const float* b, int64_t n, float scale, float* partial /* >= 256 floats */, float* out,
               int accumulate, hipStream_t stream) {
  int blocks = (int)((n + 256 * 16 - 1) / (256 * 16));
  if (blocks < 1) blocks = 1;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(dot_partial_kernel, dim3(blocks), dim3(256), 0, stream, a, b, n, partial);
  hipLaunchKernelGGL(sum_scaled_kernel, dim3(1), dim3(256), 0, stream, partial, blocks, scale, out, accumulate);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace ezclip

// ---- column direction + fused d(logits) for a materialised square logits matrix ----
namespace ezclip {
namespace {

// lse over rows for each column j (coalesced: lane = column); 4 waves split the rows
__global__ __launch_bounds__(256) void ce_cols_fwd_kernel(const float* S, int64_t ld, int n, float* lse, float* col_loss) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  float mx = -INFINITY;
  if (j < n) for (int i = w; i < n; i += 4) mx = fmaxf(mx, S[(int64_t)i * ld + j]);
  red[w][lane] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0][lane], red[1][lane]), fmaxf(red[2][lane], red[3][lane]));
  __syncthreads();
  float sum = 0.f;
  if (j < n) for (int i = w; i < n; i += 4) sum += expf(S[(int64_t)i * ld + j] - mx);
  red[w][lane] = sum;
  __syncthreads();
  if (w == 0 && j < n) {
    sum = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const float l = mx + logf(sum);
    lse[j] = l;
    col_loss[j] = l - S[(int64_t)j * ld + j];
  }
}

// dS[i][j] = c * (exp(S - lse_r[i]) + exp(S - lse_c[j]) - 2 [i == j]),  c = coef * *g
__global__ __launch_bounds__(256) void infonce_dlogits_kernel(const float* S, int n, const float* lse_r, const float* lse_c,
                                                               const float* g, float coef, float* dS) {
  const int i = blockIdx.x;
  const float c = coef * (g ? *g : 1.0f);
  const float lr = lse_r[i];
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const float s = S[(int64_t)i * n + j];
    dS[(int64_t)i * n + j] = c * (expf(s - lr) + expf(s - lse_c[j]) - (i == j ? 2.0f : 0.0f));
  }
}

// query row r of this block of rows is query i = row0 + r of the whole set:
// rank[r] = #{j : sim[r][j] > sim[r][i]} + #{j < i : sim[r][j] == sim[r][i]}        (sim: [rows, n])
__global__ __launch_bounds__(256) void recall_rank_kernel(const float* sim, int n, int row0, int32_t* rank) {
  __shared__ float red[4];
  const int r = blockIdx.x, i = row0 + r;
  const float d = sim[(int64_t)r * n + i];
  float c = 0.f;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const float s = sim[(int64_t)r * n + j];
    c += (s > d || (s == d && j < i)) ? 1.f : 0.f;
  }
  c = block_sum(c, red);
  if (threadIdx.x == 0) rank[r] = (int32_t)c;
}

}  // namespace

int ce_cols_fwd(const float* S, int64_t ld, int n, float* lse, float* col_loss, hipStream_t stream) {
  hipLaunchKernelGGL(ce_cols_fwd_kernel, dim3((n + 63) / 64), dim3(256), 0, stream, S, ld, n, lse, col_loss);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}
int infonce_dlogits(const float* S, int n, const float* lse_r, const float* lse_c, const float* g, float coef, float* dS,
                    hipStream_t stream) {
  hipLaunchKernelGGL(infonce_dlogits_kernel, dim3(n), dim3(256), 0, stream, S, n, lse_r, lse_c, g, coef, dS);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}
int recall_ranks(const float* sim, int rows, int n, int row0, int32_t* rank, hipStream_t stream) {
  hipLaunchKernelGGL(recall_rank_kernel, dim3(rows), dim3(256), 0, stream, sim, n, row0, rank);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace ezclip
