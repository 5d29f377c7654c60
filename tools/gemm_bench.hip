// Stand-alone GEMM check + timing on the path's real shapes (no torch: starts in ~1 s on the GPU box).
//   build:  python tools/build_tools.py        (links against easynlp_amd/csrc/libezclip_hip.so)
//   run:    tools/bin/gemm_bench [batch=1024] [iters=20] [variants=0,2]
// For every shape it runs the reference variant 0 (128x128 kernel, verified against the oracle by the
// pytest suite) and the listed variants on the same uniform[-1,1) operands, reports TFLOP/s and the max
// absolute difference of the bf16 outputs (expected 0: same accumulation order, same epilogue math).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include "../easynlp_amd/csrc/kernels.h"

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                    \
    }                                                                             \
  } while (0)

namespace {

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    const float f = ((x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;   // uniform [-scale, scale)
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    p[i] = (uint16_t)(u >> 16);
  }
}
__global__ void fill_f32(float* p, size_t n, uint32_t seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
    p[i] = (x >> 8) * (1.0f / 8388608.0f) - 1.0f;
  }
}
__global__ void maxdiff_bf16(const uint16_t* a, const uint16_t* b, size_t n, float* out, unsigned long long* nbad) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float m = 0.f;
  unsigned long long bad = 0;
  for (; i < n; i += stride) {
    const float x = __uint_as_float((uint32_t)a[i] << 16), y = __uint_as_float((uint32_t)b[i] << 16);
    const float d = fabsf(x - y);
    if (!(d <= m)) m = d;   // NaN propagates
    if (a[i] != b[i]) ++bad;
  }
  atomicMax(reinterpret_cast<int*>(out), __float_as_int(m));   // non-negative floats order as ints
  if (bad) atomicAdd(nbad, bad);
}

// where do the outputs differ?  bins: [0..15] (row%128)/8, [16..23] (col%64)/8, [24..31] wave = ((row%256)/128)*4 + (col%256)/64,
// [32] tiles with a difference (approx: counts elements at tile origin rows), [33..40] row%8
__global__ void diffmap(const uint16_t* a, const uint16_t* b, int M, int N, unsigned long long* bins) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t n = (size_t)M * N, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    if (a[i] == b[i]) continue;
    const int r = (int)(i / N), c = (int)(i % N);
    atomicAdd(&bins[(r % 128) / 8], 1ull);
    atomicAdd(&bins[16 + (c % 64) / 8], 1ull);
    atomicAdd(&bins[24 + ((r % 256) / 128) * 4 + (c % 256) / 64], 1ull);
    atomicAdd(&bins[33 + r % 8], 1ull);
    atomicAdd(&bins[41 + (c % 8)], 1ull);
    if (b[i] == 0xffff) atomicAdd(&bins[32], 1ull);
  }
}

constexpr size_t kGuard = 256 * 3072;   // elements after the last row of C

struct Shape { const char* name; int M, N, K, act; bool res, c2; bool u = false; };

}  // namespace

int main(int argc, char** argv) {
  const int batch = argc > 1 ? atoi(argv[1]) : 1024;
  const bool only_attn = getenv("ONLY_ATTN") != nullptr;
  if (getenv("RASTER_GM")) ezclip::set_gemm_raster(atoi(getenv("RASTER_GM")));    // tile order of the persistent kernel
  if (getenv("GEMM_DEPHASE")) ezclip::set_gemm_dephase(atoi(getenv("GEMM_DEPHASE")));   // steps + 100 * period code (gemm8p.hip)
  if (getenv("ATTN_FWD_OPTS")) ezclip::set_attention_short_tail(atoi(getenv("ATTN_FWD_OPTS")));   // bits: 1 short tail, 2 MFMA row sums, 4 no full lines, 8 no persistent grid
  const int iters = argc > 2 ? atoi(argv[2]) : 20;
  std::vector<int> variants;
  {
    std::string v = argc > 3 ? argv[3] : "0,2";
    size_t pos = 0;
    while (pos < v.size()) {
      variants.push_back(atoi(v.c_str() + pos));
      pos = v.find(',', pos);
      if (pos == std::string::npos) break;
      ++pos;
    }
  }
  // OPERAND_SCALE=0: all-zero A / B (no datapath toggling) -- how much of the sustained rate is the socket power cap
  const float opscale = getenv("OPERAND_SCALE") ? (float)atof(getenv("OPERAND_SCALE")) : 1.0f;
  const int Mv = batch * 197, Mt = batch * 64;
  const Shape shapes[] = {
      {"vit.qkv", Mv, 2304, 768, 0, false, false},     {"vit.out+res", Mv, 768, 768, 0, true, false},
      {"vit.fc+qgelu", Mv, 3072, 768, 1, false, false}, {"vit.proj+res", Mv, 768, 3072, 0, true, false},
      {"bert.qkvo+res", Mt, 768, 768, 0, true, false},  {"bert.ffn1+gelu", Mt, 3072, 768, 2, false, false},
      {"bert.ffn2+res", Mt, 768, 3072, 0, true, false}, {"patch", batch * 196, 768, 768, 0, false, false},
      {"train.fc+c2", Mv, 3072, 768, 1, false, true},   {"ragged.M", Mv - 100, 768, 768, 0, true, false},
      {"bwd.dgrad*act'", Mv, 3072, 768, 1, false, false, true}, {"bwd.gelu' ragged", 788, 3072, 768, 2, false, false, true},
      {"small.M=788", 788, 768, 3072, 0, true, false},
      {"bert.qkv", Mt, 2304, 768, 0, false, false},       // q | k | v of a BERT layer as one product (round 2)
  };
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float* d_md;
  unsigned long long* d_bad;
  CK(hipMalloc(&d_md, 4));
  CK(hipMalloc(&d_bad, 8));
  printf("batch %d iters %d\n", batch, iters);
  const int nt_shapes = getenv("NT_SHAPES") ? atoi(getenv("NT_SHAPES")) : -1;   // run only the first n NT shapes, then exit
  int shape_no = 0;
  for (const Shape& s : shapes) {
    if (only_attn) break;
    if (nt_shapes >= 0 && shape_no++ >= nt_shapes) return 0;
    const size_t nA = (size_t)s.M * s.K, nB = (size_t)s.N * s.K, nC = (size_t)s.M * s.N;
    uint16_t *A, *B, *R = nullptr, *C0, *C1, *P0 = nullptr, *P1 = nullptr, *Uu = nullptr;
    float* bias;
    CK(hipMalloc(&A, nA * 2)); CK(hipMalloc(&B, nB * 2)); CK(hipMalloc(&C0, nC * 2)); CK(hipMalloc(&C1, (nC + kGuard) * 2));
    CK(hipMalloc(&bias, s.N * 4));
    if (s.res) CK(hipMalloc(&R, nC * 2));
    if (s.c2) { CK(hipMalloc(&P0, nC * 2)); CK(hipMalloc(&P1, nC * 2)); }
    fill_bf16<<<2048, 256, 0, st>>>(A, nA, 1u, 1.0f * opscale);
    fill_bf16<<<2048, 256, 0, st>>>(B, nB, 2u, 0.05f * opscale);
    fill_f32<<<(s.N + 255) / 256, 256, 0, st>>>(bias, s.N, 3u);
    if (R) fill_bf16<<<2048, 256, 0, st>>>(R, nC, 4u, 1.0f);
    if (s.u) { CK(hipMalloc(&Uu, nC * 2)); fill_bf16<<<2048, 256, 0, st>>>(Uu, nC, 5u, 3.0f); }
    CK(hipMemsetAsync(C0, 0xff, nC * 2, st));
    CK(hipMemsetAsync(C1, 0xff, nC * 2, st));
    CK(hipMemsetAsync(C1 + nC, 0x5a, kGuard * 2, st));   // guard: rows >= M must never be written
    printf("%-15s M=%7d N=%5d K=%5d :", s.name, s.M, s.N, s.K);
    for (size_t vi = 0; vi < variants.size(); ++vi) {
      const int v = variants[vi];
      ezclip::set_gemm_variant(v);
      ezclip::GemmArgs g;
      g.A = A; g.lda = s.K; g.B = B; g.ldb = s.K;
      g.C = vi == 0 ? C0 : C1; g.ldc = s.N;
      g.C2 = s.c2 ? (vi == 0 ? P0 : P1) : nullptr;
      g.bias = bias; g.R = R; g.ldr = s.N; g.M = s.M; g.N = s.N; g.K = s.K; g.act = s.act;
      g.U = Uu; g.ldu = s.N;
      if (ezclip::gemm_nt(g, EZCLIP_BF16, st) != 0) { printf(" v%d ERROR %s", v, ezclip::last_error()); continue; }
      CK(hipStreamSynchronize(st));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, st));
      for (int it = 0; it < iters; ++it) ezclip::gemm_nt(g, EZCLIP_BF16, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= iters;
      printf("  v%d %7.1f TF (%.3f ms)", v, 2.0 * s.M * s.N * s.K / ms / 1e9, ms);
      if (vi > 0) {
        CK(hipMemsetAsync(d_md, 0, 4, st)); CK(hipMemsetAsync(d_bad, 0, 8, st));
        maxdiff_bf16<<<1024, 256, 0, st>>>(C0, C1, nC, d_md, d_bad);
        if (s.c2) maxdiff_bf16<<<1024, 256, 0, st>>>(P0, P1, nC, d_md, d_bad);
        float md; unsigned long long bad;
        CK(hipMemcpyAsync(&md, d_md, 4, hipMemcpyDeviceToHost, st));
        CK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        printf(" [maxdiff %.3g, %llu differ]", md, bad);
        {
          std::vector<uint16_t> hg(kGuard);
          CK(hipMemcpyAsync(hg.data(), C1 + nC, kGuard * 2, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
          size_t gb = 0;
          for (uint16_t x : hg) gb += (x != 0x5a5a);
          if (gb) printf(" [GUARD VIOLATED: %zu elements past M rows written]", gb);
        }
        if (bad && getenv("DIFFMAP")) {
          unsigned long long* dbins; unsigned long long hb[49];
          CK(hipMalloc(&dbins, sizeof(hb))); CK(hipMemsetAsync(dbins, 0, sizeof(hb), st));
          diffmap<<<1024, 256, 0, st>>>(C0, C1, s.M, s.N, dbins);
          CK(hipMemcpyAsync(hb, dbins, sizeof(hb), hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
          printf("\n   by (row%%128)/8:"); for (int q = 0; q < 16; ++q) printf(" %llu", hb[q]);
          printf("\n   by (col%%64)/8:"); for (int q = 16; q < 24; ++q) printf(" %llu", hb[q]);
          printf("\n   by wave:"); for (int q = 24; q < 32; ++q) printf(" %llu", hb[q]);
          printf("\n   unwritten(0xffff): %llu", hb[32]);
          printf("\n   by row%%8:"); for (int q = 33; q < 41; ++q) printf(" %llu", hb[q]);
          printf("\n   by col%%8:"); for (int q = 41; q < 49; ++q) printf(" %llu", hb[q]);
          printf("\n");
          hipFree(dbins);
        }
      }
    }
    printf("\n");
    fflush(stdout);
    hipFree(A); hipFree(B); hipFree(C0); hipFree(C1); hipFree(bias);
    if (R) hipFree(R);
    if (Uu) hipFree(Uu);
    if (P0) { hipFree(P0); hipFree(P1); }
  }
  // ---- weight-gradient (TN) shapes: C[N,K] += A[M,N]^T . B[M,K], fp32 output --------------------------------
  {
    struct TN { const char* name; int M, N, K; };
    const TN tns[] = {{"wgrad.out", Mv, 768, 768},   {"wgrad.qkv", Mv, 2304, 768}, {"wgrad.fc", Mv, 3072, 768},
                      {"wgrad.proj", Mv, 768, 3072}, {"wgrad.bert.ffn1", Mt, 3072, 768}, {"wgrad.ragged", Mv - 77, 768, 768}};
    for (const TN& t : tns) {
      if (only_attn) break;
      const size_t nA = (size_t)t.M * t.N, nB = (size_t)t.M * t.K, nC = (size_t)t.N * t.K;
      uint16_t *A, *B; float *C0, *C1;
      std::vector<float> h0;
      CK(hipMalloc(&A, nA * 2)); CK(hipMalloc(&B, nB * 2)); CK(hipMalloc(&C0, nC * 4)); CK(hipMalloc(&C1, nC * 4));
      fill_bf16<<<2048, 256, 0, st>>>(A, nA, 11u, 1.0f);
      fill_bf16<<<2048, 256, 0, st>>>(B, nB, 12u, 1.0f);
      printf("%-15s M=%7d N=%5d K=%5d :", t.name, t.M, t.N, t.K);
      for (size_t vi = 0; vi < variants.size(); ++vi) {
        const int v = variants[vi];
        float* C = vi == 0 ? C0 : C1;
        ezclip::set_gemm_variant(v);
        ezclip::GemmTNArgs g;
        g.A = A; g.lda = t.N; g.B = B; g.ldb = t.K; g.C = C; g.ldc = t.K; g.M = t.M; g.N = t.N; g.K = t.K; g.accumulate = 1;
        fill_f32<<<(unsigned)((nC + 255) / 256), 256, 0, st>>>(C, nC, 13u);       // accumulate onto a known pattern
        if (ezclip::gemm_tn(g, EZCLIP_BF16, st) != 0) { printf(" v%d ERROR %s", v, ezclip::last_error()); continue; }
        CK(hipStreamSynchronize(st));
        if (vi == 0) { h0.resize(nC); CK(hipMemcpy(h0.data(), C0, nC * 4, hipMemcpyDeviceToHost)); }
        if (vi > 0) {
          std::vector<float> b(nC);
          const std::vector<float>& a = h0;
          CK(hipMemcpy(b.data(), C1, nC * 4, hipMemcpyDeviceToHost));
          double md = 0, mx = 0;
          for (size_t i = 0; i < nC; ++i) { md = std::max(md, (double)fabsf(a[i] - b[i])); mx = std::max(mx, (double)fabsf(a[i])); }
          printf(" [maxdiff %.3g of max %.3g]", md, mx);
        }
        ezclip::GemmTNArgs g2 = g; g2.accumulate = 0;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int it = 0; it < iters; ++it) ezclip::gemm_tn(g2, EZCLIP_BF16, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        printf("  v%d %7.1f TF (%.3f ms)", v, 2.0 * t.M * t.N * t.K / ms / 1e9, ms);
      }
      printf("\n");
      fflush(stdout);
      hipFree(A); hipFree(B); hipFree(C0); hipFree(C1);
    }
  }
  // ---- folded-LayerNorm epilogue:  C = bf16( acc * s0[m] + s1[m] * c1[n] + c2[n] )  against the fp32 product of the
  //      128x128 kernel finished on the host -------------------------------------------------------------------------
  if (!only_attn) {
    const int M = 4000, N = 768, K = 768;
    uint16_t *A, *B, *C; float *Cf, *st2, *c1, *c2;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    CK(hipMalloc(&Cf, (size_t)M * N * 4)); CK(hipMalloc(&st2, (size_t)M * 8)); CK(hipMalloc(&c1, N * 4)); CK(hipMalloc(&c2, N * 4));
    fill_bf16<<<2048, 256, 0, st>>>(A, (size_t)M * K, 31u, 1.0f);
    fill_bf16<<<2048, 256, 0, st>>>(B, (size_t)N * K, 32u, 0.05f);
    fill_f32<<<(2 * M + 255) / 256, 256, 0, st>>>(st2, 2 * M, 33u);
    fill_f32<<<(N + 255) / 256, 256, 0, st>>>(c1, N, 34u);
    fill_f32<<<(N + 255) / 256, 256, 0, st>>>(c2, N, 35u);
    ezclip::GemmArgs g;
    g.A = A; g.lda = K; g.B = B; g.ldb = K; g.C = Cf; g.ldc = N; g.M = M; g.N = N; g.K = K; g.out_f32 = 1;
    ezclip::set_gemm_variant(0);
    int rc0 = ezclip::gemm_nt(g, EZCLIP_BF16, st);
    ezclip::set_gemm_variant(2);
    g.C = C; g.out_f32 = 0; g.ln_stats = st2; g.ln_c1 = c1; g.ln_c2 = c2;
    int rc1 = ezclip::gemm_nt(g, EZCLIP_BF16, st);
    CK(hipStreamSynchronize(st));
    if (rc0 || rc1) printf("ln.fold ERROR %s\n", ezclip::last_error());
    else {
      std::vector<float> hcf((size_t)M * N), hs(2 * M), h1(N), h2(N);
      std::vector<uint16_t> hc((size_t)M * N);
      CK(hipMemcpy(hcf.data(), Cf, hcf.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hs.data(), st2, hs.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(h1.data(), c1, N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), c2, N * 4, hipMemcpyDeviceToHost));
      double md = 0; size_t nan = 0;
      size_t hb_i[4] = {0}, hb_it[4] = {0}, hb_crow[8] = {0}, hb_g[8] = {0}, hb_e[8] = {0}, hb_tile[16] = {0}, nbad = 0;
      for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
          const float ref = fmaf(hcf[(size_t)m * N + n], hs[2 * m], fmaf(hs[2 * m + 1], h1[n], h2[n]));
          const uint32_t u = (uint32_t)hc[(size_t)m * N + n] << 16;
          float got; memcpy(&got, &u, 4);
          const bool bad = (got != got) || fabsf(got - ref) / (fabs(ref) + 1.0) > 0.02;
          if (bad) { ++nbad; ++hb_i[(m % 128) / 32]; ++hb_it[(m % 32) / 8]; ++hb_crow[m % 8]; ++hb_g[(n % 64) / 8]; ++hb_e[n % 8]; ++hb_tile[(m / 256) % 16]; }
          if (got != got) { ++nan; continue; }
          md = std::max(md, (double)fabsf(got - ref) / (fabs(ref) + 1.0));
        }
      printf("ln.fold         M=%7d N=%5d K=%5d : max rel err %.3g (bf16 ulp = 3.9e-3), NaNs %zu, bad %zu\n", M, N, K, md, nan, nbad);
      if (nbad) {
        printf("   by i:"); for (size_t v : hb_i) printf(" %zu", v);
        printf("  by it:"); for (size_t v : hb_it) printf(" %zu", v);
        printf("  by crow:"); for (size_t v : hb_crow) printf(" %zu", v);
        printf("\n   by g:"); for (size_t v : hb_g) printf(" %zu", v);
        printf("  by e:"); for (size_t v : hb_e) printf(" %zu", v);
        printf("  by m-tile:"); for (size_t v : hb_tile) printf(" %zu", v);
        printf("\n");
      }
    }
    hipFree(A); hipFree(B); hipFree(C); hipFree(Cf); hipFree(st2); hipFree(c1); hipFree(c2);
  }
  // ---- attention (packed qkv [B*L, 3*H*64] bf16): old two-pass kernels (variant 0) vs the short-sequence kernels ----
  {
    struct AT { const char* name; int B, L, H; };
    const AT ats[] = {{"attn.vit", batch, 197, 12}, {"attn.bert", batch, 64, 12}, {"attn.vitl14", batch / 2, 257, 16}};
    for (const AT& t : ats) {
      const size_t rows = (size_t)t.B * t.L, W = (size_t)t.H * 64;
      uint16_t *qkv, *ctx0, *ctx1; float* lse;
      CK(hipMalloc(&qkv, rows * 3 * W * 2)); CK(hipMalloc(&ctx0, rows * W * 2)); CK(hipMalloc(&ctx1, rows * W * 2));
      CK(hipMalloc(&lse, (size_t)t.B * t.H * t.L * 4));
      fill_bf16<<<2048, 256, 0, st>>>(qkv, rows * 3 * W, 21u, 1.5f);
      printf("%-15s B=%5d L=%4d H=%3d :", t.name, t.B, t.L, t.H);
      for (int v = 0; v < 2; ++v) {
        ezclip::set_attention_variant(v == 0 ? 0 : -1);
        ezclip::AttnArgs a;
        a.q = qkv; a.k = qkv + W; a.v = qkv + 2 * W; a.row_stride = 3 * W;
        a.ctx = v == 0 ? ctx0 : ctx1; a.ctx_stride = W; a.lse = lse; a.B = t.B; a.L = t.L; a.H = t.H; a.scale = 0.125f;
        if (ezclip::attention_fwd(a, EZCLIP_BF16, st) != 0) { printf(" ERROR %s", ezclip::last_error()); continue; }
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int it = 0; it < iters; ++it) ezclip::attention_fwd(a, EZCLIP_BF16, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        const double gb = (double)rows * W * 2 * 4 / 1e9;
        printf("  %s %.3f ms (%.0f TF, %.2f TB/s)", v == 0 ? "two-pass" : "short", ms,
               4.0 * t.B * t.H * (double)t.L * t.L * 64 / ms / 1e9, gb / ms);
      }
      CK(hipMemsetAsync(d_md, 0, 4, st)); CK(hipMemsetAsync(d_bad, 0, 8, st));
      maxdiff_bf16<<<1024, 256, 0, st>>>(ctx0, ctx1, rows * W, d_md, d_bad);
      float md; CK(hipMemcpyAsync(&md, d_md, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
      printf("  [max |ctx diff| %.3g]\n", md);
      // backward: two-kernel version vs the fused short-sequence kernel
      {
        uint16_t *dctx, *dq0, *dq1;
        CK(hipMalloc(&dctx, rows * W * 2)); CK(hipMalloc(&dq0, rows * 3 * W * 2)); CK(hipMalloc(&dq1, rows * 3 * W * 2));
        fill_bf16<<<2048, 256, 0, st>>>(dctx, rows * W, 22u, 1.0f);
        printf("%-15s bwd              :", t.name);
        for (int v = 0; v < 2; ++v) {
          ezclip::set_attention_variant(v == 0 ? 0 : -1);
          ezclip::AttnBwdArgs ab;
          ab.f.q = qkv; ab.f.k = qkv + W; ab.f.v = qkv + 2 * W; ab.f.row_stride = 3 * W;
          ab.f.ctx = ctx1; ab.f.ctx_stride = W; ab.f.lse = lse; ab.f.B = t.B; ab.f.L = t.L; ab.f.H = t.H; ab.f.scale = 0.125f;
          uint16_t* dq = v == 0 ? dq0 : dq1;
          ab.dctx = dctx; ab.dq = dq; ab.dk = dq + W; ab.dv = dq + 2 * W;
          if (ezclip::attention_bwd(ab, EZCLIP_BF16, st) != 0) { printf(" ERROR %s", ezclip::last_error()); continue; }
          CK(hipStreamSynchronize(st));
          hipEvent_t e0, e1;
          CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
          CK(hipEventRecord(e0, st));
          for (int it = 0; it < iters; ++it) ezclip::attention_bwd(ab, EZCLIP_BF16, st);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms = 0;
          CK(hipEventElapsedTime(&ms, e0, e1));
          ms /= iters;
          printf("  %s %.3f ms (%.0f TF)", v == 0 ? "two kernels" : "fused", ms, 10.0 * t.B * t.H * (double)t.L * t.L * 64 / ms / 1e9);
        }
        // probes (ATTN_PROBE=1): the fused kernel with the bias-gradient reduction on, and timed one launch at a time between
        // bursts of large GEMMs (the clock / power state it meets inside a training step)
        if (getenv("ATTN_PROBE")) {
          ezclip::set_attention_variant(-1);
          float *dbp, *dbs;
          CK(hipMalloc(&dbp, (size_t)t.B * 3 * W * 4)); CK(hipMalloc(&dbs, 3 * W * 4)); CK(hipMemsetAsync(dbs, 0, 3 * W * 4, st));
          ezclip::AttnBwdArgs ab;
          ab.f.q = qkv; ab.f.k = qkv + W; ab.f.v = qkv + 2 * W; ab.f.row_stride = 3 * W;
          ab.f.ctx = ctx1; ab.f.ctx_stride = W; ab.f.lse = lse; ab.f.B = t.B; ab.f.L = t.L; ab.f.H = t.H; ab.f.scale = 0.125f;
          ab.dctx = dctx; ab.dq = dq1; ab.dk = dq1 + W; ab.dv = dq1 + 2 * W;
          hipEvent_t e0, e1;
          CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
          for (int withdb = 0; withdb < 2; ++withdb) {
            if (withdb) { ab.dbq = dbs; ab.dbk = dbs + W; ab.dbv = dbs + 2 * W; ab.db_part = dbp; }
            ezclip::attention_bwd(ab, EZCLIP_BF16, st);
            CK(hipEventRecord(e0, st));
            for (int it = 0; it < iters; ++it) ezclip::attention_bwd(ab, EZCLIP_BF16, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  [probe db=%d: %.3f ms]", withdb, ms / iters);
          }
          // hot: [4 x (M x 2304 x 768) GEMM, 1 x attention] x 12, the attention launches timed one by one
          {
            const int gM = t.B * 197, gN = 2304, gK = 768;
            uint16_t *gA, *gB, *gC;
            CK(hipMalloc(&gA, (size_t)gM * gK * 2)); CK(hipMalloc(&gB, (size_t)gN * gK * 2)); CK(hipMalloc(&gC, (size_t)gM * gN * 2));
            fill_bf16<<<2048, 256, 0, st>>>(gA, (size_t)gM * gK, 31u, 1.0f);
            fill_bf16<<<2048, 256, 0, st>>>(gB, (size_t)gN * gK, 32u, 0.05f);
            ezclip::GemmArgs g;
            g.A = gA; g.lda = gK; g.B = gB; g.ldb = gK; g.C = gC; g.ldc = gN; g.M = gM; g.N = gN; g.K = gK;
            float hot = 0, fhot = 0;
            ezclip::AttnArgs fa = ab.f; fa.ctx = ctx0;
            for (int rep = 0; rep < 14; ++rep) {
              for (int q = 0; q < 4; ++q) ezclip::gemm_nt(g, EZCLIP_BF16, st);
              CK(hipEventRecord(e0, st));
              ezclip::attention_bwd(ab, EZCLIP_BF16, st);
              CK(hipEventRecord(e1, st));
              CK(hipEventSynchronize(e1));
              float ms = 0;
              CK(hipEventElapsedTime(&ms, e0, e1));
              if (rep >= 2) hot += ms;
              for (int q = 0; q < 4; ++q) ezclip::gemm_nt(g, EZCLIP_BF16, st);
              CK(hipEventRecord(e0, st));
              ezclip::attention_fwd(fa, EZCLIP_BF16, st);
              CK(hipEventRecord(e1, st));
              CK(hipEventSynchronize(e1));
              CK(hipEventElapsedTime(&ms, e0, e1));
              if (rep >= 2) fhot += ms;
            }
            printf("  [between GEMM bursts: bwd+db %.3f ms, fwd %.3f ms]", hot / 12, fhot / 12);
            hipFree(gA); hipFree(gB); hipFree(gC);
          }
          hipFree(dbp); hipFree(dbs);
        }
        CK(hipMemsetAsync(d_md, 0, 4, st)); CK(hipMemsetAsync(d_bad, 0, 8, st));
        maxdiff_bf16<<<1024, 256, 0, st>>>(dq0, dq1, rows * 3 * W, d_md, d_bad);
        float md2; CK(hipMemcpyAsync(&md2, d_md, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        printf("  [max |dqkv diff| %.3g]\n", md2);
        hipFree(dctx); hipFree(dq0); hipFree(dq1);
      }
      hipFree(qkv); hipFree(ctx0); hipFree(ctx1); hipFree(lse);
    }
    ezclip::set_attention_variant(-1);
  }
  ezclip::set_gemm_variant(-1);
  return 0;
}
