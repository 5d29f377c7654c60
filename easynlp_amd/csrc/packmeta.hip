// Packing metadata of a [B, S] text batch on the device, in one launch and without a stream synchronisation.
//
// CHINESE_CLIP.encode_text masks with text.ne(0) (modeling_chineseclip.py:347-348; the huggingface_clip branch hands the
// tokenizer's attention_mask to RobertaModel, appzoo/clip/model.py:131-133), BertModel turns the mask into a -10000 key bias
// and only row 0 of the last hidden state is read: masked tokens never reach the feature, so the tower runs on
//   keep[b][t] = mask[b][t] != 0  or  t == 0 (the CLS query)  or  sentence b has no unmasked token at all (uniform softmax over ALL
//                positions in the reference).
// Round 2 built rowmap / cu / lens with ~10 torch launches and read two scalars back with .tolist() -- a device synchronisation in
// every step that sees a new batch.  Here ONE workgroup does it: a wave per sentence (lane = token, ballot = keep mask, popcount =
// length), a block scan for cu, a second sweep that scatters rowmap[cu[b] + rank] = b S + t, and thread 0 writes
// (rows, longest, prefix, ticket) into PINNED HOST memory the library owns; the host polls the ticket word
// (ezclip_pack_text_meta_result) -- no hipStreamSynchronize, no hipMemcpy.  `prefix`: the kept tokens of every sentence are a
// prefix of it (what train-mode dropout needs: a packed position is then the padded one).
#include "ezclip_common.h"
#include "kernels.h"

namespace ezclip {
namespace {

constexpr int kPmThreads = 1024, kPmWaves = 16, kPmMaxChunks = 8;      // S <= 512

// keep masks of sentence b: m[c] bit t = keep[b][64 c + t]; returns the number kept; `pre`: they form a prefix
__device__ __forceinline__ int sentence_masks(const int64_t* __restrict__ src, int b, int S, int lane, uint64_t (&m)[kPmMaxChunks],
                                              bool& pre) {
  const int nc = (S + 63) >> 6;
  bool any = false;
#pragma unroll
  for (int c = 0; c < kPmMaxChunks; ++c) {
    m[c] = 0;
    if (c < nc) {
      const int t = 64 * c + lane;
      const bool k = t < S && src[(int64_t)b * S + t] != 0;
      m[c] = __ballot(k);
      any |= m[c] != 0;
    }
  }
  int cnt = 0;
  pre = true;
  bool seen_zero = false;
#pragma unroll
  for (int c = 0; c < kPmMaxChunks; ++c) {
    if (c < nc) {
      const int valid = min(64, S - 64 * c);
      const uint64_t full = valid == 64 ? ~0ull : ((1ull << valid) - 1ull);
      if (!any) m[c] = full;                 // no unmasked key: the whole sentence is kept
      if (c == 0) m[c] |= 1ull;              // the CLS query
      cnt += __popcll(m[c]);
      if (seen_zero && m[c] != 0) pre = false;
      if ((m[c] & (m[c] + 1ull)) != 0) pre = false;       // not of the form 0..01..1
      if (m[c] != full) seen_zero = true;
    }
  }
  return cnt;
}

// S <= 64: one 64-token chunk per sentence.  A wave keeps EIGHT sentences' loads in flight (the loop is latency-bound: one
// dependent global load per sentence and wave measured 114 us for 1024 sentences; this form ~15 us)
constexpr int kPmUnroll = 8;
__device__ __forceinline__ uint64_t keep_mask64(uint64_t m, int S) {
  const uint64_t full = S == 64 ? ~0ull : ((1ull << S) - 1ull);
  if (m == 0) m = full;                      // no unmasked key: the whole sentence is kept
  return m | 1ull;                           // the CLS query
}

__global__ __launch_bounds__(kPmThreads) void pack_meta_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ mask, int B,
                                                               int S, int* __restrict__ rowmap, int* __restrict__ cu,
                                                               int* __restrict__ lens, volatile int* host_out, int ticket) {
  extern __shared__ int sh[];               // lens / cu of the batch [B] + scan scratch
  int* s_len = sh;
  __shared__ int s_part[kPmThreads];
  __shared__ int s_red[3][kPmWaves];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t* src = mask ? mask : ids;
  int longest = 0;
  bool prefix = true;
  const bool one_chunk = S <= 64;
  if (one_chunk) {
    for (int b0 = wave; b0 < B; b0 += kPmWaves * kPmUnroll) {
      int64_t v[kPmUnroll];
#pragma unroll
      for (int u = 0; u < kPmUnroll; ++u) {
        const int b = b0 + u * kPmWaves;
        v[u] = (b < B && lane < S) ? src[(int64_t)b * S + lane] : 0;
      }
#pragma unroll
      for (int u = 0; u < kPmUnroll; ++u) {
        const int b = b0 + u * kPmWaves;
        const uint64_t m = keep_mask64(__ballot(v[u] != 0), S);
        if (b < B) {
          const int n = __popcll(m);
          if (lane == 0) s_len[b] = n;
          longest = max(longest, n);
          prefix &= (m & (m + 1ull)) == 0;
        }
      }
    }
  } else {
    for (int b = wave; b < B; b += kPmWaves) {
      uint64_t m[kPmMaxChunks];
      bool pre;
      const int n = sentence_masks(src, b, S, lane, m, pre);
      if (lane == 0) s_len[b] = n;
      longest = max(longest, n);
      prefix &= pre;
    }
  }
  __syncthreads();
  // exclusive scan of s_len: thread i owns a contiguous segment
  const int seg = (B + kPmThreads - 1) / kPmThreads;
  int sum = 0;
  for (int i = 0; i < seg; ++i) {
    const int b = tid * seg + i;
    if (b < B) sum += s_len[b];
  }
  s_part[tid] = sum;
  __syncthreads();
  for (int o = 1; o < kPmThreads; o <<= 1) {     // Hillis-Steele over 1024 partials (10 rounds; once per batch)
    const int v = tid >= o ? s_part[tid - o] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  const int total = s_part[kPmThreads - 1];
  int run = s_part[tid] - sum;
  for (int i = 0; i < seg; ++i) {
    const int b = tid * seg + i;
    if (b < B) {
      const int n = s_len[b];
      lens[b] = n;
      cu[b] = run;
      s_len[b] = run;                          // (own segment only: no other thread reads it before the barrier)
      run += n;
    }
  }
  __syncthreads();
  if (one_chunk) {
    for (int b0 = wave; b0 < B; b0 += kPmWaves * kPmUnroll) {
      int64_t v[kPmUnroll];
#pragma unroll
      for (int u = 0; u < kPmUnroll; ++u) {
        const int b = b0 + u * kPmWaves;
        v[u] = (b < B && lane < S) ? src[(int64_t)b * S + lane] : 0;
      }
#pragma unroll
      for (int u = 0; u < kPmUnroll; ++u) {
        const int b = b0 + u * kPmWaves;
        const uint64_t m = keep_mask64(__ballot(v[u] != 0), S);
        if (b < B && ((m >> lane) & 1ull)) rowmap[s_len[b] + __popcll(m & ((1ull << lane) - 1ull))] = b * S + lane;
      }
    }
  }
  for (int b = wave; b < B && !one_chunk; b += kPmWaves) {
    uint64_t m[kPmMaxChunks];
    bool pre;
    sentence_masks(src, b, S, lane, m, pre);
    int base = s_len[b];
    const int nc = (S + 63) >> 6;
#pragma unroll
    for (int c = 0; c < kPmMaxChunks; ++c) {
      if (c < nc) {
        if ((m[c] >> lane) & 1ull) rowmap[base + __popcll(m[c] & ((1ull << lane) - 1ull))] = b * S + 64 * c + lane;
        base += __popcll(m[c]);
      }
    }
  }
  // longest / prefix over the block (both are wave-uniform)
  if (lane == 0) { s_red[0][wave] = longest; s_red[1][wave] = prefix ? 1 : 0; }
  __syncthreads();
  if (tid == 0) {
    int lg = 0, pf = 1;
    for (int w = 0; w < kPmWaves; ++w) { lg = max(lg, s_red[0][w]); pf &= s_red[1][w]; }
    host_out[0] = total;
    host_out[1] = lg;
    host_out[2] = pf;
    __threadfence_system();
    host_out[3] = ticket;                      // the host polls this word
    __threadfence_system();
  }
}

}  // namespace

int pack_text_meta(const int64_t* ids, const int64_t* mask, int B, int S, int* rowmap, int* cu, int* lens, int* host_out_dev,
                   int ticket, hipStream_t stream) {
  EZ_REQUIRE(ids && rowmap && cu && lens && host_out_dev, "pack_text_meta: null argument");
  EZ_REQUIRE(B > 0 && S > 0 && S <= 64 * kPmMaxChunks && B <= 12288, "pack_text_meta: batch %d x %d (at most 12288 x 512)", B, S);
  hipLaunchKernelGGL(pack_meta_kernel, dim3(1), dim3(kPmThreads), (size_t)B * sizeof(int), stream, ids, mask, B, S, rowmap, cu, lens,
                     host_out_dev, ticket);
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace ezclip
