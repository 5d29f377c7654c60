// Counter-based dropout masks for the BERT text tower in train mode.
//
// Reference: nn.Dropout(hidden_dropout_prob) after the embedding LayerNorm, after BertSelfOutput.dense and after
// BertOutput.dense (modeling_bert.py:128,266,344) and nn.Dropout(attention_probs_dropout_prob) on the softmax
// output (modeling_bert.py:238).  torch draws its masks from the global Philox stream of the device; a bit-identical
// stream is neither possible nor part of the contract (a different launch geometry already changes it).  What is kept:
// independent Bernoulli(1 - p) keep decisions per element, survivors scaled by 1/(1 - p), the same mask in forward
// and backward.
//
// No mask is ever stored.  Element (row, col) of dropout site `sid` keeps its value iff
//     philox4x32_10(counter = (col >> 2, row, sid, 0), key = seed)[col & 3]  >=  thr,    thr = round(p * 2^32)
// so any kernel, in any layout, can regenerate the decision of any element (the attention kernels need it once per
// query-major lane in the forward / dQ pass and once per key-major lane in the dK/dV pass).  Rows are token rows
// (hidden sites) or (batch*heads + head)*L + query (attention probabilities, col = key).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ezclip {

struct DropCfg {
  uint32_t thr = 0;      // 0: dropout off
  float scale = 1.f;     // 1 / (1 - p)
  uint32_t k0 = 0, k1 = 0;
  uint32_t sid = 0;
};

inline DropCfg make_drop(float p, uint64_t seed, uint32_t sid) {
  DropCfg d;
  if (!(p > 0.f)) return d;
  const double t = (double)p * 4294967296.0 + 0.5;
  d.thr = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
  d.scale = 1.0f / (1.0f - p);
  d.k0 = (uint32_t)seed;
  d.k1 = (uint32_t)(seed >> 32);
  d.sid = sid;
  return d;
}

// dropout sites of the text tower (sid)
inline uint32_t drop_sid_embed() { return 0u; }
inline uint32_t drop_sid_attn(int layer) { return 1u + 3u * (uint32_t)layer; }
inline uint32_t drop_sid_self_out(int layer) { return 2u + 3u * (uint32_t)layer; }
inline uint32_t drop_sid_out(int layer) { return 3u + 3u * (uint32_t)layer; }

// Philox-4x32-10 (Salmon et al., SC'11); known-answer vectors are checked in tests/test_dropout.py through
// ezclip_op_dropout_mask.
__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}

// the four 32-bit words deciding columns 4*colquad .. 4*colquad + 3 of `row`
__device__ __forceinline__ uint4 drop_words(const DropCfg& d, uint32_t row, uint32_t colquad) {
  return philox4x32_10(colquad, row, d.sid, 0u, d.k0, d.k1);
}
__device__ __forceinline__ uint32_t drop_word(const DropCfg& d, uint32_t row, uint32_t col) {
  const uint4 w = drop_words(d, row, col >> 2);
  const uint32_t c = col & 3u;
  return c == 0 ? w.x : c == 1 ? w.y : c == 2 ? w.z : w.w;
}

}  // namespace ezclip
