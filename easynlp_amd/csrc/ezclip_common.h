// Shared device/host helpers for the ezclip HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace ezclip {

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int kWave = 64;  // CDNA wavefront

// TANH: 128x128 kernels only (pooler).  RELU: max(x, 0) in the activation slot; RELU_POST: applied AFTER the residual add
// (Bottleneck: relu(bn3(conv3(.)) + identity), modeling_chineseclip.py:72-73)
enum Act { ACT_NONE = 0, ACT_QUICKGELU = 1, ACT_GELU_ERF = 2, ACT_TANH = 3, ACT_RELU = 4, ACT_RELU_POST = 5 };

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}

// round-to-nearest-even, NaN-preserving: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_hw_t;

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_hw_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw_t));
}

__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kPerChunk = 4;  // elements per 16-byte chunk
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kPerChunk = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// load / store 4 consecutive elements (16-byte aligned for float, 8 for bf16)
__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
  float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void ld4(const bf16_t* p, float (&v)[4]) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
__device__ __forceinline__ void st4(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st4(bf16_t* p, const float (&v)[4]) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}

// unpack a 16-byte chunk into floats (4 for f32, 8 for bf16)
__device__ __forceinline__ void unpack_chunk(const uint4& c, float (&v)[4], float) {
  v[0] = __uint_as_float(c.x); v[1] = __uint_as_float(c.y);
  v[2] = __uint_as_float(c.z); v[3] = __uint_as_float(c.w);
}
__device__ __forceinline__ void unpack_chunk(const uint4& c, float (&v)[8], bf16_t) {
  v[0] = __uint_as_float(c.x << 16); v[1] = __uint_as_float(c.x & 0xffff0000u);
  v[2] = __uint_as_float(c.y << 16); v[3] = __uint_as_float(c.y & 0xffff0000u);
  v[4] = __uint_as_float(c.z << 16); v[5] = __uint_as_float(c.z & 0xffff0000u);
  v[6] = __uint_as_float(c.w << 16); v[7] = __uint_as_float(c.w & 0xffff0000u);
}

// erf via Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7): kept for reference / tools; the bf16 pipeline's GELU uses the
// polynomials below, the f32 path calls erff.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);   // v_rcp_f32 (1 ulp)
  float y = 1.061405429f;
  y = fmaf(y, t, -1.453152027f);
  y = fmaf(y, t, 1.421413741f);
  y = fmaf(y, t, -0.284496736f);
  y = fmaf(y, t, 0.254829592f);
  y = 1.0f - y * t * __expf(-ax * ax);
  return copysignf(y, x);
}

// ---- erf-GELU of the bf16 pipeline without transcendental instructions (round 3) ------------------------------------------
// F.gelu(x) = x Phi(x) (activations.py:45-48,98).  The Abramowitz-Stegun form above costs ~13 VALU instructions plus v_rcp_f32
// and v_exp_f32 (quarter rate) per element: 25 % of the BERT intermediate product's time went into its epilogue.  Here
//   Phi(x)   ~ 0.5 + xc R(t),   xc = clamp(x, -c, c),  t = 2 xc^2 / c^2 - 1        (Phi - 1/2 is odd: one polynomial)
//   gelu'(x) = Phi(x) + x phi(x) ~ 0.5 + xc Q(t)                                   (gelu' - 1/2 is odd as well)
// with minimax fits (Lawson iterations on Chebyshev nodes; evaluated by Horner in t in [-1, 1], which keeps float32 rounding
// at the 1e-7 level).  Phi (c = 4.2, degree 8; tools/fit_gelu_poly.py) is fitted UNDER THE CONSTRAINT R(1) = 0.5 / c, i.e. the
// polynomial is exactly 0 / 1 at the clamp (round 4; the unconstrained fit of round 3 left Phi(-c) +- 7.5e-6 there and
// gelu(x) = x Phi(x) off by |x| * 2e-5 for ANY x below -c: -7.2e-5 at x = -12): |Phi error| <= 1.36e-5 (at the end points, where
// 1 - Phi(4.2) = 1.33e-5), |gelu error| <= 5.7e-5 ABSOLUTE for all x -- outside the clamp x * (1 +- 1e-7) or x * (0 +- 1e-7)
// (tests/test_ops_gpu.py::test_gelu_polynomial_tails).  |gelu' error| <= 1.2e-5 (c = 4.8, degree 11).  13 / 16 plain multiply-adds per element, written on float pairs so that they
// issue as v_pk_fma_f32 / v_pk_mul_f32 (two elements per instruction); the scalar forms run the SAME operation sequence
// (every step one IEEE fma / mul), so the packed epilogue of the 8-phase GEMM and the scalar one of the 128x128 kernel stay
// bit-identical (tests/test_bench_regime_gpu.py).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr float kPhiC = 4.2f, kPhiTs = 2.0f / (4.2f * 4.2f);
constexpr float kPhiK[9] = {0.16785699129104614f, -0.081463746726512909f, 0.055767588317394257f, -0.039157509803771973f,
                            0.025356100872159004f, -0.013150581158697605f, 0.0073324446566402912f, -0.0061912033706903458f,
                            0.0026975346263498068f};
constexpr float kGpC = 4.8f, kGpTs = 2.0f / (4.8f * 4.8f);
constexpr float kGpK[12] = {0.1484677493572235f, -0.08022641390562057f, 0.07392347604036331f, -0.08052316308021545f,
                            0.0863509252667427f, -0.08870752155780792f, 0.08652839064598083f, -0.06128013879060745f,
                            0.026012925431132317f, -0.023687424138188362f, 0.029725147411227226f, -0.01241487916558981f};

template <typename V> __device__ __forceinline__ V ez_splat(float v);
template <> __device__ __forceinline__ float ez_splat<float>(float v) { return v; }
template <> __device__ __forceinline__ f32x2_t ez_splat<f32x2_t>(float v) { return f32x2_t{v, v}; }
__device__ __forceinline__ float ez_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ f32x2_t ez_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float ez_clamp(float x, float c) { return __builtin_amdgcn_fmed3f(x, -c, c); }
__device__ __forceinline__ f32x2_t ez_clamp(f32x2_t x, float c) {
  return f32x2_t{__builtin_amdgcn_fmed3f(x.x, -c, c), __builtin_amdgcn_fmed3f(x.y, -c, c)};
}
// 0.5 + xc * poly(t): V = float or f32x2_t
template <typename V, int N>
__device__ __forceinline__ V odd_poly_half(V x, float c, float ts, const float (&k)[N]) {
  const V xc = ez_clamp(x, c);
  const V t = ez_fma(xc * xc, ez_splat<V>(ts), ez_splat<V>(-1.0f));
  V r = ez_splat<V>(k[N - 1]);
#pragma unroll
  for (int i = N - 2; i >= 0; --i) r = ez_fma(r, t, ez_splat<V>(k[i]));
  return ez_fma(xc, r, ez_splat<V>(0.5f));
}
template <typename V> __device__ __forceinline__ V gelu_fast(V x) { return x * odd_poly_half<V>(x, kPhiC, kPhiTs, kPhiK); }
template <typename V> __device__ __forceinline__ V gelu_grad_fast(V x) { return odd_poly_half<V>(x, kGpC, kGpTs, kGpK); }

// FAST = true for the bf16 pipeline (hardware exp, polynomial erf)
template <bool FAST>
__device__ __forceinline__ float act_apply(float x, int act) {
  if (act == ACT_QUICKGELU) {
    // exp(-1.702 x) = 2^(x * -1.702 log2(e)): one multiply in front of v_exp_f32 instead of two
    if (FAST) return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -2.4554669595930157f));
    return x / (1.0f + expf(-1.702f * x));
  }
  if (act == ACT_GELU_ERF) {
    if (FAST) return gelu_fast<float>(x);
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
  }
  if (act == ACT_TANH) return tanhf(x);     // BertPooler / RobertaPooler (B rows per step: never hot)
  if (act == ACT_RELU) return fmaxf(x, 0.f);
  return x;
}

// d act(x) / dx
template <bool FAST>
__device__ __forceinline__ float act_grad(float x, int act) {
  if (act == ACT_QUICKGELU) {
    const float e = FAST ? __builtin_amdgcn_exp2f(x * -2.4554669595930157f) : expf(-1.702f * x);
    const float s = FAST ? __builtin_amdgcn_rcpf(1.0f + e) : 1.0f / (1.0f + e);
    return s * (1.0f + 1.702f * x * (1.0f - s));
  }
  if (act == ACT_GELU_ERF) {
    if (FAST) return gelu_grad_fast<float>(x);
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
  }
  if (act == ACT_TANH) {
    const float t = tanhf(x);
    return 1.0f - t * t;
  }
  return 1.0f;
}

template <typename T> struct IsFast { static constexpr bool value = false; };
template <> struct IsFast<bf16_t> { static constexpr bool value = true; };

// One 32x32 accumulate step over one 16-byte K chunk per lane.
//   A_hw lane (i = lane&31, g = lane>>5) supplies chunk a; B_hw lane (j = lane&31, g) supplies b.
//   D[i][j] += sum over the chunk's K elements; D layout: col j = lane&31,
//   row i = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// bf16: one v_mfma_f32_32x32x16_bf16 (K=16: 8 per half-wave).
// f32 : four v_mfma_f32_32x32x2_f32 (K=2 each: element e of both half-waves).
__device__ __forceinline__ void mma32(f32x16_t& acc, const uint4& a, const uint4& b, bf16_t) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
// v_mfma_f32_16x16x32_bf16 (round 4): D[16 x 16] += A[16 x 32] B[32 x 16], operands 8 bf16 per lane: lane (i = lane & 15,
// kq = lane >> 4) holds A[i][8 kq .. 8 kq + 7] and B[8 kq .. 8 kq + 7][i]; result column j = lane & 15, rows 4 (lane >> 4) + r.
// Same peak rate as 32x32x16 (1024 flop / cycle / SIMD) and the same operand bytes per flop out of LDS, but K = 32 per
// instruction means half the f32 accumulator read-modify-writes per flop: under the 1 400 W socket cap that is the difference --
// nothing but MFMAs on random register operands sustains 1.80 PFLOP/s (1.72 GHz) with 32x32x16 and 2.10 PFLOP/s (2.00 GHz) with
// 16x16x32, 2.4 with either on all-zero operands (tools/mfma_power_probe.hip, profiles/r4_mfma_shape_power_probe.log).  It is
// also the instruction of hipBLASLt's 256 x 256 x 64 kernel (profiles/r4_vendor_vs_ours_pmc.md).
// EZ_MI16 = 0 rebuilds the bf16 GEMM kernels on 32x32x16 for A/B runs (tools/build_variants.py).
#ifndef EZ_MI16
#define EZ_MI16 1
#endif
__device__ __forceinline__ void mma16(f32x4_t& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16_t& acc, const uint4& a, const uint4& b, float) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 16 lanes of a DPP row, every lane ends up with it (quad butterflies, then the 8- and 16-lane mirrors: four
// VALU instructions, no LDS-pipe round trips); half_sum: over the 32 lanes of a half-wave (one bpermute for the row pair)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);     // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);     // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);    // row_half_mirror
  return dpp_add<0x140>(v); // row_mirror
}
__device__ __forceinline__ float half_sum(float v) {
  v = row16_sum(v);
  return v + __shfl_xor(v, 16, 64);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// XCD-aware bijective remap of a 1-D block id: consecutive logical ids land on
// the same XCD (block b is dispatched to XCD b % 8 -- a speed assumption only).
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

}  // namespace ezclip

// ---- host side ------------------------------------------------------------
#define EZ_OK 0
#define EZ_ERR_INVALID 1
#define EZ_ERR_HIP 2
#define EZ_ERR_UNSUPPORTED 3
#define EZ_ERR_STATE 4

namespace ezclip {
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);
int launch_check(const char* file, int line);      // EZ_LAUNCH_CHECK (model.hip; EZCLIP_SYNC_LAUNCHES=1|2: synchronise after every launch)
}  // namespace ezclip

#define EZ_HIP(expr)                                                   \
  do {                                                                 \
    int _rc = ::ezclip::check_hip((expr), #expr);                      \
    if (_rc != EZ_OK) return _rc;                                      \
  } while (0)

#define EZ_LAUNCH_CHECK()                                              \
  do {                                                                 \
    int _rc = ::ezclip::launch_check(__FILE__, __LINE__);              \
    if (_rc != EZ_OK) return _rc;                                      \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE).  Every launcher that needs more than the default
// 64 KiB keeps one LdsOptIn per kernel instantiation: the largest size already granted per device, read and written atomically
// (two host threads racing set the attribute twice -- harmless; a process driving a second GPU sets it there as well.  Round 3
// kept one `static bool` per instantiation: ADVICE r3).
namespace ezclip {
struct LdsOptIn {
  static constexpr int kMaxDev = 32;
  std::atomic<int> granted[kMaxDev];
  LdsOptIn() { for (auto& g : granted) g.store(0, std::memory_order_relaxed); }
};
inline int ensure_lds(const void* kern, LdsOptIn& st, int bytes) {
  int dev = 0;
  int rc = check_hip(hipGetDevice(&dev), "hipGetDevice");
  if (rc != EZ_OK) return rc;
  const bool tracked = dev >= 0 && dev < LdsOptIn::kMaxDev;
  if (tracked && st.granted[dev].load(std::memory_order_acquire) >= bytes) return EZ_OK;
  rc = check_hip(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  if (rc != EZ_OK) return rc;
  if (tracked) {
    int cur = st.granted[dev].load(std::memory_order_relaxed);
    while (cur < bytes && !st.granted[dev].compare_exchange_weak(cur, bytes, std::memory_order_release)) {}
  }
  return EZ_OK;
}
}  // namespace ezclip
#define EZ_ENSURE_LDS(kern, state, bytes)                                                            \
  do {                                                                                               \
    int _rc = ::ezclip::ensure_lds(reinterpret_cast<const void*>(kern), (state), (int)(bytes));      \
    if (_rc != EZ_OK) return _rc;                                                                    \
  } while (0)

#define EZ_REQUIRE(cond, ...)                                          \
  do {                                                                 \
    if (!(cond)) {                                                     \
      ::ezclip::set_error(__VA_ARGS__);                                \
      return EZ_ERR_INVALID;                                           \
    }                                                                  \
  } while (0)
