// Image half of CLIPDataset.convert_single_row_to_example on the GPU (input pipeline):
//   _resize  (easynlp/appzoo/clip/data.py:52-72)   PIL BICUBIC, shorter side -> `size`
//   _center_crop (:29-50)                          crop x crop window
//   _normalize (:101-135)                          /255 in float32, (x - mean) / std, CHW
// Input: decoded RGB8 pixels (HWC) of a whole batch packed into one device buffer; output: float32 [n, 3, crop, crop],
// exactly the `pixel_values` ezclip_encode_image takes.  Base64 / JPEG decoding stays on the CPU (as in the reference).
//
// The resampling is Pillow's (PIL.Image.resize -> libImaging/Resample.c, 8-bit path), restated so that the result is
// BIT-IDENTICAL: a separable two-pass convolution -- horizontal, intermediate rounded to uint8, then vertical -- with,
// per output pixel, a window [xmin, xmin + n) of int32 coefficients carrying 22 fractional bits, computed in double
// precision from the Keys bicubic kernel (a = -0.5) stretched by the downscale factor (antialiasing), normalised, and
// rounded half away from zero; accumulators start at 2^21 and are shifted right by 22 and clamped to [0, 255].
// The coefficient tables are built on the HOST (plain IEEE double arithmetic in the same order as Pillow's C: nothing
// the device compiler could contract into FMAs) and only for what the crop keeps: `crop` columns and `crop` rows per
// image, plus the source rows the vertical windows touch.  The two device passes are pure byte / integer work:
//   pass 1: one workgroup per needed source row -- the row is staged in LDS with aligned dword loads, every lane
//           produces one output column (3 channels) -> tmp[row][crop][3] (uint8, L2-resident: ~0.25 MB per image)
//   pass 2: one lane per output pixel, window over tmp rows, clamp, 256-entry per-channel table lookup holding
//           (v / 255 - mean) / std evaluated as numpy does -> three coalesced float planes.
// Both are HBM-bound: the source is read once, the output written once.
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/ezclip.h"
#include "ezclip_common.h"
#include "kernels.h"

namespace ezclip {
namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

double bicubic_filter(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// Resample.c precompute_coeffs + normalize_coeffs_8bpc for outputs [first, first + count) of an axis resampled from
// in_size to out_size (full box).  identity (in_size == out_size: Pillow skips the pass): one coefficient of 2^22.
struct AxisTable {
  int ksize = 1;
  std::vector<int> bounds;   // [count][2]: first source index, taps
  std::vector<int> kk;       // [count][ksize]
};

AxisTable axis_table(int in_size, int out_size, int first, int count) {
  AxisTable t;
  if (in_size == out_size) {
    t.ksize = 1;
    t.bounds.resize((size_t)count * 2);
    t.kk.assign((size_t)count, 1 << kPrecisionBits);
    for (int o = 0; o < count; ++o) { t.bounds[2 * o] = first + o; t.bounds[2 * o + 1] = 1; }
    return t;
  }
  const float in0 = 0.f, in1 = (float)in_size;
  const double scale = (double)(in1 - in0) / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  t.ksize = (int)std::ceil(support) * 2 + 1;
  t.bounds.resize((size_t)count * 2);
  t.kk.assign((size_t)count * t.ksize, 0);
  const double ss = 1.0 / filterscale;
  std::vector<double> w((size_t)t.ksize);
  for (int o = 0; o < count; ++o) {
    const int xx = first + o;
    const double center = in0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      w[x] = bicubic_filter((x + xmin - center + 0.5) * ss);
      ww += w[x];
    }
    for (int x = 0; x < xmax; ++x) {
      const double v = ww != 0.0 ? w[x] / ww : w[x];
      t.kk[(size_t)o * t.ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << kPrecisionBits)) : (int)(0.5 + v * (1 << kPrecisionBits));
    }
    t.bounds[2 * o] = xmin;
    t.bounds[2 * o + 1] = xmax;
  }
  return t;
}

struct ImgPlan {
  uint64_t src_off;     // byte offset of the image in the packed buffer
  uint64_t tmp_off;     // byte offset of its [nrows][crop][3] intermediate in the workspace
  int in_w, in_h;
  int row0, nrows;      // source rows the vertical windows touch
  int col0, ncols;      // source columns the horizontal windows touch (staged in LDS)
  int ksize_h, hb, hk;  // horizontal table: taps, int offsets of bounds / coefficients in the table buffer
  int ksize_v, vb, vk;  // vertical table (bounds relative to row0)
};

__device__ __forceinline__ uint8_t clip8(int acc) {
  int v = acc >> kPrecisionBits;
  v = v < 0 ? 0 : v;
  return (uint8_t)(v > 255 ? 255 : v);
}

// pass 1: workgroup = (source row block, image).  Rows are staged one at a time: [ncols*3] bytes starting at the
// dword-aligned address below the first needed byte.
constexpr int kRowsPerWg = 4;
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ packed, const ImgPlan* __restrict__ plans,
                                                       const int* __restrict__ tables, uint8_t* __restrict__ tmp, int crop) {
  extern __shared__ __attribute__((aligned(16))) uint8_t row_lds[];
  const ImgPlan p = plans[blockIdx.y];
  const int r_first = blockIdx.x * kRowsPerWg;
  if (r_first >= p.nrows) return;
  const int* hb = tables + p.hb;
  const int* hk = tables + p.hk;
  const int xx = threadIdx.x;
  int xmin = 0, taps = 0;
  if (xx < crop) { xmin = hb[2 * xx] - p.col0; taps = hb[2 * xx + 1]; }
  const int nbytes = p.ncols * 3;
  for (int rr = 0; rr < kRowsPerWg && r_first + rr < p.nrows; ++rr) {
    const int r = r_first + rr;
    const uint64_t start = p.src_off + ((uint64_t)(p.row0 + r) * p.in_w + p.col0) * 3u;
    const uint64_t abase = (uint64_t)(uintptr_t)packed + start;
    const int mis = (int)(abase & 3u);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(abase - mis);
    const int ndw = (nbytes + mis + 3) >> 2;
    __syncthreads();   // previous row consumed
    for (int i = threadIdx.x; i < ndw; i += blockDim.x) reinterpret_cast<uint32_t*>(row_lds)[i] = src[i];
    __syncthreads();
    if (xx < crop) {
      const uint8_t* px = row_lds + mis + xmin * 3;
      int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
      for (int x = 0; x < taps; ++x) {
        const int k = hk[xx * p.ksize_h + x];
        s0 += (int)px[3 * x] * k;
        s1 += (int)px[3 * x + 1] * k;
        s2 += (int)px[3 * x + 2] * k;
      }
      uint8_t* o = tmp + p.tmp_off + ((uint64_t)r * crop + xx) * 3u;
      o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
    }
  }
}

// pass 2: lane = output pixel (y, x) of image blockIdx.y; lut [3][256] floats
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const ImgPlan* __restrict__ plans, const int* __restrict__ tables,
                                                            const uint8_t* __restrict__ tmp, const float* __restrict__ lut,
                                                            float* __restrict__ out, int crop) {
  const ImgPlan p = plans[blockIdx.y];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= crop * crop) return;
  const int y = idx / crop, x = idx - y * crop;
  const int* vb = tables + p.vb;
  const int* vk = tables + p.vk + y * p.ksize_v;
  const int ymin = vb[2 * y], taps = vb[2 * y + 1];
  const uint8_t* t = tmp + p.tmp_off + ((uint64_t)ymin * crop + x) * 3u;
  int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
  for (int k = 0; k < taps; ++k) {
    const int c = vk[k];
    const uint8_t* q = t + (uint64_t)k * crop * 3u;
    s0 += (int)q[0] * c;
    s1 += (int)q[1] * c;
    s2 += (int)q[2] * c;
  }
  float* o = out + (uint64_t)blockIdx.y * 3u * crop * crop + idx;
  o[0] = lut[clip8(s0)];
  o[(uint64_t)crop * crop] = lut[256 + clip8(s1)];
  o[(uint64_t)2 * crop * crop] = lut[512 + clip8(s2)];
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct BatchPlan {
  std::vector<ImgPlan> plans;
  std::vector<int> tables;
  size_t tmp_bytes = 0;
  int max_rows = 0, max_cols = 0;
};

// data.py:62-71 (resized size) and :43-48 (crop origin)
int plan_batch(const ezclip_image_desc* desc, int n, int size, int crop, BatchPlan* out) {
  out->plans.resize((size_t)n);
  for (int i = 0; i < n; ++i) {
    const int w = desc[i].width, h = desc[i].height;
    EZ_REQUIRE(w > 0 && h > 0, "preprocess: image %d has size %d x %d", i, w, h);
    const int shrt = w <= h ? w : h, lng = w <= h ? h : w;
    int nw = w, nh = h;
    if (shrt != size) {
      const int new_long = (int)((double)size * lng / shrt);    // int(size * long / short): python float division
      if (w <= h) { nw = size; nh = new_long; } else { nw = new_long; nh = size; }
    }
    EZ_REQUIRE(nw >= crop && nh >= crop, "preprocess: image %d (%d x %d -> %d x %d) is smaller than the %d crop (the reference pads; "
               "not on this path)", i, w, h, nw, nh, crop);
    const int left = (int)((nw - crop + 1) * 0.5), top = (int)((nh - crop + 1) * 0.5);
    AxisTable th = axis_table(w, nw, left, crop), tv = axis_table(h, nh, top, crop);
    ImgPlan& p = out->plans[(size_t)i];
    p.src_off = desc[i].offset;
    p.in_w = w; p.in_h = h;
    p.row0 = tv.bounds[0];
    p.nrows = tv.bounds[2 * (crop - 1)] + tv.bounds[2 * (crop - 1) + 1] - p.row0;
    p.col0 = th.bounds[0];
    p.ncols = th.bounds[2 * (crop - 1)] + th.bounds[2 * (crop - 1) + 1] - p.col0;
    for (int o = 0; o < crop; ++o) tv.bounds[2 * o] -= p.row0;        // vertical windows index the intermediate
    p.ksize_h = th.ksize; p.ksize_v = tv.ksize;
    p.hb = (int)out->tables.size(); out->tables.insert(out->tables.end(), th.bounds.begin(), th.bounds.end());
    p.hk = (int)out->tables.size(); out->tables.insert(out->tables.end(), th.kk.begin(), th.kk.end());
    p.vb = (int)out->tables.size(); out->tables.insert(out->tables.end(), tv.bounds.begin(), tv.bounds.end());
    p.vk = (int)out->tables.size(); out->tables.insert(out->tables.end(), tv.kk.begin(), tv.kk.end());
    p.tmp_off = out->tmp_bytes;
    out->tmp_bytes += align_up((size_t)p.nrows * crop * 3, 256);
    if (p.nrows > out->max_rows) out->max_rows = p.nrows;
    if (p.ncols > out->max_cols) out->max_cols = p.ncols;
  }
  return EZ_OK;
}

struct WsLayout { size_t plans, tables, lut, tmp, total; };
WsLayout ws_layout(const BatchPlan& b) {
  WsLayout l;
  l.plans = 0;
  l.tables = align_up(b.plans.size() * sizeof(ImgPlan), 256);
  l.lut = l.tables + align_up(b.tables.size() * sizeof(int), 256);
  l.tmp = l.lut + align_up(3 * 256 * sizeof(float), 256);
  l.total = l.tmp + b.tmp_bytes + 256;
  return l;
}

}  // namespace

// host-only: the window table of one axis (tests pin it against the Pillow-pinned oracle without a GPU)
int resample_table(int in_size, int out_size, int first, int count, int* ksize, int* bounds, int* kk, int kk_capacity) {
  EZ_REQUIRE(in_size > 0 && out_size > 0 && first >= 0 && count > 0 && first + count <= out_size && ksize && bounds && kk,
             "resample_table: bad argument");
  const AxisTable t = axis_table(in_size, out_size, first, count);
  EZ_REQUIRE((size_t)kk_capacity >= t.kk.size(), "resample_table: kk capacity %d < %zu", kk_capacity, t.kk.size());
  *ksize = t.ksize;
  std::memcpy(bounds, t.bounds.data(), t.bounds.size() * sizeof(int));
  std::memcpy(kk, t.kk.data(), t.kk.size() * sizeof(int));
  return EZ_OK;
}

size_t preprocess_workspace_bytes(const ezclip_image_desc* desc, int n, int size, int crop) {
  BatchPlan b;
  if (desc == nullptr || n <= 0 || plan_batch(desc, n, size, crop, &b) != EZ_OK) return 0;
  return ws_layout(b).total;
}

int preprocess_images(const uint8_t* packed, const ezclip_image_desc* desc, int n, int size, int crop, const float* mean,
                      const float* stdv, float* out, void* ws, size_t ws_bytes, hipStream_t stream) {
  EZ_REQUIRE(packed && desc && out && ws && n > 0, "preprocess_images: null/empty argument");
  EZ_REQUIRE(crop > 0 && crop <= 256 && size > 0, "preprocess_images: crop %d must be in [1, 256]", crop);
  EZ_REQUIRE(((uintptr_t)ws % 256) == 0, "preprocess_images: workspace must be 256-byte aligned");
  BatchPlan b;
  int rc = plan_batch(desc, n, size, crop, &b);
  if (rc != EZ_OK) return rc;
  const WsLayout l = ws_layout(b);
  EZ_REQUIRE(ws_bytes >= l.total, "preprocess_images: workspace too small (%zu < %zu)", ws_bytes, l.total);
  // (v / 255 - mean) / std exactly as numpy evaluates it in float32 (data.py:93-94,131-132)
  float lut[3 * 256];
  for (int c = 0; c < 3; ++c)
    for (int v = 0; v < 256; ++v) {
      volatile float q = (float)v / 255.0f;
      volatile float d = q - mean[c];
      lut[c * 256 + v] = d / stdv[c];
    }
  // plans + tables + lut go through one library-owned pinned staging buffer (a pageable source would either make the
  // copy synchronous or leave the DMA reading freed vectors); an event guards its reuse by the next call
  struct Staging { char* host = nullptr; size_t bytes = 0; hipEvent_t ev = nullptr; bool pending = false; };
  static std::mutex stg_mutex;
  static std::map<hipStream_t, Staging> stg_by_stream;          // one staging buffer per stream
  std::lock_guard<std::mutex> stg_lock(stg_mutex);
  Staging& stg = stg_by_stream[stream];
  const size_t up_bytes = l.tmp;
  if (stg.pending) { EZ_HIP(hipEventSynchronize(stg.ev)); stg.pending = false; }
  if (stg.bytes < up_bytes) {
    if (stg.host) EZ_HIP(hipHostFree(stg.host));
    stg.host = nullptr; stg.bytes = 0;
    EZ_HIP(hipHostMalloc(reinterpret_cast<void**>(&stg.host), up_bytes + (up_bytes >> 1), hipHostMallocDefault));
    stg.bytes = up_bytes + (up_bytes >> 1);
  }
  if (stg.ev == nullptr) EZ_HIP(hipEventCreateWithFlags(&stg.ev, hipEventDisableTiming));
  std::memcpy(stg.host + l.plans, b.plans.data(), b.plans.size() * sizeof(ImgPlan));
  std::memcpy(stg.host + l.tables, b.tables.data(), b.tables.size() * sizeof(int));
  std::memcpy(stg.host + l.lut, lut, sizeof(lut));
  char* w = static_cast<char*>(ws);
  EZ_HIP(hipMemcpyAsync(w, stg.host, up_bytes, hipMemcpyHostToDevice, stream));
  EZ_HIP(hipEventRecord(stg.ev, stream));
  stg.pending = true;
  const ImgPlan* dplans = reinterpret_cast<const ImgPlan*>(w + l.plans);
  const int* dtables = reinterpret_cast<const int*>(w + l.tables);
  const size_t lds = align_up((size_t)b.max_cols * 3 + 8, 16);
  EZ_REQUIRE(lds <= 64 * 1024, "preprocess_images: a source row of %d pixels does not fit the staging buffer", b.max_cols);
  {
    ProfScope ps(PROF_ROWOP, (double)b.tmp_bytes * 2 + (double)n * 3 * crop * crop * 4, stream);
    hipLaunchKernelGGL(resize_h_kernel, dim3((b.max_rows + kRowsPerWg - 1) / kRowsPerWg, n), dim3(256), lds, stream, packed, dplans,
                       dtables, reinterpret_cast<uint8_t*>(w + l.tmp), crop);
    hipLaunchKernelGGL(resize_v_norm_kernel, dim3((crop * crop + 255) / 256, n), dim3(256), 0, stream, dplans, dtables,
                       reinterpret_cast<const uint8_t*>(w + l.tmp), reinterpret_cast<const float*>(w + l.lut), out, crop);
  }
  EZ_LAUNCH_CHECK();
  return EZ_OK;
}

}  // namespace ezclip
