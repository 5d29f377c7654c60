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

// What the host still computes per axis when the DEVICE fills the tables (O(1) per image): the scalars of
// precompute_coeffs and the window of the first / last kept output (which source rows / columns are touched at all).
struct AxisGeom {
  int in_size, out_size, first, ksize, identity;
  int lo, hi;                 // source range [lo, hi) touched by outputs first .. first + count - 1
  double scale, support, ss;
};

void window_of(const AxisGeom& g, int xx, int* xmin, int* taps) {
  const double center = 0.0f + (xx + 0.5) * g.scale;
  int lo = (int)(center - g.support + 0.5);
  if (lo < 0) lo = 0;
  int hi = (int)(center + g.support + 0.5);
  if (hi > g.in_size) hi = g.in_size;
  *xmin = lo;
  *taps = hi - lo;
}

AxisGeom axis_geom(int in_size, int out_size, int first, int count) {
  AxisGeom g;
  g.in_size = in_size; g.out_size = out_size; g.first = first;
  g.identity = in_size == out_size;
  if (g.identity) {
    g.ksize = 1; g.scale = 1.0; g.support = 0.0; g.ss = 1.0;
    g.lo = first; g.hi = first + count;
    return g;
  }
  const float in0 = 0.f, in1 = (float)in_size;
  g.scale = (double)(in1 - in0) / out_size;
  const double filterscale = g.scale < 1.0 ? 1.0 : g.scale;
  g.support = 2.0 * filterscale;
  g.ksize = (int)std::ceil(g.support) * 2 + 1;
  g.ss = 1.0 / filterscale;
  int a, n;
  window_of(g, first, &a, &n);
  g.lo = a;
  window_of(g, first + count - 1, &a, &n);
  g.hi = a + n;
  return g;
}

struct ImgPlan {
  uint64_t src_off;     // byte offset of the image in the packed buffer
  uint64_t tmp_off;     // byte offset of its [nrows][crop][3] intermediate in the workspace
  int in_w, in_h;
  int row0, nrows;      // source rows the vertical windows touch
  int col0, ncols;      // source columns the horizontal windows touch (staged in LDS)
  int ksize_h, hb, hk;  // horizontal table: taps, int offsets of bounds / coefficients in the table buffer
  int ksize_v, vb, vk;  // vertical table (bounds relative to row0)
  // device-built tables: per-axis scalars (h, then v)
  int out_size[2], first[2], identity[2];
  double scale[2], support[2], ss[2];
};

// Resample.c bicubic_filter / precompute_coeffs / normalize_coeffs_8bpc on the device.  Every double operation is an
// explicitly rounded intrinsic in Pillow's operation order (no FMA contraction), so the tables equal the host's bit for bit
// (tests/test_preprocess.py compares them over many sizes).
// Plain operators under `#pragma clang fp contract(off)`: hipcc's default for device code is contract(fast), and the
// __dmul_rn / __dadd_rn wrappers of the HIP headers are plain operators compiled in the HEADER's context -- a product feeding a
// sum became an FMA (seen as windows off by one where center + support + 0.5 sits one ulp under an integer).
__device__ __forceinline__ double bicubic_filter_dev(double x) {
#pragma clang fp contract(off)
  if (x < 0.0) x = -x;
  if (x < 1.0) {
    double t = 1.5 * x;
    t = t - 2.5;
    t = t * x;
    t = t * x;
    return t + 1.0;
  }
  if (x < 2.0) {
    double t = x - 5.0;
    t = t * x;
    t = t + 8.0;
    t = t * x;
    t = t - 4.0;
    return t * -0.5;
  }
  return 0.0;
}

// one thread per kept output of one axis of one image: bounds[o] = (first source index - shift, taps), kk[o][ksize]
__device__ __forceinline__ void fill_axis_output(int in_size, int xx, double scale, double support, double ss, int ksize,
                                                 int identity, int shift, int* bounds, int* kk) {
#pragma clang fp contract(off)
  if (identity) {
    bounds[0] = xx - shift; bounds[1] = 1;
    kk[0] = 1 << kPrecisionBits;
    return;
  }
  double center = (double)xx + 0.5;
  center = center * scale;
  double lo = center - support;
  lo = lo + 0.5;
  int xmin = (int)lo;
  if (xmin < 0) xmin = 0;
  double hi = center + support;
  hi = hi + 0.5;
  int xmax = (int)hi;
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double a = (double)(x + xmin) - center;
    a = a + 0.5;
    a = a * ss;
    ww = ww + bicubic_filter_dev(a);
  }
  for (int x = 0; x < ksize; ++x) {
    int c = 0;
    if (x < xmax) {
      double a = (double)(x + xmin) - center;
      a = a + 0.5;
      a = a * ss;
      const double w = bicubic_filter_dev(a);
      const double v = ww != 0.0 ? w / ww : w;
      double sc = v * (double)(1 << kPrecisionBits);
      sc = v < 0 ? -0.5 + sc : 0.5 + sc;
      c = (int)sc;
    }
    kk[x] = c;
  }
  bounds[0] = xmin - shift;
  bounds[1] = xmax;
}

__global__ __launch_bounds__(256) void resample_tables_kernel(const ImgPlan* __restrict__ plans, int* __restrict__ tables, int crop) {
  const ImgPlan p = plans[blockIdx.y];
  const int axis = blockIdx.x;          // 0 horizontal, 1 vertical
  const int o = threadIdx.x;
  if (o >= crop) return;
  const int in_size = axis == 0 ? p.in_w : p.in_h;
  const int ksize = axis == 0 ? p.ksize_h : p.ksize_v;
  int* bounds = tables + (axis == 0 ? p.hb : p.vb) + 2 * o;
  int* kk = tables + (axis == 0 ? p.hk : p.vk) + o * ksize;
  fill_axis_output(in_size, p.first[axis] + o, p.scale[axis], p.support[axis], p.ss[axis], ksize, p.identity[axis],
                   axis == 1 ? p.row0 : 0, bounds, kk);
}

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

__global__ __launch_bounds__(256) void resample_tables_split_kernel(const ImgPlan* __restrict__ plans, int* bounds, int* kk, int count) {
  const ImgPlan p = plans[0];
  const int o = threadIdx.x;
  if (o >= count) return;
  fill_axis_output(p.in_w, p.first[0] + o, p.scale[0], p.support[0], p.ss[0], p.ksize_h, p.identity[0], 0, bounds + 2 * o,
                   kk + o * p.ksize_h);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct BatchPlan {
  std::vector<ImgPlan> plans;
  std::vector<int> tables;      // host-built tables (empty when the device builds them)
  size_t table_ints = 0;        // size of the table buffer either way
  size_t tmp_bytes = 0;
  int max_rows = 0, max_cols = 0;
};

bool g_device_tables = true;    // ezclip_debug_set(5, v): 0 = window tables computed on the host (the verified fallback)

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
    ImgPlan& p = out->plans[(size_t)i];
    p.src_off = desc[i].offset;
    p.in_w = w; p.in_h = h;
    const AxisGeom gh = axis_geom(w, nw, left, crop), gv = axis_geom(h, nh, top, crop);
    p.row0 = gv.lo; p.nrows = gv.hi - gv.lo;
    p.col0 = gh.lo; p.ncols = gh.hi - gh.lo;
    p.ksize_h = gh.ksize; p.ksize_v = gv.ksize;
    const AxisGeom* gs[2] = {&gh, &gv};
    for (int a = 0; a < 2; ++a) {
      p.out_size[a] = gs[a]->out_size; p.first[a] = gs[a]->first; p.identity[a] = gs[a]->identity;
      p.scale[a] = gs[a]->scale; p.support[a] = gs[a]->support; p.ss[a] = gs[a]->ss;
    }
    if (g_device_tables) {
      size_t off = out->table_ints;
      p.hb = (int)off; off += 2 * (size_t)crop;
      p.hk = (int)off; off += (size_t)crop * gh.ksize;
      p.vb = (int)off; off += 2 * (size_t)crop;
      p.vk = (int)off; off += (size_t)crop * gv.ksize;
      out->table_ints = off;
    } else {
      AxisTable th = axis_table(w, nw, left, crop), tv = axis_table(h, nh, top, crop);
      for (int o = 0; o < crop; ++o) tv.bounds[2 * o] -= p.row0;        // vertical windows index the intermediate
      p.hb = (int)out->tables.size(); out->tables.insert(out->tables.end(), th.bounds.begin(), th.bounds.end());
      p.hk = (int)out->tables.size(); out->tables.insert(out->tables.end(), th.kk.begin(), th.kk.end());
      p.vb = (int)out->tables.size(); out->tables.insert(out->tables.end(), tv.bounds.begin(), tv.bounds.end());
      p.vk = (int)out->tables.size(); out->tables.insert(out->tables.end(), tv.kk.begin(), tv.kk.end());
      out->table_ints = out->tables.size();
    }
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
  l.plans = 0;                                   // [plans | lut | tables | tmp]: plans + lut (+ host-built tables) are uploaded
  l.lut = align_up(b.plans.size() * sizeof(ImgPlan), 256);
  l.tables = l.lut + align_up(3 * 256 * sizeof(float), 256);
  l.tmp = l.tables + align_up(b.table_ints * sizeof(int), 256);
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

void set_device_resample_tables(int on) { g_device_tables = on != 0; }

// test hook: the same table built by the device kernel (bounds [count][2], kk [count][ksize] device int32 buffers)
int resample_table_device(int in_size, int out_size, int first, int count, int* bounds_dev, int* kk_dev, hipStream_t stream) {
  EZ_REQUIRE(in_size > 0 && out_size > 0 && first >= 0 && count > 0 && count <= 256 && first + count <= out_size && bounds_dev && kk_dev,
             "resample_table_device: bad argument");
  const AxisGeom g = axis_geom(in_size, out_size, first, count);
  ImgPlan p;
  std::memset(&p, 0, sizeof(p));
  p.in_w = in_size; p.ksize_h = g.ksize;
  p.first[0] = first; p.identity[0] = g.identity; p.scale[0] = g.scale; p.support[0] = g.support; p.ss[0] = g.ss;
  p.hb = 0; p.hk = 0;
  ImgPlan* dp = nullptr;
  EZ_HIP(hipMalloc(reinterpret_cast<void**>(&dp), sizeof(ImgPlan)));
  EZ_HIP(hipMemcpyAsync(dp, &p, sizeof(p), hipMemcpyHostToDevice, stream));
  EZ_HIP(hipStreamSynchronize(stream));
  hipLaunchKernelGGL(resample_tables_split_kernel, dim3(1), dim3(256), 0, stream, dp, bounds_dev, kk_dev, count);
  EZ_HIP(hipStreamSynchronize(stream));
  EZ_HIP(hipFree(dp));
  EZ_LAUNCH_CHECK();
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
  const size_t up_bytes = b.tables.empty() ? l.tables : l.tmp;        // device-built tables are not uploaded
  if (stg.pending) { EZ_HIP(hipEventSynchronize(stg.ev)); stg.pending = false; }
  if (stg.bytes < up_bytes) {
    if (stg.host) EZ_HIP(hipHostFree(stg.host));
    stg.host = nullptr; stg.bytes = 0;
    EZ_HIP(hipHostMalloc(reinterpret_cast<void**>(&stg.host), up_bytes + (up_bytes >> 1), hipHostMallocDefault));
    stg.bytes = up_bytes + (up_bytes >> 1);
  }
  if (stg.ev == nullptr) EZ_HIP(hipEventCreateWithFlags(&stg.ev, hipEventDisableTiming));
  std::memcpy(stg.host + l.plans, b.plans.data(), b.plans.size() * sizeof(ImgPlan));
  std::memcpy(stg.host + l.lut, lut, sizeof(lut));
  if (!b.tables.empty()) std::memcpy(stg.host + l.tables, b.tables.data(), b.tables.size() * sizeof(int));
  char* w = static_cast<char*>(ws);
  EZ_HIP(hipMemcpyAsync(w, stg.host, up_bytes, hipMemcpyHostToDevice, stream));
  EZ_HIP(hipEventRecord(stg.ev, stream));
  stg.pending = true;
  const ImgPlan* dplans = reinterpret_cast<const ImgPlan*>(w + l.plans);
  int* dtables = reinterpret_cast<int*>(w + l.tables);
  if (b.tables.empty())      // window tables on the device: 2 x crop threads per image
    hipLaunchKernelGGL(resample_tables_kernel, dim3(2, n), dim3(256), 0, stream, dplans, dtables, crop);
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
