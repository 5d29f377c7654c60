// extern "C" surface of libezclip_hip.so (declared in include/ezclip.h).
#include <cstring>

#include <chrono>
#include <thread>

#include "model.h"

using namespace ezclip;

namespace {

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct Arena {
  char* base;
  size_t off = 0;
  explicit Arena(void* b) : base(reinterpret_cast<char*>(b)) {}
  float* takef(size_t n) {
    off = (off + 255) & ~size_t(255);
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += n * 4;
    return p;
  }
};

int sim_gemm(const float* a, int64_t lda, const float* b, int64_t ldb, int na, int nb, int k, const float* scale_log,
             float* out, int64_t ldc, const float* residual, hipStream_t st) {
  GemmArgs g;
  g.A = a; g.lda = lda; g.B = b; g.ldb = ldb; g.C = out; g.ldc = ldc;
  g.M = na; g.N = nb; g.K = k;
  g.scale_log = scale_log;
  g.R = residual; g.ldr = ldc;
  return gemm_nt(g, EZCLIP_F32, st);
}

struct NceWS {
  float *St, *Si, *dSt, *dSi, *lse_t, *lse_i, *rl_t, *rl_i, *IallT, *TallT, *dStT, *dSiT, *TlocT, *IlocT, *partial;
  int Np, np;
};
size_t layout_nce(int n, int N, int e, void* base, NceWS* out) {
  Arena a(base);
  NceWS w;
  w.Np = round_up(N, 32); w.np = round_up(n, 32);
  w.St = a.takef((size_t)n * w.Np); w.Si = a.takef((size_t)n * w.Np);
  w.dSt = a.takef((size_t)n * w.Np); w.dSi = a.takef((size_t)n * w.Np);
  w.lse_t = a.takef(n); w.lse_i = a.takef(n); w.rl_t = a.takef(n); w.rl_i = a.takef(n);
  w.IallT = a.takef((size_t)e * w.Np); w.TallT = a.takef((size_t)e * w.Np);
  w.dStT = a.takef((size_t)N * w.np); w.dSiT = a.takef((size_t)N * w.np);
  w.TlocT = a.takef((size_t)e * w.np); w.IlocT = a.takef((size_t)e * w.np);
  w.partial = a.takef(256);
  if (out) *out = w;
  return a.off + 256;
}

}  // namespace

#define API_TRY(expr)                \
  do {                               \
    int _rc = (expr);                \
    if (_rc != EZ_OK) return _rc;    \
  } while (0)

extern "C" {

const char* ezclip_last_error(void) { return ezclip::last_error(); }
const char* ezclip_version(void) { return "ezclip-hip 0.1 (gfx950)"; }

int ezclip_create(const ezclip_config* cfg, ezclip_handle* out) { return model_create(cfg, out, 0); }
int ezclip_create_ex(const ezclip_config* cfg, int text_arch, ezclip_handle* out) { return model_create(cfg, out, text_arch); }

void ezclip_destroy(ezclip_handle h) {
  if (h) for (hipEvent_t ev : h->progress_pool) (void)hipEventDestroy(ev);
  if (h && h->pm_host) (void)hipHostFree(h->pm_host);
  delete h;
}

int ezclip_num_params(ezclip_handle h) { return h ? (int)h->params.size() : 0; }

int ezclip_param_info(ezclip_handle h, int index, const char** name, int64_t* shape, int* ndim) {
  EZ_REQUIRE(h && index >= 0 && index < (int)h->params.size(), "ezclip_param_info: bad index %d", index);
  const auto& p = h->params[index];
  if (name) *name = p.name.c_str();
  if (ndim) *ndim = (int)p.shape.size();
  if (shape) for (size_t i = 0; i < p.shape.size(); ++i) shape[i] = p.shape[i];
  return EZ_OK;
}

int ezclip_bind_param(ezclip_handle h, const char* name, void* w, void* g, const int64_t* shape, int ndim) {
  EZ_REQUIRE(h && name && w, "ezclip_bind_param: null argument");
  auto it = h->index.find(name);
  EZ_REQUIRE(it != h->index.end(), "ezclip_bind_param: unknown parameter '%s'", name);
  auto& p = h->params[it->second];
  EZ_REQUIRE(ndim == (int)p.shape.size(), "ezclip_bind_param: %s expects %d dims, got %d", name, (int)p.shape.size(), ndim);
  for (int i = 0; i < ndim; ++i)
    EZ_REQUIRE(shape[i] == p.shape[i], "ezclip_bind_param: %s dim %d is %lld, expected %lld", name, i,
               (long long)shape[i], (long long)p.shape[i]);
  EZ_REQUIRE(((uintptr_t)w % 16) == 0 && ((uintptr_t)g % 16) == 0, "ezclip_bind_param: %s must be 16-byte aligned", name);
  if (p.w != reinterpret_cast<float*>(w)) h->weights_fresh = false;   // grad-only rebinding keeps the packs valid
  p.w = reinterpret_cast<float*>(w);
  p.g = reinterpret_cast<float*>(g);
  return EZ_OK;
}

size_t ezclip_shadow_bytes(ezclip_handle h, int with_backward) {
  return h ? model_shadow_layout(h, nullptr, with_backward != 0) : 0;
}

int ezclip_set_shadow(ezclip_handle h, void* buf, size_t bytes, int with_backward) {
  EZ_REQUIRE(h && buf, "ezclip_set_shadow: null argument");
  EZ_REQUIRE(((uintptr_t)buf % 256) == 0, "ezclip_set_shadow: buffer must be 256-byte aligned");
  const size_t need = model_shadow_layout(h, nullptr, with_backward != 0);
  EZ_REQUIRE(bytes >= need, "ezclip_set_shadow: buffer too small (%zu < %zu)", bytes, need);
  model_shadow_layout(h, reinterpret_cast<char*>(buf), with_backward != 0);
  h->shadow = buf; h->shadow_bytes = bytes; h->shadow_backward = with_backward != 0;
  h->weights_fresh = false;
  return EZ_OK;
}

int ezclip_refresh_weights(ezclip_handle h, void* stream) {
  EZ_REQUIRE(h, "ezclip_refresh_weights: null handle");
  return model_refresh_weights(h, S(stream));
}

size_t ezclip_image_workspace_bytes(ezclip_handle h, int batch, int save) {
  return h ? image_workspace_bytes(h, batch, save != 0) : 0;
}
size_t ezclip_text_workspace_bytes(ezclip_handle h, int batch, int seq_len, int save) {
  return h ? text_workspace_bytes(h, batch, seq_len, save != 0) : 0;
}

int ezclip_encode_image(ezclip_handle h, const float* pixels, int batch, float* out, void* ws, size_t ws_bytes, int save,
                        void* stream) {
  EZ_REQUIRE(h, "ezclip_encode_image: null handle");
  return encode_image(h, pixels, batch, out, ws, ws_bytes, save != 0, S(stream));
}

int ezclip_encode_text(ezclip_handle h, const int64_t* ids, int batch, int seq_len, float* out, void* ws,
                       size_t ws_bytes, int save, void* stream) {
  EZ_REQUIRE(h, "ezclip_encode_text: null handle");
  return encode_text(h, ids, batch, seq_len, out, ws, ws_bytes, save != 0, S(stream));
}

int ezclip_similarity(const float* a, const float* b, int na, int nb, int e, const float* logit_scale, float* out,
                      void* stream) {
  EZ_REQUIRE(a && b && out && na > 0 && nb > 0, "ezclip_similarity: null/empty argument");
  return sim_gemm(a, e, b, e, na, nb, e, logit_scale, out, nb, nullptr, S(stream));
}

int ezclip_infonce_from_logits(const float* logits, int n, float* loss, float* scratch, void* stream) {
  EZ_REQUIRE(logits && loss && scratch && n > 0, "ezclip_infonce_from_logits: null/empty argument");
  hipStream_t st = S(stream);
  float *lse_r = scratch, *rl = scratch + n, *lse_c = scratch + 2 * n, *cl = scratch + 3 * n;
  API_TRY(ce_rows_fwd(logits, n, n, n, 0, lse_r, rl, st));
  API_TRY(ce_cols_fwd(logits, n, n, lse_c, cl, st));
  API_TRY(sum_scaled(rl, n, 0.5f / n, loss, 0, st));
  API_TRY(sum_scaled(cl, n, 0.5f / n, loss, 1, st));
  return EZ_OK;
}

int ezclip_infonce_from_logits_bwd(const float* logits, int n, const float* grad_out, float* dlogits, float* scratch,
                                   void* stream) {
  EZ_REQUIRE(logits && dlogits && scratch && n > 0, "ezclip_infonce_from_logits_bwd: null/empty argument");
  hipStream_t st = S(stream);
  float *lse_r = scratch, *rl = scratch + n, *lse_c = scratch + 2 * n, *cl = scratch + 3 * n;
  API_TRY(ce_rows_fwd(logits, n, n, n, 0, lse_r, rl, st));
  API_TRY(ce_cols_fwd(logits, n, n, lse_c, cl, st));
  API_TRY(infonce_dlogits(logits, n, lse_r, lse_c, grad_out, 0.5f / n, dlogits, st));
  return EZ_OK;
}

int ezclip_cross_entropy_diag(const float* logits, int rows, int cols, int64_t ld, float* loss, float* scratch, void* stream) {
  EZ_REQUIRE(logits && loss && scratch, "ezclip_cross_entropy_diag: null argument");
  EZ_REQUIRE(rows > 0 && cols >= rows && ld >= cols, "ezclip_cross_entropy_diag: bad shape rows=%d cols=%d ld=%lld (the labels are arange(rows))",
             rows, cols, (long long)ld);
  hipStream_t st = S(stream);
  API_TRY(ce_rows_fwd(logits, ld, rows, cols, 0, scratch, scratch + rows, st));
  API_TRY(sum_scaled(scratch + rows, rows, 1.0f / rows, loss, 0, st));
  return EZ_OK;
}

int ezclip_cross_entropy_diag_bwd(const float* logits, int rows, int cols, int64_t ld, const float* grad_out, const float* scratch,
                                  float* dlogits, void* stream) {
  EZ_REQUIRE(logits && scratch && dlogits, "ezclip_cross_entropy_diag_bwd: null argument");
  EZ_REQUIRE(rows > 0 && cols >= rows && ld >= cols, "ezclip_cross_entropy_diag_bwd: bad shape rows=%d cols=%d ld=%lld", rows, cols, (long long)ld);
  API_TRY(ce_rows_bwd(logits, ld, rows, cols, 0, scratch, grad_out, 1.0f / rows, dlogits, cols, 0, S(stream)));
  return EZ_OK;
}

size_t ezclip_infonce_workspace_bytes(int n_local, int n_global, int e) {
  return layout_nce(n_local, n_global, e, nullptr, nullptr);
}

size_t ezclip_infonce_tiled_workspace_bytes(int n_local, int n_global, int e) {
  return infonce_tiled_eligible(e) ? infonce_tiled_workspace_bytes(n_local, n_global, e) : 0;
}

int ezclip_infonce_tiled(const float* T, const float* I, int n, int N, int off, int e, const float* ls, float grad_scale,
                         int split_operands, float* loss, float* dT, float* dI, float* dls, void* wsp, size_t ws_bytes,
                         void* stream) {
  EZ_REQUIRE(T && I && ls && loss && wsp, "ezclip_infonce_tiled: null argument");
  EZ_REQUIRE(n > 0 && N >= n && off >= 0 && off + n <= N, "ezclip_infonce_tiled: bad shard n=%d N=%d offset=%d", n, N, off);
  EZ_REQUIRE(((uintptr_t)wsp % 256) == 0, "ezclip_infonce_tiled: workspace must be 256-byte aligned");
  EZ_REQUIRE((dT == nullptr) == (dI == nullptr) && (dT == nullptr) == (dls == nullptr),
             "ezclip_infonce_tiled: d_text/d_image/d_logit_scale must be all set or all NULL");
  return infonce_tiled(T, I, n, N, off, e, ls, grad_scale, split_operands ? 1 : 0, loss, dT, dI, dls, wsp, ws_bytes, S(stream));
}

int ezclip_infonce_fused(const float* T, const float* I, int n, int N, int off, int e, const float* ls,
                         float grad_scale, float* loss, float* dT, float* dI, float* dls, void* wsp, size_t ws_bytes,
                         void* stream) {
  EZ_REQUIRE(T && I && loss && wsp, "ezclip_infonce_fused: null argument");
  EZ_REQUIRE(n > 0 && N >= n && off >= 0 && off + n <= N, "ezclip_infonce_fused: bad shard n=%d N=%d offset=%d", n, N, off);
  EZ_REQUIRE(e % 32 == 0, "ezclip_infonce_fused: embed dim %d must be a multiple of 32", e);
  EZ_REQUIRE(((uintptr_t)wsp % 256) == 0, "ezclip_infonce_fused: workspace must be 256-byte aligned");
  NceWS w;
  const size_t need = layout_nce(n, N, e, wsp, &w);
  EZ_REQUIRE(ws_bytes >= need, "ezclip_infonce_fused: workspace too small (%zu < %zu)", ws_bytes, need);
  hipStream_t st = S(stream);
  const float* Tl = T + (size_t)off * e;
  const float* Il = I + (size_t)off * e;
  // S_t = s * T_loc I_all^T ; S_i = s * I_loc T_all^T         (appzoo/clip/model.py:148, both directions)
  API_TRY(sim_gemm(Tl, e, I, e, n, N, e, ls, w.St, w.Np, nullptr, st));
  API_TRY(sim_gemm(Il, e, T, e, n, N, e, ls, w.Si, w.Np, nullptr, st));
  API_TRY(ce_rows_fwd(w.St, w.Np, n, N, off, w.lse_t, w.rl_t, st));      // model.py:154-160
  API_TRY(ce_rows_fwd(w.Si, w.Np, n, N, off, w.lse_i, w.rl_i, st));
  API_TRY(sum_scaled(w.rl_t, n, 0.5f / n, loss, 0, st));
  API_TRY(sum_scaled(w.rl_i, n, 0.5f / n, loss, 1, st));
  if (dT == nullptr && dI == nullptr && dls == nullptr) return EZ_OK;
  EZ_REQUIRE(dT && dI && dls, "ezclip_infonce_fused: d_text/d_image/d_logit_scale must be all set or all NULL");
  const float c = grad_scale * 0.5f / n;
  EZ_HIP(hipMemsetAsync(w.dSt, 0, (size_t)n * w.Np * 4, st));
  EZ_HIP(hipMemsetAsync(w.dSi, 0, (size_t)n * w.Np * 4, st));
  API_TRY(ce_rows_bwd(w.St, w.Np, n, N, off, w.lse_t, nullptr, c, w.dSt, w.Np, 0, st));
  API_TRY(ce_rows_bwd(w.Si, w.Np, n, N, off, w.lse_i, nullptr, c, w.dSi, w.Np, 0, st));
  // d logit_scale = sum dS .* S   (S = exp(ls) * X  =>  dS/dls = S)
  API_TRY(dot_scaled(w.dSt, w.St, (int64_t)n * w.Np, 1.0f, w.partial, dls, 0, st));
  API_TRY(dot_scaled(w.dSi, w.Si, (int64_t)n * w.Np, 1.0f, w.partial, dls, 1, st));
  // operand transposes (f32, tiny)
  API_TRY(transpose_cast(I, e, N, e, w.IallT, w.Np, EZCLIP_F32, st));
  API_TRY(transpose_cast(T, e, N, e, w.TallT, w.Np, EZCLIP_F32, st));
  API_TRY(transpose_cast(Tl, e, n, e, w.TlocT, w.np, EZCLIP_F32, st));
  API_TRY(transpose_cast(Il, e, n, e, w.IlocT, w.np, EZCLIP_F32, st));
  API_TRY(transpose_cast(w.dSt, w.Np, n, N, w.dStT, w.np, EZCLIP_F32, st));
  API_TRY(transpose_cast(w.dSi, w.Np, n, N, w.dSiT, w.np, EZCLIP_F32, st));
  EZ_HIP(hipMemsetAsync(dT, 0, (size_t)N * e * 4, st));
  EZ_HIP(hipMemsetAsync(dI, 0, (size_t)N * e * 4, st));
  // remote-column terms: dI_all = s * dS_t^T T_loc ; dT_all = s * dS_i^T I_loc
  API_TRY(sim_gemm(w.dStT, w.np, w.TlocT, w.np, N, e, w.np, ls, dI, e, nullptr, st));
  API_TRY(sim_gemm(w.dSiT, w.np, w.IlocT, w.np, N, e, w.np, ls, dT, e, nullptr, st));
  // local-row terms (accumulated through the residual input): dT_loc += s * dS_t I_all ; dI_loc += s * dS_i T_all
  float* dTl = dT + (size_t)off * e;
  float* dIl = dI + (size_t)off * e;
  API_TRY(sim_gemm(w.dSt, w.Np, w.IallT, w.Np, n, e, w.Np, ls, dTl, e, dTl, st));
  API_TRY(sim_gemm(w.dSi, w.Np, w.TallT, w.Np, n, e, w.Np, ls, dIl, e, dIl, st));
  return EZ_OK;
}

int ezclip_backward_image(ezclip_handle h, const float* pixels, int batch, const float* d_emb, void* ws, size_t ws_bytes,
                          void* stream) {
  EZ_REQUIRE(h, "ezclip_backward_image: null handle");
  return backward_image(h, pixels, batch, d_emb, ws, ws_bytes, S(stream));
}
int ezclip_backward_text(ezclip_handle h, const int64_t* ids, int batch, int seq_len, const float* d_emb, void* ws,
                         size_t ws_bytes, void* stream) {
  EZ_REQUIRE(h, "ezclip_backward_text: null handle");
  return backward_text(h, ids, batch, seq_len, d_emb, ws, ws_bytes, S(stream));
}

int ezclip_pack_text_meta(ezclip_handle h, const int64_t* ids, const int64_t* attention_mask, int batch, int seq_len,
                          int32_t* rowmap, int32_t* cu, int32_t* lens, int* ticket, void* stream) {
  EZ_REQUIRE(h && ticket, "ezclip_pack_text_meta: null handle / ticket");
  if (!h->pm_host) {
    void* p = nullptr;
    EZ_HIP(hipHostMalloc(&p, 8 * 4 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    h->pm_host = (int*)p;
    for (int i = 0; i < 32; ++i) h->pm_host[i] = 0;
    void* d = nullptr;
    EZ_HIP(hipHostGetDevicePointer(&d, p, 0));
    h->pm_dev = (int*)d;
  }
  const int t = ++h->pm_ticket;          // (tickets start at 1: a zeroed slot never matches)
  const int slot = t & 7;
  int rc = pack_text_meta(ids, attention_mask, batch, seq_len, rowmap, cu, lens, h->pm_dev + 4 * slot, t, S(stream));
  if (rc != EZ_OK) return rc;
  *ticket = t;
  return EZ_OK;
}

int ezclip_pack_text_meta_result(ezclip_handle h, int ticket, int* rows, int* longest, int* prefix) {
  EZ_REQUIRE(h && h->pm_host && rows && longest && prefix, "ezclip_pack_text_meta_result: nothing was enqueued");
  EZ_REQUIRE(ticket > 0 && ticket <= h->pm_ticket && h->pm_ticket - ticket < 8,
             "ezclip_pack_text_meta_result: ticket %d is not among the last 8 (newest %d)", ticket, h->pm_ticket);
  volatile int* s = h->pm_host + 4 * (ticket & 7);
  // the kernel writes the three values, fences, then the ticket word: poll it (the stream is NOT synchronised; other work
  // queued behind the kernel keeps the device busy while the host waits for this one launch)
  const auto t0 = std::chrono::steady_clock::now();
  long spins = 0;
  while (__atomic_load_n((const int*)(s + 3), __ATOMIC_ACQUIRE) != ticket) {
    if ((++spins & 1023) == 0) {
      const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (sec > 120.0) { set_error("ezclip_pack_text_meta_result: ticket %d did not arrive within 120 s", ticket); return EZ_ERR_STATE; }
      if (sec > 0.002) std::this_thread::yield();
    }
  }
  *rows = s[0]; *longest = s[1]; *prefix = s[2];
  return EZ_OK;
}

int ezclip_set_backward_progress(ezclip_handle h, ezclip_progress_fn fn, void* user) {
  EZ_REQUIRE(h, "ezclip_set_backward_progress: null handle");
  h->progress_fn = fn;
  h->progress_user = user;
  return EZ_OK;
}

// Progress of a backward pass: the legacy host callback (ezclip_set_backward_progress) and / or the event log
// (ezclip_backward_progress_events): one hipEventRecord on the stream that has just been given the group's last kernel.  The
// events come from a ring of at most kProgressRing (a pass has <= 2 x (layers + 2) groups, far fewer): the cursor wraps, so a
// caller that drains every step without re-arming the log does not grow the pool, and an event is re-recorded only after
// kProgressRing later records -- long after the drained item that referred to it has been waited on.  An undrained log is
// capped at the ring's size too.  A dropped item (event create / record failed, log full) is reported through the thread's
// error message: the pass itself goes on, the caller sees the group missing from the drain.
static constexpr size_t kProgressRing = 256;
void ezclip_model::progress(int tower, int stage, hipStream_t stream) const {
  if (progress_fn) progress_fn(progress_user, tower, stage);
  if (!progress_log) return;
  if (progress_items.size() >= kProgressRing) {
    set_error("backward progress log full (%zu undrained items): group (%d, %d) dropped -- drain after every pass", progress_items.size(), tower, stage);
    return;
  }
  if (progress_next >= kProgressRing) progress_next = 0;
  if (progress_next >= progress_pool.size()) {
    hipEvent_t ev = nullptr;
    const hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) { set_error("backward progress: hipEventCreate failed (%s): group (%d, %d) dropped", hipGetErrorString(e), tower, stage); return; }
    progress_pool.push_back(ev);
  }
  hipEvent_t ev = progress_pool[progress_next];
  const hipError_t e = hipEventRecord(ev, stream);
  if (e != hipSuccess) { set_error("backward progress: hipEventRecord failed (%s): group (%d, %d) dropped", hipGetErrorString(e), tower, stage); return; }
  ++progress_next;
  progress_items.push_back(ProgressItem{tower, stage, ev});
}

int ezclip_backward_progress_events(ezclip_handle h, int enable) {
  EZ_REQUIRE(h, "ezclip_backward_progress_events: null handle");
  h->progress_log = enable != 0;
  h->progress_items.clear();
  h->progress_next = 0;
  return EZ_OK;
}

int ezclip_backward_progress_drain(ezclip_handle h, int* towers, int* stages, void** events, int max_items, int* n_items) {
  EZ_REQUIRE(h && towers && stages && events && n_items && max_items > 0, "ezclip_backward_progress_drain: null argument");
  const int n = (int)h->progress_items.size();
  EZ_REQUIRE(n <= max_items, "ezclip_backward_progress_drain: %d items logged, room for %d", n, max_items);
  for (int i = 0; i < n; ++i) {
    towers[i] = h->progress_items[i].tower;
    stages[i] = h->progress_items[i].stage;
    events[i] = (void*)h->progress_items[i].ev;
  }
  *n_items = n;
  h->progress_items.clear();      // (the events stay valid and un-re-recorded for the next kProgressRing - n records: the ring's cursor is NOT reset here)
  return EZ_OK;
}

int ezclip_stream_wait_event(void* stream, void* event) {
  EZ_REQUIRE(event, "ezclip_stream_wait_event: null event");
  EZ_HIP(hipStreamWaitEvent(S(stream), (hipEvent_t)event, 0));
  return EZ_OK;
}

int ezclip_set_option(ezclip_handle h, int key, double value) {
  EZ_REQUIRE(h, "ezclip_set_option: null handle");
  switch (key) {
    case EZCLIP_OPT_TEXT_POOLER: h->opt_text_pooler = value != 0.0; return EZ_OK;
    case EZCLIP_OPT_VISION_FROZEN: h->opt_vision_frozen = value != 0.0; return EZ_OK;
    case EZCLIP_OPT_TEXT_LN_EPS:
      EZ_REQUIRE(value > 0.0 && value < 1.0, "ezclip_set_option: layer_norm_eps %g", value);
      h->text_ln_eps = (float)value;
      return EZ_OK;
    case EZCLIP_OPT_TEXT_PAD_ID: h->text_pad_id = (int64_t)value; return EZ_OK;
    case EZCLIP_OPT_BLOCK_LN_EPS:
      EZ_REQUIRE(value > 0.0 && value < 1.0, "ezclip_set_option: block LayerNorm eps %g", value);
      h->block_ln_eps = (float)value;
      return EZ_OK;
    case EZCLIP_OPT_TEXT_EOT_ID:
      EZ_REQUIRE(h->text_arch == 1, "ezclip_set_option: EZCLIP_OPT_TEXT_EOT_ID needs a handle created with EZCLIP_TEXT_CLIP");
      EZ_REQUIRE(value >= -1.0 && value < (double)h->cfg.vocab_size, "ezclip_set_option: token id %g outside the vocabulary", value);
      h->text_eot_id = (int64_t)value;
      return EZ_OK;
    default: break;
  }
  set_error("ezclip_set_option: unknown key %d", key);
  return EZ_ERR_INVALID;
}

int ezclip_encode_text_ex(ezclip_handle h, const int64_t* ids, const int64_t* position_ids, const int64_t* token_type_ids,
                          const int64_t* attention_mask, int batch, int seq_len, float* out, void* ws, size_t ws_bytes,
                          int save, void* stream) {
  EZ_REQUIRE(h, "ezclip_encode_text_ex: null handle");
  TextExtras ex;
  ex.pos_ids = position_ids; ex.type_ids = token_type_ids; ex.attn_mask = attention_mask;
  return encode_text(h, ids, batch, seq_len, out, ws, ws_bytes, save != 0, S(stream), &ex);
}
int ezclip_encode_text_packed(ezclip_handle h, const int64_t* ids, const int64_t* position_ids, const int64_t* token_type_ids,
                              const int64_t* attention_mask, const int32_t* rowmap, const int32_t* cu, const int32_t* lens,
                              int batch, int seq_len, int packed_rows, int max_len, float* out, void* ws, size_t ws_bytes,
                              int save, void* stream) {
  EZ_REQUIRE(h && rowmap && cu && lens, "ezclip_encode_text_packed: null argument");
  TextExtras ex;
  ex.pos_ids = position_ids; ex.type_ids = token_type_ids; ex.attn_mask = attention_mask;
  ex.rowmap = rowmap; ex.cu = cu; ex.lens = lens; ex.packed_rows = packed_rows; ex.max_len = max_len;
  return encode_text(h, ids, batch, seq_len, out, ws, ws_bytes, save != 0, S(stream), &ex);
}
int ezclip_backward_text_packed(ezclip_handle h, const int64_t* ids, const int64_t* position_ids, const int64_t* token_type_ids,
                                const int64_t* attention_mask, const int32_t* rowmap, const int32_t* cu, const int32_t* lens,
                                int batch, int seq_len, int packed_rows, int max_len, const float* d_emb, void* ws,
                                size_t ws_bytes, void* stream) {
  EZ_REQUIRE(h && rowmap && cu && lens, "ezclip_backward_text_packed: null argument");
  TextExtras ex;
  ex.pos_ids = position_ids; ex.type_ids = token_type_ids; ex.attn_mask = attention_mask;
  ex.rowmap = rowmap; ex.cu = cu; ex.lens = lens; ex.packed_rows = packed_rows; ex.max_len = max_len;
  return backward_text(h, ids, batch, seq_len, d_emb, ws, ws_bytes, S(stream), &ex);
}
int ezclip_backward_text_ex(ezclip_handle h, const int64_t* ids, const int64_t* position_ids, const int64_t* token_type_ids,
                            const int64_t* attention_mask, int batch, int seq_len, const float* d_emb, void* ws,
                            size_t ws_bytes, void* stream) {
  EZ_REQUIRE(h, "ezclip_backward_text_ex: null handle");
  TextExtras ex;
  ex.pos_ids = position_ids; ex.type_ids = token_type_ids; ex.attn_mask = attention_mask;
  return backward_text(h, ids, batch, seq_len, d_emb, ws, ws_bytes, S(stream), &ex);
}

int ezclip_set_text_dropout(ezclip_handle h, float hidden_p, float attention_p, uint64_t seed) {
  EZ_REQUIRE(h, "ezclip_set_text_dropout: null handle");
  EZ_REQUIRE(hidden_p >= 0.f && hidden_p < 1.f && attention_p >= 0.f && attention_p < 1.f,
             "ezclip_set_text_dropout: probabilities must be in [0, 1) (got %g, %g)", (double)hidden_p, (double)attention_p);
  h->drop_hidden = hidden_p;
  h->drop_attn = attention_p;
  h->drop_seed = seed;
  return EZ_OK;
}

size_t ezclip_preprocess_workspace_bytes(const ezclip_image_desc* desc, int n, int size, int crop) {
  return preprocess_workspace_bytes(desc, n, size, crop);
}
int ezclip_op_resample_table(int in_size, int out_size, int first, int count, int* ksize, int* bounds_host, int* kk_host,
                             int kk_capacity) {
  return resample_table(in_size, out_size, first, count, ksize, bounds_host, kk_host, kk_capacity);
}
int ezclip_op_resample_table_device(int in_size, int out_size, int first, int count, int* bounds_dev, int* kk_dev, void* stream) {
  return resample_table_device(in_size, out_size, first, count, bounds_dev, kk_dev, S(stream));
}
int ezclip_preprocess_images(const uint8_t* packed, const ezclip_image_desc* desc, int n, int size, int crop, const float* mean,
                             const float* stdv, float* out, void* ws, size_t ws_bytes, void* stream) {
  EZ_REQUIRE(mean && stdv, "ezclip_preprocess_images: null mean / std");
  return preprocess_images(packed, desc, n, size, crop, mean, stdv, out, ws, ws_bytes, S(stream));
}

int ezclip_recall_ranks(const float* text, const float* image, int n, int e, int32_t* rank_out, float* scratch,
                        void* stream) {
  EZ_REQUIRE(text && image && rank_out && scratch && n > 0, "ezclip_recall_ranks: null/empty argument");
  API_TRY(sim_gemm(text, e, image, e, n, n, e, nullptr, scratch, n, nullptr, S(stream)));   // evaluator.py:50
  return recall_ranks(scratch, n, n, 0, rank_out, S(stream));
}

int ezclip_recall_ranks_rows(const float* text_rows, const float* image, int rows, int row0, int n, int e, int32_t* rank_out,
                             float* scratch, void* stream) {
  EZ_REQUIRE(text_rows && image && rank_out && scratch && rows > 0 && row0 >= 0 && row0 + rows <= n,
             "ezclip_recall_ranks_rows: bad block rows=%d row0=%d n=%d", rows, row0, n);
  API_TRY(sim_gemm(text_rows, e, image, e, rows, n, e, nullptr, scratch, n, nullptr, S(stream)));
  return recall_ranks(scratch, rows, n, row0, rank_out, S(stream));
}

int ezclip_recall_paired_scores(const float* text, const float* image, int n, int e, float* paired, void* stream) {
  EZ_REQUIRE(text && image && paired && n > 0 && e > 0, "ezclip_recall_paired_scores: null/empty argument");
  GemmRankArgs g;
  g.A = text; g.lda = e; g.B = image; g.ldb = e; g.M = n; g.N = n; g.K = e;
  g.rank_mode = 1; g.rank_row0 = 0; g.rank_diag_out = paired;
  return gemm_nt_rank(g, S(stream));
}

int ezclip_recall_ranks_fused(const float* text_rows, const float* image, int rows, int row0, int n, int e, const float* paired,
                              int32_t* rank_t2i, int32_t* rank_i2t, void* stream) {
  EZ_REQUIRE(text_rows && image && paired && rank_t2i && rows > 0 && row0 >= 0 && row0 + rows <= n,
             "ezclip_recall_ranks_fused: bad block rows=%d row0=%d n=%d", rows, row0, n);
  EZ_HIP(hipMemsetAsync(rank_t2i, 0, (size_t)rows * sizeof(int32_t), S(stream)));
  GemmRankArgs g;
  g.A = text_rows; g.lda = e; g.B = image; g.ldb = e; g.M = rows; g.N = n; g.K = e;
  g.rank_mode = 2; g.rank_row0 = row0; g.rank_diag = paired; g.rank_rows = rank_t2i; g.rank_cols = rank_i2t;
  return gemm_nt_rank(g, S(stream));
}

// ---- ModifiedResNet training path, operator level (tests; the tower's training orchestration will call the same functions)
size_t ezclip_op_rn_bn_scratch_bytes(int64_t rows, int cp) { return rn_bn_scratch_bytes(rows, cp); }
int ezclip_op_rn_bn_train_fwd(const void* z, int64_t rows, int c, int cp, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, float momentum, float eps, const void* residual, int relu, void* y, float* mean,
                              float* rstd, float* scratch, int dtype, void* stream) {
  return rn_bn_train_fwd(z, rows, c, cp, gamma, beta, running_mean, running_var, momentum, eps, residual, relu, y, mean, rstd, scratch, dtype,
                         S(stream));
}
int ezclip_op_rn_bn_train_bwd(const void* dy, const void* y, const void* z, int64_t rows, int c, int cp, const float* gamma,
                              const float* mean, const float* rstd, void* dz, void* dres, float* dgamma, float* dbeta, int accumulate,
                              float* scratch, int dtype, void* stream) {
  return rn_bn_train_bwd(dy, y, z, rows, c, cp, gamma, mean, rstd, dz, dres, dgamma, dbeta, accumulate, scratch, dtype, S(stream));
}
int ezclip_op_rn_avgpool2_bwd(const void* dy, int b, int h, int w, int cp, void* dx, int dtype, void* stream) {
  return rn_avgpool2_bwd(dy, b, h, w, cp, dx, dtype, S(stream));
}
int ezclip_op_rn_im2col3x3(const void* x, int b, int h, int w, int cp, void* col, int dtype, void* stream) {
  return rn_im2col3x3(x, b, h, w, cp, col, dtype, S(stream));
}
int ezclip_op_rn_pack_conv_dgrad(const float* w, int o, int i, int k, int opad, int ipad, void* dst, int dtype, void* stream) {
  return rn_pack_conv_dgrad(w, o, i, k, opad, ipad, dst, dtype, S(stream));
}
int ezclip_op_rn_unpack_wgrad(const float* dwp, int64_t ldp, int o, int i, int k, int cp, int accumulate, float* dw, void* stream) {
  return rn_unpack_wgrad(dwp, ldp, o, i, k, cp, accumulate, dw, S(stream));
}
// the implicit 3x3 convolution (pad 1, stride 1) of the tower on NHWC rows: out [b*h*w, n] = conv(x [b*h*w, cp]; wp [n][9*cp])
int ezclip_op_conv3x3_nhwc(const void* x, int b, int h, int w, int cp, const void* wp, int n, void* out, const void* zero256, int dtype,
                           void* stream) {
  EZ_REQUIRE(x && wp && out && zero256 && b > 0 && h > 0 && w > 0 && cp % 64 == 0 && n % 64 == 0, "ezclip_op_conv3x3_nhwc: bad shape");
  GemmArgs g;
  g.A = x; g.lda = cp; g.B = wp; g.ldb = 9 * (int64_t)cp; g.C = out; g.ldc = n;
  g.M = b * h * w; g.N = n; g.K = 9 * cp;
  g.conv_H = h; g.conv_W = w; g.conv_C = cp; g.conv_zero = zero256;
  return gemm_nt(g, dtype, S(stream));
}

int ezclip_debug_set(int key, int value) {
  if (key == 0) { set_gemm_variant(value); return EZ_OK; }
  if (key == 1) { set_attention_variant(value); return EZ_OK; }
  if (key == 2) { set_fold_layernorm(value); return EZ_OK; }
  if (key == 3) { set_cls_last(value); return EZ_OK; }
  if (key == 4) { set_cls_last_train(value); return EZ_OK; }
  if (key == 5) { set_device_resample_tables(value); return EZ_OK; }
  if (key == 6) { set_gemm_raster(value); return EZ_OK; }
  if (key == 7) { set_fuse_bert_qkv(value); return EZ_OK; }
  if (key == 8) { set_cls_q_only(value); return EZ_OK; }
  if (key == 9) { set_attention_short_tail(value); return EZ_OK; }
  if (key == 10) { set_rn_buffer_bound_mib(value); return EZ_OK; }
  if (key == 11) { set_attention_bwd_once(value); return EZ_OK; }
  if (key == 12) { set_gemm_dephase(value); return EZ_OK; }
  set_error("ezclip_debug_set: unknown key %d", key);
  return EZ_ERR_INVALID;
}
int ezclip_profile_begin(void) { return profile_begin(); }
int ezclip_profile_end(int kernel_class, double* ms, double* work, int* launches) {
  return profile_end(kernel_class, ms, work, launches);
}

// ---- operator-level -------------------------------------------------------------------
int ezclip_op_gemm_nt(const void* a, int64_t lda, const void* b, int64_t ldb, void* c, int64_t ldc, const float* bias,
                      const void* residual, int64_t ldr, int m, int n, int k, int act, int dtype, int out_f32,
                      void* stream) {
  GemmArgs g;
  g.A = a; g.lda = lda; g.B = b; g.ldb = ldb; g.C = c; g.ldc = ldc;
  g.bias = bias; g.R = residual; g.ldr = ldr;
  g.M = m; g.N = n; g.K = k; g.act = act; g.out_f32 = out_f32;
  return gemm_nt(g, dtype, S(stream));
}

int ezclip_op_gemm_nt_ex(const ezclip_gemm_desc* d, void* stream) {
  EZ_REQUIRE(d && d->a_dev && d->b_dev && d->c_dev, "ezclip_op_gemm_nt_ex: null argument");
  GemmArgs g;
  g.A = d->a_dev; g.lda = d->lda; g.B = d->b_dev; g.ldb = d->ldb; g.C = d->c_dev; g.ldc = d->ldc; g.C2 = d->c2_dev;
  g.bias = d->bias_dev; g.R = d->residual_dev; g.ldr = d->ldr; g.U = d->u_dev; g.ldu = d->ldu;
  g.ln_stats = d->ln_stats_dev; g.ln_c1 = d->ln_c1_dev; g.ln_c2 = d->ln_c2_dev;
  g.rowstat_part = d->rowstat_part_dev; g.colsum = d->colsum_dev;
  g.alpha = d->alpha; g.M = d->m; g.N = d->n; g.K = d->k; g.act = d->act; g.out_f32 = d->out_f32;
  const int fk = d->force_kernel;
  EZ_REQUIRE(fk == -1 || fk == 0 || fk == 2 || fk == 24, "ezclip_op_gemm_nt_ex: force_kernel %d", fk);
  if (fk >= 2) EZ_REQUIRE(gemm_nt_8p_eligible(g, d->dtype), "ezclip_op_gemm_nt_ex: the 8-phase kernel does not take this problem");
  set_gemm_variant(fk);
  const int rc = gemm_nt(g, d->dtype, S(stream));
  set_gemm_variant(-1);
  return rc;
}
int ezclip_op_layernorm_stats(const void* x, int64_t xs, float eps, int rows, int d, int dtype, float* stats, void* stream) {
  return layernorm_row_stats(x, xs, eps, rows, d, dtype, stats, S(stream));
}

int ezclip_op_gemm_tn(const void* a, int64_t lda, const void* b, int64_t ldb, float* c, int64_t ldc, int m, int n, int k,
                      int accumulate, int dtype, void* stream) {
  GemmTNArgs g;
  g.A = a; g.lda = lda; g.B = b; g.ldb = ldb; g.C = c; g.ldc = ldc;
  g.M = m; g.N = n; g.K = k; g.accumulate = accumulate;
  return gemm_tn(g, dtype, S(stream));
}

int ezclip_op_gemm_tn_conv3x3(const void* a, int64_t lda, const void* x, int images, int h, int w, int cp, float* c, int64_t ldc, int n,
                              int accumulate, int dtype, void* stream) {
  GemmTNArgs g;
  g.A = a; g.lda = lda; g.B = x; g.ldb = cp; g.C = c; g.ldc = ldc;
  g.M = images * h * w; g.N = n; g.K = 9 * cp; g.accumulate = accumulate;
  g.conv_H = h; g.conv_W = w; g.conv_C = cp;
  return gemm_tn(g, dtype, S(stream));
}

int ezclip_op_rn_wgrad3x3_c64(const void* x, const void* dz, int images, int h, int w, int cp, int opad, void* scratch, size_t scratch_bytes,
                              float* out, int64_t ldo, int accumulate, void* stream) {
  return rn_wgrad3x3_c64(x, dz, images, h, w, cp, opad, scratch, scratch_bytes, out, ldo, accumulate, S(stream));
}

int ezclip_op_rn_tn_skinny(const void* a, int64_t lda, const void* b, int64_t ldb, float* c, int64_t ldc, int64_t m, int n, int k, int accumulate,
                           void* scratch, size_t scratch_bytes, void* stream) {
  return rn_tn_skinny(a, lda, b, ldb, c, ldc, m, n, k, accumulate, scratch, scratch_bytes, S(stream));
}

int ezclip_op_layernorm(const void* x, int64_t xs, void* y, int64_t ys, const float* g, const float* b, float eps,
                        int rows, int d, int dtype, float* mean, float* rstd, void* stream) {
  return layernorm_fwd(x, xs, y, ys, g, b, eps, rows, d, dtype, mean, rstd, S(stream));
}

int ezclip_op_layernorm_bwd(const void* x, const void* dy, const float* g, const float* mean, const float* rstd, void* dx,
                            float* dg, float* db, int rows, int d, int dtype, void* stream) {
  return layernorm_bwd(x, d, dy, d, g, mean, rstd, dx, d, nullptr, 0, dg, db, rows, d, dtype, S(stream));
}

// causal mask / dropout of an op-level attention call (NULL: neither) -- arguments of the call, not process state
static int attn_opts(const ezclip_attention_opts* o, AttnArgs& a) {
  if (!o) return EZ_OK;
  if (!(o->dropout_p >= 0.f && o->dropout_p < 1.f)) { set_error("dropout probability %g outside [0, 1)", (double)o->dropout_p); return EZ_ERR_INVALID; }
  a.causal = o->causal != 0;
  if (o->dropout_p > 0.f) a.drop = make_drop(o->dropout_p, o->dropout_seed, o->dropout_site);
  return EZ_OK;
}
int ezclip_op_dropout(const void* x, const void* residual, void* y, int rows, int d, float p, uint64_t seed,
                      uint32_t site, int dtype, void* stream) {
  if (!(p > 0.f && p < 1.f)) { set_error("dropout probability %g outside (0, 1)", (double)p); return EZ_ERR_INVALID; }
  return dropout_rows(x, d, residual, d, y, d, rows, d, make_drop(p, seed, site), dtype, S(stream));
}
int ezclip_op_dropout_mask(float p, uint64_t seed, uint32_t site, int rows, int cols, uint8_t* keep, uint32_t* words,
                           void* stream) {
  if (!(p > 0.f && p < 1.f)) { set_error("dropout probability %g outside (0, 1)", (double)p); return EZ_ERR_INVALID; }
  return dropout_mask(keep, words, rows, cols, make_drop(p, seed, site), S(stream));
}

int ezclip_op_attention(const void* q, const void* k, const void* v, int64_t row_stride, void* ctx, int64_t ctx_stride,
                        const float* key_bias, float* lse, int batch, int seq_len, int heads, int dtype,
                        const ezclip_attention_opts* opts, void* stream) {
  AttnArgs a;
  API_TRY(attn_opts(opts, a));
  a.q = q; a.k = k; a.v = v; a.row_stride = row_stride; a.ctx = ctx; a.ctx_stride = ctx_stride;
  a.key_bias = key_bias; a.lse = lse; a.B = batch; a.L = seq_len; a.H = heads; a.scale = 0.125f;
  return attention_fwd(a, dtype, S(stream));
}

int ezclip_op_attention_bwd(const void* q, const void* k, const void* v, int64_t row_stride, const void* ctx,
                            const void* dctx, int64_t ctx_stride, const float* key_bias, const float* lse, void* dq,
                            void* dk, void* dv, int batch, int seq_len, int heads, int dtype, const ezclip_attention_opts* opts,
                            void* stream) {
  AttnBwdArgs b;
  API_TRY(attn_opts(opts, b.f));
  b.f.q = q; b.f.k = k; b.f.v = v; b.f.row_stride = row_stride; b.f.ctx = const_cast<void*>(ctx);
  b.f.ctx_stride = ctx_stride; b.f.key_bias = key_bias; b.f.lse = const_cast<float*>(lse);
  b.f.B = batch; b.f.L = seq_len; b.f.H = heads; b.f.scale = 0.125f;
  b.dctx = dctx; b.dq = dq; b.dk = dk; b.dv = dv;
  return attention_bwd(b, dtype, S(stream));
}

int ezclip_op_attention_bwd_bias(const void* q, const void* k, const void* v, int64_t row_stride, const void* ctx,
                                 const void* dctx, int64_t ctx_stride, const float* key_bias, const float* lse, void* dq,
                                 void* dk, void* dv, float* dbq, float* dbk, float* dbv, float* db_scratch, int batch,
                                 int seq_len, int heads, int dtype, const ezclip_attention_opts* opts, void* stream) {
  AttnBwdArgs b;
  API_TRY(attn_opts(opts, b.f));
  b.f.q = q; b.f.k = k; b.f.v = v; b.f.row_stride = row_stride; b.f.ctx = const_cast<void*>(ctx);
  b.f.ctx_stride = ctx_stride; b.f.key_bias = key_bias; b.f.lse = const_cast<float*>(lse);
  b.f.B = batch; b.f.L = seq_len; b.f.H = heads; b.f.scale = 0.125f;
  b.dctx = dctx; b.dq = dq; b.dk = dk; b.dv = dv;
  b.dbq = dbq; b.dbk = dbk; b.dbv = dbv; b.db_part = db_scratch;
  return attention_bwd(b, dtype, S(stream));
}

int ezclip_op_attention_cls(const void* q_cls, int64_t q_stride, const void* k, const void* v, int64_t row_stride,
                            const float* key_bias, void* ctx_cls, int64_t ctx_stride, int batch, int seq_len, int heads, int dtype,
                            void* stream) {
  AttnArgs a;
  a.k = k; a.v = v; a.row_stride = row_stride; a.key_bias = key_bias; a.B = batch; a.L = seq_len; a.H = heads; a.scale = 0.125f;
  return attention_cls_fwd(a, q_cls, q_stride, ctx_cls, ctx_stride, dtype, S(stream));
}
int ezclip_op_attention_cls_bwd(const void* q_cls, int64_t q_stride, const void* k, const void* v, int64_t row_stride,
                                const float* key_bias, const void* ctx_cls, const void* dctx_cls, int64_t ctx_stride, void* dq,
                                void* dk, void* dv, void* dq_cls, int64_t dq_stride, int batch, int seq_len, int heads, int dtype,
                                void* stream) {
  AttnBwdArgs b;
  b.f.k = k; b.f.v = v; b.f.row_stride = row_stride; b.f.key_bias = key_bias;
  b.f.B = batch; b.f.L = seq_len; b.f.H = heads; b.f.scale = 0.125f;
  b.dq = dq; b.dk = dk; b.dv = dv;
  return attention_cls_bwd(b, q_cls, q_stride, ctx_cls, dctx_cls, ctx_stride, dtype, S(stream), dq_cls, dq_stride);
}
int ezclip_op_cast_from_f32(const float* src, void* dst, int64_t n, int dtype, void* stream) {
  return cast_from_f32(src, dst, n, dtype, S(stream));
}
int ezclip_op_cast_to_f32(const void* src, float* dst, int64_t n, int dtype, void* stream) {
  return cast_to_f32(src, dst, n, dtype, S(stream));
}

}  // extern "C"
