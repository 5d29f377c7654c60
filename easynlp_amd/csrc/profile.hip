// Optional per-launch HIP-event timing of the hot kernels, on the stream they
// are launched on (bench.py's roofline leg).  Off by default: zero overhead.
#include <vector>

#include "ezclip_common.h"
#include "kernels.h"

namespace ezclip {
namespace {

struct Rec { hipEvent_t a, b; int cls; double work; };
struct State {
  bool on = false;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  std::vector<Rec> recs;
  hipEvent_t next() {
    if (used == pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      pool.push_back(e);
    }
    return pool[used++];
  }
};
State g;

}  // namespace

ProfScope::ProfScope(int cls, double work, hipStream_t stream) : stream_(stream), idx_(-1) {
  if (!g.on) return;
  Rec r;
  r.a = g.next(); r.b = g.next(); r.cls = cls; r.work = work;
  if (!r.a || !r.b) return;
  (void)hipEventRecord(r.a, stream);
  g.recs.push_back(r);
  idx_ = (int)g.recs.size() - 1;
}
ProfScope::~ProfScope() {
  if (idx_ >= 0) (void)hipEventRecord(g.recs[idx_].b, stream_);
}

int profile_begin() {
  g.on = true;
  g.used = 0;
  g.recs.clear();
  return EZ_OK;
}

int profile_end(int cls, double* ms, double* work, int* launches) {
  g.on = false;
  double t = 0, w = 0;
  int n = 0;
  for (auto& r : g.recs) {
    if (r.cls != cls) continue;
    EZ_HIP(hipEventSynchronize(r.b));
    float e = 0.f;
    EZ_HIP(hipEventElapsedTime(&e, r.a, r.b));
    t += e; w += r.work; ++n;
  }
  if (ms) *ms = t;
  if (work) *work = w;
  if (launches) *launches = n;
  return EZ_OK;
}

}  // namespace ezclip
