// Optional in-library kernel timing: HIP events recorded on the launch stream around kernel launches (all, or every s-th).
// bench.py switches it on for the timed region to get the dominant kernel's live average launch duration
// (the roofline figure) without an external profiler; it is off by default and costs nothing then.
#include <mutex>
#include <vector>

#include "common.h"

namespace {
struct Rec {
  hipEvent_t a, b;
  int kind;
  double work;
};
std::mutex g_mu;
std::vector<Rec> g_recs;
size_t g_used = 0;
bool g_on = false;
int g_stride = 1;                            // time every g_stride-th launch of each kind
long long g_seen[DGVIT_PROFILE_KINDS] = {};  // all launches between start and stop, per kind
double g_work_all[DGVIT_PROFILE_KINDS] = {};
}  // namespace

// returns a slot index (>= 0) when this launch is being timed, -1 otherwise
int profile_begin(int kind, double work, hipStream_t st) {
  if (!g_on) return -1;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_on) return -1;
  const int kk = kind >= 0 && kind < DGVIT_PROFILE_KINDS ? kind : DGVIT_PROFILE_KINDS - 1;
  const long long seen = g_seen[kk]++;
  g_work_all[kk] += work;
  if (seen % g_stride != 0 || g_used >= g_recs.size()) return -1;
  Rec& r = g_recs[g_used];
  r.kind = kind;
  r.work = work;
  if (hipEventRecord(r.a, st) != hipSuccess) return -1;
  return (int)g_used++;
}

void profile_end(int slot, hipStream_t st) {
  if (slot < 0) return;
  (void)hipEventRecord(g_recs[slot].b, st);
}

extern "C" int dgvit_profile_start(int max_records) {
  std::lock_guard<std::mutex> lk(g_mu);
  DGVIT_CHECK_ARG(max_records > 0 && max_records <= (1 << 20), "profile_start: bad record count");
  while (g_recs.size() < (size_t)max_records) {
    Rec r;
    r.kind = 0;
    r.work = 0;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess)
      return dgvit_set_error(DGVIT_ERR_HIP, "profile_start: hipEventCreate failed");
    g_recs.push_back(r);
  }
  g_used = 0;
  for (int k = 0; k < DGVIT_PROFILE_KINDS; ++k) {
    g_seen[k] = 0;
    g_work_all[k] = 0;
  }
  g_on = true;
  return DGVIT_OK;
}

extern "C" int dgvit_profile_sampling(int stride) {
  std::lock_guard<std::mutex> lk(g_mu);
  DGVIT_CHECK_ARG(stride >= 1 && stride <= (1 << 20), "profile_sampling: stride must be >= 1");
  g_stride = stride;
  return DGVIT_OK;
}

extern "C" int dgvit_profile_totals(double* work_all, long long* launches_all) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (int k = 0; k < DGVIT_PROFILE_KINDS; ++k) {
    if (work_all) work_all[k] = g_work_all[k];
    if (launches_all) launches_all[k] = g_seen[k];
  }
  return DGVIT_OK;
}

// Sums per kind (DGVIT_PROFILE_KINDS entries each): milliseconds, work units (FLOPs for GEMM/attention,
// bytes for the rest), launches.  Blocks until the recorded events have completed.
extern "C" int dgvit_profile_stop(double* ms, double* work, long long* launches) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = false;
  for (int k = 0; k < DGVIT_PROFILE_KINDS; ++k) {
    if (ms) ms[k] = 0;
    if (work) work[k] = 0;
    if (launches) launches[k] = 0;
  }
  for (size_t i = 0; i < g_used; ++i) {
    Rec& r = g_recs[i];
    if (hipEventSynchronize(r.b) != hipSuccess) return dgvit_set_error(DGVIT_ERR_HIP, "profile_stop: event sync failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return dgvit_set_error(DGVIT_ERR_HIP, "profile_stop: elapsed failed");
    const int k = r.kind >= 0 && r.kind < DGVIT_PROFILE_KINDS ? r.kind : DGVIT_PROFILE_KINDS - 1;
    if (ms) ms[k] += t;
    if (work) work[k] += r.work;
    if (launches) launches[k] += 1;
  }
  g_used = 0;
  return DGVIT_OK;
}
