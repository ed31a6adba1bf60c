// Runtime around the kernels of liblmnet_hip.so: plan recorder (one FFI crossing per pass), stream ordering helpers,
// and the in-library kernel timer that bench.py's `roofline` block reads.
//
// The reference's hot loop (utils/train_eval_utils.py:140-145: model(images); loss.backward()) crosses into ~1200 ATen
// calls per step.  Here a pass is a fixed schedule of C-ABI entries over fixed buffers, so it is recorded once (entries
// executed AND remembered, arguments by value) and afterwards re-issued by lmn_plan_run -- on the same HIP streams, with
// the cross-stream dependencies re-created by events -- without any host code per kernel.
#include "common.h"

#include <cxxabi.h>

#include <chrono>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

int g_lmn_prof_on = 0;
int g_lmn_det = 0;
thread_local void* g_lmn_rec = nullptr;

namespace {

// ------------------------------------------------------------------------------------------------ plan
struct Plan {
  std::vector<std::function<int()>> ops;
  std::vector<const char*> names;  // the recorded call expressions (string literals of the entry points)
  std::vector<hipEvent_t> events;  // owned by the plan: one per recorded cross-stream wait
  bool sealed = false;
};

// events for lmn_stream_wait outside of a plan: a ring (an event may be re-recorded as soon as the wait on its previous
// record has been ISSUED -- hipStreamWaitEvent captures the record that is current at the call)
struct EventRing {
  std::vector<hipEvent_t> ev;
  size_t next = 0;
  hipEvent_t get() {
    if (ev.size() < 512) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
      ev.push_back(e);
      return e;
    }
    hipEvent_t e = ev[next];
    next = (next + 1) % ev.size();
    return e;
  }
};
thread_local EventRing g_ring;

int wait_with(hipEvent_t e, hipStream_t waiter, hipStream_t waited) {
  hipError_t r = hipEventRecord(e, waited);
  if (r == hipSuccess) r = hipStreamWaitEvent(waiter, e, 0);
  if (r != hipSuccess) {
    snprintf(g_lmn_err, sizeof(g_lmn_err), "stream_wait: %s", hipGetErrorString(r));
    return (int)r;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ kernel timer
struct ProfRec {
  const char* name;  // string literal of the launch site
  hipStream_t st;
  hipEvent_t e0, e1;
  double flops, bytes;
};
struct Prof {
  std::mutex mu;
  std::string filter;  // '|'-separated substrings; empty = every kernel
  std::vector<std::string> parts;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  bool detail = false;  // filter began with '@': one line per (kernel, declared cost) = per layer shape
  bool timeline = false;  // filter began with '!': one line per LAUNCH: name, stream, start and end (us after the first launch)
} g_prof;
thread_local double g_cost_flops = 0.0, g_cost_bytes = 0.0;
thread_local hipEvent_t g_open_e1 = nullptr;

hipEvent_t prof_event() {
  if (!g_prof.pool.empty()) {
    hipEvent_t e = g_prof.pool.back();
    g_prof.pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

}  // namespace

void lmn_rec_push(std::function<int()>&& f, const char* what) {
  Plan* p = (Plan*)g_lmn_rec;
  if (p && !p->sealed) {
    p->ops.push_back(std::move(f));
    p->names.push_back(what);
  }
}

// "N2..7LmnKTagIXadL_Z...EEE" -> "dw_fwd_kernel<float, true>".  The host demangler of this image predates the bf16 type code
// (DF16b): it is swapped for `t` (unsigned short, a builtin like it, so substitution indices are unchanged; no kernel takes a u16) and
// swapped back in the text.
const char* lmn_kname(const char* tag) {
  static std::mutex mu;
  static std::map<const char*, std::string> names;
  std::lock_guard<std::mutex> lk(mu);
  auto it = names.find(tag);
  if (it != names.end()) return it->second.c_str();
  std::string m(tag);
  for (size_t p; (p = m.find("DF16b")) != std::string::npos;) m.replace(p, 5, "t");
  int status = 0;
  char* d = abi::__cxa_demangle(m.c_str(), nullptr, nullptr, &status);
  std::string s = (status == 0 && d) ? d : tag;
  free(d);
  for (size_t p; (p = s.find("(anonymous namespace)::")) != std::string::npos;) s.erase(p, 23);
  for (size_t p; (p = s.find("unsigned short")) != std::string::npos;) s.replace(p, 14, "__bf16");
  size_t b = s.find("&(void ");           // LmnKTag<&(void NAME<ARGS>(PARAMS))>, or LmnKTag<&NAME> / LmnKTag<&NAME(PARAMS)> of a plain function
  if (b != std::string::npos) b += 7;
  else if ((b = s.find("<&")) != std::string::npos) { b += 2; if (s.back() == '>') s.pop_back(); }
  if (b != std::string::npos) {
    size_t e = b;
    for (int depth = 0; e < s.size(); ++e) {
      if (s[e] == '<') ++depth;
      else if (s[e] == '>') --depth;
      else if (s[e] == '(' && depth == 0) break;
    }
    s = s.substr(b, e - b);
  }
  return names.emplace(tag, std::move(s)).first->second.c_str();
}

void lmn_prof_cost(double flops, double bytes) {
  g_cost_flops = flops;
  g_cost_bytes = bytes;
}

bool lmn_prof_start(const char* kernel, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  const double fl = g_cost_flops, by = g_cost_bytes;
  g_cost_flops = g_cost_bytes = 0.0;
  if (!g_prof.parts.empty()) {
    bool hit = false;
    for (const std::string& s : g_prof.parts)
      if (strstr(kernel, s.c_str())) { hit = true; break; }
    if (!hit) return false;
  }
  ProfRec r;
  r.name = kernel;
  r.st = st;
  r.e0 = prof_event();
  r.e1 = prof_event();
  r.flops = fl;
  r.bytes = by;
  if (!r.e0 || !r.e1) return false;
  (void)hipEventRecord(r.e0, st);
  g_open_e1 = r.e1;
  g_prof.recs.push_back(r);
  return true;
}

void lmn_prof_stop(hipStream_t st) {
  if (g_open_e1) (void)hipEventRecord(g_open_e1, st);
  g_open_e1 = nullptr;
}

// ---- deterministic mode: per-stream slot scratch + fixed-order slot sum
namespace {
struct DetWs { char* base = nullptr; size_t bytes = 0, used = 0; std::vector<char*> retired; };   // retired: outgrown blocks (see lmn_det_slots)
std::map<hipStream_t, DetWs> g_det_ws;
std::mutex g_det_mu;

// block = 8 values x 32 slot groups: thread (v, g) sums slots g, g + 32, ... in ascending order, then the 32 group sums are added
// in ascending order by one thread per value
__global__ __launch_bounds__(256) void det_sum_kernel(const float* __restrict__ slots, int nslots, int64_t size, float* __restrict__ dst) {
  __shared__ float part[8][33];
  const int v = threadIdx.x >> 5, g = threadIdx.x & 31;
  const int64_t i = (int64_t)blockIdx.x * 8 + v;
  float a = 0.f;
  if (i < size)
    for (int s = g; s < nslots; s += 32) a += slots[(int64_t)s * size + i];
  part[v][g] = a;
  __syncthreads();
  if (g == 0 && i < size) {
    float t = 0.f;
    for (int k = 0; k < 32; ++k) t += part[v][k];
    dst[i] += t;
  }
}
}  // namespace

void lmn_det_begin(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_det_mu);
  DetWs& w = g_det_ws[st];
  w.used = 0;
  // blocks outgrown by an EARLIER entry are freed here, at the start of the next one and after a wait for the stream: no region of
  // the new entry lives in them, and whatever the earlier entry queued against them has run (never inside lmn_det_slots, where a
  // second growth of one entry would free the block that holds the entry's first region while its kernels are still to be launched)
  if (!w.retired.empty()) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return;   // (cannot wait inside a capture: keep them)
    (void)hipStreamSynchronize(st);
    for (char* r : w.retired) (void)hipFree(r);
    w.retired.clear();
  }
}

float* lmn_det_slots(hipStream_t st, size_t floats) {
  std::lock_guard<std::mutex> lk(g_det_mu);
  DetWs& w = g_det_ws[st];
  const size_t need = ((floats * sizeof(float) + 255) / 256) * 256;
  if (w.used + need > w.bytes) {
    // (a stream that is being captured into a hipGraph can neither be waited for nor allocate: deterministic mode then needs a
    //  scratch that already exists for that stream -- enable_graphs() captures on a stream of its own, so it is refused)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return nullptr;
    // grow: a larger block replaces the scratch.  The outgrown block is RETIRED, not freed: regions handed out earlier in this
    // entry live in it and stay valid, and work already queued on the stream may still read it.  Retired blocks are freed by the
    // next lmn_det_begin on this stream (growth happens a handful of times in a process: sizes repeat from step to step).
    if (w.base) w.retired.push_back(w.base);
    size_t nb = w.bytes ? w.bytes : (size_t)32 << 20;
    while (nb < need) nb *= 2;
    if (nb < 2 * w.bytes) nb = 2 * w.bytes;
    w.base = nullptr;
    if (hipMalloc((void**)&w.base, nb) != hipSuccess) { w.bytes = w.used = 0; return nullptr; }
    w.bytes = nb;
    w.used = 0;
  }
  float* p = (float*)(w.base + w.used);
  if (hipMemsetAsync(p, 0, need, st) != hipSuccess) return nullptr;   // (the region is not handed out: `used` stays)
  w.used += need;
  return p;
}

void lmn_det_sum(hipStream_t st, const float* slots, int nslots, int64_t size, float* dst) {
  if (size <= 0 || nslots <= 0) return;
  hipLaunchKernelGGL(det_sum_kernel, dim3((unsigned)((size + 7) / 8)), dim3(256), 0, st, slots, nslots, size, dst);
}

static struct { hipStream_t st; int level; } g_prio[8];   // (common.h (4); level 0 = free entry)
int lmn_prio_level(hipStream_t st) {
  for (int i = 0; i < 8; ++i)
    if (g_prio[i].level > 0 && g_prio[i].st == st) return g_prio[i].level;
  return 0;
}

extern "C" {

// ---- stream ordering without torch: `waiter` waits for everything enqueued on `waited` so far
int lmn_stream_wait(lmn_stream_t waiter, lmn_stream_t waited) {
  if (waiter == waited) return 0;
  Plan* p = (Plan*)g_lmn_rec;
  hipEvent_t e = nullptr;
  if (p && !p->sealed) {
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      snprintf(g_lmn_err, sizeof(g_lmn_err), "stream_wait: hipEventCreate failed");
      return LMN_E_BADARG;
    }
    p->events.push_back(e);
    p->ops.push_back([e, waiter, waited]() -> int { return wait_with(e, (hipStream_t)waiter, (hipStream_t)waited); });
  } else {
    e = g_ring.get();
    LMN_REQUIRE(e, "stream_wait: no event");
  }
  return wait_with(e, (hipStream_t)waiter, (hipStream_t)waited);
}

// ---- numbered events: record on one stream now, make another stream wait for exactly that point later
// (lmn_stream_wait orders against everything enqueued so far, which would also wait for work queued AFTER the point of
// interest on the waited stream).  64 slots per process; a slot may be re-recorded once its waits have been issued.
static hipEvent_t g_slots[64] = {nullptr};
static hipEvent_t slot_event(int slot) {
  if (slot < 0 || slot >= 64) return nullptr;
  if (!g_slots[slot] && hipEventCreateWithFlags(&g_slots[slot], hipEventDisableTiming) != hipSuccess) g_slots[slot] = nullptr;
  return g_slots[slot];
}
int lmn_event_record(int slot, lmn_stream_t stream) {
  LMN_REC(lmn_event_record(slot, stream));
  hipEvent_t e = slot_event(slot);
  LMN_REQUIRE(e, "event_record: slot %d", slot);
  const hipError_t r = hipEventRecord(e, (hipStream_t)stream);
  LMN_REQUIRE(r == hipSuccess, "event_record: %s", hipGetErrorString(r));
  return 0;
}
int lmn_event_wait(int slot, lmn_stream_t stream) {
  LMN_REC(lmn_event_wait(slot, stream));
  hipEvent_t e = slot_event(slot);
  LMN_REQUIRE(e, "event_wait: slot %d", slot);
  const hipError_t r = hipStreamWaitEvent((hipStream_t)stream, e, 0);
  LMN_REQUIRE(r == hipSuccess, "event_wait: %s", hipGetErrorString(r));
  return 0;
}

// ---- the compute chain's stream (see common.h (4))
int lmn_set_priority_stream(lmn_stream_t stream, int level) {
  LMN_REQUIRE(level >= 0 && level <= 3, "set_priority_stream: level %d", level);
  hipStream_t st = (hipStream_t)stream;
  int slot = -1;
  for (int i = 0; i < 8; ++i) {
    if (g_prio[i].level > 0 && g_prio[i].st == st) { slot = i; break; }
    if (g_prio[i].level == 0 && slot < 0) slot = i;
  }
  if (slot < 0) {   // table full (a caller that cycles through streams): the priority is a scheduling hint -- the oldest entry makes room
    static int victim = 0;
    slot = victim;
    victim = (victim + 1) & 7;
  }
  g_prio[slot].st = st;
  g_prio[slot].level = level;
  return 0;
}

// ---- deterministic mode switch (see common.h (3))
int lmn_set_deterministic(int on) {
  g_lmn_det = on ? 1 : 0;
  return 0;
}
int lmn_get_deterministic(void) { return g_lmn_det; }

// ---- plans
lmn_plan_t lmn_plan_create(void) { return (lmn_plan_t) new Plan(); }

int lmn_plan_destroy(lmn_plan_t plan) {
  Plan* p = (Plan*)plan;
  if (!p) return 0;
  if (g_lmn_rec == p) g_lmn_rec = nullptr;
  for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
  delete p;
  return 0;
}

int lmn_plan_record_begin(lmn_plan_t plan) {
  Plan* p = (Plan*)plan;
  LMN_REQUIRE(p && !p->sealed, "plan_record_begin: null or sealed plan");
  LMN_REQUIRE(g_lmn_rec == nullptr || g_lmn_rec == p, "plan_record_begin: another plan is recording on this thread");
  g_lmn_rec = p;
  return 0;
}

// pause / resume (entries issued while paused run but are not remembered); returns the number of ops recorded so far,
// which is what lmn_plan_run takes as segment bounds
int64_t lmn_plan_record_end(lmn_plan_t plan, int seal) {
  Plan* p = (Plan*)plan;
  if (!p) return -1;
  if (g_lmn_rec == p) g_lmn_rec = nullptr;
  if (seal) p->sealed = true;
  return (int64_t)p->ops.size();
}

int64_t lmn_plan_size(lmn_plan_t plan) { return plan ? (int64_t)((Plan*)plan)->ops.size() : -1; }

// re-issue ops [lo, hi) (hi < 0: to the end).  Stops at the first failing entry and returns its code.
int lmn_plan_run(lmn_plan_t plan, int64_t lo, int64_t hi) {
  Plan* p = (Plan*)plan;
  LMN_REQUIRE(p, "plan_run: null plan");
  LMN_REQUIRE(g_lmn_rec == nullptr, "plan_run: a plan is recording on this thread");
  const int64_t n = (int64_t)p->ops.size();
  if (hi < 0 || hi > n) hi = n;
  LMN_REQUIRE(lo >= 0 && lo <= hi, "plan_run: bad range [%lld, %lld) of %lld", (long long)lo, (long long)hi, (long long)n);
  for (int64_t i = lo; i < hi; ++i) {
    const int rc = p->ops[(size_t)i]();
    if (rc != 0) return rc;
  }
  return 0;
}

// Host-side cost of a replay: runs the plan once, timing every entry on the host clock, and writes one line per entry point
//   name \t ops \t total_us \n     (returns the bytes needed including the terminator, like lmn_prof_end)
int64_t lmn_plan_host_profile(lmn_plan_t plan, char* out, int64_t cap) {
  Plan* p = (Plan*)plan;
  if (!p || g_lmn_rec) return -1;
  struct Agg { int64_t n = 0; double us = 0; };
  std::map<std::string, Agg> agg;
  for (size_t i = 0; i < p->ops.size(); ++i) {
    const auto t0 = std::chrono::steady_clock::now();
    if (p->ops[i]() != 0) return -1;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    std::string nm = i < p->names.size() && p->names[i] ? p->names[i] : "?";
    nm = nm.substr(0, nm.find('('));
    Agg& a = agg[nm];
    a.n += 1;
    a.us += us;
  }
  std::string s;
  char line[256];
  for (auto& kv : agg) {
    snprintf(line, sizeof(line), "%s\t%lld\t%.1f\n", kv.first.c_str(), (long long)kv.second.n, kv.second.us);
    s += line;
  }
  if (out && cap > 0) {
    const size_t k = s.size() < (size_t)cap - 1 ? s.size() : (size_t)cap - 1;
    memcpy(out, s.data(), k);
    out[k] = 0;
  }
  return (int64_t)s.size() + 1;
}

// ---- kernel timer
int lmn_prof_begin(const char* filter) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  for (ProfRec& r : g_prof.recs) { g_prof.pool.push_back(r.e0); g_prof.pool.push_back(r.e1); }
  g_prof.recs.clear();
  g_prof.parts.clear();
  g_prof.filter = filter ? filter : "";
  g_prof.detail = !g_prof.filter.empty() && g_prof.filter[0] == '@';
  g_prof.timeline = !g_prof.filter.empty() && g_prof.filter[0] == '!';
  if (g_prof.detail || g_prof.timeline) g_prof.filter.erase(0, 1);
  size_t a = 0;
  while (a <= g_prof.filter.size() && !g_prof.filter.empty()) {
    size_t b = g_prof.filter.find('|', a);
    if (b == std::string::npos) b = g_prof.filter.size();
    if (b > a) g_prof.parts.push_back(g_prof.filter.substr(a, b - a));
    a = b + 1;
  }
  g_lmn_prof_on = 1;
  return 0;
}

// Stops the timer, waits for the device, and writes one line per kernel name:
//   name \t launches \t total_us \t flops \t bytes \n      (flops / bytes: sums of the algorithmic costs the entries declared)
// Returns the number of bytes needed (including the terminator); call again with a larger buffer if > cap.
int64_t lmn_prof_end(char* out, int64_t cap) {
  g_lmn_prof_on = 0;
  (void)hipDeviceSynchronize();
  std::lock_guard<std::mutex> lk(g_prof.mu);
  std::string s;
  char line[512];
  if (g_prof.timeline && !g_prof.recs.empty()) {
    // events of different streams share the device clock: everything relative to the first launch's start event
    const hipEvent_t base = g_prof.recs.front().e0;
    for (ProfRec& r : g_prof.recs) {
      float a = 0.f, b = 0.f;
      if (hipEventElapsedTime(&a, base, r.e0) != hipSuccess || hipEventElapsedTime(&b, base, r.e1) != hipSuccess) continue;
      snprintf(line, sizeof(line), "%s\t%llx\t%.3f\t%.3f\t0\n", r.name, (unsigned long long)(uintptr_t)r.st, a * 1e3, b * 1e3);
      s += line;
    }
    if (out && cap > 0) {
      const size_t k = s.size() < (size_t)cap - 1 ? s.size() : (size_t)cap - 1;
      memcpy(out, s.data(), k);
      out[k] = 0;
    }
    return (int64_t)s.size() + 1;
  }
  struct Agg { int64_t n = 0; double us = 0, fl = 0, by = 0; };
  std::map<std::string, Agg> agg;
  for (ProfRec& r : g_prof.recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    std::string nm = r.name;
    if (!nm.empty() && nm.front() == '(' && nm.back() == ')') nm = nm.substr(1, nm.size() - 2);
    if (g_prof.detail) {
      char tag[96];
      snprintf(tag, sizeof(tag), "#%.4e/%.4e", r.flops, r.bytes);
      nm += tag;
    }
    Agg& a = agg[nm];
    a.n += 1; a.us += ms * 1e3; a.fl += r.flops; a.by += r.bytes;
  }
  for (auto& kv : agg) {
    snprintf(line, sizeof(line), "%s\t%lld\t%.3f\t%.6e\t%.6e\n", kv.first.c_str(), (long long)kv.second.n, kv.second.us,
             kv.second.fl, kv.second.by);
    s += line;
  }
  if (out && cap > 0) {
    const size_t k = s.size() < (size_t)cap - 1 ? s.size() : (size_t)cap - 1;
    memcpy(out, s.data(), k);
    out[k] = 0;
  }
  return (int64_t)s.size() + 1;
}

}  // extern "C"
