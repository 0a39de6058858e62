#include "prof.h"
#include <map>
#include <vector>
#include <cstdio>

namespace {
struct Rec { hipEvent_t a, b; std::string name; double flops, bytes; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
int g_every = 0;        // 0 = disabled
long g_forward = 0;
bool g_now = false;

hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

bool prof_active() { return g_now; }
void prof_enable(int every) {
  g_every = every;
  g_forward = 0;
  g_now = false;
}
void prof_begin_forward() {
  if (g_every <= 0) { g_now = false; return; }
  g_now = (g_forward % g_every) == 0;
  ++g_forward;
}

ProfScope::ProfScope(const char* name, double flops, double bytes, hipStream_t stream) : s(stream) {
  if (!g_now) return;
  Rec r{get_event(), get_event(), name, flops, bytes};
  (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
  rec = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
  if (rec >= 0) (void)hipEventRecord(g_recs[rec].b, s);
}

std::string prof_collect_json() {
  (void)hipDeviceSynchronize();
  struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : g_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      Agg& a = agg[r.name];
      a.n++; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
    }
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  std::string out = "{";
  bool first = true;
  char buf[512];
  for (auto& kv : agg) {
    snprintf(buf, sizeof buf, "%s\"%s\": {\"launches\": %ld, \"total_ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
             first ? "" : ", ", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
    out += buf;
    first = false;
  }
  return out + "}";
}
