// Optional in-library kernel timing with HIP events on the launch stream (used by bench.py for the live roofline
// figure).  Off by default: PROF_SCOPE compiles to two predictable branches on a global flag.
#pragma once
#include <hip/hip_runtime.h>
#include <string>

struct ProfScope {
  int rec = -1;
  hipStream_t s;
  ProfScope(const char* name, double flops, double bytes, hipStream_t stream);
  ~ProfScope();
};
bool prof_active();
void prof_begin_forward();  // called once per UNet / decoder / MLP forward: applies the sampling period
void prof_enable(int every_n_forwards);
std::string prof_collect_json();  // synchronises the device, aggregates, resets

#define PROF_SCOPE(name, flops, bytes, stream) ProfScope _prof_scope((name), (double)(flops), (double)(bytes), (stream))
