// wn_prof.h -- opt-in per-launch timing with HIP events (diagnostics for bench.py's roofline block).
// Disabled by default: one global flag test per launch.  Events are recorded on the stream the
// kernel is launched on.
#pragma once
#include "wn_device.h"

void wn_prof_scope_begin(const char* name, double flops, double bytes, wn_stream_t st);
void wn_prof_scope_end(wn_stream_t st);
void wn_prof_mark(const char* name);   // an entry of the issue-order log that is not a launch (gradient-bucket events)
bool wn_prof_is_on();  // per-launch timing active: launchers keep everything on one stream

struct WnProfScope {
    wn_stream_t st;
    WnProfScope(const char* name, double flops, double bytes, wn_stream_t s) : st(s) { wn_prof_scope_begin(name, flops, bytes, s); }
    ~WnProfScope() { wn_prof_scope_end(st); }
};
#define WN_PROF(name, flops, bytes, st) WnProfScope _wn_prof_scope((name), (flops), (bytes), (st))
