// wn_prof.hip -- see wn_prof.h
#include "wn_prof.h"

#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/wavenet_hip.h"

#ifdef WN_EMU
// Host build of the kernels (tests/emu): no events, but the launch LOG is kept -- tests assert which launches a mode
// issues (count / flops / bytes per tag; "ms" is 0).  wn_prof_is_on() stays false: the emulator has one in-order stream.
namespace {
struct EmuAgg {
    long count = 0;
    double flops = 0, bytes = 0;
};
bool g_emu_on = false;
std::map<std::string, EmuAgg> g_emu;
std::vector<std::string> g_seq;   // tags in issue order, incl. the "bucket_event" marks of wn_backward
}  // namespace
void wn_prof_mark(const char* name) {
    if (g_emu_on) g_seq.push_back(name);
}
void wn_prof_scope_begin(const char* name, double flops, double bytes, wn_stream_t) {
    if (!g_emu_on) return;
    g_seq.push_back(name);
    EmuAgg& a = g_emu[name];
    a.count++;
    a.flops += flops;
    a.bytes += bytes;
}
void wn_prof_scope_end(wn_stream_t) {}
bool wn_prof_is_on() { return false; }
extern "C" int wn_prof_enable(int on) {
    g_emu_on = on != 0;
    if (g_emu_on) {
        g_emu.clear();
        g_seq.clear();
    }
    return 0;
}
extern "C" int wn_prof_report(char* buf, size_t n) {
    std::string s = "{";
    bool first = true;
    for (auto& kv : g_emu) {
        char tmp[512];
        snprintf(tmp, sizeof(tmp), "%s\"%s\": {\"count\": %ld, \"ms\": 0.0, \"flops\": %.6e, \"bytes\": %.6e}", first ? "" : ", ",
                 kv.first.c_str(), kv.second.count, kv.second.flops, kv.second.bytes);
        s += tmp;
        first = false;
    }
    s += "}";
    if (!buf || n == 0) return (int)s.size() + 1;
    if (s.size() + 1 > n) return -1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return 0;
}
#else
namespace {
struct Rec {
    const char* name;
    double flops, bytes;
    hipEvent_t e0, e1;
};
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<std::string> g_seq;   // tags in issue order, incl. the "bucket_event" marks of wn_backward
std::vector<hipEvent_t> g_pool;
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

void wn_prof_mark(const char* name) {
    if (g_on) g_seq.push_back(name);
}

void wn_prof_scope_begin(const char* name, double flops, double bytes, wn_stream_t st) {
    if (!g_on) return;
    g_seq.push_back(name);
    Rec r;
    r.name = name;
    r.flops = flops;
    r.bytes = bytes;
    r.e0 = get_event();
    r.e1 = get_event();
    (void)hipEventRecord(r.e0, st);
    g_recs.push_back(r);
}

void wn_prof_scope_end(wn_stream_t st) {
    if (!g_on || g_recs.empty()) return;
    (void)hipEventRecord(g_recs.back().e1, st);
}

bool wn_prof_is_on() { return g_on; }

extern "C" int wn_prof_enable(int on) {
    g_on = on != 0;
    if (g_on) {
        for (auto& r : g_recs) {
            g_pool.push_back(r.e0);
            g_pool.push_back(r.e1);
        }
        g_recs.clear();
        g_seq.clear();
    }
    return 0;
}

// JSON object {"name": {"count": n, "ms": total, "flops": total, "bytes": total}, ...}.  The caller
// must have synchronised the stream(s).
extern "C" int wn_prof_report(char* buf, size_t n) {
    struct Agg {
        long count = 0;
        double ms = 0, flops = 0, bytes = 0;
    };
    std::map<std::string, Agg> agg;
    for (auto& r : g_recs) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        Agg& a = agg[r.name];
        a.count++;
        a.ms += ms;
        a.flops += r.flops;
        a.bytes += r.bytes;
    }
    std::string s = "{";
    bool first = true;
    for (auto& kv : agg) {
        char tmp[512];
        snprintf(tmp, sizeof(tmp), "%s\"%s\": {\"count\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}", first ? "" : ", ",
                 kv.first.c_str(), kv.second.count, kv.second.ms, kv.second.flops, kv.second.bytes);
        s += tmp;
        first = false;
    }
    s += "}";
    if (!buf || n == 0) return (int)s.size() + 1;
    if (s.size() + 1 > n) return -1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return 0;
}
#endif

// The recorded tags in issue order, comma separated ("bucket_event" = wn_backward recorded a gradient-bucket event there).
extern "C" int wn_prof_sequence(char* buf, size_t n) {
    std::string s;
    for (size_t i = 0; i < g_seq.size(); ++i) {
        if (i) s += ",";
        s += g_seq[i];
    }
    if (!buf || n == 0) return (int)s.size() + 1;
    if (s.size() + 1 > n) return -1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return 0;
}
