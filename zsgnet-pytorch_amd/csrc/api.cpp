// api.cpp — version / error plumbing and the optional per-launch HIP-event profiler of libzsg.
#include <stdarg.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void zsg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int g_zsg_deterministic = 0;
extern "C" int zsg_set_deterministic(int32_t on) {
    g_zsg_deterministic = on ? 1 : 0;
    return 0;
}
extern "C" int zsg_version(void) { return ZSG_VERSION; }
extern "C" const char* zsg_last_error(void) { return g_err; }

// ---- cross-stream ordering without marker packets ----------------------------------------------------------------------
thread_local hipEvent_t zsg_tls_completion_event = nullptr;
thread_local int zsg_tls_completion_uses = 0;

extern "C" void* zsg_event_create(void) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        zsg_set_error("event_create: %s", hipGetErrorString(hipGetLastError()));
        return nullptr;
    }
    return (void*)e;
}
extern "C" int zsg_event_destroy(void* ev) {
    if (ev && hipEventDestroy((hipEvent_t)ev) != hipSuccess) ZSG_FAIL(-3, "event_destroy: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}
extern "C" int zsg_set_completion_event(void* ev) {
    const int used = zsg_tls_completion_uses;        // launches that carried the previously armed event
    zsg_tls_completion_event = (hipEvent_t)ev;
    zsg_tls_completion_uses = 0;
    return used;
}
extern "C" int zsg_event_record(void* ev, void* stream) {
    ZSG_REQUIRE(ev, "event_record: null event");
    hipError_t e = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
    if (e != hipSuccess) ZSG_FAIL(-3, "event_record: %s", hipGetErrorString(e));
    return 0;
}
extern "C" int zsg_stream_wait_event(void* stream, void* ev) {
    ZSG_REQUIRE(ev, "stream_wait_event: null event");
    hipError_t e = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0);
    if (e != hipSuccess) ZSG_FAIL(-3, "stream_wait_event: %s", hipGetErrorString(e));
    return 0;
}

// ---- profiler ----------------------------------------------------------------------------------------------------
int g_zsg_prof_on = 0;
struct ProfRec {
    const char* name;
    hipEvent_t a, b;
    double flops, bytes;
};
static std::vector<ProfRec> g_recs;
static std::mutex g_prof_mu;

ZsgProfScope::ZsgProfScope(const char* name, hipStream_t s, double flops, double bytes) : slot(-1), st(s) {
    if (!g_zsg_prof_on) return;
    ProfRec r;
    r.name = name;
    r.flops = flops;
    r.bytes = bytes;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    hipEventRecord(r.a, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    slot = (int)g_recs.size();
    g_recs.push_back(r);
}
ZsgProfScope::~ZsgProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEventRecord(g_recs[slot].b, st);
}

extern "C" int zsg_prof_enable(int32_t on) {
    g_zsg_prof_on = on ? 1 : 0;
    return 0;
}

extern "C" int zsg_prof_collect(zsg_prof_entry* out, int32_t max_entries) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::map<std::string, zsg_prof_entry> agg;
    std::vector<std::string> order;
    for (auto& r : g_recs) {
        float ms = 0.f;
        hipEventSynchronize(r.b);
        hipEventElapsedTime(&ms, r.a, r.b);
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
        auto it = agg.find(r.name);
        if (it == agg.end()) {
            zsg_prof_entry e;
            memset(&e, 0, sizeof(e));
            strncpy(e.name, r.name, sizeof(e.name) - 1);
            it = agg.insert({r.name, e}).first;
            order.push_back(r.name);
        }
        it->second.launches += 1;
        it->second.ms += ms;
        it->second.flops += r.flops;
        it->second.bytes += r.bytes;
    }
    g_recs.clear();
    int n = 0;
    for (auto& k : order) {
        if (n >= max_entries) break;
        out[n++] = agg[k];
    }
    return n;
}
