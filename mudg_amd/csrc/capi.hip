// capi.hip — ABI bookkeeping: version, per-thread error text, optional hipEvent profiler.
#include "common.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
thread_local char g_err[512] = "";

struct Slot { hipEvent_t a, b; int fam; double flops, bytes; };
std::mutex g_mu;
std::vector<Slot> g_slots;       // events are recycled across resets
size_t g_used = 0;
int g_mask = 0;
}  // namespace

void mudg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#ifdef MUDG_DEBUG_VARIANTS
int mudg_variant(const char* name, int dflt) {
    char key[64] = "MUDG_";
    strncat(key, name, sizeof(key) - 6);
    const char* e = getenv(key);
    return e ? atoi(e) : dflt;
}
#endif

int mudg_prof_begin(int fam, hipStream_t s) {
    if (!(g_mask & (1 << fam))) return -1;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_used == g_slots.size()) {
        Slot sl{};
        if (hipEventCreate(&sl.a) != hipSuccess || hipEventCreate(&sl.b) != hipSuccess) return -1;
        g_slots.push_back(sl);
    }
    Slot& sl = g_slots[g_used];
    sl.fam = fam; sl.flops = 0; sl.bytes = 0;
    (void)hipEventRecord(sl.a, s);
    return (int)g_used++;
}

void mudg_prof_end(int slot, hipStream_t s, double flops, double bytes) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Slot& sl = g_slots[slot];
    sl.flops = flops; sl.bytes = bytes;
    (void)hipEventRecord(sl.b, s);
}

extern "C" {

int mudg_version(void) { return 2; }
int mudg_operand_dtype(void) { return MUDG_OPERAND_CODE; }
const char* mudg_last_error(void) { return g_err; }

int mudg_prof_enable(int family_mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_mask = family_mask;
    if (!family_mask) g_used = 0;
    return MUDG_OK;
}

int mudg_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_used = 0;
    return MUDG_OK;
}

int mudg_prof_collect(int fam, double* total_ms, int64_t* launches, double* flops, double* bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    double ms = 0, fl = 0, by = 0;
    int64_t n = 0;
    for (size_t i = 0; i < g_used; ++i) {
        Slot& sl = g_slots[i];
        if (sl.fam != fam) continue;
        if (hipEventSynchronize(sl.b) != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "prof: event sync failed");
        float t = 0.f;
        if (hipEventElapsedTime(&t, sl.a, sl.b) != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "prof: elapsed failed");
        ms += t; fl += sl.flops; by += sl.bytes; ++n;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (flops) *flops = fl;
    if (bytes) *bytes = by;
    return MUDG_OK;
}

}  // extern "C"
