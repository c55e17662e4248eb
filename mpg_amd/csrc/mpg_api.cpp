// ABI bookkeeping: version + thread-local last-error string.
#include <stdarg.h>
#include <stdio.h>

#include "mpg_common.h"

static thread_local char g_err[512] = "";

void mpg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int mpg_abi_version(void) { return MPG_ABI_VERSION; }
extern "C" const char* mpg_last_error(void) { return g_err; }

// ---- optional per-kernel timing (HIP events on the launch stream), off by default ------------------------
// bench.py uses it to report the live average duration of the dominant kernels over its timed region.
#include <vector>

namespace {
constexpr int NSLOT = 8, MAXPAIR = 16384;
struct ProfSlot {
    std::vector<hipEvent_t> start, stop;
    int used = 0;
    long calls = 0;        // launches seen since mpg_prof_enable
    bool open = false;     // the current launch is being timed
};
ProfSlot g_slot[NSLOT];
int g_prof_on = 0;
const char* g_slot_name[NSLOT] = {"k_rollout_fwd", "k_rollout_bwd", "k_step (env)", "k_forward", "k_backward",
                                  "k_wgrad", "k_target_fused", ""};
}  // namespace

void mpg_prof_begin(int slot, hipStream_t s) {
    if (!g_prof_on || slot < 0 || slot >= NSLOT) return;
    ProfSlot& p = g_slot[slot];
    // an event record is a packet of its own on the stream (~4-5 us between two otherwise back-to-back kernels): only
    // every g_prof_on-th launch of a slot is timed
    p.open = (p.calls++ % g_prof_on) == 0 && p.used < MAXPAIR;
    if (!p.open) return;
    if ((int)p.start.size() <= p.used) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { p.open = false; return; }
        p.start.push_back(a);
        p.stop.push_back(b);
    }
    (void)hipEventRecord(p.start[p.used], s);
}

void mpg_prof_end(int slot, hipStream_t s) {
    if (!g_prof_on || slot < 0 || slot >= NSLOT) return;
    ProfSlot& p = g_slot[slot];
    if (!p.open) return;
    (void)hipEventRecord(p.stop[p.used], s);
    ++p.used;
    p.open = false;
}

extern "C" int mpg_prof_enable(int every) {
    g_prof_on = every > 0 ? every : 0;
    for (int i = 0; i < NSLOT; ++i) { g_slot[i].used = 0; g_slot[i].calls = 0; g_slot[i].open = false; }
    return MPG_OK;
}

extern "C" int mpg_prof_read(int slot, double* total_ms, int* count) {
    MPG_REQUIRE(slot >= 0 && slot < NSLOT && total_ms && count, "mpg_prof_read: bad argument");
    ProfSlot& p = g_slot[slot];
    double tot = 0.0;
    for (int i = 0; i < p.used; ++i) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(p.stop[i]);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, p.start[i], p.stop[i]);
        if (e != hipSuccess) {
            mpg_set_error("mpg_prof_read: %s", hipGetErrorString(e));
            return -(int)e;
        }
        tot += ms;
    }
    *total_ms = tot;
    *count = p.used;
    return MPG_OK;
}

extern "C" const char* mpg_prof_slot_name(int slot) { return (slot >= 0 && slot < NSLOT) ? g_slot_name[slot] : ""; }
