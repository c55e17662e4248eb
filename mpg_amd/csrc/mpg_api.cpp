// ABI bookkeeping: version, thread-local last-error string, and the caller-owned kernel timer (mpg_prof_t).
#include <stdarg.h>
#include <stdio.h>

#include <new>
#include <vector>

#include "mpg_common.h"

// the only library-side state: the text behind mpg_last_error(), one buffer per calling thread
static thread_local char t_err[512] = "";

void mpg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
}

extern "C" int mpg_abi_version(void) { return MPG_ABI_VERSION; }
extern "C" const char* mpg_last_error(void) { return t_err; }

// ---- optional per-kernel timing (HIP events on the launch stream) -------------------------------------------
// A timer is an object the caller creates and hands in through mpg_cfg_t.prof; all its events exist before the first
// launch is timed.
struct mpg_prof {
    struct Slot {
        std::vector<hipEvent_t> start, stop;
        int used = 0;
        long calls = 0;        // launches seen since mpg_prof_start
        bool open = false;     // the current launch is being timed
    };
    Slot slot[MPG_PROF_SLOTS];
    int every = 0;
    int max_samples = 0;
};

void mpg_prof_begin(mpg_prof_t* p, int slot, hipStream_t s) {
    if (!p || !p->every || slot < 0 || slot >= MPG_PROF_SLOTS) return;
    mpg_prof::Slot& q = p->slot[slot];
    // an event record is a packet of its own on the stream (~4-5 us between two otherwise back-to-back kernels): only
    // every `every`-th launch of a slot is timed
    q.open = (q.calls++ % p->every) == 0 && q.used < p->max_samples;
    if (q.open) (void)hipEventRecord(q.start[q.used], s);
}

void mpg_prof_end(mpg_prof_t* p, int slot, hipStream_t s) {
    if (!p || !p->every || slot < 0 || slot >= MPG_PROF_SLOTS) return;
    mpg_prof::Slot& q = p->slot[slot];
    if (!q.open) return;
    (void)hipEventRecord(q.stop[q.used], s);
    ++q.used;
    q.open = false;
}

extern "C" int mpg_prof_create(int max_samples, mpg_prof_t** out) {
    MPG_REQUIRE(out && max_samples > 0 && max_samples <= 65536, "mpg_prof_create: bad argument");
    mpg_prof* p = new (std::nothrow) mpg_prof;
    MPG_REQUIRE(p, "mpg_prof_create: out of memory");
    p->max_samples = max_samples;
    for (int i = 0; i < MPG_PROF_SLOTS; ++i) {
        for (int k = 0; k < max_samples; ++k) {
            hipEvent_t a = nullptr, b = nullptr;
            hipError_t e = hipEventCreate(&a);
            if (e == hipSuccess) e = hipEventCreate(&b);
            if (e != hipSuccess) {
                if (a) (void)hipEventDestroy(a);
                mpg_set_error("mpg_prof_create: %s", hipGetErrorString(e));
                mpg_prof_destroy(p);
                return -(int)e;
            }
            p->slot[i].start.push_back(a);
            p->slot[i].stop.push_back(b);
        }
    }
    *out = p;
    return MPG_OK;
}

extern "C" int mpg_prof_destroy(mpg_prof_t* p) {
    if (!p) return MPG_OK;
    for (int i = 0; i < MPG_PROF_SLOTS; ++i) {
        for (hipEvent_t e : p->slot[i].start) (void)hipEventDestroy(e);
        for (hipEvent_t e : p->slot[i].stop) (void)hipEventDestroy(e);
    }
    delete p;
    return MPG_OK;
}

extern "C" int mpg_prof_start(mpg_prof_t* p, int every) {
    MPG_REQUIRE(p, "mpg_prof_start: null timer");
    p->every = every > 0 ? every : 0;
    for (int i = 0; i < MPG_PROF_SLOTS; ++i) { p->slot[i].used = 0; p->slot[i].calls = 0; p->slot[i].open = false; }
    return MPG_OK;
}

extern "C" int mpg_prof_read(mpg_prof_t* p, int slot, double* total_ms, int* count) {
    MPG_REQUIRE(p && slot >= 0 && slot < MPG_PROF_SLOTS && total_ms && count, "mpg_prof_read: bad argument");
    mpg_prof::Slot& q = p->slot[slot];
    double tot = 0.0;
    for (int i = 0; i < q.used; ++i) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(q.stop[i]);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, q.start[i], q.stop[i]);
        if (e != hipSuccess) {
            mpg_set_error("mpg_prof_read: %s", hipGetErrorString(e));
            return -(int)e;
        }
        tot += ms;
    }
    *total_ms = tot;
    *count = q.used;
    return MPG_OK;
}

extern "C" const char* mpg_prof_slot_name(int slot) {
    static const char* const names[MPG_PROF_SLOTS] = {"k_rollout_fwd", "k_rollout_bwd", "env step", "k_forward", "k_backward",
                                                     "k_wgrad", "k_target_fused", "k_critic_fused", "gradient exchange",
                                                     "k_clip_adam_polyak"};
    return (slot >= 0 && slot < MPG_PROF_SLOTS) ? names[slot] : "";
}

extern "C" int mpg_prof_region_begin(mpg_prof_t* p, int slot, mpg_stream_t stream) {
    mpg_prof_begin(p, slot, mpg_stream(stream));
    return MPG_OK;
}

extern "C" int mpg_prof_region_end(mpg_prof_t* p, int slot, mpg_stream_t stream) {
    mpg_prof_end(p, slot, mpg_stream(stream));
    return MPG_OK;
}
