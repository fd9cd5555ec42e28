// Error reporting, version query and optional per-family launch timing for the C ABI.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";
int mi355_g_tune[16];

int mi355_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int mi355_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mi355_fail(MI355_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return MI355_OK;
}

namespace {
struct Rec { int fam; hipEvent_t a, b; double work; };
std::mutex g_mu;
bool g_prof = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_free;
thread_local hipEvent_t g_open_a = nullptr;
thread_local double g_open_work = 0.0;

hipEvent_t get_event() {
    if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

void mi355_prof_begin(int, hipStream_t s, double work) {
    if (!g_prof) return;
    std::lock_guard<std::mutex> lk(g_mu);
    g_open_a = get_event();
    g_open_work = work;
    if (g_open_a) (void)hipEventRecord(g_open_a, s);
}

void mi355_prof_end(int fam, hipStream_t s) {
    if (!g_prof || !g_open_a) return;
    std::lock_guard<std::mutex> lk(g_mu);
    hipEvent_t b = get_event();
    if (!b) { g_free.push_back(g_open_a); g_open_a = nullptr; return; }
    (void)hipEventRecord(b, s);
    g_recs.push_back(Rec{fam, g_open_a, b, g_open_work});
    g_open_a = nullptr;
}

extern "C" {

int mi355_zero_f32(float* p, long long n, void* stream);     // elementwise.hip

const char* mi355_last_error(void) { return g_err; }

int mi355_abi_version(void) { return 1; }

// Timing of the launches issued between enable(1) and collect(): HIP events are recorded on the
// launch stream around every library call of a kernel family.  Not for use under graph capture.
int mi355_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_prof = on != 0;
    return MI355_OK;
}

// ms[f], work[f], count[f] for f in [0, 5): gemm/conv, wgrad, attention, norm, elementwise.
int mi355_prof_collect(double* ms, double* work, long long* count) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (int f = 0; f < FAM_COUNT; ++f) { ms[f] = 0.0; work[f] = 0.0; count[f] = 0; }
    for (Rec& r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) return mi355_fail(MI355_ERR_LAUNCH, "prof: event sync failed");
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms[r.fam] += t; work[r.fam] += r.work; count[r.fam] += 1; }
        g_free.push_back(r.a);
        g_free.push_back(r.b);
    }
    g_recs.clear();
    return MI355_OK;
}

double mi355_prof_event_overhead_us(void* stream, float* scratch, long long scratch_floats, int n) {
    if (!scratch || scratch_floats <= 0 || n <= 0 || n > 4096) { mi355_fail(MI355_ERR_ARG, "prof: bad calibration arguments"); return -1.0; }
    hipStream_t s = (hipStream_t)stream;
    std::vector<hipEvent_t> ev(2 * (size_t)n + 3, nullptr);
    bool ok = true;
    for (hipEvent_t& e : ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    hipEvent_t* outer = ev.data() + 2 * (size_t)n;
    // n launches bare, then the same n with an event pair around each: the second run is longer by two event records
    // per launch, of which the closing one falls inside its pair
    ok = ok && hipEventRecord(outer[0], s) == hipSuccess;
    for (int i = 0; ok && i < n; ++i) ok = mi355_zero_f32(scratch, scratch_floats, s) == MI355_OK;
    ok = ok && hipEventRecord(outer[1], s) == hipSuccess;
    for (int i = 0; ok && i < n; ++i) {
        ok = hipEventRecord(ev[2 * i], s) == hipSuccess && mi355_zero_f32(scratch, scratch_floats, s) == MI355_OK &&
             hipEventRecord(ev[2 * i + 1], s) == hipSuccess;
    }
    ok = ok && hipEventRecord(outer[2], s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    float bare = 0.f, paired = 0.f;
    ok = ok && hipEventElapsedTime(&bare, outer[0], outer[1]) == hipSuccess &&
         hipEventElapsedTime(&paired, outer[1], outer[2]) == hipSuccess;
    for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    if (!ok) { mi355_fail(MI355_ERR_LAUNCH, "prof: event calibration failed"); return -1.0; }
    const double us = ((double)paired - (double)bare) / (2.0 * n) * 1e3;
    return us > 0.0 ? us : 0.0;
}

}  // extern "C"
