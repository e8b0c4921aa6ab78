// Library-level entry points: version and thread-local error text.
#include "be_common.h"

namespace be {
char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace be

extern "C" int be_version(void) { return 1; }
extern "C" const char* be_last_error(void) { return be::last_error_buf(); }

// ------------------------------------------------------------------------------------------------ profiling
namespace {
struct Rec { hipEvent_t a, b; int kernel_id; double flops, bytes, flops_exec; };
// n is claimed with an atomic increment: two host threads launching with the hooks enabled get distinct slots (enable / read are
// single-threaded by contract: they belong to the measuring harness)
struct Prof { Rec* recs = nullptr; int cap = 0; std::atomic<int> n{0}; bool on = false; } g_prof;
}  // namespace

namespace be {
ProfileScope::ProfileScope(hipStream_t s, int kernel_id, double flops, double bytes, double flops_executed)
    : s_(s), slot_(-1) {
    if (!g_prof.on) return;
    const int slot = g_prof.n.fetch_add(1, std::memory_order_relaxed);
    if (slot >= g_prof.cap) { g_prof.n.fetch_sub(1, std::memory_order_relaxed); return; }
    slot_ = slot;
    Rec& r = g_prof.recs[slot_];
    r.kernel_id = kernel_id; r.flops = flops; r.bytes = bytes; r.flops_exec = flops_executed;
    (void)hipEventRecord(r.a, s_);
}
ProfileScope::~ProfileScope() {
    if (slot_ >= 0) (void)hipEventRecord(g_prof.recs[slot_].b, s_);
}
}  // namespace be

extern "C" int be_profile_enable(int max_launches) {
    for (int i = 0; i < g_prof.cap; ++i) { (void)hipEventDestroy(g_prof.recs[i].a); (void)hipEventDestroy(g_prof.recs[i].b); }
    delete[] g_prof.recs;
    g_prof.recs = nullptr; g_prof.cap = 0; g_prof.n.store(0); g_prof.on = false;
    if (max_launches <= 0) return BE_OK;
    g_prof.recs = new Rec[max_launches];
    for (int i = 0; i < max_launches; ++i) {
        if (hipEventCreate(&g_prof.recs[i].a) != hipSuccess || hipEventCreate(&g_prof.recs[i].b) != hipSuccess)
            return be::fail(BE_ELAUNCH, "be_profile_enable: hipEventCreate failed");
    }
    g_prof.cap = max_launches;
    g_prof.on = true;
    return BE_OK;
}

extern "C" int be_profile_reset(void) { g_prof.n.store(0); return BE_OK; }

extern "C" int be_profile_read(int* kernel_id, double* flops, double* bytes, double* flops_executed, float* ms, int cap) {
    const int have = g_prof.n.load();
    const int n = have < cap ? have : cap;
    for (int i = 0; i < n; ++i) {
        Rec& r = g_prof.recs[i];
        if (hipEventSynchronize(r.b) != hipSuccess) return be::fail(BE_ELAUNCH, "be_profile_read: event sync failed");
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return be::fail(BE_ELAUNCH, "be_profile_read: elapsed failed");
        kernel_id[i] = r.kernel_id; flops[i] = r.flops; bytes[i] = r.bytes; ms[i] = t;
        if (flops_executed) flops_executed[i] = r.flops_exec;
    }
    return n;
}

namespace {
__global__ void k_vec_add(float* a, const float* b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += b[i];
}
}  // namespace
int be::vec_add_inplace(float* a, const float* b, int n, void* stream) {
    hipLaunchKernelGGL(k_vec_add, dim3((n + 255) / 256), dim3(256), 0, be::as_stream(stream), a, b, n);
    return be::check_launch("vec_add_inplace");
}
