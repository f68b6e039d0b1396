/*
 * tsan_double_mpc_hip.cpp -- a TEST DOUBLE of the few C-ABI entry points the drop-in (controller_mpc_hip.cpp) calls,
 * for ONE purpose: running the drop-in's own thread-facing code -- the handle pool, the hull generation, the
 * cancellation poll -- under ThreadSanitizer on a machine without a GPU (GPU sanitizers do not exist on this pool, and
 * an instrumented process cannot host the HIP runtime).  Linked into tests/host/host_tests_tsan ONLY, in place of
 * libmpc_hip.so; it computes nothing of the model (its "result" is a checksum of the problem it was given) and is no
 * fallback of anything: the product library has none and fails without a device.
 *
 * What it checks by itself: a handle is never inside two calls at once (the pool's whole contract, include/mpc_hip.h
 * "a handle ... may be used by one thread at a time") -- it aborts the process if it is.
 */
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "mpc_hip.h"

struct mpc_hip_handle {
    std::atomic<int> busy{0};
    mpc_model hull;
    int solves = 0;                 /* plain on purpose: ThreadSanitizer sees a handle shared by two threads as a race here */
};

namespace {
struct Busy {
    mpc_hip_handle* h;
    explicit Busy(mpc_hip_handle* hh) : h(hh)
    {
        if (h->busy.fetch_add(1) != 0) { std::fprintf(stderr, "tsan double: a handle is in two calls at once\n"); std::abort(); }
    }
    ~Busy() { h->busy.fetch_sub(1); }
};
std::atomic<int> g_live(0), g_peak(0);
}

extern "C" {

int mpc_hip_abi_version(void) { return MPC_HIP_ABI_VERSION; }
const char* mpc_hip_last_error(void) { return ""; }
const char* mpc_hip_status_string(int status)
{
    switch (status) {
    case MPC_OK: return "ok";
    case MPC_ERR_EMPTY_PLAN: return "Cannot run MPC on an empty plan";
    case MPC_ERR_NO_ITERATION: return "MPC failed to complete a single iteration";
    case MPC_ERR_NON_FINITE: return "MPC selected non-finite controls";
    default: return "test double: other status";
    }
}
void mpc_hip_default_problem(mpc_problem* p)
{
    std::memset(p, 0, sizeof *p);
    p->w_distance = 1; p->w_heading = 20; p->w_speed = 5; p->rudder_granularity = 0.0625; p->throttle_granularity = 0.125;
    p->min_rudder = -1; p->max_rudder = 1; p->max_throttle = 1; p->tolerance = 1e-5;
}
int mpc_hip_substep_schedule(double t, double* out, int cap)
{
    int n = 0;
    while (t > 0) { const double d = t < 0.1 ? t : 0.1; t -= 0.1; if (out && n < cap) out[n] = d; n++; }
    return n;
}
void mpc_hip_default_model(mpc_model* m)
{
    const mpc_model d = {0.0, 3200.0, 1000.0, 0.389105058, 20.0, 3.14159265358979323846 / 6, 0.25, 2.0, 2000.0, 8948.4, 2.75};
    *m = d;
}
int mpc_hip_create(int, void*, mpc_hip_handle** out)
{
    *out = new mpc_hip_handle;
    mpc_hip_default_model(&(*out)->hull);
    const int live = ++g_live;
    int peak = g_peak.load();
    while (live > peak && !g_peak.compare_exchange_weak(peak, live)) {}
    return MPC_OK;
}
void mpc_hip_destroy(mpc_hip_handle* h) { if (h) { --g_live; delete h; } }
int mpc_hip_set_timing(mpc_hip_handle* h, uint32_t) { Busy b(h); return MPC_OK; }
int mpc_hip_set_model(mpc_hip_handle* h, const mpc_model* models, uint32_t n)
{
    Busy b(h);
    if (!models || n != 1) return MPC_ERR_INVALID;
    h->hull = models[0];
    return MPC_OK;
}

/* "solves" like the library does as far as the drop-in can tell: polls the clock before and after every level, stops at
 * the deadline (the cancellation poll returns end_time), reports the levels completed, and returns a trajectory that
 * is a pure function of the problem and the hull. */
int mpc_hip_solve_tree(mpc_hip_handle* h, const mpc_problem* p, int max_levels, int, mpc_time_fn now, void* ctx,
                       double end_time, uint64_t, mpc_tree_level*, mpc_result* r)
{
    Busy b(h);
    h->solves++;
    std::memset(r, 0, sizeof *r);
    int done = 0;
    for (int l = 0; l < max_levels; l++) {
        if (now && now(ctx) >= end_time) break;
        std::this_thread::sleep_for(std::chrono::microseconds(30 * (l + 1)));     /* a level takes a while: solves overlap */
        if (now && now(ctx) >= end_time) break;
        done = l + 1;
    }
    if (done == 0) { r->status = MPC_ERR_NO_ITERATION; return MPC_ERR_NO_ITERATION; }
    double sum = h->hull.mass * 1e-3 + p->current_x - p->current_y + p->w_heading;
    for (int l = 0; l < done; l++) sum += p->checkpoints[l].x * (l + 1) - p->checkpoints[l].y;
    r->levels = done;
    for (int l = 0; l <= done; l++) {
        mpc_level_state& s = r->trajectory[l];
        s.x = p->start_x + l * sum; s.y = p->start_y - l; s.heading = p->start_heading; s.speed = p->start_speed;
        s.rpm = 100 * l; s.course_made_good = 0.1 * l; s.speed_over_ground = 2 + l; s.rudder = 0.0625 * done; s.throttle = 1; s.score = sum * l;
    }
    r->first_rudder = r->trajectory[1].rudder; r->first_throttle = 1;
    r->status = MPC_OK;
    return MPC_OK;
}

/* (the drop-in calls the deepening entry point; to the double there is no level too wide for one search) */
int mpc_hip_solve_tree_deep(mpc_hip_handle* h, const mpc_problem* p, int max_levels, int precision, mpc_time_fn now, void* ctx,
                            double end_time, mpc_tree_level* levels, mpc_result* r)
{
    return mpc_hip_solve_tree(h, p, max_levels, precision, now, ctx, end_time, 0, levels, r);
}

/* how many handles the pool ever had checked out or parked at once: two overlapping solves need two, not more */
int tsan_double_peak_handles(void) { return g_peak.load(); }

}
