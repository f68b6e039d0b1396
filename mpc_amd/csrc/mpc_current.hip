/*
 * mpc_current.hip -- water-current estimates for a batch of vessels on the device.
 *
 * Replaces, for batches of problems, the host call CurrentEstimator::getCurrent (CurrentEstimator.cpp:15-41) that
 * mpc_node.cpp:190-192 makes before every solve: dead-reckon from the oldest history sample through the recorded
 * controls WITHOUT current (<= 600 sequential VehicleState::simulate calls) up to the query time and attribute the
 * position error to a constant current over the whole span.  One lane per vessel; the strict FP64 model
 * (mpc_model.h, the reference's statements in its order), simulate()'s run-time substep schedule
 * (VehicleState.h:48-54: while (t > 0) { d = t < 0.1 ? t : 0.1; t -= 0.1; estimate(d) }) and estimate()'s
 * `time += dt` (:146) restated as they are, because the replay's branches compare those accumulated times.
 * Kept quirks: the walk starts AT the oldest sample (`begin()++` advances a temporary, :22), so its first simulate
 * call covers zero seconds; the replay starts from rpm(speed) of the oldest sample (VehicleState.h:20-22).
 *
 * With apply != 0 the estimates become the resident problems' currents, as controller.cpp:398-400 stores them for
 * the next solve: cx, cy and every field of the problem record derived from them (mpc_pack_problem) are rewritten
 * on the device, so a fleet goes plans -> checkpoints -> currents -> solve without a per-vessel host loop.
 */
#include <string.h>

#include "mpc_internal.h"

using namespace mpc;

struct Reckoned { Veh s; double time; };

/* VehicleState::simulate(rudder, throttle, t) with no current (VehicleState.h:43-55) */
__device__ static inline void reckon(Reckoned &e, double rudder, double throttle, double t, const Model &M)
{
    while (t > 0) {
        const double d = t < 0.1 ? t : 0.1;
        t -= 0.1;
        strict_level<0>(e.s, rudder, throttle, &d, 1, 0.0, 0.0, M);           /* a bare dt, not a schedule row */
        e.time += d;
    }
}

/* where a vessel's history lives: the caller's array-of-structs upload (mpc_hip_estimate_currents) ... */
struct UploadedHistory {
    const mpc_history_sample *hist;
    __device__ inline mpc_history_sample get(int i) const { return hist[i]; }
};
/* ... or the device-resident ring (mpc_hip_history_update): field-major, slot-major SoA ring[field][slot][vessel], so
 * the 64 lanes of a wave -- 64 vessels at the same age of history -- read 512 contiguous bytes per field */
struct ResidentHistory {
    const double *ring; uint32_t n, p, head;
    __device__ inline mpc_history_sample get(int i) const
    {
        const size_t at = (size_t)((head + (uint32_t)i) % MPC_CURRENT_MAX_SAMPLES) * n + p;
        const size_t f = (size_t)MPC_CURRENT_MAX_SAMPLES * n;
        return mpc_history_sample{ring[at], ring[f + at], ring[2 * f + at], ring[3 * f + at], ring[4 * f + at], ring[5 * f + at],
                                  ring[6 * f + at]};
    }
};

/* CurrentEstimator::getCurrent (CurrentEstimator.cpp:15-41) for one vessel */
template <class H>
__device__ inline void estimate_one(const H &hist, int cnt, double ax, double ay, double at, const Model &M, double &cx, double &cy)
{
    cx = 0.0; cy = 0.0;
    if (cnt < MPC_CURRENT_MIN_SAMPLES) return;                                              /* :18 */
    const mpc_history_sample h0 = hist.get(0);
    Reckoned e{Veh{h0.x, h0.y, h0.heading, h0.speed, rpm_from_speed(h0.speed, M), 0.0, 0.0}, h0.time};   /* :19 */
    const double span = at - e.time;
    if (!(span > 0)) return;                                                                /* :21 */
    mpc_history_sample s = h0;
    for (int i = 0; i < cnt; i++) {                                                         /* :22 */
        s = hist.get(i);
        if (s.time >= at) { reckon(e, s.rudder, s.throttle, at - e.time, M); break; }       /* :23-27 */
        reckon(e, s.rudder, s.throttle, s.time - e.time, M);                                /* :30 */
    }
    if (e.time < at) {                                                                      /* :34-37 */
        const mpc_history_sample last = hist.get(cnt - 1);
        reckon(e, last.rudder, last.throttle, at - e.time, M);
    }
    cx = (ax - e.s.x) / span;                                                               /* :39-40 */
    cy = (ay - e.s.y) / span;
}

/* the fields mpc_pack_problem derives from the current, same expressions */
__device__ inline void apply_current(DevProblem &P, double cx, double cy)
{
    P.cx = cx; P.cy = cy;
    double level_time = 0;
    for (int s = 0; s < P.nsub; s++) level_time += P.sub[s * SUB_STRIDE];
    for (int l = 0; l < P.depth; l++) {
        double *c = P.cp + l * CP_STRIDE;
        c[6] = c[0] - cx * (level_time * (l + 1)); c[7] = c[1] - cy * (level_time * (l + 1));
    }
    for (int s = 0; s < P.nsub; s++) {
        double *q = P.sub + s * SUB_STRIDE;
        q[3] = cx * q[0]; q[4] = cy * q[0];
    }
}

__global__ __launch_bounds__(64) void estimate_currents_kernel(const mpc_history_sample *__restrict__ samples,
                                                               const uint32_t *__restrict__ counts, uint32_t stride,
                                                               const double *__restrict__ actual, uint32_t n,
                                                               double *__restrict__ currents, DevProblem *__restrict__ probs,
                                                               const Model *__restrict__ models, uint32_t n_models)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const Model M = models[p % n_models];            /* vessel p's hull (mpc_hip_set_model) */
    double cx, cy;
    estimate_one(UploadedHistory{samples + (size_t)p * stride}, (int)counts[p], actual[3 * p], actual[3 * p + 1], actual[3 * p + 2], M, cx, cy);
    currents[2 * p] = cx; currents[2 * p + 1] = cy;
    if (probs) apply_current(probs[p], cx, cy);
}

/* ---- device-resident histories ------------------------------------------------------------------------------------
 * CurrentEstimator's own state for a fleet, kept in HBM: per vessel a ring of <= 600 samples, the count, the oldest
 * slot and the skips left (CurrentEstimator.h:43-47, :75, :82).  One tick = one mpc_hip_history_update: n samples up
 * (56 B each), one kernel that does updateEstimate (CurrentEstimator.cpp:5-13) for every vessel. */
struct HistoryState { uint32_t head, count; int32_t skips; uint32_t pad; };

__global__ __launch_bounds__(256) void history_reset_kernel(HistoryState *__restrict__ st, uint32_t n)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) st[p] = HistoryState{0u, 0u, 20, 0u};                                       /* resetEstimate, CurrentEstimator.h:43-47 */
}

__global__ __launch_bounds__(256) void history_update_kernel(double *__restrict__ ring, HistoryState *__restrict__ st,
                                                             const mpc_history_sample *__restrict__ in, uint32_t n)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    HistoryState s = st[p];
    if (s.skips > 0) { s.skips--; st[p] = s; return; }                                      /* :7-10 */
    uint32_t slot;
    if (s.count == MPC_CURRENT_MAX_SAMPLES) { slot = s.head; s.head = (s.head + 1) % MPC_CURRENT_MAX_SAMPLES; }   /* :12 pop_front */
    else { slot = (s.head + s.count) % MPC_CURRENT_MAX_SAMPLES; s.count++; }                 /* :11 emplace_back */
    const mpc_history_sample v = in[p];
    const size_t at = (size_t)slot * n + p, f = (size_t)MPC_CURRENT_MAX_SAMPLES * n;
    ring[at] = v.x; ring[f + at] = v.y; ring[2 * f + at] = v.heading; ring[3 * f + at] = v.speed; ring[4 * f + at] = v.time;
    ring[5 * f + at] = v.rudder; ring[6 * f + at] = v.throttle;
    st[p] = s;
}

__global__ __launch_bounds__(64) void estimate_currents_resident_kernel(const double *__restrict__ ring,
                                                                        const HistoryState *__restrict__ st,
                                                                        const double *__restrict__ actual, uint32_t n,
                                                                        double *__restrict__ currents, DevProblem *__restrict__ probs,
                                                                        const Model *__restrict__ models, uint32_t n_models)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const Model M = models[p % n_models];
    const HistoryState s = st[p];
    double cx, cy;
    estimate_one(ResidentHistory{ring, n, p, s.head}, (int)s.count, actual[3 * p], actual[3 * p + 1], actual[3 * p + 2], M, cx, cy);
    currents[2 * p] = cx; currents[2 * p + 1] = cy;
    if (probs) apply_current(probs[p], cx, cy);
}

extern "C" int mpc_hip_estimate_currents(mpc_hip_handle *h, const mpc_history_sample *h_samples, const uint32_t *h_counts,
                                         uint32_t stride, const double *h_actual, uint32_t n, int apply,
                                         double *h_currents_out)
{
    if (!h || !h_samples || !h_counts || !h_actual || n == 0 || stride == 0) {
        mpc_set_err("estimate_currents: NULL argument, zero vessels or zero stride"); return MPC_ERR_INVALID;
    }
    if (stride > MPC_CURRENT_MAX_SAMPLES) { mpc_set_err("estimate_currents: stride beyond MPC_CURRENT_MAX_SAMPLES"); return MPC_ERR_TOO_LARGE; }
    if (apply && n != h->n_probs) { mpc_set_err("estimate_currents: apply needs one history per resident problem"); return MPC_ERR_INVALID; }
    if (!apply && !h_currents_out) { mpc_set_err("estimate_currents: nothing to do (no output, no apply)"); return MPC_ERR_INVALID; }
    for (uint32_t p = 0; p < n; p++) {
        if (h_counts[p] > stride) { mpc_set_err("estimate_currents: a history is longer than the stride"); return MPC_ERR_INVALID; }
        /* the replay's loops end because simulated time advances: every quantity that drives them must be finite */
        if (!std::isfinite(h_actual[3 * p + 2])) { mpc_set_err("estimate_currents: query time is not finite"); return MPC_ERR_INVALID; }
        const mpc_history_sample *hist = h_samples + (size_t)p * stride;
        for (uint32_t i = 0; i < h_counts[p]; i++)
            if (!std::isfinite(hist[i].time)) { mpc_set_err("estimate_currents: a sample time is not finite"); return MPC_ERR_INVALID; }
        if (h_counts[p] && h_actual[3 * p + 2] - hist[0].time > MPC_CURRENT_MAX_SPAN) {
            mpc_set_err("estimate_currents: query more than MPC_CURRENT_MAX_SPAN seconds after the oldest sample"); return MPC_ERR_TOO_LARGE;
        }
    }
    if (int s = mpc_use_device(h)) return s;
    if (int s = mpc_flush_pending(h)) return s;          /* a pending finish reads the problem records this may rewrite */
    const size_t b_samples = sizeof(mpc_history_sample) * (size_t)n * stride;
    const size_t b_counts = (sizeof(uint32_t) * n + 15) & ~(size_t)15;
    const size_t b_actual = sizeof(double) * 3 * n, b_out = sizeof(double) * 2 * n;
    const size_t need = b_samples + b_counts + b_actual + b_out;
    if (need > h->cap_hist) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_hist); h->d_hist = nullptr; h->cap_hist = 0;
        HIP_TRY(hipMalloc(&h->d_hist, need));
        h->cap_hist = need;
    }
    char *base = (char *)h->d_hist;
    mpc_history_sample *d_samples = (mpc_history_sample *)base;
    uint32_t *d_counts = (uint32_t *)(base + b_samples);
    double *d_actual = (double *)(base + b_samples + b_counts);
    double *d_out = d_actual + 3 * (size_t)n;
    /* pageable host memory: the runtime stages each copy before hipMemcpyAsync returns */
    HIP_TRY(hipMemcpyAsync(d_samples, h_samples, b_samples, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d_counts, h_counts, sizeof(uint32_t) * n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d_actual, h_actual, b_actual, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(estimate_currents_kernel, dim3((n + 63) / 64), dim3(64), 0, h->stream, d_samples, d_counts, stride,
                       d_actual, n, d_out, apply ? h->d_probs : (DevProblem *)nullptr, h->d_models, (uint32_t)h->models.size());
    HIP_TRY(hipGetLastError());
    if (h_currents_out) HIP_TRY(hipMemcpyAsync(h_currents_out, d_out, b_out, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));      /* the host arrays are the caller's again; the estimates are in place */
    if (apply) h->ran = false;                     /* results of an earlier run belong to the old currents */
    return MPC_OK;
}

extern "C" int mpc_hip_history_reset(mpc_hip_handle *h, uint32_t n_vessels)
{
    if (!h || n_vessels == 0) { mpc_set_err("history_reset: NULL handle or zero vessels"); return MPC_ERR_INVALID; }
    if (int s = mpc_use_device(h)) return s;
    if (n_vessels != h->hist_n) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_hist_ring); (void)hipFree(h->d_hist_state); (void)hipFree(h->d_hist_io); (void)hipHostFree(h->h_hist_io);
        h->d_hist_ring = nullptr; h->d_hist_state = nullptr; h->d_hist_io = nullptr; h->h_hist_io = nullptr; h->hist_n = 0;
        HIP_TRY(hipMalloc(&h->d_hist_ring, sizeof(double) * 7 * (size_t)MPC_CURRENT_MAX_SAMPLES * n_vessels));
        HIP_TRY(hipMalloc(&h->d_hist_state, sizeof(HistoryState) * n_vessels));
        HIP_TRY(hipMalloc(&h->d_hist_io, sizeof(mpc_history_sample) * n_vessels));          /* >= 5 doubles per vessel: also actual[3] + out[2] */
        HIP_TRY(hipHostMalloc(&h->h_hist_io, sizeof(mpc_history_sample) * n_vessels));
        h->hist_n = n_vessels;
    }
    hipLaunchKernelGGL(history_reset_kernel, dim3((n_vessels + 255) / 256), dim3(256), 0, h->stream, (HistoryState *)h->d_hist_state, n_vessels);
    HIP_TRY(hipGetLastError());
    return MPC_OK;
}

extern "C" int mpc_hip_history_update(mpc_hip_handle *h, const mpc_history_sample *h_samples, uint32_t n)
{
    if (!h || !h_samples) { mpc_set_err("history_update: NULL argument"); return MPC_ERR_INVALID; }
    if (n == 0 || n != h->hist_n) { mpc_set_err("history_update: n differs from mpc_hip_history_reset's vessels"); return MPC_ERR_INVALID; }
    for (uint32_t p = 0; p < n; p++)
        if (!std::isfinite(h_samples[p].time)) { mpc_set_err("history_update: a sample time is not finite"); return MPC_ERR_INVALID; }
    if (int s = mpc_use_device(h)) return s;
    HIP_TRY(hipStreamSynchronize(h->stream));                   /* the pinned staging buffer may still be in flight */
    memcpy(h->h_hist_io, h_samples, sizeof(mpc_history_sample) * n);
    HIP_TRY(hipMemcpyAsync(h->d_hist_io, h->h_hist_io, sizeof(mpc_history_sample) * n, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(history_update_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, (double *)h->d_hist_ring,
                       (HistoryState *)h->d_hist_state, (const mpc_history_sample *)h->d_hist_io, n);
    HIP_TRY(hipGetLastError());
    return MPC_OK;
}

extern "C" int mpc_hip_estimate_currents_resident(mpc_hip_handle *h, const double *h_actual, uint32_t n, int apply,
                                                  double *h_currents_out)
{
    if (!h || !h_actual) { mpc_set_err("estimate_currents_resident: NULL argument"); return MPC_ERR_INVALID; }
    if (n == 0 || n != h->hist_n) { mpc_set_err("estimate_currents_resident: n differs from mpc_hip_history_reset's vessels"); return MPC_ERR_INVALID; }
    if (apply && n != h->n_probs) { mpc_set_err("estimate_currents_resident: apply needs one history per resident problem"); return MPC_ERR_INVALID; }
    if (!apply && !h_currents_out) { mpc_set_err("estimate_currents_resident: nothing to do (no output, no apply)"); return MPC_ERR_INVALID; }
    for (uint32_t p = 0; p < n; p++)
        if (!std::isfinite(h_actual[3 * p + 2])) { mpc_set_err("estimate_currents_resident: query time is not finite"); return MPC_ERR_INVALID; }
    if (int s = mpc_use_device(h)) return s;
    if (int s = mpc_flush_pending(h)) return s;          /* a pending finish reads the problem records this may rewrite */
    HIP_TRY(hipStreamSynchronize(h->stream));
    double *h_io = (double *)h->h_hist_io, *d_io = (double *)h->d_hist_io;
    memcpy(h_io, h_actual, sizeof(double) * 3 * n);
    HIP_TRY(hipMemcpyAsync(d_io, h_io, sizeof(double) * 3 * n, hipMemcpyHostToDevice, h->stream));
    double *d_out = d_io + 3 * (size_t)n;
    hipLaunchKernelGGL(estimate_currents_resident_kernel, dim3((n + 63) / 64), dim3(64), 0, h->stream, (const double *)h->d_hist_ring,
                       (const HistoryState *)h->d_hist_state, d_io, n, d_out, apply ? h->d_probs : (DevProblem *)nullptr,
                       h->d_models, (uint32_t)h->models.size());
    HIP_TRY(hipGetLastError());
    if (h_currents_out) HIP_TRY(hipMemcpyAsync(h_io + 3 * (size_t)n, d_out, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h_currents_out) memcpy(h_currents_out, h_io + 3 * (size_t)n, sizeof(double) * 2 * n);
    if (apply) h->ran = false;
    return MPC_OK;
}
