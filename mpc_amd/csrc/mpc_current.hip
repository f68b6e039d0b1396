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

__global__ __launch_bounds__(64) void estimate_currents_kernel(const mpc_history_sample *__restrict__ samples,
                                                               const uint32_t *__restrict__ counts, uint32_t stride,
                                                               const double *__restrict__ actual, uint32_t n,
                                                               double *__restrict__ currents, DevProblem *__restrict__ probs,
                                                               const Model *__restrict__ models, uint32_t n_models)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const Model M = models[p % n_models];            /* vessel p's hull (mpc_hip_set_model) */
    const mpc_history_sample *hist = samples + (size_t)p * stride;
    const int cnt = (int)counts[p];
    const double ax = actual[3 * p], ay = actual[3 * p + 1], at = actual[3 * p + 2];
    double cx = 0.0, cy = 0.0;
    if (cnt >= MPC_CURRENT_MIN_SAMPLES) {                                                   /* :18 */
        Reckoned e{Veh{hist[0].x, hist[0].y, hist[0].heading, hist[0].speed, rpm_from_speed(hist[0].speed, M), 0.0, 0.0},
                   hist[0].time};                                                           /* :19 */
        const double span = at - e.time;
        if (span > 0) {                                                                     /* :21 */
            for (int i = 0; i < cnt; i++) {                                                 /* :22 */
                const mpc_history_sample s = hist[i];
                if (s.time >= at) { reckon(e, s.rudder, s.throttle, at - e.time, M); break; }  /* :23-27 */
                reckon(e, s.rudder, s.throttle, s.time - e.time, M);                          /* :30 */
            }
            if (e.time < at) reckon(e, hist[cnt - 1].rudder, hist[cnt - 1].throttle, at - e.time, M);   /* :34-37 */
            cx = (ax - e.s.x) / span;                                                       /* :39-40 */
            cy = (ay - e.s.y) / span;
        }
    }
    currents[2 * p] = cx; currents[2 * p + 1] = cy;
    if (probs) {                     /* the fields mpc_pack_problem derives from the current, same expressions */
        DevProblem &P = probs[p];
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
